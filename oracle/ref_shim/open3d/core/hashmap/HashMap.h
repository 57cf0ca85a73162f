// Stand-in for open3d/core/hashmap/HashMap.h: just what RayCastCPU needs --
// a handle that yields the CPU backend (VoxelBlockGridImpl.h:606-637).
#pragma once
#include <memory>
#include "open3d/core/Tensor.h"
namespace open3d {
namespace core {
using buf_index_t = uint32_t;
class DeviceHashBackend {
public:
    virtual ~DeviceHashBackend() = default;
};
enum class HashBackendType { Slab, StdGPU, TBB, Default };
class HashMap {
public:
    explicit HashMap(std::shared_ptr<DeviceHashBackend> b)
        : backend_(std::move(b)) {}
    std::shared_ptr<DeviceHashBackend> GetDeviceHashBackend() const {
        return backend_;
    }
    Device GetDevice() const { return Device("CPU:0"); }
private:
    std::shared_ptr<DeviceHashBackend> backend_;
};
}  // namespace core
}  // namespace open3d
