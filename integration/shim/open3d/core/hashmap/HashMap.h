// Stand-in for open3d/core/hashmap/HashMap.h used only to TYPE-CHECK
// integration/hip_backend.cpp: the reference's real DeviceHashBackend
// interface (core/hashmap/DeviceHashBackend.h:20-107, included from where it
// lies) plus the two HashMap members the dispatchers use.
#pragma once
#include <memory>
#include "open3d/core/Tensor.h"
#include "open3d/core/hashmap/DeviceHashBackend.h"
namespace open3d {
namespace core {
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
