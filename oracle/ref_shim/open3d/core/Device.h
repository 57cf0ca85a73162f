// Stand-in for open3d/core/Device.h: only CPU:0 exists here.
#pragma once
#include <string>
namespace open3d {
namespace core {
class Device {
public:
    enum class DeviceType { CPU = 0, CUDA = 1, SYCL = 2 };
    Device() = default;
    explicit Device(const std::string&) {}
    Device(const char*) {}
    Device(DeviceType, int) {}
    bool operator==(const Device&) const { return true; }
    bool operator!=(const Device&) const { return false; }
    bool IsCPU() const { return true; }
    bool IsCUDA() const { return false; }
    bool IsSYCL() const { return false; }
    DeviceType GetType() const { return DeviceType::CPU; }
    int GetID() const { return 0; }
    std::string ToString() const { return "CPU:0"; }
};
}  // namespace core
}  // namespace open3d
