// Stand-in for open3d/core/CUDAUtils.h (non-CUDA branch, CUDAUtils.h:40-52).
#pragma once
#define OPEN3D_FORCE_INLINE inline
#define OPEN3D_HOST_DEVICE
#define OPEN3D_DEVICE
#define OPEN3D_ASSERT_HOST_DEVICE_LAMBDA(type)
#define OPEN3D_CUDA_CHECK(err)
#define OPEN3D_GET_LAST_CUDA_ERROR(message)
#define CUDA_CALL(cuda_function, ...) \
    utility::LogError("Not built with CUDA, cannot call " #cuda_function);
#include "open3d/utility/Logging.h"
#include "open3d/core/Device.h"
namespace open3d {
namespace core {
namespace cuda {
inline void Synchronize() {}
inline void Synchronize(const Device&) {}
}  // namespace cuda
}  // namespace core
}  // namespace open3d
