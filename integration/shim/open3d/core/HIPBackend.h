// core/HIPBackend.h -- the ONE new header of the reference-side binding
// (INTEGRATION.md section 0): status -> LogError, the per-thread stream, dtype
// codes. In a real Open3D HIP build `HipStream()` returns
// core::hip::GetStream(); here (type-check build) it is the null stream.
#pragma once
#include "o3d_mi355x.h"
#include "open3d/core/Dtype.h"
#include "open3d/utility/Logging.h"

#define O3DMI_CALL(expr)                                                  \
    do {                                                                  \
        int st_ = (expr);                                                 \
        if (st_ != O3DMI_OK) utility::LogError("{}", o3dmi_last_error()); \
    } while (0)

namespace open3d {
namespace core {
inline void* HipStream() { return nullptr; }
inline int ToO3dmi(const Dtype& d) {
    if (d == Float32) return O3DMI_F32;
    if (d == Float64) return O3DMI_F64;
    if (d == UInt16) return O3DMI_U16;
    if (d == UInt8) return O3DMI_U8;
    if (d == Int32) return O3DMI_I32;
    if (d == Int64) return O3DMI_I64;
    utility::LogError("Unsupported dtype {}", d.ToString());
    return -1;
}
}  // namespace core
}  // namespace open3d
