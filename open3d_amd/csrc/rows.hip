// Row gather / scatter by index on the device: the Tensor::IndexGet /
// IndexSet of whole value rows that HashMap::Reserve (core/hashmap/
// HashMap.cpp:54-76) and VoxelBlockGrid::Save (t/geometry/VoxelBlockGrid.cpp:
// 505-516) perform. One workgroup streams one row (a 16^3 block row is 8 to
// 48 KiB) with 16-byte accesses when the row size allows it. HBM bound.

#include "common.h"

namespace o3dmi {
namespace {

template <typename V, bool GATHER>
__global__ void RowsKernel(const uint8_t* __restrict__ src,
                           uint8_t* __restrict__ dst,
                           const int* __restrict__ indices, int64_t n,
                           int64_t row_units) {
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const int64_t b = indices[r];
        const V* s = (const V*)src + (GATHER ? b : r) * row_units;
        V* d = (V*)dst + (GATHER ? r : b) * row_units;
        for (int64_t i = threadIdx.x; i < row_units; i += blockDim.x) d[i] = s[i];
    }
}

template <bool GATHER>
int Rows(const void* src, void* dst, const int* indices, int64_t n,
         int64_t row_bytes, hipStream_t s) {
    if (n <= 0 || row_bytes <= 0) return O3DMI_OK;
    O3DMI_REQUIRE(src && dst && indices, "null argument");
    const int grid = (int)(n < 65536 ? n : 65536);
    const bool al16 = row_bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 &&
                      ((uintptr_t)dst % 16) == 0;
    const bool al4 = row_bytes % 4 == 0 && ((uintptr_t)src % 4) == 0 &&
                     ((uintptr_t)dst % 4) == 0;
    if (al16)
        hipLaunchKernelGGL((RowsKernel<uint4, GATHER>), dim3(grid), dim3(256), 0,
                           s, (const uint8_t*)src, (uint8_t*)dst, indices, n,
                           row_bytes / 16);
    else if (al4)
        hipLaunchKernelGGL((RowsKernel<uint32_t, GATHER>), dim3(grid), dim3(256),
                           0, s, (const uint8_t*)src, (uint8_t*)dst, indices, n,
                           row_bytes / 4);
    else
        hipLaunchKernelGGL((RowsKernel<uint8_t, GATHER>), dim3(grid), dim3(256),
                           0, s, (const uint8_t*)src, (uint8_t*)dst, indices, n,
                           row_bytes);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace

// dst[r] = src[indices[r]]
int GatherRows(const void* src, const int* indices_dev, int64_t n,
               int64_t row_bytes, void* dst, hipStream_t s) {
    return Rows<true>(src, dst, indices_dev, n, row_bytes, s);
}
// dst[indices[r]] = src[r]
int ScatterRows(const void* src, const int* indices_dev, int64_t n,
                int64_t row_bytes, void* dst, hipStream_t s) {
    return Rows<false>(src, dst, indices_dev, n, row_bytes, s);
}

}  // namespace o3dmi
