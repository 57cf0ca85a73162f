// Voxel enumeration helpers of VoxelBlockGrid on MI355X -- the "secondary
// entry points" of SURVEY.md's kernel inventory (K14) and the tensor
// arithmetic of two VoxelBlockGrid methods, each as one coalesced launch:
//   GetVoxelCoordinatesAndFlattenedIndicesCUDA
//       (cpp/open3d/t/geometry/kernel/VoxelBlockGridImpl.h:43-92)
//   VoxelBlockGrid::GetVoxelIndices   (t/geometry/VoxelBlockGrid.cpp:145-178:
//       an Arange and six element-wise tensor ops upstream)
//   VoxelBlockGrid::GetVoxelCoordinates (VoxelBlockGrid.cpp:130-143: IndexGet,
//       transpose, cast, three in-place adds)
// Pure index arithmetic, HBM-write bound: 8-32 B per voxel out, 12 B per block
// in. One thread per voxel, a block's voxels in consecutive lanes (x fastest,
// the layout of the value tensors: GeometryIndexer.h:244-249).
// The flattened index is formed in 64 bits: upstream's `index_t` is int and
// `block_idx * resolution^3` overflows past 524 287 blocks of 16^3 (SURVEY
// 9.5); below that the values are the same.

#include "common.h"

namespace o3dmi {
namespace {

__global__ void VoxelCoordsFlatKernel(const int* __restrict__ buf_indices,
                                      const int* __restrict__ block_keys,
                                      int64_t n, int res, float voxel_size,
                                      float* __restrict__ voxel_coords,
                                      int64_t* __restrict__ flattened) {
    const int res2 = res * res, res3 = res2 * res;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int block_idx = buf_indices[w / res3];
        const int voxel_idx = (int)(w % res3);
        const int xb = block_keys[3 * (int64_t)block_idx + 0];
        const int yb = block_keys[3 * (int64_t)block_idx + 1];
        const int zb = block_keys[3 * (int64_t)block_idx + 2];
        // WorkloadToCoord of a {res, res, res} indexer: x fastest
        const int zv = voxel_idx / res2;
        const int yv = (voxel_idx - zv * res2) / res;
        const int xv = voxel_idx - zv * res2 - yv * res;
        if (flattened) flattened[w] = (int64_t)block_idx * res3 + voxel_idx;
        if (voxel_coords) {
            voxel_coords[3 * w + 0] = (xb * res + xv) * voxel_size;
            voxel_coords[3 * w + 1] = (yb * res + yv) * voxel_size;
            voxel_coords[3 * w + 2] = (zb * res + zv) * voxel_size;
        }
    }
}

// rows of the {4, n} result: buffer index, x, y, z of voxel w
__global__ void VoxelIndicesKernel(const int* __restrict__ buf_indices,
                                   int64_t n, int res,
                                   int64_t* __restrict__ out) {
    const int64_t res2 = (int64_t)res * res, res3 = res2 * res;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = w / res3;
        int64_t rem = w - b * res3;
        const int64_t z = rem / res2;
        rem -= z * res2;
        const int64_t y = rem / res;
        const int64_t x = rem - y * res;
        out[w] = (int64_t)buf_indices[b];
        out[n + w] = x;
        out[2 * n + w] = y;
        out[3 * n + w] = z;
    }
}

// {4, n} voxel indices -> {3, n} voxel coordinates (in voxels, Int64)
__global__ void VoxelCoordinatesKernel(const int64_t* __restrict__ vi,
                                       int64_t n,
                                       const int* __restrict__ block_keys,
                                       int64_t capacity, int res,
                                       int64_t* __restrict__ out,
                                       int* __restrict__ err) {
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = vi[w];
        if (b < 0 || b >= capacity) {  // upstream's IndexGet throws
            if (err) atomicOr(err, 1);
            out[w] = out[n + w] = out[2 * n + w] = 0;
            continue;
        }
        out[w] = (int64_t)block_keys[3 * b + 0] * res + vi[n + w];
        out[n + w] = (int64_t)block_keys[3 * b + 1] * res + vi[2 * n + w];
        out[2 * n + w] = (int64_t)block_keys[3 * b + 2] * res + vi[3 * n + w];
    }
}

}  // namespace
}  // namespace o3dmi

using namespace o3dmi;

extern "C" {

int o3dmi_vbg_voxel_coordinates_and_flattened_indices(
        const int32_t* buf_indices_dev, int64_t n_blocks,
        const int32_t* block_keys_dev, int resolution, float voxel_size,
        float* voxel_coords_dev, int64_t* flattened_indices_dev,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n_blocks >= 0 && resolution > 0 && resolution <= 64,
                  "bad block count / resolution");
    if (n_blocks == 0) return O3DMI_OK;
    O3DMI_REQUIRE(buf_indices_dev && block_keys_dev &&
                          (voxel_coords_dev || flattened_indices_dev),
                  "null argument");
    const int64_t n = n_blocks * resolution * resolution * resolution;
    hipLaunchKernelGGL(VoxelCoordsFlatKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, buf_indices_dev,
                       block_keys_dev, n, resolution, voxel_size,
                       voxel_coords_dev, flattened_indices_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_voxel_indices(const int32_t* buf_indices_dev, int64_t n_blocks,
                            int resolution, int64_t* voxel_indices_dev,
                            o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n_blocks >= 0 && resolution > 0 && resolution <= 64,
                  "bad block count / resolution");
    if (n_blocks == 0) return O3DMI_OK;
    O3DMI_REQUIRE(buf_indices_dev && voxel_indices_dev, "null argument");
    const int64_t n = n_blocks * resolution * resolution * resolution;
    hipLaunchKernelGGL(VoxelIndicesKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, buf_indices_dev, n,
                       resolution, voxel_indices_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_voxel_coordinates(const int64_t* voxel_indices_dev, int64_t n,
                                const int32_t* block_keys_dev,
                                int64_t key_capacity, int resolution,
                                int64_t* voxel_coords_dev, int32_t* err_dev,
                                o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n >= 0 && resolution > 0 && key_capacity >= 0,
                  "bad voxel count / resolution");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(voxel_indices_dev && block_keys_dev && voxel_coords_dev,
                  "null argument");
    hipLaunchKernelGGL(VoxelCoordinatesKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, voxel_indices_dev,
                       n, block_keys_dev, key_capacity, resolution,
                       voxel_coords_dev, err_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // extern "C"
