// TSDF integration for VoxelBlockGrid on MI355X.
//
//   o3dmi_vbg_integrate <- IntegrateCUDA<input_depth_t,input_color_t,tsdf_t,
//                          weight_t,color_t> (t/geometry/kernel/
//                          VoxelBlockGridImpl.h:151-308; dtype dispatch
//                          t/geometry/kernel/VoxelBlockGrid.cpp:107-232)
//
// The per-voxel arithmetic is the reference's, statement for statement (see
// the numbered comments); what is new is the mapping onto the machine:
//
//  * one 256-thread workgroup owns one voxel block at a time (persistent grid,
//    grid-stride over the active-block list whose length lives on the device);
//  * a lane owns 4 x-consecutive voxels ("quad"): tsdf moves as one 16-byte
//    access per lane (1 KiB per wave instruction), u16 weight as 8 bytes,
//    u16 colour as 24 bytes (three 8-byte accesses, all bytes of every cache
//    line consumed by the same wave);
//  * the block's buffer index and key are wave-uniform (scalar registers);
//    row-constant products of the rigid transform are hoisted per (y,z) row;
//  * voxels that fail the frustum / truncation test neither load nor store
//    state; a quad is written back only if one of its voxels changed.
//
// HBM-bound: 24 B/voxel (u16 grid + colour) or 40 B/voxel (f32 grid) of
// read+write per updated voxel; the depth/colour images (614 KB + 921 KB at
// VGA) stay resident in the 4 MiB per-XCD L2.

#include "common.h"

namespace o3dmi {
namespace {

struct IntegrateParams {
    Camera cam;        // depth intrinsics + extrinsic, scale = voxel_size
    Camera color_cam;  // colour intrinsics, identity extrinsic
    int depth_rows, depth_cols, color_rows, color_cols;
    int resolution;
    float sdf_trunc, depth_scale, depth_max;
    float color_multiplier;
};

template <typename T, int N, int A>
struct alignas(A) Vec {
    T v[N];
};

// First half of the reference's lambda (VoxelBlockGridImpl.h:220-267): voxel
// -> camera -> pixel -> normalised SDF. Needs no voxel state. Returns false
// when the voxel is skipped.
template <typename input_depth_t>
__device__ __forceinline__ bool Associate(
        const IntegrateParams& p, const input_depth_t* __restrict__ depth,
        int x, int y, int z, int& ui, int& vi, float& sdf) {
    float xc, yc, zc, u, v;
    p.cam.RigidTransform((float)x, (float)y, (float)z, xc, yc, zc);
    p.cam.Project(xc, yc, zc, u, v);
    if (!InBoundary2D(u, v, p.depth_rows, p.depth_cols)) return false;

    ui = (int)u;
    vi = (int)v;

    float d = (float)depth[(int64_t)vi * p.depth_cols + ui] / p.depth_scale;
    sdf = d - zc;
    if (d <= 0 || d > p.depth_max || zc <= 0 || sdf < -p.sdf_trunc)
        return false;
    sdf = sdf < p.sdf_trunc ? sdf : p.sdf_trunc;
    sdf /= p.sdf_trunc;
    return true;
}

// Second half (VoxelBlockGridImpl.h:269-302): running averages.
template <typename input_color_t, typename weight_t, typename color_t,
          bool kColor>
__device__ __forceinline__ void Update(const IntegrateParams& p,
                                       const input_color_t* __restrict__ color,
                                       int ui, int vi, float sdf, float& tsdf,
                                       weight_t& wgt, color_t* col) {
    // `*weight_ptr + 1` is int arithmetic for a u16 weight.
    float inv_wsum;
    if constexpr (sizeof(weight_t) == 2)
        inv_wsum = 1.0f / (float)((int)wgt + 1);
    else
        inv_wsum = 1.0f / (wgt + 1);
    float weight = (float)wgt;
    tsdf = (weight * tsdf + sdf) * inv_wsum;

    if constexpr (kColor) {
        // Unproject with the depth intrinsics, re-project with the colour
        // intrinsics (identity extrinsic, scale 1).
        float xx, yy, zz, uf, vf;
        p.cam.Unproject((float)ui, (float)vi, 1.0f, xx, yy, zz);
        p.color_cam.Project(xx, yy, zz, uf, vf);
        if (InBoundary2D(uf, vf, p.color_rows, p.color_cols)) {
            int uc = (int)roundf(uf);
            int vc = (int)roundf(vf);
            const input_color_t* in =
                    color + ((int64_t)vc * p.color_cols + uc) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                col[i] = (color_t)((weight * (float)col[i] +
                                    (float)in[i] * p.color_multiplier) *
                                   inv_wsum);
            }
        }
    }
    // float -> weight_t conversion truncates for u16.
    wgt = (weight_t)(weight + 1);
}

// Vectorised kernel: resolution % 4 == 0 (8, 16, 32 ...).
template <typename input_depth_t, typename input_color_t, typename weight_t,
          typename color_t, bool kColor>
__global__ void __launch_bounds__(256)
IntegrateQuadKernel(IntegrateParams p, const input_depth_t* __restrict__ depth,
                    const input_color_t* __restrict__ color,
                    const int* __restrict__ indices, int64_t n_indices,
                    const int* __restrict__ n_indices_dev,
                    const int* __restrict__ block_keys,
                    float* __restrict__ tsdf_base,
                    weight_t* __restrict__ weight_base,
                    color_t* __restrict__ color_base) {
    using TVec = Vec<float, 4, 16>;
    using WVec = Vec<weight_t, 4, 4 * sizeof(weight_t)>;
    using CVec = Vec<color_t, 12, 4 * sizeof(color_t)>;
    if (n_indices_dev) {
        int64_t live = *n_indices_dev;
        n_indices = live < n_indices ? live : n_indices;
    }
    const int res = p.resolution;
    const int res3 = res * res * res;
    const int quads_per_row = res >> 2;
    const int n_quads = res3 >> 2;

    for (int64_t b = blockIdx.x; b < n_indices; b += gridDim.x) {
        // Wave-uniform block header.
        const int block_idx = __builtin_amdgcn_readfirstlane(indices[b]);
        const int xb = __builtin_amdgcn_readfirstlane(
                block_keys[3 * (int64_t)block_idx + 0]);
        const int yb = __builtin_amdgcn_readfirstlane(
                block_keys[3 * (int64_t)block_idx + 1]);
        const int zb = __builtin_amdgcn_readfirstlane(
                block_keys[3 * (int64_t)block_idx + 2]);
        const int64_t block_base = (int64_t)block_idx * res3;

        for (int q = threadIdx.x; q < n_quads; q += blockDim.x) {
            // voxel_idx = z*res*res + y*res + x (GeometryIndexer.h:270-278)
            const int qx = q % quads_per_row;
            const int row = q / quads_per_row;
            const int yv = row % res;
            const int zv = row / res;
            const int x0 = xb * res + (qx << 2);
            const int y = yb * res + yv;
            const int z = zb * res + zv;
            const int64_t lin0 = block_base + ((int64_t)q << 2);

            // Association needs no voxel state: do it for the 4 voxels first
            // so that quads entirely outside the frustum / truncation band
            // cost no HBM traffic at all.
            int ui[4], vi[4];
            float sdf[4];
            bool ok[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[j] = Associate(p, depth, x0 + j, y, z, ui[j], vi[j], sdf[j]);
                any |= ok[j];
            }
            if (!any) continue;

            TVec t4 = *reinterpret_cast<const TVec*>(tsdf_base + lin0);
            WVec w4 = *reinterpret_cast<const WVec*>(weight_base + lin0);
            CVec c12;
            if constexpr (kColor)
                c12 = *reinterpret_cast<const CVec*>(color_base + 3 * lin0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ok[j])
                    Update<input_color_t, weight_t, color_t, kColor>(
                            p, color, ui[j], vi[j], sdf[j], t4.v[j], w4.v[j],
                            &c12.v[3 * j]);
            }
            *reinterpret_cast<TVec*>(tsdf_base + lin0) = t4;
            *reinterpret_cast<WVec*>(weight_base + lin0) = w4;
            if constexpr (kColor)
                *reinterpret_cast<CVec*>(color_base + 3 * lin0) = c12;
        }
    }
}

// Generic kernel: any resolution, one voxel per lane per step.
template <typename input_depth_t, typename input_color_t, typename weight_t,
          typename color_t, bool kColor>
__global__ void __launch_bounds__(256)
IntegrateScalarKernel(IntegrateParams p,
                      const input_depth_t* __restrict__ depth,
                      const input_color_t* __restrict__ color,
                      const int* __restrict__ indices, int64_t n_indices,
                      const int* __restrict__ n_indices_dev,
                      const int* __restrict__ block_keys,
                      float* __restrict__ tsdf_base,
                      weight_t* __restrict__ weight_base,
                      color_t* __restrict__ color_base) {
    if (n_indices_dev) {
        int64_t live = *n_indices_dev;
        n_indices = live < n_indices ? live : n_indices;
    }
    const int res = p.resolution;
    const int res2 = res * res;
    const int res3 = res2 * res;
    for (int64_t b = blockIdx.x; b < n_indices; b += gridDim.x) {
        const int block_idx = indices[b];
        const int xb = block_keys[3 * (int64_t)block_idx + 0];
        const int yb = block_keys[3 * (int64_t)block_idx + 1];
        const int zb = block_keys[3 * (int64_t)block_idx + 2];
        for (int voxel_idx = threadIdx.x; voxel_idx < res3;
             voxel_idx += blockDim.x) {
            int xv = voxel_idx % res;
            int yv = (voxel_idx / res) % res;
            int zv = voxel_idx / res2;
            int64_t lin = (int64_t)block_idx * res3 + voxel_idx;
            float t = tsdf_base[lin];
            weight_t w = weight_base[lin];
            color_t col[3] = {0, 0, 0};
            if constexpr (kColor) {
                col[0] = color_base[3 * lin + 0];
                col[1] = color_base[3 * lin + 1];
                col[2] = color_base[3 * lin + 2];
            }
            int ui, vi;
            float sdf;
            bool upd = Associate(p, depth, xb * res + xv, yb * res + yv,
                                 zb * res + zv, ui, vi, sdf);
            if (upd)
                Update<input_color_t, weight_t, color_t, kColor>(
                        p, color, ui, vi, sdf, t, w, col);
            if (upd) {
                tsdf_base[lin] = t;
                weight_base[lin] = w;
                if constexpr (kColor) {
                    color_base[3 * lin + 0] = col[0];
                    color_base[3 * lin + 1] = col[1];
                    color_base[3 * lin + 2] = col[2];
                }
            }
        }
    }
}

template <typename DT, typename CT, typename WT, typename VT>
int Launch(const IntegrateParams& p, const void* depth, const void* color,
           const int* indices, int64_t n, const int* n_dev, const int* keys,
           float* tsdf, void* weight, void* cbuf, bool do_color,
           hipStream_t s) {
    // Persistent grid: up to 8 workgroups per CU keeps 32 waves/CU resident.
    int64_t g = n < (int64_t)kCUs * 8 ? n : (int64_t)kCUs * 8;
    if (g < 1) g = 1;
    dim3 grid((unsigned)g), block(256);
    bool quad = (p.resolution % 4) == 0;
#define O3DMI_LAUNCH(KERNEL, COLOR)                                          \
    hipLaunchKernelGGL((KERNEL<DT, CT, WT, VT, COLOR>), grid, block, 0, s, p, \
                       (const DT*)depth, (const CT*)color, indices, n, n_dev, \
                       keys, tsdf, (WT*)weight, (VT*)cbuf)
    if (quad) {
        if (do_color) O3DMI_LAUNCH(IntegrateQuadKernel, true);
        else O3DMI_LAUNCH(IntegrateQuadKernel, false);
    } else {
        if (do_color) O3DMI_LAUNCH(IntegrateScalarKernel, true);
        else O3DMI_LAUNCH(IntegrateScalarKernel, false);
    }
#undef O3DMI_LAUNCH
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace

}  // namespace o3dmi

using namespace o3dmi;

extern "C" int o3dmi_vbg_integrate(
        const void* depth_dev, int depth_rows, int depth_cols,
        const void* color_dev, int color_rows, int color_cols, int input_dtype,
        const int32_t* indices_dev, int64_t n_indices,
        const int32_t* n_indices_dev, const int32_t* block_keys_dev,
        float* tsdf_dev, void* weight_dev, void* color_buf_dev, int grid_dtype,
        const double* depth_intrinsic, const double* color_intrinsic,
        const double* extrinsic, int resolution, float voxel_size,
        float sdf_trunc, float depth_scale, float depth_max,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(depth_dev && indices_dev && block_keys_dev && tsdf_dev &&
                          weight_dev && depth_intrinsic && extrinsic,
                  "null argument");
    // "TSDF and/or weight not allocated in blocks" is the caller's check
    // (VoxelBlockGridImpl.h:193-198).
    O3DMI_REQUIRE(input_dtype == O3DMI_U16 || input_dtype == O3DMI_F32,
                  "Unsupported input data type combination. Expected (float, "
                  "float) or (uint16, uint8)");
    O3DMI_REQUIRE(grid_dtype == O3DMI_U16 || grid_dtype == O3DMI_F32,
                  "Unsupported value data type combination. Expected (float, "
                  "float) or (uint16, uint16)");
    O3DMI_REQUIRE(resolution > 0 && n_indices >= 0, "bad resolution / count");
    if (n_indices == 0) return O3DMI_OK;

    static const double eye4[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                    0, 0, 1, 0, 0, 0, 0, 1};
    IntegrateParams p;
    p.cam = Camera::Make(depth_intrinsic, extrinsic, voxel_size);
    p.color_cam = Camera::Make(color_intrinsic ? color_intrinsic
                                               : depth_intrinsic,
                               eye4, 1.0f);
    p.depth_rows = depth_rows;
    p.depth_cols = depth_cols;
    p.color_rows = color_rows;
    p.color_cols = color_cols;
    p.resolution = resolution;
    p.sdf_trunc = sdf_trunc;
    p.depth_scale = depth_scale;
    p.depth_max = depth_max;
    // VoxelBlockGridImpl.h:208-216: Float32 colours in [0,1] -> [0,255].
    p.color_multiplier = (input_dtype == O3DMI_F32) ? 255.0f : 1.0f;

    bool do_color = color_buf_dev != nullptr && color_dev != nullptr &&
                    (int64_t)color_rows * color_cols > 0;
    hipStream_t s = (hipStream_t)stream;
    // Instantiations mirror VoxelBlockGridCPU.cpp:212-218.
    if (input_dtype == O3DMI_U16 && grid_dtype == O3DMI_U16)
        return Launch<uint16_t, uint8_t, uint16_t, uint16_t>(
                p, depth_dev, color_dev, indices_dev, n_indices, n_indices_dev,
                block_keys_dev, tsdf_dev, weight_dev, color_buf_dev, do_color,
                s);
    if (input_dtype == O3DMI_U16 && grid_dtype == O3DMI_F32)
        return Launch<uint16_t, uint8_t, float, float>(
                p, depth_dev, color_dev, indices_dev, n_indices, n_indices_dev,
                block_keys_dev, tsdf_dev, weight_dev, color_buf_dev, do_color,
                s);
    if (input_dtype == O3DMI_F32 && grid_dtype == O3DMI_U16)
        return Launch<float, float, uint16_t, uint16_t>(
                p, depth_dev, color_dev, indices_dev, n_indices, n_indices_dev,
                block_keys_dev, tsdf_dev, weight_dev, color_buf_dev, do_color,
                s);
    return Launch<float, float, float, float>(
            p, depth_dev, color_dev, indices_dev, n_indices, n_indices_dev,
            block_keys_dev, tsdf_dev, weight_dev, color_buf_dev, do_color, s);
}
