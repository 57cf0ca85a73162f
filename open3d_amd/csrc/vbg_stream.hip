// Frame-stream fast path of VoxelBlockGrid integration on MI355X (see
// stream_path.h for the contract and the reference lines it re-cuts).
//
//   front role      <- DepthTouchCPU (t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201)
//                      + HashMap::Activate (core/hashmap/HashMap.cpp:166-181)
//                      + the per-pixel part of IntegrateCPU's lambda
//                        (t/geometry/kernel/VoxelBlockGridImpl.h:258-262,277-289)
//   integrate role  <- the per-voxel part of IntegrateCPU's lambda
//                        (VoxelBlockGridImpl.h:220-257,263-303)
//
// FrameStepKernel runs the front role of frame k+1 and the integrate role of
// frame k in ONE launch: the front role is a latency chain of hash atomics on
// ~75 workgroups, the integrate role a bandwidth-bound sweep over ~4x10^3
// work items; side by side they cost max() instead of sum() and a frame costs
// a single kernel launch.

#include "common.h"
#include "stream_path.h"
#include "touch_device.h"

namespace o3dmi {
namespace {

template <typename T, int N, int A>
struct alignas(A) Vec {
    T v[N];
};

// Frame-stream front end (stream_path.h). Workgroups [0, n_touch_wg) run the
// fused touch+activate of TouchActivateKernel, emitting {slot, key} entries;
// the remaining workgroups run the per-pixel prepare pass. The two roles share
// one launch so that the latency-bound hash work (75 workgroups at VGA) and the
// streaming prepare pass (all other CUs) overlap.
struct PrepParams {
    Camera color_cam;  // colour intrinsics, identity extrinsic, scale 1
    int color_rows, color_cols;
    bool with_color;
};

struct FrontParams {
    HashView hv;
    TouchParams p;
    PrepParams pp;
    const uint16_t* depth;
    const uint8_t* color;
    PixelRec* recs;
    FrameBlock* list;
    int64_t list_capacity;
    int* out_count;
    int frame_stamp;
    int n_touch_wg, n_prep_wg;
};

// `wg` = index of this workgroup within the front role, [0, n_touch_wg +
// n_prep_wg).
__device__ __forceinline__ void FrontRole(const FrontParams& fp, int wg) {
    const HashView& hv = fp.hv;
    const TouchParams& p = fp.p;
    const PrepParams& pp = fp.pp;
    const uint16_t* __restrict__ depth = fp.depth;
    const uint8_t* __restrict__ color = fp.color;
    PixelRec* __restrict__ recs = fp.recs;
    FrameBlock* __restrict__ list = fp.list;
    const int64_t list_capacity = fp.list_capacity;
    int* __restrict__ out_count = fp.out_count;
    const int frame_stamp = fp.frame_stamp;
    const int n_touch_wg = fp.n_touch_wg;
    if (wg < n_touch_wg) {
        int n = p.rows_strided * p.cols_strided;
        int n_padded = ((n + 63) / 64) * 64;
        for (int w = wg * blockDim.x + threadIdx.x; w < n_padded;
             w += n_touch_wg * blockDim.x) {
            int xb[4], yb[4], zb[4];
            bool valid = (w < n) && RayCandidates(p, depth, w, xb, yb, zb);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bool ok = valid;
                if (ok && s > 0 && xb[s] == xb[s - 1] && yb[s] == yb[s - 1] &&
                    zb[s] == zb[s - 1])
                    ok = false;
                if (ok && !KeyInRange(xb[s], yb[s], zb[s])) {
                    atomicOr(&hv.counters[1], kErrKeyRange);
                    ok = false;
                }
                unsigned long long k = ok ? PackKey(xb[s], yb[s], zb[s]) : 0ull;
                if (WaveLeaderForKey(k, ok)) {
                    unsigned slot;
                    InsertKey<true>(hv, xb[s], yb[s], zb[s], slot);
                    int old = atomicExch(&hv.slot_stamp[slot], frame_stamp);
                    if (old != frame_stamp) {
                        int o = atomicAdd(out_count, 1);
                        if (o < list_capacity) {
                            FrameBlock fb;
                            fb.slot = (int)slot;
                            fb.x = xb[s];
                            fb.y = yb[s];
                            fb.z = zb[s];
                            list[o] = fb;
                        } else {
                            atomicOr(&hv.counters[1], kErrCapacity);
                        }
                    }
                }
            }
        }
        return;
    }
    // Prepare pass: the voxel-independent sub-expressions of the integrate
    // lambda (VoxelBlockGridImpl.h:258-262 depth, :277-289 colour pixel).
    const int n_px = p.rows * p.cols;
    const int n_wg = fp.n_prep_wg;
    for (int i = (wg - n_touch_wg) * blockDim.x + threadIdx.x;
         i < n_px; i += n_wg * blockDim.x) {
        const int vi = i / p.cols;
        const int ui = i - vi * p.cols;
        PixelRec r;
        r.d = (float)depth[i] / p.depth_scale;
        r.rgba = 0u;
        if (pp.with_color) {
            float x, y, z, uf, vf;
            p.cam.Unproject((float)ui, (float)vi, 1.0f, x, y, z);
            pp.color_cam.Project(x, y, z, uf, vf);
            if (InBoundary2D(uf, vf, pp.color_rows, pp.color_cols)) {
                int uc = (int)roundf(uf);
                int vc = (int)roundf(vf);
                const uint8_t* in =
                        color + ((int64_t)vc * pp.color_cols + uc) * 3;
                r.rgba = (unsigned)in[0] | ((unsigned)in[1] << 8) |
                         ((unsigned)in[2] << 16) | (1u << 24);
            }
        }
        recs[i] = r;
    }
}

// ---- frame-stream kernel (stream_path.h) -----------------------------------
// Work item = (block of the frame's list, 256-quad part of that block); the
// grid strides over items so that a frame's ~10^3 blocks spread as ~4x10^3
// workgroups over the 256 CUs. Depth and colour come from the prepared
// PixelRec image (one 8-byte gather per voxel).
struct StreamParams {
    Camera cam;  // depth intrinsics + extrinsic, scale = voxel_size
    int rows, cols, resolution;
    float sdf_trunc, depth_max;
};

struct IntegParams {
    StreamParams p;
    const PixelRec* recs;
    const FrameBlock* list;
    const int* count;
    int64_t list_capacity;
    const int* slot_vals;
    const int* hash_counters;
    float* tsdf;
    void* weight;
    void* color;
    int* zero_counter;
    int* size_host;
    int frame_stamp;
    int* prof_count;
};

// `wg` / `n_wg` = index of this workgroup within the integrate role and the
// number of workgroups the role was given.
template <typename weight_t, typename color_t, bool kColor>
__device__ __forceinline__ void IntegrateRole(const IntegParams& ip, int wg,
                                              int n_wg) {
    const StreamParams& p = ip.p;
    const PixelRec* __restrict__ recs = ip.recs;
    const FrameBlock* __restrict__ list = ip.list;
    const int* __restrict__ count = ip.count;
    const int64_t list_capacity = ip.list_capacity;
    const int* __restrict__ slot_vals = ip.slot_vals;
    const int* __restrict__ hash_counters = ip.hash_counters;
    float* __restrict__ tsdf_base = ip.tsdf;
    weight_t* __restrict__ weight_base = (weight_t*)ip.weight;
    color_t* __restrict__ color_base = (color_t*)ip.color;
    int* __restrict__ zero_counter = ip.zero_counter;
    int* __restrict__ size_host = ip.size_host;
    const int frame_stamp = ip.frame_stamp;
    int* __restrict__ prof_count = ip.prof_count;
    using TVec = Vec<float, 4, 16>;
    using WVec = Vec<weight_t, 4, 4 * sizeof(weight_t)>;
    using CVec = Vec<color_t, 12, 4 * sizeof(color_t)>;
    int64_t n_blocks = *count;
    if (n_blocks > list_capacity) n_blocks = list_capacity;

    if (wg == 0 && threadIdx.x == 0) {
        if (zero_counter) *zero_counter = 0;
        if (prof_count) *prof_count = (int)n_blocks;
        if (size_host) {
            // The touch kernel of this frame has completed (stream order), so
            // heap_top is the exact map size after this frame's activation.
            size_host[0] = hash_counters[0];
            size_host[1] = hash_counters[1];
            size_host[2] = (int)n_blocks;
            __hip_atomic_store(&size_host[3], frame_stamp, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    const int res = p.resolution;
    const int res3 = res * res * res;
    const int quads_per_row = res >> 2;
    const int n_quads = res3 >> 2;
    const int parts = (n_quads + 255) >> 8;
    const int64_t n_items = n_blocks * parts;

    for (int64_t item = wg; item < n_items; item += n_wg) {
        const int64_t b = item / parts;
        const int part = (int)(item - b * parts);
        // Wave-uniform block header: {slot, key} -> buffer index.
        const FrameBlock fb = list[b];
        const int slot = __builtin_amdgcn_readfirstlane(fb.slot);
        const int xb = __builtin_amdgcn_readfirstlane(fb.x);
        const int yb = __builtin_amdgcn_readfirstlane(fb.y);
        const int zb = __builtin_amdgcn_readfirstlane(fb.z);
        const int block_idx = __builtin_amdgcn_readfirstlane(slot_vals[slot]);
        const int64_t block_base = (int64_t)block_idx * res3;

        const int q = (part << 8) + threadIdx.x;
        if (q >= n_quads) continue;
        const int qx = q % quads_per_row;
        const int row = q / quads_per_row;
        const int yv = row % res;
        const int zv = row / res;
        const int x0 = xb * res + (qx << 2);
        const int y = yb * res + yv;
        const int z = zb * res + zv;
        const int64_t lin0 = block_base + ((int64_t)q << 2);

        // VoxelBlockGridImpl.h:244-267 with depth taken from the record.
        float sdf[4];
        unsigned rgba[4];
        bool ok[4];
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xc, yc, zc, u, v;
            p.cam.RigidTransform((float)(x0 + j), (float)y, (float)z, xc, yc,
                                 zc);
            p.cam.Project(xc, yc, zc, u, v);
            ok[j] = InBoundary2D(u, v, p.rows, p.cols);
            sdf[j] = 0.f;
            rgba[j] = 0u;
            if (ok[j]) {
                const int ui = (int)u;
                const int vi = (int)v;
                const PixelRec r = recs[(int64_t)vi * p.cols + ui];
                const float d = r.d;
                float sd = d - zc;
                if (d <= 0 || d > p.depth_max || zc <= 0 || sd < -p.sdf_trunc) {
                    ok[j] = false;
                } else {
                    sd = sd < p.sdf_trunc ? sd : p.sdf_trunc;
                    sdf[j] = sd / p.sdf_trunc;
                    rgba[j] = r.rgba;
                }
            }
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
            if (!ok[j]) continue;
            // VoxelBlockGridImpl.h:269-302
            float inv_wsum;
            if constexpr (sizeof(weight_t) == 2)
                inv_wsum = 1.0f / (float)((int)w4.v[j] + 1);
            else
                inv_wsum = 1.0f / (w4.v[j] + 1);
            const float weight = (float)w4.v[j];
            t4.v[j] = (weight * t4.v[j] + sdf[j]) * inv_wsum;
            if constexpr (kColor) {
                if (rgba[j] >> 24) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float in = (float)((rgba[j] >> (8 * i)) & 0xffu);
                        // colour multiplier is 1 for uint8 input
                        c12.v[3 * j + i] = (color_t)(
                                (weight * (float)c12.v[3 * j + i] + in * 1.0f) *
                                inv_wsum);
                    }
                }
            }
            w4.v[j] = (weight_t)(weight + 1);
        }
        *reinterpret_cast<TVec*>(tsdf_base + lin0) = t4;
        *reinterpret_cast<WVec*>(weight_base + lin0) = w4;
        if constexpr (kColor)
            *reinterpret_cast<CVec*>(color_base + 3 * lin0) = c12;
    }
}


template <typename weight_t, typename color_t, bool kColor>
__global__ void __launch_bounds__(256)
FrameStepKernel(FrontParams fp, IntegParams ip, int n_front_wg) {
    const int b = (int)blockIdx.x;
    if (b < n_front_wg) {
        FrontRole(fp, b);
    } else {
        IntegrateRole<weight_t, color_t, kColor>(ip, b - n_front_wg,
                                                 (int)gridDim.x - n_front_wg);
    }
}

}  // namespace

int LaunchFrameStep(o3dmi_hash* bh, const FrameFrontArgs* f,
                    const IntegrateStreamArgs* a, hipStream_t s) {
    O3DMI_REQUIRE(f || a, "nothing to launch");
    FrontParams fp = {};
    IntegParams ip = {};
    int n_front_wg = 0, n_int_wg = 0;
    int grid_dtype = O3DMI_U16;
    bool col = false;
    if (f) {
        fp.hv = bh->view;
        fp.p = MakeTouchParams(f->depth_intrinsic, f->extrinsic, f->rows,
                               f->cols, f->stride, f->resolution,
                               f->voxel_size, f->sdf_trunc, f->depth_scale,
                               f->depth_max);
        static const double eye4[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                        0, 0, 1, 0, 0, 0, 0, 1};
        fp.pp.color_cam = Camera::Make(f->color_intrinsic ? f->color_intrinsic
                                                          : f->depth_intrinsic,
                                       eye4, 1.0f);
        fp.pp.color_rows = f->color_rows;
        fp.pp.color_cols = f->color_cols;
        fp.pp.with_color = f->color != nullptr;
        fp.depth = f->depth;
        fp.color = f->color;
        fp.recs = f->recs;
        fp.list = f->list;
        fp.list_capacity = f->list_capacity;
        fp.out_count = f->count;
        fp.frame_stamp = f->frame_stamp;
        const int n_rays = fp.p.rows_strided * fp.p.cols_strided;
        fp.n_touch_wg = (n_rays + kBlock - 1) / kBlock;
        // 4 pixels per prepare lane
        fp.n_prep_wg = (f->rows * f->cols + kBlock * 4 - 1) / (kBlock * 4);
        if (fp.n_prep_wg < 1) fp.n_prep_wg = 1;
        n_front_wg = fp.n_touch_wg + fp.n_prep_wg;
    }
    if (a) {
        O3DMI_REQUIRE(a->resolution % 4 == 0,
                      "frame-stream path needs block_resolution % 4 == 0");
        ip.p.cam = Camera::Make(a->depth_intrinsic, a->extrinsic,
                                a->voxel_size);
        ip.p.rows = a->rows;
        ip.p.cols = a->cols;
        ip.p.resolution = a->resolution;
        ip.p.sdf_trunc = a->sdf_trunc;
        ip.p.depth_max = a->depth_max;
        ip.recs = a->recs;
        ip.list = a->list;
        ip.count = a->count;
        ip.list_capacity = a->list_capacity;
        ip.slot_vals = bh->view.slot_vals;
        ip.hash_counters = bh->view.counters;
        ip.tsdf = a->tsdf;
        ip.weight = a->weight;
        ip.color = a->color;
        ip.zero_counter = a->zero_counter;
        ip.size_host = a->size_host;
        ip.frame_stamp = a->frame_stamp;
        ip.prof_count = a->prof_count;
        const int n_quads =
                (a->resolution * a->resolution * a->resolution) >> 2;
        const int parts = (n_quads + 255) >> 8;
        // Grid from the expected block count (previous frame + slack); the
        // role strides, so an under-estimate only costs balance.
        int64_t g = ((int64_t)a->grid_hint + (a->grid_hint >> 2) + 64) * parts;
        const int64_t g_max = (int64_t)kCUs * 32;
        if (g > g_max) g = g_max;
        if (g < kCUs) g = kCUs;
        n_int_wg = (int)g;
        grid_dtype = a->grid_dtype;
        col = a->with_color && a->color != nullptr;
    }
    dim3 grid((unsigned)(n_front_wg + n_int_wg)), block(256);
#define O3DMI_LAUNCH_STEP(WT, VT, COLOR)                                      \
    hipLaunchKernelGGL((FrameStepKernel<WT, VT, COLOR>), grid, block, 0, s,   \
                       fp, ip, n_front_wg)
    if (grid_dtype == O3DMI_U16) {
        if (col) O3DMI_LAUNCH_STEP(uint16_t, uint16_t, true);
        else O3DMI_LAUNCH_STEP(uint16_t, uint16_t, false);
    } else {
        if (col) O3DMI_LAUNCH_STEP(float, float, true);
        else O3DMI_LAUNCH_STEP(float, float, false);
    }
#undef O3DMI_LAUNCH_STEP
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace o3dmi
