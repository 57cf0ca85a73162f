// Block "touch" kernels for VoxelBlockGrid on MI355X.
//
//   o3dmi_vbg_depth_touch       <- DepthTouchCUDA  (t/geometry/kernel/VoxelBlockGridCUDA.cu:106-227)
//                                  arithmetic as DepthTouchCPU (VoxelBlockGridCPU.cpp:117-201)
//   o3dmi_vbg_pointcloud_touch  <- PointCloudTouchCUDA (VoxelBlockGridCUDA.cu:42-104)
//   o3dmi_vbg_touch_activate    fused front end for the frame-stream fast path
//   o3dmi_unproject             <- UnprojectCUDA (t/geometry/kernel/PointCloudImpl.h:42-143)
//
// Design: the reference materialises 4 candidates per ray, activates them in a
// scratch hash map and compacts by mask (3 kernels + 2 host syncs). Here each
// ray inserts its candidates directly into the hash with a single CAS per
// *distinct* key per wave (neighbouring rays mostly hit the same block: a
// wave-level match-any dedup removes ~95 % of the atomics), and the winner of
// each key appends it to the output list. Counts stay on the device.

#include "common.h"
#include <map>
#include <utility>

#include "touch_device.h"

namespace o3dmi {
namespace {

template <typename depth_t>
__global__ void DepthTouchKernel(HashView hv, TouchParams p,
                                 const depth_t* __restrict__ depth,
                                 int* __restrict__ out_coords,
                                 int64_t out_capacity,
                                 int* __restrict__ out_count) {
    int n = p.rows_strided * p.cols_strided;
    int n_padded = ((n + 63) / 64) * 64;  // keep whole waves in the loop
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n_padded;
         w += gridDim.x * blockDim.x) {
        int xb[4], yb[4], zb[4];
        bool valid = (w < n) && RayCandidates(p, depth, w, xb, yb, zb);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bool ok = valid;
            // Within a ray consecutive samples often repeat the block.
            if (ok && s > 0 && xb[s] == xb[s - 1] && yb[s] == yb[s - 1] &&
                zb[s] == zb[s - 1])
                ok = false;
            if (ok && !KeyInRange(xb[s], yb[s], zb[s])) {
                atomicOr(&hv.counters[1], kErrKeyRange);
                ok = false;
            }
            unsigned long long k = ok ? PackKey(xb[s], yb[s], zb[s]) : 0ull;
            if (ok && !hv.Owns(k)) ok = false;  // another rank's block
            if (WaveLeaderForKey(k, ok)) {
                unsigned slot;
                if (InsertKey<false>(hv, xb[s], yb[s], zb[s], slot)) {
                    int o = atomicAdd(out_count, 1);
                    if (o < out_capacity) {
                        out_coords[3 * o + 0] = xb[s];
                        out_coords[3 * o + 1] = yb[s];
                        out_coords[3 * o + 2] = zb[s];
                    } else {
                        atomicOr(&hv.counters[1], kErrCapacity);
                    }
                }
            }
        }
    }
}

// Fused: candidates -> find-or-create in the MAIN block hash -> first toucher
// of a slot in this frame (TouchSlot) appends the slot to the list.
template <typename depth_t>
__global__ void TouchActivateKernel(HashView hv, TouchParams p,
                                    const depth_t* __restrict__ depth,
                                    int* __restrict__ out_slots,
                                    int64_t out_capacity,
                                    int* __restrict__ out_count,
                                    int frame_stamp) {
    int n = p.rows_strided * p.cols_strided;
    int n_padded = ((n + 63) / 64) * 64;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n_padded;
         w += gridDim.x * blockDim.x) {
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
            if (ok && !hv.Owns(k)) ok = false;  // another rank's block
            if (WaveLeaderForKey(k, ok)) {
                unsigned slot;
                InsertKey<true>(hv, xb[s], yb[s], zb[s], slot);
                if (TouchSlot(hv, slot, (unsigned long long)frame_stamp, 0)) {
                    int o = atomicAdd(out_count, 1);
                    if (o < out_capacity) out_slots[o] = (int)slot;
                    else atomicOr(&hv.counters[1], kErrCapacity);
                }
            }
        }
    }
}

// slot list -> buffer indices (slot_vals are published by the previous kernel).
__global__ void SlotsToIndicesKernel(HashView hv, int* __restrict__ io,
                                     const int* __restrict__ count,
                                     int64_t capacity) {
    int64_t n = *count;
    if (n > capacity) n = capacity;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        io[i] = hv.slot_vals[io[i]];
}

__global__ void PointCloudTouchKernel(HashView hv,
                                      const float* __restrict__ pcd, int64_t n,
                                      float block_size, float sdf_trunc,
                                      int* __restrict__ out_coords,
                                      int64_t out_capacity,
                                      int* __restrict__ out_count) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float x = pcd[3 * i + 0], y = pcd[3 * i + 1], z = pcd[3 * i + 2];
        int xb_lo = (int)floorf((x - sdf_trunc) / block_size);
        int xb_hi = (int)floorf((x + sdf_trunc) / block_size);
        int yb_lo = (int)floorf((y - sdf_trunc) / block_size);
        int yb_hi = (int)floorf((y + sdf_trunc) / block_size);
        int zb_lo = (int)floorf((z - sdf_trunc) / block_size);
        int zb_hi = (int)floorf((z + sdf_trunc) / block_size);
        for (int xb = xb_lo; xb <= xb_hi; ++xb)
            for (int yb = yb_lo; yb <= yb_hi; ++yb)
                for (int zb = zb_lo; zb <= zb_hi; ++zb) {
                    if (!KeyInRange(xb, yb, zb)) {
                        atomicOr(&hv.counters[1], kErrKeyRange);
                        continue;
                    }
                    unsigned slot;
                    if (InsertKey<false>(hv, xb, yb, zb, slot)) {
                        int o = atomicAdd(out_count, 1);
                        if (o < out_capacity) {
                            out_coords[3 * o + 0] = xb;
                            out_coords[3 * o + 1] = yb;
                            out_coords[3 * o + 2] = zb;
                        } else {
                            atomicOr(&hv.counters[1], kErrCapacity);
                        }
                    }
                }
    }
}

// UnprojectCPU (t/geometry/kernel/PointCloudImpl.h:42-143): valid pixels ->
// points, compacted IN PIXEL ORDER by one launch. A workgroup owns a chunk of
// kBlock * ROUNDS consecutive strided pixels: validity ballots per wave and
// round, one prefix over the chunk in LDS; the chunk's total goes out as ONE
// write-through 8-byte word {launch sequence number, total}, and the
// workgroup reads the words of every chunk before its own (<= 4 per lane,
// polled until they carry this launch's number: chunks are dispatched in
// index order and publish before they wait, so a chunk only ever waits for
// workgroups that are already running) -- their sum is where its points go.
// The last chunk writes the cloud's size.
// (Rounds 1-5 took the chunk's range off one atomic counter: arrival order,
// not reproducible run to run -- the one such output of the tracking loop --
// 300 serialised returning atomics per VGA cloud, and since round 5 a fenced
// ticket per workgroup to hand the total over: 11.5 us per launch. A
// pixel-order mode existed as three launches, count -> scan -> write.)
// ROUNDS trades words to poll against workgroups to spread the image over:
// the host picks the largest ROUNDS that still gives every CU a chunk.
constexpr int kUnprojMaxRounds = 8;
constexpr int kUnprojSpinLimit = 1 << 22;  // never hang the device

template <typename depth_t, int ROUNDS>
__global__ void __launch_bounds__(kBlock)
UnprojectKernel(TouchParams p, const depth_t* __restrict__ depth,
                const float* __restrict__ image_colors,
                float* __restrict__ points, float* __restrict__ colors,
                int* __restrict__ count, unsigned long long* chunk_words,
                unsigned seq, int n_chunks) {
    __shared__ int offs[ROUNDS][kBlock / 64];
    __shared__ int before[kBlock / 64];
    __shared__ int chunk_total;
    const int64_t n = (int64_t)p.rows_strided * p.cols_strided;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    {
        const int chunk = blockIdx.x;
        const int64_t c0 = (int64_t)chunk * (kBlock * ROUNDS);
        float d[ROUNDS];
        unsigned long long ballot[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int64_t w = c0 + k * kBlock + threadIdx.x;
            bool valid = false;
            d[k] = 0;
            if (w < n) {
                const int64_t y = (w / p.cols_strided) * p.stride;
                const int64_t x = (w % p.cols_strided) * p.stride;
                d[k] = (float)depth[y * p.cols + x] / p.depth_scale;
                valid = d[k] > 0 && d[k] < p.depth_max;
            }
            ballot[k] = __ballot(valid);
            if (lane == 0) offs[k][wave] = __popcll(ballot[k]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k)
#pragma unroll
                for (int wv = 0; wv < kBlock / 64; ++wv) {
                    const int c = offs[k][wv];
                    offs[k][wv] = run;
                    run += c;
                }
            __hip_atomic_store(&chunk_words[chunk],
                               ((unsigned long long)seq << 32) | (unsigned)run,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chunk_total = run;
        }
        // the chunks before this one
        int sum = 0;
        for (int q = threadIdx.x; q < chunk; q += kBlock) {
            unsigned long long w;
            int spins = 0;
            do {
                w = __hip_atomic_load(&chunk_words[q], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)(w >> 32) != seq && ++spins < kUnprojSpinLimit);
            sum += (int)(unsigned)w;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
        if (lane == 0) before[wave] = sum;
        __syncthreads();
        int64_t base = 0;
#pragma unroll
        for (int wv = 0; wv < kBlock / 64; ++wv) base += before[wv];
        if (chunk == n_chunks - 1 && threadIdx.x == 0)
            *count = (int)base + chunk_total;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            if (!((ballot[k] >> lane) & 1ull)) continue;
            const int64_t w = c0 + k * kBlock + threadIdx.x;
            const int64_t y = (w / p.cols_strided) * p.stride;
            const int64_t x = (w % p.cols_strided) * p.stride;
            const int64_t idx = base + offs[k][wave] + __popcll(ballot[k] & lt);
            float x_c, y_c, z_c, xo, yo, zo;
            p.cam.Unproject((float)x, (float)y, d[k], x_c, y_c, z_c);
            p.cam.RigidTransform(x_c, y_c, z_c, xo, yo, zo);
            points[3 * idx + 0] = xo;
            points[3 * idx + 1] = yo;
            points[3 * idx + 2] = zo;
            if (colors && image_colors) {
                const float* ip = image_colors + 3 * (y * p.cols + x);
                colors[3 * idx + 0] = ip[0];
                colors[3 * idx + 1] = ip[1];
                colors[3 * idx + 2] = ip[2];
            }
        }
    }
}

// Two images of one size in ONE launch (o3dmi_unproject_pair: the model frame
// a ray cast rendered and the camera's new frame, the two clouds a tracking
// step starts from): workgroups [0, a.n_chunks) are cloud a's chunks, the rest
// cloud b's. A chunk still waits only for chunks of its own cloud before it,
// all of them dispatched earlier. The depth type is a per-cloud run-time flag
// (uniform over the workgroup); everything else is UnprojectKernel's body, to
// the bit.
struct UnprojectJob {
    TouchParams p;
    const void* depth;
    int depth_is_u16;
    const float* image_colors;
    float* points;
    float* colors;
    int* count;
    unsigned long long* chunk_words;
    int n_chunks;
};

template <int ROUNDS>
__global__ void __launch_bounds__(kBlock)
UnprojectPairKernel(UnprojectJob job_a, UnprojectJob job_b, unsigned seq) {
    __shared__ int offs[ROUNDS][kBlock / 64];
    __shared__ int before[kBlock / 64];
    __shared__ int chunk_total;
    const bool second = (int)blockIdx.x >= job_a.n_chunks;
    const UnprojectJob& J = second ? job_b : job_a;
    const TouchParams& p = J.p;
    const int chunk = (int)blockIdx.x - (second ? job_a.n_chunks : 0);
    const int64_t n = (int64_t)p.rows_strided * p.cols_strided;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t c0 = (int64_t)chunk * (kBlock * ROUNDS);
    float d[ROUNDS];
    unsigned long long ballot[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int64_t w = c0 + k * kBlock + threadIdx.x;
        bool valid = false;
        d[k] = 0;
        if (w < n) {
            const int64_t y = (w / p.cols_strided) * p.stride;
            const int64_t x = (w % p.cols_strided) * p.stride;
            const int64_t at = y * p.cols + x;
            const float raw = J.depth_is_u16
                                      ? (float)((const uint16_t*)J.depth)[at]
                                      : ((const float*)J.depth)[at];
            d[k] = raw / p.depth_scale;
            valid = d[k] > 0 && d[k] < p.depth_max;
        }
        ballot[k] = __ballot(valid);
        if (lane == 0) offs[k][wave] = __popcll(ballot[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k)
#pragma unroll
            for (int wv = 0; wv < kBlock / 64; ++wv) {
                const int c = offs[k][wv];
                offs[k][wv] = run;
                run += c;
            }
        __hip_atomic_store(&J.chunk_words[chunk],
                           ((unsigned long long)seq << 32) | (unsigned)run,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chunk_total = run;
    }
    int sum = 0;
    for (int q = threadIdx.x; q < chunk; q += kBlock) {
        unsigned long long w;
        int spins = 0;
        do {
            w = __hip_atomic_load(&J.chunk_words[q], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
        } while ((unsigned)(w >> 32) != seq && ++spins < kUnprojSpinLimit);
        sum += (int)(unsigned)w;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
    if (lane == 0) before[wave] = sum;
    __syncthreads();
    int64_t base = 0;
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; ++wv) base += before[wv];
    if (chunk == J.n_chunks - 1 && threadIdx.x == 0)
        *J.count = (int)base + chunk_total;
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        if (!((ballot[k] >> lane) & 1ull)) continue;
        const int64_t w = c0 + k * kBlock + threadIdx.x;
        const int64_t y = (w / p.cols_strided) * p.stride;
        const int64_t x = (w % p.cols_strided) * p.stride;
        const int64_t idx = base + offs[k][wave] + __popcll(ballot[k] & lt);
        float x_c, y_c, z_c, xo, yo, zo;
        p.cam.Unproject((float)x, (float)y, d[k], x_c, y_c, z_c);
        p.cam.RigidTransform(x_c, y_c, z_c, xo, yo, zo);
        J.points[3 * idx + 0] = xo;
        J.points[3 * idx + 1] = yo;
        J.points[3 * idx + 2] = zo;
        if (J.colors && J.image_colors) {
            const float* ip = J.image_colors + 3 * (y * p.cols + x);
            J.colors[3 * idx + 0] = ip[0];
            J.colors[3 * idx + 1] = ip[1];
            J.colors[3 * idx + 2] = ip[2];
        }
    }
}

}  // namespace

// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadTouch() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &DepthTouchKernel<uint16_t>)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi

using namespace o3dmi;

extern "C" {

int o3dmi_vbg_depth_touch(o3dmi_hash_t* fh, const void* depth_dev,
                          int depth_dtype, int rows, int cols,
                          const double* intrinsic, const double* extrinsic,
                          int32_t* out_coords_dev, int64_t out_capacity,
                          int32_t* out_count_dev, int resolution,
                          float voxel_size, float sdf_trunc, float depth_scale,
                          float depth_max, int stride, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(fh && depth_dev && intrinsic && extrinsic && out_coords_dev &&
                          out_count_dev,
                  "null argument");
    O3DMI_REQUIRE(depth_dtype == O3DMI_U16 || depth_dtype == O3DMI_F32,
                  "depth dtype must be UInt16 or Float32");
    O3DMI_REQUIRE(stride > 0 && rows >= stride && cols >= stride,
                  "bad image size / stride");
    hipStream_t s = (hipStream_t)stream;
    TouchParams p = MakeTouchParams(intrinsic, extrinsic, rows, cols, stride,
                                    resolution, voxel_size, sdf_trunc,
                                    depth_scale, depth_max);
    // frustum map and output count cleared by one launch
    int st = ClearHashAndCounter(fh, out_count_dev, s);
    if (st != O3DMI_OK) return st;
    int n = p.rows_strided * p.cols_strided;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (depth_dtype == O3DMI_U16)
        hipLaunchKernelGGL(DepthTouchKernel<uint16_t>, grid, block, 0, s,
                           fh->view, p, (const uint16_t*)depth_dev,
                           out_coords_dev, out_capacity, out_count_dev);
    else
        hipLaunchKernelGGL(DepthTouchKernel<float>, grid, block, 0, s, fh->view,
                           p, (const float*)depth_dev, out_coords_dev,
                           out_capacity, out_count_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_touch_activate(o3dmi_hash_t* bh, const void* depth_dev,
                             int depth_dtype, int rows, int cols,
                             const double* intrinsic, const double* extrinsic,
                             int32_t* out_buf_indices_dev,
                             int64_t out_capacity, int32_t* out_count_dev,
                             int resolution, float voxel_size, float sdf_trunc,
                             float depth_scale, float depth_max, int stride,
                             int32_t frame_stamp, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(bh && depth_dev && intrinsic && extrinsic &&
                          out_buf_indices_dev && out_count_dev,
                  "null argument");
    O3DMI_REQUIRE(depth_dtype == O3DMI_U16 || depth_dtype == O3DMI_F32,
                  "depth dtype must be UInt16 or Float32");
    O3DMI_REQUIRE(stride > 0 && rows >= stride && cols >= stride,
                  "bad image size / stride");
    O3DMI_REQUIRE(frame_stamp > 0, "frame_stamp must be positive");
    hipStream_t s = (hipStream_t)stream;
    TouchParams p = MakeTouchParams(intrinsic, extrinsic, rows, cols, stride,
                                    resolution, voxel_size, sdf_trunc,
                                    depth_scale, depth_max);
    O3DMI_HIP_CHECK(hipMemsetAsync(out_count_dev, 0, sizeof(int), s));
    int n = p.rows_strided * p.cols_strided;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (depth_dtype == O3DMI_U16)
        hipLaunchKernelGGL(TouchActivateKernel<uint16_t>, grid, block, 0, s,
                           bh->view, p, (const uint16_t*)depth_dev,
                           out_buf_indices_dev, out_capacity, out_count_dev,
                           frame_stamp);
    else
        hipLaunchKernelGGL(TouchActivateKernel<float>, grid, block, 0, s,
                           bh->view, p, (const float*)depth_dev,
                           out_buf_indices_dev, out_capacity, out_count_dev,
                           frame_stamp);
    O3DMI_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(SlotsToIndicesKernel, dim3(64), dim3(kBlock), 0, s,
                       bh->view, out_buf_indices_dev, out_count_dev,
                       out_capacity);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_pointcloud_touch(o3dmi_hash_t* fh, const float* points_dev,
                               int64_t n, int32_t* out_coords_dev,
                               int64_t out_capacity, int32_t* out_count_dev,
                               int resolution, float voxel_size,
                               float sdf_trunc, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(fh && points_dev && out_coords_dev && out_count_dev,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    int st = ClearHashAndCounter(fh, out_count_dev, s);
    if (st != O3DMI_OK) return st;
    if (n == 0) return O3DMI_OK;
    hipLaunchKernelGGL(PointCloudTouchKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, s, fh->view, points_dev, n,
                       voxel_size * resolution, sdf_trunc, out_coords_dev,
                       out_capacity, out_count_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// The chunk totals' words of a launch, per host thread, device and stream (two
// calls of one thread on two streams may overlap on the device; the launches
// of one stream follow each other), grown on demand. A word is valid for the
// launch whose sequence number it carries: nothing is cleared between
// launches.
static int UnprojectWords(int n_chunks, hipStream_t s,
                          unsigned long long** words, unsigned* seq) {
    struct Words {
        unsigned long long* buf = nullptr;
        int cap = 0;
        unsigned seq = 0;
    };
    static thread_local std::map<std::pair<int, hipStream_t>, Words> bufs;
    int dev = 0;
    O3DMI_HIP_CHECK(hipGetDevice(&dev));
    Words& c = bufs[std::make_pair(dev, s)];
    if (c.cap < n_chunks) {
        if (c.buf) {
            O3DMI_HIP_CHECK(hipStreamSynchronize(s));
            (void)hipFree(c.buf);
            c.buf = nullptr;
            c.cap = 0;
        }
        int cap = 4096;
        while (cap < n_chunks) cap <<= 1;
        O3DMI_HIP_CHECK(hipMalloc((void**)&c.buf,
                                  sizeof(unsigned long long) * cap));
        O3DMI_HIP_CHECK(hipMemsetAsync(c.buf, 0,
                                       sizeof(unsigned long long) * cap, s));
        c.cap = cap;
        c.seq = 0;
    }
    if (++c.seq == 0) ++c.seq;  // 0 = the cleared buffer
    *words = c.buf;
    *seq = c.seq;
    return O3DMI_OK;
}

int o3dmi_unproject(const void* depth_dev, int depth_dtype, int rows, int cols,
                    const float* image_colors_dev, float* points_dev,
                    float* colors_dev, int32_t* out_count_dev,
                    const double* intrinsic, const double* extrinsic,
                    float depth_scale, float depth_max, int64_t stride,
                    o3dmi_stream_t stream) {
    O3DMI_REQUIRE(depth_dev && points_dev && out_count_dev && intrinsic &&
                          extrinsic,
                  "null argument");
    O3DMI_REQUIRE(depth_dtype == O3DMI_U16 || depth_dtype == O3DMI_F32,
                  "depth dtype must be UInt16 or Float32");
    O3DMI_REQUIRE(stride > 0 && rows >= stride && cols >= stride,
                  "bad image size / stride");
    hipStream_t s = (hipStream_t)stream;
    TouchParams p = MakeTouchParams(intrinsic, extrinsic, rows, cols,
                                    (int)stride, 1, 1.0f, 0.0f, depth_scale,
                                    depth_max);
    int64_t n = (int64_t)p.rows_strided * p.cols_strided;
    // the largest chunk that still gives every CU one
    int rounds = kUnprojMaxRounds;
    while (rounds > 1 && n < (int64_t)kCUs * kBlock * rounds) rounds >>= 1;
    const int n_chunks = (int)((n + (int64_t)kBlock * rounds - 1) /
                               ((int64_t)kBlock * rounds));
    // one workgroup per chunk, all of them: a chunk waits for the chunks
    // before it only, and workgroups are dispatched in index order
    dim3 grid((unsigned)n_chunks), block(kBlock);
    unsigned long long* chunk_words = nullptr;
    unsigned seq = 0;
    {
        int st = UnprojectWords(n_chunks, s, &chunk_words, &seq);
        if (st != O3DMI_OK) return st;
    }
#define O3DMI_UNPROJECT(T, R)                                                  \
    hipLaunchKernelGGL((UnprojectKernel<T, R>), grid, block, 0, s, p,         \
                       (const T*)depth_dev, image_colors_dev, points_dev,     \
                       colors_dev, out_count_dev, chunk_words, seq, n_chunks)
#define O3DMI_UNPROJECT_R(T)                                                   \
    switch (rounds) {                                                         \
        case 8: O3DMI_UNPROJECT(T, 8); break;                                 \
        case 4: O3DMI_UNPROJECT(T, 4); break;                                 \
        case 2: O3DMI_UNPROJECT(T, 2); break;                                 \
        default: O3DMI_UNPROJECT(T, 1); break;                                \
    }
    if (depth_dtype == O3DMI_U16) { O3DMI_UNPROJECT_R(uint16_t) }
    else { O3DMI_UNPROJECT_R(float) }
#undef O3DMI_UNPROJECT_R
#undef O3DMI_UNPROJECT
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_unproject_pair(
        const void* depth_a_dev, int depth_a_dtype,
        const float* image_colors_a_dev, float* points_a_dev,
        float* colors_a_dev, int32_t* out_count_a_dev,
        const double* extrinsic_a, const void* depth_b_dev, int depth_b_dtype,
        const float* image_colors_b_dev, float* points_b_dev,
        float* colors_b_dev, int32_t* out_count_b_dev,
        const double* extrinsic_b, int rows, int cols, const double* intrinsic,
        float depth_scale, float depth_max, int64_t stride,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(depth_a_dev && points_a_dev && out_count_a_dev &&
                          extrinsic_a && depth_b_dev && points_b_dev &&
                          out_count_b_dev && extrinsic_b && intrinsic,
                  "null argument");
    O3DMI_REQUIRE((depth_a_dtype == O3DMI_U16 || depth_a_dtype == O3DMI_F32) &&
                          (depth_b_dtype == O3DMI_U16 ||
                           depth_b_dtype == O3DMI_F32),
                  "depth dtype must be UInt16 or Float32");
    O3DMI_REQUIRE(stride > 0 && rows >= stride && cols >= stride,
                  "bad image size / stride");
    O3DMI_REQUIRE(points_a_dev != points_b_dev &&
                          out_count_a_dev != out_count_b_dev,
                  "the two clouds share an output");
    hipStream_t s = (hipStream_t)stream;
    UnprojectJob ja = {}, jb = {};
    ja.p = MakeTouchParams(intrinsic, extrinsic_a, rows, cols, (int)stride, 1,
                           1.0f, 0.0f, depth_scale, depth_max);
    jb.p = MakeTouchParams(intrinsic, extrinsic_b, rows, cols, (int)stride, 1,
                           1.0f, 0.0f, depth_scale, depth_max);
    const int64_t n = (int64_t)ja.p.rows_strided * ja.p.cols_strided;
    // the largest chunk that still gives every CU one (two clouds' chunks)
    int rounds = kUnprojMaxRounds;
    while (rounds > 1 && 2 * n < (int64_t)kCUs * kBlock * rounds) rounds >>= 1;
    const int n_chunks = (int)((n + (int64_t)kBlock * rounds - 1) /
                               ((int64_t)kBlock * rounds));
    unsigned long long* words = nullptr;
    unsigned seq = 0;
    int st = UnprojectWords(2 * n_chunks, s, &words, &seq);
    if (st != O3DMI_OK) return st;
    ja.depth = depth_a_dev;
    ja.depth_is_u16 = depth_a_dtype == O3DMI_U16;
    ja.image_colors = image_colors_a_dev;
    ja.points = points_a_dev;
    ja.colors = colors_a_dev;
    ja.count = out_count_a_dev;
    ja.chunk_words = words;
    ja.n_chunks = n_chunks;
    jb.depth = depth_b_dev;
    jb.depth_is_u16 = depth_b_dtype == O3DMI_U16;
    jb.image_colors = image_colors_b_dev;
    jb.points = points_b_dev;
    jb.colors = colors_b_dev;
    jb.count = out_count_b_dev;
    jb.chunk_words = words + n_chunks;
    jb.n_chunks = n_chunks;
    dim3 grid((unsigned)(2 * n_chunks)), block(kBlock);
    switch (rounds) {
        case 8:
            hipLaunchKernelGGL(UnprojectPairKernel<8>, grid, block, 0, s, ja,
                               jb, seq);
            break;
        case 4:
            hipLaunchKernelGGL(UnprojectPairKernel<4>, grid, block, 0, s, ja,
                               jb, seq);
            break;
        case 2:
            hipLaunchKernelGGL(UnprojectPairKernel<2>, grid, block, 0, s, ja,
                               jb, seq);
            break;
        default:
            hipLaunchKernelGGL(UnprojectPairKernel<1>, grid, block, 0, s, ja,
                               jb, seq);
            break;
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // extern "C"
