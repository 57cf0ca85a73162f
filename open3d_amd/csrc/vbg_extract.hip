// Surface point extraction from the TSDF voxel blocks on MI355X (SURVEY.md
// section 8 row f2). Replaces ExtractPointCloudCUDA<tsdf_t, weight_t, color_t>
// (cpp/open3d/t/geometry/kernel/VoxelBlockGridImpl.h:1122-1365, helpers
// DeviceGetLinearIdx / DeviceGetNormal :94-149) together with the host-side
// BufferRadiusNeighbors table (t/geometry/VoxelBlockGrid.cpp:22-51).
//
// Differences in structure (same per-point arithmetic):
//  * no {27, n} neighbour tables in HBM: one workgroup owns one active block
//    and its first 27 lanes look the neighbours up in the spatial hash into
//    LDS;
//  * no global atomic counter: pass 1 counts the zero crossings per block,
//    a scan turns the counts into offsets, pass 2 writes each point at
//    offset(block) + rank(voxel, axis) -- the output order is (active block,
//    voxel, axis), identical on every run (the reference's order is whatever
//    its atomic counter hands out);
//  * 64-bit linear indices (the reference's `int` overflows past 524 287
//    blocks of 16^3).
// HBM bound in principle (one pass over tsdf + weight of the active blocks,
// 6 B / voxel, twice) with scattered neighbour reads at block faces.


#include "common.h"

namespace o3dmi {
namespace {

constexpr int kExtractBlock = 256;

__device__ __forceinline__ int Sgn(int x) { return (x > 0) - (x < 0); }

// Voxel <-> block arithmetic. The block resolution is a power of two in every
// configuration of the path (8, 16), and a division by a run-time value costs
// tens of instructions -- ~12 of them per voxel here; POW2 turns them into
// masks and shifts.
template <bool POW2>
struct ResMath {
    int res, shift;
    __device__ __forceinline__ int Mod(int x) const {
        return POW2 ? (x & (res - 1)) : (x % res);
    }
    __device__ __forceinline__ int Div(int x) const {
        return POW2 ? (x >> shift) : (x / res);
    }
};

struct ExtractArgs {
    const int32_t* indices;  // [n_blocks] active buffer indices
    const float* tsdf;
    const void* weight;
    const void* color;
    int resolution;
    float voxel_size;
    float weight_threshold;
};

// DeviceGetLinearIdx, VoxelBlockGridImpl.h:94-121; nb = LDS table of the 27
// neighbour buffer indices (-1 = absent).
template <bool POW2>
__device__ __forceinline__ long long LinearIdx(int xo, int yo, int zo,
                                               const ResMath<POW2>& rm,
                                               const int* nb) {
    const int res = rm.res;
    const int xn = rm.Mod(xo + res);
    const int yn = rm.Mod(yo + res);
    const int zn = rm.Mod(zo + res);
    const int nb_idx = (Sgn(xo - xn) + 1) + (Sgn(yo - yn) + 1) * 3 +
                       (Sgn(zo - zn) + 1) * 9;
    const int b = nb[nb_idx];
    if (b < 0) return -1;
    return ((((long long)b * res) + zn) * res + yn) * res + xn;
}

// DeviceGetNormal, :123-149: components are only overwritten when both
// neighbours exist.
template <bool POW2>
__device__ __forceinline__ void GetNormal(const float* __restrict__ tsdf,
                                          int xo, int yo, int zo,
                                          const ResMath<POW2>& res,
                                          const int* nb, float* n) {
    const long long vxp = LinearIdx(xo + 1, yo, zo, res, nb);
    const long long vxn = LinearIdx(xo - 1, yo, zo, res, nb);
    const long long vyp = LinearIdx(xo, yo + 1, zo, res, nb);
    const long long vyn = LinearIdx(xo, yo - 1, zo, res, nb);
    const long long vzp = LinearIdx(xo, yo, zo + 1, res, nb);
    const long long vzn = LinearIdx(xo, yo, zo - 1, res, nb);
    if (vxp >= 0 && vxn >= 0) n[0] = tsdf[vxp] - tsdf[vxn];
    if (vyp >= 0 && vyn >= 0) n[1] = tsdf[vyp] - tsdf[vyn];
    if (vzp >= 0 && vzn >= 0) n[2] = tsdf[vzp] - tsdf[vzn];
}

// Exclusive prefix of v over the workgroup; total = sum over the workgroup.
__device__ __forceinline__ int BlockExclusiveScan(int v, int* wave_sums,
                                                  int& total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    int wave_off = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < kExtractBlock / 64; ++k) {
        const int s = wave_sums[k];
        if (k < wave) wave_off += s;
        total += s;
    }
    __syncthreads();
    return wave_off + x - v;
}

template <typename weight_t, typename color_t, bool WRITE, bool POW2>
__global__ void __launch_bounds__(kExtractBlock)
ExtractKernel(HashView hv, ExtractArgs a, int* __restrict__ block_counts,
              const long long* __restrict__ block_offsets,
              float* __restrict__ points, float* __restrict__ normals,
              float* __restrict__ colors, long long capacity) {
    __shared__ int nb[27];
    __shared__ int wave_sums[kExtractBlock / 64];
    const int res = a.resolution;
    const int res3 = res * res * res;
    ResMath<POW2> rm;
    rm.res = res;
    rm.shift = 31 - __clz(res);
    const int block_idx = a.indices[blockIdx.x];
    const int* key = hv.key_buffer + 3 * (long long)block_idx;
    const int xb = key[0], yb = key[1], zb = key[2];
    if (threadIdx.x < 27) {
        const int t = threadIdx.x;
        const int dz = t / 9, dy = (t % 9) / 3, dx = t % 3;
        nb[t] = (t == 13) ? block_idx
                          : hv.Find(xb + dx - 1, yb + dy - 1, zb + dz - 1);
    }
    __syncthreads();
    const float* __restrict__ tsdf = a.tsdf;
    const weight_t* __restrict__ weight = (const weight_t*)a.weight;
    const color_t* __restrict__ color = (const color_t*)a.color;
    const float thr = a.weight_threshold;
    long long base = WRITE ? block_offsets[blockIdx.x] : 0;
    int block_total = 0;
    for (int v0 = 0; v0 < res3; v0 += kExtractBlock) {
        const int voxel_idx = v0 + threadIdx.x;
        int flags = 0;
        int xv = 0, yv = 0, zv = 0;
        long long linear_idx = 0;
        long long lin_i[3] = {-1, -1, -1};
        float tsdf_o = 0;
        if (voxel_idx < res3) {
            xv = rm.Mod(voxel_idx);
            yv = rm.Mod(rm.Div(voxel_idx));
            zv = rm.Div(rm.Div(voxel_idx));
            linear_idx = (long long)block_idx * res3 + voxel_idx;
            tsdf_o = tsdf[linear_idx];
            const float weight_o = (float)weight[linear_idx];
            if (!(weight_o <= thr)) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const long long li = LinearIdx(xv + (i == 0), yv + (i == 1),
                                                   zv + (i == 2), rm, nb);
                    if (li < 0) continue;
                    const float tsdf_i = tsdf[li];
                    const float weight_i = (float)weight[li];
                    if (weight_i > thr && tsdf_i * tsdf_o < 0) {
                        flags |= 1 << i;
                        lin_i[i] = li;
                    }
                }
            }
        }
        const int cnt = __popc(flags);
        int chunk_total;
        const int rank = BlockExclusiveScan(cnt, wave_sums, chunk_total);
        if (WRITE && flags) {
            float no[3] = {0, 0, 0}, ne[3] = {0, 0, 0};
            GetNormal(tsdf, xv, yv, zv, rm, nb, no);
            const int x = xb * res + xv;
            const int y = yb * res + yv;
            const int z = zb * res + zv;
            long long idx = base + rank;
            // Axes are visited in order; `ne` carries over between the axes
            // of a voxel exactly as in the reference (it is never reset).
            for (int i = 0; i < 3; ++i) {
                if (!(flags & (1 << i))) {
                    continue;
                }
                const long long li = lin_i[i];
                const float tsdf_i = tsdf[li];
                const float ratio = (0 - tsdf_o) / (tsdf_i - tsdf_o);
                GetNormal(tsdf, xv + (i == 0), yv + (i == 1), zv + (i == 2),
                          rm, nb, ne);
                if (idx < capacity) {
                    float* p = points + 3 * idx;
                    p[0] = a.voxel_size * ((float)x + ratio * (float)(int)(i == 0));
                    p[1] = a.voxel_size * ((float)y + ratio * (float)(int)(i == 1));
                    p[2] = a.voxel_size * ((float)z + ratio * (float)(int)(i == 2));
                    const float nx = (1 - ratio) * no[0] + ratio * ne[0];
                    const float ny = (1 - ratio) * no[1] + ratio * ne[1];
                    const float nz = (1 - ratio) * no[2] + ratio * ne[2];
                    const float norm = (float)((double)sqrtf(nx * nx + ny * ny +
                                                             nz * nz) +
                                               1e-5);
                    float* nn = normals + 3 * idx;
                    nn[0] = nx / norm;
                    nn[1] = ny / norm;
                    nn[2] = nz / norm;
                    if (color != nullptr && colors != nullptr) {
                        const color_t* co = color + 3 * linear_idx;
                        const color_t* ci = color + 3 * li;
                        const float r_o = (float)co[0], g_o = (float)co[1],
                                    b_o = (float)co[2];
                        const float r_i = (float)ci[0], g_i = (float)ci[1],
                                    b_i = (float)ci[2];
                        float* c = colors + 3 * idx;
                        c[0] = ((1 - ratio) * r_o + ratio * r_i) / 255.0f;
                        c[1] = ((1 - ratio) * g_o + ratio * g_i) / 255.0f;
                        c[2] = ((1 - ratio) * b_o + ratio * b_i) / 255.0f;
                    }
                }
                ++idx;
            }
        }
        base += chunk_total;
        block_total += chunk_total;
    }
    if (!WRITE && threadIdx.x == 0) block_counts[blockIdx.x] = block_total;
}

// offsets[i] = sum counts[0..i), offsets[n] = total. One workgroup.
constexpr int kScanThreads = 1024;
__global__ void __launch_bounds__(kScanThreads)
ScanCountsKernel(const int* __restrict__ counts, long long* __restrict__ offsets,
                 long long n) {
    __shared__ long long wave_sums[kScanThreads / 64];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long long i0 = 0; i0 < n; i0 += kScanThreads) {
        const long long i = i0 + threadIdx.x;
        const long long v = i < n ? (long long)counts[i] : 0;
        long long x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wave_sums[wave] = x;
        __syncthreads();
        long long wave_off = 0, total = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) {
            const long long s = wave_sums[k];
            if (k < wave) wave_off += s;
            total += s;
        }
        const long long carry = carry_s;
        if (i < n) offsets[i] = carry + wave_off + x - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry_s;
}

}  // namespace
}  // namespace o3dmi

using namespace o3dmi;

extern "C" int o3dmi_vbg_extract_points(
        o3dmi_hash_t* block_hash, const int32_t* indices_dev, int64_t n_blocks,
        const float* tsdf_dev, const void* weight_dev, const void* color_dev,
        int grid_dtype, int resolution, float voxel_size,
        float weight_threshold, float* points_dev, float* normals_dev,
        float* colors_dev, int64_t capacity, int64_t* total_out,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(block_hash && total_out, "null argument");
    O3DMI_REQUIRE(n_blocks >= 0 && n_blocks < (1ll << 31), "n_blocks out of range");
    O3DMI_REQUIRE(resolution > 0 && resolution <= 64, "bad block resolution");
    O3DMI_REQUIRE(grid_dtype == O3DMI_F32 || grid_dtype == O3DMI_U16,
                  "Unsupported value data type combination. Expected (float, "
                  "float) or (uint16, uint16)");
    *total_out = 0;
    if (n_blocks == 0) return O3DMI_OK;
    O3DMI_REQUIRE(indices_dev && tsdf_dev && weight_dev,
                  "TSDF and/or weight not allocated in blocks, please implement "
                  "customized integration.");
    const bool write = capacity >= 0;
    if (write && capacity > 0)
        O3DMI_REQUIRE(points_dev && normals_dev, "null output");
    hipStream_t s = (hipStream_t)stream;
    char* scratch = nullptr;
    const size_t off_bytes = sizeof(long long) * ((size_t)n_blocks + 1);
    const size_t cnt_bytes = sizeof(int) * (size_t)n_blocks;
    int st = PoolAlloc((void**)&scratch, off_bytes + cnt_bytes);
    if (st) return st;
    long long* offsets = (long long*)scratch;
    int* counts = (int*)(scratch + off_bytes);
    ExtractArgs a;
    a.indices = indices_dev;
    a.tsdf = tsdf_dev;
    a.weight = weight_dev;
    a.color = color_dev;
    a.resolution = resolution;
    a.voxel_size = voxel_size;
    a.weight_threshold = weight_threshold;
    const dim3 grid((unsigned)n_blocks), block(kExtractBlock);
    const HashView hv = block_hash->view;
    const bool pow2 = (resolution & (resolution - 1)) == 0;
#define O3DMI_EXTRACT_P(WT, CT, WR, P2)                                        \
    hipLaunchKernelGGL((ExtractKernel<WT, CT, WR, P2>), grid, block, 0, s, hv, \
                       a, counts, offsets, points_dev, normals_dev,            \
                       colors_dev, (long long)capacity)
#define O3DMI_EXTRACT(WT, CT, WR)                                              \
    do {                                                                       \
        if (pow2) O3DMI_EXTRACT_P(WT, CT, WR, true);                           \
        else O3DMI_EXTRACT_P(WT, CT, WR, false);                               \
    } while (0)
    if (grid_dtype == O3DMI_F32) O3DMI_EXTRACT(float, float, false);
    else O3DMI_EXTRACT(uint16_t, uint16_t, false);
    hipLaunchKernelGGL(ScanCountsKernel, dim3(1), dim3(kScanThreads), 0, s,
                       counts, offsets, (long long)n_blocks);
    if (write && capacity > 0) {
        if (grid_dtype == O3DMI_F32) O3DMI_EXTRACT(float, float, true);
        else O3DMI_EXTRACT(uint16_t, uint16_t, true);
    }
#undef O3DMI_EXTRACT
#undef O3DMI_EXTRACT_P
    long long total = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(&total, offsets + n_blocks, sizeof(long long),
                           hipMemcpyDeviceToHost, s);
    hipError_t e2 = hipStreamSynchronize(s);
    PoolFree(scratch);
    O3DMI_HIP_CHECK(e);
    O3DMI_HIP_CHECK(e2);
    *total_out = (int64_t)total;
    return O3DMI_OK;
}
