// Point-cloud pyramid support for MultiScaleICP on MI355X.
//
//   o3dmi_voxel_down_sample <- t::geometry::PointCloud::VoxelDownSample
//                              (t/geometry/PointCloud.cpp:496-567)
//
// Reference semantics reproduced exactly (CPU tensor path):
//   voxel = floor(p / T(voxel_size)) computed in the point dtype T;
//   every attribute is summed in Float32 in POINT ORDER (IndexAdd_ on the CPU
//   is a sequential loop, core/kernel/IndexReductionCPU.cpp:52-56), divided by
//   the Float32 point count and cast back to T; normals are averaged, not
//   re-normalised. Output order = order of each voxel's first point (the
//   reference leaves it unspecified).
//
// The float sums are order dependent, so a scatter-add with atomics would not
// reproduce them. Instead:
//   1. hash the voxel keys (packed 64-bit, open addressing) and record each
//      slot's smallest point index (atomicMin);
//   2. first points -> dense voxel ids in first-occurrence order (block counts,
//      then a block-local scan on top of the preceding blocks' counts);
//   3. STABLE counting sort of the points by voxel id, least-significant
//      11-bit digit first (per pass: block histograms -> scatter that sums
//      the (digit, block) table for its own offsets and ranks equal digits by
//      wave, round and lane); keys are the dense ids, far below 2^22, so this
//      is two passes -- every voxel's points end up contiguous and still in
//      point order;
//   4. one lane per voxel walks its segment and adds sequentially in float32.
//
// Nothing in the chain needs a host decision: the point count may live on the
// device (the output count of the previous, finer level), every kernel bounds
// itself by it, and the voxel count is left on the device as well. A pyramid
// of several levels is therefore ONE string of launches with a single read-
// back at its end (VdsAsync, used by the ICP driver); the public entry point
// runs one level and reads the count back. The earlier form used a generic
// radix sort (16 launches) and two host waits per level; at 77 k-point VGA
// clouds the pyramid took 1.3 ms of a 1.7 ms tracking frame.

#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "o3d_mi355x_host.h"
#include "vds.h"

namespace o3dmi {
namespace {

constexpr int kSortBits = 11;
constexpr int kSortBins = 1 << kSortBits;
constexpr int kSortBlock = 256;             // 4 waves
constexpr int kSortItems = 8;               // elements per thread
constexpr int kSortTile = kSortBlock * kSortItems;  // 2048 elements per block

struct VdsTable {
    unsigned long long* keys;  // [n_slots], kEmptyKey when free
    int* first;                // [n_slots] smallest point index
    int* voxel;                // [n_slots] dense voxel id
    unsigned mask;
};

__device__ __forceinline__ int LiveCount(const int* n_dev, int n_host) {
    if (!n_dev) return n_host;
    const int n = *n_dev;
    return n < n_host ? n : n_host;  // never beyond what the buffers hold
}

template <typename T>
__global__ void VdsInsertKernel(const T* __restrict__ pos, const int* n_dev,
                                int n_host, T vs, VdsTable tb,
                                int* __restrict__ slot_of_point,
                                int* __restrict__ err) {
    const int n = LiveCount(n_dev, n_host);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += gridDim.x * blockDim.x) {
        // (p / vs).Floor().To(Int64)
        const long long cx = (long long)floor(pos[3 * (int64_t)i + 0] / vs);
        const long long cy = (long long)floor(pos[3 * (int64_t)i + 1] / vs);
        const long long cz = (long long)floor(pos[3 * (int64_t)i + 2] / vs);
        if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias ||
            cy >= kKeyBias || cz < -kKeyBias || cz >= kKeyBias) {
            atomicOr(err, kErrKeyRange);
            slot_of_point[i] = 0;
            continue;
        }
        const unsigned long long k = PackKey((int)cx, (int)cy, (int)cz);
        unsigned h = HashKey(k) & tb.mask;
        while (true) {
            unsigned long long cur = tb.keys[h];
            if (cur == kEmptyKey)
                cur = atomicCAS(&tb.keys[h], kEmptyKey, k);
            if (cur == kEmptyKey || cur == k) break;
            h = (h + 1) & tb.mask;
        }
        slot_of_point[i] = (int)h;
        atomicMin(&tb.first[h], i);
    }
}

// Block-wide exclusive prefix of one value per thread (256 threads); returns
// the prefix, *total = sum over the block.
__device__ __forceinline__ int BlockExclusive(int v, int* lds4, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSortBlock / 64; ++w) {
        const int c = lds4[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// First points per tile of kSortTile points.
__global__ void VdsCountFirstKernel(const int* __restrict__ slot_of_point,
                                    const int* n_dev, int n_host, VdsTable tb,
                                    int* __restrict__ tile_counts) {
    __shared__ int lds4[4];
    const int n = LiveCount(n_dev, n_host);
    const int base = blockIdx.x * kSortTile;
    if (base >= n) return;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + threadIdx.x * kSortItems + k;
        if (i < n) c += tb.first[slot_of_point[i]] == i;
    }
    int total;
    (void)BlockExclusive(c, lds4, &total);
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = total;
}

// Dense voxel ids in first-occurrence order: preceding tiles' counts + the
// tile-local prefix. Tile 0 also publishes the voxel count.
__global__ void VdsAssignKernel(const int* __restrict__ slot_of_point,
                                const int* n_dev, int n_host, VdsTable tb,
                                const int* __restrict__ tile_counts,
                                int* __restrict__ m_dev) {
    __shared__ int lds4[4];
    const int n = LiveCount(n_dev, n_host);
    const int base = blockIdx.x * kSortTile;
    if (base >= n && blockIdx.x != 0) return;
    const int n_tiles = (n + kSortTile - 1) / kSortTile;
    // counts of the tiles before this one (and of all tiles, for tile 0)
    int before = 0, all = 0;
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
        const int c = tile_counts[t];
        all += c;
        if (t < (int)blockIdx.x) before += c;
    }
    int tot_before, tot_all;
    (void)BlockExclusive(before, lds4, &tot_before);
    (void)BlockExclusive(all, lds4, &tot_all);
    if (blockIdx.x == 0 && threadIdx.x == 0) *m_dev = tot_all;
    if (base >= n) return;
    bool f[kSortItems];
    int c = 0;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + threadIdx.x * kSortItems + k;
        f[k] = i < n && tb.first[slot_of_point[i]] == i;
        c += f[k];
    }
    int total;
    int id = tot_before + BlockExclusive(c, lds4, &total);
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + threadIdx.x * kSortItems + k;
        if (f[k]) tb.voxel[slot_of_point[i]] = id++;
    }
}

// ---- stable counting sort, one 11-bit digit per pass ------------------------
// A block owns kSortTile consecutive elements; wave w of the block owns the
// elements [w * 512, (w + 1) * 512) of the tile, visited in 8 rounds of 64
// (element = tile + w * 512 + round * 64 + lane), so (block, wave, round, lane)
// is index order. hist is digit-major: hist[digit * tile_stride + tile].

// kMakeKeys: pass 0 also creates the keys (voxel id of every point) and the
// values (point indices).
template <bool kMakeKeys>
__global__ void __launch_bounds__(kSortBlock)
SortHistKernel(const int* __restrict__ slot_of_point, VdsTable tb,
               unsigned* __restrict__ keys, unsigned* __restrict__ vals,
               const int* n_dev, int n_host, int shift, int tile_stride,
               int* __restrict__ hist) {
    __shared__ int h[kSortBins];
    const int n = LiveCount(n_dev, n_host);
    const int base = blockIdx.x * kSortTile;
    if (base >= n) return;
    for (int b = threadIdx.x; b < kSortBins; b += kSortBlock) h[b] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + k * kSortBlock + threadIdx.x;
        if (i < n) {
            unsigned key;
            if constexpr (kMakeKeys) {
                key = (unsigned)tb.voxel[slot_of_point[i]];
                keys[i] = key;
                vals[i] = (unsigned)i;
            } else {
                key = keys[i];
            }
            atomicAdd(&h[(key >> shift) & (kSortBins - 1)], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSortBins; b += kSortBlock)
        hist[(int64_t)b * tile_stride + blockIdx.x] = h[b];
}

__global__ void __launch_bounds__(kSortBlock)
SortScatterKernel(const unsigned* __restrict__ keys_in,
                  const unsigned* __restrict__ vals_in,
                  unsigned* __restrict__ keys_out,
                  unsigned* __restrict__ vals_out, const int* n_dev, int n_host,
                  int shift, int tile_stride, const int* __restrict__ hist) {
    __shared__ int wh[kSortBlock / 64][kSortBins];  // 32 KiB
    __shared__ int dbase[kSortBins];                // 8 KiB
    __shared__ int lds4[4];
    const int n = LiveCount(n_dev, n_host);
    const int tile = blockIdx.x * kSortTile;
    if (tile >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = threadIdx.x; b < kSortBins * (kSortBlock / 64);
         b += kSortBlock)
        (&wh[0][0])[b] = 0;
    __syncthreads();
    const int wbase = tile + wave * (kSortItems * 64);
    unsigned key[kSortItems], val[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        key[r] = 0;
        val[r] = 0;
        if (e < n) {
            key[r] = keys_in[e];
            val[r] = vals_in[e];
            atomicAdd(&wh[wave][(key[r] >> shift) & (kSortBins - 1)], 1);
        }
    }
    __syncthreads();
    // Where this tile's run of every digit starts: all elements with a
    // smaller digit (over all tiles) + the same digit in the tiles before
    // this one. The (digit, tile) table is small (2048 x a few dozen tiles
    // for the clouds of a tracking frame), so every block sums what it needs
    // itself -- two launches fewer per pass than a separate table scan, and
    // this chain is launch-latency bound. Thread t owns the 8 consecutive
    // digits 8 t .. 8 t + 7, so that the block prefix is in digit order.
    {
        const int n_tiles = (n + kSortTile - 1) / kSortTile;
        int tot[kSortBins / kSortBlock], before[kSortBins / kSortBlock];
        int mine = 0;
#pragma unroll
        for (int q = 0; q < kSortBins / kSortBlock; ++q) {
            const int dgt = threadIdx.x * (kSortBins / kSortBlock) + q;
            const int* row = hist + (int64_t)dgt * tile_stride;
            int a = 0, bsum = 0;
            for (int t = 0; t < n_tiles; ++t) {
                const int c = row[t];
                a += c;
                if (t < (int)blockIdx.x) bsum += c;
            }
            tot[q] = a;
            before[q] = bsum;
            mine += a;
        }
        int total;
        int run = BlockExclusive(mine, lds4, &total);
#pragma unroll
        for (int q = 0; q < kSortBins / kSortBlock; ++q) {
            dbase[threadIdx.x * (kSortBins / kSortBlock) + q] = run + before[q];
            run += tot[q];
        }
    }
    __syncthreads();
    // per digit: where each wave's run starts in the output
    for (int b = threadIdx.x; b < kSortBins; b += kSortBlock) {
        int off = dbase[b];
#pragma unroll
        for (int w = 0; w < kSortBlock / 64; ++w) {
            const int c = wh[w][b];
            wh[w][b] = off;
            off += c;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        const bool valid = e < n;
        const unsigned d = (key[r] >> shift) & (kSortBins - 1);
        // lanes of this round holding the same digit
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kSortBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        int pos = 0;
        if (valid) pos = wh[wave][d] + __popcll(same & lt);
        // every lane has read its start; the first lane of each digit group
        // moves the start past the group (only this wave touches wh[wave])
        if (valid && (same & lt) == 0ull) wh[wave][d] += __popcll(same);
        if (valid) {
            keys_out[pos] = key[r];
            vals_out[pos] = val[r];
        }
    }
}

__global__ void VdsSegmentsKernel(const unsigned* __restrict__ sorted_voxel,
                                  const int* n_dev, int n_host,
                                  int* __restrict__ seg_start) {
    const int n = LiveCount(n_dev, n_host);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += gridDim.x * blockDim.x) {
        if (j == 0 || sorted_voxel[j] != sorted_voxel[j - 1])
            seg_start[sorted_voxel[j]] = j;
        if (j == n - 1) seg_start[sorted_voxel[j] + 1] = n;
    }
}

template <typename T>
__global__ void VdsReduceKernel(const T* __restrict__ pos,
                                const T* __restrict__ nrm,
                                const unsigned* __restrict__ sorted_point,
                                const int* __restrict__ seg_start,
                                const int* __restrict__ m_dev,
                                T* __restrict__ out_pos,
                                T* __restrict__ out_nrm) {
    const int m = *m_dev;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < m;
         v += gridDim.x * blockDim.x) {
        const int b = seg_start[v], e = seg_start[v + 1];
        float cnt = 0.f, sp[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        for (int j = b; j < e; ++j) {
            const int64_t i = sorted_point[j];
            cnt += 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sp[c] += (float)pos[3 * i + c];
                if (nrm) sn[c] += (float)nrm[3 * i + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out_pos[3 * (int64_t)v + c] = (T)(sp[c] / cnt);
            if (nrm) out_nrm[3 * (int64_t)v + c] = (T)(sn[c] / cnt);
        }
    }
}

template <typename T>
int VdsAsyncImpl(const T* pos, const T* nrm, int64_t n_max, const int* n_dev,
                 double voxel_size, T* out_pos, T* out_nrm, int* m_dev,
                 int* err_dev, std::vector<void*>& scratch, hipStream_t s) {
    O3DMI_REQUIRE(n_max > 0 && n_max < (1ll << 30),
                  "VoxelDownSample: bad point count");
    auto alloc = [&](auto** p, size_t count) -> int {
        void* q = nullptr;
        int e = PoolAlloc(&q, sizeof(**p) * (count ? count : 1));
        if (e) return e;
        scratch.push_back(q);
        *p = (std::remove_reference_t<decltype(**p)>*)q;
        return O3DMI_OK;
    };
    const int n_host = (int)n_max;
    int64_t n_slots = 1024;
    while (n_slots < 2 * n_max) n_slots <<= 1;
    const int n_tiles = (int)((n_max + kSortTile - 1) / kSortTile);
    VdsTable tb;
    int st;
    if ((st = alloc(&tb.keys, (size_t)n_slots))) return st;
    if ((st = alloc(&tb.first, (size_t)n_slots))) return st;
    if ((st = alloc(&tb.voxel, (size_t)n_slots))) return st;
    tb.mask = (unsigned)(n_slots - 1);
    int *slot_of_point, *tile_counts, *hist, *seg_start;
    unsigned *keys_a, *vals_a, *keys_b, *vals_b;
    if ((st = alloc(&slot_of_point, (size_t)n_max))) return st;
    if ((st = alloc(&tile_counts, (size_t)n_tiles))) return st;
    if ((st = alloc(&hist, (size_t)kSortBins * n_tiles))) return st;
    if ((st = alloc(&seg_start, (size_t)n_max + 2))) return st;
    if ((st = alloc(&keys_a, (size_t)n_max))) return st;
    if ((st = alloc(&vals_a, (size_t)n_max))) return st;
    if ((st = alloc(&keys_b, (size_t)n_max))) return st;
    if ((st = alloc(&vals_b, (size_t)n_max))) return st;
    O3DMI_HIP_CHECK(hipMemsetAsync(tb.keys, 0xFF,
                                   sizeof(unsigned long long) * (size_t)n_slots,
                                   s));
    O3DMI_HIP_CHECK(hipMemsetAsync(tb.first, 0x7F,
                                   sizeof(int) * (size_t)n_slots, s));
    const dim3 grid(GridFor(n_max, kBlock)), block(kBlock);
    const dim3 tiles((unsigned)n_tiles), sblock(kSortBlock);
    hipLaunchKernelGGL(VdsInsertKernel<T>, grid, block, 0, s, pos, n_dev,
                       n_host, (T)voxel_size, tb, slot_of_point, err_dev);
    hipLaunchKernelGGL(VdsCountFirstKernel, tiles, sblock, 0, s, slot_of_point,
                       n_dev, n_host, tb, tile_counts);
    hipLaunchKernelGGL(VdsAssignKernel, tiles, sblock, 0, s, slot_of_point,
                       n_dev, n_host, tb, tile_counts, m_dev);
    // keys = dense voxel ids < n_max: as many 11-bit passes as they need
    int bits = 1;
    while ((1ll << bits) < n_max) ++bits;
    const int passes = (bits + kSortBits - 1) / kSortBits;
    unsigned *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * kSortBits;
        if (p == 0)
            hipLaunchKernelGGL(SortHistKernel<true>, tiles, sblock, 0, s,
                               slot_of_point, tb, ki, vi, n_dev, n_host, shift,
                               n_tiles, hist);
        else
            hipLaunchKernelGGL(SortHistKernel<false>, tiles, sblock, 0, s,
                               slot_of_point, tb, ki, vi, n_dev, n_host, shift,
                               n_tiles, hist);
        hipLaunchKernelGGL(SortScatterKernel, tiles, sblock, 0, s, ki, vi, ko,
                           vo, n_dev, n_host, shift, n_tiles, hist);
        std::swap(ki, ko);
        std::swap(vi, vo);
    }
    hipLaunchKernelGGL(VdsSegmentsKernel, grid, block, 0, s, ki, n_dev, n_host,
                       seg_start);
    hipLaunchKernelGGL(VdsReduceKernel<T>, grid, block, 0, s, pos, nrm, vi,
                       seg_start, m_dev, out_pos, out_nrm);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace

int VdsAsync(const void* pos, const void* attr, int64_t n_max, const int* n_dev,
             int dtype, double voxel_size, void* out_pos, void* out_attr,
             int* m_dev, int* err_dev, std::vector<void*>& scratch,
             hipStream_t s) {
    if (dtype == O3DMI_F64)
        return VdsAsyncImpl<double>((const double*)pos, (const double*)attr,
                                    n_max, n_dev, voxel_size, (double*)out_pos,
                                    (double*)out_attr, m_dev, err_dev, scratch,
                                    s);
    return VdsAsyncImpl<float>((const float*)pos, (const float*)attr, n_max,
                               n_dev, voxel_size, (float*)out_pos,
                               (float*)out_attr, m_dev, err_dev, scratch, s);
}

}  // namespace o3dmi

using namespace o3dmi;

extern "C" int o3dmi_voxel_down_sample(const void* positions_dev,
                                       const void* normals_dev, int64_t n,
                                       int dtype, double voxel_size,
                                       void* out_positions_dev,
                                       void* out_normals_dev, int64_t* m_out,
                                       o3dmi_stream_t stream) {
    O3DMI_REQUIRE(m_out && (n == 0 || (positions_dev && out_positions_dev)),
                  "null argument");
    O3DMI_REQUIRE(!normals_dev || out_normals_dev, "out_normals is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (voxel_size <= 0) {
        SetLastError("voxel_size must be positive.");
        return O3DMI_ERR_INVALID_ARG;
    }
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    *m_out = 0;
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    std::vector<void*> scratch;
    struct Release {
        std::vector<void*>& v;
        hipStream_t s;
        ~Release() {
            (void)hipStreamSynchronize(s);  // pooled blocks: stream drained
            for (void* p : v) PoolFree(p);
        }
    } release{scratch, s};
    int* counts = nullptr;  // {voxel count, error flags}
    void* q = nullptr;
    int st = PoolAlloc(&q, sizeof(int) * 4);
    if (st) return st;
    scratch.push_back(q);
    counts = (int*)q;
    O3DMI_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * 4, s));
    st = VdsAsync(positions_dev, normals_dev, n, nullptr, dtype, voxel_size,
                  out_positions_dev, out_normals_dev, counts, counts + 1,
                  scratch, s);
    if (st) return st;
    int host[2] = {0, 0};
    O3DMI_HIP_CHECK(hipMemcpyAsync(host, counts, sizeof(host),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (host[1] & kErrKeyRange) {
        SetLastError("VoxelDownSample: voxel coordinate outside +-2^20");
        return O3DMI_ERR_KEY_RANGE;
    }
    *m_out = host[0];
    return O3DMI_OK;
}
