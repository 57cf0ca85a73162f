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
//      slot's smallest point index (atomicMin): the voxel's FIRST POINT;
//   2. STABLE counting sort of the points by the index of their voxel's first
//      point, least-significant digit first (<= 9 bits per digit, two passes
//      up to 2^18 points; per pass: block histograms -> scatter that sums the
//      (tile, digit) table down its columns for its own offsets and ranks
//      equal digits by wave, round and lane): every voxel's points end up
//      contiguous, still in point order, and the voxels in the order of their
//      first points -- which is the output order;
//   3. the rank of every first point among the first points (= the output row
//      of its voxel) falls out of the same two launches of pass 0: the
//      histogram launch counts the first points per tile, the scatter launch
//      (which walks the points in index order) adds the preceding tiles'
//      counts to a tile-local prefix;
//   4. one lane per sorted element: a lane whose key differs from its left
//      neighbour's starts a run, walks it (eight elements in flight at a
//      time) adding sequentially in float32, and writes the voxel's row.
// Seven launches per level (clear the table, insert, 2 x (histogram,
// scatter), reduce).
//
// Nothing in the chain needs a host decision: the point count may live on the
// device (the output count of the previous, finer level), every kernel bounds
// itself by it, and the voxel count is left on the device as well. A pyramid
// of several levels is therefore ONE string of launches with a single read-
// back at its end (VdsAsync, used by the ICP driver); the public entry point
// runs one level and reads the count back. History: a generic radix sort (16
// launches) and two host waits per level took 1.3 ms of a 1.7 ms tracking
// frame at VGA; dense voxel ids from a separate two-launch scan before the
// sort, a segment-start launch after it and 2048-element tiles (whose scatter
// spent 26 of its 32 us summing the offset table) came next.

#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "o3d_mi355x_host.h"
#include "vds.h"

namespace o3dmi {
namespace {

constexpr int kSortBits = 9;                // at most; see SortPlan
constexpr int kSortBins = 1 << kSortBits;
constexpr int kSortBlock = 1024;            // 16 waves
constexpr int kSortWaves = kSortBlock / 64;
constexpr int kSortItems = 8;               // elements per thread
constexpr int kSortTile = kSortBlock * kSortItems;  // 8192 elements per block

struct VdsTable {
    unsigned long long* keys;  // [n_slots], kEmptyKey when free
    int* first;                // [n_slots] smallest point index
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
    // The live count comes from device memory (the previous level wrote it):
    // the points are fetched alongside it, bounded by the buffer size, and
    // dropped afterwards if they turn out to lie past it -- one memory round
    // trip less at the head of every kernel of the chain.
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_host;
         i += gridDim.x * blockDim.x) {
        const T px = pos[3 * (int64_t)i + 0], py = pos[3 * (int64_t)i + 1],
                pz = pos[3 * (int64_t)i + 2];
        if (i >= LiveCount(n_dev, n_host)) break;
        // (p / vs).Floor().To(Int64)
        const long long cx = (long long)floor(px / vs);
        const long long cy = (long long)floor(py / vs);
        const long long cz = (long long)floor(pz / vs);
        if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias ||
            cy >= kKeyBias || cz < -kKeyBias || cz >= kKeyBias) {
            // reported to the caller; the point stays a voxel of its own so
            // that the rest of the chain sees consistent keys
            atomicOr(err, kErrKeyRange);
            slot_of_point[i] = -1;
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

// Block-wide exclusive prefix of one value per thread (kSortBlock threads);
// returns the prefix, *total = sum over the block. lds4: kSortWaves ints.
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

// ---- stable counting sort, one digit of <= 9 bits per pass --------------------
// A block owns kSortTile consecutive elements; wave w of the block owns the
// elements [w * 512, (w + 1) * 512) of the tile, visited in 8 rounds of 64
// (element = tile + w * 512 + round * 64 + lane), so (block, wave, round, lane)
// is index order. hist is tile-major: hist[tile * bins + digit] (both the
// histogram pass and the scatter's column sums touch it coalesced).
//
// The digit width follows the key range (SortPlan): two passes up to 2^18
// keys, ceil(bits / 9) beyond. Tiles are large (8192 elements) because every
// scatter block sums the (tile, digit) table down its columns for its own
// offsets: 512 columns x n_tiles rows, read once per block with all loads of a
// column in flight -- 10 rows for a VGA frame cloud, 29 at 720p. (The first
// version had 2048-element tiles and 2048 digits; that sum was 8 x n_tiles
// dependent loads per thread and took 26 of the pass's 32 us.)

struct SortPlan {
    int passes, bits;  // bits per digit
};
inline SortPlan PlanSort(int64_t n_keys) {
    int key_bits = 1;
    while ((1ll << key_bits) < n_keys) ++key_bits;
    SortPlan p;
    p.passes = key_bits <= 2 * kSortBits ? 2
                                         : (key_bits + kSortBits - 1) / kSortBits;
    p.bits = (key_bits + p.passes - 1) / p.passes;
    return p;
}

// kMakeKeys: pass 0 also creates the keys (index of the first point of every
// point's voxel) and the values (point indices), and counts the tile's first
// points (key == own index).
template <bool kMakeKeys>
__global__ void __launch_bounds__(kSortBlock)
SortHistKernel(const int* __restrict__ slot_of_point, VdsTable tb,
               unsigned* __restrict__ keys, unsigned* __restrict__ vals,
               const int* n_dev, int n_host, int shift, int bits,
               int* __restrict__ hist, int* __restrict__ tile_firsts) {
    __shared__ int h[kSortBins];
    __shared__ int firsts;
    const int base = blockIdx.x * kSortTile;
    // fetched alongside the live count (see VdsInsertKernel)
    unsigned key[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + k * kSortBlock + threadIdx.x;
        key[k] = 0xFFFFFFFFu;
        if (i < n_host)
            key[k] = kMakeKeys ? (unsigned)slot_of_point[i] : keys[i];
    }
    const int n = LiveCount(n_dev, n_host);
    if (base >= n) return;
    const int bins = 1 << bits;
    for (int b = threadIdx.x; b < bins; b += kSortBlock) h[b] = 0;
    if (threadIdx.x == 0) firsts = 0;
    __syncthreads();
    if constexpr (kMakeKeys) {
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const int i = base + k * kSortBlock + threadIdx.x;
            const int slot = (int)key[k];
            // a slot read past the live count is whatever the buffer held
            key[k] = i < n && slot >= 0 ? (unsigned)tb.first[slot & tb.mask]
                                        : (unsigned)i;
        }
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const int i = base + k * kSortBlock + threadIdx.x;
        if (i < n) {
            if constexpr (kMakeKeys) {
                keys[i] = key[k];
                vals[i] = (unsigned)i;
                mine += key[k] == (unsigned)i;
            }
            atomicAdd(&h[(key[k] >> shift) & (bins - 1)], 1);
        }
    }
    if constexpr (kMakeKeys) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) mine += __shfl_xor(mine, m);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&firsts, mine);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += kSortBlock)
        hist[(int64_t)blockIdx.x * bins + b] = h[b];
    if constexpr (kMakeKeys)
        if (threadIdx.x == 0) tile_firsts[blockIdx.x] = firsts;
}

// kRank (pass 0, which walks the points in index order): also numbers the first
// points -- rank_of_first[i] = how many first points precede point i = the
// output row of the voxel that point i opens -- and block 0 publishes their
// total, the voxel count.
template <bool kRank>
__global__ void __launch_bounds__(kSortBlock)
SortScatterKernel(const unsigned* __restrict__ keys_in,
                  const unsigned* __restrict__ vals_in,
                  unsigned* __restrict__ keys_out,
                  unsigned* __restrict__ vals_out, const int* n_dev, int n_host,
                  int shift, int bits, const int* __restrict__ hist,
                  const int* __restrict__ tile_firsts,
                  int* __restrict__ rank_of_first, int* __restrict__ m_dev) {
    __shared__ int wh[kSortWaves][kSortBins];  // 32 KiB
    __shared__ int dbase[kSortBins];
    __shared__ int lds4[kSortWaves];
    __shared__ int wave_firsts[kSortWaves];
    __shared__ int firsts_before;
    const int tile = blockIdx.x * kSortTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wbase = tile + wave * (kSortItems * 64);
    // fetched alongside the live count (see VdsInsertKernel)
    unsigned key[kSortItems], val[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        key[r] = 0;
        val[r] = 0;
        if (e < n_host) {
            key[r] = keys_in[e];
            val[r] = vals_in[e];
        }
    }
    const int n = LiveCount(n_dev, n_host);
    if (tile >= n) return;
    const int bins = 1 << bits;
    const int n_tiles = (n + kSortTile - 1) / kSortTile;
    for (int b = threadIdx.x; b < kSortBins * kSortWaves; b += kSortBlock)
        (&wh[0][0])[b] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r)
        if (wbase + r * 64 + lane < n)
            atomicAdd(&wh[wave][(key[r] >> shift) & (bins - 1)], 1);
    // Where this tile's run of every digit starts: all elements with a smaller
    // digit (over all tiles) + the same digit in the tiles before this one.
    // Thread d sums column d of the table, sixteen rows in flight at a time.
    {
        int all = 0, before = 0;
        if ((int)threadIdx.x < bins) {
            const int* col = hist + threadIdx.x;
            for (int t0 = 0; t0 < n_tiles; t0 += 16) {
                int c[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    c[u] = t0 + u < n_tiles ? col[(int64_t)(t0 + u) * bins] : 0;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    all += c[u];
                    before += t0 + u < (int)blockIdx.x ? c[u] : 0;
                }
            }
        }
        if constexpr (kRank) {
            // first points in the tiles before this one (last wave: its lanes
            // are not needed for the columns when bins <= 512)
            if (wave == kSortWaves - 1) {
                int fb = 0, fa = 0;
                for (int t = lane; t < n_tiles; t += 64) {
                    const int c = tile_firsts[t];
                    fa += c;
                    fb += t < (int)blockIdx.x ? c : 0;
                }
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) {
                    fa += __shfl_xor(fa, m);
                    fb += __shfl_xor(fb, m);
                }
                if (lane == 0) {
                    firsts_before = fb;
                    if (blockIdx.x == 0) *m_dev = fa;
                }
            }
        }
        int total;
        const int run = BlockExclusive(all, lds4, &total);
        if ((int)threadIdx.x < bins) dbase[threadIdx.x] = run + before;
    }
    __syncthreads();
    // per digit: where each wave's run starts in the output
    for (int b = threadIdx.x; b < bins; b += kSortBlock) {
        int off = dbase[b];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const int c = wh[w][b];
            wh[w][b] = off;
            off += c;
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    if constexpr (kRank) {
        // first points of this wave's 512 elements, in element order
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            const int e = wbase + r * 64 + lane;
            cnt += __popcll(__ballot(e < n && key[r] == (unsigned)e));
        }
        if (lane == 0) wave_firsts[wave] = cnt;
    }
    __syncthreads();
    if constexpr (kRank) {
        int rank = firsts_before;
        for (int w = 0; w < wave; ++w) rank += wave_firsts[w];
#pragma unroll
        for (int r = 0; r < kSortItems; ++r) {
            const int e = wbase + r * 64 + lane;
            const bool is_first = e < n && key[r] == (unsigned)e;
            const unsigned long long fm = __ballot(is_first);
            if (is_first) rank_of_first[e] = rank + __popcll(fm & lt);
            rank += __popcll(fm);
        }
    }
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        const bool valid = e < n;
        const unsigned d = (key[r] >> shift) & (bins - 1);
        // lanes of this round holding the same digit
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
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

// Empty hash table, no first points: one launch instead of two fills.
__global__ void VdsInitKernel(VdsTable tb, int64_t n_slots) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        tb.keys[i] = kEmptyKey;
        tb.first[i] = 0x7FFFFFFF;
    }
}

// One lane per sorted element; run starts (key != left neighbour's key) walk
// their run, kRun elements in flight at a time, adding in float32 in element
// (= point) order; the row comes from the run's key, the index of the voxel's
// first point.
constexpr int kRun = 8;
template <typename T>
__global__ void VdsReduceKernel(const T* __restrict__ pos,
                                const T* __restrict__ nrm,
                                const unsigned* __restrict__ sorted_key,
                                const unsigned* __restrict__ sorted_point,
                                const int* __restrict__ rank_of_first,
                                const int* n_dev, int n_host,
                                T* __restrict__ out_pos,
                                T* __restrict__ out_nrm) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_host;
         j += gridDim.x * blockDim.x) {
        // fetched alongside the live count (see VdsInsertKernel)
        const unsigned k = sorted_key[j];
        const unsigned left = j > 0 ? sorted_key[j - 1] : 0u;
        const int n = LiveCount(n_dev, n_host);
        if (j >= n) break;
        if (j > 0 && left == k) continue;
        const int v = rank_of_first[k];
        float cnt = 0.f, sp[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        bool more = true;
        for (int j0 = j; more; j0 += kRun) {
            unsigned kk[kRun], pi[kRun];
#pragma unroll
            for (int u = 0; u < kRun; ++u) {
                const int jj = j0 + u < n ? j0 + u : n - 1;
                kk[u] = sorted_key[jj];
                pi[u] = sorted_point[jj];
            }
            float p[kRun][3], q[kRun][3];
#pragma unroll
            for (int u = 0; u < kRun; ++u) {
                // elements past the run read the run's first point (in cache)
                const int64_t i = kk[u] == k ? pi[u] : k;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p[u][c] = (float)pos[3 * i + c];
                    q[u][c] = nrm ? (float)nrm[3 * i + c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < kRun; ++u) {
                more = more && j0 + u < n && kk[u] == k;
                if (more) {
                    cnt += 1.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sp[c] += p[u][c];
                        sn[c] += q[u][c];
                    }
                }
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
    tb.mask = (unsigned)(n_slots - 1);
    int *slot_of_point, *tile_firsts, *hist, *rank_of_first;
    unsigned *keys_a, *vals_a, *keys_b, *vals_b;
    if ((st = alloc(&slot_of_point, (size_t)n_max))) return st;
    if ((st = alloc(&tile_firsts, (size_t)n_tiles))) return st;
    if ((st = alloc(&hist, (size_t)kSortBins * n_tiles))) return st;
    if ((st = alloc(&rank_of_first, (size_t)n_max))) return st;
    if ((st = alloc(&keys_a, (size_t)n_max))) return st;
    if ((st = alloc(&vals_a, (size_t)n_max))) return st;
    if ((st = alloc(&keys_b, (size_t)n_max))) return st;
    if ((st = alloc(&vals_b, (size_t)n_max))) return st;
    const dim3 grid(GridFor(n_max, kBlock)), block(kBlock);
    const dim3 tiles((unsigned)n_tiles), sblock(kSortBlock);
    hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)), block, 0,
                       s, tb, n_slots);
    hipLaunchKernelGGL(VdsInsertKernel<T>, grid, block, 0, s, pos, n_dev,
                       n_host, (T)voxel_size, tb, slot_of_point, err_dev);
    // keys = index of the voxel's first point < n_max
    const SortPlan plan = PlanSort(n_max);
    unsigned *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    for (int p = 0; p < plan.passes; ++p) {
        const int shift = p * plan.bits;
        if (p == 0) {
            hipLaunchKernelGGL(SortHistKernel<true>, tiles, sblock, 0, s,
                               slot_of_point, tb, ki, vi, n_dev, n_host, shift,
                               plan.bits, hist, tile_firsts);
            hipLaunchKernelGGL(SortScatterKernel<true>, tiles, sblock, 0, s, ki,
                               vi, ko, vo, n_dev, n_host, shift, plan.bits,
                               hist, tile_firsts, rank_of_first, m_dev);
        } else {
            hipLaunchKernelGGL(SortHistKernel<false>, tiles, sblock, 0, s,
                               slot_of_point, tb, ki, vi, n_dev, n_host, shift,
                               plan.bits, hist, tile_firsts);
            hipLaunchKernelGGL(SortScatterKernel<false>, tiles, sblock, 0, s,
                               ki, vi, ko, vo, n_dev, n_host, shift, plan.bits,
                               hist, tile_firsts, rank_of_first, m_dev);
        }
        std::swap(ki, ko);
        std::swap(vi, vo);
    }
    hipLaunchKernelGGL(VdsReduceKernel<T>, grid, block, 0, s, pos, nrm, ki, vi,
                       rank_of_first, n_dev, n_host, out_pos, out_nrm);
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
