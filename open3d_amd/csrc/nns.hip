// Nearest-neighbour index and searches for MI355X.
//
//   o3dmi_nns_create              <- BuildSpatialHashTableCUDA (core/nns/FixedRadiusIndex.h:227-233,
//                                    FixedRadiusSearchImpl.cuh:63-135,732-824)
//   o3dmi_nns_hybrid_search_k1,   <- HybridSearchCUDA (FixedRadiusIndex.h:364-377,
//   o3dmi_nns_hybrid_search          FixedRadiusSearchImpl.cuh:514-632) with the CPU path's
//                                    nanoflann semantics (core/nns/NanoFlannImpl.h:305-370)
//   o3dmi_nns_knn_search          <- KnnSearchCUDA (core/nns/KnnIndex.h; NanoFlannImpl.h:129-203)
//   o3dmi_nns_radius_covariances  <- EstimateCovariancesUsingRadiusSearchCUDA
//                                    (t/geometry/kernel/PointCloudImpl.h:641-689)
//
// See nns.h for the index layout. The fused ICP iteration kernel that searches
// and accumulates in one launch lives in icp.hip.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nns.h"

namespace o3dmi {
namespace {

// K1: bucket histogram into ranges[b].y.
template <typename T>
__global__ void CountKernel(const T* __restrict__ pts, int64_t n,
                            double inv_cell, unsigned mask,
                            uint2* __restrict__ ranges) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        long long cx, cy, cz;
        CellOf(pts + 3 * i, inv_cell, cx, cy, cz);
        atomicAdd(&ranges[HashCell(cx, cy, cz) & mask].y, 1u);
    }
}

// K2: every bucket gets a contiguous range of records: a workgroup sums the
// counts of its 1024 buckets (wave prefix, then the 16 wave totals through
// LDS), one lane takes that many records off the running total
// (ranges[n_buckets].x -- one atomic per 1024 buckets: same-address atomics
// serialise, and a table has up to 2^27 buckets), the prefix places the
// buckets. ranges[b].x = start here; the scatter moves it to the end.
// n_buckets is a power of two >= 1024, so every workgroup is full.
constexpr int kAssignBlock = 1024;
__global__ void __launch_bounds__(kAssignBlock)
AssignRangesKernel(uint2* __restrict__ ranges, int64_t n_buckets,
                   int* __restrict__ tickets) {
    __shared__ unsigned wave_total[kAssignBlock / 64];
    __shared__ unsigned block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the search launches' final-sum tickets start at zero (a fill launch per
    // index build until round 5)
    if (tickets && blockIdx.x == 0 && threadIdx.x < 16) tickets[threadIdx.x] = 0;
    for (int64_t b0 = (int64_t)blockIdx.x * kAssignBlock; b0 < n_buckets;
         b0 += (int64_t)gridDim.x * kAssignBlock) {
        const int64_t b = b0 + threadIdx.x;
        const unsigned c = ranges[b].y;
        unsigned incl = c;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned o = __shfl_up(incl, m);
            if (lane >= m) incl += o;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kAssignBlock / 64; ++w) {
            const unsigned t = wave_total[w];
            before += w < wave ? t : 0u;
            total += t;
        }
        if (threadIdx.x == 0)
            block_base = total ? atomicAdd(&ranges[n_buckets].x, total) : 0u;
        __syncthreads();
        ranges[b].x = block_base + before + incl - c;
        __syncthreads();  // block_base / wave_total are reused
    }
}

// K3: scatter points (and, when given, their normals) into bucket order as
// {x,y,z,idx} records.
template <typename T>
__global__ void ScatterKernel(const T* __restrict__ pts,
                              const T* __restrict__ normals, int64_t n,
                              double inv_cell, unsigned mask,
                              uint2* __restrict__ ranges,
                              Rec4<T>* __restrict__ sorted,
                              Rec4<T>* __restrict__ sorted_normals) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Rec4<T> r;
        r.x = pts[3 * i + 0];
        r.y = pts[3 * i + 1];
        r.z = pts[3 * i + 2];
        r.w = i;
        const T p[3] = {r.x, r.y, r.z};
        long long cx, cy, cz;
        CellOf(p, inv_cell, cx, cy, cz);
        const unsigned b = HashCell(cx, cy, cz) & mask;
        const unsigned pos = atomicAdd(&ranges[b].x, 1u);
        sorted[pos] = r;
        if (normals) {
            Rec4<T> m;
            m.x = normals[3 * i + 0];
            m.y = normals[3 * i + 1];
            m.z = normals[3 * i + 2];
            m.w = 0;
            sorted_normals[pos] = m;
        }
    }
}

// ---- several indices in the same launches (round 6) -----------------------------
// The ICP driver builds the indices of its later scales while the first scale
// iterates. One build is a clearing fill and three small launches (plus four
// pool allocations): ~20 us of HOST time, issued behind an 11 us search launch
// whose sums the host should be waiting for -- two builds cost a tracked frame
// two late hops. Here up to kIndexJobs indices share every launch
// (blockIdx.y = index): clear, count, assign, scatter -- four launches for all
// of them, the fill one of ours instead of a runtime call per index.
constexpr int kIndexJobs = 4;
template <typename T>
struct IndexJob {
    const T* pts;
    const T* normals;
    int64_t n;
    double inv_cell;
    unsigned mask;
    int64_t nb;
    uint2* ranges;
    Rec4<T>* sorted;
    Rec4<T>* sorted_normals;
    int* tickets;
};
template <typename T>
__device__ __forceinline__ const IndexJob<T>& PickJob(const IndexJob<T>& a,
                                                      const IndexJob<T>& b,
                                                      const IndexJob<T>& c,
                                                      const IndexJob<T>& d) {
    return blockIdx.y == 0 ? a : (blockIdx.y == 1 ? b : (blockIdx.y == 2 ? c : d));
}

template <typename T>
__global__ void ClearManyKernel(IndexJob<T> a, IndexJob<T> b, IndexJob<T> c,
                                IndexJob<T> d) {
    const IndexJob<T>& job = PickJob(a, b, c, d);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         i <= job.nb; i += (int64_t)gridDim.x * blockDim.x)
        job.ranges[i] = make_uint2(0u, 0u);
}

template <typename T>
__global__ void CountManyKernel(IndexJob<T> a, IndexJob<T> b, IndexJob<T> c,
                                IndexJob<T> d) {
    const IndexJob<T>& job = PickJob(a, b, c, d);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < job.n;
         i += (int64_t)gridDim.x * blockDim.x) {
        long long cx, cy, cz;
        CellOf(job.pts + 3 * i, job.inv_cell, cx, cy, cz);
        atomicAdd(&job.ranges[HashCell(cx, cy, cz) & job.mask].y, 1u);
    }
}

// AssignRangesKernel's body for one index
__device__ __forceinline__ void AssignRangesBody(uint2* __restrict__ ranges,
                                                 int64_t n_buckets,
                                                 int* __restrict__ tickets) {
    __shared__ unsigned wave_total_m[kAssignBlock / 64];
    __shared__ unsigned block_base_m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tickets && blockIdx.x == 0 && threadIdx.x < 16) tickets[threadIdx.x] = 0;
    for (int64_t b0 = (int64_t)blockIdx.x * kAssignBlock; b0 < n_buckets;
         b0 += (int64_t)gridDim.x * kAssignBlock) {
        const int64_t b = b0 + threadIdx.x;
        const unsigned c = ranges[b].y;
        unsigned incl = c;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned o = __shfl_up(incl, m);
            if (lane >= m) incl += o;
        }
        if (lane == 63) wave_total_m[wave] = incl;
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kAssignBlock / 64; ++w) {
            const unsigned t = wave_total_m[w];
            before += w < wave ? t : 0u;
            total += t;
        }
        if (threadIdx.x == 0)
            block_base_m = total ? atomicAdd(&ranges[n_buckets].x, total) : 0u;
        __syncthreads();
        ranges[b].x = block_base_m + before + incl - c;
        __syncthreads();
    }
}

template <typename T>
__global__ void __launch_bounds__(kAssignBlock)
AssignManyKernel(IndexJob<T> a, IndexJob<T> b, IndexJob<T> c, IndexJob<T> d) {
    const IndexJob<T>& job = PickJob(a, b, c, d);
    AssignRangesBody(job.ranges, job.nb, job.tickets);
}

template <typename T>
__global__ void ScatterManyKernel(IndexJob<T> a, IndexJob<T> b, IndexJob<T> c,
                                  IndexJob<T> d) {
    const IndexJob<T>& job = PickJob(a, b, c, d);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < job.n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Rec4<T> r;
        r.x = job.pts[3 * i + 0];
        r.y = job.pts[3 * i + 1];
        r.z = job.pts[3 * i + 2];
        r.w = i;
        const T p[3] = {r.x, r.y, r.z};
        long long cx, cy, cz;
        CellOf(p, job.inv_cell, cx, cy, cz);
        const unsigned bk = HashCell(cx, cy, cz) & job.mask;
        const unsigned pos = atomicAdd(&job.ranges[bk].x, 1u);
        job.sorted[pos] = r;
        if (job.normals) {
            Rec4<T> m;
            m.x = job.normals[3 * i + 0];
            m.y = job.normals[3 * i + 1];
            m.z = job.normals[3 * i + 2];
            m.w = 0;
            job.sorted_normals[pos] = m;
        }
    }
}

// K1 - K3 in ONE launch for a small cloud (the coarsest level of an ICP
// pyramid: 2 - 3 k points): a single workgroup counts into LDS, scans the
// buckets there, writes every range (no clearing launch before it) and
// scatters the records, cursors in LDS. The four launches it replaces are 3 -
// 4 us each with the launch latency of a dependent chain between them -- 35 us
// on the critical path of every tracked frame, between the pyramid and the
// first search. The records of a bucket come out in any order, as from K3 (the
// searches break ties by original index).
constexpr int kSmallIndexPoints = 4096;          // => at most 8192 buckets
constexpr int kSmallIndexBuckets = 2 * kSmallIndexPoints;
constexpr int kSmallIndexBlock = 1024;
template <typename T>
__global__ void __launch_bounds__(kSmallIndexBlock)
BuildSmallIndexKernel(const T* __restrict__ pts, const T* __restrict__ normals,
                      int n, const int* __restrict__ n_dev, double inv_cell,
                      unsigned mask, int n_buckets, uint2* __restrict__ ranges,
                      Rec4<T>* __restrict__ sorted,
                      Rec4<T>* __restrict__ sorted_normals,
                      int* __restrict__ tickets) {
    __shared__ unsigned cnt[kSmallIndexBuckets];
    __shared__ unsigned wave_total[kSmallIndexBlock / 64];
    constexpr int kMine = kSmallIndexPoints / kSmallIndexBlock;  // points / thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tickets && threadIdx.x < 16) tickets[threadIdx.x] = 0;
    // n_dev: the size lives on the device (the ICP driver queues this build
    // behind the pyramid BEFORE it has read the level sizes back); a cloud
    // that turns out too large for this launch is left unbuilt -- the host,
    // which learns the size a moment later, builds it the long way
    if (n_dev) {
        n = *n_dev;
        if (n < 0 || n > kSmallIndexPoints) return;
    }
    for (int b = threadIdx.x; b < n_buckets; b += kSmallIndexBlock) cnt[b] = 0;
    __syncthreads();
    Rec4<T> rec[kMine];
    unsigned bucket[kMine];
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int i = k * kSmallIndexBlock + (int)threadIdx.x;
        bucket[k] = 0;
        if (i < n) {
            rec[k].x = pts[3 * (int64_t)i + 0];
            rec[k].y = pts[3 * (int64_t)i + 1];
            rec[k].z = pts[3 * (int64_t)i + 2];
            rec[k].w = i;
            const T p[3] = {rec[k].x, rec[k].y, rec[k].z};
            long long cx, cy, cz;
            CellOf(p, inv_cell, cx, cy, cz);
            bucket[k] = HashCell(cx, cy, cz) & mask;
            atomicAdd(&cnt[bucket[k]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan: a thread owns n_buckets / 1024 consecutive buckets
    const int per = n_buckets / kSmallIndexBlock;  // 1, 2, 4 or 8
    const int b0 = (int)threadIdx.x * per;
    unsigned mine = 0;
    for (int k = 0; k < per; ++k) mine += cnt[b0 + k];
    unsigned incl = mine;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned o = __shfl_up(incl, m);
        if (lane >= m) incl += o;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    unsigned run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wave_total[w];
    for (int k = 0; k < per; ++k) {
        const unsigned c = cnt[b0 + k];
        ranges[b0 + k] = make_uint2(run + c, c);  // {end, count}: see BucketRange
        cnt[b0 + k] = run;                         // the scatter's cursor
        run += c;
    }
    if (threadIdx.x == 0) ranges[n_buckets] = make_uint2((unsigned)n, 0u);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
        const int i = k * kSmallIndexBlock + (int)threadIdx.x;
        if (i < n) {
            const unsigned pos = atomicAdd(&cnt[bucket[k]], 1u);
            sorted[pos] = rec[k];
            if (normals) {
                Rec4<T> m;
                m.x = normals[3 * (int64_t)i + 0];
                m.y = normals[3 * (int64_t)i + 1];
                m.z = normals[3 * (int64_t)i + 2];
                m.w = 0;
                sorted_normals[pos] = m;
            }
        }
    }
}

template <typename T>
__global__ void GatherAttrKernel(const T* __restrict__ attr,
                                 const Rec4<T>* __restrict__ sorted_pts,
                                 int64_t n, Rec4<T>* __restrict__ out) {
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = RecIndex(sorted_pts[j]);
        Rec4<T> r;
        r.x = attr[3 * i + 0];
        r.y = attr[3 * i + 1];
        r.z = attr[3 * i + 2];
        r.w = 0;
        out[j] = r;
    }
}

// Nearest neighbour with d2 < r2 (strict), ties -> lowest original index.
// Returns the position in the sorted array (or -1); idx/d2 by reference.
// Distance arithmetic: nanoflann::L2_Adaptor::evalMetric for dim 3,
// ((dx*dx) + dy*dy) + dz*dz with dx = query - point, in T.
template <typename T>
__device__ __forceinline__ int SearchNearest(const NnsView<T>& nv, const T* q,
                                             int& best_idx, T& best_d2) {
    long long cx, cy, cz;
    CellOf(q, nv.inv_cell, cx, cy, cz);
    int best_pos = -1;
    best_idx = -1;
    best_d2 = 0;
    const T qx = q[0], qy = q[1], qz = q[2];
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                unsigned b = HashCell(cx + dx, cy + dy, cz + dz) & nv.mask;
                unsigned s, e;
                BucketRange(nv, b, s, e);
                for (unsigned j = s; j < e; ++j) {
                    Rec4<T> p = nv.sorted[j];
                    T result = T(0);
                    const T d0 = qx - p.x;
                    result += d0 * d0;
                    const T d1 = qy - p.y;
                    result += d1 * d1;
                    const T d2 = qz - p.z;
                    result += d2 * d2;
                    if (result < nv.radius_squared) {
                        int idx = RecIndex(p);
                        if (best_pos < 0 || result < best_d2 ||
                            (result == best_d2 && idx < best_idx)) {
                            best_pos = (int)j;
                            best_idx = idx;
                            best_d2 = result;
                        }
                    }
                }
            }
    return best_pos;
}

template <typename T>
__global__ void HybridSearchK1Kernel(NnsView<T> nv, const T* __restrict__ q,
                                     int64_t nq, int* __restrict__ idx_out,
                                     T* __restrict__ d2_out,
                                     int* __restrict__ cnt_out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq;
         i += (int64_t)gridDim.x * blockDim.x) {
        T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        int idx;
        T d2;
        int pos = SearchNearest(nv, qq, idx, d2);
        if (idx_out) idx_out[i] = pos >= 0 ? idx : -1;
        if (d2_out) d2_out[i] = pos >= 0 ? d2 : T(0);
        if (cnt_out) cnt_out[i] = pos >= 0 ? 1 : 0;
    }
}

constexpr int kMaxKnn = 64;  // general-k searches: k <= one wave

// ---- general-k searches: one wave per query --------------------------------
// A lane-per-query search keeps a sorted k-list per lane and pays a dependent
// chain of ~2 loads per neighbour cell plus an insertion shift per candidate,
// with the 64 lanes of a wave diverging on every one of them (2.7 ms for
// 100 k queries at k = 30 even with the lists in LDS). Here a wave serves one
// query:
//   * lane c looks up neighbour cell c (bucket bounds), a wave prefix sum
//     turns the per-cell counts into one candidate range, and lane t fetches
//     candidate t (owner cell found by a 6-step search over the prefix) --
//     two memory round trips for any number of cells. A record is accepted
//     only if its own cell is the cell it was fetched for, so records that
//     merely share the bucket (hash collisions) never show up twice;
//   * accepted candidates are compacted into an LDS buffer of the wave;
//   * the k best are found by rank counting: candidate p reads every buffered
//     candidate q (a broadcast LDS read) and counts those that sort before it
//     by (d2, index); rank < k means "rank-th neighbour". No dependent chain,
//     no divergence, ties impossible because indices are unique.
constexpr int kCoopCap = 512;    // buffered candidates per wave before a merge
constexpr int kCoopBlock = 256;  // 4 waves = 4 queries in flight per workgroup
constexpr int kNoIndex = 0x7fffffff;

template <typename T>
__device__ __forceinline__ T InfOf() { return (T)INFINITY; }

template <typename T>
__device__ __forceinline__ bool PairLess(T ad, int ai, T bd, int bi) {
    return ad < bd || (ad == bd && ai < bi);
}

__device__ __forceinline__ void WaveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Buffer rows: kCoopCap, + one more batch of 64 (the merge is triggered after
// the batch that crosses kCoopCap), + the current list of <= 64 entries that
// competes again in a merge; then the 64 ranked rows.
constexpr int kCoopRows = kCoopCap + 128;

template <typename T>
constexpr size_t CoopLdsBytesPerWave() {
    return (sizeof(T) + sizeof(int)) * (size_t)(kCoopRows + 64);
}

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; };
template <> struct Quad<double> { using type = double4; };

template <typename T>
struct WaveTopK {
    T* cd;       // LDS: candidates [kCoopRows], then ranked list [64]
    int* ci;
    T* rd;
    int* ri;
    int m;       // buffered candidates (wave-uniform)
    T best_d;    // rank = lane, valid for lane < nbest
    int best_i;
    int nbest;   // wave-uniform
    int knn;
    T kth_d;     // the k-th entry once nbest == knn (wave-uniform)
    int kth_i;

    __device__ __forceinline__ void Init(char* lds) {
        char* base = lds + CoopLdsBytesPerWave<T>() * (threadIdx.x >> 6);
        cd = (T*)base;
        rd = cd + kCoopRows;
        ci = (int*)(rd + 64);
        ri = ci + kCoopRows;
    }

    __device__ __forceinline__ void Reset(int k) {
        m = 0;
        best_d = InfOf<T>();
        best_i = kNoIndex;
        nbest = 0;
        knn = k;
        kth_d = InfOf<T>();
        kth_i = kNoIndex;
    }

    // Candidate of this lane (kNoIndex = none).
    __device__ __forceinline__ void Push(T d, int i, T, T, T) { Push(d, i); }
    __device__ __forceinline__ void Push(T d, int i) {
        // what cannot make the list any more is dropped here
        bool valid = i != kNoIndex;
        if (valid && nbest == knn && !PairLess(d, i, kth_d, kth_i)) valid = false;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(valid);
        if (mask == 0) return;
        const int lane = threadIdx.x & 63;
        const int at = m + __popcll(mask & ((1ull << lane) - 1ull));
        if (valid) {
            cd[at] = d;
            ci[at] = i;
        }
        m += __popcll(mask);
        if (m > kCoopCap) Select();  // room for one more batch of 64 is kept
    }

    __device__ __forceinline__ void Flush() {
        if (m > 0) Select();
    }

    // Merge the buffer into the ranked list.
    __device__ __forceinline__ void Select() {
        const int lane = threadIdx.x & 63;
        // the current list competes again
        int total = m;
        if (lane < nbest) {
            cd[total + lane] = best_d;
            ci[total + lane] = best_i;
        }
        total += nbest;
        // pad to whole quads with entries that sort after everything
        const int padded = (total + 3) & ~3;
        if (lane < padded - total) {
            cd[total + lane] = InfOf<T>();
            ci[total + lane] = kNoIndex;
        }
        WaveLdsSync();
        using Q = typename Quad<T>::type;
        // two candidates per lane per pass over the buffer (quad LDS reads,
        // every lane reads the same address: broadcast)
        for (int p0 = 0; p0 < total; p0 += 128) {
            const int pa = p0 + lane, pb = p0 + 64 + lane;
            const bool ha = pa < total, hb = pb < total;
            const T a_d = ha ? cd[pa] : InfOf<T>();
            const int a_i = ha ? ci[pa] : kNoIndex;
            const T b_d = hb ? cd[pb] : InfOf<T>();
            const int b_i = hb ? ci[pb] : kNoIndex;
            int ra = 0, rb = 0;
#pragma unroll 2
            for (int qi = 0; qi < padded; qi += 4) {
                const Q d4 = *(const Q*)(cd + qi);
                const int4 i4 = *(const int4*)(ci + qi);
                ra += (PairLess(d4.x, i4.x, a_d, a_i) ? 1 : 0) +
                      (PairLess(d4.y, i4.y, a_d, a_i) ? 1 : 0) +
                      (PairLess(d4.z, i4.z, a_d, a_i) ? 1 : 0) +
                      (PairLess(d4.w, i4.w, a_d, a_i) ? 1 : 0);
                rb += (PairLess(d4.x, i4.x, b_d, b_i) ? 1 : 0) +
                      (PairLess(d4.y, i4.y, b_d, b_i) ? 1 : 0) +
                      (PairLess(d4.z, i4.z, b_d, b_i) ? 1 : 0) +
                      (PairLess(d4.w, i4.w, b_d, b_i) ? 1 : 0);
            }
            if (ha && ra < knn) {
                rd[ra] = a_d;
                ri[ra] = a_i;
            }
            if (hb && rb < knn) {
                rd[rb] = b_d;
                ri[rb] = b_i;
            }
        }
        WaveLdsSync();
        nbest = total < knn ? total : knn;
        best_d = lane < nbest ? rd[lane] : InfOf<T>();
        best_i = lane < nbest ? ri[lane] : kNoIndex;
        if (nbest == knn) {
            kth_d = rd[knn - 1];
            kth_i = ri[knn - 1];
        }
        WaveLdsSync();
        m = 0;
    }
};

// Candidates of the cells [x0..x1] x [y0..y1] x [z0..z1] whose Chebyshev cell
// distance to (cx, cy, cz) is >= r_skip. RADIUS: keep d2 < nv.radius_squared.
// Sink::Push(d2, index, x, y, z) receives every lane's candidate of a batch
// (index == kNoIndex: none).
template <typename T, bool RADIUS, typename Sink>
__device__ __forceinline__ void GatherCells(const NnsView<T>& nv, const T* qq,
                                            long long cx, long long cy,
                                            long long cz, long long x0,
                                            long long x1, long long y0,
                                            long long y1, long long z0,
                                            long long z1, long long r_skip,
                                            Sink& list) {
    if (x1 < x0 || y1 < y0 || z1 < z0) return;
    const int lane = threadIdx.x & 63;
    const int nx = (int)(x1 - x0 + 1), ny = (int)(y1 - y0 + 1);
    const int nz = (int)(z1 - z0 + 1);
    const int ncell = nx * ny * nz;
    for (int base = 0; base < ncell; base += 64) {
        const int ci = base + lane;
        unsigned s0 = 0, cnt = 0;
        if (ci < ncell) {
            const long long x = x0 + ci % nx, y = y0 + (ci / nx) % ny,
                            z = z0 + ci / (nx * ny);
            long long ax = x - cx, ay = y - cy, az = z - cz;
            ax = ax < 0 ? -ax : ax;
            ay = ay < 0 ? -ay : ay;
            az = az < 0 ? -az : az;
            const long long cheb = max(ax, max(ay, az));
            if (cheb >= r_skip) {
                const unsigned b = HashCell(x, y, z) & nv.mask;
                unsigned e0;
                BucketRange(nv, b, s0, e0);
                cnt = e0 - s0;
            }
        }
        unsigned incl = cnt;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned o = __shfl_up(incl, m);
            if (lane >= m) incl += o;
        }
        const unsigned total = __shfl(incl, 63);
        const unsigned excl = incl - cnt;
        for (unsigned t0 = 0; t0 < total; t0 += 64) {
            const unsigned t = t0 + lane;
            const unsigned tc = t < total ? t : total - 1;
            // owner cell: number of lanes whose inclusive prefix is <= t
            int pos = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const unsigned v = __shfl(incl, pos + step - 1);
                if (v <= tc) pos += step;
            }
            const unsigned oe = __shfl(excl, pos);
            const unsigned os = __shfl(s0, pos);
            T d = InfOf<T>();
            int pi = kNoIndex;
            T px = T(0), py = T(0), pz = T(0);
            if (t < total) {
                const Rec4<T> p = nv.sorted[os + (t - oe)];
                px = p.x;
                py = p.y;
                pz = p.z;
                // the record's own cell must be the cell it was fetched for
                const T pp[3] = {p.x, p.y, p.z};
                long long rx, ry, rz;
                CellOf(pp, nv.inv_cell, rx, ry, rz);
                const bool own = rx >= x0 && rx <= x1 && ry >= y0 && ry <= y1 &&
                                 rz >= z0 && rz <= z1 &&
                                 (int)(rx - x0) + nx * ((int)(ry - y0) +
                                                        ny * (int)(rz - z0)) ==
                                         base + pos;
                T result = T(0);
                const T d0 = qq[0] - p.x;
                result += d0 * d0;
                const T d1 = qq[1] - p.y;
                result += d1 * d1;
                const T dd = qq[2] - p.z;
                result += dd * dd;
                if (own && (!RADIUS || result < nv.radius_squared)) {
                    d = result;
                    pi = RecIndex(p);
                }
            }
            list.Push(d, pi, px, py, pz);
        }
    }
}

template <typename T>
__device__ __forceinline__ void WriteTopK(const WaveTopK<T>& list, int64_t i,
                                          int* __restrict__ idx_out,
                                          T* __restrict__ d2_out,
                                          int* __restrict__ cnt_out) {
    const int lane = threadIdx.x & 63;
    if (lane < list.knn) {
        const bool ok = lane < list.nbest;
        if (idx_out) idx_out[i * list.knn + lane] = ok ? list.best_i : -1;
        if (d2_out) d2_out[i * list.knn + lane] = ok ? list.best_d : T(0);
    }
    if (cnt_out && lane == 0) cnt_out[i] = list.nbest;
}

// HybridSearch for general max_knn (core/nns/NanoFlannImpl.h:305-370 semantics:
// neighbours with d2 < r2, ascending by (d2, index), the first max_knn kept;
// idx padded with -1, dist with 0, count = min(found, max_knn)).
template <typename T>
__global__ void __launch_bounds__(kCoopBlock)
HybridSearchKernel(NnsView<T> nv, const T* __restrict__ q, int64_t nq,
                   int max_knn, int* __restrict__ idx_out,
                   T* __restrict__ d2_out, int* __restrict__ cnt_out) {
    extern __shared__ __align__(16) char coop_lds[];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    WaveTopK<T> list;
    list.Init(coop_lds);
    for (int64_t i = wave; i < nq; i += n_waves) {
        const T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        list.Reset(max_knn);
        long long cx, cy, cz;
        CellOf(qq, nv.inv_cell, cx, cy, cz);
        GatherCells<T, true>(nv, qq, cx, cy, cz, cx - 1, cx + 1, cy - 1, cy + 1,
                             cz - 1, cz + 1, 0, list);
        list.Flush();
        WriteTopK(list, i, idx_out, d2_out, cnt_out);
    }
}

// FixedRadiusSearch (core/nns/NearestNeighborSearch.cpp:114-142; GPU form
// FixedRadiusSearchCUDA, core/nns/FixedRadiusSearchOps.cu / FixedRadiusSearch
// Impl.cuh:826-1072 -- count pass, prefix sum, write pass): every neighbour
// with d2 < r2 (strict, the CPU path's nanoflann semantics), ascending by
// (d2, index), as a CSR list. A wave serves a query. The write pass emits 64
// neighbours per round: the 64 smallest pairs above the last one emitted are
// selected from the 27 cells, written, and the round repeats until a round
// comes back short -- no cap on the neighbourhood size, one round for the
// usual one.
template <typename T>
struct CountSink {
    int count;
    __device__ __forceinline__ void Push(T, int i, T, T, T) {
        if (i != kNoIndex) ++count;
    }
};

template <typename T>
__global__ void __launch_bounds__(kCoopBlock)
RadiusCountKernel(NnsView<T> nv, const T* __restrict__ q, int64_t nq,
                  int* __restrict__ counts) {
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < nq; i += n_waves) {
        const T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        long long cx, cy, cz;
        CellOf(qq, nv.inv_cell, cx, cy, cz);
        CountSink<T> sink;
        sink.count = 0;
        GatherCells<T, true>(nv, qq, cx, cy, cz, cx - 1, cx + 1, cy - 1, cy + 1,
                             cz - 1, cz + 1, 0, sink);
        int c = sink.count;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m);
        if ((threadIdx.x & 63) == 0) counts[i] = c;
    }
}

// Candidates at or below the floor pair were emitted in an earlier round.
template <typename T>
struct FloorSink {
    WaveTopK<T>* list;
    T floor_d;
    int floor_i;
    bool has_floor;
    __device__ __forceinline__ void Push(T d, int i, T, T, T) {
        if (has_floor && i != kNoIndex && !PairLess(floor_d, floor_i, d, i))
            i = kNoIndex;
        list->Push(d, i);
    }
};

template <typename T>
__global__ void __launch_bounds__(kCoopBlock)
RadiusWriteKernel(NnsView<T> nv, const T* __restrict__ q, int64_t nq,
                  const int64_t* __restrict__ row_splits,
                  int* __restrict__ idx_out, T* __restrict__ d2_out) {
    extern __shared__ __align__(16) char coop_lds[];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    WaveTopK<T> list;
    list.Init(coop_lds);
    for (int64_t i = wave; i < nq; i += n_waves) {
        const T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        long long cx, cy, cz;
        CellOf(qq, nv.inv_cell, cx, cy, cz);
        const int64_t begin = row_splits[i], end = row_splits[i + 1];
        int64_t at = begin;
        FloorSink<T> sink;
        sink.list = &list;
        sink.floor_d = T(0);
        sink.floor_i = -1;
        sink.has_floor = false;
        while (at < end) {
            list.Reset(kMaxKnn);
            GatherCells<T, true>(nv, qq, cx, cy, cz, cx - 1, cx + 1, cy - 1,
                                 cy + 1, cz - 1, cz + 1, 0, sink);
            list.Flush();
            if (lane < list.nbest && at + lane < end) {
                idx_out[at + lane] = list.best_i;
                if (d2_out) d2_out[at + lane] = list.best_d;
            }
            at += list.nbest;
            if (list.nbest < kMaxKnn) break;
            sink.floor_d = list.kth_d;
            sink.floor_i = list.kth_i;
            sink.has_floor = true;
        }
    }
}

// EstimateCovariancesUsingRadiusSearch (t/geometry/kernel/PointCloudImpl.h:
// 641-689): every neighbour with d2 < r2, no cap. One wave per point, two
// sweeps over the 27 cells: count + centroid, then the six cumulants about it
// (EstimatePointWiseRobustNormalizedCovarianceKernel, :512-585, float64 sums).
// The reference adds the neighbours one by one in ascending distance; the wave
// adds them in parallel, so the float64 sums can differ in their last bits
// (the stored covariance is their rounding to the point dtype).
template <typename T>
struct MomentSink {
    double acc[6];
    double c[3];
    int count;
    bool second;
    __device__ __forceinline__ void Push(T, int i, T x, T y, T z) {
        if (i == kNoIndex) return;
        if (!second) {
            acc[0] += (double)x;
            acc[1] += (double)y;
            acc[2] += (double)z;
            ++count;
        } else {
            const double dx = (double)x - c[0], dy = (double)y - c[1],
                         dz = (double)z - c[2];
            acc[0] += dx * dx;
            acc[1] += dy * dy;
            acc[2] += dz * dz;
            acc[3] += dx * dy;
            acc[4] += dx * dz;
            acc[5] += dy * dz;
        }
    }
    __device__ __forceinline__ void WaveSum(int n) {
        for (int k = 0; k < n; ++k)
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) acc[k] += __shfl_xor(acc[k], m);
    }
};

template <typename T>
__global__ void __launch_bounds__(kCoopBlock)
RadiusCovariancesKernel(NnsView<T> nv, const T* __restrict__ q, int64_t nq,
                        T* __restrict__ covariances) {
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (int64_t i = wave; i < nq; i += n_waves) {
        const T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        long long cx, cy, cz;
        CellOf(qq, nv.inv_cell, cx, cy, cz);
        MomentSink<T> sink;
#pragma unroll
        for (int k = 0; k < 6; ++k) sink.acc[k] = 0;
        sink.count = 0;
        sink.second = false;
        GatherCells<T, true>(nv, qq, cx, cy, cz, cx - 1, cx + 1, cy - 1, cy + 1,
                             cz - 1, cz + 1, 0, sink);
        sink.WaveSum(3);
        int cnt = sink.count;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) cnt += __shfl_xor(cnt, m);
        T* cov = covariances + 9 * i;
        if (cnt < 3) {
            if (lane < 9) cov[lane] = (lane % 4 == 0) ? T(1.0) : T(0.0);
            continue;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) sink.c[k] = sink.acc[k] / cnt;
#pragma unroll
        for (int k = 0; k < 6; ++k) sink.acc[k] = 0;
        sink.second = true;
        GatherCells<T, true>(nv, qq, cx, cy, cz, cx - 1, cx + 1, cy - 1, cy + 1,
                             cz - 1, cz + 1, 0, sink);
        sink.WaveSum(6);
        if (lane == 0) {
            const double nf = (double)(cnt - 1);
            double cm[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) cm[k] = sink.acc[k] / nf;
            cov[0] = (T)cm[0];
            cov[4] = (T)cm[1];
            cov[8] = (T)cm[2];
            cov[1] = (T)cm[3];
            cov[3] = cov[1];
            cov[2] = (T)cm[4];
            cov[6] = cov[2];
            cov[5] = (T)cm[5];
            cov[7] = cov[5];
        }
    }
}

// ---- k nearest neighbours without a radius (KnnIndex / KnnSearch) ----------
// NearestNeighborSearch::KnnSearch semantics (core/nns/NanoFlannImpl.h:
// _KnnSearchCPU, nanoflann KNNResultSet): the min(knn, N) nearest points,
// ascending by distance (ties by lower index here; nanoflann's tie order
// depends on its tree traversal). The reference's GPU path is a brute-force
// distance matrix + block select (core/nns/KnnSearchOps.cu); points in 3-D do
// better on the same bucketed grid as the radius index, searched in growing
// cubic shells: after shell r every unvisited point is farther than
// r * cell + (distance from the query to the nearest face of its own cell),
// so the walk stops as soon as the k-th distance is below that bound. The
// cell size is chosen by the host so that an occupied cell holds ~knn / 2
// points (shells 0 and 1 then usually suffice).
// A query in a sparse region would walk thousands of empty shells on a single
// fine grid, so the index is a pyramid: level l has cells 4^l times the finest;
// a level is searched for at most kKnnShells shells, then the walk restarts on
// the next coarser level (the list starts over there). The
// coarsest level spans the whole cloud in a handful of cells and is searched
// exhaustively.
template <typename T>
struct KnnGrid {
    NnsView<T> nv;
    double cell;
    long long cmin[3], cmax[3];  // occupied cell box
};

constexpr int kKnnMaxLevels = 12;
constexpr int kKnnShells = 3;

template <typename T>
struct KnnPyramid {
    int n_levels;
    int first_radius;  // cube radius of the first step on a level
    int first_level;   // levels [first_level, n_levels) are walked
    int exhaustive_last;  // the last level covers the cloud: search all of it
    int brute;            // scan all n_points records instead of walking cells
    int64_t n_points;
    KnnGrid<T> level[kKnnMaxLevels];
};

template <typename T>
__global__ void __launch_bounds__(kCoopBlock)
__attribute__((amdgpu_waves_per_eu(4)))
KnnSearchKernel(KnnPyramid<T> pyr, const T* __restrict__ q, int64_t nq, int knn,
                const int* __restrict__ query_ids, int* __restrict__ retry_ids,
                int* __restrict__ retry_count, int* __restrict__ idx_out,
                T* __restrict__ d2_out, int* __restrict__ cnt_out) {
    extern __shared__ __align__(16) char coop_lds[];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    WaveTopK<T> list;
    list.Init(coop_lds);
    for (int64_t w = wave; w < nq; w += n_waves) {
        // second pass: only the queries the finest level could not finish
        const int64_t i = query_ids ? (int64_t)query_ids[w] : w;
        const T qq[3] = {q[3 * i + 0], q[3 * i + 1], q[3 * i + 2]};
        bool done = false;
        if (pyr.brute) {
            // few leftover queries: one coalesced sweep over the records, the
            // k-th distance prunes almost every batch after the first merges
            list.Reset(knn);
            const Rec4<T>* rec = pyr.level[0].nv.sorted;
            const int lane = threadIdx.x & 63;
            constexpr int kAhead = 8;  // record loads in flight per lane
            for (int64_t t0 = 0; t0 < pyr.n_points; t0 += 64 * kAhead) {
                Rec4<T> p[kAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const int64_t t = t0 + 64 * u + lane;
                    p[u] = rec[t < pyr.n_points ? t : pyr.n_points - 1];
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const int64_t t = t0 + 64 * u + lane;
                    T result = T(0);
                    const T d0 = qq[0] - p[u].x;
                    result += d0 * d0;
                    const T d1 = qq[1] - p[u].y;
                    result += d1 * d1;
                    const T dd = qq[2] - p[u].z;
                    result += dd * dd;
                    const bool in = t < pyr.n_points;
                    list.Push(in ? result : InfOf<T>(),
                              in ? RecIndex(p[u]) : kNoIndex);
                }
            }
            list.Flush();
            done = true;
        }
        for (int l = pyr.first_level; l < pyr.n_levels && !done; ++l) {
            // a coarser level covers the finer one's cells again: start over
            list.Reset(knn);
            const KnnGrid<T>& g = pyr.level[l];
            const bool last = pyr.exhaustive_last && l == pyr.n_levels - 1;
            long long c[3];
            CellOf(qq, g.nv.inv_cell, c[0], c[1], c[2]);
            // distance to the nearest face of the query's own cell
            double margin = g.cell;
            long long r0 = 0, rmax = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)qq[a] - (double)c[a] * g.cell;
                const double hi = (double)(c[a] + 1) * g.cell - (double)qq[a];
                margin = fmin(margin, fmin(lo, hi));
                const long long below = g.cmin[a] - c[a];
                const long long above = c[a] - g.cmax[a];
                r0 = max(r0, max(below, above));        // shells before the box
                rmax = max(rmax, max(-below, -above));  // shell covering the box
            }
            // cell assignment rounds in float64: keep a small absolute slack
            margin -= 1e-7 * g.cell;
            if (!(margin > 0)) margin = 0;
            if (last) {
                // the whole occupied box in one step
                GatherCells<T, false>(g.nv, qq, c[0], c[1], c[2], g.cmin[0],
                                      g.cmax[0], g.cmin[1], g.cmax[1],
                                      g.cmin[2], g.cmax[2], 0, list);
                list.Flush();
                done = true;
                break;
            }
            if (r0 > kKnnShells) continue;  // the box is out of reach here
            bool first = true;
            for (long long r = r0 > pyr.first_radius ? r0 : pyr.first_radius;
                 r <= kKnnShells; ++r) {
                // first step: the whole cube of radius r (shells 0..r); then
                // one shell at a time
                GatherCells<T, false>(
                        g.nv, qq, c[0], c[1], c[2], max(c[0] - r, g.cmin[0]),
                        min(c[0] + r, g.cmax[0]), max(c[1] - r, g.cmin[1]),
                        min(c[1] + r, g.cmax[1]), max(c[2] - r, g.cmin[2]),
                        min(c[2] + r, g.cmax[2]), first ? 0 : r, list);
                first = false;
                list.Flush();
                if (list.nbest == knn) {
                    const double bound = (double)r * g.cell + margin;
                    if ((double)list.kth_d < bound * bound * (1.0 - 1e-6)) {
                        done = true;
                        break;
                    }
                }
                if (r >= rmax) {  // nothing of the box lies beyond: complete
                    done = true;
                    break;
                }
            }
        }
        if (!done) {
            // first pass on the finest level only: leave it to the second
            if (retry_ids && (threadIdx.x & 63) == 0)
                retry_ids[atomicAdd(retry_count, 1)] = (int)i;
            continue;
        }
        WriteTopK(list, i, idx_out, d2_out, cnt_out);
    }
}

// Bounding box of a cloud, as order-preserving 64-bit keys of the float64
// coordinates (atomicMin / atomicMax work on them).
__device__ __forceinline__ unsigned long long OrderedKey(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double FromOrderedKey(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    double v;
    std::memcpy(&v, &u, sizeof(v));
    return v;
}

template <typename T>
__global__ void BoundsKernel(const T* __restrict__ pts, int64_t n,
                             unsigned long long* __restrict__ mn,
                             unsigned long long* __restrict__ mx) {
    double lo[3] = {INFINITY, INFINITY, INFINITY};
    double hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = (double)pts[3 * i + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    }
    // wave, then workgroup (LDS), then one atomic pair per axis and workgroup:
    // atomics on six addresses serialise, a few hundred of them are noise
    __shared__ double wlo[kBlock / 64][3], whi[kBlock / 64][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int m = 32; m > 0; m >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], m));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], m));
        }
        if ((threadIdx.x & 63) == 0) {
            wlo[threadIdx.x >> 6][a] = lo[a];
            whi[threadIdx.x >> 6][a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        double l = wlo[0][a], h = whi[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            l = fmin(l, wlo[w][a]);
            h = fmax(h, whi[w][a]);
        }
        if (l <= h) {
            atomicMin(&mn[a], OrderedKey(l));
            atomicMax(&mx[a], OrderedKey(h));
        }
    }
}

__global__ void CountOccupiedKernel(const uint2* __restrict__ ranges,
                                    int64_t n_buckets,
                                    unsigned* __restrict__ occupied) {
    unsigned local = 0;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         b < n_buckets; b += (int64_t)gridDim.x * blockDim.x)
        local += ranges[b].y ? 1u : 0u;
    for (int m = 32; m > 0; m >>= 1) local += __shfl_xor(local, m);
    __shared__ unsigned wsum[kBlock / 64];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
        if (t) atomicAdd(occupied, t);
    }
}

}  // namespace
// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadNns() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &CountKernel<float>)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi

using namespace o3dmi;

namespace {

template <typename T>
int BuildIndex(o3dmi_nns* nns, const T* pts, const T* normals, hipStream_t s) {
    const int64_t n = nns->n;
    int64_t nb = 1024;
    while (nb < 2 * n && nb < (1ll << 27)) nb <<= 1;
    nns->n_buckets = nb;
    unsigned mask = (unsigned)(nb - 1);
    const size_t recs = sizeof(Rec4<T>) * (size_t)(n > 0 ? n : 1);
    { int st_; if ((st_ = PoolAlloc((void**)&nns->ranges, sizeof(uint2) * (size_t)(nb + 1)))) return st_; }
    { int st_; if ((st_ = PoolAlloc(&nns->sorted_pts, recs))) return st_; }
    if (normals)
        { int st_; if ((st_ = PoolAlloc(&nns->sorted_normals, recs))) return st_; }
    { int st_; if ((st_ = PoolAlloc((void**)&nns->partials, sizeof(double) * kCUs * 4 * kNumSums))) return st_; }
    // (a search launch writes <= 512 rows; the last row holds the tickets)
    nns->tickets = (int*)(nns->partials + (size_t)(kCUs * 4 - 1) * kNumSums);
    // zeroed by the build launch that runs anyway (none for an empty cloud)
    if (n <= 0)
        O3DMI_HIP_CHECK(hipMemsetAsync(nns->tickets, 0, sizeof(int) * 16, s));
    if (n > 0 && n <= kSmallIndexPoints) {
        hipLaunchKernelGGL(BuildSmallIndexKernel<T>, dim3(1),
                           dim3(kSmallIndexBlock), 0, s, pts, normals, (int)n,
                           (const int*)nullptr, nns->inv_cell, mask, (int)nb,
                           nns->ranges,
                           (Rec4<T>*)nns->sorted_pts,
                           (Rec4<T>*)nns->sorted_normals, nns->tickets);
        O3DMI_HIP_CHECK(hipGetLastError());
        return O3DMI_OK;
    }
    // one fill launch: a size that is not a multiple of 16 bytes is cleared
    // by two (the pool rounds the block up to a power of two, so the padding
    // is there)
    O3DMI_HIP_CHECK(hipMemsetAsync(
            nns->ranges, 0,
            (sizeof(uint2) * (size_t)(nb + 1) + 255) & ~(size_t)255, s));
    if (n > 0) {
        hipLaunchKernelGGL(CountKernel<T>, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, pts, n, nns->inv_cell, mask,
                           nns->ranges);
        hipLaunchKernelGGL(AssignRangesKernel,
                           dim3(GridFor(nb, kAssignBlock)), dim3(kAssignBlock),
                           0, s, nns->ranges, nb, nns->tickets);
        hipLaunchKernelGGL(ScatterKernel<T>, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, pts, normals, n, nns->inv_cell,
                           mask, nns->ranges, (Rec4<T>*)nns->sorted_pts,
                           (Rec4<T>*)nns->sorted_normals);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// The large-cloud build (clear, count, assign, scatter) of up to kIndexJobs
// indices in four launches. Indices of <= 4096 points take their own
// one-launch build.
template <typename T>
int BuildIndexMany(o3dmi_nns** nns, const void* const* pts,
                   const void* const* normals, int count, hipStream_t s) {
    IndexJob<T> jobs[kIndexJobs] = {};
    int n_jobs = 0;
    int64_t most_n = 0, most_nb = 0;
    for (int q = 0; q < count; ++q) {
        o3dmi_nns* x = nns[q];
        if (x->n <= kSmallIndexPoints) {
            int st = BuildIndex<T>(x, (const T*)pts[q], (const T*)normals[q], s);
            if (st) return st;
            continue;
        }
        const int64_t n = x->n;
        int64_t nb = 1024;
        while (nb < 2 * n && nb < (1ll << 27)) nb <<= 1;
        x->n_buckets = nb;
        const size_t recs = sizeof(Rec4<T>) * (size_t)n;
        int st;
        if ((st = PoolAlloc((void**)&x->ranges, sizeof(uint2) * (size_t)(nb + 1))))
            return st;
        if ((st = PoolAlloc(&x->sorted_pts, recs))) return st;
        if (normals[q] && (st = PoolAlloc(&x->sorted_normals, recs))) return st;
        if ((st = PoolAlloc((void**)&x->partials,
                            sizeof(double) * kCUs * 4 * kNumSums)))
            return st;
        x->tickets = (int*)(x->partials + (size_t)(kCUs * 4 - 1) * kNumSums);
        IndexJob<T>& j = jobs[n_jobs++];
        j.pts = (const T*)pts[q];
        j.normals = (const T*)normals[q];
        j.n = n;
        j.inv_cell = x->inv_cell;
        j.mask = (unsigned)(nb - 1);
        j.nb = nb;
        j.ranges = x->ranges;
        j.sorted = (Rec4<T>*)x->sorted_pts;
        j.sorted_normals = (Rec4<T>*)x->sorted_normals;
        j.tickets = x->tickets;
        most_n = n > most_n ? n : most_n;
        most_nb = nb > most_nb ? nb : most_nb;
    }
    if (n_jobs == 0) return O3DMI_OK;
    // (unused jobs: n = 0, nb = -1: their workgroups find nothing to do)
    for (int q = n_jobs; q < kIndexJobs; ++q) jobs[q].nb = -1;
    const unsigned gy = (unsigned)n_jobs;
    hipLaunchKernelGGL(ClearManyKernel<T>,
                       dim3(GridFor(most_nb + 1, kBlock), gy), dim3(kBlock), 0,
                       s, jobs[0], jobs[1], jobs[2], jobs[3]);
    hipLaunchKernelGGL(CountManyKernel<T>, dim3(GridFor(most_n, kBlock), gy),
                       dim3(kBlock), 0, s, jobs[0], jobs[1], jobs[2], jobs[3]);
    hipLaunchKernelGGL(AssignManyKernel<T>,
                       dim3(GridFor(most_nb, kAssignBlock), gy),
                       dim3(kAssignBlock), 0, s, jobs[0], jobs[1], jobs[2],
                       jobs[3]);
    hipLaunchKernelGGL(ScatterManyKernel<T>, dim3(GridFor(most_n, kBlock), gy),
                       dim3(kBlock), 0, s, jobs[0], jobs[1], jobs[2], jobs[3]);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

template <typename T>
int EnsureSortedNormals(o3dmi_nns* nns, const T* normals, hipStream_t s) {
    if (!nns->sorted_normals)
        { int st_; if ((st_ = PoolAlloc(&nns->sorted_normals, sizeof(Rec4<T>) * (size_t)(nns->n > 0 ? nns->n : 1)))) return st_; }
    if (nns->n > 0)
        hipLaunchKernelGGL(GatherAttrKernel<T>, dim3(GridFor(nns->n, kBlock)),
                           dim3(kBlock), 0, s, normals,
                           (const Rec4<T>*)nns->sorted_pts, nns->n,
                           (Rec4<T>*)nns->sorted_normals);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace

// Internal (not in the public header): attach target normals to an index so
// that the fused search+accumulate kernel can gather them in bucket order.
extern "C" int o3dmi_nns_set_normals(o3dmi_nns_t* nns, const void* normals_dev,
                                     o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns && normals_dev, "null argument");
    if (nns->dtype == O3DMI_F64)
        return EnsureSortedNormals<double>(nns, (const double*)normals_dev,
                                           (hipStream_t)stream);
    return EnsureSortedNormals<float>(nns, (const float*)normals_dev,
                                      (hipStream_t)stream);
}

extern "C" {

int o3dmi_internal_nns_create_with_normals(const void* points_dev,
                                           const void* normals_dev, int64_t n,
                                           int dtype, double radius,
                                           o3dmi_stream_t stream,
                                           o3dmi_nns_t** out);

int o3dmi_nns_create(const void* points_dev, int64_t n, int dtype,
                     double radius, o3dmi_stream_t stream, o3dmi_nns_t** out) {
    return o3dmi_internal_nns_create_with_normals(points_dev, nullptr, n,
                                                  dtype, radius, stream, out);
}

// Internal (host drivers): the index with the target normals scattered into
// record order by the build itself (o3dmi_nns_set_normals is a second pass
// over a finished index). Stream-ordered: returns without waiting.
int o3dmi_internal_nns_create_with_normals(const void* points_dev,
                                           const void* normals_dev, int64_t n,
                                           int dtype, double radius,
                                           o3dmi_stream_t stream,
                                           o3dmi_nns_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(radius > 0, "radius must be positive");
    // records are addressed by 32-bit byte offsets (32 bytes each for f64)
    O3DMI_REQUIRE(n >= 0 && n < (1ll << 27), "n out of range (< 2^27 points)");
    O3DMI_REQUIRE(n == 0 || points_dev != nullptr, "points is null");
    auto* nns = new o3dmi_nns();
    nns->dtype = dtype;
    nns->n = n;
    nns->radius = radius;
    nns->inv_cell = 1.0 / (radius * 1.001);
    int st = dtype == O3DMI_F64
                     ? BuildIndex<double>(nns, (const double*)points_dev,
                                          (const double*)normals_dev,
                                          (hipStream_t)stream)
                     : BuildIndex<float>(nns, (const float*)points_dev,
                                         (const float*)normals_dev,
                                         (hipStream_t)stream);
    if (st != O3DMI_OK) {
        o3dmi_nns_destroy(nns);
        return st;
    }
    *out = nns;
    return O3DMI_OK;
}

// Internal (ICP driver): `count` (<= 4) indices with their normals, built in
// the SAME launches (BuildIndexMany); all of one dtype. On an error the
// indices created so far are destroyed and out[] is left NULL.
int o3dmi_internal_nns_create_many(int count, const void* const* points_dev,
                                   const void* const* normals_dev,
                                   const int64_t* n, int dtype,
                                   const double* radius, o3dmi_stream_t stream,
                                   o3dmi_nns_t** out) {
    O3DMI_REQUIRE(out && points_dev && normals_dev && n && radius &&
                          count >= 1 && count <= kIndexJobs,
                  "bad argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    o3dmi_nns* made[kIndexJobs] = {};
    for (int q = 0; q < count; ++q) {
        out[q] = nullptr;
        if (!(radius[q] > 0) || n[q] < 0 || n[q] >= (1ll << 27) ||
            (n[q] > 0 && !points_dev[q])) {
            for (int k = 0; k < q; ++k) o3dmi_nns_destroy(made[k]);
            SetLastError("nns_create_many: bad size / radius / points");
            return O3DMI_ERR_INVALID_ARG;
        }
        auto* x = new o3dmi_nns();
        x->dtype = dtype;
        x->n = n[q];
        x->radius = radius[q];
        x->inv_cell = 1.0 / (radius[q] * 1.001);
        made[q] = x;
    }
    const int st = dtype == O3DMI_F64
                           ? BuildIndexMany<double>(made, points_dev,
                                                    normals_dev, count,
                                                    (hipStream_t)stream)
                           : BuildIndexMany<float>(made, points_dev,
                                                   normals_dev, count,
                                                   (hipStream_t)stream);
    if (st != O3DMI_OK) {
        for (int q = 0; q < count; ++q) o3dmi_nns_destroy(made[q]);
        return st;
    }
    for (int q = 0; q < count; ++q) out[q] = made[q];
    return O3DMI_OK;
}

// Internal (ICP driver): the index of a SMALL cloud (<= 4096 points: the
// coarsest level of a pyramid) whose size is still a device word -- the
// one-workgroup build is queued now, sized for the largest cloud it can take
// (8192 buckets); o3dmi_internal_nns_adopt_count tells the index its size once
// the host knows it. The build does nothing when the cloud is larger.
int o3dmi_internal_nns_destroy_completed(o3dmi_nns_t* nns);

int o3dmi_internal_nns_create_small_deferred(const void* points_dev,
                                             const void* normals_dev,
                                             const int* n_dev, int dtype,
                                             double radius,
                                             o3dmi_stream_t stream,
                                             o3dmi_nns_t** out) {
    O3DMI_REQUIRE(out && points_dev && n_dev, "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(radius > 0, "radius must be positive");
    hipStream_t s = (hipStream_t)stream;
    auto* nns = new o3dmi_nns();
    nns->dtype = dtype;
    nns->n = kSmallIndexPoints;  // until adopt_count
    nns->radius = radius;
    nns->inv_cell = 1.0 / (radius * 1.001);
    nns->n_buckets = kSmallIndexBuckets;
    const size_t rec = dtype == O3DMI_F64 ? sizeof(Rec4<double>)
                                          : sizeof(Rec4<float>);
    int st = PoolAlloc((void**)&nns->ranges,
                       sizeof(uint2) * (size_t)(kSmallIndexBuckets + 1));
    if (!st) st = PoolAlloc(&nns->sorted_pts, rec * kSmallIndexPoints);
    if (!st && normals_dev)
        st = PoolAlloc(&nns->sorted_normals, rec * kSmallIndexPoints);
    if (!st)
        st = PoolAlloc((void**)&nns->partials,
                       sizeof(double) * kCUs * 4 * kNumSums);
    if (st) {
        o3dmi_internal_nns_destroy_completed(nns);
        return st;
    }
    nns->tickets = (int*)(nns->partials + (size_t)(kCUs * 4 - 1) * kNumSums);
    const unsigned mask = (unsigned)(kSmallIndexBuckets - 1);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(BuildSmallIndexKernel<double>, dim3(1),
                           dim3(kSmallIndexBlock), 0, s,
                           (const double*)points_dev,
                           (const double*)normals_dev, 0, n_dev, nns->inv_cell,
                           mask, kSmallIndexBuckets, nns->ranges,
                           (Rec4<double>*)nns->sorted_pts,
                           (Rec4<double>*)nns->sorted_normals, nns->tickets);
    else
        hipLaunchKernelGGL(BuildSmallIndexKernel<float>, dim3(1),
                           dim3(kSmallIndexBlock), 0, s,
                           (const float*)points_dev, (const float*)normals_dev,
                           0, n_dev, nns->inv_cell, mask, kSmallIndexBuckets,
                           nns->ranges, (Rec4<float>*)nns->sorted_pts,
                           (Rec4<float>*)nns->sorted_normals, nns->tickets);
    if (hipGetLastError() != hipSuccess) {
        o3dmi_nns_destroy(nns);
        SetLastError("small index: launch failed");
        return O3DMI_ERR_HIP;
    }
    *out = nns;
    return O3DMI_OK;
}

// 1: the deferred index above was built for this size (it now knows it);
// 0: the cloud was too large (or empty) -- the caller builds it the long way.
int o3dmi_internal_nns_adopt_count(o3dmi_nns_t* nns, int64_t n) {
    if (!nns || n <= 0 || n > kSmallIndexPoints) return 0;
    nns->n = n;
    return 1;
}

// Internal (host drivers): the caller knows that every kernel using the index
// has completed (the ICP driver has read each search's sums through the host
// mailbox), so the blocks go back to the pool without a device-wide wait.
int o3dmi_internal_nns_destroy_completed(o3dmi_nns_t* nns) {
    if (!nns) return O3DMI_OK;
    PoolFree(nns->sorted_pts);
    PoolFree(nns->sorted_normals);
    PoolFree(nns->ranges);
    PoolFree(nns->partials);
    delete nns;
    return O3DMI_OK;
}

int o3dmi_nns_destroy(o3dmi_nns_t* nns) {
    if (!nns) return O3DMI_OK;
    // Searches on this index may still be in flight on any stream.
    (void)hipDeviceSynchronize();
    PoolFree(nns->sorted_pts);
    PoolFree(nns->sorted_normals);
    PoolFree(nns->ranges);
    PoolFree(nns->partials);
    delete nns;
    return O3DMI_OK;
}

int o3dmi_nns_hybrid_search_k1(const o3dmi_nns_t* nns, const void* queries_dev,
                               int64_t q, int32_t* idx_dev, void* dist2_dev,
                               int32_t* counts_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns != nullptr, "index is null");
    O3DMI_REQUIRE(q >= 0, "q < 0");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev != nullptr, "queries is null");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(q, kBlock)), block(kBlock);
    if (nns->dtype == O3DMI_F64)
        hipLaunchKernelGGL(HybridSearchK1Kernel<double>, grid, block, 0, s,
                           MakeView<double>(nns), (const double*)queries_dev, q,
                           idx_dev, (double*)dist2_dev, counts_dev);
    else
        hipLaunchKernelGGL(HybridSearchK1Kernel<float>, grid, block, 0, s,
                           MakeView<float>(nns), (const float*)queries_dev, q,
                           idx_dev, (float*)dist2_dev, counts_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_nns_hybrid_search(const o3dmi_nns_t* nns, const void* queries_dev,
                            int64_t q, int max_knn, int32_t* idx_dev,
                            void* dist2_dev, int32_t* counts_dev,
                            o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns != nullptr, "index is null");
    O3DMI_REQUIRE(q >= 0, "q < 0");
    O3DMI_REQUIRE(max_knn >= 1 && max_knn <= kMaxKnn,
                  "max_knn must be in [1, 64]");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev != nullptr, "queries is null");
    hipStream_t s = (hipStream_t)stream;
    // one wave per query
    dim3 grid(GridFor(q, kCoopBlock / 64, kCUs * 16)), block(kCoopBlock);
    if (nns->dtype == O3DMI_F64)
        hipLaunchKernelGGL(HybridSearchKernel<double>, grid, block,
                           CoopLdsBytesPerWave<double>() * (kCoopBlock / 64), s,
                           MakeView<double>(nns), (const double*)queries_dev, q,
                           max_knn, idx_dev, (double*)dist2_dev, counts_dev);
    else
        hipLaunchKernelGGL(HybridSearchKernel<float>, grid, block,
                           CoopLdsBytesPerWave<float>() * (kCoopBlock / 64), s,
                           MakeView<float>(nns), (const float*)queries_dev, q,
                           max_knn, idx_dev, (float*)dist2_dev, counts_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_nns_radius_count(const o3dmi_nns_t* nns, const void* queries_dev,
                           int64_t q, int32_t* counts_dev,
                           o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns != nullptr, "index is null");
    O3DMI_REQUIRE(q >= 0, "q < 0");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev && counts_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(q, kCoopBlock / 64, kCUs * 16)), block(kCoopBlock);
    if (nns->dtype == O3DMI_F64)
        hipLaunchKernelGGL(RadiusCountKernel<double>, grid, block, 0, s,
                           MakeView<double>(nns), (const double*)queries_dev, q,
                           counts_dev);
    else
        hipLaunchKernelGGL(RadiusCountKernel<float>, grid, block, 0, s,
                           MakeView<float>(nns), (const float*)queries_dev, q,
                           counts_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_nns_radius_search(const o3dmi_nns_t* nns, const void* queries_dev,
                            int64_t q, const int64_t* row_splits_dev,
                            int32_t* idx_dev, void* dist2_dev,
                            o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns != nullptr, "index is null");
    O3DMI_REQUIRE(q >= 0, "q < 0");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev && row_splits_dev && idx_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(q, kCoopBlock / 64, kCUs * 16)), block(kCoopBlock);
    if (nns->dtype == O3DMI_F64)
        hipLaunchKernelGGL(RadiusWriteKernel<double>, grid, block,
                           CoopLdsBytesPerWave<double>() * (kCoopBlock / 64), s,
                           MakeView<double>(nns), (const double*)queries_dev, q,
                           row_splits_dev, idx_dev, (double*)dist2_dev);
    else
        hipLaunchKernelGGL(RadiusWriteKernel<float>, grid, block,
                           CoopLdsBytesPerWave<float>() * (kCoopBlock / 64), s,
                           MakeView<float>(nns), (const float*)queries_dev, q,
                           row_splits_dev, idx_dev, (float*)dist2_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// EstimateCovariancesUsingRadiusSearchCUDA: covariances {q,3,3} of all index
// points within the index radius of every query.
int o3dmi_nns_radius_covariances(const o3dmi_nns_t* nns, const void* queries_dev,
                                 int64_t q, void* covariances_dev,
                                 o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns != nullptr, "index is null");
    O3DMI_REQUIRE(q >= 0, "q < 0");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev && covariances_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(q, kCoopBlock / 64, kCUs * 16)), block(kCoopBlock);
    if (nns->dtype == O3DMI_F64)
        hipLaunchKernelGGL(RadiusCovariancesKernel<double>, grid, block, 0, s,
                           MakeView<double>(nns), (const double*)queries_dev, q,
                           (double*)covariances_dev);
    else
        hipLaunchKernelGGL(RadiusCovariancesKernel<float>, grid, block, 0, s,
                           MakeView<float>(nns), (const float*)queries_dev, q,
                           (float*)covariances_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

namespace {

// Everything a KnnSearch call owns; released on every exit path (the index
// destructor drains the device first, so the pooled scratch is idle by then).
struct KnnResources {
    unsigned long long* box = nullptr;  // [0..2] min keys, [3..5] max, [6] count
    int* retry = nullptr;               // [0] = count, [1..q] = query ids
    std::vector<o3dmi_nns*> levels;     // finest first
    ~KnnResources() {
        for (o3dmi_nns* lv : levels) o3dmi_nns_destroy(lv);
        PoolFree(box);
        PoolFree(retry);
    }
    int AddLevel(const void* points, int64_t n, int dtype, double cell,
                 hipStream_t s) {
        auto* lv = new o3dmi_nns();
        lv->dtype = dtype;
        lv->n = n;
        lv->radius = cell;
        lv->inv_cell = 1.0 / cell;
        levels.push_back(lv);
        return dtype == O3DMI_F64
                       ? BuildIndex<double>(lv, (const double*)points, nullptr, s)
                       : BuildIndex<float>(lv, (const float*)points, nullptr, s);
    }
    void DropLevels() {
        for (o3dmi_nns* lv : levels) o3dmi_nns_destroy(lv);
        levels.clear();
    }
};

}  // namespace

// Internal form (also fills counts_dev {q} with the row width when given).
int o3dmi_nns_knn_search_counts(const void* points_dev, int64_t n,
                                const void* queries_dev, int64_t q, int dtype,
                                int knn, int32_t* idx_dev, void* dist2_dev,
                                int32_t* counts_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(knn > 0, "knn should be larger than 0.");
    O3DMI_REQUIRE(n > 0 && n < (1ll << 31) && points_dev, "empty dataset");
    O3DMI_REQUIRE(q >= 0 && q < (1ll << 31) - 1, "q out of range");
    const int k = (int)(n < (int64_t)knn ? n : (int64_t)knn);
    O3DMI_REQUIRE(k <= kMaxKnn, "knn > 64 is not supported");
    if (q == 0) return O3DMI_OK;
    O3DMI_REQUIRE(queries_dev && idx_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    KnnResources res;
    int st;

    // 1. Bounding box of the dataset.
    if ((st = PoolAlloc((void**)&res.box, 64))) return st;
    unsigned* occupied = (unsigned*)(res.box + 6);
    O3DMI_HIP_CHECK(hipMemsetAsync(res.box, 0xff, 24, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(res.box + 3, 0, 24, s));
    {
        int g = GridFor(n, kBlock);
        if (g > kCUs) g = kCUs;  // 6 atomics per wave on 6 addresses
        if (dtype == O3DMI_F64)
            hipLaunchKernelGGL(BoundsKernel<double>, dim3(g), dim3(kBlock), 0,
                               s, (const double*)points_dev, n, res.box,
                               res.box + 3);
        else
            hipLaunchKernelGGL(BoundsKernel<float>, dim3(g), dim3(kBlock), 0, s,
                               (const float*)points_dev, n, res.box,
                               res.box + 3);
    }
    unsigned long long hbox[6];
    O3DMI_HIP_CHECK(hipMemcpyAsync(hbox, res.box, sizeof(hbox),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    double lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = FromOrderedKey(hbox[a]);
        const double hi = FromOrderedKey(hbox[3 + a]);
        O3DMI_REQUIRE(lo[a] <= hi, "KnnSearch: dataset has no finite points");
        ext[a] = hi - lo[a];
    }

    // 2. Cell size of the finest level from the measured density. Target
    // points per occupied cell: small cells keep the candidate sets (and the
    // quadratic rank counting) small, shell 1 yields a first list whose k-th
    // distance prunes shell 2. Measured on MI355X, k = 30, 100 k queries:
    // 3 per cell 1.8 ms, 4.5: 0.71 ms, 6: 0.84 ms, 8: 1.26 ms, 15: 2.5 ms.
    double target = k * 0.15 < 2.0 ? 2.0 : k * 0.15;
    // First guess: a surface spanning the two largest extents, a filled
    // volume or a line, whichever gives the largest cell (shrinking is the
    // cheap direction: few occupied cells estimate the density reliably).
    double e[3] = {ext[0], ext[1], ext[2]};
    std::sort(e, e + 3);
    const double emax = e[2] > 0 ? e[2] : 1.0;
    const double h_min = emax * 1e-6;
    double h = std::sqrt(std::max(e[2] * e[1], 0.0) * target / (double)n);
    h = std::max(h, std::cbrt(std::max(e[0] * e[1] * e[2], 0.0) * target /
                              (double)n));
    h = std::max(h, e[2] * target / (double)n);
    if (!(h > h_min)) h = h_min;
    if (!(e[2] > 0)) h = 1.0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        res.DropLevels();
        if ((st = res.AddLevel(points_dev, n, dtype, h, s))) return st;
        const o3dmi_nns* lv = res.levels[0];
        O3DMI_HIP_CHECK(hipMemsetAsync(occupied, 0, sizeof(unsigned), s));
        int g = GridFor(lv->n_buckets, kBlock);
        if (g > kCUs) g = kCUs;
        hipLaunchKernelGGL(CountOccupiedKernel, dim3(g), dim3(kBlock), 0, s,
                           lv->ranges, lv->n_buckets, occupied);
        unsigned occ = 0;
        O3DMI_HIP_CHECK(hipMemcpyAsync(&occ, occupied, sizeof(occ),
                                       hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        const double ppc = (double)n / (double)(occ ? occ : 1);
        if (attempt == 5 || (ppc >= 0.5 * target && ppc <= 2.0 * target)) break;
        double h_next = h * std::pow(target / ppc, 0.4);
        if (h_next < h_min) h_next = h_min;
        if (h_next == h) break;
        h = h_next;
    }

    // 3. First pass on the finest level alone; the queries it cannot finish
    // (too few points within kKnnShells shells: sparse regions, outliers,
    // queries away from the cloud) are collected for a second pass.
    const bool single = !(emax / h > 3.0);  // one level already spans the cloud
    if (!single) {
        if ((st = PoolAlloc((void**)&res.retry, sizeof(int) * (size_t)(q + 1))))
            return st;
        O3DMI_HIP_CHECK(hipMemsetAsync(res.retry, 0, sizeof(int), s));
    }
    auto launch = [&](int first_level, bool exhaustive, bool brute,
                      const int* ids, int64_t count, int* retry_ids,
                      int* retry_count) -> int {
        const dim3 grid(GridFor(count, kCoopBlock / 64, kCUs * 16)),
                block(kCoopBlock);
#define O3DMI_KNN(T)                                                           \
    do {                                                                       \
        KnnPyramid<T> pyr;                                                     \
        pyr.n_levels = (int)res.levels.size();                                 \
        pyr.first_radius = 1;                                                  \
        pyr.first_level = first_level;                                         \
        pyr.exhaustive_last = exhaustive ? 1 : 0;                              \
        pyr.brute = brute ? 1 : 0;                                             \
        pyr.n_points = n;                                                      \
        for (int l = 0; l < pyr.n_levels; ++l) {                               \
            const o3dmi_nns* lv = res.levels[l];                               \
            KnnGrid<T>& kg = pyr.level[l];                                     \
            kg.nv = MakeView<T>(lv);                                           \
            kg.cell = lv->radius;                                              \
            for (int a = 0; a < 3; ++a) {                                      \
                kg.cmin[a] = (long long)std::floor(lo[a] * lv->inv_cell) - 1;  \
                kg.cmax[a] = (long long)std::floor((lo[a] + ext[a]) *          \
                                                   lv->inv_cell) + 1;          \
            }                                                                  \
        }                                                                      \
        hipLaunchKernelGGL(KnnSearchKernel<T>, grid, block,                    \
                           CoopLdsBytesPerWave<T>() * (kCoopBlock / 64), s,    \
                           pyr, (const T*)queries_dev, count, k, ids,          \
                           retry_ids, retry_count, idx_dev, (T*)dist2_dev,     \
                           counts_dev);                                        \
    } while (0)
        if (dtype == O3DMI_F64) O3DMI_KNN(double);
        else O3DMI_KNN(float);
#undef O3DMI_KNN
        O3DMI_HIP_CHECK(hipGetLastError());
        return O3DMI_OK;
    };
    if ((st = launch(0, single, false, nullptr, q,
                     res.retry ? res.retry + 1 : nullptr, res.retry)))
        return st;
    int n_retry = 0;
    if (!single) {
        O3DMI_HIP_CHECK(hipMemcpyAsync(&n_retry, res.retry, sizeof(int),
                                       hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    }

    // 4. Second pass: a coalesced sweep over all records when the leftovers
    // are few, else a pyramid of 4x coarser levels (built only now) whose
    // last level spans the cloud in <= 4 cells per axis and is searched
    // exhaustively.
    double sweep_limit = 2e9;  // leftover queries x points
    if (const char* e_ = std::getenv("O3DMI_KNN_SWEEP_LIMIT"))
        sweep_limit = std::atof(e_);
    const bool brute = (double)n_retry * (double)n <= sweep_limit;
    if (n_retry > 0 && brute) {
        if ((st = launch(0, false, true, res.retry + 1, n_retry, nullptr,
                         nullptr)))
            return st;
    } else if (n_retry > 0) {
        while ((int)res.levels.size() < kKnnMaxLevels &&
               emax / res.levels.back()->radius > 3.0) {
            if ((st = res.AddLevel(points_dev, n, dtype,
                                   res.levels.back()->radius * 4.0, s)))
                return st;
        }
        if ((st = launch(1, true, false, res.retry + 1, n_retry, nullptr,
                         nullptr)))
            return st;
    }
    if (std::getenv("O3DMI_VERBOSE"))
        std::fprintf(stderr,
                     "[o3dmi] knn: n=%lld k=%d cell=%g levels=%d target=%g "
                     "second-pass queries=%d (%s)\n",
                     (long long)n, k, h, (int)res.levels.size(), target,
                     n_retry, brute ? "sweep" : "pyramid");
    return O3DMI_OK;
}

int o3dmi_nns_knn_search(const void* points_dev, int64_t n,
                         const void* queries_dev, int64_t q, int dtype, int knn,
                         int32_t* idx_dev, void* dist2_dev,
                         o3dmi_stream_t stream) {
    return o3dmi_nns_knn_search_counts(points_dev, n, queries_dev, q, dtype,
                                       knn, idx_dev, dist2_dev, nullptr,
                                       stream);
}

}  // extern "C"
