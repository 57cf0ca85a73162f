// Point-to-plane ICP kernels for MI355X.
//
//   o3dmi_nns_create              <- BuildSpatialHashTableCUDA (core/nns/FixedRadiusIndex.h:227-233,
//                                    FixedRadiusSearchImpl.cuh:63-135,732-824)
//   o3dmi_nns_hybrid_search_k1    <- HybridSearchCUDA (FixedRadiusIndex.h:364-377,
//                                    FixedRadiusSearchImpl.cuh:514-632) with the CPU path's
//                                    nanoflann semantics (core/nns/NanoFlannImpl.h:305-370)
//   o3dmi_icp_p2plane_accumulate  <- ComputePosePointToPlaneCUDA (t/pipelines/kernel/
//                                    RegistrationCUDA.cu:29-117, RegistrationImpl.h:251-287)
//   o3dmi_icp_search_accumulate   fused search + fitness/rmse sums + accumulation
//   o3dmi_transform_points/normals<- TransformPointsCUDA/TransformNormalsCUDA
//                                    (t/geometry/kernel/TransformImpl.h:19-60)
//   o3dmi_decode_and_solve6x6, o3dmi_pose_to_transformation (host)
//                                 <- TransformationConverter.cpp:81-104,189-226
//
// Index design (new): a bucketed uniform grid with cell edge = radius*(1+1e-3).
// Target points are *reordered* by bucket into 16-byte (32-byte for f64)
// records {x,y,z,original index}, normals likewise, so a query's 27 cell
// visits read contiguous memory instead of chasing a CSR index through 12-byte
// AoS points. Cell coordinates are computed in float64 on both the build and
// the query side so that large-offset clouds (1000 m + 5 cm radius, cf.
// cpp/tests/core/NearestNeighborSearch.cpp:831-869) bin consistently.
//
// Reduction design (new): a fixed persistent grid; each lane keeps the 29 (+2)
// sums in float64 registers, a wave64 __shfl_down tree, one LDS stage per
// workgroup, one partial row per workgroup, and a single-workgroup second
// stage in fixed order => run-to-run deterministic (the reference's CUDA path
// uses float atomics, its CPU path a TBB tree of unspecified shape).

#include <algorithm>
#include <cmath>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "mailbox.h"
#include "reduce_sums.h"

namespace o3dmi {
namespace {

constexpr int kNumSums = 32;  // 29 + sum d2 + match count + pad

template <typename T> struct Rec4;  // {x,y,z,w}
template <> struct alignas(16) Rec4<float> { float x, y, z; int w; };
template <> struct alignas(32) Rec4<double> { double x, y, z; long long w; };

__device__ __forceinline__ unsigned HashCell(long long cx, long long cy,
                                             long long cz) {
    unsigned long long k = (unsigned long long)cx * 0x9E3779B97F4A7C15ull;
    k ^= (unsigned long long)cy * 0xC2B2AE3D27D4EB4Full + (k >> 29);
    k ^= (unsigned long long)cz * 0x165667B19E3779F9ull + (k << 17);
    return HashKey(k);
}

template <typename T>
__device__ __forceinline__ void CellOf(const T* p, double inv_cell,
                                       long long& cx, long long& cy,
                                       long long& cz) {
    cx = (long long)floor((double)p[0] * inv_cell);
    cy = (long long)floor((double)p[1] * inv_cell);
    cz = (long long)floor((double)p[2] * inv_cell);
}

// K1: bucket histogram.
template <typename T>
__global__ void CountKernel(const T* __restrict__ pts, int64_t n,
                            double inv_cell, unsigned mask,
                            unsigned* __restrict__ counts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        long long cx, cy, cz;
        CellOf(pts + 3 * i, inv_cell, cx, cy, cz);
        atomicAdd(&counts[HashCell(cx, cy, cz) & mask], 1u);
    }
}

// K2: exclusive scan in three passes (1024 elements per workgroup).
constexpr int kScanBlock = 256;
constexpr int kScanItems = 4;
__global__ void ScanLocalKernel(const unsigned* __restrict__ in,
                                unsigned* __restrict__ out,
                                unsigned* __restrict__ block_sums, int64_t n) {
    __shared__ unsigned lds[kScanBlock];
    int64_t base = (int64_t)blockIdx.x * kScanBlock * kScanItems;
    unsigned v[kScanItems];
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * kScanItems + k;
        v[k] = i < n ? in[i] : 0u;
        sum += v[k];
    }
    lds[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        unsigned t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0u;
        __syncthreads();
        lds[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned excl = lds[threadIdx.x] - sum;
    if (threadIdx.x == kScanBlock - 1) block_sums[blockIdx.x] = lds[threadIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * kScanItems + k;
        if (i < n) out[i] = excl;
        excl += v[k];
    }
}
__global__ void ScanBlockSumsKernel(unsigned* block_sums, int n_blocks) {
    // single workgroup, sequential over chunks of 256
    __shared__ unsigned lds[kScanBlock];
    __shared__ unsigned carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += kScanBlock) {
        int i = base + threadIdx.x;
        unsigned v = i < n_blocks ? block_sums[i] : 0u;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            unsigned t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0u;
            __syncthreads();
            lds[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + lds[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) carry += lds[threadIdx.x];
        __syncthreads();
    }
}
__global__ void ScanAddKernel(unsigned* __restrict__ out,
                              const unsigned* __restrict__ block_sums,
                              int64_t n) {
    int64_t base = (int64_t)blockIdx.x * kScanBlock * kScanItems;
    unsigned add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * kScanItems + k;
        if (i < n) out[i] += add;
    }
}

// K3: scatter points into bucket order as {x,y,z,idx} records.
template <typename T>
__global__ void ScatterKernel(const T* __restrict__ pts, int64_t n,
                              double inv_cell, unsigned mask,
                              const unsigned* __restrict__ starts,
                              unsigned* __restrict__ cursor,
                              Rec4<T>* __restrict__ sorted) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        long long cx, cy, cz;
        CellOf(pts + 3 * i, inv_cell, cx, cy, cz);
        unsigned b = HashCell(cx, cy, cz) & mask;
        unsigned pos = starts[b] + atomicAdd(&cursor[b], 1u);
        Rec4<T> r;
        r.x = pts[3 * i + 0];
        r.y = pts[3 * i + 1];
        r.z = pts[3 * i + 2];
        r.w = i;
        sorted[pos] = r;
    }
}

template <typename T>
__device__ __forceinline__ int RecIndex(const Rec4<T>& r) {
    return (int)r.w;
}

// Gather an {N,3} attribute (normals) into sorted record order.
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

template <typename T>
struct NnsView {
    const Rec4<T>* sorted;      // [n] bucket-ordered {x,y,z,idx}
    const unsigned* starts;     // [n_buckets + 1]
    double inv_cell;
    unsigned mask;
    T radius_squared;
};

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
                unsigned s = nv.starts[b], e = nv.starts[b + 1];
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
                s0 = nv.starts[b];
                cnt = nv.starts[b + 1] - s0;
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
constexpr int kKnnShells = 2;

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
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int m = 32; m > 0; m >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], m));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], m));
        }
        if ((threadIdx.x & 63) == 0) {
            if (lo[a] <= hi[a]) {
                atomicMin(&mn[a], OrderedKey(lo[a]));
                atomicMax(&mx[a], OrderedKey(hi[a]));
            }
        }
    }
}

__global__ void CountOccupiedKernel(const unsigned* __restrict__ starts,
                                    int64_t n_buckets,
                                    unsigned* __restrict__ occupied) {
    unsigned local = 0;
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         b < n_buckets; b += (int64_t)gridDim.x * blockDim.x)
        local += starts[b + 1] > starts[b] ? 1u : 0u;
    for (int m = 32; m > 0; m >>= 1) local += __shfl_xor(local, m);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(occupied, local);
}

// ---- robust kernels ---------------------------------------------------------
// RobustKernelImpl.h:35-126, literal: the double-typed literals promote parts
// of each expression to float64 before the result is narrowed to scalar_t.
template <typename T>
__device__ __forceinline__ T SquareT(T x) { return x * x; }

struct RobustParams {
    int method;
    double scaling, shape;
    int generalized_case;  // 0: ~2, 1: ~0 (never true, see below), 2: <-1e7, 3: else
};

inline bool IsCloseHost(double x, double y, double rtol) {
    // GeometryMacros.h:58-63; with y == 0 this is never true.
    return (x > (1.0 - rtol) * y) && (x < (1.0 + rtol) * y);
}

template <typename T>
__device__ __forceinline__ T RobustWeight(const RobustParams& rp, T residual) {
    const T scale = (T)rp.scaling;
    switch (rp.method) {
        case O3DMI_L2_LOSS:
            return (T)1.0;
        case O3DMI_L1_LOSS:
            return (T)(1.0 / (double)fabs(residual));
        case O3DMI_HUBER_LOSS: {
            T a = fabs(residual);
            return scale / (a < scale ? scale : a);
        }
        case O3DMI_CAUCHY_LOSS:
            return (T)(1.0 / (1.0 + (double)SquareT<T>(residual / scale)));
        case O3DMI_GM_LOSS:
            return scale / SquareT<T>(scale + SquareT<T>(residual));
        case O3DMI_TUKEY_LOSS: {
            T a = fabs(residual) / scale;
            T m = (T)1.0 < a ? (T)1.0 : a;
            double v = 1.0 - (double)SquareT<T>(m);
            return (T)(v * v);
        }
        case O3DMI_GENERALIZED_LOSS: {
            if (rp.generalized_case == 0) {
                return (T)(1.0 / (double)SquareT<T>(scale));
            } else if (rp.generalized_case == 1) {
                return (T)(2.0 / (double)(SquareT<T>(residual) +
                                          2 * SquareT<T>(scale)));
            } else if (rp.generalized_case == 2) {
                return (T)(exp((double)SquareT<T>(residual / scale) / (-2.0)) /
                           (double)SquareT<T>(scale));
            } else {
                return (T)(pow(((double)SquareT<T>(residual / scale) /
                                        fabs(rp.shape - 2.0) +
                                1),
                               ((rp.shape / 2.0) - 1.0)) /
                           (double)SquareT<T>(scale));
            }
        }
        default:
            return (T)1.0;
    }
}

RobustParams MakeRobust(int method, double scaling, double shape) {
    RobustParams rp;
    rp.method = method;
    rp.scaling = scaling;
    rp.shape = shape;
    if (IsCloseHost(shape, 2.0, 1e-3)) rp.generalized_case = 0;
    else if (IsCloseHost(shape, 0.0, 1e-3)) rp.generalized_case = 1;
    else if (shape < -1e7) rp.generalized_case = 2;
    else rp.generalized_case = 3;
    return rp;
}

// ---- 29(+2)-value reduction -------------------------------------------------
// Per-correspondence terms in T exactly as RegistrationCPU.cpp:62-74; the
// running sums are float64.
template <typename T>
__device__ __forceinline__ void AccumulateP2Plane(
        double (&A)[kNumSums], T sx, T sy, T sz, T tx, T ty, T tz, T nx, T ny,
        T nz, const RobustParams& rp) {
    // RegistrationImpl.h:274-284
    T r = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
    T J[6];
    J[0] = nz * sy - ny * sz;
    J[1] = nx * sz - nz * sx;
    J[2] = ny * sx - nx * sy;
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    T w = RobustWeight<T>(rp, r);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(J[j] * w * J[k]);
            ++i;
        }
        A[21 + j] += (double)(J[j] * w * r);
    }
    A[27] += (double)r;
    A[28] += 1.0;
}

// Point-to-point (Horn) sums in one pass: sum s, sum t, sum t s^T, count
// (RegistrationCPU.cpp:495-617 forms the means first and the centred products
// in a second pass; the centred covariance follows on the host from these raw
// moments in float64, where products of two Float32 values are exact).
template <typename T>
__device__ __forceinline__ void AccumulateP2Point(double (&A)[kNumSums], T sx,
                                                  T sy, T sz, T tx, T ty,
                                                  T tz) {
    const double s[3] = {(double)sx, (double)sy, (double)sz};
    const double t[3] = {(double)tx, (double)ty, (double)tz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[k] += s[k];
        A[3 + k] += t[k];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) A[6 + 3 * j + k] += t[j] * s[k];
    A[15] += 1.0;
}

// GetInformationJacobians + the 21 sums of ComputeInformationMatrixKernelCPU
// (RegistrationImpl.h:686-715, RegistrationCPU.cpp:652-701): per matched target
// point G^T G with G = [-[t]x | I], each packed-lower-triangle term formed in T
// as J_x[j] J_x[k] + J_y[j] J_y[k] + J_z[j] J_z[k], summed in float64.
template <typename T>
__device__ __forceinline__ void AccumulateInformation(double (&A)[kNumSums],
                                                      T tx, T ty, T tz) {
    const T Jx[6] = {T(0), tz, -ty, T(1), T(0), T(0)};
    const T Jy[6] = {-tz, T(0), tx, T(0), T(1), T(0)};
    const T Jz[6] = {ty, -tx, T(0), T(0), T(0), T(1)};
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(Jx[j] * Jx[k] + Jy[j] * Jy[k] + Jz[j] * Jz[k]);
            ++i;
        }
    }
}

// GetJacobianSymmetric + the 29 sums of ComputePoseSymmetricKernelCPU
// (RegistrationImpl.h:323-386, RegistrationCPU.cpp:124-180): plane normal
// n_t + sign(n_s . n_t) n_s, Jacobian and right-hand side about the
// correspondence means, robust weight from the un-centred residual; A[27] is
// the sum of squared un-centred residuals.
template <typename T>
struct Means3 { T s[3], t[3]; };

template <typename T>
__device__ __forceinline__ void AccumulateSymmetric(
        double (&A)[kNumSums], T sx, T sy, T sz, T tx, T ty, T tz, T snx, T sny,
        T snz, T tnx, T tny, T tnz, const Means3<T>& mean,
        const RobustParams& rp) {
    const T normal_dot = snx * tnx + sny * tny + snz * tnz;
    const T normal_sign = normal_dot < T(0) ? T(-1) : T(1);
    const T nx = tnx + normal_sign * snx;
    const T ny = tny + normal_sign * sny;
    const T nz = tnz + normal_sign * snz;
    const T sx_centered = sx - mean.s[0];
    const T sy_centered = sy - mean.s[1];
    const T sz_centered = sz - mean.s[2];
    const T tx_centered = tx - mean.t[0];
    const T ty_centered = ty - mean.t[1];
    const T tz_centered = tz - mean.t[2];
    const T sum_x = sx_centered + tx_centered;
    const T sum_y = sy_centered + ty_centered;
    const T sum_z = sz_centered + tz_centered;
    T J[6];
    J[0] = sum_y * nz - sum_z * ny;
    J[1] = sum_z * nx - sum_x * nz;
    J[2] = sum_x * ny - sum_y * nx;
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    const T centered_residual = (sx_centered - tx_centered) * nx +
                                (sy_centered - ty_centered) * ny +
                                (sz_centered - tz_centered) * nz;
    const T objective_residual =
            (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
    const T w = RobustWeight<T>(rp, objective_residual);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(J[j] * w * J[k]);
            ++i;
        }
        A[21 + j] += (double)(J[j] * w * centered_residual);
    }
    A[27] += (double)(objective_residual * objective_residual);
    A[28] += 1.0;
}

// GetJacobianColoredICP + the 29 sums of ComputePoseColoredICPKernelCPU
// (RegistrationImpl.h:388-466, RegistrationCPU.cpp:220-290): geometric and
// photometric rows, each with its own robust weight; A[27] = sum r_G^2 + r_I^2.
template <typename T>
__device__ __forceinline__ void AccumulateColored(
        double (&A)[kNumSums], const T* vs, const T* cs, const T* vt,
        const T* nt, const T* ct, const T* dit, T sqrt_lambda_geometric,
        T sqrt_lambda_photometric, const RobustParams& rp) {
    const T d = (vs[0] - vt[0]) * nt[0] + (vs[1] - vt[1]) * nt[1] +
                (vs[2] - vt[2]) * nt[2];
    T J_G[6], J_I[6];
    J_G[0] = sqrt_lambda_geometric * (-vs[2] * nt[1] + vs[1] * nt[2]);
    J_G[1] = sqrt_lambda_geometric * (vs[2] * nt[0] - vs[0] * nt[2]);
    J_G[2] = sqrt_lambda_geometric * (-vs[1] * nt[0] + vs[0] * nt[1]);
    J_G[3] = sqrt_lambda_geometric * nt[0];
    J_G[4] = sqrt_lambda_geometric * nt[1];
    J_G[5] = sqrt_lambda_geometric * nt[2];
    const T r_G = sqrt_lambda_geometric * d;
    const T vs_proj[3] = {vs[0] - d * nt[0], vs[1] - d * nt[1],
                          vs[2] - d * nt[2]};
    // "/ 3.0": float64 division, then narrowed
    const T intensity_source = (cs[0] + cs[1] + cs[2]) / 3.0;
    const T intensity_target = (ct[0] + ct[1] + ct[2]) / 3.0;
    const T is_proj = dit[0] * (vs_proj[0] - vt[0]) +
                      dit[1] * (vs_proj[1] - vt[1]) +
                      dit[2] * (vs_proj[2] - vt[2]) + intensity_target;
    const T s = dit[0] * nt[0] + dit[1] * nt[1] + dit[2] * nt[2];
    const T ditM[3] = {s * nt[0] - dit[0], s * nt[1] - dit[1],
                       s * nt[2] - dit[2]};
    J_I[0] = sqrt_lambda_photometric * (-vs[2] * ditM[1] + vs[1] * ditM[2]);
    J_I[1] = sqrt_lambda_photometric * (vs[2] * ditM[0] - vs[0] * ditM[2]);
    J_I[2] = sqrt_lambda_photometric * (-vs[1] * ditM[0] + vs[0] * ditM[1]);
    J_I[3] = sqrt_lambda_photometric * ditM[0];
    J_I[4] = sqrt_lambda_photometric * ditM[1];
    J_I[5] = sqrt_lambda_photometric * ditM[2];
    const T r_I = sqrt_lambda_photometric * (intensity_source - is_proj);
    const T w_G = RobustWeight<T>(rp, r_G);
    const T w_I = RobustWeight<T>(rp, r_I);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(J_G[j] * w_G * J_G[k] + J_I[j] * w_I * J_I[k]);
            ++i;
        }
        A[21 + j] += (double)(J_G[j] * w_G * r_G + J_I[j] * w_I * r_I);
    }
    A[27] += (double)(r_G * r_G + r_I * r_I);
    A[28] += 1.0;
}

constexpr int kReduceBlock = 256;

// Reduce-scatter wave reduction (reduce_sums.h), LDS across the 4 waves, one
// row per workgroup.
__device__ __forceinline__ void BlockReduceAndStore(double (&A)[kNumSums],
                                                    double* __restrict__ partials) {
    static_assert(kReduceBlock == kSumsBlock, "shared reduction geometry");
    BlockSumAndStore<kNumSums>(A, partials);
}

__global__ void FinalReduceKernel(const double* __restrict__ partials,
                                  int n_rows, double* __restrict__ out,
                                  int n_out, double* mail_data, int* mail_flag,
                                  int mail_seq) {
    // one wave per output column would waste lanes; 32 columns x 8 row-lanes.
    __shared__ double lds[8][kNumSums];
    int col = threadIdx.x % kNumSums;
    int rl = threadIdx.x / kNumSums;  // 0..7
    double v = 0;
    for (int r = rl; r < n_rows; r += 8) v += partials[(int64_t)r * kNumSums + col];
    lds[rl][col] = v;
    __syncthreads();
    if (threadIdx.x < n_out) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds[k][threadIdx.x];
        if (out) out[threadIdx.x] = s;
        if (mail_data) mail_data[threadIdx.x] = s;
    }
    // Host mailbox (mailbox.h): the driver spins on the sequence word instead
    // of a copy + stream synchronise per iteration.
    if (mail_flag) MailboxPublish(mail_flag, mail_seq);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
P2PlaneAccumulateKernel(const T* __restrict__ src, const T* __restrict__ tgt,
                        const T* __restrict__ tgt_n,
                        const int64_t* __restrict__ corr, int64_t n,
                        RobustParams rp, double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateP2Plane<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                             tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2],
                             tgt_n[3 * c + 0], tgt_n[3 * c + 1],
                             tgt_n[3 * c + 2], rp);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
SymmetricAccumulateKernel(const T* __restrict__ src, const T* __restrict__ src_n,
                          const T* __restrict__ tgt, const T* __restrict__ tgt_n,
                          const int64_t* __restrict__ corr, int64_t n,
                          Means3<T> mean, RobustParams rp,
                          double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateSymmetric<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                               tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2],
                               src_n[3 * i + 0], src_n[3 * i + 1],
                               src_n[3 * i + 2], tgt_n[3 * c + 0],
                               tgt_n[3 * c + 1], tgt_n[3 * c + 2], mean, rp);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
ColoredAccumulateKernel(const T* __restrict__ src, const T* __restrict__ src_c,
                        const T* __restrict__ tgt, const T* __restrict__ tgt_n,
                        const T* __restrict__ tgt_c, const T* __restrict__ tgt_g,
                        const int64_t* __restrict__ corr, int64_t n,
                        T sqrt_lambda_geometric, T sqrt_lambda_photometric,
                        RobustParams rp, double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        const T vs[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        const T cs[3] = {src_c[3 * i], src_c[3 * i + 1], src_c[3 * i + 2]};
        const T vt[3] = {tgt[3 * c], tgt[3 * c + 1], tgt[3 * c + 2]};
        const T nt[3] = {tgt_n[3 * c], tgt_n[3 * c + 1], tgt_n[3 * c + 2]};
        const T ct[3] = {tgt_c[3 * c], tgt_c[3 * c + 1], tgt_c[3 * c + 2]};
        const T dit[3] = {tgt_g[3 * c], tgt_g[3 * c + 1], tgt_g[3 * c + 2]};
        AccumulateColored<T>(A, vs, cs, vt, nt, ct, dit, sqrt_lambda_geometric,
                             sqrt_lambda_photometric, rp);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
P2PointAccumulateKernel(const T* __restrict__ src, const T* __restrict__ tgt,
                        const int64_t* __restrict__ corr, int64_t n,
                        double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateP2Point<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                             tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2]);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
InformationAccumulateKernel(const T* __restrict__ tgt,
                            const int64_t* __restrict__ corr, int64_t n,
                            double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateInformation<T>(A, tgt[3 * c + 0], tgt[3 * c + 1],
                                 tgt[3 * c + 2]);
    }
    BlockReduceAndStore(A, partials);
}

// Fused search + accumulate, one query per 32 lanes. The 27 neighbour cells of
// a query are independent look-ups (bucket bounds -> a handful of candidate
// records); walking them one after the other from a single lane is a chain of
// ~54 dependent memory round trips (~100 us per launch whatever the number of
// queries). Here lane c of a 32-lane group scans cell c, the group then takes
// the minimum by (d2, original index) -- the same winner the sequential scan
// picks -- and its lane 0 forms the Jacobian terms. A wave serves two queries.
// G = lanes per query (1, 2, 4, ... 32): few queries want G = 32 (latency),
// many queries want a small G (every lane busy); the winner is the same.
// EST: 0 = point-to-plane terms (needs sorted normals), 1 = point-to-point
// moments, 2 = information-matrix terms of the matched target point.
template <typename T, int G, int EST>
__global__ void __launch_bounds__(kReduceBlock)
SearchAccumulateKernel(NnsView<T> nv, const Rec4<T>* __restrict__ sorted_n,
                       const T* __restrict__ src, int64_t n, RobustParams rp,
                       int64_t* __restrict__ corr_out,
                       double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    constexpr int kPerWave = 64 / G;  // queries per wave
    const int c0 = threadIdx.x & (G - 1);  // first neighbour cell of this lane
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int sub = (threadIdx.x & 63) / G;
    for (int64_t base = wave * kPerWave; base < n; base += n_waves * kPerWave) {
        const int64_t i = base + sub;
        const bool valid = i < n;
        T q[3] = {T(0), T(0), T(0)};
        int pos = -1, idx = -1;
        T d2 = T(0);
        if (valid) {
            q[0] = src[3 * i + 0];
            q[1] = src[3 * i + 1];
            q[2] = src[3 * i + 2];
            long long cx, cy, cz;
            CellOf(q, nv.inv_cell, cx, cy, cz);
            for (int c = c0; c < 27; c += G) {
                const int dz = c / 9 - 1, dy = (c % 9) / 3 - 1, dx = c % 3 - 1;
                const unsigned b =
                        HashCell(cx + dx, cy + dy, cz + dz) & nv.mask;
                const unsigned s0 = nv.starts[b], e0 = nv.starts[b + 1];
                for (unsigned j = s0; j < e0; ++j) {
                    const Rec4<T> p = nv.sorted[j];
                    T result = T(0);
                    const T d0 = q[0] - p.x;
                    result += d0 * d0;
                    const T d1 = q[1] - p.y;
                    result += d1 * d1;
                    const T dd = q[2] - p.z;
                    result += dd * dd;
                    if (result < nv.radius_squared) {
                        const int pi = RecIndex(p);
                        if (pos < 0 || result < d2 ||
                            (result == d2 && pi < idx)) {
                            pos = (int)j;
                            idx = pi;
                            d2 = result;
                        }
                    }
                }
            }
        }
        // minimum by (d2, idx) over the G lanes of the group
#pragma unroll
        for (int m = G / 2; m > 0; m >>= 1) {
            const int opos = __shfl_xor(pos, m, G);
            const int oidx = __shfl_xor(idx, m, G);
            const T od2 = __shfl_xor(d2, m, G);
            const bool take = opos >= 0 &&
                              (pos < 0 || od2 < d2 || (od2 == d2 && oidx < idx));
            if (take) {
                pos = opos;
                idx = oidx;
                d2 = od2;
            }
        }
        if (valid && c0 == 0) {
            if (corr_out) corr_out[i] = pos >= 0 ? (int64_t)idx : (int64_t)-1;
            if (pos >= 0) {
                const Rec4<T> t = nv.sorted[pos];
                if constexpr (EST == 0) {
                    const Rec4<T> nn = sorted_n[pos];
                    AccumulateP2Plane<T>(A, q[0], q[1], q[2], t.x, t.y, t.z,
                                         nn.x, nn.y, nn.z, rp);
                } else if constexpr (EST == 1) {
                    AccumulateP2Point<T>(A, q[0], q[1], q[2], t.x, t.y, t.z);
                } else {
                    AccumulateInformation<T>(A, t.x, t.y, t.z);
                }
                A[29] += (double)d2;
                A[30] += 1.0;
            }
        }
    }
    BlockReduceAndStore(A, partials);
}

// TransformImpl.h:19-44
template <typename T>
struct Mat4 { T m[16]; };

template <typename T>
__global__ void TransformPointsKernel(Mat4<T> t, T* __restrict__ pts,
                                      int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T* p = pts + 3 * i;
        T p0 = p[0], p1 = p[1], p2 = p[2];
        T x0 = t.m[0] * p0 + t.m[1] * p1 + t.m[2] * p2 + t.m[3];
        T x1 = t.m[4] * p0 + t.m[5] * p1 + t.m[6] * p2 + t.m[7];
        T x2 = t.m[8] * p0 + t.m[9] * p1 + t.m[10] * p2 + t.m[11];
        T x3 = t.m[12] * p0 + t.m[13] * p1 + t.m[14] * p2 + t.m[15];
        p[0] = x0 / x3;
        p[1] = x1 / x3;
        p[2] = x2 / x3;
    }
}
// TransformImpl.h:46-60
template <typename T>
__global__ void TransformNormalsKernel(Mat4<T> t, T* __restrict__ nrm,
                                       int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T* p = nrm + 3 * i;
        T p0 = p[0], p1 = p[1], p2 = p[2];
        T x0 = t.m[0] * p0 + t.m[1] * p1 + t.m[2] * p2;
        T x1 = t.m[4] * p0 + t.m[5] * p1 + t.m[6] * p2;
        T x2 = t.m[8] * p0 + t.m[9] * p1 + t.m[10] * p2;
        p[0] = x0;
        p[1] = x1;
        p[2] = x2;
    }
}

int ReduceGrid(int64_t n) {
    int64_t g = (n + kReduceBlock - 1) / kReduceBlock;
    // Two workgroups per CU at most: enough waves to hide the gather latency,
    // and the final pass (a single workgroup, linear in the row count) reads
    // <= 512 partial rows.
    int64_t cap = (int64_t)kCUs * 2;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace
}  // namespace o3dmi

using namespace o3dmi;

struct o3dmi_nns {
    int dtype = O3DMI_F32;
    int64_t n = 0;
    double radius = 0, inv_cell = 0;
    int64_t n_buckets = 0;
    void* sorted_pts = nullptr;      // Rec4<T>[n]
    void* sorted_normals = nullptr;  // Rec4<T>[n], optional
    unsigned* starts = nullptr;      // [n_buckets + 1]
    double* partials = nullptr;      // [kCUs*4, kNumSums]
};

namespace {

template <typename T>
int BuildIndex(o3dmi_nns* nns, const T* pts, hipStream_t s) {
    const int64_t n = nns->n;
    int64_t nb = 1024;
    while (nb < 2 * n && nb < (1ll << 27)) nb <<= 1;
    nns->n_buckets = nb;
    unsigned mask = (unsigned)(nb - 1);
    unsigned *counts = nullptr, *cursor = nullptr, *block_sums = nullptr;
    int64_t n_scan = nb + 1;
    int n_scan_blocks =
            (int)((n_scan + kScanBlock * kScanItems - 1) / (kScanBlock * kScanItems));
    { int st_; if ((st_ = PoolAlloc((void**)&counts, sizeof(unsigned) * n_scan))) return st_; }
    { int st_; if ((st_ = PoolAlloc((void**)&cursor, sizeof(unsigned) * nb))) return st_; }
    { int st_; if ((st_ = PoolAlloc((void**)&block_sums, sizeof(unsigned) * (n_scan_blocks + 1)))) return st_; }
    { int st_; if ((st_ = PoolAlloc((void**)&nns->starts, sizeof(unsigned) * n_scan))) return st_; }
    { int st_; if ((st_ = PoolAlloc(&nns->sorted_pts, sizeof(Rec4<T>) * (size_t)(n > 0 ? n : 1)))) return st_; }
    { int st_; if ((st_ = PoolAlloc((void**)&nns->partials, sizeof(double) * kCUs * 4 * kNumSums))) return st_; }
    O3DMI_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(unsigned) * n_scan, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(cursor, 0, sizeof(unsigned) * nb, s));
    if (n > 0) {
        hipLaunchKernelGGL(CountKernel<T>, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, pts, n, nns->inv_cell, mask,
                           counts);
    }
    hipLaunchKernelGGL(ScanLocalKernel, dim3(n_scan_blocks), dim3(kScanBlock),
                       0, s, counts, nns->starts, block_sums, n_scan);
    hipLaunchKernelGGL(ScanBlockSumsKernel, dim3(1), dim3(kScanBlock), 0, s,
                       block_sums, n_scan_blocks);
    hipLaunchKernelGGL(ScanAddKernel, dim3(n_scan_blocks), dim3(kScanBlock), 0,
                       s, nns->starts, block_sums, n_scan);
    if (n > 0) {
        hipLaunchKernelGGL(ScatterKernel<T>, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, pts, n, nns->inv_cell, mask,
                           nns->starts, cursor, (Rec4<T>*)nns->sorted_pts);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    PoolFree(counts);
    PoolFree(cursor);
    PoolFree(block_sums);
    return O3DMI_OK;
}

template <typename T>
NnsView<T> MakeView(const o3dmi_nns* nns) {
    NnsView<T> v;
    v.sorted = (const Rec4<T>*)nns->sorted_pts;
    v.starts = nns->starts;
    v.inv_cell = nns->inv_cell;
    v.mask = (unsigned)(nns->n_buckets - 1);
    const T r = (T)nns->radius;  // NanoFlannImpl.h:332: T radius_squared
    v.radius_squared = r * r;
    return v;
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

int o3dmi_nns_create(const void* points_dev, int64_t n, int dtype,
                     double radius, o3dmi_stream_t stream, o3dmi_nns_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(radius > 0, "radius must be positive");
    O3DMI_REQUIRE(n >= 0 && n < (1ll << 31), "n out of range");
    O3DMI_REQUIRE(n == 0 || points_dev != nullptr, "points is null");
    auto* nns = new o3dmi_nns();
    nns->dtype = dtype;
    nns->n = n;
    nns->radius = radius;
    nns->inv_cell = 1.0 / (radius * 1.001);
    int st = dtype == O3DMI_F64
                     ? BuildIndex<double>(nns, (const double*)points_dev,
                                          (hipStream_t)stream)
                     : BuildIndex<float>(nns, (const float*)points_dev,
                                         (hipStream_t)stream);
    if (st != O3DMI_OK) {
        o3dmi_nns_destroy(nns);
        return st;
    }
    *out = nns;
    return O3DMI_OK;
}

int o3dmi_nns_destroy(o3dmi_nns_t* nns) {
    if (!nns) return O3DMI_OK;
    // Searches on this index may still be in flight on any stream.
    (void)hipDeviceSynchronize();
    PoolFree(nns->sorted_pts);
    PoolFree(nns->sorted_normals);
    PoolFree(nns->starts);
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
                       ? BuildIndex<double>(lv, (const double*)points, s)
                       : BuildIndex<float>(lv, (const float*)points, s);
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
    if (const char* e_ = std::getenv("O3DMI_KNN_PPC")) {  // tuning knob
        const double v = std::atof(e_);
        if (v > 0) target = v;
    }
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
        if (g > kCUs * 4) g = kCUs * 4;
        hipLaunchKernelGGL(CountOccupiedKernel, dim3(g), dim3(kBlock), 0, s,
                           lv->starts, lv->n_buckets, occupied);
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

int o3dmi_icp_p2plane_accumulate(const void* src_dev, const void* tgt_dev,
                                 const void* tgt_normals_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 int robust_kernel, double scaling_parameter,
                                 double shape_parameter, double* sums29_dev,
                                 o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && tgt_dev && tgt_normals_dev && corr_dev &&
                          sums29_dev,
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(P2PlaneAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev, corr_dev, n, rp,
                           partials);
    else
        hipLaunchKernelGGL(P2PlaneAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)tgt_dev, (const float*)tgt_normals_dev,
                           corr_dev, n, rp, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_p2point_accumulate(const void* src_dev, const void* tgt_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 double* sums16_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && tgt_dev && corr_dev && sums16_dev,
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(P2PointAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)tgt_dev, corr_dev, n, partials);
    else
        hipLaunchKernelGGL(P2PointAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)tgt_dev, corr_dev, n, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums16_dev, 16, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

// Internal: also posts the 29 sums to a host mailbox when mail_data != NULL.
int o3dmi_icp_colored_accumulate_post(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        double* partials_dev, double* mail_data, int* mail_flag, int mail_seq,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && src_colors_dev && tgt_dev && tgt_normals_dev &&
                          tgt_colors_dev && tgt_color_gradients_dev &&
                          corr_dev && (sums29_dev || mail_data),
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    O3DMI_REQUIRE(lambda_geometric >= 0 && lambda_geometric <= 1.0,
                  "lambda_geometric must be in [0, 1]");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = partials_dev;
    if (!partials)
        O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                       sizeof(double) * (size_t)g * kNumSums,
                                       s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    // ComputePoseColoredICPCPU, RegistrationCPU.cpp:310-313
    const double slg = std::sqrt(lambda_geometric);
    const double slp = std::sqrt(1.0 - lambda_geometric);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(ColoredAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)src_colors_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev,
                           (const double*)tgt_colors_dev,
                           (const double*)tgt_color_gradients_dev, corr_dev, n,
                           slg, slp, rp, partials);
    else
        hipLaunchKernelGGL(ColoredAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)src_colors_dev, (const float*)tgt_dev,
                           (const float*)tgt_normals_dev,
                           (const float*)tgt_colors_dev,
                           (const float*)tgt_color_gradients_dev, corr_dev, n,
                           (float)slg, (float)slp, rp, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, mail_data, mail_flag, mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    if (!partials_dev) O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_colored_accumulate(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums29_dev != nullptr, "null argument");
    return o3dmi_icp_colored_accumulate_post(
            src_dev, src_colors_dev, tgt_dev, tgt_normals_dev, tgt_colors_dev,
            tgt_color_gradients_dev, corr_dev, n, dtype, lambda_geometric,
            robust_kernel, scaling_parameter, shape_parameter, sums29_dev,
            nullptr, nullptr, nullptr, 0, stream);
}

// Internal: also posts the 29 sums to a host mailbox when mail_data != NULL.
int o3dmi_icp_symmetric_accumulate_post(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, double* partials_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && src_normals_dev && tgt_dev && tgt_normals_dev &&
                          corr_dev && source_mean3 && target_mean3 &&
                          (sums29_dev || mail_data),
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = partials_dev;
    if (!partials)
        O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                       sizeof(double) * (size_t)g * kNumSums,
                                       s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    if (dtype == O3DMI_F64) {
        Means3<double> m;
        for (int k = 0; k < 3; ++k) {
            m.s[k] = source_mean3[k];
            m.t[k] = target_mean3[k];
        }
        hipLaunchKernelGGL(SymmetricAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)src_normals_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev, corr_dev, n, m, rp,
                           partials);
    } else {
        Means3<float> m;
        for (int k = 0; k < 3; ++k) {
            m.s[k] = (float)source_mean3[k];
            m.t[k] = (float)target_mean3[k];
        }
        hipLaunchKernelGGL(SymmetricAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)src_normals_dev, (const float*)tgt_dev,
                           (const float*)tgt_normals_dev, corr_dev, n, m, rp,
                           partials);
    }
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, mail_data, mail_flag, mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    if (!partials_dev) O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_symmetric_accumulate(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums29_dev != nullptr, "null argument");
    return o3dmi_icp_symmetric_accumulate_post(
            src_dev, src_normals_dev, tgt_dev, tgt_normals_dev, corr_dev, n,
            dtype, source_mean3, target_mean3, robust_kernel, scaling_parameter,
            shape_parameter, sums29_dev, nullptr, nullptr, nullptr, 0, stream);
}

int o3dmi_icp_information_accumulate(const void* tgt_dev,
                                     const int64_t* corr_dev, int64_t n,
                                     int dtype, double* sums21_dev,
                                     o3dmi_stream_t stream) {
    O3DMI_REQUIRE(tgt_dev && corr_dev && sums21_dev, "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(InformationAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)tgt_dev,
                           corr_dev, n, partials);
    else
        hipLaunchKernelGGL(InformationAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)tgt_dev,
                           corr_dev, n, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums21_dev, 21, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_search_accumulate_post(
        const o3dmi_nns_t* nns, const void* src_dev,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);

int o3dmi_icp_search_accumulate_p2point(const o3dmi_nns_t* nns,
                                        const void* src_dev, int64_t n,
                                        int64_t* corr_out_dev,
                                        double* sums32_dev,
                                        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums32_dev != nullptr, "null argument");
    return o3dmi_icp_search_accumulate_post(nns, src_dev, nullptr, n, 1, 0, 1.0,
                                            1.0, corr_out_dev, sums32_dev,
                                            nullptr, nullptr, 0, stream);
}

int o3dmi_icp_search_accumulate(const o3dmi_nns_t* nns, const void* src_dev,
                                const void* tgt_normals_dev, int64_t n,
                                int robust_kernel, double scaling_parameter,
                                double shape_parameter, int64_t* corr_out_dev,
                                double* sums32_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums32_dev != nullptr, "null argument");
    return o3dmi_icp_search_accumulate_post(
            nns, src_dev, tgt_normals_dev, n, 0, robust_kernel,
            scaling_parameter, shape_parameter, corr_out_dev, sums32_dev,
            nullptr, nullptr, 0, stream);
}

// Internal (not in the public header): also posts the 32 sums to a host
// mailbox (mailbox.h) when mail_data != NULL; sums32_dev may then be NULL.
int o3dmi_icp_search_accumulate_post(
        const o3dmi_nns_t* nns, const void* src_dev,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns && src_dev && (sums32_dev || mail_data), "null argument");
    O3DMI_REQUIRE(estimation >= 0 && estimation <= 2,
                  "estimation must be point-to-plane (0), point-to-point (1) "
                  "or information (2)");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    if (estimation == 0) {
        if (tgt_normals_dev) {
            int st = o3dmi_nns_set_normals(const_cast<o3dmi_nns_t*>(nns),
                                           tgt_normals_dev, stream);
            if (st != O3DMI_OK) return st;
        }
        O3DMI_REQUIRE(nns->sorted_normals != nullptr,
                      "Target pointcloud missing normals attribute.");
    }
    // Lanes per query: as many as keep the whole chip busy about once.
    int group = 32;
    while (group > 1 && n * group > (int64_t)kCUs * 2048) group >>= 1;
    if (const char* e = std::getenv("O3DMI_NNS_GROUP")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) group = v;
    }
    int g = ReduceGrid(n * group);
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
#define O3DMI_SEARCH_E(T, G, E)                                                \
    hipLaunchKernelGGL((SearchAccumulateKernel<T, G, E>), dim3(g),            \
                       dim3(kReduceBlock), 0, s, MakeView<T>(nns),            \
                       (const Rec4<T>*)nns->sorted_normals,                   \
                       (const T*)src_dev, n, rp, corr_out_dev, nns->partials)
#define O3DMI_SEARCH(T, G)                                                     \
    do {                                                                      \
        if (estimation == 0) O3DMI_SEARCH_E(T, G, 0);                         \
        else if (estimation == 1) O3DMI_SEARCH_E(T, G, 1);                    \
        else O3DMI_SEARCH_E(T, G, 2);                                         \
    } while (0)
#define O3DMI_SEARCH_G(T)                                                      \
    switch (group) {                                                          \
        case 32: O3DMI_SEARCH(T, 32); break;                                  \
        case 16: O3DMI_SEARCH(T, 16); break;                                  \
        case 8: O3DMI_SEARCH(T, 8); break;                                    \
        case 4: O3DMI_SEARCH(T, 4); break;                                    \
        case 2: O3DMI_SEARCH(T, 2); break;                                    \
        default: O3DMI_SEARCH(T, 1); break;                                   \
    }
    if (nns->dtype == O3DMI_F64) { O3DMI_SEARCH_G(double) }
    else { O3DMI_SEARCH_G(float) }
#undef O3DMI_SEARCH_G
#undef O3DMI_SEARCH
#undef O3DMI_SEARCH_E
    hipLaunchKernelGGL(FinalSumKernel<kNumSums>, dim3(1), dim3(kFinalThreads),
                       0, s, nns->partials, g, sums32_dev, mail_data, mail_flag,
                       mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_transform_points(const double* transformation, void* points_dev,
                           int64_t n, int dtype, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(transformation && (points_dev || n == 0), "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64) {
        Mat4<double> t;
        for (int i = 0; i < 16; ++i) t.m[i] = transformation[i];
        hipLaunchKernelGGL(TransformPointsKernel<double>, grid, block, 0, s, t,
                           (double*)points_dev, n);
    } else {
        Mat4<float> t;
        for (int i = 0; i < 16; ++i) t.m[i] = (float)transformation[i];
        hipLaunchKernelGGL(TransformPointsKernel<float>, grid, block, 0, s, t,
                           (float*)points_dev, n);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_transform_normals(const double* transformation, void* normals_dev,
                            int64_t n, int dtype, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(transformation && (normals_dev || n == 0), "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "normals must be Float32 or Float64");
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64) {
        Mat4<double> t;
        for (int i = 0; i < 16; ++i) t.m[i] = transformation[i];
        hipLaunchKernelGGL(TransformNormalsKernel<double>, grid, block, 0, s,
                           t, (double*)normals_dev, n);
    } else {
        Mat4<float> t;
        for (int i = 0; i < 16; ++i) t.m[i] = (float)transformation[i];
        hipLaunchKernelGGL(TransformNormalsKernel<float>, grid, block, 0, s, t,
                           (float*)normals_dev, n);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// TransformationConverter.cpp:189-226; the solve is LAPACK ?gesv restated
// (LU with partial pivoting), core/linalg/SolveCPU.cpp:15-30.
int o3dmi_decode_and_solve6x6(const double* A, double* pose6, float* residual,
                              int* inlier_count) {
    O3DMI_REQUIRE(A && pose6 && residual && inlier_count, "null argument");
    double M[36], b[6];
    for (int j = 0; j < 6; j++) {
        b[j] = -A[21 + j];
        const int reduction_idx = (j * (j + 1)) / 2;
        for (int k = 0; k <= j; k++) {
            M[j * 6 + k] = A[reduction_idx + k];
            M[k * 6 + j] = A[reduction_idx + k];
        }
    }
    const int n = 6;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double mx = std::fabs(M[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(M[i * n + k]);
            if (v > mx) { mx = v; p = i; }
        }
        if (mx == 0.0 || !(mx == mx)) {
            for (int j = 0; j < 6; ++j) pose6[j] = 0;
            *residual = 0;
            *inlier_count = 0;
            SetLastError("Singular 6x6 linear system detected, tracking failed.");
            return O3DMI_ERR_SINGULAR;
        }
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(M[k * n + j], M[p * n + j]);
            std::swap(b[k], b[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            double l = M[i * n + k] / M[k * n + k];
            M[i * n + k] = l;
            for (int j = k + 1; j < n; ++j) M[i * n + j] -= l * M[k * n + j];
        }
    }
    for (int i = 1; i < n; ++i)
        for (int j = 0; j < i; ++j) b[i] -= M[i * n + j] * b[j];
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j) b[i] -= M[i * n + j] * b[j];
        b[i] /= M[i * n + i];
    }
    for (int j = 0; j < 6; ++j) pose6[j] = b[j];
    *residual = (float)A[27];
    *inlier_count = (int)A[28];
    return O3DMI_OK;
}

// ComputeRtPointToPointCPU after its reduction (RegistrationCPU.cpp:640-650):
// Sxy = U D V^T, R = U diag(1, 1, det(U) det(V)) V^T, t = mean_t - R mean_s.
// The reference calls LAPACK gesvd; here a one-sided (Hestenes) Jacobi SVD in
// float64: columns of G = Sxy V are rotated pairwise until orthogonal, then
// u_i = g_i / |g_i|. The third left vector is taken as u_1 x u_2, which folds
// the reflection test into det(V): with u_3 = e (u_1 x u_2), e = det(U),
// det(U) det(V) u_3 = det(V) (u_1 x u_2). That also covers planar
// correspondence sets (sigma_3 = 0) without dividing by sigma_3.
int o3dmi_compute_rt_p2point(const double* sums, double* R9, double* t3) {
    O3DMI_REQUIRE(sums && R9 && t3, "null argument");
    const double cnt = sums[15];
    if (!(cnt > 0)) {
        SetLastError("No valid correspondence present.");
        return O3DMI_ERR_NO_INLIERS;
    }
    double ms[3], mt[3];
    for (int k = 0; k < 3; ++k) {
        ms[k] = sums[k] / cnt;
        mt[k] = sums[3 + k] / cnt;
    }
    double G[3][3], V[3][3];  // G[row][col]
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
            G[j][k] = sums[6 + 3 * j + k] / cnt - mt[j] * ms[k];
            V[j][k] = j == k ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += G[r][p] * G[r][p];
                    beta += G[r][q] * G[r][q];
                    gamma += G[r][p] * G[r][q];
                }
                if (gamma == 0.0 ||
                    std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta))
                    continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tn = (zeta >= 0 ? 1.0 : -1.0) /
                                  (std::fabs(zeta) +
                                   std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + tn * tn);
                const double sn = c * tn;
                for (int r = 0; r < 3; ++r) {
                    const double gp = G[r][p], gq = G[r][q];
                    G[r][p] = c * gp - sn * gq;
                    G[r][q] = sn * gp + c * gq;
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = c * vp - sn * vq;
                    V[r][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    // column order by decreasing singular value
    double sig[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c)
        sig[c] = std::sqrt(G[0][c] * G[0][c] + G[1][c] * G[1][c] +
                           G[2][c] * G[2][c]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sig[ord[b]] > sig[ord[a]]) std::swap(ord[a], ord[b]);
    double u[3][3], v[3][3];  // u[i] / v[i] = i-th singular vectors
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 3; ++r) v[i][r] = V[r][ord[i]];
    const double s0 = sig[ord[0]], s1 = sig[ord[1]];
    if (!(s0 > 0)) {
        // all correspondences coincide with their means: rotation undetermined,
        // the least-squares answer is the pure translation.
        for (int i = 0; i < 9; ++i) R9[i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; ++k) t3[k] = mt[k] - ms[k];
        return O3DMI_OK;
    }
    for (int r = 0; r < 3; ++r) u[0][r] = G[r][ord[0]] / s0;
    if (s1 > 1e-300 && s1 > 1e-15 * s0) {
        double d = 0, nrm = 0;
        for (int r = 0; r < 3; ++r) u[1][r] = G[r][ord[1]] / s1;
        for (int r = 0; r < 3; ++r) d += u[1][r] * u[0][r];
        for (int r = 0; r < 3; ++r) {
            u[1][r] -= d * u[0][r];
            nrm += u[1][r] * u[1][r];
        }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    } else {
        // rank one: any unit vector orthogonal to u_0
        int m = 0;
        for (int r = 1; r < 3; ++r)
            if (std::fabs(u[0][r]) < std::fabs(u[0][m])) m = r;
        double e[3] = {0, 0, 0};
        e[m] = 1.0;
        double d = u[0][m], nrm = 0;
        for (int r = 0; r < 3; ++r) {
            u[1][r] = e[r] - d * u[0][r];
            nrm += u[1][r] * u[1][r];
        }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    const double detV =
            v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) -
            v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
            v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    const double sgn = detV < 0 ? -1.0 : 1.0;
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k)
            R9[j * 3 + k] = u[0][j] * v[0][k] + u[1][j] * v[1][k] +
                            sgn * u[2][j] * v[2][k];
    for (int j = 0; j < 3; ++j)
        t3[j] = mt[j] - (R9[j * 3 + 0] * ms[0] + R9[j * 3 + 1] * ms[1] +
                         R9[j * 3 + 2] * ms[2]);
    return O3DMI_OK;
}

// PoseToSymmetricTransformation (TransformationConverter.cpp:106-133) =
// TransformSymmetricPoseToMatrix4d (pipelines/registration/SymmetricICPImpl.h:
// 19-44): theta = atan |g|, H = AngleAxis(theta, g / |g|) (Rodrigues), R = H H,
// t = target_mean + H (t' cos theta) - R source_mean.
void o3dmi_symmetric_pose_to_transformation(const double* pose,
                                            const double* source_mean,
                                            const double* target_mean,
                                            double* T) {
    const double g_norm = std::sqrt(pose[0] * pose[0] + pose[1] * pose[1] +
                                    pose[2] * pose[2]);
    const double theta = std::atan(g_norm);
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (g_norm > 0.0) {
        const double ax = pose[0] / g_norm, ay = pose[1] / g_norm,
                     az = pose[2] / g_norm;
        const double c = std::cos(theta), sn = std::sin(theta), v = 1.0 - c;
        H[0] = c + v * ax * ax;
        H[1] = v * ax * ay - sn * az;
        H[2] = v * ax * az + sn * ay;
        H[3] = v * ax * ay + sn * az;
        H[4] = c + v * ay * ay;
        H[5] = v * ay * az - sn * ax;
        H[6] = v * ax * az - sn * ay;
        H[7] = v * ay * az + sn * ax;
        H[8] = c + v * az * az;
    }
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = H[i * 3 + 0] * H[0 * 3 + j] +
                           H[i * 3 + 1] * H[1 * 3 + j] +
                           H[i * 3 + 2] * H[2 * 3 + j];
    const double ct = std::cos(theta);
    const double u[3] = {pose[3] * ct, pose[4] * ct, pose[5] * ct};
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = target_mean[i] +
                       (H[i * 3 + 0] * u[0] + H[i * 3 + 1] * u[1] +
                        H[i * 3 + 2] * u[2]) -
                       (R[i * 3 + 0] * source_mean[0] +
                        R[i * 3 + 1] * source_mean[1] +
                        R[i * 3 + 2] * source_mean[2]);
    }
}

// TransformationConverterImpl.h:23-42 + TransformationConverter.cpp:81-104
void o3dmi_pose_to_transformation(const double* pose_ptr, double* T) {
    for (int i = 0; i < 16; ++i) T[i] = 0;
    T[0] = cos(pose_ptr[2]) * cos(pose_ptr[1]);
    T[1] = -1 * sin(pose_ptr[2]) * cos(pose_ptr[0]) +
           cos(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    T[2] = sin(pose_ptr[2]) * sin(pose_ptr[0]) +
           cos(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    T[4] = sin(pose_ptr[2]) * cos(pose_ptr[1]);
    T[5] = cos(pose_ptr[2]) * cos(pose_ptr[0]) +
           sin(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    T[6] = -1 * cos(pose_ptr[2]) * sin(pose_ptr[0]) +
           sin(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    T[8] = -1 * sin(pose_ptr[1]);
    T[9] = cos(pose_ptr[1]) * sin(pose_ptr[0]);
    T[10] = cos(pose_ptr[1]) * cos(pose_ptr[0]);
    T[3] = pose_ptr[3];
    T[7] = pose_ptr[4];
    T[11] = pose_ptr[5];
    T[15] = 1;
}

}  // extern "C"
