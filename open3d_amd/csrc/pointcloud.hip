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

#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "o3d_mi355x_host.h"
#include "mailbox.h"
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
    // The first kPre rows of this thread's column of the (tile, digit) table,
    // requested with the elements and the live count instead of a round trip
    // later: rows of tiles past the live count hold stale numbers and are
    // masked when the column is summed.
    constexpr int kPre = 32;
    const int bins = 1 << bits;
    const int n_tiles_host = (n_host + kSortTile - 1) / kSortTile;
    int pre[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u)
        pre[u] = (int)threadIdx.x < bins && u < n_tiles_host
                         ? hist[(int64_t)u * bins + threadIdx.x]
                         : 0;
    const int n = LiveCount(n_dev, n_host);
    if (tile >= n) return;
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
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int c = u < n_tiles ? pre[u] : 0;
            all += c;
            before += u < (int)blockIdx.x ? c : 0;
        }
        if ((int)threadIdx.x < bins) {
            const int* col = hist + threadIdx.x;
            for (int t0 = kPre; t0 < n_tiles; t0 += 16) {
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
                                const int* __restrict__ slot_of_point,
                                VdsTable tb, const int* n_dev, int n_host,
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
        // the voxel's hash slot back to the empty state: the table stays
        // clean between levels (persistent workspace, no clearing launch)
        const int slot = slot_of_point[k];
        if (slot >= 0) {
            tb.keys[slot & tb.mask] = kEmptyKey;
            tb.first[slot & tb.mask] = 0x7FFFFFFF;
        }
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

// ==== the bucketed form: three launches per level, clouds up to 2^17 points ====
// (round 3; the seven-launch chain above stays for larger clouds.) The sort
// above exists to put every voxel's points side by side in point order. Here
// the points are only PARTITIONED -- stably, by the high bits of their voxel's
// hash slot, into 512 buckets of a few hundred points -- and a workgroup per
// bucket does the rest in LDS:
//   A  VdsInsertBucketKernel   hash insert (slot, first point by atomicMin) +
//                              per-tile histogram of the bucket digit;
//   B  VdsBucketScatterKernel  gathers first[slot] -> "is the first point of
//                              its voxel" as one bit per point (a wave ballot
//                              is 64 consecutive bits: coalesced), and the
//                              stable scatter of (slot, point) by bucket --
//                              SortScatterKernel's ranking, one pass;
//   C  VdsBucketReduceKernel   per bucket: the entries and their points'
//                              coordinates / attribute staged into LDS with
//                              every load in flight at once; the lane of a
//                              voxel's first point walks the bucket's entries
//                              behind it (point order survives the stable
//                              scatter) adding its voxel's members in float32
//                              out of LDS; output row = number of first-point
//                              bits below its own (a prefix popcount over the
//                              bit array, rebuilt in LDS by every workgroup:
//                              <= 4096 words); finally every entry returns
//                              its table slot to the empty state.
// The table is therefore CLEAN after every level and lives in a persistent
// per-chain workspace (no clearing launch, no pooled scratch per call; one
// set of buffers per chain however many levels and attributes go through it).
// 52 us of kernels per level in the VGA tracking loop -> see DESIGN.md.
constexpr int kBucketBits = 9;
constexpr int kBuckets = 1 << kBucketBits;
// (Round 5 tried 2^18, so that the finest level of a 1280x720 frame -- 230 400
// points -- would take the three launches instead of the sort's seven: the
// tracking loop got SLOWER, 1177-1207 -> 1127-1133 frames/s; buckets of ~450
// entries make the reduce launch's in-bucket walk the long pole. Kept at 2^17.)
constexpr int64_t kBucketedMaxPoints = 1 << 17;
constexpr int kBucketLds = 2048;   // entries of a bucket staged in LDS
constexpr int kReduceBlock = 256;
constexpr int kMaxBucketWidth = 2048;  // slots of a bucket: n_slots / 512

template <typename T>
__global__ void __launch_bounds__(kSortBlock)
VdsInsertBucketKernel(const T* __restrict__ pos, const int* n_dev, int n_host,
                      T vs, VdsTable tb, int bshift,
                      int* __restrict__ slot_of_point,
                      int* __restrict__ tile_hist, int* __restrict__ err) {
    // ONE point per lane: a point's insert is a chain of dependent global
    // atomics (CAS on the key, atomicMin on the first point), so every point
    // wants its own lane -- the first version gave a lane the 8 points of a
    // tile slice and took 43 us where the plain insert took 7. A workgroup
    // covers 1024 consecutive points (an eighth of a scatter tile) and adds
    // its bucket counts to the tile's row with one atomic per occupied bucket
    // (the rows are zero between levels: the reduce launch clears them).
    __shared__ int h[kBuckets];
    const int i = blockIdx.x * kSortBlock + threadIdx.x;
    T p[3] = {T(0), T(0), T(0)};
    if (i < n_host) {
        p[0] = pos[3 * (int64_t)i + 0];
        p[1] = pos[3 * (int64_t)i + 1];
        p[2] = pos[3 * (int64_t)i + 2];
    }
    const int n = LiveCount(n_dev, n_host);
    if ((int)(blockIdx.x * kSortBlock) >= n) return;
    for (int b = threadIdx.x; b < kBuckets; b += kSortBlock) h[b] = 0;
    __syncthreads();
    if (i < n) {
        // (p / vs).Floor().To(Int64)
        const long long cx = (long long)floor(p[0] / vs);
        const long long cy = (long long)floor(p[1] / vs);
        const long long cz = (long long)floor(p[2] / vs);
        if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias ||
            cy >= kKeyBias || cz < -kKeyBias || cz >= kKeyBias) {
            // reported to the caller; the point stays a voxel of its own
            atomicOr(err, kErrKeyRange);
            slot_of_point[i] = -1;
            atomicAdd(&h[0], 1);
        } else {
            const unsigned long long key = PackKey((int)cx, (int)cy, (int)cz);
            unsigned s = HashKey(key) & tb.mask;
            while (true) {
                unsigned long long cur = tb.keys[s];
                if (cur == kEmptyKey)
                    cur = atomicCAS(&tb.keys[s], kEmptyKey, key);
                if (cur == kEmptyKey || cur == key) break;
                s = (s + 1) & tb.mask;
            }
            slot_of_point[i] = (int)s;
            atomicMin(&tb.first[s], i);
            atomicAdd(&h[s >> bshift], 1);
        }
    }
    __syncthreads();
    int* row = tile_hist +
               (int64_t)(blockIdx.x / (kSortTile / kSortBlock)) * kBuckets;
    for (int b = threadIdx.x; b < kBuckets; b += kSortBlock)
        if (h[b]) atomicAdd(&row[b], h[b]);
}

__global__ void __launch_bounds__(kSortBlock)
VdsBucketScatterKernel(const int* __restrict__ slot_of_point, VdsTable tb,
                       int bshift, const int* n_dev, int n_host,
                       const int* __restrict__ tile_hist,
                       unsigned* __restrict__ ent_slot,
                       unsigned* __restrict__ ent_point,
                       unsigned long long* __restrict__ first_bits,
                       int* __restrict__ bucket_start) {
    __shared__ int wh[kSortWaves][kBuckets];  // 32 KiB
    __shared__ int dbase[kBuckets];
    __shared__ int lds4[kSortWaves];
    const int tile = blockIdx.x * kSortTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wbase = tile + wave * (kSortItems * 64);
    int slot[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        slot[r] = e < n_host ? slot_of_point[e] : -1;
    }
    // The column of the tile histogram this thread will sum, requested with
    // the slots and the live count (one round trip instead of two): the rows
    // of tiles past the live count are zero -- the reduce launch clears the
    // table and the insert launch only adds to live tiles -- so all
    // ceil(n_host / tile) <= 16 rows can be summed without knowing n.
    constexpr int kMaxTiles = (int)(kBucketedMaxPoints / kSortTile);
    static_assert(kMaxTiles <= 16, "one batch of column loads");
    const int n_tiles_host = (n_host + kSortTile - 1) / kSortTile;
    int colv[kMaxTiles];
#pragma unroll
    for (int u = 0; u < kMaxTiles; ++u)
        colv[u] = (int)threadIdx.x < kBuckets && u < n_tiles_host
                          ? tile_hist[(int64_t)u * kBuckets + threadIdx.x]
                          : 0;
    const int n = LiveCount(n_dev, n_host);
    // block 0 always runs: it publishes the bucket starts (all zero for an
    // empty cloud)
    if (tile >= n && blockIdx.x != 0) return;
    // first point of its voxel? (a point outside the key range is a voxel of
    // its own)
    int first[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        first[r] = e < n && slot[r] >= 0 ? tb.first[slot[r]] : e;
    }
    for (int b = threadIdx.x; b < kBuckets * kSortWaves; b += kSortBlock)
        (&wh[0][0])[b] = 0;
    __syncthreads();
    unsigned digit[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const int e = wbase + r * 64 + lane;
        digit[r] = slot[r] >= 0 ? (unsigned)slot[r] >> bshift : 0u;
        const unsigned long long fm = __ballot(e < n && first[r] == e);
        if (lane == 0 && wbase + r * 64 < n) first_bits[(wbase + r * 64) >> 6] = fm;
        if (e < n) atomicAdd(&wh[wave][digit[r]], 1);
    }
    {
        // where this tile's run of every bucket starts (SortScatterKernel)
        int all = 0, before = 0;
#pragma unroll
        for (int u = 0; u < kMaxTiles; ++u) {
            all += colv[u];
            before += u < (int)blockIdx.x ? colv[u] : 0;
        }
        int total;
        const int run = BlockExclusive(all, lds4, &total);
        if ((int)threadIdx.x < kBuckets) {
            dbase[threadIdx.x] = run + before;
            if (blockIdx.x == 0) {
                bucket_start[threadIdx.x] = run;
                if (threadIdx.x == kBuckets - 1) bucket_start[kBuckets] = total;
            }
        }
    }
    __syncthreads();
    if (tile >= n) return;
    for (int b = threadIdx.x; b < kBuckets; b += kSortBlock) {
        int off = dbase[b];
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
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
        const unsigned d = digit[r];
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kBucketBits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        int posn = 0;
        if (valid) posn = wh[wave][d] + __popcll(same & lt);
        if (valid && (same & lt) == 0ull) wh[wave][d] += __popcll(same);
        if (valid) {
            ent_slot[posn] = (unsigned)slot[r];
            ent_point[posn] = (unsigned)e;
        }
    }
}

// The NEXT (coarser) level's hash insert, carried by this level's reduce
// launch: a pyramid is built from its own output (Registration.cpp:233-270),
// so the lane that writes a voxel's mean can insert that point -- it has its
// index (the output row) and its coordinates in registers -- into the next
// level's table and count it in the next level's tile histogram, which is all
// VdsInsertBucketKernel would do one launch later. tb.keys == NULL: no next
// level (or a caller that makes several passes per level).
template <typename T>
struct VdsNext {
    VdsTable tb;
    T vs;
    int* slot_of_point;
    int* tile_hist;
    int* err;
};

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
VdsBucketReduceKernel(const T* __restrict__ pos, const T* __restrict__ nrm,
                      const unsigned* __restrict__ ent_slot,
                      const unsigned* __restrict__ ent_point,
                      const unsigned long long* __restrict__ first_bits,
                      const int* __restrict__ bucket_start, VdsTable tb,
                      int bshift, int* __restrict__ tile_hist, int n_tiles_cap,
                      const int* n_dev, int n_host, T* __restrict__ out_pos,
                      T* __restrict__ out_nrm, int* __restrict__ m_dev,
                      VdsNext<T> next) {
    __shared__ int word_prefix[kBucketedMaxPoints / 64];  // 16 KiB
    __shared__ int lds4[kReduceBlock / 64];
    __shared__ unsigned e_slot[kBucketLds + 4];  // + a sentinel chunk
    __shared__ float e_pos[kBucketLds][3];
    __shared__ float e_nrm[kBucketLds][3];
    __shared__ int members[kMaxBucketWidth];  // points per slot of the bucket
    const int tid = threadIdx.x;
    // The first-point bits of the whole cloud (<= 2048 words, 8 per lane),
    // requested before anything that depends on the bucket's range: they
    // depend on nothing but the buffer size (words past the live count are
    // masked below).
    constexpr int kWordsPer = (int)(kBucketedMaxPoints / 64) / kReduceBlock;  // 8
    const int n_words_host = (n_host + 63) >> 6;
    unsigned long long w[kWordsPer];
#pragma unroll
    for (int k = 0; k < kWordsPer; ++k) {
        const int wi = tid * kWordsPer + k;
        w[k] = wi < n_words_host ? first_bits[wi] : 0ull;
    }
    const int b0 = bucket_start[blockIdx.x], b1 = bucket_start[blockIdx.x + 1];
    const int n = LiveCount(n_dev, n_host);
    const int count = b1 - b0;
    const int width = 1 << bshift;  // slots of a bucket
    const bool staged = count <= kBucketLds && width <= kMaxBucketWidth;
    // ---- every load of the bucket in flight: entries, then their points ----
    constexpr int kPer = kBucketLds / kReduceBlock;  // 8 entries per lane
    unsigned es[kPer], ep[kPer];
    if (staged) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            es[k] = j < count ? ent_slot[b0 + j] : 0xFFFFFFFEu;
            ep[k] = j < count ? ent_point[b0 + j] : 0u;
        }
        for (int q = tid; q < width; q += kReduceBlock) members[q] = 0;
    }
    // this bucket's column of the tile histogram back to zero for the next
    // level (the scatter launch has read it)
    if (tid < n_tiles_cap) tile_hist[(int64_t)tid * kBuckets + blockIdx.x] = 0;
    // ---- prefix popcount of the first-point bits (all words, every bucket) --
    const int n_words = (n + 63) >> 6;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < kWordsPer; ++k) {
        const int wi = tid * kWordsPer + k;
        if (wi >= n_words) w[k] = 0ull;  // never written by this level
        mine += __popcll(w[k]);
    }
    unsigned long long ew[kPer];  // the first-point word of every entry
    if (staged) {
        float pf[kPer][3], qf[kPer][3];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            const int64_t i = j < count ? (int64_t)ep[k] : 0;
            ew[k] = first_bits[i >> 6];  // with the gathers: one round trip
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pf[k][c] = (float)pos[3 * i + c];
                qf[k][c] = nrm ? (float)nrm[3 * i + c] : 0.f;
            }
        }
        __syncthreads();  // members[] is zero
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            if (j < count) {
                e_slot[j] = es[k];
                if (es[k] != 0xFFFFFFFFu)
                    atomicAdd(&members[es[k] & (unsigned)(width - 1)], 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    e_pos[j][c] = pf[k][c];
                    e_nrm[j][c] = qf[k][c];
                }
            }
        }
        if (tid < 4) e_slot[count + tid] = 0xFFFFFFFEu;  // matches no slot
    }
    {
        // block exclusive scan of `mine` (kReduceBlock threads)
        const int lane = tid & 63, wave = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) lds4[wave] = incl;
        __syncthreads();
        int run = incl - mine, total = 0;
#pragma unroll
        for (int k = 0; k < kReduceBlock / 64; ++k) {
            if (k < wave) run += lds4[k];
            total += lds4[k];
        }
#pragma unroll
        for (int k = 0; k < kWordsPer; ++k) {
            word_prefix[tid * kWordsPer + k] = run;
            run += __popcll(w[k]);
        }
        if (blockIdx.x == 0 && tid == 0) *m_dev = total;
    }
    __syncthreads();
    // ---- one lane per entry; the lane of a voxel's first point adds it up ---
    auto entry = [&](int j, unsigned s, unsigned i,
                     unsigned long long word) {
        // the slot goes back to the empty state (members write the same)
        if (s != 0xFFFFFFFFu) {
            tb.keys[s] = kEmptyKey;
            tb.first[s] = 0x7FFFFFFF;
        }
        if (!((word >> (i & 63)) & 1ull)) return;
        const int row = word_prefix[i >> 6] +
                        __popcll(word & ((1ull << (i & 63)) - 1ull));
        float cnt = 0.f, sp[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        auto add = [&](const float* pp, const float* qq) {
            cnt += 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sp[c] += pp[c];
                sn[c] += qq[c];
            }
        };
        if (staged) {
            // The members follow the first point in point order. Their number
            // is known, so the walk ends at the last one -- a voxel's points
            // are neighbours in the image and therefore in the list -- and
            // four entries are compared per LDS round trip.
            add(e_pos[j], e_nrm[j]);
            int left = s != 0xFFFFFFFFu
                               ? members[s & (unsigned)(width - 1)] - 1
                               : 0;
            for (int e = j + 1; left > 0 && e < count; e += 4) {
                const unsigned s0 = e_slot[e], s1 = e_slot[e + 1],
                               s2 = e_slot[e + 2], s3 = e_slot[e + 3];
                if (s0 == s) { add(e_pos[e], e_nrm[e]); --left; }
                if (s1 == s) { add(e_pos[e + 1], e_nrm[e + 1]); --left; }
                if (s2 == s) { add(e_pos[e + 2], e_nrm[e + 2]); --left; }
                if (s3 == s) { add(e_pos[e + 3], e_nrm[e + 3]); --left; }
            }
        } else {
            // a bucket beyond the LDS staging (a cloud whose points crowd a
            // few voxels): the same walk out of global memory
            for (int e = j; e < count; ++e) {
                const bool member =
                        e == j || (s != 0xFFFFFFFFu && ent_slot[b0 + e] == s);
                if (member) {
                    const int64_t pi = ent_point[b0 + e];
                    float pp[3], qq[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        pp[c] = (float)pos[3 * pi + c];
                        qq[c] = nrm ? (float)nrm[3 * pi + c] : 0.f;
                    }
                    add(pp, qq);
                }
            }
        }
        T o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = (T)(sp[c] / cnt);
            out_pos[3 * (int64_t)row + c] = o[c];
            if (nrm) out_nrm[3 * (int64_t)row + c] = (T)(sn[c] / cnt);
        }
        if (next.tb.keys) {
            // VdsInsertBucketKernel's body for point `row` of the next level
            int* hist = next.tile_hist + (int64_t)(row / kSortTile) * kBuckets;
            const long long cx = (long long)floor(o[0] / next.vs);
            const long long cy = (long long)floor(o[1] / next.vs);
            const long long cz = (long long)floor(o[2] / next.vs);
            if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias ||
                cy >= kKeyBias || cz < -kKeyBias || cz >= kKeyBias) {
                atomicOr(next.err, kErrKeyRange);
                next.slot_of_point[row] = -1;
                atomicAdd(&hist[0], 1);
            } else {
                const unsigned long long key =
                        PackKey((int)cx, (int)cy, (int)cz);
                unsigned ns = HashKey(key) & next.tb.mask;
                while (true) {
                    unsigned long long cur = next.tb.keys[ns];
                    if (cur == kEmptyKey)
                        cur = atomicCAS(&next.tb.keys[ns], kEmptyKey, key);
                    if (cur == kEmptyKey || cur == key) break;
                    ns = (ns + 1) & next.tb.mask;
                }
                next.slot_of_point[row] = (int)ns;
                atomicMin(&next.tb.first[ns], row);
                atomicAdd(&hist[ns >> bshift], 1);
            }
        }
    };
    if (staged) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            if (j < count) entry(j, es[k], ep[k], ew[k]);
        }
    } else {
        for (int j = tid; j < count; j += kReduceBlock) {
            const unsigned i = ent_point[b0 + j];
            entry(j, ent_slot[b0 + j], i, first_bits[i >> 6]);
        }
    }
}

// Persistent buffers of the bucketed form, one set per host thread, device and
// chain (the ICP driver runs the source and the target pyramid as two chains
// on two streams). Everything a level needs is sized by the largest cloud the
// chain has seen; the hash table is returned clean by every level.
struct VdsWorkspace {
    int64_t n_cap = 0, n_slots = 0;
    // Two sets of {table, slot of every point, tile histogram}: a level that
    // carries the next level's insert (VdsNext) fills the other set while it
    // empties its own.
    VdsTable tb = {}, tb2 = {};
    int *slot_of_point = nullptr, *slot_of_point2 = nullptr;
    int *tile_hist = nullptr, *tile_hist2 = nullptr;
    // the insert a reduce launch has left behind: which set, of which cloud
    // (the reduce launch's output), for which voxel size
    bool primed = false;
    int primed_set = 0;
    const void* primed_src = nullptr;
    double primed_vs = 0;
    int64_t primed_n_max = 0;
    size_t primed_esz = 0;
    unsigned* ent_slot = nullptr;
    unsigned* ent_point = nullptr;
    unsigned long long* first_bits = nullptr;
    int* bucket_start = nullptr;
    void Free() {
        (void)hipFree(tb.keys);
        (void)hipFree(tb.first);
        (void)hipFree(slot_of_point);
        (void)hipFree(tile_hist);
        (void)hipFree(tb2.keys);
        (void)hipFree(tb2.first);
        (void)hipFree(slot_of_point2);
        (void)hipFree(tile_hist2);
        (void)hipFree(ent_slot);
        (void)hipFree(ent_point);
        (void)hipFree(first_bits);
        (void)hipFree(bucket_start);
        *this = VdsWorkspace();
    }
};
constexpr int kVdsChains = 2;
constexpr int kVdsDevices = 64;

// The workspaces are self-cleaning: the LAST launch of a level returns the
// table slots, histograms and `primed` state it used to their idle values. A
// chain that is abandoned between its first launch and that last one (an error
// return in the driver, a failed launch) leaves them dirty; the driver says so
// (VdsChainInvalidate) and the next user of the workspace throws it away and
// starts from freshly initialised buffers. [bucketed form, sort form]
static thread_local bool g_vds_dirty[kVdsDevices][kVdsChains][2];

static bool TakeVdsDirty(int dev, int chain, int which) {
    const bool d = g_vds_dirty[dev][chain][which];
    g_vds_dirty[dev][chain][which] = false;
    return d;
}

VdsWorkspace* ThreadVdsWorkspace(int chain, int64_t n_max, hipStream_t s) {
    static thread_local VdsWorkspace ws[kVdsDevices][kVdsChains];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVdsDevices ||
        chain < 0 || chain >= kVdsChains)
        return nullptr;
    VdsWorkspace& w = ws[dev][chain];
    if (TakeVdsDirty(dev, chain, 0) && w.n_cap) {
        // whatever the abandoned chain queued (on whichever stream) is over
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;
        w.Free();
    }
    if (w.n_cap >= n_max) return &w;
    // grow: everything queued on the old buffers must have run
    if (w.n_cap && hipStreamSynchronize(s) != hipSuccess) return nullptr;
    w.Free();
    int64_t cap = 16384;
    while (cap < n_max) cap <<= 1;
    int64_t n_slots = 1024;
    while (n_slots < 2 * cap) n_slots <<= 1;
    const int64_t n_tiles = (cap + kSortTile - 1) / kSortTile;
    bool ok = hipMalloc((void**)&w.tb.keys, sizeof(unsigned long long) * n_slots) == hipSuccess &&
              hipMalloc((void**)&w.tb.first, sizeof(int) * n_slots) == hipSuccess &&
              hipMalloc((void**)&w.slot_of_point, sizeof(int) * cap) == hipSuccess &&
              hipMalloc((void**)&w.tile_hist, sizeof(int) * kBuckets * n_tiles) == hipSuccess &&
              hipMalloc((void**)&w.tb2.keys, sizeof(unsigned long long) * n_slots) == hipSuccess &&
              hipMalloc((void**)&w.tb2.first, sizeof(int) * n_slots) == hipSuccess &&
              hipMalloc((void**)&w.slot_of_point2, sizeof(int) * cap) == hipSuccess &&
              hipMalloc((void**)&w.tile_hist2, sizeof(int) * kBuckets * n_tiles) == hipSuccess &&
              hipMalloc((void**)&w.ent_slot, sizeof(unsigned) * cap) == hipSuccess &&
              hipMalloc((void**)&w.ent_point, sizeof(unsigned) * cap) == hipSuccess &&
              hipMalloc((void**)&w.first_bits, sizeof(unsigned long long) * (cap / 64 + 1)) == hipSuccess &&
              hipMalloc((void**)&w.bucket_start, sizeof(int) * (kBuckets + 1)) == hipSuccess;
    if (ok) {
        w.tb.mask = w.tb2.mask = (unsigned)(n_slots - 1);
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)),
                           dim3(kBlock), 0, s, w.tb, n_slots);
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)),
                           dim3(kBlock), 0, s, w.tb2, n_slots);
        ok = hipGetLastError() == hipSuccess &&
             hipMemsetAsync(w.tile_hist, 0, sizeof(int) * kBuckets * n_tiles,
                            s) == hipSuccess &&
             hipMemsetAsync(w.tile_hist2, 0, sizeof(int) * kBuckets * n_tiles,
                            s) == hipSuccess;
    }
    if (!ok) {
        w.Free();
        SetLastError("VoxelDownSample: workspace allocation failed");
        return nullptr;
    }
    w.n_cap = cap;
    w.n_slots = n_slots;
    return &w;
}

template <typename T>
int VdsBucketedImpl(const T* pos, const T* nrm, int64_t n_max, const int* n_dev,
                    double voxel_size, T* out_pos, T* out_nrm, int* m_dev,
                    int* err_dev, int chain, hipStream_t s,
                    double next_voxel_size, bool from_previous) {
    VdsWorkspace* w = ThreadVdsWorkspace(chain, n_max, s);
    if (!w) return O3DMI_ERR_HIP;
    int slot_bits = 0;
    while ((1ll << slot_bits) < w->n_slots) ++slot_bits;
    const int bshift = slot_bits - kBucketBits;  // n_slots >= 1024 > 512
    const int n_host = (int)n_max;
    const int n_tiles = (int)((n_max + kSortTile - 1) / kSortTile);
    const int64_t n_tiles_cap = (w->n_cap + kSortTile - 1) / kSortTile;
    const dim3 tiles((unsigned)n_tiles), sblock(kSortBlock);
    const dim3 chunks((unsigned)((n_max + kSortBlock - 1) / kSortBlock));
    static const bool no_fuse = std::getenv("O3DMI_VDS_NO_FUSE") != nullptr;
    // Was this cloud inserted by the launch that wrote it?
    const bool inserted = w->primed && from_previous &&
                          w->primed_src == (const void*)pos &&
                          w->primed_vs == voxel_size &&
                          w->primed_n_max == n_max &&
                          w->primed_esz == sizeof(T);
    if (w->primed && !inserted) {
        // an insert nobody came for (the caller changed its mind between two
        // levels): back to the clean state
        VdsTable& tb = w->primed_set ? w->tb2 : w->tb;
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(w->n_slots, kBlock)),
                           dim3(kBlock), 0, s, tb, w->n_slots);
        O3DMI_HIP_CHECK(hipMemsetAsync(
                w->primed_set ? w->tile_hist2 : w->tile_hist, 0,
                sizeof(int) * kBuckets * n_tiles_cap, s));
    }
    const int cur = inserted ? w->primed_set : 0;
    w->primed = false;
    VdsTable& tb = cur ? w->tb2 : w->tb;
    int* slot_of_point = cur ? w->slot_of_point2 : w->slot_of_point;
    int* tile_hist = cur ? w->tile_hist2 : w->tile_hist;
    VdsNext<T> next = {};
    if (next_voxel_size > 0 && !no_fuse) {
        next.tb = cur ? w->tb : w->tb2;
        next.vs = (T)next_voxel_size;
        next.slot_of_point = cur ? w->slot_of_point : w->slot_of_point2;
        next.tile_hist = cur ? w->tile_hist : w->tile_hist2;
        next.err = err_dev;
    }
    if (!inserted)
        hipLaunchKernelGGL(VdsInsertBucketKernel<T>, chunks, sblock, 0, s, pos,
                           n_dev, n_host, (T)voxel_size, tb, bshift,
                           slot_of_point, tile_hist, err_dev);
    hipLaunchKernelGGL(VdsBucketScatterKernel, tiles, sblock, 0, s,
                       slot_of_point, tb, bshift, n_dev, n_host, tile_hist,
                       w->ent_slot, w->ent_point, w->first_bits,
                       w->bucket_start);
    hipLaunchKernelGGL(VdsBucketReduceKernel<T>, dim3(kBuckets),
                       dim3(kReduceBlock), 0, s, pos, nrm, w->ent_slot,
                       w->ent_point, w->first_bits, w->bucket_start, tb,
                       bshift, tile_hist, n_tiles, n_dev, n_host, out_pos,
                       out_nrm, m_dev, next);
    O3DMI_HIP_CHECK(hipGetLastError());
    if (next.tb.keys) {
        w->primed = true;
        w->primed_set = 1 - cur;
        w->primed_src = (const void*)out_pos;
        w->primed_vs = next_voxel_size;
        w->primed_n_max = n_max;
        w->primed_esz = sizeof(T);
    }
    return O3DMI_OK;
}

// Persistent buffers of the seven-launch sort (clouds beyond the bucketed
// form), same keying as VdsWorkspace.
struct VdsSortWorkspace {
    int64_t n_cap = 0;
    unsigned long long* keys = nullptr;
    int* first = nullptr;
    int *slot_of_point = nullptr, *tile_firsts = nullptr, *hist = nullptr,
        *rank_of_first = nullptr;
    unsigned *keys_a = nullptr, *vals_a = nullptr, *keys_b = nullptr,
             *vals_b = nullptr;
    void Free() {
        void* all[] = {keys, first, slot_of_point, tile_firsts, hist,
                       rank_of_first, keys_a, vals_a, keys_b, vals_b};
        for (void* p : all) (void)hipFree(p);
        *this = VdsSortWorkspace();
    }
};

VdsSortWorkspace* ThreadVdsSortWorkspace(int chain, int64_t n_max,
                                         hipStream_t s) {
    static thread_local VdsSortWorkspace ws[kVdsDevices][kVdsChains];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVdsDevices ||
        chain < 0 || chain >= kVdsChains)
        return nullptr;
    VdsSortWorkspace& w = ws[dev][chain];
    if (TakeVdsDirty(dev, chain, 1) && w.n_cap) {
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;
        w.Free();
    }
    if (w.n_cap >= n_max) return &w;
    if (w.n_cap && hipStreamSynchronize(s) != hipSuccess) return nullptr;
    w.Free();
    int64_t cap = 1 << 17;
    while (cap < n_max) cap <<= 1;
    int64_t n_slots = 1024;
    while (n_slots < 2 * cap) n_slots <<= 1;
    const int64_t n_tiles = (cap + kSortTile - 1) / kSortTile;
    auto get = [](auto** p, size_t count) {
        return hipMalloc((void**)p, sizeof(**p) * count) == hipSuccess;
    };
    const bool ok = get(&w.keys, (size_t)n_slots) && get(&w.first, (size_t)n_slots) &&
                    get(&w.slot_of_point, (size_t)cap) &&
                    get(&w.tile_firsts, (size_t)n_tiles) &&
                    get(&w.hist, (size_t)kSortBins * n_tiles) &&
                    get(&w.rank_of_first, (size_t)cap) &&
                    get(&w.keys_a, (size_t)cap) && get(&w.vals_a, (size_t)cap) &&
                    get(&w.keys_b, (size_t)cap) && get(&w.vals_b, (size_t)cap);
    bool init_ok = ok;
    if (ok) {
        VdsTable tb;
        tb.keys = w.keys;
        tb.first = w.first;
        tb.mask = (unsigned)(n_slots - 1);
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)),
                           dim3(kBlock), 0, s, tb, n_slots);
        init_ok = hipGetLastError() == hipSuccess;
    }
    if (!init_ok) {
        w.Free();
        SetLastError("VoxelDownSample: workspace allocation failed");
        return nullptr;
    }
    w.n_cap = cap;
    return &w;
}

template <typename T>
int VdsAsyncImpl(const T* pos, const T* nrm, int64_t n_max, const int* n_dev,
                 double voxel_size, T* out_pos, T* out_nrm, int* m_dev,
                 int* err_dev, std::vector<void*>& scratch, hipStream_t s,
                 int chain) {
    O3DMI_REQUIRE(n_max > 0 && n_max < (1ll << 30),
                  "VoxelDownSample: bad point count");
    // One set of buffers per host thread, device and chain, sized by the
    // largest cloud the chain has seen and reused by every level and every
    // attribute pass (the calls of a chain are stream-ordered). Round 2 took
    // a fresh pooled set per call: a coloured three-level pyramid held nine
    // full-size sets until its read-back.
    (void)scratch;
    VdsSortWorkspace* ws = ThreadVdsSortWorkspace(chain, n_max, s);
    if (!ws) return O3DMI_ERR_HIP;
    const int n_host = (int)n_max;
    int64_t n_slots = 1024;
    while (n_slots < 2 * n_max) n_slots <<= 1;
    const int n_tiles = (int)((n_max + kSortTile - 1) / kSortTile);
    VdsTable tb;
    tb.keys = ws->keys;
    tb.first = ws->first;
    tb.mask = (unsigned)(n_slots - 1);
    int *slot_of_point = ws->slot_of_point, *tile_firsts = ws->tile_firsts,
        *hist = ws->hist, *rank_of_first = ws->rank_of_first;
    unsigned *keys_a = ws->keys_a, *vals_a = ws->vals_a, *keys_b = ws->keys_b,
             *vals_b = ws->vals_b;
    const dim3 grid(GridFor(n_max, kBlock)), block(kBlock);
    const dim3 tiles((unsigned)n_tiles), sblock(kSortBlock);
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
                       rank_of_first, slot_of_point, tb, n_dev, n_host, out_pos,
                       out_nrm);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace

// The level counts of a pyramid chain straight to the host: one tiny launch at
// the end of the chain writes them into host-mapped memory and publishes a
// sequence word the host spins on (mailbox.h) -- instead of a copy and a
// stream synchronisation per chain. Every word is returned to zero for the
// next chain (the buffer is persistent; a level whose input is empty writes
// no count).
__global__ void PostCountsKernel(int* __restrict__ counts, int n,
                                 double* mail_data, int* mail_flag,
                                 int mail_seq) {
    if ((int)threadIdx.x < n) {
        mail_data[threadIdx.x] = (double)counts[threadIdx.x];
        counts[threadIdx.x] = 0;
    }
    MailboxPublish(mail_flag, mail_seq);
}

int PostCountsAsync(int* counts_dev, int n, double* mail_data, int* mail_flag,
                    int mail_seq, hipStream_t s) {
    O3DMI_REQUIRE(n >= 1 && n <= 32, "too many levels");
    hipLaunchKernelGGL(PostCountsKernel, dim3(1), dim3(64), 0, s, counts_dev, n,
                       mail_data, mail_flag, mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

void VdsChainInvalidate(int chain) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kVdsDevices ||
        chain < 0 || chain >= kVdsChains)
        return;
    g_vds_dirty[dev][chain][0] = g_vds_dirty[dev][chain][1] = true;
}

int VdsAsync(const void* pos, const void* attr, int64_t n_max, const int* n_dev,
             int dtype, double voxel_size, void* out_pos, void* out_attr,
             int* m_dev, int* err_dev, std::vector<void*>& scratch,
             hipStream_t s, int chain, double next_voxel_size,
             bool from_previous) {
    if (n_max > 0 && n_max <= kBucketedMaxPoints) {
        if (dtype == O3DMI_F64)
            return VdsBucketedImpl<double>(
                    (const double*)pos, (const double*)attr, n_max, n_dev,
                    voxel_size, (double*)out_pos, (double*)out_attr, m_dev,
                    err_dev, chain, s, next_voxel_size, from_previous);
        return VdsBucketedImpl<float>((const float*)pos, (const float*)attr,
                                      n_max, n_dev, voxel_size,
                                      (float*)out_pos, (float*)out_attr, m_dev,
                                      err_dev, chain, s, next_voxel_size,
                                      from_previous);
    }
    if (dtype == O3DMI_F64)
        return VdsAsyncImpl<double>((const double*)pos, (const double*)attr,
                                    n_max, n_dev, voxel_size, (double*)out_pos,
                                    (double*)out_attr, m_dev, err_dev, scratch,
                                    s, chain);
    return VdsAsyncImpl<float>((const float*)pos, (const float*)attr, n_max,
                               n_dev, voxel_size, (float*)out_pos,
                               (float*)out_attr, m_dev, err_dev, scratch, s,
                               chain);
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
