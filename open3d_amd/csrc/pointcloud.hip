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

// ==== the tiled form: two launches per level, clouds up to 2^20 points ========
// (round 6; replaces round 3's three-launch bucketed form, whose launches ran
// 10-17 us each for <= 3 MB of traffic: ten 8192-point scatter workgroups on a
// 256-CU chip, 512 global histogram atomics per insert workgroup, the
// first-point bits of the whole cloud re-read by every reduce workgroup. The
// seven-launch sort above stays for larger clouds.)
//
// The sort exists to put every voxel's points side by side in point order.
// Here the points are only PARTITIONED by the high bits of their voxel's hash
// slot (a bucket = 1024 or 2048 consecutive slots, a few hundred points), and
// the partition is never materialised globally: a 1024-point TILE is split
// stably in LDS by its own workgroup and written back tile by tile, with the
// tile's bucket offsets; the bucket's workgroup then picks its segment out of
// every tile, in tile order -- which is point order. No global histogram, no
// column sums, no global scatter.
//   I  VdsTileInsertKernel     (finest level only) hash insert: slot of every
//                              point, first point of every voxel (atomicMin);
//   P  VdsTilePartitionKernel  per tile: "is the first point of its voxel"
//                              (first[slot] == i) with its rank among the
//                              tile's first points, the stable split by bucket
//                              in LDS, 32-byte entries {slot, index | rank |
//                              first flag, position, attribute} written in
//                              bucket order + the tile's bucket offsets and
//                              first-point count;
//   R  VdsTileReduceKernel     per bucket: scans the tiles' segment lengths and
//                              first-point counts (<= 1024 tiles), stages its
//                              entries into LDS with every load in flight, the
//                              lane of a voxel's first point walks the entries
//                              behind it adding its voxel's members in float32
//                              (point order survives the stable split), output
//                              row = first points of the tiles before + rank
//                              in the tile; every entry returns its table slot
//                              to the empty state; and the lane that writes a
//                              voxel's mean inserts it into the NEXT level's
//                              table (VdsNext), so a pyramid is I P R P R P R.
// One or TWO clouds per launch (blockIdx.y): the ICP driver builds the source
// and the target pyramid level by level in the same launches.
constexpr int kTile = kSortBlock;          // 1024 points, one per lane
constexpr int kMaxTiles = 1024;
constexpr int64_t kTiledMaxPoints = (int64_t)kTile * kMaxTiles;  // 2^20
constexpr int kMaxBuckets = 1024;
constexpr int kReduceBlock = 256;
constexpr unsigned kNoSlot = 0xFFFFFFFFu;  // a point outside the key range
constexpr unsigned kEntryFirst = 0x80000000u;
constexpr int kEntryRankShift = 20;        // index 20 bits, rank 10 bits

// The NEXT (coarser) level's hash insert, carried by this level's reduce
// launch: a pyramid is built from its own output (Registration.cpp:233-270),
// so the lane that writes a voxel's mean can insert that point -- it has its
// index (the output row) and its coordinates in registers -- into the next
// level's table. tb.keys == NULL: no next level (or a caller that makes
// several passes per level).
template <typename T>
struct VdsNext {
    VdsTable tb;
    T vs;
    int* slot_of_point;
};

template <typename T>
struct VdsJob {
    const T* pos;
    const T* attr;
    const int* n_dev;
    int n_host;           // 0: nothing to do for this cloud in this launch
    T vs;
    VdsTable tb;
    int* slot_of_point;
    int bshift;           // bucket = slot >> bshift
    int bucket_bits;      // log2(number of buckets)
    uint4* ent;           // two per entry, tile-major
    int* tile_off;        // [tile][buckets + 1]
    int* tile_firsts;     // [tile]
    T* out_pos;
    T* out_attr;
    int* m_dev;
    int* err;
    VdsNext<T> next;
    VdsPost post;         // the chain's counts leave with this level (vds.h)
};

// (p / vs).Floor().To(Int64) -> hash insert; the voxel's first point.
template <typename T>
__device__ __forceinline__ void VdsInsertPoint(T px, T py, T pz, T vs,
                                               const VdsTable& tb, int i,
                                               int* __restrict__ slot_of_point,
                                               int* __restrict__ err) {
    const long long cx = (long long)floor(px / vs);
    const long long cy = (long long)floor(py / vs);
    const long long cz = (long long)floor(pz / vs);
    if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias || cy >= kKeyBias ||
        cz < -kKeyBias || cz >= kKeyBias) {
        // reported to the caller; the point stays a voxel of its own so that
        // the rest of the chain sees consistent keys
        atomicOr(err, kErrKeyRange);
        slot_of_point[i] = -1;
        return;
    }
    const unsigned long long key = PackKey((int)cx, (int)cy, (int)cz);
    unsigned s = HashKey(key) & tb.mask;
    while (true) {
        unsigned long long cur = tb.keys[s];
        if (cur == kEmptyKey) cur = atomicCAS(&tb.keys[s], kEmptyKey, key);
        if (cur == kEmptyKey || cur == key) break;
        s = (s + 1) & tb.mask;
    }
    slot_of_point[i] = (int)s;
    atomicMin(&tb.first[s], i);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
VdsTileInsertKernel(VdsJob<T> job_a, VdsJob<T> job_b) {
    const VdsJob<T>& job = blockIdx.y ? job_b : job_a;
    // ONE point per lane: an insert is a chain of dependent global atomics.
    // The point is fetched alongside the live count (which the previous
    // level or Unproject wrote) and dropped if it lies past it.
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    T p[3] = {T(0), T(0), T(0)};
    if (i < job.n_host) {
        p[0] = job.pos[3 * (int64_t)i + 0];
        p[1] = job.pos[3 * (int64_t)i + 1];
        p[2] = job.pos[3 * (int64_t)i + 2];
    }
    const bool live = i < LiveCount(job.n_dev, job.n_host);
    // (p / vs).Floor().To(Int64)
    const long long cx = (long long)floor(p[0] / job.vs);
    const long long cy = (long long)floor(p[1] / job.vs);
    const long long cz = (long long)floor(p[2] / job.vs);
    const bool in_range = cx >= -kKeyBias && cx < kKeyBias && cy >= -kKeyBias &&
                          cy < kKeyBias && cz >= -kKeyBias && cz < kKeyBias;
    const bool keyed = live && in_range;
    const unsigned long long key =
            keyed ? PackKey((int)cx, (int)cy, (int)cz) : kEmptyKey;
    // Neighbours in the cloud are neighbours in space (Unproject writes in
    // pixel order, a down-sampled level in first-point order): a RUN of lanes
    // with the same voxel sends ONE lane to the table -- its first, which is
    // also the run's smallest point index -- and takes the slot from it.
    // (Every point on its own: 2.5 fabric atomics each, 17 us for the two
    // 77 k-point clouds of a VGA frame and 46 us at 1280x720.)
    const unsigned long long prev = __shfl_up(key, 1);
    const bool head = keyed && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head);
    int slot = -1;
    if (head) {
        unsigned s = HashKey(key) & job.tb.mask;
        while (true) {
            unsigned long long cur = job.tb.keys[s];
            if (cur == kEmptyKey)
                cur = atomicCAS(&job.tb.keys[s], kEmptyKey, key);
            if (cur == kEmptyKey || cur == key) break;
            s = (s + 1) & job.tb.mask;
        }
        slot = (int)s;
        atomicMin(&job.tb.first[s], i);
    }
    // the run's head: the highest head lane at or below this one
    const unsigned long long upto = heads & (~0ull >> (63 - lane));
    const int head_lane = upto ? 63 - __clzll((long long)upto) : lane;
    slot = __shfl(slot, head_lane);
    if (!live) return;
    if (!in_range) {
        // reported to the caller; the point stays a voxel of its own so that
        // the rest of the chain sees consistent keys
        atomicOr(job.err, kErrKeyRange);
        slot = -1;
    }
    job.slot_of_point[i] = slot;
}

template <typename T>
__global__ void __launch_bounds__(kTile)
VdsTilePartitionKernel(VdsJob<T> job_a, VdsJob<T> job_b) {
    __shared__ int wh[kTile / 64][kMaxBuckets];  // 64 KiB
    __shared__ int lds4[kTile / 64];
    __shared__ int wave_firsts[kTile / 64];
    const VdsJob<T>& job = blockIdx.y ? job_b : job_a;
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = tile * kTile + tid;
    // everything that depends on the buffer size only, requested with the
    // live count
    int slot = -1;
    float pf[3] = {0.f, 0.f, 0.f}, af[3] = {0.f, 0.f, 0.f};
    if (i < job.n_host) {
        slot = job.slot_of_point[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pf[c] = (float)job.pos[3 * (int64_t)i + c];
            if (job.attr) af[c] = (float)job.attr[3 * (int64_t)i + c];
        }
    }
    const int n = LiveCount(job.n_dev, job.n_host);
    if (tile * kTile >= n) return;
    const bool valid = i < n;
    // (a slot read past the live count is whatever the buffer held)
    const bool keyed = valid && slot >= 0;
    const int first = keyed ? job.tb.first[(unsigned)slot & job.tb.mask] : i;
    const bool is_first = valid && first == i;
    const int bits = job.bucket_bits, buckets = 1 << bits;
    for (int q = tid; q < (kTile / 64) << bits; q += kTile)
        wh[q >> bits][q & (buckets - 1)] = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long fm = __ballot(is_first);
    if (lane == 0) wave_firsts[wave] = __popcll(fm);
    __syncthreads();
    const unsigned d = keyed ? ((unsigned)slot & job.tb.mask) >> job.bshift : 0u;
    if (valid) atomicAdd(&wh[wave][d], 1);
    __syncthreads();
    // bucket b of this tile: its size, then where each wave's share starts
    int mine = 0;
    if (tid < buckets) {
#pragma unroll
        for (int w = 0; w < kTile / 64; ++w) mine += wh[w][tid];
    }
    int tile_total;
    const int start = BlockExclusive(mine, lds4, &tile_total);
    int* row = job.tile_off + (int64_t)tile * (buckets + 1);
    if (tid < buckets) {
        row[tid] = start;
        int off = start;
#pragma unroll
        for (int w = 0; w < kTile / 64; ++w) {
            const int c = wh[w][tid];
            wh[w][tid] = off;
            off += c;
        }
    }
    int rank = __popcll(fm & lt), firsts = 0;
#pragma unroll
    for (int w = 0; w < kTile / 64; ++w) {
        const int c = wave_firsts[w];
        rank += w < wave ? c : 0;
        firsts += c;
    }
    if (tid == 0) {
        row[buckets] = tile_total;
        job.tile_firsts[tile] = firsts;
    }
    __syncthreads();
    // lanes of this wave in the same bucket, in lane (= point) order
    unsigned long long same = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        same &= bit ? bal : ~bal;
    }
    if (!valid) return;
    const int posn = wh[wave][d] + __popcll(same & lt);
    uint4 e0, e1;
    e0.x = keyed ? (unsigned)slot & job.tb.mask : kNoSlot;
    e0.y = (unsigned)i | ((unsigned)rank << kEntryRankShift) |
           (is_first ? kEntryFirst : 0u);
    e0.z = __float_as_uint(pf[0]);
    e0.w = __float_as_uint(pf[1]);
    e1.x = __float_as_uint(pf[2]);
    e1.y = __float_as_uint(af[0]);
    e1.z = __float_as_uint(af[1]);
    e1.w = __float_as_uint(af[2]);
    uint4* dst = job.ent + 2 * ((int64_t)tile * kTile + posn);
    dst[0] = e0;
    dst[1] = e1;
}

// LDS of the reduce launch, carved out of one dynamic block sized by the host
// (VdsReduceLdsBytes): the tile tables by the cloud's tile count, the member
// counters by the bucket width, the staging area by kStage entries. 39 KB for
// a 640 x 360 cloud (four workgroups per CU; round 6's first form held 77 KB
// statically and ran the 1024 workgroups of a 720p level in two rounds).
constexpr int kStage = 1024;  // entries of a bucket staged in LDS at a time
struct VdsReduceLds {
    int* seg_lo;         // [tiles_p2] bucket's start inside tile t
    int* seg_start;      // [tiles_p2] entries in the tiles before t
    int* first_base;     // [tiles_p2] first points in the tiles before t
    int* members;        // [width] points per slot of the bucket
    unsigned* e_slot;    // [kStage + 4] (+ a sentinel chunk)
    unsigned* e_iw;      // [kStage] index | rank | first flag
    float* e_pos;        // [kStage][3]
    float* e_attr;       // [kStage][3]
};
inline size_t VdsReduceLdsBytes(int tiles_p2, int width) {
    return sizeof(int) * (3 * (size_t)tiles_p2 + (size_t)width) +
           sizeof(unsigned) * (2 * kStage + 4) + sizeof(float) * 6 * kStage;
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
VdsTileReduceKernel(VdsJob<T> job_a, VdsJob<T> job_b, int tiles_p2,
                    int members_n) {
    extern __shared__ int lds_raw[];
    __shared__ int lds4[2][kReduceBlock / 64];
    __shared__ int wave_sel[kReduceBlock / 64];
    const VdsJob<T>& job = blockIdx.y ? job_b : job_a;
    const int bucket = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int buckets = 1 << job.bucket_bits;
    if (bucket >= buckets || job.n_host <= 0) return;
    const int width = 1 << job.bshift;  // slots of a bucket (<= members_n)
    VdsReduceLds L;
    L.seg_lo = lds_raw;
    L.seg_start = L.seg_lo + tiles_p2;
    L.first_base = L.seg_start + tiles_p2;
    L.members = L.first_base + tiles_p2;
    L.e_slot = (unsigned*)(L.members + members_n);
    L.e_iw = L.e_slot + kStage + 4;
    L.e_pos = (float*)(L.e_iw + kStage);
    L.e_attr = L.e_pos + 3 * kStage;
    // ---- the tiles' tables: `per` consecutive tiles per lane ----------------
    const int per = tiles_p2 >= kReduceBlock ? tiles_p2 / kReduceBlock : 1;
    const int tiles_host = (job.n_host + kTile - 1) / kTile;
    const int n = LiveCount(job.n_dev, job.n_host);
    const int tiles = (n + kTile - 1) / kTile;
    int my_len = 0, my_fc = 0;
    for (int u = 0; u < per; ++u) {
        const int t = tid * per + u;
        int lo = 0, len = 0, fc = 0;
        // rows of tiles past the live count hold an earlier level's numbers
        if (t < tiles_host && t < tiles) {
            const int* row = job.tile_off + (int64_t)t * (buckets + 1) + bucket;
            lo = row[0];
            len = row[1] - lo;
            fc = job.tile_firsts[t];
        }
        if (t < tiles_p2) {
            L.seg_lo[t] = lo;
            L.seg_start[t] = len;    // (lengths; scanned below)
            L.first_base[t] = fc;
        }
        my_len += len;
        my_fc += fc;
    }
    int count, voxels;
    {
        // block exclusive scans of my_len and my_fc (kReduceBlock threads)
        int il = my_len, ic = my_fc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int tl = __shfl_up(il, d), tc = __shfl_up(ic, d);
            if (lane >= d) {
                il += tl;
                ic += tc;
            }
        }
        if (lane == 63) {
            lds4[0][wave] = il;
            lds4[1][wave] = ic;
        }
        __syncthreads();
        int rl = il - my_len, rc = ic - my_fc;
        count = voxels = 0;
#pragma unroll
        for (int k = 0; k < kReduceBlock / 64; ++k) {
            if (k < wave) {
                rl += lds4[0][k];
                rc += lds4[1][k];
            }
            count += lds4[0][k];
            voxels += lds4[1][k];
        }
        for (int u = 0; u < per; ++u) {
            const int t = tid * per + u;
            if (t < tiles_p2) {
                const int len = L.seg_start[t], fc = L.first_base[t];
                L.seg_start[t] = rl;
                L.first_base[t] = rc;
                rl += len;
                rc += fc;
            }
        }
        if (bucket == 0 && tid == 0) *job.m_dev = voxels;
        if (bucket == 0 && wave == 0 && job.post.counts) {
            // The chain's last level: every count of the chain is known NOW
            // -- the earlier levels' and the error word in memory (launches
            // before this one; this level inserts nothing), this level's in
            // `voxels` -- so the host gets them while the level is still
            // being reduced, instead of from a posting launch behind it
            // (4.7 us + a launch boundary per tracked frame).
            int c = 0;
            if (lane < job.post.n) {
                int* cp = job.post.counts +
                          (lane == job.post.n - 1 ? kCountsErr : lane);
                c = cp == job.m_dev ? voxels : *cp;
                job.post.counts[kCountsKeep + lane] = c;
                // zero again for the next chain -- but for the two counts
                // this launch's other workgroups are reading or writing
                if (cp != job.m_dev && cp != job.n_dev) *cp = 0;
            }
            MailboxPostSealedWave(job.post.mail_data, job.post.mail_flag,
                                  job.post.mail_seq, (double)c);
        }
    }
    __syncthreads();
    // entry j of the bucket -> where it lies: the last tile that starts at or
    // before j (empty segments share their successor's start)
    auto locate = [&](int j) -> int64_t {
        int t = 0;
        for (int step = tiles_p2 >> 1; step > 0; step >>= 1)
            if (L.seg_start[t + step] <= j) t += step;
        return (int64_t)t * kTile + L.seg_lo[t] + (j - L.seg_start[t]);
    };
    auto finish = [&](unsigned iw, float cnt, const float* sp,
                      const float* sn) {
        const int i = (int)(iw & ((1u << kEntryRankShift) - 1u));
        const int row = L.first_base[i / kTile] +
                        (int)((iw >> kEntryRankShift) & (unsigned)(kTile - 1));
        T o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = (T)(sp[c] / cnt);
            job.out_pos[3 * (int64_t)row + c] = o[c];
            if (job.attr) job.out_attr[3 * (int64_t)row + c] = (T)(sn[c] / cnt);
        }
        if (job.next.tb.keys)
            VdsInsertPoint(o[0], o[1], o[2], job.next.vs, job.next.tb, row,
                           job.next.slot_of_point, job.err);
    };
    auto clean = [&](unsigned s) {
        // the slot goes back to the empty state (members write the same): the
        // table is clean after every level
        if (s != kNoSlot) {
            job.tb.keys[s] = kEmptyKey;
            job.tb.first[s] = 0x7FFFFFFF;
        }
    };
    // The staged entries 0..m: one lane per entry; the lane of a voxel's first
    // point adds the voxel up. Its members follow it in point order (the
    // stable split), their number is known, so the walk ends at the last one
    // -- a voxel's points are neighbours in the cloud and therefore in the
    // list -- and four entries are compared per LDS round trip.
    auto process_staged = [&](int m) {
        if (tid < 4) L.e_slot[m + tid] = 0xFFFFFFFEu;  // matches no slot
        __syncthreads();
        for (int q = tid; q < m; q += kReduceBlock) {
            const unsigned s = L.e_slot[q];
            const unsigned iw = L.e_iw[q];
            clean(s);
            if (!(iw & kEntryFirst)) continue;
            float cnt = 0.f, sp[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
            auto add = [&](int e) {
                cnt += 1.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    sp[c] += L.e_pos[3 * e + c];
                    sn[c] += L.e_attr[3 * e + c];
                }
            };
            add(q);
            int left = s != kNoSlot ? L.members[s & (unsigned)(width - 1)] - 1
                                    : 0;
            for (int e = q + 1; left > 0 && e < m; e += 4) {
                const unsigned s0 = L.e_slot[e], s1 = L.e_slot[e + 1],
                               s2 = L.e_slot[e + 2], s3 = L.e_slot[e + 3];
                if (s0 == s) { add(e); --left; }
                if (s1 == s) { add(e + 1); --left; }
                if (s2 == s) { add(e + 2); --left; }
                if (s3 == s) { add(e + 3); --left; }
            }
            finish(iw, cnt, sp, sn);
        }
    };
    auto stage_entry = [&](int at, const uint4& e0, const uint4& e1) {
        L.e_slot[at] = e0.x;
        L.e_iw[at] = e0.y;
        if (e0.x != kNoSlot)
            atomicAdd(&L.members[e0.x & (unsigned)(width - 1)], 1);
        L.e_pos[3 * at + 0] = __uint_as_float(e0.z);
        L.e_pos[3 * at + 1] = __uint_as_float(e0.w);
        L.e_pos[3 * at + 2] = __uint_as_float(e1.x);
        L.e_attr[3 * at + 0] = __uint_as_float(e1.y);
        L.e_attr[3 * at + 1] = __uint_as_float(e1.z);
        L.e_attr[3 * at + 2] = __uint_as_float(e1.w);
    };
    for (int q = tid; q < width; q += kReduceBlock) L.members[q] = 0;
    if (count <= kStage) {
        // ---- the usual case: the whole bucket at once, every load in flight -
        constexpr int kPer = kStage / kReduceBlock;  // 4 entries per lane
        uint4 e0[kPer], e1[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            if (j < count) {
                const uint4* src = job.ent + 2 * locate(j);
                e0[k] = src[0];
                e1[k] = src[1];
            }
        }
        __syncthreads();  // members[] is zero
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int j = tid + k * kReduceBlock;
            if (j < count) stage_entry(j, e0[k], e1[k]);
        }
        process_staged(count);
        return;
    }
    // ---- a crowded bucket (a cloud whose points pile up in few voxels): in
    // passes over 2^k ranges of its slots, each pass staging the entries of
    // its range -- stably: in entry order -- and running the same walk. A
    // range that still overflows the staging area (one voxel with more than
    // kStage points) is walked out of global memory.
    int passes = 2;
    while (passes < width && (int64_t)passes * kStage < 2 * (int64_t)count)
        passes <<= 1;
    const int sub_shift = job.bshift - (31 - __clz(passes));  // log2(slots / pass)
    auto range_of = [&](unsigned s) -> int {
        return s == kNoSlot ? 0 : (int)((s & (unsigned)(width - 1)) >> sub_shift);
    };
    for (int p = 0; p < passes; ++p) {
        __syncthreads();  // the previous pass is through with the staging area
        for (int q = tid; q < width; q += kReduceBlock) L.members[q] = 0;
        __syncthreads();
        int base = 0;  // entries of this range before chunk j0 (uniform)
        for (int j0 = 0; j0 < count; j0 += kReduceBlock) {
            const int j = j0 + tid;
            uint4 e0 = make_uint4(0u, 0u, 0u, 0u), e1 = e0;
            bool sel = false;
            if (j < count) {
                const uint4* src = job.ent + 2 * locate(j);
                e0 = src[0];
                sel = range_of(e0.x) == p;
                if (sel) e1 = src[1];
            }
            const unsigned long long bal = __ballot(sel);
            if (lane == 0) wave_sel[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kReduceBlock / 64; ++w) {
                const int c = wave_sel[w];
                before += w < wave ? c : 0;
                total += c;
            }
            const int at = base + before +
                           __popcll(bal & ((1ull << lane) - 1ull));
            if (sel) {
                if (at < kStage) {
                    stage_entry(at, e0, e1);
                } else if (e0.x != kNoSlot) {
                    // (beyond the staging area: only the member counts)
                    atomicAdd(&L.members[e0.x & (unsigned)(width - 1)], 1);
                }
            }
            base += total;
            __syncthreads();  // wave_sel is reused
        }
        if (base <= kStage) {
            process_staged(base);
            continue;
        }
        // this range alone overflows: walk it out of global memory, four
        // entries in flight
        __syncthreads();  // members[] complete
        for (int j = tid; j < count; j += kReduceBlock) {
            const uint4* own = job.ent + 2 * locate(j);
            const uint4 o0 = own[0];
            if (range_of(o0.x) != p || !(o0.y & kEntryFirst)) continue;
            const uint4 o1 = own[1];
            const unsigned s = o0.x;
            float cnt = 1.0f;
            float sp[3] = {__uint_as_float(o0.z), __uint_as_float(o0.w),
                           __uint_as_float(o1.x)};
            float sn[3] = {__uint_as_float(o1.y), __uint_as_float(o1.z),
                           __uint_as_float(o1.w)};
            int left = s != kNoSlot ? L.members[s & (unsigned)(width - 1)] - 1
                                    : 0;
            for (int e = j + 1; left > 0 && e < count; e += 4) {
                uint4 m0[4], m1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    m0[u] = make_uint4(0xFFFFFFFEu, 0u, 0u, 0u);
                    m1[u] = make_uint4(0u, 0u, 0u, 0u);
                    if (e + u < count) {
                        const uint4* src = job.ent + 2 * locate(e + u);
                        m0[u] = src[0];
                        m1[u] = src[1];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (m0[u].x == s) {
                        cnt += 1.0f;
                        sp[0] += __uint_as_float(m0[u].z);
                        sp[1] += __uint_as_float(m0[u].w);
                        sp[2] += __uint_as_float(m1[u].x);
                        sn[0] += __uint_as_float(m1[u].y);
                        sn[1] += __uint_as_float(m1[u].z);
                        sn[2] += __uint_as_float(m1[u].w);
                        --left;
                    }
            }
            finish(o0.y, cnt, sp, sn);
        }
        // (only when every walk of the range has read its slots)
        __syncthreads();
        for (int j = tid; j < count; j += kReduceBlock) {
            const unsigned s = job.ent[2 * locate(j)].x;
            if (range_of(s) == p) clean(s);
        }
    }
}

// Persistent buffers of the tiled form, one set per host thread, device and
// chain (the source and the target pyramid of the ICP driver are two chains,
// built in the same launches). Everything a level needs is sized by the
// largest cloud the chain has seen; the hash table is returned clean by every
// level.
struct VdsWorkspace {
    int64_t n_cap = 0, n_slots = 0;
    // Two sets of {table, slot of every point}: a level that carries the next
    // level's insert (VdsNext) fills the other set while it empties its own.
    VdsTable tb = {}, tb2 = {};
    int *slot_of_point = nullptr, *slot_of_point2 = nullptr;
    // the insert a reduce launch has left behind: which set, of which cloud
    // (the reduce launch's output), for which voxel size
    bool primed = false;
    int primed_set = 0;
    const void* primed_src = nullptr;
    double primed_vs = 0;
    int64_t primed_n_max = 0;
    size_t primed_esz = 0;
    uint4* ent = nullptr;        // [2 * n_cap]
    int* tile_off = nullptr;     // [n_cap / kTile][kMaxBuckets + 1]
    int* tile_firsts = nullptr;  // [n_cap / kTile]
    void Free() {
        (void)hipFree(tb.keys);
        (void)hipFree(tb.first);
        (void)hipFree(slot_of_point);
        (void)hipFree(tb2.keys);
        (void)hipFree(tb2.first);
        (void)hipFree(slot_of_point2);
        (void)hipFree(ent);
        (void)hipFree(tile_off);
        (void)hipFree(tile_firsts);
        *this = VdsWorkspace();
    }
};
constexpr int kVdsChains = 2;
constexpr int kVdsDevices = 64;

// The workspaces are self-cleaning: the LAST launch of a level returns the
// table slots and `primed` state it used to their idle values. A chain that is
// abandoned between its first launch and that last one (an error return in
// the driver, a failed launch) leaves them dirty; the driver says so
// (VdsChainInvalidate) and the next user of the workspace throws it away and
// starts from freshly initialised buffers. [tiled form, sort form]
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
    const int64_t n_slots = 2 * cap;
    const int64_t n_tiles = cap / kTile;
    auto get = [](auto** p, size_t count) {
        return hipMalloc((void**)p, sizeof(**p) * count) == hipSuccess;
    };
    bool ok = get(&w.tb.keys, (size_t)n_slots) &&
              get(&w.tb.first, (size_t)n_slots) &&
              get(&w.slot_of_point, (size_t)cap) &&
              get(&w.tb2.keys, (size_t)n_slots) &&
              get(&w.tb2.first, (size_t)n_slots) &&
              get(&w.slot_of_point2, (size_t)cap) &&
              get(&w.ent, (size_t)(2 * cap)) &&
              get(&w.tile_off, (size_t)(n_tiles * (kMaxBuckets + 1))) &&
              get(&w.tile_firsts, (size_t)n_tiles);
    if (ok) {
        w.tb.mask = w.tb2.mask = (unsigned)(n_slots - 1);
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)),
                           dim3(kBlock), 0, s, w.tb, n_slots);
        hipLaunchKernelGGL(VdsInitKernel, dim3(GridFor(n_slots, kBlock)),
                           dim3(kBlock), 0, s, w.tb2, n_slots);
        ok = hipGetLastError() == hipSuccess;
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

// One level of one or two clouds (two chains) in the same launches.
template <typename T>
int VdsTiledImpl(const VdsLevelJob* jobs, int n_jobs, hipStream_t s,
                 bool* posted = nullptr) {
    O3DMI_REQUIRE(n_jobs == 1 || (n_jobs == 2 && jobs[0].chain != jobs[1].chain),
                  "VoxelDownSample: one cloud per chain");
    static const bool no_fuse = std::getenv("O3DMI_VDS_NO_FUSE") != nullptr;
    VdsJob<T> run[2] = {}, ins[2] = {};
    VdsWorkspace* wss[2] = {nullptr, nullptr};
    int cur[2] = {0, 0};
    bool any_insert = false;
    int64_t most_points = 0;
    int most_tiles = 0, most_buckets = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const VdsLevelJob& J = jobs[q];
        O3DMI_REQUIRE(J.n_max > 0 && J.n_max <= kTiledMaxPoints,
                      "VoxelDownSample: bad point count");
        VdsWorkspace* w = ThreadVdsWorkspace(J.chain, J.n_max, s);
        if (!w) return O3DMI_ERR_HIP;
        wss[q] = w;
        // this call's share of the (clean) table: 2 slots per point; a bucket
        // is 1024 slots (2048 for the largest clouds: <= 1024 buckets)
        int slot_bits = 15;
        while ((1ll << slot_bits) < 2 * J.n_max) ++slot_bits;
        const int bshift = slot_bits > 20 ? 11 : 10;
        // Was this cloud inserted by the launch that wrote it?
        const bool inserted = w->primed && J.from_previous &&
                              w->primed_src == J.pos &&
                              w->primed_vs == J.voxel_size &&
                              w->primed_n_max == J.n_max &&
                              w->primed_esz == sizeof(T);
        if (w->primed && !inserted) {
            // an insert nobody came for (the caller changed its mind between
            // two levels): back to the clean state
            VdsTable& tb = w->primed_set ? w->tb2 : w->tb;
            hipLaunchKernelGGL(VdsInitKernel,
                               dim3(GridFor(w->n_slots, kBlock)), dim3(kBlock),
                               0, s, tb, w->n_slots);
        }
        cur[q] = inserted ? w->primed_set : 0;
        w->primed = false;
        VdsJob<T>& d = run[q];
        d.pos = (const T*)J.pos;
        d.attr = (const T*)J.attr;
        d.n_dev = J.n_dev;
        d.n_host = (int)J.n_max;
        d.vs = (T)J.voxel_size;
        d.tb = cur[q] ? w->tb2 : w->tb;
        d.tb.mask = (unsigned)((1ll << slot_bits) - 1);
        d.slot_of_point = cur[q] ? w->slot_of_point2 : w->slot_of_point;
        d.bshift = bshift;
        d.bucket_bits = slot_bits - bshift;
        d.ent = w->ent;
        d.tile_off = w->tile_off;
        d.tile_firsts = w->tile_firsts;
        d.out_pos = (T*)J.out_pos;
        d.out_attr = (T*)J.out_attr;
        d.m_dev = J.m_dev;
        d.err = J.err_dev;
        if (J.next_voxel_size > 0 && !no_fuse) {
            d.next.tb = cur[q] ? w->tb : w->tb2;
            d.next.tb.mask = d.tb.mask;
            d.next.vs = (T)J.next_voxel_size;
            d.next.slot_of_point =
                    cur[q] ? w->slot_of_point : w->slot_of_point2;
        }
        ins[q] = d;
        if (inserted) ins[q].n_host = 0;
        else any_insert = true;
        most_points = J.n_max > most_points ? J.n_max : most_points;
        const int tiles = (int)((J.n_max + kTile - 1) / kTile);
        most_tiles = tiles > most_tiles ? tiles : most_tiles;
        most_buckets = (1 << d.bucket_bits) > most_buckets ? 1 << d.bucket_bits
                                                           : most_buckets;
    }
    // the counts posted by this level's reduce launch: only if every job asks
    // and none of them inserts into a next level (the error word is final)
    bool post = true;
    for (int q = 0; q < n_jobs; ++q)
        post = post && jobs[q].post.counts && jobs[q].post.mail_data &&
               jobs[q].post.mail_flag && jobs[q].post.n >= 1 &&
               jobs[q].post.n <= 32 && !run[q].next.tb.keys;
    if (post)
        for (int q = 0; q < n_jobs; ++q) run[q].post = jobs[q].post;
    if (posted) *posted = post;
    const unsigned gy = (unsigned)n_jobs;
    if (any_insert)
        hipLaunchKernelGGL(VdsTileInsertKernel<T>,
                           dim3((unsigned)((most_points + kBlock - 1) / kBlock),
                                gy),
                           dim3(kBlock), 0, s, ins[0], ins[1]);
    hipLaunchKernelGGL(VdsTilePartitionKernel<T>, dim3((unsigned)most_tiles, gy),
                       dim3(kTile), 0, s, run[0], run[1]);
    int tiles_p2 = 64;
    while (tiles_p2 < most_tiles) tiles_p2 <<= 1;
    int widest = 0;
    for (int q = 0; q < n_jobs; ++q)
        widest = (1 << run[q].bshift) > widest ? 1 << run[q].bshift : widest;
    hipLaunchKernelGGL(VdsTileReduceKernel<T>, dim3((unsigned)most_buckets, gy),
                       dim3(kReduceBlock),
                       (unsigned)VdsReduceLdsBytes(tiles_p2, widest), s, run[0],
                       run[1], tiles_p2, widest);
    O3DMI_HIP_CHECK(hipGetLastError());
    for (int q = 0; q < n_jobs; ++q) {
        if (!run[q].next.tb.keys) continue;
        VdsWorkspace* w = wss[q];
        w->primed = true;
        w->primed_set = 1 - cur[q];
        w->primed_src = jobs[q].out_pos;
        w->primed_vs = jobs[q].next_voxel_size;
        w->primed_n_max = jobs[q].n_max;
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
        int* cp = counts + ((int)threadIdx.x == n - 1 ? kCountsErr
                                                       : (int)threadIdx.x);
        mail_data[threadIdx.x] = (double)*cp;
        *cp = 0;
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

__global__ void PostCountsPairKernel(int* __restrict__ counts_a,
                                     double* mail_data_a, int* mail_flag_a,
                                     int mail_seq_a, int* __restrict__ counts_b,
                                     double* mail_data_b, int* mail_flag_b,
                                     int mail_seq_b, int n) {
    // one wave per chain
    int* counts = blockIdx.x ? counts_b : counts_a;
    double* mail_data = blockIdx.x ? mail_data_b : mail_data_a;
    if ((int)threadIdx.x < n) {
        int* cp = counts + ((int)threadIdx.x == n - 1 ? kCountsErr
                                                       : (int)threadIdx.x);
        const int c = *cp;
        mail_data[threadIdx.x] = (double)c;
        // a copy that survives the re-zeroing, for launches queued behind
        // this one that size themselves by a level count (the deferred small
        // index build of the ICP driver)
        counts[kCountsKeep + threadIdx.x] = c;
        *cp = 0;
    }
    MailboxPublish(blockIdx.x ? mail_flag_b : mail_flag_a,
                   blockIdx.x ? mail_seq_b : mail_seq_a);
}

int PostCountsPairAsync(int* counts_a, double* mail_data_a, int* mail_flag_a,
                        int mail_seq_a, int* counts_b, double* mail_data_b,
                        int* mail_flag_b, int mail_seq_b, int n,
                        hipStream_t s) {
    O3DMI_REQUIRE(n >= 1 && n <= 32, "too many levels");
    hipLaunchKernelGGL(PostCountsPairKernel, dim3(2), dim3(64), 0, s, counts_a,
                       mail_data_a, mail_flag_a, mail_seq_a, counts_b,
                       mail_data_b, mail_flag_b, mail_seq_b, n);
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

int VdsPairAsync(const VdsLevelJob* jobs, int n_jobs, int dtype,
                 std::vector<void*>& scratch, hipStream_t s, bool* posted) {
    O3DMI_REQUIRE(jobs && (n_jobs == 1 || n_jobs == 2), "bad job count");
    if (posted) *posted = false;
    bool tiled = true;
    for (int q = 0; q < n_jobs; ++q)
        tiled = tiled && jobs[q].n_max > 0 && jobs[q].n_max <= kTiledMaxPoints;
    if (tiled)
        return dtype == O3DMI_F64
                       ? VdsTiledImpl<double>(jobs, n_jobs, s, posted)
                       : VdsTiledImpl<float>(jobs, n_jobs, s, posted);
    for (int q = 0; q < n_jobs; ++q) {
        const VdsLevelJob& J = jobs[q];
        int st;
        if (J.n_max > 0 && J.n_max <= kTiledMaxPoints)
            st = dtype == O3DMI_F64 ? VdsTiledImpl<double>(&J, 1, s)
                                    : VdsTiledImpl<float>(&J, 1, s);
        else if (dtype == O3DMI_F64)
            st = VdsAsyncImpl<double>((const double*)J.pos,
                                      (const double*)J.attr, J.n_max, J.n_dev,
                                      J.voxel_size, (double*)J.out_pos,
                                      (double*)J.out_attr, J.m_dev, J.err_dev,
                                      scratch, s, J.chain);
        else
            st = VdsAsyncImpl<float>((const float*)J.pos, (const float*)J.attr,
                                     J.n_max, J.n_dev, J.voxel_size,
                                     (float*)J.out_pos, (float*)J.out_attr,
                                     J.m_dev, J.err_dev, scratch, s, J.chain);
        if (st) return st;
    }
    return O3DMI_OK;
}

int VdsAsync(const void* pos, const void* attr, int64_t n_max, const int* n_dev,
             int dtype, double voxel_size, void* out_pos, void* out_attr,
             int* m_dev, int* err_dev, std::vector<void*>& scratch,
             hipStream_t s, int chain, double next_voxel_size,
             bool from_previous) {
    VdsLevelJob J;
    J.pos = pos;
    J.attr = attr;
    J.n_max = n_max;
    J.n_dev = n_dev;
    J.voxel_size = voxel_size;
    J.out_pos = out_pos;
    J.out_attr = out_attr;
    J.m_dev = m_dev;
    J.err_dev = err_dev;
    J.chain = chain;
    J.next_voxel_size = next_voxel_size;
    J.from_previous = from_previous;
    return VdsPairAsync(&J, 1, dtype, scratch, s);
}

// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadPointcloud() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &VdsInitKernel)) == hipSuccess
                   ? 0
                   : 1;
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
