// Side-stream kernels of the sliced block touch (sliced_path.h): block-ownership
// sharding with the touch split over the ranks and one all-gather of candidate
// keys per chunk of frames (SURVEY 8(e) scheme A; the loop being sliced is
// DepthTouchCPU, t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201).

#include "sliced_path.h"
#include "touch_device.h"

namespace o3dmi {
namespace {

constexpr int kTileKeys = 2048;  // LDS key set of one 16 x 16 ray tile

// key -> slot of the chunk table. A new key claims a slot (one CAS) and joins
// the list; `claimed` tells the caller (the receiver inserts the key into the
// block hash exactly then). Returns the slot, or -1 when the table is full
// (flag set). The caller ORs its frame bits into bits[slot].
__device__ __forceinline__ int ClaimChunkSlot(const ChunkTable& t,
                                              unsigned long long key,
                                              bool& claimed) {
    claimed = false;
    unsigned h = HashKey(key) & t.mask;
    for (unsigned step = 0; step <= t.mask; ++step) {
        unsigned long long cur = __hip_atomic_load(
                &t.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptyKey) {
            cur = atomicCAS(&t.keys[h], kEmptyKey, key);
            if (cur == kEmptyKey) {
                const int idx = atomicAdd(t.count, 1);
                if (idx >= t.list_cap) {
                    // more distinct blocks than the table is sized for: the
                    // slot stays claimed (cleaned with the table), the flag
                    // sends the chunk through the fallback
                    atomicOr(t.flags, kSliceFlagTable);
                } else {
                    t.list[idx] = h;
                    claimed = true;
                }
                return (int)h;
            }
        }
        if (cur == key) return (int)h;
        h = (h + 1) & t.mask;
    }
    atomicOr(t.flags, kSliceFlagTable);
    return -1;
}

struct TouchSliceArgs {
    TouchParams p;  // p.cam.e is taken per frame
    const SliceFrame* frames;
    int f0, n;
    int tile_begin, tiles_per_frame;  // this rank's band of tiles
    ChunkTable table;
};

// One workgroup = one 16 x 16 tile of rays of one frame.
__global__ void __launch_bounds__(256)
TouchSliceKernel(TouchSliceArgs a) {
    __shared__ unsigned long long tile_keys[kTileKeys];
    const int fi = (int)blockIdx.x / a.tiles_per_frame;
    const int tile = a.tile_begin + (int)blockIdx.x - fi * a.tiles_per_frame;
    const SliceFrame& fr = a.frames[a.f0 + fi];
    TouchParams p = a.p;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) p.cam.e[i][j] = fr.pose[i][j];
    for (int e = threadIdx.x; e < kTileKeys; e += blockDim.x)
        tile_keys[e] = kEmptyKey;
    __syncthreads();
    const int tiles_x = (p.cols_strided + 15) >> 4;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int rx = tx * 16 + (threadIdx.x & 15);
    const int ry = ty * 16 + (threadIdx.x >> 4);
    if (rx < p.cols_strided && ry < p.rows_strided) {
        int xb[4], yb[4], zb[4];
        if (RayCandidates(p, fr.depth, ry * p.cols_strided + rx, xb, yb, zb)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s > 0 && xb[s] == xb[s - 1] && yb[s] == yb[s - 1] &&
                    zb[s] == zb[s - 1])
                    continue;
                if (!KeyInRange(xb[s], yb[s], zb[s])) {
                    atomicOr(a.table.flags, kSliceFlagKeyRange);
                    continue;
                }
                const unsigned long long k = PackKey(xb[s], yb[s], zb[s]);
                unsigned h = HashKey(k) & (kTileKeys - 1);
                while (true) {  // <= 1024 keys in 2048 slots: terminates
                    unsigned long long cur = tile_keys[h];
                    if (cur == k) break;
                    if (cur == kEmptyKey) {
                        cur = atomicCAS(&tile_keys[h], kEmptyKey, k);
                        if (cur == kEmptyKey || cur == k) break;
                    }
                    h = (h + 1) & (kTileKeys - 1);
                }
            }
        }
    }
    __syncthreads();
    const int word = fi >> 5;
    const unsigned bit = 1u << (fi & 31);
    for (int e = threadIdx.x; e < kTileKeys; e += blockDim.x) {
        const unsigned long long k = tile_keys[e];
        if (k == kEmptyKey) continue;
        bool claimed;
        const int slot = ClaimChunkSlot(a.table, k, claimed);
        if (slot >= 0)
            atomicOr(&a.table.bits[(size_t)slot * kChunkWords + word], bit);
    }
}

// list -> records, slots back to empty (grid-stride over the list).
__global__ void __launch_bounds__(256)
PackSliceKernel(ChunkTable t, void* segment, int capacity) {
    SliceHeader* hd = (SliceHeader*)segment;
    SliceRecord* rec = (SliceRecord*)((char*)segment + sizeof(SliceHeader));
    const int n_claimed = *t.count;  // stable: the touch launch has finished
    const int n = n_claimed < t.list_cap ? n_claimed : t.list_cap;
    const bool table_full = n_claimed > t.list_cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += gridDim.x * blockDim.x) {
        const unsigned h = t.list[i];
        if (i < capacity) {
            SliceRecord r;
            r.key = t.keys[h];
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w)
                r.bits[w] = t.bits[(size_t)h * kChunkWords + w];
            r.pad[0] = r.pad[1] = 0;
            rec[i] = r;
        }
        t.keys[h] = kEmptyKey;
#pragma unroll
        for (int w = 0; w < kChunkWords; ++w)
            t.bits[(size_t)h * kChunkWords + w] = 0;
    }
    if (table_full)  // slots claimed beyond the list are not listed: sweep
        for (unsigned h = blockIdx.x * blockDim.x + threadIdx.x; h <= t.mask;
             h += gridDim.x * blockDim.x) {
            t.keys[h] = kEmptyKey;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w)
                t.bits[(size_t)h * kChunkWords + w] = 0;
        }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // a count beyond the capacity tells every rank that the slice did not
        // fit (count / flags are reset by the host before the next touch)
        hd->count = table_full ? capacity + 1 : n;
        hd->flags = *t.flags;
        hd->capacity = capacity;
    }
}

struct ApplySliceArgs {
    HashView hv;
    const char* gathered;
    int64_t segment_bytes;
    int world, capacity;
    ChunkTable table;
    int overflow_stamp;
};

// blockIdx.y = sending rank; grid-stride over its records.
__global__ void __launch_bounds__(256)
ApplySliceKernel(ApplySliceArgs a) {
    const int j = (int)blockIdx.y;
    const char* seg = a.gathered + (size_t)j * a.segment_bytes;
    const SliceHeader* hd = (const SliceHeader*)seg;
    const SliceRecord* rec = (const SliceRecord*)(seg + sizeof(SliceHeader));
    int n = hd->count;
    if (hd->flags & kSliceFlagAbort) {
        // the sender is leaving the call with an error: every rank drops the
        // chunk and leaves with it (status -3, BuildChunkKernel)
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(a.table.flags, kSliceFlagPeerAbort);
        n = 0;
    }
    if (n > a.capacity || (hd->flags & kSliceFlagTable)) {
        // the sender's slice did not fit its segment / its table: flagged in
        // the receiver's status (BuildChunkKernel), the host redoes the chunk
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicOr(a.table.flags, kSliceFlagSender);
        if (n > a.capacity) n = a.capacity;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 &&
        (hd->flags & kSliceFlagKeyRange))
        atomicOr(&a.hv.counters[1], kErrKeyRange);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += gridDim.x * blockDim.x) {
        const SliceRecord r = rec[i];
        if (!a.hv.Owns(r.key)) continue;
        bool claimed;
        const int ts = ClaimChunkSlot(a.table, r.key, claimed);
        if (ts < 0) continue;
#pragma unroll
        for (int w = 0; w < kChunkWords; ++w)
            if (r.bits[w])
                atomicOr(&a.table.bits[(size_t)ts * kChunkWords + w],
                         r.bits[w]);
        if (claimed) {
            const int x = (int)((r.key >> 42) & 0x1FFFFFull) - kKeyBias;
            const int y = (int)((r.key >> 21) & 0x1FFFFFull) - kKeyBias;
            const int z = (int)(r.key & 0x1FFFFFull) - kKeyBias;
            unsigned hslot = 0;
            InsertKey<true>(a.hv, x, y, z, hslot, a.overflow_stamp);
            a.table.hslot[ts] = hslot;
        }
    }
}

struct BuildChunkArgs {
    HashView hv;
    ChunkTable table;
    ChunkEntry* entries;
    int entries_cap;
    int* entries_count;
    int* status_host;
    int stamp;
    int longest_first;  // counting sort by frame count (arrival order: 0)
};

// After ApplySliceKernel has completed (kernel boundary: every buffer index is
// in place). One workgroup: its tail (count, status) follows every entry.
__global__ void __launch_bounds__(256)
BuildChunkKernel(BuildChunkArgs a) {
    const ChunkTable& t = a.table;
    const int n_claimed = *t.count;
    int n = n_claimed < t.list_cap ? n_claimed : t.list_cap;
    const bool table_full = n_claimed > t.list_cap;
    // (flags: set by ApplySliceKernel, cleared by the host before it)
    const int fl = *t.flags;
    const bool flagged = (fl & (kSliceFlagTable | kSliceFlagSender |
                                kSliceFlagPeerAbort)) != 0;
    // a chunk that ran out of buffer indices (or whose records did not fit)
    // is dropped as a whole: empty list; the host makes room and applies it
    // again
    const bool dropped = a.hv.counters[3] != 0 || table_full || flagged;
    // Longest first: the chunk launch takes one workgroup per (entry, part)
    // in list order and lasts as long as its slowest SIMD, and the entries'
    // work is very uneven (a rank's share at 8 ranks: 27 frames on average,
    // 192 for the blocks every frame of the chunk sees). With the entries in
    // descending order of their frame count the dispatcher's in-order issue
    // is longest-processing-time-first scheduling: the long chains start at
    // time 0 and the short items fill the machine behind them. A counting
    // sort over the 1..kChunkFrames possible counts; the order among equal
    // counts is arrival order (any order gives the same grid: entries are
    // distinct blocks).
    __shared__ int sort_pos[kChunkFrames + 2];
    for (int c = threadIdx.x; c < kChunkFrames + 2; c += blockDim.x)
        sort_pos[c] = 0;
    __syncthreads();
    if (!dropped)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned h = t.list[i];
            int c = 0;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w)
                c += __popc(t.bits[(size_t)h * kChunkWords + w]);
            atomicAdd(&sort_pos[a.longest_first ? c : 0], 1);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;  // descending counts -> ascending positions
        for (int c = kChunkFrames; c >= 0; --c) {
            const int k = sort_pos[c];
            sort_pos[c] = run;
            run += k;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned h = t.list[i];
        if (!dropped) {
            ChunkEntry ce;
            ce.key = t.keys[h];
            ce.block_idx = a.hv.slot_vals[t.hslot[h]];
            ce.pad = 0;
            int c = 0;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w) {
                ce.bits[w] = t.bits[(size_t)h * kChunkWords + w];
                c += __popc(ce.bits[w]);
            }
            const int pos = atomicAdd(&sort_pos[a.longest_first ? c : 0], 1);
            if (pos < a.entries_cap) a.entries[pos] = ce;
        }
        t.keys[h] = kEmptyKey;
#pragma unroll
        for (int w = 0; w < kChunkWords; ++w)
            t.bits[(size_t)h * kChunkWords + w] = 0;
    }
    if (table_full)
        for (unsigned h = threadIdx.x; h <= t.mask; h += blockDim.x) {
            t.keys[h] = kEmptyKey;
#pragma unroll
            for (int w = 0; w < kChunkWords; ++w)
                t.bits[(size_t)h * kChunkWords + w] = 0;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n > a.entries_cap) n = a.entries_cap;
        *a.entries_count = dropped ? 0 : n;
        if (a.status_host) {
            const int top = a.hv.counters[0];
            a.status_host[0] = top < a.hv.capacity ? top : a.hv.capacity;
            a.status_host[1] =
                    (fl & kSliceFlagPeerAbort)
                            ? -3
                            : (a.hv.counters[3] != 0
                                       ? a.hv.counters[3]
                                       : ((fl & kSliceFlagSender)
                                                  ? -1
                                                  : ((flagged || table_full)
                                                             ? -2
                                                             : 0)));
            a.status_host[2] = n;
            __hip_atomic_store(&a.status_host[3], a.stamp, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void InitTableKernel(unsigned long long* keys, unsigned* bits,
                                int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = kEmptyKey;
#pragma unroll
        for (int w = 0; w < kChunkWords; ++w) bits[i * kChunkWords + w] = 0;
    }
}

}  // namespace

int AllocChunkTable(ChunkTable* t, int slots, bool receiver, hipStream_t s) {
    *t = ChunkTable{};
    const size_t n = (size_t)slots;
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->keys, sizeof(unsigned long long) * n));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->bits,
                              sizeof(unsigned) * n * kChunkWords));
    if (receiver)
        O3DMI_HIP_CHECK(hipMalloc((void**)&t->hslot, sizeof(unsigned) * n));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->list, sizeof(unsigned) * n / 2));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->count, sizeof(int) * 2));
    t->flags = t->count + 1;
    t->mask = (unsigned)slots - 1u;
    t->list_cap = slots / 2;
    O3DMI_HIP_CHECK(hipMemsetAsync(t->count, 0, sizeof(int) * 2, s));
    hipLaunchKernelGGL(InitTableKernel, dim3(GridFor((int64_t)n, kBlock)),
                       dim3(kBlock), 0, s, t->keys, t->bits, (int64_t)n);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

void FreeChunkTable(ChunkTable* t) {
    (void)hipFree(t->keys);
    (void)hipFree(t->bits);
    (void)hipFree(t->hslot);
    (void)hipFree(t->list);
    (void)hipFree(t->count);
    *t = ChunkTable{};
}

int LaunchTouchSlice(const TouchParams& shared, const SliceFrame* frames_dev,
                     int f0, int n, int slice_rank, int slice_world,
                     const ChunkTable& table, hipStream_t s) {
    O3DMI_REQUIRE(n >= 1 && n <= kChunkFrames, "touch slice: bad chunk");
    O3DMI_REQUIRE(slice_world >= 1 && slice_rank >= 0 &&
                          slice_rank < slice_world,
                  "touch slice: bad rank / world");
    // the table's count / flags of the previous chunk (its pack launch is
    // behind us on this stream)
    O3DMI_HIP_CHECK(hipMemsetAsync(table.count, 0, sizeof(int) * 2, s));
    const int tiles = ((shared.cols_strided + 15) / 16) *
                      ((shared.rows_strided + 15) / 16);
    // contiguous band of tiles (row-major): neighbouring tiles see the same
    // blocks, so a band's distinct keys are ~1/world of the frame's
    const int begin = (int)((int64_t)tiles * slice_rank / slice_world);
    const int end = (int)((int64_t)tiles * (slice_rank + 1) / slice_world);
    if (end <= begin) return O3DMI_OK;  // more ranks than tiles
    TouchSliceArgs a;
    a.p = shared;
    a.frames = frames_dev;
    a.f0 = f0;
    a.n = n;
    a.tile_begin = begin;
    a.tiles_per_frame = end - begin;
    a.table = table;
    hipLaunchKernelGGL(TouchSliceKernel, dim3((unsigned)(n * (end - begin))),
                       dim3(256), 0, s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchPackSlice(const ChunkTable& table, void* segment_dev, int capacity,
                    hipStream_t s) {
    hipLaunchKernelGGL(PackSliceKernel, dim3(16), dim3(256), 0, s, table,
                       segment_dev, capacity);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchApplySlice(o3dmi_hash* bh, const void* gathered_dev, int world,
                     int capacity, const ChunkTable& table, int overflow_stamp,
                     hipStream_t s) {
    O3DMI_HIP_CHECK(hipMemsetAsync(table.count, 0, sizeof(int) * 2, s));
    ApplySliceArgs a;
    a.hv = bh->view;
    a.gathered = (const char*)gathered_dev;
    a.segment_bytes = SliceSegmentBytes(capacity);
    a.world = world;
    a.capacity = capacity;
    a.table = table;
    a.overflow_stamp = overflow_stamp;
    hipLaunchKernelGGL(ApplySliceKernel, dim3(8, (unsigned)world), dim3(256), 0,
                       s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchBuildChunk(o3dmi_hash* bh, const ChunkTable& table,
                     ChunkEntry* entries, int entries_cap, int* entries_count,
                     int* status_host, int stamp, hipStream_t s) {
    BuildChunkArgs a;
    a.hv = bh->view;
    a.table = table;
    a.entries = entries;
    a.entries_cap = entries_cap;
    a.entries_count = entries_count;
    a.status_host = status_host;
    a.stamp = stamp;
    a.longest_first = 1;
    hipLaunchKernelGGL(BuildChunkKernel, dim3(1), dim3(256), 0, s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace o3dmi
