// Side-stream kernels of the sliced block touch (sliced_path.h): block-ownership
// sharding with the touch split over the ranks and one all-gather of candidate
// keys per chunk of frames (SURVEY 8(e) scheme A; the loop being sliced is
// DepthTouchCPU, t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201).

#include "sliced_path.h"
#include "touch_device.h"

namespace o3dmi {
namespace {

constexpr int kTileKeys = 2048;  // LDS key set of one 16 x 16 ray tile

// key -> bits |= b in group g's table. A new key claims a slot (one CAS) and
// joins the group's list; `claimed` tells the caller (the receiver inserts
// the key into the block hash exactly then). Returns the slot, or -1 when the
// table is full (flag set).
__device__ __forceinline__ int Accumulate(const GroupTables& t, int g,
                                          unsigned long long key, unsigned b,
                                          bool& claimed) {
    claimed = false;
    unsigned long long* keys = t.keys + (size_t)g * (t.mask + 1u);
    unsigned* bits = t.bits + (size_t)g * (t.mask + 1u);
    unsigned h = HashKey(key) & t.mask;
    for (unsigned step = 0; step <= t.mask; ++step) {
        unsigned long long cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptyKey) {
            cur = atomicCAS(&keys[h], kEmptyKey, key);
            if (cur == kEmptyKey) {
                const int idx = atomicAdd(&t.count[g], 1);
                if (idx >= t.list_cap) {
                    // more distinct blocks than the table is sized for: the
                    // slot stays claimed (cleaned with the table), the flag
                    // sends the chunk through the fallback
                    atomicOr(t.flags, kSliceFlagTable);
                } else {
                    t.list[(size_t)g * t.list_cap + idx] = h;
                }
                claimed = idx < t.list_cap;
                atomicOr(&bits[h], b);
                return (int)h;
            }
        }
        if (cur == key) {
            atomicOr(&bits[h], b);
            return (int)h;
        }
        h = (h + 1) & t.mask;
    }
    atomicOr(t.flags, kSliceFlagTable);
    return -1;
}

struct TouchSliceArgs {
    TouchParams p;  // p.cam.e is taken per frame
    const SliceFrame* frames;
    int f0, n;
    int frames_per_group;
    int tile_begin, tiles_per_frame;  // this rank's band of tiles
    GroupTables tables;
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
                    atomicOr(a.tables.flags, kSliceFlagKeyRange);
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
    const int g = fi / a.frames_per_group;
    const unsigned bit = 1u << (fi - g * a.frames_per_group);
    for (int e = threadIdx.x; e < kTileKeys; e += blockDim.x) {
        const unsigned long long k = tile_keys[e];
        if (k == kEmptyKey) continue;
        bool claimed;
        (void)Accumulate(a.tables, g, k, bit, claimed);
    }
}

// One workgroup per group: list -> records, slots back to empty.
__global__ void __launch_bounds__(256)
PackSliceKernel(GroupTables t, void* segment, int capacity) {
    __shared__ int s_n;
    const int g = (int)blockIdx.x;
    SliceHeader* hd = (SliceHeader*)segment;
    SliceRecord* rec = (SliceRecord*)((char*)segment + sizeof(SliceHeader)) +
                       (size_t)g * capacity;
    if (threadIdx.x == 0) s_n = t.count[g];
    __syncthreads();
    const int n_claimed = s_n;
    const int n = n_claimed < t.list_cap ? n_claimed : t.list_cap;
    unsigned long long* keys = t.keys + (size_t)g * (t.mask + 1u);
    unsigned* bits = t.bits + (size_t)g * (t.mask + 1u);
    const bool table_full = n_claimed > t.list_cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned h = t.list[(size_t)g * t.list_cap + i];
        if (i < capacity) {
            SliceRecord r;
            r.key = keys[h];
            r.bits = bits[h];
            r.pad = 0;
            rec[i] = r;
        }
        keys[h] = kEmptyKey;
        bits[h] = 0;
    }
    if (table_full) {
        // slots claimed beyond the list are not listed: sweep the table
        for (unsigned h = threadIdx.x; h <= t.mask; h += blockDim.x) {
            keys[h] = kEmptyKey;
            bits[h] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // a count beyond the capacity tells every rank that this slice did
        // not fit
        hd->count[g] = table_full ? capacity + 1 : n;
        t.count[g] = 0;
        if (g == 0) {
            hd->flags = *t.flags;
            hd->capacity = capacity;
            *t.flags = 0;
        }
    }
}

struct ApplySliceArgs {
    HashView hv;
    const char* gathered;
    int64_t segment_bytes;
    int world, capacity;
    GroupTables tables;
    int overflow_stamp;
};

// One workgroup per (sending rank, group).
__global__ void __launch_bounds__(256)
ApplySliceKernel(ApplySliceArgs a) {
    const int j = (int)blockIdx.x / kChunkGroups;
    const int g = (int)blockIdx.x - j * kChunkGroups;
    const char* seg = a.gathered + (size_t)j * a.segment_bytes;
    const SliceHeader* hd = (const SliceHeader*)seg;
    const SliceRecord* rec =
            (const SliceRecord*)(seg + sizeof(SliceHeader)) +
            (size_t)g * a.capacity;
    int n = hd->count[g];
    if (n > a.capacity || (g == 0 && (hd->flags & kSliceFlagTable))) {
        // the sender's slice did not fit its segment / its tables: flagged in
        // the receiver's status (BuildReadyKernel), the host redoes the chunk
        if (threadIdx.x == 0) atomicOr(a.tables.flags, kSliceFlagTable);
        if (n > a.capacity) n = a.capacity;
    }
    if (g == 0 && threadIdx.x == 0 && (hd->flags & kSliceFlagKeyRange))
        atomicOr(&a.hv.counters[1], kErrKeyRange);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const SliceRecord r = rec[i];
        if (!a.hv.Owns(r.key)) continue;
        bool claimed;
        const int ts = Accumulate(a.tables, g, r.key, r.bits, claimed);
        if (ts >= 0 && claimed) {
            const int x = (int)((r.key >> 42) & 0x1FFFFFull) - kKeyBias;
            const int y = (int)((r.key >> 21) & 0x1FFFFFull) - kKeyBias;
            const int z = (int)(r.key & 0x1FFFFFull) - kKeyBias;
            unsigned hslot = 0;
            InsertKey<true>(a.hv, x, y, z, hslot, a.overflow_stamp);
            a.tables.hslot[(size_t)g * (a.tables.mask + 1u) + (unsigned)ts] =
                    hslot;
        }
    }
}

struct BuildReadyArgs {
    HashView hv;
    GroupTables tables;
    ReadyEntry* ready;
    int ready_cap;
    int* ready_count;
    int* status_host;
    int stamp;
};

// One workgroup per group, after ApplySliceKernel has completed (kernel
// boundary: every buffer index is in place).
__global__ void __launch_bounds__(256)
BuildReadyKernel(BuildReadyArgs a) {
    __shared__ int s_n;
    const int g = (int)blockIdx.x;
    const GroupTables& t = a.tables;
    if (threadIdx.x == 0) s_n = t.count[g];
    __syncthreads();
    const int n_claimed = s_n;
    int n = n_claimed < t.list_cap ? n_claimed : t.list_cap;
    const bool table_full = n_claimed > t.list_cap;
    unsigned long long* keys = t.keys + (size_t)g * (t.mask + 1u);
    unsigned* bits = t.bits + (size_t)g * (t.mask + 1u);
    const unsigned* hslot = t.hslot + (size_t)g * (t.mask + 1u);
    // a chunk that ran out of buffer indices (or comes after one that did) is
    // dropped as a whole: empty lists; the host reserves and applies it again
    // (flags: set by ApplySliceKernel, cleared by the host before it)
    const bool flagged = (*t.flags & kSliceFlagTable) != 0;
    const bool dropped = a.hv.counters[3] != 0 || table_full || flagged;
    ReadyEntry* out = a.ready + (size_t)g * a.ready_cap;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned h = t.list[(size_t)g * t.list_cap + i];
        if (!dropped && i < a.ready_cap) {
            ReadyEntry re;
            re.key = keys[h];
            re.block_idx = a.hv.slot_vals[hslot[h]];
            re.bits = bits[h];
            out[i] = re;
        }
        keys[h] = kEmptyKey;
        bits[h] = 0;
    }
    if (table_full)
        for (unsigned h = threadIdx.x; h <= t.mask; h += blockDim.x) {
            keys[h] = kEmptyKey;
            bits[h] = 0;
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n > a.ready_cap) n = a.ready_cap;
        a.ready_count[g] = dropped ? 0 : n;
        t.count[g] = 0;
        if (g == 0 && a.status_host) {
            const int top = a.hv.counters[0];
            a.status_host[0] = top < a.hv.capacity ? top : a.hv.capacity;
            // table overflow is reported as a (negative) pseudo stamp
            a.status_host[1] = a.hv.counters[3] != 0
                                       ? a.hv.counters[3]
                                       : ((flagged || table_full) ? -1 : 0);
            a.status_host[2] = n;
            __hip_atomic_store(&a.status_host[3], a.stamp, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void InitTablesKernel(unsigned long long* keys, unsigned* bits,
                                 int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = kEmptyKey;
        bits[i] = 0;
    }
}

}  // namespace

int AllocGroupTables(GroupTables* t, int slots, bool receiver, hipStream_t s) {
    *t = GroupTables{};
    const size_t n = (size_t)kChunkGroups * (size_t)slots;
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->keys, sizeof(unsigned long long) * n));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->bits, sizeof(unsigned) * n));
    if (receiver)
        O3DMI_HIP_CHECK(hipMalloc((void**)&t->hslot, sizeof(unsigned) * n));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->list, sizeof(unsigned) * n / 2));
    O3DMI_HIP_CHECK(hipMalloc((void**)&t->count,
                              sizeof(int) * (kChunkGroups + 1)));
    t->flags = t->count + kChunkGroups;
    t->mask = (unsigned)slots - 1u;
    t->list_cap = slots / 2;
    O3DMI_HIP_CHECK(hipMemsetAsync(t->count, 0,
                                   sizeof(int) * (kChunkGroups + 1), s));
    hipLaunchKernelGGL(InitTablesKernel, dim3(GridFor((int64_t)n, kBlock)),
                       dim3(kBlock), 0, s, t->keys, t->bits, (int64_t)n);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

void FreeGroupTables(GroupTables* t) {
    (void)hipFree(t->keys);
    (void)hipFree(t->bits);
    (void)hipFree(t->hslot);
    (void)hipFree(t->list);
    (void)hipFree(t->count);
    *t = GroupTables{};
}

int LaunchTouchSlice(const TouchParams& shared, const SliceFrame* frames_dev,
                     int f0, int n, int frames_per_group, int slice_rank,
                     int slice_world, const GroupTables& tables,
                     hipStream_t s) {
    O3DMI_REQUIRE(n >= 1 && frames_per_group >= 1 &&
                          n <= kChunkGroups * frames_per_group,
                  "touch slice: bad chunk");
    O3DMI_REQUIRE(slice_world >= 1 && slice_rank >= 0 &&
                          slice_rank < slice_world,
                  "touch slice: bad rank / world");
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
    a.frames_per_group = frames_per_group;
    a.tile_begin = begin;
    a.tiles_per_frame = end - begin;
    a.tables = tables;
    hipLaunchKernelGGL(TouchSliceKernel, dim3((unsigned)(n * (end - begin))),
                       dim3(256), 0, s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchPackSlice(const GroupTables& tables, void* segment_dev, int capacity,
                    hipStream_t s) {
    hipLaunchKernelGGL(PackSliceKernel, dim3(kChunkGroups), dim3(256), 0, s,
                       tables, segment_dev, capacity);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchApplySlice(o3dmi_hash* bh, const void* gathered_dev, int world,
                     int capacity, const GroupTables& tables,
                     int overflow_stamp, hipStream_t s) {
    ApplySliceArgs a;
    a.hv = bh->view;
    a.gathered = (const char*)gathered_dev;
    a.segment_bytes = SliceSegmentBytes(capacity);
    a.world = world;
    a.capacity = capacity;
    a.tables = tables;
    a.overflow_stamp = overflow_stamp;
    hipLaunchKernelGGL(ApplySliceKernel, dim3((unsigned)(world * kChunkGroups)),
                       dim3(256), 0, s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int LaunchBuildReady(o3dmi_hash* bh, const GroupTables& tables,
                     ReadyEntry* ready, int ready_cap, int* ready_count,
                     int* status_host, int stamp, hipStream_t s) {
    BuildReadyArgs a;
    a.hv = bh->view;
    a.tables = tables;
    a.ready = ready;
    a.ready_cap = ready_cap;
    a.ready_count = ready_count;
    a.status_host = status_host;
    a.stamp = stamp;
    hipLaunchKernelGGL(BuildReadyKernel, dim3(kChunkGroups), dim3(256), 0, s, a);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace o3dmi
