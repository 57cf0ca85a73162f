// Shared definitions of the MI355X backend: status/error plumbing, the float32
// camera model (Open3D's TransformIndexer, t/geometry/kernel/
// GeometryIndexer.h:25-144) as a POD passed by value to kernels, and the
// block-key packing used by the spatial hash.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "o3d_mi355x.h"

namespace o3dmi {

void SetLastError(const std::string& msg);

// Caching scratch allocator (pool.cpp). PoolFree must only be called once the
// work using the block has completed.
int PoolAlloc(void** out, size_t bytes);
void PoolFree(void* p);

// Row gather / scatter by index (rows.hip): dst[r] = src[idx[r]] and
// dst[idx[r]] = src[r] for rows of row_bytes bytes.
int GatherRows(const void* src, const int* indices_dev, int64_t n,
               int64_t row_bytes, void* dst, hipStream_t s);
int ScatterRows(const void* src, const int* indices_dev, int64_t n,
                int64_t row_bytes, void* dst, hipStream_t s);

#define O3DMI_HIP_CHECK(expr)                                              \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) {                                            \
            ::o3dmi::SetLastError(std::string(#expr) + ": " +              \
                                  hipGetErrorString(_e));                  \
            return O3DMI_ERR_HIP;                                          \
        }                                                                  \
    } while (0)

#define O3DMI_REQUIRE(cond, msg)                                           \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ::o3dmi::SetLastError(msg);                                    \
            return O3DMI_ERR_INVALID_ARG;                                  \
        }                                                                  \
    } while (0)

// Fixed launch geometry for streaming kernels: enough workgroups to cover the
// 256 CUs several times, grid-stride over the rest.
constexpr int kCUs = 256;
constexpr int kBlock = 256;

inline int GridFor(int64_t n, int per_block, int max_blocks = kCUs * 8) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// Float copies of intrinsics / extrinsics, exactly as TransformIndexer keeps
// them (GeometryIndexer.h:46-58,136-143).
struct Camera {
    float e[3][4];
    float fx, fy, cx, cy;
    float scale;

    __host__ static Camera Make(const double* K, const double* T,
                                float scale = 1.0f) {
        Camera c;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) c.e[i][j] = (float)T[i * 4 + j];
        c.fx = (float)K[0];
        c.fy = (float)K[4];
        c.cx = (float)K[2];
        c.cy = (float)K[5];
        c.scale = scale;
        return c;
    }

    // GeometryIndexer.h:62-78 (evaluation order preserved; the library is
    // compiled with -ffp-contract=off).
    __device__ __forceinline__ void RigidTransform(float x, float y, float z,
                                                   float& xo, float& yo,
                                                   float& zo) const {
        x *= scale;
        y *= scale;
        z *= scale;
        xo = x * e[0][0] + y * e[0][1] + z * e[0][2] + e[0][3];
        yo = x * e[1][0] + y * e[1][1] + z * e[1][2] + e[1][3];
        zo = x * e[2][0] + y * e[2][1] + z * e[2][2] + e[2][3];
    }
    // GeometryIndexer.h:81-97
    __device__ __forceinline__ void Rotate(float x, float y, float z,
                                           float& xo, float& yo,
                                           float& zo) const {
        x *= scale;
        y *= scale;
        z *= scale;
        xo = x * e[0][0] + y * e[0][1] + z * e[0][2];
        yo = x * e[1][0] + y * e[1][1] + z * e[1][2];
        zo = x * e[2][0] + y * e[2][1] + z * e[2][2];
    }
    // GeometryIndexer.h:100-108
    __device__ __forceinline__ void Project(float x, float y, float z,
                                            float& u, float& v) const {
        float inv_z = 1.0f / z;
        u = fx * x * inv_z + cx;
        v = fy * y * inv_z + cy;
    }
    // GeometryIndexer.h:111-120
    __device__ __forceinline__ void Unproject(float u, float v, float d,
                                              float& x, float& y,
                                              float& z) const {
        x = (u - cx) * d / fx;
        y = (v - cy) * d / fy;
        z = d;
    }
};

// t/geometry/Utility.h:77-120
inline void InverseTransformation(const double* T, double* Tinv) {
    Tinv[0] = T[0];  Tinv[1] = T[4];  Tinv[2] = T[8];
    Tinv[4] = T[1];  Tinv[5] = T[5];  Tinv[6] = T[9];
    Tinv[8] = T[2];  Tinv[9] = T[6];  Tinv[10] = T[10];
    Tinv[3] = -(Tinv[0] * T[3] + Tinv[1] * T[7] + Tinv[2] * T[11]);
    Tinv[7] = -(Tinv[4] * T[3] + Tinv[5] * T[7] + Tinv[6] * T[11]);
    Tinv[11] = -(Tinv[8] * T[3] + Tinv[9] * T[7] + Tinv[10] * T[11]);
    Tinv[12] = 0; Tinv[13] = 0; Tinv[14] = 0; Tinv[15] = 1;
}

// TArrayIndexer::InBoundary for a {rows, cols} image (GeometryIndexer.h:294).
__device__ __forceinline__ bool InBoundary2D(float x, float y, int rows,
                                             int cols) {
    return y >= 0 && x >= 0 && y <= rows - 1.0f && x <= cols - 1.0f;
}

// ---- block keys ------------------------------------------------------------
// A block key (int32 x 3) is packed into one 64-bit word, 21 bits per
// coordinate biased by 2^20, so that slot ownership is decided by a single
// 64-bit CAS. |coord| < 2^20 blocks (134 km at 8 mm x 16) is enforced.
constexpr int kKeyBias = 1 << 20;
constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned long long kTombKey = 0xFFFFFFFFFFFFFFFEull;

__host__ __device__ __forceinline__ bool KeyInRange(int x, int y, int z) {
    return x >= -kKeyBias && x < kKeyBias && y >= -kKeyBias && y < kKeyBias &&
           z >= -kKeyBias && z < kKeyBias;
}
__host__ __device__ __forceinline__ unsigned long long PackKey(int x, int y,
                                                               int z) {
    return ((unsigned long long)(unsigned)(x + kKeyBias) << 42) |
           ((unsigned long long)(unsigned)(y + kKeyBias) << 21) |
           (unsigned long long)(unsigned)(z + kKeyBias);
}
__host__ __device__ __forceinline__ unsigned HashKey(unsigned long long k) {
    // 64-bit finaliser (splitmix64 constants); only the low bits are used.
    k ^= k >> 30;
    k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 27;
    k *= 0x94d049bb133111ebull;
    k ^= k >> 31;
    return (unsigned)k;
}

// Owner rank of a packed block key: a 64-bit finaliser independent of HashKey
// (murmur3 constants), upper half modulo the world size.
__host__ __device__ __forceinline__ int OwnerOf(unsigned long long k,
                                                int world) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (int)((unsigned)(k >> 32) % (unsigned)world);
}

constexpr int kErrKeyRange = 1;
constexpr int kErrCapacity = 2;
constexpr int kErrTouchStamp = 4;  // integrate role met a foreign touch word
constexpr int kErrProbe = 8;       // probe sequence wrapped (table full)

// Device view of the spatial hash, passed by value to kernels.
struct HashView {
    unsigned long long* slot_keys;  // [n_slots] packed key / empty / tombstone
    int* slot_vals;                 // [n_slots] buffer index of the slot
    // [2][n_slots] (touch stamp << kTouchBits) | frame bits. Two planes: consecutive
    // frame groups of the frame-stream path alternate planes, because the
    // front roles of group g+1 run in the same launch as the integrate role of
    // group g, which still reads group g's words (plane = group sequence & 1;
    // every other user takes plane 0).
    unsigned long long* slot_touch;
    int* heap;                      // [capacity] free buffer indices
    int* counters;                  // [0]=heap_top, [1]=error flags,
                                    // [2]=slots ever taken from the empty
                                    //     state (live + tombstones),
                                    // [3]=stamp of the first frame-stream
                                    //     group that ran out of buffer
                                    //     indices (0 = none; InsertKey)
    int* key_buffer;                // [capacity,3]
    unsigned mask;                  // n_slots - 1
    int capacity;
    // Block-ownership sharding (multi-GPU, SURVEY 8e scheme A): with
    // owner_world > 1 the touch kernels only activate / list the blocks whose
    // OwnerOf(key) equals owner_rank; every rank then holds a disjoint part of
    // the same grid, each part bit-identical to the single-GPU result.
    int owner_rank;
    int owner_world;

    // True when this rank integrates block key k.
    __device__ __forceinline__ bool Owns(unsigned long long k) const {
        return owner_world <= 1 || OwnerOf(k, owner_world) == owner_rank;
    }

    // Lookup; -1 when absent. The walk is bounded by the table size: with
    // tombstone reuse (ClaimSlot) and the rebuild behind Erase the table always
    // keeps empty slots, the bound only guards a corrupted table.
    __device__ __forceinline__ int Find(int x, int y, int z) const {
        if (!KeyInRange(x, y, z)) return -1;
        unsigned long long k = PackKey(x, y, z);
        unsigned h = HashKey(k) & mask;
        for (unsigned step = 0; step <= mask; ++step) {
            unsigned long long cur = slot_keys[h];
            if (cur == k) return slot_vals[h];
            if (cur == kEmptyKey) return -1;
            h = (h + 1) & mask;
        }
        return -1;
    }
};

// Insert-if-absent of packed key k into the slot table. Returns 1 when this
// thread created the entry, 0 when the key is (or concurrently became)
// present, -1 when the probe sequence wrapped (error flag set). `slot_out`
// receives the slot in the first two cases.
//
// A new key takes the FIRST TOMBSTONE of its probe sequence if there is one,
// else the first empty slot; the walk always continues to that empty slot
// first, because the key may live behind the tombstone. Every inserter of the
// same key therefore competes for the same slot and the CAS picks one winner;
// a thread that loses its target to a different key walks on from there. Only
// inserts may run concurrently (Erase is its own launch, as in the
// reference's backends), so a slot's state only moves empty / tombstone ->
// key during the walk and a stale read is always corrected by the CAS result.
__device__ __forceinline__ int ClaimSlot(const HashView& hv,
                                         unsigned long long k,
                                         unsigned& slot_out,
                                         bool report_wrap = true) {
    constexpr unsigned kNone = 0xFFFFFFFFu;
    unsigned h = HashKey(k) & hv.mask;
    unsigned tomb = kNone;
    unsigned walked = 0;  // slots seen since the walk (re)started
    // 3 x table size: a lost target re-walks part of the sequence
    for (unsigned long long step = 0; step <= 3ull * hv.mask + 2; ++step) {
        const unsigned long long cur = hv.slot_keys[h];
        if (cur == k) {
            slot_out = h;
            return 0;
        }
        bool lost = false;
        if (cur == kTombKey) {
            if (tomb == kNone) tomb = h;
        } else if (cur == kEmptyKey) {
            const unsigned target = tomb != kNone ? tomb : h;
            const unsigned long long expect =
                    tomb != kNone ? kTombKey : kEmptyKey;
            const unsigned long long old =
                    atomicCAS(&hv.slot_keys[target], expect, k);
            if (old == expect) {
                if (tomb == kNone) atomicAdd(&hv.counters[2], 1);
                slot_out = target;
                return 1;
            }
            if (old == k) {
                slot_out = target;
                return 0;
            }
            // lost `target` to another key: go on behind it
            h = target;
            lost = true;
        }
        h = (h + 1) & hv.mask;
        if (lost) {
            tomb = kNone;
            walked = 0;
        } else if (++walked > hv.mask && tomb != kNone) {
            // A whole cycle without an empty slot (a table crowded with
            // tombstones): the key is absent, its place is the first
            // tombstone of the sequence.
            const unsigned long long old =
                    atomicCAS(&hv.slot_keys[tomb], kTombKey, k);
            if (old == kTombKey || old == k) {
                slot_out = tomb;
                return old == kTombKey ? 1 : 0;
            }
            h = (tomb + 1) & hv.mask;
            tomb = kNone;
            walked = 0;
        }
    }
    if (report_wrap) atomicOr(&hv.counters[1], kErrProbe);
    return -1;
}

// "Am I the last of `total` workgroups to get here?" -- true in every thread
// of exactly one workgroup, after all the others have arrived. For hand-offs
// INSIDE a launch: the data an arriving workgroup leaves for the last one must
// be write-through (agent-scope atomic / sc1) stores or atomics; they are
// drained here (s_waitcnt vmcnt(0) in every wave) before the ticket is taken,
// and the last workgroup reads them with agent-scope loads. No release fence:
// on this part it writes back the XCD's whole L2. Tickets are two-level, one
// counter per class (index % 8) and one on top, so that no word sees more
// than total / 8 arrivals; tickets[0..8] must be zero beforehand and are zero
// again afterwards.
__device__ __forceinline__ bool LastArrival(int* tickets, int index,
                                            int total) {
    __shared__ int s_last_arrival;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int cls = index & 7;
        const int in_class = (total - cls + 7) >> 3;
        const int classes = total < 8 ? total : 8;
        int last = 0;
        if (__hip_atomic_fetch_add(&tickets[1 + cls], 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) == in_class - 1) {
            // this class is complete: its counter back to zero, then up
            __hip_atomic_store(&tickets[1 + cls], 0, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(&tickets[0], 1, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) ==
                   classes - 1;
            if (last)
                __hip_atomic_store(&tickets[0], 0, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last_arrival = last;
    }
    __syncthreads();
    return s_last_arrival != 0;
}

// Marks `slot` as touched by frame `bit` of the frame group `stamp`. The word
// holds (stamp << kTouchBits) | one bit per frame of the group; a word carrying
// an older stamp is stale and is replaced. Returns true for exactly one caller per
// (slot, stamp): the one that moved the word to this stamp.
constexpr int kTouchBits = 16;  // frames per group (stream_path.h kMaxGroup)
__device__ __forceinline__ unsigned long long* TouchWord(const HashView& hv,
                                                         unsigned slot,
                                                         int plane) {
    return hv.slot_touch + ((size_t)plane * ((size_t)hv.mask + 1) + slot);
}
__device__ __forceinline__ bool TouchSlot(const HashView& hv, unsigned slot,
                                          unsigned long long stamp, int bit,
                                          int plane = 0) {
    unsigned long long* w = TouchWord(hv, slot, plane);
    unsigned long long cur = *w;  // possibly stale; the CAS corrects it
    while (true) {
        const bool fresh = (cur >> kTouchBits) != stamp;
        const unsigned long long want =
                fresh ? ((stamp << kTouchBits) | (1ull << bit))
                      : (cur | (1ull << bit));
        if (want == cur) return false;
        const unsigned long long old = atomicCAS(w, cur, want);
        if (old == cur) return fresh;
        cur = old;
    }
}


}  // namespace o3dmi

// The C struct behind o3dmi_hash_t.
struct o3dmi_hash {
    o3dmi::HashView view;
    int64_t capacity = 0;
    int64_t n_slots = 0;
    int n_values = 0;
    int64_t value_dsizes[8] = {0};
    void* value_buffers[8] = {nullptr};
    int* scratch_count = nullptr;  // device int for compaction kernels
    int* counters_host = nullptr;  // pinned mirror of view.counters (4 ints):
                                   // size queries wait for one async copy
};

namespace o3dmi {
// o3dmi_hash_clear and one more device counter zeroed by the same launch
// (block_hash.hip; the touch kernels' output count).
int ClearHashAndCounter(o3dmi_hash* h, int* counter_dev, hipStream_t s);
}  // namespace o3dmi
