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

// Device view of the spatial hash, passed by value to kernels.
struct HashView {
    unsigned long long* slot_keys;  // [n_slots] packed key / empty / tombstone
    int* slot_vals;                 // [n_slots] buffer index of the slot
    unsigned long long* slot_touch; // [n_slots] (touch stamp << 8) | frame bits
    int* heap;                      // [capacity] free buffer indices
    int* counters;                  // [0]=heap_top, [1]=error flags
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

    // Lookup; -1 when absent.
    __device__ __forceinline__ int Find(int x, int y, int z) const {
        if (!KeyInRange(x, y, z)) return -1;
        unsigned long long k = PackKey(x, y, z);
        unsigned h = HashKey(k) & mask;
        while (true) {
            unsigned long long cur = slot_keys[h];
            if (cur == k) return slot_vals[h];
            if (cur == kEmptyKey) return -1;
            h = (h + 1) & mask;
        }
    }
};

// Marks `slot` as touched by frame `bit` of the frame group `stamp`. The word
// holds (stamp << 8) | one bit per frame of the group; a word carrying an older
// stamp is stale and is replaced. Returns true for exactly one caller per
// (slot, stamp): the one that moved the word to this stamp.
__device__ __forceinline__ bool TouchSlot(const HashView& hv, unsigned slot,
                                          unsigned long long stamp, int bit) {
    unsigned long long* w = &hv.slot_touch[slot];
    unsigned long long cur = *w;  // possibly stale; the CAS corrects it
    while (true) {
        const bool fresh = (cur >> 8) != stamp;
        const unsigned long long want =
                fresh ? ((stamp << 8) | (1ull << bit)) : (cur | (1ull << bit));
        if (want == cur) return false;
        const unsigned long long old = atomicCAS(w, cur, want);
        if (old == cur) return fresh;
        cur = old;
    }
}

constexpr int kErrKeyRange = 1;
constexpr int kErrCapacity = 2;

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
};
