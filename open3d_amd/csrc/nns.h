// Private header shared by nns.hip (index build + searches) and icp.hip (the
// fused search + accumulate kernel): the bucketed uniform grid of the
// fixed-radius / KNN index.
//
// Index design: cell edge = radius * (1 + 1e-3) (KNN: chosen from the measured
// density). Target points are *reordered* by bucket into 16-byte (32-byte for
// f64) records {x,y,z,original index}, normals likewise, so a query's cell
// visits read contiguous memory instead of chasing a CSR index through 12-byte
// AoS points. A bucket's records are contiguous, but buckets lie in memory in
// the order their ranges were handed out (one wave-aggregated atomic per 64
// buckets) -- no prefix sum over the table: the build is a memset and three
// launches (count, hand out ranges, scatter points + normals) with no scratch
// and nothing to wait for. ranges[b] = {end, count} of bucket b: one 8-byte
// load per visited cell. Nothing a search returns depends on the order in
// memory (every result is defined by (d2, original index)). Cell coordinates are computed in float64 on both the build and
// the query side so that large-offset clouds (1000 m + 5 cm radius, cf.
// cpp/tests/core/NearestNeighborSearch.cpp:831-869) bin consistently.
#pragma once

#include "common.h"

struct o3dmi_nns {
    int dtype = O3DMI_F32;
    int64_t n = 0;
    double radius = 0, inv_cell = 0;
    int64_t n_buckets = 0;
    void* sorted_pts = nullptr;      // Rec4<T>[n]
    void* sorted_normals = nullptr;  // Rec4<T>[n], optional
    uint2* ranges = nullptr;         // [n_buckets + 1] {end, count}; the last
                                     // entry is the build's running total
    double* partials = nullptr;      // [kCUs*4, kNumSums]
    int* tickets = nullptr;          // 9 ticket words of the search launch's
                                     // final-sum tail (icp.hip SumTail), in
                                     // the last row of `partials`; zero
                                     // between launches
};

namespace o3dmi {

constexpr int kNumSums = 32;  // 29 + sum d2 + match count + pad

template <typename T> struct Rec4;  // {x,y,z,w}
template <> struct alignas(16) Rec4<float> { float x, y, z; int w; };
template <> struct alignas(32) Rec4<double> { double x, y, z; long long w; };

// 32-bit mixing of the (wrapped) cell coordinates: three multiplies and a
// murmur3 finaliser. (The first version went through 64-bit products and a
// 64-bit finaliser -- five 64-bit multiplies, ~45 vector instructions per
// visited cell, a fifth of the fused ICP search's instruction stream.)
__device__ __forceinline__ unsigned HashCell(long long cx, long long cy,
                                             long long cz) {
    unsigned h = (unsigned)cx * 0x9E3779B1u;
    h ^= ((unsigned)cy * 0x85EBCA77u) + (h >> 15);
    h ^= ((unsigned)cz * 0xC2B2AE3Du) + (h << 11);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

template <typename T>
__device__ __forceinline__ void CellOf(const T* p, double inv_cell,
                                       long long& cx, long long& cy,
                                       long long& cz) {
    cx = (long long)floor((double)p[0] * inv_cell);
    cy = (long long)floor((double)p[1] * inv_cell);
    cz = (long long)floor((double)p[2] * inv_cell);
}

template <typename T>
__device__ __forceinline__ int RecIndex(const Rec4<T>& r) {
    return (int)r.w;
}

// Gather an {N,3} attribute (normals) into sorted record order.
template <typename T>
struct NnsView {
    const Rec4<T>* sorted;      // [n] bucket-ordered {x,y,z,idx}
    const uint2* ranges;        // [n_buckets] {end, count}
    double inv_cell;
    unsigned mask;
    T radius_squared;
};

// Records of bucket b: sorted[s .. e).
template <typename T>
__device__ __forceinline__ void BucketRange(const NnsView<T>& nv, unsigned b,
                                            unsigned& s, unsigned& e) {
    const uint2 r = nv.ranges[b];
    e = r.x;
    s = r.x - r.y;
}

template <typename T>
inline NnsView<T> MakeView(const o3dmi_nns* nns) {
    NnsView<T> v;
    v.sorted = (const Rec4<T>*)nns->sorted_pts;
    v.ranges = nns->ranges;
    v.inv_cell = nns->inv_cell;
    v.mask = (unsigned)(nns->n_buckets - 1);
    const T r = (T)nns->radius;  // NanoFlannImpl.h:332: T radius_squared
    v.radius_squared = r * r;
    return v;
}

}  // namespace o3dmi
