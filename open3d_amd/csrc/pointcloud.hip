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
//      slot's smallest point index (atomicMin);
//   2. flag first points, exclusive-scan the flags -> dense voxel ids in
//      first-occurrence order;
//   3. stable radix sort of (voxel id, point index): every voxel's points end
//      up contiguous and still in point order;
//   4. one lane per voxel walks its segment and adds sequentially in float32.
// Scan and sort are rocPRIM primitives (via hipCUB); the rest is below.

#include <hipcub/hipcub.hpp>

#include "common.h"
#include "o3d_mi355x_host.h"

namespace o3dmi {
namespace {

struct VdsTable {
    unsigned long long* keys;  // [n_slots], kEmptyKey when free
    int* first;                // [n_slots] smallest point index
    int* voxel;                // [n_slots] dense voxel id
    unsigned mask;
};

template <typename T>
__global__ void VdsInsertKernel(const T* __restrict__ pos, int64_t n, T vs,
                                VdsTable tb, int* __restrict__ slot_of_point,
                                int* __restrict__ err) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        // (p / vs).Floor().To(Int64)
        const long long cx = (long long)floor(pos[3 * i + 0] / vs);
        const long long cy = (long long)floor(pos[3 * i + 1] / vs);
        const long long cz = (long long)floor(pos[3 * i + 2] / vs);
        if (cx < -kKeyBias || cx >= kKeyBias || cy < -kKeyBias ||
            cy >= kKeyBias || cz < -kKeyBias || cz >= kKeyBias) {
            atomicOr(err, kErrKeyRange);
            slot_of_point[i] = 0;
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
        atomicMin(&tb.first[h], (int)i);
    }
}

__global__ void VdsFlagKernel(const int* __restrict__ slot_of_point, int64_t n,
                              VdsTable tb, int* __restrict__ flags) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        flags[i] = tb.first[slot_of_point[i]] == (int)i ? 1 : 0;
}

// Dense voxel id of every slot (from its first point), then of every point.
__global__ void VdsLabelSlotsKernel(const int* __restrict__ slot_of_point,
                                    const int* __restrict__ flags,
                                    const int* __restrict__ scan, int64_t n,
                                    VdsTable tb) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        if (flags[i]) tb.voxel[slot_of_point[i]] = scan[i];
}
__global__ void VdsLabelPointsKernel(const int* __restrict__ slot_of_point,
                                     int64_t n, VdsTable tb,
                                     int* __restrict__ voxel_of_point,
                                     int* __restrict__ point_index) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        voxel_of_point[i] = tb.voxel[slot_of_point[i]];
        point_index[i] = (int)i;
    }
}

__global__ void VdsSegmentsKernel(const int* __restrict__ sorted_voxel,
                                  int64_t n, int* __restrict__ seg_start) {
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        if (j == 0 || sorted_voxel[j] != sorted_voxel[j - 1])
            seg_start[sorted_voxel[j]] = (int)j;
        if (j == n - 1) seg_start[sorted_voxel[j] + 1] = (int)n;
    }
}

template <typename T>
__global__ void VdsReduceKernel(const T* __restrict__ pos,
                                const T* __restrict__ nrm,
                                const int* __restrict__ sorted_point,
                                const int* __restrict__ seg_start, int64_t m,
                                T* __restrict__ out_pos,
                                T* __restrict__ out_nrm) {
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < m;
         v += (int64_t)gridDim.x * blockDim.x) {
        const int b = seg_start[v], e = seg_start[v + 1];
        float cnt = 0.f, sp[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f};
        for (int j = b; j < e; ++j) {
            const int64_t i = sorted_point[j];
            cnt += 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sp[c] += (float)pos[3 * i + c];
                if (nrm) sn[c] += (float)nrm[3 * i + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out_pos[3 * v + c] = (T)(sp[c] / cnt);
            if (nrm) out_nrm[3 * v + c] = (T)(sn[c] / cnt);
        }
    }
}

struct Scratch {
    std::vector<void*> ptrs;
    hipStream_t stream = nullptr;
    ~Scratch() {
        // Pooled blocks may only be released once the stream has drained
        // (already the case on the success path).
        (void)hipStreamSynchronize(stream);
        for (void* p : ptrs) PoolFree(p);
    }
    template <typename T>
    int Alloc(T** p, size_t count) {
        void* q = nullptr;
        int st_ = PoolAlloc(&q, sizeof(T) * (count ? count : 1));
        if (st_) return st_;
        ptrs.push_back(q);
        *p = (T*)q;
        return O3DMI_OK;
    }
};

template <typename T>
int VoxelDownSampleImpl(const T* pos, const T* nrm, int64_t n,
                        double voxel_size, T* out_pos, T* out_nrm,
                        int64_t* m_out, hipStream_t s) {
    *m_out = 0;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(n < (1ll << 31), "VoxelDownSample: too many points");
    Scratch sc;
    sc.stream = s;
    int64_t n_slots = 1024;
    while (n_slots < 2 * n) n_slots <<= 1;
    VdsTable tb;
    int st;
    if ((st = sc.Alloc(&tb.keys, (size_t)n_slots))) return st;
    if ((st = sc.Alloc(&tb.first, (size_t)n_slots))) return st;
    if ((st = sc.Alloc(&tb.voxel, (size_t)n_slots))) return st;
    tb.mask = (unsigned)(n_slots - 1);
    int *slot_of_point, *flags, *scan, *voxel_of_point, *point_index,
            *sorted_voxel, *sorted_point, *seg_start, *err;
    if ((st = sc.Alloc(&slot_of_point, (size_t)n))) return st;
    if ((st = sc.Alloc(&flags, (size_t)n + 1))) return st;
    if ((st = sc.Alloc(&scan, (size_t)n + 1))) return st;
    if ((st = sc.Alloc(&voxel_of_point, (size_t)n))) return st;
    if ((st = sc.Alloc(&point_index, (size_t)n))) return st;
    if ((st = sc.Alloc(&sorted_voxel, (size_t)n))) return st;
    if ((st = sc.Alloc(&sorted_point, (size_t)n))) return st;
    if ((st = sc.Alloc(&seg_start, (size_t)n + 2))) return st;
    if ((st = sc.Alloc(&err, 4))) return st;
    O3DMI_HIP_CHECK(hipMemsetAsync(tb.keys, 0xFF,
                                   sizeof(unsigned long long) * (size_t)n_slots,
                                   s));
    O3DMI_HIP_CHECK(hipMemsetAsync(tb.first, 0x7F,
                                   sizeof(int) * (size_t)n_slots, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int) * 4, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(flags + n, 0, sizeof(int), s));

    const dim3 grid(GridFor(n, kBlock)), block(kBlock);
    hipLaunchKernelGGL(VdsInsertKernel<T>, grid, block, 0, s, pos, n,
                       (T)voxel_size, tb, slot_of_point, err);
    hipLaunchKernelGGL(VdsFlagKernel, grid, block, 0, s, slot_of_point, n, tb,
                       flags);
    // exclusive scan over n+1 entries: scan[n] = number of voxels
    size_t tmp_scan = 0, tmp_sort = 0;
    O3DMI_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(
            nullptr, tmp_scan, flags, scan, (int)(n + 1), s));
    O3DMI_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(
            nullptr, tmp_sort, voxel_of_point, sorted_voxel, point_index,
            sorted_point, (int)n, 0, 32, s));
    char* tmp = nullptr;
    const size_t tmp_bytes = tmp_scan > tmp_sort ? tmp_scan : tmp_sort;
    if ((st = sc.Alloc(&tmp, tmp_bytes))) return st;
    size_t tb1 = tmp_bytes;
    O3DMI_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tb1, flags, scan,
                                                     (int)(n + 1), s));
    int host[2] = {0, 0};
    O3DMI_HIP_CHECK(hipMemcpyAsync(&host[0], scan + n, sizeof(int),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipMemcpyAsync(&host[1], err, sizeof(int),
                                   hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(VdsLabelSlotsKernel, grid, block, 0, s, slot_of_point,
                       flags, scan, n, tb);
    hipLaunchKernelGGL(VdsLabelPointsKernel, grid, block, 0, s, slot_of_point,
                       n, tb, voxel_of_point, point_index);
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (host[1] & kErrKeyRange) {
        SetLastError("VoxelDownSample: voxel coordinate outside +-2^20");
        return O3DMI_ERR_KEY_RANGE;
    }
    const int64_t m = host[0];
    int bits = 1;
    while ((1ll << bits) < m) ++bits;
    size_t tb2 = tmp_bytes;
    O3DMI_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(
            tmp, tb2, voxel_of_point, sorted_voxel, point_index, sorted_point,
            (int)n, 0, bits, s));
    hipLaunchKernelGGL(VdsSegmentsKernel, grid, block, 0, s, sorted_voxel, n,
                       seg_start);
    hipLaunchKernelGGL(VdsReduceKernel<T>, dim3(GridFor(m, kBlock)), block, 0,
                       s, pos, nrm, sorted_point, seg_start, m, out_pos,
                       out_nrm);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    *m_out = m;
    return O3DMI_OK;
}

}  // namespace
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
    hipStream_t s = (hipStream_t)stream;
    if (dtype == O3DMI_F64)
        return VoxelDownSampleImpl<double>(
                (const double*)positions_dev, (const double*)normals_dev, n,
                voxel_size, (double*)out_positions_dev,
                (double*)out_normals_dev, m_out, s);
    return VoxelDownSampleImpl<float>(
            (const float*)positions_dev, (const float*)normals_dev, n,
            voxel_size, (float*)out_positions_dev, (float*)out_normals_dev,
            m_out, s);
}
