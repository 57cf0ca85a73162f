// Device-side building blocks of block touch shared by vbg_touch.hip and
// vbg_stream.hip: hash insert-if-absent, wave-level duplicate elimination and
// the 4 candidate blocks of a depth ray (DepthTouchCPU,
// t/geometry/kernel/VoxelBlockGridCPU.cpp:144-180).
#pragma once

#include "common.h"

namespace o3dmi {

// Insert-if-absent of a packed key; returns the slot index and whether this
// thread created the entry. `val_out` receives the buffer index for creators
// when kAllocate (main block hash); scratch hashes do not allocate.
//
// Running out of buffer indices: callers that reserved for an exact count
// never do, and for them it is an error (kErrCapacity, reported at the next
// sync). The frame stream issues groups on an ESTIMATE of what they add
// (host/voxel_block_grid.cpp): it passes its group stamp as `overflow_stamp`,
// an overflow then records the first failing group in counters[3] and leaves
// the slot with the marker index -1 (Find: absent); the group's work is
// dropped on the device, the host reserves and replays it (RecoverOverflow).
template <bool kAllocate>
__device__ __forceinline__ bool InsertKey(const HashView& hv, int x, int y,
                                          int z, unsigned& slot_out,
                                          int overflow_stamp = 0) {
    slot_out = 0;
    // The group is dropped already: no walk in a full table. (A plain, cached
    // load: a stale zero only costs the walk; an agent-scope load in front of
    // every insert cost the cold pass of the headline 7 %.)
    if (kAllocate && overflow_stamp != 0 && hv.counters[3] != 0) return false;
    const int claim = ClaimSlot(hv, PackKey(x, y, z), slot_out,
                                !(kAllocate && overflow_stamp != 0));
    if (claim == -1 && kAllocate && overflow_stamp != 0) {
        // more new keys than the table has slots: the same overflow
        atomicCAS(&hv.counters[3], 0, overflow_stamp);
        slot_out = 0;
        return false;
    }
    if (claim != 1) return false;
    if (kAllocate) {
        const unsigned h = slot_out;
        int top = atomicAdd(&hv.counters[0], 1);
        if (top >= hv.capacity) {
            if (overflow_stamp != 0) {
                atomicCAS(&hv.counters[3], 0, overflow_stamp);
                __hip_atomic_store(&hv.slot_vals[h], -1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                return true;
            }
            atomicOr(&hv.counters[1], kErrCapacity);
            // Leave a valid (but shared) index so later kernels stay in
            // bounds; the error is reported at sync.
            hv.slot_vals[h] = 0;
            return true;
        }
        int idx = hv.heap[top];
        hv.key_buffer[3 * idx + 0] = x;
        hv.key_buffer[3 * idx + 1] = y;
        hv.key_buffer[3 * idx + 2] = z;
        // write-through: the ready-list compaction of the frame stream reads
        // it from another workgroup of the SAME launch (vbg_stream.hip)
        __hip_atomic_store(&hv.slot_vals[h], idx, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// True for exactly one lane among the active lanes of the wave that hold the
// same packed key (the lowest such lane). Lanes with valid == false never lead.
__device__ __forceinline__ bool WaveLeaderForKey(unsigned long long k,
                                                 bool valid) {
    // Cheap neighbour filter first: adjacent rays nearly always agree.
    bool leader = valid;
    unsigned long long remaining = __ballot(valid);
    int lane = threadIdx.x & 63;
    bool decided = !valid;
    while (remaining) {
        int first = __ffsll((long long)remaining) - 1;
        unsigned long long kf = __shfl(k, first);
        bool same = valid && (k == kf);
        unsigned long long same_mask = __ballot(same);
        if (same && !decided) {
            leader = (lane == first);
            decided = true;
        }
        remaining &= ~same_mask;
    }
    return leader;
}

struct TouchParams {
    Camera cam;  // intrinsics + POSE (inverse extrinsic), scale 1
    int rows, cols, stride;
    int rows_strided, cols_strided;
    float block_size, sdf_trunc, depth_scale, depth_max;
};

// Computes the 4 candidate block keys of strided pixel `workload_idx`
// (VoxelBlockGridCPU.cpp:144-180). Returns false when the pixel is invalid.
template <typename depth_t>
__device__ __forceinline__ bool RayCandidates(const TouchParams& p,
                                              const depth_t* __restrict__ depth,
                                              int workload_idx, int (&xb)[4],
                                              int (&yb)[4], int (&zb)[4]) {
    int y = (workload_idx / p.cols_strided) * p.stride;
    int x = (workload_idx % p.cols_strided) * p.stride;
    float d = (float)depth[(int64_t)y * p.cols + x] / p.depth_scale;
    if (!(d > 0 && d < p.depth_max)) return false;

    float x_c, y_c, z_c, x_g, y_g, z_g;
    p.cam.Unproject((float)x, (float)y, 1.0f, x_c, y_c, z_c);
    p.cam.RigidTransform(x_c, y_c, z_c, x_g, y_g, z_g);
    float x_o = p.cam.e[0][3], y_o = p.cam.e[1][3], z_o = p.cam.e[2][3];
    float x_d = x_g - x_o, y_d = y_g - y_o, z_d = z_g - z_o;

    const float t_min = fmaxf(d - p.sdf_trunc, 0.0f);
    const float t_max = fminf(d + p.sdf_trunc, p.depth_max);
    const float t_step = (t_max - t_min) / 3;
    float t = t_min;
#pragma unroll
    for (int step = 0; step < 4; ++step) {
        xb[step] = (int)floorf((x_o + t * x_d) / p.block_size);
        yb[step] = (int)floorf((y_o + t * y_d) / p.block_size);
        zb[step] = (int)floorf((z_o + t * z_d) / p.block_size);
        t += t_step;
    }
    return true;
}


inline TouchParams MakeTouchParams(const double* intrinsic, const double* extrinsic,
                            int rows, int cols, int stride, int resolution,
                            float voxel_size, float sdf_trunc,
                            float depth_scale, float depth_max) {
    TouchParams p;
    double pose[16];
    InverseTransformation(extrinsic, pose);
    p.cam = Camera::Make(intrinsic, pose, 1.0f);
    p.rows = rows;
    p.cols = cols;
    p.stride = stride;
    p.rows_strided = rows / stride;
    p.cols_strided = cols / stride;
    p.block_size = voxel_size * resolution;
    p.sdf_trunc = sdf_trunc;
    p.depth_scale = depth_scale;
    p.depth_max = depth_max;
    return p;
}


}  // namespace o3dmi
