// Deterministic N-value sum reduction for the Gauss-Newton style kernels
// (29 sums of RGB-D odometry, 21 sums of the information matrix): float64
// running sums per lane, wave64 __shfl_down tree, LDS across the waves of a
// workgroup, one partial row per workgroup, and a single-workgroup final pass
// over the rows. The launch geometry is a function of the element count only,
// so results are run-to-run identical (the reference's TBB parallel_reduce /
// CUDA atomicAdd versions are not: RGBDOdometryCPU.cpp:315-360,
// RGBDOdometryCUDA.cu).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "mailbox.h"

namespace o3dmi {

constexpr int kSumsBlock = 256;
// One workgroup per CU at most: every lane then carries several elements
// before the (comparatively expensive) 29-value wave reduction, and the final
// pass has at most 256 partial rows to read.
constexpr int kSumsMaxGrid = 256;
inline int SumsGrid(int64_t n) {
    // Coarse pyramid levels (<= 64k elements) are pure latency: one element
    // per lane keeps their dependent-load chain as short as it can be.
    int64_t g = (n + kSumsBlock - 1) / kSumsBlock;
    if (g > kSumsMaxGrid) g = kSumsMaxGrid;
    if (g < 1) g = 1;
    return (int)g;
}

// Sum of up to 32 float64 values per lane over the 64 lanes of a wave, as a
// reduce-scatter: each butterfly step (lane ^ 32, 16, 8, 4, 2) halves the
// number of values a lane carries, a last step (lane ^ 1) joins the two halves.
// 32 double shuffles instead of N x 6 for one-value-at-a-time trees (the LDS
// crossbar, not the adds, is what a 29-value reduction costs). On return lane l
// holds the wave total of value (l >> 1). Fixed order: run-to-run identical.
template <int N>
__device__ __forceinline__ double WaveReduceScatter(const double (&A)[N]) {
    static_assert(N <= 32, "at most 32 values");
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = k < N ? A[k] : 0.0;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        // lanes with bit (2 * half) of the lane id set keep the upper half
        const bool upper = (lane & (2 * half)) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            const double send = upper ? v[k] : v[k + half];
            const double keep = upper ? v[k + half] : v[k];
            v[k] = keep + __shfl_xor(send, 2 * half, 64);
        }
    }
    return v[0] + __shfl_xor(v[0], 1, 64);
}

// partials: [gridDim.x][N] float64.
template <int N>
__device__ __forceinline__ void BlockSumAndStore(double (&A)[N],
                                                 double* __restrict__ partials) {
    __shared__ double lds[kSumsBlock / 64][32];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const double t = WaveReduceScatter<N>(A);
    if ((lane & 1) == 0) lds[wave][lane >> 1] = t;
    __syncthreads();
    if (threadIdx.x < N) {
        double v = 0;
#pragma unroll
        for (int wv = 0; wv < kSumsBlock / 64; ++wv) v += lds[wv][threadIdx.x];
        partials[(int64_t)blockIdx.x * N + threadIdx.x] = v;
    }
}

// One workgroup of kFinalThreads lanes: lane (r, c) = (tid / 32, tid % 32)
// strides over the rows (32 row-lanes, independent loads), the row-lanes are
// then added in a fixed order. N <= 32.
// mail_data / mail_flag (optional): host mailbox, see mailbox.h.
constexpr int kFinalThreads = 1024;
constexpr int kFinalRowLanes = kFinalThreads / 32;

template <int N>
__global__ void __launch_bounds__(kFinalThreads)
FinalSumKernel(const double* __restrict__ partials, int n_rows,
               double* __restrict__ out, double* mail_data, int* mail_flag,
               int mail_seq) {
    __shared__ double lds[kFinalRowLanes][32];
    const int col = threadIdx.x & 31;
    const int rl = threadIdx.x >> 5;
    // rows rl, rl + 32, ...: sixteen loads in flight at a time (the rows were
    // written by other XCDs and come from the fabric, ~1 us each if taken one
    // after the other), added in row order.
    double v = 0;
    if (col < N) {
        for (int r0 = rl; r0 < n_rows; r0 += 16 * kFinalRowLanes) {
            double x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int r = r0 + k * kFinalRowLanes;
                x[k] = r < n_rows ? partials[(int64_t)r * N + col] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) v += x[k];
        }
    }
    lds[rl][col] = v;
    __syncthreads();
    if (threadIdx.x < N) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < kFinalRowLanes; ++k) s += lds[k][threadIdx.x];
        if (out) out[threadIdx.x] = s;
        if (mail_data) mail_data[threadIdx.x] = s;
    }
    if (mail_flag) MailboxPublish(mail_flag, mail_seq);
}

}  // namespace o3dmi
