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

// partials: [gridDim.x][N] float64.
template <int N>
__device__ __forceinline__ void BlockSumAndStore(double (&A)[N],
                                                 double* __restrict__ partials) {
    __shared__ double lds[kSumsBlock / 64][N];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double v = A[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) lds[wave][k] = v;
    }
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
    double v = 0;
    if (col < N)
        for (int r = rl; r < n_rows; r += kFinalRowLanes)
            v += partials[(int64_t)r * N + col];
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
