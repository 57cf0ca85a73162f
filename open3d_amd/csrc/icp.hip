// ICP estimator kernels for MI355X.
//
//   o3dmi_icp_p2plane_accumulate  <- ComputePosePointToPlaneCUDA (t/pipelines/kernel/
//                                    RegistrationCUDA.cu:29-117, RegistrationImpl.h:251-287)
//   o3dmi_icp_{p2point,symmetric,colored,information}_accumulate
//                                 <- the other reductions of RegistrationCUDA.cu /
//                                    RegistrationCPU.cpp:124-340,495-735
//   o3dmi_icp_search_accumulate   fused search + fitness/rmse sums + accumulation
//   o3dmi_transform_points/normals<- TransformPointsCUDA/TransformNormalsCUDA
//                                    (t/geometry/kernel/TransformImpl.h:19-60)
//   o3dmi_decode_and_solve6x6, o3dmi_pose_to_transformation,
//   o3dmi_compute_rt_p2point, o3dmi_symmetric_pose_to_transformation (host)
//                                 <- TransformationConverter.cpp:81-133,189-226,
//                                    RegistrationCPU.cpp:640-650
//
// The search index (nns.h, nns.hip) is shared with the stand-alone searches.
//
// Reduction design (new): a fixed persistent grid; each lane keeps the 29 (+2)
// sums in float64 registers, a wave64 __shfl_down tree, one LDS stage per
// workgroup, one partial row per workgroup, and a single-workgroup second
// stage in fixed order => run-to-run deterministic (the reference's CUDA path
// uses float atomics, its CPU path a TBB tree of unspecified shape).

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nns.h"
#include "mailbox.h"

#ifndef O3DMI_ABLATE_TAIL
#define O3DMI_ABLATE_TAIL 0
#endif
#include "reduce_sums.h"

namespace o3dmi {
namespace {

// ---- robust kernels ---------------------------------------------------------
// RobustKernelImpl.h:35-126, literal: the double-typed literals promote parts
// of each expression to float64 before the result is narrowed to scalar_t.
template <typename T>
__device__ __forceinline__ T SquareT(T x) { return x * x; }

struct RobustParams {
    int method;
    double scaling, shape;
    int generalized_case;  // 0: ~2, 1: ~0 (never true, see below), 2: <-1e7, 3: else
};

inline bool IsCloseHost(double x, double y, double rtol) {
    // GeometryMacros.h:58-63; with y == 0 this is never true.
    return (x > (1.0 - rtol) * y) && (x < (1.0 + rtol) * y);
}

template <typename T>
__device__ __forceinline__ T RobustWeight(const RobustParams& rp, T residual) {
    const T scale = (T)rp.scaling;
    switch (rp.method) {
        case O3DMI_L2_LOSS:
            return (T)1.0;
        case O3DMI_L1_LOSS:
            return (T)(1.0 / (double)fabs(residual));
        case O3DMI_HUBER_LOSS: {
            T a = fabs(residual);
            return scale / (a < scale ? scale : a);
        }
        case O3DMI_CAUCHY_LOSS:
            return (T)(1.0 / (1.0 + (double)SquareT<T>(residual / scale)));
        case O3DMI_GM_LOSS:
            return scale / SquareT<T>(scale + SquareT<T>(residual));
        case O3DMI_TUKEY_LOSS: {
            T a = fabs(residual) / scale;
            T m = (T)1.0 < a ? (T)1.0 : a;
            double v = 1.0 - (double)SquareT<T>(m);
            return (T)(v * v);
        }
        case O3DMI_GENERALIZED_LOSS: {
            if (rp.generalized_case == 0) {
                return (T)(1.0 / (double)SquareT<T>(scale));
            } else if (rp.generalized_case == 1) {
                return (T)(2.0 / (double)(SquareT<T>(residual) +
                                          2 * SquareT<T>(scale)));
            } else if (rp.generalized_case == 2) {
                return (T)(exp((double)SquareT<T>(residual / scale) / (-2.0)) /
                           (double)SquareT<T>(scale));
            } else {
                return (T)(pow(((double)SquareT<T>(residual / scale) /
                                        fabs(rp.shape - 2.0) +
                                1),
                               ((rp.shape / 2.0) - 1.0)) /
                           (double)SquareT<T>(scale));
            }
        }
        default:
            return (T)1.0;
    }
}

RobustParams MakeRobust(int method, double scaling, double shape) {
    RobustParams rp;
    rp.method = method;
    rp.scaling = scaling;
    rp.shape = shape;
    if (IsCloseHost(shape, 2.0, 1e-3)) rp.generalized_case = 0;
    else if (IsCloseHost(shape, 0.0, 1e-3)) rp.generalized_case = 1;
    else if (shape < -1e7) rp.generalized_case = 2;
    else rp.generalized_case = 3;
    return rp;
}

// ---- 29(+2)-value reduction -------------------------------------------------
// Per-correspondence terms in T exactly as RegistrationCPU.cpp:62-74; the
// running sums are float64.
template <typename T, typename Sums = double[kNumSums], bool kL2 = false>
__device__ __forceinline__ void AccumulateP2Plane(
        Sums& A, T sx, T sy, T sz, T tx, T ty, T tz, T nx, T ny,
        T nz, const RobustParams& rp) {
    // RegistrationImpl.h:274-284
    T r = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
    T J[6];
    J[0] = nz * sy - ny * sz;
    J[1] = nx * sz - nz * sx;
    J[2] = ny * sx - nx * sy;
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    // kL2: the caller knows the kernel is L2Loss (weight 1, the common case);
    // the general form drags float64 pow / exp through the register file
    T w = kL2 ? T(1) : RobustWeight<T>(rp, r);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += J[j] * w * J[k];  // widened by the sum
            ++i;
        }
        A[21 + j] += J[j] * w * r;
    }
    A[27] += r;
    A[28] += 1.0;
}

// Point-to-point (Horn) sums in one pass: sum s, sum t, sum t s^T, count
// (RegistrationCPU.cpp:495-617 forms the means first and the centred products
// in a second pass; the centred covariance follows on the host from these raw
// moments in float64, where products of two Float32 values are exact).
template <typename T, typename Sums = double[kNumSums]>
__device__ __forceinline__ void AccumulateP2Point(Sums& A, T sx,
                                                  T sy, T sz, T tx, T ty,
                                                  T tz) {
    const double s[3] = {(double)sx, (double)sy, (double)sz};
    const double t[3] = {(double)tx, (double)ty, (double)tz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[k] += s[k];
        A[3 + k] += t[k];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) A[6 + 3 * j + k] += t[j] * s[k];
    A[15] += 1.0;
}

// GetInformationJacobians + the 21 sums of ComputeInformationMatrixKernelCPU
// (RegistrationImpl.h:686-715, RegistrationCPU.cpp:652-701): per matched target
// point G^T G with G = [-[t]x | I], each packed-lower-triangle term formed in T
// as J_x[j] J_x[k] + J_y[j] J_y[k] + J_z[j] J_z[k], summed in float64.
template <typename T, typename Sums = double[kNumSums]>
__device__ __forceinline__ void AccumulateInformation(Sums& A,
                                                      T tx, T ty, T tz) {
    const T Jx[6] = {T(0), tz, -ty, T(1), T(0), T(0)};
    const T Jy[6] = {-tz, T(0), tx, T(0), T(1), T(0)};
    const T Jz[6] = {ty, -tx, T(0), T(0), T(0), T(1)};
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(Jx[j] * Jx[k] + Jy[j] * Jy[k] + Jz[j] * Jz[k]);
            ++i;
        }
    }
}

// GetJacobianSymmetric + the 29 sums of ComputePoseSymmetricKernelCPU
// (RegistrationImpl.h:323-386, RegistrationCPU.cpp:124-180): plane normal
// n_t + sign(n_s . n_t) n_s, Jacobian and right-hand side about the
// correspondence means, robust weight from the un-centred residual; A[27] is
// the sum of squared un-centred residuals.
template <typename T>
struct Means3 { T s[3], t[3]; };

template <typename T>
__device__ __forceinline__ void AccumulateSymmetric(
        double (&A)[kNumSums], T sx, T sy, T sz, T tx, T ty, T tz, T snx, T sny,
        T snz, T tnx, T tny, T tnz, const Means3<T>& mean,
        const RobustParams& rp) {
    const T normal_dot = snx * tnx + sny * tny + snz * tnz;
    const T normal_sign = normal_dot < T(0) ? T(-1) : T(1);
    const T nx = tnx + normal_sign * snx;
    const T ny = tny + normal_sign * sny;
    const T nz = tnz + normal_sign * snz;
    const T sx_centered = sx - mean.s[0];
    const T sy_centered = sy - mean.s[1];
    const T sz_centered = sz - mean.s[2];
    const T tx_centered = tx - mean.t[0];
    const T ty_centered = ty - mean.t[1];
    const T tz_centered = tz - mean.t[2];
    const T sum_x = sx_centered + tx_centered;
    const T sum_y = sy_centered + ty_centered;
    const T sum_z = sz_centered + tz_centered;
    T J[6];
    J[0] = sum_y * nz - sum_z * ny;
    J[1] = sum_z * nx - sum_x * nz;
    J[2] = sum_x * ny - sum_y * nx;
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    const T centered_residual = (sx_centered - tx_centered) * nx +
                                (sy_centered - ty_centered) * ny +
                                (sz_centered - tz_centered) * nz;
    const T objective_residual =
            (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
    const T w = RobustWeight<T>(rp, objective_residual);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += J[j] * w * J[k];  // widened by the sum
            ++i;
        }
        A[21 + j] += (double)(J[j] * w * centered_residual);
    }
    A[27] += (double)(objective_residual * objective_residual);
    A[28] += 1.0;
}

// GetJacobianColoredICP + the 29 sums of ComputePoseColoredICPKernelCPU
// (RegistrationImpl.h:388-466, RegistrationCPU.cpp:220-290): geometric and
// photometric rows, each with its own robust weight; A[27] = sum r_G^2 + r_I^2.
template <typename T>
__device__ __forceinline__ void AccumulateColored(
        double (&A)[kNumSums], const T* vs, const T* cs, const T* vt,
        const T* nt, const T* ct, const T* dit, T sqrt_lambda_geometric,
        T sqrt_lambda_photometric, const RobustParams& rp) {
    const T d = (vs[0] - vt[0]) * nt[0] + (vs[1] - vt[1]) * nt[1] +
                (vs[2] - vt[2]) * nt[2];
    T J_G[6], J_I[6];
    J_G[0] = sqrt_lambda_geometric * (-vs[2] * nt[1] + vs[1] * nt[2]);
    J_G[1] = sqrt_lambda_geometric * (vs[2] * nt[0] - vs[0] * nt[2]);
    J_G[2] = sqrt_lambda_geometric * (-vs[1] * nt[0] + vs[0] * nt[1]);
    J_G[3] = sqrt_lambda_geometric * nt[0];
    J_G[4] = sqrt_lambda_geometric * nt[1];
    J_G[5] = sqrt_lambda_geometric * nt[2];
    const T r_G = sqrt_lambda_geometric * d;
    const T vs_proj[3] = {vs[0] - d * nt[0], vs[1] - d * nt[1],
                          vs[2] - d * nt[2]};
    // "/ 3.0": float64 division, then narrowed
    const T intensity_source = (cs[0] + cs[1] + cs[2]) / 3.0;
    const T intensity_target = (ct[0] + ct[1] + ct[2]) / 3.0;
    const T is_proj = dit[0] * (vs_proj[0] - vt[0]) +
                      dit[1] * (vs_proj[1] - vt[1]) +
                      dit[2] * (vs_proj[2] - vt[2]) + intensity_target;
    const T s = dit[0] * nt[0] + dit[1] * nt[1] + dit[2] * nt[2];
    const T ditM[3] = {s * nt[0] - dit[0], s * nt[1] - dit[1],
                       s * nt[2] - dit[2]};
    J_I[0] = sqrt_lambda_photometric * (-vs[2] * ditM[1] + vs[1] * ditM[2]);
    J_I[1] = sqrt_lambda_photometric * (vs[2] * ditM[0] - vs[0] * ditM[2]);
    J_I[2] = sqrt_lambda_photometric * (-vs[1] * ditM[0] + vs[0] * ditM[1]);
    J_I[3] = sqrt_lambda_photometric * ditM[0];
    J_I[4] = sqrt_lambda_photometric * ditM[1];
    J_I[5] = sqrt_lambda_photometric * ditM[2];
    const T r_I = sqrt_lambda_photometric * (intensity_source - is_proj);
    const T w_G = RobustWeight<T>(rp, r_G);
    const T w_I = RobustWeight<T>(rp, r_I);
    int i = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int k = 0; k <= j; ++k) {
            A[i] += (double)(J_G[j] * w_G * J_G[k] + J_I[j] * w_I * J_I[k]);
            ++i;
        }
        A[21 + j] += (double)(J_G[j] * w_G * r_G + J_I[j] * w_I * r_I);
    }
    A[27] += (double)(r_G * r_G + r_I * r_I);
    A[28] += 1.0;
}

constexpr int kReduceBlock = 256;

// Reduce-scatter wave reduction (reduce_sums.h), LDS across the 4 waves, one
// row per workgroup.
__device__ __forceinline__ void BlockReduceAndStore(double (&A)[kNumSums],
                                                    double* __restrict__ partials) {
    static_assert(kReduceBlock == kSumsBlock, "shared reduction geometry");
    BlockSumAndStore<kNumSums>(A, partials);
}

__global__ void FinalReduceKernel(const double* __restrict__ partials,
                                  int n_rows, double* __restrict__ out,
                                  int n_out, double* mail_data, int* mail_flag,
                                  int mail_seq) {
    // one wave per output column would waste lanes; 32 columns x 8 row-lanes.
    __shared__ double lds[8][kNumSums];
    int col = threadIdx.x % kNumSums;
    int rl = threadIdx.x / kNumSums;  // 0..7
    double v = 0;
    for (int r = rl; r < n_rows; r += 8) v += partials[(int64_t)r * kNumSums + col];
    lds[rl][col] = v;
    __syncthreads();
    if (threadIdx.x < n_out) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds[k][threadIdx.x];
        if (out) out[threadIdx.x] = s;
        if (mail_data) mail_data[threadIdx.x] = s;
    }
    // Host mailbox (mailbox.h): the driver spins on the sequence word instead
    // of a copy + stream synchronise per iteration.
    if (mail_flag) MailboxPublish(mail_flag, mail_seq);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
P2PlaneAccumulateKernel(const T* __restrict__ src, const T* __restrict__ tgt,
                        const T* __restrict__ tgt_n,
                        const int64_t* __restrict__ corr, int64_t n,
                        RobustParams rp, double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateP2Plane<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                             tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2],
                             tgt_n[3 * c + 0], tgt_n[3 * c + 1],
                             tgt_n[3 * c + 2], rp);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
SymmetricAccumulateKernel(const T* __restrict__ src, const T* __restrict__ src_n,
                          const T* __restrict__ tgt, const T* __restrict__ tgt_n,
                          const int64_t* __restrict__ corr, int64_t n,
                          Means3<T> mean, RobustParams rp,
                          double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateSymmetric<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                               tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2],
                               src_n[3 * i + 0], src_n[3 * i + 1],
                               src_n[3 * i + 2], tgt_n[3 * c + 0],
                               tgt_n[3 * c + 1], tgt_n[3 * c + 2], mean, rp);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
ColoredAccumulateKernel(const T* __restrict__ src, const T* __restrict__ src_c,
                        const T* __restrict__ tgt, const T* __restrict__ tgt_n,
                        const T* __restrict__ tgt_c, const T* __restrict__ tgt_g,
                        const int64_t* __restrict__ corr, int64_t n,
                        T sqrt_lambda_geometric, T sqrt_lambda_photometric,
                        RobustParams rp, double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        const T vs[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        const T cs[3] = {src_c[3 * i], src_c[3 * i + 1], src_c[3 * i + 2]};
        const T vt[3] = {tgt[3 * c], tgt[3 * c + 1], tgt[3 * c + 2]};
        const T nt[3] = {tgt_n[3 * c], tgt_n[3 * c + 1], tgt_n[3 * c + 2]};
        const T ct[3] = {tgt_c[3 * c], tgt_c[3 * c + 1], tgt_c[3 * c + 2]};
        const T dit[3] = {tgt_g[3 * c], tgt_g[3 * c + 1], tgt_g[3 * c + 2]};
        AccumulateColored<T>(A, vs, cs, vt, nt, ct, dit, sqrt_lambda_geometric,
                             sqrt_lambda_photometric, rp);
    }
    BlockReduceAndStore(A, partials);
}

// TransformationEstimationPointTo{Plane,Point}::ComputeRMSE
// (TransformationEstimation.cpp:101-130,160-193): the tensor expressions are
// element-wise -- point-to-plane squares every COMPONENT of (s - t) * n, it is
// not the squared point-to-plane distance -- formed in T, summed in float64
// here. A[0] = sum, A[1] = number of correspondences.
template <typename T, bool PLANE>
__global__ void __launch_bounds__(kReduceBlock)
ResidualSquaresKernel(const T* __restrict__ src, const T* __restrict__ tgt,
                      const T* __restrict__ tgt_n,
                      const int64_t* __restrict__ corr, int64_t n,
                      double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T e = src[3 * i + k] - tgt[3 * c + k];
            if (PLANE) e = e * tgt_n[3 * c + k];
            A[0] += (double)(e * e);
        }
        A[1] += 1.0;
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
P2PointAccumulateKernel(const T* __restrict__ src, const T* __restrict__ tgt,
                        const int64_t* __restrict__ corr, int64_t n,
                        double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateP2Point<T>(A, src[3 * i + 0], src[3 * i + 1], src[3 * i + 2],
                             tgt[3 * c + 0], tgt[3 * c + 1], tgt[3 * c + 2]);
    }
    BlockReduceAndStore(A, partials);
}

template <typename T>
__global__ void __launch_bounds__(kReduceBlock)
InformationAccumulateKernel(const T* __restrict__ tgt,
                            const int64_t* __restrict__ corr, int64_t n,
                            double* __restrict__ partials) {
    double A[kNumSums];
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) A[k] = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t c = corr[i];
        if (c == -1) continue;
        AccumulateInformation<T>(A, tgt[3 * c + 0], tgt[3 * c + 1],
                                 tgt[3 * c + 2]);
    }
    BlockReduceAndStore(A, partials);
}

// TransformImpl.h:19-44
template <typename T>
struct Mat4 { T m[16]; };

// Fused (transform +) search + accumulate. The 27 neighbour cells of a query
// are independent look-ups (bucket range -> a handful of candidate records);
// walking them one after the other from a single lane is a chain of ~54
// dependent memory round trips. Here G lanes (1, 2, 4 ... 32) share a query:
// lane c scans cells c, c + G, ..., the group takes the minimum by (d2,
// original index) -- the same winner the sequential scan picks.
//
// A pass is a short chain of memory round trips per query whatever the bucket
// sizes: (1) the query, (2) the ranges of all cells a lane owns, (3) their
// records taken as one list, kFlight in flight together (a scanned surface at
// the ICP radii has ~85 candidates per query, i.e. 85 / G per lane), (4) the
// winner's record and normal. G is chosen so that all queries are in flight
// at once (n G / 64 waves <= what the chip holds).
//
// The sums: every lane of the group forms the winner's terms (same inputs ->
// same values) and keeps only the 32 / G of them it owns, term c + m * G in
// register m -- so the running sums cost 2 * 32 / G registers instead of 64,
// which is what bounds how many waves a SIMD holds while they wait on memory.
// At the end the groups of a wave are added by xor shuffles, the waves through
// LDS, one row per workgroup; the geometry is a function of (n, G) only, so
// the result is run-to-run identical.
//
// apply_xf: the source point is first moved by `xf` exactly like
// TransformPointsKernel would and stored back (the ICP driver's per-iteration
// `source.Transform(update)` rides in the next search launch).
// EST: 0 = point-to-plane terms (needs sorted normals), 1 = point-to-point
// moments, 2 = information-matrix terms of the matched target point, 3 =
// point-to-plane with L2Loss (weight 1).
constexpr int kSearchBlock = 512;  // 8 waves; <= 512 rows for the final pass

// The sums a lane of a G-lane group owns: term i lives in lane i % G, register
// i / G. `S[i] += v` (i a compile-time constant once the callers' loops are
// unrolled) adds v there and nothing (+0.0) elsewhere, so a term is consumed
// the moment it is formed -- no 32-value temporary.
template <int G>
struct OwnedSums {
    double (&mine)[kNumSums / G];
    int c0;
    struct Ref {
        OwnedSums& o;
        int i;
        __device__ __forceinline__ void operator+=(double v) {
            o.mine[i / G] += (i % G) == o.c0 ? v : 0.0;
        }
        // a float term is picked as a float and widened once
        __device__ __forceinline__ void operator+=(float v) {
            o.mine[i / G] += (double)((i % G) == o.c0 ? v : 0.0f);
        }
    };
    __device__ __forceinline__ Ref operator[](int i) { return Ref{*this, i}; }
};

// ---- the final sum inside the search launch -----------------------------------
// What used to follow every search launch -- a one-workgroup final-sum launch
// (6.2 us + the launch boundary, 14 times per tracking frame) -- is done by the
// LAST workgroup of the search launch itself:
//   * every workgroup writes its row of partial sums with write-through
//     (sc1) stores, drains them (s_waitcnt vmcnt(0)) and takes a ticket; the
//     tickets are two-level, one counter per XCD class (blockIdx % 8) and one
//     on top, so no counter sees more than 64 arrivals (a single word
//     serialises them at ~12 ns each);
//   * the last arriver reads all rows with sc1 loads (valid without an acquire
//     fence because the producers stored sc1), adds them in FinalSumKernel's
//     order (bit-identical sums) and posts them to the host mailbox (and / or
//     a device buffer).
// No agent-scope release fence anywhere: on this part it writes back the
// XCD's whole L2 (the moved source points of the launch are dirty in it) and
// costs more than the launch it would save -- measured in round 1.
// (Rounds 3-4 also solved the 6x6 system and formed the update in that last
// workgroup, the host one launch ahead: +8-10 us per search launch, what the
// host hop it removed cost -- and a variant whose NEXT launch was queued ahead
// and polled a host inbox: 2x slower. Both dropped; docs/rounds.md.)
constexpr int kTailRows = 512;  // workgroups of a launch that carries a tail
struct SumTail {
    int* tickets;      // [9] zero between launches; NULL = no tail
    double* out;       // device [32] or NULL
    double* mail_data; // host-mapped [32] or NULL
    int* mail_flag;
    int mail_seq;
};

__device__ __forceinline__ double LoadSc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(
            (const unsigned long long*)p, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void StoreSc1(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p,
                       (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// True in every thread of the workgroup that arrived last. Call after the
// workgroup's row is stored (by threads of wave 0) -- see above.
__device__ __forceinline__ bool LastWorkgroup(int* tickets) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        const int cls = blockIdx.x & 7;
        const int in_class = ((int)gridDim.x - cls + 7) >> 3;
        const int classes = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
        int last = 0;
        if (__hip_atomic_fetch_add(&tickets[1 + cls], 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) == in_class - 1)
            last = __hip_atomic_fetch_add(&tickets[0], 1, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT) ==
                   classes - 1;
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// Executed by all kSearchBlock threads of the last workgroup.
__device__ __forceinline__ void RowSumTail(const double* partials, int n_rows,
                                           const SumTail& tl) {
    static_assert(kSearchBlock == 512 && kNumSums == 32, "tail geometry");
    __shared__ double s_rows[kFinalRowLanes][32];
    const int tid = threadIdx.x;
    // FinalSumKernel's order: row lane rl adds rows rl, rl + 32, ... in
    // ascending order; the 32 row lanes are then added in ascending order.
    // 512 threads: thread (rl0, col) runs row lanes rl0 and rl0 + 16, eight
    // loads of a row lane in flight at a time (<= kTailRows rows: at most two
    // rounds).
    {
        const int col = tid & 31, rl0 = tid >> 5;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rl = rl0 + 16 * h;
            double v = 0;
            for (int r0 = rl; r0 < n_rows; r0 += 8 * kFinalRowLanes) {
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = r0 + k * kFinalRowLanes;
                    x[k] = r < n_rows
                                   ? LoadSc1(partials + (int64_t)r * 32 + col)
                                   : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v += x[k];
            }
            s_rows[rl][col] = v;
        }
    }
    if (tid == 0) {
        // tickets back to zero for the next launch (ordered by the kernel
        // boundary)
#pragma unroll
        for (int k = 0; k < 9; ++k) tl.tickets[k] = 0;
    }
    __syncthreads();
    if (tid < 64) {
        double t = 0;
        if (tid < 32) {
#pragma unroll
            for (int k = 0; k < kFinalRowLanes; ++k) t += s_rows[k][tid];
            if (tl.out) tl.out[tid] = t;
        }
        if (tl.mail_data) {
            // SEALED post (mailbox.h): no system-scope release fence -- inside
            // this launch it would write back the XCD's whole L2, dirty with
            // the moved source points. The 32 values and a seal word (their
            // XOR mixed with the sequence number) go out as write-through
            // system-scope stores; the host accepts the block only when the
            // seal fits, whatever order the stores land in.
            unsigned long long b =
                    tid < 32 ? (unsigned long long)__double_as_longlong(t)
                             : 0ull;
            unsigned long long x = b;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) x ^= __shfl_xor(x, d, 64);
            if (tid < 32)
                __hip_atomic_store((unsigned long long*)tl.mail_data + tid, b,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid == 32)
                __hip_atomic_store((unsigned long long*)tl.mail_data + 32,
                                   x ^ MailSeal(tl.mail_seq), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0)
                __hip_atomic_store(tl.mail_flag, tl.mail_seq, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T, int G, int EST>
__global__ void __launch_bounds__(kSearchBlock,
                                  EST == 0 ? 2 : (G >= 8 ? 4 : (G >= 2 ? 3 : 2)))
SearchAccumulateKernel(NnsView<T> nv, const Rec4<T>* __restrict__ sorted_n,
                       T* __restrict__ src, int64_t n, Mat4<T> xf,
                       int apply_xf, RobustParams rp,
                       int64_t* __restrict__ corr_out,
                       double* __restrict__ partials, SumTail tail) {
    constexpr int kPerWave = 64 / G;  // queries per wave
    constexpr int kM = kNumSums / G;  // sums a lane owns
    static_assert(kNumSums % G == 0, "G divides the number of sums");
    double mine[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) mine[m] = 0;
    const int c0 = threadIdx.x & (G - 1);  // first neighbour cell of this lane
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int sub = (threadIdx.x & 63) / G;
    constexpr int kOwn = (27 + G - 1) / G;        // cells a lane owns
    constexpr int kBatch = kOwn < 9 ? kOwn : 9;   // cells per batch
    // candidate records in flight per lane and round trip
    constexpr int kFlight = (sizeof(T) == 4 ? 16 : 8) / (G >= 16 ? 2 : 1);
    for (int64_t base = wave * kPerWave; base < n; base += n_waves * kPerWave) {
        const int64_t i = base + sub;
        const bool valid = i < n;
        T q[3] = {T(0), T(0), T(0)};
        int pos = -1, idx = -1;
        T d2 = T(0);
        if (valid) {
            q[0] = src[3 * i + 0];
            q[1] = src[3 * i + 1];
            q[2] = src[3 * i + 2];
            if (apply_xf) {
                // TransformPointsKernel, statement by statement
                const T p0 = q[0], p1 = q[1], p2 = q[2];
                const T x0 = xf.m[0] * p0 + xf.m[1] * p1 + xf.m[2] * p2 + xf.m[3];
                const T x1 = xf.m[4] * p0 + xf.m[5] * p1 + xf.m[6] * p2 + xf.m[7];
                const T x2 = xf.m[8] * p0 + xf.m[9] * p1 + xf.m[10] * p2 + xf.m[11];
                const T x3 = xf.m[12] * p0 + xf.m[13] * p1 + xf.m[14] * p2 + xf.m[15];
                q[0] = x0 / x3;
                q[1] = x1 / x3;
                q[2] = x2 / x3;
                if (c0 == 0) {
                    src[3 * i + 0] = q[0];
                    src[3 * i + 1] = q[1];
                    src[3 * i + 2] = q[2];
                }
            }
            long long cx, cy, cz;
            CellOf(q, nv.inv_cell, cx, cy, cz);
            // records by 32-bit byte offset off one scalar base (the index
            // holds < 2^27 points): half the address registers per load in
            // flight
            auto record = [&](unsigned j) -> Rec4<T> {
                const unsigned off = j * (unsigned)sizeof(Rec4<T>);
                return *(const Rec4<T>*)((const char*)nv.sorted + off);
            };
            // branch-free; `live` = entry t of the list exists. The winner is
            // remembered by its list entry and turned into a record position
            // once per batch of cells. Float32: (d2, index) compare as ONE
            // 64-bit key -- a non-negative float orders like its bit pattern
            // -- so "nearer, ties to the lower index" is a single compare.
            bool have = false;
            unsigned best_t = 0;
            unsigned long long best_key = ~0ull;
            auto consider = [&](const Rec4<T>& p, unsigned t, bool live) {
                T result = T(0);
                const T d0 = q[0] - p.x;
                result += d0 * d0;
                const T d1 = q[1] - p.y;
                result += d1 * d1;
                const T dd = q[2] - p.z;
                result += dd * dd;
                const int pi = RecIndex(p);
                if constexpr (sizeof(T) == 4) {
                    const unsigned long long key =
                            ((unsigned long long)__float_as_uint((float)result)
                             << 32) |
                            (unsigned)pi;
                    const bool better = live && result < nv.radius_squared &&
                                        key < best_key;
                    best_t = better ? t : best_t;
                    best_key = better ? key : best_key;
                } else {
                    const bool better =
                            live && result < nv.radius_squared &&
                            (!have || result < d2 ||
                             (result == d2 && pi < idx));
                    best_t = better ? t : best_t;
                    idx = better ? pi : idx;
                    d2 = better ? result : d2;
                    have = have || better;
                }
            };
            // One batch of cells: the ranges of the lane's cells (one round
            // trip), then their records as ONE list: entry t of the list is
            // record t - first[k] of the cell k it falls into, so a round trip
            // fetches kFlight candidates whatever the bucket sizes are.
            // bucket[k] / use[k]: the cells of the batch and which of them
            // count (an unused cell's range is read and dropped).
            auto scan_batch = [&](auto n_tag, const unsigned* bucket,
                                  const bool* use) {
                constexpr int kN = decltype(n_tag)::value;
                unsigned s0[kN], first[kN + 1];
                first[0] = 0;
#pragma unroll
                for (int k = 0; k < kN; ++k) {
                    unsigned e;
                    BucketRange(nv, bucket[k], s0[k], e);
                    first[k + 1] = first[k] + (use[k] ? e - s0[k] : 0u);
                }
                const unsigned total = first[kN];
                // entry t of the list is record t + off[k] of the array, k the
                // cell it falls into
                // (as a running sum of the offsets' differences: a chain of
                // plain "pick off[k]" selects is turned into an indexed read
                // of an LDS copy of the array, one LDS round trip in front of
                // every record load)
                unsigned step[kN];
                step[0] = s0[0];  // first[0] == 0
#pragma unroll
                for (int k = 1; k < kN; ++k)
                    step[k] = (s0[k] - first[k]) - (s0[k - 1] - first[k - 1]);
                auto position = [&](unsigned t) {
                    unsigned o = step[0];
#pragma unroll
                    for (int k = 1; k < kN; ++k)
                        o += t >= first[k] ? step[k] : 0u;
                    return t + o;
                };
                const unsigned long long key_before = best_key;
                const int idx_before = idx;
                const bool had = have;
                for (unsigned t0 = 0; t0 < total; t0 += kFlight) {
                    Rec4<T> cand[kFlight];
#pragma unroll
                    for (int u = 0; u < kFlight; ++u) {
                        const unsigned t = t0 + u;
                        cand[u] = record(t < total ? position(t) : 0u);
                    }
#pragma unroll
                    for (int u = 0; u < kFlight; ++u)
                        consider(cand[u], t0 + u, t0 + u < total);
                }
                // a new winner out of this batch of cells?
                if constexpr (sizeof(T) == 4) {
                    if (best_key != key_before) {
                        pos = (int)position(best_t);
                        idx = (int)(unsigned)best_key;
                        d2 = (T)__uint_as_float((unsigned)(best_key >> 32));
                    }
                } else {
                    if (have && (!had || idx != idx_before))
                        pos = (int)position(best_t);
                }
            };
            if constexpr (G == 8) {
                // NEAREST neighbour only (max_knn = 1): most of the 27 cells
                // cannot hold it. Phase 1: the 2 x 2 x 2 cells on the query's
                // side of its own cell -- one per lane -- where the nearest
                // point of a sampled surface almost always lies. Phase 2: of
                // the other 19 cells only those whose box comes closer to the
                // query than the best so far (or than the radius, if nothing
                // was found); skipped by the whole wave when no lane needs
                // it. A skipped cell lies strictly farther than the winner:
                // the winner by (d2, index) is the one the full scan finds,
                // bit for bit. (Round 6: 85 -> ~37 candidate records and 27
                // -> 8 ranges per query on the finest scale of a tracked
                // frame, where the launch is bound by the per-CU vector
                // memory path.)
                const double ux = (double)q[0] * nv.inv_cell - (double)cx;
                const double uy = (double)q[1] * nv.inv_cell - (double)cy;
                const double uz = (double)q[2] * nv.inv_cell - (double)cz;
                const int sx = ux >= 0.5 ? 1 : -1, sy = uy >= 0.5 ? 1 : -1,
                          sz = uz >= 0.5 ? 1 : -1;
                {
                    const unsigned b1[1] = {
                            HashCell(cx + ((c0 & 1) ? sx : 0),
                                     cy + ((c0 & 2) ? sy : 0),
                                     cz + ((c0 & 4) ? sz : 0)) &
                            nv.mask};
                    const bool u1[1] = {true};
                    scan_batch(std::integral_constant<int, 1>(), b1, u1);
                }
                // the group's best so far (every lane of a group is here)
                T bound = nv.radius_squared;
                {
                    int gp = pos, gi = idx;
                    T gd = d2;
#pragma unroll
                    for (int m = G / 2; m > 0; m >>= 1) {
                        const int opos = __shfl_xor(gp, m, G);
                        const int oidx = __shfl_xor(gi, m, G);
                        const T od2 = __shfl_xor(gd, m, G);
                        const bool take =
                                opos >= 0 && (gp < 0 || od2 < gd ||
                                              (od2 == gd && oidx < gi));
                        if (take) {
                            gp = opos;
                            gi = oidx;
                            gd = od2;
                        }
                    }
                    if (gp >= 0) bound = gd;
                }
                // distance (in cells) from the query to the cell beside its
                // own, per axis and side; a hair under the true one, so that
                // rounding can only make a cell count, never drop it
                const double cell = 1.0 / nv.inv_cell;
                const double bound_c =
                        (double)bound * nv.inv_cell * nv.inv_cell;
                (void)cell;
                unsigned b2[kOwn];
                bool u2[kOwn];
                bool any = false;
#pragma unroll
                for (int k = 0; k < kOwn; ++k) {
                    const int c = c0 + k * G;
                    const int cc = c < 27 ? c : 26;
                    const int dz = cc / 9 - 1, dy = (cc % 9) / 3 - 1,
                              dx = cc % 3 - 1;
                    const bool in_octant = (dx == 0 || dx == sx) &&
                                           (dy == 0 || dy == sy) &&
                                           (dz == 0 || dz == sz);
                    const double ax = dx == 0 ? 0.0 : (dx > 0 ? 1.0 - ux : ux);
                    const double ay = dy == 0 ? 0.0 : (dy > 0 ? 1.0 - uy : uy);
                    const double az = dz == 0 ? 0.0 : (dz > 0 ? 1.0 - uz : uz);
                    const double lb = (ax * ax + ay * ay + az * az) * 0.9999;
                    u2[k] = c < 27 && !in_octant && lb <= bound_c;
                    any = any || u2[k];
                    b2[k] = HashCell(cx + dx, cy + dy, cz + dz) & nv.mask;
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0ull)
                    scan_batch(std::integral_constant<int, kOwn>(), b2, u2);
            } else {
                for (int cb = 0; cb < kOwn; cb += kBatch) {
                    unsigned bucket[kBatch];
                    bool use[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        // cells past the 27th: look at cell 26 again, keep
                        // nothing
                        const int c = c0 + (cb + k) * G;
                        const int cc = c < 27 ? c : 26;
                        const int dz = cc / 9 - 1, dy = (cc % 9) / 3 - 1,
                                  dx = cc % 3 - 1;
                        bucket[k] = HashCell(cx + dx, cy + dy, cz + dz) &
                                    nv.mask;
                        use[k] = c < 27;
                    }
                    scan_batch(std::integral_constant<int, kBatch>(), bucket,
                               use);
                }
            }
        }
        // minimum by (d2, idx) over the G lanes of the group
#pragma unroll
        for (int m = G / 2; m > 0; m >>= 1) {
            const int opos = __shfl_xor(pos, m, G);
            const int oidx = __shfl_xor(idx, m, G);
            const T od2 = __shfl_xor(d2, m, G);
            const bool take = opos >= 0 &&
                              (pos < 0 || od2 < d2 || (od2 == d2 && oidx < idx));
            if (take) {
                pos = opos;
                idx = oidx;
                d2 = od2;
            }
        }
        if (valid && c0 == 0 && corr_out)
            corr_out[i] = pos >= 0 ? (int64_t)idx : (int64_t)-1;
        if (pos >= 0) {  // the whole group agrees
            const Rec4<T> t = nv.sorted[pos];
            OwnedSums<G> S{mine, c0};
            if constexpr (EST == 0 || EST == 3) {
                const Rec4<T> nn = sorted_n[pos];
                AccumulateP2Plane<T, OwnedSums<G>, EST == 3>(
                        S, q[0], q[1], q[2], t.x, t.y, t.z, nn.x, nn.y, nn.z,
                        rp);
            } else if constexpr (EST == 1) {
                AccumulateP2Point<T>(S, q[0], q[1], q[2], t.x, t.y, t.z);
            } else {
                AccumulateInformation<T>(S, t.x, t.y, t.z);
            }
            S[29] += (double)d2;
            S[30] += 1.0;
        }
    }
    // groups of the wave, then the waves of the workgroup
    __shared__ double lds[kSearchBlock / 64][kNumSums];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if constexpr (G == 1) {
        const double t = WaveReduceScatter<kNumSums>(mine);
        if ((lane & 1) == 0) lds[wv][lane >> 1] = t;
    } else {
#pragma unroll
        for (int m = 0; m < kM; ++m) {
#pragma unroll
            for (int sft = G; sft < 64; sft <<= 1)
                mine[m] += __shfl_xor(mine[m], sft, 64);
            if (lane < G) lds[wv][c0 + m * G] = mine[m];
        }
    }
    __syncthreads();
    if (threadIdx.x < kNumSums) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < kSearchBlock / 64; ++w) t += lds[w][threadIdx.x];
        if (tail.tickets)
            StoreSc1(partials + (int64_t)blockIdx.x * kNumSums + threadIdx.x,
                     t);
        else
            partials[(int64_t)blockIdx.x * kNumSums + threadIdx.x] = t;
    }
#if O3DMI_ABLATE_TAIL == 1
    // (variant builds only, tools/build_variant.sh-style: what the final sum
    // inside the launch costs -- sums are NOT delivered)
#elif O3DMI_ABLATE_TAIL == 2
    if (tail.tickets) (void)LastWorkgroup(tail.tickets);
#else
    if (tail.tickets && LastWorkgroup(tail.tickets))
        RowSumTail(partials, (int)gridDim.x, tail);
#endif
}


template <typename T>
__global__ void TransformPointsKernel(Mat4<T> t, T* __restrict__ pts,
                                      int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T* p = pts + 3 * i;
        T p0 = p[0], p1 = p[1], p2 = p[2];
        T x0 = t.m[0] * p0 + t.m[1] * p1 + t.m[2] * p2 + t.m[3];
        T x1 = t.m[4] * p0 + t.m[5] * p1 + t.m[6] * p2 + t.m[7];
        T x2 = t.m[8] * p0 + t.m[9] * p1 + t.m[10] * p2 + t.m[11];
        T x3 = t.m[12] * p0 + t.m[13] * p1 + t.m[14] * p2 + t.m[15];
        p[0] = x0 / x3;
        p[1] = x1 / x3;
        p[2] = x2 / x3;
    }
}
// TransformImpl.h:46-60
template <typename T>
__global__ void TransformNormalsKernel(Mat4<T> t, T* __restrict__ nrm,
                                       int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        T* p = nrm + 3 * i;
        T p0 = p[0], p1 = p[1], p2 = p[2];
        T x0 = t.m[0] * p0 + t.m[1] * p1 + t.m[2] * p2;
        T x1 = t.m[4] * p0 + t.m[5] * p1 + t.m[6] * p2;
        T x2 = t.m[8] * p0 + t.m[9] * p1 + t.m[10] * p2;
        p[0] = x0;
        p[1] = x1;
        p[2] = x2;
    }
}

int ReduceGrid(int64_t n) {
    int64_t g = (n + kReduceBlock - 1) / kReduceBlock;
    // Two workgroups per CU at most: enough waves to hide the gather latency,
    // and the final pass (a single workgroup, linear in the row count) reads
    // <= 512 partial rows.
    int64_t cap = (int64_t)kCUs * 2;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace
// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadIcp() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &TransformNormalsKernel<float>)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi

using namespace o3dmi;

extern "C" int o3dmi_nns_set_normals(o3dmi_nns_t* nns, const void* normals_dev,
                                     o3dmi_stream_t stream);

extern "C" {

int o3dmi_icp_p2plane_accumulate(const void* src_dev, const void* tgt_dev,
                                 const void* tgt_normals_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 int robust_kernel, double scaling_parameter,
                                 double shape_parameter, double* sums29_dev,
                                 o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && tgt_dev && tgt_normals_dev && corr_dev &&
                          sums29_dev,
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(P2PlaneAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev, corr_dev, n, rp,
                           partials);
    else
        hipLaunchKernelGGL(P2PlaneAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)tgt_dev, (const float*)tgt_normals_dev,
                           corr_dev, n, rp, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

// Internal (host header: o3dmi_registration_compute_rmse): sums2_dev[0] = sum
// of squared residual components, [1] = number of correspondences.
int o3dmi_icp_residual_squares(const void* src_dev, const void* tgt_dev,
                               const void* tgt_normals_dev,
                               const int64_t* corr_dev, int64_t n, int dtype,
                               double* sums2_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && tgt_dev && corr_dev && sums2_dev, "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
#define O3DMI_RESID(T, P)                                                      \
    hipLaunchKernelGGL((ResidualSquaresKernel<T, P>), dim3(g),                 \
                       dim3(kReduceBlock), 0, s, (const T*)src_dev,            \
                       (const T*)tgt_dev, (const T*)tgt_normals_dev, corr_dev, \
                       n, partials)
    if (dtype == O3DMI_F64) {
        if (tgt_normals_dev) O3DMI_RESID(double, true);
        else O3DMI_RESID(double, false);
    } else {
        if (tgt_normals_dev) O3DMI_RESID(float, true);
        else O3DMI_RESID(float, false);
    }
#undef O3DMI_RESID
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums2_dev, 2, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_p2point_accumulate(const void* src_dev, const void* tgt_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 double* sums16_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && tgt_dev && corr_dev && sums16_dev,
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(P2PointAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)tgt_dev, corr_dev, n, partials);
    else
        hipLaunchKernelGGL(P2PointAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)tgt_dev, corr_dev, n, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums16_dev, 16, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

// Internal: also posts the 29 sums to a host mailbox when mail_data != NULL.
int o3dmi_icp_colored_accumulate_post(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        double* partials_dev, double* mail_data, int* mail_flag, int mail_seq,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && src_colors_dev && tgt_dev && tgt_normals_dev &&
                          tgt_colors_dev && tgt_color_gradients_dev &&
                          corr_dev && (sums29_dev || mail_data),
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    O3DMI_REQUIRE(lambda_geometric >= 0 && lambda_geometric <= 1.0,
                  "lambda_geometric must be in [0, 1]");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = partials_dev;
    if (!partials)
        O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                       sizeof(double) * (size_t)g * kNumSums,
                                       s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    // ComputePoseColoredICPCPU, RegistrationCPU.cpp:310-313
    const double slg = std::sqrt(lambda_geometric);
    const double slp = std::sqrt(1.0 - lambda_geometric);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(ColoredAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)src_colors_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev,
                           (const double*)tgt_colors_dev,
                           (const double*)tgt_color_gradients_dev, corr_dev, n,
                           slg, slp, rp, partials);
    else
        hipLaunchKernelGGL(ColoredAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)src_colors_dev, (const float*)tgt_dev,
                           (const float*)tgt_normals_dev,
                           (const float*)tgt_colors_dev,
                           (const float*)tgt_color_gradients_dev, corr_dev, n,
                           (float)slg, (float)slp, rp, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, mail_data, mail_flag, mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    if (!partials_dev) O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_colored_accumulate(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums29_dev != nullptr, "null argument");
    return o3dmi_icp_colored_accumulate_post(
            src_dev, src_colors_dev, tgt_dev, tgt_normals_dev, tgt_colors_dev,
            tgt_color_gradients_dev, corr_dev, n, dtype, lambda_geometric,
            robust_kernel, scaling_parameter, shape_parameter, sums29_dev,
            nullptr, nullptr, nullptr, 0, stream);
}

// Internal: also posts the 29 sums to a host mailbox when mail_data != NULL.
int o3dmi_icp_symmetric_accumulate_post(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, double* partials_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(src_dev && src_normals_dev && tgt_dev && tgt_normals_dev &&
                          corr_dev && source_mean3 && target_mean3 &&
                          (sums29_dev || mail_data),
                  "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = partials_dev;
    if (!partials)
        O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                       sizeof(double) * (size_t)g * kNumSums,
                                       s));
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    if (dtype == O3DMI_F64) {
        Means3<double> m;
        for (int k = 0; k < 3; ++k) {
            m.s[k] = source_mean3[k];
            m.t[k] = target_mean3[k];
        }
        hipLaunchKernelGGL(SymmetricAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)src_dev,
                           (const double*)src_normals_dev,
                           (const double*)tgt_dev,
                           (const double*)tgt_normals_dev, corr_dev, n, m, rp,
                           partials);
    } else {
        Means3<float> m;
        for (int k = 0; k < 3; ++k) {
            m.s[k] = (float)source_mean3[k];
            m.t[k] = (float)target_mean3[k];
        }
        hipLaunchKernelGGL(SymmetricAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)src_dev,
                           (const float*)src_normals_dev, (const float*)tgt_dev,
                           (const float*)tgt_normals_dev, corr_dev, n, m, rp,
                           partials);
    }
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums29_dev, 29, mail_data, mail_flag, mail_seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    if (!partials_dev) O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_symmetric_accumulate(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums29_dev != nullptr, "null argument");
    return o3dmi_icp_symmetric_accumulate_post(
            src_dev, src_normals_dev, tgt_dev, tgt_normals_dev, corr_dev, n,
            dtype, source_mean3, target_mean3, robust_kernel, scaling_parameter,
            shape_parameter, sums29_dev, nullptr, nullptr, nullptr, 0, stream);
}

int o3dmi_icp_information_accumulate(const void* tgt_dev,
                                     const int64_t* corr_dev, int64_t n,
                                     int dtype, double* sums21_dev,
                                     o3dmi_stream_t stream) {
    O3DMI_REQUIRE(tgt_dev && corr_dev && sums21_dev, "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    hipStream_t s = (hipStream_t)stream;
    int g = ReduceGrid(n);
    double* partials = nullptr;
    O3DMI_HIP_CHECK(hipMallocAsync((void**)&partials,
                                   sizeof(double) * (size_t)g * kNumSums, s));
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(InformationAccumulateKernel<double>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const double*)tgt_dev,
                           corr_dev, n, partials);
    else
        hipLaunchKernelGGL(InformationAccumulateKernel<float>, dim3(g),
                           dim3(kReduceBlock), 0, s, (const float*)tgt_dev,
                           corr_dev, n, partials);
    hipLaunchKernelGGL(FinalReduceKernel, dim3(1), dim3(256), 0, s, partials, g,
                       sums21_dev, 21, (double*)nullptr, (int*)nullptr, 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipFreeAsync(partials, s));
    return O3DMI_OK;
}

int o3dmi_icp_search_accumulate_post(
        const o3dmi_nns_t* nns, const void* src_dev,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);

int o3dmi_internal_icp_transform_search_accumulate(
        const o3dmi_nns_t* nns, void* src_dev, const double* transformation,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream);

int o3dmi_icp_search_accumulate_p2point(const o3dmi_nns_t* nns,
                                        const void* src_dev, int64_t n,
                                        int64_t* corr_out_dev,
                                        double* sums32_dev,
                                        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums32_dev != nullptr, "null argument");
    return o3dmi_icp_search_accumulate_post(nns, src_dev, nullptr, n, 1, 0, 1.0,
                                            1.0, corr_out_dev, sums32_dev,
                                            nullptr, nullptr, 0, stream);
}

int o3dmi_icp_search_accumulate(const o3dmi_nns_t* nns, const void* src_dev,
                                const void* tgt_normals_dev, int64_t n,
                                int robust_kernel, double scaling_parameter,
                                double shape_parameter, int64_t* corr_out_dev,
                                double* sums32_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(sums32_dev != nullptr, "null argument");
    return o3dmi_icp_search_accumulate_post(
            nns, src_dev, tgt_normals_dev, n, 0, robust_kernel,
            scaling_parameter, shape_parameter, corr_out_dev, sums32_dev,
            nullptr, nullptr, 0, stream);
}

// Internal (not in the public header): also posts the 32 sums to a host
// mailbox (mailbox.h) when mail_data != NULL; sums32_dev may then be NULL.
int o3dmi_icp_search_accumulate_post(
        const o3dmi_nns_t* nns, const void* src_dev,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    return o3dmi_internal_icp_transform_search_accumulate(
            nns, const_cast<void*>(src_dev), nullptr, tgt_normals_dev, n,
            estimation, robust_kernel, scaling_parameter, shape_parameter,
            corr_out_dev, sums32_dev, mail_data, mail_flag, mail_seq, stream);
}

// Internal: as above, and when `transformation` (row-major 4x4, float64) is
// given the source points are first moved by it IN PLACE, with
// o3dmi_transform_points' arithmetic, inside the same launch.
static int LaunchSearchAccumulate(
        const o3dmi_nns_t* nns, void* src_dev, const double* transformation,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(nns && src_dev && (sums32_dev || mail_data),
                  "null argument");
    O3DMI_REQUIRE(estimation >= 0 && estimation <= 2,
                  "estimation must be point-to-plane (0), point-to-point (1) "
                  "or information (2)");
    O3DMI_REQUIRE(robust_kernel >= 0 && robust_kernel <= 6,
                  "Unsupported method.");
    hipStream_t s = (hipStream_t)stream;
    if (estimation == 0) {
        if (tgt_normals_dev) {
            int st = o3dmi_nns_set_normals(const_cast<o3dmi_nns_t*>(nns),
                                           tgt_normals_dev, stream);
            if (st != O3DMI_OK) return st;
        }
        O3DMI_REQUIRE(nns->sorted_normals != nullptr,
                      "Target pointcloud missing normals attribute.");
    }
    // (Round 6: with the 8-lane form's pruned scan, 8 lanes at EVERY scale was
    // tried -- 12.6 / 13.5 us per coarse launch against 11.6 / 13.8: no gain,
    // the coarse scales are bound by the chain of trips, not by the scan.)
    // Lanes per query, from the launch timings of tools/bench_search.py on
    // the levels of a tracking frame and on the raw clouds (2 k ... 230 k
    // queries, profiles/r2w_search_*.json): G = 32 only while it leaves half
    // the chip free (a lane then owns one cell, mostly an empty one); G = 16
    // while that puts every query in flight at once (n G / 64 waves against
    // the 4 waves per SIMD the kernel's registers allow); G = 8 beyond --
    // several rounds of waves whose lanes own 4 cells each beat one round of
    // lanes that own 7, 14 or 27.
    const int64_t lane_scale = (int64_t)kCUs * 1024;  // 4 waves per SIMD
    int group;
    if (n * 32 <= lane_scale / 2) group = 32;
    else if (n * 16 <= lane_scale) group = 16;
    else group = 8;
    // The final sum rides in the launch's last workgroup (SumTail, <= 512
    // workgroups of 8 waves). Against the separate one-workgroup final-sum
    // launch of rounds 1-4 (examples/icp_slam, same box, gpurun r5c): 64 -> 53
    // launches per VGA frame, 1265 -> 1337 frames/s (mean of 3), level with
    // it at 1280x720 while the post still carried a system-scope fence.
    O3DMI_REQUIRE(nns->tickets != nullptr, "index without ticket words");
    int64_t g64 = (n * group + kSearchBlock - 1) / kSearchBlock;
    const int64_t g_max = kTailRows;
    static_assert(kTailRows <= kCUs * 4 - 1, "rows of nns->partials");
    if (g64 > g_max) g64 = g_max;
    if (g64 < 1) g64 = 1;
    const int g = (int)g64;
    RobustParams rp = MakeRobust(robust_kernel, scaling_parameter,
                                 shape_parameter);
    const int apply_xf = transformation != nullptr ? 1 : 0;
    SumTail tail = {};
    tail.tickets = nns->tickets;
    tail.out = sums32_dev;
    tail.mail_data = mail_data;
    tail.mail_flag = mail_flag;
    tail.mail_seq = mail_seq;
    Mat4<double> xd;
    Mat4<float> xfl;
    for (int k = 0; k < 16; ++k) {
        xd.m[k] = transformation ? transformation[k] : (k % 5 == 0 ? 1.0 : 0.0);
        xfl.m[k] = (float)xd.m[k];
    }
    auto xf_of = [&](auto tag) {
        if constexpr (sizeof(tag) == 8) return xd;
        else return xfl;
    };
#define O3DMI_SEARCH_E(T, G, E)                                                \
    hipLaunchKernelGGL((SearchAccumulateKernel<T, G, E>), dim3(g),            \
                       dim3(kSearchBlock), 0, s, MakeView<T>(nns),            \
                       (const Rec4<T>*)nns->sorted_normals, (T*)src_dev, n,   \
                       xf_of(T()), apply_xf, rp, corr_out_dev, nns->partials, \
                       tail)
#define O3DMI_SEARCH(T, G)                                                     \
    do {                                                                      \
        if (estimation == 0 && robust_kernel == O3DMI_L2_LOSS)                \
            O3DMI_SEARCH_E(T, G, 3);                                          \
        else if (estimation == 0) O3DMI_SEARCH_E(T, G, 0);                    \
        else if (estimation == 1) O3DMI_SEARCH_E(T, G, 1);                    \
        else O3DMI_SEARCH_E(T, G, 2);                                         \
    } while (0)
#define O3DMI_SEARCH_G(T)                                                      \
    switch (group) {                                                          \
        case 32: O3DMI_SEARCH(T, 32); break;                                  \
        case 16: O3DMI_SEARCH(T, 16); break;                                  \
        default: O3DMI_SEARCH(T, 8); break;                                   \
    }
    if (nns->dtype == O3DMI_F64) { O3DMI_SEARCH_G(double) }
    else { O3DMI_SEARCH_G(float) }
#undef O3DMI_SEARCH_G
#undef O3DMI_SEARCH
#undef O3DMI_SEARCH_E
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_internal_icp_transform_search_accumulate(
        const o3dmi_nns_t* nns, void* src_dev, const double* transformation,
        const void* tgt_normals_dev, int64_t n, int estimation,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        int64_t* corr_out_dev, double* sums32_dev, double* mail_data,
        int* mail_flag, int mail_seq, o3dmi_stream_t stream) {
    return LaunchSearchAccumulate(nns, src_dev, transformation, tgt_normals_dev,
                                  n, estimation, robust_kernel,
                                  scaling_parameter, shape_parameter,
                                  corr_out_dev, sums32_dev, mail_data, mail_flag,
                                  mail_seq, stream);
}

int o3dmi_transform_points(const double* transformation, void* points_dev,
                           int64_t n, int dtype, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(transformation && (points_dev || n == 0), "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64) {
        Mat4<double> t;
        for (int i = 0; i < 16; ++i) t.m[i] = transformation[i];
        hipLaunchKernelGGL(TransformPointsKernel<double>, grid, block, 0, s, t,
                           (double*)points_dev, n);
    } else {
        Mat4<float> t;
        for (int i = 0; i < 16; ++i) t.m[i] = (float)transformation[i];
        hipLaunchKernelGGL(TransformPointsKernel<float>, grid, block, 0, s, t,
                           (float*)points_dev, n);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_transform_normals(const double* transformation, void* normals_dev,
                            int64_t n, int dtype, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(transformation && (normals_dev || n == 0), "null argument");
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "normals must be Float32 or Float64");
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64) {
        Mat4<double> t;
        for (int i = 0; i < 16; ++i) t.m[i] = transformation[i];
        hipLaunchKernelGGL(TransformNormalsKernel<double>, grid, block, 0, s,
                           t, (double*)normals_dev, n);
    } else {
        Mat4<float> t;
        for (int i = 0; i < 16; ++i) t.m[i] = (float)transformation[i];
        hipLaunchKernelGGL(TransformNormalsKernel<float>, grid, block, 0, s, t,
                           (float*)normals_dev, n);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// TransformationConverter.cpp:189-226; the solve is LAPACK ?gesv restated
// (LU with partial pivoting), core/linalg/SolveCPU.cpp:15-30.
int o3dmi_decode_and_solve6x6(const double* A, double* pose6, float* residual,
                              int* inlier_count) {
    O3DMI_REQUIRE(A && pose6 && residual && inlier_count, "null argument");
    double M[36], b[6];
    for (int j = 0; j < 6; j++) {
        b[j] = -A[21 + j];
        const int reduction_idx = (j * (j + 1)) / 2;
        for (int k = 0; k <= j; k++) {
            M[j * 6 + k] = A[reduction_idx + k];
            M[k * 6 + j] = A[reduction_idx + k];
        }
    }
    const int n = 6;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double mx = std::fabs(M[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(M[i * n + k]);
            if (v > mx) { mx = v; p = i; }
        }
        if (mx == 0.0 || !(mx == mx)) {
            for (int j = 0; j < 6; ++j) pose6[j] = 0;
            *residual = 0;
            *inlier_count = 0;
            SetLastError("Singular 6x6 linear system detected, tracking failed.");
            return O3DMI_ERR_SINGULAR;
        }
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(M[k * n + j], M[p * n + j]);
            std::swap(b[k], b[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            double l = M[i * n + k] / M[k * n + k];
            M[i * n + k] = l;
            for (int j = k + 1; j < n; ++j) M[i * n + j] -= l * M[k * n + j];
        }
    }
    for (int i = 1; i < n; ++i)
        for (int j = 0; j < i; ++j) b[i] -= M[i * n + j] * b[j];
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j) b[i] -= M[i * n + j] * b[j];
        b[i] /= M[i * n + i];
    }
    for (int j = 0; j < 6; ++j) pose6[j] = b[j];
    *residual = (float)A[27];
    *inlier_count = (int)A[28];
    return O3DMI_OK;
}

// ComputeRtPointToPointCPU after its reduction (RegistrationCPU.cpp:640-650):
// Sxy = U D V^T, R = U diag(1, 1, det(U) det(V)) V^T, t = mean_t - R mean_s.
// The reference calls LAPACK gesvd; here a one-sided (Hestenes) Jacobi SVD in
// float64: columns of G = Sxy V are rotated pairwise until orthogonal, then
// u_i = g_i / |g_i|. The third left vector is taken as u_1 x u_2, which folds
// the reflection test into det(V): with u_3 = e (u_1 x u_2), e = det(U),
// det(U) det(V) u_3 = det(V) (u_1 x u_2). That also covers planar
// correspondence sets (sigma_3 = 0) without dividing by sigma_3.
int o3dmi_compute_rt_p2point(const double* sums, double* R9, double* t3) {
    O3DMI_REQUIRE(sums && R9 && t3, "null argument");
    const double cnt = sums[15];
    if (!(cnt > 0)) {
        SetLastError("No valid correspondence present.");
        return O3DMI_ERR_NO_INLIERS;
    }
    double ms[3], mt[3];
    for (int k = 0; k < 3; ++k) {
        ms[k] = sums[k] / cnt;
        mt[k] = sums[3 + k] / cnt;
    }
    double G[3][3], V[3][3];  // G[row][col]
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
            G[j][k] = sums[6 + 3 * j + k] / cnt - mt[j] * ms[k];
            V[j][k] = j == k ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += G[r][p] * G[r][p];
                    beta += G[r][q] * G[r][q];
                    gamma += G[r][p] * G[r][q];
                }
                if (gamma == 0.0 ||
                    std::fabs(gamma) <= 1e-17 * std::sqrt(alpha * beta))
                    continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tn = (zeta >= 0 ? 1.0 : -1.0) /
                                  (std::fabs(zeta) +
                                   std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + tn * tn);
                const double sn = c * tn;
                for (int r = 0; r < 3; ++r) {
                    const double gp = G[r][p], gq = G[r][q];
                    G[r][p] = c * gp - sn * gq;
                    G[r][q] = sn * gp + c * gq;
                    const double vp = V[r][p], vq = V[r][q];
                    V[r][p] = c * vp - sn * vq;
                    V[r][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    // column order by decreasing singular value
    double sig[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c)
        sig[c] = std::sqrt(G[0][c] * G[0][c] + G[1][c] * G[1][c] +
                           G[2][c] * G[2][c]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (sig[ord[b]] > sig[ord[a]]) std::swap(ord[a], ord[b]);
    double u[3][3], v[3][3];  // u[i] / v[i] = i-th singular vectors
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 3; ++r) v[i][r] = V[r][ord[i]];
    const double s0 = sig[ord[0]], s1 = sig[ord[1]];
    if (!(s0 > 0)) {
        // all correspondences coincide with their means: rotation undetermined,
        // the least-squares answer is the pure translation.
        for (int i = 0; i < 9; ++i) R9[i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 3; ++k) t3[k] = mt[k] - ms[k];
        return O3DMI_OK;
    }
    for (int r = 0; r < 3; ++r) u[0][r] = G[r][ord[0]] / s0;
    if (s1 > 1e-300 && s1 > 1e-15 * s0) {
        double d = 0, nrm = 0;
        for (int r = 0; r < 3; ++r) u[1][r] = G[r][ord[1]] / s1;
        for (int r = 0; r < 3; ++r) d += u[1][r] * u[0][r];
        for (int r = 0; r < 3; ++r) {
            u[1][r] -= d * u[0][r];
            nrm += u[1][r] * u[1][r];
        }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    } else {
        // rank one: any unit vector orthogonal to u_0
        int m = 0;
        for (int r = 1; r < 3; ++r)
            if (std::fabs(u[0][r]) < std::fabs(u[0][m])) m = r;
        double e[3] = {0, 0, 0};
        e[m] = 1.0;
        double d = u[0][m], nrm = 0;
        for (int r = 0; r < 3; ++r) {
            u[1][r] = e[r] - d * u[0][r];
            nrm += u[1][r] * u[1][r];
        }
        nrm = std::sqrt(nrm);
        for (int r = 0; r < 3; ++r) u[1][r] /= nrm;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    const double detV =
            v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) -
            v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
            v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    const double sgn = detV < 0 ? -1.0 : 1.0;
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k)
            R9[j * 3 + k] = u[0][j] * v[0][k] + u[1][j] * v[1][k] +
                            sgn * u[2][j] * v[2][k];
    for (int j = 0; j < 3; ++j)
        t3[j] = mt[j] - (R9[j * 3 + 0] * ms[0] + R9[j * 3 + 1] * ms[1] +
                         R9[j * 3 + 2] * ms[2]);
    return O3DMI_OK;
}

// PoseToSymmetricTransformation (TransformationConverter.cpp:106-133) =
// TransformSymmetricPoseToMatrix4d (pipelines/registration/SymmetricICPImpl.h:
// 19-44): theta = atan |g|, H = AngleAxis(theta, g / |g|) (Rodrigues), R = H H,
// t = target_mean + H (t' cos theta) - R source_mean.
void o3dmi_symmetric_pose_to_transformation(const double* pose,
                                            const double* source_mean,
                                            const double* target_mean,
                                            double* T) {
    const double g_norm = std::sqrt(pose[0] * pose[0] + pose[1] * pose[1] +
                                    pose[2] * pose[2]);
    const double theta = std::atan(g_norm);
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (g_norm > 0.0) {
        const double ax = pose[0] / g_norm, ay = pose[1] / g_norm,
                     az = pose[2] / g_norm;
        const double c = std::cos(theta), sn = std::sin(theta), v = 1.0 - c;
        H[0] = c + v * ax * ax;
        H[1] = v * ax * ay - sn * az;
        H[2] = v * ax * az + sn * ay;
        H[3] = v * ax * ay + sn * az;
        H[4] = c + v * ay * ay;
        H[5] = v * ay * az - sn * ax;
        H[6] = v * ax * az - sn * ay;
        H[7] = v * ay * az + sn * ax;
        H[8] = c + v * az * az;
    }
    double R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = H[i * 3 + 0] * H[0 * 3 + j] +
                           H[i * 3 + 1] * H[1 * 3 + j] +
                           H[i * 3 + 2] * H[2 * 3 + j];
    const double ct = std::cos(theta);
    const double u[3] = {pose[3] * ct, pose[4] * ct, pose[5] * ct};
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = target_mean[i] +
                       (H[i * 3 + 0] * u[0] + H[i * 3 + 1] * u[1] +
                        H[i * 3 + 2] * u[2]) -
                       (R[i * 3 + 0] * source_mean[0] +
                        R[i * 3 + 1] * source_mean[1] +
                        R[i * 3 + 2] * source_mean[2]);
    }
}

// TransformationConverterImpl.h:23-42 + TransformationConverter.cpp:81-104
void o3dmi_pose_to_transformation(const double* pose_ptr, double* T) {
    for (int i = 0; i < 16; ++i) T[i] = 0;
    T[0] = cos(pose_ptr[2]) * cos(pose_ptr[1]);
    T[1] = -1 * sin(pose_ptr[2]) * cos(pose_ptr[0]) +
           cos(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    T[2] = sin(pose_ptr[2]) * sin(pose_ptr[0]) +
           cos(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    T[4] = sin(pose_ptr[2]) * cos(pose_ptr[1]);
    T[5] = cos(pose_ptr[2]) * cos(pose_ptr[0]) +
           sin(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    T[6] = -1 * cos(pose_ptr[2]) * sin(pose_ptr[0]) +
           sin(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    T[8] = -1 * sin(pose_ptr[1]);
    T[9] = cos(pose_ptr[1]) * sin(pose_ptr[0]);
    T[10] = cos(pose_ptr[1]) * cos(pose_ptr[0]);
    T[3] = pose_ptr[3];
    T[7] = pose_ptr[4];
    T[11] = pose_ptr[5];
    T[15] = 1;
}

}  // extern "C"

// ---- device-side all-reduce support (multi-GPU source-sharded ICP) ----------
// The driver keeps the iteration's 32 sums on the device, lets the caller's
// collective (RCCL all-reduce through the hook of o3dmi_set_device_allreduce)
// run on the launch stream, and only then posts them to the host mailbox:
//   final sum -> SumsTailKernel -> [all-reduce on the stream] -> SumsPostKernel
namespace o3dmi {
namespace {
// tail[k] >= 0: sums[29 + k] = tail[k]; NaN: left as computed.
__global__ void SumsTailKernel(double* __restrict__ sums, double t29,
                               double t30, double t31) {
    if (threadIdx.x == 0) {
        if (t29 == t29) sums[29] = t29;
        if (t30 == t30) sums[30] = t30;
        if (t31 == t31) sums[31] = t31;
    }
}
__global__ void SumsPostKernel(const double* __restrict__ sums,
                               double* mail_data, int* mail_flag, int seq) {
    if (threadIdx.x < 32) mail_data[threadIdx.x] = sums[threadIdx.x];
    MailboxPublish(mail_flag, seq);
}
}  // namespace
}  // namespace o3dmi

extern "C" int o3dmi_internal_sums_tail(double* sums32_dev, double t29,
                                        double t30, double t31,
                                        o3dmi_stream_t stream) {
    hipLaunchKernelGGL(o3dmi::SumsTailKernel, dim3(1), dim3(64), 0,
                       (hipStream_t)stream, sums32_dev, t29, t30, t31);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

extern "C" int o3dmi_internal_sums_post(const double* sums32_dev,
                                        double* mail_data, int* mail_flag,
                                        int seq, o3dmi_stream_t stream) {
    hipLaunchKernelGGL(o3dmi::SumsPostKernel, dim3(1), dim3(64), 0,
                       (hipStream_t)stream, sums32_dev, mail_data, mail_flag,
                       seq);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}
