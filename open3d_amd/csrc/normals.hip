// Point-cloud normal estimation on MI355X:
// PointCloud::EstimateNormals(max_nn, radius) with hybrid search
// (cpp/open3d/t/geometry/PointCloud.cpp:856-976). Replaces
//   EstimateCovariancesUsingHybridSearchCUDA  (t/geometry/kernel/PointCloudImpl.h:588-638,
//       per-point body EstimatePointWiseRobustNormalizedCovarianceKernel :512-585)
//   EstimateNormalsFromCovariancesCUDA        (:1011-1063)
// Covariances follow the reference statement by statement (two-pass float64
// cumulants in neighbour order: bit-exact). The eigenvector behind a normal is
// this code base's own float64 Jacobi routine, not the reference's closed-form
// one (:746-1009): a tolerance, not bits -- see SmallestEigenvectorSym3.
// Gather-bound: 30 neighbours x 12 B per point, L2-resident for a sorted cloud.

#include <cmath>
#include <cstdlib>

#include "common.h"
#include "scan.h"

namespace o3dmi {
namespace {

__device__ __forceinline__ float Sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double Sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float Abs(float v) { return fabsf(v); }
__device__ __forceinline__ double Abs(double v) { return fabs(v); }

// PointCloudImpl.h:512-585
// Neighbour lists: fixed-width rows, or CSR when row_splits != NULL.
template <typename T>
__global__ void CovariancesKernel(const T* __restrict__ points,
                                  const int32_t* __restrict__ indices,
                                  const int32_t* __restrict__ counts,
                                  const int64_t* __restrict__ row_splits,
                                  int64_t n, int max_nn,
                                  T* __restrict__ covariances) {
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* idx = row_splits ? indices + row_splits[w]
                                        : indices + (int64_t)max_nn * w;
        const int32_t cnt = row_splits
                                    ? (int32_t)(row_splits[w + 1] - row_splits[w])
                                    : counts[w];
        T* cov = covariances + 9 * w;
        if (cnt < 3) {
#pragma unroll
            for (int i = 0; i < 9; ++i) cov[i] = (i % 4 == 0) ? T(1.0) : T(0.0);
            continue;
        }
        double centroid[3] = {0, 0, 0};
        for (int32_t i = 0; i < cnt; ++i) {
            const int64_t o = 3 * (int64_t)idx[i];
            centroid[0] += points[o];
            centroid[1] += points[o + 1];
            centroid[2] += points[o + 2];
        }
        centroid[0] /= cnt;
        centroid[1] /= cnt;
        centroid[2] /= cnt;
        double cumulants[6] = {0, 0, 0, 0, 0, 0};
        for (int32_t i = 0; i < cnt; ++i) {
            const int64_t o = 3 * (int64_t)idx[i];
            const double x = static_cast<double>(points[o]) - centroid[0];
            const double y = static_cast<double>(points[o + 1]) - centroid[1];
            const double z = static_cast<double>(points[o + 2]) - centroid[2];
            cumulants[0] += x * x;
            cumulants[1] += y * y;
            cumulants[2] += z * z;
            cumulants[3] += x * y;
            cumulants[4] += x * z;
            cumulants[5] += y * z;
        }
        const double normalization_factor = static_cast<double>(cnt - 1);
#pragma unroll
        for (int i = 0; i < 6; ++i) cumulants[i] /= normalization_factor;
        cov[0] = static_cast<T>(cumulants[0]);
        cov[4] = static_cast<T>(cumulants[1]);
        cov[8] = static_cast<T>(cumulants[2]);
        cov[1] = static_cast<T>(cumulants[3]);
        cov[3] = cov[1];
        cov[2] = static_cast<T>(cumulants[4]);
        cov[6] = cov[2];
        cov[5] = static_cast<T>(cumulants[5]);
        cov[7] = cov[5];
    }
}

// ---- symmetric 3x3 eigen-decomposition (this code base's own) -----------------
// Cyclic Jacobi: rotations in the (0,1), (0,2), (1,2) planes, each chosen to
// annihilate that off-diagonal entry (the smaller root of t^2 + 2 theta t - 1),
// until every off-diagonal entry is an exact zero or kJacobiSweeps sweeps have
// run (quadratic convergence: a 3x3 is at rounding level after 4 - 5). On
// return a is diagonal (eigenvalues, unsorted) and the columns of V are the
// eigenvectors. Used by the normal estimation (smallest eigenvector) and by
// the colour-gradient solve (pseudo-inverse).
constexpr int kJacobiSweeps = 8;
template <typename T>
__device__ __forceinline__ void JacobiEigenSym3(T (&a)[3][3], T (&V)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? T(1) : T(0);
    for (int sweep = 0; sweep < kJacobiSweeps; ++sweep) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const T apq = a[p][q];
                if (apq == T(0)) continue;
                const T theta = (a[q][q] - a[p][p]) / (T(2) * apq);
                const T t = (theta >= T(0) ? T(1) : T(-1)) /
                            (Abs(theta) + Sqrt(theta * theta + T(1)));
                const T c = T(1) / Sqrt(t * t + T(1));
                const T sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - sn * aqk;
                    a[q][k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
}

// Normal of a neighbourhood = unit eigenvector of the smallest eigenvalue of
// its covariance. The reference (EstimatePointWiseNormalsWithFastEigen3x3,
// t/geometry/kernel/PointCloudImpl.h:875-1009) gets it non-iteratively in the
// point dtype (trigonometric eigenvalues + cross products of rows); this
// routine is NOT that one: the covariance is widened to float64 and diagonalised
// by the converged Jacobi above, so the answer is the exact eigenvector to
// float64 rounding for both dtypes. Against the reference's compiled body the
// two agree to the reference's own rounding: <= 1e-4 rad (Float32) / 1e-10
// (Float64) wherever the two smallest eigenvalues are separated by more than
// 5 % of the largest (tests/test_normals_gpu.py); in a degenerate eigenspace
// any of its unit vectors is a valid answer and the two routines pick
// different ones.
//   * sign: an eigenvector has none, the reference's is whatever its cross
//     products produce. Pinned here: the last non-zero component is positive
//     (z > 0, else y > 0, else x > 0) -- the convention the reference's own
//     value test satisfies (cpp/tests/t/geometry/PointCloud.cpp:630-668);
//   * ties between eigenvalues go to the later axis, so the identity
//     covariance of a neighbourhood with < 3 members gives +z as in the
//     reference; an all-zero covariance has no direction: zero vector (the
//     caller turns it into +z when the cloud has no prior normals).
template <typename T>
__device__ void SmallestEigenvectorSym3(const T* cov, T* nrm) {
    double a[3][3], V[3][3];
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double m = fabs((double)cov[i]);
        scale = m > scale ? m : scale;
    }
    if (!(scale > 0.0)) {
        nrm[0] = nrm[1] = nrm[2] = T(0);
        return;
    }
    // symmetric by construction: the upper triangle is read
    const double inv = 1.0 / scale;
    a[0][0] = (double)cov[0] * inv;
    a[1][1] = (double)cov[4] * inv;
    a[2][2] = (double)cov[8] * inv;
    a[0][1] = a[1][0] = (double)cov[1] * inv;
    a[0][2] = a[2][0] = (double)cov[2] * inv;
    a[1][2] = a[2][1] = (double)cov[5] * inv;
    JacobiEigenSym3<double>(a, V);
    int best = 0;
    if (a[1][1] <= a[best][best]) best = 1;
    if (a[2][2] <= a[best][best]) best = 2;
    double v[3] = {V[0][best], V[1][best], V[2][best]};
    const double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const bool flip = v[2] < 0.0 ||
                      (v[2] == 0.0 && (v[1] < 0.0 || (v[1] == 0.0 && v[0] < 0.0)));
    const double s = (flip ? -1.0 : 1.0) / len;
    nrm[0] = (T)(v[0] * s);
    nrm[1] = (T)(v[1] * s);
    nrm[2] = (T)(v[2] * s);
}

// PointCloudImpl.h:1011-1063
template <typename T>
__global__ void NormalsFromCovariancesKernel(const T* __restrict__ covariances,
                                             int64_t n, T* __restrict__ normals,
                                             bool has_normals) {
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        T cov[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) cov[i] = covariances[9 * w + i];
        T out[3] = {0, 0, 0};
        SmallestEigenvectorSym3<T>(cov, out);
        if ((out[0] * out[0] + out[1] * out[1] + out[2] * out[2]) == 0.0 &&
            !has_normals) {
            out[0] = 0.0; out[1] = 0.0; out[2] = 1.0;
        }
        if (has_normals) {
            if ((normals[3 * w] * out[0] + normals[3 * w + 1] * out[1] +
                 normals[3 * w + 2] * out[2]) < 0.0) {
                out[0] *= -1; out[1] *= -1; out[2] *= -1;
            }
        }
        normals[3 * w] = out[0];
        normals[3 * w + 1] = out[1];
        normals[3 * w + 2] = out[2];
    }
}

// x = pinv(AtA) Atb for a symmetric 3x3 AtA: cyclic Jacobi eigen-decomposition
// (8 sweeps, converged for 3x3), eigenvalues below 1e-10 dropped -- the
// minimum-norm solution of the normal equations, which is what the reference
// asks of core::linalg::kernel::solve_svd3x3 (PointCloudImpl.h:1163). The
// reference's routine is an APPROXIMATE SVD (four fixed Jacobi sweeps in
// single-precision-oriented arithmetic); this solver is this code base's own
// and converged, so the two agree to the accuracy of the reference's
// approximation: tests/test_icp_gpu.py::test_color_gradients_vs_reference_body
// bounds the difference against the reference's compiled body (median ~1e-6
// of the gradient scale in Float32, a percent-level tail on ill-conditioned
// neighbourhoods; the reference's Float64 path returns NaN on ~9 % of them,
// this one never does). See DESIGN.md section 7.
template <typename T>
__device__ __forceinline__ void PinvSolveSym3(const T* AtA, const T* Atb,
                                              T* x) {
    T a[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[i][j] = AtA[i * 3 + j];
    JacobiEigenSym3<T>(a, V);
    const T epsilon = (T)1e-10;
    x[0] = x[1] = x[2] = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T lam = a[i][i];
        const T inv = Abs(lam) < epsilon ? T(0) : T(1) / lam;
        const T proj = V[0][i] * Atb[0] + V[1][i] * Atb[1] + V[2][i] * Atb[2];
        const T w = inv * proj;
        x[0] += V[0][i] * w;
        x[1] += V[1][i] * w;
        x[2] += V[2][i] * w;
    }
}

// EstimatePointWiseColorGradientKernel, PointCloudImpl.h:1067-1165: intensity
// least squares over the neighbours projected on the tangent plane + the
// constraint gradient . normal = 0 (the first neighbour is the point itself).
// Neighbour lists: fixed-width rows (indices + max_nn * w, counts[w]) or, when
// row_splits != NULL, CSR (radius search).
template <typename T>
__global__ void ColorGradientsKernel(const T* __restrict__ points,
                                     const T* __restrict__ normals,
                                     const T* __restrict__ colors,
                                     const int32_t* __restrict__ indices,
                                     const int32_t* __restrict__ counts,
                                     const int64_t* __restrict__ row_splits,
                                     int64_t n, int max_nn,
                                     T* __restrict__ gradients) {
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* idx = row_splits ? indices + row_splits[w]
                                        : indices + (int64_t)max_nn * w;
        const int32_t cnt = row_splits
                                    ? (int32_t)(row_splits[w + 1] - row_splits[w])
                                    : counts[w];
        T* out = gradients + 3 * w;
        if (cnt < 4) {
            out[0] = 0;
            out[1] = 0;
            out[2] = 0;
            continue;
        }
        const T vt[3] = {points[3 * w], points[3 * w + 1], points[3 * w + 2]};
        const T nt[3] = {normals[3 * w], normals[3 * w + 1],
                         normals[3 * w + 2]};
        const T it = (colors[3 * w] + colors[3 * w + 1] + colors[3 * w + 2]) /
                     3.0;
        T AtA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        T Atb[3] = {0, 0, 0};
        const T s = vt[0] * nt[0] + vt[1] * nt[1] + vt[2] * nt[2];
        int i = 1;
        for (; i < cnt; i++) {
            const int64_t o = 3 * (int64_t)idx[i];
            const T vt_adj[3] = {points[o], points[o + 1], points[o + 2]};
            const T d = vt_adj[0] * nt[0] + vt_adj[1] * nt[1] +
                        vt_adj[2] * nt[2] - s;
            const T vt_proj[3] = {vt_adj[0] - d * nt[0], vt_adj[1] - d * nt[1],
                                  vt_adj[2] - d * nt[2]};
            const T it_adj = (colors[o + 0] + colors[o + 1] + colors[o + 2]) /
                             3.0;
            const T A[3] = {vt_proj[0] - vt[0], vt_proj[1] - vt[1],
                            vt_proj[2] - vt[2]};
            AtA[0] += A[0] * A[0];
            AtA[1] += A[1] * A[0];
            AtA[2] += A[2] * A[0];
            AtA[4] += A[1] * A[1];
            AtA[5] += A[2] * A[1];
            AtA[8] += A[2] * A[2];
            const T b = it_adj - it;
            Atb[0] += A[0] * b;
            Atb[1] += A[1] * b;
            Atb[2] += A[2] * b;
        }
        const T A[3] = {(i - 1) * nt[0], (i - 1) * nt[1], (i - 1) * nt[2]};
        AtA[0] += A[0] * A[0];
        AtA[1] += A[0] * A[1];
        AtA[2] += A[0] * A[2];
        AtA[4] += A[1] * A[1];
        AtA[5] += A[1] * A[2];
        AtA[8] += A[2] * A[2];
        AtA[3] = AtA[1];
        AtA[6] = AtA[2];
        AtA[7] = AtA[5];
        T x[3];
        PinvSolveSym3<T>(AtA, Atb, x);
        out[0] = x[0];
        out[1] = x[1];
        out[2] = x[2];
    }
}

}  // namespace
}  // namespace o3dmi

using namespace o3dmi;

namespace {

// Sorted neighbour lists of every point of a cloud within the index radius,
// CSR: FixedRadiusSearch = count pass, prefix sum, write pass.
struct CsrLists {
    char* buf = nullptr;         // counts | splits | tile totals of the scan
    int64_t* splits = nullptr;   // [n + 1]
    int32_t* indices = nullptr;  // [total]
    void Free() {
        PoolFree(indices);
        PoolFree(buf);
        indices = nullptr;
        buf = nullptr;
    }
};

int BuildRadiusLists(const o3dmi_nns_t* index, const void* points_dev,
                     int64_t n, hipStream_t s, CsrLists* out) {
    o3dmi_stream_t stream = (o3dmi_stream_t)s;
    const size_t cnt_bytes = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255;
    const size_t spl_bytes =
            (sizeof(int64_t) * (size_t)(n + 1) + 255) & ~(size_t)255;
    const size_t tmp_bytes = (ScanScratchBytes(n) + 255) & ~(size_t)255;
    int st = PoolAlloc((void**)&out->buf, cnt_bytes + spl_bytes + tmp_bytes);
    if (st) return st;
    int32_t* cnt = (int32_t*)out->buf;
    out->splits = (int64_t*)(out->buf + cnt_bytes);
    void* tmp = out->buf + cnt_bytes + spl_bytes;
    if ((st = o3dmi_nns_radius_count(index, points_dev, n, cnt, stream)))
        return st;
    int64_t total = 0;
    if (hipMemsetAsync(out->splits, 0, sizeof(int64_t), s) != hipSuccess) {
        SetLastError("radius lists: memset failed");
        return O3DMI_ERR_HIP;
    }
    if ((st = PrefixSumAsync(cnt, n, true, out->splits + 1, nullptr, tmp, s)))
        return st;
    if (hipMemcpyAsync(&total, out->splits + n, sizeof(int64_t),
                       hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        SetLastError("radius lists: prefix sum failed");
        return O3DMI_ERR_HIP;
    }
    if ((st = PoolAlloc((void**)&out->indices,
                        sizeof(int32_t) * (size_t)(total > 0 ? total : 1))))
        return st;
    return o3dmi_nns_radius_search(index, points_dev, n, out->splits,
                                   out->indices, nullptr, stream);
}

}  // namespace

extern "C" {

int o3dmi_pointcloud_estimate_covariances(const void* points_dev,
                                          const int32_t* indices_dev,
                                          const int32_t* counts_dev, int64_t n,
                                          int max_nn, int dtype,
                                          void* covariances_dev,
                                          o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(n >= 0 && max_nn >= 1, "bad sizes");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(points_dev && indices_dev && counts_dev && covariances_dev,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(CovariancesKernel<double>, grid, block, 0, s,
                           (const double*)points_dev, indices_dev, counts_dev,
                           (const int64_t*)nullptr, n, max_nn,
                           (double*)covariances_dev);
    else
        hipLaunchKernelGGL(CovariancesKernel<float>, grid, block, 0, s,
                           (const float*)points_dev, indices_dev, counts_dev,
                           (const int64_t*)nullptr, n, max_nn,
                           (float*)covariances_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_pointcloud_normals_from_covariances(const void* covariances_dev,
                                              int64_t n, int dtype,
                                              void* normals_dev,
                                              int has_normals,
                                              o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "covariances must be Float32 or Float64");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(covariances_dev && normals_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (dtype == O3DMI_F64)
        hipLaunchKernelGGL(NormalsFromCovariancesKernel<double>, grid, block, 0,
                           s, (const double*)covariances_dev, n,
                           (double*)normals_dev, has_normals != 0);
    else
        hipLaunchKernelGGL(NormalsFromCovariancesKernel<float>, grid, block, 0,
                           s, (const float*)covariances_dev, n,
                           (float*)normals_dev, has_normals != 0);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_nns_knn_search_counts(const void* points_dev, int64_t n,
                                const void* queries_dev, int64_t q, int dtype,
                                int knn, int32_t* idx_dev, void* dist2_dev,
                                int32_t* counts_dev, o3dmi_stream_t stream);

// PointCloud::EstimateNormals(max_knn, radius), PointCloud.cpp:856-976: index
// over the cloud itself, hybrid search (both given), KNN search (radius <= 0,
// the reference's default max_nn = 30 / radius = nullopt) or radius search
// (max_nn <= 0), covariances, normals; the "covariances" attribute is a temporary, as in the reference.
int o3dmi_pointcloud_estimate_normals(const void* points_dev, int64_t n,
                                      int dtype, int max_nn, double radius,
                                      void* normals_dev, int has_normals,
                                      o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    O3DMI_REQUIRE(max_nn > 0 || radius > 0, "Both max_nn and radius are none.");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(points_dev && normals_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == O3DMI_F64 ? 8 : 4;
    if (max_nn <= 0) {
        // EstimateCovariancesUsingRadiusSearch, PointCloudImpl.h:641-689: CSR
        // lists from the fixed-radius search (sorted by distance, as the
        // reference's), then the same per-point covariance body -- bit for
        // bit. (o3dmi_nns_radius_covariances is the list-free fast form:
        // float64 wave sums, equal to rounding.)
        O3DMI_REQUIRE(n < (1ll << 31), "too many points");
        o3dmi_nns_t* index = nullptr;
        int st = o3dmi_nns_create(points_dev, n, dtype, radius, stream, &index);
        if (st) return st;
        CsrLists lists;
        st = BuildRadiusLists(index, points_dev, n, s, &lists);
        void* cov = nullptr;
        if (!st) st = PoolAlloc(&cov, esz * 9 * (size_t)n);
        if (!st) {
            dim3 grid(GridFor(n, kBlock)), block(kBlock);
            if (dtype == O3DMI_F64)
                hipLaunchKernelGGL(CovariancesKernel<double>, grid, block, 0, s,
                                   (const double*)points_dev, lists.indices,
                                   (const int32_t*)nullptr, lists.splits, n, 1,
                                   (double*)cov);
            else
                hipLaunchKernelGGL(CovariancesKernel<float>, grid, block, 0, s,
                                   (const float*)points_dev, lists.indices,
                                   (const int32_t*)nullptr, lists.splits, n, 1,
                                   (float*)cov);
            if (hipGetLastError() != hipSuccess) {
                SetLastError("covariance kernel launch failed");
                st = O3DMI_ERR_HIP;
            }
        }
        if (!st)
            st = o3dmi_pointcloud_normals_from_covariances(
                    cov, n, dtype, normals_dev, has_normals, stream);
        (void)hipStreamSynchronize(s);
        PoolFree(cov);
        lists.Free();
        o3dmi_nns_destroy(index);
        return st;
    }
    if (!(radius > 0)) {
        // EstimateCovariancesUsingKNNSearch, PointCloudImpl.h:692-744
        const int k = (int)(n < (int64_t)max_nn ? n : (int64_t)max_nn);
        O3DMI_REQUIRE(k >= 3,
                      "Not enough neighbors to compute Covariances / Normals. "
                      "Try increasing the max_nn parameter.");
        char* scratch = nullptr;
        const size_t idx_bytes =
                (sizeof(int32_t) * (size_t)n * k + 255) & ~(size_t)255;
        const size_t cnt_bytes = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255;
        int st = PoolAlloc((void**)&scratch,
                           idx_bytes + cnt_bytes + esz * 9 * (size_t)n);
        if (st) return st;
        int32_t* idx = (int32_t*)scratch;
        int32_t* cnt = (int32_t*)(scratch + idx_bytes);
        void* cov = scratch + idx_bytes + cnt_bytes;
        st = o3dmi_nns_knn_search_counts(points_dev, n, points_dev, n, dtype, k,
                                         idx, nullptr, cnt, stream);
        if (!st)
            st = o3dmi_pointcloud_estimate_covariances(points_dev, idx, cnt, n,
                                                       k, dtype, cov, stream);
        if (!st)
            st = o3dmi_pointcloud_normals_from_covariances(
                    cov, n, dtype, normals_dev, has_normals, stream);
        (void)hipStreamSynchronize(s);
        PoolFree(scratch);
        return st;
    }
    o3dmi_nns_t* nns = nullptr;
    int st = o3dmi_nns_create(points_dev, n, dtype, radius, stream, &nns);
    if (st) return st;
    char* scratch = nullptr;
    const size_t idx_bytes = (sizeof(int32_t) * (size_t)n * max_nn + 255) & ~(size_t)255;
    const size_t cnt_bytes = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255;
    const size_t cov_bytes = esz * 9 * (size_t)n;
    st = PoolAlloc((void**)&scratch, idx_bytes + cnt_bytes + cov_bytes);
    if (!st) {
        int32_t* idx = (int32_t*)scratch;
        int32_t* cnt = (int32_t*)(scratch + idx_bytes);
        void* cov = scratch + idx_bytes + cnt_bytes;
        st = o3dmi_nns_hybrid_search(nns, points_dev, n, max_nn, idx, nullptr,
                                     cnt, stream);
        if (!st)
            st = o3dmi_pointcloud_estimate_covariances(points_dev, idx, cnt, n,
                                                       max_nn, dtype, cov,
                                                       stream);
        if (!st)
            st = o3dmi_pointcloud_normals_from_covariances(
                    cov, n, dtype, normals_dev, has_normals, stream);
        (void)hipStreamSynchronize(s);
        PoolFree(scratch);
    }
    o3dmi_nns_destroy(nns);
    return st;
}

// EstimateColorGradientsUsing{Hybrid,KNN}SearchCUDA after the search
// (PointCloudImpl.h:1170-1290): gradients {n,3} from given neighbour lists.
static int ColorGradientsLaunch(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        const int32_t* indices_dev, const int32_t* counts_dev,
        const int64_t* row_splits_dev, int64_t n, int max_nn, int dtype,
        void* gradients_dev, o3dmi_stream_t stream);

int o3dmi_pointcloud_color_gradients_from_neighbors(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        const int32_t* indices_dev, const int32_t* counts_dev, int64_t n,
        int max_nn, int dtype, void* gradients_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n == 0 || counts_dev != nullptr, "null argument");
    return ColorGradientsLaunch(points_dev, normals_dev, colors_dev,
                                indices_dev, counts_dev, nullptr, n, max_nn,
                                dtype, gradients_dev, stream);
}

static int ColorGradientsLaunch(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        const int32_t* indices_dev, const int32_t* counts_dev,
        const int64_t* row_splits_dev, int64_t n, int max_nn, int dtype,
        void* gradients_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "points must be Float32 or Float64");
    O3DMI_REQUIRE(n >= 0 && max_nn >= 1, "bad sizes");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(points_dev && normals_dev && colors_dev && indices_dev &&
                          (counts_dev || row_splits_dev) && gradients_dev,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
#define O3DMI_GRAD(T)                                                          \
    hipLaunchKernelGGL((ColorGradientsKernel<T>), grid, block, 0, s,           \
                       (const T*)points_dev, (const T*)normals_dev,            \
                       (const T*)colors_dev, indices_dev, counts_dev,          \
                       row_splits_dev, n, max_nn, (T*)gradients_dev)
    if (dtype == O3DMI_F64) O3DMI_GRAD(double);
    else O3DMI_GRAD(float);
#undef O3DMI_GRAD
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// PointCloud::EstimateColorGradients(max_knn, radius), PointCloud.cpp:987-1060
// (hybrid search when radius > 0, KNN search otherwise). Synchronises.
int o3dmi_pointcloud_estimate_color_gradients(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        int64_t n, int dtype, int max_nn, double radius, void* gradients_dev,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(dtype == O3DMI_F32 || dtype == O3DMI_F64,
                  "Only Float32 and Float64 point clouds are supported.");
    O3DMI_REQUIRE(max_nn > 0 || radius > 0, "Both max_nn and radius are none.");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(points_dev && normals_dev && colors_dev && gradients_dev,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (max_nn <= 0) {
        // EstimateColorGradientsUsingRadiusSearch (PointCloudImpl.h): CSR
        // lists from the fixed-radius search (count, prefix sum, write).
        O3DMI_REQUIRE(n < (1ll << 31), "too many points");
        o3dmi_nns_t* index = nullptr;
        int st = o3dmi_nns_create(points_dev, n, dtype, radius, stream, &index);
        if (st) return st;
        CsrLists lists;
        st = BuildRadiusLists(index, points_dev, n, s, &lists);
        if (!st)
            st = ColorGradientsLaunch(points_dev, normals_dev, colors_dev,
                                      lists.indices, nullptr, lists.splits, n,
                                      1, dtype, gradients_dev, stream);
        (void)hipStreamSynchronize(s);
        lists.Free();
        o3dmi_nns_destroy(index);
        return st;
    }
    const int k = (int)(n < (int64_t)max_nn ? n : (int64_t)max_nn);
    char* scratch = nullptr;
    const size_t idx_bytes =
            (sizeof(int32_t) * (size_t)n * k + 255) & ~(size_t)255;
    const size_t cnt_bytes = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255;
    int st = PoolAlloc((void**)&scratch, idx_bytes + cnt_bytes);
    if (st) return st;
    int32_t* idx = (int32_t*)scratch;
    int32_t* cnt = (int32_t*)(scratch + idx_bytes);
    o3dmi_nns_t* nns = nullptr;
    if (radius > 0) {
        st = o3dmi_nns_create(points_dev, n, dtype, radius, stream, &nns);
        if (!st)
            st = o3dmi_nns_hybrid_search(nns, points_dev, n, k, idx, nullptr,
                                         cnt, stream);
    } else {
        st = o3dmi_nns_knn_search_counts(points_dev, n, points_dev, n, dtype, k,
                                         idx, nullptr, cnt, stream);
    }
    if (!st)
        st = o3dmi_pointcloud_color_gradients_from_neighbors(
                points_dev, normals_dev, colors_dev, idx, cnt, n, k, dtype,
                gradients_dev, stream);
    (void)hipStreamSynchronize(s);
    PoolFree(scratch);
    if (nns) o3dmi_nns_destroy(nns);
    return st;
}

}  // extern "C"
