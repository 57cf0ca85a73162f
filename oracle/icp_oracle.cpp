// TEST INFRASTRUCTURE ONLY -- see oracle_common.h.
//
// CPU restatement of the ICP half of the hot path:
//   hybrid search      cpp/open3d/core/nns/NanoFlannImpl.h:305-370 (call site);
//                      arithmetic lives in nanoflann v1.5.0 (pinned by
//                      3rdparty/nanoflann/nanoflann.cmake:6, NOT vendored):
//                      KDTreeSingleIndexAdaptor::radiusSearch with
//                      L2_Adaptor (NanoFlannImpl.h:55-58): d2 accumulated as
//                      ((dx*dx)+dy*dy)+dz*dz with dx = query - point, in T;
//                      RadiusResultSet::addPoint keeps `dist < radius`
//                      (strict); SearchParameters.sorted=true sorts ascending
//                      by distance. Ties: std::sort order is unspecified in
//                      the reference -> here lowest index wins (documented).
//   GetJacobianPointToPlane   cpp/open3d/t/pipelines/kernel/RegistrationImpl.h:251-287
//   29-reduction              cpp/open3d/t/pipelines/kernel/RegistrationCPU.cpp:30-122
//   robust kernels            cpp/open3d/t/pipelines/registration/RobustKernelImpl.h:35-126
//   DecodeAndSolve6x6         cpp/open3d/t/pipelines/kernel/TransformationConverter.cpp:189-226
//   PoseToTransformation      .../TransformationConverter.cpp:81-104, ...Impl.h:23-42
//   TransformPoints/Normals   cpp/open3d/t/geometry/kernel/TransformImpl.h:19-60
//   ICP / MultiScaleICP       cpp/open3d/t/pipelines/registration/Registration.cpp:24-62,275-444
//   VoxelDownSample           cpp/open3d/t/geometry/PointCloud.cpp:496-567
//   P2Plane ComputeRMSE       cpp/open3d/t/pipelines/registration/TransformationEstimation.cpp:160-193
//
// Pinned by the reference's in-source known answers (tests/test_oracle_goldens.py):
//   point-to-plane 0.335499 -> 0.601422  (cpp/tests/t/pipelines/registration/TransformationEstimation.cpp:33-86,148,176)
//   hybrid search {1,4,-1} / {0.00626358,0.00747938,0} / 2 (cpp/tests/core/NearestNeighborSearch.cpp:321-377)
//   robust kernel weights at r=0.98 (cpp/tests/t/pipelines/registration/Registration.cpp:411-490)
//   Solve {3,1;1,2} x = {9,8} -> {2,3} (cpp/tests/core/Linalg.cpp:454-480)
//   Pose zeros -> identity (cpp/tests/t/pipelines/TransformationConverter.cpp:37-48)
// Parity unpinned: TBB parallel_reduce split order (float rounding of the
// 29-sum) -- the oracle offers a sequential scalar_t sum and a double sum.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "approx_svd3_oracle.h"

namespace {

using std::abs;
using std::exp;
using std::max;
using std::min;
using std::pow;

template <typename scalar_t>
scalar_t Square(const scalar_t& x) {
    return x * x;
}

// GeometryMacros.h:58-63 (note: with y == 0 this is never true).
template <typename scalar_t, typename T>
bool IsClose(const scalar_t& x, const T& y, const double rtol = 1e-4) {
    return ((x > (1.0 - rtol) * y) && (x < (1.0 + rtol) * y));
}

// RobustKernelImpl.h:35-126, literal (including the double-typed literals that
// promote parts of the expression to double before the scalar_t return).
template <typename scalar_t>
scalar_t RobustWeight(int method, double scaling_parameter,
                      double shape_parameter, scalar_t residual) {
    scalar_t scale = static_cast<scalar_t>(scaling_parameter);
    switch (method) {
        case 0:  // L2Loss
            return 1.0;
        case 1:  // L1Loss
            return 1.0 / abs(residual);
        case 2:  // HuberLoss
            return scale / max(abs(residual), scale);
        case 3:  // CauchyLoss
            return 1.0 / (1.0 + Square(residual / scale));
        case 4:  // GMLoss
            return scale / Square(scale + Square(residual));
        case 5:  // TukeyLoss
            return Square(1.0 - Square(min((scalar_t)1.0,
                                           abs(residual) / scale)));
        case 6:  // GeneralizedLoss
            if (IsClose(shape_parameter, 2.0, 1e-3)) {
                auto const_val = 1.0 / Square(scale);
                return const_val;
            } else if (IsClose(shape_parameter, 0.0, 1e-3)) {
                return 2.0 / (Square(residual) + 2 * Square(scale));
            } else if (shape_parameter < -1e7) {
                return exp(Square(residual / scale) / (-2.0)) / Square(scale);
            } else {
                return pow((Square(residual / scale) /
                                    abs(shape_parameter - 2.0) +
                            1),
                           ((shape_parameter / 2.0) - 1.0)) /
                       Square(scale);
            }
        default:
            return 1.0;
    }
}

// RegistrationImpl.h:251-287
template <typename scalar_t>
inline bool GetJacobianPointToPlane(int64_t workload_idx,
                                    const scalar_t* source_points_ptr,
                                    const scalar_t* target_points_ptr,
                                    const scalar_t* target_normals_ptr,
                                    const int64_t* correspondence_indices,
                                    scalar_t* J_ij, scalar_t& r) {
    if (correspondence_indices[workload_idx] == -1) {
        return false;
    }
    const int64_t target_idx = 3 * correspondence_indices[workload_idx];
    const int64_t source_idx = 3 * workload_idx;

    const scalar_t& sx = source_points_ptr[source_idx + 0];
    const scalar_t& sy = source_points_ptr[source_idx + 1];
    const scalar_t& sz = source_points_ptr[source_idx + 2];
    const scalar_t& tx = target_points_ptr[target_idx + 0];
    const scalar_t& ty = target_points_ptr[target_idx + 1];
    const scalar_t& tz = target_points_ptr[target_idx + 2];
    const scalar_t& nx = target_normals_ptr[target_idx + 0];
    const scalar_t& ny = target_normals_ptr[target_idx + 1];
    const scalar_t& nz = target_normals_ptr[target_idx + 2];

    r = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;

    J_ij[0] = nz * sy - ny * sz;
    J_ij[1] = nx * sz - nz * sx;
    J_ij[2] = ny * sx - nx * sy;
    J_ij[3] = nx;
    J_ij[4] = ny;
    J_ij[5] = nz;
    return true;
}

// RegistrationCPU.cpp:30-90, as ONE sequential range (TBB would split it).
// acc_t = scalar_t reproduces the reference arithmetic for a single range;
// acc_t = double keeps the per-term products in scalar_t (as the reference)
// but sums them in double.
template <typename scalar_t, typename acc_t>
void ComputePosePointToPlaneKernel(const scalar_t* source_points_ptr,
                                   const scalar_t* target_points_ptr,
                                   const scalar_t* target_normals_ptr,
                                   const int64_t* correspondence_indices,
                                   int64_t n, acc_t* global_sum, int method,
                                   double scaling, double shape) {
    acc_t A[29];
    for (int i = 0; i < 29; ++i) A[i] = 0;
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        scalar_t J_ij[6];
        scalar_t r = 0;
        bool valid = GetJacobianPointToPlane<scalar_t>(
                workload_idx, source_points_ptr, target_points_ptr,
                target_normals_ptr, correspondence_indices, J_ij, r);
        scalar_t w = RobustWeight<scalar_t>(method, scaling, shape, r);
        if (valid) {
            int i = 0;
            for (int j = 0; j < 6; ++j) {
                for (int k = 0; k <= j; ++k) {
                    A[i] += J_ij[j] * w * J_ij[k];
                    ++i;
                }
                A[21 + j] += J_ij[j] * w * r;
            }
            A[27] += r;
            A[28] += 1;
        }
    }
    for (int i = 0; i < 29; ++i) global_sum[i] = A[i];
}

// LAPACK ?gesv restated: LU with partial (row) pivoting, then two triangular
// solves (core/linalg/SolveCPU.cpp:15-30 -> LAPACKE_dgesv). Returns 0 on
// success, >0 if U is exactly singular (gesv's info>0 -> OPEN3D_LAPACK_CHECK
// throws, core/linalg/LinalgUtils.h:35-42).
int SolveLU(int n, double* A /*row-major n*n, destroyed*/, double* b) {
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double mx = std::abs(A[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = std::abs(A[i * n + k]);
            if (v > mx) {
                mx = v;
                p = i;
            }
        }
        if (mx == 0.0) return k + 1;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[p * n + j]);
            std::swap(b[k], b[p]);
        }
        for (int i = k + 1; i < n; ++i) {
            double l = A[i * n + k] / A[k * n + k];
            A[i * n + k] = l;
            for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
        }
    }
    // forward (unit lower)
    for (int i = 1; i < n; ++i)
        for (int j = 0; j < i; ++j) b[i] -= A[i * n + j] * b[j];
    // backward
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j) b[i] -= A[i * n + j] * b[j];
        b[i] /= A[i * n + i];
    }
    return 0;
}

// TransformationConverter.cpp:189-226
int DecodeAndSolve6x6(const double* A_1x29, double* delta,
                      float* inlier_residual, int* inlier_count) {
    double AtA[36], Atb[6];
    for (int j = 0; j < 6; j++) {
        Atb[j] = A_1x29[21 + j];
        const int64_t reduction_idx = ((j * (j + 1)) / 2);
        for (int k = 0; k <= j; k++) {
            AtA[j * 6 + k] = A_1x29[reduction_idx + k];
            AtA[k * 6 + j] = A_1x29[reduction_idx + k];
        }
    }
    double rhs[6];
    for (int j = 0; j < 6; ++j) rhs[j] = -Atb[j];
    int info = SolveLU(6, AtA, rhs);
    if (info != 0) {
        // Reference: LogError("Singular 6x6 linear system detected, tracking
        // failed.") throws.
        for (int j = 0; j < 6; ++j) delta[j] = 0;
        *inlier_residual = 0;
        *inlier_count = 0;
        return 1;
    }
    for (int j = 0; j < 6; ++j) delta[j] = rhs[j];
    *inlier_residual = (float)A_1x29[27];
    *inlier_count = static_cast<int>(A_1x29[28]);
    return 0;
}

// TransformationConverterImpl.h:23-42 + TransformationConverter.cpp:81-104
void PoseToTransformation(const double* pose_ptr, double* transformation_ptr) {
    for (int i = 0; i < 16; ++i) transformation_ptr[i] = 0;
    transformation_ptr[0] = cos(pose_ptr[2]) * cos(pose_ptr[1]);
    transformation_ptr[1] =
            -1 * sin(pose_ptr[2]) * cos(pose_ptr[0]) +
            cos(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    transformation_ptr[2] =
            sin(pose_ptr[2]) * sin(pose_ptr[0]) +
            cos(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    transformation_ptr[4] = sin(pose_ptr[2]) * cos(pose_ptr[1]);
    transformation_ptr[5] =
            cos(pose_ptr[2]) * cos(pose_ptr[0]) +
            sin(pose_ptr[2]) * sin(pose_ptr[1]) * sin(pose_ptr[0]);
    transformation_ptr[6] =
            -1 * cos(pose_ptr[2]) * sin(pose_ptr[0]) +
            sin(pose_ptr[2]) * sin(pose_ptr[1]) * cos(pose_ptr[0]);
    transformation_ptr[8] = -1 * sin(pose_ptr[1]);
    transformation_ptr[9] = cos(pose_ptr[1]) * sin(pose_ptr[0]);
    transformation_ptr[10] = cos(pose_ptr[1]) * cos(pose_ptr[0]);
    transformation_ptr[3] = pose_ptr[3];
    transformation_ptr[7] = pose_ptr[4];
    transformation_ptr[11] = pose_ptr[5];
    transformation_ptr[15] = 1;
}

// TransformImpl.h:19-44 -- transformation is first cast to the point dtype
// (kernel/Transform.cpp:20-45).
template <typename scalar_t>
void TransformPoints(const double* T, scalar_t* points, int64_t n) {
    scalar_t t[16];
    for (int i = 0; i < 16; ++i) t[i] = (scalar_t)T[i];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        scalar_t* p = points + 3 * i;
        scalar_t x[4] = {t[0] * p[0] + t[1] * p[1] + t[2] * p[2] + t[3],
                         t[4] * p[0] + t[5] * p[1] + t[6] * p[2] + t[7],
                         t[8] * p[0] + t[9] * p[1] + t[10] * p[2] + t[11],
                         t[12] * p[0] + t[13] * p[1] + t[14] * p[2] + t[15]};
        p[0] = x[0] / x[3];
        p[1] = x[1] / x[3];
        p[2] = x[2] / x[3];
    }
}

// TransformImpl.h:46-60
template <typename scalar_t>
void TransformNormals(const double* T, scalar_t* normals, int64_t n) {
    scalar_t t[16];
    for (int i = 0; i < 16; ++i) t[i] = (scalar_t)T[i];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        scalar_t* p = normals + 3 * i;
        scalar_t x[3] = {t[0] * p[0] + t[1] * p[1] + t[2] * p[2],
                         t[4] * p[0] + t[5] * p[1] + t[6] * p[2],
                         t[8] * p[0] + t[9] * p[1] + t[10] * p[2]};
        p[0] = x[0];
        p[1] = x[1];
        p[2] = x[2];
    }
}

// ---------------------------------------------------------------------------
// Exact fixed-radius "hybrid" search with nanoflann semantics.
struct CellKey {
    int64_t x, y, z;
    bool operator==(const CellKey& o) const {
        return x == o.x && y == o.y && z == o.z;
    }
};
struct CellKeyHash {
    size_t operator()(const CellKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint64_t v : {(uint64_t)k.x, (uint64_t)k.y, (uint64_t)k.z}) {
            h ^= v;
            h *= 1099511628211ull;
        }
        return (size_t)h;
    }
};

template <typename T>
struct GridIndex {
    const T* points;
    int64_t n;
    double cell;
    std::unordered_map<CellKey, std::vector<int64_t>, CellKeyHash> cells;

    GridIndex(const T* pts, int64_t n_, double radius) : points(pts), n(n_) {
        // Slightly larger than the radius so that float rounding of d2 can
        // never reach a point two cells away.
        cell = radius * 1.001 + 1e-12;
        cells.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            cells[Key(pts + 3 * i)].push_back(i);
        }
    }
    CellKey Key(const T* p) const {
        return CellKey{(int64_t)std::floor((double)p[0] / cell),
                       (int64_t)std::floor((double)p[1] / cell),
                       (int64_t)std::floor((double)p[2] / cell)};
    }
    // Collect all (d2, idx) with d2 < r2, sorted ascending by (d2, idx).
    void Query(const T* q, T radius_squared,
               std::vector<std::pair<T, int64_t>>& out) const {
        out.clear();
        CellKey c = Key(q);
        for (int64_t dx = -1; dx <= 1; ++dx)
            for (int64_t dy = -1; dy <= 1; ++dy)
                for (int64_t dz = -1; dz <= 1; ++dz) {
                    auto it = cells.find(CellKey{c.x + dx, c.y + dy, c.z + dz});
                    if (it == cells.end()) continue;
                    for (int64_t idx : it->second) {
                        const T* p = points + 3 * idx;
                        // nanoflann::L2_Adaptor::evalMetric, size==3 tail loop.
                        T result = T();
                        const T d0 = q[0] - p[0];
                        result += d0 * d0;
                        const T d1 = q[1] - p[1];
                        result += d1 * d1;
                        const T d2 = q[2] - p[2];
                        result += d2 * d2;
                        // RadiusResultSet::addPoint: strict.
                        if (result < radius_squared)
                            out.emplace_back(result, idx);
                    }
                }
        std::sort(out.begin(), out.end());
    }
};

// NanoFlannImpl.h:305-370 (_HybridSearchCPU).
template <typename T>
void HybridSearch(const T* points, int64_t num_points, const T* queries,
                  int64_t num_queries, double radius_d, int max_knn,
                  int32_t* indices_ptr, T* distances_ptr, int32_t* counts_ptr) {
    if (num_queries == 0 || num_points == 0) return;
    const T radius = (T)radius_d;
    T radius_squared = radius * radius;
    GridIndex<T> grid(points, num_points, radius_d);
#pragma omp parallel
    {
        std::vector<std::pair<T, int64_t>> ret_matches;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < num_queries; ++i) {
            grid.Query(queries + 3 * i, radius_squared, ret_matches);
            size_t num_results = ret_matches.size();
            int32_t count_i = static_cast<int32_t>(num_results);
            count_i = count_i < max_knn ? count_i : max_knn;
            counts_ptr[i] = count_i;

            int neighbor_idx = 0;
            for (auto it = ret_matches.begin();
                 it < ret_matches.end() && neighbor_idx < max_knn;
                 it++, neighbor_idx++) {
                indices_ptr[i * max_knn + neighbor_idx] = (int32_t)it->second;
                distances_ptr[i * max_knn + neighbor_idx] = it->first;
            }
            while (neighbor_idx < max_knn) {
                indices_ptr[i * max_knn + neighbor_idx] = -1;
                distances_ptr[i * max_knn + neighbor_idx] = 0;
                neighbor_idx += 1;
            }
        }
    }
}

// Brute-force variant (no grid) used to cross-check the grid on small inputs.
template <typename T>
void HybridSearchBrute(const T* points, int64_t num_points, const T* queries,
                       int64_t num_queries, double radius_d, int max_knn,
                       int32_t* indices_ptr, T* distances_ptr,
                       int32_t* counts_ptr) {
    const T radius = (T)radius_d;
    T radius_squared = radius * radius;
    std::vector<std::pair<T, int64_t>> m;
    for (int64_t i = 0; i < num_queries; ++i) {
        m.clear();
        const T* q = queries + 3 * i;
        for (int64_t j = 0; j < num_points; ++j) {
            const T* p = points + 3 * j;
            T result = T();
            const T d0 = q[0] - p[0];
            result += d0 * d0;
            const T d1 = q[1] - p[1];
            result += d1 * d1;
            const T d2 = q[2] - p[2];
            result += d2 * d2;
            if (result < radius_squared) m.emplace_back(result, j);
        }
        std::sort(m.begin(), m.end());
        int32_t c = (int32_t)m.size();
        c = c < max_knn ? c : max_knn;
        counts_ptr[i] = c;
        int k = 0;
        for (; k < c; ++k) {
            indices_ptr[i * max_knn + k] = (int32_t)m[k].second;
            distances_ptr[i * max_knn + k] = m[k].first;
        }
        for (; k < max_knn; ++k) {
            indices_ptr[i * max_knn + k] = -1;
            distances_ptr[i * max_knn + k] = 0;
        }
    }
}

// NearestNeighborSearch::KnnSearch on CPU = NanoFlannIndex::SearchKnn ->
// _KnnSearchCPU (core/nns/NanoFlannImpl.h:129-203): num_neighbors =
// min(dataset size, knn) per query, ascending by distance (nanoflann
// KNNResultSet; nanoflann v1.5.0 is not vendored, tie order follows its tree
// traversal -- here ties go to the lower index). L2_Adaptor distance
// arithmetic as in HybridSearch above. Exhaustive scan, OpenMP over queries.
template <typename T>
void KnnSearchBrute(const T* points, int64_t num_points, const T* queries,
                    int64_t num_queries, int knn, int32_t* indices_ptr,
                    T* distances_ptr) {
    const int64_t k = num_points < (int64_t)knn ? num_points : (int64_t)knn;
#pragma omp parallel
    {
        std::vector<std::pair<T, int64_t>> m((size_t)num_points);
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < num_queries; ++i) {
            const T* q = queries + 3 * i;
            for (int64_t j = 0; j < num_points; ++j) {
                const T* p = points + 3 * j;
                T result = T();
                const T d0 = q[0] - p[0];
                result += d0 * d0;
                const T d1 = q[1] - p[1];
                result += d1 * d1;
                const T d2 = q[2] - p[2];
                result += d2 * d2;
                m[(size_t)j] = {result, j};
            }
            std::partial_sort(m.begin(), m.begin() + k, m.end());
            for (int64_t c = 0; c < k; ++c) {
                indices_ptr[i * k + c] = (int32_t)m[(size_t)c].second;
                if (distances_ptr) distances_ptr[i * k + c] = m[(size_t)c].first;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// PointCloud::VoxelDownSample, PointCloud.cpp:496-567, sequential semantics:
// voxel label = order of first occurrence; every attribute is summed in
// Float32 in point order (IndexAdd_), divided by the Float32 count, cast back.
template <typename T>
int64_t VoxelDownSample(const T* positions, const T* normals, int64_t n,
                        double voxel_size, T* out_positions, T* out_normals) {
    std::unordered_map<CellKey, int64_t, CellKeyHash> map;
    map.reserve((size_t)n);
    std::vector<int64_t> point2voxel((size_t)n);
    const T vs = (T)voxel_size;  // Tensor / scalar: scalar cast to dtype
    int64_t num_voxels = 0;
    for (int64_t i = 0; i < n; ++i) {
        CellKey k{(int64_t)std::floor(positions[3 * i + 0] / vs),
                  (int64_t)std::floor(positions[3 * i + 1] / vs),
                  (int64_t)std::floor(positions[3 * i + 2] / vs)};
        auto res = map.insert({k, num_voxels});
        if (res.second) ++num_voxels;
        point2voxel[(size_t)i] = res.first->second;
    }
    std::vector<float> cnt((size_t)num_voxels, 0.f);
    std::vector<float> sp((size_t)num_voxels * 3, 0.f);
    std::vector<float> sn(normals ? (size_t)num_voxels * 3 : 0, 0.f);
    for (int64_t i = 0; i < n; ++i) {
        int64_t v = point2voxel[(size_t)i];
        cnt[(size_t)v] += 1.0f;
        for (int c = 0; c < 3; ++c) {
            sp[(size_t)v * 3 + c] += (float)positions[3 * i + c];
            if (normals) sn[(size_t)v * 3 + c] += (float)normals[3 * i + c];
        }
    }
    for (int64_t v = 0; v < num_voxels; ++v) {
        for (int c = 0; c < 3; ++c) {
            out_positions[3 * v + c] = (T)(sp[(size_t)v * 3 + c] / cnt[(size_t)v]);
            if (normals)
                out_normals[3 * v + c] =
                        (T)(sn[(size_t)v * 3 + c] / cnt[(size_t)v]);
        }
    }
    return num_voxels;
}

// ---------------------------------------------------------------------------
// ComputeRegistrationResult, Registration.cpp:24-62.
struct RegResult {
    double T[16];
    double fitness = 0, inlier_rmse = 0;
    bool converged = false;
    int num_iterations = 0;
    std::vector<int64_t> correspondences;
};

void Eye4(double* T) {
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

void Matmul4(const double* A, const double* B, double* C) {
    double R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            R[i * 4 + j] = s;
        }
    std::memcpy(C, R, sizeof(R));
}

// The reference sums the squared distances in the tensor dtype
// (distances.Sum({0}), Registration.cpp:44-45) -- in Float32 that sum carries
// ~1e-5 relative rounding that depends on the reduction engine's chunking, and
// the driver's convergence rule (|d rmse| < relative_rmse) sits on it. With
// the drivers' accumulate_double switch the sum is Float64 as well (per-term
// arithmetic unchanged), so that the float64-accumulating oracle is free of
// summation-order knife edges end to end; the Float32 form stays the default.
static thread_local bool g_sum_distances_double = false;

template <typename T>
RegResult ComputeRegistrationResult(const T* source, int64_t ns,
                                    const T* target, int64_t nt,
                                    double max_correspondence_distance,
                                    const double* transformation) {
    RegResult result;
    std::memcpy(result.T, transformation, sizeof(result.T));
    std::vector<int32_t> idx((size_t)ns), counts((size_t)ns);
    std::vector<T> distances((size_t)ns);
    HybridSearch<T>(target, nt, source, ns, max_correspondence_distance, 1,
                    idx.data(), distances.data(), counts.data());
    result.correspondences.resize((size_t)ns);
    // counts.Sum / distances.Sum: reduction in the tensor dtype
    // (int32 / T), then cast to Float64 (Registration.cpp:38-46).
    int64_t num = 0;
    T sq = 0;
    double sq64 = 0;
    for (int64_t i = 0; i < ns; ++i) {
        result.correspondences[(size_t)i] = idx[(size_t)i];
        num += counts[(size_t)i];
        sq += distances[(size_t)i];
        sq64 += (double)distances[(size_t)i];
    }
    double num_correspondences = (double)num;
    if (num_correspondences != 0) {
        const double squared_error =
                g_sum_distances_double ? sq64 : (double)sq;
        result.fitness = num_correspondences / static_cast<double>(ns);
        result.inlier_rmse = std::sqrt(squared_error / num_correspondences);
    } else {
        result.fitness = 0.0;
        result.inlier_rmse = 0.0;
        Eye4(result.T);
    }
    return result;
}

// ---------------------------------------------------------------------------
// TransformationEstimationSymmetric (Rusinkiewicz 2019).
// GetJacobianSymmetric, RegistrationImpl.h:323-386: n = n_t + sign(n_s . n_t)
// n_s; Jacobian and the right-hand-side residual about the correspondence
// means, the robust weight from the un-centred ("objective") residual.
template <typename scalar_t>
bool GetJacobianSymmetric(int64_t workload_idx,
                          const scalar_t* source_points_ptr,
                          const scalar_t* target_points_ptr,
                          const scalar_t* source_normals_ptr,
                          const scalar_t* target_normals_ptr,
                          const int64_t* correspondence_indices,
                          const scalar_t* source_mean_ptr,
                          const scalar_t* target_mean_ptr, scalar_t* J_ij,
                          scalar_t& centered_residual,
                          scalar_t& objective_residual) {
    if (correspondence_indices[workload_idx] == -1) return false;
    const int64_t target_idx = 3 * correspondence_indices[workload_idx];
    const int64_t source_idx = 3 * workload_idx;
    const scalar_t& sx = source_points_ptr[source_idx + 0];
    const scalar_t& sy = source_points_ptr[source_idx + 1];
    const scalar_t& sz = source_points_ptr[source_idx + 2];
    const scalar_t& tx = target_points_ptr[target_idx + 0];
    const scalar_t& ty = target_points_ptr[target_idx + 1];
    const scalar_t& tz = target_points_ptr[target_idx + 2];
    const scalar_t normal_dot =
            source_normals_ptr[source_idx + 0] * target_normals_ptr[target_idx + 0] +
            source_normals_ptr[source_idx + 1] * target_normals_ptr[target_idx + 1] +
            source_normals_ptr[source_idx + 2] * target_normals_ptr[target_idx + 2];
    const scalar_t normal_sign =
            normal_dot < scalar_t(0) ? scalar_t(-1) : scalar_t(1);
    const scalar_t nx = target_normals_ptr[target_idx + 0] +
                        normal_sign * source_normals_ptr[source_idx + 0];
    const scalar_t ny = target_normals_ptr[target_idx + 1] +
                        normal_sign * source_normals_ptr[source_idx + 1];
    const scalar_t nz = target_normals_ptr[target_idx + 2] +
                        normal_sign * source_normals_ptr[source_idx + 2];
    const scalar_t sx_centered = sx - source_mean_ptr[0];
    const scalar_t sy_centered = sy - source_mean_ptr[1];
    const scalar_t sz_centered = sz - source_mean_ptr[2];
    const scalar_t tx_centered = tx - target_mean_ptr[0];
    const scalar_t ty_centered = ty - target_mean_ptr[1];
    const scalar_t tz_centered = tz - target_mean_ptr[2];
    const scalar_t sum_x = sx_centered + tx_centered;
    const scalar_t sum_y = sy_centered + ty_centered;
    const scalar_t sum_z = sz_centered + tz_centered;
    J_ij[0] = sum_y * nz - sum_z * ny;
    J_ij[1] = sum_z * nx - sum_x * nz;
    J_ij[2] = sum_x * ny - sum_y * nx;
    J_ij[3] = nx;
    J_ij[4] = ny;
    J_ij[5] = nz;
    centered_residual = (sx_centered - tx_centered) * nx +
                        (sy_centered - ty_centered) * ny +
                        (sz_centered - tz_centered) * nz;
    objective_residual = (sx - tx) * nx + (sy - ty) * ny + (sz - tz) * nz;
    return true;
}

// ComputePoseSymmetricKernelCPU, RegistrationCPU.cpp:124-180 (one sequential
// range); A[27] is the sum of squared objective residuals here.
template <typename scalar_t, typename acc_t>
void ComputePoseSymmetricKernel(const scalar_t* source_points_ptr,
                                const scalar_t* target_points_ptr,
                                const scalar_t* source_normals_ptr,
                                const scalar_t* target_normals_ptr,
                                const int64_t* correspondence_indices,
                                const scalar_t* source_mean_ptr,
                                const scalar_t* target_mean_ptr, int64_t n,
                                acc_t* global_sum, int method, double scaling,
                                double shape) {
    acc_t A[29];
    for (int i = 0; i < 29; ++i) A[i] = 0;
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        scalar_t J_ij[6] = {0};
        scalar_t centered_residual = 0;
        scalar_t objective_residual = 0;
        const bool valid = GetJacobianSymmetric<scalar_t>(
                workload_idx, source_points_ptr, target_points_ptr,
                source_normals_ptr, target_normals_ptr, correspondence_indices,
                source_mean_ptr, target_mean_ptr, J_ij, centered_residual,
                objective_residual);
        if (valid) {
            const scalar_t weight = RobustWeight<scalar_t>(
                    method, scaling, shape, objective_residual);
            int i = 0;
            for (int j = 0; j < 6; ++j) {
                for (int k = 0; k <= j; ++k) {
                    A[i++] += J_ij[j] * weight * J_ij[k];
                }
                A[21 + j] += J_ij[j] * weight * centered_residual;
            }
            A[27] += objective_residual * objective_residual;
            A[28] += 1;
        }
    }
    for (int i = 0; i < 29; ++i) global_sum[i] = A[i];
}

// TransformSymmetricPoseToMatrix4d, pipelines/registration/SymmetricICPImpl.h:
// 19-44: pose = (g, t'); theta = atan(|g|); half rotation = AngleAxis(theta,
// g / |g|) (Eigen's toRotationMatrix = Rodrigues' formula, Eigen is not
// vendored); R = H H; t = target_mean + H (t' cos theta) - R source_mean.
void TransformSymmetricPoseToMatrix4d(const double* pose,
                                      const double* source_mean,
                                      const double* target_mean, double* T) {
    const double g_norm = std::sqrt(pose[0] * pose[0] + pose[1] * pose[1] +
                                    pose[2] * pose[2]);
    const double theta = std::atan(g_norm);
    double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (g_norm > 0.0) {
        const double ax = pose[0] / g_norm, ay = pose[1] / g_norm,
                     az = pose[2] / g_norm;
        const double c = std::cos(theta), s = std::sin(theta), v = 1.0 - c;
        H[0] = c + v * ax * ax;
        H[1] = v * ax * ay - s * az;
        H[2] = v * ax * az + s * ay;
        H[3] = v * ax * ay + s * az;
        H[4] = c + v * ay * ay;
        H[5] = v * ay * az - s * ax;
        H[6] = v * ax * az - s * ay;
        H[7] = v * ay * az + s * ax;
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

// ComputeTransformationSymmetric, t/pipelines/kernel/Registration.cpp:80-135:
// means of the matched points in the point dtype (Tensor::Mean; with
// accumulate_double the sums are float64 and only the mean is rounded to the
// dtype), 29 sums, DecodeAndSolve6x6, PoseToSymmetricTransformation. Returns
// 0, or 2 when the 6x6 system is singular; no correspondence -> identity.
template <typename T>
int ComputeTransformationSymmetric(const T* src, const T* tgt, const T* sn,
                                   const T* tn, const int64_t* corr, int64_t n,
                                   int method, double scaling, double shape,
                                   int accumulate_double, double* T16,
                                   double* sums29_out) {
    for (int i = 0; i < 16; ++i) T16[i] = (i % 5 == 0) ? 1.0 : 0.0;
    T ms[3], mt[3];
    int64_t cnt = 0;
    if (accumulate_double || sizeof(T) == 8) {
        double a[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t w = 0; w < n; ++w)
            if (corr[w] != -1) {
                for (int k = 0; k < 3; ++k) {
                    a[k] += src[3 * w + k];
                    a[3 + k] += tgt[3 * corr[w] + k];
                }
                ++cnt;
            }
        if (cnt == 0) return 0;
        for (int k = 0; k < 3; ++k) {
            ms[k] = (T)(a[k] / (double)cnt);
            mt[k] = (T)(a[3 + k] / (double)cnt);
        }
    } else {
        T a[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t w = 0; w < n; ++w)
            if (corr[w] != -1) {
                for (int k = 0; k < 3; ++k) {
                    a[k] += src[3 * w + k];
                    a[3 + k] += tgt[3 * corr[w] + k];
                }
                ++cnt;
            }
        if (cnt == 0) return 0;
        for (int k = 0; k < 3; ++k) {
            ms[k] = a[k] / (T)cnt;
            mt[k] = a[3 + k] / (T)cnt;
        }
    }
    double A[29];
    if (accumulate_double || sizeof(T) == 8) {
        ComputePoseSymmetricKernel<T, double>(src, tgt, sn, tn, corr, ms, mt, n,
                                              A, method, scaling, shape);
    } else {
        T Af[29];
        ComputePoseSymmetricKernel<T, T>(src, tgt, sn, tn, corr, ms, mt, n, Af,
                                         method, scaling, shape);
        for (int i = 0; i < 29; ++i) A[i] = (double)Af[i];
    }
    if (sums29_out) std::memcpy(sums29_out, A, sizeof(A));
    double pose[6];
    float residual;
    int inlier_count;
    int st = DecodeAndSolve6x6(A, pose, &residual, &inlier_count) != 0 ? 2 : 0;
    const double msd[3] = {(double)ms[0], (double)ms[1], (double)ms[2]};
    const double mtd[3] = {(double)mt[0], (double)mt[1], (double)mt[2]};
    TransformSymmetricPoseToMatrix4d(pose, msd, mtd, T16);
    return st;
}

// ---------------------------------------------------------------------------
// TransformationEstimationForColoredICP (Park et al. 2017).
// GetJacobianColoredICP, RegistrationImpl.h:388-466: geometric term
// sqrt(lambda) (vs - vt) . nt, photometric term sqrt(1 - lambda) (Is - Is_proj)
// with the source point projected onto the target's tangent plane and the
// target intensity extrapolated by its colour gradient. The "/ 3.0" promotes
// the intensity means to float64 before they are narrowed to scalar_t.
template <typename scalar_t>
bool GetJacobianColoredICP(int64_t workload_idx,
                           const scalar_t* source_points_ptr,
                           const scalar_t* source_colors_ptr,
                           const scalar_t* target_points_ptr,
                           const scalar_t* target_normals_ptr,
                           const scalar_t* target_colors_ptr,
                           const scalar_t* target_color_gradients_ptr,
                           const int64_t* correspondence_indices,
                           const scalar_t& sqrt_lambda_geometric,
                           const scalar_t& sqrt_lambda_photometric,
                           scalar_t* J_G, scalar_t* J_I, scalar_t& r_G,
                           scalar_t& r_I) {
    if (correspondence_indices[workload_idx] == -1) return false;
    const int64_t target_idx = 3 * correspondence_indices[workload_idx];
    const int64_t source_idx = 3 * workload_idx;
    const scalar_t vs[3] = {source_points_ptr[source_idx],
                            source_points_ptr[source_idx + 1],
                            source_points_ptr[source_idx + 2]};
    const scalar_t vt[3] = {target_points_ptr[target_idx],
                            target_points_ptr[target_idx + 1],
                            target_points_ptr[target_idx + 2]};
    const scalar_t nt[3] = {target_normals_ptr[target_idx],
                            target_normals_ptr[target_idx + 1],
                            target_normals_ptr[target_idx + 2]};
    const scalar_t d = (vs[0] - vt[0]) * nt[0] + (vs[1] - vt[1]) * nt[1] +
                       (vs[2] - vt[2]) * nt[2];
    J_G[0] = sqrt_lambda_geometric * (-vs[2] * nt[1] + vs[1] * nt[2]);
    J_G[1] = sqrt_lambda_geometric * (vs[2] * nt[0] - vs[0] * nt[2]);
    J_G[2] = sqrt_lambda_geometric * (-vs[1] * nt[0] + vs[0] * nt[1]);
    J_G[3] = sqrt_lambda_geometric * nt[0];
    J_G[4] = sqrt_lambda_geometric * nt[1];
    J_G[5] = sqrt_lambda_geometric * nt[2];
    r_G = sqrt_lambda_geometric * d;
    const scalar_t vs_proj[3] = {vs[0] - d * nt[0], vs[1] - d * nt[1],
                                 vs[2] - d * nt[2]};
    const scalar_t intensity_source =
            (source_colors_ptr[source_idx] + source_colors_ptr[source_idx + 1] +
             source_colors_ptr[source_idx + 2]) /
            3.0;
    const scalar_t intensity_target =
            (target_colors_ptr[target_idx] + target_colors_ptr[target_idx + 1] +
             target_colors_ptr[target_idx + 2]) /
            3.0;
    const scalar_t dit[3] = {target_color_gradients_ptr[target_idx],
                             target_color_gradients_ptr[target_idx + 1],
                             target_color_gradients_ptr[target_idx + 2]};
    const scalar_t is_proj = dit[0] * (vs_proj[0] - vt[0]) +
                             dit[1] * (vs_proj[1] - vt[1]) +
                             dit[2] * (vs_proj[2] - vt[2]) + intensity_target;
    const scalar_t s = dit[0] * nt[0] + dit[1] * nt[1] + dit[2] * nt[2];
    const scalar_t ditM[3] = {s * nt[0] - dit[0], s * nt[1] - dit[1],
                              s * nt[2] - dit[2]};
    J_I[0] = sqrt_lambda_photometric * (-vs[2] * ditM[1] + vs[1] * ditM[2]);
    J_I[1] = sqrt_lambda_photometric * (vs[2] * ditM[0] - vs[0] * ditM[2]);
    J_I[2] = sqrt_lambda_photometric * (-vs[1] * ditM[0] + vs[0] * ditM[1]);
    J_I[3] = sqrt_lambda_photometric * ditM[0];
    J_I[4] = sqrt_lambda_photometric * ditM[1];
    J_I[5] = sqrt_lambda_photometric * ditM[2];
    r_I = sqrt_lambda_photometric * (intensity_source - is_proj);
    return true;
}

// ComputePoseColoredICPKernelCPU, RegistrationCPU.cpp:220-290 (one range).
template <typename scalar_t, typename acc_t>
void ComputePoseColoredICPKernel(
        const scalar_t* source_points_ptr, const scalar_t* source_colors_ptr,
        const scalar_t* target_points_ptr, const scalar_t* target_normals_ptr,
        const scalar_t* target_colors_ptr,
        const scalar_t* target_color_gradients_ptr,
        const int64_t* correspondence_indices, double lambda_geometric,
        int64_t n, acc_t* global_sum, int method, double scaling,
        double shape) {
    // ComputePoseColoredICPCPU :310-313
    const scalar_t sqrt_lambda_geometric =
            static_cast<scalar_t>(std::sqrt(lambda_geometric));
    const scalar_t sqrt_lambda_photometric =
            static_cast<scalar_t>(std::sqrt(1.0 - lambda_geometric));
    acc_t A[29];
    for (int i = 0; i < 29; ++i) A[i] = 0;
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        scalar_t J_G[6] = {0}, J_I[6] = {0};
        scalar_t r_G = 0, r_I = 0;
        bool valid = GetJacobianColoredICP<scalar_t>(
                workload_idx, source_points_ptr, source_colors_ptr,
                target_points_ptr, target_normals_ptr, target_colors_ptr,
                target_color_gradients_ptr, correspondence_indices,
                sqrt_lambda_geometric, sqrt_lambda_photometric, J_G, J_I, r_G,
                r_I);
        scalar_t w_G = RobustWeight<scalar_t>(method, scaling, shape, r_G);
        scalar_t w_I = RobustWeight<scalar_t>(method, scaling, shape, r_I);
        if (valid) {
            int i = 0;
            for (int j = 0; j < 6; ++j) {
                for (int k = 0; k <= j; ++k) {
                    A[i] += J_G[j] * w_G * J_G[k] + J_I[j] * w_I * J_I[k];
                    ++i;
                }
                A[21 + j] += J_G[j] * w_G * r_G + J_I[j] * w_I * r_I;
            }
            A[27] += r_G * r_G + r_I * r_I;
            A[28] += 1;
        }
    }
    for (int i = 0; i < 29; ++i) global_sum[i] = A[i];
}

// ---------------------------------------------------------------------------
// GetInformationMatrix, Registration.cpp:446-486 ->
// ComputeInformationMatrixCPU, RegistrationCPU.cpp:652-735, with
// GetInformationJacobians, RegistrationImpl.h:686-715. T = point dtype (each
// term J_x[j] J_x[k] + J_y[j] J_y[k] + J_z[j] J_z[k] is formed in T), ACC =
// accumulator type (T in the reference).
template <typename T, typename ACC>
void ComputeInformationMatrixKernel(const T* target_points_ptr,
                                    const int64_t* correspondence_indices,
                                    int64_t n, ACC* global_sum) {
    ACC AtA[21];
    for (int i = 0; i < 21; ++i) AtA[i] = 0;
    for (int64_t w = 0; w < n; ++w) {
        T J_x[6] = {0}, J_y[6] = {0}, J_z[6] = {0};
        if (correspondence_indices[w] == -1) continue;
        const int64_t target_idx = 3 * correspondence_indices[w];
        J_x[0] = J_x[4] = J_x[5] = 0.0;
        J_x[1] = target_points_ptr[target_idx + 2];
        J_x[2] = -target_points_ptr[target_idx + 1];
        J_x[3] = 1.0;
        J_y[1] = J_y[3] = J_y[5] = 0.0;
        J_y[0] = -target_points_ptr[target_idx + 2];
        J_y[2] = target_points_ptr[target_idx];
        J_y[4] = 1.0;
        J_z[2] = J_z[3] = J_z[4] = 0.0;
        J_z[0] = target_points_ptr[target_idx + 1];
        J_z[1] = -target_points_ptr[target_idx];
        J_z[5] = 1.0;
        int i = 0;
        for (int j = 0; j < 6; ++j) {
            for (int k = 0; k <= j; ++k) {
                AtA[i] += J_x[j] * J_x[k] + J_y[j] * J_y[k] + J_z[j] * J_z[k];
                ++i;
            }
        }
    }
    for (int i = 0; i < 21; ++i) global_sum[i] = AtA[i];
}

template <typename T>
int InformationMatrix(const T* source_in, int64_t ns, const T* target,
                      int64_t nt, double max_dist, const double* transformation,
                      int accumulate_double, double* GTG) {
    std::vector<T> source(source_in, source_in + 3 * ns);
    TransformPoints<T>(transformation, source.data(), ns);
    std::vector<int32_t> idx((size_t)ns), counts((size_t)ns);
    std::vector<T> distances((size_t)ns);
    HybridSearch<T>(target, nt, source.data(), ns, max_dist, 1, idx.data(),
                    distances.data(), counts.data());
    std::vector<int64_t> corr((size_t)ns);
    int64_t num = 0;
    for (int64_t i = 0; i < ns; ++i) {
        corr[(size_t)i] = idx[(size_t)i];
        num += counts[(size_t)i];
    }
    if (num == 0) return 1;  // "0 correspondence present ..."
    double sum[21];
    if (accumulate_double) {
        ComputeInformationMatrixKernel<T, double>(target, corr.data(), ns, sum);
    } else {
        T sf[21];
        ComputeInformationMatrixKernel<T, T>(target, corr.data(), ns, sf);
        for (int i = 0; i < 21; ++i) sum[i] = (double)sf[i];
    }
    int i = 0;
    for (int j = 0; j < 6; j++)
        for (int k = 0; k <= j; k++) {
            GTG[j * 6 + k] = GTG[k * 6 + j] = sum[i];
            ++i;
        }
    return 0;
}

// ---------------------------------------------------------------------------
// TransformationEstimationPointToPoint
// Get3x3SxyLinearSystem, t/pipelines/kernel/RegistrationCPU.cpp:495-617: means
// of the matched source / target points (first reduction, :509-552), then the
// centred cross products Sxy[j][k] = sum (t_j - mt_j)(s_k - ms_k) / count
// (second reduction, :555-591, index i = 3 col + row with row = source
// component, col = target component). T = point dtype (products and
// differences in T as the reference), ACC = accumulator type (the reference
// accumulates in T through tbb::parallel_reduce; one left-to-right pass is one
// valid schedule of it).
template <typename T, typename ACC>
int64_t Get3x3Sxy(const T* source_points_ptr, const T* target_points_ptr,
                  const int64_t* correspondence_indices, int64_t n, ACC* Sxy,
                  ACC* source_mean, ACC* target_mean) {
    ACC mean_1x7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t w = 0; w < n; ++w) {
        if (correspondence_indices[w] != -1) {
            int64_t target_idx = 3 * correspondence_indices[w];
            mean_1x7[0] += source_points_ptr[3 * w];
            mean_1x7[1] += source_points_ptr[3 * w + 1];
            mean_1x7[2] += source_points_ptr[3 * w + 2];
            mean_1x7[3] += target_points_ptr[target_idx];
            mean_1x7[4] += target_points_ptr[target_idx + 1];
            mean_1x7[5] += target_points_ptr[target_idx + 2];
            mean_1x7[6] += 1;
        }
    }
    if (mean_1x7[6] == 0) return 0;  // "No valid correspondence present."
    for (int i = 0; i < 6; ++i) mean_1x7[i] = mean_1x7[i] / mean_1x7[6];
    ACC sxy_1x9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t w = 0; w < n; ++w) {
        if (correspondence_indices[w] != -1) {
            for (int i = 0; i < 9; ++i) {
                const int row = i % 3;
                const int col = i / 3;
                const int64_t source_idx = 3 * w + row;
                const int64_t target_idx = 3 * correspondence_indices[w] + col;
                sxy_1x9[i] += (source_points_ptr[source_idx] - mean_1x7[row]) *
                              (target_points_ptr[target_idx] - mean_1x7[3 + col]);
            }
        }
    }
    int i = 0;
    for (int j = 0; j < 3; ++j) {
        for (int k = 0; k < 3; ++k) {
            Sxy[j * 3 + k] = sxy_1x9[i] / mean_1x7[6];
            ++i;
        }
        source_mean[j] = mean_1x7[j];
        target_mean[j] = mean_1x7[j + 3];
    }
    return (int64_t)mean_1x7[6];
}

// Tensor::SVD is LAPACK ?gesvd (core/linalg/SVD.cpp), absent from
// /root/reference: restated as the textbook route for a 3x3 matrix --
// eigen-decomposition of S^T S by cyclic Jacobi rotations (V, sigma^2), then
// u_i = S v_i / sigma_i, completed to an orthonormal basis when sigma_i = 0.
// For a non-degenerate S the product U diag(1,1,det U det V) V^T below does not
// depend on which SVD routine produced U and V; tests pin this against
// numpy.linalg.svd (also LAPACK).
template <typename S>
void SVD3x3(const S* A, S* U, S* D, S* V) {
    S B[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S v = 0;
            for (int k = 0; k < 3; ++k) v += A[k * 3 + i] * A[k * 3 + j];
            B[i * 3 + j] = v;
        }
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? S(1) : S(0);
    for (int sweep = 0; sweep < 64; ++sweep) {
        S off = std::abs(B[1]) + std::abs(B[2]) + std::abs(B[5]);
        S diag = std::abs(B[0]) + std::abs(B[4]) + std::abs(B[8]);
        if (off <= std::numeric_limits<S>::epsilon() * S(1e-3) * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                S apq = B[p * 3 + q];
                if (apq == S(0)) continue;
                S theta = (B[q * 3 + q] - B[p * 3 + p]) / (S(2) * apq);
                S t = (theta >= 0 ? S(1) : S(-1)) /
                      (std::abs(theta) + std::sqrt(theta * theta + S(1)));
                S c = S(1) / std::sqrt(t * t + S(1)), sn = t * c;
                for (int k = 0; k < 3; ++k) {  // B <- B J
                    S bkp = B[k * 3 + p], bkq = B[k * 3 + q];
                    B[k * 3 + p] = c * bkp - sn * bkq;
                    B[k * 3 + q] = sn * bkp + c * bkq;
                }
                for (int k = 0; k < 3; ++k) {  // B <- J^T B
                    S bpk = B[p * 3 + k], bqk = B[q * 3 + k];
                    B[p * 3 + k] = c * bpk - sn * bqk;
                    B[q * 3 + k] = sn * bpk + c * bqk;
                }
                for (int k = 0; k < 3; ++k) {
                    S vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - sn * vkq;
                    V[k * 3 + q] = sn * vkp + c * vkq;
                }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (B[ord[b] * 4] > B[ord[a] * 4]) std::swap(ord[a], ord[b]);
    S Vs[9];
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 3; ++r) Vs[r * 3 + i] = V[r * 3 + ord[i]];
    std::memcpy(V, Vs, sizeof(Vs));
    S u[3][3];
    int have = 0;
    for (int i = 0; i < 3; ++i) {
        S g[3], nn = 0;
        for (int r = 0; r < 3; ++r) {
            g[r] = A[r * 3 + 0] * V[0 * 3 + i] + A[r * 3 + 1] * V[1 * 3 + i] +
                   A[r * 3 + 2] * V[2 * 3 + i];
        }
        for (int r = 0; r < 3; ++r) nn += g[r] * g[r];
        nn = std::sqrt(nn);
        D[i] = nn;
        S lim = (i == 0) ? S(0)
                         : D[0] * std::numeric_limits<S>::epsilon() * S(16);
        if (nn > lim) {
            for (int h = 0; h < have; ++h) {
                S d = g[0] * u[h][0] + g[1] * u[h][1] + g[2] * u[h][2];
                for (int r = 0; r < 3; ++r) g[r] -= d * u[h][r];
            }
            S n2 = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            for (int r = 0; r < 3; ++r) u[i][r] = g[r] / n2;
        } else if (have == 2) {
            u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
            u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
            u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        } else {
            int m = 0;
            if (have == 1) {
                for (int r = 1; r < 3; ++r)
                    if (std::abs(u[0][r]) < std::abs(u[0][m])) m = r;
            }
            S e[3] = {0, 0, 0};
            e[m] = 1;
            for (int h = 0; h < have; ++h) {
                S d = e[0] * u[h][0] + e[1] * u[h][1] + e[2] * u[h][2];
                for (int r = 0; r < 3; ++r) e[r] -= d * u[h][r];
            }
            S n2 = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            for (int r = 0; r < 3; ++r) u[i][r] = e[r] / n2;
        }
        ++have;
    }
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 3; ++r) U[r * 3 + i] = u[i][r];
}

template <typename S>
S Det3(const S* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) -
           M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// ComputeRtPointToPointCPU after the reduction, RegistrationCPU.cpp:640-650:
//   U, D, VT = Sxy.SVD();  S = I;  if det(U) det(VT^T) < 0: S[2][2] = -1
//   R = U S VT;  t = target_mean - R source_mean            (all in dtype S)
template <typename S>
void RtFromSxy(const S* Sxy, const S* source_mean, const S* target_mean,
               double* R9, double* t3) {
    S U[9], D[3], V[9];
    SVD3x3<S>(Sxy, U, D, V);
    S sg = (Det3(U) * Det3(V) < 0) ? S(-1) : S(1);
    S R[9];
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k)
            R[j * 3 + k] = U[j * 3 + 0] * V[k * 3 + 0] +
                           U[j * 3 + 1] * V[k * 3 + 1] +
                           sg * U[j * 3 + 2] * V[k * 3 + 2];
    for (int j = 0; j < 3; ++j) {
        S rt = R[j * 3 + 0] * source_mean[0] + R[j * 3 + 1] * source_mean[1] +
               R[j * 3 + 2] * source_mean[2];
        t3[j] = (double)(S)(target_mean[j] - rt);
    }
    for (int i = 0; i < 9; ++i) R9[i] = (double)R[i];
}

// ComputeRtPointToPoint for point dtype T; accumulate_double widens the
// accumulators, the SVD and R, t to float64 (the comparison target for the
// GPU path, which reduces in float64).
template <typename T>
int64_t ComputeRtPointToPoint(const T* src, const T* tgt, const int64_t* corr,
                              int64_t n, int accumulate_double, double* R9,
                              double* t3) {
    if (accumulate_double || sizeof(T) == 8) {
        double Sxy[9], ms[3], mt[3];
        int64_t c = Get3x3Sxy<T, double>(src, tgt, corr, n, Sxy, ms, mt);
        if (c == 0) return 0;
        RtFromSxy<double>(Sxy, ms, mt, R9, t3);
        return c;
    }
    T Sxy[9], ms[3], mt[3];
    int64_t c = Get3x3Sxy<T, T>(src, tgt, corr, n, Sxy, ms, mt);
    if (c == 0) return 0;
    RtFromSxy<T>(Sxy, ms, mt, R9, t3);
    return c;
}

typedef void (*icp_callback_t)(int64_t iteration_index, int64_t scale_index,
                               int64_t scale_iteration_index, double inlier_rmse,
                               double fitness, const double* transformation,
                               void* user);

// MultiScaleICP, Registration.cpp:362-444 (+ DoSingleScaleICPIterations
// :275-360, InitializePointCloudPyramid :221-273), point-to-plane estimator.
// normals::EstimateColorGradients (defined further down, with the other
// point-attribute kernels).
template <typename T>
void ColorGradientsForIcp(const T* points, const T* normals, const T* colors,
                          const int32_t* indices, const int32_t* counts,
                          int64_t n, int max_nn, T* gradients);

// Optional replacement of the point-to-plane 29-sum inside the driver (same
// signature as oracle/_ref's ref_p2plane_accumulate).
typedef int (*P2PlaneHook)(const void* src, const void* tgt, const void* tgt_n,
                           const int64_t* corr, int64_t n, int is_f64,
                           int method, double scaling, double shape,
                           double* sums29);
static P2PlaneHook g_p2plane_hook = nullptr;

// Attributes some estimators read beyond positions / target normals.
struct IcpExtra {
    const void* source_normals = nullptr;         // symmetric
    const void* source_colors = nullptr;          // colored
    const void* target_colors = nullptr;          // colored
    const void* target_color_gradients = nullptr; // colored, optional
    double lambda_geometric = 0.968;
};

template <typename T>
int MultiScaleICP(const T* source_in, const IcpExtra& extra, int64_t ns_in,
                  const T* target_in, const T* target_normals_in,
                  int64_t nt_in, int num_scales,
                  const double* voxel_sizes, const int* max_iterations,
                  const double* relative_fitness, const double* relative_rmse,
                  const double* max_dists, const double* init, int kernel_method,
                  double kernel_scale, double kernel_shape, int accumulate_double,
                  int estimation, double* out_T, double* out_fitness,
                  double* out_rmse,
                  int* out_converged, int* out_num_iterations,
                  int64_t* out_correspondences, int64_t* out_num_corr,
                  icp_callback_t cb, void* user) {
    struct SumMode {  // scoped: Float64 distance sum with accumulate_double
        bool prev;
        explicit SumMode(bool on) : prev(g_sum_distances_double) {
            g_sum_distances_double = on;
        }
        ~SumMode() { g_sum_distances_double = prev; }
    } sum_mode(accumulate_double != 0);
    // Pyramid.
    std::vector<std::vector<T>> src_p(num_scales), tgt_p(num_scales),
            tgt_n(num_scales), src_n(num_scales), src_c(num_scales),
            tgt_c(num_scales), tgt_g(num_scales);
    const bool with_sn = estimation == 2;  // symmetric: source normals too
    const bool colored = estimation == 3;  // colours + target colour gradients
    const T* source_normals_in = (const T*)extra.source_normals;
    double lambda_geometric = extra.lambda_geometric;
    if (!(lambda_geometric >= 0 && lambda_geometric <= 1.0))
        lambda_geometric = 0.968;
    auto down = [&](const std::vector<T>& p, const std::vector<T>* nrm,
                    double v, std::vector<T>& op, std::vector<T>* on) {
        int64_t n = (int64_t)p.size() / 3;
        op.resize(p.size());
        if (on) on->resize(p.size());
        int64_t m = VoxelDownSample<T>(p.data(), nrm ? nrm->data() : nullptr, n,
                                       v, op.data(), on ? on->data() : nullptr);
        op.resize((size_t)m * 3);
        if (on) on->resize((size_t)m * 3);
    };
    std::vector<T> s0(source_in, source_in + 3 * ns_in);
    std::vector<T> t0(target_in, target_in + 3 * nt_in);
    std::vector<T> n0;
    if (target_normals_in)
        n0.assign(target_normals_in, target_normals_in + 3 * nt_in);
    else
        n0.assign((size_t)(3 * nt_in), T(0));  // point-to-point: unused
    std::vector<T> sn0;
    if (with_sn) sn0.assign(source_normals_in, source_normals_in + 3 * ns_in);
    int last = num_scales - 1;
    if (voxel_sizes[last] <= 0) {
        src_p[last] = s0;
        src_n[last] = sn0;
        tgt_p[last] = t0;
        tgt_n[last] = n0;
    } else {
        down(s0, with_sn ? &sn0 : nullptr, voxel_sizes[last], src_p[last],
             with_sn ? &src_n[last] : nullptr);
        down(t0, &n0, voxel_sizes[last], tgt_p[last], &tgt_n[last]);
    }
    if (colored) {
        // every attribute is averaged by VoxelDownSample (the voxel order is
        // the same whichever attribute rides along)
        std::vector<T> sc0((const T*)extra.source_colors,
                           (const T*)extra.source_colors + 3 * ns_in);
        std::vector<T> tc0((const T*)extra.target_colors,
                           (const T*)extra.target_colors + 3 * nt_in);
        std::vector<T> scratch;
        if (voxel_sizes[last] <= 0) {
            src_c[last] = sc0;
            tgt_c[last] = tc0;
            if (extra.target_color_gradients)
                tgt_g[last].assign(
                        (const T*)extra.target_color_gradients,
                        (const T*)extra.target_color_gradients + 3 * nt_in);
        } else {
            down(s0, &sc0, voxel_sizes[last], scratch, &src_c[last]);
            down(t0, &tc0, voxel_sizes[last], scratch, &tgt_c[last]);
            if (extra.target_color_gradients) {
                std::vector<T> tg0(
                        (const T*)extra.target_color_gradients,
                        (const T*)extra.target_color_gradients + 3 * nt_in);
                down(t0, &tg0, voxel_sizes[last], scratch, &tgt_g[last]);
            }
        }
        if (tgt_g[last].empty()) {
            // Registration.cpp:243-262
            const double radius = voxel_sizes[last] <= 0
                                          ? max_dists[last] * 2.0
                                          : voxel_sizes[last] * 4.0;
            const int64_t m = (int64_t)tgt_p[last].size() / 3;
            std::vector<int32_t> idx((size_t)m * 30), cnt((size_t)m);
            std::vector<T> dist((size_t)m * 30);
            HybridSearch<T>(tgt_p[last].data(), m, tgt_p[last].data(), m,
                            radius, 30, idx.data(), dist.data(), cnt.data());
            tgt_g[last].resize((size_t)m * 3);
            ColorGradientsForIcp<T>(
                    tgt_p[last].data(), tgt_n[last].data(), tgt_c[last].data(),
                    idx.data(), cnt.data(), m, 30, tgt_g[last].data());
        }
    }
    for (int k = num_scales - 2; k >= 0; k--) {
        if (colored) {
            std::vector<T> scratch;
            down(src_p[k + 1], &src_c[k + 1], voxel_sizes[k], scratch,
                 &src_c[k]);
            down(tgt_p[k + 1], &tgt_c[k + 1], voxel_sizes[k], scratch,
                 &tgt_c[k]);
            down(tgt_p[k + 1], &tgt_g[k + 1], voxel_sizes[k], scratch,
                 &tgt_g[k]);
        }
        down(src_p[k + 1], with_sn ? &src_n[k + 1] : nullptr, voxel_sizes[k],
             src_p[k], with_sn ? &src_n[k] : nullptr);
        down(tgt_p[k + 1], &tgt_n[k + 1], voxel_sizes[k], tgt_p[k], &tgt_n[k]);
    }

    RegResult result;
    std::memcpy(result.T, init, sizeof(result.T));
    int iteration_count = 0;
    int status = 0;

    for (int scale_idx = 0; scale_idx < num_scales; ++scale_idx) {
        std::vector<T>& source = src_p[scale_idx];
        const std::vector<T>& target = tgt_p[scale_idx];
        const std::vector<T>& normals = tgt_n[scale_idx];
        int64_t ns = (int64_t)source.size() / 3;
        int64_t nt = (int64_t)target.size() / 3;
        TransformPoints<T>(result.T, source.data(), ns);
        // PointCloud::Transform also rotates the normals (PointCloud.cpp:
        // 352-372); only the symmetric estimator reads the source's.
        std::vector<T>& source_normals = src_n[scale_idx];
        if (with_sn) TransformNormals<T>(result.T, source_normals.data(), ns);

        // DoSingleScaleICPIterations
        RegResult current_result = result;
        {
            RegResult r2;
            std::memcpy(r2.T, current_result.T, sizeof(r2.T));
            double prev_fitness = current_result.fitness;
            double prev_inlier_rmse = current_result.inlier_rmse;
            int it = 0;
            bool early_return = false;
            for (it = 0; it < max_iterations[scale_idx]; ++it) {
                double keepT[16];
                std::memcpy(keepT, r2.T, sizeof(keepT));
                r2 = ComputeRegistrationResult<T>(source.data(), ns,
                                                  target.data(), nt,
                                                  max_dists[scale_idx], keepT);
                if (r2.fitness <= std::numeric_limits<double>::min()) {
                    r2.converged = false;
                    r2.num_iterations = it;
                    early_return = true;
                    break;
                }
                if (estimation == 3) {
                    // TransformationEstimationForColoredICP::
                    // ComputeTransformation, TransformationEstimation.cpp:
                    // 380-432.
                    double A[29];
                    if (accumulate_double) {
                        ComputePoseColoredICPKernel<T, double>(
                                source.data(), src_c[scale_idx].data(),
                                target.data(), normals.data(),
                                tgt_c[scale_idx].data(),
                                tgt_g[scale_idx].data(),
                                r2.correspondences.data(), lambda_geometric,
                                ns, A, kernel_method, kernel_scale,
                                kernel_shape);
                    } else {
                        T Af[29];
                        ComputePoseColoredICPKernel<T, T>(
                                source.data(), src_c[scale_idx].data(),
                                target.data(), normals.data(),
                                tgt_c[scale_idx].data(),
                                tgt_g[scale_idx].data(),
                                r2.correspondences.data(), lambda_geometric,
                                ns, Af, kernel_method, kernel_scale,
                                kernel_shape);
                        for (int i = 0; i < 29; ++i) A[i] = (double)Af[i];
                    }
                    double pose[6], update[16];
                    float residual;
                    int inlier_count;
                    if (DecodeAndSolve6x6(A, pose, &residual, &inlier_count) !=
                        0)
                        status = 2;
                    PoseToTransformation(pose, update);
                    Matmul4(update, r2.T, r2.T);
                    TransformPoints<T>(update, source.data(), ns);
                    if (cb) {
                        cb(iteration_count + it, scale_idx, it, r2.inlier_rmse,
                           r2.fitness, r2.T, user);
                    }
                    if (it != 0 &&
                        std::abs(prev_fitness - r2.fitness) <
                                relative_fitness[scale_idx] &&
                        std::abs(prev_inlier_rmse - r2.inlier_rmse) <
                                relative_rmse[scale_idx]) {
                        r2.converged = true;
                        break;
                    }
                    prev_fitness = r2.fitness;
                    prev_inlier_rmse = r2.inlier_rmse;
                    continue;
                }
                if (estimation == 2) {
                    // TransformationEstimationSymmetric::ComputeTransformation,
                    // TransformationEstimation.cpp:276-292.
                    double update[16];
                    if (ComputeTransformationSymmetric<T>(
                                source.data(), target.data(),
                                source_normals.data(), normals.data(),
                                r2.correspondences.data(), ns, kernel_method,
                                kernel_scale, kernel_shape, accumulate_double,
                                update, nullptr) != 0)
                        status = 2;
                    Matmul4(update, r2.T, r2.T);
                    TransformPoints<T>(update, source.data(), ns);
                    TransformNormals<T>(update, source_normals.data(), ns);
                    if (cb) {
                        cb(iteration_count + it, scale_idx, it, r2.inlier_rmse,
                           r2.fitness, r2.T, user);
                    }
                    if (it != 0 &&
                        std::abs(prev_fitness - r2.fitness) <
                                relative_fitness[scale_idx] &&
                        std::abs(prev_inlier_rmse - r2.inlier_rmse) <
                                relative_rmse[scale_idx]) {
                        r2.converged = true;
                        break;
                    }
                    prev_fitness = r2.fitness;
                    prev_inlier_rmse = r2.inlier_rmse;
                    continue;
                }
                if (estimation == 1) {
                    // TransformationEstimationPointToPoint::
                    // ComputeTransformation, TransformationEstimation.cpp:
                    // 132-160 (RtToTransformation: R, t into a Float64 4x4).
                    double R9[9], t3[3], update[16];
                    ComputeRtPointToPoint<T>(source.data(), target.data(),
                                             r2.correspondences.data(), ns,
                                             accumulate_double, R9, t3);
                    Eye4(update);
                    for (int j = 0; j < 3; ++j) {
                        for (int k = 0; k < 3; ++k)
                            update[j * 4 + k] = R9[j * 3 + k];
                        update[j * 4 + 3] = t3[j];
                    }
                    Matmul4(update, r2.T, r2.T);
                    TransformPoints<T>(update, source.data(), ns);
                    if (cb) {
                        cb(iteration_count + it, scale_idx, it, r2.inlier_rmse,
                           r2.fitness, r2.T, user);
                    }
                    if (it != 0 &&
                        std::abs(prev_fitness - r2.fitness) <
                                relative_fitness[scale_idx] &&
                        std::abs(prev_inlier_rmse - r2.inlier_rmse) <
                                relative_rmse[scale_idx]) {
                        r2.converged = true;
                        break;
                    }
                    prev_fitness = r2.fitness;
                    prev_inlier_rmse = r2.inlier_rmse;
                    continue;
                }
                double A[29];
                if (g_p2plane_hook) {
                    // the 29 sums from an external body (tests: the
                    // reference's own, under a chosen reduction schedule)
                    g_p2plane_hook(source.data(), target.data(),
                                   normals.data(), r2.correspondences.data(),
                                   ns, sizeof(T) == 8, kernel_method,
                                   kernel_scale, kernel_shape, A);
                } else if (accumulate_double) {
                    ComputePosePointToPlaneKernel<T, double>(
                            source.data(), target.data(), normals.data(),
                            r2.correspondences.data(), ns, A, kernel_method,
                            kernel_scale, kernel_shape);
                } else {
                    T Af[29];
                    ComputePosePointToPlaneKernel<T, T>(
                            source.data(), target.data(), normals.data(),
                            r2.correspondences.data(), ns, Af, kernel_method,
                            kernel_scale, kernel_shape);
                    for (int i = 0; i < 29; ++i) A[i] = (double)Af[i];
                }
                double pose[6];
                float residual;
                int inlier_count;
                if (DecodeAndSolve6x6(A, pose, &residual, &inlier_count) != 0) {
                    status = 2;  // singular: the reference throws.
                }
                double update[16];
                PoseToTransformation(pose, update);
                Matmul4(update, r2.T, r2.T);
                TransformPoints<T>(update, source.data(), ns);

                if (cb) {
                    cb(iteration_count + it, scale_idx, it, r2.inlier_rmse,
                       r2.fitness, r2.T, user);
                }
                if (it != 0 &&
                    std::abs(prev_fitness - r2.fitness) <
                            relative_fitness[scale_idx] &&
                    std::abs(prev_inlier_rmse - r2.inlier_rmse) <
                            relative_rmse[scale_idx]) {
                    r2.converged = true;
                    break;
                }
                prev_fitness = r2.fitness;
                prev_inlier_rmse = r2.inlier_rmse;
            }
            (void)early_return;
            result = r2;
            iteration_count = iteration_count + it;
        }

        if (scale_idx == num_scales - 1) {
            bool preserved = result.converged;
            double keepT[16];
            std::memcpy(keepT, result.T, sizeof(keepT));
            result = ComputeRegistrationResult<T>(source.data(), ns,
                                                  target.data(), nt,
                                                  max_dists[scale_idx], keepT);
            result.converged = preserved;
        }
        if (result.fitness <= std::numeric_limits<double>::min()) {
            result.converged = false;
            break;
        }
    }
    result.num_iterations = iteration_count;

    std::memcpy(out_T, result.T, sizeof(result.T));
    *out_fitness = result.fitness;
    *out_rmse = result.inlier_rmse;
    *out_converged = result.converged ? 1 : 0;
    *out_num_iterations = result.num_iterations;
    *out_num_corr = (int64_t)result.correspondences.size();
    if (out_correspondences) {
        for (size_t i = 0; i < result.correspondences.size(); ++i)
            out_correspondences[i] = result.correspondences[i];
    }
    return status;
}

}  // namespace

// ===========================================================================

// ---------------------------------------------------------------------------
// Normal estimation: PointCloud::EstimateNormals with
// hybrid search (t/geometry/PointCloud.cpp:856-976) =
//   EstimateCovariancesUsingHybridSearchCPU   (PointCloudImpl.h:588-638)
//   EstimatePointWiseRobustNormalizedCovarianceKernel (:512-585)
//   EstimateNormalsFromCovariancesCPU         (:1011-1063)
//   EstimatePointWiseNormalsWithFastEigen3x3  (:875-1009), ComputeEigenvector0
//   (:746-793), ComputeEigenvector1 (:795-873); cross / dot / matmul helpers
//   core/linalg/kernel/Matrix.h.
// Mixed float / double literals are kept as written (they promote parts of an
// expression to double). Unqualified sqrt / abs / acos / cos / min / max
// resolve to the std:: float overloads with libstdc++'s <cmath> (checked
// against the compiled reference body in tests/test_oracle_vs_ref.py).
namespace normals {

template <typename T>
void Cross(const T* a, const T* b, T* c) {
    c[0] = (a[1] * b[2]) - (a[2] * b[1]);
    c[1] = (a[2] * b[0]) - (a[0] * b[2]);
    c[2] = (a[0] * b[1]) - (a[1] * b[0]);
}
template <typename T>
T Dot(const T* a, const T* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <typename T>
void Matmul3x3_3x1(const T* A, const T* b, T* c) {
    c[0] = A[0] * b[0] + A[1] * b[1] + A[2] * b[2];
    c[1] = A[3] * b[0] + A[4] * b[1] + A[5] * b[2];
    c[2] = A[6] * b[0] + A[7] * b[1] + A[8] * b[2];
}

template <typename T>
void Covariance(const T* points_ptr, const int32_t* indices_ptr,
                int32_t indices_count, T* covariance_ptr) {
    if (indices_count < 3) {
        for (int i = 0; i < 9; ++i) covariance_ptr[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double centroid[3] = {0};
    for (int32_t i = 0; i < indices_count; ++i) {
        int32_t idx = 3 * indices_ptr[i];
        centroid[0] += points_ptr[idx];
        centroid[1] += points_ptr[idx + 1];
        centroid[2] += points_ptr[idx + 2];
    }
    centroid[0] /= indices_count;
    centroid[1] /= indices_count;
    centroid[2] /= indices_count;
    double cumulants[6] = {0};
    for (int32_t i = 0; i < indices_count; ++i) {
        int32_t idx = 3 * indices_ptr[i];
        const double x = static_cast<double>(points_ptr[idx]) - centroid[0];
        const double y = static_cast<double>(points_ptr[idx + 1]) - centroid[1];
        const double z = static_cast<double>(points_ptr[idx + 2]) - centroid[2];
        cumulants[0] += x * x;
        cumulants[1] += y * y;
        cumulants[2] += z * z;
        cumulants[3] += x * y;
        cumulants[4] += x * z;
        cumulants[5] += y * z;
    }
    const double normalization_factor = static_cast<double>(indices_count - 1);
    for (int i = 0; i < 6; ++i) cumulants[i] /= normalization_factor;
    covariance_ptr[0] = static_cast<T>(cumulants[0]);
    covariance_ptr[4] = static_cast<T>(cumulants[1]);
    covariance_ptr[8] = static_cast<T>(cumulants[2]);
    covariance_ptr[1] = static_cast<T>(cumulants[3]);
    covariance_ptr[3] = covariance_ptr[1];
    covariance_ptr[2] = static_cast<T>(cumulants[4]);
    covariance_ptr[6] = covariance_ptr[2];
    covariance_ptr[5] = static_cast<T>(cumulants[5]);
    covariance_ptr[7] = covariance_ptr[5];
}

template <typename T>
void Eigenvector0(const T* A, const T eval0, T* eigen_vector0) {
    T row0[3] = {A[0] - eval0, A[1], A[2]};
    T row1[3] = {A[1], A[4] - eval0, A[5]};
    T row2[3] = {A[2], A[5], A[8] - eval0};
    T r0xr1[3], r0xr2[3], r1xr2[3];
    Cross(row0, row1, r0xr1);
    Cross(row0, row2, r0xr2);
    Cross(row1, row2, r1xr2);
    T d0 = Dot(r0xr1, r0xr1);
    T d1 = Dot(r0xr2, r0xr2);
    T d2 = Dot(r1xr2, r1xr2);
    T dmax = d0;
    int imax = 0;
    if (d1 > dmax) {
        dmax = d1;
        imax = 1;
    }
    if (d2 > dmax) imax = 2;
    const T* v = imax == 0 ? r0xr1 : (imax == 1 ? r0xr2 : r1xr2);
    T sqrt_d = std::sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
    eigen_vector0[0] = v[0] / sqrt_d;
    eigen_vector0[1] = v[1] / sqrt_d;
    eigen_vector0[2] = v[2] / sqrt_d;
}

template <typename T>
void Eigenvector1(const T* A, const T* evec0, const T eval1,
                  T* eigen_vector1) {
    T U[3];
    if (std::abs(evec0[0]) > std::abs(evec0[1])) {
        T inv_length =
                1.0 / std::sqrt(evec0[0] * evec0[0] + evec0[2] * evec0[2]);
        U[0] = -evec0[2] * inv_length;
        U[1] = 0.0;
        U[2] = evec0[0] * inv_length;
    } else {
        T inv_length =
                1.0 / std::sqrt(evec0[1] * evec0[1] + evec0[2] * evec0[2]);
        U[0] = 0.0;
        U[1] = evec0[2] * inv_length;
        U[2] = -evec0[1] * inv_length;
    }
    T V[3], AU[3], AV[3];
    Cross(evec0, U, V);
    Matmul3x3_3x1(A, U, AU);
    Matmul3x3_3x1(A, V, AV);
    T m00 = Dot(U, AU) - eval1;
    T m01 = Dot(U, AV);
    T m11 = Dot(V, AV) - eval1;
    T absM00 = std::abs(m00);
    T absM01 = std::abs(m01);
    T absM11 = std::abs(m11);
    T max_abs_comp;
    if (absM00 >= absM11) {
        max_abs_comp = std::max(absM00, absM01);
        if (max_abs_comp > 0) {
            if (absM00 >= absM01) {
                m01 /= m00;
                m00 = 1 / std::sqrt(1 + m01 * m01);
                m01 *= m00;
            } else {
                m00 /= m01;
                m01 = 1 / std::sqrt(1 + m00 * m00);
                m00 *= m01;
            }
            eigen_vector1[0] = m01 * U[0] - m00 * V[0];
            eigen_vector1[1] = m01 * U[1] - m00 * V[1];
            eigen_vector1[2] = m01 * U[2] - m00 * V[2];
        } else {
            eigen_vector1[0] = U[0];
            eigen_vector1[1] = U[1];
            eigen_vector1[2] = U[2];
        }
    } else {
        max_abs_comp = std::max(absM11, absM01);
        if (max_abs_comp > 0) {
            if (absM11 >= absM01) {
                m01 /= m11;
                m11 = 1 / std::sqrt(1 + m01 * m01);
                m01 *= m11;
            } else {
                m11 /= m01;
                m01 = 1 / std::sqrt(1 + m11 * m11);
                m11 *= m01;
            }
            eigen_vector1[0] = m11 * U[0] - m01 * V[0];
            eigen_vector1[1] = m11 * U[1] - m01 * V[1];
            eigen_vector1[2] = m11 * U[2] - m01 * V[2];
        } else {
            eigen_vector1[0] = U[0];
            eigen_vector1[1] = U[1];
            eigen_vector1[2] = U[2];
        }
    }
}

template <typename T>
void FastEigen3x3(const T* covariance_ptr, T* normals_ptr) {
    T max_coeff = covariance_ptr[0];
    for (int i = 1; i < 9; ++i)
        if (max_coeff < covariance_ptr[i]) max_coeff = covariance_ptr[i];
    if (max_coeff == 0) {
        normals_ptr[0] = normals_ptr[1] = normals_ptr[2] = 0.0;
        return;
    }
    T A[9] = {0};
    for (int i = 0; i < 9; ++i) A[i] = covariance_ptr[i] / max_coeff;
    T norm = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (norm > 0) {
        T eval[3], evec0[3], evec1[3], evec2[3];
        T q = (A[0] + A[4] + A[8]) / 3.0;
        T b00 = A[0] - q;
        T b11 = A[4] - q;
        T b22 = A[8] - q;
        T p = std::sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2.0) / 6.0);
        T c00 = b11 * b22 - A[5] * A[5];
        T c01 = A[1] * b22 - A[5] * A[2];
        T c02 = A[1] * A[5] - b11 * A[2];
        T det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
        T half_det = det * 0.5;
        half_det = std::min(std::max(half_det, static_cast<T>(-1.0)),
                            static_cast<T>(1.0));
        T angle = std::acos(half_det) / 3.0;
        const T two_thrids_pi = 2.09439510239319549;
        T beta2 = std::cos(angle) * 2.0;
        T beta0 = std::cos(angle + two_thrids_pi) * 2.0;
        T beta1 = -(beta0 + beta2);
        eval[0] = q + p * beta0;
        eval[1] = q + p * beta1;
        eval[2] = q + p * beta2;
        if (half_det >= 0) {
            Eigenvector0<T>(A, eval[2], evec2);
            if (eval[2] < eval[0] && eval[2] < eval[1]) {
                for (int i = 0; i < 3; ++i) normals_ptr[i] = evec2[i];
                return;
            }
            Eigenvector1<T>(A, evec2, eval[1], evec1);
            if (eval[1] < eval[0] && eval[1] < eval[2]) {
                for (int i = 0; i < 3; ++i) normals_ptr[i] = evec1[i];
                return;
            }
            normals_ptr[0] = evec1[1] * evec2[2] - evec1[2] * evec2[1];
            normals_ptr[1] = evec1[2] * evec2[0] - evec1[0] * evec2[2];
            normals_ptr[2] = evec1[0] * evec2[1] - evec1[1] * evec2[0];
        } else {
            Eigenvector0<T>(A, eval[0], evec0);
            if (eval[0] < eval[1] && eval[0] < eval[2]) {
                for (int i = 0; i < 3; ++i) normals_ptr[i] = evec0[i];
                return;
            }
            Eigenvector1<T>(A, evec0, eval[1], evec1);
            if (eval[1] < eval[0] && eval[1] < eval[2]) {
                for (int i = 0; i < 3; ++i) normals_ptr[i] = evec1[i];
                return;
            }
            normals_ptr[0] = evec0[1] * evec1[2] - evec0[2] * evec1[1];
            normals_ptr[1] = evec0[2] * evec1[0] - evec0[0] * evec1[2];
            normals_ptr[2] = evec0[0] * evec1[1] - evec0[1] * evec1[0];
        }
    } else {
        if (covariance_ptr[0] < covariance_ptr[4] &&
            covariance_ptr[0] < covariance_ptr[8]) {
            normals_ptr[0] = 1.0; normals_ptr[1] = 0.0; normals_ptr[2] = 0.0;
        } else if (covariance_ptr[4] < covariance_ptr[0] &&
                   covariance_ptr[4] < covariance_ptr[8]) {
            normals_ptr[0] = 0.0; normals_ptr[1] = 1.0; normals_ptr[2] = 0.0;
        } else {
            normals_ptr[0] = 0.0; normals_ptr[1] = 0.0; normals_ptr[2] = 1.0;
        }
    }
}

template <typename T>
void NormalsFromCovariances(const T* covariances, int64_t n, T* normals_ptr,
                            bool has_normals) {
    for (int64_t w = 0; w < n; ++w) {
        T out[3] = {0};
        FastEigen3x3<T>(covariances + 9 * w, out);
        if ((out[0] * out[0] + out[1] * out[1] + out[2] * out[2]) == 0.0 &&
            !has_normals) {
            out[0] = 0.0; out[1] = 0.0; out[2] = 1.0;
        }
        if (has_normals) {
            if ((normals_ptr[3 * w] * out[0] + normals_ptr[3 * w + 1] * out[1] +
                 normals_ptr[3 * w + 2] * out[2]) < 0.0) {
                out[0] *= -1; out[1] *= -1; out[2] *= -1;
            }
        }
        normals_ptr[3 * w] = out[0];
        normals_ptr[3 * w + 1] = out[1];
        normals_ptr[3 * w + 2] = out[2];
    }
}

// ---------------------------------------------------------------------------
// PointCloud::EstimateColorGradients (t/geometry/PointCloud.cpp:987-1060):
// EstimatePointWiseColorGradientKernel, PointCloudImpl.h:1067-1165, per point
// over its neighbour list (the first neighbour is the point itself): least
// squares fit of the intensity over the neighbours projected on the tangent
// plane, plus the constraint gradient . normal = 0 weighted by the neighbour
// count.
// PARITY NOTE: the reference solves the 3x3 normal equations with
// core::linalg::kernel::solve_svd3x3 (core/linalg/kernel/SVD3x3.h, 2.2 k lines:
// McAdams' approximate SVD -- 4 fixed Jacobi sweeps, pi/8 clamps applied
// through 32-bit masks, pseudo-inverse with 1e-10 cut-off). That routine is
// not restated; x = pinv(AtA) Atb is formed here from a converged Jacobi
// eigen-decomposition of the symmetric AtA with the same 1e-10 cut-off. The
// accumulation of AtA / Atb is the reference's, statement by statement; the
// result agrees with the compiled reference body to the accuracy of its
// approximate SVD (tests/test_oracle_vs_ref.py states the tolerance).
template <typename T>
void PinvSolveSym3(const T* AtA, const T* Atb, T* x) {
    T a[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            a[i][j] = AtA[i * 3 + j];
            V[i][j] = i == j ? T(1) : T(0);
        }
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const T apq = a[p][q];
                if (apq == T(0)) continue;
                const T theta = (a[q][q] - a[p][p]) / (T(2) * apq);
                const T t = (theta >= T(0) ? T(1) : T(-1)) /
                            (std::abs(theta) + std::sqrt(theta * theta + T(1)));
                const T c = T(1) / std::sqrt(t * t + T(1));
                const T sn = t * c;
                for (int k = 0; k < 3; ++k) {  // a <- a J
                    const T akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {  // a <- J^T a
                    const T apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - sn * aqk;
                    a[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const T vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    const T epsilon = (T)1e-10;
    x[0] = x[1] = x[2] = T(0);
    for (int i = 0; i < 3; ++i) {
        const T lam = a[i][i];
        const T inv = std::abs(lam) < epsilon ? T(0) : T(1) / lam;
        const T proj = V[0][i] * Atb[0] + V[1][i] * Atb[1] + V[2][i] * Atb[2];
        const T w = inv * proj;
        x[0] += V[0][i] * w;
        x[1] += V[1][i] * w;
        x[2] += V[2][i] * w;
    }
}

template <typename scalar_t>
void ColorGradient(const scalar_t* points_ptr, const scalar_t* normals_ptr,
                   const scalar_t* colors_ptr, int64_t idx_offset,
                   const int32_t* indices_ptr, int32_t indices_count,
                   scalar_t* color_gradients_ptr, bool exact_solve) {
    if (indices_count < 4) {
        color_gradients_ptr[idx_offset] = 0;
        color_gradients_ptr[idx_offset + 1] = 0;
        color_gradients_ptr[idx_offset + 2] = 0;
        return;
    }
    scalar_t vt[3] = {points_ptr[idx_offset], points_ptr[idx_offset + 1],
                      points_ptr[idx_offset + 2]};
    scalar_t nt[3] = {normals_ptr[idx_offset], normals_ptr[idx_offset + 1],
                      normals_ptr[idx_offset + 2]};
    scalar_t it = (colors_ptr[idx_offset] + colors_ptr[idx_offset + 1] +
                   colors_ptr[idx_offset + 2]) /
                  3.0;
    scalar_t AtA[9] = {0};
    scalar_t Atb[3] = {0};
    scalar_t s = vt[0] * nt[0] + vt[1] * nt[1] + vt[2] * nt[2];
    int i = 1;
    for (; i < indices_count; i++) {
        int64_t neighbour_idx_offset = 3 * (int64_t)indices_ptr[i];
        if (neighbour_idx_offset == -1) break;  // (never true; kept as written)
        scalar_t vt_adj[3] = {points_ptr[neighbour_idx_offset],
                              points_ptr[neighbour_idx_offset + 1],
                              points_ptr[neighbour_idx_offset + 2]};
        scalar_t d = vt_adj[0] * nt[0] + vt_adj[1] * nt[1] +
                     vt_adj[2] * nt[2] - s;
        scalar_t vt_proj[3] = {vt_adj[0] - d * nt[0], vt_adj[1] - d * nt[1],
                               vt_adj[2] - d * nt[2]};
        scalar_t it_adj = (colors_ptr[neighbour_idx_offset + 0] +
                           colors_ptr[neighbour_idx_offset + 1] +
                           colors_ptr[neighbour_idx_offset + 2]) /
                          3.0;
        scalar_t A[3] = {vt_proj[0] - vt[0], vt_proj[1] - vt[1],
                         vt_proj[2] - vt[2]};
        AtA[0] += A[0] * A[0];
        AtA[1] += A[1] * A[0];
        AtA[2] += A[2] * A[0];
        AtA[4] += A[1] * A[1];
        AtA[5] += A[2] * A[1];
        AtA[8] += A[2] * A[2];
        scalar_t b = it_adj - it;
        Atb[0] += A[0] * b;
        Atb[1] += A[1] * b;
        Atb[2] += A[2] * b;
    }
    scalar_t A[3] = {(i - 1) * nt[0], (i - 1) * nt[1], (i - 1) * nt[2]};
    AtA[0] += A[0] * A[0];
    AtA[1] += A[0] * A[1];
    AtA[2] += A[0] * A[2];
    AtA[4] += A[1] * A[1];
    AtA[5] += A[1] * A[2];
    AtA[8] += A[2] * A[2];
    AtA[3] = AtA[1];
    AtA[6] = AtA[2];
    AtA[7] = AtA[5];
    if (exact_solve)
        PinvSolveSym3<scalar_t>(AtA, Atb, color_gradients_ptr + idx_offset);
    else  // core::linalg::kernel::solve_svd3x3, restated in approx_svd3_oracle.h
        svd3::SolveSvd3x3<scalar_t>(AtA, Atb, color_gradients_ptr + idx_offset);
}

template <typename T>
void EstimateColorGradients(const T* points, const T* normals, const T* colors,
                            const int32_t* indices, const int32_t* counts,
                            int64_t n, int max_nn, T* gradients,
                            bool exact_solve = false) {
#pragma omp parallel for schedule(static)
    for (int64_t w = 0; w < n; ++w)
        ColorGradient<T>(points, normals, colors, 3 * w,
                         indices + (int64_t)max_nn * w, counts[w], gradients,
                         exact_solve);
}

}  // namespace normals

namespace {
// Which 3x3 solve the oracle's ICP driver uses when it estimates colour
// gradients itself: the reference's approximate solve_svd3x3 (default) or the
// converged pseudo-inverse the product uses (orc_set_exact_color_gradients).
bool g_exact_color_gradients = false;
template <typename T>
void ColorGradientsForIcp(const T* points, const T* normals_, const T* colors,
                          const int32_t* indices, const int32_t* counts,
                          int64_t n, int max_nn, T* gradients) {
    normals::EstimateColorGradients<T>(points, normals_, colors, indices,
                                       counts, n, max_nn, gradients,
                                       g_exact_color_gradients);
}
}  // namespace

extern "C" {

void orc_set_exact_color_gradients(int on) { g_exact_color_gradients = on != 0; }

// fn = address of a function with ref_p2plane_accumulate's signature, or NULL.
void orc_set_p2plane_hook(void* fn) { g_p2plane_hook = (P2PlaneHook)fn; }

double orc_robust_weight(int is_f64, int method, double scaling, double shape,
                         double residual) {
    if (is_f64) return RobustWeight<double>(method, scaling, shape, residual);
    return (double)RobustWeight<float>(method, scaling, shape, (float)residual);
}

// EstimatePointWiseRobustNormalizedCovarianceKernel over hybrid-search results
// (indices {n, max_nn} int32, counts {n}); covariances {n,3,3}.
void orc_estimate_covariances(const void* points, const int32_t* indices,
                              const int32_t* counts, int64_t n, int max_nn,
                              int is_f64, void* covariances) {
    for (int64_t w = 0; w < n; ++w) {
        if (is_f64)
            normals::Covariance<double>((const double*)points,
                                        indices + (int64_t)max_nn * w, counts[w],
                                        (double*)covariances + 9 * w);
        else
            normals::Covariance<float>((const float*)points,
                                       indices + (int64_t)max_nn * w, counts[w],
                                       (float*)covariances + 9 * w);
    }
}

// EstimateNormalsFromCovariancesCPU; normals is in/out when has_normals.
void orc_normals_from_covariances(const void* covariances, int64_t n,
                                  int is_f64, void* normals_io,
                                  int has_normals) {
    if (is_f64)
        normals::NormalsFromCovariances<double>((const double*)covariances, n,
                                                (double*)normals_io,
                                                has_normals != 0);
    else
        normals::NormalsFromCovariances<float>((const float*)covariances, n,
                                               (float*)normals_io,
                                               has_normals != 0);
}


void orc_hybrid_search(const void* points, int64_t n, const void* queries,
                       int64_t q, int is_f64, double radius, int max_knn,
                       int brute, int32_t* idx, void* dist, int32_t* counts) {
    if (is_f64) {
        if (brute)
            HybridSearchBrute<double>((const double*)points, n,
                                      (const double*)queries, q, radius,
                                      max_knn, idx, (double*)dist, counts);
        else
            HybridSearch<double>((const double*)points, n,
                                 (const double*)queries, q, radius, max_knn,
                                 idx, (double*)dist, counts);
    } else {
        if (brute)
            HybridSearchBrute<float>((const float*)points, n,
                                     (const float*)queries, q, radius, max_knn,
                                     idx, (float*)dist, counts);
        else
            HybridSearch<float>((const float*)points, n, (const float*)queries,
                                q, radius, max_knn, idx, (float*)dist, counts);
    }
}

void orc_estimate_color_gradients(const void* points, const void* normals,
                                  const void* colors, const int32_t* indices,
                                  const int32_t* counts, int64_t n, int max_nn,
                                  int is_f64, int exact_solve,
                                  void* gradients) {
    if (is_f64)
        normals::EstimateColorGradients<double>(
                (const double*)points, (const double*)normals,
                (const double*)colors, indices, counts, n, max_nn,
                (double*)gradients, exact_solve != 0);
    else
        normals::EstimateColorGradients<float>(
                (const float*)points, (const float*)normals,
                (const float*)colors, indices, counts, n, max_nn,
                (float*)gradients, exact_solve != 0);
}

// svd3x3 / solve_svd3x3 restated (approx_svd3_oracle.h): row-major 3x3.
void orc_svd3x3(const void* A, int is_f64, void* U, void* S, void* V) {
    if (is_f64)
        svd3::Svd3x3<double>((const double*)A, (double*)U, (double*)S,
                             (double*)V);
    else
        svd3::Svd3x3<float>((const float*)A, (float*)U, (float*)S, (float*)V);
}
void orc_solve_svd3x3(const void* A, const void* b, int is_f64, void* x) {
    if (is_f64)
        svd3::SolveSvd3x3<double>((const double*)A, (const double*)b,
                                  (double*)x);
    else
        svd3::SolveSvd3x3<float>((const float*)A, (const float*)b, (float*)x);
}

// idx {q, min(knn, n)} int32, dist {q, min(knn, n)} in the point dtype.
void orc_knn_search(const void* points, int64_t n, const void* queries,
                    int64_t q, int is_f64, int knn, int32_t* idx, void* dist) {
    if (is_f64)
        KnnSearchBrute<double>((const double*)points, n, (const double*)queries,
                               q, knn, idx, (double*)dist);
    else
        KnnSearchBrute<float>((const float*)points, n, (const float*)queries, q,
                              knn, idx, (float*)dist);
}

// out29 is always double[29]; accumulate_double selects the accumulator type.
void orc_p2plane_accumulate(const void* src, const void* tgt, const void* tgt_n,
                            const int64_t* corr, int64_t n, int is_f64,
                            int method, double scaling, double shape,
                            int accumulate_double, double* out29) {
    if (is_f64) {
        ComputePosePointToPlaneKernel<double, double>(
                (const double*)src, (const double*)tgt, (const double*)tgt_n,
                corr, n, out29, method, scaling, shape);
    } else if (accumulate_double) {
        ComputePosePointToPlaneKernel<float, double>(
                (const float*)src, (const float*)tgt, (const float*)tgt_n, corr,
                n, out29, method, scaling, shape);
    } else {
        float A[29];
        ComputePosePointToPlaneKernel<float, float>(
                (const float*)src, (const float*)tgt, (const float*)tgt_n, corr,
                n, A, method, scaling, shape);
        for (int i = 0; i < 29; ++i) out29[i] = (double)A[i];
    }
}

int orc_decode_and_solve6x6(const double* A29, double* pose, float* residual,
                            int* count) {
    return DecodeAndSolve6x6(A29, pose, residual, count);
}

int orc_solve(int n, const double* A, const double* b, double* x) {
    std::vector<double> a(A, A + (size_t)n * n), r(b, b + n);
    int info = SolveLU(n, a.data(), r.data());
    if (info) return info;
    for (int i = 0; i < n; ++i) x[i] = r[i];
    return 0;
}

void orc_pose_to_transformation(const double* pose, double* T) {
    PoseToTransformation(pose, T);
}

void orc_transform_points(const double* T, void* pts, int64_t n, int is_f64) {
    if (is_f64) TransformPoints<double>(T, (double*)pts, n);
    else TransformPoints<float>(T, (float*)pts, n);
}

void orc_transform_normals(const double* T, void* nrm, int64_t n, int is_f64) {
    if (is_f64) TransformNormals<double>(T, (double*)nrm, n);
    else TransformNormals<float>(T, (float*)nrm, n);
}

// TransformationEstimationPointToPlane::ComputeRMSE,
// TransformationEstimation.cpp:160-193: element-wise (s-t)*n, squared, summed
// over all elements (in the cloud dtype), / n_corr, sqrt.
double orc_p2plane_rmse(const void* src, const void* tgt, const void* tgt_n,
                        const int64_t* corr, int64_t n, int is_f64) {
    int64_t cnt = 0;
    if (is_f64) {
        const double *s = (const double*)src, *t = (const double*)tgt,
                     *nn = (const double*)tgt_n;
        double e = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (corr[i] == -1) continue;
            ++cnt;
            for (int c = 0; c < 3; ++c) {
                double v = (s[3 * i + c] - t[3 * corr[i] + c]) *
                           nn[3 * corr[i] + c];
                e += v * v;
            }
        }
        return std::sqrt(e / (double)cnt);
    }
    const float *s = (const float*)src, *t = (const float*)tgt,
                *nn = (const float*)tgt_n;
    float e = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (corr[i] == -1) continue;
        ++cnt;
        for (int c = 0; c < 3; ++c) {
            float v = (s[3 * i + c] - t[3 * corr[i] + c]) * nn[3 * corr[i] + c];
            e += v * v;
        }
    }
    return std::sqrt((double)e / (double)cnt);
}

int64_t orc_voxel_down_sample(const void* pos, const void* nrm, int64_t n,
                              int is_f64, double voxel_size, void* out_pos,
                              void* out_nrm) {
    if (is_f64)
        return VoxelDownSample<double>((const double*)pos, (const double*)nrm,
                                       n, voxel_size, (double*)out_pos,
                                       (double*)out_nrm);
    return VoxelDownSample<float>((const float*)pos, (const float*)nrm, n,
                                  voxel_size, (float*)out_pos, (float*)out_nrm);
}

void orc_information_accumulate(const void* tgt, const int64_t* corr,
                                int64_t n, int is_f64, int accumulate_double,
                                double* out21) {
    if (is_f64) {
        ComputeInformationMatrixKernel<double, double>((const double*)tgt, corr,
                                                       n, out21);
    } else if (accumulate_double) {
        ComputeInformationMatrixKernel<float, double>((const float*)tgt, corr,
                                                      n, out21);
    } else {
        float A[21];
        ComputeInformationMatrixKernel<float, float>((const float*)tgt, corr, n,
                                                     A);
        for (int i = 0; i < 21; ++i) out21[i] = (double)A[i];
    }
}

int orc_information_matrix(const void* source, int64_t ns, const void* target,
                           int64_t nt, int is_f64, double max_dist,
                           const double* transformation, int accumulate_double,
                           double* GTG36) {
    if (is_f64)
        return InformationMatrix<double>((const double*)source, ns,
                                         (const double*)target, nt, max_dist,
                                         transformation, 1, GTG36);
    return InformationMatrix<float>((const float*)source, ns,
                                    (const float*)target, nt, max_dist,
                                    transformation, accumulate_double, GTG36);
}

// EvaluateRegistration, Registration.cpp:64-91.
void orc_evaluate_registration(const void* source, int64_t ns,
                               const void* target, int64_t nt, int is_f64,
                               double max_dist, const double* transformation,
                               double* out_T, double* out_fitness,
                               double* out_rmse, int64_t* out_corr) {
    RegResult r;
    if (is_f64) {
        std::vector<double> s((const double*)source,
                              (const double*)source + 3 * ns);
        TransformPoints<double>(transformation, s.data(), ns);
        r = ComputeRegistrationResult<double>(s.data(), ns,
                                              (const double*)target, nt,
                                              max_dist, transformation);
    } else {
        std::vector<float> s((const float*)source,
                             (const float*)source + 3 * ns);
        TransformPoints<float>(transformation, s.data(), ns);
        r = ComputeRegistrationResult<float>(s.data(), ns, (const float*)target,
                                             nt, max_dist, transformation);
    }
    std::memcpy(out_T, r.T, sizeof(r.T));
    *out_fitness = r.fitness;
    *out_rmse = r.inlier_rmse;
    if (out_corr)
        for (int64_t i = 0; i < ns; ++i) out_corr[i] = r.correspondences[(size_t)i];
}

// 29 sums of ComputePoseSymmetricKernelCPU; means given in float64 and rounded
// to the point dtype (as the Tensor the reference passes).
void orc_symmetric_accumulate(const void* src, const void* tgt, const void* sn,
                              const void* tn, const int64_t* corr, int64_t n,
                              int is_f64, const double* source_mean3,
                              const double* target_mean3, int method,
                              double scaling, double shape,
                              int accumulate_double, double* out29) {
    if (is_f64) {
        ComputePoseSymmetricKernel<double, double>(
                (const double*)src, (const double*)tgt, (const double*)sn,
                (const double*)tn, corr, source_mean3, target_mean3, n, out29,
                method, scaling, shape);
        return;
    }
    float ms[3], mt[3];
    for (int k = 0; k < 3; ++k) {
        ms[k] = (float)source_mean3[k];
        mt[k] = (float)target_mean3[k];
    }
    if (accumulate_double) {
        ComputePoseSymmetricKernel<float, double>(
                (const float*)src, (const float*)tgt, (const float*)sn,
                (const float*)tn, corr, ms, mt, n, out29, method, scaling,
                shape);
    } else {
        float A[29];
        ComputePoseSymmetricKernel<float, float>(
                (const float*)src, (const float*)tgt, (const float*)sn,
                (const float*)tn, corr, ms, mt, n, A, method, scaling, shape);
        for (int i = 0; i < 29; ++i) out29[i] = (double)A[i];
    }
}

void orc_colored_accumulate(const void* src, const void* src_c, const void* tgt,
                            const void* tn, const void* tc, const void* tg,
                            const int64_t* corr, int64_t n, int is_f64,
                            double lambda_geometric, int method, double scaling,
                            double shape, int accumulate_double,
                            double* out29) {
    if (is_f64) {
        ComputePoseColoredICPKernel<double, double>(
                (const double*)src, (const double*)src_c, (const double*)tgt,
                (const double*)tn, (const double*)tc, (const double*)tg, corr,
                lambda_geometric, n, out29, method, scaling, shape);
    } else if (accumulate_double) {
        ComputePoseColoredICPKernel<float, double>(
                (const float*)src, (const float*)src_c, (const float*)tgt,
                (const float*)tn, (const float*)tc, (const float*)tg, corr,
                lambda_geometric, n, out29, method, scaling, shape);
    } else {
        float A[29];
        ComputePoseColoredICPKernel<float, float>(
                (const float*)src, (const float*)src_c, (const float*)tgt,
                (const float*)tn, (const float*)tc, (const float*)tg, corr,
                lambda_geometric, n, A, method, scaling, shape);
        for (int i = 0; i < 29; ++i) out29[i] = (double)A[i];
    }
}

void orc_symmetric_pose_to_transformation(const double* pose6,
                                          const double* source_mean3,
                                          const double* target_mean3,
                                          double* T16) {
    TransformSymmetricPoseToMatrix4d(pose6, source_mean3, target_mean3, T16);
}

int orc_compute_transformation_symmetric(
        const void* src, const void* tgt, const void* sn, const void* tn,
        const int64_t* corr, int64_t n, int is_f64, int method, double scaling,
        double shape, int accumulate_double, double* T16, double* sums29) {
    if (is_f64)
        return ComputeTransformationSymmetric<double>(
                (const double*)src, (const double*)tgt, (const double*)sn,
                (const double*)tn, corr, n, method, scaling, shape, 1, T16,
                sums29);
    return ComputeTransformationSymmetric<float>(
            (const float*)src, (const float*)tgt, (const float*)sn,
            (const float*)tn, corr, n, method, scaling, shape,
            accumulate_double, T16, sums29);
}

// R9 (row-major), t3 as float64; returns the number of correspondences.
int64_t orc_compute_rt_p2point(const void* src, const void* tgt,
                               const int64_t* corr, int64_t n, int is_f64,
                               int accumulate_double, double* R9, double* t3) {
    if (is_f64)
        return ComputeRtPointToPoint<double>((const double*)src,
                                             (const double*)tgt, corr, n, 1, R9,
                                             t3);
    return ComputeRtPointToPoint<float>((const float*)src, (const float*)tgt,
                                        corr, n, accumulate_double, R9, t3);
}

// Sxy {3,3}, means {3}: in the point dtype widened to double (what the
// reference's Get3x3SxyLinearSystem returns), or accumulated in double.
int64_t orc_p2point_sxy(const void* src, const void* tgt, const int64_t* corr,
                        int64_t n, int is_f64, int accumulate_double,
                        double* Sxy9, double* source_mean3,
                        double* target_mean3) {
    if (is_f64)
        return Get3x3Sxy<double, double>((const double*)src, (const double*)tgt,
                                         corr, n, Sxy9, source_mean3,
                                         target_mean3);
    if (accumulate_double)
        return Get3x3Sxy<float, double>((const float*)src, (const float*)tgt,
                                        corr, n, Sxy9, source_mean3,
                                        target_mean3);
    float S[9], ms[3], mt[3];
    int64_t c = Get3x3Sxy<float, float>((const float*)src, (const float*)tgt,
                                        corr, n, S, ms, mt);
    for (int i = 0; i < 9; ++i) Sxy9[i] = S[i];
    for (int i = 0; i < 3; ++i) {
        source_mean3[i] = ms[i];
        target_mean3[i] = mt[i];
    }
    return c;
}

void orc_rt_from_sxy(const double* Sxy9, const double* source_mean3,
                     const double* target_mean3, int as_f32, double* R9,
                     double* t3) {
    if (as_f32) {
        float S[9], ms[3], mt[3];
        for (int i = 0; i < 9; ++i) S[i] = (float)Sxy9[i];
        for (int i = 0; i < 3; ++i) {
            ms[i] = (float)source_mean3[i];
            mt[i] = (float)target_mean3[i];
        }
        RtFromSxy<float>(S, ms, mt, R9, t3);
    } else {
        RtFromSxy<double>(Sxy9, source_mean3, target_mean3, R9, t3);
    }
}

int orc_multiscale_icp_ex(const void* source, const void* source_normals,
                          const void* source_colors, const void* target_colors,
                          const void* target_color_gradients,
                          double lambda_geometric, int64_t ns,
                          const void* target, const void* target_normals,
                          int64_t nt, int is_f64,
                          int num_scales, const double* voxel_sizes,
                          const int* max_iterations, const double* rel_fitness,
                          const double* rel_rmse, const double* max_dists,
                          const double* init, int estimation, int kernel_method,
                          double kernel_scale, double kernel_shape,
                          int accumulate_double, double* out_T,
                          double* out_fitness, double* out_rmse,
                          int* out_converged, int* out_num_iterations,
                          int64_t* out_correspondences, int64_t* out_num_corr,
                          icp_callback_t cb, void* user) {
    IcpExtra extra;
    extra.source_normals = source_normals;
    extra.source_colors = source_colors;
    extra.target_colors = target_colors;
    extra.target_color_gradients = target_color_gradients;
    extra.lambda_geometric = lambda_geometric;
    if (is_f64)
        return MultiScaleICP<double>(
                (const double*)source, extra, ns, (const double*)target, (const double*)target_normals, nt,
                num_scales, voxel_sizes, max_iterations, rel_fitness, rel_rmse,
                max_dists, init, kernel_method, kernel_scale, kernel_shape,
                accumulate_double, estimation, out_T, out_fitness, out_rmse, out_converged,
                out_num_iterations, out_correspondences, out_num_corr, cb,
                user);
    return MultiScaleICP<float>(
            (const float*)source, extra, ns, (const float*)target, (const float*)target_normals, nt, num_scales,
            voxel_sizes, max_iterations, rel_fitness, rel_rmse, max_dists, init,
            kernel_method, kernel_scale, kernel_shape, accumulate_double,
            estimation, out_T, out_fitness, out_rmse, out_converged,
            out_num_iterations, out_correspondences, out_num_corr, cb, user);
}

int orc_multiscale_icp(const void* source, int64_t ns, const void* target,
                       const void* target_normals, int64_t nt, int is_f64,
                       int num_scales, const double* voxel_sizes,
                       const int* max_iterations, const double* rel_fitness,
                       const double* rel_rmse, const double* max_dists,
                       const double* init, int kernel_method,
                       double kernel_scale, double kernel_shape,
                       int accumulate_double, double* out_T,
                       double* out_fitness, double* out_rmse,
                       int* out_converged, int* out_num_iterations,
                       int64_t* out_correspondences, int64_t* out_num_corr,
                       icp_callback_t cb, void* user) {
    if (is_f64)
        return MultiScaleICP<double>(
                (const double*)source, IcpExtra(), ns, (const double*)target,
                (const double*)target_normals, nt, num_scales, voxel_sizes,
                max_iterations, rel_fitness, rel_rmse, max_dists, init,
                kernel_method, kernel_scale, kernel_shape, accumulate_double,
                0, out_T, out_fitness, out_rmse, out_converged,
                out_num_iterations, out_correspondences, out_num_corr, cb,
                user);
    return MultiScaleICP<float>(
            (const float*)source, IcpExtra(), ns, (const float*)target,
            (const float*)target_normals, nt, num_scales, voxel_sizes,
            max_iterations, rel_fitness, rel_rmse, max_dists, init,
            kernel_method, kernel_scale, kernel_shape, accumulate_double, 0,
            out_T, out_fitness, out_rmse, out_converged, out_num_iterations,
            out_correspondences, out_num_corr, cb, user);
}

}  // extern "C"
