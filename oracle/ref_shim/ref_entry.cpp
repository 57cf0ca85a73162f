// TEST INFRASTRUCTURE ONLY.
//
// C entry points over the reference's OWN hot-path kernel files, compiled
// from where they lie under /root/reference/cpp/open3d (see README.md in this
// directory and `make -C oracle ref`). Nothing below re-implements reference
// arithmetic: each function wraps raw pointers into the stand-in Tensor and
// calls the reference function named in its comment.
//
// The signatures deliberately mirror the oracle's orc_* functions
// (oracle/vbg_oracle.cpp, oracle/icp_oracle.cpp) so tests can run the same
// inputs through both and compare bit for bit.

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

// The reference translation units, unmodified.
#include "open3d/t/geometry/kernel/PointCloudCPU.cpp"
#include "open3d/t/geometry/kernel/TransformImpl.h"
#include "open3d/t/geometry/kernel/VoxelBlockGridCPU.cpp"
#include "open3d/t/pipelines/kernel/RegistrationCPU.cpp"
#include "open3d/t/pipelines/kernel/TransformationConverter.cpp"

// PointCloudCPU.cpp also holds the search-based normal / colour-gradient
// estimators, which call core::nns::NearestNeighborSearch (nanoflann, not
// available here). They are off the paths exercised through this file; the
// definitions below only satisfy the linker.
namespace open3d {
namespace core {
namespace nns {
#define REF_NNS_UNAVAILABLE utility::LogError("shim: nanoflann search is not available")
NearestNeighborSearch::~NearestNeighborSearch() {}
bool NearestNeighborSearch::KnnIndex() { REF_NNS_UNAVAILABLE; }
bool NearestNeighborSearch::FixedRadiusIndex(std::optional<double>) { REF_NNS_UNAVAILABLE; }
bool NearestNeighborSearch::HybridIndex(std::optional<double>) { REF_NNS_UNAVAILABLE; }
std::pair<Tensor, Tensor> NearestNeighborSearch::KnnSearch(const Tensor&, int) { REF_NNS_UNAVAILABLE; }
std::tuple<Tensor, Tensor, Tensor> NearestNeighborSearch::FixedRadiusSearch(const Tensor&, double, bool) { REF_NNS_UNAVAILABLE; }
std::tuple<Tensor, Tensor, Tensor> NearestNeighborSearch::HybridSearch(const Tensor&, double, int) const { REF_NNS_UNAVAILABLE; }
#undef REF_NNS_UNAVAILABLE
}  // namespace nns
}  // namespace core
}  // namespace open3d

using namespace open3d;
using core::Tensor;
namespace vg = open3d::t::geometry::kernel::voxel_grid;
namespace pk = open3d::t::pipelines::kernel;
namespace reg = open3d::t::pipelines::registration;

namespace {

thread_local std::string g_err;

Tensor Wrap(const void* p, core::SizeVector shape, core::Dtype dt) {
    return Tensor::FromPtr(p, shape, dt);
}
Tensor Mat(const double* p, int r, int c) {
    // K / T are Float64 host tensors (t/geometry/Utility.h checks).
    return Tensor::FromPtr(p, {r, c}, core::Float64).Clone();
}

using Key = utility::MiniVec<int, 3>;
using Hash = utility::MiniVecHash<int, 3>;
using Eq = utility::MiniVecEq<int, 3>;

std::shared_ptr<core::HashMap> MakeHashMap(const int* keys,
                                           const int* buf_indices, int64_t n) {
    auto backend = std::make_shared<core::TBBHashBackend<Key, Hash, Eq>>();
    auto impl = backend->GetImpl();
    for (int64_t i = 0; i < n; ++i) {
        Key k;
        k[0] = keys[3 * i + 0];
        k[1] = keys[3 * i + 1];
        k[2] = keys[3 * i + 2];
        (*impl)[k] = (core::buf_index_t)buf_indices[i];
    }
    return std::make_shared<core::HashMap>(backend);
}

template <typename F>
int Guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}

}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

void ref_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

// Schedule of the stand-in tbb::parallel_reduce (tbb/parallel_reduce.h):
// seed 0 = one sequential chunk (default); otherwise seeded random splits down
// to chunks of <= min_chunk elements.
void ref_set_reduce_schedule(unsigned long long seed, long long min_chunk) {
    tbb::shim::SetSchedule(seed, min_chunk);
}

// DepthTouchCPU, t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201.
// Returns the number of unique blocks, -1 on error ("No block is touched").
int64_t ref_depth_touch(const void* depth, int depth_is_f32, int rows, int cols,
                        const double* intrinsic, const double* extrinsic,
                        int resolution, float voxel_size, float sdf_trunc,
                        float depth_scale, float depth_max, int stride,
                        int* out_coords, int64_t out_capacity) {
    int64_t m = -1;
    Guard([&] {
        Tensor d = Wrap(depth, {rows, cols, 1},
                        depth_is_f32 ? core::Float32 : core::UInt16);
        std::shared_ptr<core::HashMap> unused;
        Tensor coords;
        vg::DepthTouchCPU(unused, d, Mat(intrinsic, 3, 3), Mat(extrinsic, 4, 4),
                          coords, resolution, voxel_size, sdf_trunc,
                          depth_scale, depth_max, stride);
        m = coords.GetLength();
        if (m > out_capacity) m = out_capacity;
        std::memcpy(out_coords, coords.GetDataPtr<int>(),
                    sizeof(int) * 3 * (size_t)m);
    });
    return m;
}

// PointCloudTouchCPU, VoxelBlockGridCPU.cpp:55-115.
int64_t ref_pointcloud_touch(const float* pcd, int64_t n, int resolution,
                             float voxel_size, float sdf_trunc, int* out_coords,
                             int64_t out_capacity) {
    int64_t m = -1;
    Guard([&] {
        Tensor p = Wrap(pcd, {n, 3}, core::Float32);
        std::shared_ptr<core::HashMap> unused;
        Tensor coords;
        vg::PointCloudTouchCPU(unused, p, coords, resolution, voxel_size,
                               sdf_trunc);
        m = coords.GetLength();
        if (m > out_capacity) m = out_capacity;
        std::memcpy(out_coords, coords.GetDataPtr<int>(),
                    sizeof(int) * 3 * (size_t)m);
    });
    return m;
}

// GetVoxelCoordinatesAndFlattenedIndicesCPU, VoxelBlockGridImpl.h:43-92.
int ref_voxel_coords_flat(const int* buf_indices, int64_t n_blocks,
                          const int* block_keys, int64_t key_capacity,
                          int resolution, float voxel_size,
                          float* voxel_coords, int64_t* flattened) {
    return Guard([&] {
        const int64_t n = n_blocks * resolution * resolution * resolution;
        Tensor bi = Wrap(buf_indices, {n_blocks}, core::Int32);
        Tensor keys = Wrap(block_keys, {key_capacity, 3}, core::Int32);
        Tensor vc = Wrap(voxel_coords, {n, 3}, core::Float32);
        Tensor fl = Wrap(flattened, {n}, core::Int64);
        vg::GetVoxelCoordinatesAndFlattenedIndicesCPU(bi, keys, vc, fl,
                                                      resolution, voxel_size);
    });
}

// IntegrateCPU<...>, t/geometry/kernel/VoxelBlockGridImpl.h:151-308, with the
// dtype dispatch of t/geometry/kernel/VoxelBlockGrid.cpp:107-146.
int ref_integrate(const void* depth, int depth_rows, int depth_cols,
                  const void* color, int color_rows, int color_cols,
                  int input_is_f32, const int* indices, int64_t n_indices,
                  const int* block_keys, int64_t capacity, float* tsdf,
                  void* weight, void* color_buf, int grid_is_f32,
                  const double* depth_intrinsic, const double* color_intrinsic,
                  const double* extrinsics, int resolution, float voxel_size,
                  float sdf_trunc, float depth_scale, float depth_max) {
    return Guard([&] {
        const core::Dtype ddt = input_is_f32 ? core::Float32 : core::UInt16;
        const core::Dtype cdt = input_is_f32 ? core::Float32 : core::UInt8;
        const core::Dtype gdt = grid_is_f32 ? core::Float32 : core::UInt16;
        Tensor d = Wrap(depth, {depth_rows, depth_cols, 1}, ddt);
        Tensor c = color ? Wrap(color, {color_rows, color_cols, 3}, cdt)
                         : Tensor({0}, cdt);
        Tensor idx = Wrap(indices, {n_indices}, core::Int32);
        Tensor keys = Wrap(block_keys, {capacity, 3}, core::Int32);
        const int64_t r = resolution;
        t::geometry::TensorMap vm("tsdf");
        vm["tsdf"] = Wrap(tsdf, {capacity, r, r, r, 1}, core::Float32);
        vm["weight"] = Wrap(weight, {capacity, r, r, r, 1}, gdt);
        if (color_buf) vm["color"] = Wrap(color_buf, {capacity, r, r, r, 3}, gdt);
        Tensor Kd = Mat(depth_intrinsic, 3, 3);
        Tensor Kc = Mat(color_intrinsic ? color_intrinsic : depth_intrinsic, 3, 3);
        Tensor T = Mat(extrinsics, 4, 4);
#define REF_CALL(DT, CT, WT, VT)                                              \
    vg::IntegrateCPU<DT, CT, float, WT, VT>(d, c, idx, keys, vm, Kd, Kc, T,   \
                                            resolution, voxel_size,           \
                                            sdf_trunc, depth_scale, depth_max)
        if (!input_is_f32 && !grid_is_f32)
            REF_CALL(uint16_t, uint8_t, uint16_t, uint16_t);
        else if (!input_is_f32 && grid_is_f32)
            REF_CALL(uint16_t, uint8_t, float, float);
        else if (input_is_f32 && !grid_is_f32)
            REF_CALL(float, float, uint16_t, uint16_t);
        else
            REF_CALL(float, float, float, float);
#undef REF_CALL
    });
}

// EstimateRangeCPU, VoxelBlockGridImpl.h:310-555. frag_buffer_size <= 0 passes
// an empty fragment buffer, which is what VoxelBlockGrid::RayCast does on its
// first call (member fragment_buffer_, t/geometry/VoxelBlockGrid.cpp:352): the
// kernel then sizes it by its own heuristic (Impl.h:342-349) and may drop
// fragments. > 0 allocates that many fragments up front.
int ref_estimate_range(const int* block_keys, int64_t n_blocks,
                       float* range_minmax_map, const double* intrinsics,
                       const double* extrinsics, int h, int w, int down_factor,
                       int64_t block_resolution, float voxel_size,
                       float depth_min, float depth_max, int frag_buffer_size) {
    return Guard([&] {
        Tensor keys = Wrap(block_keys, {n_blocks, 3}, core::Int32);
        Tensor range;
        Tensor frag;
        if (frag_buffer_size > 0)
            frag = Tensor({(int64_t)frag_buffer_size, 6}, core::Float32);
        vg::EstimateRangeCPU(keys, range, Mat(intrinsics, 3, 3),
                             Mat(extrinsics, 4, 4), h, w, down_factor,
                             block_resolution, voxel_size, depth_min, depth_max,
                             frag);
        std::memcpy(range_minmax_map, range.GetDataPtr<float>(),
                    sizeof(float) * (size_t)range.NumElements());
    });
}

// RayCastCPU<...>, VoxelBlockGridImpl.h:578-1120. The hash map is given as
// (key, buffer index) pairs.
int ref_raycast(const int* hash_keys, const int* hash_buf_indices,
                int64_t n_hash, int64_t capacity, const float* tsdf,
                const void* weight, const void* color_buf, int grid_is_f32,
                const float* range_map, float* out_depth, float* out_vertex,
                float* out_color, float* out_normal, int64_t* out_index,
                uint8_t* out_mask, float* out_ratio, float* out_ratio_dx,
                float* out_ratio_dy, float* out_ratio_dz,
                const double* intrinsic, const double* extrinsics, int h, int w,
                int block_resolution, float voxel_size, float depth_scale,
                float depth_min, float depth_max, float weight_threshold,
                float trunc_voxel_multiplier, int range_map_down_factor) {
    return Guard([&] {
        auto hm = MakeHashMap(hash_keys, hash_buf_indices, n_hash);
        const core::Dtype gdt = grid_is_f32 ? core::Float32 : core::UInt16;
        const int64_t r = block_resolution;
        t::geometry::TensorMap vm("tsdf");
        vm["tsdf"] = Wrap(tsdf, {capacity, r, r, r, 1}, core::Float32);
        vm["weight"] = Wrap(weight, {capacity, r, r, r, 1}, gdt);
        if (color_buf) vm["color"] = Wrap(color_buf, {capacity, r, r, r, 3}, gdt);
        Tensor range = Wrap(range_map,
                            {h / range_map_down_factor,
                             w / range_map_down_factor, 2},
                            core::Float32);
        t::geometry::TensorMap rm("depth");
        if (out_depth) rm["depth"] = Wrap(out_depth, {h, w, 1}, core::Float32);
        if (out_vertex) rm["vertex"] = Wrap(out_vertex, {h, w, 3}, core::Float32);
        if (out_color) rm["color"] = Wrap(out_color, {h, w, 3}, core::Float32);
        if (out_normal) rm["normal"] = Wrap(out_normal, {h, w, 3}, core::Float32);
        if (out_index) rm["index"] = Wrap(out_index, {h, w, 8}, core::Int64);
        if (out_mask) rm["mask"] = Wrap(out_mask, {h, w, 8}, core::Bool);
        if (out_ratio)
            rm["interp_ratio"] = Wrap(out_ratio, {h, w, 8}, core::Float32);
        if (out_ratio_dx)
            rm["interp_ratio_dx"] = Wrap(out_ratio_dx, {h, w, 8}, core::Float32);
        if (out_ratio_dy)
            rm["interp_ratio_dy"] = Wrap(out_ratio_dy, {h, w, 8}, core::Float32);
        if (out_ratio_dz)
            rm["interp_ratio_dz"] = Wrap(out_ratio_dz, {h, w, 8}, core::Float32);
        Tensor K = Mat(intrinsic, 3, 3), T = Mat(extrinsics, 4, 4);
        if (grid_is_f32)
            vg::RayCastCPU<float, float, float>(
                    hm, vm, range, rm, K, T, h, w, block_resolution, voxel_size,
                    depth_scale, depth_min, depth_max, weight_threshold,
                    trunc_voxel_multiplier, range_map_down_factor);
        else
            vg::RayCastCPU<float, uint16_t, uint16_t>(
                    hm, vm, range, rm, K, T, h, w, block_resolution, voxel_size,
                    depth_scale, depth_min, depth_max, weight_threshold,
                    trunc_voxel_multiplier, range_map_down_factor);
    });
}

// ExtractPointCloudCPU<...>, VoxelBlockGridImpl.h:1122-1365 (instantiations
// VoxelBlockGridCPU.cpp:234-245). nb tables are {27, n, 1} as
// BufferRadiusNeighbors builds them. valid_size < 0 runs the counting pass.
// Returns the number of points (total count), -1 on error. Output order is
// the atomic counter's (sequential with ref_set_threads(1)).
int64_t ref_extract_point_cloud(const int* indices, const int* nb_indices,
                                const uint8_t* nb_masks, const int* block_keys,
                                int64_t capacity, const float* tsdf,
                                const void* weight, const void* color_buf,
                                int grid_is_f32, int64_t n_blocks,
                                int resolution, float voxel_size,
                                float weight_threshold, float* points,
                                float* normals, float* colors,
                                int64_t out_capacity, int valid_size_in) {
    int valid_size = valid_size_in;
    int rc = Guard([&] {
        const core::Dtype gdt = grid_is_f32 ? core::Float32 : core::UInt16;
        const int64_t r = resolution;
        Tensor idx = Wrap(indices, {n_blocks}, core::Int32);
        Tensor nbi = Wrap(nb_indices, {27, n_blocks, 1}, core::Int32);
        Tensor nbm = Wrap(nb_masks, {27, n_blocks, 1}, core::Bool);
        Tensor keys = Wrap(block_keys, {capacity, 3}, core::Int32);
        t::geometry::TensorMap vm("tsdf");
        vm["tsdf"] = Wrap(tsdf, {capacity, r, r, r, 1}, core::Float32);
        vm["weight"] = Wrap(weight, {capacity, r, r, r, 1}, gdt);
        if (color_buf) vm["color"] = Wrap(color_buf, {capacity, r, r, r, 3}, gdt);
        Tensor p, n, c;
        if (grid_is_f32)
            vg::ExtractPointCloudCPU<float, float, float>(
                    idx, nbi, nbm, keys, vm, p, n, c, resolution, voxel_size,
                    weight_threshold, valid_size);
        else
            vg::ExtractPointCloudCPU<float, uint16_t, uint16_t>(
                    idx, nbi, nbm, keys, vm, p, n, c, resolution, voxel_size,
                    weight_threshold, valid_size);
        int64_t m = std::min<int64_t>(std::min<int64_t>(valid_size, p.GetLength()),
                                      out_capacity);
        if (points && m > 0)
            std::memcpy(points, p.GetDataPtr<float>(), sizeof(float) * 3 * m);
        if (normals && m > 0)
            std::memcpy(normals, n.GetDataPtr<float>(), sizeof(float) * 3 * m);
        if (colors && color_buf && m > 0)
            std::memcpy(colors, c.GetDataPtr<float>(), sizeof(float) * 3 * m);
    });
    return rc == 0 ? (int64_t)valid_size : -1;
}

// EstimatePointWiseRobustNormalizedCovarianceKernel (PointCloudImpl.h:512-585)
// per point over given hybrid-search results, and
// EstimateNormalsFromCovariancesCPU (:1011-1063). The search itself
// (nanoflann) is not part of this build.
int ref_estimate_covariances(const void* points, const int32_t* indices,
                             const int32_t* counts, int64_t n, int max_nn,
                             int is_f64, void* covariances) {
    return Guard([&] {
        namespace pc = open3d::t::geometry::kernel::pointcloud;
        for (int64_t w = 0; w < n; ++w) {
            const int32_t cnt = counts[w];
            if (is_f64)
                pc::EstimatePointWiseRobustNormalizedCovarianceKernel<double>(
                        (const double*)points, indices + (int64_t)max_nn * w, cnt,
                        (double*)covariances + 9 * w);
            else
                pc::EstimatePointWiseRobustNormalizedCovarianceKernel<float>(
                        (const float*)points, indices + (int64_t)max_nn * w, cnt,
                        (float*)covariances + 9 * w);
        }
    });
}

// EstimatePointWiseColorGradientKernel (PointCloudImpl.h:1067-1165, with the
// reference's own solve_svd3x3) per point over given neighbour lists.
int ref_estimate_color_gradients(const void* points, const void* normals,
                                 const void* colors, const int32_t* indices,
                                 const int32_t* counts, int64_t n, int max_nn,
                                 int is_f64, void* gradients) {
    return Guard([&] {
        namespace pc = open3d::t::geometry::kernel::pointcloud;
        for (int64_t w = 0; w < n; ++w) {
            const int32_t idx_offset = (int32_t)(3 * w);
            const int32_t cnt = counts[w];
            if (is_f64)
                pc::EstimatePointWiseColorGradientKernel<double>(
                        (const double*)points, (const double*)normals,
                        (const double*)colors, idx_offset,
                        indices + (int64_t)max_nn * w, cnt, (double*)gradients);
            else
                pc::EstimatePointWiseColorGradientKernel<float>(
                        (const float*)points, (const float*)normals,
                        (const float*)colors, idx_offset,
                        indices + (int64_t)max_nn * w, cnt, (float*)gradients);
        }
    });
}

// core::linalg::kernel::svd3x3 / solve_svd3x3 (core/linalg/kernel/SVD3x3.h),
// row-major 3x3.
int ref_svd3x3(const void* A, int is_f64, void* U, void* S, void* V) {
    return Guard([&] {
        namespace lk = open3d::core::linalg::kernel;
        if (is_f64)
            lk::svd3x3<double>((const double*)A, (double*)U, (double*)S,
                               (double*)V);
        else
            lk::svd3x3<float>((const float*)A, (float*)U, (float*)S,
                              (float*)V);
    });
}
int ref_solve_svd3x3(const void* A, const void* b, int is_f64, void* x) {
    return Guard([&] {
        namespace lk = open3d::core::linalg::kernel;
        if (is_f64)
            lk::solve_svd3x3<double>((const double*)A, (const double*)b,
                                     (double*)x);
        else
            lk::solve_svd3x3<float>((const float*)A, (const float*)b,
                                    (float*)x);
    });
}

int ref_normals_from_covariances(const void* covariances, int64_t n, int is_f64,
                                 void* normals_io, int has_normals) {
    return Guard([&] {
        namespace pc = open3d::t::geometry::kernel::pointcloud;
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor cov = Wrap(covariances, {n, 3, 3}, dt);
        Tensor nrm = Wrap(normals_io, {n, 3}, dt);
        pc::EstimateNormalsFromCovariancesCPU(cov, nrm, has_normals != 0);
    });
}

// UnprojectCPU, t/geometry/kernel/PointCloudImpl.h:42-143.
int64_t ref_unproject(const void* depth, int depth_is_f32, int rows, int cols,
                      const float* colors_f32, float* points, float* colors,
                      const double* intrinsics, const double* extrinsics,
                      float depth_scale, float depth_max, int64_t stride) {
    int64_t m = -1;
    Guard([&] {
        Tensor d = Wrap(depth, {rows, cols, 1},
                        depth_is_f32 ? core::Float32 : core::UInt16);
        Tensor pts, cols_out;
        Tensor img;
        std::optional<std::reference_wrapper<const Tensor>> in_c = std::nullopt;
        std::optional<std::reference_wrapper<Tensor>> out_c = std::nullopt;
        if (colors_f32 && colors) {
            img = Wrap(colors_f32, {rows, cols, 3}, core::Float32);
            in_c = std::cref(img);
            out_c = std::ref(cols_out);
        }
        t::geometry::kernel::pointcloud::UnprojectCPU(
                d, in_c, pts, out_c, Mat(intrinsics, 3, 3),
                Mat(extrinsics, 4, 4), depth_scale, depth_max, stride);
        m = pts.GetLength();
        std::memcpy(points, pts.GetDataPtr<float>(),
                    sizeof(float) * 3 * (size_t)m);
        if (colors_f32 && colors)
            std::memcpy(colors, cols_out.GetDataPtr<float>(),
                        sizeof(float) * 3 * (size_t)m);
    });
    return m;
}

// ComputePosePointToPlaneKernelCPU, t/pipelines/kernel/RegistrationCPU.cpp:
// 30-90: the 29 sums in the point dtype (widened to double for the caller).
// The stand-in tbb::parallel_reduce runs it as one sequential chunk.
int ref_p2plane_accumulate(const void* src, const void* tgt, const void* tgt_n,
                           const int64_t* corr, int64_t n, int is_f64,
                           int method, double scaling, double shape,
                           double* sums29) {
    return Guard([&] {
        reg::RobustKernel kernel((reg::RobustKernelMethod)method, scaling,
                                 shape);
        using reg::RobustKernelMethod;
        if (is_f64) {
            std::vector<double> g(29, 0.0);
            using scalar_t = double;
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePosePointToPlaneKernelCPU(
                                (const double*)src, (const double*)tgt,
                                (const double*)tgt_n, corr, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        } else {
            std::vector<float> g(29, 0.0f);
            using scalar_t = float;
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePosePointToPlaneKernelCPU(
                                (const float*)src, (const float*)tgt,
                                (const float*)tgt_n, corr, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        }
    });
}

// ComputePoseSymmetricKernelCPU, RegistrationCPU.cpp:124-180 (with
// GetJacobianSymmetric, RegistrationImpl.h:323-386): the 29 sums in the point
// dtype, widened to double. Means are given in float64 and rounded to the
// dtype, as the {3} Tensor the reference passes.
int ref_symmetric_accumulate(const void* src, const void* tgt, const void* sn,
                             const void* tn, const int64_t* corr, int64_t n,
                             int is_f64, const double* source_mean3,
                             const double* target_mean3, int method,
                             double scaling, double shape, double* sums29) {
    return Guard([&] {
        reg::RobustKernel kernel((reg::RobustKernelMethod)method, scaling,
                                 shape);
        using reg::RobustKernelMethod;
        if (is_f64) {
            std::vector<double> g(29, 0.0);
            using scalar_t = double;
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePoseSymmetricKernelCPU(
                                (const double*)src, (const double*)tgt,
                                (const double*)sn, (const double*)tn, corr,
                                source_mean3, target_mean3, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        } else {
            float ms[3], mt[3];
            for (int k = 0; k < 3; ++k) {
                ms[k] = (float)source_mean3[k];
                mt[k] = (float)target_mean3[k];
            }
            std::vector<float> g(29, 0.0f);
            using scalar_t = float;
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePoseSymmetricKernelCPU(
                                (const float*)src, (const float*)tgt,
                                (const float*)sn, (const float*)tn, corr, ms,
                                mt, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        }
    });
}

// ComputePoseColoredICPKernelCPU, RegistrationCPU.cpp:220-290 (with
// GetJacobianColoredICP, RegistrationImpl.h:388-466): 29 sums in the point
// dtype, widened to double; sqrt(lambda) as ComputePoseColoredICPCPU forms it.
int ref_colored_accumulate(const void* src, const void* src_c, const void* tgt,
                           const void* tn, const void* tc, const void* tg,
                           const int64_t* corr, int64_t n, int is_f64,
                           double lambda_geometric, int method, double scaling,
                           double shape, double* sums29) {
    return Guard([&] {
        reg::RobustKernel kernel((reg::RobustKernelMethod)method, scaling,
                                 shape);
        using reg::RobustKernelMethod;
        if (is_f64) {
            std::vector<double> g(29, 0.0);
            using scalar_t = double;
            scalar_t slg = static_cast<scalar_t>(sqrt(lambda_geometric));
            scalar_t slp = static_cast<scalar_t>(sqrt(1.0 - lambda_geometric));
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePoseColoredICPKernelCPU(
                                (const double*)src, (const double*)src_c,
                                (const double*)tgt, (const double*)tn,
                                (const double*)tc, (const double*)tg, corr, slg,
                                slp, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        } else {
            std::vector<float> g(29, 0.0f);
            using scalar_t = float;
            scalar_t slg = static_cast<scalar_t>(sqrt(lambda_geometric));
            scalar_t slp = static_cast<scalar_t>(sqrt(1.0 - lambda_geometric));
            DISPATCH_ROBUST_KERNEL_FUNCTION(
                    kernel.type_, scalar_t, kernel.scaling_parameter_,
                    kernel.shape_parameter_, [&]() {
                        pk::ComputePoseColoredICPKernelCPU(
                                (const float*)src, (const float*)src_c,
                                (const float*)tgt, (const float*)tn,
                                (const float*)tc, (const float*)tg, corr, slg,
                                slp, (int)n, g.data(),
                                GetWeightFromRobustKernel);
                    });
            for (int i = 0; i < 29; ++i) sums29[i] = g[(size_t)i];
        }
    });
}

// ComputeInformationMatrixCPU, RegistrationCPU.cpp:703-735: GTG {6,6} float64
// (the 21 sums are accumulated in the point dtype).
int ref_information_matrix(const void* tgt, const int64_t* corr, int64_t n,
                           int64_t n_tgt, int is_f64, double* GTG36) {
    return Guard([&] {
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor t = Wrap(tgt, {n_tgt, 3}, dt), c = Wrap(corr, {n}, core::Int64);
        Tensor info = Tensor::Empty({6, 6}, core::Float64);
        pk::ComputeInformationMatrixCPU(t, c, info, dt, core::Device("CPU:0"));
        std::memcpy(GTG36, info.GetDataPtr<double>(), sizeof(double) * 36);
    });
}

// Get3x3SxyLinearSystem, RegistrationCPU.cpp:495-617 (the reduction of
// ComputeRtPointToPointCPU): Sxy {3,3}, source_mean, target_mean in the point
// dtype (widened to double for the caller), inlier count.
int ref_p2point_sxy(const void* src, const void* tgt, const int64_t* corr,
                    int64_t n, int is_f64, double* Sxy9, double* source_mean3,
                    double* target_mean3, int* inlier_count) {
    return Guard([&] {
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor Sxy, target_mean, source_mean;
        if (is_f64)
            pk::Get3x3SxyLinearSystem<double>(
                    (const double*)src, (const double*)tgt, corr, (int)n, dt,
                    core::Device("CPU:0"), Sxy, target_mean, source_mean,
                    *inlier_count);
        else
            pk::Get3x3SxyLinearSystem<float>(
                    (const float*)src, (const float*)tgt, corr, (int)n, dt,
                    core::Device("CPU:0"), Sxy, target_mean, source_mean,
                    *inlier_count);
        for (int i = 0; i < 9; ++i)
            Sxy9[i] = is_f64 ? Sxy.GetDataPtr<double>()[i]
                             : (double)Sxy.GetDataPtr<float>()[i];
        for (int i = 0; i < 3; ++i) {
            source_mean3[i] = is_f64 ? source_mean.GetDataPtr<double>()[i]
                                     : (double)source_mean.GetDataPtr<float>()[i];
            target_mean3[i] = is_f64 ? target_mean.GetDataPtr<double>()[i]
                                     : (double)target_mean.GetDataPtr<float>()[i];
        }
    });
}

// ComputePosePointToPlaneCPU, RegistrationCPU.cpp:92-122 (reduction + decode +
// solve): pose {6} float64, residual, inlier count.
int ref_compute_pose_p2plane(const void* src, const void* tgt,
                             const void* tgt_n, const int64_t* corr, int64_t n,
                             int64_t n_tgt, int is_f64, int method,
                             double scaling, double shape, double* pose6,
                             float* residual, int* inlier_count) {
    return Guard([&] {
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor s = Wrap(src, {n, 3}, dt), t = Wrap(tgt, {n_tgt, 3}, dt),
               tn = Wrap(tgt_n, {n_tgt, 3}, dt),
               c = Wrap(corr, {n}, core::Int64);
        Tensor pose = Tensor::Empty({6}, core::Float64);
        reg::RobustKernel kernel((reg::RobustKernelMethod)method, scaling,
                                 shape);
        pk::ComputePosePointToPlaneCPU(s, t, tn, c, pose, *residual,
                                       *inlier_count, dt, core::Device("CPU:0"),
                                       kernel);
        std::memcpy(pose6, pose.GetDataPtr<double>(), sizeof(double) * 6);
    });
}

// DecodeAndSolve6x6, t/pipelines/kernel/TransformationConverter.cpp:189-226
// (the 6x6 solve itself is the stand-in Tensor::Solve: LU, partial pivoting).
int ref_decode_and_solve6x6(const double* A29, double* pose6, float* residual,
                            int* inlier_count) {
    return Guard([&] {
        Tensor A = Wrap(A29, {29}, core::Float64);
        Tensor delta = Tensor::Empty({6}, core::Float64);
        pk::DecodeAndSolve6x6(A, delta, *residual, *inlier_count);
        std::memcpy(pose6, delta.GetDataPtr<double>(), sizeof(double) * 6);
    });
}

// PoseToTransformation, TransformationConverter.cpp:81-104 (+Impl.h:23-42).
int ref_pose_to_transformation(const double* pose6, double* T16) {
    return Guard([&] {
        Tensor T = pk::PoseToTransformation(Wrap(pose6, {6}, core::Float64));
        std::memcpy(T16, T.GetDataPtr<double>(), sizeof(double) * 16);
    });
}

// TransformPointsCPU / TransformNormalsCPU, t/geometry/kernel/TransformImpl.h:
// 83-118 (in place; T is cast to the point dtype first as
// t/geometry/kernel/Transform.cpp:20-75 does).
int ref_transform_points(const double* T16, void* pts, int64_t n, int is_f64) {
    return Guard([&] {
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor p = Wrap(pts, {n, 3}, dt);
        Tensor T = Wrap(T16, {4, 4}, core::Float64).To(dt);
        t::geometry::kernel::transform::TransformPointsCPU(T, p);
    });
}
int ref_transform_normals(const double* T16, void* nrm, int64_t n, int is_f64) {
    return Guard([&] {
        const core::Dtype dt = is_f64 ? core::Float64 : core::Float32;
        Tensor p = Wrap(nrm, {n, 3}, dt);
        Tensor T = Wrap(T16, {4, 4}, core::Float64).To(dt);
        t::geometry::kernel::transform::TransformNormalsCPU(T, p);
    });
}

// DISPATCH_ROBUST_KERNEL_FUNCTION, registration/RobustKernelImpl.h:35-126.
double ref_robust_weight(int is_f64, int method, double scaling, double shape,
                         double residual) {
    using reg::RobustKernelMethod;
    reg::RobustKernelMethod m = (reg::RobustKernelMethod)method;
    double out = 0;
    if (is_f64) {
        using scalar_t = double;
        DISPATCH_ROBUST_KERNEL_FUNCTION(m, scalar_t, scaling, shape, [&]() {
            out = (double)GetWeightFromRobustKernel((scalar_t)residual);
        });
    } else {
        using scalar_t = float;
        DISPATCH_ROBUST_KERNEL_FUNCTION(m, scalar_t, scaling, shape, [&]() {
            out = (double)GetWeightFromRobustKernel((scalar_t)residual);
        });
    }
    return out;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// RGB-D odometry front end: the reference's image
// pyramid kernels and per-pixel Jacobian reductions, unmodified.
// ---------------------------------------------------------------------------
#include "open3d/t/geometry/kernel/ImageCPU.cpp"
#include "open3d/t/pipelines/kernel/RGBDOdometryCPU.cpp"

namespace img = open3d::t::geometry::kernel::image;
namespace odo = open3d::t::pipelines::kernel::odometry;

extern "C" {

// ClipTransformCPU, t/geometry/kernel/ImageImpl.h:94-128.
int ref_clip_transform(const void* src, int src_is_f32, int64_t rows,
                       int64_t cols, float scale, float min_value,
                       float max_value, float clip_fill, float* dst) {
    return Guard([&] {
        Tensor s = Wrap(src, {rows, cols, 1},
                        src_is_f32 ? core::Float32 : core::UInt16);
        Tensor d = Wrap(dst, {rows, cols, 1}, core::Float32);
        img::ClipTransformCPU(s, d, scale, min_value, max_value, clip_fill);
    });
}

// PyrDownDepthCPU, ImageImpl.h:132-206; dst is {rows/2, cols/2, 1}
// (t/geometry/Image.cpp:419-420).
int ref_pyrdown_depth(const float* src, int rows, int cols, float depth_diff,
                      float invalid_fill, float* dst) {
    return Guard([&] {
        Tensor s = Wrap(src, {rows, cols, 1}, core::Float32);
        Tensor d = Wrap(dst, {rows / 2, cols / 2, 1}, core::Float32);
        img::PyrDownDepthCPU(s, d, depth_diff, invalid_fill);
    });
}

// CreateVertexMapCPU, ImageImpl.h:208-256.
int ref_create_vertex_map(const float* src, int64_t rows, int64_t cols,
                          const double* K, float invalid_fill, float* dst) {
    return Guard([&] {
        Tensor s = Wrap(src, {rows, cols, 1}, core::Float32);
        Tensor d = Wrap(dst, {rows, cols, 3}, core::Float32);
        img::CreateVertexMapCPU(s, d, Mat(K, 3, 3), invalid_fill);
    });
}

// CreateNormalMapCPU, ImageImpl.h:257-322.
int ref_create_normal_map(const float* src, int64_t rows, int64_t cols,
                          float invalid_fill, float* dst) {
    return Guard([&] {
        Tensor s = Wrap(src, {rows, cols, 3}, core::Float32);
        Tensor d = Wrap(dst, {rows, cols, 3}, core::Float32);
        img::CreateNormalMapCPU(s, d, invalid_fill);
    });
}

// ToCPU, ImageImpl.h:35-85, dst Float32. src_dtype: 0 u8, 1 u16, 2 f32.
int ref_image_to_float(const void* src, int src_dtype, int64_t n, double scale,
                       double offset, float* dst) {
    return Guard([&] {
        const core::Dtype dt = src_dtype == 0   ? core::UInt8
                               : src_dtype == 1 ? core::UInt16
                                                : core::Float32;
        Tensor s = Wrap(src, {n}, dt);
        Tensor d = Wrap(dst, {n}, core::Float32);
        img::ToCPU(s, d, scale, offset);
    });
}

// ComputeOdometryResult{PointToPlane,Intensity,Hybrid}CPU,
// t/pipelines/kernel/RGBDOdometryCPU.cpp:98-364. Outputs the decoded result
// (delta pose, residual, count) and the float A_1x29 the reference reduced
// (read back through the stand-in Tensor's vector-constructor hook).
int ref_odometry(int method, int rows, int cols, const float* source_depth,
                 const float* target_depth, const float* source_intensity,
                 const float* target_intensity, const float* target_depth_dx,
                 const float* target_depth_dy, const float* target_intensity_dx,
                 const float* target_intensity_dy, const float* source_vertex,
                 const float* target_vertex, const float* target_normal,
                 const double* K, const double* T, float depth_outlier_trunc,
                 float depth_huber_delta, float intensity_huber_delta,
                 double* delta6, float* residual, int* count, double* sums29) {
    return Guard([&] {
        auto M1 = [&](const float* p) {
            return Wrap(p, {rows, cols, 1}, core::Float32);
        };
        auto M3 = [&](const float* p) {
            return Wrap(p, {rows, cols, 3}, core::Float32);
        };
        Tensor delta;
        float res = 0;
        int cnt = 0;
        core::LastVectorInit().clear();
        if (method == 0) {
            odo::ComputeOdometryResultPointToPlaneCPU(
                    M3(source_vertex), M3(target_vertex), M3(target_normal),
                    Mat(K, 3, 3), Mat(T, 4, 4), delta, res, cnt,
                    depth_outlier_trunc, depth_huber_delta);
        } else if (method == 1) {
            odo::ComputeOdometryResultIntensityCPU(
                    M1(source_depth), M1(target_depth), M1(source_intensity),
                    M1(target_intensity), M1(target_intensity_dx),
                    M1(target_intensity_dy), M3(source_vertex), Mat(K, 3, 3),
                    Mat(T, 4, 4), delta, res, cnt, depth_outlier_trunc,
                    intensity_huber_delta);
        } else {
            odo::ComputeOdometryResultHybridCPU(
                    M1(source_depth), M1(target_depth), M1(source_intensity),
                    M1(target_intensity), M1(target_depth_dx),
                    M1(target_depth_dy), M1(target_intensity_dx),
                    M1(target_intensity_dy), M3(source_vertex), Mat(K, 3, 3),
                    Mat(T, 4, 4), delta, res, cnt, depth_outlier_trunc,
                    depth_huber_delta, intensity_huber_delta);
        }
        const std::vector<double>& A = core::LastVectorInit();
        for (int i = 0; i < 29; ++i) sums29[i] = i < (int)A.size() ? A[i] : 0.0;
        for (int i = 0; i < 6; ++i) delta6[i] = delta.GetDouble(i);
        *residual = res;
        *count = cnt;
    });
}

// ComputeOdometryInformationMatrixCPU, RGBDOdometryCPU.cpp:26-96.
int ref_odometry_information(int rows, int cols, const float* source_vertex,
                             const float* target_vertex, const double* K,
                             const double* T, float square_dist_thr,
                             double* info36) {
    return Guard([&] {
        Tensor info;
        odo::ComputeOdometryInformationMatrixCPU(
                Wrap(source_vertex, {rows, cols, 3}, core::Float32),
                Wrap(target_vertex, {rows, cols, 3}, core::Float32),
                Mat(K, 3, 3), Mat(T, 4, 4), square_dist_thr, info);
        for (int i = 0; i < 36; ++i) info36[i] = info.GetDouble(i);
    });
}

}  // extern "C"
