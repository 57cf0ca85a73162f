// TEST INFRASTRUCTURE ONLY -- see oracle_common.h.
//
// CPU restatement of the VoxelBlockGrid half of the hot path:
//   DepthTouchCPU        cpp/open3d/t/geometry/kernel/VoxelBlockGridCPU.cpp:117-201
//   hash Activate/Find   cpp/open3d/core/hashmap/HashMap.cpp:166-216,
//                        cpp/open3d/core/hashmap/CPU/TBBHashBackend.h:203-247,
//                        cpp/open3d/core/hashmap/CPU/CPUHashBackendBufferAccessor.hpp:23-46
//   IntegrateCPU         cpp/open3d/t/geometry/kernel/VoxelBlockGridImpl.h:151-308
//   EstimateRangeCPU     cpp/open3d/t/geometry/kernel/VoxelBlockGridImpl.h:310-555
//   RayCastCPU           cpp/open3d/t/geometry/kernel/VoxelBlockGridImpl.h:578-1120
//   UnprojectCPU         cpp/open3d/t/geometry/kernel/PointCloudImpl.h:42-143
//
// Pinning: the reference holds no self-contained value-level golden vectors
// for these functions (its tests use downloaded Redwood frames, SURVEY.md
// section 8c). The restatement is pinned instead against the reference's own
// template bodies compiled from /root/reference through oracle/_ref (see
// oracle/ref_shim/ and oracle/Makefile target `ref`); fixtures produced by
// that build are committed under tests/golden/.
//
// All loops are written per work-item exactly as the reference's ParallelFor
// lambdas; `#pragma omp parallel for` is only a stand-in for TBB's
// parallel_for over the same index range (used by bench.py's cpu_baseline).

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "oracle_common.h"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

struct Coord3i {
    int x, y, z;
    bool operator==(const Coord3i& o) const {
        return x == o.x && y == o.y && z == o.z;
    }
    bool operator<(const Coord3i& o) const {
        if (x != o.x) return x < o.x;
        if (y != o.y) return y < o.y;
        return z < o.z;
    }
};
// VoxelBlockGridCPU.cpp:43-53 (hash only affects iteration order).
struct Coord3iHash {
    size_t operator()(const Coord3i& k) const {
        static const size_t p0 = 73856093;
        static const size_t p1 = 19349669;
        static const size_t p2 = 83492791;
        return (static_cast<size_t>(k.x) * p0) ^
               (static_cast<size_t>(k.y) * p1) ^
               (static_cast<size_t>(k.z) * p2);
    }
};

// ---------------------------------------------------------------------------
// DepthTouchCPU, VoxelBlockGridCPU.cpp:117-201. Output is sorted
// lexicographically (the reference's order is unordered-set iteration order,
// i.e. unspecified; parity is on the *set*).
template <typename depth_t>
int64_t DepthTouchImpl(const depth_t* depth, int rows, int cols,
                       const double* intrinsic, const double* extrinsic,
                       int resolution, float voxel_size, float sdf_trunc,
                       float depth_scale, float depth_max, int stride,
                       int* out_coords, int64_t out_capacity) {
    double pose[16];
    InverseTransformation(extrinsic, pose);
    TransformIndexer ti(intrinsic, pose, 1.0f);

    int rows_strided = rows / stride;
    int cols_strided = cols / stride;
    int n = rows_strided * cols_strided;
    float block_size = voxel_size * resolution;

    std::unordered_set<Coord3i, Coord3iHash> set;
    for (int workload_idx = 0; workload_idx < n; ++workload_idx) {
        int y = (workload_idx / cols_strided) * stride;
        int x = (workload_idx % cols_strided) * stride;

        float d = depth[(int64_t)y * cols + x] / depth_scale;
        if (d > 0 && d < depth_max) {
            float x_c = 0, y_c = 0, z_c = 0;
            ti.Unproject(static_cast<float>(x), static_cast<float>(y), 1.0,
                         &x_c, &y_c, &z_c);
            float x_g = 0, y_g = 0, z_g = 0;
            ti.RigidTransform(x_c, y_c, z_c, &x_g, &y_g, &z_g);

            float x_o = 0, y_o = 0, z_o = 0;
            ti.GetCameraPosition(&x_o, &y_o, &z_o);

            float x_d = x_g - x_o;
            float y_d = y_g - y_o;
            float z_d = z_g - z_o;

            const int step_size = 3;
            const float t_min = std::max(d - sdf_trunc, 0.0f);
            const float t_max = std::min(d + sdf_trunc, depth_max);
            const float t_step = (t_max - t_min) / step_size;

            float t = t_min;
            for (int step = 0; step <= step_size; ++step) {
                int xb = static_cast<int>(
                        std::floor((x_o + t * x_d) / block_size));
                int yb = static_cast<int>(
                        std::floor((y_o + t * y_d) / block_size));
                int zb = static_cast<int>(
                        std::floor((z_o + t * z_d) / block_size));
                set.insert(Coord3i{xb, yb, zb});
                t += t_step;
            }
        }
    }
    std::vector<Coord3i> v(set.begin(), set.end());
    std::sort(v.begin(), v.end());
    int64_t count = (int64_t)v.size();
    for (int64_t i = 0; i < count && i < out_capacity; ++i) {
        out_coords[3 * i + 0] = v[i].x;
        out_coords[3 * i + 1] = v[i].y;
        out_coords[3 * i + 2] = v[i].z;
    }
    return count;
}

// PointCloudTouchCPU, VoxelBlockGridCPU.cpp:55-115.
int64_t PointCloudTouchImpl(const float* pcd, int64_t n, int resolution,
                            float voxel_size, float sdf_trunc, int* out_coords,
                            int64_t out_capacity) {
    float block_size = voxel_size * resolution;
    std::unordered_set<Coord3i, Coord3iHash> set;
    for (int64_t i = 0; i < n; ++i) {
        float x = pcd[3 * i + 0], y = pcd[3 * i + 1], z = pcd[3 * i + 2];
        int xb_lo = static_cast<int>(std::floor((x - sdf_trunc) / block_size));
        int xb_hi = static_cast<int>(std::floor((x + sdf_trunc) / block_size));
        int yb_lo = static_cast<int>(std::floor((y - sdf_trunc) / block_size));
        int yb_hi = static_cast<int>(std::floor((y + sdf_trunc) / block_size));
        int zb_lo = static_cast<int>(std::floor((z - sdf_trunc) / block_size));
        int zb_hi = static_cast<int>(std::floor((z + sdf_trunc) / block_size));
        for (int xb = xb_lo; xb <= xb_hi; ++xb)
            for (int yb = yb_lo; yb <= yb_hi; ++yb)
                for (int zb = zb_lo; zb <= zb_hi; ++zb)
                    set.insert(Coord3i{xb, yb, zb});
    }
    std::vector<Coord3i> v(set.begin(), set.end());
    std::sort(v.begin(), v.end());
    int64_t count = (int64_t)v.size();
    for (int64_t i = 0; i < count && i < out_capacity; ++i) {
        out_coords[3 * i + 0] = v[i].x;
        out_coords[3 * i + 1] = v[i].y;
        out_coords[3 * i + 2] = v[i].z;
    }
    return count;
}

// ---------------------------------------------------------------------------
// IntegrateCPU, VoxelBlockGridImpl.h:151-308. One iteration == one
// workload_idx of the reference's ParallelFor.
template <typename input_depth_t, typename input_color_t, typename weight_t,
          typename color_t>
void IntegrateImpl(const input_depth_t* depth, int depth_rows, int depth_cols,
                   const input_color_t* color, int color_rows, int color_cols,
                   const int* indices, int64_t n_indices,
                   const int* block_keys, float* tsdf_base_ptr,
                   weight_t* weight_base_ptr, color_t* color_base_ptr,
                   const double* depth_intrinsic, const double* color_intrinsic,
                   const double* extrinsics, int resolution, float voxel_size,
                   float sdf_trunc, float depth_scale, float depth_max) {
    using tsdf_t = float;
    int64_t resolution2 = (int64_t)resolution * resolution;
    int64_t resolution3 = resolution2 * resolution;

    TransformIndexer transform_indexer(depth_intrinsic, extrinsics, voxel_size);
    static const double eye4[16] = {1, 0, 0, 0, 0, 1, 0, 0,
                                    0, 0, 1, 0, 0, 0, 0, 1};
    TransformIndexer colormap_indexer(color_intrinsic, eye4);

    bool integrate_color = color_base_ptr != nullptr && color != nullptr &&
                           (int64_t)color_rows * color_cols > 0;
    float color_multiplier = 1.0;
    if (integrate_color && std::is_same<input_color_t, float>::value) {
        color_multiplier = 255.0;
    }

    int64_t n = n_indices * resolution3;
#pragma omp parallel for schedule(static)
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        int64_t block_idx = indices[workload_idx / resolution3];
        int64_t voxel_idx = workload_idx % resolution3;

        const int* block_key_ptr = block_keys + 3 * block_idx;
        int xb = block_key_ptr[0];
        int yb = block_key_ptr[1];
        int zb = block_key_ptr[2];

        // TArrayIndexer::WorkloadToCoord (3D), GeometryIndexer.h:270-278
        int xv = (int)(voxel_idx % resolution);
        int yv = (int)((voxel_idx / resolution) % resolution);
        int zv = (int)(voxel_idx / resolution2);

        int x = xb * resolution + xv;
        int y = yb * resolution + yv;
        int z = zb * resolution + zv;

        float xc, yc, zc, u, v;
        transform_indexer.RigidTransform(static_cast<float>(x),
                                         static_cast<float>(y),
                                         static_cast<float>(z), &xc, &yc, &zc);
        transform_indexer.Project(xc, yc, zc, &u, &v);
        if (!InBoundary2D(u, v, depth_rows, depth_cols)) continue;

        int ui = static_cast<int>(u);
        int vi = static_cast<int>(v);

        float depth_v = depth[(int64_t)vi * depth_cols + ui] / depth_scale;

        float sdf = depth_v - zc;
        if (depth_v <= 0 || depth_v > depth_max || zc <= 0 ||
            sdf < -sdf_trunc) {
            continue;
        }
        sdf = sdf < sdf_trunc ? sdf : sdf_trunc;
        sdf /= sdf_trunc;

        int64_t linear_idx = block_idx * resolution3 + voxel_idx;

        tsdf_t* tsdf_ptr = tsdf_base_ptr + linear_idx;
        weight_t* weight_ptr = weight_base_ptr + linear_idx;

        float inv_wsum = 1.0f / (*weight_ptr + 1);
        float weight = *weight_ptr;
        *tsdf_ptr = (weight * (*tsdf_ptr) + sdf) * inv_wsum;

        if (integrate_color) {
            color_t* color_ptr = color_base_ptr + 3 * linear_idx;

            float xx, yy, zz;
            transform_indexer.Unproject(ui, vi, 1.0, &xx, &yy, &zz);

            float uf, vf;
            colormap_indexer.Project(xx, yy, zz, &uf, &vf);
            if (InBoundary2D(uf, vf, color_rows, color_cols)) {
                ui = round(uf);
                vi = round(vf);

                const input_color_t* input_color_ptr =
                        color + ((int64_t)vi * color_cols + ui) * 3;

                for (int i = 0; i < 3; ++i) {
                    color_ptr[i] = (weight * color_ptr[i] +
                                    input_color_ptr[i] * color_multiplier) *
                                   inv_wsum;
                }
            }
        }
        *weight_ptr = weight + 1;
    }
}

// ---------------------------------------------------------------------------
// EstimateRangeCPU, VoxelBlockGridImpl.h:310-555.
// Returns needed fragment count (reference reallocates for the *next* call
// when needed >= capacity, and drops fragments for this one).
int EstimateRangeImpl(const int* block_keys, int64_t n_blocks,
                      float* range_minmax_map /* h_down*w_down*2 */,
                      const double* intrinsics, const double* extrinsics,
                      int h, int w, int down_factor, int64_t block_resolution,
                      float voxel_size, float depth_min, float depth_max,
                      int frag_buffer_size) {
    int h_down = h / down_factor;
    int w_down = w / down_factor;
    const int fragment_size = 16;

    if (frag_buffer_size <= 0) {
        // VoxelBlockGridImpl.h:342-349
        frag_buffer_size =
                h_down * w_down / (fragment_size * fragment_size) / voxel_size;
    }
    std::vector<float> fragment_buffer((size_t)frag_buffer_size * 6);
    TransformIndexer w2c_transform_indexer(intrinsics, extrinsics);
    int count = 0;

    using std::max;
    using std::min;

    // Pass 0
    for (int64_t workload_idx = 0; workload_idx < n_blocks; ++workload_idx) {
        const int* key = block_keys + 3 * workload_idx;

        int u_min = w_down - 1, v_min = h_down - 1, u_max = 0, v_max = 0;
        float z_min = depth_max, z_max = depth_min;
        float xc, yc, zc, u, v;

        for (int i = 0; i < 8; ++i) {
            float xw = (key[0] + ((i & 1) > 0)) * block_resolution * voxel_size;
            float yw = (key[1] + ((i & 2) > 0)) * block_resolution * voxel_size;
            float zw = (key[2] + ((i & 4) > 0)) * block_resolution * voxel_size;

            w2c_transform_indexer.RigidTransform(xw, yw, zw, &xc, &yc, &zc);
            if (zc <= 0) continue;

            w2c_transform_indexer.Project(xc, yc, zc, &u, &v);
            u /= down_factor;
            v /= down_factor;

            v_min = min(static_cast<int>(floorf(v)), v_min);
            v_max = max(static_cast<int>(ceilf(v)), v_max);
            u_min = min(static_cast<int>(floorf(u)), u_min);
            u_max = max(static_cast<int>(ceilf(u)), u_max);
            z_min = min(z_min, zc);
            z_max = max(z_max, zc);
        }

        v_min = max(0, v_min);
        v_max = min(h_down - 1, v_max);
        u_min = max(0, u_min);
        u_max = min(w_down - 1, u_max);

        if (v_min >= v_max || u_min >= u_max || z_min >= z_max) continue;

        int frag_v_count =
                ceilf(float(v_max - v_min + 1) / float(fragment_size));
        int frag_u_count =
                ceilf(float(u_max - u_min + 1) / float(fragment_size));

        int frag_count = frag_v_count * frag_u_count;
        int frag_count_start = count;
        count += frag_count;
        int frag_count_end = frag_count_start + frag_count;
        if (frag_count_end >= frag_buffer_size) continue;

        int offset = 0;
        for (int frag_v = 0; frag_v < frag_v_count; ++frag_v) {
            for (int frag_u = 0; frag_u < frag_u_count; ++frag_u, ++offset) {
                float* frag_ptr =
                        &fragment_buffer[(size_t)(frag_count_start + offset) *
                                         6];
                frag_ptr[0] = z_min;
                frag_ptr[1] = z_max;
                frag_ptr[2] = v_min + frag_v * fragment_size;
                frag_ptr[3] = u_min + frag_u * fragment_size;
                frag_ptr[4] = min(frag_ptr[2] + fragment_size - 1,
                                  static_cast<float>(v_max));
                frag_ptr[5] = min(frag_ptr[3] + fragment_size - 1,
                                  static_cast<float>(u_max));
            }
        }
    }
    int needed_frag_count = count;
    int frag_count = needed_frag_count;
    if (frag_count >= frag_buffer_size) {
        frag_count = frag_buffer_size - 1;
    }

    // Pass 0.5
    for (int64_t workload_idx = 0; workload_idx < (int64_t)h_down * w_down;
         ++workload_idx) {
        float* range_ptr = range_minmax_map + 2 * workload_idx;
        range_ptr[0] = depth_max;
        range_ptr[1] = depth_min;
    }

    // Pass 1
    for (int64_t workload_idx = 0;
         workload_idx < (int64_t)frag_count * fragment_size * fragment_size;
         ++workload_idx) {
        int frag_idx = workload_idx / (fragment_size * fragment_size);
        int local_idx = workload_idx % (fragment_size * fragment_size);
        int dv = local_idx / fragment_size;
        int du = local_idx % fragment_size;

        float* frag_ptr = &fragment_buffer[(size_t)frag_idx * 6];
        int v_min = static_cast<int>(frag_ptr[2]);
        int u_min = static_cast<int>(frag_ptr[3]);
        int v_max = static_cast<int>(frag_ptr[4]);
        int u_max = static_cast<int>(frag_ptr[5]);

        int v = v_min + dv;
        int u = u_min + du;
        if (v > v_max || u > u_max) continue;

        float z_min = frag_ptr[0];
        float z_max = frag_ptr[1];
        float* range_ptr = range_minmax_map + 2 * ((int64_t)v * w_down + u);
        range_ptr[0] = min(z_min, range_ptr[0]);
        range_ptr[1] = max(z_max, range_ptr[1]);
    }
    return needed_frag_count;
}

// ---------------------------------------------------------------------------
// RayCastCPU, VoxelBlockGridImpl.h:578-1120.
struct MiniVecCache {
    int x, y, z, block_idx;
    int Check(int xin, int yin, int zin) {
        return (xin == x && yin == y && zin == z) ? block_idx : -1;
    }
    void Update(int xin, int yin, int zin, int b) {
        x = xin; y = yin; z = zin; block_idx = b;
    }
};

using BlockMap = std::unordered_map<Coord3i, int, Coord3iHash>;

struct RayCastOutputs {
    float* depth;    // {h,w,1} or null
    float* vertex;   // {h,w,3}
    float* color;    // {h,w,3}
    float* normal;   // {h,w,3}
    int64_t* index;  // {h,w,8}
    uint8_t* mask;   // {h,w,8} (bool)
    float* interp_ratio;
    float* interp_ratio_dx;
    float* interp_ratio_dy;
    float* interp_ratio_dz;
};

template <typename weight_t, typename color_t>
void RayCastImpl(const BlockMap& hashmap_impl, const float* tsdf_base_ptr,
                 const weight_t* weight_base_ptr, const color_t* color_base_ptr,
                 const float* range_map, const RayCastOutputs& out,
                 const double* intrinsic, const double* extrinsics, int h,
                 int w, int block_resolution, float voxel_size,
                 float depth_scale, float depth_min, float depth_max,
                 float weight_threshold, float trunc_voxel_multiplier,
                 int range_map_down_factor) {
    (void)depth_min;
    (void)depth_max;
    bool render_color = color_base_ptr != nullptr && out.color != nullptr;
    bool visit_neighbors = render_color || out.normal || out.mask ||
                           out.index || out.interp_ratio ||
                           out.interp_ratio_dx || out.interp_ratio_dy ||
                           out.interp_ratio_dz;

    double pose[16];
    InverseTransformation(extrinsics, pose);
    TransformIndexer c2w_transform_indexer(intrinsic, pose);
    TransformIndexer w2c_transform_indexer(intrinsic, extrinsics);

    int rows = h, cols = w;
    int64_t n = (int64_t)rows * cols;
    int w_down = w / range_map_down_factor;

    float block_size = voxel_size * block_resolution;
    int resolution2 = block_resolution * block_resolution;
    int resolution3 = resolution2 * block_resolution;

    using std::max;
    using std::sqrt;

#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        auto find = [&](int xb, int yb, int zb) -> int {
            auto it = hashmap_impl.find(Coord3i{xb, yb, zb});
            if (it == hashmap_impl.end()) return -1;
            return it->second;
        };

        auto GetLinearIdxAtP = [&](int x_b, int y_b, int z_b, int x_v, int y_v,
                                   int z_v, int block_buf_idx,
                                   MiniVecCache& cache) -> int {
            int x_vn = (x_v + block_resolution) % block_resolution;
            int y_vn = (y_v + block_resolution) % block_resolution;
            int z_vn = (z_v + block_resolution) % block_resolution;

            int dx_b = Sign(x_v - x_vn);
            int dy_b = Sign(y_v - y_vn);
            int dz_b = Sign(z_v - z_vn);

            if (dx_b == 0 && dy_b == 0 && dz_b == 0) {
                return block_buf_idx * resolution3 + z_v * resolution2 +
                       y_v * block_resolution + x_v;
            } else {
                int kx = x_b + dx_b, ky = y_b + dy_b, kz = z_b + dz_b;
                int nb_buf_idx = cache.Check(kx, ky, kz);
                if (nb_buf_idx < 0) {
                    nb_buf_idx = find(kx, ky, kz);
                    if (nb_buf_idx < 0) return -1;
                    cache.Update(kx, ky, kz, nb_buf_idx);
                }
                return nb_buf_idx * resolution3 + z_vn * resolution2 +
                       y_vn * block_resolution + x_vn;
            }
        };

        auto GetLinearIdxAtT = [&](float x_o, float y_o, float z_o, float x_d,
                                   float y_d, float z_d, float t,
                                   MiniVecCache& cache) -> int {
            float x_g = x_o + t * x_d;
            float y_g = y_o + t * y_d;
            float z_g = z_o + t * z_d;

            int x_b = static_cast<int>(floorf(x_g / block_size));
            int y_b = static_cast<int>(floorf(y_g / block_size));
            int z_b = static_cast<int>(floorf(z_g / block_size));

            int block_buf_idx = cache.Check(x_b, y_b, z_b);
            if (block_buf_idx < 0) {
                block_buf_idx = find(x_b, y_b, z_b);
                if (block_buf_idx < 0) return -1;
                cache.Update(x_b, y_b, z_b, block_buf_idx);
            }

            int x_v = int((x_g - x_b * block_size) / voxel_size);
            int y_v = int((y_g - y_b * block_size) / voxel_size);
            int z_v = int((z_g - z_b * block_size) / voxel_size);

            return block_buf_idx * resolution3 + z_v * resolution2 +
                   y_v * block_resolution + x_v;
        };

        int y = (int)(workload_idx / cols);
        int x = (int)(workload_idx % cols);

        const float* range =
                range_map + 2 * ((int64_t)(y / range_map_down_factor) * w_down +
                                 (x / range_map_down_factor));

        float* depth_ptr = nullptr;
        float* vertex_ptr = nullptr;
        float* color_ptr = nullptr;
        float* normal_ptr = nullptr;
        int64_t* index_ptr = nullptr;
        uint8_t* mask_ptr = nullptr;
        float* interp_ratio_ptr = nullptr;
        float* interp_ratio_dx_ptr = nullptr;
        float* interp_ratio_dy_ptr = nullptr;
        float* interp_ratio_dz_ptr = nullptr;

        if (out.vertex) {
            vertex_ptr = out.vertex + 3 * workload_idx;
            vertex_ptr[0] = 0; vertex_ptr[1] = 0; vertex_ptr[2] = 0;
        }
        if (out.depth) {
            depth_ptr = out.depth + workload_idx;
            depth_ptr[0] = 0;
        }
        if (out.normal) {
            normal_ptr = out.normal + 3 * workload_idx;
            normal_ptr[0] = 0; normal_ptr[1] = 0; normal_ptr[2] = 0;
        }
        if (out.mask) {
            mask_ptr = out.mask + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) mask_ptr[i] = 0;
        }
        if (out.index) {
            index_ptr = out.index + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) index_ptr[i] = 0;
        }
        if (out.interp_ratio) {
            interp_ratio_ptr = out.interp_ratio + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) interp_ratio_ptr[i] = 0;
        }
        if (out.interp_ratio_dx) {
            interp_ratio_dx_ptr = out.interp_ratio_dx + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) interp_ratio_dx_ptr[i] = 0;
        }
        if (out.interp_ratio_dy) {
            interp_ratio_dy_ptr = out.interp_ratio_dy + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) interp_ratio_dy_ptr[i] = 0;
        }
        if (out.interp_ratio_dz) {
            interp_ratio_dz_ptr = out.interp_ratio_dz + 8 * workload_idx;
            for (int i = 0; i < 8; ++i) interp_ratio_dz_ptr[i] = 0;
        }
        if (out.color) {
            color_ptr = out.color + 3 * workload_idx;
            color_ptr[0] = 0; color_ptr[1] = 0; color_ptr[2] = 0;
        }

        float t = range[0];
        const float t_max = range[1];
        if (t >= t_max) continue;

        float x_c = 0, y_c = 0, z_c = 0;
        float x_g = 0, y_g = 0, z_g = 0;
        float x_o = 0, y_o = 0, z_o = 0;

        float t_prev = t;
        float tsdf_prev = -1.0f;
        float tsdf = 1.0;
        float sdf_trunc = voxel_size * trunc_voxel_multiplier;
        float wgt = 0.0;

        c2w_transform_indexer.RigidTransform(0, 0, 0, &x_o, &y_o, &z_o);
        c2w_transform_indexer.Unproject(static_cast<float>(x),
                                        static_cast<float>(y), 1.0f, &x_c, &y_c,
                                        &z_c);
        c2w_transform_indexer.RigidTransform(x_c, y_c, z_c, &x_g, &y_g, &z_g);
        float x_d = (x_g - x_o);
        float y_d = (y_g - y_o);
        float z_d = (z_g - z_o);

        MiniVecCache cache{0, 0, 0, -1};
        bool surface_found = false;
        while (t < t_max) {
            int linear_idx =
                    GetLinearIdxAtT(x_o, y_o, z_o, x_d, y_d, z_d, t, cache);

            if (linear_idx < 0) {
                t_prev = t;
                t += block_size;
            } else {
                tsdf_prev = tsdf;
                tsdf = tsdf_base_ptr[linear_idx];
                wgt = weight_base_ptr[linear_idx];
                if (tsdf_prev > 0 && wgt >= weight_threshold && tsdf <= 0) {
                    surface_found = true;
                    break;
                }
                t_prev = t;
                float delta = tsdf * sdf_trunc;
                t += delta < voxel_size ? voxel_size : delta;
            }
        }

        if (surface_found) {
            float t_intersect =
                    (t * tsdf_prev - t_prev * tsdf) / (tsdf_prev - tsdf);
            x_g = x_o + t_intersect * x_d;
            y_g = y_o + t_intersect * y_d;
            z_g = z_o + t_intersect * z_d;

            if (depth_ptr) {
                *depth_ptr = t_intersect * depth_scale;
            }
            if (vertex_ptr) {
                w2c_transform_indexer.RigidTransform(x_g, y_g, z_g,
                                                     vertex_ptr + 0,
                                                     vertex_ptr + 1,
                                                     vertex_ptr + 2);
            }
            if (!visit_neighbors) continue;

            int x_b = static_cast<int>(floorf(x_g / block_size));
            int y_b = static_cast<int>(floorf(y_g / block_size));
            int z_b = static_cast<int>(floorf(z_g / block_size));
            float x_v = (x_g - float(x_b) * block_size) / voxel_size;
            float y_v = (y_g - float(y_b) * block_size) / voxel_size;
            float z_v = (z_g - float(z_b) * block_size) / voxel_size;

            int block_buf_idx = cache.Check(x_b, y_b, z_b);
            if (block_buf_idx < 0) {
                block_buf_idx = find(x_b, y_b, z_b);
                if (block_buf_idx < 0) continue;
                cache.Update(x_b, y_b, z_b, block_buf_idx);
            }

            int x_v_floor = static_cast<int>(floorf(x_v));
            int y_v_floor = static_cast<int>(floorf(y_v));
            int z_v_floor = static_cast<int>(floorf(z_v));

            float ratio_x = x_v - float(x_v_floor);
            float ratio_y = y_v - float(y_v_floor);
            float ratio_z = z_v - float(z_v_floor);

            float sum_r = 0.0;
            for (int k = 0; k < 8; ++k) {
                int dx_v = (k & 1) > 0 ? 1 : 0;
                int dy_v = (k & 2) > 0 ? 1 : 0;
                int dz_v = (k & 4) > 0 ? 1 : 0;

                int linear_idx_k = GetLinearIdxAtP(
                        x_b, y_b, z_b, x_v_floor + dx_v, y_v_floor + dy_v,
                        z_v_floor + dz_v, block_buf_idx, cache);

                if (linear_idx_k >= 0 && weight_base_ptr[linear_idx_k] > 0) {
                    float rx = dx_v * (ratio_x) + (1 - dx_v) * (1 - ratio_x);
                    float ry = dy_v * (ratio_y) + (1 - dy_v) * (1 - ratio_y);
                    float rz = dz_v * (ratio_z) + (1 - dz_v) * (1 - ratio_z);
                    float r = rx * ry * rz;

                    if (interp_ratio_ptr) interp_ratio_ptr[k] = r;
                    if (mask_ptr) mask_ptr[k] = 1;
                    if (index_ptr) index_ptr[k] = linear_idx_k;

                    float tsdf_k = tsdf_base_ptr[linear_idx_k];
                    float interp_ratio_dx = ry * rz * (2 * dx_v - 1);
                    float interp_ratio_dy = rx * rz * (2 * dy_v - 1);
                    float interp_ratio_dz = rx * ry * (2 * dz_v - 1);

                    if (interp_ratio_dx_ptr)
                        interp_ratio_dx_ptr[k] = interp_ratio_dx;
                    if (interp_ratio_dy_ptr)
                        interp_ratio_dy_ptr[k] = interp_ratio_dy;
                    if (interp_ratio_dz_ptr)
                        interp_ratio_dz_ptr[k] = interp_ratio_dz;

                    if (normal_ptr) {
                        normal_ptr[0] += interp_ratio_dx * tsdf_k;
                        normal_ptr[1] += interp_ratio_dy * tsdf_k;
                        normal_ptr[2] += interp_ratio_dz * tsdf_k;
                    }

                    if (color_ptr && render_color) {
                        int64_t color_linear_idx = (int64_t)linear_idx_k * 3;
                        color_ptr[0] +=
                                r * color_base_ptr[color_linear_idx + 0];
                        color_ptr[1] +=
                                r * color_base_ptr[color_linear_idx + 1];
                        color_ptr[2] +=
                                r * color_base_ptr[color_linear_idx + 2];
                    }

                    sum_r += r;
                }
            }

            if (sum_r > 0) {
                sum_r *= 255.0;
                if (color_ptr && render_color) {
                    color_ptr[0] /= sum_r;
                    color_ptr[1] /= sum_r;
                    color_ptr[2] /= sum_r;
                }

                if (normal_ptr) {
                    constexpr float EPSILON = 1e-5f;
                    float norm = sqrt(normal_ptr[0] * normal_ptr[0] +
                                      normal_ptr[1] * normal_ptr[1] +
                                      normal_ptr[2] * normal_ptr[2]);
                    norm = max(norm, EPSILON);
                    w2c_transform_indexer.Rotate(
                            -normal_ptr[0] / norm, -normal_ptr[1] / norm,
                            -normal_ptr[2] / norm, normal_ptr + 0,
                            normal_ptr + 1, normal_ptr + 2);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// UnprojectCPU, t/geometry/kernel/PointCloudImpl.h:42-143 (depth only / with
// colours). Output order here is row-major scan order (the reference's is
// atomic-counter order, i.e. unspecified).
template <typename depth_t>
int64_t UnprojectImpl(const depth_t* depth, int rows, int cols,
                      const float* image_colors_f32, float* points,
                      float* colors, const double* intrinsics,
                      const double* extrinsics, float depth_scale,
                      float depth_max, int64_t stride) {
    double pose[16];
    InverseTransformation(extrinsics, pose);
    TransformIndexer ti(intrinsics, pose, 1.0f);

    int64_t rows_strided = rows / stride;
    int64_t cols_strided = cols / stride;
    int64_t n = rows_strided * cols_strided;
    int64_t count = 0;
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        int64_t y = (workload_idx / cols_strided) * stride;
        int64_t x = (workload_idx % cols_strided) * stride;

        float d = depth[y * cols + x] / depth_scale;
        if (d > 0 && d < depth_max) {
            int64_t idx = count++;
            float x_c = 0, y_c = 0, z_c = 0;
            ti.Unproject(static_cast<float>(x), static_cast<float>(y), d, &x_c,
                         &y_c, &z_c);
            float* vertex = points + 3 * idx;
            ti.RigidTransform(x_c, y_c, z_c, vertex + 0, vertex + 1,
                              vertex + 2);
            if (colors && image_colors_f32) {
                // PointCloudImpl.h:120-128: colours are Float32 images here
                // (the caller, PointCloud::CreateFromRGBDImage, rescales u8).
                float* pcd_pixel = colors + 3 * idx;
                const float* image_pixel = image_colors_f32 + 3 * (y * cols + x);
                *pcd_pixel = *image_pixel;
                *(pcd_pixel + 1) = *(image_pixel + 1);
                *(pcd_pixel + 2) = *(image_pixel + 2);
            }
        }
    }
    return count;
}

// ---------------------------------------------------------------------------
// ExtractPointCloudCPU, VoxelBlockGridImpl.h:1122-1365 with DeviceGetLinearIdx
// / DeviceGetNormal (:94-149). Sequential: output order = workload order
// (active block, voxel, axis); the reference's order is whatever its atomic
// counter hands out. nb_indices / nb_masks are the {27, n} tables of
// BufferRadiusNeighbors (VoxelBlockGrid.cpp:22-51).
//
// `sqrt` in ExtractPointCloudCPU is unqualified; with libstdc++'s <cmath> the
// float overload is visible in the global namespace, so this is a float sqrt
// whose result is then added to the double literal 1e-5 (checked against the
// compiled reference body: the float -> double variant differs by 1 ulp).
#define EXTRACT_SQRT(v) std::sqrt((float)(v))

inline int64_t ExtractLinearIdx(int xo, int yo, int zo, int64_t curr_block_idx,
                                int resolution, int64_t n_blocks,
                                const int* nb_indices, const uint8_t* nb_masks) {
    int xn = (xo + resolution) % resolution;
    int yn = (yo + resolution) % resolution;
    int zn = (zo + resolution) % resolution;
    int dxb = orc::Sign(xo - xn);
    int dyb = orc::Sign(yo - yn);
    int dzb = orc::Sign(zo - zn);
    int nb_idx = (dxb + 1) + (dyb + 1) * 3 + (dzb + 1) * 9;
    if (!nb_masks[(int64_t)nb_idx * n_blocks + curr_block_idx]) return -1;
    int64_t block_idx_i = nb_indices[(int64_t)nb_idx * n_blocks + curr_block_idx];
    return (((block_idx_i * resolution) + zn) * resolution + yn) * resolution +
           xn;
}

template <typename weight_t, typename color_t>
int64_t ExtractPointCloudImpl(const int* indices, const int* nb_indices,
                              const uint8_t* nb_masks, const int* block_keys,
                              const float* tsdf_base_ptr,
                              const weight_t* weight_base_ptr,
                              const color_t* color_base_ptr, int64_t n_blocks,
                              int resolution, float voxel_size,
                              float weight_threshold, float* points,
                              float* normals, float* colors,
                              int64_t valid_size) {
    const int64_t resolution3 = (int64_t)resolution * resolution * resolution;
    auto L = [&](int xo, int yo, int zo, int64_t b) {
        return ExtractLinearIdx(xo, yo, zo, b, resolution, n_blocks, nb_indices,
                                nb_masks);
    };
    auto GetNormal = [&](int xo, int yo, int zo, int64_t b, float* n) {
        int64_t vxp = L(xo + 1, yo, zo, b), vxn = L(xo - 1, yo, zo, b);
        int64_t vyp = L(xo, yo + 1, zo, b), vyn = L(xo, yo - 1, zo, b);
        int64_t vzp = L(xo, yo, zo + 1, b), vzn = L(xo, yo, zo - 1, b);
        if (vxp >= 0 && vxn >= 0) n[0] = tsdf_base_ptr[vxp] - tsdf_base_ptr[vxn];
        if (vyp >= 0 && vyn >= 0) n[1] = tsdf_base_ptr[vyp] - tsdf_base_ptr[vyn];
        if (vzp >= 0 && vzn >= 0) n[2] = tsdf_base_ptr[vzp] - tsdf_base_ptr[vzn];
    };
    int64_t count = 0;
    const int64_t n = n_blocks * resolution3;
    for (int64_t workload_idx = 0; workload_idx < n; ++workload_idx) {
        int64_t workload_block_idx = workload_idx / resolution3;
        int64_t block_idx = indices[workload_block_idx];
        int64_t voxel_idx = workload_idx % resolution3;
        const int* key = block_keys + 3 * block_idx;
        int xb = key[0], yb = key[1], zb = key[2];
        int xv = (int)(voxel_idx % resolution);
        int yv = (int)((voxel_idx / resolution) % resolution);
        int zv = (int)(voxel_idx / ((int64_t)resolution * resolution));
        int64_t linear_idx = block_idx * resolution3 + voxel_idx;
        float tsdf_o = tsdf_base_ptr[linear_idx];
        float weight_o = weight_base_ptr[linear_idx];
        if (weight_o <= weight_threshold) continue;
        // no / ne persist across the three axes of a voxel exactly as in the
        // reference (ne is NOT reset between axes).
        float no[3] = {0}, ne[3] = {0};
        if (points) GetNormal(xv, yv, zv, workload_block_idx, no);
        int x = xb * resolution + xv;
        int y = yb * resolution + yv;
        int z = zb * resolution + zv;
        for (int i = 0; i < 3; ++i) {
            int64_t linear_idx_i = L(xv + (i == 0), yv + (i == 1), zv + (i == 2),
                                     workload_block_idx);
            if (linear_idx_i < 0) continue;
            float tsdf_i = tsdf_base_ptr[linear_idx_i];
            float weight_i = weight_base_ptr[linear_idx_i];
            if (weight_i > weight_threshold && tsdf_i * tsdf_o < 0) {
                int64_t idx = count++;
                if (!points) continue;         // counting pass
                if (idx >= valid_size) break;  // the reference `return`s
                float ratio = (0 - tsdf_o) / (tsdf_i - tsdf_o);
                float* point_ptr = points + 3 * idx;
                point_ptr[0] = voxel_size * (x + ratio * int(i == 0));
                point_ptr[1] = voxel_size * (y + ratio * int(i == 1));
                point_ptr[2] = voxel_size * (z + ratio * int(i == 2));
                float* normal_ptr = normals + 3 * idx;
                GetNormal(xv + (i == 0), yv + (i == 1), zv + (i == 2),
                          workload_block_idx, ne);
                float nx = (1 - ratio) * no[0] + ratio * ne[0];
                float ny = (1 - ratio) * no[1] + ratio * ne[1];
                float nz = (1 - ratio) * no[2] + ratio * ne[2];
                float norm = static_cast<float>(
                        EXTRACT_SQRT(nx * nx + ny * ny + nz * nz) + 1e-5);
                normal_ptr[0] = nx / norm;
                normal_ptr[1] = ny / norm;
                normal_ptr[2] = nz / norm;
                if (color_base_ptr && colors) {
                    float* color_ptr = colors + 3 * idx;
                    const color_t* color_o_ptr = color_base_ptr + 3 * linear_idx;
                    float r_o = color_o_ptr[0];
                    float g_o = color_o_ptr[1];
                    float b_o = color_o_ptr[2];
                    const color_t* color_i_ptr =
                            color_base_ptr + 3 * linear_idx_i;
                    float r_i = color_i_ptr[0];
                    float g_i = color_i_ptr[1];
                    float b_i = color_i_ptr[2];
                    color_ptr[0] = ((1 - ratio) * r_o + ratio * r_i) / 255.0f;
                    color_ptr[1] = ((1 - ratio) * g_o + ratio * g_i) / 255.0f;
                    color_ptr[2] = ((1 - ratio) * b_o + ratio * b_i) / 255.0f;
                }
            }
        }
    }
    return count;
}

}  // namespace

// ===========================================================================
// C entry points (ctypes). dtype codes: 0=u16 depth + u8 colour input,
// 1=f32 depth + f32 colour input; grid 0=(f32,u16,u16), 1=(f32,f32,f32).
extern "C" {

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_inverse_transformation(const double* T, double* Tinv) {
    InverseTransformation(T, Tinv);
}

int64_t orc_depth_touch(const void* depth, int depth_is_f32, int rows, int cols,
                        const double* intrinsic, const double* extrinsic,
                        int resolution, float voxel_size, float sdf_trunc,
                        float depth_scale, float depth_max, int stride,
                        int* out_coords, int64_t out_capacity) {
    if (depth_is_f32)
        return DepthTouchImpl<float>((const float*)depth, rows, cols, intrinsic,
                                     extrinsic, resolution, voxel_size,
                                     sdf_trunc, depth_scale, depth_max, stride,
                                     out_coords, out_capacity);
    return DepthTouchImpl<uint16_t>((const uint16_t*)depth, rows, cols,
                                    intrinsic, extrinsic, resolution,
                                    voxel_size, sdf_trunc, depth_scale,
                                    depth_max, stride, out_coords,
                                    out_capacity);
}

int64_t orc_pointcloud_touch(const float* pcd, int64_t n, int resolution,
                             float voxel_size, float sdf_trunc, int* out_coords,
                             int64_t out_capacity) {
    return PointCloudTouchImpl(pcd, n, resolution, voxel_size, sdf_trunc,
                               out_coords, out_capacity);
}

// GetVoxelCoordinatesAndFlattenedIndicesCPU, VoxelBlockGridImpl.h:43-92:
// workload w -> block buf_indices[w / res^3], voxel w % res^3 (x fastest,
// WorkloadToCoord of a {res, res, res} indexer, GeometryIndexer.h:270-278);
// coordinate (key * res + voxel) * voxel_size with the integer part formed in
// int; flattened index block * res^3 + voxel in the reference's index_t (int:
// the oracle keeps that width, callers stay below 2^31 voxels).
void orc_voxel_coords_flat(const int* buf_indices, int64_t n_blocks,
                           const int* block_keys, int resolution,
                           float voxel_size, float* voxel_coords,
                           int64_t* flattened) {
    const int res3 = resolution * resolution * resolution;
    const int n = (int)(n_blocks * res3);
    for (int w = 0; w < n; ++w) {
        const int block_idx = buf_indices[w / res3];
        const int voxel_idx = w % res3;
        const int xb = block_keys[block_idx * 3 + 0];
        const int yb = block_keys[block_idx * 3 + 1];
        const int zb = block_keys[block_idx * 3 + 2];
        int rem = voxel_idx;
        const int xv = rem % resolution;
        rem /= resolution;
        const int yv = rem % resolution;
        rem /= resolution;
        const int zv = rem;
        flattened[w] = block_idx * res3 + voxel_idx;
        voxel_coords[w * 3 + 0] = (xb * resolution + xv) * voxel_size;
        voxel_coords[w * 3 + 1] = (yb * resolution + yv) * voxel_size;
        voxel_coords[w * 3 + 2] = (zb * resolution + zv) * voxel_size;
    }
}

// --- hash map (insert-if-absent, heap-ordered buffer indices) --------------
// HashMap.cpp:166-216 + TBBHashBackend.h:203-247 + buffer accessor heap
// (heap initialised to identity; DeviceAllocate = heap[top++]).
struct OrcHashMap {
    BlockMap map;
    std::vector<int> keys;  // key buffer {capacity,3}
    int64_t capacity;
    int heap_top;
};

void* orc_hash_create(int64_t capacity) {
    auto* h = new OrcHashMap();
    h->capacity = capacity;
    h->keys.assign((size_t)capacity * 3, 0);
    h->heap_top = 0;
    return h;
}
void orc_hash_destroy(void* hp) { delete (OrcHashMap*)hp; }
int64_t orc_hash_size(void* hp) { return (int64_t)((OrcHashMap*)hp)->map.size(); }
int64_t orc_hash_capacity(void* hp) { return ((OrcHashMap*)hp)->capacity; }
const int* orc_hash_key_buffer(void* hp) { return ((OrcHashMap*)hp)->keys.data(); }

// Returns 0 on success, 1 if capacity would be exceeded (the caller -- the
// test harness -- is expected to size the map; the reference would Reserve()).
int orc_hash_activate(void* hp, const int* keys, int64_t n, int* buf_indices,
                      uint8_t* masks) {
    auto* h = (OrcHashMap*)hp;
    for (int64_t i = 0; i < n; ++i) {
        buf_indices[i] = 0;
        masks[i] = 0;
        Coord3i k{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]};
        auto res = h->map.insert({k, 0});
        if (res.second) {
            if (h->heap_top >= h->capacity) return 1;
            int buf_index = h->heap_top++;
            h->keys[(size_t)buf_index * 3 + 0] = k.x;
            h->keys[(size_t)buf_index * 3 + 1] = k.y;
            h->keys[(size_t)buf_index * 3 + 2] = k.z;
            res.first->second = buf_index;
            buf_indices[i] = buf_index;
            masks[i] = 1;
        }
    }
    return 0;
}

void orc_hash_find(void* hp, const int* keys, int64_t n, int* buf_indices,
                   uint8_t* masks) {
    auto* h = (OrcHashMap*)hp;
    for (int64_t i = 0; i < n; ++i) {
        Coord3i k{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]};
        auto it = h->map.find(k);
        if (it == h->map.end()) {
            buf_indices[i] = 0;
            masks[i] = 0;
        } else {
            buf_indices[i] = it->second;
            masks[i] = 1;
        }
    }
}

int64_t orc_hash_active_indices(void* hp, int* out) {
    auto* h = (OrcHashMap*)hp;
    int64_t i = 0;
    std::vector<int> v;
    for (auto& kv : h->map) v.push_back(kv.second);
    std::sort(v.begin(), v.end());
    for (int b : v) out[i++] = b;
    return i;
}

// --- integrate --------------------------------------------------------------
void orc_integrate(const void* depth, int depth_rows, int depth_cols,
                   const void* color, int color_rows, int color_cols,
                   int input_is_f32, const int* indices, int64_t n_indices,
                   const int* block_keys, float* tsdf, void* weight,
                   void* color_buf, int grid_is_f32,
                   const double* depth_intrinsic, const double* color_intrinsic,
                   const double* extrinsics, int resolution, float voxel_size,
                   float sdf_trunc, float depth_scale, float depth_max) {
#define CALL(DT, CT, WT, VT)                                                  \
    IntegrateImpl<DT, CT, WT, VT>(                                            \
            (const DT*)depth, depth_rows, depth_cols, (const CT*)color,       \
            color_rows, color_cols, indices, n_indices, block_keys, tsdf,     \
            (WT*)weight, (VT*)color_buf, depth_intrinsic, color_intrinsic,    \
            extrinsics, resolution, voxel_size, sdf_trunc, depth_scale,       \
            depth_max)
    // Instantiations mirror VoxelBlockGridCPU.cpp:212-218.
    if (!input_is_f32 && !grid_is_f32) CALL(uint16_t, uint8_t, uint16_t, uint16_t);
    else if (!input_is_f32 && grid_is_f32) CALL(uint16_t, uint8_t, float, float);
    else if (input_is_f32 && !grid_is_f32) CALL(float, float, uint16_t, uint16_t);
    else CALL(float, float, float, float);
#undef CALL
}

int orc_estimate_range(const int* block_keys, int64_t n_blocks,
                       float* range_minmax_map, const double* intrinsics,
                       const double* extrinsics, int h, int w, int down_factor,
                       int64_t block_resolution, float voxel_size,
                       float depth_min, float depth_max, int frag_buffer_size) {
    return EstimateRangeImpl(block_keys, n_blocks, range_minmax_map, intrinsics,
                             extrinsics, h, w, down_factor, block_resolution,
                             voxel_size, depth_min, depth_max,
                             frag_buffer_size);
}

void orc_raycast(void* hp, const float* tsdf, const void* weight,
                 const void* color_buf, int grid_is_f32, const float* range_map,
                 float* out_depth, float* out_vertex, float* out_color,
                 float* out_normal, int64_t* out_index, uint8_t* out_mask,
                 float* out_ratio, float* out_ratio_dx, float* out_ratio_dy,
                 float* out_ratio_dz, const double* intrinsic,
                 const double* extrinsics, int h, int w, int block_resolution,
                 float voxel_size, float depth_scale, float depth_min,
                 float depth_max, float weight_threshold,
                 float trunc_voxel_multiplier, int range_map_down_factor) {
    auto* hm = (OrcHashMap*)hp;
    RayCastOutputs out{out_depth,    out_vertex,   out_color,    out_normal,
                       out_index,    out_mask,     out_ratio,    out_ratio_dx,
                       out_ratio_dy, out_ratio_dz};
    // Instantiations mirror VoxelBlockGridCPU.cpp:231-232.
    if (grid_is_f32)
        RayCastImpl<float, float>(hm->map, tsdf, (const float*)weight,
                                  (const float*)color_buf, range_map, out,
                                  intrinsic, extrinsics, h, w, block_resolution,
                                  voxel_size, depth_scale, depth_min, depth_max,
                                  weight_threshold, trunc_voxel_multiplier,
                                  range_map_down_factor);
    else
        RayCastImpl<uint16_t, uint16_t>(
                hm->map, tsdf, (const uint16_t*)weight,
                (const uint16_t*)color_buf, range_map, out, intrinsic,
                extrinsics, h, w, block_resolution, voxel_size, depth_scale,
                depth_min, depth_max, weight_threshold, trunc_voxel_multiplier,
                range_map_down_factor);
}

// BufferRadiusNeighbors, VoxelBlockGrid.cpp:22-51: {27, n} tables.
void orc_buffer_radius_neighbors(void* hp, const int* active_buf_indices,
                                 int64_t n, int* nb_indices, uint8_t* nb_masks) {
    auto* h = (OrcHashMap*)hp;
    for (int nb = 0; nb < 27; ++nb) {
        int dz = nb / 9, dy = (nb % 9) / 3, dx = nb % 3;
        for (int64_t i = 0; i < n; ++i) {
            const int* k = h->keys.data() + 3 * (size_t)active_buf_indices[i];
            Coord3i q{k[0] + dx - 1, k[1] + dy - 1, k[2] + dz - 1};
            auto it = h->map.find(q);
            nb_indices[(int64_t)nb * n + i] = it == h->map.end() ? 0 : it->second;
            nb_masks[(int64_t)nb * n + i] = it == h->map.end() ? 0 : 1;
        }
    }
}

// points == NULL: counting pass (valid_size < 0 in the reference).
int64_t orc_extract_point_cloud(const int* indices, const int* nb_indices,
                                const uint8_t* nb_masks, const int* block_keys,
                                const float* tsdf, const void* weight,
                                const void* color_buf, int grid_is_f32,
                                int64_t n_blocks, int resolution,
                                float voxel_size, float weight_threshold,
                                float* points, float* normals, float* colors,
                                int64_t valid_size) {
    if (grid_is_f32)
        return ExtractPointCloudImpl<float, float>(
                indices, nb_indices, nb_masks, block_keys, tsdf,
                (const float*)weight, (const float*)color_buf, n_blocks,
                resolution, voxel_size, weight_threshold, points, normals,
                colors, valid_size);
    return ExtractPointCloudImpl<uint16_t, uint16_t>(
            indices, nb_indices, nb_masks, block_keys, tsdf,
            (const uint16_t*)weight, (const uint16_t*)color_buf, n_blocks,
            resolution, voxel_size, weight_threshold, points, normals, colors,
            valid_size);
}

int64_t orc_unproject(const void* depth, int depth_is_f32, int rows, int cols,
                      const float* colors_f32,
                      float* points, float* colors, const double* intrinsics,
                      const double* extrinsics, float depth_scale,
                      float depth_max, int64_t stride) {
    if (depth_is_f32)
        return UnprojectImpl<float>((const float*)depth, rows, cols,
                                    colors_f32, points, colors, intrinsics,
                                    extrinsics, depth_scale, depth_max, stride);
    return UnprojectImpl<uint16_t>((const uint16_t*)depth, rows, cols,
                                   colors_f32, points, colors,
                                   intrinsics, extrinsics, depth_scale,
                                   depth_max, stride);
}

}  // extern "C"
