// TEST INFRASTRUCTURE ONLY. Sanitizer driver of the CPU oracle: built by
// `make -C oracle asan` with -fsanitize=address,undefined (the reference's own
// switches: CMakeLists.txt:107-109, cpp/open3d/CMakeLists.txt:60-67) and run
// by tests/test_oracle_sanitizers.py. It walks the hot path once on a small
// analytic scene -- depth touch, hash activate / find, integrate (both grid
// dtypes), estimate range, ray cast, and a two-scale point-to-plane ICP -- so
// that out-of-bounds accesses, use-after-free and undefined arithmetic in the
// checker show up as a non-zero exit.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" {
void orc_set_threads(int n);
int64_t orc_depth_touch(const void* depth, int depth_is_f32, int rows, int cols,
                        const double* intrinsic, const double* extrinsic,
                        int resolution, float voxel_size, float sdf_trunc,
                        float depth_scale, float depth_max, int stride,
                        int* out_coords, int64_t out_capacity);
void* orc_hash_create(int64_t capacity);
void orc_hash_destroy(void* hp);
int64_t orc_hash_size(void* hp);
const int* orc_hash_key_buffer(void* hp);
int orc_hash_activate(void* hp, const int* keys, int64_t n, int* buf_indices,
                      uint8_t* masks);
void orc_hash_find(void* hp, const int* keys, int64_t n, int* buf_indices,
                   uint8_t* masks);
void orc_integrate(const void* depth, int depth_rows, int depth_cols,
                   const void* color, int color_rows, int color_cols,
                   int input_is_f32, const int* indices, int64_t n_indices,
                   const int* block_keys, float* tsdf, void* weight,
                   void* color_buf, int grid_is_f32,
                   const double* depth_intrinsic, const double* color_intrinsic,
                   const double* extrinsics, int resolution, float voxel_size,
                   float sdf_trunc, float depth_scale, float depth_max);
int orc_estimate_range(const int* block_keys, int64_t n_blocks,
                       float* range_minmax_map, const double* intrinsics,
                       const double* extrinsics, int h, int w, int down_factor,
                       int64_t block_resolution, float voxel_size,
                       float depth_min, float depth_max, int frag_buffer_size);
void orc_raycast(void* hp, const float* tsdf, const void* weight,
                 const void* color_buf, int grid_is_f32, const float* range_map,
                 float* out_depth, float* out_vertex, float* out_color,
                 float* out_normal, int64_t* out_index, uint8_t* out_mask,
                 float* out_ratio, float* out_ratio_dx, float* out_ratio_dy,
                 float* out_ratio_dz, const double* intrinsic,
                 const double* extrinsics, int h, int w, int block_resolution,
                 float voxel_size, float depth_scale, float depth_min,
                 float depth_max, float weight_threshold,
                 float trunc_voxel_multiplier, int range_map_down_factor);
typedef void (*icp_callback_t)(int64_t, int64_t, int64_t, double, double,
                               const double*, void*);
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
                       icp_callback_t cb, void* user);
}

int main() {
    orc_set_threads(2);
    const int W = 160, H = 120, res = 16;
    const float voxel = 0.008f, trunc = voxel * 8, ds = 1000.f, dmax = 3.f;
    const double K[9] = {131.25, 0, 79.5, 0, 131.25, 59.5, 0, 0, 1};
    const double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // a tilted wall with a bump, 1 - 1.6 m in front of the camera
    std::vector<uint16_t> depth((size_t)W * H);
    std::vector<uint8_t> color((size_t)W * H * 3);
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            double d = 1.2 + 0.002 * (u - 80) + 0.05 * std::sin(0.1 * v);
            depth[(size_t)v * W + u] = (uint16_t)(d * ds);
            for (int c = 0; c < 3; ++c)
                color[((size_t)v * W + u) * 3 + c] = (uint8_t)((u + 2 * v + 40 * c) & 255);
        }
    // some invalid pixels (zero / beyond depth_max)
    depth[5] = 0;
    depth[77] = 60000;
    std::vector<int> keys(4 * (W / 4) * (H / 4) * 3);
    const int64_t nk = orc_depth_touch(depth.data(), 0, H, W, K, T, res, voxel,
                                       trunc, ds, dmax, 4, keys.data(),
                                       (int64_t)keys.size() / 3);
    if (nk <= 0) return 2;
    const int64_t cap = 2048;
    if (nk > cap) return 3;
    void* h = orc_hash_create(cap);
    std::vector<int> buf((size_t)nk);
    std::vector<uint8_t> mask((size_t)nk);
    if (orc_hash_activate(h, keys.data(), nk, buf.data(), mask.data())) return 4;
    orc_hash_find(h, keys.data(), nk, buf.data(), mask.data());
    const size_t vox = (size_t)cap * res * res * res;
    for (int grid_f32 = 0; grid_f32 < 2; ++grid_f32) {
        std::vector<float> tsdf(vox, 0.f);
        std::vector<uint8_t> weight(vox * (grid_f32 ? 4 : 2), 0);
        std::vector<uint8_t> colbuf(vox * 3 * (grid_f32 ? 4 : 2), 0);
        for (int rep = 0; rep < 2; ++rep)
            orc_integrate(depth.data(), H, W, color.data(), H, W, 0, buf.data(),
                          nk, orc_hash_key_buffer(h), tsdf.data(), weight.data(),
                          colbuf.data(), grid_f32, K, K, T, res, voxel, trunc,
                          ds, dmax);
        std::vector<float> range((size_t)(H / 8) * (W / 8) * 2);
        orc_estimate_range(keys.data(), nk, range.data(), K, T, H, W, 8, res,
                           voxel, 0.1f, dmax, 65536);
        std::vector<float> od((size_t)W * H), ov((size_t)W * H * 3),
                oc((size_t)W * H * 3), on((size_t)W * H * 3),
                r0((size_t)W * H * 8), r1((size_t)W * H * 8),
                r2((size_t)W * H * 8), r3((size_t)W * H * 8);
        std::vector<int64_t> oi((size_t)W * H * 8);
        std::vector<uint8_t> om((size_t)W * H * 8);
        orc_raycast(h, tsdf.data(), weight.data(), colbuf.data(), grid_f32,
                    range.data(), od.data(), ov.data(), oc.data(), on.data(),
                    oi.data(), om.data(), r0.data(), r1.data(), r2.data(),
                    r3.data(), K, T, H, W, res, voxel, ds, 0.1f, dmax, 1.0f,
                    8.0f, 8);
        int hit = 0;
        for (float d : od) hit += d > 0;
        if (hit < W * H / 4) return 5;
    }
    orc_hash_destroy(h);

    // ICP: a wavy sheet and a slightly moved copy
    const int n = 4000;
    std::vector<float> tgt((size_t)n * 3), nrm((size_t)n * 3), src((size_t)n * 3);
    for (int i = 0; i < n; ++i) {
        const double x = (i % 80) * 0.025, y = (i / 80) * 0.04;
        const double z = 0.1 * std::sin(2 * x) + 0.05 * std::cos(3 * y);
        const double gx = 0.2 * std::cos(2 * x), gy = -0.15 * std::sin(3 * y);
        const double l = std::sqrt(gx * gx + gy * gy + 1);
        tgt[3 * i] = (float)x; tgt[3 * i + 1] = (float)y; tgt[3 * i + 2] = (float)z;
        nrm[3 * i] = (float)(-gx / l); nrm[3 * i + 1] = (float)(-gy / l);
        nrm[3 * i + 2] = (float)(1 / l);
        src[3 * i] = (float)(x + 0.01); src[3 * i + 1] = (float)(y - 0.008);
        src[3 * i + 2] = (float)(z + 0.012);
    }
    const double vs[2] = {0.05, -1.0}, md[2] = {0.1, 0.06};
    const int iters[2] = {10, 15};
    const double rf[2] = {1e-6, 1e-6}, rr[2] = {1e-6, 1e-6};
    double To[16], fit = 0, rmse = 0;
    int conv = 0, nit = 0;
    int64_t ncorr = 0;
    std::vector<int64_t> corr((size_t)n);
    for (int acc64 = 0; acc64 < 2; ++acc64) {
        const int st = orc_multiscale_icp(
                src.data(), n, tgt.data(), nrm.data(), n, 0, 2, vs, iters, rf,
                rr, md, T, 0, 1.0, 1.0, acc64, To, &fit, &rmse, &conv, &nit,
                corr.data(), &ncorr, nullptr, nullptr);
        if (st != 0 || !(fit > 0.5)) return 6;
    }
    std::printf("oracle sanitizer walk ok: %lld blocks, icp fitness %.3f rmse "
                "%.4f in %d iterations\n", (long long)nk, fit, rmse, nit);
    return 0;
}
