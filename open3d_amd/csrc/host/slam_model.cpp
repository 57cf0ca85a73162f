// Host-side slam::Model for the MI355X backend: the dense-SLAM volumetric model
// of cpp/open3d/t/pipelines/slam/Model.{h,cpp} (ctor :23-38,
// SynthesizeModelFrame :40-68, TrackFrameToModel :70-92, Integrate :94-108,
// ExtractPointCloud :110-113) on top of the VoxelBlockGrid and RGB-D odometry
// drivers of this library. The control flow is the reference's; what changes is
// below it (fused touch / integrate / ray-cast / odometry kernels).

#include <cmath>
#include <cstring>

#include "../common.h"
#include "o3d_mi355x_host.h"

extern "C" int o3dmi_vbg_export_last_frame_blocks(o3dmi_vbg_t* g,
                                                  int32_t* out_keys_dev,
                                                  int64_t out_capacity,
                                                  int32_t* out_count_dev,
                                                  o3dmi_stream_t stream);

using namespace o3dmi;

struct o3dmi_slam_model {
    o3dmi_vbg_t* grid = nullptr;
    double T_frame_to_world[16];
    int frame_id = -1;
    // frustum_block_coords_ of the last Integrate.
    int32_t* frustum_coords = nullptr;
    int64_t frustum_capacity = 0;
    // Number of frustum blocks: device-resident (frustum_count_dev) after the
    // fused Integrate path, mirrored on the host only on demand.
    int64_t frustum_count = 0;       // valid when !count_on_device
    int32_t* frustum_count_dev = nullptr;
    bool count_on_device = false;
    bool integrated = false;
};

extern "C" {

int o3dmi_slam_model_create(float voxel_size, int block_resolution,
                            int64_t block_count, const double* T_init,
                            o3dmi_stream_t stream, o3dmi_slam_model_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    auto* m = new o3dmi_slam_model();
    const char* names[3] = {"tsdf", "weight", "color"};
    const int dtypes[3] = {O3DMI_F32, O3DMI_U16, O3DMI_U16};
    const int channels[3] = {1, 1, 3};
    int st = o3dmi_vbg_create(3, names, dtypes, channels, voxel_size,
                              block_resolution, block_count, stream, &m->grid);
    if (st) {
        delete m;
        return st;
    }
    // a tracking pipeline is being set up: its kernels are loaded now, not by
    // its first frame (best effort)
    (void)o3dmi_preload();
    if (T_init)
        std::memcpy(m->T_frame_to_world, T_init, sizeof(m->T_frame_to_world));
    else
        for (int i = 0; i < 16; ++i)
            m->T_frame_to_world[i] = (i % 5 == 0) ? 1.0 : 0.0;
    *out = m;
    return O3DMI_OK;
}

int o3dmi_slam_model_destroy(o3dmi_slam_model_t* m) {
    if (!m) return O3DMI_OK;
    (void)hipDeviceSynchronize();
    o3dmi_vbg_destroy(m->grid);
    (void)hipFree(m->frustum_coords);
    (void)hipFree(m->frustum_count_dev);
    delete m;
    return O3DMI_OK;
}

o3dmi_vbg_t* o3dmi_slam_model_voxel_grid(o3dmi_slam_model_t* m) {
    return m ? m->grid : nullptr;
}

int o3dmi_slam_model_get_current_frame_pose(const o3dmi_slam_model_t* m,
                                            double* T_frame_to_world) {
    O3DMI_REQUIRE(m && T_frame_to_world, "null argument");
    std::memcpy(T_frame_to_world, m->T_frame_to_world,
                sizeof(m->T_frame_to_world));
    return O3DMI_OK;
}

int o3dmi_slam_model_update_frame_pose(o3dmi_slam_model_t* m, int frame_id,
                                       const double* T_frame_to_world) {
    O3DMI_REQUIRE(m && T_frame_to_world, "null argument");
    // "Skipped {} frames in update T!" is a warning in the reference.
    m->frame_id = frame_id;
    std::memcpy(m->T_frame_to_world, T_frame_to_world,
                sizeof(m->T_frame_to_world));
    return O3DMI_OK;
}

int o3dmi_slam_model_frame_id(const o3dmi_slam_model_t* m) {
    return m ? m->frame_id : -1;
}

int o3dmi_slam_model_synthesize_model_frame(
        o3dmi_slam_model_t* m, const double* intrinsics, int width, int height,
        float depth_scale, float depth_min, float depth_max,
        float trunc_voxel_multiplier, float weight_threshold,
        float* depth_out_dev, float* color_out_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(m && intrinsics && depth_out_dev, "null argument");
    O3DMI_REQUIRE(width > 0 && height > 0, "empty frame");
    O3DMI_REQUIRE(m->integrated,
                  "SynthesizeModelFrame needs a previous Integrate");
    if (weight_threshold < 0)
        weight_threshold = std::fmin(m->frame_id * 1.0f, 3.0f);  // Model.cpp:45-47
    // (the range map is the grid's own scratch -- as in the reference, where
    // RayCast allocates it -- and the ray cast leaves it clean: no clearing
    // launch per frame)
    const int down = 8;
    double extrinsic[16];
    InverseTransformation(m->T_frame_to_world, extrinsic);
    return o3dmi_vbg_ray_cast_dev(
            m->grid, m->frustum_coords,
            m->count_on_device ? m->frustum_capacity : m->frustum_count,
            m->count_on_device ? m->frustum_count_dev : nullptr, intrinsics,
            extrinsic, width, height, /*range_map=*/nullptr, depth_out_dev,
            nullptr, color_out_dev,
            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
            depth_scale, depth_min, depth_max, weight_threshold,
            trunc_voxel_multiplier, down, stream);
}

int o3dmi_slam_model_track_frame_to_model(
        o3dmi_slam_model_t* m, const void* input_depth_dev,
        int input_depth_dtype, const void* input_color_dev,
        int input_color_dtype, const float* raycast_depth_dev,
        const float* raycast_color_dev, int rows, int cols,
        const double* intrinsics, float depth_scale, float depth_max,
        float depth_diff, int method, int n_levels,
        const o3dmi_odometry_criteria_t* criteria,
        o3dmi_odometry_result_t* result, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(m != nullptr, "model is null");
    const o3dmi_odometry_criteria_t defaults[3] = {
            {6, 1e-6, 1e-6}, {3, 1e-6, 1e-6}, {1, 1e-6, 1e-6}};
    if (!criteria) {
        criteria = defaults;
        n_levels = 3;
    }
    // OdometryLossParams(depth_diff): huber deltas keep their defaults
    // (RGBDOdometry.h:99-101).
    return o3dmi_rgbd_odometry_multiscale(
            input_depth_dev, input_color_dev, raycast_depth_dev,
            raycast_color_dev, input_depth_dtype, input_color_dtype, O3DMI_F32,
            O3DMI_F32, rows, cols, intrinsics, nullptr, depth_scale, depth_max,
            n_levels, criteria, method, depth_diff, 0.05f, 0.1f, result,
            stream);
}

int o3dmi_slam_model_integrate(o3dmi_slam_model_t* m, const void* depth_dev,
                               int depth_dtype, const void* color_dev,
                               int rows, int cols, const double* intrinsics,
                               float depth_scale, float depth_max,
                               float trunc_voxel_multiplier,
                               o3dmi_stream_t stream) {
    O3DMI_REQUIRE(m && depth_dev && intrinsics, "null argument");
    O3DMI_REQUIRE(depth_dtype == O3DMI_U16 || depth_dtype == O3DMI_F32,
                  "depth must be UInt16 or Float32");
    const int64_t cap = (int64_t)(cols / 4) * (rows / 4) * 4;
    O3DMI_REQUIRE(cap > 0, "depth image too small");
    if (m->frustum_capacity < cap) {
        (void)hipFree(m->frustum_coords);
        m->frustum_coords = nullptr;
        O3DMI_HIP_CHECK(hipMalloc((void**)&m->frustum_coords,
                                  sizeof(int32_t) * 3 * (size_t)cap));
        m->frustum_capacity = cap;
    }
    if (!m->frustum_count_dev)
        O3DMI_HIP_CHECK(hipMalloc((void**)&m->frustum_count_dev,
                                  sizeof(int32_t)));
    double extrinsic[16];
    InverseTransformation(m->T_frame_to_world, extrinsic);
    m->frustum_count = 0;
    m->integrated = false;
    // GetUniqueBlockCoordinates + Integrate in the fused form: block touch,
    // activation and the voxel update are issued back to back, the touched
    // block keys (frustum_block_coords_) and their number stay on the device
    // for the next SynthesizeModelFrame. The reference's "No block is touched"
    // error needs that number on the host; here it surfaces when
    // o3dmi_slam_model_frustum_block_count is asked for.
    int st = o3dmi_vbg_integrate_frame(
            m->grid, depth_dev, rows, cols, color_dev, color_dev ? rows : 0,
            color_dev ? cols : 0, depth_dtype, intrinsics, intrinsics, extrinsic,
            depth_scale, depth_max, trunc_voxel_multiplier, stream);
    if (st) return st;
    st = o3dmi_vbg_export_last_frame_blocks(m->grid, m->frustum_coords,
                                            m->frustum_capacity,
                                            m->frustum_count_dev, stream);
    if (st) return st;
    m->count_on_device = true;
    m->integrated = true;
    return O3DMI_OK;
}

int64_t o3dmi_slam_model_frustum_block_count(const o3dmi_slam_model_t* m) {
    if (!m || !m->integrated) return 0;
    if (!m->count_on_device) return m->frustum_count;
    int32_t n = 0;
    if (hipMemcpy(&n, m->frustum_count_dev, sizeof(n), hipMemcpyDeviceToHost) !=
        hipSuccess)
        return 0;
    return n;
}

const int32_t* o3dmi_slam_model_frustum_block_coords(
        const o3dmi_slam_model_t* m) {
    return m ? m->frustum_coords : nullptr;
}

int o3dmi_slam_model_extract_point_cloud(o3dmi_slam_model_t* m,
                                         float weight_threshold,
                                         int64_t capacity, float* points_dev,
                                         float* normals_dev, float* colors_dev,
                                         int64_t* total_out,
                                         o3dmi_stream_t stream) {
    O3DMI_REQUIRE(m != nullptr, "model is null");
    return o3dmi_vbg_extract_point_cloud(m->grid, weight_threshold, capacity,
                                         points_dev, normals_dev, colors_dev,
                                         total_out, stream);
}

}  // extern "C"
