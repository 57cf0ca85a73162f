"""ctypes binding of the C ABI in include/o3d_mi355x.h.

The HIP library is mandatory: there is no CPU or PyTorch fallback. If
libo3d_mi355x.so is missing or fails to load, importing a compute entry point
raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# O3DMI_LIB: another build of the same library (A/B measurements of two
# builds on one box); the default is the in-tree build.
SO_PATH = os.environ.get("O3DMI_LIB") or os.path.join(
    _HERE, "lib", "libo3d_mi355x.so")

OK = 0
F32, F64, U16, U8, I32, I64 = 0, 1, 2, 3, 4, 5
I8, I16, U32, U64, BOOL = 6, 7, 8, 9, 10

_vp = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int
_f = C.c_float
_d = C.c_double
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes). Kept in one table so that tests can check that
# every symbol the header declares is exported.
PROTOTYPES = {
    "o3dmi_abi_version": (_i32, []),
    "o3dmi_status_string": (C.c_char_p, [_i32]),
    "o3dmi_last_error": (C.c_char_p, []),
    "o3dmi_device_info": (_i32, [C.c_char_p, C.c_size_t, C.POINTER(_i32),
                                 C.POINTER(_i64)]),
    "o3dmi_release_cached_memory": (_i32, []),
    "o3dmi_hash_create": (_i32, [_i64, _i32, C.POINTER(_i64), _vp,
                                 C.POINTER(_vp)]),
    "o3dmi_hash_destroy": (_i32, [_vp]),
    "o3dmi_hash_clear": (_i32, [_vp, _vp]),
    "o3dmi_hash_activate": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "o3dmi_hash_insert": (_i32, [_vp, _vp, C.POINTER(_vp), _i64, _vp, _vp,
                                 _vp]),
    "o3dmi_hash_find": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "o3dmi_hash_erase": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "o3dmi_hash_size": (_i32, [_vp, _vp, C.POINTER(_i64)]),
    "o3dmi_hash_capacity": (_i64, [_vp]),
    "o3dmi_hash_bucket_count": (_i64, [_vp]),
    "o3dmi_hash_active_indices": (_i32, [_vp, _vp, _vp, C.POINTER(_i64)]),
    "o3dmi_hash_reserve": (_i32, [_vp, _i64, _vp]),
    "o3dmi_hash_set_ownership": (_i32, [_vp, _i32, _i32]),
    "o3dmi_block_owner": (_i32, [C.POINTER(_i32), _i32]),
    "o3dmi_hash_key_buffer": (_vp, [_vp]),
    "o3dmi_hash_value_buffer": (_vp, [_vp, _i32]),
    "o3dmi_vbg_depth_touch": (_i32, [_vp, _vp, _i32, _i32, _i32, _dp, _dp,
                                     _vp, _i64, _vp, _i32, _f, _f, _f, _f,
                                     _i32, _vp]),
    "o3dmi_vbg_pointcloud_touch": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp,
                                          _i32, _f, _f, _vp]),
    "o3dmi_vbg_voxel_coordinates_and_flattened_indices": (
        _i32, [_vp, _i64, _vp, _i32, _f, _vp, _vp, _vp]),
    "o3dmi_vbg_voxel_indices": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "o3dmi_vbg_voxel_coordinates": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp,
                                           _vp, _vp]),
    "o3dmi_vbg_integrate": (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32,
                                   _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32,
                                   _dp, _dp, _dp, _i32, _f, _f, _f, _f, _vp]),
    "o3dmi_vbg_touch_activate": (_i32, [_vp, _vp, _i32, _i32, _i32, _dp, _dp,
                                        _vp, _i64, _vp, _i32, _f, _f, _f, _f,
                                        _i32, _i32, _vp]),
    "o3dmi_vbg_estimate_range": (_i32, [_vp, _i64, _vp, _dp, _dp, _i32, _i32,
                                        _i32, _i64, _f, _f, _f, _vp]),
    "o3dmi_vbg_estimate_range_dev": (_i32, [_vp, _i64, _vp, _vp, _dp, _dp,
                                            _i32, _i32, _i32, _i64, _f, _f,
                                            _f, _vp]),
    "o3dmi_vbg_raycast": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp] + [_vp] * 10 +
                          [_dp, _dp, _i32, _i32, _i32, _f, _f, _f, _f, _f, _f,
                           _i32, _vp]),
    "o3dmi_vbg_raycast_rows": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp] +
                               [_vp] * 10 +
                               [_dp, _dp, _i32, _i32, _i32, _i32, _i32, _f, _f,
                                _f, _f, _f, _f, _i32, _vp]),
    "o3dmi_unproject": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _dp,
                               _dp, _f, _f, _i64, _vp]),
    "o3dmi_unproject_pair": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _dp,
                                    _vp, _i32, _vp, _vp, _vp, _vp, _dp,
                                    _i32, _i32, _dp, _f, _f, _i64, _vp]),
    "o3dmi_nns_create": (_i32, [_vp, _i64, _i32, _d, _vp, C.POINTER(_vp)]),
    "o3dmi_nns_destroy": (_i32, [_vp]),
    "o3dmi_nns_hybrid_search_k1": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp,
                                          _vp]),
    "o3dmi_nns_hybrid_search": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp,
                                       _vp]),
    "o3dmi_pointcloud_estimate_covariances": (_i32, [_vp, _vp, _vp, _i64, _i32,
                                                     _i32, _vp, _vp]),
    "o3dmi_pointcloud_normals_from_covariances": (_i32, [_vp, _i64, _i32, _vp,
                                                         _i32, _vp]),
    "o3dmi_icp_p2plane_accumulate": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32,
                                            _i32, _d, _d, _vp, _vp]),
    "o3dmi_icp_search_accumulate": (_i32, [_vp, _vp, _vp, _i64, _i32, _d, _d,
                                           _vp, _vp, _vp]),
    "o3dmi_transform_points": (_i32, [_dp, _vp, _i64, _i32, _vp]),
    "o3dmi_transform_normals": (_i32, [_dp, _vp, _i64, _i32, _vp]),
    "o3dmi_decode_and_solve6x6": (_i32, [_dp, _dp, C.POINTER(_f),
                                         C.POINTER(_i32)]),
    "o3dmi_pose_to_transformation": (None, [_dp, _dp]),
    "o3dmi_vbg_extract_points": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i32,
                                        _i32, _f, _f, _vp, _vp, _vp, _i64,
                                        C.POINTER(_i64), _vp]),
    "o3dmi_sort_indices": (_i32, [_vp, _i64, _vp]),
    "o3dmi_image_clip_transform": (_i32, [_vp, _i32, _i32, _i32, _f, _f, _f,
                                          _f, _vp, _vp]),
    "o3dmi_image_pyrdown_depth": (_i32, [_vp, _i32, _i32, _f, _f, _vp, _vp]),
    "o3dmi_image_create_vertex_map": (_i32, [_vp, _i32, _i32, _dp, _f, _vp,
                                             _vp]),
    "o3dmi_image_create_normal_map": (_i32, [_vp, _i32, _i32, _f, _vp, _vp]),
    "o3dmi_image_to_float": (_i32, [_vp, _i32, _i64, _d, _d, _vp, _vp]),
    "o3dmi_image_rgb_to_gray": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "o3dmi_image_rgb_to_intensity": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "o3dmi_image_filter_bilateral": (_i32, [_vp, _i32, _i32, _i32, _f, _f,
                                            _vp, _vp]),
    "o3dmi_image_filter_gaussian": (_i32, [_vp, _i32, _i32, _i32, _f, _vp,
                                           _vp]),
    "o3dmi_image_filter_sobel": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "o3dmi_image_resize_half_nearest": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "o3dmi_image_pyrdown": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "o3dmi_odometry_p2plane_level": (_i32, [_vp, _vp, _i32, _i32, _dp, _vp,
                                            _vp, _vp, _vp, _vp, _f, _vp]),
    "o3dmi_image_clip_transform_pair": (_i32, [_vp, _i32, _vp, _i32, _i32,
                                               _i32, _f, _f, _f, _f, _vp, _vp,
                                               _vp]),
    "o3dmi_odometry_sums_scratch_doubles": (_i32, []),
    "o3dmi_odometry_sums": (_i32, [_i32, _i32, _i32] + [_vp] * 11 +
                            [_dp, _dp, _f, _f, _f, _vp, _vp, _vp]),
    "o3dmi_odometry_information": (_i32, [_i32, _i32, _vp, _vp, _dp, _dp, _f,
                                          _dp, _vp]),
}

# include/o3d_mi355x_host.h
class IcpCriteria(C.Structure):
    _fields_ = [("relative_fitness", _d), ("relative_rmse", _d),
                ("max_iteration", _i32)]


class RegistrationResultC(C.Structure):
    _fields_ = [("transformation", _d * 16), ("inlier_rmse", _d),
                ("fitness", _d), ("converged", _i32),
                ("num_iterations", _i32), ("num_correspondences", _i64)]


class OdometryCriteriaC(C.Structure):
    _fields_ = [("max_iteration", _i32), ("relative_rmse", _d),
                ("relative_fitness", _d)]


class OdometryResultC(C.Structure):
    _fields_ = [("transformation", _d * 16), ("inlier_rmse", _d),
                ("fitness", _d), ("num_iterations", _i32)]


class IcpAttributes(C.Structure):
    _fields_ = [("source_normals", _vp), ("source_colors", _vp),
                ("target_colors", _vp), ("target_color_gradients", _vp),
                ("lambda_geometric", _d)]


ICP_CALLBACK = C.CFUNCTYPE(None, _i64, _i64, _i64, _d, _d, _dp, _vp)
ALLREDUCE_SUM = C.CFUNCTYPE(_i32, _dp, _i32, _vp)
ALLREDUCE_DEVICE = C.CFUNCTYPE(_i32, _vp, _i32, _vp, _vp)

PROTOTYPES.update({
    "o3dmi_set_device_allreduce": (_i32, [ALLREDUCE_DEVICE, _vp]),
    "o3dmi_registration_multiscale_icp": (
        _i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _dp,
               C.POINTER(IcpCriteria), _dp, _dp, _i32, _d, _d, ICP_CALLBACK,
               _vp, ALLREDUCE_SUM, _vp, _vp, C.POINTER(RegistrationResultC),
               _vp]),
    "o3dmi_registration_multiscale_icp_ex": (
        _i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _dp,
               C.POINTER(IcpCriteria), _dp, _dp, _i32,
               C.POINTER(IcpAttributes), _i32, _d, _d, ICP_CALLBACK, _vp,
               ALLREDUCE_SUM, _vp, _vp, C.POINTER(RegistrationResultC), _vp]),
    "o3dmi_icp_colored_accumulate": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _i64, _i32, _d, _i32, _d, _d, _vp,
                                            _vp]),
    "o3dmi_pointcloud_color_gradients_from_neighbors": (
        _i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "o3dmi_pointcloud_estimate_color_gradients": (
        _i32, [_vp, _vp, _vp, _i64, _i32, _i32, _d, _vp, _vp]),
    "o3dmi_nns_radius_count": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "o3dmi_nns_radius_search": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "o3dmi_nns_radius_covariances": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "o3dmi_nns_knn_search": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _vp,
                                    _vp, _vp]),
    "o3dmi_registration_compute_rmse": (
        _i32, [_i32, _vp, _i64, _vp, _vp, _i32, C.POINTER(IcpAttributes), _vp,
               C.POINTER(_d), _vp]),
    "o3dmi_registration_evaluate": (
        _i32, [_vp, _i64, _vp, _i64, _i32, _d, _dp, _vp,
               C.POINTER(RegistrationResultC), _vp]),
    "o3dmi_registration_information_matrix": (
        _i32, [_vp, _i64, _vp, _i64, _i32, _d, _dp, _dp, _vp]),
    "o3dmi_icp_information_accumulate": (_i32, [_vp, _vp, _i64, _i32, _vp,
                                                _vp]),
    "o3dmi_icp_symmetric_accumulate": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64,
                                              _i32, _dp, _dp, _i32, _d, _d,
                                              _vp, _vp]),
    "o3dmi_symmetric_pose_to_transformation": (None, [_dp, _dp, _dp, _dp]),
    "o3dmi_icp_p2point_accumulate": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp,
                                            _vp]),
    "o3dmi_icp_search_accumulate_p2point": (_i32, [_vp, _vp, _i64, _vp, _vp,
                                                   _vp]),
    "o3dmi_compute_rt_p2point": (_i32, [_dp, _dp, _dp]),
    "o3dmi_pointcloud_estimate_normals": (_i32, [_vp, _i64, _i32, _i32, _d,
                                                 _vp, _i32, _vp]),
    "o3dmi_voxel_down_sample": (_i32, [_vp, _vp, _i64, _i32, _d, _vp, _vp,
                                       C.POINTER(_i64), _vp]),
    "o3dmi_vbg_to_device": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "o3dmi_hash_to_device": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "o3dmi_vbg_create": (_i32, [_i32, C.POINTER(C.c_char_p), C.POINTER(_i32),
                                C.POINTER(_i32), _f, _i64, _i64, _vp,
                                C.POINTER(_vp)]),
    "o3dmi_vbg_destroy": (_i32, [_vp]),
    "o3dmi_vbg_hashmap": (_vp, [_vp]),
    "o3dmi_vbg_set_block_ownership": (_i32, [_vp, _i32, _i32]),
    "o3dmi_vbg_attribute": (_vp, [_vp, C.c_char_p, C.POINTER(_i32),
                                  C.POINTER(_i32)]),
    "o3dmi_vbg_get_unique_block_coordinates": (
        _i32, [_vp, _vp, _i32, _i32, _i32, _dp, _dp, _f, _f, _f, _vp,
               C.POINTER(_i64), _vp]),
    "o3dmi_vbg_get_unique_block_coordinates_pcd": (
        _i32, [_vp, _vp, _i64, _f, _vp, _i64, C.POINTER(_i64), _vp]),
    "o3dmi_vbg_get_voxel_indices": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "o3dmi_vbg_get_voxel_coordinates": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "o3dmi_vbg_get_voxel_coordinates_and_flattened_indices": (
        _i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "o3dmi_vbg_integrate_blocks": (
        _i32, [_vp, _vp, _i64, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _dp,
               _dp, _dp, _f, _f, _f, _vp]),
    "o3dmi_vbg_integrate_frame": (
        _i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _dp, _dp, _dp, _f,
               _f, _f, _vp]),
    "o3dmi_vbg_integrate_frames": (
        _i32, [_vp, _i32, C.POINTER(_vp), _i32, _i32, C.POINTER(_vp), _i32,
               _i32, _i32, _dp, _dp, _dp, _f, _f, _f, _i32, _vp]),
    "o3dmi_vbg_slice_chunk_frames": (_i32, [_i32]),
    "o3dmi_vbg_set_slice_capacity": (_i32, [_vp, _i32, _i32]),
    "o3dmi_vbg_slice_segment_bytes": (_i64, [_vp]),
    "o3dmi_vbg_touch_slice": (
        _i32, [_vp, _i32, C.POINTER(_vp), _i32, _i32, _dp, _dp, _f, _f, _f,
               _i32, _i32, _i32, _vp, _vp]),
    "o3dmi_vbg_integrate_frames_sliced": (
        _i32, [_vp, _i32, C.POINTER(_vp), _i32, _i32, C.POINTER(_vp), _i32,
               _i32, _i32, _dp, _dp, _dp, _f, _f, _f, _i32, C.POINTER(_vp),
               _vp]),
    "o3dmi_vbg_sliced_stats": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64),
                                      C.POINTER(_i32), C.POINTER(_i32)]),
    "o3dmi_vbg_profile_begin": (_i32, [_vp, _i32, _i32]),
    "o3dmi_vbg_profile_end": (_i32, [_vp, _vp, C.POINTER(_d), C.POINTER(_i64),
                                     C.POINTER(_i64), C.POINTER(_i64)]),
    "o3dmi_rgbd_odometry_multiscale": (
        _i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _dp,
               _dp, _f, _f, _i32, C.POINTER(OdometryCriteriaC), _i32, _f, _f,
               _f,
               C.POINTER(OdometryResultC), _vp]),
    "o3dmi_rgbd_odometry_information_matrix": (
        _i32, [_vp, _vp, _i32, _i32, _i32, _dp, _dp, _f, _f, _f, _dp, _vp]),
    "o3dmi_vbg_last_frame_block_coordinates": (_i32, [_vp, _vp, _i64, _vp,
                                                       _vp]),
    "o3dmi_vbg_ray_cast_dev": (
        _i32, [_vp, _vp, _i64, _vp, _dp, _dp, _i32, _i32, _vp] + [_vp] * 10 +
        [_f, _f, _f, _f, _f, _i32, _vp]),
    "o3dmi_vbg_extract_point_cloud": (_i32, [_vp, _f, _i64, _vp, _vp, _vp,
                                             C.POINTER(_i64), _vp]),
    "o3dmi_slam_model_create": (_i32, [_f, _i32, _i64, _dp, _vp,
                                       C.POINTER(_vp)]),
    "o3dmi_slam_model_destroy": (_i32, [_vp]),
    "o3dmi_slam_model_voxel_grid": (_vp, [_vp]),
    "o3dmi_slam_model_get_current_frame_pose": (_i32, [_vp, _dp]),
    "o3dmi_slam_model_update_frame_pose": (_i32, [_vp, _i32, _dp]),
    "o3dmi_slam_model_frame_id": (_i32, [_vp]),
    "o3dmi_slam_model_synthesize_model_frame": (
        _i32, [_vp, _dp, _i32, _i32, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "o3dmi_slam_model_track_frame_to_model": (
        _i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _dp, _f, _f,
               _f, _i32, _i32, C.POINTER(OdometryCriteriaC),
               C.POINTER(OdometryResultC), _vp]),
    "o3dmi_slam_model_integrate": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32,
                                          _dp, _f, _f, _f, _vp]),
    "o3dmi_slam_model_frustum_block_count": (_i64, [_vp]),
    "o3dmi_slam_model_frustum_block_coords": (_vp, [_vp]),
    "o3dmi_slam_model_extract_point_cloud": (_i32, [_vp, _f, _i64, _vp, _vp,
                                                    _vp, C.POINTER(_i64),
                                                    _vp]),
    "o3dmi_npz_create": (_i32, [C.POINTER(_vp)]),
    "o3dmi_npz_destroy": (_i32, [_vp]),
    "o3dmi_npz_add": (_i32, [_vp, C.c_char_p, _i32, _i32, C.POINTER(_i64),
                             _vp]),
    "o3dmi_npz_count": (_i32, [_vp]),
    "o3dmi_npz_name": (C.c_char_p, [_vp, _i32]),
    "o3dmi_npz_get": (_i32, [_vp, C.c_char_p, C.POINTER(_i32),
                             C.POINTER(_i32), C.POINTER(_i64),
                             C.POINTER(_vp)]),
    "o3dmi_npz_write": (_i32, [_vp, C.c_char_p]),
    "o3dmi_npz_read": (_i32, [C.c_char_p, C.POINTER(_vp)]),
    "o3dmi_vbg_export_blocks": (_i32, [_vp, _i64, _vp, C.POINTER(_vp),
                                       C.POINTER(_i64), _vp]),
    "o3dmi_vbg_merge_blocks": (_i32, [_vp, _vp, C.POINTER(_vp), _i64, _vp]),
    "o3dmi_vbg_save": (_i32, [_vp, C.c_char_p, _vp]),
    "o3dmi_vbg_load": (_i32, [C.c_char_p, _vp, C.POINTER(_vp)]),
    "o3dmi_vbg_attribute_count": (_i32, [_vp]),
    "o3dmi_vbg_attribute_name": (C.c_char_p, [_vp, _i32]),
    "o3dmi_vbg_voxel_size": (_f, [_vp]),
    "o3dmi_vbg_block_resolution": (_i64, [_vp]),
    "o3dmi_vbg_ray_cast": (
        _i32, [_vp, _vp, _i64, _dp, _dp, _i32, _i32, _vp] + [_vp] * 10 +
        [_f, _f, _f, _f, _f, _i32, _vp]),
    "o3dmi_vbg_ray_cast_sharded": (
        _i32, [_vp, _vp, _i64, _dp, _dp, _i32, _i32, _vp] + [_vp] * 4 +
        [_f, _f, _f, _f, _f, _i32, _vp]),
    "o3dmi_vbg_profile_distinct_blocks": (_i64, [_vp]),
    "o3dmi_vbg_division_forms": (_i32, [_f, _f, _i32]),
    "o3dmi_vbg_profile_launches": (_i64, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "o3dmi_rccl_available": (_i32, []),
    "o3dmi_rccl_unique_id": (_i32, [_vp]),
    "o3dmi_comm_create_rccl": (_i32, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "o3dmi_comm_adopt_rccl": (_i32, [_vp, C.POINTER(_vp)]),
    "o3dmi_comm_create_custom": (_i32, [_vp, _vp, _i32, _i32,
                                        C.POINTER(_vp)]),
    "o3dmi_comm_destroy": (_i32, [_vp]),
    "o3dmi_comm_rank": (_i32, [_vp]),
    "o3dmi_preload": (_i32, []),
    "o3dmi_comm_world": (_i32, [_vp]),
    "o3dmi_comm_rccl_ranks": (_i32, [_vp]),
    "o3dmi_set_comm": (_i32, [_vp]),
    "o3dmi_set_rccl_comm": (_i32, [_vp]),
    "o3dmi_set_icp_level_sharding": (_i32, [_i32]),
    "o3dmi_registration_set_device_counts": (_i32, [_vp, _vp]),
    "o3dmi_comm_allreduce_sum_f64": (_i32, [_vp, _vp, _i64, _vp]),
    "o3dmi_comm_allgather": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "o3dmi_comm_alltoallv": (_i32, [_vp, _vp, C.POINTER(_i64),
                                    C.POINTER(_i64), _vp, C.POINTER(_i64),
                                    C.POINTER(_i64), _vp]),
    "o3dmi_vbg_merge_frame_sharded": (_i32, [_vp, _vp, _vp]),
    "o3dmi_vbg_allgather_owned_blocks": (_i32, [_vp, _vp, _vp]),
})

# o3dmi_transport_t (include/o3d_mi355x_host.h): the caller-provided collectives
TRANSPORT_ALLREDUCE = C.CFUNCTYPE(_i32, _vp, _vp, _i64, _vp)
TRANSPORT_ALLGATHER = C.CFUNCTYPE(_i32, _vp, _vp, _vp, _i64, _vp)
TRANSPORT_ALLTOALLV = C.CFUNCTYPE(_i32, _vp, _vp, C.POINTER(_i64),
                                  C.POINTER(_i64), _vp, C.POINTER(_i64),
                                  C.POINTER(_i64), _vp)


class TransportC(C.Structure):
    _fields_ = [("allreduce_sum_f64", TRANSPORT_ALLREDUCE),
                ("allgather", TRANSPORT_ALLGATHER),
                ("alltoallv", TRANSPORT_ALLTOALLV)]

_lib = None


class O3DMIError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().o3dmi_status_string(status).decode()
        detail = lib().o3dmi_last_error().decode()
        super().__init__("%s failed: %s (%s)" % (where, msg, detail))


def lib():
    """Loads the HIP library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "open3d_amd: %s not found -- the HIP extension is mandatory "
                "(run `python -m open3d_amd.build`); there is no CPU "
                "fallback." % SO_PATH)
        # torch first: it ships its own HIP runtime, and the process must end
        # up with ONE libamdhip64 (the first one loaded wins the soname). With
        # the order reversed the second runtime sees no device.
        import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, where):
    if status != OK:
        raise O3DMIError(status, where)


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def f64p(a):
    """numpy float64 contiguous array -> double*."""
    return a.ctypes.data_as(_dp)
