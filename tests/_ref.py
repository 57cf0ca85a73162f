"""ctypes binding of oracle/_ref/libo3d_ref.so -- TEST INFRASTRUCTURE ONLY.

libo3d_ref.so holds the reference's OWN hot-path kernel files
(VoxelBlockGridCPU.cpp + VoxelBlockGridImpl.h, RegistrationCPU.cpp,
TransformationConverter.cpp, TransformImpl.h, PointCloudCPU.cpp ...) compiled
from /root/reference through the stand-in headers in oracle/ref_shim
(`make -C oracle ref`). It exists only where /root/reference exists (this build
container) or where a prebuilt copy travelled with the tree (the GPU box).
`available()` says whether it can be used; tests skip otherwise.

Function names and argument order mirror tests/_oracle.py so the same inputs
can be run through both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "_ref", "libo3d_ref.so")
_REFERENCE = "/root/reference/cpp/open3d"

_lib = None
c_vp = C.c_void_p


def available():
    if os.path.exists(_SO):
        return True
    if os.path.isdir(_REFERENCE):
        try:
            subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "ref"])
        except Exception:
            return False
        return os.path.exists(_SO)
    return False


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libo3d_ref.so is not available")
        _lib = C.CDLL(_SO)
        L = _lib
        L.ref_last_error.restype = C.c_char_p
        L.ref_depth_touch.restype = C.c_int64
        L.ref_pointcloud_touch.restype = C.c_int64
        L.ref_unproject.restype = C.c_int64
        L.ref_robust_weight.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_vp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _check(st, what):
    if st != 0:
        raise RuntimeError("%s: %s" % (what, lib().ref_last_error().decode()))


def set_threads(n):
    lib().ref_set_threads(int(n))


def set_reduce_schedule(seed, min_chunk=1024):
    """Schedule of the stand-in tbb::parallel_reduce: seed 0 = one sequential
    chunk; otherwise seeded random splits down to <= min_chunk elements."""
    lib().ref_set_reduce_schedule(C.c_ulonglong(int(seed)),
                                  C.c_longlong(int(min_chunk)))


def p2plane_accumulate_address():
    """Address of ref_p2plane_accumulate (for orc.set_p2plane_hook)."""
    return C.cast(lib().ref_p2plane_accumulate, C.c_void_p).value


def depth_touch(depth, K, T, resolution, voxel_size, sdf_trunc, depth_scale,
                depth_max, stride=4):
    depth = np.ascontiguousarray(depth)
    rows, cols = depth.shape[:2]
    cap = (rows // stride) * (cols // stride) * 4 + 16
    out = np.zeros((cap, 3), np.int32)
    K, T = _f64(K), _f64(T)
    n = lib().ref_depth_touch(_p(depth), int(depth.dtype == np.float32), rows,
                              cols, _p(K), _p(T), int(resolution),
                              C.c_float(voxel_size), C.c_float(sdf_trunc),
                              C.c_float(depth_scale), C.c_float(depth_max),
                              int(stride), _p(out), C.c_int64(cap))
    if n < 0:
        raise RuntimeError(lib().ref_last_error().decode())
    return out[:n].copy()


def pointcloud_touch(points, resolution, voxel_size, sdf_trunc):
    points = np.ascontiguousarray(points, dtype=np.float32)
    n = points.shape[0]
    cap = n * 27 + 16
    out = np.zeros((cap, 3), np.int32)
    m = lib().ref_pointcloud_touch(_p(points), C.c_int64(n), int(resolution),
                                   C.c_float(voxel_size), C.c_float(sdf_trunc),
                                   _p(out), C.c_int64(cap))
    if m < 0:
        raise RuntimeError(lib().ref_last_error().decode())
    return out[:m].copy()


def voxel_coords_flat(buf_indices, block_keys, resolution, voxel_size):
    buf_indices = np.ascontiguousarray(buf_indices, dtype=np.int32)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    n = buf_indices.shape[0] * resolution ** 3
    coords = np.zeros((n, 3), np.float32)
    flat = np.zeros(n, np.int64)
    _check(lib().ref_voxel_coords_flat(
        _p(buf_indices), C.c_int64(buf_indices.shape[0]), _p(block_keys),
        C.c_int64(block_keys.shape[0]), int(resolution), C.c_float(voxel_size),
        _p(coords), _p(flat)), "ref_voxel_coords_flat")
    return coords, flat


def integrate(depth, color, indices, block_keys, tsdf, weight, color_buf, K_d,
              K_c, T, resolution, voxel_size, sdf_trunc, depth_scale,
              depth_max):
    """In place on tsdf / weight / color_buf."""
    depth = np.ascontiguousarray(depth)
    input_is_f32 = int(depth.dtype == np.float32)
    grid_is_f32 = int(weight.dtype == np.float32)
    if color is not None and color.size > 0:
        color = np.ascontiguousarray(color)
        crow, ccol = color.shape[:2]
    else:
        color, crow, ccol = None, 0, 0
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    K_d, K_c, T = _f64(K_d), _f64(K_c), _f64(T)
    _check(lib().ref_integrate(
        _p(depth), depth.shape[0], depth.shape[1], _p(color), crow, ccol,
        input_is_f32, _p(indices), C.c_int64(indices.shape[0]), _p(block_keys),
        C.c_int64(tsdf.shape[0]), _p(tsdf), _p(weight), _p(color_buf),
        grid_is_f32, _p(K_d), _p(K_c), _p(T), int(resolution),
        C.c_float(voxel_size), C.c_float(sdf_trunc), C.c_float(depth_scale),
        C.c_float(depth_max)), "ref_integrate")


def estimate_range(block_keys, K, T, h, w, down_factor, block_resolution,
                   voxel_size, depth_min, depth_max, frag_buffer_size=0):
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    out = np.zeros((h // down_factor, w // down_factor, 2), np.float32)
    K, T = _f64(K), _f64(T)
    _check(lib().ref_estimate_range(
        _p(block_keys), C.c_int64(block_keys.shape[0]), _p(out), _p(K), _p(T),
        int(h), int(w), int(down_factor), C.c_int64(block_resolution),
        C.c_float(voxel_size), C.c_float(depth_min), C.c_float(depth_max),
        int(frag_buffer_size)), "ref_estimate_range")
    return out


def raycast(hash_keys, hash_buf_indices, tsdf, weight, color_buf, range_map, K,
            T, h, w, block_resolution, voxel_size, depth_scale, depth_min,
            depth_max, weight_threshold, trunc_voxel_multiplier,
            range_map_down_factor, attrs=("depth", "color")):
    grid_is_f32 = int(weight.dtype == np.float32)
    K, T = _f64(K), _f64(T)
    shapes = {"depth": (1, np.float32), "vertex": (3, np.float32),
              "color": (3, np.float32), "normal": (3, np.float32),
              "index": (8, np.int64), "mask": (8, np.uint8),
              "interp_ratio": (8, np.float32),
              "interp_ratio_dx": (8, np.float32),
              "interp_ratio_dy": (8, np.float32),
              "interp_ratio_dz": (8, np.float32)}
    out = {}
    for a in attrs:
        c, dt = shapes[a]
        out[a] = np.full((h, w, c), 77, dt)
    g = lambda a: _p(out[a]) if a in out else None
    hash_keys = np.ascontiguousarray(hash_keys, dtype=np.int32)
    hash_buf_indices = np.ascontiguousarray(hash_buf_indices, dtype=np.int32)
    range_map = np.ascontiguousarray(range_map, dtype=np.float32)
    _check(lib().ref_raycast(
        _p(hash_keys), _p(hash_buf_indices), C.c_int64(hash_keys.shape[0]),
        C.c_int64(tsdf.shape[0]), _p(tsdf), _p(weight), _p(color_buf),
        grid_is_f32, _p(range_map), g("depth"), g("vertex"), g("color"),
        g("normal"), g("index"), g("mask"), g("interp_ratio"),
        g("interp_ratio_dx"), g("interp_ratio_dy"), g("interp_ratio_dz"),
        _p(K), _p(T), int(h), int(w), int(block_resolution),
        C.c_float(voxel_size), C.c_float(depth_scale), C.c_float(depth_min),
        C.c_float(depth_max), C.c_float(weight_threshold),
        C.c_float(trunc_voxel_multiplier), int(range_map_down_factor)),
        "ref_raycast")
    if "mask" in out:
        out["mask"] = out["mask"].astype(bool)
    return out


def unproject(depth, colors_f32, K, T, depth_scale, depth_max, stride=1):
    depth = np.ascontiguousarray(depth)
    rows, cols = depth.shape[:2]
    n = (rows // stride) * (cols // stride)
    pts = np.zeros((n, 3), np.float32)
    cols_out = None
    if colors_f32 is not None:
        colors_f32 = np.ascontiguousarray(colors_f32, dtype=np.float32)
        cols_out = np.zeros((n, 3), np.float32)
    K, T = _f64(K), _f64(T)
    m = lib().ref_unproject(_p(depth), int(depth.dtype == np.float32), rows,
                            cols, _p(colors_f32), _p(pts), _p(cols_out), _p(K),
                            _p(T), C.c_float(depth_scale),
                            C.c_float(depth_max), C.c_int64(stride))
    if m < 0:
        raise RuntimeError(lib().ref_last_error().decode())
    if cols_out is None:
        return pts[:m].copy(), None
    return pts[:m].copy(), cols_out[:m].copy()


def robust_weight(method, scaling, shape, residual, f64=False):
    return float(lib().ref_robust_weight(int(f64), int(method),
                                         C.c_double(scaling),
                                         C.c_double(shape),
                                         C.c_double(residual)))


def p2plane_accumulate(src, tgt, tgt_n, corr, method=0, scaling=1.0,
                       shape=1.0):
    """29 sums accumulated in the point dtype (sequential order)."""
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    tgt_n = np.ascontiguousarray(tgt_n, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    _check(lib().ref_p2plane_accumulate(
        _p(src), _p(tgt), _p(tgt_n), _p(corr), C.c_int64(src.shape[0]),
        int(src.dtype == np.float64), int(method), C.c_double(scaling),
        C.c_double(shape), _p(out)), "ref_p2plane_accumulate")
    return out


def symmetric_accumulate(src, tgt, sn, tn, corr, source_mean, target_mean,
                         method=0, scaling=1.0, shape=1.0):
    """ComputePoseSymmetricKernelCPU: 29 sums in the point dtype."""
    src = np.ascontiguousarray(src)
    dt = src.dtype
    tgt, sn, tn = (np.ascontiguousarray(a, dtype=dt) for a in (tgt, sn, tn))
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    ms = np.ascontiguousarray(source_mean, dtype=np.float64)
    mt = np.ascontiguousarray(target_mean, dtype=np.float64)
    _check(lib().ref_symmetric_accumulate(
        _p(src), _p(tgt), _p(sn), _p(tn), _p(corr), C.c_int64(src.shape[0]),
        int(dt == np.float64), _p(ms), _p(mt), int(method),
        C.c_double(scaling), C.c_double(shape), _p(out)),
        "ref_symmetric_accumulate")
    return out


def colored_accumulate(src, src_c, tgt, tn, tc, tg, corr, lambda_geometric,
                       method=0, scaling=1.0, shape=1.0):
    """ComputePoseColoredICPKernelCPU: 29 sums in the point dtype."""
    src = np.ascontiguousarray(src)
    dt = src.dtype
    src_c, tgt, tn, tc, tg = (np.ascontiguousarray(a, dtype=dt)
                              for a in (src_c, tgt, tn, tc, tg))
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    _check(lib().ref_colored_accumulate(
        _p(src), _p(src_c), _p(tgt), _p(tn), _p(tc), _p(tg), _p(corr),
        C.c_int64(src.shape[0]), int(dt == np.float64),
        C.c_double(lambda_geometric), int(method), C.c_double(scaling),
        C.c_double(shape), _p(out)), "ref_colored_accumulate")
    return out


def information_matrix(tgt, corr):
    """ComputeInformationMatrixCPU: GTG {6,6} float64."""
    tgt = np.ascontiguousarray(tgt)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    G = np.zeros((6, 6), np.float64)
    _check(lib().ref_information_matrix(
        _p(tgt), _p(corr), C.c_int64(corr.shape[0]), C.c_int64(tgt.shape[0]),
        int(tgt.dtype == np.float64), _p(G)), "ref_information_matrix")
    return G


def p2point_sxy(src, tgt, corr):
    """Get3x3SxyLinearSystem (RegistrationCPU.cpp:495-617) in the point dtype:
    (Sxy {3,3}, source_mean {3}, target_mean {3}, inlier_count)."""
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    S = np.zeros((3, 3), np.float64)
    ms = np.zeros(3, np.float64)
    mt = np.zeros(3, np.float64)
    cnt = C.c_int(0)
    _check(lib().ref_p2point_sxy(
        _p(src), _p(tgt), _p(corr), C.c_int64(src.shape[0]),
        int(src.dtype == np.float64), _p(S), _p(ms), _p(mt), C.byref(cnt)),
        "ref_p2point_sxy")
    return S, ms, mt, cnt.value


def compute_pose_p2plane(src, tgt, tgt_n, corr, method=0, scaling=1.0,
                         shape=1.0):
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    tgt_n = np.ascontiguousarray(tgt_n, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    pose = np.zeros(6, np.float64)
    residual, count = C.c_float(0), C.c_int(0)
    _check(lib().ref_compute_pose_p2plane(
        _p(src), _p(tgt), _p(tgt_n), _p(corr), C.c_int64(src.shape[0]),
        C.c_int64(tgt.shape[0]), int(src.dtype == np.float64), int(method),
        C.c_double(scaling), C.c_double(shape), _p(pose), C.byref(residual),
        C.byref(count)), "ref_compute_pose_p2plane")
    return pose, residual.value, count.value


def decode_and_solve6x6(A29):
    A29 = _f64(A29)
    pose = np.zeros(6, np.float64)
    residual, count = C.c_float(0), C.c_int(0)
    st = lib().ref_decode_and_solve6x6(_p(A29), _p(pose), C.byref(residual),
                                       C.byref(count))
    return st, pose, residual.value, count.value


def pose_to_transformation(pose):
    pose = _f64(pose)
    T = np.zeros((4, 4), np.float64)
    _check(lib().ref_pose_to_transformation(_p(pose), _p(T)),
           "ref_pose_to_transformation")
    return T


def transform_points(T, pts):
    T = _f64(T)
    pts = np.array(pts, copy=True, order="C")
    _check(lib().ref_transform_points(_p(T), _p(pts), C.c_int64(pts.shape[0]),
                                      int(pts.dtype == np.float64)),
           "ref_transform_points")
    return pts


def transform_normals(T, nrm):
    T = _f64(T)
    nrm = np.array(nrm, copy=True, order="C")
    _check(lib().ref_transform_normals(_p(T), _p(nrm),
                                       C.c_int64(nrm.shape[0]),
                                       int(nrm.dtype == np.float64)),
           "ref_transform_normals")
    return nrm


# ---------------------------------------------------------------------------
# RGB-D odometry front end: ImageCPU.cpp / RGBDOdometryCPU.cpp bodies
# ---------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def clip_transform(src, scale, min_value, max_value, clip_fill):
    src = np.ascontiguousarray(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols), np.float32)
    _check(lib().ref_clip_transform(_p(src), int(src.dtype == np.float32),
                                    C.c_int64(rows), C.c_int64(cols),
                                    C.c_float(scale), C.c_float(min_value),
                                    C.c_float(max_value), C.c_float(clip_fill),
                                    _p(dst)), "ClipTransformCPU")
    return dst


def pyrdown_depth(src, depth_diff, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows // 2, cols // 2), np.float32)
    _check(lib().ref_pyrdown_depth(_p(src), rows, cols, C.c_float(depth_diff),
                                   C.c_float(invalid_fill), _p(dst)),
           "PyrDownDepthCPU")
    return dst


def create_vertex_map(src, K, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols, 3), np.float32)
    _check(lib().ref_create_vertex_map(_p(src), C.c_int64(rows),
                                       C.c_int64(cols), _p(_f64(K)),
                                       C.c_float(invalid_fill), _p(dst)),
           "CreateVertexMapCPU")
    return dst


def create_normal_map(src, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols, 3), np.float32)
    _check(lib().ref_create_normal_map(_p(src), C.c_int64(rows),
                                       C.c_int64(cols),
                                       C.c_float(invalid_fill), _p(dst)),
           "CreateNormalMapCPU")
    return dst


def image_to_float(src, scale, offset=0.0):
    src = np.ascontiguousarray(src)
    code = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1,
            np.dtype(np.float32): 2}[src.dtype]
    dst = np.empty(src.shape, np.float32)
    _check(lib().ref_image_to_float(_p(src), code, C.c_int64(src.size),
                                    C.c_double(scale), C.c_double(offset),
                                    _p(dst)), "ToCPU")
    return dst


def _opt32(a):
    return None if a is None else _f32(a)


def odometry(method, K, T, source_vertex, target_vertex=None,
             target_normal=None, source_depth=None, target_depth=None,
             source_intensity=None, target_intensity=None,
             target_depth_dx=None, target_depth_dy=None,
             target_intensity_dx=None, target_intensity_dy=None,
             depth_outlier_trunc=0.07, depth_huber_delta=0.05,
             intensity_huber_delta=0.1):
    """Returns (delta pose {6}, residual float, count int, A_1x29 float sums)."""
    sv = _f32(source_vertex)
    rows, cols = sv.shape[:2]
    arrs = [_opt32(a) for a in (source_depth, target_depth, source_intensity,
                                target_intensity, target_depth_dx,
                                target_depth_dy, target_intensity_dx,
                                target_intensity_dy)]
    tv, tn = _opt32(target_vertex), _opt32(target_normal)
    delta = np.zeros(6, np.float64)
    sums = np.zeros(29, np.float64)
    res, cnt = C.c_float(0), C.c_int(0)
    _check(lib().ref_odometry(int(method), rows, cols,
                              *[_p(a) for a in arrs], _p(sv), _p(tv), _p(tn),
                              _p(_f64(K)), _p(_f64(T)),
                              C.c_float(depth_outlier_trunc),
                              C.c_float(depth_huber_delta),
                              C.c_float(intensity_huber_delta), _p(delta),
                              C.byref(res), C.byref(cnt), _p(sums)),
           "ComputeOdometryResultCPU")
    return delta, res.value, cnt.value, sums


def odometry_information(source_vertex, target_vertex, K, T, square_dist_thr):
    sv, tv = _f32(source_vertex), _f32(target_vertex)
    rows, cols = sv.shape[:2]
    out = np.zeros((6, 6), np.float64)
    _check(lib().ref_odometry_information(rows, cols, _p(sv), _p(tv),
                                          _p(_f64(K)), _p(_f64(T)),
                                          C.c_float(square_dist_thr), _p(out)),
           "ComputeOdometryInformationMatrixCPU")
    return out


def extract_point_cloud(indices, nb_indices, nb_masks, block_keys, tsdf,
                        weight, color_buf, resolution, voxel_size,
                        weight_threshold, estimated_number=-1,
                        out_capacity=None):
    """The reference's ExtractPointCloudCPU body: (points, normals,
    colors|None, total_count). Output order is sequential only under
    set_threads(1)."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    nb_indices = np.ascontiguousarray(nb_indices, dtype=np.int32)
    nb_masks = np.ascontiguousarray(nb_masks, dtype=np.uint8)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    n = indices.shape[0]
    capacity = block_keys.shape[0]
    grid_is_f32 = int(weight.dtype == np.float32)
    cap = int(out_capacity if out_capacity is not None else
              (estimated_number if estimated_number >= 0
               else n * resolution ** 3 * 3))
    pts = np.zeros((cap, 3), np.float32)
    nrm = np.zeros((cap, 3), np.float32)
    col = np.zeros((cap, 3), np.float32) if color_buf is not None else None
    L = lib()
    L.ref_extract_point_cloud.restype = C.c_int64
    total = int(L.ref_extract_point_cloud(
        _p(indices), _p(nb_indices), _p(nb_masks), _p(block_keys),
        C.c_int64(capacity), _p(tsdf), _p(weight), _p(color_buf), grid_is_f32,
        C.c_int64(n), int(resolution), C.c_float(voxel_size),
        C.c_float(weight_threshold), _p(pts), _p(nrm), _p(col),
        C.c_int64(cap), int(estimated_number)))
    if total < 0:
        raise RuntimeError("extract_point_cloud: %s"
                           % L.ref_last_error().decode())
    m = min(total, cap)
    return pts[:m], nrm[:m], (None if col is None else col[:m]), total


def estimate_covariances(points, indices, counts):
    points = np.ascontiguousarray(points)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, max_nn = indices.shape
    cov = np.zeros((n, 3, 3), points.dtype)
    _check(lib().ref_estimate_covariances(
        _p(points), _p(indices), _p(counts), C.c_int64(n), int(max_nn),
        int(points.dtype == np.float64), _p(cov)), "estimate_covariances")
    return cov


def svd3x3(A):
    A = np.ascontiguousarray(A)
    U, S, V = np.zeros((3, 3), A.dtype), np.zeros(3, A.dtype), \
        np.zeros((3, 3), A.dtype)
    _check(lib().ref_svd3x3(_p(A), int(A.dtype == np.float64), _p(U), _p(S),
                            _p(V)), "svd3x3")
    return U, S, V


def solve_svd3x3(A, b):
    A = np.ascontiguousarray(A)
    b = np.ascontiguousarray(b, dtype=A.dtype)
    x = np.zeros(3, A.dtype)
    _check(lib().ref_solve_svd3x3(_p(A), _p(b), int(A.dtype == np.float64),
                                  _p(x)), "solve_svd3x3")
    return x


def estimate_color_gradients(points, normals, colors, indices, counts):
    points = np.ascontiguousarray(points)
    dt = points.dtype
    normals = np.ascontiguousarray(normals, dtype=dt)
    colors = np.ascontiguousarray(colors, dtype=dt)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, max_nn = indices.shape
    g = np.zeros((n, 3), dt)
    _check(lib().ref_estimate_color_gradients(
        _p(points), _p(normals), _p(colors), _p(indices), _p(counts),
        C.c_int64(n), int(max_nn), int(dt == np.float64), _p(g)),
        "estimate_color_gradients")
    return g


def normals_from_covariances(cov, normals=None):
    cov = np.ascontiguousarray(cov)
    n = cov.shape[0]
    has = normals is not None
    out = np.ascontiguousarray(normals, dtype=cov.dtype).copy() if has \
        else np.zeros((n, 3), cov.dtype)
    _check(lib().ref_normals_from_covariances(
        _p(cov), C.c_int64(n), int(cov.dtype == np.float64), _p(out),
        int(has)), "normals_from_covariances")
    return out
