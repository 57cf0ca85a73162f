"""ctypes binding of oracle/libo3d_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
this module. The product package (open3d_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libo3d_oracle.so")

_lib = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_fp = C.POINTER(C.c_float)
c_vp = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "libo3d_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_ORACLE_DIR, f) for f in
                ("vbg_oracle.cpp", "icp_oracle.cpp", "odometry_oracle.cpp",
                 "oracle_common.h")]
        if (not os.path.exists(_SO)) or any(
                os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO)
                for s in srcs):
            build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.orc_depth_touch.restype = C.c_int64
        L.orc_pointcloud_touch.restype = C.c_int64
        L.orc_hash_create.restype = c_vp
        L.orc_hash_size.restype = C.c_int64
        L.orc_hash_capacity.restype = C.c_int64
        L.orc_hash_key_buffer.restype = c_vp
        L.orc_hash_active_indices.restype = C.c_int64
        L.orc_unproject.restype = C.c_int64
        L.orc_robust_weight.restype = C.c_double
        L.orc_p2plane_rmse.restype = C.c_double
        L.orc_voxel_down_sample.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_vp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def set_threads(n):
    lib().orc_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().orc_max_threads())


def inverse_transformation(T):
    T = _f64(T)
    out = np.zeros((4, 4), np.float64)
    lib().orc_inverse_transformation(_p(T), _p(out))
    return out


def depth_touch(depth, K, T, resolution, voxel_size, sdf_trunc, depth_scale,
                depth_max, stride=4):
    depth = np.ascontiguousarray(depth)
    is_f32 = int(depth.dtype == np.float32)
    rows, cols = depth.shape[:2]
    cap = (rows // stride) * (cols // stride) * 4 + 16
    out = np.zeros((cap, 3), np.int32)
    K, T = _f64(K), _f64(T)
    n = lib().orc_depth_touch(_p(depth), is_f32, rows, cols, _p(K), _p(T),
                              int(resolution), C.c_float(voxel_size),
                              C.c_float(sdf_trunc), C.c_float(depth_scale),
                              C.c_float(depth_max), int(stride), _p(out),
                              C.c_int64(cap))
    return out[:n].copy()


def pointcloud_touch(points, resolution, voxel_size, sdf_trunc):
    points = np.ascontiguousarray(points, dtype=np.float32)
    n = points.shape[0]
    cap = n * 27 + 16
    out = np.zeros((cap, 3), np.int32)
    m = lib().orc_pointcloud_touch(_p(points), C.c_int64(n), int(resolution),
                                   C.c_float(voxel_size), C.c_float(sdf_trunc),
                                   _p(out), C.c_int64(cap))
    return out[:m].copy()


def voxel_coords_flat(buf_indices, block_keys, resolution, voxel_size):
    """GetVoxelCoordinatesAndFlattenedIndicesCPU (VoxelBlockGridImpl.h:43-92)."""
    buf_indices = np.ascontiguousarray(buf_indices, dtype=np.int32)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    n = buf_indices.shape[0] * resolution ** 3
    coords = np.zeros((n, 3), np.float32)
    flat = np.zeros(n, np.int64)
    lib().orc_voxel_coords_flat(_p(buf_indices), C.c_int64(buf_indices.shape[0]),
                                _p(block_keys), int(resolution),
                                C.c_float(voxel_size), _p(coords), _p(flat))
    return coords, flat


def voxel_indices(buf_indices, resolution):
    """VoxelBlockGrid::GetVoxelIndices (t/geometry/VoxelBlockGrid.cpp:145-178),
    its tensor expressions one for one (numpy Int64)."""
    buf_indices = np.asarray(buf_indices)
    n_blocks = buf_indices.shape[0]
    r, r2, r3 = resolution, resolution ** 2, resolution ** 3
    lin = np.arange(0, n_blocks * r3, 1, dtype=np.int64)
    block_idx = lin // r3
    rem = lin - block_idx * r3
    voxel_z = rem // r2
    rem = rem - voxel_z * r2
    voxel_y = rem // r
    voxel_x = rem - voxel_y * r
    out = np.zeros((4, n_blocks * r3), np.int64)
    out[0] = buf_indices[block_idx].astype(np.int64)
    out[1], out[2], out[3] = voxel_x, voxel_y, voxel_z
    return out


def voxel_coordinates(voxel_indices_, key_tensor, resolution):
    """VoxelBlockGrid::GetVoxelCoordinates (VoxelBlockGrid.cpp:130-143)."""
    vc = key_tensor[voxel_indices_[0]].T.astype(np.int64) * resolution
    vc[0] += voxel_indices_[1]
    vc[1] += voxel_indices_[2]
    vc[2] += voxel_indices_[3]
    return vc


class HashMap:
    """Insert-if-absent map int3 -> buf_index with heap-ordered indices."""

    def __init__(self, capacity):
        self.h = c_vp(lib().orc_hash_create(C.c_int64(int(capacity))))
        self.capacity = int(capacity)

    def __del__(self):
        try:
            lib().orc_hash_destroy(self.h)
        except Exception:
            pass

    def size(self):
        return int(lib().orc_hash_size(self.h))

    def activate(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        n = keys.shape[0]
        buf = np.zeros(n, np.int32)
        masks = np.zeros(n, np.uint8)
        st = lib().orc_hash_activate(self.h, _p(keys), C.c_int64(n), _p(buf),
                                     _p(masks))
        if st != 0:
            raise RuntimeError("oracle hash map capacity exceeded")
        return buf, masks.astype(bool)

    def find(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        n = keys.shape[0]
        buf = np.zeros(n, np.int32)
        masks = np.zeros(n, np.uint8)
        lib().orc_hash_find(self.h, _p(keys), C.c_int64(n), _p(buf), _p(masks))
        return buf, masks.astype(bool)

    def key_buffer(self):
        ptr = lib().orc_hash_key_buffer(self.h)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)),
                                    shape=(self.capacity, 3))
        return arr

    def active_indices(self):
        out = np.zeros(self.size(), np.int32)
        lib().orc_hash_active_indices(self.h, _p(out))
        return out


def integrate(depth, color, indices, block_keys, tsdf, weight, color_buf, K_d,
              K_c, T, resolution, voxel_size, sdf_trunc, depth_scale,
              depth_max):
    """In-place on tsdf / weight / color_buf (numpy arrays)."""
    depth = np.ascontiguousarray(depth)
    input_is_f32 = int(depth.dtype == np.float32)
    grid_is_f32 = int(weight.dtype == np.float32)
    if color is not None and color.size > 0:
        color = np.ascontiguousarray(color)
        assert color.dtype == (np.float32 if input_is_f32 else np.uint8)
        crow, ccol = color.shape[:2]
    else:
        color, crow, ccol = None, 0, 0
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    assert tsdf.dtype == np.float32 and tsdf.flags.c_contiguous
    assert weight.flags.c_contiguous
    K_d, K_c, T = _f64(K_d), _f64(K_c), _f64(T)
    lib().orc_integrate(_p(depth), depth.shape[0], depth.shape[1], _p(color),
                        crow, ccol, input_is_f32, _p(indices),
                        C.c_int64(indices.shape[0]), _p(block_keys), _p(tsdf),
                        _p(weight), _p(color_buf), grid_is_f32, _p(K_d),
                        _p(K_c), _p(T), int(resolution), C.c_float(voxel_size),
                        C.c_float(sdf_trunc), C.c_float(depth_scale),
                        C.c_float(depth_max))


def estimate_range(block_keys, K, T, h, w, down_factor, block_resolution,
                   voxel_size, depth_min, depth_max, frag_buffer_size=0):
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    out = np.zeros((h // down_factor, w // down_factor, 2), np.float32)
    K, T = _f64(K), _f64(T)
    needed = lib().orc_estimate_range(
            _p(block_keys), C.c_int64(block_keys.shape[0]), _p(out), _p(K),
            _p(T), int(h), int(w), int(down_factor),
            C.c_int64(block_resolution), C.c_float(voxel_size),
            C.c_float(depth_min), C.c_float(depth_max), int(frag_buffer_size))
    return out, int(needed)


def raycast(hashmap, tsdf, weight, color_buf, range_map, K, T, h, w,
            block_resolution, voxel_size, depth_scale, depth_min, depth_max,
            weight_threshold, trunc_voxel_multiplier, range_map_down_factor,
            attrs=("depth", "color")):
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
        # Garbage-fill to make sure the kernel initialises every pixel.
        out[a] = np.full((h, w, c), 77, dt)
    g = lambda a: _p(out[a]) if a in out else None
    range_map = np.ascontiguousarray(range_map, dtype=np.float32)
    lib().orc_raycast(hashmap.h, _p(tsdf), _p(weight), _p(color_buf),
                      grid_is_f32, _p(range_map), g("depth"), g("vertex"),
                      g("color"), g("normal"), g("index"), g("mask"),
                      g("interp_ratio"), g("interp_ratio_dx"),
                      g("interp_ratio_dy"), g("interp_ratio_dz"), _p(K), _p(T),
                      int(h), int(w), int(block_resolution),
                      C.c_float(voxel_size), C.c_float(depth_scale),
                      C.c_float(depth_min), C.c_float(depth_max),
                      C.c_float(weight_threshold),
                      C.c_float(trunc_voxel_multiplier),
                      int(range_map_down_factor))
    if "mask" in out:
        out["mask"] = out["mask"].astype(bool)
    return out


def buffer_radius_neighbors(hashmap, active_buf_indices):
    """BufferRadiusNeighbors (VoxelBlockGrid.cpp:22-51): ({27,n} int32,
    {27,n} bool)."""
    idx = np.ascontiguousarray(active_buf_indices, dtype=np.int32)
    n = idx.shape[0]
    nbi = np.zeros((27, n), np.int32)
    nbm = np.zeros((27, n), np.uint8)
    lib().orc_buffer_radius_neighbors(hashmap.h, _p(idx), C.c_int64(n),
                                      _p(nbi), _p(nbm))
    return nbi, nbm


def extract_point_cloud(indices, nb_indices, nb_masks, block_keys, tsdf,
                        weight, color_buf, resolution, voxel_size,
                        weight_threshold, estimated_number=-1):
    """ExtractPointCloudCPU: (points, normals, colors|None, total_count)."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    nb_indices = np.ascontiguousarray(nb_indices, dtype=np.int32)
    nb_masks = np.ascontiguousarray(nb_masks, dtype=np.uint8)
    block_keys = np.ascontiguousarray(block_keys, dtype=np.int32)
    n = indices.shape[0]
    grid_is_f32 = int(weight.dtype == np.float32)
    L = lib()
    L.orc_extract_point_cloud.restype = C.c_int64
    args = [_p(indices), _p(nb_indices), _p(nb_masks), _p(block_keys),
            _p(tsdf), _p(weight), _p(color_buf), grid_is_f32, C.c_int64(n),
            int(resolution), C.c_float(voxel_size),
            C.c_float(weight_threshold)]
    cap = int(estimated_number)
    if cap < 0:
        cap = int(L.orc_extract_point_cloud(*args, None, None, None,
                                            C.c_int64(0)))
    pts = np.zeros((cap, 3), np.float32)
    nrm = np.zeros((cap, 3), np.float32)
    col = np.zeros((cap, 3), np.float32) if color_buf is not None else None
    total = int(L.orc_extract_point_cloud(*args, _p(pts), _p(nrm), _p(col),
                                          C.c_int64(cap)))
    m = min(total, cap)
    return pts[:m], nrm[:m], (None if col is None else col[:m]), total


def unproject(depth, colors_f32, K, T, depth_scale, depth_max, stride=1):
    depth = np.ascontiguousarray(depth)
    is_f32 = int(depth.dtype == np.float32)
    rows, cols = depth.shape[:2]
    n = (rows // stride) * (cols // stride)
    pts = np.zeros((n, 3), np.float32)
    cols_out = None
    if colors_f32 is not None:
        colors_f32 = np.ascontiguousarray(colors_f32, dtype=np.float32)
        cols_out = np.zeros((n, 3), np.float32)
    K, T = _f64(K), _f64(T)
    m = lib().orc_unproject(_p(depth), is_f32, rows, cols, _p(colors_f32),
                            _p(pts), _p(cols_out), _p(K), _p(T),
                            C.c_float(depth_scale), C.c_float(depth_max),
                            C.c_int64(stride))
    if cols_out is None:
        return pts[:m].copy(), None
    return pts[:m].copy(), cols_out[:m].copy()


# ---------------------------------------------------------------- ICP side --
def robust_weight(method, scaling, shape, residual, f64=False):
    return float(lib().orc_robust_weight(int(f64), int(method),
                                         C.c_double(scaling),
                                         C.c_double(shape),
                                         C.c_double(residual)))


def hybrid_search(points, queries, radius, max_knn, brute=False):
    points = np.ascontiguousarray(points)
    queries = np.ascontiguousarray(queries, dtype=points.dtype)
    is_f64 = int(points.dtype == np.float64)
    q = queries.shape[0]
    idx = np.zeros((q, max_knn), np.int32)
    dist = np.zeros((q, max_knn), points.dtype)
    cnt = np.zeros(q, np.int32)
    lib().orc_hybrid_search(_p(points), C.c_int64(points.shape[0]),
                            _p(queries), C.c_int64(q), is_f64,
                            C.c_double(radius), int(max_knn), int(brute),
                            _p(idx), _p(dist), _p(cnt))
    return idx, dist, cnt


def knn_search(points, queries, knn):
    """NearestNeighborSearch::KnnSearch: (idx {q,k} int32, dist2 {q,k})."""
    points = np.ascontiguousarray(points)
    queries = np.ascontiguousarray(queries, dtype=points.dtype)
    k = min(int(knn), points.shape[0])
    q = queries.shape[0]
    idx = np.zeros((q, k), np.int32)
    dist = np.zeros((q, k), points.dtype)
    lib().orc_knn_search(_p(points), C.c_int64(points.shape[0]), _p(queries),
                         C.c_int64(q), int(points.dtype == np.float64),
                         int(knn), _p(idx), _p(dist))
    return idx, dist


def p2plane_accumulate(src, tgt, tgt_n, corr, method=0, scaling=1.0, shape=1.0,
                       accumulate_double=False):
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    tgt_n = np.ascontiguousarray(tgt_n, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    lib().orc_p2plane_accumulate(_p(src), _p(tgt), _p(tgt_n), _p(corr),
                                 C.c_int64(src.shape[0]),
                                 int(src.dtype == np.float64), int(method),
                                 C.c_double(scaling), C.c_double(shape),
                                 int(accumulate_double), _p(out))
    return out


def information_accumulate(tgt, corr, accumulate_double=False):
    """The 21 packed sums of ComputeInformationMatrixKernelCPU."""
    tgt = np.ascontiguousarray(tgt)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(21, np.float64)
    lib().orc_information_accumulate(_p(tgt), _p(corr),
                                     C.c_int64(corr.shape[0]),
                                     int(tgt.dtype == np.float64),
                                     int(accumulate_double), _p(out))
    return out


def unpack21(sums21):
    G = np.zeros((6, 6))
    i = 0
    for j in range(6):
        for k in range(j + 1):
            G[j, k] = G[k, j] = sums21[i]
            i += 1
    return G


def information_matrix(source, target, max_dist, transformation=None,
                       accumulate_double=False):
    """registration::GetInformationMatrix -> (status, GTG {6,6})."""
    source = np.ascontiguousarray(source)
    target = np.ascontiguousarray(target, dtype=source.dtype)
    T = _f64(np.eye(4) if transformation is None else transformation)
    G = np.zeros((6, 6), np.float64)
    st = lib().orc_information_matrix(
        _p(source), C.c_int64(source.shape[0]), _p(target),
        C.c_int64(target.shape[0]), int(source.dtype == np.float64),
        C.c_double(max_dist), _p(T), int(accumulate_double), _p(G))
    return st, G


def evaluate_registration(source, target, max_dist, transformation=None):
    """registration::EvaluateRegistration."""
    source = np.ascontiguousarray(source)
    target = np.ascontiguousarray(target, dtype=source.dtype)
    T = _f64(np.eye(4) if transformation is None else transformation)
    outT = np.zeros((4, 4), np.float64)
    fit, rmse = C.c_double(0), C.c_double(0)
    corr = np.zeros(source.shape[0], np.int64)
    lib().orc_evaluate_registration(
        _p(source), C.c_int64(source.shape[0]), _p(target),
        C.c_int64(target.shape[0]), int(source.dtype == np.float64),
        C.c_double(max_dist), _p(T), _p(outT), C.byref(fit), C.byref(rmse),
        _p(corr))
    return dict(transformation=outT, fitness=fit.value, inlier_rmse=rmse.value,
                correspondences=corr)


def symmetric_accumulate(src, tgt, sn, tn, corr, source_mean, target_mean,
                         method=0, scaling=1.0, shape=1.0,
                         accumulate_double=False):
    src = np.ascontiguousarray(src)
    dt = src.dtype
    tgt, sn, tn = (np.ascontiguousarray(a, dtype=dt) for a in (tgt, sn, tn))
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    lib().orc_symmetric_accumulate(
        _p(src), _p(tgt), _p(sn), _p(tn), _p(corr), C.c_int64(src.shape[0]),
        int(dt == np.float64), _p(_f64(source_mean)), _p(_f64(target_mean)),
        int(method), C.c_double(scaling), C.c_double(shape),
        int(accumulate_double), _p(out))
    return out


def colored_accumulate(src, src_c, tgt, tn, tc, tg, corr, lambda_geometric,
                       method=0, scaling=1.0, shape=1.0,
                       accumulate_double=False):
    src = np.ascontiguousarray(src)
    dt = src.dtype
    src_c, tgt, tn, tc, tg = (np.ascontiguousarray(a, dtype=dt)
                              for a in (src_c, tgt, tn, tc, tg))
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    out = np.zeros(29, np.float64)
    lib().orc_colored_accumulate(
        _p(src), _p(src_c), _p(tgt), _p(tn), _p(tc), _p(tg), _p(corr),
        C.c_int64(src.shape[0]), int(dt == np.float64),
        C.c_double(lambda_geometric), int(method), C.c_double(scaling),
        C.c_double(shape), int(accumulate_double), _p(out))
    return out


def symmetric_pose_to_transformation(pose, source_mean, target_mean):
    T = np.zeros((4, 4), np.float64)
    lib().orc_symmetric_pose_to_transformation(
        _p(_f64(pose)), _p(_f64(source_mean)), _p(_f64(target_mean)), _p(T))
    return T


def compute_transformation_symmetric(src, tgt, sn, tn, corr, method=0,
                                     scaling=1.0, shape=1.0,
                                     accumulate_double=False):
    """TransformationEstimationSymmetric::ComputeTransformation ->
    (status, T {4,4}, sums29)."""
    src = np.ascontiguousarray(src)
    dt = src.dtype
    tgt, sn, tn = (np.ascontiguousarray(a, dtype=dt) for a in (tgt, sn, tn))
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    T = np.zeros((4, 4), np.float64)
    sums = np.zeros(29, np.float64)
    st = lib().orc_compute_transformation_symmetric(
        _p(src), _p(tgt), _p(sn), _p(tn), _p(corr), C.c_int64(src.shape[0]),
        int(dt == np.float64), int(method), C.c_double(scaling),
        C.c_double(shape), int(accumulate_double), _p(T), _p(sums))
    return st, T, sums


def compute_rt_p2point(src, tgt, corr, accumulate_double=False):
    """ComputeRtPointToPoint: (R {3,3}, t {3}, count)."""
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    R = np.zeros((3, 3), np.float64)
    t = np.zeros(3, np.float64)
    f = lib().orc_compute_rt_p2point
    f.restype = C.c_int64
    c = f(_p(src), _p(tgt), _p(corr), C.c_int64(src.shape[0]),
          int(src.dtype == np.float64), int(accumulate_double), _p(R), _p(t))
    return R, t, int(c)


def p2point_sxy(src, tgt, corr, accumulate_double=False):
    """Get3x3SxyLinearSystem: (Sxy {3,3}, source_mean, target_mean, count)."""
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    S = np.zeros((3, 3), np.float64)
    ms = np.zeros(3, np.float64)
    mt = np.zeros(3, np.float64)
    f = lib().orc_p2point_sxy
    f.restype = C.c_int64
    c = f(_p(src), _p(tgt), _p(corr), C.c_int64(src.shape[0]),
          int(src.dtype == np.float64), int(accumulate_double), _p(S), _p(ms),
          _p(mt))
    return S, ms, mt, int(c)


def rt_from_sxy(Sxy, source_mean, target_mean, as_f32=False):
    R = np.zeros((3, 3), np.float64)
    t = np.zeros(3, np.float64)
    lib().orc_rt_from_sxy(_p(_f64(Sxy)), _p(_f64(source_mean)),
                          _p(_f64(target_mean)), int(as_f32), _p(R), _p(t))
    return R, t


def decode_and_solve6x6(A29):
    A29 = _f64(A29)
    pose = np.zeros(6, np.float64)
    residual = C.c_float(0)
    count = C.c_int(0)
    st = lib().orc_decode_and_solve6x6(_p(A29), _p(pose), C.byref(residual),
                                       C.byref(count))
    return st, pose, residual.value, count.value


def solve(A, b):
    A, b = _f64(A), _f64(b)
    n = A.shape[0]
    x = np.zeros(n, np.float64)
    st = lib().orc_solve(int(n), _p(A), _p(b), _p(x))
    if st != 0:
        raise RuntimeError("singular matrix")
    return x


def pose_to_transformation(pose):
    pose = _f64(pose)
    T = np.zeros((4, 4), np.float64)
    lib().orc_pose_to_transformation(_p(pose), _p(T))
    return T


def transform_points(T, pts):
    """Returns a transformed copy."""
    T = _f64(T)
    pts = np.array(pts, copy=True, order="C")
    lib().orc_transform_points(_p(T), _p(pts), C.c_int64(pts.shape[0]),
                               int(pts.dtype == np.float64))
    return pts


def transform_normals(T, nrm):
    T = _f64(T)
    nrm = np.array(nrm, copy=True, order="C")
    lib().orc_transform_normals(_p(T), _p(nrm), C.c_int64(nrm.shape[0]),
                                int(nrm.dtype == np.float64))
    return nrm


def p2plane_rmse(src, tgt, tgt_n, corr):
    src = np.ascontiguousarray(src)
    tgt = np.ascontiguousarray(tgt, dtype=src.dtype)
    tgt_n = np.ascontiguousarray(tgt_n, dtype=src.dtype)
    corr = np.ascontiguousarray(corr, dtype=np.int64).reshape(-1)
    return float(lib().orc_p2plane_rmse(_p(src), _p(tgt), _p(tgt_n), _p(corr),
                                        C.c_int64(src.shape[0]),
                                        int(src.dtype == np.float64)))


def voxel_down_sample(pos, nrm, voxel_size):
    pos = np.ascontiguousarray(pos)
    n = pos.shape[0]
    op = np.zeros_like(pos)
    if nrm is not None:
        nrm = np.ascontiguousarray(nrm, dtype=pos.dtype)
        on = np.zeros_like(pos)
    else:
        on = None
    m = lib().orc_voxel_down_sample(_p(pos), _p(nrm), C.c_int64(n),
                                    int(pos.dtype == np.float64),
                                    C.c_double(voxel_size), _p(op), _p(on))
    return op[:m].copy(), (None if on is None else on[:m].copy())


ICP_CB = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                     C.c_double, c_dp, c_vp)


def multiscale_icp(source, target, target_normals, voxel_sizes, criterias,
                   max_dists, init=None, kernel=(0, 1.0, 1.0),
                   accumulate_double=False, callback=None, estimation=0,
                   source_normals=None, source_colors=None, target_colors=None,
                   target_color_gradients=None, lambda_geometric=0.968):
    """criterias: list of (relative_fitness, relative_rmse, max_iteration).
    estimation: 0 = point-to-plane, 1 = point-to-point (normals unused),
    2 = symmetric (needs source_normals), 3 = colored (needs the colours)."""
    source = np.ascontiguousarray(source)
    dt = source.dtype
    if source_normals is not None:
        source_normals = np.ascontiguousarray(source_normals, dtype=dt)
    if source_colors is not None:
        source_colors = np.ascontiguousarray(source_colors, dtype=dt)
    if target_colors is not None:
        target_colors = np.ascontiguousarray(target_colors, dtype=dt)
    if target_color_gradients is not None:
        target_color_gradients = np.ascontiguousarray(target_color_gradients,
                                                      dtype=dt)
    target = np.ascontiguousarray(target, dtype=dt)
    if target_normals is not None:
        target_normals = np.ascontiguousarray(target_normals, dtype=dt)
    ns, nt = source.shape[0], target.shape[0]
    S = len(criterias)
    vs = _f64(voxel_sizes)
    mi = np.ascontiguousarray([c[2] for c in criterias], dtype=np.int32)
    rf = _f64([c[0] for c in criterias])
    rr = _f64([c[1] for c in criterias])
    md = _f64(max_dists)
    init = _f64(np.eye(4) if init is None else init)
    T = np.zeros((4, 4), np.float64)
    fit, rmse = C.c_double(0), C.c_double(0)
    conv, nit = C.c_int(0), C.c_int(0)
    corr = np.full(ns, -1, np.int64)
    ncorr = C.c_int64(0)
    cb = ICP_CB(0)
    if callback is not None:
        def _cb(it, sc, sit, r, f, Tp, user):
            callback(dict(iteration_index=it, scale_index=sc,
                          scale_iteration_index=sit, inlier_rmse=r, fitness=f,
                          transformation=np.ctypeslib.as_array(
                                  Tp, shape=(16,)).reshape(4, 4).copy()))
        cb = ICP_CB(_cb)
    st = lib().orc_multiscale_icp_ex(
            _p(source),
            _p(source_normals) if source_normals is not None else None,
            _p(source_colors) if source_colors is not None else None,
            _p(target_colors) if target_colors is not None else None,
            _p(target_color_gradients)
            if target_color_gradients is not None else None,
            C.c_double(lambda_geometric), C.c_int64(ns), _p(target),
            _p(target_normals) if target_normals is not None else None,
            C.c_int64(nt), int(dt == np.float64), int(S), _p(vs), _p(mi),
            _p(rf), _p(rr), _p(md), _p(init), int(estimation), int(kernel[0]),
            C.c_double(kernel[1]), C.c_double(kernel[2]),
            int(accumulate_double), _p(T), C.byref(fit), C.byref(rmse),
            C.byref(conv), C.byref(nit), _p(corr), C.byref(ncorr), cb, None)
    return dict(status=st, transformation=T, fitness=fit.value,
                inlier_rmse=rmse.value, converged=bool(conv.value),
                num_iterations=nit.value,
                correspondences=corr[:ncorr.value].copy())


# ---------------------------------------------------------------------------
# RGB-D odometry front end (oracle/odometry_oracle.cpp)
# ---------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _img_dtype_code(a):
    return {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1,
            np.dtype(np.float32): 2}[a.dtype]


def clip_transform(src, scale, min_value, max_value, clip_fill):
    src = np.ascontiguousarray(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols), np.float32)
    lib().orc_clip_transform(_p(src), int(src.dtype == np.float32),
                             C.c_int64(rows), C.c_int64(cols),
                             C.c_float(scale), C.c_float(min_value),
                             C.c_float(max_value), C.c_float(clip_fill),
                             _p(dst))
    return dst


def pyrdown_depth(src, depth_diff, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows // 2, cols // 2), np.float32)
    lib().orc_pyrdown_depth(_p(src), rows, cols, C.c_float(depth_diff),
                            C.c_float(invalid_fill), _p(dst))
    return dst


def create_vertex_map(src, K, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols, 3), np.float32)
    lib().orc_create_vertex_map(_p(src), C.c_int64(rows), C.c_int64(cols),
                                _p(_f64(K)), C.c_float(invalid_fill), _p(dst))
    return dst


def create_normal_map(src, invalid_fill):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols, 3), np.float32)
    lib().orc_create_normal_map(_p(src), C.c_int64(rows), C.c_int64(cols),
                                C.c_float(invalid_fill), _p(dst))
    return dst


def image_to_float(src, scale, offset=0.0):
    src = np.ascontiguousarray(src)
    dst = np.empty(src.shape, np.float32)
    lib().orc_image_to_float(_p(src), _img_dtype_code(src),
                             C.c_int64(src.size), C.c_double(scale),
                             C.c_double(offset), _p(dst))
    return dst


def rgb_to_gray(src):
    src = np.ascontiguousarray(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols), src.dtype)
    lib().orc_rgb_to_gray(_p(src), _img_dtype_code(src),
                          C.c_int64(rows * cols), _p(dst))
    return dst


def filter_bilateral(src, kernel_size, value_sigma, distance_sigma):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols), np.float32)
    lib().orc_filter_bilateral(_p(src), rows, cols, int(kernel_size),
                               C.c_float(value_sigma),
                               C.c_float(distance_sigma), _p(dst))
    return dst


def filter_gaussian(src, kernel_size=3, sigma=1.0):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((rows, cols), np.float32)
    lib().orc_filter_gaussian(_p(src), rows, cols, int(kernel_size),
                              C.c_float(sigma), _p(dst))
    return dst


def filter_sobel(src):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dx = np.empty((rows, cols), np.float32)
    dy = np.empty((rows, cols), np.float32)
    lib().orc_filter_sobel(_p(src), rows, cols, _p(dx), _p(dy))
    return dx, dy


def resize_half_nearest(src):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((int(rows * 0.5), int(cols * 0.5)), np.float32)
    lib().orc_resize_half_nearest(_p(src), rows, cols, _p(dst))
    return dst


def pyrdown(src):
    src = _f32(src)
    rows, cols = src.shape[:2]
    dst = np.empty((int(rows * 0.5), int(cols * 0.5)), np.float32)
    lib().orc_pyrdown(_p(src), rows, cols, _p(dst))
    return dst


ODO_P2PLANE, ODO_INTENSITY, ODO_HYBRID = 0, 1, 2


def _opt32(a):
    return None if a is None else _f32(a)


def odometry_sums(method, K, T, source_vertex, target_vertex=None,
                  target_normal=None, source_depth=None, target_depth=None,
                  source_intensity=None, target_intensity=None,
                  target_depth_dx=None, target_depth_dy=None,
                  target_intensity_dx=None, target_intensity_dy=None,
                  depth_outlier_trunc=0.07, depth_huber_delta=0.05,
                  intensity_huber_delta=0.1, accumulate_double=False):
    sv = _f32(source_vertex)
    rows, cols = sv.shape[:2]
    arrs = [_opt32(a) for a in (source_depth, target_depth, source_intensity,
                                target_intensity, target_depth_dx,
                                target_depth_dy, target_intensity_dx,
                                target_intensity_dy)]
    tv, tn = _opt32(target_vertex), _opt32(target_normal)
    out = np.zeros(29, np.float64)
    lib().orc_odometry_sums(int(method), rows, cols, *[_p(a) for a in arrs],
                            _p(sv), _p(tv), _p(tn), _p(_f64(K)), _p(_f64(T)),
                            C.c_float(depth_outlier_trunc),
                            C.c_float(depth_huber_delta),
                            C.c_float(intensity_huber_delta),
                            int(bool(accumulate_double)), _p(out))
    return out


def odometry_information(source_vertex, target_vertex, K, T, square_dist_thr,
                         accumulate_double=False):
    sv, tv = _f32(source_vertex), _f32(target_vertex)
    rows, cols = sv.shape[:2]
    out = np.zeros((6, 6), np.float64)
    lib().orc_odometry_information(rows, cols, _p(sv), _p(tv), _p(_f64(K)),
                                   _p(_f64(T)), C.c_float(square_dist_thr),
                                   int(bool(accumulate_double)), _p(out))
    return out


def rgbd_odometry_multiscale(method, src_depth, tgt_depth, K, init=None,
                             src_color=None, tgt_color=None,
                             depth_scale=1000.0, depth_max=3.0,
                             criteria=((6, 1e-6, 1e-6), (3, 1e-6, 1e-6),
                                       (1, 1e-6, 1e-6)),
                             depth_outlier_trunc=0.07, depth_huber_delta=0.05,
                             intensity_huber_delta=0.1,
                             accumulate_double=False):
    """criteria: list of (max_iteration, relative_rmse, relative_fitness),
    coarse to fine (OdometryConvergenceCriteria, RGBDOdometry.h:38-68)."""
    sd = np.ascontiguousarray(src_depth)
    td = np.ascontiguousarray(tgt_depth)
    assert sd.dtype in (np.uint16, np.float32)
    assert td.dtype in (np.uint16, np.float32)
    rows, cols = sd.shape[:2]
    sc = None if src_color is None else np.ascontiguousarray(src_color)
    tc = None if tgt_color is None else np.ascontiguousarray(tgt_color)
    init = np.eye(4) if init is None else init
    iters = np.array([c[0] for c in criteria], np.int32)
    rr = np.array([c[1] for c in criteria], np.float64)
    rf = np.array([c[2] for c in criteria], np.float64)
    T = np.zeros((4, 4), np.float64)
    rmse, fit = C.c_double(0), C.c_double(0)
    it = C.c_int(0)
    st = lib().orc_rgbd_odometry_multiscale(
        int(method), _p(sd), _p(sc), _p(td), _p(tc),
        int(sd.dtype == np.float32), int(td.dtype == np.float32),
        int(sc is not None and sc.dtype == np.float32),
        int(tc is not None and tc.dtype == np.float32), rows, cols,
        _p(_f64(K)),
        _p(_f64(init)), C.c_float(depth_scale), C.c_float(depth_max),
        len(criteria), _p(iters), _p(rr), _p(rf),
        C.c_float(depth_outlier_trunc), C.c_float(depth_huber_delta),
        C.c_float(intensity_huber_delta), int(bool(accumulate_double)), _p(T),
        C.byref(rmse), C.byref(fit), C.byref(it))
    return {"status": int(st), "transformation": T, "inlier_rmse": rmse.value,
            "fitness": fit.value, "iterations": it.value}


# ---------------------------------------------------------------------------
# Normal estimation (row f4)
# ---------------------------------------------------------------------------
def estimate_covariances(points, indices, counts):
    points = np.ascontiguousarray(points)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, max_nn = indices.shape
    cov = np.zeros((n, 3, 3), points.dtype)
    lib().orc_estimate_covariances(_p(points), _p(indices), _p(counts),
                                   C.c_int64(n), int(max_nn),
                                   int(points.dtype == np.float64), _p(cov))
    return cov


def svd3x3(A):
    A = np.ascontiguousarray(A)
    U, S, V = np.zeros((3, 3), A.dtype), np.zeros(3, A.dtype), \
        np.zeros((3, 3), A.dtype)
    lib().orc_svd3x3(_p(A), int(A.dtype == np.float64), _p(U), _p(S), _p(V))
    return U, S, V


def set_exact_color_gradients(on):
    """ICP driver: colour gradients it estimates itself use the converged
    pseudo-inverse (the product's solver) instead of the reference's
    approximate solve_svd3x3."""
    lib().orc_set_exact_color_gradients(int(bool(on)))


def set_p2plane_hook(fn_address):
    """Point-to-plane ICP driver: take the 29 sums from an external function
    (address of one with _ref's ref_p2plane_accumulate signature; None
    restores the oracle's own)."""
    lib().orc_set_p2plane_hook.argtypes = [C.c_void_p]
    lib().orc_set_p2plane_hook(C.c_void_p(fn_address or 0))


def solve_svd3x3(A, b):
    A = np.ascontiguousarray(A)
    b = np.ascontiguousarray(b, dtype=A.dtype)
    x = np.zeros(3, A.dtype)
    lib().orc_solve_svd3x3(_p(A), _p(b), int(A.dtype == np.float64), _p(x))
    return x


def estimate_color_gradients(points, normals, colors, indices, counts,
                             exact_solve=False):
    points = np.ascontiguousarray(points)
    dt = points.dtype
    normals = np.ascontiguousarray(normals, dtype=dt)
    colors = np.ascontiguousarray(colors, dtype=dt)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, max_nn = indices.shape
    g = np.zeros((n, 3), dt)
    lib().orc_estimate_color_gradients(
        _p(points), _p(normals), _p(colors), _p(indices), _p(counts),
        C.c_int64(n), int(max_nn), int(dt == np.float64), int(exact_solve),
        _p(g))
    return g


def normals_from_covariances(cov, normals=None):
    cov = np.ascontiguousarray(cov)
    n = cov.shape[0]
    has = normals is not None
    out = np.ascontiguousarray(normals, dtype=cov.dtype).copy() if has \
        else np.zeros((n, 3), cov.dtype)
    lib().orc_normals_from_covariances(_p(cov), C.c_int64(n),
                                       int(cov.dtype == np.float64), _p(out),
                                       int(has))
    return out


def estimate_normals(points, radius, max_nn, normals=None):
    """PointCloud::EstimateNormals(max_nn, radius) -- hybrid search."""
    idx, _, cnt = hybrid_search(points, points, radius, max_nn)
    return normals_from_covariances(estimate_covariances(points, idx, cnt),
                                    normals)
