"""Python mirror of open3d.t.pipelines.odometry (RGB-D odometry) and of the
t.geometry.Image operations its front end uses, for the MI355X backend.

Names / defaults follow cpp/pybind/t/pipelines/odometry/odometry.cpp and
t/pipelines/odometry/RGBDOdometry.h:23-195; image ops follow
t/geometry/Image.h (ClipTransform, PyrDownDepth, CreateVertexMap,
CreateNormalMap, To, RGBToGray, FilterBilateral, FilterGaussian, FilterSobel,
PyrDown). Images are torch device tensors {H,W[,C]}.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .core import TORCH_TO_O3DMI, host_mat, require_cuda, stream


class Method:
    PointToPlane, Intensity, Hybrid = range(3)


class OdometryConvergenceCriteria:
    def __init__(self, max_iteration, relative_rmse=1e-6,
                 relative_fitness=1e-6):
        self.max_iteration = max_iteration
        self.relative_rmse = relative_rmse
        self.relative_fitness = relative_fitness


class OdometryLossParams:
    def __init__(self, depth_outlier_trunc=0.07, depth_huber_delta=0.05,
                 intensity_huber_delta=0.1):
        self.depth_outlier_trunc = depth_outlier_trunc
        self.depth_huber_delta = depth_huber_delta
        self.intensity_huber_delta = intensity_huber_delta


class OdometryResult:
    def __init__(self, transformation=None, inlier_rmse=0.0, fitness=0.0):
        self.transformation = np.eye(4) if transformation is None \
            else transformation
        self.inlier_rmse = inlier_rmse
        self.fitness = fitness
        self.num_iterations = 0


def _img2(t, name):
    t = require_cuda(t, name)
    if t.dim() == 3 and t.shape[2] == 1:
        t = t[..., 0]
    if t.dim() != 2:
        raise ValueError("%s must be {rows, cols} or {rows, cols, 1}" % name)
    return t.contiguous()


def rgbd_odometry_multi_scale(source_depth, target_depth, intrinsics,
                              init_source_to_target=None, depth_scale=1000.0,
                              depth_max=3.0, criteria_list=(10, 5, 3),
                              method=Method.Hybrid, params=None,
                              source_color=None, target_color=None):
    """RGBDOdometryMultiScale (RGBDOdometry.cpp:56-108). criteria_list: list
    of OdometryConvergenceCriteria or plain iteration counts, coarse to fine
    (the reference's implicit int -> criteria conversion)."""
    sd = _img2(source_depth, "source depth")
    td = _img2(target_depth, "target depth")
    for d in (sd, td):
        if d.dtype not in (torch.uint16, torch.float32):
            raise ValueError("depth must be UInt16 or Float32")
    if sd.shape != td.shape:
        raise ValueError("source / target size mismatch")
    K = host_mat(intrinsics, (3, 3), "intrinsics")
    init = host_mat(np.eye(4) if init_source_to_target is None
                    else init_source_to_target, (4, 4),
                    "init_source_to_target")
    params = params or OdometryLossParams()
    crit = [c if isinstance(c, OdometryConvergenceCriteria)
            else OdometryConvergenceCriteria(int(c)) for c in criteria_list]
    cc = (_lib.OdometryCriteriaC * len(crit))(*[
        _lib.OdometryCriteriaC(c.max_iteration, c.relative_rmse,
                               c.relative_fitness) for c in crit])
    sc = tc = None
    scdt = tcdt = _lib.U8
    if source_color is not None or target_color is not None:
        sc = require_cuda(source_color, "source color")
        tc = require_cuda(target_color, "target color")
        for c in (sc, tc):
            if c.dtype not in (torch.uint8, torch.float32):
                raise ValueError("colour must be UInt8 or Float32")
            if tuple(c.shape) != (sd.shape[0], sd.shape[1], 3):
                raise ValueError("colour must be {rows, cols, 3}")
        scdt, tcdt = TORCH_TO_O3DMI[sc.dtype], TORCH_TO_O3DMI[tc.dtype]
    res = _lib.OdometryResultC()
    st = _lib.lib().o3dmi_rgbd_odometry_multiscale(
        _lib.ptr(sd), _lib.ptr(sc), _lib.ptr(td), _lib.ptr(tc),
        TORCH_TO_O3DMI[sd.dtype], scdt, TORCH_TO_O3DMI[td.dtype], tcdt,
        sd.shape[0], sd.shape[1],
        _lib.f64p(K), _lib.f64p(init), C.c_float(depth_scale),
        C.c_float(depth_max), len(crit), cc, int(method),
        C.c_float(params.depth_outlier_trunc),
        C.c_float(params.depth_huber_delta),
        C.c_float(params.intensity_huber_delta), C.byref(res), stream())
    _lib.check(st, "rgbd_odometry_multi_scale")
    out = OdometryResult(np.array(res.transformation[:]).reshape(4, 4),
                         res.inlier_rmse, res.fitness)
    out.num_iterations = res.num_iterations
    return out


def compute_odometry_information_matrix(source_depth, target_depth, intrinsics,
                                        source_to_target, dist_thr,
                                        depth_scale=1000.0, depth_max=3.0):
    """ComputeOdometryInformationMatrix (RGBDOdometry.cpp:488-513)."""
    sd = _img2(source_depth, "source depth")
    td = _img2(target_depth, "target depth")
    K = host_mat(intrinsics, (3, 3), "intrinsics")
    T = host_mat(source_to_target, (4, 4), "source_to_target")
    info = np.zeros((6, 6), np.float64)
    _lib.check(_lib.lib().o3dmi_rgbd_odometry_information_matrix(
        _lib.ptr(sd), _lib.ptr(td), TORCH_TO_O3DMI[sd.dtype], sd.shape[0],
        sd.shape[1], _lib.f64p(K), _lib.f64p(T), C.c_float(dist_thr),
        C.c_float(depth_scale), C.c_float(depth_max), _lib.f64p(info),
        stream()), "compute_odometry_information_matrix")
    return info


def compute_odometry_sums(method, intrinsics, init_source_to_target,
                          source_vertex, target_vertex=None,
                          target_normal=None, source_depth=None,
                          target_depth=None, source_intensity=None,
                          target_intensity=None, target_depth_dx=None,
                          target_depth_dy=None, target_intensity_dx=None,
                          target_intensity_dy=None, depth_outlier_trunc=0.07,
                          depth_huber_delta=0.05, intensity_huber_delta=0.1):
    """The reduction inside ComputeOdometryResult{PointToPlane,Intensity,
    Hybrid} (t/pipelines/kernel/RGBDOdometry.h:18-61): the 29 sums (float64
    numpy)."""
    sv = require_cuda(source_vertex, "source_vertex_map")
    rows, cols = sv.shape[0], sv.shape[1]
    K = host_mat(intrinsics, (3, 3), "intrinsics")
    T = host_mat(init_source_to_target, (4, 4), "init_source_to_target")
    maps = [source_depth, target_depth, source_intensity, target_intensity,
            target_depth_dx, target_depth_dy, target_intensity_dx,
            target_intensity_dy, sv, target_vertex, target_normal]
    for m in maps:
        if m is not None:
            require_cuda(m, "map")
            if m.dtype != torch.float32:
                raise ValueError("odometry maps must be Float32")
    sums = torch.empty(29, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().o3dmi_odometry_sums(
        int(method), rows, cols, *[_lib.ptr(m) for m in maps], _lib.f64p(K),
        _lib.f64p(T), C.c_float(depth_outlier_trunc),
        C.c_float(depth_huber_delta), C.c_float(intensity_huber_delta), None,
        _lib.ptr(sums), stream()), "compute_odometry_sums")
    return sums.cpu().numpy()


def compute_odometry_result(method, intrinsics, init_source_to_target, *maps,
                            **kw):
    """ComputeOdometryResult* (RGBDOdometry.cpp:391-486): one Gauss-Newton
    step -> OdometryResult(delta transformation, residual / count, count /
    (H W))."""
    A = compute_odometry_sums(method, intrinsics, init_source_to_target,
                              *maps, **kw)
    pose = np.zeros(6)
    residual, count = C.c_float(0), C.c_int(0)
    _lib.check(_lib.lib().o3dmi_decode_and_solve6x6(
        _lib.f64p(A), _lib.f64p(pose), C.byref(residual), C.byref(count)),
        "compute_odometry_result")
    if count.value <= 0:
        raise RuntimeError("Invalid inlier_count value %d, must be > 0."
                           % count.value)
    T = np.zeros((4, 4))
    _lib.lib().o3dmi_pose_to_transformation(_lib.f64p(pose), _lib.f64p(T))
    sv = kw.get("source_vertex", maps[0] if maps else None)
    n = sv.shape[0] * sv.shape[1]
    return OdometryResult(
        T, float(np.float32(residual.value) / np.float32(count.value)),
        count.value / n)


# ---------------------------------------------------------------------------
# t.geometry.Image operations (device tensors in, device tensors out)
# ---------------------------------------------------------------------------
def clip_transform(depth, scale, min_value, max_value, clip_fill=0.0):
    d = _img2(depth, "depth")
    out = torch.empty(d.shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_clip_transform(
        _lib.ptr(d), TORCH_TO_O3DMI[d.dtype], d.shape[0], d.shape[1],
        C.c_float(scale), C.c_float(min_value), C.c_float(max_value),
        C.c_float(clip_fill), _lib.ptr(out), stream()), "clip_transform")
    return out


def pyrdown_depth(depth, diff_threshold, invalid_fill=0.0):
    d = _img2(depth, "depth")
    out = torch.empty((d.shape[0] // 2, d.shape[1] // 2), dtype=torch.float32,
                      device="cuda")
    _lib.check(_lib.lib().o3dmi_image_pyrdown_depth(
        _lib.ptr(d), d.shape[0], d.shape[1], C.c_float(diff_threshold),
        C.c_float(invalid_fill), _lib.ptr(out), stream()), "pyrdown_depth")
    return out


def create_vertex_map(depth, intrinsics, invalid_fill=0.0):
    d = _img2(depth, "depth")
    K = host_mat(intrinsics, (3, 3), "intrinsics")
    out = torch.empty((d.shape[0], d.shape[1], 3), dtype=torch.float32,
                      device="cuda")
    _lib.check(_lib.lib().o3dmi_image_create_vertex_map(
        _lib.ptr(d), d.shape[0], d.shape[1], _lib.f64p(K),
        C.c_float(invalid_fill), _lib.ptr(out), stream()), "create_vertex_map")
    return out


def create_normal_map(vertex_map, invalid_fill=0.0):
    v = require_cuda(vertex_map, "vertex map")
    out = torch.empty_like(v)
    _lib.check(_lib.lib().o3dmi_image_create_normal_map(
        _lib.ptr(v), v.shape[0], v.shape[1], C.c_float(invalid_fill),
        _lib.ptr(out), stream()), "create_normal_map")
    return out


def to_float(image, scale=1.0, offset=0.0):
    t = require_cuda(image, "image")
    out = torch.empty(t.shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_to_float(
        _lib.ptr(t), TORCH_TO_O3DMI[t.dtype], t.numel(), C.c_double(scale),
        C.c_double(offset), _lib.ptr(out), stream()), "to_float")
    return out


def rgb_to_gray(color):
    c = require_cuda(color, "color")
    out = torch.empty(c.shape[:2], dtype=c.dtype, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_rgb_to_gray(
        _lib.ptr(c), TORCH_TO_O3DMI[c.dtype], c.shape[0] * c.shape[1],
        _lib.ptr(out), stream()), "rgb_to_gray")
    return out


def rgb_to_intensity(color):
    c = require_cuda(color, "color")
    out = torch.empty(c.shape[:2], dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_rgb_to_intensity(
        _lib.ptr(c), TORCH_TO_O3DMI[c.dtype], c.shape[0] * c.shape[1],
        _lib.ptr(out), stream()), "rgb_to_intensity")
    return out


def filter_bilateral(image, kernel_size=3, value_sigma=20.0,
                     distance_sigma=10.0):
    t = _img2(image, "image")
    out = torch.empty_like(t)
    _lib.check(_lib.lib().o3dmi_image_filter_bilateral(
        _lib.ptr(t), t.shape[0], t.shape[1], int(kernel_size),
        C.c_float(value_sigma), C.c_float(distance_sigma), _lib.ptr(out),
        stream()), "filter_bilateral")
    return out


def filter_gaussian(image, kernel_size=3, sigma=1.0):
    t = _img2(image, "image")
    out = torch.empty_like(t)
    _lib.check(_lib.lib().o3dmi_image_filter_gaussian(
        _lib.ptr(t), t.shape[0], t.shape[1], int(kernel_size),
        C.c_float(sigma), _lib.ptr(out), stream()), "filter_gaussian")
    return out


def filter_sobel(image):
    t = _img2(image, "image")
    dx, dy = torch.empty_like(t), torch.empty_like(t)
    _lib.check(_lib.lib().o3dmi_image_filter_sobel(
        _lib.ptr(t), t.shape[0], t.shape[1], _lib.ptr(dx), _lib.ptr(dy),
        stream()), "filter_sobel")
    return dx, dy


def resize_half_nearest(image):
    t = _img2(image, "image")
    out = torch.empty((int(t.shape[0] * 0.5), int(t.shape[1] * 0.5)),
                      dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_resize_half_nearest(
        _lib.ptr(t), t.shape[0], t.shape[1], _lib.ptr(out), stream()),
        "resize")
    return out


def pyrdown(image):
    t = _img2(image, "image")
    out = torch.empty((int(t.shape[0] * 0.5), int(t.shape[1] * 0.5)),
                      dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_image_pyrdown(
        _lib.ptr(t), t.shape[0], t.shape[1], _lib.ptr(out), stream()),
        "pyrdown")
    return out


def p2plane_level(source_depth, target_depth, intrinsics,
                  next_level_depth_diff=None):
    """One pyramid level of the point-to-plane method in a single launch:
    (source_vertex, target_vertex, target_normal[, source_depth_next,
    target_depth_next]); the next-level depths (PyrDownDepth with
    `next_level_depth_diff`, NaN fill) are produced when it is given."""
    sd = _img2(source_depth, "source depth")
    td = _img2(target_depth, "target depth")
    K = host_mat(intrinsics, (3, 3), "intrinsics")
    shp = (sd.shape[0], sd.shape[1], 3)
    sv = torch.empty(shp, dtype=torch.float32, device="cuda")
    tv = torch.empty(shp, dtype=torch.float32, device="cuda")
    tn = torch.empty(shp, dtype=torch.float32, device="cuda")
    sn = tdn = None
    if next_level_depth_diff is not None:
        half = (sd.shape[0] // 2, sd.shape[1] // 2)
        sn = torch.empty(half, dtype=torch.float32, device="cuda")
        tdn = torch.empty(half, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().o3dmi_odometry_p2plane_level(
        _lib.ptr(sd), _lib.ptr(td), sd.shape[0], sd.shape[1], _lib.f64p(K),
        _lib.ptr(sv), _lib.ptr(tv), _lib.ptr(tn), _lib.ptr(sn), _lib.ptr(tdn),
        C.c_float(next_level_depth_diff or 0.0), stream()), "p2plane_level")
    if sn is None:
        return sv, tv, tn
    return sv, tv, tn, sn, tdn
