"""Python mirror of open3d.t.pipelines.registration.{icp, multi_scale_icp}
for the MI355X backend (point-to-plane and point-to-point estimators).

Argument names / defaults follow the reference's binding
(cpp/pybind/t/pipelines/registration/registration.cpp) and
t/pipelines/registration/Registration.h:31-98,133-208. Point clouds are given
as torch device tensors {N,3} (float32 or float64).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .core import TORCH_TO_O3DMI, require_cuda, stream


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6,
                 max_iteration=30):
        self.relative_fitness = relative_fitness
        self.relative_rmse = relative_rmse
        self.max_iteration = max_iteration


class RobustKernel:
    L2Loss, L1Loss, HuberLoss, CauchyLoss, GMLoss, TukeyLoss, \
        GeneralizedLoss = range(7)

    def __init__(self, type=0, scaling_parameter=1.0, shape_parameter=1.0):
        self.type = type
        self.scaling_parameter = scaling_parameter
        self.shape_parameter = shape_parameter


class TransformationEstimationPointToPlane:
    def __init__(self, kernel=None):
        self.kernel = kernel or RobustKernel()


class TransformationEstimationPointToPoint:
    """TransformationEstimation.h:76-118; no robust kernel, no normals."""
    kernel = RobustKernel()


class TransformationEstimationSymmetric:
    """TransformationEstimation.h (Symmetric ICP, Rusinkiewicz 2019): needs
    source and target normals."""
    def __init__(self, kernel=None):
        self.kernel = kernel or RobustKernel()


class TransformationEstimationForColoredICP:
    """TransformationEstimation.h:283-349 (Park et al. 2017): needs target
    normals and colours on both clouds; `color_gradients` of the target are
    estimated when not given."""
    def __init__(self, lambda_geometric=0.968, kernel=None):
        self.lambda_geometric = lambda_geometric
        self.kernel = kernel or RobustKernel()


class RegistrationResult:
    def __init__(self):
        self.transformation = np.eye(4)
        self.correspondence_set = None
        self.inlier_rmse = 0.0
        self.fitness = 0.0
        self.converged = False
        self.num_iterations = 0


def multi_scale_icp(source, target, target_normals, voxel_sizes, criteria_list,
                    max_correspondence_distances, init_source_to_target=None,
                    estimation_method=None, callback_after_iteration=None,
                    allreduce=None, source_normals=None, source_colors=None,
                    target_colors=None, target_color_gradients=None,
                    device_allreduce=None, device_counts=None):
    """source/target/target_normals: device tensors {N,3}. `device_counts`
    (optional): (ns, nt) int32 device tensors of one element holding the LIVE
    sizes of source / target, whose tensors are then buffers of at least that
    many rows (o3dmi_registration_set_device_counts: no read-back of the
    sizes). `allreduce`
    (optional) sums a length-32 numpy float64 array over ranks in place;
    `device_allreduce(dev_ptr, n, stream_ptr)` (optional, takes precedence;
    sharding.make_device_allreduce) enqueues the same sum on the device.
    `source_normals` is read by the symmetric estimator, the colours (and the
    optional target colour gradients) by the coloured one."""
    est = estimation_method or TransformationEstimationPointToPlane()
    p2point = isinstance(est, TransformationEstimationPointToPoint)
    symmetric = isinstance(est, TransformationEstimationSymmetric)
    if symmetric:
        if source_normals is None or target_normals is None:
            raise ValueError("SymmetricICP requires both source and target to "
                             "have normals.")
        source_normals = require_cuda(source_normals, "source_normals")
    else:
        source_normals = None
    colored = isinstance(est, TransformationEstimationForColoredICP)
    attrs = _lib.IcpAttributes()
    attrs.lambda_geometric = getattr(est, "lambda_geometric", 0.968)
    if colored:
        if source_colors is None or target_colors is None:
            raise ValueError("Source and/or Target pointcloud missing colors "
                             "attribute.")
        source_colors = require_cuda(source_colors, "source_colors")
        target_colors = require_cuda(target_colors, "target_colors")
        attrs.source_colors = source_colors.data_ptr()
        attrs.target_colors = target_colors.data_ptr()
        if target_color_gradients is not None:
            target_color_gradients = require_cuda(target_color_gradients,
                                                  "target_color_gradients")
            attrs.target_color_gradients = target_color_gradients.data_ptr()
    if symmetric:
        attrs.source_normals = source_normals.data_ptr()
    source = require_cuda(source, "source")
    target = require_cuda(target, "target")
    if source.dtype not in (torch.float32, torch.float64):
        raise ValueError("Only Float32 and Float64 point clouds are supported.")
    if target.dtype != source.dtype:
        raise ValueError("source / target dtype mismatch")
    if p2point:
        target_normals = None
    else:
        if target_normals is None:
            raise ValueError("Target pointcloud missing normals attribute.")
        target_normals = require_cuda(target_normals, "target_normals")
        if target_normals.dtype != source.dtype:
            raise ValueError("source / target dtype mismatch")
    S = len(criteria_list)
    if not (len(voxel_sizes) == S and len(max_correspondence_distances) == S):
        raise ValueError("Size of criterias, voxel_size, "
                         "max_correspondence_distances vectors must be same.")
    vs = np.ascontiguousarray(voxel_sizes, dtype=np.float64)
    md = np.ascontiguousarray(max_correspondence_distances, dtype=np.float64)
    crit = (_lib.IcpCriteria * S)(*[
        _lib.IcpCriteria(c.relative_fitness, c.relative_rmse, c.max_iteration)
        for c in criteria_list])
    init = np.ascontiguousarray(
        np.eye(4) if init_source_to_target is None else init_source_to_target,
        dtype=np.float64)
    if init.shape != (4, 4):
        raise ValueError("init_source_to_target must be 4x4")
    ns, nt = source.shape[0], target.shape[0]
    corr = torch.full((ns,), -1, dtype=torch.int64, device="cuda")
    res = _lib.RegistrationResultC()

    cb = _lib.ICP_CALLBACK(0)
    if callback_after_iteration is not None:
        def _cb(it, sc, sit, rmse, fit, Tp, user):
            callback_after_iteration({
                "iteration_index": it, "scale_index": sc,
                "scale_iteration_index": sit, "inlier_rmse": rmse,
                "fitness": fit,
                "transformation": np.ctypeslib.as_array(
                    Tp, shape=(16,)).reshape(4, 4).copy()})
        cb = _lib.ICP_CALLBACK(_cb)
    ar = _lib.ALLREDUCE_SUM(0)
    if allreduce is not None:
        def _ar(buf, n, user):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            allreduce(a)
            return 0
        ar = _lib.ALLREDUCE_SUM(_ar)

    dar = None
    if device_allreduce is not None:
        def _dar(buf, n, strm, user):
            try:
                device_allreduce(int(buf or 0), int(n), int(strm or 0))
                return 0
            except Exception:  # surfaces as the driver's error status
                import traceback
                traceback.print_exc()
                return 1
        dar = _lib.ALLREDUCE_DEVICE(_dar)
        _lib.lib().o3dmi_set_device_allreduce(dar, None)
    try:
        if device_counts is not None:
            ns_dev, nt_dev = device_counts
            _lib.check(_lib.lib().o3dmi_registration_set_device_counts(
                _lib.ptr(ns_dev), _lib.ptr(nt_dev)), "set_device_counts")
        st = _icp_call(source, ns, target, target_normals, nt, S, vs, crit, md,
                       init, p2point, symmetric, colored, attrs, est, cb, ar,
                       corr, res)
    finally:
        if dar is not None:
            _lib.lib().o3dmi_set_device_allreduce(_lib.ALLREDUCE_DEVICE(0),
                                                  None)
        if device_counts is not None:
            # the driver consumes the pointers on entry; if Python raised
            # before it got there they must not wait for the NEXT call of this
            # thread
            _lib.lib().o3dmi_registration_set_device_counts(None, None)
    _lib.check(st, "multi_scale_icp")
    out = RegistrationResult()
    out.transformation = np.array(res.transformation[:]).reshape(4, 4)
    out.inlier_rmse = res.inlier_rmse
    out.fitness = res.fitness
    out.converged = bool(res.converged)
    out.num_iterations = res.num_iterations
    out.correspondence_set = corr[:res.num_correspondences]
    return out


def _icp_call(source, ns, target, target_normals, nt, S, vs, crit, md, init,
              p2point, symmetric, colored, attrs, est, cb, ar, corr, res):
    return _lib.lib().o3dmi_registration_multiscale_icp_ex(
        _lib.ptr(source), ns, _lib.ptr(target),
        _lib.ptr(target_normals) if target_normals is not None else None, nt,
        TORCH_TO_O3DMI[source.dtype], S, _lib.f64p(vs), crit, _lib.f64p(md),
        _lib.f64p(init),
        1 if p2point else (2 if symmetric else (3 if colored else 0)),
        C.byref(attrs), int(est.kernel.type),
        C.c_double(est.kernel.scaling_parameter),
        C.c_double(est.kernel.shape_parameter), cb, None, ar, None,
        _lib.ptr(corr), C.byref(res), stream())


def icp(source, target, target_normals, max_correspondence_distance,
        init_source_to_target=None, estimation_method=None, criteria=None,
        voxel_size=-1.0, callback_after_iteration=None, allreduce=None,
        source_normals=None, source_colors=None, target_colors=None,
        target_color_gradients=None, device_allreduce=None,
        device_counts=None):
    """t::pipelines::registration::ICP (Registration.cpp:93-106)."""
    return multi_scale_icp(source, target, target_normals, [voxel_size],
                           [criteria or ICPConvergenceCriteria()],
                           [max_correspondence_distance],
                           init_source_to_target, estimation_method,
                           callback_after_iteration, allreduce, source_normals,
                           source_colors, target_colors,
                           target_color_gradients, device_allreduce,
                           device_counts)


def _check_pair(source, target):
    source = require_cuda(source, "source")
    target = require_cuda(target, "target")
    if source.dtype not in (torch.float32, torch.float64):
        raise ValueError("Only Float32 and Float64 point clouds are supported.")
    if target.dtype != source.dtype:
        raise ValueError("source / target dtype mismatch")
    return source, target


def evaluate_registration(source, target, max_correspondence_distance,
                          transformation=None):
    """t::pipelines::registration::EvaluateRegistration
    (Registration.cpp:64-91)."""
    source, target = _check_pair(source, target)
    T = np.ascontiguousarray(
        np.eye(4) if transformation is None else transformation,
        dtype=np.float64)
    ns = source.shape[0]
    corr = torch.full((ns,), -1, dtype=torch.int64, device="cuda")
    res = _lib.RegistrationResultC()
    _lib.check(_lib.lib().o3dmi_registration_evaluate(
        _lib.ptr(source), ns, _lib.ptr(target), target.shape[0],
        TORCH_TO_O3DMI[source.dtype], C.c_double(max_correspondence_distance),
        _lib.f64p(T), _lib.ptr(corr), C.byref(res), stream()),
        "evaluate_registration")
    out = RegistrationResult()
    out.transformation = np.array(res.transformation[:]).reshape(4, 4)
    out.inlier_rmse = res.inlier_rmse
    out.fitness = res.fitness
    out.correspondence_set = corr
    return out


def get_information_matrix(source, target, max_correspondence_distance,
                           transformation=None):
    """t::pipelines::registration::GetInformationMatrix
    (Registration.cpp:446-486) -> {6,6} float64 (host)."""
    source, target = _check_pair(source, target)
    T = np.ascontiguousarray(
        np.eye(4) if transformation is None else transformation,
        dtype=np.float64)
    G = np.zeros((6, 6), np.float64)
    _lib.check(_lib.lib().o3dmi_registration_information_matrix(
        _lib.ptr(source), source.shape[0], _lib.ptr(target), target.shape[0],
        TORCH_TO_O3DMI[source.dtype], C.c_double(max_correspondence_distance),
        _lib.f64p(T), _lib.f64p(G), stream()), "get_information_matrix")
    return G


def voxel_down_sample(positions, normals, voxel_size):
    """t::geometry::PointCloud::VoxelDownSample (PointCloud.cpp:496-567) for
    positions (+ optional normals): -> (positions {M,3}, normals {M,3}|None),
    voxels in order of first occurrence."""
    positions = require_cuda(positions, "positions")
    n = positions.shape[0]
    out_p = torch.empty_like(positions)
    out_n = None
    if normals is not None:
        normals = require_cuda(normals, "normals")
        out_n = torch.empty_like(normals)
    m = C.c_int64(0)
    _lib.check(_lib.lib().o3dmi_voxel_down_sample(
        _lib.ptr(positions), _lib.ptr(normals), n,
        TORCH_TO_O3DMI[positions.dtype], C.c_double(voxel_size),
        _lib.ptr(out_p), _lib.ptr(out_n), C.byref(m), stream()),
        "voxel_down_sample")
    return out_p[:m.value], (None if out_n is None else out_n[:m.value])


def estimate_normals(positions, max_nn=30, radius=None, normals=None):
    """t::geometry::PointCloud::EstimateNormals(max_nn, radius)
    (PointCloud.cpp:856-976): hybrid search when both are given, KNN search
    when radius is None (the reference's default), radius search when max_nn
    is None; returns normals {N,3};
    `normals` (optional) are existing normals whose orientation is kept."""
    positions = require_cuda(positions, "positions")
    if radius is None and max_nn is None:
        raise ValueError("Both max_nn and radius are none.")
    if max_nn is None:
        max_nn = -1      # radius search: every neighbour within radius
    if radius is None:
        radius = -1.0    # KNN search
    if normals is None:
        out = torch.empty_like(positions)
        has = 0
    else:
        out = require_cuda(normals, "normals").clone()
        has = 1
    _lib.check(_lib.lib().o3dmi_pointcloud_estimate_normals(
        _lib.ptr(positions), positions.shape[0],
        TORCH_TO_O3DMI[positions.dtype], int(max_nn), C.c_double(radius),
        _lib.ptr(out), has, stream()), "estimate_normals")
    return out


def knn_search(points, queries, knn):
    """core::nns::NearestNeighborSearch(points).KnnIndex() + KnnSearch(queries,
    knn) -> (indices {Q,k} int32, squared distances {Q,k}), k = min(knn, N)."""
    points = require_cuda(points, "points")
    queries = require_cuda(queries, "queries")
    if queries.dtype != points.dtype:
        raise ValueError("points / queries dtype mismatch")
    n, q = points.shape[0], queries.shape[0]
    k = min(int(knn), n)
    ka = max(k, 1)  # knn <= 0 is rejected by the library, as in the reference
    idx = torch.empty((q, ka), dtype=torch.int32, device="cuda")
    d2 = torch.empty((q, ka), dtype=points.dtype, device="cuda")
    _lib.check(_lib.lib().o3dmi_nns_knn_search(
        _lib.ptr(points), n, _lib.ptr(queries), q,
        TORCH_TO_O3DMI[points.dtype], int(knn), _lib.ptr(idx), _lib.ptr(d2),
        stream()), "knn_search")
    return idx, d2


def estimate_color_gradients(positions, normals, colors, max_nn=30,
                             radius=None):
    """t::geometry::PointCloud::EstimateColorGradients(max_nn, radius)
    (PointCloud.cpp:987-1060) -> gradients {N,3}."""
    positions = require_cuda(positions, "positions")
    normals = require_cuda(normals, "normals")
    colors = require_cuda(colors, "colors")
    if max_nn is None and radius is None:
        raise ValueError("Both max_nn and radius are none.")
    if max_nn is None:
        max_nn = -1      # radius search
    out = torch.empty_like(positions)
    _lib.check(_lib.lib().o3dmi_pointcloud_estimate_color_gradients(
        _lib.ptr(positions), _lib.ptr(normals), _lib.ptr(colors),
        positions.shape[0], TORCH_TO_O3DMI[positions.dtype], int(max_nn),
        C.c_double(-1.0 if radius is None else radius), _lib.ptr(out),
        stream()), "estimate_color_gradients")
    return out


def compute_rmse(estimation_method, source, target, target_normals,
                 correspondences, source_normals=None, source_colors=None,
                 target_colors=None, target_color_gradients=None):
    """TransformationEstimation*::ComputeRMSE on device tensors (the
    reference's definitions, see o3dmi_registration_compute_rmse)."""
    est = estimation_method
    code = (1 if isinstance(est, TransformationEstimationPointToPoint) else
            2 if isinstance(est, TransformationEstimationSymmetric) else
            3 if isinstance(est, TransformationEstimationForColoredICP) else 0)
    source, target = _check_pair(source, target)
    corr = require_cuda(correspondences, "correspondences")
    attrs = _lib.IcpAttributes()
    attrs.lambda_geometric = getattr(est, "lambda_geometric", 0.968)
    keep = [require_cuda(t, "attribute") for t in
            (source_normals, source_colors, target_colors,
             target_color_gradients, target_normals) if t is not None]
    if source_normals is not None:
        attrs.source_normals = source_normals.data_ptr()
    if source_colors is not None:
        attrs.source_colors = source_colors.data_ptr()
    if target_colors is not None:
        attrs.target_colors = target_colors.data_ptr()
    if target_color_gradients is not None:
        attrs.target_color_gradients = target_color_gradients.data_ptr()
    out = C.c_double(0)
    _lib.check(_lib.lib().o3dmi_registration_compute_rmse(
        code, _lib.ptr(source), source.shape[0], _lib.ptr(target),
        _lib.ptr(target_normals), TORCH_TO_O3DMI[source.dtype],
        C.byref(attrs), _lib.ptr(corr), C.byref(out), stream()),
        "compute_rmse")
    del keep
    return out.value


def fixed_radius_search(points, queries, radius):
    """core::nns::NearestNeighborSearch(points).FixedRadiusIndex(radius) +
    FixedRadiusSearch(queries, radius) -> (indices {total} int32, squared
    distances {total}, neighbors_row_splits {Q+1} int64), neighbours of a
    query ascending by (distance, index)."""
    points = require_cuda(points, "points")
    queries = require_cuda(queries, "queries")
    if queries.dtype != points.dtype:
        raise ValueError("points / queries dtype mismatch")
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.o3dmi_nns_create(_lib.ptr(points), points.shape[0],
                                  TORCH_TO_O3DMI[points.dtype],
                                  C.c_double(radius), stream(), C.byref(h)),
               "nns_create")
    try:
        q = queries.shape[0]
        counts = torch.zeros(q, dtype=torch.int32, device="cuda")
        _lib.check(L.o3dmi_nns_radius_count(h, _lib.ptr(queries), q,
                                            _lib.ptr(counts), stream()),
                   "radius_count")
        splits = torch.zeros(q + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(counts, 0, out=splits[1:])
        total = int(splits[-1].item())
        idx = torch.empty(max(total, 1), dtype=torch.int32, device="cuda")
        d2 = torch.empty(max(total, 1), dtype=points.dtype, device="cuda")
        _lib.check(L.o3dmi_nns_radius_search(h, _lib.ptr(queries), q,
                                             _lib.ptr(splits), _lib.ptr(idx),
                                             _lib.ptr(d2), stream()),
                   "radius_search")
        torch.cuda.synchronize()
        return idx[:total], d2[:total], splits
    finally:
        L.o3dmi_nns_destroy(h)
