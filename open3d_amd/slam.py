"""Python mirror of open3d.t.pipelines.slam.{Frame, Model} for the MI355X
backend (cpp/open3d/t/pipelines/slam/Frame.h, Model.h / Model.cpp; binding
cpp/pybind/t/pipelines/slam/slam.cpp). The model itself (voxel grid, current
pose, frustum blocks) lives in the native library (o3dmi_slam_model_*); Frame
is the reference's plain container of per-frame device tensors.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, geometry
from .core import TORCH_TO_O3DMI, host_mat, require_cuda, stream
from .odometry import (Method, OdometryConvergenceCriteria, OdometryResult)


class Frame:
    """slam::Frame (Frame.h:20-77): height, width, intrinsics and a name ->
    tensor map ("depth", "color", ...)."""

    def __init__(self, height, width, intrinsics, device="cuda:0"):
        self._h, self._w = int(height), int(width)
        self._K = host_mat(intrinsics, (3, 3), "intrinsics")
        self._data = {}

    def height(self):
        return self._h

    def width(self):
        return self._w

    def set_intrinsics(self, intrinsics):
        self._K = host_mat(intrinsics, (3, 3), "intrinsics")

    def get_intrinsics(self):
        return self._K

    def set_data(self, name, data):
        self._data[name] = data.cuda() if not data.is_cuda else data

    def get_data(self, name):
        # "Property not found for {}, return an empty tensor!"
        return self._data.get(name, torch.empty(0))

    set_data_from_image = set_data
    get_data_as_image = get_data


class _BorrowedGrid(geometry.VoxelBlockGrid):
    """VoxelBlockGrid whose native handle belongs to a Model."""

    def __del__(self):
        pass


class Model:
    def __init__(self, voxel_size, block_resolution=16, block_count=1000,
                 transformation=None, device="cuda:0"):
        T = host_mat(np.eye(4) if transformation is None else transformation,
                     (4, 4), "T_init")
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_slam_model_create(
            C.c_float(voxel_size), int(block_resolution), int(block_count),
            _lib.f64p(T), stream(), C.byref(h)), "Model")
        self._m = h
        # non-owning VoxelBlockGrid view of the model's grid (valid while
        # the model lives)
        g = _BorrowedGrid.__new__(_BorrowedGrid)
        g._g = C.c_void_p(_lib.lib().o3dmi_slam_model_voxel_grid(h))
        g.voxel_size = float(voxel_size)
        g.block_resolution = int(block_resolution)
        g.attr_names = ["tsdf", "weight", "color"]
        g._chans = {"tsdf": 1, "weight": 1, "color": 3}
        self.voxel_grid = g

    def __del__(self):
        try:
            if getattr(self, "_m", None):
                _lib.lib().o3dmi_slam_model_destroy(self._m)
                self._m = None
        except Exception:
            pass

    def get_hashmap(self):
        return self.voxel_grid.hashmap()

    def get_current_frame_pose(self):
        T = np.zeros((4, 4), np.float64)
        _lib.check(_lib.lib().o3dmi_slam_model_get_current_frame_pose(
            self._m, _lib.f64p(T)), "get_current_frame_pose")
        return T

    def update_frame_pose(self, frame_id, T_frame_to_world):
        T = host_mat(T_frame_to_world, (4, 4), "T_frame_to_world")
        _lib.check(_lib.lib().o3dmi_slam_model_update_frame_pose(
            self._m, int(frame_id), _lib.f64p(T)), "update_frame_pose")

    @property
    def frame_id(self):
        return int(_lib.lib().o3dmi_slam_model_frame_id(self._m))

    @property
    def frustum_block_coords(self):
        n = int(_lib.lib().o3dmi_slam_model_frustum_block_count(self._m))
        p = _lib.lib().o3dmi_slam_model_frustum_block_coords(self._m)
        if n == 0 or not p:
            return torch.empty((0, 3), dtype=torch.int32, device="cuda")
        from .core import tensor_from_ptr
        return tensor_from_ptr(p, (n, 3), _lib.I32, self)

    def synthesize_model_frame(self, raycast_frame, depth_scale=1000.0,
                               depth_min=0.1, depth_max=3.0,
                               trunc_voxel_multiplier=8.0, enable_color=True,
                               weight_threshold=-1.0):
        h, w = raycast_frame.height(), raycast_frame.width()
        depth = torch.empty((h, w, 1), dtype=torch.float32, device="cuda")
        # The reference always renders {"depth", "color"} and drops the colour
        # when !enable_color (Model.cpp:49-67); rendering it is skipped here.
        color = torch.empty((h, w, 3), dtype=torch.float32, device="cuda") \
            if enable_color else None
        _lib.check(_lib.lib().o3dmi_slam_model_synthesize_model_frame(
            self._m, _lib.f64p(raycast_frame.get_intrinsics()), w, h,
            C.c_float(depth_scale), C.c_float(depth_min), C.c_float(depth_max),
            C.c_float(trunc_voxel_multiplier), C.c_float(weight_threshold),
            _lib.ptr(depth), _lib.ptr(color), stream()),
            "synthesize_model_frame")
        raycast_frame.set_data("depth", depth)
        if enable_color:
            raycast_frame.set_data("color", color)
        elif raycast_frame.get_data("color").numel() == 0:
            # dummy RGB frame so that RGB-D odometry can run (Model.cpp:60-66)
            raycast_frame.set_data("color", torch.zeros(
                (h, w, 3), dtype=torch.float32, device="cuda"))

    def track_frame_to_model(self, input_frame, raycast_frame,
                             depth_scale=1000.0, depth_max=3.0,
                             depth_diff=0.07, method=Method.PointToPlane,
                             criteria=(6, 3, 1)):
        d = require_cuda(input_frame.get_data("depth"), "input depth")
        c = input_frame.get_data("color")
        c = require_cuda(c, "input color") if c.numel() else None
        rd = require_cuda(raycast_frame.get_data("depth"), "raycast depth")
        rc = raycast_frame.get_data("color")
        rc = require_cuda(rc, "raycast color") if rc.numel() else None
        if rd.dtype != torch.float32 or (rc is not None and
                                         rc.dtype != torch.float32):
            raise ValueError("ray-cast frame must be Float32")
        if method != Method.PointToPlane and (c is None or rc is None):
            raise ValueError("intensity / hybrid tracking needs colour")
        rows, cols = d.shape[0], d.shape[1]
        crit = [x if isinstance(x, OdometryConvergenceCriteria)
                else OdometryConvergenceCriteria(int(x)) for x in criteria]
        cc = (_lib.OdometryCriteriaC * len(crit))(*[
            _lib.OdometryCriteriaC(x.max_iteration, x.relative_rmse,
                                   x.relative_fitness) for x in crit])
        res = _lib.OdometryResultC()
        _lib.check(_lib.lib().o3dmi_slam_model_track_frame_to_model(
            self._m, _lib.ptr(d), TORCH_TO_O3DMI[d.dtype], _lib.ptr(c),
            TORCH_TO_O3DMI[c.dtype] if c is not None else _lib.U8,
            _lib.ptr(rd), _lib.ptr(rc), rows, cols,
            _lib.f64p(raycast_frame.get_intrinsics()), C.c_float(depth_scale),
            C.c_float(depth_max), C.c_float(depth_diff), int(method),
            len(crit), cc, C.byref(res), stream()), "track_frame_to_model")
        out = OdometryResult(np.array(res.transformation[:]).reshape(4, 4),
                             res.inlier_rmse, res.fitness)
        out.num_iterations = res.num_iterations
        return out

    def integrate(self, input_frame, depth_scale=1000.0, depth_max=3.0,
                  trunc_voxel_multiplier=8.0):
        d = require_cuda(input_frame.get_data("depth"), "input depth")
        c = input_frame.get_data("color")
        c = require_cuda(c, "input color") if c.numel() else None
        _lib.check(_lib.lib().o3dmi_slam_model_integrate(
            self._m, _lib.ptr(d), TORCH_TO_O3DMI[d.dtype], _lib.ptr(c),
            d.shape[0], d.shape[1], _lib.f64p(input_frame.get_intrinsics()),
            C.c_float(depth_scale), C.c_float(depth_max),
            C.c_float(trunc_voxel_multiplier), stream()), "Model.integrate")

    def extract_pointcloud(self, weight_threshold=3.0, estimated_number=-1):
        return geometry._extract(
            lambda *a: _lib.lib().o3dmi_slam_model_extract_point_cloud(
                self._m, *a), True, weight_threshold, estimated_number)
