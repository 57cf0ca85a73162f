"""Python mirror of open3d.t.geometry.VoxelBlockGrid for the MI355X backend.

Same constructor and method names / defaults as the reference's binding
(cpp/pybind/t/geometry/voxel_block_grid.cpp:31-178); tensors are torch device
tensors (depth {H,W} or {H,W,1} uint16|float32, colour {H,W,3} uint8|float32,
block_coords {M,3} int32), intrinsic / extrinsic are Float64 host matrices.
All compute goes through libo3d_mi355x.so (no fallback).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .core import (O3DMI_TO_TORCH, TORCH_TO_O3DMI, host_mat, require_cuda,
                   stream, tensor_from_ptr)


class HashMapView:
    """Read-only view of the grid's block hash map (o3d.core.HashMap subset)."""

    def __init__(self, handle, owner):
        self._h = handle
        self._owner = owner

    def size(self):
        n = C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_hash_size(self._h, stream(), C.byref(n)),
                   "HashMap.size")
        return int(n.value)

    def capacity(self):
        return int(_lib.lib().o3dmi_hash_capacity(self._h))

    def key_tensor(self):
        p = _lib.lib().o3dmi_hash_key_buffer(self._h)
        return tensor_from_ptr(p, (self.capacity(), 3), _lib.I32, self._owner)

    def active_buf_indices(self):
        out = torch.empty(self.capacity(), dtype=torch.int32, device="cuda")
        n = C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_hash_active_indices(
            self._h, _lib.ptr(out), stream(), C.byref(n)),
            "HashMap.active_buf_indices")
        return out[:n.value]

    def find(self, keys):
        keys = require_cuda(keys, "keys")
        n = keys.shape[0]
        buf = torch.empty(n, dtype=torch.int32, device="cuda")
        masks = torch.empty(n, dtype=torch.bool, device="cuda")
        _lib.check(_lib.lib().o3dmi_hash_find(
            self._h, _lib.ptr(keys), n, None, _lib.ptr(buf), _lib.ptr(masks),
            stream()), "HashMap.find")
        return buf, masks


def _image(t, name, channels):
    t = require_cuda(t, name)
    if t.dim() == 3 and t.shape[2] == 1 and channels == 1:
        t = t[:, :, 0]
    if channels == 1 and t.dim() != 2:
        raise ValueError("%s must be {H,W} or {H,W,1}" % name)
    if channels == 3 and (t.dim() != 3 or t.shape[2] != 3):
        raise ValueError("%s must be {H,W,3}" % name)
    return t


def _extract(call, has_color, weight_threshold, estimated_point_number):
    total = C.c_int64(0)
    cap = int(estimated_point_number)
    if cap < 0:
        _lib.check(call(C.c_float(weight_threshold), C.c_int64(-1), None, None,
                        None, C.byref(total), stream()), "extract_point_cloud")
        cap = int(total.value)
    pts = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
    nrm = torch.empty((cap, 3), dtype=torch.float32, device="cuda")
    col = torch.empty((cap, 3), dtype=torch.float32, device="cuda") \
        if has_color else None
    _lib.check(call(C.c_float(weight_threshold), C.c_int64(cap), _lib.ptr(pts),
                    _lib.ptr(nrm), _lib.ptr(col), C.byref(total), stream()),
               "extract_point_cloud")
    m = min(cap, int(total.value))
    out = {"positions": pts[:m], "normals": nrm[:m]}
    if col is not None:
        out["colors"] = col[:m]
    return out


class FrameBatch:
    """Prepared arguments of VoxelBlockGrid.integrate_frames (prepare_frames)."""
    pass


class VoxelBlockGrid:
    _CHANNELS = {"vertex": 3, "normal": 3, "depth": 1, "color": 3, "index": 8,
                 "mask": 8, "interp_ratio": 8, "interp_ratio_dx": 8,
                 "interp_ratio_dy": 8, "interp_ratio_dz": 8}

    def __init__(self, attr_names, attr_dtypes, attr_channels,
                 voxel_size=0.0058, block_resolution=16, block_count=10000,
                 device="cuda:0"):
        if len(attr_dtypes) != len(attr_names):
            raise ValueError("Number of attribute dtypes (%d) mismatch with "
                             "names (%d)." % (len(attr_dtypes),
                                              len(attr_names)))
        if len(attr_channels) != len(attr_names):
            raise ValueError("Number of attribute channels (%d) mismatch with "
                             "names (%d)." % (len(attr_channels),
                                              len(attr_names)))
        self.voxel_size = float(voxel_size)
        self.block_resolution = int(block_resolution)
        self.attr_names = list(attr_names)
        n = len(attr_names)
        names = (C.c_char_p * n)(*[s.encode() for s in attr_names])
        dts = (C.c_int * n)(*[TORCH_TO_O3DMI[d] for d in attr_dtypes])
        chans = []
        for c in attr_channels:
            c = tuple(c) if isinstance(c, (tuple, list)) else (c,)
            chans.append(int(np.prod(c)))
        chs = (C.c_int * n)(*chans)
        self._chans = dict(zip(attr_names, chans))
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_vbg_create(
            n, names, dts, chs, C.c_float(voxel_size), int(block_resolution),
            int(block_count), stream(), C.byref(h)), "VoxelBlockGrid")
        self._g = h

    def __del__(self):
        try:
            if getattr(self, "_g", None):
                _lib.lib().o3dmi_vbg_destroy(self._g)
                self._g = None
        except Exception:
            pass

    def set_block_ownership(self, rank, world):
        """Multi-GPU block-ownership sharding: this grid only activates and
        integrates the blocks whose owner (sharding.block_owner) is `rank`."""
        _lib.check(_lib.lib().o3dmi_vbg_set_block_ownership(
            self._g, int(rank), int(world)), "set_block_ownership")

    def hashmap(self):
        return HashMapView(C.c_void_p(_lib.lib().o3dmi_vbg_hashmap(self._g)),
                           self)

    def attribute(self, attribute_name):
        dt, ch = C.c_int(0), C.c_int(0)
        p = _lib.lib().o3dmi_vbg_attribute(self._g, attribute_name.encode(),
                                           C.byref(dt), C.byref(ch))
        if not p:
            return torch.empty(0)
        r = self.block_resolution
        return tensor_from_ptr(p, (self.hashmap().capacity(), r, r, r,
                                   ch.value), dt.value, self)

    def compute_unique_block_coordinates(self, depth, intrinsic, extrinsic,
                                         depth_scale=1000.0, depth_max=3.0,
                                         trunc_voxel_multiplier=8.0):
        depth = _image(depth, "depth", 1)
        K = host_mat(intrinsic, (3, 3), "intrinsic")
        T = host_mat(extrinsic, (4, 4), "extrinsic")
        rows, cols = depth.shape
        cap = (rows // 4) * (cols // 4) * 4
        out = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        m = C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_vbg_get_unique_block_coordinates(
            self._g, _lib.ptr(depth), TORCH_TO_O3DMI[depth.dtype], rows, cols,
            _lib.f64p(K), _lib.f64p(T), C.c_float(depth_scale),
            C.c_float(depth_max), C.c_float(trunc_voxel_multiplier),
            _lib.ptr(out), C.byref(m), stream()),
            "VoxelBlockGrid.compute_unique_block_coordinates")
        return out[:m.value]

    def compute_unique_block_coordinates_pcd(self, points,
                                             trunc_voxel_multiplier=8.0):
        """GetUniqueBlockCoordinates(pcd, trunc_voxel_multiplier)
        (VoxelBlockGrid.cpp:246-267): the blocks within the truncation
        distance of a point cloud {n, 3} Float32."""
        if not (points.is_cuda and points.dtype == torch.float32 and
                points.dim() == 2 and points.shape[1] == 3):
            raise ValueError("points must be a CUDA Float32 {n, 3} tensor")
        points = points.contiguous()
        n = points.shape[0]
        out = torch.empty((max(1, n * 8), 3), dtype=torch.int32, device="cuda")
        m = C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_vbg_get_unique_block_coordinates_pcd(
            self._g, _lib.ptr(points), n, C.c_float(trunc_voxel_multiplier),
            _lib.ptr(out), out.shape[0], C.byref(m), stream()),
            "VoxelBlockGrid.compute_unique_block_coordinates_pcd")
        return out[:m.value]

    def _buf_indices(self, buf_indices):
        if buf_indices is None:        # upstream: GetActiveIndices()
            buf_indices = self.hashmap().active_buf_indices()
        if not (buf_indices.is_cuda and buf_indices.dtype == torch.int32 and
                buf_indices.dim() == 1):
            raise ValueError("buf_indices must be a CUDA Int32 vector")
        return buf_indices.contiguous()

    def voxel_indices(self, buf_indices=None):
        """GetVoxelIndices (VoxelBlockGrid.cpp:145-183): {4, n * res^3}
        Int64 -- buffer index, x, y, z of every voxel of the blocks."""
        b = self._buf_indices(buf_indices)
        res = int(_lib.lib().o3dmi_vbg_block_resolution(self._g))
        out = torch.empty((4, b.shape[0] * res ** 3), dtype=torch.int64,
                          device="cuda")
        _lib.check(_lib.lib().o3dmi_vbg_get_voxel_indices(
            self._g, _lib.ptr(b), b.shape[0], _lib.ptr(out), stream()),
            "VoxelBlockGrid.voxel_indices")
        return out

    def voxel_coordinates(self, voxel_indices):
        """GetVoxelCoordinates (VoxelBlockGrid.cpp:130-143): {4, n} voxel
        indices -> {3, n} Int64 voxel coordinates."""
        if not (voxel_indices.is_cuda and voxel_indices.dtype == torch.int64
                and voxel_indices.dim() == 2 and voxel_indices.shape[0] == 4):
            raise ValueError("voxel_indices must be a CUDA Int64 {4, n} tensor")
        v = voxel_indices.contiguous()
        out = torch.empty((3, v.shape[1]), dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().o3dmi_vbg_get_voxel_coordinates(
            self._g, _lib.ptr(v), v.shape[1], _lib.ptr(out), stream()),
            "VoxelBlockGrid.voxel_coordinates")
        return out

    def voxel_coordinates_and_flattened_indices(self, buf_indices=None):
        """GetVoxelCoordinatesAndFlattenedIndices (VoxelBlockGrid.cpp:185-211):
        ({n * res^3, 3} Float32 world coordinates, {n * res^3} Int64 linear
        indices into the value tensors)."""
        b = self._buf_indices(buf_indices)
        res = int(_lib.lib().o3dmi_vbg_block_resolution(self._g))
        nv = b.shape[0] * res ** 3
        coords = torch.empty((nv, 3), dtype=torch.float32, device="cuda")
        flat = torch.empty((nv,), dtype=torch.int64, device="cuda")
        _lib.check(
            _lib.lib().o3dmi_vbg_get_voxel_coordinates_and_flattened_indices(
                self._g, _lib.ptr(b), b.shape[0], _lib.ptr(coords),
                _lib.ptr(flat), stream()),
            "VoxelBlockGrid.voxel_coordinates_and_flattened_indices")
        return coords, flat

    def integrate(self, block_coords, depth, color=None, depth_intrinsic=None,
                  color_intrinsic=None, extrinsic=None, depth_scale=1000.0,
                  depth_max=3.0, trunc_voxel_multiplier=8.0):
        block_coords = require_cuda(block_coords, "block_coords")
        if block_coords.dtype != torch.int32:
            raise ValueError("Unsupported block coordinate dtype %s"
                             % block_coords.dtype)
        depth = _image(depth, "depth", 1)
        if color_intrinsic is None:
            color_intrinsic = depth_intrinsic
        Kd = host_mat(depth_intrinsic, (3, 3), "intrinsic")
        Kc = host_mat(color_intrinsic, (3, 3), "intrinsic")
        T = host_mat(extrinsic, (4, 4), "extrinsic")
        crows = ccols = 0
        if color is not None and color.numel() > 0:
            color = _image(color, "color", 3)
            crows, ccols = color.shape[:2]
            want = torch.float32 if depth.dtype == torch.float32 \
                else torch.uint8
            if color.dtype != want:
                raise ValueError(
                    "Unsupported input data type combination. Expected "
                    "(float, float) or (uint16, uint8), but received (%s %s)"
                    % (depth.dtype, color.dtype))
        else:
            color = None
        _lib.check(_lib.lib().o3dmi_vbg_integrate_blocks(
            self._g, _lib.ptr(block_coords), block_coords.shape[0],
            _lib.ptr(depth), depth.shape[0], depth.shape[1], _lib.ptr(color),
            crows, ccols, TORCH_TO_O3DMI[depth.dtype], _lib.f64p(Kd),
            _lib.f64p(Kc), _lib.f64p(T), C.c_float(depth_scale),
            C.c_float(depth_max), C.c_float(trunc_voxel_multiplier),
            stream()), "VoxelBlockGrid.integrate")

    def integrate_frame(self, depth, color, depth_intrinsic, color_intrinsic,
                        extrinsic, depth_scale=1000.0, depth_max=3.0,
                        trunc_voxel_multiplier=8.0):
        """compute_unique_block_coordinates + integrate of one frame with all
        counts device-resident (frame-stream fast path; same results)."""
        depth = _image(depth, "depth", 1)
        if color_intrinsic is None:
            color_intrinsic = depth_intrinsic
        Kd = host_mat(depth_intrinsic, (3, 3), "intrinsic")
        Kc = host_mat(color_intrinsic, (3, 3), "intrinsic")
        T = host_mat(extrinsic, (4, 4), "extrinsic")
        crows = ccols = 0
        if color is not None and color.numel() > 0:
            color = _image(color, "color", 3)
            crows, ccols = color.shape[:2]
        else:
            color = None
        _lib.check(_lib.lib().o3dmi_vbg_integrate_frame(
            self._g, _lib.ptr(depth), depth.shape[0], depth.shape[1],
            _lib.ptr(color), crows, ccols, TORCH_TO_O3DMI[depth.dtype],
            _lib.f64p(Kd), _lib.f64p(Kc), _lib.f64p(T), C.c_float(depth_scale),
            C.c_float(depth_max), C.c_float(trunc_voxel_multiplier),
            stream()), "VoxelBlockGrid.integrate_frame")

    def prepare_frames(self, depths, colors, depth_intrinsic,
                       color_intrinsic, extrinsics):
        """The argument block of integrate_frames for a list of resident
        frames -- checked once, pointer / pose arrays built once -- for callers
        that integrate the same frames repeatedly (a looped stream): returns a
        FrameBatch to pass as `depths`."""
        n = len(depths)
        ds = [_image(d, "depth", 1) for d in depths]
        b = FrameBatch()
        b.n = n
        b.keep = [ds]
        b.rows, b.cols = ds[0].shape
        if color_intrinsic is None:
            color_intrinsic = depth_intrinsic
        b.Kd = host_mat(depth_intrinsic, (3, 3), "intrinsic")
        b.Kc = host_mat(color_intrinsic, (3, 3), "intrinsic")
        b.Ts = np.ascontiguousarray(
            np.stack([host_mat(T, (4, 4), "extrinsic") for T in extrinsics]),
            dtype=np.float64)
        b.dptr = (C.c_void_p * n)(*[d.data_ptr() for d in ds])
        b.cptr, b.crows, b.ccols = None, 0, 0
        if colors is not None:
            cs = [_image(c, "color", 3) for c in colors]
            b.keep.append(cs)
            b.crows, b.ccols = cs[0].shape[:2]
            b.cptr = (C.c_void_p * n)(*[c.data_ptr() for c in cs])
        b.dtype = TORCH_TO_O3DMI[ds[0].dtype]
        return b

    def integrate_frames(self, depths, colors=None, depth_intrinsic=None,
                         color_intrinsic=None, extrinsics=None,
                         depth_scale=1000.0, depth_max=3.0,
                         trunc_voxel_multiplier=8.0, frames_per_launch=0):
        """integrate_frame over a list of frames (same intrinsics / sizes),
        strictly in order, in one native call. frames_per_launch (1..16, 0 =
        8) frames are applied per launch to each touched block while its
        voxels stay in registers; results are identical for every value.
        `depths` may be a FrameBatch from prepare_frames (the other frame
        arguments are then taken from it)."""
        if isinstance(depths, FrameBatch):
            b = depths
        else:
            if len(depths) == 0:
                return
            b = self.prepare_frames(depths, colors, depth_intrinsic,
                                    color_intrinsic, extrinsics)
        if b.n == 0:
            return
        _lib.check(_lib.lib().o3dmi_vbg_integrate_frames(
            self._g, b.n, b.dptr, b.rows, b.cols, b.cptr, b.crows, b.ccols,
            b.dtype, _lib.f64p(b.Kd), _lib.f64p(b.Kc), _lib.f64p(b.Ts),
            C.c_float(depth_scale), C.c_float(depth_max),
            C.c_float(trunc_voxel_multiplier), int(frames_per_launch), stream()),
            "VoxelBlockGrid.integrate_frames")

    # ---- sliced block touch (block-ownership sharding; include/
    # o3d_mi355x_host.h "SLICED block touch") --------------------------------

    @staticmethod
    def slice_chunk_frames(frames_per_launch):
        return int(_lib.lib().o3dmi_vbg_slice_chunk_frames(
            int(frames_per_launch)))

    def set_slice_capacity(self, records_per_group, table_slots):
        _lib.check(_lib.lib().o3dmi_vbg_set_slice_capacity(
            self._g, int(records_per_group), int(table_slots)),
            "VoxelBlockGrid.set_slice_capacity")

    def slice_segment_bytes(self):
        return int(_lib.lib().o3dmi_vbg_slice_segment_bytes(self._g))

    def touch_slice(self, batch, lo, n, slice_rank, slice_world,
                    depth_scale=1000.0, depth_max=3.0,
                    trunc_voxel_multiplier=8.0, frames_per_launch=0,
                    out=None):
        """Wire segment (uint8 device tensor) of rank `slice_rank`'s band of
        ray tiles for frames [lo, lo + n) of a FrameBatch (n <= one chunk)."""
        if out is None:
            out = torch.empty(self.slice_segment_bytes(), dtype=torch.uint8,
                              device="cuda")
        dptr = (C.c_void_p * n)(*[batch.dptr[lo + i] for i in range(n)])
        Ts = np.ascontiguousarray(batch.Ts[lo:lo + n])
        _lib.check(_lib.lib().o3dmi_vbg_touch_slice(
            self._g, n, dptr, batch.rows, batch.cols, _lib.f64p(batch.Kd),
            _lib.f64p(Ts), C.c_float(depth_scale), C.c_float(depth_max),
            C.c_float(trunc_voxel_multiplier), int(frames_per_launch),
            int(slice_rank), int(slice_world), _lib.ptr(out), stream()),
            "VoxelBlockGrid.touch_slice")
        return out

    def gather_slices(self, batch, world, depth_scale=1000.0, depth_max=3.0,
                      trunc_voxel_multiplier=8.0, frames_per_launch=0):
        """What the all-gather of the sliced path delivers, computed on ONE
        device (tests, bench.py --emulate-world): per chunk of the batch a
        uint8 tensor holding the wire segments of ranks 0 .. world - 1."""
        cf = self.slice_chunk_frames(frames_per_launch)
        seg = self.slice_segment_bytes()
        out = []
        for lo in range(0, batch.n, cf):
            n = min(cf, batch.n - lo)
            buf = torch.empty(world * seg, dtype=torch.uint8, device="cuda")
            for r in range(world):
                self.touch_slice(batch, lo, n, r, world, depth_scale,
                                 depth_max, trunc_voxel_multiplier,
                                 frames_per_launch,
                                 out=buf[r * seg:(r + 1) * seg])
            out.append(buf)
        return out

    def integrate_frames_sliced(self, batch, gathered=None, depth_scale=1000.0,
                                depth_max=3.0, trunc_voxel_multiplier=8.0,
                                frames_per_launch=0):
        """integrate_frames through the sliced path. `gathered`: list of per-
        chunk tensors from gather_slices (or an exchange of one's own), or
        None: all-gather over the communicator installed on this thread."""
        gp = None
        if gathered is not None:
            gp = (C.c_void_p * len(gathered))(
                *[t.data_ptr() for t in gathered])
        _lib.check(_lib.lib().o3dmi_vbg_integrate_frames_sliced(
            self._g, batch.n, batch.dptr, batch.rows, batch.cols, batch.cptr,
            batch.crows, batch.ccols, batch.dtype, _lib.f64p(batch.Kd),
            _lib.f64p(batch.Kc), _lib.f64p(batch.Ts), C.c_float(depth_scale),
            C.c_float(depth_max), C.c_float(trunc_voxel_multiplier),
            int(frames_per_launch), gp, stream()),
            "VoxelBlockGrid.integrate_frames_sliced")

    def sliced_stats(self):
        ch, re = C.c_int64(0), C.c_int64(0)
        cap, sl = C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.lib().o3dmi_vbg_sliced_stats(
            self._g, C.byref(ch), C.byref(re), C.byref(cap), C.byref(sl)),
            "sliced_stats")
        return dict(chunks=ch.value, reapplied=re.value, capacity=cap.value,
                    table_slots=sl.value)

    def _attr_layout(self):
        out = []
        for nm in self.attr_names:
            dt, ch = C.c_int(0), C.c_int(0)
            _lib.lib().o3dmi_vbg_attribute(self._g, nm.encode(), C.byref(dt),
                                           C.byref(ch))
            out.append((O3DMI_TO_TORCH[dt.value], ch.value))
        return out

    def export_blocks(self):
        """The active blocks in ascending buffer index (the order of save):
        keys {n,3} int32 and one {n,res,res,res,C} tensor per attribute, on the
        device. The payload of the frame-sharded merge (sharding.py)."""
        n = C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_vbg_export_blocks(
            self._g, 0, None, None, C.byref(n), stream()), "export_blocks")
        n = n.value
        r = self.block_resolution
        keys = torch.empty((n, 3), dtype=torch.int32, device="cuda")
        vals = [torch.empty((n, r, r, r, ch), dtype=dt, device="cuda")
                for dt, ch in self._attr_layout()]
        if n:
            ptrs = (C.c_void_p * len(vals))(*[v.data_ptr() for v in vals])
            m = C.c_int64(0)
            _lib.check(_lib.lib().o3dmi_vbg_export_blocks(
                self._g, n, _lib.ptr(keys), ptrs, C.byref(m), stream()),
                "export_blocks")
            assert m.value == n
        return keys, vals

    def merge_frame_sharded(self, comm, replicate=False):
        """The payload exchange of frame-sharded integration, owner-partitioned
        (o3dmi_vbg_merge_frame_sharded): every active block travels to the
        rank that owns it and is folded in there; afterwards the ranks hold
        disjoint grids whose union is the model of the whole stream. With
        `replicate` the finished blocks are then all-gathered so that every
        rank holds the whole model. `comm`: sharding.Comm. Collective."""
        _lib.check(_lib.lib().o3dmi_vbg_merge_frame_sharded(
            self._g, comm.handle, stream()), "merge_frame_sharded")
        if replicate:
            _lib.check(_lib.lib().o3dmi_vbg_allgather_owned_blocks(
                self._g, comm.handle, stream()), "allgather_owned_blocks")

    def merge_blocks(self, keys, values):
        """Folds foreign blocks (keys {n,3} int32 unique, one value tensor per
        attribute in this grid's layout) into the grid: weighted running mean
        per voxel, weights added (o3dmi_vbg_merge_blocks)."""
        keys = require_cuda(keys, "keys")
        if keys.dtype != torch.int32 or keys.dim() != 2 or keys.shape[1] != 3:
            raise ValueError("keys must be {n,3} Int32")
        layout = self._attr_layout()
        if len(values) != len(layout):
            raise ValueError("one value tensor per attribute expected")
        n = keys.shape[0]
        r = self.block_resolution
        vals = []
        for v, (dt, ch) in zip(values, layout):
            v = require_cuda(v, "values")
            if v.dtype != dt or v.numel() != n * r * r * r * ch:
                raise ValueError("value tensor does not match the grid's "
                                 "attribute layout")
            vals.append(v)
        ptrs = (C.c_void_p * len(vals))(*[v.data_ptr() for v in vals])
        _lib.check(_lib.lib().o3dmi_vbg_merge_blocks(
            self._g, _lib.ptr(keys), ptrs, n, stream()), "merge_blocks")

    def save(self, file_name):
        """VoxelBlockGrid::Save (VoxelBlockGrid.cpp:474-524): NPZ of the
        active blocks; ".npz" is appended when missing."""
        _lib.check(_lib.lib().o3dmi_vbg_save(self._g, str(file_name).encode(),
                                             stream()), "VoxelBlockGrid.save")

    def to(self, device):
        """VoxelBlockGrid::To(device, copy=True): the grid on HIP device
        `device` (index or "cuda:i"; its own device gives a deep copy). Use the
        result with that device current (torch.cuda.device(i))."""
        idx = torch.device(device).index if not isinstance(device, int) \
            else device
        torch.cuda.synchronize()
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_vbg_to_device(self._g, int(idx or 0),
                                                  C.byref(h)),
                   "VoxelBlockGrid.to")
        return VoxelBlockGrid._adopt(h)

    @staticmethod
    def load(file_name):
        """VoxelBlockGrid::Load (VoxelBlockGrid.cpp:538-596)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_vbg_load(str(file_name).encode(), stream(),
                                             C.byref(h)),
                   "VoxelBlockGrid.load")
        return VoxelBlockGrid._adopt(h)

    @staticmethod
    def _adopt(h):
        L = _lib.lib()
        g = VoxelBlockGrid.__new__(VoxelBlockGrid)
        g._g = h
        g.voxel_size = float(L.o3dmi_vbg_voxel_size(h))
        g.block_resolution = int(L.o3dmi_vbg_block_resolution(h))
        g.attr_names = [L.o3dmi_vbg_attribute_name(h, i).decode()
                        for i in range(L.o3dmi_vbg_attribute_count(h))]
        g._chans = {}
        for nm in g.attr_names:
            dt, ch = C.c_int(0), C.c_int(0)
            L.o3dmi_vbg_attribute(h, nm.encode(), C.byref(dt), C.byref(ch))
            g._chans[nm] = ch.value
        return g

    def extract_point_cloud(self, weight_threshold=3.0,
                            estimated_point_number=-1):
        """ExtractPointCloud (VoxelBlockGrid.cpp:404-434) -> dict(positions,
        normals[, colors]) of device tensors {N,3} float32. A negative
        estimate runs the counting pass first (the reference's 2-pass mode);
        otherwise at most `estimated_point_number` points are returned."""
        return _extract(lambda *a: _lib.lib().o3dmi_vbg_extract_point_cloud(
            self._g, *a), "color" in self.attr_names, weight_threshold,
            estimated_point_number)

    def profile_begin(self, max_frames, stride=1):
        _lib.check(_lib.lib().o3dmi_vbg_profile_begin(
            self._g, int(max_frames), int(stride)), "profile_begin")

    def profile_end(self):
        """-> dict(integrate_ms, launches, block_frames, frames,
        distinct_blocks) over the bracketed launches."""
        ti = C.c_double(0)
        n, bf, fr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.lib().o3dmi_vbg_profile_end(
            self._g, stream(), C.byref(ti), C.byref(n), C.byref(bf),
            C.byref(fr)), "profile_end")
        return dict(integrate_ms=ti.value, launches=n.value,
                    block_frames=bf.value, frames=fr.value,
                    distinct_blocks=int(
                        _lib.lib().o3dmi_vbg_profile_distinct_blocks(self._g)))

    def profile_launches(self):
        """Per bracketed launch of the last profile_end -> dict of numpy
        arrays: ms, block_frames, distinct_blocks, map_size."""
        import numpy as np
        L = _lib.lib()
        n = int(L.o3dmi_vbg_profile_launches(self._g, 0, None, None, None,
                                             None))
        ms = np.zeros(n, np.float32)
        bf, db, sz = (np.zeros(n, np.int32) for _ in range(3))
        L.o3dmi_vbg_profile_launches(
            self._g, n, ms.ctypes.data_as(C.c_void_p),
            bf.ctypes.data_as(C.c_void_p), db.ctypes.data_as(C.c_void_p),
            sz.ctypes.data_as(C.c_void_p))
        return dict(ms=ms, block_frames=bf, distinct_blocks=db, map_size=sz)

    def last_frame_block_coordinates(self, capacity):
        """Extension: the block coordinates the most recent integrate_frame
        touched (= compute_unique_block_coordinates of that frame), without a
        second touch or a host round trip -> (coords {capacity,3} int32, count
        {1} int32, both on the device); give both to ray_cast(...,
        block_count_dev=count)."""
        out = torch.empty((int(capacity), 3), dtype=torch.int32, device="cuda")
        cnt = torch.empty(1, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib().o3dmi_vbg_last_frame_block_coordinates(
            self._g, _lib.ptr(out), int(capacity), _lib.ptr(cnt), stream()),
            "VoxelBlockGrid.last_frame_block_coordinates")
        return out, cnt

    def ray_cast_last_frame(self, intrinsic, extrinsic, width, height,
                            render_attributes=("depth", "normal"),
                            depth_scale=1000.0, depth_min=0.1, depth_max=3.0,
                            weight_threshold=3.0, trunc_voxel_multiplier=8.0,
                            range_map_down_factor=8):
        """Extension: ray cast over the blocks the most recent integrate_frame
        touched, read from the grid's own list, with the grid's own range map
        (o3dmi_vbg_ray_cast_dev without block coordinates and without a range
        map: no export launch, and the ray cast leaves the map clean for the
        next call). Same maps as ray_cast(last_frame_block_coordinates(...))."""
        K = host_mat(intrinsic, (3, 3), "intrinsic")
        T = host_mat(extrinsic, (4, 4), "extrinsic")
        out = {}
        for a in render_attributes:
            if a not in ("depth", "vertex", "color", "normal"):
                raise ValueError("ray_cast_last_frame renders depth / vertex "
                                 "/ color / normal, not %s" % a)
            out[a] = torch.empty((height, width, self._CHANNELS[a]),
                                 dtype=torch.float32, device="cuda")
        g = lambda a: _lib.ptr(out.get(a))
        _lib.check(_lib.lib().o3dmi_vbg_ray_cast_dev(
            self._g, None, 0, None, _lib.f64p(K), _lib.f64p(T), int(width),
            int(height), None, g("depth"), g("vertex"), g("color"),
            g("normal"), None, None, None, None, None, None,
            C.c_float(depth_scale), C.c_float(depth_min), C.c_float(depth_max),
            C.c_float(weight_threshold), C.c_float(trunc_voxel_multiplier),
            int(range_map_down_factor), stream()),
            "VoxelBlockGrid.ray_cast_last_frame")
        return out

    def ray_cast(self, block_coords, intrinsic, extrinsic, width, height,
                 render_attributes=("depth", "color"), depth_scale=1000.0,
                 depth_min=0.1, depth_max=3.0, weight_threshold=3.0,
                 trunc_voxel_multiplier=8.0, range_map_down_factor=8,
                 block_count_dev=None, sharded=False):
        """sharded: the rows of the maps are rendered by the ranks of the
        calling thread's communicator (sharding.Comm.install) and all-gathered
        -- a collective call on a replicated grid; depth / vertex / color /
        normal only."""
        block_coords = require_cuda(block_coords, "block_coords")
        if block_coords.dtype != torch.int32:
            raise ValueError("Unsupported block coordinate dtype %s"
                             % block_coords.dtype)
        K = host_mat(intrinsic, (3, 3), "intrinsic")
        T = host_mat(extrinsic, (4, 4), "extrinsic")
        out = {}
        for a in render_attributes:
            if a not in self._CHANNELS:
                raise ValueError("Unsupported attribute %s, please implement "
                                 "customized ray casting." % a)
            dt = torch.bool if a == "mask" else (
                torch.int64 if a == "index" else torch.float32)
            out[a] = torch.empty((height, width, self._CHANNELS[a]), dtype=dt,
                                 device="cuda")
        d = range_map_down_factor
        out["range"] = torch.empty((height // d, width // d, 2),
                                   dtype=torch.float32, device="cuda")
        g = lambda a: _lib.ptr(out.get(a))
        if sharded:
            extra = set(out) - {"depth", "vertex", "color", "normal", "range"}
            if extra:
                raise ValueError("sharded ray cast renders depth / vertex / "
                                 "color / normal, not %s" % sorted(extra))
            _lib.check(_lib.lib().o3dmi_vbg_ray_cast_sharded(
                self._g, _lib.ptr(block_coords), block_coords.shape[0],
                _lib.f64p(K), _lib.f64p(T), int(width), int(height),
                g("range"), g("depth"), g("vertex"), g("color"), g("normal"),
                C.c_float(depth_scale), C.c_float(depth_min),
                C.c_float(depth_max), C.c_float(weight_threshold),
                C.c_float(trunc_voxel_multiplier),
                int(range_map_down_factor), stream()),
                "VoxelBlockGrid.ray_cast(sharded)")
            return out
        if block_count_dev is not None:
            # the number of rows of block_coords in use lives on the device
            _lib.check(_lib.lib().o3dmi_vbg_ray_cast_dev(
                self._g, _lib.ptr(block_coords), block_coords.shape[0],
                _lib.ptr(block_count_dev), _lib.f64p(K), _lib.f64p(T),
                int(width), int(height), g("range"), g("depth"), g("vertex"),
                g("color"), g("normal"), g("index"), g("mask"),
                g("interp_ratio"), g("interp_ratio_dx"), g("interp_ratio_dy"),
                g("interp_ratio_dz"), C.c_float(depth_scale),
                C.c_float(depth_min), C.c_float(depth_max),
                C.c_float(weight_threshold), C.c_float(trunc_voxel_multiplier),
                int(range_map_down_factor), stream()),
                "VoxelBlockGrid.ray_cast")
            return out
        _lib.check(_lib.lib().o3dmi_vbg_ray_cast(
            self._g, _lib.ptr(block_coords), block_coords.shape[0],
            _lib.f64p(K), _lib.f64p(T), int(width), int(height), g("range"),
            g("depth"), g("vertex"), g("color"), g("normal"), g("index"),
            g("mask"), g("interp_ratio"), g("interp_ratio_dx"),
            g("interp_ratio_dy"), g("interp_ratio_dz"), C.c_float(depth_scale),
            C.c_float(depth_min), C.c_float(depth_max),
            C.c_float(weight_threshold), C.c_float(trunc_voxel_multiplier),
            int(range_map_down_factor), stream()), "VoxelBlockGrid.ray_cast")
        return out
