"""Python view of the library's NPZ codec (t::io::WriteNpz / ReadNpz mirror,
include/o3d_mi355x_host.h): dict of numpy arrays <-> .npz file, through the
native writer / reader (not numpy's)."""
import ctypes as C

import numpy as np

from . import _lib

_NP = {_lib.F32: np.float32, _lib.F64: np.float64, _lib.U16: np.uint16,
       _lib.U8: np.uint8, _lib.I32: np.int32, _lib.I64: np.int64,
       _lib.I8: np.int8, _lib.I16: np.int16, _lib.U32: np.uint32,
       _lib.U64: np.uint64, _lib.BOOL: np.bool_}
_CODE = {np.dtype(v): k for k, v in _NP.items()}


def write_npz(file_name, tensor_map):
    L = _lib.lib()
    z = C.c_void_p()
    _lib.check(L.o3dmi_npz_create(C.byref(z)), "npz_create")
    try:
        for name, a in tensor_map.items():
            a = np.asarray(a, order="C")  # (ascontiguousarray makes 0-d 1-d)
            if a.dtype not in _CODE:
                raise ValueError("Unsupported dtype: %s" % a.dtype)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(L.o3dmi_npz_add(z, name.encode(), _CODE[a.dtype],
                                       a.ndim, shape,
                                       a.ctypes.data_as(C.c_void_p)),
                       "npz_add")
        _lib.check(L.o3dmi_npz_write(z, str(file_name).encode()), "write_npz")
    finally:
        L.o3dmi_npz_destroy(z)


def read_npz(file_name):
    L = _lib.lib()
    z = C.c_void_p()
    _lib.check(L.o3dmi_npz_read(str(file_name).encode(), C.byref(z)),
               "read_npz")
    out = {}
    try:
        for i in range(L.o3dmi_npz_count(z)):
            name = L.o3dmi_npz_name(z, i)
            dt, nd = C.c_int(0), C.c_int(0)
            shape = (C.c_int64 * 8)()
            p = C.c_void_p()
            _lib.check(L.o3dmi_npz_get(z, name, C.byref(dt), C.byref(nd),
                                       shape, C.byref(p)), "npz_get")
            shp = tuple(shape[k] for k in range(nd.value))
            n = int(np.prod(shp)) if shp else 1
            dtype = np.dtype(_NP[dt.value])
            if n == 0:
                arr = np.zeros(shp, dtype)
            else:
                buf = (C.c_char * (n * dtype.itemsize)).from_address(p.value)
                arr = np.frombuffer(buf, dtype=dtype).reshape(shp).copy()
            out[name.decode()] = arr
    finally:
        L.o3dmi_npz_destroy(z)
    return out
