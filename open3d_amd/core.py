"""Device plumbing shared by the Python mirror classes: torch supplies device
memory and the HIP stream; every compute call goes through the C ABI."""
import ctypes as C

import numpy as np
import torch

from . import _lib

TORCH_TO_O3DMI = {torch.float32: _lib.F32, torch.float64: _lib.F64,
                  torch.uint16: _lib.U16, torch.uint8: _lib.U8,
                  torch.int32: _lib.I32, torch.int64: _lib.I64}
O3DMI_TO_TORCH = {v: k for k, v in TORCH_TO_O3DMI.items()}


def stream():
    """Current torch HIP stream as a void* (kernels are ordered with torch)."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a device tensor" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def host_mat(a, shape, name):
    """Intrinsic / extrinsic checks of t/geometry/Utility.h
    (CheckIntrinsicTensor / CheckExtrinsicTensor): Float64, host, shape."""
    if isinstance(a, torch.Tensor):
        if a.is_cuda:
            raise ValueError("%s must be on CPU:0" % name)
        a = a.numpy()
    a = np.ascontiguousarray(np.asarray(a))
    if a.dtype != np.float64:
        raise ValueError("Unsupported %s dtype %s (Float64 expected)"
                         % (name, a.dtype))
    if a.shape != shape:
        raise ValueError("Unsupported %s shape %s" % (name, a.shape))
    return a


class _DevPtrView:
    """Wraps a raw device pointer so torch can adopt it without a copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": typestr,
            "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner


_TYPESTR = {_lib.F32: "<f4", _lib.F64: "<f8", _lib.U16: "<u2", _lib.U8: "|u1",
            _lib.I32: "<i4", _lib.I64: "<i8"}


def tensor_from_ptr(ptr, shape, dtype_code, owner):
    """Zero-copy torch view of library-owned device memory."""
    if dtype_code == _lib.U16:
        # torch's CUDA array interface import lacks uint16: view via int16.
        t = torch.as_tensor(_DevPtrView(ptr, shape, "<i2", owner),
                            device="cuda")
        return t.view(torch.uint16)
    return torch.as_tensor(_DevPtrView(ptr, shape, _TYPESTR[dtype_code],
                                       owner), device="cuda")
