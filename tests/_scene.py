"""Shared seeded inputs for the parity tests (CPU tensors; GPU tests upload)."""
import functools

import numpy as np

from open3d_amd import synthetic as syn

VOXEL = 0.008
RES = 16
TRUNC_MULT = 8.0
DEPTH_SCALE = 1000.0
DEPTH_MAX = 3.0


@functools.lru_cache(maxsize=None)
def frames(k0, n, width=640, height=480):
    d, c, K, Ts = syn.render_frames(k0, n, width, height, device="cpu")
    return d.numpy(), c.numpy(), K, Ts


def as_f32_inputs(depth_u16, color_u8):
    """Float32 input convention: depth in raw units, colour in [0,1]."""
    return (depth_u16.astype(np.float32),
            (color_u8.astype(np.float32) / 255.0).astype(np.float32))


def sort_rows(a):
    a = np.asarray(a)
    if a.shape[0] == 0:
        return a
    idx = np.lexsort(a.T[::-1])
    return a[idx]
