"""open3d_amd -- MI355X-native backend for Open3D's tensor dense-SLAM hot path
(point-to-plane ICP + VoxelBlockGrid TSDF integration / ray casting).

Compute lives in hand-written HIP (open3d_amd/csrc) behind the C ABI declared
in include/o3d_mi355x.h and include/o3d_mi355x_host.h. This package is the
thin Python mirror of the reference's operator interface; importing the
compute modules requires the built library (no fallback).
"""
__version__ = "0.1.0"
