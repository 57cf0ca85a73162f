"""Seeded synthetic RGB-D streams and ICP cloud pairs (closed-form scene).

No reference dataset is reachable offline, so inputs are
generated here: an axis-aligned room (camera inside) holding two spheres, a
camera on a smooth closed trajectory, PrimeSense-like intrinsics
(fx=fy=525, cx=319.5, cy=239.5 at 640x480 -- the constants the reference's own
tests use, cpp/tests/t/geometry/VoxelBlockGrid.cpp:34-43), uint16 depth at
scale 1000 (z-depth, as Open3D depth images store it) and procedural uint8
colour. Everything is closed-form ray/box and ray/sphere intersection in
float64, so the same function renders on the CPU (tests, oracle inputs) and on
the GPU (bench), and ground-truth poses are exact.

torch is used here as an array library only.
"""
import math

import numpy as np
import torch

ROOM_MIN = (-2.6, -1.3, -2.2)
ROOM_MAX = (2.6, 1.3, 2.2)
# (cx, cy, cz, r)
SPHERES = ((1.6, 0.7, 1.2, 0.6), (-1.5, 0.8, -1.0, 0.55), (0.2, 1.0, 1.9, 0.5),
           (-1.9, -0.3, 1.0, 0.4))


def intrinsics(width=640, height=480):
    """3x3 float64 pinhole matrix, scaled from the 640x480 PrimeSense values."""
    sx, sy = width / 640.0, height / 480.0
    K = np.array([[525.0 * sx, 0, (319.5 + 0.5) * sx - 0.5],
                  [0, 525.0 * sy, (239.5 + 0.5) * sy - 0.5],
                  [0, 0, 1]], np.float64)
    return K


def pose(k, frames_per_loop=1200):
    """World->camera extrinsic (4x4 float64) of frame k.

    The camera circles the room centre (radius 0.45 m, ~0.3 deg and ~2.4 mm
    per frame) looking outwards, with a gentle vertical bob and pitch.
    """
    th = 0.35 + 2.0 * math.pi * (k % frames_per_loop) / frames_per_loop
    c = np.array([0.45 * math.sin(th), 0.10 * math.sin(2 * th),
                  0.45 * math.cos(th)])
    pitch = 0.12 * math.sin(3 * th)
    # camera axes in world coordinates (x right, y down, z forward)
    z = np.array([math.sin(th) * math.cos(pitch), math.sin(pitch),
                  math.cos(th) * math.cos(pitch)])
    x = np.array([math.cos(th), 0.0, -math.sin(th)])
    y = np.cross(z, x)
    R = np.stack([x, y, z])  # rows: world -> camera
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = -R @ c
    return T


def _intersect(o, d):
    """o: (...,3) origins, d: (...,3) directions (float64 tensors).

    Returns t (ray parameter of the first hit), hit point and an integer
    surface id (0..5 walls, 6.. spheres)."""
    dev, dt = d.device, d.dtype
    lo = torch.tensor(ROOM_MIN, device=dev, dtype=dt)
    hi = torch.tensor(ROOM_MAX, device=dev, dtype=dt)
    eps = 1e-12
    dsafe = torch.where(d.abs() < eps, torch.full_like(d, eps), d)
    t_axis = torch.where(d > 0, (hi - o) / dsafe, (lo - o) / dsafe)
    t_axis = torch.where(d.abs() < eps, torch.full_like(t_axis, 1e30), t_axis)
    t, axis = t_axis.min(dim=-1)
    sid = axis * 2 + (torch.gather(d, -1, axis.unsqueeze(-1)).squeeze(-1) > 0)
    for i, (sx, sy, sz, sr) in enumerate(SPHERES):
        c = torch.tensor((sx, sy, sz), device=dev, dtype=dt)
        oc = o - c
        a = (d * d).sum(-1)
        b = 2.0 * (oc * d).sum(-1)
        cc = (oc * oc).sum(-1) - sr * sr
        disc = b * b - 4 * a * cc
        ok = disc > 0
        sq = torch.sqrt(torch.clamp(disc, min=0))
        ts = (-b - sq) / (2 * a)
        hit = ok & (ts > 1e-6) & (ts < t)
        t = torch.where(hit, ts, t)
        sid = torch.where(hit, torch.full_like(sid, 6 + i), sid)
    p = o + t.unsqueeze(-1) * d
    return t, p, sid


def _normals(p, sid):
    """Analytic surface normals (pointing towards the room interior / out of
    the spheres)."""
    n = torch.zeros_like(p)
    for axis in range(3):
        for side in range(2):
            m = sid == axis * 2 + side
            val = -1.0 if side == 1 else 1.0
            n[..., axis] = torch.where(m, torch.full_like(n[..., axis], val),
                                       n[..., axis])
    for i, (sx, sy, sz, sr) in enumerate(SPHERES):
        c = torch.tensor((sx, sy, sz), device=p.device, dtype=p.dtype)
        m = (sid == 6 + i).unsqueeze(-1)
        n = torch.where(m, (p - c) / sr, n)
    return n


def _shade(p, sid):
    """Procedural uint8 RGB from the hit point."""
    f = 2.0 * math.pi
    r = 128 + 100 * torch.sin(f * 0.9 * p[..., 0] + 0.3 * sid)
    g = 128 + 100 * torch.sin(f * 1.1 * p[..., 1] + 1.7)
    b = 128 + 100 * torch.sin(f * 0.7 * p[..., 2] + 0.9 * sid)
    return torch.stack([r, g, b], -1).clamp(0, 255).to(torch.uint8)


def render_frames(k0, n, width=640, height=480, device="cpu",
                  depth_scale=1000.0, noise_sigma=0.0, seed=0):
    """Renders frames k0..k0+n-1.

    Returns (depth uint16 {n,H,W}, color uint8 {n,H,W,3}, K 3x3 float64 numpy,
    T_list [n] of 4x4 float64 numpy)."""
    K = intrinsics(width, height)
    dev = torch.device(device)
    u = torch.arange(width, device=dev, dtype=torch.float64)
    v = torch.arange(height, device=dev, dtype=torch.float64)
    dc = torch.stack([((u - K[0, 2]) / K[0, 0]).expand(height, width),
                      ((v - K[1, 2]) / K[1, 1]).unsqueeze(1).expand(height,
                                                                    width),
                      torch.ones(height, width, device=dev,
                                 dtype=torch.float64)], -1)  # camera-frame dirs
    Ts = [pose(k0 + i) for i in range(n)]
    depths, colors = [], []
    gen = None
    if noise_sigma > 0:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed))
    for T in Ts:
        R = torch.tensor(T[:3, :3], device=dev, dtype=torch.float64)
        c = torch.tensor(-T[:3, :3].T @ T[:3, 3], device=dev,
                         dtype=torch.float64)
        dw = dc @ R  # R^T applied to row vectors
        t, p, sid = _intersect(c.expand_as(dw), dw)
        z = t  # direction has camera z == 1 -> t is z-depth
        if gen is not None:
            z = z + noise_sigma * torch.randn(z.shape, generator=gen,
                                              device=dev, dtype=torch.float64)
        d16 = torch.clamp(torch.round(z * depth_scale), 0, 65535).to(
                torch.int32).to(torch.uint16)
        depths.append(d16)
        colors.append(_shade(p, sid))
    return torch.stack(depths), torch.stack(colors), K, Ts


def make_icp_pair(n_source=100000, n_target=100000, seed=0, dtype=np.float32,
                  rot_deg=3.0, trans=0.05):
    """Two independently sampled clouds of the room surfaces.

    target: points + analytic normals in the world frame.
    source: an independent sample, moved by the INVERSE of a known rigid
    transform T_gt, so that ICP(source -> target) should recover T_gt
    (rotation ~rot_deg degrees, translation ~trans metres).
    Returns dict(source, target, target_normals, T_gt) as numpy arrays.
    """
    rng = np.random.RandomState(seed)

    def sample(n):
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = np.zeros((n, 3)) + np.array([0.1, -0.05, 0.0])
        t, p, sid = _intersect(torch.from_numpy(o), torch.from_numpy(v))
        nrm = _normals(p, sid)
        return p.numpy(), nrm.numpy()

    tgt, tgt_n = sample(n_target)
    src_w, _ = sample(n_source)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = math.radians(rot_deg)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(a) * Kx + (1 - math.cos(a)) * (Kx @ Kx)
    tv = rng.normal(size=3)
    tv = tv / np.linalg.norm(tv) * trans
    T_gt = np.eye(4)
    T_gt[:3, :3] = R
    T_gt[:3, 3] = tv
    # source = T_gt^-1 * src_w
    src = (src_w - tv) @ R  # R^T (p - t) for row vectors
    return dict(source=np.ascontiguousarray(src.astype(dtype)),
                target=np.ascontiguousarray(tgt.astype(dtype)),
                target_normals=np.ascontiguousarray(tgt_n.astype(dtype)),
                T_gt=T_gt)
