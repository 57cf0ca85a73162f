#!/usr/bin/env python
"""Secondary measurements of the hot path (the headline line is bench.py):

  --mode icp    BASELINE configs[0] shape on the GPU: point-to-plane ICP on two
                synthetic 100k-point clouds (single scale, max_dist 0.07,
                criteria (1e-6, 1e-6, 30)); ms per ICP call, per iteration, and
                the same call through the CPU oracle (pose parity printed).
  --mode slam   BASELINE configs[2]: tracking loop on a synthetic 1280x720
                stream -- frame cloud (Unproject, stride 2) against the model
                cloud (vertex / normal maps ray-cast from the grid at the
                previous pose), multi-scale ICP (voxel 5 / 2.5 / 1.25 cm,
                20/10/5 iterations), integrate at the estimated pose, ray-cast
                for the next frame. frames/s and pose drift against the
                ground-truth trajectory.

One JSON line per mode on stdout. torch is used as an array library for the
glue between operators (mask / reshape / 4x4 products), as an Open3D user
would use Tensor ops there.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pose_err(A, B):
    D = np.linalg.inv(A) @ B
    ang = float(np.arccos(np.clip((np.trace(D[:3, :3]) - 1) / 2, -1, 1)))
    return ang, float(np.linalg.norm(D[:3, 3]))


def mode_icp(a):
    from open3d_amd import registration as reg, synthetic
    p = synthetic.make_icp_pair(a.points, a.points, seed=0)
    src = torch.from_numpy(p["source"]).cuda()
    tgt = torch.from_numpy(p["target"]).cuda()
    nrm = torch.from_numpy(p["target_normals"]).cuda()
    crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 30)
    res = None
    for _ in range(2):
        res = reg.icp(src, tgt, nrm, 0.07, criteria=crit)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        res = reg.icp(src, tgt, nrm, 0.07, criteria=crit)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.repeat * 1e3
    out = {"mode": "icp", "points": a.points, "ms_per_icp": ms,
           "iterations": res.num_iterations,
           "ms_per_iteration": ms / max(1, res.num_iterations),
           "fitness": res.fitness, "inlier_rmse": res.inlier_rmse,
           "pose_err_vs_ground_truth_rad_m": pose_err(p["T_gt"],
                                                      res.transformation)}
    if not a.no_cpu:
        import _oracle as orc
        orc.set_threads(min(64, os.cpu_count() or 1))
        t0 = time.perf_counter()
        want = orc.multiscale_icp(p["source"], p["target"],
                                  p["target_normals"], [-1.0],
                                  [(1e-6, 1e-6, 30)], [0.07],
                                  accumulate_double=True)
        out["cpu_oracle_ms_per_icp"] = (time.perf_counter() - t0) * 1e3
        out["cpu_oracle_threads"] = min(64, os.cpu_count() or 1)
        out["pose_err_vs_oracle_rad_m"] = pose_err(want["transformation"],
                                                   res.transformation)
        out["same_iterations_as_oracle"] = (want["num_iterations"] ==
                                            res.num_iterations)
    print(json.dumps(out), flush=True)


def mode_slam(a):
    from open3d_amd import geometry, registration as reg, synthetic
    from open3d_amd import _lib
    from open3d_amd.core import stream
    import ctypes as C
    W, H = 1280, 720
    voxel, res, trunc = 0.008, 16, 8.0
    ds, dmax = 1000.0, 3.0
    n = a.frames
    K = synthetic.intrinsics(W, H)
    depths, colors, Ts = [], [], []
    for k in range(n):
        d, c, _, T = synthetic.render_frames(k * a.frame_step, 1, W, H,
                                             device="cuda")
        depths.append(d[0].contiguous())
        colors.append(c[0].contiguous())
        Ts.append(T[0])
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], voxel, res, a.block_count)
    vs = [0.05, 0.025, 0.0125]
    crit = [reg.ICPConvergenceCriteria(1e-6, 1e-6, it) for it in (20, 10, 5)]
    md = [0.15, 0.075, 0.0375]
    L = _lib.lib()
    stride = 2
    pts_buf = torch.empty(((H // stride) * (W // stride), 3),
                          dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")

    def frame_cloud(depth, T_wc):
        """PointCloud::CreateFromDepthImage(depth, K, T, scale, max, stride)."""
        T = np.ascontiguousarray(T_wc, dtype=np.float64)
        _lib.check(L.o3dmi_unproject(
            _lib.ptr(depth), _lib.U16, H, W, None, _lib.ptr(pts_buf), None,
            _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(T), C.c_float(ds),
            C.c_float(dmax), C.c_int64(stride), stream()), "unproject")
        return pts_buf[:int(cnt.item())]

    def model_cloud(T_wc):
        keys = g.compute_unique_block_coordinates(depth_pred, K, T_wc, ds, dmax,
                                                  trunc)
        out = g.ray_cast(keys, K, T_wc, W, H,
                         render_attributes=("depth", "vertex", "normal"),
                         depth_scale=ds, depth_min=0.1, depth_max=dmax,
                         weight_threshold=1.0, trunc_voxel_multiplier=trunc)
        valid = (out["depth"][..., 0] > 0) & \
                torch.isfinite(out["normal"]).all(-1) & \
                (out["normal"].abs().sum(-1) > 0)
        valid[::2, :] = False   # same density as the stride-2 frame cloud
        valid[:, ::2] = False
        v = out["vertex"][valid]
        nn = out["normal"][valid]
        Tinv = torch.from_numpy(np.linalg.inv(T_wc)).to(v.device,
                                                        torch.float32)
        pw = v @ Tinv[:3, :3].T + Tinv[:3, 3]
        nw = nn @ Tinv[:3, :3].T
        return pw.contiguous(), nw.contiguous(), out["depth"]

    # bootstrap with frame 0 at its true pose
    T_est = [np.array(Ts[0])]
    g.integrate_frame(depths[0], colors[0], K, K, Ts[0], ds, dmax, trunc)
    depth_pred = depths[0]
    torch.cuda.synchronize()
    iters = 0
    t0 = time.perf_counter()
    for k in range(1, n):
        T_prev = T_est[-1]
        tp, tn, dpred = model_cloud(T_prev)
        # source in the previous camera's world alignment: ICP estimates the
        # world-frame correction from the previous pose to the current one
        src = frame_cloud(depths[k], T_prev)
        r = reg.multi_scale_icp(src, tp, tn, vs, crit, md)
        iters += r.num_iterations
        # points_world = r.T * (T_prev^-1 * p_cam)  =>  extrinsic_k = T_prev * r.T^-1
        T_k = T_prev @ np.linalg.inv(r.transformation)
        T_est.append(T_k)
        g.integrate_frame(depths[k], colors[k], K, K, T_k, ds, dmax, trunc)
        depth_pred = depths[k]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    errs = [pose_err(Ts[k], T_est[k]) for k in range(n)]
    out = {"mode": "slam", "workload": "configs[2]: 1280x720 synthetic stream, "
           "multi-scale ICP (5/2.5/1.25 cm; 20/10/5 it) + integrate + ray cast",
           "frames": n - 1, "frames_per_s": (n - 1) / dt,
           "ms_per_frame": dt / (n - 1) * 1e3,
           "icp_iterations_per_frame": iters / (n - 1),
           "final_pose_err_rad_m": errs[-1],
           "max_pose_err_rad_m": [max(e[0] for e in errs),
                                  max(e[1] for e in errs)],
           "active_blocks": g.hashmap().size()}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["icp", "slam", "both"], default="both")
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--frame-step", type=int, default=4)
    ap.add_argument("--block-count", type=int, default=65536)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    if a.mode in ("icp", "both"):
        mode_icp(a)
    if a.mode in ("slam", "both"):
        mode_slam(a)


if __name__ == "__main__":
    main()
