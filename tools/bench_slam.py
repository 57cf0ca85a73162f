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

  --mode model  the reference's own dense-SLAM loop (examples/python/
                t_reconstruction_system/dense_slam.py) through slam::Model:
                TrackFrameToModel (RGB-D odometry, point-to-plane, {6,3,1}
                iterations) -> UpdateFramePose -> Integrate ->
                SynthesizeModelFrame, on a synthetic VGA or 720p stream; with
                per-operator times (HIP events) and, with --cpu-frames > 0, the
                same operators through the CPU oracle on the first frames.

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
    p2point = a.estimation == "p2point"
    est = (reg.TransformationEstimationPointToPoint() if p2point
           else reg.TransformationEstimationPointToPlane())
    if p2point:
        nrm = None
    res = None
    for _ in range(2):
        res = reg.icp(src, tgt, nrm, 0.07, criteria=crit,
                      estimation_method=est)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        res = reg.icp(src, tgt, nrm, 0.07, criteria=crit,
                      estimation_method=est)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.repeat * 1e3
    out = {"mode": "icp", "estimation": a.estimation, "points": a.points,
           "ms_per_icp": ms,
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
                                  None if p2point else p["target_normals"],
                                  [-1.0], [(1e-6, 1e-6, 30)], [0.07],
                                  accumulate_double=True,
                                  estimation=1 if p2point else 0)
        out["cpu_oracle_ms_per_icp"] = (time.perf_counter() - t0) * 1e3
        out["cpu_oracle_threads"] = min(64, os.cpu_count() or 1)
        if not p2point:
            # one iteration, phase by phase: search, 29-sum
            # accumulation, transform
            ta = time.perf_counter()
            idx, _, _ = orc.hybrid_search(p["target"], p["source"], 0.07, 1)
            tb = time.perf_counter()
            orc.p2plane_accumulate(p["source"], p["target"],
                                   p["target_normals"],
                                   idx[:, 0].astype(np.int64),
                                   accumulate_double=True)
            tc = time.perf_counter()
            orc.transform_points(np.eye(4), p["source"])
            td = time.perf_counter()
            out["cpu_oracle_ms_per_iteration_phase"] = {
                "search": (tb - ta) * 1e3, "accumulate": (tc - tb) * 1e3,
                "transform": (td - tc) * 1e3}
        out["pose_err_vs_oracle_rad_m"] = pose_err(want["transformation"],
                                                   res.transformation)
        out["same_iterations_as_oracle"] = (want["num_iterations"] ==
                                            res.num_iterations)
    # SURVEY 8(d) unit = source point x iteration: 12 B query + 27 bucket
    # heads (8 B) + visited records x 16 B + 44 B accumulate (12 src + 8 corr +
    # 24 gathered target point + normal) + 24 B transform. The time is the
    # whole iteration (search + accumulate + final sum + host hop + transform),
    # so the fraction is a lower bound for the search kernel's own.
    if not p2point:
        visited = mean_visited_records(p["target"], p["source"], 0.07)
        unit = 12 + 27 * 8 + visited * 16 + 44 + 24
        gbs = a.points * unit / (out["ms_per_iteration"] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "unit": "GB/s",
                           "bytes_per_point_iteration": unit,
                           "visited_records_per_query": visited,
                           "achieved": gbs, "peak": 8000.0,
                           "frac": gbs / 8000.0,
                           "basis": "whole-iteration wall time (lower bound "
                                    "for SearchAccumulateKernel)"}
    return out


def _voxel_centres(points, voxel):
    """One representative point per occupied voxel (what VoxelDownSample leaves
    of a cloud, up to the position inside the voxel)."""
    p = np.asarray(points, np.float64)
    if voxel <= 0:
        return p
    k = np.floor(p / voxel).astype(np.int64)
    _, first = np.unique(k, axis=0, return_index=True)
    return p[np.sort(first)]


def mean_visited_records(target, queries, radius, sample=4000):
    """Average number of index records in the 27 cells (edge = radius) around a
    query -- the 'visited records' of SURVEY 8(d), counted on the host."""
    t = np.floor(np.asarray(target, np.float64) / radius).astype(np.int64)
    key = (t[:, 0] + (1 << 20)) << 42 | (t[:, 1] + (1 << 20)) << 21 | \
        (t[:, 2] + (1 << 20))
    uk, cnt = np.unique(key, return_counts=True)
    q = np.asarray(queries, np.float64)
    if q.shape[0] > sample:
        q = q[:: q.shape[0] // sample]
    qc = np.floor(q / radius).astype(np.int64)
    tot = np.zeros(q.shape[0], np.int64)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                k = (qc[:, 0] + dx + (1 << 20)) << 42 | \
                    (qc[:, 1] + dy + (1 << 20)) << 21 | \
                    (qc[:, 2] + dz + (1 << 20))
                i = np.searchsorted(uk, k)
                i = np.minimum(i, uk.shape[0] - 1)
                tot += np.where(uk[i] == k, cnt[i], 0)
    return float(tot.mean())


def mode_slam(a):
    from open3d_amd import geometry, registration as reg, synthetic
    from open3d_amd import _lib
    from open3d_amd.core import stream
    import ctypes as C
    W, H = (640, 480) if getattr(a, "vga", False) else (1280, 720)
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

    mpts = torch.empty(((H // stride) * (W // stride), 3),
                       dtype=torch.float32, device="cuda")
    mnrm = torch.empty_like(mpts)
    mcnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    EYE = np.ascontiguousarray(np.eye(4), dtype=np.float64)

    def clouds(T_wc, depth):
        """Model cloud at the previous pose -- ray-cast depth + normal maps,
        PointCloud::CreateFromDepthImage(ray-cast depth, stride) with the
        normal map carried along as the per-pixel attribute -- and the frame
        cloud of the new depth image, each in its own camera's frame, by one
        Unproject launch (as examples/icp_slam.cpp since round 6: the normals
        as rendered, no rotation launch). Library calls only, no tensor glue."""
        # the block coordinates of the frame integrated last, as the
        # integration left them on the device (= GetUniqueBlockCoordinates of
        # that frame; no second touch, no host wait)
        out = g.ray_cast(frame_keys[0], K, T_wc, W, H,
                         render_attributes=("depth", "normal"),
                         depth_scale=ds, depth_min=0.1, depth_max=dmax,
                         weight_threshold=1.0, trunc_voxel_multiplier=trunc,
                         block_count_dev=frame_keys[1])
        _lib.check(L.o3dmi_unproject_pair(
            _lib.ptr(out["depth"]), _lib.F32, _lib.ptr(out["normal"]),
            _lib.ptr(mpts), _lib.ptr(mnrm), _lib.ptr(mcnt), _lib.f64p(EYE),
            _lib.ptr(depth), _lib.U16, None, _lib.ptr(pts_buf), None,
            _lib.ptr(cnt), _lib.f64p(EYE), H, W, _lib.f64p(K), C.c_float(ds),
            C.c_float(dmax), C.c_int64(stride), stream()), "unproject_pair")
        # the live sizes stay on the device (mcnt, cnt)
        if getattr(a, "host_counts", False):
            m = int(mcnt.item())
            return mpts[:m], mnrm[:m], pts_buf[:int(cnt.item())]
        return mpts, mnrm, pts_buf

    # bootstrap with frame 0 at its true pose
    T_est = [np.array(Ts[0])]
    keys_cap = (H // 4) * (W // 4) * 4
    g.integrate_frame(depths[0], colors[0], K, K, Ts[0], ds, dmax, trunc)
    frame_keys = g.last_frame_block_coordinates(keys_cap)
    torch.cuda.synchronize()
    iters = 0
    iters_log = []
    phase = np.zeros(4)

    def tick():
        if a.phases:
            torch.cuda.synchronize()
        return time.perf_counter()

    t0 = time.perf_counter()
    for k in range(1, n):
        T_prev = T_est[-1]
        p0 = tick()
        # source in its own camera's frame: ICP estimates the motion from
        # this camera to the previous one, starting from the identity
        tp, tn, src = clouds(T_prev, depths[k])
        p1 = tick()
        p2 = p1
        r = reg.multi_scale_icp(
            src, tp, tn, vs, crit, md,
            device_counts=None if getattr(a, "host_counts", False) else (cnt, mcnt))
        iters_log.append(r.num_iterations)
        p3 = tick()
        iters += r.num_iterations
        # p_prev_cam = r.T p_cam = T_prev p_world  =>  extrinsic_k = r.T^-1 T_prev
        T_k = np.linalg.inv(r.transformation) @ T_prev
        T_est.append(T_k)
        g.integrate_frame(depths[k], colors[k], K, K, T_k, ds, dmax, trunc)
        frame_keys = g.last_frame_block_coordinates(keys_cap)
        p4 = tick()
        phase += (p1 - p0, p2 - p1, p3 - p2, p4 - p3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    errs = [pose_err(Ts[k], T_est[k]) for k in range(n)]
    out = {"mode": "slam", "workload": "configs[2]: %dx%d synthetic stream, "
           "multi-scale ICP (5/2.5/1.25 cm; 20/10/5 it) + integrate + ray cast"
           % (W, H),
           "frames": n - 1, "frames_per_s": (n - 1) / dt,
           "ms_per_frame": dt / (n - 1) * 1e3,
           "icp_iterations_per_frame": iters / (n - 1),
           "final_pose_err_rad_m": errs[-1],
           "max_pose_err_rad_m": [max(e[0] for e in errs),
                                  max(e[1] for e in errs)],
           "active_blocks": g.hashmap().size(),
           "cloud_sizes": "host" if getattr(a, "host_counts", False) else "device",
           "icp_iterations_first_frames": iters_log[:12]}
    if a.phases:
        out["ms_model_cloud_frame_cloud_icp_integrate"] = \
            [float(x) for x in phase / (n - 1) * 1e3]
    n_src, n_tgt = int(cnt.item()), int(mcnt.item())
    src, tp, tn = src[:n_src], tp[:n_tgt], tn[:n_tgt]
    out["source_points"] = n_src
    out["target_points"] = n_tgt
    # SURVEY 8(d) accounting of the ICP leg, level by level on the last
    # frame's clouds: points of the level (one per occupied voxel of
    # VoxelDownSample) x iterations the level ran (replayed once with the
    # iteration callback) x bytes per point-iteration (12 B query + 27 bucket
    # heads + visited records x 16 B + 44 B accumulate + 24 B transform).
    # Divided by the WHOLE frame time (track + integrate + ray cast), so the
    # fraction is a lower bound for the search kernel's own.
    src_np, tp_np, tn_np = (t.cpu().numpy() for t in (src, tp, tn))
    log = []
    reg.multi_scale_icp(src, tp, tn, vs, crit, md,
                        callback_after_iteration=log.append)
    per_level = []
    frame_bytes = 0.0
    for li, (v, r) in enumerate(zip(vs, md)):
        its = sum(1 for e in log if e["scale_index"] == li)
        sd = _voxel_centres(src_np, v)
        td = _voxel_centres(tp_np, v)
        visited = mean_visited_records(td, sd, r)
        unit = 12 + 27 * 8 + visited * 16 + 44 + 24
        frame_bytes += sd.shape[0] * its * unit
        per_level.append({"voxel": v, "source_points": int(sd.shape[0]),
                          "target_points": int(td.shape[0]),
                          "iterations": its,
                          "visited_records_per_query": visited,
                          "bytes_per_point_iteration": unit})
    gbs = frame_bytes / (dt / (n - 1)) / 1e9
    out["roofline"] = {"bound": "hbm", "unit": "GB/s", "levels": per_level,
                       "icp_bytes_per_frame": frame_bytes,
                       "achieved": gbs, "peak": 8000.0, "frac": gbs / 8000.0,
                       "basis": "last frame's per-level sizes and iteration "
                                "counts over the mean whole-frame time (track "
                                "+ integrate + ray cast): a lower bound"}
    if getattr(a, "cpu_frames", 0) > 0:
        import _oracle as orc
        orc.set_threads(min(64, os.cpu_count() or 1))
        crit_o = [(1e-6, 1e-6, it) for it in (20, 10, 5)]
        t0c = time.perf_counter()
        for _ in range(a.cpu_frames):
            orc.multiscale_icp(src_np, tp_np, tn_np, vs, crit_o, md,
                               accumulate_double=True)
        out["cpu_oracle_ms_per_multiscale_icp"] = \
            (time.perf_counter() - t0c) / a.cpu_frames * 1e3
        out["cpu_oracle_threads"] = min(64, os.cpu_count() or 1)
    return out


def mode_model(a):
    from open3d_amd import slam, synthetic
    from open3d_amd.odometry import Method
    W, H = (1280, 720) if a.hd else (640, 480)
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
        Ts.append(np.linalg.inv(np.array(T[0])))  # frame -> world
    method = {"p2plane": Method.PointToPlane, "intensity": Method.Intensity,
              "hybrid": Method.Hybrid}[a.method]

    def run(timed):
        model = slam.Model(voxel, res, a.block_count, Ts[0])
        fin, frc = slam.Frame(H, W, K), slam.Frame(H, W, K)
        T = np.array(Ts[0])
        ev = []
        iters = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fin.set_data("depth", depths[i])
            fin.set_data("color", colors[i])
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)] \
                if timed else None
            if timed:
                e[0].record()
            if i > 0:
                r = model.track_frame_to_model(fin, frc, ds, dmax, 0.07,
                                               method)
                iters += r.num_iterations
                T = T @ r.transformation
            model.update_frame_pose(i, T)
            if timed:
                e[1].record()
            model.integrate(fin, ds, dmax, trunc)
            if timed:
                e[2].record()
            model.synthesize_model_frame(frc, ds, 0.1, dmax, trunc,
                                         a.method != "p2plane")
            if timed:
                e[3].record()
                ev.append(e)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return model, T, dt, ev, iters

    run(False)                     # warm-up (allocations, code objects)
    model, T, dt, _, iters = run(False)
    _, _, _, ev, _ = run(True)
    op = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(3)]
                   for e in ev[1:]])
    err = pose_err(Ts[-1], T)
    moved = pose_err(Ts[0], Ts[-1])
    out = {"mode": "model", "workload": "dense_slam.py loop through slam::Model"
           ", %dx%d synthetic stream, method %s, criteria {6,3,1}, 8 mm / 16^3"
           % (W, H, a.method),
           "frames": n, "frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3,
           "odometry_iterations_per_frame": iters / max(1, n - 1),
           "ms_track_integrate_synthesize": [float(x) for x in op.mean(0)],
           "final_pose_err_rad_m": err, "trajectory_moved_rad_m": moved,
           "active_blocks": model.get_hashmap().size()}
    if a.cpu_frames > 0:
        import _oracle as orc
        orc.set_threads(min(64, os.cpu_count() or 1))
        cap = 16384
        h = orc.HashMap(cap)
        tsdf = np.zeros((cap, res, res, res), np.float32)
        wgt = np.zeros((cap, res, res, res), np.uint16)
        col = np.zeros((cap, res, res, res, 3), np.uint16)
        Tc = np.array(Ts[0])
        rd = rc = None
        tt = np.zeros(3)
        crit = ((6, 1e-6, 1e-6), (3, 1e-6, 1e-6), (1, 1e-6, 1e-6))
        for i in range(min(a.cpu_frames, n)):
            dn, cn = depths[i].cpu().numpy(), colors[i].cpu().numpy()
            t0 = time.perf_counter()
            if i > 0:
                r = orc.rgbd_odometry_multiscale(
                    int(method), dn, rd, K, src_color=cn, tgt_color=rc,
                    criteria=crit, accumulate_double=False)
                Tc = Tc @ r["transformation"]
            t1 = time.perf_counter()
            extr = orc.inverse_transformation(Tc)
            keys = orc.depth_touch(dn, K, extr, res, voxel, voxel * trunc, ds,
                                   dmax)
            h.activate(keys)
            buf, _ = h.find(keys)
            orc.integrate(dn, cn, buf, h.key_buffer(), tsdf, wgt, col, K, K,
                          extr, res, voxel, voxel * trunc, ds, dmax)
            t2 = time.perf_counter()
            rng, _ = orc.estimate_range(keys, K, extr, H, W, 8, res, voxel, 0.1,
                                        dmax, frag_buffer_size=262144)
            o = orc.raycast(h, tsdf, wgt, col, rng, K, extr, H, W, res, voxel,
                            ds, 0.1, dmax, min(float(i), 3.0), trunc, 8,
                            ("depth", "color"))
            t3 = time.perf_counter()
            rd, rc = o["depth"][..., 0].copy(), o["color"]
            if i > 0:
                tt += (t1 - t0, t2 - t1, t3 - t2)
        k = max(1, min(a.cpu_frames, n) - 1)
        out["cpu_oracle_ms_track_integrate_synthesize"] = \
            [float(x) for x in tt / k * 1e3]
        out["cpu_oracle_threads"] = {"track": 1, "integrate_raycast":
                                     min(64, os.cpu_count() or 1)}
        out["cpu_oracle_frames_per_s"] = float(k / tt.sum()) if tt.sum() else 0
    return out


def mode_extract(a):
    """VoxelBlockGrid::ExtractPointCloud on the grid the C2 stream leaves
    behind (`--frames` VGA frames, 8 mm / 16^3): count pass + write pass."""
    from open3d_amd import geometry, synthetic
    W, H = 640, 480
    K = synthetic.intrinsics(W, H)
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], 0.008, 16, a.block_count)
    ds, cs, Ts = [], [], []
    for k in range(a.frames):
        d, c, _, T = synthetic.render_frames(k * a.frame_step, 1, W, H,
                                             device="cuda")
        ds.append(d[0].contiguous())
        cs.append(c[0].contiguous())
        Ts.append(T[0])
    g.integrate_frames(ds, cs, K, K, Ts, 1000.0, 3.0, 8.0)
    torch.cuda.synchronize()
    for _ in range(2):
        pcd = g.extract_point_cloud(3.0)
    n_pts = pcd["positions"].shape[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        pcd = g.extract_point_cloud(3.0)                  # 2-pass estimation
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / a.repeat * 1e3
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        pcd = g.extract_point_cloud(3.0, n_pts)           # size given
    torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) / a.repeat * 1e3
    nb = g.hashmap().size()
    out = {"mode": "extract", "active_blocks": nb, "points": n_pts,
           "ms_two_pass": ms2, "ms_with_estimate": ms1,
           "algorithmic_GBps_with_estimate":
               (nb * 4096 * 6 * 2 + n_pts * 36) / (ms1 * 1e-3) / 1e9}
    return out


def mode_normals(a):
    """PointCloud::EstimateNormals(max_nn=30, radius) on an `--points` cloud:
    index build + hybrid search (k = 30) + covariances + eigen solve."""
    from open3d_amd import registration as reg, synthetic
    p = synthetic.make_icp_pair(a.points, a.points, seed=0)
    pts = torch.from_numpy(p["target"]).cuda()
    radius = None if a.knn_only else 0.05
    for _ in range(2):
        n = reg.estimate_normals(pts, 30, radius)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        n = reg.estimate_normals(pts, 30, radius)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.repeat * 1e3
    cosang = (n.cpu().numpy() * p["target_normals"]).sum(1)
    out = {"mode": "normals", "points": a.points, "max_nn": 30,
           "radius": radius, "ms_per_call": ms,
           "points_per_s": a.points / ms * 1e3,
           "median_abs_cos_to_true_normal": float(np.median(np.abs(cosang)))}
    if not a.no_cpu and not a.knn_only:
        import _oracle as orc
        orc.set_threads(min(64, os.cpu_count() or 1))
        m = min(a.points, 20000)
        sub = np.ascontiguousarray(p["target"][:m])
        t0 = time.perf_counter()
        orc.estimate_normals(sub, radius, 30)
        dt = time.perf_counter() - t0
        out["cpu_oracle_points_per_s"] = m / dt
        out["cpu_oracle_sample_points"] = m
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["icp", "slam", "model", "normals", "extract", "both"],
                    default="both")
    ap.add_argument("--hd", action="store_true", help="1280x720 (model mode)")
    ap.add_argument("--vga", action="store_true", help="640x480 (slam mode)")
    ap.add_argument("--method", default="p2plane",
                    choices=["p2plane", "intensity", "hybrid"])
    ap.add_argument("--estimation", default="p2plane",
                    choices=["p2plane", "p2point"], help="icp mode")
    ap.add_argument("--knn-only", action="store_true",
                    help="normals mode: EstimateNormals(max_nn=30) without a "
                         "radius (KNN search)")
    ap.add_argument("--cpu-frames", type=int, default=0)
    ap.add_argument("--host-counts", action="store_true",
                    help="slam mode: read the two cloud sizes back every "
                         "frame (the round-2 loop) instead of leaving them "
                         "on the device")
    ap.add_argument("--phases", action="store_true",
                    help="slam mode: synchronise between phases and report "
                         "their times")
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--frame-step", type=int, default=4)
    ap.add_argument("--block-count", type=int, default=65536)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    def emit(o):
        print(json.dumps(o), flush=True)
    if a.mode in ("icp", "both"):
        emit(mode_icp(a))
    if a.mode in ("slam", "both"):
        emit(mode_slam(a))
    if a.mode == "model":
        emit(mode_model(a))
    if a.mode == "normals":
        emit(mode_normals(a))
    if a.mode == "extract":
        emit(mode_extract(a))


if __name__ == "__main__":
    main()
