#!/usr/bin/env python
"""Generates tests/golden/*.npz from the REFERENCE's own kernel bodies
(oracle/_ref/libo3d_ref.so = /root/reference sources compiled through
oracle/ref_shim; `make -C oracle ref`). Run in the build container:

    python tests/golden/make_golden.py

Every file stores the inputs next to the reference outputs, so the checks in
tests/test_golden.py need neither /root/reference nor the synthetic generator.
Sizes are kept small (QQVGA frames, res-8 blocks) so the fixtures stay < 1 MB.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _oracle as orc  # noqa: E402  (only its hash map, to hand out buffer indices)
import _ref as ref  # noqa: E402
from open3d_amd import synthetic as syn  # noqa: E402

W, H = 160, 120
VOXEL, RES, TRUNC_MULT = 0.02, 8, 4.0
DEPTH_SCALE, DEPTH_MAX = 1000.0, 3.0


def vbg():
    tr = VOXEL * TRUNC_MULT
    frames = (30, 36, 42)
    d, c, K, Ts = [], [], None, []
    for k in frames:
        dd, cc, K, T = syn.render_frames(k, 1, W, H, device="cpu")
        d.append(dd[0].numpy())
        c.append(cc[0].numpy())
        Ts.append(np.asarray(T[0], dtype=np.float64))
    K = np.asarray(K, dtype=np.float64)
    cap = 1024
    h = orc.HashMap(cap)
    out = {"depth": np.stack(d), "color": np.stack(c), "K": K,
           "T": np.stack(Ts), "params": np.array(
               [VOXEL, RES, TRUNC_MULT, DEPTH_SCALE, DEPTH_MAX], np.float64)}
    for grid in ("u16",):
        wd = np.uint16 if grid == "u16" else np.float32
        tsdf = np.zeros((cap, RES, RES, RES), np.float32)
        wgt = np.zeros((cap, RES, RES, RES), wd)
        col = np.zeros((cap, RES, RES, RES, 3), wd)
        hh = orc.HashMap(cap)
        for i in range(len(frames)):
            keys = ref.depth_touch(d[i], K, Ts[i], RES, VOXEL, tr, DEPTH_SCALE,
                                   DEPTH_MAX, 4)
            if grid == "u16":
                out["touch_keys_%d" % i] = keys[np.lexsort(keys.T[::-1])]
            hh.activate(keys)
            buf, _ = hh.find(keys)
            ref.integrate(d[i], c[i], buf, hh.key_buffer(), tsdf, wgt, col, K,
                          K, Ts[i], RES, VOXEL, tr, DEPTH_SCALE, DEPTH_MAX)
        n = hh.size()
        keys_all = hh.key_buffer()[:n].copy()
        order = np.lexsort(keys_all.T[::-1])
        out["block_keys"] = keys_all[order]
        out["tsdf_" + grid] = tsdf[:n][order]
        out["weight_" + grid] = wgt[:n][order]
        out["color_" + grid] = col[:n][order]
        if grid == "u16":
            # fragment buffer large enough not to drop fragments (the
            # reference's steady state after its first call on a scene)
            rng = ref.estimate_range(keys, K, Ts[-1], H, W, 8, RES, VOXEL, 0.1,
                                     DEPTH_MAX, frag_buffer_size=65536)
            out["range_map"] = rng
            out["raycast_block_keys"] = keys
            hb, _ = hh.find(keys_all)
            rc = ref.raycast(keys_all, hb, tsdf, wgt, col, rng, K, Ts[-1], H,
                             W, RES, VOXEL, DEPTH_SCALE, 0.1, DEPTH_MAX, 1.0,
                             TRUNC_MULT, 8,
                             attrs=("depth", "vertex", "color", "normal",
                                    "mask"))
            for k, v in rc.items():
                out["raycast_" + k] = v
    np.savez_compressed(os.path.join(HERE, "vbg_qqvga_res8.npz"), **out)
    print("vbg: %d blocks, raycast hit %.2f" % (
        n, float((out["raycast_depth"] > 0).mean())))


def icp():
    out = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        p = syn.make_icp_pair(2000, 2000, seed=9, dtype=dt)
        idx, d2, cnt = orc.hybrid_search(p["target"], p["source"], 0.1, 1)
        corr = idx[:, 0].astype(np.int64)
        out["source_" + name] = p["source"]
        out["target_" + name] = p["target"]
        out["normals_" + name] = p["target_normals"]
        out["corr_" + name] = corr
        for kname, kern in (("l2", (0, 1.0, 1.0)), ("huber", (2, 0.05, 1.0)),
                            ("tukey", (5, 0.05, 1.0))):
            out["sums29_%s_%s" % (kname, name)] = ref.p2plane_accumulate(
                p["source"], p["target"], p["target_normals"], corr, *kern)
        pose, res, count = ref.compute_pose_p2plane(
            p["source"], p["target"], p["target_normals"], corr)
        out["pose_" + name] = pose
        out["count_" + name] = np.array([count])
        T = ref.pose_to_transformation(pose)
        out["T_" + name] = T
        out["transformed_" + name] = ref.transform_points(T, p["source"])
    np.savez_compressed(os.path.join(HERE, "icp_2k.npz"), **out)
    print("icp: %d correspondences" % int((out["corr_f32"] >= 0).sum()))


def odometry():
    """Row f1: image pyramid ops and the 29 sums of the three methods from
    the reference's ImageCPU / RGBDOdometryCPU bodies on a QQVGA pair."""
    d, c, K, Ts = syn.render_frames(3, 1, W, H, device="cpu", noise_sigma=0.002)
    d2, c2, _, Ts2 = syn.render_frames(5, 1, W, H, device="cpu",
                                       noise_sigma=0.002, seed=1)
    sd, td = d[0].numpy().copy(), d2[0].numpy().copy()
    sd[10:20, 30:50] = 0
    td[60:64, 100:140] = 0
    sc_, tc_ = c[0].numpy(), c2[0].numpy()
    K = np.asarray(K, np.float64)
    nan = float("nan")
    s = ref.clip_transform(sd, 1000.0, 0.0, 3.0, nan)
    t = ref.clip_transform(td, 1000.0, 0.0, 3.0, nan)
    sv = ref.create_vertex_map(s, K, nan)
    tv = ref.create_vertex_map(t, K, nan)
    # FilterBilateral / Sobel / RGBToGray have no in-tree CPU arithmetic (IPP):
    # the smoothed normals and the gradient maps are INPUTS of this fixture,
    # produced once by the oracle's IPP-semantics filters.
    tn = ref.create_normal_map(
        ref.create_vertex_map(orc.filter_bilateral(t, 5, 5, 10), K, nan), nan)
    si = ref.image_to_float(orc.rgb_to_gray(sc_), 1 / 255)
    ti = ref.image_to_float(orc.rgb_to_gray(tc_), 1 / 255)
    tdx, tdy = orc.filter_sobel(t)
    tix, tiy = orc.filter_sobel(ti)
    T = np.asarray(Ts2[0], np.float64) @ np.linalg.inv(np.asarray(Ts[0], np.float64))
    T[0, 3] += 0.004
    T[2, 3] -= 0.003
    out = {"source_depth_u16": sd, "target_depth_u16": td, "K": K, "T": T,
           "source_clip": s, "target_clip": t,
           "source_pyrdown": ref.pyrdown_depth(s, 0.14, nan),
           "source_vertex": sv, "target_vertex": tv, "target_normal": tn,
           "source_intensity": si, "target_intensity": ti,
           "target_depth_dx": tdx, "target_depth_dy": tdy,
           "target_intensity_dx": tix, "target_intensity_dy": tiy}
    L = dict(K=K, T=T, source_vertex=sv, target_vertex=tv, target_normal=tn,
             source_depth=s, target_depth=t, source_intensity=si,
             target_intensity=ti, target_depth_dx=tdx, target_depth_dy=tdy,
             target_intensity_dx=tix, target_intensity_dy=tiy)
    ref.set_threads(1)
    for m, name in ((0, "p2plane"), (1, "intensity"), (2, "hybrid")):
        delta, res, cnt, sums = ref.odometry(m, **L, depth_outlier_trunc=0.07,
                                             depth_huber_delta=0.05,
                                             intensity_huber_delta=0.1)
        out["sums29_" + name] = np.asarray(sums, np.float32)
        out["delta_" + name] = np.asarray(delta, np.float64)
        out["count_" + name] = np.array([cnt])
    out["information"] = ref.odometry_information(sv, tv, K, T, 0.07 * 0.07)
    np.savez_compressed(os.path.join(HERE, "odometry_qqvga.npz"), **out)
    print("odometry: inliers", [int(out["count_" + n][0]) for n in
                                ("p2plane", "intensity", "hybrid")])


def extract_and_normals():
    """Rows f2 / f4: ExtractPointCloudCPU on the vbg fixture's grid, and the
    covariance / eigen-solver bodies on a 1500-point cloud."""
    g = np.load(os.path.join(HERE, "vbg_qqvga_res8.npz"))
    keys = g["block_keys"]
    n = keys.shape[0]
    h = orc.HashMap(n)
    h.activate(keys)                      # buffer index i <-> sorted key i
    active = np.arange(n, dtype=np.int32)
    nbi, nbm = orc.buffer_radius_neighbors(h, active)
    ref.set_threads(1)
    pts, nrm, col, total = ref.extract_point_cloud(
        active, nbi, nbm, keys, g["tsdf_u16"], g["weight_u16"], g["color_u16"],
        RES, VOXEL, 1.0)
    out = {"points": pts, "normals": nrm, "colors": col,
           "weight_threshold": np.array([1.0])}
    p = syn.make_icp_pair(1500, 1500, seed=11, dtype=np.float32)
    cloud = p["target"]
    idx, _, cnt = orc.hybrid_search(cloud, cloud, 0.4, 30)
    cov = ref.estimate_covariances(cloud, idx, cnt)
    out.update({"cloud": cloud, "nn_idx": idx, "nn_cnt": cnt, "cov": cov,
                "cloud_normals": ref.normals_from_covariances(cov),
                "radius_max_nn": np.array([0.4, 30.0])})
    np.savez_compressed(os.path.join(HERE, "extract_normals.npz"), **out)
    print("extract: %d points; normals: %d points, %d..%d neighbours" % (
        total, cloud.shape[0], cnt.min(), cnt.max()))


if __name__ == "__main__":
    assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
    if "--new-only" not in sys.argv:
        vbg()
        icp()
    odometry()
    extract_and_normals()
