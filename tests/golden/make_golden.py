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


if __name__ == "__main__":
    assert ref.available(), "build oracle/_ref first (make -C oracle ref)"
    vbg()
    icp()
