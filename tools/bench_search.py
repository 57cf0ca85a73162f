"""Kernel-level timing of the fused ICP search launch (search + final sum) and
of VoxelDownSample on the clouds of a tracking frame: one VGA (or 720p) frame
of the synthetic stream -> frame cloud / model cloud by Unproject (stride 2)
-> the 5 / 2.5 / 1.25 cm pyramid levels. For every level the search launch is
timed (HIP events around 50 launches) for each lanes-per-query setting G.

    python tools/bench_search.py [--hd]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hd", action="store_true")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    from open3d_amd import _lib, registration as reg, synthetic
    from open3d_amd.core import stream
    L = _lib.lib()
    W, H = (1280, 720) if a.hd else (640, 480)
    K = synthetic.intrinsics(W, H)
    d, c, _, T = synthetic.render_frames(0, 2, W, H, device="cuda")
    stride = 2
    npx = (H // stride) * (W // stride)

    def cloud(depth, Twc):
        pts = torch.empty((npx, 3), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        Tm = np.ascontiguousarray(Twc, dtype=np.float64)
        _lib.check(L.o3dmi_unproject(
            _lib.ptr(depth.contiguous()), _lib.U16, H, W, None, _lib.ptr(pts),
            None, _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Tm), C.c_float(1000.0),
            C.c_float(3.0), C.c_int64(stride), stream()), "unproject")
        return pts[:int(cnt.item())].clone()

    src0 = cloud(d[1], T[0])   # next frame at the previous pose
    tgt0 = cloud(d[0], T[0])
    nrm0 = reg.estimate_normals(tgt0, 30, 0.05)
    out = {"width": W, "height": H, "levels": []}

    def timed(fn, reps):
        for _ in range(5):
            fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3  # us

    # VoxelDownSample, level by level as the pyramid chains them
    s, t, tn = src0, tgt0, nrm0
    levels = []
    for v in (0.0125, 0.025, 0.05):
        us = timed(lambda: reg.voxel_down_sample(t, tn, v), 20)
        s2, _ = reg.voxel_down_sample(s, None, v)
        t2, tn2 = reg.voxel_down_sample(t, tn, v)
        levels.append((v, s2, t2, tn2, us, t.shape[0]))
        s, t, tn = s2, t2, tn2
    # the raw clouds as one more level (a single-scale ICP on full clouds)
    levels.insert(0, (0.004, src0, tgt0, nrm0, 0.0, 0))
    for (v, s, t, tn, vds_us, n_in) in reversed(levels):
        md = 3.0 * v
        nns = C.c_void_p()
        _lib.check(L.o3dmi_nns_create(_lib.ptr(t), t.shape[0], _lib.F32,
                                      C.c_double(md), stream(),
                                      C.byref(nns)), "nns_create")
        sums = torch.zeros(32, dtype=torch.float64, device="cuda")
        row = {"voxel": v, "source_points": int(s.shape[0]),
               "target_points": int(t.shape[0]),
               "vds_us_incl_readback": round(vds_us, 1),
               "vds_input_points": int(n_in), "search_us": {}}
        # (rounds 2-4 swept the lanes per query here through an environment
        # switch; the library now picks 8 / 16 / 32 by size -- the measured
        # table is profiles/r2w_search_*.json)
        _lib.check(L.o3dmi_icp_search_accumulate(
            nns, _lib.ptr(s), _lib.ptr(tn), s.shape[0], 0, C.c_double(1.0),
            C.c_double(1.0), None, _lib.ptr(sums), stream()), "search")

        def launch_auto():
            _lib.check(L.o3dmi_icp_search_accumulate(
                nns, _lib.ptr(s), None, s.shape[0], 0, C.c_double(1.0),
                C.c_double(1.0), None, _lib.ptr(sums), stream()), "search")
        row["search_us"]["auto"] = round(timed(launch_auto, a.reps), 1)
        torch.cuda.synchronize()
        L.o3dmi_nns_destroy(nns)
        out["levels"].append(row)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
