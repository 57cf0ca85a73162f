#!/usr/bin/env python
"""The drop-in call shapes, one frame per call, timed alone (bench.py's
`drop_in` legs with the host side taken apart):

    api_two_step     get_unique_block_coordinates (one host wait for the
                     count, as upstream) + integrate
    integrate_frame  the fused one-call form

Per leg: frames/s, and the host time spent INSIDE the native calls per frame
(ctypes call to return) against the whole Python iteration -- what the Python
mirror adds on top of the C ABI. Run it under `rocprofv3 --kernel-trace
--stats` for the launches per frame and their durations (tools/kstats.py).

    python tools/bench_api.py [frames=300] [leg=both|two_step|fused]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from open3d_amd import _lib, geometry, synthetic  # noqa: E402

W, H = 640, 480
VOXEL, RES, TRUNC = 0.008, 16, 8.0
DEPTH_SCALE, DEPTH_MAX = 1000.0, 3.0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    leg = sys.argv[2] if len(sys.argv) > 2 else "both"
    dev = "cuda"
    K = synthetic.intrinsics(W, H)
    ds, cs, Ts = [], [], []
    for k in range(n):
        d, c, _, T = synthetic.render_frames(k, 1, W, H, device=dev)
        ds.append(d[0].contiguous())
        cs.append(c[0].contiguous())
        Ts.append(T[0])
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], VOXEL, RES, 50000)
    g.integrate_frames(ds, cs, K, K, Ts, DEPTH_SCALE, DEPTH_MAX, TRUNC)
    torch.cuda.synchronize()

    # host time inside the native calls: wrap the ctypes functions
    L = _lib.lib()
    inside = {"t": 0.0, "calls": 0}

    def wrap(name):
        fn = getattr(L, name)

        def timed(*a):
            t = time.perf_counter()
            r = fn(*a)
            inside["t"] += time.perf_counter() - t
            inside["calls"] += 1
            return r
        return fn, timed

    def two_step(i):
        bc = g.compute_unique_block_coordinates(ds[i], K, Ts[i], DEPTH_SCALE,
                                                DEPTH_MAX, TRUNC)
        g.integrate(bc, ds[i], cs[i], K, K, Ts[i], DEPTH_SCALE, DEPTH_MAX,
                    TRUNC)

    def fused(i):
        g.integrate_frame(ds[i], cs[i], K, K, Ts[i], DEPTH_SCALE, DEPTH_MAX,
                          TRUNC)

    out = {}
    legs = [("api_two_step", two_step,
             ["o3dmi_vbg_get_unique_block_coordinates",
              "o3dmi_vbg_integrate_blocks"]),
            ("integrate_frame", fused, ["o3dmi_vbg_integrate_frame"])]
    for name, fn, natives in legs:
        if leg not in ("both", name, {"api_two_step": "two_step",
                                      "integrate_frame": "fused"}[name]):
            continue
        fn(0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        plain = time.perf_counter() - t
        # second pass with the native calls timed
        saved = []
        for nm in natives:
            if not hasattr(L, nm):
                continue
            orig, timed = wrap(nm)
            saved.append((nm, orig))
            setattr(L, nm, timed)
        inside["t"], inside["calls"] = 0.0, 0
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        instr = time.perf_counter() - t
        for nm, orig in saved:
            setattr(L, nm, orig)
        out[name] = {
            "frames_per_s": round(n / plain, 1),
            "us_per_frame": round(plain / n * 1e6, 2),
            "us_per_frame_instrumented": round(instr / n * 1e6, 2),
            "us_inside_native_calls": round(inside["t"] / n * 1e6, 2),
            "native_calls_per_frame": round(inside["calls"] / n, 2),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
