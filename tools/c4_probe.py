#!/usr/bin/env python
"""configs[4] integrate leg, launch by launch.

    python tools/c4_probe.py [--frames 200] [--fpl 4] [--passes 2] [--out f.json]

One cold pass of bench.py's configs[4] integrate leg (4 mm voxels, 2 M-block
map, 540 k blocks of ballast) with EVERY launch bracketed by HIP events, then
`passes - 1` warm passes over the same frames (no block is created any more):
per launch the event duration, distinct blocks, block-frames and the map size
the launch saw. Under `rocprofv3 --kernel-trace` / `--pmc` the same command is
the workload of profiles/r4_c4_* (tools/c4_profile.sh).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--fpl", type=int, default=4)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    import bench
    from open3d_amd import _lib, geometry, synthetic
    from open3d_amd.core import stream
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    W, H = bench.W, bench.H
    K = synthetic.intrinsics(W, H)
    ds, cs, Ts = [], [], []
    for k in range(0, a.frames * bench.C4_FRAME_STEP, bench.C4_FRAME_STEP):
        d, c, _, T = synthetic.render_frames(k, 1, W, H, device=dev)
        ds.append(d[0].contiguous())
        cs.append(c[0].contiguous())
        T2 = T[0].copy()
        T2[:3, 3] *= 2.0
        Ts.append(T2)
    t0 = time.perf_counter()
    g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                [torch.float32, torch.uint16, torch.uint16],
                                [1, 1, 3], bench.C4_VOXEL, bench.RES,
                                bench.C4_CAPACITY)
    i = torch.arange(bench.C4_BALLAST, dtype=torch.int32, device=dev)
    keys = torch.stack([100000 + i % 1000, 100000 + i // 1000,
                        torch.full_like(i, 100000)], 1).contiguous()
    _lib.check(_lib.lib().o3dmi_hash_activate(
        g.hashmap()._h, _lib.ptr(keys), bench.C4_BALLAST, None, None, None,
        stream()), "activate ballast")
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    batch = g.prepare_frames(ds, cs, K, K, Ts)
    out = {"frames": a.frames, "frames_per_launch": a.fpl,
           "setup_s": t_setup, "passes": []}
    for p in range(a.passes):
        n_launch = (a.frames + a.fpl - 1) // a.fpl
        g.profile_begin(n_launch + 8, a.stride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.integrate_frames(batch, depth_scale=bench.C4_DEPTH_SCALE,
                           depth_max=bench.C4_DEPTH_MAX,
                           trunc_voxel_multiplier=bench.TRUNC,
                           frames_per_launch=a.fpl)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = g.profile_end()
        per = g.profile_launches()
        out["passes"].append({
            "kind": "cold" if p == 0 else "warm",
            "frames_per_s": a.frames / dt, "wall_ms": dt * 1e3,
            "host_issue_ms": t_issue * 1e3,
            "event_ms_sum": prof["integrate_ms"],
            "launches": prof["launches"],
            "avg_event_ms": prof["integrate_ms"] / max(1, prof["launches"]),
            "block_frames": prof["block_frames"],
            "distinct_blocks": prof["distinct_blocks"],
            "map_size_end": int(g.hashmap().size()),
            "per_launch_ms": [round(float(x), 4) for x in per["ms"]],
            "per_launch_distinct": per["distinct_blocks"].tolist(),
            "per_launch_block_frames": per["block_frames"].tolist(),
            "per_launch_map_size": per["map_size"].tolist()})
    s = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")
    brief = {k: v for k, v in out.items() if k != "passes"}
    brief["passes"] = [{k: v for k, v in p.items()
                        if not k.startswith("per_launch")}
                       for p in out["passes"]]
    print(json.dumps(brief))


if __name__ == "__main__":
    main()
