"""Times the frame-sharded merge step on one GPU: export_blocks of a grid that
holds `--frames` VGA frames and merge_blocks of that payload into a second grid
of the same stream's other half. Prints one JSON line."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from open3d_amd import geometry, synthetic
    W, H = 640, 480
    K = synthetic.intrinsics(W, H)
    grids = []
    for half in range(2):
        g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                    [torch.float32, torch.uint16, torch.uint16],
                                    [1, 1, 3], 0.008, 16, 65536)
        ds, cs, Ts = [], [], []
        for k in range(half, a.frames, 2):
            d, c, _, T = synthetic.render_frames(k, 1, W, H, device="cuda")
            ds.append(d[0].contiguous())
            cs.append(c[0].contiguous())
            Ts.append(T[0])
        g.integrate_frames(ds, cs, K, K, Ts, 1000.0, 3.0, 8.0)
        grids.append(g)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    export_ms, merge_ms = [], []
    for _ in range(a.reps + 1):
        ev[0].record()
        keys, vals = grids[1].export_blocks()
        ev[1].record()
        grids[0].merge_blocks(keys, vals)
        ev[2].record()
        torch.cuda.synchronize()
        export_ms.append(ev[0].elapsed_time(ev[1]))
        merge_ms.append(ev[1].elapsed_time(ev[2]))
    n = int(keys.shape[0])
    voxels = n * 4096
    # u16 grid: tsdf 4 + weight 2 + colour 6 = 12 B per voxel; merge reads both
    # sides and writes one, export reads and writes one row each
    merge_bytes = voxels * 36
    export_bytes = voxels * 24
    em, mm = min(export_ms[1:]), min(merge_ms[1:])
    print(json.dumps({
        "tool": "bench_merge", "frames": a.frames, "blocks": n,
        "payload_MB": voxels * 12 / 1e6,
        "export_ms": em, "export_GBps": export_bytes / em / 1e6,
        "merge_ms": mm, "merge_GBps": merge_bytes / mm / 1e6}))


if __name__ == "__main__":
    main()
