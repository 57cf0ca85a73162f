"""RayCast micro-benchmark: a grid of `--frames` VGA (or --hd) frames, then
`--repeat` ray casts (depth + normal, the tracking loop's call) at the last
pose, timed with HIP events on the launch stream. Prints one JSON line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open3d_amd import geometry, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--repeat", type=int, default=50)
ap.add_argument("--hd", action="store_true")
ap.add_argument("--attrs", default="depth,normal")
ap.add_argument("--digest", action="store_true",
                help="add a SHA-256 of the rendered maps (A / B runs, e.g. "
                     "O3DMI_LIB=<another build>, must print the same one)")
a = ap.parse_args()
W, H = (1280, 720) if a.hd else (640, 480)
K = synthetic.intrinsics(W, H)
g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                            [torch.float32, torch.uint16, torch.uint16],
                            [1, 1, 3], 0.008, 16, 65536)
T = None
for k in range(0, a.frames * 2, 2):
    d, c, _, Ts = synthetic.render_frames(k, 1, W, H, device="cuda")
    T = Ts[0]
    g.integrate_frame(d[0].contiguous(), c[0].contiguous(), K, K, T, 1000.0,
                      3.0, 8.0)
keys, cnt = g.last_frame_block_coordinates((H // 4) * (W // 4) * 4)
attrs = tuple(a.attrs.split(","))
for _ in range(3):
    out = g.ray_cast(keys, K, T, W, H, render_attributes=attrs,
                     weight_threshold=1.0, block_count_dev=cnt)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
      for _ in range(a.repeat)]
for e0, e1 in ev:
    e0.record()
    out = g.ray_cast(keys, K, T, W, H, render_attributes=attrs,
                     weight_threshold=1.0, block_count_dev=cnt)
    e1.record()
torch.cuda.synchronize()
ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
extra = {}
if a.digest:
    import hashlib
    h = hashlib.sha256()
    for k in attrs:
        h.update(out[k].contiguous().cpu().numpy().tobytes())
    extra["sha256"] = h.hexdigest()[:16]
print(json.dumps({"size": [W, H], "attrs": attrs, "blocks": int(cnt.item()),
                  **extra,
                  "ray_cast_call_us_median": ms[len(ms) // 2] * 1e3,
                  "min": ms[0] * 1e3,
                  "valid_frac": float((out["depth"] > 0).float().mean())}))
