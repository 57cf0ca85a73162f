"""Micro-benchmark of o3dmi_voxel_down_sample (the multi-scale ICP pyramid)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from open3d_amd import registration as reg, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 230400
p = synthetic.make_icp_pair(n, n, seed=0)
pts = torch.from_numpy(p["target"]).cuda()
nrm = torch.from_numpy(p["target_normals"]).cuda()
out = {}
for voxel in (0.0125, 0.025, 0.05):
    for _ in range(3):
        q, qn = reg.voxel_down_sample(pts, nrm, voxel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        q, qn = reg.voxel_down_sample(pts, nrm, voxel)
    torch.cuda.synchronize()
    out["voxel_%g" % voxel] = {"ms": (time.perf_counter() - t0) / 20 * 1e3,
                               "out": int(q.shape[0])}
print(json.dumps({"mode": "vds", "points": n, **out}))
