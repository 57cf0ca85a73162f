#!/usr/bin/env python3
"""Randomised VoxelDownSample check (GPU): clouds of random sizes and shapes --
uniform boxes, tight clusters (crowded buckets: the reduce launch's multi-pass
path), a few voxels holding everything, lines, lattice points on voxel faces,
negative coordinates, shuffled and coherent orders, float32 and float64 -- each
compared with the oracle bit for bit, positions and the averaged attribute.

  python tools/fuzz_vds.py [--cases 200] [--seed 1] [--max-points 400000]

Prints one line per failure and a summary; exit status 1 on any mismatch. The
oracle (oracle/) is the checker here, as in tests/."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_cloud(rng, n, dtype):
    kind = rng.integers(0, 8)
    if kind == 0:      # uniform box
        ext = rng.uniform(0.2, 8.0)
        p = rng.uniform(-ext, ext, (n, 3))
    elif kind == 1:    # a few tight clusters
        k = int(rng.integers(1, 6))
        c = rng.uniform(-2, 2, (k, 3))
        p = c[rng.integers(0, k, n)] + rng.normal(0, rng.uniform(0.002, 0.05),
                                                  (n, 3))
    elif kind == 2:    # everything in a handful of voxels
        p = rng.uniform(0, 0.04, (n, 3)) + rng.integers(-3, 3, (1, 3)) * 0.5
    elif kind == 3:    # a line (long runs of equal voxels)
        t = np.sort(rng.uniform(-3, 3, n))
        d = rng.normal(0, 1, 3)
        p = t[:, None] * d[None, :] + rng.normal(0, 1e-3, (n, 3))
    elif kind == 4:    # lattice points: coordinates on voxel faces
        p = rng.integers(-40, 40, (n, 3)) * 0.05
    elif kind == 5:    # a surface (depth-image-like), coherent order
        u = np.sort(rng.uniform(-1.5, 1.5, n))
        v = rng.uniform(-1, 1, n)
        p = np.stack([u, v, 2 + 0.2 * np.sin(3 * u) * np.cos(2 * v)], 1)
    elif kind == 6:    # far from the origin, negative side
        p = rng.uniform(-1, 1, (n, 3)) - np.array([300.0, 120.0, 77.0])
    else:              # duplicates of a small set
        base = rng.uniform(-1, 1, (max(1, n // 50), 3))
        p = base[rng.integers(0, base.shape[0], n)]
    if rng.random() < 0.4:
        p = p[rng.permutation(n)]
    return np.ascontiguousarray(p.astype(dtype)), int(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-points", type=int, default=400000)
    a = ap.parse_args()
    import torch
    import _oracle as orc
    from open3d_amd import registration as reg
    rng = np.random.default_rng(a.seed)
    bad = 0
    t0 = time.time()
    sizes = []
    for case in range(a.cases):
        r = rng.random()
        if r < 0.3:
            n = int(rng.integers(1, 3000))
        elif r < 0.8:
            n = int(rng.integers(3000, 120000))
        else:
            n = int(rng.integers(120000, a.max_points))
        dtype = np.float32 if rng.random() < 0.75 else np.float64
        pts, kind = make_cloud(rng, n, dtype)
        attr = None
        if rng.random() < 0.6:
            attr = np.ascontiguousarray(
                rng.normal(0, 1, (n, 3)).astype(dtype))
        voxel = float(rng.choice([0.005, 0.0125, 0.02, 0.05, 0.1, 0.3, 1.0]))
        wp, wn = orc.voxel_down_sample(pts, attr, voxel)
        tn = None if attr is None else torch.from_numpy(attr).cuda()
        gp, gn = reg.voxel_down_sample(torch.from_numpy(pts).cuda(), tn, voxel)
        ok = gp.shape[0] == wp.shape[0] and np.array_equal(gp.cpu().numpy(), wp)
        if ok and attr is not None:
            ok = np.array_equal(gn.cpu().numpy(), wn)
        sizes.append((n, wp.shape[0]))
        if not ok:
            bad += 1
            print("MISMATCH case %d: kind %d n %d dtype %s voxel %g attr %s "
                  "(voxels %d vs %d)" % (case, kind, n, dtype.__name__, voxel,
                                         attr is not None, gp.shape[0],
                                         wp.shape[0]), flush=True)
    print("fuzz_vds: %d cases, %d mismatches, %.0f s; points %d..%d, voxels "
          "%d..%d" % (a.cases, bad, time.time() - t0,
                      min(s[0] for s in sizes), max(s[0] for s in sizes),
                      min(s[1] for s in sizes), max(s[1] for s in sizes)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
