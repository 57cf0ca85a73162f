#!/usr/bin/env python3
"""Randomised check of the fused ICP search (GPU): EvaluateRegistration's
correspondences, fitness and rmse -- the launch of the Gauss-Newton loop, with
its 8 / 16 / 32-lane forms and the 8-lane form's pruned cell scan -- against
the oracle on random target clouds (the shapes of tools/fuzz_vds.py), queries
scattered around them at up to a few radii, random radii, float32 / float64
and a random rigid transformation applied inside the launch. A differing
index is accepted only as a TIE (both neighbours at the same float distance:
nanoflann's tie rule is not pinned, README); anything else is a mismatch.

  python tools/fuzz_search.py [--cases 150] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    import _oracle as orc
    from fuzz_vds import make_cloud
    from open3d_amd import registration as reg
    rng = np.random.default_rng(a.seed)
    bad = ties = 0
    t0 = time.time()
    lanes = {8: 0, 16: 0, 32: 0}
    for case in range(a.cases):
        dtype = np.float32 if rng.random() < 0.75 else np.float64
        nt = int(rng.integers(50, 150000))
        tgt, kind = make_cloud(rng, nt, dtype)
        r = rng.random()
        nq = int(rng.integers(1, 4000)) if r < 0.3 else (
            int(rng.integers(4000, 20000)) if r < 0.6 else
            int(rng.integers(20000, 120000)))
        radius = float(rng.choice([0.01, 0.0375, 0.075, 0.15, 0.4]))
        # queries: target points + noise of up to ~2 radii, some far away
        q = tgt[rng.integers(0, nt, nq)].astype(np.float64)
        q += rng.normal(0, radius * rng.uniform(0.05, 1.0), (nq, 3))
        far = rng.random(nq) < 0.05
        q[far] += rng.uniform(-5, 5, (int(far.sum()), 3))
        # the launch moves the source by T first: hand it T^-1 q
        ang = rng.uniform(-0.2, 0.2, 3)
        cx, cy, cz = np.cos(ang)
        sx, sy, sz = np.sin(ang)
        R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @
             np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
             np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        src = np.ascontiguousarray(((q - T[:3, 3]) @ R).astype(dtype))
        lanes[32 if nq * 64 <= 256 * 1024 else
              (16 if nq * 16 <= 256 * 1024 else 8)] += 1
        want = orc.evaluate_registration(src, tgt, radius, T)
        got = reg.evaluate_registration(torch.from_numpy(src).cuda(),
                                        torch.from_numpy(tgt).cuda(), radius,
                                        T)
        gc = got.correspondence_set.cpu().numpy()
        wc = want["correspondences"]
        diff = np.nonzero(gc != wc)[0]
        ok = True
        if diff.size:
            # ties only: both found, same distance from the moved query
            moved = orc.transform_points(T, src)
            both = (gc[diff] >= 0) & (wc[diff] >= 0)
            if not both.all():
                ok = False
            else:
                dg = ((moved[diff] - tgt[gc[diff]]) ** 2).sum(1)
                dw = ((moved[diff] - tgt[wc[diff]]) ** 2).sum(1)
                ok = bool((dg == dw).all())
                ties += int(diff.size)
        if ok and not diff.size:
            # (the reference sums the squared distances in the tensor dtype,
            # Registration.cpp:44-45: ~1e-5 of order-dependent rounding in
            # Float32; the library sums them in Float64)
            tol = 1e-4 if dtype == np.float32 else 1e-9
            ok = (got.fitness == want["fitness"] and
                  abs(got.inlier_rmse - want["inlier_rmse"]) <=
                  tol * max(1e-12, want["inlier_rmse"]))
        if not ok:
            bad += 1
            print("MISMATCH case %d: kind %d nt %d nq %d %s radius %g: %d "
                  "indices differ; fitness %r / %r, rmse %r / %r" %
                  (case, kind, nt, nq, dtype.__name__, radius, diff.size,
                   got.fitness, want["fitness"], got.inlier_rmse,
                   want["inlier_rmse"]), flush=True)
    print("fuzz_search: %d cases, %d mismatches, %d tied indices, %.0f s; "
          "lane forms %s" % (a.cases, bad, ties, time.time() - t0, lanes))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
