#!/usr/bin/env python3
"""Randomised check of MultiScaleICP's pyramid forms (GPU): the same random
registration problems -- sizes 200 .. 250 000 points, 1 to 5 scales, with and
without a finest level that is the input itself, float32 / float64, device-
resident or host cloud sizes -- run by the default build of the loop (paired
pyramids, fused next-level inserts, counts posted by the last level's reduce
launch) and by the plain one (two chains, an insert launch per level, a
posting launch: O3DMI_VDS_UNPAIRED=1 O3DMI_VDS_NO_FUSE=1
O3DMI_VDS_POST_LAUNCH=1). Every iteration's rmse, the final pose, fitness and
the iteration count must be the same bits.

  python tools/fuzz_pyramid.py [--cases 40] [--seed 1]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emit(cases, seed):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from open3d_amd import registration as reg, synthetic as syn
    rng = np.random.default_rng(seed)
    out = []
    for case in range(cases):
        r = rng.random()
        n = int(rng.integers(200, 5000)) if r < 0.3 else (
            int(rng.integers(5000, 60000)) if r < 0.8 else
            int(rng.integers(60000, 250000)))
        dtype = np.float32 if rng.random() < 0.75 else np.float64
        n_scales = int(rng.integers(1, 6))
        v0 = float(rng.choice([0.02, 0.04, 0.08]))
        voxels = [v0 / (2 ** k) for k in range(n_scales)]
        if rng.random() < 0.3:
            voxels[-1] = -1.0          # the finest level is the input itself
        p = syn.make_icp_pair(n, n, seed=int(rng.integers(1, 1 << 30)),
                              dtype=dtype)
        ns = int(rng.integers(n // 2 + 1, n + 1))
        nt = int(rng.integers(n // 2 + 1, n + 1))
        src = torch.from_numpy(p["source"]).cuda()
        tgt = torch.from_numpy(p["target"]).cuda()
        nrm = torch.from_numpy(p["target_normals"]).cuda()
        crit = [reg.ICPConvergenceCriteria(1e-6, 1e-6, 4)] * n_scales
        md = [max(3 * abs(v), 0.05) for v in voxels]
        log = []
        use_dev = rng.random() < 0.5
        try:
            r_ = one(reg, torch, src, tgt, nrm, ns, nt, voxels, crit, md, log,
                     use_dev)
        except Exception as e:   # (e.g. a singular system: the same either way)
            out.append([n, ns, nt, dtype.__name__, voxels, "error: %s" % e,
                        -1, "", "", [repr(x["inlier_rmse"]) for x in log]])
            continue
        out.append([n, ns, nt, dtype.__name__, voxels,
                    r_.transformation.tobytes().hex(), r_.num_iterations,
                    repr(r_.fitness), repr(r_.inlier_rmse),
                    [repr(e["inlier_rmse"]) for e in log]])
    print(json.dumps(out))


def one(reg, torch, src, tgt, nrm, ns, nt, voxels, crit, md, log, use_dev):
    if use_dev:
        counts = torch.tensor([ns, nt], dtype=torch.int32, device="cuda")
        return reg.multi_scale_icp(src, tgt, nrm, voxels, crit, md,
                                   device_counts=(counts[0:1], counts[1:2]),
                                   callback_after_iteration=log.append)
    return reg.multi_scale_icp(src[:ns].contiguous(), tgt[:nt].contiguous(),
                               nrm[:nt].contiguous(), voxels, crit, md,
                               callback_after_iteration=log.append)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emit", action="store_true")
    a = ap.parse_args()
    if a.emit:
        return emit(a.cases, a.seed)

    def run(extra):
        env = dict(os.environ)
        for k in ("O3DMI_VDS_UNPAIRED", "O3DMI_VDS_NO_FUSE",
                  "O3DMI_VDS_POST_LAUNCH"):
            env.pop(k, None)
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__),
                            "--emit", "--cases", str(a.cases), "--seed",
                            str(a.seed)], env=env, capture_output=True,
                           text=True)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(2)
        return json.loads(r.stdout.strip().splitlines()[-1])

    fast = run({})
    plain = run({"O3DMI_VDS_UNPAIRED": "1", "O3DMI_VDS_NO_FUSE": "1",
                 "O3DMI_VDS_POST_LAUNCH": "1"})
    mid = run({"O3DMI_VDS_POST_LAUNCH": "1"})
    bad = 0
    for k, (x, y, z) in enumerate(zip(fast, plain, mid)):
        if x != y or x != z:
            bad += 1
            print("MISMATCH case %d: n %d ns %d nt %d %s voxels %s" %
                  (k, x[0], x[1], x[2], x[3], x[4]))
    print("fuzz_pyramid: %d cases, %d mismatches; iterations %d..%d" %
          (len(fast), bad, min(c[6] for c in fast), max(c[6] for c in fast)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
