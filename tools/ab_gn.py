"""A/B of the in-launch Gauss-Newton step against the host-solved iteration
(diagnostics; prints the differences)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open3d_amd import registration as reg, synthetic as syn

for dtype in (np.float32, np.float64):
    p = syn.make_icp_pair(60000, 60000, seed=11, dtype=dtype)
    src = torch.from_numpy(p["source"]).cuda()
    tgt = torch.from_numpy(p["target"]).cuda()
    nrm = torch.from_numpy(p["target_normals"]).cuda()
    cases = [([-1.0], [reg.ICPConvergenceCriteria(1e-6, 1e-6, 30)], [0.07]),
             ([0.05, 0.025, 0.0125],
              [reg.ICPConvergenceCriteria(1e-6, 1e-6, n) for n in (20, 10, 5)],
              [0.15, 0.075, 0.0375])]
    for vs, crit, md in cases:
        def run():
            log = []
            r = reg.multi_scale_icp(src.clone(), tgt, nrm, vs, crit, md,
                                    callback_after_iteration=lambda d: log.append(
                                        (d["scale_index"], d["scale_iteration_index"], d["fitness"], d["inlier_rmse"])))
            torch.cuda.synchronize()
            return r, log
        f, fl = run()
        f2, fl2 = run()
        os.environ["O3DMI_ICP_HOST_SOLVE"] = "1"
        s, sl = run()
        del os.environ["O3DMI_ICP_HOST_SOLVE"]
        print(dtype.__name__, vs, "iters", f.num_iterations, f2.num_iterations, s.num_iterations,
              "dT fast-fast", np.abs(f.transformation - f2.transformation).max(),
              "dT fast-slow", np.abs(f.transformation - s.transformation).max(),
              "fit", f.fitness, s.fitness, "rmse", f.inlier_rmse, s.inlier_rmse)
        for a, b in list(zip(fl, sl))[:6]:
            print("   ", a, b)
