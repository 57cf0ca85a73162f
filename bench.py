#!/usr/bin/env python
"""bench.py -- RGB-D frames/s of the MI355X dense-SLAM hot path.

Workload at N=1 (BASELINE.json configs[1]): synthetic 640x480 RGB-D frames with
known poses integrated into an 8 mm / 16^3-block VoxelBlockGrid (tsdf f32,
weight u16, colour u16 -- the slam::Model layout) on one MI355X: per frame
block touch + hash activation + per-voxel TSDF/weight/colour update.
A "step" is one batch of `--batch` frames; frames are resident in HBM before
the timed region starts. With --gpus N every rank integrates its own shard of
the stream (frames r, r+N, ...) into a private grid (weak scaling, no data-path
collective); the activated block IDs are all-gathered over RCCL once, inside
the timed region, as the closing exchange step.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
`roofline` (Integrate kernel: algorithmic bytes / HIP-event kernel time vs the
8 TB/s HBM peak) and `cpu_baseline` (Open3D's own CPU kernel bodies through oracle/_ref
when that prebuilt library is present, else the restated oracle, timed on this
host's cores on a bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOXEL = 0.008
RES = 16
TRUNC = 8.0
DEPTH_SCALE = 1000.0
DEPTH_MAX = 3.0
W, H = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# u16 grid with colour, read+write per voxel
BYTES_PER_BLOCK = 4096 * 24
IMAGE_BYTES = W * H * 2 + W * H * 3
BLOCK_HEADER_BYTES = 16  # buf index + key


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=50,
                    help="frames per step (per GPU)")
    ap.add_argument("--block-count", type=int, default=262144)
    ap.add_argument("--per-frame-calls", action="store_true",
                    help="one o3dmi_vbg_integrate_frame call per frame instead "
                         "of one o3dmi_vbg_integrate_frames call per step")
    ap.add_argument("--frames-per-launch", type=int, default=4,
                    help="consecutive frames applied per launch to register-"
                         "resident blocks (1..4); results are identical")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket every n-th integrate launch with HIP events "
                         "(0 = none; the roofline is then not measured)")
    ap.add_argument("--sharding", choices=["frames", "blocks"],
                    default="frames",
                    help="multi-GPU scheme: 'frames' = every rank integrates "
                         "its own frames into a private grid (weak scaling, "
                         "the default); 'blocks' = every rank sees the SAME "
                         "stream and integrates only the blocks it owns "
                         "(strong scaling, union of the grids bit-identical "
                         "to one GPU)")
    ap.add_argument("--merge-model", action="store_true",
                    help="frame sharding, N > 1: after the timed region fold "
                         "every other rank's blocks into this rank's grid "
                         "(sharding.merge_frame_sharded_grid) and report its "
                         "time as config.merge_ms; not part of `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth-only", action="store_true",
                    help="diagnostics: grid without colour (tsdf + weight)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the live rocprofv3 counter passes")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the ICP legs (configs[0] / configs[2])")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(frames_cpu, K, Ts, budget_s):
    """CPU path timed beside the GPU on a bounded sample of the same stream:
    touch + activate + integrate of its first frames.

    kind "reference": oracle/_ref -- Open3D's own DepthTouchCPU / IntegrateCPU
    bodies compiled from the reference sources (OpenMP stand-in for TBB's
    parallel_for), block activation through the oracle's hash map. Falls back
    to kind "port" (the restated oracle) when the prebuilt _ref is absent.
    A few thread counts are tried on the first frames and the fastest is kept
    (the reference lets TBB pick)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    import _ref as ref
    use_ref = ref.available()
    impl = ref if use_ref else orc
    cores = os.cpu_count() or 1

    def run(n_threads, frames, budget):
        impl.set_threads(n_threads)
        cap = 16384
        h = orc.HashMap(cap)
        tsdf = np.zeros((cap, RES, RES, RES), np.float32)
        wgt = np.zeros((cap, RES, RES, RES), np.uint16)
        col = np.zeros((cap, RES, RES, RES, 3), np.uint16)
        n = 0
        phase = [0.0, 0.0, 0.0]  # touch, activate + find, integrate (seconds)
        t0 = time.perf_counter()
        for (d, c), T in frames:
            ta = time.perf_counter()
            keys = impl.depth_touch(d, K, T, RES, VOXEL, VOXEL * TRUNC,
                                    DEPTH_SCALE, DEPTH_MAX)
            tb = time.perf_counter()
            h.activate(keys)
            buf, _ = h.find(keys)
            tc = time.perf_counter()
            impl.integrate(d, c, buf, h.key_buffer(), tsdf, wgt, col, K, K, T,
                           RES, VOXEL, VOXEL * TRUNC, DEPTH_SCALE, DEPTH_MAX)
            td = time.perf_counter()
            phase[0] += tb - ta
            phase[1] += tc - tb
            phase[2] += td - tc
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        return n / (time.perf_counter() - t0), n, [1e3 * x / n for x in phase]

    frames = list(zip(frames_cpu, Ts))
    cands = sorted({c for c in (8, 16, 32, 64, 128, cores) if c <= cores})
    probe = {c: run(c, frames[:6], budget_s * 0.08)[0] for c in cands}
    best = max(probe, key=probe.get)
    fps, n, phase_ms = run(best, frames, budget_s * 0.5)
    return {"value": fps, "unit": "frames/s", "cores": best,
            "kind": "reference" if use_ref else "port",
            "ms_per_frame": {"touch": phase_ms[0],
                             "activate_find": phase_ms[1],
                             "integrate": phase_ms[2]},
            "sample": "first %d frames of the same 640x480 stream (touch + "
                      "activate + integrate, u16 grid with colour); %s; best "
                      "of %s threads on a %d-thread host"
                      % (n, "Open3D DepthTouchCPU/IntegrateCPU bodies via "
                            "oracle/_ref" if use_ref else "restated oracle",
                         cands, cores)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(
            os.path.join(ROOT, "open3d_amd", "lib", "libo3d_mi355x.so")):
        ge.build()
    if dist is not None:
        dist.barrier()
    from open3d_amd import geometry, synthetic
    from open3d_amd.sharding import allgather_block_keys

    n_steps = a.steps + a.warmup
    n_local = n_steps * a.batch
    by_blocks = a.sharding == "blocks" and world > 1
    if by_blocks:
        # One stream seen by every rank; blocks are split by ownership.
        frame_ids = list(range(n_local))
    else:
        # Frame-sharded stream: rank r owns global frames r, r+world, ...
        frame_ids = [rank + world * i for i in range(n_local)]
    K = synthetic.intrinsics(W, H)
    depths, colors, Ts = [], [], []
    for i0 in range(0, n_local, 25):
        ids = frame_ids[i0:i0 + 25]
        for k in ids:
            d, c, _, T = synthetic.render_frames(k, 1, W, H, device=dev)
            depths.append(d[0].contiguous())
            colors.append(c[0].contiguous())
            Ts.append(T[0])
    torch.cuda.synchronize()

    if a.depth_only:
        g = geometry.VoxelBlockGrid(["tsdf", "weight"],
                                    [torch.float32, torch.uint16], [1, 1],
                                    VOXEL, RES, a.block_count)
        colors = None
    else:
        g = geometry.VoxelBlockGrid(["tsdf", "weight", "color"],
                                    [torch.float32, torch.uint16,
                                     torch.uint16],
                                    [1, 1, 3], VOXEL, RES, a.block_count)
    if by_blocks:
        g.set_block_ownership(rank, world)

    def run_step(s):
        lo, hi = s * a.batch, (s + 1) * a.batch
        if a.per_frame_calls:
            for i in range(lo, hi):
                g.integrate_frame(depths[i], colors[i], K, K, Ts[i],
                                  DEPTH_SCALE, DEPTH_MAX, TRUNC)
        else:
            g.integrate_frames(depths[lo:hi],
                               colors[lo:hi] if colors is not None else None,
                               K, K, Ts[lo:hi],
                               DEPTH_SCALE, DEPTH_MAX, TRUNC,
                               frames_per_launch=a.frames_per_launch)

    def barrier():
        if dist is not None:
            dist.barrier()

    for s in range(a.warmup):
        run_step(s)
    torch.cuda.synchronize()
    barrier()

    g.profile_begin(a.steps * a.batch, a.event_stride)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for s in range(a.warmup, n_steps):
        run_step(s)
    n_union = None
    if dist is not None:
        hm = g.hashmap()
        act = hm.active_buf_indices()
        keys = hm.key_tensor()[act.long()]
        n_union = int(allgather_block_keys(keys, dist).shape[0])
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    prof = g.profile_end()
    n_blocks = g.hashmap().size()
    merge_ms = None
    if a.merge_model and dist is not None and not by_blocks:
        from open3d_amd.sharding import merge_frame_sharded_grid
        barrier()
        tm = time.perf_counter()
        merge_frame_sharded_grid(g, dist)
        torch.cuda.synchronize()
        barrier()
        merge_ms = (time.perf_counter() - tm) * 1e3

    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_frames = a.steps * a.batch * (1 if by_blocks else world)
    fps = total_frames / elapsed

    # Roofline of the dominant kernel over the HIP-event-bracketed launches:
    # unit = one active block x one frame (98 304 B of voxel
    # state read+written, + 16 B header), + the frame's images once.
    launches = max(1, prof["launches"])
    alg_bytes = (prof["block_frames"] * (BYTES_PER_BLOCK + BLOCK_HEADER_BYTES)
                 + prof["frames"] * IMAGE_BYTES) / launches
    k_ms = prof["integrate_ms"] / launches
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    # HBM traffic per launch of the same kernel from the committed rocprofv3
    # PMC passes of this command (tools/profile_gpu.sh: FETCH_SIZE and
    # WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 note in
    # MI355X_MICROARCH.md); null when no summary is present.
    traffic, traffic_src = None, None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for fn in sorted(os.listdir(pdir), reverse=True):
            if fn.endswith("_hbm_traffic.json"):
                try:
                    with open(os.path.join(pdir, fn)) as f:
                        t = json.load(f).get("FrameStepKernel")
                    if t:
                        traffic = t["hbm_bytes_per_launch_corrected"]
                        traffic_src = "profiles/" + fn
                        break
                except Exception:
                    pass

    out = {
        "metric": "RGB-D frames/s (TSDF integrate into 8 mm / 16^3 "
                  "VoxelBlockGrid: touch + activate + integrate)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if by_blocks else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: %d synthetic 640x480 RGB-D frames "
                               "per GPU -> 8 mm VoxelBlockGrid(16^3), grid "
                               "(tsdf f32, weight u16, color u16), known poses"
                               % (a.steps * a.batch),
                   "frames_per_step": a.batch, "block_count": a.block_count,
                   "api": "integrate_frame per frame" if a.per_frame_calls
                          else "integrate_frames per step",
                   "frames_per_launch": 1 if a.per_frame_calls
                                        else a.frames_per_launch,
                   "active_blocks": int(n_blocks),
                   "avg_blocks_per_frame": prof["block_frames"] /
                                           max(1, prof["frames"]),
                   "sharding": ("none" if world == 1 else
                                "one stream, blocks split by ownership; "
                                "block-ID all-gather at the end" if by_blocks
                                else "frames r, r+N, ... per rank; block-ID "
                                     "all-gather at the end"),
                   "union_blocks": n_union, "merge_ms": merge_ms},
        "roofline": {"bound": "hbm", "kernel": "FrameStepKernel",
                     "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_kernel_ms": k_ms,
                     "frames_per_launch": prof["frames"] / launches,
                     "note": "frac can exceed the DRAM bound when several "
                             "frames are applied per launch: voxel state is "
                             "then read/written once per launch, not once per "
                             "frame (compare `traffic`)"},
    }
    if world == 1 and not a.no_cpu_baseline:
        nb = 64
        frames_cpu = [(depths[i].cpu().numpy(), colors[i].cpu().numpy())
                      for i in range(min(nb, len(depths)))]
        out["cpu_baseline"] = cpu_baseline(frames_cpu, K, Ts, a.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
