"""Parity at BASELINE.json's full configurations (GPU through the C ABI vs the
CPU path on the same inputs):

  configs[1]  1000 synthetic 640x480 RGB-D frames into an 8 mm / 16^3 grid:
              the whole stream through integrate_frames (8 frames per launch)
              vs Open3D's own DepthTouchCPU / IntegrateCPU bodies (oracle/_ref)
              frame by frame -- whole grid bit for bit.
  configs[2]  1280x720 tracking loop: ray cast (model cloud) -> Unproject
              (frame cloud) -> MultiScaleICP (5 / 2.5 / 1.25 cm; 20 / 10 / 5
              iterations) -> Integrate, 20 frames, operator by operator in
              lock-step with the oracle: pose <= 1e-6 rad / 1e-5 m per frame vs
              the float64-accumulating oracle, block sets and the final grid
              bit for bit, ray-cast maps <= 1e-4 with exact masks.
  configs[3]  8 ranks x 1280x720, both sharding schemes, every rank run on
              this one device: block ownership (8 classes, 56 frames: the
              union of the 8 grids is the single grid bit for bit) and frame
              sharding closed by the library's owner-partitioned exchange
              (o3dmi_vbg_merge_frame_sharded over an in-process transport,
              one host thread per rank): keys / weights exact, TSDF <= 1e-4,
              colour bounded per voxel by its observation count.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as orc
import _scene as sc
from test_vbg_gpu import OracleGrid, _compare_grids, _mk_grid

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import _lib, geometry
    return _lib, geometry


@pytest.mark.timeout(1500)
def test_configs1_full_length_1000_frames_vs_reference_cpu_bodies():
    """BASELINE configs[1] at full length. Frames are rendered once (on the
    GPU, float64 closed form) and the same uint16 / uint8 images feed both
    sides."""
    _lib, geometry = _gpu()
    import _ref as ref
    from open3d_amd import synthetic as syn
    impl = ref if ref.available() else orc
    impl.set_threads(16)
    n, cap = 1000, 16384
    g = _mk_grid(geometry, False, block_count=cap)
    og = OracleGrid(False, cap)
    K = syn.intrinsics(640, 480)
    dts, cts, Ts = [], [], []
    for k0 in range(0, n, 50):
        d, c, _, T = syn.render_frames(k0, 50, 640, 480, device="cuda")
        dn, cn = d.cpu().numpy(), c.cpu().numpy()
        for i in range(50):
            keys = impl.depth_touch(dn[i], K, T[i], sc.RES, sc.VOXEL,
                                    sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                                    sc.DEPTH_MAX, 4)
            og.h.activate(keys)
            buf, _ = og.h.find(keys)
            impl.integrate(dn[i], cn[i], buf, og.h.key_buffer(), og.tsdf,
                           og.weight, og.color, K, K, T[i], sc.RES, sc.VOXEL,
                           sc.VOXEL * sc.TRUNC_MULT, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX)
            dts.append(d[i].contiguous())
            cts.append(c[i].contiguous())
            Ts.append(T[i])
    # the bench's call pattern: prepared argument blocks (prepare_frames),
    # 12 frames per launch (bench.py's default group); 250 frames per native
    # call here. The map (16 384 blocks for a stream that creates 5 745) is far
    # below the frustum bound of one group (12 x 9 353): the groups are issued
    # on the estimate of what they add and no Reserve happens.
    for lo in range(0, n, 250):
        batch = g.prepare_frames(dts[lo:lo + 250], cts[lo:lo + 250], K, K,
                                 Ts[lo:lo + 250])
        g.integrate_frames(batch, depth_scale=sc.DEPTH_SCALE,
                           depth_max=sc.DEPTH_MAX,
                           trunc_voxel_multiplier=sc.TRUNC_MULT,
                           frames_per_launch=12)
    err, exact = _compare_grids(og, g)
    assert exact, err
    assert g.hashmap().capacity() == cap
    assert og.weight.max() >= 150
    print("configs[1]: 1000 frames, %d blocks, max weight %d, whole grid "
          "bit-exact vs %s" % (og.h.size(), int(og.weight.max()),
                               "oracle/_ref" if impl is ref else "oracle"))


def _pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    skew = d[:3, :3] - d[:3, :3].T
    ang = float(np.linalg.norm([skew[2, 1], skew[0, 2], skew[1, 0]]) / 2)
    return ang, float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))


@pytest.mark.timeout(2400)
def test_configs2_720p_tracking_loop_in_lock_step_with_the_oracle():
    """BASELINE configs[2] (the loop tools/bench_slam.py --mode slam and
    bench.py's secondary line time), 1280x720, 20 tracked frames."""
    _lib, geometry = _gpu()
    from open3d_amd import registration as reg, synthetic as syn
    from open3d_amd.core import stream
    L = _lib.lib()
    W, H, n, step = 1280, 720, 21, 2
    ds, dmax, trunc, stride = sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT, 2
    K = syn.intrinsics(W, H)
    depths, colors, Tgt = [], [], []
    for k in range(n):
        d, c, _, T = syn.render_frames(k * step, 1, W, H, device="cuda")
        depths.append(d[0].contiguous())
        colors.append(c[0].contiguous())
        Tgt.append(np.array(T[0]))
    cap = 16384
    g = _mk_grid(geometry, False, block_count=cap)
    og = OracleGrid(False, cap)
    orc.set_threads(64)
    vs = [0.05, 0.025, 0.0125]
    crit = [(1e-6, 1e-6, 20), (1e-6, 1e-6, 10), (1e-6, 1e-6, 5)]
    md = [0.15, 0.075, 0.0375]
    npx = (H // stride) * (W // stride)
    pts_buf = torch.empty((npx, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    mpts = torch.empty((npx, 3), dtype=torch.float32, device="cuda")
    mnrm = torch.empty_like(mpts)
    mcnt = torch.zeros(1, dtype=torch.int32, device="cuda")

    # bootstrap: frame 0 at its true pose, both sides
    g.integrate_frame(depths[0], colors[0], K, K, Tgt[0], ds, dmax, trunc)
    og.integrate(depths[0].cpu().numpy(), colors[0].cpu().numpy(), K, Tgt[0])
    T_prev = Tgt[0]
    depth_pred = depths[0]
    worst = {"pose": (0.0, 0.0), "depth": 0.0, "normal": 0.0}
    iters, step_err = [], []
    for k in range(1, n):
        # ---- predict: ray cast at the previous pose -------------------------
        keys = g.compute_unique_block_coordinates(depth_pred, K, T_prev, ds,
                                                  dmax, trunc)
        okeys = orc.depth_touch(depth_pred.cpu().numpy(), K, T_prev, sc.RES,
                                sc.VOXEL, sc.VOXEL * trunc, ds, dmax, 4)
        assert np.array_equal(sc.sort_rows(keys.cpu().numpy()),
                              sc.sort_rows(okeys)), k
        out = g.ray_cast(keys, K, T_prev, W, H,
                         render_attributes=("depth", "normal"),
                         depth_scale=ds, depth_min=0.1, depth_max=dmax,
                         weight_threshold=1.0, trunc_voxel_multiplier=trunc)
        rng, _ = orc.estimate_range(okeys, K, T_prev, H, W, 8, sc.RES,
                                    sc.VOXEL, 0.1, dmax,
                                    frag_buffer_size=1 << 20)
        want_rc = orc.raycast(og.h, og.tsdf, og.weight, og.color, rng, K,
                              T_prev, H, W, sc.RES, sc.VOXEL, ds, 0.1, dmax,
                              1.0, trunc, 8, ("depth", "normal"))
        gd = out["depth"].cpu().numpy()
        gn = out["normal"].cpu().numpy()
        assert np.array_equal(gd > 0, want_rc["depth"] > 0), k
        assert (gd > 0).mean() > 0.5
        # depth map is in depth_scale units (x1000): 1e-4 m = 0.1
        e_d = float(np.abs(gd - want_rc["depth"]).max()) / ds
        e_n = float(np.abs(gn - want_rc["normal"]).max())
        assert e_d <= 1e-4 and e_n <= 1e-4, (k, e_d, e_n)
        worst["depth"] = max(worst["depth"], e_d)
        worst["normal"] = max(worst["normal"], e_n)
        # ---- model cloud / frame cloud (Unproject, stride 2) ----------------
        Tp = np.ascontiguousarray(T_prev, dtype=np.float64)
        _lib.check(L.o3dmi_unproject(
            _lib.ptr(out["depth"]), _lib.F32, H, W, _lib.ptr(out["normal"]),
            _lib.ptr(mpts), _lib.ptr(mnrm), _lib.ptr(mcnt), _lib.f64p(K),
            _lib.f64p(Tp), C.c_float(ds), C.c_float(dmax), C.c_int64(stride),
            stream()), "unproject")
        m = int(mcnt.item())
        Tinv = np.ascontiguousarray(np.linalg.inv(T_prev), dtype=np.float64)
        _lib.check(L.o3dmi_transform_normals(_lib.f64p(Tinv), _lib.ptr(mnrm),
                                             m, _lib.F32, stream()),
                   "transform_normals")
        tp, tn = mpts[:m], mnrm[:m]
        _lib.check(L.o3dmi_unproject(
            _lib.ptr(depths[k]), _lib.U16, H, W, None, _lib.ptr(pts_buf), None,
            _lib.ptr(cnt), _lib.f64p(K), _lib.f64p(Tp), C.c_float(ds),
            C.c_float(dmax), C.c_int64(stride), stream()), "unproject")
        src = pts_buf[:int(cnt.item())]
        # the frame cloud is the reference's UnprojectCPU point set
        want_src, _ = orc.unproject(depths[k].cpu().numpy(), None, K, T_prev,
                                    ds, dmax, stride)
        assert np.array_equal(sc.sort_rows(src.cpu().numpy()),
                              sc.sort_rows(want_src)), k
        # ---- track: MultiScaleICP, oracle on the very same clouds -----------
        r = reg.multi_scale_icp(
            src, tp, tn, vs, [reg.ICPConvergenceCriteria(*c) for c in crit],
            md)
        want = orc.multiscale_icp(src.cpu().numpy(), tp.cpu().numpy(),
                                  tn.cpu().numpy(), vs, crit, md,
                                  accumulate_double=True)
        assert want["status"] == 0
        e = _pose_err(want["transformation"], r.transformation)
        assert e[0] <= 1e-6 and e[1] <= 1e-5, (k, e)
        assert r.num_iterations == want["num_iterations"], k
        assert abs(r.inlier_rmse - want["inlier_rmse"]) < 1e-9
        assert abs(r.fitness - want["fitness"]) < 1e-12
        worst["pose"] = (max(worst["pose"][0], e[0]),
                         max(worst["pose"][1], e[1]))
        # ---- integrate at the estimated pose, both sides --------------------
        T_k = T_prev @ np.linalg.inv(r.transformation)
        # tracking health: the estimated frame-to-frame motion against the
        # true one (the stream is noise-free, so this is the tracker's own
        # error -- the reference's, the run being in lock-step with it)
        iters.append(r.num_iterations)
        step_err.append(_pose_err(Tgt[k] @ np.linalg.inv(Tgt[k - 1]),
                                  T_k @ np.linalg.inv(T_prev)))
        g.integrate_frame(depths[k], colors[k], K, K, T_k, ds, dmax, trunc)
        og.integrate(depths[k].cpu().numpy(), colors[k].cpu().numpy(), K, T_k)
        T_prev = T_k
        depth_pred = depths[k]
    err, exact = _compare_grids(og, g)
    assert exact, err
    drift = _pose_err(Tgt[-1], T_prev)
    print("configs[2] 720p lock-step, %d tracked frames: worst pose error vs "
          "the float64 oracle %.3g rad / %.3g m; ray-cast depth %.3g m, "
          "normal %.3g; %d blocks, grid bit-exact; drift vs ground truth "
          "%.3g rad / %.3g m"
          % (n - 1, worst["pose"][0], worst["pose"][1], worst["depth"],
             worst["normal"], og.h.size(), drift[0], drift[1]))
    sa = np.array([e[0] for e in step_err])
    st = np.array([e[1] for e in step_err])
    print("  tracking health: per-frame motion error mean %.3g rad / %.3g m, "
          "max %.3g rad / %.3g m; ICP iterations per frame %s (mean %.1f)"
          % (sa.mean(), st.mean(), sa.max(), st.max(), iters,
             float(np.mean(iters))))
    # per frame, not only at the end: a tracker that loses 1 mrad / 2.5 mm a
    # frame on a noise-free stream is converging on a biased model
    assert sa.mean() <= 1.2e-3 and st.mean() <= 2.5e-3, (sa.mean(), st.mean())
    assert sa.max() <= 4e-3 and st.max() <= 8e-3, (sa.max(), st.max())
    assert drift[0] < 0.03 and drift[1] < 0.06
    assert 5 <= np.mean(iters) <= 25


def _track_vga(n_frames):
    """A short VGA tracking loop (predict -> Unproject x2 -> MultiScaleICP ->
    integrate at the estimate); returns the ICP transformations and the grid."""
    _lib, geometry = _gpu()
    from open3d_amd import registration as reg, synthetic as syn
    from open3d_amd.core import stream
    L = _lib.lib()
    W, H, stride = 640, 480, 2
    ds, dmax, trunc = sc.DEPTH_SCALE, sc.DEPTH_MAX, sc.TRUNC_MULT
    K = syn.intrinsics(W, H)
    g = _mk_grid(geometry, False, block_count=16384)
    vs = [0.05, 0.025, 0.0125]
    crit = [reg.ICPConvergenceCriteria(1e-6, 1e-6, i) for i in (20, 10, 5)]
    md = [0.15, 0.075, 0.0375]
    npx = (H // stride) * (W // stride)
    bufs = [torch.empty((npx, 3), dtype=torch.float32, device="cuda")
            for _ in range(3)]
    cnts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
    T_prev, d_prev, out = None, None, []
    for k in range(n_frames):
        d, c, _, T = syn.render_frames(k * 2, 1, W, H, device="cuda")
        d, c = d[0].contiguous(), c[0].contiguous()
        if k == 0:
            T_k = np.array(T[0])
        else:
            keys = g.compute_unique_block_coordinates(d_prev, K, T_prev, ds,
                                                      dmax, trunc)
            rc = g.ray_cast(keys, K, T_prev, W, H,
                            render_attributes=("depth", "normal"),
                            depth_scale=ds, depth_min=0.1, depth_max=dmax,
                            weight_threshold=1.0, trunc_voxel_multiplier=trunc)
            Tp = np.ascontiguousarray(T_prev, dtype=np.float64)
            _lib.check(L.o3dmi_unproject(
                _lib.ptr(rc["depth"]), _lib.F32, H, W, _lib.ptr(rc["normal"]),
                _lib.ptr(bufs[0]), _lib.ptr(bufs[1]), _lib.ptr(cnts[0]),
                _lib.f64p(K), _lib.f64p(Tp), C.c_float(ds), C.c_float(dmax),
                C.c_int64(stride), stream()), "unproject")
            m = int(cnts[0].item())
            Tinv = np.ascontiguousarray(np.linalg.inv(T_prev), dtype=np.float64)
            _lib.check(L.o3dmi_transform_normals(
                _lib.f64p(Tinv), _lib.ptr(bufs[1]), m, _lib.F32, stream()),
                "transform_normals")
            _lib.check(L.o3dmi_unproject(
                _lib.ptr(d), _lib.U16, H, W, None, _lib.ptr(bufs[2]), None,
                _lib.ptr(cnts[1]), _lib.f64p(K), _lib.f64p(Tp), C.c_float(ds),
                C.c_float(dmax), C.c_int64(stride), stream()), "unproject")
            src = bufs[2][:int(cnts[1].item())]
            r = reg.multi_scale_icp(src, bufs[0][:m], bufs[1][:m], vs, crit, md)
            out.append((np.array(r.transformation), r.num_iterations,
                        r.inlier_rmse, r.fitness, src.cpu().numpy().copy()))
            T_k = T_prev @ np.linalg.inv(r.transformation)
        g.integrate_frame(d, c, K, K, T_k, ds, dmax, trunc)
        T_prev, d_prev = T_k, d
    return out, _sorted_export(g)


def test_tracking_loop_is_reproducible_run_to_run():
    """The clouds the tracker sees are in pixel order (Unproject compacts by a
    look-back over chunk totals, not an atomic counter), so nothing in the loop
    depends on arrival order: two runs of the same loop give the same bits --
    clouds, poses, iteration counts and the integrated grid."""
    a, ga = _track_vga(5)
    b, gb = _track_vga(5)
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x[4], y[4]), k           # frame cloud, in order
        assert np.array_equal(x[0], y[0]), (k, x[0] - y[0])
        assert x[1:4] == y[1:4], k
    assert len(ga) == len(gb)
    for u, v in zip(ga, gb):
        assert np.array_equal(u, v)


def _frames_720p(n, step=2):
    from open3d_amd import synthetic as syn
    K = syn.intrinsics(1280, 720)
    ds, cs, Ts = [], [], []
    for k in range(0, n * step, step):
        d, c, _, T = syn.render_frames(k, 1, 1280, 720, device="cuda")
        ds.append(d[0].contiguous())
        cs.append(c[0].contiguous())
        Ts.append(T[0])
    return K, ds, cs, Ts


def _sorted_export(g):
    keys, vals = g.export_blocks()
    keys = keys.cpu().numpy()
    order = np.lexsort(keys.T[::-1])
    return [keys[order]] + [v.cpu().numpy()[order] for v in vals]


@pytest.mark.timeout(900)
def test_configs3_block_ownership_8_ranks_720p_union_is_the_single_grid():
    """BASELINE configs[3], scheme A (the bench's multi-GPU headline) at its
    size: 8 ownership classes see the same 56-frame 1280x720 stream; each
    grid holds exactly its class of the single grid's keys and every block is
    bit-identical to the single grid's."""
    _lib, geometry = _gpu()
    from open3d_amd import sharding
    world, n = 8, 56
    K, ds, cs, Ts = _frames_720p(n)

    def run(rank, w):
        g = _mk_grid(geometry, False, block_count=32768)
        if w > 1:
            g.set_block_ownership(rank, w)
        g.integrate_frames(ds, cs, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                           sc.TRUNC_MULT)
        return _sorted_export(g)

    full = run(0, 1)
    assert full[0].shape[0] > 1000
    owner = sharding.block_owner(full[0], world)
    seen = 0
    for r in range(world):
        part = run(r, world)
        sel = owner == r
        assert np.array_equal(part[0], full[0][sel]), r
        for a, b in zip(part[1:], full[1:]):
            assert a.tobytes() == b[sel].tobytes(), r
        seen += part[0].shape[0]
    assert seen == full[0].shape[0]


class _Loopback:
    """In-process transport for o3dmi_comm_create_custom: `world` host threads
    (one rank each, all on this device) meet at barriers and copy each other's
    device ranges -- the exchange pattern of a real transport, so the library's
    merge runs unchanged."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def comm(self, rank):
        from open3d_amd import _lib, sharding
        from open3d_amd.core import tensor_from_ptr
        world, slots, bar = self.world, self.slots, self.bar

        def view(ptr, n):
            return tensor_from_ptr(ptr, (int(n),), _lib.U8, None)

        def meet():
            torch.cuda.synchronize()
            bar.wait()

        def allreduce(_u, dev, n, _s):
            t = tensor_from_ptr(dev, (int(n),), _lib.F64, None)
            slots[rank] = t
            meet()
            total = torch.stack([slots[p] for p in range(world)]).sum(0)
            meet()
            t.copy_(total)
            meet()
            return 0

        def allgather(_u, send, recv, nbytes, _s):
            slots[rank] = view(send, nbytes)
            meet()
            out = view(recv, nbytes * world)
            for p in range(world):
                out[p * nbytes:(p + 1) * nbytes].copy_(slots[p])
            meet()
            return 0

        def alltoallv(_u, send, sb, so, recv, rb, ro, _s):
            slots[rank] = (send, [int(sb[p]) for p in range(world)],
                           [int(so[p]) for p in range(world)])
            meet()
            ok = 0
            for p in range(world):
                src, psb, pso = slots[p]
                if psb[rank] != rb[p]:
                    ok = 1
                elif rb[p]:
                    view(recv + ro[p], rb[p]).copy_(
                        view(src + pso[rank], psb[rank]))
            meet()
            return ok

        cbs = (_lib.TRANSPORT_ALLREDUCE(allreduce),
               _lib.TRANSPORT_ALLGATHER(allgather),
               _lib.TRANSPORT_ALLTOALLV(alltoallv))
        table = _lib.TransportC(*cbs)
        h = C.c_void_p()
        _lib.check(_lib.lib().o3dmi_comm_create_custom(
            C.byref(table), None, rank, world, C.byref(h)), "comm")
        return sharding.Comm(h, keep=(cbs, table))


def _run_ranks(world, body):
    import threading
    out, errs = [None] * world, []

    def main(r):
        try:
            torch.cuda.set_device(0)
            out[r] = body(r)
        except BaseException as e:  # surfaced by the caller
            errs.append((r, e))
    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0][1]
    return out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("frames_per_launch", [2, 12])
def test_configs3_sliced_touch_8_rank_threads_720p(frames_per_launch):
    """BASELINE configs[3] with the sliced block touch and a REAL exchange: 8
    ranks as host threads on this device, the library's communicator over the
    in-process transport, `integrate_frames` on grids with an ownership (the
    call a C++ rank makes). Rank r touches its band of the 1280x720 ray tiles,
    the candidate records are all-gathered per chunk, each rank integrates the
    blocks it owns with one launch per chunk: the 8 grids are the 8 ownership
    classes of the single grid, bit for bit."""
    _lib, geometry = _gpu()
    from open3d_amd import sharding
    world, n = 8, 56
    K, ds, cs, Ts = _frames_720p(n)
    full_g = _mk_grid(geometry, False, block_count=32768)
    full_g.integrate_frames(ds, cs, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                            sc.TRUNC_MULT, frames_per_launch=frames_per_launch)
    full = _sorted_export(full_g)
    owner = sharding.block_owner(full[0], world)
    lb = _Loopback(world)

    def body(r):
        g = _mk_grid(geometry, False, block_count=8192)
        g.set_block_ownership(r, world)
        comm = lb.comm(r)
        comm.install()
        try:
            g.integrate_frames(ds, cs, K, K, Ts, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               sc.TRUNC_MULT,
                               frames_per_launch=frames_per_launch)
            torch.cuda.synchronize()
            st = g.sliced_stats()
        finally:
            sharding.Comm.uninstall()
            comm.destroy()
        return _sorted_export(g), st

    got = _run_ranks(world, body)
    seen = 0
    for r in range(world):
        part, st = got[r]
        assert st["chunks"] == -(-n // (16 * frames_per_launch)), st
        sel = owner == r
        assert np.array_equal(part[0], full[0][sel]), r
        for a, b in zip(part[1:], full[1:]):
            assert a.tobytes() == b[sel].tobytes(), r
        seen += part[0].shape[0]
    assert seen == full[0].shape[0]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_frames", [56, 112])
def test_configs3_frame_sharded_8_ranks_720p_merge(n_frames):
    """BASELINE configs[3], scheme B at its size: 8 ranks integrate frames
    r, r + 8, ... of a 1280x720 stream into private grids; the library's
    owner-partitioned exchange (o3dmi_vbg_merge_frame_sharded) then moves
    every block to its owner and folds the partial blocks in. Result: 8
    disjoint grids = the ownership classes of the single-stream grid's keys;
    weights exact; TSDF <= 1e-4 (the same weighted sum in another
    association); colour: both sides truncate a running mean to uint16 at
    every step, the single stream once per frame, the merged one once per
    frame of a shard + once per fold -- each truncation loses < 1 unit and
    every later step scales what was lost by w / (w + 1), so per voxel
    |merged - single| < (w + 1) / 2 + 8 with w the voxel's observation count
    (that is the worst case of the REFERENCE's own arithmetic against the
    exact mean, not of the merge). The bound does not depend on the stream
    length -- the 112-frame run has the same per-voxel bound as the 56-frame
    one; the observed maximum is reported (16 / 34 units of 255 for 56 / 112
    frames on the first run: the single stream's own truncation drift grows
    with a voxel's observation count, the merged grid's much less)."""
    _lib, geometry = _gpu()
    from open3d_amd import sharding
    world = 8
    K, ds, cs, Ts = _frames_720p(n_frames)

    def integrate(sel, grid=None):
        g = grid or _mk_grid(geometry, False, block_count=32768)
        g.integrate_frames([ds[i] for i in sel], [cs[i] for i in sel], K, K,
                           [Ts[i] for i in sel], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                           sc.TRUNC_MULT)
        return g

    full = _sorted_export(integrate(range(n_frames)))
    parts = [integrate(range(r, n_frames, world)) for r in range(world)]
    torch.cuda.synchronize()
    lb = _Loopback(world)

    def rank_body(r):
        comm = lb.comm(r)
        parts[r].merge_frame_sharded(comm)
        torch.cuda.synchronize()
        lb.bar.wait()
        comm.destroy()
        return _sorted_export(parts[r])

    merged = _run_ranks(world, rank_body)
    owner = sharding.block_owner(full[0], world)
    worst_t, worst_c, worst_ratio = 0.0, 0, 0.0
    for r in range(world):
        sel = owner == r
        keys, tsdf, weight, color = merged[r]
        assert np.array_equal(keys, full[0][sel]), r
        assert np.array_equal(weight, full[2][sel]), r
        worst_t = max(worst_t, float(np.abs(tsdf - full[1][sel]).max()))
        dc = np.abs(color.astype(np.int32) - full[3][sel].astype(np.int32))
        bound = (weight.astype(np.int32) + 1) // 2 + world
        assert (dc <= bound).all(), r
        worst_c = max(worst_c, int(dc.max()))
        worst_ratio = max(worst_ratio, float((dc / bound).max()))
    print("frame-sharded merge, %d frames x 8 ranks at 720p: max |dTSDF| "
          "%.3g, max |dcolour| %d units of 255 (%.2f of the per-voxel bound)"
          % (n_frames, worst_t, worst_c, worst_ratio))
    assert worst_t <= 1e-4
    assert worst_ratio <= 1.0


@pytest.mark.timeout(900)
def test_configs3_level_sharded_multiscale_icp_8_ranks_equals_unsharded():
    """The ICP half of configs[3]: 8 ranks (host threads on this device, the
    library's communicator over the in-process transport, all-reduce of the 32
    float64 sums inside every Gauss-Newton iteration) run MultiScaleICP (5 /
    2.5 / 1.25 cm) on 720p-sized clouds with LEVEL sharding: every rank builds
    the whole voxel pyramid and works on its slice of each level, so -- unlike
    caller-side sharding of the raw cloud -- the sharded run is the unsharded
    one: same iteration count, pose equal to the rounding of the float64 sums,
    identical on every rank, and the union of the ranks' correspondence rows is
    the unsharded correspondence set."""
    _lib, geometry = _gpu()
    from open3d_amd import registration as reg
    from open3d_amd import synthetic as syn
    world = 8
    p = syn.make_icp_pair(230000, 230000, seed=4)
    src = torch.from_numpy(p["source"]).cuda()
    tgt = torch.from_numpy(p["target"]).cuda()
    nrm = torch.from_numpy(p["target_normals"]).cuda()
    vs = [0.05, 0.025, 0.0125]
    crit = [reg.ICPConvergenceCriteria(1e-6, 1e-6, n) for n in (20, 10, 5)]
    md = [0.15, 0.075, 0.0375]
    one = reg.multi_scale_icp(src, tgt, nrm, vs, crit, md)
    torch.cuda.synchronize()
    lb = _Loopback(world)

    def rank_body(r):
        comm = lb.comm(r)
        comm.install(level_sharding=True)
        try:
            out = reg.multi_scale_icp(src.clone(), tgt, nrm, vs, crit, md)
            torch.cuda.synchronize()
        finally:
            sharding_uninstall()
        lb.bar.wait()
        comm.destroy()
        return out

    from open3d_amd.sharding import Comm
    sharding_uninstall = Comm.uninstall
    got = _run_ranks(world, rank_body)
    n_rows = one.correspondence_set.shape[0]
    union = torch.full((n_rows,), -1, dtype=torch.int64, device="cuda")
    for r in range(world):
        assert got[r].num_iterations == one.num_iterations, r
        d = np.abs(got[r].transformation - one.transformation).max()
        assert d <= 1e-9, (r, d)
        assert np.array_equal(got[r].transformation, got[0].transformation)
        assert abs(got[r].fitness - one.fitness) < 1e-12
        assert abs(got[r].inlier_rmse - one.inlier_rmse) < 1e-9
        c = got[r].correspondence_set
        assert c.shape[0] == n_rows
        mine = c >= 0
        assert not bool((union[mine] >= 0).any())   # slices are disjoint
        union[mine] = c[mine]
    assert torch.equal(union, one.correspondence_set)
