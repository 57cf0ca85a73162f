"""GPU parity tests of SURVEY section 8 row f2: VoxelBlockGrid::ExtractPointCloud
and the slam::Model loop (TrackFrameToModel / Integrate / SynthesizeModelFrame)
through the C ABI, against the CPU oracle (pinned to the reference's own
ExtractPointCloudCPU / RGBDOdometryCPU / IntegrateCPU / RayCastCPU bodies).

Bars: extracted points / normals / colours bit-exact and in the oracle's
sequential order; per-frame tracking pose within 1e-6 rad / 1e-5 m of the
oracle on identical inputs; the integrated grid bit-exact; the synthesized model
frame within 1e-3 depth units / 1e-5 colour of the oracle's ray cast."""
import os

import numpy as np
import pytest
import torch

import _oracle as orc
import _ref as ref
import _scene as sc
from test_vbg_gpu import OracleGrid, _compare_grids, _mk_grid

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import _lib, geometry, slam
    return _lib, geometry, slam


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _nb_tables(hm, active):
    """BufferRadiusNeighbors through the GPU hash map's Find."""
    keys = hm.key_tensor().cpu().numpy()[active]
    n = keys.shape[0]
    nbi = np.zeros((27, n), np.int32)
    nbm = np.zeros((27, n), np.uint8)
    for nb in range(27):
        d = np.array([nb % 3 - 1, (nb % 9) // 3 - 1, nb // 9 - 1], np.int32)
        b, m = hm.find(_dev((keys + d).astype(np.int32)))
        nbi[nb] = np.where(m.cpu().numpy(), b.cpu().numpy(), 0)
        nbm[nb] = m.cpu().numpy()
    return nbi, nbm


@pytest.mark.parametrize("grid_f32", [False, True])
@pytest.mark.parametrize("with_color", [True, False])
def test_extract_point_cloud_bit_exact(grid_f32, with_color):
    _lib, geometry, _ = _gpu()
    g = _mk_grid(geometry, grid_f32, block_count=8192, with_color=with_color)
    for k in range(100, 108):
        d, c, K, Ts = sc.frames(k * 3, 1, 320, 240)
        g.integrate_frame(_dev(d[0]), _dev(c[0]) if with_color else None, K,
                          K, Ts[0])
    hm = g.hashmap()
    active = np.sort(hm.active_buf_indices().cpu().numpy())
    assert active.shape[0] > 300
    nbi, nbm = _nb_tables(hm, active)
    tsdf = g.attribute("tsdf").cpu().numpy()[..., 0]
    wgt = g.attribute("weight").cpu().numpy()[..., 0]
    col = g.attribute("color").cpu().numpy() if with_color else None
    keys = hm.key_tensor().cpu().numpy()
    for thr in (0.0, 3.0):
        want = orc.extract_point_cloud(active, nbi, nbm, keys, tsdf, wgt, col,
                                       sc.RES, sc.VOXEL, thr)
        got = g.extract_point_cloud(thr)
        assert want[3] > 5000
        assert got["positions"].shape[0] == want[3]
        assert got["positions"].cpu().numpy().tobytes() == want[0].tobytes()
        assert got["normals"].cpu().numpy().tobytes() == want[1].tobytes()
        if with_color:
            assert got["colors"].cpu().numpy().tobytes() == want[2].tobytes()
        else:
            assert "colors" not in got
    # an estimate below the surface size keeps the first points in order
    est = want[3] // 3
    part = g.extract_point_cloud(3.0, est)
    assert part["positions"].shape[0] == est
    assert torch.equal(part["positions"], got["positions"][:est])
    assert torch.equal(part["normals"], got["normals"][:est])
    # larger estimate: exactly the surface
    big = g.extract_point_cloud(3.0, want[3] + 1000)
    assert torch.equal(big["positions"], got["positions"])
    # same set as the reference's own body (its order is the atomic counter's)
    if ref.available():
        ref.set_threads(8)
        b = ref.extract_point_cloud(active, nbi, nbm, keys, tsdf, wgt, col,
                                    sc.RES, sc.VOXEL, 3.0)
        assert b[3] == want[3]
        key = lambda p, n: sc.sort_rows(np.concatenate([p, n], axis=1))
        assert np.array_equal(key(b[0], b[1]),
                              key(got["positions"].cpu().numpy(),
                                  got["normals"].cpu().numpy()))
    # points lie on the synthetic room's surfaces: within a voxel of z-range
    p = got["positions"].cpu().numpy()
    assert np.isfinite(p).all() and np.abs(p).max() < 10
    nn = np.linalg.norm(got["normals"].cpu().numpy(), axis=1)
    assert ((nn > 0.99) | (nn == 0)).all()


def test_extract_point_cloud_empty_and_res8():
    _lib, geometry, _ = _gpu()
    g = _mk_grid(geometry, False, block_count=256)
    out = g.extract_point_cloud(3.0)
    assert out["positions"].shape == (0, 3)
    g8 = _mk_grid(geometry, False, block_count=4096, res=8)
    og = OracleGrid(False, 4096, res=8)
    for k in (0, 5, 10, 15):
        d, c, K, Ts = sc.frames(k, 1, 320, 240)
        og.integrate(d[0], c[0], K, Ts[0])
        g8.integrate_frame(_dev(d[0]), _dev(c[0]), K, K, Ts[0])
    hm = g8.hashmap()
    active = np.sort(hm.active_buf_indices().cpu().numpy())
    nbi, nbm = _nb_tables(hm, active)
    want = orc.extract_point_cloud(
        active, nbi, nbm, hm.key_tensor().cpu().numpy(),
        g8.attribute("tsdf").cpu().numpy()[..., 0],
        g8.attribute("weight").cpu().numpy()[..., 0],
        g8.attribute("color").cpu().numpy(), 8, sc.VOXEL, 1.0)
    got = g8.extract_point_cloud(1.0)
    assert want[3] > 1000
    assert got["positions"].cpu().numpy().tobytes() == want[0].tobytes()
    assert got["normals"].cpu().numpy().tobytes() == want[1].tobytes()
    assert got["colors"].cpu().numpy().tobytes() == want[2].tobytes()


def _pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    c = (np.trace(d[:3, :3]) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))), \
        float(np.linalg.norm(d[:3, 3]))


@pytest.mark.parametrize("method", [0, 2])
def test_model_loop_matches_oracle_operator_by_operator(method):
    """The reference's dense-SLAM loop (examples/python/t_reconstruction_system/
    dense_slam.py): track -> update pose -> integrate -> synthesize. The oracle
    runs in lock-step on the same inputs."""
    _lib, geometry, slam = _gpu()
    from open3d_amd import odometry as odo
    W, H, n = 320, 240, 6
    frames = [sc.frames(k * 2, 1, W, H) for k in range(n)]
    K = frames[0][2]
    # render_frames returns extrinsics (world -> camera)
    T_f2w = np.linalg.inv(np.array(frames[0][3][0]))
    model = slam.Model(sc.VOXEL, sc.RES, 8192, T_f2w)
    assert np.array_equal(model.get_current_frame_pose(), T_f2w)
    assert model.frame_id == -1
    input_frame = slam.Frame(H, W, K)
    raycast_frame = slam.Frame(H, W, K)
    og = OracleGrid(False, 8192)
    crit = ((6, 1e-6, 1e-6), (3, 1e-6, 1e-6), (1, 1e-6, 1e-6))
    max_track = (0.0, 0.0)
    for i in range(n):
        d, c, _, Ts = frames[i]
        input_frame.set_data_from_image("depth", _dev(d[0]))
        input_frame.set_data_from_image("color", _dev(c[0]))
        if i > 0:
            r = model.track_frame_to_model(input_frame, raycast_frame,
                                           sc.DEPTH_SCALE, sc.DEPTH_MAX, 0.07,
                                           method)
            rd = raycast_frame.get_data("depth").cpu().numpy()[..., 0]
            rc = raycast_frame.get_data("color").cpu().numpy()
            want = orc.rgbd_odometry_multiscale(
                method, d[0], rd, K, src_color=c[0], tgt_color=rc,
                criteria=crit, depth_outlier_trunc=0.07,
                accumulate_double=True)
            assert want["status"] == 0
            e = _pose_err(r.transformation, want["transformation"])
            max_track = (max(max_track[0], e[0]), max(max_track[1], e[1]))
            assert e[0] <= 1e-6 and e[1] <= 1e-5, (i, e)
            assert r.num_iterations == want["iterations"]
            T_f2w = T_f2w @ r.transformation
        model.update_frame_pose(i, T_f2w)
        assert model.frame_id == i
        assert np.array_equal(model.get_current_frame_pose(), T_f2w)
        model.integrate(input_frame, sc.DEPTH_SCALE, sc.DEPTH_MAX,
                        sc.TRUNC_MULT)
        extr = orc.inverse_transformation(T_f2w)
        keys = og.integrate(d[0], c[0], K, extr)
        fk = model.frustum_block_coords.cpu().numpy()
        assert np.array_equal(sc.sort_rows(fk), sc.sort_rows(keys))
        model.synthesize_model_frame(raycast_frame, sc.DEPTH_SCALE, 0.1,
                                     sc.DEPTH_MAX, sc.TRUNC_MULT, True)
        wthr = min(float(i), 3.0)
        range_o, needed = orc.estimate_range(keys, K, extr, H, W, 8, sc.RES,
                                             sc.VOXEL, 0.1, sc.DEPTH_MAX,
                                             frag_buffer_size=65536)
        want_rc = orc.raycast(og.h, og.tsdf, og.weight, og.color, range_o, K,
                              extr, H, W, sc.RES, sc.VOXEL, sc.DEPTH_SCALE,
                              0.1, sc.DEPTH_MAX, wthr, sc.TRUNC_MULT, 8,
                              ("depth", "color"))
        gd = raycast_frame.get_data("depth").cpu().numpy()
        gc = raycast_frame.get_data("color").cpu().numpy()
        assert np.array_equal(gd > 0, want_rc["depth"] > 0)
        assert (gd > 0).mean() > 0.5
        assert np.abs(gd - want_rc["depth"]).max() <= 1e-3
        assert np.abs(gc - want_rc["color"]).max() <= 1e-5
    assert _compare_grids(og, model.voxel_grid)[1]  # whole grid bit-exact
    # the trajectory follows the ground truth of the synthetic stream
    # (sanity only: the estimate is the oracle's to 1e-6 / 1e-5 per frame; how
    # well 10 projective Gauss-Newton steps track the analytic room is the
    # algorithm's business)
    gt = np.linalg.inv(np.array(frames[-1][3][0]))
    e = _pose_err(gt, T_f2w)
    moved = _pose_err(np.linalg.inv(np.array(frames[0][3][0])), gt)
    print("drift rad/m", e, "moved", moved, "max track err", max_track)
    assert e[0] < 2e-2 and e[1] < 5e-2, (e, moved)
    pcd = model.extract_pointcloud(3.0)
    assert pcd["positions"].shape[0] > 1000
    assert set(pcd) == {"positions", "normals", "colors"}


def test_model_depth_only_and_dummy_colour():
    _lib, geometry, slam = _gpu()
    W, H = 320, 240
    d, c, K, Ts = sc.frames(0, 1, W, H)
    model = slam.Model(sc.VOXEL, sc.RES, 4096)
    model.update_frame_pose(0, np.linalg.inv(np.array(Ts[0])))
    f = slam.Frame(H, W, K)
    f.set_data("depth", _dev(d[0]))
    model.integrate(f)                       # no colour: depth-only update
    rf = slam.Frame(H, W, K)
    model.synthesize_model_frame(rf, enable_color=False)
    assert rf.get_data("color").shape == (H, W, 3)
    assert float(rf.get_data("color").abs().max()) == 0.0
    assert (rf.get_data("depth") > 0).float().mean() > 0.5
    r = model.track_frame_to_model(f, rf)    # point-to-plane ignores colour
    # a one-frame TSDF ray-casts about half a voxel off the measured surface
    assert np.abs(r.transformation - np.eye(4)).max() < 2e-2
    assert r.fitness > 0.5
    with pytest.raises(RuntimeError, match="previous Integrate"):
        slam.Model(sc.VOXEL).synthesize_model_frame(slam.Frame(H, W, K))


def test_cpp_host_example_runs_the_loop_through_the_c_abi():
    """examples/dense_slam.cpp: a plain C++ program (hipMalloc + the C ABI, no
    Python / torch in the process) runs the slam::Model loop on an analytic
    scene and checks its own trajectory against the closed-form poses."""
    import json
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "dense_slam")
    r = subprocess.run([exe, "30", "320", "240"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["frames"] == 30 and out["surface_points"] > 1000
    assert out["max_translation_error_m"] < 0.08
    assert out["max_rotation_error_rad"] < 0.01745


def test_cpp_icp_tracking_example_runs_configs2_through_the_c_abi():
    """examples/icp_slam.cpp: BASELINE configs[2] (ray cast -> model cloud ->
    frame cloud -> MultiScaleICP -> Integrate) from plain C++ on the C ABI; the
    program checks its trajectory against the closed-form poses."""
    import json
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "icp_slam")
    r = subprocess.run([exe, "20", "320", "240"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["frames"] == 19
    assert out["max_translation_error_m"] < 0.08
    assert out["max_rotation_error_rad"] < 0.01745
    assert out["icp_iterations_per_frame"] >= 3


def test_cpp_icp_tracking_example_sharded_ranks_through_the_library_comm():
    """examples/icp_slam.cpp with ranks = 3: one host thread per rank, the
    library's communicator installed from C++ (o3dmi_set_comm +
    o3dmi_set_icp_level_sharding), the per-iteration all-reduce inside the
    library, and the model frame rendered by the ranks together
    (o3dmi_vbg_ray_cast_sharded: a band of rows each, all-gathered -- the whole
    multi-GPU tracking frame of SURVEY 8(e)). On this one-GPU box the in-process loopback transport stands in
    for RCCL (which refuses several ranks on one device) -- and, where the box
    has the GPUs, the RCCL transport is run as well. Every rank must end with
    the same poses; against the single-rank run the ICP calls agree to the
    rounding of the float64 sums (the level pyramid is the unsharded one --
    tests/test_configs_gpu.py checks one call to 1e-9), which over a tracking
    loop with feedback (integrate at the estimated pose, ray cast, track
    again) can move an iteration count by one -- and the unprojected clouds
    come out in a different order on every run (compaction by atomic
    counter), which moves the float32 voxel means of the pyramid: the
    trajectories are compared at half a centimetre, the run-to-run spread of
    the single-rank loop itself."""
    import json
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "icp_slam")

    def run(*extra):
        r = subprocess.run([exe, "12", "320", "240", "0"] + list(extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        return json.loads(r.stdout.strip().splitlines()[-1])

    one = run()
    many = run("3", "loopback")
    assert many["ranks"] == 3 and many["poses_identical_on_all_ranks"]
    assert abs(many["max_translation_error_m"] -
               one["max_translation_error_m"]) < 5e-3
    # (twelve single-rank runs on one box: 11.0 ... 13.2 iterations per frame,
    # 0.0356 ... 0.0357 m; six 3-rank runs: 11.3 ... 13.7, 0.0339 ... 0.0356 m)
    assert abs(many["icp_iterations_per_frame"] -
               one["icp_iterations_per_frame"]) < 3.5
    if torch.cuda.device_count() >= 2:
        rccl = run("2", "rccl")
        assert rccl["poses_identical_on_all_ranks"]
        assert abs(rccl["max_translation_error_m"] -
                   one["max_translation_error_m"]) < 5e-3
