"""Golden vectors produced by the reference's own kernel bodies
(tests/golden/make_golden.py, from /root/reference through oracle/_ref):
the oracle must reproduce them on the CPU (`-m "not gpu"`), the HIP library on
the GPU (`-m gpu`). Nothing here needs /root/reference at run time."""
import os

import numpy as np
import pytest

import _oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _sort_rows(a):
    return a[np.lexsort(a.T[::-1])]


# ------------------------------------------------------------- oracle (CPU) --
def test_oracle_reproduces_vbg_golden():
    g = _load("vbg_qqvga_res8.npz")
    voxel, res, tm, ds, dm = g["params"]
    res = int(res)
    tr = voxel * tm
    K = g["K"]
    cap = 1024
    h = orc.HashMap(cap)
    tsdf = np.zeros((cap, res, res, res), np.float32)
    wgt = np.zeros((cap, res, res, res), np.uint16)
    col = np.zeros((cap, res, res, res, 3), np.uint16)
    for i in range(g["depth"].shape[0]):
        keys = orc.depth_touch(g["depth"][i], K, g["T"][i], res, voxel, tr, ds,
                               dm, 4)
        assert np.array_equal(_sort_rows(keys), g["touch_keys_%d" % i])
        h.activate(keys)
        buf, _ = h.find(keys)
        orc.integrate(g["depth"][i], g["color"][i], buf, h.key_buffer(), tsdf,
                      wgt, col, K, K, g["T"][i], res, voxel, tr, ds, dm)
    buf, m = h.find(g["block_keys"])
    assert m.all() and h.size() == g["block_keys"].shape[0]
    assert np.array_equal(tsdf[buf], g["tsdf_u16"])
    assert np.array_equal(wgt[buf], g["weight_u16"])
    assert np.array_equal(col[buf], g["color_u16"])
    H, W = g["depth"].shape[1:3]
    rng, _ = orc.estimate_range(g["raycast_block_keys"], K, g["T"][-1], H, W,
                                8, res, voxel, 0.1, dm, frag_buffer_size=65536)
    assert np.array_equal(rng, g["range_map"])
    out = orc.raycast(h, tsdf, wgt, col, rng, K, g["T"][-1], H, W, res, voxel,
                      ds, 0.1, dm, 1.0, tm, 8,
                      attrs=("depth", "vertex", "color", "normal", "mask"))
    for k, v in out.items():
        assert np.array_equal(v, g["raycast_" + k], equal_nan=True), k


@pytest.mark.parametrize("name", ["f32", "f64"])
def test_oracle_reproduces_icp_golden(name):
    g = _load("icp_2k.npz")
    src, tgt, nrm = g["source_" + name], g["target_" + name], \
        g["normals_" + name]
    idx, d2, cnt = orc.hybrid_search(tgt, src, 0.1, 1)
    assert np.array_equal(idx[:, 0].astype(np.int64), g["corr_" + name])
    for kname, kern in (("l2", (0, 1.0, 1.0)), ("huber", (2, 0.05, 1.0)),
                        ("tukey", (5, 0.05, 1.0))):
        a = orc.p2plane_accumulate(src, tgt, nrm, g["corr_" + name], *kern,
                                   accumulate_double=False)
        assert np.array_equal(a, g["sums29_%s_%s" % (kname, name)]), kname
    A = orc.p2plane_accumulate(src, tgt, nrm, g["corr_" + name],
                               accumulate_double=False)
    st, pose, res, count = orc.decode_and_solve6x6(A)
    assert st == 0 and count == int(g["count_" + name][0])
    assert np.allclose(pose, g["pose_" + name], rtol=1e-12, atol=1e-15)
    T = orc.pose_to_transformation(g["pose_" + name])
    assert np.array_equal(T, g["T_" + name])
    assert np.array_equal(orc.transform_points(T, src),
                          g["transformed_" + name])


# ---------------------------------------------------------------- HIP (GPU) --
def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import _lib, geometry
    return torch, _lib, geometry


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["two_step", "frame", "frames"])
def test_hip_reproduces_vbg_golden(path):
    torch, _lib, geometry = _gpu()
    g = _load("vbg_qqvga_res8.npz")
    voxel, res, tm, ds, dm = [float(x) for x in g["params"]]
    res = int(res)
    K = g["K"]
    grid = geometry.VoxelBlockGrid(
        ["tsdf", "weight", "color"],
        [torch.float32, torch.uint16, torch.uint16], [1, 1, 3], voxel, res,
        2048)
    n = g["depth"].shape[0]
    dt = [torch.from_numpy(g["depth"][i]).cuda() for i in range(n)]
    ct = [torch.from_numpy(g["color"][i]).cuda() for i in range(n)]
    if path == "frames":
        grid.integrate_frames(dt, ct, K, K, [g["T"][i] for i in range(n)], ds,
                              dm, tm)
    for i in range(n):
        if path == "two_step":
            keys = grid.compute_unique_block_coordinates(dt[i], K, g["T"][i],
                                                         ds, dm, tm)
            assert np.array_equal(_sort_rows(keys.cpu().numpy()),
                                  g["touch_keys_%d" % i])
            grid.integrate(keys, dt[i], ct[i], K, K, g["T"][i], ds, dm, tm)
        elif path == "frame":
            grid.integrate_frame(dt[i], ct[i], K, K, g["T"][i], ds, dm, tm)
    hm = grid.hashmap()
    assert hm.size() == g["block_keys"].shape[0]
    buf, m = hm.find(torch.from_numpy(g["block_keys"]).cuda())
    assert bool(m.all())
    buf = buf.cpu().numpy().astype(np.int64)
    t = grid.attribute("tsdf").cpu().numpy()[buf][..., 0]
    w = grid.attribute("weight").cpu().numpy()[buf][..., 0]
    c = grid.attribute("color").cpu().numpy()[buf]
    assert np.array_equal(w, g["weight_u16"])
    assert np.array_equal(c, g["color_u16"])
    assert np.abs(t - g["tsdf_u16"]).max() <= 1e-4  # north-star bar
    assert np.array_equal(t, g["tsdf_u16"])         # and in fact bit-exact
    H, W = g["depth"].shape[1:3]
    out = grid.ray_cast(torch.from_numpy(g["raycast_block_keys"]).cuda(), K,
                        g["T"][-1], W, H,
                        render_attributes=("depth", "vertex", "color",
                                           "normal", "mask"),
                        depth_scale=ds, depth_min=0.1, depth_max=dm,
                        weight_threshold=1.0, trunc_voxel_multiplier=tm,
                        range_map_down_factor=8)
    assert np.array_equal(out["range"].cpu().numpy(), g["range_map"])
    assert np.array_equal(out["mask"].cpu().numpy(), g["raycast_mask"])
    for k in ("depth", "vertex", "color", "normal"):
        a, b = out[k].cpu().numpy(), g["raycast_" + k]
        assert np.allclose(a, b, rtol=0, atol=1e-4, equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f32", "f64"])
def test_hip_reproduces_icp_golden(name):
    import ctypes as C
    torch, _lib, geometry = _gpu()
    from open3d_amd.core import stream
    g = _load("icp_2k.npz")
    L = _lib.lib()
    dt = _lib.F64 if name == "f64" else _lib.F32
    src = torch.from_numpy(g["source_" + name]).cuda()
    tgt = torch.from_numpy(g["target_" + name]).cuda()
    nrm = torch.from_numpy(g["normals_" + name]).cuda()
    n = src.shape[0]
    h = C.c_void_p()
    _lib.check(L.o3dmi_nns_create(_lib.ptr(tgt), tgt.shape[0], dt,
                                  C.c_double(0.1), stream(), C.byref(h)), "nns")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    d2 = torch.empty(n, dtype=src.dtype, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    _lib.check(L.o3dmi_nns_hybrid_search_k1(h, _lib.ptr(src), n, _lib.ptr(idx),
                                            _lib.ptr(d2), _lib.ptr(cnt),
                                            stream()), "search")
    L.o3dmi_nns_destroy(h)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64),
                          g["corr_" + name])
    corr = torch.from_numpy(g["corr_" + name]).cuda()
    sums = torch.empty(29, dtype=torch.float64, device="cuda")
    for kname, kern in (("l2", (0, 1.0, 1.0)), ("huber", (2, 0.05, 1.0)),
                        ("tukey", (5, 0.05, 1.0))):
        _lib.check(L.o3dmi_icp_p2plane_accumulate(
            _lib.ptr(src), _lib.ptr(tgt), _lib.ptr(nrm), _lib.ptr(corr), n, dt,
            kern[0], C.c_double(kern[1]), C.c_double(kern[2]), _lib.ptr(sums),
            stream()), "accumulate")
        want = g["sums29_%s_%s" % (kname, name)]
        got = sums.cpu().numpy()
        # identical per-term arithmetic; the reference sums sequentially in the
        # point dtype, the HIP kernel in float64 with a fixed tree
        tol = 2e-4 if name == "f32" else 1e-11
        assert np.allclose(got, want, rtol=tol, atol=tol), kname
        assert got[28] == want[28]
    T = np.ascontiguousarray(g["T_" + name])
    pts = src.clone()
    _lib.check(L.o3dmi_transform_points(_lib.f64p(T), _lib.ptr(pts), n, dt,
                                        stream()), "transform")
    assert np.array_equal(pts.cpu().numpy(), g["transformed_" + name])


# ------------------------------------------------- rows f1 / f2 / f4 goldens --
ODO_MAPS = ("source_vertex", "target_vertex", "target_normal", "source_depth",
            "target_depth", "source_intensity", "target_intensity",
            "target_depth_dx", "target_depth_dy", "target_intensity_dx",
            "target_intensity_dy")


def _odo_level(g):
    L = {k: g[k] for k in ODO_MAPS if k in g.files}
    L["source_depth"], L["target_depth"] = g["source_clip"], g["target_clip"]
    return L


def _nan_equal_bits(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and \
        np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


def test_oracle_reproduces_odometry_golden():
    g = _load("odometry_qqvga.npz")
    nan = float("nan")
    s = orc.clip_transform(g["source_depth_u16"], 1000.0, 0.0, 3.0, nan)
    assert _nan_equal_bits(s, g["source_clip"])
    assert _nan_equal_bits(orc.pyrdown_depth(s, 0.14, nan), g["source_pyrdown"])
    assert _nan_equal_bits(orc.create_vertex_map(s, g["K"], nan),
                           g["source_vertex"])
    L = _odo_level(g)
    for m, name in ((0, "p2plane"), (1, "intensity"), (2, "hybrid")):
        sums = orc.odometry_sums(m, g["K"], g["T"], **L,
                                 accumulate_double=False)
        assert np.array_equal(sums.astype(np.float32), g["sums29_" + name])
        st, pose, res, cnt = orc.decode_and_solve6x6(sums)
        assert st == 0 and cnt == int(g["count_" + name][0])
        assert np.array_equal(pose, g["delta_" + name])
    info = orc.odometry_information(g["source_vertex"], g["target_vertex"],
                                    g["K"], g["T"], 0.07 * 0.07)
    assert np.array_equal(info, g["information"])


def test_oracle_reproduces_extract_and_normals_golden():
    v, g = _load("vbg_qqvga_res8.npz"), _load("extract_normals.npz")
    keys = v["block_keys"]
    n = keys.shape[0]
    h = orc.HashMap(n)
    h.activate(keys)
    active = np.arange(n, dtype=np.int32)
    nbi, nbm = orc.buffer_radius_neighbors(h, active)
    voxel, res = float(v["params"][0]), int(v["params"][1])
    got = orc.extract_point_cloud(active, nbi, nbm, keys, v["tsdf_u16"],
                                  v["weight_u16"], v["color_u16"], res, voxel,
                                  float(g["weight_threshold"][0]))
    assert got[0].tobytes() == g["points"].tobytes()
    assert got[1].tobytes() == g["normals"].tobytes()
    assert got[2].tobytes() == g["colors"].tobytes()
    r, k = float(g["radius_max_nn"][0]), int(g["radius_max_nn"][1])
    idx, _, cnt = orc.hybrid_search(g["cloud"], g["cloud"], r, k)
    assert np.array_equal(idx, g["nn_idx"]) and np.array_equal(cnt, g["nn_cnt"])
    cov = orc.estimate_covariances(g["cloud"], idx, cnt)
    assert cov.tobytes() == g["cov"].tobytes()
    assert orc.normals_from_covariances(cov).tobytes() == \
        g["cloud_normals"].tobytes()


@pytest.mark.gpu
def test_hip_reproduces_odometry_golden():
    _gpu()
    import torch
    from open3d_amd import odometry as odo
    g = _load("odometry_qqvga.npz")
    nan = float("nan")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    s = odo.clip_transform(dev(g["source_depth_u16"]), 1000.0, 0.0, 3.0, nan)
    assert _nan_equal_bits(s.cpu().numpy(), g["source_clip"])
    assert _nan_equal_bits(odo.pyrdown_depth(s, 0.14, nan).cpu().numpy(),
                           g["source_pyrdown"])
    assert _nan_equal_bits(odo.create_vertex_map(s, g["K"], nan).cpu().numpy(),
                           g["source_vertex"])
    L = {k: dev(v) for k, v in _odo_level(g).items()}
    for m, name in ((0, "p2plane"), (1, "intensity"), (2, "hybrid")):
        sums = odo.compute_odometry_sums(m, g["K"], g["T"], **L)
        want = g["sums29_" + name].astype(np.float64)
        # the reference's sums are float32 accumulations; ours float64
        assert sums[28] == want[28]
        assert np.allclose(sums, want, rtol=2e-3, atol=1e-3)
        A = np.ascontiguousarray(sums)
        from open3d_amd import _lib
        import ctypes as C
        pose = np.zeros(6)
        res, cnt = C.c_float(0), C.c_int(0)
        _lib.check(_lib.lib().o3dmi_decode_and_solve6x6(
            _lib.f64p(A), _lib.f64p(pose), C.byref(res), C.byref(cnt)), "solve")
        assert cnt.value == int(g["count_" + name][0])
        assert np.abs(pose - g["delta_" + name]).max() < 1e-4


@pytest.mark.gpu
def test_hip_reproduces_extract_and_normals_golden(tmp_path):
    _gpu()
    import torch
    from open3d_amd import geometry, registration
    v, g = _load("vbg_qqvga_res8.npz"), _load("extract_normals.npz")
    voxel, res = float(v["params"][0]), int(v["params"][1])
    # put the fixture's grid on the device through the NPZ loader: buffer
    # index i <-> key i, exactly the fixture's ordering
    path = str(tmp_path / "grid.npz")
    np.savez(path, voxel_size=np.array([voxel], np.float32),
             block_resolution=np.array([res], np.int64),
             **{"CPU:0": np.zeros((), np.uint8)},
             attr_name_tsdf=np.array([0], np.int32),
             attr_name_weight=np.array([1], np.int32),
             attr_name_color=np.array([2], np.int32), key=v["block_keys"],
             value_000=v["tsdf_u16"][..., None], value_001=v["weight_u16"][..., None],
             value_002=v["color_u16"])
    grid = geometry.VoxelBlockGrid.load(path)
    pcd = grid.extract_point_cloud(float(g["weight_threshold"][0]))
    key = lambda p, n, c: _sort_rows(np.concatenate([p, n, c], axis=1))
    assert np.array_equal(
        key(pcd["positions"].cpu().numpy(), pcd["normals"].cpu().numpy(),
            pcd["colors"].cpu().numpy()),
        key(g["points"], g["normals"], g["colors"]))
    r, k = float(g["radius_max_nn"][0]), int(g["radius_max_nn"][1])
    nrm = registration.estimate_normals(torch.from_numpy(g["cloud"]).cuda(), k,
                                        r).cpu().numpy()
    from _normals_check import assert_normals_match
    assert_normals_match(nrm, g["cloud_normals"], g["cov"],
                         g["cloud"].dtype.type, min_checked=0.3)
