"""Pins the CPU oracle (oracle/*.cpp, a restatement) against the reference's OWN
kernel bodies compiled from /root/reference (oracle/_ref/libo3d_ref.so, see
oracle/ref_shim/README.md). Bit-exact unless stated.

Runs wherever libo3d_ref.so exists (built here from the reference sources; it
travels to the GPU box as a prebuilt file). Skipped otherwise.
"""
import numpy as np
import pytest

import _oracle as orc
import _ref as ref
import _scene as sc

pytestmark = pytest.mark.skipif(not ref.available(),
                                reason="oracle/_ref/libo3d_ref.so not built")

TR = sc.VOXEL * sc.TRUNC_MULT


@pytest.mark.parametrize("k", [0, 300, 700])
@pytest.mark.parametrize("f32", [False, True])
def test_depth_touch_same_block_set(k, f32):
    d, c, K, Ts = sc.frames(k, 1)
    depth = d[0].astype(np.float32) if f32 else d[0]
    a = orc.depth_touch(depth, K, Ts[0], sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE,
                        sc.DEPTH_MAX, 4)
    b = ref.depth_touch(depth, K, Ts[0], sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE,
                        sc.DEPTH_MAX, 4)
    assert a.shape[0] > 100
    assert np.array_equal(sc.sort_rows(a), sc.sort_rows(b))


def test_depth_touch_no_block_is_an_error_in_both():
    depth = np.zeros((480, 640), np.uint16)
    K = sc.frames(0, 1)[2]
    with pytest.raises(RuntimeError):
        ref.depth_touch(depth, K, np.eye(4), sc.RES, sc.VOXEL, TR,
                        sc.DEPTH_SCALE, sc.DEPTH_MAX, 4)
    assert orc.depth_touch(depth, K, np.eye(4), sc.RES, sc.VOXEL, TR,
                           sc.DEPTH_SCALE, sc.DEPTH_MAX, 4).shape[0] == 0


def test_pointcloud_touch_same_block_set():
    rng = np.random.default_rng(3)
    pts = (rng.random((5000, 3)) * 4 - 2).astype(np.float32)
    a = orc.pointcloud_touch(pts, sc.RES, sc.VOXEL, TR)
    b = ref.pointcloud_touch(pts, sc.RES, sc.VOXEL, TR)
    assert np.array_equal(sc.sort_rows(a), sc.sort_rows(b))


@pytest.mark.parametrize("res,voxel", [(16, 0.008), (8, 0.0125), (3, 0.1)])
def test_voxel_coordinates_and_flattened_indices(res, voxel):
    """GetVoxelCoordinatesAndFlattenedIndicesCPU compiled from the reference
    against the restatement, bit for bit: random (repeated, unordered) buffer
    indices into a key buffer with negative and large block coordinates."""
    rng = np.random.default_rng(17)
    cap = 300
    keys = rng.integers(-2000, 2000, (cap, 3)).astype(np.int32)
    keys[5] = (-(1 << 19), (1 << 19) - 1, 0)
    buf = rng.integers(0, cap, 77).astype(np.int32)
    buf[:3] = (5, 0, cap - 1)
    ca, fa = orc.voxel_coords_flat(buf, keys, res, voxel)
    cb, fb = ref.voxel_coords_flat(buf, keys, res, voxel)
    assert np.array_equal(ca, cb) and np.array_equal(fa, fb)
    assert fa.shape == (77 * res ** 3,) and ca.dtype == np.float32
    # the numpy restatements of GetVoxelIndices / GetVoxelCoordinates agree
    # with the kernel's enumeration: same voxel order, same integer coordinates
    vi = orc.voxel_indices(buf, res)
    assert np.array_equal(vi[0] * res ** 3 + vi[3] * res * res + vi[2] * res +
                          vi[1], fa)
    vc = orc.voxel_coordinates(vi, keys, res)
    assert np.array_equal((vc.T.astype(np.int32) *
                           np.float32(voxel)).astype(np.float32), ca)


def _grids(grid_f32, cap, res, with_color=True):
    wd = np.float32 if grid_f32 else np.uint16
    mk = lambda: (np.zeros((cap, res, res, res), np.float32),
                  np.zeros((cap, res, res, res), wd),
                  np.zeros((cap, res, res, res, 3), wd) if with_color else None)
    return mk(), mk()


@pytest.mark.parametrize("input_f32", [False, True])
@pytest.mark.parametrize("grid_f32", [False, True])
def test_integrate_bit_exact_all_dtype_combos(input_f32, grid_f32):
    """3 frames accumulated into the same blocks (weights 1..3): tsdf, weight
    and colour bit-identical between the restatement and IntegrateCPU<...>."""
    cap = 2048
    (t1, w1, c1), (t2, w2, c2) = _grids(grid_f32, cap, sc.RES)
    h = orc.HashMap(cap)
    for k in (40, 44, 48):
        d, c, K, Ts = sc.frames(k, 1)
        depth, color = (sc.as_f32_inputs(d[0], c[0]) if input_f32
                        else (d[0], c[0]))
        keys = orc.depth_touch(depth, K, Ts[0], sc.RES, sc.VOXEL, TR,
                               sc.DEPTH_SCALE, sc.DEPTH_MAX, 4)
        h.activate(keys)
        buf, m = h.find(keys)
        assert m.all()
        kb = h.key_buffer().copy()
        args = (K, K, Ts[0], sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE,
                sc.DEPTH_MAX)
        orc.integrate(depth, color, buf, kb, t1, w1, c1, *args)
        ref.integrate(depth, color, buf, kb, t2, w2, c2, *args)
    assert w1.max() == 3
    assert np.array_equal(w1, w2)
    assert np.array_equal(t1, t2)
    assert np.array_equal(c1, c2)
    assert np.abs(t1).max() > 0.5 and c1.max() > 10


def test_integrate_depth_only_res8_and_other_color_camera():
    cap, res = 4096, 8
    (t1, w1, _), (t2, w2, _) = _grids(False, cap, res, with_color=False)
    h = orc.HashMap(cap)
    d, c, K, Ts = sc.frames(10, 1)
    keys = orc.depth_touch(d[0], K, Ts[0], res, sc.VOXEL, TR, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, 4)
    h.activate(keys)
    buf, _ = h.find(keys)
    kb = h.key_buffer().copy()
    a = (K, K, Ts[0], res, sc.VOXEL, TR, sc.DEPTH_SCALE, sc.DEPTH_MAX)
    orc.integrate(d[0], None, buf, kb, t1, w1, None, *a)
    ref.integrate(d[0], None, buf, kb, t2, w2, None, *a)
    assert np.array_equal(t1, t2) and np.array_equal(w1, w2)

    # colour camera with its own intrinsics and resolution (320x240)
    cap = 2048
    (t1, w1, c1), (t2, w2, c2) = _grids(False, cap, sc.RES)
    h = orc.HashMap(cap)
    keys = orc.depth_touch(d[0], K, Ts[0], sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE,
                           sc.DEPTH_MAX, 4)
    h.activate(keys)
    buf, _ = h.find(keys)
    kb = h.key_buffer().copy()
    c_small = np.ascontiguousarray(c[0][::2, ::2])
    K2 = np.array(K, dtype=np.float64)
    K2[:2] *= 0.5
    a = (K, K2, Ts[0], sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE, sc.DEPTH_MAX)
    orc.integrate(d[0], c_small, buf, kb, t1, w1, c1, *a)
    ref.integrate(d[0], c_small, buf, kb, t2, w2, c2, *a)
    assert np.array_equal(t1, t2) and np.array_equal(c1, c2)
    assert c1.max() > 10


def _integrated_scene(grid_f32, frames=(100, 104, 108, 112)):
    cap = 4096
    (t, w, c), _ = _grids(grid_f32, cap, sc.RES)
    h = orc.HashMap(cap)
    for k in frames:
        d, col, K, Ts = sc.frames(k, 1)
        keys = orc.depth_touch(d[0], K, Ts[0], sc.RES, sc.VOXEL, TR,
                               sc.DEPTH_SCALE, sc.DEPTH_MAX, 4)
        h.activate(keys)
        buf, _ = h.find(keys)
        orc.integrate(d[0], col[0], buf, h.key_buffer(), t, w, c, K, K, Ts[0],
                      sc.RES, sc.VOXEL, TR, sc.DEPTH_SCALE, sc.DEPTH_MAX)
    return h, t, w, c, K, Ts[0], keys


@pytest.mark.parametrize("grid_f32", [False, True])
def test_estimate_range_and_raycast_bit_exact(grid_f32):
    h, t, w, c, K, T, keys = _integrated_scene(grid_f32)
    H, W = 480, 640
    r1, _ = orc.estimate_range(keys, K, T, H, W, 8, sc.RES, sc.VOXEL, 0.1,
                               sc.DEPTH_MAX, frag_buffer_size=65536)
    r2 = ref.estimate_range(keys, K, T, H, W, 8, sc.RES, sc.VOXEL, 0.1,
                            sc.DEPTH_MAX, frag_buffer_size=65536)
    assert np.array_equal(r1, r2)
    # First-call semantics (empty fragment buffer -> heuristic size, fragments
    # may be dropped): which ones survive depends on thread interleaving in the
    # reference, so compare on one thread.
    ref.set_threads(1)
    try:
        r3, _ = orc.estimate_range(keys, K, T, H, W, 8, sc.RES, sc.VOXEL, 0.1,
                                   sc.DEPTH_MAX)
        r4 = ref.estimate_range(keys, K, T, H, W, 8, sc.RES, sc.VOXEL, 0.1,
                                sc.DEPTH_MAX)
        assert np.array_equal(r3, r4)
    finally:
        import os
        ref.set_threads(os.cpu_count() or 1)
    assert (r1[..., 1] > r1[..., 0]).mean() > 0.5

    n = h.size()
    hk = h.key_buffer()[:n].copy()
    hb, _ = h.find(hk)
    attrs = ("depth", "vertex", "color", "normal", "index", "mask",
             "interp_ratio", "interp_ratio_dx", "interp_ratio_dy",
             "interp_ratio_dz")
    a = (K, T, H, W, sc.RES, sc.VOXEL, sc.DEPTH_SCALE, 0.1, sc.DEPTH_MAX, 1.0,
         sc.TRUNC_MULT, 8)
    o1 = orc.raycast(h, t, w, c, r1, *a, attrs=attrs)
    o2 = ref.raycast(hk, hb, t, w, c, r1, *a, attrs=attrs)
    assert (o1["depth"] > 0).mean() > 0.3
    for k in attrs:
        assert np.array_equal(o1[k], o2[k], equal_nan=True), k


def test_unproject_same_point_set():
    d, c, K, Ts = sc.frames(5, 1)
    cf = (c[0].astype(np.float32) / 255.0).astype(np.float32)
    for stride in (1, 4):
        p1, c1 = orc.unproject(d[0], cf, K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               stride)
        p2, c2 = ref.unproject(d[0], cf, K, Ts[0], sc.DEPTH_SCALE, sc.DEPTH_MAX,
                               stride)
        assert p1.shape == p2.shape and p1.shape[0] > 1000
        a = np.concatenate([p1, c1], 1)
        b = np.concatenate([p2, c2], 1)
        assert np.array_equal(sc.sort_rows(a), sc.sort_rows(b))


# ------------------------------------------------------------------ ICP side
def _pairs(n, dtype, seed):
    from open3d_amd import synthetic as syn
    p = syn.make_icp_pair(n, n, seed=seed, dtype=dtype)
    idx, d2, cnt = orc.hybrid_search(p["target"], p["source"], 0.1, 1)
    corr = idx[:, 0].astype(np.int64)
    return p, corr


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", [(0, 1.0, 1.0), (1, 1.0, 1.0),
                                    (2, 0.05, 1.0), (3, 0.05, 1.0),
                                    (4, 0.05, 1.0), (5, 0.05, 1.0),
                                    (6, 0.05, 2.0), (6, 0.05, 0.0),
                                    (6, 0.05, -2.0), (6, 0.05, 1.0),
                                    (6, 0.05, -1e9)])
def test_p2plane_29_sums_bit_exact_sequential(dtype, kernel):
    """ComputePosePointToPlaneKernelCPU run as one sequential chunk (a valid
    TBB schedule) == the oracle's scalar_t-accumulating variant, bit for bit;
    this pins the per-correspondence arithmetic incl. every robust kernel."""
    p, corr = _pairs(3000, dtype, 2)
    assert (corr >= 0).sum() > 1000
    a = orc.p2plane_accumulate(p["source"], p["target"], p["target_normals"],
                               corr, *kernel, accumulate_double=False)
    b = ref.p2plane_accumulate(p["source"], p["target"], p["target_normals"],
                               corr, *kernel)
    assert np.array_equal(a, b)
    assert a[28] == (corr >= 0).sum()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_compute_pose_decode_solve_match(dtype):
    p, corr = _pairs(4000, dtype, 5)
    A = orc.p2plane_accumulate(p["source"], p["target"], p["target_normals"],
                               corr, accumulate_double=False)
    st, pose, res, cnt = orc.decode_and_solve6x6(A)
    pose_r, res_r, cnt_r = ref.compute_pose_p2plane(
        p["source"], p["target"], p["target_normals"], corr)
    assert st == 0 and cnt == cnt_r and res == res_r
    # both are LU with partial pivoting on identical inputs; elimination order
    # inside the two LU codes may differ in the last ulp
    assert np.allclose(pose, pose_r, rtol=1e-12, atol=1e-15)
    st2, pose2, res2, cnt2 = ref.decode_and_solve6x6(A)
    assert st2 == 0 and np.allclose(pose2, pose, rtol=1e-12, atol=1e-15)
    assert np.array_equal(orc.pose_to_transformation(pose),
                          ref.pose_to_transformation(pose))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_p2point_sxy_bit_exact_sequential(dtype):
    """Get3x3SxyLinearSystem (the reduction of ComputeRtPointToPointCPU) run as
    one sequential chunk == the oracle's scalar_t-accumulating variant."""
    p, corr = _pairs(3000, dtype, 4)
    assert (corr >= 0).sum() > 1000
    corr[::17] = -1
    S, ms, mt, c = orc.p2point_sxy(p["source"], p["target"], corr)
    Sr, msr, mtr, cr = ref.p2point_sxy(p["source"], p["target"], corr)
    assert c == cr == (corr >= 0).sum()
    assert np.array_equal(S, Sr)
    assert np.array_equal(ms, msr) and np.array_equal(mt, mtr)


def _rt_numpy(S, ms, mt):
    """ComputeRtPointToPointCPU after the reduction, written with numpy's
    LAPACK SVD exactly as RegistrationCPU.cpp:640-650 reads."""
    U, D, VT = np.linalg.svd(S)
    Sg = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(VT.T) < 0:
        Sg[-1, -1] = -1
    R = U @ (Sg @ VT)
    return R, mt - R @ ms


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_p2point_rt_matches_lapack_svd(dtype):
    """The oracle's restated 3x3 SVD route gives the R, t that LAPACK's SVD
    gives (Tensor::SVD is gesvd, not in /root/reference), on the reference's
    own Sxy / means; also a reflection case (det(U) det(V) < 0) and a planar
    correspondence set (sigma_3 = 0)."""
    p, corr = _pairs(4000, dtype, 9)
    Sr, msr, mtr, _ = ref.p2point_sxy(p["source"], p["target"], corr)
    f32 = dtype == np.float32
    R, t = orc.rt_from_sxy(Sr, msr, mtr, as_f32=f32)
    Rn, tn = _rt_numpy(Sr.astype(dtype), msr.astype(dtype), mtr.astype(dtype))
    tol = 5e-6 if f32 else 1e-12
    assert np.abs(R - Rn).max() < tol and np.abs(t - tn).max() < tol
    assert abs(np.linalg.det(R) - 1) < (1e-5 if f32 else 1e-12)
    rng = np.random.default_rng(3)
    for trial in range(20):
        S = rng.standard_normal((3, 3))
        if trial % 3 == 0:
            S[:, 2] = 0          # rank 2: planar source set
        if trial % 5 == 1:
            S = -S               # flips the sign of det
        ms, mt = rng.standard_normal(3), rng.standard_normal(3)
        R, t = orc.rt_from_sxy(S, ms, mt)
        Rn, tn = _rt_numpy(S, ms, mt)
        assert np.abs(R - Rn).max() < 1e-9, (trial, R, Rn)
        assert np.abs(t - tn).max() < 1e-9
        assert abs(np.linalg.det(R) - 1) < 1e-12


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_information_matrix_bit_exact_sequential(dtype):
    """ComputeInformationMatrixCPU (one sequential chunk) == the oracle's
    scalar_t-accumulating 21 sums unpacked to GTG."""
    p, corr = _pairs(3000, dtype, 6)
    corr[::13] = -1
    a = orc.unpack21(orc.information_accumulate(p["target"], corr))
    b = ref.information_matrix(p["target"], corr)
    assert np.array_equal(a, b)
    assert a[3, 3] == a[4, 4] == a[5, 5] == (corr >= 0).sum()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", [(0, 1.0, 1.0), (2, 0.05, 1.0),
                                    (3, 0.5, 1.0), (5, 0.05, 1.0)])
def test_symmetric_29_sums_bit_exact_sequential(dtype, kernel):
    """ComputePoseSymmetricKernelCPU (GetJacobianSymmetric, weight from the
    objective residual, centred right-hand side) as one sequential chunk ==
    the oracle's scalar_t-accumulating variant, bit for bit."""
    p, corr = _pairs(3000, dtype, 8)
    corr[::11] = -1
    rng = np.random.default_rng(5)
    sn = p["target_normals"][rng.integers(0, 3000, 3000)]  # any unit normals
    sn[::2] *= -1                                          # both signs of n_s.n_t
    m = corr >= 0
    ms = p["source"][m].astype(np.float64).mean(0)
    mt = p["target"][corr[m]].astype(np.float64).mean(0)
    a = orc.symmetric_accumulate(p["source"], p["target"], sn,
                                 p["target_normals"], corr, ms, mt, *kernel)
    b = ref.symmetric_accumulate(p["source"], p["target"], sn,
                                 p["target_normals"], corr, ms, mt, *kernel)
    assert np.array_equal(a, b)
    assert a[28] == m.sum() and a[27] > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", [(0, 1.0, 1.0), (2, 0.05, 1.0),
                                    (5, 0.05, 1.0)])
def test_colored_29_sums_bit_exact_sequential(dtype, kernel):
    """ComputePoseColoredICPKernelCPU (GetJacobianColoredICP: geometric +
    photometric terms, two robust weights) as one sequential chunk == the
    oracle's scalar_t-accumulating variant, bit for bit."""
    p, corr = _pairs(3000, dtype, 9)
    corr[::7] = -1
    rng = np.random.default_rng(4)
    sc = rng.random((3000, 3)).astype(dtype)
    tc = rng.random((3000, 3)).astype(dtype)
    tg = (rng.standard_normal((3000, 3)) * 2).astype(dtype)
    args = (p["source"], sc, p["target"], p["target_normals"], tc, tg, corr,
            0.968)
    a = orc.colored_accumulate(*args, *kernel)
    b = ref.colored_accumulate(*args, *kernel)
    assert np.array_equal(a, b)
    assert a[28] == (corr >= 0).sum() and a[27] > 0


def test_singular_system_is_an_error_in_both():
    A = np.zeros(29)
    assert orc.decode_and_solve6x6(A)[0] != 0
    assert ref.decode_and_solve6x6(A)[0] != 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_transform_points_normals_bit_exact(dtype):
    rng = np.random.default_rng(11)
    pts = (rng.random((5000, 3)) * 6 - 3).astype(dtype)
    T = orc.pose_to_transformation([0.03, -0.02, 0.05, 0.1, -0.2, 0.05])
    assert np.array_equal(orc.transform_points(T, pts),
                          ref.transform_points(T, pts))
    assert np.array_equal(orc.transform_normals(T, pts),
                          ref.transform_normals(T, pts))


@pytest.mark.parametrize("f64", [False, True])
def test_robust_weights_bit_exact(f64):
    for method, shape in [(0, 1.0), (1, 1.0), (2, 1.0), (3, 1.0), (4, 1.0),
                          (5, 1.0), (6, 2.0), (6, 0.0), (6, -2.0), (6, 1.0),
                          (6, -1e9)]:
        for r in (0.98, -0.3, 1e-4, 2.5):
            for s in (1.0, 0.05):
                a = orc.robust_weight(method, s, shape, r, f64)
                b = ref.robust_weight(method, s, shape, r, f64)
                assert a == b or (np.isnan(a) and np.isnan(b)), \
                    (method, shape, r, s, a, b)


# ---------------------------------------------------------------------------
# ExtractPointCloud
# ---------------------------------------------------------------------------
def _integrated_grid(grid_f32, n_frames=3, w=160, h=120, res=8, voxel=0.02):
    """A small grid integrated with the oracle (res 8, 2 cm: a few hundred
    blocks) -> (hashmap, tsdf, weight, color)."""
    cap = 4096
    wd = np.float32 if grid_f32 else np.uint16
    tsdf = np.zeros((cap, res, res, res), np.float32)
    wgt = np.zeros((cap, res, res, res), wd)
    col = np.zeros((cap, res, res, res, 3), wd)
    hm = orc.HashMap(cap)
    from open3d_amd import synthetic
    d, c, K, Ts = synthetic.render_frames(0, n_frames, w, h, device="cpu")
    for i in range(n_frames):
        dn, cn = d[i].numpy(), c[i].numpy()
        keys = orc.depth_touch(dn, K, Ts[i], res, voxel, voxel * 4, 1000.0,
                               3.0)
        hm.activate(keys)
        buf, _ = hm.find(keys)
        orc.integrate(dn, cn, buf, hm.key_buffer(), tsdf, wgt, col, K, K,
                      Ts[i], res, voxel, voxel * 4, 1000.0, 3.0)
    return hm, tsdf, wgt, col, res, voxel


@pytest.mark.parametrize("grid_f32", [False, True])
@pytest.mark.parametrize("with_color", [True, False])
def test_extract_point_cloud_vs_reference_body(grid_f32, with_color):
    hm, tsdf, wgt, col, res, voxel = _integrated_grid(grid_f32)
    if not with_color:
        col = None
    active = hm.active_indices()
    nbi, nbm = orc.buffer_radius_neighbors(hm, active)
    # interior blocks have all 27 neighbours' lookups answered
    assert nbm[13].all() and np.array_equal(nbi[13], active)
    ref.set_threads(1)  # sequential: the atomic counter follows workload order
    for thr in (0.0, 1.0, 2.0):
        a = orc.extract_point_cloud(active, nbi, nbm, hm.key_buffer(), tsdf,
                                    wgt, col, res, voxel, thr)
        b = ref.extract_point_cloud(active, nbi, nbm, hm.key_buffer(), tsdf,
                                    wgt, col, res, voxel, thr)
        assert a[3] == b[3]
        if thr == 0.0:
            assert a[3] > 2000
        assert a[0].tobytes() == b[0].tobytes()      # points
        assert a[1].tobytes() == b[1].tobytes()      # normals
        if with_color:
            assert a[2].tobytes() == b[2].tobytes()  # colours
    # estimated size smaller than the surface: the first `est` in order
    est = a[3] // 2
    a2 = orc.extract_point_cloud(active, nbi, nbm, hm.key_buffer(), tsdf, wgt,
                                 col, res, voxel, 2.0, estimated_number=est)
    assert a2[3] >= est and a2[0].shape[0] == est
    assert np.array_equal(a2[0], a[0][:est])
    ref.set_threads(8)
    b8 = ref.extract_point_cloud(active, nbi, nbm, hm.key_buffer(), tsdf, wgt,
                                 col, res, voxel, 2.0)
    assert b8[3] == a[3]
    key = lambda p: sc.sort_rows(np.concatenate(p[:2], axis=1))
    assert np.array_equal(key(b8), key(a))


# ---------------------------------------------------------------------------
# Normal estimation
# ---------------------------------------------------------------------------
def _surface_cloud(n, seed, dtype):
    from open3d_amd import synthetic
    p = synthetic.make_icp_pair(n, n, seed=seed, dtype=dtype)
    return p["target"], p["target_normals"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_normals_vs_reference_bodies(dtype):
    pts, nrm_true = _surface_cloud(4000, 5, dtype)
    # degenerate inputs ride along: isolated points (< 3 neighbours), exact
    # duplicates, points on an axis-aligned plane / line
    extra = np.array([[50, 50, 50], [60, 60, 60], [60, 60, 60.001]], dtype)
    line = np.stack([np.linspace(70, 70.2, 40), np.full(40, 1.0),
                     np.full(40, 2.0)], 1).astype(dtype)
    gx, gy = np.meshgrid(np.linspace(80, 80.3, 12), np.linspace(0, 0.3, 12))
    plane = np.stack([gx.ravel(), gy.ravel(), np.full(144, 3.0)], 1).astype(dtype)
    pts = np.ascontiguousarray(np.concatenate([pts, extra, line, plane]))
    idx, _, cnt = orc.hybrid_search(pts, pts, 0.08, 30)
    assert cnt.max() == 30 and cnt.min() <= 2
    a = orc.estimate_covariances(pts, idx, cnt)
    b = ref.estimate_covariances(pts, idx, cnt)
    assert a.tobytes() == b.tobytes()
    na = orc.normals_from_covariances(a)
    nb = ref.normals_from_covariances(b)
    assert na.tobytes() == nb.tobytes()
    # with existing normals: orientation is kept consistent with them
    prior = np.concatenate([nrm_true, np.tile([[0, 0, 1.0]], (187, 1))]) \
        .astype(dtype)
    pa = orc.normals_from_covariances(a, prior)
    pb = ref.normals_from_covariances(b, prior)
    assert pa.tobytes() == pb.tobytes()
    assert ((pa * prior).sum(1) >= 0).all()
    # and they are the surface normals up to sign
    cosang = np.abs((na[:4000] * nrm_true).sum(1))
    assert np.median(cosang) > 0.98


def _color_field(P):
    P = P.astype(np.float64)
    return np.stack([0.5 + 0.4 * np.sin(3 * P[:, 0] + 2 * P[:, 1]),
                     0.5 + 0.4 * np.cos(2 * P[:, 1] - P[:, 2]),
                     0.5 + 0.3 * np.sin(P[:, 2] * 4 + P[:, 0])], 1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_svd3x3_restatement_bit_exact(dtype):
    """core::linalg::kernel::svd3x3 / solve_svd3x3 (SVD3x3.h) vs the
    restatement in oracle/approx_svd3_oracle.h on random, symmetric, rank-deficient
    and diagonal matrices (the last make the masked guards fire). Float64: the
    reference's 32-bit masks on a double's low word make it return NaN on some
    inputs; NaN positions must agree, NaN payloads are not compared."""
    rng = np.random.default_rng(0)
    nan_cases = 0
    for t in range(1500):
        A = rng.standard_normal((3, 3)).astype(dtype)
        if t % 5 == 1:
            A = (A @ A.T).astype(dtype)
        if t % 7 == 2:
            A[:, 2] = A[:, 0]
        if t % 11 == 3:
            A = np.diag(rng.standard_normal(3)).astype(dtype)
        b = rng.standard_normal(3).astype(dtype)
        got, want = orc.svd3x3(A), ref.svd3x3(A)
        for g, w in zip(got, want):
            assert np.array_equal(g, w, equal_nan=True), (t, A)
        nan_cases += bool(np.isnan(want[1]).any())
        assert np.array_equal(orc.solve_svd3x3(A, b), ref.solve_svd3x3(A, b),
                              equal_nan=True)
    if dtype == np.float32:
        assert nan_cases == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_color_gradients_vs_reference_body(dtype):
    """EstimatePointWiseColorGradientKernel (PointCloudImpl.h:1067-1165) with
    the reference's own solve_svd3x3 == the oracle, bit for bit (Float64: NaN
    positions equal). The exact-solve variant (what
    the product uses) equals numpy's pseudo-inverse of
    the same normal equations and shows how approximate the reference's
    solver is on ill-conditioned neighbourhoods."""
    from open3d_amd import synthetic as syn
    p = syn.make_icp_pair(4000, 4000, seed=3, dtype=dtype)
    pts, nrm = p["target"], p["target_normals"]
    col = _color_field(pts).astype(dtype)
    idx, _, cnt = orc.hybrid_search(pts, pts, 0.15, 30)
    a = orc.estimate_color_gradients(pts, nrm, col, idx, cnt)
    b = ref.estimate_color_gradients(pts, nrm, col, idx, cnt)
    assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(a[cnt < 4], np.zeros(((cnt < 4).sum(), 3)))
    if dtype != np.float32:
        return
    ex = orc.estimate_color_gradients(pts, nrm, col, idx, cnt,
                                      exact_solve=True)
    err = np.abs(ex - b).max(1)
    assert np.median(err) < 1e-5 and np.quantile(err, 0.99) < 2e-2
    P, N, Cc = (x.astype(np.float64) for x in (pts, nrm, col))
    for w in np.argsort(-err)[:50]:
        k = cnt[w]
        ids = idx[w, 1:k]
        d = P[ids] @ N[w] - P[w] @ N[w]
        A = P[ids] - d[:, None] * N[w] - P[w]
        bb = Cc[ids].mean(1) - Cc[w].mean()
        AtA = A.T @ A + np.outer((k - 1) * N[w], (k - 1) * N[w])
        want = np.linalg.pinv(AtA, rcond=1e-15) @ (A.T @ bb)
        assert np.abs(ex[w] - want).max() < 2e-3 * max(1.0, np.abs(want).max())


def test_p2plane_sums_under_random_reduce_schedules():
    """tbb::parallel_reduce stand-in with seeded random split points: the
    reference's float32 29-sum (RegistrationCPU.cpp:30-90) depends on the
    schedule in its last bits, schedule 0 (one sequential chunk) equals the
    oracle's float32 restatement bit for bit, and every schedule stays within
    float32 rounding of the float64-accumulated sums."""
    from open3d_amd import synthetic as syn
    p = syn.make_icp_pair(30000, 30000, seed=9, dtype=np.float32)
    idx, _, cnt = orc.hybrid_search(p["target"], p["source"], 0.07, 1)
    corr = np.where(cnt > 0, idx[:, 0], -1).astype(np.int64)
    want64 = orc.p2plane_accumulate(p["source"], p["target"],
                                    p["target_normals"], corr,
                                    accumulate_double=True)
    want32 = orc.p2plane_accumulate(p["source"], p["target"],
                                    p["target_normals"], corr,
                                    accumulate_double=False)
    seen = set()
    try:
        for seed in range(0, 9):
            ref.set_reduce_schedule(seed, 128)
            got = ref.p2plane_accumulate(p["source"], p["target"],
                                         p["target_normals"], corr)
            if seed == 0:
                assert np.array_equal(got, want32)
            assert got[28] == want64[28]  # the count is exact
            assert np.allclose(got[:28], want64[:28], rtol=2e-4,
                               atol=1e-4 * np.abs(want64[:28]).max())
            seen.add(got.tobytes())
    finally:
        ref.set_reduce_schedule(0)
    assert len(seen) >= 5
