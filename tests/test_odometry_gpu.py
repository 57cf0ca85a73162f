"""GPU parity tests of the RGB-D odometry front end:
HIP kernels through the C ABI vs the CPU oracle (oracle/odometry_oracle.cpp,
itself pinned bit for bit to the reference's ImageCPU / RGBDOdometryCPU bodies).

Bars: image ops bit-exact (float32 per-pixel arithmetic, including the
IPP-semantics filters as the oracle defines them); the 29 sums match the
float64-accumulating oracle to rtol 1e-12 with the inlier count exact; the
multi-scale pose within 1e-6 rad / 1e-5 m."""
import numpy as np
import pytest
import torch

import _oracle as orc
from test_odometry_oracle import NAN, _holes, _levels, _pair

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import odometry
    return odometry


def same_bits(a, b):
    """Bit-identical values; NaNs match NaNs whatever their sign / payload (an
    x86 invalid operation yields the negative default NaN, the GPU the
    positive one -- neither is a value)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind != "f":
        return a.tobytes() == b.tobytes()
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    iv = a.view(np.uint32 if a.dtype == np.float32 else np.uint64)
    jv = b.view(iv.dtype)
    return bool(np.array_equal(iv[~na], jv[~nb]))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def _pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    c = (np.trace(d[:3, :3]) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))), \
        float(np.linalg.norm(d[:3, 3]))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_depth_image_ops_bit_exact(dtype):
    odo = _gpu()
    sd, _, _, _, K, _, _ = _pair(noise=0.002)
    sd = _holes(sd)
    src = sd if dtype == np.uint16 else sd.astype(np.float32)
    for fill in (NAN, 0.0, float("inf")):
        a = orc.clip_transform(src, 1000.0, 0.0, 3.0, fill)
        g = odo.clip_transform(_dev(src), 1000.0, 0.0, 3.0, fill)
        assert same_bits(_host(g), a)
        for diff in (0.14, 0.02):
            assert same_bits(_host(odo.pyrdown_depth(g, diff, fill)),
                             orc.pyrdown_depth(a, diff, fill))
        va = orc.create_vertex_map(a, K, fill)
        vg = odo.create_vertex_map(g, K, fill)
        assert same_bits(_host(vg), va)
        assert same_bits(_host(odo.create_normal_map(vg, fill)),
                         orc.create_normal_map(va, fill))
    odd = orc.clip_transform(src[:119, :157], 1000.0, 0.0, 3.0, NAN)
    assert same_bits(_host(odo.pyrdown_depth(_dev(odd), 0.14, NAN)),
                     orc.pyrdown_depth(odd, 0.14, NAN))


def test_reference_goldens_on_device():
    """cpp/tests/t/geometry/Image.cpp:829-876 (vertex / normal maps) and the
    IPP-path filter goldens :246-281, :335-361, run on the GPU."""
    odo = _gpu()
    depth = np.array([0, 1, 2, 1, 0, 0, 2, 4, 2, 0, 0, 3, 6, 3, 29, 0, 2, 4, 2,
                      0, 0, 1, 2, 1, 0], np.uint16).reshape(5, 5)
    K = np.array([[1, 0, 2], [0, 1, 2], [0, 0, 1]], np.float64)
    c = odo.clip_transform(_dev(depth), 10.0, 0.0, 2.5, 0.0)
    v = odo.create_vertex_map(c, K, 0.0)
    n = _host(odo.create_normal_map(v, 0.0))
    assert np.allclose(n[0, 1], [0.57735, 0.57735, 0.57735], atol=1e-5)
    assert np.allclose(n[2, 2], [-0.666667, -0.333333, -0.666667], atol=1e-5)
    assert np.allclose(_host(v)[1, 2], [0.0, -0.4, 0.4], atol=1e-6)
    x = np.zeros((5, 5), np.float32)
    x[2, 2] = 1
    b = _host(odo.filter_bilateral(_dev(x), 3, 10, 10))
    assert abs(b[2, 2] - 0.201605) < 1e-5 and abs(b[1, 2] - 0.199001) < 1e-5


def test_colour_and_filter_ops_bit_exact():
    odo = _gpu()
    sd, sc, td, tc, K, _, _ = _pair(noise=0.002)
    # RGBToGray (u8, u16, f32) and To(Float32)
    rng = np.random.default_rng(0)
    for arr in (sc, rng.integers(0, 65536, (50, 60, 3), dtype=np.uint16),
                (sc.astype(np.float32) / 255).astype(np.float32)):
        assert same_bits(_host(odo.rgb_to_gray(_dev(arr))),
                         orc.rgb_to_gray(arr))
    g8 = orc.rgb_to_gray(sc)
    for arr, scale in ((g8, 1 / 255),
                       (rng.integers(0, 65536, 999, dtype=np.uint16),
                        1 / 65535),
                       (rng.standard_normal(999).astype(np.float32), 2.5)):
        assert same_bits(_host(odo.to_float(_dev(arr), scale, 0.25)),
                         orc.image_to_float(arr, scale, 0.25))
    want_i = orc.image_to_float(g8, 1 / 255)
    gi = odo.rgb_to_intensity(_dev(sc))
    assert same_bits(_host(gi), want_i)
    cf = (sc.astype(np.float32) / 255).astype(np.float32)
    assert same_bits(_host(odo.rgb_to_intensity(_dev(cf))),
                     orc.rgb_to_gray(cf))
    # filters
    for img in (want_i, orc.clip_transform(_holes(sd), 1000.0, 0.0, 3.0, NAN),
                want_i[:119, :157]):
        di = _dev(img)
        dx, dy = odo.filter_sobel(di)
        wx, wy = orc.filter_sobel(img)
        assert same_bits(_host(dx), wx) and same_bits(_host(dy), wy)
        for ks, sg in ((3, 1.0), (5, 1.0), (5, 0.8)):
            assert same_bits(_host(odo.filter_gaussian(di, ks, sg)),
                             orc.filter_gaussian(img, ks, sg))
        assert same_bits(_host(odo.resize_half_nearest(di)),
                         orc.resize_half_nearest(img))
        assert same_bits(_host(odo.pyrdown(di)), orc.pyrdown(img))
    for img in (orc.clip_transform(sd, 1000.0, 0.0, 3.0, NAN),
                orc.clip_transform(_holes(sd), 1000.0, 0.0, 3.0, NAN)):
        for args in ((5, 5.0, 10.0), (3, 10.0, 10.0)):
            got = _host(odo.filter_bilateral(_dev(img), *args))
            want = orc.filter_bilateral(img, *args)
            assert np.array_equal(np.isnan(got), np.isnan(want))
            # exp through the float64 routine on both sides: identical but for
            # the ~1e-9 of arguments where the two libms round differently
            m = ~np.isnan(want)
            assert np.allclose(got[m], want[m], rtol=3e-7, atol=0)
            assert (got[m] != want[m]).mean() < 1e-3


def test_p2plane_level_equals_separate_ops():
    odo = _gpu()
    sd, _, td, _, K, _, _ = _pair(noise=0.002)
    s = _dev(orc.clip_transform(_holes(sd, 1), 1000.0, 0.0, 3.0, NAN))
    t = _dev(orc.clip_transform(_holes(td, 2), 1000.0, 0.0, 3.0, NAN))
    sv, tv, tn = odo.p2plane_level(s, t, K)
    assert same_bits(_host(sv), _host(odo.create_vertex_map(s, K, NAN)))
    assert same_bits(_host(tv), _host(odo.create_vertex_map(t, K, NAN)))
    want = odo.create_normal_map(
        odo.create_vertex_map(odo.filter_bilateral(t, 5, 5.0, 10.0), K, NAN),
        NAN)
    assert same_bits(_host(tn), _host(want))
    # with the next level's depths riding along (odd sizes too)
    for sl in (np.s_[:, :], np.s_[:119, :157]):
        s2, t2 = s[sl].contiguous(), t[sl].contiguous()
        sv, tv, tn, sn, tdn = odo.p2plane_level(s2, t2, K, 0.14)
        assert same_bits(_host(sv), _host(odo.create_vertex_map(s2, K, NAN)))
        assert same_bits(_host(tn), _host(odo.create_normal_map(
            odo.create_vertex_map(odo.filter_bilateral(t2, 5, 5.0, 10.0), K,
                                  NAN), NAN)))
        assert same_bits(_host(sn), _host(odo.pyrdown_depth(s2, 0.14, NAN)))
        assert same_bits(_host(tdn), _host(odo.pyrdown_depth(t2, 0.14, NAN)))


@pytest.mark.parametrize("method", [orc.ODO_P2PLANE, orc.ODO_INTENSITY,
                                    orc.ODO_HYBRID])
@pytest.mark.parametrize("huber", [(0.05, 0.1), (0.004, 0.02)])
def test_odometry_sums_match_oracle(method, huber):
    odo = _gpu()
    L = _levels()
    kw = dict(depth_outlier_trunc=0.07, depth_huber_delta=huber[0],
              intensity_huber_delta=huber[1])
    want = orc.odometry_sums(method, **L, **kw, accumulate_double=True)
    maps = {k: _dev(v) for k, v in L.items() if k not in ("K", "T")}
    got = odo.compute_odometry_sums(method, L["K"], L["T"], **maps, **kw)
    assert got[28] == want[28] and got[28] > 1000
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    again = odo.compute_odometry_sums(method, L["K"], L["T"], **maps, **kw)
    assert np.array_equal(got, again)  # fixed tree: run-to-run identical
    # and within float32 accumulation error of the reference's float sums
    f32 = orc.odometry_sums(method, **L, **kw, accumulate_double=False)
    assert np.allclose(got, f32, rtol=2e-3, atol=1e-3)


def test_odometry_information_matches_oracle():
    odo = _gpu()
    from open3d_amd import _lib
    from open3d_amd.core import stream
    L = _levels()
    want = orc.odometry_information(L["source_vertex"], L["target_vertex"],
                                    L["K"], L["T"], 0.07 * 0.07,
                                    accumulate_double=True)
    sv, tv = _dev(L["source_vertex"]), _dev(L["target_vertex"])
    got = np.zeros((6, 6))
    _lib.check(_lib.lib().o3dmi_odometry_information(
        sv.shape[0], sv.shape[1], _lib.ptr(sv), _lib.ptr(tv),
        _lib.f64p(np.ascontiguousarray(L["K"])),
        _lib.f64p(np.ascontiguousarray(L["T"])), 0.07 * 0.07,
        _lib.f64p(got), stream()), "information")
    assert got[3, 3] == want[3, 3] > 1000
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("method", [orc.ODO_P2PLANE, orc.ODO_INTENSITY,
                                    orc.ODO_HYBRID])
@pytest.mark.parametrize("depth_f32", [False, True])
def test_multiscale_pose_matches_oracle(method, depth_f32):
    odo = _gpu()
    sd, sc, td, tc, K, Ts, Tt = _pair(w=320, h=240, step=1, noise=0.001)
    sd, td = _holes(sd, 1), _holes(td, 2)
    if depth_f32:
        sd, td = sd.astype(np.float32), td.astype(np.float32)
    crit = ((6, 1e-6, 1e-6), (3, 1e-6, 1e-6), (2, 1e-6, 1e-6))
    want = orc.rgbd_odometry_multiscale(method, sd, td, K, src_color=sc,
                                        tgt_color=tc, criteria=crit,
                                        accumulate_double=True)
    assert want["status"] == 0
    got = odo.rgbd_odometry_multi_scale(
        _dev(sd), _dev(td), K, None, 1000.0, 3.0,
        [odo.OdometryConvergenceCriteria(*c) for c in crit], method,
        odo.OdometryLossParams(), _dev(sc), _dev(tc))
    assert got.num_iterations == want["iterations"]
    rot, trans = _pose_err(got.transformation, want["transformation"])
    assert rot <= 1e-6 and trans <= 1e-5, (rot, trans)
    assert abs(got.fitness - want["fitness"]) <= 1e-12
    assert abs(got.inlier_rmse - want["inlier_rmse"]) <= 1e-9
    # and it is the relative pose of the synthetic trajectory, as the oracle
    gt = Tt @ np.linalg.inv(Ts)
    err0 = np.abs(np.eye(4) - gt).max()
    assert np.abs(got.transformation - gt).max() < 0.6 * err0


def test_multiscale_vga_default_criteria_and_float_colour():
    odo = _gpu()
    sd, sc, td, tc, K, Ts, Tt = _pair(w=640, h=480, step=1)
    scf = (sc.astype(np.float32) / 255).astype(np.float32)
    tcf = (tc.astype(np.float32) / 255).astype(np.float32)
    crit = ((10, 1e-6, 1e-6), (5, 1e-6, 1e-6), (3, 1e-6, 1e-6))
    want = orc.rgbd_odometry_multiscale(orc.ODO_HYBRID, sd, td, K,
                                        src_color=scf, tgt_color=tcf,
                                        criteria=crit, accumulate_double=True)
    got = odo.rgbd_odometry_multi_scale(_dev(sd), _dev(td), K,
                                        source_color=_dev(scf),
                                        target_color=_dev(tcf))
    rot, trans = _pose_err(got.transformation, want["transformation"])
    assert rot <= 1e-6 and trans <= 1e-5, (rot, trans)


def test_information_matrix_driver():
    odo = _gpu()
    sd, _, td, _, K, Ts, Tt = _pair(w=320, h=240, step=1)
    T = Tt @ np.linalg.inv(Ts)
    info = odo.compute_odometry_information_matrix(_dev(sd), _dev(td), K, T,
                                                   0.07)
    s = orc.clip_transform(sd, 1000.0, 0.0, 3.0, NAN)
    t = orc.clip_transform(td, 1000.0, 0.0, 3.0, NAN)
    want = orc.odometry_information(orc.create_vertex_map(s, K, NAN),
                                    orc.create_vertex_map(t, K, NAN), K, T,
                                    0.07 * 0.07, accumulate_double=True)
    assert np.allclose(info, want, rtol=1e-12, atol=1e-12)


def test_error_paths():
    odo = _gpu()
    z = torch.zeros((120, 160), dtype=torch.uint16, device="cuda")
    K = np.array([[100, 0, 80], [0, 100, 60], [0, 0, 1]], np.float64)
    # all-invalid depth: the reference's 6x6 system is all zeros -> its
    # solver throws "Singular 6x6 linear system"
    with pytest.raises(RuntimeError, match="Singular|inlier_count"):
        odo.rgbd_odometry_multi_scale(z, z, K, method=odo.Method.PointToPlane)
    with pytest.raises(RuntimeError, match="colour"):
        odo.rgbd_odometry_multi_scale(z, z, K, method=odo.Method.Hybrid)


def test_degenerate_frames():
    """Edge cases of the driver: a frame pair with a band of valid depth only
    (most pixels NaN), identical frames (zero motion), minimum-size pyramids."""
    odo = _gpu()
    sd, sc, td, tc, K, Ts, Tt = _pair(w=160, h=120, step=1)
    band = np.zeros_like(sd)
    band[40:80] = sd[40:80]
    crit = ((4, 1e-6, 1e-6), (2, 1e-6, 1e-6))
    for a, b in ((band, td), (sd, sd)):
        want = orc.rgbd_odometry_multiscale(orc.ODO_P2PLANE, a, b, K,
                                            criteria=crit,
                                            accumulate_double=True)
        got = odo.rgbd_odometry_multi_scale(
            _dev(a), _dev(b), K, criteria_list=[
                odo.OdometryConvergenceCriteria(*c) for c in crit],
            method=odo.Method.PointToPlane)
        assert want["status"] == 0
        rot, trans = _pose_err(got.transformation, want["transformation"])
        assert rot <= 1e-6 and trans <= 1e-5
        assert got.num_iterations == want["iterations"]
    # identical frames: the estimate stays at the identity to rounding
    assert np.abs(got.transformation - np.eye(4)).max() < 1e-4
    # an 8 x 8 image with 3 levels bottoms out at 2 x 2
    tiny = np.full((8, 8), 1500, np.uint16)
    Kt = np.array([[10.0, 0, 3.5], [0, 10.0, 3.5], [0, 0, 1]])
    try:
        odo.rgbd_odometry_multi_scale(_dev(tiny), _dev(tiny), Kt,
                                      criteria_list=(1, 1, 1),
                                      method=odo.Method.PointToPlane)
    except RuntimeError as e:   # a flat wall is rank deficient: the reference
        assert "Singular" in str(e) or "inlier" in str(e)   # throws as well
