"""GPU parity tests of SURVEY section 8 row f4 (and the general-k form of rows
a3 / a4): HybridSearch with max_knn > 1, EstimateCovariancesUsingHybridSearch,
EstimateNormalsFromCovariances and PointCloud::EstimateNormals through the C
ABI, against the CPU oracle (pinned bit for bit to the reference's bodies).

Bars: neighbour indices / counts exact and squared distances bit-exact;
covariances bit-exact (float64 cumulants in neighbour order); normals: the
product's eigenvector routine is its own (float64 Jacobi), the oracle runs the
reference's closed-form routine -- the bar is the line angle 1e-4 rad (float32)
/ 1e-10 (float64) away from degenerate covariances plus a pinned sign, see
tests/_normals_check.py (the reference's own CPU-vs-GPU bar for normals is
1e-2, cpp/tests/t/geometry/VoxelBlockGrid.cpp:548-551)."""
import ctypes as C

import numpy as np
import pytest
import torch

import _oracle as orc
from _normals_check import assert_normals_match

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from open3d_amd import _lib, registration
    return _lib, registration


def _cloud(n, seed, dtype):
    from open3d_amd import synthetic
    p = synthetic.make_icp_pair(n, n, seed=seed, dtype=dtype)
    pts = p["target"]
    extra = np.array([[50, 50, 50], [60, 60, 60], [60, 60, 60.001],
                      [60, 60, 60]], dtype)       # isolated / duplicate points
    gx, gy = np.meshgrid(np.linspace(80, 80.3, 12), np.linspace(0, 0.3, 12))
    plane = np.stack([gx.ravel(), gy.ravel(), np.full(144, 3.0)], 1)
    line = np.stack([np.linspace(70, 70.2, 40), np.full(40, 1.0),
                     np.full(40, 2.0)], 1)
    return np.ascontiguousarray(
        np.concatenate([pts, extra, plane.astype(dtype), line.astype(dtype)])), \
        p["target_normals"]


def _search(_lib, pts, qrs, radius, k):
    from open3d_amd.core import TORCH_TO_O3DMI, stream
    L = _lib.lib()
    tp, tq = torch.from_numpy(pts).cuda(), torch.from_numpy(qrs).cuda()
    h = C.c_void_p()
    _lib.check(L.o3dmi_nns_create(_lib.ptr(tp), tp.shape[0],
                                  TORCH_TO_O3DMI[tp.dtype], C.c_double(radius),
                                  stream(), C.byref(h)), "nns_create")
    q = tq.shape[0]
    idx = torch.full((q, k), 7, dtype=torch.int32, device="cuda")
    d2 = torch.full((q, k), 7, dtype=tp.dtype, device="cuda")
    cnt = torch.zeros(q, dtype=torch.int32, device="cuda")
    _lib.check(L.o3dmi_nns_hybrid_search(h, _lib.ptr(tq), q, k, _lib.ptr(idx),
                                         _lib.ptr(d2), _lib.ptr(cnt),
                                         stream()), "hybrid_search")
    torch.cuda.synchronize()
    L.o3dmi_nns_destroy(h)
    return idx, d2, cnt


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [1, 5, 30, 64])
def test_hybrid_search_general_k(dtype, k):
    _lib, _ = _gpu()
    pts, _ = _cloud(6000, 3, dtype)
    rng = np.random.default_rng(1)
    qrs = np.ascontiguousarray(np.concatenate(
        [pts[::3], (pts[:500] + rng.normal(0, 0.01, (500, 3))).astype(dtype)]))
    want = orc.hybrid_search(pts, qrs, 0.07, k)
    idx, d2, cnt = _search(_lib, pts, qrs, 0.07, k)
    assert np.array_equal(cnt.cpu().numpy(), want[2])
    assert np.array_equal(idx.cpu().numpy(), want[0])
    assert d2.cpu().numpy().tobytes() == want[1].tobytes()
    assert want[2].max() == min(k, want[2].max()) and want[2].min() <= 2
    # the reference's own golden (cpp/tests/core/NearestNeighborSearch.cpp:321-377)
    ref_pts = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.1], [0.0, 0.0, 0.2],
                        [0.0, 0.1, 0.0], [0.0, 0.1, 0.1], [0.0, 0.1, 0.2],
                        [0.0, 0.2, 0.0], [0.0, 0.2, 0.1], [0.0, 0.2, 0.2],
                        [0.1, 0.0, 0.0]], dtype)
    q1 = np.array([[0.064705, 0.043921, 0.087843]], dtype)
    i1, dd1, c1 = _search(_lib, ref_pts, q1, 0.1, 3)
    assert i1.cpu().numpy().tolist() == [[1, 4, -1]]
    assert c1.cpu().numpy().tolist() == [2]
    assert np.allclose(dd1.cpu().numpy(), [[0.00626358, 0.00747938, 0]],
                       atol=1e-6)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_covariances_and_normals_match_oracle(dtype):
    _lib, _ = _gpu()
    from open3d_amd.core import TORCH_TO_O3DMI, stream
    L = _lib.lib()
    pts, _ = _cloud(5000, 5, dtype)
    idx, _, cnt = _search(_lib, pts, pts, 0.08, 30)
    widx, _, wcnt = orc.hybrid_search(pts, pts, 0.08, 30)
    assert np.array_equal(idx.cpu().numpy(), widx)
    tp = torch.from_numpy(pts).cuda()
    n = pts.shape[0]
    cov = torch.zeros((n, 3, 3), dtype=tp.dtype, device="cuda")
    _lib.check(L.o3dmi_pointcloud_estimate_covariances(
        _lib.ptr(tp), _lib.ptr(idx), _lib.ptr(cnt), n, 30,
        TORCH_TO_O3DMI[tp.dtype], _lib.ptr(cov), stream()), "covariances")
    want_cov = orc.estimate_covariances(pts, widx, wcnt)
    assert cov.cpu().numpy().tobytes() == want_cov.tobytes()
    nrm = torch.zeros((n, 3), dtype=tp.dtype, device="cuda")
    _lib.check(L.o3dmi_pointcloud_normals_from_covariances(
        _lib.ptr(cov), n, TORCH_TO_O3DMI[tp.dtype], _lib.ptr(nrm), 0,
        stream()), "normals")
    want = orc.normals_from_covariances(want_cov)
    got = nrm.cpu().numpy()
    # a sparse cloud: most neighbourhoods are small or line-like
    assert_normals_match(got, want, want_cov, dtype, min_checked=0.25)
    # degenerate neighbourhoods: < 3 neighbours -> identity covariance -> the
    # z axis; points on a plane z = const -> +-z
    few = np.where(wcnt < 3)[0]
    assert few.size >= 2 and np.array_equal(got[few], want[few])
    assert np.all(np.abs(got[5004:5004 + 144, 2]) > 0.999)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_normal_eigenvector_vs_reference_body(dtype):
    """The product's own smallest-eigenvector routine against the reference's
    COMPILED EstimateNormalsFromCovariancesCPU (oracle/_ref: PointCloudCPU.cpp
    + PointCloudImpl.h:746-1063 from /root/reference) on covariances of every
    conditioning: random SPD matrices with eigenvalue ratios from 1 to 1e6,
    exact planes / lines / diagonal matrices, the identity, all zeros."""
    import _ref as ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    _lib, _ = _gpu()
    from open3d_amd.core import TORCH_TO_O3DMI, stream
    L = _lib.lib()
    rng = np.random.default_rng(23)
    n = 20000
    q, _ = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    lam = np.sort(10.0 ** rng.uniform(-6, 0, (n, 3)), axis=1)
    lam[: n // 4, 0] = 0.0                       # exact planes
    lam[n // 4: n // 4 + 500, :2] = 0.0          # exact lines (degenerate)
    cov = np.einsum("nij,nj,nkj->nik", q, lam, q)
    cov = (cov + cov.transpose(0, 2, 1)) / 2 * 10.0 ** rng.uniform(
        -4, 2, (n, 1, 1))
    special = np.zeros((6, 3, 3))
    special[0] = np.eye(3)
    special[1] = np.diag([3.0, 1.0, 2.0])
    special[2] = np.diag([0.5, 2.0, 1.0])
    special[3] = np.diag([2.0, 2.0, 1.0])
    special[4] = np.diag([1e-3, 1.0, 1.0])
    cov = np.ascontiguousarray(np.concatenate([cov, special]).astype(dtype))
    want = ref.normals_from_covariances(cov)
    tc = torch.from_numpy(cov).cuda()
    m = cov.shape[0]
    nrm = torch.zeros((m, 3), dtype=tc.dtype, device="cuda")
    _lib.check(L.o3dmi_pointcloud_normals_from_covariances(
        _lib.ptr(tc), m, TORCH_TO_O3DMI[tc.dtype], _lib.ptr(nrm), 0,
        stream()), "normals")
    got = nrm.cpu().numpy()
    # the all-zero covariance: +z without prior normals (reference: same)
    assert got[-1].tolist() == [0, 0, 1] and want[-1].tolist() == [0, 0, 1]
    worst, share = assert_normals_match(got[:-1], want[:-1], cov[:-1], dtype,
                                        min_checked=0.3)
    assert got[n + 1].tolist() == [0, 1, 0] and got[n + 2].tolist() == [1, 0, 0]
    assert got[n + 3].tolist() == [0, 0, 1] and got[n + 4].tolist() == [1, 0, 0]
    # with prior normals the sign follows them (and zero stays zero)
    prior = rng.normal(size=(m, 3)).astype(dtype)
    nrm2 = torch.from_numpy(prior.copy()).cuda()
    _lib.check(L.o3dmi_pointcloud_normals_from_covariances(
        _lib.ptr(tc), m, TORCH_TO_O3DMI[tc.dtype], _lib.ptr(nrm2), 1,
        stream()), "normals")
    got2 = nrm2.cpu().numpy()
    want2 = ref.normals_from_covariances(cov, prior)
    assert not got2[-1].any() and not want2[-1].any()
    assert_normals_match(got2, want2, cov, dtype, prior=prior,
                         min_checked=0.3)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_estimate_normals_operator(dtype):
    _lib, reg = _gpu()
    pts, nrm_true = _cloud(20000, 7, dtype)
    tp = torch.from_numpy(pts).cuda()
    got = reg.estimate_normals(tp, 30, 0.08).cpu().numpy()
    want = orc.estimate_normals(pts, 0.08, 30)
    widx, _, wcnt = orc.hybrid_search(pts, pts, 0.08, 30)
    want_cov = orc.estimate_covariances(pts, widx, wcnt)
    assert_normals_match(got, want, want_cov, dtype)
    cosang = np.abs((got[:20000] * nrm_true).sum(1))
    assert np.median(cosang) > 0.99
    # existing normals keep their orientation
    prior = np.concatenate([nrm_true, np.tile([[0, 0, 1.0]],
                                              (pts.shape[0] - 20000, 1))])
    prior = np.ascontiguousarray(prior.astype(dtype))
    got2 = reg.estimate_normals(tp, 30, 0.08,
                                torch.from_numpy(prior).cuda()).cpu().numpy()
    want2 = orc.estimate_normals(pts, 0.08, 30, prior)
    assert_normals_match(got2, want2, want_cov, dtype, prior=prior)
    with pytest.raises(ValueError, match="Both max_nn and radius are none"):
        reg.estimate_normals(tp, None, None)


def test_empty_and_tiny_inputs():
    _lib, reg = _gpu()
    from open3d_amd.core import stream
    L = _lib.lib()
    empty = torch.empty((0, 3), dtype=torch.float32, device="cuda")
    assert reg.estimate_normals(empty, 30, 0.1).shape == (0, 3)
    # one and two points: every neighbourhood has < 3 members -> +z
    for n in (1, 2):
        p = torch.rand((n, 3), dtype=torch.float32, device="cuda")
        nr = reg.estimate_normals(p, 30, 0.5).cpu().numpy()
        assert np.array_equal(nr, np.tile([[0, 0, 1.0]], (n, 1)).astype(np.float32))
    # q = 0 queries, max_knn out of range
    pts = torch.rand((100, 3), dtype=torch.float32, device="cuda")
    h = C.c_void_p()
    _lib.check(L.o3dmi_nns_create(_lib.ptr(pts), 100, _lib.F32, C.c_double(0.2),
                                  stream(), C.byref(h)), "nns_create")
    assert L.o3dmi_nns_hybrid_search(h, None, 0, 5, None, None, None,
                                     stream()) == 0
    idx = torch.zeros((100, 65), dtype=torch.int32, device="cuda")
    assert L.o3dmi_nns_hybrid_search(h, _lib.ptr(pts), 100, 65, _lib.ptr(idx),
                                     None, None, stream()) != 0
    assert b"max_knn" in L.o3dmi_last_error()
    L.o3dmi_nns_destroy(h)


# ------------------------------------------------------------ KNN (no radius)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [1, 8, 30, 64])
@pytest.mark.parametrize("second_pass", ["sweep", "pyramid"])
def test_knn_search_parity(dtype, k, second_pass, monkeypatch):
    """KnnIndex + KnnSearch: indices exact, squared distances bit-exact vs an
    exhaustive scan, on a cloud with isolated / duplicate points, a regular
    lattice (distance ties) and queries far outside the cloud. Queries the
    finest grid level cannot finish go through the second pass: a sweep over
    all records (few leftovers) or the multi-level walk (many)."""
    _lib, reg = _gpu()
    monkeypatch.setenv("O3DMI_KNN_SWEEP_LIMIT",
                       "1e30" if second_pass == "sweep" else "0")
    pts, _ = _cloud(6000, 3, dtype)
    rng = np.random.default_rng(2)
    qrs = np.ascontiguousarray(np.concatenate(
        [pts[::3], (pts[:500] + rng.normal(0, 0.01, (500, 3))).astype(dtype),
         np.array([[-500, 20, 3], [1e4, 1e4, 1e4], [65, 65, 65],
                   [80.15, 0.15, 3.2]], dtype)]))
    widx, wd2 = orc.knn_search(pts, qrs, k)
    idx, d2 = reg.knn_search(torch.from_numpy(pts).cuda(),
                             torch.from_numpy(qrs).cuda(), k)
    assert idx.shape == (qrs.shape[0], k)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert d2.cpu().numpy().tobytes() == wd2.tobytes()


def test_knn_search_golden_and_edges():
    """The reference's KnnSearch golden (cpp/tests/core/
    NearestNeighborSearch.cpp:36-111) through the GPU; k > N; k <= 0; N = 1;
    all points identical."""
    _lib, reg = _gpu()
    from test_oracle_goldens import KNN_D2, KNN_IDX, KNN_PTS, KNN_Q
    tp, tq = torch.from_numpy(KNN_PTS).cuda(), torch.from_numpy(KNN_Q).cuda()
    idx, d2 = reg.knn_search(tp, tq, 3)
    assert idx.cpu().numpy().tolist() == [KNN_IDX[:3]]
    assert np.allclose(d2.cpu().numpy(), [KNN_D2[:3]], rtol=1e-5, atol=1e-8)
    idx, d2 = reg.knn_search(tp, tq, 14)
    assert idx.shape == (1, 12) and idx.cpu().numpy().tolist() == [KNN_IDX]
    assert np.allclose(d2.cpu().numpy(), [KNN_D2], rtol=1e-5, atol=1e-8)
    for bad in (0, -1):
        with pytest.raises(RuntimeError, match="knn should be larger than 0"):
            reg.knn_search(tp, tq, bad)
    with pytest.raises(RuntimeError, match="knn > 64"):
        reg.knn_search(torch.rand((100, 3), device="cuda"), tq, 65)
    one = torch.tensor([[1.0, 2.0, 3.0]], device="cuda")
    idx, d2 = reg.knn_search(one, tq, 5)
    assert idx.cpu().numpy().tolist() == [[0]]
    same = one.repeat(10, 1)
    idx, d2 = reg.knn_search(same, one, 4)
    assert idx.cpu().numpy().tolist() == [[0, 1, 2, 3]]
    assert (d2.cpu().numpy() == 0).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_estimate_normals_knn_variant(dtype):
    """EstimateNormals(max_nn = 30, radius = nullopt) -- the reference's default
    (EstimateCovariancesUsingKNNSearch, PointCloudImpl.h:692-744)."""
    _lib, reg = _gpu()
    pts, nrm_true = _cloud(12000, 9, dtype)
    tp = torch.from_numpy(pts).cuda()
    got = reg.estimate_normals(tp, 30).cpu().numpy()
    widx, _ = orc.knn_search(pts, pts, 30)
    wcnt = np.full(pts.shape[0], 30, np.int32)
    want_cov = orc.estimate_covariances(pts, widx, wcnt)
    want = orc.normals_from_covariances(want_cov)
    assert_normals_match(got, want, want_cov, dtype)
    cosang = np.abs((got[:12000] * nrm_true).sum(1))
    assert np.median(cosang) > 0.99
    # fewer than 3 points in the whole cloud is the reference's error
    with pytest.raises(RuntimeError, match="Not enough neighbors"):
        reg.estimate_normals(tp[:2].contiguous(), 30)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_estimate_normals_radius_variant(dtype):
    """EstimateNormals(max_nn = nullopt, radius) (EstimateCovariancesUsing
    RadiusSearch, PointCloudImpl.h:641-689): every neighbour within the radius.
    Checked against the oracle's hybrid search with a cap above the largest
    neighbourhood (= the sorted radius search). The list-free seam function
    sums the float64 moments in a wave (equal to rounding); the operator uses
    sorted CSR lists + the reference's per-point body."""
    _lib, reg = _gpu()
    from open3d_amd.core import TORCH_TO_O3DMI, stream
    L = _lib.lib()
    pts, nrm_true = _cloud(12000, 11, dtype)
    radius = 0.3
    widx, _, wcnt = orc.hybrid_search(pts, pts, radius, 1500)
    assert 64 < wcnt.max() < 1500                 # beyond the top-k kernels
    want_cov = orc.estimate_covariances(pts, widx, wcnt)
    tp = torch.from_numpy(pts).cuda()
    h = C.c_void_p()
    _lib.check(L.o3dmi_nns_create(_lib.ptr(tp), tp.shape[0],
                                  TORCH_TO_O3DMI[tp.dtype], C.c_double(radius),
                                  stream(), C.byref(h)), "nns_create")
    cov = torch.zeros((tp.shape[0], 3, 3), dtype=tp.dtype, device="cuda")
    _lib.check(L.o3dmi_nns_radius_covariances(h, _lib.ptr(tp), tp.shape[0],
                                              _lib.ptr(cov), stream()),
               "radius_covariances")
    torch.cuda.synchronize()
    L.o3dmi_nns_destroy(h)
    got_cov = cov.cpu().numpy()
    few = wcnt < 3
    assert few.any() and np.array_equal(got_cov[few], want_cov[few])
    scale = np.abs(want_cov).max()
    tol = 1e-6 if dtype == np.float32 else 1e-13
    assert np.abs(got_cov - want_cov).max() <= tol * scale
    # the operator goes through sorted CSR lists and the reference's per-point
    # covariance body: same bar as the hybrid / KNN variants
    got = reg.estimate_normals(tp, None, radius).cpu().numpy()
    want = orc.normals_from_covariances(want_cov)
    assert_normals_match(got, want, want_cov, dtype)
    cos_true = np.abs((got[:12000] * nrm_true).sum(1))
    assert np.median(cos_true) > 0.99


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fixed_radius_search_parity(dtype):
    """FixedRadiusSearch (CSR lists, no cap): vs the oracle's hybrid search
    with a cap above the largest neighbourhood; neighbourhoods beyond 64
    exercise the multi-round emission. Plus the reference's golden
    (cpp/tests/core/NearestNeighborSearch.cpp:166-216)."""
    _lib, reg = _gpu()
    pts, _ = _cloud(8000, 17, dtype)
    rng = np.random.default_rng(4)
    qrs = np.ascontiguousarray(np.concatenate(
        [pts[::2], (pts[:300] + rng.normal(0, 0.02, (300, 3))).astype(dtype),
         np.array([[500, 500, 500]], dtype)]))
    radius = 0.35
    widx, wd2, wcnt = orc.hybrid_search(pts, qrs, radius, 1500)
    assert 130 < wcnt.max() < 1500 and wcnt.min() == 0
    idx, d2, splits = reg.fixed_radius_search(torch.from_numpy(pts).cuda(),
                                              torch.from_numpy(qrs).cuda(),
                                              radius)
    splits = splits.cpu().numpy()
    assert np.array_equal(np.diff(splits), wcnt)
    flat_i = np.concatenate([widx[i, :c] for i, c in enumerate(wcnt)])
    flat_d = np.concatenate([wd2[i, :c] for i, c in enumerate(wcnt)])
    assert np.array_equal(idx.cpu().numpy(), flat_i)
    assert d2.cpu().numpy().tobytes() == flat_d.tobytes()
    ref_pts = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.1], [0.0, 0.0, 0.2],
                        [0.0, 0.1, 0.0], [0.0, 0.1, 0.1], [0.0, 0.1, 0.2],
                        [0.0, 0.2, 0.0], [0.0, 0.2, 0.1], [0.0, 0.2, 0.2],
                        [0.1, 0.0, 0.0]], dtype)
    q1 = np.array([[0.064705, 0.043921, 0.087843]], dtype)
    i1, dd1, s1 = reg.fixed_radius_search(torch.from_numpy(ref_pts).cuda(),
                                          torch.from_numpy(q1).cuda(), 0.1)
    assert i1.cpu().numpy().tolist() == [1, 4]
    assert s1.cpu().numpy().tolist() == [0, 2]
    assert np.allclose(dd1.cpu().numpy(), [0.00626358, 0.00747938], atol=1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_estimate_normals_cube_golden_through_gpu(dtype):
    """cpp/tests/t/geometry/PointCloud.cpp:630-668: the reference's own
    EstimateNormals value test (signs included), all three search variants, on
    an 8-point cloud (far below one wave: the searches' small-n paths)."""
    from test_oracle_goldens import CUBE, CUBE_NORMALS
    _lib, reg = _gpu()
    tp = torch.from_numpy(CUBE.astype(dtype)).cuda()
    for kw in ({"max_nn": 4, "radius": 2.0}, {"max_nn": 4},
               {"max_nn": None, "radius": 1.1}):
        got = reg.estimate_normals(tp, **kw).cpu().numpy()
        assert np.allclose(got, CUBE_NORMALS, rtol=1e-4, atol=1e-4), kw


def test_nns_coincident_and_tie_break_cases_through_gpu():
    """cpp/tests/core/NearestNeighborSearch.cpp:495-533,780-794 (C1, C4)."""
    _lib, reg = _gpu()
    from test_oracle_goldens import KNN_PTS
    tp = torch.from_numpy(KNN_PTS).cuda()
    q = torch.tensor([[0.0, 0.1, 0.1]], device="cuda")    # dataset point 4
    idx, d2 = reg.knn_search(tp, q, 3)
    assert int(idx[0, 0]) == 4 and float(d2[0, 0]) == 0.0
    ridx, rd2, splits = reg.fixed_radius_search(tp, q, 0.05)
    assert splits.cpu().tolist() == [0, 1]
    assert int(ridx[0]) == 4 and float(rd2[0]) == 0.0
    tri = torch.eye(3, device="cuda")
    idx, d2 = reg.knn_search(tri, torch.zeros((1, 3), device="cuda"), 3)
    assert idx.cpu().tolist() == [[0, 1, 2]]
    assert np.allclose(d2.cpu().numpy(), 1.0, atol=1e-5)
