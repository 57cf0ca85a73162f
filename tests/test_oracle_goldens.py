"""Pin the CPU oracle against the reference's own in-source known answers.

Every vector below is transcribed from the reference's unit tests (file:line in
each test's docstring); nothing here needs a GPU.
"""
import numpy as np
import pytest

import _oracle as orc

# cpp/tests/t/pipelines/registration/TransformationEstimation.cpp:33-86
SRC = np.array([[1.15495, 2.40671, 1.15061], [1.81481, 2.06281, 1.71927],
                [0.888322, 2.05068, 2.04879], [3.78842, 1.70788, 1.30246],
                [1.8437, 2.22894, 0.986237], [2.95706, 2.20180, 0.987878],
                [1.72644, 1.24356, 1.93486], [0.922024, 1.14872, 2.34317],
                [3.70293, 1.85134, 1.15357], [3.06505, 1.30386, 1.55279],
                [0.634826, 1.04995, 2.47046], [1.40107, 1.37469, 1.09687],
                [2.93002, 1.96242, 1.48532], [3.74384, 1.30258, 1.30244]])
TGT = np.array([[2.41766, 2.05397, 1.74994], [1.37848, 2.19793, 1.66553],
                [2.24325, 2.27183, 1.33708], [3.09898, 1.98482, 1.77401],
                [1.81615, 1.48337, 1.49697], [3.01758, 2.20312, 1.51502],
                [2.38836, 1.39096, 1.74914], [1.30911, 1.4252, 1.37429],
                [3.16847, 1.39194, 1.90959], [1.59412, 1.53304, 1.58040],
                [1.34342, 2.19027, 1.30075]])
TGT_N = np.array([[-0.00850160, -0.22355, -0.519574],
                  [0.257463, -0.0738755, -0.698319],
                  [0.0574301, -0.484248, -0.409929],
                  [-0.0123503, -0.230172, -0.520720],
                  [0.355904, -0.142007, -0.720467],
                  [0.0674038, -0.418757, -0.458602],
                  [0.226091, 0.258253, -0.874024],
                  [0.43979, 0.122441, -0.574998],
                  [0.109144, 0.180992, -0.762368],
                  [0.273325, 0.292013, -0.903111],
                  [0.385407, -0.212348, -0.277818]])
CORR = np.array([10, 1, 1, 3, 2, 5, 9, 7, 5, 8, 7, 7, 5, 8], np.int64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_p2plane_rmse_golden(dtype):
    """ComputeRMSEPointToPlane: 0.335499 (TransformationEstimation.cpp:148)."""
    r = orc.p2plane_rmse(SRC.astype(dtype), TGT.astype(dtype),
                         TGT_N.astype(dtype), CORR)
    assert abs(r - 0.335499) < 1e-4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("acc_double", [False, True])
def test_p2plane_transformation_golden(dtype, acc_double):
    """ComputeTransformationPointToPlane -> RMSE 0.601422 after applying it
    (TransformationEstimation.cpp:176)."""
    s, t, n = SRC.astype(dtype), TGT.astype(dtype), TGT_N.astype(dtype)
    A = orc.p2plane_accumulate(s, t, n, CORR, accumulate_double=acc_double)
    assert A[28] == 14
    st, pose, residual, count = orc.decode_and_solve6x6(A)
    assert st == 0 and count == 14
    T = orc.pose_to_transformation(pose)
    s2 = orc.transform_points(T, s)
    r = orc.p2plane_rmse(s2, t, n, CORR)
    assert abs(r - 0.601422) < 1e-4


def _p2point_rmse(s, t, corr):
    """TransformationEstimationPointToPoint::ComputeRMSE,
    TransformationEstimation.cpp:101-130."""
    m = corr >= 0
    d = s[m].astype(np.float64) - t[corr[m]].astype(np.float64)
    return float(np.sqrt((d * d).sum() / m.sum()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("acc_double", [False, True])
def test_p2point_transformation_golden(dtype, acc_double):
    """ComputeRMSEPointToPoint 0.706437; ComputeTransformationPointToPoint ->
    RMSE 0.578255 after applying it (TransformationEstimation.cpp:103,130)."""
    s, t = SRC.astype(dtype), TGT.astype(dtype)
    assert abs(_p2point_rmse(s, t, CORR) - 0.706437) < 1e-4
    R, tr, c = orc.compute_rt_p2point(s, t, CORR, accumulate_double=acc_double)
    assert c == 14
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, tr
    s2 = orc.transform_points(T, s)
    assert abs(_p2point_rmse(s2, t, CORR) - 0.578255) < 1e-4


def test_hybrid_search_golden():
    """NNSPermuteDevices.HybridSearch (cpp/tests/core/NearestNeighborSearch.cpp:321-377)."""
    pts = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.1], [0.0, 0.0, 0.2],
                    [0.0, 0.1, 0.0], [0.0, 0.1, 0.1], [0.0, 0.1, 0.2],
                    [0.0, 0.2, 0.0], [0.0, 0.2, 0.1], [0.0, 0.2, 0.2],
                    [0.1, 0.0, 0.0]], np.float32)
    q = np.array([[0.064705, 0.043921, 0.087843]], np.float32)
    for brute in (False, True):
        idx, dist, cnt = orc.hybrid_search(pts, q, 0.1, 3, brute=brute)
        assert idx.tolist() == [[1, 4, -1]]
        assert np.allclose(dist, [[0.00626358, 0.00747938, 0]], rtol=1e-5,
                           atol=1e-8)
        assert cnt.tolist() == [2]


# cpp/tests/t/pipelines/registration/TransformationEstimation.cpp:220-420
def _axis_angle(angle, axis):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


SYM_EXPECTED = np.eye(4)
SYM_EXPECTED[:3, :3] = _axis_angle(0.3, [1.0, 2.0, -1.0])
SYM_EXPECTED[:3, 3] = [0.2, -0.1, 0.15]
SYM_SRC = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0],
                    [0.0, 0.0, 1.0], [1.0, 1.0, 0.0], [1.0, 0.0, 1.0],
                    [0.0, 1.0, 1.0], [1.0, 1.0, 1.0], [2.0, -1.0, 0.5],
                    [-0.5, 1.5, 2.0]])
SYM_NRM = np.array([[1.0, 2.0, 3.0], [2.0, -1.0, 1.0], [-1.0, 3.0, 2.0],
                    [3.0, 1.0, -2.0], [-2.0, -1.0, 3.0], [1.0, -3.0, 2.0],
                    [-3.0, 2.0, 1.0], [2.0, 3.0, -1.0], [1.0, 1.0, -2.0],
                    [-2.0, 1.0, -3.0]])
SYM_NOISE = np.array([[0.02, -0.01, 0.0], [-0.03, 0.02, 0.01],
                      [0.0, 0.04, -0.02], [0.01, -0.03, 0.03],
                      [-0.04, 0.0, 0.02], [0.03, 0.01, -0.04],
                      [-0.02, -0.02, 0.03], [0.04, -0.03, -0.01],
                      [0.85, -0.55, 0.45], [-0.65, 0.70, -0.50]])
SYM_ROBUST_EXPECTED = np.array(
    [[0.978808528971923, 0.011598608561290, -0.204448858865149,
      0.374573231881982],
     [-0.033468146467010, 0.994030976876383, -0.103837855246761,
      0.105509532797345],
     [0.202024124262134, 0.108479902699193, 0.973354182159039,
      -0.112542276940963],
     [0.0, 0.0, 0.0, 1.0]])


def sym_case(dtype, n, noise=False):
    """Source / target clouds of the reference's symmetric tests: normals
    normalised in the dtype (PointCloud::NormalizeNormals), target = source
    moved by SYM_EXPECTED with the dtype's TransformPoints / TransformNormals
    arithmetic."""
    s = SYM_SRC[:n].astype(dtype)
    nr = SYM_NRM[:n].astype(dtype)
    nr = (nr / np.sqrt((nr * nr).sum(1, keepdims=True))).astype(dtype)
    t = orc.transform_points(SYM_EXPECTED, s)
    tn = orc.transform_normals(SYM_EXPECTED, nr)
    if noise:
        t = (t + SYM_NOISE[:n].astype(dtype)).astype(dtype)
    return s, nr, t, tn


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("acc_double", [False, True])
def test_symmetric_transformation_golden(dtype, acc_double):
    """ComputeTransformationSymmetric (TransformationEstimation.cpp:220-420):
    an exact rigid motion is recovered in one step (also with alternating
    target-normal signs), no correspondence -> identity, and the Cauchy-kernel
    case reproduces the reference's printed matrix."""
    tol = 1e-4 if dtype == np.float32 else 1e-8
    s, nr, t, tn = sym_case(dtype, 6)
    corr = np.arange(6)
    st, T, _ = orc.compute_transformation_symmetric(
        s, t, nr, tn, corr, accumulate_double=acc_double)
    assert st == 0 and np.allclose(T, SYM_EXPECTED, rtol=tol, atol=tol)
    signs = np.array([[-1.0], [1.0], [-1.0], [1.0], [-1.0], [1.0]], dtype)
    st, T, _ = orc.compute_transformation_symmetric(
        s, t, nr, (tn * signs).astype(dtype), corr,
        accumulate_double=acc_double)
    assert st == 0 and np.allclose(T, SYM_EXPECTED, rtol=tol, atol=tol)
    st, T, _ = orc.compute_transformation_symmetric(
        s, t, nr, tn, np.full(6, -1), accumulate_double=acc_double)
    assert st == 0 and np.array_equal(T, np.eye(4))
    s, nr, t, tn = sym_case(dtype, 10, noise=True)
    st, T, _ = orc.compute_transformation_symmetric(
        s, t, nr, tn, np.arange(10), method=3, scaling=0.5, shape=1.0,
        accumulate_double=acc_double)
    assert st == 0
    assert np.allclose(T, SYM_ROBUST_EXPECTED, rtol=tol, atol=tol)


KNN_PTS = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.1], [0.0, 0.0, 0.2],
                    [0.0, 0.1, 0.0], [0.0, 0.1, 0.1], [0.0, 0.1, 0.2],
                    [0.0, 0.2, 0.0], [0.0, 0.2, 0.1], [0.0, 0.2, 0.2],
                    [0.1, 0.0, 0.0], [0.1, 0.0, 0.1], [0.1, 0.1, 0.0]],
                   np.float32)
KNN_Q = np.array([[0.064705, 0.043921, 0.087843]], np.float32)
KNN_IDX = [10, 1, 4, 9, 11, 0, 3, 2, 5, 7, 6, 8]
KNN_D2 = [0.00332258, 0.00626358, 0.00747938, 0.0108912, 0.0121070, 0.0138322,
          0.015048, 0.018695, 0.0199108, 0.0286952, 0.0362638, 0.0411266]


def test_knn_search_golden():
    """NNSPermuteDevices.KnnSearch (cpp/tests/core/NearestNeighborSearch.cpp:
    36-111): k = 3 and k > dataset size (row width = dataset size)."""
    idx, d2 = orc.knn_search(KNN_PTS, KNN_Q, 3)
    assert idx.tolist() == [KNN_IDX[:3]]
    assert np.allclose(d2, [KNN_D2[:3]], rtol=1e-5, atol=1e-8)
    idx, d2 = orc.knn_search(KNN_PTS, KNN_Q, 14)
    assert idx.shape == (1, 12) and idx.tolist() == [KNN_IDX]
    assert np.allclose(d2, [KNN_D2], rtol=1e-5, atol=1e-8)


def test_hybrid_search_radius_is_strict():
    """nanoflann v1.5.0 RadiusResultSet::addPoint keeps dist < radius only."""
    pts = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0]], np.float32)
    q = np.array([[0.25, 0.0, 0.0]], np.float32)
    idx, dist, cnt = orc.hybrid_search(pts, q, 0.25, 2)
    assert cnt.tolist() == [0] and idx.tolist() == [[-1, -1]]
    idx, dist, cnt = orc.hybrid_search(pts, q, 0.2500001, 2)
    assert cnt.tolist() == [2] and idx.tolist() == [[0, 1]]  # tie: low index


def test_hybrid_search_large_offset_recipe():
    """Seeded stress recipe of NNSParityTest.HybridSearchLargeOffsetParityCPU
    (NearestNeighborSearch.cpp:831-869): n=4000, +1000 m offset, r=0.05, k=1;
    expects > n/2 matches. Grid and brute-force variants must agree exactly."""
    rng = np.random.RandomState(7)
    n = 4000
    base = (1000.0 + rng.uniform(0, 3, (n, 3))).astype(np.float32)
    qrs = (base + rng.uniform(-0.02, 0.02, (n, 3)).astype(np.float32))
    i1, d1, c1 = orc.hybrid_search(base, qrs, 0.05, 1)
    i2, d2, c2 = orc.hybrid_search(base, qrs, 0.05, 1, brute=True)
    assert c1.sum() > n // 2
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert np.array_equal(c1, c2)


def test_robust_kernel_goldens():
    """RegistrationPermuteDevices.RobustKernel
    (cpp/tests/t/pipelines/registration/Registration.cpp:411-490)."""
    expected = {0: 1.0, 1: 1.0204, 2: 1.0, 3: 0.5101, 4: 0.260202,
                5: 0.00156816, 6: 0.714213}
    for f64 in (False, True):
        for m, e in expected.items():
            assert abs(orc.robust_weight(m, 1.0, 1.0, 0.98, f64) - e) < 1e-3
        for shape, e in {2.0: 1.0, 0.0: 0.675584, -2.0: 0.650259,
                         1.0: 0.714213}.items():
            assert abs(orc.robust_weight(6, 1.0, shape, 0.98, f64) - e) < 1e-3


def test_solve_golden():
    """LinalgPermuteDevices.Solve (cpp/tests/core/Linalg.cpp:454-480)."""
    x = orc.solve([[3, 1], [1, 2]], [9, 8])
    assert np.allclose(x, [2, 3], atol=1e-12)
    with pytest.raises(RuntimeError):
        orc.solve(np.zeros((2, 2)), [9, 8])


def test_pose_to_transformation_identity():
    """cpp/tests/t/pipelines/TransformationConverter.cpp:37-48."""
    assert np.array_equal(orc.pose_to_transformation(np.zeros(6)), np.eye(4))


def test_pose_to_transformation_is_rz_ry_rx():
    a, b, g = 0.1, -0.2, 0.3
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)],
                   [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0],
                   [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0],
                   [0, 0, 1]])
    T = orc.pose_to_transformation([a, b, g, 1, 2, 3])
    assert np.allclose(T[:3, :3], Rz @ Ry @ Rx, atol=1e-15)
    assert np.array_equal(T[:3, 3], [1, 2, 3])


def test_hashmap_semantics():
    """Duplicate keys -> exactly one success mask; Find after Activate
    (cpp/tests/core/HashMap.cpp:138-191; VoxelBlockGrid Indexing test: 5 keys /
    3 unique, cpp/tests/t/geometry/VoxelBlockGrid.cpp:199-239)."""
    h = orc.HashMap(10)
    keys = np.array([[1, 2, 3], [1, 2, 3], [-1, 0, 5], [1, 2, 3], [7, 7, 7]],
                    np.int32)
    buf, masks = h.activate(keys)
    assert masks.sum() == 3 and h.size() == 3
    assert masks.tolist() == [True, False, True, False, True]
    buf2, m2 = h.find(keys)
    assert m2.all() and buf2[0] == buf2[1] == buf2[3]
    assert len(set(buf2.tolist())) == 3
    kb = h.key_buffer()
    assert np.array_equal(kb[buf2], keys)
    _, m3 = h.find(np.array([[9, 9, 9]], np.int32))
    assert not m3[0]
    buf4, m4 = h.activate(keys)
    assert not m4.any() and h.size() == 3


# cpp/tests/t/geometry/PointCloud.cpp:630-668 (EstimateNormals): the unit cube's
# corners; hybrid (4, 2.0), KNN (4) and radius (1.1) all see a corner plus its
# three edge neighbours.
CUBE = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
                 [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]], np.float64)
S3 = 0.57735
CUBE_NORMALS = np.array([[S3, S3, S3], [-S3, -S3, S3], [S3, -S3, S3],
                         [-S3, S3, S3], [-S3, S3, S3], [S3, -S3, S3],
                         [-S3, -S3, S3], [S3, S3, S3]])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_estimate_normals_cube_golden(dtype):
    pts = CUBE.astype(dtype)
    lists = {
        "hybrid": orc.hybrid_search(pts, pts, 2.0, 4)[::2],
        "knn": (orc.knn_search(pts, pts, 4)[0], np.full(8, 4, np.int32)),
        "radius": orc.hybrid_search(pts, pts, 1.1, 8)[::2],
    }
    for name, (idx, cnt) in lists.items():
        assert np.array_equal(cnt, np.full(8, 4)), name
        nrm = orc.normals_from_covariances(
            orc.estimate_covariances(pts, idx, cnt))
        assert np.allclose(nrm, CUBE_NORMALS, rtol=1e-4, atol=1e-4), name


def test_voxel_down_sample_golden():
    """cpp/tests/t/geometry/PointCloud.cpp:1300-1314: four points in one unit
    voxel -> their mean."""
    pts = np.array([[0.1, 0.3, 0.9], [0.9, 0.2, 0.4], [0.3, 0.6, 0.8],
                    [0.2, 0.4, 0.2]], np.float32)
    down, _ = orc.voxel_down_sample(pts, None, 1.0)
    assert down.shape == (1, 3)
    assert np.allclose(down, [[0.375, 0.375, 0.575]], rtol=1e-5, atol=1e-8)


def test_nns_coincident_and_tie_break_cases():
    """cpp/tests/core/NearestNeighborSearch.cpp:495-533,780-794: a query on a
    dataset point has distance exactly 0 (C1); equidistant neighbours come in
    index order (C4)."""
    q = np.array([[0.0, 0.1, 0.1]], np.float32)           # dataset point 4
    idx, d2 = orc.knn_search(KNN_PTS, q, 3)
    assert idx[0, 0] == 4 and d2[0, 0] == 0.0
    ridx, rd2, rcnt = orc.hybrid_search(KNN_PTS, q, 0.05, 12)
    assert rcnt[0] == 1 and ridx[0, 0] == 4 and rd2[0, 0] == 0.0
    tri = np.eye(3, dtype=np.float32)
    idx, d2 = orc.knn_search(tri, np.zeros((1, 3), np.float32), 3)
    assert idx.tolist() == [[0, 1, 2]] and np.allclose(d2, 1.0, atol=1e-5)
