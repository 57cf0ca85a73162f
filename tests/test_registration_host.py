"""Host-side (no GPU) pieces of the registration path inside the product
library: the point-to-point R, t from the reduced moments
(o3dmi_compute_rt_p2point <- ComputeRtPointToPointCPU after its reduction,
RegistrationCPU.cpp:640-650) against LAPACK's SVD and the oracle."""
import ctypes as C

import numpy as np
import pytest

import _oracle as orc


def _lib():
    from open3d_amd import _lib as L
    return L


def _moments(s, t):
    """The 16 sums o3dmi_icp_p2point_accumulate produces, in float64."""
    s = s.astype(np.float64)
    t = t.astype(np.float64)
    out = np.zeros(16)
    out[0:3] = s.sum(0)
    out[3:6] = t.sum(0)
    out[6:15] = (t.T @ s).reshape(-1)
    out[15] = s.shape[0]
    return out


def _rt(sums):
    L = _lib()
    R = np.zeros(9)
    t = np.zeros(3)
    st = L.lib().o3dmi_compute_rt_p2point(L.f64p(np.ascontiguousarray(sums)),
                                          L.f64p(R), L.f64p(t))
    return st, R.reshape(3, 3), t


def _rt_numpy(s, t):
    ms, mt = s.mean(0), t.mean(0)
    S = (t - mt).T @ (s - ms) / s.shape[0]
    U, D, VT = np.linalg.svd(S)
    Sg = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(VT.T) < 0:
        Sg[-1, -1] = -1
    R = U @ (Sg @ VT)
    return R, mt - R @ ms


def _rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


@pytest.mark.parametrize("seed", range(6))
def test_rt_recovers_rigid_motion_and_matches_lapack(seed):
    rng = np.random.default_rng(seed)
    s = rng.random((500, 3)) * 4 - 2
    R0, t0 = _rot(rng), rng.standard_normal(3)
    t = s @ R0.T + t0 + 1e-3 * rng.standard_normal(s.shape)
    st, R, tt = _rt(_moments(s, t))
    Rn, tn = _rt_numpy(s, t)
    assert st == 0
    assert np.abs(R - Rn).max() < 1e-12 and np.abs(tt - tn).max() < 1e-12
    assert np.abs(R - R0).max() < 1e-3
    assert abs(np.linalg.det(R) - 1) < 1e-13
    # and the oracle's two-pass restatement of the reference
    Ro, to, c = orc.compute_rt_p2point(s, t, np.arange(s.shape[0]))
    assert c == s.shape[0]
    assert np.abs(R - Ro).max() < 1e-12 and np.abs(tt - to).max() < 1e-12


def test_rt_reflection_and_planar_sets():
    rng = np.random.default_rng(42)
    # planar source (sigma_3 = 0) under a proper rotation
    s = rng.random((300, 3)) * 2
    s[:, 2] = 0.5
    R0, t0 = _rot(rng), np.array([0.3, -0.1, 0.2])
    t = s @ R0.T + t0
    st, R, tt = _rt(_moments(s, t))
    assert st == 0 and np.abs(R - R0).max() < 1e-10
    assert np.abs(tt - t0).max() < 1e-10
    # mirrored target: the best proper rotation, not the reflection
    s = rng.standard_normal((400, 3))
    M = np.diag([1.0, 1.0, -1.0])
    t = s @ (R0 @ M).T
    st, R, tt = _rt(_moments(s, t))
    Rn, tn = _rt_numpy(s, t)
    assert st == 0 and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.abs(R - Rn).max() < 1e-10 and np.abs(tt - tn).max() < 1e-10
    # collinear source (rank one): still a proper rotation mapping the line
    s = np.outer(np.linspace(-1, 1, 50), [1.0, 2.0, -0.5])
    t = s @ R0.T + t0
    st, R, tt = _rt(_moments(s, t))
    assert st == 0 and abs(np.linalg.det(R) - 1) < 1e-12
    assert np.abs(s @ R.T + tt - t).max() < 1e-9


def test_rt_no_correspondence_is_an_error():
    """RegistrationCPU.cpp:548-550 "No valid correspondence present."."""
    L = _lib()
    st, R, t = _rt(np.zeros(16))
    assert st == L.O3DMI_ERR_NO_INLIERS if hasattr(L, "O3DMI_ERR_NO_INLIERS") \
        else st != 0
    assert b"No valid correspondence" in L.lib().o3dmi_last_error()
    # coincident points: pure translation
    s = np.tile([[1.0, 2.0, 3.0]], (5, 1))
    st, R, t = _rt(_moments(s, s + [0.5, 0, -1]))
    assert st == 0 and np.array_equal(R, np.eye(3))
    assert np.allclose(t, [0.5, 0, -1], atol=1e-12)


def test_symmetric_pose_to_transformation_matches_oracle():
    """PoseToSymmetricTransformation (TransformationConverter.cpp:106-133):
    the library's host function vs the oracle restatement, and the closed form
    for a pure half-angle rotation about z."""
    L = _lib()
    rng = np.random.default_rng(7)
    for _ in range(20):
        pose = rng.standard_normal(6) * [0.3, 0.3, 0.3, 1, 1, 1]
        ms, mt = rng.standard_normal(3), rng.standard_normal(3)
        T = np.zeros((4, 4))
        L.lib().o3dmi_symmetric_pose_to_transformation(
            L.f64p(pose), L.f64p(ms), L.f64p(mt), L.f64p(T))
        want = orc.symmetric_pose_to_transformation(pose, ms, mt)
        assert np.abs(T - want).max() < 1e-15
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    # g = tan(theta) z, no translation, zero means: rotation by 2 theta about z
    theta = 0.2
    pose = np.array([0, 0, np.tan(theta), 0, 0, 0.0])
    T = np.zeros((4, 4))
    z = np.zeros(3)
    L.lib().o3dmi_symmetric_pose_to_transformation(L.f64p(pose), L.f64p(z),
                                                   L.f64p(z), L.f64p(T))
    c, s = np.cos(2 * theta), np.sin(2 * theta)
    assert np.allclose(T[:3, :3], [[c, -s, 0], [s, c, 0], [0, 0, 1]],
                       atol=1e-15)
    # zero pose: identity rotation, translation = target_mean - source_mean
    L.lib().o3dmi_symmetric_pose_to_transformation(
        L.f64p(np.zeros(6)), L.f64p(np.array([1.0, 2, 3])),
        L.f64p(np.array([2.0, 2, 2])), L.f64p(T))
    assert np.array_equal(T[:3, :3], np.eye(3))
    assert np.array_equal(T[:3, 3], [1.0, 0, -1])
