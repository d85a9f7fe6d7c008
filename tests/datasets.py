"""Deterministic inputs shared by the tests, smoke() and bench.py."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# tst/point2point.cpp:93-101 — Rx(0.3) * Ry(0.4) * Rz(0.5), t = (10.5, 10.2, 0.1)
FIXTURE_T = np.array([10.5, 10.2, 0.1])
# (t, log R) of that pose: where LM must arrive from x0 = 0
FIXTURE_X = np.array([10.5, 10.2, 0.1, 0.38994502377414, 0.31542006718654, 0.54962215934141])
X_ZERO = np.zeros(6)
X_GENERIC = np.array([0.5, -0.3, 0.2, 0.1, -0.2, 0.3])


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def fixture_rotation():
    return _rot(0, 0.3) @ _rot(1, 0.4) @ _rot(2, 0.5)


def apply_fixture_transform(src):
    """tgt = T * [p; 1] with the association of a 4x4 matrix-vector product
    (tst/point2point.cpp:109-123)."""
    R = fixture_rotation()
    t = FIXTURE_T
    out = np.empty_like(src)
    for r in range(3):
        out[:, r] = ((R[r, 0] * src[:, 0] + R[r, 1] * src[:, 1]) + R[r, 2] * src[:, 2]) + t[r]
    return out


def facade_pair():
    q = np.load(os.path.join(GOLDEN, "fachada_xyz_1e8.npz"))["xyz_1e8"]
    src = q.astype(np.float64) / 1e8
    return src, apply_fixture_transform(src)


def synthetic_pair(n, seed=42, noise=0.0, dtype=np.float64):
    """src ~ U[0,10]^3 (the shape of tst/parallel.cpp:39-47); tgt = R src + t with the fixture
    pose, optionally + N(0, noise^2)."""
    rng = np.random.default_rng(seed)
    src = rng.random((n, 3)) * 10.0
    tgt = apply_fixture_transform(src)
    if noise > 0:
        tgt = tgt + rng.normal(0.0, noise, size=tgt.shape)
    return src.astype(dtype), tgt.astype(dtype)


def synthetic_camera(n, seed=7, x_true=(-0.01, 0.02, -0.058, 0.018, -0.0013, 0.027)):
    """Points in front of the camera of tst/camera_calibration.cpp and the rounded pixels their
    projection under x_true lands on."""
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 4))
    pts[:, 0] = rng.uniform(1.5, 4.0, n)     # depth along the laser x axis
    pts[:, 1] = rng.uniform(-1.0, 1.0, n)
    pts[:, 2] = rng.uniform(-0.5, 0.8, n)
    pts[:, 3] = 1.0
    K = np.array([[586.122314453125, 0, 638.8477694496105, 0],
                  [0, 722.3973388671875, 323.031267074588, 0], [0, 0, 1, 0]])
    C = np.eye(4)
    C[:3, :3] = _rot(0, np.pi / 2) @ _rot(2, np.pi / 2)
    x = np.asarray(x_true, dtype=np.float64)
    th = np.linalg.norm(x[3:])
    a = x[3:] / th
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    T[:3, 3] = x[:3]
    o = (K @ T @ C @ pts.T).T
    pix = np.rint(o[:, :2] / o[:, 2:3]).astype(np.int32)
    return pts, pix
