"""SE(3) manifold update (the reference's "TODO Manifold operation",
src/levenberg_marquadt_dyn.cpp:82-83, include/moptimizer/manifold.h) — the host-side pieces, no GPU:
mopt_se3_plus, the left-perturbation Jacobian of the oracle's point2point model, and what the update
buys the LM loop."""
import numpy as np

from tests import datasets as ds
from tests import oracle_binding as ob


def _exp(w):
    th = np.linalg.norm(w)
    if th < 1e-15:
        return np.eye(3)
    a = w / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def test_se3_plus_composes_a_left_perturbation():
    import moptimizer_0_amd as mo
    rng = np.random.default_rng(3)
    for _ in range(200):
        x = np.concatenate([rng.normal(0, 3, 3), rng.normal(0, 0.6, 3)])
        d = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3)])
        out = mo.capi.se3_plus(x, d)
        R = _exp(d[3:]) @ _exp(x[3:])
        t = _exp(d[3:]) @ x[:3] + d[:3]
        assert np.abs(_exp(out[3:]) - R).max() < 1e-12
        assert np.abs(out[:3] - t).max() < 1e-12
    # zero step: the pose is unchanged (Log(Exp(w)) = w away from pi)
    x = ds.X_GENERIC
    assert np.abs(mo.capi.se3_plus(x, np.zeros(6)) - x).max() < 1e-14


def test_left_jacobian_is_the_derivative_of_the_residual_under_se3_plus(oracle):
    """b = J^T r and H = J^T J of the left-perturbation mode against central differences of the cost
    along x (+) eps e_j: d/d eps (sum r^T r / 2) = b_j, at a pose far from the identity."""
    import moptimizer_0_amd as mo
    src, tgt = ds.synthetic_pair(500, seed=9, noise=0.05)
    x = np.array([0.4, -1.1, 2.0, 1.2, -0.9, 1.5])  # |w| = 2.1 rad
    H, b, s = oracle.p2p_linearize(src, tgt, x, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_LEFT)
    eps = 1e-6
    for j in range(6):
        e = np.zeros(6)
        e[j] = eps
        cp = oracle.p2p_cost(src, tgt, mo.capi.se3_plus(x, e))
        cm = oracle.p2p_cost(src, tgt, mo.capi.se3_plus(x, -e))
        assert abs((cp - cm) / (4 * eps) - b[j]) < 1e-6 * max(1.0, abs(b[j])), j
        # the cost is exactly quadratic along a translation: its second difference is H_jj
        if j < 3:
            big = np.zeros(6)
            big[j] = 0.5
            c2p = oracle.p2p_cost(src, tgt, mo.capi.se3_plus(x, big))
            c2m = oracle.p2p_cost(src, tgt, mo.capi.se3_plus(x, -big))
            assert abs((c2p - 2 * s + c2m) / (2 * 0.25) - H[j, j]) < 1e-9 * H[j, j]
    # the Euclidean-parameter Jacobian [I | -skew(p)] is NOT that derivative away from R = I
    He, be, _ = oracle.p2p_linearize(src, tgt, x, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_ROW_MAJOR)
    assert np.abs(be[3:] - b[3:]).max() > 1e-2 * np.abs(b[3:]).max()


def test_manifold_update_converges_where_the_euclidean_one_crawls(oracle):
    """From a start 2.5 rad away from the fixture rotation, LM with the left Jacobian and the
    x (+) delta update reaches the pose; the reference's Euclidean update with [I | -skew(p)]
    (a first-order Jacobian, exact at w = 0 only) needs more than twice the iterations."""
    src, tgt = ds.synthetic_pair(2000, seed=12, noise=0.0)
    R0 = _exp(np.array([0.0, 0.0, 2.5])) @ ds.fixture_rotation()
    th = np.arccos((np.trace(R0) - 1) / 2)
    w0 = th / (2 * np.sin(th)) * np.array([R0[2, 1] - R0[1, 2], R0[0, 2] - R0[2, 0], R0[1, 0] - R0[0, 1]])
    x0 = np.concatenate([ds.FIXTURE_T + 1.0, w0])
    xm, sm, im = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                     layout=ob.LAYOUT_LEFT | ob.MANIFOLD_UPDATE, max_iter=200)
    xe, se, ie = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                     layout=ob.LAYOUT_ROW_MAJOR, max_iter=200)
    assert sm == 0 and np.abs(_exp(xm[3:]) - ds.fixture_rotation()).max() < 1e-8
    assert np.abs(xm[:3] - ds.FIXTURE_T).max() < 1e-7
    assert ie > 2 * im, (im, ie)


def test_se3_plus_right_composes_on_the_right():
    """The composition of the reference's own sketches: `parameter_matrix * Exp(delta)`
    (tst/manifold.cpp:47); `rot_ * rhs_rot`, `lin_ += delta` (tst/state_model.cpp:28-34)."""
    import moptimizer_0_amd as mo
    rng = np.random.default_rng(4)
    for _ in range(200):
        x = np.concatenate([rng.normal(0, 3, 3), rng.normal(0, 0.6, 3)])
        d = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3)])
        out = mo.capi.se3_plus_right(x, d)
        assert np.abs(_exp(out[3:]) - _exp(x[3:]) @ _exp(d[3:])).max() < 1e-12
        assert np.abs(out[:3] - (x[:3] + d[:3])).max() < 1e-14
    x = ds.X_GENERIC
    assert np.abs(mo.capi.se3_plus_right(x, np.zeros(6)) - x).max() < 1e-14


def test_right_jacobian_is_the_derivative_of_the_residual_under_se3_plus_right(oracle):
    """b = J^T r of the right-perturbation mode, J = [I | -R skew(p)], against central differences of
    the cost along x (+) eps e_j composed on the right, at a pose 2.1 rad from the identity; the
    translation block is the Euclidean one (t <- t + delta_t), so H_jj is the exact second difference."""
    import moptimizer_0_amd as mo
    src, tgt = ds.synthetic_pair(500, seed=9, noise=0.05)
    x = np.array([0.4, -1.1, 2.0, 1.2, -0.9, 1.5])
    H, b, s = oracle.p2p_linearize(src, tgt, x, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_RIGHT)
    eps = 1e-6
    for j in range(6):
        e = np.zeros(6)
        e[j] = eps
        cp = oracle.p2p_cost(src, tgt, mo.capi.se3_plus_right(x, e))
        cm = oracle.p2p_cost(src, tgt, mo.capi.se3_plus_right(x, -e))
        assert abs((cp - cm) / (4 * eps) - b[j]) < 1e-6 * max(1.0, abs(b[j])), j
        if j < 3:
            big = np.zeros(6)
            big[j] = 0.5
            c2p = oracle.p2p_cost(src, tgt, mo.capi.se3_plus_right(x, big))
            c2m = oracle.p2p_cost(src, tgt, mo.capi.se3_plus_right(x, -big))
            assert abs((c2p - 2 * s + c2m) / (2 * 0.25) - H[j, j]) < 1e-9 * H[j, j]
    # it differs from the left form (same residual, another parametrisation of the step) ...
    Hl, bl, _ = oracle.p2p_linearize(src, tgt, x, cost_class=ob.ANALYTIC_DYN, layout=ob.LAYOUT_LEFT)
    assert np.abs(bl[3:] - b[3:]).max() > 1e-2 * np.abs(b[3:]).max()
    # ... and has the same translation block
    assert np.abs(Hl[:3, :3] - H[:3, :3]).max() == 0.0 and np.abs(bl[:3] - b[:3]).max() == 0.0


def test_right_manifold_update_converges_like_the_left_one(oracle):
    src, tgt = ds.synthetic_pair(2000, seed=12, noise=0.0)
    R0 = _exp(np.array([0.0, 0.0, 2.5])) @ ds.fixture_rotation()
    th = np.arccos((np.trace(R0) - 1) / 2)
    w0 = th / (2 * np.sin(th)) * np.array([R0[2, 1] - R0[1, 2], R0[0, 2] - R0[2, 0], R0[1, 0] - R0[0, 1]])
    x0 = np.concatenate([ds.FIXTURE_T + 1.0, w0])
    xr, sr, ir = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                     layout=ob.LAYOUT_RIGHT | ob.MANIFOLD_UPDATE_RIGHT, max_iter=200)
    xe, se, ie = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                     layout=ob.LAYOUT_ROW_MAJOR, max_iter=200)
    assert sr == 0 and np.abs(_exp(xr[3:]) - ds.fixture_rotation()).max() < 1e-8
    assert np.abs(xr[:3] - ds.FIXTURE_T).max() < 1e-7
    assert ie > 2 * ir, (ir, ie)
