"""mopt_lm_minimize — LevenbergMarquadtDynamic::minimize (src/levenberg_marquadt_dyn.cpp:34-119)
with the iteration resident on the device — against the CPU restatement of the same loop over the
CPU restatement of the costs (oracle), and against the host loop driving the HIP cost through the
boundary.

Bar.  Analytic Jacobians: same status, same number of outer iterations, every iterate within 1e-9.
Forward differences: the device forms the transforms at x and x + h_j e_j with its own sincos, which
differs from glibc's in the last bit for some arguments; forward differences amplify a last-bit
difference of R by 1 / h_j = 6.7e7 / |x_j| (the effect tests/test_so3_bitwise.py removes from the
boundary path, where both sides call glibc), so a Jacobian column carries a relative noise of
~1e-8 that is a different realisation on the two sides — as it would be between any two libms.  The
iterates then agree to that noise (measured <= 2e-8 |x|, bound used 1e-7), the fixed point is the
same, and the iteration at which the noise-level stopping tests (cost < 8 eps, rho < 0 with
|delta| < sqrt eps) fire may differ by one or two."""
FD_ITERATE_TOL = 1e-7
import numpy as np
import pytest

from tests import datasets as ds
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

CONVERGED, MAX_ITERATIONS, SMALL_DELTA, NUMERIC_ERROR = 0, 1, 2, 3


def host_lm(cost, jac_mode, x0, max_iter=15, lm_iter=3):
    """The reference's loop written against the blocking boundary calls of ONE HIP cost."""
    x = np.array(x0, dtype=np.float64)
    lam, eps, sweeps = -1.0, np.finfo(np.float64).eps, 0
    it = 0
    while it < max_iter:
        if hasattr(cost, "update"):
            cost.update(x, count_matches=False)  # cost->update(x0), levenberg_marquadt_dyn.cpp:54
        H, b, y0 = cost.linearize(x, jac_mode)
        if abs(y0) < 8 * eps:
            return x, CONVERGED, it
        if lam < 0:
            lam = 1e-9 * np.abs(np.diag(H)).max()
        nu = 2.0
        for _ in range(lm_iter):
            # Eigen's LDLT reads the lower triangle only (levenberg_marquadt_dyn.cpp:78-80): under a
            # non-symmetric covariance H is not symmetric, and the step is that of tril(H) mirrored
            Hl = np.tril(H) + np.tril(H, -1).T
            delta = np.linalg.solve(Hl + lam * np.diag(np.diag(H)), -b)
            xi = x + delta
            yi = cost.compute_cost(xi)
            if np.isnan(yi):
                return x, NUMERIC_ERROR, it
            rho = (y0 - yi) / delta.dot(lam * delta - b)
            if rho < 0:
                if np.abs(delta).max() < np.sqrt(eps):
                    return x, (CONVERGED if abs(yi) < 8 * eps else SMALL_DELTA), it
                lam *= nu
                nu *= 2
                continue
            x = xi
            lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            break
        it += 1
    return x, MAX_ITERATIONS, it


def test_facade_registration_matches_the_cpu_loop(hip_lib, oracle, facade):
    """tst/point2point.cpp:192-217: numerical cost, x0 = 0 -> the fixture pose."""
    mo = hip_lib
    src, tgt = facade
    cost = mo.Point2PointCost(src, tgt)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6))
    xr, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN,
                                            layout=ob.LAYOUT_ROW_MAJOR)
    # exact correspondences: the cost falls through the 8 eps stopping threshold (optimizer.h:26-29)
    # on rounding noise of the forward differences, so the last iteration may or may not be taken
    assert rep["status"] == status == CONVERGED and abs(rep["iterations"] - iters) <= 1, (rep, iters)
    assert np.abs(x - xr).max() < 1e-9
    assert np.abs(x - ds.FIXTURE_X).max() < 1e-8
    # one sweep per evaluated point, none on the host's initiative
    assert rep["sweeps"] <= 1 + 3 * max(iters, 1)
    cost.close()


@pytest.mark.parametrize("n", [1000, 100_000, 1_000_000])
@pytest.mark.parametrize("jac", [0, 2])
def test_iterates_match_the_cpu_loop(hip_lib, oracle, n, jac):
    """Truncating both loops after k outer iterations exposes the k-th iterate."""
    mo = hip_lib
    src, tgt = ds.synthetic_pair(n, seed=5, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    cc = ob.NUMERIC_DYN if jac == 2 else ob.ANALYTIC_DYN
    ks = (1, 2, 3, 5, 15) if n <= 100_000 else (2, 15)
    for k in ks:
        x, rep = mo.capi.lm_minimize([cost], [jac], np.zeros(6), max_iterations=k)
        xr, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=cc,
                                                layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
        tol = 1e-9 if jac == 0 else FD_ITERATE_TOL
        if jac == 0:
            assert (rep["status"], rep["iterations"]) == (status, iters), (k, rep, status, iters)
        else:
            # the noise-level stop (rho < 0 with |delta| < sqrt eps) fires on the last bits of two
            # costs that are stationary to twelve digits by then: one or two iterations apart (4
            # against 5 at n = 1000 with the moments, 4 against 6 evaluated literally;
            # scripts/probes/device_loop_stop.py) — and then one side may run into the iteration
            # limit instead.  Same pose, same cost.
            assert abs(rep["iterations"] - iters) <= 2, (k, rep, iters)
            assert rep["status"] == status or MAX_ITERATIONS in (rep["status"], status), (k, rep, status)
            want_cost = oracle.p2p_cost(src, tgt, xr)
            if k > 2:
                assert abs(rep["cost"] - want_cost) <= 1e-9 * want_cost, (k, rep["cost"], want_cost)
        assert np.abs(x - xr).max() < tol * max(1.0, np.abs(xr).max()), (k, x, xr)
    cost.close()


def test_forward_differences_at_small_parameters(hip_lib, oracle):
    """A registration between nearly aligned clouds: every iterate has 0 < |x_j| < 0.08, where the
    moments form of the forward differences lacks the reference's own per-point cancellation noise
    (eps |R p + t| / h_j, part of what linearization.h:101-105 computes).  Since round 6 the device
    loop applies the blocking call's rule at every point it evaluates (moptimizer_hip.h,
    MOPT_KERNEL_AUTO): under AUTO, as under MOPT_KERNEL_LITERAL, its first iterate — H, b at x0 from
    transforms the host formed, one damped solve — is the CPU loop's to rounding; MOMENTS_ALWAYS
    keeps the moments and is 1e5 times further away, still 1e-8.  Later iterates carry the device's
    own sincos (module docstring) and agree to FD_ITERATE_TOL under either; same pose at the end."""
    mo = hip_lib
    rng = np.random.default_rng(3)
    n = 100_000
    src = rng.random((n, 3)) * 10.0
    x_true = np.array([0.02, -0.03, 0.01, 0.004, -0.006, 0.005])
    T = oracle.se3_from_x(x_true)
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3))
    x0 = 0.5 * x_true
    cost = mo.Point2PointCost(src, tgt)
    xr1, _, _ = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.NUMERIC_DYN, layout=ob.LAYOUT_ROW_MAJOR,
                                    max_iter=1)
    kept = {}
    for variant, first_iterate_tol in ((mo.KERNEL_LITERAL, 1e-12), (mo.KERNEL_AUTO, 1e-12),
                                       (mo.KERNEL_MOMENTS, 1e-12), (mo.KERNEL_MOMENTS_ALWAYS, 1e-7)):
        cost.set_kernel_variant(variant)
        x1, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=1)
        assert rep["iterations"] == 1 and np.abs(x1 - xr1).max() < first_iterate_tol, (variant, x1 - xr1)
        if variant == mo.KERNEL_MOMENTS_ALWAYS:
            assert np.abs(x1 - xr1).max() > 1e-10   # (the deviation is there: the header does not overstate)
        for k in (2, 3, 15):
            x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=k)
            xr, status, iters = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.NUMERIC_DYN,
                                                    layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
            assert abs(rep["iterations"] - iters) <= 2, (k, rep, iters)
            assert np.abs(x - xr).max() < FD_ITERATE_TOL, (k, x - xr)
            kept[(variant, k)] = (x, rep)
    # every point of the AUTO solves was small: each took the literal sweep, and the solves are the
    # LITERAL variant's bit for bit (the same two kernels ran)
    points, literal = cost.lm_choice_stats()
    assert points > 20 and literal == points, (points, literal)
    for k in (2, 3, 15):
        for variant in (mo.KERNEL_AUTO, mo.KERNEL_MOMENTS):
            assert np.array_equal(kept[(variant, k)][0], kept[(mo.KERNEL_LITERAL, k)][0]), (variant, k)
            assert kept[(variant, k)][1] == kept[(mo.KERNEL_LITERAL, k)][1], (variant, k)
    # the blocking call applies the rule per x: AUTO at this x0 is the reference's H, b to rounding
    cost.set_kernel_variant(mo.KERNEL_AUTO)
    H, b, s = cost.linearize(x0, mo.JAC_NUMERIC)
    Hr, br, sr = oracle.p2p_linearize(src, tgt, x0, cost_class=ob.NUMERIC_DYN)
    assert np.abs(H - Hr).max() <= 1e-12 * np.abs(Hr).max() and np.abs(b - br).max() <= 1e-12 * np.abs(br).max()
    cost.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_forward_difference_sweep_is_chosen_point_by_point(hip_lib, oracle, dtype):
    """Under AUTO the device-resident loop sweeps with a kernel that holds both forward-difference forms and
    the step kernel names one per point (sweep.hpp kLmGateMoments; linearization.h:78-105).  Where no
    evaluated point has a small parameter the solve is MOMENTS_ALWAYS's bit for bit, where every point
    has one it is LITERAL's (previous test), and on a registration that starts at x = 0 (fixed step:
    moments) and ends at a pose with one small component both kinds run in one solve — which still
    lands on the CPU loop's pose."""
    mo = hip_lib
    rng = np.random.default_rng(12)
    n = 60_000
    src = rng.random((n, 3)) * 10.0
    fp64 = dtype == np.float64

    def problem(x_true):
        T = oracle.se3_from_x(np.asarray(x_true, dtype=np.float64))
        tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3))
        return mo.Point2PointCost(src.astype(dtype), tgt.astype(dtype), dtype=dtype), tgt

    # (a) every parameter stays large: no literal point, the moments solve exactly
    x_true = np.array([1.5, -1.3, 1.2, 0.4, -0.5, 0.3])
    cost, _ = problem(x_true)
    x0 = (x_true + np.array([0.2, -0.2, 0.2, 0.05, 0.05, -0.05])).astype(dtype)
    got = {}
    for variant in (mo.KERNEL_AUTO, mo.KERNEL_MOMENTS_ALWAYS):
        cost.set_kernel_variant(variant)
        got[variant] = [mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=k) for k in (1, 3, 15)]
    points, literal = cost.lm_choice_stats()
    assert points == sum(r["sweeps"] for _, r in got[mo.KERNEL_AUTO]) and literal == 0, (points, literal)
    for (xa, ra), (xm, rm) in zip(got[mo.KERNEL_AUTO], got[mo.KERNEL_MOMENTS_ALWAYS]):
        assert ra == rm and np.array_equal(xa, xm), (ra, rm, xa - xm)
    assert np.abs(got[mo.KERNEL_AUTO][2][0] - x_true).max() < (1e-3 if fp64 else 5e-3)
    cost.close()

    # (b) from x = 0 to a pose with one small component: moments first, literal near the end
    x_true = np.array([0.5, -0.3, 0.2, 0.01, -0.2, 0.3])
    cost, tgt = problem(x_true)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6, dtype=dtype))
    points, literal = cost.lm_choice_stats()
    assert points == rep["sweeps"] and 0 < literal < points, (points, literal, rep)
    if fp64:
        xr, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN,
                                                layout=ob.LAYOUT_ROW_MAJOR)
        assert abs(rep["iterations"] - iters) <= 2 and np.abs(x - xr).max() < FD_ITERATE_TOL, (rep, iters, x - xr)
    else:
        assert np.abs(x - x_true).max() < 5e-3, x - x_true
    # the switch keeps the loop's other paths: MOPT_LM_PER_ITERATE is read once per process, so the
    # moments-at-every-point form is exercised through MOMENTS_ALWAYS above
    cost.close()


def test_same_answer_as_the_host_loop_over_the_hip_cost(hip_lib):
    mo = hip_lib
    src, tgt = ds.synthetic_pair(50_000, seed=8, noise=0.05)
    cost = mo.Point2PointCost(src, tgt)
    cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 30.0)
    # every covariance form has its own resident sweep (the forward-difference ones keep the
    # perturbed rotations in registers, in registers and LDS, or in LDS by form and precision)
    covs = (np.diag([1.0, 0.5, 2.0]),
            np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]]),
            np.array([[2.0, 0.7, -0.1], [0.3, 0.5, 0.9], [-0.4, 0.2, 1.5]]))
    for cov in covs:
        cost.set_covariance(cov)
        for jac in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
            for variant in (mo.KERNEL_MOMENTS, mo.KERNEL_LITERAL):
                cost.set_kernel_variant(variant)
                xh, sh, ih = host_lm(cost, jac, np.zeros(6))
                x, rep = mo.capi.lm_minimize([cost], [jac], np.zeros(6))
                assert rep["status"] == sh, (cov, jac, variant, rep, sh)
                if jac == mo.JAC_ANALYTIC:
                    assert rep["iterations"] == ih, (cov, variant, rep, ih)
                    assert np.abs(x - xh).max() < 1e-9 * max(1.0, np.abs(xh).max())
                else:
                    assert abs(rep["iterations"] - ih) <= 1, (cov, variant, rep, ih)
                    assert np.abs(x - xh).max() < FD_ITERATE_TOL * max(1.0, np.abs(xh).max())
    cost.close()


def test_every_optimization_status(hip_lib):
    mo = hip_lib
    # CONVERGED by vanishing cost: exact correspondences
    src, tgt = ds.synthetic_pair(4000, seed=2, noise=0.0)
    cost = mo.Point2PointCost(src, tgt)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=30)
    assert rep["status"] == CONVERGED and np.abs(x - ds.FIXTURE_X).max() < 1e-8
    # MAXIMUM_ITERATIONS_REACHED
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=2)
    assert rep["status"] == MAX_ITERATIONS and rep["iterations"] == 2
    # zero iterations: the loop body never runs, x untouched, no sweep
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.full(6, 0.25), max_iterations=0)
    assert rep["status"] == MAX_ITERATIONS and rep["sweeps"] == 0 and np.all(x == 0.25)
    cost.close()
    # SMALL_DELTA: noisy data, started at the optimum found by a first run
    src, tgt = ds.synthetic_pair(4000, seed=3, noise=0.05)
    cost = mo.Point2PointCost(src, tgt)
    x1, rep1 = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=50)
    xh, sh, ih = host_lm(cost, mo.JAC_NUMERIC, np.zeros(6), max_iter=50)
    assert rep1["status"] == sh == SMALL_DELTA and abs(rep1["iterations"] - ih) <= 1
    assert np.abs(x1 - xh).max() < FD_ITERATE_TOL * np.abs(xh).max()
    xa, repa = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], np.zeros(6), max_iterations=50)
    xh, sh, ih = host_lm(cost, mo.JAC_ANALYTIC, np.zeros(6), max_iter=50)
    assert (repa["status"], repa["iterations"]) == (sh, ih)
    assert np.abs(xa - xh).max() < 1e-9 * np.abs(xh).max()
    cost.close()
    # NUMERIC_ERROR: a NaN source coordinate makes every trial cost NaN (:88-91)
    bad = src.copy()
    bad[17, 1] = np.nan
    cost = mo.Point2PointCost(bad, tgt)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], np.zeros(6))
    assert rep["status"] == NUMERIC_ERROR
    cost.close()


def test_camera_calibration_two_costs_with_robust_loss(hip_lib, oracle):
    """tst/camera_calibration.cpp / multiple_objectives.cpp shape: two reprojection costs summed."""
    mo = hip_lib
    pts, pix = ds.synthetic_camera(20_000, seed=17)
    split = 8000
    costs = [mo.ReprojectionCost(pts[:split], pix[:split]), mo.ReprojectionCost(pts[split:], pix[split:])]
    for c in costs:
        c.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
    x, rep = mo.capi.lm_minimize(costs, [mo.JAC_NUMERIC] * 2, np.zeros(6), max_iterations=25)
    xr, status, iters = oracle.camera_minimize(pts, pix, [split, len(pts) - split], np.zeros(6),
                                               max_iter=25, loss_kind=1, loss_param=100.0)
    assert rep["status"] == status and abs(rep["iterations"] - iters) <= 1, (rep, status, iters)
    assert np.abs(x - xr).max() < 1e-6, (x, xr)  # forward differences only (BaseModel), and pixel
    # rounding leaves a flat valley: the pose is determined to ~1e-7
    for c in costs:
        c.close()


def test_float32_cost(hip_lib):
    mo = hip_lib
    src, tgt = ds.synthetic_pair(20_000, seed=4, noise=0.01, dtype=np.float32)
    cost = mo.Point2PointCost(src, tgt, dtype=np.float32)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], np.zeros(6, dtype=np.float32),
                                 max_iterations=30)
    assert x.dtype == np.float32
    assert np.abs(x - ds.FIXTURE_X).max() < 5e-3
    cost.close()


def test_float32_nearly_rank_deficient_normal_equations(hip_lib):
    """fp32 with a Hessian close to singular (the rational model observed over a narrow range of t:
    its two Jacobian columns are nearly parallel, the second pivot of the damped system cancels to a
    few 1e-7 of its diagonal entry): the device's unpivoted solve must hand such a system to the
    pivoted one — its test for that scales with the type's epsilon (64 eps; the fp64 constant 1e-10
    would never fire in fp32) — and the loop must still descend as the host loop over the same cost
    does (whose solve is numpy's, in fp64, on the same fp32 sums)."""
    mo = hip_lib
    rng = np.random.default_rng(8)
    n = 2_000   # (the loop's first damping is 1e-9 max diag H, relative to diag H: it grows with n)
    t = (2.0 + 1e-3 * rng.random(n)).astype(np.float32)
    y = (0.36 * t / (0.56 + t) + rng.normal(0, 1e-3, n)).astype(np.float32)
    cost = mo.ScalarModelCost(mo.capi.MODEL_RATIONAL, t, y, dtype=np.float32)
    x0 = np.array([0.5, 0.9], dtype=np.float32)
    H, b, y0 = cost.linearize(x0, mo.JAC_ANALYTIC)
    # the regime meant, from the data in fp64 (in the fp32 sums themselves the second pivot is at the
    # level of their rounding: which solve the device takes is then a matter of that noise — either
    # must do)
    q = x0[1].astype(np.float64) + t.astype(np.float64)
    J = np.stack([-t / q, x0[0] * t / (q * q)], axis=1)
    A = J.T @ J
    A = A + 1e-9 * np.diag(A).max() * np.diag(np.diag(A))
    pivot2 = A[1, 1] - A[1, 0] * A[0, 1] / A[0, 0]
    assert 0 < pivot2 / A[1, 1] < 64 * np.finfo(np.float32).eps, pivot2 / A[1, 1]
    xd, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], x0)
    xh, status, iters = host_lm(cost, mo.JAC_ANALYTIC, x0)
    cd, ch = cost.compute_cost(xd.astype(np.float32)), cost.compute_cost(xh.astype(np.float32))
    assert np.isfinite(xd).all() and rep["status"] in (CONVERGED, MAX_ITERATIONS, SMALL_DELTA)
    assert cd < 0.05 * y0 and cd <= ch * 1.05 + 1e-6, (cd, ch, y0, rep)


def test_blocking_calls_still_work_after_a_device_resident_solve(hip_lib, oracle):
    mo = hip_lib
    src, tgt = ds.synthetic_pair(30_000, seed=6, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    before = cost.linearize(ds.X_GENERIC, mo.JAC_NUMERIC)
    mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6))
    after = cost.linearize(ds.X_GENERIC, mo.JAC_NUMERIC)
    for a, b in zip(before, after):
        assert np.asarray(a).tobytes() == np.asarray(b).tobytes()
    cost.set_loss(mo.LOSS_GEMAN_MCCLURE, 10.0)  # the resident constants are re-uploaded
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6))
    xh, sh, ih = host_lm(cost, mo.JAC_NUMERIC, np.zeros(6))
    assert rep["status"] == sh and abs(rep["iterations"] - ih) <= 1
    assert np.abs(x - xh).max() < FD_ITERATE_TOL * 11
    cost.close()


def _exp(w):
    th = np.linalg.norm(w)
    a = w / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def test_manifold_update_on_the_device(hip_lib, oracle):
    """mopt_lm_options.manifold with MOPT_JAC_ANALYTIC_LEFT costs: the iterates of the CPU loop with
    setManifoldUpdate over the oracle's left-perturbation model, from a start 2.5 rad off."""
    mo = hip_lib
    src, tgt = ds.synthetic_pair(20_000, seed=12, noise=0.01)
    R0 = _exp(np.array([0.0, 0.0, 2.5])) @ ds.fixture_rotation()
    th = np.arccos((np.trace(R0) - 1) / 2)
    w0 = th / (2 * np.sin(th)) * np.array([R0[2, 1] - R0[1, 2], R0[0, 2] - R0[2, 0], R0[1, 0] - R0[0, 1]])
    x0 = np.concatenate([ds.FIXTURE_T + 1.0, w0])
    cost = mo.Point2PointCost(src, tgt)
    for variant in (mo.KERNEL_MOMENTS, mo.KERNEL_LITERAL):
        cost.set_kernel_variant(variant)
        for k in (1, 2, 4, 60):
            x, rep = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC_LEFT], x0, max_iterations=k,
                                         manifold=True)
            xr, status, iters = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                                    layout=ob.LAYOUT_LEFT | ob.MANIFOLD_UPDATE,
                                                    max_iter=k)
            if k < 60:
                assert (rep["status"], rep["iterations"]) == (status, iters), (k, rep, status, iters)
            else:
                assert rep["status"] == status, (rep, status, iters)
                assert abs(rep["iterations"] - iters) <= 1 or \
                    abs(rep["cost"] - oracle.p2p_cost(src, tgt, xr)) <= 1e-9 * rep["cost"], (rep, status, iters)
            assert np.abs(x - xr).max() < 1e-9 * max(1.0, np.abs(xr).max()), (k, x, xr)
    assert np.abs(_exp(x[3:]) - ds.fixture_rotation()).max() < 1e-3
    # composed on the right (manifold = 2, MOPT_JAC_ANALYTIC_RIGHT): the reference's own sketches
    for variant in (mo.KERNEL_MOMENTS, mo.KERNEL_LITERAL):
        cost.set_kernel_variant(variant)
        for k in (1, 2, 4, 60):
            xq, repq = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC_RIGHT], x0, max_iterations=k,
                                           manifold="right")
            xr, status, iters = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.ANALYTIC_DYN,
                                                    layout=ob.LAYOUT_RIGHT | ob.MANIFOLD_UPDATE_RIGHT,
                                                    max_iter=k)
            if k < 60:
                assert (repq["status"], repq["iterations"]) == (status, iters), (k, repq, status, iters)
            else:
                # the iterates agree to 1e-13 for six iterations; then both loops idle at the minimum
                # until rho < 0 meets a small delta — decided by the 16th digit of two costs: the same
                # number of iterations, or the same minimum
                assert repq["status"] == status, (repq, status, iters)
                assert abs(repq["iterations"] - iters) <= 1 or \
                    abs(repq["cost"] - oracle.p2p_cost(src, tgt, xr)) <= 1e-9 * repq["cost"], (repq, status, iters)
            assert np.abs(xq - xr).max() < 1e-9 * max(1.0, np.abs(xr).max()), (k, xq, xr)
    assert np.abs(_exp(xq[3:]) - ds.fixture_rotation()).max() < 1e-3
    # the Euclidean update from the same start needs more than twice the outer iterations
    xe, repe = mo.capi.lm_minimize([cost], [mo.JAC_ANALYTIC], x0, max_iterations=200)
    assert repe["iterations"] > 2 * rep["iterations"], (repe, rep)
    cost.close()


def test_run_time_compiled_models_under_the_device_loop(hip_lib, oracle):
    """User-written models (mopt_jit_model_create) in mopt_lm_minimize: the curve fit of
    tst/curve_fitting.cpp:101-147 written as source reaches the reference's known answer, and the
    path's own point2point model written as source gives the iterates of the built-in one."""
    mo = hip_lib
    from tests.test_gpu_parity import P2P_JIT_JACOBIAN, P2P_JIT_RESIDUAL, P2P_JIT_SETUP
    curve = np.load(ds.GOLDEN + "/p2p_1k_golden.npz")  # (only to make sure the fixture dir exists)
    del curve
    rng = np.random.default_rng(0)
    t = np.linspace(0.0, 4.95, 67)
    y = np.exp(0.3 * t + 0.1) + rng.normal(0, 0.2, t.shape)
    jit = mo.JitModelCost(2, 1, "r[0] = d[1] - exp(x[0] * d[0] + x[1]);",
                          "const S e = exp(x[0] * d[0] + x[1]); J[0] = -d[0] * e; J[1] = -e;",
                          planes=np.stack([t, y]))
    builtin = mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, y)
    for jac in (mo.JAC_NUMERIC, mo.JAC_ANALYTIC):
        xj, rj = mo.capi.lm_minimize([jit], [jac], np.zeros(2), max_iterations=50)
        xh, sh, ih = host_lm_n(jit, jac, np.zeros(2), 50)
        # same status and end point; the same number of outer iterations, or — both loops idle at
        # the minimum until rho < 0 meets a small delta, which the last bits of the cost decide — the
        # same minimum
        assert rj["status"] == sh, (rj, sh, ih)
        assert abs(rj["iterations"] - ih) <= 1 or \
            abs(rj["cost"] - jit.compute_cost(xh)) <= 1e-9 * rj["cost"], (rj, sh, ih)
        assert np.abs(xj - xh).max() < 1e-6
        assert np.abs(xj - np.array([0.3, 0.1])).max() < 0.1  # near the generating parameters
    xb, rb = mo.capi.lm_minimize([builtin], [mo.JAC_NUMERIC], np.zeros(2), max_iterations=50)
    xj, rj = mo.capi.lm_minimize([jit], [mo.JAC_NUMERIC], np.zeros(2), max_iterations=50)
    assert np.abs(xb - xj).max() < 1e-6
    # point2point as source, analytic Jacobian: same iterates as the hand-written sweep
    src, tgt = ds.synthetic_pair(30_000, seed=3, noise=0.01)
    planes = np.concatenate([src.T, tgt.T])
    p2p_jit = mo.JitModelCost(6, 3, P2P_JIT_RESIDUAL, P2P_JIT_JACOBIAN, planes=planes, n_aux=12,
                              setup_body=P2P_JIT_SETUP)
    p2p = mo.Point2PointCost(src, tgt)
    for k in (1, 3, 15):
        xa, ra = mo.capi.lm_minimize([p2p_jit], [mo.JAC_ANALYTIC], np.zeros(6), max_iterations=k)
        xb, rb = mo.capi.lm_minimize([p2p], [mo.JAC_ANALYTIC], np.zeros(6), max_iterations=k)
        assert (ra["status"], ra["iterations"]) == (rb["status"], rb["iterations"])
        assert np.abs(xa - xb).max() < 1e-9 * 11
    for c in (jit, builtin, p2p_jit, p2p):
        c.close()


def host_lm_n(cost, jac_mode, x0, max_iter):
    """host_lm for any parameter count."""
    return host_lm(cost, jac_mode, x0, max_iter=max_iter)


def test_icp_cost_with_the_search_inside_the_device_loop(hip_lib):
    """An ICP cost under mopt_lm_minimize: its update(x) — the nearest-neighbour search — runs on
    the device at the top of every outer iteration, as the reference's loop calls it
    (levenberg_marquadt_dyn.cpp:54).  Same status, iteration count and pose as the host loop that
    calls update / linearize / computeCost through the boundary."""
    mo = hip_lib
    rng = np.random.default_rng(4)
    tgt = rng.random((40_000, 3)) * 20.0
    x_true = np.array([0.15, -0.1, 0.2, 0.02, -0.03, 0.025])
    th = np.linalg.norm(x_true[3:])
    a = x_true[3:] / th
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    # sources = targets moved by the inverse pose (+ noise), shuffled: registration must undo it
    src = (tgt - x_true[:3]) @ R + rng.normal(0, 0.002, tgt.shape)
    src = src[rng.permutation(len(src))][:30_000]
    # iterate for iterate the host loop's answer.  The device solves the damped system without
    # pivoting where it is positive definite (lm_device.hpp solveDampedPositive): the same step to
    # eps * cond(H), not the same bits, so the poses agree to 1e-13 all the way and only the
    # noise-level stop at the very end — rho < 0 with |delta| < sqrt eps, decided by the last bits of
    # two costs that differ in them — may come at a different iteration (k = 27: the pose has been
    # stationary for iterations by then).
    for k in (3, 12, 27):
        dev = mo.IcpCost(src, tgt, max_distance=0.6)
        ref = mo.IcpCost(src, tgt, max_distance=0.6)
        xd, rep = mo.capi.lm_minimize([dev], [mo.JAC_ANALYTIC], np.zeros(6), max_iterations=k)
        xh, sh, ih = host_lm(ref, mo.JAC_ANALYTIC, np.zeros(6), max_iter=k)
        if k < 27:
            assert (rep["status"], rep["iterations"]) == (sh, ih), (k, rep, sh, ih)
            assert rep["sweeps"] == 2 * k  # search + linearization, then one accepted trial, per iteration
        else:
            assert rep["status"] in (MAX_ITERATIONS, SMALL_DELTA, CONVERGED), rep
            assert 20 <= rep["iterations"] <= k, rep
        assert np.abs(xd - xh).max() < 1e-9, (k, xd, xh)
        # the correspondences the two loops ended with are the same
        md, mr = dev.matches(), ref.matches()
        assert np.array_equal(np.isnan(md), np.isnan(mr))
        assert np.nanmax(np.abs(md - mr)) == 0.0
        dev.close()
        ref.close()
    for jac in (mo.JAC_ANALYTIC, mo.JAC_ANALYTIC_LEFT):
        dev = mo.IcpCost(src, tgt, max_distance=0.6)
        xd, rep = mo.capi.lm_minimize([dev], [jac], np.zeros(6), max_iterations=60,
                                      manifold=(jac == mo.JAC_ANALYTIC_LEFT))
        assert rep["status"] in (CONVERGED, SMALL_DELTA, MAX_ITERATIONS)
        assert np.abs(xd - x_true).max() < 2e-3, (jac, xd)
        # ... and the blocking calls continue from that state
        n_matched = dev.update(xd)
        assert n_matched > 0.95 * len(src)
        dev.close()


def host_lm_sum(costs, jac_modes, x0, max_iter=15, lm_iter=3):
    """The same loop over several HIP costs: H, b and the cost summed in cost order (:48-60, :86)."""
    class Sum:
        def linearize(self, x, _):
            H, b, y = 0.0, 0.0, 0.0
            for c, jac in zip(costs, jac_modes):
                Hc, bc, yc = c.linearize(x, jac)
                H, b, y = H + Hc, b + bc, y + yc
            return H, b, y

        def compute_cost(self, x):
            return sum(c.compute_cost(x) for c in costs)
    return host_lm(Sum(), None, x0, max_iter, lm_iter)


def test_several_costs_in_every_combination_of_sweep_kinds(hip_lib):
    """Costs whose sweeps leave rows of one form are reduced by ONE finalize kernel (and swept by one
    launch where a kernel for that exists: reprojection); rows of moments and mixed forms keep one
    finalize per cost.  Every form against the host loop summing the same costs through the blocking
    boundary; then each cost again on its own (its resident constants point at its own rows again)."""
    mo = hip_lib
    rng = np.random.default_rng(5)
    # (a) three reprojection costs: one sweep launch + one finalize per point
    pts, pix = ds.synthetic_camera(30_000, seed=3)
    cuts = [0, 7000, 18_000, 30_000]
    cams = [mo.ReprojectionCost(pts[a:b], pix[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for c in cams:
        c.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)
    x, rep = mo.capi.lm_minimize(cams, [mo.JAC_NUMERIC] * 3, np.zeros(6), max_iterations=25)
    xh, sh, ih = host_lm_sum(cams, [mo.JAC_NUMERIC] * 3, np.zeros(6), max_iter=25)
    # same status and pose; the same number of outer iterations, or — both loops idle at the minimum
    # until rho < 0 meets a small delta, which the last bits of the cost decide — the same minimum
    assert rep["status"] == sh, (rep, sh, ih)
    assert abs(rep["iterations"] - ih) <= 1 or \
        abs(rep["cost"] - sum(c.compute_cost(xh) for c in cams)) <= 1e-9 * rep["cost"], (rep, sh, ih)
    assert np.abs(x - xh).max() < 1e-6, (x, xh)
    x1, rep1 = mo.capi.lm_minimize(cams[:1], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=25)
    xh1, sh1, _ = host_lm(cams[0], mo.JAC_NUMERIC, np.zeros(6), max_iter=25)
    assert rep1["status"] == sh1 and np.abs(x1 - xh1).max() < 1e-6, (x1, xh1)
    for c in cams:
        c.close()

    # (b) two exp-curve costs (tst/multiple_objectives.cpp:81-98): one finalize, a sweep each
    t = np.linspace(0.0, 5.0, 4000)
    y = np.exp(0.3 * t + 0.1) + 0.01 * rng.standard_normal(t.size)
    curves = [mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t[:1500], y[:1500]),
              mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t[1500:], y[1500:])]
    for jac in (mo.JAC_NUMERIC,):  # the model has no f_df (numeric only, like the reference's)
        x, rep = mo.capi.lm_minimize(curves, [jac] * 2, np.zeros(2), max_iterations=50)
        xh, sh, ih = host_lm_sum(curves, [jac] * 2, np.zeros(2), max_iter=50)
        # both loops end on the noise-level test (rho < 0 with |delta| < sqrt eps) at the same point; at
        # which of the last few iterations that test first fires is decided by rounding
        assert rep["status"] == sh and abs(rep["iterations"] - ih) <= 3, (jac, rep, sh, ih)
        assert np.abs(x - xh).max() < 1e-8, (jac, x, xh)
    for c in curves:
        c.close()

    # (c) point2point costs: literal + literal (one finalize), moments + literal and
    #     moments + moments (a finalize each: rows of moments are contracted by their own kernel)
    src, tgt = ds.synthetic_pair(60_000, seed=21, noise=0.02)
    halves = [mo.Point2PointCost(src[:25_000], tgt[:25_000]), mo.Point2PointCost(src[25_000:], tgt[25_000:])]
    whole = mo.Point2PointCost(src, tgt)
    xw, repw = mo.capi.lm_minimize([whole], [mo.JAC_ANALYTIC], np.zeros(6))
    for variants in ((mo.KERNEL_LITERAL, mo.KERNEL_LITERAL), (mo.KERNEL_MOMENTS, mo.KERNEL_LITERAL),
                     (mo.KERNEL_MOMENTS, mo.KERNEL_MOMENTS)):
        for c, v in zip(halves, variants):
            c.set_kernel_variant(v)
        x, rep = mo.capi.lm_minimize(halves, [mo.JAC_ANALYTIC] * 2, np.zeros(6))
        xh, sh, ih = host_lm_sum(halves, [mo.JAC_ANALYTIC] * 2, np.zeros(6))
        assert rep["status"] == sh and rep["iterations"] == ih, (variants, rep, sh, ih)
        assert np.abs(x - xh).max() < 1e-9, (variants, x, xh)
        # two halves of a cloud = the whole cloud
        assert rep["status"] == repw["status"] and np.abs(x - xw).max() < 1e-9, (variants, x, xw)
    # a cost that shared a finalize goes back to its own rows for the blocking calls and alone
    H, b, y0 = halves[0].linearize(xw, mo.JAC_ANALYTIC)
    x0, rep0 = mo.capi.lm_minimize(halves[:1], [mo.JAC_ANALYTIC], np.zeros(6))
    xh0, sh0, ih0 = host_lm(halves[0], mo.JAC_ANALYTIC, np.zeros(6))
    assert rep0["status"] == sh0 and np.abs(x0 - xh0).max() < 1e-9
    H2, b2, y2 = halves[0].linearize(xw, mo.JAC_ANALYTIC)
    assert np.array_equal(H, H2) and np.array_equal(b, b2) and y0 == y2

    # (d) round 3: costs of one kind, mode and covariance form are also SWEPT by one launch — literal
    #     point2point (analytic and forward differences, with a covariance), the built-in scalar
    #     models of (b), and run-time compiled models created from the same source
    cov = np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]])
    for c in halves + [whole]:
        c.set_kernel_variant(mo.KERNEL_LITERAL)
        c.set_covariance(cov)
    for jac in (mo.JAC_NUMERIC, mo.JAC_ANALYTIC_RIGHT):
        manifold = "right" if jac == mo.JAC_ANALYTIC_RIGHT else False
        xw2, repw2 = mo.capi.lm_minimize([whole], [jac], np.zeros(6), manifold=manifold)
        x, rep = mo.capi.lm_minimize(halves, [jac] * 2, np.zeros(6), manifold=manifold)
        # (two halves summed and the whole differ in the last bits of every sum: the pose is the
        # same, the iteration at which the noise-level stop fires need not be)
        assert rep["status"] in (CONVERGED, SMALL_DELTA, MAX_ITERATIONS), rep
        assert np.abs(x - xw2).max() < 1e-7, (jac, x, xw2, rep, repw2)
    # a pair in different modes shares the finalize but not the launch: still the same answer
    x, rep = mo.capi.lm_minimize(halves, [mo.JAC_ANALYTIC, mo.JAC_NUMERIC], np.zeros(6))
    xh, sh, ih = host_lm_sum(halves, [mo.JAC_ANALYTIC, mo.JAC_NUMERIC], np.zeros(6))
    assert rep["status"] == sh and np.abs(x - xh).max() < 1e-7, (x, xh)
    for c in halves + [whole]:
        c.close()
    body = "r[0] = d[1] - exp(x[0] * d[0] + x[1]);"
    jac_body = "const S e = exp(x[0] * d[0] + x[1]); J[0] = -d[0] * e; J[1] = -e;"
    parts = [mo.JitModelCost(2, 1, body, jac_body, planes=np.stack([t[a:b], y[a:b]]))
             for a, b in ((0, 1500), (1500, 2600), (2600, 4000))]
    one = mo.JitModelCost(2, 1, body, jac_body, planes=np.stack([t, y]))
    for jac in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
        xo, repo = mo.capi.lm_minimize([one], [jac], np.zeros(2), max_iterations=50)
        x, rep = mo.capi.lm_minimize(parts, [jac] * 3, np.zeros(2), max_iterations=50)
        assert rep["status"] == repo["status"], (rep, repo)
        assert np.abs(x - xo).max() < 1e-8, (jac, x, xo)
        assert np.abs(x - np.array([0.3, 0.1])).max() < 1e-2
    for c in parts + [one]:
        c.close()


def test_four_wide_costs_each_with_full_rows(hip_lib):
    """Four wide run-time compiled costs (n = 14: rows of n^2 + n + 1 = 211 values) large enough for two
    workgroups per CU each (>= 32 x CUs elements): 8 x CUs rows of 211 doubles do not fit the last cost's
    partial-row buffer (16 x CUs rows of 96), so their rows must NOT be merged behind one another into
    it — the gate of round 3 counted rows only and would have sent the later costs' rows past its end
    (ADVICE r3).  The device-resident loop over all four must reach the host loop's minimum."""
    mo = hip_lib
    n, per_cost = 14, 8448          # 33 x 256 elements: a 256-CU part gets its 512 workgroups per cost
    rng = np.random.default_rng(17)
    x_true = rng.uniform(-1, 1, n)
    residual = """
  S v = 0;
  for (int k = 0; k < %d; ++k) v += cos(S(k + 1) * d[0]) * x[k];
  r[0] = v - d[1];""" % n
    jacobian = "for (int k = 0; k < %d; ++k) J[k] = cos(S(k + 1) * d[0]);" % n
    costs = []
    for _ in range(4):
        t = rng.uniform(0.0, 3.0, per_cost)
        basis = np.cos(np.arange(1, n + 1)[None, :] * t[:, None])
        y = basis @ x_true + 0.01 * rng.standard_normal(per_cost)
        costs.append(mo.JitModelCost(n, 1, residual, jacobian_body=jacobian, planes=np.stack([t, y])))
    x0 = x_true + 0.2 * rng.standard_normal(n)
    for jac in (mo.JAC_ANALYTIC, mo.JAC_NUMERIC):
        xd, rep = mo.capi.lm_minimize(costs, [jac] * 4, x0, max_iterations=30)
        xh, sh, ih = host_lm_sum(costs, [jac] * 4, x0, max_iter=30)
        assert np.isfinite(xd).all() and rep["status"] == sh, (jac, rep, sh, ih)
        assert np.abs(xd - xh).max() < 1e-6 and np.abs(xd - x_true).max() < 5e-3, (jac, xd, xh)
    # and every cost still answers on its own afterwards (nobody's rows or constants were overwritten)
    for c in costs:
        Hc, bc, sc = c.linearize(xd, mo.JAC_ANALYTIC)
        assert np.isfinite(Hc).all() and sc > 0
        c.close()



@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_small_problems_in_one_launch_take_the_same_iterates(hip_lib, oracle, dtype, monkeypatch):
    """One point2point cost of at most four tiles (the reference's own sizes) is minimised by ONE
    launch of one workgroup (sweep_kernels.hip p2pSolveSmallKernel: correspondences held in registers;
    sweep, contraction and LM step without leaving the kernel).  Against the launch-per-point loop
    (MOPT_LM_ONE_LAUNCH_TILES=0), whose sums it adds in another order: truncated after k outer
    iterations the same status, iteration count, sweeps and — to rounding — x and cost, for every
    Jacobian mode, covariance form, the robust loss and both manifold updates; run out, the same pose
    and cost (the noise-level stopping tests may fire an iteration apart, as between any two
    summation orders).  And the CPU loop's iterates as for any size."""
    mo = hip_lib
    per_tile = 512 if dtype == np.float64 else 1024
    eps = np.finfo(dtype).eps
    cov = np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]])
    cases = [(mo.JAC_ANALYTIC, 0, None, 0), (mo.JAC_NUMERIC, 0, None, 0), (mo.JAC_ANALYTIC_TST_LAYOUT, 0, None, 0),
             (mo.JAC_ANALYTIC_LEFT, 1, None, 0), (mo.JAC_ANALYTIC_RIGHT, 2, None, 0),
             (mo.JAC_ANALYTIC, 0, cov, 1), (mo.JAC_NUMERIC, 0, np.diag([0.5, 2.0, 3.0]), 1)]
    compared = 0
    for n in (1, 7, 1000, 4 * per_tile, 4 * per_tile + 1):
        src, tgt = ds.synthetic_pair(n, seed=40 + n % 7, noise=0.01)
        cost = mo.Point2PointCost(src.astype(dtype), tgt.astype(dtype), dtype=dtype)
        for jac, manifold, cv, loss in cases:
            numeric = jac == mo.JAC_NUMERIC
            cost.set_covariance(cv)
            cost.set_loss(loss, 100.0)
            for k in (1, 3, 15):
                got = {}
                for tiles in ("4", "0"):
                    monkeypatch.setenv("MOPT_LM_ONE_LAUNCH_TILES", tiles)
                    got[tiles] = mo.capi.lm_minimize([cost], [jac], np.zeros(6, dtype=dtype), max_iterations=k,
                                                     manifold=manifold)
                (xa, ra), (xb, rb) = got["4"], got["0"]
                what = (n, jac, k, ra, rb, xa, xb)
                if n > 4 * per_tile:  # five tiles: both runs took the launch-per-point loop
                    assert ra == rb and np.array_equal(xa, xb), what
                elif n < 3 or np.isnan(ra["cost"]) or np.isnan(rb["cost"]):
                    # fewer correspondences than constrain a pose: singular normal equations, where
                    # rounding decides the direction — both must end the same way
                    assert ra["status"] == rb["status"], what
                elif k < 15 and dtype == np.float64:
                    assert (ra["status"], ra["iterations"], ra["sweeps"]) == (rb["status"], rb["iterations"], rb["sweeps"]), what
                    tol = 1e-7 if numeric else 1e-10
                    assert np.abs(xa - xb).max() <= tol * max(1.0, np.abs(xb).max()), what
                    assert abs(ra["cost"] - rb["cost"]) <= tol * max(rb["cost"], 1e-12), what
                else:
                    tol = 1e-6 if dtype == np.float64 else 2e-3
                    assert np.abs(xa - xb).max() <= tol * max(1.0, np.abs(xb).max()), what
                    assert abs(ra["cost"] - rb["cost"]) <= 64 * eps * n + 1e-6 * rb["cost"], what
                compared += 1
        cost.close()
    assert compared == 5 * 7 * 3
    monkeypatch.delenv("MOPT_LM_ONE_LAUNCH_TILES")
    # and against the CPU loop (fp64, analytic: every iterate to 1e-9, same status and count)
    if dtype == np.float64:
        src, tgt = ds.synthetic_pair(1000, seed=5, noise=0.01)
        cost = mo.Point2PointCost(src, tgt)
        for k in (1, 2, 3, 5, 15):
            x, rep = mo.capi.lm_minimize([cost], [0], np.zeros(6), max_iterations=k)
            xr, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.ANALYTIC_DYN,
                                                    layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
            assert (rep["status"], rep["iterations"]) == (status, iters), (k, rep, status, iters)
            assert np.abs(x - xr).max() < 1e-9 * max(1.0, np.abs(xr).max()), (k, x, xr)
        cost.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_one_launch_solve_holds_both_forward_difference_forms(hip_lib, oracle, dtype, monkeypatch):
    """A small point2point cost that differentiates numerically under AUTO is still minimised by one launch
    of one workgroup: the kernel holds the literal forward-difference form beside the moments
    (p2pSolveSmallKernel<S, FD_COV>, fd_device.hpp over the packs held in registers) and the step chooses
    per point, as in the launch-per-point loop (MOPT_LM_ONE_LAUNCH_TILES=0), whose sums it adds in another
    order.  Nearly aligned clouds — every point small, every sweep literal — under each covariance form
    and the robust loss: the same status, iteration count, sweeps and, to forward-difference noise, x as
    that loop; the first iterate is the CPU loop's; and a solve in which both forms run."""
    mo = hip_lib
    fp64 = dtype == np.float64
    per_tile = 512 if fp64 else 1024
    rng = np.random.default_rng(21)
    x_small = np.array([0.02, -0.03, 0.01, 0.004, -0.006, 0.005])
    covs = (None, np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]]),
            np.array([[2.0, 0.7, -0.1], [0.3, 0.5, 0.9], [-0.4, 0.2, 1.5]]))
    differed = 0
    for n in (700, 4 * per_tile):
        src = rng.random((n, 3)) * 10.0
        T = oracle.se3_from_x(x_small)
        tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3))
        cost = mo.Point2PointCost(src.astype(dtype), tgt.astype(dtype), dtype=dtype)
        x0 = (0.5 * x_small).astype(dtype)
        for cov in covs:
            for loss in (0, 1):
                cost.set_covariance(cov)
                cost.set_loss(loss, 100.0)
                for k in (1, 3, 15):
                    p0, l0 = cost.lm_choice_stats()
                    monkeypatch.delenv("MOPT_LM_ONE_LAUNCH_TILES", raising=False)
                    xa, ra = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=k)
                    p1, l1 = cost.lm_choice_stats()
                    assert p1 - p0 == ra["sweeps"] and l1 - l0 == ra["sweeps"], (n, k, p1 - p0, l1 - l0, ra)
                    monkeypatch.setenv("MOPT_LM_ONE_LAUNCH_TILES", "0")
                    xb, rb = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=k)
                    what = (n, cov is not None, loss, k, ra, rb, xa - xb)
                    if k < 15 and fp64:
                        assert (ra["status"], ra["iterations"], ra["sweeps"]) == (rb["status"], rb["iterations"], rb["sweeps"]), what
                    # (run into the iteration limit, both loops wander at the noise floor of literal forward
                    # differences at |x_j| ~ 0.005 — eps |R p + t| / h_j ~ 4e-5 per Jacobian entry — each along
                    # its own summation order: same minimum, looser x)
                    tol = (FD_ITERATE_TOL if k < 15 else 1e-6) if fp64 else 5e-3
                    assert np.abs(xa - xb).max() <= tol, what
                    assert abs(ra["cost"] - rb["cost"]) <= (1e-6 if fp64 else 1e-2) * rb["cost"], what
                    differed += int(not np.array_equal(xa, xb))
        if fp64:  # the first iterate against the CPU loop (identity covariance, no loss)
            cost.set_covariance(None)
            cost.set_loss(0, 0.0)
            monkeypatch.delenv("MOPT_LM_ONE_LAUNCH_TILES", raising=False)
            x1, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=1)
            xr1, _, _ = oracle.p2p_minimize(src, tgt, 0.5 * x_small, cost_class=ob.NUMERIC_DYN,
                                            layout=ob.LAYOUT_ROW_MAJOR, max_iter=1)
            assert rep["iterations"] == 1 and np.abs(x1 - xr1).max() < 1e-10, x1 - xr1
        cost.close()
    # the two loops are different kernels: identical bits throughout would mean the one launch never ran
    assert differed > 0
    # from x = 0 (fixed step: moments) to a pose with one small component (literal): both forms in one launch
    monkeypatch.delenv("MOPT_LM_ONE_LAUNCH_TILES", raising=False)
    x_true = np.array([0.5, -0.3, 0.2, 0.01, -0.2, 0.3])
    src = rng.random((1500, 3)) * 10.0
    T = oracle.se3_from_x(x_true)
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (1500, 3))
    cost = mo.Point2PointCost(src.astype(dtype), tgt.astype(dtype), dtype=dtype)
    x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6, dtype=dtype))
    points, literal = cost.lm_choice_stats()
    assert points == rep["sweeps"] and 0 < literal < points, (points, literal, rep)
    assert np.abs(x - x_true).max() < (2e-3 if fp64 else 1e-2), x - x_true
    cost.close()


def test_metric_size_10M_device_loop_chooses_per_point(hip_lib, oracle):
    """The per-point choice of the forward-difference sweep at the metric's own size (10 M correspondences,
    BASELINE.json), through properties that need no CPU sweep: between nearly aligned clouds every point is
    small and the AUTO solve is the LITERAL solve bit for bit (the literal form of the kernel that holds both
    ran at every point); between clouds a generic pose apart no point is small and it is the MOMENTS_ALWAYS
    solve bit for bit; either lands on the pose the data were made with."""
    mo = hip_lib
    import torch
    n = 10_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    src = torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float64) * 10.0
    for x_true, other, all_literal in ((np.array([0.02, -0.03, 0.01, 0.004, -0.006, 0.005]), mo.KERNEL_LITERAL, True),
                                       (np.array([1.5, -1.3, 1.2, 0.4, -0.5, 0.3]), mo.KERNEL_MOMENTS_ALWAYS, False)):
        T = torch.tensor(oracle.se3_from_x(x_true), device="cuda", dtype=torch.float64)
        tgt = (src @ T[:3, :3].T + T[:3, 3] + 0.01 * torch.randn((n, 3), generator=g, device="cuda",
                                                                   dtype=torch.float64)).contiguous()
        torch.cuda.synchronize()
        cost = mo.Point2PointCost(src.data_ptr(), tgt.data_ptr(), device_ptrs=True, count=n)
        x0 = x_true * (0.5 if all_literal else 1.0) + (0.0 if all_literal else 0.05)
        got = {}
        for variant in (mo.KERNEL_AUTO, other):
            cost.set_kernel_variant(variant)
            got[variant] = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=4)
        (xa, ra), (xo, ro) = got[mo.KERNEL_AUTO], got[other]
        points, literal = cost.lm_choice_stats()
        assert points == ra["sweeps"] and literal == (points if all_literal else 0), (points, literal, ra)
        assert ra == ro and np.array_equal(xa, xo), (ra, ro, xa - xo)
        assert np.abs(xa - x_true).max() < 5e-2, xa - x_true
        # (four damped iterations: near, not at, the pose; run out, at it)
        cost.set_kernel_variant(mo.KERNEL_AUTO)
        xf, rf = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=30)
        assert np.abs(xf - x_true).max() < 2e-3, (xf - x_true, rf)
        cost.close()
        del tgt
    torch.cuda.empty_cache()

