"""Pre-queued sweeps (mopt_cost_set_prequeue): the next blocking linearization's sweep + finalize
queued before its x is known and released by a word the host stores.  Same kernels on the same
inputs, so every number must equal the ordinary call's bit for bit; and every way a queued pair can
become useless — state changed, another kind of call, the caller went away — must end in an
ordinary launch, never in a wait."""
import time

import numpy as np
import pytest

from tests import datasets as ds

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_prequeued_sweeps_return_the_ordinary_numbers(hip_lib):
    mo = hip_lib
    rng = np.random.default_rng(11)
    src, tgt = ds.synthetic_pair(70_003, seed=5, noise=0.02)
    plain = mo.Point2PointCost(src, tgt)
    gated = mo.Point2PointCost(src, tgt)
    gated.set_prequeue(True)
    for c in (plain, gated):
        c.set_speculation(False)
    xs = [ds.X_GENERIC + 0.004 * rng.standard_normal(6) for _ in range(40)]  # every |x_j| > 0.08
    modes = (mo.JAC_ANALYTIC, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_LEFT, mo.JAC_ANALYTIC_RIGHT,
             mo.JAC_ANALYTIC_TST_LAYOUT)
    for k, x in enumerate(xs):
        m = modes[k % len(modes)]
        assert _same(gated.linearize(x, m), plain.linearize(x, m)), (k, m)
    armed, abandoned = gated.prequeue_stats()
    assert armed == len(xs) - 1 and abandoned == 0, (armed, abandoned)  # the first call had no pair yet
    # state changes between calls: the waiting pair is for the old state — abandoned, and the call
    # launches as usual
    cov = np.array([[2.0, 0.3, -0.1], [0.3, 0.5, 0.2], [-0.1, 0.2, 1.5]])
    for k, x in enumerate(xs[:12]):
        for c in (plain, gated):
            if k % 3 == 0:
                c.set_loss(mo.LOSS_GEMAN_MCCLURE, 5.0 + k)
            elif k % 3 == 1:
                c.set_covariance(cov * (1 + k))
        assert _same(gated.linearize(x, mo.JAC_ANALYTIC), plain.linearize(x, mo.JAC_ANALYTIC)), k
    armed2, abandoned2 = gated.prequeue_stats()
    assert abandoned2 >= 8 and armed2 > armed, (armed2, abandoned2)
    # other kinds of calls in between: the cost sweep (its own kernel without speculation), the
    # literal evaluation, a device-resident solve, an asynchronous sweep
    for k, x in enumerate(xs[:10]):
        assert gated.compute_cost(x) == plain.compute_cost(x)
        assert _same(gated.linearize(x, mo.JAC_NUMERIC), plain.linearize(x, mo.JAC_NUMERIC))
        if k == 4:
            for c in (plain, gated):
                c.set_kernel_variant(mo.KERNEL_LITERAL)
        if k == 7:
            for c in (plain, gated):
                c.set_kernel_variant(mo.KERNEL_AUTO)
            xg, rg = mo.capi.lm_minimize([gated], [mo.JAC_ANALYTIC], np.zeros(6))
            xp, rp = mo.capi.lm_minimize([plain], [mo.JAC_ANALYTIC], np.zeros(6))
            assert np.array_equal(xg, xp) and rg["iterations"] == rp["iterations"]
    # with speculation the trial cost runs the linearization sweep: pre-queued as well
    for c in (plain, gated):
        c.set_speculation(True)
        c.set_loss(mo.LOSS_NONE)
        c.set_covariance(None)
    before = gated.prequeue_stats()[0]
    for x in xs[:10]:
        assert gated.compute_cost(x) == plain.compute_cost(x)
        assert _same(gated.linearize(x, mo.JAC_ANALYTIC), plain.linearize(x, mo.JAC_ANALYTIC))
    assert gated.prequeue_stats()[0] - before >= 9
    plain.close()
    t0 = time.perf_counter()
    gated.close()  # a pair is waiting: abandoned, not waited out
    assert time.perf_counter() - t0 < 0.04


def test_a_pair_nobody_arms_ends_by_itself(hip_lib):
    """The caller goes away for longer than the bound: the waiting kernels give up on the device
    (50 ms), the next call does not arm a pair that old (20 ms) and launches as usual."""
    mo = hip_lib
    src, tgt = ds.synthetic_pair(5_000, seed=6, noise=0.02)
    plain = mo.Point2PointCost(src, tgt)
    gated = mo.Point2PointCost(src, tgt)
    gated.set_prequeue(True)
    for c in (plain, gated):
        c.set_speculation(False)
    x = ds.X_GENERIC
    want = plain.linearize(x, mo.JAC_ANALYTIC)
    assert _same(gated.linearize(x, mo.JAC_ANALYTIC), want)
    for pause in (0.03, 0.12):   # past the host bound; past the device timeout as well
        time.sleep(pause)
        t0 = time.perf_counter()
        assert _same(gated.linearize(x, mo.JAC_ANALYTIC), want)
        assert time.perf_counter() - t0 < 0.06, pause
    armed, abandoned = gated.prequeue_stats()
    assert armed == 0 and abandoned == 2, (armed, abandoned)
    assert _same(gated.linearize(x, mo.JAC_ANALYTIC), want)   # and back to armed pairs
    assert gated.prequeue_stats()[0] == 1
    gated.set_prequeue(False)
    assert _same(gated.linearize(x, mo.JAC_ANALYTIC), want)
    for c in (plain, gated):
        c.close()


def test_lm_loop_over_a_prequeued_cost(hip_lib, oracle, facade):
    """The host loop (update / linearize / computeCost through the boundary) over a pre-queued cost:
    the iterates of the loop over the ordinary cost, bit for bit."""
    from tests.test_gpu_device_lm import host_lm
    mo = hip_lib
    src, tgt = facade
    plain = mo.Point2PointCost(src, tgt)
    gated = mo.Point2PointCost(src, tgt)
    gated.set_prequeue(True)
    for jac in (mo.JAC_NUMERIC, mo.JAC_ANALYTIC):
        xp, sp, ip = host_lm(plain, jac, np.zeros(6), max_iter=30)
        xg, sg, ig = host_lm(gated, jac, np.zeros(6), max_iter=30)
        assert (sp, ip) == (sg, ig) and np.array_equal(xp, xg), (jac, xp, xg)
    assert gated.prequeue_stats()[0] > 10
    for c in (plain, gated):
        c.close()
