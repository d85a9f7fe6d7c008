#!/usr/bin/env python3
"""Randomized call sequences through the C ABI against a stateless checker.

The library keeps state between calls — the kept result of a speculative sweep, sweeps queued for
linked costs, resident constants of the device-resident loop, kernel variant, loss, covariance,
replaced data.  Here a seeded random walk over those calls runs on a handful of costs (point2point
in both kernel variants, reprojection, a run-time compiled model), and every number a call returns is
compared with what a fresh evaluation of the same cost state gives: the CPU restatement (oracle)
where it has the model, otherwise a twin cost object that is never linked, never speculates and is
asked nothing else.

    python tests/tools/api_fuzz.py [--seed S] [--steps N]

Exit status 0 = every call agreed.  (Test infrastructure: uses oracle/.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo  # noqa: E402
from tests import datasets as ds  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402

REL = 1e-6


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max()
    return np.abs(a - b).max() / scale if scale > 0 else np.abs(a).max()


class Subject:
    """A cost under test + the way to evaluate its current state from scratch."""

    def __init__(self, name, cost, twin, jac_modes):
        self.name, self.cost, self.twin, self.jac_modes = name, cost, twin, jac_modes
        self.loss, self.cov = (0, 0.0), None
        twin.set_speculation(False)

    def set_loss(self, kind, param):
        self.loss = (kind, param)
        self.cost.set_loss(kind, param)
        self.twin.set_loss(kind, param)

    def set_covariance(self, cov):
        self.cov = cov
        self.cost.set_covariance(cov)
        self.twin.set_covariance(cov)

    def want_linearize(self, x, jac):
        return self.twin.linearize(x, jac)

    def want_cost(self, x):
        return self.twin.compute_cost(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    run(ap.parse_args())


def run(args):
    rng = np.random.default_rng(args.seed)
    oracle = ob.load()

    src, tgt = ds.synthetic_pair(30_000, seed=args.seed, noise=0.02)
    pts, pix = ds.synthetic_camera(20_000, seed=args.seed + 1)
    t = np.linspace(0.0, 4.0, 3000)
    y = np.exp(0.25 * t + 0.1) + 0.01 * rng.standard_normal(t.size)
    p2p_jit_planes = np.vstack([src.T, tgt.T])
    p2p_setup = """
  const S th = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
  S R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > S(10) * S(2.220446049250313e-16)) {
    const S k[3] = {x[3] / th, x[4] / th, x[5] / th};
    const S K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    const S s = sin(th), c1 = S(1) - cos(th);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        S kk = 0;
        for (int q = 0; q < 3; ++q) kk += K[i * 3 + q] * K[q * 3 + j];
        R[i * 3 + j] = ((i == j ? S(1) : S(0)) + s * K[i * 3 + j]) + c1 * kk;
      }
  }
  for (int i = 0; i < 9; ++i) a[i] = R[i];
  a[9] = x[0]; a[10] = x[1]; a[11] = x[2];"""
    p2p_res = """
  for (int i = 0; i < 3; ++i)
    r[i] = (((a[i * 3] * d[0] + a[i * 3 + 1] * d[1]) + a[i * 3 + 2] * d[2]) + a[9 + i]) - d[3 + i];"""

    def jit_p2p():
        return mo.JitModelCost(6, 3, p2p_res, planes=p2p_jit_planes, n_aux=12, setup_body=p2p_setup)

    subjects = [
        Subject("p2p/auto", mo.Point2PointCost(src, tgt), mo.Point2PointCost(src, tgt),
                [mo.JAC_ANALYTIC, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_LEFT, mo.JAC_ANALYTIC_RIGHT]),
        Subject("p2p/literal", mo.Point2PointCost(src[:9000], tgt[:9000]),
                mo.Point2PointCost(src[:9000], tgt[:9000]),
                [mo.JAC_ANALYTIC, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_RIGHT]),
        # a second literally evaluated cost: with the first one it is swept by one launch in the
        # device-resident solves that pick both in the same mode (round 3)
        Subject("p2p/literal-b", mo.Point2PointCost(src[9000:20000], tgt[9000:20000]),
                mo.Point2PointCost(src[9000:20000], tgt[9000:20000]),
                [mo.JAC_ANALYTIC, mo.JAC_NUMERIC, mo.JAC_ANALYTIC_RIGHT]),
        Subject("camera/a", mo.ReprojectionCost(pts[:8000], pix[:8000]),
                mo.ReprojectionCost(pts[:8000], pix[:8000]), [mo.JAC_NUMERIC]),
        Subject("camera/b", mo.ReprojectionCost(pts[8000:], pix[8000:]),
                mo.ReprojectionCost(pts[8000:], pix[8000:]), [mo.JAC_NUMERIC]),
        Subject("jit/p2p", jit_p2p(), jit_p2p(), [mo.JAC_NUMERIC]),
        # three tiles: alone in a device-resident solve it is minimised by one launch of one workgroup —
        # the moments-only kernel, or (forward differences under AUTO) the one that holds both forms (round 6)
        Subject("p2p/small", mo.Point2PointCost(src[20000:21500], tgt[20000:21500]),
                mo.Point2PointCost(src[20000:21500], tgt[20000:21500]), [mo.JAC_ANALYTIC, mo.JAC_NUMERIC]),
    ]
    # an ICP cost: its correspondences are a function of the last update(x); never queued ahead
    tgt_cloud = tgt[rng.permutation(len(tgt))[:12_000]]
    icp = Subject("icp", mo.IcpCost(src[:6000], tgt_cloud, 2.0), mo.IcpCost(src[:6000], tgt_cloud, 2.0),
                  [mo.JAC_ANALYTIC, mo.JAC_NUMERIC])
    subjects.append(icp)
    for k in (1, 2):
        subjects[k].cost.set_kernel_variant(mo.KERNEL_LITERAL)
        subjects[k].twin.set_kernel_variant(mo.KERNEL_LITERAL)
    curve = Subject("curve", mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, y),
                    mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t, y), [mo.JAC_NUMERIC])
    pose = [s for s in subjects]          # the 6-parameter costs share parameter vectors
    recent = [rng.uniform(-0.3, 0.3, 6) for _ in range(4)]
    calls = {}

    def note(kind):
        calls[kind] = calls.get(kind, 0) + 1

    def pick_x():
        if rng.random() < 0.55:
            return recent[rng.integers(len(recent))].copy()   # the loop revisits points
        x = rng.uniform(-0.4, 0.4, 6) * (rng.random() < 0.8) + 0.0
        recent[rng.integers(len(recent))] = x.copy()
        return x

    def check_linearize(s, x, jac):
        H, b, c = s.cost.linearize(x, jac)
        Hw, bw, cw = s.want_linearize(x, jac)
        e = max(rel(H, Hw), rel(b, bw) if np.abs(bw).max() > 1e-9 * np.abs(Hw).max() else 0.0,
                abs(c - cw) / max(abs(cw), 1e-300))
        assert e <= REL, (s.name, "linearize", jac, x, e)
        note("linearize")

    def check_cost(s, x):
        c, cw = s.cost.compute_cost(x), s.want_cost(x)
        assert abs(c - cw) <= REL * abs(cw), (s.name, "cost", x, c, cw)
        note("compute_cost")

    for step in range(args.steps):
        op = rng.random()
        s = pose[rng.integers(len(pose))]
        if op < 0.34:
            check_linearize(s, pick_x(), s.jac_modes[rng.integers(len(s.jac_modes))])
        elif op < 0.60:
            check_cost(s, pick_x())
        elif op < 0.72:
            # the optimizer's pattern over a random subset: linearize all at x0, cost of all at xi
            group = [pose[i] for i in rng.permutation(len(pose))[:rng.integers(1, len(pose) + 1)]]
            x0, xi = pick_x(), pick_x()
            for g in group:
                check_linearize(g, x0, g.jac_modes[-1] if g.name.startswith("camera") else g.jac_modes[0])
            for g in group:
                check_cost(g, xi)
            note("loop pattern")
        elif op < 0.78:
            chosen = [pose[i] for i in rng.permutation(len(pose))[:rng.integers(0, len(pose) + 1)]]
            mo.capi.link_costs([g.cost for g in chosen])
            note("link")
        elif op < 0.83:
            kind = int(rng.integers(2))
            s.set_loss(kind, float(rng.uniform(0.5, 50.0)) if kind else 0.0)
            note("set_loss")
        elif op < 0.87:
            m = 2 if s.name.startswith("camera") else 3
            if rng.random() < 0.3:
                s.set_covariance(None)
            else:
                A = rng.standard_normal((m, m))
                cov = A @ A.T / m + np.eye(m)
                if rng.random() < 0.3:
                    cov = cov + 0.1 * rng.standard_normal((m, m))   # not symmetric
                s.set_covariance(cov)
            note("set_covariance")
        elif op < 0.89:
            s.cost.set_speculation(bool(rng.integers(2)))
            note("set_speculation")
        elif op < 0.93:
            # a device-resident solve over a random subset, against the host loop over the twins
            # costs of one kind per problem (a camera pose and a cloud alignment share no minimum: their
            # sum is an ill-conditioned problem whose iterates amplify forward-difference noise)
            cameras = rng.random() < 0.4
            family = [g for g in pose if g.name.startswith("camera") == cameras and g.name != "icp"]
            # (mopt_lm_minimize sums at most 4 costs)
            group = [family[i] for i in rng.permutation(len(family))[:rng.integers(1, min(len(family), 4) + 1)]]
            modes = [g.jac_modes[-1] if g.name.startswith("camera") else g.jac_modes[0] for g in group]
            modes = [m if m != mo.JAC_ANALYTIC_LEFT else mo.JAC_ANALYTIC for m in modes]
            if not cameras and rng.random() < 0.5:
                # the literally evaluated costs in forward differences: still one launch for the pair
                modes = [mo.JAC_NUMERIC if g.name.startswith(("p2p/literal", "p2p/small")) else m
                         for g, m in zip(group, modes)]
            # (the reprojection problem from within its basin: far from it the robust loss saturates and
            # the iterates become a noise amplifier, which compares rounding, not code paths)
            x0 = (np.array([-0.01, 0.02, -0.058, 0.018, -0.0013, 0.027]) + 0.003 * rng.standard_normal(6)
                  if cameras else
                  # 0 or |x_j| >= 0.1: with h_j = sqrt(eps) |x_j| below ~1e-9 two runs whose iterates
                  # differ in the last bit draw different forward-difference noise of relative size
                  # eps / h_j, and then differ by that much one iteration later
                  rng.uniform(0.1, 0.3, 6) * rng.choice([-1.0, 1.0], 6) * (rng.random() < 0.5))
            k = int(rng.integers(1, 4))
            x_dev, rep = mo.capi.lm_minimize([g.cost for g in group], modes, x0, max_iterations=k)
            x_host = host_lm([g.twin for g in group], modes, x0, k)
            # one iteration, or analytic Jacobians: the two loops see the same numbers (tight).  Forward
            # differences over several iterations: iterates that differ in the last bit draw different
            # rounding noise in the next Jacobian (relative eps / h_j), so they agree to that only.
            numeric = any(m == mo.JAC_NUMERIC for m in modes)
            tol = 3e-5 if (numeric and k > 1) else 2e-6
            assert np.abs(x_dev - x_host).max() < tol * max(1.0, np.abs(x_host).max()), \
                ([(g.name, g.loss, None if g.cov is None else g.cov.tolist()) for g in group], modes,
                 x0.tolist(), k, rep, x_dev, x_host)
            note("lm_minimize")
        elif op < 0.945:
            xu = pick_x()
            xu[:3] += np.array([10.5, 10.2, 0.1]) * rng.random()   # towards the fixture pose: real matches
            na = icp.cost.update(xu)
            nb = icp.twin.update(xu)
            assert na == nb, ("icp update", xu, na, nb)
            note("icp update")
        elif op < 0.96 and s.name.startswith("p2p"):
            n = int(rng.integers(100, 9000))
            a, b = ds.synthetic_pair(n, seed=int(rng.integers(1 << 30)), noise=0.02)
            s.cost.set_data(a, b)
            s.twin.set_data(a, b)
            note("set_data")
        else:
            # |x_j| >= 0.05: a forward step h = sqrt(eps) |x_j| amplifies the last-bit difference between
            # the device's exp and glibc's by eps / h (profiles/NOTES.md, "Parity") — not what is being tested here
            xc = rng.uniform(0.05, 0.5, 2) * rng.choice([-1.0, 1.0], 2)
            H, b, c = curve.cost.linearize(xc, mo.JAC_NUMERIC)
            Hw, bw, cw = oracle.scalar_linearize(1, t, y, xc, numeric=True)
            assert max(rel(H, Hw), rel(b, bw), abs(c - cw) / cw) <= REL, ("curve", xc)
            note("curve vs oracle")
        if step % 500 == 499:
            print("step %d ok: %s" % (step + 1, calls), flush=True)
    mo.capi.link_costs([])
    print("PASS seed %d: %d steps, calls %s, answered ahead %s" % (
        args.seed, args.steps, calls, [s.cost.answered_ahead() for s in pose]))


def host_lm(costs, modes, x0, max_iter, lm_iter=3):
    x = np.array(x0, dtype=np.float64)
    lam, eps = -1.0, np.finfo(np.float64).eps
    for _ in range(max_iter):
        H, b, y0 = 0.0, 0.0, 0.0
        for c, m in zip(costs, modes):
            Hc, bc, yc = c.linearize(x, m)
            H, b, y0 = H + Hc, b + bc, y0 + yc
        if abs(y0) < 8 * eps:
            return x
        if lam < 0:
            lam = 1e-9 * np.abs(np.diag(H)).max()
        nu = 2.0
        # Eigen::LDLT (levenberg_marquadt_dyn.cpp:78-80) reads the lower triangle only: with a
        # covariance that is not symmetric H is not either, and that is the system actually solved
        H = np.tril(H) + np.tril(H, -1).T
        for _ in range(lm_iter):
            delta = np.linalg.solve(H + lam * np.diag(np.diag(H)), -b)
            xi = x + delta
            yi = sum(c.compute_cost(xi) for c in costs)
            rho = (y0 - yi) / delta.dot(lam * delta - b)
            if rho < 0:
                if np.abs(delta).max() < np.sqrt(eps):
                    return x
                lam *= nu
                nu *= 2
                continue
            x = xi
            lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            break
    return x


if __name__ == "__main__":
    main()
