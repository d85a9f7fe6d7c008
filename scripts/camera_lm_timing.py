#!/usr/bin/env python3
"""BASELINE config 5 as a solve: camera calibration, 100 000 elements as two costs (40 k + 60 k),
Geman-McClure(100), forward differences — the device-resident loop (mopt_lm_minimize over both
costs) next to the host loop driving the same costs through the blocking boundary."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds

pts, pix = ds.synthetic_camera(100_000, seed=17)
costs = [mo.ReprojectionCost(pts[:40_000], pix[:40_000]), mo.ReprojectionCost(pts[40_000:], pix[40_000:])]
for c in costs:
    c.set_loss(mo.LOSS_GEMAN_MCCLURE, 100.0)


def host_loop(x0, max_iter=25, lm_iter=3):
    x = np.array(x0, dtype=np.float64); lam = -1.0; eps = np.finfo(np.float64).eps; sweeps = 0
    for it in range(max_iter):
        H = np.zeros((6, 6)); b = np.zeros(6); y0 = 0.0
        for c in costs:
            Hc, bc, yc = c.linearize(x, mo.JAC_NUMERIC); H += Hc; b += bc; y0 += yc
        if abs(y0) < 8 * eps: return x, it
        if lam < 0: lam = 1e-9 * np.abs(np.diag(H)).max()
        nu = 2.0
        for _ in range(lm_iter):
            delta = np.linalg.solve(H + lam * np.diag(np.diag(H)), -b); xi = x + delta
            yi = sum(c.compute_cost(xi) for c in costs)
            rho = (y0 - yi) / delta.dot(lam * delta - b)
            if rho < 0:
                if np.abs(delta).max() < np.sqrt(eps): return x, it
                lam *= nu; nu *= 2; continue
            x = xi; lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3); break
    return x, max_iter


def linked_host_loop(x0):
    mo.capi.link_costs(costs)   # mopt_costs_link: the first cost asked queues the other's sweep too
    try:
        return host_loop(x0)
    finally:
        mo.capi.link_costs([])


for name, run in (("device-resident loop", lambda: mo.capi.lm_minimize(costs, [mo.JAC_NUMERIC] * 2, np.zeros(6), max_iterations=25)),
                  ("host loop (python) ", lambda: host_loop(np.zeros(6))),
                  ("host loop, costs linked", lambda: linked_host_loop(np.zeros(6)))):
    best, out = 1e9, None
    for _ in range(7):
        t0 = time.perf_counter(); out = run(); best = min(best, time.perf_counter() - t0)
    extra = ("%d iterations, %d points" % (out[1]["iterations"], out[1]["sweeps"])) if isinstance(out[1], dict) else ("%d iterations" % out[1])
    print("%s: %.3f ms (%s), x = %s" % (name, best * 1e3, extra, np.array2string(np.asarray(out[0]), precision=6)))
