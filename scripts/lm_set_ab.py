#!/usr/bin/env python3
"""A/B of the one-launch sweep over several costs of one kind (MOPT_LM_SET=0 keeps a launch per cost):
device-resident solves over 2-4 literal point2point costs / exp-curve costs, best of N.
Usage: [MOPT_LM_SET=0] python scripts/lm_set_ab.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds

def best(costs, modes, x0, reps=30, **kw):
    b, rep = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        x, rep = mo.capi.lm_minimize(costs, modes, x0, **kw)
        b = min(b, time.perf_counter() - t0)
    return b, rep

src, tgt = ds.synthetic_pair(120_000, seed=21, noise=0.02)
for parts in (2, 4):
    cuts = np.linspace(0, len(src), parts + 1).astype(int)
    costs = [mo.Point2PointCost(src[a:b], tgt[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for c in costs:
        c.set_kernel_variant(mo.KERNEL_LITERAL)
    for jac, name in ((mo.JAC_ANALYTIC, "analytic"), (mo.JAC_NUMERIC, "numeric")):
        t, rep = best(costs, [jac] * parts, np.zeros(6))
        print("%d literal point2point costs (%s): %.3f ms, %d sweeps -> %.1f us per evaluated point [MOPT_LM_SET=%s]"
              % (parts, name, t * 1e3, rep["sweeps"], t * 1e6 / rep["sweeps"], os.environ.get("MOPT_LM_SET", "1")), flush=True)
    for c in costs:
        c.close()
rng = np.random.default_rng(5)
t_ = np.linspace(0.0, 5.0, 4000); y_ = np.exp(0.3 * t_ + 0.1) + 0.01 * rng.standard_normal(t_.size)
curves = [mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t_[:1500], y_[:1500]), mo.ScalarModelCost(mo.capi.MODEL_EXP_CURVE, t_[1500:], y_[1500:])]
t, rep = best(curves, [mo.JAC_NUMERIC] * 2, np.zeros(2), max_iterations=50)
print("2 exp-curve costs: %.3f ms, %d sweeps -> %.1f us per evaluated point [MOPT_LM_SET=%s]"
      % (t * 1e3, rep["sweeps"], t * 1e6 / rep["sweeps"], os.environ.get("MOPT_LM_SET", "1")), flush=True)
