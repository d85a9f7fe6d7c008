#!/usr/bin/env python3
"""A few device-resident LM solves (mopt_lm_minimize) for a kernel trace:
    rocprofv3 --kernel-trace --stats -d out -- python3 scripts/lm_profile.py [n] [solves] [mode] [variant]
(mode: Jacobian mode, default 2 = forward differences; variant: kernel variant, default 0 = AUTO,
1 = the literal evaluation)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
src, tgt = ds.synthetic_pair(n, seed=42, noise=0.01)
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cost = mo.Point2PointCost(src, tgt)
cost.set_kernel_variant(variant)
best = 1e9
for k in range(solves):
    t0 = time.perf_counter()
    x, rep = mo.capi.lm_minimize([cost], [mode], np.zeros(6), max_iterations=50)
    best = min(best, time.perf_counter() - t0)
print("n=%d mode %d variant %d: device-resident solve %.3f ms best of %d, %d sweeps, %d iterations, status %d -> %.1f us per sweep"
      % (n, mode, variant, best * 1e3, solves, rep["sweeps"], rep["iterations"], rep["status"], best * 1e6 / rep["sweeps"]))
cost.close()
