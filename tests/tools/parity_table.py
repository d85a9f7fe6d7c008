#!/usr/bin/env python3
"""Measured parity (norm-wise relative error of H, b; relative error of the cost) of the HIP path
against the CPU restatement: per Jacobian mode / kernel variant / size, and — for forward
differences, whose step is h_j = sqrt(eps) |x_j| (linearization.h:85) — per decade of |x_j|."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds, oracle_binding as ob
o = ob.load()
def rel(a, b): return np.abs(np.asarray(a, float) - b).max() / np.abs(b).max()
print("| N | mode | kernel | x | max|dH|/max|H| | max|db|/max|b| | |dcost|/cost |")
print("|---|---|---|---|---|---|---|")
for n in (1000, 100_000, 1_000_000):
    src, tgt = ds.synthetic_pair(n, seed=42, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    for mode, mname in ((0, "analytic"), (1, "analytic, tst layout"), (3, "analytic, left perturbation"),
                        (2, "forward differences")):
        cc = ob.NUMERIC_DYN if mode == 2 else ob.ANALYTIC_DYN
        layout = {1: ob.LAYOUT_TST, 3: ob.LAYOUT_LEFT}.get(mode, ob.LAYOUT_ROW_MAJOR)
        for variant, vname in ((2, "moments"), (1, "literal")):
            cost.set_kernel_variant(variant)
            for x, xname in ((ds.X_ZERO, "0"), (ds.X_GENERIC, "generic")):
                H, b, s = cost.linearize(x, mode)
                Hr, br, sr = o.p2p_linearize(src, tgt, x, cost_class=cc, layout=layout)
                print("| %d | %s | %s | %s | %.1e | %.1e | %.1e |" % (n, mname, vname, xname, rel(H, Hr), rel(b, br), abs(s - sr) / sr), flush=True)
    cost.close()

print()
print("Forward differences by decade of |x_j| (20 random x per decade, worst case; N = 100 000):")
print()
print("| |x_j| ~ | h_j ~ | kernel | worst max|dH|/max|H| | worst max|db|/max|b| |")
print("|---|---|---|---|---|")
rng = np.random.default_rng(5)
src, tgt = ds.synthetic_pair(100_000, seed=42, noise=0.01)
cost = mo.Point2PointCost(src, tgt)
for scale in (1e-8, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1, 1.0):
    for variant, vname in ((2, "moments"), (1, "literal")):
        cost.set_kernel_variant(variant)
        wH = wb = 0.0
        for _ in range(20):
            x = rng.choice([-1.0, 1.0], 6) * scale * rng.uniform(0.3, 3.0, 6)
            H, b, s = cost.linearize(x, 2)
            Hr, br, sr = o.p2p_linearize(src, tgt, x, cost_class=ob.NUMERIC_DYN, layout=ob.LAYOUT_ROW_MAJOR)
            wH, wb = max(wH, rel(H, Hr)), max(wb, rel(b, br))
        print("| %.0e | %.1e | %s | %.1e | %.1e |" % (scale, 1.49e-8 * scale, vname, wH, wb), flush=True)
cost.close()
