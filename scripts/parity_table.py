#!/usr/bin/env python3
"""Measured parity (norm-wise relative error of H, b; relative error of the cost) of the HIP path
against the CPU restatement, per Jacobian mode / kernel variant / size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    for mode, mname in ((0, "analytic"), (1, "analytic, tst layout"), (2, "forward differences")):
        cc = ob.NUMERIC_DYN if mode == 2 else ob.ANALYTIC_DYN
        layout = ob.LAYOUT_TST if mode == 1 else ob.LAYOUT_ROW_MAJOR
        for variant, vname in ((2, "moments"), (1, "literal")):
            cost.set_kernel_variant(variant)
            for x, xname in ((ds.X_ZERO, "0"), (ds.X_GENERIC, "generic")):
                H, b, s = cost.linearize(x, mode)
                Hr, br, sr = o.p2p_linearize(src, tgt, x, cost_class=cc, layout=layout)
                print("| %d | %s | %s | %s | %.1e | %.1e | %.1e |" % (n, mname, vname, xname, rel(H, Hr), rel(b, br), abs(s - sr) / sr), flush=True)
