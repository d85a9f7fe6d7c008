"""One device-resident solve of a small point2point problem (for the -DMOPT_LM_TIMING build, which
prints the phases of every evaluated point from the device)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import moptimizer_0_amd as mo  # noqa: E402
from tests import datasets as ds  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
jac = int(sys.argv[2]) if len(sys.argv) > 2 else 0
src, tgt = ds.synthetic_pair(n, seed=5, noise=0.01)
cost = mo.Point2PointCost(src, tgt)
for _ in range(2):
    x, rep = mo.capi.lm_minimize([cost], [jac], np.zeros(6), max_iterations=3)
print(x, rep)
