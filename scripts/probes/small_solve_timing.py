"""Whole device-resident solves of small point2point problems: one launch of one workgroup
(p2pSolveSmallKernel) against the launch-per-point loop (MOPT_LM_ONE_LAUNCH_TILES=0)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import moptimizer_0_amd as mo  # noqa: E402
from tests import datasets as ds  # noqa: E402

for dtype, name in ((np.float64, "f64"), (np.float32, "f32")):
    per_tile = 512 if dtype == np.float64 else 1024
    for n in (100, per_tile, 2 * per_tile, 3 * per_tile, 4 * per_tile):
        src, tgt = ds.synthetic_pair(n, seed=5, noise=0.01)
        cost = mo.Point2PointCost(src.astype(dtype), tgt.astype(dtype), dtype=dtype)
        for jac, jname in ((0, "analytic"), (2, "numeric")):
            row = []
            for tiles in ("4", "0"):
                os.environ["MOPT_LM_ONE_LAUNCH_TILES"] = tiles
                x0 = np.zeros(6, dtype=dtype)
                for _ in range(20):
                    mo.capi.lm_minimize([cost], [jac], x0)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter()
                    x, rep = mo.capi.lm_minimize([cost], [jac], x0)
                    ts.append(time.perf_counter() - t0)
                row.append((np.median(ts) * 1e6, rep["iterations"], rep["sweeps"]))
            (a, it, sw), (b, _, _) = row
            print("%s n=%5d %-8s: one launch %6.1f us, launch per point %6.1f us (%d iterations, %d sweeps): %.1f / %.1f us per point"
                  % (name, n, jname, a, b, it, sw, a / sw, b / sw), flush=True)
        cost.close()
