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
            # "4": one launch wherever the kernel can; "0": never; "default": what the library chooses
            for tiles in ("4", "0", "default"):
                if tiles == "default":
                    os.environ.pop("MOPT_LM_ONE_LAUNCH_TILES", None)
                else:
                    os.environ["MOPT_LM_ONE_LAUNCH_TILES"] = tiles
                # forward differences under AUTO choose their sweep per evaluated point (the one-launch kernel
                # then holds both forms): the first two columns time MOMENTS_ALWAYS (moments at every point,
                # the moments-only kernel), the third the library's default
                cost.set_kernel_variant(mo.KERNEL_AUTO if (tiles == "default" or jac != 2)
                                        else mo.KERNEL_MOMENTS_ALWAYS)
                x0 = np.zeros(6, dtype=dtype)

                def timed(**kw):
                    for _ in range(20):
                        mo.capi.lm_minimize([cost], [jac], x0, **kw)
                    ts = []
                    for _ in range(200):
                        t0 = time.perf_counter()
                        x, rep = mo.capi.lm_minimize([cost], [jac], x0, **kw)
                        ts.append(time.perf_counter() - t0)
                    return np.median(ts) * 1e6, rep["iterations"], rep["sweeps"]

                # to the loop's own stop, and cut off after 3 outer iterations: with forward differences the
                # noise-level stop fires an iteration or three apart between two summation orders, so the
                # full solves of the two loops need not evaluate the same number of points; the truncated ones do
                row.append(timed() + timed(max_iterations=3))
            (a, it, sw, a3, _, sw3), (b, it_b, sw_b, b3, _, sw3_b), (d, it_d, sw_d, d3, _, sw3_d) = row
            print("%s n=%5d %-8s: one launch %6.1f us (%2d iterations, %2d sweeps), launch per point %6.1f us "
                  "(%2d, %2d): %.1f / %.1f us per point; default %6.1f us (%2d, %2d) | 3 iterations: %5.1f (%d sweeps) "
                  "/ %5.1f (%d) / default %5.1f (%d)%s"
                  % (name, n, jname, a, it, sw, b, it_b, sw_b, a / sw, b / sw_b, d, it_d, sw_d, a3, sw3, b3, sw3_b,
                     d3, sw3_d, "" if d3 <= 1.08 * min(a3, b3) else "   <-- default is not the faster form"),
                  flush=True)
        cost.close()
