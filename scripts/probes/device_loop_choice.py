"""The registration of bench_solve (x0 = 0 -> the fixture pose, forward differences) under the
device-resident loop: per size, the solve time with the sweep chosen per evaluated point (default) and
with the moments at every point (MOPT_KERNEL_MOMENTS_ALWAYS), and how many points took the literal
sweep (mopt_cost_lm_choice_stats)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import moptimizer_0_amd as mo  # noqa: E402
from tests import datasets as ds  # noqa: E402

for n in (1000, 100_000, 1_000_000, 10_000_000):
    src, tgt = ds.synthetic_pair(n, seed=42, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    row = []
    for variant in (mo.KERNEL_AUTO, mo.KERNEL_MOMENTS_ALWAYS, mo.KERNEL_LITERAL):
        cost.set_kernel_variant(variant)
        for _ in range(3):
            mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=50)
        p0, l0 = cost.lm_choice_stats()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=50)
            ts.append(time.perf_counter() - t0)
        p1, l1 = cost.lm_choice_stats()
        row.append((np.median(ts) * 1e3, rep["iterations"], rep["sweeps"], (p1 - p0) // 15, (l1 - l0) // 15))
    (a, ia, sa, pa, la), (m, im, sm, _, _), (lt, il, sl, _, _) = row
    print("n=%9d: chosen per point %.3f ms (%d iterations, %d sweeps, %d of %d points literal) | moments at every "
          "point %.3f ms (%d, %d) | literal at every point %.3f ms (%d, %d) | per sweep %.1f / %.1f / %.1f us"
          % (n, a, ia, sa, la, pa, m, im, sm, lt, il, sl, 1e3 * a / sa, 1e3 * m / sm, 1e3 * lt / sl), flush=True)
    cost.close()
