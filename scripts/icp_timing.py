#!/usr/bin/env python3
"""Correspondence-search (update step) timing: N sources against N targets in a box whose density
gives ~`per_cell` targets per cube of the search radius (the grid's cells are finer than the radius
where that is more than about one: `reach` cells to the radius)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo

mo.capi.device_count()
DTYPES = {"f64": np.float64, "f32": np.float32}
dtype_name = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "f64"
for n, per_cell in ((1_000, 1.0), (1_000_000, 1.0), (1_000_000, 4.0), (4_000_000, 1.0)):
    rng = np.random.default_rng(1)
    side = 100.0
    tgt = rng.random((n, 3)) * side
    src = tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3))
    max_dist = side * (per_cell / n) ** (1.0 / 3.0)
    t0 = time.perf_counter()
    cost = mo.IcpCost(src, tgt, max_dist, dtype=DTYPES[dtype_name])
    build = time.perf_counter() - t0
    reach = cost.grid()[1]
    x = np.array([0.01, -0.01, 0.02, 0.001, -0.002, 0.001])
    cost.update(x)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); m = cost.update(x); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print(dtype_name + " n=%d targets per radius cube~%.0f (reach %d) max_dist=%.3f: create (upload + GPU grid build + first search) %.0f ms; "
          "update %.3f ms = %.2e sources/s, %d matched" % (n, per_cell, reach, max_dist, build * 1e3, dt * 1e3, n / dt, m), flush=True)
    cost.close()
