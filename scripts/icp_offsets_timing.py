#!/usr/bin/env python3
"""Correspondence search against the distance between the clouds: 1 M sources whose targets lie
`offset` cells away (0 = aligned, 0.2 = the benchmark of icp_timing.py, 0.7 and 1.5 = most lanes
have to look beyond the 2 x 2 x 2 cells of the first round).  MOPT_ICP_FIRST_ROUND=0 runs the
row-by-row search alone.
MOPT_ICP_REACH=k forces k cells to the radius (default: by the density of the targets).
Usage: python scripts/icp_offsets_timing.py [--dtype f32] [--per-cell 1 | --surface --radius-spacings 3] [--offsets 0,0.2,0.7]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo

def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default

dtype = {"f64": np.float64, "f32": np.float32}[arg("--dtype", "f64")]
per_cell = float(arg("--per-cell", "1"))
n = int(arg("--n", "1000000"))
mo.capi.device_count()
rng = np.random.default_rng(1)
side = 100.0
if "--surface" in sys.argv:
    # a scanned surface: a wavy sheet through the box; the radius is given in point spacings
    uv = rng.random((n, 2)) * side
    tgt = np.column_stack([uv, 10.0 * np.sin(uv[:, 0] / 10.0) * np.cos(uv[:, 1] / 15.0) + 50.0])
    max_dist = float(arg("--radius-spacings", "3")) * side / np.sqrt(n)
else:
    tgt = rng.random((n, 3)) * side
    max_dist = side * (per_cell / n) ** (1.0 / 3.0)
offsets = [float(v) for v in arg("--offsets", "0,0.2,0.7,1.5").split(",")]
for offset in offsets:
    shift = np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0) * offset * max_dist
    src = tgt[rng.permutation(n)] + rng.normal(0, 0.01 * max_dist, (n, 3)) + shift
    t0 = time.perf_counter()
    cost = mo.IcpCost(src, tgt, max_dist, dtype=dtype)
    create_ms = (time.perf_counter() - t0) * 1e3
    x = np.zeros(6)
    cost.update(x)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); m = cost.update(x); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    what = ("surface, radius %s spacings" % arg("--radius-spacings", "3")) if "--surface" in sys.argv \
        else "targets per radius cube~%.0f" % per_cell
    print("first_round=%s reach=%d %s n=%d %s offset %.2f radii: update %.3f ms, %d matched (create %.1f ms)"
          % (os.environ.get("MOPT_ICP_FIRST_ROUND", "1"), cost.grid()[1], arg("--dtype", "f64"), n, what, offset, dt * 1e3, m, create_ms), flush=True)
    cost.close()
