"""The first blocking calls of a fresh cost, one by one (us): direct dispatch (MOPT_AQL=1) against the HIP stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import moptimizer_0_amd as mo
from tests import datasets as ds
src, tgt = ds.synthetic_pair(100_000, seed=2, noise=0.01)
for trial in range(3):
    c = mo.Point2PointCost(src, tgt)
    c.set_speculation(False)
    call, x_in, H, b, s = c.bound_linearize(mo.JAC_NUMERIC)
    ts = []
    for k in range(14):
        x_in[:] = ds.X_GENERIC + 1e-3 * k
        t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e6)
    print("MOPT_AQL=%s cost %d:" % (os.environ.get("MOPT_AQL", "1"), trial), " ".join("%.1f" % t for t in ts), flush=True)
    c.close()
