import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds
src, tgt = ds.synthetic_pair(100000, seed=1)
c = mo.Point2PointCost(src, tgt)
x = ds.X_GENERIC
t_start = time.perf_counter()
ts = []
for i in range(3000):
    t0 = time.perf_counter(); c.linearize(x, 0); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
idx = np.nonzero(ts > 1000)[0]
print("stalls >1ms at calls", idx, ts[idx].round(0), "cumulative time at stall (ms):", [round(ts[:i].sum()/1e3,1) for i in idx])
print("median", np.median(ts))
