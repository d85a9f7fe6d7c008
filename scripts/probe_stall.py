"""Where is the one-time ~40 ms pause?  Times (a) blocking linearize calls that really sweep,
(b) calls answered from the kept result (no GPU work), (c) a trivial ctypes call, and reports every
call slower than 1 ms with its index and the time since the loop started."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import moptimizer_0_amd as mo
from tests import datasets as ds


def loop(name, fn, n=3000):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts)
    idx = np.nonzero(ts > 1000)[0]
    print("%-34s median %6.1f us; >1 ms at calls %s (%s us), %s ms into the loop" %
          (name, np.median(ts), idx.tolist(), ts[idx].round(0).tolist(),
           [round(ts[:i].sum() / 1e3, 1) for i in idx]), flush=True)


order = sys.argv[1:] or ["trivial", "cached", "sweeps"]
src, tgt = ds.synthetic_pair(100000, seed=1)
x = ds.X_GENERIC
lib = mo.capi.load()
for what in order:
    if what == "trivial":
        loop("trivial ctypes call", lib.mopt_version)
    else:
        c = mo.Point2PointCost(src, tgt)
        c.set_speculation(what == "cached")
        if what == "cached":
            c.compute_cost(x)      # keeps the linearization at x
        loop("linearize, %s" % ("answered from the kept result" if what == "cached" else "real sweeps"),
             lambda: c.linearize(x, 0))
        c.close()
