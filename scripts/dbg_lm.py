import sys, numpy as np
sys.path.insert(0, '.')
import moptimizer_0_amd as mo
from tests import datasets as ds, oracle_binding as ob
o = ob.load()
src, tgt = ds.synthetic_pair(1000, seed=5, noise=0.01)
cost = mo.Point2PointCost(src, tgt)
for k in (1, 2, 3, 15):
    x, rep = mo.capi.lm_minimize([cost], [2], np.zeros(6), max_iterations=k)
    xr, st, it = o.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN, layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
    print(k, rep, st, it, np.abs(x - xr).max())
