"""Where do the device loop and the CPU loop stop on forward differences?  (tests/test_gpu_device_lm.py
test_iterates_match_the_cpu_loop): iterations, status, distance of the iterates, per kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import moptimizer_0_amd as mo
from tests import datasets as ds, oracle_binding as ob
oracle = ob.load()
for n in (1000, 100_000):
    src, tgt = ds.synthetic_pair(n, seed=5, noise=0.01)
    cost = mo.Point2PointCost(src, tgt)
    for variant, name in ((mo.KERNEL_AUTO, "auto=literal"), (mo.KERNEL_MOMENTS, "moments")):
        cost.set_kernel_variant(variant)
        for k in (2, 3, 4, 5, 6, 7, 15):
            x, rep = mo.capi.lm_minimize([cost], [2], np.zeros(6), max_iterations=k)
            xr, status, iters = oracle.p2p_minimize(src, tgt, np.zeros(6), cost_class=ob.NUMERIC_DYN,
                                                    layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
            print("n=%d %s k=%d: device it %d status %d cost %.12g | cpu it %d status %d | max|dx| %.2e"
                  % (n, name, k, rep["iterations"], rep["status"], rep["cost"], iters, status, np.abs(x - xr).max()))
    cost.close()
