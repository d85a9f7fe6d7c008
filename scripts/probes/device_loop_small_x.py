import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import moptimizer_0_amd as mo
from tests import datasets as ds, oracle_binding as ob
oracle = ob.load()
rng = np.random.default_rng(3)
n = 100_000
src = rng.random((n, 3)) * 10.0
x_true = np.array([0.02, -0.03, 0.01, 0.004, -0.006, 0.005])
T = oracle.se3_from_x(x_true)
tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3))
x0 = 0.5 * x_true
cost = mo.Point2PointCost(src, tgt)
for variant, name in ((mo.KERNEL_AUTO, "auto"), (mo.KERNEL_LITERAL, "literal"), (mo.KERNEL_MOMENTS, "moments"), (mo.KERNEL_MOMENTS_ALWAYS, "always")):
    cost.set_kernel_variant(variant)
    for k in (1, 2, 3):
        x, rep = mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], x0, max_iterations=k)
        xr, st, it = oracle.p2p_minimize(src, tgt, x0, cost_class=ob.NUMERIC_DYN, layout=ob.LAYOUT_ROW_MAJOR, max_iter=k)
        print(name, k, rep["status"], rep["iterations"], st, it, "max|dx| %.3e  |x| %.3e" % (np.abs(x - xr).max(), np.abs(xr).max()))
    H, b, s = cost.linearize(x0, mo.JAC_NUMERIC)
    Hr, br, sr = oracle.p2p_linearize(src, tgt, x0, cost_class=ob.NUMERIC_DYN)
    print(name, "blocking linearize at x0: H rel %.2e b rel %.2e" % (np.abs(H-Hr).max()/np.abs(Hr).max(), np.abs(b-br).max()/np.abs(br).max()))
