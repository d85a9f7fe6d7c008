import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import moptimizer_0_amd as mo
from tests import datasets as ds
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
src, tgt = ds.synthetic_pair(n, seed=42, noise=0.01)
cost = mo.Point2PointCost(src, tgt)
for variant in (mo.KERNEL_AUTO, mo.KERNEL_MOMENTS_ALWAYS):
    cost.set_kernel_variant(variant)
    for _ in range(20):
        mo.capi.lm_minimize([cost], [mo.JAC_NUMERIC], np.zeros(6), max_iterations=50)
cost.close()
