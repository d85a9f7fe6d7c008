"""moptimizer_0_amd — MI355X-native linearization path of moptimizer.

The product is the C-ABI shared library (include/moptimizer_hip.h, built from
moptimizer_0_amd/csrc) plus the C++ host classes in include/moptimizer_amd/.  This Python
package is only the plumbing tests/ and bench.py use to reach that library (ctypes) and to
shard a cost over one-process-per-GPU ranks with torch.distributed (RCCL).
"""
from . import _capi as capi  # noqa: F401
from ._capi import (  # noqa: F401
    COMBINE_HOST,
    COMBINE_NONE,
    COMBINE_PEER,
    COMBINE_RCCL,
    JAC_ANALYTIC,
    JAC_ANALYTIC_LEFT,
    JAC_ANALYTIC_RIGHT,
    JAC_ANALYTIC_TST_LAYOUT,
    JAC_NUMERIC,
    KERNEL_AUTO,
    KERNEL_LITERAL,
    KERNEL_MOMENTS,
    KERNEL_MOMENTS_ALWAYS,
    LOSS_GEMAN_MCCLURE,
    LOSS_NONE,
    IcpCost,
    JitModelCost,
    MoptError,
    Point2PointCost,
    Point2PointGroup,
    ReprojectionCost,
    ScalarModelCost,
)
from .sharded import ShardedSweep, shard_range  # noqa: F401
