"""A seeded random walk over the stateful calls of the C ABI (speculation, linked costs, kernel variant,
loss, covariance, replaced data, device-resident solves over random subsets of costs), every returned
number checked against a fresh evaluation of the same cost state: tests/tools/api_fuzz.py, two short
seeds here (20 seeds x 8000 steps were run when the linked-cost and merged-finalize paths went in)."""
import argparse
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 102])
def test_random_call_sequences_agree_with_fresh_evaluations(hip_lib, oracle, seed):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "api_fuzz.py")
    spec = importlib.util.spec_from_file_location("api_fuzz", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    fuzz.run(argparse.Namespace(seed=seed, steps=1500))
