"""bench.py's own rank launcher, exercised where there is no GPU: `python bench.py --gpus N` must
start N fresh rank processes with the torch.distributed environment, and — the product having no
CPU path — every rank must refuse loudly, the parent passing the failure on."""
import os
import subprocess
import sys

import pytest

from tests import datasets as ds


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="a HIP device is present: covered by test_gpu_multirank")
def test_self_launch_starts_every_rank_and_fails_loudly_without_a_gpu():
    env = dict(os.environ, MOPT_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ds.ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "2", "--warmup", "1", "--n", "1000"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = out.stderr.decode()
    assert out.returncode != 0
    assert out.stdout.decode().strip() == ""          # no JSON line for a run that measured nothing
    assert "rank 0 of 2: no HIP device" in err and "rank 1 of 2: no HIP device" in err, err[-2000:]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ds.ROOT, "bench.py"), "--gpus", "2"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in out.stderr.decode()
