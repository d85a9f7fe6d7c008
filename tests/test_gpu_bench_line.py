"""The single-GPU line of bench.py (what the driver records as BENCH_rNN.json), at a reduced size so
that it runs in seconds: every object the contract names, the oracle check, and every BASELINE
config that fits one GPU in the `configs` block."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import datasets as ds

pytestmark = pytest.mark.gpu


def test_single_gpu_line_carries_every_object(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(ds.ROOT, "bench.py"), "--gpus", "1", "--steps", "20",
                          "--warmup", "5", "--n", "1000000", "--cpu-seconds", "1", "--hbm-check-n", "3000000",
                          "--rotating-costs", "3"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["unit"] and line["value"] > 0 and line["higher_is_better"] is True
    assert line["dtype"] == "f64" and line["data"].startswith("synthetic") and line["vs_baseline"] is None
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1.0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 1e6 and cpu["sample"]
    vs = line["check"]["vs_oracle"]
    assert vs["ok"] and max(vs["H_rel"], vs["b_rel"], vs["cost_rel"]) <= vs["bar"] == 1e-6, vs
    assert len(line["timing"]["per_step_us"]) == 20
    cfgs = line["configs"]
    assert {"cfg1", "cfg2", "cfg3", "cfg3_literal", "cfg5"} <= set(cfgs)
    for name in ("cfg1", "cfg2", "cfg3", "cfg3_literal", "cfg5"):
        assert cfgs[name]["ms_per_step"] > 0 and cfgs[name]["value"] > 0 and cfgs[name]["workload"], name
    for name in ("cfg2", "cfg3", "cfg3_literal"):
        assert 0 < cfgs[name]["frac"] < 1.2 and cfgs[name]["kernel_ms"] > 0, (name, cfgs[name])
    # configs[0] as a solve: the registration of 1 k correspondences ends at the fixture's pose
    c1 = cfgs["cfg1"]
    assert 0 < c1["solve_ms"] < 5.0 and c1["solve_sweeps"] >= c1["solve_iterations"] >= 2
    x = np.array(c1["solve_x"])
    assert np.abs(x[:3] - ds.FIXTURE_T).max() < 0.05, x
    # the timed region as the library counted it: K steps were K sweeps over the data, none answered from
    # a kept result, every one through the library's own packets (or none: MOPT_AQL=0 / a profiler)
    reg = line["check"]["timed_region"]
    assert reg["sweeps"] == 20 and reg["cache_hits"] == 0 and reg["direct_dispatches"] in (0, 20), reg
    if os.environ.get("MOPT_AQL", "1") != "0":
        assert reg["direct_dispatches"] == 20, reg
    for name in ("cfg2", "cfg3", "cfg3_literal"):
        r = cfgs[name]["timed_region"]
        assert r["sweeps"] == 20 and r["cache_hits"] == 0, (name, r)
    r5 = cfgs["cfg5"]["timed_region"]
    assert r5["sweeps"] == 40 and r5["cache_hits"] == 0 and r5["costs"] == 2, r5
    # the two measurements no cache can have served (at the bench's real sizes: 100 M and 4 x 10 M)
    rot, big = cfgs["rotating"], cfgs["hbm_check"]
    assert rot["costs"] == 3 and rot["timed_region"]["sweeps"] == rot["timed_region"]["steps"] >= 20
    assert rot["timed_region"]["cache_hits"] == 0 and 0 < rot["frac"] < 1.2 and rot["kernel_ms"] > 0
    assert big["n"] == 3000000 and big["timed_region"]["sweeps"] == big["timed_region"]["steps"] > 0
    assert big["timed_region"]["cache_hits"] == 0 and 0 < big["frac"] < 1.2
    assert line["roofline"]["frac_rotating"] == rot["frac"] and line["roofline"]["hbm_check_frac"] == big["frac"]
    # the sum of squares is additive over correspondences: 3 M of the same distribution ~ 3 x the 1 M shard's
    assert 2.5 < big["check_sum_sq"] / line["check"]["sum_sq"] < 3.5
