"""Host-side logic of the drop-in (LM statuses and error behaviour, pivoted LDL^T, dense matrix,
SO(3) exp/log, loss weights, logger, exception) — tests/cpp/host_logic.cpp, built with g++ only
and run without a GPU or the HIP library."""
import os
import subprocess

from tests import datasets as ds

CPP = os.path.join(ds.ROOT, "tests", "cpp")


def test_host_logic_program():
    build = subprocess.run(["make", "-C", CPP, "host"], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-2000:]
    out = subprocess.run([os.path.join(CPP, "_build", "host_logic")], capture_output=True, text=True,
                         timeout=120)
    failed = [l for l in out.stdout.splitlines() if l.startswith("FAIL")]
    assert out.returncode == 0 and not failed, "\n".join(failed) or out.stdout[-2000:]
    summary = [l for l in out.stdout.splitlines() if l.startswith("SUMMARY")][0]
    assert int(summary.split()[1]) >= 49 and summary.endswith("failures=0")
