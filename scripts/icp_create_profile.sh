#!/bin/bash
# Where the construction of an ICP cost (upload, grid build, first search) spends its time: HIP API
# and kernel totals over ten constructions of a 1 M x 1 M cost.   -> gpurun_out/icp_create_*.txt
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/icp_create
mkdir -p $out
cat > $out/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import moptimizer_0_amd as mo
n = 1_000_000
rng = np.random.default_rng(1)
tgt = rng.random((n, 3)) * 100.0
src = tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3))
mo.IcpCost(src, tgt, 1.0).close()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); c = mo.IcpCost(src, tgt, 1.0); ts.append(time.perf_counter() - t0); c.close()
print("construction: median %.2f ms, min %.2f ms" % (np.median(ts) * 1e3, min(ts) * 1e3))
PY
python3 $out/run.py | tee gpurun_out/icp_create_time.txt
timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $out/run.py > /dev/null 2> $out/trace.err
python3 - $out <<'PY' | tee gpurun_out/icp_create_api.txt
import csv, glob, sys
for kind in ("hip_api_stats", "kernel_stats"):
    for f in glob.glob(sys.argv[1] + "/trace/**/*%s.csv" % kind, recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
        print("==", kind)
        for r in rows[:14]:
            print("%-60s calls %6s total %9.2f ms avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
