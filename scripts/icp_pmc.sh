#!/bin/bash
# Kernel time and PMC counters of the correspondence search (icpMatchKernel) at 1 M x 1 M, clouds
# 0.2 cells apart: one --stats pass, then one counter group per pass (--kernel-trace only).
#   bash scripts/icp_pmc.sh <tag>   -> gpurun_out/icp_pmc_<tag>.txt
set -e -o pipefail
export TMPDIR=/tmp
tag=$1
dtype=${2:-float64}
out=gpurun_out/icp_pmc_$tag
mkdir -p $out
cat > $out/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import moptimizer_0_amd as mo
n = 1_000_000
rng = np.random.default_rng(1)
tgt = rng.random((n, 3)) * 100.0
src = tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3)) + np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0) * 0.2
cost = mo.IcpCost(src, tgt, 1.0, dtype=np.DTYPE)
for _ in range(20):
    cost.update(np.zeros(6))
cost.close()
PY
sed -i "s/np.DTYPE/np.$dtype/" $out/run.py
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $out/run.py > /dev/null 2> $out/stats.err
# (one counter per pass for the TA / TCP / TD blocks: a group the hardware cannot collect together
# aborts rocprofv3, which then never exits — hence also the timeout around every pass)
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
    TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum \
    TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum \
    TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TD_TD_BUSY_sum \
    TD_TC_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum; do
  d=$out/$(echo $group | cut -d' ' -f1)
  echo "pass: $group"
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $d -o pmc -- python3 $out/run.py > /dev/null 2> $d.err || echo "  (pass failed)"
done
python3 - $out <<'PY' > gpurun_out/icp_pmc_$tag.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icpMatch" in r["Name"]:
            print("kernel_stats", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
sums = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icpMatch" in r["Kernel_Name"]:
            sums[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
for k in sorted(sums):
    print("%-34s %.4g per launch (%d launches)" % (k, sums[k] / calls[k], calls[k]))
PY
cat gpurun_out/icp_pmc_$tag.txt
