#!/bin/bash
# Same-box A/B on BASELINE config 5 (two reprojection costs, 40 k + 60 k elements): the round-3 library
# against the current tree — the blocking step of bench.py --workload camera, the whole solves, and
# rocprofv3 kernel averages of the solves.
#   scripts/camera_ab_r4.sh    (GPU box; build/ab/lib_r3.so = round 3's library, build/ab/lib_new.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/camab
for rep in 1 2; do
for v in r3 new; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  python3 bench.py --workload camera --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep=$rep camera step ms_per_step %.5f kernel_ms %.5f unlinked %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['ms_per_step_unlinked']))"
  python3 scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/$v rep=$rep /"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/camab/${v}_$rep -o k -- python3 scripts/camera_lm_timing.py > /dev/null 2>&1
  rm -f gpurun_out/camab/${v}_$rep/k_kernel_trace.csv
done
done
unset MOPT_LIBRARY
python3 - <<'PY'
import csv, glob, os
for path in sorted(glob.glob("gpurun_out/camab/*/**/*kernel_stats.csv", recursive=True)):
    tag = path.split("/")[2]
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("void ", "").replace("mopt::(anonymous namespace)::", "").split("(")[0]
        if "reproj" in name or "finalize" in name or "lmStep" in name:
            print("%-10s %-44s calls %6s avg %9.1f ns" % (tag, name, r["Calls"], float(r["AverageNs"])))
PY
