#!/bin/bash
# Same-box A/B of library builds (build/ab/lib_<name>.so) under the device-resident loop at 1 M and
# 10 M correspondences, moments sweep:   scripts/gate_ab.sh <name> <name> ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/gateab
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  for n in 1000000 10000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gateab/${v}_${n}_$rep -o fd -- python3 scripts/lm_profile.py $n 10 > gpurun_out/gateab/${v}_${n}_$rep.txt 2>&1
    rm -f gpurun_out/gateab/${v}_${n}_$rep/fd_kernel_trace.csv
    python3 scripts/lm_profile.py $n 20 > gpurun_out/gateab/${v}_${n}_${rep}_untraced.txt 2>&1
  done
done
done
