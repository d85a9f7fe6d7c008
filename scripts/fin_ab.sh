#!/bin/bash
# Same-box A/B of two builds of the library (build/ab/lib_<name>.so): rocprofv3 kernel averages of
# the finalize kernels at 1 M correspondences under the host loop (moments, forward differences with
# identity / general covariance) and under the device-resident loop.
#   scripts/fin_ab.sh <name> <name> ...      (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/finab
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/finab/${v}_mom_$rep -o fd -- python3 bench.py --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/finab/${v}_fdid_$rep -o fd -- python3 bench.py --mode numeric --variant literal --cov identity --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/finab/${v}_fdgen_$rep -o fd -- python3 bench.py --mode numeric --variant literal --cov general --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/finab/${v}_lm_$rep -o fd -- python3 scripts/lm_profile.py 1000000 20 > gpurun_out/finab/${v}_lm_$rep.txt 2>&1
  find gpurun_out/finab -name '*kernel_trace.csv' -delete
done
done
