#!/bin/bash
# Same-box A/B of library builds (build/ab/lib_<name>.so) on the fp32 forward-difference sweep
# (identity / symmetric covariance, 10 M and 1 M): rocprofv3 kernel averages.
#   scripts/f32_fd_ab.sh <name> <name> ...     (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/f32ab
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  for cov in identity symmetric; do
    for n in 10000000 1000000; do
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32ab/${v}_${cov}_${n}_$rep -o k -- python3 bench.py --dtype f32 --mode numeric --variant literal --cov $cov --n $n --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
      rm -f gpurun_out/f32ab/${v}_${cov}_${n}_$rep/k_kernel_trace.csv
    done
  done
done
done
