#!/bin/bash
# Same-box A/B of library builds on the fp32 literal analytic sweep (10 M, 1 M): rocprofv3 kernel averages.
#   scripts/f32_literal_ab.sh <name> <name> ...     (GPU box; build/ab/lib_<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/f32lit
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  for n in 10000000 1000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32lit/${v}_${n}_$rep -o k -- python3 bench.py --dtype f32 --variant literal --n $n --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
    rm -f gpurun_out/f32lit/${v}_${n}_$rep/k_kernel_trace.csv
  done
done
done
unset MOPT_LIBRARY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_lm.py -x -q 2>&1 | tail -2
