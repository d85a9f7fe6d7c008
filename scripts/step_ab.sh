#!/bin/bash
# Same-box A/B of library builds (build/ab/lib_<name>.so) on the blocking step (bench.py: moments and
# literal forward differences) at 1 M and 10 M, untraced step times and rocprofv3 finalize averages.
#   scripts/step_ab.sh <name> <name> ...      (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/stepab
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  for n in 1000000 10000000; do
    python3 bench.py --n $n --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v moments n=$n rep=$rep ms_per_step %.5f kernel_ms %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
    python3 bench.py --mode numeric --variant literal --cov symmetric --n $n --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fd-sym  n=$n rep=$rep ms_per_step %.5f kernel_ms %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stepab/${v}_$rep -o fd -- python3 bench.py --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  rm -f gpurun_out/stepab/${v}_$rep/fd_kernel_trace.csv
done
done
