#!/bin/bash
# Forward-difference sweep under a general covariance: rotation entries re-read from LDS (home 0) or the
# first two perturbed rotations kept in registers (home 2); rocprofv3 kernel averages, 10 M and 1 M.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/fdhome
for rep in 1 2; do
for home in 0 2; do
  export MOPT_FD_ROTATION_HOME=$home
  for n in 10000000 1000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fdhome/home${home}_${n}_$rep -o fd -- python3 bench.py --mode numeric --variant literal --cov general --n $n --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
    rm -f gpurun_out/fdhome/home${home}_${n}_$rep/fd_kernel_trace.csv
  done
done
done
unset MOPT_FD_ROTATION_HOME
python -m pytest tests/test_gpu_parity.py -x -q -k "step_size or cov or numeric" 2>&1 | tail -2
MOPT_FD_ROTATION_HOME=2 python -m pytest tests/test_gpu_parity.py -x -q -k "step_size or cov or numeric" 2>&1 | tail -2
