cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/fdhome2
for rep in 1 2; do
for home in 2 3; do
  export MOPT_FD_ROTATION_HOME=$home
  for n in 10000000 1000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fdhome2/home${home}_${n}_$rep -o fd -- python3 bench.py --mode numeric --variant literal --cov general --n $n --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
    rm -f gpurun_out/fdhome2/home${home}_${n}_$rep/fd_kernel_trace.csv
  done
done
done
MOPT_FD_ROTATION_HOME=3 python -m pytest tests/test_gpu_parity.py -x -q -k "step_size or cov or numeric" 2>&1 | tail -2
