cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/f32c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_lm.py tests/test_gpu_dropin_cpp.py -x -q 2>&1 | tail -2
for home in 0 2; do
  export MOPT_FD_ROTATION_HOME=$home
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32c/home$home -o k -- python3 bench.py --dtype f32 --mode numeric --variant literal --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  rm -f gpurun_out/f32c/home$home/k_kernel_trace.csv
done
