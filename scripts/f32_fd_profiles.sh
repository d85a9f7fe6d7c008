cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/f32b
python -m pytest tests/test_gpu_parity.py -x -q -k "float32 or f32 or fp32 or randomized or step_size" 2>&1 | tail -2
for args in "--dtype f32 --mode numeric --variant literal" "--dtype f32 --mode numeric --variant literal --cov symmetric" "--dtype f32 --mode numeric --variant literal --n 1000000"; do
  tag=$(echo $args | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32b/$tag -o k -- python3 bench.py $args --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/f32b/$tag.json 2>/dev/null
  rm -f gpurun_out/f32b/$tag/k_kernel_trace.csv
done
