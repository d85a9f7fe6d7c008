cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/f32
for args in "--dtype f32" "--dtype f32 --variant literal" "--dtype f32 --mode numeric --variant literal" "--dtype f32 --mode numeric --variant literal --cov symmetric" "--dtype f32 --n 1000000"; do
  tag=$(echo $args | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32/$tag -o k -- python3 bench.py $args --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/f32/$tag.json 2>/dev/null
  rm -f gpurun_out/f32/$tag/k_kernel_trace.csv
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
