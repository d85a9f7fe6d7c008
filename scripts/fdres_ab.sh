cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/fdres
for v in new2 new3; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  for n in 1000000 10000000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fdres/${v}_$n -o fd -- python3 scripts/lm_profile.py $n 10 2 1 > gpurun_out/fdres/${v}_$n.txt 2>&1
    rm -f gpurun_out/fdres/${v}_$n/fd_kernel_trace.csv
  done
done
unset MOPT_LIBRARY
python -m pytest tests/test_gpu_device_lm.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
