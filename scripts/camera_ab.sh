#!/bin/bash
# Same-box A/B of library builds on BASELINE config 5 (two reprojection costs, 40 k + 60 k elements):
# the blocking step of bench.py --workload camera, the whole solves, and rocprofv3 kernel averages.
#   scripts/camera_ab.sh <name> <name> ...    (GPU box; build/ab/lib_<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/camab
for rep in 1 2; do
for v in "$@"; do
  export MOPT_LIBRARY=$GRAFT_REPO_ROOT/build/ab/lib_$v.so
  python3 bench.py --workload camera --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep=$rep camera step ms_per_step %.5f kernel_ms %.5f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
  python3 scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/$v rep=$rep /"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/camab/${v}_$rep -o k -- python3 scripts/camera_lm_timing.py > /dev/null 2>&1
  rm -f gpurun_out/camab/${v}_$rep/k_kernel_trace.csv
done
done
unset MOPT_LIBRARY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_lm.py tests/test_gpu_dropin_cpp.py -x -q -k "reproj or camera or config5 or several or dropin" 2>&1 | tail -2
