#!/bin/bash
# Device-resident LM over the two costs of BASELINE config 5: one finalize kernel per cost
# (MOPT_LM_MERGE=0) against one finalize over the rows of both (default).
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_device_lm.py -q -m gpu -x > gpurun_out/lm_merge_tests.log 2>&1 || { tail -30 gpurun_out/lm_merge_tests.log; exit 1; }
tail -2 gpurun_out/lm_merge_tests.log
echo "== one finalize per cost"; MOPT_LM_MERGE=0 python scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids
echo "== one finalize over all rows"; python scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids
