#!/bin/bash
# Kernel statistics of BASELINE config 5 solved by the device-resident loop (one sweep launch and one
# finalize-and-step per evaluated point) -> gpurun_out/camera_lm_prof/
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/camera_lm_prof -- python3 scripts/camera_lm_timing.py > gpurun_out/camera_lm_prof.log 2>&1
f=$(find gpurun_out/camera_lm_prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/camera_lm_kernel_stats.csv; cut -d, -f1-4 "$f" | cut -c1-160; fi
grep -v amdgpu.ids gpurun_out/camera_lm_prof.log | tail -3
