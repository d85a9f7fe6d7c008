#!/bin/bash
# Does placing kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) change call / kernel time?
set -e -o pipefail
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v python3 scripts/size_sweep.py --tag kernarg$v --max-n 10000000 2>/dev/null | grep -E "f64 .*(1000|1000000|10000000) \|"
done
