#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, as MI355X_MICROARCH.md prescribes: separate
# passes for FETCH_SIZE and WRITE_SIZE, each with --kernel-trace only.  Run on the GPU box:
#   bash scripts/profile_pmc.sh <tag> [bench flags...]   -> gpurun_out/pmc_<tag>_{fetch,write}/
set -e -o pipefail
export TMPDIR=/tmp
tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_${tag}_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- \
      python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $d.json 2> $d.err
  ls $d | head -3
done
