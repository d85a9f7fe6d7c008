#!/bin/bash
# rocprofv3 --kernel-trace --stats for the BASELINE.json configurations other than the default
# bench line (which profiles/README.md already covers).  Run on the GPU box from the repo root:
#   bash scripts/profile_configs.sh          -> gpurun_out/prof_<tag>/..., gpurun_out/prof_<tag>.json
# then scripts/summarize_profiles.py <tag> <stats.csv> condenses each into profiles/.
set -e -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
run() {  # tag, bench flags...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o $tag -- \
      python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/prof_$tag.json 2> $OUT/prof_$tag.err
  tail -n 1 $OUT/prof_$tag.json
}
run cfg2_analytic_1M   --n 1000000  --mode analytic
run cfg3_numeric_1M    --n 1000000  --mode numeric
run cfg3_numeric_1M_literal --n 1000000 --mode numeric --variant literal
run numeric_10M        --n 10000000 --mode numeric
run analytic_10M_f32   --n 10000000 --mode analytic --dtype f32
run cfg5_camera        --workload camera
