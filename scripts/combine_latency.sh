#!/bin/bash
# What each way of adding the ranks' sums costs when nothing else does: ranks sharing one GPU
# (rehearsal backend), shards so small that a step is pure latency.
export MOPT_BENCH_BACKEND=gloo
out=gpurun_out/r2q
mkdir -p $out
for spec in "2 1000" "2 100000" "3 1000" "4 1000"; do
  set -- $spec
  python3 bench.py --gpus $1 --n $2 --steps 2000 --warmup 100 --settle-ms 20 > $out/lat_g$1_n$2.json 2> $out/lat_g$1_n$2.err
  python3 - "$out/lat_g$1_n$2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["n_gpus"], "ranks, n =", d["config"]["correspondences_per_gpu"], {k: round(v * 1e3, 2) for k, v in d["ms_per_step_by_collective"].items()}, "us per step")
PY
done
