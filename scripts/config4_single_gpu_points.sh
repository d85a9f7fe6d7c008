#!/bin/bash
# Single-GPU step time at the per-GPU shard sizes of BASELINE config 4 (10 M correspondences over
# 1 / 2 / 4 / 8 GPUs): the inputs of DESIGN.md's scaling prediction.
out=gpurun_out/r2q
mkdir -p $out
for n in 10000000 5000000 2500000 1250000; do
  python3 bench.py --n $n --steps 500 --warmup 50 --no-cpu-baseline > $out/single_$n.json 2> $out/single_$n.err
  python3 - "$out/single_$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("n = %d: %.2f us per step, sweep kernel %.2f us" % (d["config"]["correspondences_per_gpu"], d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3))
PY
done
