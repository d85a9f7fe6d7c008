#!/bin/bash
# Workgroups per CU for the moments sweep at Infinity-Cache-resident sizes (tuning experiment).
for n in 1000000 2500000; do
  for b in 1 2 3 4; do
    MOPT_BLOCKS_PER_CU=$b python3 bench.py --n $n --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('n=%d blocks/CU=$b: step %.2f us, sweep kernel %.2f us' % (d['config']['correspondences_per_gpu'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  done
done
