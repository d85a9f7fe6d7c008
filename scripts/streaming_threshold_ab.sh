#!/bin/bash
# Plain against non-temporal loads of the point2point sweep for data sets inside the 256 MiB
# Infinity Cache (24 ... 240 MB): does keeping the tiles cached between the sweeps of a solve pay?
# (MOPT_STREAMING_LOADS: 0 = the library's choice — non-temporal past 32 MiB —, 1 never, 2 always)
for n in 500000 1000000 2000000 3000000 4000000 5000000; do
  for s in 0 1 2; do
    line=$(MOPT_STREAMING_LOADS=$s python3 bench.py --n $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1)
    python3 - "$n" "$s" "$line" <<'PY'
import sys, json
n, s, line = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.loads(line)
print("n=%s streaming=%s (0 auto, 1 never, 2 always): step %.2f us, kernel %.2f us, frac %.3f" % (n, s, d["ms_per_step"]*1e3, d.get("kernel_ms", float("nan"))*1e3, d["roofline"]["frac"]))
PY
  done
done
