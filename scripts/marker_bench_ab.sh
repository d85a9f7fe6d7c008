#!/bin/bash
# The driver's command under MOPT_MARKER_EVERY off / 32 and with / without the 2 ms pause after the synchronisation, interleaved, same box.
mkdir -p gpurun_out/r5
for rep in 1 2 3; do
  for every in 1000000000 32; do for pause in 0 2; do
    MOPT_MARKER_EVERY=$every python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 1 --pause-after-sync-ms $pause 2>/dev/null \
      > gpurun_out/r5/mb_${every}_${pause}_$rep.json
    python3 - "$every" "$rep" "$pause" <<'PY'
import json, sys
l = json.load(open("gpurun_out/r5/mb_%s_%s_%s.json" % (sys.argv[1], sys.argv[3], sys.argv[2])))
c = l["configs"]
print("every=%-10s pause %s ms rep %s: headline %.2f us (median %.2f, first %.1f, kernel %.2f) | cfg2 %.2f cfg3 %.2f cfg3l %.2f cfg5 %.2f"
      % (sys.argv[1], sys.argv[3], sys.argv[2], l["ms_per_step"] * 1e3, l["timing"]["median"], l["timing"]["first"],
         l["roofline"]["kernel_ms"] * 1e3, c["cfg2"]["ms_per_step"] * 1e3, c["cfg3"]["ms_per_step"] * 1e3,
         c["cfg3_literal"]["ms_per_step"] * 1e3, c["cfg5"]["ms_per_step"] * 1e3), flush=True)
PY
  done; done
done
