#!/bin/bash
# The blocking sweeps through the library's own AQL queues (MOPT_AQL=1, default) against the HIP stream
# (MOPT_AQL=0): default bench line and the driver's command, interleaved, same box.
mkdir -p gpurun_out/r5
for rep in 1 2; do for a in 0 1; do
  for form in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
    MOPT_AQL=$a python3 bench.py $form --cpu-seconds 1 2>/dev/null > gpurun_out/r5/aqlab.json
    python3 - "$a" "$rep" "$form" <<'PY'
import json, sys
l = json.load(open("gpurun_out/r5/aqlab.json")); c = l["configs"]
print("MOPT_AQL=%s rep %s [%s]: step %.2f median %.2f first %.1f kernel %.2f literal %.2f | cfg2 %.2f cfg3 %.2f cfg3l %.2f cfg5 %.2f | vs_oracle %s"
      % (sys.argv[1], sys.argv[2], sys.argv[3], l["ms_per_step"] * 1e3, l["timing"]["median"], l["timing"]["first"],
         l["roofline"]["kernel_ms"] * 1e3, l["roofline"]["literal_kernel_ms"] * 1e3, c["cfg2"]["ms_per_step"] * 1e3,
         c["cfg3"]["ms_per_step"] * 1e3, c["cfg3_literal"]["ms_per_step"] * 1e3, c["cfg5"]["ms_per_step"] * 1e3,
         l["check"]["vs_oracle"]["ok"]), flush=True)
PY
  done
done; done
