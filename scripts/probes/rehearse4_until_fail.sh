#!/bin/bash
# the 4-rank rehearsal of the driver's command, repeated: stops at the first line that lacks a transport
mkdir -p gpurun_out/r5
for i in 1 2 3 4 5; do
  MOPT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
    --master-addr 127.0.0.1 --master-port $((29540 + i)) bench.py --gpus 4 --steps 20 --warmup 5 --collective rccl \
    > gpurun_out/r5/mr4_$i.json 2> gpurun_out/r5/mr4_$i.err
  python3 - $i <<'PY'
import json, sys
i = sys.argv[1]
txt = [l for l in open("gpurun_out/r5/mr4_%s.json" % i) if l.lstrip().startswith("{")]
l = json.loads(txt[-1])
a, b = l["config4_strong"]["ms_per_step_by_collective"], l["ms_per_step_by_collective"]
print(i, sorted(a), sorted(b), flush=True)
sys.exit(0 if {"none", "host", "peer"} <= set(a) and {"none", "host", "peer"} <= set(b) else 1)
PY
  if [ $? -ne 0 ]; then grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|^\*\*\*\|Gloo" gpurun_out/r5/mr4_$i.err | tail -20 | cut -c1-400; break; fi
done
