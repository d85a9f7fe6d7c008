#!/bin/bash
# The driver's multi-GPU bench command with 6 ranks on however many GPUs this box has (gloo as the
# rank backend when they have to share one: the host / peer combines run, RCCL cannot).  Outside
# pytest, because the pool allows 6 processes on a GPU at once.
#   scripts/rehearse_driver_command.sh [ranks] > gpurun_out/r3_6rank_rehearsal.json
set -euo pipefail
cd "$(dirname "$0")/.."
RANKS=${1:-6}
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
if [ "$NGPU" -lt "$RANKS" ]; then export MOPT_BENCH_BACKEND=gloo; fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
START=$(date +%s)
if [ "$RANKS" -ge 6 ]; then
  # torch.distributed.run's agent process holds the GPU too: with 6 ranks bench.py starts them itself
  python bench.py --gpus "$RANKS" --steps 20 --warmup 5
else
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$RANKS" --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus "$RANKS" --steps 20 --warmup 5
fi
echo "wall seconds: $(( $(date +%s) - START ))" >&2
