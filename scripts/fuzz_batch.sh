#!/bin/bash
# A batch of seeded random walks over the stateful calls (tests/tools/api_fuzz.py), one process at a time.
#   bash scripts/fuzz_batch.sh FIRST_SEED LAST_SEED STEPS > gpurun_out/fuzz.log
set -o pipefail
fail=0
for seed in $(seq $1 $2); do
  if python3 tests/tools/api_fuzz.py --seed $seed --steps $3 > /tmp/fuzz_$seed.log 2>&1; then
    echo "seed $seed: $(tail -1 /tmp/fuzz_$seed.log)"
  else
    echo "seed $seed: FAILED"; tail -20 /tmp/fuzz_$seed.log; fail=1
  fi
done
exit $fail
