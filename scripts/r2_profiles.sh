#!/bin/bash
# Round-2 evidence run on the GPU box: kernel traces, PMC traffic, solve times, the 2-rank rehearsal.
#   bash scripts/r2_profiles.sh     -> gpurun_out/r2p/...
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2p
mkdir -p $out
echo "== default bench line"; python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json < /dev/null
echo "== kernel trace of the default run"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_traced.json 2> $out/bench_traced.err
echo "== PMC passes (FETCH_SIZE, WRITE_SIZE) of the default run"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d.json 2> $d.err
done
echo "== forward differences, literal evaluation (10 M and 1 M)"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fd10m -o fd -- python3 bench.py --mode numeric --variant literal --steps 100 --warmup 10 --no-cpu-baseline > $out/fd10m.json 2> $out/fd10m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fd1m -o fd -- python3 bench.py --mode numeric --variant literal --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > $out/fd1m.json 2> $out/fd1m.err
echo "== device-resident LM, 1 M"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/lm1m -o lm -- python3 scripts/lm_profile.py 1000000 20 > $out/lm1m.log 2>&1
echo "== solve times"; ./tests/cpp/_build/bench_solve 1000 100000 1000000 10000000 > $out/solve.md 2>&1; cat $out/solve.md < /dev/null
echo "== 2-rank rehearsal (ranks share this GPU)"; MOPT_BENCH_BACKEND=gloo python3 bench.py --gpus 2 > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank_rehearsal.err; tail -c 400 $out/bench_2rank_rehearsal.json < /dev/null
echo "== camera config"; python3 bench.py --workload camera > $out/bench_camera.json 2> $out/bench_camera.err
echo done
