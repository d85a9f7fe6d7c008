#!/bin/bash
# Round-3 evidence run on the GPU box: kernel traces of the default run and of the forward-difference
# sweeps under every covariance form, the device-resident solve, solve times, rehearsals.
#   bash scripts/r3_profiles.sh     -> gpurun_out/r3p/...
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r3p
mkdir -p $out
echo "== default bench line"; python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 700 $out/bench_default.json < /dev/null
echo "== kernel trace of the default run"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_traced.json 2> $out/bench_traced.err
echo "== forward differences, literal evaluation: identity / symmetric / general covariance, 10 M and 1 M"
for cov in identity symmetric general; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/fd10m_$cov -o fd -- python3 bench.py --mode numeric --variant literal --cov $cov --steps 100 --warmup 10 --no-cpu-baseline > $out/fd10m_$cov.json 2> $out/fd10m_$cov.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/fd1m_$cov -o fd -- python3 bench.py --mode numeric --variant literal --cov $cov --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > $out/fd1m_$cov.json 2> $out/fd1m_$cov.err
done
echo "== device-resident LM, 1 M"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/lm1m -o lm -- python3 scripts/lm_profile.py 1000000 20 > $out/lm1m.log 2>&1
echo "== solve times"; ./tests/cpp/_build/bench_solve 1000 100000 1000000 10000000 > $out/solve.md 2>&1; cat $out/solve.md < /dev/null
echo "== camera config"; python3 bench.py --workload camera > $out/bench_camera.json 2> $out/bench_camera.err; tail -c 300 $out/bench_camera.json < /dev/null
echo "== the driver's multi-GPU command, 6 ranks on this GPU"; bash scripts/rehearse_driver_command.sh 6 > $out/6rank_rehearsal.json 2> $out/6rank_rehearsal.err; tail -c 300 $out/6rank_rehearsal.json < /dev/null; tail -2 $out/6rank_rehearsal.err
echo done
