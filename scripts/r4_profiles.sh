#!/bin/bash
# Round-4 evidence run on the GPU box: the default bench line and its kernel trace, the PMC traffic passes
# of the same command, the camera configuration (line + kernel trace), generic-sweep and search timings.
#   bash scripts/r4_profiles.sh     -> gpurun_out/r4p/...   (then scripts/summarize_profiles.py, see the end)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4p
mkdir -p $out
echo "== default bench line"; python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json < /dev/null; echo
echo "== kernel trace of the default run"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_traced.json 2> $out/bench_traced.err
find $out/bench -name "*kernel_trace.csv" -delete
echo "== PMC passes of the same command (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $d.json 2> $d.err
  find $d -name "*kernel_trace.csv" -delete
done
echo "== camera config: line, then kernel trace"
python3 bench.py --workload camera > $out/bench_camera.json 2> $out/bench_camera.err; tail -c 400 $out/bench_camera.json < /dev/null; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/camera -o cam -- python3 bench.py --workload camera --no-cpu-baseline > /dev/null 2>&1
find $out/camera -name "*kernel_trace.csv" -delete
echo "== forward differences (literal), identity covariance, 10 M and 1 M: lines"
python3 bench.py --mode numeric --variant literal --steps 100 --warmup 10 --no-cpu-baseline > $out/fd10m.json 2>/dev/null
python3 bench.py --mode numeric --variant literal --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > $out/fd1m.json 2>/dev/null
python3 bench.py --n 1000000 --steps 100 --warmup 10 --no-cpu-baseline > $out/an1m.json 2>/dev/null
echo "== solve times"; ./tests/cpp/_build/bench_solve 1000 100000 1000000 10000000 > $out/solve.md 2>&1; cat $out/solve.md < /dev/null
echo "== correspondence search"
(python3 scripts/icp_timing.py; python3 scripts/icp_timing.py --dtype f32) 2>&1 | grep -v amdgpu.ids > $out/icp_timing.txt; cat $out/icp_timing.txt < /dev/null
(for d in 1 2 4 8 16; do python3 scripts/icp_offsets_timing.py --per-cell $d --offsets 0,0.2,0.4,0.7,1.5 2>/dev/null; done
 for d in 4 16; do MOPT_ICP_REACH=1 python3 scripts/icp_offsets_timing.py --per-cell $d --offsets 0,0.2,0.4,0.7,1.5 2>/dev/null; done) > $out/icp_density.txt
(for k in 2 4 8 16; do python3 scripts/icp_offsets_timing.py --surface --radius-spacings $k --offsets 0,0.1,0.3,0.7 2>/dev/null; done
 for k in 8 16; do MOPT_ICP_REACH=1 python3 scripts/icp_offsets_timing.py --surface --radius-spacings $k --offsets 0,0.1,0.3,0.7 2>/dev/null; done) > $out/icp_surface.txt
python3 tests/tools/icp_lm_probe.py 2>&1 | grep -v "amdgpu.ids\|^[0-9]* dev" > $out/icp_solve.txt
bash scripts/icp_pmc.sh r4 float64 > $out/icp_pmc.log 2>&1
echo done
