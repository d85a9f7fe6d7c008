#!/bin/bash
# Round-5 evidence run on the GPU box: the driver's command and the default bench line (with their
# configs block), the kernel trace of the default run, the PMC traffic passes of the same command,
# kernel traces of BASELINE configs 2 / 3 / 3-literal / 5 alone, solve times, search timings.
#   bash scripts/r5_profiles.sh     -> gpurun_out/r5p/...   (condensed by scripts/summarize_profiles.py)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5p
mkdir -p $out
echo "== the driver's command, three times, then the default line"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err; done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json < /dev/null; echo
echo "== kernel trace of the default run (configs block included: its kernels are in the stats)"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --no-cpu-baseline --no-configs > $out/bench_traced.json 2> $out/bench_traced.err
find $out/bench -name "*kernel_trace.csv" -delete
echo "== PMC passes of the same command (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  echo "   pass $c"; timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $d.json 2> $d.err
  find $d -name "*kernel_trace.csv" -delete
done
echo "== configs 2, 3, 3-literal (1 M) and 5 (camera) alone: kernel traces"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg2 -o t -- python3 bench.py --n 1000000 --no-cpu-baseline --no-configs > $out/cfg2.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3 -o t -- python3 bench.py --n 1000000 --mode numeric --no-cpu-baseline --no-configs > $out/cfg3.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3l -o t -- python3 bench.py --n 1000000 --mode numeric --variant literal --no-cpu-baseline --no-configs > $out/cfg3l.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/camera -o t -- python3 bench.py --workload camera > $out/camera.json 2>/dev/null
find $out/cfg2 $out/cfg3 $out/cfg3l $out/camera -name "*kernel_trace.csv" -delete
echo "== forward differences (literal), 10 M, identity covariance: line"
python3 bench.py --mode numeric --variant literal --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $out/fd10m.json 2>/dev/null
echo "== solve times"; ./tests/cpp/_build/bench_solve 1000 100000 1000000 10000000 > $out/solve.md 2>&1; cat $out/solve.md < /dev/null
echo "== correspondence search"
(python3 scripts/icp_timing.py; python3 scripts/icp_timing.py --dtype f32) 2>&1 | grep -v amdgpu.ids > $out/icp_timing.txt; cat $out/icp_timing.txt < /dev/null
(for k in 2 4 8 16; do python3 scripts/icp_offsets_timing.py --surface --radius-spacings $k --offsets 0,0.1,0.3,0.7 2>/dev/null; done) > $out/icp_surface.txt
python3 tests/tools/icp_lm_probe.py 2>&1 | grep -v "amdgpu.ids\|^[0-9]* dev" > $out/icp_solve.txt
python3 scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids > $out/camera_solve.txt
echo done
