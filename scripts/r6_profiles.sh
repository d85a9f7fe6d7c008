#!/bin/bash
# Round-6 evidence run on the GPU box.
#   * the driver's command three times and the default bench line (configs block: BASELINE configs 1, 2, 3,
#     5, the rotating costs, the 100 M hbm_check);
#   * kernel traces (rocprofv3 --kernel-trace --stats) of (a) the bench with its launches dominated by the
#     ROTATING costs — the figure roofline.frac / roofline.kernel_ms report —, (b) the same command with the
#     headline's own cost swept back to back (roofline.same_cost), (c) the 100 M cost alone;
#   * the PMC traffic passes (FETCH_SIZE, WRITE_SIZE: separate passes) of (a);
#   * configs 2 / 3 / 3-literal / 5 alone, solve times, the device loop's per-point choice, small solves.
#   bash scripts/r6_profiles.sh     -> gpurun_out/r6p/...   (condensed by scripts/summarize_profiles.py)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6p
mkdir -p $out
ROT="--no-cpu-baseline --no-configs --hbm-check-n 0 --settle-ms 0 --warmup 2 --steps 20 --kernel-steps 20 --rotating-launches 300"
echo "== the driver's command, three times, then the default line"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err; done
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 300 $out/bench_default.json < /dev/null; echo
echo "== (a) kernel trace, launches dominated by the rotating costs"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rot -o rot -- python3 bench.py $ROT > $out/rot_traced.json 2> $out/rot_traced.err
find $out/rot -name "*kernel_trace.csv" -delete
echo "== (b) kernel trace, the headline's own cost back to back"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/same -o same -- python3 bench.py --no-cpu-baseline --no-configs --hbm-check-n 0 --rotating-costs 0 > $out/same_traced.json 2> $out/same_traced.err
find $out/same -name "*kernel_trace.csv" -delete
echo "== (c) kernel trace, one cost of 100 M correspondences"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/big -o big -- python3 bench.py --n 100000000 --steps 20 --warmup 3 --settle-ms 0 --kernel-steps 20 --no-cpu-baseline --no-configs --hbm-check-n 0 --rotating-costs 0 > $out/big_traced.json 2> $out/big_traced.err
find $out/big -name "*kernel_trace.csv" -delete
echo "== PMC passes of (a) (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  echo "   pass $c"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python3 bench.py $ROT > $d.json 2> $d.err
  find $d -name "*kernel_trace.csv" -delete
done
echo "== configs 2, 3, 3-literal (1 M) and 5 (camera) alone: kernel traces"
ONE="--no-cpu-baseline --no-configs --hbm-check-n 0 --rotating-costs 0"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg2 -o t -- python3 bench.py --n 1000000 $ONE > $out/cfg2.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3 -o t -- python3 bench.py --n 1000000 --mode numeric $ONE > $out/cfg3.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3l -o t -- python3 bench.py --n 1000000 --mode numeric --variant literal $ONE > $out/cfg3l.json 2>/dev/null
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/camera -o t -- python3 bench.py --workload camera > $out/camera.json 2>/dev/null
find $out/cfg2 $out/cfg3 $out/cfg3l $out/camera -name "*kernel_trace.csv" -delete
echo "== forward differences (literal), 10 M, identity covariance: line"
python3 bench.py --mode numeric --variant literal --steps 100 --warmup 10 --no-cpu-baseline --no-configs --hbm-check-n 0 > $out/fd10m.json 2>/dev/null
echo "== solve times"; ./tests/cpp/_build/bench_solve 1000 100000 1000000 10000000 > $out/solve.md 2>&1; cat $out/solve.md < /dev/null
python3 scripts/probes/device_loop_choice.py 2>&1 | grep -v amdgpu.ids > $out/device_loop_choice.txt; cat $out/device_loop_choice.txt < /dev/null
python3 scripts/probes/small_solve_timing.py 2>&1 | grep -v amdgpu.ids > $out/small_solve.txt
python3 scripts/camera_lm_timing.py 2>&1 | grep -v amdgpu.ids > $out/camera_solve.txt
python3 scripts/size_sweep.py --tag r6 2>/dev/null > $out/size_sweep_rows.md
echo done
