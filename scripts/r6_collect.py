#!/usr/bin/env python3
"""Condense what scripts/r6_profiles.sh left under gpurun_out/r6p/ into the tracked files under profiles/
(run here, after gpurun has merged gpurun_out/ back).  Prints the figures DESIGN.md §5 quotes."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r6p")
P = os.path.join(ROOT, "profiles")


def summarize(*args):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarize_profiles.py")] + list(args),
                          stdout=subprocess.DEVNULL)


def main():
    summarize("r6_bench_rotating_10M", os.path.join(R, "rot", "rot_kernel_stats.csv"),
              os.path.join(R, "pmc_fetch", "pmc_counter_collection.csv"),
              os.path.join(R, "pmc_write", "pmc_counter_collection.csv"), "analytic_f64_n10000000")
    summarize("r6_bench_same_cost_10M", os.path.join(R, "same", "same_kernel_stats.csv"))
    summarize("r6_hbm_check_100M", os.path.join(R, "big", "big_kernel_stats.csv"))
    for c in ("cfg2", "cfg3", "cfg3l", "camera"):
        summarize("r6_" + c, os.path.join(R, c, "t_kernel_stats.csv"))
    shutil.copy(os.path.join(R, "bench_default.json"), os.path.join(P, "r6_bench_line.json"))
    with open(os.path.join(P, "r6_driver_command_lines.txt"), "w") as f:
        for i in (1, 2, 3):
            f.write(open(os.path.join(R, "bench_driver_%d.json" % i)).read())
    for src, dst in (("solve.md", "r6_solve_times.md"), ("device_loop_choice.txt", "r6_device_loop_choice.txt"),
                     ("camera_solve.txt", "r6_camera_solve.txt")):
        shutil.copy(os.path.join(R, src), os.path.join(P, dst))
    head = ("# scripts/probes/small_solve_timing.py, MI355X: whole device-resident solves of one small point2point "
            "cost (median of 200, Python call included)\n"
            "# one launch = p2pSolveSmallKernel (<= 4 tiles held in registers); launch per point = "
            "MOPT_LM_ONE_LAUNCH_TILES=0; each with its own (iterations, sweeps); then the same solves cut off after 3\n"
            "# outer iterations (with forward differences the noise-level stop fires an iteration or three apart "
            "between two summation orders: the full solves need not evaluate the same number of points, the "
            "truncated ones do)\n"
            "# analytic rows: every column under MOPT_KERNEL_AUTO.  numeric rows: the first two columns under "
            "MOPT_KERNEL_MOMENTS_ALWAYS (the moments-only kernel), default = MOPT_KERNEL_AUTO: the sweep of every\n"
            "# point chosen by the step, in the one-launch kernel that holds both forward-difference forms\n")
    with open(os.path.join(P, "r6_small_solve.txt"), "w") as f:
        f.write(head + open(os.path.join(R, "small_solve.txt")).read())
    # the size sweep's table keeps its hand-written head: only the rows are replaced
    md = os.path.join(P, "r6_size_sweep.md")
    rows = open(os.path.join(R, "size_sweep_rows.md")).read()
    if os.path.exists(md) and rows.strip():
        text = open(md).read()
        cut = text.index("|---|---|---|---|---|---|---|---|\n") + len("|---|---|---|---|---|---|---|---|\n")
        open(md, "w").write(text[:cut] + rows)
    js = os.path.join(ROOT, "gpurun_out", "r6_size_sweep.json")
    if os.path.exists(js):
        shutil.copy(js, os.path.join(P, "r6_size_sweep.json"))
    # ---- what the documents quote ------------------------------------------------------------------
    for name in ["bench_driver_%d" % i for i in (1, 2, 3)] + ["bench_default"]:
        d = json.load(open(os.path.join(R, name + ".json")))
        r, c = d["roofline"], d["configs"]
        print("%-15s step %.2f us | frac %.4f (%.2f us) same %.4f (%.2f us) 100M %.4f literal %.4f | cfg1 solve %.4f "
              "ms (one launch, moments only %.4f) cfg2 %.2f/%.2f cfg3 %.2f/%.2f cfg3l %.2f/%.2f cfg5 %.2f/%.2f | cpu %.3g"
              % (name, d["ms_per_step"] * 1e3, r["frac"], r["kernel_ms"] * 1e3, r["same_cost"]["frac"],
                 r["same_cost"]["kernel_ms"] * 1e3, r["hbm_check_frac"], r["literal_frac"], c["cfg1"]["solve_ms"],
                 c["cfg1"]["solve_ms_moments_always"], c["cfg2"]["ms_per_step"] * 1e3, c["cfg2"]["kernel_ms"] * 1e3,
                 c["cfg3"]["ms_per_step"] * 1e3, c["cfg3"]["kernel_ms"] * 1e3, c["cfg3_literal"]["ms_per_step"] * 1e3,
                 c["cfg3_literal"]["kernel_ms"] * 1e3, c["cfg5"]["ms_per_step"] * 1e3, c["cfg5"]["kernel_ms"] * 1e3,
                 d["cpu_baseline"]["value"]))
    for t in ("rot", "same", "big", "cfg2", "cfg3", "cfg3l", "camera"):
        for f in glob.glob(os.path.join(R, t, "*kernel_stats.csv")):
            for row in csv.DictReader(open(f)):
                if "mopt" in row["Name"] and int(row["Calls"]) > 20 and "inalize" not in row["Name"]:
                    print("%-7s %-60s calls %6s avg %10.1f ns sd %s" % (
                        t, row["Name"].replace("void mopt::(anonymous namespace)::", "").split("(")[0][:60],
                        row["Calls"], float(row["AverageNs"]), row["StdDev"][:7]))
    t = json.load(open(os.path.join(P, "r6_bench_rotating_10M_hbm_traffic.json")))
    k = [k for k in t if k.startswith("p2pMomentsKernel")][0]
    print("traffic per launch %.2f MB over %d launches" % (t[k]["hbm_bytes_per_launch"] / 1e6, t[k]["launches_FETCH_SIZE"]))
    print(open(os.path.join(R, "solve.md")).read())
    print(open(os.path.join(R, "device_loop_choice.txt")).read())


if __name__ == "__main__":
    main()
