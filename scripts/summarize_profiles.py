#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/...) into the tracked summaries under profiles/.

  kernel stats : profiles/<tag>_kernel_stats.csv      rows of the --kernel-trace --stats summary
  PMC traffic  : profiles/<tag>_hbm_traffic.json      per-launch FETCH_SIZE / WRITE_SIZE of each
                 mopt kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE
                 counts 64 B per 128-B request of a 16-B-per-lane stream: x2; WRITE_SIZE exact; both
                 in KiB), collected in separate --pmc passes.
Usage: summarize_profiles.py TAG STATS_CSV [FETCH_COUNTER_CSV WRITE_COUNTER_CSV [BENCH_KEY]]
STATS_CSV may be "-" (PMC passes only).  BENCH_KEY ("<mode>_<dtype>_n<N>", e.g. analytic_f64_n10000000)
also records the sweep kernel's bytes per launch in profiles/hbm_traffic.json, the file bench.py reads
`roofline.traffic` from: the sweep kernel is the mopt kernel with the most bytes per launch among those
launched more than twice (which excludes the one-off re-layout)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "").replace("mopt::(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(stats))) if stats != "-" else []
    keep = [r for r in rows if "mopt" in r["Name"] or "copyBuffer" in r["Name"]
            or "rccl" in r["Name"].lower() or "nccl" in r["Name"].lower()]
    if stats != "-":
        with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in keep:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        print("wrote", "%s_kernel_stats.csv" % tag, len(keep), "rows")
    if len(sys.argv) >= 5:
        res = collections.OrderedDict()
        for cname, path in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(path)):
                if "mopt" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                res.setdefault(k, {})[cname + "_KiB_raw_mean"] = sum(v) / len(v)
                res[k]["launches_" + cname] = len(v)
        for k, d in res.items():
            fetch = d.get("FETCH_SIZE_KiB_raw_mean", 0.0) * 2.0 * 1024.0  # gfx950: raw is 1/2
            write = d.get("WRITE_SIZE_KiB_raw_mean", 0.0) * 1024.0
            d["hbm_read_bytes_per_launch"] = fetch
            d["hbm_write_bytes_per_launch"] = write
            d["hbm_bytes_per_launch"] = fetch + write
        with open(os.path.join(out_dir, "%s_hbm_traffic.json" % tag), "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res, indent=1))
        if len(sys.argv) >= 6:
            # (the one-off re-layout at construction is not a sweep, however many costs a run builds)
            sweeps = {k: d for k, d in res.items()
                      if d.get("launches_FETCH_SIZE", 0) > 2 and "relayout" not in k}
            # the sweep kernel of the run: among the kernels that move (nearly) the most bytes per launch,
            # the one launched most often (the bench also times the literal form of the same sweep a few
            # times: same bytes, a handful of launches)
            top = max(d["hbm_bytes_per_launch"] for d in sweeps.values())
            name = max((k for k, d in sweeps.items() if d["hbm_bytes_per_launch"] > 0.9 * top),
                       key=lambda k: sweeps[k]["launches_FETCH_SIZE"])
            tpath = os.path.join(out_dir, "hbm_traffic.json")
            table = json.load(open(tpath)) if os.path.exists(tpath) else {}
            table[sys.argv[5]] = res[name]["hbm_bytes_per_launch"]
            table["_source"] = ("profiles/*_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                                "passes, gfx950 x2 correction on FETCH_SIZE); bytes per launch of the sweep kernel")
            json.dump(table, open(tpath, "w"), indent=1)
            print("hbm_traffic.json:", sys.argv[5], "=", table[sys.argv[5]], "from", name)


if __name__ == "__main__":
    main()
