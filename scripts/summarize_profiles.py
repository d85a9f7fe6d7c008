#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/...) into the tracked summaries under profiles/.

  kernel stats : profiles/<tag>_kernel_stats.csv      rows of the --kernel-trace --stats summary
  PMC traffic  : profiles/<tag>_hbm_traffic.json      per-launch FETCH_SIZE / WRITE_SIZE of each
                 mopt kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE
                 counts 64 B per 128-B request of a 16-B-per-lane stream: x2; WRITE_SIZE exact; both
                 in KiB), collected in separate --pmc passes.
Usage: summarize_profiles.py TAG STATS_CSV [FETCH_COUNTER_CSV WRITE_COUNTER_CSV]"""
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
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "mopt" in r["Name"] or "copyBuffer" in r["Name"]
            or "rccl" in r["Name"].lower() or "nccl" in r["Name"].lower()]
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


if __name__ == "__main__":
    main()
