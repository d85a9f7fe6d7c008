#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: per blocking call, sweep duration, sweep->finalize gap,
finalize duration, and finalize-end -> next-sweep-start gap (host turnaround + launch latency)."""
import csv, sys, statistics as st
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mopt" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [("F" if "finalize" in r["Kernel_Name"] else ("P" if "publish" in r["Kernel_Name"] else "S"),
       int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "relayout" not in r["Kernel_Name"]]
sd, g1, fd, g2 = [], [], [], []
for a, b in zip(ev, ev[1:]):
    if a[0] == "S" and b[0] == "F":
        sd.append(a[2] - a[1]); g1.append(b[1] - a[2]); fd.append(b[2] - b[1])
    if a[0] == "F" and b[0] == "S":
        g2.append(b[1] - a[2])
def q(v): 
    v = sorted(v); return "n=%d median %.2f us  p10 %.2f  p90 %.2f" % (len(v), v[len(v)//2]/1e3, v[len(v)//10]/1e3, v[9*len(v)//10]/1e3)
print("sweep duration      ", q(sd)); print("sweep->finalize gap ", q(g1)); print("finalize duration   ", q(fd)); print("finalize->next sweep", q(g2))
