#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile_gpu.sh) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(f)):
        print("  %-70s calls=%s avg_ns=%s total_ns=%s pct=%s" % (
            row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"),
            row.get("Percentage")))
f = find("trace", "*kernel_trace.csv")
if f:
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                      row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                                      row.get("Grid_Size"), row.get("Workgroup_Size")))
    print("== kernel trace: per-kernel durations (steady state = last 20 launches) ==")
    for k, v in d.items():
        last = [x[0] for x in v[-20:]]
        print("  %-60s n=%d avg_us(last20)=%.2f min_us=%.2f vgpr=%s sgpr=%s lds=%s grid=%s wg=%s" % (
            k[:60], len(v), sum(last) / len(last) / 1e3, min(last) / 1e3, v[-1][1], v[-1][2], v[-1][3], v[-1][4], v[-1][5]))
for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== %s (per dispatch, mean over last 20 dispatches) ==" % sub)
    for k, cs in acc.items():
        if "tstage" not in k:
            continue
        for c, vals in cs.items():
            v = vals[-20:]
            print("  %-40s %-24s %.4g" % (k[:40], c, sum(v) / len(v)))
