#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of bench.py's lane coder mode: when the back kernels (cdf, chain,
size/offsets/write) and the front kernels (T-stage, PVQ, count/scan/scatter) of every call ran.
usage: python tools/lc_timeline.py <t_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
t0 = ev[0][0]
key = lambda n: ("chain" if "lc_chain" in n else "cdf" if "lc_cdf" in n else "size" if "lc_size" in n else
                 "write" if "lc_write" in n else "offsets" if "lc_offsets" in n else
                 "front" if any(k in n for k in ("tstage", "pvq", "lc_count", "lc_scan", "lc_scatter")) else "other")
# front kernels: merge into bursts (gap < 20 ms)
bursts = []
for s, e, n in ev:
    k = key(n)
    if k == "front":
        if bursts and s - bursts[-1][1] < 20e6:
            bursts[-1][1] = max(bursts[-1][1], e); bursts[-1][2] += e - s; bursts[-1][3] += 1
        else:
            bursts.append([s, e, e - s, 1])
    elif k in ("chain", "cdf", "write"):
        print("%-6s %9.1f -> %9.1f ms  (%.1f ms)" % (k, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
for s, e, busy, cnt in bursts:
    print("front  %9.1f -> %9.1f ms  (%.1f ms wall, %.1f ms of kernel time, %d launches)" %
          ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, busy / 1e6, cnt))
