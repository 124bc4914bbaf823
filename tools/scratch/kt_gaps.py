#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous kernel -> next kernel)."""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].split("::")[-1][:28]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 20000: gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(g)
for r in rows: durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    print(f"{k[0]:>28} -> {k[1]:<28} n={len(v):5d} median gap {statistics.median(v)/1e3:6.2f} us  mean {statistics.mean(v)/1e3:6.2f}")
for k, v in sorted(durs.items(), key=lambda kv: -len(kv[1]))[:6]:
    print(f"{k:>28} n={len(v):5d} median {statistics.median(v)/1e3:6.2f} us mean {statistics.mean(v)/1e3:6.2f}")
