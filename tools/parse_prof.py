#!/usr/bin/env python3
"""Summarise a tools/profile.sh run: copies the rocprofv3 kernel stats into profiles/ and turns
the FETCH_SIZE / WRITE_SIZE passes into per-launch HBM-side bytes per kernel, corrected with the
calibration pass on known byte counts (MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE
under-reports wide coalesced reads; calibrate in your own access pattern).
Usage: tools/parse_prof.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    assert hits, pattern
    return hits[0]


def counter_avg(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


shutil.copy(one("kt/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
fetch = counter_avg(one("pmc_fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
write = counter_avg(one("pmc_write/*/*_counter_collection.csv"), "WRITE_SIZE")
cal_f = counter_avg(one("cal_fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
cal_w = counter_avg(one("cal_write/*/*_counter_collection.csv"), "WRITE_SIZE")
KNOWN = 512 << 20
cal = {}
for k in cal_f:
    width = "8B" if "rw8" in k else "16B"
    if "stream_rw" in k:
        cal[width] = {"fetch_factor": KNOWN / (cal_f[k] * 1024), "write_factor": KNOWN / (cal_w[k] * 1024)}
short = {"fntt_row_kernel": "fwd_row", "fntt_col_kernel": "fwd_col", "intt_row_kernel": "inv_row",
         "intt_col_kernel": "inv_col"}
# which calibration applies: column kernels and the row kernel's strided side use 8-B accesses,
# the row kernels' coalesced side 16-B; use the 8-B factor for reads of fwd (8 B loads) and the
# 16-B factor where the kernel reads 16 B per lane (inv_row); writes likewise.
read_width = {"fwd_row": "8B", "fwd_col": "8B", "inv_row": "16B", "inv_col": "8B"}
write_width = {"fwd_row": "16B", "fwd_col": "8B", "inv_row": "8B", "inv_col": "8B"}
summary = {"tag": tag, "calibration": cal, "kernels": {}}
latest = {}
for name, f_kb in fetch.items():
    key = next((v for k, v in short.items() if k in name), None)
    if not key:
        continue
    w_kb = write.get(name, 0.0)
    rb = f_kb * 1024 * cal[read_width[key]]["fetch_factor"]
    wb = w_kb * 1024 * cal[write_width[key]]["write_factor"]
    summary["kernels"][key] = {"kernel": name.split("(")[0], "FETCH_SIZE_raw_KB": f_kb, "WRITE_SIZE_raw_KB": w_kb,
                               "read_bytes_corrected": rb, "write_bytes_corrected": wb, "hbm_bytes_per_launch": rb + wb}
    latest[key] = rb + wb
# stamp both files with what they were measured on: bench.py reports `traffic` only when the hash of csrc/ matches
sys.path.insert(0, ROOT)
import bench  # noqa: E402
head = os.popen(f"git -C {ROOT} rev-parse --short HEAD 2>/dev/null").read().strip()
for d in (summary, latest):
    d["source_hash"] = bench.kernel_source_hash()
    d["git_head_at_parse"] = head
    d["tag"] = tag
json.dump(summary, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
json.dump(latest, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
