#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (rocpd sqlite): kt_summary.py <db> [calls-per-step]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in db.execute("select name, start, end from kernels"):
    short = re.sub(r"\(.*", "", name).replace("void pha::", "")
    agg[short][0] += 1
    agg[short][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':70s} {'calls/step':>10s} {'us/step':>9s} {'avg us':>8s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:70]:70s} {c / steps:10.2f} {t / steps:9.2f} {t / c:8.2f}")
print(f"{'total':70s} {sum(v[0] for v in agg.values()) / steps:10.2f} {tot / steps:9.2f}")
