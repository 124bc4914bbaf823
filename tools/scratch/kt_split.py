#!/usr/bin/env python3
"""kernel time by (name, grid) from a rocprofv3 results database: kt_split.py <db> <substring>"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
agg = collections.defaultdict(lambda: [0, 0.0])
for name, s, e, gx, gy, gz in db.execute("select name,start,end,grid_x,grid_y,grid_z from kernels"):
    if sys.argv[2] in name:
        k = (re.sub(r"\(.*", "", name).replace("void pha::", ""), gx // 256, gy, gz)
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
for k, v in sorted(agg.items()):
    print(k, v[0], f"{v[1] / v[0]:.2f} us")
