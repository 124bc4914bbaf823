#!/bin/bash
# bconv kernel durations (rocprofv3 kernel trace of tools/scratch/hmult_loop.py) for the timing-experiment builds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  cp $R/tools/scratch/lib_bc$v.so $R/phantom-fhe_amd/libphantom_amd.so
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_bc$v -o bc -- python3 $R/tools/scratch/hmult_loop.py > /dev/null 2>&1 || exit 1
  echo "== variant $v (0 shipped, 1 no arithmetic, 2 no stores, 3 no Barrett)"
  python3 - $R/gpurun_out/kt_bc$v/bc_results.db <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
agg = collections.defaultdict(list)
for name, s, e, gy, gz in db.execute("select name,start,end,grid_y,grid_z from kernels"):
    if "bconv" in name:
        agg[(gy, gz)].append((e - s) / 1e3)
for k, v in sorted(agg.items()):
    print("   grid y,z", k, "calls", len(v), "avg %.1f us min %.1f" % (sum(v) / len(v), min(v)))
PY
done
