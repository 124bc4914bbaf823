#!/bin/bash
# entry / exit timeline of the forward column kernel (tools/scratch/libphantom_trace.so built with -DPHA_TRACE=3) next to the
# profiler's durations of the same launches: FP64 chain, then 60-bit chain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for int in "" 1; do
  rm -rf gpurun_out/trprof$int
  TRACE_INT=$int timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/trprof$int -o tr -- python3 tools/scratch/trace2.py > gpurun_out/trprof$int.log 2>&1 || exit 1
  echo "== chain: ${int:+36 x 60-bit (integer butterflies)}${int:-1 x 60-bit + 35 x 50-bit (FP64 butterflies)}"; grep -v "^W2\|^E2\|amdgpu.ids" gpurun_out/trprof$int.log
  python3 - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/trprof$int/tr_results.db")
rows = [r for r in db.execute("select name,start,end from kernels order by start") if "fntt_col" in r[0]]
print("profiler: untraced launches avg %.2f us; the six traced launches %s us" % (sum(r[2] - r[1] for r in rows[:20]) / 20e3, " ".join("%.2f" % ((r[2] - r[1]) / 1e3) for r in rows[-6:])))
PY
done
