#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py plus separate PMC
# passes (FETCH_SIZE, WRITE_SIZE) and a counter calibration on known byte counts.
# Usage: tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_fetch" -- $ROOT/tools/ubench stream > "$OUT/cal_fetch.log" 2>&1 || { echo "calibration fetch failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_write" -- $ROOT/tools/ubench stream > "$OUT/cal_write.log" 2>&1 || { echo "calibration write failed"; exit 1; }
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
