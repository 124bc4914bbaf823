#!/bin/bash
# per-kernel averages of the default and the eight-per-lane column pass, FP64 chain and 60-bit chain separately
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for c in 0 1; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/col8prof$c -o col8 -- python3 tools/scratch/ab_col8.py 300 $c > gpurun_out/ab_col8_prof$c.log 2>&1 || exit 1
  echo "case $c"; grep "equal\|us per" gpurun_out/ab_col8_prof$c.log
  python3 tools/scratch/kt_split.py gpurun_out/col8prof$c/col8_results.db fntt_col
done
