#!/bin/bash
# ab_env.sh KEY=VAL [phase] [inverse]: rocprofv3 per-kernel averages of the default context and one created under KEY=VAL
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
export AB_PHASE=${2:-0} AB_INVERSE=${3:-0}
for c in 0 1; do
  rm -rf gpurun_out/abprof$c
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/abprof$c -o ab -- python3 tools/scratch/ab_col8.py 300 $c $1 > gpurun_out/ab_prof$c.log 2>&1 || exit 1
  echo "case $c"; grep "equal\|us per" gpurun_out/ab_prof$c.log
  python3 tools/scratch/kt_split.py gpurun_out/abprof$c/ab_results.db ntt_
done
