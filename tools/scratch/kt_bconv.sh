#!/bin/bash
# bconv kernel durations under rocprofv3 for PHA_BCONV_FP64=0/1 (tools/scratch/hmult_loop.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export PHA_BCONV_FP64=$v
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_fp$v -o fp -- python3 $R/tools/scratch/hmult_loop.py > /dev/null 2>&1 || exit 1
  echo "== PHA_BCONV_FP64=$v"
  python3 $R/tools/scratch/kt_summary.py $R/gpurun_out/kt_fp$v/fp_results.db 360 | head -12
done
