#!/bin/bash
# bconv_mfma_kernel launch-shape sweep under a kernel trace: PHA_BCONV_MFMA_PAIRS x PHA_BCONV_MFMA_WGS
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pr in ${SWEEP_PAIRS:-2}; do
for w in ${SWEEP_WGS:-512 1024 2048 4096}; do
  export PHA_BCONV_MFMA_PAIRS=$pr PHA_BCONV_MFMA_WGS=$w
  rm -rf $R/gpurun_out/kt_sw
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_sw -o bm -- python3 $R/tools/scratch/hmult_loop.py > /tmp/hm.log 2>/dev/null || exit 1
  echo "== pairs $pr wgs $w: $(grep HMult -A1 /tmp/hm.log | tr '\n' ' ')"
  python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_sw/bm_results.db bconv
done
done
rm -rf $R/gpurun_out/kt_sw
