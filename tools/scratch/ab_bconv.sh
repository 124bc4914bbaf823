#!/bin/bash
# A/B of the matrix-core base conversion (PHA_BCONV_MFMA=0/1) on HMult + relinearise (tools/scratch/hmult_loop.py), then a kernel trace
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
  for v in 0 1; do
    echo "$i PHA_BCONV_MFMA=$v $(PHA_BCONV_MFMA=$v timeout -k 10 120 python3 tools/scratch/hmult_loop.py 2>/dev/null | grep -v amdgpu | tr '\n' ' ')" || exit 1
  done
done
for w in 512 1024 1536 3072 4608 6912; do
  echo "PHA_BCONV_MFMA_WGS=$w $(PHA_BCONV_MFMA_WGS=$w timeout -k 10 120 python3 tools/scratch/hmult_loop.py 2>/dev/null | grep -v amdgpu | tr '\n' ' ')" || exit 1
done
cd /tmp && export TMPDIR=/tmp
for w in 512 2048 4608; do
export PHA_BCONV_MFMA_WGS=$w
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_bm$w -o bm -- python3 $R/tools/scratch/hmult_loop.py > /dev/null 2>&1 || exit 1
echo "== WGS $w"
python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_bm$w/bm_results.db bconv
python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_bm$w/bm_results.db inner_prod
done
python3 $R/tools/scratch/kt_summary.py $R/gpurun_out/kt_bm2048/bm_results.db 360 | head -14
