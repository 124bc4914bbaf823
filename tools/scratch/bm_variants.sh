#!/bin/bash
# timing-only builds of bconv_mfma_kernel (PHA_BM_EXP: 1 no recombination/Barrett, 2 no MFMA, 3 no stores) under a kernel trace
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cp $R/phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
for v in 0 1 2 3; do
  cp $R/tools/scratch/lib_bm_v$v.so $R/phantom-fhe_amd/libphantom_amd.so
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_bmv$v -o bm -- python3 $R/tools/scratch/hmult_loop.py > /dev/null 2>&1 || exit 1
  echo "== variant $v"
  python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_bmv$v/bm_results.db bconv
done
cp /tmp/lib_keep.so $R/phantom-fhe_amd/libphantom_amd.so
