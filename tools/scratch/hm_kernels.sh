#!/bin/bash
# HMult + relinearise kernel breakdown (rocprofv3 --kernel-trace over tools/scratch/hmult_loop.py, HMult only): hm_kernels.sh <tag>
R=$GRAFT_REPO_ROOT; T=${1:-r3u}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_hm_$T
ONLY_HMULT=1 timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_hm_$T -o hm -- python3 $R/tools/scratch/hmult_loop.py > /tmp/hm.log 2>/dev/null || exit 1
grep HMult /tmp/hm.log
python3 $R/tools/scratch/kt_summary.py $R/gpurun_out/kt_hm_$T/hm_results.db 360 > $R/gpurun_out/${T}_hmult_relin_kernels_single_gpu.txt
python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_hm_$T/hm_results.db bconv >> $R/gpurun_out/${T}_hmult_relin_kernels_single_gpu.txt
cat $R/gpurun_out/${T}_hmult_relin_kernels_single_gpu.txt
