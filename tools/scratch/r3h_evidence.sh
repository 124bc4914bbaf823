#!/bin/bash
# round-3 evidence: HMult + relinearise kernel breakdown (HMult only) and the single-launch transform against two kernels
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ONLY_HMULT=1 timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_hm_r3h -o hm -- python3 $R/tools/scratch/hmult_loop.py > /tmp/hm.log 2>/dev/null || exit 1
grep HMult /tmp/hm.log
python3 $R/tools/scratch/kt_summary.py $R/gpurun_out/kt_hm_r3h/hm_results.db 360 > $R/gpurun_out/r3h_hmult_relin_kernels_single_gpu.txt
python3 $R/tools/scratch/kt_split.py $R/gpurun_out/kt_hm_r3h/hm_results.db bconv >> $R/gpurun_out/r3h_hmult_relin_kernels_single_gpu.txt
cat $R/gpurun_out/r3h_hmult_relin_kernels_single_gpu.txt
cd $R
timeout -k 10 300 python3 tools/exp_fused.py --shapes 2x36,3x36,4x36,8x36,16x36 --flags 0,2 --rounds 3 --reps 100 2>/dev/null > gpurun_out/r3h_fused_vs_two_kernel.log || exit 1
cut -c1-300 gpurun_out/r3h_fused_vs_two_kernel.log
