#!/bin/bash
# ab_lib_kt.sh lib...: rocprofv3 kernel averages of the four transform passes (72 limbs) per library build under tools/scratch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
for r in 1 2; do for v in "$@"; do
  cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
  rm -rf gpurun_out/abkt
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/abkt -o ab -- python3 bench.py --no-secondary --no-cpu-baseline --steps 300 > /dev/null 2>&1 || exit 1
  echo "$v: $(python3 tools/scratch/kt_split.py gpurun_out/abkt/ab_results.db ntt_ | grep "1152" | awk '{print $1, $2, $NF-1, $(NF-1)}' | tr '\n' ';')"
done; done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
