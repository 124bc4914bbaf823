#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
for v in "$@"; do
  cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
  rm -rf gpurun_out/gaps_$v
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_$v -- python3 bench.py --no-secondary --no-cpu-baseline --steps 400 > /dev/null 2>&1 || exit 1
  echo "== $v"; python3 tools/scratch/kt_gaps.py gpurun_out/gaps_$v
  rm -rf gpurun_out/gaps_$v
done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
