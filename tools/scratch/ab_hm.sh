#!/bin/bash
# A/B of library builds on HMult + relinearise / rotate (tools/scratch/hmult_loop.py): ab_hm.sh <rounds> <lib>...
set -e
r=$1; shift
for i in $(seq 1 $r); do
  for v in "$@"; do
    cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
    echo "$i $v $(python3 tools/scratch/hmult_loop.py 2>/dev/null | grep -v amdgpu | tr '\n' ' ')"
  done
done
