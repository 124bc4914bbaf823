#!/bin/bash
# ab_hm.sh <rounds> lib...: HMult + relin / rotate / int-path A/B of library builds under tools/scratch
cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
r=$1; shift
for i in $(seq 1 $r); do for v in "$@"; do
  cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
  echo "$v $(timeout -k 10 100 python3 tools/scratch/hmult_loop.py 2>/dev/null | tr '\n' ' ')"
done; done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
