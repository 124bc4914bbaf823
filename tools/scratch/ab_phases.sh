#!/bin/bash
# ab_phases.sh <rounds> lib...: event-timed transform kernels per library build under tools/scratch
cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
r=$1; shift
for i in $(seq 1 $r); do for v in "$@"; do
  cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
  echo "$i $v $(timeout -k 10 100 python3 tools/scratch/phase_time.py 2>/dev/null | tail -1)"
done; done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
