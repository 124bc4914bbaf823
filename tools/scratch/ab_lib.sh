#!/bin/bash
# same-box A/B of library builds: ab_lib.sh <rounds> <lib>...   (headline limb-transforms/s and HMult + relinearise / rotate)
R=$GRAFT_REPO_ROOT
cd $R
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
r=$1; shift
for i in $(seq 1 $r); do
  for v in "$@"; do
    cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
    h=$(timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(r['frac'],4), {k:round(v['avg_launch_ms']*1e3,2) for k,v in r['kernels'].items()})")
    echo "$i $v $h $(timeout -k 10 100 python3 tools/scratch/hmult_loop.py 2>/dev/null | tr '\n' ' ')"
  done
done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
