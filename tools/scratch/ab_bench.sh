#!/bin/bash
# ab_bench.sh <rounds> lib...: headline of bench.py (whole step and the forward transform's events) per library build, alternating
cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
r=$1; shift
for i in $(seq 1 $r); do for v in "$@"; do
  cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
  echo "$i $v $(timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 3000 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(d['ms_per_step']*1e3,2), round(r['frac'],4), {k:round(v['avg_launch_ms']*1e3,2) for k,v in r['kernels'].items()})")"
done; done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
