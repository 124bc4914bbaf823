#!/bin/bash
# ab_lib_env.sh <rounds> lib[:KEY=VAL] ...  -- same-box A/B of library builds (tools/scratch/<lib>) with an optional env setting
cd $GRAFT_REPO_ROOT || exit 1
cp phantom-fhe_amd/libphantom_amd.so /tmp/lib_keep.so
r=$1; shift
for i in $(seq 1 $r); do
for spec in "$@"; do
  lib=${spec%%:*}; env=""; [ "$spec" != "$lib" ] && env=${spec#*:}
  cp tools/scratch/$lib phantom-fhe_amd/libphantom_amd.so
  h=$( ( [ -n "$env" ] && export $env; timeout -k 10 250 python bench.py --no-cpu-baseline --no-secondary --steps 300 2>/dev/null ) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(r['frac'],4), {k:round(v['avg_launch_ms']*1e3,2) for k,v in r['kernels'].items()})") || exit 1
  i2=$( ( [ -n "$env" ] && export $env; timeout -k 10 100 python3 tools/scratch/hmult_loop.py 2>/dev/null ) | tr '\n' ' ')
  echo "$spec $h $i2"
done
done
cp /tmp/lib_keep.so phantom-fhe_amd/libphantom_amd.so
