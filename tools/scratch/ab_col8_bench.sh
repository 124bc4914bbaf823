#!/bin/bash
# whole-bench A/B of PHA_COL8 (library = the 6-waves build): headline, int_path, HMult + relin
cd $GRAFT_REPO_ROOT || exit 1
cp tools/scratch/lib_col8w6.so phantom-fhe_amd/libphantom_amd.so
for i in 1 2; do
for v in 0 1; do
  export PHA_COL8=$v
  h=$(timeout -k 10 250 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(r['frac'],4), 'int', round(r['int_path']['frac'],4), {k:round(v['avg_launch_ms']*1e3,2) for k,v in r['kernels'].items()}, {k:v for k,v in d.items() if 'ms' in k and k!='ms_per_step'})") || exit 1
  echo "col8=$v $h $(timeout -k 10 100 python3 tools/scratch/hmult_loop.py 2>/dev/null | tr '\n' ' ')"
done
done
