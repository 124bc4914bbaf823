#!/bin/bash
# ab_envbench.sh <rounds> KEY=VAL...: whole-bench A/B of environment settings (first = baseline "-")
cd $GRAFT_REPO_ROOT || exit 1
r=$1; shift
for i in $(seq 1 $r); do
for v in - "$@"; do
  h=$( ( [ "$v" != "-" ] && export $v; timeout -k 10 250 python bench.py --no-cpu-baseline --no-secondary --steps 300 2>/dev/null ) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(r['frac'],4), {k:round(v['avg_launch_ms']*1e3,2) for k,v in r['kernels'].items()})") || exit 1
  i2=$( ( [ "$v" != "-" ] && export $v; timeout -k 10 100 python3 tools/scratch/hmult_loop.py 2>/dev/null ) | tr '\n' ' ')
  echo "$v $h $i2"
done
done
