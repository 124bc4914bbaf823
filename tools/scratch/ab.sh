#!/bin/bash
# A/B of two builds of the library on the bench legs: tools/scratch/ab.sh <libA> <libB> <rounds>
set -e
for r in $(seq 1 $3); do
  for v in $1 $2; do
    cp tools/scratch/$v phantom-fhe_amd/libphantom_amd.so
    a=$(timeout -k 10 300 python3 bench.py --steps 2000 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['roofline']['frac'],4))")
    b=$(timeout -k 10 300 python3 bench.py --config 3 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4))")
    echo "$r $v config2: $a  config3_ms: $b"
  done
done
