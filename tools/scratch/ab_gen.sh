#!/bin/bash
# A/B of the derived last-pass twiddles (PHA_GEN_TW) on HMult + relinearise and on the headline line
mkdir -p gpurun_out/r3e
for g in 1 0 1 0; do
  PHA_GEN_TW=$g timeout -k 10 200 python -u bench.py --config 3 --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r3e/c3_gen$g.json
  python3 -c "import json; d=json.load(open('gpurun_out/r3e/c3_gen$g.json')); print('GEN_TW=$g hmult_relin', d.get('value'), d.get('unit'))"
done
for g in 1 0; do
  PHA_GEN_TW=$g timeout -k 10 300 python -u bench.py --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/r3e/bench_gen$g.json
  python3 -c "import json; d=json.load(open('gpurun_out/r3e/bench_gen$g.json')); print('GEN_TW=$g', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('secondary'))"
done
