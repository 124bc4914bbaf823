#!/bin/bash
# rocprofv3 kernel statistics of bench.py --config 4 (rotate x 64, batches of 16)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/kt_c4
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_c4 -- python3 bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt_c4/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{r["Name"].split("(")[0][-58:]:>58} calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
