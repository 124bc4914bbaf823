#!/bin/bash
# Counter passes over the base-conversion kernel inside HMult + relinearise (tools/scratch/hmult_loop.py, run on the GPU
# box via gpurun): SQ issue / wait counters, MFMA instruction counts, fabric bytes.  Usage: tools/prof_bconv.sh <tag>
set -u
TAG=${1:-bconv}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bconv_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ONLY_HMULT=1 REPS=5
RUN="python3 $ROOT/tools/scratch/hmult_loop.py"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/p1" -- $RUN > "$OUT/p1.log" 2>&1 || { echo "pass 1 failed"; tail -5 "$OUT/p1.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d "$OUT/p2" -- $RUN > "$OUT/p2.log" 2>&1 || { echo "pass 2 failed"; tail -5 "$OUT/p2.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/p3" -- $RUN > "$OUT/p3.log" 2>&1 || { echo "pass 3 failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/p4" -- $RUN > "$OUT/p4.log" 2>&1 || { echo "pass 4 failed"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "bconv" not in k and "inner_prod" not in k and "tensor" not in k: continue
        short = k.split("(")[0].split("::")[-1][:28] + " grid " + row.get("Grid_Size", "?")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("FETCH_SIZE reads 1/2 on gfx950 (x2 for bytes); *_SIZE in KB")
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
