# groups the key-switch inner-product launches of a rocprofv3 kernel trace by grid size (= limbs): hoisted rotations
# gather their digits through the Galois permutation, relinearisations stream them
import csv, sys, collections
agg = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "key_switch_inner_prod" in n or "multi_mac_qp" in n or "lift_qp" in n:
            agg[(n.split("(")[0][-40:], int(r["Grid_Size_X"]) // 256 if "Grid_Size_X" in r else int(r["Grid_Size"]) // 256)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    v = sorted(agg[k])
    print(k, "calls", len(v), "median us", v[len(v) // 2] / 1e3, "min", v[0] / 1e3, "max", v[-1] / 1e3)
