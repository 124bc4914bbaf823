#!/usr/bin/env python3
"""Experiment harness: per-phase NTT kernel time vs batch size (limbs x polys) on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import phantom_fhe_amd as pha

N, LIMBS = 65536, 36
mods = pha.coeff_modulus_create(N, [60] + [50] * 35)
ctx = pha.PhantomContext(16, mods, 0)
maxp = 8
buf = torch.randint(0, 1 << 49, (maxp, LIMBS, N), dtype=torch.int64, device="cuda")

def t_ms(fn, reps=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

configs = [(1, 32), (1, 36), (2, 32), (2, 36), (3, 36), (4, 32), (4, 36), (8, 32), (8, 36)]
print("polys limbs  WGs   fwd_col  fwd_row  inv_row  inv_col  (us)   us/limb(fwd)")
for polys, limbs in configs:
    r = []
    for inv, ph in ((0, 0), (0, 1), (1, 1), (1, 0)):
        r.append(1e3 * t_ms(lambda: ctx.ntt_launch_phase(buf, polys, LIMBS, limbs, 0, inv, ph)))
    wgs = polys * limbs * 16
    print(f"{polys:5d} {limbs:5d} {wgs:5d}  " + "  ".join(f"{x:7.2f}" for x in r) + f"      {(r[0]+r[1])/(polys*limbs):.3f}")
