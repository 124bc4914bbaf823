#!/usr/bin/env python3
"""A fixed number of forward transforms of the configs[1] batch in ONE mode (for rocprofv3 passes):
usage: python3 tools/run_ntt_mode.py fused|two [--polys 2] [--reps 50] [--flags 0] [--chain 60,50x35]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phantom_fhe_amd as pha
ap = argparse.ArgumentParser()
ap.add_argument("mode")
ap.add_argument("--polys", type=int, default=2)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--flags", default="0")
ap.add_argument("--chain", default="60,50x35")
a = ap.parse_args()
bits = []
for tok in a.chain.split(","):
    b, _, c = tok.partition("x")
    bits += [int(b)] * (int(c) if c else 1)
N, L = 65536, len(bits)
mods = pha.coeff_modulus_create(N, bits)
ctx = pha.PhantomContext(16, mods, 0)
ctx.ntt_fused_config(a.mode == "fused", int(a.flags, 0), 1)
q = torch.tensor([int(m) for m in mods], dtype=torch.int64, device="cuda")
x = torch.randint(0, 1 << 62, (a.polys, L, N), dtype=torch.int64, device="cuda") % q[None, :, None]
for _ in range(a.reps):
    ctx.ntt_forward_batched(x, a.polys, L, L)
torch.cuda.synchronize()
print("done", a.mode, ctx.ntt_fused_status())
