#!/usr/bin/env python3
"""Event-timed launches of the four transform kernels at 2 x 36 limbs of N = 2^16 (the library under test may be a timing-only build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha
N, LIMBS = 65536, 36
mods = pha.coeff_modulus_create(N, [60] + [50] * 35)
ctx = pha.PhantomContext(16, mods, 0)
buf = torch.randint(0, 1 << 49, (2, LIMBS, N), dtype=torch.int64, device="cuda")
def t_us(fn, reps=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
t_us(lambda: ctx.ntt_launch_phase(buf, 2, LIMBS, LIMBS, 0, 0, 0), 2000)   # clocks out of idle
r = [t_us(lambda: ctx.ntt_launch_phase(buf, 2, LIMBS, LIMBS, 0, inv, ph)) for inv, ph in ((0, 0), (0, 1), (1, 1), (1, 0))]
print("fwd_col %.2f  fwd_row %.2f  inv_row %.2f  inv_col %.2f us" % tuple(r))
