"""Event-timed rescale_to_next at 36 limbs of N = 2^16 and an alpha = 1 key switch (24 + 1 limbs): the paths whose single-limb
broadcast runs inside the forward column kernel (PHA_BCAST_FUSED=0: as its own pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha
N = 65536
def timed(fn, reps=300):
    for _ in range(60): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
g = torch.Generator(device="cuda").manual_seed(1)
mods = pha.coeff_modulus_create(N, [60] + [50] * 35 + [60] * 12)
ctx = pha.PhantomContext(16, mods, 12)
rnd = lambda lo, *s: torch.randint(0, lo, (*s, N), dtype=torch.int64, device="cuda", generator=g)
src, dst = rnd(int(mods.min()), 2, 36), torch.empty((2, 35, N), dtype=torch.int64, device="cuda")
print(f"rescale 36 -> 35 limbs {timed(lambda: ctx.rescale_to_next(1, src, 2, dst)):.1f} us", flush=True)
ctx.close()
mods1 = pha.coeff_modulus_create(N, [50] * 24 + [60])
c1 = pha.PhantomContext(16, mods1, 1)
evk = pha.EvalKey([rnd(int(mods1.min()), 2, 25) for _ in range(24)])
ct, c2 = rnd(int(mods1.min()), 2, 24), rnd(int(mods1.min()), 24)
print(f"   key switch, 24 + 1 limbs (alpha = 1) {timed(lambda: c1.keyswitch_inplace(1, ct, c2, evk), 100):.1f} us", flush=True)
