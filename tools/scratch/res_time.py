"""inverse / forward transform of L limbs, N = 2^16: two kernels against the one-launch resident kernel (timing experiments)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha
N = 65536
limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 36
mods = pha.coeff_modulus_create(N, [60] + [50] * (limbs - 1))
ctx = pha.PhantomContext(16, mods, 0)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.stack([torch.randint(0, int(m), (N,), dtype=torch.int64, device="cuda", generator=g) for m in mods])
def t(fn, reps=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
a = x.clone()
print(f"limbs {limbs} RESIDENT={os.environ.get('PHA_RESIDENT_NTT','1')} FLAGS={os.environ.get('PHA_RESIDENT_FLAGS','0')}: "
      f"inverse {t(lambda: ctx.nwt_2d_radix8_backward_inplace(a, limbs, 0)):.1f} us  forward {t(lambda: ctx.nwt_2d_radix8_forward_inplace(a, limbs, 0)):.1f} us", flush=True)
