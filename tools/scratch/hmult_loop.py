import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
N, size_q, size_p = 65536, 36, 12
mods = pha.coeff_modulus_create(N, [60] + [50] * (size_q - 1) + [60] * size_p)
ctx = pha.PhantomContext(16, mods, size_p)
g = torch.Generator(device="cuda").manual_seed(1)
lo = int(mods.min())
rnd = lambda *s: torch.randint(0, lo, (*s, N), dtype=torch.int64, device="cuda", generator=g)
evk = pha.EvalKey([rnd(2, size_q + size_p) for _ in range(3)])
ct1, ct2 = rnd(3, size_q), rnd(2, size_q)
for _ in range(int(os.environ.get("REPS", "60"))):
    ctx.multiply_and_relin_inplace(1, ct1, ct2, evk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 300
e0.record()
for _ in range(reps):
    ctx.multiply_and_relin_inplace(1, ct1, ct2, evk)
e1.record()
e1.synchronize()
print(f"HMult + relin {1e3 * e0.elapsed_time(e1) / reps:.1f} us", flush=True)
if os.environ.get("ONLY_HMULT"):
    sys.exit(0)
elt = pha.galois_elt_from_step(16, 1)
ct = rnd(2, size_q)
e0.record()
for _ in range(reps):
    ctx.apply_galois_inplace(1, ct, elt, evk)
e1.record()
e1.synchronize()
print(f"   rotate {1e3 * e0.elapsed_time(e1) / reps:.1f} us", flush=True)
ctx.close()
imods = pha.coeff_modulus_create(N, [60] * 36)
ictx = pha.PhantomContext(16, imods, 0)
ibuf = torch.randint(0, int(imods.min()), (2, 36, N), dtype=torch.int64, device="cuda", generator=g)
for _ in range(20):
    ictx.ntt_forward_batched(ibuf, 2, 36, 36, 0)
e0.record()
for _ in range(reps):
    ictx.ntt_forward_batched(ibuf, 2, 36, 36, 0)
e1.record()
e1.synchronize()
t = 1e3 * e0.elapsed_time(e1) / reps
print(f"   int-path forward {t:.1f} us = {72 * 1048576 / (t * 1e-6) / 8e12:.4f} of 8 TB/s", flush=True)
