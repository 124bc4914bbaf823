"""Determinism stress: HMult + relinearise, rotation, merged multiply-relin-rescale and rescale at configs[2], 40 runs each,
every result compared with the first (any timing-dependent hazard shows up as a differing word)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha
n, log_n = 65536, 16
mods = pha.coeff_modulus_create(n, [60] + [50] * 35 + [60] * 12)
ctx = pha.PhantomContext(log_n, mods, 12)
def uniform(rows, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.stack([torch.randint(0, int(m), (n,), dtype=torch.int64, device="cuda", generator=g) for m in rows])
ct = torch.stack([uniform(mods[:36], 2), uniform(mods[:36], 3)])
ct3 = torch.cat([ct, torch.zeros_like(ct[:1])])
evk = pha.EvalKey([torch.stack([uniform(mods, 10 + 2 * d), uniform(mods, 11 + 2 * d)]) for d in range(3)])
elt = pha.galois_elt_from_step(log_n, 1)
first, bad = {}, {}
for rep in range(40):
    a = ct3.clone(); ctx.multiply_and_relin_inplace(1, a, torch.flip(ct, [0]).contiguous(), evk)
    b = ct.clone(); ctx.apply_galois_inplace(1, b, elt, evk)
    c = ctx.multiply_relin_rescale(1, ct.clone(), torch.flip(ct, [0]).contiguous(), evk)
    d = torch.zeros(2, 35, n, dtype=torch.int64, device="cuda"); ctx.rescale_to_next(1, ct.clone(), 2, d)
    for k, v in (("hmult", a[:2]), ("rotate", b), ("merged", c), ("rescale", d)):
        if rep == 0:
            first[k] = v.clone()
        else:
            bad[k] = bad.get(k, 0) + int((v != first[k]).sum())
torch.cuda.synchronize()
print("differing words over 39 repeats:", bad)
assert not any(bad.values())
