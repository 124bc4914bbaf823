#!/usr/bin/env python3
"""Config 3 / 4 composites on one GPU: HMult+relinearize, rotate, rescale latency at N=2^16,
L=35 (36 Q limbs), dnum=3 (12 special primes), plus a per-kernel breakdown via torch profiler-free
event timing of the individual C-ABI stages."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import phantom_fhe_amd as pha

N = 65536
def t_ms(fn, reps=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

def run(size_p=12, size_q=36):
    mods = pha.coeff_modulus_create(N, [60] + [50] * (size_q - 1) + [60] * size_p)
    ctx = pha.PhantomContext(16, mods, size_p)
    size_qp = size_q + size_p
    dnum = size_q // size_p
    g = torch.Generator(device="cuda").manual_seed(1)
    dm = torch.from_numpy(mods.view(np.int64)).cuda()
    def uni(shape_limbs, idx):
        m = dm[idx].view(*([1] * (len(shape_limbs) - 1)), -1, 1)
        return torch.randint(0, 1 << 49, (*shape_limbs, N), dtype=torch.int64, device="cuda", generator=g) % m
    evk = pha.EvalKey([uni((2, size_qp), torch.arange(size_qp, device="cuda")) for _ in range(dnum)])
    q_idx = torch.arange(size_q, device="cuda")
    ct1 = uni((3, size_q), q_idx); ct2 = uni((2, size_q), q_idx)
    res = {}
    res["hmult_relin_ms"] = t_ms(lambda: ctx.multiply_and_relin_inplace(1, ct1, ct2, evk))
    res["rotate_ms"] = t_ms(lambda: ctx.apply_galois_inplace(1, ct2, 5, evk))
    dst = torch.empty((2, size_q - 1, N), dtype=torch.int64, device="cuda")
    res["rescale_ms"] = t_ms(lambda: ctx.rescale_to_next(1, ct2, 2, dst))
    beta = ctx.beta(1)
    t_mod_up = torch.empty((beta, size_qp, N), dtype=torch.int64, device="cuda")
    cx = torch.empty((2, size_qp, N), dtype=torch.int64, device="cuda")
    c2 = uni((size_q,), q_idx)
    res["stage_modup_ms"] = t_ms(lambda: ctx.modup(1, t_mod_up, c2))
    res["stage_inner_prod_ms"] = t_ms(lambda: ctx.key_switch_inner_prod(1, cx, t_mod_up, evk))
    res["stage_moddown_x2_ms"] = t_ms(lambda: (ctx.moddown_from_NTT(1, cx[0], cx[0]), ctx.moddown_from_NTT(1, cx[1], cx[1])))
    res["stage_tensor_ms"] = t_ms(lambda: ctx.tensor_prod_2x2(ct1, ct2, size_q, ct1))
    res["stage_galois_ms"] = t_ms(lambda: ctx.apply_galois_ntt(ct2, 2 * size_q, 5, cx))
    # 8 rotations of one ciphertext: one by one vs hoisted (shared modup)
    elts = [pha.galois_elt_from_step(16, st) for st in range(1, 9)]
    keys = [evk] * 8
    res["rotate_x8_ms"] = t_ms(lambda: [ctx.apply_galois_inplace(1, ct2, e, evk) for e in elts], reps=10)
    res["rotate_x8_hoisted_ms"] = t_ms(lambda: ctx.rotate_hoisted(1, ct2, elts, keys), reps=10)
    res["config"] = {"N": N, "size_Q": size_q, "size_P": size_p, "dnum": dnum}
    ctx.close()
    return res

if __name__ == "__main__":
    out = {"config3_dnum3": run(12, 36)}
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        out["alpha1_31limbs"] = run(1, 31)
    print(json.dumps(out, indent=1))
