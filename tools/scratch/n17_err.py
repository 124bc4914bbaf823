import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, phantom_fhe_amd as pha
for LOG_N, special in ((16, 2), (17, 2), (17, 1)):
    N = 1 << LOG_N
    mods = pha.coeff_modulus_create(N, [60, 40, 40, 40] + [60] * special)
    ctx = pha.PhantomContext(LOG_N, mods, special)
    sk = ctx.gen_secretkey(5); rlk = ctx.gen_relinkey(sk, 6)
    s = N // 2
    rng = np.random.default_rng(3)
    z1 = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s); z2 = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    scale = 2.0 ** 40
    ct1 = ctx.encrypt_symmetric(sk, ctx.encode(z1, scale, 1, s), 1, 11, 12)
    ct2 = ctx.encrypt_symmetric(sk, ctx.encode(z2, scale, 1, s), 1, 13, 14)
    e0 = np.abs(ctx.decode(ctx.ckks_decrypt(sk, ct1, 1), scale, s, 1) - z1)
    prod = torch.cat([ct1, torch.zeros_like(ct1[:1])])
    ctx.multiply_and_relin_inplace(1, prod, ct2, rlk)
    e1 = np.abs(ctx.decode(ctx.ckks_decrypt(sk, prod[:2].contiguous(), 1), scale * scale, s, 1) - z1 * z2)
    res = torch.empty((2, 3, N), dtype=torch.int64, device="cuda")
    ctx.rescale_to_next(1, prod[:2].contiguous(), 2, res)
    e2 = np.abs(ctx.decode(ctx.ckks_decrypt(sk, res, 2), scale * scale / float(mods[3]), s, 2) - z1 * z2)
    for name, e in (("fresh", e0), ("mul+relin", e1), ("rescaled", e2)):
        top = np.argsort(e)[-5:][::-1]
        print(LOG_N, special, name, "max %.3g median %.3g" % (e.max(), np.median(e)), "top idx", list(top), ["%.2g" % v for v in e[top]])
    # plaintext-only encode/decode of the product at scale^2 to separate encoder precision
    p = ctx.encode(z1 * z2, scale * scale, 1, s)
    e3 = np.abs(ctx.decode(p, scale * scale, s, 1) - z1 * z2)
    print(LOG_N, "encode/decode at scale^2: max %.3g median %.3g" % (e3.max(), np.median(e3)))
    ctx.close()
