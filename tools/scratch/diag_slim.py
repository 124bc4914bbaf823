import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle as orc
import phantom_fhe_amd as pha
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator
from oracle.sim_evaluator import SimEvaluator
LOG_N, N = 12, 4096
for s, groups, d, e in ((2048, 3, 50, 50), (128, 2, 40, 40)):
    mods = orc.coeff_modulus_create(N, [d + 10] + [d] * 2 + [d] * groups + [e] * 9 + [min(60, d + 18)] * groups + [60])
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    sk = ctx.gen_secretkey(23, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, s)
    sim = SimEvaluator(LOG_N, [int(v) for v in mods[:-1]], s, hamming_weight=32, seed=1)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, transform="fft", c2s_groups=groups, s2c_groups=groups, slim=True)
    bsim = Bootstrapper(sim, K=12, r=3, degree=31, transform="fft", c2s_groups=groups, s2c_groups=groups, slim=True)
    rng = np.random.default_rng(3)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom - bs.pre_levels))
    cs = sim.encrypt(sim.encode(msg, 2.0 ** d, sim.bottom - bs.pre_levels))
    print(s, "input err", np.abs(ev.decrypt_decode(ct) - msg).max(), "levels", ct.level, ev.bottom, "delta_out", np.log2(bs.delta_out), np.log2(bs.delta))
    x, xs = ct, cs
    for lt, lts in zip(bs.s2c, bsim.s2c):
        x, xs = lt.apply(x), lts.apply(xs)
        a, b = ev.decrypt_decode(x), sim.decrypt_decode(xs)
        print("  after s2c group level", x.level, "scale ratio", x.scale / xs.scale, "max|slot|", np.abs(b).max(), "err", np.abs(a - b).max())
    out = bs.bootstrap(ct)
    print("  bootstrap err", np.abs(ev.decrypt_decode(out) - msg).max())
    ctx.close()
