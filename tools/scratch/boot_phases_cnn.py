# phase times of the CNN's sparse bootstrappings (tools/bench_resnet20.py chain: 32 + 4 limbs)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator
n, d, e, alpha = 1 << 16, 38, 48, 8
q_bits = [d + 10] + [d] * 16 + [d] * 2 + [e] * 10 + [46] * 3
mods = pha.coeff_modulus_create(n, q_bits + [50] * alpha)
ctx = pha.PhantomContext(16, mods, alpha)
sk = ctx.gen_secretkey(21, hamming_weight=192)
ev = GpuEvaluator(pha, ctx, sk, 32768)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for slots in (16384, 4096):
    bs = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=3, s2c_groups=2, slots=slots, slim=True)
    msg = np.random.default_rng(2).uniform(-1, 1, slots) + 0j
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom - 2, slots))
    for rep in range(3):
        tp = T()
        x = ev.drop(ct, bs.s2c_level)
        for lt in bs.s2c: x = lt.apply(x)
        t0 = T()
        t = ev.mod_raise(ev.drop(x, ev.bottom)); t.scale = float(ev.q[0])
        k = bs.S
        while k < ev.n // 2:
            t = ev.add(t, ev.rotate(t, k)); k <<= 1
        t1 = T()
        half = t
        cs = []
        for lt in bs.c2s:
            half = lt.apply(half); cs.append(T())
        t2 = T()
        cj = ev.conj(half); re = ev.add(half, cj); im = ev.neg(ev.mul_i(ev.sub(half, cj)))
        t3 = T()
        a, b = ev.run_pair(lambda ee: bs.eval_mod(re, ee), lambda ee: bs.eval_mod(im, ee))
        t5 = T()
        v = ev.add(a, ev.mul_i(b))
        t6 = T()
    print(slots, "s2c first %.2f  modraise+subsum %.2f  c2s %.2f %s  split %.2f  evalmod pair %.2f  combine %.2f  total %.2f ms" % (
        1e3 * (t0 - tp), 1e3 * (t1 - t0), 1e3 * (t2 - t1), [round(1e3 * (y - x), 2) for x, y in zip([t1] + cs[:-1], cs)], 1e3 * (t3 - t2), 1e3 * (t5 - t3), 1e3 * (t6 - t5), 1e3 * (t6 - tp)),
        "diagonals", [len(lt.diagonals) if hasattr(lt, "diagonals") else None for lt in bs.c2s + bs.s2c], flush=True)
