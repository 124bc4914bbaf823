import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator
n, d, e, alpha, groups = 1 << 16, 46, 58, 4, 4
q_bits = [d + 10] + [d] * 13 + [d] * groups + [e] * 10 + [60] * groups
while len(q_bits) % alpha: q_bits.insert(1, d)
mods = pha.coeff_modulus_create(n, q_bits + [60] * alpha)
ctx = pha.PhantomContext(16, mods, alpha)
sk = ctx.gen_secretkey(21, hamming_weight=192)
ev = GpuEvaluator(pha, ctx, sk, 32768)
bs = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=groups, s2c_groups=groups)
msg = np.random.default_rng(2).uniform(-1, 1, 32768) + 0j
ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom))
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T()
    t = ev.mod_raise(ev.drop(ct, ev.bottom)); t.scale = float(ev.q[0])
    t1 = T()
    half = t
    for lt in bs.c2s: half = lt.apply(half)
    t2 = T()
    cj = ev.conj(half); re = ev.add(half, cj); im = ev.neg(ev.mul_i(ev.sub(half, cj)))
    t3 = T()
    a = bs.eval_mod(re)
    t4 = T()
    b = bs.eval_mod(im)
    t5 = T()
    v = ev.add(a, ev.mul_i(b))
    for lt in bs.s2c: v = lt.apply(v)
    t6 = T()
    print("modraise %.2f  c2s %.2f  split %.2f  evalmod %.2f + %.2f  s2c %.2f  total %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)), flush=True)
print("levels: c2s", [lt.level for lt in bs.c2s], "evalmod", bs.evalmod_level, "s2c", [lt.level for lt in bs.s2c], "limbs top", ev.limbs(1))
