import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phantom_fhe_amd as pha, torch
from phantom_fhe_amd.bootstrap import *
from phantom_fhe_amd.ckks import GpuEvaluator
N, LOG_N = 4096, 12
s = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c2s_bits = int(sys.argv[3]) if len(sys.argv) > 3 else 58
mods = pha.coeff_modulus_create(N, [50] + [40] * 2 + [40] * groups + [40] * 9 + [c2s_bits] * groups + [60])
ctx = pha.PhantomContext(LOG_N, mods, 1)
sk = ctx.gen_secretkey(23, hamming_weight=32)
ev = GpuEvaluator(pha, ctx, sk, s)
bs = Bootstrapper(ev, K=12, r=3, degree=31, transform="fft", c2s_groups=groups, s2c_groups=groups)
rng = np.random.default_rng(3)
msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
ct = ev.encrypt(ev.encode(msg, 2.0 ** 40, ev.bottom))
q0 = ev.q[0]
t = ev.mod_raise(ev.drop(ct, ev.bottom)); t.scale = float(q0)
k = bs.S
while k < ev.n // 2:
    t = ev.add(t, ev.rotate(t, k)); k <<= 1
zt = ev.decrypt_decode(t)
print("raised slots max", np.abs(zt).max())
bits = s.bit_length() - 1
perm = np.array([int(format(i, "0%db" % bits)[::-1], 2) for i in range(s)])
A = slot_matrix(s)
u = np.linalg.solve(A, zt)
norm = 1.0 / (bs.g * bs.Kb * 2.0)
half = t
x = zt
def dense(g):
    j = np.arange(s); d = np.zeros((s, s), dtype=complex)
    for off, v in g.items(): d[j, (j + off) % s] += v
    return d
inv = special_fft_groups(s, groups, True)
for i, lt in enumerate(bs.c2s):
    half = lt.apply(half)
    x = dense(inv[i]) @ x * norm ** (1.0 / groups)
    got = ev.decrypt_decode(half)
    print("c2s group", i, "scale", half.scale, "err", np.abs(got - x * (q0 / half.scale if False else 1)).max(), "mag", np.abs(x).max())
print("c2s total vs ideal", np.abs(ev.decrypt_decode(half) * 1.0 - u[perm] * norm * (q0 / bs.delta) * (bs.delta / q0)).max())
cj = ev.conj(half)
re = ev.add(half, cj)
im = ev.neg(ev.mul_i(ev.sub(half, cj)))
wre = ev.decrypt_decode(re); wim = ev.decrypt_decode(im)
print("re/im imag parts", np.abs(wre.imag).max(), np.abs(wim.imag).max(), "max |w|", np.abs(wre.real).max(), np.abs(wim.real).max())
ere = bs.eval_mod(re); eim = bs.eval_mod(im)
f = lambda w: np.sin(2 * np.pi * bs.Kb * w)
print("evalmod err re", np.abs(ev.decrypt_decode(ere) - f(wre.real)).max(), "im", np.abs(ev.decrypt_decode(eim) - f(wim.real)).max())
v = ev.add(ere, ev.mul_i(eim))
vv = ev.decrypt_decode(v)
out = v
y = vv
fwd = special_fft_groups(s, groups, False)
cst = ((q0 / bs.delta) / (2 * np.pi)) ** (1.0 / groups)
for i, lt in enumerate(bs.s2c):
    out = lt.apply(out)
    y = dense(fwd[i]) @ y * cst
    print("s2c group", i, "err", np.abs(ev.decrypt_decode(out) - y).max(), "mag", np.abs(y).max())
print("final err", np.abs(ev.decrypt_decode(out) - msg).max())
ctx.close()
