import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator, Ct
n, d, e, alpha, groups = 1 << 16, 46, 58, int(os.environ.get("ALPHA", "4")), 4
q_bits = [d + 10] + [d] * 13 + [d] * groups + [e] * 10 + [60] * groups
while len(q_bits) % alpha: q_bits.insert(1, d)
mods = pha.coeff_modulus_create(n, q_bits + [60] * alpha)
ctx = pha.PhantomContext(16, mods, alpha)
sk = ctx.gen_secretkey(21, hamming_weight=192)
ev = GpuEvaluator(pha, ctx, sk, 32768)
bs = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=groups, s2c_groups=groups)
msg = np.random.default_rng(2).uniform(-1, 1, 32768) + 0j
ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom))
for _ in range(3):
    out = bs.bootstrap(ct)
torch.cuda.synchronize()
t0 = time.perf_counter(); out = bs.bootstrap(ct); torch.cuda.synchronize(); print("eager ms", 1e3 * (time.perf_counter() - t0), flush=True)
static_in = Ct(ct.data.clone(), ct.level, ct.scale)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    bs.bootstrap(static_in)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("capturing", flush=True)
with torch.cuda.graph(g):
    static_out = bs.bootstrap(static_in)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(2): g.replay()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("graph replay ms", ts, flush=True)
err = np.abs(ev.decrypt_decode(static_out) - msg).max()
print("graph result err", err, "eager err", np.abs(ev.decrypt_decode(out) - msg).max())
