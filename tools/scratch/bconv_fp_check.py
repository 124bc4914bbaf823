import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 1:
    for v in ("0", "1"):
        subprocess.run([sys.executable, __file__, v], env=dict(os.environ, PHA_BCONV_FP64=v))
    sys.exit(0)
import torch
import phantom_fhe_amd as pha
N = 65536
mods = pha.coeff_modulus_create(N, [49] * 36 + [50] * 12)
ctx = pha.PhantomContext(16, mods, 12)
g = torch.Generator(device="cuda").manual_seed(1)
lo = int(mods.min())
c2 = torch.randint(0, lo, (36, N), dtype=torch.int64, device="cuda", generator=g)
t = torch.empty((3, 48, N), dtype=torch.int64, device="cuda")
for _ in range(5):
    ctx.modup(1, t, c2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ctx.modup(1, t, c2)
e1.record(); e1.synchronize()
print(f"PHA_BCONV_FP64={sys.argv[1]}: modup (INTT + conversion + NTT) {1e3 * e0.elapsed_time(e1) / 200:.1f} us, checksum {int(t.sum().item()) & 0xffffffff:x}", flush=True)
