import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha
from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, ThreadComm
N, size_q, size_p = 65536, 36, 12
mods = pha.coeff_modulus_create(N, [60] + [50] * (size_q - 1) + [60] * size_p)
ctx = pha.PhantomContext(16, mods, size_p)
ev = LimbParallelEvaluator(ctx, ThreadComm(ThreadComm.Shared(1), 0))
g = torch.Generator(device="cuda").manual_seed(1)
lo = int(mods.min())
rnd = lambda *s: torch.randint(0, lo, (*s, N), dtype=torch.int64, device="cuda", generator=g)
evk = pha.EvalKey([rnd(2, size_q + size_p) for _ in range(3)])
ct1, ct2 = rnd(2, size_q), rnd(2, size_q)
out3 = torch.empty((3, size_q, N), dtype=torch.int64, device="cuda")
for _ in range(int(os.environ.get("REPS", "60"))):
    ev.multiply_and_relin(1, ct1, ct2, evk, out3)
torch.cuda.synchronize()
