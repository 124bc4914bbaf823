import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
tag = sys.argv[1]
log_n, n = 16, 65536
q_bits, p_bits = [60] + [50] * 35, [60] * 12
mods = pha.coeff_modulus_create(n, q_bits + p_bits)
def uniform(rows, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.stack([torch.randint(0, int(m), (n,), dtype=torch.int64, device="cuda", generator=g) for m in rows])
for mf in (0, 1):
    os.environ["PHA_BCONV_MFMA"] = str(mf)
    ctx = pha.PhantomContext(log_n, mods, 12)
    ct = torch.stack([uniform(mods[:36], 2), uniform(mods[:36], 3)])
    evk = pha.EvalKey([torch.stack([uniform(mods, 10 + 2 * d), uniform(mods, 11 + 2 * d)]) for d in range(3)])
    for rep in range(2):
        out = ctx.multiply_relin_rescale(1, ct.clone(), torch.flip(ct, [0]).contiguous(), evk)
        torch.cuda.synchronize()
        np.save(f"gpurun_out/merged_{tag}_mf{mf}_r{rep}.npy", out.cpu().numpy())
    ctx.close()
