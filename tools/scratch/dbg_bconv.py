"""modup with the base conversion on the vector ALU and on the matrix cores, side by side: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import phantom_fhe_amd as pha
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
size_q, size_p = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 2)
N = 1 << log_n
mods = pha.coeff_modulus_create(N, [45] * size_q + [55] * size_p)
os.environ["PHA_BCONV_MFMA"] = "0"
c0 = pha.PhantomContext(log_n, mods, size_p)
os.environ["PHA_BCONV_MFMA"] = "1"
c1 = pha.PhantomContext(log_n, mods, size_p)
g = torch.Generator(device="cuda").manual_seed(1)
cks = torch.stack([torch.randint(0, int(m), (N,), dtype=torch.int64, device="cuda", generator=g) for m in mods[:size_q]])
beta = c0.beta(1)
outs = []
for c in (c0, c1):
    dst = torch.full((beta, size_q + size_p, N), -1, dtype=torch.int64, device="cuda")
    c.modup(1, dst, cks.clone())
    torch.cuda.synchronize()
    outs.append(dst.cpu().numpy().view(np.uint64))
a, b = outs
print("equal:", (a == b).all())
for d in range(beta):
    for j in range(size_q + size_p):
        ne = np.nonzero(a[d, j] != b[d, j])[0]
        if len(ne):
            print(f"digit {d} limb {j}: {len(ne)} of {N} differ; first at {ne[:8]}; alu {a[d, j, ne[0]]} mfma {b[d, j, ne[0]]} "
                  f"diff {(int(b[d, j, ne[0]]) - int(a[d, j, ne[0]])) % int(mods[j])}")
evk = pha.EvalKey([torch.stack([torch.stack([torch.randint(0, int(m), (N,), dtype=torch.int64, device="cuda", generator=g) for m in mods])
                                for _ in range(2)]) for _ in range(beta)])
ct = torch.stack([cks, cks.flip(1)])
res = []
for c in (c0, c1):
    x = ct.clone()
    c.keyswitch_inplace(1, x, cks.clone(), evk)
    torch.cuda.synchronize()
    res.append(x.cpu().numpy().view(np.uint64))
print("keyswitch equal:", (res[0] == res[1]).all())
for p in range(2):
    for j in range(size_q):
        ne = np.nonzero(res[0][p, j] != res[1][p, j])[0]
        if len(ne):
            print(f"poly {p} limb {j}: {len(ne)} differ; first at {ne[:8]}, last {ne[-1]}")
