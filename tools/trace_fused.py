#!/usr/bin/env python3
"""Per-workgroup, per-task timeline of the single-launch forward transform (csrc/ntt_fused.cuh).
Needs the tracing build of the library:
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPHA_TRACE phantom-fhe_amd/csrc/phantom_amd.hip \\
          phantom-fhe_amd/csrc/numth.cpp -o tools/scratch/libphantom_trace.so
usage: python3 tools/trace_fused.py [--polys 2] [--flags 0] [--out gpurun_out/trace_fused.npy]"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import phantom_fhe_amd as pha
pha.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "libphantom_trace.so")
ap = argparse.ArgumentParser()
ap.add_argument("--polys", type=int, default=2)
ap.add_argument("--flags", default="0")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace_fused.npy"))
a = ap.parse_args()
L, N = 36, 65536
mods = pha.coeff_modulus_create(N, [60] + [50] * 35)
ctx = pha.PhantomContext(16, mods, 0)
lib = pha.lib()
lib.pha_set_trace.argtypes = [ctypes.c_void_p]
lib.pha_set_trace.restype = ctypes.c_int
q = torch.tensor([int(m) for m in mods], dtype=torch.int64, device="cuda")
x = torch.randint(0, 1 << 62, (a.polys, L, N), dtype=torch.int64, device="cuda") % q[None, :, None]
ctx.ntt_fused_config(1, int(a.flags, 0), 1)
fwd = lambda: ctx.ntt_forward_batched(x, a.polys, L, L)
for _ in range(20):
    fwd()
torch.cuda.synchronize()
NWG = 512
buf = torch.zeros(NWG * 256, dtype=torch.int64, device="cuda")
assert lib.pha_set_trace(buf.data_ptr()) == 0
torch.cuda.synchronize()
fwd()
torch.cuda.synchronize()
assert ctx.ntt_fused_status()[0] == 0
t = buf.cpu().numpy().reshape(NWG, 256)[:, :250].reshape(NWG, 25, 10)
np.save(a.out, t)
valid = t[:, :, 1] > 0
t0 = t[:, :, 1][valid].min()
us = lambda v: (v - t0) * 0.01
kind = t[:, :, 0] >> 32
print("workgroups with tasks:", int(valid.any(axis=1).sum()), " iterations recorded:", int(valid.sum()))
print("launch span (first top -> last end barrier): %.2f us" % us(t[:, :, 5][valid].max()))
for k, name in ((0, "none (first iteration / no work)"), (1, "column"), (2, "row")):
    m = valid & (kind == k)
    if not m.any():
        continue
    print(f"--- {name}: {int(m.sum())} iterations")
    phases = [("top -> pass 1 done", 1, 2), ("counted wait + middle barrier", 2, 3), ("pass 2 + stores issued", 3, 4),
              ("end wait (next image) + barrier", 4, 5), ("whole iteration", 1, 5), ("  signal + next row tile's DMA issued", 3, 6)]
    if k == 2:
        phases += [("  second register pass", 6, 7), ("  barrier inside pass 2", 7, 8), ("  transposed read + 8 stores issued", 8, 4)]
    for label, i0, i1 in phases:
        d = (t[:, :, i1] - t[:, :, i0])[m] * 0.01
        print("  %-34s min %5.2f  med %5.2f  p90 %5.2f  max %5.2f" % ((label,) + tuple(np.percentile(d, [0, 50, 90, 100]))))
# per workgroup: tasks done and when it finished
done = np.array([us(t[w, :, 5][valid[w]].max()) if valid[w].any() else 0 for w in range(NWG)])
ntask = (valid & (kind > 0)).sum(axis=1)
print("tasks per workgroup: min %d med %d max %d" % (ntask.min(), np.median(ntask), ntask.max()))
print("workgroup finish time: min %.2f med %.2f p90 %.2f max %.2f" % tuple(np.percentile(done[done > 0], [0, 50, 90, 100])))
# timeline of one XCD (blocks b with b % 8 == 0): first 12 workgroups
for w in list(range(0, 96, 8)) + list(range(256, 352, 8)):
    row = []
    for i in range(25):
        if not valid[w, i]:
            break
        row.append("%s%d@%.1f-%.1f" % (" CR"[int(kind[w, i]) & 3], (int(t[w, i, 0]) >> 8) & 0xffff, us(t[w, i, 1]), us(t[w, i, 5])))
    print("wg %3d: " % w + " ".join(row))
