"""Per-workgroup phase timestamps of the forward column kernel (library built with -DPHA_TRACE).
   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPHA_TRACE csrc/phantom_amd.hip csrc/numth.cpp -o tools/scratch/libphantom_trace.so"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import phantom_fhe_amd as pha
pha.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "libphantom_trace.so")
L, N = 36, 65536
mods = pha.coeff_modulus_create(N, [60] + [50] * 34 + [60] + [60] * 12)
ctx = pha.PhantomContext(16, mods, 12)
lib = pha.lib()
lib.pha_set_trace.argtypes = [ctypes.c_void_p]
lib.pha_set_trace.restype = ctypes.c_int
x = torch.randint(0, 1 << 49, (2, L, N), dtype=torch.int64, device="cuda")
nwg = 2 * L * (N // int(os.environ.get("TILE", "2048")))
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
def fwd():
    ctx.ntt_launch_phase(x, 2, L, L, 0, 0, 0)      # forward, column phase
fwd()
torch.cuda.synchronize()
for _ in range(10): fwd()
torch.cuda.synchronize()
# timing loop (no tracing)
import time
for ph, name in ((0, "col"), (1, "row")):
    for inv in (0, 1):
        f = lambda: ctx.ntt_launch_phase(x, 2, L, L, 0, inv, ph)
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); print("phase", name, "inverse" if inv else "forward", "%.2f us" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
assert lib.pha_set_trace(buf.data_ptr()) == 0
torch.cuda.synchronize()
fwd()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8)[:, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) * 0.01      # 100 MHz -> us
np.save(os.path.join(ROOT, "gpurun_out", "trace_col.npy"), t)
order = np.argsort(t[:, 0])
print("WGs", nwg, "kernel span us", t[:, 4].max())
print("start:   min %.2f  med %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(t[:, 0], [0, 50, 90, 100])))
for name, a, b in (("load+pass1", 0, 1), ("barrier", 1, 2), ("pass2+store issue", 2, 3), ("store drain", 3, 4), ("total", 0, 4)):
    d = t[:, b] - t[:, a]
    print("%-18s min %.2f  med %.2f  p90 %.2f  max %.2f" % ((name,) + tuple(np.percentile(d, [0, 50, 90, 100]))))
print("end:     min %.2f  med %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(t[:, 4], [0, 50, 90, 100])))
late = t[:, 0] > 2.0
print("late starters:", late.sum(), "their start med %.2f end med %.2f" % (np.median(t[late, 0]) if late.any() else 0, np.median(t[late, 4]) if late.any() else 0))
hist, edges = np.histogram(t[:, 0], bins=20)
print("start histogram", list(zip(np.round(edges[:-1], 1), hist)))
hist, edges = np.histogram(t[:, 4], bins=20)
print("end histogram", list(zip(np.round(edges[:-1], 1), hist)))
