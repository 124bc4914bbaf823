"""Entry / exit timestamps per tile of the forward column kernel (library built with -DPHA_TRACE=3), last launch of a chain."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import phantom_fhe_amd as pha
pha.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "libphantom_trace.so")
L, N = 36, 65536
bits = [60] * 36 if os.environ.get("TRACE_INT") else [60] + [50] * 35
mods = pha.coeff_modulus_create(N, bits)
ctx = pha.PhantomContext(16, mods, 0)
lib = pha.lib()
lib.pha_set_trace.argtypes = [ctypes.c_void_p]
lib.pha_set_trace.restype = ctypes.c_int
x = torch.randint(0, 1 << 49, (2, L, N), dtype=torch.int64, device="cuda")
nwg = 2 * L * (N // 4096)
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
fwd = lambda: ctx.ntt_launch_phase(x, 2, L, L, 0, 0, 0)
for _ in range(20): fwd()
torch.cuda.synchronize()
assert lib.pha_set_trace(buf.data_ptr()) == 0
torch.cuda.synchronize()
for _ in range(6): fwd()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 8)[:, [0, 4]].astype(np.float64)
print("raw ticks: first entry %d last exit %d" % (int(t[:, 0].min()), int(t[:, 1].max())))
t = (t - t[:, 0].min()) * 0.01
print("span %.2f us; life med %.2f p90 %.2f" % (t[:, 1].max(), np.median(t[:, 1] - t[:, 0]), np.percentile(t[:, 1] - t[:, 0], 90)))
for lo in range(0, nwg, 128):
    s, e = t[lo:lo + 128, 0], t[lo:lo + 128, 1]
    print("%4d entry med %.2f min %.2f max %.2f | exit med %.2f max %.2f | life med %.2f" % (lo, np.median(s), s.min(), s.max(), np.median(e), e.max(), np.median(e - s)))
