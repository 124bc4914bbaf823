import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import phantom_fhe_amd as pha
LOG_N, N = 16, 1 << 16
def ev_ms(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
for limbs, polys in ((36, 2), (36, 8)):
    mods = pha.coeff_modulus_create(N, [60] + [50] * (limbs - 1))
    ctx = pha.PhantomContext(LOG_N, mods, 0)
    rng = np.random.default_rng(7)
    host = np.stack([np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in mods]) for _ in range(polys)])
    orig = pha.to_device(host); buf = orig.clone()
    for st in (0, 25, 50, 75, 100, 150, 0, 50):
        ctx.ntt_block_placement(16 + st)
        buf.copy_(orig)
        def step():
            ctx.ntt_forward_batched(buf, polys, limbs, limbs, 0); ctx.ntt_backward_batched(buf, polys, limbs, limbs, 0)
        for _ in range(5): step()
        torch.cuda.synchronize()
        ok = bool(torch.equal(buf, orig))
        rt = 1e3 * ev_ms(step, 100)
        fwd = 1e3 * ev_ms(lambda: ctx.ntt_forward_batched(buf, polys, limbs, limbs, 0), 100)
        print(json.dumps({"limbs": limbs, "polys": polys, "stagger_ticks": st, "ok": ok, "roundtrip_us": rt, "fwd_us": fwd}), flush=True)
    ctx.close()
