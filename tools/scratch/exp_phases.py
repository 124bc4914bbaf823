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
for limbs, polys in ((36, 2), (36, 8), (12, 2)):
    mods = pha.coeff_modulus_create(N, [60] + [50] * (limbs - 1))
    ctx = pha.PhantomContext(LOG_N, mods, 0)
    rng = np.random.default_rng(7)
    host = np.stack([np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in mods]) for _ in range(polys)])
    orig = pha.to_device(host); buf = orig.clone()
    for rep in range(2):
        buf.copy_(orig)
        def step():
            ctx.ntt_forward_batched(buf, polys, limbs, limbs, 0); ctx.ntt_backward_batched(buf, polys, limbs, limbs, 0)
        for _ in range(5): step()
        torch.cuda.synchronize()
        ok = bool(torch.equal(buf, orig))
        rec = {"tag": os.environ.get("EXP_TAG", ""), "limbs": limbs, "polys": polys, "ok": ok, "roundtrip_us": round(1e3 * ev_ms(step, 100), 2)}
        rec["fwd_us"] = round(1e3 * ev_ms(lambda: ctx.ntt_forward_batched(buf, polys, limbs, limbs, 0), 100), 2)
        for name, inv, ph in (("fwd_col", 0, 0), ("fwd_row", 0, 1), ("inv_row", 1, 1), ("inv_col", 1, 0)):
            fn = (lambda inv=inv, ph=ph: ctx.ntt_launch_phase(buf, polys, limbs, limbs, 0, inv, ph))
            for _ in range(3): fn()
            rec[name] = round(1e3 * ev_ms(fn, 50), 2)
        rec["fwd_frac"] = round(polys * limbs * N * 16 / (rec["fwd_us"] * 1e-6) / 8e12, 4)
        print(json.dumps(rec), flush=True)
    ctx.close()
