#!/usr/bin/env python3
"""Forward column pass with eight coefficients per lane (PHA_COL8=1) against the default: equality of the stored words at
BASELINE configs[1] (and at N = 2^17, 60-bit limbs), then `iters` timed forward transforms on each context (HIP events
around the loop; the per-kernel split comes from rocprofv3 --kernel-trace --stats over this script)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha


def ctx_with(log_n, mods, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pha.PhantomContext(log_n, mods, 0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def uniform(mods, n, polys, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.stack([torch.stack([torch.randint(0, int(m), (n,), dtype=torch.int64, device="cuda", generator=g) for m in mods])
                        for _ in range(polys)])


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    cases = ((16, [60] + [50] * 35), (16, [60] * 36), (17, [60] * 4 + [50] * 4), (15, [50] * 8))
    if len(sys.argv) > 2:
        cases = cases[int(sys.argv[2]):int(sys.argv[2]) + 1]
    for log_n, bits in cases:
        n = 1 << log_n
        mods = pha.coeff_modulus_create(n, bits)
        key, val = (sys.argv[3].split("=") if len(sys.argv) > 3 else ("PHA_COL8", "1"))
        base, col8 = ctx_with(log_n, mods, **{key: 0}), ctx_with(log_n, mods, **{key: val})
        x = uniform(mods, n, 2, log_n)
        a, b = x.clone(), x.clone()
        for p in range(2):
            base.nwt_2d_radix8_forward_inplace(a[p], len(mods), 0)
            col8.nwt_2d_radix8_forward_inplace(b[p], len(mods), 0)
        torch.cuda.synchronize()
        print(f"log_n {log_n} limbs {len(mods)} bits {bits[-1]}: equal = {bool(torch.equal(a, b))}", flush=True)
        if log_n == 16:
            phase = int(os.environ.get("AB_PHASE", "0"))
            inverse = bool(int(os.environ.get("AB_INVERSE", "0")))
            for name, ctx in (("default", base), ("variant", col8), ("default", base), ("variant", col8)):
                y = x.clone()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    ctx.ntt_launch_phase(y, 2, len(mods), len(mods), 0, inverse, phase)
                e0.record()
                for _ in range(iters):
                    ctx.ntt_launch_phase(y, 2, len(mods), len(mods), 0, inverse, phase)    # column kernel only, 72 limbs
                e1.record()
                torch.cuda.synchronize()
                print(f"  {name:8s} {e0.elapsed_time(e1) / iters * 1e3:8.2f} us per 72-limb pass (phase {phase}, inverse {inverse})", flush=True)
        base.close()
        col8.close()


if __name__ == "__main__":
    main()
