#!/usr/bin/env python3
"""Forward transform of the configs[1] batch (2 x 36 limbs) as ONE call on one stream, and as two calls on two streams
(fork / join with events) with the limbs split a : 36-a, so that the row pass of one part runs beside the column tail of
the other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha

N, LIMBS = 65536, 36
mods = pha.coeff_modulus_create(N, [60] + [50] * 35)
ctx = pha.PhantomContext(16, mods, 0)
buf = torch.randint(0, 1 << 49, (2, LIMBS, N), dtype=torch.int64, device="cuda")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def one():
    ctx.ntt_forward_batched(buf, 2, LIMBS, LIMBS)


def split(a):
    def fn():
        side.wait_stream(main)
        ctx.ntt_forward_batched(buf, 2, LIMBS, a, 0)
        with torch.cuda.stream(side):
            ctx.ntt_forward_batched(buf, 2, LIMBS, LIMBS - a, a)
        main.wait_stream(side)
    return fn


def t_us(fn, reps=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for rnd in range(2):
    print(f"one call: {t_us(one):.2f} us", flush=True)
    for a in (18, 20, 22, 24, 26, 28):
        print(f"  split {a}:{LIMBS - a} on two streams: {t_us(split(a)):.2f} us", flush=True)
