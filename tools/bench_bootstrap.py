#!/usr/bin/env python3
"""CKKS bootstrapping latency at N=2^16 on one GPU (BASELINE config 5's primitive; the reference's
bootstrap_sparse* / bootstrap_full*, src/boot/Bootstrapper.cu).  Prints one JSON line.

    python tools/bench_bootstrap.py [--slots 64] [--log-n 16] [--reps 3]
    python tools/bench_bootstrap.py --slots 32768 --transform fft      # full slots, FFT-factored transforms
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import phantom_fhe_amd as pha
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--extra-levels", type=int, default=4)
    ap.add_argument("--alpha", type=int, default=4)
    ap.add_argument("--delta-bits", type=int, default=50, help="working scale; q0 has 10 bits more")
    ap.add_argument("--transform", choices=("dense", "fft"), default="dense")
    ap.add_argument("--groups", type=int, default=3, help="levels of each FFT-factored transform")
    ap.add_argument("--special-bits", type=int, default=60, help="size of the special primes (below 2^50: FP64 transform path)")
    ap.add_argument("--c2s-bits", type=int, default=0, help="CoeffToSlot primes (default: min(60, delta-bits + 8))")
    ap.add_argument("--slim", action="store_true", help="SlotToCoeff first, on the last levels (needs --transform fft)")
    ap.add_argument("--s2c-groups", type=int, default=0, help="levels of SlotToCoeff (default: --groups)")
    ap.add_argument("--evalmod-bits", type=int, default=0, help="EvalMod working scale / primes (default: delta-bits)")
    ap.add_argument("--chain", default="", help="explicit Q chain, bottom first, e.g. 54,48x16,54x14 (the reference's "
                    "tests/bootstrapping_test.cu:180-185: logq, remaining_level x logp, boot_level x logq)")
    ap.add_argument("--K", type=int, default=0)
    ap.add_argument("--r", type=int, default=0)
    ap.add_argument("--degree", type=int, default=31)
    args = ap.parse_args()
    n = 1 << args.log_n
    d = args.delta_bits
    e = args.evalmod_bits or d
    evalmod = [e] * (9 if args.log_n < 15 else 10)
    if args.transform == "fft":
        g = min(args.groups, args.slots.bit_length() - 1)
        g2 = min(args.s2c_groups or args.groups, args.slots.bit_length() - 1)
        q_bits = [d + 10] + [d] * args.extra_levels + [d] * g2 + evalmod + [args.c2s_bits or min(60, d + 8)] * g
    else:
        q_bits = [d + 10] + [d] * args.extra_levels + [d] + evalmod + [d, min(60, d + 18)]
    if args.chain:
        q_bits = []
        for part in args.chain.split(","):
            bits, _, rep = part.partition("x")
            q_bits += [int(bits)] * int(rep or 1)
    while len(q_bits) % args.alpha:
        q_bits.insert(1, d)
    mods = pha.coeff_modulus_create(n, q_bits + [args.special_bits] * args.alpha)
    ctx = pha.PhantomContext(args.log_n, mods, args.alpha)
    sk = ctx.gen_secretkey(21, hamming_weight=192 if args.log_n >= 15 else 32)
    ev = GpuEvaluator(pha, ctx, sk, args.slots)
    t0 = time.perf_counter()
    bs = Bootstrapper(ev, K=args.K or (25 if args.log_n >= 15 else 12), r=args.r or (3 if args.log_n < 15 else 4), degree=args.degree,
                      delta=2.0 ** e, msg_scale=2.0 ** d,
                      transform=args.transform, c2s_groups=args.groups, s2c_groups=args.s2c_groups or args.groups, slim=args.slim)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    rng = np.random.default_rng(2)
    msg = rng.uniform(-1, 1, args.slots) + 1j * rng.uniform(-1, 1, args.slots)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom - bs.pre_levels))
    out = bs.bootstrap(ct)
    err = float(np.abs(ev.decrypt_decode(out) - msg).max())
    torch.cuda.synchronize()
    times = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = bs.bootstrap(ct)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    print(json.dumps({"metric": "CKKS bootstrap ms", "transform": args.transform, "order": "slim" if args.slim else "usual", "value": 1e3 * min(times), "unit": "ms", "log_n": args.log_n,
                      "slots": args.slots, "delta_bits": d, "evalmod_bits": e, "q_limbs": len(q_bits), "special_primes": args.alpha, "levels_after": ev.limbs(out.level) - 1,
                      "max_abs_error": err, "log_qp": round(float(sum(np.log2(np.array(mods, dtype=np.float64)))), 1), "setup_s": setup_s, "rotation_keys": len(ev.galois), "n_gpus": 1}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
