#!/usr/bin/env python3
"""Single-launch forward NTT (ntt_fused.cuh) against the two-kernel transform: bit-exactness under repetition, then time.
Needs a library built with the experiment compiled in:
    PHA_EXTRA_DEFINES=-DPHA_EXPERIMENT_SINGLE_LAUNCH python3 phantom-fhe_amd/build.py
usage: python3 tools/scratch/exp_fused.py [--rounds R] [--depth D] [--debug BITS]   (WGS=3,4 selects workgroups per CU)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phantom_fhe_amd as pha

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=20)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--debug", type=int, default=0)
ap.add_argument("--depth", type=int, default=4)
a = ap.parse_args()
N, LIMBS = 65536, 36
WGS = [int(x) for x in os.environ.get("WGS", "3,4,5,6").split(",")]
mods = pha.coeff_modulus_create(N, [60] + [50] * 35)
ctx = pha.PhantomContext(16, mods, 0)
import ctypes
_lib = pha.lib()
_lib.pha_ntt_single_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
_lib.pha_ntt_single_launch_depth.argtypes = [ctypes.c_void_p, ctypes.c_int]
_lib.pha_ntt_single_launch_status.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]


def single_launch(enable, wgs=4, depth=None):
    assert _lib.pha_ntt_single_launch(ctx._h, int(enable), int(wgs)) == 0
    if depth is not None:
        assert _lib.pha_ntt_single_launch_depth(ctx._h, int(depth)) == 0


def single_launch_status():
    v = ctypes.c_uint32(0)
    assert _lib.pha_ntt_single_launch_status(ctx._h, ctypes.byref(v)) == 0
    return v.value

q = torch.tensor([int(m) for m in mods], dtype=torch.int64, device="cuda")


def rand(polys, limbs):
    x = torch.randint(0, 1 << 62, (polys, LIMBS, N), dtype=torch.int64, device="cuda")
    return x % q[None, :, None]


def t_us(fn, reps):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for polys, limbs in ((2, 36), (8, 36)):
    src = rand(polys, limbs)
    single_launch(0)
    ref = src.clone()
    ctx.ntt_forward_batched(ref, polys, LIMBS, limbs)
    buf = src.clone()
    base = t_us(lambda: ctx.ntt_forward_batched(buf, polys, LIMBS, limbs), a.reps)
    row = {"polys": polys, "limbs": limbs, "two_kernel_us": round(base, 2)}
    for wgs in WGS:
        single_launch(1 | (a.debug << 1), wgs, a.depth)
        bad = 0
        for r in range(a.rounds):
            got = src.clone()
            ctx.ntt_forward_batched(got, polys, LIMBS, limbs)
            bad += int((got[:, :limbs] != ref[:, :limbs]).sum().item())
        to = single_launch_status()
        buf = src.clone()
        us = t_us(lambda: ctx.ntt_forward_batched(buf, polys, LIMBS, limbs), a.reps)
        row[f"wgs{wgs}"] = {"us": round(us, 2), "mismatched_words": bad, "timed_out": to}
        if (bad or to) and not a.debug:
            break
    print(json.dumps(row), flush=True)
    algo = polys * limbs * N * 8 * 2
    best = min([base] + [v["us"] for k, v in row.items() if k.startswith("wgs")])
    print(f"   frac of 8 TB/s: two-kernel {algo / base / 8e6:.3f}, best {algo / best / 8e6:.3f}", flush=True)
