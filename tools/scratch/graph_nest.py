"""Minimal reproducer for the hipGraph capture of a tree of streams (GpuEvaluator.run_pair nested twice).
usage: python3 tools/scratch/graph_nest.py --nest 2 --ops mul|bsgs|rot"""
import argparse
import faulthandler
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nest", type=int, default=2)
    ap.add_argument("--ops", default="mul")
    ap.add_argument("--log-n", type=int, default=13)
    a = ap.parse_args()
    import torch
    import phantom_fhe_amd as pha
    from phantom_fhe_amd.ckks import GpuEvaluator
    n = 1 << a.log_n
    mods = pha.coeff_modulus_create(n, [50] + [40] * 7 + [50] * 2)
    ctx = pha.PhantomContext(a.log_n, mods, 2)
    sk = ctx.gen_secretkey(5, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, n // 2, seed=100)
    ev.add_rotation_keys([1, 2, 3, 4, 8])
    GpuEvaluator.MAX_NEST = a.nest
    rng = np.random.default_rng(0)
    v = rng.uniform(-1, 1, n // 2).astype(np.complex128)
    x = ev.encrypt(ev.encode(v, 2.0 ** 40, 1))

    def leaf(e, c, k):
        if a.ops == "mul":
            return e.mul_rescale(c, c)
        if a.ops == "torch":
            from phantom_fhe_amd.ckks import Ct
            return Ct(c.data + k, c.level, c.scale)
        if a.ops == "add":
            return e.add(c, c)
        if a.ops == "rot":
            return e.rotate(c, k)
        plains = [[e.encode_lt(v * (i + j + 1) / 8, 2.0 ** 40, c.level, n // 2) for i in range(2)] for j in range(2)]
        return e.bsgs(c, [0, 1], [0, 2], plains)

    def circuit(c):
        (p, q), (r, s) = ev.run_pair(lambda e: e.run_pair(lambda f: leaf(f, c, 1), lambda f: leaf(f, c, 2)),
                                     lambda e: e.run_pair(lambda f: leaf(f, c, 3), lambda f: leaf(f, c, 4)))
        return ev.add(ev.add(p, q), ev.add(r, s))

    want = circuit(x)
    torch.cuda.synchronize()
    print("eager ok", flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        circuit(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("warm ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = circuit(x)
    print("capture ok", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replay ok", flush=True)
    if a.ops != "bsgs":          # the encodings inside bsgs leaves are host work: same values either way
        assert torch.equal(out.data, want.data)
    print("equal ok", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
