#!/usr/bin/env python3
"""BASELINE config 4: CKKS rotate (apply_galois + key switch) on 64 independent ciphertexts,
N=2^16, 36 Q limbs + 12 special primes, dealt to the ranks of one node (no data-path collective;
`scaling: strong` — the 64 ciphertexts are the whole job).

    python tools/bench_rotate64.py                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/bench_rotate64.py
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import phantom_fhe_amd as pha

N, SIZE_Q, SIZE_P, TOTAL = 65536, 36, 12, 64


def main():
    steps = int(os.environ.get("STEPS", "3"))
    rank, world, local_rank, dist = bench.dist_setup("nccl")
    torch.cuda.set_device(local_rank)
    mods = pha.coeff_modulus_create(N, [60] + [50] * (SIZE_Q - 1) + [60] * SIZE_P)
    # STREAMS > 1: independent ciphertexts go round-robin to that many (context, stream) pairs, so that
    # the latency-bound small launches of one key switch overlap with another's (each context owns its
    # workspace; contexts are not shared between streams)
    n_streams = int(os.environ.get("STREAMS", "2"))
    ctxs = [pha.PhantomContext(16, mods, SIZE_P) for _ in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
    ctx = ctxs[0]
    lo, hi = bench.shard_units(TOTAL, world, rank)
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    small = int(mods.min())
    cts = [torch.randint(0, small, (2, SIZE_Q, N), dtype=torch.int64, device="cuda", generator=g) for _ in range(lo, hi)]
    evk = pha.EvalKey([torch.randint(0, small, (2, SIZE_Q + SIZE_P, N), dtype=torch.int64, device="cuda", generator=g)
                       for _ in range(SIZE_Q // SIZE_P)])
    elt = pha.galois_elt_from_step(16, 1)

    # BATCH > 1: the rank's ciphertexts go through pha_apply_galois_batched in groups of BATCH (one launch sequence
    # per group; every kernel spans the group)
    batch = int(os.environ.get("BATCH", "16"))
    groups = [torch.stack(cts[i:i + batch]) for i in range(0, len(cts), batch)] if batch > 1 else []

    def step():
        if batch > 1:
            for i, grp in enumerate(groups):
                with torch.cuda.stream(streams[i % n_streams]):
                    ctxs[i % n_streams].apply_galois_batched(1, grp, elt, evk)
            return
        for i, ct in enumerate(cts):
            with torch.cuda.stream(streams[i % n_streams]):
                ctxs[i % n_streams].apply_galois_inplace(1, ct, elt, evk)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    el = bench.max_over_ranks(time.perf_counter() - t0, dist, "cuda")
    if rank == 0:
        print(json.dumps({"metric": "CKKS rotate, 64 independent ciphertexts (N=2^16, 36+12 limbs, dnum=3)",
                          "value": TOTAL * steps / el, "unit": "ciphertexts/s", "ms_per_64": 1e3 * el / steps,
                          "n_gpus": world, "steps": steps, "streams": n_streams, "batch": batch, "scaling": "strong", "data": "synthetic"}), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
