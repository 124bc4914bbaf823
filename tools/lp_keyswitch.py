#!/usr/bin/env python3
"""Limb-parallel key switch across the ranks of one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
        tools/lp_keyswitch.py [--backend nccl] [--log-n 16 --q-limbs 36 --p-limbs 12] [--check] [--steps K]

--check compares every rank's owned limbs with the CPU oracle (small sizes only), --check-single with the single-GPU
entry points on the same inputs (any size, e.g. BASELINE configs[2]); otherwise it
times K key switches (barrier + synchronize on both sides, max over ranks) and rank 0 prints one
JSON line.  With --backend gloo all ranks may share one GPU (tests on a one-GPU box)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--q-limbs", type=int, default=36)
    ap.add_argument("--p-limbs", type=int, default=12)
    ap.add_argument("--q-bits", type=int, default=50)
    ap.add_argument("--p-bits", type=int, default=60)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--check-single", action="store_true",
                    help="compare HMult + relinearise and a rotation with the single-GPU entry points on the same inputs (any size)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import phantom_fhe_amd as pha
    from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, TorchDistComm

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    dist.init_process_group(args.backend)
    comm = TorchDistComm()
    rank, world = comm.rank, comm.world

    n = 1 << args.log_n
    mods = pha.coeff_modulus_create(n, [60] + [args.q_bits] * (args.q_limbs - 1) + [args.p_bits] * args.p_limbs)
    ctx = pha.PhantomContext(args.log_n, mods, args.p_limbs)
    ev = LimbParallelEvaluator(ctx, comm)
    size_ql, size_qp = args.q_limbs, args.q_limbs + args.p_limbs
    q0, qc, p0, pc = ev.windows(1)[rank]
    beta = ctx.beta(1)

    if args.check:
        import oracle
        o = oracle.Oracle(args.log_n, mods, args.p_limbs)
        lim = list(range(size_ql))
        ct = np.stack([o.uniform_poly(lim, 1), o.uniform_poly(lim, 2)])
        c2 = o.uniform_poly(lim, 3)
        evk_h = [np.stack([o.uniform_poly(list(range(size_qp)), 10 + 2 * d), o.uniform_poly(list(range(size_qp)), 11 + 2 * d)])
                 for d in range(beta)]
        exp = ct.copy()
        o.keyswitch_inplace(size_ql, exp, c2, evk_h)
        d_ct, d_c2 = pha.to_device(ct), pha.to_device(c2)
        evk = pha.EvalKey([pha.to_device(k) for k in evk_h])
        ev.keyswitch_inplace(1, d_ct, d_c2, evk)
        got = pha.to_host(d_ct)
        assert (got[:, q0:q0 + qc] == exp[:, q0:q0 + qc]).all(), f"rank {rank}: owned limbs differ from the oracle"
        ev.gather_limbs(1, [d_ct[0], d_ct[1]])
        assert (pha.to_host(d_ct) == exp).all(), f"rank {rank}: gathered ciphertext differs from the oracle"
        # HMult + relinearise end to end (BASELINE configs[2])
        ct_b = np.stack([o.uniform_poly(lim, 5), o.uniform_poly(lim, 6)])
        d3 = o.tensor_prod_2x2(ct, ct_b)
        exp_mul = d3[:2].copy()
        o.keyswitch_inplace(size_ql, exp_mul, d3[2], evk_h)
        out3 = torch.empty((3, size_ql, n), dtype=torch.int64, device="cuda")
        prod = ev.multiply_and_relin(1, pha.to_device(ct), pha.to_device(ct_b), evk, out3)
        assert (pha.to_host(prod)[:, q0:q0 + qc] == exp_mul[:, q0:q0 + qc]).all(), f"rank {rank}: HMult + relin differs"
        dist.barrier()
        print(f"ok rank {rank}/{world} window Q[{q0},+{qc}) P[{p0},+{pc})", flush=True)
    elif args.check_single:
        # every rank runs the single-GPU operation on a context of its own and the sharded one on the shared inputs: the
        # owned limbs, then the gathered ciphertext, must equal pha_multiply_and_relin_inplace / pha_apply_galois_inplace
        dev = torch.device("cuda")
        g = torch.Generator(device=dev).manual_seed(11)                # the same inputs on every rank
        lo = int(mods.min())
        rnd = lambda *shape: torch.randint(0, lo, shape, dtype=torch.int64, device=dev, generator=g)  # noqa: E731
        ct_a, ct_b = rnd(2, size_ql, n), rnd(2, size_ql, n)
        evk = pha.EvalKey([rnd(2, size_qp, n) for _ in range(beta)])
        single = pha.PhantomContext(args.log_n, mods, args.p_limbs)
        want = torch.cat([ct_a.clone(), torch.zeros((1, size_ql, n), dtype=torch.int64, device=dev)])
        single.multiply_and_relin_inplace(1, want, ct_b.clone(), evk)
        out3 = torch.full((3, size_ql, n), -1, dtype=torch.int64, device=dev)
        prod = ev.multiply_and_relin(1, ct_a.clone(), ct_b.clone(), evk, out3)
        torch.cuda.synchronize()
        assert torch.equal(prod[:, q0:q0 + qc], want[:2, q0:q0 + qc]), f"rank {rank}: HMult + relinearise differs on the owned limbs"
        ev.gather_limbs(1, [prod[0], prod[1]])
        assert torch.equal(prod, want[:2]), f"rank {rank}: gathered HMult + relinearise differs"
        elt = pha.galois_elt_from_step(args.log_n, 5)
        want_rot = ct_a.clone()
        single.apply_galois_inplace(1, want_rot, elt, evk)
        rot = ct_a.clone()
        ev.apply_galois_inplace(1, rot, elt, evk)
        torch.cuda.synchronize()
        assert torch.equal(rot[:, q0:q0 + qc], want_rot[:, q0:q0 + qc]), f"rank {rank}: rotation differs on the owned limbs"
        single.close()
        dist.barrier()
        print(f"ok rank {rank}/{world} window Q[{q0},+{qc}) P[{p0},+{pc})", flush=True)
    else:
        dev = torch.device("cuda")
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        lo = int(mods.min())
        rnd = lambda *shape: torch.randint(0, lo, shape, dtype=torch.int64, device=dev, generator=g)  # noqa: E731
        d_ct, d_ct2, out3 = rnd(2, size_ql, n), rnd(2, size_ql, n), torch.empty((3, size_ql, n), dtype=torch.int64, device=dev)
        evk = pha.EvalKey([rnd(2, size_qp, n) for _ in range(beta)])
        for _ in range(args.warmup):
            ev.multiply_and_relin(1, d_ct, d_ct2, evk, out3)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ev.multiply_and_relin(1, d_ct, d_ct2, evk, out3)
        torch.cuda.synchronize()
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                          device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "limb-parallel HMult + relinearise ms", "value": 1e3 * float(el) / args.steps, "unit": "ms",
                              "n_gpus": world, "steps": args.steps, "log_n": args.log_n, "size_Q": size_ql,
                              "size_P": args.p_limbs, "backend": args.backend, "scaling": "strong"}), flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
