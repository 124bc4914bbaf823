"""Limb-parallel key switch / rotation / rescale on one GPU: `world` simulated
ranks (threads, each with its own context and buffers, exchanging through ThreadComm) must
reproduce the oracle's single-device result bit for bit on the limbs they own, reading nothing
outside their windows except what the exchanges deliver."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, LOG_N = 4096, 12
POISON = np.uint64(0x5EEDBADC0FFEE123)


def _run_ranks(world, body):
    from phantom_fhe_amd.limb_parallel import ThreadComm
    shared = ThreadComm.Shared(world)
    errors, results = [], [None] * world

    def target(rank):
        try:
            results[rank] = body(rank, ThreadComm(shared, rank))
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((rank, e))
            shared.barrier.abort()

    threads = [threading.Thread(target=target, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)] or errors
    if real:
        raise real[0][1]
    return results


def _owned_only(a, begin, count, axis_limb):
    """copy of `a` with every limb outside [begin, begin+count) poisoned"""
    out = np.full_like(a, POISON)
    sl = [slice(None)] * a.ndim
    sl[axis_limb] = slice(begin, begin + count)
    out[tuple(sl)] = a[tuple(sl)]
    return out


def _uniform_evk(o, seed):
    beta_max = -(-o.size_Q // o.size_P)
    return [np.stack([o.uniform_poly(list(range(o.size_QP)), seed + 2 * d),
                      o.uniform_poly(list(range(o.size_QP)), seed + 2 * d + 1)]) for d in range(beta_max)]


def _check_world(gpu, pha, orc, log_n, mods, size_p, chain_index, world):
    """every limb-parallel operation on `world` simulated ranks against the oracle's single-device result"""
    from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, limb_window
    n = 1 << log_n
    o = orc.Oracle(log_n, mods, size_p)
    size_ql = o.size_Q - (chain_index - 1)
    lim = list(range(size_ql))
    ct = np.stack([o.uniform_poly(lim, 31), o.uniform_poly(lim, 32)])
    c2 = o.uniform_poly(lim, 33)
    evk_h = _uniform_evk(o, 200)
    elt = orc.galois_elt_from_step(log_n, 3)

    exp_ks = ct.copy()
    o.keyswitch_inplace(size_ql, exp_ks, c2, evk_h)
    rot_c1 = o.apply_galois_ntt(ct[1], elt)
    exp_rot = np.stack([o.apply_galois_ntt(ct[0], elt), np.zeros_like(rot_c1)])
    o.keyswitch_inplace(size_ql, exp_rot, rot_c1, evk_h)
    exp_rs = o.divide_and_round_q_last_ntt(size_ql, ct, 2) if size_ql > 1 else None
    ct_b = np.stack([o.uniform_poly(lim, 41), o.uniform_poly(lim, 42)])
    d3 = o.tensor_prod_2x2(ct, ct_b)                    # HMult + relinearise, end to end (BASELINE configs[2])
    exp_mul = d3[:2].copy()
    o.keyswitch_inplace(size_ql, exp_mul, d3[2], evk_h)

    # the windows tile Q_l and P exactly
    wins = [limb_window(size_ql, size_p, world, r) for r in range(world)]
    assert sorted(l for q0, qc, _, _ in wins for l in range(q0, q0 + qc)) == lim
    assert sorted(l for _, _, p0, pc in wins for l in range(p0, p0 + pc)) == list(range(size_p))

    def body(rank, comm):
        ctx = pha.PhantomContext(log_n, mods, size_p)
        ev = LimbParallelEvaluator(ctx, comm)
        q0, qc, _, _ = ev.windows(chain_index)[rank]
        evk = pha.EvalKey([pha.to_device(k) for k in evk_h])
        out = {}
        d_ct = pha.to_device(_owned_only(ct, q0, qc, 1))
        ev.keyswitch_inplace(chain_index, d_ct, pha.to_device(_owned_only(c2, q0, qc, 0)), evk)
        out["ks_owned"] = pha.to_host(d_ct)[:, q0:q0 + qc].copy()
        ev.gather_limbs(chain_index, [d_ct[0], d_ct[1]])
        out["ks_full"] = pha.to_host(d_ct)

        d_out3 = pha.to_device(np.full((3, size_ql, n), POISON, dtype=np.uint64))
        prod = ev.multiply_and_relin(chain_index, pha.to_device(_owned_only(ct, q0, qc, 1)),
                                     pha.to_device(_owned_only(ct_b, q0, qc, 1)), evk, d_out3)
        out["mul_owned"] = pha.to_host(prod)[:, q0:q0 + qc].copy()

        d_rot = pha.to_device(_owned_only(ct, q0, qc, 1))
        ev.apply_galois_inplace(chain_index, d_rot, elt, evk)
        out["rot_owned"] = pha.to_host(d_rot)[:, q0:q0 + qc].copy()

        if size_ql > 1:
            d_dst = pha.to_device(np.full((2, size_ql - 1, n), POISON, dtype=np.uint64))
            ev.rescale_to_next(chain_index, pha.to_device(_owned_only(ct, q0, qc, 1)), 2, d_dst)
            ev.gather_limbs(chain_index, [d_dst[0], d_dst[1]], size_ql - 1)
            out["rs_full"] = pha.to_host(d_dst)
        gpu.cuda.synchronize()
        ctx.close()
        return out

    results = _run_ranks(world, body)
    for rank, res in enumerate(results):
        q0, qc, _, _ = wins[rank]
        assert (res["ks_owned"] == exp_ks[:, q0:q0 + qc]).all(), f"rank {rank} key switch"
        assert (res["ks_full"] == exp_ks).all(), f"rank {rank} gathered key switch"
        assert (res["rot_owned"] == exp_rot[:, q0:q0 + qc]).all(), f"rank {rank} rotation"
        assert (res["mul_owned"] == exp_mul[:, q0:q0 + qc]).all(), f"rank {rank} HMult + relinearise"
        if exp_rs is not None:
            assert (res["rs_full"] == exp_rs).all(), f"rank {rank} rescale"


@pytest.mark.parametrize("world,size_p,chain_index", [(2, 2, 1), (3, 2, 2), (8, 2, 1), (4, 1, 1), (3, 3, 1), (2, 3, 3),
                                                      (5, 1, 4), (4, 3, 2), (5, 2, 1), (1, 2, 1)])
def test_keyswitch_rotation_rescale(gpu, pha, orc, world, size_p, chain_index):
    _check_world(gpu, pha, orc, LOG_N, orc.coeff_modulus_create(N, [45] * 6 + [55] * size_p), size_p, chain_index, world)


@pytest.mark.parametrize("seed", range(24))
def test_random_geometry_and_world(gpu, pha, orc, seed):
    """the same checks with the shape drawn: N = 2^11 .. 2^13, alpha 1 .. 4, dnum 1 .. 4, primes of 36 .. 60 bits in any mix,
    any level with two limbs or more, 1 .. 8 ranks (more ranks than limbs included: some ranks then own nothing)"""
    rng = np.random.default_rng(7000 + seed)
    log_n = int(rng.choice([11, 12, 13]))
    alpha = int(rng.integers(1, 5))
    dnum = max(int(rng.integers(1, 5)), 2 if alpha == 1 else 1)
    size_q = alpha * dnum
    q_bits = [int(b) for b in rng.integers(36, 61, size_q)]
    p_bits = [int(b) for b in rng.integers(45, 61, alpha)]
    chain_index = int(rng.integers(1, size_q))
    world = int(rng.integers(1, 9))
    mods = orc.coeff_modulus_create(1 << log_n, q_bits + p_bits)
    try:
        _check_world(gpu, pha, orc, log_n, mods, alpha, chain_index, world)
    except AssertionError as e:
        raise AssertionError(f"seed {seed}: N=2^{log_n} alpha={alpha} Q={q_bits} P={p_bits} chain_index={chain_index} world={world}: {e}")


def test_window_validation(gpu, pha, orc):
    mods = orc.coeff_modulus_create(N, [45] * 4 + [55] * 2)
    ctx = pha.PhantomContext(LOG_N, mods, 2)
    d = pha.to_device(np.zeros((2, 6, N), dtype=np.uint64))
    with pytest.raises(pha.PhantomError) as e:
        ctx.lp_moddown_finish(1, d, d, d, (3, 2, 0, 0))  # Q window [3,5) exceeds the 4 Q limbs
    assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    with pytest.raises(pha.PhantomError):
        ctx.lp_moddown_begin(1, d, d, (0, 0, 1, 2))  # P window [1,3) exceeds alpha = 2
    # exchange buffers: rows beyond the buffer and row tables pointing outside it are refused, nothing is launched
    c2 = pha.to_device(np.zeros((4, N), dtype=np.uint64))
    small = pha.to_device(np.zeros((3, N), dtype=np.uint64))
    t_mod_up = pha.to_device(np.zeros((2, 6, N), dtype=np.uint64))
    for call in (lambda: ctx.lp_modup_begin(1, c2, small, (0, 2, 0, 0), row_begin=2),            # rows 2, 3 of a 3-row buffer
                 lambda: ctx.lp_modup_finish(1, t_mod_up, c2, small, (0, 2, 0, 1)),              # fewer rows than Q_l has limbs
                 lambda: ctx.lp_modup_finish(1, t_mod_up, c2, c2, (0, 2, 0, 1), q_rows=[0, 1, 2, 4]),
                 lambda: ctx.lp_moddown_begin(1, d, small, (0, 0, 0, 2), row_begin=0, poly_stride=2),   # poly 1 needs rows 2, 3
                 lambda: ctx.lp_moddown_finish(1, d, d, small, (0, 2, 0, 0), p_rows=[0, 1], poly_stride=2)):
        with pytest.raises(pha.PhantomError) as e:
            call()
        assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    ctx.close()


@pytest.mark.parametrize("backend,nproc", [("gloo", 2), ("nccl", 1)])
def test_torch_distributed_ranks(gpu, backend, nproc):
    """The same path through torch.distributed: two processes sharing this GPU over gloo (host-staged
    exchange), and the RCCL code path with the one rank a one-GPU box allows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                          "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(root, "tools", "lp_keyswitch.py"),
                          "--backend", backend, "--log-n", "12", "--q-limbs", "5", "--p-limbs", "2", "--q-bits", "45",
                          "--p-bits", "55", "--check"], capture_output=True, text=True, timeout=170, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count("ok rank") == nproc


def test_full_size_two_ranks_equal_single_gpu(gpu):
    """BASELINE configs[2] at full size (N = 2^16, 36 + 12 limbs, dnum 3) on two torch.distributed ranks (gloo, sharing
    this GPU): the sharded HMult + relinearise and a rotation reproduce pha_multiply_and_relin_inplace /
    pha_apply_galois_inplace bit for bit, on the owned limbs and after the gather."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29643", os.path.join(root, "tools", "lp_keyswitch.py"),
                          "--backend", "gloo", "--check-single"], capture_output=True, text=True, timeout=170, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert out.stdout.count("ok rank") == 2


@pytest.mark.parametrize("config,steps", [(2, 20), (3, 5), (4, 2)])
def test_bench_legs_with_two_ranks(gpu, config, steps):
    """bench.py --gpus 2: the script starts its two ranks itself (torch.distributed.run child), both share this GPU and
    rendezvous over gloo (RCCL refuses two ranks on one device); every leg prints one line with ranks = 2 and n_gpus = the
    number of distinct devices the ranks ran on (1 on a one-GPU box)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", str(config),
                          "--steps", str(steps), "--warmup", "1"], capture_output=True, text=True, timeout=170, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    import torch
    assert line["ranks"] == 2 and line["n_gpus"] == min(2, torch.cuda.device_count())
    assert line["steps"] == steps and line["value"] > 0
    if config != 2:
        assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["peak"] == 8000.0 * line["n_gpus"]
    assert line["scaling"] == ("weak" if config == 2 else "strong")
