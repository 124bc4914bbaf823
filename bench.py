#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on one node of MI355X GPUs, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4]

--config 2 (default; BASELINE configs[1], the configuration `metric` is quoted on): batched RNS NTT/INTT, N = 2^16,
    36 limbs x 2 polys (uint64[2][36][65536]) per GPU.  A step = one forward NTT of the whole batch followed by one
    inverse NTT of it (144 limb-transforms through pha_ntt_forward_batched / pha_ntt_backward_batched; the round trip
    keeps the buffer canonical).  value = limb-transforms per second over all ranks; weak scaling: every rank owns its
    own batch, the path has no exchange step, so there is no collective on the data path.
--config 3 (configs[2]): CKKS HMult + relinearise, N = 2^16, 36 + 12 limbs (dnum 3), the limbs sharded over the ranks
    with RCCL all-gathers at the two base conversions (phantom-fhe_amd/limb_parallel.py).  value = ms per HMult+relin
    (strong scaling).
--config 4 (configs[3]): CKKS rotate on 64 independent ciphertexts dealt to the ranks; value = ciphertexts/s (strong
    scaling, no data-path collective).

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) the script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` as a child
process BEFORE anything touches the GPU and relays the child's JSON line; under a launcher it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  `n_gpus` is the number of ranks that took part.

Extra objects on the config-2 line:
  roofline     — the forward transform (its two kernels, fntt_col_kernel + fntt_row_kernel), timed live with HIP events
                 on the launching stream: achieved = algorithmic bytes (N*8 read + N*8 written per limb, SURVEY.md
                 §8d) / time, frac = achieved / 8 TB/s.  `kernels` has every kernel's own average launch time and
                 the rate at which it streams its limb once; `alu` prices the same transform against the measured
                 32-bit multiply and FP64 issue peaks (the second bound north_star asks for); `traffic` is the
                 fabric-side byte count of one forward transform from a separate rocprofv3 --pmc pass
                 (tools/profile.sh -> profiles/traffic_latest.json), reported only when that file was produced from
                 the kernel sources this run uses (hash of csrc/), else null.
                 `int_path` is the same forward transform on a chain of 36 sixty-bit primes (every limb on the
                 32-bit-multiply butterflies; the headline chain has 35 of 36 limbs on the FP64 butterflies).
  cpu_baseline — the oracle (a scalar port; kind "port") timed on the host cores on a bounded sample of the same
                 workload.
The config-3 and config-4 lines carry `roofline` (SURVEY.md §8(d)'s ~0.9 GiB of algorithmic traffic per key-switched
operation against 8 TB/s per GPU that took part) and, at one rank, `cpu_baseline` (the oracle's key switch on the host).
`n_gpus` counts the distinct devices the ranks ran on; `ranks` the processes.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LOG_N = 16
N = 1 << LOG_N
LIMBS = 36
POLYS = 2
BITS = [60] + [50] * 35  # BASELINE.md config 2 / SURVEY.md §8(d): first prime 60-bit, scale primes 50-bit
SIZE_P = 12              # configs 3 / 4: 12 special primes of 60 bits (dnum 3)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_LIMB = N * 8 * 2
KS_ALGO_BYTES = 0.9 * 2 ** 30   # HMult + relinearise at config 3, every buffer once per producing / consuming kernel
BUTTERFLIES_PER_LIMB = (N // 2) * LOG_N
# measured issue peaks on MI355X (tools/ubench.hip, profiles/r1a_ubench.log), lane-operations per second chip-wide
INT32_MUL_PEAK = 31.03e12    # v_mad_u64_u32
FP64_PEAK = 36.74e12         # v_fma_f64
MUL32_PER_INT_BUTTERFLY = 9      # csrc/modarith.cuh: approximate-quotient Shoup product
FP64_PER_FP_BUTTERFLY = 10.0     # csrc/modarith.cuh FpField: 639 FP64 instructions per 64 butterflies in the ISA (column kernel)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4))
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: CPU-side rehearsal of the launcher path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spinup-ms", type=float, default=30.0,
                    help="set-up, before the W warm-up steps: run the step untimed for this long so that the GPU's clocks have left "
                         "their idle state (0: none); the timed region is unchanged -- exactly K steps between barrier + synchronize")
    ap.add_argument("--no-secondary", action="store_true", help="skip the HMult/rotate/bootstrap side measurements and roofline.int_path (profiling runs: kernel statistics then hold the headline chain only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks rendezvous, time an empty step and print the line")
    ap.add_argument("--pmc-traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="fabric-side bytes per launch from a separate rocprofv3 --pmc pass (tools/profile.sh)")
    return ap.parse_args(argv)


def kernel_source_hash():
    """Identifies the kernel sources a traffic file was measured on (the GPU box has no .git)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "phantom-fhe_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def shard_units(total_units, world, rank):
    """Independent units (ciphertexts, config 4; batches, config 2) are dealt to ranks in contiguous blocks,
    the first `total % world` ranks taking one more; no data-path collective."""
    base, extra = divmod(total_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher: N ranks under torch.distributed.run as a CHILD process (never a re-exec: this
    process has not touched the GPU, and must not before this point).  Relays the child's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif out.returncode == 0:
        print(out.stdout, flush=True)
    return out.returncode if lines or out.returncode else 1


def dist_setup(backend, device_bound=True):
    """Returns (rank, world, local_rank, dist or None); RANK/WORLD_SIZE/MASTER_* come from the launcher."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        kw = ({"device_id": torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))}
              if backend == "nccl" and device_bound else {})
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank, dist


def max_over_ranks(elapsed, dist, device):
    """Whole-job time = slowest rank."""
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_value(units_per_rank_step, steps, world, elapsed):
    return world * units_per_rank_step * steps / elapsed


def devices_used(torch, dist, world):
    """Distinct GPUs the ranks ran on (two ranks may share one device in a rehearsal on a one-GPU box)."""
    if dist is None or world == 1:
        return 1
    mine = torch.tensor([torch.cuda.current_device() if torch.cuda.is_available() else 0], dtype=torch.int64,
                        device="cuda" if dist.get_backend() == "nccl" else "cpu")
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    return len({int(t.item()) for t in every})


def keyswitch_cpu_baseline(rotate, budget_s=10.0):
    """The oracle's HMult + relinearise (or rotation) at the config-3 shapes on the host cores, bounded sample."""
    import numpy as np

    import oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    oracle.build()
    mods = oracle.coeff_modulus_create(N, [60] + [50] * (LIMBS - 1) + [60] * SIZE_P)
    orc = oracle.Oracle(LOG_N, mods, SIZE_P)
    ct1 = np.ascontiguousarray(orc.uniform_poly(list(range(LIMBS)) * 2, 3).reshape(2, LIMBS, N))
    ct2 = np.ascontiguousarray(orc.uniform_poly(list(range(LIMBS)) * 2, 4).reshape(2, LIMBS, N))
    key_idx = list(range(LIMBS)) + list(range(LIMBS, LIMBS + SIZE_P))
    evk = [np.ascontiguousarray(orc.uniform_poly(key_idx * 2, 5 + d).reshape(2, LIMBS + SIZE_P, N)) for d in range(orc.beta(LIMBS))]
    elt = 5
    done, t0 = 0, time.perf_counter()
    while True:
        if rotate:
            c0 = orc.apply_galois_ntt(ct1[0], elt)
            c1 = orc.apply_galois_ntt(ct1[1], elt)
            ct = np.ascontiguousarray(np.stack([c0, np.zeros_like(c1)]))
            orc.keyswitch_inplace(LIMBS, ct, c1, evk)
        else:
            d = orc.tensor_prod_2x2(ct1, ct2)
            ct = np.ascontiguousarray(d[:2])
            orc.keyswitch_inplace(LIMBS, ct, np.ascontiguousarray(d[2]), evk)
        done += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    what = "rotations (apply_galois + key switch)" if rotate else "HMult + relinearise"
    base = {"cores": threads, "kind": "port", "host_cores_visible": os.cpu_count(),
            "sample": f"{done} {what}, N=2^16, 36 + 12 limbs, dnum 3, in {el:.1f} s, OpenMP over limbs on {threads} threads"}
    if rotate:
        base.update(value=done / el, unit="ciphertexts/s")
    else:
        base.update(value=1e3 * el / done, unit="ms")
    return base


def spin_up(step, ms, sync, fixed_steps=0):
    """Set-up work, not part of the contract's W + K steps: a GPU that has just been handed to the process runs its first
    milliseconds at idle clocks; with K = 20 steps of 70 us the timed region would otherwise measure that ramp.
    fixed_steps: a step with collectives must run the same number of times on every rank -- a count instead of a duration."""
    if ms <= 0:
        return
    if fixed_steps:
        for _ in range(fixed_steps):
            step()
        sync()
        return
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        for _ in range(20):
            step()
        sync()


def timed_steps(step, steps, warmup, dist, sync, device):
    """The contract's timed region: W untimed steps, then exactly K steps between barrier + synchronize on both
    sides; returns the maximum over ranks of the elapsed seconds."""
    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    return max_over_ranks(time.perf_counter() - t0, dist, device)


def event_time_ms(torch, fn, reps):
    """Average duration of fn() over reps back-to-back launches, HIP events on the current stream (the stream the
    library launches on: the bindings pass torch's current stream)."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def event_time_each_ms(torch, fn, reps):
    """Median of per-launch event pairs."""
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in pairs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in pairs)
    return ts[len(ts) // 2]


def cpu_baseline(mods, budget_s=12.0):
    """Oracle forward+inverse over the same batch shape on the host cores (bounded sample).  Threads: the box's CPU
    share (OMP_NUM_THREADS as the harness sets it, 16 per GPU), never more than the machine has."""
    import numpy as np

    import oracle
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    oracle.build()
    orc = oracle.Oracle(LOG_N, mods, 0)
    buf = np.ascontiguousarray(orc.uniform_poly(list(range(LIMBS)), 2))
    orc.nwt_forward_inplace(buf, LIMBS, 0)  # warm-up
    orc.nwt_backward_inplace(buf, LIMBS, 0)
    done, t0 = 0, time.perf_counter()
    while True:
        orc.nwt_forward_inplace(buf, LIMBS, 0)
        orc.nwt_backward_inplace(buf, LIMBS, 0)
        done += 2 * LIMBS
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {"value": done / el, "unit": "limb-transforms/s", "cores": threads, "kind": "port",
            "host_cores_visible": os.cpu_count(),
            "sample": f"{done} limb-transforms (36-limb poly, N=2^16, forward+inverse passes) in {el:.1f} s, "
                      f"OpenMP over limbs on {threads} threads (the box's CPU share for one GPU)"}


def secondary_metrics(torch, pha):
    """HMult+relin / rotate / rescale at config-3 shapes (N=2^16, 36 + 12 limbs, dnum 3) and one sparse-slot
    bootstrap (N=2^16, 64 slots), single GPU, event-timed."""
    import numpy as np
    res = {}
    size_q, size_p = 36, SIZE_P
    mods = pha.coeff_modulus_create(N, [60] + [50] * (size_q - 1) + [60] * size_p)
    ctx = pha.PhantomContext(LOG_N, mods, size_p)
    g = torch.Generator(device="cuda").manual_seed(1)
    lo = int(mods.min())
    rnd = lambda *shape: torch.randint(0, lo, (*shape, N), dtype=torch.int64, device="cuda", generator=g)  # noqa: E731
    evk = pha.EvalKey([rnd(2, size_q + size_p) for _ in range(size_q // size_p)])
    ct1, ct2 = rnd(3, size_q), rnd(2, size_q)
    dst = torch.empty((2, size_q - 1, N), dtype=torch.int64, device="cuda")

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return event_time_ms(torch, fn, reps)

    res["hmult_relin_ms"] = timed(lambda: ctx.multiply_and_relin_inplace(1, ct1, ct2, evk))
    res["rotate_ms"] = timed(lambda: ctx.apply_galois_inplace(1, ct2, 5, evk))
    res["rescale_ms"] = timed(lambda: ctx.rescale_to_next(1, ct2, 2, dst))
    # ~0.9 GiB of algorithmic traffic per HMult + relinearise at these shapes
    res["hmult_relin_roofline_frac"] = 0.9 * 2 ** 30 / (res["hmult_relin_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    res["config"] = "N=2^16, 36 Q limbs + 12 special primes (dnum 3), uniform synthetic ciphertexts and keys"
    ctx.close()
    del evk, ct1, ct2, dst
    # the same operation on a chain whose 13 wide limbs (q0 and the special primes) are 50-bit too: every transform on
    # the FP64 butterflies
    mods = pha.coeff_modulus_create(N, [50] * (size_q + size_p))
    ctx = pha.PhantomContext(LOG_N, mods, size_p)
    lo = int(mods.min())
    evk = pha.EvalKey([rnd(2, size_q + size_p) for _ in range(size_q // size_p)])
    ct1, ct2 = rnd(3, size_q), rnd(2, size_q)
    res["hmult_relin_all_50bit_ms"] = timed(lambda: ctx.multiply_and_relin_inplace(1, ct1, ct2, evk))
    ctx.close()
    del evk, ct1, ct2

    from phantom_fhe_amd.bootstrap import Bootstrapper
    from phantom_fhe_amd.ckks import GpuEvaluator
    d, alpha, slots = 50, 4, 64
    q_bits = [d + 10] + [d] * 4 + [d] + [d] * 10 + [d, 60]
    while len(q_bits) % alpha:
        q_bits.insert(1, d)
    mods = pha.coeff_modulus_create(N, q_bits + [60] * alpha)
    ctx = pha.PhantomContext(LOG_N, mods, alpha)
    ev = GpuEvaluator(pha, ctx, ctx.gen_secretkey(21, hamming_weight=192), slots)
    bs = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** d)
    msg = np.random.default_rng(2).uniform(-1, 1, slots) + 1j * np.random.default_rng(3).uniform(-1, 1, slots)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom))
    out = bs.bootstrap(ct)
    res["bootstrap_max_abs_error"] = float(np.abs(ev.decrypt_decode(out) - msg).max())
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bs.bootstrap(ct)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    res["bootstrap_ms"] = 1e3 * min(times)
    res["bootstrap_config"] = f"N=2^16, {slots} sparse slots, {len(q_bits)} Q limbs + {alpha} special primes, h=192, scale 2^{d}"
    ctx.close()
    del ev, bs, ct, out

    # full slots (BASELINE configs[4]'s primitive): 32768 slots, four levels per linear transform, slim order (SlotToCoeff first,
    # on the last levels), 9 levels left between bootstrappings; both chains inside the 1761-bit bound for N = 2^16:
    #   precise: messages 2^46, EvalMod 2^58, 28 + 4 limbs with 60-bit special primes (dnum 7), log2(QP) = 1714
    #   dnum 4 : messages 2^40, EvalMod 2^48, 28 + 7 limbs, every prime at most 50 bits (FP64 transform path), log2(QP) = 1592
    for tag, d, e, alpha, special_bits in (("", 46, 58, 4, 60), ("_dnum4", 40, 48, 7, 50)):
        slots, groups = N // 2, 4
        q_bits = [d + 10] + [d] * 9 + [d] * groups + [e] * 10 + [min(60, d + 8) if not tag else e] * groups
        while len(q_bits) % alpha:
            q_bits.insert(1, d)
        mods = pha.coeff_modulus_create(N, q_bits + [special_bits] * alpha)
        ctx = pha.PhantomContext(LOG_N, mods, alpha)
        ev = GpuEvaluator(pha, ctx, ctx.gen_secretkey(21, hamming_weight=192), slots)
        bs = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=groups,
                          s2c_groups=groups, slim=True)
        msg = np.random.default_rng(4).uniform(-1, 1, slots) + 1j * np.random.default_rng(5).uniform(-1, 1, slots)
        ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom - bs.pre_levels))
        out = bs.bootstrap(ct)
        res[f"bootstrap_full_slots{tag}_max_abs_error"] = float(np.abs(ev.decrypt_decode(out) - msg).max())
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bs.bootstrap(ct)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        res[f"bootstrap_full_slots{tag}_ms"] = 1e3 * min(times)
        res[f"bootstrap_full_slots{tag}_config"] = (f"N=2^16, {slots} slots, {len(q_bits)} Q limbs + {alpha} special primes of {special_bits} bits, h=192, "
                                                    f"message scale 2^{d}, EvalMod at 2^{e}, SlotToCoeff first (slim order), "
                                                    f"{ev.limbs(out.level) - 1 - bs.pre_levels} levels left between bootstrappings, log2(QP) = "
                                                    f"{float(sum(np.log2(np.array(mods, dtype=np.float64)))):.0f}")
        ctx.close()
        del ev, bs, ct, out

    # BASELINE configs[4]: bootstrapping driving the ResNet-20 / CIFAR-10 example (tools/bench_resnet20.py: the
    # reference's pretrained parameters, a synthetic image, logits checked against the plain network)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import bench_resnet20
    rn = bench_resnet20.run(bench_resnet20.parse(["--reps", "2"]))
    res["resnet20_cifar10_ms"] = rn["value"]
    res["resnet20_cifar10_max_logit_error"] = rn["max_logit_error_vs_plain_same_relu"]
    res["resnet20_cifar10_config"] = (f"N=2^16, {rn['q_limbs']} Q limbs + {rn['special_primes']} special primes, log2(QP) = {rn['log_qp']:.0f}, "
                                      f"{rn['bootstrappings']} bootstrappings, ReLU degrees {rn['relu']['degrees']}, {rn['rotation_keys']} rotation keys")
    return res


def roofline_config2(torch, ctx, buf, orig, mods, args):
    """Live measurements behind the `roofline` object (rank 0, after the timed region)."""
    reps = max(100, min(args.steps, 1000))
    phase = {}
    for name, inv, ph in (("fwd_col", 0, 0), ("fwd_row", 0, 1), ("inv_row", 1, 1), ("inv_col", 1, 0)):
        fn = (lambda inv=inv, ph=ph: ctx.ntt_launch_phase(buf, POLYS, LIMBS, LIMBS, 0, inv, ph))
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = event_time_ms(torch, fn, reps)
        phase[name] = {"avg_launch_ms": ms, "median_ms_per_launch_events": event_time_each_ms(torch, fn, min(reps, 200)),
                       # each kernel reads and writes every limb once: its own streaming rate
                       "stream_GBs": POLYS * LIMBS * ALGO_BYTES_PER_LIMB / (ms * 1e-3) / 1e9}
    buf.copy_(orig)
    fwd = lambda: ctx.ntt_forward_batched(buf, POLYS, LIMBS, LIMBS, 0)  # noqa: E731
    for _ in range(10):
        fwd()
    torch.cuda.synchronize()
    fwd_ms = event_time_ms(torch, fwd, reps)
    buf.copy_(orig)
    algo = POLYS * LIMBS * ALGO_BYTES_PER_LIMB
    achieved = algo / (fwd_ms * 1e-3) / 1e9
    traffic, note = None, None
    try:
        with open(args.pmc_traffic_json) as f:
            tj = json.load(f)
        if tj.get("source_hash") == kernel_source_hash():
            traffic = tj["fwd_col"] + tj["fwd_row"]
            note = (f"bytes per forward transform from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of these kernel "
                    f"sources ({os.path.basename(args.pmc_traffic_json)}, tag {tj.get('tag')}, tools/profile.sh), not counted in this run")
        else:
            note = "profiles/traffic_latest.json was measured on other kernel sources (source_hash differs): not reported"
    except Exception as e:  # noqa: BLE001
        note = f"no usable traffic file: {e!r}"[:160]
    n_fp = sum(1 for q in mods if int(q) >> 50 == 0) * POLYS
    n_int = POLYS * LIMBS - n_fp
    t_int = n_int * BUTTERFLIES_PER_LIMB * MUL32_PER_INT_BUTTERFLY / INT32_MUL_PEAK
    t_fp = n_fp * BUTTERFLIES_PER_LIMB * FP64_PER_FP_BUTTERFLY / FP64_PEAK
    roof = {
        "bound": "hbm", "kernel": "forward transform = fntt_col_kernel<8,4> + fntt_row_kernel<8,4,0>, 72 limbs per launch",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "algorithmic_bytes_per_launch": algo, "avg_launch_ms": fwd_ms,
        "kernels": phase,
        "alu": {"limbs_on_int32_path": n_int, "limbs_on_fp64_path": n_fp,
                "mul32_lane_ops": n_int * BUTTERFLIES_PER_LIMB * MUL32_PER_INT_BUTTERFLY,
                "fp64_lane_ops": n_fp * BUTTERFLIES_PER_LIMB * FP64_PER_FP_BUTTERFLY,
                "int32_mul_peak_lane_ops_s": INT32_MUL_PEAK, "fp64_peak_lane_ops_s": FP64_PEAK,
                "issue_floor_ms": 1e3 * (t_int + t_fp), "frac_of_alu_peak": (t_int + t_fp) / (fwd_ms * 1e-3)},
    }
    if note:
        roof["traffic_note"] = note
    if args.no_secondary:   # profiling runs (tools/profile.sh): the kernel statistics must hold the headline chain's launches only
        return roof
    # the same transform with every limb on the integer butterflies (the reference's grids are 55/60-bit chains,
    # tests/relin.cu:98-130, and q0 + the special primes of configs[2] are 60-bit)
    import phantom_fhe_amd as pha
    imods = pha.coeff_modulus_create(N, [60] * LIMBS)
    ictx = pha.PhantomContext(LOG_N, imods, 0)
    g = torch.Generator(device="cuda").manual_seed(11)
    ibuf = torch.randint(0, int(imods.min()), (POLYS, LIMBS, N), dtype=torch.int64, device="cuda", generator=g)
    ifwd = lambda: ictx.ntt_forward_batched(ibuf, POLYS, LIMBS, LIMBS, 0)  # noqa: E731
    for _ in range(10):
        ifwd()
    torch.cuda.synchronize()
    ims = event_time_ms(torch, ifwd, reps)
    ictx.close()
    roof["int_path"] = {"chain": "36 primes of 60 bits", "avg_launch_ms": ims, "achieved": algo / (ims * 1e-3) / 1e9,
                        "frac": algo / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_of_int32_mul_peak": POLYS * LIMBS * BUTTERFLIES_PER_LIMB * MUL32_PER_INT_BUTTERFLY / INT32_MUL_PEAK / (ims * 1e-3)}
    return roof


def run_config2(args, torch, pha, rank, world, dist):
    import numpy as np
    mods = pha.coeff_modulus_create(N, BITS)
    ctx = pha.PhantomContext(LOG_N, mods, 0)
    # synthetic input: uniform residues, seeded per rank (data: synthetic)
    rng = np.random.default_rng(2 + rank)
    host = np.stack([np.stack([rng.integers(0, int(q), size=N, dtype=np.uint64) for q in mods]) for _ in range(POLYS)])
    buf = pha.to_device(host)
    orig = buf.clone()

    def step():
        ctx.ntt_forward_batched(buf, POLYS, LIMBS, LIMBS, 0)
        ctx.ntt_backward_batched(buf, POLYS, LIMBS, LIMBS, 0)

    spin_up(step, args.spinup_ms, torch.cuda.synchronize)
    elapsed = timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    # size-independent property at full size: the round trips returned the input bit-for-bit
    assert torch.equal(buf, orig), "INTT(NTT(x)) != x after the timed region"
    units_per_step = 2 * POLYS * LIMBS
    out = {
        "metric": "RNS-NTT limb-transforms/s @N=2^16 (forward+inverse, 36 limbs x 2 polys per GPU)",
        "value": aggregate_value(units_per_step, args.steps, world, elapsed), "unit": "limb-transforms/s",
        "n_gpus": devices_used(torch, dist, world), "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[1]: batched RNS NTT+INTT, N=2^16, 36 limbs x 2 polys, uint64[2][36][65536] per GPU",
                   "limbs": LIMBS, "polys": POLYS, "log_n": LOG_N, "parallelism": f"independent batches x{world}"},
    }
    if rank == 0:
        out["roofline"] = roofline_config2(torch, ctx, buf, orig, mods, args)
        if not args.no_secondary and world == 1:
            # the other headline numbers of BASELINE.json (HMult+relin, rotate, bootstrap) on this GPU; they do
            # not enter `value`
            try:
                ctx.close()
                ctx = None
                out["secondary"] = secondary_metrics(torch, pha)
            except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline
                out["secondary"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mods)
    if ctx is not None:
        ctx.close()
    return out


def run_config3(args, torch, pha, rank, world, dist):
    """HMult + relinearise with the limbs sharded over the ranks (strong scaling; RCCL all-gathers)."""
    from phantom_fhe_amd.limb_parallel import LimbParallelEvaluator, ThreadComm, TorchDistComm
    mods = pha.coeff_modulus_create(N, [60] + [50] * (LIMBS - 1) + [60] * SIZE_P)
    ctx = pha.PhantomContext(LOG_N, mods, SIZE_P)
    comm = TorchDistComm() if dist is not None else ThreadComm(ThreadComm.Shared(1), 0)
    ev = LimbParallelEvaluator(ctx, comm)
    g = torch.Generator(device="cuda").manual_seed(7)      # the same ciphertexts and key on every rank
    lo = int(mods.min())
    rnd = lambda *shape: torch.randint(0, lo, shape, dtype=torch.int64, device="cuda", generator=g)  # noqa: E731
    ct1, ct2 = rnd(2, LIMBS, N), rnd(2, LIMBS, N)
    out3 = torch.empty((3, LIMBS, N), dtype=torch.int64, device="cuda")
    evk = pha.EvalKey([rnd(2, LIMBS + SIZE_P, N) for _ in range(ctx.beta(1))])

    def step():
        ev.multiply_and_relin(1, ct1, ct2, evk, out3)

    spin_up(step, args.spinup_ms, torch.cuda.synchronize, fixed_steps=100)   # all-gathers inside: same count on every rank
    elapsed = timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    q0, qc, p0, pc = ev.windows(1)[rank]
    ndev = devices_used(torch, dist, world)
    out = {
        "metric": "CKKS HMult + relinearise ms @N=2^16, L=35 (36 + 12 limbs, dnum 3), limbs sharded over the GPUs",
        "value": 1e3 * elapsed / args.steps, "unit": "ms", "n_gpus": ndev, "ranks": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[2]: HMult + relinearise (hybrid key switch, dnum 3), N=2^16, 36 Q + 12 P limbs",
                   "parallelism": f"limb-parallel x{world}: balanced limb windows over Q_l u P, 2 in-place all-gathers per key switch",
                   "rank0_window": {"q_begin": q0, "q_count": qc, "p_begin": p0, "p_count": pc}},
    }
    if rank == 0:
        # ~0.9 GiB of algorithmic traffic per HMult + relinearise, against 8 TB/s on every GPU that took part
        ach = KS_ALGO_BYTES / (elapsed / args.steps) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "HMult + relinearise, whole operation (all ranks)", "achieved": ach,
                           "peak": HBM_PEAK_GBS * ndev, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * ndev), "traffic": None,
                           "algorithmic_bytes_per_launch": KS_ALGO_BYTES}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = keyswitch_cpu_baseline(rotate=False)
    ctx.close()
    return out


def run_config4(args, torch, pha, rank, world, dist):
    """Rotation of 64 independent ciphertexts dealt to the ranks (strong scaling; no collective)."""
    total, batch = 64, 16
    mods = pha.coeff_modulus_create(N, [60] + [50] * (LIMBS - 1) + [60] * SIZE_P)
    ctx = pha.PhantomContext(LOG_N, mods, SIZE_P)
    lo_u, hi_u = shard_units(total, world, rank)
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    small = int(mods.min())
    rnd = lambda *shape: torch.randint(0, small, shape, dtype=torch.int64, device="cuda", generator=g)  # noqa: E731
    cts = [rnd(2, LIMBS, N) for _ in range(lo_u, hi_u)]
    groups = [torch.stack(cts[i:i + batch]) for i in range(0, len(cts), batch)]
    evk = pha.EvalKey([rnd(2, LIMBS + SIZE_P, N) for _ in range(ctx.beta(1))])
    elt = pha.galois_elt_from_step(LOG_N, 1)

    def step():
        for grp in groups:
            ctx.apply_galois_batched(1, grp, elt, evk)

    spin_up(step, args.spinup_ms, torch.cuda.synchronize, fixed_steps=3)
    elapsed = timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    out = {
        "metric": "CKKS rotate (apply_galois + key switch) ciphertexts/s, 64 independent ciphertexts @N=2^16, 36 + 12 limbs",
        "value": total * args.steps / elapsed, "unit": "ciphertexts/s", "n_gpus": devices_used(torch, dist, world), "ranks": world,
        "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "configs[3]: rotate by one slot, 64 independent ciphertexts, N=2^16, 36 Q + 12 P limbs (dnum 3)",
                   "parallelism": f"ciphertexts dealt to {world} ranks ({hi_u - lo_u} on rank 0), batches of {batch} per launch sequence"},
    }
    if rank == 0:
        # a rotation = key switch + two gathers: SURVEY.md §8(d)'s 0.9 GiB minus the tensor product's 126 MiB plus 72 MiB
        algo = KS_ALGO_BYTES - 126 * 2 ** 20 + 72 * 2 ** 20
        ach = algo * out["value"] / 1e9
        ndev = out["n_gpus"]
        out["roofline"] = {"bound": "hbm", "kernel": "rotate = Galois gather + key switch, whole operation (all ranks)", "achieved": ach,
                           "peak": HBM_PEAK_GBS * ndev, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * ndev), "traffic": None,
                           "algorithmic_bytes_per_launch": algo}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = keyswitch_cpu_baseline(rotate=True)
    ctx.close()
    return out


def run_dry(args, rank, world, dist):
    """Launcher rehearsal on CPU (tests/test_dist_cpu.py): same rendezvous, barrier, max-over-ranks and JSON line."""
    elapsed = timed_steps(lambda: None, args.steps, args.warmup, dist, lambda: None, "cpu")
    lo, hi = shard_units(64, world, rank)
    return {"metric": "dry run (no GPU work)", "value": aggregate_value(1, args.steps, world, max(elapsed, 1e-9)), "unit": "steps/s",
            "n_gpus": world, "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "none", "rank0_units_of_64": hi - lo}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    if args.dry_run:
        rank, world, local_rank, dist = dist_setup(args.backend, device_bound=False)
        out = run_dry(args, rank, world, dist)
    else:
        import torch

        import phantom_fhe_amd as pha

        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        # this rank's GPU is current before the process group (RCCL communicator, barriers) comes up on it
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        rank, world, local_rank, dist = dist_setup(args.backend)
        out = {2: run_config2, 3: run_config3, 4: run_config4}[args.config](args, torch, pha, rank, world, dist)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
