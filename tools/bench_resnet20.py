#!/usr/bin/env python3
"""Encrypted ResNet-20 / CIFAR-10 inference at N = 2^16 on one GPU (BASELINE config 5: bootstrapping driving the
reference's examples/cnn.cu).  Pretrained parameters: tests/golden/resnet20_cifar10.npz (the reference's
pretrained_parameters/resnet20_new); the image is synthetic (the reference's CIFAR-10 test file is not in its
repository).  Checks the decrypted logits against the plain network (oracle/cnn_ref.py) and prints one JSON line.

    python tools/bench_resnet20.py [--layers 20] [--reps 2] [--precise] [--phases]

Default chain (log2 QP = 1750 <= 1761): q0 48 bits, 18 levels of 38 bits (ReLU 14, convolution 2, SlotToCoeff 2), EvalMod
10 x 48, CoeffToSlot 3 x 46, EIGHT special primes of 50 bits: dnum = 4, and every prime below 2^50 takes the FP64
transform path.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import phantom_fhe_amd as pha
from phantom_fhe_amd import cnn
from phantom_fhe_amd.bootstrap import Bootstrapper
from phantom_fhe_amd.ckks import GpuEvaluator
from oracle import cnn_ref


def smooth_image(seed, shape=(3, 32, 32)):
    """a seeded low-frequency random field with the statistics of a normalised CIFAR-10 image (|pixel| < 2.5)"""
    rng = np.random.default_rng(seed)
    c, h, w = shape
    f = np.fft.fft2(rng.normal(size=shape))
    ky, kx = np.meshgrid(np.fft.fftfreq(h), np.fft.fftfreq(w), indexing="ij")
    img = np.real(np.fft.ifft2(f / (1.0 + 40.0 * (kx ** 2 + ky ** 2) * h)))
    img = img / img.std(axis=(1, 2), keepdims=True) + rng.normal(0.0, 0.3, (c, 1, 1))
    return np.clip(img, -2.5, 2.5)


def build(args):
    relu = cnn.default_relu()
    d, e, g = args.delta_bits, args.evalmod_bits, args.groups
    g2 = args.s2c_groups or g
    q_bits = [d + 10] + [d] * (relu.depth + 2) + [d] * g2 + [e] * 10 + [min(60, d + 8)] * g
    while len(q_bits) % args.alpha:
        q_bits.insert(1, d)
    n = 1 << args.log_n
    mods = pha.coeff_modulus_create(n, q_bits + [args.special_bits] * args.alpha)
    ctx = pha.PhantomContext(args.log_n, mods, args.alpha)
    sk = ctx.gen_secretkey(21, hamming_weight=192)
    ev = GpuEvaluator(pha, ctx, sk, n // 2)
    ev.auto_keys = True
    boot = {}
    for s in (n // 4, n // 8, n // 16):
        boot[s] = Bootstrapper(ev, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=g, s2c_groups=g2, slots=s, slim=not args.no_slim)
    return ctx, ev, boot, relu, mods


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--alpha", type=int, default=8)
    ap.add_argument("--groups", type=int, default=3, help="levels of CoeffToSlot")
    ap.add_argument("--s2c-groups", type=int, default=2, help="levels of SlotToCoeff (0: --groups)")
    ap.add_argument("--delta-bits", type=int, default=38)
    ap.add_argument("--evalmod-bits", type=int, default=48)
    ap.add_argument("--special-bits", type=int, default=50, help="size of the special primes (below 2^50 they take the FP64 transform path)")
    ap.add_argument("--image-seed", type=int, default=0)
    ap.add_argument("--params", default="", help="npz of parameters for --layers (tools/make_resnet20_fixture.py <layers> <out.npz>)")
    ap.add_argument("--no-slim", action="store_true", help="SlotToCoeff after EvalMod (the usual order) instead of before the modulus raise")
    ap.add_argument("--precise", action="store_true", help="42-bit messages, 56-bit EvalMod primes, four 60-bit special primes (dnum 8): "
                    "logit errors of 2e-5 instead of 9e-4, 589 ms instead of 502 ms")
    ap.add_argument("--graph", action="store_true", help="also capture the inference in one HIP graph and time its replay")
    ap.add_argument("--streams", type=int, default=1, help="2: a second, independent inference on a sibling (context, stream) pair, both as HIP graphs (throughput)")
    ap.add_argument("--phases", action="store_true", help="synchronise after every block and print where the time goes")
    args = ap.parse_args(argv)
    if args.precise:
        args.alpha, args.delta_bits, args.evalmod_bits, args.special_bits = 4, 42, 56, 60
    return args


def run(args):
    # one process per GPU under torch.distributed.run: every rank runs its own image (replicas: the network does not
    # shard, SURVEY.md 8e "one image per GPU"); no data-path collective, the timing is barrier + max over ranks
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        args.image_seed += rank
    t0 = time.perf_counter()
    ctx, ev, boot, relu, mods = build(args)
    if args.params:
        params = cnn.load_parameters_npz(args.params)
    elif args.layers == 20:
        params = cnn.load_parameters_npz(os.path.join(ROOT, "tests", "golden", "resnet20_cifar10.npz"))
    else:
        params = cnn_ref.synthetic_parameters({8: 0, 14: 1, 32: 4, 44: 6, 56: 8, 110: 17}[args.layers], seed=1)
    net = cnn.ResNetCifar10(ev, params, layer_num=args.layers, B=40.0, relu=relu, bootstrappers=boot)
    image = smooth_image(args.image_seed)
    level = net.input_level()
    x = net.encrypt_image(image, 2.0 ** args.delta_bits, level)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    seen = {}
    net.trace = lambda tag, t: seen.__setitem__(tag, net.B * cnn.unpack_tensor(np.real(ev.decrypt_decode(t.ct, 1 << t.logn)), t.k, t.h, t.w, t.c))
    logits = net.decrypt_logits(net(x))                          # first run: encodes the constants, generates rotation keys
    torch.cuda.synchronize()
    first_s = time.perf_counter() - t0
    net.trace = None
    want_seen = {}
    B = net.B
    want = cnn_ref.resnet_cifar10(params, image, net.end_num, relu=lambda v: B * relu(v / B), trace=lambda tag, t: want_seen.__setitem__(tag, t))
    exact = cnn_ref.resnet_cifar10(params, image, net.end_num)
    block_err = {tag: float(np.abs(seen[tag] - want_seen[tag]).max()) for tag in want_seen}
    times = []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        out = net(x)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        times.append(dt)
    again = net.decrypt_logits(out)
    key_bytes = sum(k.numel() * 8 for k in ev.galois.values()) if ev.galois and hasattr(next(iter(ev.galois.values())), "numel") else None
    res = {"metric": "encrypted ResNet-%d CIFAR-10 inference ms" % args.layers, "value": 1e3 * min(times), "unit": "ms", "log_n": args.log_n,
           "q_limbs": len(mods) - args.alpha, "special_primes": args.alpha, "log_qp": round(float(sum(np.log2(np.array(mods, dtype=np.float64)))), 1),
           "delta_bits": args.delta_bits, "evalmod_bits": args.evalmod_bits, "bootstrappings": 6 * (net.end_num + 1),
           "relu": {"degrees": list(relu.degrees), "alpha": round(relu.alpha, 2), "levels": relu.depth},
           "logits": [round(float(v), 4) for v in logits], "logits_plain_same_relu": [round(float(v), 4) for v in want],
           "max_logit_error_vs_plain_same_relu": float(np.abs(logits - want).max()), "max_logit_error_vs_exact_relu": float(np.abs(logits - exact).max()),
           "argmax": int(np.argmax(logits)), "argmax_plain": int(np.argmax(exact)), "max_block_error": max(block_err.values()),
           "max_activation_over_B": float(max(np.abs(v).max() for v in want_seen.values()) / B),
           "repeat_logit_diff": float(np.abs(again - logits).max()),
           "rotation_keys": len(ev.galois), "rotation_key_bytes": key_bytes, "encoded_constants": sum(1 for v in net.cache.store.values() if v is not None and v is not True),
           "setup_s": round(setup_s, 2), "first_inference_s": round(first_s, 2), "n_gpus": world, "images_per_s": world / min(times), "scaling": "weak",
           "data": ("reference's pretrained ResNet-%d parameters, synthetic image" % args.layers) if (args.layers == 20 or args.params) else "synthetic parameters and image"}
    if args.graph:
        t0 = time.perf_counter()
        run = net.capture(x)
        torch.cuda.synchronize()
        res["graph_capture_s"] = round(time.perf_counter() - t0, 2)
        gt = []
        for _ in range(max(2, args.reps)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gout = run(x)
            torch.cuda.synchronize()
            gt.append(time.perf_counter() - t0)
        res["graph_replay_ms"] = 1e3 * min(gt)
        res["graph_logit_diff"] = float(np.abs(net.decrypt_logits(gout) - logits).max())
    if args.streams == 2:
        # two images in flight: the second on a sibling evaluator (same keys, own library context and stream) with its
        # own bootstrappers and constants; both inferences are HIP graphs, replayed side by side
        ev2 = ev.sibling()
        ev2.auto_keys = True
        d, e, g, g2 = args.delta_bits, args.evalmod_bits, args.groups, args.s2c_groups or args.groups
        with torch.cuda.stream(ev2.stream):
            boot2 = {s: Bootstrapper(ev2, K=25, r=4, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=g, s2c_groups=g2, slots=s, slim=not args.no_slim)
                     for s in boot}
            net2 = cnn.ResNetCifar10(ev2, params, layer_num=args.layers, B=40.0, relu=relu, bootstrappers=boot2)
            image2 = smooth_image(args.image_seed + 1)
            x2 = net2.encrypt_image(image2, 2.0 ** args.delta_bits, level)
            logits2 = net2.decrypt_logits(net2(x2))
            run2 = net2.capture(x2)
        run1 = net.capture(x)
        torch.cuda.synchronize()
        pair = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o1 = run1(x)
            with torch.cuda.stream(ev2.stream):
                o2 = run2(x2)
            torch.cuda.synchronize()
            pair.append(time.perf_counter() - t0)
        want2 = cnn_ref.resnet_cifar10(params, image2, net.end_num, relu=lambda v: B * relu(v / B))
        with torch.cuda.stream(ev2.stream):
            got2 = net2.decrypt_logits(o2)
        res["two_streams_ms_for_two_images"] = 1e3 * min(pair)
        res["images_per_s_two_streams"] = 2.0 / min(pair)
        res["images_per_s_one_stream"] = 1.0 / min(times)
        res["two_streams_logit_errors"] = [float(np.abs(net.decrypt_logits(o1) - want).max()), float(np.abs(got2 - want2).max())]
    if args.phases:
        res["block_errors"] = block_err
        spent = {}

        def timed(tag, fn):
            def run(*a, **k):
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = fn(*a, **k)
                torch.cuda.synchronize()
                spent[tag] = spent.get(tag, 0.0) + 1e3 * (time.perf_counter() - t)
                return out
            return run
        for name, tag in (("multiplexed_parallel_convolution", "convolution"), ("multiplexed_parallel_batch_norm", "batch_norm"),
                          ("approx_relu", "relu"), ("multiplexed_parallel_downsampling", "downsampling"), ("cnn_add", "add"),
                          ("averagepooling_scale", "avgpool"), ("fully_connected", "linear")):
            setattr(cnn, name, timed(tag, getattr(cnn, name)))
        net._bootstrap = timed("bootstrap", net._bootstrap)
        net(x)
        res["phase_ms"] = {k: round(v, 2) for k, v in spent.items()}
        # evaluator methods inside the convolutions and the ReLUs (synchronised: sums exceed the phase times a little)
        phase, inner, calls = [None], {}, {}

        def scoped(tag, fn):
            def run(*a, **k):
                phase[0] = tag
                try:
                    return fn(*a, **k)
                finally:
                    phase[0] = None
            return run

        def method(name, fn):
            def run(*a, **k):
                if phase[0] is None:
                    return fn(*a, **k)
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = fn(*a, **k)
                torch.cuda.synchronize()
                key = phase[0] + "." + name
                inner[key] = inner.get(key, 0.0) + 1e3 * (time.perf_counter() - t)
                calls[key] = calls.get(key, 0) + 1
                return out
            return run
        cnn.multiplexed_parallel_convolution = scoped("conv", cnn.multiplexed_parallel_convolution)
        cnn.approx_relu = scoped("relu", cnn.approx_relu)
        for name in ("rotate_many", "mul_plain_add", "rescale", "rotate_batch", "add", "bsgs", "batch", "mul_rescale", "mul", "mul_plain", "add_plain", "sub", "drop"):
            setattr(ev, name, method(name, getattr(ev, name)))
        net(x)
        res["method_ms"] = {k: [round(v, 2), calls[k]] for k, v in sorted(inner.items(), key=lambda kv: -kv[1])}
    ctx.close()
    if dist is not None:
        ok = torch.tensor([1.0 if res["argmax"] == res["argmax_plain"] and res["max_logit_error_vs_plain_same_relu"] < 1e-2 else 0.0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        res["all_ranks_correct"] = bool(ok.item())
        dist.barrier()
        dist.destroy_process_group()
    res["rank"] = rank
    return res


if __name__ == "__main__":
    result = run(parse())
    if result["rank"] == 0:
        print(json.dumps(result), flush=True)
