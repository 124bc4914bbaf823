"""The CNN layer (phantom-fhe_amd/cnn.py; reference: src/cnn_phantom.cu, examples/cnn.cu) on the plaintext-ring
simulator against a plain numpy network (oracle/cnn_ref.py), and on the GPU against the same.  Parity unpinned by
reference outputs (the reference's result logs need CIFAR-10 images it does not hold): the checks are against the
function the circuit must compute."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import phantom_fhe_amd as pha                                    # noqa: E402
from phantom_fhe_amd import cnn                                  # noqa: E402
from oracle.sim_evaluator import SimEvaluator                    # noqa: E402
from oracle import cnn_ref                                       # noqa: E402

LOGN = 10                                                        # slots of the simulator's ring (N = 2^11)


def _sim(levels=8, bits=40):
    return SimEvaluator(LOGN + 1, [2 ** (bits + 10)] + [2 ** bits + 2 * i + 1 for i in range(levels)], 1 << LOGN)


def _tensor(ev, a, k, p, level=1, scale=2.0 ** 40):
    vec = cnn.pack_tensor(a, LOGN, k, p)
    c = a.shape[0]
    return cnn.TensorCipher(LOGN, k, a.shape[1], a.shape[2], c, -(-c // (k * k)), p,
                            ev.encrypt(ev.encode(vec.astype(np.complex128), scale, level, 1 << LOGN)))


def _read(ev, x):
    v = np.real(ev.decrypt_decode(x.ct, 1 << LOGN))
    n = 1 << LOGN
    per = n // x.p
    for u in range(1, x.p):                                       # every copy holds the same vector
        assert np.abs(v[u * per:(u + 1) * per] - v[:per]).max() < 1e-6
    return cnn.unpack_tensor(v, x.k, x.h, x.w, x.c), v


def test_pack_unpack():
    rng = np.random.default_rng(0)
    for k, c, h in ((1, 4, 8), (2, 8, 4), (4, 16, 4), (2, 6, 8)):
        a = rng.normal(size=(c, h, h))
        t = -(-c // (k * k))
        p = (1 << LOGN) // (k * k * h * h * t)
        v = cnn.pack_tensor(a, LOGN, k, p)
        assert np.array_equal(cnn.unpack_tensor(v, k, h, h, c), a)
        assert np.array_equal(v[:(1 << LOGN) // p], v[(1 << LOGN) // p * (p - 1):])
        assert np.count_nonzero(v) == p * a.size


@pytest.mark.parametrize("k,c,h,co,st", [(1, 4, 8, 8, 1), (1, 4, 8, 8, 2), (2, 8, 4, 8, 1), (2, 8, 4, 16, 2), (1, 3, 8, 16, 1), (2, 6, 4, 5, 1)])
def test_convolution_matches_numpy(k, c, h, co, st):
    rng = np.random.default_rng(k * 100 + c + co + st)
    ev = _sim()
    a = rng.normal(size=(c, h, h))
    t = -(-c // (k * k))
    p = (1 << LOGN) // (k * k * h * h * t)
    p = 1 << (p.bit_length() - 1)
    x = _tensor(ev, a, k, p)
    w = rng.normal(size=(co, c, 3, 3))
    var, gamma = rng.uniform(0.5, 2.0, co), rng.uniform(0.5, 1.5, co)
    y = cnn.multiplexed_parallel_convolution(ev, cnn.PlainCache(ev), "c", x, co, st, 3, 3, w.ravel(), var, gamma, 1e-5)
    assert y.ct.level == 3 and abs(y.ct.scale / x.ct.scale - 1) < 1e-12
    assert (y.k, y.h, y.c, y.t) == (k * st, h // st, co, -(-co // (k * st) ** 2))
    got, vec = _read(ev, y)
    want = cnn_ref.conv3x3(a, w, st) * (gamma / np.sqrt(var + 1e-5))[:, None, None]
    assert np.abs(got - want).max() < 1e-7
    assert np.count_nonzero(np.abs(vec) > 1e-7) <= y.p * want.size        # nothing outside the packing


def test_convolution_argument_checks():
    ev = _sim()
    x = _tensor(ev, np.zeros((4, 8, 8)), 1, 4)
    ok = dict(co=8, st=1, fh=3, fw=3, data=np.zeros(8 * 4 * 9), running_var=np.ones(8), constant_weight=np.ones(8), epsilon=1e-5)
    for bad in (dict(st=3), dict(data=np.zeros(7)), dict(running_var=np.ones(7)), dict(running_var=np.zeros(8)), dict(fh=2, data=np.zeros(8 * 4 * 6))):
        with pytest.raises(ValueError):
            cnn.multiplexed_parallel_convolution(ev, cnn.PlainCache(ev), "c", x, **{**ok, **bad})


def test_batch_norm_downsampling_add():
    rng = np.random.default_rng(5)
    ev = _sim()
    a = rng.normal(size=(8, 8, 8))
    x = _tensor(ev, a, 1, 2)
    cache = cnn.PlainCache(ev)
    bias, mean, var, gamma = rng.normal(size=8), rng.normal(size=8), rng.uniform(0.5, 2, 8), rng.uniform(0.5, 1.5, 8)
    B = 40.0
    y = cnn.multiplexed_parallel_batch_norm(ev, cache, "bn", x, bias, mean, var, gamma, 1e-5, B)
    got, _ = _read(ev, y)
    want = a - ((mean * gamma / np.sqrt(var + 1e-5) - bias) / B)[:, None, None]
    assert np.abs(got - want).max() < 1e-8
    d = cnn.multiplexed_parallel_downsampling(ev, cache, "down", x)
    assert (d.k, d.h, d.w, d.c, d.t, d.p) == (2, 4, 4, 16, 4, 4) and d.ct.level == 2
    got, vec = _read(ev, d)
    assert np.abs(got - cnn_ref.shortcut_a(a)).max() < 1e-8
    s = cnn.cnn_add(ev, d, d)
    assert np.abs(_read(ev, s)[0] - 2 * cnn_ref.shortcut_a(a)).max() < 1e-8
    with pytest.raises(ValueError):
        cnn.cnn_add(ev, d, x)
    with pytest.raises(ValueError):
        cnn.multiplexed_parallel_downsampling(ev, cache, "down2", _tensor(ev, a[:4], 1, 4))      # ti % 8


def test_average_pooling_and_fully_connected():
    rng = np.random.default_rng(6)
    ev = _sim()
    a = rng.normal(size=(16, 4, 4))
    x = _tensor(ev, a, 2, 4)
    cache = cnn.PlainCache(ev)
    B = 40.0
    y = cnn.averagepooling_scale(ev, cache, "avg", x, B)
    v = np.real(ev.decrypt_decode(y.ct, 1 << LOGN))
    assert np.abs(v[:16] - B * a.mean(axis=(1, 2))).max() < 1e-7
    assert np.abs(v[16:]).max() < 1e-7
    m, b = rng.normal(size=(10, 16)), rng.normal(size=10)
    z = cnn.fully_connected(ev, cache, "fc", y, m.ravel(), b, 10, 16)
    out = np.real(ev.decrypt_decode(z.ct, 1 << LOGN))[:10]
    assert np.abs(out - (m @ (B * a.mean(axis=(1, 2))) + b)).max() < 1e-6
    assert z.ct.level == 3


def test_approximate_relu():
    relu = cnn.default_relu()
    assert relu.depth == 14 and relu.alpha > 13.0                           # examples/cnn.cu:54-56: alpha = 13, {15, 15, 27}
    ev = _sim(levels=16, bits=46)
    rng = np.random.default_rng(7)
    vals = np.concatenate([rng.uniform(-1, 1, 1000), [0.0, 1e-4, -1e-4, 2.0 ** -9, -2.0 ** -9, 1.0, -1.0]])
    ct = ev.encrypt(ev.encode(vals.astype(np.complex128), 2.0 ** 46, 1, 1 << LOGN))
    out = relu.evaluate(ev, ct)
    assert out.level == 1 + relu.depth and abs(out.scale / ct.scale - 1) < 1e-12
    got = np.real(ev.decrypt_decode(out, 1 << LOGN))[:len(vals)]
    assert np.abs(got - relu(vals)).max() < 1e-9                            # the circuit evaluates the composite
    assert np.abs(got - np.maximum(vals, 0)).max() < 2.0 ** -13             # which approximates the ReLU


def _small_net(ev, boot=None, seed=3):
    params = cnn_ref.synthetic_parameters(0, seed=seed)
    net = cnn.ResNetCifar10(ev, params, layer_num=8, B=16.0, image_shape=(3, 8, 8), bootstrappers=boot)
    image = np.random.default_rng(seed).normal(0.0, 1.0, (3, 8, 8))
    return params, net, image


def test_resnet8_on_the_simulator():
    """three residual blocks (16, 32, 64 channels; gaps 1, 2, 4) on 8 x 8 images, no bootstrapping (a 115-level chain)"""
    relu = cnn.default_relu()
    levels = 2 + relu.depth + 3 * 2 * (2 + relu.depth) + 2
    ev = _sim(levels=levels, bits=46)
    params, net, image = _small_net(ev)
    seen = {}
    x = net.encrypt_image(image, 2.0 ** 46, 1)
    assert (x.k, x.c, x.t, x.p) == (1, 3, 3, 4)
    net.trace = lambda tag, t: seen.__setitem__(tag, net.B * _read(ev, t)[0])
    logits = net.decrypt_logits(net(x))
    want_seen = {}
    B = net.B
    want = cnn_ref.resnet_cifar10(params, image, 0, relu=lambda v: B * relu(v / B), trace=lambda tag, t: want_seen.__setitem__(tag, t))
    for tag in want_seen:
        assert np.abs(want_seen[tag]).max() < B                      # activations inside the approximation range
        assert np.abs(seen[tag] - want_seen[tag]).max() < 1e-6, tag
    assert np.abs(logits - want).max() < 1e-5
    exact = cnn_ref.resnet_cifar10(params, image, 0)
    assert np.abs(logits - exact).max() < 0.05 and np.argmax(logits) == np.argmax(exact)
    # the second inference reuses every encoded constant
    n_plain = len(net.cache.store)
    net.trace = None
    again = net.decrypt_logits(net(x))
    assert len(net.cache.store) == n_plain and np.abs(again - logits).max() < 1e-9


def test_parameter_files(tmp_path):
    params = cnn_ref.synthetic_parameters(2, seed=1)
    assert sorted(params) == sorted(cnn.parameter_names(20))
    for k, v in params.items():
        np.savetxt(tmp_path / (k + ".txt"), v)
    back = cnn.import_parameters_cifar10(str(tmp_path), 20)
    assert all(np.array_equal(back[k], params[k]) for k in params)
    np.savez(tmp_path / "p.npz", **params)
    back = cnn.load_parameters_npz(str(tmp_path / "p.npz"))
    assert all(np.array_equal(back[k], params[k]) for k in params)


@pytest.mark.gpu
@pytest.mark.parametrize("slim,d,e,alpha,special", [(True, 42, 56, 4, 60), (False, 42, 56, 4, 60), (True, 38, 48, 8, 50)])
def test_resnet8_with_bootstrapping_on_the_gpu(slim, d, e, alpha, special):
    """N = 2^13, 8 x 8 images: three residual blocks, six bootstrappings (1024, 512 and 256 sparse slots), every
    ciphertext operation through the C ABI; logits against the plain network with the same ReLU polynomial."""
    import torch
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from phantom_fhe_amd.ckks import GpuEvaluator
    log_n, g = 13, 3
    relu = cnn.default_relu()
    q_bits = [d + 10] + [d] * (relu.depth + 2) + [d] * g + [e] * 9 + [d + 8] * g
    assert len(q_bits) % alpha == 0
    n = 1 << log_n
    mods = pha.coeff_modulus_create(n, q_bits + [special] * alpha)     # 50-bit special primes: every limb on the FP64 path, dnum 4
    ctx = pha.PhantomContext(log_n, mods, alpha)
    sk = ctx.gen_secretkey(21, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, n // 2)
    ev.auto_keys = True
    boot = {s: Bootstrapper(ev, K=12, r=3, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=g, s2c_groups=g, slots=s,
                            slim=slim) for s in (n // 8, n // 16, n // 32)}
    params = cnn_ref.synthetic_parameters(0, seed=3)
    net = cnn.ResNetCifar10(ev, params, layer_num=8, B=16.0, relu=relu, image_shape=(3, 8, 8), bootstrappers=boot)
    image = np.random.default_rng(3).normal(0.0, 1.0, (3, 8, 8))
    x = net.encrypt_image(image, 2.0 ** d, net.input_level())
    out = net(x)
    assert out.ct.level == next(iter(boot.values())).out_level + relu.depth + 2      # average pooling and the linear layer: two levels
    logits = net.decrypt_logits(out)
    B = net.B
    want = cnn_ref.resnet_cifar10(params, image, 0, relu=lambda v: B * relu(v / B))
    exact = cnn_ref.resnet_cifar10(params, image, 0)
    assert np.abs(logits - want).max() < 1e-2, (logits, want)
    assert np.argmax(logits) == np.argmax(exact)
    # one HIP graph of the whole inference gives the same ciphertext
    run = net.capture(x)
    again = net.decrypt_logits(run(x))
    assert np.array_equal(again, logits)
    torch.cuda.synchronize()
    ctx.close()


def test_relu_by_height_equals_the_recursion():
    """ChebyshevEvaluator.batched: every node of one height at one level (batched products) computes the same
    polynomial as the depth-first recursion, at the same level and scale"""
    relu = cnn.ApproxReLU()
    ev = _sim(levels=16, bits=46)
    vals = np.random.default_rng(8).uniform(-1, 1, 1000)
    ct = ev.encrypt(ev.encode(vals.astype(np.complex128), 2.0 ** 46, 1, 1 << LOGN))
    relu.batched, relu.parallel = False, False
    a = relu.evaluate(ev, ct)
    relu.batched = True
    b = relu.evaluate(ev, ct)
    assert a.level == b.level and abs(a.scale / b.scale - 1) < 1e-12
    assert np.abs(ev.decrypt_decode(a, 1 << LOGN) - ev.decrypt_decode(b, 1 << LOGN)).max() < 1e-9


@pytest.mark.gpu
def test_resnet20_cifar10_at_the_stated_size():
    """BASELINE configs[4] at its stated size (N = 2^16, ResNet-20, 18 bootstrappings; tools/bench_resnet20.py builds the
    chain, the keys and the network from tests/golden/resnet20_cifar10.npz): the encrypted logits against the plain network
    with the same ReLU polynomial, and the same class as the exact network.  ~55 GiB of rotation keys."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2 ** 30:
        pytest.skip("needs ~70 GiB of device memory")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_resnet20
    res = bench_resnet20.run(bench_resnet20.parse(["--reps", "1"]))
    assert res["max_logit_error_vs_plain_same_relu"] < 5e-3, res
    assert res["argmax"] == res["argmax_plain"], res
    torch.cuda.empty_cache()
