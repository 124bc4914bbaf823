"""Sparse-slot CKKS bootstrapping (src/boot/Bootstrapper.cu bootstrap_sparse*):
the circuit (phantom-fhe_amd/bootstrap.py) runs on the plaintext-ring simulator on the CPU — which
checks the algebra, the level schedule and the exact scale bookkeeping — and on real ciphertexts
through the C ABI on the GPU, where a ciphertext at the last level must come back at a high level
decrypting to the same message (the reference's tests/bootstrapping_test.cu asks for ~1e-3;
here the tolerance is 1e-3 of the message magnitude too, the measured error is far below)."""
import numpy as np
import pytest

N, LOG_N = 4096, 12
# q0, two levels left after bootstrapping, SlotToCoeff, 9 EvalMod levels, CoeffToSlot (two primes)
Q_BITS = [50] + [40] * 2 + [40] + [40] * 9 + [40, 58]


def _chain(orc, extra=()):
    return [int(v) for v in orc.coeff_modulus_create(N, Q_BITS + list(extra))]


def test_chebyshev_and_linear_transform_on_the_simulator(orc, pha):
    from phantom_fhe_amd.bootstrap import ChebyshevEvaluator, LinearTransform
    from oracle.sim_evaluator import SimEvaluator
    ev = SimEvaluator(LOG_N, _chain(orc), 16, seed=1)
    rng = np.random.default_rng(0)
    w = rng.uniform(-1, 1, 16)
    for degree, f in ((31, lambda x: np.cos(2 * np.pi * (12.5 * x - 0.25) / 8)), (7, lambda x: x ** 3 - 0.5 * x),
                      (12, np.exp), (1, lambda x: 2 * x + 1)):
        co = np.polynomial.chebyshev.chebinterpolate(f, degree)
        ch = ChebyshevEvaluator(ev, co)
        out = ch(ev.encrypt(ev.encode(w, 2.0 ** 40, 2)), 2.0 ** 40)
        assert out.level == 2 + ch.depth and abs(out.scale / 2.0 ** 40 - 1) < 1e-12     # exact target scale
        assert np.abs(ev.decrypt_decode(out) - np.polynomial.chebyshev.chebval(w, co)).max() < 1e-9
    z = rng.normal(size=16) + 1j * rng.normal(size=16)
    M = rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16))
    lt = LinearTransform(ev, M, 3, 2.0 ** 40)
    assert np.abs(ev.decrypt_decode(lt.apply(ev.encrypt(ev.encode(z, 2.0 ** 40, 3)))) - M @ z).max() < 1e-9


@pytest.mark.parametrize("s,packed", [(4, True), (16, True), (64, True), (16, False)])
def test_bootstrap_on_the_simulator(orc, pha, s, packed):
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from oracle.sim_evaluator import SimEvaluator
    q = _chain(orc)
    ev = SimEvaluator(LOG_N, q, s, hamming_weight=32, seed=s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, packed=packed)
    assert bs.levels_needed() == 13 and bs.packed == packed
    rng = np.random.default_rng(1)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    out = bs.bootstrap(ev.encrypt(ev.encode(msg, 2.0 ** 40, ev.bottom)))
    assert np.abs(ev.last_I).max() >= 3                     # the modulus raise really added multiples of q0
    assert out.level == 13 and abs(out.scale / 2.0 ** 40 - 1) < 1e-3
    # the sine's cubic term: (2 pi |m| / q0)^2 / 6 relative, |m|/q0 = 2^-10
    assert np.abs(ev.decrypt_decode(out) - msg).max() < 3e-5


def _fft_chain(orc, n, groups, extra=(), d=40, e=None):
    # q0, two levels left, SlotToCoeff groups, 9 EvalMod levels, CoeffToSlot groups (wide primes: the
    # stage diagonals multiply q0 * I and their rounding comes back times q0 / delta)
    return orc.coeff_modulus_create(n, [d + 10] + [d] * 2 + [d] * groups + [e or d] * 9 + [min(60, d + 18)] * groups + list(extra))


def test_special_fft_factorisation():
    """slot_matrix(S) = F_S o .. o F_2 o bitrev, for every grouping of the stages"""
    from phantom_fhe_amd.bootstrap import slot_matrix, special_fft_groups
    for S, groups in ((2, 1), (16, 1), (16, 2), (64, 3), (128, 4), (256, 3)):
        bits = S.bit_length() - 1
        perm = np.array([int(format(i, "0%db" % bits)[::-1], 2) for i in range(S)])
        A, j = slot_matrix(S), np.arange(S)

        def dense(maps):
            m = np.eye(S, dtype=np.complex128)
            for g in maps:
                d = np.zeros((S, S), dtype=np.complex128)
                for off, v in g.items():
                    d[j, (j + off) % S] += v
                m = d @ m
            return m
        fwd, inv = special_fft_groups(S, groups, False), special_fft_groups(S, groups, True)
        assert len(fwd) == len(inv) == groups
        assert np.abs(dense(fwd)[:, perm] - A).max() < 1e-12            # forward stages take bit-reversed input
        assert np.abs(dense(inv)[perm] - np.linalg.inv(A)).max() < 1e-12    # inverse leaves bit-reversed output
        # radix-2^k groups: at most 2^(k+1) - 1 diagonals
        assert max(len(g) for g in fwd + inv) <= 2 ** (-(-bits // groups) + 1) - 1


@pytest.mark.parametrize("log_n,s,groups", [(12, 256, 3), (12, 32, 2), (8, 128, 3), (9, 256, 4)])
def test_fft_bootstrap_on_the_simulator(orc, pha, log_n, s, groups):
    """FFT-factored CoeffToSlot / SlotToCoeff, sparse and FULL slots (s = N/2: no SubSum)"""
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from oracle.sim_evaluator import SimEvaluator
    n = 1 << log_n
    q = [int(v) for v in _fft_chain(orc, n, groups)]
    ev = SimEvaluator(log_n, q, s, hamming_weight=32, seed=s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, transform="fft", c2s_groups=groups, s2c_groups=groups)
    assert bs.levels_needed() == 2 * groups + 10 and not bs.packed
    rng = np.random.default_rng(1)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    out = bs.bootstrap(ev.encrypt(ev.encode(msg, 2.0 ** 40, ev.bottom)))
    assert np.abs(ev.last_I).max() >= 3
    assert out.level == bs.levels_needed() and abs(out.scale / 2.0 ** 40 - 1) < 1e-3
    assert np.abs(ev.decrypt_decode(out) - msg).max() < 3e-5


@pytest.mark.parametrize("log_n,s,groups", [(12, 256, 3), (12, 32, 2), (8, 128, 3), (9, 256, 4)])
def test_slim_bootstrap_on_the_simulator(orc, pha, log_n, s, groups):
    """SlotToCoeff first (on the last levels), then modulus raise, CoeffToSlot, EvalMod: the output scale is the
    input scale by the choice of the EvalMod scale, whatever the input scale is"""
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from oracle.sim_evaluator import SimEvaluator
    n = 1 << log_n
    q = [int(v) for v in _fft_chain(orc, n, groups)]
    ev = SimEvaluator(log_n, q, s, hamming_weight=32, seed=s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, transform="fft", c2s_groups=groups, s2c_groups=groups, slim=True)
    assert bs.levels_needed() == groups + 10 and bs.pre_levels == min(groups, s.bit_length() - 1)
    rng = np.random.default_rng(1)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    for scale in (2.0 ** 40, 1.37 * 2.0 ** 39):
        out = bs.bootstrap(ev.encrypt(ev.encode(msg, scale, ev.bottom - bs.pre_levels)))
        assert np.abs(ev.last_I).max() >= 3
        assert out.level == bs.levels_needed() and abs(out.scale / scale - 1) < 1e-9
        assert np.abs(ev.decrypt_decode(out) - msg).max() < 3e-5
    with pytest.raises(AssertionError):
        bs.bootstrap(ev.encrypt(ev.encode(msg, 2.0 ** 40, ev.bottom)))        # no level left for SlotToCoeff


@pytest.mark.gpu
@pytest.mark.parametrize("s,groups,d,e", [(2048, 3, 50, 50), (2048, 4, 40, 58), (128, 2, 40, 40)])
def test_slim_bootstrap_on_the_gpu(gpu, pha, orc, s, groups, d, e):
    """the slim order on real ciphertexts: the EvalMod error reaches the slots without the sqrt(s) of a
    SlotToCoeff behind it (compare the tolerances of test_fft_bootstrap_on_the_gpu)"""
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from phantom_fhe_amd.ckks import GpuEvaluator
    mods = _fft_chain(orc, N, groups, extra=[60], d=d, e=e)
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    sk = ctx.gen_secretkey(23, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, delta=2.0 ** e, transform="fft", c2s_groups=groups, s2c_groups=groups, slim=True)
    rng = np.random.default_rng(3)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom - bs.pre_levels))
    out = bs.bootstrap(ct)
    assert out.level == groups + 10 and abs(out.scale / ct.scale - 1) < 1e-9
    err = np.abs(ev.decrypt_decode(out) - msg).max()
    assert err < (1e-3 if e == 40 else 2e-5), err
    sq = ev.rescale(ev.mul(out, out))
    assert np.abs(ev.decrypt_decode(sq) - msg * msg).max() < (2e-3 if e == 40 else 1e-4)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("s,groups,d,e", [(2048, 3, 50, 50), (2048, 4, 40, 58), (128, 2, 40, 40)])
def test_fft_bootstrap_on_the_gpu(gpu, pha, orc, s, groups, d, e):
    """full-slot (s = N/2) and sparse bootstrapping through the FFT-factored transforms on real ciphertexts.
    Full slots work at delta = 2^50: the EvalMod noise floor (about 2^-20 of the sine at delta = 2^40, the
    rescale / relinearisation noise through 5 squarings of the Chebyshev basis and 3 double angles) comes back
    times q0 / (2 pi delta) = 163 per coefficient and times sqrt(s) through SlotToCoeff — measured 1.6e-3 at
    2^40, which sparse packing (SubSum averages it) does not see.  The (40, 58) case keeps 40-bit messages and
    runs EvalMod alone at 2^58 (msg_scale != delta)."""
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from phantom_fhe_amd.ckks import GpuEvaluator
    mods = _fft_chain(orc, N, groups, extra=[60], d=d, e=e)
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    sk = ctx.gen_secretkey(23, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, delta=2.0 ** e, msg_scale=2.0 ** d, transform="fft", c2s_groups=groups,
                      s2c_groups=groups)
    rng = np.random.default_rng(3)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** d, ev.bottom))
    out = bs.bootstrap(ct)
    assert out.level == 2 * groups + 10 and ev.limbs(out.level) == 3
    got = ev.decrypt_decode(out)
    tol = 1e-3 if e == 40 else 2e-5
    assert np.abs(got - msg).max() < tol
    sq = ev.rescale(ev.mul(out, out))
    assert np.abs(ev.decrypt_decode(sq) - msg * msg).max() < 2 * tol
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("s,packed", [(8, True), (32, True), (16, False)])
def test_bootstrap_on_the_gpu(gpu, pha, orc, s, packed):
    from phantom_fhe_amd.bootstrap import Bootstrapper
    from phantom_fhe_amd.ckks import GpuEvaluator
    mods = orc.coeff_modulus_create(N, Q_BITS + [60])
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    sk = ctx.gen_secretkey(21, hamming_weight=32)
    ev = GpuEvaluator(pha, ctx, sk, s)
    bs = Bootstrapper(ev, K=12, r=3, degree=31, packed=packed)
    rng = np.random.default_rng(2)
    msg = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    ct = ev.encrypt(ev.encode(msg, 2.0 ** 40, ev.bottom))
    assert np.abs(ev.decrypt_decode(ct) - msg).max() < 1e-7
    out = bs.bootstrap(ct)
    assert out.level == 13 and ev.limbs(out.level) == 3     # two multiplicative levels regained
    got = ev.decrypt_decode(out)
    assert np.abs(got - msg).max() < 1e-3
    # and the refreshed ciphertext computes: square it, rescale, compare slot-wise
    sq = ev.rescale(ev.mul(out, out))
    assert np.abs(ev.decrypt_decode(sq) - msg * msg).max() < 2e-3
    ctx.close()
