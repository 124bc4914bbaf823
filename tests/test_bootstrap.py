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
    from phantom_fhe_amd.ckks import SimEvaluator
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
    from phantom_fhe_amd.ckks import SimEvaluator
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
