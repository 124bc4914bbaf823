"""CKKS encoder / decoder (src/ckks.cu:130-283).  The oracle is the definition of
the canonical embedding (oracle/ckks_ref.py: a dense s x s matrix in complex128); FP64 parity is by
tolerance: encoded integer coefficients may differ from the oracle's by the rounding of values that
sit within FP noise of a half-integer (|diff| <= 1 at scale 2^40, i.e. 2^-40 relative), decoded
slots agree to 1e-9.  Then the scheme-level pins of the reference's tests (tests/multi.cu,
tests/rotation.cu): decode(decrypt(op(encrypt(encode z)))) ~= op(z) slot-wise, all on the GPU."""
import numpy as np
import pytest

N, LOG_N = 4096, 12


def test_reference_definition_is_consistent():
    from oracle.ckks_ref import decode_ref, encode_ref
    rng = np.random.default_rng(1)
    for s in (2, 8, 64):
        z = rng.normal(size=s) + 1j * rng.normal(size=s)
        c = encode_ref(z, 2.0 ** 40, N, s)
        assert sum(1 for v in c if v) <= 2 * s
        assert np.abs(decode_ref(c, 2.0 ** 40, N, s) - z).max() < 1e-9
    # a real constant vector encodes to the constant polynomial
    c = encode_ref(np.full(8, 3.0), 2.0 ** 30, N, 8)
    assert c[0] == 3 << 30 and not any(c[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("s", [2, 16, 512, 2048])
def test_gpu_encode_decode_matches_definition(gpu, pha, orc, s):
    from oracle.ckks_ref import CkksHarness, decode_ref, encode_ref
    mods = orc.coeff_modulus_create(N, [60, 40, 40, 60])
    o = orc.Oracle(LOG_N, mods, 1)
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    h = CkksHarness(o, seed=1)
    rng = np.random.default_rng(s)
    nvals = s if s < 16 else s - 3                       # ragged: zero padding up to sparse_slots
    z = rng.normal(size=nvals) + 1j * rng.normal(size=nvals)
    scale = 2.0 ** 40
    plain = ctx.encode(z, scale, 1, s)
    coeff = np.stack([o.ntt_inv(l, i) for i, l in enumerate(pha.to_host(plain))])
    got = h.crt_compose(coeff, range(3))
    want = encode_ref(z, scale, N, s)
    assert max(abs(a - b) for a, b in zip(got, want)) <= 1
    dec = ctx.decode(plain, scale, s, 1)
    zp = np.concatenate([z, np.zeros(s - nvals)])
    assert np.abs(dec - zp).max() < 1e-9
    assert np.abs(dec - decode_ref(got, scale, N, s)).max() < 1e-9
    # large scale: coefficients beyond 64 bits go through the multi-word paths
    big = 2.0 ** 90
    pb = ctx.encode(z, big, 1, s)
    gb = h.crt_compose(np.stack([o.ntt_inv(l, i) for i, l in enumerate(pha.to_host(pb))]), range(3))
    wb = encode_ref(z, big, N, s)
    assert max(abs(a - b) for a, b in zip(gb, wb)) <= 2 ** 52     # 2^-38 relative: FP64 of a 2^90-scaled value
    assert np.abs(ctx.decode(pb, big, s, 1) - zp).max() < 1e-9
    ctx.close()


@pytest.mark.gpu
def test_gpu_encoder_errors(gpu, pha, orc):
    mods = orc.coeff_modulus_create(N, [40, 40, 50])
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    z = np.ones(4, dtype=np.complex128)
    for args in ((np.ones(N, dtype=np.complex128), 2.0 ** 30, 1, 0),   # values_size is too large
                 (z, 2.0 ** 90, 1, 0),                                 # scale out of bounds
                 (z, 0.0, 1, 0),
                 (z[:1], 2.0 ** 30, 1, 0),                             # single value encoding is not available
                 (z * 2.0 ** 45, 2.0 ** 40, 1, 0)):                    # encoded values are too large
        with pytest.raises(pha.PhantomError) as e:
            ctx.encode(*args)
        assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("LOG_N,special", [(12, 1), (17, 2)])
def test_gpu_ckks_pipeline_slotwise(gpu, pha, orc, LOG_N, special):
    """encode -> encrypt -> HMult+relin -> rescale -> rotate -> conjugate -> decrypt -> decode; also at the largest
    supported degree N = 2^17 with full slots (65536)."""
    N = 1 << LOG_N
    mods = orc.coeff_modulus_create(N, [60, 40, 40, 40] + [60] * special)
    ctx = pha.PhantomContext(LOG_N, mods, special)
    sk = ctx.gen_secretkey(5)
    rlk = ctx.gen_relinkey(sk, 6)
    s = N // 2
    rng = np.random.default_rng(3)
    z1 = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    z2 = rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
    scale = 2.0 ** 40
    ct1 = ctx.encrypt_symmetric(sk, ctx.encode(z1, scale, 1, s), 1, 11, 12)
    ct2 = ctx.encrypt_symmetric(sk, ctx.encode(z2, scale, 1, s), 1, 13, 14)
    assert np.abs(ctx.decode(ctx.ckks_decrypt(sk, ct1, 1), scale, s, 1) - z1).max() < 1e-7
    import torch
    prod = torch.cat([ct1, torch.zeros_like(ct1[:1])])
    ctx.multiply_and_relin_inplace(1, prod, ct2, rlk)
    res = torch.empty((2, 3, N), dtype=torch.int64, device="cuda")
    ctx.rescale_to_next(1, prod[:2].contiguous(), 2, res)
    new_scale = scale * scale / float(mods[3])
    got = ctx.decode(ctx.ckks_decrypt(sk, res, 2), new_scale, s, 2)
    # the reference's tests ask for << 1e-3.  The rescale is the reference's FLOOR division (src/rns.cu:1125-1181):
    # its +1/2 mean per coefficient is a constant polynomial bias that lands on the slots whose root is nearest 1
    # (slot 0: 8e-6 at N = 2^16, 2.2e-5 at 2^17; the median slot error is 1e-8)
    tol = 1e-5 if LOG_N == 12 else 1e-4
    assert np.abs(got - z1 * z2).max() < tol and np.median(np.abs(got - z1 * z2)) < 1e-6
    # rotation by 3 slots at level 2, then complex conjugation
    step = 3
    elt = pha.galois_elt_from_step(LOG_N, step)
    ctx.apply_galois_inplace(2, res, elt, ctx.gen_galois_key(sk, elt, 7))
    got = ctx.decode(ctx.ckks_decrypt(sk, res, 2), new_scale, s, 2)
    assert np.abs(got - np.roll(z1 * z2, -step)).max() < tol
    conj = 2 * N - 1
    ctx.apply_galois_inplace(2, res, conj, ctx.gen_galois_key(sk, conj, 8))
    got = ctx.decode(ctx.ckks_decrypt(sk, res, 2), new_scale, s, 2)
    assert np.abs(got - np.conj(np.roll(z1 * z2, -step))).max() < tol
    ctx.close()
