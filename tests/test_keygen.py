"""Key generation, symmetric encryption and decryption: the oracle's restatement
is pinned functionally on the CPU (distributions, decrypt(encrypt(m)) = m + small noise, HMult+relin
and rotation with the generated keys decrypt correctly), the C ABI must reproduce it bit for bit on
the GPU from the same seeds, and an all-GPU pipeline (keygen -> encrypt -> HMult+relin -> rescale ->
rotate -> decrypt) must return the expected polynomials."""
import numpy as np
import pytest

N, LOG_N = 4096, 12
DELTA = 1 << 40
CHAINS = [([50, 40, 40, 40], [50, 50]), ([50, 40, 40], [55])]


def _setup(orc, chain):
    q_bits, p_bits = chain
    mods = orc.coeff_modulus_create(N, q_bits + p_bits)
    return mods, orc.Oracle(LOG_N, mods, len(p_bits))


def _centre(v, q):
    v = v.astype(np.int64)
    return np.where(v > q // 2, v - q, v)


def _harness(o, sk):
    """CkksHarness (CRT compose, exact products) around an externally generated secret key."""
    from oracle.ckks_ref import CkksHarness
    h = CkksHarness(o, seed=1)
    h.sk_ntt = np.ascontiguousarray(sk)
    h.s = _centre(o.ntt_inv(sk[0], 0), int(o.moduli[0]))
    return h


def _messages(seed):
    rng = np.random.default_rng(seed)
    return ([int(v) * DELTA for v in rng.integers(-8, 9, size=N)], [int(v) * DELTA for v in rng.integers(-8, 9, size=N)])


def test_sampler_distributions(orc):
    mods, o = _setup(orc, CHAINS[0])
    q0 = int(mods[0])
    t = o.sample("ternary", 3, 5)
    tc = _centre(t[0], q0)
    assert set(np.unique(tc)) == {-1, 0, 1} and abs(tc.mean()) < 0.06 and abs((tc == 0).mean() - 86 / 256) < 0.03
    assert (_centre(t[1], int(mods[1])) == tc).all()            # the same coefficient in every limb (src/prng.cu:153-157)
    e = _centre(o.sample("error", 2, 6)[0], q0)
    assert abs(e.mean()) < 0.2 and abs(e.std() - np.sqrt(21 / 2)) < 0.2 and np.abs(e).max() <= 21
    u = o.sample("uniform", o.size_QP, 7)
    assert (u < mods[:, None]).all()
    assert all(abs(u[i].mean() / int(mods[i]) - 0.5) < 0.02 for i in range(o.size_QP))
    assert orc.rnd_at(1, 0) != orc.rnd_at(1, 1) and orc.rnd_at(1, 5) == orc.rnd_at(1, 5)
    assert (o.sample("uniform", 2, 7) != o.sample("uniform", 2, 8)).mean() > 0.99


def test_chacha20_block_known_answer(orc):
    """RFC 8439 section 2.3.2: key 00..1f, block counter 1, nonce 00 00 00 09 00 00 00 4a 00 00 00 00.  The sampler's
    counter / nonce words are (ctr_lo, ctr_hi, seed_lo, seed_hi), so that vector is ctr = 1 | 0x09000000 << 32,
    seed = 0x4a000000."""
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    out = orc.chacha20_block(key, 0x4A000000, 1 | (0x09000000 << 32))
    assert out == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
                   0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    # rnd(seed, ctr) = the first 64 bits of the block under the oracle's (all-zero) key
    w = orc.chacha20_block([0] * 8, 7, 9)
    assert orc.rnd_at(7, 9) == w[0] | (w[1] << 32)


def test_library_nonces_and_sampler_key_entry_points(pha):
    """no GPU needed: the entry points exist and OS-entropy nonces do not repeat"""
    pha.lib()
    assert len({pha.random_nonce() for _ in range(64)}) == 64
    pha.set_sampler_key(bytes(range(32)))
    pha.set_sampler_key(None)


@pytest.mark.gpu
def test_default_sampler_key_is_os_entropy(gpu, pha, orc):
    """The library's default state: a 256-bit key from OS entropy.  The same nonce then gives unrelated streams from
    one key to the next (nothing about a draw can be recomputed from its nonce or from another draw), while under
    the fixed test key the nonce determines the draw."""
    mods, o = _setup(orc, CHAINS[0])
    ctx = pha.PhantomContext(LOG_N, mods, 2)
    try:
        pha.set_sampler_key(None)
        a = pha.to_host(ctx.gen_secretkey(5))
        ct_a = pha.to_host(ctx.encrypt_zero_symmetric(ctx.gen_secretkey(5), 1, 77, 78))
        pha.set_sampler_key(None)
        b = pha.to_host(ctx.gen_secretkey(5))
        ct_b = pha.to_host(ctx.encrypt_zero_symmetric(ctx.gen_secretkey(5), 1, 77, 78))
        assert (a != b).mean() > 0.3                      # two ternary keys, NTT form: unrelated
        assert (ct_a[1] != ct_b[1]).mean() > 0.99         # the published uniform part under the same nonce: unrelated
        pha.set_sampler_key(orc.SAMPLER_KEY)
        c1, c2 = pha.to_host(ctx.gen_secretkey(5)), pha.to_host(ctx.gen_secretkey(5))
        assert (c1 == c2).all() and (c1 == o.gen_secretkey(5)).all()
    finally:
        pha.set_sampler_key(orc.SAMPLER_KEY)
        ctx.close()


@pytest.mark.parametrize("chain_id", [0, 1])
def test_oracle_keys_decrypt(orc, chain_id):
    from oracle.ckks_ref import negacyclic_mul
    mods, o = _setup(orc, CHAINS[chain_id])
    sk = o.gen_secretkey(11)
    h = _harness(o, sk)
    assert set(np.unique(h.s)) <= {-1, 0, 1}
    p1, p2 = _messages(3)
    size_ql = o.size_Q
    ct1 = o.encrypt_symmetric(sk, h.to_ntt(p1, range(size_ql)), 1, 21, 22)
    ct2 = o.encrypt_symmetric(sk, h.to_ntt(p2, range(size_ql)), 1, 23, 24)
    dec = h.crt_compose(np.stack([o.ntt_inv(l, i) for i, l in enumerate(o.ckks_decrypt(sk, ct1, 1))]), range(size_ql))
    assert max(abs(a - b) for a, b in zip(dec, p1)) <= 21        # exactly the centred-binomial error
    t = o.tensor_prod_2x2(ct1, ct2)
    want = negacyclic_mul(p1, p2)
    assert max(abs(a - b) for a, b in zip(h.decrypt(t), want)) < DELTA * DELTA * 1e-4   # size-3 decrypt
    rlk = o.gen_relinkey(sk, 31)
    ct = t[:2].copy()
    o.keyswitch_inplace(size_ql, ct, t[2], rlk)
    assert max(abs(a - b) for a, b in zip(h.decrypt(ct), want)) < DELTA * DELTA * 1e-4
    elt = orc.galois_elt_from_step(LOG_N, 2)
    glk = o.gen_galois_key(sk, elt, 41)
    rot = np.stack([o.apply_galois_ntt(ct1[0], elt), np.zeros_like(ct1[0])])
    o.keyswitch_inplace(size_ql, rot, o.apply_galois_ntt(ct1[1], elt), glk)
    assert max(abs(a - b) for a, b in zip(h.decrypt(rot), h.galois_coeffs(p1, elt))) < DELTA * 1e-4


@pytest.mark.parametrize("chain_id", [0, 1])
def test_oracle_public_key_encryption_decrypts(orc, chain_id):
    """PhantomPublicKey::encrypt_asymmetric (src/secretkey.cu:15-164): pk * u + e at the key level, divided by P.
    The decryption error is the rounding of that division times (1 + s): far below one centred-binomial draw times N."""
    mods, o = _setup(orc, CHAINS[chain_id])
    sk = o.gen_secretkey(11)
    h = _harness(o, sk)
    pk = o.gen_publickey(sk, 61)
    assert (pk == o.encrypt_zero_symmetric(sk, 0, orc.rnd_at(61, 0), orc.rnd_at(61, 1))).all()
    p1, _ = _messages(5)
    for chain_index in (1, 2):
        limbs = o.level_limbs(chain_index)
        ct = o.encrypt_asymmetric(pk, h.to_ntt(p1, range(limbs)), chain_index, 71)
        assert ct.shape == (2, limbs, N)
        dec = h.crt_compose(np.stack([o.ntt_inv(l, i) for i, l in enumerate(o.ckks_decrypt(sk, ct, chain_index))]), range(limbs))
        assert max(abs(a - b) for a, b in zip(dec, p1)) <= N          # measured: a few hundred
    z = o.encrypt_zero_asymmetric(pk, 1, 72)
    assert (z != o.encrypt_zero_asymmetric(pk, 1, 73)).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("chain_id", [0, 1])
def test_gpu_public_key_encryption_matches_oracle(gpu, pha, orc, chain_id):
    mods, o = _setup(orc, CHAINS[chain_id])
    ctx = pha.PhantomContext(LOG_N, mods, o.size_P)
    sk_h, sk = o.gen_secretkey(11), ctx.gen_secretkey(11)
    pk_h, pk = o.gen_publickey(sk_h, 61), ctx.gen_publickey(sk, 61)
    assert (pha.to_host(pk) == pk_h).all()
    h = _harness(o, sk_h)
    p1, _ = _messages(5)
    for chain_index in (1, 3):
        limbs = o.level_limbs(chain_index)
        assert (pha.to_host(ctx.encrypt_zero_asymmetric(pk, chain_index, 72)) == o.encrypt_zero_asymmetric(pk_h, chain_index, 72)).all()
        pl = h.to_ntt(p1, range(limbs))
        ct = ctx.encrypt_asymmetric(pk, pha.to_device(pl), chain_index, 71)
        assert (pha.to_host(ct) == o.encrypt_asymmetric(pk_h, pl, chain_index, 71)).all()
        plain = pha.to_host(ctx.ckks_decrypt(sk, ct, chain_index))
        dec = h.crt_compose(np.stack([o.ntt_inv(plain[i], i) for i in range(limbs)]), range(limbs))
        assert max(abs(a - b) for a, b in zip(dec, p1)) <= N
    with pytest.raises(pha.PhantomError):
        ctx.encrypt_zero_asymmetric(pk, 0, 1)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("chain_id", [0, 1])
def test_gpu_keygen_matches_oracle_and_decrypts(gpu, pha, orc, chain_id):
    from oracle.ckks_ref import negacyclic_mul
    mods, o = _setup(orc, CHAINS[chain_id])
    ctx = pha.PhantomContext(LOG_N, mods, o.size_P)
    sk_h = o.gen_secretkey(11)
    sk = ctx.gen_secretkey(11)
    assert (pha.to_host(sk) == sk_h).all()
    for chain_index in (0, 1, 2):
        assert (pha.to_host(ctx.encrypt_zero_symmetric(sk, chain_index, 5, 6)) ==
                o.encrypt_zero_symmetric(sk_h, chain_index, 5, 6)).all()
    rlk = ctx.gen_relinkey(sk, 31)
    for d, ref in zip(rlk.digits, o.gen_relinkey(sk_h, 31)):
        assert (pha.to_host(d) == ref).all()
    elt = orc.galois_elt_from_step(LOG_N, 2)
    glk = ctx.gen_galois_key(sk, elt, 41)
    for d, ref in zip(glk.digits, o.gen_galois_key(sk_h, elt, 41)):
        assert (pha.to_host(d) == ref).all()
    new_key = o.uniform_poly(list(range(o.size_Q)), 77)
    ksk = ctx.generate_one_kswitch_key(sk, pha.to_device(new_key), 51)
    for d, ref in zip(ksk.digits, o.generate_one_kswitch_key(sk_h, new_key, 51)):
        assert (pha.to_host(d) == ref).all()

    # all-GPU pipeline on real data
    h = _harness(o, sk_h)
    p1, p2 = _messages(4)
    size_ql = o.size_Q
    pl1, pl2 = h.to_ntt(p1, range(size_ql)), h.to_ntt(p2, range(size_ql))
    ct1 = ctx.encrypt_symmetric(sk, pha.to_device(pl1), 1, 21, 22)
    ct2 = ctx.encrypt_symmetric(sk, pha.to_device(pl2), 1, 23, 24)
    assert (pha.to_host(ct1) == o.encrypt_symmetric(sk_h, pl1, 1, 21, 22)).all()

    def decode(plain_dev, limbs):
        pl = pha.to_host(plain_dev)
        return h.crt_compose(np.stack([o.ntt_inv(pl[i], i) for i in range(limbs)]), range(limbs))

    assert max(abs(a - b) for a, b in zip(decode(ctx.ckks_decrypt(sk, ct1, 1), size_ql), p1)) <= 21
    want = negacyclic_mul(p1, p2)
    import torch
    prod = torch.cat([ct1, torch.zeros_like(ct1[:1])])
    ctx.tensor_prod_2x2(ct1, ct2, size_ql, prod)
    got3 = ctx.ckks_decrypt(sk, prod, 1)                                     # size-3 decryption
    assert (pha.to_host(got3) == o.ckks_decrypt(sk_h, pha.to_host(prod), 1)).all()
    assert max(abs(a - b) for a, b in zip(decode(got3, size_ql), want)) < DELTA * DELTA * 1e-4
    prod = torch.cat([ct1, torch.zeros_like(ct1[:1])])
    ctx.multiply_and_relin_inplace(1, prod, ct2, rlk)
    assert max(abs(a - b) for a, b in zip(decode(ctx.ckks_decrypt(sk, prod[:2].contiguous(), 1), size_ql), want)) < DELTA * DELTA * 1e-4
    res = torch.empty((2, size_ql - 1, N), dtype=torch.int64, device="cuda")
    ctx.rescale_to_next(1, prod[:2].contiguous(), 2, res)
    q_last = int(mods[size_ql - 1])
    dec = decode(ctx.ckks_decrypt(sk, res, 2), size_ql - 1)
    assert max(abs(a * q_last - b) for a, b in zip(dec, want)) < DELTA * DELTA * 1e-4
    rot = ct1.clone()
    ctx.apply_galois_inplace(1, rot, elt, glk)
    assert max(abs(a - b) for a, b in zip(decode(ctx.ckks_decrypt(sk, rot, 1), size_ql), h.galois_coeffs(p1, elt))) < DELTA * 1e-4
    ctx.close()
