"""Functional pins with real keys, the way the reference's own tests pin this path
(tests/relin.cu, tests/rotation.cu, tests/rescale_to_next.cu, tests/multi.cu: decrypt(op(encrypt(x)))
~= op(x)): a secret key, relinearisation and Galois keys in the reference's evk
layout, symmetric encryption and decryption are restated on the CPU (oracle/ckks_ref.py), and
HMult+relin, rotation and rescale must decrypt to the exact polynomial product / automorphism /
quotient up to the scheme's noise.  The CPU half runs the oracle (so the oracle itself is pinned
to the scheme's semantics, not only to its own restatement); the GPU half runs the C ABI on the
same ciphertexts and keys and must in addition agree with the oracle bit for bit."""
import numpy as np
import pytest

N, LOG_N = 4096, 12
DELTA = 1 << 40
# (Q bit sizes, P bit sizes): alpha = 2 / beta = 2, and the alpha = 1 path
CHAINS = [([50, 40, 40, 40], [50, 50]), ([50, 40, 40], [55])]


def _setup(orc_mod, chain, seed):
    from oracle.ckks_ref import CkksHarness
    q_bits, p_bits = chain
    mods = orc_mod.coeff_modulus_create(N, q_bits + p_bits)
    o = orc_mod.Oracle(LOG_N, mods, len(p_bits))
    return o, CkksHarness(o, seed=seed)


def _messages(seed):
    rng = np.random.default_rng(seed)
    m1 = rng.integers(-8, 9, size=N)
    m2 = rng.integers(-8, 9, size=N)
    return [int(v) * DELTA for v in m1], [int(v) * DELTA for v in m2]


def _max_err(got, want):
    return max(abs(g - w) for g, w in zip(got, want))


def _q_prod(o, size_ql):
    out = 1
    for q in o.moduli[:size_ql]:
        out *= int(q)
    return out


def _check_mult(h, o, ct, size_ql, p1, p2):
    from oracle.ckks_ref import negacyclic_mul
    want = negacyclic_mul(p1, p2)
    big_q = _q_prod(o, size_ql)
    assert max(abs(w) for w in want) < big_q // 4, "test parameters: product must fit the modulus"
    err = _max_err(h.decrypt(ct, size_ql), want)
    assert err < DELTA * DELTA * 1e-4, f"HMult+relin noise {err / DELTA ** 2:.3e} Delta^2"
    return want


@pytest.mark.parametrize("chain_id,drop", [(0, 0), (0, 1), (1, 0)])
def test_oracle_relin_rotate_rescale_decrypt(orc, chain_id, drop):
    o, h = _setup(orc, CHAINS[chain_id], seed=5 + chain_id)
    size_ql = o.size_Q - drop
    p1, p2 = _messages(3)
    ct1, ct2 = h.encrypt(p1, size_ql), h.encrypt(p2, size_ql)
    assert _max_err(h.decrypt(ct1), p1) < 64  # fresh noise: a few sigma

    # HMult + relin
    rlk = h.relin_key()
    t = o.tensor_prod_2x2(ct1, ct2)
    # the size-3 ciphertext already decrypts to the product under (1, s, s^2)
    _check_mult(h, o, t, size_ql, p1, p2)
    ct = t[:2].copy()
    o.keyswitch_inplace(size_ql, ct, t[2], rlk)
    want = _check_mult(h, o, ct, size_ql, p1, p2)

    # rescale: floor division by the last prime (src/rns.cu:1157-1181)
    if size_ql > 1:
        q_last = int(o.moduli[size_ql - 1])
        before = h.decrypt(ct, size_ql)
        res = o.divide_and_round_q_last_ntt(size_ql, ct, 2)
        got = h.decrypt(res, size_ql - 1)
        # each coefficient of c0, c1 moves by less than one unit of q_last, so the decryption
        # c0 + c1*s moves by at most (1 + |s|_1) <= N + 1 units
        err = max(abs(g * q_last - w) for g, w in zip(got, before))
        assert err <= q_last * (N + 1), f"rescale error {err / q_last:.1f} q_last"
        assert _max_err([g * q_last for g in got], want) < DELTA * DELTA * 1e-4

    # rotation by one slot: Galois element 5 (src/galois.cu, include/galois.cuh)
    elt = orc.galois_elt_from_step(LOG_N, 1)
    glk = h.galois_key(elt)
    c0 = o.apply_galois_ntt(ct1[0], elt)
    c1 = o.apply_galois_ntt(ct1[1], elt)
    rot = np.stack([c0, np.zeros_like(c0)])
    o.keyswitch_inplace(size_ql, rot, c1, glk)
    err = _max_err(h.decrypt(rot, size_ql), h.galois_coeffs(p1, elt))
    assert err < DELTA * 1e-4, f"rotation noise {err / DELTA:.3e} Delta"


@pytest.mark.gpu
@pytest.mark.parametrize("chain_id,drop", [(0, 0), (0, 1), (1, 0)])
def test_gpu_relin_rotate_rescale_decrypt(gpu, pha, orc, chain_id, drop):
    o, h = _setup(orc, CHAINS[chain_id], seed=9 + chain_id)
    ctx = pha.PhantomContext(LOG_N, o.moduli, o.size_P)
    size_ql = o.size_Q - drop
    chain_index = 1 + drop
    p1, p2 = _messages(4)
    ct1, ct2 = h.encrypt(p1, size_ql), h.encrypt(p2, size_ql)
    rlk_h = h.relin_key()
    rlk = pha.EvalKey([pha.to_device(k) for k in rlk_h])

    d_ct = pha.to_device(np.concatenate([ct1, np.zeros_like(ct1[:1])]))
    ctx.multiply_and_relin_inplace(chain_index, d_ct, pha.to_device(ct2), rlk)
    got = pha.to_host(d_ct)[:2]
    want = _check_mult(h, o, got, size_ql, p1, p2)
    t = o.tensor_prod_2x2(ct1, ct2)
    exp = t[:2].copy()
    o.keyswitch_inplace(size_ql, exp, t[2], rlk_h)
    assert (got == exp).all()

    if size_ql > 1:
        q_last = int(o.moduli[size_ql - 1])
        d_res = pha.to_device(np.zeros((2, size_ql - 1, N), dtype=np.uint64))
        ctx.rescale_to_next(chain_index, pha.to_device(got), 2, d_res)
        res = pha.to_host(d_res)
        dec = h.decrypt(res, size_ql - 1)
        assert _max_err([g * q_last for g in dec], want) < DELTA * DELTA * 1e-4
        assert (res == o.divide_and_round_q_last_ntt(size_ql, got, 2)).all()

    elt = orc.galois_elt_from_step(LOG_N, 1)
    glk_h = h.galois_key(elt)
    glk = pha.EvalKey([pha.to_device(k) for k in glk_h])
    d_rot = pha.to_device(ct1)
    ctx.apply_galois_inplace(chain_index, d_rot, elt, glk)
    rot = pha.to_host(d_rot)
    assert _max_err(h.decrypt(rot, size_ql), h.galois_coeffs(p1, elt)) < DELTA * 1e-4
    ctx.close()
