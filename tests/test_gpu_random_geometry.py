"""Seeded differential sweep over chain GEOMETRIES: the parametrised parity tests pin a handful of (size_Q, alpha, level)
shapes; here the shape itself is drawn -- N = 2^11 .. 2^13, alpha = 1 .. 5, dnum = 1 .. 4, prime sizes 36 .. 60 bits in
any mix (so FP64 and integer limbs interleave, and base conversions with and without matrix-core tables), any level with
at least two limbs left (full and ragged last digits, down to a single digit) -- and every evaluator entry point of the
hot path (key switch, HMult + relinearise, rotation, rescale, the merged multiply-relinearise-rescale and hoisted
rotations) is compared with the oracle bit for bit on uniform inputs and a uniform key (arithmetic parity does not need a
valid key; reference: src/eval_key_switch.cu:91-178, src/evaluate.cu:229-280,553-616,747-794)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _geometry(seed):
    rng = np.random.default_rng(9000 + seed)
    log_n = int(rng.choice([11, 12, 12, 13]))
    alpha = int(rng.integers(1, 6))
    dnum = int(rng.integers(1, 5))
    while alpha * dnum > 12:
        dnum -= 1
    size_q = alpha * dnum
    if size_q < 2:
        dnum, size_q = 2, 2 * alpha
    q_bits = [int(b) for b in rng.integers(36, 61, size_q)]
    p_bits = [int(b) for b in rng.integers(45, 61, alpha)]
    chain_index = int(rng.integers(1, size_q))          # size_Ql = size_q - chain_index + 1 >= 2
    step = int(rng.choice([1, -1, 2, 5, -7, 16]))
    return log_n, alpha, q_bits, p_bits, chain_index, step


@pytest.mark.parametrize("seed", range(32))
def test_evaluator_entry_points_on_a_random_geometry(gpu, pha, orc, seed):
    log_n, alpha, q_bits, p_bits, chain_index, step = _geometry(seed)
    n = 1 << log_n
    mods = orc.coeff_modulus_create(n, q_bits + p_bits)
    ctx, o = pha.PhantomContext(log_n, mods, alpha), orc.Oracle(log_n, mods, alpha)
    what = f"seed {seed}: N=2^{log_n} alpha={alpha} Q={q_bits} P={p_bits} chain_index={chain_index} step={step}"
    size_ql = ctx.size_Ql(chain_index)
    rows = list(range(size_ql))
    dnum = len(q_bits) // alpha
    evk_h = [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(o.size_QP)), 500 + 10 * d),
                                            o.uniform_poly(list(range(o.size_QP)), 501 + 10 * d)])) for d in range(dnum)]
    evk_d = pha.EvalKey([pha.to_device(h) for h in evk_h])
    dev, host = pha.to_device, lambda t: t.cpu().numpy().view(np.uint64)

    # key switch (modup -> inner product -> moddown -> add)
    ct = np.stack([o.uniform_poly(rows, 1), o.uniform_poly(rows, 2)])
    c2 = o.uniform_poly(rows, 3)
    ref = ct.copy()
    o.keyswitch_inplace(size_ql, ref, c2, evk_h)
    d = dev(ct)
    ctx.keyswitch_inplace(chain_index, d, dev(c2), evk_d)
    assert (host(d) == ref).all(), "keyswitch_inplace, " + what

    # HMult + relinearise
    a = np.stack([o.uniform_poly(rows, 4), o.uniform_poly(rows, 5)])
    b = np.stack([o.uniform_poly(rows, 6), o.uniform_poly(rows, 7)])
    t = o.tensor_prod_2x2(a, b)
    ref = t[:2].copy()
    o.keyswitch_inplace(size_ql, ref, t[2], evk_h)
    d = dev(np.concatenate([a, np.zeros_like(a[:1])]))
    ctx.multiply_and_relin_inplace(chain_index, d, dev(b), evk_d)
    assert (host(d)[:2] == ref).all(), "multiply_and_relin_inplace, " + what

    # rotation
    elt = pha.galois_elt_from_step(log_n, step)
    ref = np.stack([o.apply_galois_ntt(ct[0], elt), np.zeros_like(ct[1])])
    o.keyswitch_inplace(size_ql, ref, o.apply_galois_ntt(ct[1], elt), evk_h)
    d = dev(ct)
    ctx.apply_galois_inplace(chain_index, d, elt, evk_d)
    assert (host(d) == ref).all(), "apply_galois_inplace, " + what

    # rescale
    ref = o.divide_and_round_q_last_ntt(size_ql, ct, 2)
    d_src, d_dst = dev(ct), dev(np.zeros_like(ref))
    ctx.rescale_to_next(chain_index, d_src, 2, d_dst)
    assert (host(d_dst) == ref).all() and (host(d_src) == ct).all(), "rescale_to_next, " + what

    # merged HMult + relinearise + rescale, product and doubled square (oracle restatement from its primitives)
    from oracle.ckks_ref import multiply_relin_rescale_ref
    d_a, d_b = dev(a), dev(b)
    out = ctx.multiply_relin_rescale(chain_index, d_a, d_b, evk_d, False)
    assert (host(out) == multiply_relin_rescale_ref(o, size_ql, a, b, evk_h, False)).all(), "multiply_relin_rescale, " + what
    out = ctx.multiply_relin_rescale(chain_index, d_a, d_a, evk_d, True)
    assert (host(out) == multiply_relin_rescale_ref(o, size_ql, a, a, evk_h, True)).all(), "multiply_relin_rescale (square), " + what

    # rotations sharing one modup (a second uniform key set stands for the second Galois key)
    elts = [elt, pha.galois_elt_from_step(log_n, 3), 2 * n - 1]
    keys_h = [evk_h, [np.ascontiguousarray(k[::-1]) for k in evk_h], evk_h]
    keys_d = [pha.EvalKey([dev(k) for k in key]) for key in keys_h]
    outs = ctx.rotate_hoisted(chain_index, dev(ct), elts, keys_d)
    refs = o.rotate_hoisted(size_ql, ct, elts, keys_h)
    for e, got, exp in zip(elts, outs, refs):
        assert (host(got) == exp).all(), f"rotate_hoisted elt {e}, " + what
    ctx.close()
