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


@pytest.mark.parametrize("seed", range(28))
def test_transform_variants_on_a_random_window(gpu, pha, orc, seed):
    """The eight live transform entry points (include/ntt.cuh:47-94) on drawn shapes: N = 2^11 .. 2^17 (every tile plan: the
    in-LDS kernel, 6 + 6 .. 8 + 8 stages, the split N = 2^17 plan), mixed prime sizes, a drawn limb window [start, start +
    count), special-modulus table remap with and without an excluded range, out-of-place and scaled inverse, the fused
    moddown epilogue -- against the oracle word for word, and the words outside the window must stay as they were."""
    rng = np.random.default_rng(12000 + seed)
    log_n = [11, 12, 13, 14, 15, 16, 17][seed % 7]
    n = 1 << log_n
    size_p = int(rng.integers(1, 4))
    size_q = int(rng.integers(2, 7 if log_n < 16 else 5))
    bits = [int(b) for b in rng.integers(40, 61, size_q)] + [int(b) for b in rng.integers(50, 61, size_p)]
    mods = orc.coeff_modulus_create(n, bits)
    ctx, o = pha.PhantomContext(log_n, mods, size_p), orc.Oracle(log_n, mods, size_p)
    size_qp = size_q + size_p
    what = f"seed {seed}: N=2^{log_n} bits={bits} size_p={size_p}"
    dev, host = pha.to_device, lambda t: t.cpu().numpy().view(np.uint64)

    # plain window [start, start + count) of a size_qp buffer (buffers are passed whole: limb indices are absolute)
    count = int(rng.integers(1, size_qp + 1))
    start = int(rng.integers(0, size_qp - count + 1))
    win = slice(start, start + count)
    x = o.uniform_poly(list(range(size_qp)), 1)
    ref = x.copy()
    o.nwt_forward_inplace(ref, count, start)
    d = dev(x)
    ctx.nwt_2d_radix8_forward_inplace(d, count, start)
    assert (host(d) == ref).all(), f"forward window [{start}, {start + count}), " + what
    out = dev(np.zeros_like(x))
    ctx.nwt_2d_radix8_backward(out, d, count, start)
    got = host(out)
    assert (got[win] == x[win]).all() and not got[:start].any() and not got[start + count:].any(), "inverse out of place, " + what
    ctx.nwt_2d_radix8_backward_inplace(d, count, start)
    assert (host(d) == x).all(), "inverse in place, " + what
    scale = np.array([int(rng.integers(1, int(m))) for m in o.moduli[:size_qp]], dtype=np.uint64)
    shoup = np.array([(int(s_) << 64) // int(m) for s_, m in zip(scale, o.moduli[:size_qp])], dtype=np.uint64)
    sref = np.zeros_like(x)
    o.nwt_backward_scale(sref, ref, count, start, scale)
    sout = dev(np.zeros_like(x))
    ctx.nwt_2d_radix8_backward_scale(sout, dev(ref), count, start, dev(scale), dev(shoup))
    assert (host(sout) == sref).all(), "scaled inverse, " + what

    # Q_l u P buffer: the last size_p limbs use the key-level tables; an excluded range inside the Q_l part
    size_ql = int(rng.integers(1, size_q + 1))
    qlp = size_ql + size_p
    tidx = list(range(size_ql)) + list(range(size_q, size_qp))
    y = o.uniform_poly(tidx, 2)
    ref = y.copy()
    o.nwt_forward_inplace_special(ref, qlp, 0, size_qp, size_p)
    d = dev(y)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod(d, qlp, 0, size_qp, size_p)
    assert (host(d) == ref).all(), "special-modulus forward, " + what
    e0 = int(rng.integers(0, size_ql))
    e1 = int(rng.integers(e0 + 1, size_ql + 1))
    ref2 = y.copy()
    o.nwt_forward_inplace_special(ref2, qlp, 0, size_qp, size_p, e0, e1)
    d = dev(y)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(d, qlp, 0, size_qp, size_p, e0, e1)
    assert (host(d) == ref2).all() and (ref2[e0:e1] == y[e0:e1]).all(), f"excluded range [{e0}, {e1}), " + what
    ref3 = ref.copy()
    o.nwt_backward_inplace_special(ref3, size_p, size_ql, size_qp, size_p)
    d = dev(ref)
    ctx.nwt_2d_radix8_backward_inplace_include_special_mod(d, size_p, size_ql, size_qp, size_p)
    assert (host(d) == ref3).all() and (ref3[size_ql:] == y[size_ql:]).all(), "special-modulus inverse, " + what

    # forward transform of delta fused with (cx - .) * P^-1
    lc = o.level_constants(size_ql)
    pinv = lc["bigPInv_mod_q"]
    pshoup = np.array([(int(s) << 64) // int(m) for s, m in zip(pinv, o.moduli[:size_ql])], dtype=np.uint64)
    delta, cx = o.uniform_poly(list(range(size_ql)), 3), o.uniform_poly(list(range(size_ql)), 4)
    fref = np.zeros_like(cx)
    o.nwt_forward_fuse_moddown(fref, cx, pinv, delta.copy(), size_ql, 0)
    dct = dev(np.zeros_like(cx))
    ctx.nwt_2d_radix8_forward_inplace_fuse_moddown(dct, dev(cx), dev(pinv), dev(pshoup), dev(delta), size_ql, 0)
    assert (host(dct) == fref).all(), "fused moddown epilogue, " + what
    ctx.close()
