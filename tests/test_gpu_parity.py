"""GPU parity: every C-ABI compute entry point against the CPU oracle on the same seeded inputs
(bit-exact — integer work), through the C ABI (ctypes), on a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(pha, a):
    return pha.to_device(a)


def _host(pha, t):
    return pha.to_host(t)


@pytest.fixture(scope="module")
def small(gpu, pha, orc):
    """N = 4096, Q = 6 x 45-bit, P = 2 x 55-bit: alpha = 2, beta = 3."""
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * 2)
    return pha.PhantomContext(12, mods, 2), orc.Oracle(12, mods, 2)


@pytest.mark.parametrize("log_n", [11, 12, 13, 14, 15, 16, 17])
def test_ntt_forward_backward_all_sizes(gpu, pha, orc, log_n):
    n = 1 << log_n
    mods = orc.coeff_modulus_create(n, [60, 50, 40, 60])
    ctx, o = pha.PhantomContext(log_n, mods, 1), orc.Oracle(log_n, mods, 1)
    for which in range(4):
        assert (ctx.get_table(2, which) == o.table(which, 2)).all()
    x = o.uniform_poly([0, 1, 2, 3], 100 + log_n)
    ref = x.copy()
    o.nwt_forward_inplace(ref, 4, 0)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace(d, 4, 0)
    assert (_host(pha, d) == ref).all()
    ctx.nwt_2d_radix8_backward_inplace(d, 4, 0)
    assert (_host(pha, d) == x).all()
    # sub-range with a start index: limbs 1..2 only
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace(d, 2, 1)
    got = _host(pha, d)
    assert (got[1:3] == ref[1:3]).all() and (got[0] == x[0]).all() and (got[3] == x[3]).all()
    ctx.close()


def test_n131072_key_switch_rescale_galois(gpu, pha, orc):
    """N = 2^17 (the reference's largest SAMPLE_SIZE case, include/ntt.cuh:19): the outermost stage runs in its own
    kernel and the column kernels work on two subtrees — every fused variant of the transform (scale epilogue in the
    modup INTT, moddown and rescale epilogues, digit exclusion, special-modulus indices) goes through that plan."""
    log_n, n, size_p = 17, 1 << 17, 2
    mods = orc.coeff_modulus_create(n, [45] * 4 + [55] * size_p)        # FP64 limbs and integer limbs
    ctx, o = pha.PhantomContext(log_n, mods, size_p), orc.Oracle(log_n, mods, size_p)
    size_ql = 4
    x = o.uniform_poly([0], 1)
    ref = x.copy()
    o.nwt_forward_inplace(ref, 1, 0)
    d = _dev(pha, x + np.uint64(3 * int(o.moduli[0])))                  # lazy input
    ctx.nwt_2d_radix8_forward_inplace(d, 1, 0)
    assert (_host(pha, d) == ref).all()
    evk_h, evk_d = _uniform_evk(pha, o, 170)
    c2 = o.uniform_poly(list(range(size_ql)), 6)
    ct = np.stack([o.uniform_poly(list(range(size_ql)), 7), o.uniform_poly(list(range(size_ql)), 8)])
    ct_ref = ct.copy()
    o.keyswitch_inplace(size_ql, ct_ref, c2, evk_h)
    d_ct = _dev(pha, ct)
    ctx.keyswitch_inplace(1, d_ct, _dev(pha, c2), evk_d)
    assert (_host(pha, d_ct) == ct_ref).all()
    ref = o.divide_and_round_q_last_ntt(size_ql, ct, 2)
    d_dst = _dev(pha, np.zeros_like(ref))
    ctx.rescale_to_next(1, _dev(pha, ct), 2, d_dst)
    assert (_host(pha, d_dst) == ref).all()
    elt = pha.galois_elt_from_step(log_n, 3)
    d_out = _dev(pha, np.zeros_like(ct[0]))
    ctx.apply_galois_ntt(_dev(pha, ct[0]), size_ql, elt, d_out)
    assert (_host(pha, d_out) == o.apply_galois_ntt(ct[0], elt)).all()
    ctx.close()


@pytest.mark.parametrize("size_p", [1, 2])
def test_n2048_every_transform_variant(gpu, pha, orc, size_p):
    """N = 2^11 (the smallest SAMPLE_SIZE case, include/ntt.cuh:16-24): the polynomial is smaller than a tile, one workgroup
    transforms a limb inside LDS.  Tables, forward / inverse (in place, out of place, scaled), and through the key switch,
    rotation and rescale every fused variant (scale epilogue of the modup INTT, digit exclusion, special-modulus indices,
    moddown and rescale epilogues) against the oracle."""
    log_n, n = 11, 2048
    mods = orc.coeff_modulus_create(n, [55, 45, 45, 45] + [56] * size_p)
    ctx, o = pha.PhantomContext(log_n, mods, size_p), orc.Oracle(log_n, mods, size_p)
    for which in range(4):
        assert (ctx.get_table(1, which) == o.table(which, 1)).all()
    size_ql = 4
    lim = list(range(size_ql))
    x = o.uniform_poly(lim, 5)
    ref = x.copy()
    o.nwt_forward_inplace(ref, size_ql, 0)
    d = _dev(pha, x + (np.uint64(3) * o.moduli[:size_ql, None]).astype(np.uint64))     # lazy input
    ctx.nwt_2d_radix8_forward_inplace(d, size_ql, 0)
    assert (_host(pha, d) == ref).all()
    ctx.nwt_2d_radix8_backward_inplace(d, size_ql, 0)
    assert (_host(pha, d) == x).all()
    d_out = _dev(pha, np.zeros_like(x))
    ctx.nwt_2d_radix8_backward(d_out, _dev(pha, ref), size_ql, 0)
    assert (_host(pha, d_out) == x).all()
    evk_h, evk_d = _uniform_evk(pha, o, 70)
    for chain_index in (1, 2):
        ql = size_ql - (chain_index - 1)
        lim = list(range(ql))
        ct = np.stack([o.uniform_poly(lim, 7), o.uniform_poly(lim, 8)])
        c2 = o.uniform_poly(lim, 6)
        ct_ref = ct.copy()
        o.keyswitch_inplace(ql, ct_ref, c2, evk_h)
        d_ct = _dev(pha, ct)
        ctx.keyswitch_inplace(chain_index, d_ct, _dev(pha, c2), evk_d)
        assert (_host(pha, d_ct) == ct_ref).all()
        rs_ref = o.divide_and_round_q_last_ntt(ql, ct, 2)
        d_rs = _dev(pha, np.zeros_like(rs_ref))
        ctx.rescale_to_next(chain_index, _dev(pha, ct), 2, d_rs)
        assert (_host(pha, d_rs) == rs_ref).all()
    elt = orc.galois_elt_from_step(log_n, 3)
    ct = np.stack([o.uniform_poly(list(range(size_ql)), 17), o.uniform_poly(list(range(size_ql)), 18)])
    rot_ref = np.stack([o.apply_galois_ntt(ct[0], elt), np.zeros_like(ct[1])])
    o.keyswitch_inplace(size_ql, rot_ref, o.apply_galois_ntt(ct[1], elt), evk_h)
    d_rot = _dev(pha, ct)
    ctx.apply_galois_inplace(1, d_rot, elt, evk_d)
    assert (_host(pha, d_rot) == rot_ref).all()
    ctx.close()


def test_block_placement_does_not_change_results(gpu, pha, orc):
    """pha_ntt_block_placement: the limb-to-XCD block map and the tile-interleaved one are two orders of the same work
    (several polys, a limb count that is not a multiple of 8, a start offset)."""
    import torch
    log_n, n = 13, 8192
    mods = orc.coeff_modulus_create(n, [60] + [45] * 10 + [55] * 2)
    ctx, o = pha.PhantomContext(log_n, mods, 2), orc.Oracle(log_n, mods, 2)
    x = np.stack([o.uniform_poly(list(range(13)), 900 + p) for p in range(3)])
    outs = []
    for placement in (1, 0):
        ctx.ntt_block_placement(placement)
        d = _dev(pha, x)
        ctx.ntt_forward_batched(d, 3, 13, 11, 1)             # limbs 1..11 of three polys
        f = d.clone()
        ctx.ntt_backward_batched(d, 3, 13, 11, 1)
        assert torch.equal(d, _dev(pha, x))
        outs.append(f)
    assert torch.equal(outs[0], outs[1])
    ref = x[0].copy()
    o.nwt_forward_inplace(ref, 11, 1)
    assert (_host(pha, outs[0])[0] == ref).all()
    ctx.close()


def test_random_batches_against_oracle(gpu, pha, orc):
    """Seeded sweep over the launch shapes of the batched transforms — polys 1..48 (the block decode divides by the poly
    count with a multiply-by-reciprocal), limb counts that are and are not multiples of 8, start offsets, both block
    placements, N = 2^12 and 2^13 — forward against the oracle limb by limb, then the round trip."""
    import torch
    rng = np.random.default_rng(2024)
    for log_n in (12, 13):
        n = 1 << log_n
        mods = orc.coeff_modulus_create(n, [60, 50, 45, 40, 50, 50, 45, 60, 50, 50, 45, 50, 50])
        ctx, o = pha.PhantomContext(log_n, mods, 1), orc.Oracle(log_n, mods, 1)
        limbs = len(mods)
        base = np.stack([o.uniform_poly(list(range(limbs)), 3000 + p) for p in range(4)])
        ref = base.copy()
        for p in range(4):
            o.nwt_forward_inplace(ref[p], limbs, 0)
        for case in range(14):
            polys = int(rng.choice([1, 2, 3, 5, 7, 8, 16, 31, 48]))
            count = int(rng.integers(1, limbs + 1))
            start = int(rng.integers(0, limbs - count + 1))
            ctx.ntt_block_placement(case & 1)
            x = np.stack([base[p % 4] for p in range(polys)])
            d = _dev(pha, x)
            ctx.ntt_forward_batched(d, polys, limbs, count, start)
            got = _host(pha, d)
            for p in range(polys):
                assert (got[p, start:start + count] == ref[p % 4, start:start + count]).all(), (log_n, polys, count, start, p)
                untouched = np.ones(limbs, dtype=bool)
                untouched[start:start + count] = False
                assert (got[p, untouched] == x[p, untouched]).all(), (log_n, polys, count, start, p)
            ctx.ntt_backward_batched(d, polys, limbs, count, start)
            assert torch.equal(d, _dev(pha, x)), (log_n, polys, count, start)
        ctx.close()


def test_forward_tolerates_lazy_inputs(gpu, pha, small):
    ctx, o = small
    x = o.uniform_poly([0], 1)
    ref = x.copy()
    o.nwt_forward_inplace(ref, 1, 0)
    d = _dev(pha, x + np.uint64(3 * int(o.moduli[0])))
    ctx.nwt_2d_radix8_forward_inplace(d, 1, 0)
    assert (_host(pha, d) == ref).all()


def test_backward_out_of_place_and_scale(gpu, pha, small):
    ctx, o = small
    x = o.uniform_poly(list(range(6)), 2)
    ref = np.zeros_like(x)
    o.nwt_backward(ref, x, 6, 0)
    din, dout = _dev(pha, x), _dev(pha, np.zeros_like(x))
    ctx.nwt_2d_radix8_backward(dout, din, 6, 0)
    assert (_host(pha, dout) == ref).all() and (_host(pha, din) == x).all()
    scale = np.array([int(m) // 3 + 17 for m in o.moduli[:6]], dtype=np.uint64)
    shoup = np.array([(int(s) << 64) // int(m) for s, m in zip(scale, o.moduli[:6])], dtype=np.uint64)
    o.nwt_backward_scale(ref, x, 6, 0, scale)
    ctx.nwt_2d_radix8_backward_scale(dout, din, 6, 0, _dev(pha, scale), _dev(pha, shoup))
    assert (_host(pha, dout) == ref).all()


def test_special_mod_variants(gpu, pha, small):
    ctx, o = small
    size_ql = 4
    tidx = list(range(size_ql)) + [6, 7]
    x = o.uniform_poly(tidx, 3)
    ref = x.copy()
    o.nwt_forward_inplace_special(ref, 6, 0, 8, 2)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod(d, 6, 0, 8, 2)
    assert (_host(pha, d) == ref).all()
    ref2 = x.copy()
    o.nwt_forward_inplace_special(ref2, 6, 0, 8, 2, 2, 4)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(d, 6, 0, 8, 2, 2, 4)
    assert (_host(pha, d) == ref2).all()
    ref3 = x.copy()
    o.nwt_backward_inplace_special(ref3, 2, size_ql, 8, 2)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_backward_inplace_include_special_mod(d, 2, size_ql, 8, 2)
    assert (_host(pha, d) == ref3).all()
    with pytest.raises(pha.PhantomError) as e:
        ctx.nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(d, 4, 1, 8, 2, 0, 2)
    assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    with pytest.raises(pha.PhantomError):
        ctx.nwt_2d_radix8_forward_inplace(d, 9, 0)


def test_fuse_moddown(gpu, pha, small):
    ctx, o = small
    size_ql = 5
    lc = o.level_constants(size_ql)
    pinv = lc["bigPInv_mod_q"]
    shoup = np.array([(int(s) << 64) // int(m) for s, m in zip(pinv, o.moduli[:size_ql])], dtype=np.uint64)
    delta, cx = o.uniform_poly(list(range(size_ql)), 4), o.uniform_poly(list(range(size_ql)), 5)
    ref = np.zeros_like(cx)
    o.nwt_forward_fuse_moddown(ref, cx, pinv, delta.copy(), size_ql, 0)
    dct = _dev(pha, np.zeros_like(cx))
    ctx.nwt_2d_radix8_forward_inplace_fuse_moddown(dct, _dev(pha, cx), _dev(pha, pinv), _dev(pha, shoup),
                                                   _dev(pha, delta), size_ql, 0)
    assert (_host(pha, dct) == ref).all()


def _uniform_evk(pha, o, seed):
    dnum = o.size_Q // o.size_P
    host = [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(o.size_QP)), seed + 10 * d),
                                            o.uniform_poly(list(range(o.size_QP)), seed + 10 * d + 1)]))
            for d in range(dnum)]
    return host, pha.EvalKey([pha.to_device(h) for h in host])


@pytest.mark.parametrize("size_p,chain_index", [(2, 1), (2, 2), (2, 4), (1, 1), (1, 3), (3, 1), (3, 3)])
def test_keyswitch_pipeline(gpu, pha, orc, size_p, chain_index):
    """modup, inner product, moddown and the whole keyswitch_inplace, alpha = 1 and alpha > 1,
    full and ragged last digit.  A uniform evk is enough for arithmetic parity."""
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = ctx.size_Ql(chain_index)
    size_qlp = size_ql + size_p
    evk_h, evk_d = _uniform_evk(pha, o, 70)
    c2 = o.uniform_poly(list(range(size_ql)), 6)
    t_ref = o.modup(size_ql, c2)
    d_t = _dev(pha, np.zeros_like(t_ref))
    ctx.modup(chain_index, d_t, _dev(pha, c2))
    assert (_host(pha, d_t) == t_ref).all()
    cx_ref = o.key_switch_inner_prod(size_ql, t_ref, evk_h)
    d_cx = _dev(pha, np.zeros_like(cx_ref))
    ctx.key_switch_inner_prod(chain_index, d_cx, d_t, evk_d)
    assert (_host(pha, d_cx) == cx_ref).all()
    for i in range(2):
        md_ref = o.moddown_from_ntt(size_ql, cx_ref[i])
        d_ct = _dev(pha, np.zeros((size_ql, 4096), dtype=np.uint64))
        ctx.moddown_from_NTT(chain_index, d_ct, d_cx[i])
        assert (_host(pha, d_ct) == md_ref).all()
    ct = np.stack([o.uniform_poly(list(range(size_ql)), 7), o.uniform_poly(list(range(size_ql)), 8)])
    ct_ref = ct.copy()
    o.keyswitch_inplace(size_ql, ct_ref, c2, evk_h)
    d_ct = _dev(pha, ct)
    ctx.keyswitch_inplace(chain_index, d_ct, _dev(pha, c2), evk_d)
    assert (_host(pha, d_ct) == ct_ref).all()
    # add_to_ct on its own
    a, b = ct[0], ct[1]
    d_a = _dev(pha, a)
    ctx.add_to_ct(d_a, _dev(pha, b), size_ql)
    exp = np.stack([((a[i].astype(object) + b[i].astype(object)) % int(mods[i])) for i in range(size_ql)])
    assert (_host(pha, d_a).astype(object) == exp).all()
    assert size_qlp == t_ref.shape[1]
    ctx.close()


def test_one_context_on_two_streams_is_serialised(gpu, pha, orc):
    """Two streams taking turns on ONE context share its workspace: the library makes the second stream wait for what the
    first enqueued (ensure_workspace) instead of letting the key switches race on t_mod_up / cx.  N = 2^15 so that each
    call runs long enough to overlap if nothing ordered them."""
    log_n, n, size_p = 15, 1 << 15, 2
    mods = orc.coeff_modulus_create(n, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(log_n, mods, size_p), orc.Oracle(log_n, mods, size_p)
    evk_h, evk_d = _uniform_evk(pha, o, 70)
    lim = list(range(6))
    cts = [np.stack([o.uniform_poly(lim, 20 + 3 * i), o.uniform_poly(lim, 21 + 3 * i)]) for i in range(2)]
    c2s = [o.uniform_poly(lim, 22 + 3 * i) for i in range(2)]
    refs = []
    for ct, c2 in zip(cts, c2s):
        r = ct.copy()
        o.keyswitch_inplace(6, r, c2, evk_h)
        refs.append(r)
    streams = [gpu.cuda.Stream(), gpu.cuda.Stream()]
    d_cts = [[_dev(pha, ct) for _ in range(4)] for ct in cts]
    d_c2s = [_dev(pha, c2) for c2 in c2s]
    ctx.keyswitch_inplace(1, _dev(pha, cts[0]), d_c2s[0], evk_d)          # tables and workspace
    gpu.cuda.synchronize()
    for rep in range(4):
        for i in range(2):
            with gpu.cuda.stream(streams[i]):
                ctx.keyswitch_inplace(1, d_cts[i][rep], d_c2s[i], evk_d)
    gpu.cuda.synchronize()
    for i in range(2):
        for rep in range(4):
            assert (_host(pha, d_cts[i][rep]) == refs[i]).all(), (i, rep)
    ctx.close()


def test_workspace_growth_under_capture_is_refused(gpu, pha, orc):
    """A call that would have to grow the workspace (device synchronisation + allocation) while its stream is capturing
    returns PHA_ERR_LOGIC before touching the capture; after one eager call the same call captures and replays exactly."""
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * 2)
    ctx, o = pha.PhantomContext(12, mods, 2), orc.Oracle(12, mods, 2)
    evk_h, evk_d = _uniform_evk(pha, o, 70)
    lim = list(range(6))
    ct = np.stack([o.uniform_poly(lim, 7), o.uniform_poly(lim, 8)])
    c2 = o.uniform_poly(lim, 6)
    ref = ct.copy()
    o.keyswitch_inplace(6, ref, c2, evk_h)
    ctx.prepare_level(1)                                                    # tables yes, workspace no
    d_ct, d_c2 = _dev(pha, ct), _dev(pha, c2)
    gpu.cuda.synchronize()
    graph = gpu.cuda.CUDAGraph()
    with gpu.cuda.graph(graph):
        with pytest.raises(pha.PhantomError) as err:
            ctx.keyswitch_inplace(1, d_ct, d_c2, evk_d)
    assert err.value.code == pha.PHA_ERR_LOGIC and "capturing" in str(err.value)
    ctx.keyswitch_inplace(1, _dev(pha, ct), d_c2, evk_d)
    gpu.cuda.synchronize()
    graph = gpu.cuda.CUDAGraph()
    with gpu.cuda.graph(graph):
        ctx.keyswitch_inplace(1, d_ct, d_c2, evk_d)
    graph.replay()
    gpu.cuda.synchronize()
    assert (_host(pha, d_ct) == ref).all()
    ctx.close()


def test_galois_and_rotate(gpu, pha, orc, small):
    ctx, o = small
    size_ql = 6
    x = o.uniform_poly(list(range(size_ql)), 9)
    for step in (1, -3, 0):
        elt = pha.galois_elt_from_step(12, step)
        ref = o.apply_galois_ntt(x, elt)
        d_out = _dev(pha, np.zeros_like(x))
        ctx.apply_galois_ntt(_dev(pha, x), size_ql, elt, d_out)
        assert (_host(pha, d_out) == ref).all()
        d_in = _dev(pha, x)
        ctx.apply_galois_ntt(d_in, size_ql, elt, d_in)  # in place
        assert (_host(pha, d_in) == ref).all()
    # apply_galois_inplace = permute both polys, key-switch the permuted c1 (src/evaluate.cu:747-794)
    evk_h, evk_d = _uniform_evk(pha, o, 80)
    ct = np.stack([o.uniform_poly(list(range(size_ql)), 10), o.uniform_poly(list(range(size_ql)), 11)])
    elt = 5
    ref = np.stack([o.apply_galois_ntt(ct[0], elt), np.zeros_like(ct[1])])
    o.keyswitch_inplace(size_ql, ref, o.apply_galois_ntt(ct[1], elt), evk_h)
    d_ct = _dev(pha, ct)
    ctx.apply_galois_inplace(1, d_ct, elt, evk_d)
    assert (_host(pha, d_ct) == ref).all()
    with pytest.raises(pha.PhantomError):
        ctx.apply_galois_ntt(d_ct, size_ql, 6, d_ct)


@pytest.mark.parametrize("chain_index", [1, 3])
def test_rescale(gpu, pha, small, chain_index):
    ctx, o = small
    size_ql = ctx.size_Ql(chain_index)
    src = np.stack([o.uniform_poly(list(range(size_ql)), 12), o.uniform_poly(list(range(size_ql)), 13)])
    ref = o.divide_and_round_q_last_ntt(size_ql, src, 2)
    d_src, d_dst = _dev(pha, src), _dev(pha, np.zeros_like(ref))
    ctx.rescale_to_next(chain_index, d_src, 2, d_dst)
    assert (_host(pha, d_dst) == ref).all()
    assert (_host(pha, d_src) == src).all()  # rescale_to_next preserves its input
    d_dst2 = _dev(pha, np.zeros_like(ref))
    ctx.divide_and_round_q_last_ntt(chain_index, d_src, 2, d_dst2)  # reference semantics: src destroyed
    assert (_host(pha, d_dst2) == ref).all()


def test_tensor_and_multiply_relin(gpu, pha, small):
    ctx, o = small
    size_ql = 6
    a = np.stack([o.uniform_poly(list(range(size_ql)), 14), o.uniform_poly(list(range(size_ql)), 15)])
    b = np.stack([o.uniform_poly(list(range(size_ql)), 16), o.uniform_poly(list(range(size_ql)), 17)])
    ref = o.tensor_prod_2x2(a, b)
    d_res = _dev(pha, np.zeros_like(ref))
    ctx.tensor_prod_2x2(_dev(pha, a), _dev(pha, b), size_ql, d_res)
    assert (_host(pha, d_res) == ref).all()
    ctx.tensor_square_2x2(_dev(pha, a), size_ql, d_res)
    assert (_host(pha, d_res) == o.tensor_prod_2x2(a, a)).all()
    # multiply_inplace + relinearize_inplace (src/evaluate.cu:229-280,553-583)
    evk_h, evk_d = _uniform_evk(pha, o, 90)
    exp = ref[:2].copy()
    o.keyswitch_inplace(size_ql, exp, ref[2], evk_h)
    d_ct1 = _dev(pha, np.concatenate([a, np.zeros_like(a[:1])]))
    ctx.multiply_and_relin_inplace(1, d_ct1, _dev(pha, b), evk_d)
    assert (_host(pha, d_ct1)[:2] == exp).all()


@pytest.mark.parametrize("m,nn", [(3, 2), (3, 3), (2, 2), (1, 3), (4, 4)])
def test_tensor_prod_mxn(gpu, pha, small, m, nn):
    """tensor_prod_mxn_rns_poly (src/polymath.cu:551-599): what multiply_internal runs when an operand has more than
    two polys; in place on operand 1 like the reference (src/evaluate.cu:270-275)."""
    ctx, o = small
    size_ql = 5
    lim = list(range(size_ql))
    a = np.stack([o.uniform_poly(lim, 300 + i) for i in range(m)])
    b = np.stack([o.uniform_poly(lim, 320 + i) for i in range(nn)])
    ref = o.tensor_prod_mxn(a, b)
    if (m, nn) == (2, 2):
        assert (ref == o.tensor_prod_2x2(a, b)).all()          # the 2x2 kernels are the same map
    d_res = _dev(pha, np.zeros_like(ref))
    ctx.tensor_prod_mxn(_dev(pha, a), _dev(pha, b), d_res)
    assert (_host(pha, d_res) == ref).all()
    # in place: the result overwrites operand 1 (buffer sized for the product)
    d_inplace = _dev(pha, np.concatenate([a, np.zeros((nn - 1, size_ql, a.shape[-1]), dtype=np.uint64)]))
    ctx.tensor_prod_mxn(d_inplace[:m], _dev(pha, b), d_inplace)
    assert (_host(pha, d_inplace) == ref).all()
    with pytest.raises(pha.PhantomError):
        ctx.tensor_prod_mxn(_dev(pha, np.zeros((5, size_ql, a.shape[-1]), dtype=np.uint64)), _dev(pha, b), d_res)


@pytest.mark.parametrize("size_p,chain_index", [(2, 1), (2, 3), (1, 2), (3, 1)])
def test_double_hoisted_linear_transform(gpu, pha, orc, size_p, chain_index):
    """pha_linear_transform_double_hoisted against its restatement from the oracle's primitives (bit-exact; uniform
    stand-ins for keys and plaintexts are enough for arithmetic parity): identity and rotating babies / giants,
    absent diagonals, alpha = 1 and alpha > 1, ragged last digit."""
    from oracle.ckks_ref import linear_transform_double_hoisted_ref
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = ctx.size_Ql(chain_index)
    tq = list(range(size_ql))
    tqp = tq + [o.size_Q + i for i in range(size_p)]
    ct = np.stack([o.uniform_poly(tq, 401), o.uniform_poly(tq, 402)])
    baby_elts = [1] + [orc.galois_elt_from_step(12, s) for s in (1, 2)]
    # three rotating giant steps: two run on the library's helper stream chain, results must not depend on it
    giant_elts = [1] + [orc.galois_elt_from_step(12, s) for s in (4, 8, 12)]
    keys = {e: _uniform_evk(pha, o, 500 + 7 * i) for i, e in enumerate(baby_elts[1:] + giant_elts[1:])}
    plains = [[o.uniform_poly(tqp, 600 + 10 * g + b) if (g, b) != (1, 1) else None for b in range(3)] for g in range(4)]
    ref = linear_transform_double_hoisted_ref(o, size_ql, ct, baby_elts, [keys[e][0] if e != 1 else None for e in baby_elts],
                                              giant_elts, [keys[e][0] if e != 1 else None for e in giant_elts], plains)
    d_pl = [[_dev(pha, p) if p is not None else None for p in row] for row in plains]
    out = ctx.linear_transform_double_hoisted(chain_index, _dev(pha, ct), baby_elts,
                                              [keys[e][1] if e != 1 else None for e in baby_elts], giant_elts,
                                              [keys[e][1] if e != 1 else None for e in giant_elts], d_pl)
    assert (_host(pha, out) == ref).all()
    # a transform whose only giant step rotates, and one without any rotation
    out = ctx.linear_transform_double_hoisted(chain_index, _dev(pha, ct), baby_elts[:2], [None, keys[baby_elts[1]][1]],
                                              giant_elts[1:2], [keys[giant_elts[1]][1]], [d_pl[1][:2]])
    ref = linear_transform_double_hoisted_ref(o, size_ql, ct, baby_elts[:2], [None, keys[baby_elts[1]][0]], giant_elts[1:2],
                                              [keys[giant_elts[1]][0]], [plains[1][:2]])
    assert (_host(pha, out) == ref).all()
    out = ctx.linear_transform_double_hoisted(chain_index, _dev(pha, ct), [1], [None], [1], [None], [[d_pl[0][0]]])
    ref = linear_transform_double_hoisted_ref(o, size_ql, ct, [1], [None], [1], [None], [[plains[0][0]]])
    assert (_host(pha, out) == ref).all()
    with pytest.raises(pha.PhantomError):
        ctx.linear_transform_double_hoisted(chain_index, _dev(pha, ct), [1], [None], [1], [None], [[None]])
    # an error after the helper streams were forked (the last rotating giant step has no key) still joins them: captured
    # as a HIP graph, the failing call leaves nothing unjoined (the capture ends cleanly) and the next call is exact
    bad_keys = [keys[e][1] if e != 1 else None for e in giant_elts[:-1]] + [None]
    args = (chain_index, _dev(pha, ct), baby_elts, [keys[e][1] if e != 1 else None for e in baby_elts], giant_elts)
    ctx.linear_transform_double_hoisted(*args, [keys[e][1] if e != 1 else None for e in giant_elts], d_pl)   # tables, workspace
    gpu.cuda.synchronize()
    graph = gpu.cuda.CUDAGraph()
    with gpu.cuda.graph(graph):
        with pytest.raises(pha.PhantomError) as err:
            ctx.linear_transform_double_hoisted(*args, bad_keys, d_pl)
        out = ctx.linear_transform_double_hoisted(*args, [keys[e][1] if e != 1 else None for e in giant_elts], d_pl)
    assert err.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    graph.replay()
    gpu.cuda.synchronize()
    ref = linear_transform_double_hoisted_ref(o, size_ql, ct, baby_elts, [keys[e][0] if e != 1 else None for e in baby_elts],
                                              giant_elts, [keys[e][0] if e != 1 else None for e in giant_elts], plains)
    assert (_host(pha, out) == ref).all()
    ctx.close()


@pytest.mark.parametrize("size_p,chain_index,batch", [(2, 1, 5), (2, 2, 3), (3, 1, 4), (1, 1, 3), (3, 3, 2), (3, 5, 4), (3, 6, 3), (2, 6, 5)])
def test_batched_rotation_equals_one_by_one(gpu, pha, orc, size_p, chain_index, batch):
    """pha_apply_galois_batched = `batch` x pha_apply_galois_inplace, bit for bit (and so the oracle's result):
    uniform digits and a single short digit (size_Ql < alpha) take the batched kernels, alpha = 1 and several digits
    with a ragged last one the per-ciphertext path."""
    import torch
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = ctx.size_Ql(chain_index)
    evk_h, evk_d = _uniform_evk(pha, o, 900)
    cts = np.stack([np.stack([o.uniform_poly(list(range(size_ql)), 910 + 2 * b), o.uniform_poly(list(range(size_ql)), 911 + 2 * b)])
                    for b in range(batch)])
    elt = pha.galois_elt_from_step(12, 3)
    d_in = _dev(pha, cts)
    out = ctx.apply_galois_batched(chain_index, d_in, elt, evk_d)
    assert (_host(pha, d_in) == cts).all()
    for b in range(batch):
        one = _dev(pha, cts[b])
        ctx.apply_galois_inplace(chain_index, one, elt, evk_d)
        assert torch.equal(out[b], one)
    ref = np.stack([o.apply_galois_ntt(cts[0][0], elt), np.zeros_like(cts[0][1])])
    o.keyswitch_inplace(size_ql, ref, o.apply_galois_ntt(cts[0][1], elt), evk_h)
    assert (_host(pha, out[0]) == ref).all()
    ctx.close()


@pytest.mark.parametrize("size_p,chain_index", [(2, 1), (2, 3), (1, 2), (3, 1)])
def test_merged_multiply_relin_rescale(gpu, pha, orc, size_p, chain_index):
    """pha_multiply_relin_rescale against its restatement from the oracle's primitives (bit-exact), product and
    square, with and without the doubling"""
    from oracle.ckks_ref import multiply_relin_rescale_ref
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = ctx.size_Ql(chain_index)
    tq = list(range(size_ql))
    a = np.stack([o.uniform_poly(tq, 701), o.uniform_poly(tq, 702)])
    b = np.stack([o.uniform_poly(tq, 703), o.uniform_poly(tq, 704)])
    evk_h, evk_d = _uniform_evk(pha, o, 710)
    d_a, d_b = _dev(pha, a), _dev(pha, b)
    for times2 in (False, True):
        out = ctx.multiply_relin_rescale(chain_index, d_a, d_b, evk_d, times2)
        assert (_host(pha, out) == multiply_relin_rescale_ref(o, size_ql, a, b, evk_h, times2)).all()
    out = ctx.multiply_relin_rescale(chain_index, d_a, d_a, evk_d, True)          # square
    assert (_host(pha, out) == multiply_relin_rescale_ref(o, size_ql, a, a, evk_h, True)).all()
    assert (_host(pha, d_a) == a).all() and (_host(pha, d_b) == b).all()           # operands preserved
    with pytest.raises(pha.PhantomError):
        ctx.multiply_relin_rescale(ctx.size_Q, d_a[:, :1].contiguous(), d_a[:, :1].contiguous(), evk_d)
    ctx.close()


def test_bad_chain_index(gpu, pha, small):
    ctx, o = small
    import torch
    d = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
    with pytest.raises(pha.PhantomError) as e:
        ctx.modup(0, d, d)
    assert e.value.code == pha.PHA_ERR_INVALID_ARGUMENT
    with pytest.raises(pha.PhantomError):
        ctx.modup(7, d, d)
    with pytest.raises(pha.PhantomError):
        ctx.rescale_to_next(6, d, 2, d)  # one limb left: end of chain


def test_61_bit_modulus_uses_the_wide_path(gpu, pha, orc):
    """Moduli may be 61 bits (MOD_BIT_COUNT_MAX, include/util/defines.h:4): then 16q overflows a word and
    the kernels fall back to the [0,8q) lazy range.  Mixed chain: one 61-bit and one 60-bit prime."""
    n = 4096
    q61 = (1 << 61) - 2 * n + 1
    while not orc.lib().orc_is_prime(q61):
        q61 -= 2 * n
    assert q61 >> 60 == 1
    mods = np.array([q61, int(orc.coeff_modulus_create(n, [60])[0])], dtype=np.uint64)
    ctx, o = pha.PhantomContext(12, mods, 0), orc.Oracle(12, mods, 0)
    x = o.uniform_poly([0, 1], 321)
    x[0, :4] = q61 - 1  # extreme residues
    ref = x.copy()
    o.nwt_forward_inplace(ref, 2, 0)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace(d, 2, 0)
    assert (_host(pha, d) == ref).all()
    ctx.nwt_2d_radix8_backward_inplace(d, 2, 0)
    assert (_host(pha, d) == x).all()
    ctx.close()


def test_extreme_residues(gpu, pha, small):
    """All coefficients q-1 (and 0): the lazy ranges are at their upper bounds in every stage."""
    ctx, o = small
    size = 8
    tidx = list(range(8))
    x = np.stack([np.full(4096, int(o.moduli[t]) - 1, dtype=np.uint64) for t in tidx])
    x[:, ::7] = 0
    ref = x.copy()
    o.nwt_forward_inplace(ref, size, 0)
    d = _dev(pha, x)
    ctx.nwt_2d_radix8_forward_inplace(d, size, 0)
    assert (_host(pha, d) == ref).all()
    ctx.nwt_2d_radix8_backward_inplace(d, size, 0)
    assert (_host(pha, d) == x).all()


@pytest.mark.parametrize("size_p", [2, 3])
def test_keyswitch_60bit_primes_extreme_residues(gpu, pha, orc, size_p):
    """All primes 60 bits, inputs q-1 everywhere: the 128-bit column accumulators of the base
    conversion and the inner product see their largest carries."""
    mods = orc.coeff_modulus_create(4096, [60] * (6 + size_p))
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = 6
    evk_h = [np.ascontiguousarray(np.stack([np.stack([np.full(4096, int(m) - 1, dtype=np.uint64) for m in mods])] * 2))
             for _ in range(o.size_Q // size_p)]
    evk_d = pha.EvalKey([pha.to_device(h) for h in evk_h])
    c2 = np.stack([np.full(4096, int(m) - 1, dtype=np.uint64) for m in mods[:size_ql]])
    ct = np.stack([c2, c2])
    ref = ct.copy()
    o.keyswitch_inplace(size_ql, ref, c2, evk_h)
    d = _dev(pha, ct)
    ctx.keyswitch_inplace(1, d, _dev(pha, c2), evk_d)
    assert (_host(pha, d) == ref).all()
    t_ref = o.modup(size_ql, c2)
    d_t = _dev(pha, np.zeros_like(t_ref))
    ctx.modup(1, d_t, _dev(pha, c2))
    assert (_host(pha, d_t) == t_ref).all()
    ctx.close()


@pytest.mark.parametrize("size_p,chain_index", [(2, 1), (2, 2), (1, 1), (3, 2)])
def test_hoisted_rotations(gpu, pha, orc, size_p, chain_index):
    """Several rotations of one ciphertext sharing the modup: bit-exact against the oracle's
    restatement of the hoisted algorithm, and (real keys) decrypting to the rotated message."""
    from oracle.ckks_ref import CkksHarness
    mods = orc.coeff_modulus_create(4096, [50, 40, 40, 40, 40, 40] + [55] * size_p)
    o = orc.Oracle(12, mods, size_p)
    ctx = pha.PhantomContext(12, mods, size_p)
    size_ql = o.size_Q - (chain_index - 1)
    h = CkksHarness(o, seed=7)
    delta = 1 << 36
    msg = [int(v) * delta for v in np.random.default_rng(2).integers(-8, 9, size=4096)]
    ct = h.encrypt(msg, size_ql)
    elts = [orc.galois_elt_from_step(12, s) for s in (1, -3, 64)] + [2 * 4096 - 1]
    keys_h = [h.galois_key(e) for e in elts]
    keys_d = [pha.EvalKey([pha.to_device(k) for k in key]) for key in keys_h]
    outs = ctx.rotate_hoisted(chain_index, pha.to_device(ct), elts, keys_d)
    refs = o.rotate_hoisted(size_ql, ct, elts, keys_h)
    for e, out, ref in zip(elts, outs, refs):
        got = pha.to_host(out)
        assert (got == ref).all()
        err = max(abs(a - b) for a, b in zip(h.decrypt(got, size_ql), h.galois_coeffs(msg, e)))
        assert err < delta * 1e-4
    with pytest.raises(pha.PhantomError):
        ctx.rotate_hoisted(chain_index, pha.to_device(ct), [4], keys_d[:1])   # even Galois element
    ctx.close()


@pytest.mark.parametrize("size_p,chain_index,batch", [(2, 1, 3), (2, 3, 5), (3, 1, 4), (3, 5, 4), (3, 3, 2), (1, 2, 3)])
def test_batched_multiply_relin_rescale_equals_single_calls(gpu, pha, orc, size_p, chain_index, batch):
    """pha_multiply_relin_rescale_batched = `batch` x pha_multiply_relin_rescale, bit for bit (and so the oracle's
    restatement): uniform digits, a single short digit, and the per-product fallback; products, a shared right
    operand and a square in one batch"""
    import torch
    mods = orc.coeff_modulus_create(4096, [45] * 6 + [55] * size_p)
    ctx, o = pha.PhantomContext(12, mods, size_p), orc.Oracle(12, mods, size_p)
    size_ql = ctx.size_Ql(chain_index)
    tq = list(range(size_ql))
    _, evk_d = _uniform_evk(pha, o, 720)
    a = [_dev(pha, np.stack([o.uniform_poly(tq, 730 + 2 * b), o.uniform_poly(tq, 731 + 2 * b)])) for b in range(batch)]
    shared = _dev(pha, np.stack([o.uniform_poly(tq, 760), o.uniform_poly(tq, 761)]))
    right = [shared] * (batch - 1) + [a[-1]]                      # the last one is a square
    for times2 in (False, True):
        out = ctx.multiply_relin_rescale_batched(chain_index, a, right, evk_d, times2)
        for b in range(batch):
            assert torch.equal(out[b], ctx.multiply_relin_rescale(chain_index, a[b], right[b], evk_d, times2))
    ctx.close()
