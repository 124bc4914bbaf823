"""Alternative code paths of one operation must store the same words: the base conversion on the matrix cores against
the vector-ALU kernel (which is also the fallback for 16-limb digits, moduli below 2^34 and unaligned buffers), the
single-launch forward transform against the two kernels, derived last-pass twiddles against table reads.  Each pair
runs in one process on contexts created under the corresponding environment switch, against the oracle where it is
cheap and against each other at full size."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(pha, log_n, mods, size_p, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return pha.PhantomContext(log_n, mods, size_p)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _uniform(gpu, mods, rows, n, seed):
    g = gpu.Generator(device="cuda").manual_seed(seed)
    return gpu.stack([gpu.randint(0, int(m), (n,), dtype=gpu.int64, device="cuda", generator=g) for m in rows])


@pytest.mark.parametrize("log_n,q_bits,p_bits,chain_index", [
    (12, [45] * 6, [55] * 2, 1),            # alpha 2: one k-step
    (12, [45] * 6, [55] * 2, 2),            # ragged last digit
    (13, [60] + [50] * 8, [60] * 3, 1),     # alpha 3, 60-bit output moduli
    (12, [50] * 16, [55] * 8, 1),           # alpha 8: two k-steps, constant slot opens the second
    (12, [40] * 30, [50] * 15, 1),          # alpha 15: the largest digit with a table
    (12, [40] * 32, [50] * 16, 1),          # alpha 16: no table, vector ALU on both contexts
    (12, [33] * 4 + [45] * 2, [55] * 2, 1),  # moduli below 2^34: no table
    (16, [60] + [50] * 35, [60] * 12, 1),   # BASELINE configs[2]
])
def test_base_conversion_paths_agree(gpu, pha, orc, log_n, q_bits, p_bits, chain_index):
    n, size_p = 1 << log_n, len(p_bits)
    mods = pha.coeff_modulus_create(n, q_bits + p_bits)
    alu = _ctx(pha, log_n, mods, size_p, PHA_BCONV_MFMA=0)
    mfma = _ctx(pha, log_n, mods, size_p, PHA_BCONV_MFMA=1)
    size_ql = len(q_bits) - (chain_index - 1)
    beta = alu.beta(chain_index)
    c2 = _uniform(gpu, mods, mods[:size_ql], n, 1)
    ct = gpu.stack([_uniform(gpu, mods, mods[:size_ql], n, 2), _uniform(gpu, mods, mods[:size_ql], n, 3)])
    key_rows = list(mods[:len(q_bits)]) + list(mods[len(q_bits):])
    evk = pha.EvalKey([gpu.stack([_uniform(gpu, mods, key_rows, n, 10 + 2 * d), _uniform(gpu, mods, key_rows, n, 11 + 2 * d)])
                       for d in range(beta)])
    outs = []
    for ctx in (alu, mfma):
        t = gpu.full((beta, size_ql + size_p, n), -1, dtype=gpu.int64, device="cuda")
        ctx.modup(chain_index, t, c2.clone())
        x = ct.clone()
        ctx.keyswitch_inplace(chain_index, x, c2.clone(), evk)
        prod = None
        if size_ql > 1 and size_p <= 14:   # the merged rescale takes at most 14 special primes
            prod = ctx.multiply_relin_rescale(chain_index, ct.clone(), gpu.flip(ct, [0]).contiguous(), evk)
        gpu.cuda.synchronize()
        outs.append((t, x, prod))
    assert gpu.equal(outs[0][0], outs[1][0]), "modup differs between the vector-ALU and the matrix-core conversion"
    assert gpu.equal(outs[0][1], outs[1][1]), "key switch differs"
    if outs[0][2] is not None:
        assert gpu.equal(outs[0][2], outs[1][2]), "merged multiply + relinearise + rescale differs"
    if log_n <= 13:   # and both equal the oracle
        o = orc.Oracle(log_n, mods, size_p)
        want = pha.to_host(ct)
        o.keyswitch_inplace(size_ql, want, pha.to_host(c2), [pha.to_host(k) for k in evk.digits])
        assert (pha.to_host(outs[1][1]) == want).all()
    alu.close()
    mfma.close()


def test_unaligned_buffers_take_the_vector_alu_kernel(gpu, pha, orc):
    """pha_modup on a destination that is 8- but not 16-byte aligned: the matrix-core kernel needs 16-byte accesses, the
    call must fall back and still be right."""
    log_n, n, size_p = 12, 4096, 2
    mods = orc.coeff_modulus_create(n, [45] * 4 + [55] * 2)
    ctx, o = pha.PhantomContext(log_n, mods, size_p), orc.Oracle(log_n, mods, size_p)
    c2 = o.uniform_poly([0, 1, 2, 3], 5)
    want = o.modup(4, c2) if hasattr(o, "modup") else None
    backing = gpu.zeros(2 * 6 * n + 1, dtype=gpu.int64, device="cuda")
    dst = backing[1:].view(2, 6, n)
    assert dst.data_ptr() % 16 == 8
    ctx.modup(1, dst, pha.to_device(c2))
    ref = gpu.zeros((2, 6, n), dtype=gpu.int64, device="cuda")
    ctx.modup(1, ref, pha.to_device(c2))
    gpu.cuda.synchronize()
    assert gpu.equal(dst, ref)
    if want is not None:
        assert (pha.to_host(ref) == want).all()
    ctx.close()


@pytest.mark.parametrize("shape", [(2, 36), (1, 36), (5, 20)])
def test_single_launch_transform_equals_two_kernels(gpu, pha, shape):
    polys, limbs = shape
    log_n, n = 16, 1 << 16
    mods = pha.coeff_modulus_create(n, [60] + [50] * (limbs - 1))
    ctx = pha.PhantomContext(log_n, mods, 0)
    x = gpu.stack([_uniform(gpu, mods, mods, n, 20 + p) for p in range(polys)])
    a, b = x.clone(), x.clone()
    ctx.ntt_forward_batched(a, polys, limbs, limbs)
    ctx.ntt_fused_config(1, 0, 1)
    for _ in range(3):   # the control block must come back clean for the next launch
        b.copy_(x)
        ctx.ntt_forward_batched(b, polys, limbs, limbs)
        gpu.cuda.synchronize()
        err, queues = ctx.ntt_fused_status()
        assert err == 0 and queues >= 1
        assert gpu.equal(a, b), "single-launch forward transform differs from the two kernels"
    ctx.ntt_fused_config(0)
    ctx.close()


@pytest.mark.parametrize("log_n", [15, 16, 17])
def test_derived_twiddles_equal_table_twiddles(gpu, pha, log_n):
    n = 1 << log_n
    mods = pha.coeff_modulus_create(n, [60] + [50] * 5 + [45] * 2)
    derived = _ctx(pha, log_n, mods, 0, PHA_GEN_TW=1)
    table = _ctx(pha, log_n, mods, 0, PHA_GEN_TW=0)
    x = _uniform(gpu, mods, mods, n, 7)
    a, b = x.clone(), x.clone()
    derived.nwt_2d_radix8_forward_inplace(a, len(mods), 0)
    table.nwt_2d_radix8_forward_inplace(b, len(mods), 0)
    assert gpu.equal(a, b), "forward transform differs"
    derived.nwt_2d_radix8_backward_inplace(a, len(mods), 0)
    table.nwt_2d_radix8_backward_inplace(b, len(mods), 0)
    gpu.cuda.synchronize()
    assert gpu.equal(a, b) and gpu.equal(a, x), "inverse transform differs"
    derived.close()
    table.close()


@pytest.mark.parametrize("log_n,bits", [(16, [60] * 3 + [50] * 5), (17, [60] * 2 + [50] * 2), (15, [60, 50, 45])])
def test_eight_per_lane_column_pass_equals_default(gpu, pha, log_n, bits):
    """PHA_COL8=1 (csrc/ntt_kernels.cuh fntt_col8_kernel; applies where the column pass has eight stages, a no-op at
    N = 2^15) stores the same words as the default forward transform, on FP64 and on integer limbs."""
    n = 1 << log_n
    mods = pha.coeff_modulus_create(n, bits)
    col8 = _ctx(pha, log_n, mods, 0, PHA_COL8=1)
    base = _ctx(pha, log_n, mods, 0, PHA_COL8=0)
    x = _uniform(gpu, mods, mods, n, 11)
    a, b = x.clone(), x.clone()
    col8.nwt_2d_radix8_forward_inplace(a, len(mods), 0)
    base.nwt_2d_radix8_forward_inplace(b, len(mods), 0)
    assert gpu.equal(a, b), "forward transform differs"
    a2, b2 = gpu.stack([x, a]), gpu.stack([x, a])                 # batched launch: two polys
    col8.ntt_forward_batched(a2, 2, len(mods), len(mods), 0)
    base.ntt_forward_batched(b2, 2, len(mods), len(mods), 0)
    assert gpu.equal(a2, b2) and gpu.equal(a2[0], a)
    col8.nwt_2d_radix8_backward_inplace(a, len(mods), 0)
    gpu.cuda.synchronize()
    assert gpu.equal(a, x)
    col8.close()
    base.close()


@pytest.mark.parametrize("log_n,q_bits,size_p,p_bits", [(16, [60] + [50] * 7, 1, 60), (15, [55, 50, 45, 60], 1, 58), (16, [60] * 3 + [50] * 3, 2, 60),
                                                        (13, [54, 40, 40, 40, 50], 1, 55), (12, [50, 36, 36], 1, 50), (14, [60, 45, 45, 45], 3, 60)])
def test_broadcast_inside_the_column_pass_equals_separate_pass(gpu, pha, log_n, q_bits, size_p, p_bits):
    """rescale_to_next and the alpha = 1 moddown broadcast ONE coefficient-form limb over the remaining limbs
    (src/rns.cu:1125-1136, src/rns_bconv.cu:691-707).  By default the forward column kernel reduces it on load
    (fntt_col_kernel<.., BCAST>); PHA_BCAST_FUSED=0 writes the residues in a pass of their own first.  Same words, on
    integer and FP64 limbs, for rescale (any alpha) and for a key switch with one special prime."""
    n = 1 << log_n
    mods = pha.coeff_modulus_create(n, q_bits + [p_bits] * size_p)
    fused = _ctx(pha, log_n, mods, size_p, PHA_BCAST_FUSED=1)
    plain = _ctx(pha, log_n, mods, size_p, PHA_BCAST_FUSED=0)
    size_q = len(q_bits)
    src = gpu.stack([_uniform(gpu, mods, mods[:size_q], n, 21), _uniform(gpu, mods, mods[:size_q], n, 22)])
    a, b = gpu.empty((2, size_q - 1, n), dtype=gpu.int64, device="cuda"), gpu.empty((2, size_q - 1, n), dtype=gpu.int64, device="cuda")
    fused.rescale_to_next(1, src, 2, a)
    plain.rescale_to_next(1, src, 2, b)
    assert gpu.equal(a, b), "rescale differs"
    if size_p == 1:
        evk_t = [gpu.stack([_uniform(gpu, mods, mods, n, 30 + 2 * d), _uniform(gpu, mods, mods, n, 31 + 2 * d)]) for d in range(size_q)]
        evk = pha.EvalKey(evk_t)
        ct = gpu.stack([_uniform(gpu, mods, mods[:size_q], n, 23), _uniform(gpu, mods, mods[:size_q], n, 24)])
        c2 = _uniform(gpu, mods, mods[:size_q], n, 25)
        ca, cb = ct.clone(), ct.clone()
        fused.keyswitch_inplace(1, ca, c2, evk)
        plain.keyswitch_inplace(1, cb, c2, evk)
        gpu.cuda.synchronize()
        assert gpu.equal(ca, cb), "key switch with one special prime differs"
    if size_q > 2:      # one level down: a shorter chain, the same special prime(s)
        lo = src[:, :size_q - 1].contiguous()
        a2, b2 = gpu.empty((2, size_q - 2, n), dtype=gpu.int64, device="cuda"), gpu.empty((2, size_q - 2, n), dtype=gpu.int64, device="cuda")
        fused.rescale_to_next(2, lo, 2, a2)
        plain.rescale_to_next(2, lo, 2, b2)
        assert gpu.equal(a2, b2), "rescale one level down differs"
        if size_p == 1:
            ct2 = gpu.stack([_uniform(gpu, mods, mods[:size_q - 1], n, 26), _uniform(gpu, mods, mods[:size_q - 1], n, 27)])
            c22 = _uniform(gpu, mods, mods[:size_q - 1], n, 28)
            ca, cb = ct2.clone(), ct2.clone()
            fused.keyswitch_inplace(2, ca, c22, evk)
            plain.keyswitch_inplace(2, cb, c22, evk)
            gpu.cuda.synchronize()
            assert gpu.equal(ca, cb), "key switch one level down differs"
    fused.close()
    plain.close()
