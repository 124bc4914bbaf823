"""Full-size (BASELINE.json configs) checks on the GPU: direct oracle comparison where the oracle
finishes in seconds, plus size-independent properties (round trip, linearity, convolution)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 65536


@pytest.fixture(scope="module")
def cfg3(gpu, pha, orc):
    """Config 3 parameters: L = 35 -> 36 Q limbs, dnum = 3 -> 12 special primes, beta = 3."""
    mods = orc.coeff_modulus_create(N, [60] + [50] * 35 + [60] * 12)
    return pha.PhantomContext(16, mods, 12), orc.Oracle(16, mods, 12)


def test_ntt_properties_full_size(gpu, pha, cfg3):
    import torch
    ctx, o = cfg3
    L = 48
    mods = torch.from_numpy(o.moduli.view(np.int64)).cuda().view(L, 1)
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randint(0, 1 << 49, (L, N), dtype=torch.int64, device="cuda", generator=g) % mods
    b = torch.randint(0, 1 << 49, (L, N), dtype=torch.int64, device="cuda", generator=g) % mods
    fa, fb, fs = a.clone(), b.clone(), (a + b) % mods
    for t in (fa, fb, fs):
        ctx.nwt_2d_radix8_forward_inplace(t, L, 0)
    assert torch.equal((fa + fb) % mods, fs), "NTT is not linear"
    assert bool((fa >= 0).all()) and bool((fa < mods).all()), "outputs not canonical"
    back = fa.clone()
    ctx.nwt_2d_radix8_backward_inplace(back, L, 0)
    assert torch.equal(back, a), "INTT(NTT(x)) != x"
    # convolution theorem on one limb with a sparse b: b = X^s  =>  a*b is a negacyclic shift of a
    s = 12345
    mono = torch.zeros((L, N), dtype=torch.int64, device="cuda")
    mono[:, s] = 1
    fm = mono.clone()
    ctx.nwt_2d_radix8_forward_inplace(fm, L, 0)
    prod = torch.zeros((3, L, N), dtype=torch.int64, device="cuda")
    ctx.tensor_prod_2x2(torch.stack([fa, fa]), torch.stack([fm, fm]), L, prod)
    c = prod[0].clone()
    ctx.nwt_2d_radix8_backward_inplace(c, L, 0)
    expect = torch.cat([(mods - a[:, N - s:]) % mods, a[:, :N - s]], dim=1)
    assert torch.equal(c, expect), "NTT(a) * NTT(X^s) is not the negacyclic shift"


def test_config3_keyswitch_vs_oracle(gpu, pha, orc, cfg3):
    ctx, o = cfg3
    size_ql = 36
    evk_h = [np.ascontiguousarray(np.stack([o.uniform_poly(list(range(48)), 500 + 10 * d),
                                             o.uniform_poly(list(range(48)), 501 + 10 * d)])) for d in range(3)]
    evk = pha.EvalKey([pha.to_device(h) for h in evk_h])
    ct1 = np.stack([o.uniform_poly(list(range(size_ql)), 3), o.uniform_poly(list(range(size_ql)), 13)])
    ct2 = np.stack([o.uniform_poly(list(range(size_ql)), 4), o.uniform_poly(list(range(size_ql)), 14)])
    # HMult + relinearize (config 3)
    tp = o.tensor_prod_2x2(ct1, ct2)
    ref = tp[:2].copy()
    o.keyswitch_inplace(size_ql, ref, tp[2], evk_h)
    d1 = pha.to_device(np.concatenate([ct1, np.zeros_like(ct1[:1])]))
    ctx.multiply_and_relin_inplace(1, d1, pha.to_device(ct2), evk)
    got = pha.to_host(d1)[:2]
    assert (got == ref).all()
    # rescale of the result (rescale_to_next)
    rs_ref = o.divide_and_round_q_last_ntt(size_ql, ref, 2)
    d_rs = pha.to_device(np.zeros_like(rs_ref))
    ctx.rescale_to_next(1, d1, 2, d_rs)
    assert (pha.to_host(d_rs) == rs_ref).all()
    # rotate by one step (config 4 unit): Galois element 5
    rot_ref = np.stack([o.apply_galois_ntt(ct1[0], 5), np.zeros_like(ct1[1])])
    o.keyswitch_inplace(size_ql, rot_ref, o.apply_galois_ntt(ct1[1], 5), evk_h)
    d_rot = pha.to_device(ct1)
    ctx.apply_galois_inplace(1, d_rot, 5, evk)
    assert (pha.to_host(d_rot) == rot_ref).all()


def test_config4_rotate_64_ciphertexts_full_size(gpu, pha, cfg3):
    """BASELINE configs[3] at its stated size: rotation of 64 independent ciphertexts (N = 2^16, 36 + 12 limbs), in batches
    of 16 as bench.py --config 4 issues them.  Property: every ciphertext of a batch equals the single-ciphertext
    pha_apply_galois_inplace of the same input bit for bit (that entry point is compared with the oracle at this size in
    test_config3_keyswitch_vs_oracle).  The keys are uniform stand-ins, which is enough for the arithmetic identity of the
    two code paths."""
    import torch
    ctx, o = cfg3
    size_ql, total, batch = 36, 64, 16
    g = torch.Generator(device="cuda").manual_seed(11)
    lo = int(o.moduli.min())
    evk = pha.EvalKey([torch.randint(0, lo, (2, 48, N), dtype=torch.int64, device="cuda", generator=g) for _ in range(3)])
    elt = pha.galois_elt_from_step(16, 1)
    for b0 in range(0, total, batch):
        cts = torch.randint(0, lo, (batch, 2, size_ql, N), dtype=torch.int64, device="cuda", generator=g)
        out = ctx.apply_galois_batched(1, cts, elt, evk)
        assert out.shape == cts.shape
        for i in range(batch):
            one = cts[i].clone()
            ctx.apply_galois_inplace(1, one, elt, evk)
            assert torch.equal(out[i], one), f"ciphertext {b0 + i} of the batched rotation differs from the single call"
