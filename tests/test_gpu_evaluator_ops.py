"""Element-wise evaluator arithmetic through the C ABI (add / sub / negate / add_plain / sub_plain /
multiply_plain / mod_switch_to_next, src/evaluate.cu:20-105,442-560,618-700) against exact integer
arithmetic on the CPU, and their meaning on real ciphertexts: decrypt(op(ct)) = op(message)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N, LOG_N = 4096, 12


def _mod(a, mods):
    return np.array(a % np.array([int(m) for m in mods], dtype=object)[:, None], dtype=np.uint64)


def test_elementwise_ops_exact(gpu, pha, orc):
    mods = orc.coeff_modulus_create(N, [60, 50, 45, 61 - 1, 55])
    o = orc.Oracle(LOG_N, mods, 1)
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    limbs = 4
    a = np.stack([o.uniform_poly(list(range(limbs)), 1), o.uniform_poly(list(range(limbs)), 2)])
    b = np.stack([o.uniform_poly(list(range(limbs)), 3), o.uniform_poly(list(range(limbs)), 4)])
    a[0, 0, :4] = [0, 1, int(mods[0]) - 1, 0]
    b[0, 0, :4] = [0, int(mods[0]) - 1, int(mods[0]) - 1, 5]
    pl = o.uniform_poly(list(range(limbs)), 5)
    ao, bo, po = a.astype(object), b.astype(object), pl.astype(object)
    q = mods[:limbs]

    def run(fn, *args):
        d = pha.to_device(a)
        fn(d, *args)
        return pha.to_host(d)

    assert (run(ctx.add_inplace, pha.to_device(b)) == np.stack([_mod(ao[i] + bo[i], q) for i in range(2)])).all()
    assert (run(ctx.sub_inplace, pha.to_device(b)) == np.stack([_mod(ao[i] - bo[i], q) for i in range(2)])).all()
    assert (run(ctx.negate_inplace) == np.stack([_mod(-ao[i], q) for i in range(2)])).all()
    got = run(ctx.add_plain_inplace, pha.to_device(pl))
    assert (got[0] == _mod(ao[0] + po, q)).all() and (got[1] == a[1]).all()
    got = run(ctx.sub_plain_inplace, pha.to_device(pl))
    assert (got[0] == _mod(ao[0] - po, q)).all() and (got[1] == a[1]).all()
    assert (run(ctx.multiply_plain_inplace, pha.to_device(pl)) == np.stack([_mod(ao[i] * po, q) for i in range(2)])).all()
    nxt = pha.to_host(ctx.mod_switch_to_next(1, pha.to_device(a)))
    assert nxt.shape == (2, 3, N) and (nxt == a[:, :3]).all()
    with pytest.raises(pha.PhantomError):
        ctx.mod_switch_to_next(4, pha.to_device(a[:, :1]))       # end of chain
    ctx.close()


def test_ops_on_real_ciphertexts(gpu, pha, orc):
    from oracle.ckks_ref import CkksHarness, negacyclic_mul
    mods = orc.coeff_modulus_create(N, [50, 40, 40, 50])
    o = orc.Oracle(LOG_N, mods, 1)
    ctx = pha.PhantomContext(LOG_N, mods, 1)
    sk = ctx.gen_secretkey(3)
    h = CkksHarness(o, seed=1)
    h.sk_ntt = pha.to_host(sk)
    rng = np.random.default_rng(8)
    delta = 1 << 30
    m1 = [int(v) * delta for v in rng.integers(-50, 51, size=N)]
    m2 = [int(v) * delta for v in rng.integers(-50, 51, size=N)]
    small = [int(v) for v in rng.integers(-3, 4, size=N)]
    size_ql = o.size_Q
    ct1 = ctx.encrypt_symmetric(sk, pha.to_device(h.to_ntt(m1, range(size_ql))), 1, 11, 12)
    ct2 = ctx.encrypt_symmetric(sk, pha.to_device(h.to_ntt(m2, range(size_ql))), 1, 13, 14)
    pl = pha.to_device(h.to_ntt(small, range(size_ql)))

    def dec(ct, limbs=size_ql):
        return h.decrypt(pha.to_host(ct), limbs)

    def err(got, want):
        return max(abs(a - b) for a, b in zip(got, want))

    s = ct1.clone(); ctx.add_inplace(s, ct2)
    assert err(dec(s), [a + b for a, b in zip(m1, m2)]) <= 42
    s = ct1.clone(); ctx.sub_inplace(s, ct2)
    assert err(dec(s), [a - b for a, b in zip(m1, m2)]) <= 42
    s = ct1.clone(); ctx.negate_inplace(s)
    assert err(dec(s), [-a for a in m1]) <= 21
    s = ct1.clone(); ctx.add_plain_inplace(s, pl)
    assert err(dec(s), [a + b for a, b in zip(m1, small)]) <= 21
    s = ct1.clone(); ctx.multiply_plain_inplace(s, pl)
    assert err(dec(s), negacyclic_mul(m1, small)) <= 21 * 3 * N
    s = ctx.mod_switch_to_next(1, ct1)
    assert err(dec(s, size_ql - 1), m1) <= 21
    ctx.close()
