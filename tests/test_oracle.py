"""Pins of the CPU oracle (oracle/oracle.c): known-answer anchors recorded from the reference's
own host generators and mathematical definitions of every hot-path function.
The reference ships no integer-level vectors, so these are the oracle's pins."""
import numpy as np
import pytest

N16 = 65536
ANCHOR_PRIMES = [1152921504597016577, 1152921504598720513, 1152921504606584833, 1125899903827969]


def test_anchor_primes(orc):
    got = orc.coeff_modulus_create(N16, [60, 60, 60, 50])
    assert [int(x) for x in got] == ANCHOR_PRIMES


def test_anchor_tables(orc):
    o = orc.Oracle(16, np.array(ANCHOR_PRIMES[:1], dtype=np.uint64), 0)
    tw, itw = o.table(0, 0), o.table(2, 0)
    assert int(tw[N16 // 2]) == 17749908910371  # psi = tw[brev(1)]
    assert [int(tw[i]) for i in (1, 2, 3)] == [192357766831624088, 436148905943818200, 590819551163260870]
    assert int(itw[1]) == 358455447691375399
    assert int(o.table(4)[0]) == 1152903912410972311
    ratio = (orc._u64 * 2)()
    orc.lib().orc_barrett_ratio(ANCHOR_PRIMES[0], ratio)
    assert (int(ratio[0]), int(ratio[1])) == (2516582144, 16)
    # Shoup companions are floor(w * 2^64 / q)
    q = ANCHOR_PRIMES[0]
    for i in (1, 2, 3, 77, N16 - 1):
        assert int(o.table(1, 0)[i]) == (int(tw[i]) << 64) // q


def test_psi_is_minimal_primitive_root(orc):
    q = ANCHOR_PRIMES[3]
    o = orc.Oracle(16, np.array([q], dtype=np.uint64), 0)
    psi = int(o.table(0, 0)[N16 // 2])
    assert pow(psi, N16, q) == q - 1
    # all primitive 2N-th roots are odd powers of psi; psi must be the smallest
    cur, sq = psi, psi * psi % q
    for _ in range(N16):
        assert cur >= psi
        cur = cur * sq % q


@pytest.mark.parametrize("log_n", [4, 6, 8])
def test_forward_matches_definition(orc, log_n):
    n = 1 << log_n
    mods = orc.coeff_modulus_create(n, [50, 30])
    o = orc.Oracle(log_n, mods, 0)
    for limb in range(2):
        x = o.uniform_poly([limb], 7 + limb)[0]
        assert (o.ntt_fwd(x, limb) == o.ntt_fwd_naive(x, limb)).all()
        assert (o.ntt_inv(o.ntt_fwd(x, limb), limb) == x).all()


def test_forward_accepts_lazy_inputs(orc):
    # inputs in [0,4q) give the same canonical output
    mods = orc.coeff_modulus_create(4096, [40])
    o = orc.Oracle(12, mods, 0)
    q = int(mods[0])
    x = o.uniform_poly([0], 3)[0]
    assert (o.ntt_fwd(x + np.uint64(3 * q), 0) == o.ntt_fwd(x, 0)).all()


def test_config1_roundtrip_n65536(orc):
    """BASELINE.json configs[0]: forward+inverse round trip, N=2^16, one 60-bit limb."""
    o = orc.Oracle(16, np.array(ANCHOR_PRIMES[:1], dtype=np.uint64), 0)
    x = orc.fill_uniform(N16, ANCHOR_PRIMES[0], 1)
    y = o.ntt_fwd(x, 0)
    assert (y < np.uint64(ANCHOR_PRIMES[0])).all()
    assert (o.ntt_inv(y, 0) == x).all()


def _negacyclic(a, b, q):
    n = len(a)
    out = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            v = int(a[i]) * int(b[j])
            if k >= n:
                out[k - n] = (out[k - n] - v) % q
            else:
                out[k] = (out[k] + v) % q
    return np.array(out, dtype=np.uint64)


def test_convolution_theorem(orc):
    n = 64
    mods = orc.coeff_modulus_create(n, [45])
    o = orc.Oracle(6, mods, 0)
    q = int(mods[0])
    a, b = o.uniform_poly([0], 11)[0], o.uniform_poly([0], 12)[0]
    fa, fb = o.ntt_fwd(a, 0), o.ntt_fwd(b, 0)
    prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    assert (o.ntt_inv(prod, 0) == _negacyclic(a, b, q)).all()


def test_special_mod_and_exclude_range(orc):
    n = 4096
    mods = orc.coeff_modulus_create(n, [40, 40, 40, 40, 50, 50])
    o = orc.Oracle(12, mods, 2)
    # buffer over Q_l ∪ P with l = 3: limbs 0..2 use tables 0..2, limbs 3,4 use tables 4,5
    tidx = [0, 1, 2, 4, 5]
    buf = o.uniform_poly(tidx, 21)
    ref = np.stack([buf[i] if i in (1, 2) else o.ntt_fwd(buf[i], tidx[i]) for i in range(5)])
    got = buf.copy()
    o.nwt_forward_inplace_special(got, 5, 0, 6, 2, 1, 3)
    assert (got == ref).all()
    back = got.copy()
    o.nwt_backward_inplace_special(back, 2, 3, 6, 2)
    assert (back[3:] == buf[3:]).all() and (back[:3] == got[:3]).all()


@pytest.mark.parametrize("step", [1, -1, 3, 0])
def test_galois_permutation_is_the_automorphism(orc, step):
    log_n = 8
    n = 1 << log_n
    mods = orc.coeff_modulus_create(n, [40, 41])
    o = orc.Oracle(log_n, mods, 0)
    elt = orc.galois_elt_from_step(log_n, step)
    if step == 1:
        assert elt == 5
    if step == 0:
        assert elt == 2 * n - 1
    if step == -1:
        assert elt * 5 % (2 * n) == pow(5, n // 2, 2 * n)
    x = o.uniform_poly([0, 1], 5)
    coeff_side = o.apply_galois_coeff(x, elt)
    lhs = np.stack([o.ntt_fwd(coeff_side[l], l) for l in range(2)])
    rhs = o.apply_galois_ntt(np.stack([o.ntt_fwd(x[l], l) for l in range(2)]), elt)
    assert (lhs == rhs).all()
    # the table is a permutation
    assert sorted(orc.galois_table(log_n, elt).tolist()) == list(range(n))


def _crt(residues, moduli):
    M = 1
    for m in moduli:
        M *= int(m)
    x = 0
    for r, m in zip(residues, moduli):
        m = int(m)
        Mi = M // m
        x += int(r) * Mi * pow(Mi, -1, m)
    return x % M, M


@pytest.mark.parametrize("size_p,size_ql", [(1, 3), (2, 5), (2, 4), (3, 3)])
def test_modup_is_uncorrected_fast_base_extension(orc, size_p, size_ql):
    """T_d = [c_i (own limbs, NTT form) ; NTT(x_d + k*Q_d mod p_j)] with 0 <= k < |digit|
    (alpha == 1: k = 0)."""
    log_n, n = 12, 4096
    size_q = 5
    mods = orc.coeff_modulus_create(n, [45] * size_q + [50] * size_p)
    o = orc.Oracle(log_n, mods, size_p)
    c = o.uniform_poly(list(range(size_ql)), 31)
    c_ntt = np.stack([o.ntt_fwd(c[i], i) for i in range(size_ql)])
    t = o.modup(size_ql, c_ntt)
    beta = o.beta(size_ql)
    qlp = o.qlp_moduli(size_ql)
    tidx = list(range(size_ql)) + list(range(size_q, size_q + size_p))
    for d in range(beta):
        s, e = d * size_p, min((d + 1) * size_p, size_ql)
        for j in range(size_ql + size_p):
            if s <= j < e:
                assert (t[d, j] == c_ntt[j]).all()
                continue
            u = o.ntt_inv(t[d, j], tidx[j])
            for k in (0, 1, 77, n - 1):
                x, Qd = _crt([c[i][k] for i in range(s, e)], mods[s:e])
                cands = {(x + kk * Qd) % int(qlp[j]) for kk in range(e - s)}
                assert int(u[k]) in cands


def _keygen(o, rng, size_ql_max):
    """Small valid hybrid switching key from s_from to s (all NTT form, key level);
    layout of src/secretkey.cu:265-310 / src/polymath.cu:323-343."""
    n, size_qp, size_p, size_q = o.n, o.size_QP, o.size_P, o.size_Q
    mods = [int(m) for m in o.moduli]

    def small_poly():
        v = rng.integers(-1, 2, size=n)
        return np.stack([o.ntt_fwd(np.array([(int(x) % m) for x in v], dtype=np.uint64), i) for i, m in enumerate(mods)])

    s, s_from = small_poly(), small_poly()
    dnum = size_q // size_p
    P_mod = [1] * size_qp
    for i in range(size_qp):
        for k in range(size_q, size_qp):
            P_mod[i] = P_mod[i] * mods[k] % mods[i]
    evk = []
    for d in range(dnum):
        a = o.uniform_poly(list(range(size_qp)), 900 + d)
        e = small_poly()
        b = np.zeros_like(a)
        for i in range(size_qp):
            m = mods[i]
            bi = (-(a[i].astype(object) * s[i].astype(object)) + e[i].astype(object)) % m
            if d * size_p <= i < (d + 1) * size_p:
                bi = (bi + P_mod[i] * s_from[i].astype(object)) % m
            b[i] = np.array(bi.tolist(), dtype=np.uint64)
        evk.append(np.ascontiguousarray(np.stack([b, a])))
    return s, s_from, evk


@pytest.mark.parametrize("size_p,drop", [(1, 0), (2, 0), (2, 1), (3, 2)])
def test_keyswitch_decrypts_correctly(orc, size_p, drop):
    """Functional pin: <KS(c2), (1, s)> = c2 * s_from + small error, for alpha = 1 and alpha > 1,
    at the top level and after dropping limbs (ragged last digit)."""
    log_n, n = 12, 4096
    size_q = 6
    mods = orc.coeff_modulus_create(n, [45] * size_q + [55] * size_p)
    o = orc.Oracle(log_n, mods, size_p)
    rng = np.random.default_rng(5)
    s, s_from, evk = _keygen(o, rng, size_q)
    size_ql = size_q - drop
    c2 = o.uniform_poly(list(range(size_ql)), 41)
    ct = np.zeros((2, size_ql, n), dtype=np.uint64)
    o.keyswitch_inplace(size_ql, ct, c2, evk)
    for i in range(size_ql):
        m = int(mods[i])
        lhs = (ct[0, i].astype(object) + ct[1, i].astype(object) * s[i].astype(object)) % m
        rhs = (c2[i].astype(object) * s_from[i].astype(object)) % m
        diff = np.array(((lhs - rhs) % m).tolist(), dtype=np.uint64)
        err = o.ntt_inv(diff, i).astype(object)
        err = np.where(err > m // 2, err - m, err)
        assert max(abs(int(v)) for v in err) < 2 ** 26, "key-switch noise too large: wrong arithmetic"


def test_rescale_is_floor_division_by_q_last(orc):
    log_n, n = 12, 4096
    mods = orc.coeff_modulus_create(n, [45, 40, 40, 50])
    o = orc.Oracle(log_n, mods, 1)
    size_ql = 3
    c = o.uniform_poly(list(range(size_ql)), 51)
    src = np.stack([o.ntt_fwd(c[i], i) for i in range(size_ql)])[None]
    dst = o.divide_and_round_q_last_ntt(size_ql, src, 1)[0]
    out = [o.ntt_inv(dst[i], i) for i in range(size_ql - 1)]
    q_last = int(mods[size_ql - 1])
    for k in (0, 1, 1000, n - 1):
        V, _ = _crt([c[i][k] for i in range(size_ql)], mods[:size_ql])
        expect = (V - int(c[size_ql - 1][k])) // q_last
        assert (V - int(c[size_ql - 1][k])) % q_last == 0
        for i in range(size_ql - 1):
            assert int(out[i][k]) == expect % int(mods[i])


def test_tensor_product(orc):
    log_n, n = 12, 4096
    mods = orc.coeff_modulus_create(n, [60, 50])
    o = orc.Oracle(log_n, mods, 0)
    a = np.stack([o.uniform_poly([0, 1], 61), o.uniform_poly([0, 1], 62)])
    b = np.stack([o.uniform_poly([0, 1], 63), o.uniform_poly([0, 1], 64)])
    r = o.tensor_prod_2x2(a, b)
    for l in range(2):
        m = int(mods[l])
        for k in (0, 5, n - 1):
            a0, a1, b0, b1 = (int(a[0, l, k]), int(a[1, l, k]), int(b[0, l, k]), int(b[1, l, k]))
            assert [int(r[i, l, k]) for i in range(3)] == [a0 * b0 % m, (a0 * b1 + a1 * b0) % m, a1 * b1 % m]


def test_tensor_prod_mxn_is_the_polynomial_product_in_y(orc):
    """(sum_a c_a y^a)(sum_b c'_b y^b) coefficient-wise: checked with Python integers; 2x2 agrees with tensor_prod_2x2"""
    mods = orc.coeff_modulus_create(4096, [50, 45, 60])
    o = orc.Oracle(12, mods, 1)
    lim = [0, 1]
    a = np.stack([o.uniform_poly(lim, 1 + i) for i in range(3)])
    b = np.stack([o.uniform_poly(lim, 11 + i) for i in range(2)])
    res = o.tensor_prod_mxn(a, b)
    assert res.shape == (4, 2, 4096)
    for l, q in enumerate(int(m) for m in mods[:2]):
        for k in (0, 1, 777, 4095):
            for j in range(4):
                want = sum(int(a[x, l, k]) * int(b[j - x, l, k]) for x in range(3) if 0 <= j - x < 2) % q
                assert int(res[j, l, k]) == want
    assert (o.tensor_prod_mxn(a[:2], b) == o.tensor_prod_2x2(a[:2], b)).all()
