"""TEST INFRASTRUCTURE ONLY — scheme-level harness around the CPU oracle.

The reference pins its hot path only functionally: every test in tests/{multi,relin,rotation,
rescale_to_next,apply_galois}.cu checks decrypt(op(encrypt(x))) ~= op(x).  This
module restates, on the CPU, just enough of the reference's key generation / encryption /
decryption to drive the same kind of check against this repository's key switch, with *real*
keys instead of uniform stand-ins:

* secret key: ternary coefficients, NTT form over all key-level limbs
  (src/secretkey.cu:368-440 gen_secretkey)
* encrypt_zero_symmetric: (c0, c1) = (-(a*s + e), a) in NTT form (src/secretkey.cu:207-263)
* key-switching key for a new key s': dnum = size_Q / size_P ciphertexts of zero at key level
  whose b-part receives (P mod q_j) * s'_j on limbs j in [d*alpha, (d+1)*alpha)
  (src/secretkey.cu:265-310, src/polymath.cu:323-343)
* relin key: s' = s^2 (src/secretkey.cu gen_relinkey); Galois key: s' = s(X^g)
  (src/secretkey.cu create_galois_keys, src/galois.cu:42-75 coefficient-form automorphism)
* decrypt: sum_i c_i * s^i, INTT, CRT-compose, centre (src/secretkey.cu decrypt / ckks decode)

Randomness is a seeded numpy generator (the reference uses std::random_device + a stream
cipher; the distributions — uniform, ternary, rounded Gaussian sigma 3.2 — are what matters for
a functional check).  Plaintexts are integer polynomials scaled by Delta (coefficient packing),
which exercises exactly the same ciphertext arithmetic as the canonical-embedding encoder
without the FFT.
"""
import numpy as np

from . import Oracle


def _mulmod_vec(a, b, q):
    """(a * b) mod q for uint64 vectors, exact (Python integers on object arrays)."""
    return np.array((a.astype(object) * b.astype(object)) % int(q), dtype=np.uint64)


def _pointwise(orc, a, b, tidx):
    """limb-wise product of two NTT-form polynomials; limb i is modulo moduli[tidx[i]]"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    for i, t in enumerate(tidx):
        out[i] = _mulmod_vec(a[i], b[i], orc.moduli[t])
    return out


class CkksHarness:
    """Key material and encrypt/decrypt for one key-level context (Oracle)."""

    def __init__(self, orc: Oracle, seed=1, sigma=3.2):
        self.orc = orc
        self.n = orc.n
        self.rng = np.random.default_rng(seed)
        self.sigma = sigma
        self._uniform_seed = seed * 7919 + 11
        self.s = self.rng.integers(-1, 2, size=self.n).astype(np.int64)  # ternary
        self.sk_ntt = self.to_ntt(self.s, range(orc.size_QP))

    # ---- representation helpers ------------------------------------------------------------
    def to_rns(self, coeffs, tidx):
        """signed integer coefficients (int64 array or list of Python ints) -> [len(tidx)][n] residues"""
        obj = np.array([int(c) for c in coeffs], dtype=object)
        out = np.zeros((len(tidx), self.n), dtype=np.uint64)
        for i, t in enumerate(tidx):
            out[i] = np.array(obj % int(self.orc.moduli[t]), dtype=np.uint64)
        return out

    def to_ntt(self, coeffs, tidx):
        tidx = list(tidx)
        r = self.to_rns(coeffs, tidx)
        for i, t in enumerate(tidx):
            r[i] = self.orc.ntt_fwd(r[i], t)
        return r

    def _uniform(self, tidx):
        self._uniform_seed += 1
        return self.orc.uniform_poly(list(tidx), self._uniform_seed)

    def _error(self):
        return np.rint(self.rng.normal(0.0, self.sigma, size=self.n)).astype(np.int64)

    def _addmod(self, a, b, tidx):
        out = np.zeros_like(a)
        for i, t in enumerate(tidx):
            q = self.orc.moduli[t]
            v = a[i] + b[i]
            out[i] = np.where(v >= q, v - q, v)
        return out

    def _negmod(self, a, tidx):
        out = np.zeros_like(a)
        for i, t in enumerate(tidx):
            q = self.orc.moduli[t]
            out[i] = np.where(a[i] == 0, a[i], q - a[i])
        return out

    # ---- encryption ------------------------------------------------------------------------
    def encrypt_zero(self, tidx):
        tidx = list(tidx)
        a = self._uniform(tidx)
        e = self.to_ntt(self._error(), tidx)
        as_ = _pointwise(self.orc, a, self.sk_ntt[tidx], tidx)
        c0 = self._negmod(self._addmod(as_, e, tidx), tidx)
        return np.stack([c0, a])

    def encrypt(self, plain_coeffs, size_ql):
        """Ciphertext [2][size_ql][n] (NTT form) of the integer polynomial plain_coeffs."""
        tidx = list(range(size_ql))
        ct = self.encrypt_zero(tidx)
        ct[0] = self._addmod(ct[0], self.to_ntt(plain_coeffs, tidx), tidx)
        return ct

    def decrypt(self, ct, size_ql=None):
        """ct [k][size_ql][n] in NTT form -> list of centred Python ints (mod Q_l)."""
        ct = np.asarray(ct, dtype=np.uint64)
        size_ql = ct.shape[1] if size_ql is None else size_ql
        tidx = list(range(size_ql))
        acc = ct[0].copy()
        s_pow = self.sk_ntt[tidx].copy()
        for i in range(1, ct.shape[0]):
            acc = self._addmod(acc, _pointwise(self.orc, ct[i], s_pow, tidx), tidx)
            if i + 1 < ct.shape[0]:
                s_pow = _pointwise(self.orc, s_pow, self.sk_ntt[tidx], tidx)
        coeff = np.stack([self.orc.ntt_inv(acc[i], t) for i, t in enumerate(tidx)])
        return self.crt_compose(coeff, tidx)

    def crt_compose(self, coeff, tidx):
        qs = [int(self.orc.moduli[t]) for t in tidx]
        big_q = 1
        for q in qs:
            big_q *= q
        total = np.zeros(self.n, dtype=object)
        for i, q in enumerate(qs):
            hat = big_q // q
            inv = pow(hat % q, -1, q)
            total = total + (coeff[i].astype(object) * inv % q) * hat
        total = total % big_q
        half = big_q // 2
        return [int(v) - big_q if int(v) > half else int(v) for v in total]

    # ---- key-switching keys ----------------------------------------------------------------
    def kswitch_key(self, new_key_ntt):
        """new_key_ntt [size_Q][n] -> list of dnum arrays [2][size_QP][n] (the reference's
        PhantomRelinKey / one PhantomGaloisKey entry)."""
        orc = self.orc
        assert orc.size_Q % orc.size_P == 0, "the reference assumes alpha | size_Q at key level"
        dnum, alpha = orc.size_Q // orc.size_P, orc.size_P
        big_p = orc.level_constants(orc.size_Q)["bigP_mod_q"]
        tidx = list(range(orc.size_QP))
        evk = []
        for d in range(dnum):
            pk = self.encrypt_zero(tidx)
            for j in range(d * alpha, (d + 1) * alpha):
                q = int(orc.moduli[j])
                add = _mulmod_vec(new_key_ntt[j], np.full(self.n, int(big_p[j]), dtype=np.uint64), q)
                v = pk[0][j] + add
                pk[0][j] = np.where(v >= orc.moduli[j], v - orc.moduli[j], v)
            evk.append(np.ascontiguousarray(pk))
        return evk

    def relin_key(self):
        tq = list(range(self.orc.size_Q))
        s2 = _pointwise(self.orc, self.sk_ntt[tq], self.sk_ntt[tq], tq)
        return self.kswitch_key(s2)

    def galois_coeffs(self, coeffs, elt):
        """x(X) -> x(X^elt) on signed coefficients (negacyclic: X^n = -1)."""
        n = self.n
        out = [0] * n
        for i, c in enumerate(coeffs):
            idx = (i * elt) % (2 * n)
            if idx >= n:
                out[idx - n] = -int(c)
            else:
                out[idx] = int(c)
        return out

    def galois_key(self, elt):
        s_g = self.galois_coeffs(self.s, elt)
        return self.kswitch_key(self.to_ntt(s_g, range(self.orc.size_Q)))


def negacyclic_mul(a, b):
    """Exact product of two integer polynomials modulo X^n + 1 (Python ints)."""
    n = len(a)
    a = [int(v) for v in a]
    b = [int(v) for v in b]
    # split into a numpy object convolution: O(n^2) is too slow for n = 4096 in pure Python, so
    # use a float-free Kronecker substitution
    bound = max(1, max(abs(v) for v in a)) * max(1, max(abs(v) for v in b)) * n
    bits = bound.bit_length() + 2
    off = 1 << (bits - 1)

    def pack(v):
        acc = 0
        for c in reversed(v):
            acc = (acc << bits) + c
        return acc

    prod = pack(a) * pack(b)
    # unpack signed digits
    out = [0] * (2 * n)
    mask = (1 << bits) - 1
    for i in range(2 * n - 1):
        d = prod & mask
        if d >= off:
            d -= 1 << bits
        out[i] = d
        prod = (prod - d) >> bits
    return [out[i] - out[i + n] for i in range(n)]


# ---- CKKS canonical-embedding encoder, by its definition (src/ckks.cu:130-283) -------------------
def _slot_matrix(s):
    """A[j][k] = xi^(5^j * k), xi = exp(2*pi*i / (4s)): decode is z = A u."""
    m = 4 * s
    rot = [pow(5, j, m) for j in range(s)]
    e = np.array([[(r * k) % m for k in range(s)] for r in rot], dtype=np.float64)
    return np.exp(2j * np.pi * e / m)


def encode_ref(values, scale, n, sparse_slots):
    """Integer coefficient vector (Python ints, length n) of the plaintext that PhantomCKKSEncoder
    produces for `values` (zero-padded to sparse_slots): coefficients k*gap and (k+s)*gap are the
    rounded real / imaginary parts of scale * u_k, u = A^-1 z = conj(A)^T z / s."""
    s = sparse_slots
    z = np.zeros(s, dtype=np.complex128)
    z[:len(values)] = values
    u = _slot_matrix(s).conj().T @ z / s
    gap = n // (2 * s)
    out = [0] * n
    for k in range(s):
        out[k * gap] = int(np.rint(scale * u[k].real))
        out[(k + s) * gap] = int(np.rint(scale * u[k].imag))
    return out


def decode_ref(coeffs, scale, n, sparse_slots):
    s = sparse_slots
    gap = n // (2 * s)
    u = np.array([complex(coeffs[k * gap] / scale, coeffs[(k + s) * gap] / scale) for k in range(s)])
    return _slot_matrix(s) @ u


# ---- double-hoisted linear transform, restated from the oracle's primitives ------------------------
def linear_transform_double_hoisted_ref(o, size_ql, ct, baby_elts, baby_keys, giant_elts, giant_keys, plains):
    """out = sum_g rot_G( sum_b plains[g][b] * rot_B(ct) ) evaluated as include/phantom_amd.h describes
    pha_linear_transform_double_hoisted: rotated ciphertexts stay over Q_l u P (P * pi(c0) lifted onto the Q_l limbs),
    one moddown per giant step and one at the end.  ct [2][size_ql][n]; plains[g][b] [size_ql + size_P][n] or None;
    keys are lists of beta arrays [2][size_QP][n]; Galois element 1 = no rotation."""
    n, size_p, size_q = o.n, o.size_P, o.size_Q
    qlp = size_ql + size_p
    mods = [int(o.moduli[j]) for j in range(size_ql)] + [int(o.moduli[size_q + i]) for i in range(size_p)]
    big_p = o.level_constants(size_ql)["bigP_mod_q"]

    def lift(poly, elt):
        src = poly if elt == 1 else o.apply_galois_ntt(np.ascontiguousarray(poly), elt)
        out = np.zeros((qlp, n), dtype=np.uint64)
        for j in range(size_ql):
            out[j] = _mulmod_vec(src[j], np.full(n, int(big_p[j]), dtype=np.uint64), mods[j])
        return out

    def addmod(a, b):
        out = np.zeros_like(a)
        for j in range(qlp):
            v = a[..., j, :].astype(object) + b[..., j, :].astype(object)
            out[..., j, :] = np.array(v % mods[j], dtype=np.uint64)
        return out

    def rotate_qp(src, elt, key):
        if elt == 1:
            return np.stack([lift(src[0], 1), lift(src[1], 1)])
        t = o.modup(size_ql, np.ascontiguousarray(src[1]))
        tp = np.stack([o.apply_galois_ntt(np.ascontiguousarray(t[d]), elt) for d in range(t.shape[0])])
        cx = o.key_switch_inner_prod(size_ql, np.ascontiguousarray(tp), key)
        cx[0] = addmod(cx[0], lift(src[0], elt))
        return cx

    def moddown2(x):
        return np.stack([o.moddown_from_ntt(size_ql, np.ascontiguousarray(x[i]).copy()) for i in range(2)])

    rots = [rotate_qp(ct, e, k) for e, k in zip(baby_elts, baby_keys)]
    total = None
    for g, (ge, gk) in enumerate(zip(giant_elts, giant_keys)):
        acc = None
        for b, pt in enumerate(plains[g]):
            if pt is None:
                continue
            term = np.zeros((2, qlp, n), dtype=np.uint64)
            for i in range(2):
                for j in range(qlp):
                    term[i, j] = _mulmod_vec(pt[j], rots[b][i, j], mods[j])
            acc = term if acc is None else addmod(acc, term)
        if acc is None:
            continue
        add = acc if ge == 1 else rotate_qp(moddown2(acc), ge, gk)
        total = add if total is None else addmod(total, add)
    return moddown2(total)


# ---- merged relinearise + rescale, restated from the oracle's primitives ------------------------------
def multiply_relin_rescale_ref(o, size_ql, ct1, ct2, rlk, times2=False):
    """out[2][size_ql-1][n] as include/phantom_amd.h describes pha_multiply_relin_rescale: tensor product, key switch
    of the third poly kept over Q_l u P, the first two polys lifted by P, and ONE division by B = {q_{l-1}} u P with
    an exact, centred base conversion (y_b = x_b * BHat_b^-1 mod b; e = rint(sum_b y_b / b);
    delta_j = sum_b y_b * (BHat_b mod q_j) - e * prod B; out_j = (x_j - delta_j) * (prod B)^-1 mod q_j), INTT / NTT as
    in moddown_from_NTT: the quotient is rounded to nearest."""
    n, size_p, size_q = o.n, o.size_P, o.size_Q
    qlp = size_ql + size_p
    tidx = list(range(size_ql)) + [size_q + i for i in range(size_p)]
    mods = [int(o.moduli[t]) for t in tidx]
    big_p = o.level_constants(size_ql)["bigP_mod_q"]
    d = o.tensor_prod_2x2(np.ascontiguousarray(ct1), np.ascontiguousarray(ct2))
    if times2:      # the doubling is taken on the tensor product, before any division
        for j in range(size_ql):
            d[:, j] = np.array((d[:, j].astype(object) * 2) % mods[j], dtype=np.uint64)
    t = o.modup(size_ql, np.ascontiguousarray(d[2]))
    cx = o.key_switch_inner_prod(size_ql, t, rlk)
    for i in range(2):
        for j in range(size_ql):
            lifted = _mulmod_vec(d[i][j], np.full(n, int(big_p[j]), dtype=np.uint64), mods[j])
            cx[i, j] = np.array((cx[i, j].astype(object) + lifted.astype(object)) % mods[j], dtype=np.uint64)
    base = list(range(size_ql - 1, qlp))                     # buffer limbs of B: q_{l-1}, then P
    bmods = [mods[b] for b in base]
    prod_b = 1
    for m in bmods:
        prod_b *= m
    out = np.zeros((2, size_ql - 1, n), dtype=np.uint64)
    for i in range(2):
        ys = []
        v = np.zeros(n, dtype=np.float64)
        for b, m in zip(base, bmods):
            coeff = o.ntt_inv(np.ascontiguousarray(cx[i, b]), tidx[b])
            hat_inv = pow((prod_b // m) % m, -1, m)
            y = _mulmod_vec(coeff, np.full(n, hat_inv, dtype=np.uint64), m)
            # e = rint(sum_b y_b / b) in FP64, products and sums in this order (bconv_centred_kernel)
            v = v + y.astype(np.float64) * np.float64(1.0 / float(m))
            ys.append(y.astype(object))
        e = np.rint(v).astype(np.uint64).astype(object)
        for j in range(size_ql - 1):
            q = mods[j]
            delta = e * ((q - prod_b % q) % q)
            for y, m in zip(ys, bmods):
                delta = delta + y * ((prod_b // m) % q)
            delta_ntt = o.ntt_fwd(np.array(delta % q, dtype=np.uint64), j)
            inv = pow(prod_b % q, -1, q)
            diff = (cx[i, j].astype(object) + q - delta_ntt.astype(object)) % q
            out[i, j] = np.array(diff * inv % q, dtype=np.uint64)
    return out
