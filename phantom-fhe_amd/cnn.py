"""ResNet inference on CKKS ciphertexts: multiplexed parallel convolutions, batch norm, approximate ReLU,
downsampling, average pooling and the fully connected layer, with bootstrapping between the blocks.

Host orchestration of the reference's CNN driver (`src/cnn_phantom.cu`, `examples/cnn.cu`: the multiplexed packing
of Lee et al., ICML 2022) on the evaluator interface of ckks.py — every ciphertext operation below is an evaluator
primitive, i.e. one or a few C-ABI calls on the GPU back end; nothing here computes on polynomials.

A tensor of c channels of h x w pixels is packed with gap k: slot  k^2hw*u + kw*v1 + v2  of a block holds channel
k^2 u + k (v1 mod k) + (v2 mod k) at pixel (v1 div k, v2 div k); t = ceil(c / k^2) blocks, and the n/p-slot vector
is repeated p times, so that the ciphertext is a sparsely packed one of n/p slots (what bootstrapping uses).  The
index formulas follow the reference (file:line cited per function); what differs is how the work is issued:

* model constants (shifted weight vectors, selection masks, batch-norm vectors) are encoded once per model and kept
  on the device (`plain cache`): the reference re-encodes them inside every layer call;
* rotations of one ciphertext share their modup (`rotate_many`: the 8 neighbour shifts of a convolution, the p
  collecting rotations, the k*t selecting rotations of a downsampling, the q + r - 1 diagonals of the fully
  connected layer), a rotation by an arbitrary step is ONE key switch with its own Galois key (the reference
  composes power-of-two steps, `rotate_vector_inplace` with keys for 2^i only), and the p output copies are made
  by log2(p) doublings instead of p - 1 rotations;
* batch norm adds a plaintext (the reference encrypts the vector first, `cnn_phantom.cu:614-623`);
* the sign polynomials are evaluated in the Chebyshev basis with Paterson-Stockmeyer at the optimal depth
  (4 + 4 + 5 levels for degrees 15, 15, 27, like the reference's `oddbaby` trees over the power basis) and are
  generated here by a Remez exchange in double precision (the reference reads `result/d13.txt`, produced by an NTL
  program that is not part of the repository).
"""
import math
import os

import numpy as np

from .bootstrap import ChebyshevEvaluator


PARALLEL_CONV = os.environ.get("PHA_CONV_PARALLEL", "1") != "0"


class TensorCipher:
    """A packed tensor: the ciphertext and (k, h, w, c, t, p) of `TensorCipher`, src/cnn_phantom.cu:3-60."""
    __slots__ = ("logn", "k", "h", "w", "c", "t", "p", "ct")

    def __init__(self, logn, k, h, w, c, t, p, ct):
        self.logn, self.k, self.h, self.w, self.c, self.t, self.p, self.ct = logn, k, h, w, c, t, p, ct

    def parms(self):
        return dict(k=self.k, h=self.h, w=self.w, c=self.c, t=self.t, p=self.p)


def _log2(v):
    """log2 of a power of two, -1 otherwise (the reference's log2_long)"""
    return v.bit_length() - 1 if v > 0 and v & (v - 1) == 0 else -1


def _coords(n, k, h, w, t, p):
    """per slot of a packed vector: copy, block, row, column in the k*h x k*w grid, and whether it holds data"""
    j = np.arange(n)
    pos = j % (n // p)
    blk = pos // (k * k * h * w)
    v1 = (pos % (k * k * h * w)) // (k * w)
    v2 = pos % (k * w)
    return j // (n // p), blk, v1, v2, pos < k * k * h * w * t


def pack_tensor(a, logn, k, p):
    """[c][h][w] array -> n-slot vector in the multiplexed packing with gap k, repeated p times"""
    c, h, w = a.shape
    n, t = 1 << logn, -(-c // (k * k))
    if k * k * h * w * t * p > n or n % p:
        raise ValueError("k^2 h w t p is larger than n")
    _, blk, v1, v2, valid = _coords(n, k, h, w, t, p)
    ch = k * k * blk + k * (v1 % k) + v2 % k
    ok = valid & (ch < c)
    return np.where(ok, a[np.minimum(ch, c - 1), v1 // k, v2 // k], 0.0)


def unpack_tensor(vec, k, h, w, c):
    """first copy of a packed vector -> [c][h][w]"""
    ch = np.arange(c)[:, None, None]
    row = np.arange(h)[None, :, None]
    col = np.arange(w)[None, None, :]
    idx = k * k * h * w * (ch // (k * k)) + k * w * (k * row + (ch % (k * k)) // k) + k * col + ch % k
    return np.asarray(vec)[idx]


class PlainCache:
    """Encoded model constants by (layer, tag): filled on the first inference, reused afterwards."""

    def __init__(self, ev):
        self.ev, self.store = ev, {}

    def get(self, key, make, scale, level, slots, lt=False):
        """lt: in the form `ev.bsgs` consumes (over Q_l u P on the GPU back end)"""
        full = (key, level, float(scale), lt)
        if full in self.store:
            return self.store[full]
        vec = make()                                     # None: an all-zero constant, nothing to multiply by
        if vec is not None:
            enc = self.ev.encode_lt if lt else self.ev.encode
            vec = enc(np.asarray(vec, dtype=np.complex128), scale, level, slots)
        self.store[full] = vec
        return vec


def _rot_steps(steps, n):
    return [int(s) % n for s in steps]


def _double_copies(ev, ct, n, po):
    """po copies at distance n/po: the reference adds po rotations by -u (n/po) (cnn_phantom.cu:560-571)"""
    x = 1
    while x < po:
        ct = ev.add(ct, ev.rotate(ct, (-(n // po) * x) % n))
        x <<= 1
    return ct


# =====================================================================================================================
# layers
# =====================================================================================================================
def multiplexed_parallel_convolution(ev, cache, name, x, co, st, fh, fw, data, running_var, constant_weight, epsilon, end=False):
    """`multiplexed_parallel_convolution_phantom`, src/cnn_phantom.cu:306-574: fh x fw convolution with stride st
    (zero padding (f-1)/2), every output channel multiplied by constant_weight / sqrt(running_var + epsilon).  Two
    levels.  `data` is the [co][ci][fh][fw] weight tensor, flat."""
    ki, hi, wi, ci, ti, pi, logn = x.k, x.h, x.w, x.c, x.t, x.p, x.logn
    n = 1 << logn
    if st not in (1, 2):
        raise ValueError("supported st is only 1 or 2")
    data = np.asarray(data, dtype=np.float64)
    if data.size != fh * fw * ci * co:
        raise ValueError("the size of data vector is not ker x ker x h x h")
    if _log2(ki) == -1:
        raise ValueError("ki is not power of two")
    running_var, constant_weight = np.asarray(running_var, dtype=np.float64), np.asarray(constant_weight, dtype=np.float64)
    if running_var.size != co or constant_weight.size != co:
        raise ValueError("the size of running_var or weight is not correct")
    if np.any(np.abs(running_var) < 1e-16):
        raise ValueError("the size of running_var is too small. nearly zero.")
    if fh % 2 == 0 or fw % 2 == 0:
        raise ValueError("fh and fw should be odd")
    if st == 1:
        ho, wo, ko = hi, wi, ki
    else:
        if hi % 2 or wi % 2:
            raise ValueError("hi or wi is not even")
        ho, wo, ko = hi // 2, wi // 2, 2 * ki
    to = -(-co // (ko * ko))
    po = 1 << int(math.floor(math.log2(n / (ko * ko * ho * wo * to))))
    q = -(-co // pi)
    if n % pi or n % po:
        raise ValueError("n is not divisible by pi or po")
    if ki * ki * hi * wi * ti * pi > n or ko * ko * ho * wo * to * po > n:
        raise ValueError("k^2 h w t p is larger than n")

    W = data.reshape(co, ci, fh, fw)
    copy, blk, v1, v2, valid = _coords(n, ki, hi, wi, ti, pi)
    in_ch = ki * ki * blk + ki * (v1 % ki) + v2 % ki
    row, col = v1 // ki, v2 // ki
    ct = x.ct
    level = ct.level
    taps = [(i1, i2) for i1 in range(fh) for i2 in range(fw)]
    rots = ev.rotate_many(ct, _rot_steps([ki * ki * wi * (i1 - (fh - 1) // 2) + ki * (i2 - (fw - 1) // 2) for i1, i2 in taps], n))
    d, c = _log2(ki), _log2(ti)
    # selection masks of the output packing, carrying the batch-norm factor (cnn_phantom.cu:403-434)
    _, oblk, o1, o2, ovalid = _coords(n, ko, ho, wo, to, 1)
    o_ch = ko * ko * oblk + ko * (o1 % ko) + o2 % ko
    bn_scale = constant_weight / np.sqrt(running_var + epsilon)
    def weighted_sums(e, first, last):
        out = []
        for i9 in range(first, last):
            out_ch = copy + pi * i9
            acc = None
            for (i1, i2), r in zip(taps, rots):
                def make(out_ch=out_ch, i1=i1, i2=i2):
                    rr, cc = row + i1 - (fh - 1) // 2, col + i2 - (fw - 1) // 2
                    ok = valid & (out_ch < co) & (in_ch < ci) & (rr >= 0) & (rr < hi) & (cc >= 0) & (cc < wi)
                    return np.where(ok, W[np.minimum(out_ch, co - 1), np.minimum(in_ch, ci - 1), i1, i2], 0.0) if ok.any() else None
                pt = cache.get((name, "w", i9, i1, i2), make, ev.q_last(level), level, n)
                if pt is not None:
                    acc = e.mul_plain_add(acc, r, pt)
            out.append(e.rescale(acc))
        return out
    # once the constants are encoded, the two halves of the partial sums (and of the collection below) run side by
    # side on the evaluator's sibling (context, stream) pair: kernels on 2-5 limbs leave most of the GPU idle
    two = PARALLEL_CONV and q >= 2 and cache.store.get((name, "warm"), False)
    def fan_out(e, fn, join, first, last, depth):
        """fn over [first, last) split over up to 2^depth evaluators (nested run_pair)"""
        if depth == 0 or last - first < 2:
            return fn(e, first, last)
        mid = (first + last) // 2
        a, b = e.run_pair(lambda e1: fan_out(e1, fn, join, first, mid, depth - 1), lambda e2: fan_out(e2, fn, join, mid, last, depth - 1))
        return join(e, a, b)
    parts = fan_out(ev, weighted_sums, lambda e, a, b: a + b, 0, q, 2 if two else 0)
    # summation over the input channels (cnn_phantom.cu:499-537): the q partial sums rotate by the same steps, so
    # they move as one batch
    var = ev.batch(parts)
    sum_steps = [1 << s for s in range(d)] + [(1 << s) * ki * wi for s in range(d)]
    if c != -1:
        sum_steps += [(1 << s) * ki * ki * hi * wi for s in range(c)]
    for step in sum_steps:
        var = ev.add(var, ev.rotate_batch(var, step % n))
    if c == -1:
        acc = var
        for s in range(1, ti):
            acc = ev.add(acc, ev.rotate_batch(var, (ki * ki * hi * wi * s) % n))
        var = acc
    # collect the valid values into the output packing (cnn_phantom.cu:539-554): per partial sum one hoisted sum of
    # masked rotations (one modup, one moddown)
    vs = ev.unbatch(var)

    def collect(e, first, last):
        total = None
        for i9 in range(first, last):
            js = [pi * i9 + i8 for i8 in range(pi) if pi * i9 + i8 < co]
            steps = _rot_steps([(n // pi) * (j4 % pi) - j4 % ko - (j4 // (ko * ko)) * ko * ko * ho * wo - ((j4 % (ko * ko)) // ko) * ko * wo for j4 in js], n)
            plains = []
            for j4 in js:
                def make_sel(j4=j4):
                    return np.where(ovalid & (o_ch == j4), bn_scale[j4], 0.0)
                plains.append(cache.get((name, "sel", j4), make_sel, ev.q_last(level + 1), level + 1, n, lt=True))
            part = e.bsgs(vs[i9], steps, [0], [plains])
            total = part if total is None else e.add(total, part)
        return total
    total = fan_out(ev, collect, lambda e, a, b: e.add(a, b), 0, q, 2 if two else 0)
    cache.store[(name, "warm")] = True
    out = ev.rescale(total)
    if not end:
        out = _double_copies(ev, out, n, po)
    return TensorCipher(logn, ko, ho, wo, co, to, po if not end else 1, out)


def multiplexed_parallel_batch_norm(ev, cache, name, x, bias, running_mean, running_var, weight, epsilon, B, end=False):
    """`multiplexed_parallel_batch_norm_phantom`, src/cnn_phantom.cu:576-626: subtracts
    (mean * weight / sqrt(var + eps) - bias) / B per channel (the multiplicative part sits in the convolution)."""
    ki, hi, wi, ci, ti, pi, logn = x.k, x.h, x.w, x.c, x.t, x.p, x.logn
    bias, running_mean, running_var, weight = (np.asarray(v, dtype=np.float64) for v in (bias, running_mean, running_var, weight))
    if not (bias.size == running_mean.size == running_var.size == weight.size == ci):
        raise ValueError("the size of bias, running_mean, running_var, or weight are not correct")
    if np.any(np.abs(running_var) < 1e-16):
        raise ValueError("the size of running_var is too small. nearly zero.")
    n = 1 << logn
    if hi * wi * ci > n:
        raise ValueError("hi*wi*ci should not be larger than n")
    if n % pi:
        raise ValueError("n is not divisible by pi")

    def make():
        _, blk, v1, v2, valid = _coords(n, ki, hi, wi, ti, pi)
        ch = ki * ki * blk + ki * (v1 % ki) + v2 % ki
        g = (running_mean * weight / np.sqrt(running_var + epsilon) - bias) / B
        return np.where(valid & (ch < ci), -g[np.minimum(ch, ci - 1)], 0.0)
    ct = ev.add_plain(x.ct, cache.get((name, "bn"), make, x.ct.scale, x.ct.level, n))
    return TensorCipher(logn, ki, hi, wi, ci, ti, pi, ct)


def cnn_add(ev, a, b):
    """`cnn_add_phantom`, src/cnn_phantom.cu:663-680"""
    if a.parms() != b.parms() or a.logn != b.logn:
        raise ValueError("the parameters of cnn1 and cnn2 are not the same")
    lvl = max(a.ct.level, b.ct.level)
    return TensorCipher(a.logn, a.k, a.h, a.w, a.c, a.t, a.p, ev.add(ev.drop(a.ct, lvl), ev.drop(b.ct, lvl)))


def multiplexed_parallel_downsampling(ev, cache, name, x):
    """`multiplexed_parallel_downsampling_phantom`, src/cnn_phantom.cu:681-756: every other pixel, the c channels
    placed in the middle of 2c (zero channels on both sides: the parameter-free shortcut).  One level.  The masks are
    applied after the (hoisted) rotations, rotated accordingly."""
    ki, hi, wi, ci, ti, pi, logn = x.k, x.h, x.w, x.c, x.t, x.p, x.logn
    n = 1 << logn
    ko, ho, wo, to, co = 2 * ki, hi // 2, wi // 2, ti // 2, 2 * ci
    if ti % 8:
        raise ValueError("ti is not multiple of 8")
    if hi % 2 or wi % 2:
        raise ValueError("hi or wi is not even")
    po = 1 << int(math.floor(math.log2(n / (ko * ko * ho * wo * to))))
    if n % po:
        raise ValueError("n is not divisible by po")
    v4 = np.arange(n)
    j5, j6, i7 = (v4 % (ki * ki * hi * wi)) // (ki * wi), v4 % (ki * wi), v4 // (ki * ki * hi * wi)
    base = (v4 < ki * ki * hi * wi * ti) & ((j5 // ki) % 2 == 0) & ((j6 // ki) % 2 == 0)
    pairs, steps = [], []
    for w1 in range(ki):
        for w2 in range(ti):
            m = ki * w2 + w1
            w3, w4, w5 = (m % (2 * ko)) // 2, m % 2, m // (2 * ko)
            pairs.append((w1, w2))
            steps.append(ki * ki * hi * wi * w2 + ki * wi * w1 - ko * ko * ho * wo * w5 - ko * wo * w3 - ki * w4 - ko * ko * ho * wo * (ti // 8))
    ct = x.ct
    level = ct.level
    steps = _rot_steps(steps, n)
    total = None
    for (w1, w2), st, r in zip(pairs, steps, ev.rotate_many(ct, steps)):
        def make(w1=w1, w2=w2, st=st):
            return np.roll(np.where(base & (j5 % ki == w1) & (i7 == w2), 1.0, 0.0), -st)   # mask of the source slots, moved along
        total = ev.mul_plain_add(total, r, cache.get((name, "down", w1, w2), make, ev.q_last(level), level, n))
    out = _double_copies(ev, ev.rescale(total), n, po)
    return TensorCipher(logn, ko, ho, wo, co, to, po, out)


def averagepooling_scale(ev, cache, name, x, B):
    """`averagepooling_phantom_scale`, src/cnn_phantom.cu:758-828: mean over the h x w pixels of every channel, times
    B (back to the network's own units); channel j lands in slot j.  One level."""
    ki, hi, wi, ci, ti, logn = x.k, x.h, x.w, x.c, x.t, x.logn
    n = 1 << logn
    if _log2(hi) == -1 or _log2(wi) == -1:
        raise ValueError("hi or wi is not power of two")
    ct = x.ct
    for s in range(_log2(wi)):
        ct = ev.add(ct, ev.rotate(ct, ((1 << s) * ki) % n))
    for s in range(_log2(hi)):
        ct = ev.add(ct, ev.rotate(ct, ((1 << s) * ki * ki * wi) % n))
    level = ct.level
    pairs = [(s, u) for s in range(ki) for u in range(ti)]
    steps = _rot_steps([-(ki * u + s) * ki + ki * ki * hi * wi * u + ki * wi * s for s, u in pairs], n)
    total = None
    for (s, u), r in zip(pairs, ev.rotate_many(ct, steps)):
        def make(s=s, u=u):
            v = np.zeros(n)
            v[(ki * u + s) * ki:(ki * u + s) * ki + ki] = B / (hi * wi)
            return v
        total = ev.mul_plain_add(total, r, cache.get((name, "avg", s, u), make, ev.q_last(level), level, n))
    return TensorCipher(logn, 1, 1, 1, ci, ti, 1, ev.rescale(total))


def fully_connected(ev, cache, name, x, matrix, bias, q, r):
    """`matrix_multiplication_phantom`, src/cnn_phantom.cu:829-875: q x r matrix times the first r slots by the
    diagonal method, q + r - 1 rotations of one ciphertext.  One level.  The bias is added (the reference builds the
    bias vector, cnn_phantom.cu:842-845, and never adds it)."""
    logn = x.logn
    n = 1 << logn
    matrix, bias = np.asarray(matrix, dtype=np.float64), np.asarray(bias, dtype=np.float64)
    if matrix.size != q * r:
        raise ValueError("the size of matrix is not q*r")
    if bias.size != q:
        raise ValueError("the size of bias is not q")
    M = matrix.reshape(q, r)
    ct = x.ct
    level = ct.level
    total = None
    for s, rot in zip(range(q + r - 1), ev.rotate_many(ct, _rot_steps([r - 1 - s for s in range(q + r - 1)], n))):
        def make(s=s):
            v = np.zeros(n)
            i = np.arange(q)
            j = i + r - 1 - s
            ok = (j >= 0) & (j < r)
            v[i[ok]] = M[i[ok], j[ok]]
            return v
        total = ev.mul_plain_add(total, rot, cache.get((name, "fc", s), make, ev.q_last(level), level, n))
    out = ev.rescale(total)

    def make_bias():
        v = np.zeros(n)
        v[:q] = bias
        return v
    out = ev.add_plain(out, cache.get((name, "fc_bias"), make_bias, out.scale, out.level, n))
    return TensorCipher(logn, x.k, x.h, x.w, x.c, x.t, x.p, out)


# =====================================================================================================================
# approximate ReLU
# =====================================================================================================================
def _remez_odd(deg, a, b, grid=40000, iters=60):
    """The odd polynomial of degree `deg` closest to 1 on [a, b] in the maximum norm (hence to sign(x) on
    [-b, -a] u [a, b]), as Chebyshev coefficients in x / b, and its error.  Remez exchange on a log + linear grid."""
    m = (deg + 1) // 2
    xs = np.unique(np.concatenate([np.geomspace(a, b, grid), np.linspace(a, b, grid)]))
    odd = np.arange(1, deg + 1, 2)
    basis = np.cos(odd[None, :] * np.arccos(np.clip(xs / b, -1.0, 1.0))[:, None])
    ref = np.searchsorted(xs, np.geomspace(a, b, m + 1) if b / a > 50 else a + (b - a) * (1 - np.cos(np.pi * np.arange(m + 1) / m)) / 2)
    ref = np.clip(ref, 0, len(xs) - 1)
    ref = np.unique(ref)
    while len(ref) < m + 1:                      # degenerate start: spread evenly
        ref = np.unique(np.linspace(0, len(xs) - 1, m + 1).astype(int))
    best = None
    for _ in range(iters):
        A = np.concatenate([basis[ref], ((-1.0) ** np.arange(m + 1))[:, None]], axis=1)
        sol = np.linalg.solve(A, np.ones(m + 1))
        coef, E = sol[:m], sol[m]
        err = basis @ coef - 1.0
        worst = np.abs(err).max()
        if best is None or worst < best[1]:
            best = (coef.copy(), worst)
        if worst - abs(E) <= 1e-9 * worst + 1e-15 or worst < 1e-11:
            break
        # one extremum per run of constant sign, then the m + 1 consecutive ones with the largest errors
        sgn = np.sign(err)
        sgn[sgn == 0] = 1
        cuts = np.flatnonzero(np.diff(sgn)) + 1
        if len(cuts) > 64 * m and worst < 1e-9:  # rounding noise: the error is at the floor of double precision
            break
        runs = np.split(np.arange(len(xs)), cuts)
        ext = np.array([run[np.argmax(np.abs(err[run]))] for run in runs])
        while len(ext) > m + 1:
            if np.abs(err[ext[0]]) < np.abs(err[ext[-1]]):
                ext = ext[1:]
            else:
                ext = ext[:-1]
        if len(ext) < m + 1:
            break
        ref = ext
    coef, worst = best
    full = np.zeros(deg + 1)
    full[1::2] = coef
    return full, float(worst)


class ApproxReLU:
    """x (1 + sign(x)) / 2 with sign(x) ~ p_k(... p_1(x)) a composite of odd minimax polynomials (Lee, Lee, No, Kim:
    the reference's `minimax_ReLU_phantom`, src/cnn_phantom.cu:876-940, degrees {15, 15, 27} for precision 2^-13,
    examples/cnn.cu:54-56).  p_1 is the minimax approximation of 1 on [eps, 1], p_i that on the range of p_{i-1};
    eps is chosen so that the maximum error of the ReLU over [-1, 1] is smallest (degrees {15, 15, 27}: eps = 2^-9.15,
    error 2^-13.16 — the precision `alpha` of the reference is this error, not eps: the slope of the composite at 0
    is at most the product of the degrees).  Every component is kept as Chebyshev coefficients in its own normalised
    variable (the previous component is divided by the bound of its range, the last one by 2: the reference's
    scale_val and 0.5)."""

    _grid = np.concatenate([np.linspace(-1, 1, 100001), np.geomspace(1e-6, 1, 50000), -np.geomspace(1e-6, 1, 50000)])

    # the result of the search below for the reference's degrees (saves its four seconds)
    _known_eps = {(15, 15, 27): 2.0 ** -9.118668693380496}

    def __init__(self, degrees=(15, 15, 27), eps=None):
        self.degrees = tuple(degrees)
        eps = self._known_eps.get(self.degrees) if eps is None else eps
        self.batched = os.environ.get("PHA_RELU_BATCHED", "0") != "0"     # nodes of one height as one batched product
        self.parallel = os.environ.get("PHA_RELU_PARALLEL", "1") != "0"   # the halves of every component on two streams
        if eps is None:
            best = None
            top = math.log2(float(np.prod(self.degrees)))      # the slope of the composite at 0 is below prod(degrees)
            for le in np.arange(max(1.0, top - 6.0), top, 0.05):
                self._build(2.0 ** -le)
                err = np.abs(self(self._grid) - np.maximum(self._grid, 0.0)).max()
                if best is None or err < best[0]:
                    best = (err, 2.0 ** -le)
                elif err > 4 * best[0]:
                    break
            eps = best[1]
        self._build(eps)
        self.eps = eps
        self.precision = float(np.abs(self(self._grid) - np.maximum(self._grid, 0.0)).max())
        self.alpha = -math.log2(self.precision)
        # what the circuit evaluates: component i outputs the next component's normalised variable
        self.scaled = []
        for i, coef in enumerate(self.components):
            nxt = self.bounds[i + 1] if i + 1 < len(self.components) else 2.0
            self.scaled.append(coef / nxt)
        self.depth = sum(max(2, math.ceil(math.log2(d + 1))) for d in self.degrees) + 1

    def _build(self, eps):
        a, b = eps, 1.0
        self.components, self.bounds, self.errors = [], [], []
        for d in self.degrees:
            coef, e = _remez_odd(d, a, b, grid=10000)
            self.components.append(coef)
            self.bounds.append(b)
            self.errors.append(e)
            if not e < 1.0:
                raise ValueError("degree %d cannot separate [%g, %g] from its mirror image" % (d, a, b))
            a, b = 1.0 - e, 1.0 + e

    def sign(self, x):
        """the composite on plain numbers (|x| <= 1)"""
        y = np.asarray(x, dtype=np.float64)
        for coef, b in zip(self.components, self.bounds):
            y = np.polynomial.chebyshev.chebval(np.clip(y / b, -1.0, 1.0), coef)
        return y

    def __call__(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x * (1.0 + self.sign(x)) / 2.0

    def evaluate(self, ev, ct):
        """levels: self.depth; the result has the scale of the input"""
        y = ct
        last = len(self.scaled) - 1
        for i, coef in enumerate(self.scaled):
            cheb = ChebyshevEvaluator(ev, coef, baby_log=2, optimal_depth=True)
            cheb.parallel = self.parallel
            cheb.batched = self.batched
            if i == last:
                y = cheb(y, ev.q_last(y.level + cheb.depth))       # (sign/2 + 1/2) * x then rescales back onto ct.scale
            else:
                y = cheb(y)
        y = ev.add_const(y, 0.5)
        return ev.mul_rescale(y, ev.drop(ct, y.level))


_RELUS = {}


def default_relu(degrees=(15, 15, 27)):
    """the composite for `degrees`, generated once per process (a few seconds of Remez exchanges)"""
    key = tuple(degrees)
    if key not in _RELUS:
        _RELUS[key] = ApproxReLU(key)
    return _RELUS[key]


def approx_relu(ev, x, relu):
    """`ReLU_phantom`, src/cnn_phantom.cu:628-644"""
    if x.h * x.w * x.c > 1 << x.logn:
        raise ValueError("hi*wi*ci should not be larger than n")
    return TensorCipher(x.logn, x.k, x.h, x.w, x.c, x.t, x.p, relu.evaluate(ev, x.ct))


# =====================================================================================================================
# parameters and the network
# =====================================================================================================================
def parameter_names(layer_num):
    """file stems of `import_parameters_cifar10`, src/cnn_phantom.cu:1474-1731"""
    end_num = {20: 2, 32: 4, 44: 6, 56: 8, 110: 17}[layer_num]
    names = ["conv1_weight"] + ["bn1_" + s for s in ("bias", "running_mean", "running_var", "weight")]
    for j in (1, 2, 3):
        for k in range(end_num + 1):
            for i in (1, 2):
                names.append("layer%d_%d_conv%d_weight" % (j, k, i))
                names += ["layer%d_%d_bn%d_%s" % (j, k, i, s) for s in ("bias", "running_mean", "running_var", "weight")]
    return names + ["linear_weight", "linear_bias"]


def import_parameters_cifar10(directory, layer_num):
    """the reference's text files (one number per line) -> {stem: float64 array}"""
    return {name: np.loadtxt(os.path.join(directory, name + ".txt"), dtype=np.float64).ravel() for name in parameter_names(layer_num)}


def load_parameters_npz(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: np.asarray(z[k], dtype=np.float64).ravel() for k in z.files}


class ResNetCifar10:
    """`ResNet_cifar10_seal_sparse`, examples/cnn.cu:48-426: conv-bn-ReLU, three stages of residual blocks
    (16, 32, 64 channels; 32, 16, 8 pixels; gaps 1, 2, 4), average pooling, fully connected layer; a bootstrapping
    before every ReLU after the first.  `bootstrappers`: {slots: Bootstrapper} for the packings of the three stages
    (n/2, n/4, n/8 slots at n = 2^15), or None to run without (tests on short networks and long chains)."""

    def __init__(self, ev, params, layer_num=20, B=40.0, relu=None, bootstrappers=None, logn=None, widths=(16, 32, 64),
                 image_shape=(3, 32, 32), epsilon=1e-5, init_p=None):
        self.ev, self.params, self.B, self.epsilon = ev, params, B, epsilon
        self.end_num = {8: 0, 14: 1, 20: 2, 32: 4, 44: 6, 56: 8, 110: 17}[layer_num]
        self.layer_num = layer_num
        self.relu = relu or default_relu()
        self.boot = bootstrappers
        self.logn = ev.log_n - 1 if logn is None else logn
        self.widths, self.image_shape = widths, image_shape
        c, h, w = image_shape
        n = 1 << self.logn
        self.init_p = init_p or 1 << int(math.floor(math.log2(n / (h * w * c))))
        self.cache = PlainCache(ev)
        self.trace = None

    def levels_after_bootstrap(self):
        """levels one block needs between two bootstrappings: ReLU, then a convolution (2), then whatever the
        bootstrapper spends before its modulus raise (the slim order's SlotToCoeff)"""
        pre = max((b.pre_levels for b in self.boot.values()), default=0) if self.boot else 0
        return self.relu.depth + 2 + pre

    def input_level(self):
        """level to encrypt the image at: the first convolution comes before the first ReLU"""
        return self.ev.bottom - self.levels_after_bootstrap() - 2

    def encrypt_image(self, image, scale, level):
        """image [c][h][w] in the network's units; the ciphertext holds image / B (examples/cnn.cu:263-278)"""
        c, h, w = self.image_shape
        vec = pack_tensor(np.asarray(image, dtype=np.float64).reshape(c, h, w) / self.B, self.logn, 1, self.init_p)
        ev = self.ev
        ct = ev.encrypt(ev.encode(vec.astype(np.complex128), scale, level, 1 << self.logn))
        return TensorCipher(self.logn, 1, h, w, c, c, self.init_p, ct)

    def _bn_args(self, stem):
        p = self.params
        return p[stem + "_bias"], p[stem + "_running_mean"], p[stem + "_running_var"], p[stem + "_weight"]

    def _conv_bn(self, x, conv, bn, co, st):
        p, ev = self.params, self.ev
        _, _, var, weight = self._bn_args(bn)
        x = multiplexed_parallel_convolution(ev, self.cache, conv, x, co, st, 3, 3, p[conv + "_weight"], var, weight, self.epsilon)
        return multiplexed_parallel_batch_norm(ev, self.cache, bn, x, *self._bn_args(bn), self.epsilon, self.B)

    def _bootstrap(self, x):
        if self.boot is None:
            return x
        slots = (1 << x.logn) // x.p
        bs = self.boot[slots]
        return TensorCipher(x.logn, x.k, x.h, x.w, x.c, x.t, x.p, bs.bootstrap(x.ct))

    def _mark(self, tag, x):
        if self.trace is not None:
            self.trace(tag, x)

    def __call__(self, x):
        ev = self.ev
        x = self._conv_bn(x, "conv1", "bn1", self.widths[0], 1)
        x = approx_relu(ev, x, self.relu)
        self._mark("layer0", x)
        for j, co in enumerate(self.widths):
            for k in range(self.end_num + 1):
                stem = "layer%d_%d" % (j + 1, k)
                st = 2 if j >= 1 and k == 0 else 1
                shortcut = x
                x = self._conv_bn(x, stem + "_conv1", stem + "_bn1", co, st)
                x = approx_relu(ev, self._bootstrap(x), self.relu)
                x = self._conv_bn(x, stem + "_conv2", stem + "_bn2", co, 1)
                if st == 2:
                    shortcut = multiplexed_parallel_downsampling(ev, self.cache, stem + "_down", shortcut)
                x = cnn_add(ev, shortcut, x)
                x = approx_relu(ev, self._bootstrap(x), self.relu)
                self._mark(stem, x)
        x = averagepooling_scale(ev, self.cache, "avgpool", x, self.B)
        return fully_connected(ev, self.cache, "linear", x, self.params["linear_weight"], self.params["linear_bias"],
                               self.params["linear_bias"].size, self.widths[-1])

    def capture(self, x):
        """The whole inference as ONE HIP graph for inputs shaped like x (GPU back end): the convolutions and ReLUs
        work on 1-17 limbs, kernels of a few microseconds that a host issuing them one by one cannot keep fed.
        Returns run(x) -> TensorCipher; the result buffer is reused by the next run."""
        import torch
        from .ckks import Ct
        static_in = TensorCipher(x.logn, x.k, x.h, x.w, x.c, x.t, x.p, Ct(x.ct.data.clone(), x.ct.level, x.ct.scale))
        # A forked stream that forks again (run_pair inside run_pair) sends hipStreamEndCapture of the HIP runtime bundled
        # with torch (7.0) into an endless recursion over the streams' parallel-capture lists; ROCm 7.2's runtime ends the
        # same capture correctly (HIP-only reproducer and backtrace: profiles/r2c_hipgraph_nested_fork.md).  Under the old
        # runtime the capture uses pairs of streams only; PHA_CAPTURE_NEST overrides the depth.
        cls = type(self.ev)
        hip = tuple(int(v) for v in (torch.version.hip or "0.0").split(".")[:2])
        depth = int(os.environ.get("PHA_CAPTURE_NEST", cls.MAX_NEST if hip >= (7, 2) else 1))
        nest, cls.MAX_NEST = cls.MAX_NEST, min(cls.MAX_NEST, depth)
        try:
            return self._capture(static_in)
        finally:
            cls.MAX_NEST = nest

    def _capture(self, static_in):
        import torch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self(static_in)                      # constants encoded, keys generated, workspaces at their final size
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self(static_in)

        def run(y):
            static_in.ct.data.copy_(y.ct.data)
            graph.replay()
            return static_out
        run.graph = graph
        return run

    def decrypt_logits(self, x):
        q = self.params["linear_bias"].size
        return np.real(self.ev.decrypt_decode(x.ct, 1 << x.logn)[:q])
