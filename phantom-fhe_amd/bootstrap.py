"""CKKS bootstrapping for sparsely packed ciphertexts over the evaluator interface of ckks.py
(the reference's Bootstrapper::bootstrap_sparse*, src/boot/Bootstrapper.cu:2863-2934,
and ModularReducer, src/boot/ModularReducer.cu).

Pipeline (s = sparse slots, g = N/(2s), message polynomial m(Y), Y = X^g, at the one-limb level):

    ModRaise      t(X) = m + q0*I(X) over the top level; the scale is re-declared as q0      (:2814-2868)
    SubSum        log2(g) rotate-and-add steps project onto the subring: g*(m + q0*I')(Y)       (:2870-2876)
    CoeffToSlot   one BSGS linear transform with conj(A)^T/(s*g*Kb) puts (coefficient_k +
                  i*coefficient_{k+s}) / (q0*Kb) into slot k; real and imaginary parts are split
                  with the conjugate (exact operations)
    EvalMod       on both parts: Chebyshev interpolant of cos(2*pi*(Kb*w - 1/4) / 2^r) by a
                  baby-step/giant-step evaluation with exact target scales, then r double angles:
                  sin(2*pi*x) ~ 2*pi*(x - I) for x = m/q0 + I
    SlotToCoeff   re + i*im recombined (exact), one BSGS linear transform with A/(2*pi)
    the scale is re-declared so that the result decodes to the input message              (:2933)

Differences from the reference, which a maintainer should know: the reference factors the two
linear transforms FFT-style over several levels and takes its polynomial from a Remez
approximation computed with NTL; here each transform is ONE dense s x s BSGS level (right for the
sparse packings s <= 2^9 that bootstrap_sparse targets; baby steps use hoisted rotations) and the
polynomial is the Chebyshev interpolant computed in double precision — no NTL, no files.
"""
import math

import numpy as np


def slot_matrix(s):
    """A[j][k] = xi^(5^j k), xi = exp(2 pi i / 4s): slots = A @ (coeff_k + i coeff_{k+s})"""
    m = 4 * s
    rot = [pow(5, j, m) for j in range(s)]
    e = np.array([[(r * k) % m for k in range(s)] for r in rot], dtype=np.float64)
    return np.exp(2j * np.pi * e / m)


class LinearTransform:
    """z -> M z on the s sparse slots by baby-step/giant-step diagonals, one level.
    M z = sum_g rot( sum_b rot(diag_{g*n1+b}, -g*n1) * rot(z, b), g*n1 )."""

    def __init__(self, ev, matrix, level, diag_scale, rescales=1):
        s = matrix.shape[0]                 # the transform's slot count (a power of two <= N/2)
        self.ev, self.s, self.level, self.diag_scale, self.rescales = ev, s, level, diag_scale, rescales
        self.n1 = 1 << math.ceil(math.log2(s) / 2)
        self.n2 = s // self.n1
        j = np.arange(s)
        self.plains = [[None] * self.n1 for _ in range(self.n2)]       # [giant][baby]
        for g in range(self.n2):
            for b in range(self.n1):
                d = g * self.n1 + b
                diag = matrix[j, (j + d) % s]
                if np.abs(diag).max() < 1e-300:
                    continue
                self.plains[g][b] = ev.encode_lt(np.roll(diag, g * self.n1), diag_scale, level, s)

    def steps(self):
        return [b for b in range(1, self.n1)] + [g * self.n1 for g in range(1, self.n2)]

    def apply(self, ct):
        ev = self.ev
        total = ev.bsgs(ev.drop(ct, self.level), list(range(self.n1)), [g * self.n1 for g in range(self.n2)], self.plains)
        for _ in range(self.rescales):      # 2: diagonals encoded over two primes (double precision)
            total = ev.rescale(total)
        return total


# ---- linear maps as sparse sets of generalised diagonals ---------------------------------------------
# {d: v} stands for x -> sum_d v * rot(x, d) with rot(x, d)[j] = x[(j + d) mod S].
def compose_diagonals(outer, inner, S):
    """(outer o inner): v1 * rot(v2 * rot(x, d2), d1) = v1 * rot(v2, d1) * rot(x, d1 + d2)"""
    out = {}
    for d1, v1 in outer.items():
        for d2, v2 in inner.items():
            d = (d1 + d2) % S
            term = v1 * np.roll(v2, -d1)
            out[d] = out[d] + term if d in out else term
    return {d: v for d, v in out.items() if np.abs(v).max() > 1e-300}


def special_fft_stage(S, length, inverse):
    """One radix-2 stage of the special FFT over the rotation group (the butterflies of src/fft.cu with
    twiddle exp(2 pi i (5^j mod 4 len) / (4 len))) as three diagonals.  Forward: (u, v) -> (u + w v, u - w v);
    inverse: (a, b) -> ((a + b)/2, (a - b) conj(w) / 2).  slot_matrix(S) = F_S o ... o F_4 o F_2 o bitrev."""
    lenh, lenq = length // 2, length * 4
    j = np.arange(S) % length
    first = j < lenh
    tw = np.exp(2j * np.pi * np.array([pow(5, int(t % lenh), lenq) for t in j]) / lenq)
    if not inverse:
        parts = [(0, np.where(first, 1.0, -tw)), (lenh, np.where(first, tw, 0.0)), (S - lenh, np.where(first, 0.0, 1.0))]
    else:
        parts = [(0, np.where(first, 0.5, -0.5 * np.conj(tw))), (lenh, np.where(first, 0.5, 0.0)),
                 (S - lenh, np.where(first, 0.0, 0.5 * np.conj(tw)))]
    out = {}
    for d, v in parts:      # at length == S the offsets lenh and S - lenh coincide
        d %= S
        v = v.astype(np.complex128)
        out[d] = out[d] + v if d in out else v
    return out


def special_fft_groups(S, n_groups, inverse):
    """The log2(S) stages merged into n_groups maps, in application order.  Forward order is
    len = 2, 4, .., S on bit-reversed input; the inverse applies len = S, .., 2 and leaves its output
    bit-reversed — the two permutations cancel around the slot-wise EvalMod."""
    lengths = [1 << k for k in range(1, S.bit_length())]
    if inverse:
        lengths = lengths[::-1]
    per = [len(lengths) // n_groups + (1 if i < len(lengths) % n_groups else 0) for i in range(n_groups)]
    groups, pos = [], 0
    for cnt in per:
        m = {0: np.ones(S, dtype=np.complex128)}
        for ln in lengths[pos:pos + cnt]:
            m = compose_diagonals(special_fft_stage(S, ln, inverse), m, S)
        groups.append(m)
        pos += cnt
    return [g for g in groups if g]


def dense_to_diagonals(matrix):
    S = matrix.shape[0]
    j = np.arange(S)
    out = {}
    for d in range(S):
        v = matrix[j, (j + d) % S]
        if np.abs(v).max() > 1e-300:
            out[d] = v
    return out


class DiagonalTransform:
    """x -> sum_d v_d * rot(x, d) on S slots with baby-step/giant-step rotations over the offsets that
    occur (offsets are multiples of a stride: d = stride * (g * n1 + b)); the evaluator's `bsgs` does the
    sum_g rot_G(sum_b pt[g][b] * rot_B(x)) — double-hoisted on the GPU (one modup for the babies, one moddown per
    giant step).  `rescales` levels are consumed (2: diagonals encoded over two primes)."""

    def __init__(self, ev, diagonals, S, level, diag_scale, rescales=1, baby_bias=None):
        self.ev, self.S, self.level, self.rescales = ev, S, level, rescales
        if baby_bias is None:       # log2 of (cost of a giant step / cost of a baby step)
            baby_bias = getattr(ev, "baby_bias", 0)
        offs = sorted(diagonals)
        stride = S
        for d in offs:
            if d:
                stride = math.gcd(stride, d)
        m = S // stride
        ks = [d // stride for d in offs]
        self.stride = stride
        self.n1 = max(1, 1 << math.ceil((math.log2(max(2, len(ks))) + baby_bias) / 2))
        self.n1 = min(self.n1, m)
        split = {d: divmod(k, self.n1) for d, k in zip(offs, ks)}
        self.babies = sorted({b for _, b in split.values()})
        self.giants = sorted({g for g, _ in split.values()})
        self.plains = [[None] * len(self.babies) for _ in self.giants]     # [giant][baby] -> plaintext
        for d, (g, b) in split.items():
            shift = g * self.n1 * stride
            self.plains[self.giants.index(g)][self.babies.index(b)] = ev.encode_lt(np.roll(diagonals[d], shift), diag_scale, level, S)

    def steps(self):
        return [b * self.stride for b in self.babies if b] + [g * self.n1 * self.stride for g in self.giants if g]

    def apply(self, ct):
        ev = self.ev
        total = ev.bsgs(ev.drop(ct, self.level), [b * self.stride for b in self.babies],
                        [g * self.n1 * self.stride for g in self.giants], self.plains)
        for _ in range(self.rescales):
            total = ev.rescale(total)
        return total


class ChebyshevEvaluator:
    """sum_j c_j T_j(w) with depth ceil(log2(d+1)) + 1 and exact scales: every addition joins
    operands of identical (level, scale) because constants are encoded at the scale that makes the
    rescaled product land on the target (Paterson-Stockmeyer over the Chebyshev basis:
    p = q * T_h + r with T_j = 2 T_{j-h} T_h - T_{2h-j})."""

    def __init__(self, ev, coeffs, baby_log=2, optimal_depth=False):
        """optimal_depth: ceil(log2(d+1)) levels instead of one more.  The leaves c1 T1 + c2 T2 + c3 T3 are computed as
        (c1 - c3) T1 + c2 T2 + (2 c3 T1) T2, the constant riding on T1 one level above T2: T3 is never formed and a
        leaf ends at the level of T3 itself, at the price of one ciphertext product per leaf (the approximate ReLU
        of cnn.py, whose three polynomials would otherwise cost three more levels)."""
        self.ev = ev
        self.c = np.array(coeffs, dtype=np.float64)
        d = len(self.c) - 1
        self.m = max(2 if optimal_depth else 1, math.ceil(math.log2(d + 1)))
        self.l = min(baby_log, self.m)
        self.optimal = optimal_depth
        self.parallel = False       # evaluate the two halves of the top split (p = q T_h + r) with ev.run_pair
        self.batched = False        # optimal depth only: all nodes of one height at one level, as batched products
        assert not optimal_depth or self.l == 2
        self.depth = self.m if optimal_depth else self.m + 1

    def _at(self, j, level):
        """T_j at `level` (dropping limbs copies the ciphertext: once per level)"""
        key = (j, level)
        t = self._dropped.get(key)
        if t is None:
            t = self._dropped[key] = self.ev.drop(self.T[j], level)
        return t

    def _powers(self, w):
        ev = self.ev
        self._dropped = {}
        T = {1: w}
        for j in range(2, 3 if self.optimal else 1 << self.l):          # babies T_2 .. T_{2^l - 1}
            a = 1 << (j.bit_length() - 1)
            T[j] = self._product(T, j // 2, j // 2) if j == a else self._product(T, a, j - a)
        for k in range(self.l, self.m):          # giants T_{2^l} .. T_{2^(m-1)}
            h = 1 << k
            if h not in T:
                T[h] = self._product(T, h // 2, h // 2)
        return T

    def _product(self, T, a, b):
        """T_{a+b} = 2 T_a T_b - T_{|a-b|}"""
        ev = self.ev
        x, y = T[a], T[b]
        lvl = max(x.level, y.level)
        p = ev.mul_rescale(ev.drop(x, lvl), ev.drop(y, lvl), times2=True)
        if a == b:
            return ev.add_const(p, -1.0)
        return ev.sub(p, ev.to_level_scale(T[abs(a - b)], p.level, p.scale))

    def __call__(self, w, target_scale=None):
        ev = self.ev
        self.T = self._powers(w)
        self.L0 = w.level
        level = self.L0 + self.depth
        scale = w.scale if target_scale is None else target_scale
        if self.batched and self.optimal:
            return self._eval_by_height(scale)
        return self._eval(self.c, self.m, level, scale)

    # ---- the same recursion, evaluated height by height ---------------------------------------------------------
    def _plan(self, c, k, scale):
        """tree of the Paterson-Stockmeyer recursion with every node of height k placed at level L0 + k (the earliest
        it can be computed): nodes of one height are independent products at one level"""
        ev = self.ev
        c = np.trim_zeros(np.asarray(c, dtype=np.float64), "b")
        if len(c) == 0:
            return None
        if k <= self.l or len(c) <= (1 << self.l):
            return {"kind": "leaf", "c": c[:1 << self.l], "height": self.l, "scale": scale}
        h = 1 << (k - 1)
        if len(c) <= h:
            return self._plan(c, k - 1, scale)
        qc = np.zeros(len(c) - h)
        r = np.array(c[:h], dtype=np.float64)
        qc[0] = c[h]
        for j in range(h + 1, len(c)):
            qc[j - h] = 2.0 * c[j]
            r[2 * h - j] -= c[j]
        Th = self._at(h, self.L0 + k - 1)
        return {"kind": "node", "height": k, "scale": scale, "Th": Th,
                "q": self._plan(qc, k - 1, scale * ev.q_last(self.L0 + k - 1) / Th.scale), "r": self._plan(r, k - 1, scale)}

    def _eval_by_height(self, scale):
        ev = self.ev
        root = self._plan(self.c, self.m, scale)
        nodes = []

        def walk(nd):
            if nd is not None:
                nodes.append(nd)
                if nd["kind"] == "node":
                    walk(nd["q"])
                    walk(nd["r"])
        walk(root)
        # leaves: odd ones (2 c3 T1)(T2 + kappa) as ONE batch of products at level L0 + 1, the others one by one
        lvl = self.L0 + self.l - 1
        q = ev.q_last(lvl)
        batch = []
        for nd in nodes:
            if nd["kind"] != "leaf":
                continue
            c = nd["c"]
            if len(c) > 3 and abs(c[3]) >= 1e-300 and abs(c[2]) < 1e-300 and abs((c[1] - c[3]) / (2.0 * c[3])) <= 16.0:
                t1, t2 = self._at(1, lvl - 1), self._at(2, lvl)
                a = ev.rescale(ev.mul_const(t1, 2.0 * c[3], (nd["scale"] * q / t2.scale) * ev.q_last(lvl - 1) / t1.scale))
                batch.append((nd, a, ev.add_const(t2, (c[1] - c[3]) / (2.0 * c[3]))))
            else:
                nd["val"] = self._leaf(c, self.L0 + self.l, nd["scale"])
        for (nd, _, _), out in zip(batch, ev.mul_rescale_batch([(a, b) for _, a, b in batch])):
            nd["val"] = ev.add_const(out, nd["c"][0]) if abs(nd["c"][0]) >= 1e-300 else out

        def at_level(v, level):
            return ev.drop(v, level) if isinstance(v, type(self.T[1])) and v.level < level else v
        for k in range(self.l + 1, self.m + 1):
            todo = [nd for nd in nodes if nd["kind"] == "node" and nd["height"] == k]
            pairs, owners = [], []
            for nd in todo:
                qv = None if nd["q"] is None else nd["q"]["val"]
                if isinstance(qv, tuple):            # q is a constant: a plaintext multiple of T_h
                    nd["prod"] = ev.rescale(ev.mul_const(nd["Th"], qv[1], nd["q"]["scale"]))
                elif qv is None:
                    nd["prod"] = None
                else:
                    pairs.append((at_level(qv, self.L0 + k - 1), nd["Th"]))
                    owners.append(nd)
            for nd, out in zip(owners, ev.mul_rescale_batch(pairs)):
                nd["prod"] = out
            for nd in todo:
                prod, rv = nd["prod"], (None if nd["r"] is None else nd["r"]["val"])
                if prod is None:
                    nd["val"] = rv if rv is None or isinstance(rv, tuple) else at_level(rv, self.L0 + k)
                elif rv is None:
                    nd["val"] = prod
                elif isinstance(rv, tuple):
                    nd["val"] = ev.add_const(prod, rv[1])
                else:
                    nd["val"] = ev.add(prod, at_level(rv, self.L0 + k))
        out = root["val"]
        return at_level(out, self.L0 + self.depth)

    def _leaf(self, c, level, scale):
        """sum_j c_j T_j, j < 2^l, landing exactly on (level, scale)"""
        ev = self.ev
        lvl = level - 1
        q = ev.q_last(lvl)
        acc = None
        if self.optimal and len(c) > 3 and abs(c[3]) >= 1e-300:
            # (2 c3 T1) T2, T3 = 2 T1 T2 - T1 puts -c3 on T1
            t1, t2 = self._at(1, lvl - 1), self._at(2, lvl)
            a = ev.rescale(ev.mul_const(t1, 2.0 * c[3], (scale * q / t2.scale) * ev.q_last(lvl - 1) / t1.scale))
            kappa = (c[1] - c[3]) / (2.0 * c[3])
            if abs(c[2]) < 1e-300 and abs(kappa) <= 16.0:
                # an odd leaf: (2 c3 T1) (T2 + kappa) is the whole of it, one merged product
                out = ev.mul_rescale(a, ev.add_const(t2, kappa))
                return ev.add_const(out, c[0]) if abs(c[0]) >= 1e-300 else out
            acc = ev.mul(a, t2)                                  # unrescaled at (lvl, scale * q)
            c = np.array([c[0], c[1] - c[3], c[2]])
        for j in range(1, len(c)):
            if abs(c[j]) < 1e-300:
                continue
            t = self._at(j, lvl)
            term = ev.mul_const(t, c[j], scale * q / t.scale)
            acc = term if acc is None else ev.add(acc, term)
        if acc is None:
            return None if abs(c[0]) < 1e-300 else ("const", c[0])
        if abs(c[0]) >= 1e-300:
            acc = ev.add_const(acc, c[0])
        return ev.rescale(acc)

    def _eval(self, c, k, level, scale):
        ev = self.ev
        c = np.trim_zeros(np.asarray(c, dtype=np.float64), "b")
        if len(c) == 0:
            return None
        if k <= self.l or len(c) <= (1 << self.l):
            return self._leaf(c[:1 << self.l] if len(c) > (1 << self.l) else c, level, scale)
        h = 1 << (k - 1)
        if len(c) <= h:
            return self._eval(c, k - 1, level, scale)
        # p = q*T_h + r
        qc = np.zeros(len(c) - h)
        r = np.array(c[:h], dtype=np.float64)
        qc[0] = c[h]
        for j in range(h + 1, len(c)):
            qc[j - h] = 2.0 * c[j]
            r[2 * h - j] -= c[j]
        Th = self._at(h, level - 1)
        if self.parallel and k >= self.m - 1 and k > self.l:
            # two independent halves at 5-20 limbs each: side by side on the sibling (context, stream) pair
            def other_half(e2):
                twin = ChebyshevEvaluator(e2, self.c, self.l, self.optimal)
                twin.T, twin._dropped, twin.L0, twin.parallel = self.T, {}, self.L0, self.parallel
                return twin._eval(qc, k - 1, level - 1, scale * e2.q_last(level - 1) / Th.scale)
            rct, qct = ev.run_pair(lambda e1: self._eval(r, k - 1, level, scale), other_half)
        else:
            qct = self._eval(qc, k - 1, level - 1, scale * ev.q_last(level - 1) / Th.scale)
            rct = self._eval(r, k - 1, level, scale)
        if isinstance(qct, tuple):       # q is a constant: q*T_h is a plaintext multiple of T_h
            prod = ev.rescale(ev.mul_const(Th, qct[1], scale * ev.q_last(level - 1) / Th.scale))
        elif qct is None:
            prod = None
        else:
            prod = ev.mul_rescale(qct, Th)
        if prod is None:
            return rct
        if rct is None:
            return prod
        if isinstance(rct, tuple):
            return ev.add_const(prod, rct[1])
        return ev.add(prod, rct)


class Bootstrapper:
    """Sparse-slot bootstrapping.  Levels used: 2 (CoeffToSlot) + cheb.depth + r (EvalMod) + 1
    (SlotToCoeff), starting from level 1 after the modulus raise.  The EvalMod primes should be close
    to the working scale delta, q0 about 2^10 * delta (the sine is used where it is linear).

    packed (default whenever 2s <= N/2, like the reference's sparse mode): the linear transforms act
    on the 2s-slot subring, CoeffToSlot leaves the s real parts in slots [0, s) and the s imaginary
    parts in slots [s, 2s) of ONE ciphertext, so EvalMod runs once.  Otherwise real and imaginary
    parts go through EvalMod as two ciphertexts."""

    def __init__(self, ev, K=12, r=3, degree=31, delta=2.0 ** 40, baby_log=2, packed=None,
                 transform="dense", c2s_groups=3, s2c_groups=3, msg_scale=None, slots=None, slim=False):
        """delta: the scale EvalMod works at (its primes should be close to it); msg_scale: the scale of
        the ciphertexts to refresh (default delta).  A delta well above msg_scale (say 2^58 against 2^46)
        lowers the EvalMod noise floor, which full slots see amplified by sqrt(s) * q0 / (2 pi msg_scale)."""
        self.ev, self.K, self.r, self.delta = ev, K, r, delta
        self.msg_scale = delta if msg_scale is None else msg_scale
        s, n = (ev.s if slots is None else slots), ev.n      # slots: several packings on one evaluator (cnn.py)
        self.s = s
        self.transform = transform
        self.slim = slim
        self.pre_levels = 0
        self.delta_out = delta               # scale of the Chebyshev polynomial's result (the slim order tunes it)
        assert transform in ("dense", "fft")
        assert not slim or transform == "fft", "the slim order is built on the FFT-factored transforms"
        # packing doubles the dimension of the dense linear transforms: it pays while they are small
        self.packed = (4 * s <= n and s <= 64 and transform == "dense") if packed is None else packed
        assert not (self.packed and transform == "fft"), "the FFT-factored transforms run unpacked"
        assert not self.packed or 4 * s <= n
        S = 2 * s if self.packed else s      # slots of the subring the circuit works in
        self.S, self.g = S, n // (2 * S)
        self.Kb = K + 0.5
        # Chebyshev interpolant of f(w) = cos(2 pi (Kb w - 1/4) / 2^r) on [-1, 1]
        f = lambda w: np.cos(2.0 * np.pi * (self.Kb * w - 0.25) / (1 << r))  # noqa: E731
        self.cheb = ChebyshevEvaluator(ev, np.polynomial.chebyshev.chebinterpolate(f, degree), baby_log)
        q0 = ev.q[0]
        norm = 1.0 / (self.g * self.Kb * 2.0)
        if slim:
            self._init_slim(norm, c2s_groups, s2c_groups)
        elif transform == "fft":
            self._init_fft(norm, c2s_groups, s2c_groups)
        else:
            self._init_dense(norm)
        steps = set()
        for t in self.c2s + self.s2c:
            steps |= set(t.steps())
        k = S
        while k < n // 2:
            steps.add(k)
            k <<= 1
        ev.add_rotation_keys(sorted(steps), conj=True)

    def _evalmod_out(self, level):
        """(scale, level) of the EvalMod output: delta after the polynomial, squared and rescaled by every
        double angle"""
        sv, lvl = self.delta_out, level + self.cheb.depth
        for _ in range(self.r):
            sv, lvl = sv * sv / self.ev.q_last(lvl), lvl + 1
        return sv, lvl

    def _init_dense(self, norm):
        ev, s, S, r, delta = self.ev, self.s, self.S, self.r, self.delta
        q0 = ev.q[0]
        A = slot_matrix(s)
        if self.packed:
            # u' = U z' are the 2s complex coefficient pairs of the 2s-slot subring; the message sits on the
            # even ones: Re u'_{2k} = coefficient k, Im u'_{2k} = coefficient k+s.  v = (u'_{2k}/2, -i u'_{2k}/2)
            # makes v + conj(v) the packed real vector; the odd ones (pure q0*I garbage) are never read.
            U = slot_matrix(S).conj().T / S
            c2s_m = np.concatenate([U[0::2], -1j * U[0::2]]) * norm
            # slots of the output (period s): A (y_first + i y_second)
            s2c_m = np.tile(np.concatenate([A, 1j * A], axis=1), (2, 1))
        else:
            c2s_m = A.conj().T / s * norm
            s2c_m = A
        # levels 1-2: the CoeffToSlot input carries q0*I*g (~2^63), and an error eps in x = m/q0 + I comes
        # back multiplied by q0/delta in the message, so the diagonals are encoded over TWO primes (their
        # integers would otherwise have only ~27 bits) at the scale that turns the re-declared scale q0
        # into delta after both rescales
        c2s_scale = delta * ev.q_last(1) * ev.q_last(2) / q0
        self.c2s = [LinearTransform(ev, c2s_m, 1, c2s_scale, rescales=2)]
        self.evalmod_level = 3
        self.s2c_level = 3 + self.cheb.depth + r
        sv, _ = self._evalmod_out(3)
        # 1/(2 pi) undoes the sine's factor, q0/delta brings the message back from units of q0 to the
        # working scale: the output is a ciphertext of scale delta again
        self.s2c = [LinearTransform(ev, s2c_m * (q0 / self.msg_scale) / (2.0 * np.pi), self.s2c_level,
                                   self.msg_scale * ev.q_last(self.s2c_level) / sv)]
        self.out_level = self.s2c_level + 1

    def _init_fft(self, norm, c2s_groups, s2c_groups):
        """CoeffToSlot / SlotToCoeff as products of merged special-FFT stages: log2(s) stages in
        c2s_groups (s2c_groups) sparse diagonal maps, one level each, O(groups * 2^(log s / groups))
        rotations instead of O(sqrt s) with s-diagonal plaintext storage.  The inverse transform leaves
        the coefficients in bit-reversed slot order, which the slot-wise EvalMod does not see and the
        forward transform (defined on bit-reversed input) undoes."""
        ev, S, delta = self.ev, self.S, self.delta
        q0 = ev.q[0]
        inv = special_fft_groups(S, min(c2s_groups, max(1, S.bit_length() - 1)), True)
        fwd = special_fft_groups(S, min(s2c_groups, max(1, S.bit_length() - 1)), False)
        # intermediate results keep the re-declared scale q0; the last group lands on delta.  The constant
        # is spread over the groups so that every encoded diagonal has entries of order one.
        self.c2s, lvl = [], 1
        cst = norm ** (1.0 / len(inv))
        for i, g in enumerate(inv):
            last = i + 1 == len(inv)
            scale = ev.q_last(lvl) * (delta / q0 if last else 1.0)
            self.c2s.append(DiagonalTransform(ev, {d: v * cst for d, v in g.items()}, S, lvl, scale))
            lvl += 1
        self.evalmod_level = lvl
        sv, lvl = self._evalmod_out(lvl)
        self.s2c_level = lvl
        self.s2c = []
        cst = ((q0 / self.msg_scale) / (2.0 * np.pi)) ** (1.0 / len(fwd))
        for i, g in enumerate(fwd):
            scale = ev.q_last(lvl) * (self.msg_scale / sv if i == 0 else 1.0)
            self.s2c.append(DiagonalTransform(ev, {d: v * cst for d, v in g.items()}, S, lvl, scale))
            lvl += 1
        self.out_level = lvl

    def _init_slim(self, norm, c2s_groups, s2c_groups):
        """SlotToCoeff FIRST, on the last levels of the chain (a few limbs: nearly free), then modulus raise,
        CoeffToSlot and EvalMod, whose output already holds the message in the slots (Chen, Chillotti, Song 2019).
        Against the usual order: SlotToCoeff at 2-3 limbs instead of ~18, and the EvalMod error reaches the slots
        directly instead of through a transform that spreads it over sqrt(s) coefficients.  The factor
        q0 / (2 pi msg_scale) that the usual order folds into the SlotToCoeff diagonals is a re-declaration of the
        scale here, and the scale delta_out on which the Chebyshev polynomial lands (any exact target costs nothing:
        its constants are encoded accordingly) is CHOSEN so that this re-declared scale equals the input scale:
        delta_out^(2^r) / prod q_i^(2^(r-i)) = q0 / (2 pi) over the primes q_i of the r double-angle levels.  The
        powers of the polynomial's argument stay at the nominal scale delta (close to the EvalMod primes)."""
        ev, S, r = self.ev, self.S, self.r
        q0 = ev.q[0]
        inv = special_fft_groups(S, min(c2s_groups, max(1, S.bit_length() - 1)), True)
        fwd = special_fft_groups(S, min(s2c_groups, max(1, S.bit_length() - 1)), False)
        self.evalmod_level = 1 + len(inv)
        lvl0 = self.evalmod_level + self.cheb.depth
        log_delta = math.log2(q0 / (2.0 * np.pi))
        for i in range(r):
            log_delta += (1 << (r - 1 - i)) * math.log2(ev.q_last(lvl0 + i))
        self.delta_out = 2.0 ** (log_delta / (1 << r))
        delta = self.delta
        self.c2s, lvl = [], 1
        cst = norm ** (1.0 / len(inv))
        for i, g in enumerate(inv):
            last = i + 1 == len(inv)
            scale = ev.q_last(lvl) * (delta / q0 if last else 1.0)
            self.c2s.append(DiagonalTransform(ev, {d: v * cst for d, v in g.items()}, S, lvl, scale))
            lvl += 1
        sv, self.out_level = self._evalmod_out(self.evalmod_level)
        assert abs(sv * 2.0 * np.pi / q0 - 1.0) < 1e-9
        self.sv = sv
        # the forward transform on the last len(fwd) levels, scale-preserving
        self.pre_levels = len(fwd)
        self.s2c_level = ev.bottom - len(fwd)
        self.s2c = [DiagonalTransform(ev, g, S, self.s2c_level + i, ev.q_last(self.s2c_level + i)) for i, g in enumerate(fwd)]

    def levels_needed(self):
        return self.out_level

    def eval_mod(self, w, ev=None):
        """w in [-1, 1] real slots, w = (m/q0 + I)/Kb  ->  sin(2 pi (m/q0 + I)) ~ 2 pi m / q0"""
        ev = self.ev if ev is None else ev
        cheb = self.cheb if ev is self.ev else ChebyshevEvaluator(ev, self.cheb.c, self.cheb.l)
        c = cheb(w, self.delta_out)
        for _ in range(self.r):
            c = ev.square_double_minus_one(c)
        return c

    def bootstrap(self, ct):
        ev = self.ev
        in_scale = ct.scale
        q0 = ev.q[0]
        if self.slim:
            assert ct.level <= self.s2c_level, "the slim order spends its SlotToCoeff levels before the modulus raise"
            ct = ev.drop(ct, self.s2c_level)
            for lt in self.s2c:
                ct = lt.apply(ct)
        else:
            assert abs(in_scale / self.msg_scale - 1.0) < 0.01, "the bootstrapper was set up for ciphertexts of scale msg_scale"
        t = ev.mod_raise(ev.drop(ct, ev.bottom))
        t.scale = float(q0)                                   # src/boot/Bootstrapper.cu:2867-2868
        k = self.S
        while k < ev.n // 2:                                  # SubSum onto the S-slot subring
            t = ev.add(t, ev.rotate(t, k))
            k <<= 1
        half = t
        for lt in self.c2s:
            half = lt.apply(half)
        cj = ev.conj(half)
        if self.packed:
            v = self.eval_mod(ev.add(half, cj))               # slots [0,s): x_k / Kb, [s,2s): x_{k+s} / Kb
        else:                                                 # half: (x_k + i x_{k+s}) / (2 Kb)
            re = ev.add(half, cj)
            im = ev.neg(ev.mul_i(ev.sub(half, cj)))
            # two independent EvalMods: side by side where the evaluator can (two streams on the GPU)
            a, b = ev.run_pair(lambda e: self.eval_mod(re, e), lambda e: self.eval_mod(im, e))
            v = ev.add(a, ev.mul_i(b))                            # 2 pi (m_k + i m_{k+s}) / q0
        if self.slim:
            v.scale = v.scale * 2.0 * np.pi * in_scale / q0       # = in_scale by the choice of delta
            return v
        out = v
        for lt in self.s2c:
            out = lt.apply(out)                           # slots: message * in_scale / delta
        out.scale = out.scale * in_scale / self.msg_scale         # src/boot/Bootstrapper.cu:2933 (a ratio of ~1)
        return out
