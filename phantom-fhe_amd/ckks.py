"""Scheme-level CKKS evaluator over the C ABI.

The reference drives its kernels from C++ host classes (`CKKSEvaluator`, include/boot/ckks_evaluator.cuh;
`Bootstrapper`, src/boot/Bootstrapper.cu).  This module is the corresponding host orchestration for
this repository: ciphertext objects that carry (level, scale), the evaluator operations the
bootstrapper needs, and nothing that computes on polynomials itself — every operation of
`GpuEvaluator` is one or a few C-ABI calls (`phantom_fhe_amd.PhantomContext`).

The CPU test double with the same interface (plaintext polynomials, no encryption) lives with the rest of the test
infrastructure in oracle/sim_evaluator.py; nothing in this package uses it.

Levels are the reference's chain indices: 1 is the top data level (all of Q), each rescale adds one.
"""
import numpy as np


class Ct:
    """Ciphertext handle: data (device tensor [2][l][N], or a coefficient vector in the simulator)."""
    __slots__ = ("data", "level", "scale")

    def __init__(self, data, level, scale):
        self.data, self.level, self.scale = data, level, float(scale)


class Pt:
    __slots__ = ("data", "level", "scale")

    def __init__(self, data, level, scale):
        self.data, self.level, self.scale = data, level, float(scale)


class EvaluatorBase:
    """Shared bookkeeping; back ends implement the underscore-free primitives."""

    def __init__(self, log_n, q_primes, sparse_slots):
        self.log_n, self.n = log_n, 1 << log_n
        self.q = [int(v) for v in q_primes]
        self.s = sparse_slots
        self.bottom = len(self.q)          # chain index of the one-limb level
        self._consts = {}

    def limbs(self, level):
        return len(self.q) - (level - 1)

    def q_last(self, level):
        """the prime a rescale at `level` divides by"""
        return self.q[self.limbs(level) - 1]

    # ---- derived operations -----------------------------------------------------------------
    def const_pt(self, c, scale, level):
        # a constant vector is the constant polynomial: two slots are enough.  Cached: a circuit asks for
        # the same constants on every run, and an encode synchronises the stream.
        key = (complex(c), float(scale), level)
        pt = self._consts.get(key)
        if pt is None:
            pt = self._consts[key] = self.encode(np.array([c, c], dtype=np.complex128), scale, level, 2)
        return pt

    def add_const(self, a, c):
        return self.add_plain(a, self.const_pt(c, a.scale, a.level))

    def mul_const(self, a, c, const_scale):
        return self.mul_plain(a, self.const_pt(c, const_scale, a.level))

    def double(self, a):
        return self.add(a, a)

    def mul_plain_add(self, acc, a, p):
        """acc + a * p (back ends may fuse it)"""
        t = self.mul_plain(a, p)
        return t if acc is None else self.add(acc, t)

    def to_level_scale(self, a, level, scale):
        """Exact (level, scale): one multiplication by the constant 1 encoded at the scale that makes
        the rescaled product land on `scale`.  Needs a.level < level."""
        assert a.level < level, (a.level, level)
        a = self.drop(a, level - 1)
        return self.rescale(self.mul_const(a, 1.0, scale * self.q_last(level - 1) / a.scale))

    def rotate(self, a, step):
        return self.rotate_many(a, [step])[0]

    # ---- independent ciphertexts of one (level, scale) moved together (the q partial sums of a convolution) ---------
    def batch(self, cts):
        """one handle for several ciphertexts: `add` and `rotate_batch` act on all of them in one launch sequence"""
        import numpy as _np
        return Ct(_np.stack([c.data for c in cts]), cts[0].level, cts[0].scale)

    def unbatch(self, b):
        return [Ct(d, b.level, b.scale) for d in b.data]

    def rotate_batch(self, b, step):
        import numpy as _np
        return Ct(_np.stack([self.rotate(c, step).data for c in self.unbatch(b)]), b.level, b.scale)

    # ---- two independent sub-circuits (the real and imaginary EvalMod of a full-slot bootstrap) -------------
    def run_pair(self, fa, fb):
        """(fa(ev_a), fb(ev_b)) for two evaluators that may run concurrently; here one after the other"""
        return fa(self), fb(self)

    # ---- baby-step / giant-step sums (the linear transforms of bootstrap.py) ----------------------------
    def encode_lt(self, values, scale, level, sparse_slots):
        """a diagonal of a linear transform, in the form `bsgs` consumes"""
        return self.encode(values, scale, level, sparse_slots)

    def bsgs(self, a, baby_steps, giant_steps, plains):
        """sum_g rot_G( sum_b plains[g][b] * rot_B(a) ); plains[g][b] may be None.  Step by step here: the baby
        rotations share one modup (rotate_many), every rotation ends in its own moddown."""
        rot = self.rotate_many(a, baby_steps)
        total = None
        for g, row in zip(giant_steps, plains):
            inner = None
            for r, pt in zip(rot, row):
                if pt is not None:
                    inner = self.mul_plain_add(inner, r, pt)
            if inner is None:
                continue
            if g % (self.n // 2):
                inner = self.rotate(inner, g)
            total = inner if total is None else self.add(total, inner)
        return total

    def mul_rescale(self, a, b, times2=False):
        """rescale(a * b) (times 2): back ends may merge relinearisation and rescale into one division"""
        t = self.mul(a, b)
        return self.rescale(self.double(t) if times2 else t)

    def mul_rescale_batch(self, pairs, times2=False):
        """[rescale(a * b) for a, b in pairs], all at one level: back ends may run them as one launch sequence"""
        return [self.mul_rescale(a, b, times2) for a, b in pairs]

    def square_double_minus_one(self, a):
        """2a^2 - 1 (a Chebyshev doubling / a cosine double angle), one level"""
        return self.add_const(self.mul_rescale(a, a, times2=True), -1.0)


# =================================================================================================
# GPU back end
# =================================================================================================
class GpuEvaluator(EvaluatorBase):
    """Every primitive is a C-ABI call on `ctx` (phantom_fhe_amd.PhantomContext).  Keys are generated
    on the GPU from `sk` for the rotations announced with `add_rotation_keys`."""

    def __init__(self, pha, ctx, sk, sparse_slots, seed=None, double_hoisting=True):
        super().__init__(ctx.log_n, ctx.moduli[:ctx.size_Q], sparse_slots)
        self.pha, self.ctx, self.sk = pha, ctx, sk
        self.double_hoisting = double_hoisting
        self.concurrent_pairs = __import__("os").environ.get("PHA_CONCURRENT_PAIRS", "1") != "0"
        self.merged_rescale = __import__("os").environ.get("PHA_MERGED_RESCALE", "1") != "0"
        # double hoisting makes a baby step (one inner product) several times cheaper than a giant step (moddown +
        # modup + inner product): bootstrap.DiagonalTransform shifts its split towards more baby steps by 2^baby_bias
        # (measured at N = 2^16 full slots, 4 levels per transform: bias 0 / 1 / 2 / 3 = 31.3 / 30.5 / 29.9 / 31.0 ms with
        # 46 / 46 / 75 / 77 rotation keys)
        self.baby_bias = int(__import__("os").environ.get("PHA_BABY_BIAS", "1")) if double_hoisting else 0
        self._seed = seed
        self.rlk = ctx.gen_relinkey(sk, self._next_seed())
        self.galois = {}
        self._mono = {}
        self.auto_keys = False               # generate a missing rotation key on first use (cnn.py: hundreds of steps)

    def _next_seed(self):
        """nonce of the next draw: OS entropy, or consecutive numbers from `seed` when one was given (reproducible runs
        under a fixed sampler key: tests)"""
        if self._seed is None:
            return self.pha.random_nonce()
        self._seed += 1
        return self._seed

    def elt(self, step):
        return self.pha.galois_elt_from_step(self.log_n, step)

    def add_rotation_keys(self, steps, conj=True):
        elts = {self.elt(s) for s in steps if s % (self.n // 2)}
        if conj:
            elts.add(2 * self.n - 1)
        for e in sorted(elts):
            if e not in self.galois:
                self.galois[e] = self.ctx.gen_galois_key(self.sk, e, self._next_seed())

    # ---- encode / encrypt ---------------------------------------------------------------------
    def encode(self, values, scale, level, sparse_slots=None):
        s = self.s if sparse_slots is None else sparse_slots
        return Pt(self.ctx.encode(values, scale, level, s), level, scale)

    def encode_lt(self, values, scale, level, sparse_slots):
        if not self.double_hoisting:
            return self.encode(values, scale, level, sparse_slots)
        return Pt(self.ctx.encode_qp(values, scale, level, sparse_slots), level, scale)       # over Q_l u P

    def bsgs(self, a, baby_steps, giant_steps, plains):
        """one C-ABI call: pha_linear_transform_double_hoisted (one modup for the babies, one moddown per giant step)"""
        if not self.double_hoisting:
            return super().bsgs(a, baby_steps, giant_steps, plains)
        if self.auto_keys:
            self.add_rotation_keys(list(baby_steps) + list(giant_steps), conj=False)
        half = self.n // 2
        scale = next(p.scale for row in plains for p in row if p is not None)
        b_elts = [self.elt(s) if s % half else 1 for s in baby_steps]
        g_elts = [self.elt(s) if s % half else 1 for s in giant_steps]
        out = self.ctx.linear_transform_double_hoisted(
            a.level, a.data, b_elts, [self.galois[e] if e != 1 else None for e in b_elts], g_elts,
            [self.galois[e] if e != 1 else None for e in g_elts], [[None if p is None else p.data for p in row] for row in plains])
        return Ct(out, a.level, a.scale * scale)

    def encrypt(self, pt):
        return Ct(self.ctx.encrypt_symmetric(self.sk, pt.data, pt.level, self._next_seed(), self._next_seed()), pt.level, pt.scale)

    def decrypt_decode(self, a, sparse_slots=None):
        s = self.s if sparse_slots is None else sparse_slots
        return self.ctx.decode(self.ctx.ckks_decrypt(self.sk, a.data, a.level), a.scale, s, a.level)

    # ---- arithmetic -------------------------------------------------------------------------
    def _same(self, a, b):
        assert a.level == b.level, (a.level, b.level)
        assert abs(a.scale / b.scale - 1.0) < 1e-9, (a.scale, b.scale)

    def _binary(self, fn, a_data, b_data, polys, b_polys):
        import torch
        d = torch.empty_like(a_data)
        rc = fn(self.ctx._h, d.data_ptr(), a_data.data_ptr(), b_data.data_ptr(), a_data.shape[-2], polys, b_polys,
                self.pha._stream())
        self.pha._check(rc)
        return d

    def add(self, a, b):
        self._same(a, b)
        polys = a.data.numel() // (a.data.shape[-2] * a.data.shape[-1])        # 2, or 2 * batch
        assert a.data.shape == b.data.shape
        return Ct(self._binary(self.pha.lib().pha_add_rns_poly, a.data, b.data, polys, polys), a.level, a.scale)

    def batch(self, cts):
        import torch
        return Ct(torch.stack([c.data for c in cts]), cts[0].level, cts[0].scale)

    def rotate_batch(self, b, step):
        """pha_apply_galois_batched: one rotation of every ciphertext of the batch, kernels spanning the batch"""
        if step % (self.n // 2) == 0:
            return b
        if self.auto_keys:
            self.add_rotation_keys([step], conj=False)
        e = self.elt(step)
        return Ct(self.ctx.apply_galois_batched(b.level, b.data, e, self.galois[e]), b.level, b.scale)

    def sub(self, a, b):
        self._same(a, b)
        return Ct(self._binary(self.pha.lib().pha_sub_rns_poly, a.data, b.data, 2, 2), a.level, a.scale)

    def neg(self, a):
        import torch
        d = torch.empty_like(a.data)
        self.pha._check(self.pha.lib().pha_negate_rns_poly(self.ctx._h, d.data_ptr(), a.data.data_ptr(), a.data.shape[-2], 2,
                                                           self.pha._stream()))
        return Ct(d, a.level, a.scale)

    def add_plain(self, a, p):
        self._same(a, p)
        d = a.data.clone()
        self.ctx.add_plain_inplace(d, p.data)
        return Ct(d, a.level, a.scale)

    def mul_plain(self, a, p):
        assert a.level == p.level
        return Ct(self._binary(self.pha.lib().pha_multiply_rns_poly, a.data, p.data, 2, 1), a.level, a.scale * p.scale)

    def mul_plain_add(self, acc, a, p):
        """acc + a * p in one pass (acc may be None); scales must agree"""
        if acc is None:
            return self.mul_plain(a, p)
        import torch
        assert a.level == p.level == acc.level and abs(acc.scale / (a.scale * p.scale) - 1.0) < 1e-9
        d = torch.empty_like(a.data)
        self.pha._check(self.pha.lib().pha_multiply_add_rns_poly(self.ctx._h, d.data_ptr(), a.data.data_ptr(), p.data.data_ptr(),
                                                                 acc.data.data_ptr(), a.data.shape[-2], 2, 1, self.pha._stream()))
        return Ct(d, a.level, acc.scale)

    def mul(self, a, b):
        import torch
        assert a.level == b.level
        d = torch.empty((3,) + tuple(a.data.shape[1:]), dtype=a.data.dtype, device=a.data.device)
        d[:2].copy_(a.data)                    # the tensor product writes all three polys
        self.ctx.multiply_and_relin_inplace(a.level, d, b.data, self.rlk)
        return Ct(d[:2], a.level, a.scale * b.scale)

    def mul_rescale(self, a, b, times2=False):
        """one C-ABI call: pha_multiply_relin_rescale (the key-switched part and the lifted tensor polys are divided by
        P * q_last in a single moddown)"""
        if not self.merged_rescale:
            return super().mul_rescale(a, b, times2)
        assert a.level == b.level
        out = self.ctx.multiply_relin_rescale(a.level, a.data, b.data, self.rlk, times2)
        return Ct(out, a.level + 1, a.scale * b.scale / self.q_last(a.level))

    def mul_rescale_batch(self, pairs, times2=False):
        """one C-ABI call: pha_multiply_relin_rescale_batched"""
        if not self.merged_rescale or len(pairs) < 2:
            return super().mul_rescale_batch(pairs, times2)
        level = pairs[0][0].level
        assert all(a.level == level and b.level == level for a, b in pairs)
        out = self.ctx.multiply_relin_rescale_batched(level, [a.data for a, _ in pairs], [b.data for _, b in pairs], self.rlk, times2)
        q = self.q_last(level)
        return [Ct(out[i], level + 1, a.scale * b.scale / q) for i, (a, b) in enumerate(pairs)]

    def rescale(self, a):
        import torch
        out = torch.empty((2, self.limbs(a.level) - 1, self.n), dtype=torch.int64, device=a.data.device)
        self.ctx.rescale_to_next(a.level, a.data, 2, out)
        return Ct(out, a.level + 1, a.scale / self.q_last(a.level))

    def drop(self, a, level):
        assert level >= a.level
        if level == a.level:
            return a
        # CKKS mod_switch_to_next only drops limbs (src/evaluate.cu:618-700): one strided copy for all levels
        return Ct(a.data[:, :self.limbs(level)].contiguous(), level, a.scale)

    def rotate_many(self, a, steps):
        half = self.n // 2
        live = [s for s in steps if s % half]
        outs = {}
        if live:
            if self.auto_keys:
                self.add_rotation_keys(live, conj=False)
            elts = [self.elt(s) for s in live]
            res = self.ctx.rotate_hoisted(a.level, a.data, elts, [self.galois[e] for e in elts])
            outs = {s: Ct(r, a.level, a.scale) for s, r in zip(live, res)}
        return [outs[s] if s % half else a for s in steps]

    def conj(self, a):
        d = a.data.clone()
        self.ctx.apply_galois_inplace(a.level, d, 2 * self.n - 1, self.galois[2 * self.n - 1])
        return Ct(d, a.level, a.scale)

    def mul_i(self, a):
        """multiplication of every slot by i = multiplication of the polynomial by X^(N/2): exact"""
        key = a.level
        if key not in self._mono:
            l = self.limbs(a.level)
            h = np.zeros((l, self.n), dtype=np.uint64)
            h[:, self.n // 2] = 1
            m = self.pha.to_device(h)
            self.ctx.nwt_2d_radix8_forward_inplace(m, l, 0)
            self._mono[key] = m
        d = a.data.clone()
        self.ctx.multiply_plain_inplace(d, self._mono[key])
        return Ct(d, a.level, a.scale)

    def mod_raise(self, a):
        assert a.level == self.bottom
        return Ct(self.ctx.mod_raise(a.data), 1, a.scale)

    # ---- a second (context, stream) pair for independent sub-circuits ---------------------------------------
    def sibling(self, nest=0):
        """An evaluator with the same keys on its own library context (own workspace) and HIP stream: two
        independent sub-circuits at 15-30 limbs each fill less than half of the GPU, so running them side by
        side nearly halves their time (measured: the two EvalMods of a full-slot bootstrap, 16.6 -> ... ms).
        One sibling per nesting depth of run_pair, so that nested pairs form a tree of streams."""
        sibs = self.__dict__.setdefault("_siblings", {})
        if nest not in sibs:
            import torch
            sib = GpuEvaluator.__new__(GpuEvaluator)
            EvaluatorBase.__init__(sib, self.log_n, self.q, self.s)
            sib.pha, sib.sk, sib.rlk, sib.galois = self.pha, self.sk, self.rlk, self.galois
            sib.ctx = self.pha.PhantomContext(self.ctx.log_n, self.ctx.moduli, self.ctx.size_P)
            sib._seed, sib._mono, sib.auto_keys = (None if self._seed is None else self._seed + ((len(sibs) + 1) << 20)), {}, False
            sib.double_hoisting, sib.baby_bias = self.double_hoisting, self.baby_bias
            sib.merged_rescale, sib.concurrent_pairs = self.merged_rescale, self.concurrent_pairs
            sib.stream = torch.cuda.Stream()
            sib._nest = 0
            sibs[nest] = sib
        return sibs[nest]

    MAX_NEST = 2          # run_pair inside run_pair: up to four streams

    def run_pair(self, fa, fb):
        import torch
        nest = getattr(self, "_nest", 0)
        if not self.concurrent_pairs or nest >= self.MAX_NEST:
            return fa(self), fb(self)
        sib, main = self.sibling(nest), torch.cuda.current_stream()
        sib.stream.wait_stream(main)
        self._nest, sib._nest = nest + 1, nest + 1
        try:
            with torch.cuda.stream(sib.stream):
                b = fb(sib)
            a = fa(self)
        finally:
            self._nest = nest
        main.wait_stream(sib.stream)
        for c in (b if isinstance(b, (list, tuple)) else [b]):
            if isinstance(c, Ct):
                c.data.record_stream(main)  # allocated on the sibling's stream, consumed on this one
        return a, b
