"""Plaintext-ring simulator with the interface of phantom_fhe_amd.ckks.GpuEvaluator -- TEST INFRASTRUCTURE.

`SimEvaluator` works on *plaintext polynomials* (float64 coefficient vectors, negacyclic products by FFT): no
encryption, no noise, but the same ring, automorphisms, levels, scales and modulus raising, so the algebra and the scale
bookkeeping of circuits built on the evaluator interface (bootstrap.py, cnn.py) can be tested on a CPU without the HIP
library.  It is a test double: only tests/ and tools/ import it, the product package never does.
"""
import numpy as np

from phantom_fhe_amd.ckks import Ct, EvaluatorBase, Pt


class SimEvaluator(EvaluatorBase):
    """Plaintext polynomials as float64 coefficient vectors: the ring, the automorphisms, levels and
    scales of the real thing; no encryption and no noise.  mod_raise adds q0 * I(X) with a random
    small integer polynomial I, which is what raising the modulus of a real ciphertext does to its
    decryption (I = (c0 + c1*s - m) / q0 over the integers)."""

    def __init__(self, log_n, q_primes, sparse_slots, hamming_weight=32, seed=0):
        super().__init__(log_n, q_primes, sparse_slots)
        self.rng = np.random.default_rng(seed)
        self.h = hamming_weight
        self.last_I = None

    def add_rotation_keys(self, steps, conj=True):
        pass

    # encode / decode by the definition of the embedding
    _matrices = {}

    @staticmethod
    def _matrix(s):
        mat = SimEvaluator._matrices.get(s)
        if mat is None:
            m = 4 * s
            rot = np.array([pow(5, j, m) for j in range(s)], dtype=np.int64)
            e = (rot[:, None] * np.arange(s, dtype=np.int64)[None, :]) % m
            mat = SimEvaluator._matrices[s] = np.exp(2j * np.pi * e / m)
        return mat

    def encode(self, values, scale, level, sparse_slots=None):
        s = self.s if sparse_slots is None else sparse_slots
        z = np.zeros(s, dtype=np.complex128)
        z[:len(values)] = values
        u = self._matrix(s).conj().T @ z / s
        gap = self.n // (2 * s)
        c = np.zeros(self.n)
        c[np.arange(s) * gap] = np.rint(scale * u.real)
        c[(np.arange(s) + s) * gap] = np.rint(scale * u.imag)
        return Pt(c, level, scale)

    def encrypt(self, pt):
        return Ct(pt.data.copy(), pt.level, pt.scale)

    def decrypt_decode(self, a, sparse_slots=None):
        s = self.s if sparse_slots is None else sparse_slots
        gap = self.n // (2 * s)
        u = (a.data[np.arange(s) * gap] + 1j * a.data[(np.arange(s) + s) * gap]) / a.scale
        return self._matrix(s) @ u

    def _same(self, a, b):
        assert a.level == b.level, (a.level, b.level)
        assert abs(a.scale / b.scale - 1.0) < 1e-9, (a.scale, b.scale)

    def add(self, a, b):
        self._same(a, b)
        return Ct(a.data + b.data, a.level, a.scale)

    def sub(self, a, b):
        self._same(a, b)
        return Ct(a.data - b.data, a.level, a.scale)

    def neg(self, a):
        return Ct(-a.data, a.level, a.scale)

    def add_plain(self, a, p):
        self._same(a, p)
        return Ct(a.data + p.data, a.level, a.scale)

    def _negacyclic(self, x, y):
        n = self.n
        f = np.fft.rfft(x, 2 * n) * np.fft.rfft(y, 2 * n)
        full = np.fft.irfft(f, 2 * n)
        return full[:n] - full[n:]

    def mul_plain(self, a, p):
        assert a.level == p.level
        return Ct(self._negacyclic(a.data, p.data), a.level, a.scale * p.scale)

    def mul(self, a, b):
        assert a.level == b.level
        return Ct(self._negacyclic(a.data, b.data), a.level, a.scale * b.scale)

    def rescale(self, a):
        q = self.q_last(a.level)
        return Ct(a.data / q, a.level + 1, a.scale / q)

    def drop(self, a, level):
        assert level >= a.level
        return Ct(a.data, level, a.scale)

    def _automorphism(self, c, g):
        n = self.n
        idx = (np.arange(n) * g) % (2 * n)
        out = np.zeros(n)
        sign = np.where(idx >= n, -1.0, 1.0)
        out[idx % n] = sign * c
        return out

    def rotate_many(self, a, steps):
        return [Ct(self._automorphism(a.data, pow(5, st % (self.n // 2), 2 * self.n)), a.level, a.scale) for st in steps]

    def conj(self, a):
        return Ct(self._automorphism(a.data, 2 * self.n - 1), a.level, a.scale)

    def mul_i(self, a):
        h = self.n // 2
        return Ct(np.concatenate([-a.data[h:], a.data[:h]]), a.level, a.scale)

    def mod_raise(self, a):
        assert a.level == self.bottom
        sigma = np.sqrt((self.h + 1) / 12.0)
        self.last_I = np.rint(self.rng.normal(0.0, sigma, self.n))
        return Ct(a.data + self.q[0] * self.last_I, 1, a.scale)
