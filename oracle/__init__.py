"""ctypes binding of the CPU oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_sz = ctypes.c_size_t
_int = ctypes.c_int
_vp = ctypes.c_void_p
_u64p = ctypes.POINTER(_u64)
_u32p = ctypes.POINTER(_u32)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        sig = {
            "orc_powmod": (_u64, [_u64, _u64, _u64]),
            "orc_invmod": (_u64, [_u64, _u64]),
            "orc_shoup": (_u64, [_u64, _u64]),
            "orc_barrett_ratio": (None, [_u64, _u64p]),
            "orc_is_prime": (_int, [_u64]),
            "orc_get_primes": (_int, [_u64, _int, _sz, _u64p]),
            "orc_coeff_modulus_create": (_int, [_u64, ctypes.POINTER(_int), _sz, _u64p]),
            "orc_min_primitive_root": (_int, [_u64, _u64, _u64p]),
            "orc_ntt_tables": (_int, [_int, _u64, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p]),
            "orc_ctx_create": (_vp, [_int, _u64p, _sz, _sz]),
            "orc_ctx_free": (None, [_vp]),
            "orc_ctx_table": (_u64p, [_vp, _int]),
            "orc_ntt_fwd": (None, [_vp, _u64p, _sz]),
            "orc_ntt_inv": (None, [_vp, _u64p, _sz]),
            "orc_ntt_fwd_naive": (None, [_vp, _u64p, _u64p, _sz]),
            "orc_nwt_forward_inplace": (None, [_vp, _u64p, _sz, _sz]),
            "orc_nwt_backward_inplace": (None, [_vp, _u64p, _sz, _sz]),
            "orc_nwt_backward": (None, [_vp, _u64p, _u64p, _sz, _sz]),
            "orc_nwt_backward_scale": (None, [_vp, _u64p, _u64p, _sz, _sz, _u64p]),
            "orc_nwt_forward_inplace_special": (None, [_vp, _u64p, _sz, _sz, _sz, _sz, _sz, _sz]),
            "orc_nwt_backward_inplace_special": (None, [_vp, _u64p, _sz, _sz, _sz, _sz]),
            "orc_nwt_forward_fuse_moddown": (None, [_vp, _u64p, _u64p, _u64p, _u64p, _sz, _sz]),
            "orc_level_constants": (None, [_vp, _sz, _u64p, _u64p, _u64p, _u64p]),
            "orc_modup": (None, [_vp, _sz, _u64p, _u64p]),
            "orc_key_switch_inner_prod": (None, [_vp, _sz, _u64p, _u64p, ctypes.POINTER(_u64p)]),
            "orc_moddown_from_ntt": (None, [_vp, _sz, _u64p, _u64p]),
            "orc_add_to_ct": (None, [_vp, _sz, _u64p, _u64p]),
            "orc_keyswitch_inplace": (None, [_vp, _sz, _u64p, _u64p, ctypes.POINTER(_u64p)]),
            "orc_divide_and_round_q_last_ntt": (None, [_vp, _sz, _u64p, _sz, _u64p]),
            "orc_galois_table": (None, [_int, _u32, _u32p]),
            "orc_apply_galois_ntt": (None, [_int, _u64p, _sz, _u32, _u64p]),
            "orc_galois_elt_from_step": (_u32, [_int, _int]),
            "orc_apply_galois_coeff": (None, [_vp, _u64p, _sz, _u32, _u64p]),
            "orc_tensor_prod_2x2": (None, [_vp, _u64p, _u64p, _sz, _u64p]),
            "orc_tensor_square_2x2": (None, [_vp, _u64p, _sz, _u64p]),
            "orc_tensor_prod_mxn": (None, [_vp, _u64p, _sz, _u64p, _sz, _sz, _u64p]),
            "orc_pointwise_mul": (None, [_vp, _u64p, _u64p, _u64p, _sz, ctypes.POINTER(_sz)]),
            "orc_rnd_at": (_u64, [_u64, _u64]),
            "orc_set_sampler_key": (None, [ctypes.c_char_p]),
            "orc_chacha20_block": (None, [ctypes.POINTER(ctypes.c_uint32), _u64, _u64, ctypes.POINTER(ctypes.c_uint32)]),
            "orc_sample_ternary": (None, [_vp, _u64p, _sz, ctypes.POINTER(_sz), _u64]),
            "orc_sample_error": (None, [_vp, _u64p, _sz, ctypes.POINTER(_sz), _u64]),
            "orc_sample_uniform": (None, [_vp, _u64p, _sz, ctypes.POINTER(_sz), _u64]),
            "orc_gen_secretkey": (None, [_vp, _u64, _u64p]),
            "orc_encrypt_zero_symmetric": (None, [_vp, _u64p, _sz, _u64, _u64, _u64p]),
            "orc_encrypt_symmetric": (None, [_vp, _u64p, _u64p, _sz, _u64, _u64, _u64p]),
            "orc_gen_publickey": (None, [_vp, _u64p, _u64, _u64p]),
            "orc_encrypt_zero_asymmetric": (None, [_vp, _u64p, _sz, _u64, _u64p]),
            "orc_encrypt_asymmetric": (None, [_vp, _u64p, _u64p, _sz, _u64, _u64p]),
            "orc_ckks_decrypt": (None, [_vp, _u64p, _u64p, _sz, _sz, _u64p]),
            "orc_generate_one_kswitch_key": (None, [_vp, _u64p, _u64p, _u64, ctypes.POINTER(_u64p)]),
            "orc_gen_relinkey": (None, [_vp, _u64p, _u64, ctypes.POINTER(_u64p)]),
            "orc_gen_galois_key": (None, [_vp, _u64p, _u32, _u64, ctypes.POINTER(_u64p)]),
            "orc_rotate_hoisted": (None, [_vp, _sz, _u64p, _sz, _u32p, ctypes.POINTER(ctypes.POINTER(_u64p)),
                                          ctypes.POINTER(_u64p)]),
            "orc_fnv1a64": (_u64, [_u64p, _sz]),
            "orc_fill_uniform": (None, [_u64p, _sz, _u64, _u64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def coeff_modulus_create(n, bit_sizes):
    bits = (_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    if lib().orc_coeff_modulus_create(n, bits, len(bit_sizes), p(out)):
        raise RuntimeError("failed to find enough qualifying primes")
    return out


def fill_uniform(count, q, seed):
    out = np.zeros(count, dtype=np.uint64)
    lib().orc_fill_uniform(p(out), count, int(q), int(seed))
    return out


def fnv1a64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return int(lib().orc_fnv1a64(p(a), a.size))


class Oracle:
    """Key-level context Q ∪ P (last size_p moduli are P)."""

    def __init__(self, log_n, moduli_qp, size_p=1):
        self.log_n = log_n
        self.n = 1 << log_n
        self.moduli = np.ascontiguousarray(moduli_qp, dtype=np.uint64)
        self.size_QP = len(self.moduli)
        self.size_P = size_p
        self.size_Q = self.size_QP - size_p
        self._h = lib().orc_ctx_create(log_n, p(self.moduli), self.size_QP, size_p)
        if not self._h:
            raise ValueError("invalid modulus in try_minimal_primitive_root")

    def __del__(self):
        try:
            if self._h:
                lib().orc_ctx_free(self._h)
                self._h = None
        except Exception:
            pass

    def table(self, which, limb=None):
        ptr = lib().orc_ctx_table(self._h, which)
        if which < 4:
            a = np.ctypeslib.as_array(ptr, shape=(self.size_QP, self.n))
            return a.copy() if limb is None else a[limb].copy()
        return np.ctypeslib.as_array(ptr, shape=(self.size_QP,)).copy()

    def beta(self, size_ql):
        return -(-size_ql // self.size_P)

    def qlp_moduli(self, size_ql):
        return np.concatenate([self.moduli[:size_ql], self.moduli[self.size_Q:]])

    def uniform_poly(self, table_indices, seed):
        """[len(table_indices)][n] uniform residues; limb i is modulo moduli[table_indices[i]]."""
        out = np.zeros((len(table_indices), self.n), dtype=np.uint64)
        for i, t in enumerate(table_indices):
            out[i] = fill_uniform(self.n, self.moduli[t], seed * 1000003 + i)
        return out

    def ntt_fwd(self, x, table_idx):
        x = np.ascontiguousarray(x, dtype=np.uint64).copy()
        lib().orc_ntt_fwd(self._h, p(x), table_idx)
        return x

    def ntt_inv(self, x, table_idx):
        x = np.ascontiguousarray(x, dtype=np.uint64).copy()
        lib().orc_ntt_inv(self._h, p(x), table_idx)
        return x

    def ntt_fwd_naive(self, x, table_idx):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.zeros_like(x)
        lib().orc_ntt_fwd_naive(self._h, p(x), p(out), table_idx)
        return out

    def nwt_forward_inplace(self, buf, count, start=0):
        lib().orc_nwt_forward_inplace(self._h, p(buf), count, start)

    def nwt_backward_inplace(self, buf, count, start=0):
        lib().orc_nwt_backward_inplace(self._h, p(buf), count, start)

    def nwt_backward(self, out, inp, count, start=0):
        lib().orc_nwt_backward(self._h, p(out), p(inp), count, start)

    def nwt_backward_scale(self, out, inp, count, start, scale):
        lib().orc_nwt_backward_scale(self._h, p(out), p(inp), count, start, p(scale))

    def nwt_forward_inplace_special(self, buf, count, start, size_qp, size_p, excl_start=0, excl_end=0):
        lib().orc_nwt_forward_inplace_special(self._h, p(buf), count, start, size_qp, size_p, excl_start, excl_end)

    def nwt_backward_inplace_special(self, buf, count, start, size_qp, size_p):
        lib().orc_nwt_backward_inplace_special(self._h, p(buf), count, start, size_qp, size_p)

    def nwt_forward_fuse_moddown(self, ct, cx, pinv, delta, count, start=0):
        lib().orc_nwt_forward_fuse_moddown(self._h, p(ct), p(cx), p(pinv), p(delta), count, start)

    def level_constants(self, size_ql):
        outs = [np.zeros(max(size_ql - 1, 1), dtype=np.uint64)] + [np.zeros(size_ql, dtype=np.uint64) for _ in range(3)]
        lib().orc_level_constants(self._h, size_ql, *[p(o) for o in outs])
        return dict(inv_q_last_mod_q=outs[0][:size_ql - 1], bigP_mod_q=outs[1], bigPInv_mod_q=outs[2],
                    partQlHatInv_mod_Ql_concat=outs[3])

    @staticmethod
    def _evk_ptrs(evk):
        arr = (_u64p * len(evk))(*[p(e) for e in evk])
        return arr

    def modup(self, size_ql, cks):
        dst = np.zeros((self.beta(size_ql), size_ql + self.size_P, self.n), dtype=np.uint64)
        lib().orc_modup(self._h, size_ql, p(dst), p(np.ascontiguousarray(cks)))
        return dst

    def key_switch_inner_prod(self, size_ql, t_mod_up, evk):
        cx = np.zeros((2, size_ql + self.size_P, self.n), dtype=np.uint64)
        lib().orc_key_switch_inner_prod(self._h, size_ql, p(cx), p(np.ascontiguousarray(t_mod_up)), self._evk_ptrs(evk))
        return cx

    def moddown_from_ntt(self, size_ql, cx_i):
        cx_i = np.ascontiguousarray(cx_i).copy()
        ct = np.zeros((size_ql, self.n), dtype=np.uint64)
        lib().orc_moddown_from_ntt(self._h, size_ql, p(ct), p(cx_i))
        return ct

    def keyswitch_inplace(self, size_ql, ct, c2, evk):
        lib().orc_keyswitch_inplace(self._h, size_ql, p(ct), p(np.ascontiguousarray(c2)), self._evk_ptrs(evk))

    def divide_and_round_q_last_ntt(self, size_ql, src, cipher_size):
        src = np.ascontiguousarray(src).copy()
        dst = np.zeros((cipher_size, size_ql - 1, self.n), dtype=np.uint64)
        lib().orc_divide_and_round_q_last_ntt(self._h, size_ql, p(src), cipher_size, p(dst))
        return dst

    def apply_galois_ntt(self, operand, elt):
        operand = np.ascontiguousarray(operand)
        out = np.zeros_like(operand)
        lib().orc_apply_galois_ntt(self.log_n, p(operand), operand.shape[0], elt, p(out))
        return out

    def apply_galois_coeff(self, operand, elt):
        operand = np.ascontiguousarray(operand)
        out = np.zeros_like(operand)
        lib().orc_apply_galois_coeff(self._h, p(operand), operand.shape[0], elt, p(out))
        return out

    def tensor_prod_2x2(self, op1, op2):
        limbs = op1.shape[1]
        res = np.zeros((3, limbs, self.n), dtype=np.uint64)
        lib().orc_tensor_prod_2x2(self._h, p(np.ascontiguousarray(op1)), p(np.ascontiguousarray(op2)), limbs, p(res))
        return res

    def tensor_prod_mxn(self, op1, op2):
        """op1 [m][limbs][n], op2 [nn][limbs][n] -> [m + nn - 1][limbs][n] (src/polymath.cu:551-599)"""
        op1, op2 = np.ascontiguousarray(op1), np.ascontiguousarray(op2)
        limbs = op1.shape[1]
        res = np.zeros((op1.shape[0] + op2.shape[0] - 1, limbs, self.n), dtype=np.uint64)
        lib().orc_tensor_prod_mxn(self._h, p(op1), op1.shape[0], p(op2), op2.shape[0], limbs, p(res))
        return res

    def rotate_hoisted(self, size_ql, ct, elts, evks):
        """rotations of ONE ciphertext by several Galois elements sharing the modup (oracle.c)"""
        ct = np.ascontiguousarray(ct)
        outs = [np.zeros((2, size_ql, self.n), dtype=np.uint64) for _ in elts]
        e = np.ascontiguousarray(elts, dtype=np.uint32)
        keys = [self._evk_ptrs(k) for k in evks]
        karr = (ctypes.POINTER(_u64p) * len(keys))(*keys)
        oarr = (_u64p * len(outs))(*[p(o) for o in outs])
        lib().orc_rotate_hoisted(self._h, size_ql, p(ct), len(elts), e.ctypes.data_as(_u32p), karr, oarr)
        return outs

    # ---- key generation / encryption / decryption (oracle.c "Key generation") ---------------
    def level_limbs(self, chain_index):
        return self.size_QP if chain_index == 0 else self.size_Q - (chain_index - 1)

    def sample(self, kind, limbs, seed):
        out = np.zeros((limbs, self.n), dtype=np.uint64)
        getattr(lib(), "orc_sample_" + kind)(self._h, p(out), limbs, None, seed)
        return out

    def gen_secretkey(self, seed):
        sk = np.zeros((self.size_QP, self.n), dtype=np.uint64)
        lib().orc_gen_secretkey(self._h, seed, p(sk))
        return sk

    def encrypt_zero_symmetric(self, sk, chain_index, seed_a, seed_e):
        ct = np.zeros((2, self.level_limbs(chain_index), self.n), dtype=np.uint64)
        lib().orc_encrypt_zero_symmetric(self._h, p(sk), chain_index, seed_a, seed_e, p(ct))
        return ct

    def encrypt_symmetric(self, sk, plain, chain_index, seed_a, seed_e):
        ct = np.zeros((2, self.level_limbs(chain_index), self.n), dtype=np.uint64)
        lib().orc_encrypt_symmetric(self._h, p(sk), p(np.ascontiguousarray(plain)), chain_index, seed_a, seed_e, p(ct))
        return ct

    def gen_publickey(self, sk, seed):
        pk = np.zeros((2, self.size_QP, self.n), dtype=np.uint64)
        lib().orc_gen_publickey(self._h, p(sk), seed, p(pk))
        return pk

    def encrypt_zero_asymmetric(self, pk, chain_index, seed):
        ct = np.zeros((2, self.level_limbs(chain_index), self.n), dtype=np.uint64)
        lib().orc_encrypt_zero_asymmetric(self._h, p(np.ascontiguousarray(pk)), chain_index, seed, p(ct))
        return ct

    def encrypt_asymmetric(self, pk, plain, chain_index, seed):
        ct = np.zeros((2, self.level_limbs(chain_index), self.n), dtype=np.uint64)
        lib().orc_encrypt_asymmetric(self._h, p(np.ascontiguousarray(pk)), p(np.ascontiguousarray(plain)), chain_index, seed, p(ct))
        return ct

    def ckks_decrypt(self, sk, ct, chain_index):
        ct = np.ascontiguousarray(ct)
        plain = np.zeros((self.level_limbs(chain_index), self.n), dtype=np.uint64)
        lib().orc_ckks_decrypt(self._h, p(sk), p(ct), ct.shape[0], chain_index, p(plain))
        return plain

    def _new_evk(self):
        return [np.zeros((2, self.size_QP, self.n), dtype=np.uint64) for _ in range(self.size_Q // self.size_P)]

    def generate_one_kswitch_key(self, sk, new_key, seed):
        evk = self._new_evk()
        lib().orc_generate_one_kswitch_key(self._h, p(sk), p(np.ascontiguousarray(new_key)), seed, self._evk_ptrs(evk))
        return evk

    def gen_relinkey(self, sk, seed):
        evk = self._new_evk()
        lib().orc_gen_relinkey(self._h, p(sk), seed, self._evk_ptrs(evk))
        return evk

    def gen_galois_key(self, sk, elt, seed):
        evk = self._new_evk()
        lib().orc_gen_galois_key(self._h, p(sk), elt, seed, self._evk_ptrs(evk))
        return evk


def rnd_at(seed, ctr):
    return int(lib().orc_rnd_at(seed, ctr))


SAMPLER_KEY = bytes(32)   # the oracle's fixed sampler key; tests install the same key in the library


def set_sampler_key(key32):
    assert len(key32) == 32
    lib().orc_set_sampler_key(bytes(key32))


def chacha20_block(key_words, seed, ctr):
    """RFC 8439 block function with counter / nonce words (ctr_lo, ctr_hi, seed_lo, seed_hi): 16 output words"""
    k = (ctypes.c_uint32 * 8)(*key_words)
    out = (ctypes.c_uint32 * 16)()
    lib().orc_chacha20_block(k, seed, ctr, out)
    return list(out)


def galois_table(log_n, elt):
    t = np.zeros(1 << log_n, dtype=np.uint32)
    lib().orc_galois_table(log_n, elt, t.ctypes.data_as(_u32p))
    return t


def galois_elt_from_step(log_n, step):
    return int(lib().orc_galois_elt_from_step(log_n, step))
