"""phantom-fhe_amd — MI355X-native RNS polynomial arithmetic (Phantom-FHE hot path).

This package is a thin ctypes binding over the C ABI declared in ``include/phantom_amd.h``
(built in-tree as ``phantom-fhe_amd/libphantom_amd.so`` by ``__graft_entry__.build()``).
There is no CPU fallback: importing works without a GPU (so the symbol / host-logic tests can
run), but every compute entry point needs the HIP library and a device and fails loudly
otherwise.  PyTorch is used only for device memory and streams.

The directory name contains a hyphen (it is the name the project layout prescribes); import it
as ``phantom_fhe_amd`` through the shim module at the repository root.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphantom_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "phantom_amd.h")

PHA_OK = 0
PHA_ERR_INVALID_ARGUMENT = -1
PHA_ERR_LOGIC = -2
PHA_ERR_HIP = -3
PHA_ERR_NO_DEVICE = -4

_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_sz = ctypes.c_size_t
_int = ctypes.c_int
_vp = ctypes.c_void_p
_u64p = ctypes.POINTER(_u64)
_u32p = ctypes.POINTER(_u32)
_intp = ctypes.POINTER(_int)

# name -> (restype, argtypes); must cover every function declared in include/phantom_amd.h
SIGNATURES = {
    "pha_last_error": (ctypes.c_char_p, []),
    "pha_set_sampler_key": (_int, [ctypes.c_char_p]),
    "pha_random_nonce": (_int, [ctypes.POINTER(ctypes.c_uint64)]),
    "pha_version": (ctypes.c_char_p, []),
    "pha_coeff_modulus_create": (_int, [_sz, _intp, _sz, _u64p]),
    "pha_host_ntt_tables": (_int, [_int, _u64, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p]),
    "pha_host_barrett_ratio": (_int, [_u64, _u64p]),
    "pha_host_galois_table": (_int, [_int, _u32, _u32p]),
    "pha_galois_elt_from_step": (_int, [_int, _int, _u32p]),
    "pha_host_level_constants": (_int, [_u64p, _sz, _sz, _sz, _u64p, _u64p, _u64p, _u64p]),
    "pha_context_create": (_int, [_int, _u64p, _sz, _sz, ctypes.POINTER(_vp)]),
    "pha_context_destroy": (None, [_vp]),
    "pha_context_n": (_sz, [_vp]),
    "pha_context_size_QP": (_sz, [_vp]),
    "pha_context_size_P": (_sz, [_vp]),
    "pha_context_get_moduli": (_int, [_vp, _u64p]),
    "pha_context_get_table": (_int, [_vp, _sz, _int, _u64p]),
    "pha_nwt_2d_radix8_forward_inplace": (_int, [_vp, _vp, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_backward_inplace": (_int, [_vp, _vp, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_backward": (_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_backward_scale": (_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "pha_nwt_2d_radix8_forward_inplace_include_special_mod": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range": (
        _int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_backward_inplace_include_special_mod": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_nwt_2d_radix8_forward_inplace_fuse_moddown": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "pha_ntt_forward_batched": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_ntt_backward_batched": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_ntt_launch_phase": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _int, _int, _vp]),
    "pha_ntt_block_placement": (_int, [_vp, _int]),
    "pha_ntt_fused_config": (_int, [_vp, _int, _u32, _u32]),
    "pha_ntt_fused_status": (_int, [_vp, _u32p, _u32p]),
    "pha_modup": (_int, [_vp, _sz, _vp, _vp, _vp]),
    "pha_key_switch_inner_prod": (_int, [_vp, _sz, _vp, _vp, _vp, _sz, _vp]),
    "pha_moddown_from_ntt": (_int, [_vp, _sz, _vp, _vp, _vp]),
    "pha_add_to_ct": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "pha_keyswitch_inplace": (_int, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "pha_divide_and_round_q_last_ntt": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "pha_apply_galois_ntt": (_int, [_vp, _vp, _sz, _u32, _vp, _vp]),
    "pha_tensor_prod_2x2": (_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "pha_tensor_square_2x2": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "pha_tensor_prod_mxn": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _vp, _vp]),
    "pha_multiply_and_relin_inplace": (_int, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "pha_apply_galois_inplace": (_int, [_vp, _sz, _vp, _u32, _vp, _vp]),
    "pha_rescale_to_next": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "pha_rescale_to_next_noalloc": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "pha_prepare_level": (_int, [_vp, _sz]),
    "pha_gen_secretkey": (_int, [_vp, _u64, _vp, _vp]),
    "pha_encrypt_zero_symmetric": (_int, [_vp, _vp, _sz, _u64, _u64, _vp, _vp]),
    "pha_encrypt_symmetric": (_int, [_vp, _vp, _vp, _sz, _u64, _u64, _vp, _vp]),
    "pha_gen_publickey": (_int, [_vp, _vp, _u64, _vp, _vp]),
    "pha_encrypt_zero_asymmetric": (_int, [_vp, _vp, _sz, _u64, _vp, _vp]),
    "pha_encrypt_asymmetric": (_int, [_vp, _vp, _vp, _sz, _u64, _vp, _vp]),
    "pha_ckks_decrypt": (_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp]),
    "pha_generate_one_kswitch_key": (_int, [_vp, _vp, _vp, _u64, ctypes.POINTER(_vp), _vp]),
    "pha_gen_relinkey": (_int, [_vp, _vp, _u64, ctypes.POINTER(_vp), _vp]),
    "pha_gen_galois_key": (_int, [_vp, _vp, _u32, _u64, ctypes.POINTER(_vp), _vp]),
    "pha_add_rns_poly": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "pha_sub_rns_poly": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "pha_multiply_rns_poly": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "pha_multiply_add_rns_poly": (_int, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp]),
    "pha_negate_rns_poly": (_int, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "pha_mod_switch_drop_to_next": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "pha_ckks_encode": (_int, [_vp, _vp, _sz, _sz, _sz, ctypes.c_double, _vp, _vp]),
    "pha_ckks_decode": (_int, [_vp, _vp, _sz, ctypes.c_double, _sz, _vp, _vp]),
    "pha_rotate_hoisted": (_int, [_vp, _sz, _vp, _sz, _u32p, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    "pha_multiply_relin_rescale": (_int, [_vp, _sz, _vp, _vp, _vp, _vp, _int, _vp]),
    "pha_multiply_relin_rescale_batched": (_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _int, _vp]),
    "pha_apply_galois_batched": (_int, [_vp, _sz, _vp, _vp, _sz, _u32, _vp, _vp]),
    "pha_ckks_encode_qp": (_int, [_vp, _vp, _sz, _sz, _sz, ctypes.c_double, _vp, _vp]),
    "pha_linear_transform_double_hoisted": (_int, [_vp, _sz, _vp, _sz, _u32p, ctypes.POINTER(_vp), _sz, _u32p, ctypes.POINTER(_vp),
                                                   ctypes.POINTER(_vp), _vp, _vp]),
    "pha_mod_raise": (_int, [_vp, _vp, _vp, _vp]),
    "pha_gen_secretkey_sparse": (_int, [_vp, _u64, _sz, _vp, _vp]),
    "pha_lp_modup_begin": (_int, [_vp, _sz, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_lp_modup_finish": (_int, [_vp, _sz, _vp, _vp, _vp, _sz, ctypes.POINTER(ctypes.c_uint32), _sz, _sz, _sz, _sz, _vp]),
    "pha_lp_key_switch_inner_prod": (_int, [_vp, _sz, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp]),
    "pha_lp_moddown_begin": (_int, [_vp, _sz, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "pha_lp_moddown_finish": (_int, [_vp, _sz, _vp, _vp, _vp, _sz, _sz, ctypes.POINTER(ctypes.c_uint32), _sz, _sz, _vp]),
    "pha_lp_tensor_prod_2x2": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "pha_lp_rescale_begin": (_int, [_vp, _sz, _vp, _sz, _vp, _vp]),
    "pha_lp_rescale_finish": (_int, [_vp, _sz, _vp, _vp, _sz, _vp, _sz, _sz, _vp]),
}

_lib = None


class PhantomError(RuntimeError):
    """C ABI call failed (std::invalid_argument / std::logic_error / HIP failure in the reference)."""

    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def lib():
    """Load libphantom_amd.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(rc):
    if rc != PHA_OK:
        raise PhantomError(rc, lib().pha_last_error().decode())


def set_sampler_key(key32=None):
    """Installs the process's 256-bit sampler key (None: fresh OS entropy, the default state).  A fixed key makes
    every sampled object a function of its nonces: tests and oracle comparisons only."""
    if key32 is not None:
        assert len(key32) == 32
        key32 = bytes(key32)
    _check(lib().pha_set_sampler_key(key32))


def random_nonce():
    """64 bits of OS entropy: the `seed` (nonce) of one draw"""
    out = ctypes.c_uint64(0)
    _check(lib().pha_random_nonce(ctypes.byref(out)))
    return out.value


def _np_u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_u64p)


# ---- host number theory (no GPU) ---------------------------------------------------------------
def coeff_modulus_create(poly_modulus_degree, bit_sizes):
    """CoeffModulus::Create (src/modulus.cu:79-110)."""
    bits = (_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    _check(lib().pha_coeff_modulus_create(poly_modulus_degree, bits, len(bit_sizes), out.ctypes.data_as(_u64p)))
    return out


def host_ntt_tables(log_n, modulus):
    n = 1 << log_n
    arrs = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
    ni, nis = _u64(), _u64()
    _check(lib().pha_host_ntt_tables(log_n, int(modulus), *[a.ctypes.data_as(_u64p) for a in arrs],
                                     ctypes.byref(ni), ctypes.byref(nis)))
    return (*arrs, ni.value, nis.value)


def host_barrett_ratio(modulus):
    r = (_u64 * 2)()
    _check(lib().pha_host_barrett_ratio(int(modulus), r))
    return int(r[0]), int(r[1])


def host_galois_table(log_n, elt):
    t = np.zeros(1 << log_n, dtype=np.uint32)
    _check(lib().pha_host_galois_table(log_n, int(elt), t.ctypes.data_as(_u32p)))
    return t


def galois_elt_from_step(log_n, step):
    e = _u32()
    _check(lib().pha_galois_elt_from_step(log_n, int(step), ctypes.byref(e)))
    return e.value


def host_level_constants(moduli_qp, size_p, size_ql):
    m, mp = _np_u64(moduli_qp)
    outs = [np.zeros(max(size_ql - 1, 0), dtype=np.uint64)] + [np.zeros(size_ql, dtype=np.uint64) for _ in range(3)]
    _check(lib().pha_host_level_constants(mp, len(m), size_p, size_ql, *[o.ctypes.data_as(_u64p) for o in outs]))
    return dict(inv_q_last_mod_q=outs[0], bigP_mod_q=outs[1], bigPInv_mod_q=outs[2],
                partQlHatInv_mod_Ql_concat=outs[3])


# ---- device side ------------------------------------------------------------------------------
def _ptr(t):
    """Device pointer of a torch tensor (int64 storage of uint64 words) or raw int."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA(HIP) tensors"
    return t.data_ptr()


def to_device(a, device="cuda"):
    """numpy uint64 array -> torch int64 device tensor with the same bit pattern."""
    import torch
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t):
    return t.detach().cpu().numpy().view(np.uint64)


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


class EvalKey:
    """PhantomRelinKey layout (include/secretkey.h:85-113): beta device buffers [2][size_QP][N] plus
    a device array of their pointers."""

    def __init__(self, digits):
        import torch
        self.digits = list(digits)
        ptrs = np.array([d.data_ptr() for d in self.digits], dtype=np.uint64)
        self.ptrs = torch.from_numpy(ptrs.view(np.int64)).to(self.digits[0].device)

    @property
    def ptr(self):
        return self.ptrs.data_ptr()


class PhantomContext:
    """Mirror of PhantomContext + DNTTTable + DRNSTool (src/context.cu:120-290) over the C ABI."""

    def __init__(self, log_n, moduli_qp, size_p=1):
        self._h = _vp()
        self.log_n = log_n
        self.n = 1 << log_n
        self.moduli = np.ascontiguousarray(moduli_qp, dtype=np.uint64)
        self.size_QP = len(self.moduli)
        self.size_P = size_p
        self.size_Q = self.size_QP - size_p
        _check(lib().pha_context_create(log_n, self.moduli.ctypes.data_as(_u64p), self.size_QP, size_p,
                                        ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().pha_context_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size_Ql(self, chain_index):
        return self.size_Q - (chain_index - 1)

    def beta(self, chain_index):
        return -(-self.size_Ql(chain_index) // self.size_P)

    def get_table(self, limb, which):
        out = np.zeros(self.n, dtype=np.uint64)
        _check(lib().pha_context_get_table(self._h, limb, which, out.ctypes.data_as(_u64p)))
        return out

    # NTT family -------------------------------------------------------------------------------
    def nwt_2d_radix8_forward_inplace(self, inout, count, start=0):
        _check(lib().pha_nwt_2d_radix8_forward_inplace(self._h, _ptr(inout), count, start, _stream()))

    def nwt_2d_radix8_backward_inplace(self, inout, count, start=0):
        _check(lib().pha_nwt_2d_radix8_backward_inplace(self._h, _ptr(inout), count, start, _stream()))

    def nwt_2d_radix8_backward(self, out, inp, count, start=0):
        _check(lib().pha_nwt_2d_radix8_backward(self._h, _ptr(out), _ptr(inp), count, start, _stream()))

    def nwt_2d_radix8_backward_scale(self, out, inp, count, start, scale, scale_shoup):
        _check(lib().pha_nwt_2d_radix8_backward_scale(self._h, _ptr(out), _ptr(inp), count, start, _ptr(scale),
                                                      _ptr(scale_shoup), _stream()))

    def nwt_2d_radix8_forward_inplace_include_special_mod(self, inout, count, start, size_qp, size_p):
        _check(lib().pha_nwt_2d_radix8_forward_inplace_include_special_mod(self._h, _ptr(inout), count, start,
                                                                           size_qp, size_p, _stream()))

    def nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(self, inout, count, start, size_qp, size_p,
                                                                        excl_start, excl_end):
        _check(lib().pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(
            self._h, _ptr(inout), count, start, size_qp, size_p, excl_start, excl_end, _stream()))

    def nwt_2d_radix8_backward_inplace_include_special_mod(self, inout, count, start, size_qp, size_p):
        _check(lib().pha_nwt_2d_radix8_backward_inplace_include_special_mod(self._h, _ptr(inout), count, start,
                                                                            size_qp, size_p, _stream()))

    def nwt_2d_radix8_forward_inplace_fuse_moddown(self, ct, cx, pinv, pinv_shoup, delta, count, start=0):
        _check(lib().pha_nwt_2d_radix8_forward_inplace_fuse_moddown(self._h, _ptr(ct), _ptr(cx), _ptr(pinv),
                                                                    _ptr(pinv_shoup), _ptr(delta), count, start,
                                                                    _stream()))

    def ntt_forward_batched(self, inout, polys, poly_stride_limbs, count, start=0):
        _check(lib().pha_ntt_forward_batched(self._h, _ptr(inout), polys, poly_stride_limbs, count, start, _stream()))

    def ntt_backward_batched(self, inout, polys, poly_stride_limbs, count, start=0):
        _check(lib().pha_ntt_backward_batched(self._h, _ptr(inout), polys, poly_stride_limbs, count, start, _stream()))

    def ntt_launch_phase(self, inout, polys, poly_stride_limbs, count, start, inverse, phase):
        _check(lib().pha_ntt_launch_phase(self._h, _ptr(inout), polys, poly_stride_limbs, count, start, int(inverse),
                                          int(phase), _stream()))

    def ntt_block_placement(self, limb_to_xcd):
        _check(lib().pha_ntt_block_placement(self._h, int(limb_to_xcd)))

    def ntt_fused_config(self, enable, flags=0, min_units=16):
        """Single-launch forward transform (csrc/ntt_fused.cuh) on / off; `flags` = its FUSED_* experiment switches."""
        _check(lib().pha_ntt_fused_config(self._h, int(enable), int(flags), int(min_units)))

    def ntt_fused_status(self):
        """(error word of the single-launch transform, number of task queues = XCDs); synchronises the device."""
        e, q = _u32(), _u32()
        _check(lib().pha_ntt_fused_status(self._h, ctypes.byref(e), ctypes.byref(q)))
        return e.value, q.value

    # DRNSTool ---------------------------------------------------------------------------------
    def modup(self, chain_index, dst, cks):
        _check(lib().pha_modup(self._h, chain_index, _ptr(dst), _ptr(cks), _stream()))

    def key_switch_inner_prod(self, chain_index, cx, t_mod_up, evk, reduction_threshold=0):
        _check(lib().pha_key_switch_inner_prod(self._h, chain_index, _ptr(cx), _ptr(t_mod_up), evk.ptr,
                                               reduction_threshold, _stream()))

    def moddown_from_NTT(self, chain_index, ct_i, cx_i):
        _check(lib().pha_moddown_from_ntt(self._h, chain_index, _ptr(ct_i), _ptr(cx_i), _stream()))

    def add_to_ct(self, ct, cx, size_ql):
        _check(lib().pha_add_to_ct(self._h, _ptr(ct), _ptr(cx), size_ql, _stream()))

    def keyswitch_inplace(self, chain_index, ct, c2, evk):
        _check(lib().pha_keyswitch_inplace(self._h, chain_index, _ptr(ct), _ptr(c2), evk.ptr, _stream()))

    def divide_and_round_q_last_ntt(self, chain_index, src, cipher_size, dst):
        _check(lib().pha_divide_and_round_q_last_ntt(self._h, chain_index, _ptr(src), cipher_size, _ptr(dst),
                                                     _stream()))

    # Galois / tensor --------------------------------------------------------------------------
    def apply_galois_ntt(self, operand, coeff_mod_size, galois_elt, result):
        _check(lib().pha_apply_galois_ntt(self._h, _ptr(operand), coeff_mod_size, galois_elt, _ptr(result), _stream()))

    def tensor_prod_2x2(self, op1, op2, coeff_mod_size, result):
        _check(lib().pha_tensor_prod_2x2(self._h, _ptr(op1), _ptr(op2), coeff_mod_size, _ptr(result), _stream()))

    def tensor_prod_mxn(self, op1, op2, result):
        """result [m + n - 1][limbs][N] from op1 [m][limbs][N], op2 [n][limbs][N]; result may alias op1"""
        _check(lib().pha_tensor_prod_mxn(self._h, _ptr(op1), op1.shape[0], _ptr(op2), op2.shape[0], op1.shape[1], _ptr(result),
                                         _stream()))

    def tensor_square_2x2(self, op, coeff_mod_size, result):
        _check(lib().pha_tensor_square_2x2(self._h, _ptr(op), coeff_mod_size, _ptr(result), _stream()))

    # evaluator composites (include/evaluate.cuh) ---------------------------------------------
    def multiply_and_relin_inplace(self, chain_index, ct1, ct2, relin_key):
        _check(lib().pha_multiply_and_relin_inplace(self._h, chain_index, _ptr(ct1), _ptr(ct2), relin_key.ptr,
                                                    _stream()))

    def apply_galois_inplace(self, chain_index, ct, galois_elt, galois_key):
        _check(lib().pha_apply_galois_inplace(self._h, chain_index, _ptr(ct), galois_elt, galois_key.ptr, _stream()))

    def rescale_to_next(self, chain_index, src, cipher_size, dst):
        _check(lib().pha_rescale_to_next(self._h, chain_index, _ptr(src), cipher_size, _ptr(dst), _stream()))

    def rescale_to_next_noalloc(self, chain_index, src, cipher_size, dst):
        _check(lib().pha_rescale_to_next_noalloc(self._h, chain_index, _ptr(src), cipher_size, _ptr(dst), _stream()))

    def prepare_level(self, chain_index):
        _check(lib().pha_prepare_level(self._h, chain_index))

    # element-wise evaluator ops (src/evaluate.cu add/sub/negate/add_plain/sub_plain/multiply_plain) --
    def add_inplace(self, ct1, ct2):
        _check(lib().pha_add_rns_poly(self._h, _ptr(ct1), _ptr(ct1), _ptr(ct2), ct1.shape[-2], ct1.shape[0], ct2.shape[0]
                                      if ct2.dim() == 3 else 1, _stream()))

    def sub_inplace(self, ct1, ct2):
        _check(lib().pha_sub_rns_poly(self._h, _ptr(ct1), _ptr(ct1), _ptr(ct2), ct1.shape[-2], ct1.shape[0], ct2.shape[0]
                                      if ct2.dim() == 3 else 1, _stream()))

    def negate_inplace(self, ct):
        _check(lib().pha_negate_rns_poly(self._h, _ptr(ct), _ptr(ct), ct.shape[-2], ct.shape[0], _stream()))

    def add_plain_inplace(self, ct, plain):
        _check(lib().pha_add_rns_poly(self._h, _ptr(ct), _ptr(ct), _ptr(plain), ct.shape[-2], 1, 1, _stream()))

    def sub_plain_inplace(self, ct, plain):
        _check(lib().pha_sub_rns_poly(self._h, _ptr(ct), _ptr(ct), _ptr(plain), ct.shape[-2], 1, 1, _stream()))

    def multiply_plain_inplace(self, ct, plain):
        _check(lib().pha_multiply_rns_poly(self._h, _ptr(ct), _ptr(ct), _ptr(plain), ct.shape[-2], ct.shape[0], 1, _stream()))

    def mod_switch_to_next(self, chain_index, ct):
        out = self._empty(ct.shape[0], ct.shape[1] - 1)
        _check(lib().pha_mod_switch_drop_to_next(self._h, chain_index, _ptr(ct), ct.shape[0], _ptr(out), _stream()))
        return out

    # PhantomCKKSEncoder (include/ckks.h, src/ckks.cu:130-283) -------------------------------
    def encode(self, values, scale, chain_index=1, sparse_slots=0):
        """complex values (numpy) -> plaintext [l][N] device tensor in NTT form"""
        v = np.ascontiguousarray(values, dtype=np.complex128)
        plain = self._empty(self.level_limbs(chain_index))
        _check(lib().pha_ckks_encode(self._h, v.ctypes.data, v.size, sparse_slots, chain_index, float(scale), _ptr(plain),
                                     _stream()))
        return plain

    def encode_qp(self, values, scale, chain_index=1, sparse_slots=0):
        """plaintext over Q_l u P, [l + size_P][N]: the operand format of linear_transform_double_hoisted"""
        v = np.ascontiguousarray(values, dtype=np.complex128)
        plain = self._empty(self.level_limbs(chain_index) + self.size_P)
        _check(lib().pha_ckks_encode_qp(self._h, v.ctypes.data, v.size, sparse_slots, chain_index, float(scale), _ptr(plain),
                                        _stream()))
        return plain

    def decode(self, plain, scale, sparse_slots, chain_index=1):
        out = np.zeros(sparse_slots, dtype=np.complex128)
        _check(lib().pha_ckks_decode(self._h, _ptr(plain), chain_index, float(scale), sparse_slots, out.ctypes.data,
                                     _stream()))
        return out

    # PhantomSecretKey (include/secretkey.h:150-290) ------------------------------------------
    def level_limbs(self, chain_index):
        return self.size_QP if chain_index == 0 else self.size_Q - (chain_index - 1)

    def _empty(self, *shape):
        import torch
        return torch.empty((*shape, self.n), dtype=torch.int64, device="cuda")

    def gen_secretkey(self, seed, hamming_weight=0):
        sk = self._empty(self.size_QP)
        if hamming_weight:
            _check(lib().pha_gen_secretkey_sparse(self._h, seed, hamming_weight, _ptr(sk), _stream()))
        else:
            _check(lib().pha_gen_secretkey(self._h, seed, _ptr(sk), _stream()))
        return sk

    def mod_raise(self, ct):
        """ct [2][1][N] at the lowest level -> [2][size_Q][N] at chain_index 1 (Bootstrapper::modraise_inplace)"""
        out = self._empty(2, self.size_Q)
        _check(lib().pha_mod_raise(self._h, _ptr(ct), _ptr(out), _stream()))
        return out

    def encrypt_zero_symmetric(self, sk, chain_index, seed_a, seed_e):
        ct = self._empty(2, self.level_limbs(chain_index))
        _check(lib().pha_encrypt_zero_symmetric(self._h, _ptr(sk), chain_index, seed_a, seed_e, _ptr(ct), _stream()))
        return ct

    def encrypt_symmetric(self, sk, plain, chain_index, seed_a, seed_e):
        ct = self._empty(2, self.level_limbs(chain_index))
        _check(lib().pha_encrypt_symmetric(self._h, _ptr(sk), _ptr(plain), chain_index, seed_a, seed_e, _ptr(ct),
                                           _stream()))
        return ct

    def gen_publickey(self, sk, seed):
        """PhantomSecretKey::gen_publickey (src/secretkey.cu:437-452): [2][size_QP][N]"""
        pk = self._empty(2, self.size_QP)
        _check(lib().pha_gen_publickey(self._h, _ptr(sk), seed, _ptr(pk), _stream()))
        return pk

    def encrypt_zero_asymmetric(self, pk, chain_index, seed):
        ct = self._empty(2, self.level_limbs(chain_index))
        _check(lib().pha_encrypt_zero_asymmetric(self._h, _ptr(pk), chain_index, seed, _ptr(ct), _stream()))
        return ct

    def encrypt_asymmetric(self, pk, plain, chain_index, seed):
        """PhantomPublicKey::encrypt_asymmetric (src/secretkey.cu:133-164)"""
        ct = self._empty(2, self.level_limbs(chain_index))
        _check(lib().pha_encrypt_asymmetric(self._h, _ptr(pk), _ptr(plain), chain_index, seed, _ptr(ct), _stream()))
        return ct

    def ckks_decrypt(self, sk, ct, chain_index):
        plain = self._empty(self.level_limbs(chain_index))
        _check(lib().pha_ckks_decrypt(self._h, _ptr(sk), _ptr(ct), ct.shape[0], chain_index, _ptr(plain), _stream()))
        return plain

    def _new_evk(self):
        digits = [self._empty(2, self.size_QP) for _ in range(self.size_Q // self.size_P)]
        return digits, (_vp * len(digits))(*[d.data_ptr() for d in digits])

    def generate_one_kswitch_key(self, sk, new_key, seed):
        digits, arr = self._new_evk()
        _check(lib().pha_generate_one_kswitch_key(self._h, _ptr(sk), _ptr(new_key), seed, arr, _stream()))
        return EvalKey(digits)

    def gen_relinkey(self, sk, seed):
        digits, arr = self._new_evk()
        _check(lib().pha_gen_relinkey(self._h, _ptr(sk), seed, arr, _stream()))
        return EvalKey(digits)

    def gen_galois_key(self, sk, galois_elt, seed):
        digits, arr = self._new_evk()
        _check(lib().pha_gen_galois_key(self._h, _ptr(sk), galois_elt, seed, arr, _stream()))
        return EvalKey(digits)

    def rotate_hoisted(self, chain_index, ct, galois_elts, galois_keys):
        """Rotations of ONE ciphertext by several Galois elements sharing the modup; returns the list of
        rotated ciphertexts (include/phantom_amd.h "hoisted rotations")."""
        outs = [self._empty(2, ct.shape[1]) for _ in galois_elts]
        elts = (_u32 * len(galois_elts))(*galois_elts)
        keys = (_vp * len(galois_keys))(*[k.ptr for k in galois_keys])
        out_ptrs = (_vp * len(outs))(*[o.data_ptr() for o in outs])
        _check(lib().pha_rotate_hoisted(self._h, chain_index, _ptr(ct), len(outs), elts, keys, out_ptrs, _stream()))
        return outs

    def multiply_relin_rescale(self, chain_index, ct1, ct2, relin_key, times2=False):
        """rescale(relinearize(ct1 * ct2)) (x 2) as one composite with a single moddown; out [2][l-1][N]"""
        out = self._empty(2, ct1.shape[1] - 1)
        _check(lib().pha_multiply_relin_rescale(self._h, chain_index, _ptr(ct1), _ptr(ct2), relin_key.ptr, _ptr(out),
                                                1 if times2 else 0, _stream()))
        return out

    def multiply_relin_rescale_batched(self, chain_index, ct1s, ct2s, relin_key, times2=False):
        """len(ct1s) independent multiply_relin_rescale products at one level in one launch sequence; out [batch][2][l-1][N]"""
        import torch
        b = len(ct1s)
        out = torch.empty((b, 2, ct1s[0].shape[1] - 1, self.n), dtype=torch.int64, device=ct1s[0].device)
        p1, p2 = (_vp * b)(*[t.data_ptr() for t in ct1s]), (_vp * b)(*[t.data_ptr() for t in ct2s])
        _check(lib().pha_multiply_relin_rescale_batched(self._h, chain_index, b, p1, p2, relin_key.ptr, _ptr(out), 1 if times2 else 0,
                                                        _stream()))
        return out

    def apply_galois_batched(self, chain_index, cts, galois_elt, galois_key):
        """cts [batch][2][l][N] -> rotated copies, one launch sequence for the whole batch"""
        import torch
        out = torch.empty_like(cts)
        _check(lib().pha_apply_galois_batched(self._h, chain_index, _ptr(cts), _ptr(out), cts.shape[0], galois_elt, galois_key.ptr,
                                              _stream()))
        return out

    def linear_transform_double_hoisted(self, chain_index, ct, baby_elts, baby_keys, giant_elts, giant_keys, plains):
        """sum_g rot_G( sum_b plains[g][b] * rot_B(ct) ) with one modup for the babies and one moddown per giant step
        (include/phantom_amd.h).  Galois element 1 = no rotation (key None); plains[g][b] = encode_qp plaintext or None."""
        out = self._empty(2, ct.shape[1])
        nb, ng = len(baby_elts), len(giant_elts)
        be, ge = (_u32 * nb)(*baby_elts), (_u32 * ng)(*giant_elts)
        bk = (_vp * nb)(*[k.ptr if k is not None else None for k in baby_keys])
        gk = (_vp * ng)(*[k.ptr if k is not None else None for k in giant_keys])
        flat = [plains[g][b] for g in range(ng) for b in range(nb)]
        pp = (_vp * (nb * ng))(*[p.data_ptr() if p is not None else None for p in flat])
        _check(lib().pha_linear_transform_double_hoisted(self._h, chain_index, _ptr(ct), nb, be, bk, ng, ge, gk, pp, _ptr(out),
                                                         _stream()))
        return out

    # limb-parallel steps (include/phantom_amd.h "limb-parallel key switch"); w = (q_begin, q_count,
    # p_begin, p_count), orchestrated by limb_parallel.LimbParallelEvaluator ------------------
    def lp_modup_begin(self, chain_index, c2_ntt, c2_coeff, w, row_begin=None):
        """c2_coeff [rows][N]: owned limb w[0] + j lands on row row_begin + j (default: its own index)"""
        _check(lib().pha_lp_modup_begin(self._h, chain_index, _ptr(c2_ntt), _ptr(c2_coeff), c2_coeff.shape[0],
                                        w[0] if row_begin is None else row_begin, w[0], w[1], _stream()))

    def lp_modup_finish(self, chain_index, t_mod_up, c2_ntt, c2_coeff, w, q_rows=None):
        """q_rows[j] = row of limb j in the gathered c2_coeff (default: j)"""
        rows = None if q_rows is None else (ctypes.c_uint32 * len(q_rows))(*q_rows)
        _check(lib().pha_lp_modup_finish(self._h, chain_index, _ptr(t_mod_up), _ptr(c2_ntt), _ptr(c2_coeff), c2_coeff.shape[0],
                                         rows, w[0], w[1], w[2], w[3], _stream()))

    def lp_key_switch_inner_prod(self, chain_index, cx, t_mod_up, evk, w):
        _check(lib().pha_lp_key_switch_inner_prod(self._h, chain_index, _ptr(cx), _ptr(t_mod_up), evk.ptr, w[0], w[1],
                                                  w[2], w[3], _stream()))

    def lp_moddown_begin(self, chain_index, cx, p_coeff, w, row_begin=None, poly_stride=None):
        """Owned P limb w[2] + j of poly z lands on row z * poly_stride + row_begin + j of p_coeff (N-word rows).
        Defaults, for p_coeff [2][rows][N]: poly_stride = rows, row_begin = the limb's own index."""
        stride = p_coeff.shape[1] if poly_stride is None else poly_stride
        _check(lib().pha_lp_moddown_begin(self._h, chain_index, _ptr(cx), _ptr(p_coeff), p_coeff.numel() // self.n, stride,
                                          w[2] if row_begin is None else row_begin, w[2], w[3], _stream()))

    def lp_moddown_finish(self, chain_index, ct, cx, p_coeff, w, p_rows=None, poly_stride=None):
        """p_rows[k] = row of special limb k of poly 0 in p_coeff (default: k); poly 1 is poly_stride rows further"""
        rows = None if p_rows is None else (ctypes.c_uint32 * len(p_rows))(*p_rows)
        stride = p_coeff.shape[1] if poly_stride is None else poly_stride
        _check(lib().pha_lp_moddown_finish(self._h, chain_index, _ptr(ct), _ptr(cx), _ptr(p_coeff), p_coeff.numel() // self.n,
                                           stride, rows, w[0], w[1], _stream()))

    def lp_tensor_prod_2x2(self, op1, op2, result, w):
        """result [3][l][N] = op1 [2][l][N] (x) op2 [2][l][N] on the limbs [w[0], w[0] + w[1])"""
        _check(lib().pha_lp_tensor_prod_2x2(self._h, _ptr(op1), _ptr(op2), op1.shape[1], w[0], w[1], _ptr(result), _stream()))

    def lp_rescale_begin(self, chain_index, src, cipher_size, last_coeff):
        _check(lib().pha_lp_rescale_begin(self._h, chain_index, _ptr(src), cipher_size, _ptr(last_coeff), _stream()))

    def lp_rescale_finish(self, chain_index, src, last_coeff, cipher_size, dst, w):
        _check(lib().pha_lp_rescale_finish(self._h, chain_index, _ptr(src), _ptr(last_coeff), cipher_size, _ptr(dst),
                                           w[0], w[1], _stream()))
