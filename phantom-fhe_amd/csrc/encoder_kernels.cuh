// CKKS encoder / decoder kernels (PhantomCKKSEncoder, src/ckks.cu:130-283,
// src/fft.cu, src/rns_base.cu decompose_array / compose_array).
//
// With s = sparse_slots (a power of two <= N/2), gap = N / (2s) and xi = exp(2*pi*i / (4s)), a
// plaintext polynomial sum_k c_k X^(k*gap) (k < 2s) encodes the vector
//     z_j = sum_{k<s} (c_k + i*c_{k+s}) * xi^(5^j * k),   j = 0 .. s-1
// (the canonical embedding restricted to the slots 5^j, which is what makes the Galois element 5^r
// a rotation by r slots).  decode evaluates that sum, encode inverts it and rounds scale * u_k.
// The transform is the radix-2 "special FFT" over the rotation group (butterfly twiddle
// exp(2*pi*i * (5^j mod 4len) / (4len))), one launch per stage; FP64, so parity with the oracle is
// by tolerance (tests/test_encoder.py).
#pragma once

#include "modarith.cuh"

namespace pha {

__device__ __forceinline__ double2 cplx_mul(double2 a, double2 b) {
    return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// one butterfly per lane; rot[j] = 5^j mod 2N
template <bool INVERSE>
__global__ __launch_bounds__(256) void special_fft_stage_kernel(double2 *v, const uint32_t *rot, uint32_t s, uint32_t len) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    if (tid >= s / 2) return;
    const uint32_t lenh = len >> 1, lenq = len << 2;
    const uint32_t i = (tid / lenh) * len, j = tid % lenh;
    const uint32_t r = rot[j] & (lenq - 1);
    double sn, cs;
    sincospi(2.0 * static_cast<double>(r) / static_cast<double>(lenq), &sn, &cs);
    const double2 a = v[i + j], b = v[i + j + lenh];
    if (!INVERSE) {
        const double2 w = cplx_mul(b, double2{cs, sn});
        v[i + j] = double2{a.x + w.x, a.y + w.y};
        v[i + j + lenh] = double2{a.x - w.x, a.y - w.y};
    } else {
        const double2 d = double2{a.x - b.x, a.y - b.y};
        v[i + j] = double2{a.x + b.x, a.y + b.y};
        v[i + j + lenh] = cplx_mul(d, double2{cs, -sn});
    }
}

// dst[brev(i)] = (i < in_size ? src[i] : 0) * factor
__global__ __launch_bounds__(256) void bit_reverse_scale_kernel(double2 *dst, const double2 *src, uint32_t in_size, uint32_t s,
                                                                uint32_t log_s, double factor) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= s) return;
    const double2 x = i < in_size ? src[i] : double2{0.0, 0.0};
    dst[__brev(i) >> (32 - log_s)] = double2{x.x * factor, x.y * factor};
}

// largest |rint(component)| as the bit pattern of a non-negative double (orders like the value)
__global__ __launch_bounds__(256) void max_abs_kernel(const double2 *u, uint32_t s, unsigned long long *out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= s) return;
    const double m = fmax(fabs(rint(u[i].x)), fabs(rint(u[i].y)));
    atomicMax(out, static_cast<unsigned long long>(__double_as_longlong(m)));
}

// residue of the integer-valued double v modulo m (exact: a double >= 2^53 is mant * 2^e)
__device__ __forceinline__ uint64_t double_residue(double v, const DModulus &m) {
    const double a = fabs(v);
    uint64_t r;
    if (a < 9223372036854775808.0) {
        r = barrett64(static_cast<uint64_t>(a), m.value, m.ratio1);
    } else {
        int e;
        const double mant = frexp(a, &e);                         // a = mant * 2^e, mant in [0.5, 1)
        const uint64_t mi = static_cast<uint64_t>(ldexp(mant, 53));  // 53-bit integer
        uint32_t sh = static_cast<uint32_t>(e - 53);
        uint64_t p = barrett64(mi, m.value, m.ratio1), base = barrett64(2, m.value, m.ratio1);
        while (sh) {  // * 2^sh by square and multiply
            if (sh & 1) p = mul_barrett(p, base, m);
            base = mul_barrett(base, base, m);
            sh >>= 1;
        }
        r = p;
    }
    return (v < 0.0 && r) ? m.value - r : r;
}

// src/rns_base.cu decompose_array: coefficient t*gap of every limb <- rint of component t of u
// (t < s: real parts, t >= s: imaginary parts); the other coefficients were zeroed
// limbs >= size_Ql are the special primes (table index size_Q + l - size_Ql): plaintexts over Q_l u P
__global__ __launch_bounds__(256) void ckks_decompose_kernel(uint64_t *plain, const double2 *u, const DModulus *mod,
                                                             uint32_t limbs, uint32_t n, uint32_t s, uint32_t gap,
                                                             uint32_t size_Ql, uint32_t size_Q) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * s) return;
    const double v = rint(t < s ? u[t].x : u[t - s].y);
    for (uint32_t l = 0; l < limbs; l++)
        plain[static_cast<size_t>(l) * n + static_cast<size_t>(t) * gap] =
            double_residue(v, mod[l >= size_Ql ? size_Q + (l - size_Ql) : l]);
}

constexpr int kMaxCrtWords = 64;

// src/rns_base.cu compose_array: CRT-compose coefficient t*gap, centre it, multiply by inv_scale
// (given as mant * 2^exp so that huge moduli do not overflow the double before scaling)
__global__ __launch_bounds__(64) void ckks_compose_kernel(double2 *u, const uint64_t *coeff, const DModulus *mod,
                                                          const uint64_t *big_q, const uint64_t *half_q, const uint64_t *qhat,
                                                          const uint64_t *inv_qhat, uint32_t limbs, uint32_t n, uint32_t s,
                                                          uint32_t gap, double inv_mant, int inv_exp) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * s) return;
    uint64_t acc[kMaxCrtWords + 1];
    for (uint32_t w = 0; w <= limbs; w++) acc[w] = 0;
    for (uint32_t j = 0; j < limbs; j++) {
        const uint64_t x = mul_barrett(coeff[static_cast<size_t>(j) * n + static_cast<size_t>(t) * gap], inv_qhat[j], mod[j]);
        uint64_t carry = 0;
        for (uint32_t w = 0; w < limbs; w++) {
            const unsigned __int128 p = static_cast<unsigned __int128>(x) * qhat[j * limbs + w] + acc[w] + carry;
            acc[w] = static_cast<uint64_t>(p);
            carry = static_cast<uint64_t>(p >> 64);
        }
        acc[limbs] += carry;
    }
    auto geq = [&](const uint64_t *b) {  // acc >= b (b has `limbs` words)
        if (acc[limbs]) return true;
        for (int w = static_cast<int>(limbs) - 1; w >= 0; w--)
            if (acc[w] != b[w]) return acc[w] > b[w];
        return true;
    };
    while (geq(big_q)) {
        uint64_t borrow = 0;
        for (uint32_t w = 0; w < limbs; w++) {
            const uint64_t b = big_q[w], d = acc[w] - b - borrow;
            borrow = (acc[w] < b || (acc[w] == b && borrow)) ? 1 : 0;
            acc[w] = d;
        }
        acc[limbs] -= borrow;
    }
    const bool neg = geq(half_q);
    if (neg) {  // acc = Q - acc
        uint64_t borrow = 0;
        for (uint32_t w = 0; w < limbs; w++) {
            const uint64_t a = big_q[w], d = a - acc[w] - borrow;
            borrow = (a < acc[w] || (a == acc[w] && borrow)) ? 1 : 0;
            acc[w] = d;
        }
    }
    int top = static_cast<int>(limbs) - 1;
    while (top > 0 && acc[top] == 0) top--;
    double val = 0.0;
    for (int w = top; w >= 0 && w >= top - 2; w--)
        val += ldexp(static_cast<double>(acc[w]) * inv_mant, 64 * w + inv_exp);
    if (neg) val = -val;
    if (t < s)
        u[t].x = val;
    else
        u[t - s].y = val;
}

}  // namespace pha
