// Device modular arithmetic for gfx950: 64-bit Shoup / Barrett multiply built from 32-bit
// multiplies (v_mad_u64_u32 / v_mul_hi_u32), lazy Harvey butterflies.
// Semantics follow include/uintmodmath.cuh and include/butterfly.cuh of the reference; the
// instruction selection is CDNA4-specific.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pha {

// {value, Shoup companion floor(value * 2^64 / q)} — one 16-byte twiddle record
using tw_t = ulonglong2;

struct DModulus {  // include/gputype.h:14-40
    uint64_t value;
    uint64_t ratio0;  // floor(2^128 / value), low word
    uint64_t ratio1;  // high word = floor(2^64 / value)
};

__device__ __forceinline__ uint32_t lo32(uint64_t x) { return static_cast<uint32_t>(x); }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return static_cast<uint32_t>(x >> 32); }
// 32x32 + 64 -> 64, one v_mad_u64_u32
__device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) {
    return static_cast<uint64_t>(a) * b + c;
}

// exact high 64 bits of a 64x64 product: 4 x v_mad_u64_u32
__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint64_t p00 = mad32(a0, b0, 0);
    const uint64_t m1 = mad32(a0, b1, hi32(p00));
    const uint64_t m2 = mad32(a1, b0, lo32(m1));
    return mad32(a1, b1, static_cast<uint64_t>(hi32(m1)) + hi32(m2));
}

// low 64 bits of a 64x64 product: 1 x v_mad_u64_u32 + 2 x v_mul_lo_u32
__device__ __forceinline__ uint64_t mullo64(uint64_t a, uint64_t b) { return a * b; }

struct u128_t {
    uint64_t lo, hi;
};

__device__ __forceinline__ u128_t mul64x64(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint64_t p00 = mad32(a0, b0, 0);
    const uint64_t m1 = mad32(a0, b1, hi32(p00));
    const uint64_t m2 = mad32(a1, b0, lo32(m1));
    u128_t r;
    r.lo = (static_cast<uint64_t>(lo32(m2)) << 32) | lo32(p00);
    r.hi = mad32(a1, b1, static_cast<uint64_t>(hi32(m1)) + hi32(m2));
    return r;
}

__device__ __forceinline__ void add128(u128_t &acc, const u128_t &x) {
    const uint64_t lo = acc.lo + x.lo;
    acc.hi += x.hi + (lo < acc.lo ? 1 : 0);
    acc.lo = lo;
}

__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return (static_cast<uint64_t>(hi) << 32) | lo; }

// High word of a 64x64 product without the low x low partial product and without the carries of
// the cross terms: true - 2 <= result <= true.  2 x v_mul_hi_u32 + 1 x v_mad_u64_u32.
__device__ __forceinline__ uint64_t mulhi64_approx(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint64_t cross = static_cast<uint64_t>(__umulhi(a0, b1)) + __umulhi(a1, b0);
    return mad32(a1, b1, cross);
}

// x in [0, 2m) -> [0, m)   (include/uintmodmath.cuh:18-21)
__device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }

// x * w mod q in [0, 2q) for any 64-bit x, w < q (include/uintmodmath.cuh:225-231)
__device__ __forceinline__ uint64_t mul_shoup_lazy(uint64_t x, uint64_t w, uint64_t w_shoup, uint64_t q) {
    return x * w - mulhi64(x, w_shoup) * q;
}
// canonical result (include/uintmodmath.cuh:207-215)
__device__ __forceinline__ uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t w_shoup, uint64_t q) {
    return csub(mul_shoup_lazy(x, w, w_shoup, q), q);
}

// x mod q for any 64-bit x, mu_hi = floor(2^64/q) (include/uintmodmath.cuh:144-151)
__device__ __forceinline__ uint64_t barrett64(uint64_t x, uint64_t q, uint64_t mu_hi) {
    return csub(x - mulhi64(mu_hi, x) * q, q);
}

// 128 -> 64 Barrett with the 128-bit ratio (include/uintmodmath.cuh:96-136): canonical result
// for any x < 2^128 (the quotient estimate is at most one too small)
__device__ __forceinline__ uint64_t barrett128(const u128_t &x, const DModulus &m) {
    // floor(x * ratio / 2^128), dropping only the carry out of lo*ratio0's low half
    const uint64_t c0 = mulhi64(x.lo, m.ratio0);
    const u128_t p1 = mul64x64(x.lo, m.ratio1);
    const u128_t p2 = mul64x64(x.hi, m.ratio0);
    const uint64_t s1 = p1.lo + c0;
    const uint64_t carry1 = s1 < c0 ? 1 : 0;
    const uint64_t s2 = s1 + p2.lo;
    const uint64_t carry2 = s2 < p2.lo ? 1 : 0;
    const uint64_t qhat = x.hi * m.ratio1 + p1.hi + p2.hi + carry1 + carry2;
    return csub(x.lo - qhat * m.value, m.value);
}

__device__ __forceinline__ uint64_t mul_barrett(uint64_t a, uint64_t b, const DModulus &m) {
    return barrett128(mul64x64(a, b), m);
}

// Per-limb constants of the lazy butterflies: q, 4q and 2^64 - q (so that x*w - h*q becomes a
// chain of multiply-adds modulo 2^64).
struct LazyMod {
    uint64_t q, q4, nq, q8, nq8;
    __device__ __forceinline__ explicit LazyMod(uint64_t q_)
        : q(q_), q4(q_ << 2), nq(0 - q_), q8(q_ << 3), nq8(0 - (q_ << 3)) {}
};

// x * w mod q in [0, 4q) for ANY 64-bit x, w < q < 2^61: Shoup's multiplication with the
// approximate quotient above (the quotient is at most 3 too small: one from Shoup's floor, two
// from mulhi64_approx).  9 integer multiplies: 3 for the quotient, 2 v_mad_u64_u32 for the low
// words and 4 v_mul_lo_u32 for the cross terms of the low 64 bits.
__device__ __forceinline__ uint64_t mul_shoup_lazy4(uint64_t x, const tw_t &w, const LazyMod &m) {
    const uint64_t h = mulhi64_approx(x, w.y);
    const uint32_t x0 = lo32(x), x1 = hi32(x), h0 = lo32(h), h1 = hi32(h);
    const uint32_t w0 = lo32(w.x), w1 = hi32(w.x), n0 = lo32(m.nq), n1 = hi32(m.nq);
    const uint64_t acc = mad32(h0, n0, mad32(x0, w0, 0));
    const uint32_t hi = hi32(acc) + x0 * w1 + x1 * w0 + h0 * n1 + h1 * n0;
    return pack64(lo32(acc), hi);
}

// forward butterfly on the wide lazy range: x,y in [0,8q) -> [0,8q)  (8q < 2^64 since q < 2^61)
__device__ __forceinline__ void ct_butterfly8(uint64_t &x, uint64_t &y, const tw_t &w, const LazyMod &m) {
    const uint64_t t = mul_shoup_lazy4(y, w, m);
    const uint64_t u = csub(x, m.q4);
    x = u + t;
    y = u + (m.q4 - t);
}

// forward butterfly for q < 2^60 (16q < 2^64): the conditional subtraction (the most expensive
// non-multiply part: a 64-bit compare plus selects) runs only on every other stage.
//   CS stage:     x,y in [0,16q): u = x - 8q if x >= 8q  -> outputs in [0,12q)
//   non-CS stage: x,y in [0,12q)                         -> outputs in [0,16q)
template <bool CS>
__device__ __forceinline__ void ct_butterfly16(uint64_t &x, uint64_t &y, const tw_t &w, const LazyMod &m) {
    const uint64_t t = mul_shoup_lazy4(y, w, m);
    const uint64_t u = CS ? x + (x >= m.q8 ? m.nq8 : 0ull) : x;
    x = u + t;
    y = u + (m.q4 - t);
}

// inverse butterfly: x,y in [0,4q) -> [0,4q)
__device__ __forceinline__ void gs_butterfly4(uint64_t &x, uint64_t &y, const tw_t &w, const LazyMod &m) {
    const uint64_t d = x + (m.q4 - y);
    x = csub(x + y, m.q4);
    y = mul_shoup_lazy4(d, w, m);
}

// ---------------------------------------------------------------------------------------------
// Arithmetic policies of the NTT kernels.  A limb is transformed either with the integer lazy
// butterflies above (any q < 2^61) or, when q < 2^50, with exact integer arithmetic carried in
// FP64 registers: gfx950 issues v_fma_f64 at full rate while a 64-bit integer modular multiply
// costs nine quarter-rate 32-bit multiplies (measured 51 vs 84 cycles per butterfly-wave).
// Both produce the same canonical residues; the choice is a speed matter only.
// ---------------------------------------------------------------------------------------------
template <bool NARROW>
struct IntField {
    using elem = uint64_t;
    using twt = tw_t;  // twiddle as the butterflies use it
    using twl = tw_t;  // twiddle as the tables store it
    LazyMod m;
    __device__ __forceinline__ explicit IntField(const DModulus &mod) : m(mod.value) {}
    __device__ __forceinline__ twt prep(const twl &w) const { return w; }
    __device__ __forceinline__ elem from_u64(uint64_t v) const { return v; }
    __device__ __forceinline__ uint64_t to_u64(elem v) const { return v; }
    __device__ __forceinline__ static uint64_t raw(elem v) { return v; }
    __device__ __forceinline__ static elem unraw(uint64_t r) { return r; }
    __device__ __forceinline__ twt make_cst(uint64_t c, uint64_t c_shoup) const { return twt{c, c_shoup}; }
    // PAR = parity of the stage index within its kernel
    template <int PAR>
    __device__ __forceinline__ void ct(elem &x, elem &y, const twt &w) const {
        if (!NARROW)
            ct_butterfly8(x, y, w, m);
        else
            ct_butterfly16<PAR == 0>(x, y, w, m);
    }
    // end of the forward transform: lazy range -> [0,q)
    __device__ __forceinline__ elem canon_fwd(elem v) const {
        if (NARROW) v = csub(v, m.q8);
        return csub(csub(csub(v, m.q4), m.q << 1), m.q);
    }
    __device__ __forceinline__ void gs(elem &x, elem &y, const twt &w) const { gs_butterfly4(x, y, w, m); }
    // inverse lazy range [0,4q) -> [0,q)
    __device__ __forceinline__ elem canon_inv(elem v) const { return csub(csub(v, m.q << 1), m.q); }
    // last inverse stage: the sum half takes N^-1, the difference half takes itw[1] (which
    // already holds N^-1, src/util/ntt.cu:53-55); exact-quotient Shoup lands in [0,q)
    __device__ __forceinline__ void gs_last(elem &x, elem &y, const twt &w, const twt &n_inv) const {
        const uint64_t a = x, b = y;
        x = csub(mul_shoup_lazy(a + b, n_inv.x, n_inv.y, m.q), m.q);
        y = csub(mul_shoup_lazy(a + (m.q4 - b), w.x, w.y, m.q), m.q);
    }
    // canonical v -> v * c mod q, canonical
    __device__ __forceinline__ elem scale(elem v, const twt &c) const { return mul_shoup(v, c.x, c.y, m.q); }
    // canonical v, sub -> (sub - v) * c mod q, canonical
    __device__ __forceinline__ elem sub_mul(elem v, uint64_t sub, const twt &c) const {
        return mul_shoup(csub(sub + m.q - v, m.q), c.x, c.y, m.q);
    }
};

// Residues as integer-valued doubles, q < 2^50.  Every intermediate is an integer of magnitude
// <= 2^52 (or an exactly representable product error term), so each operation below is exact:
//   mulmod(y, {w, w/q}), |y| <= 2^52:  h = fl(y*w), l = y*w - h (one fma), c = rint(y * (w/q));
//       |y*w/q - c| <= 1/2 + |y| 2^-52  =>  t = (h - c*q) + l = y*w - c*q,  |t| <= 1.5q + 1
//   reduce(x): x - rint(x/q)*q, |result| <= q/2 + 1
// The tables store only w (8 bytes per twiddle: half the table traffic of the integer path's
// {w, Shoup companion}); w/q is formed as w * fl(1/q) when the twiddle is loaded.  With that
// product's extra rounding the quotient estimate is off by at most 1/2 + 1.5 |y| 2^-52, so
//   |t| <= (1/2 + 1.5 |y| / 2^52) q   (< 2q for |y| <= 4q <= 2^52).
struct FpField {
    using elem = double;
#ifndef PHA_FP_QUOTIENT_FROM_W
#define PHA_FP_QUOTIENT_FROM_W 0
#endif
#if PHA_FP_QUOTIENT_FROM_W
    using twt = double2;  // {w, ~w/q}: the quotient estimate is y * (w/q), one more product per twiddle fetched (round 1)
#else
    using twt = double;   // the quotient estimate comes from the product's high part: rint(fl(y*w) * (1/q))
#endif
    using twl = double;   // stored form
    double q, qinv;
#if PHA_FP_QUOTIENT_FROM_W
    __device__ __forceinline__ twt prep(twl w) const { return twt{w, w * qinv}; }
#else
    __device__ __forceinline__ twt prep(twl w) const { return w; }
#endif
    __device__ __forceinline__ explicit FpField(const DModulus &mod)
        : q(static_cast<double>(static_cast<int64_t>(mod.value))), qinv(1.0 / q) {}
    // 0 <= v < 2^52: splice into the mantissa of 2^52 and subtract
    __device__ __forceinline__ elem from_u64(uint64_t v) const {
        return __longlong_as_double(static_cast<long long>(v | 0x4330000000000000ull)) - 4503599627370496.0;
    }
    // integer-valued 0 <= v < 2^52
    __device__ __forceinline__ uint64_t to_u64(elem v) const {
        return static_cast<uint64_t>(__double_as_longlong(v + 4503599627370496.0)) & 0x000FFFFFFFFFFFFFull;
    }
    __device__ __forceinline__ static uint64_t raw(elem v) { return static_cast<uint64_t>(__double_as_longlong(v)); }
    __device__ __forceinline__ static elem unraw(uint64_t r) { return __longlong_as_double(static_cast<long long>(r)); }
    __device__ __forceinline__ twt make_cst(uint64_t c, uint64_t) const { return prep(from_u64(c)); }
    // y * w mod q as a representative of absolute value <= (1/2 + 3 * 2^-53 * |y| * w / q) q: h + l = y * w exactly
    // (h the rounded product, l its error), c the rounded quotient estimate (three roundings off y*w/q either way:
    // fl(y*w), 1/q and the product; or w/q, its product with y and 1/q), h - c*q exact in the FMA
    __device__ __forceinline__ elem mulmod(elem y, const twt &w) const {
#pragma clang fp contract(off)
#if PHA_FP_QUOTIENT_FROM_W
        const double h = y * w.x;
        const double l = __builtin_fma(y, w.x, -h);
        const double c = __builtin_rint(y * w.y);
#else
        const double h = y * w;
        const double l = __builtin_fma(y, w, -h);
        const double c = __builtin_rint(h * qinv);
#endif
        return __builtin_fma(-c, q, h) + l;
    }
    __device__ __forceinline__ elem reduce(elem x) const {
#pragma clang fp contract(off)
        return __builtin_fma(-__builtin_rint(x * qinv), q, x);
    }
    // a * b mod q as a representative of absolute value <= q/2 + 1, for |a|, |b| <= q: a twiddle derived from generators
    __device__ __forceinline__ elem mulred(elem a, elem b) const { return reduce(mulmod(a, prep(b))); }
    // any |v| <= 2^52 -> [0,q)
    __device__ __forceinline__ elem canon(elem v) const {
        const double r = reduce(v);
        return r < 0.0 ? r + q : r;
    }
    // Forward butterfly, bounds in units of q.  Even stage: x is reduced (<= 1/2), |y| <= a <= 4:
    // t <= 1/2 + 0.375 a, outputs b <= 1 + 0.375 a <= 2.5; odd stage (no reduction): t <= 1/2 +
    // 0.375 b, outputs <= 1/2 + 1.375 b = 1.875 + 0.516 a <= 3.94 -> the range contracts to a
    // fixed point of 3.87 q < 4q <= 2^52.  API inputs may be lazy up to 4q like the reference's
    // (they enter an even stage).
    template <int PAR>
    __device__ __forceinline__ void ct(elem &x, elem &y, const twt &w) const {
        const double t = mulmod(y, w);
        const double u = PAR == 0 ? reduce(x) : x;
        x = u + t;
        y = u - t;
    }
    __device__ __forceinline__ elem canon_fwd(elem v) const { return canon(v); }
    __device__ __forceinline__ elem canon_inv(elem v) const { return canon(v); }
    // Inverse butterfly: inputs |.| <= 2q -> |d| <= 4q <= 2^52, sum reduced to <= q/2+1, product
    // <= (1/2 + 1.5 * 4q / 2^52) q <= 2q
    __device__ __forceinline__ void gs(elem &x, elem &y, const twt &w) const {
        const double d = x - y;
        x = reduce(x + y);
        y = mulmod(d, w);
    }
    __device__ __forceinline__ void gs_last(elem &x, elem &y, const twt &w, const twt &n_inv) const {
        const double a = x + y, b = x - y;
        x = canon(mulmod(a, n_inv));
        y = canon(mulmod(b, w));
    }
    __device__ __forceinline__ elem scale(elem v, const twt &c) const { return canon(mulmod(v, c)); }
    __device__ __forceinline__ elem sub_mul(elem v, uint64_t sub, const twt &c) const {
        return canon(mulmod(from_u64(sub) - v, c));
    }
};

// Harvey forward butterfly, x,y in [0,4q) -> [0,4q)  (include/butterfly.cuh:10-22)
__device__ __forceinline__ void ct_butterfly(uint64_t &x, uint64_t &y, const tw_t &w, uint64_t q, uint64_t q2) {
    const uint64_t t = mul_shoup_lazy(y, w.x, w.y, q);
    const uint64_t u = csub(x, q2);
    x = u + t;
    y = u + q2 - t;
}

// Harvey inverse butterfly, x,y in [0,2q) -> [0,2q)  (include/butterfly.cuh:28-37)
__device__ __forceinline__ void gs_butterfly(uint64_t &x, uint64_t &y, const tw_t &w, uint64_t q, uint64_t q2) {
    const uint64_t d = x + q2 - y;
    x = csub(x + y, q2);
    y = mul_shoup_lazy(d, w.x, w.y, q);
}

}  // namespace pha
