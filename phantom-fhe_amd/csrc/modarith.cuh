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
