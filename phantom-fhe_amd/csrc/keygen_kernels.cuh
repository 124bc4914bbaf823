// Samplers and element-wise kernels of key generation, symmetric encryption and decryption
// (SURVEY.md §8f.1).  Distributions follow src/prng.cu of the reference (ternary :142-163,
// centred binomial :219-242, uniform by rejection :174-207).  Element-wise arithmetic restates src/polymath.cu.
//
// Randomness: a keyed stream cipher, as in the reference (Salsa20 under a std::random_device key, src/prng.cu:17,157,192,231).
//     rnd(key, seed, ctr) = first 64 bits of the ChaCha20 block (RFC 8439, 20 rounds) under the 256-bit `key` whose
//                           counter / nonce words are (ctr_lo, ctr_hi, seed_lo, seed_hi)
// `key` is the process's sampler key: 32 bytes of OS entropy drawn when the library first samples (pha_set_sampler_key
// replaces it; tests and the oracle fix it to compare key material bit for bit).  The 64-bit `seed` arguments of the C ABI
// are NONCES that name a stream under that key, not secrets: everything published (the uniform `a` of a ciphertext or key)
// is ChaCha20 output, from which neither the key nor any other stream (secret key, errors, other nonces) can be derived.
// The generator stays random-access, so every coefficient is sampled independently.
#pragma once

#include "modarith.cuh"

namespace pha {

struct SamplerKey {
    uint32_t w[8];
};

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
#define PHA_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);

// ChaCha20 block function, first `words` (<= 16) output words
__host__ __device__ __forceinline__ void chacha20_block(const SamplerKey &key, uint64_t seed, uint64_t ctr, uint32_t *out, int words) {
    const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                             key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7],
                             static_cast<uint32_t>(ctr), static_cast<uint32_t>(ctr >> 32), static_cast<uint32_t>(seed),
                             static_cast<uint32_t>(seed >> 32)};
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7], x8 = in[8],
             x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
    for (int r = 0; r < 10; r++) {
        PHA_CHACHA_QR(x0, x4, x8, x12) PHA_CHACHA_QR(x1, x5, x9, x13) PHA_CHACHA_QR(x2, x6, x10, x14) PHA_CHACHA_QR(x3, x7, x11, x15)
        PHA_CHACHA_QR(x0, x5, x10, x15) PHA_CHACHA_QR(x1, x6, x11, x12) PHA_CHACHA_QR(x2, x7, x8, x13) PHA_CHACHA_QR(x3, x4, x9, x14)
    }
    const uint32_t x[16] = {x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
    for (int i = 0; i < words; i++) out[i] = x[i] + in[i];
}

__host__ __device__ __forceinline__ uint64_t rnd_at(const SamplerKey &key, uint64_t seed, uint64_t ctr) {
    uint32_t o[2];
    chacha20_block(key, seed, ctr, o, 2);
    return (static_cast<uint64_t>(o[1]) << 32) | o[0];
}

enum : int { SAMPLE_TERNARY = 0, SAMPLE_ERROR = 1, SAMPLE_UNIFORM = 2 };

// out[l][k] for l < limbs (limb l is modulo mod[l]); one lane per coefficient, looping over limbs
// for the two small distributions so that the draw is shared by every limb
template <int KIND>
__global__ __launch_bounds__(256) void sample_poly_kernel(uint64_t *out, const DModulus *mod, uint32_t limbs, uint32_t n,
                                                          SamplerKey key, uint64_t seed) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (KIND == SAMPLE_UNIFORM) {
        const uint32_t l = blockIdx.y;
        const DModulus m = mod[l];
        const uint64_t total = static_cast<uint64_t>(limbs) * n;
        // largest r accepted: 2^64 - 1 - ((2^64 - 1) mod q) - 1 (src/prng.cu:188)
        const uint64_t max_multiple = ~0ull - barrett64(~0ull, m.value, m.ratio1) - 1;
        uint64_t ctr = static_cast<uint64_t>(l) * n + k, r;
        while ((r = rnd_at(key, seed, ctr)) > max_multiple) ctr += total;
        out[static_cast<size_t>(l) * n + k] = barrett64(r, m.value, m.ratio1);
    } else {
        const uint64_t r = rnd_at(key, seed, k);
        int v;
        if (KIND == SAMPLE_TERNARY)
            v = static_cast<int>((r & 0xFF) % 3) - 1;
        else
            v = __popcll(r & 0x1FFFFFull) - __popcll((r >> 24) & 0x1FFFFFull);
        for (uint32_t l = 0; l < limbs; l++)
            out[static_cast<size_t>(l) * n + k] = v < 0 ? mod[l].value - static_cast<uint64_t>(-v) : static_cast<uint64_t>(v);
    }
}

enum : int { EW_MUL = 0, EW_MUL_ADD = 1, EW_MUL_ADD_NEG = 2, EW_ADD = 3, EW_MUL_CONST_ADD = 4 };

// limb-wise element-wise arithmetic, 16 B per lane (src/polymath.cu: multiply_rns_poly,
// multiply_and_add_rns_poly, multiply_and_add_negate_rns_poly :345-380, add_rns_poly,
// multiply_temp_mod_and_add_rns_poly :323-343); limb l of the buffers is modulo mod[l]
template <int OP>
__global__ __launch_bounds__(256) void elementwise_kernel(uint64_t *dst, const uint64_t *a, const uint64_t *b,
                                                          const uint64_t *c, const DModulus *mod, const uint64_t *cst,
                                                          uint32_t n) {
    const uint32_t per_limb = n / 512;
    const uint32_t l = blockIdx.x / per_limb;
    const size_t id = static_cast<size_t>(l) * n + ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const DModulus m = mod[l];
    const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a + id);
    ulonglong2 y = {0, 0}, z = {0, 0}, r;
    if (OP != EW_MUL_CONST_ADD) y = *reinterpret_cast<const ulonglong2 *>(b + id);
    if (OP == EW_MUL_ADD || OP == EW_MUL_ADD_NEG || OP == EW_MUL_CONST_ADD) z = *reinterpret_cast<const ulonglong2 *>(c + id);
    if (OP == EW_ADD) {
        r.x = csub(x.x + y.x, m.value);
        r.y = csub(x.y + y.y, m.value);
    } else {
        if (OP == EW_MUL_CONST_ADD) y.x = y.y = cst[l];
        r.x = mul_barrett(x.x, y.x, m);
        r.y = mul_barrett(x.y, y.y, m);
        if (OP != EW_MUL) {
            r.x = csub(r.x + z.x, m.value);
            r.y = csub(r.y + z.y, m.value);
        }
        if (OP == EW_MUL_ADD_NEG) {
            r.x = r.x ? m.value - r.x : 0;
            r.y = r.y ? m.value - r.y : 0;
        }
    }
    *reinterpret_cast<ulonglong2 *>(dst + id) = r;
}

enum : int { PB_ADD = 0, PB_SUB = 1, PB_MUL = 2, PB_NEG = 3, PB_MAC = 4 };  // MAC: dst = c + a * b

// Ciphertext / plaintext element-wise evaluator arithmetic (src/polymath.cu add_rns_poly,
// sub_rns_poly, multiply_rns_poly, negate_rns_poly as used by src/evaluate.cu:20-105,442-560):
// a, dst are [polys][limbs][n]; b is the same shape, or one [limbs][n] poly applied to every poly
// of a when b_broadcast (plaintext operand).  16 B per lane.
template <int OP>
__global__ __launch_bounds__(256) void poly_binary_kernel(uint64_t *dst, const uint64_t *a, const uint64_t *b,
                                                          const DModulus *mod, uint32_t limbs, uint32_t n,
                                                          uint32_t b_broadcast, const uint64_t *c = nullptr) {
    const uint32_t per_limb = n / 512;
    const uint32_t row = blockIdx.x / per_limb, l = row % limbs;
    const uint32_t k = ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const size_t id = static_cast<size_t>(row) * n + k;
    const DModulus m = mod[l];
    const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a + id);
    ulonglong2 y = {0, 0}, r;
    if (OP != PB_NEG) y = *reinterpret_cast<const ulonglong2 *>(b + (b_broadcast ? static_cast<size_t>(l) * n + k : id));
    if (OP == PB_ADD) {
        r.x = csub(x.x + y.x, m.value);
        r.y = csub(x.y + y.y, m.value);
    } else if (OP == PB_SUB) {
        r.x = csub(x.x + m.value - y.x, m.value);
        r.y = csub(x.y + m.value - y.y, m.value);
    } else if (OP == PB_MUL) {
        r.x = mul_barrett(x.x, y.x, m);
        r.y = mul_barrett(x.y, y.y, m);
    } else if (OP == PB_MAC) {
        const ulonglong2 z = *reinterpret_cast<const ulonglong2 *>(c + id);
        r.x = csub(mul_barrett(x.x, y.x, m) + z.x, m.value);
        r.y = csub(mul_barrett(x.y, y.y, m) + z.y, m.value);
    } else {
        r.x = x.x ? m.value - x.x : 0;
        r.y = x.y ? m.value - x.y : 0;
    }
    *reinterpret_cast<ulonglong2 *>(dst + id) = r;
}

// Bootstrapper::modraise_inplace (src/boot/Bootstrapper.cu:2801-2861): a coefficient-form residue x
// modulo q0 is lifted to its centred representative (x - q0 if x > q0/2) and reduced modulo every
// limb of the top level.  src: [polys][n] (limb 0 in coefficient form), dst: [polys][limbs][n].
__global__ __launch_bounds__(256) void mod_raise_kernel(uint64_t *dst, const uint64_t *src, const DModulus *mod,
                                                        uint32_t limbs, uint32_t n) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    const uint64_t q0 = mod[0].value, x = src[static_cast<size_t>(p) * n + k];
    const bool high = x > (q0 >> 1);
    for (uint32_t l = 0; l < limbs; l++) {
        const DModulus m = mod[l];
        uint64_t r = barrett64(x, m.value, m.ratio1);
        if (high) r = csub(r + m.value - barrett64(q0, m.value, m.ratio1), m.value);
        dst[(static_cast<size_t>(p) * limbs + l) * n + k] = r;
    }
}

}  // namespace pha
