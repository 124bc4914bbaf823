// Host-side number theory and table generation for the MI355X RNS hot path.
// Mirrors the values produced by the reference's host generators (cited per function) so that
// device tables are interchangeable with Phantom's DNTTTable / DRNSTool contents.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace pha {

using u128 = unsigned __int128;

inline uint64_t mul_mod(uint64_t a, uint64_t b, uint64_t q) { return static_cast<uint64_t>(static_cast<u128>(a) * b % q); }
uint64_t pow_mod(uint64_t base, uint64_t exp, uint64_t q);
uint64_t inv_mod_prime(uint64_t a, uint64_t q);
// floor(w * 2^64 / q): Shoup companion (include/util/uintarithsmallmod.h:141-146)
inline uint64_t shoup(uint64_t w, uint64_t q) { return static_cast<uint64_t>((static_cast<u128>(w) << 64) / q); }
// floor(2^128 / q) -> {lo, hi} (src/modulus.cu:28-40)
void barrett_ratio(uint64_t q, uint64_t ratio[2]);
bool is_prime_u64(uint64_t n);
// src/util/numth.cu:209-235
bool find_primes(uint64_t ntt_size, int bit_size, size_t count, std::vector<uint64_t> &out);
// src/modulus.cu:79-110
bool coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out);
// src/util/numth.cu:311-333: smallest primitive degree-th root of unity
bool minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t &root);

inline uint32_t bit_reverse(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

struct LimbTables {
    uint64_t q = 0;
    uint64_t ratio[2] = {0, 0};
    uint64_t psi = 0;
    uint64_t n_inv = 0, n_inv_shoup = 0;
    std::vector<uint64_t> tw, tw_shoup, itw, itw_shoup;  // canonical (reference) order
};
// src/util/ntt.cu:11-56
bool build_limb_tables(int log_n, uint64_t q, LimbTables &t);

// Generators of the LAST register pass's twiddles (no reference counterpart: the reference reads all N twiddles of a
// limb, src/ntt/fntt_2d.cu:148-166).  The last four stages of a transform work on groups of 16 adjacent coefficients;
// group g (node G = n/16 + g) uses, in stage s and sub-block jb < 2^s, the table entry tw[(G << s) + jb] =
// psi^(brev_s(jb) 2^(logn-s) + B 2^(3-s)) with B = brev_(logn-3)(G), i.e.  theta_g^(2^(3-s)) * omega[s][jb]  with
// theta_g = tw[G << 3] and 15 per-limb constants omega; and with g = tile * 256 + t,  theta_g = alpha[tile] * beta[t].
// Layout per limb and direction (kGenStride words): omega[16] (slot (1 << s) - 1 + jb; slot 15 unused), alpha[32], beta[256].
constexpr size_t kGenOmega = 0, kGenAlpha = 16, kGenBeta = 48, kGenStride = 304;
void build_generator_table(const LimbTables &t, int log_n, bool inverse, uint64_t *out);

// include/galois.cuh:49-64
void galois_permutation(int log_n, uint32_t galois_elt, uint32_t *table);
// src/galois.cu:42-75; returns false for an out-of-range step
bool galois_elt_from_step(int log_n, int step, uint32_t &elt);

// prod_{k in [s,e), k != skip} base[k] mod m   (skip = SIZE_MAX for the full product)
uint64_t product_mod_except(const uint64_t *base, size_t s, size_t e, size_t skip, uint64_t m);

// DRNSTool level constants, src/rns.cu:60-192
struct LevelConstants {
    size_t size_Ql = 0, alpha = 0, beta = 0;
    std::vector<uint64_t> inv_q_last_mod_q;            // [size_Ql-1]
    std::vector<uint64_t> bigP_mod_q, bigPInv_mod_q;   // [size_Ql]
    std::vector<uint64_t> partQlHatInv_mod_Ql_concat;  // [size_Ql]
    // per digit d: matrix [obase = size_QlP - |digit|][ibase = |digit|] of QHat_{d,i} mod p_j
    // (src/util/rns.cu:452-461), output order = Q_l ∪ P with the digit erased (src/rns.cu:157-163)
    std::vector<std::vector<uint64_t>> modup_matrix;
    // P -> Q_l: PHatInv_k mod p_k [alpha], matrix [size_Ql][alpha] of PHat_k mod q_j
    std::vector<uint64_t> PHatInv_mod_p;
    std::vector<uint64_t> moddown_matrix;
};
void build_level_constants(const std::vector<uint64_t> &moduli_QP, size_t size_P, size_t size_Ql, LevelConstants &lc);

// Base conversion on the matrix cores (rns_kernels.cuh: bconv_mfma_kernel; no reference counterpart, the reference's
// bconv_matmul_*_kernel, src/rns_bconv.cu:143-168, multiplies 64-bit words on the vector ALU).  The product
//   out_j = sum_i y_i M[j][i]  mod p_j
// is rewritten over the bytes of the inputs: y_i = sum_a u_ia 2^(8a), t_ia = u_ia - 128 (a signed byte), and
//   out_j = sum_s 2^(8s) sum_(i,a) t_ia w[j][i][a][s] + C_j   (mod p_j),   s = 0..7,
// where w[j][i][a][.] are the eight BALANCED base-256 digits (in [-128, 127]) of (2^(8a) M[j][i]) mod p_j and C_j =
// 128 sum_(i,a) (2^(8a) M[j][i] mod p_j).  The inner sums are one i8 matrix product with K = 8 ibase + 1: the extra
// slot multiplies a constant byte 1 by the digits of (C_j - kBmfBias) mod p_j, kBmfBias being what the kernel adds to
// keep its accumulator non-negative.  Table layout, per QUAD of four outputs (j = 4 quad + g): [tile t = 0, 1][k-step]
// [lane = 16 g' + 4 g + r][16 bytes]: the A operand of v_mfma_i32_16x16x64_i8 whose row 4 g + r is digit s = 4 t + r
// of output j; byte jb of lane group g' is slot (input i = 8 kstep + 2 g' + jb / 8, byte a = jb % 8).
// consts: [quads * 4][2] = {p_j, floor(2^96 / p_j)}.  ksteps = 0 (no table): ibase > 15 or a modulus below 2^34 (the
// kernel's quotient estimate is then more than one short).
constexpr uint32_t kBmfBiasWord = 1u << 30;       // what the partial sums of digits 0, 2, 4, 6 start from
// ... which adds this much to each of the two 32-bit-spaced halves of the accumulator
constexpr uint64_t kBmfBiasLimb = static_cast<uint64_t>(kBmfBiasWord) + (static_cast<uint64_t>(kBmfBiasWord) << 16);
struct BconvMfmaTable {
    uint32_t ksteps = 0, quads = 0;
    std::vector<uint8_t> table;       // [quads][2][ksteps][64][16]
    std::vector<uint64_t> consts;     // [quads * 4][2]
};
// matrix [obase][ibase] (row j reduced mod moduli[j]); moduli [obase]
void build_bconv_mfma_table(const uint64_t *matrix, const uint64_t *moduli, size_t ibase, size_t obase, BconvMfmaTable &out);

// CRT composition tables of the base q_0..q_{l-1} (src/util/rns.cu RNSBase::initialize, used by
// compose_array src/rns_base.cu:236-402): multi-word little-endian integers of l words
struct CrtTables {
    size_t l = 0;
    std::vector<uint64_t> big_q;        // [l]      Q = prod q_j
    std::vector<uint64_t> half_q;       // [l]      (Q + 1) / 2: upper-half threshold (context_data.upper_half_threshold())
    std::vector<uint64_t> qhat;         // [l][l]   Q / q_j
    std::vector<uint64_t> inv_qhat;     // [l]      (Q / q_j)^-1 mod q_j
};
void build_crt_tables(const uint64_t *moduli, size_t l, CrtTables &t);

}  // namespace pha
