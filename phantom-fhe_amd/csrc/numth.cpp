#include "numth.hpp"

#include <algorithm>
#include <map>

namespace pha {

uint64_t pow_mod(uint64_t base, uint64_t exp, uint64_t q) {
    uint64_t acc = 1 % q;
    base %= q;
    for (; exp; exp >>= 1) {
        if (exp & 1) acc = mul_mod(acc, base, q);
        base = mul_mod(base, base, q);
    }
    return acc;
}

uint64_t inv_mod_prime(uint64_t a, uint64_t q) { return pow_mod(a % q, q - 2, q); }

void barrett_ratio(uint64_t q, uint64_t ratio[2]) {
    // long division of 2^128 by q, 64 bits at a time
    const u128 b = static_cast<u128>(1) << 64;
    const uint64_t hi = static_cast<uint64_t>(b / q);
    const uint64_t r = static_cast<uint64_t>(b % q);
    ratio[1] = hi;
    ratio[0] = static_cast<uint64_t>((static_cast<u128>(r) << 64) / q);
}

bool is_prime_u64(uint64_t n) {
    if (n < 2) return false;
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (n % p == 0) return n == p;
    }
    uint64_t d = n - 1;
    int s = 0;
    while (!(d & 1)) {
        d >>= 1;
        ++s;
    }
    // Sinclair's 7-base set is a deterministic Miller-Rabin test for all n < 2^64.
    for (uint64_t a : {2ull, 325ull, 9375ull, 28178ull, 450775ull, 9780504ull, 1795265022ull}) {
        uint64_t x = pow_mod(a, d, n);
        if (x == 0 || x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < s && witness; ++i) {
            x = mul_mod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

bool find_primes(uint64_t ntt_size, int bit_size, size_t count, std::vector<uint64_t> &out) {
    out.clear();
    if (bit_size < 2 || bit_size > 61) return false;
    const uint64_t step = 2 * ntt_size;
    const uint64_t top = uint64_t(1) << bit_size;
    if (top <= step) return false;
    const uint64_t floor_v = uint64_t(1) << (bit_size - 1);
    for (uint64_t cand = top - step + 1; out.size() < count && cand > floor_v; cand -= step) {
        if (is_prime_u64(cand)) out.push_back(cand);
    }
    return out.size() == count;
}

bool coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out) {
    std::map<int, std::vector<uint64_t>> pool;
    std::map<int, size_t> want;
    for (size_t i = 0; i < count; ++i) ++want[bit_sizes[i]];
    for (auto &kv : want) {
        if (!find_primes(n, kv.first, kv.second, pool[kv.first])) return false;
    }
    // the generator found the primes in descending order; they are handed out from the back
    for (size_t i = 0; i < count; ++i) {
        auto &v = pool[bit_sizes[i]];
        out[i] = v.back();
        v.pop_back();
    }
    return true;
}

bool minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t &root) {
    if (degree < 2 || (q - 1) % degree) return false;
    const uint64_t cofactor = (q - 1) / degree;
    uint64_t r = 0;
    for (uint64_t x = 2; x < 4096 && !r; ++x) {
        const uint64_t cand = pow_mod(x, cofactor, q);
        if (pow_mod(cand, degree / 2, q) == q - 1) r = cand;
    }
    if (!r) return false;
    // every primitive root is an odd power of r: scan them all for the minimum
    const uint64_t r2 = mul_mod(r, r, q);
    uint64_t best = r, cur = r;
    for (uint64_t i = 1; i < degree / 2; ++i) {
        cur = mul_mod(cur, r2, q);
        best = std::min(best, cur);
    }
    root = best;
    return true;
}

bool build_limb_tables(int log_n, uint64_t q, LimbTables &t) {
    const size_t n = size_t(1) << log_n;
    t.q = q;
    barrett_ratio(q, t.ratio);
    if (!minimal_primitive_root(2 * n, q, t.psi)) return false;
    const uint64_t psi_inv = inv_mod_prime(t.psi, q);
    t.tw.assign(n, 0);
    t.itw.assign(n, 0);
    t.tw_shoup.assign(n, 0);
    t.itw_shoup.assign(n, 0);
    uint64_t f = 1, b = 1;
    for (size_t e = 0; e < n; ++e) {
        const uint32_t slot = bit_reverse(static_cast<uint32_t>(e), log_n);
        t.tw[slot] = f;
        t.itw[slot] = b;
        f = mul_mod(f, t.psi, q);
        b = mul_mod(b, psi_inv, q);
    }
    t.n_inv = inv_mod_prime(n % q, q);
    t.n_inv_shoup = shoup(t.n_inv, q);
    t.itw[1] = mul_mod(t.itw[1], t.n_inv, q);
    for (size_t i = 0; i < n; ++i) {
        t.tw_shoup[i] = shoup(t.tw[i], q);
        t.itw_shoup[i] = shoup(t.itw[i], q);
    }
    return true;
}

void build_generator_table(const LimbTables &t, int log_n, bool inverse, uint64_t *out) {
    const size_t n = size_t(1) << log_n, groups = n >> 4, tiles = n >> 12;
    const uint64_t q = t.q;
    const std::vector<uint64_t> &tw = inverse ? t.itw : t.tw;      // every index used here is >= 8: itw[1]'s N^-1 is not involved
    const uint64_t root = inverse ? inv_mod_prime(t.psi, q) : t.psi;
    for (size_t i = 0; i < kGenStride; i++) out[i] = 0;
    for (int s = 0; s < 4; s++)
        for (uint32_t jb = 0; jb < (1u << s); jb++)
            out[kGenOmega + (size_t(1) << s) - 1 + jb] = pow_mod(root, static_cast<uint64_t>(bit_reverse(jb, s)) << (log_n - s), q);
    const uint64_t theta0_inv = inv_mod_prime(tw[groups << 3], q);
    for (size_t ti = 0; ti < tiles; ti++) out[kGenAlpha + ti] = tw[(groups + ti * 256) << 3];
    for (size_t l = 0; l < 256 && l < groups; l++) out[kGenBeta + l] = mul_mod(tw[(groups + l) << 3], theta0_inv, q);
}

void galois_permutation(int log_n, uint32_t galois_elt, uint32_t *table) {
    const uint32_t n = 1u << log_n;
    for (uint32_t i = 0; i < n; ++i) {
        // output slot i holds the evaluation at psi^(2*brev(i)+1); X -> X^g moves it to the
        // slot of exponent g*(2*brev(i)+1) mod 2n
        const uint64_t odd = 2ull * bit_reverse(i, log_n) + 1;
        const uint64_t mapped = (odd * galois_elt) & (2ull * n - 1);
        table[i] = bit_reverse(static_cast<uint32_t>(mapped >> 1), log_n);
    }
}

bool galois_elt_from_step(int log_n, int step, uint32_t &elt) {
    const uint32_t n = 1u << log_n, m = 2 * n;
    if (step == 0) {
        elt = m - 1;
        return true;
    }
    const uint32_t mag = static_cast<uint32_t>(step < 0 ? -static_cast<int64_t>(step) : step);
    if (mag >= n / 2) return false;
    uint32_t reps = step < 0 ? n / 2 - mag : mag;
    uint64_t g = 1;
    while (reps--) g = (g * 5) & (m - 1);
    elt = static_cast<uint32_t>(g);
    return true;
}

uint64_t product_mod_except(const uint64_t *base, size_t s, size_t e, size_t skip, uint64_t m) {
    uint64_t acc = 1 % m;
    for (size_t k = s; k < e; ++k) {
        if (k == skip) continue;
        acc = mul_mod(acc, base[k] % m, m);
    }
    return acc;
}

void build_level_constants(const std::vector<uint64_t> &moduli_QP, size_t size_P, size_t size_Ql, LevelConstants &lc) {
    const size_t size_QP = moduli_QP.size(), size_Q = size_QP - size_P;
    const uint64_t *Q = moduli_QP.data();
    const uint64_t *P = Q + size_Q;
    lc.size_Ql = size_Ql;
    lc.alpha = size_P;
    lc.beta = size_P ? (size_Ql + size_P - 1) / size_P : 0;

    lc.inv_q_last_mod_q.resize(size_Ql ? size_Ql - 1 : 0);
    for (size_t i = 0; i + 1 < size_Ql; ++i) lc.inv_q_last_mod_q[i] = inv_mod_prime(Q[size_Ql - 1] % Q[i], Q[i]);

    lc.bigP_mod_q.resize(size_Ql);
    lc.bigPInv_mod_q.resize(size_Ql);
    for (size_t i = 0; i < size_Ql; ++i) {
        lc.bigP_mod_q[i] = product_mod_except(P, 0, size_P, SIZE_MAX, Q[i]);
        lc.bigPInv_mod_q[i] = inv_mod_prime(lc.bigP_mod_q[i], Q[i]);
    }

    // Q_l ∪ P as one list
    std::vector<uint64_t> qlp(Q, Q + size_Ql);
    qlp.insert(qlp.end(), P, P + size_P);

    lc.partQlHatInv_mod_Ql_concat.resize(size_Ql);
    lc.modup_matrix.assign(lc.beta, {});
    for (size_t d = 0; d < lc.beta; ++d) {
        const size_t s = d * size_P, e = std::min(s + size_P, size_Ql), len = e - s;
        for (size_t i = s; i < e; ++i) {
            lc.partQlHatInv_mod_Ql_concat[i] = inv_mod_prime(product_mod_except(Q, s, e, i, Q[i]), Q[i]);
        }
        auto &mat = lc.modup_matrix[d];
        mat.reserve((qlp.size() - len) * len);
        for (size_t j = 0; j < qlp.size(); ++j) {
            if (j >= s && j < e) continue;
            for (size_t i = s; i < e; ++i) mat.push_back(product_mod_except(Q, s, e, i, qlp[j]));
        }
    }

    lc.PHatInv_mod_p.resize(size_P);
    for (size_t k = 0; k < size_P; ++k) lc.PHatInv_mod_p[k] = inv_mod_prime(product_mod_except(P, 0, size_P, k, P[k]), P[k]);
    lc.moddown_matrix.resize(size_Ql * size_P);
    for (size_t j = 0; j < size_Ql; ++j)
        for (size_t k = 0; k < size_P; ++k) lc.moddown_matrix[j * size_P + k] = product_mod_except(P, 0, size_P, k, Q[j]);
}

// ---- multi-word helpers for the CRT tables ------------------------------------------------------
static void mw_mul_word(std::vector<uint64_t> &a, uint64_t w) {  // a *= w, a has room
    uint64_t carry = 0;
    for (auto &x : a) {
        const u128 p = static_cast<u128>(x) * w + carry;
        x = static_cast<uint64_t>(p);
        carry = static_cast<uint64_t>(p >> 64);
    }
}

void build_bconv_mfma_table(const uint64_t *matrix, const uint64_t *moduli, size_t ibase, size_t obase, BconvMfmaTable &out) {
    out = BconvMfmaTable{};
    if (ibase == 0 || ibase > 15 || obase == 0) return;
    for (size_t j = 0; j < obase; j++)
        if (moduli[j] < (1ull << 34) || (moduli[j] >> 61) != 0) return;
    typedef unsigned __int128 u128;
    const uint32_t ksteps = ibase <= 7 ? 1 : 2;
    const size_t quads = (obase + 3) / 4, quad_bytes = 2 * ksteps * 1024;
    out.ksteps = ksteps;
    out.quads = static_cast<uint32_t>(quads);
    out.table.assign(quads * quad_bytes, 0);
    out.consts.assign(quads * 4 * 2, 0);
    auto digits = [](uint64_t x, int8_t d[8]) {   // balanced base-256 digits of x < 2^62
        for (int s = 0; s < 8; s++) {
            int64_t v = static_cast<int64_t>(x & 255);
            if (v >= 128) v -= 256;
            d[s] = static_cast<int8_t>(v);
            x = (x - static_cast<uint64_t>(v)) >> 8;   // exact: x - v is a multiple of 256 (wraps only when x < |v|, x small)
        }
    };
    for (size_t j = 0; j < quads * 4; j++) {
        const size_t quad = j / 4, g = j % 4;
        if (j >= obase) {   // padding rows: a harmless modulus, all-zero digits
            out.consts[j * 2] = (1ull << 34) + 1;
            out.consts[j * 2 + 1] = static_cast<uint64_t>((static_cast<u128>(1) << 96) / out.consts[j * 2]);
            continue;
        }
        const uint64_t p = moduli[j];
        out.consts[j * 2] = p;
        out.consts[j * 2 + 1] = static_cast<uint64_t>((static_cast<u128>(1) << 96) / p);
        u128 sum = 0;                 // sum over (i, a) of (2^(8a) M[j][i]) mod p
        auto put = [&](size_t slot_i, size_t a, uint64_t value) {
            int8_t d[8];
            digits(value, d);
            const size_t kstep = slot_i / 8, gk = (slot_i % 8) / 2, jb = (slot_i % 2) * 8 + a;
            for (int s = 0; s < 8; s++) {
                const size_t t = s / 4, r = s % 4, lane = 16 * gk + 4 * g + r;
                out.table[quad * quad_bytes + ((t * ksteps + kstep) * 64 + lane) * 16 + jb] = static_cast<uint8_t>(d[s]);
            }
        };
        for (size_t i = 0; i < ibase; i++) {
            const uint64_t m = matrix[j * ibase + i] % p;
            for (size_t a = 0; a < 8; a++) {
                const uint64_t v = static_cast<uint64_t>((static_cast<u128>(m) << (8 * a)) % p);
                sum += v;
                put(i, a, v);
            }
        }
        // the constant slot: input index ibase, byte 0, against a data byte 1
        const u128 bias = static_cast<u128>(kBmfBiasLimb) + (static_cast<u128>(kBmfBiasLimb) << 32);
        const uint64_t c = static_cast<uint64_t>((sum * 128 % p + p - bias % p) % p);
        put(ibase, 0, c);
    }
}

void build_crt_tables(const uint64_t *moduli, size_t l, CrtTables &t) {
    t.l = l;
    t.big_q.assign(l, 0);
    t.big_q[0] = 1;
    for (size_t j = 0; j < l; j++) mw_mul_word(t.big_q, moduli[j]);
    t.qhat.assign(l * l, 0);
    t.inv_qhat.assign(l, 0);
    for (size_t j = 0; j < l; j++) {
        std::vector<uint64_t> h(l, 0);
        h[0] = 1;
        for (size_t k = 0; k < l; k++)
            if (k != j) mw_mul_word(h, moduli[k]);
        std::copy(h.begin(), h.end(), t.qhat.begin() + j * l);
        t.inv_qhat[j] = inv_mod_prime(product_mod_except(moduli, 0, l, j, moduli[j]), moduli[j]);
    }
    // (Q + 1) >> 1
    t.half_q = t.big_q;
    uint64_t carry = 1;
    for (auto &x : t.half_q) {
        const uint64_t s = x + carry;
        carry = s < x ? 1 : 0;
        x = s;
    }
    for (size_t i = 0; i < l; i++) t.half_q[i] = (t.half_q[i] >> 1) | (i + 1 < l ? t.half_q[i + 1] << 63 : 0);
}

}  // namespace pha
