/*
 * oracle.c — scalar CPU restatement of Phantom-FHE's RNS polynomial-arithmetic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (phantom-fhe_amd/, include/) may call,
 * link or import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY PIN STATUS: the reference is CUDA-only (no CPU path) and ships no integer-level
 * golden vectors, so kernel-level parity is "unpinned" by reference fixtures.  The table
 * generators here are pinned to the known-answer anchors recorded in SURVEY.md §8(c)
 * (primes, psi, tw[1..3], itw[1], N^-1, Barrett ratio) and every dataflow function is pinned
 * by mathematical self-checks in tests/ (definition of the transform, INTT∘NTT = id,
 * convolution theorem, decrypt∘keyswitch correctness with real keys).  All stored results in
 * the reference are canonical residues in [0,q), so any correct implementation is bit-identical.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------
 * small modular arithmetic (plain, obviously-correct forms)
 * ---------------------------------------------------------------------------------------- */

static inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod(r, a, q);
        a = mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

/* q prime */
uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a, q - 2, q); }

/* include/util/uintarithsmallmod.h:141-146 compute_shoup: floor(w * 2^64 / q) */
uint64_t orc_shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

/* src/modulus.cu:28-40: const_ratio = floor(2^128 / q) as (lo, hi) */
void orc_barrett_ratio(uint64_t q, uint64_t ratio[2]) {
    u128 top = (u128)1 << 64;
    uint64_t hi = (uint64_t)(top / q);
    uint64_t rem = (uint64_t)(top % q);
    uint64_t lo = (uint64_t)(((u128)rem << 64) / q);
    ratio[0] = lo;
    ratio[1] = hi;
}

/* src/util/numth.cu:160-207 is_prime — the reference runs a randomized Miller-Rabin; the
 * deterministic base set below decides primality exactly for every n < 2^64. */
int orc_is_prime(uint64_t n) {
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    uint64_t d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        uint64_t x = orc_powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int composite = 1;
        for (int k = 1; k < r; k++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { composite = 0; break; }
        }
        if (composite) return 0;
    }
    return 1;
}

/* src/util/numth.cu:209-235 get_primes: walk down from 2^bits - 2N + 1 in steps of 2N */
int orc_get_primes(uint64_t ntt_size, int bit_size, size_t count, uint64_t *out) {
    uint64_t factor = 2 * ntt_size;
    uint64_t value = ((uint64_t)1 << bit_size) - factor + 1;
    uint64_t lower = (uint64_t)1 << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower) {
        if (orc_is_prime(value)) out[found++] = value;
        value -= factor;
    }
    return found == count ? 0 : -1;
}

/* src/modulus.cu:79-110 CoeffModulus::Create: per bit size the primes are found in
 * descending order and handed out from the back (smallest first). */
int orc_coeff_modulus_create(uint64_t n, const int *bit_sizes, size_t count, uint64_t *out) {
    for (size_t i = 0; i < count; i++) out[i] = 0;
    for (size_t i = 0; i < count; i++) {
        if (out[i]) continue;
        int bits = bit_sizes[i];
        size_t c = 0;
        for (size_t j = i; j < count; j++) c += (bit_sizes[j] == bits);
        uint64_t *tmp = (uint64_t *)malloc(c * sizeof(uint64_t));
        if (orc_get_primes(n, bits, c, tmp)) { free(tmp); return -1; }
        size_t k = c;
        for (size_t j = i; j < count; j++)
            if (bit_sizes[j] == bits) out[j] = tmp[--k];
        free(tmp);
    }
    return 0;
}

/* src/util/numth.cu:278-333: numerically smallest primitive degree-th root of unity mod q.
 * The reference starts from a random primitive root; the minimum over all odd powers does not
 * depend on the start, so a deterministic start is used here. */
int orc_min_primitive_root(uint64_t degree, uint64_t q, uint64_t *out) {
    if ((q - 1) % degree) return -1;
    uint64_t quot = (q - 1) / degree;
    uint64_t root = 0;
    for (uint64_t g = 2; g < 1000; g++) {
        uint64_t r = orc_powmod(g, quot, q);
        if (orc_powmod(r, degree >> 1, q) == q - 1) { root = r; break; }
    }
    if (!root) return -1;
    uint64_t gen_sq = mulmod(root, root, q);
    uint64_t cur = root;
    for (uint64_t i = 0; i < degree; i++) {
        if (cur < root) root = cur;
        cur = mulmod(cur, gen_sq, q);
    }
    *out = root;
    return 0;
}

static inline uint32_t brev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* src/util/ntt.cu:11-56 NTT::NTT: bit-reversed psi powers, Shoup companions,
 * N^-1 folded into itw[1]. */
int orc_ntt_tables(int log_n, uint64_t q, uint64_t *tw, uint64_t *tw_shoup, uint64_t *itw,
                   uint64_t *itw_shoup, uint64_t *n_inv, uint64_t *n_inv_shoup) {
    size_t n = (size_t)1 << log_n;
    uint64_t psi;
    if (orc_min_primitive_root(2 * n, q, &psi)) return -1;
    uint64_t ipsi = orc_invmod(psi, q);
    uint64_t p = psi, ip = ipsi;
    for (size_t i = 1; i < n; i++) {
        uint32_t r = brev((uint32_t)i, log_n);
        tw[r] = p;
        itw[r] = ip;
        p = mulmod(p, psi, q);
        ip = mulmod(ip, ipsi, q);
    }
    tw[0] = 1;
    itw[0] = 1;
    uint64_t ninv = orc_invmod(n % q, q);
    itw[1] = mulmod(itw[1], ninv, q);
    for (size_t i = 0; i < n; i++) {
        tw_shoup[i] = orc_shoup(tw[i], q);
        itw_shoup[i] = orc_shoup(itw[i], q);
    }
    *n_inv = ninv;
    *n_inv_shoup = orc_shoup(ninv, q);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * context: key-level modulus chain Q ∪ P with one table set per limb
 * (src/context.cu:148-189: one DNTTTable for all QP limbs)
 * ---------------------------------------------------------------------------------------- */

struct orc_ctx {
    int log_n;
    size_t n, size_QP, size_P;
    uint64_t *q;
    uint64_t *tw, *tw_shoup, *itw, *itw_shoup; /* [size_QP][n] */
    uint64_t *n_inv, *n_inv_shoup;
};

orc_ctx *orc_ctx_create(int log_n, const uint64_t *moduli, size_t size_QP, size_t size_P) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->log_n = log_n;
    c->n = (size_t)1 << log_n;
    c->size_QP = size_QP;
    c->size_P = size_P;
    size_t n = c->n;
    c->q = (uint64_t *)malloc(size_QP * sizeof(uint64_t));
    memcpy(c->q, moduli, size_QP * sizeof(uint64_t));
    c->tw = (uint64_t *)malloc(size_QP * n * sizeof(uint64_t));
    c->tw_shoup = (uint64_t *)malloc(size_QP * n * sizeof(uint64_t));
    c->itw = (uint64_t *)malloc(size_QP * n * sizeof(uint64_t));
    c->itw_shoup = (uint64_t *)malloc(size_QP * n * sizeof(uint64_t));
    c->n_inv = (uint64_t *)malloc(size_QP * sizeof(uint64_t));
    c->n_inv_shoup = (uint64_t *)malloc(size_QP * sizeof(uint64_t));
    int bad = 0;
#pragma omp parallel for schedule(dynamic) reduction(| : bad)
    for (size_t i = 0; i < size_QP; i++) {
        bad |= orc_ntt_tables(log_n, moduli[i], c->tw + i * n, c->tw_shoup + i * n, c->itw + i * n,
                              c->itw_shoup + i * n, c->n_inv + i, c->n_inv_shoup + i) != 0;
    }
    if (bad) { orc_ctx_free(c); return NULL; }
    return c;
}

void orc_ctx_free(orc_ctx *c) {
    if (!c) return;
    free(c->q); free(c->tw); free(c->tw_shoup); free(c->itw); free(c->itw_shoup);
    free(c->n_inv); free(c->n_inv_shoup);
    free(c);
}

size_t orc_ctx_n(const orc_ctx *c) { return c->n; }
const uint64_t *orc_ctx_moduli(const orc_ctx *c) { return c->q; }
const uint64_t *orc_ctx_table(const orc_ctx *c, int which) {
    switch (which) {
        case 0: return c->tw;
        case 1: return c->tw_shoup;
        case 2: return c->itw;
        case 3: return c->itw_shoup;
        case 4: return c->n_inv;
        case 5: return c->n_inv_shoup;
        default: return NULL;
    }
}

/* ------------------------------------------------------------------------------------------
 * single-limb transforms — exactly the stage order / twiddle indexing of
 * src/ntt/ntt_1d.cu:41-74 (forward, CT) and :220-254 (inverse, GS) with the Harvey lazy
 * butterflies of include/butterfly.cuh:10-37.  Results are canonical in [0,q).
 * ---------------------------------------------------------------------------------------- */

static inline uint64_t shoup_lazy(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
    uint64_t hi = (uint64_t)(((u128)x * ws) >> 64);
    return x * w - hi * q; /* [0, 2q) */
}

void orc_ntt_fwd(const orc_ctx *c, uint64_t *x, size_t table_idx) {
    const size_t n = c->n;
    const uint64_t q = c->q[table_idx], q2 = 2 * q;
    const uint64_t *tw = c->tw + table_idx * n, *tws = c->tw_shoup + table_idx * n;
    for (size_t m = 1, t = n >> 1; m < n; m <<= 1, t >>= 1) {
        for (size_t k = 0; k < m; k++) {
            const uint64_t w = tw[m + k], ws = tws[m + k];
            uint64_t *a = x + 2 * k * t, *b = a + t;
            for (size_t j = 0; j < t; j++) {
                uint64_t u = a[j];
                if (u >= q2) u -= q2;
                uint64_t v = shoup_lazy(b[j], w, ws, q);
                a[j] = u + v;
                b[j] = u + q2 - v;
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        uint64_t v = x[i];
        if (v >= q2) v -= q2;
        if (v >= q) v -= q;
        x[i] = v;
    }
}

void orc_ntt_inv(const orc_ctx *c, uint64_t *x, size_t table_idx) {
    const size_t n = c->n;
    const uint64_t q = c->q[table_idx], q2 = 2 * q;
    const uint64_t *tw = c->itw + table_idx * n, *tws = c->itw_shoup + table_idx * n;
    for (size_t m = n >> 1, t = 1; m >= 1; m >>= 1, t <<= 1) {
        for (size_t k = 0; k < m; k++) {
            const uint64_t w = tw[m + k], ws = tws[m + k];
            uint64_t *a = x + 2 * k * t, *b = a + t;
            for (size_t j = 0; j < t; j++) {
                uint64_t u = a[j], v = b[j];
                uint64_t s = u + v;
                if (s >= q2) s -= q2;
                a[j] = s;
                b[j] = shoup_lazy(u + q2 - v, w, ws, q);
            }
        }
    }
    /* lower half times N^-1; the upper half got it through itw[1] (src/util/ntt.cu:53-55,
     * src/ntt/intt_2d.cu:106-207) */
    const uint64_t ni = c->n_inv[table_idx], nis = c->n_inv_shoup[table_idx];
    for (size_t i = 0; i < n; i++) {
        uint64_t v = x[i];
        if (i < n / 2) v = shoup_lazy(v, ni, nis, q);
        if (v >= q) v -= q;
        x[i] = v;
    }
}

/* Definition check helper: X[i] = sum_j x[j] * psi^((2*brev(i)+1)*j)  (SURVEY.md A.1).
 * O(N^2); tests call it for small N only. */
void orc_ntt_fwd_naive(const orc_ctx *c, const uint64_t *x, uint64_t *out, size_t table_idx) {
    const size_t n = c->n;
    const uint64_t q = c->q[table_idx];
    /* psi = tw[brev(1)] = tw[n/2] */
    const uint64_t psi = c->tw[table_idx * n + (n >> 1)];
    for (size_t i = 0; i < n; i++) {
        uint64_t e = 2 * (uint64_t)brev((uint32_t)i, c->log_n) + 1;
        uint64_t w = orc_powmod(psi, e, q);
        uint64_t acc = 0, p = 1;
        for (size_t j = 0; j < n; j++) {
            acc = addmod(acc, mulmod(x[j] % q, p, q), q);
            p = mulmod(p, w, q);
        }
        out[i] = acc;
    }
}

/* include_special_mod table-index remap: the last size_P limbs of [start, start+count) use
 * key-level indices size_QP - size_P + k  (src/ntt/fntt_2d.cu:433-436) */
static inline size_t special_idx(size_t twr, size_t start, size_t count, size_t size_QP, size_t size_P) {
    return twr >= start + count - size_P ? size_QP - (start + count - twr) : twr;
}

/* src/ntt/fntt_2d.cu:620-653 */
void orc_nwt_forward_inplace(const orc_ctx *c, uint64_t *inout, size_t count, size_t start) {
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) orc_ntt_fwd(c, inout + (start + i) * c->n, start + i);
}

/* src/ntt/intt_2d.cu:724-757 */
void orc_nwt_backward_inplace(const orc_ctx *c, uint64_t *inout, size_t count, size_t start) {
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) orc_ntt_inv(c, inout + (start + i) * c->n, start + i);
}

/* src/ntt/ntt_modup.cu:320-354 (out-of-place: limb start+i of in -> limb start+i of out) */
void orc_nwt_backward(const orc_ctx *c, uint64_t *out, const uint64_t *in, size_t count, size_t start) {
    memcpy(out + start * c->n, in + start * c->n, count * c->n * sizeof(uint64_t));
    orc_nwt_backward_inplace(c, out, count, start);
}

/* src/ntt/ntt_modup.cu:356-393: INTT then x scale[limb] */
void orc_nwt_backward_scale(const orc_ctx *c, uint64_t *out, const uint64_t *in, size_t count, size_t start,
                            const uint64_t *scale) {
    orc_nwt_backward(c, out, in, count, start);
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) {
        const uint64_t q = c->q[start + i];
        uint64_t *p = out + (start + i) * c->n;
        for (size_t k = 0; k < c->n; k++) p[k] = mulmod(p[k], scale[start + i], q);
    }
}

/* src/ntt/fntt_2d.cu:695-736 and src/ntt/ntt_modup.cu:610-657 (exclude range; pass
 * excl_start == excl_end for none) */
void orc_nwt_forward_inplace_special(const orc_ctx *c, uint64_t *inout, size_t count, size_t start,
                                     size_t size_QP, size_t size_P, size_t excl_start, size_t excl_end) {
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) {
        size_t twr = start + i;
        if (twr >= excl_start && twr < excl_end) continue;
        orc_ntt_fwd(c, inout + twr * c->n, special_idx(twr, start, count, size_QP, size_P));
    }
}

/* src/ntt/intt_2d.cu:796-834 */
void orc_nwt_backward_inplace_special(const orc_ctx *c, uint64_t *inout, size_t count, size_t start,
                                      size_t size_QP, size_t size_P) {
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) {
        size_t twr = start + i;
        orc_ntt_inv(c, inout + twr * c->n, special_idx(twr, start, count, size_QP, size_P));
    }
}

/* src/ntt/ntt_moddown.cu:106-261: delta <- NTT(delta); ct = (cx - delta) * pinv mod q */
void orc_nwt_forward_fuse_moddown(const orc_ctx *c, uint64_t *ct, const uint64_t *cx, const uint64_t *pinv,
                                  uint64_t *delta, size_t count, size_t start) {
    orc_nwt_forward_inplace(c, delta, count, start);
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < count; i++) {
        const size_t l = start + i;
        const uint64_t q = c->q[l];
        for (size_t k = 0; k < c->n; k++) {
            size_t id = l * c->n + k;
            ct[id] = mulmod(submod(cx[id], delta[id], q), pinv[l], q);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * level constants (src/rns.cu:60-192, src/util/rns.cu:284-341,442-501)
 * ---------------------------------------------------------------------------------------- */

static size_t level_beta(const orc_ctx *c, size_t size_Ql) { return (size_Ql + c->size_P - 1) / c->size_P; }

/* modulus of buffer limb j of a Q_l ∪ P buffer */
static inline uint64_t qlp_mod(const orc_ctx *c, size_t size_Ql, size_t j) {
    size_t size_Q = c->size_QP - c->size_P;
    return j < size_Ql ? c->q[j] : c->q[size_Q + (j - size_Ql)];
}

/* prod_{k in [s,e), k != skip} base[k] mod m */
static uint64_t prod_except_mod(const uint64_t *base, size_t s, size_t e, size_t skip, uint64_t m) {
    uint64_t r = 1 % m;
    for (size_t k = s; k < e; k++)
        if (k != skip) r = mulmod(r, base[k] % m, m);
    return r;
}

void orc_level_constants(const orc_ctx *c, size_t size_Ql, uint64_t *inv_q_last_mod_q, uint64_t *bigP_mod_q,
                         uint64_t *bigPInv_mod_q, uint64_t *partQlHatInv_mod_Ql_concat) {
    const size_t alpha = c->size_P, size_Q = c->size_QP - c->size_P;
    const uint64_t *P = c->q + size_Q;
    for (size_t i = 0; i + 1 < size_Ql; i++) {
        if (inv_q_last_mod_q) inv_q_last_mod_q[i] = orc_invmod(c->q[size_Ql - 1] % c->q[i], c->q[i]);
    }
    for (size_t i = 0; i < size_Ql; i++) {
        uint64_t pm = prod_except_mod(P, 0, alpha, (size_t)-1, c->q[i]);
        if (bigP_mod_q) bigP_mod_q[i] = pm;
        if (bigPInv_mod_q) bigPInv_mod_q[i] = orc_invmod(pm, c->q[i]);
    }
    if (partQlHatInv_mod_Ql_concat) {
        size_t beta = level_beta(c, size_Ql);
        for (size_t d = 0; d < beta; d++) {
            size_t s = d * alpha, e = s + alpha < size_Ql ? s + alpha : size_Ql;
            for (size_t i = s; i < e; i++) {
                uint64_t hat = prod_except_mod(c->q, s, e, i, c->q[i]);
                partQlHatInv_mod_Ql_concat[i] = orc_invmod(hat, c->q[i]);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * hybrid key switch (src/rns_bconv.cu:530-628 modup, src/eval_key_switch.cu:12-88 inner
 * product, src/rns_bconv.cu:776-828 moddown_from_NTT, src/eval_key_switch.cu:91-178)
 * ---------------------------------------------------------------------------------------- */

/* dst[beta][size_QlP][n] <- modup(cks[size_Ql][n] in NTT form) */
void orc_modup(const orc_ctx *c, size_t size_Ql, uint64_t *dst, const uint64_t *cks) {
    const size_t n = c->n, alpha = c->size_P, size_QlP = size_Ql + alpha;
    const size_t beta = level_beta(c, size_Ql);
    uint64_t *t_cks = (uint64_t *)malloc(size_Ql * n * sizeof(uint64_t));
    uint64_t *hatinv = (uint64_t *)malloc(size_Ql * sizeof(uint64_t));
    orc_level_constants(c, size_Ql, NULL, NULL, NULL, hatinv);
    if (alpha == 1) {
        /* rns_bconv.cu:547-553: plain INTT, no QHatInv factor */
        orc_nwt_backward(c, t_cks, cks, size_Ql, 0);
    } else {
        /* rns_bconv.cu:554-560: INTT fused with x QHatInv (bconv phase 1) */
        orc_nwt_backward_scale(c, t_cks, cks, size_Ql, 0, hatinv);
    }
    for (size_t d = 0; d < beta; d++) {
        const size_t s = d * alpha, e = s + alpha < size_Ql ? s + alpha : size_Ql;
        uint64_t *part = dst + d * size_QlP * n;
#pragma omp parallel for schedule(dynamic)
        for (size_t j = 0; j < size_QlP; j++) {
            uint64_t *o = part + j * n;
            if (j >= s && j < e) {
                /* own limbs stay in NTT form: rns_bconv.cu:522-528 / :449-451 */
                memcpy(o, cks + j * n, n * sizeof(uint64_t));
                continue;
            }
            const uint64_t pj = qlp_mod(c, size_Ql, j);
            if (alpha == 1) {
                /* rns_bconv.cu:432-453: coeff mod p_j */
                const uint64_t *src = t_cks + s * n;
                for (size_t k = 0; k < n; k++) o[k] = src[k] % pj;
            } else {
                /* rns_bconv.cu:455-485 + rns_bconv.cuh:127-143: sum_i y_i * (QHat_i mod p_j) mod p_j,
                 * matrix from src/util/rns.cu:452-461 */
                uint64_t m[64];
                for (size_t i = s; i < e; i++) m[i - s] = prod_except_mod(c->q, s, e, i, pj);
                for (size_t k = 0; k < n; k++) {
                    u128 acc = 0;
                    for (size_t i = s; i < e; i++) acc += (u128)t_cks[i * n + k] * m[i - s];
                    o[k] = (uint64_t)(acc % pj);
                }
            }
        }
        /* rns_bconv.cu:616-620: NTT everything except the digit's own limbs */
        orc_nwt_forward_inplace_special(c, part, size_QlP, 0, c->size_QP, c->size_P, s, e);
    }
    free(t_cks);
    free(hatinv);
}

/* cx[2][size_QlP][n] = sum_d t_mod_up[d] (.) evk[d][{0,1}]; evk[d] is [2][size_QP][n] at key level */
void orc_key_switch_inner_prod(const orc_ctx *c, size_t size_Ql, uint64_t *cx, const uint64_t *t_mod_up,
                               const uint64_t *const *evk) {
    const size_t n = c->n, alpha = c->size_P, size_QlP = size_Ql + alpha;
    const size_t size_QP = c->size_QP, size_Q = size_QP - alpha;
    const size_t beta = level_beta(c, size_Ql);
#pragma omp parallel for schedule(dynamic)
    for (size_t nid = 0; nid < size_QlP; nid++) {
        const size_t twr = nid >= size_Ql ? size_Q + (nid - size_Ql) : nid; /* eval_key_switch.cu:18 */
        const uint64_t q = c->q[twr];
        for (size_t k = 0; k < n; k++) {
            u128 a0 = 0, a1 = 0;
            for (size_t d = 0; d < beta; d++) {
                uint64_t t = t_mod_up[d * size_QlP * n + nid * n + k];
                a0 += (u128)t * evk[d][twr * n + k];
                a1 += (u128)t * evk[d][size_QP * n + twr * n + k];
            }
            cx[nid * n + k] = (uint64_t)(a0 % q);
            cx[size_QlP * n + nid * n + k] = (uint64_t)(a1 % q);
        }
    }
}

/* ct_i[size_Ql][n] <- moddown(cx_i[size_QlP][n]); cx_i's P limbs are left in coefficient form
 * (the reference destroys them the same way, rns_bconv.cu:788) */
void orc_moddown_from_ntt(const orc_ctx *c, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i) {
    const size_t n = c->n, alpha = c->size_P, size_Q = c->size_QP - alpha;
    const uint64_t *P = c->q + size_Q;
    uint64_t *delta = (uint64_t *)malloc(size_Ql * n * sizeof(uint64_t));
    uint64_t *pinv = (uint64_t *)malloc(size_Ql * sizeof(uint64_t));
    orc_level_constants(c, size_Ql, NULL, NULL, pinv, NULL);
    orc_nwt_backward_inplace_special(c, cx_i, alpha, size_Ql, c->size_QP, alpha);
    const uint64_t *cp = cx_i + size_Ql * n;
    uint64_t phatinv[64];
    for (size_t k = 0; k < alpha; k++) phatinv[k] = orc_invmod(prod_except_mod(P, 0, alpha, k, P[k]), P[k]);
#pragma omp parallel for schedule(dynamic)
    for (size_t j = 0; j < size_Ql; j++) {
        const uint64_t qj = c->q[j];
        uint64_t *o = delta + j * n;
        if (alpha == 1) {
            /* rns_bconv.cu:691-707 */
            for (size_t k = 0; k < n; k++) o[k] = cp[k] % qj;
        } else {
            /* bConv_BEHZ rns_bconv.cu:212-229: (p_k * PHat_k^-1 mod p_k) * (PHat_k mod q_j) */
            uint64_t m[64];
            for (size_t i = 0; i < alpha; i++) m[i] = prod_except_mod(P, 0, alpha, i, qj);
            for (size_t k = 0; k < n; k++) {
                u128 acc = 0;
                for (size_t i = 0; i < alpha; i++) acc += (u128)mulmod(cp[i * n + k], phatinv[i], P[i]) * m[i];
                o[k] = (uint64_t)(acc % qj);
            }
        }
    }
    orc_nwt_forward_fuse_moddown(c, ct_i, cx_i, pinv, delta, size_Ql, 0);
    free(delta);
    free(pinv);
}

/* src/rns_bconv.cu:763-769 */
void orc_add_to_ct(const orc_ctx *c, size_t size_Ql, uint64_t *ct, const uint64_t *cx) {
    for (size_t j = 0; j < size_Ql; j++)
        for (size_t k = 0; k < c->n; k++) ct[j * c->n + k] = addmod(ct[j * c->n + k], cx[j * c->n + k], c->q[j]);
}

/* src/eval_key_switch.cu:91-178 (CKKS branch): ct[2][size_Ql][n] += keyswitch(c2[size_Ql][n]) */
void orc_keyswitch_inplace(const orc_ctx *c, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                           const uint64_t *const *evk) {
    const size_t n = c->n, size_QlP = size_Ql + c->size_P, beta = level_beta(c, size_Ql);
    uint64_t *t_mod_up = (uint64_t *)malloc(beta * size_QlP * n * sizeof(uint64_t));
    uint64_t *cx = (uint64_t *)malloc(2 * size_QlP * n * sizeof(uint64_t));
    orc_modup(c, size_Ql, t_mod_up, c2);
    orc_key_switch_inner_prod(c, size_Ql, cx, t_mod_up, evk);
    for (int i = 0; i < 2; i++) {
        uint64_t *cx_i = cx + (size_t)i * size_QlP * n;
        orc_moddown_from_ntt(c, size_Ql, cx_i, cx_i);
    }
    for (int i = 0; i < 2; i++) orc_add_to_ct(c, size_Ql, ct + (size_t)i * size_Ql * n, cx + (size_t)i * size_QlP * n);
    free(t_mod_up);
    free(cx);
}

/* ------------------------------------------------------------------------------------------
 * rescale (src/rns.cu:1125-1181 divide_and_round_q_last_ntt; floor, not round)
 * src[cipher_size][size_Ql][n] (destroyed: last limb is INTT'ed) -> dst[cipher_size][size_Ql-1][n]
 * ---------------------------------------------------------------------------------------- */
void orc_divide_and_round_q_last_ntt(const orc_ctx *c, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                     uint64_t *dst) {
    const size_t n = c->n, next = size_Ql - 1;
    uint64_t *inv = (uint64_t *)malloc(size_Ql * sizeof(uint64_t));
    orc_level_constants(c, size_Ql, inv, NULL, NULL, NULL);
    for (size_t p = 0; p < cipher_size; p++) {
        uint64_t *ci = src + p * size_Ql * n, *co = dst + p * next * n;
        orc_nwt_backward_inplace(c, ci, 1, size_Ql - 1);
        const uint64_t *last = ci + next * n;
        for (size_t j = 0; j < next; j++)
            for (size_t k = 0; k < n; k++) co[j * n + k] = last[k] % c->q[j];
        orc_nwt_forward_inplace(c, co, next, 0);
        for (size_t j = 0; j < next; j++)
            for (size_t k = 0; k < n; k++)
                co[j * n + k] = mulmod(submod(ci[j * n + k], co[j * n + k], c->q[j]), inv[j], c->q[j]);
    }
    free(inv);
}

/* ------------------------------------------------------------------------------------------
 * Galois (include/galois.cuh:49-64 table, src/galois.cu:8-18 gather, :42-75 step -> element)
 * ---------------------------------------------------------------------------------------- */
void orc_galois_table(int log_n, uint32_t galois_elt, uint32_t *table) {
    const uint32_t n = (uint32_t)1 << log_n;
    for (uint32_t i = n; i < 2 * n; i++) {
        uint32_t reversed = brev(i, log_n + 1);
        uint64_t index_raw = ((uint64_t)galois_elt * reversed) >> 1;
        index_raw &= (uint64_t)(n - 1);
        table[i - n] = brev((uint32_t)index_raw, log_n);
    }
}

void orc_apply_galois_ntt(int log_n, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                          uint64_t *result) {
    const size_t n = (size_t)1 << log_n;
    uint32_t *table = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_galois_table(log_n, galois_elt, table);
    for (size_t l = 0; l < coeff_mod_size; l++)
        for (size_t i = 0; i < n; i++) result[l * n + i] = operand[l * n + table[i]];
    free(table);
}

/* src/galois.cu:42-75, generator 5 (include/galois.cuh generator_) */
uint32_t orc_galois_elt_from_step(int log_n, int step) {
    const uint32_t n = (uint32_t)1 << log_n, m = 2 * n;
    if (step == 0) return m - 1;
    int neg = step < 0;
    uint32_t pos = (uint32_t)(neg ? -step : step);
    if (pos >= (n >> 1)) return 0; /* invalid: step count too large */
    uint32_t s = neg ? (n >> 1) - pos : pos;
    uint64_t elt = 1;
    while (s--) elt = (elt * 5) & (m - 1);
    return (uint32_t)elt;
}

/* coefficient-domain automorphism X -> X^g (src/galois.cu:20-40), used by tests to pin the
 * NTT-domain permutation to its mathematical meaning */
void orc_apply_galois_coeff(const orc_ctx *c, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                            uint64_t *result) {
    const size_t n = c->n;
    for (size_t l = 0; l < coeff_mod_size; l++) {
        uint64_t idx = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t v = operand[l * n + i];
            if (idx >= n) v = v ? c->q[l] - v : 0;
            result[l * n + (idx & (n - 1))] = v;
            idx = (idx + galois_elt) & (2 * n - 1);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * tensor products (src/polymath.cu:468-537)
 * op1, op2: [2][limbs][n]; result [3][limbs][n]
 * ---------------------------------------------------------------------------------------- */
void orc_tensor_prod_2x2(const orc_ctx *c, const uint64_t *op1, const uint64_t *op2, size_t limbs, uint64_t *res) {
    const size_t n = c->n, rc = limbs * n;
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[l];
        for (size_t k = 0; k < n; k++) {
            size_t id = l * n + k;
            uint64_t a0 = op1[id], a1 = op1[id + rc], b0 = op2[id], b1 = op2[id + rc];
            uint64_t d0 = mulmod(a0, b0, q), d2 = mulmod(a1, b1, q);
            uint64_t d1 = addmod(mulmod(a0, b1, q), mulmod(a1, b0, q), q);
            res[id] = d0;
            res[id + rc] = d1;
            res[id + 2 * rc] = d2;
        }
    }
}

void orc_tensor_square_2x2(const orc_ctx *c, const uint64_t *op, size_t limbs, uint64_t *res) {
    orc_tensor_prod_2x2(c, op, op, limbs, res);
}

/* general product (src/polymath.cu:551-599): res[j] = sum_{a+b=j} op1[a] * op2[b]; res may alias op1 (every input of a
 * coefficient is read before its outputs are written, as in the reference) */
void orc_tensor_prod_mxn(const orc_ctx *c, const uint64_t *op1, size_t m, const uint64_t *op2, size_t nn, size_t limbs,
                         uint64_t *res) {
    const size_t n = c->n, rc = limbs * n;
    uint64_t c1[16], c2[16];
    if (m > 16 || nn > 16) return;
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[l];
        for (size_t k = 0; k < n; k++) {
            const size_t id = l * n + k;
            for (size_t i = 0; i < m; i++) c1[i] = op1[id + rc * i];
            for (size_t i = 0; i < nn; i++) c2[i] = op2[id + rc * i];
            for (size_t j = 0; j + 1 < m + nn; j++) {
                unsigned __int128 acc = 0;
                for (size_t a = 0; a < m; a++)
                    if (j >= a && j - a < nn) acc += (unsigned __int128)c1[a] * c2[j - a];
                res[id + rc * j] = (uint64_t)(acc % q);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * helpers for tests that need *valid* keys (functional pin of the key switch):
 * evk digit d = (b_d, a_d), b_d = -a_d*s + e + [P mod q_j * s'] on the digit's own limbs
 * (src/secretkey.cu:265-310, src/polymath.cu:323-343).  All polys in NTT form at key level.
 * ---------------------------------------------------------------------------------------- */
void orc_pointwise_mul(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t limbs,
                       const size_t *table_idx) {
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[table_idx ? table_idx[l] : l];
        for (size_t k = 0; k < c->n; k++) out[l * c->n + k] = mulmod(a[l * c->n + k], b[l * c->n + k], q);
    }
}

/* digest for golden fixtures: FNV-1a 64 over the little-endian bytes */
uint64_t orc_fnv1a64(const uint64_t *data, size_t words) {
    uint64_t h = 0xcbf29ce484222325ULL;
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < words * 8; i++) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

/* SplitMix64 -> uniform residues in [0,q) by rejection (SURVEY.md §8(d) synthetic inputs) */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void orc_fill_uniform(uint64_t *out, size_t count, uint64_t q, uint64_t seed) {
    uint64_t s = seed;
    int bits = 64 - __builtin_clzll(q);
    uint64_t mask = bits >= 64 ? ~0ULL : (((uint64_t)1 << bits) - 1);
    for (size_t i = 0; i < count; i++) {
        uint64_t v;
        do { v = splitmix64(&s) & mask; } while (v >= q);
        out[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * Key generation, symmetric encryption, decryption.
 *
 * Structure follows src/secretkey.cu (gen_secretkey :368-440, encrypt_zero_symmetric :207-263,
 * generate_one_kswitch_key :265-310 + src/polymath.cu:323-343, gen_relinkey :452-477,
 * create_galois_keys :479-521, ckks_decrypt :584-624) and the samplers of src/prng.cu
 * (ternary :142-163, centred binomial :219-242, uniform by rejection :174-207).
 * Randomness: like the reference (Salsa20 under a std::random_device key, src/prng.cu:17,157,192,231) the library
 * samples with a keyed stream cipher — draw ctr of stream `seed` is the first 64 bits of the ChaCha20 block (RFC 8439)
 * under a 256-bit sampler key with counter / nonce words (ctr_lo, ctr_hi, seed_lo, seed_hi); `seed` is a public nonce.
 * The library draws its key from OS entropy; the oracle, being the checker, keeps a FIXED key (all zero unless
 * orc_set_sampler_key changes it) and the tests install the same key in the library to compare key material bit for bit.
 * (This fork's "uniform" sampler stores the constant 1, src/prng.cu:201-203 — the sampler here is the uniform one the
 * comment there describes.)  The generator is random-access, so the GPU computes every coefficient independently.
 * ---------------------------------------------------------------------------------------- */
static uint32_t g_sampler_key[8];

void orc_set_sampler_key(const uint8_t *key32) { memcpy(g_sampler_key, key32, 32); }

#define ROTL32(x, r) (((x) << (r)) | ((x) >> (32 - (r))))
#define QR(a, b, c, d) \
    a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); a += b; d ^= a; d = ROTL32(d, 8); c += d; b ^= c; b = ROTL32(b, 7);

/* RFC 8439 section 2.3: 10 double rounds, then the input state is added */
void orc_chacha20_block(const uint32_t key[8], uint64_t seed, uint64_t ctr, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6],
                       key[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[16];
    memcpy(x, in, sizeof(x));
    for (int r = 0; r < 10; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

static inline uint64_t rnd_at(uint64_t seed, uint64_t ctr) {
    uint32_t o[16];
    orc_chacha20_block(g_sampler_key, seed, ctr, o);
    return ((uint64_t)o[1] << 32) | o[0];
}

uint64_t orc_rnd_at(uint64_t seed, uint64_t ctr) { return rnd_at(seed, ctr); }

/* src/prng.cu:142-163: first byte mod 3, minus 1; the same coefficient in every limb */
void orc_sample_ternary(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed) {
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[table_idx ? table_idx[l] : l];
        for (size_t k = 0; k < c->n; k++) {
            const uint64_t t = (rnd_at(seed, k) & 0xFF) % 3;
            out[l * c->n + k] = t == 0 ? q - 1 : t - 1;
        }
    }
}

/* src/prng.cu:219-242: centred binomial over 21 + 21 bits of the first six bytes */
void orc_sample_error(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed) {
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[table_idx ? table_idx[l] : l];
        for (size_t k = 0; k < c->n; k++) {
            const uint64_t r = rnd_at(seed, k);
            const int cbd = __builtin_popcountll(r & 0x1FFFFFULL) - __builtin_popcountll((r >> 24) & 0x1FFFFFULL);
            out[l * c->n + k] = cbd < 0 ? q - (uint64_t)(-cbd) : (uint64_t)cbd;
        }
    }
}

/* src/prng.cu:174-207: reject above the largest multiple of q, then reduce; attempt a of element
 * (l, k) uses counter l*n + k + a*limbs*n */
void orc_sample_uniform(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed) {
    const uint64_t total = (uint64_t)limbs * c->n;
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[table_idx ? table_idx[l] : l];
        const uint64_t max_multiple = ~0ULL - (~0ULL % q) - 1;
        for (size_t k = 0; k < c->n; k++) {
            uint64_t ctr = l * c->n + k, r;
            while ((r = rnd_at(seed, ctr)) > max_multiple) ctr += total;
            out[l * c->n + k] = r % q;
        }
    }
}

/* limbs of a level: chain_index 0 = key level (all of Q u P), i >= 1 = the first size_Q-(i-1)
 * limbs of Q (include/context.cuh get_context_data) */
static size_t level_limbs(const orc_ctx *c, size_t chain_index) {
    return chain_index == 0 ? c->size_QP : c->size_QP - c->size_P - (chain_index - 1);
}

/* src/secretkey.cu:368-440: ternary coefficients, NTT form over all key-level limbs */
void orc_gen_secretkey(const orc_ctx *c, uint64_t seed, uint64_t *sk) {
    orc_sample_ternary(c, sk, c->size_QP, NULL, seed);
    for (size_t l = 0; l < c->size_QP; l++) orc_ntt_fwd(c, sk + l * c->n, l);
}

/* src/secretkey.cu:207-263 (is_ntt_form = true): (c0, c1) = (-(a*s + e), a) */
void orc_encrypt_zero_symmetric(const orc_ctx *c, const uint64_t *sk, size_t chain_index, uint64_t seed_a,
                                uint64_t seed_e, uint64_t *ct) {
    const size_t limbs = level_limbs(c, chain_index), n = c->n;
    uint64_t *c0 = ct, *c1 = ct + limbs * n;
    orc_sample_error(c, c0, limbs, NULL, seed_e);
    for (size_t l = 0; l < limbs; l++) orc_ntt_fwd(c, c0 + l * n, l);
    orc_sample_uniform(c, c1, limbs, NULL, seed_a);
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[l];
        for (size_t k = 0; k < n; k++) {
            const uint64_t v = addmod(mulmod(c1[l * n + k], sk[l * n + k], q), c0[l * n + k], q);
            c0[l * n + k] = v ? q - v : 0;
        }
    }
}

/* src/secretkey.cu:548-582: c0 += plain (plain in NTT form at the same level) */
void orc_encrypt_symmetric(const orc_ctx *c, const uint64_t *sk, const uint64_t *plain, size_t chain_index,
                           uint64_t seed_a, uint64_t seed_e, uint64_t *ct) {
    const size_t limbs = level_limbs(c, chain_index), n = c->n;
    orc_encrypt_zero_symmetric(c, sk, chain_index, seed_a, seed_e, ct);
    for (size_t l = 0; l < limbs; l++)
        for (size_t k = 0; k < n; k++) ct[l * n + k] = addmod(ct[l * n + k], plain[l * n + k], c->q[l]);
}

/* PhantomSecretKey::gen_publickey (src/secretkey.cu:437-452): an encryption of zero at the key level,
 * pk[2][size_QP][n]; the two seeds are outputs 0 and 1 of the stream at `seed` */
void orc_gen_publickey(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *pk) {
    orc_encrypt_zero_symmetric(c, sk, 0, rnd_at(seed, 0), rnd_at(seed, 1), pk);
}

/* PhantomPublicKey::encrypt_zero_asymmetric_internal (src/secretkey.cu:15-131): at the key level
 * cx_i = pk_i * u + e_i with u ternary and e_0, e_1 from the error distribution (seeds: outputs 0, 1, 2 of the
 * stream at `seed`; the reference reuses one seed for e_0 and e_1), then each poly is divided by P
 * (moddown) onto the first data level; lower levels drop limbs (CKKS mod_switch_to_next). */
void orc_encrypt_zero_asymmetric(const orc_ctx *c, const uint64_t *pk, size_t chain_index, uint64_t seed, uint64_t *ct) {
    const size_t n = c->n, QP = c->size_QP, Q = c->size_QP - c->size_P, limbs = level_limbs(c, chain_index);
    uint64_t *u = (uint64_t *)malloc(QP * n * sizeof(uint64_t));
    uint64_t *cx = (uint64_t *)malloc(QP * n * sizeof(uint64_t));
    uint64_t *top = (uint64_t *)malloc(Q * n * sizeof(uint64_t));
    orc_sample_ternary(c, u, QP, NULL, rnd_at(seed, 0));
    for (size_t l = 0; l < QP; l++) orc_ntt_fwd(c, u + l * n, l);
    for (int i = 0; i < 2; i++) {
        orc_sample_error(c, cx, QP, NULL, rnd_at(seed, 1 + (uint64_t)i));
        for (size_t l = 0; l < QP; l++) {
            orc_ntt_fwd(c, cx + l * n, l);
            const uint64_t q = c->q[l];
            for (size_t k = 0; k < n; k++)
                cx[l * n + k] = addmod(mulmod(u[l * n + k], pk[((size_t)i * QP + l) * n + k], q), cx[l * n + k], q);
        }
        orc_moddown_from_ntt(c, Q, top, cx);
        memcpy(ct + (size_t)i * limbs * n, top, limbs * n * sizeof(uint64_t));
    }
    free(u);
    free(cx);
    free(top);
}

/* PhantomPublicKey::encrypt_asymmetric (src/secretkey.cu:133-164): c0 += plain */
void orc_encrypt_asymmetric(const orc_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t chain_index, uint64_t seed,
                            uint64_t *ct) {
    const size_t limbs = level_limbs(c, chain_index), n = c->n;
    orc_encrypt_zero_asymmetric(c, pk, chain_index, seed, ct);
    for (size_t l = 0; l < limbs; l++)
        for (size_t k = 0; k < n; k++) ct[l * n + k] = addmod(ct[l * n + k], plain[l * n + k], c->q[l]);
}

/* src/secretkey.cu:584-624: plain = c0 + sum_i c_i * s^i (NTT form) */
void orc_ckks_decrypt(const orc_ctx *c, const uint64_t *sk, const uint64_t *ct, size_t cipher_size, size_t chain_index,
                      uint64_t *plain) {
    const size_t limbs = level_limbs(c, chain_index), n = c->n;
    for (size_t l = 0; l < limbs; l++) {
        const uint64_t q = c->q[l];
        for (size_t k = 0; k < n; k++) {
            uint64_t acc = ct[l * n + k], sp = 1;
            for (size_t i = 1; i < cipher_size; i++) {
                sp = mulmod(sp, sk[l * n + k], q);
                acc = addmod(acc, mulmod(ct[(i * limbs + l) * n + k], sp, q), q);
            }
            plain[l * n + k] = acc;
        }
    }
}

/* src/secretkey.cu:265-310 + src/polymath.cu:323-343: dnum = size_Q/size_P encryptions of zero at
 * key level; digit d's b-part gets (P mod q_j) * new_key_j on limbs j in [d*alpha, (d+1)*alpha).
 * Digit d draws (seed_a, seed_e) = (rnd(seed, 2d), rnd(seed, 2d+1)).  evks[d]: [2][size_QP][n]. */
void orc_generate_one_kswitch_key(const orc_ctx *c, const uint64_t *sk, const uint64_t *new_key, uint64_t seed,
                                  uint64_t *const *evks) {
    const size_t n = c->n, alpha = c->size_P, size_Q = c->size_QP - c->size_P, dnum = size_Q / alpha;
    for (size_t d = 0; d < dnum; d++) {
        orc_encrypt_zero_symmetric(c, sk, 0, rnd_at(seed, 2 * d), rnd_at(seed, 2 * d + 1), evks[d]);
        for (size_t j = d * alpha; j < (d + 1) * alpha; j++) {
            const uint64_t q = c->q[j];
            uint64_t bigP = 1;
            for (size_t k = 0; k < alpha; k++) bigP = mulmod(bigP, c->q[size_Q + k] % q, q);
            for (size_t k = 0; k < n; k++)
                evks[d][j * n + k] = addmod(evks[d][j * n + k], mulmod(new_key[j * n + k], bigP, q), q);
        }
    }
}

/* src/secretkey.cu:452-477: s' = s^2 */
void orc_gen_relinkey(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *const *evks) {
    const size_t n = c->n, size_Q = c->size_QP - c->size_P;
    uint64_t *s2 = (uint64_t *)malloc(size_Q * n * sizeof(uint64_t));
    for (size_t l = 0; l < size_Q; l++)
        for (size_t k = 0; k < n; k++) s2[l * n + k] = mulmod(sk[l * n + k], sk[l * n + k], c->q[l]);
    orc_generate_one_kswitch_key(c, sk, s2, seed, evks);
    free(s2);
}

/* src/secretkey.cu:479-521: s' = s(X^g), the NTT-domain permutation of the secret key */
void orc_gen_galois_key(const orc_ctx *c, const uint64_t *sk, uint32_t galois_elt, uint64_t seed, uint64_t *const *evks) {
    const size_t n = c->n, size_Q = c->size_QP - c->size_P;
    uint64_t *sg = (uint64_t *)malloc(size_Q * n * sizeof(uint64_t));
    orc_apply_galois_ntt(c->log_n, sk, size_Q, galois_elt, sg);
    orc_generate_one_kswitch_key(c, sk, sg, seed, evks);
    free(sg);
}

/* ------------------------------------------------------------------------------------------
 * Hoisted rotations: several Galois elements applied to ONE ciphertext share the modup of c1.
 * No counterpart in the reference (its *_hoisting drivers, src/boot/Bootstrapper.cu:1988-2055,
 * still rotate one at a time with rotate_vector); the algorithm is the standard one: the
 * automorphism commutes with the digit decomposition and base extension up to a multiple of Q
 * per coefficient (the same kind of term the uncorrected fast base conversion already carries),
 * so   rot_g(ct) = (sigma_g(c0), 0) + moddown( <sigma_g(modup(c1)), evk_g> ),
 * sigma_g acting on NTT-form limbs as the index permutation of apply_galois_ntt.  Results
 * decrypt like apply_galois_inplace's but are not bit-identical to it.
 * ct [2][size_Ql][n]; out[r] [2][size_Ql][n]; evks[r]: the key of elts[r].
 * ---------------------------------------------------------------------------------------- */
void orc_rotate_hoisted(const orc_ctx *c, size_t size_Ql, const uint64_t *ct, size_t n_rot, const uint32_t *elts,
                        const uint64_t *const *const *evks, uint64_t *const *out) {
    const size_t n = c->n, size_QlP = size_Ql + c->size_P, beta = level_beta(c, size_Ql);
    uint64_t *t = (uint64_t *)malloc(beta * size_QlP * n * sizeof(uint64_t));
    uint64_t *tg = (uint64_t *)malloc(beta * size_QlP * n * sizeof(uint64_t));
    uint64_t *cx = (uint64_t *)malloc(2 * size_QlP * n * sizeof(uint64_t));
    orc_modup(c, size_Ql, t, ct + size_Ql * n);
    for (size_t r = 0; r < n_rot; r++) {
        orc_apply_galois_ntt(c->log_n, t, beta * size_QlP, elts[r], tg);
        orc_key_switch_inner_prod(c, size_Ql, cx, tg, evks[r]);
        for (int i = 0; i < 2; i++) orc_moddown_from_ntt(c, size_Ql, cx + (size_t)i * size_QlP * n, cx + (size_t)i * size_QlP * n);
        orc_apply_galois_ntt(c->log_n, ct, size_Ql, elts[r], out[r]);
        memset(out[r] + size_Ql * n, 0, size_Ql * n * sizeof(uint64_t));
        for (int i = 0; i < 2; i++) orc_add_to_ct(c, size_Ql, out[r] + (size_t)i * size_Ql * n, cx + (size_t)i * size_QlP * n);
    }
    free(t);
    free(tg);
    free(cx);
}
