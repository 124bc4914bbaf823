/*
 * oracle.h — CPU restatement of Phantom-FHE's RNS polynomial-arithmetic hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.c header).  Plain C ABI so tests bind it with ctypes.
 */
#ifndef PHANTOM_AMD_ORACLE_H
#define PHANTOM_AMD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* number theory / table generation */
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);
uint64_t orc_shoup(uint64_t w, uint64_t q);
void orc_barrett_ratio(uint64_t q, uint64_t ratio[2]);
int orc_is_prime(uint64_t n);
int orc_get_primes(uint64_t ntt_size, int bit_size, size_t count, uint64_t *out);
int orc_coeff_modulus_create(uint64_t n, const int *bit_sizes, size_t count, uint64_t *out);
int orc_min_primitive_root(uint64_t degree, uint64_t q, uint64_t *out);
int orc_ntt_tables(int log_n, uint64_t q, uint64_t *tw, uint64_t *tw_shoup, uint64_t *itw, uint64_t *itw_shoup,
                   uint64_t *n_inv, uint64_t *n_inv_shoup);

/* context = key-level chain Q ∪ P (last size_P moduli are P) + per-limb tables */
orc_ctx *orc_ctx_create(int log_n, const uint64_t *moduli, size_t size_QP, size_t size_P);
void orc_ctx_free(orc_ctx *c);
size_t orc_ctx_n(const orc_ctx *c);
const uint64_t *orc_ctx_moduli(const orc_ctx *c);
/* which: 0 tw, 1 tw_shoup, 2 itw, 3 itw_shoup ([size_QP][n]); 4 n_inv, 5 n_inv_shoup ([size_QP]) */
const uint64_t *orc_ctx_table(const orc_ctx *c, int which);

/* single limb, in place, table index given */
void orc_ntt_fwd(const orc_ctx *c, uint64_t *x, size_t table_idx);
void orc_ntt_inv(const orc_ctx *c, uint64_t *x, size_t table_idx);
void orc_ntt_fwd_naive(const orc_ctx *c, const uint64_t *x, uint64_t *out, size_t table_idx);

/* the live nwt_2d_radix8_* family (include/ntt.cuh:47-94); buffers are [limb][n] and limb
 * indices are absolute (start..start+count) like the reference */
void orc_nwt_forward_inplace(const orc_ctx *c, uint64_t *inout, size_t count, size_t start);
void orc_nwt_backward_inplace(const orc_ctx *c, uint64_t *inout, size_t count, size_t start);
void orc_nwt_backward(const orc_ctx *c, uint64_t *out, const uint64_t *in, size_t count, size_t start);
void orc_nwt_backward_scale(const orc_ctx *c, uint64_t *out, const uint64_t *in, size_t count, size_t start,
                            const uint64_t *scale);
void orc_nwt_forward_inplace_special(const orc_ctx *c, uint64_t *inout, size_t count, size_t start, size_t size_QP,
                                     size_t size_P, size_t excl_start, size_t excl_end);
void orc_nwt_backward_inplace_special(const orc_ctx *c, uint64_t *inout, size_t count, size_t start,
                                      size_t size_QP, size_t size_P);
void orc_nwt_forward_fuse_moddown(const orc_ctx *c, uint64_t *ct, const uint64_t *cx, const uint64_t *pinv,
                                  uint64_t *delta, size_t count, size_t start);

/* DRNSTool constants for the level with size_Ql limbs; any output pointer may be NULL */
void orc_level_constants(const orc_ctx *c, size_t size_Ql, uint64_t *inv_q_last_mod_q, uint64_t *bigP_mod_q,
                         uint64_t *bigPInv_mod_q, uint64_t *partQlHatInv_mod_Ql_concat);

/* hybrid key switch */
void orc_modup(const orc_ctx *c, size_t size_Ql, uint64_t *dst, const uint64_t *cks);
void orc_key_switch_inner_prod(const orc_ctx *c, size_t size_Ql, uint64_t *cx, const uint64_t *t_mod_up,
                               const uint64_t *const *evk);
void orc_moddown_from_ntt(const orc_ctx *c, size_t size_Ql, uint64_t *ct_i, uint64_t *cx_i);
void orc_add_to_ct(const orc_ctx *c, size_t size_Ql, uint64_t *ct, const uint64_t *cx);
void orc_keyswitch_inplace(const orc_ctx *c, size_t size_Ql, uint64_t *ct, const uint64_t *c2,
                           const uint64_t *const *evk);

/* rescale */
void orc_divide_and_round_q_last_ntt(const orc_ctx *c, size_t size_Ql, uint64_t *src, size_t cipher_size,
                                     uint64_t *dst);

/* Galois */
void orc_galois_table(int log_n, uint32_t galois_elt, uint32_t *table);
void orc_apply_galois_ntt(int log_n, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                          uint64_t *result);
uint32_t orc_galois_elt_from_step(int log_n, int step);
void orc_apply_galois_coeff(const orc_ctx *c, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                            uint64_t *result);

/* tensor products */
void orc_tensor_prod_2x2(const orc_ctx *c, const uint64_t *op1, const uint64_t *op2, size_t limbs, uint64_t *res);
void orc_tensor_square_2x2(const orc_ctx *c, const uint64_t *op, size_t limbs, uint64_t *res);
void orc_tensor_prod_mxn(const orc_ctx *c, const uint64_t *op1, size_t m, const uint64_t *op2, size_t nn, size_t limbs,
                         uint64_t *res);

/* misc helpers for tests / fixtures */
void orc_pointwise_mul(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t limbs,
                       const size_t *table_idx);
uint64_t orc_fnv1a64(const uint64_t *data, size_t words);
void orc_fill_uniform(uint64_t *out, size_t count, uint64_t q, uint64_t seed);

/* key generation / symmetric encryption / decryption with the project's seeded generator
 * (oracle.c, "Key generation"; src/secretkey.cu, src/prng.cu) */
uint64_t orc_rnd_at(uint64_t seed, uint64_t ctr);
void orc_set_sampler_key(const uint8_t *key32);
void orc_chacha20_block(const uint32_t key[8], uint64_t seed, uint64_t ctr, uint32_t out[16]);
void orc_sample_ternary(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed);
void orc_sample_error(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed);
void orc_sample_uniform(const orc_ctx *c, uint64_t *out, size_t limbs, const size_t *table_idx, uint64_t seed);
void orc_gen_secretkey(const orc_ctx *c, uint64_t seed, uint64_t *sk);
void orc_encrypt_zero_symmetric(const orc_ctx *c, const uint64_t *sk, size_t chain_index, uint64_t seed_a,
                                uint64_t seed_e, uint64_t *ct);
void orc_gen_publickey(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *pk);
void orc_encrypt_zero_asymmetric(const orc_ctx *c, const uint64_t *pk, size_t chain_index, uint64_t seed, uint64_t *ct);
void orc_encrypt_asymmetric(const orc_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t chain_index, uint64_t seed,
                            uint64_t *ct);
void orc_encrypt_symmetric(const orc_ctx *c, const uint64_t *sk, const uint64_t *plain, size_t chain_index,
                           uint64_t seed_a, uint64_t seed_e, uint64_t *ct);
void orc_ckks_decrypt(const orc_ctx *c, const uint64_t *sk, const uint64_t *ct, size_t cipher_size, size_t chain_index,
                      uint64_t *plain);
void orc_generate_one_kswitch_key(const orc_ctx *c, const uint64_t *sk, const uint64_t *new_key, uint64_t seed,
                                  uint64_t *const *evks);
void orc_gen_relinkey(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *const *evks);
void orc_rotate_hoisted(const orc_ctx *c, size_t size_Ql, const uint64_t *ct, size_t n_rot, const uint32_t *elts,
                        const uint64_t *const *const *evks, uint64_t *const *out);
void orc_gen_galois_key(const orc_ctx *c, const uint64_t *sk, uint32_t galois_elt, uint64_t seed, uint64_t *const *evks);

#ifdef __cplusplus
}
#endif
#endif
