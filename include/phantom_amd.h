/*
 * phantom_amd.h — C ABI of the MI355X-native RNS polynomial-arithmetic hot path.
 *
 * Drop-in boundary for Phantom-FHE's L4 seam: every entry point replaces the
 * reference function cited next to it (paths relative to the reference tree).  Plain pointers
 * and sizes only; no C++ / torch types.  All `uint64_t*` data arguments are DEVICE pointers
 * (limb-major `[poly][limb][N]`, canonical residues in [0, q_i)) unless the name says `host`.
 * All compute entry points are asynchronous on `stream` (a `hipStream_t` passed as void*;
 * NULL = the HIP default stream), return 0 on success or a negative PHA_ERR_* code, and never
 * synchronise the host.  Argument errors that make the reference throw std::invalid_argument /
 * std::logic_error return PHA_ERR_INVALID_ARGUMENT / PHA_ERR_LOGIC here; pha_last_error() has
 * the message.  HIP API failures return PHA_ERR_HIP (the reference calls std::exit there).
 *
 * Not thread-safe per context (like the reference: one stream, one context); scratch buffers
 * are owned by the context and sized for the largest level on first use.
 */
#ifndef PHANTOM_AMD_H
#define PHANTOM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHA_OK 0
#define PHA_ERR_INVALID_ARGUMENT (-1)
#define PHA_ERR_LOGIC (-2)
#define PHA_ERR_HIP (-3)
#define PHA_ERR_NO_DEVICE (-4)

typedef struct pha_context pha_context;
typedef void *pha_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------- */
const char *pha_last_error(void);
const char *pha_version(void);

/* ---- host number theory (no GPU needed) ------------------------------------------------ */
/* CoeffModulus::Create (src/modulus.cu:79-110, src/util/numth.cu:209-235) */
int pha_coeff_modulus_create(size_t poly_modulus_degree, const int *bit_sizes, size_t count, uint64_t *out_moduli);
/* arith::NTT::NTT table generation for one modulus (src/util/ntt.cu:11-56): host buffers of
 * N words each; tw/itw are bit-reversed psi powers with N^-1 folded into itw[1]. */
int pha_host_ntt_tables(int log_n, uint64_t modulus, uint64_t *tw, uint64_t *tw_shoup, uint64_t *itw,
                        uint64_t *itw_shoup, uint64_t *n_inv, uint64_t *n_inv_shoup);
/* Modulus::set_value Barrett ratio floor(2^128/q) as (lo, hi) (src/modulus.cu:28-40) */
int pha_host_barrett_ratio(uint64_t modulus, uint64_t ratio[2]);
/* PhantomGaloisTool permutation table (include/galois.cuh:49-64) and step -> element
 * (src/galois.cu:42-75); host buffers */
int pha_host_galois_table(int log_n, uint32_t galois_elt, uint32_t *table);
int pha_galois_elt_from_step(int log_n, int step, uint32_t *out_elt);
/* DRNSTool level constants (src/rns.cu:60-192) for the data level with size_Ql limbs of the
 * chain `moduli_QP` (last size_P are the special primes); host buffers, any may be NULL:
 * inv_q_last_mod_q[size_Ql-1], bigP_mod_q[size_Ql], bigPInv_mod_q[size_Ql],
 * partQlHatInv_mod_Ql_concat[size_Ql]. */
int pha_host_level_constants(const uint64_t *moduli_QP, size_t size_QP, size_t size_P, size_t size_Ql,
                             uint64_t *inv_q_last_mod_q, uint64_t *bigP_mod_q, uint64_t *bigPInv_mod_q,
                             uint64_t *partQlHatInv_mod_Ql_concat);

/* ---- context: PhantomContext + DNTTTable + per-level DRNSTool (src/context.cu:120-290) --- */
/* moduli_QP = key-level coeff_modulus (Q then P); size_P = special_modulus_size
 * (include/util/encryptionparams.h:375, default 1).  Uses the current HIP device.  log_n = 11 .. 17, the degrees of
 * SAMPLE_SIZE (include/ntt.cuh:16-24); any other: PHA_ERR_INVALID_ARGUMENT ("unsupported polynomial degree"). */
int pha_context_create(int log_n, const uint64_t *moduli_QP, size_t size_QP, size_t size_P, pha_context **out);
void pha_context_destroy(pha_context *ctx);
size_t pha_context_n(const pha_context *ctx);
size_t pha_context_size_QP(const pha_context *ctx);
size_t pha_context_size_P(const pha_context *ctx);
/* copy the key-level moduli / the canonical (reference-layout) tables of one limb to host:
 * which = 0 twiddle, 1 twiddle_shoup, 2 itwiddle, 3 itwiddle_shoup (N words)  (include/gputype.h:51-146) */
int pha_context_get_moduli(const pha_context *ctx, uint64_t *out_host);
int pha_context_get_table(const pha_context *ctx, size_t limb, int which, uint64_t *out_host);

/* ---- NTT family (include/ntt.cuh:47-94).  Buffers are [limb][N]; limb indices absolute. --- */
/* nwt_2d_radix8_forward_inplace (src/ntt/fntt_2d.cu:620-653) */
int pha_nwt_2d_radix8_forward_inplace(pha_context *ctx, uint64_t *inout, size_t coeff_modulus_size,
                                      size_t start_modulus_idx, pha_stream_t stream);
/* nwt_2d_radix8_backward_inplace (src/ntt/intt_2d.cu:724-757) */
int pha_nwt_2d_radix8_backward_inplace(pha_context *ctx, uint64_t *inout, size_t coeff_modulus_size,
                                       size_t start_modulus_idx, pha_stream_t stream);
/* nwt_2d_radix8_backward, out-of-place (src/ntt/ntt_modup.cu:320-354) */
int pha_nwt_2d_radix8_backward(pha_context *ctx, uint64_t *out, const uint64_t *in, size_t coeff_modulus_size,
                               size_t start_modulus_idx, pha_stream_t stream);
/* nwt_2d_radix8_backward_scale: INTT then x scale[limb] (src/ntt/ntt_modup.cu:356-393);
 * scale / scale_shoup are device arrays indexed by absolute limb */
int pha_nwt_2d_radix8_backward_scale(pha_context *ctx, uint64_t *out, const uint64_t *in, size_t coeff_modulus_size,
                                     size_t start_modulus_idx, const uint64_t *scale, const uint64_t *scale_shoup,
                                     pha_stream_t stream);
/* nwt_2d_radix8_forward_inplace_include_special_mod (src/ntt/fntt_2d.cu:695-736) */
int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context *ctx, uint64_t *inout,
                                                          size_t coeff_modulus_size, size_t start_modulus_idx,
                                                          size_t size_QP, size_t size_P, pha_stream_t stream);
/* nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range (src/ntt/ntt_modup.cu:610-657) */
int pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(
    pha_context *ctx, uint64_t *inout, size_t coeff_modulus_size, size_t start_modulus_idx, size_t size_QP,
    size_t size_P, size_t excluded_range_start, size_t excluded_range_end, pha_stream_t stream);
/* nwt_2d_radix8_backward_inplace_include_special_mod (src/ntt/intt_2d.cu:796-834) */
int pha_nwt_2d_radix8_backward_inplace_include_special_mod(pha_context *ctx, uint64_t *inout,
                                                           size_t coeff_modulus_size, size_t start_modulus_idx,
                                                           size_t size_QP, size_t size_P, pha_stream_t stream);
/* nwt_2d_radix8_forward_inplace_fuse_moddown (src/ntt/ntt_moddown.cu:222-261):
 * delta <- NTT(delta) (delta is scratch afterwards); ct = (cx - NTT(delta)) * bigPInv mod q */
int pha_nwt_2d_radix8_forward_inplace_fuse_moddown(pha_context *ctx, uint64_t *ct, const uint64_t *cx,
                                                   const uint64_t *bigPInv_mod_q,
                                                   const uint64_t *bigPInv_mod_q_shoup, uint64_t *delta,
                                                   size_t coeff_modulus_size, size_t start_modulus_idx,
                                                   pha_stream_t stream);
/* Extension (no reference counterpart): the same forward / inverse transform over `polys`
 * buffers `[polys][limb_stride_limbs][N]` in one launch pair, sharing each limb's twiddles
 * across the polys (config 2 of BASELINE.json: 2 polys x 36 limbs). */
int pha_ntt_forward_batched(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs,
                            size_t coeff_modulus_size, size_t start_modulus_idx, pha_stream_t stream);
int pha_ntt_backward_batched(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs,
                             size_t coeff_modulus_size, size_t start_modulus_idx, pha_stream_t stream);

/* Measurement aid (no reference counterpart): launch only ONE of the two kernels of a
 * transform (phase 0 = column kernel, 1 = row kernel; inverse != 0 selects the INTT pair) so that
 * bench.py can time the dominant kernel with HIP events.  The buffer contents afterwards are an
 * unfinished transform. */
int pha_ntt_launch_phase(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs,
                         size_t coeff_modulus_size, size_t start_modulus_idx, int inverse, int phase,
                         pha_stream_t stream);


/* Block placement of the NTT kernels (no reference counterpart; the reference's grid-stride kernels have no placement,
 * src/ntt/fntt_2d.cu:625-652): limb_to_xcd != 0 (default) runs every tile of a limb, in both kernels of a transform, on
 * one XCD; 0 restores the tile-interleaved placement.  A speed setting for A/B measurements; results are identical. */
int pha_ntt_block_placement(pha_context *ctx, int limb_to_xcd);

/* The single-launch transform (no reference counterpart; the reference's transform is the two grid-stride launches of
 * src/ntt/fntt_2d.cu:620-653): enable != 0 runs a forward transform of at least `min_units` (poly, limb) pairs as ONE
 * persistent launch whose column pass hands its result to the row pass inside each XCD's L2 (csrc/ntt_fused.cuh);
 * `flags` are the FUSED_* experiment switches of that file (0 for production use).  Results are identical either way.
 * pha_ntt_fused_status synchronises the device and returns the launch error word (0 = none; set when a bounded wait of
 * the single launch ran out, after which results are invalid) and the number of task queues (XCDs) of the device. */
int pha_ntt_fused_config(pha_context *ctx, int enable, uint32_t flags, uint32_t min_units);
int pha_ntt_fused_status(pha_context *ctx, uint32_t *error_word, uint32_t *queues);

/* ---- DRNSTool (include/rns.cuh:155-185); chain_index as in the reference:
 *      1 = first data level (all of Q), k = Q minus its last k-1 primes (src/context.cu:148-174) */
/* DRNSTool::modup, CKKS branch (src/rns_bconv.cu:530-628): dst[beta][size_QlP][N] */
int pha_modup(pha_context *ctx, size_t chain_index, uint64_t *dst, const uint64_t *cks, pha_stream_t stream);
/* key_switch_inner_prod (src/eval_key_switch.cu:12-88): evks = device array of beta device
 * pointers, each a key-level `[2][size_QP][N]` buffer (PhantomRelinKey::public_keys_ptr,
 * include/secretkey.h:85-113).  reduction_threshold is accepted for signature parity and
 * ignored: the result is the canonical residue either way. */
int pha_key_switch_inner_prod(pha_context *ctx, size_t chain_index, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                              const uint64_t *const *evks, size_t reduction_threshold, pha_stream_t stream);
/* DRNSTool::moddown_from_NTT, CKKS branch (src/rns_bconv.cu:776-828); destroys cx_i's P limbs */
int pha_moddown_from_ntt(pha_context *ctx, size_t chain_index, uint64_t *ct_i, uint64_t *cx_i, pha_stream_t stream);
/* add_to_ct_kernel (src/rns_bconv.cu:763-769): ct[limbs][N] += cx[limbs][N] */
int pha_add_to_ct(pha_context *ctx, uint64_t *ct, const uint64_t *cx, size_t size_Ql, pha_stream_t stream);
/* keyswitch_inplace, CKKS branch (src/eval_key_switch.cu:91-178): ct[2][size_Ql][N] += KS(c2) */
int pha_keyswitch_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *evks, pha_stream_t stream);
/* DRNSTool::divide_and_round_q_last_ntt (src/rns.cu:1157-1181); destroys src's last limb */
int pha_divide_and_round_q_last_ntt(pha_context *ctx, size_t chain_index, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, pha_stream_t stream);

/* ---- Galois (include/galois.cuh:134-135, src/galois.cu:8-18,122-138) ---------------------- */
/* result[l][i] = operand[l][perm_g[i]]; result may equal operand (a context scratch buffer is
 * used then, like the reference's temporary).  The table for galois_elt is built and cached
 * on first use. */
int pha_apply_galois_ntt(pha_context *ctx, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                         uint64_t *result, pha_stream_t stream);

/* ---- tensor products (src/polymath.cu:468-537) ------------------------------------------- */
/* result[3][limbs][N] from operand1[2][limbs][N], operand2[2][limbs][N]; result may alias
 * neither operand */
int pha_tensor_prod_2x2(pha_context *ctx, const uint64_t *operand1, const uint64_t *operand2, size_t coeff_mod_size,
                        uint64_t *result, pha_stream_t stream);
int pha_tensor_square_2x2(pha_context *ctx, const uint64_t *operand, size_t coeff_mod_size, uint64_t *result,
                          pha_stream_t stream);
/* tensor_prod_mxn_rns_poly (src/polymath.cu:551-599), what multiply_internal runs for operands that are not both of
 * size 2 (src/evaluate.cu:270-275): result[op1_size + op2_size - 1][limbs][N], result[j] = sum_{a+b=j} op1[a] * op2[b].
 * result may alias operand1 (the reference multiplies in place).  Sizes 1..4 each. */
int pha_tensor_prod_mxn(pha_context *ctx, const uint64_t *operand1, size_t op1_size, const uint64_t *operand2, size_t op2_size,
                        size_t coeff_mod_size, uint64_t *result, pha_stream_t stream);

/* ---- evaluator composites the host keeps (include/evaluate.cuh:114-294) ------------------ */
/* multiply_internal + relinearize_inplace (src/evaluate.cu:229-280,553-583): ct1[2][l][N] *= ct2,
 * then relinearised with evks; ct1 must have room for 3 polys (the reference resizes) */
int pha_multiply_and_relin_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct1, const uint64_t *ct2,
                                   const uint64_t *const *relin_evks, pha_stream_t stream);
/* apply_galois_inplace (src/evaluate.cu:747-794): ct[2][l][N] rotated by galois_elt */
int pha_apply_galois_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct, uint32_t galois_elt,
                             const uint64_t *const *galois_evks, pha_stream_t stream);
/* rescale_to_next / mod_switch_scale_to_next (src/evaluate.cu:585-616,725-745):
 * dst[2][l-1][N] from src[2][l][N]; src is preserved (the reference copies first) */
int pha_rescale_to_next(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size,
                        uint64_t *dst, pha_stream_t stream);
/* The same rescale for callers that cannot allocate.  No counterpart in the reference (which allocates its
 * temporaries with cudaMallocAsync, src/evaluate.cu:585-616): it is for callers that must not reach hipMalloc /
 * hipFree -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  The scratch that
 * pha_rescale_to_next takes from the context's workspace (the inverted last limb of every poly, cipher_size * N
 * words) lives in a file-scope device array of PHA_NOALLOC_SCRATCH_WORDS words (3 MiB per GPU) instead; a larger
 * cipher_size * N is refused with PHA_ERR_INVALID_ARGUMENT, and a level whose constants are not on the device yet with
 * PHA_ERR_LOGIC (pha_prepare_level uploads them; it allocates, so call it at set-up).  Output is bit-identical to
 * pha_rescale_to_next.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * contexts) race. */
#define PHA_NOALLOC_SCRATCH_WORDS ((size_t)3 << 17)
int pha_rescale_to_next_noalloc(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size,
                                uint64_t *dst, pha_stream_t stream);
/* Uploads the per-level constants of chain_index (DRNSTool, src/rns.cu:11-192) now instead of on first use. */
int pha_prepare_level(pha_context *ctx, size_t chain_index);

/* ---- limb-parallel key switch and rescale ---------------------------------
 * No counterpart in the reference, which is single-GPU: these are the per-rank steps of the
 * reference's keyswitch_inplace (src/eval_key_switch.cu:91-178) and divide_and_round_q_last_ntt
 * (src/rns.cu:1157-1181) when the limbs of Q_l u P are partitioned over the GPUs of a node.
 * A rank owns the Q limbs [q_begin, q_begin+q_count) of Q_l and the special limbs
 * [p_begin, p_begin+p_count) of P.  Buffers keep the single-GPU layouts (c2 [l][N], t_mod_up
 * [beta][l+alpha][N], cx [2][l+alpha][N], ct [2][l][N]); each call reads and writes only the owned
 * limbs, except for the inputs named "gathered".  Between the calls the caller exchanges limbs
 * (RCCL all-gather / broadcast; phantom-fhe_amd/limb_parallel.py):
 *   modup_begin  -> all-gather c2_coeff over Q_l            (base conversion needs every input limb)
 *   modup_finish, key_switch_inner_prod, moddown_begin
 *                -> all-gather the P limbs of cx (both polys)
 *   moddown_finish: ct[owned] += key-switched part
 * Results on the owned limbs are bit-identical to pha_keyswitch_inplace / pha_rescale_to_next. */
/* Exchange buffers: rank r owns rows [r * block, (r + 1) * block) of a buffer of world * block rows of N words (block =
 * the largest window of any rank), so that ONE in-place all-gather fills it whatever each rank owns; the calls below
 * take the buffer's row count (for validation), the first row of this rank's block and, on the reading side, a host
 * table "limb -> row" (NULL: limb j at row j).
 * c2_coeff row (row_begin + j) = INTT(c2_ntt[q_begin + j]) * partQlHatInv for the owned Q limbs (src/rns_bconv.cu:530-560) */
int pha_lp_modup_begin(pha_context *ctx, size_t chain_index, const uint64_t *c2_ntt, uint64_t *c2_coeff, size_t c2_coeff_rows,
                       size_t row_begin, size_t q_begin, size_t q_count, pha_stream_t stream);
/* owned limbs of every digit of t_mod_up from the gathered c2_coeff (src/rns_bconv.cu:560-628); q_rows_host[j] = row of
 * limb j in c2_coeff (size_Ql entries, at most 64 limbs; NULL: row j) */
int pha_lp_modup_finish(pha_context *ctx, size_t chain_index, uint64_t *t_mod_up, const uint64_t *c2_ntt,
                        const uint64_t *c2_coeff_gathered, size_t c2_coeff_rows, const uint32_t *q_rows_host, size_t q_begin,
                        size_t q_count, size_t p_begin, size_t p_count, pha_stream_t stream);
/* owned limbs of cx (src/eval_key_switch.cu:12-88); rlk as in pha_key_switch_inner_prod */
int pha_lp_key_switch_inner_prod(pha_context *ctx, size_t chain_index, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                                 const uint64_t *const *rlk, size_t q_begin, size_t q_count, size_t p_begin,
                                 size_t p_count, pha_stream_t stream);
/* owned P limbs of both polys of cx: INTT and * PHatInv (src/rns_bconv.cu:776-800); owned limb p_begin + j of poly p
 * goes to row (p * p_coeff_poly_stride_limbs + row_begin + j) of p_coeff.  With p_coeff_poly_stride_limbs = block and
 * row_begin = 2 * block * rank both polys of a rank sit in ONE contiguous block: one all-gather for the moddown. */
int pha_lp_moddown_begin(pha_context *ctx, size_t chain_index, const uint64_t *cx, uint64_t *p_coeff, size_t p_coeff_rows,
                         size_t p_coeff_poly_stride_limbs, size_t row_begin, size_t p_begin, size_t p_count,
                         pha_stream_t stream);
/* ct[owned] += (cx - NTT(bconv(gathered p_coeff))) * P^-1 (src/rns_bconv.cu:800-828,
 * src/eval_key_switch.cu:157-177); special limb k of poly p is row (p * p_coeff_poly_stride_limbs + p_rows_host[k]) of
 * p_coeff (host array of alpha entries, at most 64; NULL: row k) */
int pha_lp_moddown_finish(pha_context *ctx, size_t chain_index, uint64_t *ct, uint64_t *cx, const uint64_t *p_coeff_gathered,
                          size_t p_coeff_rows, size_t p_coeff_poly_stride_limbs, const uint32_t *p_rows_host, size_t q_begin,
                          size_t q_count, pha_stream_t stream);
/* tensor_prod_2x2 / tensor_square_2x2 (src/polymath.cu:468-537) on the owned limbs [q_begin, q_begin + q_count) of
 * operands [2][coeff_modulus_size][N]; result [3][coeff_modulus_size][N]; operand1 == operand2 squares */
int pha_lp_tensor_prod_2x2(pha_context *ctx, const uint64_t *operand1, const uint64_t *operand2, size_t coeff_modulus_size,
                           size_t q_begin, size_t q_count, uint64_t *result, pha_stream_t stream);
/* owner of the last limb: last_coeff[p][N] = INTT(src[p][l-1]); broadcast it, then every rank
 * finishes its owned limbs of dst[cipher_size][l-1][N] (src/rns.cu:1157-1181) */
int pha_lp_rescale_begin(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size,
                         uint64_t *last_coeff, pha_stream_t stream);
int pha_lp_rescale_finish(pha_context *ctx, size_t chain_index, const uint64_t *src, const uint64_t *last_coeff,
                          size_t cipher_size, uint64_t *dst, size_t q_begin, size_t q_count, pha_stream_t stream);

/* ---- key generation, symmetric encryption, decryption --------------------
 * PhantomSecretKey's methods (include/secretkey.h:150-290, src/secretkey.cu) over caller-owned
 * buffers.  Distributions are the reference's (src/prng.cu: ternary, centred binomial, uniform by
 * rejection).  Randomness is a keyed stream cipher as in the reference (Salsa20 under a std::random_device key,
 * src/prng.cu): draw number ctr of stream `seed` is the first 64 bits of ChaCha20(key, counter/nonce = (ctr, seed)) under
 * the PROCESS'S 256-bit sampler key, which is drawn from OS entropy on first use.  The `seed` arguments below are
 * therefore public NONCES that name a stream under that key (distinct nonces = independent streams), not secrets:
 * use a fresh one per draw (pha_random_nonce) and never reuse a nonce for two objects.  With the key fixed by
 * pha_set_sampler_key (tests, the oracle) results are reproducible from the nonces.
 * chain_index 0 is the key level (all of Q u P), i >= 1 the first size_Q-(i-1) limbs of Q. */
/* gen_secretkey (src/secretkey.cu:368-440): sk[size_QP][N], ternary, NTT form */
/* Installs the 32-byte sampler key (NULL: 32 fresh bytes of OS entropy).  A fixed key makes every sampled object a
 * function of its nonces: for tests and the oracle comparison only. */
int pha_set_sampler_key(const uint8_t *key32);
/* 64 bits of OS entropy: a nonce for one draw. */
int pha_random_nonce(uint64_t *out);
int pha_gen_secretkey(pha_context *ctx, uint64_t seed, uint64_t *sk, pha_stream_t stream);
/* encrypt_zero_symmetric (src/secretkey.cu:207-263, NTT form): ct[2][l][N] = (-(a*s+e), a) */
int pha_encrypt_zero_symmetric(pha_context *ctx, const uint64_t *sk, size_t chain_index, uint64_t seed_a,
                               uint64_t seed_e, uint64_t *ct, pha_stream_t stream);
/* PhantomSecretKey::gen_publickey (src/secretkey.cu:437-452): pk[2][size_QP][N] = an encryption of zero at the key
 * level; seeded (outputs 0, 1 of the stream at `seed` drive a and e) */
int pha_gen_publickey(pha_context *ctx, const uint64_t *sk, uint64_t seed, uint64_t *pk, pha_stream_t stream);
/* PhantomPublicKey::encrypt_zero_asymmetric_internal / encrypt_asymmetric (src/secretkey.cu:15-164):
 * cx_i = pk_i * u + e_i at the key level (u ternary), divided by P onto the first data level (moddown), limbs
 * dropped for chain_index > 1; ct[2][l][N], plain[l][N] in NTT form.  e_0 and e_1 use distinct seeds here (the
 * reference draws both from one seed). */
int pha_encrypt_zero_asymmetric(pha_context *ctx, const uint64_t *pk, size_t chain_index, uint64_t seed, uint64_t *ct,
                                pha_stream_t stream);
int pha_encrypt_asymmetric(pha_context *ctx, const uint64_t *pk, const uint64_t *plain, size_t chain_index, uint64_t seed,
                           uint64_t *ct, pha_stream_t stream);
/* encrypt_symmetric (src/secretkey.cu:548-582): c0 += plain, plain[l][N] in NTT form */
int pha_encrypt_symmetric(pha_context *ctx, const uint64_t *sk, const uint64_t *plain, size_t chain_index,
                          uint64_t seed_a, uint64_t seed_e, uint64_t *ct, pha_stream_t stream);
/* ckks_decrypt (src/secretkey.cu:584-624): plain[l][N] = sum_i ct_i * s^i, NTT form */
int pha_ckks_decrypt(pha_context *ctx, const uint64_t *sk, const uint64_t *ct, size_t cipher_size, size_t chain_index,
                     uint64_t *plain, pha_stream_t stream);
/* generate_one_kswitch_key (src/secretkey.cu:265-310, src/polymath.cu:323-343): evks_host is a
 * HOST array of dnum = size_Q/size_P device buffers [2][size_QP][N]; new_key[size_Q][N] */
int pha_generate_one_kswitch_key(pha_context *ctx, const uint64_t *sk, const uint64_t *new_key, uint64_t seed,
                                 uint64_t *const *evks_host, pha_stream_t stream);
/* gen_relinkey (src/secretkey.cu:452-477): new_key = s^2 */
int pha_gen_relinkey(pha_context *ctx, const uint64_t *sk, uint64_t seed, uint64_t *const *evks_host,
                     pha_stream_t stream);
/* one entry of create_galois_keys (src/secretkey.cu:479-521): new_key = s(X^galois_elt) */
int pha_gen_galois_key(pha_context *ctx, const uint64_t *sk, uint32_t galois_elt, uint64_t seed,
                       uint64_t *const *evks_host, pha_stream_t stream);

/* ---- element-wise evaluator arithmetic ------------------------------------
 * add_rns_poly / sub_rns_poly / multiply_rns_poly / negate_rns_poly of src/polymath.cu as
 * add_inplace, sub_inplace, negate_inplace, add_plain, sub_plain and multiply_plain use them
 * (src/evaluate.cu:20-105,442-560): a, dst are [polys][coeff_mod_size][N]; b has b_polys == polys
 * polys, or b_polys == 1 (a plaintext applied to every poly of a).  dst may alias a. */
int pha_add_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size,
                     size_t polys, size_t b_polys, pha_stream_t stream);
int pha_sub_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size,
                     size_t polys, size_t b_polys, pha_stream_t stream);
int pha_multiply_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size,
                          size_t polys, size_t b_polys, pha_stream_t stream);
/* dst = c + a * b (multiply_and_add_rns_poly, src/polymath.cu:230-260): the accumulation step of the
 * bootstrapper's linear transforms; c has the shape of a, dst may alias c */
int pha_multiply_add_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                              size_t coeff_mod_size, size_t polys, size_t b_polys, pha_stream_t stream);
int pha_negate_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, size_t coeff_mod_size, size_t polys,
                        pha_stream_t stream);
/* mod_switch_drop_to_next (src/evaluate.cu:618-700): dst[cipher_size][l-1][N] = src without its
 * last limb (CKKS mod_switch_to_next; no scaling) */
int pha_mod_switch_drop_to_next(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size,
                                uint64_t *dst, pha_stream_t stream);

/* ---- CKKS encoder / decoder (PhantomCKKSEncoder, src/ckks.cu:130-283) ------
 * values_host: interleaved (re, im) doubles on the HOST, like the reference's std::vector inputs;
 * both calls synchronise the stream (the reference does, src/ckks.cu:187,282).
 * encode: values_size <= sparse_slots <= N/2 (sparse_slots = 0: next power of two >= values_size)
 * -> plain[l][N], NTT form, level chain_index; decode: the first sparse_slots slots of plain. */
int pha_ckks_encode(pha_context *ctx, const double *values_host, size_t values_size, size_t sparse_slots,
                    size_t chain_index, double scale, uint64_t *plain, pha_stream_t stream);
int pha_ckks_decode(pha_context *ctx, const uint64_t *plain, size_t chain_index, double scale, size_t sparse_slots,
                    double *values_host, pha_stream_t stream);

/* ---- hoisted rotations ---------------------------------------------------------------------
 * No counterpart in the reference (its *_hoisting drivers, src/boot/Bootstrapper.cu:1988-2055,
 * rotate one at a time): n_rot Galois elements applied to ONE ciphertext ct[2][l][N] share the
 * modup of c1 — per rotation only the inner product (with the automorphism applied to t_mod_up on
 * the fly), the moddown and the permutation of c0 remain.  out_host[r][2][l][N] decrypts like
 * pha_apply_galois_inplace(ct, galois_elts_host[r]) but is not bit-identical to it (the base
 * extension commutes with the automorphism only up to multiples of Q; oracle.c "Hoisted
 * rotations").  galois_elts_host, evks_host (device pointer arrays as in pha_keyswitch_inplace,
 * one per rotation) and out_host are HOST arrays; outputs must not alias ct. */
int pha_rotate_hoisted(pha_context *ctx, size_t chain_index, const uint64_t *ct, size_t n_rot,
                       const uint32_t *galois_elts_host, const uint64_t *const *const *evks_host,
                       uint64_t *const *out_host, pha_stream_t stream);

/* multiply + relinearize + rescale_to_next (src/evaluate.cu:229-280,553-616) as ONE composite: out[2][l-1][N] from
 * ct1, ct2 [2][l][N] (ct1 == ct2 squares); times2 != 0 also doubles the result (a Chebyshev product 2 T_a T_b).
 * The tensor's first two polys are lifted over Q_l u P and the key-switched third, and one moddown divides by
 * P * q_{l-1}: one NTT pass over the ciphertext less than the three calls, no last-limb INTT / broadcast.  The result
 * equals rescale(relinearize(multiply)) up to the rounding of the divisions (a few units); the reference has no
 * counterpart.  ct1 and ct2 are preserved. */
int pha_multiply_relin_rescale(pha_context *ctx, size_t chain_index, const uint64_t *ct1, const uint64_t *ct2,
                               const uint64_t *const *relin_evks, uint64_t *out, int times2, pha_stream_t stream);

/* `batch` independent pha_multiply_relin_rescale products at one level, one launch sequence: ct1s / ct2s are host arrays
 * of device pointers ([2][l][N] each; ct1s[b] == ct2s[b] squares), outs is [batch][2][l-1][N] contiguous.  Bit-identical
 * to `batch` single calls; the transforms and base conversions span the batch and the relinearisation key is read once
 * per four products (the leaves and the same-height products of a Paterson-Stockmeyer polynomial evaluation).  The
 * reference has no counterpart. */
int pha_multiply_relin_rescale_batched(pha_context *ctx, size_t chain_index, size_t batch, const uint64_t *const *ct1s,
                                       const uint64_t *const *ct2s, const uint64_t *const *relin_evks, uint64_t *outs, int times2,
                                       pha_stream_t stream);

/* apply_galois_inplace (src/evaluate.cu:747-794) on `batch` independent ciphertexts sharing the Galois element and
 * its key: in / out are [batch][2][l][N] contiguous, out of place (BASELINE configs[3], rotate x 64).  Results are
 * bit-identical to `batch` calls of pha_apply_galois_inplace; the reference has no batched entry point.  Levels whose
 * digits are uniform (alpha divides l) or that hold a single short digit (l < alpha: the 1-3 limb ciphertexts of the
 * CNN's convolutions) run kernels that span the batch; alpha = 1 and several digits with a ragged last one fall back to
 * one ciphertext at a time. */
int pha_apply_galois_batched(pha_context *ctx, size_t chain_index, const uint64_t *in, uint64_t *out, size_t batch,
                             uint32_t galois_elt, const uint64_t *const *galois_evks, pha_stream_t stream);

/* Plaintext over Q_l u P ([size_Ql + size_P][N], NTT form): pha_ckks_encode with the special primes appended, the
 * operand format of pha_linear_transform_double_hoisted.  No counterpart in the reference. */
int pha_ckks_encode_qp(pha_context *ctx, const double *values_host, size_t values_size, size_t sparse_slots,
                       size_t chain_index, double scale, uint64_t *plain, pha_stream_t stream);
/* Double-hoisted baby-step / giant-step linear transform, the inner loop of the bootstrapper's CoeffToSlot /
 * SlotToCoeff (the reference's bsgs_linear_transform_hoisting family, src/boot/Bootstrapper.cu:1988-2127, shares the
 * modup only; this also keeps the rotated ciphertexts over Q_l u P so that there are n_giant + 1 moddowns instead
 * of n_baby + n_giant):  out[2][l][N] = sum_g rot_{G_g}( sum_b plains[g * n_baby + b] * rot_{B_b}(ct) ).
 * Galois element 1 = no rotation (its key entry may be NULL); plains entries are device pointers to
 * pha_ckks_encode_qp plaintexts or NULL for an absent diagonal; element / key / plaintext tables are HOST arrays.
 * The result differs from the step-by-step evaluation by the rounding of the moddowns only. */
int pha_linear_transform_double_hoisted(pha_context *ctx, size_t chain_index, const uint64_t *ct, size_t n_baby,
                                        const uint32_t *baby_elts_host, const uint64_t *const *const *baby_evks_host,
                                        size_t n_giant, const uint32_t *giant_elts_host,
                                        const uint64_t *const *const *giant_evks_host, const uint64_t *const *plains_host,
                                        uint64_t *out, pha_stream_t stream);

/* ---- bootstrapping primitives -----------------------------------------------
 * Bootstrapper::modraise_inplace (src/boot/Bootstrapper.cu:2801-2861): src[2][1][N], a ciphertext at
 * the lowest level (one limb, NTT form) -> dst[2][size_Q][N] at the top level (chain_index 1),
 * every coefficient lifted to its centred representative modulo q0 */
int pha_mod_raise(pha_context *ctx, const uint64_t *src, uint64_t *dst, pha_stream_t stream);
/* gen_secretkey with parms.secret_key_hamming_weight (src/secretkey.cu:401-433): exactly
 * hamming_weight coefficients are +-1; seeded like pha_gen_secretkey */
int pha_gen_secretkey_sparse(pha_context *ctx, uint64_t seed, size_t hamming_weight, uint64_t *sk,
                             pha_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PHANTOM_AMD_H */
