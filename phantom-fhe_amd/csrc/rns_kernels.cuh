// Base conversion, key-switch inner product, Galois gather, tensor product and element-wise
// kernels for gfx950.  Arithmetic restates src/rns_bconv.cu, src/eval_key_switch.cu,
// src/galois.cu, src/polymath.cu and src/rns.cu:1125-1155 of the reference; every stored value
// is the canonical residue, so results are bit-identical.
#pragma once

#include "modarith.cuh"
#include "numth.hpp"

namespace pha {

// ---------------------------------------------------------------------------------------------
// fast base conversion, phase 2 (src/rns_bconv.cu:143-168,455-485, include/rns_bconv.cuh:127-143):
//   dst[jo][k] = sum_i y[i][k] * M[j][i] mod p_j,  jo = j + (j >= skip_start ? skip_len : 0)
// One lane owns one coefficient: its IB inputs stay in registers and are reused for every
// output limb of the block's limb group; the matrix row is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------------------------
// blockIdx.z batches independent conversions that share the shape (the beta digits of modup when
// alpha divides size_Ql, or the two polys of moddown): every pointer advances by its z-stride.
constexpr uint32_t kBconvMaxRows = 64;
struct BconvBatch {
    size_t dst_stride, y_stride, matrix_stride, tidx_stride;
    uint32_t skip_stride;
    const uint64_t *copy_src;  // optional: the digit's own (NTT-form) limbs, copied to the skipped slots
    size_t copy_stride;        // (src/rns_bconv.cu:522-528 modup_copy_partQl_kernel, fused)
    // group > 0: z = b * group + d runs over [batch of ciphertexts][digits]; the per-digit strides above apply to d,
    // the group strides to b (dst stays linear in z)
    uint32_t group;
    size_t y_group_stride, copy_group_stride;
    // in_rows_set != 0: input limb i of digit d is row in_row[d * in_row_stride + i] of y (which then does not advance
    // with the digit: the caller sets y_stride = 0) instead of row i: a gathered buffer whose rank blocks are padded
    // or not in limb order (limb-parallel modup and moddown); register-path kernels only (ibase <= 16)
    uint32_t in_rows_set;
    uint32_t in_row_stride;
    uint16_t in_row[kBconvMaxRows];
    // own_set != 0: only the outputs whose buffer limb jo lies in [own[0], own[1]) or [own[2], own[3]) are computed, and
    // only those of the digit's own limbs are copied (limb-parallel modup: a rank's Q window and its P window)
    uint32_t own_set;
    uint32_t own[4];
};

__device__ __forceinline__ bool bconv_owned(const BconvBatch &bt, uint32_t jo) {
    return !bt.own_set || (jo >= bt.own[0] && jo < bt.own[1]) || (jo >= bt.own[2] && jo < bt.own[3]);
}

__device__ __forceinline__ void bconv_batch_offsets(const BconvBatch &bt, uint32_t z, uint64_t *__restrict__ &dst,
                                                    const uint64_t *__restrict__ &y, const uint64_t *__restrict__ &matrix,
                                                    const uint32_t *__restrict__ &obase_tidx, uint32_t &skip_start,
                                                    const uint64_t *&copy_src, uint32_t &digit) {
    const uint32_t d = bt.group ? z % bt.group : z, b = bt.group ? z / bt.group : 0;
    digit = d;
    dst += z * bt.dst_stride;
    y += d * bt.y_stride + b * bt.y_group_stride;
    matrix += d * bt.matrix_stride;
    obase_tidx += d * bt.tidx_stride;
    skip_start += d * bt.skip_stride;
    copy_src = bt.copy_src ? bt.copy_src + d * bt.copy_stride + b * bt.copy_group_stride : nullptr;
}

// 128-bit dot product by columns: every input and matrix entry is a residue < 2^61, so the high
// words are < 2^29.  c0 collects the low x low products with an explicit carry counter (the
// carry-out of v_mad_u64_u32 feeds a v_addc); the cross terms (< 2^61 each) go to accumulators
// that take at most 8 terms, the high x high products (< 2^58) never overflow.  4 multiplies and
// one add-with-carry per term instead of a full 128-bit add.
template <int IB>
__device__ __forceinline__ u128_t dot_residues(const uint64_t (&in)[IB], const uint64_t *__restrict__ row) {
    static_assert(IB <= 16, "cross-term accumulators hold 8 terms each");
    uint64_t c0 = 0, c2 = 0, c1[4] = {0, 0, 0, 0};
    uint32_t k0 = 0;
#pragma unroll
    for (int i = 0; i < IB; i++) {
        const uint32_t x0 = lo32(in[i]), x1 = hi32(in[i]), m0 = lo32(row[i]), m1 = hi32(row[i]);
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(c0), "+v"(k0)
            : "v"(x0), "s"(m0)
            : "vcc");
        c1[(i >> 3) * 2] = mad32(x0, m1, c1[(i >> 3) * 2]);
        c1[(i >> 3) * 2 + 1] = mad32(x1, m0, c1[(i >> 3) * 2 + 1]);
        c2 = mad32(x1, m1, c2);
    }
    unsigned __int128 acc = static_cast<unsigned __int128>(c0) + (static_cast<unsigned __int128>(c2 + k0) << 64);
#pragma unroll
    for (int t = 0; t < (IB > 8 ? 4 : 2); t++) acc += static_cast<unsigned __int128>(c1[t]) << 32;
    return u128_t{static_cast<uint64_t>(acc), static_cast<uint64_t>(acc >> 64)};
}

template <int IB>
__global__ __launch_bounds__(256) void bconv_matmul_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ y,
                                                           const uint64_t *__restrict__ matrix,
                                                           const DModulus *__restrict__ obase_mod,
                                                           const uint32_t *__restrict__ obase_tidx,
                                                           uint32_t obase_size, uint32_t limbs_per_block, uint32_t n,
                                                           uint32_t skip_start, uint32_t skip_len, BconvBatch bt) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint64_t *src;
    uint32_t digit;
    bconv_batch_offsets(bt, blockIdx.z, dst, y, matrix, obase_tidx, skip_start, src, digit);
    const uint32_t j0 = blockIdx.y * limbs_per_block;
    const uint32_t j1 = min(j0 + limbs_per_block, obase_size);
    if (bt.own_set) {   // a workgroup none of whose outputs (or copies) this rank owns has nothing to read
        bool any = src != nullptr && blockIdx.y == 0 && skip_start < bt.own[1] && skip_start + IB > bt.own[0];
        for (uint32_t j = j0; j < j1 && !any; j++) any = bconv_owned(bt, j + (j >= skip_start ? skip_len : 0));
        if (!any) return;
    }
    uint64_t in[IB];
#pragma unroll
    for (int i = 0; i < IB; i++)
        in[i] = y[static_cast<size_t>(bt.in_rows_set ? bt.in_row[(digit * bt.in_row_stride + i) & (kBconvMaxRows - 1)] : i) * n + k];
    if (src != nullptr && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < IB; i++)
            if (bconv_owned(bt, skip_start + i)) dst[static_cast<size_t>(skip_start + i) * n + k] = src[static_cast<size_t>(i) * n + k];
    }
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t jo = j + (j >= skip_start ? skip_len : 0);
        if (!bconv_owned(bt, jo)) continue;
        const u128_t acc = dot_residues<IB>(in, matrix + static_cast<size_t>(j) * IB);
        const DModulus m = obase_mod[obase_tidx[j]];
        dst[static_cast<size_t>(jo) * n + k] = barrett128(acc, m);
    }
}

// ---------------------------------------------------------------------------------------------
// The same conversion on the matrix cores (numth.hpp: build_bconv_mfma_table states the arithmetic and the table
// layout).  It is a real matrix product -- [obase x 8 ibase] digits times [8 ibase x N] input bytes -- and on the
// vector ALU it is bound by instruction issue (profiles/r2c_bconv.md: 176 instructions per output and wave), so the
// byte products go to v_mfma_i32_16x16x64_i8 and the vector ALU is left with recombining eight 21-bit partial sums
// and one 80 -> 61-bit Barrett step per output.
// A workgroup converts 256 coefficients (64 per wave, as two pairs of 16-coefficient MFMA column groups: lane (g, c)
// loads 16 bytes = coefficients 2c, 2c + 1 of inputs 2g, 2g + 1 per k-step, which ARE its B operands after an xor)
// for the quads of blockIdx.y, whose A operands and constants it stages in LDS once.  Lane (g, c) ends up with all
// eight partial sums of output 4 quad + g at its coefficient: no cross-lane step, 16-byte stores.
// ---------------------------------------------------------------------------------------------
typedef int bm_v4i __attribute__((ext_vector_type(4)));
constexpr uint32_t kBmOperandBytes = 1024;   // one A operand: 64 lanes x 16 bytes

// d0, d1: the partial sums of digits 0..3 and 4..7, registers 0 and 2 of each started at kBmfBiasWord (the MFMA's C
// operand) so that every intermediate is a non-negative machine word
__device__ __forceinline__ uint64_t bm_finish(const bm_v4i &d0, const bm_v4i &d1, uint64_t p, uint64_t mu) {
    // |partial sum| < 2^21: lo, hi in (2^29, 2^31); v = lo + 2^16 hi < 2^47.0001
    const uint32_t lo0 = static_cast<uint32_t>(d0.x) + (static_cast<uint32_t>(d0.y) << 8);
    const uint32_t hi0 = static_cast<uint32_t>(d0.z) + (static_cast<uint32_t>(d0.w) << 8);
    const uint32_t lo1 = static_cast<uint32_t>(d1.x) + (static_cast<uint32_t>(d1.y) << 8);
    const uint32_t hi1 = static_cast<uint32_t>(d1.z) + (static_cast<uint32_t>(d1.w) << 8);
    const uint64_t v0 = mad32(hi0, 65536u, lo0), v1 = mad32(hi1, 65536u, lo1);
    // x = v0 + 2^32 v1 < 2^80; xh = floor(x / 2^32); qhat = floor(xh mu / 2^64) is floor(x / p) or one less (p >= 2^34)
    const uint64_t xh = v1 + hi32(v0);
    const uint64_t xlo = pack64(lo32(v0), lo32(xh));
    const uint64_t qhat = mulhi64(xh, mu);
    return csub(xlo - qhat * p, p);
}

// PAIRS: 32-coefficient pairs per wave (2: a workgroup converts 256 coefficients, 1: 128)
template <int KSTEPS, int PAIRS>
__global__ __launch_bounds__(256) void bconv_mfma_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ y,
                                                         const uint8_t *__restrict__ table, const uint64_t *__restrict__ consts,
                                                         size_t table_stride, size_t consts_stride, uint32_t ibase,
                                                         uint32_t obase, uint32_t quads_per_block, uint32_t quad_groups,
                                                         uint32_t n, uint32_t skip_start, uint32_t skip_len, BconvBatch bt) {
    extern __shared__ __align__(16) uint8_t bm_lds[];
    constexpr uint32_t kQuadBytes = 2 * KSTEPS * kBmOperandBytes, kCoeffs = 128 * PAIRS;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    // 1-D grid over (coefficient block, quad group, z).  Workgroups are dealt to the 8 XCDs round-robin: the quad
    // groups of one (block, z) -- which read the same inputs -- are given ids 8 apart, so they share an L2 and run
    // back to back; gridDim.y == groups when the block count is no multiple of 8 (plain 3-D grid)
    uint32_t bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (gridDim.y == 1 && quad_groups > 1) {
        const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3, unit = (t / quad_groups) * 8 + xcd, per_z = n / kCoeffs;
        by = t % quad_groups;
        z = unit / per_z;
        bx = unit % per_z;
    }
    const uint32_t d = bt.group ? z % bt.group : z, b = bt.group ? z / bt.group : 0;
    dst += z * bt.dst_stride;
    y += d * bt.y_stride + b * bt.y_group_stride;
    table += d * table_stride;
    consts += d * consts_stride;
    skip_start += d * bt.skip_stride;
    const uint64_t *src = bt.copy_src ? bt.copy_src + d * bt.copy_stride + b * bt.copy_group_stride : nullptr;
    const uint32_t quads = (obase + 3) >> 2, q0 = by * quads_per_block, nq = min(quads_per_block, quads - q0);
    const bool copies = src != nullptr && by == 0;
    if (bt.own_set) {   // a workgroup none of whose outputs (or copies) this rank owns has nothing to do
        bool any = copies && skip_start < bt.own[1] && skip_start + ibase > bt.own[0];
        for (uint32_t j = q0 * 4; j < min((q0 + nq) * 4, obase) && !any; j++) any = bconv_owned(bt, j + (j >= skip_start ? skip_len : 0));
        if (!any) return;
    }
    // the inputs: ld[kstep][pair][h] = coefficients k0 + 32 pair + {0, 1} of input 8 kstep + 2 g + h, bytes biased by -128
    const uint32_t k0 = bx * kCoeffs + wave * (32 * PAIRS) + 2 * c;
    bm_v4i ld[KSTEPS][PAIRS][2];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = ks * 8 + 2 * g + h;
            const uint32_t row = bt.in_rows_set ? bt.in_row[(d * bt.in_row_stride + i) & (kBconvMaxRows - 1)] : i;
#pragma unroll
            for (int pr = 0; pr < PAIRS; pr++) {
                bm_v4i v = {0, 0, 0, 0};
                if (i < ibase) v = *reinterpret_cast<const bm_v4i *>(y + static_cast<size_t>(row) * n + k0 + 32 * pr);
                ld[ks][pr][h] = v;
            }
        }
    // A operands and constants of this workgroup's quads: every load issued before the first LDS store
    {
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(table + static_cast<size_t>(q0) * kQuadBytes);
        uint4 *tdst = reinterpret_cast<uint4 *>(bm_lds);
        const uint32_t total = nq * (kQuadBytes / 16);
        for (uint32_t i0 = tid; i0 < total; i0 += 256 * 4) {   // loads clamped, stores predicated: all four in flight at once
            const uint32_t i1 = i0 + 256, i2 = i0 + 512, i3 = i0 + 768, last = total - 1;
            const uint4 v0 = tsrc[i0], v1 = tsrc[min(i1, last)], v2 = tsrc[min(i2, last)], v3 = tsrc[min(i3, last)];
            tdst[i0] = v0;
            if (i1 < total) tdst[i1] = v1;
            if (i2 < total) tdst[i2] = v2;
            if (i3 < total) tdst[i3] = v3;
        }
        const uint4 *csrc = reinterpret_cast<const uint4 *>(consts) + q0 * 4;
        uint4 *cdst = reinterpret_cast<uint4 *>(bm_lds + static_cast<size_t>(quads_per_block) * kQuadBytes);
        for (uint32_t i = tid; i < nq * 4; i += 256) cdst[i] = csrc[i];
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t i = ks * 8 + 2 * g + h;
#pragma unroll
            for (int pr = 0; pr < PAIRS; pr++) {
                if (i < ibase) {
                    ld[ks][pr][h] ^= static_cast<int>(0x80808080u);
                } else if (i == ibase) {   // the constant slot (both coefficients of the pair: words 0 and 2)
                    ld[ks][pr][h].x = 1;
                    ld[ks][pr][h].z = 1;
                }
            }
        }
    if (copies) {   // the digit's own limbs, already in NTT form (src/rns_bconv.cu:522-528)
        constexpr uint32_t kChunks = kCoeffs / 2, kRowsPerPass = 256 / kChunks;
        for (uint32_t i = tid / kChunks; i < ibase; i += kRowsPerPass)
            if (bconv_owned(bt, skip_start + i)) {
                const size_t off = static_cast<size_t>(bx) * kCoeffs + (tid % kChunks) * 2;
                *reinterpret_cast<uint4 *>(dst + static_cast<size_t>(skip_start + i) * n + off) =
                    *reinterpret_cast<const uint4 *>(src + static_cast<size_t>(i) * n + off);
            }
    }
    __syncthreads();
    const uint4 *lds_c = reinterpret_cast<const uint4 *>(bm_lds + static_cast<size_t>(quads_per_block) * kQuadBytes);
    for (uint32_t q = 0; q < nq; q++) {
        bm_v4i a[2][KSTEPS];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++)
                a[t][ks] = reinterpret_cast<const bm_v4i *>(bm_lds + (q * 2 * KSTEPS + t * KSTEPS + ks) * kBmOperandBytes)[lane];
        const uint4 cst = lds_c[q * 4 + g];
        const uint64_t p = pack64(cst.x, cst.y), mu = pack64(cst.z, cst.w);
        const uint32_t j = (q0 + q) * 4 + g, jo = j + (j >= skip_start ? skip_len : 0);
        const bool store = j < obase && bconv_owned(bt, jo);
#pragma unroll
        for (int pr = 0; pr < PAIRS; pr++) {
            uint64_t out[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                bm_v4i d0 = {static_cast<int>(kBmfBiasWord), 0, static_cast<int>(kBmfBiasWord), 0}, d1 = d0;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    const bm_v4i bop = {ld[ks][pr][0][2 * e], ld[ks][pr][0][2 * e + 1], ld[ks][pr][1][2 * e], ld[ks][pr][1][2 * e + 1]};
                    d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0][ks], bop, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1][ks], bop, d1, 0, 0, 0);
                }
                out[e] = bm_finish(d0, d1, p, mu);
            }
            if (store) {
                const uint4 o = {lo32(out[0]), hi32(out[0]), lo32(out[1]), hi32(out[1])};
                *reinterpret_cast<uint4 *>(dst + static_cast<size_t>(jo) * n + k0 + 32 * pr) = o;
            }
        }
    }
}

// Base conversion with the overflow corrected and the remainder centred (Halevi-Polyakov-Shoup): besides the IB
// residues y_i < b_i the lane forms e = rint(sum_i y_i / b_i) in FP64 — the number of times prod(b) must be taken off
// sum_i y_i * BHat_i for the CENTRED remainder — and feeds it to the dot product as an extra input against the
// matrix column (-prod(b)) mod q_j.  The conversion is then exact (a quotient built on it is rounded to nearest:
// error <= 1/2 instead of up to IB units of overflow plus the floor's bias), except when sum_i y_i / b_i falls within
// ~IB * 2^-53 of a half-integer, where e may be off by one (one unit of error, probability ~2^-49 per coefficient).
// inv_b[i] = 1.0 / b_i; products and sums in this order, no contraction: the oracle's restatement does the same.
template <int IB>
__global__ __launch_bounds__(256) void bconv_centred_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ y,
                                                            const uint64_t *__restrict__ matrix /* [obase][IB + 1] */,
                                                            const double *__restrict__ inv_b,
                                                            const DModulus *__restrict__ obase_mod,
                                                            const uint32_t *__restrict__ obase_tidx, uint32_t obase_size,
                                                            uint32_t limbs_per_block, uint32_t n, size_t dst_stride,
                                                            size_t y_stride) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    dst += blockIdx.z * dst_stride;
    y += blockIdx.z * y_stride;
    uint64_t in[IB + 1];
    double v = 0.0;
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int i = 0; i < IB; i++) {
            in[i] = y[static_cast<size_t>(i) * n + k];
            const double t = static_cast<double>(in[i]) * inv_b[i];
            v = v + t;
        }
    }
    in[IB] = static_cast<uint64_t>(__builtin_rint(v));
    const uint32_t j0 = blockIdx.y * limbs_per_block;
    const uint32_t j1 = min(j0 + limbs_per_block, obase_size);
    for (uint32_t j = j0; j < j1; j++) {
        const u128_t acc = dot_residues<IB + 1>(in, matrix + static_cast<size_t>(j) * (IB + 1));
        dst[static_cast<size_t>(j) * n + k] = barrett128(acc, obase_mod[obase_tidx[j]]);
    }
}

// generic fallback for ibase sizes without an instantiation
__global__ __launch_bounds__(256) void bconv_matmul_generic_kernel(uint64_t *dst, const uint64_t *y,
                                                                   const uint64_t *matrix, const DModulus *obase_mod,
                                                                   const uint32_t *obase_tidx, uint32_t ibase_size,
                                                                   uint32_t obase_size, uint32_t limbs_per_block,
                                                                   uint32_t n, uint32_t skip_start, uint32_t skip_len,
                                                                   BconvBatch bt) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint64_t *src;
    uint32_t digit;
    bconv_batch_offsets(bt, blockIdx.z, dst, y, matrix, obase_tidx, skip_start, src, digit);
    if (src != nullptr && blockIdx.y == 0) {
        for (uint32_t i = 0; i < ibase_size; i++)
            if (bconv_owned(bt, skip_start + i)) dst[static_cast<size_t>(skip_start + i) * n + k] = src[static_cast<size_t>(i) * n + k];
    }
    const uint32_t j0 = blockIdx.y * limbs_per_block;
    const uint32_t j1 = min(j0 + limbs_per_block, obase_size);
    for (uint32_t j = j0; j < j1; j++) {
        const uint32_t jo = j + (j >= skip_start ? skip_len : 0);
        if (!bconv_owned(bt, jo)) continue;
        const uint64_t *row = matrix + static_cast<size_t>(j) * ibase_size;
        u128_t acc = {0, 0};
        for (uint32_t i = 0; i < ibase_size; i++) {
            const uint32_t r = bt.in_rows_set ? bt.in_row[(digit * bt.in_row_stride + i) & (kBconvMaxRows - 1)] : i;
            add128(acc, mul64x64(y[static_cast<size_t>(r) * n + k], row[i]));
        }
        const DModulus m = obase_mod[obase_tidx[j]];
        dst[static_cast<size_t>(jo) * n + k] = barrett128(acc, m);
    }
}

// alpha == 1 modup (src/rns_bconv.cu:432-453): every digit is one limb; dst[d][j][k] is the raw
// NTT-form limb for j == d and (coeff mod p_j) otherwise.
// rows.set != 0: the coefficient form of limb d is row rows.row[d] of cks_coeff (a gathered, padded buffer)
struct CoeffRows {
    uint32_t set;
    uint16_t row[kBconvMaxRows];
};
__global__ __launch_bounds__(256) void modup_single_p_kernel(uint64_t *dst, const uint64_t *cks_ntt,
                                                             const uint64_t *cks_coeff, const DModulus *mod,
                                                             uint32_t size_Ql, uint32_t size_Q, uint32_t n,
                                                             uint32_t j_begin, uint32_t j_end, CoeffRows rows) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t d = blockIdx.y;
    const uint32_t size_QlP = size_Ql + 1;
    const uint64_t coeff = cks_coeff[static_cast<size_t>(rows.set ? rows.row[d & (kBconvMaxRows - 1)] : d) * n + k];
    const uint64_t in_prime = mod[d].value;
    uint64_t *o = dst + static_cast<size_t>(d) * size_QlP * n + k;
    for (uint32_t j = j_begin; j < j_end; j++) {  // output limbs [j_begin, j_end) of [0, size_QlP)
        uint64_t v;
        if (j == d) {
            v = cks_ntt[static_cast<size_t>(d) * n + k];
        } else {
            const DModulus m = mod[j < size_Ql ? j : size_Q];
            v = in_prime > m.value ? barrett64(coeff, m.value, m.ratio1) : coeff;
        }
        o[static_cast<size_t>(j) * n] = v;
    }
}

// dst[j][k] = src[k] mod q_j for j in [j_begin, j_end) (src/rns_bconv.cu:691-707, src/rns.cu:1125-1136)
__global__ __launch_bounds__(256) void reduce_single_limb_kernel(uint64_t *dst, const uint64_t *src,
                                                                 const DModulus *mod, uint32_t j_begin, uint32_t j_end,
                                                                 uint32_t n, size_t dst_poly_stride,
                                                                 size_t src_poly_stride) {
    const uint32_t k = (blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(src + blockIdx.y * src_poly_stride + k);
    uint64_t *o = dst + blockIdx.y * dst_poly_stride + k;
    for (uint32_t j = j_begin; j < j_end; j++) {
        const DModulus m = mod[j];
        ulonglong2 v;
        v.x = barrett64(c.x, m.value, m.ratio1);
        v.y = barrett64(c.y, m.value, m.ratio1);
        *reinterpret_cast<ulonglong2 *>(o + static_cast<size_t>(j) * n) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// key-switch inner product (src/eval_key_switch.cu:12-66): two 128-bit dot products over the
// beta digits per coefficient, one Barrett each.  Pure streaming: 16 B per lane per access.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void key_switch_inner_prod_kernel(uint64_t *dst, const uint64_t *t_mod_up,
                                                                    const uint64_t *const *evks, const DModulus *mod,
                                                                    uint32_t n, uint32_t size_QP, uint32_t size_QlP,
                                                                    uint32_t size_Q, uint32_t size_Ql, uint32_t beta,
                                                                    uint32_t limb0, const uint32_t *perm = nullptr,
                                                                    const uint64_t *own = nullptr, uint32_t alpha = 0) {
    const uint32_t per_limb = n / 512;
    const uint32_t nid = limb0 + blockIdx.x / per_limb;  // limbs [limb0, limb0 + gridDim.x / per_limb)
    const uint32_t k = ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const uint32_t twr = nid >= size_Ql ? size_Q + (nid - size_Ql) : nid;
    const DModulus m = mod[twr];
    const size_t c2_id = static_cast<size_t>(nid) * n + k;
    const size_t evk_id = static_cast<size_t>(twr) * n + k;
    const size_t size_QlP_n = static_cast<size_t>(size_QlP) * n, size_QP_n = static_cast<size_t>(size_QP) * n;
    u128_t a0x = {0, 0}, a0y = {0, 0}, a1x = {0, 0}, a1y = {0, 0};
    // own != nullptr: the modup left the digits' own limbs where they were (own [size_Ql][n], NTT form) instead of
    // copying them into t_mod_up (src/rns_bconv.cu:522-528): limb nid of digit nid / alpha is read from there
    const uint32_t own_digit = own != nullptr && nid < size_Ql ? nid / alpha : ~0u;
    for (uint32_t d = 0; d < beta; d++) {
        ulonglong2 t;
        const uint64_t *row = d == own_digit ? own + static_cast<size_t>(nid) * n : t_mod_up + d * size_QlP_n + static_cast<size_t>(nid) * n;
        if (perm == nullptr) {
            t = *reinterpret_cast<const ulonglong2 *>(row + k);
        } else {  // hoisted rotation: the automorphism is applied to t_mod_up on the fly (index gather)
            t.x = row[perm[k]];
            t.y = row[perm[k + 1]];
        }
        const uint64_t *evk = evks[d];
        const ulonglong2 e0 = *reinterpret_cast<const ulonglong2 *>(evk + evk_id);
        const ulonglong2 e1 = *reinterpret_cast<const ulonglong2 *>(evk + size_QP_n + evk_id);
        add128(a0x, mul64x64(t.x, e0.x));
        add128(a0y, mul64x64(t.y, e0.y));
        add128(a1x, mul64x64(t.x, e1.x));
        add128(a1y, mul64x64(t.y, e1.y));
    }
    ulonglong2 r0, r1;
    r0.x = barrett128(a0x, m);
    r0.y = barrett128(a0y, m);
    r1.x = barrett128(a1x, m);
    r1.y = barrett128(a1y, m);
    *reinterpret_cast<ulonglong2 *>(dst + c2_id) = r0;
    *reinterpret_cast<ulonglong2 *>(dst + size_QlP_n + c2_id) = r1;
}

// The same inner product for a batch of ciphertexts that share the key (pha_apply_galois_batched): a workgroup
// takes BPW ciphertexts of one (limb, tile) and reads every key word once for all of them — the key is the largest
// operand (144 MiB at configs[2] against 72 MiB of t_mod_up per ciphertext).  t_mod_up [batch][beta][size_QlP][n],
// dst [batch][2][size_QlP][n]; gridDim.x = size_QlP * (n / 512) * ceil(batch / BPW).
template <int BPW>
__global__ __launch_bounds__(256) void key_switch_inner_prod_batched_kernel(uint64_t *dst, const uint64_t *t_mod_up,
                                                                            const uint64_t *const *evks, const DModulus *mod,
                                                                            uint32_t n, uint32_t size_QP, uint32_t size_QlP,
                                                                            uint32_t size_Q, uint32_t size_Ql, uint32_t beta,
                                                                            uint32_t batch) {
    const uint32_t per_limb = n / 512, groups = (batch + BPW - 1) / BPW;
    const uint32_t bx = blockIdx.x / groups, b0 = (blockIdx.x % groups) * BPW;
    const uint32_t nid = bx / per_limb;
    const uint32_t k = ((bx % per_limb) * 256 + threadIdx.x) * 2;
    const uint32_t twr = nid >= size_Ql ? size_Q + (nid - size_Ql) : nid;
    const DModulus m = mod[twr];
    const size_t c2_id = static_cast<size_t>(nid) * n + k;
    const size_t evk_id = static_cast<size_t>(twr) * n + k;
    const size_t size_QlP_n = static_cast<size_t>(size_QlP) * n, size_QP_n = static_cast<size_t>(size_QP) * n;
    u128_t acc[BPW][4];
#pragma unroll
    for (int b = 0; b < BPW; b++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[b][i] = u128_t{0, 0};
    for (uint32_t d = 0; d < beta; d++) {
        const uint64_t *evk = evks[d];
        const ulonglong2 e0 = *reinterpret_cast<const ulonglong2 *>(evk + evk_id);
        const ulonglong2 e1 = *reinterpret_cast<const ulonglong2 *>(evk + size_QP_n + evk_id);
#pragma unroll
        for (int b = 0; b < BPW; b++) {
            if (b0 + b >= batch) break;
            const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(t_mod_up + (static_cast<size_t>(b0 + b) * beta + d) * size_QlP_n + c2_id);
            add128(acc[b][0], mul64x64(t.x, e0.x));
            add128(acc[b][1], mul64x64(t.y, e0.y));
            add128(acc[b][2], mul64x64(t.x, e1.x));
            add128(acc[b][3], mul64x64(t.y, e1.y));
        }
    }
#pragma unroll
    for (int b = 0; b < BPW; b++) {
        if (b0 + b >= batch) break;
        uint64_t *o = dst + static_cast<size_t>(b0 + b) * 2 * size_QlP_n + c2_id;
        *reinterpret_cast<ulonglong2 *>(o) = ulonglong2{barrett128(acc[b][0], m), barrett128(acc[b][1], m)};
        *reinterpret_cast<ulonglong2 *>(o + size_QlP_n) = ulonglong2{barrett128(acc[b][2], m), barrett128(acc[b][3], m)};
    }
}

// ---------------------------------------------------------------------------------------------
// Double-hoisted linear transforms (pha_linear_transform_double_hoisted): rotated ciphertexts stay in the
// extended basis Q_l u P between the key-switch inner product and ONE moddown per giant step.
// Buffers over Q_l u P are [polys][size_QlP][n]; buffer limb j uses modulus j (j < size_Ql) or
// size_Q + (j - size_Ql).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t qp_table_index(uint32_t limb, uint32_t size_Ql, uint32_t size_Q) {
    return limb >= size_Ql ? size_Q + (limb - size_Ql) : limb;
}

// dst[p][j][i] (+)= (P mod q_j) * src[p][j][perm ? perm[i] : i] on the Q_l limbs: the ciphertext part that a key
// switch leaves untouched, lifted to Q_l u P (its P limbs are zero).  SET also zeroes the P limbs.
template <bool SET>
__global__ __launch_bounds__(256) void lift_qp_kernel(uint64_t *dst, const uint64_t *src, const uint32_t *perm,
                                                      const uint64_t *bigP_mod_q, const DModulus *mod, uint32_t n,
                                                      uint32_t size_Ql, uint32_t size_QlP) {
    const uint32_t per_limb = n / 512;
    const uint32_t poly = blockIdx.y, limb = blockIdx.x / per_limb;
    const uint32_t k = ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    uint64_t *d = dst + (static_cast<size_t>(poly) * size_QlP + limb) * n + k;
    if (limb >= size_Ql) {
        if (SET) *reinterpret_cast<ulonglong2 *>(d) = ulonglong2{0, 0};
        return;
    }
    const uint64_t *row = src + (static_cast<size_t>(poly) * size_Ql + limb) * n;
    const DModulus m = mod[limb];
    const uint64_t pj = bigP_mod_q[limb];
    ulonglong2 v;
    v.x = mul_barrett(perm ? row[perm[k]] : row[k], pj, m);
    v.y = mul_barrett(perm ? row[perm[k + 1]] : row[k + 1], pj, m);
    if (!SET) {
        const ulonglong2 o = *reinterpret_cast<const ulonglong2 *>(d);
        v.x = csub(v.x + o.x, m.value);
        v.y = csub(v.y + o.y, m.value);
    }
    *reinterpret_cast<ulonglong2 *>(d) = v;
}

constexpr int kMaxMacTerms = 32;
struct MacTerms {
    const uint64_t *rot[kMaxMacTerms];  // [2][size_QlP][n] each
    const uint64_t *pt[kMaxMacTerms];   // [size_QlP][n] each
    uint32_t count;
};

// dst[p][j][i] = sum_b pt_b[j][i] * rot_b[p][j][i] mod q_j, both polys: the inner sum of a giant step, one pass
__global__ __launch_bounds__(256) void multi_mac_qp_kernel(uint64_t *dst, MacTerms terms, const DModulus *mod, uint32_t n,
                                                           uint32_t size_Ql, uint32_t size_QlP, uint32_t size_Q) {
    const uint32_t per_limb = n / 512;
    const uint32_t limb = blockIdx.x / per_limb;
    const uint32_t k = ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const DModulus m = mod[qp_table_index(limb, size_Ql, size_Q)];
    const size_t id = static_cast<size_t>(limb) * n + k, poly1 = static_cast<size_t>(size_QlP) * n;
    u128_t a0x = {0, 0}, a0y = {0, 0}, a1x = {0, 0}, a1y = {0, 0};
    for (uint32_t b = 0; b < terms.count; b++) {
        const ulonglong2 pt = *reinterpret_cast<const ulonglong2 *>(terms.pt[b] + id);
        const ulonglong2 r0 = *reinterpret_cast<const ulonglong2 *>(terms.rot[b] + id);
        const ulonglong2 r1 = *reinterpret_cast<const ulonglong2 *>(terms.rot[b] + poly1 + id);
        add128(a0x, mul64x64(pt.x, r0.x));
        add128(a0y, mul64x64(pt.y, r0.y));
        add128(a1x, mul64x64(pt.x, r1.x));
        add128(a1y, mul64x64(pt.y, r1.y));
    }
    *reinterpret_cast<ulonglong2 *>(dst + id) = ulonglong2{barrett128(a0x, m), barrett128(a0y, m)};
    *reinterpret_cast<ulonglong2 *>(dst + poly1 + id) = ulonglong2{barrett128(a1x, m), barrett128(a1y, m)};
}

// dst += src over [2][size_QlP][n] (SET: dst = src)
template <bool SET>
__global__ __launch_bounds__(256) void add_qp_kernel(uint64_t *dst, const uint64_t *src, const DModulus *mod, uint32_t n,
                                                     uint32_t size_Ql, uint32_t size_QlP, uint32_t size_Q) {
    const uint32_t per_limb = n / 512;
    const uint32_t poly = blockIdx.y, limb = blockIdx.x / per_limb;
    const size_t id = (static_cast<size_t>(poly) * size_QlP + limb) * n + ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + id);
    if (!SET) {
        const uint64_t q = mod[qp_table_index(limb, size_Ql, size_Q)].value;
        const ulonglong2 o = *reinterpret_cast<const ulonglong2 *>(dst + id);
        v.x = csub(v.x + o.x, q);
        v.y = csub(v.y + o.y, q);
    }
    *reinterpret_cast<ulonglong2 *>(dst + id) = v;
}

// ct[l][k] = ct[l][k] + cx[l][k] mod q_l (src/rns_bconv.cu:763-769)
__global__ __launch_bounds__(256) void add_to_ct_kernel(uint64_t *ct, const uint64_t *cx, const DModulus *mod,
                                                        uint32_t n) {
    const uint32_t per_limb = n / 512;
    const uint32_t l = blockIdx.x / per_limb;
    const size_t id = static_cast<size_t>(l) * n + ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const uint64_t q = mod[l].value;
    ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(ct + id);
    const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(cx + id);
    a.x = csub(a.x + b.x, q);
    a.y = csub(a.y + b.y, q);
    *reinterpret_cast<ulonglong2 *>(ct + id) = a;
}

// ---------------------------------------------------------------------------------------------
// Galois automorphism in the NTT domain = index gather (src/galois.cu:8-18)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apply_galois_ntt_kernel(uint64_t *dst, const uint64_t *src,
                                                               const uint32_t *perm, uint32_t n) {
    const uint32_t per_limb = n / 512;
    const uint32_t l = blockIdx.x / per_limb;
    const uint32_t i = ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const uint2 p = *reinterpret_cast<const uint2 *>(perm + i);
    const uint64_t *s = src + static_cast<size_t>(l) * n;
    ulonglong2 v;
    v.x = s[p.x];
    v.y = s[p.y];
    *reinterpret_cast<ulonglong2 *>(dst + static_cast<size_t>(l) * n + i) = v;
}

// ---------------------------------------------------------------------------------------------
// dyadic ciphertext product (src/polymath.cu:468-537): d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1
// ---------------------------------------------------------------------------------------------
#ifndef PHA_TENSOR_FP64
#define PHA_TENSOR_FP64 1
#endif
template <bool SQUARE>
__global__ __launch_bounds__(256) void tensor_prod_2x2_kernel(const uint64_t *op1, const uint64_t *op2,
                                                              const DModulus *mod, uint64_t *res, uint32_t n,
                                                              uint32_t limbs, uint32_t times2 = 0, uint32_t limb0 = 0) {
    // times2: (2 d0, 2 d1, 2 d2) — the doubling of a Chebyshev product taken BEFORE any division, where it is exact
    // limb0: first limb of the window this launch covers (limb-parallel ranks take gridDim.x / per_limb limbs from there)
    const uint32_t per_limb = n / 512;
    const uint32_t l = limb0 + blockIdx.x / per_limb;
    const size_t id = static_cast<size_t>(l) * n + ((blockIdx.x % per_limb) * 256 + threadIdx.x) * 2;
    const size_t rc = static_cast<size_t>(limbs) * n;
    const DModulus m = mod[l];
    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(op1 + id);
    const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(op1 + id + rc);
    const ulonglong2 b0 = SQUARE ? a0 : *reinterpret_cast<const ulonglong2 *>(op2 + id);
    const ulonglong2 b1 = SQUARE ? a1 : *reinterpret_cast<const ulonglong2 *>(op2 + id + rc);
    ulonglong2 d0, d1, d2;
    const uint64_t q = m.value;
    if (PHA_TENSOR_FP64 && (q >> 50) == 0) {
        // residues below 2^50 as integer-valued doubles (modarith.cuh: FpField): three exact products of 6 FP64 operations
        // each instead of three 128-bit Barrett products; every stored word is the canonical residue, as on the other path
        const FpField f(m);
        auto one = [&](uint64_t x0, uint64_t x1, uint64_t y0, uint64_t y1, unsigned long long &r0, unsigned long long &r1, unsigned long long &r2) {
            const double fa0 = f.from_u64(x0), fa1 = f.from_u64(x1), fb0 = f.from_u64(y0), fb1 = f.from_u64(y1);
            const double p0 = f.mulmod(fa0, f.prep(fb0)), p2 = f.mulmod(fa1, f.prep(fb1));
            const double pk = f.mulmod(fa0 + fa1, f.prep(fb0 + fb1));      // operands < 2q: |pk| <= (1/2 + 3 * 2^-53 * 4q) q
            r0 = f.to_u64(f.canon(p0));
            r1 = f.to_u64(f.canon(pk - p0 - p2));
            r2 = f.to_u64(f.canon(p2));
        };
        one(a0.x, a1.x, b0.x, b1.x, d0.x, d1.x, d2.x);
        one(a0.y, a1.y, b0.y, b1.y, d0.y, d1.y, d2.y);
    } else {
    d0.x = mul_barrett(a0.x, b0.x, m);
    d0.y = mul_barrett(a0.y, b0.y, m);
    d2.x = mul_barrett(a1.x, b1.x, m);
    d2.y = mul_barrett(a1.y, b1.y, m);
    // Karatsuba form of the reference: (a0+a1)(b0+b1) - d0 - d2; operands < 2q, product < 2^124
    uint64_t tx = mul_barrett(a0.x + a1.x, b0.x + b1.x, m) + 2 * q - d0.x - d2.x;
    uint64_t ty = mul_barrett(a0.y + a1.y, b0.y + b1.y, m) + 2 * q - d0.y - d2.y;
    d1.x = csub(csub(tx, q), q);
    d1.y = csub(csub(ty, q), q);
    }
    if (times2) {
        d0 = ulonglong2{csub(d0.x + d0.x, q), csub(d0.y + d0.y, q)};
        d1 = ulonglong2{csub(d1.x + d1.x, q), csub(d1.y + d1.y, q)};
        d2 = ulonglong2{csub(d2.x + d2.x, q), csub(d2.y + d2.y, q)};
    }
    *reinterpret_cast<ulonglong2 *>(res + id) = d0;
    *reinterpret_cast<ulonglong2 *>(res + id + rc) = d1;
    *reinterpret_cast<ulonglong2 *>(res + id + 2 * rc) = d2;
}

// General ciphertext product (src/polymath.cu:551-599): res[j] = sum_{a + b = j} op1[a] * op2[b], every sum accumulated
// in 128 bits and reduced once.  The M + N input words of a coefficient are read before its M + N - 1 outputs are
// written, so res may alias op1 as in the reference's multiply_internal (src/evaluate.cu:270-275).  Operand sizes are
// template parameters so that the inputs stay in registers (the reference allocates them with device-side new).
template <int M, int NN>
__global__ __launch_bounds__(256) void tensor_prod_mxn_kernel(const uint64_t *op1, const uint64_t *op2, const DModulus *mod,
                                                              uint64_t *res, uint32_t n, uint32_t limbs) {
    const uint32_t per_limb = n / 256;
    const uint32_t l = blockIdx.x / per_limb;
    const size_t id = static_cast<size_t>(l) * n + (blockIdx.x % per_limb) * 256 + threadIdx.x;
    const size_t rc = static_cast<size_t>(limbs) * n;
    const DModulus m = mod[l];
    uint64_t c1[M], c2[NN];
#pragma unroll
    for (int i = 0; i < M; i++) c1[i] = op1[id + rc * i];
#pragma unroll
    for (int i = 0; i < NN; i++) c2[i] = op2[id + rc * i];
#pragma unroll
    for (int j = 0; j < M + NN - 1; j++) {
        u128_t acc = {0, 0};
#pragma unroll
        for (int a = 0; a < M; a++) {
            const int b = j - a;
            if (b >= 0 && b < NN) add128(acc, mul64x64(c1[a], c2[b]));
        }
        res[id + rc * j] = barrett128(acc, m);
    }
}

}  // namespace pha
