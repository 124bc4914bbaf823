// Negacyclic NTT / INTT kernels for gfx950 (the nwt_2d_radix8_* family of the reference,
// include/ntt.cuh:47-94, re-designed for CDNA4).
//
// Decomposition (N = 2^logN, S1 + S2 = logN, both <= 8):
//   column kernel: the S1 strided stages (t >= n2) on a [2^S1 rows][C cols] tile, two register
//                  radix-2^RA / 2^RB passes with one LDS exchange; all tiles of a limb share the
//                  same 2^S1 twiddles
//   row kernel:    the S2 contiguous stages on 4096 consecutive coefficients, two register
//                  passes, one LDS exchange between them and one LDS transpose so that the HBM
//                  side is read/written 16 B per lane, fully coalesced
// A tile is 4096 coefficients (32 KiB) per 256-thread workgroup, 16 coefficients per lane.
// Stage order and twiddle indices are exactly those of src/ntt/ntt_1d.cu:41-74,220-254
// (tw[m + k], m = 1,2,..,N/2 forward; reversed for the inverse), so results are bit-identical
// to the reference after the final reduction to [0,q).
#pragma once

#include <type_traits>

#include "modarith.cuh"
#include "numth.hpp"

namespace pha {

// One workgroup transforms a tile of kTile coefficients, 16 per thread: 4096 (256 threads), four workgroups per CU.
// Measured alternatives at configs[1] (1152 workgroups = 4.5 per CU; per-workgroup timestamps from
// tools/scratch/trace_ntt.py show the last 128 running as a second generation that ends ~3.5 us after the first):
//  * PHA_TILE_LOG=11 (128 threads, 16 KiB LDS, eight to nine workgroups per CU): row kernels unchanged (18.1 us),
//    column kernels 23.3 us against 17.4 us — their global accesses shrink to half cache lines;
//  * PHA_LDS_SWIZZLE=1 (unpadded 32 KiB image, 96 VGPRs) does NOT reach five workgroups per CU: a CU hands out
//    LDS to five workgroups only up to 31 KiB each (tools/scratch/occ.hip: 31744 B -> 5 resident, 32256 B -> 4);
//  * a timing-only build with a 30 KiB image (what keeping each lane's own element of the exchange in a register
//    would allow) did put all 1152 workgroups on the chip at once, but the kernel span only went from 16.6 to
//    15.2 us: every workgroup ran slower (median 9.1 against 7.7 us) — the memory system, not the tail, is the limit.
// PHA_LDS_SWIZZLE stays selectable for re-measurement and passes the parity tests; PHA_TILE_LOG=11 was measured in round 2
// and no longer builds since ntt_fused.cuh (written for 4096-coefficient tiles) is part of the library.
#ifndef PHA_TILE_LOG
#define PHA_TILE_LOG 12
#endif
constexpr int kTileLog = PHA_TILE_LOG;
constexpr int kTile = 1 << kTileLog;
constexpr int kThreads = kTile / 16;
#ifndef PHA_LDS_SWIZZLE
#define PHA_LDS_SWIZZLE (PHA_TILE_LOG < 12)
#endif
#if PHA_LDS_SWIZZLE
constexpr int kLdsWords = kTile;
#define PHA_NTT_BOUNDS __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5, 5)))
#else
constexpr int kLdsWords = kTile + 512;
#ifndef PHA_NTT_MIN_WAVES
#define PHA_NTT_BOUNDS __launch_bounds__(kThreads)
#else
#define PHA_NTT_BOUNDS __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PHA_NTT_MIN_WAVES)))
#endif
#endif

struct LimbMap {
    uint32_t start, count;          // buffer limbs [start, start+count)
    uint32_t size_QP, size_P;       // size_P > 0: last size_P limbs of the range use key-level table
                                    // indices size_QP - size_P + k (src/ntt/fntt_2d.cu:433-436)
    uint32_t excl_start, excl_end;  // limbs skipped (src/ntt/ntt_modup.cu:422); empty if start >= end
    uint32_t active;                // limbs actually transformed = count - |excluded| (max over polys)
    uint32_t polys;                 // buffers [polys][...] handled by this launch
    uint32_t digit_alpha;           // > 0: poly p is digit p of a modup buffer and skips its own limbs
    uint32_t digit_limit;           //      [p*alpha, min((p+1)*alpha, digit_limit))  (src/rns_bconv.cu:616-620)
    uint32_t tidx_bias;             // size_P == 0: table index = buffer limb + tidx_bias (a window of the P limbs of
                                    // a Q_l u P buffer in the limb-parallel key switch)
    uint32_t digit_count;           // > 0 with digit_alpha: buffers are [batch][digit_count] and poly p is digit
                                    // p % digit_count (batched key switch)
    uint32_t xcd_map;               // 1: limb-to-XCD placement of decode_block (default), 0: tile-interleaved placement
    uint32_t out_limb0;             // inverse transform: buffer limb l of the input lands on limb l - out_limb0 of the
    size_t out_poly_stride;         // output, whose polys are out_poly_stride words apart (0: the input's stride) --
                                    // e.g. the last limb of every poly gathered into a [polys][N] buffer (rescale)
    uint32_t polys_magic;           // ceil(2^32 / polys): decode_block divides by polys with one multiply (set by
                                    // the launcher; exact for dividends < 2^20 and polys <= 2^11)
    const uint64_t *bcast_src;      // forward column kernel, BCAST variant: every limb's input is this ONE limb in
    size_t bcast_poly_stride;       // coefficient form ([polys] limbs, bcast_poly_stride words apart) reduced modulo the
                                    // limb's prime on load -- the broadcast of rescale and of the alpha = 1 moddown
                                    // (src/rns.cu:1125-1136, src/rns_bconv.cu:691-707) without its pass through memory
};

struct NttTables {
    const DModulus *mod;   // [size_QP]
    const tw_t *tw;        // [size_QP][n] canonical order {w, w_shoup}
    const tw_t *twD;       // [size_QP][(n >> RB) / 64][2^RB - 1][64] last-pass order
    const uint64_t *n_inv, *n_inv_shoup;  // [size_QP]
    uint32_t log_n;
    uint32_t twD_stride;   // records per limb in twD
    uint32_t fp64;         // limbs with q < 2^50 run on FpField; their table regions hold one double w per
                           // twiddle (8-byte records in the same order, first half of the region)
    uint32_t split;        // N = 2^17: 1 = the outermost radix-2 stage runs in its own kernel (split_*_kernel) and
                           // the column kernels transform the two halves as subtrees rooted at twiddle nodes 2, 3
    const double *twG;     // [size_QP][kGenWords] generators of the last pass's twiddles (FP64 limbs, last pass of four
                           // stages); nullptr: every last-pass twiddle is read from twD
};
constexpr uint32_t kGenWords = 304;   // = kGenStride; offsets kGenOmega, kGenAlpha, kGenBeta: numth.hpp (build_generator_table)

// out[l][e] = (sub[l][e] - x) * cst[l] (+ addend[l][e])   or   x * cst[l]
struct Epilogue {
    uint64_t *out;
    const uint64_t *sub;
    const uint64_t *addend;
    const uint64_t *cst, *cst_shoup;  // indexed by buffer limb - cst_limb0
    uint32_t cst_limb0;
    size_t out_poly_stride, sub_poly_stride;
    uint32_t addend_every;  // > 0: only polys p with p % addend_every == 0 take the addend (a rotation adds pi(c0) to
                            // the first poly of each ciphertext and nothing to the second)
};

enum : int { EPI_NONE = 0, EPI_SUB_MUL = 1, EPI_SUB_MUL_ADD = 2, EPI_SCALE = 3 };

// Non-temporal result stores (so that no dirty lines wait in the per-XCD L2s for the end-of-kernel write-back)
// were measured and lost: column kernel 19.0 us against 17.4 us at 72 limbs.  Kept behind a macro for re-measurement.
#ifndef PHA_NT_STORES
#define PHA_NT_STORES 0
#endif
#ifndef PHA_SC1_STORES
#define PHA_SC1_STORES 2
#endif
#ifndef PHA_EPI_PREFETCH
#define PHA_EPI_PREFETCH 2
#endif
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
// threadIdx.x as a value the optimiser cannot see through: inside a persistent loop (ntt_fused.cuh) everything derived
// from the lane index would otherwise be hoisted out of the loop (300 VGPRs of addresses and LDS offsets, spilled)
__device__ __forceinline__ uint32_t pha_tid() {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ void st_stream(uint64_t *p, uint64_t v) {
#if PHA_SC1_STORES >= 2
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#elif PHA_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_sc1(uint64_t *p, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_stream2(uint64_t *p, ulonglong2 v) {
#if PHA_SC1_STORES
    // write-through (sc1) 16-B store: the bytes leave the XCD's L2 while the kernel runs instead of in the write-back
    // at its end
    const u64x2_t w = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
#elif PHA_NT_STORES
    u64x2_t w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<u64x2_t *>(p));
#else
    *reinterpret_cast<ulonglong2 *>(p) = v;
#endif
}

// Tile accesses through a buffer descriptor: a workgroup-uniform base (scalar registers), one 32-bit byte offset per lane
// and a scalar offset per row group -- the 64-bit address arithmetic of a strided access (2-3 VALU instructions each)
// moves to the scalar unit, and the stores are compiler-visible instructions instead of asm statements (LDS reads and
// arithmetic are scheduled around them).  Stores carry sc1 like st_stream (write-through).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = PHA_SC1_STORES ? 16 : 0;   // cache-policy bit 4 = sc1 on gfx94x/gfx950
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const uint64_t *ubase) {
    // the base IS workgroup-uniform (it derives from blockIdx and kernel arguments); reading it from the first lane says so
    // to the compiler where its own analysis gives up -- otherwise every access is wrapped in a waterfall loop
    const uint64_t b = reinterpret_cast<uint64_t>(ubase);
    // (the builtin returns a signed int: widen through uint32_t, or a low word with bit 31 set smears over the high word)
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32)));
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(b)));
    const uint64_t u = (static_cast<uint64_t>(hi) << 32) | lo;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint64_t *>(u), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint64_t ld_tile(__amdgpu_buffer_rsrc_t r, uint32_t lane_byte_off, uint32_t uniform_byte_off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_byte_off, uniform_byte_off, 0);
    return (static_cast<uint64_t>(v.y) << 32) | v.x;
}
__device__ __forceinline__ void st_tile(__amdgpu_buffer_rsrc_t r, uint32_t lane_byte_off, uint32_t uniform_byte_off, uint64_t v) {
    const u32x2_t w = {static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(w, r, lane_byte_off, uniform_byte_off, kAuxSc1);
}
#ifndef PHA_BUF_MASK
#define PHA_BUF_MASK 15   // bisecting aid: 1 forward column, 2 forward row, 4 inverse row, 8 inverse column kernel
#endif
__device__ __forceinline__ void st_tile2(__amdgpu_buffer_rsrc_t r, uint32_t lane_byte_off, uint32_t uniform_byte_off, ulonglong2 v) {
    const u32x4_t w = {static_cast<uint32_t>(v.x), static_cast<uint32_t>(v.x >> 32), static_cast<uint32_t>(v.y),
                       static_cast<uint32_t>(v.y >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, lane_byte_off, uniform_byte_off, kAuxSc1);
}

// Development aid (tools/scratch/trace_ntt.py): per-workgroup timestamps of the forward kernels' phases.
#ifdef PHA_TRACE
__device__ unsigned long long *pha_trace_buf;
#define PHA_TRACE_MARK(slot)                                                                      \
    do {                                                                                          \
        if (threadIdx.x == 0 && pha_trace_buf) pha_trace_buf[blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define PHA_TRACE_MARK(slot)
#endif
// -DPHA_TRACE=3 (tools/scratch/trace2.py): two timestamps per workgroup only -- entry, before the loads are issued, and
// exit, after the stores are issued -- written by thread 0 at the exit.  The five phase marks above cost a workgroup
// 2-3 us of its life (each waits for its clock read); this form changes the kernel time by less than 1 us.
#if defined(PHA_TRACE) && PHA_TRACE == 3
#undef PHA_TRACE_MARK
#define PHA_TRACE_MARK(slot)
#define PHA_TRACE_ENTRY() const unsigned long long pha_trace_t0 = wall_clock64()
#define PHA_TRACE_EXIT()                                                 \
    do {                                                                 \
        if (threadIdx.x == 0 && pha_trace_buf) {                         \
            pha_trace_buf[blockIdx.x * 8] = pha_trace_t0;                \
            pha_trace_buf[blockIdx.x * 8 + 4] = wall_clock64();          \
        }                                                                \
    } while (0)
#else
#define PHA_TRACE_ENTRY()
#define PHA_TRACE_EXIT()
#endif

// Wave priorities (experiment, -DPHA_GEN_PRIO=1/2/3; profiles/EXPERIMENTS.md 3.13): 1 = a workgroup's waves take the
// priority 3 - min(3, blockIdx / 256) at entry (the four workgroups a CU holds are one of each 256-block dispatch round,
// so the oldest issues first and reaches its stores while the others still compute); 2 = the same plus priority 3 behind
// the exchange barrier (whoever is in its last pass goes first); 3 = only the latter.
#ifndef PHA_GEN_PRIO
#define PHA_GEN_PRIO 0
#endif
__device__ __forceinline__ void prio_entry() {
#if PHA_GEN_PRIO == 1 || PHA_GEN_PRIO == 2
    const uint32_t gen = blockIdx.x >> 8;
    if (gen == 0)
        __builtin_amdgcn_s_setprio(3);
    else if (gen == 1)
        __builtin_amdgcn_s_setprio(2);
    else if (gen == 2)
        __builtin_amdgcn_s_setprio(1);
#endif
}
__device__ __forceinline__ void prio_late() {
#if PHA_GEN_PRIO >= 2
    __builtin_amdgcn_s_setprio(3);
#endif
}

// A value that is the same in every lane of the workgroup BY CONSTRUCTION (it derives from blockIdx and kernel arguments
// only), said to the compiler where its own analysis gives up (the block decode divides): what is indexed by it -- the
// modulus, N^-1, the epilogue constants, the table and tile base pointers -- is then fetched through the scalar cache and
// kept in scalar registers instead of queueing behind the tile's vector loads.
__device__ __forceinline__ uint32_t uniform(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v)));
}

// returns false when this workgroup has nothing to do (ragged last digit)
__device__ __forceinline__ bool resolve_limb(const LimbMap &m, uint32_t poly, uint32_t li, uint32_t &twr, uint32_t &tidx) {
    uint32_t es = m.excl_start, ee = m.excl_end;
    if (m.digit_alpha) {
        es = (m.digit_count ? poly % m.digit_count : poly) * m.digit_alpha;
        ee = min(es + m.digit_alpha, m.digit_limit);
    }
    // only the part of the skipped range that lies inside this launch's window counts
    es = max(es, m.start);
    ee = min(ee, m.start + m.count);
    const uint32_t len = ee > es ? ee - es : 0;
    if (li >= m.count - len) return false;
    twr = m.start + li;
    if (len && twr >= es) twr += len;
    tidx = (m.size_P && twr >= m.start + m.count - m.size_P) ? m.size_QP - (m.start + m.count - twr)
                                                             : twr + m.tidx_bias;
    return true;
}

// Work decomposition of a launch: one workgroup per (poly, limb, tile).  Workgroups b and b + 8 share an XCD
// (round-robin dispatch: observed, not promised -- placement is a speed matter only, never a correctness one).
//
// Limb-to-XCD placement (xcd_map = 1): every tile of limb slot li, of every poly, runs on XCD li % 8 -- in BOTH kernels
// of a transform and in every launch with the same limb window -- so that a kernel reads what a workgroup of the same
// XCD wrote.  Measured on MI355X (tools/scratch/l2_persist.hip, profiles/r2a_l2_persist.log): a kernel that reads
// 36 MiB which the previous kernel on the stream has just written takes 4.7 us when each block reads what a block of
// the same XCD wrote and 17.7 us when another XCD wrote it.  On the transforms themselves the gain shows where a launch
// is short: 2 x 12 limbs forward 25.0 -> 16.2 us, 2 x 36 limbs 34.8 -> 34.5 us (profiles/r2a_xcd_placement.log).
// The limbs beyond a multiple of 8 are dealt to the XCDs unit by unit ((limb, poly) pairs) so that the load stays even
// (configs[1]: 36 limbs x 2 polys = 4 full rounds + 8 units, 9 units per XCD).  Within an XCD's sequence the polys of a
// tile are adjacent: they read the same twiddles and the second read hits that XCD's L2.
struct BlockWork {
    uint32_t poly, li, ti;
};
__host__ __device__ __forceinline__ uint32_t xcd_grid_blocks(uint32_t active, uint32_t polys, uint32_t tiles) {
    const uint32_t full = active >> 3, left = (active & 7u) * polys;
    return 8u * (full * polys * tiles + ((left + 7u) >> 3) * tiles);
}
// x / polys for x < 2^20 (polys_magic = ceil(2^32 / polys) does not fit 32 bits for polys = 1)
__device__ __forceinline__ uint32_t div_polys(uint32_t x, const LimbMap &m) {
    return m.polys == 1 ? x : __umulhi(x, m.polys_magic);
}
__device__ __forceinline__ bool decode_block(uint32_t b, const LimbMap &m, uint32_t tiles, BlockWork &w) {
    const uint32_t polys = m.polys, active = m.active;
    if (m.xcd_map) {
        // tiles is a power of two and x / polys = umulhi(x, polys_magic): no integer division in the prologue (a u32
        // division is ~40 VALU instructions; the round-1 decode ran five of them before the first global load)
        const uint32_t tl = 31u - static_cast<uint32_t>(__builtin_clz(tiles));
        const uint32_t x = b & 7u, s = b >> 3, full = active >> 3, main_part = (full * polys) << tl;
        if (s < main_part) {
            const uint32_t k = div_polys(s >> tl, m);                  // s / (polys * tiles)
            const uint32_t rem = s - ((k * polys) << tl);
            w.li = k * 8 + x;
            w.ti = div_polys(rem, m);
            w.poly = rem - w.ti * polys;
            return true;
        }
        const uint32_t s2 = s - main_part, v = (s2 >> tl) * 8 + x;
        if (v >= (active & 7u) * polys) return false;
        const uint32_t vl = div_polys(v, m);
        w.li = full * 8 + vl;
        w.poly = v - vl * polys;
        w.ti = s2 & (tiles - 1);
        return true;
    }
    const uint32_t total_tiles = active * tiles;
    if (b >= total_tiles * polys) return false;
    uint32_t tile;
    if ((total_tiles & 7u) == 0) {
        const uint32_t x = b & 7u, u = b >> 3;
        w.poly = u % polys;
        tile = (u / polys) * 8 + x;
    } else {
        w.poly = b % polys;
        tile = b / polys;
    }
    w.li = tile / tiles;
    w.ti = tile % tiles;
    return true;
}

template <class TW>
struct CanonTw {
    const TW *tw;
    uint32_t node;
    __device__ __forceinline__ TW operator()(int s, int jb) const { return tw[(node << s) + jb]; }
};

// twiddles of a radix-2^R block already in registers (slot (1 << s) - 1 + jb), fetched ahead of the barrier in front of
// the block so that their latency is not paid stage by stage behind it
template <int R, class TW>
struct PreTw {
    TW w[(1 << R) - 1];
    __device__ __forceinline__ TW operator()(int s, int jb) const { return w[(1 << s) - 1 + jb]; }
};
template <int R, class TW>
__device__ __forceinline__ PreTw<R, TW> prefetch_tw(const TW *tw, uint32_t node) {
    PreTw<R, TW> r;
#pragma unroll
    for (int s = 0; s < R; s++)
#pragma unroll
        for (int jb = 0; jb < (1 << s); jb++) r.w[(1 << s) - 1 + jb] = tw[(node << s) + jb];
    return r;
}
// the inverse column kernel has the registers for the integer path's 16-byte {w, Shoup companion} records too (124 VGPRs;
// in the inverse row kernel the same costs the FP64 limbs more than it gives the integer ones: profiles/EXPERIMENTS.md 3.16)
#ifndef PHA_TW_PREFETCH_INV16
#define PHA_TW_PREFETCH_INV16 1
#endif
#ifndef PHA_TW_PREFETCH
#define PHA_TW_PREFETCH 2     // 1: the pass behind the exchange barrier only (round-3 default until r3y); 2: also the first pass
#endif

template <int RB, class TW>
struct LastPassTw {
    const TW *base;  // already offset to (chunk, lane)
    __device__ __forceinline__ TW operator()(int s, int jb) const { return base[((1 << s) - 1 + jb) * 64]; }
};

// The same 15 twiddles derived instead of loaded (FP64 limbs): stage s, sub-block jb uses theta^(2^(3-s)) * omega[s][jb],
// theta = alpha[tile] * beta[lane's group] (NttTables::twG).  One 8-byte load per lane and tile replaces fifteen (a third of
// a row pass's traffic); the price is 15 modular products per 32 butterflies.  Any representative of the residue class
// with |w| <= q serves the butterflies, and every stored result is canonical, so results do not change.
struct GenTw {
    const FpField &f;
    const double *omega;      // 16 doubles in LDS
    double pw[4];             // theta^8, theta^4, theta^2, theta
    __device__ __forceinline__ GenTw(const FpField &f_, const double *omega_, double theta) : f(f_), omega(omega_) {
        pw[3] = theta;
        pw[2] = f.mulred(theta, theta);
        pw[1] = f.mulred(pw[2], pw[2]);
        pw[0] = f.mulred(pw[1], pw[1]);
    }
    __device__ __forceinline__ double operator()(int s, int jb) const {
        return jb == 0 ? pw[s] : f.mulred(pw[s], omega[(1 << s) - 1 + jb]);
    }
};
// generator operands of a row tile: omega goes to LDS (lanes 0..15), the lane keeps theta
template <class F>
__device__ __forceinline__ double gen_theta(const F &, const double *, double *, uint32_t, uint32_t) { return 0.0; }
template <>
__device__ __forceinline__ double gen_theta<FpField>(const FpField &f, const double *twg, double *lds_omega, uint32_t ti, uint32_t t) {
    if (t < 16) lds_omega[t] = twg[kGenOmega + t];
    return f.mulred(twg[kGenAlpha + ti], twg[kGenBeta + t]);
}

// radix-2^R decimation-in-time block: stage s uses twiddle node*2^s + jb on sub-block jb.
// G0 = index of the block's first stage within its kernel; the field policies (modarith.cuh)
// reduce on even kernel-local stages only.
// -DPHA_NO_MATH=1: timing-only build without butterflies (wrong results): what the loads, the LDS exchange and the stores of
// a pass cost in the same launch structure (profiles/EXPERIMENTS.md 3.14)
#ifndef PHA_NO_MATH
#define PHA_NO_MATH 0
#endif
#ifndef PHA_TW_PIPELINE
#define PHA_TW_PIPELINE 1     // 0: every stage fetches its own twiddles (A/B: profiles/EXPERIMENTS.md 3.19)
#endif
template <int R, int G0, class F, class Tw>
__device__ __forceinline__ void ct_radix(typename F::elem (&x)[1 << R], const Tw &tw, const F &f) {
    if (PHA_NO_MATH) return;
#if PHA_TW_PIPELINE
    // the twiddles of stage s + 1 are requested in front of the butterflies of stage s: a lazily loading Tw (table reads) then
    // waits for memory once per block instead of once per stage; register-resident and derived twiddles are unaffected
    using W = decltype(tw(0, 0));
    W cur[1 << (R - 1)], nxt[1 << (R - 1)];
    cur[0] = tw(0, 0);
#pragma unroll
    for (int s = 0; s < R; s++) {
        const int half = 1 << (R - 1 - s);
        if (s + 1 < R) {
#pragma unroll
            for (int jb = 0; jb < (2 << s); jb++) nxt[jb] = tw(s + 1, jb);
        }
#pragma unroll
        for (int jb = 0; jb < (1 << s); jb++) {
            const typename F::twt w = f.prep(cur[jb]);
#pragma unroll
            for (int i = 0; i < half; i++) {
                typename F::elem &a = x[jb * 2 * half + i], &b = x[jb * 2 * half + half + i];
                if (((G0 + s) & 1) == 0)
                    f.template ct<0>(a, b, w);
                else
                    f.template ct<1>(a, b, w);
            }
        }
        if (s + 1 < R) {
#pragma unroll
            for (int jb = 0; jb < (2 << s); jb++) cur[jb] = nxt[jb];
        }
    }
#else
#pragma unroll
    for (int s = 0; s < R; s++) {
        const int half = 1 << (R - 1 - s);
#pragma unroll
        for (int jb = 0; jb < (1 << s); jb++) {
            const typename F::twt w = f.prep(tw(s, jb));
#pragma unroll
            for (int i = 0; i < half; i++) {
                typename F::elem &a = x[jb * 2 * half + i], &b = x[jb * 2 * half + half + i];
                if (((G0 + s) & 1) == 0)
                    f.template ct<0>(a, b, w);
                else
                    f.template ct<1>(a, b, w);
            }
        }
    }
#endif
}

// inverse block; LAST: the final stage multiplies the sum half by N^-1 (the difference half gets
// it through itw[1], src/util/ntt.cu:53-55) and reduces everything to [0,q)
template <int R, bool LAST, class F, class Tw>
__device__ __forceinline__ void gs_radix(typename F::elem (&x)[1 << R], const Tw &tw, const F &f,
                                         const typename F::twt &n_inv) {
    if (PHA_NO_MATH) return;
#if PHA_TW_PIPELINE
    using W = decltype(tw(0, 0));
    W cur[1 << (R - 1)], nxt[1 << (R - 1)];
#pragma unroll
    for (int jb = 0; jb < (1 << (R - 1)); jb++) cur[jb] = tw(R - 1, jb);
#pragma unroll
    for (int s = R - 1; s >= (LAST ? 1 : 0); s--) {
        const int half = 1 << (R - 1 - s);
        if (s > 0) {
#pragma unroll
            for (int jb = 0; jb < (1 << (s - 1)); jb++) nxt[jb] = tw(s - 1, jb);
        }
#pragma unroll
        for (int jb = 0; jb < (1 << s); jb++) {
            const typename F::twt w = f.prep(cur[jb]);
#pragma unroll
            for (int i = 0; i < half; i++) f.gs(x[jb * 2 * half + i], x[jb * 2 * half + half + i], w);
        }
        if (s > 0) {
#pragma unroll
            for (int jb = 0; jb < (1 << (s - 1)); jb++) cur[jb] = nxt[jb];
        }
    }
    if (LAST) {
        const int half = 1 << (R - 1);
        const typename F::twt w = f.prep(cur[0]);
#pragma unroll
        for (int i = 0; i < half; i++) f.gs_last(x[i], x[half + i], w, n_inv);
    }
#else
#pragma unroll
    for (int s = R - 1; s >= (LAST ? 1 : 0); s--) {
        const int half = 1 << (R - 1 - s);
#pragma unroll
        for (int jb = 0; jb < (1 << s); jb++) {
            const typename F::twt w = f.prep(tw(s, jb));
#pragma unroll
            for (int i = 0; i < half; i++) f.gs(x[jb * 2 * half + i], x[jb * 2 * half + half + i], w);
        }
    }
    if (LAST) {
        const int half = 1 << (R - 1);
        const typename F::twt w = f.prep(tw(0, 0));
#pragma unroll
        for (int i = 0; i < half; i++) f.gs_last(x[i], x[half + i], w, n_inv);
    }
#endif
}

// LDS images: padded (36 KiB, default) or XOR-swizzled (32 KiB) against bank conflicts.
// Column tile [row][C cols]: with C = 16 a row is half a 256-B bank row; the second pass's half-waves are 2^RB
// rows apart and must hit both halves (padded: every 2^RB-row group shifted by one row).
template <int C, int RB>
__device__ __forceinline__ uint32_t lds_col(uint32_t r, uint32_t c) {
#if PHA_LDS_SWIZZLE
    // rows 2^RB apart (the lanes of the second pass's half-waves) land in different parts of a 256-B bank row
    constexpr uint32_t mask = C < 32 ? 32 / C - 1 : 0;      // rows per 256-B bank row, minus one
    return (r ^ ((r >> RB) & mask)) * C + c;
#else
    return r * C + c + (C == 16 ? (r >> RB) * 16 : 0);
#endif
}
// The same index for row (hi << RB) | lo with the part that depends on the (compile-time) register index split off as a
// constant: lds_col<C, RB>((j << RB) | r_lo, c) = lds_col_lo(r_lo, c) + j * kColHiStep, and
// lds_col<C, RB>((r_hi << RB) | j, c) = lds_col_hi(r_hi, c) + j * C -- the 16 accesses of a pass then share one address
// register and differ in their immediate offsets (the padded image only; the swizzled one has no such form)
template <int C, int RB>
struct LdsColSplit {
    static constexpr bool kSplit = !PHA_LDS_SWIZZLE;
    static constexpr uint32_t kHiStep = (static_cast<uint32_t>(C) << RB) + (C == 16 ? 16u : 0u);
    __device__ __forceinline__ static uint32_t by_hi(uint32_t j, uint32_t r_lo, uint32_t c) {      // row (j << RB) | r_lo
        if (kSplit) return r_lo * C + c + j * kHiStep;
        return lds_col<C, RB>((j << RB) | r_lo, c);
    }
    __device__ __forceinline__ static uint32_t by_lo(uint32_t r_hi, uint32_t j, uint32_t c) {      // row (r_hi << RB) | j
        if (kSplit) return r_hi * kHiStep + c + j * C;
        return lds_col<C, RB>((r_hi << RB) | j, c);
    }
};
// Row tile, seen as 128-B rows of eight 16-B chunks.  Padded: 16 B per 128 B so that the last pass's ds_read_b128
// (lane stride 128 B) is conflict-free.  Swizzled: chunk ^= (row >> 1) & 7 spreads the 16 lanes of such a read
// over the 16 chunks of a bank row, and row ^= (row >> 4) & 1 separates the first pass's half-waves (16 rows apart).
__device__ __forceinline__ uint32_t lds_row(uint32_t e) {
#if PHA_LDS_SWIZZLE
    const uint32_t row = e >> 4;
    return ((row ^ ((row >> 4) & 1u)) << 4) | (((e >> 1) ^ (row >> 1)) & 7u) << 1 | (e & 1u);
#else
    return e + ((e >> 4) << 1);
#endif
}

// lds_row(x + y) with y a compile-time constant whose sum with x carries nothing out of the low four bits: the padded
// index is then additive, lds_row(x) + lds_row(y), and the constant part becomes the immediate offset of the LDS access
// (16 accesses share one address register).  The swizzled image has no such form.
__device__ __forceinline__ uint32_t lds_row_add(uint32_t x_index, uint32_t x, uint32_t y) {
#if PHA_LDS_SWIZZLE
    (void)x_index;
    return lds_row(x + y);
#else
    (void)x;
    return x_index + (y + ((y >> 4) << 1));
#endif
}
// Between the two kernels of a transform the buffer holds the field's raw lazy representation
// (an unreduced integer, or the bits of a double on the FP64 path); only the first kernel's
// input and the last kernel's output are residues in the caller's format.
// Geometry of a column tile: the 2^split subtrees of a limb are independent column transforms on n >> split
// consecutive coefficients, rooted at twiddle node 2^split + h.
struct SubTree {
    uint32_t offset, tile, n2, root;
};
template <int S1>
__device__ __forceinline__ SubTree sub_tree(const NttTables &T, uint32_t ti) {
    const uint32_t sub_log = T.log_n - T.split, n_sub = 1u << sub_log, tiles_log = sub_log - kTileLog;
    const uint32_t h = ti >> tiles_log;
    return SubTree{h << sub_log, ti & ((1u << tiles_log) - 1), n_sub >> S1, (1u << T.split) + h};
}

// where limb twr of poly `poly` lives in the output buffer of an inverse transform (see LimbMap::out_limb0)
__device__ __forceinline__ size_t inv_out_offset(const LimbMap &lm, uint32_t poly, size_t poly_stride, uint32_t twr, uint32_t n) {
    return poly * (lm.out_poly_stride ? lm.out_poly_stride : poly_stride) + static_cast<size_t>(twr - lm.out_limb0) * n;
}

__device__ __forceinline__ bool use_fp64(const NttTables &T, const DModulus &mod) {
    return T.fp64 && (mod.value >> 50) == 0;
}

// ---------------------------------------------------------------------------------------------
// Every kernel below issues the 16 global loads per lane of its tile FIRST and only then fetches what depends on the
// limb (modulus, field choice, twiddles): per-workgroup timestamps (tools/scratch/trace_ntt.py) showed 40 % of a
// launch's workgroups entering their load phase 2-4 us late, queued behind the other workgroups' tile loads for
// the few bytes of T.mod[tidx] that selected their code path.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// forward, column phase: stages m = 1 .. 2^(S1-1)
// ---------------------------------------------------------------------------------------------
template <int S1, int RA>
__device__ __forceinline__ void fntt_col_load(uint64_t (&raw)[16], const uint64_t *base, uint32_t n2) {
    constexpr int RB = S1 - RA, C = kTile >> S1, GA = 16 >> RA;
#pragma unroll
    for (int g = 0; g < GA; g++) {
        const uint32_t gid = pha_tid() + kThreads * g, c = gid % C, r_lo = gid / C;
        // uniform row pointer + one 32-bit lane offset for all 2^RA loads (scalar base, no 64-bit VALU address arithmetic)
        const uint32_t off = (r_lo * n2 + c) * 8u;
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
            if (PHA_BUF_MASK & 1)
                raw[(g << RA) + j] = ld_tile(rs, off, static_cast<uint32_t>(j << RB) * n2 * 8u);
            else
                raw[(g << RA) + j] = base[static_cast<size_t>((j << RB) | r_lo) * n2 + c];
        }
    }
}
// SC1: the result is stored write-through (hand-off to the row pass inside one launch, ntt_fused.cuh)
template <int S1, int RA, bool SC1 = false, class F>
__device__ __forceinline__ void fntt_col_body(uint64_t *lds, const uint64_t (&raw)[16], uint64_t *base, const tw_t *tw_raw,
                                              uint32_t n2, uint32_t root, const F &f) {
    constexpr int RB = S1 - RA, C = kTile >> S1, GA = 16 >> RA, GB = 16 >> RB;
    using TW = typename F::twl;
    const TW *tw = reinterpret_cast<const TW *>(tw_raw);
    const uint32_t t = pha_tid();
    PHA_TRACE_MARK(0);
#pragma unroll
    for (int g = 0; g < GA; g++) {
        const uint32_t gid = t + kThreads * g, c = gid % C, r_lo = gid / C;
        typename F::elem x[1 << RA];
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) x[j] = f.from_u64(raw[(g << RA) + j]);
        ct_radix<RA, 0>(x, CanonTw<TW>{tw, root}, f);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) lds[LdsColSplit<C, RB>::by_hi(j, r_lo, c)] = F::raw(x[j]);
    }
    PHA_TRACE_MARK(1);
    // one group per lane and 8-byte twiddles: the second pass's 2^RB - 1 twiddles are requested before the barrier
    constexpr bool kPre = PHA_TW_PREFETCH && GB == 1 && (sizeof(TW) == 8 || PHA_TW_PREFETCH >= 3);
    PreTw<kPre ? RB : 1, TW> pre;
    if constexpr (kPre) pre = prefetch_tw<RB, TW>(tw, (root << RA) + t / C);
    __syncthreads();
    prio_late();
    PHA_TRACE_MARK(2);
#pragma unroll
    for (int g = 0; g < GB; g++) {
        const uint32_t gid = t + kThreads * g, c = gid % C, r_hi = gid / C;
        typename F::elem x[1 << RB];
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) x[j] = F::unraw(lds[LdsColSplit<C, RB>::by_lo(r_hi, j, c)]);
        if constexpr (kPre)
            ct_radix<RB, RA>(x, pre, f);
        else
            ct_radix<RB, RA>(x, CanonTw<TW>{tw, (root << RA) + r_hi}, f);
        const uint32_t off = ((r_hi << RB) * n2 + c) * 8u;
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base);
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) {
            if (SC1)
                st_sc1(&base[static_cast<size_t>((r_hi << RB) | j) * n2 + c], F::raw(x[j]));
            else if (PHA_BUF_MASK & 1)
                st_tile(rs, off, static_cast<uint32_t>(j) * n2 * 8u, F::raw(x[j]));
            else
                st_stream(&base[static_cast<size_t>((r_hi << RB) | j) * n2 + c], F::raw(x[j]));
        }
    }
    PHA_TRACE_MARK(3);
#if defined(PHA_TRACE) && PHA_TRACE != 3
    __builtin_amdgcn_s_waitcnt(0);      // stores acknowledged (keeps the workgroup until they are)
    PHA_TRACE_MARK(4);
#endif
}

template <int S1, int RA, bool NARROW, bool BCAST = false>
__global__ PHA_NTT_BOUNDS void fntt_col_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm) {
    constexpr int C = kTile >> S1;
    __shared__ uint64_t lds[kLdsWords];
    const uint32_t n = 1u << T.log_n, tiles = n >> kTileLog;
    PHA_TRACE_ENTRY();
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    const SubTree st = sub_tree<S1>(T, ti);
    uint64_t *base = data + poly * poly_stride + static_cast<size_t>(twr) * n + st.offset + st.tile * C;
    uint64_t raw[16];
    fntt_col_load<S1, RA>(raw, BCAST ? lm.bcast_src + poly * lm.bcast_poly_stride + st.offset + st.tile * C : base, st.n2);
    prio_entry();
    const DModulus mod = T.mod[tidx];
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    if (BCAST) {
#pragma unroll
        for (int j = 0; j < 16; j++) raw[j] = barrett64(raw[j], mod.value, mod.ratio1);
    }
    if (use_fp64(T, mod))
        fntt_col_body<S1, RA>(lds, raw, base, tw, st.n2, st.root, FpField(mod));
    else
        fntt_col_body<S1, RA>(lds, raw, base, tw, st.n2, st.root, IntField<NARROW>(mod));
    PHA_TRACE_EXIT();
}

// Experiment (PHA_COL8=1 at context creation; profiles/EXPERIMENTS.md 3.9): the same column tile with EIGHT
// coefficients per lane -- 512 threads, three register passes of 3 + 3 + 2 stages, two LDS exchanges -- so that a
// lane needs 64-76 VGPRs and a SIMD holds six to eight waves instead of four.  After the first exchange a wave owns
// 32 consecutive rows of the tile, which the remaining five stages never leave: the second exchange is wave-local.
// Stage for stage the butterflies are those of fntt_col_kernel<8, 4>, so the raw output is bit-identical.
// Measured at 72 limbs of N = 2^16 (profiles/r3o_col8_ab.log): FP64 limbs 16.2-16.3 us against 16.4-16.5 us, 60-bit
// limbs 20.5 us (six waves, no spills) or 22.6 us (eight waves, 4-6 spilled VGPRs) against 22.0 us; the headline and
// HMult do not move.  Twice the waves on the same four workgroups per CU buy nothing: the pass is not latency-bound.
#ifndef PHA_COL8_WAVES
#define PHA_COL8_WAVES 6
#endif
constexpr int kThreads8 = kTile / 8;
__device__ __forceinline__ uint32_t lds_col8(uint32_t r, uint32_t c) {
    return r * 16 + c + (r >> 3) * 16;   // rows 8 apart (the third pass's quarter-waves) fall into different bank halves
}
template <class F>
__device__ __forceinline__ void fntt_col8_body(uint64_t *lds, const uint64_t (&raw)[8], uint64_t *base, const tw_t *tw_raw,
                                               uint32_t n2, uint32_t root, const F &f) {
    using TW = typename F::twl;
    const TW *tw = reinterpret_cast<const TW *>(tw_raw);
    const uint32_t t = pha_tid(), c = t & 15u;
    typename F::elem x[8];
    {
        const uint32_t r_lo = t >> 4;
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = f.from_u64(raw[j]);
        ct_radix<3, 0>(x, CanonTw<TW>{tw, root}, f);
#pragma unroll
        for (int j = 0; j < 8; j++) lds[lds_col8((j << 5) | r_lo, c)] = F::raw(x[j]);
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6), q = (t >> 4) & 3u;
    __syncthreads();
    {
        const uint32_t r0 = (wave << 5) | q;
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = F::unraw(lds[lds_col8(r0 + (j << 2), c)]);
        ct_radix<3, 3>(x, CanonTw<TW>{tw, (root << 3) + wave}, f);
#pragma unroll
        for (int j = 0; j < 8; j++) lds[lds_col8(r0 + (j << 2), c)] = F::raw(x[j]);
    }
    // rows [32 wave, 32 wave + 32) belong to this wave alone from here on: its own LDS operations complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const uint32_t r0 = (wave << 5) | (q << 3);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = F::unraw(lds[lds_col8(r0 + j, c)]);
        typename F::elem lo[4] = {x[0], x[1], x[2], x[3]}, hi[4] = {x[4], x[5], x[6], x[7]};
        const uint32_t node = (root << 6) + (r0 >> 2);
        ct_radix<2, 6>(lo, CanonTw<TW>{tw, node}, f);
        ct_radix<2, 6>(hi, CanonTw<TW>{tw, node + 1}, f);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            st_stream(&base[static_cast<size_t>(r0 + j) * n2 + c], F::raw(lo[j]));
            st_stream(&base[static_cast<size_t>(r0 + 4 + j) * n2 + c], F::raw(hi[j]));
        }
    }
}
template <bool NARROW>
__global__ __launch_bounds__(kThreads8) __attribute__((amdgpu_waves_per_eu(PHA_COL8_WAVES, 8))) void fntt_col8_kernel(
    uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm) {
    __shared__ uint64_t lds[kTile + 512];
    const uint32_t n = 1u << T.log_n, tiles = n >> kTileLog;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    uint32_t twr, tidx;
    if (!resolve_limb(lm, bw.poly, bw.li, twr, tidx)) return;
    const SubTree st = sub_tree<8>(T, bw.ti);
    uint64_t *base = data + bw.poly * poly_stride + static_cast<size_t>(twr) * n + st.offset + st.tile * 16;
    uint64_t raw[8];
    {
        const uint32_t t = pha_tid(), c = t & 15u, r_lo = t >> 4;
#pragma unroll
        for (int j = 0; j < 8; j++) raw[j] = base[static_cast<size_t>((j << 5) | r_lo) * st.n2 + c];
    }
    const DModulus mod = T.mod[tidx];
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    if (use_fp64(T, mod))
        fntt_col8_body(lds, raw, base, tw, st.n2, st.root, FpField(mod));
    else
        fntt_col8_body(lds, raw, base, tw, st.n2, st.root, IntField<NARROW>(mod));
}

// ---------------------------------------------------------------------------------------------
// forward, row phase: stages m = 2^S1 .. N/2, final reduction to [0,q), optional fused epilogue
// ---------------------------------------------------------------------------------------------
template <int S2, int RA>
__device__ __forceinline__ void fntt_row_load(uint64_t (&raw)[16], const uint64_t *base) {
    constexpr int RB = S2 - RA, L = 1 << S2, GC = 16 >> RA;
#pragma unroll
    for (int g = 0; g < GC; g++) {
        const uint32_t gid = pha_tid() + kThreads * g, b_lo = gid & ((1u << RB) - 1), a = gid >> RB;
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) raw[(g << RA) + j] = base[a * L + ((j << RB) | b_lo)];
    }
}
template <int S2, int RA, int EPI, class F>
__device__ __forceinline__ void fntt_row_body(uint64_t *lds, const uint64_t (&raw)[16], const tw_t *tw_raw,
                                              const tw_t *twd_raw, uint32_t n1, uint32_t ti, const F &f, uint64_t q,
                                              uint64_t *out, const uint64_t *sub, const uint64_t *addend, uint64_t cst,
                                              uint64_t cst_shoup, const double *twg = nullptr, double *lds_omega = nullptr) {
    constexpr int RB = S2 - RA, L = 1 << S2, RW = kTile >> S2, GC = 16 >> RA, GD = 16 >> RB;
    constexpr bool kCanGen = std::is_same<F, FpField>::value && RB == 4 && kTileLog == 12;
    using TW = typename F::twl;
    const TW *tw = reinterpret_cast<const TW *>(tw_raw);
    const uint32_t t = pha_tid();
    double theta = 0.0;
    if (kCanGen && twg) theta = gen_theta(f, twg, lds_omega, ti, t);      // omega reaches LDS before the first barrier
    // one group per lane and 8-byte twiddles: the first pass's 2^RA - 1 twiddles are requested here, behind the tile loads
    // and before their data are touched, instead of stage by stage with a wait behind each load
    constexpr bool kPreA = PHA_TW_PREFETCH >= 2 && GC == 1 && (sizeof(TW) == 8 || PHA_TW_PREFETCH >= 3);
    PreTw<kPreA ? RA : 1, TW> preA;
    if constexpr (kPreA) preA = prefetch_tw<RA, TW>(tw, n1 + ti * RW + (t >> RB));
#pragma unroll
    for (int g = 0; g < GC; g++) {
        const uint32_t gid = t + kThreads * g, b_lo = gid & ((1u << RB) - 1), a = gid >> RB;
        typename F::elem x[1 << RA];
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) x[j] = F::unraw(raw[(g << RA) + j]);
        if constexpr (kPreA)
            ct_radix<RA, 0>(x, preA, f);
        else
            ct_radix<RA, 0>(x, CanonTw<TW>{tw, n1 + ti * RW + a}, f);
        const uint32_t ia = lds_row(a * L + b_lo);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) lds[lds_row_add(ia, a * L + b_lo, j << RB)] = F::raw(x[j]);
    }
    __syncthreads();
    prio_late();
#pragma unroll
    for (int g = 0; g < GD; g++) {
        const uint32_t gid = t + kThreads * g, e0 = gid << RB, i0 = lds_row(e0);
        const uint32_t ggid = ti * (kTile >> RB) + gid;
        typename F::elem x[1 << RB];
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) x[j] = F::unraw(lds[lds_row_add(i0, e0, j)]);
        if constexpr (kCanGen) {
            if (twg) {
                ct_radix<RB, RA>(x, GenTw(f, lds_omega, theta), f);
            } else {
                const TW *twd = reinterpret_cast<const TW *>(twd_raw) +
                                static_cast<size_t>(ggid >> 6) * (((1 << RB) - 1) * 64) + (ggid & 63);
                ct_radix<RB, RA>(x, LastPassTw<RB, TW>{twd}, f);
            }
        } else {
            const TW *twd = reinterpret_cast<const TW *>(twd_raw) +
                            static_cast<size_t>(ggid >> 6) * (((1 << RB) - 1) * 64) + (ggid & 63);
            ct_radix<RB, RA>(x, LastPassTw<RB, TW>{twd}, f);
        }
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) lds[lds_row_add(i0, e0, j)] = F::raw(f.canon_fwd(x[j]));
    }
#if PHA_EPI_PREFETCH
    // The epilogue's operand streams are requested here, where the butterfly registers are dead and the workgroup is
    // about to wait at the barrier, instead of after it (measured: profiles/r2c_ab_epilogue_prefetch.log)
    ulonglong2 pre_s[8], pre_a[8];
    if (EPI != EPI_NONE) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t e = (i * kThreads + t) * 2;
            pre_s[i] = *reinterpret_cast<const ulonglong2 *>(sub + e);
            if (PHA_EPI_PREFETCH >= 2 && EPI == EPI_SUB_MUL_ADD && addend != nullptr)
                pre_a[i] = *reinterpret_cast<const ulonglong2 *>(addend + e);
        }
    }
#endif
    __syncthreads();

    const typename F::twt c = f.make_cst(cst, cst_shoup);
    // all eight LDS reads first, so that their latency is paid once and not per result store
    typename F::elem lv0[8], lv1[8];
    const __amdgpu_buffer_rsrc_t out_rs = tile_rsrc(out);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t it = lds_row(2 * t);
        lv0[i] = F::unraw(lds[lds_row_add(it, 2 * t, i * kThreads * 2)]);
        lv1[i] = F::unraw(lds[lds_row_add(it, 2 * t, i * kThreads * 2 + 1)]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t e = (i * kThreads + t) * 2;
        const typename F::elem v0 = lv0[i], v1 = lv1[i];
        ulonglong2 v;
        if (EPI != EPI_NONE) {
            // (cx - NTT(delta)) * PInv mod q  (src/ntt/ntt_moddown.cu:209-214,
            // include/uintmodmath.cuh:234-242); same shape for rescale (src/rns.cu:1138-1155)
            // (fetching these operands a pass ahead was measured: no gain, and 64 more VGPRs)
#if PHA_EPI_PREFETCH
            const ulonglong2 s = pre_s[i];
#else
            const ulonglong2 s = *reinterpret_cast<const ulonglong2 *>(sub + e);
#endif
            v.x = f.to_u64(f.sub_mul(v0, s.x, c));
            v.y = f.to_u64(f.sub_mul(v1, s.y, c));
            if (EPI == EPI_SUB_MUL_ADD && addend != nullptr) {
#if PHA_EPI_PREFETCH >= 2
                const ulonglong2 a = pre_a[i];
#else
                const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(addend + e);
#endif
                v.x = csub(v.x + a.x, q);
                v.y = csub(v.y + a.y, q);
            }
        } else {
            v.x = f.to_u64(v0);
            v.y = f.to_u64(v1);
        }
        if (PHA_BUF_MASK & 2)
            // the whole offset in the lane operand, the scalar offset an immediate 0: a 16-byte store whose scalar offset is
            // a REGISTER gets no wait state from the compiler before its data registers are overwritten, and the hardware
            // needs one (measured: sparse wrong words in the epilogue kernels; the asm form carries its own s_nop)
            st_tile2(out_rs, t * 16u + i * kThreads * 16u, 0, v);
        else
            st_stream2(out + e, v);
    }
}

template <int S2, int RA, int EPI, bool NARROW>
__global__ PHA_NTT_BOUNDS void fntt_row_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm,
                                                            Epilogue ep) {
    __shared__ uint64_t lds[kLdsWords];
    __shared__ double lds_omega[16];
    const uint32_t n = 1u << T.log_n, n1 = n >> S2, tiles = n >> kTileLog;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    const double *twg = T.twG ? T.twG + static_cast<size_t>(tidx) * kGenWords : nullptr;
    const size_t limb_off = static_cast<size_t>(twr) * n + static_cast<size_t>(ti) * kTile;
    uint64_t *base = data + poly * poly_stride + limb_off;
    uint64_t raw[16];
    fntt_row_load<S2, RA>(raw, base);
    prio_entry();
    const DModulus mod = T.mod[tidx];
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    const tw_t *twd = T.twD + static_cast<size_t>(tidx) * T.twD_stride;

    uint64_t *out = base;
    const uint64_t *sub = nullptr, *addend = nullptr;
    uint64_t cst = 0, cst_shoup = 0;
    if (EPI != EPI_NONE) {
        out = ep.out + poly * ep.out_poly_stride + limb_off;
        sub = ep.sub + poly * ep.sub_poly_stride + limb_off;
        if (EPI == EPI_SUB_MUL_ADD && (ep.addend_every == 0 || poly % ep.addend_every == 0))
            addend = ep.addend + poly * ep.out_poly_stride + limb_off;
        cst = ep.cst[twr - ep.cst_limb0];
        cst_shoup = ep.cst_shoup[twr - ep.cst_limb0];
    }
    if (use_fp64(T, mod))
        fntt_row_body<S2, RA, EPI>(lds, raw, tw, twd, n1, ti, FpField(mod), mod.value, out, sub, addend, cst, cst_shoup, twg,
                                   lds_omega);
    else
        fntt_row_body<S2, RA, EPI>(lds, raw, tw, twd, n1, ti, IntField<NARROW>(mod), mod.value, out, sub, addend, cst,
                                   cst_shoup);
}

// ---------------------------------------------------------------------------------------------
// inverse, row phase (first): stages m = N/2 .. 2^S1
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void intt_row_load(uint64_t (&raw)[16], const uint64_t *src) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + (i * kThreads + threadIdx.x) * 2);
        raw[2 * i] = v.x;
        raw[2 * i + 1] = v.y;
    }
}
template <int S2, int RA, class F>
__device__ __forceinline__ void intt_row_body(uint64_t *lds, const uint64_t (&raw)[16], uint64_t *dst, const tw_t *tw_raw,
                                              const tw_t *twd_raw, uint32_t n1, uint32_t ti, const F &f,
                                              const double *twg = nullptr, double *lds_omega = nullptr) {
    constexpr int RB = S2 - RA, L = 1 << S2, RW = kTile >> S2, GC = 16 >> RA, GD = 16 >> RB;
    constexpr bool kCanGen = std::is_same<F, FpField>::value && RB == 4 && kTileLog == 12;
    using TW = typename F::twl;
    const TW *tw = reinterpret_cast<const TW *>(tw_raw);
    const uint32_t t = threadIdx.x;
    const typename F::twt unused{};
    const __amdgpu_buffer_rsrc_t dst_rs = tile_rsrc(dst);
    double theta = 0.0;
    if (kCanGen && twg) theta = gen_theta(f, twg, lds_omega, ti, t);      // omega reaches LDS before the first barrier
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t it = lds_row(2 * t);
        lds[lds_row_add(it, 2 * t, i * kThreads * 2)] = F::raw(f.from_u64(raw[2 * i]));
        lds[lds_row_add(it, 2 * t, i * kThreads * 2 + 1)] = F::raw(f.from_u64(raw[2 * i + 1]));
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GD; g++) {
        const uint32_t gid = t + kThreads * g, e0 = gid << RB, i0 = lds_row(e0);
        const uint32_t ggid = ti * (kTile >> RB) + gid;
        typename F::elem x[1 << RB];
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) x[j] = F::unraw(lds[lds_row_add(i0, e0, j)]);
        if constexpr (kCanGen) {
            if (twg) {
                gs_radix<RB, false>(x, GenTw(f, lds_omega, theta), f, unused);
            } else {
                const TW *twd = reinterpret_cast<const TW *>(twd_raw) +
                                static_cast<size_t>(ggid >> 6) * (((1 << RB) - 1) * 64) + (ggid & 63);
                gs_radix<RB, false>(x, LastPassTw<RB, TW>{twd}, f, unused);
            }
        } else {
            const TW *twd = reinterpret_cast<const TW *>(twd_raw) +
                            static_cast<size_t>(ggid >> 6) * (((1 << RB) - 1) * 64) + (ggid & 63);
            gs_radix<RB, false>(x, LastPassTw<RB, TW>{twd}, f, unused);
        }
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) lds[lds_row_add(i0, e0, j)] = F::raw(x[j]);
    }
    constexpr bool kPre = PHA_TW_PREFETCH && GC == 1 && (sizeof(TW) == 8 || PHA_TW_PREFETCH >= 3);
    PreTw<kPre ? RA : 1, TW> pre;
    if constexpr (kPre) pre = prefetch_tw<RA, TW>(tw, n1 + ti * RW + (t >> RB));
    __syncthreads();
    prio_late();
#pragma unroll
    for (int g = 0; g < GC; g++) {
        const uint32_t gid = t + kThreads * g, b_lo = gid & ((1u << RB) - 1), a = gid >> RB;
        typename F::elem x[1 << RA];
        const uint32_t ia = lds_row(a * L + b_lo);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) x[j] = F::unraw(lds[lds_row_add(ia, a * L + b_lo, j << RB)]);
        if constexpr (kPre)
            gs_radix<RA, false>(x, pre, f, unused);
        else
            gs_radix<RA, false>(x, CanonTw<TW>{tw, n1 + ti * RW + a}, f, unused);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
            if (PHA_BUF_MASK & 4)
                st_tile(dst_rs, (a * L + b_lo) * 8u, (j << RB) * 8u, F::raw(x[j]));
            else
                st_stream(&dst[a * L + ((j << RB) | b_lo)], F::raw(x[j]));
        }
    }
}

template <int S2, int RA>
__global__ PHA_NTT_BOUNDS void intt_row_kernel(uint64_t *out, const uint64_t *in, size_t poly_stride,
                                                            NttTables T, LimbMap lm) {
    __shared__ uint64_t lds[kLdsWords];
    __shared__ double lds_omega[16];
    const uint32_t n = 1u << T.log_n, n1 = n >> S2, tiles = n >> kTileLog;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    const double *twg = T.twG ? T.twG + static_cast<size_t>(tidx) * kGenWords : nullptr;
    const size_t off = poly * poly_stride + static_cast<size_t>(twr) * n + static_cast<size_t>(ti) * kTile;
    const size_t out_off = inv_out_offset(lm, poly, poly_stride, twr, n) + static_cast<size_t>(ti) * kTile;
    uint64_t raw[16];
    intt_row_load(raw, in + off);
    prio_entry();
    const DModulus mod = T.mod[tidx];
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    const tw_t *twd = T.twD + static_cast<size_t>(tidx) * T.twD_stride;
    if (use_fp64(T, mod))
        intt_row_body<S2, RA>(lds, raw, out + out_off, tw, twd, n1, ti, FpField(mod), twg, lds_omega);
    else
        intt_row_body<S2, RA>(lds, raw, out + out_off, tw, twd, n1, ti, IntField<true>(mod));
}

// ---------------------------------------------------------------------------------------------
// inverse, column phase (last): stages m = 2^(S1-1) .. 1, N^-1, reduction to [0,q), optional scale
// ---------------------------------------------------------------------------------------------
template <int S1, int RA>
__device__ __forceinline__ void intt_col_load(uint64_t (&raw)[16], const uint64_t *base, uint32_t n2) {
    constexpr int RB = S1 - RA, C = kTile >> S1, GB = 16 >> RB;
#pragma unroll
    for (int g = 0; g < GB; g++) {
        const uint32_t gid = threadIdx.x + kThreads * g, c = gid % C, r_hi = gid / C;
        const uint32_t off = ((r_hi << RB) * n2 + c) * 8u;
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(base);
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) {
            if (PHA_BUF_MASK & 8)
                raw[(g << RB) + j] = ld_tile(rs, off, static_cast<uint32_t>(j) * n2 * 8u);
            else
                raw[(g << RB) + j] = base[static_cast<size_t>((r_hi << RB) | j) * n2 + c];
        }
    }
}
template <int S1, int RA, int EPI, bool SPLIT, class F>
__device__ __forceinline__ void intt_col_body(uint64_t *lds, const uint64_t (&raw)[16], uint64_t *base, const tw_t *tw_raw,
                                              uint32_t n2, uint32_t root, const F &f, uint64_t n_inv, uint64_t n_inv_shoup,
                                              uint64_t cst, uint64_t cst_shoup) {
    constexpr int RB = S1 - RA, C = kTile >> S1, GA = 16 >> RA, GB = 16 >> RB;
    using TW = typename F::twl;
    const TW *tw = reinterpret_cast<const TW *>(tw_raw);
    const uint32_t t = threadIdx.x;
    const __amdgpu_buffer_rsrc_t base_rs = tile_rsrc(base);
    const typename F::twt ninv = f.make_cst(n_inv, n_inv_shoup);
    constexpr bool kPreB = PHA_TW_PREFETCH >= 2 && GB == 1 && (sizeof(TW) == 8 || PHA_TW_PREFETCH >= 3 || PHA_TW_PREFETCH_INV16);      // as in fntt_row_body
    PreTw<kPreB ? RB : 1, TW> preB;
    if constexpr (kPreB) preB = prefetch_tw<RB, TW>(tw, (root << RA) + t / C);
#pragma unroll
    for (int g = 0; g < GB; g++) {
        const uint32_t gid = t + kThreads * g, c = gid % C, r_hi = gid / C;
        typename F::elem x[1 << RB];
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) x[j] = F::unraw(raw[(g << RB) + j]);
        if constexpr (kPreB)
            gs_radix<RB, false>(x, preB, f, ninv);
        else
            gs_radix<RB, false>(x, CanonTw<TW>{tw, (root << RA) + r_hi}, f, ninv);
#pragma unroll
        for (int j = 0; j < (1 << RB); j++) lds[LdsColSplit<C, RB>::by_lo(r_hi, j, c)] = F::raw(x[j]);
    }
    __syncthreads();
    prio_late();
    const typename F::twt c_scale = f.make_cst(cst, cst_shoup);
#pragma unroll
    for (int g = 0; g < GA; g++) {
        const uint32_t gid = t + kThreads * g, c = gid % C, r_lo = gid / C;
        typename F::elem x[1 << RA];
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) x[j] = F::unraw(lds[LdsColSplit<C, RB>::by_hi(j, r_lo, c)]);
        if (SPLIT)      // a subtree: its last stage is an ordinary one; N^-1 waits for split_inv_kernel
            gs_radix<RA, false>(x, CanonTw<TW>{tw, root}, f, ninv);
        else
            gs_radix<RA, true>(x, CanonTw<TW>{tw, 1u}, f, ninv);
#pragma unroll
        for (int j = 0; j < (1 << RA); j++) {
            typename F::elem v = SPLIT ? f.canon_inv(x[j]) : x[j];
            // fused bconv phase 1 (src/ntt/ntt_modup.cu:212-318)
            if (EPI == EPI_SCALE) v = f.scale(v, c_scale);
            if (PHA_BUF_MASK & 8)
                st_tile(base_rs, (r_lo * n2 + c) * 8u, static_cast<uint32_t>(j << RB) * n2 * 8u, f.to_u64(v));
            else
                st_stream(&base[static_cast<size_t>((j << RB) | r_lo) * n2 + c], f.to_u64(v));
        }
    }
}

template <int S1, int RA, int EPI, bool SPLIT = false>
__global__ PHA_NTT_BOUNDS void intt_col_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm,
                                                            Epilogue ep) {
    constexpr int C = kTile >> S1;
    __shared__ uint64_t lds[kLdsWords];
    const uint32_t n = 1u << T.log_n, tiles = n >> kTileLog;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    const SubTree st = sub_tree<S1>(T, ti);
    uint64_t *base = data + inv_out_offset(lm, poly, poly_stride, twr, n) + st.offset + st.tile * C;
    uint64_t raw[16];
    intt_col_load<S1, RA>(raw, base, st.n2);
    prio_entry();
    const DModulus mod = T.mod[tidx];
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    const uint64_t n_inv = T.n_inv[tidx], n_inv_shoup = T.n_inv_shoup[tidx];
    uint64_t cst = 0, cst_shoup = 0;
    if (EPI == EPI_SCALE) {
        cst = ep.cst[twr - ep.cst_limb0];
        cst_shoup = ep.cst_shoup[twr - ep.cst_limb0];
    }
    if (use_fp64(T, mod))
        intt_col_body<S1, RA, EPI, SPLIT>(lds, raw, base, tw, st.n2, st.root, FpField(mod), n_inv, n_inv_shoup, cst, cst_shoup);
    else
        intt_col_body<S1, RA, EPI, SPLIT>(lds, raw, base, tw, st.n2, st.root, IntField<true>(mod), n_inv, n_inv_shoup, cst,
                                          cst_shoup);
}

// ---------------------------------------------------------------------------------------------
// N = 2^11 (the smallest SAMPLE_SIZE case of the reference, include/ntt.cuh:16-24; no CKKS parameter set of its tests
// uses it): the polynomial is smaller than a tile, so one workgroup transforms a whole limb inside LDS, one stage
// per barrier, on the integer path (such a context keeps every table in the {w, Shoup} form).  Same stage order,
// twiddle indexing (tw[m + k]) and epilogues as the tiled kernels; speed is not a goal here.
// ---------------------------------------------------------------------------------------------
constexpr int kSmallN = 2048;

template <int EPI>
__global__ __launch_bounds__(256) void small_fntt_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm, Epilogue ep) {
    __shared__ uint64_t lds[kSmallN];
    const uint32_t n = 1u << T.log_n, poly = blockIdx.x % lm.polys, li = blockIdx.x / lm.polys, t0 = threadIdx.x;
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, li, twr, tidx)) return;
    const DModulus mod = T.mod[tidx];
    const IntField<false> f(mod);
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    const size_t limb_off = static_cast<size_t>(twr) * n;
    uint64_t *limb = data + poly * poly_stride + limb_off;
    for (uint32_t i = t0; i < n; i += 256) lds[i] = limb[i];
    uint32_t log_t = T.log_n - 1;
    for (uint32_t m = 1; m < n; m <<= 1, log_t--) {
        __syncthreads();
        for (uint32_t idx = t0; idx < n / 2; idx += 256) {
            const uint32_t k = idx >> log_t, j = idx & ((1u << log_t) - 1), a = ((2 * k) << log_t) + j, b = a + (1u << log_t);
            uint64_t x = lds[a], y = lds[b];
            f.template ct<0>(x, y, tw[m + k]);
            lds[a] = x;
            lds[b] = y;
        }
    }
    __syncthreads();
    uint64_t *out = limb;
    const uint64_t *sub = nullptr, *addend = nullptr;
    tw_t c{0, 0};
    if (EPI != EPI_NONE) {
        out = ep.out + poly * ep.out_poly_stride + limb_off;
        sub = ep.sub + poly * ep.sub_poly_stride + limb_off;
        if (EPI == EPI_SUB_MUL_ADD && (ep.addend_every == 0 || poly % ep.addend_every == 0))
            addend = ep.addend + poly * ep.out_poly_stride + limb_off;
        c = f.make_cst(ep.cst[twr - ep.cst_limb0], ep.cst_shoup[twr - ep.cst_limb0]);
    }
    for (uint32_t i = t0; i < n; i += 256) {
        uint64_t v = f.canon_fwd(lds[i]);
        if (EPI != EPI_NONE) {
            v = f.sub_mul(v, sub[i], c);
            if (EPI == EPI_SUB_MUL_ADD && addend != nullptr) v = csub(v + addend[i], mod.value);
        }
        out[i] = v;
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void small_intt_kernel(uint64_t *out, const uint64_t *in, size_t poly_stride, NttTables T,
                                                         LimbMap lm, Epilogue ep) {
    __shared__ uint64_t lds[kSmallN];
    const uint32_t n = 1u << T.log_n, poly = blockIdx.x % lm.polys, li = blockIdx.x / lm.polys, t0 = threadIdx.x;
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, li, twr, tidx)) return;
    const DModulus mod = T.mod[tidx];
    const IntField<false> f(mod);
    const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
    const uint64_t *src = in + poly * poly_stride + static_cast<size_t>(twr) * n;
    uint64_t *dst = out + inv_out_offset(lm, poly, poly_stride, twr, n);
    for (uint32_t i = t0; i < n; i += 256) lds[i] = src[i];
    uint32_t log_t = 0;
    for (uint32_t m = n >> 1; m >= 2; m >>= 1, log_t++) {
        __syncthreads();
        for (uint32_t idx = t0; idx < n / 2; idx += 256) {
            const uint32_t k = idx >> log_t, j = idx & ((1u << log_t) - 1), a = ((2 * k) << log_t) + j, b = a + (1u << log_t);
            uint64_t x = lds[a], y = lds[b];
            f.gs(x, y, tw[m + k]);
            lds[a] = x;
            lds[b] = y;
        }
    }
    __syncthreads();
    const tw_t n_inv = f.make_cst(T.n_inv[tidx], T.n_inv_shoup[tidx]);
    tw_t c{0, 0};
    if (EPI == EPI_SCALE) c = f.make_cst(ep.cst[twr - ep.cst_limb0], ep.cst_shoup[twr - ep.cst_limb0]);
    for (uint32_t j = t0; j < n / 2; j += 256) {   // last stage (m = 1) with N^-1, then the optional scale
        uint64_t x = lds[j], y = lds[j + n / 2];
        f.gs_last(x, y, tw[1], n_inv);
        if (EPI == EPI_SCALE) {
            x = f.scale(x, c);
            y = f.scale(y, c);
        }
        dst[j] = x;
        dst[j + n / 2] = y;
    }
}

// ---------------------------------------------------------------------------------------------
// N = 2^17: the outermost radix-2 stage (m = 1, pairs N/2 apart) as an element-wise kernel on canonical
// residues; Barrett products, so the same code serves integer and FP64 limbs.  One workgroup = 4096 pairs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t root_twiddle(const NttTables &T, const DModulus &mod, const tw_t *tw) {
    // tw[1]: FP64 limbs store doubles, integer limbs {w, w_shoup}
    return use_fp64(T, mod) ? static_cast<uint64_t>(reinterpret_cast<const double *>(tw)[1]) : tw[1].x;
}

__global__ __launch_bounds__(256) void split_fwd_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm) {
    const uint32_t n = 1u << T.log_n, half = n >> 1, tiles = half >> 12;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    uint64_t *lo = data + poly * poly_stride + static_cast<size_t>(twr) * n + static_cast<size_t>(ti) * 4096, *hi = lo + half;
    const DModulus mod = T.mod[tidx];
    const uint64_t q = mod.value, w = root_twiddle(T, mod, T.tw + static_cast<size_t>(tidx) * n);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t e = (i * 256 + threadIdx.x) * 2;
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(lo + e), b = *reinterpret_cast<const ulonglong2 *>(hi + e);
        // inputs may be lazy (< 4q) like the reference's; Barrett reduces any 64-bit value
        const uint64_t a0 = barrett64(a.x, q, mod.ratio1), a1 = barrett64(a.y, q, mod.ratio1);
        const uint64_t t0 = mul_barrett(b.x, w, mod), t1 = mul_barrett(b.y, w, mod);
        *reinterpret_cast<ulonglong2 *>(lo + e) = ulonglong2{csub(a0 + t0, q), csub(a1 + t1, q)};
        *reinterpret_cast<ulonglong2 *>(hi + e) = ulonglong2{csub(a0 + q - t0, q), csub(a1 + q - t1, q)};
    }
}

// inverse: x = (a + b) N^-1, y = (a - b) itw[1] (itw[1] already holds N^-1, src/util/ntt.cu:53-55), optional scale
template <int EPI>
__global__ __launch_bounds__(256) void split_inv_kernel(uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm,
                                                        Epilogue ep) {
    const uint32_t n = 1u << T.log_n, half = n >> 1, tiles = half >> 12;
    BlockWork bw;
    if (!decode_block(blockIdx.x, lm, tiles, bw)) return;
    const uint32_t poly = uniform(bw.poly), ti = uniform(bw.ti);
    uint32_t twr, tidx;
    if (!resolve_limb(lm, poly, bw.li, twr, tidx)) return;
    twr = uniform(twr);
    tidx = uniform(tidx);
    uint64_t *lo = data + inv_out_offset(lm, poly, poly_stride, twr, n) + static_cast<size_t>(ti) * 4096, *hi = lo + half;
    const DModulus mod = T.mod[tidx];
    const uint64_t q = mod.value, w = root_twiddle(T, mod, T.tw + static_cast<size_t>(tidx) * n);
    const uint64_t n_inv = T.n_inv[tidx];
    const uint64_t cst = EPI == EPI_SCALE ? ep.cst[twr - ep.cst_limb0] : 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t e = (i * 256 + threadIdx.x) * 2;
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(lo + e), b = *reinterpret_cast<const ulonglong2 *>(hi + e);
        ulonglong2 x{mul_barrett(csub(a.x + b.x, q), n_inv, mod), mul_barrett(csub(a.y + b.y, q), n_inv, mod)};
        ulonglong2 y{mul_barrett(csub(a.x + q - b.x, q), w, mod), mul_barrett(csub(a.y + q - b.y, q), w, mod)};
        if (EPI == EPI_SCALE) {
            x = ulonglong2{mul_barrett(x.x, cst, mod), mul_barrett(x.y, cst, mod)};
            y = ulonglong2{mul_barrett(y.x, cst, mod), mul_barrett(y.y, cst, mod)};
        }
        *reinterpret_cast<ulonglong2 *>(lo + e) = x;
        *reinterpret_cast<ulonglong2 *>(hi + e) = y;
    }
}

}  // namespace pha
