// Single-launch negacyclic NTT / INTT for gfx950: both passes of a transform in ONE persistent launch.
//
// The two-kernel transform (ntt_kernels.cuh) streams every limb through the memory fabric twice and pays a kernel
// boundary plus a second resident round of workgroups that load, compute and store in lock-step.  Here one launch of
// (workgroups per CU) x (CUs) persistent workgroups pulls tile tasks from per-XCD ticket queues:
//   * a "unit" is one (poly, limb); its 16 first-pass tiles (forward: column tiles, inverse: row tiles) and its 16
//     second-pass tiles are consecutive tickets of ONE queue, the first pass `depth` units ahead of the second, so
//     that the second pass of a unit starts on the XCD whose L2 still holds the lines the first pass wrote;
//   * the hand-off is the placement-independent agent-scope protocol (cdna_hip_programming.md Guideline 16): every
//     storing wave drains its stores, the workgroup meets at a barrier, one lane runs an agent-scope release fence and
//     adds to the unit's arrival counter; a second-pass workgroup polls that counter relaxed from one lane, runs ONE
//     agent-scope acquire and a barrier, then loads.  XCD affinity (HW_REG_XCC_ID picks the queue) is a speed matter
//     only: any workgroup may take tickets of any queue (it moves on to the next queue when its own runs dry), and
//     results do not depend on placement, dispatch order or timing;
//   * forward progress: tickets of a queue are claimed in order, the first-pass tiles of a unit come before its
//     second-pass tiles, a first-pass task never waits, and whoever claimed a ticket is resident and running — so a
//     second-pass task only ever waits for workgroups that are executing.  Spins are bounded all the same (the error
//     word is set and the kernel drains);
//   * state: ticket / arrival / exit counters live in a small control block that is zero when a launch starts; the
//     last workgroup to leave zeroes it again (replay-safe under hipGraph, no per-launch memset).
// Results are bit-identical to the two-kernel path: the same bodies run on the same tiles.
#pragma once

#include "ntt_kernels.cuh"

namespace pha {

constexpr uint32_t kFusedQueues = 8;                  // one per XCD
constexpr uint32_t kCtlExit = kFusedQueues * 32;      // ticket of queue q at word q * 32 (own 128-B line)
constexpr uint32_t kCtlError = kCtlExit + 1;
constexpr uint32_t kCtlDone = kCtlExit + 32;          // arrival counters, one per unit
constexpr uint32_t kFusedMaxUnits = 4096;
constexpr uint32_t kCtlWords = kCtlDone + kFusedMaxUnits;
constexpr uint32_t kFusedSpinLimit = 1u << 21;        // x ~0.3 us of s_sleep: well over any launch, then give up

struct FusedArgs {
    uint32_t *ctl;      // kCtlWords words, zero at launch
    uint32_t depth;     // first-pass units claimed ahead of the second pass, per queue
    uint32_t nq;        // queues in use (1..8)
    uint32_t use_xcc;   // 1: home queue = XCC id % nq, 0: blockIdx % nq
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

// ticket -> (unit within the queue, tile, pass).  Order of tile blocks in a queue of U units with look-ahead D:
// first(0..D-1), then second(k), first(D+k) alternating, then the remaining second passes.
__device__ __forceinline__ void fused_decode(uint32_t t, uint32_t U, uint32_t D, uint32_t tiles, uint32_t &unit,
                                             uint32_t &tile, bool &second) {
    const uint32_t b = t / tiles;
    tile = t % tiles;
    D = min(D, U);
    if (b < D) {
        unit = b;
        second = false;
        return;
    }
    const uint32_t i = b - D, alt = 2 * (U - D);
    if (i < alt) {
        second = (i & 1u) == 0;
        unit = second ? (i >> 1) : D + (i >> 1);
    } else {
        second = true;
        unit = (U - D) + (i - alt);
    }
}

__device__ __forceinline__ uint32_t ctl_add(uint32_t *p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// first pass done: publish this workgroup's tile (plain stores -> drain -> barrier -> agent release -> arrival)
__device__ __forceinline__ void fused_publish(uint32_t *done) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fence's own wait must not be dropped (Pitfall 12)
        ctl_add(done, 1u);
    }
}

// second pass: wait until all first-pass tiles of the unit have arrived, then ONE agent acquire for the workgroup
__device__ __forceinline__ void fused_wait(uint32_t *done, uint32_t need, uint32_t *err) {
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > kFusedSpinLimit) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

#ifndef PHA_FUSED_WAVES
#define PHA_FUSED_WAVES 4      // register budget: this many workgroups of a CU may be resident (128 VGPRs)
#endif
#define PHA_FUSED_ATTR __attribute__((amdgpu_waves_per_eu(PHA_FUSED_WAVES, PHA_FUSED_WAVES)))

// One tile task.  The four bodies are separate (not inlined) functions: inlined into one persistent loop they cost
// 80-100 spilled VGPRs at the 128-register budget; as calls each keeps the allocation of its stand-alone kernel.
struct FusedTask {
    uint64_t *lds;
    uint64_t *out;          // first element of the limb in the output buffer
    const uint64_t *in;     // ... in the input buffer (inverse first pass only; == out when in place)
    const tw_t *tw, *twd;
    DModulus mod;
    uint32_t log_n, split, ti, twr, poly, fp;
    uint64_t n_inv, n_inv_shoup;
};

template <int S1, int RA1, bool NARROW>
__device__ __noinline__ void fused_fcol(const FusedTask &k) {
    constexpr int C = kTile >> S1;
    NttTables T{};
    T.log_n = k.log_n;
    T.split = k.split;
    const SubTree st = sub_tree<S1>(T, k.ti);
    uint64_t *base = k.out + st.offset + st.tile * C;
    if (k.fp)
        fntt_col_body<S1, RA1>(k.lds, base, k.tw, st.n2, st.root, FpField(k.mod));
    else
        fntt_col_body<S1, RA1>(k.lds, base, k.tw, st.n2, st.root, IntField<NARROW>(k.mod));
}

template <int S2, int RA2, int EPI, bool NARROW>
__device__ __noinline__ void fused_frow(const FusedTask &k, const Epilogue &ep, size_t limb_rel) {
    const uint32_t n1 = (1u << k.log_n) >> S2;
    const size_t toff = static_cast<size_t>(k.ti) * kTile;
    uint64_t *base = k.out + toff, *dst = base;
    const uint64_t *sub = nullptr, *addend = nullptr;
    uint64_t cst = 0, cst_shoup = 0;
    if (EPI != EPI_NONE) {
        const size_t rel = limb_rel + toff;
        dst = ep.out + k.poly * ep.out_poly_stride + rel;
        sub = ep.sub + k.poly * ep.sub_poly_stride + rel;
        if (EPI == EPI_SUB_MUL_ADD && (ep.addend_every == 0 || k.poly % ep.addend_every == 0))
            addend = ep.addend + k.poly * ep.out_poly_stride + rel;
        cst = ep.cst[k.twr];
        cst_shoup = ep.cst_shoup[k.twr];
    }
    if (k.fp)
        fntt_row_body<S2, RA2, EPI>(k.lds, base, k.tw, k.twd, n1, k.ti, FpField(k.mod), k.mod.value, dst, sub, addend, cst,
                                    cst_shoup);
    else
        fntt_row_body<S2, RA2, EPI>(k.lds, base, k.tw, k.twd, n1, k.ti, IntField<NARROW>(k.mod), k.mod.value, dst, sub, addend,
                                    cst, cst_shoup);
}

template <int S2, int RA2>
__device__ __noinline__ void fused_irow(const FusedTask &k) {
    const uint32_t n1 = (1u << k.log_n) >> S2;
    const size_t toff = static_cast<size_t>(k.ti) * kTile;
    if (k.fp)
        intt_row_body<S2, RA2>(k.lds, k.out + toff, k.in + toff, k.tw, k.twd, n1, k.ti, FpField(k.mod));
    else
        intt_row_body<S2, RA2>(k.lds, k.out + toff, k.in + toff, k.tw, k.twd, n1, k.ti, IntField<true>(k.mod));
}

template <int S1, int RA1, int EPI>
__device__ __noinline__ void fused_icol(const FusedTask &k, const Epilogue &ep) {
    constexpr int C = kTile >> S1;
    NttTables T{};
    T.log_n = k.log_n;
    T.split = k.split;
    const SubTree st = sub_tree<S1>(T, k.ti);
    uint64_t *base = k.out + st.offset + st.tile * C;
    uint64_t cst = 0, cst_shoup = 0;
    if (EPI == EPI_SCALE) {
        cst = ep.cst[k.twr];
        cst_shoup = ep.cst_shoup[k.twr];
    }
    if (k.fp)
        intt_col_body<S1, RA1, EPI, false>(k.lds, base, k.tw, st.n2, st.root, FpField(k.mod), k.n_inv, k.n_inv_shoup, cst,
                                           cst_shoup);
    else
        intt_col_body<S1, RA1, EPI, false>(k.lds, base, k.tw, st.n2, st.root, IntField<true>(k.mod), k.n_inv, k.n_inv_shoup,
                                           cst, cst_shoup);
}

template <bool INVERSE, int S1, int RA1, int S2, int RA2, int EPI, bool NARROW>
__global__ __launch_bounds__(kThreads) PHA_FUSED_ATTR void ntt_fused_kernel(uint64_t *out, const uint64_t *in,
                                                                            size_t poly_stride, NttTables T, LimbMap lm,
                                                                            Epilogue ep, FusedArgs fa) {
    __shared__ uint64_t lds[kLdsWords];
    __shared__ uint32_t s_ticket;
    const uint32_t n = 1u << T.log_n, tiles = n >> kTileLog;
    uint32_t *const ctl = fa.ctl;
    uint32_t q = (fa.use_xcc ? xcc_id() : blockIdx.x) % fa.nq;
    for (uint32_t hop = 0; hop < fa.nq; hop++, q = (q + 1 == fa.nq ? 0 : q + 1)) {
        if (q >= lm.active) continue;
        const uint32_t U = ((lm.active - q + fa.nq - 1) / fa.nq) * lm.polys, total = 2 * U * tiles;
        for (;;) {
            if (threadIdx.x == 0) s_ticket = ctl_add(ctl + q * 32, 1u);
            __syncthreads();
            const uint32_t t = s_ticket;
            __syncthreads();
            if (t >= total) break;
            uint32_t unit, ti;
            bool second;
            fused_decode(t, U, fa.depth, tiles, unit, ti, second);
            const uint32_t li = q + fa.nq * (unit / lm.polys), poly = unit % lm.polys;
            uint32_t twr, tidx;
            if (!resolve_limb(lm, poly, li, twr, tidx)) continue;  // ragged digit: neither pass has work
            uint32_t *done = ctl + kCtlDone + li * lm.polys + poly;
            FusedTask k;
            k.lds = lds;
            k.mod = T.mod[tidx];
            k.tw = T.tw + static_cast<size_t>(tidx) * n;
            k.twd = T.twD + static_cast<size_t>(tidx) * T.twD_stride;
            const size_t limb_rel = static_cast<size_t>(twr) * n, limb_off = poly * poly_stride + limb_rel;
            k.out = out + limb_off;
            k.in = in + limb_off;
            k.log_n = T.log_n;
            k.split = T.split;
            k.ti = ti;
            k.twr = twr;
            k.poly = poly;
            k.fp = use_fp64(T, k.mod) ? 1u : 0u;
            if (second) fused_wait(done, tiles, ctl + kCtlError);
            if (!INVERSE) {
                if (!second)
                    fused_fcol<S1, RA1, NARROW>(k);
                else
                    fused_frow<S2, RA2, (EPI == EPI_SCALE ? EPI_NONE : EPI), NARROW>(k, ep, limb_rel);
            } else {
                if (!second) {
                    fused_irow<S2, RA2>(k);
                } else {
                    k.n_inv = T.n_inv[tidx];
                    k.n_inv_shoup = T.n_inv_shoup[tidx];
                    fused_icol<S1, RA1, (EPI == EPI_SCALE ? EPI_SCALE : EPI_NONE)>(k, ep);
                }
            }
            if (!second) fused_publish(done);
        }
    }
    // leave: the last workgroup out zeroes the control words this launch used
    if (threadIdx.x == 0) {
        const uint32_t gone = ctl_add(ctl + kCtlExit, 1u);
        if (gone == gridDim.x - 1) {
            for (uint32_t x = 0; x < fa.nq; x++) __hip_atomic_store(ctl + x * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t units = lm.active * lm.polys;
            for (uint32_t u = 0; u < units; u++)
                __hip_atomic_store(ctl + kCtlDone + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + kCtlExit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace pha
