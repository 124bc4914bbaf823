// Single-launch forward negacyclic NTT for gfx950: both passes of a transform in ONE persistent launch.
//
// The two-kernel transform (ntt_kernels.cuh) pays a kernel boundary between the column and the row pass, and each of
// its launches is one resident round of workgroups that load, compute and store in lock-step (at configs[1]: 1152
// tiles on 1024 slots, then a 128-tile tail).  Here (workgroups per CU) x (CUs) persistent workgroups pull tile tasks
// from per-XCD ticket queues, the column tiles of the queue's units first, then their row tiles, so that loads, FP64
// work and stores of different tasks overlap for the whole launch and no slot idles behind a boundary.
//
// Hand-off of a unit ((poly, limb): 16 column tiles -> 16 row tiles) without a release / acquire fence
// (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire", third row):
//   * every byte of the column pass is stored write-through (`global_store_dwordx2 ... sc1`; each 128-B line is written
//     whole by one store instruction of one wave: 16 lanes x 8 B), every storing wave then waits `vmcnt(0)`, the
//     workgroup meets at a barrier and ONE lane adds 1 to the unit's arrival counter (agent-scope atomic);
//   * a row task polls that counter from ONE lane with `global_load_dword ... sc1` until all column tiles of the unit
//     have arrived, the workgroup meets at a barrier, and EVERY load of the handed-off tile is a `global_load ... sc1`
//     to registers (L1 bypassed).  In place: the only lines of a unit that any L2 ever held are the ones its column
//     tasks loaded, and those are the lines the same tasks then stored sc1 (dropped from that L2), so placement of
//     producer and consumer does not matter for correctness; XCD affinity (HW_REG_XCC_ID picks the home queue) is a
//     speed matter only.
// Forward progress: tickets of a queue are claimed in order, a unit's column tiles come before its row tiles, a column
// task never waits, and whoever claimed a ticket is resident and running, so a row task only ever waits for workgroups
// that are executing.  Spins are bounded all the same (the error word is set, the kernel drains, the host reports it).
// State: ticket / arrival / exit counters live in a control block that is zero when a launch starts; the last workgroup
// to leave zeroes it again.  One control block per context: launches of one context must not overlap (they are
// ordered on one stream; ensure_workspace's rule for two streams applies here too).
// Results are bit-identical to the two-kernel path: the same bodies run on the same tiles.
#pragma once

#include "../../phantom-fhe_amd/csrc/ntt_kernels.cuh"

namespace pha {

constexpr uint32_t kFusedQueues = 8;                  // one per XCD
constexpr uint32_t kCtlExit = kFusedQueues * 32;      // ticket of queue q at word q * 32 (own 128-B line)
constexpr uint32_t kCtlError = kCtlExit + 1;
constexpr uint32_t kCtlOwner = kCtlExit + 8;          // owner XCD (+1) of queue q at word kCtlOwner + q
constexpr uint32_t kCtlDone = kCtlExit + 32;          // arrival counters, one per unit
constexpr uint32_t kFusedMaxUnits = 4096;
constexpr uint32_t kCtlWords = kCtlDone + kFusedMaxUnits;
constexpr uint32_t kFusedSpinLimit = 1u << 21;        // x ~0.3 us of s_sleep: well over any launch, then give up

struct FusedArgs {
    uint32_t *ctl;      // kCtlWords words, zero at launch
    uint32_t nq;        // queues in use (1..8)
    uint32_t depth;     // units whose column tiles are queued ahead of the first row tiles
    uint32_t debug;     // timing experiments only (results invalid): 1 = row tasks do not wait, 2 = no ticket atomics (static
                        // task assignment by blockIdx), 4 = no publish wait, 8 = plain row loads, 16 = column tasks only
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}

// ticket -> (unit within the queue, tile, pass).  Order of tile blocks in a queue of U units with look-ahead D:
// column(0..D-1), then row(k), column(D+k) alternating, then the remaining row passes.
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
__device__ __forceinline__ uint64_t ld_sc1(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dwordx2 ... sc1
}

// column pass done: every storing wave drains its sc1 stores, barrier, ONE lane signals
__device__ __forceinline__ void fused_publish(uint32_t *done, uint32_t debug) {
    if (!(debug & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ctl_add(done, 1u);
}

// row pass: ONE lane polls the arrival counter (sc1 load), then the barrier every load of the tile stays behind
__device__ __forceinline__ void fused_wait(uint32_t *done, uint32_t need, uint32_t *err) {
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kFusedSpinLimit) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

#ifndef PHA_FUSED_WAVES
#define PHA_FUSED_WAVES 4      // register budget: this many workgroups of a CU may be resident (128 VGPRs)
#endif
#define PHA_FUSED_ATTR __attribute__((amdgpu_waves_per_eu(PHA_FUSED_WAVES, PHA_FUSED_WAVES)))
#ifndef PHA_FUSED_INLINE
#define PHA_FUSED_INLINE __forceinline__
#endif

// The two bodies are separate (not inlined) functions: inlined into one persistent loop they cost 80-100 spilled
// VGPRs at the 128-register budget; as calls each keeps the allocation of its stand-alone kernel.
template <int S1, int RA1, bool NARROW>
__device__ PHA_FUSED_INLINE void fused_fcol(uint64_t *lds, uint64_t *limb, const tw_t *tw, DModulus mod, uint32_t log_n, uint32_t ti,
                                        uint32_t fp) {
    constexpr int C = kTile >> S1;
    NttTables T{};
    T.log_n = log_n;
    T.split = 0;
    const SubTree st = sub_tree<S1>(T, ti);
    uint64_t *base = limb + st.offset + st.tile * C;
    uint64_t raw[16];
    fntt_col_load<S1, RA1>(raw, base, st.n2);
    if (fp)
        fntt_col_body<S1, RA1, false>(lds, raw, base, tw, st.n2, st.root, FpField(mod));
    else
        fntt_col_body<S1, RA1, false>(lds, raw, base, tw, st.n2, st.root, IntField<NARROW>(mod));
}

template <int S2, int RA2>
__device__ __forceinline__ void fntt_row_load_sc1(uint64_t (&raw)[16], const uint64_t *base) {
    constexpr int RB = S2 - RA2, L = 1 << S2, GC = 16 >> RA2;
#pragma unroll
    for (int g = 0; g < GC; g++) {
        const uint32_t gid = pha_tid() + kThreads * g, b_lo = gid & ((1u << RB) - 1), a = gid >> RB;
#pragma unroll
        for (int j = 0; j < (1 << RA2); j++) raw[(g << RA2) + j] = ld_sc1(&base[a * L + ((j << RB) | b_lo)]);
    }
}

template <int S2, int RA2, int EPI, bool NARROW>
__device__ PHA_FUSED_INLINE void fused_frow(uint64_t *lds, uint64_t *limb, const tw_t *tw, const tw_t *twd, DModulus mod,
                                        uint32_t log_n, uint32_t ti, uint32_t fp, uint64_t *out, const uint64_t *sub,
                                        const uint64_t *addend, uint64_t cst, uint64_t cst_shoup, bool plain_loads) {
    const uint32_t n1 = (1u << log_n) >> S2;
    const size_t toff = static_cast<size_t>(ti) * kTile;
    uint64_t raw[16];
    if (plain_loads)
        fntt_row_load<S2, RA2>(raw, limb + toff);
    else
        fntt_row_load_sc1<S2, RA2>(raw, limb + toff);
    if (fp)
        fntt_row_body<S2, RA2, EPI>(lds, raw, tw, twd, n1, ti, FpField(mod), mod.value, out + toff, sub ? sub + toff : nullptr,
                                    addend ? addend + toff : nullptr, cst, cst_shoup);
    else
        fntt_row_body<S2, RA2, EPI>(lds, raw, tw, twd, n1, ti, IntField<NARROW>(mod), mod.value, out + toff,
                                    sub ? sub + toff : nullptr, addend ? addend + toff : nullptr, cst, cst_shoup);
}

template <int S1, int RA1, int S2, int RA2, int EPI, bool NARROW>
__global__ __launch_bounds__(kThreads) PHA_FUSED_ATTR void fntt_fused_kernel(uint64_t *data, size_t poly_stride, NttTables T,
                                                                             LimbMap lm, Epilogue ep, FusedArgs fa) {
    __shared__ uint64_t lds[kLdsWords];
    __shared__ uint32_t s_ticket, s_owner;
    const uint32_t n = 1u << T.log_n, tiles = n >> kTileLog;
    uint32_t *const ctl = fa.ctl;
    const uint32_t me = __builtin_amdgcn_readfirstlane(xcc_id());
    uint32_t q = me % fa.nq;
    for (uint32_t hop = 0; hop < fa.nq; hop++, q = (q + 1 == fa.nq ? 0 : q + 1)) {
        if (q >= lm.active) continue;
        // a queue is served by the workgroups of ONE XCD (debug 32: assume every XCD is present, home queue only)
        if (fa.debug & 32) {
            if (hop) break;
        } else {
            if (threadIdx.x == 0) {
                uint32_t expected = 0;
                __hip_atomic_compare_exchange_strong(ctl + kCtlOwner + q, &expected, me + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                s_owner = expected == 0 ? me + 1 : expected;
            }
            __syncthreads();
            const uint32_t owner = __builtin_amdgcn_readfirstlane(s_owner);
            __syncthreads();
            if (owner != me + 1) continue;
        }
        // units of this queue: limb slots q, q + nq, ... x polys; tickets: all column tiles, then all row tiles
        const uint32_t U = ((lm.active - q + fa.nq - 1) / fa.nq) * lm.polys, half = U * tiles, total = 2 * half;
        uint32_t static_round = 0;
        if ((fa.debug & 2) && hop) break;
        for (;;) {
            uint32_t t;
            if (fa.debug & 2) {
                t = (blockIdx.x / fa.nq) + static_round * (gridDim.x / fa.nq);
                static_round++;
            } else {
                if (threadIdx.x == 0) s_ticket = ctl_add(ctl + q * 32, 1u);
                __syncthreads();
                t = __builtin_amdgcn_readfirstlane(s_ticket);     // uniform: everything derived from it stays scalar
                __syncthreads();
            }
            if (t >= total) break;
            uint32_t unit, ti;
            bool second;
            fused_decode(t, U, fa.depth, tiles, unit, ti, second);
            if ((fa.debug & 16) && second) continue;
            const uint32_t li = q + fa.nq * (unit / lm.polys), poly = unit % lm.polys;
            uint32_t twr, tidx;
            if (!resolve_limb(lm, poly, li, twr, tidx)) continue;  // ragged digit: neither pass has work
            uint32_t *done = ctl + kCtlDone + li * lm.polys + poly;
            const DModulus mod = T.mod[tidx];
            const tw_t *tw = T.tw + static_cast<size_t>(tidx) * n;
            const size_t limb_rel = static_cast<size_t>(twr) * n;
            uint64_t *limb = data + poly * poly_stride + limb_rel;
            const uint32_t fp = use_fp64(T, mod) ? 1u : 0u;
            if (!second) {
                fused_fcol<S1, RA1, NARROW>(lds, limb, tw, mod, T.log_n, ti, fp);
                fused_publish(done, fa.debug);
            } else {
                if (!(fa.debug & 1)) fused_wait(done, tiles, ctl + kCtlError);
                const tw_t *twd = T.twD + static_cast<size_t>(tidx) * T.twD_stride;
                uint64_t *out = limb;
                const uint64_t *sub = nullptr, *addend = nullptr;
                uint64_t cst = 0, cst_shoup = 0;
                if (EPI != EPI_NONE) {
                    out = ep.out + poly * ep.out_poly_stride + limb_rel;
                    sub = ep.sub + poly * ep.sub_poly_stride + limb_rel;
                    if (EPI == EPI_SUB_MUL_ADD && (ep.addend_every == 0 || poly % ep.addend_every == 0))
                        addend = ep.addend + poly * ep.out_poly_stride + limb_rel;
                    cst = ep.cst[twr - ep.cst_limb0];
                    cst_shoup = ep.cst_shoup[twr - ep.cst_limb0];
                }
                fused_frow<S2, RA2, EPI, NARROW>(lds, limb, tw, twd, mod, T.log_n, ti, fp, out, sub, addend, cst, cst_shoup, (fa.debug & 8) != 0);
                __syncthreads();   // the tile image in LDS is free for the next task
            }
        }
    }
    // leave: the last workgroup out zeroes the control words this launch used
    if (threadIdx.x == 0) {
        const uint32_t gone = ctl_add(ctl + kCtlExit, 1u);
        if (gone == gridDim.x - 1) {
            for (uint32_t x = 0; x < fa.nq; x++) {
                __hip_atomic_store(ctl + x * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ctl + kCtlOwner + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t units = lm.active * lm.polys;
            for (uint32_t u = 0; u < units; u++)
                __hip_atomic_store(ctl + kCtlDone + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + kCtlExit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace pha
