// Forward negacyclic NTT in ONE persistent, software-pipelined launch (gfx950), N = 2^(8+8).
//
// The two-kernel transform (ntt_kernels.cuh) moves every limb through the memory fabric twice, and each of its
// launches is one resident round of workgroups that load, compute and store one after the other.  Here
//   * the column pass hands its result to the row pass inside one XCD's L2: every tile of a unit ((poly, limb), 512 KiB
//     at N = 2^16) is processed by workgroups of ONE XCD, column tiles are stored with plain stores (the lines stay in
//     that XCD's write-back L2) and row tiles read them back with sc1 loads (L1 bypassed, L2 served);
//   * every workgroup is a pipeline whose memory side is written by hand: the NEXT tile (and the twiddles its register
//     passes read through LDS) arrives by LDS-DMA (global_load_lds_dwordx4, no register destination) in the other half
//     of a double-buffered LDS image while the current tile is transformed in place in its half; results leave through
//     asm stores; every wait is a counted s_waitcnt vmcnt(N) of this file and every barrier a raw s_barrier, so hipcc
//     inserts no vmcnt(0) anywhere in the loop (it cannot see the asm memory operations; the one kind of load it does
//     see, the row pass's per-lane last-pass twiddles, is collected where nothing else is in flight);
//   * two workgroups per CU (72 KiB of LDS each) with complementary roles: one prefers column tiles (fabric-bound
//     loads), the other row tiles (L2-bound loads), each taking the other kind when its own queue is empty or blocked.
//
// Scheduling state (per launch, in a control block that is zero when a launch starts and is zeroed again by the last
// workgroup to leave): per XCD queue a column ticket counter and a row ticket counter, per unit an arrival counter.
// Tickets of a queue are handed out in unit order.  A column task never waits.  A row task waits until all column
// tiles of its unit have arrived; a workgroup that holds such a ticket runs column tasks meanwhile, and only spins
// when no column ticket is left -- then every column task is already held by a running workgroup, so the wait ends.
// Hence a single resident workgroup per XCD completes the launch alone: no assumption about co-residency or dispatch
// order, only that every XCC id of the context's census (FusedArgs::xcc_queue) receives at least one workgroup, which a
// grid of two workgroups per CU always gives.  Spins are bounded all the same (error word, results then invalid).
//
// Hand-off correctness inside one XCD: a column task's stores are acknowledged by the L2 (vmcnt) in every wave, the
// workgroup meets at a barrier and one lane adds to the unit's arrival counter (agent scope); a row task learns the
// count from an sc1 load, meets at a barrier, and fetches the tile with sc1 LDS-DMA loads (never from its CU's L1).
// Tiles are whole 128-B lines, so no line is shared between tasks.  Results are bit-identical to the two-kernel path:
// the same register passes run on the same tiles.
#pragma once

#include <type_traits>

#include "ntt_kernels.cuh"

namespace pha {

constexpr uint32_t kFusedQueues = 8;                        // at most one per XCD
constexpr uint32_t kFqStride = 64;                          // words per queue: column tickets at +0, row tickets at +32
constexpr uint32_t kFCtlExit = kFusedQueues * kFqStride;    // workgroups that have left
constexpr uint32_t kFCtlError = kFCtlExit + 1;              // sticky: a bounded spin ran out / unknown XCC id
constexpr uint32_t kFCtlDone = kFCtlExit + 32;              // arrival counters, one per unit (limb slot * polys + poly)
constexpr uint32_t kFusedMaxUnits = 8192;
constexpr uint32_t kFCtlWords = kFCtlDone + kFusedMaxUnits;
constexpr uint32_t kFusedSpinLimit = 1u << 22;
constexpr int kFusedWgsPerCu = 2;                           // what the LDS footprint admits

enum : uint32_t {
    FUSED_ROLE_PARITY = 1,         // role by parity of the workgroup's index within its XCD (default: first half / second half)
    FUSED_ALL_COLUMN_FIRST = 2,    // experiment: every workgroup prefers column tiles (row tiles when none is left)
    FUSED_START_WITH_COLUMN = 4,   // row-role workgroups take one column tile first (nothing has arrived at the start)
    FUSED_ROW_PLAIN_LOADS = 16,    // timing experiment ONLY (results may be stale): row tiles fetched without sc1
    FUSED_NO_WAIT = 32,            // timing experiment ONLY (results wrong): row tiles do not wait for their unit
    FUSED_PATIENCE_SHIFT = 8,      // bits 8..15: polls (x ~0.1 us) a row tile waits for its unit before it runs a column
                                   // tile meanwhile (0: none)
};

struct FusedArgs {
    uint32_t *ctl;
    uint32_t nq;                   // queues in use
    uint32_t role_split;           // workgroups per XCD / 2
    uint64_t xcc_queue;            // byte x: queue served by XCC id x (0xff: none)
    uint32_t flags;
};

__device__ __forceinline__ uint32_t fused_xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}
__device__ __forceinline__ uint32_t fctl_add(uint32_t *p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t fctl_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_load_dword ... sc1
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// units of queue q: limb slots q, q + nq, ... of every poly, then the limb slots beyond a multiple of nq dealt unit by
// unit (as decode_block does for the two-kernel path), polys of a limb adjacent (they share the row pass's twiddles)
__host__ __device__ __forceinline__ uint32_t fused_units(uint32_t active, uint32_t polys, uint32_t nq, uint32_t q) {
    const uint32_t full = active / nq, left = (active % nq) * polys;
    return full * polys + (left > q ? (left - q + nq - 1) / nq : 0);
}
__device__ __forceinline__ void fused_unit(uint32_t k, uint32_t active, uint32_t polys, uint32_t nq, uint32_t q, uint32_t &li,
                                           uint32_t &poly) {
    const uint32_t full = active / nq;
    if (k < full * polys) {
        const uint32_t a = k / polys;
        li = a * nq + q;
        poly = k - a * polys;
    } else {
        const uint32_t v = (k - full * polys) * nq + q, a = v / polys;
        li = full * nq + a;
        poly = v - a * polys;
    }
}

constexpr uint32_t kTaskNone = 0, kTaskCol = 1, kTaskRow = 2, kTaskEmpty = 4;   // kTaskEmpty: the unit has no work (ragged digit)
constexpr uint32_t kTicketValid = 0x80000000u, kTicketRow = 0x40000000u;

struct FusedTask {
    uint32_t kind, ti, gunit, tidx;
    uint32_t off_lo, off_hi;        // word offset of the limb in `data` (integers: wave-uniform, address space kept)
};
__device__ __forceinline__ size_t task_offset(const FusedTask &k) { return (static_cast<size_t>(uni(k.off_hi)) << 32) | uni(k.off_lo); }

// a modulus record through the scalar cache (the tables are never written by a kernel); inside the persistent loop hipcc
// would use a vector load behind a full vmcnt wait, because stores and barriers precede it
__device__ __forceinline__ DModulus sload_modulus(const DModulus *p) {
    uint64_t a, b, c;
    asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(p) : "memory");
    return DModulus{a, b, c};
}

// ---- hand-written memory side -----------------------------------------------------------------------------------
// LDS-DMA: 64 lanes x 16 B from per-lane global addresses (sbase + voff) to LDS bytes [lds_dst, lds_dst + 1024), lane
// order.  M0 carries the destination and is saved / restored around the instruction (hipcc reserves it).
template <bool SC1>
__device__ __forceinline__ void dma16(const void *sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    if (SC1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void st_plain8(uint64_t *p, uint64_t v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1_16(uint64_t *p, ulonglong2 v) {
    const u64x2_t w = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");
}
// workgroup barrier that waits for this wave's LDS operations only (a __syncthreads() may add a vmcnt(0))
__device__ __forceinline__ void fsync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define PHA_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Control operations of the controller lane whose latency is hidden behind a register pass: a ticket claim (returning
// atomic add) and an arrival poll (sc1 load).  Their results land in accumulator registers a0 / a2, which the compiler
// does not allocate in this kernel (no MFMA, no spills: audited by tools/check_fused_isa.py), so nothing can copy or
// reuse the destination before the data is there; a counted s_waitcnt before v_accvgpr_read collects them.
__device__ __forceinline__ void fctl_async_init() { asm volatile("v_accvgpr_write_b32 a1, %0" ::"v"(1u) : "a1"); }
__device__ __forceinline__ void fctl_async_claim(uint32_t *counter) {
    asm volatile("s_nop 4\n\tglobal_atomic_add a0, %0, a1, %1 sc0" ::"v"(0u), "s"(counter) : "memory", "a0");
}
__device__ __forceinline__ void fctl_async_poll(const uint32_t *word) {
    asm volatile("s_nop 4\n\tglobal_load_dword a2, %0, %1 sc1" ::"v"(0u), "s"(word) : "memory", "a2");
}
__device__ __forceinline__ uint32_t fctl_async_claimed() {
    uint32_t v;
    asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v)::"a0");
    return v;
}
__device__ __forceinline__ uint32_t fctl_async_polled() {
    uint32_t v;
    asm volatile("v_accvgpr_read_b32 %0, a2" : "=v"(v)::"a2");
    return v;
}

// Development aid (tools/trace_fused.py, library built with -DPHA_TRACE): per workgroup and task, six words -- the
// ticket, then the 100-MHz clock at the top of the iteration, after the first register pass, after the middle barrier,
// after the second pass's stores were issued and after the end barrier.  The stamps are LDS stores of the controller lane,
// kept in LDS and copied out when the workgroup leaves, so that tracing adds no global memory operation to the loop.
#ifdef PHA_TRACE
#define PHA_FTRACE(slot, value)                                                        \
    do {                                                                               \
        if (ctrl && trace_i < 25) trace_lds[trace_i * 10 + (slot)] = (value);           \
    } while (0)
constexpr uint32_t kFusedTraceBytes = 256 * 8;
#else
#define PHA_FTRACE(slot, value)
constexpr uint32_t kFusedTraceBytes = 0;
#endif

// ---- LDS map (one array): two tile images, two twiddle stagings, mailbox -----------------------------------------
constexpr uint32_t kImgBytes = kTile * 8;                   // 32 KiB, unpadded (XOR-swizzled at 16-B / row granularity)
constexpr uint32_t kStageBytes = 4096;                      // 256 records of 16 B
constexpr uint32_t kLdsImg = 0, kLdsStage = 2 * kImgBytes, kLdsMb = kLdsStage + 2 * kStageBytes, kFusedLdsBytes = kLdsMb + 64;

// column image [256 rows][16 cols]: rows 16 apart (the second pass's half-waves) in different halves of a bank row
__device__ __forceinline__ uint32_t sw_col(uint32_t r, uint32_t c) { return ((r ^ ((r >> 4) & 1u)) << 4) | c; }
// row image, 128-B rows of eight 16-B pieces: piece ^= (row >> 1) & 7, row ^= (row >> 4) & 1
__device__ __forceinline__ uint32_t sw_row(uint32_t e) {
    const uint32_t row = e >> 4;
    return ((row ^ ((row >> 4) & 1u)) << 4) | ((((e >> 1) ^ (row >> 1)) & 7u) << 1) | (e & 1u);
}

template <class TW>
struct StageTw {      // twiddles staged in LDS in canonical order: record (node << s) + jb
    const TW *tw;
    uint32_t node;
    __device__ __forceinline__ TW operator()(int s, int jb) const { return tw[(node << s) + jb]; }
};
template <class TW>
struct StageRowTw {   // the 15 twiddles of one row's first pass, 16-B records (an FP64 twiddle sits in the first 8 bytes)
    const ulonglong2 *rec;
    __device__ __forceinline__ TW operator()(int s, int jb) const { return *reinterpret_cast<const TW *>(rec + ((1 << s) - 1 + jb)); }
};
template <class TW>
struct RegTw {        // the 15 last-pass twiddles of a lane, loaded a register pass ahead
    const TW *tw;
    __device__ __forceinline__ TW operator()(int s, int jb) const { return tw[(1 << s) - 1 + jb]; }
};

// ---- tile bodies on an LDS image (S1 = S2 = 8, two radix-16 passes each) -----------------------------------------
template <class F>
__device__ __forceinline__ void fcol_part1(uint64_t *img, const void *stage, const F &f) {
    using TW = typename F::twl;
    const uint32_t t = pha_tid(), c = t & 15u, r_lo = t >> 4;
    typename F::elem x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = f.from_u64(img[sw_col((j << 4) | r_lo, c)]);
    ct_radix<4, 0>(x, StageTw<TW>{reinterpret_cast<const TW *>(stage), 1u}, f);
#pragma unroll
    for (int j = 0; j < 16; j++) img[sw_col((j << 4) | r_lo, c)] = F::raw(x[j]);
}
template <class F>
__device__ __forceinline__ void fcol_part2(const uint64_t *img, const void *stage, uint64_t *base, uint32_t n2, const F &f) {
    using TW = typename F::twl;
    const uint32_t t = pha_tid(), c = t & 15u, r_hi = t >> 4;
    typename F::elem x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = F::unraw(img[sw_col((r_hi << 4) | j, c)]);
    ct_radix<4, 4>(x, StageTw<TW>{reinterpret_cast<const TW *>(stage), 16u + r_hi}, f);
    // plain stores: the lines stay (dirty) in this XCD's L2 for the row pass
#pragma unroll
    for (int j = 0; j < 16; j++) st_plain8(&base[static_cast<size_t>((r_hi << 4) | j) * n2 + c], F::raw(x[j]));
}
template <class F>
__device__ __forceinline__ void frow_part1(uint64_t *img, const void *stage, const F &f) {
    using TW = typename F::twl;
    const uint32_t t = pha_tid(), b_lo = t & 15u, a = t >> 4;
    typename F::elem x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = F::unraw(img[sw_row(a * 256 + ((j << 4) | b_lo))]);
    ct_radix<4, 0>(x, StageRowTw<TW>{reinterpret_cast<const ulonglong2 *>(stage) + a * 15}, f);
#pragma unroll
    for (int j = 0; j < 16; j++) img[sw_row(a * 256 + ((j << 4) | b_lo))] = F::raw(x[j]);
}
template <class F>
__device__ __forceinline__ void frow_part2(uint64_t *img, const typename F::twl *twd, const F &f) {
    using TW = typename F::twl;
    const uint32_t t = pha_tid(), e0 = t << 4;
    typename F::elem x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = F::unraw(img[sw_row(e0 + j)]);
    ct_radix<4, 4>(x, RegTw<TW>{twd}, f);
#pragma unroll
    for (int j = 0; j < 16; j++) img[sw_row(e0 + j)] = F::raw(f.canon_fwd(x[j]));
}
template <class F>
__device__ __forceinline__ void frow_part3(const uint64_t *img, const F &f, uint64_t *out) {
    const uint32_t t = pha_tid();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t e = (i * kThreads + t) * 2;
        const typename F::elem v0 = F::unraw(img[sw_row(e)]), v1 = F::unraw(img[sw_row(e + 1)]);
        st_sc1_16(out + e, ulonglong2{f.to_u64(v0), f.to_u64(v1)});
    }
}

// The accumulator registers of the asynchronous control operations make the compiler split its register budget evenly
// between VGPRs and AGPRs; the budget of ONE wave per SIMD (256 + 256) leaves the VGPR side what two waves per SIMD can
// use.  What the hardware admits follows from the registers actually allocated (audited: <= 256 in total).
template <bool NARROW>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void fntt_fused_kernel(
    uint64_t *data, size_t poly_stride, NttTables T, LimbMap lm, FusedArgs fa) {
    static_assert(kTileLog == 12 && kThreads == 256, "the single-launch transform is written for 4096-coefficient tiles");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kFusedLdsBytes + kFusedTraceBytes];
#ifdef PHA_TRACE
    unsigned long long *const trace_lds = reinterpret_cast<unsigned long long *>(smem + kFusedLdsBytes);
    uint32_t trace_i = 0;
    trace_lds[threadIdx.x] = 0;
#endif
    uint32_t *const mb = reinterpret_cast<uint32_t *>(smem + kLdsMb);
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));     // LDS byte address of smem
    const uint32_t n = 1u << T.log_n, tiles_log = T.log_n - kTileLog, tiles = 1u << tiles_log, n2 = n >> 8, n1 = n >> 8;
    uint32_t *const ctl = fa.ctl;
    const bool ctrl = threadIdx.x == 0;
    const uint32_t wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t xcc = uni(fused_xcc_id());
    const uint32_t q = static_cast<uint32_t>(fa.xcc_queue >> (8 * (xcc & 7u))) & 0xffu;

    uint32_t pend = ~0u;                 // unit of a column tile whose stores are issued but not yet signalled
    if (q < fa.nq) {
        const uint32_t U = fused_units(lm.active, lm.polys, fa.nq, q), total = U << tiles_log;
        uint32_t *const tk_col = ctl + q * kFqStride, *const tk_row = tk_col + 32;
        const uint32_t widx = blockIdx.x >> 3;   // index within the XCD under round-robin dispatch (a speed matter only)
        uint32_t pref = (fa.flags & FUSED_ALL_COLUMN_FIRST) ? kTaskCol
                        : ((fa.flags & FUSED_ROLE_PARITY) ? (widx & 1u) : (widx >= fa.role_split)) ? kTaskRow : kTaskCol;

        // per-lane source offsets of the DMA chunks (lane L fills piece L of a 1-KiB chunk; the swizzles go on the source)
        const uint32_t lrow = lane >> 3, piece = lane & 7u;
        uint32_t voff_col[2], voff_row[4];
#pragma unroll
        for (uint32_t v = 0; v < 2; v++) voff_col[v] = (lrow ^ v) * (n2 * 8u) + piece * 16u;
#pragma unroll
        for (uint32_t v = 0; v < 4; v++) voff_row[v] = (lrow ^ (v >> 1)) * 128u + (piece ^ (((v & 1u) << 2) | (lrow >> 1))) * 16u;
        // first-pass twiddles of a row tile: lane t < 240 stages record (row t / 15, slot t % 15)
        const uint32_t st_a = threadIdx.x / 15u, st_slot = threadIdx.x - st_a * 15u;
        const uint32_t st_s = 31u - __builtin_clz(st_slot + 1u), st_jb = st_slot + 1u - (1u << st_s);

        // controller lane only: claim a ticket, preferred kind first; 0 = both queues are empty
        auto claim_sync = [&](uint32_t first) -> uint32_t {
            for (int pass = 0; pass < 2; pass++) {
                const uint32_t kind = pass == 0 ? first : (first == kTaskCol ? kTaskRow : kTaskCol);
                // do not run a counter that is already exhausted further up than needed: peek first
                if (fctl_load(kind == kTaskCol ? tk_col : tk_row) >= total) continue;
                const uint32_t t = fctl_add(kind == kTaskCol ? tk_col : tk_row, 1u);
                if (t < total) return kTicketValid | (kind == kTaskRow ? kTicketRow : 0u) | t;
            }
            return 0u;
        };
        // ticket -> task; every field is wave-uniform (the ticket word comes out of the mailbox through readfirstlane)
        auto decode = [&](uint32_t enc) -> FusedTask {
            FusedTask k{kTaskNone, 0, 0, 0, 0, 0};
            if (!(enc & kTicketValid)) return k;
            const uint32_t t = enc & 0x3fffffffu, unit = t >> tiles_log;
            uint32_t li, poly, twr, tidx;
            fused_unit(unit, lm.active, lm.polys, fa.nq, q, li, poly);
            k.ti = t & (tiles - 1);
            k.gunit = li * lm.polys + poly;
            k.kind = (enc & kTicketRow) ? kTaskRow : kTaskCol;
            if (!resolve_limb(lm, poly, li, twr, tidx)) {    // ragged digit: neither pass has work; a column "task" still
                k.kind |= kTaskEmpty;                        // arrives so that nothing ever waits for it
                return k;
            }
            k.tidx = tidx;
            const size_t off = poly * poly_stride + static_cast<size_t>(twr) * n;
            k.off_lo = static_cast<uint32_t>(off);
            k.off_hi = static_cast<uint32_t>(off >> 32);
            return k;
        };
        // 9 DMA instructions per wave: the twiddle staging first, then the wave's 8 chunks of the tile (chunk = 8 image rows)
        auto dma_col = [&](const FusedTask &k, uint32_t buf) {
            const uint32_t tidx = uni(k.tidx);
            const unsigned char *twb = reinterpret_cast<const unsigned char *>(T.tw + static_cast<size_t>(tidx) * n);
            dma16<false>(twb, threadIdx.x * 16u, lds_base + kLdsStage + buf * kStageBytes + wave * 1024u);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + task_offset(k) + uni(k.ti) * 16u);
            const uint32_t img = lds_base + kLdsImg + buf * kImgBytes;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                const uint32_t ch = wave * 8u + i;                       // image rows 8 ch .. 8 ch + 7
                dma16<false>(src + static_cast<size_t>(ch) * 8u * n2 * 8u, ((i >> 1) & 1u) ? voff_col[1] : voff_col[0], img + ch * 1024u);
            }
        };
        auto dma_row = [&](const FusedTask &k, uint32_t buf) {
            const uint32_t tidx = uni(k.tidx), ti = uni(k.ti);
            // record size of the limb's table: 8 B (FP64 limbs keep doubles) or 16 B
            const uint32_t rec_log = use_fp64(T, sload_modulus(T.mod + tidx)) ? 3u : 4u;
            const unsigned char *twb = reinterpret_cast<const unsigned char *>(T.tw + static_cast<size_t>(tidx) * n);
            const uint32_t idx = ((n1 + ti * 16u + st_a) << st_s) + st_jb;
            dma16<false>(twb, threadIdx.x < 240u ? idx << rec_log : 0u, lds_base + kLdsStage + buf * kStageBytes + wave * 1024u);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(data + task_offset(k) + static_cast<size_t>(ti) * kTile);
            const uint32_t img = lds_base + kLdsImg + buf * kImgBytes;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                const uint32_t ch = wave * 8u + i;
                const uint32_t vo = voff_row[((i >> 1) & 1u) * 2u + (i & 1u)];
                if (fa.flags & FUSED_ROW_PLAIN_LOADS)
                    dma16<false>(src + ch * 1024u, vo, img + ch * 1024u);
                else
                    dma16<true>(src + ch * 1024u, vo, img + ch * 1024u);
            }
        };
        auto publish = [&]() {          // after the storing waves' vmcnt wait and a barrier
            if (ctrl && pend != ~0u) fctl_add(ctl + kFCtlDone + pend, 1u);
            pend = ~0u;
        };
        auto mbox = [&](int i) -> uint32_t { return uni(mb[i]); };

        const uint32_t role = pref;
        if (fa.flags & FUSED_START_WITH_COLUMN) pref = kTaskCol;
        if (ctrl) {
            fctl_async_init();
            mb[0] = claim_sync(pref);
        }
        fsync();
        pref = role;
        const uint32_t patience = (fa.flags >> FUSED_PATIENCE_SHIFT) & 0xffu;
        FusedTask cur{kTaskNone, 0, 0, 0, 0, 0}, next = decode(mbox(0));
        uint32_t fol_enc = 0, cb = 0;           // cb: the half of the LDS image / staging that holds the current tile
        bool have_fol = false, next_loaded = false;

        for (;;) {
            PHA_FTRACE(0, (static_cast<unsigned long long>(cur.kind) << 32) | (cur.gunit << 8) | cur.ti);
            PHA_FTRACE(1, wall_clock64());
            // ---- top: control operations first (collected at the middle barrier), then the next column tile's DMA, a
            // whole task ahead (no dependency), then the current row tile's last-pass twiddles (used after the barrier)
            const bool want_claim = next.kind != kTaskNone && !have_fol;
            const bool want_poll = next.kind == kTaskRow;
            if (ctrl) {
                if (want_claim) fctl_async_claim(pref == kTaskCol ? tk_col : tk_row);
                if (want_poll) fctl_async_poll(ctl + kFCtlDone + next.gunit);
            }
            next_loaded = false;
            if (next.kind == kTaskCol) {
                dma_col(next, cb ^ 1u);
                next_loaded = true;
            } else if (next.kind & kTaskEmpty) {
                next_loaded = true;                             // nothing to load
            }
            const uint32_t ctidx = uni(cur.tidx), cti = uni(cur.ti);
            const DModulus mod = sload_modulus(T.mod + ctidx);
            uint64_t *const img = reinterpret_cast<uint64_t *>(smem + kLdsImg + cb * kImgBytes);
            const void *const stage = smem + kLdsStage + cb * kStageBytes;

            // The middle of an iteration, shared by every task kind: the counted wait, the controller's mailbox work, the
            // barrier, the signal for the column tile before this one, and a row tile's fetch once its unit has arrived.
            auto middle = [&](bool col_col) {
                PHA_FTRACE(2, wall_clock64());
                // Behind this wait: the stores of the column tile before this one (pend) in every wave, the controller's
                // claim and poll, and a row tile's last-pass twiddles.  Younger than all of those are only the 9 DMA
                // instructions of the next column tile, when the current tile is a column tile too.
                if (col_col)
                    PHA_VMCNT(9);
                else
                    PHA_VMCNT(0);
                if (ctrl) {
                    if (want_poll) mb[1] = (fctl_async_polled() >= tiles || (fa.flags & FUSED_NO_WAIT)) ? 1u : 0u;
                    if (want_claim) {
                        const uint32_t tk = fctl_async_claimed();
                        uint32_t enc;
                        if (tk < total)
                            enc = kTicketValid | (pref == kTaskRow ? kTicketRow : 0u) | tk;
                        else
                            enc = claim_sync(pref == kTaskCol ? kTaskRow : kTaskCol);    // own queue is empty: the other kind
                        mb[2] = enc;
                    }
                }
                fsync();
                PHA_FTRACE(3, wall_clock64());
                publish();
                if (want_claim) {
                    fol_enc = mbox(2);
                    have_fol = true;
                    if (fol_enc & kTicketValid) pref = (fol_enc & kTicketRow) ? kTaskRow : kTaskCol;
                }
                if (want_poll && mbox(1)) {                     // its unit has arrived: fetch it under the second pass
                    dma_row(next, cb ^ 1u);
                    next_loaded = true;
                }
                PHA_FTRACE(6, wall_clock64());
            };
            // One instantiation per (kind, field): nothing of one is merged with another (hipcc would otherwise load both
            // twiddle formats and select), and the register passes run in place in the tile's LDS image.
            auto col_task = [&](const auto &f) {
                fcol_part1(img, stage, f);
                middle(next.kind == kTaskCol);
                fcol_part2(img, stage, data + task_offset(cur) + cti * 16u, n2, f);
                pend = cur.gunit;
                PHA_FTRACE(4, wall_clock64());
                PHA_VMCNT(16);                                  // the next tile's DMA; the 16 stores stay in flight
            };
            auto row_task = [&](const auto &f) {
                using F = std::remove_cv_t<std::remove_reference_t<decltype(f)>>;
                using TW = typename F::twl;
                // the lane's 15 last-pass twiddles, a register pass ahead of their use
                const uint32_t ggid = cti * 256u + pha_tid();
                const TW *p = reinterpret_cast<const TW *>(T.twD + static_cast<size_t>(ctidx) * T.twD_stride) +
                              static_cast<size_t>(ggid >> 6) * (15 * 64) + (ggid & 63u);
                TW twd[15];
#pragma unroll
                for (int s = 0; s < 15; s++) twd[s] = p[s * 64];
                frow_part1(img, stage, f);
                PHA_VMCNT(0);
                // hipcc's own wait for the twiddle registers lands here, where nothing else is in flight
#pragma unroll
                for (int s = 0; s < 15; s++) {
                    if constexpr (sizeof(TW) == 8)
                        asm volatile("" : "+v"(twd[s]));
                    else
                        asm volatile("" : "+v"(twd[s].x), "+v"(twd[s].y));
                }
                middle(false);
                frow_part2(img, twd, f);
                PHA_FTRACE(7, wall_clock64());
                fsync();
                PHA_FTRACE(8, wall_clock64());
                frow_part3(img, f, data + task_offset(cur) + static_cast<size_t>(cti) * kTile);
                PHA_FTRACE(4, wall_clock64());
                PHA_VMCNT(8);                                   // the next tile's DMA; the 8 stores stay in flight
            };
            if (cur.kind == kTaskCol) {
                if (use_fp64(T, mod))
                    col_task(FpField(mod));
                else
                    col_task(IntField<NARROW>(mod));
            } else if (cur.kind == kTaskRow) {
                if (use_fp64(T, mod))
                    row_task(FpField(mod));
                else
                    row_task(IntField<NARROW>(mod));
            } else {
                middle(false);
                if (cur.kind == (kTaskCol | kTaskEmpty)) pend = cur.gunit;      // a unit without work still arrives
                PHA_VMCNT(0);
            }
            fsync();                                            // the next image is complete; this one is free
            PHA_FTRACE(5, wall_clock64());
#ifdef PHA_TRACE
            trace_i++;
#endif
            if (next.kind == kTaskNone) break;
            if (!next_loaded) {
                // a row tile whose unit had not arrived at the middle barrier: signal what is pending, then either the
                // unit has arrived by now, or run a column tile meanwhile, or (no column ticket left) wait for it
                PHA_VMCNT(0);
                fsync();
                publish();
                if (ctrl) {
                    uint32_t verdict = 0, enc = 0;
                    uint32_t *const done = ctl + kFCtlDone + next.gunit;
                    uint32_t arrived = fctl_load(done);
                    for (uint32_t w = 0; w < patience && arrived < tiles; w++) {
                        __builtin_amdgcn_s_sleep(4);
                        arrived = fctl_load(done);
                    }
                    if (arrived >= tiles || (fa.flags & FUSED_NO_WAIT)) {
                        verdict = 1;
                    } else {
                        if (fctl_load(tk_col) < total) {
                            const uint32_t t = fctl_add(tk_col, 1u);
                            if (t < total) enc = kTicketValid | t;
                        }
                        if (enc) {
                            verdict = 2;
                        } else {
                            uint32_t spins = 0;
                            while (fctl_load(done) < tiles) {
                                __builtin_amdgcn_s_sleep(4);
                                if (++spins > kFusedSpinLimit) {
                                    __hip_atomic_fetch_or(ctl + kFCtlError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                            }
                            verdict = 1;
                        }
                    }
                    mb[3] = verdict;
                    mb[4] = enc;
                }
                fsync();
                if (mbox(3) == 2u) {
                    cur = decode(mbox(4));          // a column tile first; `next` (the row tile) and `fol_enc` stay
                    if (cur.kind == kTaskCol) dma_col(cur, cb);
                    PHA_VMCNT(0);
                    fsync();
                    continue;
                }
                dma_row(next, cb ^ 1u);
                PHA_VMCNT(0);
                fsync();
            }
            cur = next;
            cb ^= 1u;
            next = decode(fol_enc);
            have_fol = false;
        }
    } else if (ctrl) {
        __hip_atomic_fetch_or(ctl + kFCtlError, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- leave: signal what is pending; the last workgroup out zeroes the control words this launch used
    PHA_VMCNT(0);
    fsync();
#ifdef PHA_TRACE
    if (pha_trace_buf) pha_trace_buf[blockIdx.x * 256 + threadIdx.x] = trace_lds[threadIdx.x];
#endif
    if (ctrl) {
        if (pend != ~0u) fctl_add(ctl + kFCtlDone + pend, 1u);
        mb[5] = fctl_add(ctl + kFCtlExit, 1u);
    }
    fsync();
    if (uni(mb[5]) == gridDim.x - 1) {
        const uint32_t units = lm.active * lm.polys;
        for (uint32_t u = threadIdx.x; u < units; u += kThreads)
            __hip_atomic_store(ctl + kFCtlDone + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < kFusedQueues * 2)
            __hip_atomic_store(ctl + (threadIdx.x >> 1) * kFqStride + (threadIdx.x & 1u) * 32, 0u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (ctrl) __hip_atomic_store(ctl + kFCtlExit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// which XCC ids a grid of this device lands on (context creation): ids[block] = HW_REG_XCC_ID
__global__ void fused_census_kernel(uint32_t *ids) {
    if (threadIdx.x == 0) ids[blockIdx.x] = fused_xcc_id();
}

}  // namespace pha
