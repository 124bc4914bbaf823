// C ABI of the MI355X-native RNS hot path (include/phantom_amd.h): context / table upload,
// kernel launch sequences for the NTT family, hybrid key switch, Galois, rescale and tensor
// product.  Host orchestration follows the dataflow of src/rns_bconv.cu, src/eval_key_switch.cu,
// src/rns.cu:1157-1181 and src/evaluate.cu of the reference; kernels are in *_kernels.cuh.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <sys/random.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/phantom_amd.h"
#include "ntt_kernels.cuh"
#include "ntt_fused.cuh"
#include "numth.hpp"
#include "rns_kernels.cuh"
#include "keygen_kernels.cuh"
#include "encoder_kernels.cuh"

using namespace pha;

// Scratch of pha_rescale_to_next_noalloc, the one entry point that allocates nothing: the inverted last limbs of up to
// three polys at N = 2^17 (3 MiB on every GPU the library runs on, until the process ends).  In host code `scratch`
// is a stand-in, not a device address: pha_context_create resolves it once per device with hipGetSymbolAddress and
// every host use goes through g_scratch_ptr[device].
constexpr size_t kNoAllocScratchWords = PHA_NOALLOC_SCRATCH_WORDS;
static __device__ __align__(256) uint64_t scratch[kNoAllocScratchWords];

namespace {

constexpr int kMaxDevices = 64;
uint64_t *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;   // guards the fill of g_scratch_ptr (pha_context_create)

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// tables of a level / Galois element / merged rescale are built and uploaded on first use: under a stream capture in
// global mode (torch.cuda.graph's default) those copies are refused and the capture is invalidated
inline const char *capture_hint(hipError_t e) {
    return (e == hipErrorStreamCaptureUnsupported || e == hipErrorStreamCaptureInvalidated || e == hipErrorStreamCaptureImplicit)
               ? " (a table is built on first use: run the call once, or pha_prepare_level, before capturing)" : "";
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(PHA_ERR_HIP, "%s failed: %s%s", #expr, hipGetErrorString(e_), capture_hint(e_)); \
    } while (0)

#define PHA_TRY(expr)             \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != PHA_OK) return rc_; \
    } while (0)

// NTT decomposition per log N: S1 column stages (first pass radix 2^RA1), S2 row stages
struct NttPlan {
    int s1, ra1, s2, ra2;
    int split;  // 1: the outermost radix-2 stage runs in its own kernel (N = 2^17)
    int small;  // 1: the polynomial is smaller than a tile; one workgroup per limb, all stages in LDS (N = 2^11)
};
bool plan_for(int log_n, NttPlan &p) {
    switch (log_n) {
        case 11: p = {0, 0, 0, 0, 0, 1}; return true;
        case 12: p = {6, 3, 6, 3, 0, 0}; return true;
        case 13: p = {6, 3, 7, 4, 0, 0}; return true;
        case 14: p = {7, 4, 7, 4, 0, 0}; return true;
        case 15: p = {7, 4, 8, 4, 0, 0}; return true;
        case 16: p = {8, 4, 8, 4, 0, 0}; return true;
        case 17: p = {8, 4, 8, 4, 1, 0}; return true;
        default: return false;
    }
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        count = n;
        return n ? hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)) : hipSuccess;
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// value + Shoup companion, indexed by absolute limb
struct ShoupVec {
    DevBuf<uint64_t> v, s;
    hipError_t upload(const std::vector<uint64_t> &vals, const uint64_t *moduli) {
        std::vector<uint64_t> sh(vals.size());
        for (size_t i = 0; i < vals.size(); i++) sh[i] = shoup(vals[i], moduli[i]);
        hipError_t e = v.upload(vals);
        return e != hipSuccess ? e : s.upload(sh);
    }
};

// device copy of a BconvMfmaTable (numth.hpp); `digits` tables of equal shape back to back for the batched modup
struct BconvMfmaDev {
    DevBuf<uint8_t> table;
    DevBuf<uint64_t> consts;
    uint32_t ksteps = 0, quads = 0;
    size_t table_stride = 0, consts_stride = 0;   // bytes / words from one digit's table to the next
    hipError_t upload(const std::vector<BconvMfmaTable> &digits) {
        ksteps = 0;
        if (digits.empty() || !digits[0].ksteps) return hipSuccess;
        std::vector<uint8_t> t;
        std::vector<uint64_t> c;
        for (const auto &dg : digits) {
            if (dg.ksteps != digits[0].ksteps || dg.quads != digits[0].quads) return hipSuccess;   // no table: vector-ALU kernel
            t.insert(t.end(), dg.table.begin(), dg.table.end());
            c.insert(c.end(), dg.consts.begin(), dg.consts.end());
        }
        hipError_t e = table.upload(t);
        if (e == hipSuccess) e = consts.upload(c);
        if (e != hipSuccess) return e;
        ksteps = digits[0].ksteps;
        quads = digits[0].quads;
        table_stride = digits[0].table.size();
        consts_stride = digits[0].consts.size();
        return hipSuccess;
    }
};

// DRNSTool for one data level (src/rns.cu:11-192, CKKS subset)
struct LevelTool {
    size_t size_Ql = 0, beta = 0;
    LevelConstants host;
    ShoupVec inv_q_last, bigPInv, partHatInv, PHatInv;  // PHatInv indexed by P limb
    std::vector<DevBuf<uint64_t>> modup_matrix;          // per digit
    std::vector<DevBuf<uint32_t>> modup_obase_tidx;      // per digit: table index of each output limb
    DevBuf<uint64_t> moddown_matrix;
    DevBuf<uint32_t> moddown_obase_tidx;
    std::vector<BconvMfmaDev> modup_mfma;                // per digit: the same matrices as matrix-core operands
    BconvMfmaDev moddown_mfma, modup_mfma_all;
    // merged relinearise + rescale: division by B = {q_{l-1}} u P in one moddown (built on first use)
    struct MergedDown {
        bool ready = false;
        ShoupVec BHatInv;              // [alpha + 1], buffer order: q_{l-1}, then P
        DevBuf<uint64_t> matrix;       // [size_Ql - 1][alpha + 2]: BHat_b mod q_j, then (-prod B) mod q_j
        DevBuf<double> inv_b;          // [alpha + 1]: 1.0 / b
        DevBuf<uint32_t> obase_tidx;   // [size_Ql - 1]
        ShoupVec BInv, BInv2;          // [size_Ql - 1]: (prod B)^-1 mod q_j, and twice that
    } merged;
    bool uniform_digits = false;                         // alpha divides size_Ql: one launch for all digits
    DevBuf<uint64_t> modup_matrix_all;
    DevBuf<uint32_t> modup_tidx_all;
};

}  // namespace

struct pha_context {
    int log_n = 0;
    size_t n = 0, size_QP = 0, size_P = 0, size_Q = 0;
    NttPlan plan{};
    bool narrow = false;  // every modulus < 2^60: forward butterflies use the 16q lazy range
    bool fp64 = true;     // limbs with q < 2^50 run on the FP64 butterflies (PHA_DISABLE_FP64=1 turns it off)
    std::vector<uint64_t> moduli;
    std::vector<LimbTables> host_tables;
    DevBuf<DModulus> d_mod;
    DevBuf<tw_t> d_tw, d_itw, d_twD, d_itwD;
    DevBuf<double> d_twG, d_itwG;     // generators of the last pass's twiddles (FP64 limbs; numth.hpp build_generator_table)
    bool col8 = false;                // PHA_COL8=1: forward column pass with eight coefficients per lane (experiment)
    bool bcast_fused = true;          // PHA_BCAST_FUSED=0: the single-limb broadcast of rescale / alpha = 1 moddown as its own pass
    bool gen_tw = false;              // the row passes derive their last-pass twiddles (PHA_GEN_TW=0: read them from twD)
    DevBuf<uint64_t> d_n_inv, d_n_inv_shoup;
    uint32_t twD_stride = 0;
    std::map<size_t, std::unique_ptr<LevelTool>> levels;
    std::map<uint32_t, std::unique_ptr<DevBuf<uint32_t>>> galois_tables;
    DevBuf<uint64_t> workspace;
    hipStream_t ws_stream = nullptr;   // the stream of the last call that took the workspace (ensure_workspace)
    bool ws_used = false;
    hipEvent_t ws_event = nullptr;
    bool ws_event_valid = false;       // ws_event was recorded after the last workspace-using call
    bool ws_shared = false;            // more than one stream has used this context: calls record ws_event as they return
    DevBuf<uint64_t> d_bigP_mod_q;  // key level, [size_Q] (key generation)
    // helper stream for the independent giant-step chains of pha_linear_transform_double_hoisted
    static constexpr int kAuxStreams = 3;                // helper streams of the double-hoisted transform's giant chains
    hipStream_t aux_stream[kAuxStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t aux_fork = nullptr, aux_join[kAuxStreams] = {nullptr, nullptr, nullptr};
    // CKKS encoder: rotation group 5^j mod 2N and per-level CRT composition tables
    struct EncoderLevel {
        DevBuf<uint64_t> big_q, half_q, qhat, inv_qhat;
        int total_bits = 0;  // bit count of prod q_j (total_coeff_modulus_bit_count)
    };
    DevBuf<uint32_t> d_rot_group;
    std::map<size_t, std::unique_ptr<EncoderLevel>> enc_levels;

    int num_cus = 0;
    int device = -1;          // the device this context was created on
    bool bconv_mfma = true;   // base conversions on the matrix cores where a table exists (PHA_BCONV_MFMA=0: vector ALU)
    uint32_t xcd_map = 1;     // limb-to-XCD block placement of the NTT kernels (PHA_XCD_MAP=0: tile-interleaved, round 1)
    // single-launch pipelined transform (ntt_fused.cuh): PHA_FUSED_NTT / pha_ntt_fused_config
    int fused = 0;                    // 0: two kernels; 1: one persistent launch where it applies
    uint32_t fused_flags = 0;         // FUSED_* experiment switches
    uint32_t fused_min_units = 16;    // smaller launches keep the two kernels
    uint32_t fused_nq = 0;            // XCC ids the census found (queues)
    uint64_t fused_xcc_queue = ~0ull; // byte x: queue of XCC id x
    DevBuf<uint32_t> fused_ctl;

    NttTables fwd_tables() const {
        return NttTables{d_mod.p, d_tw.p, d_twD.p, d_n_inv.p, d_n_inv_shoup.p, static_cast<uint32_t>(log_n), twD_stride,
                         fp64 ? 1u : 0u, static_cast<uint32_t>(plan.split), gen_tw ? d_twG.p : nullptr};
    }
    NttTables inv_tables() const {
        return NttTables{d_mod.p, d_itw.p, d_itwD.p, d_n_inv.p, d_n_inv_shoup.p, static_cast<uint32_t>(log_n), twD_stride,
                         fp64 ? 1u : 0u, static_cast<uint32_t>(plan.split), gen_tw ? d_itwG.p : nullptr};
    }
    bool limb_is_fp64(size_t l) const { return fp64 && (moduli[l] >> 50) == 0; }
};

namespace {

// The context's one workspace, for a call on stream `st`.  Calls on one stream are ordered by the stream.  The first time
// a call arrives on another stream than the one before it the device is synchronised once and the context is marked as
// shared between streams; from then on every call that took the workspace records ws_event on its own stream when it
// returns (WsLease below), and a call on a different stream waits for that event before it touches t_mod_up / cx /
// scratch.  The old stream's handle is never used again, so the caller may destroy it.  Concurrent work needs one context
// per stream (ckks.py: GpuEvaluator.sibling).  Captured work is outside this tracking: a capturing stream cannot wait for
// eager work and a graph replay is invisible here, so replays of a context must not overlap other work on that context
// (INTEGRATION.md).  Growing the buffer synchronises the device and allocates: not possible while `st` is capturing --
// run the call once before the capture.
int ensure_workspace(pha_context *ctx, size_t words, uint64_t **out, hipStream_t st) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(st, &cap));
    if (ctx->workspace.count < words) {
        if (cap != hipStreamCaptureStatusNone)
            return fail(PHA_ERR_LOGIC, "the workspace must grow to %zu words while the stream is capturing: run this call once before "
                        "the capture", words);
        // the old buffer may still be in use by enqueued kernels
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(ctx->workspace.alloc(words));
        ctx->ws_event_valid = false;   // everything recorded so far has completed
    } else if (ctx->ws_used && ctx->ws_stream != st && cap == hipStreamCaptureStatusNone) {
        if (ctx->ws_event_valid) {
            HIP_TRY(hipStreamWaitEvent(st, ctx->ws_event, 0));
        } else if (!ctx->ws_shared) {
            HIP_TRY(hipDeviceSynchronize());   // once per context: the calls so far recorded nothing
        }
        ctx->ws_shared = true;
    }
    ctx->ws_stream = st;
    ctx->ws_used = true;
    *out = ctx->workspace.p;
    return PHA_OK;
}

// Marks the end of a call's use of the workspace: on a context that more than one stream has used, records ws_event
// on the call's stream so that the next call on another stream can wait for it.
struct WsLease {
    pha_context *ctx;
    hipStream_t st;
    WsLease(pha_context *c, hipStream_t s) : ctx(c), st(s) {}
    WsLease(const WsLease &) = delete;
    WsLease &operator=(const WsLease &) = delete;
    ~WsLease() {
        if (!ctx->ws_shared) return;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        ctx->ws_event_valid = false;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
        if (!ctx->ws_event && hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming) != hipSuccess) return;
        ctx->ws_event_valid = hipEventRecord(ctx->ws_event, st) == hipSuccess;
    }
};

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PHA_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return PHA_OK;
}

// Which XCC ids the workgroups of a grid on this device report (8 on an MI355X in SPX mode, 1 in CPX mode): the
// single-launch transform gives every id its own task queue, so the map must be the device's, not an assumption.
// Also allocates and zeroes the control block.  No id found (cannot happen) leaves the single launch off.
int fused_census(pha_context *ctx) {
    constexpr uint32_t kBlocks = 1024;
    DevBuf<uint32_t> ids;
    HIP_TRY(ids.alloc(kBlocks));
    HIP_TRY(hipMemset(ids.p, 0xff, kBlocks * sizeof(uint32_t)));
    hipLaunchKernelGGL(fused_census_kernel, dim3(kBlocks), dim3(64), 0, nullptr, ids.p);
    PHA_TRY(check_launch("fused_census"));
    std::vector<uint32_t> h(kBlocks);
    HIP_TRY(hipMemcpy(h.data(), ids.p, kBlocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    bool seen[8] = {};
    for (uint32_t v : h)
        if (v < 8) seen[v] = true;
    uint64_t map = 0;
    uint32_t nq = 0;
    for (int x = 0; x < 8; x++) map |= static_cast<uint64_t>(seen[x] ? nq++ : 0xffu) << (8 * x);
    ctx->fused_nq = nq;
    ctx->fused_xcc_queue = map;
    HIP_TRY(ctx->fused_ctl.alloc(kFCtlWords));
    HIP_TRY(hipMemset(ctx->fused_ctl.p, 0, kFCtlWords * sizeof(uint32_t)));
    return PHA_OK;
}

// ---- table upload -------------------------------------------------------------------------
void interleave(const std::vector<uint64_t> &w, const std::vector<uint64_t> &ws, tw_t *dst) {
    for (size_t i = 0; i < w.size(); i++) dst[i] = tw_t{w[i], ws[i]};
}

// last-pass order: [(ggid >> 6)][slot][ggid & 63], record = tw[((n >> rb) + ggid) << s | jb]
void last_pass_order(const tw_t *canon, size_t n, int rb, tw_t *dst) {
    const size_t groups = n >> rb, nslot = (size_t(1) << rb) - 1;
    for (size_t g = 0; g < groups; g++) {
        for (int s = 0; s < rb; s++) {
            for (size_t jb = 0; jb < (size_t(1) << s); jb++) {
                const size_t slot = (size_t(1) << s) - 1 + jb;
                dst[((g >> 6) * nslot + slot) * 64 + (g & 63)] = canon[(((n >> rb) + g) << s) + jb];
            }
        }
    }
}

// {w, w_shoup} records -> one double(w) per twiddle, same order, packed into the first half of
// the limb's region (limbs that run on FpField)
void to_fp64_records(tw_t *rec, size_t count) {
    std::vector<double> w(count);
    for (size_t i = 0; i < count; i++) w[i] = static_cast<double>(rec[i].x);
    memset(rec, 0, count * sizeof(tw_t));
    memcpy(rec, w.data(), count * sizeof(double));
}

int upload_tables(pha_context *ctx) {
    const size_t n = ctx->n, L = ctx->size_QP;
    const int rb = ctx->plan.s2 - ctx->plan.ra2;
    const size_t stride = ctx->plan.small ? 0 : (n >> rb) * ((size_t(1) << rb) - 1);   // no last-pass tables for the LDS kernels
    ctx->twD_stride = static_cast<uint32_t>(stride);
    std::vector<tw_t> tw(L * n), itw(L * n), twD(L * stride), itwD(L * stride);
    std::vector<DModulus> mods(L);
    std::vector<uint64_t> ninv(L), ninv_s(L);
    for (size_t l = 0; l < L; l++) {
        const LimbTables &t = ctx->host_tables[l];
        interleave(t.tw, t.tw_shoup, tw.data() + l * n);
        interleave(t.itw, t.itw_shoup, itw.data() + l * n);
        if (stride) {
            last_pass_order(tw.data() + l * n, n, rb, twD.data() + l * stride);
            last_pass_order(itw.data() + l * n, n, rb, itwD.data() + l * stride);
        }
        if (ctx->limb_is_fp64(l)) {
            to_fp64_records(tw.data() + l * n, n);
            to_fp64_records(itw.data() + l * n, n);
            to_fp64_records(twD.data() + l * stride, stride);
            to_fp64_records(itwD.data() + l * stride, stride);
        }
        mods[l] = DModulus{t.q, t.ratio[0], t.ratio[1]};
        ninv[l] = t.n_inv;
        ninv_s[l] = t.n_inv_shoup;
    }
    HIP_TRY(ctx->d_tw.upload(tw));
    HIP_TRY(ctx->d_itw.upload(itw));
    HIP_TRY(ctx->d_twD.upload(twD));
    HIP_TRY(ctx->d_itwD.upload(itwD));
    // generators of the last-pass twiddles: only where that pass has four stages on 4096-coefficient tiles and the limb
    // runs on FP64 (the integer butterflies need each twiddle's Shoup companion, a division per derived twiddle)
    if (rb == 4 && kTileLog == 12 && !ctx->plan.small) {
        static_assert(kGenWords == kGenStride, "device and host layouts of the generator table");
        std::vector<double> g(L * kGenStride, 0.0), ig(L * kGenStride, 0.0);
        std::vector<uint64_t> w(kGenStride);
        for (size_t l = 0; l < L; l++) {
            if (!ctx->limb_is_fp64(l)) continue;
            build_generator_table(ctx->host_tables[l], ctx->log_n, false, w.data());
            for (size_t i = 0; i < kGenStride; i++) g[l * kGenStride + i] = static_cast<double>(w[i]);
            build_generator_table(ctx->host_tables[l], ctx->log_n, true, w.data());
            for (size_t i = 0; i < kGenStride; i++) ig[l * kGenStride + i] = static_cast<double>(w[i]);
        }
        HIP_TRY(ctx->d_twG.upload(g));
        HIP_TRY(ctx->d_itwG.upload(ig));
        const char *e = getenv("PHA_GEN_TW");
        ctx->gen_tw = !(e && e[0] == '0');
    }
    {
        const char *c8 = getenv("PHA_COL8");
        ctx->col8 = c8 && c8[0] == '1' && kTileLog == 12;
    }
    {
        const char *e = getenv("PHA_BCAST_FUSED");
        ctx->bcast_fused = !(e && e[0] == '0');
    }
    HIP_TRY(ctx->d_mod.upload(mods));
    HIP_TRY(ctx->d_n_inv.upload(ninv));
    HIP_TRY(ctx->d_n_inv_shoup.upload(ninv_s));
    return PHA_OK;
}

// ---- NTT launchers ------------------------------------------------------------------------
struct NttRange {
    LimbMap lm;
    uint32_t active;  // limbs actually transformed
};

int make_range(const pha_context *ctx, size_t count, size_t start, size_t size_QP, size_t size_P, size_t excl_s,
               size_t excl_e, NttRange &r) {
    if (count == 0) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_modulus_size must be positive");
    if (size_P) {
        if (size_QP != ctx->size_QP || size_P > count || size_P > size_QP)
            return fail(PHA_ERR_INVALID_ARGUMENT, "special modulus range does not match the context");
        if (start + count - size_P > size_QP - size_P)
            return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    } else if (start + count > ctx->size_QP) {
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    }
    uint32_t len = 0;
    if (excl_e > excl_s) {
        if (excl_s < start || excl_e > start + count)
            return fail(PHA_ERR_INVALID_ARGUMENT, "Excluded range in NTT is invalid.");
        len = static_cast<uint32_t>(excl_e - excl_s);
    } else {
        excl_s = excl_e = 0;
    }
    r.active = static_cast<uint32_t>(count) - len;
    r.lm = LimbMap{static_cast<uint32_t>(start), static_cast<uint32_t>(count), static_cast<uint32_t>(size_QP),
                   static_cast<uint32_t>(size_P), static_cast<uint32_t>(excl_s), static_cast<uint32_t>(excl_e),
                   r.active, 1u, 0u, 0u, 0u, 0u};
    return PHA_OK;
}

template <int S, int RA>
void launch_fcol(bool narrow, dim3 grid, hipStream_t st, uint64_t *data, size_t ps, const NttTables &T, const LimbMap &lm) {
    if (lm.bcast_src) {
        if (narrow)
            hipLaunchKernelGGL((fntt_col_kernel<S, RA, true, true>), grid, dim3(kThreads), 0, st, data, ps, T, lm);
        else
            hipLaunchKernelGGL((fntt_col_kernel<S, RA, false, true>), grid, dim3(kThreads), 0, st, data, ps, T, lm);
        return;
    }
    if (narrow)
        hipLaunchKernelGGL((fntt_col_kernel<S, RA, true>), grid, dim3(kThreads), 0, st, data, ps, T, lm);
    else
        hipLaunchKernelGGL((fntt_col_kernel<S, RA, false>), grid, dim3(kThreads), 0, st, data, ps, T, lm);
}
template <int S, int RA, bool NARROW>
void launch_frow_n(int epi, dim3 grid, hipStream_t st, uint64_t *data, size_t ps, const NttTables &T, const LimbMap &lm,
                   const Epilogue &ep) {
    switch (epi) {
        case EPI_SUB_MUL:
            hipLaunchKernelGGL((fntt_row_kernel<S, RA, EPI_SUB_MUL, NARROW>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
            break;
        case EPI_SUB_MUL_ADD:
            hipLaunchKernelGGL((fntt_row_kernel<S, RA, EPI_SUB_MUL_ADD, NARROW>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
            break;
        default:
            hipLaunchKernelGGL((fntt_row_kernel<S, RA, EPI_NONE, NARROW>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
    }
}
template <int S, int RA>
void launch_frow(bool narrow, int epi, dim3 grid, hipStream_t st, uint64_t *data, size_t ps, const NttTables &T,
                 const LimbMap &lm, const Epilogue &ep) {
    if (narrow)
        launch_frow_n<S, RA, true>(epi, grid, st, data, ps, T, lm, ep);
    else
        launch_frow_n<S, RA, false>(epi, grid, st, data, ps, T, lm, ep);
}
template <int S, int RA>
void launch_irow(dim3 grid, hipStream_t st, uint64_t *out, const uint64_t *in, size_t ps, const NttTables &T,
                 const LimbMap &lm) {
    hipLaunchKernelGGL((intt_row_kernel<S, RA>), grid, dim3(kThreads), 0, st, out, in, ps, T, lm);
}
template <int S, int RA>
void launch_icol(int epi, dim3 grid, hipStream_t st, uint64_t *data, size_t ps, const NttTables &T, const LimbMap &lm,
                 const Epilogue &ep) {
    if (T.split) {
        // subtrees only (instantiated for the N = 2^17 plan); N^-1 and the epilogue follow in split_inv_kernel
        hipLaunchKernelGGL((intt_col_kernel<8, 4, EPI_NONE, true>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
        const dim3 g2(lm.xcd_map ? xcd_grid_blocks(lm.active, lm.polys, (1u << T.log_n) >> 13) : grid.x >> (13 - kTileLog));
        if (epi == EPI_SCALE)
            hipLaunchKernelGGL((split_inv_kernel<EPI_SCALE>), g2, dim3(256), 0, st, data, ps, T, lm, ep);
        else
            hipLaunchKernelGGL((split_inv_kernel<EPI_NONE>), g2, dim3(256), 0, st, data, ps, T, lm, ep);
        return;
    }
    if (epi == EPI_SCALE)
        hipLaunchKernelGGL((intt_col_kernel<S, RA, EPI_SCALE>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
    else
        hipLaunchKernelGGL((intt_col_kernel<S, RA, EPI_NONE>), grid, dim3(kThreads), 0, st, data, ps, T, lm, ep);
}

#define DISPATCH_S(S, RA, CALL6, CALL7, CALL8) \
    switch (S) {                               \
        case 6: CALL6; break;                  \
        case 7: CALL7; break;                  \
        default: CALL8; break;                 \
    }

// limb-to-XCD placement needs the launch inside the range of decode_block's multiply-by-reciprocal division
bool use_xcd_map(const pha_context *ctx, uint32_t active, size_t polys, size_t tiles) {
    return ctx->xcd_map && polys <= 2048 && static_cast<uint64_t>(active + 8) * polys * tiles < (1ull << 22);
}
uint32_t ntt_grid_blocks(bool xcd, uint32_t active, size_t polys, size_t tiles) {
    const uint32_t p = static_cast<uint32_t>(polys), t = static_cast<uint32_t>(tiles);
    return xcd ? xcd_grid_blocks(active, p, t) : active * p * t;
}
void set_placement(LimbMap &lm, bool xcd, size_t polys) {
    lm.polys = static_cast<uint32_t>(polys);
    lm.xcd_map = xcd ? 1u : 0u;
    lm.polys_magic = static_cast<uint32_t>(((1ull << 32) + polys - 1) / polys);
}

int ntt_forward(pha_context *ctx, uint64_t *data, size_t polys, size_t poly_stride, const NttRange &r, int epi,
                const Epilogue &ep, hipStream_t st, int phases = 3) {
    if (!r.active || !polys) return PHA_OK;
    const NttTables T = ctx->fwd_tables();
    NttRange rr = r;
    if (ctx->plan.small) {
        if (!(phases & 1)) return PHA_OK;   // one kernel does the whole transform
        set_placement(rr.lm, false, polys);
        const dim3 g(static_cast<uint32_t>(r.active * polys));
        if (epi == EPI_SUB_MUL)
            hipLaunchKernelGGL((small_fntt_kernel<EPI_SUB_MUL>), g, dim3(256), 0, st, data, poly_stride, T, rr.lm, ep);
        else if (epi == EPI_SUB_MUL_ADD)
            hipLaunchKernelGGL((small_fntt_kernel<EPI_SUB_MUL_ADD>), g, dim3(256), 0, st, data, poly_stride, T, rr.lm, ep);
        else
            hipLaunchKernelGGL((small_fntt_kernel<EPI_NONE>), g, dim3(256), 0, st, data, poly_stride, T, rr.lm, ep);
        return check_launch("forward NTT (N = 2^11)");
    }
    if (ctx->fused && phases == 3 && epi == EPI_NONE && !T.split && ctx->plan.s1 == 8 && ctx->plan.s2 == 8 && ctx->fused_nq &&
        !r.lm.bcast_src) {
        const size_t tiles = ctx->n >> kTileLog, units = static_cast<size_t>(r.active) * polys;
        if (units >= ctx->fused_min_units && units <= kFusedMaxUnits) {
            set_placement(rr.lm, false, polys);
            const size_t tasks = 2 * units * tiles;
            // a multiple of the queue count, so that round-robin dispatch gives every XCD the same number of workgroups
            size_t wgs = std::min<size_t>(tasks, static_cast<size_t>(ctx->num_cus) * kFusedWgsPerCu);
            wgs = std::max<size_t>(ctx->fused_nq, wgs / ctx->fused_nq * ctx->fused_nq);
            const FusedArgs fa{ctx->fused_ctl.p, ctx->fused_nq, static_cast<uint32_t>(std::max<size_t>(1, wgs / ctx->fused_nq / 2)),
                               ctx->fused_xcc_queue, ctx->fused_flags};
            const dim3 grid(static_cast<uint32_t>(wgs));
            if (ctx->narrow)
                hipLaunchKernelGGL(fntt_fused_kernel<true>, grid, dim3(kThreads), 0, st, data, poly_stride, T, rr.lm, fa);
            else
                hipLaunchKernelGGL(fntt_fused_kernel<false>, grid, dim3(kThreads), 0, st, data, poly_stride, T, rr.lm, fa);
            return check_launch("forward NTT (single launch)");
        }
    }
    const bool xcd = use_xcd_map(ctx, r.active, polys, ctx->n >> kTileLog);
    set_placement(rr.lm, xcd, polys);
    const dim3 grid(ntt_grid_blocks(xcd, r.active, polys, ctx->n >> kTileLog));
    if ((phases & 1) && T.split)
        hipLaunchKernelGGL(split_fwd_kernel, dim3(ntt_grid_blocks(xcd, r.active, polys, ctx->n >> 13)), dim3(256), 0, st,
                           data, poly_stride, T, rr.lm);
    if ((phases & 1) && ctx->col8 && ctx->plan.s1 == 8) {
        if (ctx->narrow)
            hipLaunchKernelGGL(fntt_col8_kernel<true>, grid, dim3(kThreads8), 0, st, data, poly_stride, T, rr.lm);
        else
            hipLaunchKernelGGL(fntt_col8_kernel<false>, grid, dim3(kThreads8), 0, st, data, poly_stride, T, rr.lm);
    } else if (phases & 1)
    DISPATCH_S(ctx->plan.s1, 0, (launch_fcol<6, 3>(ctx->narrow, grid, st, data, poly_stride, T, rr.lm)),
               (launch_fcol<7, 4>(ctx->narrow, grid, st, data, poly_stride, T, rr.lm)),
               (launch_fcol<8, 4>(ctx->narrow, grid, st, data, poly_stride, T, rr.lm)));
    if (phases & 2)
    DISPATCH_S(ctx->plan.s2, 0, (launch_frow<6, 3>(ctx->narrow, epi, grid, st, data, poly_stride, T, rr.lm, ep)),
               (launch_frow<7, 4>(ctx->narrow, epi, grid, st, data, poly_stride, T, rr.lm, ep)),
               (launch_frow<8, 4>(ctx->narrow, epi, grid, st, data, poly_stride, T, rr.lm, ep)));
    return check_launch("forward NTT");
}

// data[p][j] = NTT(src[p] mod q_j) (+ epilogue) for the limbs j of `r`, src = ONE limb per poly in coefficient form,
// src_stride words apart: the broadcast of rescale_to_next and of the alpha = 1 moddown (src/rns.cu:1125-1136,
// src/rns_bconv.cu:691-707) followed by the forward transform.  The column kernel reduces the source on load
// (fntt_col_kernel<.., BCAST>), so the broadcast costs no pass through memory; where that kernel does not run (N = 2^11,
// N = 2^17, the eight-per-lane column pass, PHA_BCAST_FUSED=0) reduce_single_limb_kernel writes the residues first.
bool bcast_in_column_pass(const pha_context *ctx) {
    return ctx->bcast_fused && !ctx->plan.small && !ctx->fwd_tables().split && !(ctx->col8 && ctx->plan.s1 == 8);
}
int ntt_forward_bcast(pha_context *ctx, uint64_t *data, size_t polys, size_t poly_stride, const NttRange &r, int epi,
                      const Epilogue &ep, hipStream_t st, const uint64_t *src, size_t src_stride) {
    if (!r.active || !polys) return PHA_OK;
    // (the fallback below indexes the moduli by buffer limb: no special-modulus remap, no excluded range)
    if (bcast_in_column_pass(ctx) && r.lm.excl_end <= r.lm.excl_start && !r.lm.size_P) {
        NttRange rr = r;
        rr.lm.bcast_src = src;
        rr.lm.bcast_poly_stride = src_stride;
        return ntt_forward(ctx, data, polys, poly_stride, rr, epi, ep, st);
    }
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    hipLaunchKernelGGL(reduce_single_limb_kernel, dim3(n / 512, static_cast<uint32_t>(polys)), dim3(256), 0, st, data, src,
                       ctx->d_mod.p, r.lm.start, r.lm.start + r.lm.count, n, poly_stride, src_stride);
    PHA_TRY(check_launch("reduce_single_limb"));
    return ntt_forward(ctx, data, polys, poly_stride, r, epi, ep, st);
}

// out_limb0 / out_poly_stride: see LimbMap (where the transformed limbs land in `out`; default: where they were in `in`)
int ntt_backward(pha_context *ctx, uint64_t *out, const uint64_t *in, size_t polys, size_t poly_stride,
                 const NttRange &r, int epi, const Epilogue &ep, hipStream_t st, int phases = 3, size_t out_limb0 = 0,
                 size_t out_poly_stride = 0) {
    if (!r.active || !polys) return PHA_OK;
    const NttTables T = ctx->inv_tables();
    NttRange rr = r;
    rr.lm.out_limb0 = static_cast<uint32_t>(out_limb0);
    rr.lm.out_poly_stride = out_poly_stride;
    if (ctx->plan.small) {
        if (!(phases & 1)) return PHA_OK;
        set_placement(rr.lm, false, polys);
        const dim3 g(static_cast<uint32_t>(r.active * polys));
        if (epi == EPI_SCALE)
            hipLaunchKernelGGL((small_intt_kernel<EPI_SCALE>), g, dim3(256), 0, st, out, in, poly_stride, T, rr.lm, ep);
        else
            hipLaunchKernelGGL((small_intt_kernel<EPI_NONE>), g, dim3(256), 0, st, out, in, poly_stride, T, rr.lm, ep);
        return check_launch("inverse NTT (N = 2^11)");
    }
    const bool xcd = use_xcd_map(ctx, r.active, polys, ctx->n >> kTileLog);
    set_placement(rr.lm, xcd, polys);
    const dim3 grid(ntt_grid_blocks(xcd, r.active, polys, ctx->n >> kTileLog));
    if (phases & 2)
    DISPATCH_S(ctx->plan.s2, 0, (launch_irow<6, 3>(grid, st, out, in, poly_stride, T, rr.lm)),
               (launch_irow<7, 4>(grid, st, out, in, poly_stride, T, rr.lm)),
               (launch_irow<8, 4>(grid, st, out, in, poly_stride, T, rr.lm)));
    if (phases & 1)
    DISPATCH_S(ctx->plan.s1, 0, (launch_icol<6, 3>(epi, grid, st, out, poly_stride, T, rr.lm, ep)),
               (launch_icol<7, 4>(epi, grid, st, out, poly_stride, T, rr.lm, ep)),
               (launch_icol<8, 4>(epi, grid, st, out, poly_stride, T, rr.lm, ep)));
    return check_launch("inverse NTT");
}

// ---- level tools ---------------------------------------------------------------------------
int get_level(pha_context *ctx, size_t chain_index, LevelTool **out) {
    if (chain_index < 1 || chain_index > ctx->size_Q)
        return fail(PHA_ERR_INVALID_ARGUMENT, "chain_index %zu out of range [1, %zu]", chain_index, ctx->size_Q);
    if (ctx->size_P == 0) return fail(PHA_ERR_LOGIC, "context has no special modulus");
    const size_t size_Ql = ctx->size_Q - (chain_index - 1);
    auto it = ctx->levels.find(size_Ql);
    if (it != ctx->levels.end()) {
        *out = it->second.get();
        return PHA_OK;
    }
    auto lt = std::make_unique<LevelTool>();
    lt->size_Ql = size_Ql;
    build_level_constants(ctx->moduli, ctx->size_P, size_Ql, lt->host);
    lt->beta = lt->host.beta;
    const uint64_t *Q = ctx->moduli.data();
    const uint64_t *P = Q + ctx->size_Q;
    if (size_Ql > 1) HIP_TRY(lt->inv_q_last.upload(lt->host.inv_q_last_mod_q, Q));
    HIP_TRY(lt->bigPInv.upload(lt->host.bigPInv_mod_q, Q));
    HIP_TRY(lt->partHatInv.upload(lt->host.partQlHatInv_mod_Ql_concat, Q));
    HIP_TRY(lt->PHatInv.upload(lt->host.PHatInv_mod_p, P));
    lt->modup_matrix.resize(lt->beta);
    lt->modup_obase_tidx.resize(lt->beta);
    lt->modup_mfma.resize(lt->beta);
    std::vector<BconvMfmaTable> digit_tables(lt->beta);
    for (size_t d = 0; d < lt->beta; d++) {
        const size_t s = d * ctx->size_P, e = std::min(s + ctx->size_P, size_Ql);
        std::vector<uint32_t> tidx;
        for (size_t j = 0; j < size_Ql + ctx->size_P; j++) {
            if (j >= s && j < e) continue;
            tidx.push_back(static_cast<uint32_t>(j < size_Ql ? j : ctx->size_Q + (j - size_Ql)));
        }
        HIP_TRY(lt->modup_matrix[d].upload(lt->host.modup_matrix[d]));
        HIP_TRY(lt->modup_obase_tidx[d].upload(tidx));
        if (ctx->size_P > 1) {
            std::vector<uint64_t> omod(tidx.size());
            for (size_t j = 0; j < tidx.size(); j++) omod[j] = ctx->moduli[tidx[j]];
            digit_tables[d] = BconvMfmaTable{};
            build_bconv_mfma_table(lt->host.modup_matrix[d].data(), omod.data(), e - s, omod.size(), digit_tables[d]);
            HIP_TRY(lt->modup_mfma[d].upload({digit_tables[d]}));
        }
    }
    lt->uniform_digits = ctx->size_P > 1 && size_Ql % ctx->size_P == 0;
    if (lt->uniform_digits) {
        std::vector<uint64_t> mall;
        std::vector<uint32_t> tall;
        for (size_t d = 0; d < lt->beta; d++) {
            const size_t s = d * ctx->size_P, e = s + ctx->size_P;
            mall.insert(mall.end(), lt->host.modup_matrix[d].begin(), lt->host.modup_matrix[d].end());
            for (size_t j = 0; j < size_Ql + ctx->size_P; j++) {
                if (j >= s && j < e) continue;
                tall.push_back(static_cast<uint32_t>(j < size_Ql ? j : ctx->size_Q + (j - size_Ql)));
            }
        }
        HIP_TRY(lt->modup_matrix_all.upload(mall));
        HIP_TRY(lt->modup_tidx_all.upload(tall));
        HIP_TRY(lt->modup_mfma_all.upload(digit_tables));
    }
    std::vector<uint32_t> tidx(size_Ql);
    for (size_t j = 0; j < size_Ql; j++) tidx[j] = static_cast<uint32_t>(j);
    HIP_TRY(lt->moddown_matrix.upload(lt->host.moddown_matrix));
    HIP_TRY(lt->moddown_obase_tidx.upload(tidx));
    if (ctx->size_P > 1) {
        BconvMfmaTable t;
        build_bconv_mfma_table(lt->host.moddown_matrix.data(), ctx->moduli.data(), ctx->size_P, size_Ql, t);
        HIP_TRY(lt->moddown_mfma.upload({t}));
    }
    *out = lt.get();
    ctx->levels[size_Ql] = std::move(lt);
    return PHA_OK;
}

// mf: the matrix as matrix-core operands (nullptr or empty: none).  That path needs 16-byte aligned buffers and N >= 256.
int launch_bconv(pha_context *ctx, uint64_t *dst, const uint64_t *y, const uint64_t *matrix, const uint32_t *obase_tidx,
                 uint32_t ibase, uint32_t obase, uint32_t skip_start, uint32_t skip_len, uint32_t batch,
                 const BconvBatch &bt, hipStream_t st, const BconvMfmaDev *mf = nullptr) {
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    const bool use_mfma = ctx->bconv_mfma;
    auto aligned16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (use_mfma && mf && mf->ksteps && n >= 256 && mf->quads == (obase + 3) / 4 && aligned16(dst) && aligned16(y) &&
        aligned16(bt.copy_src)) {
        // every group of quads re-reads the inputs and stages its operands in LDS (at most 60 KiB of it)
        static const uint32_t target_wgs = [] {
            const char *e = getenv("PHA_BCONV_MFMA_WGS");
            return e ? static_cast<uint32_t>(atoi(e)) : 1024u;
        }();
        static const uint32_t pairs_env = [] {
            const char *e = getenv("PHA_BCONV_MFMA_PAIRS");
            return e ? static_cast<uint32_t>(atoi(e)) : 0u;
        }();
        const uint32_t quad_bytes = 2 * mf->ksteps * 1024 + 64, max_qpb = (60u << 10) / quad_bytes;
        // 256 coefficients per workgroup (128 measured slower at configs[2] even where it doubles the workgroups in flight)
        const uint32_t pairs = pairs_env == 1 ? 1u : 2u;
        const uint32_t coeffs = 128 * pairs, base = (n / coeffs) * batch;
        uint32_t groups = std::min<uint32_t>(mf->quads, std::max<uint32_t>(1, target_wgs / base));
        uint32_t qpb = std::min<uint32_t>((mf->quads + groups - 1) / groups, max_qpb);
        groups = (mf->quads + qpb - 1) / qpb;
        const bool xcd_grid = groups > 1 && base % 8 == 0;   // see the kernel: quad groups of a block share an XCD
        const dim3 g = xcd_grid ? dim3(base * groups) : dim3(n / coeffs, groups, batch);
        const size_t lds = static_cast<size_t>(qpb) * quad_bytes;
        // z runs over digits (each with its own matrix) only when the vector-ALU kernel's matrix advances with it
        const size_t tstride = bt.matrix_stride ? mf->table_stride : 0, cstride = bt.matrix_stride ? mf->consts_stride : 0;
#define BM_LAUNCH(KS, PR)                                                                                                     \
    hipLaunchKernelGGL((bconv_mfma_kernel<KS, PR>), g, dim3(256), lds, st, dst, y, mf->table.p, mf->consts.p, tstride, cstride, \
                       ibase, obase, qpb, groups, n, skip_start, skip_len, bt)
        if (mf->ksteps == 1) {
            if (pairs == 1) BM_LAUNCH(1, 1); else BM_LAUNCH(1, 2);
        } else {
            if (pairs == 1) BM_LAUNCH(2, 1); else BM_LAUNCH(2, 2);
        }
#undef BM_LAUNCH
        return check_launch("bconv_mfma");
    }
    // Every output-limb group re-reads the ibase input limbs: as few groups as still give ~4096 workgroups
    // (measured at configs[2]: targets 512..9216 give HMult 0.343 / 0.331 / 0.332 / 0.327 / 0.336 ms; a batch of 16
    // rotations needs one group and gains 2.5 %); PHA_BCONV_WGS overrides the target for experiments.
    static const uint32_t target = [] {
        const char *e = getenv("PHA_BCONV_WGS");
        return e ? static_cast<uint32_t>(atoi(e)) : 4096u;
    }();
    const uint32_t base_wgs = (n / 256) * batch;
    const uint32_t groups = std::min<uint32_t>(obase, std::max<uint32_t>(1, (target + base_wgs - 1) / base_wgs));
    const uint32_t lpb = (obase + groups - 1) / groups;
    const dim3 grid(n / 256, (obase + lpb - 1) / lpb, batch);
    const DModulus *mod = ctx->d_mod.p;
#define BCONV_CASE(IB)                                                                                          \
    case IB:                                                                                                    \
        hipLaunchKernelGGL((bconv_matmul_kernel<IB>), grid, dim3(256), 0, st, dst, y, matrix, mod, obase_tidx, \
                           obase, lpb, n, skip_start, skip_len, bt);                                           \
        break;
    switch (ibase) {
        BCONV_CASE(1) BCONV_CASE(2) BCONV_CASE(3) BCONV_CASE(4) BCONV_CASE(5) BCONV_CASE(6) BCONV_CASE(7) BCONV_CASE(8)
        BCONV_CASE(9) BCONV_CASE(10) BCONV_CASE(11) BCONV_CASE(12) BCONV_CASE(13) BCONV_CASE(14) BCONV_CASE(15)
        BCONV_CASE(16)
        default:
            hipLaunchKernelGGL(bconv_matmul_generic_kernel, grid, dim3(256), 0, st, dst, y, matrix, mod, obase_tidx,
                               ibase, obase, lpb, n, skip_start, skip_len, bt);
    }
#undef BCONV_CASE
    return check_launch("bconv_matmul");
}

// DRNSTool::modup (src/rns_bconv.cu:530-628), t_cks = scratch [size_Ql][n]
// copy_own = false leaves the digits' own slots of dst unwritten: the caller's inner product reads those limbs from cks
int do_modup(pha_context *ctx, LevelTool *lt, uint64_t *dst, const uint64_t *cks, uint64_t *t_cks, hipStream_t st,
             bool copy_own = true) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha;
    NttRange r;
    PHA_TRY(make_range(ctx, size_Ql, 0, 0, 0, 0, 0, r));
    Epilogue ep{};
    if (alpha == 1) {
        PHA_TRY(ntt_backward(ctx, t_cks, cks, 1, 0, r, EPI_NONE, ep, st));
        if (bcast_in_column_pass(ctx)) {
            // every digit is ONE limb: digit d's other limbs are limb d of t_cks modulo their primes, which the forward
            // column kernel forms on load (poly d reads t_cks + d * n); only the own limbs are copied, if asked for
            if (copy_own)
                HIP_TRY(hipMemcpy2DAsync(dst, (size_QlP + 1) * n * sizeof(uint64_t), cks, n * sizeof(uint64_t),
                                         n * sizeof(uint64_t), size_Ql, hipMemcpyDeviceToDevice, st));
            NttRange rb;
            PHA_TRY(make_range(ctx, size_QlP, 0, ctx->size_QP, alpha, 0, 0, rb));
            rb.lm.digit_alpha = 1u;
            rb.lm.digit_limit = static_cast<uint32_t>(size_Ql);
            rb.active = rb.lm.active = static_cast<uint32_t>(size_QlP - 1);
            rb.lm.bcast_src = t_cks;
            rb.lm.bcast_poly_stride = n;
            return ntt_forward(ctx, dst, lt->beta, size_QlP * n, rb, EPI_NONE, Epilogue{}, st);
        }
        hipLaunchKernelGGL(modup_single_p_kernel, dim3(static_cast<uint32_t>(n / 256), static_cast<uint32_t>(lt->beta)),
                           dim3(256), 0, st, dst, cks, t_cks, ctx->d_mod.p, static_cast<uint32_t>(size_Ql),
                           static_cast<uint32_t>(ctx->size_Q), static_cast<uint32_t>(n), 0u,
                           static_cast<uint32_t>(size_QlP), CoeffRows{});
        PHA_TRY(check_launch("modup_single_p"));
    } else {
        ep.cst = lt->partHatInv.v.p;
        ep.cst_shoup = lt->partHatInv.s.p;
        PHA_TRY(ntt_backward(ctx, t_cks, cks, 1, 0, r, EPI_SCALE, ep, st));
        if (lt->uniform_digits) {
            const uint32_t obase = static_cast<uint32_t>(size_QlP - alpha);
            BconvBatch bt{size_QlP * n, alpha * n, static_cast<size_t>(obase) * alpha, obase,
                          static_cast<uint32_t>(alpha), copy_own ? cks : nullptr, alpha * n, 0, 0, 0};
            PHA_TRY(launch_bconv(ctx, dst, t_cks, lt->modup_matrix_all.p, lt->modup_tidx_all.p,
                                 static_cast<uint32_t>(alpha), obase, 0, static_cast<uint32_t>(alpha),
                                 static_cast<uint32_t>(lt->beta), bt, st, &lt->modup_mfma_all));
        } else {
            for (size_t d = 0; d < lt->beta; d++) {
                const size_t s = d * alpha, e = std::min(s + alpha, size_Ql), len = e - s;
                BconvBatch bt{0, 0, 0, 0, 0, copy_own ? cks + s * n : nullptr, 0, 0, 0, 0};
                PHA_TRY(launch_bconv(ctx, dst + d * size_QlP * n, t_cks + s * n, lt->modup_matrix[d].p,
                                     lt->modup_obase_tidx[d].p, static_cast<uint32_t>(len),
                                     static_cast<uint32_t>(size_QlP - len), static_cast<uint32_t>(s),
                                     static_cast<uint32_t>(len), 1, bt, st, &lt->modup_mfma[d]));
            }
        }
    }
    // one launch pair transforms every digit, each skipping its own (already NTT-form) limbs
    NttRange rr;
    PHA_TRY(make_range(ctx, size_QlP, 0, ctx->size_QP, alpha, 0, 0, rr));
    rr.lm.digit_alpha = static_cast<uint32_t>(alpha);
    rr.lm.digit_limit = static_cast<uint32_t>(size_Ql);
    rr.active = rr.lm.active = static_cast<uint32_t>(size_QlP - std::min(alpha, size_Ql - (lt->beta - 1) * alpha));
    PHA_TRY(ntt_forward(ctx, dst, lt->beta, size_QlP * n, rr, EPI_NONE, Epilogue{}, st));
    return PHA_OK;
}

// own: the digits' own limbs (c2 in NTT form) when the modup did not copy them into t_mod_up
int do_inner_prod(pha_context *ctx, LevelTool *lt, uint64_t *cx, const uint64_t *t_mod_up, const uint64_t *const *evks,
                  hipStream_t st, const uint32_t *perm = nullptr, const uint64_t *own = nullptr) {
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    const uint32_t size_QlP = static_cast<uint32_t>(lt->size_Ql + ctx->size_P);
    hipLaunchKernelGGL(key_switch_inner_prod_kernel, dim3(size_QlP * (n / 512)), dim3(256), 0, st, cx, t_mod_up, evks,
                       ctx->d_mod.p, n, static_cast<uint32_t>(ctx->size_QP), size_QlP,
                       static_cast<uint32_t>(ctx->size_Q), static_cast<uint32_t>(lt->size_Ql),
                       static_cast<uint32_t>(lt->beta), 0u, perm, own, static_cast<uint32_t>(ctx->size_P));
    return check_launch("key_switch_inner_prod");
}

// modup of `batch` ciphertexts' c2 at once (alpha > 1; uniform digits, or a single digit of size_Ql <= alpha limbs):
// cks / t_cks hold the b-th polynomial at b * in_stride, dst is [batch][beta][size_QlP][n]
int do_modup_batched(pha_context *ctx, LevelTool *lt, uint64_t *dst, const uint64_t *cks, uint64_t *t_cks, size_t in_stride,
                     size_t batch, hipStream_t st) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha;
    const bool single = !lt->uniform_digits;            // beta == 1: the digit is all of Q_l
    const size_t len = single ? size_Ql : alpha;
    NttRange r;
    PHA_TRY(make_range(ctx, size_Ql, 0, 0, 0, 0, 0, r));
    Epilogue ep{};
    ep.cst = lt->partHatInv.v.p;
    ep.cst_shoup = lt->partHatInv.s.p;
    PHA_TRY(ntt_backward(ctx, t_cks, cks, batch, in_stride, r, EPI_SCALE, ep, st));
    const uint32_t obase = static_cast<uint32_t>(size_QlP - len);
    BconvBatch bt{size_QlP * n, len * n, static_cast<size_t>(obase) * len, obase, static_cast<uint32_t>(len), cks, len * n,
                  static_cast<uint32_t>(lt->beta), in_stride, in_stride};
    PHA_TRY(launch_bconv(ctx, dst, t_cks, single ? lt->modup_matrix[0].p : lt->modup_matrix_all.p,
                         single ? lt->modup_obase_tidx[0].p : lt->modup_tidx_all.p, static_cast<uint32_t>(len), obase, 0,
                         static_cast<uint32_t>(len), static_cast<uint32_t>(lt->beta * batch), bt, st,
                         single ? &lt->modup_mfma[0] : &lt->modup_mfma_all));
    NttRange rr;
    PHA_TRY(make_range(ctx, size_QlP, 0, ctx->size_QP, alpha, 0, 0, rr));
    rr.lm.digit_alpha = static_cast<uint32_t>(alpha);
    rr.lm.digit_limit = static_cast<uint32_t>(size_Ql);
    rr.lm.digit_count = static_cast<uint32_t>(lt->beta);
    rr.active = rr.lm.active = static_cast<uint32_t>(size_QlP - len);
    return ntt_forward(ctx, dst, lt->beta * batch, size_QlP * n, rr, EPI_NONE, Epilogue{}, st);
}

int do_inner_prod_batched(pha_context *ctx, LevelTool *lt, uint64_t *cx, const uint64_t *t_mod_up, const uint64_t *const *evks,
                          size_t batch, hipStream_t st) {
    constexpr int BPW = 4;
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    const uint32_t size_QlP = static_cast<uint32_t>(lt->size_Ql + ctx->size_P);
    const uint32_t groups = static_cast<uint32_t>((batch + BPW - 1) / BPW);
    hipLaunchKernelGGL((key_switch_inner_prod_batched_kernel<BPW>), dim3(size_QlP * (n / 512) * groups), dim3(256), 0, st, cx,
                       t_mod_up, evks, ctx->d_mod.p, n, static_cast<uint32_t>(ctx->size_QP), size_QlP,
                       static_cast<uint32_t>(ctx->size_Q), static_cast<uint32_t>(lt->size_Ql), static_cast<uint32_t>(lt->beta),
                       static_cast<uint32_t>(batch));
    return check_launch("key_switch_inner_prod_batched");
}

// DRNSTool::moddown_from_NTT (src/rns_bconv.cu:776-828) for `polys` buffers at once
// (cx stride size_QlP*n, ct stride size_Ql*n), delta = scratch [polys][size_Ql][n].
// addend != nullptr fuses add_to_ct: ct = addend + (cx - NTT(delta)) * PInv
int do_moddown(pha_context *ctx, LevelTool *lt, uint64_t *ct, uint64_t *cx, uint64_t *delta, const uint64_t *addend,
               size_t polys, hipStream_t st, uint32_t addend_every = 0) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha;
    NttRange rp;
    PHA_TRY(make_range(ctx, alpha, size_Ql, ctx->size_QP, alpha, 0, 0, rp));
    uint64_t *cp = cx + size_Ql * n;
    if (alpha == 1) {
        PHA_TRY(ntt_backward(ctx, cx, cx, polys, size_QlP * n, rp, EPI_NONE, Epilogue{}, st));
    } else {
        // INTT fused with bconv phase 1 (x PHatInv mod p_k); constants indexed by buffer limb
        Epilogue ep{};
        ep.cst = lt->PHatInv.v.p;
        ep.cst_shoup = lt->PHatInv.s.p;
        ep.cst_limb0 = static_cast<uint32_t>(size_Ql);
        PHA_TRY(ntt_backward(ctx, cx, cx, polys, size_QlP * n, rp, EPI_SCALE, ep, st));
        BconvBatch bt{size_Ql * n, size_QlP * n, 0, 0, 0, nullptr, 0, 0, 0, 0};
        PHA_TRY(launch_bconv(ctx, delta, cp, lt->moddown_matrix.p, lt->moddown_obase_tidx.p, static_cast<uint32_t>(alpha),
                             static_cast<uint32_t>(size_Ql), static_cast<uint32_t>(size_Ql), 0,
                             static_cast<uint32_t>(polys), bt, st, &lt->moddown_mfma));
    }
    NttRange rq;
    PHA_TRY(make_range(ctx, size_Ql, 0, 0, 0, 0, 0, rq));
    Epilogue ep{};
    ep.out = ct;
    ep.sub = cx;
    ep.addend = addend;
    ep.addend_every = addend_every;
    ep.cst = lt->bigPInv.v.p;
    ep.cst_shoup = lt->bigPInv.s.p;
    ep.out_poly_stride = size_Ql * n;
    ep.sub_poly_stride = size_QlP * n;
    if (alpha == 1)   // the single special limb, broadcast over Q_l inside the transform
        return ntt_forward_bcast(ctx, delta, polys, size_Ql * n, rq, addend ? EPI_SUB_MUL_ADD : EPI_SUB_MUL, ep, st, cp,
                                 size_QlP * n);
    return ntt_forward(ctx, delta, polys, size_Ql * n, rq, addend ? EPI_SUB_MUL_ADD : EPI_SUB_MUL, ep, st);
}

// Division of `polys` polynomials over Q_l u P by B = {q_{l-1}} u P in one moddown: out[polys][l-1][n] =
// (cx - [cx]_B) / prod(B) (x 2 when times2) with [.]_B the CENTRED remainder from an exact base conversion
// (bconv_centred_kernel): rounding to nearest, where moddown_from_NTT + rescale_to_next leave up to alpha units of
// conversion overflow and a floor.
// This is rescale_to_next o moddown merged: one NTT pass over the ciphertext instead of two, no separate
// INTT / broadcast of the last limb.  cx's limbs l-1 .. l+alpha-1 are destroyed; delta = scratch [polys][l-1][n].
int build_merged_down(pha_context *ctx, LevelTool *lt) {
    if (lt->merged.ready) return PHA_OK;
    const size_t size_Ql = lt->size_Ql, alpha = ctx->size_P, nb = alpha + 1, out_l = size_Ql - 1;
    const uint64_t *Q = ctx->moduli.data();
    std::vector<uint64_t> B;
    B.push_back(Q[size_Ql - 1]);
    for (size_t k = 0; k < alpha; k++) B.push_back(Q[ctx->size_Q + k]);
    if (nb + 1 > 16) return fail(PHA_ERR_INVALID_ARGUMENT, "the merged rescale supports up to 14 special primes");
    std::vector<uint64_t> hat_inv(nb), matrix(out_l * (nb + 1)), binv(out_l), binv2(out_l);
    std::vector<double> inv_b(nb);
    for (size_t b = 0; b < nb; b++) {
        hat_inv[b] = inv_mod_prime(product_mod_except(B.data(), 0, nb, b, B[b]), B[b]);
        inv_b[b] = 1.0 / static_cast<double>(B[b]);
    }
    for (size_t j = 0; j < out_l; j++) {
        for (size_t b = 0; b < nb; b++) matrix[j * (nb + 1) + b] = product_mod_except(B.data(), 0, nb, b, Q[j]);
        const uint64_t prod = product_mod_except(B.data(), 0, nb, SIZE_MAX, Q[j]);
        matrix[j * (nb + 1) + nb] = prod ? Q[j] - prod : 0;  // the column of the overflow count
        binv[j] = inv_mod_prime(prod, Q[j]);
        binv2[j] = (binv[j] * 2) % Q[j];
    }
    HIP_TRY(lt->merged.inv_b.upload(inv_b));
    std::vector<uint32_t> tidx(out_l);
    for (size_t j = 0; j < out_l; j++) tidx[j] = static_cast<uint32_t>(j);
    HIP_TRY(lt->merged.BHatInv.upload(hat_inv, B.data()));
    HIP_TRY(lt->merged.matrix.upload(matrix));
    HIP_TRY(lt->merged.obase_tidx.upload(tidx));
    HIP_TRY(lt->merged.BInv.upload(binv, Q));
    HIP_TRY(lt->merged.BInv2.upload(binv2, Q));
    lt->merged.ready = true;
    return PHA_OK;
}

int do_merged_moddown(pha_context *ctx, LevelTool *lt, uint64_t *out, uint64_t *cx, uint64_t *delta, size_t polys, bool times2,
                      hipStream_t st) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha, out_l = size_Ql - 1;
    if (size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    PHA_TRY(build_merged_down(ctx, lt));
    NttRange rb;
    PHA_TRY(make_range(ctx, alpha + 1, out_l, ctx->size_QP, alpha, 0, 0, rb));
    Epilogue ep{};
    ep.cst = lt->merged.BHatInv.v.p;
    ep.cst_shoup = lt->merged.BHatInv.s.p;
    ep.cst_limb0 = static_cast<uint32_t>(out_l);
    PHA_TRY(ntt_backward(ctx, cx, cx, polys, size_QlP * n, rb, EPI_SCALE, ep, st));
    {   // exact, centred base conversion B -> Q_{l-1} (bconv_centred_kernel): the division below rounds to nearest
        const uint32_t n32 = static_cast<uint32_t>(n), obase = static_cast<uint32_t>(out_l);
        const uint32_t base_wgs = (n32 / 256) * static_cast<uint32_t>(polys);
        const uint32_t groups = std::min<uint32_t>(obase, std::max<uint32_t>(1, (4096u + base_wgs - 1) / base_wgs));
        const uint32_t lpb = (obase + groups - 1) / groups;
        const dim3 grid(n32 / 256, (obase + lpb - 1) / lpb, static_cast<uint32_t>(polys));
#define CENTRED_CASE(IB)                                                                                                  \
    case IB:                                                                                                              \
        hipLaunchKernelGGL((bconv_centred_kernel<IB>), grid, dim3(256), 0, st, delta, cx + out_l * n, lt->merged.matrix.p, \
                           lt->merged.inv_b.p, ctx->d_mod.p, lt->merged.obase_tidx.p, obase, lpb, n32, out_l * n,          \
                           size_QlP * n);                                                                                 \
        break;
        switch (alpha + 1) {
            CENTRED_CASE(2) CENTRED_CASE(3) CENTRED_CASE(4) CENTRED_CASE(5) CENTRED_CASE(6) CENTRED_CASE(7) CENTRED_CASE(8)
            CENTRED_CASE(9) CENTRED_CASE(10) CENTRED_CASE(11) CENTRED_CASE(12) CENTRED_CASE(13) CENTRED_CASE(14) CENTRED_CASE(15)
            default: return fail(PHA_ERR_INVALID_ARGUMENT, "the merged rescale supports 1 to 14 special primes");
        }
#undef CENTRED_CASE
        PHA_TRY(check_launch("bconv_centred"));
    }
    NttRange rq;
    PHA_TRY(make_range(ctx, out_l, 0, 0, 0, 0, 0, rq));
    Epilogue e2{};
    e2.out = out;
    e2.sub = cx;
    e2.cst = times2 ? lt->merged.BInv2.v.p : lt->merged.BInv.v.p;
    e2.cst_shoup = times2 ? lt->merged.BInv2.s.p : lt->merged.BInv.s.p;
    e2.out_poly_stride = out_l * n;
    e2.sub_poly_stride = size_QlP * n;
    return ntt_forward(ctx, delta, polys, out_l * n, rq, EPI_SUB_MUL, e2, st);
}

// addend (default ct itself) is what the switched key part is added to; addend_every as in Epilogue
int do_keyswitch(pha_context *ctx, LevelTool *lt, uint64_t *ct, const uint64_t *c2, const uint64_t *const *evks,
                 hipStream_t st, const uint64_t *addend = nullptr, uint32_t addend_every = 0) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, (lt->beta * size_QlP + 2 * size_QlP + 2 * size_Ql) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *t_mod_up = ws, *cx = t_mod_up + lt->beta * size_QlP * n, *scratch = cx + 2 * size_QlP * n;
    // the digits' own limbs stay in c2 and the inner product reads them there (36 MiB less traffic at configs[2]); with one
    // special prime the modup copies them only where its broadcast kernel writes them anyway
    const bool in_place_own = ctx->size_P > 1 || bcast_in_column_pass(ctx);
    PHA_TRY(do_modup(ctx, lt, t_mod_up, c2, scratch, st, !in_place_own));
    PHA_TRY(do_inner_prod(ctx, lt, cx, t_mod_up, evks, st, nullptr, in_place_own ? c2 : nullptr));
    // both polys' moddown in one launch sequence, add_to_ct fused into the NTT epilogue
    // (src/eval_key_switch.cu:157-177)
    return do_moddown(ctx, lt, ct, cx, scratch, addend ? addend : ct, 2, st, addend_every);
}

int do_rescale(pha_context *ctx, LevelTool *lt, const uint64_t *src, const uint64_t *last_coeff, size_t last_stride,
               size_t cipher_size, uint64_t *dst, hipStream_t st) {
    // last_coeff[p * last_stride ...]: INTT'ed last limb of poly p
    const size_t n = ctx->n, size_Ql = lt->size_Ql, next = size_Ql - 1;
    NttRange r;
    PHA_TRY(make_range(ctx, next, 0, 0, 0, 0, 0, r));
    Epilogue ep{};
    ep.out = dst;
    ep.sub = src;
    ep.cst = lt->inv_q_last.v.p;
    ep.cst_shoup = lt->inv_q_last.s.p;
    ep.out_poly_stride = next * n;
    ep.sub_poly_stride = size_Ql * n;
    return ntt_forward_bcast(ctx, dst, cipher_size, next * n, r, EPI_SUB_MUL, ep, st, last_coeff, last_stride);
}

// rescale_to_next (src/evaluate.cu:585-616, src/rns.cu:1157-1181) with the caller's scratch [cipher_size][n]: the last limb of
// every poly is inverted into it (one launch), then broadcast over the remaining limbs and divided out
int rescale_with_scratch(pha_context *ctx, LevelTool *lt, const uint64_t *src, size_t cipher_size, uint64_t *dst, uint64_t *ws,
                         hipStream_t st) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql;
    NttRange r;
    PHA_TRY(make_range(ctx, 1, size_Ql - 1, 0, 0, 0, 0, r));
    PHA_TRY(ntt_backward(ctx, ws, src, cipher_size, size_Ql * n, r, EPI_NONE, Epilogue{}, st, 3, size_Ql - 1, n));
    return do_rescale(ctx, lt, src, ws, n, cipher_size, dst, st);
}

// ---- limb-parallel key switch -----------------------------------------------
// A rank owns the Q limbs [q0, q0+qc) of Q_l and the special limbs [p0, p0+pc) of P; buffers keep
// the single-GPU layouts and every step touches only the owned limbs.  The exchanges between the
// steps (all-gather of the scaled coefficient-form c2, all-gather of the scaled coefficient-form
// P limbs of cx, broadcast of the rescale limb) belong to the caller.
struct LimbWindow {
    size_t q0, qc, p0, pc;
};

int check_window(const pha_context *ctx, const LevelTool *lt, const LimbWindow &w) {
    if (w.q0 + w.qc > lt->size_Ql || w.p0 + w.pc > ctx->size_P)
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb window [%zu,+%zu) / [%zu,+%zu) exceeds Q_l (%zu) or P (%zu)", w.q0, w.qc,
                    w.p0, w.pc, lt->size_Ql, ctx->size_P);
    return PHA_OK;
}

// P window of a Q_l u P buffer: buffer limbs [size_Ql + p0, +pc), key-level table indices
int make_p_window(const pha_context *ctx, const LevelTool *lt, const LimbWindow &w, NttRange &r) {
    PHA_TRY(make_range(ctx, w.pc, lt->size_Ql + w.p0, 0, 0, 0, 0, r));
    r.lm.tidx_bias = static_cast<uint32_t>(ctx->size_Q - lt->size_Ql);
    return PHA_OK;
}

// owned limb q0 + j lands in row row_begin + j of c2_coeff (the buffer the ranks all-gather: rank r's block of rows)
int lp_modup_begin(pha_context *ctx, LevelTool *lt, const uint64_t *c2_ntt, uint64_t *c2_coeff, size_t row_begin,
                   const LimbWindow &w, hipStream_t st) {
    if (!w.qc) return PHA_OK;
    NttRange r;
    PHA_TRY(make_range(ctx, w.qc, w.q0, 0, 0, 0, 0, r));
    Epilogue ep{};
    // limb q0 + j -> row row_begin + j; the pointer moves up when the rows start beyond the limb index
    uint64_t *out = c2_coeff;
    size_t limb0 = w.q0 - row_begin;
    if (row_begin > w.q0) {
        out = c2_coeff + (row_begin - w.q0) * ctx->n;
        limb0 = 0;
    }
    if (ctx->size_P == 1) return ntt_backward(ctx, out, c2_ntt, 1, 0, r, EPI_NONE, ep, st, 3, limb0, 0);
    ep.cst = lt->partHatInv.v.p;
    ep.cst_shoup = lt->partHatInv.s.p;
    return ntt_backward(ctx, out, c2_ntt, 1, 0, r, EPI_SCALE, ep, st, 3, limb0, 0);
}

// q_rows != nullptr: the coefficient form of limb j is row q_rows[j] of c2_coeff (rank blocks padded to equal size)
int lp_modup_finish(pha_context *ctx, LevelTool *lt, uint64_t *t_mod_up, const uint64_t *c2_ntt, const uint64_t *c2_coeff,
                    const uint32_t *q_rows, const LimbWindow &w, hipStream_t st) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha;
    const size_t win[2][2] = {{w.q0, w.q0 + w.qc}, {size_Ql + w.p0, size_Ql + w.p0 + w.pc}};
    if (q_rows && size_Ql > kBconvMaxRows)
        return fail(PHA_ERR_INVALID_ARGUMENT, "a c2_coeff row table holds at most %u limbs", kBconvMaxRows);
    if (alpha == 1) {
        CoeffRows rows{};
        if (q_rows) {
            rows.set = 1;
            for (size_t j = 0; j < size_Ql; j++) rows.row[j] = static_cast<uint16_t>(q_rows[j]);
        }
        for (const auto &jw : win) {
            if (jw[0] == jw[1]) continue;
            hipLaunchKernelGGL(modup_single_p_kernel, dim3(static_cast<uint32_t>(n / 256), static_cast<uint32_t>(lt->beta)),
                               dim3(256), 0, st, t_mod_up, c2_ntt, c2_coeff, ctx->d_mod.p, static_cast<uint32_t>(size_Ql),
                               static_cast<uint32_t>(ctx->size_Q), static_cast<uint32_t>(n), static_cast<uint32_t>(jw[0]),
                               static_cast<uint32_t>(jw[1]), rows);
            PHA_TRY(check_launch("modup_single_p"));
        }
    } else {
        // the launches of do_modup, restricted to the outputs (and own-limb copies) inside this rank's windows
        const uint32_t own[4] = {static_cast<uint32_t>(win[0][0]), static_cast<uint32_t>(win[0][1]), static_cast<uint32_t>(win[1][0]),
                                 static_cast<uint32_t>(win[1][1])};
        // first: the digit's first limb (its inputs are limbs first, first + 1, ...); stride: limbs from one digit of a
        // batched launch to the next
        auto owned = [&](BconvBatch bt, size_t first, size_t stride) {
            bt.own_set = 1;
            for (int i = 0; i < 4; i++) bt.own[i] = own[i];
            if (q_rows) {
                bt.in_rows_set = 1;
                bt.in_row_stride = static_cast<uint32_t>(stride);
                bt.y_stride = 0;
                for (size_t j = first; j < size_Ql; j++) bt.in_row[j - first] = static_cast<uint16_t>(q_rows[j]);
            }
            return bt;
        };
        if (lt->uniform_digits) {
            const uint32_t obase = static_cast<uint32_t>(size_QlP - alpha);
            BconvBatch bt{size_QlP * n, alpha * n, static_cast<size_t>(obase) * alpha, obase,
                          static_cast<uint32_t>(alpha), c2_ntt, alpha * n, 0, 0, 0};
            PHA_TRY(launch_bconv(ctx, t_mod_up, c2_coeff, lt->modup_matrix_all.p, lt->modup_tidx_all.p,
                                 static_cast<uint32_t>(alpha), obase, 0, static_cast<uint32_t>(alpha),
                                 static_cast<uint32_t>(lt->beta), owned(bt, 0, alpha), st, &lt->modup_mfma_all));
        } else {
            for (size_t d = 0; d < lt->beta; d++) {
                const size_t s = d * alpha, e = std::min(s + alpha, size_Ql), len = e - s;
                BconvBatch bt{0, 0, 0, 0, 0, c2_ntt + s * n, 0, 0, 0, 0};
                PHA_TRY(launch_bconv(ctx, t_mod_up + d * size_QlP * n, q_rows ? c2_coeff : c2_coeff + s * n, lt->modup_matrix[d].p,
                                     lt->modup_obase_tidx[d].p, static_cast<uint32_t>(len),
                                     static_cast<uint32_t>(size_QlP - len), static_cast<uint32_t>(s),
                                     static_cast<uint32_t>(len), 1, owned(bt, s, 0), st, &lt->modup_mfma[d]));
            }
        }
    }
    if (w.qc && w.pc == alpha && w.q0 + w.qc == size_Ql) {
        // the Q window ends where the (whole) P window begins, e.g. a single rank: one transform over both, the special
        // limbs on the key-level tables (a LimbMap addresses the special limbs from the end of P, so a partial P window
        // keeps its own launch)
        NttRange r;
        PHA_TRY(make_range(ctx, w.qc + w.pc, w.q0, ctx->size_QP, w.pc, 0, 0, r));
        r.lm.digit_alpha = static_cast<uint32_t>(alpha);
        r.lm.digit_limit = static_cast<uint32_t>(size_Ql);
        return ntt_forward(ctx, t_mod_up, lt->beta, size_QlP * n, r, EPI_NONE, Epilogue{}, st);
    }
    if (w.qc) {
        NttRange r;
        PHA_TRY(make_range(ctx, w.qc, w.q0, 0, 0, 0, 0, r));
        r.lm.digit_alpha = static_cast<uint32_t>(alpha);
        r.lm.digit_limit = static_cast<uint32_t>(size_Ql);
        PHA_TRY(ntt_forward(ctx, t_mod_up, lt->beta, size_QlP * n, r, EPI_NONE, Epilogue{}, st));
    }
    if (w.pc) {
        NttRange r;
        PHA_TRY(make_p_window(ctx, lt, w, r));
        PHA_TRY(ntt_forward(ctx, t_mod_up, lt->beta, size_QlP * n, r, EPI_NONE, Epilogue{}, st));
    }
    return PHA_OK;
}

int lp_inner_prod(pha_context *ctx, LevelTool *lt, uint64_t *cx, const uint64_t *t_mod_up, const uint64_t *const *evks,
                  const LimbWindow &w, hipStream_t st) {
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    const uint32_t size_QlP = static_cast<uint32_t>(lt->size_Ql + ctx->size_P);
    size_t win[2][2] = {{w.q0, w.qc}, {lt->size_Ql + w.p0, w.pc}};
    if (w.qc && w.pc && w.q0 + w.qc == lt->size_Ql && w.p0 == 0) {  // adjacent windows: one launch
        win[0][1] = w.qc + w.pc;
        win[1][1] = 0;
    }
    for (const auto &lw : win) {
        if (!lw[1]) continue;
        hipLaunchKernelGGL(key_switch_inner_prod_kernel, dim3(static_cast<uint32_t>(lw[1]) * (n / 512)), dim3(256), 0, st, cx,
                           t_mod_up, evks, ctx->d_mod.p, n, static_cast<uint32_t>(ctx->size_QP), size_QlP,
                           static_cast<uint32_t>(ctx->size_Q), static_cast<uint32_t>(lt->size_Ql),
                           static_cast<uint32_t>(lt->beta), static_cast<uint32_t>(lw[0]));
        PHA_TRY(check_launch("key_switch_inner_prod"));
    }
    return PHA_OK;
}

// INTT (+ bconv phase 1) of the owned P limbs of cx, both polys, written to p_coeff[poly][p][n] (P limb p at row p,
// polys p_stride words apart): the buffer the ranks all-gather before the P -> Q_l conversion
int lp_moddown_begin(pha_context *ctx, LevelTool *lt, const uint64_t *cx, uint64_t *p_coeff, size_t p_stride, size_t row_begin,
                     const LimbWindow &w, hipStream_t st) {
    if (!w.pc) return PHA_OK;
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    NttRange rp;
    PHA_TRY(make_p_window(ctx, lt, w, rp));
    Epilogue ep{};
    // buffer limb size_Ql + p0 + j -> row row_begin + j; the pointer moves up when the rows start beyond the limb index
    const size_t first = size_Ql + w.p0;
    uint64_t *out = p_coeff;
    size_t limb0 = first - row_begin;
    if (row_begin > first) {
        out = p_coeff + (row_begin - first) * n;
        limb0 = 0;
    }
    if (ctx->size_P == 1) return ntt_backward(ctx, out, cx, 2, size_QlP * n, rp, EPI_NONE, ep, st, 3, limb0, p_stride);
    ep.cst = lt->PHatInv.v.p;
    ep.cst_shoup = lt->PHatInv.s.p;
    ep.cst_limb0 = static_cast<uint32_t>(size_Ql);
    return ntt_backward(ctx, out, cx, 2, size_QlP * n, rp, EPI_SCALE, ep, st, 3, limb0, p_stride);
}

// ct[owned] += (cx - NTT(P -> Q_l conversion of the gathered p_coeff)) * P^-1
int lp_moddown_finish(pha_context *ctx, LevelTool *lt, uint64_t *ct, uint64_t *cx, const uint64_t *p_coeff, size_t p_stride,
                      const uint32_t *p_rows, const LimbWindow &w, hipStream_t st) {
    if (!w.qc) return PHA_OK;
    const size_t n = ctx->n, size_Ql = lt->size_Ql, alpha = ctx->size_P, size_QlP = size_Ql + alpha;
    uint64_t *delta;
    PHA_TRY(ensure_workspace(ctx, 2 * size_Ql * n, &delta, st));
    WsLease ws_lease(ctx, st);
    if (alpha != 1) {
        BconvBatch bt{size_Ql * n, p_stride, 0, 0, 0, nullptr, 0, 0, 0, 0};
        if (p_rows) {
            if (alpha > kBconvMaxRows) return fail(PHA_ERR_INVALID_ARGUMENT, "a p_coeff row table holds at most %u special primes", kBconvMaxRows);
            bt.in_rows_set = 1;
            for (size_t k = 0; k < alpha; k++) bt.in_row[k] = static_cast<uint16_t>(p_rows[k]);
        }
        // the whole matrix with this rank's Q window as the owned outputs
        bt.own_set = 1;
        bt.own[0] = static_cast<uint32_t>(w.q0);
        bt.own[1] = static_cast<uint32_t>(w.q0 + w.qc);
        bt.own[2] = bt.own[3] = 0;
        PHA_TRY(launch_bconv(ctx, delta, p_coeff, lt->moddown_matrix.p, lt->moddown_obase_tidx.p, static_cast<uint32_t>(alpha),
                             static_cast<uint32_t>(size_Ql), static_cast<uint32_t>(size_Ql), 0, 2, bt, st, &lt->moddown_mfma));
    }
    NttRange rq;
    PHA_TRY(make_range(ctx, w.qc, w.q0, 0, 0, 0, 0, rq));
    Epilogue ep{};
    ep.out = ct;
    ep.sub = cx;
    ep.addend = ct;
    ep.cst = lt->bigPInv.v.p;
    ep.cst_shoup = lt->bigPInv.s.p;
    ep.out_poly_stride = size_Ql * n;
    ep.sub_poly_stride = size_QlP * n;
    if (alpha == 1)
        return ntt_forward_bcast(ctx, delta, 2, size_Ql * n, rq, EPI_SUB_MUL_ADD, ep, st, p_coeff + (p_rows ? p_rows[0] : 0) * n,
                                 p_stride);
    return ntt_forward(ctx, delta, 2, size_Ql * n, rq, EPI_SUB_MUL_ADD, ep, st);
}

int lp_rescale_finish(pha_context *ctx, LevelTool *lt, const uint64_t *src, const uint64_t *last_coeff, size_t cipher_size,
                      uint64_t *dst, const LimbWindow &w, hipStream_t st) {
    const size_t n = ctx->n, size_Ql = lt->size_Ql, next = size_Ql - 1;
    const size_t a = std::min(w.q0, next), b = std::min(w.q0 + w.qc, next);
    if (a >= b) return PHA_OK;
    NttRange r;
    PHA_TRY(make_range(ctx, b - a, a, 0, 0, 0, 0, r));
    Epilogue ep{};
    ep.out = dst;
    ep.sub = src;
    ep.cst = lt->inv_q_last.v.p;
    ep.cst_shoup = lt->inv_q_last.s.p;
    ep.out_poly_stride = next * n;
    ep.sub_poly_stride = size_Ql * n;
    return ntt_forward_bcast(ctx, dst, cipher_size, next * n, r, EPI_SUB_MUL, ep, st, last_coeff, n);
}

// ---- key generation / encryption / decryption -------------------------------
// The process's sampler key (csrc/keygen_kernels.cuh): 32 bytes of OS entropy on first use, or what
// pha_set_sampler_key installed.  Guarded by a mutex: contexts of several threads share it.
std::mutex g_sampler_mutex;
bool g_sampler_key_set = false;
SamplerKey g_sampler_key;

int os_entropy(void *buf, size_t len) {
    uint8_t *p = static_cast<uint8_t *>(buf);
    while (len) {
        const ssize_t got = getrandom(p, len, 0);
        if (got < 0) {
            if (errno == EINTR) continue;
            return fail(PHA_ERR_LOGIC, "getrandom failed: %s", strerror(errno));
        }
        p += got;
        len -= static_cast<size_t>(got);
    }
    return PHA_OK;
}

int sampler_key(SamplerKey &out) {
    std::lock_guard<std::mutex> lock(g_sampler_mutex);
    if (!g_sampler_key_set) {
        PHA_TRY(os_entropy(g_sampler_key.w, sizeof(g_sampler_key.w)));
        g_sampler_key_set = true;
    }
    out = g_sampler_key;
    return PHA_OK;
}

// same stream as rnd_at() on the device
int host_rnd_at(uint64_t seed, uint64_t ctr, uint64_t &out) {
    SamplerKey key;
    PHA_TRY(sampler_key(key));
    out = rnd_at(key, seed, ctr);
    return PHA_OK;
}

// chain_index 0 = key level (all of Q u P), i >= 1 = the first size_Q-(i-1) limbs of Q
int level_limbs(const pha_context *ctx, size_t chain_index, size_t *limbs) {
    if (chain_index > ctx->size_Q) return fail(PHA_ERR_INVALID_ARGUMENT, "chain_index %zu out of range", chain_index);
    *limbs = chain_index == 0 ? ctx->size_QP : ctx->size_Q - (chain_index - 1);
    return PHA_OK;
}

int launch_sample(pha_context *ctx, int kind, uint64_t *out, size_t limbs, uint64_t seed, hipStream_t st) {
    const uint32_t n = static_cast<uint32_t>(ctx->n), L = static_cast<uint32_t>(limbs);
    SamplerKey key;
    PHA_TRY(sampler_key(key));
    if (kind == SAMPLE_UNIFORM)
        hipLaunchKernelGGL((sample_poly_kernel<SAMPLE_UNIFORM>), dim3(n / 256, L), dim3(256), 0, st, out, ctx->d_mod.p, L, n, key, seed);
    else if (kind == SAMPLE_TERNARY)
        hipLaunchKernelGGL((sample_poly_kernel<SAMPLE_TERNARY>), dim3(n / 256), dim3(256), 0, st, out, ctx->d_mod.p, L, n, key, seed);
    else
        hipLaunchKernelGGL((sample_poly_kernel<SAMPLE_ERROR>), dim3(n / 256), dim3(256), 0, st, out, ctx->d_mod.p, L, n, key, seed);
    return check_launch("sample_poly");
}

// limbs [limb0, limb0+limbs) of the chain; buffer pointers already point at limb0
int launch_ew(pha_context *ctx, int op, uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
              const uint64_t *cst, size_t limb0, size_t limbs, hipStream_t st) {
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    const dim3 grid(static_cast<uint32_t>(limbs) * (n / 512));
    const DModulus *mod = ctx->d_mod.p + limb0;
#define EW_CASE(OP) case OP: hipLaunchKernelGGL((elementwise_kernel<OP>), grid, dim3(256), 0, st, dst, a, b, c, mod, cst, n); break;
    switch (op) {
        EW_CASE(EW_MUL) EW_CASE(EW_MUL_ADD) EW_CASE(EW_MUL_ADD_NEG) EW_CASE(EW_ADD) EW_CASE(EW_MUL_CONST_ADD)
    }
#undef EW_CASE
    return check_launch("elementwise");
}

// src/secretkey.cu:207-263 with is_ntt_form = true
int do_encrypt_zero(pha_context *ctx, const uint64_t *sk, size_t chain_index, uint64_t seed_a, uint64_t seed_e, uint64_t *ct,
                    hipStream_t st) {
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    uint64_t *c0 = ct, *c1 = ct + limbs * ctx->n;
    PHA_TRY(launch_sample(ctx, SAMPLE_ERROR, c0, limbs, seed_e, st));
    NttRange r;
    PHA_TRY(make_range(ctx, limbs, 0, 0, 0, 0, 0, r));
    PHA_TRY(ntt_forward(ctx, c0, 1, 0, r, EPI_NONE, Epilogue{}, st));
    PHA_TRY(launch_sample(ctx, SAMPLE_UNIFORM, c1, limbs, seed_a, st));
    return launch_ew(ctx, EW_MUL_ADD_NEG, c0, c1, sk, c0, nullptr, 0, limbs, st);
}

// src/secretkey.cu:265-310, src/polymath.cu:323-343; evks: HOST array of dnum device buffers
int do_kswitch_key(pha_context *ctx, const uint64_t *sk, const uint64_t *new_key, uint64_t seed, uint64_t *const *evks,
                   hipStream_t st) {
    if (ctx->size_P == 0 || ctx->size_Q % ctx->size_P)
        return fail(PHA_ERR_LOGIC, "key switching keys need size_P > 0 dividing size_Q (dnum = size_Q / size_P)");
    const size_t n = ctx->n, alpha = ctx->size_P, dnum = ctx->size_Q / alpha;
    if (!ctx->d_bigP_mod_q.p) {
        LevelConstants lc;
        build_level_constants(ctx->moduli, ctx->size_P, ctx->size_Q, lc);
        HIP_TRY(ctx->d_bigP_mod_q.upload(lc.bigP_mod_q));
    }
    for (size_t d = 0; d < dnum; d++) {
        if (!evks[d]) return fail(PHA_ERR_INVALID_ARGUMENT, "null key buffer");
        uint64_t seed_a, seed_e;
        PHA_TRY(host_rnd_at(seed, 2 * d, seed_a));
        PHA_TRY(host_rnd_at(seed, 2 * d + 1, seed_e));
        PHA_TRY(do_encrypt_zero(ctx, sk, 0, seed_a, seed_e, evks[d], st));
        const size_t j0 = d * alpha;
        PHA_TRY(launch_ew(ctx, EW_MUL_CONST_ADD, evks[d] + j0 * n, new_key + j0 * n, nullptr, evks[d] + j0 * n,
                          ctx->d_bigP_mod_q.p + j0, j0, alpha, st));
    }
    return PHA_OK;
}

// ---- CKKS encoder (src/ckks.cu:130-283) ---------------------------------------
int get_encoder_level(pha_context *ctx, size_t limbs, pha_context::EncoderLevel **out) {
    if (!ctx->d_rot_group.p) {
        std::vector<uint32_t> rot(ctx->n / 4 ? ctx->n / 4 : 1);
        uint32_t pos = 1;
        for (auto &r : rot) {
            r = pos;
            pos = (pos * 5u) & static_cast<uint32_t>(2 * ctx->n - 1);
        }
        HIP_TRY(ctx->d_rot_group.upload(rot));
    }
    auto it = ctx->enc_levels.find(limbs);
    if (it == ctx->enc_levels.end()) {
        if (limbs > static_cast<size_t>(kMaxCrtWords)) return fail(PHA_ERR_INVALID_ARGUMENT, "too many limbs for the encoder");
        CrtTables t;
        build_crt_tables(ctx->moduli.data(), limbs, t);
        auto lv = std::make_unique<pha_context::EncoderLevel>();
        HIP_TRY(lv->big_q.upload(t.big_q));
        HIP_TRY(lv->half_q.upload(t.half_q));
        HIP_TRY(lv->qhat.upload(t.qhat));
        HIP_TRY(lv->inv_qhat.upload(t.inv_qhat));
        size_t top = limbs - 1;
        while (top > 0 && t.big_q[top] == 0) top--;
        lv->total_bits = static_cast<int>(64 * top + (64 - __builtin_clzll(t.big_q[top])));
        it = ctx->enc_levels.emplace(limbs, std::move(lv)).first;
    }
    *out = it->second.get();
    return PHA_OK;
}

int check_slots(const pha_context *ctx, size_t s) {
    if (s < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "single value encoding is not available");
    if ((s & (s - 1)) || s > ctx->n / 2) return fail(PHA_ERR_INVALID_ARGUMENT, "sparse_slots must be a power of two <= N/2");
    return PHA_OK;
}

template <bool INVERSE>
void special_fft_stages(pha_context *ctx, double2 *v, uint32_t s, hipStream_t st) {
    const dim3 grid((s / 2 + 255) / 256);
    if (!INVERSE)
        for (uint32_t len = 2; len <= s; len <<= 1)
            hipLaunchKernelGGL((special_fft_stage_kernel<false>), grid, dim3(256), 0, st, v, ctx->d_rot_group.p, s, len);
    else
        for (uint32_t len = s; len >= 2; len >>= 1)
            hipLaunchKernelGGL((special_fft_stage_kernel<true>), grid, dim3(256), 0, st, v, ctx->d_rot_group.p, s, len);
}

int get_galois_table(pha_context *ctx, uint32_t elt, const uint32_t **out) {
    if (!(elt & 1) || elt >= 2 * ctx->n) return fail(PHA_ERR_INVALID_ARGUMENT, "Galois element is not valid");
    auto it = ctx->galois_tables.find(elt);
    if (it == ctx->galois_tables.end()) {
        std::vector<uint32_t> h(ctx->n);
        galois_permutation(ctx->log_n, elt, h.data());
        auto buf = std::make_unique<DevBuf<uint32_t>>();
        HIP_TRY(buf->upload(h));
        it = ctx->galois_tables.emplace(elt, std::move(buf)).first;
    }
    *out = it->second->p;
    return PHA_OK;
}

int do_galois(pha_context *ctx, const uint64_t *operand, size_t limbs, const uint32_t *perm, uint64_t *result,
              hipStream_t st) {
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    hipLaunchKernelGGL(apply_galois_ntt_kernel, dim3(static_cast<uint32_t>(limbs) * (n / 512)), dim3(256), 0, st, result,
                       operand, perm, n);
    return check_launch("apply_galois_ntt");
}

inline hipStream_t as_stream(pha_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *pha_last_error(void) { return g_last_error.c_str(); }
const char *pha_version(void) { return "phantom-fhe_amd 0.1 (gfx950)"; }

int pha_coeff_modulus_create(size_t poly_modulus_degree, const int *bit_sizes, size_t count, uint64_t *out_moduli) {
    if (!bit_sizes || !out_moduli) return fail(PHA_ERR_INVALID_ARGUMENT, "null argument");
    if (poly_modulus_degree < 2 || (poly_modulus_degree & (poly_modulus_degree - 1)) || poly_modulus_degree > 131072)
        return fail(PHA_ERR_INVALID_ARGUMENT, "poly_modulus_degree is invalid");
    for (size_t i = 0; i < count; i++)
        if (bit_sizes[i] < 2 || bit_sizes[i] > 60) return fail(PHA_ERR_INVALID_ARGUMENT, "bit_sizes is invalid");
    if (!coeff_modulus_create(poly_modulus_degree, bit_sizes, count, out_moduli))
        return fail(PHA_ERR_LOGIC, "failed to find enough qualifying primes");
    return PHA_OK;
}

int pha_host_ntt_tables(int log_n, uint64_t modulus, uint64_t *tw, uint64_t *tw_shoup, uint64_t *itw, uint64_t *itw_shoup,
                        uint64_t *n_inv, uint64_t *n_inv_shoup) {
    if (log_n < 1 || log_n > 17) return fail(PHA_ERR_INVALID_ARGUMENT, "log_n out of range");
    LimbTables t;
    if (modulus >> 61 || !is_prime_u64(modulus) || !build_limb_tables(log_n, modulus, t))
        return fail(PHA_ERR_INVALID_ARGUMENT, "invalid modulus in try_minimal_primitive_root");
    const size_t bytes = (size_t(1) << log_n) * sizeof(uint64_t);
    if (tw) memcpy(tw, t.tw.data(), bytes);
    if (tw_shoup) memcpy(tw_shoup, t.tw_shoup.data(), bytes);
    if (itw) memcpy(itw, t.itw.data(), bytes);
    if (itw_shoup) memcpy(itw_shoup, t.itw_shoup.data(), bytes);
    if (n_inv) *n_inv = t.n_inv;
    if (n_inv_shoup) *n_inv_shoup = t.n_inv_shoup;
    return PHA_OK;
}

int pha_host_barrett_ratio(uint64_t modulus, uint64_t ratio[2]) {
    if (modulus < 2 || modulus >> 61) return fail(PHA_ERR_INVALID_ARGUMENT, "value can be at most 61-bit and cannot be 1");
    barrett_ratio(modulus, ratio);
    return PHA_OK;
}

int pha_host_galois_table(int log_n, uint32_t galois_elt, uint32_t *table) {
    if (log_n < 1 || log_n > 17 || !table) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_count_power out of range");
    if (!(galois_elt & 1) || galois_elt >= (2u << log_n)) return fail(PHA_ERR_INVALID_ARGUMENT, "Galois element is not valid");
    galois_permutation(log_n, galois_elt, table);
    return PHA_OK;
}

int pha_galois_elt_from_step(int log_n, int step, uint32_t *out_elt) {
    if (log_n < 1 || log_n > 17 || !out_elt) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_count_power out of range");
    if (!galois_elt_from_step(log_n, step, *out_elt)) return fail(PHA_ERR_INVALID_ARGUMENT, "step count too large");
    return PHA_OK;
}

int pha_host_level_constants(const uint64_t *moduli_QP, size_t size_QP, size_t size_P, size_t size_Ql,
                             uint64_t *inv_q_last_mod_q, uint64_t *bigP_mod_q, uint64_t *bigPInv_mod_q,
                             uint64_t *partQlHatInv_mod_Ql_concat) {
    if (!moduli_QP || size_P == 0 || size_P >= size_QP || size_Ql == 0 || size_Ql > size_QP - size_P)
        return fail(PHA_ERR_INVALID_ARGUMENT, "RNSBase is invalid");
    LevelConstants lc;
    build_level_constants(std::vector<uint64_t>(moduli_QP, moduli_QP + size_QP), size_P, size_Ql, lc);
    auto put = [](uint64_t *dst, const std::vector<uint64_t> &v) {
        if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(uint64_t));
    };
    put(inv_q_last_mod_q, lc.inv_q_last_mod_q);
    put(bigP_mod_q, lc.bigP_mod_q);
    put(bigPInv_mod_q, lc.bigPInv_mod_q);
    put(partQlHatInv_mod_Ql_concat, lc.partQlHatInv_mod_Ql_concat);
    return PHA_OK;
}

int pha_context_create(int log_n, const uint64_t *moduli_QP, size_t size_QP, size_t size_P, pha_context **out) {
    if (!moduli_QP || !out) return fail(PHA_ERR_INVALID_ARGUMENT, "null argument");
    NttPlan plan;
    if (!plan_for(log_n, plan))
        return fail(PHA_ERR_INVALID_ARGUMENT, "unsupported polynomial degree when selecting sample size");
    if (size_QP == 0 || size_QP > 64 || size_P >= size_QP) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_modulus is invalid");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PHA_ERR_NO_DEVICE, "no HIP device available");
    auto ctx = std::make_unique<pha_context>();
    ctx->log_n = log_n;
    ctx->n = size_t(1) << log_n;
    ctx->size_QP = size_QP;
    ctx->size_P = size_P;
    ctx->size_Q = size_QP - size_P;
    ctx->plan = plan;
    ctx->moduli.assign(moduli_QP, moduli_QP + size_QP);
    const char *no_fp = getenv("PHA_DISABLE_FP64");
    ctx->fp64 = !(no_fp && no_fp[0] == '1') && !plan.small;   // the LDS kernels of N = 2^11 are integer-only
    {
        auto env_u = [](const char *name, uint32_t dflt) {
            const char *e = getenv(name);
            return e && *e ? static_cast<uint32_t>(atoi(e)) : dflt;
        };
        ctx->xcd_map = env_u("PHA_XCD_MAP", 1) ? 1u : 0u;
        ctx->bconv_mfma = env_u("PHA_BCONV_MFMA", 1) != 0;
        ctx->fused = static_cast<int>(env_u("PHA_FUSED_NTT", 0));
        ctx->fused_flags = env_u("PHA_FUSED_FLAGS", 0);
        ctx->fused_min_units = env_u("PHA_FUSED_MIN_UNITS", 16);
        int dev = 0;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDevice(&dev));
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
        ctx->device = dev;
        if (dev >= 0 && dev < kMaxDevices) {
            // this device's copy of the scratch array (every GPU holds its own, at its own address); contexts may be
            // created from several threads at once
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            if (!g_scratch_ptr[dev]) {
                uint64_t *p = nullptr;
                HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void **>(&p), HIP_SYMBOL(scratch)));
                g_scratch_ptr[dev] = p;
            }
        }
    }
    PHA_TRY(fused_census(ctx.get()));
    ctx->narrow = true;
    for (size_t i = 0; i < size_QP; i++) ctx->narrow = ctx->narrow && (moduli_QP[i] >> 60) == 0;
    ctx->host_tables.resize(size_QP);
    for (size_t i = 0; i < size_QP; i++) {
        for (size_t j = 0; j < i; j++)
            if (moduli_QP[i] == moduli_QP[j]) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_modulus is not coprime");
        if (moduli_QP[i] >> 61 || !is_prime_u64(moduli_QP[i]) ||
            !build_limb_tables(log_n, moduli_QP[i], ctx->host_tables[i]))
            return fail(PHA_ERR_INVALID_ARGUMENT, "invalid modulus in try_minimal_primitive_root");
    }
    PHA_TRY(upload_tables(ctx.get()));
    *out = ctx.release();
    return PHA_OK;
}

void pha_context_destroy(pha_context *ctx) {
    if (!ctx) return;
    (void)hipDeviceSynchronize();
    if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    for (int c = 0; c < pha_context::kAuxStreams; c++) {
        if (ctx->aux_join[c]) (void)hipEventDestroy(ctx->aux_join[c]);
        if (ctx->aux_stream[c]) (void)hipStreamDestroy(ctx->aux_stream[c]);
    }
    delete ctx;
}

size_t pha_context_n(const pha_context *ctx) { return ctx ? ctx->n : 0; }
size_t pha_context_size_QP(const pha_context *ctx) { return ctx ? ctx->size_QP : 0; }
size_t pha_context_size_P(const pha_context *ctx) { return ctx ? ctx->size_P : 0; }

int pha_context_get_moduli(const pha_context *ctx, uint64_t *out_host) {
    if (!ctx || !out_host) return fail(PHA_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out_host, ctx->moduli.data(), ctx->size_QP * sizeof(uint64_t));
    return PHA_OK;
}

int pha_context_get_table(const pha_context *ctx, size_t limb, int which, uint64_t *out_host) {
    if (!ctx || !out_host || limb >= ctx->size_QP || which < 0 || which > 3)
        return fail(PHA_ERR_INVALID_ARGUMENT, "invalid table request");
    // read back from the device copy so that the getter checks what the kernels actually use
    const tw_t *src = (which < 2 ? ctx->d_tw.p : ctx->d_itw.p) + limb * ctx->n;
    std::vector<tw_t> h(ctx->n);
    HIP_TRY(hipMemcpy(h.data(), src, ctx->n * sizeof(tw_t), hipMemcpyDeviceToHost));
    if (ctx->limb_is_fp64(limb)) {
        // FP64 limbs keep one double per twiddle: the value comes back from the device, its Shoup
        // companion (unused by the kernels on such a limb) from the host tables
        const LimbTables &t = ctx->host_tables[limb];
        const double *w = reinterpret_cast<const double *>(h.data());
        for (size_t i = 0; i < ctx->n; i++)
            out_host[i] = (which & 1) ? (which == 1 ? t.tw_shoup[i] : t.itw_shoup[i]) : static_cast<uint64_t>(w[i]);
        return PHA_OK;
    }
    for (size_t i = 0; i < ctx->n; i++) out_host[i] = (which & 1) ? h[i].y : h[i].x;
    return PHA_OK;
}

#define REQUIRE_CTX(ptr_ok)                                                          \
    if (!ctx || !(ptr_ok)) return fail(PHA_ERR_INVALID_ARGUMENT, "null argument")

int pha_nwt_2d_radix8_forward_inplace(pha_context *ctx, uint64_t *inout, size_t count, size_t start, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    return ntt_forward(ctx, inout, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_backward_inplace(pha_context *ctx, uint64_t *inout, size_t count, size_t start, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    return ntt_backward(ctx, inout, inout, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_backward(pha_context *ctx, uint64_t *out, const uint64_t *in, size_t count, size_t start,
                               pha_stream_t stream) {
    REQUIRE_CTX(out && in);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    return ntt_backward(ctx, out, in, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_backward_scale(pha_context *ctx, uint64_t *out, const uint64_t *in, size_t count, size_t start,
                                     const uint64_t *scale, const uint64_t *scale_shoup, pha_stream_t stream) {
    REQUIRE_CTX(out && in && scale && scale_shoup);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    Epilogue ep{};
    ep.cst = scale;
    ep.cst_shoup = scale_shoup;
    return ntt_backward(ctx, out, in, 1, 0, r, EPI_SCALE, ep, as_stream(stream));
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context *ctx, uint64_t *inout, size_t count, size_t start,
                                                          size_t size_QP, size_t size_P, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, size_QP, size_P, 0, 0, r));
    return ntt_forward(ctx, inout, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(pha_context *ctx, uint64_t *inout, size_t count,
                                                                        size_t start, size_t size_QP, size_t size_P,
                                                                        size_t excluded_range_start,
                                                                        size_t excluded_range_end, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    if (excluded_range_start < start || excluded_range_end > start + count)
        return fail(PHA_ERR_INVALID_ARGUMENT, "Excluded range in NTT is invalid.");
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, size_QP, size_P, excluded_range_start, excluded_range_end, r));
    return ntt_forward(ctx, inout, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_backward_inplace_include_special_mod(pha_context *ctx, uint64_t *inout, size_t count, size_t start,
                                                           size_t size_QP, size_t size_P, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, size_QP, size_P, 0, 0, r));
    return ntt_backward(ctx, inout, inout, 1, 0, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_nwt_2d_radix8_forward_inplace_fuse_moddown(pha_context *ctx, uint64_t *ct, const uint64_t *cx,
                                                   const uint64_t *bigPInv_mod_q, const uint64_t *bigPInv_mod_q_shoup,
                                                   uint64_t *delta, size_t count, size_t start, pha_stream_t stream) {
    REQUIRE_CTX(ct && cx && bigPInv_mod_q && bigPInv_mod_q_shoup && delta);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    Epilogue ep{};
    ep.out = ct;
    ep.sub = cx;
    ep.cst = bigPInv_mod_q;
    ep.cst_shoup = bigPInv_mod_q_shoup;
    return ntt_forward(ctx, delta, 1, 0, r, EPI_SUB_MUL, ep, as_stream(stream));
}

int pha_ntt_forward_batched(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs, size_t count,
                            size_t start, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    return ntt_forward(ctx, inout, polys, poly_stride_limbs * ctx->n, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_ntt_backward_batched(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs, size_t count,
                             size_t start, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    return ntt_backward(ctx, inout, inout, polys, poly_stride_limbs * ctx->n, r, EPI_NONE, Epilogue{}, as_stream(stream));
}

int pha_ntt_launch_phase(pha_context *ctx, uint64_t *inout, size_t polys, size_t poly_stride_limbs, size_t count,
                         size_t start, int inverse, int phase, pha_stream_t stream) {
    REQUIRE_CTX(inout);
    if (phase != 0 && phase != 1) return fail(PHA_ERR_INVALID_ARGUMENT, "phase must be 0 (column) or 1 (row)");
    NttRange r;
    PHA_TRY(make_range(ctx, count, start, 0, 0, 0, 0, r));
    const int mask = phase == 0 ? 1 : 2;
    if (inverse)
        return ntt_backward(ctx, inout, inout, polys, poly_stride_limbs * ctx->n, r, EPI_NONE, Epilogue{},
                            as_stream(stream), mask);
    return ntt_forward(ctx, inout, polys, poly_stride_limbs * ctx->n, r, EPI_NONE, Epilogue{}, as_stream(stream), mask);
}

int pha_ntt_fused_config(pha_context *ctx, int enable, uint32_t flags, uint32_t min_units) {
    REQUIRE_CTX(true);
    ctx->fused = enable ? 1 : 0;
    ctx->fused_flags = flags;
    ctx->fused_min_units = min_units;
    return PHA_OK;
}

int pha_ntt_fused_status(pha_context *ctx, uint32_t *error_word, uint32_t *queues) {
    REQUIRE_CTX(error_word);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(error_word, ctx->fused_ctl.p + kFCtlError, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (queues) *queues = ctx->fused_nq;
    return PHA_OK;
}

int pha_ntt_block_placement(pha_context *ctx, int limb_to_xcd) {
    REQUIRE_CTX(true);
    ctx->xcd_map = limb_to_xcd ? 1u : 0u;
    return PHA_OK;
}

int pha_modup(pha_context *ctx, size_t chain_index, uint64_t *dst, const uint64_t *cks, pha_stream_t stream) {
    REQUIRE_CTX(dst && cks);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, lt->size_Ql * ctx->n, &ws, as_stream(stream)));
    WsLease ws_lease(ctx, as_stream(stream));
    return do_modup(ctx, lt, dst, cks, ws, as_stream(stream));
}

int pha_key_switch_inner_prod(pha_context *ctx, size_t chain_index, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                              const uint64_t *const *evks, size_t reduction_threshold, pha_stream_t stream) {
    (void)reduction_threshold;
    REQUIRE_CTX(p_cx && p_t_mod_up && evks);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    return do_inner_prod(ctx, lt, p_cx, p_t_mod_up, evks, as_stream(stream));
}

int pha_moddown_from_ntt(pha_context *ctx, size_t chain_index, uint64_t *ct_i, uint64_t *cx_i, pha_stream_t stream) {
    REQUIRE_CTX(ct_i && cx_i);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, lt->size_Ql * ctx->n, &ws, as_stream(stream)));
    WsLease ws_lease(ctx, as_stream(stream));
    return do_moddown(ctx, lt, ct_i, cx_i, ws, nullptr, 1, as_stream(stream));
}

int pha_add_to_ct(pha_context *ctx, uint64_t *ct, const uint64_t *cx, size_t size_Ql, pha_stream_t stream) {
    REQUIRE_CTX(ct && cx);
    if (size_Ql == 0 || size_Ql > ctx->size_QP) return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    hipLaunchKernelGGL(add_to_ct_kernel, dim3(static_cast<uint32_t>(size_Ql) * (n / 512)), dim3(256), 0, as_stream(stream),
                       ct, cx, ctx->d_mod.p, n);
    return check_launch("add_to_ct");
}

int pha_keyswitch_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct, const uint64_t *c2,
                          const uint64_t *const *evks, pha_stream_t stream) {
    REQUIRE_CTX(ct && c2 && evks);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    return do_keyswitch(ctx, lt, ct, c2, evks, as_stream(stream));
}

int pha_divide_and_round_q_last_ntt(pha_context *ctx, size_t chain_index, uint64_t *src, size_t cipher_size,
                                    uint64_t *dst, pha_stream_t stream) {
    REQUIRE_CTX(src && dst);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    if (lt->size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    const size_t n = ctx->n, size_Ql = lt->size_Ql;
    NttRange r;
    PHA_TRY(make_range(ctx, 1, size_Ql - 1, 0, 0, 0, 0, r));
    PHA_TRY(ntt_backward(ctx, src, src, cipher_size, size_Ql * n, r, EPI_NONE, Epilogue{}, as_stream(stream)));
    return do_rescale(ctx, lt, src, src + (size_Ql - 1) * n, size_Ql * n, cipher_size, dst, as_stream(stream));
}

int pha_apply_galois_ntt(pha_context *ctx, const uint64_t *operand, size_t coeff_mod_size, uint32_t galois_elt,
                         uint64_t *result, pha_stream_t stream) {
    REQUIRE_CTX(operand && result);
    if (coeff_mod_size == 0) return fail(PHA_ERR_INVALID_ARGUMENT, "coeff_mod_size must be positive");
    const uint32_t *perm;
    PHA_TRY(get_galois_table(ctx, galois_elt, &perm));
    hipStream_t st = as_stream(stream);
    if (result == operand) {
        uint64_t *ws;
        PHA_TRY(ensure_workspace(ctx, coeff_mod_size * ctx->n, &ws, st));
        WsLease ws_lease(ctx, st);
        PHA_TRY(do_galois(ctx, operand, coeff_mod_size, perm, ws, st));
        HIP_TRY(hipMemcpyAsync(result, ws, coeff_mod_size * ctx->n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        return PHA_OK;
    }
    return do_galois(ctx, operand, coeff_mod_size, perm, result, st);
}

int pha_tensor_prod_2x2(pha_context *ctx, const uint64_t *operand1, const uint64_t *operand2, size_t coeff_mod_size,
                        uint64_t *result, pha_stream_t stream) {
    REQUIRE_CTX(operand1 && operand2 && result);
    if (coeff_mod_size == 0 || coeff_mod_size > ctx->size_QP)
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    hipLaunchKernelGGL((tensor_prod_2x2_kernel<false>), dim3(static_cast<uint32_t>(coeff_mod_size) * (n / 512)), dim3(256),
                       0, as_stream(stream), operand1, operand2, ctx->d_mod.p, result, n,
                       static_cast<uint32_t>(coeff_mod_size));
    return check_launch("tensor_prod_2x2");
}

int pha_tensor_prod_mxn(pha_context *ctx, const uint64_t *operand1, size_t op1_size, const uint64_t *operand2, size_t op2_size,
                        size_t coeff_mod_size, uint64_t *result, pha_stream_t stream) {
    REQUIRE_CTX(operand1 && operand2 && result);
    if (coeff_mod_size == 0 || coeff_mod_size > ctx->size_QP)
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    if (op1_size < 1 || op2_size < 1 || op1_size > 4 || op2_size > 4)
        return fail(PHA_ERR_INVALID_ARGUMENT, "tensor product of a size-%zu and a size-%zu ciphertext: sizes 1..4 are supported",
                    op1_size, op2_size);
    const uint32_t n = static_cast<uint32_t>(ctx->n), limbs = static_cast<uint32_t>(coeff_mod_size);
    const dim3 grid(limbs * (n / 256));
    hipStream_t st = as_stream(stream);
#define MXN_CASE(M, NN)                                                                                                       \
    if (op1_size == M && op2_size == NN) {                                                                                    \
        hipLaunchKernelGGL((tensor_prod_mxn_kernel<M, NN>), grid, dim3(256), 0, st, operand1, operand2, ctx->d_mod.p, result, n, \
                           limbs);                                                                                            \
        return check_launch("tensor_prod_mxn");                                                                               \
    }
    MXN_CASE(1, 1) MXN_CASE(1, 2) MXN_CASE(1, 3) MXN_CASE(1, 4) MXN_CASE(2, 1) MXN_CASE(2, 2) MXN_CASE(2, 3) MXN_CASE(2, 4)
    MXN_CASE(3, 1) MXN_CASE(3, 2) MXN_CASE(3, 3) MXN_CASE(3, 4) MXN_CASE(4, 1) MXN_CASE(4, 2) MXN_CASE(4, 3) MXN_CASE(4, 4)
#undef MXN_CASE
    return fail(PHA_ERR_LOGIC, "unreachable");
}

int pha_tensor_square_2x2(pha_context *ctx, const uint64_t *operand, size_t coeff_mod_size, uint64_t *result,
                          pha_stream_t stream) {
    REQUIRE_CTX(operand && result);
    if (coeff_mod_size == 0 || coeff_mod_size > ctx->size_QP)
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb range exceeds the modulus chain");
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    hipLaunchKernelGGL((tensor_prod_2x2_kernel<true>), dim3(static_cast<uint32_t>(coeff_mod_size) * (n / 512)), dim3(256),
                       0, as_stream(stream), operand, operand, ctx->d_mod.p, result, n,
                       static_cast<uint32_t>(coeff_mod_size));
    return check_launch("tensor_square_2x2");
}

int pha_multiply_and_relin_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct1, const uint64_t *ct2,
                                   const uint64_t *const *relin_evks, pha_stream_t stream) {
    REQUIRE_CTX(ct1 && ct2 && relin_evks);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t size_Ql = lt->size_Ql;
    if (ct1 == ct2)
        PHA_TRY(pha_tensor_square_2x2(ctx, ct1, size_Ql, ct1, stream));
    else
        PHA_TRY(pha_tensor_prod_2x2(ctx, ct1, ct2, size_Ql, ct1, stream));
    return do_keyswitch(ctx, lt, ct1, ct1 + 2 * size_Ql * ctx->n, relin_evks, as_stream(stream));
}

int pha_apply_galois_inplace(pha_context *ctx, size_t chain_index, uint64_t *ct, uint32_t galois_elt,
                             const uint64_t *const *galois_evks, pha_stream_t stream) {
    REQUIRE_CTX(ct && galois_evks);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const uint32_t *perm;
    PHA_TRY(get_galois_table(ctx, galois_elt, &perm));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    hipStream_t st = as_stream(stream);
    // workspace: [key-switch scratch][temp for both permuted polys]
    const size_t ks_words = (lt->beta * size_QlP + 2 * size_QlP + 2 * size_Ql) * n;
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, ks_words + 2 * size_Ql * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *temp = ws + ks_words;
    // c0 <- galois(c0); temp1 <- galois(c1); c1 <- 0; ct += KS(temp1)   (src/evaluate.cu:774-793): the moddown
    // epilogue writes ct = (galois(c0), 0) + KS directly — no copy of galois(c0), no memset of c1
    PHA_TRY(do_galois(ctx, ct, 2 * size_Ql, perm, temp, st));
    return do_keyswitch(ctx, lt, ct, temp + size_Ql * n, galois_evks, st, temp, 2);
}

int pha_rescale_to_next(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size, uint64_t *dst,
                        pha_stream_t stream) {
    REQUIRE_CTX(src && dst);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    if (lt->size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, cipher_size * ctx->n, &ws, as_stream(stream)));
    WsLease ws_lease(ctx, as_stream(stream));
    return rescale_with_scratch(ctx, lt, src, cipher_size, dst, ws, as_stream(stream));
}

int pha_rescale_to_next_noalloc(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size, uint64_t *dst,
                                pha_stream_t stream) {
    REQUIRE_CTX(src && dst);
    if (chain_index < 1 || chain_index > ctx->size_Q)
        return fail(PHA_ERR_INVALID_ARGUMENT, "chain_index %zu out of range [1, %zu]", chain_index, ctx->size_Q);
    // the level's constants must be on the device already: uploading them allocates
    auto it = ctx->levels.find(ctx->size_Q - (chain_index - 1));
    if (it == ctx->levels.end())
        return fail(PHA_ERR_LOGIC, "pha_rescale_to_next_noalloc: level %zu is not prepared (call pha_prepare_level first)",
                    chain_index);
    LevelTool *lt = it->second.get();
    if (lt->size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    if (cipher_size == 0 || cipher_size * ctx->n > kNoAllocScratchWords)
        return fail(PHA_ERR_INVALID_ARGUMENT, "pha_rescale_to_next_noalloc: cipher_size * N = %zu words exceeds the %zu-word scratch",
                    cipher_size * ctx->n, kNoAllocScratchWords);
    uint64_t *ws = ctx->device >= 0 && ctx->device < kMaxDevices ? g_scratch_ptr[ctx->device] : nullptr;
    if (!ws) return fail(PHA_ERR_LOGIC, "pha_rescale_to_next_noalloc: no scratch address for device %d", ctx->device);
    return rescale_with_scratch(ctx, lt, src, cipher_size, dst, ws, as_stream(stream));
}

int pha_prepare_level(pha_context *ctx, size_t chain_index) {
    REQUIRE_CTX(true);
    LevelTool *lt;
    return get_level(ctx, chain_index, &lt);
}

// ---- limb-parallel entry points (no counterpart in the single-GPU reference) ----
#define LP_PROLOGUE(ptr_ok)                          \
    REQUIRE_CTX(ptr_ok);                             \
    LevelTool *lt;                                   \
    PHA_TRY(get_level(ctx, chain_index, &lt));       \
    const LimbWindow w{q_begin, q_count, p_begin, p_count}; \
    PHA_TRY(check_window(ctx, lt, w))

int pha_lp_modup_begin(pha_context *ctx, size_t chain_index, const uint64_t *c2_ntt, uint64_t *c2_coeff, size_t c2_coeff_rows,
                       size_t row_begin, size_t q_begin, size_t q_count, pha_stream_t stream) {
    const size_t p_begin = 0, p_count = 0;
    LP_PROLOGUE(c2_ntt && c2_coeff);
    if (row_begin + q_count > c2_coeff_rows) return fail(PHA_ERR_INVALID_ARGUMENT, "rows exceed c2_coeff");
    return lp_modup_begin(ctx, lt, c2_ntt, c2_coeff, row_begin, w, as_stream(stream));
}

int pha_lp_modup_finish(pha_context *ctx, size_t chain_index, uint64_t *t_mod_up, const uint64_t *c2_ntt,
                        const uint64_t *c2_coeff, size_t c2_coeff_rows, const uint32_t *q_rows_host, size_t q_begin,
                        size_t q_count, size_t p_begin, size_t p_count, pha_stream_t stream) {
    LP_PROLOGUE(t_mod_up && c2_ntt && c2_coeff);
    if (c2_coeff_rows < lt->size_Ql) return fail(PHA_ERR_INVALID_ARGUMENT, "c2_coeff holds fewer limbs than Q_l");
    if (q_rows_host)
        for (size_t j = 0; j < lt->size_Ql; j++)
            if (q_rows_host[j] >= c2_coeff_rows) return fail(PHA_ERR_INVALID_ARGUMENT, "q_rows exceed c2_coeff");
    return lp_modup_finish(ctx, lt, t_mod_up, c2_ntt, c2_coeff, q_rows_host, w, as_stream(stream));
}

int pha_lp_key_switch_inner_prod(pha_context *ctx, size_t chain_index, uint64_t *p_cx, const uint64_t *p_t_mod_up,
                                 const uint64_t *const *rlk, size_t q_begin, size_t q_count, size_t p_begin,
                                 size_t p_count, pha_stream_t stream) {
    LP_PROLOGUE(p_cx && p_t_mod_up && rlk);
    return lp_inner_prod(ctx, lt, p_cx, p_t_mod_up, rlk, w, as_stream(stream));
}

int pha_lp_moddown_begin(pha_context *ctx, size_t chain_index, const uint64_t *cx, uint64_t *p_coeff, size_t p_coeff_rows,
                         size_t p_coeff_poly_stride_limbs, size_t row_begin, size_t p_begin, size_t p_count, pha_stream_t stream) {
    const size_t q_begin = 0, q_count = 0;
    LP_PROLOGUE(cx && p_coeff);
    if (p_count && p_coeff_poly_stride_limbs + row_begin + p_count > p_coeff_rows) return fail(PHA_ERR_INVALID_ARGUMENT, "rows exceed p_coeff");
    return lp_moddown_begin(ctx, lt, cx, p_coeff, p_coeff_poly_stride_limbs * ctx->n, row_begin, w, as_stream(stream));
}

int pha_lp_moddown_finish(pha_context *ctx, size_t chain_index, uint64_t *ct, uint64_t *cx, const uint64_t *p_coeff,
                          size_t p_coeff_rows, size_t p_coeff_poly_stride_limbs, const uint32_t *p_rows_host, size_t q_begin,
                          size_t q_count, pha_stream_t stream) {
    const size_t p_begin = 0, p_count = 0;
    LP_PROLOGUE(ct && cx && p_coeff);
    for (size_t k = 0; k < ctx->size_P; k++)
        if ((p_rows_host ? p_rows_host[k] : k) + p_coeff_poly_stride_limbs >= p_coeff_rows)
            return fail(PHA_ERR_INVALID_ARGUMENT, "p_rows exceed p_coeff");
    return lp_moddown_finish(ctx, lt, ct, cx, p_coeff, p_coeff_poly_stride_limbs * ctx->n, p_rows_host, w, as_stream(stream));
}

int pha_lp_tensor_prod_2x2(pha_context *ctx, const uint64_t *operand1, const uint64_t *operand2, size_t coeff_mod_size,
                           size_t q_begin, size_t q_count, uint64_t *result, pha_stream_t stream) {
    REQUIRE_CTX(operand1 && operand2 && result);
    if (coeff_mod_size == 0 || coeff_mod_size > ctx->size_QP || q_begin + q_count > coeff_mod_size)
        return fail(PHA_ERR_INVALID_ARGUMENT, "limb window exceeds the ciphertext");
    if (!q_count) return PHA_OK;
    const uint32_t n = static_cast<uint32_t>(ctx->n);
    if (operand1 == operand2)
        hipLaunchKernelGGL((tensor_prod_2x2_kernel<true>), dim3(static_cast<uint32_t>(q_count) * (n / 512)), dim3(256), 0,
                           as_stream(stream), operand1, operand2, ctx->d_mod.p, result, n, static_cast<uint32_t>(coeff_mod_size),
                           0u, static_cast<uint32_t>(q_begin));
    else
        hipLaunchKernelGGL((tensor_prod_2x2_kernel<false>), dim3(static_cast<uint32_t>(q_count) * (n / 512)), dim3(256), 0,
                           as_stream(stream), operand1, operand2, ctx->d_mod.p, result, n, static_cast<uint32_t>(coeff_mod_size),
                           0u, static_cast<uint32_t>(q_begin));
    return check_launch("tensor_prod_2x2 (limb window)");
}

int pha_lp_rescale_begin(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size,
                         uint64_t *last_coeff, pha_stream_t stream) {
    REQUIRE_CTX(src && last_coeff);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    if (lt->size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    const size_t n = ctx->n, size_Ql = lt->size_Ql;
    NttRange r;
    PHA_TRY(make_range(ctx, 1, size_Ql - 1, 0, 0, 0, 0, r));
    return ntt_backward(ctx, last_coeff, src, cipher_size, size_Ql * n, r, EPI_NONE, Epilogue{}, as_stream(stream), 3,
                        size_Ql - 1, n);
}

int pha_lp_rescale_finish(pha_context *ctx, size_t chain_index, const uint64_t *src, const uint64_t *last_coeff,
                          size_t cipher_size, uint64_t *dst, size_t q_begin, size_t q_count, pha_stream_t stream) {
    const size_t p_begin = 0, p_count = 0;
    LP_PROLOGUE(src && last_coeff && dst);
    if (lt->size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    return lp_rescale_finish(ctx, lt, src, last_coeff, cipher_size, dst, w, as_stream(stream));
}

// ---- key generation, symmetric encryption, decryption -----------------------
int pha_set_sampler_key(const uint8_t *key32) {
    std::lock_guard<std::mutex> lock(g_sampler_mutex);
    if (key32) {
        memcpy(g_sampler_key.w, key32, sizeof(g_sampler_key.w));
    } else {
        PHA_TRY(os_entropy(g_sampler_key.w, sizeof(g_sampler_key.w)));
    }
    g_sampler_key_set = true;
    return PHA_OK;
}

int pha_random_nonce(uint64_t *out) {
    if (!out) return fail(PHA_ERR_INVALID_ARGUMENT, "null argument");
    return os_entropy(out, sizeof(*out));
}

int pha_gen_secretkey(pha_context *ctx, uint64_t seed, uint64_t *sk, pha_stream_t stream) {
    REQUIRE_CTX(sk);
    hipStream_t st = as_stream(stream);
    PHA_TRY(launch_sample(ctx, SAMPLE_TERNARY, sk, ctx->size_QP, seed, st));
    NttRange r;
    PHA_TRY(make_range(ctx, ctx->size_QP, 0, 0, 0, 0, 0, r));
    return ntt_forward(ctx, sk, 1, 0, r, EPI_NONE, Epilogue{}, st);
}

int pha_encrypt_zero_symmetric(pha_context *ctx, const uint64_t *sk, size_t chain_index, uint64_t seed_a, uint64_t seed_e,
                               uint64_t *ct, pha_stream_t stream) {
    REQUIRE_CTX(sk && ct);
    return do_encrypt_zero(ctx, sk, chain_index, seed_a, seed_e, ct, as_stream(stream));
}

int pha_encrypt_symmetric(pha_context *ctx, const uint64_t *sk, const uint64_t *plain, size_t chain_index, uint64_t seed_a,
                          uint64_t seed_e, uint64_t *ct, pha_stream_t stream) {
    REQUIRE_CTX(sk && plain && ct);
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    PHA_TRY(do_encrypt_zero(ctx, sk, chain_index, seed_a, seed_e, ct, as_stream(stream)));
    return launch_ew(ctx, EW_ADD, ct, ct, plain, nullptr, nullptr, 0, limbs, as_stream(stream));
}

int pha_gen_publickey(pha_context *ctx, const uint64_t *sk, uint64_t seed, uint64_t *pk, pha_stream_t stream) {
    REQUIRE_CTX(sk && pk);
    uint64_t seed_a, seed_e;
    PHA_TRY(host_rnd_at(seed, 0, seed_a));
    PHA_TRY(host_rnd_at(seed, 1, seed_e));
    return do_encrypt_zero(ctx, sk, 0, seed_a, seed_e, pk, as_stream(stream));
}

int pha_encrypt_zero_asymmetric(pha_context *ctx, const uint64_t *pk, size_t chain_index, uint64_t seed, uint64_t *ct,
                                pha_stream_t stream) {
    REQUIRE_CTX(pk && ct);
    if (chain_index == 0) return fail(PHA_ERR_INVALID_ARGUMENT, "asymmetric encryption targets a data level (chain_index >= 1)");
    if (ctx->size_P == 0) return fail(PHA_ERR_LOGIC, "asymmetric encryption needs a special modulus");
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    hipStream_t st = as_stream(stream);
    const size_t n = ctx->n, QP = ctx->size_QP, Q = ctx->size_Q;
    LevelTool *lt;
    PHA_TRY(get_level(ctx, 1, &lt));
    // workspace: u[QP] | cx[2][QP] | delta[2][Q] | top[2][Q]
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, (3 * QP + 4 * Q) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *u = ws, *cx = u + QP * n, *delta = cx + 2 * QP * n, *top = delta + 2 * Q * n;
    NttRange r;
    PHA_TRY(make_range(ctx, QP, 0, 0, 0, 0, 0, r));
    uint64_t sub[3];
    for (uint64_t i = 0; i < 3; i++) PHA_TRY(host_rnd_at(seed, i, sub[i]));
    PHA_TRY(launch_sample(ctx, SAMPLE_TERNARY, u, QP, sub[0], st));
    PHA_TRY(ntt_forward(ctx, u, 1, 0, r, EPI_NONE, Epilogue{}, st));
    for (size_t i = 0; i < 2; i++) PHA_TRY(launch_sample(ctx, SAMPLE_ERROR, cx + i * QP * n, QP, sub[1 + i], st));
    PHA_TRY(ntt_forward(ctx, cx, 2, QP * n, r, EPI_NONE, Epilogue{}, st));
    for (size_t i = 0; i < 2; i++)   // cx_i = u * pk_i + e_i
        PHA_TRY(launch_ew(ctx, EW_MUL_ADD, cx + i * QP * n, u, pk + i * QP * n, cx + i * QP * n, nullptr, 0, QP, st));
    uint64_t *dst = limbs == Q ? ct : top;
    PHA_TRY(do_moddown(ctx, lt, dst, cx, delta, nullptr, 2, st));
    if (limbs != Q)  // lower levels: drop limbs (CKKS mod_switch_to_next)
        HIP_TRY(hipMemcpy2DAsync(ct, limbs * n * sizeof(uint64_t), top, Q * n * sizeof(uint64_t), limbs * n * sizeof(uint64_t), 2,
                                 hipMemcpyDeviceToDevice, st));
    return PHA_OK;
}

int pha_encrypt_asymmetric(pha_context *ctx, const uint64_t *pk, const uint64_t *plain, size_t chain_index, uint64_t seed,
                           uint64_t *ct, pha_stream_t stream) {
    REQUIRE_CTX(pk && plain && ct);
    PHA_TRY(pha_encrypt_zero_asymmetric(ctx, pk, chain_index, seed, ct, stream));
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    return launch_ew(ctx, EW_ADD, ct, ct, plain, nullptr, nullptr, 0, limbs, as_stream(stream));
}

int pha_ckks_decrypt(pha_context *ctx, const uint64_t *sk, const uint64_t *ct, size_t cipher_size, size_t chain_index,
                     uint64_t *plain, pha_stream_t stream) {
    REQUIRE_CTX(sk && ct && plain);
    if (cipher_size < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "encrypted is not valid for encryption parameters");
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    hipStream_t st = as_stream(stream);
    const size_t n = ctx->n;
    // plain = c0 + c1*s (+ c2*s^2 ...): powers of s kept in the workspace
    PHA_TRY(launch_ew(ctx, EW_MUL_ADD, plain, ct + limbs * n, sk, ct, nullptr, 0, limbs, st));
    if (cipher_size > 2) {
        uint64_t *sp;
        PHA_TRY(ensure_workspace(ctx, limbs * n, &sp, st));
        WsLease ws_lease(ctx, st);
        const uint64_t *prev = sk;
        for (size_t i = 2; i < cipher_size; i++) {
            PHA_TRY(launch_ew(ctx, EW_MUL, sp, prev, sk, nullptr, nullptr, 0, limbs, st));
            PHA_TRY(launch_ew(ctx, EW_MUL_ADD, plain, ct + i * limbs * n, sp, plain, nullptr, 0, limbs, st));
            prev = sp;
        }
    }
    return PHA_OK;
}

int pha_generate_one_kswitch_key(pha_context *ctx, const uint64_t *sk, const uint64_t *new_key, uint64_t seed,
                                 uint64_t *const *evks_host, pha_stream_t stream) {
    REQUIRE_CTX(sk && new_key && evks_host);
    return do_kswitch_key(ctx, sk, new_key, seed, evks_host, as_stream(stream));
}

int pha_gen_relinkey(pha_context *ctx, const uint64_t *sk, uint64_t seed, uint64_t *const *evks_host, pha_stream_t stream) {
    REQUIRE_CTX(sk && evks_host);
    uint64_t *s2;
    PHA_TRY(ensure_workspace(ctx, ctx->size_Q * ctx->n, &s2, as_stream(stream)));
    WsLease ws_lease(ctx, as_stream(stream));
    PHA_TRY(launch_ew(ctx, EW_MUL, s2, sk, sk, nullptr, nullptr, 0, ctx->size_Q, as_stream(stream)));
    return do_kswitch_key(ctx, sk, s2, seed, evks_host, as_stream(stream));
}

int pha_gen_galois_key(pha_context *ctx, const uint64_t *sk, uint32_t galois_elt, uint64_t seed, uint64_t *const *evks_host,
                       pha_stream_t stream) {
    REQUIRE_CTX(sk && evks_host);
    const uint32_t *perm;
    PHA_TRY(get_galois_table(ctx, galois_elt, &perm));
    uint64_t *sg;
    PHA_TRY(ensure_workspace(ctx, ctx->size_Q * ctx->n, &sg, as_stream(stream)));
    WsLease ws_lease(ctx, as_stream(stream));
    PHA_TRY(do_galois(ctx, sk, ctx->size_Q, perm, sg, as_stream(stream)));
    return do_kswitch_key(ctx, sk, sg, seed, evks_host, as_stream(stream));
}

// ---- element-wise evaluator arithmetic (src/evaluate.cu:20-105,442-560,618-700) ----------------
namespace {
int poly_binary(pha_context *ctx, int op, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size,
                size_t polys, size_t b_polys, hipStream_t st, const uint64_t *c = nullptr) {
    if (coeff_mod_size == 0 || coeff_mod_size > ctx->size_QP || polys == 0 || (op != PB_NEG && b_polys != polys && b_polys != 1))
        return fail(PHA_ERR_INVALID_ARGUMENT, "poly number mismatch");
    const uint32_t n = static_cast<uint32_t>(ctx->n), limbs = static_cast<uint32_t>(coeff_mod_size);
    const dim3 grid(static_cast<uint32_t>(polys) * limbs * (n / 512));
    const uint32_t bc = (b_polys == 1 && polys > 1) ? 1u : 0u;
#define PB_CASE(OP) case OP: hipLaunchKernelGGL((poly_binary_kernel<OP>), grid, dim3(256), 0, st, dst, a, b, ctx->d_mod.p, limbs, n, bc, c); break;
    switch (op) { PB_CASE(PB_ADD) PB_CASE(PB_SUB) PB_CASE(PB_MUL) PB_CASE(PB_NEG) PB_CASE(PB_MAC) }
#undef PB_CASE
    return check_launch("poly_binary");
}
}  // namespace

int pha_add_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size, size_t polys,
                     size_t b_polys, pha_stream_t stream) {
    REQUIRE_CTX(dst && a && b);
    return poly_binary(ctx, PB_ADD, dst, a, b, coeff_mod_size, polys, b_polys, as_stream(stream));
}
int pha_sub_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size, size_t polys,
                     size_t b_polys, pha_stream_t stream) {
    REQUIRE_CTX(dst && a && b);
    return poly_binary(ctx, PB_SUB, dst, a, b, coeff_mod_size, polys, b_polys, as_stream(stream));
}
int pha_multiply_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, size_t coeff_mod_size,
                          size_t polys, size_t b_polys, pha_stream_t stream) {
    REQUIRE_CTX(dst && a && b);
    return poly_binary(ctx, PB_MUL, dst, a, b, coeff_mod_size, polys, b_polys, as_stream(stream));
}
int pha_multiply_add_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                              size_t coeff_mod_size, size_t polys, size_t b_polys, pha_stream_t stream) {
    REQUIRE_CTX(dst && a && b && c);
    return poly_binary(ctx, PB_MAC, dst, a, b, coeff_mod_size, polys, b_polys, as_stream(stream), c);
}
int pha_negate_rns_poly(pha_context *ctx, uint64_t *dst, const uint64_t *a, size_t coeff_mod_size, size_t polys,
                        pha_stream_t stream) {
    REQUIRE_CTX(dst && a);
    return poly_binary(ctx, PB_NEG, dst, a, nullptr, coeff_mod_size, polys, polys, as_stream(stream));
}
int pha_mod_switch_drop_to_next(pha_context *ctx, size_t chain_index, const uint64_t *src, size_t cipher_size, uint64_t *dst,
                                pha_stream_t stream) {
    REQUIRE_CTX(src && dst);
    if (chain_index < 1 || chain_index >= ctx->size_Q)
        return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    const size_t l = ctx->size_Q - (chain_index - 1), n = ctx->n;
    HIP_TRY(hipMemcpy2DAsync(dst, (l - 1) * n * sizeof(uint64_t), src, l * n * sizeof(uint64_t), (l - 1) * n * sizeof(uint64_t),
                             cipher_size, hipMemcpyDeviceToDevice, as_stream(stream)));
    return PHA_OK;
}

// ---- CKKS encoder / decoder (src/ckks.cu:130-283) ----------------------------------------------
namespace {
// with_special: the plaintext is written over Q_l u P ([size_Ql + size_P][N]) instead of Q_l
int do_ckks_encode(pha_context *ctx, const double *values_host, size_t values_size, size_t sparse_slots, size_t chain_index,
                   double scale, uint64_t *plain, bool with_special, pha_stream_t stream) {
    REQUIRE_CTX(plain && (values_host || values_size == 0));
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    if (with_special && (chain_index == 0 || ctx->size_P == 0))
        return fail(PHA_ERR_INVALID_ARGUMENT, "a plaintext over Q_l u P needs a data level and a special modulus");
    if (values_size > ctx->n / 2) return fail(PHA_ERR_INVALID_ARGUMENT, "values_size is too large");
    pha_context::EncoderLevel *lv;
    PHA_TRY(get_encoder_level(ctx, limbs, &lv));
    if (!(scale > 0) || static_cast<int>(std::log2(scale)) + 1 >= lv->total_bits)
        return fail(PHA_ERR_INVALID_ARGUMENT, "scale out of bounds (encode: scale %g, %d modulus bits at chain index %zu)", scale,
                    lv->total_bits, chain_index);
    size_t s = sparse_slots;
    if (s == 0)
        for (s = 1; s < values_size; s <<= 1) {}
    PHA_TRY(check_slots(ctx, s));
    if (values_size > s) return fail(PHA_ERR_INVALID_ARGUMENT, "values_size exceeds sparse_slots");
    hipStream_t st = as_stream(stream);
    const uint32_t n = static_cast<uint32_t>(ctx->n), S = static_cast<uint32_t>(s);
    uint32_t log_s = 0;
    while ((1u << log_s) < S) log_s++;
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, 4 * s + 2, &ws, st));
    WsLease ws_lease(ctx, st);
    double2 *buf = reinterpret_cast<double2 *>(ws), *tmp = buf + s;
    unsigned long long *d_max = reinterpret_cast<unsigned long long *>(tmp + s);
    HIP_TRY(hipMemsetAsync(buf, 0, s * sizeof(double2), st));
    if (values_size) HIP_TRY(hipMemcpyAsync(buf, values_host, values_size * sizeof(double2), hipMemcpyHostToDevice, st));
    // u = (scale / s) * bitrev(stages(z)): inverse special FFT
    special_fft_stages<true>(ctx, buf, S, st);
    hipLaunchKernelGGL(bit_reverse_scale_kernel, dim3((S + 255) / 256), dim3(256), 0, st, tmp, buf, S, S, log_s,
                       scale / static_cast<double>(s));
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(max_abs_kernel, dim3((S + 255) / 256), dim3(256), 0, st, tmp, S, d_max);
    PHA_TRY(check_launch("special fft"));
    unsigned long long max_bits = 0;
    HIP_TRY(hipMemcpyAsync(&max_bits, d_max, sizeof(max_bits), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double max_coeff;
    memcpy(&max_coeff, &max_bits, sizeof(double));
    // one extra bit for the sign (src/ckks.cu:196-203)
    const int max_coeff_bit_count = static_cast<int>(std::ceil(std::log2(std::max(max_coeff, 1.0)))) + 1;
    if (!(max_coeff == max_coeff) || max_coeff_bit_count >= lv->total_bits)
        return fail(PHA_ERR_INVALID_ARGUMENT, "encoded values are too large");
    const size_t out_limbs = limbs + (with_special ? ctx->size_P : 0);
    HIP_TRY(hipMemsetAsync(plain, 0, out_limbs * n * sizeof(uint64_t), st));
    hipLaunchKernelGGL(ckks_decompose_kernel, dim3((2 * S + 255) / 256), dim3(256), 0, st, plain, tmp, ctx->d_mod.p,
                       static_cast<uint32_t>(out_limbs), n, S, n / (2 * S),
                       static_cast<uint32_t>(chain_index == 0 ? out_limbs : limbs), static_cast<uint32_t>(ctx->size_Q));
    PHA_TRY(check_launch("ckks_decompose"));
    NttRange r;
    if (with_special)
        PHA_TRY(make_range(ctx, out_limbs, 0, ctx->size_QP, ctx->size_P, 0, 0, r));
    else
        PHA_TRY(make_range(ctx, limbs, 0, 0, 0, 0, 0, r));
    return ntt_forward(ctx, plain, 1, 0, r, EPI_NONE, Epilogue{}, st);
}
}  // namespace

int pha_ckks_encode(pha_context *ctx, const double *values_host, size_t values_size, size_t sparse_slots, size_t chain_index,
                    double scale, uint64_t *plain, pha_stream_t stream) {
    return do_ckks_encode(ctx, values_host, values_size, sparse_slots, chain_index, scale, plain, false, stream);
}

int pha_ckks_encode_qp(pha_context *ctx, const double *values_host, size_t values_size, size_t sparse_slots, size_t chain_index,
                       double scale, uint64_t *plain, pha_stream_t stream) {
    return do_ckks_encode(ctx, values_host, values_size, sparse_slots, chain_index, scale, plain, true, stream);
}


int pha_ckks_decode(pha_context *ctx, const uint64_t *plain, size_t chain_index, double scale, size_t sparse_slots,
                    double *values_host, pha_stream_t stream) {
    REQUIRE_CTX(plain && values_host);
    size_t limbs;
    PHA_TRY(level_limbs(ctx, chain_index, &limbs));
    pha_context::EncoderLevel *lv;
    PHA_TRY(get_encoder_level(ctx, limbs, &lv));
    if (!(scale > 0) || static_cast<int>(std::log2(scale)) >= lv->total_bits)
        return fail(PHA_ERR_INVALID_ARGUMENT, "scale out of bounds (decode: scale %g, %d modulus bits at chain index %zu)", scale,
                    lv->total_bits, chain_index);
    PHA_TRY(check_slots(ctx, sparse_slots));
    hipStream_t st = as_stream(stream);
    const uint32_t n = static_cast<uint32_t>(ctx->n), S = static_cast<uint32_t>(sparse_slots);
    uint32_t log_s = 0;
    while ((1u << log_s) < S) log_s++;
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, limbs * n + 4 * sparse_slots, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *copy = ws;
    double2 *buf = reinterpret_cast<double2 *>(ws + limbs * n), *tmp = buf + sparse_slots;
    NttRange r;
    PHA_TRY(make_range(ctx, limbs, 0, 0, 0, 0, 0, r));
    PHA_TRY(ntt_backward(ctx, copy, plain, 1, 0, r, EPI_NONE, Epilogue{}, st));
    int inv_exp;
    const double inv_mant = std::frexp(1.0 / scale, &inv_exp);
    hipLaunchKernelGGL(ckks_compose_kernel, dim3((2 * S + 63) / 64), dim3(64), 0, st, tmp, copy, ctx->d_mod.p, lv->big_q.p,
                       lv->half_q.p, lv->qhat.p, lv->inv_qhat.p, static_cast<uint32_t>(limbs), n, S, n / (2 * S), inv_mant, inv_exp);
    // z = stages(bitrev(u)): forward special FFT
    hipLaunchKernelGGL(bit_reverse_scale_kernel, dim3((S + 255) / 256), dim3(256), 0, st, buf, tmp, S, S, log_s, 1.0);
    special_fft_stages<false>(ctx, buf, S, st);
    PHA_TRY(check_launch("ckks decode"));
    HIP_TRY(hipMemcpyAsync(values_host, buf, sparse_slots * sizeof(double2), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PHA_OK;
}

// ---- hoisted rotations (no counterpart in the reference; see oracle.c "Hoisted rotations") ------
int pha_rotate_hoisted(pha_context *ctx, size_t chain_index, const uint64_t *ct, size_t n_rot, const uint32_t *galois_elts_host,
                       const uint64_t *const *const *evks_host, uint64_t *const *out_host, pha_stream_t stream) {
    REQUIRE_CTX(ct && galois_elts_host && evks_host && out_host);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    hipStream_t st = as_stream(stream);
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, (lt->beta * size_QlP + 2 * size_QlP + 2 * size_Ql) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *t_mod_up = ws, *cx = t_mod_up + lt->beta * size_QlP * n, *scratch = cx + 2 * size_QlP * n;
    const bool in_place_own = ctx->size_P > 1;   // as in do_keyswitch
    PHA_TRY(do_modup(ctx, lt, t_mod_up, ct + size_Ql * n, scratch, st, !in_place_own));  // once for every rotation
    for (size_t r = 0; r < n_rot; r++) {
        if (!out_host[r] || !evks_host[r] || out_host[r] == ct) return fail(PHA_ERR_INVALID_ARGUMENT, "null or aliasing rotation output");
        const uint32_t *perm;
        PHA_TRY(get_galois_table(ctx, galois_elts_host[r], &perm));
        PHA_TRY(do_inner_prod(ctx, lt, cx, t_mod_up, evks_host[r], st, perm, in_place_own ? ct + size_Ql * n : nullptr));
        PHA_TRY(do_galois(ctx, ct, size_Ql, perm, out_host[r], st));
        PHA_TRY(do_moddown(ctx, lt, out_host[r], cx, scratch, out_host[r], 2, st, 2));  // poly 0 += pi(c0), poly 1 as is
    }
    return PHA_OK;
}

// Double-hoisted baby-step / giant-step linear transform (Bossuat et al., "Efficient bootstrapping for approximate
// homomorphic encryption with non-sparse keys", algorithm 6; the reference's *_hoisting drivers stop at sharing the
// modup): out = sum_g rot_{G_g}( sum_b pt[g][b] * rot_{B_b}(ct) ).
//   babies:  ONE modup of c1; per baby an inner product with the automorphism applied on the fly, result kept
//            over Q_l u P with P * pi(c0) added on the Q_l limbs — no moddown;
//   giants:  the inner sum is one multi-term multiply-accumulate over Q_l u P (plaintexts encoded over Q_l u P by
//            pha_ckks_encode_qp), ONE moddown (both polys), then for G != identity modup + inner product, the
//            results accumulated over Q_l u P;
//   end:     ONE moddown of the total.
// Moddowns: n_giant + 1 instead of n_baby + n_giant.  Galois element 1 = no rotation (its key may be null).
int pha_linear_transform_double_hoisted(pha_context *ctx, size_t chain_index, const uint64_t *ct, size_t n_baby,
                                        const uint32_t *baby_elts_host, const uint64_t *const *const *baby_evks_host,
                                        size_t n_giant, const uint32_t *giant_elts_host,
                                        const uint64_t *const *const *giant_evks_host, const uint64_t *const *plains_host,
                                        uint64_t *out, pha_stream_t stream) {
    REQUIRE_CTX(ct && baby_elts_host && baby_evks_host && giant_elts_host && giant_evks_host && plains_host && out);
    if (n_baby == 0 || n_baby > kMaxMacTerms || n_giant == 0)
        return fail(PHA_ERR_INVALID_ARGUMENT, "baby steps must number 1..%d, giant steps at least 1", kMaxMacTerms);
    if (ctx->size_P == 0) return fail(PHA_ERR_LOGIC, "key switching needs a special modulus");
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    hipStream_t st = as_stream(stream);
    if (!ctx->d_bigP_mod_q.p) {
        LevelConstants lc;
        build_level_constants(ctx->moduli, ctx->size_P, ctx->size_Q, lc);
        HIP_TRY(ctx->d_bigP_mod_q.upload(lc.bigP_mod_q));
    }
    // Giant steps that rotate are independent chains (moddown -> modup -> inner product) of small, latency-bound
    // launches: with two or more of them, every other chain runs on the context's helper stream with its own buffers.
    size_t rotating = 0;
    for (size_t g = 0; g < n_giant; g++) rotating += giant_elts_host[g] != 1;
    // (PHA_GIANT_STREAMS = 1 .. 4 sets the number of chains, default 2, 1 = all on the caller's stream; measured on the
    // CNN's bootstrappings: 2 chains 1-2 % over 1, 4 chains the same as 2)
    static const size_t max_chains = [] {
        const char *e = getenv("PHA_GIANT_STREAMS");
        const int v = e ? atoi(e) : 2;
        return static_cast<size_t>(std::min(1 + pha_context::kAuxStreams, std::max(1, v)));
    }();
    const size_t n_chains = std::max<size_t>(1, std::min(max_chains, rotating));
    const bool two = n_chains >= 2;
    if (two && !ctx->aux_fork) HIP_TRY(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
    for (size_t c = 1; c < n_chains; c++)
        if (!ctx->aux_stream[c - 1]) {
            HIP_TRY(hipStreamCreateWithFlags(&ctx->aux_stream[c - 1], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ctx->aux_join[c - 1], hipEventDisableTiming));
        }
    const size_t chain_words = ((lt->beta + 6) * size_QlP + 4 * size_Ql) * n;   // t_mod_up | cx | acc | total | scratch | u
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, n_chains * chain_words + 2 * n_baby * size_QlP * n, &ws, st));
    WsLease ws_lease(ctx, st);
    struct Chain {
        hipStream_t st;
        uint64_t *t_mod_up, *cx, *acc, *total, *scratch, *u;
        bool first;
    } chains[1 + pha_context::kAuxStreams];
    for (size_t c = 0; c < n_chains; c++) {
        uint64_t *base = ws + c * chain_words;
        chains[c] = Chain{c ? ctx->aux_stream[c - 1] : st, base, base + lt->beta * size_QlP * n, base + (lt->beta + 2) * size_QlP * n,
                          base + (lt->beta + 4) * size_QlP * n, base + (lt->beta + 6) * size_QlP * n,
                          base + (lt->beta + 6) * size_QlP * n + 2 * size_Ql * n, true};
    }
    uint64_t *rot = ws + n_chains * chain_words;
    const uint32_t n32 = static_cast<uint32_t>(n), ql = static_cast<uint32_t>(size_Ql), qlp = static_cast<uint32_t>(size_QlP),
                   sq = static_cast<uint32_t>(ctx->size_Q);
    const dim3 grid_qp(qlp * (n32 / 512), 2), grid_qp1(qlp * (n32 / 512), 1);
    const DModulus *mod = ctx->d_mod.p;
    // dst_qp = key switch of src's c1 under elt (ch.t_mod_up must hold modup(src c1)) + lifted pi(c0), on ch's stream
    auto rotate_qp = [&](Chain &ch, uint64_t *dst_qp, const uint64_t *src, uint32_t elt, const uint64_t *const *evks) -> int {
        const uint32_t *perm = nullptr;
        if (elt == 1) {
            hipLaunchKernelGGL((lift_qp_kernel<true>), grid_qp, dim3(256), 0, ch.st, dst_qp, src, perm, ctx->d_bigP_mod_q.p, mod, n32, ql, qlp);
            return check_launch("lift_qp");
        }
        PHA_TRY(get_galois_table(ctx, elt, &perm));
        if (!evks) return fail(PHA_ERR_INVALID_ARGUMENT, "null Galois key");
        PHA_TRY(do_inner_prod(ctx, lt, dst_qp, ch.t_mod_up, evks, ch.st, perm));
        hipLaunchKernelGGL((lift_qp_kernel<false>), grid_qp1, dim3(256), 0, ch.st, dst_qp, src, perm, ctx->d_bigP_mod_q.p, mod, n32, ql, qlp);
        return check_launch("lift_qp");
    };
    auto accumulate = [&](Chain &ch, const uint64_t *add) -> int {
        if (ch.first)
            hipLaunchKernelGGL((add_qp_kernel<true>), grid_qp, dim3(256), 0, ch.st, ch.total, add, mod, n32, ql, qlp, sq);
        else
            hipLaunchKernelGGL((add_qp_kernel<false>), grid_qp, dim3(256), 0, ch.st, ch.total, add, mod, n32, ql, qlp, sq);
        ch.first = false;
        return check_launch("add_qp");
    };
    Chain &main_chain = chains[0];
    bool any_rot = false;
    for (size_t b = 0; b < n_baby; b++) any_rot = any_rot || baby_elts_host[b] != 1;
    if (any_rot) PHA_TRY(do_modup(ctx, lt, main_chain.t_mod_up, ct + size_Ql * n, main_chain.scratch, st));
    for (size_t b = 0; b < n_baby; b++)
        PHA_TRY(rotate_qp(main_chain, rot + 2 * b * size_QlP * n, ct, baby_elts_host[b], baby_evks_host[b]));
    if (two) {  // the helper streams start once the baby rotations exist
        HIP_TRY(hipEventRecord(ctx->aux_fork, st));
        for (size_t c = 1; c < n_chains; c++) HIP_TRY(hipStreamWaitEvent(ctx->aux_stream[c - 1], ctx->aux_fork, 0));
    }
    // Once the helper streams have been forked they are joined below whatever happens in between: a helper left
    // unjoined would keep running on this context's workspace after the call has returned its error, and under stream
    // capture it would make the caller's hipStreamEndCapture fail (hipErrorStreamCaptureUnjoined).
    auto giant_steps = [&]() -> int {
    size_t rot_index = 0;
    for (size_t g = 0; g < n_giant; g++) {
        MacTerms terms{};
        for (size_t b = 0; b < n_baby; b++) {
            const uint64_t *pt = plains_host[g * n_baby + b];
            if (!pt) continue;
            terms.rot[terms.count] = rot + 2 * b * size_QlP * n;
            terms.pt[terms.count++] = pt;
        }
        if (!terms.count) continue;
        const bool rotates = giant_elts_host[g] != 1;
        Chain &ch = chains[(two && rotates) ? (rot_index++ % n_chains) : 0];
        hipLaunchKernelGGL(multi_mac_qp_kernel, dim3(qlp * (n32 / 512)), dim3(256), 0, ch.st, ch.acc, terms, mod, n32, ql, qlp, sq);
        PHA_TRY(check_launch("multi_mac_qp"));
        const uint64_t *add = ch.acc;
        if (rotates) {
            PHA_TRY(do_moddown(ctx, lt, ch.u, ch.acc, ch.scratch, nullptr, 2, ch.st));
            PHA_TRY(do_modup(ctx, lt, ch.t_mod_up, ch.u + size_Ql * n, ch.scratch, ch.st));
            PHA_TRY(rotate_qp(ch, ch.cx, ch.u, giant_elts_host[g], giant_evks_host[g]));
            add = ch.cx;
        }
        PHA_TRY(accumulate(ch, add));
    }
    return PHA_OK;
    };
    int rc = giant_steps();
    for (size_t c = 1; c < n_chains; c++) {  // join: the helper chains' partial sums go into the main one
        const hipError_t e1 = hipEventRecord(ctx->aux_join[c - 1], ctx->aux_stream[c - 1]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, ctx->aux_join[c - 1], 0) : e1;
        if (e2 != hipSuccess && rc == PHA_OK) rc = fail(PHA_ERR_HIP, "joining helper stream %zu: %s", c, hipGetErrorString(e2));
        if (rc == PHA_OK && !chains[c].first) rc = accumulate(main_chain, chains[c].total);
    }
    if (rc != PHA_OK) return rc;
    if (main_chain.first) return fail(PHA_ERR_INVALID_ARGUMENT, "the transform has no plaintext");
    return do_moddown(ctx, lt, out, main_chain.total, main_chain.scratch, nullptr, 2, st);
}

// apply_galois_inplace on `batch` independent ciphertexts that share the Galois element and its key
// (BASELINE configs[3]: rotate x 64).  Every step is one launch over the whole batch: larger grids run the NTT
// kernels in their steady state and the ciphertexts of a batch read each key tile together.
int pha_apply_galois_batched(pha_context *ctx, size_t chain_index, const uint64_t *in, uint64_t *out, size_t batch,
                             uint32_t galois_elt, const uint64_t *const *galois_evks, pha_stream_t stream) {
    REQUIRE_CTX(in && out && galois_evks);
    if (batch == 0) return PHA_OK;
    if (in == out) return fail(PHA_ERR_INVALID_ARGUMENT, "the batched rotation is out of place");
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P, ct_words = 2 * size_Ql * n;
    hipStream_t st = as_stream(stream);
    const uint32_t *perm;
    PHA_TRY(get_galois_table(ctx, galois_elt, &perm));
    if (!lt->uniform_digits && !(ctx->size_P > 1 && lt->beta == 1)) {  // alpha == 1 or several digits with a ragged last one: one ciphertext at a time
        for (size_t b = 0; b < batch; b++) {
            PHA_TRY(do_galois(ctx, in + b * ct_words, 2 * size_Ql, perm, out + b * ct_words, st));
            uint64_t *ws;
            PHA_TRY(ensure_workspace(ctx, (lt->beta * size_QlP + 2 * size_QlP + 3 * size_Ql) * n, &ws, st));
            WsLease ws_lease(ctx, st);
            uint64_t *c2 = ws + (lt->beta * size_QlP + 2 * size_QlP + 2 * size_Ql) * n;
            HIP_TRY(hipMemcpyAsync(c2, out + b * ct_words + size_Ql * n, size_Ql * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemsetAsync(out + b * ct_words + size_Ql * n, 0, size_Ql * n * sizeof(uint64_t), st));
            PHA_TRY(do_keyswitch(ctx, lt, out + b * ct_words, c2, galois_evks, st));
        }
        return PHA_OK;
    }
    // workspace: t_mod_up [batch][beta][QlP] | cx [batch][2][QlP] | t_cks [batch][2][Ql] (poly 1 slots used) | delta [2 batch][Ql]
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, batch * (lt->beta * size_QlP + 2 * size_QlP + 4 * size_Ql) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *t_mod_up = ws, *cx = t_mod_up + batch * lt->beta * size_QlP * n, *t_cks = cx + batch * 2 * size_QlP * n,
             *delta = t_cks + batch * ct_words;
    // permute both polys of every ciphertext (src/evaluate.cu:747-794)
    PHA_TRY(do_galois(ctx, in, batch * 2 * size_Ql, perm, out, st));
    // key switch the permuted c1 of every ciphertext
    PHA_TRY(do_modup_batched(ctx, lt, t_mod_up, out + size_Ql * n, t_cks + size_Ql * n, ct_words, batch, st));
    PHA_TRY(do_inner_prod_batched(ctx, lt, cx, t_mod_up, galois_evks, batch, st));
    return do_moddown(ctx, lt, out, cx, delta, out, 2 * batch, st, 2);  // first polys += pi(c0), second polys as they come
}

// HMult + relinearise + rescale as one composite (and optionally x 2, the doubling of a Chebyshev product):
// out[2][l-1][N] = ((P * (d0, d1) + KS_QP(d2)) / (P * q_{l-1})), (d0, d1, d2) = ct1 (x) ct2.  The key-switched part
// stays over Q_l u P, the tensor's first two polys are lifted onto it, and ONE moddown divides by the special primes
// and the level's last prime together.  Against multiply + relinearize + rescale_to_next it saves a full NTT pass
// over the ciphertext, the INTT of the last limb and its broadcast; the result differs by rounding only
// (both are within a few units of the exact quotient).
int pha_multiply_relin_rescale(pha_context *ctx, size_t chain_index, const uint64_t *ct1, const uint64_t *ct2,
                               const uint64_t *const *relin_evks, uint64_t *out, int times2, pha_stream_t stream) {
    REQUIRE_CTX(ct1 && ct2 && relin_evks && out);
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    if (size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    hipStream_t st = as_stream(stream);
    if (!ctx->d_bigP_mod_q.p) {
        LevelConstants lc;
        build_level_constants(ctx->moduli, ctx->size_P, ctx->size_Q, lc);
        HIP_TRY(ctx->d_bigP_mod_q.upload(lc.bigP_mod_q));
    }
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, (lt->beta * size_QlP + 2 * size_QlP + 5 * size_Ql) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *t_mod_up = ws, *cx = t_mod_up + lt->beta * size_QlP * n, *scratch = cx + 2 * size_QlP * n, *d = scratch + 2 * size_Ql * n;
    {   // tensor product, doubled here when asked: before the division the doubling is exact
        const dim3 tg(static_cast<uint32_t>(size_Ql) * (static_cast<uint32_t>(n) / 512));
        if (ct1 == ct2)
            hipLaunchKernelGGL((tensor_prod_2x2_kernel<true>), tg, dim3(256), 0, st, ct1, ct1, ctx->d_mod.p, d, static_cast<uint32_t>(n),
                               static_cast<uint32_t>(size_Ql), times2 ? 1u : 0u);
        else
            hipLaunchKernelGGL((tensor_prod_2x2_kernel<false>), tg, dim3(256), 0, st, ct1, ct2, ctx->d_mod.p, d, static_cast<uint32_t>(n),
                               static_cast<uint32_t>(size_Ql), times2 ? 1u : 0u);
        PHA_TRY(check_launch("tensor_prod_2x2"));
    }
    const bool in_place_own = ctx->size_P > 1;   // as in do_keyswitch
    PHA_TRY(do_modup(ctx, lt, t_mod_up, d + 2 * size_Ql * n, scratch, st, !in_place_own));
    PHA_TRY(do_inner_prod(ctx, lt, cx, t_mod_up, relin_evks, st, nullptr, in_place_own ? d + 2 * size_Ql * n : nullptr));
    const uint32_t n32 = static_cast<uint32_t>(n), ql = static_cast<uint32_t>(size_Ql), qlp = static_cast<uint32_t>(size_QlP);
    hipLaunchKernelGGL((lift_qp_kernel<false>), dim3(qlp * (n32 / 512), 2), dim3(256), 0, st, cx, d, static_cast<const uint32_t *>(nullptr),
                       ctx->d_bigP_mod_q.p, ctx->d_mod.p, n32, ql, qlp);
    PHA_TRY(check_launch("lift_qp"));
    return do_merged_moddown(ctx, lt, out, cx, scratch, 2, false, st);
}

// `batch` independent products at one level in one launch sequence: outs [batch][2][l-1][N] contiguous,
// out[b] = pha_multiply_relin_rescale(ct1s[b], ct2s[b]) bit for bit.  The transforms, the base conversions and the
// inner product span the batch (the relinearisation key is read once per four products); levels whose digits are
// neither uniform nor a single short one fall back to one product at a time.
int pha_multiply_relin_rescale_batched(pha_context *ctx, size_t chain_index, size_t batch, const uint64_t *const *ct1s,
                                       const uint64_t *const *ct2s, const uint64_t *const *relin_evks, uint64_t *outs, int times2,
                                       pha_stream_t stream) {
    REQUIRE_CTX(ct1s && ct2s && relin_evks && outs);
    if (batch == 0) return PHA_OK;
    LevelTool *lt;
    PHA_TRY(get_level(ctx, chain_index, &lt));
    const size_t n = ctx->n, size_Ql = lt->size_Ql, size_QlP = size_Ql + ctx->size_P;
    if (size_Ql < 2) return fail(PHA_ERR_INVALID_ARGUMENT, "end of modulus switching chain reached");
    const size_t out_words = 2 * (size_Ql - 1) * n;
    if (batch == 1 || (!lt->uniform_digits && !(ctx->size_P > 1 && lt->beta == 1))) {
        for (size_t b = 0; b < batch; b++)
            PHA_TRY(pha_multiply_relin_rescale(ctx, chain_index, ct1s[b], ct2s[b], relin_evks, outs + b * out_words, times2, stream));
        return PHA_OK;
    }
    hipStream_t st = as_stream(stream);
    if (!ctx->d_bigP_mod_q.p) {
        LevelConstants lc;
        build_level_constants(ctx->moduli, ctx->size_P, ctx->size_Q, lc);
        HIP_TRY(ctx->d_bigP_mod_q.upload(lc.bigP_mod_q));
    }
    // workspace: t_mod_up [batch][beta][QlP] | cx [batch][2][QlP] | d [batch][3][Ql] | t_cks [batch][3][Ql] (third slots used) |
    // delta [2 batch][Ql]
    const size_t d_words = 3 * size_Ql * n;
    uint64_t *ws;
    PHA_TRY(ensure_workspace(ctx, batch * (lt->beta * size_QlP + 2 * size_QlP + 2 * 3 * size_Ql + 2 * size_Ql) * n, &ws, st));
    WsLease ws_lease(ctx, st);
    uint64_t *t_mod_up = ws, *cx = t_mod_up + batch * lt->beta * size_QlP * n, *d = cx + batch * 2 * size_QlP * n,
             *t_cks = d + batch * d_words, *delta = t_cks + batch * d_words;
    const uint32_t n32 = static_cast<uint32_t>(n), ql = static_cast<uint32_t>(size_Ql), qlp = static_cast<uint32_t>(size_QlP);
    const dim3 tg(ql * (n32 / 512));
    for (size_t b = 0; b < batch; b++) {
        if (ct1s[b] == ct2s[b])
            hipLaunchKernelGGL((tensor_prod_2x2_kernel<true>), tg, dim3(256), 0, st, ct1s[b], ct1s[b], ctx->d_mod.p, d + b * d_words, n32, ql,
                               times2 ? 1u : 0u);
        else
            hipLaunchKernelGGL((tensor_prod_2x2_kernel<false>), tg, dim3(256), 0, st, ct1s[b], ct2s[b], ctx->d_mod.p, d + b * d_words, n32, ql,
                               times2 ? 1u : 0u);
    }
    PHA_TRY(check_launch("tensor_prod_2x2"));
    PHA_TRY(do_modup_batched(ctx, lt, t_mod_up, d + 2 * size_Ql * n, t_cks + 2 * size_Ql * n, d_words, batch, st));
    PHA_TRY(do_inner_prod_batched(ctx, lt, cx, t_mod_up, relin_evks, batch, st));
    for (size_t b = 0; b < batch; b++)
        hipLaunchKernelGGL((lift_qp_kernel<false>), dim3(qlp * (n32 / 512), 2), dim3(256), 0, st, cx + b * 2 * size_QlP * n, d + b * d_words,
                           static_cast<const uint32_t *>(nullptr), ctx->d_bigP_mod_q.p, ctx->d_mod.p, n32, ql, qlp);
    PHA_TRY(check_launch("lift_qp"));
    return do_merged_moddown(ctx, lt, outs, cx, delta, 2 * batch, false, st);
}

// ---- bootstrapping primitives (src/boot/Bootstrapper.cu) ------------------------------------------
int pha_mod_raise(pha_context *ctx, const uint64_t *src, uint64_t *dst, pha_stream_t stream) {
    REQUIRE_CTX(src && dst);
    hipStream_t st = as_stream(stream);
    const size_t n = ctx->n, limbs = ctx->size_Q;
    uint64_t *coeff;
    PHA_TRY(ensure_workspace(ctx, 2 * n, &coeff, st));
    WsLease ws_lease(ctx, st);
    // limb 0 of both polys to coefficient form (the lowest level has one limb: polys are n apart)
    NttRange r0;
    PHA_TRY(make_range(ctx, 1, 0, 0, 0, 0, 0, r0));
    PHA_TRY(ntt_backward(ctx, coeff, src, 2, n, r0, EPI_NONE, Epilogue{}, st));
    hipLaunchKernelGGL(mod_raise_kernel, dim3(static_cast<uint32_t>(n / 256), 2u), dim3(256), 0, st, dst, coeff, ctx->d_mod.p,
                       static_cast<uint32_t>(limbs), static_cast<uint32_t>(n));
    PHA_TRY(check_launch("mod_raise"));
    NttRange r;
    PHA_TRY(make_range(ctx, limbs, 0, 0, 0, 0, 0, r));
    return ntt_forward(ctx, dst, 2, limbs * n, r, EPI_NONE, Epilogue{}, st);
}

// gen_secretkey with secret_key_hamming_weight (src/secretkey.cu:401-433): exactly `hamming_weight`
// nonzero (+-1) coefficients at positions drawn without replacement from the seeded generator
int pha_gen_secretkey_sparse(pha_context *ctx, uint64_t seed, size_t hamming_weight, uint64_t *sk, pha_stream_t stream) {
    REQUIRE_CTX(sk);
    const size_t n = ctx->n, L = ctx->size_QP;
    if (hamming_weight == 0 || hamming_weight > n) return fail(PHA_ERR_INVALID_ARGUMENT, "hamming weight out of range");
    std::vector<uint32_t> pos(n);
    for (size_t i = 0; i < n; i++) pos[i] = static_cast<uint32_t>(i);
    std::vector<uint64_t> h(L * n, 0);
    for (size_t i = 0; i < hamming_weight; i++) {  // partial Fisher-Yates
        uint64_t r_pos, r_sign;
        PHA_TRY(host_rnd_at(seed, 2 * i, r_pos));
        PHA_TRY(host_rnd_at(seed, 2 * i + 1, r_sign));
        const size_t j = i + static_cast<size_t>(r_pos % (n - i));
        std::swap(pos[i], pos[j]);
        const bool minus = r_sign & 1;
        for (size_t l = 0; l < L; l++) h[l * n + pos[i]] = minus ? ctx->moduli[l] - 1 : 1;
    }
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(sk, h.data(), h.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // h goes out of scope
    NttRange r;
    PHA_TRY(make_range(ctx, L, 0, 0, 0, 0, 0, r));
    return ntt_forward(ctx, sk, 1, 0, r, EPI_NONE, Epilogue{}, st);
}

#ifdef PHA_TRACE
// development aid: see PHA_TRACE_MARK in ntt_kernels.cuh
int pha_set_trace(uint64_t *buf) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(buf);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pha::pha_trace_buf), &p, sizeof(p)));
    return PHA_OK;
}
#endif

}  // extern "C"
