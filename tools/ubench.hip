// Micro-benchmarks that decide the NTT design on gfx950: integer multiply issue rates, the cost
// of one Shoup butterfly, FP64 FMA rate, and in-place streaming bandwidth at cache-resident and
// HBM-resident footprints; `ubench exchange`: lane-exchange primitives against the LDS exchange.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP>
__global__ __launch_bounds__(256) void alu_kernel(uint64_t *out, uint64_t seed) {
    uint64_t a[ILP];
    uint32_t b = static_cast<uint32_t>(seed) | 1u;
    uint64_t w = seed * 0x9E3779B97F4A7C15ull | 1ull;
    double fa[ILP];
    const double fq = 1125899906826241.0 + static_cast<double>(seed & 1), fqinv = 1.0 / fq, fw = 0.37 * fq + static_cast<double>(seed & 3),
                 fwq = fw * fqinv;
#pragma unroll
    for (int i = 0; i < ILP; i++) { a[i] = seed + threadIdx.x * 7 + i * 1315423911ull + blockIdx.x; fa[i] = 1.0 + i + threadIdx.x; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) a[i] = static_cast<uint64_t>(static_cast<uint32_t>(a[i])) * b + a[i];                  // v_mad_u64_u32
            if (OP == 1) a[i] = static_cast<uint32_t>(a[i]) * b;                                                 // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(static_cast<uint32_t>(a[i]), b) + 1;                                    // v_mul_hi_u32
            if (OP == 3) a[i] = a[i] * w + 1;                                                                    // 64-bit mul lo
            if (OP == 4) a[i] = __umul64hi(a[i], w) + a[i];                                                      // mulhi64
            if (OP == 5) { uint64_t h = __umul64hi(a[i], w); a[i] = a[i] * (w >> 3) - h * 0x0FFFFFFFFFFC0001ull; } // shoup lazy
            if (OP == 6) fa[i] = __fma_rn(fa[i], 1.0000001, 0.5);                                                // v_fma_f64
            if (OP == 7) a[i] = a[i] + w;                                                                        // 64-bit add
            if (OP == 8) a[i] = static_cast<uint32_t>(a[i]) + b;                                                 // 32-bit add
            if (OP == 9) a[i] = __mul24(static_cast<int>(a[i]), static_cast<int>(b));                             // v_mul_i32_i24
            if (OP == 10) fa[i] = fa[i] * 1.0000001;                                                             // v_mul_f64
            if (OP == 11) fa[i] = fa[i] + 0.4999;                                                                // v_add_f64
            if (OP == 12) fa[i] = __builtin_rint(fa[i]) + 0.4999;                                                // v_rndne_f64 + v_add_f64
            if (OP == 13) {  // the FpField modular product as shipped: mul, fma, mul, rndne, fma, add
                const double h = fa[i] * fw, l = __builtin_fma(fa[i], fw, -h), c = __builtin_rint(fa[i] * fwq);
                fa[i] = __builtin_fma(-c, fq, h) + l;
            }
            if (OP == 14) {  // quotient rounded by adding and subtracting 1.5 * 2^52: mul, fma, fma, add, fma, add
                const double h = fa[i] * fw, l = __builtin_fma(fa[i], fw, -h);
                const double c = __builtin_fma(fa[i], fwq, 6755399441055744.0) - 6755399441055744.0;
                fa[i] = __builtin_fma(-c, fq, h) + l;
            }
            if (OP == 15) {  // quotient from h (no w/q companion): mul, fma, mul, rndne, fma, add
                const double h = fa[i] * fw, l = __builtin_fma(fa[i], fw, -h), c = __builtin_rint(h * fqinv);
                fa[i] = __builtin_fma(-c, fq, h) + l;
            }
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) r += a[i] + static_cast<uint64_t>(fa[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void stream_rw_kernel(ulonglong2 *buf, size_t n16) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) {
        ulonglong2 v = buf[i];
        v.x += 1;
        v.y ^= 3;
        buf[i] = v;
    }
}
__global__ __launch_bounds__(256) void stream_rw8_kernel(uint64_t *buf, size_t n8) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n8; i += gridDim.x * 256ull) buf[i] += 1;
}

// read R 16-B words and write W 16-B words per step: the read : write mixes of the streaming kernels (mode "mix")
template <int R, int W>
__global__ __launch_bounds__(256) void stream_mix_kernel(const ulonglong2 *src, ulonglong2 *dst, size_t steps) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < steps; i += gridDim.x * 256ull) {
        ulonglong2 acc = {i, 3};
#pragma unroll
        for (int r = 0; r < R; r++) {
            const ulonglong2 v = src[r * steps + i];
            acc.x += v.x;
            acc.y ^= v.y;
        }
#pragma unroll
        for (int w = 0; w < W; w++) {
            acc.x += w;
            dst[w * steps + i] = acc;
        }
    }
}
template <int R, int W>
int run_mix(const ulonglong2 *src, ulonglong2 *dst, size_t steps, int grid) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 50;
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((stream_mix_kernel<R, W>), dim3(grid), dim3(256), 0, 0, src, dst, steps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((stream_mix_kernel<R, W>), dim3(grid), dim3(256), 0, 0, src, dst, steps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 16.0 * steps * (R + W);
    printf("read %d : write %d, %6.1f MiB read + %6.1f MiB written, grid %5d: %7.2f us/launch  %6.2f TB/s\n", R, W,
           16.0 * steps * R / 1048576.0, 16.0 * steps * W / 1048576.0, grid, 1e3 * ms / reps, bytes * reps / (ms * 1e-3) / 1e12);
    return 0;
}
int mix_mode() {
    // 18 MiB units: read 1 : write 3 is the modup's conversion at configs[2] (18 MiB in, 54 MiB out), 1 : 3 at 12 MiB its
    // moddown (12 in, 36 out); 4 : 3 the tensor product; 1 : 0 and 0 : 1 the pure directions
    void *src, *dst;
    CHECK(hipMalloc(&src, 128ull << 20)); CHECK(hipMalloc(&dst, 128ull << 20));
    CHECK(hipMemset(src, 1, 128ull << 20)); CHECK(hipMemset(dst, 1, 128ull << 20));
    const ulonglong2 *s = (const ulonglong2 *)src; ulonglong2 *d = (ulonglong2 *)dst;
    for (int grid : {1024, 2048, 4096}) {
        const size_t u18 = (18ull << 20) / 16, u12 = (12ull << 20) / 16;
        run_mix<1, 0>(s, d, 4 * u18, grid);
        run_mix<0, 1>(s, d, 4 * u18, grid);
        run_mix<0, 3>(s, d, u18, grid);
        run_mix<1, 3>(s, d, u18, grid);
        run_mix<1, 3>(s, d, u12, grid);
        run_mix<4, 3>(s, d, u18, grid);
        run_mix<1, 1>(s, d, 2 * u18, grid);
    }
    return 0;
}

template <int OP>
int run_alu(const char *name, uint64_t *d_out, double ops_per_iter) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(alu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(alu_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = 5.0 * blocks * 256.0 * ITERS * ILP * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.2f cycles per wave-instr per SIMD @2.4GHz)\n", name, ms,
           lane_ops / (ms * 1e-3) / 1e12, 1024.0 * 2.4e9 * 64 / (lane_ops / (ms * 1e-3)));
    return 0;
}

// ---- exchange of the 16 values a lane holds with the other lanes (mode "exchange") ----------------------------------
// What a radix-16 register pass needs between two passes: a transpose between the register index and four bits of
// the thread index.  Every variant moves all 16 64-bit values of every lane once per iteration; a little FP64 work
// between iterations keeps the values live.  X_LDS / X_LDS_BARRIER: ds_write_b64 x16, (workgroup barrier), ds_read_b64
// x16 -- the shipped exchange (the barrier variant is what a cross-wave transpose costs).  X_SWAP32 / X_SWAP16: one
// butterfly level at lane distance 32 / 16 with v_permlane32_swap / v_permlane16_swap (16 instructions).  X_DPP8: one
// level at distance 8 with v_mov_dpp row_ror:8 and selects.  X_DPP_ROW: the full 16 x 16 transpose inside a 16-lane row,
// four levels (8, 4, 2, 1; distances below 8 have no xor form on gfx950: two shifted moves and a select each).
// X_BPERMUTE: the same transpose through ds_bpermute_b32 (the LDS crossbar without memory), 32 of them.
enum { X_LDS, X_LDS_BARRIER, X_SWAP32, X_SWAP16, X_DPP8, X_DPP_ROW, X_BPERMUTE, X_NONE };
typedef unsigned xv2u __attribute__((ext_vector_type(2)));

template <int DIST>
__device__ __forceinline__ unsigned dpp_xor(unsigned v, bool upper) {
    // the value of lane (l ^ DIST) within a 16-lane row
    if (DIST == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    const unsigned dn = __builtin_amdgcn_update_dpp(0u, v, 0x100 + DIST, 0xf, 0xf, false);   // row_shl: lane l reads l + DIST
    const unsigned up = __builtin_amdgcn_update_dpp(0u, v, 0x110 + DIST, 0xf, 0xf, false);   // row_shr: lane l reads l - DIST
    return upper ? up : dn;
}

template <int DIST>
__device__ __forceinline__ void dpp_level(double (&r)[16], unsigned lane) {
    const bool upper = (lane & DIST) != 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i & DIST) continue;
        // lanes with the bit clear keep r[i] and send r[i + DIST]; the others keep r[i + DIST] and send r[i]
        const uint64_t send = __double_as_longlong(upper ? r[i] : r[i + DIST]);
        const unsigned lo = dpp_xor<DIST>(static_cast<unsigned>(send), upper), hi = dpp_xor<DIST>(static_cast<unsigned>(send >> 32), upper);
        const double got = __longlong_as_double((static_cast<uint64_t>(hi) << 32) | lo);
        if (upper) r[i] = got; else r[i + DIST] = got;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void exchange_kernel(double *out, int iters, double seed) {
    __shared__ double lds[256 * 17];
    const unsigned tid = threadIdx.x, lane = tid & 63, row = tid >> 4, c = tid & 15;
    double r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed + tid * 16 + i;
    for (int it = 0; it < iters; it++) {
        if (MODE == X_LDS || MODE == X_LDS_BARRIER) {
            // 16 x 16 transpose inside each 16-lane row: lane c writes its register i to [row][c][i], reads [row][i][c]
#pragma unroll
            for (int i = 0; i < 16; i++) lds[(row * 16 + c) * 17 + i] = r[i];
            if (MODE == X_LDS_BARRIER) __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = lds[(row * 16 + i) * 17 + c];
            if (MODE == X_LDS_BARRIER) __syncthreads();
        } else if (MODE == X_SWAP32 || MODE == X_SWAP16) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint64_t a = __double_as_longlong(r[i]), b = __double_as_longlong(r[i + 8]);
                xv2u lo, hi;
                if (MODE == X_SWAP32) {
                    lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(a), static_cast<unsigned>(b), false, false);
                    hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(a >> 32), static_cast<unsigned>(b >> 32), false, false);
                } else {
                    lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(a), static_cast<unsigned>(b), false, false);
                    hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(a >> 32), static_cast<unsigned>(b >> 32), false, false);
                }
                r[i] = __longlong_as_double((static_cast<uint64_t>(hi.x) << 32) | lo.x);
                r[i + 8] = __longlong_as_double((static_cast<uint64_t>(hi.y) << 32) | lo.y);
            }
        } else if (MODE == X_DPP8) {
            dpp_level<8>(r, lane);
        } else if (MODE == X_DPP_ROW) {
            dpp_level<8>(r, lane);
            dpp_level<4>(r, lane);
            dpp_level<2>(r, lane);
            dpp_level<1>(r, lane);
        } else if (MODE == X_BPERMUTE) {
            // register i of lane c <- register c of lane i (same row): rotate the register index by the lane, pull, rotate back
            double t[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // step i: lane c pulls from lane (c ^ i) that lane's register (c ^ i) ^ i ... a select chain stands in for the
                // dynamic register index: the cost measured is the 32 ds_bpermute_b32, as in a real implementation
                const unsigned src = ((lane & 48) | (c ^ i)) << 2;
                const uint64_t v = __double_as_longlong(r[i]);
                const unsigned lo = __builtin_amdgcn_ds_bpermute(src, static_cast<unsigned>(v));
                const unsigned hi = __builtin_amdgcn_ds_bpermute(src, static_cast<unsigned>(v >> 32));
                t[i] = __longlong_as_double((static_cast<uint64_t>(hi) << 32) | lo);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) r[i] = t[i];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) r[i] = __builtin_fma(r[i], 1.0000001, 0.25);   // 16 FP64 FMAs: one butterfly column's worth
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += r[i];
    out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static double time_exchange(const char *name, double *d_out, double base_ns) {
    const int iters = 2000, grid = 256 * 8;   // 8 workgroups = 32 waves per CU: 8 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.5);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(exchange_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.5);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves share it, each doing `iters` exchanges
    const double ns_per_exchange = 1e6 * ms / iters / 8.0;
    printf("%-44s %8.1f ns per exchange and wave (SIMD time)   %8.1f ns over the FMA-only loop\n", name, ns_per_exchange,
           ns_per_exchange - base_ns);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ns_per_exchange;
}

static int exchange_mode() {
    double *d_out;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(double)));
    printf("16 x 64-bit values per lane, 8 waves per SIMD, 16 FP64 FMAs between exchanges\n");
    const double base = time_exchange<X_NONE>("no exchange (16 FMAs only)", d_out, 0.0);
    time_exchange<X_LDS>("LDS, intra-row transpose, no barrier", d_out, base);
    time_exchange<X_LDS_BARRIER>("LDS, two workgroup barriers (shipped form)", d_out, base);
    time_exchange<X_SWAP32>("v_permlane32_swap, one level (distance 32)", d_out, base);
    time_exchange<X_SWAP16>("v_permlane16_swap, one level (distance 16)", d_out, base);
    time_exchange<X_DPP8>("DPP row_ror:8 + selects, one level", d_out, base);
    time_exchange<X_DPP_ROW>("DPP, full 16x16 row transpose (4 levels)", d_out, base);
    time_exchange<X_BPERMUTE>("ds_bpermute_b32 x32, full row transpose", d_out, base);
    CHECK(hipDeviceSynchronize());
    return 0;
}


// ---- mode "policy": the in-place pass of a transform kernel (read 16 B, write 16 B to the same address, 36 MiB, resident in
// the Infinity Cache as in the benchmark loop) under the cache-policy bits a store / load can carry on gfx950 ----------
#define POLICY_KERNEL(NAME, LD, ST)                                                                              \
    __global__ __launch_bounds__(256) void NAME(ulonglong2 *buf, size_t n) {                                      \
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {                      \
            u64x2_p v;                                                                                            \
            asm volatile("global_load_dwordx4 %0, %1, off" LD "\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(buf + i) : "memory"); \
            v.x += 1;                                                                                             \
            asm volatile("global_store_dwordx4 %0, %1, off" ST "\n\ts_nop 1" : : "v"(buf + i), "v"(v) : "memory");  \
        }                                                                                                         \
    }
typedef unsigned long long u64x2_p __attribute__((ext_vector_type(2)));
POLICY_KERNEL(pol_plain, "", "")
POLICY_KERNEL(pol_st_sc1, "", " sc1")
POLICY_KERNEL(pol_st_sc0, "", " sc0")
POLICY_KERNEL(pol_st_sc0sc1, "", " sc0 sc1")
POLICY_KERNEL(pol_st_nt, "", " nt")
POLICY_KERNEL(pol_st_ntsc1, "", " sc1 nt")
POLICY_KERNEL(pol_st_ntsc0sc1, "", " sc0 sc1 nt")
POLICY_KERNEL(pol_ld_nt_st_sc1, " nt", " sc1")
POLICY_KERNEL(pol_ld_sc1_st_sc1, " sc1", " sc1")
POLICY_KERNEL(pol_ld_nt_st_ntsc1, " nt", " sc1 nt")
int policy_mode() {
    void *buf;
    const size_t bytes = 36ull << 20, n = bytes / 16;
    CHECK(hipMalloc(&buf, 2 * bytes));
    CHECK(hipMemset(buf, 1, 2 * bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    typedef void (*kern_t)(ulonglong2 *, size_t);
    const struct { const char *name; kern_t k; } ks[] = {
        {"load plain, store plain", pol_plain}, {"store sc1", pol_st_sc1}, {"store sc0", pol_st_sc0}, {"store sc0 sc1", pol_st_sc0sc1},
        {"store nt", pol_st_nt}, {"store sc1 nt", pol_st_ntsc1}, {"store sc0 sc1 nt", pol_st_ntsc0sc1},
        {"load nt, store sc1", pol_ld_nt_st_sc1}, {"load sc1, store sc1", pol_ld_sc1_st_sc1}, {"load nt, store sc1 nt", pol_ld_nt_st_ntsc1}};
    for (int round = 0; round < 2; round++)
        for (int grid : {1152, 4608})
            for (const auto &e : ks) {
                // two buffers alternating, like the passes of a forward + inverse step over 2 x 36 limbs
                for (int r = 0; r < 10; r++) hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)buf + (r & 1) * n, n);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                const int reps = 100;
                for (int r = 0; r < reps; r++) hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)buf + (r & 1) * n, n);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("round %d grid %4d  %-26s %6.2f us per 36 MiB in-place pass = %5.2f TB/s\n", round, grid, e.name, 1e3 * ms / reps,
                       2.0 * bytes / (ms / reps * 1e-3) / 1e12);
            }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "exchange") return exchange_mode();
    if (argc > 1 && std::string(argv[1]) == "mix") return mix_mode();
    if (argc > 1 && std::string(argv[1]) == "policy") return policy_mode();
    if (argc > 1) {
        // calibration mode for the FETCH_SIZE / WRITE_SIZE counters: known byte counts
        // (512 MiB read + 512 MiB written per launch), 3 launches of each access width
        const size_t bytes = 512ull << 20;
        void *buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 1, bytes));
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(stream_rw_kernel, dim3(8192), dim3(256), 0, 0, (ulonglong2 *)buf, bytes / 16);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(stream_rw8_kernel, dim3(8192), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
        CHECK(hipDeviceSynchronize());
        printf("calibration: 536870912 bytes read and written per launch\n");
        return 0;
    }
    uint64_t *d_out;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint64_t)));
    run_alu<0>("v_mad_u64_u32", d_out, 1);
    run_alu<1>("v_mul_lo_u32", d_out, 1);
    run_alu<2>("v_mul_hi_u32(+add)", d_out, 1);
    run_alu<3>("mul64 lo (+add)", d_out, 1);
    run_alu<4>("mulhi64 (+add)", d_out, 1);
    run_alu<5>("shoup lazy modmul", d_out, 1);
    run_alu<6>("v_fma_f64", d_out, 1);
    run_alu<7>("add64", d_out, 1);
    run_alu<8>("add32", d_out, 1);
    run_alu<9>("v_mul_i32_i24", d_out, 1);
    run_alu<10>("v_mul_f64", d_out, 1);
    run_alu<11>("v_add_f64", d_out, 1);
    run_alu<12>("v_rndne_f64 + v_add_f64", d_out, 2);
    run_alu<13>("fp64 modmul (rndne), 6 ops", d_out, 6);
    run_alu<14>("fp64 modmul (magic), 6 ops", d_out, 6);
    run_alu<15>("fp64 modmul (q from h), 6 ops", d_out, 6);
    if (argc == 1 && getenv("UBENCH_ALU_ONLY")) return 0;

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t mib : {36ull, 72ull, 144ull, 512ull, 2048ull}) {
        const size_t bytes = mib << 20;
        void *buf;
        CHECK(hipMalloc(&buf, bytes));
        CHECK(hipMemset(buf, 1, bytes));
        for (int variant = 0; variant < 2; variant++) {
            const int reps = mib <= 144 ? 50 : 10;
            for (int grid : {2048, 8192}) {
                auto launch = [&]() {
                    if (variant == 0) hipLaunchKernelGGL(stream_rw_kernel, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)buf, bytes / 16);
                    else hipLaunchKernelGGL(stream_rw8_kernel, dim3(grid), dim3(256), 0, 0, (uint64_t *)buf, bytes / 8);
                };
                launch(); launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                for (int r = 0; r < reps; r++) launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("in-place rw %4zu MiB %s grid %5d: %8.2f us/launch  %7.2f TB/s (read+write)\n", mib,
                       variant == 0 ? "16B/lane" : " 8B/lane", grid, 1e3 * ms / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e12);
            }
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
