// L2 write policy probe for the single-launch transform (MI355X): does a line that one workgroup stores twice (plain, then
// plain or sc1) in ONE kernel leave the XCD's L2 once or twice, and is a plain-stored tile read back by the same XCD with sc1
// loads served from L2?  Read with rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE per kernel name.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_l2.hip -o tools/ubench_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_plain(u64x2_t *p, u64x2_t v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1(u64x2_t *p, u64x2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64x2_t ld_sc1(const u64x2_t *p) {
    u64x2_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
constexpr int kTileVec = 2048;   // 32 KiB per workgroup
template <int WRITES, bool LAST_SC1, bool READBACK, int GAP>
__global__ __launch_bounds__(256) void probe(u64x2_t *buf, u64x2_t *sink) {
    u64x2_t *tile = buf + static_cast<size_t>(blockIdx.x) * kTileVec;
    u64x2_t acc = {0, 0};
    for (int w = 0; w < WRITES; w++) {
        for (int i = 0; i < 8; i++) {
            const u64x2_t v = {static_cast<unsigned long long>(threadIdx.x + w), static_cast<unsigned long long>(i)};
            if (LAST_SC1 && w == WRITES - 1) st_sc1(tile + i * 256 + threadIdx.x, v); else st_plain(tile + i * 256 + threadIdx.x, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int g = 0; g < GAP; g++) __builtin_amdgcn_s_sleep(100);
    }
    if (READBACK) {
        for (int i = 0; i < 8; i++) acc += ld_sc1(tile + ((i * 256 + threadIdx.x) ^ 1));
        if (acc.x == 0x1234567) sink[threadIdx.x] = acc;
    }
}
int main() {
    const int wgs = 2048;
    u64x2_t *buf, *sink;
    hipMalloc(&buf, sizeof(u64x2_t) * kTileVec * wgs);
    hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((probe<1, false, false, 0>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<2, false, false, 0>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<3, false, false, 0>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<2, true, false, 0>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<2, false, false, 20>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<1, false, true, 0>), dim3(wgs), dim3(256), 0, 0, buf, sink);
        hipLaunchKernelGGL((probe<1, false, true, 20>), dim3(wgs), dim3(256), 0, 0, buf, sink);
    }
    hipDeviceSynchronize();
    printf("done: %d workgroups x 32 KiB = %.1f MB per write pass\n", wgs, wgs * 32768 / 1e6);
    return 0;
}
