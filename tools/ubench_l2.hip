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
__device__ __forceinline__ u64x2_t ld_plain(const u64x2_t *p) {
    u64x2_t v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
// the single-launch transform's hand-off: the tile is LOADED first (plain), then stored (plain), then read back by the same
// XCD (MODE 0: sc1 loads, 1: plain loads, 2: sc1 loads by the workgroup blockIdx ^ 8, i.e. another CU of the same XCD under
// round-robin dispatch, after a grid-wide counter says every tile is stored)
template <int MODE, int GAP>
__global__ __launch_bounds__(256) void handoff(u64x2_t *buf, u64x2_t *sink, unsigned *counter) {
    u64x2_t *tile = buf + static_cast<size_t>(blockIdx.x) * kTileVec;
    u64x2_t acc = {0, 0};
    for (int i = 0; i < 8; i++) acc += ld_plain(tile + i * 256 + threadIdx.x);
    for (int i = 0; i < 8; i++) {
        const u64x2_t v = {acc.x + threadIdx.x, static_cast<unsigned long long>(i)};
        st_plain(tile + i * 256 + threadIdx.x, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int g = 0; g < GAP; g++) __builtin_amdgcn_s_sleep(100);
    const u64x2_t *src = tile;
    if (MODE == 2) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 20)) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        src = buf + static_cast<size_t>(blockIdx.x ^ 8u) * kTileVec;
    }
    for (int i = 0; i < 8; i++) acc += MODE == 1 ? ld_plain(src + ((i * 256 + threadIdx.x) ^ 1)) : ld_sc1(src + ((i * 256 + threadIdx.x) ^ 1));
    if (acc.x == 0x1234567) sink[threadIdx.x] = acc;
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
    unsigned *counter;
    hipMalloc(&counter, 4);
    for (int rep = 0; rep < 3; rep++) {
        // 512 workgroups (two per CU, all resident): 16.8 MB, half of the aggregate L2
        hipLaunchKernelGGL((handoff<0, 0>), dim3(512), dim3(256), 0, 0, buf, sink, counter);
        hipLaunchKernelGGL((handoff<1, 0>), dim3(512), dim3(256), 0, 0, buf, sink, counter);
        hipLaunchKernelGGL((handoff<0, 4>), dim3(512), dim3(256), 0, 0, buf, sink, counter);
        hipMemsetAsync(counter, 0, 4, 0);
        hipLaunchKernelGGL((handoff<2, 0>), dim3(512), dim3(256), 0, 0, buf, sink, counter);
        // 2048 workgroups: 67 MB, twice the aggregate L2, in four resident rounds
        hipLaunchKernelGGL((handoff<0, 0>), dim3(2048), dim3(256), 0, 0, buf, sink, counter);
    }
    hipDeviceSynchronize();
    printf("done: %d workgroups x 32 KiB = %.1f MB per write pass\n", wgs, wgs * 32768 / 1e6);
    return 0;
}
