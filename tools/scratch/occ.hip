// occupancy probe: how many 256-thread workgroups with a given LDS size are resident per CU on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ unsigned long long dyn[];
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void probe(unsigned long long *out, int spin_us) {
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { dyn[0] = t0; out[blockIdx.x] = t0; }
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) { }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d  sharedMemPerMultiprocessor %zu  sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.multiProcessorCount, p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    const int n = 2048;
    unsigned long long *d; hipMalloc(&d, n * 8);
    for (int lds : {16384, 28672, 30720, 31744, 32256, 32768, 36864, 40960, 65536}) {
        int nb = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 256, lds);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipMemset(d, 0, n * 8);
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), lds, 0, d, 50);
        hipError_t e = hipDeviceSynchronize();
        std::vector<unsigned long long> h(n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0; for (auto v : h) if (v - t0 < 2000) first++;   // started within 20 us
        printf("lds %6d  api blocks/CU %d  resident at once %d (%.2f per CU)  err %d\n", lds, nb, first, first / 256.0, (int)e);
    }
    return 0;
}
