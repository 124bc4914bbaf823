// When do the workgroups of a launch start?  1152 workgroups of 256 threads, varying static LDS and a VGPR floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_KIB, int VREGS, int SPIN_US>
__global__ __launch_bounds__(256) void probe(unsigned long long *t, float *sink) {
    __shared__ unsigned long long lds[LDS_KIB * 128 + 1];
    const unsigned long long t0 = wall_clock64();
    float acc[VREGS];
#pragma unroll
    for (int i = 0; i < VREGS; i++) acc[i] = threadIdx.x * 0.5f + i;
    if (threadIdx.x == 0) lds[0] = t0;
    // stay resident for SPIN_US
    while (wall_clock64() - t0 < SPIN_US * 100ull) {
#pragma unroll
        for (int i = 0; i < VREGS; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < VREGS; i++) s += acc[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x % (LDS_KIB * 128)];
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = wall_clock64(); }
}

template <int LDS_KIB, int VREGS, int SPIN_US>
int run(const char *name, int blocks) {
    unsigned long long *t; float *sink;
    CK(hipMalloc(&t, blocks * 16)); CK(hipMalloc(&sink, 64));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((probe<LDS_KIB, VREGS, SPIN_US>), dim3(blocks), dim3(256), 0, 0, t, sink);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 2);
    CK(hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull; for (int i = 0; i < blocks; i++) t0 = std::min(t0, h[2 * i]);
    std::vector<double> st(blocks); double end = 0;
    for (int i = 0; i < blocks; i++) { st[i] = (h[2 * i] - t0) * 0.01; end = std::max(end, (h[2 * i + 1] - t0) * 0.01); }
    std::sort(st.begin(), st.end());
    int within1 = std::lower_bound(st.begin(), st.end(), 1.0) - st.begin();
    int within5 = std::lower_bound(st.begin(), st.end(), (double)SPIN_US - 0.1) - st.begin();
    printf("%-28s blocks %d: started <1us %4d, before the first could end %4d | start p50 %.2f p75 %.2f p90 %.2f max %.2f | end %.2f us\n", name, blocks,
           within1, within5, st[blocks / 2], st[blocks * 3 / 4], st[blocks * 9 / 10], st[blocks - 1], end);
    CK(hipFree(t)); CK(hipFree(sink));
    return 0;
}

int main() {
    run<1, 8, 6>("lds 1K vgpr~16", 1152);
    run<36, 8, 6>("lds 36K vgpr~16", 1152);
    run<1, 100, 6>("lds 1K vgpr~110", 1152);
    run<36, 100, 6>("lds 36K vgpr~110", 1152);
    run<36, 100, 6>("lds 36K vgpr~110", 1024);
    run<32, 100, 6>("lds 32K vgpr~110", 1152);
    run<30, 100, 6>("lds 30K vgpr~110", 1280);
    run<18, 50, 6>("lds 18K vgpr~60", 2304);
    return 0;
}
