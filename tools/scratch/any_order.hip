// Does hipExtAnyOrderLaunch let two consecutive kernels of ONE stream overlap on gfx950?  Each kernel: 256 workgroups (a quarter of
// the slots) that spin for ~20 us.  Serial: ~40 us per pair; overlapped: ~20 us.
// build: hipcc --offload-arch=gfx950 -O2 -o any_order any_order.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 1) sink[0] = 1;
}
int main() {
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned long long ticks = 2000;     // wall_clock64 runs at 100 MHz: 20 us
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 10; i++) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, sink);
                if (mode == 0)
                    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, ticks, sink);
                else
                    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %.1f us per pair of 20-us kernels\n", mode ? "second kernel any-order" : "both in order        ", 1e3 * ms / 10);
        }
    }
    return 0;
}
