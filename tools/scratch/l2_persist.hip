// Does an XCD's L2 keep a kernel's lines for the NEXT kernel on the same stream?  Kernel W writes chunk b from block b,
// kernel R reads chunk (b + shift) from block b: shift 0 = same XCD under round-robin placement, shift 1 = another one.
// Also: read-only / write-only stream rates over a MALL-resident buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kChunk = 4096;  // 16-B words per block = 64 KiB

__global__ __launch_bounds__(256) void wr(ulonglong2 *buf, unsigned long long v) {
    ulonglong2 *p = buf + (size_t)blockIdx.x * kChunk;
#pragma unroll
    for (int i = 0; i < kChunk / 256; i++) p[i * 256 + threadIdx.x] = ulonglong2{v + i, v ^ threadIdx.x};
}
__global__ __launch_bounds__(256) void rd(const ulonglong2 *buf, unsigned long long *sink, int shift) {
    const unsigned b = (blockIdx.x + shift) % gridDim.x;
    const ulonglong2 *p = buf + (size_t)b * kChunk;
    unsigned long long acc = 0;
#pragma unroll
    for (int i = 0; i < kChunk / 256; i++) { ulonglong2 x = p[i * 256 + threadIdx.x]; acc += x.x ^ x.y; }
    if (acc == 0x123456789abcdefull) sink[0] = acc;
}
__global__ __launch_bounds__(256) void rw(ulonglong2 *buf, int shift) {
    const unsigned b = (blockIdx.x + shift) % gridDim.x;
    ulonglong2 *p = buf + (size_t)b * kChunk;
#pragma unroll
    for (int i = 0; i < kChunk / 256; i++) { ulonglong2 x = p[i * 256 + threadIdx.x]; x.x += 1; x.y ^= x.x; p[i * 256 + threadIdx.x] = x; }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long *sink; CK(hipMalloc(&sink, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t mib : {4, 8, 16, 24, 36, 72}) {
        const size_t bytes = mib << 20; const unsigned blocks = bytes / (kChunk * 16);
        ulonglong2 *buf; CK(hipMalloc(&buf, bytes));
        for (int mode = 0; mode < 4; mode++) {          // 0: W then R shift 0   1: W then R shift 1   2: W then RW shift 0  3: W then RW shift 1
            std::vector<float> ts;
            for (int rep = 0; rep < 30; rep++) {
                hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, st, buf, (unsigned long long)rep);
                CK(hipEventRecord(a, st));
                if (mode < 2) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, st, buf, sink, mode & 1);
                else hipLaunchKernelGGL(rw, dim3(blocks), dim3(256), 0, st, buf, mode & 1);
                CK(hipEventRecord(b, st));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            const double us = ts[ts.size() / 2] * 1e3;
            printf("%3zu MiB  %s shift %d: %8.2f us  %6.2f TB/s (%s)\n", mib, mode < 2 ? "W->R " : "W->RW", mode & 1, us,
                   (mode < 2 ? 1.0 : 2.0) * bytes / us / 1e6, mode < 2 ? "read" : "read+write");
        }
        // W,R pairs back to back (no events inside): does R find W's lines in its XCD's L2?
        for (int shift = 0; shift < 2; shift++) {
            for (int k = 0; k < 2; k++) {
                CK(hipEventRecord(a, st));
                for (int rep = 0; rep < 50; rep++) {
                    hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, st, buf, (unsigned long long)rep);
                    if (k == 0) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, st, buf, sink, shift);
                    else hipLaunchKernelGGL(rw, dim3(blocks), dim3(256), 0, st, buf, shift);
                }
                CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                printf("%3zu MiB  pairs W,%s shift %d x50: %8.2f us/pair\n", mib, k ? "RW" : "R ", shift, ms * 1e3 / 50);
            }
        }
        // back-to-back read-only and write-only streams
        for (int k = 0; k < 2; k++) {
            CK(hipEventRecord(a, st));
            for (int rep = 0; rep < 50; rep++) {
                if (k == 0) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, st, buf, sink, 0);
                else hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, st, buf, (unsigned long long)rep);
            }
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("%3zu MiB  %s x50 back-to-back: %8.2f us/launch  %6.2f TB/s\n", mib, k ? "write-only" : "read-only ", ms * 1e3 / 50, bytes / (ms * 1e3 / 50) / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
