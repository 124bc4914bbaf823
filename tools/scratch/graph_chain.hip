// hipGraph capture of a chain of forks with HIP alone: m forks s0, s0 forks s1, s1 joins s0, s0 joins m.
// build: hipcc --offload-arch=gfx950 -O2 -o graph_chain graph_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void addk(long* y, const long* x, long k, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i] + k; }
__global__ void add2(long* y, const long* a, const long* b, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = a[i] + b[i]; }
int main(int argc, char** argv) {
    const int n = 1 << 20, mode = argc > 1 ? atoi(argv[1]) : 0;      // mode bits: 1 global capture mode, 2 destroy each event right after its wait (torch's wait_stream does), 4 AutoFreeOnLaunch, 8 hipMalloc while capturing
    long *x, *a, *b, *c, *d, *e, *out;
    for (long** p : {&x, &a, &b, &c, &d, &e, &out}) CK(hipMalloc(p, n * sizeof(long)));
    CK(hipMemset(x, 0, n * sizeof(long)));
    hipStream_t m, s0, s1;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t f0, f1, j1, j0;
    for (hipEvent_t* ev : {&f0, &f1, &j1, &j0}) CK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    dim3 g(n / 256), t(256);
    CK(hipStreamBeginCapture(m, (mode & 1) ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(f0, m)); CK(hipStreamWaitEvent(s0, f0, 0)); if (mode & 2) CK(hipEventDestroy(f0));
    CK(hipEventRecord(f1, s0)); CK(hipStreamWaitEvent(s1, f1, 0)); if (mode & 2) CK(hipEventDestroy(f1));
    if (mode & 8) { hipStreamCaptureMode cm = hipStreamCaptureModeRelaxed; CK(hipThreadExchangeStreamCaptureMode(&cm)); long* extra; CK(hipMalloc(&extra, n * sizeof(long))); CK(hipThreadExchangeStreamCaptureMode(&cm)); a = extra; }
    addk<<<g, t, 0, s1>>>(a, x, 4, n);
    addk<<<g, t, 0, s0>>>(b, x, 3, n);
    CK(hipEventRecord(j1, s1)); CK(hipStreamWaitEvent(s0, j1, 0)); if (mode & 2) CK(hipEventDestroy(j1));
    add2<<<g, t, 0, s0>>>(c, a, b, n);
    addk<<<g, t, 0, m>>>(d, x, 1, n);
    CK(hipEventRecord(j0, s0)); CK(hipStreamWaitEvent(m, j0, 0)); if (mode & 2) CK(hipEventDestroy(j0));
    add2<<<g, t, 0, m>>>(out, c, d, n);
    hipGraph_t graph;
    CK(hipStreamEndCapture(m, &graph));
    printf("capture ok\n");
    hipGraphExec_t exec;
    if (mode & 4) CK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
    else CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, m)); CK(hipStreamSynchronize(m));
    long h; CK(hipMemcpy(&h, out + 5, sizeof(long), hipMemcpyDeviceToHost));
    printf("replay ok: %ld (want 8)\n", h);
    return h == 8 ? 0 : 1;
}
