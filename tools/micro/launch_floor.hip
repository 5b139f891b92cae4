// Micro-benchmark: cost of one dependent kernel node inside a replayed hipGraph, for an empty kernel at several
// launch shapes (grid, LDS).  hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDSB>
__global__ __launch_bounds__(256) void k_empty(float* p) {
    __shared__ float s[LDSB / 4 > 0 ? LDSB / 4 : 1];
    if (p == nullptr) { s[threadIdx.x] = 1.f; __syncthreads(); p[0] = s[0]; }   // never taken; keeps the LDS alive
}
__global__ __launch_bounds__(256) void k_touch(float* p) {   // one load + one store per thread
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    p[i] = p[i] + 1.f;
}
template <typename F>
static float run(hipStream_t st, int nodes, int replays, F launch) {
    hipGraph_t g; hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    hipGraphLaunch(e, st); hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < replays; ++r) hipGraphLaunch(e, st);
    hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(e); hipGraphDestroy(g);
    return ms * 1000.f / (nodes * replays);
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    float* buf; hipMalloc(&buf, 64 << 20);
    const int nodes = 80, reps = 50;
    printf("empty 1x64      : %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(1), dim3(64), 0, st, buf); }));
    printf("empty 256x256   : %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(256), dim3(256), 0, st, buf); }));
    printf("empty 256x256 LDS 48K: %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_empty<49152>, dim3(256), dim3(256), 0, st, buf); }));
    printf("empty 16x16 grid 256 thr: %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_empty<49152>, dim3(16, 16), dim3(256), 0, st, buf); }));
    printf("empty 2048x256  : %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_empty<0>, dim3(2048), dim3(256), 0, st, buf); }));
    printf("touch 256x256 (256 KB rw): %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, buf); }));
    printf("touch 6144x256 (6 MB rw) : %.2f us/node\n", run(st, nodes, reps, [&] { hipLaunchKernelGGL(k_touch, dim3(6144), dim3(256), 0, st, buf); }));
    return 0;
}
