// Micro-benchmark: how long after a wave starts does its first kernel argument arrive (s_load from the kernarg segment),
// for dependent kernel nodes of a replayed hipGraph?  Each workgroup's wave 0 stamps wall_clock64 (100 MHz) at entry, after
// the first kernarg-dependent scalar is available, and after a first global load through a kernarg pointer.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { const float* in; unsigned long long* out; int node; int pad[40]; };
__global__ __launch_bounds__(256) void k_probe(const Big a) {
    const unsigned long long t0 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    int node = a.node; asm volatile("" : "+s"(node));                    // first kernarg word is here
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    float v = a.in[(size_t)blockIdx.x * 256 + threadIdx.x]; asm volatile("" : "+v"(v));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { unsigned long long* o = a.out + ((size_t)node * gridDim.x + blockIdx.x) * 3; o[0] = t0; o[1] = t1; o[2] = t2; }
    if (v == 12345.f) a.out[0] = 1;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int nodes = 20, grid = 256;
    float* in; hipMalloc(&in, grid * 256 * 4); hipMemset(in, 0, grid * 256 * 4);
    unsigned long long* out; hipMalloc(&out, (size_t)nodes * grid * 3 * 8);
    hipGraph_t g; hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) { Big a{}; a.in = in; a.out = out; a.node = i; hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, st, a); }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(e, st);
    hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nodes * grid * 3);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    for (int n : {1, 5, 10, 19}) {
        std::vector<double> d1, d2; unsigned long long first = ~0ull, prev_last = 0;
        for (int b = 0; b < grid; ++b) { const unsigned long long* o = &h[((size_t)n * grid + b) * 3]; d1.push_back((o[1] - o[0]) * 0.01); d2.push_back((o[2] - o[1]) * 0.01); first = std::min(first, o[0]); }
        for (int b = 0; b < grid; ++b) prev_last = std::max(prev_last, h[((size_t)(n - 1) * grid + b) * 3 + 2]);
        std::sort(d1.begin(), d1.end()); std::sort(d2.begin(), d2.end());
        printf("node %2d: kernarg wait median %.2f us (p90 %.2f), first global load %.2f us (p90 %.2f); previous node's last stamp -> this node's first entry %.2f us\n",
               n, d1[grid / 2], d1[grid * 9 / 10], d2[grid / 2], d2[grid * 9 / 10], ((double)first - (double)prev_last) * 0.01);
    }
    return 0;
}
