// Micro-benchmark of the weight stream of the deep-level convolutions: 16 x 16 workgroups of 4 waves; the 16
// workgroups of one n-tile read the same 327 KB (80 x 16 B per thread, 1 KiB contiguous per wave instruction), as
// conv_gemm_h3_kernel does.  Reports us per launch inside a replayed graph of 20 dependent launches.
//   hipcc --offload-arch=gfx950 -O3 -o bstream.bin bstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// G loads in flight per thread, software-pipelined: group g+1 is issued before group g is consumed
template <int G, int SHARE>   // SHARE 1: the 16 m-tiles share the n-tile's weights; 0: private region per workgroup
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ W, int nloads, uint32_t* sink, int a_loads, const uint4* __restrict__ A) {
    const int tid = threadIdx.x;
    const size_t region = (size_t)nloads * 256;
    const uint4* p = W + (SHARE ? (size_t)blockIdx.x : (size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * region + tid;
    uint4 acc = {0, 0, 0, 0};
    uint4 r0[G], r1[G];
    // optional A traffic: a_loads x 16 B per thread, shared by the 16 n-tiles of an m-tile
    const uint4* ap = A + (size_t)blockIdx.y * a_loads * 256 + tid;
    for (int i = 0; i < a_loads; ++i) { const uint4 v = ap[(size_t)i * 256]; acc.x ^= v.x; acc.y ^= v.y; }
#pragma unroll
    for (int j = 0; j < G; ++j) r0[j] = p[(size_t)j * 256];
    for (int g = 0; g < nloads / G; g += 2) {
        const int g1 = min(g + 1, nloads / G - 1), g2 = min(g + 2, nloads / G - 1);
#pragma unroll
        for (int j = 0; j < G; ++j) r1[j] = p[((size_t)g1 * G + j) * 256];
#pragma unroll
        for (int j = 0; j < G; ++j) { acc.x ^= r0[j].x; acc.y ^= r0[j].y; acc.z ^= r0[j].z; acc.w ^= r0[j].w; }
#pragma unroll
        for (int j = 0; j < G; ++j) r0[j] = p[((size_t)g2 * G + j) * 256];
#pragma unroll
        for (int j = 0; j < G; ++j) { acc.x ^= r1[j].x; acc.y ^= r1[j].y; acc.z ^= r1[j].z; acc.w ^= r1[j].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

template <typename F>
static float run(hipStream_t st, int nodes, int replays, F launch) {
    hipGraph_t g; hipGraphExec_t e;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; ++i) launch(i);
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
    const int nloads = 80;                                  // per thread: 4 stages x 5 taps x 4 fragments
    const size_t layer_u4 = (size_t)16 * nloads * 256;      // one layer's weights (5.2 MB)
    const int layers = 20;
    uint4* W; hipMalloc(&W, (size_t)256 * nloads * 256 * 16 + layers * layer_u4 * 16);
    hipMemset(W, 1, (size_t)256 * nloads * 256 * 16 + layers * layer_u4 * 16);
    uint4* A; hipMalloc(&A, (size_t)16 * 64 * 256 * 16); hipMemset(A, 2, (size_t)16 * 64 * 256 * 16);
    uint32_t* sink; hipMalloc(&sink, 4096);
    const dim3 grid(16, 16);
    const int nodes = 20, reps = 50;
#define RUN(G, SH, LBL, LAYERFN, AL) printf("%-58s %.2f us/launch\n", LBL, run(st, nodes, reps, [&](int i) { \
        hipLaunchKernelGGL((k_stream<G, SH>), grid, dim3(256), 0, st, W + (size_t)(LAYERFN) * layer_u4, nloads, sink, AL, A); }));
    RUN(20, 1, "shared n-tile weights, 20 in flight, same layer (L2 hot)", 0, 0);
    RUN(20, 1, "shared, 20 in flight, 20 different layers (L2 cold)", i, 0);
    RUN(10, 1, "shared, 10 in flight, different layers", i, 0);
    RUN(40, 1, "shared, 40 in flight, different layers", i, 0);
    RUN(20, 0, "private 327 KB per workgroup (84 MB), 20 in flight", 0, 0);
    RUN(20, 1, "shared, 20 in flight, different layers, + 24 A loads/thread", i, 24);
    // half the bytes per workgroup (what a 2x larger m-tile would see per unit of work)
    {
        const int nl = 40;
        printf("%-58s %.2f us/launch\n", "shared, 40 loads per thread (164 KB), 20 in flight", run(st, nodes, reps, [&](int i) {
            hipLaunchKernelGGL((k_stream<20, 1>), grid, dim3(256), 0, st, W + (size_t)i * layer_u4, nl, sink, 0, A); }));
    }
    printf("%-58s %.2f us/launch\n", "empty-ish (0 loads)", run(st, nodes, reps, [&](int i) {
        hipLaunchKernelGGL((k_stream<20, 1>), grid, dim3(256), 0, st, W, 0, sink, 0, A); }));
    return 0;
}
