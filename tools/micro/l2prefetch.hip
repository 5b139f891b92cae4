// Micro-benchmark: does a side-branch prefetch kernel (8 blocks per XCD reading the NEXT launch's weights while the
// current launch runs) turn the cold weight stream of a dependent launch chain into an L2-hot one?
//   chain: 20 launches of k_stream (16 x 16 workgroups, the 16 m-tiles of an n-tile share 327 KB), each on a DIFFERENT
//   5.2 MB weight set (so nothing is L2-hot by itself), replayed as one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o l2prefetch.bin l2prefetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int G>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ W, int nloads, uint32_t* sink) {
    const int tid = threadIdx.x;
    const uint4* p = W + (size_t)blockIdx.x * nloads * 256 + tid;
    uint4 acc = {0, 0, 0, 0};
    uint4 r0[G], r1[G];
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

// block b runs on XCD b % 8 (observed dispatch rule): it reads 1/PARTS of the regions of n-tiles x and x + 8
__global__ __launch_bounds__(256) void k_prefetch(const uint4* __restrict__ W, int nloads, int parts, uint32_t* sink) {
    const int x = blockIdx.x & 7, part = blockIdx.x >> 3;
    const int per = nloads / parts;                       // 16-byte loads per thread and n-tile
    uint4 acc = {0, 0, 0, 0};
    for (int h = 0; h < 2; ++h) {
        const uint4* p = W + ((size_t)(x + 8 * h) * nloads + (size_t)part * per) * 256 + threadIdx.x;
        for (int j = 0; j < per; ++j) { const uint4 v = p[(size_t)j * 256]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}

int main() {
    hipStream_t st, side; hipStreamCreate(&st); hipStreamCreate(&side);
    const int nloads = 80, layers = 20;
    const size_t layer_u4 = (size_t)16 * nloads * 256;
    uint4* W; hipMalloc(&W, layers * layer_u4 * 16); hipMemset(W, 1, layers * layer_u4 * 16);
    uint32_t* sink; hipMalloc(&sink, 4096);
    const dim3 grid(16, 16);
    for (int mode = 0; mode < 4; ++mode) {                // 0: no prefetch; 1: side-branch prefetch 1 ahead; 2: 2 ahead; 3: prefetch in the SAME stream (serial, upper bound on benefit / cost)
        hipGraph_t g; hipGraphExec_t e;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        std::vector<hipEvent_t> evs;
        for (int i = 0; i < layers; ++i) {
            if (mode == 1 || mode == 2) {
                const int tgt = i + mode;                 // while launch i runs, prefetch the weights of launch i + mode
                if (tgt < layers) {
                    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming); evs.push_back(ev);
                    hipEventRecord(ev, st);               // after launch i - 1
                    hipStreamWaitEvent(side, ev, 0);
                    hipLaunchKernelGGL(k_prefetch, dim3(64), dim3(256), 0, side, W + (size_t)tgt * layer_u4, nloads, 8, sink);
                }
            }
            if (mode == 3 && i + 1 < layers)
                hipLaunchKernelGGL(k_prefetch, dim3(64), dim3(256), 0, st, W + (size_t)(i + 1) * layer_u4, nloads, 8, sink);
            hipLaunchKernelGGL((k_stream<20>), grid, dim3(256), 0, st, W + (size_t)i * layer_u4, nloads, sink);
        }
        if (mode == 1 || mode == 2) {                     // join the side stream
            hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming); evs.push_back(ev);
            hipEventRecord(ev, side); hipStreamWaitEvent(st, ev, 0);
        }
        hipError_t ce = hipStreamEndCapture(st, &g);
        if (ce != hipSuccess) { printf("capture failed: %s\n", hipGetErrorString(ce)); return 1; }
        hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
        hipGraphLaunch(e, st); hipStreamSynchronize(st);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int reps = 50;
        hipEventRecord(a, st);
        for (int r = 0; r < reps; ++r) hipGraphLaunch(e, st);
        hipEventRecord(b, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d: %.2f us per chain launch\n", mode, ms * 1000.f / (layers * reps));
        hipGraphExecDestroy(e); hipGraphDestroy(g);
    }
    return 0;
}
