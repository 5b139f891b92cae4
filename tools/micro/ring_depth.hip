// Micro-benchmark behind DESIGN 4.1d (round 6): does the weight stream of dconv2_kernel's K loops run at a THROUGHPUT limit of the
// L2 -> L1 path, or at one ring of requests per round trip (a latency limit that a deeper ring would lift)?
// 16 x 16 workgroups of 4 waves as the deep convolutions; the 16 workgroups of an n-tile read the same 327 KB (80 x 16 B per thread).
// A ring of G requests per thread: slot j is consumed (an empty asm statement that needs the registers, or MF MFMAs per 4 slots = one tap's matrix work) and
// re-requested at once, as load_b_tap does behind a tap's last use.  G = 20 is the kernel's ring (5 taps x 4 fragments).
//   hipcc --offload-arch=gfx950 -O3 -o ring_depth.bin ring_depth.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int G, int MF>
__global__ __launch_bounds__(256) void k_ring(const uint4* __restrict__ W, int nloads_rt, float* sink) {
    constexpr int nloads = 240;               // compile-time: the loop is fully unrolled so that every wait is an exact vmcnt(N), as in the kernel
    if (nloads_rt == 0) return;
    const int tid = threadIdx.x;
    const uint4* p = W + (size_t)blockIdx.x * nloads * 256 + tid;
    uint4 r[G];
    f32x4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 x = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < G; ++j) r[j] = p[(size_t)j * 256];
    const half8 fh = __builtin_bit_cast(half8, make_uint4(tid, 1, 2, 3));
#pragma unroll
    for (int i = 0; i < nloads; i += G) {
#pragma unroll
        for (int j = 0; j < G; j += 4) {
            if (MF > 0) {
#pragma unroll
                for (int m = 0; m < MF; ++m)
                    acc[m % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh, __builtin_bit_cast(half8, r[j + (m & 3)]), acc[m % 6], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(r[j + q].x), "v"(r[j + q].y), "v"(r[j + q].z), "v"(r[j + q].w));      // waited for, nothing issued
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) if (i + G + j + q < nloads) r[j + q] = p[(size_t)(i + G + j + q) * 256];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1234.5f || (x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u) sink[tid] = s;
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
    const int nloads = 240;                                 // per thread: 12 stages x 5 taps x 4 fragments (983 KB per workgroup, 2 MB per XCD)
    const size_t layer_u4 = (size_t)16 * nloads * 256;      // one layer's weights (15.7 MB)
    const int layers = 5;                                   // 79 MB cycling: L2-cold, Infinity-Cache-resident (as the step's 83 MB of weights)
    uint4* W; hipMalloc(&W, layers * layer_u4 * 16);
    hipMemset(W, 0x11, layers * layer_u4 * 16);
    float* sink; hipMalloc(&sink, 4096);
    const dim3 grid(16, 16);
    const int nodes = 20, reps = 50;
    const float empty = run(st, nodes, reps, [&](int) { hipLaunchKernelGGL((k_ring<20, 0>), grid, dim3(256), 0, st, W, 0, sink); });
    printf("empty launch (0 loads): %.2f us/launch; below: us/launch minus that, and GB/s per CU for the 983 KB\n", empty);
#define RUN(G, MF, HOT) { const float t = run(st, nodes, reps, [&](int i) { \
        hipLaunchKernelGGL((k_ring<G, MF>), grid, dim3(256), 0, st, W + (size_t)((HOT) ? 0 : i % 5) * layer_u4, nloads, sink); }) - empty; \
        printf("ring of %2d requests per thread, %2d MFMAs per tap, %s: %5.2f us  %6.1f GB/s per CU\n", G, MF, (HOT) ? "L2-hot " : "5 layers ", t, 983.04f / t); }
    RUN(8, 0, 1) RUN(12, 0, 1) RUN(20, 0, 1) RUN(24, 0, 1) RUN(40, 0, 1) RUN(60, 0, 1)
    RUN(8, 0, 0) RUN(12, 0, 0) RUN(20, 0, 0) RUN(24, 0, 0) RUN(40, 0, 0) RUN(60, 0, 0)
    RUN(20, 18, 1) RUN(24, 18, 1) RUN(40, 18, 1) RUN(60, 18, 1)
    RUN(20, 18, 0) RUN(24, 18, 0) RUN(40, 18, 0) RUN(60, 18, 0)
    return 0;
}
