// Timing harness for fu_stem_bwd_h3_kernel (forceunet_stem.h) outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DFU_STEM_PROF] tools/micro/stem_bwd.hip -o tools/micro/stem_bwd.bin
// With -DFU_STEM_PROF workgroup 5 records s_memtime around the phases of its tenth gradient row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../cindm_amd/csrc/forceunet_stem.h"
using namespace cindm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f) * 1e-3f;
    }
}
int main(int argc, char** argv) {
    const int NI = argc > 1 ? atoi(argv[1]) : 768, H = 64;
    const size_t n = (size_t)NI * H * 64 * 64;
    float *g, *W, *dx;
    CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&W, 7 * 2 * 2 * 2 * 64 * 16)); CK(hipMalloc(&dx, (size_t)NI * H * 64 * 4 * 4));
    fill<<<1024, 256>>>(g, n, 1); CK(hipMemset(W, 0x2c, 7 * 2 * 2 * 2 * 64 * 16));
    FuStemBwdH3Args a{}; a.g = g; a.W = W; a.dx = dx; a.H = H; a.NI = NI; a.beta = 0.f;
#ifdef FU_STEM_PROF
    unsigned long long* prof; CK(hipMalloc(&prof, 64 * 8)); CK(hipMemset(prof, 0, 64 * 8)); a.prof = prof;
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fu_stem_bwd_h3_kernel, dim3(NI * (H / 16)), dim3(256), 0, 0, a); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fu_stem_bwd_h3_kernel, dim3(NI * (H / 16)), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NI %d: %.1f us per launch (%.2f TB/s over g)\n", NI, ms * 100.f, n * 4 / (ms * 1e-4) / 1e12);
#ifdef FU_STEM_PROF
    unsigned long long h[64]; CK(hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost));
    printf("row phases (cycles): stage + loads %llu, row sum %llu, multiply %llu, max + park %llu, barrier %llu\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4]);
#endif
    return 0;
}
