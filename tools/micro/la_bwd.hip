// Timing harness for the ForceUnet LinearAttention backward passes (forceunet_la.h) at one site's shape, outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DFU_LA_PROF] tools/micro/la_bwd.hip -o tools/micro/la_bwd.bin
//   tools/micro/la_bwd.bin [NI = 768] [HW = 4096] [C = 64]
// Weights are zero, x / dout pseudo-random: timing does not depend on the values.  With -DFU_LA_PROF workgroup 0 records
// s_memtime at every barrier of its second tile (printed as cycle deltas).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cindm_amd/csrc/forceunet_la.h"
using namespace cindm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f);
    }
}
template <class F> float timeit(F f, int n) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < n; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.f / n;
}
int main(int argc, char** argv) {
    const int NI = argc > 1 ? atoi(argv[1]) : 768, HW = argc > 2 ? atoi(argv[2]) : 4096, C = argc > 3 ? atoi(argv[3]) : 64;
    const size_t n = (size_t)NI * HW * C;
    auto dev = [](size_t nf, bool zero) { float* p; CK(hipMalloc(&p, nf * 4)); if (zero) CK(hipMemset(p, 0, nf * 4)); return p; };
    FuLaArgs a{};
    float* x = dev(n, false); float* dout = dev(n, false);
    fill<<<1024, 256>>>(x, n, 1); fill<<<1024, 256>>>(dout, n, 2);
    a.x = x; a.ldx = C; a.dout = dout;
    float* g1 = dev(C, false); float* g2 = dev(C, false); fill<<<1, 64>>>(g1, C, 3); fill<<<1, 64>>>(g2, C, 4);
    a.g1 = g1; a.g2 = g2;
    float* wq = dev((size_t)24 * C * 16 * 2, false); fill<<<64, 256>>>(wq, (size_t)24 * C * 16 * 2, 5);    // fp16 pairs: arbitrary finite bit patterns
    CK(hipMemset(wq, 0x2c, (size_t)24 * C * 16 * 2 * 4));       // 0x2c2c fp16 = 0.065
    a.Wqkv = wq;
    auto wz = [&](size_t nf) { float* p = dev(nf, false); CK(hipMemset(p, 0x28, nf * 4)); return p; };
    a.Wo = wz((size_t)C * 128 * 2); a.bo = dev(C, true); a.WoT = wz((size_t)128 * C * 2); a.WqT = wz((size_t)C * 128 * 2); a.WkvT = wz((size_t)C * 256 * 2);
    float* ctx = dev((size_t)NI * 4096, false); fill<<<256, 256>>>(ctx, (size_t)NI * 4096, 6); a.ctx = ctx;
    float* kst = dev((size_t)NI * 256, false); fill<<<64, 256>>>(kst, (size_t)NI * 256, 7); a.kst = kst;
    const int npx = C == 64 ? 64 : 32, ntl = HW / 64, wpi = ntl < 4 ? ntl : 4;
    a.dctx_part = dev((size_t)NI * wpi * 4096, true);
    float* dctx = dev((size_t)NI * 4096, false); fill<<<256, 256>>>(dctx, (size_t)NI * 4096, 8); a.dctx = dctx;
    a.T = dev((size_t)NI * 128, true);
    a.dyq = dev(n, true); a.dx = dev(n, true); a.beta = 0.f; a.HW = HW; a.tpw = HW / (npx * wpi); a.inv_n = 1.0f / HW;
#ifdef FU_LA_PROF
    unsigned long long* prof; CK(hipMalloc(&prof, 64 * 8)); CK(hipMemset(prof, 0, 64 * 8)); a.prof = prof;
#endif
    CK(hipDeviceSynchronize());
    const dim3 grid((unsigned)NI * wpi);
    float ta, tb;
    if (C == 64 && getenv("LA_NPX32")) {
        a.tpw = HW / (32 * wpi);
        ta = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_a_kernel<64, 32, 2>), grid, dim3(256), 0, 0, a); }, 10);
        tb = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_b_kernel<64, 32, 2>), grid, dim3(256), 0, 0, a); }, 10);
    } else if (C == 64) {
        ta = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_a_kernel<64, 64>), grid, dim3(256), 0, 0, a); }, 10);
        tb = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_b_kernel<64, 64>), grid, dim3(256), 0, 0, a); }, 10);
    } else {
        ta = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_a_kernel<128, 32>), grid, dim3(256), 0, 0, a); }, 10);
        tb = timeit([&] { hipLaunchKernelGGL((fu_la_bwd_b_kernel<128, 32>), grid, dim3(256), 0, 0, a); }, 10);
    }
    CK(hipDeviceSynchronize());
    const double gb = (double)n * 4 / 1e9;
    printf("NI %d HW %d C %d: pass A %.1f us (%.2f TB/s over x + dout + dyq), pass B %.1f us (%.2f TB/s over x x2 + dout + dyq + dx)\n", NI, HW, C,
           ta, 3 * gb / ta * 1e3, tb, 5 * gb / tb * 1e3);
#ifdef FU_LA_PROF
    unsigned long long h[64]; CK(hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost));
    printf("pass A phases (cycles of the 100 MHz s_memtime clock x 24 = shader cycles):");
    for (int i = 1; i < 16 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
    printf("\npass B phases:");
    for (int i = 33; i < 48 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
    printf("\n");
#endif
    return 0;
}
