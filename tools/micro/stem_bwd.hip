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
// --check: two images whose gradient rows differ in magnitude by up to 2^60 (row r of image 0 scaled by 2^(5 (r % 13) - 30), image 1
// by 2^(20 - 3 (r % 7)) with rows 10..19 exactly zero) against a double-precision evaluation of the same sum on the host: exercises
// the window-minimum scale and the accumulator rescale, which ordinary gradients never trigger.  Prints the largest error relative to
// the largest reference value of each output ROW (rows differ by 18 orders of magnitude: a global maximum would hide the small ones).
#include <cmath>
#include <cstring>
#include <vector>
static int check() {
    const int NI = 2, H = 64, WD = 64;
    std::vector<float> W(64 * 4 * 49), g((size_t)NI * H * WD * 64);
    unsigned s = 7;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 8388608.0f) - 1.0f; };
    for (auto& v : W) v = rnd() * 0.1f;
    for (int i = 0; i < NI; ++i)
        for (int r = 0; r < H; ++r) {
            float sc = i == 0 ? std::ldexp(1.0f, 5 * (r % 13) - 30) : std::ldexp(1.0f, 20 - 3 * (r % 7));
            if (i == 1 && r >= 10 && r < 20) sc = 0.f;
            for (int x = 0; x < WD * 64; ++x) g[((size_t)i * H + r) * WD * 64 + x] = rnd() * sc;
        }
    // fragments as forceunet_host.inc packs them: [a][kk][nb][plane][lane][8 halfs]
    std::vector<uint16_t> frag((size_t)7 * 2 * 2 * 2 * 64 * 8);
    auto put = [](uint16_t* hi, uint16_t* lo, float v) { const _Float16 hv = (_Float16)v, lv = (_Float16)((v - (float)hv) * 2048.0f); std::memcpy(hi, &hv, 2); std::memcpy(lo, &lv, 2); };
    for (int ta = 0; ta < 7; ++ta) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) for (int kk = 0; kk < 2; ++kk) for (int nb = 0; nb < 2; ++nb) {
        const int lr = lane & 15, lq = lane >> 4, n = nb * 16 + lr, b_ = n >> 2, ci = n & 3, co = kk * 32 + lq * 8 + j;
        const float v = b_ < 7 ? W[((size_t)co * 4 + ci) * 49 + (6 - ta) * 7 + (6 - b_)] : 0.f;
        const size_t q = ((size_t)((ta * 2 + kk) * 2 + nb) * 2) * 64 * 8;
        put(&frag[q + (size_t)lane * 8 + j], &frag[q + 64 * 8 + (size_t)lane * 8 + j], v);
    }
    float *dg, *dW, *ddx;
    CK(hipMalloc(&dg, g.size() * 4)); CK(hipMalloc(&dW, frag.size() * 2)); CK(hipMalloc(&ddx, (size_t)NI * H * WD * 4 * 4));
    CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
    FuStemBwdH3Args a{}; a.g = dg; a.W = dW; a.dx = ddx; a.H = H; a.NI = NI; a.beta = 0.f;
#ifdef FU_STEM_PROF
    unsigned long long* prof; CK(hipMalloc(&prof, 64 * 8)); a.prof = prof;
#endif
    hipLaunchKernelGGL(fu_stem_bwd_h3_kernel, dim3(NI * (H / 16)), dim3(256), 0, 0, a); CK(hipDeviceSynchronize());
    std::vector<float> dx((size_t)NI * H * WD * 4);
    CK(hipMemcpy(dx.data(), ddx, dx.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0; int bad = 0;
    for (int i = 0; i < NI; ++i)
        for (int y = 0; y < H; ++y) {
            std::vector<double> ref(WD * 4, 0.0);
            double rowmax = 0.0;
            for (int x = 0; x < WD; ++x) for (int ci = 0; ci < 4; ++ci) {
                double acc = 0.0;
                for (int ty = 0; ty < 7; ++ty) { const int yy = y + 3 - ty; if (yy < 0 || yy >= H) continue;      // dx[y][x] = sum g[y + 3 - ty][x + 3 - tx][co] W[co][ci][ty][tx]
                    for (int tx = 0; tx < 7; ++tx) { const int xx = x + 3 - tx; if (xx < 0 || xx >= WD) continue;
                        const float* gp = &g[(((size_t)i * H + yy) * WD + xx) * 64];
                        for (int co = 0; co < 64; ++co) acc += (double)gp[co] * (double)W[((size_t)co * 4 + ci) * 49 + ty * 7 + tx];
                    } }
                ref[x * 4 + ci] = acc; rowmax = std::max(rowmax, std::fabs(acc));
            }
            for (int k = 0; k < WD * 4; ++k) {
                const double got = dx[((size_t)i * H + y) * WD * 4 + k];
                if (!std::isfinite(got)) { ++bad; continue; }
                if (rowmax > 0.0) worst = std::max(worst, std::fabs(got - ref[k]) / rowmax); else if (got != 0.0) ++bad;
            }
        }
    printf("check: worst row-relative error %.3e, non-finite or non-zero-where-zero %d\n", worst, bad);
    return (worst < 2e-6 && bad == 0) ? 0 : 1;
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--check")) return check();
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
