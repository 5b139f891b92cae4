// Write-stream microbenchmark: 3.1 M rows x 384 floats (4.8 GB, the qkv projection's output at the 64 x 64 level).
//   A: a wave owns 16 rows; every store instruction writes 16 rows x 64 B (the MFMA accumulator layout: 4 lanes x float4 per row)
//   B: a wave owns 16 rows; every store instruction writes ONE KiB of one row (64 lanes x float4 contiguous)
//   C: as A but the 24 instructions of a wave walk a row's 64-byte pieces in address order per row quad (same as A, order only)
// hipcc --offload-arch=gfx950 -O3 -o write_pattern.bin write_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k_write(float* __restrict__ y, int64_t rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * 16;
    if (row0 >= rows) return;
    const float4 v = make_float4((float)lane, 1.f, 2.f, 3.f);
    if (MODE == 0) {
        const int lp = lane & 15, q = lane >> 4;
#pragma unroll
        for (int c = 0; c < 24; ++c) *reinterpret_cast<float4*>(y + (row0 + lp) * 384 + c * 16 + q * 4) = v;
    } else {
#pragma unroll
        for (int c = 0; c < 24; ++c) {                       // instruction c: row c / 1.5 ... : 16 rows x 1536 B = 24 KiB = 24 x 1 KiB
            float* p = y + row0 * 384 + (int64_t)c * 256 + lane * 4;
            *reinterpret_cast<float4*>(p) = v;
        }
    }
}
int main() {
    const int64_t rows = 768LL * 4096;
    float* y; hipMalloc(&y, rows * 384 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(rows / 64);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, y, rows);
            else hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, y, rows);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms  %.2f TB/s\n", mode == 0 ? "A 16 rows x 64 B per instruction" : "B 1 KiB contiguous per instruction", ms, rows * 384 * 4.0 / ms / 1e9);
        }
    return 0;
}
