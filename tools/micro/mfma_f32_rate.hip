// fp32 MFMA (v_mfma_f32_16x16x4_f32) issue rate: 4 independent accumulator chains per wave, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate.bin mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + 1.f, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + 2.f, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + 3.f, acc[3], 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int WAVES> void run(float* out, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<WAVES>, dim3(grid), dim3(64 * WAVES), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * WAVES * iters * 32 * 2048.0;
        printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
    run<4>(out, "4 waves per workgroup, 8 workgroups per CU");
    run<1>(out, "1 wave per workgroup");
    return 0;
}
