// Checks the gfx950 v_permlane{16,32}_swap all-reduce helpers against __shfl_xor.  hipcc --offload-arch=gfx950 -O3 -o permlane_test permlane_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ float swap32_other(float v) {      // value of lane ^ 32
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    // a = [v[0..31], v[0..31]], b = [v[32..63], v[32..63]]
    return (threadIdx.x & 32) ? a : b;
}
__device__ __forceinline__ float swap16_other(float v) {      // value of lane ^ 16
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return (threadIdx.x & 16) ? a : b;
}
__global__ void k(const float* in, float* out) {
    const float v = in[threadIdx.x];
    out[threadIdx.x] = swap32_other(v);
    out[64 + threadIdx.x] = __shfl_xor(v, 32, 64);
    out[128 + threadIdx.x] = swap16_other(v);
    out[192 + threadIdx.x] = __shfl_xor(v, 16, 64);
}
int main() {
    float h[64], o[256], *d, *e;
    for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 3.f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { if (o[i] != o[64 + i]) ++bad; if (o[128 + i] != o[192 + i]) ++bad; }
    printf("permlane swap mismatches: %d\n", bad);
    return bad != 0;
}
