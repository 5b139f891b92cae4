// NEGATIVE RESULT, kept for the record.  A 2.5-instruction (hi, scaled lo) split -- v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 computing
// fma(hi as fp16 operand, -2048, 2048 x) -- is bit-identical to the plain expression (0 mismatches below) but SLOWER in every kernel it was
// rolled out to (round 3: 1-D step 364.5 -> 365.9 us, config 5 step 5.69 -> 5.90 ms, design gradient 37.3 -> 37.8 ms, LinearAttention
// backward pass A 2446 -> 2586 us): the inline asm pins the schedule and the mix instructions are no cheaper than what they replace.
// The test itself: the mix split against the plain
// expression hi = (_Float16)x, lo = (_Float16)((x - (float)hi) * 2048) on 2^21 values with exponents 2^-67 .. 2^32 (fp16
// underflow, overflow to infinity and the denormal range included): must print 0 mismatches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/split_mix.hip -o tools/micro/split_mix.bin
#include <hip/hip_runtime.h>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const float k = 2048.0f;
    const float nk = -2048.0f;
    float s0 = x0 * k, s1 = x1 * k;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    unsigned l = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "s"(nk), "v"(s0));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "s"(nk), "v"(s1));
    lo = l;
}
__global__ void k(const float* x, unsigned* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    unsigned h, l;
    split2(a, b, h, l);
    half2v rh, rl;
    rh[0] = (_Float16)a; rh[1] = (_Float16)b;
    rl[0] = (_Float16)((a - (float)rh[0]) * 2048.0f); rl[1] = (_Float16)((b - (float)rh[1]) * 2048.0f);
    o[4 * i] = h; o[4 * i + 1] = l; o[4 * i + 2] = __builtin_bit_cast(unsigned, rh); o[4 * i + 3] = __builtin_bit_cast(unsigned, rl);
}
int main() {
    const int n = 1 << 20;
    float* x; unsigned* o;
    hipMallocManaged(&x, n * 2 * 4); hipMallocManaged(&o, n * 4 * 4);
    unsigned s = 12345;
    for (int i = 0; i < 2 * n; ++i) { s = s * 1664525u + 1013904223u; unsigned e = 60 + (s >> 8) % 100; unsigned b = (s & 0x80000000u) | (e << 23) | ((s >> 3) & 0x7fffff); x[i] = __builtin_bit_cast(float, b); }
    x[0] = 0.f; x[1] = 1.0f; x[2] = 65504.f; x[3] = -3.0e-8f; x[4] = 1e-3f; x[5] = 0.33333f;
    k<<<n / 256, 256>>>(x, o, n);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o[4 * i] != o[4 * i + 2] || o[4 * i + 1] != o[4 * i + 3]) { if (bad < 5) printf("mismatch %d: %g %g  %08x %08x vs %08x %08x\n", i, x[2 * i], x[2 * i + 1], o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]); ++bad; }
    printf("mismatches: %d of %d\n", bad, n);
}
