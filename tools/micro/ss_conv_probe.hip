// Micro-benchmark (round 5, evidence for DESIGN section 7): what would ONE wave per SIMD that does BOTH jobs of conv2d_ws_kernel --
// the 432 MFMAs of an item AND the staging / store work of its memory wave, interleaved in one instruction stream -- need per item?
// conv2d_ws_kernel (wave-specialised: a matrix wave and a memory wave per SIMD) needs 6.4 - 7.4 us per item where the matrix work
// alone is 3.6 us; MI355X_MICROARCH.md says two waves of a SIMD share its VALU issue by priority and age and that moving work
// between them is zero-sum.  This probe has the same per-item instruction mix and memory pattern as a 64 -> 64 channel 3x3 layer
// at 64 x 64 (8 x 16 pixel tile, 10 x 18 window, hi / lo fp16 planes in LDS, weight fragments from L2 in a 3-tap ring, output tile
// through LDS to float4 row stores) but NOT its arithmetic contract (borders are clamped, no GroupNorm statistics): timing only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -o ss_conv_probe.bin ss_conv_probe.hip && ./ss_conv_probe.bin
// (the threshold: without it the 72-step loop is only partly unrolled and the register arrays land in scratch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SW = 18, R = 10 * SW, PITCH = 144, PLANE = R * PITCH, LDT = 132, NP = 12;
constexpr float H3_SCALE = 2048.0f, H3_INV = 1.0f / 2048.0f;
struct Args { const float* x; const uint4* W; float* out; int items; int NI; };

__device__ __forceinline__ float silu_f(float x) {
    const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// FLAGS: bit 0 = GroupNorm-style fma + SiLU while staging, bit 1 = no staging / store work at all (MFMA stream alone),
// bit 2 = no MFMAs (staging / store stream alone)
template <int FLAGS>
__global__ __launch_bounds__(256) void ss_kernel(const Args a) {
    constexpr bool GN = FLAGS & 1, NOSTAGE = FLAGS & 2, NOMMA = FLAGS & 4, NOTILE = FLAGS & 8, NOCONV = FLAGS & 16, NOLOAD = FLAGS & 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) float Tile[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ph = w & 1, nh = w >> 1;
    const int r0 = tid >> 4, c4 = tid & 15;
    const float fa = 1.0001f, fb = 0.001f;
    float4 areg0[NP], areg1[NP];                          // (item k + 1's window being converted | item k + 2's loads in flight)
    auto origin = [&](int k, int& img, int& ty0, int& tx0) {
        const int t = blockIdx.x * a.items + k;
        img = (t >> 5) % a.NI; const int ti = t & 31;
        ty0 = (ti >> 2) * 8; tx0 = (ti & 3) * 16;
    };
    auto load_item = [&](int k, float4 (&ar)[NP]) __attribute__((always_inline)) {
        int img, ty0, tx0; origin(k, img, ty0, tx0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = min(r0 + 16 * p, R - 1);
            const int hy = r / SW, hx = r - hy * SW;
            const int y = min(max(ty0 + hy - 1, 0), 63), x = min(max(tx0 + hx - 1, 0), 63);
            ar[p] = *reinterpret_cast<const float4*>(a.x + (((size_t)img * 64 + y) * 64 + x) * 64 + c4 * 4);
        }
    };
    auto stage_one = [&](int buf, const float4 (&ar)[NP], int p) __attribute__((always_inline)) {
        const int r = r0 + 16 * p;
        float4 v = ar[p];
        if constexpr (GN) {
            v.x = silu_f(__builtin_fmaf(v.x, fa, fb)); v.y = silu_f(__builtin_fmaf(v.y, fa, fb));
            v.z = silu_f(__builtin_fmaf(v.z, fa, fb)); v.w = silu_f(__builtin_fmaf(v.w, fa, fb));
        }
        const bool ok = (r & 31) != 31;                      // a select per element, as the border masks of the real kernel
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        half4v hi, lo;
        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
        lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
        if (r < R) {
            *reinterpret_cast<half4v*>(&smem[buf][0] + r * PITCH + c4 * 8) = hi;
            *reinterpret_cast<half4v*>(&smem[buf][0] + PLANE + r * PITCH + c4 * 8) = lo;
        }
    };
    f32x4 accM[4][2], accL[4][2];
    half8 breg[3][2][2][2];
    const uint4* wbase = a.W + nh * 128 + lane;
    auto load_b = [&](int tap, int slot_) __attribute__((always_inline)) {
        const uint4* wp = wbase + (size_t)tap * 4 * 256;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int q = 0; q < 4; ++q) breg[slot_][kh][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256 + kh * 64]);
    };
    const int foff = (lane & 15) * PITCH + (lane >> 4) * 16;
    // item k: multiply from buffer k & 1 while item k + 1's window (registers `cur`, loaded one item earlier) is converted into the
    // other buffer, one staged float4 per six steps; item k + 2's loads are requested at the top into `nxt`
    auto item = [&](int k) __attribute__((always_inline)) {
        float4 (&cur)[NP] = areg0;
        float4 (&nxt)[NP] = areg1;
        if (!NOSTAGE && !NOLOAD && k + 2 < a.items) load_item(k + 2, nxt);
        const unsigned char* P0 = &smem[k & 1][0] + foff;
        const unsigned char* P1 = P0 + PLANE;
        half8 fh[3], fl[3];
        auto read_frag = [&](int s, int slot_) __attribute__((always_inline)) {
            const int tap = s >> 3, kh = (s >> 2) & 1, mb = s & 3;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int o = ((4 * ph + mb + dy) * SW + dx) * PITCH + kh * 64;
            fh[slot_] = *reinterpret_cast<const half8*>(P0 + o);
            fl[slot_] = *reinterpret_cast<const half8*>(P1 + o);
        };
        if (!NOMMA) { read_frag(0, 0); read_frag(1, 1); }
#pragma unroll
        for (int s = 0; s < 72; ++s) {
            const int tap = s >> 3, kh = (s >> 2) & 1, mb = s & 3, bs = tap % 3, fs = s % 3;
            if constexpr (!NOMMA) {
                if (s + 2 < 72) read_frag(s + 2, (s + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bool z = tap == 0 && kh == 0;
                accM[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][kh][0][0], z ? zero : accM[mb][0], 0, 0, 0);
                accL[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][kh][0][1], z ? zero : accL[mb][0], 0, 0, 0);
                accM[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][kh][1][0], z ? zero : accM[mb][1], 0, 0, 0);
                accL[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][kh][1][1], z ? zero : accL[mb][1], 0, 0, 0);
                accL[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[fs], breg[bs][kh][0][0], accL[mb][0], 0, 0, 0);
                accL[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[fs], breg[bs][kh][1][0], accL[mb][1], 0, 0, 0);
                if ((s & 7) == 7) load_b((tap + 3) % 9, bs);
            }
            if constexpr (!NOSTAGE && !NOCONV) { if (s % 6 == 5 && k + 1 < a.items) stage_one((k + 1) & 1, cur, s / 6); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // finished tile: accumulators -> channel-major LDS tile -> float4 rows
        if constexpr (!NOSTAGE && !NOTILE) {
            float* const trow = Tile + (nh * 32 + (lane & 15)) * LDT + (lane >> 4) * 4 + ph * 64;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float4 v;
                    v.x = accM[mb][nb][0] + accL[mb][nb][0] * H3_INV; v.y = accM[mb][nb][1] + accL[mb][nb][1] * H3_INV;
                    v.z = accM[mb][nb][2] + accL[mb][nb][2] * H3_INV; v.w = accM[mb][nb][3] + accL[mb][nb][3] * H3_INV;
                    *reinterpret_cast<float4*>(trow + nb * 16 * LDT + mb * 16) = v;
                }
        }
        __syncthreads();                                      // the tile is complete; the next window is staged; this window is consumed
        if constexpr (!NOSTAGE && !NOTILE) {
            int img, ty0, tx0; origin(k, img, ty0, tx0);
            const int oc4 = lane & 15, q = lane >> 4;
            float* o0 = a.out + (((size_t)img * 64 + ty0 + 2 * w) * 64 + tx0) * 64 + oc4 * 4;
            const float* t0 = Tile + (oc4 * 4) * LDT + 32 * w + 4 * q;
            float4 f[2][4];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) f[jj][ci] = *reinterpret_cast<const float4*>(t0 + ci * LDT + 16 * jj);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                *reinterpret_cast<float4*>(o0 + (jj * 64 + 4 * q + 0) * 64) = make_float4(f[jj][0].x, f[jj][1].x, f[jj][2].x, f[jj][3].x);
                *reinterpret_cast<float4*>(o0 + (jj * 64 + 4 * q + 1) * 64) = make_float4(f[jj][0].y, f[jj][1].y, f[jj][2].y, f[jj][3].y);
                *reinterpret_cast<float4*>(o0 + (jj * 64 + 4 * q + 2) * 64) = make_float4(f[jj][0].z, f[jj][1].z, f[jj][2].z, f[jj][3].z);
                *reinterpret_cast<float4*>(o0 + (jj * 64 + 4 * q + 3) * 64) = make_float4(f[jj][0].w, f[jj][1].w, f[jj][2].w, f[jj][3].w);
            }
            __syncthreads();                                  // the tile has been read
        }
    };
    load_b(0, 0); load_b(1, 1); load_b(2, 2);
    load_item(0, areg0);
#pragma unroll
    for (int p = 0; p < NP; ++p) stage_one(0, areg0, p);
    if (a.items > 1) load_item(1, areg0);
    __syncthreads();
    for (int k = 0; k < a.items; ++k) {
        item(k);
#pragma unroll
        for (int p = 0; p < NP; ++p) areg0[p] = areg1[p];   // (the loads were requested a whole item ago)
    }
    if (NOSTAGE || NOTILE) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) t += accM[i][j][e] + accL[i][j][e];
        if (t == 123.456f) a.out[0] = 1.f;
    }
}

// The MFMA stream alone in another division of the item: wave = (pixel half, K half) with ALL 64 output channels (128 accumulator
// registers): an A fragment feeds 12 MFMAs instead of 6 -- 4 KB of LDS fragments per 192 clocks and wave instead of per 96 -- the weight
// fragments per wave and tap stay 8 KB.  (The k-group reduction this needs per finished tile is not in the probe.)
template <int RING, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void mm64_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ph = w & 1, kg = w >> 1;
    for (int i = tid; i < 2 * 2 * PLANE / 16; i += 256) reinterpret_cast<uint4*>(&smem[0][0])[i] = make_uint4(0x2c002c00u, 0x2c002c00u, 0x2c002c00u, 0x2c002c00u);
    __syncthreads();
    f32x4 accM[4][4], accL[4][4];
    half8 breg[RING][4][2];                                   // [ring slot][column block][plane]
    const uint4* wbase = a.W + kg * 64 + lane;
    auto load_b = [&](int tap, int slot_) __attribute__((always_inline)) {
        const uint4* wp = wbase + (size_t)tap * 4 * 256;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int q = 0; q < 4; ++q) breg[slot_][2 * nh + (q >> 1)][q & 1] = __builtin_bit_cast(half8, wp[q * 256 + nh * 128]);
    };
    const int foff = (lane & 15) * PITCH + (lane >> 4) * 16 + kg * 64;
    load_b(0, 0); load_b(1, 1); if (RING == 3) load_b(2, 2);
    for (int k = 0; k < a.items; ++k) {
        const unsigned char* P0 = &smem[k & 1][0] + foff;
        const unsigned char* P1 = P0 + PLANE;
        half8 fh[3], fl[3];
        auto read_frag = [&](int s, int slot_) __attribute__((always_inline)) {
            const int tap = s >> 2, mb = s & 3;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int o = ((4 * ph + mb + dy) * SW + dx) * PITCH;
            fh[slot_] = *reinterpret_cast<const half8*>(P0 + o);
            fl[slot_] = *reinterpret_cast<const half8*>(P1 + o);
        };
        read_frag(0, 0); read_frag(1, 1);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int tap = s >> 2, mb = s & 3, bs = tap % RING, fs = s % 3;
            if (s + 2 < 36) read_frag(s + 2, (s + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bool z = tap == 0;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][nb][0], z ? zero : accM[mb][nb], 0, 0, 0);
                accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][nb][1], z ? zero : accL[mb][nb], 0, 0, 0);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[fs], breg[bs][nb][0], accL[mb][nb], 0, 0, 0);
            if ((s & 3) == 3) load_b((tap + RING) % 9, bs);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) t += accM[i][j][e] + accL[i][j][e];
    if (t == 123.456f) a.out[0] = 1.f;
}

template <int FLAGS>
static float run(const Args& a, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(ss_kernel<FLAGS>, dim3(256), dim3(256), 0, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ss_kernel<FLAGS>, dim3(256), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const int NI = 128, items = 16;                          // 256 workgroups x 16 tiles = 128 images x 32 tiles: one 64 -> 64 layer of config 5
    Args a{};
    float* x; hipMalloc(&x, (size_t)NI * 64 * 64 * 64 * 4);
    float* out; hipMalloc(&out, (size_t)NI * 64 * 64 * 64 * 4);
    uint4* W; hipMalloc(&W, (size_t)9 * 4 * 256 * 16);
    std::vector<float> hx((size_t)NI * 64 * 64 * 64);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)9 * 4 * 256 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x2c00 + (i * 7919u) % 0x3ff);    // small positive halfs
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    a.x = x; a.W = W; a.out = out; a.items = items; a.NI = NI;
    const int reps = 20;
    const float t0 = run<0>(a, reps), t1 = run<1>(a, reps), t2 = run<2>(a, reps), t4 = run<4>(a, reps), t5 = run<5>(a, reps);
    printf("  ablations of the plain variant: no tile / stores %.2f, no conversion %.2f, no window loads %.2f, neither %.2f us per item\n",
           run<8>(a, reps) / items, run<16>(a, reps) / items, run<32>(a, reps) / items, run<8 + 16 + 32>(a, reps) / items);
    printf("one wave per SIMD, both jobs in one stream; %d items per workgroup, 256 workgroups (a 64 -> 64 layer at 64 x 64 x 128 images)\n", items);
    printf("  plain staging              : %7.1f us per launch = %.2f us per item   (conv2d_ws_kernel: ~110 us, 6.4 us per item)\n", t0, t0 / items);
    printf("  fma + SiLU while staging   : %7.1f us per launch = %.2f us per item   (conv2d_ws_kernel: ~135 us, 7.4 us per item)\n", t1, t1 / items);
    printf("  MFMA stream alone          : %7.1f us per launch = %.2f us per item\n", t2, t2 / items);
    auto time64 = [&](auto kern, const char* what) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, a); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        printf("  MFMA stream alone, wave = 64 pixels x 64 channels x half K, %s: %7.1f us per launch = %.2f us per item\n", what, ms * 1e3f / reps, ms * 1e3f / reps / items);
    };
    time64(mm64_kernel<3>, "weights two taps ahead");
    time64(mm64_kernel<2>, "weights one tap ahead ");
    time64(mm64_kernel<2, 2>, "one tap ahead, 256 registers");
    printf("  staging / store alone      : %7.1f us per launch = %.2f us per item (plain), %.2f (fma + SiLU)\n", t4, t4 / items, t5 / items);
    return 0;
}
