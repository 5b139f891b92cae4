// ForceUnet's 7x7 stem (init_conv, model/diffusion_2d.py:322, 4 -> 64 channels, padding 3) on the split-fp16 MFMA, forward and
// input gradient.  Round 2 ran both on the exact fp32 MFMA (fu_conv_kernel<7, 4, 0>: 1.03 ms, fu_stem_bwd_kernel: 1.14 ms and
// 3.8 GB per design-gradient call at 768 images -- the 4-row strips re-read their 6 halo rows from HBM).
//
// Forward (fu_stem_h3_kernel): the 7 horizontal taps join the 4 input channels as ONE k-step of 32 (28 used): in NHWC with
// C = 4 the operand of output pixel (y, x), vertical tap a, is the 32 contiguous values x[y + a - 3][x - 3 .. x + 4][0..3], so a
// lane's 8 k-values are 16 contiguous bytes of the staged plane.  21 MFMAs per 16 pixels x 16 channels instead of 196 fp32
// ones.  The input is the RAW pressure / mask / offset field (any magnitude): each workgroup scales its staged window by
// the power of two that puts the window's largest magnitude in [2^13, 2^14) and multiplies the product by the inverse --
// exact, and nothing overflows fp16 whatever the input's range.
//
// Input gradient (fu_stem_bwd_h3_kernel), the formulation of fu_stem_bwd_kernel
//     T[y][x'][b*4 + ci] = sum_a sum_co g[y + a - 3][x'][co] W[co][ci][6 - a][6 - b] ;  dx[y][x][ci] = sum_b T[y][x + b - 3][b*4 + ci]
// turned into a SCATTER over the staged gradient rows: a workgroup walks down a 16-row strip one gradient row at a time,
// multiplies the row by all 7 vertical taps (84 MFMAs per 16 pixels) and adds tap a's product into the accumulator of output
// row r + 3 - a; seven output rows are in flight in registers (rotating, the row loop is unrolled by 7), the weights never leave
// the wave.  Only ONE gradient row is staged at a time, so its power-of-two scale is the row's own (exact; depends on nothing
// outside the image: a design's gradient stays independent of its batch) and no pass over the tensor for a maximum is
// needed.  HBM: g is read 22 / 16 times instead of 10 / 4.
#pragma once
#include "forceunet_la.h"
#include <type_traits>

namespace cindm {

__device__ __forceinline__ float wave_max64(float v) {
    v = fmaxf(v, dpp_get<0x128>(v)); v = fmaxf(v, dpp_get<0x124>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0xB1>(v));
    return xmax32(xmax16(v));
}

struct FuStemArgs { const float* x; const float* W; const float* bias; float* y; int H, NI; };

// W: [tap a (7)][channel tile (4)][plane (2)][lane (64)][8 halfs]; lane (lr = channel in tile, lq), half j: k = lq*8 + j =
// (dx = 2 lq + (j >> 2), ci = j & 3) -> W[co][ci][a][dx], zero at dx = 7
template <int WD>
__global__ __launch_bounds__(256) void fu_stem_h3_kernel(const FuStemArgs a) {
    constexpr int RO = 8, RI = RO + 6, PP = WD + 8, NL = (RI * WD + 255) / 256, XB = WD / 16, NB = RO * XB;
    __shared__ __attribute__((aligned(16))) unsigned char Xp[2][RI * PP * 8];        // [plane][row][3 + WD + 5 pixels][4 halfs]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int strips = a.H / RO, img = blockIdx.x / strips, y0 = (blockIdx.x - img * strips) * RO;
    const float4* W4 = reinterpret_cast<const float4*>(a.W);
    half8 wh[7], wl[7];
#pragma unroll
    for (int ta = 0; ta < 7; ++ta) {
        wh[ta] = __builtin_bit_cast(half8, W4[((ta * 4 + w) * 2 + 0) * 64 + lane]);
        wl[ta] = __builtin_bit_cast(half8, W4[((ta * 4 + w) * 2 + 1) * 64 + lane]);
    }
    const float4 bias = *reinterpret_cast<const float4*>(a.bias + w * 16 + lq * 4);
    const float* xi = a.x + (size_t)img * a.H * WD * 4;
    float4 v[NL];
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + 256 * j, r = i / WD, x = i - r * WD, yy = y0 - 3 + r;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < RI * WD && yy >= 0 && yy < a.H) v[j] = *reinterpret_cast<const float4*>(xi + ((size_t)yy * WD + x) * 4);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
    }
    for (int i = tid; i < RI * 8 * 2; i += 256) {                  // the zero columns: staged pixels 0..2 and WD + 3 .. WD + 7
        const int pl = i & 1, hp = (i >> 1) & 7, r = i >> 4, px = hp < 3 ? hp : WD + hp;
        *reinterpret_cast<uint2*>(&Xp[pl][(r * PP + px) * 8]) = make_uint2(0u, 0u);
    }
    mx = wave_max64(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float inv;
    const float sc = grad_scale(mx, inv);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + 256 * j, r = i / WD, x = i - r * WD;
        if (i < RI * WD) {
            const float s0 = v[j].x * sc, s1 = v[j].y * sc, s2 = v[j].z * sc, s3 = v[j].w * sc;
            half4v hi, lo;
            hi[0] = (_Float16)s0; hi[1] = (_Float16)s1; hi[2] = (_Float16)s2; hi[3] = (_Float16)s3;
            lo[0] = (_Float16)((s0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((s1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((s2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((s3 - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&Xp[0][(r * PP + x + 3) * 8]) = hi;
            *reinterpret_cast<half4v*>(&Xp[1][(r * PP + x + 3) * 8]) = lo;
        }
    }
    __syncthreads();
    float* yo = a.y + ((size_t)img * a.H + y0) * WD * 64 + w * 16 + lq * 4;
#pragma unroll 2
    for (int pb = 0; pb < NB; ++pb) {
        const int py = pb / XB, px0 = (pb - py * XB) * 16;
        f32x4 M = f32x4{0.f, 0.f, 0.f, 0.f}, L = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ta = 0; ta < 7; ++ta) {
            const int off = ((py + ta) * PP + px0 + lr + 2 * lq) * 8;
            const half4v h0 = *reinterpret_cast<const half4v*>(&Xp[0][off]), h1 = *reinterpret_cast<const half4v*>(&Xp[0][off + 8]);
            const half4v l0 = *reinterpret_cast<const half4v*>(&Xp[1][off]), l1 = *reinterpret_cast<const half4v*>(&Xp[1][off + 8]);
            const half8 xh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7), xl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            M = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta], xh, M, 0, 0, 0);
            L = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta], xl, L, 0, 0, 0);
            L = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ta], xh, L, 0, 0, 0);
        }
        const f32x4 o = (M + L * H3_INV) * inv;
        *reinterpret_cast<float4*>(yo + (size_t)(py * WD + px0 + lr) * 64) = make_float4(o[0] + bias.x, o[1] + bias.y, o[2] + bias.z, o[3] + bias.w);
    }
}

struct FuStemBwdH3Args {
    const float* g; const float* W; float* dx; int H, NI; float beta;
#ifdef FU_STEM_PROF
    unsigned long long* prof;           // tools/micro/stem_bwd.hip only
#endif
};
#ifdef FU_STEM_PROF
#define FU_STEM_MARK(i) do { if (blockIdx.x == 5 && threadIdx.x == 0 && rr == 9) a.prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FU_STEM_MARK(i) do { } while (0)
#endif

// exponent field of the power of two that puts a magnitude mx into [2^13, 2^14) (grad_scale's rule)
__device__ __forceinline__ int scale_exp_of(float mx) {
    const int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
    return e == 0 ? 127 : min(max(267 - e, 1), 253);
}
__device__ __forceinline__ float pow2_of_exp(int se) { return __builtin_bit_cast(float, (unsigned)se << 23); }

// W: [tap a (7)][kk (2: channels 0..31 | 32..63)][nb (2)][plane (2)][lane (64)][8 halfs]; lane (lr = n in tile, lq), half j:
// A[n = nb*16 + lr][co = kk*32 + lq*8 + j] = W[co][ci = n & 3][6 - a][6 - (n >> 2)], zero for n >= 28.  64-pixel-wide images.
// Wave w = (column block nb = w & 1 of T, pixel blocks 2 (w >> 1), 2 (w >> 1) + 1): its 112 weight registers stay in VGPRs.  Each
// gradient row is staged with ITS OWN power-of-two scale; a tap's product (6 MFMAs from a zero accumulator) is multiplied by
// the row's inverse scale and added to the fp32 accumulator of the output row it feeds (8 VALU per 6 MFMAs).  A variant that
// accumulated in the MFMAs' own accumulators (one scale per 7-row window, accumulators rescaled when it moved) was no faster
// and lost bits of a small row that follows a much larger one within six rows (`tools/micro/stem_bwd.hip --check`).
__global__ __launch_bounds__(256) void fu_stem_bwd_h3_kernel(const FuStemBwdH3Args a) {
    constexpr int WD = 64, RO = 16, NR = RO + 6, PB = 144, TP = 33;
    __shared__ __attribute__((aligned(16))) unsigned char Gp[2][2][WD * PB];       // [buffer][plane][pixel][64 halfs + pad]
    __shared__ float Ts[2][WD * TP];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int nb = w & 1, pb0 = (w >> 1) * 2;
    const int strips = a.H / RO, img = blockIdx.x / strips, y0 = (blockIdx.x - img * strips) * RO;
    const float* gi = a.g + (size_t)img * a.H * WD * 64;
    const float4* W4 = reinterpret_cast<const float4*>(a.W);
    half8 wh[7][2], wl[7][2];
#pragma unroll
    for (int ta = 0; ta < 7; ++ta)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            wh[ta][kk] = __builtin_bit_cast(half8, W4[((((ta * 2 + kk) * 2 + nb) * 2 + 0) * 64) + lane]);
            wl[ta][kk] = __builtin_bit_cast(half8, W4[((((ta * 2 + kk) * 2 + nb) * 2 + 1) * 64) + lane]);
        }
    f32x4 acc[7][2];                                         // the seven output rows in flight x this wave's two pixel blocks
#pragma unroll
    for (int s = 0; s < 7; ++s) { acc[s][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[s][1] = acc[s][0]; }
    // staging role of a thread: pixels p0 + 16 j (j < 4), channel quad c4.  Rows outside the image are staged as zeros, so
    // that every row of the strip runs the same instruction stream.
    const int p0 = tid >> 4, c4 = tid & 15;
    float4 v[4];
    auto load_row = [&](int rr) {
        const int yy = y0 - 3 + rr;
        if (yy >= 0 && yy < a.H) {
            const float* gp = gi + (size_t)yy * WD * 64 + c4 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(gp + (size_t)(p0 + 16 * j) * 64);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto row_max = [&](int q) {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
        mx = wave_max64(mx);
        if (lane == 0) red[q][w] = mx;
    };
    auto red_exp = [&](int q) {
        const float mx = fmaxf(fmaxf(red[q][0], red[q][1]), fmaxf(red[q][2], red[q][3]));
        return scale_exp_of(mx);
    };
    auto store_row = [&](int q, float sc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s0 = v[j].x * sc, s1 = v[j].y * sc, s2 = v[j].z * sc, s3 = v[j].w * sc;
            half4v hi, lo;
            hi[0] = (_Float16)s0; hi[1] = (_Float16)s1; hi[2] = (_Float16)s2; hi[3] = (_Float16)s3;
            lo[0] = (_Float16)((s0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((s1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((s2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((s3 - (float)hi[3]) * H3_SCALE);
            const int off = (p0 + 16 * j) * PB + c4 * 8;
            *reinterpret_cast<half4v*>(&Gp[q][0][off]) = hi;
            *reinterpret_cast<half4v*>(&Gp[q][1][off]) = lo;
        }
    };
    const int ox = tid >> 2, oc = tid & 3;                  // output role: pixel ox, channel oc
    auto row_sum = [&](int q, int oy) {                      // dx row oy of the strip from the T row in Ts[q]
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const int xg = ox + b - 3;
            if (xg >= 0 && xg < WD) s += Ts[q][xg * TP + b * 4 + oc];
        }
        if (oy >= 0) {
            float* o = a.dx + (((size_t)img * a.H + y0 + oy) * WD + ox) * 4 + oc;
            *o = a.beta != 0.f ? a.beta * (*o) + s : s;
        }
    };
    // prologue: row 0 staged, row 1 in registers with its maximum exchanged
    load_row(0);
    row_max(0);
    __syncthreads();
    int se = red_exp(0);
    float inv_cur = pow2_of_exp(254 - se);                   // inverse scale of the row the planes hold
    store_row(0, pow2_of_exp(se));
    load_row(1);
    row_max(1);
    __syncthreads();
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    // One barrier per gradient row.  In iteration rr a wave (A) stages row rr + 1 from its registers (its maximum crossed the
    // last barrier), (B) requests row rr + 2, (C) writes out the dx row whose T row was parked in LDS one iteration ago, (D)
    // multiplies row rr, (E) reduces the maximum of row rr + 2, (F) parks the T row that just received its last tap.  (A) - (C),
    // (E) are independent of (D): they fill the matrix pipe's shadow.
#pragma unroll 1
    for (int it = 0; it < (NR + 6) / 7; ++it) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int rr = it * 7 + j, buf = rr & 1;
            if (rr < NR) {                                   // (uniform)
                FU_STEM_MARK(0);
                // (A) row rr + 1 with the scale of its own maximum
                se = red_exp(buf ^ 1);
                const float inv_next = pow2_of_exp(254 - se);
                store_row(buf ^ 1, pow2_of_exp(se));
                load_row(rr + 2);                            // (B)
                FU_STEM_MARK(1);
                row_sum(buf ^ 1, rr - 7);                    // (C)
                FU_STEM_MARK(2);
                // (D) tap a of row rr feeds output row rr - a, whose accumulator is slot (rr - a) mod 7; tap 0 OPENS a slot.  Slots of
                // rows outside the strip are fed like the others and never written out.
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int boff = ((pb0 + p) * 16 + lr) * PB + lq * 16;
                    const half8 gh0 = *reinterpret_cast<const half8*>(&Gp[buf][0][boff]), gh1 = *reinterpret_cast<const half8*>(&Gp[buf][0][boff + 64]);
                    const half8 gl0 = *reinterpret_cast<const half8*>(&Gp[buf][1][boff]), gl1 = *reinterpret_cast<const half8*>(&Gp[buf][1][boff + 64]);
#pragma unroll
                    for (int ta = 0; ta < 7; ++ta) {
                        const int slot = (j - ta + 7) % 7;
                        f32x4 M = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta][0], gh0, zero4, 0, 0, 0);
                        f32x4 Lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta][0], gl0, zero4, 0, 0, 0);
                        Lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ta][0], gh0, Lo, 0, 0, 0);
                        M = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta][1], gh1, M, 0, 0, 0);
                        Lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ta][1], gl1, Lo, 0, 0, 0);
                        Lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ta][1], gh1, Lo, 0, 0, 0);
                        const f32x4 pr = (M + Lo * H3_INV) * inv_cur;
                        acc[slot][p] = ta == 0 ? pr : acc[slot][p] + pr;
                    }
                }
                FU_STEM_MARK(3);
                row_max(buf);                                // (E) row rr + 2
                // (F) output row rr - 6 has its last contribution (tap 6 of this row)
                {
                    const int sd = (j + 1) % 7;
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int i = 0; i < 4; ++i) Ts[buf][((pb0 + p) * 16 + lr) * TP + nb * 16 + lq * 4 + i] = acc[sd][p][i];
                }
                inv_cur = inv_next;
                FU_STEM_MARK(4);
                __syncthreads();
                FU_STEM_MARK(5);
            }
        }
    }
    row_sum((NR - 1) & 1, NR - 7);
}

}  // namespace cindm
