// ForceUnet's Residual(PreNorm(LinearAttention)) sites (model/diffusion_2d.py:226-254) WITHOUT the [pixels, 384] q | k | v
// tensors -- forward on the diffusion U-Net's la2d_* kernels (kernels2d.h: x is read twice, the output written once), and
// an input-gradient pass in the same style.  Round 2's path materialised qkv, softmax_d(q) s, softmax_n(k), the attention
// output, their gradients and two LayerNorm tensors per site: 90 GB of the 176 GB a design-gradient call moved, 20 of its
// 47 ms (profiles/r02_pmc_traffic_force.json).  The backward here recomputes everything per 64-pixel tile from x:
//
//   pass A (fu_la_bwd_a_kernel), wave = head:  y = LN(x) g1 -> q -> qs = softmax_d(q) s -> att = ctx^T qs -> z = Wo att + b
//     -> dz = LayerNorm'(z; g2, dout) -> datt = Wo^T dz -> dqs = ctx datt -> dq (softmax derivative)
//     -> dctx += qs datt^T (per-workgroup partial, merged afterwards)  and  dyq = Wq^T dq  [pixels, C] (written)
//   merge (fu_la_dctx_merge_kernel): dctx = sum of partials, T[d] = sum_e dctx[d][e] ctx[d][e]  (= sum_n ks dks, no pixel pass)
//   pass B (fu_la_bwd_b_kernel), wave = head:  y again -> k, v -> ks = exp(k - max_n) / sum_n -> dks = dctx v / n,
//     dv = dctx^T ks / n, dk = ks (dks - T) -> dy = dyq + Wk^T dk + Wv^T dv -> dx = dout + LayerNorm'(x; g1, dy) (+ what x.g holds)
//
// HBM traffic of a site's backward: x twice, dout twice, dyq written and read, dx written -- 7 tensors of [pixels, C] instead
// of ~40.  Products of FORWARD quantities (q, k, v, att, z) run on the split-fp16 MFMA like the forward kernels.  The three
// large products with a GRADIENT operand -- datt = Wo^T dz, dyq = Wq^T dq, dy = [Wk^T | Wv^T] [dk ; dv] -- do too: the
// gradient matrix is staged as hi / scaled-lo planes times a power of two chosen PER PIXEL from that pixel's largest
// magnitude (gradients sit far below fp16's range; the factor and its inverse are exact), 3 fp16 MFMAs of K = 32 instead of
// 8 fp32 MFMAs of K = 4.  The small per-head products (32 x 32 contexts) stay on the exact fp32 MFMA.
// Layout chains (no transposes through LDS except where a contraction runs over pixels):
//   rows = channels, cols = pixels accumulators (A = weight fragment, B = pixel planes) are the B operand of the next
//   channel contraction; contractions over PIXELS (dctx) read both operands back from a per-wave fp32 LDS tile.
#pragma once
#include "kernels2d.h"

namespace cindm {

struct FuLaArgs {
    const float* x; int ldx;            // site input [NI * HW, C]
    const float* dout;                  // gradient with respect to the site output [NI * HW, C]
    const float* g1; const float* g2;   // PreNorm gain, to_out LayerNorm gain
    const float* Wqkv; const float* Wo; const float* bo;      // la2d split-fp16 fragments (24 tiles x C ; C/16 tiles x 128), bias [C]
    const float* WoT; const float* WqT; const float* WkvT;     // split-fp16 fragments of Wo^T [128 x C], Wq^T [C x 128], [Wk^T | Wv^T] [C x 256]
    const float* ctx;                   // [NI * 4][32][32] merged context (incl. 1 / n)
    const float* kst;                   // [NI * 4][32][2]: max_n k, 1 / sum_n exp(k - max)
    float* dctx_part;                   // [NI][wpi][4][1024] per-workgroup partials of pass A
    const float* dctx; const float* T;  // merged: [NI * 4][1024], [NI * 128]
    float* dyq;                         // [NI * HW, C]: Wq^T dq, written by pass A, read by pass B
    float* dx; float beta;              // pass B: dx = beta * dx + dout + LayerNorm'(...)
    int HW, tpw;                        // pixels per image ; NPX-pixel tiles per workgroup (NPX = 64 at C = 64, 32 at C = 128)
    float inv_n;
#ifdef FU_LA_PROF
    unsigned long long* prof;            // tools/micro/la_bwd.hip only
#endif
};
#ifdef FU_LA_PROF
#define FU_LA_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tt == 1) a.prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FU_LA_MARK(i) do { } while (0)
#endif

// max over groups of LPR consecutive lanes (as rowgroup_sum)
template <int LPR>
__device__ __forceinline__ float rowgroup_max(float v) {
    v = fmaxf(v, dpp_get<0x128>(v)); v = fmaxf(v, dpp_get<0x124>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0xB1>(v));
    if (LPR >= 32) v = xmax16(v);
    if (LPR >= 64) v = xmax32(v);
    return v;
}
// the power of two that puts a maximum magnitude mx into [2^13, 2^14) (fp16's upper normal range), and its inverse: gradient
// operands are staged as split-fp16 planes times this factor (exact), the product is multiplied by the inverse (exact)
__device__ __forceinline__ float grad_scale(float mx, float& inv) {
    const int e = (int)(__builtin_bit_cast(unsigned, mx) >> 23);
    const int se = e == 0 ? 127 : min(max(267 - e, 1), 253);
    inv = __builtin_bit_cast(float, (unsigned)(254 - se) << 23);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}

// merged context + the column statistics of k that pass B needs (la2d_merge_kernel + kst): one workgroup per (image, head)
__global__ __launch_bounds__(256) void fu_la_merge_kernel(const float* __restrict__ part, float* __restrict__ ctx, float* __restrict__ kst,
                                                          int nsplit, int HW) {
    __shared__ float wgt[64 * 32];
    const int ih = blockIdx.x;
    const float* p = part + (size_t)ih * nsplit * LA2_REC;
    if (threadIdx.x < 32) {
        const int d = threadIdx.x;
        float M = p[d];
        for (int sp = 1; sp < nsplit; ++sp) M = fmaxf(M, p[(size_t)sp * LA2_REC + d]);
        float S = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float e = __builtin_amdgcn_exp2f((p[(size_t)sp * LA2_REC + d] - M) * 1.4426950408889634f);
            wgt[sp * 32 + d] = e;
            S += p[(size_t)sp * LA2_REC + 32 + d] * e;
        }
        const float inv = 1.0f / (S * (float)HW);
        for (int sp = 0; sp < nsplit; ++sp) wgt[sp * 32 + d] *= inv;
        if (kst) { kst[((size_t)ih * 32 + d) * 2] = M; kst[((size_t)ih * 32 + d) * 2 + 1] = 1.0f / S; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + 256 * j;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += p[(size_t)sp * LA2_REC + 64 + i] * wgt[sp * 32 + (i >> 5)];
        ctx[(size_t)ih * 1024 + i] = s;
    }
}

// dctx[ih] = sum over the image's pass-A workgroups ; T[ih][d] = sum_e dctx[d][e] ctx[d][e].  One workgroup per (image, head).
__global__ __launch_bounds__(256) void fu_la_dctx_merge_kernel(const float* __restrict__ part, const float* __restrict__ ctx,
                                                               float* __restrict__ dctx, float* __restrict__ T, int wpi) {
    __shared__ float prod[1024];
    const int ih = blockIdx.x, img = ih >> 2, hd = ih & 3, tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        float s = 0.f;
        for (int wg = 0; wg < wpi; ++wg) s += part[(((size_t)img * wpi + wg) * 4 + hd) * 1024 + i];       // fixed order: repeatable
        dctx[(size_t)ih * 1024 + i] = s;
        prod[i] = s * ctx[(size_t)ih * 1024 + i];
    }
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
        for (int e = 0; e < 32; ++e) t += prod[tid * 32 + e];
        T[(size_t)ih * 32 + tid] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int C, int NPX, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void fu_la_bwd_a_kernel(const FuLaArgs a) {
    using LN = LnTile<C, NPX>;
    constexpr int K32 = C / 32, YPB = 2 * C + 16, APB = 2 * 128 + 16, NTL = NPX / 16, CT = C / 16, TPW = CT / 4, ZP = C + 4, KC4 = C / 4;
    constexpr int QP = 33;                                   // pitch of the per-wave [pixel][32] fp32 tiles
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][NPX * YPB];       // y planes -> z -> dz (fp32 [pixel][ZP])
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][NPX * APB];       // att planes -> dq (fp32 [128][NPX])
    __shared__ float QS[4][NPX * QP], DA[4][NPX * QP];                            // per wave: qs and datt with pixels as rows
    __shared__ float SCZ[NPX], PM[4][NPX];                                        // 1 / scale of a pixel's dz ; per-head max |dq| of a pixel
    static_assert(NPX * ZP * 4 <= 2 * NPX * YPB, "Z aliases the y planes");
    static_assert(2 * NPX * YPB <= 2 * NPX * APB, "the dz planes alias the att planes");
    float* Z = reinterpret_cast<float*>(&Yp[0][0]);
    unsigned char* DZ0 = &Ap[0][0];                          // dz planes [pixel][C] (pitch YPB), then the dq planes [pixel][128] (pitch APB)
    unsigned char* DZ1 = DZ0 + NPX * YPB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int wpi = a.HW / (NPX * a.tpw);
    const int img = blockIdx.x / wpi, wg = blockIdx.x % wpi, t0 = wg * a.tpw;
    const size_t row00 = (size_t)img * a.HW + (size_t)t0 * NPX;
    float4 xr[LN::NPASS];
    LN::load(xr, a.x + row00 * a.ldx, a.ldx, tid);
    const int lcol = tid % LN::LPR, lrow = tid / LN::LPR;
    const float4 gv = *reinterpret_cast<const float4*>(a.g1 + 4 * lcol);
    const float4 gv2 = *reinterpret_cast<const float4*>(a.g2 + 4 * lcol);
    const float4* W4 = reinterpret_cast<const float4*>(a.Wqkv);
    const float4* Wo4 = reinterpret_cast<const float4*>(a.Wo);
    half8 qh[2][K32], ql[2][K32];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            qh[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(2 * w + s) * K32 + k) * 2 + 0) * 64 + lane]);
            ql[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(2 * w + s) * K32 + k) * 2 + 1) * 64 + lane]);
        }
    half8 oh[TPW][4], ol[TPW][4];
    float4 bias[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        bias[s] = *reinterpret_cast<const float4*>(a.bo + (w * TPW + s) * 16 + lq * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            oh[s][k] = __builtin_bit_cast(half8, Wo4[(((size_t)(w * TPW + s) * 4 + k) * 2 + 0) * 64 + lane]);
            ol[s][k] = __builtin_bit_cast(half8, Wo4[(((size_t)(w * TPW + s) * 4 + k) * 2 + 1) * 64 + lane]);
        }
    }
    // context of head w: cf = A fragments of att = ctx^T qs (rows e, k = d) ; cA = A fragments of dqs = ctx datt (rows d, k = e)
    float cf[2][2][4], cA[2][2][4];
    {
        const float* cp = a.ctx + (size_t)(img * 4 + w) * 1024;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cf[dt][et][i] = cp[(dt * 16 + lq * 4 + i) * 32 + et * 16 + lr];
                    cA[dt][et][i] = cp[(dt * 16 + lr) * 32 + et * 16 + lq * 4 + i];
                }
    }
    // Wo^T rows of head w (datt = Wo^T dz): tiles 2w, 2w + 1 of the [128 x C] fragments ; Wq^T rows of this wave's channel tiles
    // (dyq = Wq^T dq over the 128 head channels): tiles w TPW + s of the [C x 128] fragments
    const float4* WoT4 = reinterpret_cast<const float4*>(a.WoT);
    const float4* WqT4 = reinterpret_cast<const float4*>(a.WqT);
    half8 woh[2][K32], wol[2][K32], wqh[TPW][4], wql[TPW][4];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            woh[et][k] = __builtin_bit_cast(half8, WoT4[(((size_t)(2 * w + et) * K32 + k) * 2 + 0) * 64 + lane]);
            wol[et][k] = __builtin_bit_cast(half8, WoT4[(((size_t)(2 * w + et) * K32 + k) * 2 + 1) * 64 + lane]);
        }
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wqh[s][k] = __builtin_bit_cast(half8, WqT4[(((size_t)(w * TPW + s) * 4 + k) * 2 + 0) * 64 + lane]);
            wql[s][k] = __builtin_bit_cast(half8, WqT4[(((size_t)(w * TPW + s) * 4 + k) * 2 + 1) * 64 + lane]);
        }
    f32x4 dctx[2][2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int et = 0; et < 2; ++et) dctx[dt][et] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int tt = 0; tt < a.tpw; ++tt) {
        const size_t row0 = row00 + (size_t)tt * NPX;
        FU_LA_MARK(0);
        LN::to_planes(xr, gv, Yp[0], Yp[1], tid);
        __syncthreads();                                                            // (1) y planes
        FU_LA_MARK(1);
        if (tt + 1 < a.tpw) LN::load(xr, a.x + (row0 + NPX) * a.ldx, a.ldx, tid);
        // dout rows of this thread (LayerNorm' below), requested early
        float4 dor[LN::NPASS];
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) dor[r] = *reinterpret_cast<const float4*>(a.dout + (row0 + r * LN::RPP + lrow) * C + 4 * lcol);
        f32x4 qs[NTL][2];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f32x4 M[2], Lo[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { M[s] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int k = 0; k < K32; ++k) {
                const int off = (nt * 16 + lr) * YPB + k * 64 + lq * 16;
                const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
                const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    M[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s][k], yh, M[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s][k], yl, Lo[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[s][k], yh, Lo[s], 0, 0, 0);
                }
            }
            f32x4 q[2];
            q[0] = M[0] + Lo[0] * H3_INV; q[1] = M[1] + Lo[1] * H3_INV;
            float mx = -INFINITY;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, q[dt][i]);
            mx = xmax32(xmax16(mx));
            float sum = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __builtin_amdgcn_exp2f((q[dt][i] - mx) * 1.4426950408889634f);
                    q[dt][i] = e;
                    sum += e;
                }
            sum = xsum32(xsum16(sum));
            const float inv = 1.0f / sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) { q[dt] = (q[dt] * inv) * 0.17677669529663687f; qs[nt][dt] = q[dt]; }
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[dt][et][i], q[dt][i], o, 0, 0, 0);
                half4v hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)o[i]; lo[i] = (_Float16)((o[i] - (float)hi[i]) * H3_SCALE); }
                const int off = (nt * 16 + lr) * APB + 2 * (w * 32 + et * 16 + lq * 4);
                *reinterpret_cast<half4v*>(&Ap[0][off]) = hi;
                *reinterpret_cast<half4v*>(&Ap[1][off]) = lo;
            }
            // qs with pixels as rows for the pixel contraction below (this wave's private tile)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) QS[w][(nt * 16 + lr) * QP + dt * 16 + lq * 4 + i] = q[dt][i];
        }
        __syncthreads();                                                            // (2) att planes ; y planes consumed
        FU_LA_MARK(2);
        // z = Wo att + bo -> Z[pixel][channel]
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int c = (w * TPW + s) * 16 + lq * 4;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                f32x4 zM = f32x4{0.f, 0.f, 0.f, 0.f}, zL = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int off = (nt * 16 + lr) * APB + k * 64 + lq * 16;
                    const half8 ah = *reinterpret_cast<const half8*>(&Ap[0][off]);
                    const half8 al = *reinterpret_cast<const half8*>(&Ap[1][off]);
                    zM = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s][k], ah, zM, 0, 0, 0);
                    zL = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s][k], al, zL, 0, 0, 0);
                    zL = __builtin_amdgcn_mfma_f32_16x16x32_f16(ol[s][k], ah, zL, 0, 0, 0);
                }
                const f32x4 z = zM + zL * H3_INV;
                *reinterpret_cast<float4*>(&Z[(nt * 16 + lr) * ZP + c]) = make_float4(z[0] + bias[s].x, z[1] + bias[s].y, z[2] + bias[s].z, z[3] + bias[s].w);
            }
        }
        __syncthreads();                                                            // (3) Z ; att planes consumed
        FU_LA_MARK(3);
        // dz = LayerNorm'(z; g2) dout, in place: dz = r (t - mean(t) - zhat mean(t zhat)), t = g2 dout
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) {
            const int n = r * LN::RPP + lrow;
            const float4 zv = *reinterpret_cast<const float4*>(&Z[n * ZP + 4 * lcol]);
            const float mean = rowgroup_sum<LN::LPR>((zv.x + zv.y) + (zv.z + zv.w)) * (1.0f / C);
            const float d0 = zv.x - mean, d1 = zv.y - mean, d2 = zv.z - mean, d3 = zv.w - mean;
            const float rstd = 1.0f / sqrtf(rowgroup_sum<LN::LPR>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / C) + 1e-5f);
            const float z0 = d0 * rstd, z1 = d1 * rstd, z2 = d2 * rstd, z3 = d3 * rstd;
            const float t0 = gv2.x * dor[r].x, t1 = gv2.y * dor[r].y, t2 = gv2.z * dor[r].z, t3 = gv2.w * dor[r].w;
            const float m1 = rowgroup_sum<LN::LPR>((t0 + t1) + (t2 + t3)) * (1.0f / C);
            const float m2 = rowgroup_sum<LN::LPR>((t0 * z0 + t1 * z1) + (t2 * z2 + t3 * z3)) * (1.0f / C);
            const float g0 = rstd * (t0 - m1 - z0 * m2), g1 = rstd * (t1 - m1 - z1 * m2), g2 = rstd * (t2 - m1 - z2 * m2), g3 = rstd * (t3 - m1 - z3 * m2);
            // dz of this pixel as split-fp16 planes times the power of two that puts the pixel's largest |dz| in [2^13, 2^14)
            float inv;
            const float sc = grad_scale(rowgroup_max<LN::LPR>(fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fmaxf(fabsf(g2), fabsf(g3)))), inv);
            if (lcol == 0) SCZ[n] = inv;
            const float s0 = g0 * sc, s1 = g1 * sc, s2 = g2 * sc, s3 = g3 * sc;
            half4v hi, lo;
            hi[0] = (_Float16)s0; hi[1] = (_Float16)s1; hi[2] = (_Float16)s2; hi[3] = (_Float16)s3;
            lo[0] = (_Float16)((s0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((s1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((s2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((s3 - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(DZ0 + n * YPB + 8 * lcol) = hi;
            *reinterpret_cast<half4v*>(DZ1 + n * YPB + 8 * lcol) = lo;
        }
        __syncthreads();                                                            // (4) dz planes
        FU_LA_MARK(4);
        // datt (rows e of head w, cols pixels) = Wo^T dz ; dqs = ctx datt ; dq ; dctx += qs datt^T
        f32x4 dq[NTL][2];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f32x4 da[2];
            {
                f32x4 M[2], Lo[2];
#pragma unroll
                for (int et = 0; et < 2; ++et) { M[et] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[et] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int k = 0; k < K32; ++k) {
                    const int off = (nt * 16 + lr) * YPB + k * 64 + lq * 16;
                    const half8 gh = *reinterpret_cast<const half8*>(DZ0 + off);
                    const half8 gl = *reinterpret_cast<const half8*>(DZ1 + off);
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
                        M[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(woh[et][k], gh, M[et], 0, 0, 0);
                        Lo[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(woh[et][k], gl, Lo[et], 0, 0, 0);
                        Lo[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wol[et][k], gh, Lo[et], 0, 0, 0);
                    }
                }
                const float inv = SCZ[nt * 16 + lr];
                da[0] = (M[0] + Lo[0] * H3_INV) * inv; da[1] = (M[1] + Lo[1] * H3_INV) * inv;
            }
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int i = 0; i < 4; ++i) DA[w][(nt * 16 + lr) * QP + et * 16 + lq * 4 + i] = da[et][i];
            f32x4 ds[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int et = 0; et < 2; ++et)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ds[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[dt][et][i], da[et][i], ds[dt], 0, 0, 0);
            // dq_d = qs_d (dqs_d - sum_j p_j dqs_j), p = qs / scale
            float dot = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) dot += qs[nt][dt][i] * ds[dt][i];
            dot = xsum32(xsum16(dot)) * 5.656854249492381f;       // 1 / scale
            float mxq = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) { dq[nt][dt][i] = qs[nt][dt][i] * (ds[dt][i] - dot); mxq = fmaxf(mxq, fabsf(dq[nt][dt][i])); }
            mxq = xmax32(xmax16(mxq));                       // over this head's 32 channels of pixel nt * 16 + lr
            if (lq == 0) PM[w][nt * 16 + lr] = mxq;
        }
        __syncthreads();                                                            // (5) dz planes consumed ; QS / DA / PM complete
        FU_LA_MARK(5);
        // dq of every head -> planes [pixel][128] (pitch APB, over the dz planes) times the pixel's power of two (common to the heads)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int px = nt * 16 + lr;
            float inv;
            const float sc = grad_scale(fmaxf(fmaxf(PM[0][px], PM[1][px]), fmaxf(PM[2][px], PM[3][px])), inv);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                half4v hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float v = dq[nt][dt][i] * sc; hi[i] = (_Float16)v; lo[i] = (_Float16)((v - (float)hi[i]) * H3_SCALE); }
                const int off = px * APB + 2 * (w * 32 + dt * 16 + lq * 4);
                *reinterpret_cast<half4v*>(&Ap[0][off]) = hi;
                *reinterpret_cast<half4v*>(&Ap[1][off]) = lo;
            }
        }
        // dctx[d][e] += sum_pixels qs[d][n] datt[e][n]: A[i = d][k = pixel], B[k = pixel][j = e] from the wave's own tiles
#pragma unroll
        for (int kk = 0; kk < NPX / 4; ++kk) {
            const int pr = (4 * kk + lq) * QP + lr;
            const float a0 = QS[w][pr], a1 = QS[w][pr + 16], b0 = DA[w][pr], b1 = DA[w][pr + 16];
            dctx[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, dctx[0][0], 0, 0, 0);
            dctx[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, dctx[0][1], 0, 0, 0);
            dctx[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, dctx[1][0], 0, 0, 0);
            dctx[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, dctx[1][1], 0, 0, 0);
        }
        __syncthreads();                                                            // (6) dq planes
        FU_LA_MARK(6);
        // dyq (rows c of this wave's channel tiles, cols pixels) = Wq^T dq over the 128 head channels
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int c = (w * TPW + s) * 16 + lq * 4;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                f32x4 oM = f32x4{0.f, 0.f, 0.f, 0.f}, oL = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int off = (nt * 16 + lr) * APB + k * 64 + lq * 16;
                    const half8 gh = *reinterpret_cast<const half8*>(&Ap[0][off]);
                    const half8 gl = *reinterpret_cast<const half8*>(&Ap[1][off]);
                    oM = __builtin_amdgcn_mfma_f32_16x16x32_f16(wqh[s][k], gh, oM, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wqh[s][k], gl, oL, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wql[s][k], gh, oL, 0, 0, 0);
                }
                const int px = nt * 16 + lr;
                float inv;
                (void)grad_scale(fmaxf(fmaxf(PM[0][px], PM[1][px]), fmaxf(PM[2][px], PM[3][px])), inv);
                const f32x4 o = (oM + oL * H3_INV) * inv;
                *reinterpret_cast<float4*>(a.dyq + (row0 + px) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();                                                            // (7) dq planes and PM consumed: the next tile rewrites Yp / Ap
        FU_LA_MARK(7);
    }
    float* rec = a.dctx_part + (((size_t)img * wpi + wg) * 4 + w) * 1024;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int i = 0; i < 4; ++i) rec[(dt * 16 + lq * 4 + i) * 32 + et * 16 + lr] = dctx[dt][et][i];
}

// ---------------------------------------------------------------------------------------------------------------------
template <int C, int NPX, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void fu_la_bwd_b_kernel(const FuLaArgs a) {
    using LN = LnTile<C, NPX>;
    constexpr int K32 = C / 32, YPB = 2 * C + 16, NTL = NPX / 16, CT = C / 16, TPW = CT / 4, ZP = C + 4;
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][NPX * YPB];       // y planes -> dy (fp32 [pixel][ZP])
    constexpr int GPB = 2 * 256 + 16;                        // bytes per pixel and plane of the [dk | dv] planes
    __shared__ __attribute__((aligned(16))) unsigned char Gp[2][NPX * GPB];       // dk (channels 0..127) | dv (128..255) of every head, scaled split-fp16
    __shared__ float PM[4][NPX];                                                  // per-head max(|dk|, |dv|) of a pixel
    static_assert(NPX * ZP * 4 <= 2 * NPX * YPB, "dy aliases the y planes");
    float* DY = reinterpret_cast<float*>(&Yp[0][0]);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int wpi = a.HW / (NPX * a.tpw);
    const int img = blockIdx.x / wpi, t0 = (blockIdx.x % wpi) * a.tpw;
    const size_t row00 = (size_t)img * a.HW + (size_t)t0 * NPX;
    float4 xr[LN::NPASS];
    LN::load(xr, a.x + row00 * a.ldx, a.ldx, tid);
    const int lcol = tid % LN::LPR, lrow = tid / LN::LPR;
    const float4 gv = *reinterpret_cast<const float4*>(a.g1 + 4 * lcol);
    const float4* W4 = reinterpret_cast<const float4*>(a.Wqkv);
    // k (tiles 8 + 2w, 9 + 2w) and v (16 + 2w, 17 + 2w) rows of head w as A operands
    half8 kh[2][K32], kl[2][K32], vh[2][K32], vl[2][K32];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            kh[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(8 + 2 * w + s) * K32 + k) * 2 + 0) * 64 + lane]);
            kl[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(8 + 2 * w + s) * K32 + k) * 2 + 1) * 64 + lane]);
            vh[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(16 + 2 * w + s) * K32 + k) * 2 + 0) * 64 + lane]);
            vl[s][k] = __builtin_bit_cast(half8, W4[(((size_t)(16 + 2 * w + s) * K32 + k) * 2 + 1) * 64 + lane]);
        }
    // dctx of head w: dA = A fragments of dks = dctx v (rows d, k = e) ; dT = A fragments of dv = dctx^T ks (rows e, k = d)
    float dA[2][2][4], dT[2][2][4], kmx[2][4], kis[2][4], Tt[2][4];
    {
        const float* dp = a.dctx + (size_t)(img * 4 + w) * 1024;
        const float* ks = a.kst + (size_t)(img * 4 + w) * 64;
        const float* tp = a.T + (size_t)(img * 4 + w) * 32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = dt * 16 + lq * 4 + i;
                kmx[dt][i] = ks[d * 2]; kis[dt][i] = ks[d * 2 + 1]; Tt[dt][i] = tp[d];
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    dA[dt][et][i] = dp[(dt * 16 + lr) * 32 + et * 16 + lq * 4 + i];
                    dT[et][dt][i] = dp[(dt * 16 + lq * 4 + i) * 32 + et * 16 + lr];
                }
            }
    }
    // [Wk^T | Wv^T] rows of this wave's channel tiles (dy = Wk^T dk + Wv^T dv): tiles w TPW + s of the [C x 256] fragments
    constexpr int UNR = C == 64 ? NTL : 1;                   // (C = 128: the pixel-block loops stay rolled, or the registers spill)
    const float4* WkvT4 = reinterpret_cast<const float4*>(a.WkvT);
    half8 wgh[TPW][8], wgl[TPW][8];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wgh[s][k] = __builtin_bit_cast(half8, WkvT4[(((size_t)(w * TPW + s) * 8 + k) * 2 + 0) * 64 + lane]);
            wgl[s][k] = __builtin_bit_cast(half8, WkvT4[(((size_t)(w * TPW + s) * 8 + k) * 2 + 1) * 64 + lane]);
        }

#pragma unroll 1
    for (int tt = 0; tt < a.tpw; ++tt) {
        const size_t row0 = row00 + (size_t)tt * NPX;
        FU_LA_MARK(32);
        constexpr bool EARLY_YQ = C == 64;                   // (C = 128: the dy loop stays rolled -- a register array could not be indexed)
        float4 xres[LN::NPASS];                              // the tile's own rows, for the LayerNorm derivative at the end
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) xres[r] = xr[r];
        LN::to_planes(xr, gv, Yp[0], Yp[1], tid);
        __syncthreads();                                                            // (1) y planes
        FU_LA_MARK(33);
        if (tt + 1 < a.tpw) LN::load(xr, a.x + (row0 + NPX) * a.ldx, a.ldx, tid);
        // this tile's dout rows and its dyq fragments are requested HERE, a whole k | v phase before
        // their first use -- requested where they are used, their HBM latency was most of the dy phase (9.9k of 28k cycles per tile)
        float4 dor[LN::NPASS], yqr[EARLY_YQ ? TPW * NTL : 1];
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) dor[r] = *reinterpret_cast<const float4*>(a.dout + (row0 + r * LN::RPP + lrow) * C + 4 * lcol);
        if constexpr (EARLY_YQ) {
#pragma unroll
            for (int s = 0; s < TPW; ++s)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) yqr[s * NTL + nt] = *reinterpret_cast<const float4*>(a.dyq + (row0 + nt * 16 + lr) * C + (w * TPW + s) * 16 + lq * 4);
        }
        f32x4 dkr[NTL][2], dvr[NTL][2];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f32x4 M[4], Lo[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { M[s] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int k = 0; k < K32; ++k) {
                const int off = (nt * 16 + lr) * YPB + k * 64 + lq * 16;
                const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
                const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    M[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[s][k], yh, M[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[s][k], yl, Lo[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[s][k], yh, Lo[s], 0, 0, 0);
                    M[2 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[s][k], yh, M[2 + s], 0, 0, 0);
                    Lo[2 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[s][k], yl, Lo[2 + s], 0, 0, 0);
                    Lo[2 + s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[s][k], yh, Lo[2 + s], 0, 0, 0);
                }
            }
            f32x4 ksm[2], vv[2];                             // rows d / e of head w, cols pixels
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4 kk = M[dt] + Lo[dt] * H3_INV;
                vv[dt] = M[2 + dt] + Lo[2 + dt] * H3_INV;
#pragma unroll
                for (int i = 0; i < 4; ++i) ksm[dt][i] = __builtin_amdgcn_exp2f((kk[i] - kmx[dt][i]) * 1.4426950408889634f) * kis[dt][i];
            }
            // dks[d][n] = sum_e dctx[d][e] v[e][n] / N ; dv[e][n] = sum_d dctx[d][e] ks[d][n] / N
            f32x4 dks[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int et = 0; et < 2; ++et)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dks[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dA[dt][et][i], vv[et][i], dks[dt], 0, 0, 0);
                        dv[et] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT[et][dt][i], ksm[dt][i], dv[et], 0, 0, 0);
                    }
            float mx = 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dkr[nt][dt][i] = ksm[dt][i] * (dks[dt][i] * a.inv_n - Tt[dt][i]);
                    dvr[nt][dt][i] = dv[dt][i] * a.inv_n;
                    mx = fmaxf(mx, fmaxf(fabsf(dkr[nt][dt][i]), fabsf(dvr[nt][dt][i])));
                }
            mx = xmax32(xmax16(mx));                         // over this head's 64 dk | dv channels of pixel nt * 16 + lr
            if (lq == 0) PM[w][nt * 16 + lr] = mx;
        }
        __syncthreads();                                                            // (2) PM ; y planes consumed
        FU_LA_MARK(34);
        // dk | dv of every head -> planes [pixel][256] times the pixel's power of two (common to the heads)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int px = nt * 16 + lr;
            float inv;
            const float sc = grad_scale(fmaxf(fmaxf(PM[0][px], PM[1][px]), fmaxf(PM[2][px], PM[3][px])), inv);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                half4v kh4, kl4, vh4, vl4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float vk = dkr[nt][dt][i] * sc, vv_ = dvr[nt][dt][i] * sc;
                    kh4[i] = (_Float16)vk; kl4[i] = (_Float16)((vk - (float)kh4[i]) * H3_SCALE);
                    vh4[i] = (_Float16)vv_; vl4[i] = (_Float16)((vv_ - (float)vh4[i]) * H3_SCALE);
                }
                const int off = px * GPB + 2 * (w * 32 + dt * 16 + lq * 4);
                *reinterpret_cast<half4v*>(&Gp[0][off]) = kh4; *reinterpret_cast<half4v*>(&Gp[1][off]) = kl4;
                *reinterpret_cast<half4v*>(&Gp[0][off + 256]) = vh4; *reinterpret_cast<half4v*>(&Gp[1][off + 256]) = vl4;
            }
        }
        __syncthreads();                                                            // (2b) dk | dv planes
        FU_LA_MARK(35);

        // dy (rows c of this wave's channel tiles, cols pixels) = dyq + Wk^T dk + Wv^T dv -> DY[pixel][channel]
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            const int c = (w * TPW + s) * 16 + lq * 4;
#pragma unroll UNR
            for (int nt = 0; nt < NTL; ++nt) {
                const int px = nt * 16 + lr;
                float4 yq;
                if constexpr (EARLY_YQ) yq = yqr[s * NTL + nt]; else yq = *reinterpret_cast<const float4*>(a.dyq + (row0 + px) * C + c);
                f32x4 oM = f32x4{0.f, 0.f, 0.f, 0.f}, oL = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int off = px * GPB + k * 64 + lq * 16;
                    const half8 gh = *reinterpret_cast<const half8*>(&Gp[0][off]);
                    const half8 gl = *reinterpret_cast<const half8*>(&Gp[1][off]);
                    oM = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgh[s][k], gh, oM, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgh[s][k], gl, oL, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgl[s][k], gh, oL, 0, 0, 0);
                }
                float inv;
                (void)grad_scale(fmaxf(fmaxf(PM[0][px], PM[1][px]), fmaxf(PM[2][px], PM[3][px])), inv);
                const f32x4 o = (oM + oL * H3_INV) * inv;
                *reinterpret_cast<float4*>(&DY[px * ZP + c]) = make_float4(o[0] + yq.x, o[1] + yq.y, o[2] + yq.z, o[3] + yq.w);
            }
        }
        __syncthreads();                                                            // (3) dy ; dk | dv planes and PM consumed
        FU_LA_MARK(36);
        // dx = beta dx + dout + LayerNorm'(x; g1) dy
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) {
            const int n = r * LN::RPP + lrow;
            const float4 dy = *reinterpret_cast<const float4*>(&DY[n * ZP + 4 * lcol]);
            const float mean = rowgroup_sum<LN::LPR>((xres[r].x + xres[r].y) + (xres[r].z + xres[r].w)) * (1.0f / C);
            const float d0 = xres[r].x - mean, d1 = xres[r].y - mean, d2 = xres[r].z - mean, d3 = xres[r].w - mean;
            const float rstd = 1.0f / sqrtf(rowgroup_sum<LN::LPR>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / C) + 1e-5f);
            const float z0 = d0 * rstd, z1 = d1 * rstd, z2 = d2 * rstd, z3 = d3 * rstd;
            const float t0 = gv.x * dy.x, t1 = gv.y * dy.y, t2 = gv.z * dy.z, t3 = gv.w * dy.w;
            const float m1 = rowgroup_sum<LN::LPR>((t0 + t1) + (t2 + t3)) * (1.0f / C);
            const float m2 = rowgroup_sum<LN::LPR>((t0 * z0 + t1 * z1) + (t2 * z2 + t3 * z3)) * (1.0f / C);
            float4 o = make_float4(rstd * (t0 - m1 - z0 * m2) + dor[r].x, rstd * (t1 - m1 - z1 * m2) + dor[r].y,
                                   rstd * (t2 - m1 - z2 * m2) + dor[r].z, rstd * (t3 - m1 - z3 * m2) + dor[r].w);
            float4* op = reinterpret_cast<float4*>(a.dx + (row0 + n) * C + 4 * lcol);
            if (a.beta != 0.f) { const float4 e = *op; o.x += a.beta * e.x; o.y += a.beta * e.y; o.z += a.beta * e.z; o.w += a.beta * e.w; }
            *op = o;
        }
        __syncthreads();                                                            // (4) dy consumed: the next tile rewrites Yp
        FU_LA_MARK(37);
    }
}

}  // namespace cindm
