// Device kernels of the 2-D airfoil path (model/diffusion_2d.py of the reference): Unet (:281-408) blocks and the
// boundary-sharing DDPM update (:712-845).  gfx950 only.
//
// Layout: channel-last fp32 [image, y, x, C] = rows (image*H*W + y*W + x) x C.
//
// conv2d_tile_kernel<KIND, MODE>: implicit-GEMM Conv2d on fp32 MFMA (v_mfma_f32_16x16x4_f32) -- the pixel-unshuffle
// 1x1 and, with CINDM_MFMA=f32, the 3x3 convolutions (default: conv2d_h3_kernel below).  One workgroup =
// a 4 x 16 pixel tile of one image x 64 output channels; K (input channels x taps) is split over the 4 waves inside
// every 32/64-channel chunk and reduced through LDS once at the end; B fragments come straight from L2 in fragment
// order; the A halo tile ((4+2) x (16+2) pixels for 3x3) is staged through LDS with normalise-on-load, image borders
// zero-filled at staging time, so every tap of every fragment row is the SAME constant LDS offset from a per-lane
// base (ds_read immediates, no address arithmetic in the loop).  Tap geometries: 3x3, 1x1, 3x3 over the
// nearest-neighbour x2 upsampled source (Upsample :99-103; the halo tile is staged from source pixel (y>>1, x>>1)),
// and the four taps of the pixel-unshuffle 1x1 (Downsample :105-109; an 8 x 32 source tile) -- neither the upsampled
// nor the unshuffled tensor is ever materialised.  conv2d_stem7_kernel: the 7x7 stem, K flattened over
// (tap, channel) so that the 21(24)-channel input wastes no MFMA lanes.
#pragma once
#include "kernels.h"

namespace cindm {

enum SrcMode2d { SRC2_PLAIN = 0, SRC2_GN_SS_SILU = 4, SRC2_LN = 2,
                 SRC2_SCALED = 5 };   // conv2d_ws_kernel only: plain source times a power of two taken from a device maximum (below)
enum ConvKind { CONV_3X3 = 0, CONV_1X1 = 1, CONV_UP2 = 2, CONV_UNSHUF = 3, CONV_STEM7 = 4, CONV_1X1_WIDE = 5,
                CONV_3X3_PAIR = 6 };   // conv2d_ws_kernel only: 8-pixel-wide images, two side by side per 8 x 16 tile
// Packed2::h3: 3x3 weights packed as split fp16 for conv2d_h3_kernel

constexpr int T2Y = 4, T2X = 16, T2M = T2Y * T2X;    // output pixel tile 4 x 16
constexpr int T2N = 64;                               // output channels per workgroup
constexpr int LDR2 = 33;                              // reduce tile pitch (32 columns per pass)

struct Conv2dArgs {
    Src src[2];           // stats: merged GroupNorm (mean, M2) [NI][8][2] (cnt = H*W*gw) or LN partials [rows][P][2]
    int nsrc;
    const float* W;       // [n-tile][stage][q][thread][4]
    const float* bias;
    int CinP, Npad, N;
    int NI, Hin, Win, Hout, Wout;
    int tiles_x, tpi;     // tiles per image row / per image
    float* out; int ldo;
    const float* res; int ldres;
    const float* e_y; int e_ld; const float* e_stats; int e_P; int e_gw; float e_cnt; const float* e_gamma; const float* e_beta;
    float* stats_out; int so_gw;        // GroupNorm (mean, M2) partials per tile: [NI][8][tpi][2]
    float* ln_out;                      // LayerNorm partials per pixel and 32-column block: [rows][Npad/32][2]
    const int* t_ptr; int t_imm;
    int dbg;                            // timing ablations (CINDM_DBG2; results are wrong when set)
    int stress;                         // conv2d_ws_kernel: > 0 = pseudo-random pauses before the hand-overs (stress_delay, kernels.h)
    // conv1x1_wide_kernel only: rows = pixels (2-D) or sequence positions (the 1-D path's qkv projections); blockIdx.y
    // selects a group of tiles_per_group output tiles (0 = all tiles in one workgroup)
    int64_t rows_total; int tiles_per_group;
    const float* lnr_g;                 // conv1x1_wide_kernel<.., LNR>: out = LayerNorm_channels(W x + bias) * lnr_g + res
};

__device__ __forceinline__ float silu_f(float x) {
    // x * sigmoid(x); exp2-based, one v_exp_f32 + one v_rcp_f32
    const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Wave-parallel merge of the P equal-count (mean, M2) partials of the 8 groups of one image: st = [8][P][2], lane =
// group * 8 + j, every lane returns its group's (mean, rstd).  Same formula as merge_stats.
__device__ __forceinline__ void merge_stats8(const float* __restrict__ st, int P, float cnt, int lane, float& mean, float& rstd) {
    const float2* p = reinterpret_cast<const float2*>(st) + (size_t)(lane >> 3) * P;
    const int j = lane & 7;
    float s = 0.f;
    for (int t = j; t < P; t += 8) s += p[t].x;
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float m = s / (float)P;
    float q = 0.f;
    for (int t = j; t < P; t += 8) { const float2 v = p[t]; const float d = v.x - m; q += v.y + cnt * d * d; }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
    mean = m;
    rstd = 1.0f / sqrtf(q / (cnt * (float)P) + 1e-5f);
}

// Value of lane 8 i + 7 (i = 0 .. 7 may differ per lane) without LDS-pipeline instructions: eight v_readlane + a select
// chain.  With seg_total(v, 8) in front this is "the total of 8-lane group i" -- the shuffle-free twin of merge_stats8's
// butterflies, used by the memory waves of conv2d_ws_kernel (their ds_bpermutes would queue behind the matrix waves'
// fragment reads).  No arrays: a dynamically indexed private array is promoted to LDS by the compiler.
__device__ __forceinline__ float group8_total(float v, int i) {
    const int b = __builtin_bit_cast(int, v);
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 7));
    r = (i == 1) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 15)) : r;
    r = (i == 2) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 23)) : r;
    r = (i == 3) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 31)) : r;
    r = (i == 4) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 39)) : r;
    r = (i == 5) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 47)) : r;
    r = (i == 6) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 55)) : r;
    r = (i == 7) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 63)) : r;
    return r;
}

template <int KIND> struct Cfg2 {
    static constexpr int KC = (KIND == CONV_1X1) ? 64 : 32;                       // channels per chunk
    static constexpr int CS = KC / 16;                                            // k-steps (of 4 channels) per wave per tap
    static constexpr int TPS = (KIND == CONV_1X1) ? 1 : (KIND == CONV_UNSHUF) ? 4 : 3;   // taps per B stage
    static constexpr int SPC = (KIND == CONV_3X3 || KIND == CONV_UP2) ? 3 : 1;    // B stages per chunk
    static constexpr int SW = (KIND == CONV_1X1) ? 16 : (KIND == CONV_UNSHUF) ? 32 : 18;  // staged tile width
    static constexpr int R = (KIND == CONV_1X1) ? 64 : (KIND == CONV_UNSHUF) ? 256 : 108; // staged rows
    static constexpr int LDAK = KC + 4;                                           // LDS row pitch (conflict-free fragments)
    static constexpr int NBF = TPS * CS;                                          // float4 of B per thread per stage
    static constexpr int F4 = KC / 4, RPP = 256 / F4, NP = (R + RPP - 1) / RPP;   // staging: float4 per row, rows per pass
};

// 16 MFMAs per k-step: 4 pixel rows x 4 column blocks; A fragments read one k-step ahead into a second register set
template <int KIND, int TR>
__device__ __forceinline__ void mma_stage(const float* __restrict__ Ab, const float (&b)[Cfg2<KIND>::NBF][4], f32x4 (&acc)[4][4]) {
    using C = Cfg2<KIND>;
    constexpr int NK = C::TPS * C::CS;
    float af[2][4];
    auto off = [](int mb, int ks) constexpr -> int {
        const int j = ks / C::CS, cs = ks % C::CS;
        int row = 0;
        if (KIND == CONV_1X1) row = mb * 16;
        else if (KIND == CONV_UNSHUF) row = (2 * mb + (j >> 1)) * 32 + (j & 1);
        else row = (mb + TR) * C::SW + j;
        return row * C::LDAK + cs * 4;
    };
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) af[0][mb] = Ab[off(mb, 0)];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + 1 < NK) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) af[(ks + 1) & 1][mb] = Ab[off(mb, ks + 1)];
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][mb], b[ks][nb], acc[mb][nb], 0, 0, 0);
    }
}

// Shared epilogue: cross-wave K reduction through LDS in two 32-column passes, bias, + SiLU(GroupNorm(e_y)),
// + residual, store, GroupNorm (mean, M2) partial of the tile, LayerNorm row partials.
__device__ __forceinline__ void conv2d_epilogue(const Conv2dArgs& a, f32x4 (&acc)[4][4], float (*Red)[T2M * LDR2], const float* tabE,
                                                int img, int ti, int ty0, int tx0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x;
    const int n = tid & 31, rq = tid >> 5;
    const size_t img_base = (size_t)img * a.Hout * a.Wout;
    size_t prow[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = rq + 8 * q;
        prow[q] = img_base + (size_t)(ty0 + (r >> 4)) * a.Wout + tx0 + (r & 15);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gn = nt * T2N + h * 32 + n;
        if (nt * T2N + h * 32 >= a.N) break;             // uniform: nothing real in this half
        const bool nok = gn < a.N;
        const float bias = (a.bias && nok) ? a.bias[gn] : 0.f;
        float eg = 1.f, eb = 0.f, em = 0.f, er = 1.f;
        float ey[8], rs[8];
        if (a.e_y && nok) {
            eg = a.e_gamma[gn]; eb = a.e_beta[gn];
            const int g = gn >> (31 - __builtin_clz(a.e_gw));
            em = tabE[2 * g]; er = tabE[2 * g + 1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ey[q] = (a.e_y && nok) ? a.e_y[prow[q] * a.e_ld + gn] : 0.f;
            rs[q] = (a.res && nok) ? a.res[prow[q] * a.ldres + gn] : 0.f;
        }
        if (h) __syncthreads();                            // pass 0's statistics readers are done with Red
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    Red[w][(mb * 16 + (lane >> 4) * 4 + rg) * LDR2 + nb * 16 + (lane & 15)] = acc[mb][2 * h + nb][rg];
        __syncthreads();
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = rq + 8 * q;
            float x = (Red[0][r * LDR2 + n] + Red[1][r * LDR2 + n]) + (Red[2][r * LDR2 + n] + Red[3][r * LDR2 + n]) + bias;
            if (a.e_y) x += silu_f((ey[q] - em) * er * eg + eb);
            if (a.res) x += rs[q];
            if (nok) a.out[prow[q] * a.ldo + gn] = x;
            v[q] = nok ? x : 0.f;
        }
        if (!a.stats_out && !a.ln_out) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) Red[0][(rq + 8 * q) * LDR2 + n] = v[q];     // own slots only
        __syncthreads();
        if (a.stats_out) {
            // GroupNorm partial of this tile: shifted one-pass sums about a pivot of the group, 8 rows per thread
            const int gwt = a.so_gw;                       // 8 or 16: whole groups inside the 32 columns
            const float K = Red[0][n & ~(gwt - 1)];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float d = v[q] - K; s1 += d; s2 += d * d; }
            Red[1][rq * 32 + n] = s1; Red[2][rq * 32 + n] = s2;
            __syncthreads();
            if (tid < 32) {
                s1 = 0.f; s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += Red[1][j * 32 + n]; s2 += Red[2][j * 32 + n]; }
                s1 = seg_total(s1, gwt);
                s2 = seg_total(s2, gwt);
                if ((n & (gwt - 1)) == gwt - 1 && nok) {
                    const float ne = (float)(T2M * gwt);
                    const int g = gn >> (31 - __builtin_clz(gwt));
                    float* o = a.stats_out + (((size_t)img * 8 + g) * a.tpi + ti) * 2;
                    o[0] = K + s1 / ne;
                    o[1] = fmaxf(s2 - s1 * s1 / ne, 0.f);
                }
            }
        }
        if (a.ln_out) {
            const int r = tid >> 2, sub = tid & 3;         // 64 rows x 4 sub-ranges of 8 columns
            const float K = Red[0][r * LDR2];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float d = Red[0][r * LDR2 + sub * 8 + c] - K; s1 += d; s2 += d * d; }
            s1 = seg_total(s1, 4);
            s2 = seg_total(s2, 4);
            if (sub == 3) {
                const size_t pr = img_base + (size_t)(ty0 + (r >> 4)) * a.Wout + tx0 + (r & 15);
                float* o = a.ln_out + (pr * (a.Npad / 32) + nt * 2 + h) * 2;
                o[0] = K + s1 * (1.0f / 32.0f);
                o[1] = fmaxf(s2 - s1 * s1 * (1.0f / 32.0f), 0.f);
            }
        }
    }
}

template <int KIND, int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_tile_kernel(const Conv2dArgs a) {
    using C = Cfg2<KIND>;
    constexpr int KC = C::KC, LDAK = C::LDAK, NBF = C::NBF, NP = C::NP, R = C::R, SW = C::SW, RPP = C::RPP, F4 = C::F4, SPC = C::SPC;
    __shared__ __attribute__((aligned(16))) float As[2][R * LDAK];
    __shared__ __attribute__((aligned(16))) float Red[4][T2M * LDR2];
    __shared__ float tabA[16];
    __shared__ float tabE[16];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int img = mt / a.tpi, ti = mt - img * a.tpi;
    const int tyi = ti / a.tiles_x;
    const int ty0 = tyi * T2Y, tx0 = (ti - tyi * a.tiles_x) * T2X;
    const int HWi = a.Hin * a.Win;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- staging geometry: this thread stages float4 c4 of rows r0 + RPP*p of the halo tile ------------------
    const int c4 = tid % F4, r0 = tid / F4;
    size_t goff[NP];
    bool rok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = r0 + RPP * p;
        const int hy = r / SW, hx = r - hy * SW;
        int y, x, srcpix;
        bool ok = r < R;
        if (KIND == CONV_3X3 || KIND == CONV_UP2) {
            y = ty0 - 1 + hy; x = tx0 - 1 + hx;
            ok = ok && (y >= 0) && (y < a.Hout) && (x >= 0) && (x < a.Wout);
            srcpix = (KIND == CONV_UP2) ? (y >> 1) * a.Win + (x >> 1) : y * a.Win + x;
        } else if (KIND == CONV_1X1) {
            y = ty0 + hy; x = tx0 + hx; srcpix = y * a.Win + x;
        } else {
            y = 2 * ty0 + hy; x = 2 * tx0 + hx; srcpix = y * a.Win + x;
        }
        rok[p] = ok;
        goff[p] = (size_t)img * HWi + (ok ? srcpix : 0);
    }
    float lnm[NP], lnr[NP];
    if constexpr (MODE == SRC2_LN) {
        const Src& s = a.src[0];
#pragma unroll
        for (int p = 0; p < NP; ++p) merge_stats(s.stats + goff[p] * s.P * 2, s.P, s.cnt, 1e-5f, lnm[p], lnr[p]);
    }

    const int nch0 = (a.src[0].C + KC - 1) / KC;
    const int nch = a.CinP / KC;                    // chunks (even or 1: host pads)
    const int nst = nch * SPC;
    float bA[NBF][4], bB[NBF][4];
    float4 areg[NP];
    const float4* wbase = reinterpret_cast<const float4*>(a.W) + (size_t)nt * nst * 256 * NBF + tid;
    auto load_b = [&](int st, float (&b)[NBF][4]) {
        const float4* wp = wbase + (size_t)st * 256 * NBF;
#pragma unroll
        for (int q = 0; q < NBF; ++q) {
            const float4 v = wp[q * 256];
            b[q][0] = v.x; b[q][1] = v.y; b[q][2] = v.z; b[q][3] = v.w;
        }
    };
    auto src_ptr = [&](int cc, int& cl, int& Cc, int& ld) -> const float* {
        const bool first = (cc < nch0) || (a.nsrc == 1);
        cl = (first ? cc : cc - nch0) * KC + c4 * 4;
        Cc = first ? a.src[0].C : a.src[1].C;
        ld = first ? a.src[0].ld : a.src[1].ld;
        return first ? a.src[0].p : a.src[1].p;
    };
    float4 pg = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), psc = pb, psh = pb;
    auto load_a = [&](int cc, float4 (&v)[NP]) {
        int cl, Cc, ld;
        const float* base = src_ptr(cc, cl, Cc, ld);
        const int clc = min(cl, Cc - 4);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            v[p] = *reinterpret_cast<const float4*>(base + goff[p] * ld + clc);
        if constexpr (MODE == SRC2_GN_SS_SILU || MODE == SRC2_LN) {
            const Src& s = a.src[0];
            pg = *reinterpret_cast<const float4*>(s.gamma + clc);
            if constexpr (MODE == SRC2_GN_SS_SILU) {
                pb = *reinterpret_cast<const float4*>(s.beta + clc);
                if (s.tb) {         // per-timestep row of this block: [scale C | shift C]
                    psc = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + clc);
                    psh = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + s.C + clc);
                }
            }
        }
    };
    load_b(0, bA);
    load_a(0, areg);

    if constexpr (MODE == SRC2_GN_SS_SILU) {
        const Src& s = a.src[0];
        if (w == 0) {                                          // per-tile partials [img][8][P][2], merged here
            float m, r;
            merge_stats8(s.stats + (size_t)img * 8 * s.P * 2, s.P, s.cnt, lane, m, r);
            if ((lane & 7) == 0) { tabA[2 * (lane >> 3)] = m; tabA[2 * (lane >> 3) + 1] = r; }
        }
    }
    if (a.e_y && w == 1) {
        float m, r;
        merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
        if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
    }
    const int gw_shift = 31 - __builtin_clz(a.src[0].gw | 1);

    auto store_a = [&](int cc, int buf, const float4 (&av)[NP]) {
        int cl, Cc, ld;
        (void)src_ptr(cc, cl, Cc, ld);
        const bool cok = cl < Cc;
        const int clc = min(cl, Cc - 4);
        float* dst = As[buf];
        float gm = 0.f, gr = 1.f;
        if constexpr (MODE == SRC2_GN_SS_SILU) { const int ti2 = (clc >> gw_shift) * 2; gm = tabA[ti2]; gr = tabA[ti2 + 1]; }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + RPP * p;
            float4 v = av[p];
            if constexpr (MODE == SRC2_GN_SS_SILU) {
                v.x = silu_f(((v.x - gm) * gr * pg.x + pb.x) * (psc.x + 1.0f) + psh.x);
                v.y = silu_f(((v.y - gm) * gr * pg.y + pb.y) * (psc.y + 1.0f) + psh.y);
                v.z = silu_f(((v.z - gm) * gr * pg.z + pb.z) * (psc.z + 1.0f) + psh.z);
                v.w = silu_f(((v.w - gm) * gr * pg.w + pb.w) * (psc.w + 1.0f) + psh.w);
            } else if constexpr (MODE == SRC2_LN) {
                v.x = (v.x - lnm[p]) * lnr[p] * pg.x; v.y = (v.y - lnm[p]) * lnr[p] * pg.y;
                v.z = (v.z - lnm[p]) * lnr[p] * pg.z; v.w = (v.w - lnm[p]) * lnr[p] * pg.w;
            }
            const bool ok = rok[p] && cok;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            if (R % RPP == 0 || r < R) *reinterpret_cast<float4*>(dst + r * LDAK + c4 * 4) = v;
        }
    };
    // per-lane fragment base: pixel column (lane & 15) of the tile, this wave's channel slice, k = lane >> 4
    const int abase = ((KIND == CONV_UNSHUF) ? 2 * (lane & 15) : (lane & 15)) * LDAK + w * (KC / 4) + (lane >> 4);

    __syncthreads();                 // tabA / tabE visible
    store_a(0, 0, areg);
    __syncthreads();

    if (nch == 1) {
        if constexpr (SPC == 3) {
            load_b(1, bB);
            mma_stage<KIND, 0>(As[0] + abase, bA, acc);
            load_b(2, bA);
            mma_stage<KIND, 1>(As[0] + abase, bB, acc);
            mma_stage<KIND, 2>(As[0] + abase, bA, acc);
        } else {
            mma_stage<KIND, 0>(As[0] + abase, bA, acc);
        }
    } else {
        for (int ch = 0; ch < nch; ch += 2) {
            const int chn = min(ch + 2, nch - 1);
            if constexpr (SPC == 3) {
                const int st = ch * 3;
                load_b(st + 1, bB);
                load_a(ch + 1, areg);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 0>(As[0] + abase, bA, acc);
                load_b(st + 2, bA);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 1>(As[0] + abase, bB, acc);
                load_b(st + 3, bB);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 2>(As[0] + abase, bA, acc);
                __builtin_amdgcn_sched_barrier(0);
                store_a(ch + 1, 1, areg);
                __syncthreads();
                load_b(st + 4, bA);
                load_a(chn, areg);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 0>(As[1] + abase, bB, acc);
                load_b(st + 5, bB);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 1>(As[1] + abase, bA, acc);
                load_b(min(st + 6, nst - 1), bA);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 2>(As[1] + abase, bB, acc);
                __builtin_amdgcn_sched_barrier(0);
                store_a(chn, 0, areg);
                __syncthreads();
            } else {
                load_b(ch + 1, bB);
                load_a(ch + 1, areg);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 0>(As[0] + abase, bA, acc);
                __builtin_amdgcn_sched_barrier(0);
                store_a(ch + 1, 1, areg);
                __syncthreads();
                load_b(chn, bA);
                load_a(chn, areg);
                __builtin_amdgcn_sched_barrier(0);
                mma_stage<KIND, 0>(As[1] + abase, bB, acc);
                __builtin_amdgcn_sched_barrier(0);
                store_a(chn, 0, areg);
                __syncthreads();
            }
        }
    }
    conv2d_epilogue(a, acc, Red, tabE, img, ti, ty0, tx0);
}

// Epilogue of conv2d_h3_kernel: wave (kg, nh) holds the partial 64 x 32 tile of column half nh over k-group kg.  One
// LDS round trip: every wave writes its tile, then all 256 threads sum the two k-group partials of BOTH halves (16
// outputs per thread), add the bias, store, and (optionally) reduce the GroupNorm (mean, M2) partial of the tile.  The
// 3x3 convolutions never carry the "+ SiLU(GN(y))" / residual / LayerNorm epilogue terms (those ride on 1x1 launches).
__device__ __forceinline__ void conv2d_h3_epilogue(const Conv2dArgs& a, f32x4 (&acc)[4][2], float (*Red)[T2M * LDR2],
                                                   int img, int ti, int ty0, int tx0) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                Red[w][(mb * 16 + (lane >> 4) * 4 + rg) * LDR2 + nb * 16 + (lane & 15)] = acc[mb][nb][rg];
    const int h = tid >> 7, n = tid & 31, rq = (tid >> 5) & 3;          // column half, column, row phase
    const int gn = nt * T2N + h * 32 + n;
    const bool nok = gn < a.N;
    // (pointer selected, load unconditional and requested before the barrier: `(a.bias && nok) ? a.bias[gn] : 0.f` is a load under
    // a branch whose join waits vmcnt(0))
    const float bias_ld = (a.bias ? a.bias : a.out)[nok ? gn : 0];
    const float bias = (a.bias && nok) ? bias_ld : 0.f;
    const size_t img_base = (size_t)img * a.Hout * a.Wout;
    __syncthreads();
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = rq + 4 * q;
        const float x = (Red[2 * h][r * LDR2 + n] + Red[2 * h + 1][r * LDR2 + n]) + bias;
        if (nok) a.out[(img_base + (size_t)(ty0 + (r >> 4)) * a.Wout + tx0 + (r & 15)) * a.ldo + gn] = x;
        v[q] = nok ? x : 0.f;
    }
    if (!a.stats_out) return;
    // GroupNorm partial of the tile per group: shifted sums about a pivot of the group (row 0 of its first column)
    const int gwt = a.so_gw;                               // 8 or 16
    __syncthreads();                                       // everyone is done reading the partial tiles
    if (rq == 0) Red[0][h * 32 + n] = v[0];                // row 0 of every column: the pivots
    __syncthreads();
    const float K = Red[0][h * 32 + (n & ~(gwt - 1))];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const float d = v[q] - K; s1 += d; s2 += d * d; }
    Red[1][(h * 4 + rq) * 32 + n] = s1; Red[2][(h * 4 + rq) * 32 + n] = s2;
    __syncthreads();
    if (rq == 0) {
        s1 = 0.f; s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += Red[1][(h * 4 + j) * 32 + n]; s2 += Red[2][(h * 4 + j) * 32 + n]; }
        s1 = seg_total(s1, gwt);
        s2 = seg_total(s2, gwt);
        if ((n & (gwt - 1)) == gwt - 1 && nok) {
            const float ne = (float)(T2M * gwt);
            const int g = gn >> (31 - __builtin_clz(gwt));
            float* o = a.stats_out + (((size_t)img * 8 + g) * a.tpi + ti) * 2;
            o[0] = K + s1 / ne;
            o[1] = fmaxf(s2 - s1 * s1 / ne, 0.f);
        }
    }
}

// conv2d_h3_kernel<KIND, MODE>: the 3x3 convolutions (KIND = CONV_3X3 / CONV_UP2) with every fp32 product evaluated on
// the fp16 matrix cores as the 3-term split of kernels.h (a = ah + 2^-11 al', b likewise; a.b ~ ah.bh + 2^-11 (ah.bl' +
// al'.bh), fp32 accumulation in two accumulator sets) -- fp32-faithful at 16/3 x the fp32 MFMA rate.  Same 4 x 16 pixel
// x 64 channel output tile and zero-filled halo staging as conv2d_tile_kernel; a pipeline chunk is 64 input channels;
// the 4 waves split the chunk's two 32-channel k-groups (kg = w & 1) and the two 32-column halves (nh = w >> 1), so a
// wave feeds v_mfma_f32_16x16x32_f16 with A fragments of 8 consecutive channels per lane (one ds_read_b128 per pixel
// block and plane, at constant offsets) and B fragments straight from L2 (ring of 3 tap slots, re-loaded right after
// use).  A is normalised in fp32 and split into hi / scaled-lo fp16 planes while it is staged.  The reduce tile aliases
// the A planes (62 KB of LDS per workgroup -> two workgroups per CU).
template <int KIND, int MODE>
__global__ __launch_bounds__(256, 2) void conv2d_h3_kernel(const Conv2dArgs a) {
    constexpr int KC = 64, SW = 18, R = 108, PITCH = 144, PLANE = R * PITCH, F4 = 16, RPP = 16, NP = (R + RPP - 1) / RPP;
    static_assert(4 * PLANE >= 4 * T2M * LDR2 * 4, "reduce tile must fit into the A planes");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * PLANE];       // [buffer][plane hi/lo][R][PITCH]
    __shared__ float tabA[16];
    __shared__ float tabE[16];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int kg = w & 1, nh = w >> 1;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int img = mt / a.tpi, ti = mt - img * a.tpi;
    const int tyi = ti / a.tiles_x;
    const int ty0 = tyi * T2Y, tx0 = (ti - tyi * a.tiles_x) * T2X;
    const int HWi = a.Hin * a.Win;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);

    f32x4 accM[4][2], accL[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int c4 = tid % F4, r0 = tid / F4;
    size_t goff[NP];
    bool rok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = r0 + RPP * p;
        const int hy = r / SW, hx = r - hy * SW;
        const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
        const bool ok = (r < R) && (y >= 0) && (y < a.Hout) && (x >= 0) && (x < a.Wout);
        const int srcpix = (KIND == CONV_UP2) ? (y >> 1) * a.Win + (x >> 1) : y * a.Win + x;
        rok[p] = ok;
        goff[p] = (size_t)img * HWi + (ok ? srcpix : 0);
    }
    const int nch0 = (a.src[0].C + KC - 1) / KC;
    const int nch = a.CinP / KC;
    const int ntap = nch * 9;

    // B: [n-tile][chunk][tap][q = nb*2 + plane][thread][8 halfs]
    half8 breg[3][2][2];
    const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + (size_t)nt * ntap * 4 * 256 + tid;
    auto load_b = [&](int g, int slot) {                 // g = chunk * 9 + tap (clamped by the caller)
        const uint4* wp = wbase + (size_t)g * 4 * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) breg[slot][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
    };
    float4 areg[NP];
    float4 pg = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), psc = pb, psh = pb;
    auto src_ptr = [&](int cc, int& cl, int& Cc, int& ld) -> const float* {
        const bool first = (cc < nch0) || (a.nsrc == 1);
        cl = (first ? cc : cc - nch0) * KC + c4 * 4;
        Cc = first ? a.src[0].C : a.src[1].C;
        ld = first ? a.src[0].ld : a.src[1].ld;
        return first ? a.src[0].p : a.src[1].p;
    };
    auto load_a = [&](int cc) {
        int cl, Cc, ld;
        const float* base = src_ptr(cc, cl, Cc, ld);
        const int clc = min(cl, Cc - 4);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            areg[p] = *reinterpret_cast<const float4*>(base + goff[p] * ld + clc);
        if constexpr (MODE == SRC2_GN_SS_SILU) {
            const Src& s = a.src[0];
            pg = *reinterpret_cast<const float4*>(s.gamma + clc);
            pb = *reinterpret_cast<const float4*>(s.beta + clc);
            if (s.tb) {
                psc = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + clc);
                psh = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + s.C + clc);
            }
        }
    };
    load_b(0, 0); load_b(min(1, ntap - 1), 1); load_b(min(2, ntap - 1), 2);
    load_a(0);
    if constexpr (MODE == SRC2_GN_SS_SILU) {
        const Src& s = a.src[0];
        if (w == 0) {                                          // per-tile partials [img][8][P][2], merged here
            float m, r;
            merge_stats8(s.stats + (size_t)img * 8 * s.P * 2, s.P, s.cnt, lane, m, r);
            if ((lane & 7) == 0) { tabA[2 * (lane >> 3)] = m; tabA[2 * (lane >> 3) + 1] = r; }
        }
    }
    if (a.e_y && w == 1) {
        float m, r;
        merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
        if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
    }
    const int gw_shift = 31 - __builtin_clz(a.src[0].gw | 1);

    auto store_a = [&](int cc, int buf) {
        int cl, Cc, ld;
        (void)src_ptr(cc, cl, Cc, ld);
        const bool cok = cl < Cc;
        const int clc = min(cl, Cc - 4);
        unsigned char* P0 = smem + (size_t)(buf * 2) * PLANE;
        unsigned char* P1 = P0 + PLANE;
        float gm = 0.f, gr = 1.f;
        if constexpr (MODE == SRC2_GN_SS_SILU) { const int ti2 = (clc >> gw_shift) * 2; gm = tabA[ti2]; gr = tabA[ti2 + 1]; }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + RPP * p;
            float4 v = areg[p];
            if constexpr (MODE == SRC2_GN_SS_SILU) {
                v.x = silu_f(((v.x - gm) * gr * pg.x + pb.x) * (psc.x + 1.0f) + psh.x);
                v.y = silu_f(((v.y - gm) * gr * pg.y + pb.y) * (psc.y + 1.0f) + psh.y);
                v.z = silu_f(((v.z - gm) * gr * pg.z + pb.z) * (psc.z + 1.0f) + psh.z);
                v.w = silu_f(((v.w - gm) * gr * pg.w + pb.w) * (psc.w + 1.0f) + psh.w);
            }
            const bool ok = rok[p] && cok;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            if (r < R) {
                *reinterpret_cast<half4v*>(P0 + r * PITCH + c4 * 8) = hi;
                *reinterpret_cast<half4v*>(P1 + r * PITCH + c4 * 8) = lo;
            }
        }
    };
    // per-lane fragment byte offset inside a plane: pixel column (lane & 15), this wave's k-group, 8 channels at (lane >> 4) * 8
    const int abase = (lane & 15) * PITCH + kg * 64 + (lane >> 4) * 16;

    if (a.dbg == 1) return;
    __syncthreads();
    store_a(0, 0);
    __syncthreads();
    if (a.dbg == 2) return;

    for (int ch = 0; ch < (a.dbg == 3 ? 0 : nch); ++ch) {
        const int chn = min(ch + 1, nch - 1);
        load_a(chn);
        const unsigned char* P0 = smem + (size_t)((ch & 1) * 2) * PLANE + abase;
        const unsigned char* P1 = P0 + PLANE;
        half8 fh[2][4], fl[2][4];
        auto read_frags = [&](int tap, half8 (&ah)[4], half8 (&al)[4]) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                ah[mb] = *reinterpret_cast<const half8*>(P0 + ((mb + dy) * SW + dx) * PITCH);
                al[mb] = *reinterpret_cast<const half8*>(P1 + ((mb + dy) * SW + dx) * PITCH);
            }
        };
        read_frags(0, fh[0], fl[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) read_frags(tap + 1, fh[(tap + 1) & 1], fl[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 (&ah)[4] = fh[tap & 1];
            const half8 (&al)[4] = fl[tap & 1];
            const int slot = tap % 3;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], breg[slot][nb][0], accM[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], breg[slot][nb][1], accL[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], breg[slot][nb][0], accL[mb][nb], 0, 0, 0);
                }
            load_b(min(ch * 9 + tap + 3, ntap - 1), slot);       // this slot's next tap, two taps ahead of its use
        }
        store_a(chn, (ch + 1) & 1);
        __syncthreads();
    }

    // combine the two accumulator sets; the reduce tile aliases the A planes (everyone is past the last barrier of the loop)
    f32x4 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = accM[mb][nb] + accL[mb][nb] * H3_INV;
    if (a.dbg == 4) { if (acc[0][0][0] == 123.456f) a.out[0] = 1.f; return; }
    conv2d_h3_epilogue(a, acc, reinterpret_cast<float (*)[T2M * LDR2]>(smem), img, ti, ty0, tx0);
}

// conv2d_stem7_h3_kernel: the 7x7 stem on the split-fp16 path.  Halo tile 10 x 22 pixels; the 24 state channels are
// padded to one 32-channel k-step (hi / scaled-lo fp16 planes, 80 B per pixel), so a tap is ONE v_mfma_f32_16x16x32_f16
// k-step.  The 49 taps are split over the two wave pairs (kg = w & 1 takes taps kg, kg+2, ...), the 64 output columns
// over nh = w >> 1 -- the same (k-group, column half) structure and epilogue as conv2d_h3_kernel.
constexpr int STEM3_NI = 25;             // tap slots per wave pair (the 25th of the odd pair is a zero pad)
__global__ __launch_bounds__(256, 2) void conv2d_stem7_h3_kernel(const Conv2dArgs a) {
    constexpr int SW = 22, SH = 10, R = SW * SH, PITCH = 80, PLANE = R * PITCH, NP = (R * 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char planes[2 * PLANE];
    __shared__ __attribute__((aligned(16))) float Red[4][T2M * LDR2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int kg = w & 1, nh = w >> 1;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int img = mt / a.tpi, ti = mt - img * a.tpi;
    const int tyi = ti / a.tiles_x;
    const int ty0 = tyi * T2Y, tx0 = (ti - tyi * a.tiles_x) * T2X;
    const int HWi = a.Hin * a.Win;

    f32x4 accM[4][2], accL[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // B: [n-tile][tap slot i][q = nb*2 + plane][thread][8 halfs]; two register sets alternate over the tap slots
    half8 bs[2][2][2];
    const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + (size_t)nt * STEM3_NI * 4 * 256 + tid;
    auto load_b = [&](int i, half8 (&b)[2][2]) {
        const uint4* wp = wbase + (size_t)i * 4 * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
    };
    load_b(0, bs[0]);
    // stage the halo tile: R rows x 8 float4 (channels >= the source's pitch are zero)
    const float* src = a.src[0].p;
    const int ld = a.src[0].ld;
    // all NP requests first, then the conversions (round 4, from the ISA: `if (ok) v = load` followed by its conversion compiled
    // to load -> vmcnt(0) -> convert -> store per pass, i.e. NP = 7 serial trips to memory at the head of every workgroup -- the
    // launch ran at ~0.8 TB/s).  Addresses are clamped, loads unconditional.
    float4 sv[NP];
    bool okv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = tid + 256 * p;
        const int r = i >> 3, c4 = i & 7;
        const int hy = r / SW, hx = r - hy * SW;
        const int y = ty0 - 3 + hy, x = tx0 - 3 + hx;
        okv[p] = (r < R) && (c4 * 4 < ld) && (y >= 0) && (y < a.Hin) && (x >= 0) && (x < a.Win);
        const size_t off = okv[p] ? ((size_t)img * HWi + (size_t)y * a.Win + x) * ld + c4 * 4 : (size_t)img * HWi * ld;
        sv[p] = *reinterpret_cast<const float4*>(src + off);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = tid + 256 * p;
        const int r = i >> 3, c4 = i & 7;
        const float4 v = okv[p] ? sv[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        half4v hi, lo;
        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
        lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
        if (r < R) {
            *reinterpret_cast<half4v*>(planes + r * PITCH + c4 * 8) = hi;
            *reinterpret_cast<half4v*>(planes + PLANE + r * PITCH + c4 * 8) = lo;
        }
    }
    __syncthreads();
    const unsigned char* P0 = planes + (lane & 15) * PITCH + (lane >> 4) * 16;
    const unsigned char* P1 = P0 + PLANE;
#pragma unroll 1
    for (int i = 0; i < STEM3_NI; i += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ii = i + u;
            if (ii < STEM3_NI) {
                load_b(min(ii + 1, STEM3_NI - 1), bs[(u + 1) & 1]);
                const int tap = min(2 * ii + kg, 48);                 // wave-uniform; the pad slot carries zero weights
                const int dy = tap / 7, dx = tap - dy * 7;
                const int o = (dy * SW + dx) * PITCH;
                half8 ah[4], al[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    ah[mb] = *reinterpret_cast<const half8*>(P0 + o + mb * SW * PITCH);
                    al[mb] = *reinterpret_cast<const half8*>(P1 + o + mb * SW * PITCH);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bs[u][nb][0], accM[mb][nb], 0, 0, 0);
                        accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bs[u][nb][1], accL[mb][nb], 0, 0, 0);
                        accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], bs[u][nb][0], accL[mb][nb], 0, 0, 0);
                    }
            }
        }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = accM[mb][nb] + accL[mb][nb] * H3_INV;
    conv2d_h3_epilogue(a, acc, Red, img, ti, ty0, tx0);
}

// conv2d_stem7_h3d_kernel (round 5): the stem with a DENSE reduction axis and TWICE the pixels per weight fragment.
// What bounds conv2d_stem7_h3_kernel is neither its 600 MFMAs per wave nor its stores but the WEIGHT STREAM: every 64-pixel workgroup
// reads the whole 410 KB pack from L2 -- 3.4 GB per launch, 16 TB/s at 206 us, every CU asking for the same lines (ablations of
// tools/r5_exp_stem2.py: three slots of products instead of all: the launch all but disappears; 25 % fewer MFMAs at the same bytes
// per product, three workgroups per CU, a deeper weight prefetch: each neutral or slower).  So: (1) a pixel is 24 halfs = 48 B in
// the LDS planes and the seven taps of one kernel ROW of an output pixel are 168 CONTIGUOUS halfs starting at its window pixel: five
// k-steps cover 160 of them (lane group lg reads halfs 32 s + 8 lg .. + 7 -- any 16-byte-aligned address is a legal fragment), and
// the seven rows' last 8 halfs (tap column 6, channels 16 - 23) are gathered into two more k-steps (lane group lg = kernel row
// lg + 4 j): 37 k-steps instead of one per tap = 49, a pack of 311 KB; (2) a workgroup owns 8 x 16 pixels: a weight fragment feeds
// eight pixel blocks, 155 KB of weights per 64 pixels instead of 410.  Slots: k-step 2 i + kg for the wave pair kg, 19 slots (the
// 38th k-step is a zero pad).  K-group reduction and epilogue of conv2d_stem7_h3_kernel, once per half of the tile.
constexpr int STEMD_NI = 19, STEMD_KS = 37, STEMD_TY = 8;
__global__ __launch_bounds__(256, 2) void conv2d_stem7_h3d_kernel(const Conv2dArgs a) {
    constexpr int SW = 22, SH = STEMD_TY + 6, R = SW * SH, PITCH = 48, PLANE = R * PITCH, NP = (R * 6 + 255) / 256;
    // the reduction / store tile of the epilogue lives in the planes' bytes (barrier in between)
    constexpr int RED_BYTES = 4 * T2M * LDR2 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RED_BYTES > 2 * PLANE ? RED_BYTES : 2 * PLANE];
    unsigned char* planes = smem;
    float (*Red)[T2M * LDR2] = reinterpret_cast<float (*)[T2M * LDR2]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int kg = w & 1;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int tpi8 = (a.Hout / STEMD_TY) * a.tiles_x;
    const int img = mt / tpi8, ti = mt - img * tpi8;
    const int tyi = ti / a.tiles_x;
    const int ty0 = tyi * STEMD_TY, tx0 = (ti - tyi * a.tiles_x) * T2X;
    const int HWi = a.Hin * a.Win;

    f32x4 accM[8][2], accL[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    half8 bs[2][2][2];
    const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + (size_t)nt * STEMD_NI * 4 * 256 + tid;
    auto load_b = [&](int i, half8 (&b)[2][2]) {
        const uint4* wp = wbase + (size_t)i * 4 * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
    };
    load_b(0, bs[0]);
    const float* src = a.src[0].p;
    const int ld = a.src[0].ld;
    {
        float4 sv[NP];
        bool okv[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + 256 * p;
            const int r = i / 6, c4 = i - r * 6;
            const int hy = r / SW, hx = r - hy * SW;
            const int y = ty0 - 3 + hy, x = tx0 - 3 + hx;
            okv[p] = (r < R) && (c4 * 4 < ld) && (y >= 0) && (y < a.Hin) && (x >= 0) && (x < a.Win);
            const size_t off = okv[p] ? ((size_t)img * HWi + (size_t)y * a.Win + x) * ld + c4 * 4 : (size_t)img * HWi * ld;
            sv[p] = *reinterpret_cast<const float4*>(src + off);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + 256 * p;
            const int r = i / 6, c4 = i - r * 6;
            const float4 v = okv[p] ? sv[p] : make_float4(0.f, 0.f, 0.f, 0.f);
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            if (r < R) {
                *reinterpret_cast<half4v*>(planes + r * PITCH + c4 * 8) = hi;
                *reinterpret_cast<half4v*>(planes + PLANE + r * PITCH + c4 * 8) = lo;
            }
        }
    }
    __syncthreads();
    const int lq = lane & 15, lg = lane >> 4;
    const unsigned char* P0 = planes + lq * PITCH + lg * 16;
    // the gathered k-steps: lane group lg reads kernel row min(lg + 4 j, 6), tap column 6, channels 16 - 23 (row 7 of j = 1: zero weights)
    const unsigned char* PB[2] = {planes + (lg * SW + lq + 6) * PITCH + 32, planes + (min(lg + 4, 6) * SW + lq + 6) * PITCH + 32};
#pragma unroll 1
    for (int i = 0; i < (a.dbg == 22 ? 2 : STEMD_NI + 1); i += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ii = i + u;
            if (ii < STEMD_NI) {
                if (a.dbg != 21) load_b(min(ii + 1, STEMD_NI - 1), bs[(u + 1) & 1]);
                const int ks = min(2 * ii + kg, STEMD_KS - 1);            // wave-uniform; the pad slot carries zero weights
                const int dy = ks / 5, st = ks - dy * 5;
                const unsigned char* pa = ks < 35 ? P0 + dy * SW * PITCH + st * 64 : PB[ks - 35];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    half8 ah[4], al[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        ah[mb] = *reinterpret_cast<const half8*>(pa + (hf * 4 + mb) * SW * PITCH);
                        al[mb] = *reinterpret_cast<const half8*>(pa + PLANE + (hf * 4 + mb) * SW * PITCH);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (without it: 0.95 x instead of 0.92 x of the per-tap kernel)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            accM[hf * 4 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bs[u][nb][0], accM[hf * 4 + mb][nb], 0, 0, 0);
                            accL[hf * 4 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], bs[u][nb][1], accL[hf * 4 + mb][nb], 0, 0, 0);
                            accL[hf * 4 + mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], bs[u][nb][0], accL[hf * 4 + mb][nb], 0, 0, 0);
                        }
                }
            }
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        f32x4 acc[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = accM[hf * 4 + mb][nb] + accL[hf * 4 + mb][nb] * H3_INV;
        if (a.dbg == 23) { if (acc[0][0][0] == 123.456f) a.out[0] = 1.f; continue; }
        __syncthreads();                      // every wave is done with the planes / with the other half's tile
        conv2d_h3_epilogue(a, acc, Red, img, 0, ty0 + 4 * hf, tx0);        // (no GroupNorm partials on this path: the host checks)
    }
}

// 7x7 stem (init_conv, :303): input = the padded state (CP = 24 channels, 21 real).  Halo tile 10 x 22 pixels x 24
// channels staged once; K = 49 taps x 6 channel-quads = 294 k-steps (padded to 320), k-step 4i + w belongs to wave w,
// so no MFMA is spent on padding channels beyond 24.  8 k-steps per B stage, 10 stages.
constexpr int STEM_KSTEPS = 320, STEM_NBF = 8, STEM_NST = 10;
__global__ __launch_bounds__(256, 2) void conv2d_stem7_kernel(const Conv2dArgs a) {
    constexpr int CP = 24, LDAK = 28, SW = 22, SH = 10, R = SW * SH, NP = (R * 6 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[R * LDAK];
    __shared__ __attribute__((aligned(16))) float Red[4][T2M * LDR2];
    __shared__ float tabE[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int img = mt / a.tpi, ti = mt - img * a.tpi;
    const int tyi = ti / a.tiles_x;
    const int ty0 = tyi * T2Y, tx0 = (ti - tyi * a.tiles_x) * T2X;
    const int HWi = a.Hin * a.Win;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float bA[STEM_NBF][4], bB[STEM_NBF][4];
    const float4* wbase = reinterpret_cast<const float4*>(a.W) + (size_t)nt * STEM_NST * 256 * STEM_NBF + tid;
    auto load_b = [&](int st, float (&b)[STEM_NBF][4]) {
        const float4* wp = wbase + (size_t)st * 256 * STEM_NBF;
#pragma unroll
        for (int q = 0; q < STEM_NBF; ++q) {
            const float4 v = wp[q * 256];
            b[q][0] = v.x; b[q][1] = v.y; b[q][2] = v.z; b[q][3] = v.w;
        }
    };
    load_b(0, bA);
    // stage the halo tile: R rows x 6 float4
    const float* src = a.src[0].p;
    const int ld = a.src[0].ld;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = tid + 256 * p;
        const int r = i / 6, c4 = i - r * 6;
        const int hy = r / SW, hx = r - hy * SW;
        const int y = ty0 - 3 + hy, x = tx0 - 3 + hx;
        const bool ok = (r < R) && (y >= 0) && (y < a.Hin) && (x >= 0) && (x < a.Win);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(src + ((size_t)img * HWi + (size_t)y * a.Win + x) * ld + c4 * 4);
        if (r < R) *reinterpret_cast<float4*>(As + r * LDAK + c4 * 4) = v;
    }
    __syncthreads();
    const float* Ab = As + (lane & 15) * LDAK + (lane >> 4);
    auto stage = [&](int st, const float (&b)[STEM_NBF][4]) {
#pragma unroll
        for (int q = 0; q < STEM_NBF; ++q) {
            const int kidx = min(4 * (st * STEM_NBF + q) + w, 293);          // wave-uniform; padded k-steps carry zero weights
            const int tap = kidx / 6, cs = kidx - tap * 6;
            const int dy = tap / 7, dx = tap - dy * 7;
            const float* ap = Ab + (dy * SW + dx) * LDAK + cs * 4;
            float af[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) af[mb] = ap[mb * SW * LDAK];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], b[q][nb], acc[mb][nb], 0, 0, 0);
        }
    };
    for (int st = 0; st < STEM_NST; st += 2) {
        load_b(st + 1, bB);
        __builtin_amdgcn_sched_barrier(0);
        stage(st, bA);
        load_b(min(st + 2, STEM_NST - 1), bA);
        __builtin_amdgcn_sched_barrier(0);
        stage(st + 1, bB);
    }
    conv2d_epilogue(a, acc, Red, tabE, img, ti, ty0, tx0);
}

// conv1x1_wide_kernel<KT, MODE, RES, EPI>: every 1x1 convolution of the Unet (attention qkv / out projections, the
// ResnetBlock tail "SiLU(GN(y1)) + res_conv(x)" -- with an identity mode when there is no res_conv --, the final
// conv), KT = 64 / 128 / 192 input channels from one or two concatenated sources.  One workgroup = 64 consecutive
// pixels; the input tile is staged through LDS once (LayerNorm-on-load in MODE SRC2_LN) and the workgroup loops over
// the N / 64 output tiles with no barrier: wave w owns 16 output channels of each tile and computes
// out^T[n][px] = W[n][:] . x^T[:][px] (A = weights straight from L2 in fragment order, B = the input fragments, kept
// in registers for the whole loop when RES -- the wide qkv projections), so there is NO cross-wave reduction: the epilogue (EPI) runs on the
// accumulators -- each lane holds 4 consecutive channels of pixel (lane & 15) + 16 pb -- with float4 loads of the
// epilogue operands and float4 stores.  LayerNorm partials of the output are per pixel and per 16 channels.
template <int KT, int MODE, bool RES, bool EPI, bool LNR = false>
__global__ __launch_bounds__(256, RES ? 2 : 3) void conv1x1_wide_kernel(const Conv2dArgs a) {
    constexpr int LDA = KT + 4, F4 = KT / 4, NKS = KT / 4, NQ = KT / 16;
    constexpr int NPASS = (64 * F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[64 * LDA];
    __shared__ float tabE[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int HWo = a.Hout * a.Wout;
    const int img = (int)(row0 / HWo);
    const Src& s0 = a.src[0];
    const int ntile = a.Npad / T2N;
    const int tpg = a.tiles_per_group > 0 ? a.tiles_per_group : ntile;
    const int it_lo = blockIdx.y * tpg, it_hi = min(ntile, it_lo + tpg);
    const float4* wbase = reinterpret_cast<const float4*>(a.W) + tid;
    float4 wA[NQ], wB[NQ];
    auto load_w = [&](int it, float4 (&wv)[NQ]) {
        if (a.W) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) wv[q] = wbase[((size_t)it * NQ + q) * 256];
        }
    };
    load_w(it_lo, wA);
    if constexpr (EPI) {
        if (a.e_y && w == 1) {
            float m, r;
            merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
            if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
        }
    }
    // staging: all global loads first (registers), then -- LayerNorm mode -- one statistics merge per row into an LDS
    // table, then normalise + store
    __shared__ float tabL[128];
    float4 sv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int i = tid + 256 * p;
        const int r = i / F4, c4 = i - r * F4;
        sv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 64 * F4) {
            const int cl = c4 * 4;
            const bool first = cl < s0.C || a.nsrc == 1;
            const Src& s = first ? a.src[0] : a.src[1];
            const int cs = first ? cl : cl - s0.C;
            if (cs < s.C && (int64_t)(row0 + r) < a.rows_total) sv[p] = *reinterpret_cast<const float4*>(s.p + (row0 + r) * s.ld + cs);
        }
    }
    if constexpr (MODE == SRC2_LN) {
        if (tid < 64) {
            float mean = 0.f, rstd = 0.f;
            if ((int64_t)(row0 + tid) < a.rows_total) merge_stats(s0.stats + (row0 + tid) * s0.P * 2, s0.P, s0.cnt, 1e-5f, mean, rstd);
            tabL[2 * tid] = mean; tabL[2 * tid + 1] = rstd;
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int i = tid + 256 * p;
        const int r = i / F4, c4 = i - r * F4;
        if (i < 64 * F4) {
            float4 v = sv[p];
            if constexpr (MODE == SRC2_LN) {
                const int cl = c4 * 4;
                if (cl < s0.C) {
                    const float4 pg = *reinterpret_cast<const float4*>(s0.gamma + cl);
                    const float mean = tabL[2 * r], rstd = tabL[2 * r + 1];
                    v.x = (v.x - mean) * rstd * pg.x; v.y = (v.y - mean) * rstd * pg.y;
                    v.z = (v.z - mean) * rstd * pg.z; v.w = (v.w - mean) * rstd * pg.w;
                }
            }
            *reinterpret_cast<float4*>(As + r * LDA + c4 * 4) = v;
        }
    }
    __syncthreads();
    float xb[RES ? NKS : 1][4];       // resident B fragments: x[px = pb*16 + lq][k = ks*4 + lg]
    if constexpr (RES) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) xb[ks][pb] = As[(pb * 16 + lq) * LDA + ks * 4 + lg];
    }

    if constexpr (LNR) {
        // LinearAttention's to_out: Conv 1x1 (+bias) -> LayerNorm over ALL output channels -> * g -> + residual (:233-236,
        // :96-97).  N <= 128: both output tiles stay in accumulators; the per-pixel statistics cross the four waves
        // (16 channels each per tile) through a 2 KB LDS table, two-pass (mean, then centred squares).
        __shared__ float psum[2][4][64];
        f32x4 accs[2][4];
        if (ntile > 1) load_w(1, wB);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < ntile) {
                const float4 (&wv)[NQ] = t ? wB : wA;
                const float4 bias = *reinterpret_cast<const float4*>(a.bias + t * T2N + w * 16 + lg * 4);
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) accs[t][pb] = (f32x4){bias.x, bias.y, bias.z, bias.w};
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float wq[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int pb = 0; pb < 4; ++pb)
                            accs[t][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[j], As[(pb * 16 + lq) * LDA + (q * 4 + j) * 4 + lg], accs[t][pb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) accs[t][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const float invn = 1.0f / (float)a.N;
        float mean[4], rstd[4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                float sm = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (t < ntile) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) { const float d = pass ? accs[t][pb][rg] - mean[pb] : accs[t][pb][rg]; sm += pass ? d * d : d; }
                    }
                sm = xsum32(xsum16(sm));
                if (lg == 0) psum[pass][w][pb * 16 + lq] = sm;
            }
            __syncthreads();
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int px = pb * 16 + lq;
                const float tot = (psum[pass][0][px] + psum[pass][1][px]) + (psum[pass][2][px] + psum[pass][3][px]);
                if (pass == 0) mean[pb] = tot * invn; else rstd[pb] = 1.0f / sqrtf(tot * invn + 1e-5f);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < ntile) {
                const int col = t * T2N + w * 16 + lg * 4;
                const float4 g = *reinterpret_cast<const float4*>(a.lnr_g + col);
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const size_t prow = row0 + pb * 16 + lq;
                    const float4 r4 = *reinterpret_cast<const float4*>(a.res + prow * a.ldres + col);
                    float4 v;
                    v.x = (accs[t][pb][0] - mean[pb]) * rstd[pb] * g.x + r4.x; v.y = (accs[t][pb][1] - mean[pb]) * rstd[pb] * g.y + r4.y;
                    v.z = (accs[t][pb][2] - mean[pb]) * rstd[pb] * g.z + r4.z; v.w = (accs[t][pb][3] - mean[pb]) * rstd[pb] * g.w + r4.w;
                    *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) = v;
                }
            }
        return;
    }
    auto tile = [&](int it, const float4 (&wv)[NQ]) {
        const int col = it * T2N + w * 16 + lg * 4;           // this lane's 4 consecutive output channels
        f32x4 acc[4];
        if (a.W) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < (NQ); ++q) {
                if (a.dbg == 6) { acc[0][0] += wv[q].x; continue; }
                const float wq[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        const float xv = RES ? xb[RES ? q * 4 + j : 0][pb] : As[(pb * 16 + lq) * LDA + (q * 4 + j) * 4 + lg];
                        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[j], xv, acc[pb], 0, 0, 0);
                    }
            }
        } else {                                              // identity "convolution": out = x (+ epilogue terms)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const float4 v = *reinterpret_cast<const float4*>(As + (pb * 16 + lq) * LDA + col);
                acc[pb] = (f32x4){v.x, v.y, v.z, v.w};
            }
        }
        if (col >= a.N) return;                               // N is a multiple of 4 except the final conv (handled below)
        if constexpr (!EPI) {                                 // plain projection: store and done
            if (a.dbg == 5) { if (acc[0][0] == 123.456f) a.out[0] = 1.f; return; }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
                if ((int64_t)(row0 + pb * 16 + lq) < a.rows_total)
                    *reinterpret_cast<float4*>(a.out + (row0 + pb * 16 + lq) * a.ldo + col) =
                        make_float4(acc[pb][0], acc[pb][1], acc[pb][2], acc[pb][3]);
            return;
        }
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool full = col + 3 < a.N;
        if (a.bias) {
            if (full) bias = *reinterpret_cast<const float4*>(a.bias + col);
            else { bias.x = a.bias[col]; if (col + 1 < a.N) bias.y = a.bias[col + 1]; if (col + 2 < a.N) bias.z = a.bias[col + 2]; }
        }
        float4 eg = make_float4(1.f, 1.f, 1.f, 1.f), eb = make_float4(0.f, 0.f, 0.f, 0.f);
        float em = 0.f, er = 1.f;
        if (a.e_y) {
            eg = *reinterpret_cast<const float4*>(a.e_gamma + col); eb = *reinterpret_cast<const float4*>(a.e_beta + col);
            const int g = col >> (31 - __builtin_clz(a.e_gw));
            em = tabE[2 * g]; er = tabE[2 * g + 1];
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const size_t prow = row0 + pb * 16 + lq;
            float4 v = make_float4(acc[pb][0] + bias.x, acc[pb][1] + bias.y, acc[pb][2] + bias.z, acc[pb][3] + bias.w);
            if (a.e_y) {
                const float4 y = *reinterpret_cast<const float4*>(a.e_y + prow * a.e_ld + col);
                v.x += silu_f((y.x - em) * er * eg.x + eb.x); v.y += silu_f((y.y - em) * er * eg.y + eb.y);
                v.z += silu_f((y.z - em) * er * eg.z + eb.z); v.w += silu_f((y.w - em) * er * eg.w + eb.w);
            }
            if (a.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(a.res + prow * a.ldres + col);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            if (!full) {          // the row pitch is N rounded up to 4 (host): the padding channels are written as zeros
                if (col + 1 >= a.N) v.y = 0.f;
                if (col + 2 >= a.N) v.z = 0.f;
                v.w = 0.f;
            }
            *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) = v;
            if (a.ln_out) {
                // LayerNorm partial of this pixel over the wave's 16 channels: 4 in this lane, 4 lanes (lg) per pixel
                float sm = (v.x + v.y) + (v.z + v.w);
                sm = xsum32(xsum16(sm));
                const float mean = sm * (1.0f / 16.0f);
                const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                m2 = xsum32(xsum16(m2));
                if (lg == 0) {
                    float* o = a.ln_out + (prow * (a.Npad / 16) + it * 4 + w) * 2;
                    o[0] = mean; o[1] = m2;
                }
            }
        }
    };
    for (int it = it_lo; it < it_hi; it += 2) {
        load_w(min(it + 1, it_hi - 1), wB);
        __builtin_amdgcn_sched_barrier(0);
        tile(it, wA);
        load_w(min(it + 2, it_hi - 1), wA);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < it_hi) tile(it + 1, wB);
    }
}

// tail_identity_kernel<C>: the ResnetBlock tail without a res_conv -- out = x + SiLU(GroupNorm(y1)) (+ LayerNorm partials when an
// attention follows) -- as a plain element-wise pass.  Through conv1x1_wide_kernel's identity mode the x tile went global ->
// registers -> LDS -> barrier -> registers and the y1 rows were requested only in the epilogue: two HBM round trips in series
// per 64-pixel workgroup (3.2 TB/s at 64 x 64, 128 images).  Here a thread owns channel quad (tid % (C/4)) of rows tid / (C/4) +
// k * (256 / (C/4)), requests all its x and y1 quads at once, and writes the same expression in the same order (bit-identical
// output).  Workgroup = 64 rows of one image; the LayerNorm partial of a pixel's 16 channels is a quad reduction.
template <int C>
__global__ __launch_bounds__(256) void tail_identity_kernel(const Conv2dArgs a) {
    constexpr int F4 = C / 4, RPP = 256 / F4, NPASS = 64 / RPP;
    __shared__ float tabE[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int img = (int)(row0 / (a.Hout * a.Wout));
    const int c4 = tid % F4, r0 = tid / F4, col = c4 * 4;
    const Src& s0 = a.src[0];
    float4 xv[NPASS], yv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const size_t prow = row0 + r0 + p * RPP;
        xv[p] = *reinterpret_cast<const float4*>(s0.p + prow * s0.ld + col);
        yv[p] = *reinterpret_cast<const float4*>(a.e_y + prow * a.e_ld + col);
    }
    if (w == 1) {
        float m, r;
        merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
        if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
    }
    const float4 eg = *reinterpret_cast<const float4*>(a.e_gamma + col), eb = *reinterpret_cast<const float4*>(a.e_beta + col);
    __syncthreads();
    const int g = col >> (31 - __builtin_clz(a.e_gw));
    const float em = tabE[2 * g], er = tabE[2 * g + 1];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const size_t prow = row0 + r0 + p * RPP;
        const float4 y = yv[p];
        float4 v = make_float4(xv[p].x + 0.f, xv[p].y + 0.f, xv[p].z + 0.f, xv[p].w + 0.f);       // (the GEMM path adds a zero bias)
        v.x += silu_f((y.x - em) * er * eg.x + eb.x); v.y += silu_f((y.y - em) * er * eg.y + eb.y);
        v.z += silu_f((y.z - em) * er * eg.z + eb.z); v.w += silu_f((y.w - em) * er * eg.w + eb.w);
        *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) = v;
        if (a.ln_out) {
            // LayerNorm partial over 16 channels = the four lanes of a quad, summed (a0 + a1) + (a2 + a3) as the GEMM path does
            float sm = (v.x + v.y) + (v.z + v.w);
            sm += dpp_get<0xB1>(sm); sm += dpp_get<0x4E>(sm);
            const float mean = sm * (1.0f / 16.0f);
            const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
            float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            m2 += dpp_get<0xB1>(m2); m2 += dpp_get<0x4E>(m2);
            if ((c4 & 3) == 0) {
                float* o = a.ln_out + (prow * (C / 16) + (c4 >> 2)) * 2;
                o[0] = mean; o[1] = m2;
            }
        }
    }
}

// conv1x1_tail_h3_kernel<KT>: the ResnetBlock tails that carry a real GEMM -- out = res_conv(cat(x0, x1)) + bias +
// SiLU(GN(y1)) (+ LayerNorm partials), KT = 128 or 192 input channels -- on the split-fp16 products.  On the fp32 MFMA
// (conv1x1_wide_kernel) these launches were co-bound by the matrix pipe (8.6 GFLOP = 55 us at the fp32 peak for 536 MB
// of traffic); with 3 fp16 MFMAs per product the pipe needs a fifth of that and the launch is a memory stream.  Same
// 64-pixel workgroup, same transposed product (A = weights: wave w owns output channels 16 w .. of each 64-channel
// tile; B = pixels), same accumulator layout (a lane holds 4 consecutive channels of pixel (lane & 15) + 16 pb) and the
// same epilogue as conv1x1_wide_kernel<.., EPI>; the input tile is split into hi / scaled-lo fp16 planes while it is
// staged ([plane][pixel][KT halfs + 8]), B fragments are one ds_read_b128 per plane and 32-channel k-step.
// Weights: [n-tile][k-step][plane][thread = wave * 64 + lane][8 halfs] = W[n = tile*64 + wave*16 + (lane & 15)][k = 32 ks + 8 (lane >> 4) + e].
template <int KT>
__global__ __launch_bounds__(256, 2) void conv1x1_tail_h3_kernel(const Conv2dArgs a) {
    constexpr int KS = KT / 32, PITCH = KT * 2 + 16, F4 = KT / 4, NPASS = (64 * F4) / 256;
    static_assert((64 * F4) % 256 == 0, "whole passes");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][64 * PITCH];
    __shared__ float tabE[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int HWo = a.Hout * a.Wout;
    const int img = (int)(row0 / HWo);
    const Src& s0 = a.src[0];
    const int ntile = a.Npad / T2N;
    const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + tid;
    half8 wA[KS][2], wB[KS][2];
    auto load_w = [&](int it, half8 (&wv)[KS][2]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[ks][pl] = __builtin_bit_cast(half8, wbase[(((size_t)it * KS + ks) * 2 + pl) * 256]);
    };
    load_w(0, wA);
    if (a.e_y && w == 1) {
        float m, r;
        merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
        if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
    }
    // staging: all global loads first, then split + store
    float4 sv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int i = tid + 256 * p;
        const int r = i / F4, c4 = i - r * F4;
        const int cl = c4 * 4;
        const bool first = cl < s0.C || a.nsrc == 1;
        const Src& s = first ? a.src[0] : a.src[1];
        const int cs = first ? cl : cl - s0.C;
        sv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cs < s.C && (int64_t)(row0 + r) < a.rows_total) sv[p] = *reinterpret_cast<const float4*>(s.p + (row0 + r) * s.ld + cs);
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int i = tid + 256 * p;
        const int r = i / F4, c4 = i - r * F4;
        const float4 v = sv[p];
        half4v hi, lo;
        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
        lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
        *reinterpret_cast<half4v*>(&Xs[0][r * PITCH + c4 * 8]) = hi;
        *reinterpret_cast<half4v*>(&Xs[1][r * PITCH + c4 * 8]) = lo;
    }
    __syncthreads();
    const unsigned char* xh0 = &Xs[0][lq * PITCH + lg * 16];
    const unsigned char* xl0 = &Xs[1][lq * PITCH + lg * 16];
    auto tile = [&](int it, const half8 (&wv)[KS][2]) {
        const int col = it * T2N + w * 16 + lg * 4;           // this lane's 4 consecutive output channels
        f32x4 accM[4], accL[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) { accM[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const half8 xh = *reinterpret_cast<const half8*>(xh0 + pb * 16 * PITCH + ks * 64);
                const half8 xl = *reinterpret_cast<const half8*>(xl0 + pb * 16 * PITCH + ks * 64);
                accM[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xh, accM[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xl, accL[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][1], xh, accL[pb], 0, 0, 0);
            }
        if (col >= a.N) return;                               // N is a multiple of 4 (host)
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + col);
        float4 eg = make_float4(1.f, 1.f, 1.f, 1.f), eb = make_float4(0.f, 0.f, 0.f, 0.f);
        float em = 0.f, er = 1.f;
        if (a.e_y) {
            eg = *reinterpret_cast<const float4*>(a.e_gamma + col); eb = *reinterpret_cast<const float4*>(a.e_beta + col);
            const int g = col >> (31 - __builtin_clz(a.e_gw));
            em = tabE[2 * g]; er = tabE[2 * g + 1];
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const size_t prow = row0 + pb * 16 + lq;
            float4 v = make_float4((accM[pb][0] + accL[pb][0] * H3_INV) + bias.x, (accM[pb][1] + accL[pb][1] * H3_INV) + bias.y,
                                   (accM[pb][2] + accL[pb][2] * H3_INV) + bias.z, (accM[pb][3] + accL[pb][3] * H3_INV) + bias.w);
            if (a.e_y) {
                const float4 y = *reinterpret_cast<const float4*>(a.e_y + prow * a.e_ld + col);
                v.x += silu_f((y.x - em) * er * eg.x + eb.x); v.y += silu_f((y.y - em) * er * eg.y + eb.y);
                v.z += silu_f((y.z - em) * er * eg.z + eb.z); v.w += silu_f((y.w - em) * er * eg.w + eb.w);
            }
            if (a.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(a.res + prow * a.ldres + col);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) = v;
            if (a.ln_out) {
                // LayerNorm partial of this pixel over the wave's 16 channels: 4 in this lane, 4 lanes (lg) per pixel
                float sm = (v.x + v.y) + (v.z + v.w);
                sm = xsum32(xsum16(sm));
                const float mean = sm * (1.0f / 16.0f);
                const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                m2 = xsum32(xsum16(m2));
                if (lg == 0) {
                    float* o = a.ln_out + (prow * (a.Npad / 16) + it * 4 + w) * 2;
                    o[0] = mean; o[1] = m2;
                }
            }
        }
    };
    for (int it = 0; it < ntile; it += 2) {
        if (it + 1 < ntile) load_w(it + 1, wB);
        tile(it, wA);
        if (it + 2 < ntile) load_w(it + 2, wA);
        if (it + 1 < ntile) tile(it + 1, wB);
    }
}

// conv1x1_tail_h3p_kernel<KT> (round 5): the ResnetBlock tail with a res_conv -- out = res_conv(cat(x0, x1)) + bias + SiLU(GN(y1)) -- as a
// PIPELINED loop over 64-pixel tiles.  conv1x1_tail_h3_kernel is one short-lived workgroup per tile whose phases follow each other
// (x rows -> split -> barrier -> 48 MFMAs -> y1 rows -> stores): 172 us for 536 MB at 64 x 64 / 128 images = 3.1 TB/s, where the
// element-wise tail (tail_identity_kernel) moves 4.8 TB/s; five attempts to re-order its epilogue did not change that.  Here a
// workgroup owns a CONTIGUOUS run of tiles (same image for all or most of them: the GroupNorm statistics are merged once per image,
// the weights are fetched once per workgroup) and every iteration issues the NEXT tile's x rows and THIS tile's y1 rows before the
// products, so both arrive under the MFMAs and the stores of the previous tile; vmcnt retires in order: the wait for the x rows
// (issued first) never covers the younger y1 loads or stores.  Same staging, product, accumulator layout and epilogue arithmetic as
// conv1x1_tail_h3_kernel (bit-identical output).  e_y required, no residual / LayerNorm-out operand.
// NW = waves per workgroup = output channels / 16: 4 (64 channels, two workgroups per CU) or 8 (128 channels: the 32 x 32 level's
// 192 -> 128 blocks; one 512-thread workgroup per CU, every wave keeps the weights of its own 16 channels).
// (Round 5 also ran the bottleneck attention's two projections on this loop -- epilogue variants "+ residual" and "bias only" --: 54.7
// instead of 50.2 us per launch, one 512-thread workgroup per CU loses against short workgroups there; removed in round 6.)
template <int KT, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv1x1_tail_h3p_kernel(const Conv2dArgs a) {
    constexpr int KS = KT / 32, PITCH = KT * 2 + 16, F4 = KT / 4, NTH = 64 * NW, NPASS = (64 * F4) / NTH;
    static_assert((64 * F4) % NTH == 0, "whole passes");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][64 * PITCH];
    __shared__ float tabE[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    const int HWo = a.Hout * a.Wout;
    const int ntl = (int)(a.rows_total >> 6);                    // tiles in the tensor (host: whole tiles)
    const int t_lo = (int)(((long long)blockIdx.x * ntl) / gridDim.x), t_hi = (int)(((long long)(blockIdx.x + 1) * ntl) / gridDim.x);
    if (t_lo >= t_hi) return;
    const Src& s0 = a.src[0];
    const int wg = blockIdx.y * NW + w;                          // this wave's 16-channel tile of the layer
    const int col = wg * 16 + lg * 4;                            // this lane's 4 consecutive output channels
    half8 wv[KS][2];
    {
        // [n-tile of 64 channels][k-step][plane][thread within the n-tile = (wave & 3) * 64 + lane]
        const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + (size_t)(wg >> 2) * KS * 2 * 256 + (tid & 255);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[ks][pl] = __builtin_bit_cast(half8, wbase[((size_t)ks * 2 + pl) * 256]);
    }
    const bool nok = col < a.N;
    const int colc = nok ? col : 0;
    const float4 bias = a.bias ? *reinterpret_cast<const float4*>(a.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 eg = *reinterpret_cast<const float4*>(a.e_gamma + colc), eb = *reinterpret_cast<const float4*>(a.e_beta + colc);
    const int g = colc >> (31 - __builtin_clz(a.e_gw));
    float4 sv[NPASS];
    auto load_x = [&](int t) {
        const size_t row0 = (size_t)t * 64;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int i = tid + NTH * p;
            const int r = i / F4, c4 = i - r * F4;
            const int cl = c4 * 4;
            const bool first = cl < s0.C || a.nsrc == 1;
            const float* sp = first ? a.src[0].p : a.src[1].p;
            const int sld = first ? a.src[0].ld : a.src[1].ld;
            const int cs = first ? cl : cl - s0.C;
            sv[p] = *reinterpret_cast<const float4*>(sp + (row0 + r) * sld + cs);
        }
    };
    auto split_x = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int i = tid + NTH * p;
            const int r = i / F4, c4 = i - r * F4;
            const float4 v = sv[p];
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&Xs[0][r * PITCH + c4 * 8]) = hi;
            *reinterpret_cast<half4v*>(&Xs[1][r * PITCH + c4 * 8]) = lo;
        }
    };
    const unsigned char* xh0 = &Xs[0][lq * PITCH + lg * 16];
    const unsigned char* xl0 = &Xs[1][lq * PITCH + lg * 16];
    int img_have = -1;
    load_x(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
        const size_t row0 = (size_t)t * 64;
        const int img = (int)(row0 / HWo);
        if (img != img_have && w == 1) {             // (wave-uniform: img is a function of t)
            float m, r;
            merge_stats8(a.e_stats + (size_t)img * 8 * a.e_P * 2, a.e_P, a.e_cnt, lane, m, r);
            if ((lane & 7) == 0) { tabE[2 * (lane >> 3)] = m; tabE[2 * (lane >> 3) + 1] = r; }
        }
        img_have = img;
        split_x();                                               // tile t: registers -> planes
        __syncthreads();
        if (t + 1 < t_hi) load_x(t + 1);                         // the next tile's rows, then this tile's y1 rows: both under the products
        float4 yv[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) yv[pb] = *reinterpret_cast<const float4*>(a.e_y + (row0 + pb * 16 + lq) * a.e_ld + colc);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 accM[4], accL[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) { accM[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const half8 xh = *reinterpret_cast<const half8*>(xh0 + pb * 16 * PITCH + ks * 64);
                const half8 xl = *reinterpret_cast<const half8*>(xl0 + pb * 16 * PITCH + ks * 64);
                accM[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xh, accM[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xl, accL[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][1], xh, accL[pb], 0, 0, 0);
            }
        const float em = tabE[2 * g], er = tabE[2 * g + 1];
        __builtin_amdgcn_sched_barrier(0);
        if (nok) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const size_t prow = row0 + pb * 16 + lq;
                float4 v = make_float4((accM[pb][0] + accL[pb][0] * H3_INV) + bias.x, (accM[pb][1] + accL[pb][1] * H3_INV) + bias.y,
                                       (accM[pb][2] + accL[pb][2] * H3_INV) + bias.z, (accM[pb][3] + accL[pb][3] * H3_INV) + bias.w);
                const float4 y = yv[pb];
                v.x += silu_f((y.x - em) * er * eg.x + eb.x); v.y += silu_f((y.y - em) * er * eg.y + eb.y);
                v.z += silu_f((y.z - em) * er * eg.z + eb.z); v.w += silu_f((y.w - em) * er * eg.w + eb.w);
                *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) = v;
            }
        }
        __syncthreads();                                         // every wave has read the planes (and tabE) of tile t
    }
}

// conv1x1_h3p_kernel<KT, NCT, UNSHUF> (round 5): bias-only 1x1 convolutions in the pipelined form of conv1x1_tail_h3p_kernel -- a
// workgroup of 2 NCT waves loops over a contiguous run of 64-pixel output tiles (wave = 16 output channels x 32 pixels; its weights
// stay in registers), the next tile's inputs are requested before the products of the current one.
//   UNSHUF (KT = 256, NCT = 4): Downsample (:105-109) = pixel-unshuffle + 1x1 convolution (4 C -> Cout) for C = 64, Cout = 64:
//     out[oy][ox][n] = b[n] + sum_{p1, p2, c} W[n][c * 4 + p1 * 2 + p2] x[2 oy + p1][2 ox + p2][c], k = tap * 64 + c.  It ran on
//     conv2d_tile_kernel<CONV_UNSHUF> -- exact fp32 MFMA, 180 + 96 registers and 107 KB of LDS: one short-lived workgroup per CU, 88 us
//     for 168 MB; this one: ~ 40 us.
//   plain (KT = 64, NCT = 2): final_conv (64 -> channels <= 32; the pad columns of the state's row pitch are written as zeros).
// Weights: [k-step KT / 32][plane][thread = ct * 64 + lane][8 halfs] = W'[n = ct * 16 + (lane & 15)][k = 32 ks + 8 (lane >> 4) + e].
template <int KT, int NCT, bool UNSHUF>
__global__ __launch_bounds__(128 * NCT, 2) void conv1x1_h3p_kernel(const Conv2dArgs a) {
    constexpr int KS = KT / 32, PITCH = KT * 2 + 16, F4 = KT / 4, NTH = 128 * NCT, NPASS = (64 * F4) / NTH;
    static_assert((64 * F4) % NTH == 0, "whole staging passes");
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][64 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lq = lane & 15, lg = lane >> 4;
    const int HWo = a.Hout * a.Wout;
    const int ntl = (int)(a.rows_total >> 6);
    const int t_lo = (int)(((long long)blockIdx.x * ntl) / gridDim.x), t_hi = (int)(((long long)(blockIdx.x + 1) * ntl) / gridDim.x);
    if (t_lo >= t_hi) return;
    const int ct = w % NCT, ph = w / NCT;                    // channel tile of 16, pixel half of the tile
    const int col = ct * 16 + lg * 4;
    half8 wv[KS][2];
    {
        const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + ct * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wv[ks][pl] = __builtin_bit_cast(half8, wbase[((size_t)ks * 2 + pl) * (NCT * 64)]);
    }
    const bool sok = col < a.ldo;                            // columns N .. ldo - 1 (the pitch's pad): zero weights, zero bias
    const bool nok = col < a.N;                              // (N % 4 == 0, or the pad columns belong to this buffer: the host checks)
    // (element-wise with clamped addresses: N = 21 is not a multiple of 4 and a float4 at column 20 would read three floats past the
    // bias vector -- inside the weight blob only by the allocation's padding.  Pointer selected, loads unconditional.)
    const float* bp = a.bias ? a.bias : a.W;
    const int cmax = a.bias ? a.N - 1 : 0;
    const float b0 = bp[min(col, cmax)], b1 = bp[min(col + 1, cmax)], b2 = bp[min(col + 2, cmax)], b3 = bp[min(col + 3, cmax)];
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias && nok) { bias.x = b0; bias.y = col + 1 < a.N ? b1 : 0.f; bias.z = col + 2 < a.N ? b2 : 0.f; bias.w = col + 3 < a.N ? b3 : 0.f; }
    const float* src = a.src[0].p;
    const int ld = a.src[0].ld, Win = a.Win, HWi = a.Hin * a.Win, Wout = a.Wout;
    float4 sv[NPASS];
    auto load_x = [&](int t) {
        const int p0 = t * 64;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int i = tid + NTH * p;
            const int r = i / F4, c4 = i % F4;              // output pixel of the tile, float4 of its KT inputs
            if constexpr (UNSHUF) {
                const int tap = c4 >> 4, ch = (c4 & 15) * 4;
                const int pix = p0 + r, img = pix / HWo, q = pix - img * HWo;
                const int oy = q / Wout, ox = q - oy * Wout;
                sv[p] = *reinterpret_cast<const float4*>(src + ((size_t)img * HWi + (size_t)(2 * oy + (tap >> 1)) * Win + 2 * ox + (tap & 1)) * ld + ch);
            } else {
                sv[p] = *reinterpret_cast<const float4*>(src + (size_t)(p0 + r) * ld + c4 * 4);
            }
        }
    };
    auto split_x = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int i = tid + NTH * p;
            const int r = i / F4, c4 = i % F4;
            const float4 v = sv[p];
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&Xs[0][r * PITCH + c4 * 8]) = hi;
            *reinterpret_cast<half4v*>(&Xs[1][r * PITCH + c4 * 8]) = lo;
        }
    };
    const unsigned char* xh0 = &Xs[0][(ph * 32 + lq) * PITCH + lg * 16];
    const unsigned char* xl0 = &Xs[1][(ph * 32 + lq) * PITCH + lg * 16];
    load_x(t_lo);
    for (int t = t_lo; t < t_hi; ++t) {
        split_x();
        __syncthreads();
        if (t + 1 < t_hi) load_x(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 accM[2], accL[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) { accM[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[pb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const half8 xh = *reinterpret_cast<const half8*>(xh0 + pb * 16 * PITCH + ks * 64);
                const half8 xl = *reinterpret_cast<const half8*>(xl0 + pb * 16 * PITCH + ks * 64);
                accM[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xh, accM[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][0], xl, accL[pb], 0, 0, 0);
                accL[pb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][1], xh, accL[pb], 0, 0, 0);
            }
        if (sok) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const size_t prow = (size_t)t * 64 + ph * 32 + pb * 16 + lq;
                *reinterpret_cast<float4*>(a.out + prow * a.ldo + col) =
                    make_float4((accM[pb][0] + accL[pb][0] * H3_INV) + bias.x, (accM[pb][1] + accL[pb][1] * H3_INV) + bias.y,
                                (accM[pb][2] + accL[pb][2] * H3_INV) + bias.z, (accM[pb][3] + accL[pb][3] * H3_INV) + bias.w);
            }
        }
        __syncthreads();
    }
}

// y = LayerNorm_channels(x) * g from the row partials [rows][P][2] (the PreNorm of the bottleneck attention,
// model/diffusion_2d.py:82-97 / :256): one thread per float4.  A separate 20 us pass so that the qkv projection can run
// as a plain-source GEMM on conv1x1_tail_h3_kernel (the LayerNorm-on-load variant of that kernel was not reliable).
__global__ __launch_bounds__(256) void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, int P, float cnt,
                                                       const float* __restrict__ g, float* __restrict__ y, int C, int64_t n4) {
    // a thread owns FOUR float4 of one row (C / 16 threads per row, 128 contiguous bytes per group of 8 lanes and instruction), all
    // requested before the row's partials are merged -- one merge per four float4 (round 5: one float4 and one merge per thread,
    // the x load behind the merge's dependent loads, ran at 2.7 TB/s)
    const int f4 = C >> 2, lpr = f4 >> 2;                 // float4 per row, threads per row (C % 16 == 0: the host checks)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = i / lpr;
    if (row * f4 >= n4) return;
    const int l = (int)(i - row * lpr);
    const float4* xp = reinterpret_cast<const float4*>(x) + row * f4 + l;
    float4 v[4], pg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = xp[j * lpr];
#pragma unroll
    for (int j = 0; j < 4; ++j) pg[j] = *reinterpret_cast<const float4*>(g + (l + j * lpr) * 4);
    float mean, rstd;
    merge_stats(stats + row * P * 2, P, cnt, 1e-5f, mean, rstd);
    float4* yp = reinterpret_cast<float4*>(y) + row * f4 + l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 o;
        o.x = (v[j].x - mean) * rstd * pg[j].x; o.y = (v[j].y - mean) * rstd * pg[j].y;
        o.z = (v[j].z - mean) * rstd * pg[j].z; o.w = (v[j].w - mean) * rstd * pg[j].w;
        yp[j * lpr] = o;
    }
}

// Merge the per-tile GroupNorm partials of an image: [NI][8][tpi][2] -> [NI][8][2] = (mean, M2) over the whole
// (image, group); every tile holds 64*gw elements.  Chan's formula, fixed order.
__global__ void gn_merge_kernel(const float* __restrict__ part, float* __restrict__ merged, int n_stats, int tpi, int gw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_stats) return;
    const float* p = part + (size_t)i * tpi * 2;
    const float nb = (float)(T2M * gw);
    float n = 0.f, mean = 0.f, M2 = 0.f;
    for (int t = 0; t < tpi; ++t) {
        const float d = p[2 * t] - mean;
        const float nn = n + nb;
        mean += d * (nb / nn);
        M2 += p[2 * t + 1] + d * d * (n * nb / nn);
        n = nn;
    }
    merged[2 * i] = mean; merged[2 * i + 1] = M2;
}

// ---------------------------------------------------------------------------------------------------------
// LinearAttention (model/diffusion_2d.py:239-254) on qkv [rows, 384] (q | k | v, heads*32 each), per image of n pixels:
//   q = softmax over d (per pixel, per head) * 32^-1/2 ; k = softmax over the n pixels ; v /= n ;
//   ctx[d][e] = sum_n k[d][n] v[e][n] ; out[e][n] = sum_d ctx[d][e] q[d][n].
// Kernel 1: partial contexts over a slice of the pixels with the slice's OWN softmax shift (online softmax over
// slices): per (image, head, slice): M_s[d] = max over the slice, S_s[d] = sum exp(k - M_s), ctxp[d][e] = sum
// exp(k - M_s) v[e].  The slices are merged (rescaled by exp(M_s - M)) in kernel 2's prologue -- no separate pass over
// k for the global column maximum.
constexpr int LA_SPLIT = 16;
__global__ __launch_bounds__(256) void la_context_kernel(const float* __restrict__ qkv, float* __restrict__ kst,
                                                         float* __restrict__ ctxp, int n) {
    __shared__ float P[256 * 32];
    __shared__ float smx[8][32], ssm[8][32];
    const int split = blockIdx.x % LA_SPLIT, ih = blockIdx.x / LA_SPLIT;       // ih = img*4 + h
    const int img = ih >> 2, h = ih & 3;
    const int d = threadIdx.x & 31, eg = threadIdx.x >> 5;                        // e in [eg*4, eg*4+4)
    const int per = n / LA_SPLIT;                                                 // <= 256
    const float* base = qkv + ((size_t)img * n + split * per) * 384;
    const float* kp = base + 128 + h * 32 + d;
    // slice maximum of column d (thread group eg takes pixels eg, eg+8, ...), the k values kept in registers
    float kv[32];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int i = eg + 8 * j;
        kv[j] = (i < per) ? kp[(size_t)i * 384] : -INFINITY;
        mx = fmaxf(mx, kv[j]);
    }
    smx[eg][d] = mx;
    __syncthreads();
    float M = smx[0][d];
#pragma unroll
    for (int j = 1; j < 8; ++j) M = fmaxf(M, smx[j][d]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int i = eg + 8 * j;
        if (i < per) { const float p = expf(kv[j] - M); P[i * 32 + d] = p; sum += p; }
    }
    ssm[eg][d] = sum;
    __syncthreads();
    if (eg == 0) {
        float S = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) S += ssm[j][d];
        float* o = kst + (((size_t)ih * LA_SPLIT + split) * 32 + d) * 2;
        o[0] = M; o[1] = S;
    }
    // v slice -> LDS with coalesced float4 loads (issued together), then the context loop runs out of LDS only
    __shared__ __attribute__((aligned(16))) float V[256 * 32];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int i = threadIdx.x + 256 * p;                 // per x 8 float4
        const int px = i >> 3, c4 = i & 7;
        if (px < per) *reinterpret_cast<float4*>(&V[px * 32 + c4 * 4]) = *reinterpret_cast<const float4*>(base + (size_t)px * 384 + 256 + h * 32 + c4 * 4);
    }
    __syncthreads();
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll 8
    for (int i = 0; i < per; ++i) {
        const float p = P[i * 32 + d];
        const float4 v = *reinterpret_cast<const float4*>(&V[i * 32 + eg * 4]);
        c0 += p * v.x; c1 += p * v.y; c2 += p * v.z; c3 += p * v.w;
    }
    float* o = ctxp + (((size_t)ih * LA_SPLIT + split) * 32 + d) * 32 + eg * 4;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// Kernel 2: out[pixel][h*32 + e] = sum_d ctx[d][e] * softmax_d(q[pixel][h*32 + :])[d] * 32^-1/2 ; one thread per (pixel, head).
__global__ __launch_bounds__(256) void la_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ kst,
                                                       const float* __restrict__ ctxp, float* __restrict__ att, int n) {
    __shared__ float ctx[32 * 33];
    __shared__ float wgt[LA_SPLIT * 32];
    const int ih = blockIdx.y, img = ih >> 2, h = ih & 3;
    if (threadIdx.x < 32) {
        // merge the slices' softmax statistics of column d: weight of slice s = exp(M_s - M) / (S * n)
        const int d = threadIdx.x;
        const float* st = kst + ((size_t)ih * LA_SPLIT * 32 + d) * 2;
        float M = st[0];
        for (int sp = 1; sp < LA_SPLIT; ++sp) M = fmaxf(M, st[sp * 64]);
        float S = 0.f;
        for (int sp = 0; sp < LA_SPLIT; ++sp) { const float e = expf(st[sp * 64] - M); wgt[sp * 32 + d] = e; S += st[sp * 64 + 1] * e; }
        const float inv = 1.0f / (S * (float)n);
        for (int sp = 0; sp < LA_SPLIT; ++sp) wgt[sp * 32 + d] *= inv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        float s = 0.f;
        for (int sp = 0; sp < LA_SPLIT; ++sp) s += ctxp[((size_t)ih * LA_SPLIT + sp) * 1024 + i] * wgt[sp * 32 + (i >> 5)];
        ctx[(i >> 5) * 33 + (i & 31)] = s;
    }
    __syncthreads();
    // out^T[e][px] = sum_d ctx[d][e] * qhat[px][d] on fp32 MFMA: A = ctx^T (resident fragments), B = the softmaxed q of 16
    // pixels; lane (lq = lane & 15, lg = lane >> 4) owns pixel lq of the block and the 8 channels d = 8 lg .. 8 lg + 7
    // (k-step j pairs d = 8 lg + j on both operands -- a contraction does not care about the order), so q is read with
    // two float4 loads per lane and the result leaves as float4 runs of 4 consecutive channels.
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lq = lane & 15, lg = lane >> 4;
    float cf[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) cf[j][eb] = ctx[(lg * 8 + j) * 33 + eb * 16 + lq];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int pix = blockIdx.x * 256 + (w * 4 + blk) * 16 + lq;
        const bool ok = pix < n;
        const float* qp = qkv + ((size_t)img * n + (ok ? pix : 0)) * 384 + h * 32 + lg * 8;
        const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
        float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        float mx = q[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, q[j]);
        mx = xmax32(xmax16(mx));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { q[j] = expf(q[j] - mx); sum += q[j]; }
        sum = xsum32(xsum16(sum));
        const float sc = 0.17677669529663687f / sum;
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float qv = q[j] * sc;
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) acc[eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[j][eb], qv, acc[eb], 0, 0, 0);
        }
        if (ok) {
            float* op = att + ((size_t)img * n + pix) * 128 + h * 32 + lg * 4;
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
                *reinterpret_cast<float4*>(op + eb * 16) = make_float4(acc[eb][0], acc[eb][1], acc[eb][2], acc[eb][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Residual(PreNorm(LinearAttention)) without the [pixels, 384] q|k|v tensor (model/diffusion_2d.py:226-254):
//   la2d_context_kernel   per (image, 128-pixel slice), wave = head: y = LN(x) g as split-fp16 planes, k|v of the
//                         head = y Wkv^T (rows = pixels) on the fp16 MFMA, slice softmax of k over pixels, partial
//                         context k^ v^T on the fp32 MFMA (its operands are the k, v accumulators)
//   la2d_merge_kernel     slices -> ctx[image][head][32][32], normalised, / (h w)
//   la2d_apply_out_kernel per (image, 64 pixels): y again, q = Wq y^T, softmax over d, * 32^-1/2, att = ctx^T q,
//                         z = Wo att + b, out = LN(z) g2 + x
// HBM traffic per site: x twice, out once (the unfused path moved 7 x that).  Projection operands are the fragments
// of attn1d_site_h3_kernel ([tile][k32][plane][lane][8 halfs]).
struct La2dArgs {
    const float* x; int ldx;       // [NI * HW, C]
    float* out; int ldo;
    const float* g;                // PreNorm gain [C]
    const float* g2;               // to_out LayerNorm gain [C]
    const float* Wqkv;             // 24 tiles x C
    const float* Wo;               // C/16 tiles x 128
    const float* bo;               // [C]
    float* part;                   // [NI*4][nsplit][64 + 1024]: slice max | slice sum | unnormalised context
    float* ctx;                    // [NI*4][1024]
    int HW, nsplit;                // nsplit = context workgroups (records) per image
    int spw, tpw;                  // 128-pixel slices per context workgroup ; 64-pixel tiles per apply workgroup
    int dbg;                       // timing ablations (wrong results)
};
constexpr int LA2_PX = 128;        // pixels per context slice in rounds 2 - 4 at C = 64; C = 128 used 64 (half the k|v registers, measured 108 -> 61 us).
                                   // Round 5: 64 pixels at C = 64 too -- with the split-fp16 context product (H16) the kernel then fits 244 registers,
                                   // i.e. TWO workgroups per CU: one's LayerNorm / softmax VALU phases run under the other's products
constexpr int LA2_REC = 64 + 1024;
constexpr int la2_px(int C) { return LA2_PX / 2; }

// LayerNorm over C channels (biased variance, eps 1e-5) * g of the NPX tile rows held in xr (thread (lrow, lcol) owns
// columns 4*lcol.. of rows r*RPP + lrow) -> (hi, scaled lo) fp16 planes
template <int C, int NPX>
struct LnTile {
    static constexpr int LPR = C / 4, RPP = 256 / LPR, NPASS = NPX / RPP, YPB = 2 * C + 16;
    __device__ static __forceinline__ void load(float4 (&xr)[NPASS], const float* __restrict__ x0, int ldx, int tid) {
        const int lrow = tid / LPR, lcol = tid % LPR;
#pragma unroll
        for (int r = 0; r < NPASS; ++r) xr[r] = *reinterpret_cast<const float4*>(x0 + (size_t)(r * RPP + lrow) * ldx + 4 * lcol);
    }
    __device__ static __forceinline__ void to_planes(const float4 (&xr)[NPASS], const float4 gv, unsigned char* Yh, unsigned char* Yl, int tid) {
        const int lrow = tid / LPR, lcol = tid % LPR;
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const float s1 = rowgroup_sum<LPR>((xr[r].x + xr[r].y) + (xr[r].z + xr[r].w));
            const float mean = s1 * (1.0f / C);
            const float d0 = xr[r].x - mean, d1 = xr[r].y - mean, d2 = xr[r].z - mean, d3 = xr[r].w - mean;
            const float s2 = rowgroup_sum<LPR>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            const float y0 = d0 * rstd * gv.x, y1 = d1 * rstd * gv.y, y2 = d2 * rstd * gv.z, y3 = d3 * rstd * gv.w;
            half4v hi, lo;
            hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
            lo[0] = (_Float16)((y0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((y2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y3 - (float)hi[3]) * H3_SCALE);
            const int off = (r * RPP + lrow) * YPB + 8 * lcol;
            *reinterpret_cast<half4v*>(Yh + off) = hi;
            *reinterpret_cast<half4v*>(Yl + off) = lo;
        }
    }
};

// One workgroup = a.spw consecutive 128-pixel slices of one image (the next slice's rows are in flight while the
// current one is processed); running maximum / sum / context are rescaled slice by slice (online softmax), one record
// per workgroup.
// H16 (round 5): the per-head product ctx += k^T v on the split-fp16 MFMA instead of the exact fp32 one.  The fp32 form was 128
// v_mfma_f32_16x16x4_f32 per 128-pixel slice and wave = 4096 matrix-pipe cycles next to the 3264 of the whole k | v projection (192
// v_mfma_f32_16x16x32_f16): 55 % of the kernel's matrix time for 7 % of its FLOPs.  Two pixel tiles of the softmaxed k (v)
// accumulators ARE one A (B) operand of a 16x16x32 step -- lane (channel lr, pixels 8 lq + e) holds eight of its channel's 32 pixels,
// the same pixel order on both sides -- so the product is 48 MFMAs (3 per (pixel-tile pair, d tile, e tile)): 816 cycles, plus the
// hi / lo splits of k and v (VALU).  exp(k - max) is in (0, 1] and v is a projection of LayerNorm output: both inside the range
// rule of the split; two fp32 accumulator sets as everywhere.
template <int C, bool H16 = true>
__global__ __launch_bounds__(256, C == 64 ? 2 : 1) void la2d_context_kernel(const La2dArgs a) {
    constexpr int PX = la2_px(C);
    using LN = LnTile<C, PX>;
    constexpr int K32 = C / 32, YPB = 2 * C + 16, NTL = PX / 16;
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][PX * YPB];
    __shared__ float fac[4][32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int img = blockIdx.x / a.nsplit, split = blockIdx.x % a.nsplit;
    const float* x0 = a.x + ((size_t)img * a.HW + (size_t)split * a.spw * PX) * a.ldx;
    float4 xr[LN::NPASS];
    LN::load(xr, x0, a.ldx, tid);
    const float4 gv = *reinterpret_cast<const float4*>(a.g + 4 * (tid % LN::LPR));
    // k | v fragments of head w, resident: tiles 8 + 2w, 9 + 2w (k), 16 + 2w, 17 + 2w (v)
    const float4* W4 = reinterpret_cast<const float4*>(a.Wqkv);
    half8 wh[4][K32], wl[4][K32];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int tile = 8 + (s >> 1) * 8 + 2 * w + (s & 1);
#pragma unroll
        for (int k = 0; k < K32; ++k) {
            wh[s][k] = __builtin_bit_cast(half8, W4[(((size_t)tile * K32 + k) * 2 + 0) * 64 + lane]);
            wl[s][k] = __builtin_bit_cast(half8, W4[(((size_t)tile * K32 + k) * 2 + 1) * 64 + lane]);
        }
    }
    float run_m[2] = {-INFINITY, -INFINITY}, run_s[2] = {0.f, 0.f};       // per channel d = dt*16 + lr
    f32x4 ctx[2][2];                                                       // rows d = dt*16 + lq*4 + i, cols e = et*16 + lr
    f32x4 ctxL[2][2];                                                      // H16: the low-order accumulator set (x 2^11)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int et = 0; et < 2; ++et) { ctx[dt][et] = f32x4{0.f, 0.f, 0.f, 0.f}; ctxL[dt][et] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int sl = 0; sl < a.spw; ++sl) {
        LN::to_planes(xr, gv, Yp[0], Yp[1], tid);
        __syncthreads();
        if (sl + 1 < a.spw) LN::load(xr, x0 + (size_t)(sl + 1) * PX * a.ldx, a.ldx, tid);
        // k, v: rows = pixels (8 tiles of 16), cols = the head's 32 channels.  H16: only k here -- v is projected pair of tiles by
        // pair of tiles where the context product consumes it (below), so that 2 instead of NTL v tiles are live
        f32x4 kk[NTL][2], vv[H16 ? 2 : NTL][2];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            constexpr int NS = H16 ? 2 : 4;
            f32x4 M[NS], Lo[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { M[s] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int k = 0; k < K32; ++k) {
                const int off = (nt * 16 + lr) * YPB + k * 64 + lq * 16;
                const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
                const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    M[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[s][k], M[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[s][k], Lo[s], 0, 0, 0);
                    Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[s][k], Lo[s], 0, 0, 0);
                }
            }
            kk[nt][0] = M[0] + Lo[0] * H3_INV; kk[nt][1] = M[1] + Lo[1] * H3_INV;
            if constexpr (!H16) { vv[nt][0] = M[2] + Lo[2] * H3_INV; vv[nt][1] = M[3] + Lo[3] * H3_INV; }
        }
        // online softmax of k over pixels, per channel d = dt*16 + lr (pixels: rows nt*16 + lq*4 + i)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, kk[nt][dt][i]);
            mx = xmax32(xmax16(mx));
            const float mnew = fmaxf(run_m[dt], mx);
            const float f = __builtin_amdgcn_exp2f((run_m[dt] - mnew) * 1.4426950408889634f);      // 0 on the first slice
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __builtin_amdgcn_exp2f((kk[nt][dt][i] - mnew) * 1.4426950408889634f);
                    kk[nt][dt][i] = e;
                    sum += e;
                }
            sum = xsum32(xsum16(sum));
            run_s[dt] = run_s[dt] * f + sum;
            run_m[dt] = mnew;
            if (lq == 0) fac[w][dt * 16 + lr] = f;
        }
        if constexpr (!H16) __syncthreads();              // fac visible; every wave is done reading the planes
        // (H16: fac[w] is written and read by wave w alone -- LDS operations of one wave complete in order -- and the planes are read
        // again by the v projection below: the workgroup barrier sits at the end of the iteration)
        // ctx = ctx * f[d] + sum_pixels e^(k - m) v
        if constexpr (H16) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float4 f4 = *reinterpret_cast<const float4*>(&fac[w][dt * 16 + lq * 4]);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    ctx[dt][et][0] *= f4.x; ctx[dt][et][1] *= f4.y; ctx[dt][et][2] *= f4.z; ctx[dt][et][3] *= f4.w;
                    ctxL[dt][et][0] *= f4.x; ctxL[dt][et][1] *= f4.y; ctxL[dt][et][2] *= f4.z; ctxL[dt][et][3] *= f4.w;
                }
            }
#pragma unroll
            for (int nt = 0; nt < NTL; nt += 2) {
                // v of pixel tiles nt, nt + 1
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 M[2], Lo[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) { M[s] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int k = 0; k < K32; ++k) {
                        const int off = ((nt + u) * 16 + lr) * YPB + k * 64 + lq * 16;
                        const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
                        const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            M[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[2 + s][k], M[s], 0, 0, 0);
                            Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[2 + s][k], Lo[s], 0, 0, 0);
                            Lo[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[2 + s][k], Lo[s], 0, 0, 0);
                        }
                    }
                    vv[u][0] = M[0] + Lo[0] * H3_INV; vv[u][1] = M[1] + Lo[1] * H3_INV;
                }
                half8 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float kv = kk[nt + (e >> 2)][t][e & 3], v_ = vv[e >> 2][t][e & 3];
                        kh[t][e] = (_Float16)kv; kl[t][e] = (_Float16)((kv - (float)kh[t][e]) * H3_SCALE);
                        vh[t][e] = (_Float16)v_; vl[t][e] = (_Float16)((v_ - (float)vh[t][e]) * H3_SCALE);
                    }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
                        ctx[dt][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[dt], vh[et], ctx[dt][et], 0, 0, 0);
                        ctxL[dt][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[dt], vl[et], ctxL[dt][et], 0, 0, 0);
                        ctxL[dt][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[dt], vh[et], ctxL[dt][et], 0, 0, 0);
                    }
            }
            __syncthreads();                              // every wave is done reading the planes
        } else
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float4 f4 = *reinterpret_cast<const float4*>(&fac[w][dt * 16 + lq * 4]);
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4 c0 = ctx[dt][et], c1 = f32x4{0.f, 0.f, 0.f, 0.f};
                c0[0] *= f4.x; c0[1] *= f4.y; c0[2] *= f4.z; c0[3] *= f4.w;
#pragma unroll
                for (int nt = 0; nt < NTL; nt += 2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[nt][dt][i], vv[nt][et][i], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[nt + 1][dt][i], vv[nt + 1][et][i], c1, 0, 0, 0);
                    }
                ctx[dt][et] = c0 + c1;
            }
        }
    }
    float* rec = a.part + ((size_t)(img * 4 + w) * a.nsplit + split) * LA2_REC;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        if (lq == 0) { rec[dt * 16 + lr] = run_m[dt]; rec[32 + dt * 16 + lr] = run_s[dt]; }
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int i = 0; i < 4; ++i) rec[64 + (dt * 16 + lq * 4 + i) * 32 + et * 16 + lr] = H16 ? ctx[dt][et][i] + ctxL[dt][et][i] * H3_INV : ctx[dt][et][i];
    }
}

__global__ __launch_bounds__(256) void la2d_merge_kernel(const float* __restrict__ part, float* __restrict__ ctx, int nsplit, int HW) {
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
        const float inv = 1.0f / (S * (float)HW);         // softmax denominator and v / (h w)
        for (int sp = 0; sp < nsplit; ++sp) wgt[sp * 32 + d] *= inv;
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

// One workgroup = a.tpw consecutive 64-pixel tiles of one image; weights and the head contexts stay in registers, the
// next tile's rows are in flight while the current one is processed.
// H16 (round 5): att = ctx^T q on the split-fp16 MFMA: the contraction runs over the head's 32 channels = ONE 16x16x32 step, the
// context fragments (times h w, a power of two: the merged context carries v / (h w) and would sit in fp16's subnormals) are split
// once per workgroup, q per pixel tile: 24 MFMAs per 64-pixel tile instead of 64 fp32 ones (2048 -> 408 matrix-pipe cycles).
template <int C, bool H16 = true>
__global__ __launch_bounds__(256, C == 64 ? 3 : 1) void la2d_apply_out_kernel(const La2dArgs a) {
    using LN = LnTile<C, 64>;
    constexpr int K32 = C / 32, YPB = 2 * C + 16, APB = 2 * 128 + 16, NPX = 64, NTL = 4, CT = C / 16, TPW = CT / 4, ZP = C + 4;
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][NPX * YPB];
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][NPX * APB];
    static_assert(NPX * ZP * 4 <= 2 * NPX * YPB, "Z aliases the y planes");
    float* Z = reinterpret_cast<float*>(&Yp[0][0]);       // the y planes are dead once every wave has its q
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int wpi = a.HW / (NPX * a.tpw);                 // workgroups per image
    const int img = blockIdx.x / wpi, t0 = (blockIdx.x % wpi) * a.tpw;
    const size_t row00 = (size_t)img * a.HW + (size_t)t0 * NPX;
    float4 xr[LN::NPASS];
    LN::load(xr, a.x + row00 * a.ldx, a.ldx, tid);
    const int lcol = tid % LN::LPR, lrow = tid / LN::LPR;
    const float4 gv = *reinterpret_cast<const float4*>(a.g + 4 * lcol);
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
    // merged context of head w as A fragments of the second product: lane holds ctx[d = dt*16 + lq*4 + i][e = et*16 + lr]
    float cf[2][2][4];
    {
        const float* cp = a.ctx + (size_t)(img * 4 + w) * 1024;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int i = 0; i < 4; ++i) cf[dt][et][i] = cp[(dt * 16 + lq * 4 + i) * 32 + et * 16 + lr];
    }
    // H16: A fragments of att = ctx^T q: lane (row e = lr of tile et, k-slots 8 lq + 4 dt + i <-> d = dt*16 + lq*4 + i)
    const bool pow2 = (a.HW & (a.HW - 1)) == 0;
    const float csc = pow2 ? (float)a.HW : 1.0f, cinv = 1.0f / csc;       // exact power of two
    half8 ch[2], cl[2];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v_ = cf[e >> 2][et][e & 3] * csc;
            ch[et][e] = (_Float16)v_; cl[et][e] = (_Float16)((v_ - (float)ch[et][e]) * H3_SCALE);
        }
#pragma unroll 1
    for (int tt = 0; tt < a.tpw; ++tt) {
        const size_t row0 = row00 + (size_t)tt * NPX;
        // the tile's own rows, for the residual.  C = 64: NOT kept -- read again at the end (an L2 hit a few microseconds after the
        // first read): 16 registers less, which is what lets THREE workgroups share a CU (168 registers, 3 x 52 KB of LDS)
        constexpr bool RELOAD = C == 64;
        float4 xres[RELOAD ? 1 : LN::NPASS];
        if constexpr (!RELOAD) {
#pragma unroll
            for (int r = 0; r < LN::NPASS; ++r) xres[r] = xr[r];
        }
        LN::to_planes(xr, gv, Yp[0], Yp[1], tid);
        __syncthreads();
        if (tt + 1 < a.tpw) LN::load(xr, a.x + (row0 + NPX) * a.ldx, a.ldx, tid);
        // q of head w: rows = channels d, cols = pixels ; softmax over d ; * 32^-1/2 ; att = ctx^T q
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
            for (int dt = 0; dt < 2; ++dt) q[dt] = (q[dt] * inv) * 0.17677669529663687f;
            half8 qhh, qll;
            if constexpr (H16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v_ = q[e >> 2][e & 3];
                    qhh[e] = (_Float16)v_; qll[e] = (_Float16)((v_ - (float)qhh[e]) * H3_SCALE);
                }
            }
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (H16) {
                    f32x4 oL = f32x4{0.f, 0.f, 0.f, 0.f};
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[et], qhh, o, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[et], qll, oL, 0, 0, 0);
                    oL = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[et], qhh, oL, 0, 0, 0);
                    o = (o + oL * H3_INV) * cinv;
                } else {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_16x16x4f32(cf[dt][et][i], q[dt][i], o, 0, 0, 0);
                }
                half4v hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)o[i]; lo[i] = (_Float16)((o[i] - (float)hi[i]) * H3_SCALE); }
                const int off = (nt * 16 + lr) * APB + 2 * (w * 32 + et * 16 + lq * 4);
                *reinterpret_cast<half4v*>(&Ap[0][off]) = hi;
                *reinterpret_cast<half4v*>(&Ap[1][off]) = lo;
            }
        }
        __syncthreads();
        // z = Wo att + bo : channel tiles [w*TPW, (w+1)*TPW) of this wave, all 64 pixels -> Z[pixel][channel]
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
        __syncthreads();
        // out = LayerNorm(z) g2 + x
        float4 xrl[RELOAD ? LN::NPASS : 1];
        if constexpr (RELOAD) LN::load(xrl, a.x + row0 * a.ldx, a.ldx, tid);
#pragma unroll
        for (int r = 0; r < LN::NPASS; ++r) {
            const int n = r * LN::RPP + lrow;
            const float4 xv = RELOAD ? xrl[RELOAD ? r : 0] : xres[RELOAD ? 0 : r];
            const float4 zv = *reinterpret_cast<const float4*>(&Z[n * ZP + 4 * lcol]);
            const float s1 = rowgroup_sum<LN::LPR>((zv.x + zv.y) + (zv.z + zv.w));
            const float mean = s1 * (1.0f / C);
            const float d0 = zv.x - mean, d1 = zv.y - mean, d2 = zv.z - mean, d3 = zv.w - mean;
            const float s2 = rowgroup_sum<LN::LPR>((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            float4 o4;
            o4.x = d0 * rstd * gv2.x + xv.x; o4.y = d1 * rstd * gv2.y + xv.y;
            o4.z = d2 * rstd * gv2.z + xv.z; o4.w = d3 * rstd * gv2.w + xv.w;
            *reinterpret_cast<float4*>(a.out + (row0 + n) * a.ldo + 4 * lcol) = o4;
        }
        __syncthreads();                                  // Z (= the y planes) is rewritten by the next tile
    }
}

// Full softmax attention (Attention.forward, :266-278) for n tokens, 4 heads x 32, on fp32 MFMA.  One workgroup per
// (image, head, 64 queries), one wave per 16 queries; keys/values streamed through LDS in chunks of 64 (double
// buffered).  Both products are computed TRANSPOSED so that the probabilities never leave registers:
//   S^T[key][q] = K[key][:] . Q^T[:][q]       (A = K from LDS, B = Q^T held in registers, pre-scaled by 32^-1/2 log2 e)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]   (A = V from LDS, B = P^T = the S^T accumulators after exp2)
// The k index of the second product enumerates keys in the accumulator layout's order (16*mb + 4*(lane>>4) + rg), which
// a contraction does not care about.  Online softmax per query column (lane & 15): the running max is shared by the
// four 16-lane groups (two cross-lane exchanges per chunk), the running sum stays a per-lane partial until the end.
__global__ __launch_bounds__(256, 2) void attn_full_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n) {
    constexpr int LDK = 36;
    __shared__ __attribute__((aligned(16))) float Ks[2][64 * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[2][64 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-aware mapping: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the query
    // blocks of one (image, head) are gathered on one XCD and its K/V is fetched from HBM once instead of 8 times
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.y;
            const int j = (lin & 7) * (total >> 3) + (lin >> 3);
            by = j / (int)gridDim.x; bx = j - by * (int)gridDim.x;
        }
    }
    const int ih = by, img = ih >> 2, h = ih & 3;
    const int q0 = bx * 64 + w * 16;
    const float* base = qkv + (size_t)img * n * 384;
    const int lq = lane & 15, lg = lane >> 4;
    float qb[8];
    {
        const float* qp = base + (size_t)(q0 + lq) * 384 + h * 32 + lg;
        const float sc = 0.17677669529663687f * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qb[ks] = qp[ks * 4] * sc;
    }
    const int sr = tid >> 3, sc4 = tid & 7;
    float4 kr0, kr1, vr0, vr1;
    auto load_kv = [&](int k0) {
        const float* rp = base + (size_t)(k0 + sr) * 384 + h * 32 + sc4 * 4;
        kr0 = *reinterpret_cast<const float4*>(rp + 128);
        vr0 = *reinterpret_cast<const float4*>(rp + 256);
        kr1 = *reinterpret_cast<const float4*>(rp + 32 * 384 + 128);
        vr1 = *reinterpret_cast<const float4*>(rp + 32 * 384 + 256);
    };
    auto store_kv = [&](int buf) {
        *reinterpret_cast<float4*>(&Ks[buf][sr * LDK + sc4 * 4]) = kr0;
        *reinterpret_cast<float4*>(&Vs[buf][sr * LDK + sc4 * 4]) = vr0;
        *reinterpret_cast<float4*>(&Ks[buf][(sr + 32) * LDK + sc4 * 4]) = kr1;
        *reinterpret_cast<float4*>(&Vs[buf][(sr + 32) * LDK + sc4 * 4]) = vr1;
    };
    load_kv(0);
    store_kv(0);
    __syncthreads();
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float m = -INFINITY, l = 0.f;
    const int nchunk = n >> 6;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        load_kv(min(c + 1, nchunk - 1) << 6);
        __builtin_amdgcn_sched_barrier(0);
        const float* Kb = &Ks[buf][lq * LDK + lg];
        f32x4 s[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            s[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                s[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Kb[mb * 16 * LDK + ks * 4], qb[ks], s[mb], 0, 0, 0);
        }
        float pr[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { pr[mb][0] = s[mb][0]; pr[mb][1] = s[mb][1]; pr[mb][2] = s[mb][2]; pr[mb][3] = s[mb][3]; }
        float mx = pr[0][0];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, pr[mb][rg]);
        mx = xmax32(xmax16(mx));
        const float mn = fmaxf(m, mx);
        const float corr = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) { pr[mb][rg] = __builtin_amdgcn_exp2f(pr[mb][rg] - mn); ps += pr[mb][rg]; }
        l = l * corr + ps;
#pragma unroll
        for (int db = 0; db < 2; ++db) { o[db][0] *= corr; o[db][1] *= corr; o[db][2] *= corr; o[db][3] *= corr; }
        const float* Vb = &Vs[buf][lg * 4 * LDK + lq];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vb[(mb * 16 + rg) * LDK + db * 16], pr[mb][rg], o[db], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_kv(buf ^ 1);
        __syncthreads();
    }
    l = xsum32(xsum16(l));
    const float inv = 1.0f / l;
    float* op = out + ((size_t)img * n + q0 + lq) * 128 + h * 32 + lg * 4;
#pragma unroll
    for (int db = 0; db < 2; ++db)
        *reinterpret_cast<float4*>(op + db * 16) = make_float4(o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv);
}

// attn_full_h3_kernel: the same attention with both products on the split-fp16 scheme (three v_mfma_f32_16x16x32_f16 per
// fp32-equivalent product, fp32 accumulation in two accumulator sets).  K is staged as (hi, scaled lo) planes
// [key][32 d]; V is staged TRANSPOSED, [d][key slot], with the 64 keys of a chunk permuted so that the eight k-slots a
// lane feeds to one MFMA -- keys 16*mbA + 4*lg + {0..3} and 16*mbB + 4*lg + {0..3}, the order in which the S^T
// accumulators hold the probabilities -- are contiguous: P goes from the accumulators into the second product with
// no data movement.  The exponentials and the fp16 splits of P are now the dominant work (VALU), not the MFMAs.
__global__ __launch_bounds__(256, 2) void attn_full_h3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n) {
    constexpr int KPB = 80, VPB = 144;                   // bytes per key row (32 d) / per d row (64 key slots), padded
    __shared__ __attribute__((aligned(16))) unsigned char Kp[2][2][64 * KPB];     // [buffer][plane]
    __shared__ __attribute__((aligned(16))) unsigned char Vp[2][2][32 * VPB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.y;
            const int j = (lin & 7) * (total >> 3) + (lin >> 3);
            by = j / (int)gridDim.x; bx = j - by * (int)gridDim.x;
        }
    }
    const int ih = by, img = ih >> 2, h = ih & 3;
    const int q0 = bx * 64 + w * 16;
    const float* base = qkv + (size_t)img * n * 384;
    const int lq = lane & 15, lg = lane >> 4;
    // Q^T fragment of this wave's 16 queries: lane (query lq, d = lg*8 .. +7), pre-scaled by 32^-1/2 log2 e
    half8 qh, ql;
    {
        const float* qp = base + (size_t)(q0 + lq) * 384 + h * 32 + lg * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(qp), a1 = *reinterpret_cast<const float4*>(qp + 4);
        const float sc = 0.17677669529663687f * 1.4426950408889634f;
        const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[e] = (_Float16)v[e]; ql[e] = (_Float16)((v[e] - (float)qh[e]) * H3_SCALE); }
    }
    const int sr = tid >> 3, sc4 = tid & 7;              // staging: keys sr, sr + 32 ; d = sc4*4 .. +3
    float4 kr0, kr1, vr0, vr1;
    auto load_kv = [&](int k0) {
        const float* rp = base + (size_t)(k0 + sr) * 384 + h * 32 + sc4 * 4;
        kr0 = *reinterpret_cast<const float4*>(rp + 128);
        vr0 = *reinterpret_cast<const float4*>(rp + 256);
        kr1 = *reinterpret_cast<const float4*>(rp + 32 * 384 + 128);
        vr1 = *reinterpret_cast<const float4*>(rp + 32 * 384 + 256);
    };
    // key -> k-slot of the second product: step (key / 32), then lg = (key % 16) / 4, half = (key / 16) % 2, rg = key % 4
    auto slot_of = [](int key) { return (key >> 5) * 32 + ((key & 15) >> 2) * 8 + ((key >> 4) & 1) * 4 + (key & 3); };
    const int slot0 = slot_of(sr), slot1 = slot_of(sr + 32);
    auto store_kv = [&](int buf) {
        auto put_k = [&](const float4 v, int key) {
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&Kp[buf][0][key * KPB + sc4 * 8]) = hi;
            *reinterpret_cast<half4v*>(&Kp[buf][1][key * KPB + sc4 * 8]) = lo;
        };
        auto put_v = [&](const float4 v, int slot) {
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const _Float16 hi = (_Float16)f[j];
                const _Float16 lo = (_Float16)((f[j] - (float)hi) * H3_SCALE);
                *reinterpret_cast<_Float16*>(&Vp[buf][0][(sc4 * 4 + j) * VPB + slot * 2]) = hi;
                *reinterpret_cast<_Float16*>(&Vp[buf][1][(sc4 * 4 + j) * VPB + slot * 2]) = lo;
            }
        };
        put_k(kr0, sr); put_k(kr1, sr + 32);
        put_v(vr0, slot0); put_v(vr1, slot1);
    };
    load_kv(0);
    store_kv(0);
    __syncthreads();
    f32x4 oM[2], oL[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) { oM[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; oL[db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float m = -INFINITY, l = 0.f;
    const int nchunk = n >> 6;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        load_kv(min(c + 1, nchunk - 1) << 6);
        __builtin_amdgcn_sched_barrier(0);
        // S^T tiles: rows = keys 16*mb + 4*lg + rg, cols = queries
        float pr[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int off = (mb * 16 + lq) * KPB + lg * 16;
            const half8 kh = *reinterpret_cast<const half8*>(&Kp[buf][0][off]);
            const half8 kl = *reinterpret_cast<const half8*>(&Kp[buf][1][off]);
            f32x4 sM = (f32x4){0.f, 0.f, 0.f, 0.f}, sL = (f32x4){0.f, 0.f, 0.f, 0.f};
            sM = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh, sM, 0, 0, 0);
            sL = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql, sL, 0, 0, 0);
            sL = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh, sL, 0, 0, 0);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) pr[mb][rg] = sM[rg] + sL[rg] * H3_INV;
        }
        float mx = pr[0][0];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, pr[mb][rg]);
        mx = xmax32(xmax16(mx));
        const float mn = fmaxf(m, mx);
        const float corr = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) { pr[mb][rg] = __builtin_amdgcn_exp2f(pr[mb][rg] - mn); ps += pr[mb][rg]; }
        l = l * corr + ps;
#pragma unroll
        for (int db = 0; db < 2; ++db) { oM[db] *= corr; oL[db] *= corr; }
        // O^T[d][q] += V^T[d][slots] P[slots][q], two k-steps of 32 key slots
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            half8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p = pr[2 * st + (e >> 2)][e & 3];
                ph[e] = (_Float16)p;
                pl[e] = (_Float16)((p - (float)ph[e]) * H3_SCALE);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int off = (db * 16 + lq) * VPB + st * 64 + lg * 16;
                const half8 vh = *reinterpret_cast<const half8*>(&Vp[buf][0][off]);
                const half8 vl = *reinterpret_cast<const half8*>(&Vp[buf][1][off]);
                oM[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, oM[db], 0, 0, 0);
                oL[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, oL[db], 0, 0, 0);
                oL[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, oL[db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_kv(buf ^ 1);
        __syncthreads();
    }
    l = xsum32(xsum16(l));
    const float inv = 1.0f / l;
    float* op = out + ((size_t)img * n + q0 + lq) * 128 + h * 32 + lg * 4;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const f32x4 o = (oM[db] + oL[db] * H3_INV) * inv;
        *reinterpret_cast<float4*>(op + db * 16) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// attn_full_h3b_kernel (round 5): attn_full_h3_kernel with 128 queries per workgroup (a wave owns TWO 16-query blocks) and the V tile
// transposed through registers instead of through 2-byte LDS stores.  What the counters said about attn_full_h3_kernel
// (profiles/r05_lds_conflicts_before.txt): SQ_LDS_IDX_ACTIVE = 98 M cycles per launch = 190 us of LDS time per CU in a 340 us launch,
// 61 % of it bank conflicts -- the kernel is LDS-bound, and half of that is the sixteen ds_write_b16 per thread and chunk that
// transpose V ([key][d] in memory, [d][key slot] for the MFMA).  Here (1) a slot quad of the second product is four CONSECUTIVE keys
// (slot_of's inverse), so thread (d, slot quad) gathers V[key0 .. key0 + 3][d] with four 4-byte loads (a wave instruction still
// covers whole 128-byte rows: d is the fastest lane index) and writes ONE 8-byte row segment per plane; (2) every K / V fragment
// read from LDS feeds two query blocks, and a chunk is staged once per 128 queries instead of once per 64.
__global__ __launch_bounds__(256, 2) void attn_full_h3b_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n) {
    constexpr int KPB = 80, VPB = 144;                   // bytes per key row (32 d) / per d row (64 key slots), padded
    __shared__ __attribute__((aligned(16))) unsigned char Kp[2][2][64 * KPB];     // [buffer][plane]
    __shared__ __attribute__((aligned(16))) unsigned char Vp[2][2][32 * VPB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {                          // the query blocks of an (image, head) on one XCD: K / V come from its L2
            const int lin = blockIdx.x + gridDim.x * blockIdx.y;
            const int j = (lin & 7) * (total >> 3) + (lin >> 3);
            by = j / (int)gridDim.x; bx = j - by * (int)gridDim.x;
        }
    }
    const int ih = by, img = ih >> 2, h = ih & 3;
    const int q0 = bx * 128 + w * 32;
    const float* base = qkv + (size_t)img * n * 384;
    const int lq = lane & 15, lg = lane >> 4;
    half8 qh[2], ql[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* qp = base + (size_t)(q0 + 16 * j + lq) * 384 + h * 32 + lg * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(qp), a1 = *reinterpret_cast<const float4*>(qp + 4);
        const float sc = 0.17677669529663687f * 1.4426950408889634f;
        const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
#pragma unroll
        for (int e = 0; e < 8; ++e) { qh[j][e] = (_Float16)v[e]; ql[j][e] = (_Float16)((v[e] - (float)qh[j][e]) * H3_SCALE); }
    }
    // K staging: keys sr, sr + 32; d = sc4*4 .. +3
    const int sr = tid >> 3, sc4 = tid & 7;
    // V staging: d = vd, slot quads vq and vq + 8 (slots 4 vq .. and 32 + 4 vq ..) = keys vk0 .. vk0 + 3 and vk0 + 32 ..
    const int vd = tid & 31, vq = tid >> 5;
    const int vk0 = (vq & 1) * 16 + (vq >> 1) * 4;
    float4 kr0, kr1;
    float vv[2][4];
    auto load_kv = [&](int k0) {
        const float* rp = base + (size_t)(k0 + sr) * 384 + h * 32 + sc4 * 4 + 128;
        kr0 = *reinterpret_cast<const float4*>(rp);
        kr1 = *reinterpret_cast<const float4*>(rp + 32 * 384);
        const float* vp = base + (size_t)(k0 + vk0) * 384 + h * 32 + vd + 256;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) vv[u][i] = vp[(size_t)(32 * u + i) * 384];
    };
    auto store_kv = [&](int buf) {
        auto split4 = [](const float (&f)[4], half4v& hi, half4v& lo) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)f[i]; lo[i] = (_Float16)((f[i] - (float)hi[i]) * H3_SCALE); }
        };
        half4v hi, lo;
        { const float f[4] = {kr0.x, kr0.y, kr0.z, kr0.w}; split4(f, hi, lo); }
        *reinterpret_cast<half4v*>(&Kp[buf][0][sr * KPB + sc4 * 8]) = hi;
        *reinterpret_cast<half4v*>(&Kp[buf][1][sr * KPB + sc4 * 8]) = lo;
        { const float f[4] = {kr1.x, kr1.y, kr1.z, kr1.w}; split4(f, hi, lo); }
        *reinterpret_cast<half4v*>(&Kp[buf][0][(sr + 32) * KPB + sc4 * 8]) = hi;
        *reinterpret_cast<half4v*>(&Kp[buf][1][(sr + 32) * KPB + sc4 * 8]) = lo;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            split4(vv[u], hi, lo);
            *reinterpret_cast<half4v*>(&Vp[buf][0][vd * VPB + (32 * u + 4 * vq) * 2]) = hi;
            *reinterpret_cast<half4v*>(&Vp[buf][1][vd * VPB + (32 * u + 4 * vq) * 2]) = lo;
        }
    };
    load_kv(0);
    store_kv(0);
    __syncthreads();
    f32x4 oM[2][2], oL[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int db = 0; db < 2; ++db) { oM[j][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; oL[j][db] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const int nchunk = n >> 6;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        load_kv(min(c + 1, nchunk - 1) << 6);
        __builtin_amdgcn_sched_barrier(0);
        // S^T tiles: rows = keys 16*mb + 4*lg + rg, cols = queries of block j
        float pr[2][4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int off = (mb * 16 + lq) * KPB + lg * 16;
            const half8 kh = *reinterpret_cast<const half8*>(&Kp[buf][0][off]);
            const half8 kl = *reinterpret_cast<const half8*>(&Kp[buf][1][off]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 sM = (f32x4){0.f, 0.f, 0.f, 0.f}, sL = (f32x4){0.f, 0.f, 0.f, 0.f};
                sM = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[j], sM, 0, 0, 0);
                sL = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[j], sL, 0, 0, 0);
                sL = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[j], sL, 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) pr[j][mb][rg] = sM[rg] + sL[rg] * H3_INV;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float mx = pr[j][0][0];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, pr[j][mb][rg]);
            mx = xmax32(xmax16(mx));
            const float mn = fmaxf(m[j], mx);
            const float corr = __builtin_amdgcn_exp2f(m[j] - mn);
            m[j] = mn;
            float ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) { pr[j][mb][rg] = __builtin_amdgcn_exp2f(pr[j][mb][rg] - mn); ps += pr[j][mb][rg]; }
            l[j] = l[j] * corr + ps;
#pragma unroll
            for (int db = 0; db < 2; ++db) { oM[j][db] *= corr; oL[j][db] *= corr; }
        }
        // O^T[d][q] += V^T[d][slots] P[slots][q], two k-steps of 32 key slots; a V fragment feeds both query blocks
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            half8 ph[2], pl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = pr[j][2 * st + (e >> 2)][e & 3];
                    ph[j][e] = (_Float16)p;
                    pl[j][e] = (_Float16)((p - (float)ph[j][e]) * H3_SCALE);
                }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int off = (db * 16 + lq) * VPB + st * 64 + lg * 16;
                const half8 vh = *reinterpret_cast<const half8*>(&Vp[buf][0][off]);
                const half8 vl = *reinterpret_cast<const half8*>(&Vp[buf][1][off]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    oM[j][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[j], oM[j][db], 0, 0, 0);
                    oL[j][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[j], oL[j][db], 0, 0, 0);
                    oL[j][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[j], oL[j][db], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_kv(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lt = xsum32(xsum16(l[j]));
        const float inv = 1.0f / lt;
        float* op = out + ((size_t)img * n + q0 + 16 * j + lq) * 128 + h * 32 + lg * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const f32x4 o = (oM[j][db] + oL[j][db] * H3_INV) * inv;
            *reinterpret_cast<float4*>(op + db * 16) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// DDPM update with boundary sharing (share_states_over_boundaries :712-725, p_mean_variance :757-773, p_sample
// :804-808): one thread per state element of x [B*nb, HW, C] (channel-last).  The model output's state channels
// (c < C-3) are replaced by their mean (or sum) over the nb boundary copies of the design; x0, clamp, posterior mean;
// + sigma_t * z where z is SHARED over the boundary copies for the state channels (sample_noise :775-785).
struct Update2dArgs {
    const float* x; const float* eps; float* x_out; float* x0_out; float* mean_out;
    int B; int nb, HW, C, CP, use_avg, clip, add_noise;     // C logical channels (21), CP padded row pitch (24)
    const float* sqrt_recip; const float* sqrt_recipm1; const float* coef1; const float* coef2; const float* logvar;
    const float* sqrt_ac; const float* sqrt_1mac;           // sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod): objective pred_v
    const int* t_ptr; int t_imm;
    const float* noise_state; int64_t ns_t_stride;      // [B, HW, C-3] (+ t * stride) or null
    const float* noise_bound; int64_t nb_t_stride;      // [B*nb, HW, 3]
    uint64_t seed; int64_t sample_off;
    int* t_dec; unsigned* done;
    // model_predictions (:727-754): eps_out = the (shared) prediction; predict = 1: with use_avg bit 1 the prediction stays
    // un-shared AND x_start / mean are not shared either (that is p_mean_variance's job); rederive: eps_out =
    // predict_noise_from_start(x, t, x_start) = (sqrt_recip x - x_start) / sqrt_recipm1 (:738-739)
    float* eps_out; int predict, rederive;
};
// counter-based noise of the 2-D path: element index = pix * CP + c (float4-aligned groups); state channels keyed by
// the design (shared over its boundary copies), boundary channels keyed by the image
__device__ __forceinline__ void noise2d_state4(uint64_t seed, int64_t design, uint32_t tag, uint32_t pix, int G, int g, float (&z)[4]) {
    counter_normal4(seed, (uint64_t)design, tag, pix * (uint32_t)G + (uint32_t)g, z);
}
__device__ __forceinline__ void noise2d_bound4(uint64_t seed, int64_t image, uint32_t tag, uint32_t pix, int G, int g, float (&z)[4]) {
    counter_normal4(seed ^ 0x9e3779b97f4a7c15ull, (uint64_t)image, tag, pix * (uint32_t)G + (uint32_t)g, z);
}
// one thread per (design, pixel, group of 4 channels); loops over the nb boundary copies
__global__ __launch_bounds__(256) void update2d_kernel(const Update2dArgs a) {
    const int G = a.CP >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = step_scalar(a.t_ptr, a.t_imm);
    const int total = a.B * a.HW * G;
    if (i < total) {
        const int g = i % G;
        const int bp = i / G;
        const int pix = bp % a.HW;
        const int b = bp / a.HW;
        const int c0 = 4 * g, Cs = a.C - 3;
        const float ra = a.sqrt_recip[t], rb = a.sqrt_recipm1[t], k1 = a.coef1[t], k2 = a.coef2[t];
        // objective (bits 4-5 of use_avg; model_predictions :743-753): 0 = the model predicts the noise, 1 = x_start itself, 2 = v
        // (x_start = sqrt(abar) x - sqrt(1 - abar) v).  For 1 and 2 the reference shares nothing inside model_predictions and
        // ALWAYS re-derives the noise from the (clamped) x_start.
        const int obj = (a.use_avg >> 4) & 3;
        const float sa = obj == 2 ? a.sqrt_ac[t] : 0.f, sm = obj == 2 ? a.sqrt_1mac[t] : 0.f;
        auto x0_of = [&](float xv, float e) -> float {
            float x0 = obj == 0 ? __fsub_rn(__fmul_rn(ra, xv), __fmul_rn(rb, e)) : obj == 1 ? e : __fsub_rn(__fmul_rn(sa, xv), __fmul_rn(sm, e));
            if (a.clip) x0 = clamp_pm1(x0);
            return x0;
        };
        const float sigma = (a.add_noise && t > 0) ? expf(0.5f * a.logvar[t]) : 0.f;
        const bool noisy = a.add_noise && t > 0;
        // shared prediction of the state channels: mean (or sum) over the boundary copies
        // use_avg bit 1 = share_noise False (p_mean_variance :757-773): the prediction is NOT shared; the clamped x_start of the
        // state channels is (x0s), and then the posterior mean computed from it (ms)
        const bool nopred = (a.use_avg & 2) != 0 || obj != 0, avg = (a.use_avg & 1) != 0;
        const bool late = (a.use_avg & 2) != 0 && !a.predict;
        float es[4] = {0.f, 0.f, 0.f, 0.f}, x0s[4] = {0.f, 0.f, 0.f, 0.f}, ms[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 < Cs) {
            const float inv = (float)a.nb;
            for (int k = 0; k < a.nb; ++k) {
                const size_t o = (((size_t)(b * a.nb + k) * a.HW) + pix) * a.CP + c0;
                const float4 e = *reinterpret_cast<const float4*>(a.eps + o);
                if (!nopred) { es[0] += e.x; es[1] += e.y; es[2] += e.z; es[3] += e.w; continue; }
                if (!late) continue;
                const float4 x4 = *reinterpret_cast<const float4*>(a.x + o);
                const float ev[4] = {e.x, e.y, e.z, e.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) x0s[j] += x0_of(xv[j], ev[j]);
            }
            if (!nopred && avg) { es[0] /= inv; es[1] /= inv; es[2] /= inv; es[3] /= inv; }
            if (late) {
                if (avg) { x0s[0] /= inv; x0s[1] /= inv; x0s[2] /= inv; x0s[3] /= inv; }
                for (int k = 0; k < a.nb; ++k) {
                    const float4 x4 = *reinterpret_cast<const float4*>(a.x + (((size_t)(b * a.nb + k) * a.HW) + pix) * a.CP + c0);
                    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) ms[j] += __fadd_rn(__fmul_rn(k1, x0s[j]), __fmul_rn(k2, xv[j]));
                }
                if (avg) { ms[0] /= inv; ms[1] /= inv; ms[2] /= inv; ms[3] /= inv; }
            }
        }
        float zs[4] = {0.f, 0.f, 0.f, 0.f};
        if (noisy && c0 < Cs) {
            if (a.noise_state) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < Cs) zs[j] = a.noise_state[(size_t)t * a.ns_t_stride + ((size_t)b * a.HW + pix) * Cs + c0 + j];
            } else {
                noise2d_state4(a.seed, a.sample_off + b, (uint32_t)t, (uint32_t)pix, G, g, zs);
            }
        }
        for (int k = 0; k < a.nb; ++k) {
            const int im = b * a.nb + k;
            const size_t o = (((size_t)im * a.HW) + pix) * a.CP + c0;
            const float4 e4 = *reinterpret_cast<const float4*>(a.eps + o);
            const float4 x4 = *reinterpret_cast<const float4*>(a.x + o);
            const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
            float zb[4] = {0.f, 0.f, 0.f, 0.f};
            if (noisy && c0 + 3 >= Cs && c0 < a.C) {
                if (a.noise_bound) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j >= Cs && c0 + j < a.C)
                            zb[j] = a.noise_bound[(size_t)t * a.nb_t_stride + ((size_t)im * a.HW + pix) * 3 + (c0 + j - Cs)];
                } else {
                    noise2d_bound4(a.seed, (a.sample_off + b) * a.nb + k, (uint32_t)t, (uint32_t)pix, G, g, zb);
                }
            }
            float r0[4], rm[4], rx[4], re[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                const float e = (c < Cs && !nopred) ? es[j] : ev[j];
                float x0 = x0_of(xv[j], e);
                float mean = __fadd_rn(__fmul_rn(k1, x0), __fmul_rn(k2, xv[j]));
                if (late && c < Cs) { x0 = x0s[j]; mean = ms[j]; }
                const float z = (c < Cs) ? zs[j] : zb[j];
                const bool real = c < a.C;
                r0[j] = real ? x0 : 0.f; rm[j] = real ? mean : 0.f;
                rx[j] = real ? (noisy ? mean + sigma * z : mean) : 0.f;
                re[j] = real ? ((a.rederive || obj != 0) ? __fsub_rn(__fmul_rn(ra, xv[j]), x0) / rb : e) : 0.f;
            }
            if (a.eps_out) *reinterpret_cast<float4*>(a.eps_out + o) = make_float4(re[0], re[1], re[2], re[3]);
            if (a.x0_out) *reinterpret_cast<float4*>(a.x0_out + o) = make_float4(r0[0], r0[1], r0[2], r0[3]);
            if (a.mean_out) *reinterpret_cast<float4*>(a.mean_out + o) = make_float4(rm[0], rm[1], rm[2], rm[3]);
            if (a.x_out) *reinterpret_cast<float4*>(a.x_out + o) = make_float4(rx[0], rx[1], rx[2], rx[3]);
        }
    }
}

// x_T for the 2-D path from the counter-based generator (state channels shared over the boundaries of a design)
__global__ void fill_noise2d_kernel(float* x, int B, int nb, int HW, int C, int CP, uint64_t seed, int64_t off, uint32_t tag) {
    const int G = CP >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nb * HW * G) return;
    const int g = i % G;
    const int ip = i / G;
    const int pix = ip % HW;
    const int im = ip / HW;
    const int b = im / nb;
    const int c0 = 4 * g, Cs = C - 3;
    float zs[4] = {0.f, 0.f, 0.f, 0.f}, zb[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < Cs) noise2d_state4(seed, off + b, tag, (uint32_t)pix, G, g, zs);
    if (c0 + 3 >= Cs && c0 < C) noise2d_bound4(seed, (off + b) * nb + (im - b * nb), tag, (uint32_t)pix, G, g, zb);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = c0 + j; v[j] = (c >= C) ? 0.f : (c < Cs) ? zs[j] : zb[j]; }
    *reinterpret_cast<float4*>(x + (size_t)i * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace cindm
