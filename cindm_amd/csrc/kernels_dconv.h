// dconv_kernel: the k=5 convolutions of the DEEP U-Net levels (horizon 3 or 6 positions per sample, 128..1024 input
// channels) -- Conv1dBlock's Conv1d(k=5, pad=2) + GroupNorm(8) + Mish (model/diffusion_1d.py:197-214) and the
// ResidualTemporalBlock glue around it (:483-511) -- with the WHOLE activation tile of a workgroup resident in LDS.
//
// Why a second convolution kernel: measured on MI355X (profiles/r02_ablation_conv5.txt) a C = 512 launch of
// conv_gemm_h3_kernel spends 7 us in launch + prologue + epilogue, 4.4 us in MFMAs (40 % of them multiply the zero
// padding of a 3-position sequence), 2.3 us in per-stage restaging + barriers, while its weight stream alone needs
// 4.4 us (tools/micro/bstream.hip).  Here
//   * activations travel between layers as SPLIT-FP16 PLANES (hi = fp16(v), lo = fp16((v - hi) * 2^11)), written by the
//     producer's epilogue in the consumer's LDS image order, so staging is a linear 16-byte copy: no conversion, no
//     normalise-on-load, and -- because wave w only ever reads the k-steps it staged itself (K is split over the four
//     waves) -- NO barrier between staging and the K loop and none inside it;
//   * tile rows are POSITION-MAJOR (row = position * S + sample, S = 48 / L samples per tile): a 16-row MFMA block holds
//     one (L = 3) or two (L = 6) positions of 16 / 8 samples, so (block, tap) pairs that only touch the zero padding are
//     skipped at compile time (9 of 15 pairs remain at L = 3, 13 of 15 at L = 6) and every remaining tap window is an
//     aligned run of 16 consecutive 16-byte LDS slots (conflict-free ds_read_b128);
//   * GroupNorm runs in registers (the thread that owns column n of rows r, r+8, .. holds all positions of its samples;
//     the group's columns are adjacent lanes: DPP / permlane all-reduce), two-pass (mean, then M2);
//   * at C_out = 512 a group spans two 32-column tiles: the two workgroups exchange their (mean, M2) halves through
//     8-byte {value, tag} granules (agent-scope relaxed atomics, tag = per-forward epoch), MI355X_MICROARCH.md
//     "handoff-1to1"; partners are adjacent block indices, the spin is bounded and reports through an error flag.
// Weights: pack_weight_h3 / pack_weight_h3_res layouts of conv_gemm_h3_kernel (shared).
#pragma once
#include "kernels.h"

namespace cindm {

struct DSrc {
    const float* f32;        // fp32 [rows = sample * L + position, ld], or null
    const uint4* planes;     // tiled planes: hi plane [tile][C/32][4][48 rows][8 halfs]; lo plane at + pstride
    size_t pstride;          // uint4 elements between the two planes
    int C, ld;
};

struct DconvArgs {
    DSrc src[2];
    const uint4* W; const float* bias;          // [n-tile][stage of 128 ch][(tap*2+nb)*2+plane][256 threads][8 halfs]
    int nch;                                    // stages in total (KPW0 + KPW1)
    int Bp, N, NT;                              // samples, output channels, n-tiles (N / 32)
    int gw;                                     // GroupNorm group width in channels: 16, 32 or 64
    const float* gamma; const float* beta;
    const float* tb; int tb_ld; const int* t_ptr; int t_imm;       // + time bias row (after the Mish) or null
    const float* res; int ldres;                // + residual (fp32, sample-major rows) or null
    float* out_f32; int ldo;                    // fp32 output (sample-major rows) or null
    uint4* out_planes; size_t out_pstride;      // planes output (tiled) or null
    const uint4* W2; const float* bias2; float* out2; int ldo2;     // riding 1x1 residual_conv: out2 = W2 . x + bias2
    unsigned long long* xchg; const int* epoch; int* err_flag;     // gw == 64: pair exchange of GroupNorm halves
    Pf pf;                                      // L2 warm-up for the next launch (kernels.h)
    int dbg;                                    // timing ablations (wrong results): 1 return at entry, 2 after staging,
                                                // 3 no K loop, 4 no epilogue, 5 no pair exchange, 6 return after the cross-wave reduce, 7 before the stores, 8 no planes store
};

__global__ void dconv_epoch_kernel(int* e) { if (threadIdx.x == 0 && blockIdx.x == 0) e[0] += 1; }

template <int L, int KPW0, int KPW1, bool RES>
__global__ __launch_bounds__(256) void dconv_kernel(const DconvArgs a) {
    constexpr int T = 5;
    constexpr int S = 48 / L;                 // samples per tile (16 or 8)
    constexpr int PB = 16 / S;                // positions per 16-row block (1 or 2)
    constexpr int H = PB - 1;                 // zero halo positions on each side of the image
    constexpr int RPAD = (L + 2 * H) * S;     // image rows per (k-step, k-quarter)
    constexpr int NWIN = L + H;               // distinct tap windows (start positions 0 .. L + H - 1)
    constexpr int KPWM = KPW0 > KPW1 ? KPW0 : KPW1;
    constexpr int KST = 4 * KPWM;             // k-steps (32 channels) held in LDS at a time
    constexpr int PLANE_U4 = KST * 4 * RPAD;
    static_assert(L == 3 || L == 6, "3 or 6 positions per sample");
    __shared__ uint4 Img[2][PLANE_U4];        // [plane][k-step][k-quarter][row] x 8 halfs
    __shared__ float Red[4][TM * LDR];
    __shared__ uint4 Tile[2 * 48 * 5];        // output planes of this tile: [plane][row][16 dwords + 4 pad]

    if (a.dbg == 1) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int b0 = mt * S;
    const int ns = min(S, a.Bp - b0);
    const int n = tid & 31, rq = tid >> 5;
    const int gn = nt * TN + n;

    // row r = rq + 8 q of the tile: S = 16: position q >> 1, sample rq + 8 (q & 1);  S = 8: position q, sample rq
    int grow[6];
    bool sok[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int pos = (S == 16) ? (q >> 1) : q;
        const int s = (S == 16) ? rq + 8 * (q & 1) : rq;
        sok[q] = s < ns;
        grow[q] = (b0 + min(s, ns - 1)) * L + pos;
    }

    // ---- staging: wave w stages (and later reads) only the k-steps 4 j + w ----------------------------------------
    // item = lane + 64 i (i < 3) of a k-step: planes source: k-quarter item / 48, row item % 48 (linear copy);
    // fp32 source: row item >> 2, k-quarter item & 3 (a row's 32 channels are one 128-byte line).  Either way an item
    // is two 16-byte loads from (base_i + k-step * kstride) and (.. + second): the source kind only selects addresses.
    struct Stg { const uint4* base[3]; size_t kstride, second; int slot[3]; bool f32; };
    auto stg_init = [&](const DSrc& s, Stg& g) {
        g.f32 = s.planes == nullptr;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int item = lane + 64 * i;
            if (!g.f32) {
                const int kq = item / 48, row = item - kq * 48;
                g.base[i] = s.planes + (size_t)mt * (s.C >> 5) * 192 + item;
                g.slot[i] = kq * RPAD + H * S + row;
            } else {
                const int row = item >> 2, kq = item & 3;
                const int sm = row % S, pos = row / S;
                g.base[i] = reinterpret_cast<const uint4*>(s.f32 + (size_t)((b0 + min(sm, ns - 1)) * L + pos) * s.ld + kq * 8);
                g.slot[i] = kq * RPAD + H * S + row;
            }
        }
        g.kstride = g.f32 ? 8 : 192;
        g.second = g.f32 ? 1 : s.pstride;
    };
    auto load_raw = [&](const Stg& g, int j, uint4 (&raw)[3][2]) {
        const size_t ko = (size_t)(4 * j + w) * g.kstride;
#pragma unroll
        for (int i = 0; i < 3; ++i) { raw[i][0] = g.base[i][ko]; raw[i][1] = g.base[i][ko + g.second]; }
    };
    auto store_raw = [&](const Stg& g, int j, const uint4 (&raw)[3][2]) {
        const int kb = (4 * j + w) * 4 * RPAD;
        if (!g.f32) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { Img[0][kb + g.slot[i]] = raw[i][0]; Img[1][kb + g.slot[i]] = raw[i][1]; }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float4 v0 = __builtin_bit_cast(float4, raw[i][0]), v1 = __builtin_bit_cast(float4, raw[i][1]);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hi[e] = (_Float16)v[e];
                    lo[e] = (_Float16)((v[e] - (float)hi[e]) * H3_SCALE);
                }
                Img[0][kb + g.slot[i]] = __builtin_bit_cast(uint4, hi);
                Img[1][kb + g.slot[i]] = __builtin_bit_cast(uint4, lo);
            }
        }
    };

    Stg g0, g1;
    stg_init(a.src[0], g0);
    uint4 raw0[KPW0][3][2];
#pragma unroll
    for (int j = 0; j < KPW0; ++j) load_raw(g0, j, raw0[j]);

    // B: this wave's fragments of stage ch, tap by tap; reloaded for the next stage right after their last use
    half8 breg[T][2][2];
    const uint4* wbase = a.W + (size_t)nt * a.nch * (T * 4) * 256 + tid;
    auto load_b_tap = [&](int ch, int tap) {
        const uint4* wp = wbase + ((size_t)ch * (T * 4) + tap * 4) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) breg[tap][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
    };
#pragma unroll
    for (int tap = 0; tap < T; ++tap) load_b_tap(0, tap);
    half8 rreg[2][2];
    const uint4* rbase = a.W2 + (size_t)nt * a.nch * 4 * 256 + tid;
    auto load_r = [&](int ch) {
        if constexpr (RES) {
            const uint4* wp = rbase + (size_t)ch * 4 * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) rreg[q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
        }
    };
    load_r(0);
    // the second source's rows are fetched now and parked in registers until the first source's k-steps are done
    uint4 raw1[KPW1 > 0 ? KPW1 : 1][3][2];
    if constexpr (KPW1 > 0) {
        stg_init(a.src[1], g1);
#pragma unroll
        for (int j = 0; j < KPW1; ++j) load_raw(g1, j, raw1[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue operands: requested behind the operand streams, consumed after the K loop ------------------------
    const int t_now = a.t_ptr ? *a.t_ptr : a.t_imm;
    const unsigned tag = a.epoch ? (unsigned)*a.epoch : 0u;
    const float bias = a.bias ? a.bias[gn] : 0.f;
    const float gam = a.gamma[gn], bet = a.beta[gn];
    const float tbv = a.tb ? a.tb[(size_t)t_now * a.tb_ld + gn] : 0.f;
    float bias2 = 0.f;
    if constexpr (RES) bias2 = a.bias2 ? a.bias2[gn] : 0.f;
    float rs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.res) {
#pragma unroll
        for (int q = 0; q < 6; ++q) rs[q] = a.res[(size_t)grow[q] * a.ldres + gn];
    }
    __builtin_amdgcn_sched_barrier(0);

    // zero halo rows of this wave's k-steps (L = 6: one position of 8 samples on each side)
    if constexpr (H > 0) {
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int i = lane; i < KPWM * 4 * 2 * H * S; i += 64) {
            const int blk = i / (2 * H * S), r = i - blk * (2 * H * S);       // blk = j * 4 + kq
            const int ks = 4 * (blk >> 2) + w, kq = blk & 3;
            const int row = r < H * S ? r : (L + H) * S + (r - H * S);
            Img[0][(ks * 4 + kq) * RPAD + row] = z;
            Img[1][(ks * 4 + kq) * RPAD + row] = z;
        }
    }
#pragma unroll
    for (int j = 0; j < KPW0; ++j) store_raw(g0, j, raw0[j]);

    f32x4 accM[3][2], accL[3][2], accRM[3][2], accRL[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accRM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accRL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // one k-step (32 channels of this wave) against all five taps: window p = 16 rows starting at image position p;
    // PF: the next stage's fragments are requested tap by tap behind their last use (not on the very last k-step)
    auto kstep = [&](int j, int chn, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        const int base = ((4 * j + w) * 4 + (lane >> 4)) * RPAD + (lane & 15);
        half8 fh[NWIN], fl[NWIN];
#pragma unroll
        for (int p = 0; p < NWIN; ++p) {
            fh[p] = __builtin_bit_cast(half8, Img[0][base + p * S]);
            fl[p] = __builtin_bit_cast(half8, Img[1][base + p * S]);
        }
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                const int p = mb * PB + tap - 2 + H;
                if (p < 0 || p >= NWIN) continue;                 // the window lies in the zero padding: nothing to add
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], breg[tap][nb][0], accM[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], breg[tap][nb][1], accL[mb][nb], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[p], breg[tap][nb][0], accL[mb][nb], 0, 0, 0);
            }
            if constexpr (RES) if (tap == 2) {                    // the 1x1 residual_conv reads the centre-tap windows
#pragma unroll
                for (int mb = 0; mb < 3; ++mb) {
                    const int p = mb * PB + H;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        accRM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], rreg[nb][0], accRM[mb][nb], 0, 0, 0);
                        accRL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], rreg[nb][1], accRL[mb][nb], 0, 0, 0);
                    }
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        accRL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[p], rreg[nb][0], accRL[mb][nb], 0, 0, 0);
                }
                if constexpr (PF) load_r(chn);
            }
            if constexpr (PF) load_b_tap(chn, tap);
        }
    };
    if (a.dbg == 2) { if (Img[0][tid].x == 0x12345u) a.out2[0] = 1.f; return; }
    PfRegs pfr;
    pfr.v[0][0] = pfr.v[0][1] = pfr.v[1][0] = pfr.v[1][1] = 0u;
    if (a.dbg != 3) {
        constexpr std::true_type PFY{};
        constexpr std::false_type PFN{};
        if constexpr (KPW1 == 0) {
#pragma unroll 1
            for (int j = 0; j < KPW0 - 1; ++j) kstep(j, j + 1, PFY);
            l2_prefetch(a.pf, pfr);
            kstep(KPW0 - 1, 0, PFN);
        } else {
#pragma unroll 1
            for (int j = 0; j < KPW0; ++j) kstep(j, j + 1, PFY);
#pragma unroll
            for (int j = 0; j < KPW1; ++j) store_raw(g1, j, raw1[j]);
#pragma unroll 1
            for (int j = 0; j < KPW1 - 1; ++j) kstep(j, KPW0 + j + 1, PFY);
            l2_prefetch(a.pf, pfr);
            kstep(KPW1 - 1, 0, PFN);
        }
    }
    if (a.dbg == 4) { if (accM[0][0][0] + accL[1][1][2] + accRM[2][0][1] + accRL[0][1][3] == 123.456f) a.out2[0] = 1.f; return; }

    // ---- epilogue -------------------------------------------------------------------------------------------------
    auto reduce_to = [&](const f32x4 (&m)[3][2], const f32x4 (&l)[3][2], float bs, float (&v)[6]) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    Red[w][(mb * 16 + (lane >> 4) * 4 + rg) * LDR + nb * 16 + (lane & 15)] = m[mb][nb][rg] + l[mb][nb][rg] * H3_INV;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = rq + 8 * q;
            v[q] = ((Red[0][r * LDR + n] + Red[1][r * LDR + n]) + (Red[2][r * LDR + n] + Red[3][r * LDR + n])) + bs;
        }
    };
    float v[6];
    reduce_to(accM, accL, bias, v);
    if (a.dbg == 6) { if (v[0] + v[5] == 123.456f) a.out2[0] = 1.f; return; }

    // GroupNorm over (group columns x L positions) of each sample: this thread holds every position of its sample(s)
    constexpr int NSAMP = (S == 16) ? 2 : 1;
    const int gwt = a.gw < TN ? a.gw : TN;                       // group columns inside this tile: 16 or 32
    const float cnt = (float)(L * gwt);
    float mean[NSAMP], rstd[NSAMP];
#pragma unroll
    for (int js = 0; js < NSAMP; ++js) {
        float s1 = 0.f;
#pragma unroll
        for (int pos = 0; pos < L; ++pos) s1 += v[(S == 16) ? 2 * pos + js : pos];
        s1 = row16_sum(s1);
        if (gwt == 32) s1 = xsum16(s1);
        const float m = s1 / cnt;
        float s2 = 0.f;
#pragma unroll
        for (int pos = 0; pos < L; ++pos) { const float d = v[(S == 16) ? 2 * pos + js : pos] - m; s2 += d * d; }
        s2 = row16_sum(s2);
        if (gwt == 32) s2 = xsum16(s2);
        mean[js] = m; rstd[js] = s2;                              // rstd holds M2 until the exchange below is done
    }
    if (a.gw == 64 && a.dbg != 5) {
        // the group's other 32 columns belong to the workgroup nt ^ 1 of the same m-tile: swap (mean, M2) halves
        const int sbase = ((mt * a.NT + nt) * 16) * 2, pbase = ((mt * a.NT + (nt ^ 1)) * 16) * 2;
#pragma unroll
        for (int js = 0; js < NSAMP; ++js) {
            const int s = (S == 16) ? rq + 8 * js : rq;
            if (n == 0) {
                const unsigned long long g0 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, mean[js]);
                const unsigned long long g1 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, rstd[js]);
                __hip_atomic_store(a.xchg + sbase + s * 2, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.xchg + sbase + s * 2 + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int js = 0; js < NSAMP; ++js) {
            const int s = (S == 16) ? rq + 8 * js : rq;
            unsigned long long g0 = 0, g1 = 0;
            int spins = 0;
            while (true) {
                g0 = __hip_atomic_load(a.xchg + pbase + s * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g1 = __hip_atomic_load(a.xchg + pbase + s * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
                if (__all(ok)) break;
                if (++spins > (1 << 20)) { if (lane == 0) atomicExch(a.err_flag, 1); break; }     // never hang the GPU
                __builtin_amdgcn_s_sleep(2);
            }
            const float mp = __builtin_bit_cast(float, (unsigned)g0), M2p = __builtin_bit_cast(float, (unsigned)g1);
            const float m = 0.5f * (mean[js] + mp);
            const float d0 = mean[js] - m, d1 = mp - m;
            mean[js] = m;
            rstd[js] = (rstd[js] + M2p) + cnt * (d0 * d0 + d1 * d1);
        }
    }
    const float cnt_all = a.gw == 64 ? 2.f * cnt : cnt;
#pragma unroll
    for (int js = 0; js < NSAMP; ++js) rstd[js] = 1.0f / sqrtf(rstd[js] / cnt_all + 1e-5f);

    float y[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int js = (S == 16) ? (q & 1) : 0;
        y[q] = mish_f((v[q] - mean[js]) * rstd[js] * gam + bet) + tbv;
        if (a.res) y[q] += rs[q];
    }
    if (a.dbg == 7) { if (y[0] + y[5] == 123.456f) a.out2[0] = 1.f; return; }
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if (a.out_f32 && sok[q]) a.out_f32[(size_t)grow[q] * a.ldo + gn] = y[q];
    if (a.out_planes && a.dbg != 8) {
        // planes of the tile: row-major [plane][row][32 halfs + pad] in LDS, re-read as 16-byte (row, k-quarter) items.
        // A lane pairs with its neighbour column (DPP quad swap): the even lane writes the hi dword (own, neighbour),
        // the odd lane the lo dword (neighbour, own) -- one 32-bit LDS store per value instead of two 16-bit ones.
        constexpr int TP = 20;                                    // dwords per tile row (16 + 4 pad)
        uint32_t* tw = reinterpret_cast<uint32_t*>(Tile);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = rq + 8 * q;
            const _Float16 hi = (_Float16)y[q];
            const _Float16 lo = (_Float16)((y[q] - (float)hi) * H3_SCALE);
            const uint32_t own = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
            const uint32_t nbr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
            const bool odd = n & 1;
            const uint32_t word = odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
            tw[(odd ? 48 * TP : 0) + r * TP + (n >> 1)] = word;
        }
        __syncthreads();
        for (int i = tid; i < 384; i += 256) {
            const int pl = i >= 192 ? 1 : 0, within = i - pl * 192;
            const int kq = within / 48, row = within - kq * 48;
            const uint4 t4 = *reinterpret_cast<const uint4*>(tw + pl * 48 * TP + row * TP + kq * 4);
            a.out_planes[pl * a.out_pstride + ((size_t)mt * a.NT + nt) * 192 + within] = t4;
        }
    }
    if constexpr (RES) {
        __syncthreads();                                          // Red is reused
        float r2[6];
        reduce_to(accRM, accRL, bias2, r2);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (sok[q]) a.out2[(size_t)grow[q] * a.ldo2 + gn] = r2[q];
    }
    l2_prefetch_done(a.pf, pfr);
}

}  // namespace cindm
