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
#include <type_traits>

namespace cindm {

// Bound of the hand-over spins.  A chain whose flag is already up (an earlier launch timed out: everything computed since
// is garbage and the chain will be re-run) only waits briefly: a chain of launches that all wait the full bound would
// turn one lost partner into minutes.  (~1 us per spin iteration: 2^20 ~ 1 s.)
// Call it at the TOP of the kernel: the flag is a uniform address, i.e. a scalar load, requested with the step's other scalars
// (a load at the spin itself is one more exposed round trip).
__device__ __forceinline__ int spin_bound(const int* err_flag, bool forced_short, int full_log2 = 20) {
    const int seen = err_flag ? uniform_word(err_flag) : 0;
    return (forced_short || seen) ? (1 << 8) : (1 << full_log2);
}

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
    int stress;                                 // > 0: pseudo-random pauses before the hand-overs (stress_delay, kernels.h)
    int dbg;                                    // timing ablations (wrong results): 1 return at entry, 2 after staging,
                                                // 3 no K loop, 4 no epilogue, 5 no pair exchange, 6 return after the cross-wave reduce, 7 before the stores, 8 no planes store,
                                                // 9 (tests) odd n-tiles skip their publish and the spin bound is short: forces the exchange time-out
};

// e[0]: the epoch eager forwards and the first step of a loop read; e[8]: the ping-pong sample loop's second slot
__global__ void dconv_epoch_kernel(int* e) { if (threadIdx.x == 0 && blockIdx.x == 0) e[0] = max(e[0], e[8]) + 1; }

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
    // the step's scalars first (lines the previous step wrote: each a miss to memory), consumed after the operand streams
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const unsigned tag = a.epoch ? (unsigned)uniform_word(a.epoch) : 0u;
    const int spin_cap = spin_bound(a.err_flag, a.dbg == 9);
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
    // (optional vectors: the POINTER is selected and the load is unconditional -- `p ? p[i] : 0.f` is a load under a branch
    // whose join waits vmcnt(0) before it may write the 0, which drains every operand load issued above it)
    const float gam = a.gamma[gn], bet = a.beta[gn];
    const float bias_ld = (a.bias ? a.bias : a.gamma)[gn];
    const float tbv_ld = (a.tb ? a.tb + (size_t)t_now * a.tb_ld : a.gamma)[gn];
    const float bias = a.bias ? bias_ld : 0.f, tbv = a.tb ? tbv_ld : 0.f;
    float bias2 = 0.f;
    if constexpr (RES) { const float b2 = (a.bias2 ? a.bias2 : a.gamma)[gn]; bias2 = a.bias2 ? b2 : 0.f; }
    float rs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        // residual rows: from `res`, or (no residual) one harmless word of gamma per lane
        const float* rp = a.res ? a.res + gn : a.gamma + gn;
        const size_t rld = a.res ? (size_t)a.ldres : 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) { const float r = rp[(size_t)grow[q] * rld]; rs[q] = a.res ? r : 0.f; }
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
    // (the staged rows are written to LDS k-step by k-step inside the K loop: the MFMAs of k-step 0 start as soon as its
    // six loads have landed, the rest of the tile streams in behind them)

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
            if constexpr (PF) {           // requested HERE, behind the tap's last use, and pinned (see dconv2_kernel)
                load_b_tap(chn, tap);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (a.dbg == 2) { if (Img[0][tid].x == 0x12345u) a.out2[0] = 1.f; return; }
    PfRegs pfr;
#pragma unroll
    for (int k = 0; k < PF_REGIONS; ++k) pfr.v[k][0] = pfr.v[k][1] = 0u;
    if (a.dbg != 3) {
        constexpr std::true_type PFY{};
        constexpr std::false_type PFN{};
        if constexpr (KPW1 == 0) {
#pragma unroll
            for (int j = 0; j < KPW0 - 1; ++j) { store_raw(g0, j, raw0[j]); kstep(j, j + 1, PFY); }
            l2_prefetch(a.pf, pfr);
            store_raw(g0, KPW0 - 1, raw0[KPW0 - 1]);
            kstep(KPW0 - 1, 0, PFN);
        } else {
#pragma unroll
            for (int j = 0; j < KPW0; ++j) { store_raw(g0, j, raw0[j]); kstep(j, j + 1, PFY); }
#pragma unroll
            for (int j = 0; j < KPW1 - 1; ++j) { store_raw(g1, j, raw1[j]); kstep(j, KPW0 + j + 1, PFY); }
            l2_prefetch(a.pf, pfr);
            store_raw(g1, KPW1 - 1, raw1[KPW1 - 1]);
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
    stress_delay(a.stress, 3u);
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
        stress_delay(a.stress, 1u);
        const int sbase = ((mt * a.NT + nt) * 16) * 2, pbase = ((mt * a.NT + (nt ^ 1)) * 16) * 2;
#pragma unroll
        for (int js = 0; js < NSAMP; ++js) {
            const int s = (S == 16) ? rq + 8 * js : rq;
            if (n == 0 && !(a.dbg == 9 && (nt & 1))) {       // dbg 9 (tests): odd n-tiles never publish -> their partners time out
                const unsigned long long g0 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, mean[js]);
                const unsigned long long g1 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, rstd[js]);
                __hip_atomic_store(a.xchg + sbase + s * 2, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.xchg + sbase + s * 2 + 1, g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        stress_delay(a.stress, 2u);
        // the partner's granules of BOTH samples in one sweep (one loop per sample was two serial round trips)
        unsigned long long gq[NSAMP][2];
        {
            int spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int js = 0; js < NSAMP; ++js) {
                    const int s = (S == 16) ? rq + 8 * js : rq;
                    gq[js][0] = __hip_atomic_load(a.xchg + pbase + s * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gq[js][1] = __hip_atomic_load(a.xchg + pbase + s * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int js = 0; js < NSAMP; ++js) ok = ok & ((unsigned)(gq[js][0] >> 32) == tag) & ((unsigned)(gq[js][1] >> 32) == tag);
                if (__all(ok)) break;
                if (++spins > spin_cap) { if (lane == 0) atomicExch(a.err_flag, 1); break; }     // never hang the GPU
                __builtin_amdgcn_s_sleep(2);
            }
        }
#pragma unroll
        for (int js = 0; js < NSAMP; ++js) {
            const unsigned long long g0 = gq[js][0], g1 = gq[js][1];
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
        stress_delay(a.stress, 4u);
        __syncthreads();
        for (int i = tid; i < 384; i += 256) {
            const int pl = i >= 192 ? 1 : 0, within = i - pl * 192;
            const int kq = within / 48, row = within - kq * 48;
            const uint4 t4 = *reinterpret_cast<const uint4*>(tw + pl * 48 * TP + row * TP + kq * 4);
            a.out_planes[pl * a.out_pstride + ((size_t)mt * a.NT + nt) * 192 + within] = t4;
        }
    }
    if constexpr (RES) {
        stress_delay(a.stress, 5u);
        __syncthreads();                                          // Red is reused
        float r2[6];
        reduce_to(accRM, accRL, bias2, r2);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (sok[q]) a.out2[(size_t)grow[q] * a.ldo2 + gn] = r2[q];
    }
    l2_prefetch_done(a.pf, pfr);
}


// ---------------------------------------------------------------------------------------------------------------------
// dconv2_kernel<L, KPW0, KPW1, RES, KPWB>: a whole deep-level ResidualTemporalBlock (model/diffusion_1d.py:483-511) in ONE
// launch -- both Conv1dBlocks, the time bias, the riding 1x1 residual_conv and the residual add:
//     y0  = Mish(GN(conv5(x) + b0)) + tbias_t          (phase A: dconv_kernel's launch A)
//     out = Mish(GN(conv5(y0) + b1)) + (x | Wr x + br)  (phase B: its launch B)
// The seam between the two convolutions is an all-gather inside the m-tile's column of workgroups: workgroup (nt, mt)
// produces channels [32 nt, 32 nt + 32) of y0 for the tile's 48 rows, and needs ALL channels of those rows for phase B.
// Each workgroup publishes its 6 KB of planes with write-through (sc1) 16-byte stores, drains, and raises ONE flag word
// (tag = the forward's epoch; MI355X_MICROARCH.md "handoff-flag", recipe R1); wave w of a consumer polls the flags of
// the k-steps it owns (k-step ks = channels of producer nt = ks), then fetches them with sc1 loads (no acquire fence: both
// sides are sc1) straight into the staging registers of phase B's K loop.  What the merge removes per block: one kernel
// boundary (dispatch + drain), the fp32 round trip of r = Wr x + br (it stays in registers) and the cold first loads of
// launch B (phase B's first weight taps are requested before phase A's epilogue).  The workgroups of a column are
// adjacent block indices, so they are dispatched together; the spins are bounded and report through err_flag.
struct Dconv2Args {
    DSrc src[2];
    const uint4* Wa; const float* bias_a; int ncha;      // conv A: stages of 128 input channels (KPW0 + KPW1)
    const uint4* Wb; const float* bias_b;                // conv B: N -> N
    const uint4* W2; const float* bias2;                 // riding 1x1 residual_conv (RES)
    int Bp, N, NT, gw;
    const float* gamma_a; const float* beta_a; const float* gamma_b; const float* beta_b;
    const float* tb; int tb_ld; const int* t_ptr; int t_imm;
    const float* res; int ldres;                          // identity residual x (fp32, sample-major rows) when !RES
    uint4* y0; size_t y0_pstride;                         // hand-over planes [plane][tile][N/32][192] (global)
    unsigned* flags;                                      // [tiles][NT] publish flags (8-byte slots, tag = epoch)
    float* out_f32; int ldo; uint4* out_planes; size_t out_pstride;
    unsigned long long* xchg_a; unsigned long long* xchg_b; const int* epoch; int* err_flag;     // gw == 64 pair exchanges
    Pf pf; int stress; int dbg;
    PhaseBuf ph;                                          // phase clocks (profiling builds; kernels.h)
    int tune;                                             // experiment switches for same-box A/B runs (option "tune"; 0 = the shipped choice)
    int xs;                                               // XCDs a column of workgroups spreads over (0: identity mapping)
};

// (experiment builds -DCINDM_KPROF, library variant "kprof": the phase clocks of dconv2_kernel go INSIDE phase A's K loop -- marks 2 + 2j /
// 3 + 2j = k-step j staged / multiplied, single-source instances only -- and the marks of the later phases are off)
#ifdef CINDM_KPROF
#define PHK(i) PH(i)
#define PHX(i) do { } while (0)
#else
#define PHK(i) do { } while (0)
#define PHX(i) PH(i)
#endif
template <int L, int KPW0, int KPW1, bool RES, int KPWB>
__global__ __launch_bounds__(256) void dconv2_kernel(const Dconv2Args a) {
    constexpr int T = 5;
    constexpr int S = 48 / L, PB = 16 / S, H = PB - 1, RPAD = (L + 2 * H) * S, NWIN = L + H;
    constexpr int KPWA = KPW0 > KPW1 ? KPW0 : KPW1;
    constexpr int KPWM = KPWA > KPWB ? KPWA : KPWB;
    constexpr int KST = 4 * KPWM;
    constexpr int PLANE_U4 = KST * 4 * RPAD;
    static_assert(L == 3 || L == 6, "3 or 6 positions per sample");
    __shared__ uint4 Img[2][PLANE_U4];
    __shared__ float Red[4][TM * LDR];
    __shared__ uint4 Tile[2 * 48 * 5];

    PH_DECL;
    PH(0);                                    // phase clocks, profiling builds (kernels.h): 0 = entry
    // The step's scalars (timestep, exchange epoch, error flag) live in lines the PREVIOUS step's last kernel wrote: each is a
    // miss all the way to memory.  Requested first, all three together, consumed after the operand streams have been issued
    // (requested where they are used they were three serial round trips in the prologue).
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const unsigned tag = (unsigned)uniform_word(a.epoch);
    const int err_seen = uniform_word(a.err_flag);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Workgroup -> (n-tile, m-tile), round 6: a column's NT workgroups spread over XS = a.xs XCDs (block index % 8 = XCD, observed), NT / XS = 4
    // on each -- XCD x hosts the n-tiles (x % XS) + XS k of the m-tiles congruent to x / XS modulo 8 / XS -- instead of over all 8 with n-tile(s)
    // x (, x + 8) of EVERY m-tile.  A column's y0 and activation tile are then fetched by XS XCDs whose second and later workgroups find the
    // lines in the XCD's L2 (sc1 loads are L2-served): fewer bytes cross the fabric -- phase B's K loop, bound by that fetch, 3.96 -> 3.64 us
    // at C = 512.  The pair-exchange partners nt ^ 1 stay on DIFFERENT XCDs: on the same one the exchange took 1.6 us instead of 0.65 (the
    // first form tried, n-tiles 4 (x & 3) + k: +15 us per step).  The host picks XS (a bijection of the grid needs NT % XS == 0 and the
    // number of m-tiles % (8 / XS) == 0; 0 = identity) and registers the warm-up pieces to match (emit_rtb_dconv2).
    int nt = blockIdx.x, mt = blockIdx.y;
#ifndef CINDM_NO_XCD_REMAP
    if (a.xs > 0) {
        const int XS = a.xs, bl = blockIdx.x + gridDim.x * blockIdx.y, xcd = bl & 7, sl = bl >> 3, q4 = gridDim.x / XS;
        nt = (sl % q4) * XS + (xcd & (XS - 1)); mt = (xcd / XS) + (8 / XS) * (sl / q4);
    }
#endif
    const int b0 = mt * S;
    const int ns = min(S, a.Bp - b0);
    const int n = tid & 31, rq = tid >> 5;
    const int gn = nt * TN + n;
    int grow[6];
    bool sok[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int pos = (S == 16) ? (q >> 1) : q;
        const int s = (S == 16) ? rq + 8 * (q & 1) : rq;
        sok[q] = s < ns;
        grow[q] = (b0 + min(s, ns - 1)) * L + pos;
    }

    // ---- staging of phase A (as dconv_kernel) ------------------------------------------------------------------------
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
    auto store_raw = [&](bool f32, const int (&slot)[3], int j, const uint4 (&raw)[3][2]) {
        const int kb = (4 * j + w) * 4 * RPAD;
        if (!f32) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { Img[0][kb + slot[i]] = raw[i][0]; Img[1][kb + slot[i]] = raw[i][1]; }
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
                Img[0][kb + slot[i]] = __builtin_bit_cast(uint4, hi);
                Img[1][kb + slot[i]] = __builtin_bit_cast(uint4, lo);
            }
        }
    };

    Stg g0, g1;
    stg_init(a.src[0], g0);
    uint4 raw0[KPW0][3][2];
#pragma unroll
    for (int j = 0; j < KPW0; ++j) load_raw(g0, j, raw0[j]);

    // The weight ring: one 128-channel stage (5 taps x 4 fragments = 80 registers), refilled tap by tap one stage ahead.
    // (Round 6 built a ring of up to three stages for phase B -- requested inside phase A's epilogue, one wave publishing so that the
    // other three need no drain: 438 - 453 registers, loads straight into AGPRs, no scratch -- and it did NOT pay: phase B's K loop
    // went 3.9 -> 3.4 us, the issue of 60 requests per lane cost 1.5 us in front of it, the step was 0.6 us slower.  With the weights
    // already in registers the K loop still waits for y0: the all-gather through memory runs at ~33 GB/s per CU whatever is in flight
    // (DESIGN.md section 5).  Removed; the slot index stays a template constant of the helpers.)
    constexpr int BD = 1;
    half8 wr[BD][T][2][2];
    const uint4* wbase = a.Wa + (size_t)nt * a.ncha * (T * 4) * 256 + tid;
    const uint4* wbase_b = a.Wb + (size_t)nt * KPWB * (T * 4) * 256 + tid;
    auto load_b_tap = [&](const uint4* wb, int ch, int tap, auto slot_c) {
        constexpr int SL = decltype(slot_c)::value;
#if defined(CINDM_ABL) && (CINDM_ABL & 2)
        ch = 0;                                   // ablation build: every stage re-reads the first stage's fragments (L2-hot)
#endif
        const uint4* wp = wb + ((size_t)ch * (T * 4) + tap * 4) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[SL][tap][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
    };
    constexpr std::integral_constant<int, 0> SLOT0{};
#pragma unroll
    for (int tap = 0; tap < T; ++tap) load_b_tap(wbase, 0, tap, SLOT0);
    half8 rreg[2][2];
    const uint4* rbase = a.W2 + (size_t)nt * a.ncha * 4 * 256 + tid;
    auto load_r = [&](int ch) {
        if constexpr (RES) {
            const uint4* wp = rbase + (size_t)ch * 4 * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) rreg[q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
        }
    };
    load_r(0);
    uint4 raw1[KPW1 > 0 ? KPW1 : 1][3][2];
    if constexpr (KPW1 > 0) {
        stg_init(a.src[1], g1);
#pragma unroll
        for (int j = 0; j < KPW1; ++j) load_raw(g1, j, raw1[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue operands of both phases ----------------------------------------------------------------------------
    const int spin_cap = (a.dbg == 9 || err_seen) ? (1 << 8) : (1 << 20);      // (spin_bound's rule)
    // (optional vectors: the POINTER is selected, the load is unconditional.  `p ? p[i] : 0.f` compiles to a load under a
    // branch whose join waits vmcnt(0) before it may write the 0 -- which drains every operand load issued above it)
    const float gam_a = a.gamma_a[gn], bet_a = a.beta_a[gn], gam_b = a.gamma_b[gn], bet_b = a.beta_b[gn];
    const float bias_a_ld = (a.bias_a ? a.bias_a : a.gamma_a)[gn], bias_b_ld = (a.bias_b ? a.bias_b : a.gamma_a)[gn];
    const float tbv_ld = a.tb ? a.tb[(size_t)t_now * a.tb_ld + gn] : bias_a_ld;      // (every dconv2 launch has a time bias: never taken)
    const float bias_a = a.bias_a ? bias_a_ld : 0.f, bias_b = a.bias_b ? bias_b_ld : 0.f, tbv = a.tb ? tbv_ld : 0.f;
    float bias2 = 0.f;
    if constexpr (RES) { const float b2 = (a.bias2 ? a.bias2 : a.gamma_a)[gn]; bias2 = a.bias2 ? b2 : 0.f; }
    float rs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (!RES) {
#pragma unroll
        for (int q = 0; q < 6; ++q) rs[q] = a.res[(size_t)grow[q] * a.ldres + gn];
    }
    __builtin_amdgcn_sched_barrier(0);
    PH(1);                                    // 1 = every prologue load issued (phase A's tile, first weight taps, epilogue operands)

    if constexpr (H > 0) {
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int i = lane; i < KPWM * 4 * 2 * H * S; i += 64) {
            const int blk = i / (2 * H * S), r = i - blk * (2 * H * S);
            const int ks = 4 * (blk >> 2) + w, kq = blk & 3;
            const int row = r < H * S ? r : (L + H) * S + (r - H * S);
            Img[0][(ks * 4 + kq) * RPAD + row] = z;
            Img[1][(ks * 4 + kq) * RPAD + row] = z;
        }
    }

    f32x4 accM[3][2], accL[3][2], accRM[3][2], accRL[3][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };
    zero_acc();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accRM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accRL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // one k-step against all five taps (dconv_kernel's); WB: the weight stream this phase reads; RIDE: the 1x1 on the centre tap
    auto kstep = [&](const uint4* wb, int j, int chn, auto pf, auto ride, auto slot_c) {
        constexpr bool PF = decltype(pf)::value;
        constexpr bool RIDE = decltype(ride)::value;
        constexpr int SL = decltype(slot_c)::value;      // the ring slot this k-step multiplies (and refills with stage chn)
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
                if (p < 0 || p >= NWIN) continue;
#if defined(CINDM_ABL) && (CINDM_ABL & 1)
                // ablation build: no MFMAs; one VALU op per operand register keeps the fragment reads and the weight loads alive
                // (empty asm statements with the operands as inputs: the values must be IN their registers here -- the waits stay,
                // no instruction is issued)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const uint4 u = __builtin_bit_cast(uint4, wr[SL][tap][nb][pl]);
                        asm volatile("" :: "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
                    }
                {
                    const uint4 u = __builtin_bit_cast(uint4, fh[p]), u2 = __builtin_bit_cast(uint4, fl[p]);
                    asm volatile("" :: "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w));
                }
#else
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], wr[SL][tap][nb][0], accM[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], wr[SL][tap][nb][1], accL[mb][nb], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[p], wr[SL][tap][nb][0], accL[mb][nb], 0, 0, 0);
#endif
            }
            if constexpr (RIDE) if (tap == 2) {
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
            if constexpr (PF) {
                // the next stage's fragments of this tap: requested HERE, behind the tap's last use (pinned: left to itself the
                // scheduler gathers the requests at the end of the k-step, where the next k-step's first taps find them late)
                // (round 6 also spread the tap's four requests over its MFMAs -- column block by column block, each fragment requested
                // behind its own last use --: 321.2 -> 321.9 us per step in alternating processes, not kept)
                load_b_tap(wb, chn, tap, slot_c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
    constexpr std::integral_constant<bool, RES> RIDE_A{};
    // ---- phase A K loop; its last k-step requests phase B's first stage of weights instead of nothing ------------------
    // (Round 6 put clocks INSIDE this loop, -DCINDM_KPROF: a k-step's multiplication takes 0.60 - 0.72 us = its 80 KB per CU at the stream's
    // rate; its staging 0.16 us from planes and 0.36 us from fp32 rows -- the split into hi / lo is ~170 VALU instructions per lane.  Staging
    // one to three k-steps AHEAD of the multiplication, or all of them in front of the loop, was built: bitwise equal, 324.7 vs 324.9 us per
    // step with one k-step of lead, slower with three, and 12 - 175 spilled registers with all of them up front.  Not kept.)
    if constexpr (KPW1 == 0) {
#pragma unroll
        for (int j = 0; j < KPW0 - 1; ++j) { store_raw(g0.f32, g0.slot, j, raw0[j]); PHK(2 + 2 * j); kstep(wbase, j, j + 1, YES, RIDE_A, SLOT0); PHK(3 + 2 * j); }
        store_raw(g0.f32, g0.slot, KPW0 - 1, raw0[KPW0 - 1]);
        PHK(2 + 2 * (KPW0 - 1));
        kstep(wbase, KPW0 - 1, 0, NO, RIDE_A, SLOT0);
        PHK(3 + 2 * (KPW0 - 1));
    } else {
#pragma unroll
        for (int j = 0; j < KPW0; ++j) { store_raw(g0.f32, g0.slot, j, raw0[j]); kstep(wbase, j, j + 1, YES, RIDE_A, SLOT0); }
#pragma unroll
        for (int j = 0; j < KPW1 - 1; ++j) { store_raw(g1.f32, g1.slot, j, raw1[j]); kstep(wbase, j, KPW0 + j + 1, YES, RIDE_A, SLOT0); }
        store_raw(g1.f32, g1.slot, KPW1 - 1, raw1[KPW1 - 1]);
        kstep(wbase, KPW1 - 1, 0, NO, RIDE_A, SLOT0);
    }
    PHX(2);                                    // 2 = phase A's K loop done (staging waits + MFMAs)
    // (phase B's first stage of weights is requested further down, where this wave would otherwise idle: requested HERE, the
    // 20 KB per wave sat in front of the epilogue's first LDS writes in the issue queue -- the cross-wave reduction of phase A
    // measured 1.5 us in the replayed step against 0.5 us for the same code in phase B)
    PfRegs pfr;
#pragma unroll
    for (int k = 0; k < PF_REGIONS; ++k) pfr.v[k][0] = pfr.v[k][1] = 0u;
    auto issue_b = [&]() {                    // phase B's first stage of weights
#pragma unroll
        for (int tap = 0; tap < T; ++tap) load_b_tap(wbase_b, 0, tap, SLOT0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- shared epilogue pieces ---------------------------------------------------------------------------------------
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
    constexpr int NSAMP = (S == 16) ? 2 : 1;
    const int gwt = a.gw < TN ? a.gw : TN;
    const float cnt = (float)(L * gwt);
    // GroupNorm + Mish of the reduced tile v (dconv_kernel's, incl. the pair exchange through `xchg` when gw == 64)
    auto gn_mish = [&](const float (&v)[6], unsigned long long* xchg, float gam, float bet, float (&y)[6], auto phb, auto after_xchg) {
        constexpr int PHB = decltype(phb)::value;      // phase marks PHB (own statistics done, published) and PHB + 1 (partner's in)
        (void)PHB;
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
            mean[js] = m; rstd[js] = s2;
        }
        if (a.gw == 64) {
            stress_delay(a.stress, 1u);
            const int sbase = ((mt * a.NT + nt) * 16) * 2, pbase = ((mt * a.NT + (nt ^ 1)) * 16) * 2;
#pragma unroll
            for (int js = 0; js < NSAMP; ++js) {
                const int s = (S == 16) ? rq + 8 * js : rq;
                if (n == 0) {
                    const unsigned long long q0 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, mean[js]);
                    const unsigned long long q1 = ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, rstd[js]);
                    __hip_atomic_store(xchg + sbase + s * 2, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(xchg + sbase + s * 2 + 1, q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            stress_delay(a.stress, 2u);
            PHX(PHB);
            // the partner's granules of BOTH samples in one sweep (one loop per sample was two serial round trips)
            unsigned long long gq[NSAMP][2];
            {
                int spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int js = 0; js < NSAMP; ++js) {
                        const int s = (S == 16) ? rq + 8 * js : rq;
                        gq[js][0] = __hip_atomic_load(xchg + pbase + s * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gq[js][1] = __hip_atomic_load(xchg + pbase + s * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int js = 0; js < NSAMP; ++js) ok = ok & ((unsigned)(gq[js][0] >> 32) == tag) & ((unsigned)(gq[js][1] >> 32) == tag);
                    if (__all(ok)) break;
                    if (++spins > spin_cap) { if (lane == 0) atomicExch(a.err_flag, 1); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
#pragma unroll
            for (int js = 0; js < NSAMP; ++js) {
                const unsigned long long q0 = gq[js][0], q1 = gq[js][1];
                const float mp = __builtin_bit_cast(float, (unsigned)q0), M2p = __builtin_bit_cast(float, (unsigned)q1);
                const float m = 0.5f * (mean[js] + mp);
                const float d0 = mean[js] - m, d1 = mp - m;
                mean[js] = m;
                rstd[js] = (rstd[js] + M2p) + cnt * (d0 * d0 + d1 * d1);
            }
            PHX(PHB + 1);
        }
        after_xchg();                             // (requests that must not sit in front of the exchange's poll in the return queue)
        const float cnt_all = a.gw == 64 ? 2.f * cnt : cnt;
#pragma unroll
        for (int js = 0; js < NSAMP; ++js) rstd[js] = 1.0f / sqrtf(rstd[js] / cnt_all + 1e-5f);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int js = (S == 16) ? (q & 1) : 0;
            y[q] = mish_f((v[q] - mean[js]) * rstd[js] * gam + bet);
        }
    };
    // the tile's planes into LDS `Tile` (row-major [plane][row][16 dwords + 4 pad]; dconv_kernel's DPP pairing)
    constexpr int TP = 20;
    auto planes_to_tile = [&](const float (&y)[6]) {
        uint32_t* tw = reinterpret_cast<uint32_t*>(Tile);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = rq + 8 * q;
            const _Float16 hi = (_Float16)y[q];
            const _Float16 lo = (_Float16)((y[q] - (float)hi) * H3_SCALE);
            const uint32_t own = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
            const uint32_t nbr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);
            const bool odd = n & 1;
            const uint32_t word = odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
            tw[(odd ? 48 * TP : 0) + r * TP + (n >> 1)] = word;
        }
    };

    // ---- phase A epilogue: y0 tile -> planes, published to the column --------------------------------------------------
    float v[6], y[6];
    stress_delay(a.stress, 3u);
    reduce_to(accM, accL, bias_a, v);
    PHX(3);                                    // 3 = cross-wave reduction of phase A (LDS round trip + barrier)
    gn_mish(v, a.xchg_a, gam_a, bet_a, y, std::integral_constant<int, 4>{}, [&]() {});      // 4, 5 = GroupNorm statistics / pair exchange
#pragma unroll
    for (int q = 0; q < 6; ++q) y[q] += tbv;
    planes_to_tile(y);
    stress_delay(a.stress, 4u);
    __syncthreads();
    {
        const uint32_t* tw = reinterpret_cast<const uint32_t*>(Tile);
        const size_t plane_bytes = a.y0_pstride * 16;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.y0), 0, (unsigned)(2 * plane_bytes), 0x00020000);
        auto store_item = [&](int i) {
            const int pl = i >= 192 ? 1 : 0, within = i - pl * 192;
            const int kq = within / 48, row = within - kq * 48;
            const uint4 t4 = *reinterpret_cast<const uint4*>(tw + pl * 48 * TP + row * TP + kq * 4);
            const unsigned off = (unsigned)(pl * plane_bytes + (((size_t)mt * a.NT + nt) * 192 + within) * 16);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t4), rsrc, off, 0, 16);     // aux 16 = sc1: write-through
        };
        for (int i = tid; i < 384; i += 256) store_item(i);
        PHX(6);                                // 6 = Mish, time bias, planes through LDS, write-through stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
        stress_delay(a.stress, 6u);
        __syncthreads();
        if (tid == 0 && !(a.dbg == 9 && (nt & 1)))
            __hip_atomic_store(a.flags + 2 * ((size_t)mt * a.NT + nt), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PHX(7);                                // 7 = stores drained, flag raised
        // phase B's first stage of weights: in flight during the hand-over (flag poll, y0 fetch)
        issue_b();
        if (a.tune & 1) l2_prefetch(a.pf, pfr);       // A/B: round 5's placement of the next launch's warm-up
    }
    // r = Wr x + br stays in registers (Red is free again: the barrier above)
    float r2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (RES) { reduce_to(accRM, accRL, bias2, r2); __syncthreads(); }
    PHX(8);                                    // 8 = the riding 1x1's reduction (RES)

    // ---- hand-over: this wave's k-steps of y0 (k-step ks = the 32 channels of producer nt = ks) -----------------------
    // The fetch runs through a ring of YB k-steps of staging registers (24 VGPRs each): with all KPWB k-steps in registers
    // (96 at C = 512) next to the weight ring (80) the kernel sat at the 256 architectural VGPRs and the compiler shortened
    // live ranges by sinking every weight-fragment load of phase B to its use -- ~30 exposed L2 round trips in this K loop
    // (7 us against phase A's 3 - 4 us; found with the in-replay phase clocks and the ISA).  k-step j + YB is requested when
    // k-step j has gone to LDS; its round trip hides behind YB k-steps of MFMAs.
    constexpr int YB = KPWB < 2 ? KPWB : 2;
    uint4 rawb[YB][3][2];
    int slotb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int item = lane + 64 * i, kq = item / 48, row = item - kq * 48; slotb[i] = kq * RPAD + H * S + row; }
    const size_t y0_plane_bytes = a.y0_pstride * 16;
    const auto y0_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.y0), 0, (unsigned)(2 * y0_plane_bytes), 0x00020000);
    auto fetch_y0 = [&](int j, uint4 (&dst)[3][2]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = (unsigned)((((size_t)mt * a.NT + 4 * j + w) * 192 + lane + 64 * i) * 16);
            dst[i][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(y0_rsrc, off, 0, 16));            // sc1: past L1
            dst[i][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(y0_rsrc, off + (unsigned)y0_plane_bytes, 0, 16));
        }
    };
    {
        stress_delay(a.stress, 7u);
        int spins = 0;
        const int spin_max = spin_cap;
        while (true) {
            // all KPWB flag words requested before the first is looked at (`ok = ok && load(..)` short-circuits: the loads
            // were issued one by one, each behind the previous one's round trip -- up to four serial cross-XCD trips per poll)
            unsigned fw[KPWB];
#pragma unroll
            for (int j = 0; j < KPWB; ++j)
                fw[j] = __hip_atomic_load(a.flags + 2 * ((size_t)mt * a.NT + 4 * j + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int j = 0; j < KPWB; ++j) ok = ok & (fw[j] == tag);
            if (ok) break;                                        // (every lane reads the same words: uniform)
            if (++spins > spin_max) { if (lane == 0) atomicExch(a.err_flag, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        // Compiler-level ordering of the payload loads behind the flag loads (the hardware issues VMEM in program order and the
        // loop's exit depends on the flag values, so no fence INSTRUCTION is needed -- an agent acquire costs ~1.7 us,
        // MI355X_MICROARCH.md -- but nothing else would stop the compiler from hoisting the raw-buffer loads above the relaxed
        // atomic loads of the spin).
        asm volatile("" ::: "memory");
        PHX(9);                                // 9 = the flags of this wave's producers seen
#pragma unroll
        for (int j = 0; j < YB; ++j) fetch_y0(j, rawb[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase B K loop --------------------------------------------------------------------------------------------------
    zero_acc();
    // k-step j multiplies ring slot j % BD and refills it, tap by tap behind each tap's last use, with stage j + BD
    auto kb_step = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr std::integral_constant<int, j % BD> slot{};
        store_raw(false, slotb, j, rawb[j % YB]);
        if (j + YB < KPWB) fetch_y0(j + YB, rawb[j % YB]);
        // k-steps are scheduling regions of their own: in ONE region the scheduler sinks the next stage's weight requests
        // (issued behind each tap's last use, ahead of their consumers) down to those consumers.  Phase A has the
        // same protection by accident -- store_raw's run-time source-kind branch ends a basic block per k-step.
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j + BD < KPWB) kstep(wbase_b, j, j + BD, YES, NO, slot);
        else kstep(wbase_b, j, 0, NO, NO, slot);
        __builtin_amdgcn_sched_barrier(0);
    };
    kb_step(std::integral_constant<int, 0>{});
    if constexpr (KPWB > 1) kb_step(std::integral_constant<int, 1>{});
    if constexpr (KPWB > 2) kb_step(std::integral_constant<int, 2>{});
    if constexpr (KPWB > 3) kb_step(std::integral_constant<int, 3>{});
    static_assert(KPWB <= 4, "phase B: at most four k-steps per wave");
    PHX(10);                                   // 10 = phase B's K loop done (y0 fetch waits + MFMAs)
    // the NEXT launch's L2 warm-up (kernels.h Pf), round 6: issued HERE, a few microseconds before this launch ends.  Through round 5 it
    // was issued behind the publish, i.e. in front of phase B's 1.3 MB per XCD of weights and 1.6 MB of y0: the touched lines did not
    // survive in the 4 MB L2 (n-tiles with and without warm-up streamed equally fast; the option was a net loss)
    if (!(a.tune & 1)) l2_prefetch(a.pf, pfr);

    // ---- phase B epilogue: out = Mish(GN(.)) + (x | r) -------------------------------------------------------------------
    stress_delay(a.stress, 8u);
    reduce_to(accM, accL, bias_b, v);
    PHX(11);                                   // 11 = cross-wave reduction of phase B
    gn_mish(v, a.xchg_b, gam_b, bet_b, y, std::integral_constant<int, 12>{}, [&]() {});     // 12, 13 = statistics / pair exchange
#pragma unroll
    for (int q = 0; q < 6; ++q) y[q] += RES ? r2[q] : rs[q];
    if (a.out_f32 && a.pf.wt) {
        // fp32 rows as 16-byte write-through stores: the quad of lanes that holds columns 4k .. 4k + 3 of rows rq + 8 q transposes
        // its 4 x 4 blocks, lane j then owns row q = j (and lanes 0 / 1 rows q = 4 / 5) of those four columns.  (A 4-byte sc1
        // store is one fabric write each: ~6 x the time per byte of a 16-byte one, MI355X_MICROARCH.md.)
        float ta[4] = {y[0], y[1], y[2], y[3]}, tb[4] = {y[4], y[5], 0.f, 0.f};
        quad_transpose4(ta, lane);
        quad_transpose4(tb, lane);
        const int j = lane & 3;
        const size_t c4 = (size_t)(gn & ~3);
        // (row of q = j: the same formula as grow[], evaluated for this lane's q)
        auto row_ok = [&](int q, int& g) { const int pos = (S == 16) ? (q >> 1) : q; const int sm = (S == 16) ? rq + 8 * (q & 1) : rq; g = (b0 + min(sm, ns - 1)) * L + pos; return sm < ns; };
        int g0; const bool ok0 = row_ok(j, g0);
        if (ok0) st_out4(a.out_f32, (size_t)g0 * a.ldo + c4, make_float4(ta[0], ta[1], ta[2], ta[3]), 1);
        if (j < 2) { int g1; if (row_ok(4 + j, g1)) st_out4(a.out_f32, (size_t)g1 * a.ldo + c4, make_float4(tb[0], tb[1], tb[2], tb[3]), 1); }
    } else {
#pragma unroll
        for (int q = 0; q < 6; ++q)
            if (a.out_f32 && sok[q]) st_out(a.out_f32, (size_t)grow[q] * a.ldo + gn, y[q], 0);
    }
    if (a.out_planes) {
        __syncthreads();                                          // Tile: the publish above has been read
        planes_to_tile(y);
        __syncthreads();
        const uint32_t* tw = reinterpret_cast<const uint32_t*>(Tile);
        for (int i = tid; i < 384; i += 256) {
            const int pl = i >= 192 ? 1 : 0, within = i - pl * 192;
            const int kq = within / 48, row = within - kq * 48;
            const uint4 t4 = *reinterpret_cast<const uint4*>(tw + pl * 48 * TP + row * TP + kq * 4);
            st_out4(a.out_planes, pl * a.out_pstride + ((size_t)mt * a.NT + nt) * 192 + within, t4, a.pf.wt);
        }
    }
    PHX(14);                                   // 14 = Mish, residual, fp32 + planes stores issued
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}


// ---------------------------------------------------------------------------------------------------------------------
// dresample_kernel<UP, KPW>: the two resampling convolutions between the deep levels -- Downsample1d = Conv1d(C, C, 3,
// stride 2, pad 1) from 6 to 3 positions and Upsample1d = ConvTranspose1d(C, C, 4, stride 2, pad 1) from 3 to 6
// (model/diffusion_1d.py:92-106) -- in dconv_kernel's form: 16 samples per workgroup, the whole input tile resident in
// LDS as position-major split-fp16 planes (staged from the attention site's fp32 output, k-step by k-step behind the
// first MFMAs), K split over the four waves, one (output position, tap) pair = one aligned 16-row window, pairs that fall
// into the padding / the wrong parity dropped at compile time (8 of 9 remain going down, 10 of 24 going up).  The output
// leaves as fp32 AND as planes in the NEXT layer's tile format (3 positions x 16 samples, or two tiles of 6 x 8).
// conv_gemm_h3_kernel<3 | 4> needed 11 us per launch for these 0.3 GFLOP: its per-stage restaging, generic tap
// addressing and statistics epilogue are all overhead here.  Weights: pack_weight_h3 (T = 3 / 4).
struct DresArgs {
    const float* x; int ld;                      // input fp32 [sample * Lin + position][ld]
    const uint4* W; const float* bias; int nch;  // stages of 128 input channels
    int Bp, N, NT;
    float* out_f32; int ldo;
    uint4* out_planes; size_t out_pstride;
    Pf pf;
    PhaseBuf ph;
    int xs;                                      // XCDs a column of workgroups spreads over (0: identity mapping)
};

// NH: column split of an n-tile over workgroups -- 1: the workgroup owns all 32 columns (grid NT x tiles = 128 workgroups at 256 rows);
// 2 (round 6): 16 columns, workgroup x = (n-tile x % NT, half x / NT), 256 workgroups: both halves of an n-tile land on the XCD the tile's
// weights are warmed for, each reads half of the tile's fragments (98 -> 49 KB) and the same activation tile (98 KB).
// CAUTION: with NH = 2 two workgroups write the two 64-byte halves of every 128-byte line of out_f32 -- the only place of the path where
// workgroups share output lines, and the best lead for the open two-chain question of DESIGN.md 4.12: the host uses NH = 2 only on the plan
// whose consumers read out_planes (whole lines per workgroup).
template <bool UP, int KPW, int NH>
__global__ __launch_bounds__(256) void dresample_kernel(const DresArgs a) {
    constexpr int T = UP ? 4 : 3, LIN = UP ? 3 : 6, LOUT = UP ? 6 : 3, S = 16;
    constexpr int TNW = TN / NH, NBW = 2 / NH, RQN = 256 / TNW, LDW = TNW + 1;      // columns, 16-column blocks, row groups of the epilogue, Red pitch
    constexpr int RIN = LIN * S, ROUT = LOUT * S, NBLK = LOUT, NQ = ROUT / RQN;
    static_assert(NH == 1 || NH == 2, "32 or 16 columns per workgroup");
    constexpr int KST = 4 * KPW, TP = TNW / 2 + 4;      // TP: words per row and plane in Tile (two columns per word, padded)
    __shared__ uint4 Img[2][KST * 4 * RIN];
    __shared__ float Red[4][ROUT * LDW];
    __shared__ uint4 Tile[2 * ROUT * 5];
    PH_DECL;
    PH(0);                                    // phase clocks (profiling builds): 0 entry, 1 loads issued, 2 K loop, 3 reduce, 4 stores issued
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (round 6, as dconv2_kernel) a column's workgroups spread over a.xs XCDs instead of all 8: its activation tile crosses the fabric a.xs times
    int bx = blockIdx.x, mt = blockIdx.y;
    if (a.xs > 0) {
        const int XS = a.xs, bl = blockIdx.x + gridDim.x * blockIdx.y, xcd = bl & 7, sl = bl >> 3, q4 = gridDim.x / XS;
        bx = (sl % q4) * XS + (xcd & (XS - 1)); mt = (xcd / XS) + (8 / XS) * (sl / q4);
    }
    const int nt = NH == 1 ? bx : bx % a.NT, nbh = NH == 1 ? 0 : bx / a.NT;
    const int b0 = mt * S, ns = min(S, a.Bp - b0);
    const int n = tid & (TNW - 1), rq = tid / TNW, gn = nt * TN + nbh * TNW + n;

    // staging: item = (input row, k-quarter); row = position * 16 + sample; a row's 32 channels are one 128-byte line
    constexpr int NIT = RIN * 4 / 64;
    const float* ibase[NIT];
    int islot[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int item = lane + 64 * i, row = item >> 2, kq = item & 3, sm = row % S, pos = row / S;
        ibase[i] = a.x + (size_t)((b0 + min(sm, ns - 1)) * LIN + pos) * a.ld + kq * 8;
        islot[i] = kq * RIN + row;
    }
    float4 raw[KPW][NIT][2];
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const float4* p4 = reinterpret_cast<const float4*>(ibase[i] + (4 * j + w) * 32);
            raw[j][i][0] = p4[0]; raw[j][i][1] = p4[1];
        }
    half8 breg[T][NBW][2];
    const uint4* wbase = a.W + (size_t)nt * a.nch * (T * 4) * 256 + tid;
    auto load_b_tap = [&](int ch, int tap) {
        const uint4* wp = wbase + ((size_t)ch * (T * 4) + tap * 4) * 256;
#pragma unroll
        for (int q = 0; q < 2 * NBW; ++q) breg[tap][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[(NH == 1 ? q : 2 * nbh + q) * 256]);
    };
#pragma unroll
    for (int tap = 0; tap < T; ++tap) load_b_tap(0, tap);
    const float bias_ld = (a.bias ? a.bias : reinterpret_cast<const float*>(a.W))[gn];      // (pointer selected, load unconditional)
    const float bias = a.bias ? bias_ld : 0.f;
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);
    __builtin_amdgcn_sched_barrier(0);
    PH(1);

    f32x4 accM[NBLK][NBW], accL[NBLK][NBW];
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto kstep = [&](int j, int chn, auto pf) {
        constexpr bool PF = decltype(pf)::value;
        const int kb = (4 * j + w) * 4 * RIN;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const float v[8] = {raw[j][i][0].x, raw[j][i][0].y, raw[j][i][0].z, raw[j][i][0].w, raw[j][i][1].x, raw[j][i][1].y, raw[j][i][1].z, raw[j][i][1].w};
            half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)((v[e] - (float)hi[e]) * H3_SCALE); }
            Img[0][kb + islot[i]] = __builtin_bit_cast(uint4, hi);
            Img[1][kb + islot[i]] = __builtin_bit_cast(uint4, lo);
        }
        const int base = kb + (lane >> 4) * RIN + (lane & 15);
        half8 fh[LIN], fl[LIN];
#pragma unroll
        for (int p = 0; p < LIN; ++p) {
            fh[p] = __builtin_bit_cast(half8, Img[0][base + p * S]);
            fl[p] = __builtin_bit_cast(half8, Img[1][base + p * S]);
        }
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
#pragma unroll
            for (int lo = 0; lo < LOUT; ++lo) {
                // down: x[2 lo - 1 + tap];  up: lo = 2 li - 1 + tap  <=>  li = (lo + 1 - tap) / 2 when that is an integer
                const int num = UP ? lo + 1 - tap : 2 * lo - 1 + tap;
                if (UP && (num & 1)) continue;
                const int p = UP ? num / 2 : num;
                if (num < 0 || p >= LIN) continue;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    accM[lo][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], breg[tap][nb][0], accM[lo][nb], 0, 0, 0);
                    accL[lo][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[p], breg[tap][nb][1], accL[lo][nb], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    accL[lo][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[p], breg[tap][nb][0], accL[lo][nb], 0, 0, 0);
            }
            if constexpr (PF) { load_b_tap(chn, tap); __builtin_amdgcn_sched_barrier(0); }      // pinned behind the tap's last use
        }
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};
#pragma unroll
    for (int j = 0; j < KPW - 1; ++j) kstep(j, j + 1, YES);
    kstep(KPW - 1, 0, NO);
    PH(2);
    l2_prefetch_late(a.pf, pfr);

    // cross-wave K reduction; thread (n, rq) ends with column n of rows rq + RQN q (row = position * 16 + sample)
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                Red[w][(mb * 16 + (lane >> 4) * 4 + rg) * LDW + nb * 16 + (lane & 15)] = accM[mb][nb][rg] + accL[mb][nb][rg] * H3_INV;
    __syncthreads();
    PH(3);
    uint32_t* tw = reinterpret_cast<uint32_t*>(Tile);
    float yq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = rq + RQN * q;
        yq[q] = ((Red[0][r * LDW + n] + Red[1][r * LDW + n]) + (Red[2][r * LDW + n] + Red[3][r * LDW + n])) + bias;
    }
    if (a.pf.wt) {
        // fp32 rows as 16-byte write-through stores: the lane quad transposes 4 x 4 blocks (rows 4 g .. 4 g + 3 of its four columns), lane j
        // stores row q = 4 g + j (as dconv2_kernel's epilogue; a 4-byte sc1 store is one fabric write each)
        const int j = lane & 3;
        const size_t c4 = (size_t)(gn & ~3);
#pragma unroll
        for (int g = 0; g < (NQ + 3) / 4; ++g) {
            float t4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = 4 * g + e < NQ ? yq[(4 * g + e) < NQ ? 4 * g + e : 0] : 0.f;
            quad_transpose4(t4, lane);
            const int q = 4 * g + j;
            const int r = rq + RQN * q, sm = r & 15, pos = r >> 4;
            if (q < NQ && sm < ns) st_out4(a.out_f32, (size_t)((b0 + sm) * LOUT + pos) * a.ldo + c4, make_float4(t4[0], t4[1], t4[2], t4[3]), 1);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = rq + RQN * q, sm = r & 15, pos = r >> 4;
        const float y = yq[q];
        if (!a.pf.wt && sm < ns) st_out(a.out_f32, (size_t)((b0 + sm) * LOUT + pos) * a.ldo + gn, y, 0);
        const _Float16 hi = (_Float16)y;
        const _Float16 lo = (_Float16)((y - (float)hi) * H3_SCALE);
        const uint32_t own = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
        const uint32_t nbr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);
        const bool odd = n & 1;
        const uint32_t word = odd ? ((nbr >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (nbr << 16));
        tw[(odd ? ROUT * TP : 0) + r * TP + (n >> 1)] = word;
    }
    if (a.out_planes) {
        __syncthreads();
        // the next layer's tiles: down -> one tile of 3 positions x 16 samples (rows as here); up -> two tiles of 6 positions x
        // 8 samples (tile 2 mt + (sample >> 3), row position * 8 + (sample & 7)); item = (plane, k-quarter, row) = 16 bytes
        constexpr int KQW = 4 / NH;              // k-quarters (8 channels = one 16-byte item) of this workgroup's columns
        for (int i = tid; i < 2 * KQW * ROUT; i += 256) {
            const int pl = i / (KQW * ROUT), within = i - pl * KQW * ROUT, kq = within / ROUT, r = within - kq * ROUT;
            const uint4 t4 = *reinterpret_cast<const uint4*>(tw + pl * ROUT * TP + r * TP + kq * 4);
            size_t dst;
            if constexpr (!UP) dst = ((size_t)mt * a.NT + nt) * 192 + (nbh * KQW + kq) * 48 + r;
            else {
                const int sm = r & 15, pos = r >> 4, tile = 2 * mt + (sm >> 3);
                if (tile * 8 >= a.Bp) continue;                  // (a ragged batch: the second 8-sample tile does not exist)
                dst = ((size_t)tile * a.NT + nt) * 192 + (nbh * KQW + kq) * 48 + pos * 8 + (sm & 7);
            }
            st_out4(a.out_planes, pl * a.out_pstride + dst, t4, a.pf.wt);
        }
    }
    PH(4);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}


// ---------------------------------------------------------------------------------------------------------------------
// attn1d_head_kernel<C>: Residual(PreNorm(LinearAttentionTemporal)) (model/diffusion_1d.py:75-81, :123-142, :272-291) of
// the deep levels (C = 256 / 512, at most 16 positions per sample group) with the HEADS SPLIT OVER WORKGROUPS.
// attn1d_site_h3_kernel runs a whole site in one workgroup per 4 samples: 64 workgroups, each streaming all of Wqkv and
// Wo (1 MB at C = 512) through one CU's 117 GB/s -- 9 us of pure weight streaming on a quarter of the chip.  Here the
// workgroup (sample group g, head h) streams only head h's q|k|v rows (196 KB) and, after the four heads of a group
// have swapped their 32 x 16 attention tiles through {value, tag} granules (the pair-exchange protocol of
// dconv_kernel, four-way), the output-projection rows [h C/4, (h+1) C/4) (64 KB): 256 workgroups, 4x fewer bytes each.
//   LayerNorm (every head workgroup recomputes it: 16 x C values) -> y planes in LDS
//   q|k|v of head h: K split over the four waves (k32 steps w, w+4, ..), partial accumulators summed through LDS in a
//     fixed order, so every wave ends with the complete q, k, v tiles in the accumulator layouts of attn_site_core
//   core: wave w handles sample w of the group (S = 16 / slot samples, 4-aligned slots as in the site kernel)
//   exchange, then z = Wo[rows of h] att + bo + x
struct AttnHeadArgs {
    const float* x; int ldx;
    float* out; int ldo;
    const float* g;
    const float* Wqkv; const float* Wo; const float* bo;      // the "#site" split-fp16 fragment packings
    int L, S, slot, Bp;
    unsigned long long* xchg; const int* epoch; int* err_flag;
    Pf pf;
    int stress;                                  // > 0: pseudo-random pauses before the hand-overs (stress_delay)
    PhaseBuf ph;
};

template <int C>
__global__ __launch_bounds__(256) void attn1d_head_kernel(const AttnHeadArgs a) {
    constexpr int NP = 16, K32 = C / 32, KPW = K32 / 4, CT4 = C / 64;      // CT4: 16-channel output tiles per head workgroup
    constexpr int YPB = 2 * C + 16;                      // bytes per position per plane
    constexpr int APB = 2 * 128 + 16;
    constexpr int CH = (C + 255) / 256;
    constexpr int RW = NP / 4;
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][NP * YPB];
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][NP * APB];
    __shared__ __attribute__((aligned(16))) float Part[4][24][64];             // per-wave q|k|v partial accumulators
    PH_DECL;
    PH(0);        // phase clocks (profiling builds): 0 entry, 1 weight loads issued, 2 LayerNorm -> planes, 3 q|k|v K share, 4 cross-wave sum,
                  // 5 core, 6 tile published, 7 four heads gathered, 8 att planes in LDS, 9 projection + stores issued
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int hd = blockIdx.x, grp = blockIdx.y;
    const int L = a.L, slot = a.slot;
    const int s_here = min(a.S, a.Bp - grp * a.S);
    const int nend = s_here * slot;
    const size_t row0 = (size_t)grp * a.S * L;
    const unsigned tag = (unsigned)uniform_word(a.epoch);
    const int spin_cap = spin_bound(a.err_flag, false, 18);
    const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
    int tile[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) tile[s] = (s >> 1) * 8 + 2 * hd + (s & 1);
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);          // (the A/B path only -- `tune` bit 0; in front of everything: see below)
    __builtin_amdgcn_sched_barrier(0);
    // ---- the group's rows first (round 4): they come from the previous launch, i.e. from memory, and the LayerNorm needs them
    // before anything else -- requested behind the 56 KB of weight fragments per wave they arrived behind them (loads return
    // in order): the LayerNorm phase measured 3 us in the replayed step ----
    constexpr int LPR = (C / 4 < 64) ? C / 4 : 64;
    constexpr int RPP = 64 / LPR;
    constexpr int NPASS = (RW + RPP - 1) / RPP;
    const int lrow = lane / LPR, lcol = lane % LPR;
    float4 xr[NPASS][CH];
    bool okr[NPASS];
#pragma unroll
    for (int r = 0; r < NPASS; ++r) {
        const int n = w * RW + r * RPP + lrow, sn = n / slot, pn = n - sn * slot;
        okr[r] = (r * RPP + lrow < RW) && n < nend && pn < L;
        const size_t xrow = okr[r] ? row0 + sn * L + pn : row0;          // (address clamped, load unconditional)
#pragma unroll
        for (int m = 0; m < CH; ++m) xr[r][m] = *reinterpret_cast<const float4*>(a.x + xrow * a.ldx + 4 * (lcol + LPR * m));
    }
    float4 gv[CH];
#pragma unroll
    for (int m = 0; m < CH; ++m) gv[m] = *reinterpret_cast<const float4*>(a.g + 4 * (lcol + LPR * m));
    // this wave's k32 steps of the head's six tiles: everything in flight at once (KPW * 12 KiB per wave)
    // (round 6) ... in two halves: the first in front of the LayerNorm (it is on its way while the rows are), the rest k-step by k-step BETWEEN
    // the LayerNorm's passes: a wave issues in order and 48 requests take ~2 us to issue against the back-pressure of the L1 path -- with all
    // of them in front, the LayerNorm's 1.6 us of VALU work started when the last one was issued.
    float4 wr[KPW][6][2];
    auto load_wq = [&](int j) {
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                wr[j][s][pl] = Wq4[(((size_t)tile[s] * K32 + (4 * j + w)) * 2 + pl) * 64 + lane];
    };
    constexpr int KFRONT = (KPW + 1) / 2;
#pragma unroll
    for (int j = 0; j < KFRONT; ++j) load_wq(j);
    // (round 6) NOTHING else is requested in front of the LayerNorm: with the 16 output-projection fragments and the A/B path's conditional
    // touches here the wave had 74+ requests outstanding -- more than vmcnt can count -- behind a region with an unknown number of loads, and
    // the LayerNorm's wait for the group's rows (the OLDEST requests) was a vmcnt(0): it started when the last weight fragment had arrived
    // (tools/isa_audit.py).  58 requests now, the rows' wait is exact, the LayerNorm runs while the q | k | v fragments stream in.
    __builtin_amdgcn_sched_barrier(0);
    PH(1);

    // ---- LayerNorm of the group's positions -> split-fp16 planes (as attn1d_site_h3_kernel) ----
    {
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int n = w * RW + r * RPP + lrow;
            if (!okr[r]) {
#pragma unroll
                for (int m = 0; m < CH; ++m) xr[r][m] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s1 = 0.f;
#pragma unroll
            for (int m = 0; m < CH; ++m) s1 += (xr[r][m].x + xr[r][m].y) + (xr[r][m].z + xr[r][m].w);
            s1 = rowgroup_sum<LPR>(s1);
            const float mean = s1 * (1.0f / C);
            float s2 = 0.f;
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const float d0 = xr[r][m].x - mean, d1 = xr[r][m].y - mean, d2 = xr[r][m].z - mean, d3 = xr[r][m].w - mean;
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            s2 = rowgroup_sum<LPR>(s2);
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            if (r * RPP + lrow < RW) {
#pragma unroll
                for (int m = 0; m < CH; ++m) {
                    float4 y;
                    y.x = (xr[r][m].x - mean) * rstd * gv[m].x; y.y = (xr[r][m].y - mean) * rstd * gv[m].y;
                    y.z = (xr[r][m].z - mean) * rstd * gv[m].z; y.w = (xr[r][m].w - mean) * rstd * gv[m].w;
                    if (!okr[r]) y = make_float4(0.f, 0.f, 0.f, 0.f);
                    half4v hi, lo;
                    hi[0] = (_Float16)y.x; hi[1] = (_Float16)y.y; hi[2] = (_Float16)y.z; hi[3] = (_Float16)y.w;
                    lo[0] = (_Float16)((y.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y.y - (float)hi[1]) * H3_SCALE);
                    lo[2] = (_Float16)((y.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y.w - (float)hi[3]) * H3_SCALE);
                    const int off = n * YPB + 8 * (lcol + LPR * m);
                    *reinterpret_cast<half4v*>(&Yp[0][off]) = hi;
                    *reinterpret_cast<half4v*>(&Yp[1][off]) = lo;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (KFRONT + r < KPW) load_wq(KFRONT + r);
            if (r == NPASS - 1) {
#pragma unroll
                for (int j = KFRONT + NPASS; j < KPW; ++j) load_wq(j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // output-projection fragments of this wave's tiles (rows hd * C/4 + ..): CT4 tiles per workgroup, CT4 / 4 per wave
    constexpr int TPW = CT4 / 4;
    static_assert(CT4 % 4 == 0, "output tiles split evenly over the waves");
    const float4* Wo4 = reinterpret_cast<const float4*>(a.Wo);
    float4 wo[TPW][4][2];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                wo[t][k][pl] = Wo4[(((size_t)(hd * CT4 + w * TPW + t) * 4 + k) * 2 + pl) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PH(2);

    // ---- this wave's K share of q, k, v of head hd ----
    f32x4 qM[2], qL[2], kM[2], kL[2], vM[2], vL[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        qM[i] = f32x4{0.f, 0.f, 0.f, 0.f}; qL[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        kM[i] = f32x4{0.f, 0.f, 0.f, 0.f}; kL[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        vM[i] = f32x4{0.f, 0.f, 0.f, 0.f}; vL[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k32 = 4 * j + w;
        const int off = lr * YPB + k32 * 64 + lq * 16;
        const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
        const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const half8 qh = __builtin_bit_cast(half8, wr[j][i][0]), ql = __builtin_bit_cast(half8, wr[j][i][1]);
            const half8 kh = __builtin_bit_cast(half8, wr[j][2 + i][0]), kl = __builtin_bit_cast(half8, wr[j][2 + i][1]);
            const half8 vh = __builtin_bit_cast(half8, wr[j][4 + i][0]), vl = __builtin_bit_cast(half8, wr[j][4 + i][1]);
            qM[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh, yh, qM[i], 0, 0, 0);
            qL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh, yl, qL[i], 0, 0, 0);
            qL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql, yh, qL[i], 0, 0, 0);
            kM[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, kh, kM[i], 0, 0, 0);
            kL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, kl, kL[i], 0, 0, 0);
            kL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, kh, kL[i], 0, 0, 0);
            vM[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, vh, vM[i], 0, 0, 0);
            vL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, vl, vL[i], 0, 0, 0);
            vL[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, vh, vL[i], 0, 0, 0);
        }
    }
    PH(3);
    // cross-wave sum in the fixed order (w0 + w1) + (w2 + w3): every wave ends with the complete tiles
    stress_delay(a.stress, 13u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Part[w][i * 4 + r][lane] = qM[i][r] + qL[i][r] * H3_INV;
            Part[w][8 + i * 4 + r][lane] = kM[i][r] + kL[i][r] * H3_INV;
            Part[w][16 + i * 4 + r][lane] = vM[i][r] + vL[i][r] * H3_INV;
        }
    __syncthreads();
    f32x4 qa[2][1], ka[1][2], va[1][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qa[i][0][r] = (Part[0][i * 4 + r][lane] + Part[1][i * 4 + r][lane]) + (Part[2][i * 4 + r][lane] + Part[3][i * 4 + r][lane]);
            ka[0][i][r] = (Part[0][8 + i * 4 + r][lane] + Part[1][8 + i * 4 + r][lane]) + (Part[2][8 + i * 4 + r][lane] + Part[3][8 + i * 4 + r][lane]);
            va[0][i][r] = (Part[0][16 + i * 4 + r][lane] + Part[1][16 + i * 4 + r][lane]) + (Part[2][16 + i * 4 + r][lane] + Part[3][16 + i * 4 + r][lane]);
        }

    PH(4);
    // the epilogue's bias and residual rows, requested before the core and the exchange (requested in the epilogue, their L2
    // round trip was the tail of the launch)
    float4 eb[TPW], ex[TPW];
    {
        const int n = lr, sn = n / slot, pn = n - sn * slot;
        const bool ok = n < nend && pn < L;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = (hd * CT4 + w * TPW + t) * 16 + lq * 4;
            eb[t] = *reinterpret_cast<const float4*>(a.bo + c);
            ex[t] = ok ? *reinterpret_cast<const float4*>(a.x + (row0 + sn * L + pn) * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    l2_prefetch_late(a.pf, pfr);                         // (behind the kernel's last load request)
    // ---- core: wave w owns sample w of the group ----
    f32x4 att[2][1];
    attn_site_core_range<1>(qa, ka, va, att, w, min(w + 1, s_here), nend, slot, L, lq, lr);
    PH(5);
    stress_delay(a.stress, 11u);
    // publish this wave's sample columns (positions [w * slot, (w + 1) * slot)) of the head's 32 x 16 tile, and keep a
    // copy for the own projection.  Granule (hd, e, n) of group grp: value att[e][n], tag = epoch.
    unsigned long long* gx = a.xchg + (size_t)grp * 4 * 512;
    const bool mycol = (lr / slot) == w && w < a.S;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = et * 16 + lq * 4 + i;
            if (mycol) {
                const float val = att[et][0][i];
                __hip_atomic_store(gx + hd * 512 + e * 16 + lr, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, val),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    // columns of samples beyond S (slot * S < 16) belong to nobody: publish zeros once (wave 0)
    if (w == 0 && lr >= a.S * slot) {
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __hip_atomic_store(gx + hd * 512 + (et * 16 + lq * 4 + i) * 16 + lr, (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PH(6);
    // gather all four heads' tiles (own head included: one code path) -> att planes [position][128 channels]
    stress_delay(a.stress, 12u);
    {
        unsigned long long gq[8];
        int spins = 0;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gq[j] = __hip_atomic_load(gx + tid + 256 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(gq[j] >> 32) == tag;
            }
            if (__all(ok)) break;
            if (++spins > spin_cap) { if (lane == 0) atomicExch(a.err_flag, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        PH(7);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 256 * j, h2 = idx >> 9, e = (idx >> 4) & 31, n = idx & 15;
            const float val = __builtin_bit_cast(float, (unsigned)gq[j]);
            const _Float16 hi = (_Float16)val;
            const _Float16 lo = (_Float16)((val - (float)hi) * H3_SCALE);
            const int off = n * APB + 2 * (h2 * 32 + e);
            *reinterpret_cast<_Float16*>(&Ap[0][off]) = hi;
            *reinterpret_cast<_Float16*>(&Ap[1][off]) = lo;
        }
    }
    __syncthreads();
    PH(8);

    // ---- out rows [hd * C/4, (hd + 1) * C/4): z = Wo att + bo + x ----
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int ct = hd * CT4 + w * TPW + t;
        f32x4 zM = f32x4{0.f, 0.f, 0.f, 0.f}, zL = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half8 wh = __builtin_bit_cast(half8, wo[t][k][0]), wl = __builtin_bit_cast(half8, wo[t][k][1]);
            const int off = lr * APB + k * 64 + lq * 16;
            const half8 ah = *reinterpret_cast<const half8*>(&Ap[0][off]);
            const half8 al = *reinterpret_cast<const half8*>(&Ap[1][off]);
            zM = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, zM, 0, 0, 0);
            zL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, zL, 0, 0, 0);
            zL = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, zL, 0, 0, 0);
        }
        const int c = ct * 16 + lq * 4;
        const float4 b = eb[t];
        const f32x4 z = zM + zL * H3_INV;
        const int n = lr, sn = n / slot, pn = n - sn * slot;
        if (n < nend && pn < L) {
            const size_t row = row0 + sn * L + pn;
            const float4 xv = ex[t];
            float4 o;
            o.x = z[0] + b.x + xv.x; o.y = z[1] + b.y + xv.y; o.z = z[2] + b.z + xv.z; o.w = z[3] + b.w + xv.w;
            st_out4(a.out, row * a.ldo + c, o, a.pf.wt);
        }
    }
    PH(9);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

}  // namespace cindm
