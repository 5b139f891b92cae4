// conv2d_ws_kernel<KIND, MODE>: second generation of the 2-D 3x3 convolution (model/diffusion_2d.py:182-198 Block.proj,
// :99-103 Upsample's conv over the nearest-x2 source) on the split-fp16 products of kernels.h.  gfx950 only.
//
// What round 2's profile of conv2d_h3_kernel showed (profiles/r02_cfg5_phases.txt): prologue (load + split, 1.0 ms per
// step), main loop (2.2 ms) and epilogue (0.8 ms) ADD UP -- two co-resident workgroups of 4 waves do not overlap their
// memory phases with each other's MFMA phase; the main loop itself is short (1.4 us of matrix work per 64-pixel tile)
// against ~5 us of load latency + split + store per tile.  A larger tile in the same structure (8 x 16 pixels, measured:
// 165 -> 148 us per 64 -> 64 layer at 64 x 64) does not change that.  Hence a persistent, wave-specialised kernel:
//
//  * one workgroup of 8 waves per CU, 256 workgroups, each looping over its list of (pixel tile, n-tile, chunk) items;
//  * waves 0-3 ("matrix waves", one per SIMD) only multiply: output tile 8 x 16 pixels x 64 channels, wave = (k-group,
//    column half), 8 pixel blocks per wave (every B fragment fetched from L2 feeds 16 MFMA triples), A fragments from
//    the LDS planes one pixel block at a time, two blocks ahead, B in a 3-tap register ring that runs across items; the
//    accumulators are defined by tap 0 of a tile's first chunk (zero C operand), so nothing is initialised;
//  * waves 4-7 ("memory waves", one per SIMD) do everything else WHILE the matrix waves multiply item k, two items deep:
//    normalise / SiLU / split item k+1's window (10 x 18 pixels x 64 channels, loaded during phase k-1) into the other LDS
//    plane buffer, THEN issue item k+2's global loads into the same registers, THEN write the finished tile of the
//    previous item from LDS to HBM as float4 rows (+ its GroupNorm partials).  vmcnt retires in order, so with the
//    stores issued after the loads no wait for a load ever covers a store (with the stores first, staging waited for
//    HBM write acknowledgements: 138 -> 123 us per layer);
//  * three workgroup barriers per finished tile: [multiply | stage-load-store] -> kg = 1 waves park their partial tile in
//    LDS -> kg = 0 waves add theirs + bias in place -> next item;
//  * GroupNorm statistics travel as per-(tile, memory wave) partials (32 pixels each; no cross-wave reduction in the
//    store path) and the consumer merges them itself -- the gn_merge_kernel launches are gone.  Inside this kernel the
//    merge uses DPP row sums + v_readlane + a select chain (group8_total), NOT ds_bpermute shuffles: with __shfl in the
//    memory waves, one run in ~5 staged a stale window pixel (3 wrong output pixels per incident; 40 repeats of one
//    forward in tests/test_gpu_paths.py::test_unet2d_conv_ws_repeatable is the regression test); without LDS-pipeline
//    instructions in that path, 0 of 450 runs;
//  * XCD x (workgroup % 8) owns a contiguous eighth of the tile list, its 32 workgroups walk it side by side, so halos
//    and both n-tiles' input hit that XCD's L2.
// LDS: 2 plane buffers x 51 840 B + 34 816 B output tile = 138.5 KB.
//
// Measured (config 5, 128 images, wall_clock64 inside the kernel, per 64 x 64 layer of 16 tiles per workgroup): plain
// source: multiply 78-82 us (3.6 us per tile when a workgroup has ONE tile, 5 us with the memory waves active),
// reduce 28 us, barrier wait 9 us; memory waves 85 us of work.  GroupNorm + SiLU on load: the staging VALU /
// transcendental work (84 us) is the long pole and the matrix waves wait 55 us -- s_setprio on either side changes
// nothing.  Layer times: 64 -> 64 plain 165 -> 123 us, GroupNorm 197 -> 176 us, 128 -> 64 240 -> 198 us.
#pragma once
#include "kernels2d.h"
#include <type_traits>

namespace cindm {

constexpr int V2Y = 8, V2X = 16, V2M = V2Y * V2X;          // output pixel tile
constexpr int V2SW = 18, V2R = 10 * V2SW;                   // staged window 10 x 18 pixels
constexpr int V2PITCH = 144, V2PLANE = V2R * V2PITCH;       // bytes per staged pixel and plane (64 halfs + 16 B pad).  Round 5: under gfx950's
                                                            // ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...; tools/lds_bank_model.py) this pitch makes the fragment
                                                            // reads 2-way conflicted (SQ_LDS_BANK_CONFLICT 45 % of SQ_LDS_IDX_ACTIVE, profiles/r05_lds_conflicts_before.txt);
                                                            // a pitch of 160 removes them (3.4 %) and the kernel is NOT faster (same-box A/B, profiles/r05_ab_lds_pitch.txt):
                                                            // the matrix waves do not wait for the LDS.  144 stays (less LDS).
constexpr int V2LDT = 132;                                  // output tile [64 channels][128 pixels + 4]: pitch in floats
constexpr int WS_GRID = 256;                                // one persistent workgroup per CU
constexpr int WS_SPT = 4;                                   // GroupNorm partials per tile (one per memory wave, 32 pixels)
constexpr int WS_MAXP = 128;                                // most GroupNorm partials per (image, group) the consumer side holds

// In-kernel clocks of conv2d_ws_kernel -- profiling build only (CINDM_PHASE_PROF).  Workgroup WSP_WG's first matrix wave and first
// memory wave add the time they spend in each phase (100 MHz wall clock) into g_ws_prof[category][role][phase]; category =
// (GroupNorm-on-load source) + 2 * (more than one 64-channel chunk) + 4 * (input-gradient mode); role 0 = matrix wave (phases:
// multiply, wait S1, reduce / tile write, wait S3, -, -, -, items), role 1 = memory wave (stage, issue loads, write tile, wait S1,
// wait S2 / S3, -, -, items).  cindm_ws_prof_read() copies and clears it; tools/ws_prof.py prints it.
constexpr int WSP_CAT = 8, WSP_NPH = 8, WSP_WG = 8;
#ifdef CINDM_PHASE_PROF
__device__ unsigned long long g_ws_prof[WSP_CAT][2][WSP_NPH];
#define WSP_DECL unsigned long long wsp_[WSP_NPH] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; unsigned long long wsp_t_ = wall_clock64()
#define WSP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); wsp_[i] += n_ - wsp_t_; wsp_t_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define WSP_COUNT() do { wsp_[7] += 1; } while (0)
#define WSP_FLUSH(role_) do { if (blockIdx.x == WSP_WG && lw == 0 && lane == 0) { \
        const int cat_ = (MODE == SRC2_GN_SS_SILU ? 1 : 0) + (nch > 1 ? 2 : 0) + (MODE == SRC2_SCALED ? 4 : 0); \
        _Pragma("unroll") for (int i_ = 0; i_ < WSP_NPH; ++i_) atomicAdd(&g_ws_prof[cat_][role_][i_], wsp_[i_]); } } while (0)
#else
#define WSP_DECL do { } while (0)
#define WSP(i) do { } while (0)
#define WSP_COUNT() do { } while (0)
#define WSP_FLUSH(role_) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
// VAR: how the matrix waves divide an item.  0 = round 2's k-groups (wave = k half x column half, reduction through LDS per tile);
// 1 = no split (wave = pixel half x column half, all 64 channels of the chunk).  Round 5 also measured the same on
// v_mfma_f32_32x32x16_f16 (half as many matrix instructions per item: 5.35 -> 5.45 ms per step) and the transposed product with the
// store path inside the matrix waves (no LDS tile, one barrier per item: 6.4 -> 7.4 us per item) -- both slower, both removed in
// round 6 (DESIGN.md section 4.5b keeps the measurements).
template <int KIND, int MODE, int VAR = 0>
__global__ __launch_bounds__(512) void conv2d_ws_kernel(const Conv2dArgs a) {
    constexpr bool KSPLIT = VAR == 0;
    // KIND CONV_3X3_PAIR (ForceUnet's 8 x 8 level): the images are 8 pixels wide and a tile is rows ty0 .. ty0 + 7 of images
    // 2j | 2j + 1 side by side (a.NI counts PAIRS).  Each half keeps its own zero columns: the window is 10 x 20 pixels,
    // [pad A0..A7 pad | pad B0..B7 pad], and the right half's fragment / staging addresses are shifted by two pixels.
    constexpr bool PAIR = KIND == CONV_3X3_PAIR;
    constexpr int SW = PAIR ? 20 : V2SW, R = 10 * SW, PLANE = R * V2PITCH;
    constexpr int KC = 64, NP = (R + 15) / 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * PLANE];   // [buffer][plane hi/lo][R][V2PITCH]
    __shared__ __attribute__((aligned(16))) float Tile[T2N * V2LDT];   // channel-major: an accumulator's 4 rows are 16 contiguous bytes

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = w >> 2, lw = w & 3;
    // ---- this workgroup's items -----------------------------------------------------------------------------------
    // The unit of work a workgroup owns is a pixel tile with all its n-tiles (the staged planes serve them all) -- except
    // in PAIR kind, where the unit is (pixel tile, n-tile): 384 tiles of 8 n-tiles x 8 chunks on 256 workgroups left a
    // third of them with twice the work; 3072 units spread evenly (with 4-8 chunks per tile no plane is reused anyway).
    constexpr bool NSPLIT = PAIR;
    const int ntiles = a.Npad / T2N, nch = a.CinP / KC, ipt = NSPLIT ? nch : ntiles * nch;      // items per unit
    const int MT = a.NI * a.tpi * (NSPLIT ? ntiles : 1);
    const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int lo = (int)(((long long)xcd * MT) >> 3), hi = (int)(((long long)(xcd + 1) * MT) >> 3);
    const int mine = hi - lo > wj ? (hi - lo - wj + wpx - 1) / wpx : 0;
    if (mine == 0) return;
    const int nitems = mine * ipt;
    const int HWi = a.Hin * a.Win;
    const int nch0 = (a.src[0].C + KC - 1) / KC;
    auto decode = [&](int k, int& mt, int& nt, int& ch) {
        const int tl = k / ipt, rem = k - tl * ipt;
        if constexpr (NSPLIT) { const int u = lo + wj + tl * wpx; mt = u / ntiles; nt = u - mt * ntiles; ch = rem; }
        else { nt = rem / nch; ch = rem - nt * nch; mt = lo + wj + tl * wpx; }
    };
    auto nt_of = [&](int tl) { return (lo + wj + tl * wpx) % ntiles; };      // NSPLIT: the n-tile of this workgroup's unit tl
    // The memory waves walk the items in order: a cursor that steps (chunk -> n-tile -> unit) replaces decode(k)'s divisions
    // (round 5: load_item + decode were 127 scalar and 27 vector instructions per item, most of them integer divisions by run-time
    // values, in a wave whose time goes to instruction issue -- one wave per SIMD beside the matrix wave).
    struct ItemCur { int tl, mt, nt, ch; };
    auto cur_first = [&](ItemCur& c) { decode(0, c.mt, c.nt, c.ch); c.tl = 0; };
    auto cur_next = [&](ItemCur& c) {
        if (++c.ch < nch) return;
        c.ch = 0;
        if constexpr (NSPLIT) { ++c.tl; const int u = lo + wj + c.tl * wpx; c.mt = u / ntiles; c.nt = u - c.mt * ntiles; }
        else { if (++c.nt < ntiles) return; c.nt = 0; ++c.tl; c.mt += wpx; }
    };
    // image / tile coordinates of a unit: shifts when the tile counts are powers of two (they are for power-of-two images)
    const int tpi_sh = (a.tpi & (a.tpi - 1)) == 0 ? 31 - __builtin_clz(a.tpi) : -1;
    const int tx_sh = (a.tiles_x & (a.tiles_x - 1)) == 0 ? 31 - __builtin_clz(a.tiles_x) : -1;

    if (role == 0 && VAR == 1) {
        // ================================ matrix waves, K not split over the waves (round 5) ======================
        // wave = (pixel half ph: pixel blocks 4 ph .. 4 ph + 3, column half nh), ALL 64 channels of the chunk: the k-group
        // reduction of the variant below (kg = 1 parks its partial tile in LDS, barrier, kg = 0 adds, barrier: 1.75 us of a
        // 7.3 us tile in the in-kernel clocks of profiles/r02_cfg5_phases.txt) does not exist; the price is that the two waves
        // of a column half fetch the same weight fragments (8 instead of 4 16-byte loads per tap and wave).  Same weight pack:
        // the fragments of k-group kh live at thread (kh + 2 nh) * 64 + lane.
        const int ph = lw & 1, nh = lw >> 1;
        if (a.dbg == 11 || (a.dbg != 10 && a.dbg != 12 && MODE != SRC2_GN_SS_SILU)) __builtin_amdgcn_s_setprio(3);
        f32x4 accM[4][2], accL[4][2];
        half8 breg[3][2][2][2];                               // [ring slot][k half][column block][plane]
        const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + nh * 128 + lane;
        auto load_b = [&](int nt, int ch, int tap, int slot_) {
            const uint4* wp = wbase + ((size_t)(nt * nch + ch) * 9 + tap) * 4 * 256;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int q = 0; q < 4; ++q) breg[slot_][kh][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256 + kh * 64]);
        };
        const int foff = ((lane & 15) + (PAIR && (lane & 15) >= 8 ? 2 : 0)) * V2PITCH + (lane >> 4) * 16;
        // 72 steps (tap, k half, pixel block); fragments of step s + 2 are read while step s multiplies
        auto compute = [&](const unsigned char* P0, int nt, int ch, int nt2, int ch2, auto FIRST_) {
            constexpr bool FIRST = decltype(FIRST_)::value;
            const unsigned char* P1 = P0 + PLANE;
            half8 fh[3], fl[3];
            auto read_frag = [&](int s, int slot_) {
                const int tap = s >> 3, kh = (s >> 2) & 1, mb = s & 3;
                const int dy = tap / 3, dx = tap - dy * 3;
                const int o = ((4 * ph + mb + dy) * SW + dx) * V2PITCH + kh * 64;
                fh[slot_] = *reinterpret_cast<const half8*>(P0 + o);
                fl[slot_] = *reinterpret_cast<const half8*>(P1 + o);
            };
            read_frag(0, 0); read_frag(1, 1);
#pragma unroll
            for (int s = 0; s < 72; ++s) {
                const int tap = s >> 3, kh = (s >> 2) & 1, mb = s & 3, bs = tap % 3, fs = s % 3;
                if (s + 2 < 72) read_frag(s + 2, (s + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bool z = FIRST && tap == 0 && kh == 0;
#define WS_MMA(X_, W_, C_) __builtin_amdgcn_mfma_f32_16x16x32_f16(X_, W_, C_, 0, 0, 0)
                accM[mb][0] = WS_MMA(fh[fs], breg[bs][kh][0][0], z ? zero : accM[mb][0]);
                accL[mb][0] = WS_MMA(fh[fs], breg[bs][kh][0][1], z ? zero : accL[mb][0]);
                accM[mb][1] = WS_MMA(fh[fs], breg[bs][kh][1][0], z ? zero : accM[mb][1]);
                accL[mb][1] = WS_MMA(fh[fs], breg[bs][kh][1][1], z ? zero : accL[mb][1]);
                accL[mb][0] = WS_MMA(fl[fs], breg[bs][kh][0][0], accL[mb][0]);
                accL[mb][1] = WS_MMA(fl[fs], breg[bs][kh][1][0], accL[mb][1]);
#undef WS_MMA
                if ((s & 7) == 7) {                          // this slot's next tap (of this or the next item), two taps ahead
                    if (tap + 3 < 9) load_b(nt, ch, tap + 3, bs);
                    else load_b(nt2, ch2, tap + 3 - 9, bs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        { const int nt0 = NSPLIT ? nt_of(0) : 0; load_b(nt0, 0, 0, 0); load_b(nt0, 0, 1, 1); load_b(nt0, 0, 2, 2); }
        stress_delay(a.stress, 100u);
        __syncthreads();                                     // S0: item 0 is staged
        WSP_DECL;
        int k = 0;
        for (int tl = 0; tl < mine; ++tl)
            for (int nt = NSPLIT ? nt_of(tl) : 0, nt_end = NSPLIT ? nt + 1 : ntiles; nt < nt_end; ++nt) {
                const int nt_after = NSPLIT ? (tl + 1 < mine ? nt_of(tl + 1) : 0) : (nt + 1 < ntiles ? nt + 1 : 0);
                {
                    const int nt2 = nch > 1 ? nt : nt_after, ch2 = nch > 1 ? 1 : 0;
                    if (a.dbg != 3) compute(&smem[k & 1][0] + foff, nt, 0, nt2, ch2, std::true_type{});
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    }
                    ++k;
                    WSP(0); WSP_COUNT();
                    stress_delay(a.stress, 101u + 8u * (unsigned)k);
                    __syncthreads();                         // S1: planes consumed; the memory waves are done with Tile
                    WSP(1);
                }
                for (int ch = 1; ch < nch; ++ch) {
                    const int nt2 = ch + 1 < nch ? nt : nt_after, ch2 = ch + 1 < nch ? ch + 1 : 0;
                    if (a.dbg != 3) compute(&smem[k & 1][0] + foff, nt, ch, nt2, ch2, std::false_type{});
                    ++k;
                    WSP(0); WSP_COUNT();
                    stress_delay(a.stress, 102u + 8u * (unsigned)k);
                    __syncthreads();                         // S1
                    WSP(1);
                }
                // every wave writes its finished 64 pixels x 32 channels (+ bias) into the channel-major tile: one ds_write_b128
                // per (pixel block, column block)
                float* const trow = Tile + (nh * 32 + (lane & 15)) * V2LDT + (lane >> 4) * 4 + ph * 64;
                float bias[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int gn = nt * T2N + nh * 32 + (lane & 15) + nb * 16;
                    bias[nb] = (a.bias && gn < a.N) ? a.bias[gn] : 0.f;
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        float4 v;
                        v.x = (accM[mb][nb][0] + accL[mb][nb][0] * H3_INV) + bias[nb]; v.y = (accM[mb][nb][1] + accL[mb][nb][1] * H3_INV) + bias[nb];
                        v.z = (accM[mb][nb][2] + accL[mb][nb][2] * H3_INV) + bias[nb]; v.w = (accM[mb][nb][3] + accL[mb][nb][3] * H3_INV) + bias[nb];
                        *reinterpret_cast<float4*>(trow + nb * 16 * V2LDT + mb * 16) = v;
                    }
                WSP(2);
                stress_delay(a.stress, 104u + 8u * (unsigned)k);
                __syncthreads();                             // S3: the finished tile is in LDS
                WSP(3);
            }
        WSP_FLUSH(0);
        return;
    }
    if (role == 0) {
        // ===================================== matrix waves, K split over the waves ================================
        const int kg = lw & 1, nh = lw >> 1;
        if (MODE != SRC2_GN_SS_SILU) __builtin_amdgcn_s_setprio(3);   // plain source: the matrix pipe is the long pole -> it goes first
        f32x4 accM[8][2], accL[8][2];
        half8 breg[3][2][2];
        // B: [n-tile][chunk][tap][q = nb*2 + plane][thread][8 halfs] (the pack of conv2d_h3_kernel: same wave roles)
        const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + lw * 64 + lane;
        auto load_b = [&](int nt, int ch, int tap, int slot_) {
            const uint4* wp = wbase + ((size_t)(nt * nch + ch) * 9 + tap) * 4 * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) breg[slot_][q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
        };
        const int foff = ((lane & 15) + (PAIR && (lane & 15) >= 8 ? 2 : 0)) * V2PITCH + kg * 64 + (lane >> 4) * 16;
        // 72 steps (tap, pixel block); fragments of step s + 2 are read while step s multiplies
        auto compute = [&](const unsigned char* P0, int nt, int ch, int nt2, int ch2, auto FIRST_) {
            constexpr bool FIRST = decltype(FIRST_)::value;
            const unsigned char* P1 = P0 + PLANE;
            half8 fh[3], fl[3];
            auto read_frag = [&](int s, int slot_) {
                const int tap = s >> 3, mb = s & 7;
                const int dy = tap / 3, dx = tap - dy * 3;
                fh[slot_] = *reinterpret_cast<const half8*>(P0 + ((mb + dy) * SW + dx) * V2PITCH);
                fl[slot_] = *reinterpret_cast<const half8*>(P1 + ((mb + dy) * SW + dx) * V2PITCH);
            };
            read_frag(0, 0); read_frag(1, 1);
#pragma unroll
            for (int s = 0; s < 72; ++s) {
                const int tap = s >> 3, mb = s & 7, bs = tap % 3, fs = s % 3;
                if (s + 2 < 72) read_frag(s + 2, (s + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bool z = FIRST && tap == 0;
                // the two products into accL of one column block are kept three instructions apart
                accM[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][0][0], z ? zero : accM[mb][0], 0, 0, 0);
                accL[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][0][1], z ? zero : accL[mb][0], 0, 0, 0);
                accM[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][1][0], z ? zero : accM[mb][1], 0, 0, 0);
                accL[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[fs], breg[bs][1][1], z ? zero : accL[mb][1], 0, 0, 0);
                accL[mb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[fs], breg[bs][0][0], accL[mb][0], 0, 0, 0);
                accL[mb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[fs], breg[bs][1][0], accL[mb][1], 0, 0, 0);
                if (mb == 7) {                               // this slot's next tap (of this or the next item), two taps ahead
                    if (tap + 3 < 9) load_b(nt, ch, tap + 3, bs);
                    else load_b(nt2, ch2, tap + 3 - 9, bs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        { const int nt0 = NSPLIT ? nt_of(0) : 0; load_b(nt0, 0, 0, 0); load_b(nt0, 0, 1, 1); load_b(nt0, 0, 2, 2); }
        stress_delay(a.stress, 100u);
        __syncthreads();                                     // S0: item 0 is staged
        WSP_DECL;
        // tile -> n-tile -> chunk: the same item order as decode(); nested so that the accumulators are defined by the
        // first chunk's tap 0 (zero C operand), updated by the other chunks and consumed by the reduce below
        int k = 0;
        for (int tl = 0; tl < mine; ++tl)
            for (int nt = NSPLIT ? nt_of(tl) : 0, nt_end = NSPLIT ? nt + 1 : ntiles; nt < nt_end; ++nt) {
                // the n-tile of the item after this (unit, n-tile)'s last chunk: the weight ring runs across items
                const int nt_after = NSPLIT ? (tl + 1 < mine ? nt_of(tl + 1) : 0) : (nt + 1 < ntiles ? nt + 1 : 0);
                {
                    const int nt2 = nch > 1 ? nt : nt_after, ch2 = nch > 1 ? 1 : 0;
                    if (a.dbg != 3) compute(&smem[k & 1][0] + foff, nt, 0, nt2, ch2, std::true_type{});
                    else {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                    }
                    ++k;
                    WSP(0); WSP_COUNT();
                    stress_delay(a.stress, 101u + 8u * (unsigned)k);
                    __syncthreads();                         // S1: planes consumed; the memory waves are done with Tile
                    WSP(1);
                }
                for (int ch = 1; ch < nch; ++ch) {
                    const int nt2 = ch + 1 < nch ? nt : nt_after, ch2 = ch + 1 < nch ? ch + 1 : 0;
                    if (a.dbg != 3) compute(&smem[k & 1][0] + foff, nt, ch, nt2, ch2, std::false_type{});
                    ++k;
                    WSP(0); WSP_COUNT();
                    stress_delay(a.stress, 102u + 8u * (unsigned)k);
                    __syncthreads();                         // S1
                    WSP(1);
                }
                // the accumulators' 4 row registers are 4 consecutive pixels of one channel: one ds_write_b128 per (pixel
                // block, column block) into the channel-major tile (the pixel-major tile of the first version cost 64
                // 4-byte LDS writes + 32 + 32 read-modify-writes per lane: 1.75 us per tile)
                float* const trow = Tile + (nh * 32 + (lane & 15)) * V2LDT + (lane >> 4) * 4;
                if (kg == 1) {
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            float4 v;
                            v.x = accM[mb][nb][0] + accL[mb][nb][0] * H3_INV; v.y = accM[mb][nb][1] + accL[mb][nb][1] * H3_INV;
                            v.z = accM[mb][nb][2] + accL[mb][nb][2] * H3_INV; v.w = accM[mb][nb][3] + accL[mb][nb][3] * H3_INV;
                            *reinterpret_cast<float4*>(trow + nb * 16 * V2LDT + mb * 16) = v;
                        }
                }
                stress_delay(a.stress, 103u + 8u * (unsigned)k);
                __syncthreads();                             // S2
                if (kg == 0) {
                    float bias[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int gn = nt * T2N + nh * 32 + (lane & 15) + nb * 16;
                        bias[nb] = (a.bias && gn < a.N) ? a.bias[gn] : 0.f;
                    }
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            float4* tp = reinterpret_cast<float4*>(trow + nb * 16 * V2LDT + mb * 16);
                            float4 v = *tp;
                            v.x = (v.x + (accM[mb][nb][0] + accL[mb][nb][0] * H3_INV)) + bias[nb];
                            v.y = (v.y + (accM[mb][nb][1] + accL[mb][nb][1] * H3_INV)) + bias[nb];
                            v.z = (v.z + (accM[mb][nb][2] + accL[mb][nb][2] * H3_INV)) + bias[nb];
                            v.w = (v.w + (accM[mb][nb][3] + accL[mb][nb][3] * H3_INV)) + bias[nb];
                            *tp = v;
                        }
                }
                WSP(2);
                stress_delay(a.stress, 104u + 8u * (unsigned)k);
                __syncthreads();                             // S3: the finished tile is in LDS
                WSP(3);
            }
        WSP_FLUSH(0);
        return;
    }

    // ============================================== memory waves ===================================================
    // kernel arguments the loop needs, as locals (the argument block is 600+ bytes: left alone, the compiler re-reads
    // fields through s_load inside the loop); element offsets are 32-bit (host: images * pixels * channels < 2^31)
    if (a.dbg == 12 || (a.dbg != 10 && a.dbg != 11 && MODE == SRC2_GN_SS_SILU)) __builtin_amdgcn_s_setprio(3);       // GroupNorm + SiLU on load: the staging VALU work is the long pole (dbg 10 / 11 / 12: priority experiments)
    const int lt = tid - 256;
    const int c4 = lt & 15, r0 = lt >> 4;                    // staging: float4 c4 of window pixels r0 + 16 p
    const int tpi = a.tpi, tiles_x = a.tiles_x, Hout = a.Hout, Wout = a.Wout, Win = a.Win, ldo = a.ldo, N = a.N;
    const int nsrc = a.nsrc, C0 = a.src[0].C, C1 = a.src[1].C, ld0 = a.src[0].ld, ld1 = a.src[1].ld;
    const float* const sp0 = a.src[0].p;
    const float* const sp1 = a.src[1].p;
    float* const outp = a.out;
    float* const stats_out = a.stats_out;
    const int so_gw = a.so_gw, dbg = a.dbg, stress = a.stress;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const int gw_shift = 31 - __builtin_clz(a.src[0].gw | 1);
    const float* const gn_stats = a.src[0].stats;
    const int gn_P = a.src[0].P;
    const float gn_cnt = a.src[0].cnt;
    // Per thread and window pixel p, once: the source byte offset relative to the tile's origin pixel, per unit of row
    // pitch (dpix4 = 4 * pixel delta), and which pixels fall off each image border.  Per item the address is then ONE mad
    // per pixel and the validity mask a handful of scalar selects (the first version recomputed y, x, the bounds test, the
    // pixel index and a 64-bit address per pixel and item: ~150 of the memory waves' ~650 VALU instructions per item;
    // measured: the step time did not change, so it is not the instruction COUNT of a lone wave per SIMD that bounds
    // the memory waves).  Loads are raw-buffer loads: an offset
    // that falls outside the tensor (the row above the first image) reads zeros instead of faulting; in-range but
    // off-image pixels read a neighbour and are zeroed when staged.
    int dpix4[NP];
    unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_valid = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = r0 + 16 * p;
        const int hy = r / SW, hx0 = r - hy * SW;
        const bool second = PAIR && hx0 >= 10;               // the pair's right image
        const int hx = second ? hx0 - 10 : hx0;
        dpix4[p] = 4 * ((KIND == CONV_UP2) ? ((hy - 1) >> 1) * Win + ((hx - 1) >> 1) : (hy - 1) * Win + (hx - 1) + (second ? HWi : 0));
        m_valid |= (r < R) ? (1u << p) : 0u;
        m_top |= (hy == 0) ? (1u << p) : 0u;  m_bot |= (hy == V2Y + 1) ? (1u << p) : 0u;
        m_left |= (hx == 0) ? (1u << p) : 0u; m_right |= (hx == (PAIR ? 9 : V2X + 1)) ? (1u << p) : 0u;
    }
    const unsigned src_bytes0 = (unsigned)a.NI * (PAIR ? 2u : 1u) * (unsigned)HWi * (unsigned)ld0 * 4u;
    const unsigned src_bytes1 = (unsigned)a.NI * (PAIR ? 2u : 1u) * (unsigned)HWi * (unsigned)ld1 * 4u;
    // MODE SRC2_SCALED (ForceUnet's input-gradient pass): the source is a gradient tensor whose magnitudes sit far below
    // fp16's normal range; src[0].stats points at one word PER IMAGE, the bit pattern of max |source| over that image
    // (atomicMax by the producer).  An image's rows are staged times the power of two that puts ITS maximum in [2^13, 2^14)
    // and its tile is written times the inverse: both exact -- and a function of that image alone, so its gradient does
    // not depend on the rest of the batch (PAIR kind: the two images of a tile share the larger of their two maxima).
    // a.res (may alias a.out: every element is read by the thread that writes it) is added.
    // The maxima are LOADED with the item's window (load_item) / the tile's addend (load_acc) and only turned into scales
    // where the loaded data is consumed anyway (store_item / write_tile): a wait on them at the point of the load would
    // drain the window loads issued just before (vmcnt retires in order) and stall the two-deep pipeline -- measured:
    // 47 -> 61 ms per design-gradient call with the scale computed inside load_item.
    const unsigned* const amax = reinterpret_cast<const unsigned*>(a.src[0].stats);
    auto scale_exp = [&](unsigned m0, unsigned m1) -> int {     // biased exponent of the staging scale from the image's (pair's) maximum
        const unsigned mb = PAIR ? max(m0, m1) : m0;
        const int e = (int)(mb >> 23);                          // biased exponent of the maximum (the sign bit is clear)
        return e == 0 ? 127 : min(max(267 - e, 1), 253);
    };
    // The finished tile's scale is NOT loaded again when the tile is written: write_tile sits behind the next item's window
    // loads, and a wait on any load issued there drains them too (measured: every input-gradient convolution 2x slower,
    // 47 -> 70 ms per design-gradient call).  Instead the exponent of every staged item is kept for four items (an item
    // is staged one phase before it is multiplied and its tile is written one phase after).
    unsigned mx_in[2] = {0u, 0u};                               // raw maxima of the item loaded last
    int se_h0 = 127, se_h1 = 127, se_h2 = 127, se_h3 = 127;     // four slots: items k - 1 (tile being written) .. k + 2 (just decoded) are live at once
    auto hist_set = [&](int j, int v) { const int r = j & 3; if (r == 0) se_h0 = v; else if (r == 1) se_h1 = v; else if (r == 2) se_h2 = v; else se_h3 = v; };
    auto hist_get = [&](int j) -> int { const int r = j & 3; return r == 0 ? se_h0 : (r == 1 ? se_h1 : (r == 2 ? se_h2 : se_h3)); };
    const float* const resp = a.res;
    const int ldres = a.ldres;
    float4 racc[8];
    float4 areg[NP];
    float4 pg = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), psc = pb, psh = pb;
    constexpr int MAXPV = WS_MAXP / 8;
    float2 pv[MAXPV];                                        // GroupNorm partials lane & 7, + 8, ... of group lane >> 3
    float4 fa = pg, fb = pb;                                 // GroupNorm + scale/shift folded per channel: y = v * fa + fb
    int gsel = 0;
    unsigned okmask = 0;
    bool cok = true;
    // issue the global loads of item (mt, ch): its window and, GroupNorm mode, the image's statistics partials
    auto load_item = [&](int mt, int ch) {
        const int img = tpi_sh >= 0 ? mt >> tpi_sh : mt / tpi, ti = mt - img * tpi;
        const int tyi = tx_sh >= 0 ? ti >> tx_sh : ti / tiles_x;
        const int ty0 = tyi * V2Y, tx0 = (ti - tyi * tiles_x) * V2X;
        const bool first = (ch < nch0) || (nsrc == 1);
        const int cl = (first ? ch : ch - nch0) * KC + c4 * 4;
        const int Cc = first ? C0 : C1;
        const int ld = first ? ld0 : ld1;
        const float* base = first ? sp0 : sp1;
        const int clc = min(cl, Cc - 4);
        cok = cl < Cc;
        okmask = m_valid & ~((ty0 == 0 ? m_top : 0u) | (ty0 + V2Y == Hout ? m_bot : 0u) | (PAIR || tx0 == 0 ? m_left : 0u) |
                             (PAIR || tx0 + V2X == Wout ? m_right : 0u));
        if constexpr (MODE == SRC2_SCALED) { mx_in[0] = amax[PAIR ? 2 * img : img]; if constexpr (PAIR) mx_in[1] = amax[2 * img + 1]; }
        const int origin = (KIND == CONV_UP2) ? img * HWi + (ty0 >> 1) * Win + (tx0 >> 1) : (PAIR ? 2 * img : img) * HWi + ty0 * Win + tx0;
        const int t4 = (origin * ld + clc) * 4;               // byte offset of the tile origin's float4 of this thread
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, first ? src_bytes0 : src_bytes1, 0x00020000);
        // (Round 5 tried to mask here instead of in store_item -- a window pixel outside the image requested at an offset beyond the
        // buffer, which a raw-buffer load answers with zeros: one select per pixel instead of four.  Same-box against the round's first
        // commit the plain-source launches were 4 - 7 % SLOWER with it (profiles/r05_round_ratio.txt, first column pair): removed.)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, dpix4[p] * ld + t4, 0, 0);
            areg[p] = __builtin_bit_cast(float4, v);
        }
        if constexpr (MODE == SRC2_GN_SS_SILU) {
            const Src& s = a.src[0];
            pg = *reinterpret_cast<const float4*>(s.gamma + clc);
            pb = *reinterpret_cast<const float4*>(s.beta + clc);
            if (s.tb) {
                psc = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + clc);
                psh = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + s.C + clc);
            }
            gsel = clc >> gw_shift;
            const float2* pp = reinterpret_cast<const float2*>(gn_stats) + ((size_t)img * 8 + (lane >> 3)) * gn_P + (lane & 7);
#pragma unroll
            for (int i = 0; i < MAXPV; ++i) pv[i] = (lane & 7) + 8 * i < gn_P ? pp[8 * i] : make_float2(0.f, 0.f);
        }
    };
    // GroupNorm (mean, rstd) of this thread's channels from the partials in pv: every memory wave merges for itself,
    // without LDS-pipeline shuffles (DPP row sums, v_readlane of the 8 group totals, a select chain; merge_stats' formula)
    auto finish_stats = [&]() {
        if constexpr (MODE == SRC2_GN_SS_SILU) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MAXPV; ++i) s += pv[i].x;          // absent partials were loaded as zeros
            s = seg_total(s, 8);                                  // group totals at lanes 7 (mod 8)
            const float invP = 1.0f / (float)gn_P;
            // dbg == 77: the shuffle (ds_bpermute) form of the three group totals, the variant that staged stale window pixels while
            // this kernel was written (round 2).  Same values bit for bit when the hand-overs are right: tools/ws_shfl_experiment.py
            // counts non-repeating forwards with and without the stress mode.
            const bool shf = dbg == 77;
            const float m = (shf ? __shfl(s, (lane & ~7) | 7) : group8_total(s, lane >> 3)) * invP;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < MAXPV; ++i) {
                const float d = pv[i].x - m;
                q += ((lane & 7) + 8 * i < gn_P) ? pv[i].y + gn_cnt * d * d : 0.f;
            }
            q = seg_total(q, 8);
            const float gm = (shf ? __shfl(s, 8 * gsel + 7) : group8_total(s, gsel)) * invP;
            const float gr = 1.0f / sqrtf((shf ? __shfl(q, 8 * gsel + 7) : group8_total(q, gsel)) / (gn_cnt * (float)gn_P) + 1e-5f);
            // ((v - gm) gr g + b)(sc + 1) + sh  =  v fa + fb   (one fma per element in the staging loop)
            const float sx = psc.x + 1.0f, sy = psc.y + 1.0f, sz = psc.z + 1.0f, sw = psc.w + 1.0f;
            fa.x = gr * pg.x * sx; fa.y = gr * pg.y * sy; fa.z = gr * pg.z * sz; fa.w = gr * pg.w * sw;
            fb.x = (pb.x - gm * gr * pg.x) * sx + psh.x; fb.y = (pb.y - gm * gr * pg.y) * sy + psh.y;
            fb.z = (pb.z - gm * gr * pg.z) * sz + psh.z; fb.w = (pb.w - gm * gr * pg.w) * sw + psh.w;
        }
    };
    auto store_item = [&](int buf, int item) {
        unsigned char* S0 = &smem[buf][0];
        unsigned char* S1 = S0 + PLANE;
        float in_s = 1.0f;
        if constexpr (MODE == SRC2_SCALED) {
            const int se = scale_exp(mx_in[0], mx_in[1]);
            hist_set(item, se);
            in_s = __builtin_bit_cast(float, (unsigned)se << 23);
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + 16 * p;
            float4 v = areg[p];
            if constexpr (MODE == SRC2_GN_SS_SILU) {
                v.x = silu_f(__builtin_fmaf(v.x, fa.x, fb.x));
                v.y = silu_f(__builtin_fmaf(v.y, fa.y, fb.y));
                v.z = silu_f(__builtin_fmaf(v.z, fa.z, fb.z));
                v.w = silu_f(__builtin_fmaf(v.w, fa.w, fb.w));
            }
            if constexpr (MODE == SRC2_SCALED) { v.x *= in_s; v.y *= in_s; v.z *= in_s; v.w *= in_s; }
            const bool ok = ((okmask >> p) & 1u) && cok;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            if (r < R) {
                *reinterpret_cast<half4v*>(S0 + r * V2PITCH + c4 * 8) = hi;
                *reinterpret_cast<half4v*>(S1 + r * V2PITCH + c4 * 8) = lo;
            }
        }
    };
    // pixel offset of tile row jj (of this wave's two), tile column 4 q + pi, from the wave's first pixel; PAIR: tile columns
    // 8 .. 15 are columns 0 .. 7 of the next image
    auto opix = [&](int jj, int q, int pi) {
        return PAIR ? jj * Wout + ((4 * q + pi) & 7) + (q >= 2 ? Hout * Wout : 0) : jj * Wout + 4 * q + pi;
    };
    // MODE SRC2_SCALED: the addend of tile (mt, nt), issued BEFORE the next item's loads so that the wait for it does not
    // cover them (write_tile's addressing)
    auto load_acc = [&](int mt, int nt) {
        if constexpr (MODE == SRC2_SCALED) {
            if (!resp) return;
            const int img = tpi_sh >= 0 ? mt >> tpi_sh : mt / tpi, ti = mt - img * tpi;
            const int tyi = tx_sh >= 0 ? ti >> tx_sh : ti / tiles_x;
            const int ty0 = tyi * V2Y, tx0 = (ti - tyi * tiles_x) * V2X;
            const int oc4 = lane & 15, q = lane >> 4;
            const int col = nt * T2N + oc4 * 4;
            const float* r0p = resp + (((PAIR ? 2 * img : img) * Hout + ty0 + 2 * lw) * Wout + tx0) * ldres + min(col, N - 4);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) racc[4 * jj + pi] = *reinterpret_cast<const float4*>(r0p + opix(jj, q, pi) * ldres);
        }
    };
    // the finished tile (mt, nt): wave lw stores tile pixels 32 lw .. 32 lw + 31 (two pixel rows) and their GroupNorm partial
    auto write_tile = [&](int mt, int nt, int item) {
        const int img = tpi_sh >= 0 ? mt >> tpi_sh : mt / tpi, ti = mt - img * tpi;
        const int tyi = tx_sh >= 0 ? ti >> tx_sh : ti / tiles_x;
        const int ty0 = tyi * V2Y, tx0 = (ti - tyi * tiles_x) * V2X;
        const int oc4 = lane & 15, q = lane >> 4;
        const int col = nt * T2N + oc4 * 4;
        const bool nok = col < N;                            // N is a multiple of 4 (host)
        // this wave's 32 pixels = 8 blocks of 4 (block b: tile row 2 lw + (b >> 2), columns 4 (b & 3) ..); lane = (channel
        // quad oc4, blocks q and q + 4): four 16-byte LDS reads give 4 channels x 4 pixels, whose transpose is a renaming
        const size_t o_off = (size_t)(((PAIR ? 2 * img : img) * Hout + ty0 + 2 * lw) * Wout + tx0) * ldo + col;
        const float* t0 = Tile + (oc4 * 4) * V2LDT + 32 * lw + 4 * q;
        float4 f[2][4];                                      // all eight LDS reads in flight before the first store
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) f[jj][ci] = *reinterpret_cast<const float4*>(t0 + ci * V2LDT + 16 * jj);
        float4 v[8];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            v[4 * jj + 0] = make_float4(f[jj][0].x, f[jj][1].x, f[jj][2].x, f[jj][3].x);
            v[4 * jj + 1] = make_float4(f[jj][0].y, f[jj][1].y, f[jj][2].y, f[jj][3].y);
            v[4 * jj + 2] = make_float4(f[jj][0].z, f[jj][1].z, f[jj][2].z, f[jj][3].z);
            v[4 * jj + 3] = make_float4(f[jj][0].w, f[jj][1].w, f[jj][2].w, f[jj][3].w);
        }
        if constexpr (MODE == SRC2_SCALED) {
            const float out_s = __builtin_bit_cast(float, (unsigned)(254 - hist_get(item)) << 23);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j].x *= out_s; v[j].y *= out_s; v[j].z *= out_s; v[j].w *= out_s; }
            if (resp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j].x += racc[j].x; v[j].y += racc[j].y; v[j].z += racc[j].z; v[j].w += racc[j].w; }
            }
        }
        if (nok && dbg != 4) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi)               // block b = q + 4 jj: row jj, column 4 q + pi
                    *reinterpret_cast<float4*>(outp + o_off + (size_t)opix(jj, q, pi) * ldo) = v[4 * jj + pi];
        }
        if (!stats_out) return;
        // shifted sums about the group's first element of the wave's first pixel
        const int gwt = so_gw;                               // 8 or 16 channels: 2 or 4 lanes (oc4) per group
        const float K = Tile[((oc4 * 4) & ~(gwt - 1)) * V2LDT + 32 * lw];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d0 = v[j].x - K, d1 = v[j].y - K, d2 = v[j].z - K, d3 = v[j].w - K;
            s1 += (d0 + d1) + (d2 + d3);
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        s1 = dpp_add<0x111, 0xf>(s1); s2 = dpp_add<0x111, 0xf>(s2);                     // pairs of oc4 -> odd lanes
        if (gwt >= 16) { s1 = dpp_add<0x112, 0xf>(s1); s2 = dpp_add<0x112, 0xf>(s2); }  // quads -> lanes 3 (mod 4)
        s1 = xsum32(xsum16(s1)); s2 = xsum32(xsum16(s2));                                // the wave's 4 pixel phases
        const int lpg = gwt >> 2;                            // lanes per group
        if (lane < 16 && (oc4 & (lpg - 1)) == lpg - 1 && nok) {
            const int g = col >> (31 - __builtin_clz(gwt));
            const float ine = 1.0f / (float)(32 * gwt);      // a power of two: the products below are exact divisions
            float* o = stats_out + (((size_t)img * 8 + g) * (tpi * WS_SPT) + ti * WS_SPT + lw) * 2;
            const float mean_d = s1 * ine;
            o[0] = K + mean_d;
            o[1] = fmaxf(s2 - s1 * mean_d, 0.f);
        }
    };

    // Software pipeline, two items deep: phase k (the matrix waves multiply item k) stages item k+1 from the registers
    // loaded during phase k-1, THEN issues item k+2's loads into the same registers, THEN writes the previous tile.
    // vmcnt retires in order: with the tile's stores issued after the loads, no wait for a load ever covers a store.
    // An item whose planes (pixel tile, chunk) are the ones its buffer already holds -- the second n-tile of a
    // two-chunk layer: items c0, c1, c0, c1 alternate buffers 0, 1, 0, 1 -- is neither loaded nor staged again.
    ItemCur c0, c1, c2;                                      // items k, k + 1, k + 2
    cur_first(c0);
    c1 = c0; cur_next(c1);
    c2 = c1; cur_next(c2);
    int mt = c0.mt, nt = c0.nt, ch = c0.ch, mtn, chn;
    load_item(mt, ch);
    finish_stats();
    store_item(0, 0);
    bool staged_skip = false;                                // the item waiting for its staging needs none
    if (nitems > 1) load_item(c1.mt, c1.ch);
    stress_delay(stress, 200u);
    __syncthreads();                                         // S0
    int pmt = -1, pnt = 0, pk = 0;                           // finished tile waiting in LDS (pk: its last item)
    WSP_DECL;
    for (int k = 0; k < nitems; ++k) {
        if (stress > 0) stress_delay(stress, 204u + 8u * (unsigned)k);
        if (k + 1 < nitems && dbg != 5 && !staged_skip) { finish_stats(); store_item((k + 1) & 1, k + 1); }
        WSP(0); WSP_COUNT();
        if (pmt >= 0) load_acc(pmt, pnt);
        if (k + 2 < nitems && dbg != 5) {
            mtn = c2.mt; chn = c2.ch;
            staged_skip = (mtn == mt && chn == ch);                // its buffer, (k + 2) & 1, holds item k's planes: these
            if (!staged_skip && dbg != 2) load_item(mtn, chn);
            else if constexpr (MODE == SRC2_SCALED) hist_set(k + 2, hist_get(k));      // (the same planes, the same image, the same scale)
        }
        WSP(1);
        if (stress > 0) stress_delay(stress, 201u + 8u * (unsigned)k);
        if (pmt >= 0 && dbg != 5) { write_tile(pmt, pnt, pk); pmt = -1; }
        WSP(2);
        if (stress > 0) stress_delay(stress, 202u + 8u * (unsigned)k);
        __syncthreads();                                     // S1
        WSP(3);
        if (ch == nch - 1) {
            if constexpr (KSPLIT) __syncthreads();           // S2
            if (stress > 0) stress_delay(stress, 203u + 8u * (unsigned)k);
            __syncthreads();                                 // S3
            WSP(4);
            pmt = mt; pnt = nt; pk = k;
        }
        c0 = c1; c1 = c2; cur_next(c2);
        mt = c0.mt; nt = c0.nt; ch = c0.ch;
    }
    if (pmt >= 0) { load_acc(pmt, pnt); write_tile(pmt, pnt, pk); }
    WSP_FLUSH(1);
}

}  // namespace cindm
