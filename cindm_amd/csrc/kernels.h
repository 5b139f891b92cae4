// Device kernels of libcindm_hip.so (gfx950 / CDNA4 only).
//
// Data layout: every activation is channel-last fp32 [rows = sample*L + position, C] with an
// explicit row stride, which is the reference's API layout [B, horizon, F]
// (model/diffusion_1d.py:610-614) -- the reference's two `b h t -> b t h` transposes disappear.
//
// conv_gemm_kernel<T>: implicit-GEMM 1-D convolution / linear layer on fp32 MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32 == an fmaf chain in k order).
//   out[b, lo, n] = bias[n] + sum_{tap<T} sum_{c<Cin} W[tap][c][n] * act(in)[b, li(lo,tap), c]
// One workgroup (4 waves) owns a 48-row x 32-column output tile; since C*L = 1536 for every
// tensor of the U-Net this is exactly one tile per sample-equivalent, i.e. B' workgroups per layer.
// The 4 waves split K (each wave owns 8 of every 32 input channels, all taps) and are reduced
// through LDS at the end, so nothing is shared between waves inside the K loop:
//   * A (activations) is staged per 32-channel chunk into LDS ONCE per workgroup, with the
//     producer's GroupNorm+Mish(+time bias) / LayerNorm applied on the way in ("normalise on
//     load"), and read as MFMA fragments with one ds_read_b32 per 16x4 fragment; conv taps are
//     pure LDS row offsets, out-of-range taps point at a zero row.
//   * B (weights, repacked [tap][Cin][Cout]) goes straight from L2 to VGPRs in MFMA fragment
//     order (each weight is used by exactly one wave), prefetched one chunk ahead.
// Epilogue: cross-wave reduce, bias, optional "+ Mish(GroupNorm(y))" and "+ residual" terms,
// store, and per-(sample, group) GroupNorm statistics / per-row LayerNorm statistics of the
// OUTPUT as (mean, M2) partials (two-pass in LDS, merged by the consumer with Chan's formula)
// so that no separate normalisation pass over HBM exists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cindm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Cross-row exchanges on gfx950 without the LDS crossbar: v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane
// halves between two registers, so with both operands = v the pair (a, b) holds (own half, other half) in some order
// and a commutative op(a, b) is the xor-16 / xor-32 all-reduce step (checked against __shfl_xor by
// tools/micro/permlane_test.hip).  The builtin form mis-folds identical operands, hence the asm.
__device__ __forceinline__ void permlane16_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xsum16(float v) { float a, b; permlane16_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xsum32(float v) { float a, b; permlane32_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xmax16(float v) { float a, b; permlane16_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xmax32(float v) { float a, b; permlane32_pair(v, a, b); return fmaxf(a, b); }
// sum over the 16 lanes of a DPP row, result in every lane (row_ror:8, row_ror:4, two quad permutes: VALU speed)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_get<0x128>(v);
    v += dpp_get<0x124>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0xB1>(v);
    return v;
}
// all-reduce sum over groups of LPR consecutive lanes (LPR = 16, 32 or 64)
template <int LPR>
__device__ __forceinline__ float rowgroup_sum(float v) {
    static_assert(LPR == 16 || LPR == 32 || LPR == 64, "16, 32 or 64 lanes per group");
    v = row16_sum(v);
    if (LPR >= 32) v = xsum16(v);
    if (LPR >= 64) v = xsum32(v);
    return v;
}

// x_start.clamp_(-1, 1) of the posterior (model/diffusion_1d.py:1038-1041, model/diffusion_2d.py:763): torch.clamp hands a NaN
// through, fminf / fmaxf return their OTHER operand -- a U-Net output that left fp16's range inside a chain (NaN) would become a
// silent x_start of 1 and the range rule's "never a silently wrong finite value" would not hold for chains (found in round 6 by the
// range test on a chain).
__device__ __forceinline__ float clamp_pm1(float v) { const float c = fminf(fmaxf(v, -1.f), 1.f); return v != v ? v : c; }

// The step's timestep: from device memory inside the sample loops (t_ptr, graph-replayable), an immediate otherwise.  The
// immediate goes through an opaque SGPR move so that it is a VALUE, not a kernarg LOCATION: hipcc otherwise folds
// `p ? *p : a.t_imm` into ONE load from a selected address -- a FLAT vector load (one address global, one constant) whose
// first use then waits vmcnt(0), i.e. for every weight / activation load the kernel has issued behind it (level0_down_kernel
// and the L = 6 dconv2 kernels did; round 4, found in the ISA).  With this form it is a scalar branch and an s_load.
// The load itself goes through the CONSTANT address space (nothing in the running kernel writes the word it reads: the
// ping-pong update writes the OTHER slot), which makes it an s_load whatever the surrounding code looks like.
typedef const int __attribute__((address_space(4))) cindm_cint4;
// a word that no thread of the running kernel writes before this read (exchange epoch, error flag at kernel entry): scalar load
__device__ __forceinline__ int uniform_word(const int* p) { return *(cindm_cint4*)(const void*)p; }
__device__ __forceinline__ int step_scalar(const int* p, int imm) {
    imm = __builtin_amdgcn_readfirstlane(imm);          // (provably uniform for the "s" constraint)
    asm volatile("" : "+s"(imm));
    typedef const int __attribute__((address_space(4))) cint4;
    return p ? *(cint4*)(const void*)p : imm;
}

// L2 warm-up of the NEXT launch's weights.  Measured (profiles/r02_*): a launch whose weights come from the Infinity
// Cache instead of its XCD's L2 streams them at 74 instead of 117 GB/s per CU, 65 us per reverse step in total; a side
// stream cannot do the warm-up (cross-stream graph edges cost more than they save).  So every launch touches, one dword
// per 128-byte line, the weights the launch AFTER it will stream: up to two regions, region k of XCD x = base[k] +
// x * stride[k] (tiled weights: the n-tiles an XCD will own; stride 0: weights every workgroup reads).  Block b runs on
// XCD b % 8 (observed dispatch rule; a wrong guess only loses the benefit).  The loaded dwords are parked in registers
// and "used" at the very end of the kernel, so the loads cost no wait.
// Stress mode of the kernels that hand data between waves / workgroups inside a launch (option "stress" = seed > 0): a
// pseudo-random pause, keyed by (seed, workgroup, wave, site), in front of every producer -> consumer hand-over.  A
// protocol that is correct gives bit-identical results under any such skew; one that only works at the natural timing
// does not.  Zero cost when off (one scalar compare).
__device__ __forceinline__ void stress_delay(int stress, unsigned site) {
    if (stress <= 0) return;
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6);
    unsigned h = (unsigned)stress * 0x9E3779B9u + ((blockIdx.y * gridDim.x + blockIdx.x) * 16u + wv) * 0x85EBCA6Bu + site * 0xC2B2AE35u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    const int n = (int)(h & 15u);                            // 0 .. 15 x 512 cycles: up to ~4 us at 2 GHz
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}

// In-kernel phase clocks -- profiling builds only (-DCINDM_PHASE_PROF: `python -m cindm_amd.build --prof` ->
// libcindm_hip_prof.so; the production library contains none of this).  Every wave keeps up to PH_NST stamps of the
// 100 MHz constant clock (s_memrealtime: the same time base on every CU, so records of different workgroups -- and of
// consecutive launches -- line up) in registers and lane 0 writes them at the end of the kernel:
// buf[((slot * PH_MAXWG + workgroup) * PH_MAXWAVE + wave) * PH_NST + i].  sched_barrier pins a mark between the phases it
// separates; the cost of the instrumentation is the difference between the profiling and the production build of the same
// launch (tools/phase_table.py reports both).
constexpr int PH_MAXWG = 1024, PH_MAXWAVE = 8, PH_NST = 16;
struct PhaseBuf { unsigned long long* buf; int slot; };
#ifdef CINDM_PHASE_PROF
#define PH_DECL unsigned long long ph_[cindm::PH_NST] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
#define PH(i) do { __builtin_amdgcn_sched_barrier(0); ph_[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PH_FLUSH(pb) do { \
        if ((pb).buf && (threadIdx.x & 63) == 0) { \
            const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; \
            if (wg_ < (unsigned)cindm::PH_MAXWG) { \
                unsigned long long* o_ = (pb).buf + ((((size_t)(pb).slot * cindm::PH_MAXWG + wg_) * cindm::PH_MAXWAVE + (threadIdx.x >> 6)) * cindm::PH_NST); \
                _Pragma("unroll") for (int i_ = 0; i_ < cindm::PH_NST; ++i_) o_[i_] = ph_[i_]; \
            } \
        } \
    } while (0)
#else
#define PH_DECL do { } while (0)
#define PH(i) do { } while (0)
#define PH_FLUSH(pb) do { } while (0)
#endif

// PF_REGIONS regions.  (Round 4 tried four -- a dconv2 launch streams two convolutions' weights and an XCD owns two n-tiles of
// each --: same-box A/B 329.9 us per reverse step with two regions, 332.4 with four.  Round 6, with the touches issued late and the
// outputs written through: conv A's tiles x and x + 8 are the two regions of a dconv2 successor; adding conv B's tiles (touched a
// whole phase before their use) cost 1.2 us per step again, adding the riding 1x1's changed nothing.  Two it stays.)
constexpr int PF_REGIONS = 2;
struct Pf { const char* base[PF_REGIONS]; unsigned bytes[PF_REGIONS]; unsigned stride[PF_REGIONS]; int* sink; int late; int wt; };
struct PfRegs { unsigned v[PF_REGIONS][2]; };
// (Every caller is a 256-thread kernel.  Round 4: the block size is a CONSTANT here.  `blockDim.x` is a 16-bit VECTOR-memory load
// from the dispatch packet: its use made hipcc wait `vmcnt(0)` in the middle of this function -- vector loads return in order, so
// that drained every weight / activation load the kernel had in flight at that point, a full L2 / Infinity-Cache round trip on
// the critical path of every launch of the step.  Found with the in-replay phase clocks: the "cross-wave reduce" phase that
// follows this call measured 1.7 - 2.4 us in dconv2_kernel's phase A against 0.45 us for the same code in phase B.)
__device__ __forceinline__ void l2_prefetch(const Pf& p, PfRegs& r) {
    constexpr int BLOCK = 256;
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, xcd = lin & 7, rank = lin >> 3;
    const int nshare = (int)((gridDim.x * gridDim.y * gridDim.z + 7) >> 3);
#pragma unroll
    for (int k = 0; k < PF_REGIONS; ++k) {
        const int lines = (int)(p.bytes[k] >> 7);
        const int per = (lines + nshare - 1) / nshare;
        const char* base = p.base[k] + (size_t)xcd * p.stride[k];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int li = (int)threadIdx.x + BLOCK * i;
            const int line = rank * per + li;
            r.v[k][i] = 0u;
            if (li < per && line < lines) r.v[k][i] = *reinterpret_cast<const unsigned*>(base + ((size_t)line << 7));
        }
    }
}
// A/B of the touches' placement (round 6): at the head of the launch (rounds 2 - 5), or a few microseconds before its end -- a line
// touched 10 - 25 us early has to survive the launch's own stream through a 4 MB L2
__device__ __forceinline__ void l2_prefetch_early(const Pf& p, PfRegs& r) {
#pragma unroll
    for (int k = 0; k < PF_REGIONS; ++k) r.v[k][0] = r.v[k][1] = 0u;
    if (!p.late) l2_prefetch(p, r);
}
// The late point of a kernel = right behind its LAST load request where there is one (vector loads return in order: a request that
// is issued behind the touches waits for them, ~1 - 1.5 us from the Infinity Cache -- placed in front of a LayerNorm gain load they cost
// level0_down 1.6 us), else behind its last K loop.
__device__ __forceinline__ void l2_prefetch_late(const Pf& p, PfRegs& r) { if (p.late) l2_prefetch(p, r); }
__device__ __forceinline__ void l2_prefetch_done(const Pf& p, const PfRegs& r) {
    unsigned x = 0u;
#pragma unroll
    for (int k = 0; k < PF_REGIONS; ++k) x ^= r.v[k][0] ^ r.v[k][1];
    if (p.sink && x == 0x9e3779b9u) p.sink[0] = 1;    // keeps the loads alive
}

// A launch's OUTPUT stores, written through the L2 (sc1) when `wt` (round 6).  With plain stores a launch leaves its outputs dirty in
// the XCD's L2 and the end-of-kernel release writes them back before the dependent launch may start -- part of every gap of the
// replayed step ("boundary ... + B / 6 TB/s when the predecessor leaves B bytes dirty", MI355X_MICROARCH.md) -- and the consumers, on
// other XCDs, read them from memory anyway.  `base` must be wave-uniform (a kernel argument plus a block-uniform offset): it becomes a
// buffer resource; `off` is the lane's element offset.
__device__ __forceinline__ void st_out(float* base, size_t off, float v, int wt) {
    if (wt) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, 0x7ffffff0u, 0x00020000),
                                                  (unsigned)(off * 4), 0, 16);      // aux 16 = sc1
    else base[off] = v;
}
__device__ __forceinline__ void st_out4(float* base, size_t off, const float4& v, int wt) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, 0x7ffffff0u, 0x00020000),
                                                   (unsigned)(off * 4), 0, 16);
    else *reinterpret_cast<float4*>(base + off) = v;
}
__device__ __forceinline__ void st_out4(uint4* base, size_t off, const uint4& v, int wt) {
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, 0x7ffffff0u, 0x00020000),
                                                   (unsigned)(off * 16), 0, 16);
    else base[off] = v;
}

// 4 x 4 transpose inside every lane quad: lane j of a quad enters with a[i] = M[j][i] and leaves with a[i] = M[i][j] (two butterfly
// stages of DPP quad permutes; 4 moves + 8 selects).  Used to turn "lane = column, registers = rows" epilogues into 16-byte row stores.
__device__ __forceinline__ void quad_transpose4(float (&a)[4], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int m = 0; m < 2; ++m) {             // lanes (0,1) and (2,3): swap M[even][2m + 1] <-> M[odd][2m]
        const float send = o1 ? a[2 * m] : a[2 * m + 1];
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
        if (o1) a[2 * m] = got; else a[2 * m + 1] = got;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {             // lanes (0,2) and (1,3): swap M[low][m + 2] <-> M[high][m]
        const float send = o2 ? a[m] : a[m + 2];
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
        if (o2) a[m] = got; else a[m + 2] = got;
    }
}

constexpr int TM = 48;        // tile rows
constexpr int TN = 32;        // tile cols
constexpr int MAXR = 96;      // max input rows per tile (stride-2 conv: 2 * 48)
constexpr int LDR = 33;       // reduce / output tile pitch

enum SrcMode { SRC_PLAIN = 0, SRC_GN_MISH = 1, SRC_LN = 2, SRC_MISH = 3, SRC_GELU = 5, SRC_SILU = 6 };

struct Src {
    const float* p;       // [rows, ld]
    int ld;
    int C;                // valid channels of this source (multiple of 4)
    int mode;             // SrcMode
    const float* stats;   // GN: [Bp][8][P][2] (mean, M2);  LN: [rows][P][2]
    int P;                // partials per statistic
    int gw;               // GN group width (channels per group)
    float cnt;            // elements per partial (GN: L*min(gw,32); LN: 32)
    const float* gamma;   // [C] (GN weight / LN g)
    const float* beta;    // [C] (GN bias) or null
    const float* tb;      // per-timestep bias table base (already offset to this layer) or null
    int tb_ld;            // table row pitch
};

struct GemmArgs {
    Src src[2];
    int nsrc;
    const float* W;       // [T][CinP][Npad]
    const float* bias;    // [Npad] or null
    int CinP, Npad, N;
    int KC;               // channels per pipeline stage the weights were packed for (host-side dispatch)
    int h3;               // weights packed as split fp16 (hi, scaled lo) for conv_gemm_h3_kernel
    int dbg;              // timing ablations of the h3 kernel (CINDM_DBG; results are wrong when set)
    int Bp, Lin, Lout, stride, pad, transposed, spt;
    int lout_magic, lin_magic;   // ceil(65536 / L): floor(r / L) == (r * magic) >> 16 for r < 256 (host: Emitter::base)
    float* out; int ldo;
    const float* res; int ldres;                         // + res
    const float* e_y; int e_ld; const float* e_stats;    // + Mish(GN(e_y))
    int e_P; int e_gw; float e_cnt; const float* e_gamma; const float* e_beta;
    float* stats_out; int so_gw;                         // GN (mean,M2) partials of the output
    float* ln_out;                                       // LN (mean,M2) partials per row & 32-col tile
    const int* t_ptr; int t_imm;                         // timestep (device pointer wins)
    // producer-side activation (GroupNorm groups that lie inside one 32-column tile, i.e. gw <= 32): the epilogue
    // normalises its own output tile and stores out = Mish(GN(v) * gamma + beta) [+ tb_t] [+ res] instead of v, so the
    // consumer stages a plain tensor (the Mish is evaluated once per element, not once per consumer n-tile)
    const float* act_gamma; const float* act_beta; const float* act_tb; int act_tb_ld;
    // second GEMM riding on the centre tap of a k=5 convolution (the block's 1x1 residual_conv reads the same staged
    // rows): out2 = W2 . x + bias2; W2 packed as split fp16 [n-tile][stage][q = nb*2 + plane][thread][8 halfs]
    const float* W2; const float* bias2; float* out2; int ldo2;
    Pf pf;                                               // (registered by the host; these kernels issue no warm-up: measured +1 us per launch)
};

__device__ __forceinline__ float mish_f(float x) {
    // Mish(x) = x * tanh(softplus(x)), PyTorch softplus threshold 20.
    // tanh(log(1+e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2),  n = e*(e+2)   (no cancellation)
    // branch-free (the select keeps the hot loop free of divergent control flow) and cheap: one v_exp_f32
    // and one v_rcp_f32 (each <= 1 ulp); |x| <= 20 keeps the exp2 argument's rounding below 4e-7 relative
    const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.4426950408889634f);
    const float n = e * (e + 2.0f);
    const float r = x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
    return x > 20.0f ? x : r;
}

// Segmented lane reduction on the DPP network (no LDS round trips): after the call the LAST lane of every aligned
// seg-lane segment (seg = 4, 8, 16 or 32) holds the segment's sum, accumulated in a fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float seg_total(float v, int seg) {
    v = dpp_add<0x111, 0xf>(v);                    // row_shr:1
    v = dpp_add<0x112, 0xf>(v);                    // row_shr:2   -> 4-lane sums at lanes 3 (mod 4)
    if (seg >= 8) v = dpp_add<0x114, 0xf>(v);      // row_shr:4   -> 8-lane sums at lanes 7 (mod 8)
    if (seg >= 16) v = dpp_add<0x118, 0xf>(v);     // row_shr:8   -> 16-lane sums at lane 15 of each row
    if (seg >= 32) v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3 -> 32-lane sums at lanes 31, 63
    return v;
}

// Merge P equal-count (mean, M2) partials -> (mean, rstd) with eps.
__device__ __forceinline__ void merge_stats(const float* st, int P, float cnt, float eps, float& mean, float& rstd) {
    float m = 0.f;
    for (int p = 0; p < P; ++p) m += st[2 * p];
    m /= (float)P;
    float M2 = 0.f;
    for (int p = 0; p < P; ++p) {
        float d = st[2 * p] - m;
        M2 += st[2 * p + 1] + cnt * d * d;
    }
    mean = m;
    rstd = 1.0f / sqrtf(M2 / (cnt * (float)P) + eps);
}

// Shared epilogue of the GEMM kernels: cross-wave K reduction through LDS, bias, optional "+ Mish(GroupNorm(y))"
// and "+ residual" terms, store, and (mean, M2) statistics of the output tile (GroupNorm / LayerNorm partials).
template <bool HAS_ACC>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[3][2], float (*Red)[TM * LDR], const float* tabE,
                                              bool skip_stats) {
    constexpr int T = HAS_ACC ? 1 : 0;
    constexpr int DBG = 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int b0 = mt * a.spt;
    const int ns = min(a.spt, a.Bp - b0);
    const int rows_out = ns * a.Lout;
    const int n0 = nt * TN;
    const int n = tid & 31, rq = tid >> 5;
    const int gn = n0 + n;
    const bool nok = gn < a.N;
    // every global read of the epilogue is issued up front (the step index and the time-bias row it addresses are a
    // dependent pair: left at their point of use they cost two exposed memory latencies per launch)
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const float bias = (a.bias && nok) ? a.bias[gn] : 0.f;
    float eg = 1.f, eb = 0.f;
    if (a.e_y && nok) { eg = a.e_gamma[gn]; eb = a.e_beta[gn]; }
    float ag = 1.f, ab = 0.f, atb = 0.f;
    if (a.act_gamma && nok) {
        ag = a.act_gamma[gn]; ab = a.act_beta[gn];
        if (a.act_tb) atb = a.act_tb[(size_t)t_now * a.act_tb_ld + gn];
    }
    float ey[6], rs[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int r = rq + 8 * q;
        ey[q] = 0.f; rs[q] = 0.f;
        if (r < rows_out && nok) {
            const size_t grow = (size_t)b0 * a.Lout + r;
            if (a.e_y) ey[q] = a.e_y[grow * a.e_ld + gn];
            if (a.res) rs[q] = a.res[grow * a.ldres + gn];
        }
    }
    // ---- cross-wave K reduction through LDS --------------------------------------------------
    // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
    if constexpr (T > 0) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    Red[w][(mb * 16 + (lane >> 4) * 4 + rg) * LDR + nb * 16 + (lane & 15)] = acc[mb][nb][rg];
        __syncthreads();
    }

#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int r = rq + 8 * q;
        float v = 0.f;
        if (T > 0) v = (Red[0][r * LDR + n] + Red[1][r * LDR + n]) + (Red[2][r * LDR + n] + Red[3][r * LDR + n]);
        v += bias;
        if (r < rows_out && nok) {
            const size_t grow = (size_t)b0 * a.Lout + r;
            if (a.e_y) {
                const int s = (r * a.lout_magic) >> 16;
                const int ti = (s * 8 + (gn >> (31 - __builtin_clz(a.e_gw)))) * 2;
                v += mish_f((ey[q] - tabE[ti]) * tabE[ti + 1] * eg + eb);
            }
            if (!a.act_gamma) {
                if (a.res) v += rs[q];
                a.out[grow * a.ldo + gn] = v;
            }
        } else {
            v = 0.f;
        }
        Red[0][r * LDR + n] = v;     // finished tile kept for the statistics passes (own slot only)
    }

    if (a.act_gamma) {
        // ---- producer-side GroupNorm + Mish on the tile (whole (sample, group) sets are inside it) ----
        __syncthreads();
        const int gwt = a.so_gw;                       // <= 32 (host guarantees)
        const int gpt = TN / gwt;                      // groups per tile
        float* tabS = &Red[1][0];                      // [sample in tile][group in tile] (mean, rstd)
        const float ne = (float)(a.Lout * gwt), inv_ne = 1.0f / ne;
        for (int sidx = rq; sidx < a.spt; sidx += 8) {
            const float* col = &Red[0][sidx * a.Lout * LDR + n];
            const float K = Red[0][sidx * a.Lout * LDR + (n & ~(gwt - 1))];
            float s1 = 0.f, s2 = 0.f;
            for (int l = 0; l < a.Lout; ++l) { const float d = col[l * LDR] - K; s1 += d; s2 += d * d; }
            s1 = seg_total(s1, gwt);
            s2 = seg_total(s2, gwt);
            if ((n & (gwt - 1)) == gwt - 1) {
                const float mean = K + s1 * inv_ne;
                const float M2 = fmaxf(s2 - s1 * s1 * inv_ne, 0.f);
                float* o = tabS + (sidx * gpt + (n >> (31 - __builtin_clz(gwt)))) * 2;
                o[0] = mean; o[1] = 1.0f / sqrtf(M2 / ne + 1e-5f);
            }
        }
        __syncthreads();
        const int gi = n >> (31 - __builtin_clz(gwt));
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = rq + 8 * q;
            float y = 0.f;
            if (r < rows_out && nok) {
                const int sl = (r * a.lout_magic) >> 16;
                const float m = tabS[(sl * gpt + gi) * 2], rs2 = tabS[(sl * gpt + gi) * 2 + 1];
                y = mish_f((Red[0][r * LDR + n] - m) * rs2 * ag + ab) + atb;
                if (a.res) y += rs[q];
                a.out[((size_t)b0 * a.Lout + r) * a.ldo + gn] = y;
            }
            Red[0][r * LDR + n] = y;
        }
        if (a.ln_out) __syncthreads();
    } else {
        if (skip_stats) return;                     // DBG 8: no statistics passes
        if (a.stats_out || a.ln_out) __syncthreads();
    }

    if (a.stats_out && !a.act_gamma) {
        // GroupNorm partial statistics of the output tile: (mean, M2) per (sample, group or 32-column part of it).
        // Thread (n = tid & 31, rq = tid >> 5) accumulates column n over the rows of samples rq, rq + 8, ... as sums
        // of (x - K) and (x - K)^2 around a pivot K taken from the data (one pass, no cancellation); the gwt columns
        // of a group are adjacent lanes and are combined on the DPP network in a fixed order.
        const int gwt = min(a.so_gw, TN);             // group columns inside this tile (power of two)
        const int P = max(1, a.so_gw / TN);           // partials per statistic
        const float ne = (float)(a.Lout * gwt), inv_ne = 1.0f / ne;
        for (int sidx = rq; sidx < a.spt; sidx += 8) {
            const float* col = &Red[0][sidx * a.Lout * LDR + n];
            const float K = Red[0][sidx * a.Lout * LDR + (n & ~(gwt - 1))];
            float s1 = 0.f, s2 = 0.f;
            for (int l = 0; l < a.Lout; ++l) { const float d = col[l * LDR] - K; s1 += d; s2 += d * d; }
            s1 = seg_total(s1, gwt);
            s2 = seg_total(s2, gwt);
            if ((n & (gwt - 1)) == gwt - 1 && sidx < ns) {
                const int g = (n0 + n) >> (31 - __builtin_clz(a.so_gw));
                const int p = (n0 / TN) & (P - 1);
                float* o = a.stats_out + (((size_t)(b0 + sidx) * 8 + g) * P + p) * 2;
                o[0] = K + s1 * inv_ne;
                o[1] = fmaxf(s2 - s1 * s1 * inv_ne, 0.f);
            }
        }
    }
    if (a.ln_out) {
        // LayerNorm partial statistics per output row over this tile's 32 columns (4 threads x 8 columns per row).
        const int r = tid >> 2, sub = tid & 3;
        float s1 = 0.f, s2 = 0.f, K = 0.f;
        if (r < TM) {
            K = Red[0][r * LDR];
            for (int c = sub * 8; c < sub * 8 + 8; ++c) { const float d = Red[0][r * LDR + c] - K; s1 += d; s2 += d * d; }
        }
        s1 = seg_total(s1, 4);
        s2 = seg_total(s2, 4);
        if (sub == 3 && r < rows_out) {
            float* o = a.ln_out + (((size_t)b0 * a.Lout + r) * (a.Npad / TN) + nt) * 2;
            o[0] = K + s1 * (1.0f / 32.0f);
            o[1] = fmaxf(s2 - s1 * s1 * (1.0f / 32.0f), 0.f);
        }
    }
}

// Pipeline: stage = KC input channels x all T taps.  While stage ch is multiplied, the B fragments and the
// A rows of stage ch+1 are in flight (registers); A is transformed and written to the other LDS buffer
// after the multiply; one barrier per stage.  The loop body is BRANCH-FREE (clamped addresses + selects,
// the tail re-loads the last stage) so that hipcc keeps counted vmcnt waits instead of draining at every
// join.  KC is 32 for the tap-ful convolutions and 64/128 for 1x1 layers (same bytes in flight per stage).
template <int T, int KC, int ROWS, int MODE, int DBG = 0>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const GemmArgs a) {
    constexpr int TT = T > 0 ? T : 1;
    constexpr int LDAK = KC + 2;            // LDS row pitch: 2*row + k spreads 16 rows x 2 k over 32 banks
    constexpr int CPW = KC / 4;             // channels per wave per stage
    constexpr int CS = CPW / 4;             // k-steps per tap per stage
    constexpr int KS = TT * CS;             // k-steps per stage per wave
    constexpr int F4R = KC / 4;             // float4 per staged row
    constexpr int RPP = 256 / F4R;          // rows per staging pass
    constexpr int NP = (ROWS + RPP - 1) / RPP;
    constexpr int LROWS = NP * RPP;         // staged rows (>= ROWS, so every staging store is unconditional)
    constexpr int ZR = LROWS;               // index of the all-zero LDS row
    __shared__ __attribute__((aligned(16))) float As[2][(LROWS + 1) * LDAK];
    __shared__ __attribute__((aligned(16))) float Red[4][TM * LDR];
    __shared__ float tabA[TM * 8 * 2];      // GN prologue: [spt][8](mean, rstd), spt <= 48;  LN: [rows_in <= 96](mean, rstd)
    __shared__ float tabE[TM * 8 * 2];      // epilogue GN table [spt][8](mean, rstd)

    if constexpr (DBG == 6) return;             // DBG 6: launch floor
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int b0 = mt * a.spt;
    const int ns = min(a.spt, a.Bp - b0);
    const int rows_out = ns * a.Lout;
    const int rows_in = ns * a.Lin;
    const int n0 = nt * TN;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);

    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    if constexpr (T > 0) {
        const int nch0 = (a.src[0].C + KC - 1) / KC;
        const int nch = a.CinP / KC;           // includes the host's zero padding stage (even stage count)
        const int c4 = tid % F4R, r0 = tid / F4R;
        // per-thread staging constants: clamped global row offsets (always in-bounds) + validity
        size_t goff[NP];
        bool rok[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + RPP * p;
            rok[p] = r < rows_in;
            goff[p] = (size_t)(b0 * a.Lin + min(r, rows_in - 1));
        }

        float bcur[KS][2], bnxt[KS][2];
        float4 areg[NP];
        // weights are packed in MFMA-fragment order [n-tile][stage][q][thread][4] (host: pack_weight): the B operands of
        // a whole stage are KS/2 float4 per lane, each wave-instruction reads 1 KiB contiguous, each stage 2*KS KiB
        static_assert((2 * KS) % 4 == 0, "stage fragment must be float4-sized");
        const float4* wbase = reinterpret_cast<const float4*>(a.W) + (size_t)nt * nch * 256 * (2 * KS / 4) + tid;
        auto load_b = [&](int ch, float (&b)[KS][2]) {
            const float4* wp = wbase + (size_t)ch * 256 * (2 * KS / 4);
#pragma unroll
            for (int q = 0; q < 2 * KS / 4; ++q) {
                const float4 v = wp[q * 256];
                b[2 * q][0] = v.x; b[2 * q][1] = v.y; b[2 * q + 1][0] = v.z; b[2 * q + 1][1] = v.w;
            }
        };
        // source of stage ch (branch-free select between the two concatenated sources)
        auto src_ptr = [&](int ch, int& cl, int& C, int& ld) -> const float* {
            const bool first = (ch < nch0) || (a.nsrc == 1);     // a padding stage of a single source stays on it
            cl = (first ? ch : ch - nch0) * KC + c4 * 4;
            C = first ? a.src[0].C : a.src[1].C;
            ld = first ? a.src[0].ld : a.src[1].ld;
            return first ? a.src[0].p : a.src[1].p;
        };
        float4 pg = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), ptb = pb;   // norm params of the stage in flight
        auto load_a = [&](int ch, float4 (&v)[NP]) {
            int cl, C, ld;
            const float* base = src_ptr(ch, cl, C, ld);
            const int clc = min(cl, C - 4);                    // clamped: always a valid address
#pragma unroll
            for (int p = 0; p < NP; ++p)
                v[p] = *reinterpret_cast<const float4*>(base + goff[p] * ld + clc);
            if constexpr (MODE == SRC_GN_MISH || MODE == SRC_LN) {
                const Src& s = a.src[0];
                pg = *reinterpret_cast<const float4*>(s.gamma + clc);
                if constexpr (MODE == SRC_GN_MISH) {
                    pb = *reinterpret_cast<const float4*>(s.beta + clc);
                    if (s.tb) ptb = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + clc);
                }
            }
        };
        // issue the first stage's global loads before anything else: their latency overlaps the set-up below
        load_b(0, bcur);
        load_a(0, areg);

        // ---- zero rows + normalisation tables ---------------------------------------------
        for (int i = tid; i < 2 * LDAK; i += 256) As[i / LDAK][ZR * LDAK + (i % LDAK)] = 0.f;
        if constexpr (MODE == SRC_GN_MISH) {
            const Src& s = a.src[0];
            for (int i = tid; i < ns * 8; i += 256) {
                float m, r;
                merge_stats(s.stats + ((size_t)(b0 + i / 8) * 8 + (i & 7)) * s.P * 2, s.P, s.cnt, 1e-5f, m, r);
                tabA[2 * i] = m; tabA[2 * i + 1] = r;
            }
        } else if constexpr (MODE == SRC_LN) {
            const Src& s = a.src[0];
            for (int i = tid; i < rows_in; i += 256) {
                float m, r;
                merge_stats(s.stats + (size_t)(b0 * a.Lin + i) * s.P * 2, s.P, s.cnt, 1e-5f, m, r);
                tabA[2 * i] = m; tabA[2 * i + 1] = r;
            }
        }
        if (a.e_y) {
            for (int i = tid; i < ns * 8; i += 256) {
                float m, r;
                merge_stats(a.e_stats + ((size_t)(b0 + i / 8) * 8 + (i & 7)) * a.e_P * 2, a.e_P, a.e_cnt, 1e-5f, m, r);
                tabE[2 * i] = m; tabE[2 * i + 1] = r;
            }
        }

        // ---- per-lane A fragment addresses (float index into one As buffer) ----------------
        int aaddr[3][T];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
            const int r = mb * 16 + (lane & 15);
            const int s = (r * a.lout_magic) >> 16, lo = r - s * a.Lout;
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                int li; bool ok;
                if (!a.transposed) { li = lo * a.stride + tap - a.pad; ok = (li >= 0) && (li < a.Lin); }
                else { const int q = lo + a.pad - tap; li = q >> 1; ok = (q >= 0) && !(q & 1) && (li < a.Lin); }
                const int row = (ok && r < rows_out) ? s * a.Lin + li : ZR;
                aaddr[mb][tap] = row * LDAK + w * CPW + (lane >> 4);
            }
        }
        int srow8[NP];                       // (sample of staged row) * 8, for the GN table
#pragma unroll
        for (int p = 0; p < NP; ++p) srow8[p] = ((min(r0 + RPP * p, rows_in - 1) * a.lin_magic) >> 16) * 8;
        const int gw_shift = 31 - __builtin_clz(a.src[0].gw | 1);      // group widths are powers of two (host checks)

        auto store_a = [&](int ch, int buf, const float4 (&av)[NP]) {
            int cl, C, ld;
            (void)src_ptr(ch, cl, C, ld);
            const bool cok = cl < C;
            const int clc = min(cl, C - 4);
            float* dst = As[buf];
            const float4 g = pg, bt = pb, tb = ptb;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int r = r0 + RPP * p;
                float4 v = av[p];
                if constexpr (MODE == SRC_GN_MISH) {
                    const int ti = (srow8[p] + (clc >> gw_shift)) * 2;
                    const float m = tabA[ti], rs = tabA[ti + 1];
                    v.x = mish_f((v.x - m) * rs * g.x + bt.x) + tb.x;
                    v.y = mish_f((v.y - m) * rs * g.y + bt.y) + tb.y;
                    v.z = mish_f((v.z - m) * rs * g.z + bt.z) + tb.z;
                    v.w = mish_f((v.w - m) * rs * g.w + bt.w) + tb.w;
                } else if constexpr (MODE == SRC_LN) {
                    const int rc = min(r, rows_in - 1);
                    const float m = tabA[2 * rc], rs = tabA[2 * rc + 1];
                    v.x = (v.x - m) * rs * g.x; v.y = (v.y - m) * rs * g.y;
                    v.z = (v.z - m) * rs * g.z; v.w = (v.w - m) * rs * g.w;
                } else if constexpr (MODE == SRC_MISH) {
                    v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w);
                } else if constexpr (MODE == SRC_GELU) {    // exact (erf) GELU, nn.GELU() default
                    v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752f));
                    v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752f));
                } else if constexpr (MODE == SRC_SILU) {    // accurate form (init-time table only)
                    v.x = v.x / (1.0f + expf(-v.x)); v.y = v.y / (1.0f + expf(-v.y));
                    v.z = v.z / (1.0f + expf(-v.z)); v.w = v.w / (1.0f + expf(-v.w));
                }
                const bool ok = rok[p] && cok;              // zero padding is applied AFTER the activation
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                float2* d2 = reinterpret_cast<float2*>(dst + r * LDAK + c4 * 4);
                d2[0] = make_float2(v.x, v.y);
                d2[1] = make_float2(v.z, v.w);
            }
        };

        __syncthreads();          // tables + zero rows visible
        store_a(0, 0, areg);
        __syncthreads();
        // one pipeline stage: prefetch stage ch+1 (B -> bn, A -> areg), multiply stage ch from LDS buffer ch&1 with bc,
        // write the prefetched A rows to the other LDS buffer, barrier.  Two explicit B register sets alternate
        // (no register copies: hipcc hoists a copy of in-flight registers above the MFMA block and stalls on it).
        auto stage = [&](int ch, const float (&bc)[KS][2], float (&bn)[KS][2]) {
            const int chn = min(ch + 1, nch - 1);          // tail: harmless re-load of the last stage
            if constexpr (DBG != 1 && DBG != 5) {          // DBG 1/5: ablate the global loads (timing experiments only)
                load_b(chn, bn);
                load_a(chn, areg);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ABOVE the MFMA block (hipcc sinks it otherwise)
            const float* Ab = As[ch & 1];
            if constexpr (DBG != 2) {                      // DBG 2: ablate the MFMA block
#pragma unroll
                for (int tap = 0; tap < T; ++tap)
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) {
                        float a0, a1, a2;
                        if constexpr (DBG == 4 || DBG == 5) {      // DBG 4/5: no LDS reads (operands from registers)
                            a0 = bc[tap * CS + cs][0] + 1.f; a1 = bc[tap * CS + cs][1] + 2.f; a2 = a0 + a1;
                        } else {
                            a0 = Ab[aaddr[0][tap] + cs * 4];
                            a1 = Ab[aaddr[1][tap] + cs * 4];
                            a2 = Ab[aaddr[2][tap] + cs * 4];
                        }
                        const float b0v = bc[tap * CS + cs][0], b1v = bc[tap * CS + cs][1];
                        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0v, acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1v, acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0v, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1v, acc[1][1], 0, 0, 0);
                        acc[2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b0v, acc[2][0], 0, 0, 0);
                        acc[2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b1v, acc[2][1], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DBG != 3) store_a(chn, (ch + 1) & 1, areg);     // DBG 3: ablate the LDS staging store
            __syncthreads();
        };
        if constexpr (DBG == 7 || DBG == 8) {              // DBG 7: prologue + epilogue only
        } else if (nch == 1) {
            stage(0, bcur, bnxt);
        } else {                                           // host guarantees an even stage count (pack_weight)
            for (int ch = 0; ch < nch; ch += 2) {
                stage(ch, bcur, bnxt);
                stage(ch + 1, bnxt, bcur);
            }
        }
    } else {
        if (a.e_y) {
            for (int i = tid; i < ns * 8; i += 256) {
                float m, r;
                merge_stats(a.e_stats + ((size_t)(b0 + i / 8) * 8 + (i & 7)) * a.e_P * 2, a.e_P, a.e_cnt, 1e-5f, m, r);
                tabE[2 * i] = m; tabE[2 * i + 1] = r;
            }
        }
        __syncthreads();
    }

    gemm_epilogue<(T > 0)>(a, acc, Red, tabE, DBG == 8);
}

// ---------------------------------------------------------------------------------------------
// conv_gemm_h3_kernel: the same implicit GEMM with every fp32 product a*b evaluated on the fp16 matrix cores as
//     a*b ~= ah*bh + 2^-11 * (ah*bl' + al'*bh),   ah = fp16(a), al' = fp16((a - ah) * 2^11)   (same for b)
// three v_mfma_f32_16x16x32_f16 with fp32 accumulation (two accumulator sets: main, and the 2^11-scaled low-order
// terms).  a - ah is exact in fp32, the scaling keeps al' out of the fp16 subnormal range, the products of two
// 11-bit significands are exact in the fp32 accumulator, so the only losses are the roundings of al', bl' (2^-24
// relative to a, b) and the dropped al*bl (2^-24): fp32-faithful, at 16/3 of the fp32 MFMA rate.
// Stage = 128 input channels (one 32-channel k-group per wave) x T taps.  A is staged as two fp16 planes
// (hi, scaled lo) and read as ds_read_b128 fragments; B (pre-split and packed on the host in fragment order) is
// re-loaded tap by tap for the next stage right after its last use, so its prefetch distance is T-1 taps.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
constexpr float H3_SCALE = 2048.0f, H3_INV = 1.0f / 2048.0f;

// ABL (timing ablations of the main loop, wrong results): 1 no weight reloads, 2 no MFMAs, 3 no activation restaging
// (and no per-stage barrier), 4 = 1 + 3, 5 no LDS fragment reads
template <int T, int ROWS, int MODE, bool RES = false, int ABL = 0>
__global__ __launch_bounds__(256) void conv_gemm_h3_kernel(const GemmArgs a) {
    constexpr int KC = 128;
    constexpr int PITCH = 272;              // bytes per LDS row per plane: 128 halfs + 16 B pad
    constexpr int F4R = KC / 4;             // float4 per staged row (32)
    constexpr int RPP = 256 / F4R;          // rows per staging pass (8)
    constexpr int NP = (ROWS + RPP - 1) / RPP;
    constexpr int LROWS = NP * RPP;
    constexpr int ZR = LROWS;
    constexpr int PLANE = (LROWS + 1) * PITCH;
    __shared__ __attribute__((aligned(16))) unsigned char Ah[2][2][PLANE];      // [buffer][plane hi/lo]
    __shared__ __attribute__((aligned(16))) float Red[4][TM * LDR];
    __shared__ float tabA[TM * 8 * 2];
    __shared__ float tabE[TM * 8 * 2];

    if (a.dbg == 10) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int b0 = mt * a.spt;
    const int ns = min(a.spt, a.Bp - b0);
    const int rows_out = ns * a.Lout;
    const int rows_in = ns * a.Lin;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);

    f32x4 accM[3][2], accL[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nch0 = (a.src[0].C + KC - 1) / KC;
    const int nch = a.CinP / KC;
    const int c4 = tid % F4R, r0 = tid / F4R;
    size_t goff[NP];
    bool rok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = r0 + RPP * p;
        rok[p] = r < rows_in;
        goff[p] = (size_t)(b0 * a.Lin + min(r, rows_in - 1));
    }

    // B: [n-tile][stage][q = (tap*2 + nb)*2 + plane][thread][8 halfs]
    half8 breg[T][2][2];
    const uint4* wbase = reinterpret_cast<const uint4*>(a.W) + (size_t)nt * nch * (T * 4) * 256 + tid;
    auto load_b_tap = [&](int ch, int tap) {
        const uint4* wp = wbase + ((size_t)ch * (T * 4) + tap * 4) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = wp[q * 256];
            breg[tap][q >> 1][q & 1] = __builtin_bit_cast(half8, v);
        }
    };
    float4 areg[NP];
    float4 pg = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), ptb = pb;
    auto src_ptr = [&](int ch, int& cl, int& C, int& ld) -> const float* {
        const bool first = (ch < nch0) || (a.nsrc == 1);
        cl = (first ? ch : ch - nch0) * KC + c4 * 4;
        C = first ? a.src[0].C : a.src[1].C;
        ld = first ? a.src[0].ld : a.src[1].ld;
        return first ? a.src[0].p : a.src[1].p;
    };
    auto load_a = [&](int ch) {
        int cl, C, ld;
        const float* base = src_ptr(ch, cl, C, ld);
        const int clc = min(cl, C - 4);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            areg[p] = *reinterpret_cast<const float4*>(base + goff[p] * ld + clc);
        if constexpr (MODE == SRC_GN_MISH || MODE == SRC_LN) {
            const Src& s = a.src[0];
            pg = *reinterpret_cast<const float4*>(s.gamma + clc);
            if constexpr (MODE == SRC_GN_MISH) {
                pb = *reinterpret_cast<const float4*>(s.beta + clc);
                if (s.tb) ptb = *reinterpret_cast<const float4*>(s.tb + (size_t)t_now * s.tb_ld + clc);
            }
        }
    };
#pragma unroll
    for (int tap = 0; tap < T; ++tap) load_b_tap(0, tap);
    // optional second GEMM on the centre tap (residual_conv): its own fragments and accumulators
    half8 rreg[2][2];
    f32x4 accRM[3][2], accRL[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { accRM[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accRL[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const uint4* rbase = reinterpret_cast<const uint4*>(a.W2) + (size_t)nt * nch * 4 * 256 + tid;
    auto load_r = [&](int ch) {
        if constexpr (RES) {
            const uint4* wp = rbase + (size_t)ch * 4 * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) rreg[q >> 1][q & 1] = __builtin_bit_cast(half8, wp[q * 256]);
        }
    };
    load_r(0);
    load_a(0);

    for (int i = tid; i < 4 * (PITCH / 4); i += 256)
        reinterpret_cast<float*>(&Ah[(i / (PITCH / 4)) >> 1][(i / (PITCH / 4)) & 1][ZR * PITCH])[i % (PITCH / 4)] = 0.f;
    if constexpr (MODE == SRC_GN_MISH) {
        const Src& s = a.src[0];
        for (int i = tid; i < ns * 8; i += 256) {
            float m, r;
            merge_stats(s.stats + ((size_t)(b0 + i / 8) * 8 + (i & 7)) * s.P * 2, s.P, s.cnt, 1e-5f, m, r);
            tabA[2 * i] = m; tabA[2 * i + 1] = r;
        }
    } else if constexpr (MODE == SRC_LN) {
        const Src& s = a.src[0];
        for (int i = tid; i < rows_in; i += 256) {
            float m, r;
            merge_stats(s.stats + (size_t)(b0 * a.Lin + i) * s.P * 2, s.P, s.cnt, 1e-5f, m, r);
            tabA[2 * i] = m; tabA[2 * i + 1] = r;
        }
    }
    if (a.e_y) {
        for (int i = tid; i < ns * 8; i += 256) {
            float m, r;
            merge_stats(a.e_stats + ((size_t)(b0 + i / 8) * 8 + (i & 7)) * a.e_P * 2, a.e_P, a.e_cnt, 1e-5f, m, r);
            tabE[2 * i] = m; tabE[2 * i + 1] = r;
        }
    }

    // per-lane A fragment byte offsets inside one plane: row(i, tap) * PITCH + (wave's k-group) * 64 + (lane>>4) * 16
    int aaddr[3][T];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        const int r = mb * 16 + (lane & 15);
        const int s = (r * a.lout_magic) >> 16, lo = r - s * a.Lout;
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            int li; bool ok;
            if (!a.transposed) { li = lo * a.stride + tap - a.pad; ok = (li >= 0) && (li < a.Lin); }
            else { const int q = lo + a.pad - tap; li = q >> 1; ok = (q >= 0) && !(q & 1) && (li < a.Lin); }
            const int row = (ok && r < rows_out) ? s * a.Lin + li : ZR;
            aaddr[mb][tap] = row * PITCH + w * 64 + (lane >> 4) * 16;
        }
    }
    int srow8[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) srow8[p] = ((min(r0 + RPP * p, rows_in - 1) * a.lin_magic) >> 16) * 8;
    const int gw_shift = 31 - __builtin_clz(a.src[0].gw | 1);

    auto store_a = [&](int ch, int buf, int p_lo, int p_hi) {      // staging rows [p_lo, p_hi) of the NP per thread
        int cl, C, ld;
        (void)src_ptr(ch, cl, C, ld);
        const bool cok = cl < C;
        const int clc = min(cl, C - 4);
        const float4 g = pg, bt = pb, tb = ptb;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p < p_lo || p >= p_hi) continue;      // constant-folded: callers pass compile-time ranges after unrolling
            const int r = r0 + RPP * p;
            float4 v = areg[p];
            if constexpr (MODE == SRC_GN_MISH) {
                const int ti = (srow8[p] + (clc >> gw_shift)) * 2;
                const float m = tabA[ti], rs = tabA[ti + 1];
                v.x = mish_f((v.x - m) * rs * g.x + bt.x) + tb.x;
                v.y = mish_f((v.y - m) * rs * g.y + bt.y) + tb.y;
                v.z = mish_f((v.z - m) * rs * g.z + bt.z) + tb.z;
                v.w = mish_f((v.w - m) * rs * g.w + bt.w) + tb.w;
            } else if constexpr (MODE == SRC_LN) {
                const int rc = min(r, rows_in - 1);
                const float m = tabA[2 * rc], rs = tabA[2 * rc + 1];
                v.x = (v.x - m) * rs * g.x; v.y = (v.y - m) * rs * g.y;
                v.z = (v.z - m) * rs * g.z; v.w = (v.w - m) * rs * g.w;
            } else if constexpr (MODE == SRC_MISH) {
                v.x = mish_f(v.x); v.y = mish_f(v.y); v.z = mish_f(v.z); v.w = mish_f(v.w);
            }
            const bool ok = rok[p] && cok;
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&Ah[buf][0][r * PITCH + c4 * 8]) = hi;
            *reinterpret_cast<half4v*>(&Ah[buf][1][r * PITCH + c4 * 8]) = lo;
        }
    };

    __syncthreads();
    store_a(0, 0, 0, NP);
    __syncthreads();
    if (a.dbg == 11) return;                          // prologue only
    for (int ch = 0; ch < (a.dbg == 7 ? 0 : nch); ++ch) {
        const int chn = min(ch + 1, nch - 1);
        if constexpr (ABL != 3 && ABL != 4) load_a(chn);
        const unsigned char* P0 = Ah[(ABL == 3 || ABL == 4) ? 0 : (ch & 1)][0];
        const unsigned char* P1 = Ah[(ABL == 3 || ABL == 4) ? 0 : (ch & 1)][1];
        // A fragments are read one tap AHEAD of their use into an explicit second register set (hipcc otherwise
        // re-uses one set and exposes the LDS latency in front of every 6 MFMAs)
        half8 fh[2][3], fl[2][3];
        auto read_frags = [&](int tap, half8 (&ah)[3], half8 (&al)[3]) {
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                ah[mb] = *reinterpret_cast<const half8*>(P0 + aaddr[mb][tap]);
                al[mb] = *reinterpret_cast<const half8*>(P1 + aaddr[mb][tap]);
            }
        };
        if (ABL != 5 || ch == 0) read_frags(0, fh[0], fl[0]);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            if (tap + 1 < T && (ABL != 5 || ch == 0)) read_frags(tap + 1, fh[(tap + 1) & 1], fl[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const half8 (&ah)[3] = fh[tap & 1];
            const half8 (&al)[3] = fl[tap & 1];
            if constexpr (ABL == 2) {
#pragma unroll
                for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        accM[mb][nb][0] += (float)ah[mb][0] * (float)breg[tap][nb][0][0];
                        accL[mb][nb][0] += (float)al[mb][0] * (float)breg[tap][nb][1][0];
                    }
            } else {
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    accM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], breg[tap][nb][0], accM[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], breg[tap][nb][1], accL[mb][nb], 0, 0, 0);
                    accL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], breg[tap][nb][0], accL[mb][nb], 0, 0, 0);
                }
            }
            if constexpr (RES) if (tap == T / 2) {     // the 1x1 residual_conv on the same (centre-tap) rows
#pragma unroll
                for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        accRM[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], rreg[nb][0], accRM[mb][nb], 0, 0, 0);
                        accRL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mb], rreg[nb][1], accRL[mb][nb], 0, 0, 0);
                        accRL[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mb], rreg[nb][0], accRL[mb][nb], 0, 0, 0);
                    }
                load_r(chn);
            }
            if constexpr (ABL != 1 && ABL != 4) load_b_tap(chn, tap);          // next stage's fragments for this tap, T-1 taps ahead of their use
        }
        // (staging the next stage's rows in per-tap slices was measured: no faster, and one build of it was flaky)
        if constexpr (ABL != 3 && ABL != 4) {
            store_a(chn, (ch + 1) & 1, 0, NP);
            __syncthreads();
        }
    }

    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = accM[i][j] + accL[i][j] * H3_INV;
    if (a.dbg == 9) { if (acc[0][0][0] == 123.456f) a.out[0] = 1.f; return; }      // no epilogue
    gemm_epilogue<true>(a, acc, Red, tabE, a.dbg == 8);
    if constexpr (RES) {
        // second output: out2 = W2 . x + bias2 (plain epilogue: reduce, bias, store)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = accRM[i][j] + accRL[i][j] * H3_INV;
        GemmArgs a2 = a;
        a2.out = a.out2; a2.ldo = a.ldo2; a2.bias = a.bias2;
        a2.act_gamma = nullptr; a2.stats_out = nullptr; a2.ln_out = nullptr; a2.e_y = nullptr; a2.res = nullptr;
        __syncthreads();
        gemm_epilogue<true>(a2, acc, Red, tabE, true);
    }
}

// ---------------------------------------------------------------------------------------------
// Linear attention core (LinearAttentionTemporal.forward, model/diffusion_1d.py:281-291) for one
// (sample, head) per wave: q *= 32^-0.5; k = softmax over positions; ctx[d][e] = sum_n k[d][n] v[e][n];
// out[e][n] = sum_d ctx[d][e] q[d][n].   qkv: [rows, 384] (q | k | v, each heads*32 channels, head-major);
// att: [rows, 128].  grid = Bp, block = 256 (4 heads).
constexpr int ATT_MAXL = 48;
__global__ __launch_bounds__(256) void linattn_core_kernel(const float* __restrict__ qkv, float* __restrict__ att, int L) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    float* q = sm + (size_t)h * (3 * L * 32 + 32 * 33);
    float* k = q + L * 32;
    float* v = k + L * 32;
    float* ctx = v + L * 32;                       // [32][33]
    const size_t row0 = (size_t)blockIdx.x * L;
    const int d = lane & 31, half = lane >> 5;
    const float scale = 0.17677669529663687f;      // 32 ** -0.5
    // each lane owns channel d of rows n = half, half + 2, ...: all loads issued up front, k kept in registers
    constexpr int MAXH = ATT_MAXL / 2;
    float kv[MAXH];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXH; ++j) {
        const int n = half + 2 * j;
        kv[j] = -INFINITY;
        if (n < L) {
            const float* p = qkv + (row0 + n) * 384 + h * 32 + d;
            q[n * 32 + d] = p[0] * scale;
            kv[j] = p[128];
            v[n * 32 + d] = p[256];
        }
    }
#pragma unroll
    for (int j = 0; j < MAXH; ++j) mx = fmaxf(mx, kv[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));            // softmax over positions: both halves of the wave hold channel d
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXH; ++j) {
        const int n = half + 2 * j;
        if (n < L) {
            const float e = __builtin_amdgcn_exp2f((kv[j] - mx) * 1.4426950408889634f);
            k[n * 32 + d] = e;                         // unnormalised; 1 / sum is folded into the context rows below
            sum += e;
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    __syncthreads();
    // ctx[d][e], e in [half*16, half*16+16)
    float c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    for (int n = 0; n < L; ++n) {
        const float kk = k[n * 32 + d];
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] += kk * v[n * 32 + half * 16 + j];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] *= inv;
#pragma unroll
    for (int j = 0; j < 16; ++j) ctx[d * 33 + half * 16 + j] = c[j];
    __syncthreads();
    // out[e][n] for e = lane&31, n = half, half+2, ...
    const int e = d;
    for (int n = half; n < L; n += 2) {
        float o = 0.f;
#pragma unroll 8
        for (int dd = 0; dd < 32; ++dd) o += ctx[dd * 33 + e] * q[n * 32 + dd];
        att[(row0 + n) * 128 + h * 32 + e] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// One launch per Residual(PreNorm(LinearAttentionTemporal)) site (model/diffusion_1d.py:75-81, :123-142, :272-291):
//   out = x + Wo * linattn(Wqkv * (LN(x) g)) + bo
// one sample per workgroup, wave = head.  Every product runs on the fp32 MFMA with operands chosen so that the
// accumulator layout of one product IS the operand layout of the next (no transposes through LDS):
//   q  = Wq y^T      rows = channels, cols = positions        (A = weights, B = y)
//   k,v = y Wk^T     rows = positions, cols = channels        (A = y, B = weights; the same LDS / weight fragments)
//   ctx[d][e] = sum_n k^[d][n] v[e][n]        A = k^ accumulators, B = v accumulators (contraction index = rows)
//   att[e][n] = sum_d ctx[d][e] q[d][n]       A = ctx accumulators, B = q accumulators
// att goes through LDS once ([position][head*32 + e]) for the output projection, whose N (channels) is split over
// the waves.  Weight fragments: [tile of 16 channels][k16][lane][4], element j <-> k = k16*16 + (lane/16)*4 + j.
struct AttnSiteArgs {
    const float* x; int ldx;
    float* out; int ldo;
    const float* g;        // LayerNorm gain [C]
    const float* Wqkv;     // [24 tiles][C/16][64][4]
    const float* Wo;       // [C/16 tiles][8][64][4]
    const float* bo;       // [C]
    int L;                 // positions per sample
    int S;                 // samples per workgroup
    int slot;              // tile positions per sample: L rounded up to a multiple of 4 when S > 1, so that the order of
                           // every per-sample reduction (hence every bit of the result) is independent of the slot
    int Bp;                // samples in the batch
    int dbg;               // timing ablations (wrong results): 1 no qkv loop, 2 no projection, 3 no core
    Pf pf;                 // L2 warm-up for the next launch
    PhaseBuf ph;           // phase clocks (profiling builds)
};

// Core of a site for the head of this wave, per sample s of the workgroup: q *= 32^-1/2 ; k = softmax over the sample's
// positions ; ctx_s = k v^T ; att = ctx_s^T q.  qa: rows = channels, cols = positions; ka, va: rows = positions, cols =
// channels (accumulator layouts of the 16x16 MFMAs); att: rows = e, cols = positions.
template <int NT>
__device__ __forceinline__ void attn_site_core_range(f32x4 (&qa)[2][NT], f32x4 (&ka)[NT][2], f32x4 (&va)[NT][2], f32x4 (&att)[2][NT],
                                                     int s_begin, int s_end, int nend, int slot, int L, int lq, int lr) {
    const float scale = 0.17677669529663687f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) qa[i][nt] *= scale;
    // sample of each accumulator row (k, v: rows = positions nt*16 + lq*4 + i) and column (q: cols = positions nt*16 + lr)
    int sid_row[NT][4], sid_col[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int n = nt * 16 + lq * 4 + i; sid_row[nt][i] = (n < nend && n % slot < L) ? n / slot : -1; }
        const int n = nt * 16 + lr;
        sid_col[nt] = (n < nend && n % slot < L) ? n / slot : -1;
    }
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) att[et][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        f32x4 ks[NT][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) if (sid_row[nt][i] == s) mx = fmaxf(mx, ka[nt][dt][i]);
            mx = xmax32(xmax16(mx));
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = (sid_row[nt][i] == s) ? __builtin_amdgcn_exp2f((ka[nt][dt][i] - mx) * 1.4426950408889634f) : 0.f;
                    ks[nt][dt][i] = e;
                    sum += e;
                }
            sum = xsum32(xsum16(sum));
            const float inv = 1.0f / sum;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ks[nt][dt] *= inv;
        }
        f32x4 ctx[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ks[nt][dt][i], va[nt][et][i], c, 0, 0, 0);
                ctx[dt][et] = c;
            }
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool mine = sid_col[nt] == s;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        att[et][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ctx[dt][et][i], mine ? qa[dt][nt][i] : 0.f, att[et][nt], 0, 0, 0);
            }
    }
}

template <int NT>
__device__ __forceinline__ void attn_site_core(f32x4 (&qa)[2][NT], f32x4 (&ka)[NT][2], f32x4 (&va)[NT][2], f32x4 (&att)[2][NT],
                                               int s_here, int nend, int slot, int L, int lq, int lr) {
    attn_site_core_range<NT>(qa, ka, va, att, 0, s_here, nend, slot, L, lq, lr);
}

template <int C, int NT, int PF>
__global__ __launch_bounds__(256) void attn1d_site_kernel(const AttnSiteArgs a) {
    constexpr int NP = NT * 16, YP = C + 4, AP = 132, K16 = C / 16, CT = C / 16;
    constexpr int CH = (C + 255) / 256;                  // float4 chunks per lane per row (LayerNorm phase)
    constexpr int RW = NP / 4;                           // rows per wave in the LayerNorm phase
    __shared__ __attribute__((aligned(16))) float Ys[NP * YP];
    __shared__ __attribute__((aligned(16))) float At[NP * AP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int L = a.L;
    const int s_here = min(a.S, a.Bp - (int)blockIdx.x * a.S);      // samples of this workgroup
    const int slot = a.slot;
    const int nend = s_here * slot;                                  // tile positions in use (pads inside a slot are zero rows)
    const size_t row0 = (size_t)blockIdx.x * a.S * L;                // tile position n <-> row row0 + (n / slot) * L + n % slot
    const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
    // this wave's six channel tiles: q (2h, 2h+1), k (8+2h, ..), v (16+2h, ..)
    int tile[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) tile[s] = (s >> 1) * 8 + 2 * w + (s & 1);
    // weight ring: PF k16-steps in flight
    float4 wr[PF][6];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < 6; ++s) wr[p][s] = (p < K16) ? Wq4[((size_t)tile[s] * K16 + p) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- LayerNorm over channels (biased variance, eps 1e-5), two-pass.  LPR lanes share a row (RPP rows per wave and
    // pass), so the butterflies are log2(LPR) deep; wave w owns tile positions [w*RW, (w+1)*RW) ----
    {
        constexpr int LPR = (C / 4 < 64) ? C / 4 : 64;   // lanes per row
        constexpr int RPP = 64 / LPR;                    // rows per pass
        constexpr int NPASS = (RW + RPP - 1) / RPP;
        const int lrow = lane / LPR, lcol = lane % LPR;
        float4 xr[NPASS][CH];
        bool okr[NPASS];
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int n = w * RW + r * RPP + lrow, sn = n / slot, pn = n - sn * slot;
            okr[r] = (r * RPP + lrow < RW) && n < nend && pn < L;
#pragma unroll
            for (int m = 0; m < CH; ++m)
                xr[r][m] = okr[r] ? *reinterpret_cast<const float4*>(a.x + (row0 + sn * L + pn) * a.ldx + 4 * (lcol + LPR * m)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 gv[CH];
#pragma unroll
        for (int m = 0; m < CH; ++m) gv[m] = *reinterpret_cast<const float4*>(a.g + 4 * (lcol + LPR * m));
#pragma unroll
        for (int r = 0; r < NPASS; ++r) {
            const int n = w * RW + r * RPP + lrow;
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
                    *reinterpret_cast<float4*>(&Ys[n * YP + 4 * (lcol + LPR * m)]) = y;
                }
            }
        }
    }
    __syncthreads();

    // ---- q, k, v of head w ----
    f32x4 qa[2][NT], ka[NT][2], va[NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { qa[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; ka[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; va[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (a.dbg != 1)
#pragma unroll
    for (int k16 = 0; k16 < K16; ++k16) {
        float4 wc[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) wc[s] = wr[k16 % PF][s];
        if (k16 + PF < K16) {
#pragma unroll
            for (int s = 0; s < 6; ++s) wr[k16 % PF][s] = Wq4[((size_t)tile[s] * K16 + k16 + PF) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the refill ahead of this step's MFMAs (PF steps in flight)
        float4 yv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) yv[nt] = *reinterpret_cast<const float4*>(&Ys[(nt * 16 + lr) * YP + k16 * 16 + lq * 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w0 = (&wc[0].x)[j], w1 = (&wc[1].x)[j], w2 = (&wc[2].x)[j], w3 = (&wc[3].x)[j], w4 = (&wc[4].x)[j], w5 = (&wc[5].x)[j];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float y = (&yv[nt].x)[j];
                qa[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, y, qa[0][nt], 0, 0, 0);
                qa[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, y, qa[1][nt], 0, 0, 0);
                ka[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, w2, ka[nt][0], 0, 0, 0);
                ka[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, w3, ka[nt][1], 0, 0, 0);
                va[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, w4, va[nt][0], 0, 0, 0);
                va[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, w5, va[nt][1], 0, 0, 0);
            }
        }
    }
    // first projection tile of this wave: in flight during the attention core
    const float4* Wo4 = reinterpret_cast<const float4*>(a.Wo);
    static_assert(CT % 4 == 0, "output channel tiles are split evenly over the four waves");
    constexpr int TPW = CT / 4;                          // output channel tiles per wave
    constexpr int WOR = 3;                               // projection weight ring: two tiles in flight
    float4 wo[WOR][8];
#pragma unroll
    for (int t = 0; t < WOR - 1; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) wo[t][k] = (t < TPW) ? Wo4[((size_t)(w * TPW + t) * 8 + k) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- core ----
    f32x4 att[2][NT];
    attn_site_core<NT>(qa, ka, va, att, a.dbg == 3 ? 0 : s_here, nend, slot, L, lq, lr);
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4*>(&At[(nt * 16 + lr) * AP + w * 32 + et * 16 + lq * 4]) = att[et][nt];
    __syncthreads();

    // ---- out = Wo att + bo + x : channel tiles [w*TPW, (w+1)*TPW) of this wave ----
    if (a.dbg != 2)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int ct = w * TPW + t;
        if (t + WOR - 1 < TPW) {
#pragma unroll
            for (int k = 0; k < 8; ++k) wo[(t + WOR - 1) % WOR][k] = Wo4[((size_t)(ct + WOR - 1) * 8 + k) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 z[NT], z1[NT];                             // two chains per tile (head pairs) keep the matrix pipe busy
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { z[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; z1[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 wv = wo[t % WOR][k], wv1 = wo[t % WOR][k + 4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 av = *reinterpret_cast<const float4*>(&At[(nt * 16 + lr) * AP + k * 16 + lq * 4]);
                const float4 av1 = *reinterpret_cast<const float4*>(&At[(nt * 16 + lr) * AP + (k + 4) * 16 + lq * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32((&wv.x)[j], (&av.x)[j], z[nt], 0, 0, 0);
                    z1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32((&wv1.x)[j], (&av1.x)[j], z1[nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[nt] += z1[nt];
        const int c = ct * 16 + lq * 4;
        const float4 b = *reinterpret_cast<const float4*>(a.bo + c);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + lr, sn = n / slot, pn = n - sn * slot;
            if (n < nend && pn < L) {
                const size_t row = row0 + sn * L + pn;
                const float4 xv = *reinterpret_cast<const float4*>(a.x + row * a.ldx + c);
                float4 o;
                o.x = z[nt][0] + b.x + xv.x; o.y = z[nt][1] + b.y + xv.y; o.z = z[nt][2] + b.z + xv.z; o.w = z[nt][3] + b.w + xv.w;
                *reinterpret_cast<float4*>(a.out + row * a.ldo + c) = o;
            }
        }
    }
}

// The same site with the two projections (to_qkv, to_out) on the split-fp16 scheme of conv_gemm_h3_kernel: LN(x)g and
// att are staged as (hi, scaled lo) fp16 planes, the weights are pre-split on the host; three
// v_mfma_f32_16x16x32_f16 per 32 channels instead of eight fp32 MFMAs.  The attention core stays in fp32 (its
// operands are accumulators).  Weight fragments: [tile of 16 channels][k32][plane][lane][8 halfs], element e <->
// k = k32*32 + (lane/16)*8 + e.
template <int C, int NT, int PF>
__global__ __launch_bounds__(256) void attn1d_site_h3_kernel(const AttnSiteArgs a) {
    PH_DECL;
    PH(0);        // phase clocks (profiling builds): 0 entry, 1 LayerNorm -> planes, 2 q | k | v + core + att planes, 3 out projection + stores
    constexpr int NP = NT * 16, K32 = C / 32, CT = C / 16;
    constexpr int YPB = 2 * C + 16;                      // bytes per position per plane (pad keeps b128 reads conflict-free)
    constexpr int APB = 2 * 128 + 16;
    constexpr int CH = (C + 255) / 256;
    constexpr int RW = NP / 4;
    __shared__ __attribute__((aligned(16))) unsigned char Yp[2][NP * YPB];      // [plane][position]
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][NP * APB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int L = a.L;
    const int s_here = min(a.S, a.Bp - (int)blockIdx.x * a.S);
    const int slot = a.slot;
    const int nend = s_here * slot;
    const size_t row0 = (size_t)blockIdx.x * a.S * L;
    // ---- the rows first (round 4): they come from the previous launch, i.e. from memory, and the LayerNorm needs them before
    // anything else; requested behind the weight ring they arrived behind it (loads return in order) ----
    constexpr int LPR = (C / 4 < 64) ? C / 4 : 64;
    constexpr int RPP = 64 / LPR;
    constexpr int NPASS = (RW + RPP - 1) / RPP;
    const int lrow = lane / LPR, lcol = lane % LPR;
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);          // (the A/B path only -- `tune` bit 0 --, in front of every other request)
    __builtin_amdgcn_sched_barrier(0);
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
    const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);      // one float4 = 8 halfs
    int tile[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) tile[s] = (s >> 1) * 8 + 2 * w + (s & 1);
    // weight ring: PF k32-steps in flight, [tile][plane]
    // (round 6, as attn1d_head_kernel) half of the ring in front of the LayerNorm, the rest between its passes: requests issue in order against
    // the L1 path's back-pressure, and with all of them in front the LayerNorm's VALU work started when the last one was issued
    float4 wr[PF][6][2];
    auto load_ring = [&](int p) {
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                wr[p][s][pl] = (p < K32) ? Wq4[(((size_t)tile[s] * K32 + p) * 2 + pl) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    constexpr int PFRONT = (PF + 1) / 2;
#pragma unroll
    for (int p = 0; p < PFRONT; ++p) load_ring(p);
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm (as in attn1d_site_kernel), result split into the two fp16 planes ----
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
            if (PFRONT + r < PF) load_ring(PFRONT + r);
            if (r == NPASS - 1) {
#pragma unroll
                for (int p = PFRONT + NPASS; p < PF; ++p) load_ring(p);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    PH(1);

    // ---- q, k, v of head w: main and low-order accumulators ----
    f32x4 qM[2][NT], qL[2][NT], kM[NT][2], kL[NT][2], vM[NT][2], vL[NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            qM[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; qL[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            kM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; kL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            vM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; vL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    if (a.dbg != 1)
#pragma unroll
    for (int k32 = 0; k32 < K32; ++k32) {
        half8 wh[6], wl[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) { wh[s] = __builtin_bit_cast(half8, wr[k32 % PF][s][0]); wl[s] = __builtin_bit_cast(half8, wr[k32 % PF][s][1]); }
        if (k32 + PF < K32) {
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wr[k32 % PF][s][pl] = Wq4[(((size_t)tile[s] * K32 + k32 + PF) * 2 + pl) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int off = (nt * 16 + lr) * YPB + k32 * 64 + lq * 16;
            const half8 yh = *reinterpret_cast<const half8*>(&Yp[0][off]);
            const half8 yl = *reinterpret_cast<const half8*>(&Yp[1][off]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                qM[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yh, qM[i][nt], 0, 0, 0);
                qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yl, qL[i][nt], 0, 0, 0);
                qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], yh, qL[i][nt], 0, 0, 0);
                kM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[2 + i], kM[nt][i], 0, 0, 0);
                kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[2 + i], kL[nt][i], 0, 0, 0);
                kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[2 + i], kL[nt][i], 0, 0, 0);
                vM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[4 + i], vM[nt][i], 0, 0, 0);
                vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[4 + i], vL[nt][i], 0, 0, 0);
                vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[4 + i], vL[nt][i], 0, 0, 0);
            }
        }
    }
    f32x4 qa[2][NT], ka[NT][2], va[NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            qa[i][j] = qM[i][j] + qL[i][j] * H3_INV;
            ka[j][i] = kM[j][i] + kL[j][i] * H3_INV;
            va[j][i] = vM[j][i] + vL[j][i] * H3_INV;
        }
    // projection weights of this wave's first tiles: in flight during the attention core
    const float4* Wo4 = reinterpret_cast<const float4*>(a.Wo);
    static_assert(CT % 4 == 0, "output channel tiles are split evenly over the four waves");
    constexpr int TPW = CT / 4;
    constexpr int WOR = 3;
    float4 wo[WOR][4][2];                                // [ring][k32][plane]
#pragma unroll
    for (int t = 0; t < WOR - 1; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                wo[t][k][pl] = (t < TPW) ? Wo4[(((size_t)(w * TPW + t) * 4 + k) * 2 + pl) * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);

    // the first output tile's bias and residual rows, requested before the core (requested in the epilogue, their L2 round trip
    // was exposed once per launch; the later tiles' loads overlap the earlier tiles' products)
    float4 eb0 = make_float4(0.f, 0.f, 0.f, 0.f), ex0[NT];
    {
        const int c = (w * TPW) * 16 + lq * 4;
        eb0 = *reinterpret_cast<const float4*>(a.bo + c);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + lr, sn = n / slot, pn = n - sn * slot;
            // (row clamped, load unconditional: the rows beyond the samples are never stored)
            const size_t rowc = (n < nend && pn < L) ? row0 + sn * L + pn : row0;
            ex0[nt] = *reinterpret_cast<const float4*>(a.x + rowc * a.ldx + c);
        }
    }
    // ---- core ----
    f32x4 att[2][NT];
    attn_site_core<NT>(qa, ka, va, att, a.dbg == 3 ? 0 : s_here, nend, slot, L, lq, lr);
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = att[et][nt];
            half4v hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)v[i]; lo[i] = (_Float16)((v[i] - (float)hi[i]) * H3_SCALE); }
            const int off = (nt * 16 + lr) * APB + 2 * (w * 32 + et * 16 + lq * 4);
            *reinterpret_cast<half4v*>(&Ap[0][off]) = hi;
            *reinterpret_cast<half4v*>(&Ap[1][off]) = lo;
        }
    __syncthreads();
    PH(2);
    l2_prefetch_late(a.pf, pfr);

    // ---- out = Wo att + bo + x ----
    // the epilogue's bias and residual rows of tile t + 1 are requested before tile t's MFMAs (round 4: loaded where they were
    // used, they were a load -> vmcnt(0) -> use round trip per tile after the first)
    float4 eb[2], ex[2][NT];
    eb[0] = eb0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ex[0][nt] = ex0[nt];
    if (a.dbg != 2)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int ct = w * TPW + t;
        if (t + WOR - 1 < TPW) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wo[(t + WOR - 1) % WOR][k][pl] = Wo4[(((size_t)(ct + WOR - 1) * 4 + k) * 2 + pl) * 64 + lane];
        }
        if (t + 1 < TPW) {
            const int cn = (ct + 1) * 16 + lq * 4;
            eb[(t + 1) & 1] = *reinterpret_cast<const float4*>(a.bo + cn);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + lr, sn = n / slot, pn = n - sn * slot;
                const size_t rowc = (n < nend && pn < L) ? row0 + sn * L + pn : row0;
                ex[(t + 1) & 1][nt] = *reinterpret_cast<const float4*>(a.x + rowc * a.ldx + cn);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 zM[NT], zL[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { zM[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; zL[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half8 wh = __builtin_bit_cast(half8, wo[t % WOR][k][0]), wl = __builtin_bit_cast(half8, wo[t % WOR][k][1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (nt * 16 + lr) * APB + k * 64 + lq * 16;
                const half8 ah = *reinterpret_cast<const half8*>(&Ap[0][off]);
                const half8 al = *reinterpret_cast<const half8*>(&Ap[1][off]);
                zM[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, zM[nt], 0, 0, 0);
                zL[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, zL[nt], 0, 0, 0);
                zL[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, zL[nt], 0, 0, 0);
            }
        }
        const int c = ct * 16 + lq * 4;
        const float4 b = eb[t & 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 z = zM[nt] + zL[nt] * H3_INV;
            const int n = nt * 16 + lr, sn = n / slot, pn = n - sn * slot;
            if (n < nend && pn < L) {
                const size_t row = row0 + sn * L + pn;
                const float4 xv = ex[t & 1][nt];
                float4 o;
                o.x = z[0] + b.x + xv.x; o.y = z[1] + b.y + xv.y; o.z = z[2] + b.z + xv.z; o.w = z[3] + b.w + xv.w;
                st_out4(a.out, row * a.ldo + c, o, a.pf.wt);
            }
        }
    }
    PH(3);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

// ---------------------------------------------------------------------------------------------
// level0_down_kernel: the whole finest level of the down path -- ResidualTemporalBlock, ResidualTemporalBlock,
// Residual(PreNorm(LinearAttentionTemporal)), Downsample1d (model/diffusion_1d.py:483-511, :272-291, :92-98; forward
// :611-619) -- in ONE launch, one sample per workgroup.  GroupNorm, LayerNorm and the attention are per-sample, so a
// workgroup never needs another one's data: six dependent launches (~58 us of critical path at any batch size) become
// one.  Activations stay in LDS (as split-fp16 planes with a zero halo of two positions, so a convolution tap is a
// row offset) and in registers; every product is a split-fp16 MFMA in the orientation of attn1d_site_h3_kernel
// (A = weight fragments, B = activations: accumulator rows = channels, cols = positions), each wave owning 16 of the
// 64 channels, which makes a GroupNorm group (8 channels) two lane-groups of one wave.  C = 64 (dim), L <= 32,
// input channels F <= 32.  The three intermediate tensors are also written out (the tap API and the tests read them).
struct Level0Args {
    const float* x; int F;                 // [Bp, L, F]
    float* h1; float* h2; float* skip; float* down;      // [Bp, L, 64] x 3, [Bp, L/2, 64]
    const float* Wc[4]; const float* bc[4]; const float* gam[4]; const float* bet[4];      // the four k=5 convolutions
    const float* Wr; const float* br;      // residual_conv of the first block (F -> 64, one tap)
    const float* tb0; const float* tb1; int tb_ld;      // time-bias table bases (row t)
    const float* ln_g; const float* Wqkv; const float* Wo; const float* bo;
    const float* Wd; const float* bd;      // Downsample1d (k = 3, stride 2, pad 1)
    const int* t_ptr; int t_imm;
    int L;
    Pf pf; PhaseBuf ph;                                 // L2 warm-up for the next launch
    // gB > 0: x is the sampler's STATE [gB, gLtot, F] and row r of this launch is window r / gB of design r % gB, positions
    // (r / gB) * gcs .. + L - 1 (the time composition of two-body states: compose_gather_kernel's copy, read in place)
    int gB, gcs, gLtot;
};

// one 16-channel x (NT*16)-position tile of a k-tap convolution: A = this wave's weight fragments [tap][k32][plane],
// B = activation planes at row (position * stride + tap + row0)
// lvl_wload() requests a layer's fragments; lvl_conv() multiplies with them.  A workgroup of these kernels is alone on its CU:
// nothing hides an L2 round trip (0.5 - 0.8 us) unless the NEXT layer's fragments are requested before the current layer's
// epilogue (GroupNorm, Mish, fp16 split, barrier) -- level0_down_kernel issued them after the barrier until round 3.
template <int N>
__device__ __forceinline__ void lvl_wload(const float4* __restrict__ Wt, int lane, float4 (&wr)[N][2]) {
#pragma unroll
    for (int i = 0; i < N; ++i) { wr[i][0] = Wt[(i * 2 + 0) * 64 + lane]; wr[i][1] = Wt[(i * 2 + 1) * 64 + lane]; }
}
template <int NT, int TAPS, int KS, int PITCHB>
__device__ __forceinline__ void lvl_conv(const float4 (&wr)[TAPS * KS][2], const unsigned char* Xh, const unsigned char* Xl,
                                         int stride, int row0, int maxrow, int lane, f32x4 (&out)[NT]) {
    const int lr = lane & 15, lq = lane >> 4;
    f32x4 M[NT], Lo[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { M[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const half8 wh = __builtin_bit_cast(half8, wr[tap * KS + k][0]), wl = __builtin_bit_cast(half8, wr[tap * KS + k][1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = min((nt * 16 + lr) * stride + tap + row0, maxrow) * PITCHB + k * 64 + lq * 16;
                const half8 xh = *reinterpret_cast<const half8*>(Xh + off);
                const half8 xl = *reinterpret_cast<const half8*>(Xl + off);
                M[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, M[nt], 0, 0, 0);
                Lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, Lo[nt], 0, 0, 0);
                Lo[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, Lo[nt], 0, 0, 0);
            }
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) out[nt] = M[nt] + Lo[nt] * H3_INV;
}

// + bias ; GroupNorm over (8 channels x L positions) of this wave's two groups (rows 4*lq + i: lq 0,1 | lq 2,3) ; Mish
template <int NT>
__device__ __forceinline__ void lvl_gn_mish(f32x4 (&v)[NT], const float4 bias, const float4 gam, const float4 bet, int L, int lane) {
    const int lr = lane & 15;
    const float bb[4] = {bias.x, bias.y, bias.z, bias.w}, gg[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
    float s1 = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[nt][i] += bb[i]; if (nt * 16 + lr < L) s1 += v[nt][i]; }
    s1 = xsum16(row16_sum(s1));
    const float inv_n = 1.0f / (8.0f * (float)L);
    const float mean = s1 * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[nt][i] - mean; if (nt * 16 + lr < L) s2 += d * d; }
    s2 = xsum16(row16_sum(s2));
    const float rstd = 1.0f / sqrtf(s2 * inv_n + 1e-5f);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[nt][i] = mish_f((v[nt][i] - mean) * rstd * gg[i] + be[i]);
}

// accumulator tile (rows = channels c0 + 4*lq + i, cols = positions) -> split-fp16 planes at row (position + row0);
// positions >= L are written as zeros (they are the right halo of the valid ones)
template <int NT, int PITCHB>
__device__ __forceinline__ void lvl_to_planes(const f32x4 (&v)[NT], unsigned char* Ph, unsigned char* Pl, int c0, int row0, int L, int lane) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + lr;
        half4v hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = (n < L) ? v[nt][i] : 0.f;
            hi[i] = (_Float16)f; lo[i] = (_Float16)((f - (float)hi[i]) * H3_SCALE);
        }
        const int off = (n + row0) * PITCHB + 2 * (c0 + lq * 4);
        *reinterpret_cast<half4v*>(Ph + off) = hi;
        *reinterpret_cast<half4v*>(Pl + off) = lo;
    }
}

template <int NT>
__device__ __forceinline__ void lvl_store(const f32x4 (&v)[NT], float* dst, int c0, int n_valid, int lane, int wt = 0) {      // dst: [positions, 64] of this sample
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + lr;
        if (n < n_valid) st_out4(dst, (size_t)n * 64 + c0 + lq * 4, make_float4(v[nt][0], v[nt][1], v[nt][2], v[nt][3]), wt);
    }
}

// MINB = 2 (more rows than CUs: configs 3 / 4): registers capped at 256 so that two workgroups share a CU (52 KB of LDS each)
template <int NT, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void level0_down_kernel(const Level0Args a) {
    PH_DECL;
    PH(0);        // phase clocks (profiling builds, kernels.h PhaseBuf): mark k follows the k-th workgroup barrier, the last one the final stores
    constexpr int C = 64, NP = NT * 16, ROWS = NP + 4;               // two halo positions on each side
    constexpr int XPB = 2 * 32 + 16, PPB = 2 * C + 16, APB = 2 * 128 + 16, HP = C + 4;
    __shared__ __attribute__((aligned(16))) unsigned char X0[2][ROWS * XPB];          // input, F padded to 32 channels
    __shared__ __attribute__((aligned(16))) unsigned char P[2][2][ROWS * PPB];        // ping-pong activation planes
    __shared__ __attribute__((aligned(16))) unsigned char Ap[2][NP * APB];            // attention output planes
    __shared__ __attribute__((aligned(16))) float H[NP * HP];                          // h2 in fp32 for the LayerNorm
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int L = a.L, b = blockIdx.x;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const int c0 = w * 16;                                               // this wave's channels
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);
    const int cl = c0 + lq * 4;                                          // this lane's four channels
    // ---- every request of the kernel's head in ONE round trip (round 4): the input row, the parameter vectors and the first
    // layer's weight fragments are requested before anything waits (the row was converted and written to LDS right after its
    // load: a full round trip from memory, and only then were the weights requested -- a second one) ----
    const int xr = tid >> 3, xc4 = tid & 7;                              // 32 rows x 8 float4
    const bool xok = xr < L && 4 * xc4 < a.F;
    const size_t xrow0 = a.gB ? (size_t)(b % a.gB) * a.gLtot + (size_t)(b / a.gB) * a.gcs : (size_t)b * L;
    const float4 xv = *reinterpret_cast<const float4*>(a.x + (xrow0 + (xok ? xr : 0)) * a.F + (xok ? 4 * xc4 : 0));
    const float4* Wc0 = reinterpret_cast<const float4*>(a.Wc[0]) + (size_t)w * (5 * 1 * 2 * 64);
    const float4* Wc1 = reinterpret_cast<const float4*>(a.Wc[1]) + (size_t)w * (5 * 2 * 2 * 64);
    const float4* Wc2 = reinterpret_cast<const float4*>(a.Wc[2]) + (size_t)w * (5 * 2 * 2 * 64);
    const float4* Wc3 = reinterpret_cast<const float4*>(a.Wc[3]) + (size_t)w * (5 * 2 * 2 * 64);
    const float4* Wr4 = reinterpret_cast<const float4*>(a.Wr) + (size_t)w * (1 * 1 * 2 * 64);
    const float4* Wd4 = reinterpret_cast<const float4*>(a.Wd) + (size_t)w * (3 * 2 * 2 * 64);
    float4 w5a[5][2], w1[1][2], w10[10][2];
    lvl_wload<5>(Wc0, lane, w5a); lvl_wload<1>(Wr4, lane, w1);
    auto ld4 = [&](const float* p) { return *reinterpret_cast<const float4*>(p + cl); };
    const float4 tb0 = ld4(a.tb0 + (size_t)t_now * a.tb_ld), tb1 = ld4(a.tb1 + (size_t)t_now * a.tb_ld);
    // every per-channel vector of the level, once (requested where they are used, each cost an L2 round trip inside an epilogue)
    float4 pbc[4], pga[4], pbe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { pbc[i] = ld4(a.bc[i]); pga[i] = ld4(a.gam[i]); pbe[i] = ld4(a.bet[i]); }
    const float4 pbr = ld4(a.br), pbo = ld4(a.bo), pbd = ld4(a.bd);
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage x (zero halo, zero pad channels / positions); clear the halos of the activation planes ----
    {
        const int r = xr, c4 = xc4;
        const float4 v = xok ? xv : make_float4(0.f, 0.f, 0.f, 0.f);
        half4v hi, lo;
        hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
        lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
        lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
        if (r < NP) {
            *reinterpret_cast<half4v*>(&X0[0][(r + 2) * XPB + 8 * c4]) = hi;
            *reinterpret_cast<half4v*>(&X0[1][(r + 2) * XPB + 8 * c4]) = lo;
        }
        // halo rows 0, 1, NP + 2, NP + 3 of every plane
        for (int i = tid; i < 4 * (PPB / 4); i += 256) {
            const int hr = i / (PPB / 4), cw = i % (PPB / 4);
            const int row = hr < 2 ? hr : NP + hr;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) reinterpret_cast<float*>(&P[q][pl][row * PPB])[cw] = 0.f;
            if (cw < XPB / 4) { reinterpret_cast<float*>(&X0[0][row * XPB])[cw] = 0.f; reinterpret_cast<float*>(&X0[1][row * XPB])[cw] = 0.f; }
        }
    }
    __syncthreads();
    PH(1);

    // ---- block 0 : y = Mish(GN(conv(x))) + tb0 ; h1 = Mish(GN(conv(y))) + (Wr x + br) ----
    f32x4 v[NT], r1[NT];
    lvl_conv<NT, 5, 1, XPB>(w5a, X0[0], X0[1], 1, 0, ROWS - 1, lane, v);
    lvl_conv<NT, 1, 1, XPB>(w1, X0[0], X0[1], 1, 2, ROWS - 1, lane, r1);
    lvl_wload<10>(Wc1, lane, w10);
    lvl_gn_mish<NT>(v, pbc[0], pga[0], pbe[0], L, lane);
    {
        const float4 br = pbr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            v[nt][0] += tb0.x; v[nt][1] += tb0.y; v[nt][2] += tb0.z; v[nt][3] += tb0.w;
            r1[nt][0] += br.x; r1[nt][1] += br.y; r1[nt][2] += br.z; r1[nt][3] += br.w;
        }
    }
    lvl_to_planes<NT, PPB>(v, P[0][0], P[0][1], c0, 2, L, lane);
    __syncthreads();
    PH(2);
    lvl_conv<NT, 5, 2, PPB>(w10, P[0][0], P[0][1], 1, 0, ROWS - 1, lane, v);
    lvl_wload<10>(Wc2, lane, w10);
    lvl_gn_mish<NT>(v, pbc[1], pga[1], pbe[1], L, lane);
    f32x4 h1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) h1[nt] = v[nt] + r1[nt];
    if (a.h1) lvl_store<NT>(h1, a.h1 + (size_t)b * L * C, c0, L, lane, a.pf.wt);
    lvl_to_planes<NT, PPB>(h1, P[1][0], P[1][1], c0, 2, L, lane);
    __syncthreads();
    PH(3);
    // ---- block 1 (identity residual) ----
    lvl_conv<NT, 5, 2, PPB>(w10, P[1][0], P[1][1], 1, 0, ROWS - 1, lane, v);
    lvl_wload<10>(Wc3, lane, w10);
    lvl_gn_mish<NT>(v, pbc[2], pga[2], pbe[2], L, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { v[nt][0] += tb1.x; v[nt][1] += tb1.y; v[nt][2] += tb1.z; v[nt][3] += tb1.w; }
    lvl_to_planes<NT, PPB>(v, P[0][0], P[0][1], c0, 2, L, lane);
    __syncthreads();
    PH(4);
    lvl_conv<NT, 5, 2, PPB>(w10, P[0][0], P[0][1], 1, 0, ROWS - 1, lane, v);
    // the attention's q | k | v fragments of this wave's head (tiles 2w, 2w+1 of q, k, v), requested two barriers ahead
    half8 wh[2][6], wl[2][6];
    {
        const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int tile = (s >> 1) * 8 + 2 * w + (s & 1);
                wh[k][s] = __builtin_bit_cast(half8, Wq4[(((size_t)tile * 2 + k) * 2 + 0) * 64 + lane]);
                wl[k][s] = __builtin_bit_cast(half8, Wq4[(((size_t)tile * 2 + k) * 2 + 1) * 64 + lane]);
            }
    }
    lvl_gn_mish<NT>(v, pbc[3], pga[3], pbe[3], L, lane);
    f32x4 h2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) h2[nt] = v[nt] + h1[nt];
    if (a.h2) lvl_store<NT>(h2, a.h2 + (size_t)b * L * C, c0, L, lane, a.pf.wt);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<float4*>(&H[(nt * 16 + lr) * HP + cl]) = make_float4(h2[nt][0], h2[nt][1], h2[nt][2], h2[nt][3]);
    __syncthreads();
    PH(5);
    // ---- attention: y = LN(h2) g -> planes P[1] (rows position + 2) ; q, k, v ; core ; out projection + h2 ----
    {
        constexpr int RPP = 16;                                          // 16 lanes per row, 16 rows per pass
        const int lrow = tid >> 4, lcol = tid & 15;
        const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + 4 * lcol);
#pragma unroll
        for (int r = 0; r < NP / RPP; ++r) {
            const int n = r * RPP + lrow;
            const float4 xv = *reinterpret_cast<const float4*>(&H[n * HP + 4 * lcol]);
            const float s1 = row16_sum((xv.x + xv.y) + (xv.z + xv.w));
            const float mean = s1 * (1.0f / C);
            const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
            const float s2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            const bool ok = n < L;
            const float y0 = ok ? d0 * rstd * gv.x : 0.f, y1 = ok ? d1 * rstd * gv.y : 0.f, y2 = ok ? d2 * rstd * gv.z : 0.f, y3 = ok ? d3 * rstd * gv.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
            lo[0] = (_Float16)((y0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((y2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y3 - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&P[1][0][(n + 2) * PPB + 8 * lcol]) = hi;
            *reinterpret_cast<half4v*>(&P[1][1][(n + 2) * PPB + 8 * lcol]) = lo;
        }
    }
    __syncthreads();
    PH(6);
    f32x4 qa[2][NT], ka[NT][2], va[NT][2];
    {
        f32x4 qM[2][NT], qL[2][NT], kM[NT][2], kL[NT][2], vM[NT][2], vL[NT][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                qM[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; qL[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                kM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; kL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                vM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; vL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (nt * 16 + lr + 2) * PPB + k * 64 + lq * 16;
                const half8 yh = *reinterpret_cast<const half8*>(&P[1][0][off]);
                const half8 yl = *reinterpret_cast<const half8*>(&P[1][1][off]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    qM[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k][i], yh, qM[i][nt], 0, 0, 0);
                    qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k][i], yl, qL[i][nt], 0, 0, 0);
                    qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[k][i], yh, qL[i][nt], 0, 0, 0);
                    kM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[k][2 + i], kM[nt][i], 0, 0, 0);
                    kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[k][2 + i], kL[nt][i], 0, 0, 0);
                    kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[k][2 + i], kL[nt][i], 0, 0, 0);
                    vM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[k][4 + i], vM[nt][i], 0, 0, 0);
                    vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[k][4 + i], vL[nt][i], 0, 0, 0);
                    vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[k][4 + i], vL[nt][i], 0, 0, 0);
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                qa[i][j] = qM[i][j] + qL[i][j] * H3_INV;
                ka[j][i] = kM[j][i] + kL[j][i] * H3_INV;
                va[j][i] = vM[j][i] + vL[j][i] * H3_INV;
            }
    }
    float4 wo4[4][2], wd6[6][2];                             // out projection and Downsample1d fragments: in flight during the core
    lvl_wload<4>(reinterpret_cast<const float4*>(a.Wo) + (size_t)w * (4 * 2 * 64), lane, wo4);
    lvl_wload<6>(Wd4, lane, wd6);
    l2_prefetch_late(a.pf, pfr);                         // (behind the kernel's LAST load request: vector loads return in order, nothing younger can queue behind the touches)
    f32x4 att[2][NT];
    attn_site_core<NT>(qa, ka, va, att, 1, NP, NP, L, lq, lr);
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            half4v hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)att[et][nt][i]; lo[i] = (_Float16)((att[et][nt][i] - (float)hi[i]) * H3_SCALE); }
            const int off = (nt * 16 + lr) * APB + 2 * (w * 32 + et * 16 + lq * 4);
            *reinterpret_cast<half4v*>(&Ap[0][off]) = hi;
            *reinterpret_cast<half4v*>(&Ap[1][off]) = lo;
        }
    __syncthreads();
    PH(7);
    f32x4 h3[NT];
    {
        lvl_conv<NT, 1, 4, APB>(wo4, Ap[0], Ap[1], 1, 0, NP - 1, lane, h3);
        const float4 bo = pbo;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            h3[nt][0] += bo.x; h3[nt][1] += bo.y; h3[nt][2] += bo.z; h3[nt][3] += bo.w;
            h3[nt] += h2[nt];
        }
    }
    lvl_store<NT>(h3, a.skip + (size_t)b * L * C, c0, L, lane, a.pf.wt);
    lvl_to_planes<NT, PPB>(h3, P[0][0], P[0][1], c0, 2, L, lane);
    __syncthreads();
    PH(8);
    // ---- Downsample1d: out[n'] = sum_tap W[tap] h3[2 n' + tap - 1] + bd ----
    {
        f32x4 d[1];
        lvl_conv<1, 3, 2, PPB>(wd6, P[0][0], P[0][1], 2, 1, ROWS - 1, lane, d);
        const float4 bd = pbd;
        d[0][0] += bd.x; d[0][1] += bd.y; d[0][2] += bd.z; d[0][3] += bd.w;
        lvl_store<1>(d, a.down + (size_t)b * (L / 2) * C, c0, L / 2, lane, a.pf.wt);
    }
    PH(9);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

// ---------------------------------------------------------------------------------------------
// level1_down_kernel: the second down level (64 -> 128 channels, horizon / 2 <= 16 positions) in one launch.  The
// level's weights are 1.6 MB, so a workgroup takes NT samples, ONE SAMPLE PER 16-POSITION TILE (own rows and halos in
// the planes: a tap never crosses samples, GroupNorm / softmax segments are whole tiles), and streams the fragments
// through a per-tap register ring.  Each wave owns 32 channels = two 16-channel tiles = two GroupNorm groups.
struct Level1Args {
    const float* x;                        // [Bp, L, 64]
    float* h1; float* h2; float* skip; float* down;      // [Bp, L, 128] x 3, [Bp, L/2, 128]
    const float* Wc[4]; const float* bc[4]; const float* gam[4]; const float* bet[4];
    const float* Wr; const float* br;
    const float* tb0; const float* tb1; int tb_ld;
    const float* ln_g; const float* Wqkv; const float* Wo; const float* bo;
    const float* Wd; const float* bd;
    const int* t_ptr; int t_imm;
    int L, Bp;
    int dbg;                               // timing ablation: return after phase dbg (wrong results)
    Pf pf; PhaseBuf ph;                                 // L2 warm-up for the next launch
};

// MT x NT tiles of a k-tap convolution.  The weight fragments stream through a register ring of RING taps that the
// CALLER owns: lvlm_prefetch() issues the first RING - 1 taps (while the previous layer's epilogue runs -- a workgroup of
// these kernels is alone on its CU, nothing else hides a load), lvlm_conv() keeps RING - 1 taps in flight.
// RING is a parameter of the ring's type (round 4): the 256-input-channel layers of the up-path kernels hold 64 registers per tap,
// and with three taps of them (192) next to the residual ring the kernel ran out of ARCHITECTURAL registers -- hipcc then parked
// the ring in accumulator registers and filled it through ONE staging quad, load -> vmcnt(0) -> v_accvgpr_write per pair: eight
// serial L2 round trips per tile (ISA of ups_last_kernel; that phase measured 8.3 us against a 5.1 us streaming floor).  Those
// layers use RING = 2.
constexpr int LVL_RING = 3;
template <int MT, int KSMAX, int RING = LVL_RING>
struct LvlRing { float4 wr[RING][MT][KSMAX][2]; };

template <int MT, int TAPS, int KS, int KSMAX, int RING>
__device__ __forceinline__ void lvlm_load_tap(LvlRing<MT, KSMAX, RING>& rg, const float4* __restrict__ Wt, int tap, int slot, int lane) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) rg.wr[slot][mt][k][pl] = Wt[((((size_t)mt * TAPS + tap) * KS + k) * 2 + pl) * 64 + lane];
}
template <int MT, int TAPS, int KS, int KSMAX, int RING>
__device__ __forceinline__ void lvlm_prefetch(LvlRing<MT, KSMAX, RING>& rg, const float4* __restrict__ Wt, int lane) {
#pragma unroll
    for (int tap = 0; tap < RING - 1; ++tap) if (tap < TAPS) lvlm_load_tap<MT, TAPS, KS, KSMAX>(rg, Wt, tap, tap % RING, lane);
}
// RMODE 1 = ConvTranspose1d(k = 4, stride 2, pad 1): output position n reads input (n + 1 - tap) / 2 when that is a whole
// number >= 0 (rows beyond the input are zero), otherwise the zero halo row 0
template <int MT, int NT, int TAPS, int KS, int PITCHB, int KSMAX, int RMODE = 0, int RING = LVL_RING>
__device__ __forceinline__ void lvlm_conv(LvlRing<MT, KSMAX, RING>& rg, const float4* __restrict__ Wt, const unsigned char* Xh, const unsigned char* Xl,
                                          int tile_rows, int stride, int row0, int maxrow, int lane, f32x4 (&out)[MT][NT]) {
    const int lr = lane & 15, lq = lane >> 4;
    f32x4 M[MT][NT], Lo[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { M[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        if (tap + RING - 1 < TAPS) lvlm_load_tap<MT, TAPS, KS, KSMAX>(rg, Wt, tap + RING - 1, (tap + RING - 1) % RING, lane);
        // all activation fragments of the tap first (one exposed LDS latency per tap instead of one per k-step), then the MFMAs
        half8 xh[KS][NT], xl[KS][NT];
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int row = nt * tile_rows + lr * stride + tap + row0;
                if constexpr (RMODE == 1) { const int q = nt * 16 + lr + 1 - tap; row = (q >= 0 && !(q & 1)) ? (q >> 1) + 2 : 0; }
                const int off = min(row, maxrow) * PITCHB + k * 64 + lq * 16;
                xh[k][nt] = *reinterpret_cast<const half8*>(Xh + off);
                xl[k][nt] = *reinterpret_cast<const half8*>(Xl + off);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const half8 wh = __builtin_bit_cast(half8, rg.wr[tap % RING][mt][k][0]), wl = __builtin_bit_cast(half8, rg.wr[tap % RING][mt][k][1]);
                    M[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[k][nt], M[mt][nt], 0, 0, 0);
                    Lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[k][nt], Lo[mt][nt], 0, 0, 0);
                    Lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[k][nt], Lo[mt][nt], 0, 0, 0);
                }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) out[mt][nt] = M[mt][nt] + Lo[mt][nt] * H3_INV;
}

// + bias ; GroupNorm of one 16-channel tile over the L positions of ONE sample tile (all 64 lanes) ; Mish
__device__ __forceinline__ void lvlm_gn_mish(f32x4& v, const float4 bias, const float4 gam, const float4 bet, int L, int lane) {
    const int lr = lane & 15;
    const float bb[4] = {bias.x, bias.y, bias.z, bias.w}, gg[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
    const bool ok = lr < L;
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] += bb[i]; if (ok) s1 += v[i]; }
    s1 = xsum32(xsum16(row16_sum(s1)));
    const float inv_n = 1.0f / (16.0f * (float)L);
    const float mean = s1 * inv_n;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; if (ok) s2 += d * d; }
    s2 = xsum32(xsum16(row16_sum(s2)));
    const float rstd = 1.0f / sqrtf(s2 * inv_n + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = mish_f((v[i] - mean) * rstd * gg[i] + be[i]);
}

template <int NT, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void level1_down_kernel(const Level1Args a) {
    PH_DECL;
    PH(0);        // phase clocks (profiling builds, kernels.h PhaseBuf): mark k follows the k-th workgroup barrier, the last one the final stores
    constexpr int C = 128, CI = 64, RS = 20, ROWS = NT * RS, NP = NT * 16;
    constexpr int XPB = 2 * CI + 16, PPB = 2 * C + 16, APB = 2 * 128 + 16, HP = C + 4;
    constexpr int RBYTES = (2 * ROWS * XPB > 2 * NP * APB) ? 2 * ROWS * XPB : 2 * NP * APB;
    static_assert(NP * HP * 4 <= RBYTES, "H fits the shared region");
    __shared__ __attribute__((aligned(16))) unsigned char P[2][2][ROWS * PPB];        // ping-pong activation planes
    __shared__ __attribute__((aligned(16))) unsigned char R[RBYTES];                  // x planes, then h2 (fp32), then att planes
    unsigned char* X0h = R; unsigned char* X0l = R + ROWS * XPB;
    float* H = reinterpret_cast<float*>(R);
    unsigned char* Aph = R; unsigned char* Apl = R + NP * APB;
    // per-channel parameter vectors, fetched once: 0-3 conv bias, 4-7 GroupNorm weight, 8-11 GroupNorm bias, 12 / 13 time bias
    // of the two blocks (row t), 14 residual_conv bias, 15 to_out bias, 16 downsample bias
    __shared__ __attribute__((aligned(16))) float PV[17][C];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int L = a.L;
    const int s0 = blockIdx.x * NT, s_here = min(NT, a.Bp - s0);
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const int t0 = 2 * w;                                                // this wave's first 16-channel tile
    auto wbase = [&](const float* W, int taps, int ks) { return reinterpret_cast<const float4*>(W) + (size_t)t0 * taps * ks * 2 * 64; };
    LvlRing<2, 4> ring;                                                  // convolution fragments
    LvlRing<2, 2> ring_r;                                                // residual_conv fragments (one tap)
    lvlm_prefetch<2, 5, 2, 4>(ring, wbase(a.Wc[0], 5, 2), lane);
    lvlm_prefetch<2, 1, 2, 2>(ring_r, wbase(a.Wr, 1, 2), lane);
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);
    // (round 4) the input rows are requested BEFORE the parameter vectors go to LDS: `PV[i][tid] = src[i][tid]` waits for its loads,
    // and rows requested after that wait were a second serial round trip at the head of the launch
    float4 xin[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        xin[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nt < s_here && (tid >> 4) < L) xin[nt] = *reinterpret_cast<const float4*>(a.x + ((size_t)(s0 + nt) * L + (tid >> 4)) * CI + 4 * (tid & 15));
    }
    float pvr[17];                                                       // requested now, written to LDS after the zero fill
    if (tid < C) {
        const float* src[17] = {a.bc[0], a.bc[1], a.bc[2], a.bc[3], a.gam[0], a.gam[1], a.gam[2], a.gam[3], a.bet[0], a.bet[1], a.bet[2], a.bet[3],
                                a.tb0 + (size_t)t_now * a.tb_ld, a.tb1 + (size_t)t_now * a.tb_ld, a.br, a.bo, a.bd};
#pragma unroll
        for (int i = 0; i < 17; ++i) pvr[i] = src[i][tid];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage x: sample tile nt at rows nt*RS + 2 + position; everything else zero.  The rows are requested BEFORE the zero
    // fill and its barrier (after them, their round trip opened every launch) ----
    for (int i = tid; i < 2 * ROWS * XPB / 16; i += 256) reinterpret_cast<float4*>(R)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 4 * ROWS * PPB / 16; i += 256) reinterpret_cast<float4*>(&P[0][0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < C) {
#pragma unroll
        for (int i = 0; i < 17; ++i) PV[i][tid] = pvr[i];
    }
    __syncthreads();
    PH(1);
    {
        const int c4 = tid & 15;                                         // 16 float4 per row
#pragma unroll
        for (int pass = 0; pass < NT; ++pass) {
            const int nt = pass, p = tid >> 4;                           // 16 positions per pass = one sample tile
            if (nt < s_here && p < L) {
                const float4 v = xin[nt];
                half4v hi, lo;
                hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
                lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
                lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
                *reinterpret_cast<half4v*>(X0h + (nt * RS + 2 + p) * XPB + 8 * c4) = hi;
                *reinterpret_cast<half4v*>(X0l + (nt * RS + 2 + p) * XPB + 8 * c4) = lo;
            }
        }
    }
    auto cl = [&](int mt) { return (t0 + mt) * 16 + lq * 4; };           // the lane's four channels of tile mt
    auto ld4 = [&](int vec, int mt) { return *reinterpret_cast<const float4*>(&PV[vec][cl(mt)]); };
    // accumulator tile -> planes (rows nt*RS + 2 + position), zero beyond L / beyond the samples of this workgroup
    auto to_planes = [&](const f32x4 (&v)[2][NT], unsigned char* Ph, unsigned char* Pl) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool ok = lr < L && nt < s_here;
                half4v hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float f = ok ? v[mt][nt][i] : 0.f;
                    hi[i] = (_Float16)f; lo[i] = (_Float16)((f - (float)hi[i]) * H3_SCALE);
                }
                const int off = (nt * RS + 2 + lr) * PPB + 2 * cl(mt);
                *reinterpret_cast<half4v*>(Ph + off) = hi;
                *reinterpret_cast<half4v*>(Pl + off) = lo;
            }
    };
    auto store = [&](const f32x4 (&v)[2][NT], float* dst, int Lo_) {    // dst [Bp, Lo_, 128]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nt < s_here && lr < Lo_)
                    st_out4(dst, ((size_t)(s0 + nt) * Lo_ + lr) * C + cl(mt), make_float4(v[mt][nt][0], v[mt][nt][1], v[mt][nt][2], v[mt][nt][3]), a.pf.wt);
    };
    auto gn_all = [&](f32x4 (&v)[2][NT], int ci) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float4 b = ld4(ci, mt), g = ld4(4 + ci, mt), be = ld4(8 + ci, mt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) lvlm_gn_mish(v[mt][nt], b, g, be, L, lane);
        }
    };
    auto add4 = [&](f32x4 (&v)[2][NT], int vec) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float4 t = ld4(vec, mt);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { v[mt][nt][0] += t.x; v[mt][nt][1] += t.y; v[mt][nt][2] += t.z; v[mt][nt][3] += t.w; }
        }
    };
    __syncthreads();
    PH(2);
    if (a.dbg == 1) return;

    // ---- block 0 ----
    f32x4 v[2][NT], r1[2][NT], h1[2][NT], h2[2][NT];
    lvlm_conv<2, NT, 5, 2, XPB, 4>(ring, wbase(a.Wc[0], 5, 2), X0h, X0l, RS, 1, 0, ROWS - 1, lane, v);
    lvlm_prefetch<2, 5, 4, 4>(ring, wbase(a.Wc[1], 5, 4), lane);
    lvlm_conv<2, NT, 1, 2, XPB, 2>(ring_r, wbase(a.Wr, 1, 2), X0h, X0l, RS, 1, 2, ROWS - 1, lane, r1);
    gn_all(v, 0);
    add4(v, 12);
    add4(r1, 14);
    to_planes(v, P[0][0], P[0][1]);
    __syncthreads();
    PH(3);
    lvlm_conv<2, NT, 5, 4, PPB, 4>(ring, wbase(a.Wc[1], 5, 4), P[0][0], P[0][1], RS, 1, 0, ROWS - 1, lane, v);
    lvlm_prefetch<2, 5, 4, 4>(ring, wbase(a.Wc[2], 5, 4), lane);
    gn_all(v, 1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) h1[mt][nt] = v[mt][nt] + r1[mt][nt];
    if (a.h1) store(h1, a.h1, L);
    to_planes(h1, P[1][0], P[1][1]);
    __syncthreads();
    PH(4);
    if (a.dbg == 2) return;
    // ---- block 1 ----
    lvlm_conv<2, NT, 5, 4, PPB, 4>(ring, wbase(a.Wc[2], 5, 4), P[1][0], P[1][1], RS, 1, 0, ROWS - 1, lane, v);
    lvlm_prefetch<2, 5, 4, 4>(ring, wbase(a.Wc[3], 5, 4), lane);
    gn_all(v, 2);
    add4(v, 13);
    to_planes(v, P[0][0], P[0][1]);
    __syncthreads();
    PH(5);
    lvlm_conv<2, NT, 5, 4, PPB, 4>(ring, wbase(a.Wc[3], 5, 4), P[0][0], P[0][1], RS, 1, 0, ROWS - 1, lane, v);
    lvlm_prefetch<2, 1, 4, 4>(ring, wbase(a.Wo, 1, 4), lane);          // to_out fragments: in flight through the attention
    gn_all(v, 3);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) h2[mt][nt] = v[mt][nt] + h1[mt][nt];
    if (a.h2) store(h2, a.h2, L);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<float4*>(&H[(nt * 16 + lr) * HP + cl(mt)]) = make_float4(h2[mt][nt][0], h2[mt][nt][1], h2[mt][nt][2], h2[mt][nt][3]);
    __syncthreads();
    PH(6);
    if (a.dbg == 3) return;
    // ---- attention: LayerNorm -> planes P[1] ----
    {
        const int lrow = tid >> 5, lcol = tid & 31;                      // 32 lanes per row, 8 rows per pass
        const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + 4 * lcol);
#pragma unroll
        for (int r = 0; r < NP / 8; ++r) {
            const int n = r * 8 + lrow, nt = n >> 4, p = n & 15;
            const float4 xv = *reinterpret_cast<const float4*>(&H[n * HP + 4 * lcol]);
            const float s1 = xsum16(row16_sum((xv.x + xv.y) + (xv.z + xv.w)));
            const float mean = s1 * (1.0f / C);
            const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
            const float s2 = xsum16(row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)));
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            const bool ok = p < L && nt < s_here;
            const float y0 = ok ? d0 * rstd * gv.x : 0.f, y1 = ok ? d1 * rstd * gv.y : 0.f, y2 = ok ? d2 * rstd * gv.z : 0.f, y3 = ok ? d3 * rstd * gv.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
            lo[0] = (_Float16)((y0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((y2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y3 - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&P[1][0][(nt * RS + 2 + p) * PPB + 8 * lcol]) = hi;
            *reinterpret_cast<half4v*>(&P[1][1][(nt * RS + 2 + p) * PPB + 8 * lcol]) = lo;
        }
    }
    __syncthreads();
    PH(7);
    if (a.dbg == 4) return;
    // ---- q, k, v of head w (tiles 2w, 2w+1 | 8+2w.. | 16+2w..), K = 128: ring over the four k32 steps ----
    f32x4 qa[2][NT], ka[NT][2], va[NT][2];
    {
        const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
        f32x4 qM[2][NT], qL[2][NT], kM[NT][2], kL[NT][2], vM[NT][2], vL[NT][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                qM[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; qL[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                kM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; kL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                vM[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; vL[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        float4 wq[2][6][2];
        auto load_k = [&](int k, int slot) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int tile = (s >> 1) * 8 + 2 * w + (s & 1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wq[slot][s][pl] = Wq4[(((size_t)tile * 4 + k) * 2 + pl) * 64 + lane];
            }
        };
        load_k(0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k + 1 < 4) load_k(k + 1, (k + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            half8 wh[6], wl[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) { wh[s] = __builtin_bit_cast(half8, wq[k & 1][s][0]); wl[s] = __builtin_bit_cast(half8, wq[k & 1][s][1]); }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = (nt * RS + 2 + lr) * PPB + k * 64 + lq * 16;
                const half8 yh = *reinterpret_cast<const half8*>(&P[1][0][off]);
                const half8 yl = *reinterpret_cast<const half8*>(&P[1][1][off]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    qM[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yh, qM[i][nt], 0, 0, 0);
                    qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], yl, qL[i][nt], 0, 0, 0);
                    qL[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], yh, qL[i][nt], 0, 0, 0);
                    kM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[2 + i], kM[nt][i], 0, 0, 0);
                    kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[2 + i], kL[nt][i], 0, 0, 0);
                    kL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[2 + i], kL[nt][i], 0, 0, 0);
                    vM[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[4 + i], vM[nt][i], 0, 0, 0);
                    vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[4 + i], vL[nt][i], 0, 0, 0);
                    vL[nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[4 + i], vL[nt][i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                qa[i][j] = qM[i][j] + qL[i][j] * H3_INV;
                ka[j][i] = kM[j][i] + kL[j][i] * H3_INV;
                va[j][i] = vM[j][i] + vL[j][i] * H3_INV;
            }
    }
    f32x4 att[2][NT];
    attn_site_core<NT>(qa, ka, va, att, s_here, s_here * 16, 16, L, lq, lr);
    __syncthreads();                                          // every wave is done with H (the att planes alias it)
    PH(8);
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            half4v hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)att[et][nt][i]; lo[i] = (_Float16)((att[et][nt][i] - (float)hi[i]) * H3_SCALE); }
            const int off = (nt * 16 + lr) * APB + 2 * (w * 32 + et * 16 + lq * 4);
            *reinterpret_cast<half4v*>(Aph + off) = hi;
            *reinterpret_cast<half4v*>(Apl + off) = lo;
        }
    __syncthreads();
    PH(9);
    if (a.dbg == 5) return;
    f32x4 h3[2][NT];
    lvlm_conv<2, NT, 1, 4, APB, 4>(ring, wbase(a.Wo, 1, 4), Aph, Apl, 16, 1, 0, NP - 1, lane, h3);
    lvlm_prefetch<2, 3, 4, 4>(ring, wbase(a.Wd, 3, 4), lane);
    l2_prefetch_late(a.pf, pfr);
    add4(h3, 15);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) h3[mt][nt] += h2[mt][nt];
    store(h3, a.skip, L);
    to_planes(h3, P[0][0], P[0][1]);
    __syncthreads();
    PH(10);
    if (a.dbg == 6) return;
    // ---- Downsample1d (k = 3, stride 2, pad 1): rows nt*RS + 2*n' + tap + 1 ----
    {
        f32x4 d[2][NT];
        lvlm_conv<2, NT, 3, 4, PPB, 4>(ring, wbase(a.Wd, 3, 4), P[0][0], P[0][1], RS, 2, 1, ROWS - 1, lane, d);
        add4(d, 16);
        store(d, a.down, L / 2);
    }
    PH(11);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

// ---------------------------------------------------------------------------------------------
// ups_last_kernel: the finest up level and the output head in one launch, one sample per workgroup:
//   cat(x, skip) [L, 256] -> ResidualTemporalBlock(256 -> 128) -> ResidualTemporalBlock(128 -> 64) -> attention site(64)
//   -> Upsample1d (ConvTranspose1d k4 s2 p1, L -> 2L) -> final Conv1dBlock(64 -> 64, k5) -> Conv1d(64 -> F, 1)
// (model/diffusion_1d.py:576-583, :635-646, :100-106, :605-608).  Same machinery as the down-level kernels; 8 launches
// become one.  In the 128-channel block a wave owns two 16-channel tiles (= two GroupNorm groups), afterwards one.
// Composition / DDPM update arguments (the kernels are further down; defined here because ups_last_kernel can run the
// update of a plain single-model step in its own epilogue: UpsLastArgs::upd)
struct ComposeArgs {
    int mode, W, cs, T, nb, cond_steps, objective, clip;
    float uncond_coef;
    int64_t B;
    int Ltot;            // rows of the state x per sample (excludes cond rows)
    int F;               // 4 * nb
    const float* x;      // [B, Ltot, F]
    const float* cond;   // [B, cond_steps, F] or null
    float* pair_in;      // [(kk*P+p)*B + b, T, 8]
    float* single_in;    // MULTIBODY: [i*B + b, T, 4]
    const float* pair_eps;
    const float* single_eps;
    // schedule tables (device, [timesteps])
    const float* sqrt_recip; const float* sqrt_recipm1; const float* sqrt_ac; const float* sqrt_1mac;
    const float* coef1; const float* coef2; const float* logvar;
    const int* t_ptr; int t_imm;
    // outputs
    float* mean_out; float* x0_out; float* eps_out;   // predict
    float* x_out;                                      // step: x_{t-1} (may alias x)
    const float* noise; int64_t noise_t_stride;        // explicit noise (+ t * stride), or null
    uint64_t seed; int64_t sample_off; int add_noise;
    const unsigned long long* dyn;      // sample loops: (seed, sample_off) in device memory, so that a captured step is reusable across calls
    const float* inp_cond; int inp_steps; const float* inp_noise; int64_t inp_noise_t_stride;
    int* t_dec; unsigned* done;     // unused by the kernel: the host launches step_counter_kernel after the update
    // DDIM (ddim_sample :1724-1804): per-step (sqrt(alpha_next), c, sigma, -) and time_next tables indexed by the
    // device step index; noise tapes are then indexed by the step index instead of t
    const float* ddim_tab; const int* ddim_tnext; int* step_idx;
    int* sidx_next;                     // ping-pong DDIM loop: the other slot's step index (written with t_next by the step's update)
    // built-in design objective (the paper's point objective, inference/inverse_design_diffusion_1d.py:211-229) with
    // "standard" / "standard-alpha" (-recurrence-N) guidance: pred = mean - [eta_t] * grad_x objective(x)
    int dz_mode;                 // 0 off, 1 "L2", 2 "L2square"
    int dz_alpha;                // 1: scale the gradient by eta_t = beta_t / sqrt(alphas_cumprod_prev_t) (standard-alpha)
    int dz_last_n; float dz_coef, dz_tc, dz_tx, dz_ty;
    int relax;                   // this launch is a relaxation iteration (:1365-1367): x <- a_t pred + b_t z'
    const float* recur_noise;    // explicit z' of this iteration (+ t * recur_t_stride), or null (counter-based, tag below)
    int64_t recur_t_stride; uint32_t recur_tag;
    const float* iso; int iso_steps;     // initial_state_overwrite [B, iso_steps, F] (:1352-1361) or null
    const float* betas; const float* ac; const float* acp;     // schedule tables for eta_t and the relaxation
    // Ping-pong step state of the plain sample loop: the kernels of this step read t / the exchange epochs from one slot; ONE
    // thread of the step's last kernel (this update) writes the OTHER slot for the step that follows -- nothing in this step
    // reads that slot, so no launch of its own is needed to advance the counter (step_counter_kernel is for the other loops).
    int* t_next;                                         // null: off
    const int* ep_cur0; int* ep_next0; const int* ep_cur1; int* ep_next1;     // exchange epochs of the U-Nets the next step runs (or null)
};
__device__ __forceinline__ void compose_advance(const ComposeArgs& a, int t) {
    if (!a.t_next) return;
    if (a.ddim_tab) { const int sidx = a.step_idx[0]; a.t_next[0] = max(a.ddim_tnext[sidx], 0); a.sidx_next[0] = sidx + 1; }   // (step_counter_kernel's rule)
    else a.t_next[0] = t - 1;
    if (a.ep_next0) a.ep_next0[0] = a.ep_cur0[0] + 1;
    if (a.ep_next1) a.ep_next1[0] = a.ep_cur1[0] + 1;
}
// the update of state element i = (b * Ltot + row) * F + f (compose_update_kernel's body)
__device__ void compose_update_element(const ComposeArgs& a, int64_t i);
__device__ __forceinline__ void counter_normal4(uint64_t seed, uint64_t sample, uint32_t step, uint32_t elem4, float (&z)[4]);
// x_{t-1} of one element of a PLAIN single-model step from the model output o (compose_update_element's mode 0 followed
// by its unguided DDPM tail, operation for operation: the fused and the separate update are bit-identical)
// The five schedule values of timestep t that plain_step_value reads, fetched ONCE (uniform addresses: scalar loads).  Round 4:
// ups_last_kernel evaluated plain_step_value eight times per lane and every evaluation re-read its table entries under the
// objective's branches -- ~30 load -> vmcnt(0) -> use sequences at the very end of the step's LAST kernel (the in-replay phase
// clocks showed 5 us for the waves that run the update: the tail of every reverse step).
struct StepCoefs { float cx, co, k1, k2, sigma; float san, cc, sg; int tn, sidx; };     // (san, cc, sg, tn, sidx: the DDIM step's table row)
typedef const float __attribute__((address_space(4))) cindm_cfloat4;
__device__ __forceinline__ float uniform_float(const float* p) { return *(cindm_cfloat4*)(const void*)p; }
__device__ __forceinline__ StepCoefs plain_step_coefs(const ComposeArgs& a, int t) {
    StepCoefs c;
    if (a.objective == 2) { c.cx = uniform_float(a.sqrt_ac + t); c.co = uniform_float(a.sqrt_1mac + t); }
    else { c.cx = uniform_float(a.sqrt_recip + t); c.co = uniform_float(a.sqrt_recipm1 + t); }
    c.k1 = uniform_float(a.coef1 + t); c.k2 = uniform_float(a.coef2 + t);
    c.sigma = (a.add_noise && t > 0) ? expf(0.5f * uniform_float(a.logvar + t)) : 0.f;
    c.san = c.cc = c.sg = 0.f; c.tn = -1; c.sidx = 0;
    if (a.ddim_tab) {
        c.sidx = uniform_word(a.step_idx);
        c.tn = uniform_word(a.ddim_tnext + c.sidx);
        c.san = uniform_float(a.ddim_tab + 4 * c.sidx); c.cc = uniform_float(a.ddim_tab + 4 * c.sidx + 1); c.sg = uniform_float(a.ddim_tab + 4 * c.sidx + 2);
    }
    return c;
}
// does the fused update of this step draw noise?  (compose_update_element's conditions: DDPM t > 0; DDIM a next time and sigma != 0
// -- or an explicit tape, which is read whenever there is a next time)
__device__ __forceinline__ bool plain_step_draws(const ComposeArgs& a, const StepCoefs& c, int t) {
    if (a.ddim_tab) return c.tn >= 0 && (a.noise != nullptr || c.sg != 0.f);
    return a.add_noise && t > 0;
}
// plain_step_value with the coefficients in hand: operation for operation the same expression
__device__ __forceinline__ float plain_step_value(const ComposeArgs& a, const StepCoefs& c, int t, float xv, float o, float z) {
    float x0 = (a.objective == 1) ? o : __fsub_rn(__fmul_rn(c.cx, xv), __fmul_rn(c.co, o));
    if (a.clip) x0 = clamp_pm1(x0);
    if (a.ddim_tab) {       // compose_update_element's DDIM tail (objective pred_noise: eps = o; the host fuses only that objective)
        if (c.tn < 0) return x0;
        return __fadd_rn(__fadd_rn(__fmul_rn(x0, c.san), __fmul_rn(c.cc, o)), __fmul_rn(c.sg, z));
    }
    float v = __fadd_rn(__fmul_rn(c.k1, x0), __fmul_rn(c.k2, xv));
    if (a.add_noise && t > 0) v += c.sigma * z;
    return v;
}
__device__ __forceinline__ float plain_step_value(const ComposeArgs& a, int t, float xv, float o, float z) {
    float x0;
    if (a.objective == 0) x0 = __fsub_rn(__fmul_rn(a.sqrt_recip[t], xv), __fmul_rn(a.sqrt_recipm1[t], o));
    else if (a.objective == 1) x0 = o;
    else x0 = __fsub_rn(__fmul_rn(a.sqrt_ac[t], xv), __fmul_rn(a.sqrt_1mac[t], o));
    if (a.clip) x0 = clamp_pm1(x0);
    float v = __fadd_rn(__fmul_rn(a.coef1[t], x0), __fmul_rn(a.coef2[t], xv));
    if (a.add_noise && t > 0) v += expf(0.5f * a.logvar[t]) * z;
    return v;
}

struct UpsLastArgs {
    const float* x; const float* skip;                   // [Bp, L, 128] each (torch.cat((x, h.pop()), dim=1))
    float* h1; float* h2; float* h3; float* up; float* ypre; float* eps; int F;      // h1 [Bp, L, 128]; h2, h3 [Bp, L, 64]; up, ypre [Bp, 2L, 64]
    const float* Wc[5]; const float* bc[5]; const float* gam[5]; const float* bet[5];      // four block convolutions + final_conv.0
    const float* Wr0; const float* br0; const float* Wr1; const float* br1;                 // residual_conv of the two blocks
    const float* tb0; const float* tb1; int tb_ld;
    const float* ln_g; const float* Wqkv; const float* Wo; const float* bo;
    const float* Wu; const float* bu;
    const float* Wf; const float* bf;                    // final 1x1, one 16-channel tile (rows >= F are zero)
    const int* t_ptr; int t_imm;
    int L;
    Pf pf; PhaseBuf ph;                                               // L2 warm-up for the next launch (the next step's first kernel)
    int fuse_upd; ComposeArgs upd;                       // plain single-model step: x_{t-1} from this kernel's eps rows, in place
};

// (Round 4 also compiled this kernel and ups_tail128_kernel for two workgroups per CU above 320 rows -- 89 / 60 spilled registers, config 3
// 833 -> 841 / 838 us per step --: removed in round 6; level0_down / level1_down keep their two-workgroup instantiations.)
__global__ __launch_bounds__(256) void ups_last_kernel(const UpsLastArgs a) {
    PH_DECL;
    PH(0);        // phase clocks (profiling builds, kernels.h PhaseBuf): mark k follows the k-th workgroup barrier, the last one the final stores
    constexpr int C = 64, CB = 128, CI = 256, NP1 = 16, NP2 = 32, ROWS1 = NP1 + 4, ROWS2 = NP2 + 4;
    constexpr int XPB = 2 * CI + 16, QPB = 2 * CB + 16, PPB = 2 * C + 16, APB = 2 * 128 + 16, HP = C + 4;
    __shared__ __attribute__((aligned(16))) unsigned char XI[2][ROWS1 * XPB];
    __shared__ __attribute__((aligned(16))) unsigned char Q[2][2][ROWS1 * QPB];         // 128-channel planes of the first block
    __shared__ __attribute__((aligned(16))) unsigned char P[2][2][ROWS2 * PPB];         // 64-channel planes
    __shared__ __attribute__((aligned(16))) unsigned char R[2 * NP1 * APB];            // h2 in fp32 for the LayerNorm, then the att planes
    // parameter vectors: 128-wide 0-2 conv1 (bias, GN weight, GN bias), 3-5 conv2, 6 time bias, 7 residual bias;
    // 64-wide 0-2 conv3, 3-5 conv4, 6-8 final conv, 9 time bias, 10 residual bias, 11 to_out, 12 upsample, 13 final 1x1
    __shared__ __attribute__((aligned(16))) float PVB[8][CB];
    __shared__ __attribute__((aligned(16))) float PV[14][C];
    static_assert(NP1 * HP * 4 <= 2 * NP1 * APB, "H fits the shared region");
    float* H = reinterpret_cast<float*>(R);
    unsigned char* Aph = R; unsigned char* Apl = R + NP1 * APB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int L = a.L, L2 = 2 * a.L, b = blockIdx.x;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    const int c0 = w * 16, cl = c0 + lq * 4;
    auto clb = [&](int mt) { return (2 * w + mt) * 16 + lq * 4; };       // the lane's channels in the 128-channel block
    auto wbase = [&](const float* W, int taps, int ks) { return reinterpret_cast<const float4*>(W) + (size_t)w * taps * ks * 2 * 64; };
    LvlRing<1, 8> ring, ring_r;                                          // weight rings (one 16-channel tile at a time)
    LvlRing<1, 8, 2> ring8;                                              // ... of the 256-input-channel block: two taps deep (see LvlRing)
    auto wtile = [&](const float* W, int tile, int taps, int ks) { return reinterpret_cast<const float4*>(W) + (size_t)tile * taps * ks * 2 * 64; };
    lvlm_prefetch<1, 5, 8, 8>(ring8, wtile(a.Wc[0], 2 * w, 5, 8), lane);
    lvlm_prefetch<1, 1, 8, 8>(ring_r, wtile(a.Wr0, 2 * w, 1, 8), lane);
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);
    // (round 4) the input rows are requested BEFORE the parameter vectors go to LDS: `PV[i][tid] = src[i][tid]` waits for its loads,
    // and rows requested after that wait were a second serial round trip at the head of the launch
    float4 xin[4];                                                       // cat(x, skip) rows, requested before the zero fill and its barrier
    if ((tid >> 4) < L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = tid >> 4, cf = (tid & 15) + 16 * q;            // float4 index over the 256 channels
            const float* src = cf < 32 ? a.x + ((size_t)b * L + p) * 128 + 4 * cf : a.skip + ((size_t)b * L + p) * 128 + 4 * (cf - 32);
            xin[q] = *reinterpret_cast<const float4*>(src);
        }
    }
    // (round 6) the LDS zero fill HERE, while the scalar load of t is on its way: the parameter loads below need t for the time-bias row and
    // the wave stalls in front of them until it arrives (in-order issue) -- behind them the zero fill was 0.5 us in front of the first barrier
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < 2 * ROWS1 * XPB / 16; i += 256) reinterpret_cast<float4*>(&XI[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 4 * ROWS1 * QPB / 16; i += 256) reinterpret_cast<float4*>(&Q[0][0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 4 * ROWS2 * PPB / 16; i += 256) reinterpret_cast<float4*>(&P[0][0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_sched_barrier(0);
    float pvr[14];                                                       // requested now, written to LDS after the noise generation
    if (tid < CB) {
        const float* src[8] = {a.bc[0], a.gam[0], a.bet[0], a.bc[1], a.gam[1], a.bet[1], a.tb0 + (size_t)t_now * a.tb_ld, a.br0};
#pragma unroll
        for (int i = 0; i < 8; ++i) pvr[i] = src[i][tid];
    } else if (tid < CB + C) {
        const int c = tid - CB;
        const float* src[13] = {a.bc[2], a.gam[2], a.bet[2], a.bc[3], a.gam[3], a.bet[3], a.bc[4], a.gam[4], a.bet[4],
                                a.tb1 + (size_t)t_now * a.tb_ld, a.br1, a.bo, a.bu};
#pragma unroll
        for (int i = 0; i < 13; ++i) pvr[i] = src[i][c];
        pvr[13] = a.bf[c < a.F ? c : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
    // Fused update (UpsLastArgs::upd): the noise of this sample's state elements depends on nothing this kernel computes, so it is
    // generated HERE, while the first loads are in flight, by all four waves (slot s = nt * 64 + lane of the final stage's wave 0;
    // wave w takes slots 32 w ..), parked in R -- free until block 1's output goes there -- and picked up by wave 0 after barrier 4.
    // At the kernel's tail (one wave, after the last barrier) Philox + Box-Muller were 2.5 us of a 3 us phase (phase clocks, round 4).
    const int tu_z = a.fuse_upd ? step_scalar(a.upd.t_ptr, a.upd.t_imm) : 0;
    StepCoefs sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, -1, 0};      // the step's schedule values (DDIM: its table row, scalar loads)
    // DDPM: the five schedule values of step t are VECTOR loads of wave 0, consumed where the update is applied.  As scalar loads at this
    // point (through round 6) they sat behind the scalar load of t -- two serial round trips to memory, lines nobody has read for 32 steps --
    // and `sigma = expf(.)` waited for them HERE, in front of the noise generation, the LDS zero fill and the first barrier (a barrier waits
    // lgkmcnt(0): scalar loads included): this phase measured 4.2 - 4.6 us with the update fused against 2.7 - 3.0 us without
    // (3.9 us now; same-box with the zero fill moved up as well: 313.7 -> 311.9 us per step).
    const bool sc_vec = a.fuse_upd && !a.upd.ddim_tab;
    float scv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.fuse_upd && !sc_vec) sc = plain_step_coefs(a.upd, tu_z);
    {   // EVERY wave issues exactly five loads, whatever the launch does (no update fused / DDIM: five reads of the final weights' first word):
        // behind a branch with an unknown number of requests the compiler waits for every OLDER load with a count that covers these too
        const ComposeArgs& u = a.upd;
        const float* tab[5] = {u.objective == 2 ? u.sqrt_ac : u.sqrt_recip, u.objective == 2 ? u.sqrt_1mac : u.sqrt_recipm1, u.coef1, u.coef2, u.logvar};
#pragma unroll
        for (int i = 0; i < 5; ++i)         // (buffer loads: a uniform address would be turned back into a scalar load)
            scv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc_vec ? tab[i] : a.Wf), 0, 0x7ffffff0u, 0x00020000), sc_vec ? (unsigned)tu_z * 4u : 0u, 0, 0));
    }
    // (seed, sample offset) of the sample loops: SCALAR loads, requested here with t.  As `u.dyn ? u.dyn[0] : u.seed` inside the noise
    // block they were vector loads waited for with vmcnt(0) -- i.e. behind the 160 KB of weight fragments, the input rows and the
    // parameter vectors requested above: the noise generation started when ALL of those had arrived (tools/isa_audit.py).
    typedef const unsigned long long __attribute__((address_space(4))) cindm_cu64;
    const uint64_t dseed = (a.fuse_upd && a.upd.dyn) ? (uint64_t)*(cindm_cu64*)(const void*)a.upd.dyn : a.upd.seed;
    const int64_t dsoff = (a.fuse_upd && a.upd.dyn) ? (int64_t)*(cindm_cu64*)(const void*)(a.upd.dyn + 1) : a.upd.sample_off;
    const bool gen_z = a.fuse_upd && (sc_vec ? (a.upd.add_noise && tu_z > 0) : plain_step_draws(a.upd, sc, tu_z));
    if (gen_z && lane < 32) {
        const ComposeArgs& u = a.upd;
        const int s = 32 * w + lane, nt = s >> 6, ln = s & 63, n = nt * 16 + (ln & 15), q4 = (ln >> 4) * 4;
        if (n < L2 && q4 < a.F) {
            float4 z;
            // (explicit tapes are indexed by t in the DDPM loop and by the step index in the DDIM loop)
            if (u.noise) z = *reinterpret_cast<const float4*>(u.noise + (size_t)(u.ddim_tab ? sc.sidx : tu_z) * u.noise_t_stride + ((size_t)b * L2 + n) * a.F + q4);
            else {
                float z4[4];
                counter_normal4(dseed, (uint64_t)(dsoff + b), (uint32_t)tu_z, (uint32_t)((n * a.F + q4) >> 2), z4);
                z = make_float4(z4[0], z4[1], z4[2], z4[3]);
            }
            reinterpret_cast<float4*>(R)[s] = z;
        }
    }
    if (tid < CB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) PVB[i][tid] = pvr[i];
    } else if (tid < CB + C) {
        const int c = tid - CB;
#pragma unroll
        for (int i = 0; i < 13; ++i) PV[i][c] = pvr[i];
        PV[13][c] = c < a.F ? pvr[13] : 0.f;
    }
    __syncthreads();
    PH(1);
    {
        const int p = tid >> 4, c4 = tid & 15;                           // 16 positions x 16 float4, four channel quarters
        if (p < L) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cf = c4 + 16 * q;
                const float4 v = xin[q];
                half4v hi, lo;
                hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
                lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
                lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
                *reinterpret_cast<half4v*>(&XI[0][(p + 2) * XPB + 8 * cf]) = hi;
                *reinterpret_cast<half4v*>(&XI[1][(p + 2) * XPB + 8 * cf]) = lo;
            }
        }
    }
    auto pv4 = [&](int vec) { return *reinterpret_cast<const float4*>(&PV[vec][cl]); };
    auto pvb4 = [&](int vec, int mt) { return *reinterpret_cast<const float4*>(&PVB[vec][clb(mt)]); };
    auto add4 = [&](f32x4& v, const float4 t) { v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; };
    auto q_planes = [&](const f32x4 (&v)[2][1], unsigned char* Ph, unsigned char* Pl) {      // 128-channel tile -> planes, rows position + 2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            half4v hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = lr < L ? v[mt][0][i] : 0.f;
                hi[i] = (_Float16)f; lo[i] = (_Float16)((f - (float)hi[i]) * H3_SCALE);
            }
            const int off = (lr + 2) * QPB + 2 * clb(mt);
            *reinterpret_cast<half4v*>(Ph + off) = hi;
            *reinterpret_cast<half4v*>(Pl + off) = lo;
        }
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    PH(2);

    // ---- block 0 (256 -> 128, residual_conv): the wave's two 16-channel tiles one after the other ----
    f32x4 vb[2][1], rb[2][1], h1b[2][1];
    {
        f32x4 t[1][1];
        lvlm_conv<1, 1, 5, 8, XPB, 8>(ring8, wtile(a.Wc[0], 2 * w, 5, 8), XI[0], XI[1], 0, 1, 0, ROWS1 - 1, lane, t); vb[0][0] = t[0][0];
        lvlm_prefetch<1, 5, 8, 8>(ring8, wtile(a.Wc[0], 2 * w + 1, 5, 8), lane);
        lvlm_conv<1, 1, 1, 8, XPB, 8>(ring_r, wtile(a.Wr0, 2 * w, 1, 8), XI[0], XI[1], 0, 1, 2, ROWS1 - 1, lane, t); rb[0][0] = t[0][0];
        lvlm_prefetch<1, 1, 8, 8>(ring_r, wtile(a.Wr0, 2 * w + 1, 1, 8), lane);
        lvlm_conv<1, 1, 5, 8, XPB, 8>(ring8, wtile(a.Wc[0], 2 * w + 1, 5, 8), XI[0], XI[1], 0, 1, 0, ROWS1 - 1, lane, t); vb[1][0] = t[0][0];
        lvlm_prefetch<1, 5, 4, 8>(ring, wtile(a.Wc[1], 2 * w, 5, 4), lane);
        lvlm_conv<1, 1, 1, 8, XPB, 8>(ring_r, wtile(a.Wr0, 2 * w + 1, 1, 8), XI[0], XI[1], 0, 1, 2, ROWS1 - 1, lane, t); rb[1][0] = t[0][0];
        lvlm_prefetch<1, 1, 4, 8>(ring_r, wtile(a.Wr1, w, 1, 4), lane);                   // second block's residual_conv (128 -> 64)
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        lvlm_gn_mish(vb[mt][0], pvb4(0, mt), pvb4(1, mt), pvb4(2, mt), L, lane);
        add4(vb[mt][0], pvb4(6, mt));
        add4(rb[mt][0], pvb4(7, mt));
    }
    q_planes(vb, Q[0][0], Q[0][1]);
    __syncthreads();
    PH(3);
    {
        f32x4 t[1][1];
        lvlm_conv<1, 1, 5, 4, QPB, 8>(ring, wtile(a.Wc[1], 2 * w, 5, 4), Q[0][0], Q[0][1], 0, 1, 0, ROWS1 - 1, lane, t); vb[0][0] = t[0][0];
        lvlm_prefetch<1, 5, 4, 8>(ring, wtile(a.Wc[1], 2 * w + 1, 5, 4), lane);
        lvlm_conv<1, 1, 5, 4, QPB, 8>(ring, wtile(a.Wc[1], 2 * w + 1, 5, 4), Q[0][0], Q[0][1], 0, 1, 0, ROWS1 - 1, lane, t); vb[1][0] = t[0][0];
        lvlm_prefetch<1, 5, 4, 8>(ring, wtile(a.Wc[2], w, 5, 4), lane);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        lvlm_gn_mish(vb[mt][0], pvb4(3, mt), pvb4(4, mt), pvb4(5, mt), L, lane);
        h1b[mt][0] = vb[mt][0] + rb[mt][0];
        if (a.h1 && lr < L) st_out4(a.h1, ((size_t)b * L + lr) * CB + clb(mt), make_float4(h1b[mt][0][0], h1b[mt][0][1], h1b[mt][0][2], h1b[mt][0][3]), a.pf.wt);
    }
    q_planes(h1b, Q[1][0], Q[1][1]);
    __syncthreads();
    PH(4);
    // ---- block 1 (128 -> 64, residual_conv) ----
    float4 zq[2] = {};                                                   // (wave 0) the fused update's noise, out of R before H goes there
    if (w == 0 && gen_z) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (nt * 16 + lr < L2 && lq * 4 < a.F) zq[nt] = reinterpret_cast<const float4*>(R)[nt * 64 + lane];
    }
    f32x4 v1[1][1], r1[1][1], h2[1];
    lvlm_conv<1, 1, 5, 4, QPB, 8>(ring, wtile(a.Wc[2], w, 5, 4), Q[1][0], Q[1][1], 0, 1, 0, ROWS1 - 1, lane, v1);
    lvlm_prefetch<1, 5, 2, 8>(ring, wbase(a.Wc[3], 5, 2), lane);
    lvlm_conv<1, 1, 1, 4, QPB, 8>(ring_r, wtile(a.Wr1, w, 1, 4), Q[1][0], Q[1][1], 0, 1, 2, ROWS1 - 1, lane, r1);
    lvl_gn_mish<1>(v1[0], pv4(0), pv4(1), pv4(2), L, lane);
    add4(v1[0][0], pv4(9));
    add4(r1[0][0], pv4(10));
    lvl_to_planes<1, PPB>(v1[0], P[0][0], P[0][1], c0, 2, L, lane);
    __syncthreads();
    PH(5);
    lvlm_conv<1, 1, 5, 2, PPB, 8>(ring, wbase(a.Wc[3], 5, 2), P[0][0], P[0][1], 0, 1, 0, ROWS2 - 1, lane, v1);
    lvlm_prefetch<1, 1, 4, 8>(ring, wbase(a.Wo, 1, 4), lane);           // to_out fragments: in flight through the attention
    lvl_gn_mish<1>(v1[0], pv4(3), pv4(4), pv4(5), L, lane);
    h2[0] = v1[0][0] + r1[0][0];
    if (a.h2) lvl_store<1>(h2, a.h2 + (size_t)b * L * C, c0, L, lane, a.pf.wt);
    *reinterpret_cast<float4*>(&H[lr * HP + cl]) = make_float4(h2[0][0], h2[0][1], h2[0][2], h2[0][3]);
    __syncthreads();
    PH(6);
    // ---- attention site (C = 64, one 16-position tile) ----
    {
        const int lrow = tid >> 4, lcol = tid & 15;
        const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + 4 * lcol);
        const int n = lrow;
        const float4 xv = *reinterpret_cast<const float4*>(&H[n * HP + 4 * lcol]);
        const float s1 = row16_sum((xv.x + xv.y) + (xv.z + xv.w));
        const float mean = s1 * (1.0f / C);
        const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
        const float s2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
        const bool ok = n < L;
        const float y0 = ok ? d0 * rstd * gv.x : 0.f, y1 = ok ? d1 * rstd * gv.y : 0.f, y2 = ok ? d2 * rstd * gv.z : 0.f, y3 = ok ? d3 * rstd * gv.w : 0.f;
        half4v hi, lo;
        hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
        lo[0] = (_Float16)((y0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y1 - (float)hi[1]) * H3_SCALE);
        lo[2] = (_Float16)((y2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y3 - (float)hi[3]) * H3_SCALE);
        *reinterpret_cast<half4v*>(&P[1][0][(n + 2) * PPB + 8 * lcol]) = hi;
        *reinterpret_cast<half4v*>(&P[1][1][(n + 2) * PPB + 8 * lcol]) = lo;
    }
    __syncthreads();
    PH(7);
    f32x4 qa[2][1], ka[1][2], va[1][2];
    {
        const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
        f32x4 M[6], Lo[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) { M[s6] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s6] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        half8 wh[2][6], wl[2][6];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const int tile = (s6 >> 1) * 8 + 2 * w + (s6 & 1);
                wh[k][s6] = __builtin_bit_cast(half8, Wq4[(((size_t)tile * 2 + k) * 2 + 0) * 64 + lane]);
                wl[k][s6] = __builtin_bit_cast(half8, Wq4[(((size_t)tile * 2 + k) * 2 + 1) * 64 + lane]);
            }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int off = (lr + 2) * PPB + k * 64 + lq * 16;
            const half8 yh = *reinterpret_cast<const half8*>(&P[1][0][off]);
            const half8 yl = *reinterpret_cast<const half8*>(&P[1][1][off]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                M[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k][i], yh, M[i], 0, 0, 0);
                Lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[k][i], yl, Lo[i], 0, 0, 0);
                Lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[k][i], yh, Lo[i], 0, 0, 0);
#pragma unroll
                for (int kv = 2; kv < 6; kv += 2) {
                    M[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh[k][kv + i], M[kv + i], 0, 0, 0);
                    Lo[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl[k][kv + i], Lo[kv + i], 0, 0, 0);
                    Lo[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh[k][kv + i], Lo[kv + i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            qa[i][0] = M[i] + Lo[i] * H3_INV;
            ka[0][i] = M[2 + i] + Lo[2 + i] * H3_INV;
            va[0][i] = M[4 + i] + Lo[4 + i] * H3_INV;
        }
    }
    f32x4 att[2][1];
    attn_site_core<1>(qa, ka, va, att, 1, NP1, NP1, L, lq, lr);
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        half4v hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)att[et][0][i]; lo[i] = (_Float16)((att[et][0][i] - (float)hi[i]) * H3_SCALE); }
        const int off = lr * APB + 2 * (w * 32 + et * 16 + lq * 4);
        *reinterpret_cast<half4v*>(Aph + off) = hi;
        *reinterpret_cast<half4v*>(Apl + off) = lo;
    }
    __syncthreads();
    PH(8);
    f32x4 h3[1];
    lvlm_conv<1, 1, 1, 4, APB, 8>(ring, wbase(a.Wo, 1, 4), Aph, Apl, 0, 1, 0, NP1 - 1, lane, v1);
    lvlm_prefetch<1, 4, 2, 8>(ring, wbase(a.Wu, 4, 2), lane);
    h3[0] = v1[0][0];
    add4(h3[0], pv4(11));
    h3[0] += h2[0];
    if (a.h3) lvl_store<1>(h3, a.h3 + (size_t)b * L * C, c0, L, lane, a.pf.wt);
    lvl_to_planes<1, PPB>(h3, P[0][0], P[0][1], c0, 2, L, lane);
    __syncthreads();
    PH(9);
    // ---- Upsample1d: ConvTranspose1d(k = 4, stride 2, pad 1), L -> 2L positions (two tiles) ----
    f32x4 u[1][2];
    lvlm_conv<1, 2, 4, 2, PPB, 8, 1>(ring, wbase(a.Wu, 4, 2), P[0][0], P[0][1], 0, 0, 0, ROWS2 - 1, lane, u);
    lvlm_prefetch<1, 5, 2, 8>(ring, wbase(a.Wc[4], 5, 2), lane);
    add4(u[0][0], pv4(12)); add4(u[0][1], pv4(12));
    if (a.up) lvl_store<2>(u[0], a.up + (size_t)b * L2 * C, c0, L2, lane, a.pf.wt);
    lvl_to_planes<2, PPB>(u[0], P[1][0], P[1][1], c0, 2, L2, lane);
    __syncthreads();
    PH(10);
    // ---- final Conv1dBlock(64 -> 64, k5) and Conv1d(64 -> F, 1) ----
    float4 xq[2] = {};                                                   // (wave 0) x_t of the elements it will update: a layer ahead
    if (w == 0 && a.fuse_upd) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 16 + lr;
            if (n < L2 && lq * 4 < a.F) xq[nt] = *reinterpret_cast<const float4*>(a.upd.x + ((size_t)b * L2 + n) * a.F + lq * 4);
        }
    }
    f32x4 y[1][2];
    lvlm_conv<1, 2, 5, 2, PPB, 8>(ring, wbase(a.Wc[4], 5, 2), P[1][0], P[1][1], 16, 1, 0, ROWS2 - 1, lane, y);
    if (w == 0) lvlm_prefetch<1, 1, 2, 8>(ring, reinterpret_cast<const float4*>(a.Wf), lane);
    l2_prefetch_late(a.pf, pfr);
    add4(y[0][0], pv4(6)); add4(y[0][1], pv4(6));
    if (a.ypre) lvl_store<2>(y[0], a.ypre + (size_t)b * L2 * C, c0, L2, lane, a.pf.wt);
    lvl_gn_mish<2>(y[0], zero4, pv4(7), pv4(8), L2, lane);
    lvl_to_planes<2, PPB>(y[0], P[0][0], P[0][1], c0, 2, L2, lane);
    __syncthreads();
    PH(11);
    if (w == 0) {
        // Plain single-model step (UpsLastArgs::upd): the lanes that hold the prediction of 4 state elements also apply the
        // reverse-step update to them, in place -- no compose_update_kernel launch.  Their state values (xq) were requested a
        // layer ago and their noise (zq) was generated at the top of the kernel, so the update adds a few FMAs to this tail.
        const ComposeArgs& u = a.upd;
        const int tu = tu_z;
        if (sc_vec) {                        // plain_step_coefs' expressions on the vector-loaded values
            // (the empty asm pins the values' first use HERE: left alone the compiler evaluates expf(.) right behind the loads, i.e. waits for
            // them -- and for every older request of the wave -- at the top of the kernel)
#pragma unroll
            for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(scv[i]));
            sc.cx = scv[0]; sc.co = scv[1]; sc.k1 = scv[2]; sc.k2 = scv[3];
            sc.sigma = (u.add_noise && tu > 0) ? expf(0.5f * scv[4]) : 0.f;
        }
        f32x4 e[1][2];
        lvlm_conv<1, 2, 1, 2, PPB, 8>(ring, reinterpret_cast<const float4*>(a.Wf), P[0][0], P[0][1], 16, 1, 2, ROWS2 - 1, lane, e);
        const float4 bf = *reinterpret_cast<const float4*>(&PV[13][lq * 4]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 16 + lr;
            if (n < L2 && lq * 4 < a.F) {
                const size_t i0 = ((size_t)b * L2 + n) * a.F + lq * 4;
                const float4 o = make_float4(e[0][nt][0] + bf.x, e[0][nt][1] + bf.y, e[0][nt][2] + bf.z, e[0][nt][3] + bf.w);
                st_out4(a.eps, i0, o, a.pf.wt);
                if (a.fuse_upd)
                    st_out4(u.x_out, i0,
                            make_float4(plain_step_value(u, sc, tu, xq[nt].x, o.x, zq[nt].x), plain_step_value(u, sc, tu, xq[nt].y, o.y, zq[nt].y),
                                        plain_step_value(u, sc, tu, xq[nt].z, o.z, zq[nt].z), plain_step_value(u, sc, tu, xq[nt].w, o.w, zq[nt].w)), a.pf.wt);
            }
        }
        if (a.fuse_upd && blockIdx.x == 0 && lane == 0) compose_advance(u, tu);
    }
    PH(12);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

// ---------------------------------------------------------------------------------------------
// ups_tail128_kernel: the second half of the second-finest up level in one launch, one sample per workgroup:
//   ResidualTemporalBlock(256 -> 128) -> attention site(128) -> Upsample1d(128) (ConvTranspose1d k4 s2 p1, L -> 2L <= 16)
// (the level's first block, 512 -> 256, keeps its per-layer launches: its weights are 3 MB).  Pieces of
// level1_down_kernel (128-channel attention) and ups_last_kernel (256-channel input streamed tile by tile, transposed
// convolution).
struct UpsTailArgs {
    const float* x;                        // [Bp, L, 256]
    float* h2; float* h3; float* up;       // [Bp, L, 128] x 2, [Bp, 2L, 128]
    const float* Wc[2]; const float* bc[2]; const float* gam[2]; const float* bet[2];
    const float* Wr; const float* br;
    const float* tb; int tb_ld;
    const float* ln_g; const float* Wqkv; const float* Wo; const float* bo;
    const float* Wu; const float* bu;
    const int* t_ptr; int t_imm;
    int L;
    Pf pf; PhaseBuf ph;                                 // L2 warm-up for the next launch
};

__global__ __launch_bounds__(256) void ups_tail128_kernel(const UpsTailArgs a) {
    PH_DECL;
    PH(0);        // phase clocks (profiling builds, kernels.h PhaseBuf): mark k follows the k-th workgroup barrier, the last one the final stores
    constexpr int C = 128, CI = 256, NP = 16, ROWS = NP + 4;
    constexpr int XPB = 2 * CI + 16, PPB = 2 * C + 16, APB = 2 * 128 + 16, HP = C + 4;
    __shared__ __attribute__((aligned(16))) unsigned char XI[2][ROWS * XPB];
    __shared__ __attribute__((aligned(16))) unsigned char P[2][2][ROWS * PPB];
    __shared__ __attribute__((aligned(16))) unsigned char R[2 * NP * APB];             // h2 in fp32 for the LayerNorm, then the att planes
    // parameter vectors: 0-2 conv1 (bias, GN weight, GN bias), 3-5 conv2, 6 time bias, 7 residual bias, 8 to_out bias, 9 upsample bias
    __shared__ __attribute__((aligned(16))) float PV[10][C];
    static_assert(NP * HP * 4 <= 2 * NP * APB, "H fits the shared region");
    float* H = reinterpret_cast<float*>(R);
    unsigned char* Aph = R; unsigned char* Apl = R + NP * APB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int L = a.L, L2 = 2 * a.L, b = blockIdx.x;
    const int t_now = step_scalar(a.t_ptr, a.t_imm);
    auto cl = [&](int mt) { return (2 * w + mt) * 16 + lq * 4; };
    auto wtile = [&](const float* W, int tile, int taps, int ks) { return reinterpret_cast<const float4*>(W) + (size_t)tile * taps * ks * 2 * 64; };
    LvlRing<1, 8> ring, ring_r;
    LvlRing<1, 8, 2> ring8;                                              // the 256-input-channel block: two taps deep (see LvlRing)
    lvlm_prefetch<1, 5, 8, 8>(ring8, wtile(a.Wc[0], 2 * w, 5, 8), lane);
    lvlm_prefetch<1, 1, 8, 8>(ring_r, wtile(a.Wr, 2 * w, 1, 8), lane);
    PfRegs pfr;
    l2_prefetch_early(a.pf, pfr);
    // (round 4) the input rows are requested BEFORE the parameter vectors go to LDS: `PV[i][tid] = src[i][tid]` waits for its loads,
    // and rows requested after that wait were a second serial round trip at the head of the launch
    float4 xin[4];                                                       // the input rows, requested before the zero fill and its barrier
    if ((tid >> 4) < L) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xin[q] = *reinterpret_cast<const float4*>(a.x + ((size_t)b * L + (tid >> 4)) * CI + 4 * ((tid & 15) + 16 * q));
    }
    float pvr[10];                                                       // requested now, written to LDS after the zero fill
    if (tid < C) {
        const float* src[10] = {a.bc[0], a.gam[0], a.bet[0], a.bc[1], a.gam[1], a.bet[1], a.tb + (size_t)t_now * a.tb_ld, a.br, a.bo, a.bu};
#pragma unroll
        for (int i = 0; i < 10; ++i) pvr[i] = src[i][tid];
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int i = tid; i < 2 * ROWS * XPB / 16; i += 256) reinterpret_cast<float4*>(&XI[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < 4 * ROWS * PPB / 16; i += 256) reinterpret_cast<float4*>(&P[0][0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < C) {
#pragma unroll
        for (int i = 0; i < 10; ++i) PV[i][tid] = pvr[i];
    }
    __syncthreads();
    PH(1);
    {
        const int p = tid >> 4, c4 = tid & 15;
        if (p < L) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cf = c4 + 16 * q;
                const float4 v = xin[q];
                half4v hi, lo;
                hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
                lo[0] = (_Float16)((v.x - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((v.y - (float)hi[1]) * H3_SCALE);
                lo[2] = (_Float16)((v.z - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((v.w - (float)hi[3]) * H3_SCALE);
                *reinterpret_cast<half4v*>(&XI[0][(p + 2) * XPB + 8 * cf]) = hi;
                *reinterpret_cast<half4v*>(&XI[1][(p + 2) * XPB + 8 * cf]) = lo;
            }
        }
    }
    auto pv4 = [&](int vec, int mt) { return *reinterpret_cast<const float4*>(&PV[vec][cl(mt)]); };
    auto add4 = [&](f32x4& v, const float4 t) { v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; };
    auto planes = [&](const f32x4 (&v)[2], unsigned char* Ph, unsigned char* Pl, int nvalid) {      // rows position + 2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            half4v hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = lr < nvalid ? v[mt][i] : 0.f;
                hi[i] = (_Float16)f; lo[i] = (_Float16)((f - (float)hi[i]) * H3_SCALE);
            }
            const int off = (lr + 2) * PPB + 2 * cl(mt);
            *reinterpret_cast<half4v*>(Ph + off) = hi;
            *reinterpret_cast<half4v*>(Pl + off) = lo;
        }
    };
    auto store = [&](const f32x4 (&v)[2], float* dst, int nvalid) {      // dst [Bp, nvalid, 128]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            if (lr < nvalid) st_out4(dst, ((size_t)b * nvalid + lr) * C + cl(mt), make_float4(v[mt][0], v[mt][1], v[mt][2], v[mt][3]), a.pf.wt);
    };
    __syncthreads();
    PH(2);

    // ---- ResidualTemporalBlock(256 -> 128): the wave's two 16-channel tiles one after the other ----
    f32x4 v[2], r[2], h2[2];
    {
        f32x4 t[1][1];
        lvlm_conv<1, 1, 5, 8, XPB, 8>(ring8, wtile(a.Wc[0], 2 * w, 5, 8), XI[0], XI[1], 0, 1, 0, ROWS - 1, lane, t); v[0] = t[0][0];
        lvlm_prefetch<1, 5, 8, 8>(ring8, wtile(a.Wc[0], 2 * w + 1, 5, 8), lane);
        lvlm_conv<1, 1, 1, 8, XPB, 8>(ring_r, wtile(a.Wr, 2 * w, 1, 8), XI[0], XI[1], 0, 1, 2, ROWS - 1, lane, t); r[0] = t[0][0];
        lvlm_prefetch<1, 1, 8, 8>(ring_r, wtile(a.Wr, 2 * w + 1, 1, 8), lane);
        lvlm_conv<1, 1, 5, 8, XPB, 8>(ring8, wtile(a.Wc[0], 2 * w + 1, 5, 8), XI[0], XI[1], 0, 1, 0, ROWS - 1, lane, t); v[1] = t[0][0];
        lvlm_prefetch<1, 5, 4, 8>(ring, wtile(a.Wc[1], 2 * w, 5, 4), lane);
        lvlm_conv<1, 1, 1, 8, XPB, 8>(ring_r, wtile(a.Wr, 2 * w + 1, 1, 8), XI[0], XI[1], 0, 1, 2, ROWS - 1, lane, t); r[1] = t[0][0];
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        lvlm_gn_mish(v[mt], pv4(0, mt), pv4(1, mt), pv4(2, mt), L, lane);
        add4(v[mt], pv4(6, mt));
        add4(r[mt], pv4(7, mt));
    }
    planes(v, P[0][0], P[0][1], L);
    __syncthreads();
    PH(3);
    {
        f32x4 t[1][1];
        lvlm_conv<1, 1, 5, 4, PPB, 8>(ring, wtile(a.Wc[1], 2 * w, 5, 4), P[0][0], P[0][1], 0, 1, 0, ROWS - 1, lane, t); v[0] = t[0][0];
        lvlm_prefetch<1, 5, 4, 8>(ring, wtile(a.Wc[1], 2 * w + 1, 5, 4), lane);
        lvlm_conv<1, 1, 5, 4, PPB, 8>(ring, wtile(a.Wc[1], 2 * w + 1, 5, 4), P[0][0], P[0][1], 0, 1, 0, ROWS - 1, lane, t); v[1] = t[0][0];
        lvlm_prefetch<1, 1, 4, 8>(ring, wtile(a.Wo, 2 * w, 1, 4), lane);            // to_out fragments of the first tile: in flight through the attention
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        lvlm_gn_mish(v[mt], pv4(3, mt), pv4(4, mt), pv4(5, mt), L, lane);
        h2[mt] = v[mt] + r[mt];
        *reinterpret_cast<float4*>(&H[lr * HP + cl(mt)]) = make_float4(h2[mt][0], h2[mt][1], h2[mt][2], h2[mt][3]);
    }
    if (a.h2) store(h2, a.h2, L);
    __syncthreads();
    PH(4);
    // ---- attention site (C = 128) ----
    {
        const int lrow = tid >> 5, lcol = tid & 31;                      // 32 lanes per row, 8 rows per pass
        const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + 4 * lcol);
#pragma unroll
        for (int rr = 0; rr < NP / 8; ++rr) {
            const int n = rr * 8 + lrow;
            const float4 xv = *reinterpret_cast<const float4*>(&H[n * HP + 4 * lcol]);
            const float s1 = xsum16(row16_sum((xv.x + xv.y) + (xv.z + xv.w)));
            const float mean = s1 * (1.0f / C);
            const float d0 = xv.x - mean, d1 = xv.y - mean, d2 = xv.z - mean, d3 = xv.w - mean;
            const float s2 = xsum16(row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)));
            const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + 1e-5f);
            const bool ok = n < L;
            const float y0 = ok ? d0 * rstd * gv.x : 0.f, y1 = ok ? d1 * rstd * gv.y : 0.f, y2 = ok ? d2 * rstd * gv.z : 0.f, y3 = ok ? d3 * rstd * gv.w : 0.f;
            half4v hi, lo;
            hi[0] = (_Float16)y0; hi[1] = (_Float16)y1; hi[2] = (_Float16)y2; hi[3] = (_Float16)y3;
            lo[0] = (_Float16)((y0 - (float)hi[0]) * H3_SCALE); lo[1] = (_Float16)((y1 - (float)hi[1]) * H3_SCALE);
            lo[2] = (_Float16)((y2 - (float)hi[2]) * H3_SCALE); lo[3] = (_Float16)((y3 - (float)hi[3]) * H3_SCALE);
            *reinterpret_cast<half4v*>(&P[1][0][(n + 2) * PPB + 8 * lcol]) = hi;
            *reinterpret_cast<half4v*>(&P[1][1][(n + 2) * PPB + 8 * lcol]) = lo;
        }
    }
    __syncthreads();
    PH(5);
    f32x4 qa[2][1], ka[1][2], va[1][2];
    {
        const float4* Wq4 = reinterpret_cast<const float4*>(a.Wqkv);
        f32x4 M[6], Lo[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) { M[s6] = f32x4{0.f, 0.f, 0.f, 0.f}; Lo[s6] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float4 wq[2][6][2];
        auto load_k = [&](int k, int slot) {
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const int tile = (s6 >> 1) * 8 + 2 * w + (s6 & 1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wq[slot][s6][pl] = Wq4[(((size_t)tile * 4 + k) * 2 + pl) * 64 + lane];
            }
        };
        load_k(0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k + 1 < 4) load_k(k + 1, (k + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int off = (lr + 2) * PPB + k * 64 + lq * 16;
            const half8 yh = *reinterpret_cast<const half8*>(&P[1][0][off]);
            const half8 yl = *reinterpret_cast<const half8*>(&P[1][1][off]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half8 wqh = __builtin_bit_cast(half8, wq[k & 1][i][0]), wql = __builtin_bit_cast(half8, wq[k & 1][i][1]);
                M[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wqh, yh, M[i], 0, 0, 0);
                Lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wqh, yl, Lo[i], 0, 0, 0);
                Lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wql, yh, Lo[i], 0, 0, 0);
#pragma unroll
                for (int kv = 2; kv < 6; kv += 2) {
                    const half8 wh = __builtin_bit_cast(half8, wq[k & 1][kv + i][0]), wl = __builtin_bit_cast(half8, wq[k & 1][kv + i][1]);
                    M[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh, M[kv + i], 0, 0, 0);
                    Lo[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl, Lo[kv + i], 0, 0, 0);
                    Lo[kv + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh, Lo[kv + i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            qa[i][0] = M[i] + Lo[i] * H3_INV;
            ka[0][i] = M[2 + i] + Lo[2 + i] * H3_INV;
            va[0][i] = M[4 + i] + Lo[4 + i] * H3_INV;
        }
    }
    f32x4 att[2][1];
    attn_site_core<1>(qa, ka, va, att, 1, NP, NP, L, lq, lr);
    __syncthreads();                                          // every wave is done with H (the att planes alias it)
    PH(6);
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        half4v hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)att[et][0][i]; lo[i] = (_Float16)((att[et][0][i] - (float)hi[i]) * H3_SCALE); }
        const int off = lr * APB + 2 * (w * 32 + et * 16 + lq * 4);
        *reinterpret_cast<half4v*>(Aph + off) = hi;
        *reinterpret_cast<half4v*>(Apl + off) = lo;
    }
    __syncthreads();
    PH(7);
    f32x4 h3[2];
    {
        f32x4 t[1][1];
        lvlm_conv<1, 1, 1, 4, APB, 8>(ring, wtile(a.Wo, 2 * w, 1, 4), Aph, Apl, 0, 1, 0, NP - 1, lane, t); h3[0] = t[0][0];
        lvlm_prefetch<1, 1, 4, 8>(ring, wtile(a.Wo, 2 * w + 1, 1, 4), lane);
        lvlm_conv<1, 1, 1, 4, APB, 8>(ring, wtile(a.Wo, 2 * w + 1, 1, 4), Aph, Apl, 0, 1, 0, NP - 1, lane, t); h3[1] = t[0][0];
        lvlm_prefetch<1, 4, 4, 8>(ring, wtile(a.Wu, 2 * w, 4, 4), lane);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { add4(h3[mt], pv4(8, mt)); h3[mt] += h2[mt]; }
    if (a.h3) store(h3, a.h3, L);
    planes(h3, P[0][0], P[0][1], L);
    __syncthreads();
    PH(8);
    // ---- Upsample1d: ConvTranspose1d(k = 4, stride 2, pad 1), L -> 2L <= 16 positions (one tile) ----
    {
        f32x4 u[2];
        f32x4 t[1][1];
        lvlm_conv<1, 1, 4, 4, PPB, 8, 1>(ring, wtile(a.Wu, 2 * w, 4, 4), P[0][0], P[0][1], 0, 0, 0, ROWS - 1, lane, t); u[0] = t[0][0];
        lvlm_prefetch<1, 4, 4, 8>(ring, wtile(a.Wu, 2 * w + 1, 4, 4), lane);
        l2_prefetch_late(a.pf, pfr);
        lvlm_conv<1, 1, 4, 4, PPB, 8, 1>(ring, wtile(a.Wu, 2 * w + 1, 4, 4), P[0][0], P[0][1], 0, 0, 0, ROWS - 1, lane, t); u[1] = t[0][0];
        add4(u[0], pv4(9, 0)); add4(u[1], pv4(9, 1));
        store(u, a.up, L2);
    }
    PH(9);
    l2_prefetch_done(a.pf, pfr);
    PH_FLUSH(a.ph);
}

// ---------------------------------------------------------------------------------------------
// Counter-based Gaussian noise: Philox4x32-10 keyed by seed, counter = (element/4, sample, step, 0),
// Box-Muller on the four 32-bit outputs.  Pure function of (seed, global sample, step, element):
// results do not depend on the number of GPUs / batch partition.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float counter_normal(uint64_t seed, uint64_t sample, uint32_t step, uint32_t elem) {
    uint32_t r[4];
    philox4x32_10(elem >> 2, (uint32_t)sample, step, (uint32_t)(sample >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int j = elem & 3;
    const uint32_t ra = r[j & 2], rb = r[(j & 2) + 1];
    const float u1 = ((float)(ra >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(rb >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    return (j & 1) ? rad * sn : rad * cs;
}

// the four normals of elements 4*elem4 .. 4*elem4+3 (identical values to counter_normal on those elements)
__device__ __forceinline__ void counter_normal4(uint64_t seed, uint64_t sample, uint32_t step, uint32_t elem4, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10(elem4, (uint32_t)sample, step, (uint32_t)(sample >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(r[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
    }
}

__global__ void fill_normal_kernel(float* out, int64_t B, int64_t per, uint64_t seed, int64_t off, uint32_t step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const int64_t b = i / per, e = i - b * per;
    out[i] = counter_normal(seed, (uint64_t)(off + b), step, (uint32_t)e);
}

// ---------------------------------------------------------------------------------------------
// Composition: gather U-Net input rows, and scatter-aggregate + DDPM posterior update.
// (struct ComposeArgs is defined above UpsLastArgs: ups_last_kernel can carry the update of a plain single-model step)

__device__ __forceinline__ int pair_index(int i, int j, int nb) {   // i < j, order (0,1),(0,2),..,(1,2),..
    return i * nb - (i * (i + 1)) / 2 + (j - i - 1);
}

// state value at full-sequence row l (cond rows first when cond_steps > 0)
__device__ __forceinline__ float full_x(const ComposeArgs& a, int64_t b, int l, int f) {
    if (l < a.cond_steps) return a.cond[((size_t)b * a.cond_steps + l) * a.F + f];
    return a.x[((size_t)b * a.Ltot + (l - a.cond_steps)) * a.F + f];
}

__global__ void compose_gather_kernel(const ComposeArgs a) {
    // pair rows
    const int P = a.nb * (a.nb - 1) / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.mode == 0) {                       // plain model on cat(cond, x): [B, Lfull, F]
        const int Lfull = a.Ltot + a.cond_steps;
        if (i >= a.B * (int64_t)Lfull * a.F) return;
        const int f = (int)(i % a.F);
        const int l = (int)((i / a.F) % Lfull);
        a.pair_in[i] = full_x(a, i / ((int64_t)a.F * Lfull), l, f);
        return;
    }
    const int64_t npair = (int64_t)a.W * P * a.B * a.T * 8;
    const int64_t nsingle = (a.mode == 5) ? (int64_t)a.nb * a.B * a.T * 4 : 0;
    if (i < npair) {
        const int c = (int)(i & 7);
        int64_t r = i >> 3;
        const int l = (int)(r % a.T); r /= a.T;
        const int64_t b = r % a.B; r /= a.B;
        const int p = (int)(r % P); const int kk = (int)(r / P);
        // decode pair p -> (ii, jj)
        int ii = 0, rem = p;
        while (rem >= a.nb - 1 - ii) { rem -= a.nb - 1 - ii; ++ii; }
        const int jj = ii + 1 + rem;
        const int f = (c < 4) ? ii * 4 + c : jj * 4 + (c - 4);
        a.pair_in[i] = full_x(a, b, kk * a.cs + l, f);
    } else if (i < npair + nsingle) {
        const int64_t k = i - npair;
        const int c = (int)(k & 3);
        int64_t r = k >> 2;
        const int l = (int)(r % a.T); r /= a.T;
        const int64_t b = r % a.B; const int body = (int)(r / a.B);
        a.single_in[k] = full_x(a, b, l, body * 4 + c);
    }
}

// One thread per state element (b, l, f) of the FULL sequence (cond rows skipped on output).
__device__ void compose_update_element(const ComposeArgs& a, int64_t i) {
    const int Lfull = a.Ltot + a.cond_steps;
    const int t = step_scalar(a.t_ptr, a.t_imm);
    const int sidx = a.ddim_tab ? *a.step_idx : 0;
    const uint64_t dseed = a.dyn ? (uint64_t)a.dyn[0] : a.seed;
    const int64_t dsoff = a.dyn ? (int64_t)a.dyn[1] : a.sample_off;
    if (i == 0) compose_advance(a, t);
    if (i < a.B * (int64_t)a.Ltot * a.F) {
    const int f = (int)(i % a.F);
    const int lx = (int)((i / a.F) % a.Ltot);
    const int64_t b = i / ((int64_t)a.F * a.Ltot);
    const int l = lx + a.cond_steps;             // row in the full sequence
    const int body = f >> 2, comp = f & 3;
    const int P = a.nb * (a.nb - 1) / 2;
    const float xv = a.x[i];
    const float ra = a.sqrt_recip[t], rb = a.sqrt_recipm1[t], c1 = a.coef1[t], c2 = a.coef2[t];

    auto pair_eps_at = [&](int kk, int other, int lw) -> float {
        // eps of `body` from the pair {body, other} in window kk at window row lw
        const int ii = min(body, other), jj = max(body, other);
        const int p = pair_index(ii, jj, a.nb);
        const int slot = (body == ii) ? 0 : 1;
        const int64_t row = ((int64_t)kk * P + p) * a.B + b;
        return a.pair_eps[(row * a.T + lw) * 8 + slot * 4 + comp];
    };
    // explicit roundings (no FMA contraction): the reference evaluates these as separate elementwise ops, and the
    // step identities (outside/mean == inside == plain for one window, one pair) must hold bitwise
    auto x0_of = [&](float eps_or_out) -> float {
        float x0;
        if (a.objective == 0) x0 = __fsub_rn(__fmul_rn(ra, xv), __fmul_rn(rb, eps_or_out));
        else if (a.objective == 1) x0 = eps_or_out;
        else x0 = __fsub_rn(__fmul_rn(a.sqrt_ac[t], xv), __fmul_rn(a.sqrt_1mac[t], eps_or_out));
        return x0;
    };
    auto post_mean = [&](float x0v) -> float { return __fadd_rn(__fmul_rn(c1, x0v), __fmul_rn(c2, xv)); };

    float eps = 0.f, x0 = 0.f, mean = 0.f;
    int cover = 0;
    for (int kk = 0; kk < a.W; ++kk) { const int lw = l - kk * a.cs; if (lw >= 0 && lw < a.T) ++cover; }

    if (a.mode == 0) {                       // plain: pair_eps is the model output on the full state
        const float o = a.pair_eps[((size_t)b * Lfull + l) * a.F + f];
        x0 = x0_of(o);
        eps = (a.objective == 0) ? o : (ra * xv - x0) / rb;
        if (a.clip) x0 = clamp_pm1(x0);
        mean = post_mean(x0);
    } else if (a.mode == 1 || a.mode == 2 || a.mode == 4) {
        // eps aggregated over senders then windows (model/diffusion_1d.py:994-999, :1457-1458)
        float tot = 0.f;
        for (int kk = 0; kk < a.W; ++kk) {
            const int lw = l - kk * a.cs;
            if (lw < 0 || lw >= a.T) continue;
            float s = 0.f;
            for (int o = 0; o < a.nb; ++o) if (o != body) s += pair_eps_at(kk, o, lw);
            if (a.mode == 1) s /= (float)(a.nb - 1);
            tot += s;
        }
        const float o = (a.mode == 1) ? tot / (float)cover : tot / ((float)cover / (float)a.W);
        x0 = x0_of(o);
        eps = (a.objective == 0) ? o : (ra * xv - x0) / rb;
        if (a.clip) x0 = clamp_pm1(x0);
        mean = post_mean(x0);
    } else if (a.mode == 3) {
        // p_mean_variance per (window, pair), then average mean and x0 (:1436-1452)
        float tm = 0.f, tx = 0.f, te = 0.f;
        for (int kk = 0; kk < a.W; ++kk) {
            const int lw = l - kk * a.cs;
            if (lw < 0 || lw >= a.T) continue;
            float sm = 0.f, sx = 0.f, se = 0.f;
            for (int o = 0; o < a.nb; ++o) {
                if (o == body) continue;
                const float e = pair_eps_at(kk, o, lw);
                float x0e = x0_of(e);
                if (a.clip) x0e = clamp_pm1(x0e);
                sm += post_mean(x0e); sx += x0e; se += e;
            }
            tm += sm / (float)(a.nb - 1); tx += sx / (float)(a.nb - 1); te += se / (float)(a.nb - 1);
        }
        mean = tm / (float)cover; x0 = tx / (float)cover; eps = te / (float)cover;
    } else {                                 // mode 5: gradient() :1900-1922
        float s = 0.f;
        for (int o = 0; o < a.nb; ++o) if (o != body) s += pair_eps_at(0, o, l);
        const float u = a.single_eps[(((int64_t)body * a.B + b) * a.T + l) * 4 + comp];
        const float o = s - a.uncond_coef * u;
        x0 = x0_of(o);
        eps = (a.objective == 0) ? o : (ra * xv - x0) / rb;
        if (a.clip) x0 = clamp_pm1(x0);
        mean = post_mean(x0);
    }

    if (a.mean_out) a.mean_out[i] = mean;
    if (a.x0_out) a.x0_out[i] = x0;
    if (a.eps_out) a.eps_out[i] = eps;
    if (a.x_out && a.dz_mode) {
        // guided update with the built-in objective (x_out never aliases x here: the gradient reads neighbours)
        float g = 0.f;
        if (comp < 2) {
            if (lx >= a.Ltot - a.dz_last_n) {
                const float d = xv - (comp == 0 ? a.dz_tx : a.dz_ty);
                const float scale = a.dz_coef / (float)a.dz_last_n;
                if (a.dz_mode == 1) {
                    const float dother = a.x[i ^ 1] - (comp == 0 ? a.dz_ty : a.dz_tx);
                    g = scale * d / sqrtf(d * d + dother * dother);
                } else {
                    g = scale * 2.0f * d;
                }
            }
            if (a.dz_tc > 0.f && a.Ltot > 1) {
                float lap = 0.f;
                if (lx >= 1) lap += xv - a.x[i - a.F];
                if (lx + 1 < a.Ltot) lap -= a.x[i + a.F] - xv;
                g += a.dz_tc * 2.0f * lap / (float)(a.Ltot - 1);
            }
        }
        if (a.dz_alpha) g *= a.betas[t] / sqrtf(a.acp[t]);
        float pred = mean - g;
        if (a.iso && lx < a.iso_steps) pred = a.iso[((size_t)b * a.iso_steps + lx) * a.F + f];
        const uint32_t el = (uint32_t)(lx * a.F + f);
        float v;
        if (a.relax) {
            const float ratio = a.ac[t] / a.acp[t];
            const float z = a.recur_noise ? a.recur_noise[(size_t)t * a.recur_t_stride + i]
                                          : counter_normal(dseed ^ 0x7f4a7c15u, (uint64_t)(dsoff + b), a.recur_tag + (uint32_t)t, el);
            v = sqrtf(ratio) * pred + sqrtf(1.0f - ratio) * z;
        } else {
            v = pred;
            if (a.add_noise && t > 0) {
                const float z = a.noise ? a.noise[(size_t)t * a.noise_t_stride + i]
                                        : counter_normal(dseed, (uint64_t)(dsoff + b), (uint32_t)t, el);
                v += expf(0.5f * a.logvar[t]) * z;
            }
            if (a.inp_cond && lx < a.inp_steps) {
                const size_t ci = ((size_t)b * a.inp_steps + lx) * a.F + f;
                const float z = a.inp_noise ? a.inp_noise[(size_t)t * a.inp_noise_t_stride + ci]
                                            : counter_normal(dseed ^ 0x5bd1e995u, (uint64_t)(dsoff + b), (uint32_t)t, el);
                v = a.sqrt_ac[t] * a.inp_cond[ci] + a.sqrt_1mac[t] * z;
            }
        }
        a.x_out[i] = v;
    } else if (a.x_out && a.ddim_tab) {
        // x_{next} = x0 * sqrt(alpha_next) + c * eps + sigma * z (:1781-1783); the last step returns x0 (:1784-1789)
        const int tn = a.ddim_tnext[sidx];
        const uint32_t el = (uint32_t)(lx * a.F + f);
        float v = x0;
        if (tn >= 0) {
            const float san = a.ddim_tab[4 * sidx], cc = a.ddim_tab[4 * sidx + 1], sg = a.ddim_tab[4 * sidx + 2];
            const float z = a.noise ? a.noise[(size_t)sidx * a.noise_t_stride + i]
                            : (sg != 0.f ? counter_normal(dseed, (uint64_t)(dsoff + b), (uint32_t)t, el) : 0.f);
            v = __fadd_rn(__fadd_rn(__fmul_rn(x0, san), __fmul_rn(cc, eps)), __fmul_rn(sg, z));
            if (a.inp_cond && lx < a.inp_steps) {      // inpainting overwrite with q_sample(cond, time) (:1790-1793)
                const size_t ci = ((size_t)b * a.inp_steps + lx) * a.F + f;
                const float z2 = a.inp_noise ? a.inp_noise[(size_t)sidx * a.inp_noise_t_stride + ci]
                                             : counter_normal(dseed ^ 0x5bd1e995u, (uint64_t)(dsoff + b), (uint32_t)t, el);
                v = a.sqrt_ac[t] * a.inp_cond[ci] + a.sqrt_1mac[t] * z2;
            }
        }
        a.x_out[i] = v;
    } else if (a.x_out) {
        float v = mean;
        const uint32_t el = (uint32_t)(lx * a.F + f);
        if (a.add_noise && t > 0) {
            const float z = a.noise ? a.noise[(size_t)t * a.noise_t_stride + i]
                                    : counter_normal(dseed, (uint64_t)(dsoff + b), (uint32_t)t, el);
            v += expf(0.5f * a.logvar[t]) * z;
        }
        if (a.inp_cond && lx < a.inp_steps) {      // inpainting overwrite (:1715-1718)
            const size_t ci = ((size_t)b * a.inp_steps + lx) * a.F + f;
            const float z = a.inp_noise ? a.inp_noise[(size_t)t * a.inp_noise_t_stride + ci]
                                        : counter_normal(dseed ^ 0x5bd1e995u, (uint64_t)(dsoff + b), (uint32_t)t, el);
            v = a.sqrt_ac[t] * a.inp_cond[ci] + a.sqrt_1mac[t] * z;
        }
        a.x_out[i] = v;
    }
    }
}

__global__ void compose_update_kernel(const ComposeArgs a) {
    compose_update_element(a, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Advances the device-side step counter after the update kernel of a step (its own graph node: every reader of t in this
// step has finished).  A "last block done" atomic inside the update kernel cost one same-address device-scope atomic
// and one release fence per block: 46 ns each, 282 us per step for the 6144 blocks of the 2-D update.
// t_dev[0] = t, t_dev[2] = DDIM step index.
// e0 / e1: per-forward epochs of the U-Nets the NEXT step will run (dconv_kernel's pair exchanges), advanced here so
// that a step needs no epoch launch of its own; null = none
__global__ void step_counter_kernel(int* t_dev, const int* ddim_tnext, int* e0, int* e1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (ddim_tnext) { const int sidx = t_dev[2]; t_dev[0] = max(ddim_tnext[sidx], 0); t_dev[2] = sidx + 1; }
        else t_dev[0] -= 1;
        if (e0) e0[0] = max(e0[0], e0[8]) + 1;      // (slot 8: the ping-pong loop's other epoch slot; tags must never repeat)
        if (e1) e1[0] = max(e1[0], e1[8]) + 1;
    }
}

}  // namespace cindm
