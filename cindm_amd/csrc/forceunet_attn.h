// ForceUnet's bottleneck Attention (model/diffusion_2d.py:256-278) at its only shape -- 64 tokens (the 8 x 8 level), 4 heads
// of 32 -- forward and backward on the exact fp32 MFMA.  Round 2's kernels were one scalar thread per query row with
// 1536-byte-strided global reads (forward 130 us, backward 624 us per design-gradient call at 768 images, for 2 GFLOP).
// Workgroup = (image, head); q (pre-scaled by 32^-1/2), k, v (and dout) are staged once with coalesced float4 rows; wave w owns
// query rows 16 w .. 16 w + 15 for S = q k^T, the softmax and dP = dout v^T (accumulator rows = queries, columns = keys; the
// softmax reductions are DPP row operations over the 16 lanes of a row), parks P (and dS) in LDS, and takes the second
// products from there: O / dq for its own query block, dk / dv for KEY block w (a contraction over all 64 queries).
#pragma once
#include "forceunet_la.h"

namespace cindm {

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_get<0x128>(v)); v = fmaxf(v, dpp_get<0x124>(v)); v = fmaxf(v, dpp_get<0x4E>(v)); v = fmaxf(v, dpp_get<0xB1>(v));
    return v;
}

// rows i = 16 w + 4 lq + r, columns j = 16 jt + lr of  A[i][:] . B[j][:]  over 32 channels (both operands [64][33] in LDS)
__device__ __forceinline__ void fu_attn_rows_x_rows(const float (*A)[33], const float (*B)[33], int w, int lr, int lq, f32x4 (&S)[4]) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) S[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
        const float av = A[16 * w + lr][4 * k8 + lq];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) S[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, B[16 * jt + lr][4 * k8 + lq], S[jt], 0, 0, 0);
    }
}
// softmax over the 64 columns of every accumulator row (in place)
__device__ __forceinline__ void fu_attn_softmax(f32x4 (&S)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mx = row16_max(fmaxf(fmaxf(S[0][r], S[1][r]), fmaxf(S[2][r], S[3][r])));
        float sm = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) { S[jt][r] = __expf(S[jt][r] - mx); sm += S[jt][r]; }
        sm = row16_sum(sm);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) S[jt][r] /= sm;
    }
}
// out rows (token block rb), 32 channels:  sum_k M[row][k] X[k][:]  (TRANS: sum_k M[k][row] X[k][:]) over 64 tokens k
template <bool TRANS>
__device__ __forceinline__ void fu_attn_mat_x_rows(const float (*M)[65], const float (*X)[33], int rb, int lr, int lq, f32x4 (&O)[2]) {
    O[0] = f32x4{0.f, 0.f, 0.f, 0.f}; O[1] = O[0];
#pragma unroll
    for (int k16 = 0; k16 < 16; ++k16) {
        const int k = 4 * k16 + lq;
        const float av = TRANS ? M[k][16 * rb + lr] : M[16 * rb + lr][k];
        O[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, X[k][lr], O[0], 0, 0, 0);
        O[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, X[k][16 + lr], O[1], 0, 0, 0);
    }
}

template <bool BWD>
__device__ __forceinline__ void fu_attn_stage(const float* __restrict__ qkv, const float* __restrict__ dout, int img, int h, int tid,
                                              float (*Q)[33], float (*K)[33], float (*V)[33], float (*G)[33]) {
    const float sc = 0.17677669529663687f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (tid >> 3) + 32 * p, c = (tid & 7) * 4;
        const float* b = qkv + ((size_t)img * 64 + row) * 384 + h * 32 + c;
        const float4 q4 = *reinterpret_cast<const float4*>(b), k4 = *reinterpret_cast<const float4*>(b + 128), v4 = *reinterpret_cast<const float4*>(b + 256);
        Q[row][c] = q4.x * sc; Q[row][c + 1] = q4.y * sc; Q[row][c + 2] = q4.z * sc; Q[row][c + 3] = q4.w * sc;
        K[row][c] = k4.x; K[row][c + 1] = k4.y; K[row][c + 2] = k4.z; K[row][c + 3] = k4.w;
        V[row][c] = v4.x; V[row][c + 1] = v4.y; V[row][c + 2] = v4.z; V[row][c + 3] = v4.w;
        if constexpr (BWD) {
            const float4 g4 = *reinterpret_cast<const float4*>(dout + ((size_t)img * 64 + row) * 128 + h * 32 + c);
            G[row][c] = g4.x; G[row][c + 1] = g4.y; G[row][c + 2] = g4.z; G[row][c + 3] = g4.w;
        }
    }
}

// out[img][i][h*32 + d] = sum_j softmax_j(q_i . k_j / sqrt(32)) v_j[d]      grid (4 heads, images), 256 threads, n = 64
__global__ __launch_bounds__(256) void fu_attn64_kernel(const float* __restrict__ qkv, float* __restrict__ out) {
    __shared__ float Q[64][33], K[64][33], V[64][33], P[64][65];
    const int img = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    fu_attn_stage<false>(qkv, nullptr, img, h, tid, Q, K, V, nullptr);
    __syncthreads();
    f32x4 S[4];
    fu_attn_rows_x_rows(Q, K, w, lr, lq, S);
    fu_attn_softmax(S);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[16 * w + 4 * lq + r][16 * jt + lr] = S[jt][r];
    __syncthreads();
    f32x4 O[2];
    fu_attn_mat_x_rows<false>(P, V, w, lr, lq, O);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((size_t)img * 64 + 16 * w + 4 * lq + r) * 128 + h * 32 + 16 * dt + lr] = O[dt][r];
}

// dqkv[img][.][q | k | v columns of head h] from dout (Attention backward, weights frozen)
__global__ __launch_bounds__(256) void fu_attn64_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv) {
    __shared__ float Q[64][33], K[64][33], V[64][33], G[64][33], P[64][65], dS[64][65];
    const int img = blockIdx.y, h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    fu_attn_stage<true>(qkv, dout, img, h, tid, Q, K, V, G);
    __syncthreads();
    f32x4 S[4], D[4];
    fu_attn_rows_x_rows(Q, K, w, lr, lq, S);
    fu_attn_softmax(S);
    fu_attn_rows_x_rows(G, V, w, lr, lq, D);                  // dP[i][j] = dout_i . v_j
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float dot = row16_sum((S[0][r] * D[0][r] + S[1][r] * D[1][r]) + (S[2][r] * D[2][r] + S[3][r] * D[3][r]));
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            P[16 * w + 4 * lq + r][16 * jt + lr] = S[jt][r];
            dS[16 * w + 4 * lq + r][16 * jt + lr] = S[jt][r] * (D[jt][r] - dot);
        }
    }
    __syncthreads();
    const float sc = 0.17677669529663687f;
    f32x4 O[2];
    float* ob = dqkv + ((size_t)img * 64 + 16 * w + 4 * lq) * 384 + h * 32 + lr;
    fu_attn_mat_x_rows<false>(dS, K, w, lr, lq, O);           // dq_i = scale * sum_j dS_ij k_j
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)r * 384 + 16 * dt] = O[dt][r] * sc;
    fu_attn_mat_x_rows<true>(dS, Q, w, lr, lq, O);            // dk_j = sum_i dS_ij q_i (q carries the scale)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)r * 384 + 128 + 16 * dt] = O[dt][r];
    fu_attn_mat_x_rows<true>(P, G, w, lr, lq, O);             // dv_j = sum_i P_ij dout_i
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)r * 384 + 256 + 16 * dt] = O[dt][r];
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention for MORE than 64 tokens (coarsest level 16 x 16 or 32 x 32: n = 256 / 1024, a multiple of 64).  Not a
// shape of the paper's configuration (image_size 64, four levels -> 8 x 8), so plain fp32 arithmetic: thread = query row (or
// key row), the other side streamed through LDS in 64-token chunks, online softmax.  grid (4 heads, images, n / 64).
// stat[((img * 4 + h) * n + i) * 3 + {0, 1, 2}] = row maximum, row sum, D_i = dout_i . out_i (the backward's softmax term).
__global__ __launch_bounds__(64) void fu_attn_gen_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ stat, int n) {
    __shared__ float K[64][33], V[64][33];
    const int h = blockIdx.x, img = blockIdx.y, t = threadIdx.x, i = blockIdx.z * 64 + t;
    const float* base = qkv + (size_t)img * n * 384 + h * 32;
    float q[32], o[32];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)i * 384 + d4 * 4);
        q[d4 * 4] = v.x * 0.17677669529663687f; q[d4 * 4 + 1] = v.y * 0.17677669529663687f;
        q[d4 * 4 + 2] = v.z * 0.17677669529663687f; q[d4 * 4 + 3] = v.w * 0.17677669529663687f;
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int c = 0; c < n; c += 64) {
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)(c + t) * 384 + 128 + d4 * 4);
            const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)(c + t) * 384 + 256 + d4 * 4);
            K[t][d4 * 4] = kv.x; K[t][d4 * 4 + 1] = kv.y; K[t][d4 * 4 + 2] = kv.z; K[t][d4 * 4 + 3] = kv.w;
            V[t][d4 * 4] = vv.x; V[t][d4 * 4 + 1] = vv.y; V[t][d4 * 4 + 2] = vv.z; V[t][d4 * 4 + 3] = vv.w;
        }
        __syncthreads();
        for (int j = 0; j < 64; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s += q[d] * K[j][d];
            const float mn = fmaxf(m, s), corr = __expf(m - mn), p = __expf(s - mn);
            l = l * corr + p;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] = o[d] * corr + p * V[j][d];
            m = mn;
        }
    }
    const float il = 1.0f / l;
    float* op = out + ((size_t)img * n + i) * 128 + h * 32;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4)
        *reinterpret_cast<float4*>(op + d4 * 4) = make_float4(o[d4 * 4] * il, o[d4 * 4 + 1] * il, o[d4 * 4 + 2] * il, o[d4 * 4 + 3] * il);
    float* sp = stat + (((size_t)img * 4 + h) * n + i) * 3;
    sp[0] = m; sp[1] = l;
}
// dq (thread = query row i): P is recomputed from the row statistics; also leaves D_i for the key-side pass.
__global__ __launch_bounds__(64) void fu_attn_gen_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
                                                               float* __restrict__ stat, float* __restrict__ dqkv, int n) {
    __shared__ float K[64][33], V[64][33];
    const int h = blockIdx.x, img = blockIdx.y, t = threadIdx.x, i = blockIdx.z * 64 + t;
    const float sc = 0.17677669529663687f;
    const float* base = qkv + (size_t)img * n * 384 + h * 32;
    float q[32], g[32], dq[32];
    float D = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)i * 384 + d4 * 4);
        const float4 gv = *reinterpret_cast<const float4*>(dout + ((size_t)img * n + i) * 128 + h * 32 + d4 * 4);
        const float4 ov = *reinterpret_cast<const float4*>(out + ((size_t)img * n + i) * 128 + h * 32 + d4 * 4);
        q[d4 * 4] = v.x * sc; q[d4 * 4 + 1] = v.y * sc; q[d4 * 4 + 2] = v.z * sc; q[d4 * 4 + 3] = v.w * sc;
        g[d4 * 4] = gv.x; g[d4 * 4 + 1] = gv.y; g[d4 * 4 + 2] = gv.z; g[d4 * 4 + 3] = gv.w;
        D += (gv.x * ov.x + gv.y * ov.y) + (gv.z * ov.z + gv.w * ov.w);
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) dq[d] = 0.f;
    float* sp = stat + (((size_t)img * 4 + h) * n + i) * 3;
    const float m = sp[0], il = 1.0f / sp[1];
    sp[2] = D;
    for (int c = 0; c < n; c += 64) {
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)(c + t) * 384 + 128 + d4 * 4);
            const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)(c + t) * 384 + 256 + d4 * 4);
            K[t][d4 * 4] = kv.x; K[t][d4 * 4 + 1] = kv.y; K[t][d4 * 4 + 2] = kv.z; K[t][d4 * 4 + 3] = kv.w;
            V[t][d4 * 4] = vv.x; V[t][d4 * 4 + 1] = vv.y; V[t][d4 * 4 + 2] = vv.z; V[t][d4 * 4 + 3] = vv.w;
        }
        __syncthreads();
        for (int j = 0; j < 64; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += q[d] * K[j][d]; dp += g[d] * V[j][d]; }
            const float ds = __expf(s - m) * il * (dp - D);
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] += ds * K[j][d];
        }
    }
    float* op = dqkv + ((size_t)img * n + i) * 384 + h * 32;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4)
        *reinterpret_cast<float4*>(op + d4 * 4) = make_float4(dq[d4 * 4] * sc, dq[d4 * 4 + 1] * sc, dq[d4 * 4 + 2] * sc, dq[d4 * 4 + 3] * sc);
}
// dk, dv (thread = key row j): queries, their gradients and row statistics streamed through LDS.
__global__ __launch_bounds__(64) void fu_attn_gen_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ stat,
                                                                float* __restrict__ dqkv, int n) {
    __shared__ float Q[64][33], G[64][33], St[64][3];
    const int h = blockIdx.x, img = blockIdx.y, t = threadIdx.x, j = blockIdx.z * 64 + t;
    const float sc = 0.17677669529663687f;
    const float* base = qkv + (size_t)img * n * 384 + h * 32;
    float k[32], v[32], dk[32], dv[32];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const float4 kv = *reinterpret_cast<const float4*>(base + (size_t)j * 384 + 128 + d4 * 4);
        const float4 vv = *reinterpret_cast<const float4*>(base + (size_t)j * 384 + 256 + d4 * 4);
        k[d4 * 4] = kv.x; k[d4 * 4 + 1] = kv.y; k[d4 * 4 + 2] = kv.z; k[d4 * 4 + 3] = kv.w;
        v[d4 * 4] = vv.x; v[d4 * 4 + 1] = vv.y; v[d4 * 4 + 2] = vv.z; v[d4 * 4 + 3] = vv.w;
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int c = 0; c < n; c += 64) {
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const float4 qv = *reinterpret_cast<const float4*>(base + (size_t)(c + t) * 384 + d4 * 4);
            const float4 gv = *reinterpret_cast<const float4*>(dout + ((size_t)img * n + c + t) * 128 + h * 32 + d4 * 4);
            Q[t][d4 * 4] = qv.x * sc; Q[t][d4 * 4 + 1] = qv.y * sc; Q[t][d4 * 4 + 2] = qv.z * sc; Q[t][d4 * 4 + 3] = qv.w * sc;
            G[t][d4 * 4] = gv.x; G[t][d4 * 4 + 1] = gv.y; G[t][d4 * 4 + 2] = gv.z; G[t][d4 * 4 + 3] = gv.w;
        }
        {
            const float* sp = stat + (((size_t)img * 4 + h) * n + c + t) * 3;
            St[t][0] = sp[0]; St[t][1] = 1.0f / sp[1]; St[t][2] = sp[2];
        }
        __syncthreads();
        for (int i = 0; i < 64; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += Q[i][d] * k[d]; dp += G[i][d] * v[d]; }
            const float p = __expf(s - St[i][0]) * St[i][1];
            const float ds = p * (dp - St[i][2]);
#pragma unroll
            for (int d = 0; d < 32; ++d) { dv[d] += p * G[i][d]; dk[d] += ds * Q[i][d]; }      // (Q carries the scale)
        }
    }
    float* op = dqkv + ((size_t)img * n + j) * 384 + h * 32;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        *reinterpret_cast<float4*>(op + 128 + d4 * 4) = make_float4(dk[d4 * 4], dk[d4 * 4 + 1], dk[d4 * 4 + 2], dk[d4 * 4 + 3]);
        *reinterpret_cast<float4*>(op + 256 + d4 * 4) = make_float4(dv[d4 * 4], dv[d4 * 4 + 1], dv[d4 * 4 + 2], dv[d4 * 4 + 3]);
    }
}

}  // namespace cindm
