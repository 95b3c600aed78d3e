// kernels_vq.hip -- codebook searches of the path.
//
//  (1) rvq_ema_forward: the reference's own ResidualVQEMA (K <= 512, D = 96): argmax_k(res.e_k - 0.5||e_k||^2)
//      per book with the residual carried between books.  One block owns TOKS tokens for ALL books; per book the
//      codebook is pinned in LDS (transposed [D][K] image, at most 256 codes at a time = 96 KiB for D = 96), every
//      lane scores its own codes for the wave's 2 tokens at once and the per-token winner is a wavefront arg-max reduction
//      (lowest index on ties, like torch.argmax on CPU).
//  (2) rvq_assign / ema_update: ResidualVQEMA.ema_step (deterministic token-order sums).
//  (3) dac_rvq: the 32-stage DAC residual quantiser (K = 1024, Dc = 8) fused into ONE launch: in_proj, L2
//      normalisation, cosine search against the LDS-resident normalised codebook, straight-through out_proj
//      and residual update, all stages for a block's 16 tokens with the residual held in registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "det_math.hpp"
#include "kernels_small.hpp"

namespace mvq {

// arg-max combine: keep the larger score, the lower index on ties
__device__ __forceinline__ void amax_combine(float& s, int& i, float os, int oi)
{
    if (os > s || (os == s && oi < i)) { s = os; i = oi; }
}

__device__ __forceinline__ void wave_argmax(float& s, int& i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(s, off);
        const int oi = __shfl_xor(i, off);
        amax_combine(s, i, os, oi);
    }
}

// dst[0..n) = src[0..n) with 8 x 16-byte loads in flight per thread (n % 4 == 0, both 16-byte aligned)
__device__ __forceinline__ void copy_to_lds_f4(float* dst, const float* __restrict__ src, int n, int tid)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int nvec = n >> 2;
    for (int base = 0; base < nvec; base += 256 * 8) {
        v4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            r[u] = *reinterpret_cast<const v4*>(src + 4 * (size_t)(i < nvec ? i : nvec - 1));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            if (i < nvec) *reinterpret_cast<v4*>(dst + 4 * (size_t)i) = r[u];
        }
    }
}

constexpr int RVQ_KH = 256;      // codes resident in LDS at a time

// Staging order of a [kh][dv] slice of float4 code-row pieces.  The LDS images below have a pitch == 1 (mod 32), i.e. element
// (k, d) sits on bank (k + d) mod 32.  Taking the vectors in row-major order puts 24 consecutive pieces of ONE code on the
// lanes of a store group: banks k + 4m (+ component) repeat every 8 lanes -> 3-way conflicts on every ds_write_b32.  Instead
// the 32 lanes of a group take a 4-code x 8-piece patch (k = 4*kq + w/8, m = 8*cb + w%8): banks k + 4m cover all 32, and each
// code's 8 pieces are still one full 128-byte line of its global row.  Needs dv % 8 == 0 and kh % 4 == 0 (D = 96: dv = 24).
__device__ __forceinline__ void rvq_stage_map(int i, int dv, bool patch, int& k, int& m)
{
    if (patch) {
        const int p = i >> 5, w = i & 31, nb8 = dv >> 3;
        const int kq = p / nb8, cb = p - kq * nb8;
        k = kq * 4 + (w >> 3);
        m = cb * 8 + (w & 7);
    } else {
        k = i / dv;
        m = i - k * dv;
    }
}

// LDS: Et[D][KH+1] | hn[KH] | resT[D][TOKS] | qsT[D][TOKS] | best_s[TOKS] | best_i[TOKS]
template <int RVQ_TOKS>            // tokens per block: 8 (small batches: more blocks) or 32 (large: 4x less codebook staging)
__global__ __launch_bounds__(256) void rvq_ema_forward_kernel(
    const float* __restrict__ z, const float* __restrict__ books, float* __restrict__ q_out,
    int32_t* __restrict__ idx_out, int B, int D, int T, int nb, int K, int update_residual)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int KHP = RVQ_KH + 1;
    float* Et = sm;
    float* hn = Et + (size_t)D * KHP;
    float* resT = hn + RVQ_KH;
    float* qsT = resT + (size_t)D * RVQ_TOKS;
    float* best_s = qsT + (size_t)D * RVQ_TOKS;
    int* best_i = reinterpret_cast<int*>(best_s + RVQ_TOKS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = B * T;
    const int n0 = blockIdx.x * RVQ_TOKS;

    // load this block's tokens: resT[d][tok]
    for (int i = tid; i < D * RVQ_TOKS; i += 256) {
        const int tok = i % RVQ_TOKS, d = i / RVQ_TOKS;
        const int n = n0 + tok;
        float v = 0.0f;
        if (n < N) { const int b = n / T, t = n - b * T; v = z[((size_t)b * D + d) * T + t]; }
        resT[d * RVQ_TOKS + tok] = v;
        qsT[d * RVQ_TOKS + tok] = 0.0f;
    }

    for (int bk = 0; bk < nb; ++bk) {
        const float* emb = books + (size_t)bk * K * D;
        for (int k0 = 0; k0 < K; k0 += RVQ_KH) {
            const int kh = K - k0 < RVQ_KH ? K - k0 : RVQ_KH;
            __syncthreads();
            // stage transposed codebook slice: global row-major (coalesced) -> Et[d][k] (odd pitch: conflict-free)
            {
                typedef float v4 __attribute__((ext_vector_type(4)));
                const int dv = D >> 2;                                  // float4 per code row (D % 4 == 0)
                const int nvec = kh * dv;
                const bool patch = (dv & 7) == 0 && (kh & 3) == 0;
                const float* src = emb + (size_t)k0 * D;
                for (int base = 0; base < nvec; base += 256 * 8) {
                    v4 r[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = base + u * 256 + tid;
                        int k, m;
                        rvq_stage_map(i < nvec ? i : nvec - 1, dv, patch, k, m);
                        r[u] = *reinterpret_cast<const v4*>(src + 4 * ((size_t)k * dv + m));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = base + u * 256 + tid;
                        if (i < nvec) {
                            int k, m;
                            rvq_stage_map(i, dv, patch, k, m);
                            const int d4 = m * 4;
                            Et[(d4 + 0) * KHP + k] = r[u].x; Et[(d4 + 1) * KHP + k] = r[u].y;
                            Et[(d4 + 2) * KHP + k] = r[u].z; Et[(d4 + 3) * KHP + k] = r[u].w;
                        }
                    }
                }
            }
            __syncthreads();
            for (int k = tid; k < kh; k += 256) {
                float s = 0.0f;
                for (int d = 0; d < D; ++d) { const float e = Et[d * KHP + k]; s = dfma(e, e, s); }
                hn[k] = 0.5f * s;
            }
            __syncthreads();
            // each wave scores its tokens, 2 at a time, against the resident codes (lane = code)
            for (int tp = 0; tp < RVQ_TOKS / 8; ++tp) {
                const int tok0 = wave * (RVQ_TOKS / 4) + tp * 2;
                float bs[2]; int bi[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { bs[u] = -__builtin_inff(); bi[u] = 0x7fffffff; }
                for (int kk = lane; kk < kh; kk += 64) {
                    float dot0 = 0.0f, dot1 = 0.0f;
                    for (int d = 0; d < D; d += 4) {                // D % 4 == 0; operands first, then the chains
                        float e[4]; float2 r[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            e[u] = Et[(d + u) * KHP + kk];
                            r[u] = *reinterpret_cast<const float2*>(resT + (d + u) * RVQ_TOKS + tok0);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { dot0 = dfma(r[u].x, e[u], dot0); dot1 = dfma(r[u].y, e[u], dot1); }
                    }
                    const float h = hn[kk];
                    const float sc0 = dot0 - h, sc1 = dot1 - h;
                    if (sc0 > bs[0]) { bs[0] = sc0; bi[0] = k0 + kk; }
                    if (sc1 > bs[1]) { bs[1] = sc1; bi[1] = k0 + kk; }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    wave_argmax(bs[u], bi[u]);
                    if (lane == 0) {
                        if (k0 == 0) { best_s[tok0 + u] = bs[u]; best_i[tok0 + u] = bi[u]; }
                        else {
                            float cs = best_s[tok0 + u]; int ci = best_i[tok0 + u];
                            amax_combine(cs, ci, bs[u], bi[u]);
                            best_s[tok0 + u] = cs; best_i[tok0 + u] = ci;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // gather + straight-through sum + residual update (reads the row-major global book: contiguous row)
        for (int i = tid; i < D * RVQ_TOKS; i += 256) {
            const int tok = i % RVQ_TOKS, d = i / RVQ_TOKS;          // token fastest: conflict-free LDS rows (the code rows are L2 hits)
            int id = best_i[tok];
            if (id < 0 || id >= K) id = 0;                       // all-NaN scores: defined, in-range gather
            const float q = emb[(size_t)id * D + d];
            const float r = resT[d * RVQ_TOKS + tok];
            const float qs = qsT[d * RVQ_TOKS + tok];
            qsT[d * RVQ_TOKS + tok] = (qs + (q - r)) + r;
            if (update_residual) resT[d * RVQ_TOKS + tok] = r - q;
        }
        if (idx_out && tid < RVQ_TOKS && n0 + tid < N) {
            int id = best_i[tid];
            if (id < 0 || id >= K) id = 0;
            idx_out[(size_t)bk * N + n0 + tid] = id;
        }
    }
    __syncthreads();
    if (q_out) {
        for (int i = tid; i < D * RVQ_TOKS; i += 256) {
            const int tok = i % RVQ_TOKS, d = i / RVQ_TOKS;
            const int n = n0 + tok;
            if (n < N) { const int b = n / T, t = n - b * T; q_out[((size_t)b * D + d) * T + t] = qsT[d * RVQ_TOKS + tok]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Throughput form of the same search: the scores of 32 tokens against 32 codes at a time are one 32x32 MFMA tile
// (M = codes, N = tokens, K = the D dimensions walked two at a time), which is exactly the d-ascending fp32 fma chain of the
// scalar kernel, so indices and sums stay bit-identical.  With M = codes every lane owns ONE token column: the running
// arg-max over codes is a per-lane loop over its 16 accumulator rows (ascending code index, strict '>' keeps the lowest on
// ties), then one exchange with the other lane half and one 4-way combine across the waves.
// LDS: Es[KH][D+1] (natural code-major rows, odd pitch) | hn[KH] | resT[D][32] | qsT[D][32] | ws[4][32] | wi[4][32] | best
// ------------------------------------------------------------------------------------------------
typedef float rvq_f32x16 __attribute__((ext_vector_type(16)));

// Round 4: (1) `tpb` live tokens per block (16 when 32 would leave half of the 256 CUs without a block: an AR chunk of 256 segments
// is 4 096 tokens = 128 blocks of 32), the tile keeps its 32 MFMA columns and the dead ones are masked; (2) the 0.5 |e|^2 chain and
// the MFMA operand fetch issue their LDS reads a batch ahead of the dependent arithmetic (both were one exposed LDS latency per
// element / per k-step).  Same chains, same order: indices and sums bit-identical (tests/test_gpu_parity_ops.py).
__global__ __launch_bounds__(256) void rvq_ema_forward_mfma_kernel(
    const float* __restrict__ z, const float* __restrict__ books, float* __restrict__ q_out,
    int32_t* __restrict__ idx_out, int B, int D, int T, int nb, int K, int update_residual, int tpb)
{
    constexpr int TOKS = 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int DP = D + 1;
    float* Es = sm;                                   // [RVQ_KH][DP]
    float* hn = Es + (size_t)RVQ_KH * DP;             // [RVQ_KH]
    float* resT = hn + RVQ_KH;                        // [D][32]
    float* qsT = resT + (size_t)D * TOKS;             // [D][32]
    float* ws = qsT + (size_t)D * TOKS;               // [4][32]
    int* wi = reinterpret_cast<int*>(ws + 4 * TOKS);  // [4][32]
    float* best_s = reinterpret_cast<float*>(wi + 4 * TOKS);
    int* best_i = reinterpret_cast<int*>(best_s + TOKS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int N = B * T;
    const int n0 = blockIdx.x * tpb;
    const int n_end = n0 + tpb < N ? n0 + tpb : N;                    // tokens [n0, n_end) belong to this block

    for (int i = tid; i < D * TOKS; i += 256) {
        const int tok = i % TOKS, d = i / TOKS;
        const int n = n0 + tok;
        float v = 0.0f;
        if (n < n_end) { const int b = n / T, t = n - b * T; v = z[((size_t)b * D + d) * T + t]; }
        resT[d * TOKS + tok] = v;
        qsT[d * TOKS + tok] = 0.0f;
    }

    for (int bk = 0; bk < nb; ++bk) {
        const float* emb = books + (size_t)bk * K * D;
        for (int k0 = 0; k0 < K; k0 += RVQ_KH) {
            const int kh = K - k0 < RVQ_KH ? K - k0 : RVQ_KH;              // multiple of 32 (checked by the launcher)
            __syncthreads();
            {   // stage the code rows: global row-major (coalesced 16-byte loads) -> Es[k][d], pitch D+1
                typedef float v4 __attribute__((ext_vector_type(4)));
                const int dv = D >> 2;
                const int nvec = kh * dv;
                const bool patch = (dv & 7) == 0;                           // kh is a multiple of 32 here
                const float* src = emb + (size_t)k0 * D;
                for (int base = 0; base < nvec; base += 256 * 8) {
                    v4 r[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = base + u * 256 + tid;
                        int k, m;
                        rvq_stage_map(i < nvec ? i : nvec - 1, dv, patch, k, m);
                        r[u] = *reinterpret_cast<const v4*>(src + 4 * ((size_t)k * dv + m));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = base + u * 256 + tid;
                        if (i < nvec) {
                            int k, m;
                            rvq_stage_map(i, dv, patch, k, m);
                            const int d4 = m * 4;
                            float* dst = Es + (size_t)k * DP + d4;
                            dst[0] = r[u].x; dst[1] = r[u].y; dst[2] = r[u].z; dst[3] = r[u].w;
                        }
                    }
                }
            }
            __syncthreads();
            for (int k = tid; k < kh; k += 256) {
                float s = 0.0f;
                const float* er = Es + (size_t)k * DP;
                int d = 0;
                for (; d + 16 <= D; d += 16) {                            // 16 reads in flight, then the ordered chain
                    float e[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) e[u] = er[d + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) s = dfma(e[u], e[u], s);
                }
                for (; d < D; ++d) { const float e = er[d]; s = dfma(e, e, s); }
                hn[k] = 0.5f * s;
            }
            __syncthreads();
            // this wave's 32-code row blocks: mb = wave, wave + 4, ... (ascending, so the strict compare keeps the lowest index)
            float bs = -__builtin_inff(); int bi = 0x7fffffff;
            // two row blocks at a time: two independent accumulator chains keep the MFMA pipe issuing back to back (a single
            // chain waits ~80 cycles per step for its own previous result)
            for (int mb = wave; mb * 32 < kh; mb += 8) {
                const bool two = (mb + 4) * 32 < kh;
                rvq_f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
                const float* ap0 = Es + (size_t)(mb * 32 + l31) * DP + h;
                const float* ap1 = Es + (size_t)((two ? mb + 4 : mb) * 32 + l31) * DP + h;
                const float* bp = resT + h * TOKS + l31;
                // K index = dimension, two per step; operands of 4 steps (12 LDS reads) are fetched ahead of their 8 MFMAs
                const int steps = D / 2;
                int sx = 0;
                for (; sx + 4 <= steps; sx += 4) {
                    float bq[4], a0[4], a1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bq[u] = bp[2 * (sx + u) * TOKS]; a0[u] = ap0[2 * (sx + u)]; a1[u] = ap1[2 * (sx + u)]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bq[u], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], bq[u], acc1, 0, 0, 0);
                    }
                }
                for (; sx < steps; ++sx) {
                    const float b = bp[2 * sx * TOKS];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap0[2 * sx], b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap1[2 * sx], b, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float sc = acc0[r] - hn[code];
                    if (sc > bs) { bs = sc; bi = k0 + code; }
                }
                if (two) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int code = (mb + 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float sc = acc1[r] - hn[code];
                        if (sc > bs) { bs = sc; bi = k0 + code; }
                    }
                }
            }
            {   // the other lane half holds the other 16 rows of every block for the same token
                const float os = __shfl_xor(bs, 32);
                const int oi = __shfl_xor(bi, 32);
                amax_combine(bs, bi, os, oi);
            }
            if (h == 0) { ws[wave * TOKS + l31] = bs; wi[wave * TOKS + l31] = bi; }
            __syncthreads();
            if (tid < TOKS) {
                float cs = ws[tid]; int ci = wi[tid];
#pragma unroll
                for (int w = 1; w < 4; ++w) amax_combine(cs, ci, ws[w * TOKS + tid], wi[w * TOKS + tid]);
                if (k0 != 0) amax_combine(cs, ci, best_s[tid], best_i[tid]);
                best_s[tid] = cs; best_i[tid] = ci;
            }
        }
        __syncthreads();
        for (int i = tid; i < D * TOKS; i += 256) {
            const int tok = i % TOKS, d = i / TOKS;                  // token fastest: conflict-free LDS rows (the code rows are L2 hits)
            int id = best_i[tok];
            if (id < 0 || id >= K) id = 0;
            const float q = emb[(size_t)id * D + d];
            const float r = resT[d * TOKS + tok];
            const float qs = qsT[d * TOKS + tok];
            qsT[d * TOKS + tok] = (qs + (q - r)) + r;
            if (update_residual) resT[d * TOKS + tok] = r - q;
        }
        if (idx_out && tid < TOKS && n0 + tid < n_end) {
            int id = best_i[tid];
            if (id < 0 || id >= K) id = 0;
            idx_out[(size_t)bk * N + n0 + tid] = id;
        }
    }
    __syncthreads();
    if (q_out) {
        for (int i = tid; i < D * TOKS; i += 256) {
            const int tok = i % TOKS, d = i / TOKS;
            const int n = n0 + tok;
            if (n < n_end) { const int b = n / T, t = n - b * T; q_out[((size_t)b * D + d) * T + t] = qsT[d * TOKS + tok]; }
        }
    }
}

static hipError_t launch_rvq_mfma(const float* z, const float* books, float* q_out, int32_t* idx_out,
                                  int B, int D, int T, int nb, int K, int update_residual, hipStream_t s)
{
    const int N = B * T;
    const size_t lds = ((size_t)RVQ_KH * (D + 1) + RVQ_KH + 2 * (size_t)D * 32 + 8 * 32 + 2 * 32) * sizeof(float);
    static BigLdsOptIn opt;
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(rvq_ema_forward_mfma_kernel)); e != hipSuccess) return e;
    const int tpb = (N + 31) / 32 < 256 ? 16 : 32;        // every CU gets a block before any block gets 32 tokens
    hipLaunchKernelGGL(rvq_ema_forward_mfma_kernel, dim3((N + tpb - 1) / tpb), dim3(256), lds, s, z, books, q_out, idx_out, B, D, T,
                       nb, K, update_residual, tpb);
    return hipGetLastError();
}

template <int TOKS>
static hipError_t launch_rvq_t(const float* z, const float* books, float* q_out, int32_t* idx_out,
                               int B, int D, int T, int nb, int K, int update_residual, hipStream_t s)
{
    const int N = B * T;
    const size_t lds = ((size_t)D * (RVQ_KH + 1) + RVQ_KH + 2 * (size_t)D * TOKS + 2 * TOKS) * sizeof(float);
    auto kern = rvq_ema_forward_kernel<TOKS>;
    static BigLdsOptIn opt;                           // per (instantiation, device)
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((N + TOKS - 1) / TOKS), dim3(256), lds, s, z, books, q_out, idx_out, B, D, T, nb, K, update_residual);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Latency form (a handful of tokens: one AR chunk of one segment is 16): ONE BLOCK PER TOKEN, thread = code.  Nothing is
// staged: every thread walks its code rows straight out of L2 (the books are read by every block, 196 KB each at K = 512)
// while the token's residual sits in LDS, so a 16-token chunk occupies 16 CUs instead of 2 and a book costs a row walk
// instead of two 98 KB LDS stagings per block.  Same chains as the other forms: score = (sum_d r_d * e_d, d ascending)
// - 0.5 * (sum_d e_d^2, d ascending); strict '>' over ascending codes, lowest index on ties.
// LDS: res[D] | qs[D] | ws[4] | wi[4]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rvq_ema_forward_token_kernel(
    const float* __restrict__ z, const float* __restrict__ books, float* __restrict__ q_out,
    int32_t* __restrict__ idx_out, int B, int D, int T, int nb, int K, int update_residual)
{
    __shared__ float res[128], qs[128], ws[4];
    __shared__ int wi[4];
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x, N = B * T;
    const int b = n / T, t = n - b * T;
    if (tid < D) { res[tid] = z[((size_t)b * D + tid) * T + t]; qs[tid] = 0.0f; }
    for (int bk = 0; bk < nb; ++bk) {
        const float* emb = books + (size_t)bk * K * D;
        __syncthreads();
        float bs = -__builtin_inff(); int bi = 0x7fffffff;
        for (int k = tid; k < K; k += 256) {
            const float* row = emb + (size_t)k * D;
            float dot = 0.0f, hs = 0.0f;
            for (int d = 0; d < D; d += 16) {                      // D % 4 == 0; up to four 16-byte row loads in flight
                v4 e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = (d + 4 * u < D) ? *reinterpret_cast<const v4*>(row + d + 4 * u) : v4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (d + 4 * u < D) {
                        const float* r = res + d + 4 * u;
                        dot = dfma(r[0], e[u].x, dot); dot = dfma(r[1], e[u].y, dot); dot = dfma(r[2], e[u].z, dot); dot = dfma(r[3], e[u].w, dot);
                        hs = dfma(e[u].x, e[u].x, hs); hs = dfma(e[u].y, e[u].y, hs); hs = dfma(e[u].z, e[u].z, hs); hs = dfma(e[u].w, e[u].w, hs);
                    }
            }
            const float sc = dot - 0.5f * hs;
            if (sc > bs) { bs = sc; bi = k; }
        }
        wave_argmax(bs, bi);
        if (lane == 0) { ws[wave] = bs; wi[wave] = bi; }
        __syncthreads();
        float cs = ws[0]; int id = wi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) amax_combine(cs, id, ws[w], wi[w]);
        if (id < 0 || id >= K) id = 0;                           // all-NaN scores: defined, in-range gather
        if (tid < D) {
            const float q = emb[(size_t)id * D + tid];
            const float r = res[tid];
            qs[tid] = (qs[tid] + (q - r)) + r;
            if (update_residual) res[tid] = r - q;
        }
        if (idx_out && tid == 0) idx_out[(size_t)bk * N + n] = id;
    }
    __syncthreads();
    if (q_out && tid < D) q_out[((size_t)b * D + tid) * T + t] = qs[tid];
}

// ------------------------------------------------------------------------------------------------
// Latency form, round 5: ONE BLOCK PER TOKEN, ONE THREAD PER CODE (K <= 512), code rows through LDS in two halves.
// profiles/r05_kernel_stats_B1_encode_before.csv: the form above costs 67 us per 16-token chunk -- 8.4 us per book, of which the
// 96-step chains are ~0.4.  The rest is the access pattern: a thread walking its own 384-byte row makes every wave-load touch 64
// different cache lines (~64 cycles of the CU's address pipe per instruction, 24 instructions x 8 waves per book), whether the
// loads are issued one by one or all at once (the first round-5 cut held the rows in registers: 58 us).  Here the block fetches
// the book COOPERATIVELY -- consecutive lanes take consecutive 16-byte pieces of a row, a wave-load covers five rows in ten
// lines -- half a row (48 dimensions, 104 KB of LDS at pitch 52: conflict-free ds_read_b128 per code) at a time, the second half
// and the next book's first half in flight while the chains of the current half run; a thread keeps the row it scored in
// registers, so the winner publishes it without a gather.  Same chains as every other form: dot = sum_d r_d e_d and
// hs = sum_d e_d^2, d ascending from +0; score = dot - 0.5 hs; strict '>' over ascending codes, lowest index on ties.
// ------------------------------------------------------------------------------------------------
template <int DV>                                       // 16-byte pieces per code row: D = 4 DV, fetched in two halves of DV / 2
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))     // one block per CU (104 KB of LDS): 256 VGPRs per lane
void rvq_ema_forward_rows_kernel(
    const float* __restrict__ z, const float* __restrict__ books, float* __restrict__ q_out,
    int32_t* __restrict__ idx_out, int B, int T, int nb, int K, int update_residual)
{
    constexpr int D = 4 * DV, HV = DV / 2, PITCH = 4 * HV + 4;
    static_assert(DV % 2 == 0, "two equal halves");
    typedef float v4 __attribute__((ext_vector_type(4)));
    // dynamic LDS only (the opt-in above 64 KB covers dynamic memory): res[D] | qs[D] | qrow[D] | ws[8] | wi[8] | rows[K][PITCH]
    extern __shared__ __attribute__((aligned(16))) float lds_rows[];
    float* const res = lds_rows;
    float* const qs = res + D;
    float* const qrow = qs + D;
    float* const ws = qrow + D;
    int* const wi = reinterpret_cast<int*>(ws + 8);
    float* const rows = ws + 16;                                    // [K][PITCH]: one half of every code row (16-byte aligned: 3 D + 16 floats)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x, N = B * T;
    const int b = n / T, t = n - b * T;
    const bool has = tid < K;
    const int npieces = K * HV;
    if (tid < D) { res[tid] = z[((size_t)b * D + tid) * T + t]; qs[tid] = 0.0f; }
    v4 st[HV];                                                      // this thread's pieces of the half being fetched
    unsigned goff[HV], loff[HV];                                    // their element offsets in a book / in the LDS image: the same for every half
    unsigned livem = 0;
#pragma unroll
    for (int u = 0; u < HV; ++u) {
        const int e0 = tid + 512 * u;
        const int ec = e0 < npieces ? e0 : npieces - 1;
        const int r = ec / HV, v = ec - r * HV;
        goff[u] = (unsigned)(r * D + 4 * v) * 4u;                   // bytes: uniform base + 32-bit lane offset, no 64-bit address per piece
        loff[u] = (unsigned)(r * PITCH + 4 * v) * 4u;
        livem |= e0 < npieces ? (1u << u) : 0u;
    }
    auto gfetch = [&](int bk, int h) __attribute__((always_inline)) {
        // uniform base kept in scalar registers (readfirstlane: the loop's strength reduction would otherwise carry one 64-bit
        // per-lane pointer per piece across the books -- 24 more registers, spilled to scratch)
        const uintptr_t bu = reinterpret_cast<uintptr_t>(books + (size_t)bk * K * D + h * 4 * HV);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bu), hi = __builtin_amdgcn_readfirstlane((unsigned)(bu >> 32));
        const char* base = reinterpret_cast<const char*>(((uintptr_t)hi << 32) | lo);
#pragma unroll
        for (int u = 0; u < HV; ++u) st[u] = *reinterpret_cast<const v4*>(base + goff[u]);
    };
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < HV; ++u)
            if ((livem >> u) & 1u) *reinterpret_cast<v4*>(reinterpret_cast<char*>(rows) + loff[u]) = st[u];
    };
    v4 e0[HV];                                                      // first half of the row this thread scores: kept, the winner publishes it
                                                                    // (the second half is still in LDS when the winner is known)
    if (nb > 0) gfetch(0, 0);
    for (int bk = 0; bk < nb; ++bk) {
        const float* myrow = rows + (has ? tid : 0) * PITCH;
        float dot = 0.0f, hs = 0.0f;
        auto half = [&](auto H) __attribute__((always_inline)) {   // H = 0 / 1 as a compile-time constant: e[] stays in registers
            constexpr int h = decltype(H)::value;
            if (h == 1) __syncthreads();                           // every chain of the first half has read its row
            lstore();
            __syncthreads();                                       // the half (and, first time round, res) is in place
            if (h == 0) gfetch(bk, 1);                             // in flight across the chains below
            else if (bk + 1 < nb) gfetch(bk + 1, 0);
            v4 ec[HV];
#pragma unroll
            for (int u = 0; u < HV; ++u) ec[u] = *reinterpret_cast<const v4*>(myrow + 4 * u);
            if (h == 0) {
#pragma unroll
                for (int u = 0; u < HV; ++u) e0[u] = ec[u];
            }
#pragma unroll
            for (int u = 0; u < HV; ++u) {
                const v4 r = *reinterpret_cast<const v4*>(res + 4 * (h * HV + u));      // every lane the same address: a broadcast read
                const v4 c = ec[u];
                dot = dfma(r.x, c.x, dot); dot = dfma(r.y, c.y, dot); dot = dfma(r.z, c.z, dot); dot = dfma(r.w, c.w, dot);
                hs = dfma(c.x, c.x, hs); hs = dfma(c.y, c.y, hs); hs = dfma(c.z, c.z, hs); hs = dfma(c.w, c.w, hs);
            }
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
        float bs = -__builtin_inff(); int bi = 0x7fffffff;
        if (has) {
            const float sc = dot - 0.5f * hs;
            if (sc > bs) { bs = sc; bi = tid; }
        }
        wave_argmax(bs, bi);
        if (lane == 0) { ws[wave] = bs; wi[wave] = bi; }
        __syncthreads();                                           // also: every chain of the second half has read its row
        float cs = ws[0]; int id = wi[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) amax_combine(cs, id, ws[w], wi[w]);
        if (id < 0 || id >= K) id = 0;                             // all-NaN scores: defined, in-range code
        if (tid == id) {
#pragma unroll
            for (int u = 0; u < HV; ++u) *reinterpret_cast<v4*>(qrow + 4 * u) = e0[u];
        }
        if (tid >= 64 && tid < 64 + 4 * HV) qrow[4 * HV + tid - 64] = rows[id * PITCH + tid - 64];      // second half: the winner's row in LDS
        __syncthreads();
        if (tid < D) {
            const float q = qrow[tid];
            const float r = res[tid];
            qs[tid] = (qs[tid] + (q - r)) + r;
            if (update_residual) res[tid] = r - q;
        }
        if (idx_out && tid == 0) idx_out[(size_t)bk * N + n] = id;
    }
    __syncthreads();
    if (q_out && tid < D) q_out[((size_t)b * D + tid) * T + t] = qs[tid];
}

hipError_t launch_rvq_ema_forward(const float* z, const float* books, float* q_out, int32_t* idx_out,
                                  int B, int D, int T, int nb, int K, int update_residual, hipStream_t s)
{
    const int N = B * T;
    if (N == 0) return hipSuccess;
    // a handful of tokens (latency regime): one block per token, nothing staged
    static const bool no_token_form = [] {                                          // A/B measurements; reported by mvq_build_flags()
        const bool o = getenv("MVQ_NO_TOKEN_RVQ") != nullptr;
        if (o) mvq::note_env_override(0x400 /* MVQ_BF_ENV_NO_TOKEN_RVQ */);
        return o;
    }();
    if (N <= 256 && D == 96 && K <= 512 && (reinterpret_cast<uintptr_t>(books) & 15) == 0 && !no_token_form) {   // CODE_DIM = 96 (Training/...5.py:68)
        // (Measured and not kept, both bit-equal: both halves of the NEXT book in flight -- does not fit 256 VGPRs beside the half row a
        // thread keeps; whole rows in LDS, 256 codes per pass, the next pass in flight -- 53 instead of 43 us per 16-token chunk: only
        // half the threads walk chains, and they are twice as long.  gpurun_out/j1.)
        auto kern = rvq_ema_forward_rows_kernel<24>;
        static BigLdsOptIn opt;                           // per device: 104 KB of dynamic LDS at K = 512
        if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(N), dim3(512), ((size_t)K * 52 + 3 * 96 + 16) * sizeof(float), s, z, books, q_out, idx_out, B, T, nb, K, update_residual);
        return hipGetLastError();
    }
    if (N <= 256 && D % 4 == 0 && D <= 128 && !no_token_form) {
        hipLaunchKernelGGL(rvq_ema_forward_token_kernel, dim3(N), dim3(256), 0, s, z, books, q_out, idx_out, B, D, T, nb, K, update_residual);
        return hipGetLastError();
    }
    // many tokens: the MFMA form (one block per 32 tokens, >= 32 blocks); few tokens: the scalar form with 8 tokens per block
    // (more blocks, and the codebook staging rather than the arithmetic is what a short chunk waits for)
    const size_t lds_mfma = ((size_t)RVQ_KH * (D + 1) + RVQ_KH + 2 * (size_t)D * 32 + 8 * 32 + 2 * 32) * sizeof(float);
    if (N >= 1024 && D % 4 == 0 && K % 32 == 0 && lds_mfma <= 160 * 1024)
        return launch_rvq_mfma(z, books, q_out, idx_out, B, D, T, nb, K, update_residual, s);
    return N >= 4096 ? launch_rvq_t<32>(z, books, q_out, idx_out, B, D, T, nb, K, update_residual, s)
                     : launch_rvq_t<8>(z, books, q_out, idx_out, B, D, T, nb, K, update_residual, s);
}

// ------------------------------------------------------------------------------------------------
// EMA update (ResidualVQEMA.ema_step, Training/compare_dacvsproposal_5.py:266-277) in O(N*D + N*K/…):
//   a stable counting sort of the token ids by assigned code, then one in-order sum per (code, dim) over ITS tokens only --
//   the same additions in the same (token) order as index_add_ / the oracle's loop, so the result is bit-identical, but a
//   (code, dim) thread no longer walks all N tokens (round 2: O(K*D*N), 23 ms at 8 x 256 segments).
//     ema_rows_kernel     z[B,D,T] -> X[N,D] (token rows contiguous: the summing threads read 4*D-byte rows)
//     ema_hist_kernel     per (book, 1024-token segment): code histogram (integer LDS atomics: order-free, exact)
//     ema_scan_kernel     per book: counts per code, exclusive offsets, per-(segment, code) write cursors
//     ema_scatter_kernel  per (book, segment): thread = code, walks the segment's ids in order -> perm (stable)
//     ema_apply_kernel    per (book, code): thread = dim, sum X[perm[j]][d] for j ascending; e <- decay*e + (1-decay)*mean
// idx[bk][N] comes from rvq_ema_forward_kernel(update_residual = 0): every book against the same un-residualised X.
// ------------------------------------------------------------------------------------------------
constexpr int EMA_SEG = 1024;

__global__ void ema_rows_kernel(const float* __restrict__ z, float* __restrict__ X, int D, int T)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, t = t0 + tx;
        tile[r][tx] = (d < D && t < T) ? z[((size_t)b * D + d) * T + t] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, d = d0 + tx;
        if (t < T && d < D) X[((size_t)b * T + t) * D + d] = tile[tx][r];
    }
}

__global__ void ema_hist_kernel(const int32_t* __restrict__ idx, int32_t* __restrict__ hist, int N, int K, int S)
{
    extern __shared__ int32_t h[];
    const int seg = blockIdx.x, bk = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const int n0 = seg * EMA_SEG, n1 = min(N, n0 + EMA_SEG);
    for (int n = n0 + threadIdx.x; n < n1; n += blockDim.x) atomicAdd(&h[idx[(size_t)bk * N + n]], 1);
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) hist[((size_t)bk * S + seg) * K + k] = h[k];
}

// hist[bk][seg][k] (counts) -> cursor[bk][seg][k] = offset[k] + #tokens of code k in earlier segments; count / offset per code
__global__ void ema_scan_kernel(int32_t* __restrict__ hist, int32_t* __restrict__ count, int32_t* __restrict__ offset, int K, int S)
{
    extern __shared__ int32_t cnt[];
    const int bk = blockIdx.x;
    int32_t* hb = hist + (size_t)bk * S * K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int run = 0;
        for (int sgm = 0; sgm < S; ++sgm) { const int c = hb[(size_t)sgm * K + k]; hb[(size_t)sgm * K + k] = run; run += c; }
        cnt[k] = run;
        count[(size_t)bk * K + k] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < K; ++k) { const int c = cnt[k]; cnt[k] = run; run += c; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int off = cnt[k];
        offset[(size_t)bk * K + k] = off;
        for (int sgm = 0; sgm < S; ++sgm) hb[(size_t)sgm * K + k] += off;
    }
}

__global__ void ema_scatter_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cursor, int32_t* __restrict__ perm,
                                   int N, int K, int S)
{
    __shared__ int32_t ids[EMA_SEG];
    const int seg = blockIdx.x, bk = blockIdx.y;
    const int n0 = seg * EMA_SEG, len = min(N, n0 + EMA_SEG) - n0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) ids[i] = idx[(size_t)bk * N + n0 + i];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        int w = cursor[((size_t)bk * S + seg) * K + k];
        for (int i = 0; i < len; ++i)                    // every lane reads the same id: an LDS broadcast
            if (ids[i] == k) perm[(size_t)bk * N + w++] = n0 + i;
    }
}

__global__ void ema_apply_kernel(const float* __restrict__ X, const int32_t* __restrict__ perm, const int32_t* __restrict__ count,
                                 const int32_t* __restrict__ offset, float* __restrict__ books, int N, int D, int K,
                                 float decay, float omd)
{
    const int k = blockIdx.x, bk = blockIdx.y, d = threadIdx.x;
    const int c = count[(size_t)bk * K + k];
    if (c == 0 || d >= D) return;
    const int32_t* pm = perm + (size_t)bk * N + offset[(size_t)bk * K + k];
    float sum = 0.0f;
    int j = 0;
    for (; j + 4 <= c; j += 4) {                          // four row loads in flight, the additions stay in token order
        const float x0 = X[(size_t)pm[j] * D + d], x1 = X[(size_t)pm[j + 1] * D + d];
        const float x2 = X[(size_t)pm[j + 2] * D + d], x3 = X[(size_t)pm[j + 3] * D + d];
        sum = sum + x0; sum = sum + x1; sum = sum + x2; sum = sum + x3;
    }
    for (; j < c; ++j) sum = sum + X[(size_t)pm[j] * D + d];
    float* p = books + ((size_t)bk * K + k) * D + d;
    const float mean = sum / ((float)c + 1e-9f);          // counts are exact integers in fp32 (< 2^24 tokens)
    *p = decay * (*p) + omd * mean;
}

size_t ema_update_scratch_bytes(int N, int nb, int K, int D)
{
    const size_t S = ((size_t)N + EMA_SEG - 1) / EMA_SEG;
    // hist/cursor [nb][S][K] + perm [nb][N] + count, offset [nb][K] (int32) + X [N][D] (fp32)
    return ((size_t)nb * S * K + (size_t)nb * N + 2 * (size_t)nb * K) * sizeof(int32_t) + (size_t)N * D * sizeof(float);
}

hipError_t launch_ema_update(const float* z, const int32_t* idx, float* books, void* scratch, int B, int D, int T, int nb, int K,
                             float decay, hipStream_t s)
{
    const int N = B * T;
    if ((long long)B * T >= (1 << 24) || (size_t)K * sizeof(int32_t) > 64 * 1024) return hipErrorInvalidValue;
    const int S = (N + EMA_SEG - 1) / EMA_SEG;
    const float omd = (float)(1.0 - (double)decay);
    int32_t* hist = reinterpret_cast<int32_t*>(scratch);
    int32_t* perm = hist + (size_t)nb * S * K;
    int32_t* count = perm + (size_t)nb * N;
    int32_t* offset = count + (size_t)nb * K;
    float* X = reinterpret_cast<float*>(offset + (size_t)nb * K);
    hipLaunchKernelGGL(ema_rows_kernel, dim3((T + 31) / 32, (D + 31) / 32, B), dim3(256), 0, s, z, X, D, T);
    hipLaunchKernelGGL(ema_hist_kernel, dim3(S, nb), dim3(256), (size_t)K * sizeof(int32_t), s, idx, hist, N, K, S);
    hipLaunchKernelGGL(ema_scan_kernel, dim3(nb), dim3(256), (size_t)K * sizeof(int32_t), s, hist, count, offset, K, S);
    hipLaunchKernelGGL(ema_scatter_kernel, dim3(S, nb), dim3(K >= 256 ? 256 : ((K + 63) / 64) * 64), 0, s, idx, hist, perm, N, K, S);
    hipLaunchKernelGGL(ema_apply_kernel, dim3(K, nb), dim3(((D + 63) / 64) * 64), 0, s, X, perm, count, offset, books, N, D, K, decay, omd);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// DAC residual VQ, all stages fused.  Block = 256 threads, 16 tokens; thread (grp, tok) OWNS the C/16 contiguous
// channels [grp*C/16, (grp+1)*C/16) of its token: residual and z_q accumulator live in registers for all stages.
//   in_proj  : 16 block-partial fma chains (one per thread, over its own channels) summed in block order
//              (the "blocked" order of the contract, see include/mvq.h), weights staged in LDS per stage
//   search   : L2-normalised cosine search against the LDS-resident normalised codebook, lane-group arg-max
//   out_proj : straight-through 8-long chain per owned channel, accumulate + residual update in registers
//   LDS: cbn[K][Dc] | cn2[K] | w[Dc*C + 512] | part[16][Dc][16] (red_s / red_i alias it) | ze[Dc][16] | pre[Dc][16]
// The stage weights are staged with a per-thread-group skew: the 4 thread groups of a wave read rows that are a multiple of
// 256 bytes apart in the natural images (in_proj: C/16 floats, out_proj: C/16*Dc floats), i.e. the same banks -- the in_proj
// image therefore gives every group CPT + 4 floats per row, the out_proj image CPT*Dc + 8 floats per group.
// ------------------------------------------------------------------------------------------------
constexpr int DQ_TOK = 16;

// dst[skewed(v)] = src[v] for the n/4 float4 pieces of a stage weight; piece v holds elements 4v .. 4v+3 of the natural image
//   IN  (in_proj  [Dc][C]):  (d, c)  -> d * (C + 16*4) + (c / CPT) * (CPT + 4) + c % CPT
//   OUT (out_proj [C][Dc]):  (c, d)  -> (c / CPT) * (CPT*Dc + 8) + (c % CPT) * Dc + d
template <int CPT, int DC, bool OUT>
__device__ __forceinline__ void copy_weights_skewed(float* dst, const float* __restrict__ src, int tid)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    constexpr int C = 16 * CPT;
    constexpr int nvec = DC * C / 4;
    for (int base = 0; base < nvec; base += 256 * 8) {
        v4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            r[u] = *reinterpret_cast<const v4*>(src + 4 * (size_t)(i < nvec ? i : nvec - 1));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + tid;
            if (i < nvec) {
                const int e = 4 * i;
                int o;
                if (OUT) { const int g = e / (CPT * DC); o = e + 8 * g; }
                else { const int d = e / C, c = e - d * C; o = d * (C + 64) + (c / CPT) * (CPT + 4) + (c % CPT); }
                *reinterpret_cast<v4*>(dst + o) = r[u];
            }
        }
    }
}

template <int CPT, int DC>    // channels per thread (C = 16*CPT), codebook dimension
__global__ __launch_bounds__(256, 2) void dac_rvq_kernel(
    const float* __restrict__ z, const float* __restrict__ in_w, const float* __restrict__ in_b,
    const float* __restrict__ cb, const float* __restrict__ out_w, const float* __restrict__ out_b,
    float* __restrict__ zq, int32_t* __restrict__ codes, float* __restrict__ latents,
    const int32_t* __restrict__ nq_item, int B, int T, int nq, int K,
    const float* __restrict__ cbn_pre, const float* __restrict__ cn2_pre)
{
    constexpr int C = 16 * CPT;
    constexpr int Dc = DC;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* cbn = sm;                                   // [K][Dc]
    float* cn2 = cbn + (size_t)K * Dc;                 // [K]
    float* wst = cn2 + K;                              // [Dc*C + 512] skewed stage weights (copy_weights_skewed)
    float* part = wst + (size_t)Dc * C + 512;          // [16][Dc][16]
    float* ze = part + 16 * Dc * DQ_TOK;               // [Dc][16]
    float* pre = ze + Dc * DQ_TOK;                     // [Dc][16]
    float* red_s = part;                               // [16][16]  (part is dead once ze is written: a barrier lies between)
    int* red_i = reinterpret_cast<int*>(red_s + 16 * DQ_TOK);
    static_assert(Dc * 64 <= 512 && 16 * 8 <= 512 && 16 * Dc * DQ_TOK >= 2 * 16 * DQ_TOK, "skew / alias sizes");

    const int tid = threadIdx.x;
    const int tok = tid & 15;
    const int grp = tid >> 4;                          // 0..15
    const int N = B * T;
    const int n = blockIdx.x * DQ_TOK + tok;
    const bool live = n < N;
    const int bb = live ? n / T : 0, tt = live ? n - bb * T : 0;
    const int c0 = grp * CPT;
    // train-mode quantiser dropout: item bb sums only its first nq_item[bb] stages (every stage still runs: the residual,
    // codes and latents of later stages are produced exactly as upstream does under its mask)
    const int lim = (nq_item && live) ? nq_item[bb] : nq;

    float res[CPT], acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        acc[j] = 0.0f;
        res[j] = live ? z[((size_t)bb * C + c0 + j) * T + tt] : 0.0f;
    }

    for (int st = 0; st < nq; ++st) {
        const float* cbs = cb + (size_t)st * K * Dc;
        __syncthreads();                               // previous stage done with cbn / wst
        // F.normalize(codebook) and its squared norms are the same for every block and every call: with a PREPARED codebook
        // (mvq_dac_rvq_prepare_f32, one-off at model load: the identical divisions, done once) the block copies them; otherwise
        // it normalises its LDS copy in place (~120 vector instructions per code -- a fifth of the stage -- per block and stage)
        copy_to_lds_f4(cbn, cbn_pre ? cbn_pre + (size_t)st * K * Dc : cbs, K * Dc, tid);
        if (cbn_pre) for (int k = tid; k < K; k += 256) cn2[k] = cn2_pre[(size_t)st * K + k];
        copy_weights_skewed<CPT, DC, false>(wst, in_w + (size_t)st * Dc * C, tid);
        __syncthreads();
        // normalised codebook (in place) + squared norms
        if (!cbn_pre)
        for (int k = tid; k < K; k += 256) {
            float ss = 0.0f;
            for (int d = 0; d < Dc; ++d) { const float v = cbn[k * Dc + d]; ss = dfma(v, v, ss); }
            const float den = __builtin_fmaxf(__builtin_sqrtf(ss), 1e-12f);
            float s2 = 0.0f;
            for (int d = 0; d < Dc; ++d) { const float v = cbn[k * Dc + d] / den; cbn[k * Dc + d] = v; s2 = dfma(v, v, s2); }
            cn2[k] = s2;
        }
        // in_proj block partials: this thread's channels, every codebook dimension
#pragma unroll
        for (int d = 0; d < Dc; ++d) {
            const float* wr = wst + (size_t)d * (C + 64) + grp * (CPT + 4);
            float p = 0.0f;
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + j);
                p = dfma(w4.x, res[j], p); p = dfma(w4.y, res[j + 1], p);
                p = dfma(w4.z, res[j + 2], p); p = dfma(w4.w, res[j + 3], p);
            }
            part[(grp * Dc + d) * DQ_TOK + tok] = p;
        }
        __syncthreads();
        if (grp < Dc) {                                // d = grp: sum the 16 block partials in block order, + bias
            float a = part[(0 * Dc + grp) * DQ_TOK + tok];
#pragma unroll
            for (int g = 1; g < 16; ++g) a = a + part[(g * Dc + grp) * DQ_TOK + tok];
            const float v = a + in_b[(size_t)st * Dc + grp];
            ze[grp * DQ_TOK + tok] = v;
            if (live) latents[((size_t)bb * nq * Dc + (size_t)st * Dc + grp) * T + tt] = v;
        }
        __syncthreads();
        // out_proj weights [C][Dc] replace the in_proj weights (every in_proj read is behind the barrier above)
        copy_weights_skewed<CPT, DC, true>(wst, out_w + (size_t)st * C * Dc, tid);
        {
            // F.normalize over Dc (every thread of a token computes the same values)
            float ss = 0.0f;
            for (int d = 0; d < Dc; ++d) { const float v = ze[d * DQ_TOK + tok]; ss = dfma(v, v, ss); }
            const float den = __builtin_fmaxf(__builtin_sqrtf(ss), 1e-12f);
            float s2 = 0.0f;
            float ev[Dc];
#pragma unroll
            for (int d = 0; d < Dc; ++d) { ev[d] = ze[d * DQ_TOK + tok] / den; s2 = dfma(ev[d], ev[d], s2); }
            // search: this thread scans codes k = grp + 16*j
            float bs = -__builtin_inff(); int bi = 0x7fffffff;
#pragma unroll 4
            for (int k = grp; k < K; k += 16) {
                const float* ck = cbn + k * Dc;
                float dot = 0.0f;
#pragma unroll
                for (int d = 0; d < Dc; ++d) dot = dfma(ev[d], ck[d], dot);
                const float dist = (s2 - 2.0f * dot) + cn2[k];
                const float sc = -dist;
                if (sc > bs) { bs = sc; bi = k; }
            }
            red_s[grp * DQ_TOK + tok] = bs;
            red_i[grp * DQ_TOK + tok] = bi;
        }
        __syncthreads();
        if (grp == 0) {
            float bs = red_s[tok]; int bi = red_i[tok];
            for (int g = 1; g < 16; ++g) amax_combine(bs, bi, red_s[g * DQ_TOK + tok], red_i[g * DQ_TOK + tok]);
            if (bi < 0 || bi >= K) bi = 0;
            if (live) codes[((size_t)bb * nq + st) * T + tt] = bi;
            const float* raw = cbs + (size_t)bi * Dc;
            for (int d = 0; d < Dc; ++d) { const float zv = ze[d * DQ_TOK + tok]; pre[d * DQ_TOK + tok] = zv + (raw[d] - zv); }
        }
        __syncthreads();
        {
            float pv[Dc];
#pragma unroll
            for (int d = 0; d < Dc; ++d) pv[d] = pre[d * DQ_TOK + tok];
            const float* ob = out_b + (size_t)st * C + c0;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const float* wr = wst + grp * (CPT * Dc + 8) + j * Dc;
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < Dc; ++d) a = dfma(wr[d], pv[d], a);
                const float zqi = a + ob[j];
                if (st < lim) acc[j] = acc[j] + zqi;
                res[j] = res[j] - zqi;
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) zq[((size_t)bb * C + c0 + j) * T + tt] = acc[j];
    }
}

// one thread per code of every stage: cbn = e / max(||e||, 1e-12), cn2 = sum_d cbn_d^2 (d ascending) -- the arithmetic of the
// in-kernel normalisation above, so a prepared codebook gives bit-identical codes and latents
__global__ __launch_bounds__(256) void dac_rvq_prepare_kernel(const float* __restrict__ cb, float* __restrict__ cbn,
                                                              float* __restrict__ cn2, int total, int Dc)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= total) return;
    const float* e = cb + (size_t)k * Dc;
    float* o = cbn + (size_t)k * Dc;
    float ss = 0.0f;
    for (int d = 0; d < Dc; ++d) { const float v = e[d]; ss = dfma(v, v, ss); }
    const float den = __builtin_fmaxf(__builtin_sqrtf(ss), 1e-12f);
    float s2 = 0.0f;
    for (int d = 0; d < Dc; ++d) { const float v = e[d] / den; o[d] = v; s2 = dfma(v, v, s2); }
    cn2[k] = s2;
}

hipError_t launch_dac_rvq_prepare(const float* cb, float* cbn, float* cn2, int nq, int K, int Dc, hipStream_t s)
{
    const int total = nq * K;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(dac_rvq_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, s, cb, cbn, cn2, total, Dc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Latency form of the DAC quantiser (round 5): TOKB tokens per block of 256 threads (launched with ONE), nothing staged in LDS,
// every operand of stage s + 1 requested at the top of stage s and held in registers.
// profiles/r05_kernel_stats_B1_encode_before.csv: dac_rvq_kernel takes 598 us for the 75 tokens of one segment -- 19 us per stage on
// 5 of the 256 CUs: a block copies ~100 KB of stage weights and codebook into LDS per stage, and a thread then walks 8 in_proj
// chains of 64, 64 codes and 64 out_proj channels one after another.  The contract fixes the CHAINS (in_proj: 16 block partials of
// C/16 channels per codebook dimension, added in block order; out_proj: an 8-long chain per channel; scores: an 8-long chain per
// code), not who runs them: here threads 0..127 take ONE partial chain each, and every thread K/256 codes and C/256 channels.
// Cuts measured, all bit-equal: two tokens per block with rows requested a PHASE ahead, 262 us (a phase is ~0.2 us of arithmetic, a
// round trip to L2 several times that); one token per block with a whole STAGE in flight, 210-240 us (a thread's share of a stage
// is 42 sixteen-byte rows = 168 VGPRs; a block per CU has 512 per lane, so two stages fit); four tokens per block, 396 us (see the
// launcher).  Clocks read inside the one-token form: of a stage's 6.5 us, 3.4 go by while the 49 loads of the next stage are
// ISSUED (the CU's vector-memory path delivers 64 B/clk and a block pulls 170 KB per stage through it) -- the stage is bound by that
// path and by the 8-cycle dependent issue of a wave that is alone on its SIMD.
// Needs the prepared codebook (mvq_dac_rvq_prepare_f32).  LDS: res[TOKB][C] | part[TOKB][16][8] | ze | pre | red
// ------------------------------------------------------------------------------------------------
template <int CPT, int KJ, int TOKB>    // C = 16 * CPT channels, K = 256 * KJ codes, Dc = 8, TOKB tokens per block
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dac_rvq_lat_kernel(
    const float* __restrict__ z, const float* __restrict__ in_w, const float* __restrict__ in_b,
    const float* __restrict__ cb, const float* __restrict__ out_w, const float* __restrict__ out_b,
    float* __restrict__ zq, int32_t* __restrict__ codes, float* __restrict__ latents,
    const int32_t* __restrict__ nq_item, int B, int T, int nq,
    const float* __restrict__ cbn_pre, const float* __restrict__ cn2_pre)
{
    constexpr int C = 16 * CPT, Dc = 8, K = 256 * KJ, CO = C / 256, IV = CPT / 4;
    static_assert(C % 256 == 0 && CPT % 4 == 0, "a thread owns C / 256 channels; in_proj pieces are 16-byte rows");
    static_assert(TOKB * Dc <= 64, "one wave sums the block partials of every token");
    typedef float v4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float res_s[TOKB][C];
    __shared__ float part[TOKB][16][Dc], ze[TOKB][Dc], pre[TOKB][Dc], red_s[TOKB][4];
    __shared__ int red_i[TOKB][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = B * T;
    int bbv[TOKB], ttv[TOKB], limv[TOKB];
    bool livev[TOKB];
#pragma unroll
    for (int k = 0; k < TOKB; ++k) {                               // a block's spare tokens repeat the LAST token and store nothing
        const int n = blockIdx.x * TOKB + k;
        livev[k] = n < N;
        const int nc = livev[k] ? n : N - 1;
        bbv[k] = nc / T; ttv[k] = nc - bbv[k] * T;
        limv[k] = nq_item ? nq_item[bbv[k]] : nq;
    }
    const int g = tid >> 3, d_in = tid & 7;                        // in_proj role (threads 0..127): block partial g of codebook dimension d_in
    const bool inproj = tid < 128;
    const int c0 = tid * CO;                                       // out_proj role: channels c0 .. c0 + CO - 1 of every token

    float resr[TOKB][CO], acc[TOKB][CO];
#pragma unroll
    for (int k = 0; k < TOKB; ++k)
#pragma unroll
        for (int j = 0; j < CO; ++j) {
            acc[k][j] = 0.0f;
            resr[k][j] = z[((size_t)bbv[k] * C + c0 + j) * T + ttv[k]];
            res_s[k][c0 + j] = resr[k][j];
        }
    struct StageOps {
        v4 iw[IV];                                                 // in_proj row piece of (stage, d_in, block g): CPT floats
        v4 sw[KJ][2], rw[KJ][2];                                   // codes k = tid + 256 j: normalised row (search), raw row (straight-through value)
        v4 ow[CO][2];                                              // out_proj rows of the owned channels
        float cn2v[KJ], obv[CO], inb;
    };
    auto fetch = [&](StageOps& P, int st) __attribute__((always_inline)) {
        {   // every thread requests a row piece (threads 128..255 the piece of thread - 128, unused): a load under a divergent branch
            // would make every later wait for P cover Pn's loads as well, i.e. undo the prefetch
            const float* wr = in_w + ((size_t)st * Dc + d_in) * C + (g & 15) * CPT;
#pragma unroll
            for (int u = 0; u < IV; ++u) P.iw[u] = *reinterpret_cast<const v4*>(wr + 4 * u);
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const size_t k = (size_t)st * K + tid + 256 * j;
            P.sw[j][0] = *reinterpret_cast<const v4*>(cbn_pre + k * Dc); P.sw[j][1] = *reinterpret_cast<const v4*>(cbn_pre + k * Dc + 4);
            P.rw[j][0] = *reinterpret_cast<const v4*>(cb + k * Dc); P.rw[j][1] = *reinterpret_cast<const v4*>(cb + k * Dc + 4);
            P.cn2v[j] = cn2_pre[k];
        }
#pragma unroll
        for (int j = 0; j < CO; ++j) {
            const float* wr = out_w + ((size_t)st * C + c0 + j) * Dc;
            P.ow[j][0] = *reinterpret_cast<const v4*>(wr); P.ow[j][1] = *reinterpret_cast<const v4*>(wr + 4);
            P.obv[j] = out_b[(size_t)st * C + c0 + j];
        }
        P.inb = in_b[(size_t)st * Dc + (tid & (Dc - 1))];
    };
    auto stage = [&](const StageOps& P, StageOps& Pn, int st) __attribute__((always_inline)) {
        fetch(Pn, st + 1 < nq ? st + 1 : st);                      // in flight for this whole stage.  UNCONDITIONAL (the last stage requests its
                                                                   // own rows again): behind a branch the wait for P would have to cover Pn's loads too
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                           // res_s holds this stage's residuals
        if (inproj) {   // in_proj block partial: channels g*CPT .. +CPT-1 ascending, from +0 -- one chain per token, interleaved
            float p[TOKB];
#pragma unroll
            for (int k = 0; k < TOKB; ++k) p[k] = 0.0f;
#pragma unroll
            for (int u = 0; u < IV; ++u) {
#pragma unroll
                for (int k = 0; k < TOKB; ++k) {
                    const v4 r4 = *reinterpret_cast<const v4*>(&res_s[k][g * CPT + 4 * u]);
                    p[k] = dfma(P.iw[u].x, r4.x, p[k]); p[k] = dfma(P.iw[u].y, r4.y, p[k]); p[k] = dfma(P.iw[u].z, r4.z, p[k]); p[k] = dfma(P.iw[u].w, r4.w, p[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < TOKB; ++k) part[k][g][d_in] = p[k];
        }
        __syncthreads();
        if (tid < TOKB * Dc) {                                     // the 16 block partials in block order, + bias: thread (token, dimension)
            const int k = tid >> 3, d = tid & 7;
            float a = part[k][0][d];
#pragma unroll
            for (int gg = 1; gg < 16; ++gg) a = a + part[k][gg][d];
            const float v = a + P.inb;                             // inb was fetched for dimension tid & 7
            ze[k][d] = v;
            const int n = blockIdx.x * TOKB + k;                   // (k is a run-time index here: recompute the token's coordinates)
            if (n < N) latents[((size_t)(n / T) * nq * Dc + (size_t)st * Dc + d) * T + (n - (n / T) * T)] = v;
        }
        __syncthreads();
        float bs[TOKB]; int bi[TOKB];
        v4 br0[TOKB], br1[TOKB];                                   // raw row of this thread's best code so far, per token (a winner publishes its own)
#pragma unroll
        for (int k = 0; k < TOKB; ++k) {   // F.normalize over Dc (every thread computes the same values), then this thread's codes, ascending
            bs[k] = -__builtin_inff(); bi[k] = 0x7fffffff; br0[k] = P.rw[0][0]; br1[k] = P.rw[0][1];
            float ss = 0.0f;
#pragma unroll
            for (int d = 0; d < Dc; ++d) { const float v = ze[k][d]; ss = dfma(v, v, ss); }
            const float den = __builtin_fmaxf(__builtin_sqrtf(ss), 1e-12f);
            float s2 = 0.0f, ev[Dc];
#pragma unroll
            for (int d = 0; d < Dc; ++d) { ev[d] = ze[k][d] / den; s2 = dfma(ev[d], ev[d], s2); }
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                float dot = 0.0f;
                dot = dfma(ev[0], P.sw[j][0].x, dot); dot = dfma(ev[1], P.sw[j][0].y, dot); dot = dfma(ev[2], P.sw[j][0].z, dot); dot = dfma(ev[3], P.sw[j][0].w, dot);
                dot = dfma(ev[4], P.sw[j][1].x, dot); dot = dfma(ev[5], P.sw[j][1].y, dot); dot = dfma(ev[6], P.sw[j][1].z, dot); dot = dfma(ev[7], P.sw[j][1].w, dot);
                const float dist = (s2 - 2.0f * dot) + P.cn2v[j];
                const float sc = -dist;
                if (sc > bs[k]) { bs[k] = sc; bi[k] = tid + 256 * j; br0[k] = P.rw[j][0]; br1[k] = P.rw[j][1]; }
            }
        }
#pragma unroll
        for (int k = 0; k < TOKB; ++k) {
            float s_ = bs[k]; int i_ = bi[k];
            wave_argmax(s_, i_);
            if (lane == 0) { red_s[k][wv] = s_; red_i[k][wv] = i_; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TOKB; ++k) {
            float cs = red_s[k][0]; int id = red_i[k][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) amax_combine(cs, id, red_s[k][w], red_i[k][w]);
            if (id < 0 || id >= K) id = 0;
            if (tid == (id & 255)) {                               // the thread that holds the winning code: its own best IS the winner
                const v4 r0 = br0[k], r1 = br1[k];
                const float raw[Dc] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int d = 0; d < Dc; ++d) { const float zv = ze[k][d]; pre[k][d] = zv + (raw[d] - zv); }
                if (livev[k]) codes[((size_t)bbv[k] * nq + st) * T + ttv[k]] = id;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TOKB; ++k) {   // out_proj: an 8-long chain per owned channel, + bias; accumulate (under the item's stage limit), update the residual
            float pv[Dc];
#pragma unroll
            for (int d = 0; d < Dc; ++d) pv[d] = pre[k][d];
#pragma unroll
            for (int j = 0; j < CO; ++j) {
                float a = 0.0f;
                a = dfma(P.ow[j][0].x, pv[0], a); a = dfma(P.ow[j][0].y, pv[1], a); a = dfma(P.ow[j][0].z, pv[2], a); a = dfma(P.ow[j][0].w, pv[3], a);
                a = dfma(P.ow[j][1].x, pv[4], a); a = dfma(P.ow[j][1].y, pv[5], a); a = dfma(P.ow[j][1].z, pv[6], a); a = dfma(P.ow[j][1].w, pv[7], a);
                const float zqi = a + P.obv[j];
                if (st < limv[k]) acc[k][j] = acc[k][j] + zqi;
                resr[k][j] = resr[k][j] - zqi;
                res_s[k][c0 + j] = resr[k][j];
            }
        }
    };
    StageOps P0, P1;
    if (nq > 0) fetch(P0, 0);
    for (int st = 0; st < nq; st += 2) {
        stage(P0, P1, st);
        if (st + 1 < nq) stage(P1, P0, st + 1);
    }
#pragma unroll
    for (int k = 0; k < TOKB; ++k)
        if (livev[k]) {
#pragma unroll
            for (int j = 0; j < CO; ++j) zq[((size_t)bbv[k] * C + c0 + j) * T + ttv[k]] = acc[k][j];
        }
}

template <int CPT>
static hipError_t launch_dac_rvq_lat_t(const float* z, const float* in_w, const float* in_b, const float* cb,
                                       const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                                       const int32_t* nq_item, int B, int T, int nq, hipStream_t s, const float* cbn_pre, const float* cn2_pre)
{
    // Tokens per block: ONE while that is at most one block per CU (N <= 256: the 75 tokens of one segment take 240 us), TWO beyond
    // (the reference's batch of six is 450 tokens: 225 blocks in one round instead of 450 in two).  Four share a stage's 104 KB
    // among four tokens and interleave their chains, but a stage then takes 12.4 us instead of 6.5 and a quarter as many CUs work:
    // 396 against 240 us at 75 tokens (gpurun_out/h8).  What a stage costs is its ~3 000 dependent-issue instructions per token on a
    // wave that is alone on its SIMD, not the bytes.
    const int N = B * T;
    if (N <= 256)
        hipLaunchKernelGGL((dac_rvq_lat_kernel<CPT, 4, 1>), dim3((unsigned)N), dim3(256), 0, s,
                           z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, cbn_pre, cn2_pre);
    else
        hipLaunchKernelGGL((dac_rvq_lat_kernel<CPT, 4, 2>), dim3((unsigned)((N + 1) / 2)), dim3(256), 0, s,
                           z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, cbn_pre, cn2_pre);
    return hipGetLastError();
}

template <int CPT, int DC>
static hipError_t launch_dac_rvq_t(const float* z, const float* in_w, const float* in_b, const float* cb,
                                   const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                                   const int32_t* nq_item, int B, int T, int nq, int K, hipStream_t s,
                                   const float* cbn_pre = nullptr, const float* cn2_pre = nullptr)
{
    constexpr int C = 16 * CPT;
    const int N = B * T;
    const size_t lds = ((size_t)K * DC + K + (size_t)DC * C + 512 + 16 * (size_t)DC * DQ_TOK + 2 * (size_t)DC * DQ_TOK) * sizeof(float);
    auto kern = dac_rvq_kernel<CPT, DC>;
    static BigLdsOptIn opt;                           // per (instantiation, device)
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((N + DQ_TOK - 1) / DQ_TOK), dim3(256), lds, s,
                       z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, K, cbn_pre, cn2_pre);
    return hipGetLastError();
}

hipError_t launch_dac_rvq(const float* z, const float* in_w, const float* in_b, const float* cb, const float* out_w,
                          const float* out_b, float* zq, int32_t* codes, float* latents, const int32_t* nq_item,
                          int B, int C, int T, int nq, int K, int Dc, hipStream_t s, const float* cbn_pre, const float* cn2_pre)
{
    if (B * T == 0) return hipSuccess;
    if (Dc != 8) return hipErrorInvalidValue;
    // a handful of tokens (one segment is 75, the reference's batch of six 450): the latency form, one token per block
    static const bool no_lat = [] { const bool o = getenv("MVQ_NO_DAC_RVQ_LAT") != nullptr; if (o) mvq::note_env_override(0x2000); return o; }();
    const bool aligned = ((reinterpret_cast<uintptr_t>(in_w) | reinterpret_cast<uintptr_t>(out_w) | reinterpret_cast<uintptr_t>(cb) |
                           reinterpret_cast<uintptr_t>(cbn_pre)) & 15) == 0;
    if (B * T <= 1024 && K == 1024 && cbn_pre && cn2_pre && aligned && !no_lat) {
        switch (C) {
            case 1024: return launch_dac_rvq_lat_t<64>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, s, cbn_pre, cn2_pre);
            case 512:  return launch_dac_rvq_lat_t<32>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, s, cbn_pre, cn2_pre);
            case 256:  return launch_dac_rvq_lat_t<16>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, s, cbn_pre, cn2_pre);
        }
    }
    switch (C) {
        case 1024: return launch_dac_rvq_t<64, 8>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, K, s, cbn_pre, cn2_pre);
        case 512:  return launch_dac_rvq_t<32, 8>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, K, s, cbn_pre, cn2_pre);
        case 256:  return launch_dac_rvq_t<16, 8>(z, in_w, in_b, cb, out_w, out_b, zq, codes, latents, nq_item, B, T, nq, K, s, cbn_pre, cn2_pre);
    }
    return hipErrorInvalidValue;
}

}  // namespace mvq
