// ln_lat.hpp -- the LATENCY form of the channel LayerNorm (4 tokens per task): one device function shared by the stand-alone kernel
// (kernels_small.hip: layernorm_c_lat_kernel) and the persistent AR-loop kernel (ar_fused.hip).
//
// Reference arithmetic: nn.LayerNorm over the channel axis of a [B, C, T] tensor as the predictor / TokenNorm apply it
// (Training/compare_dacvsproposal_5.py:246-252 TokenNorm, :196-222 CrossPredictor), restated by oracle/c/oracle.c (ln_channels):
// per token, s = sum of v over c ascending (sequential fp32 adds), mean = s / C, var = sum of fma(d, d, var) with d = v - mean over
// c ascending, rstd = 1 / sqrt(var / C + eps), y = fma(d * rstd, gamma, beta) (then tanh and scale for TokenNorm).
//
// What a small call waits for (gpurun_out/f4, one 16-token chunk: 24-35 us per LayerNorm of ~230 us per chunk) is not bandwidth:
// it is (a) dependent round trips to far memory -- x, then sub, then pe, then gamma / beta channel by channel -- and
// (b) the two ordered chains, 2 C dependent operations fed one ds_read_b32 at a time.  Here
//   * every global operand of a thread (x, sub, pe: up to 48 loads at C = 1024) is requested before the first is used;
//   * the tile is kept TOKEN-major, tile[tok][C + 4], so the chain lane reads its column with ds_read_b128 (4 channels per LDS
//     instruction), two batches of 32 channels ahead of the additions;
//   * d = v - mean is computed ONCE by all threads between the two chains (the same rounding the oracle's var and y both use), so the
//     second chain is one fma per channel, and the epilogue starts from d.
// The chains themselves are untouched: same operations, same order, same bits (tests/test_gpu_parity_ops.py, test_gpu_ar_fused.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "det_math.hpp"

namespace mvq {

// v = x - sub + pe (each optional, in this order).  x: a strided tensor, or (x == nullptr) the predictor's shift-by-one input: zero
// except token 0 of an item, which is prev[b * prev_sb + c * prev_sc] (prev == nullptr: all zero).  Written to up to two layouts.
struct LnIo {
    const float* x; size_t x_sb, x_sc;           // element (b, c, i) at x + b * x_sb + c * x_sc + i
    const float* prev; size_t prev_sb, prev_sc;
    const float* sub; size_t sub_sb, sub_sc;
    const float* pe;                             // [pos][C] or null
    const float *gamma, *beta;
    float* y0; size_t y0_sb, y0_sc;
    float* y1; size_t y1_sb, y1_sc;              // second copy or null
    float eps, post_scale; int do_tanh;
};

constexpr int LN_LAT_TOK = 4;
__host__ __device__ constexpr size_t ln_lat_lds_floats(int C) { return (size_t)LN_LAT_TOK * (C + 4) + 2 * LN_LAT_TOK; }

// Task `task` = tokens [4 task, 4 task + 4) of the B * n tokens (token nn = item nn / n, position nn % n).  The first 256 threads of
// the block work; EVERY thread of the block must call (block-wide barriers).  C % 64 == 0.  ACT_LOAD: how activations are read
// (the fused kernel must not let them become scalar / read-only-cache loads).
template <class ActLoad>
__device__ __forceinline__ void ln_lat_task(const LnIo& io, int task, int B, int n, int C, float* sm, int tid, ActLoad ld)
{
    constexpr int TOK = LN_LAT_TOK, CGR = 256 / TOK, EPT = 16;       // 64 channel groups; 16 elements per thread and pass
    const int CP = C + 4;
    float* const tile = sm;                                          // [TOK][CP]
    float* const mean_s = tile + (size_t)TOK * CP;
    float* const rstd_s = mean_s + TOK;
    const int tok = tid & (TOK - 1), cg = (tid & 255) / TOK;
    const bool worker = tid < 256;
    const int nn = task * TOK + tok;
    const bool live = worker && nn < B * n;
    const int nc = nn < B * n ? nn : B * n - 1;                      // spare lanes of the last task walk the LAST token and store nothing: every
    const int b = nc / n, i = nc - b * n;                            // load below is unconditional (no branch per element between the requests)
    float* const row = tile + (size_t)tok * CP;
    __syncthreads();                                                 // the block's previous task is done with the tile
    float gm[EPT], bt[EPT];                                          // gamma / beta of the thread's first 16 channels: requested now, used last
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int c = cg + u * CGR < C ? cg + u * CGR : C - 1;
        gm[u] = io.gamma[c];
        bt[u] = io.beta[c];
    }
    if (worker) {
        for (int c0 = cg; c0 < C; c0 += CGR * EPT) {
            float v[EPT], sv[EPT], pv[EPT];
#pragma unroll
            for (int u = 0; u < EPT; ++u) {                          // all requests first
                const int c = c0 + u * CGR < C ? c0 + u * CGR : C - 1;
                v[u] = 0.0f; sv[u] = 0.0f; pv[u] = 0.0f;
                if (io.x) v[u] = ld(io.x + (size_t)b * io.x_sb + (size_t)c * io.x_sc + i);
                else if (io.prev) v[u] = ld(io.prev + (size_t)b * io.prev_sb + (size_t)c * io.prev_sc);
                if (io.sub) sv[u] = ld(io.sub + (size_t)b * io.sub_sb + (size_t)c * io.sub_sc + i);
                if (io.pe) pv[u] = io.pe[(size_t)i * C + c];
            }
            const bool shifted = !io.x;                              // the shift-by-one input: only token 0 of an item carries a value
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int c = c0 + u * CGR;
                float e = (shifted && i != 0) ? 0.0f : v[u];
                if (io.sub) e = e - sv[u];
                if (io.pe) e = e + pv[u];
                if (c < C) row[c] = live ? e : 0.0f;
            }
        }
    }
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int C4 = C / 4;
    if (tid < TOK) {                                                 // chain 1: the ordered sum
        const f4* col = reinterpret_cast<const f4*>(tile + (size_t)tid * CP);
        float s = 0.0f;
        f4 p[8], q[8];
        int c4 = 0;
        if (C4 >= 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = col[u];
        }
        for (; c4 + 16 <= C4; c4 += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = col[c4 + 8 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s = s + p[u].x; s = s + p[u].y; s = s + p[u].z; s = s + p[u].w; }
            if (c4 + 24 <= C4) {
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = col[c4 + 16 + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s = s + q[u].x; s = s + q[u].y; s = s + q[u].z; s = s + q[u].w; }
        }
        if (c4 + 8 <= C4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { s = s + p[u].x; s = s + p[u].y; s = s + p[u].z; s = s + p[u].w; }
            c4 += 8;
        }
        for (; c4 < C4; ++c4) { const f4 e = col[c4]; s = s + e.x; s = s + e.y; s = s + e.z; s = s + e.w; }
        mean_s[tid] = s / (float)C;
    }
    __syncthreads();
    if (worker) {                                                    // d = v - mean, once, in place
        const float mean = mean_s[tok];
        for (int c = cg; c < C; c += CGR) row[c] = row[c] - mean;
    }
    __syncthreads();
    if (tid < TOK) {                                                 // chain 2: the ordered squared deviations
        const f4* col = reinterpret_cast<const f4*>(tile + (size_t)tid * CP);
        float var = 0.0f;
        f4 p[8], q[8];
        int c4 = 0;
        if (C4 >= 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = col[u];
        }
        for (; c4 + 16 <= C4; c4 += 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = col[c4 + 8 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { var = dfma(p[u].x, p[u].x, var); var = dfma(p[u].y, p[u].y, var); var = dfma(p[u].z, p[u].z, var); var = dfma(p[u].w, p[u].w, var); }
            if (c4 + 24 <= C4) {
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = col[c4 + 16 + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { var = dfma(q[u].x, q[u].x, var); var = dfma(q[u].y, q[u].y, var); var = dfma(q[u].z, q[u].z, var); var = dfma(q[u].w, q[u].w, var); }
        }
        if (c4 + 8 <= C4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { var = dfma(p[u].x, p[u].x, var); var = dfma(p[u].y, p[u].y, var); var = dfma(p[u].z, p[u].z, var); var = dfma(p[u].w, p[u].w, var); }
            c4 += 8;
        }
        for (; c4 < C4; ++c4) { const f4 e = col[c4]; var = dfma(e.x, e.x, var); var = dfma(e.y, e.y, var); var = dfma(e.z, e.z, var); var = dfma(e.w, e.w, var); }
        rstd_s[tid] = 1.0f / __builtin_sqrtf(var / (float)C + io.eps);
    }
    __syncthreads();
    if (live) {
        // The epilogue reads every operand before its first store: with the stores in between, each channel's gamma / beta load would
        // wait behind the previous channel's store (they may alias as far as the compiler knows) -- 16 dependent round trips to far
        // memory, 8 us of a 20 us call (gpurun_out/f7).
        const float rstd = rstd_s[tok];
        for (int c0 = cg; c0 < C; c0 += CGR * EPT) {
            float o[EPT];
            if (c0 != cg) {
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int c = c0 + u * CGR;
                    gm[u] = c < C ? io.gamma[c] : 0.0f;
                    bt[u] = c < C ? io.beta[c] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int c = c0 + u * CGR;
                float e = dfma((c < C ? row[c] : 0.0f) * rstd, gm[u], bt[u]);
                if (io.do_tanh) e = det_tanh(e);
                if (io.do_tanh || io.post_scale != 1.0f) e = io.post_scale * e;
                o[u] = e;
            }
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int c = c0 + u * CGR;
                if (c < C) {
                    io.y0[(size_t)b * io.y0_sb + (size_t)c * io.y0_sc + i] = o[u];
                    if (io.y1) io.y1[(size_t)b * io.y1_sb + (size_t)c * io.y1_sc + i] = o[u];
                }
            }
        }
    }
}

}  // namespace mvq
