// kernels_bwd.hip -- backward kernels of the reference-owned trainable modules:
// LayerNorm / TokenNorm, GELU, the cross-attention core, tanh*scale, plus the data-movement pieces the weight
// gradients need (2-D transpose, row sums).  Weight gradients themselves are GEMMs over the token axis and run on the
// conv MFMA kernel (K = tokens) -- see ops.linear_wgrad.  Parity bar for these rows: torch autograd on the torch
// restatement (fp32 tolerance); they are not part of the bit-exact forward contract.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "det_math.hpp"
#include "kernels_small.hpp"

namespace mvq {

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over C, backward.  Element (b,c,t) at b*sb + c*sc + t (same addressing as the forward).
//   pass 1 (one thread per token): mu, rstd, m1 = mean(g*gamma), m2 = mean(g*gamma*xhat) -> gx ; stats[n] = (mu, rstd)
//   pass 2 (one block per channel): dgamma[c] += sum_n g*xhat, dbeta[c] += sum_n g       (accumulates into the outputs)
// ---------------------------------------------------------------------------------------------------------------
// 16 tokens x 16 channel-groups per block: thread (tok, cg) walks channels cg, cg+16, ... of its token; the four
// per-token reductions (sum, variance, m1, m2) are combined across the 16 groups through LDS.  x and g are re-read from
// L2 (a block's working set is 2 x 16 x C floats) rather than staged, so any C fits.
constexpr int LNB_TOK = 16, LNB_CG = 16;

__device__ __forceinline__ float lnb_reduce(float v, float* red, int tok, int cg)
{
    __syncthreads();
    red[cg * LNB_TOK + tok] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < LNB_CG; ++k) s += red[k * LNB_TOK + tok];
    return s;
}

__global__ __launch_bounds__(256) void layernorm_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                       const float* __restrict__ gamma, const float* __restrict__ g,
                                       float* __restrict__ gx, float* __restrict__ stats,
                                       int B, int C, int T, size_t sb, size_t sc, float eps)
{
    __shared__ float red[LNB_CG * LNB_TOK];
    const int tok = threadIdx.x & (LNB_TOK - 1), cg = threadIdx.x / LNB_TOK;
    const int n = blockIdx.x * LNB_TOK + tok;
    const bool live = n < B * T;
    const int b = live ? n / T : 0, t = live ? n - b * T : 0;
    const size_t base = (size_t)b * sb + t;
    const float* per = pe ? pe + (size_t)t * C : nullptr;
    float s = 0.0f;
    if (live) for (int c = cg; c < C; c += LNB_CG) { float v = x[base + (size_t)c * sc]; if (per) v += per[c]; s += v; }
    const float mu = lnb_reduce(s, red, tok, cg) / (float)C;
    float var = 0.0f;
    if (live) for (int c = cg; c < C; c += LNB_CG) { float v = x[base + (size_t)c * sc]; if (per) v += per[c]; const float d = v - mu; var = dfma(d, d, var); }
    const float rstd = 1.0f / __builtin_sqrtf(lnb_reduce(var, red, tok, cg) / (float)C + eps);
    float m1 = 0.0f, m2 = 0.0f;
    if (live) for (int c = cg; c < C; c += LNB_CG) {
        float v = x[base + (size_t)c * sc]; if (per) v += per[c];
        const float gg = g[base + (size_t)c * sc] * gamma[c];
        m1 += gg; m2 = dfma(gg, (v - mu) * rstd, m2);
    }
    m1 = lnb_reduce(m1, red, tok, cg) / (float)C;
    m2 = lnb_reduce(m2, red, tok, cg) / (float)C;
    if (!live) return;
    if (gx) {
        for (int c = cg; c < C; c += LNB_CG) {
            float v = x[base + (size_t)c * sc]; if (per) v += per[c];
            const float xh = (v - mu) * rstd;
            const float gg = g[base + (size_t)c * sc] * gamma[c];
            gx[base + (size_t)c * sc] = rstd * (gg - m1 - xh * m2);
        }
    }
    if (cg == 0) { stats[2 * n] = mu; stats[2 * n + 1] = rstd; }
}

// The same arithmetic with the thread's channels held in registers (C = 16 * EPT): x and g are read ONCE -- every operand requested
// before the first is used -- instead of four dependent passes over L2, and TOK = 4 tokens per block put a handful of tokens on four
// times as many CUs.  profiles/r05_kernel_stats_B6_train_*: 67.8 us per call at the reference's batch of six (96 tokens per AR
// chunk: six blocks, each thread 4 x 64 dependent L2 reads), 20 calls per training step.  Thread (tok, cg) still owns channels
// cg, cg + 16, ... in ascending order and the 16 group partials are still added in group order: same bits as the kernel above.
template <int TOK, int EPT>
__global__ __launch_bounds__(TOK * LNB_CG) void layernorm_bwd_x_regs_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                       const float* __restrict__ gamma, const float* __restrict__ g,
                                       float* __restrict__ gx, float* __restrict__ stats,
                                       int B, int T, size_t sb, size_t sc, float eps)
{
    constexpr int C = LNB_CG * EPT;
    __shared__ float red[LNB_CG * TOK];
    const int tok = threadIdx.x & (TOK - 1), cg = threadIdx.x / TOK;
    const int n = blockIdx.x * TOK + tok;
    const bool live = n < B * T;
    const int nc = live ? n : B * T - 1;           // a block's spare threads walk the LAST token (unconditional loads: a branch per
    const int b = nc / T, t = nc - b * T;          // element made every load wait for the one before, 36 us per call) and store nothing
    const size_t base = (size_t)b * sb + t;
    const float* per = pe ? pe + (size_t)t * C : nullptr;
    auto reduce = [&](float v) __attribute__((always_inline)) {
        __syncthreads();
        red[cg * TOK + tok] = v;
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < LNB_CG; ++k) s += red[k * TOK + tok];
        return s;
    };
    float xv[EPT], gg[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int c = cg + LNB_CG * j;
        xv[j] = x[base + (size_t)c * sc];
        gg[j] = g[base + (size_t)c * sc];
    }
    if (per) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) xv[j] += per[cg + LNB_CG * j];
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) gg[j] = gg[j] * gamma[cg + LNB_CG * j];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) s += xv[j];
    const float mu = reduce(s) / (float)C;
    float var = 0.0f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) { const float d = xv[j] - mu; var = dfma(d, d, var); }
    const float rstd = 1.0f / __builtin_sqrtf(reduce(var) / (float)C + eps);
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) { m1 += gg[j]; m2 = dfma(gg[j], (xv[j] - mu) * rstd, m2); }
    m1 = reduce(m1) / (float)C;
    m2 = reduce(m2) / (float)C;
    if (!live) return;
    if (gx) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const float xh = (xv[j] - mu) * rstd;
            gx[base + (size_t)(cg + LNB_CG * j) * sc] = rstd * (gg[j] - m1 - xh * m2);
        }
    }
    if (cg == 0) { stats[2 * n] = mu; stats[2 * n + 1] = rstd; }
}

__global__ __launch_bounds__(256) void layernorm_bwd_param_kernel(
    const float* __restrict__ x, const float* __restrict__ pe, const float* __restrict__ g,
    const float* __restrict__ stats, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int B, int C, int T, size_t sb, size_t sc)
{
    __shared__ float r1[256], r2[256];
    const int c = blockIdx.x;
    float a1 = 0.0f, a2 = 0.0f;
    for (int n = threadIdx.x; n < B * T; n += 256) {
        const int b = n / T, t = n - b * T;
        const size_t off = (size_t)b * sb + (size_t)c * sc + t;
        float v = x[off]; if (pe) v += pe[(size_t)t * C + c];
        const float xh = (v - stats[2 * n]) * stats[2 * n + 1];
        const float gv = g[off];
        a1 = dfma(gv, xh, a1); a2 += gv;
    }
    r1[threadIdx.x] = a1; r2[threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dgamma[c] += r1[0]; dbeta[c] += r2[0]; }
}

hipError_t launch_layernorm_bwd(const float* x, const float* pe, const float* gamma, const float* g, float* gx,
                                float* dgamma, float* dbeta, float* stats, int B, int C, int T, size_t sb, size_t sc,
                                float eps, hipStream_t s)
{
    const int n = B * T;
    if (n == 0) return hipSuccess;
    if (C == 1024 && n <= 1024)                      // the predictor / TokenNorm width, a few AR chunks of tokens: 4 tokens per block
        hipLaunchKernelGGL((layernorm_bwd_x_regs_kernel<4, 64>), dim3((n + 3) / 4), dim3(4 * LNB_CG), 0, s, x, pe, gamma, g, gx, stats, B, T, sb, sc, eps);
    else if (C == 1024)
        hipLaunchKernelGGL((layernorm_bwd_x_regs_kernel<LNB_TOK, 64>), dim3((n + LNB_TOK - 1) / LNB_TOK), dim3(LNB_TOK * LNB_CG), 0, s, x, pe, gamma, g, gx, stats, B, T, sb, sc, eps);
    else
        hipLaunchKernelGGL(layernorm_bwd_x_kernel, dim3((n + LNB_TOK - 1) / LNB_TOK), dim3(256), 0, s, x, pe, gamma, g, gx, stats, B, C, T, sb, sc, eps);
    hipLaunchKernelGGL(layernorm_bwd_param_kernel, dim3(C), dim3(256), 0, s, x, pe, g, stats, dgamma, dbeta, B, C, T, sb, sc);
    return hipGetLastError();
}

// gx = g * d gelu(x)/dx,  gelu'(x) = Phi(x) + x*phi(x)
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float cdf = 0.5f * (1.0f + det_erf(v * 0.707106781186547524f));
        const float pdf = 0.3989422804014327f * det_exp(-0.5f * v * v);
        gx[i] = g[i] * dfma(v, pdf, cdf);
    }
}

hipError_t launch_gelu_bwd(const float* x, const float* g, float* gx, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, g, gx, n);
    return hipGetLastError();
}

// y = s * tanh(u) ;  backward: gu = g * s * (1 - t^2), partial[block] = sum over the block's elements of g * t
__global__ void scale_tanh_kernel(const float* __restrict__ u, float s, float* __restrict__ y, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = s * det_tanh(u[i]);
}

__global__ __launch_bounds__(256) void scale_tanh_bwd_kernel(const float* __restrict__ u, const float* __restrict__ g,
                                                             float s, float* __restrict__ gu, float* __restrict__ partial, size_t n)
{
    __shared__ float red[256];
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float t = det_tanh(u[i]);
        const float gv = g[i];
        gu[i] = gv * s * dfma(-t, t, 1.0f);
        acc = dfma(gv, t, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

hipError_t launch_scale_tanh(const float* u, float s, float* y, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_tanh_kernel, dim3((unsigned)blocks), dim3(256), 0, st, u, s, y, n);
    return hipGetLastError();
}

hipError_t launch_scale_tanh_bwd(const float* u, const float* g, float s, float* gu, float* partial, int n_partial, size_t n,
                                 hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(scale_tanh_bwd_kernel, dim3((unsigned)n_partial), dim3(256), 0, st, u, g, s, gu, partial, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// attention backward, one block per (batch, head); same staging as the forward.  P is recomputed.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_bwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ G,
    float* __restrict__ gQ, float* __restrict__ gK, float* __restrict__ gV,
    int B, int H, int dh, int Tq, int Tk, size_t qsb, size_t qsc, size_t ksb, size_t ksc)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;                       // [dh][Tq]
    float* Ks = Qs + dh * Tq;             // [dh][Tk]
    float* Vs = Ks + dh * Tk;             // [dh][Tk]
    float* Gs = Vs + dh * Tk;             // [dh][Tq]  dL/dctx
    float* P = Gs + dh * Tq;              // [Tq][Tk]
    float* dS = P + Tq * Tk;              // [Tq][Tk]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / H, hd = blockIdx.x - b * H;
    const size_t qo = (size_t)b * qsb + (size_t)hd * dh * qsc, ko = (size_t)b * ksb + (size_t)hd * dh * ksc;
    for (int e = tid; e < dh * Tq; e += 256) { const int d = e / Tq, i = e - d * Tq; Qs[e] = Q[qo + (size_t)d * qsc + i]; Gs[e] = G[qo + (size_t)d * qsc + i]; }
    for (int e = tid; e < dh * Tk; e += 256) { const int d = e / Tk, j = e - d * Tk; Ks[e] = K[ko + (size_t)d * ksc + j]; Vs[e] = V[ko + (size_t)d * ksc + j]; }
    __syncthreads();
    const float rs = __builtin_sqrtf((float)dh);
    for (int p = tid; p < Tq * Tk; p += 256) {
        const int i = p / Tk, j = p - i * Tk;
        float a = 0.0f, dp = 0.0f;
        for (int d = 0; d < dh; ++d) { a = dfma(Qs[d * Tq + i], Ks[d * Tk + j], a); dp = dfma(Gs[d * Tq + i], Vs[d * Tk + j], dp); }
        P[p] = a / rs; dS[p] = dp;                                   // dS holds dP for now
    }
    __syncthreads();
    if (tid < Tq) {
        float* pr = P + tid * Tk; float* dr = dS + tid * Tk;
        float m = -__builtin_inff();
        for (int j = 0; j < Tk; ++j) m = __builtin_fmaxf(m, pr[j]);
        float l = 0.0f;
        for (int j = 0; j < Tk; ++j) { const float e = det_exp(pr[j] - m); pr[j] = e; l += e; }
        float dot = 0.0f;
        for (int j = 0; j < Tk; ++j) { pr[j] = pr[j] / l; dot = dfma(dr[j], pr[j], dot); }
        for (int j = 0; j < Tk; ++j) dr[j] = pr[j] * (dr[j] - dot) / rs;      // dL/d(QK^T), scaled for the 1/sqrt(dh)
    }
    __syncthreads();
    for (int e = tid; e < dh * Tq; e += 256) {                                  // dQ[d][i] = sum_j dS[i][j] K[d][j]
        const int d = e / Tq, i = e - d * Tq;
        float a = 0.0f;
        for (int j = 0; j < Tk; ++j) a = dfma(dS[i * Tk + j], Ks[d * Tk + j], a);
        gQ[qo + (size_t)d * qsc + i] = a;
    }
    for (int e = tid; e < dh * Tk; e += 256) {                                  // dK[d][j] = sum_i dS[i][j] Q[d][i]; dV[d][j] = sum_i P[i][j] G[d][i]
        const int d = e / Tk, j = e - d * Tk;
        float a = 0.0f, c = 0.0f;
        for (int i = 0; i < Tq; ++i) { a = dfma(dS[i * Tk + j], Qs[d * Tq + i], a); c = dfma(P[i * Tk + j], Gs[d * Tq + i], c); }
        gK[ko + (size_t)d * ksc + j] = a;
        gV[ko + (size_t)d * ksc + j] = c;
    }
}

hipError_t launch_attention_bwd(const float* q, const float* k, const float* v, const float* g, float* gq, float* gk, float* gv,
                                int B, int H, int dh, int Tq, int Tk, size_t qsb, size_t qsc, size_t ksb, size_t ksc, hipStream_t s)
{
    if (B * H == 0 || Tq == 0) return hipSuccess;
    const size_t lds = ((size_t)dh * (2 * Tq + 2 * Tk) + 2 * (size_t)Tq * Tk) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(256), lds, s, q, k, v, g, gq, gk, gv, B, H, dh, Tq, Tk, qsb, qsc, ksb, ksc);
    return hipGetLastError();
}

// out = a * b * scale   (dropout mask application and its backward)
__global__ void mul_scaled_kernel(const float* __restrict__ a, const float* __restrict__ b, float scale, float* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i] * scale;
}

hipError_t launch_mul_scaled(const float* a, const float* b, float scale, float* out, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mul_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, scale, out, n);
    return hipGetLastError();
}

// out[c][r] = in[r][c]  (rows x cols -> cols x rows), 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) { const int r = r0 + k, c = c0 + tx; tile[k][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.0f; }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) { const int c = c0 + k, r = r0 + tx; if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][k]; }
}

hipError_t launch_transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, in, out, rows, cols);
    return hipGetLastError();
}

// out[r] (+)= sum_c in[r][c]   (bias gradients): one block per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ in, float* __restrict__ out, int cols, int accumulate)
{
    __shared__ float red[256];
    const float* row = in + (size_t)blockIdx.x * cols;
    float a = 0.0f;
    for (int c = threadIdx.x; c < cols; c += 256) a += row[c];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = accumulate ? out[blockIdx.x] + red[0] : red[0];
}

hipError_t launch_rowsum(const float* in, float* out, int rows, int cols, int accumulate, hipStream_t s)
{
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(rowsum_kernel, dim3(rows), dim3(256), 0, s, in, out, cols, accumulate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// optimiser step of the training config (torch.optim.AdamW as the reference builds it, Training/compare_dacvsproposal_5.py:
// 367; clip_grad_norm_ at ...:394): partial sums of squares for the global gradient norm, and the decoupled-weight-decay
// Adam update with the clip factor applied to the gradient on the fly.  Same operation order as torch's single-tensor
// AdamW:  p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, size_t n)
{
    __shared__ float red[256];
    float a = 0.0f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a = dfma(x[i], x[i], a);
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

hipError_t launch_sumsq_partial(const float* x, float* partial, int n_partial, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)n_partial), dim3(256), 0, s, x, partial, n);
    return hipGetLastError();
}

// clip_coef[0] (device): factor the gradients are multiplied by (min(1, max_norm/(norm + 1e-6)), or 1 when NULL)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             const float* __restrict__ clip_coef, size_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bc1, float sqrt_bc2)
{
    const float cc = clip_coef ? clip_coef[0] : 1.0f;
    const float step = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * cc;
        float pi = p[i] * (1.0f - lr * weight_decay);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;               // lerp form used by torch: m + (g - m)*(1-b1)
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        const float denom = __builtin_sqrtf(vi) / sqrt_bc2 + eps;
        pi = pi - step * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, const float* clip_coef, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bc1, float sqrt_bc2, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, clip_coef, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrt_bc2);
    return hipGetLastError();
}

}  // namespace mvq
