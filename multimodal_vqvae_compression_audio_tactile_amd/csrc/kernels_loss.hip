// kernels_loss.hip -- the reference's training losses on the device:
// safe_l1, MultiResSTFTLoss, MelCosineLoss (Training/compare_dacvsproposal_5.py:150-211) with their gradients w.r.t. the
// predicted waveform.  MI355X design: the STFT is a windowed-frame matrix times a real DFT basis, i.e. a k=1 conv on the
// fp32 MFMA kernel (frames [n_fft, B*nframes] -> spectrum [2*Fp, B*nframes]); everything else here is the HBM-bound
// glue around those GEMMs: framing with reflect padding, magnitudes, the per-item reductions, the gradient of the
// magnitude, overlap-add back to the waveform, and the mel / log / cosine chain.  Column index = half*B*nframes +
// b*nframes + n, half 0 = prediction, half 1 = target.  Tolerance-checked against torch (tests/test_gpu_losses.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_small.hpp"

namespace mvq {

__device__ __forceinline__ int reflect_index(int j, int T)      // torch 'reflect' padding, |overhang| < T
{
    if (j < 0) j = -j;
    if (j >= T) j = 2 * (T - 1) - j;
    return j;
}

__device__ __forceinline__ float finite_or_zero(float v) { return (v == v && __builtin_fabsf(v) != __builtin_inff()) ? v : 0.0f; }

// out[f][col0 + b*nframes + n] = window[f] * x[b][reflect(n*hop + f - n_fft/2)]
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                          float* __restrict__ out, int B, int T, int n_fft, int hop,
                                                          int nframes, size_t ncols, size_t col0)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (c >= B * nframes) return;
    const int b = c / nframes, n = c - b * nframes;
    const int j = reflect_index(n * hop + f - n_fft / 2, T);
    out[(size_t)f * ncols + col0 + c] = window[f] * finite_or_zero(x[(size_t)b * T + j]);
}

// mag[k][c] = max(|S[k][c] + i S[Fp+k][c]|, eps) for k < F, 0 for F <= k < Fp
__global__ __launch_bounds__(256) void spec_mag_kernel(const float* __restrict__ S, float* __restrict__ mag, int F, int Fp,
                                                       size_t ncols, float eps)
{
    const size_t c = blockIdx.x * (size_t)256 + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= ncols) return;
    float m = 0.0f;
    if (k < F) {
        const float re = S[(size_t)k * ncols + c], im = S[(size_t)(Fp + k) * ncols + c];
        m = __builtin_fmaxf(__builtin_sqrtf(re * re + im * im), eps);
    }
    mag[(size_t)k * ncols + c] = m;
}

// per item b: partial[0][b][p] = sum (X-Y)^2, partial[1][b][p] = sum Y^2, partial[2][b][p] = sum |X-Y| over the block's
// share of the F x nframes spectrogram of item b.  grid (P, B).
__global__ __launch_bounds__(256) void spec_loss_partial_kernel(const float* __restrict__ mag, float* __restrict__ partial,
                                                                int F, int B, int nframes, size_t ncols)
{
    __shared__ float r0[256], r1[256], r2[256];
    const int b = blockIdx.y, P = gridDim.x;
    const size_t half = (size_t)B * nframes;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    const int total = F * nframes;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += P * 256) {
        const int k = e / nframes, n = e - k * nframes;
        const size_t o = (size_t)k * ncols + (size_t)b * nframes + n;
        const float X = mag[o], Y = mag[o + half], d = X - Y;
        a0 += d * d; a1 += Y * Y; a2 += __builtin_fabsf(d);
    }
    r0[threadIdx.x] = a0; r1[threadIdx.x] = a1; r2[threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)0 * B + b) * P + blockIdx.x] = r0[0];
        partial[((size_t)1 * B + b) * P + blockIdx.x] = r1[0];
        partial[((size_t)2 * B + b) * P + blockIdx.x] = r2[0];
    }
}

// G[k][c], G[Fp+k][c] (c over the prediction half only) = dL/d(re, im):
//   gX = coefA[b]*(X-Y) + coefB*sign(X-Y) + extra[k][c];  d|z|/dz = z/|z| where |z| >= eps (clamp_min passes the gradient
//   at and above the bound), 0 below.
__global__ __launch_bounds__(256) void spec_grad_kernel(const float* __restrict__ S, const float* __restrict__ mag,
                                                        const float* __restrict__ coefA, float coefB,
                                                        const float* __restrict__ extra, float* __restrict__ G,
                                                        int F, int Fp, int B, int nframes, size_t ncols, float eps)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    const int half = B * nframes;
    if (c >= half) return;
    float gre = 0.0f, gim = 0.0f;
    if (k < F) {
        const float X = mag[(size_t)k * ncols + c], Y = mag[(size_t)k * ncols + half + c], d = X - Y;
        float gX = coefA ? coefA[c / nframes] * d : 0.0f;
        gX += coefB * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
        if (extra) gX += extra[(size_t)k * half + c];
        const float re = S[(size_t)k * ncols + c], im = S[(size_t)(Fp + k) * ncols + c];
        const float a = __builtin_sqrtf(re * re + im * im);
        if (a >= eps && a > 0.0f) { gre = gX * re / a; gim = gX * im / a; }
    }
    G[(size_t)k * half + c] = gre;
    G[(size_t)(Fp + k) * half + c] = gim;
}

// dy[b][t] += sum over the padded positions that map to t (itself + the two reflections) of
//             sum_n window[j - n*hop] * dF[j - n*hop][b*nframes + n]
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ dF, const float* __restrict__ window,
                                                          float* __restrict__ dy, int B, int T, int n_fft, int hop, int nframes)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const int pad = n_fft / 2;
    const size_t ncols = (size_t)B * nframes;
    int js[3]; int nj = 0;
    js[nj++] = t + pad;
    if (t >= 1 && t <= pad) js[nj++] = pad - t;
    if (t <= T - 2 && t >= T - 1 - pad) js[nj++] = pad + 2 * (T - 1) - t;
    float acc = 0.0f;
    for (int q = 0; q < nj; ++q) {
        const int j = js[q];
        int n_hi = j / hop; if (n_hi > nframes - 1) n_hi = nframes - 1;
        int n_lo = (j - n_fft + hop) / hop; if (j - n_fft + 1 <= 0) n_lo = 0;     // ceil((j - n_fft + 1)/hop) for positive values
        for (int n = n_lo; n <= n_hi; ++n) {
            const int f = j - n * hop;
            if (f >= 0 && f < n_fft) acc += window[f] * dF[(size_t)f * ncols + (size_t)b * nframes + n];
        }
    }
    dy[(size_t)b * T + t] += acc;
}

// L1: partial[p] = sum |y - tgt| ; dy += coef * sign(y - tgt)   (coef = 0: forward only)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ y, const float* __restrict__ tgt,
                                                      float* __restrict__ partial, float* __restrict__ dy, float coef, size_t n)
{
    __shared__ float red[256];
    float a = 0.0f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = finite_or_zero(y[i]) - finite_or_zero(tgt[i]);
        a += __builtin_fabsf(d);
        if (dy) dy[i] += coef * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- mel / log / cosine -------------------------------------------------------------------------------------------
// maxv[h*B + b] = max over (mel, frame) of M[mel][h*B*nframes + b*nframes + n]; argmax (first, row-major mel,frame) too.
__global__ __launch_bounds__(256) void mel_max_kernel(const float* __restrict__ M, float* __restrict__ maxv, int* __restrict__ argm,
                                                      int n_mels, int B, int nframes, size_t ncols)
{
    __shared__ float rv[256]; __shared__ int ri[256];
    const int hb = blockIdx.x;                                       // h*B + b
    const size_t c0 = (size_t)hb * nframes;
    float best = -__builtin_inff(); int bi = 0x7fffffff;
    for (int e = threadIdx.x; e < n_mels * nframes; e += 256) {
        const int m = e / nframes, n = e - m * nframes;
        const float v = M[(size_t)m * ncols + c0 + n];
        if (v > best) { best = v; bi = e; }
    }
    rv[threadIdx.x] = best; ri[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v = rv[threadIdx.x + o]; const int i = ri[threadIdx.x + o];
            if (v > rv[threadIdx.x] || (v == rv[threadIdx.x] && i < ri[threadIdx.x])) { rv[threadIdx.x] = v; ri[threadIdx.x] = i; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { maxv[hb] = rv[0]; argm[hb] = ri[0]; }
}

// one thread per (b, frame): X = log(Mx/denx + eps), Y likewise; cos = clamp(<X,Y> / max(|X||Y|, eps), -1, 1).
// With dM != NULL also the gradient: dM[m][c] = dL/dMx (through the log and the division by the per-item max, max held
// constant) and dden[c] = this column's share of dL/d(denx); coef = dL/dcos (same for every column).
__global__ __launch_bounds__(256) void mel_cos_kernel(const float* __restrict__ M, const float* __restrict__ maxv,
                                                      float* __restrict__ cosv, float* __restrict__ dM, float* __restrict__ dden,
                                                      float coef, int n_mels, int B, int nframes, size_t ncols, float eps,
                                                      int use_log)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int half = B * nframes;
    if (c >= half) return;
    const int b = c / nframes;
    const float denx = __builtin_fmaxf(maxv[b], eps), deny = __builtin_fmaxf(maxv[B + b], eps);
    float num = 0.0f, nx = 0.0f, ny = 0.0f;
    for (int m = 0; m < n_mels; ++m) {
        float X = M[(size_t)m * ncols + c] / denx, Y = M[(size_t)m * ncols + half + c] / deny;
        if (use_log) { X = logf(X + eps); Y = logf(Y + eps); }              // MelCosineLoss; linear = stsim_batch
        num += X * Y; nx += X * X; ny += Y * Y;
    }
    nx = __builtin_sqrtf(nx); ny = __builtin_sqrtf(ny);
    const float prod = nx * ny, den2 = __builtin_fmaxf(prod, eps);
    const float v = num / den2;
    cosv[c] = __builtin_fminf(__builtin_fmaxf(v, -1.0f), 1.0f);
    if (!dM || !use_log) return;
    const float gv = (v >= -1.0f && v <= 1.0f) ? coef : 0.0f;          // clamp(-1, 1) passes the gradient inside the range
    const float gnum = gv / den2;
    const float gprod = (prod >= eps) ? -gv * num / (den2 * den2) : 0.0f;
    float dd = 0.0f;
    for (int m = 0; m < n_mels; ++m) {
        const float Mx = M[(size_t)m * ncols + c];
        const float u = Mx / denx + eps;
        const float X = logf(u), Y = logf(M[(size_t)m * ncols + half + c] / deny + eps);
        float gX = gnum * Y;
        if (nx > 0.0f) gX += gprod * ny * X / nx;
        const float gu = gX / u;
        dM[(size_t)m * half + c] = gu / denx;
        dd += -gu * Mx / (denx * denx);
    }
    dden[c] = dd;
}

// per item b: g = sum_n dden[b*nframes + n]; if max >= eps, add g to dM at the argmax element
__global__ __launch_bounds__(256) void mel_max_grad_kernel(const float* __restrict__ dden, const float* __restrict__ maxv,
                                                           const int* __restrict__ argm, float* __restrict__ dM,
                                                           int B, int nframes, float eps)
{
    __shared__ float red[256];
    const int b = blockIdx.x;
    float a = 0.0f;
    for (int n = threadIdx.x; n < nframes; n += 256) a += dden[(size_t)b * nframes + n];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && maxv[b] >= eps) {
        const int e = argm[b], m = e / nframes, n = e - m * nframes;
        dM[(size_t)m * ((size_t)B * nframes) + (size_t)b * nframes + n] += red[0];
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
hipError_t launch_stft_frames(const float* x, const float* window, float* out, int B, int T, int n_fft, int hop, int nframes,
                              size_t ncols, size_t col0, hipStream_t s)
{
    if (B * nframes == 0) return hipSuccess;
    hipLaunchKernelGGL(stft_frames_kernel, dim3((B * nframes + 255) / 256, n_fft), dim3(256), 0, s, x, window, out, B, T, n_fft, hop, nframes, ncols, col0);
    return hipGetLastError();
}
hipError_t launch_spec_mag(const float* S, float* mag, int F, int Fp, size_t ncols, float eps, hipStream_t s)
{
    if (ncols == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_mag_kernel, dim3((unsigned)((ncols + 255) / 256), Fp), dim3(256), 0, s, S, mag, F, Fp, ncols, eps);
    return hipGetLastError();
}
hipError_t launch_spec_loss_partial(const float* mag, float* partial, int P, int F, int B, int nframes, size_t ncols, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_loss_partial_kernel, dim3(P, B), dim3(256), 0, s, mag, partial, F, B, nframes, ncols);
    return hipGetLastError();
}
hipError_t launch_spec_grad(const float* S, const float* mag, const float* coefA, float coefB, const float* extra, float* G,
                            int F, int Fp, int B, int nframes, size_t ncols, float eps, hipStream_t s)
{
    if (B * nframes == 0) return hipSuccess;
    hipLaunchKernelGGL(spec_grad_kernel, dim3((B * nframes + 255) / 256, Fp), dim3(256), 0, s, S, mag, coefA, coefB, extra, G, F, Fp, B, nframes, ncols, eps);
    return hipGetLastError();
}
hipError_t launch_overlap_add(const float* dF, const float* window, float* dy, int B, int T, int n_fft, int hop, int nframes, hipStream_t s)
{
    if (B * T == 0) return hipSuccess;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, dF, window, dy, B, T, n_fft, hop, nframes);
    return hipGetLastError();
}
hipError_t launch_l1_loss(const float* y, const float* tgt, float* partial, int P, float* dy, float coef, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(l1_loss_kernel, dim3(P), dim3(256), 0, s, y, tgt, partial, dy, coef, n);
    return hipGetLastError();
}
hipError_t launch_mel_max(const float* M, float* maxv, int* argm, int n_mels, int B, int nframes, size_t ncols, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(mel_max_kernel, dim3(2 * B), dim3(256), 0, s, M, maxv, argm, n_mels, B, nframes, ncols);
    return hipGetLastError();
}
hipError_t launch_mel_cos(const float* M, const float* maxv, float* cosv, float* dM, float* dden, float coef, int n_mels, int B,
                          int nframes, size_t ncols, float eps, int use_log, hipStream_t s)
{
    if (B * nframes == 0) return hipSuccess;
    hipLaunchKernelGGL(mel_cos_kernel, dim3((B * nframes + 255) / 256), dim3(256), 0, s, M, maxv, cosv, dM, dden, coef, n_mels, B, nframes, ncols, eps, use_log);
    return hipGetLastError();
}
hipError_t launch_mel_max_grad(const float* dden, const float* maxv, const int* argm, float* dM, int B, int nframes, float eps, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(mel_max_grad_kernel, dim3(B), dim3(256), 0, s, dden, maxv, argm, dM, B, nframes, eps);
    return hipGetLastError();
}

}  // namespace mvq
