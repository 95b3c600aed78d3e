// kernels_small.hpp -- launchers of the non-GEMM kernels (kernels_small.hip, kernels_vq.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace mvq {

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): one flag per device ordinal, so a process that
// drives several GPUs opts every one of them in (a single per-process flag would leave the second device at the 64 KB default).
constexpr int kMaxLdsOptInDevices = 64;
struct BigLdsOptIn {
    bool done[kMaxLdsOptInDevices] = {};
    hipError_t ensure(const void* kern)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool tracked = dev >= 0 && dev < kMaxLdsOptInDevices;
        if (tracked && done[dev]) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess && tracked) done[dev] = true;
        return e;
    }
};

void note_env_override(unsigned bit);                                      // api.hip: environment A/B knobs seen -> mvq_build_flags()
int prof_begin(const char* kernel_name, double flops, hipStream_t s);     // api.hip (see conv1d_mfma.hpp)
void prof_end(int idx, hipStream_t s);
bool prof_enabled();

struct DirectConvArgs {
    const float* x; const float* wp; const float* bias; const float* alpha_in; const float* residual;
    const float* alpha_out; float* y;
    int B, Cin, Tin, Cout, Tout, ks, stride, dil, pad, Mpad, act;
    float* y2; const float* alpha2;       // dual output (see ConvArgs::y2)
    const float* dsn_src; const float* dsn_alpha;   // dgrad epilogue (see ConvArgs::dsn_src)
};

hipError_t launch_weight_norm(const float* v, const float* g, float* w, int rows, int inner, hipStream_t s);
hipError_t launch_pack_conv1d(const float* w, float* wp, int cin, int cout, int ks, int mpad, hipStream_t s);
hipError_t launch_pack_convtr(const float* w, float* wp, int cin, int cout, int S, int mpad, hipStream_t s);
hipError_t launch_conv1d_direct(const DirectConvArgs& a, hipStream_t s);
hipError_t launch_convtr_direct(const DirectConvArgs& a, hipStream_t s);
hipError_t launch_layernorm_c(const float* x, const float* pe, const float* gamma, const float* beta, float* y,
                              int B, int C, int T, size_t sb, size_t sc, float eps, int do_tanh, float post_scale,
                              const float* sub, hipStream_t s);
hipError_t launch_attention(const float* q, const float* k, const float* v, float* ctx,
                            int B, int H, int dh, int Tq, int Tk, size_t qsb, size_t qsc, size_t ksb, size_t ksc,
                            hipStream_t s);
hipError_t launch_gelu(const float* x, float* y, size_t n, hipStream_t s);
hipError_t launch_strided3d(const float* a, size_t asb, size_t asc, const float* b2, size_t bsb, size_t bsc,
                            float* y, size_t ysb, size_t ysc, int B, int C, int n, hipStream_t s);

hipError_t launch_pack_conv1d_dgrad(const float* w, float* wp, int cin, int cout, int ks, int mpad, hipStream_t s);
hipError_t launch_pack_convtr_dgrad(const float* w, float* wp, int cin, int cout, int ks, int mpad, hipStream_t s);
hipError_t launch_mul_dtanh(const float* g, const float* y, float* out, size_t n, hipStream_t s);
hipError_t launch_layernorm_bwd(const float* x, const float* pe, const float* gamma, const float* g, float* gx,
                                float* dgamma, float* dbeta, float* stats, int B, int C, int T, size_t sb, size_t sc,
                                float eps, hipStream_t s);
hipError_t launch_gelu_bwd(const float* x, const float* g, float* gx, size_t n, hipStream_t s);
hipError_t launch_scale_tanh(const float* u, float s, float* y, size_t n, hipStream_t st);
hipError_t launch_scale_tanh_bwd(const float* u, const float* g, float s, float* gu, float* partial, int n_partial, size_t n,
                                 hipStream_t st);
hipError_t launch_attention_bwd(const float* q, const float* k, const float* v, const float* g, float* gq, float* gk, float* gv,
                                int B, int H, int dh, int Tq, int Tk, size_t qsb, size_t qsc, size_t ksb, size_t ksc, hipStream_t s);
hipError_t launch_mul_scaled(const float* a, const float* b, float scale, float* out, size_t n, hipStream_t s);
hipError_t launch_transpose2d(const float* in, float* out, int rows, int cols, hipStream_t s);
hipError_t launch_rowsum(const float* in, float* out, int rows, int cols, int accumulate, hipStream_t s);
hipError_t launch_stft_frames(const float* x, const float* window, float* out, int B, int T, int n_fft, int hop, int nframes,
                              size_t ncols, size_t col0, hipStream_t s);
hipError_t launch_spec_mag(const float* S, float* mag, int F, int Fp, size_t ncols, float eps, hipStream_t s);
hipError_t launch_spec_loss_partial(const float* mag, float* partial, int P, int F, int B, int nframes, size_t ncols, hipStream_t s);
hipError_t launch_spec_grad(const float* S, const float* mag, const float* coefA, float coefB, const float* extra, float* G,
                            int F, int Fp, int B, int nframes, size_t ncols, float eps, hipStream_t s);
hipError_t launch_overlap_add(const float* dF, const float* window, float* dy, int B, int T, int n_fft, int hop, int nframes, hipStream_t s);
hipError_t launch_l1_loss(const float* y, const float* tgt, float* partial, int P, float* dy, float coef, size_t n, hipStream_t s);
hipError_t launch_mel_max(const float* M, float* maxv, int* argm, int n_mels, int B, int nframes, size_t ncols, hipStream_t s);
hipError_t launch_mel_cos(const float* M, const float* maxv, float* cosv, float* dM, float* dden, float coef, int n_mels, int B,
                          int nframes, size_t ncols, float eps, int use_log, hipStream_t s);
hipError_t launch_mel_max_grad(const float* dden, const float* maxv, const int* argm, float* dM, int B, int nframes, float eps, hipStream_t s);
hipError_t launch_resample(const float* x, const float* kern, float* y, int B, int L, int Lout, int orig, int newf,
                           int width, int ks, hipStream_t s);
hipError_t launch_sumsq_partial(const float* x, float* partial, int n_partial, size_t n, hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, const float* clip_coef, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bc1, float sqrt_bc2, hipStream_t s);
hipError_t launch_align_xcorr_batch(const float* r, const float* e, int B, int T, int max_shift, float* corr, int* scratch_valid,
                                    int* best_shift, hipStream_t s);
hipError_t launch_resample_ragged(const float* x, const float* kern, float* y, const int* off, const int* len, int* lout, int B,
                                  int pitch, int lout_pitch, int orig, int newf, int width, int ks, hipStream_t s);
hipError_t launch_align_xcorr(const float* r, const float* e, int T, int max_shift, float* corr, int* scratch_valid,
                              int* best_shift, hipStream_t s);

hipError_t launch_rvq_ema_forward(const float* z, const float* books, float* q_out, int32_t* idx_out,
                                  int B, int D, int T, int nb, int K, int update_residual, hipStream_t s);
size_t ema_update_scratch_bytes(int N, int nb, int K, int D);
hipError_t launch_ema_update(const float* z, const int32_t* idx, float* books, void* scratch, int B, int D, int T, int nb, int K,
                             float decay, hipStream_t s);
hipError_t launch_dac_rvq(const float* z, const float* in_w, const float* in_b, const float* cb, const float* out_w,
                          const float* out_b, float* zq, int32_t* codes, float* latents, const int32_t* nq_item,
                          int B, int C, int T, int nq, int K, int Dc, hipStream_t s,
                          const float* cbn_pre = nullptr, const float* cn2_pre = nullptr);
hipError_t launch_dac_rvq_prepare(const float* cb, float* cbn, float* cn2, int nq, int K, int Dc, hipStream_t s);
}  // namespace mvq
