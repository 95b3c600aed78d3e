/*
 * oracle.c -- CPU ORACLE for the encode -> vector-quantise -> decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * path (multimodal_vqvae_compression_audio_tactile_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * What it restates (file:line are relative to /root/reference):
 *   - the reference's OWN ops: ResidualVQEMA._nearest_l2/.forward/.ema_step
 *     (Training/compare_dacvsproposal_5.py:246-277, eval variant with n_books_use
 *     Evaluation/dac_vcpwq_proposed6_latency.py:409-435), CrossPredictor/TokenNorm pieces
 *     (Training/compare_dacvsproposal_5.py:214-244);
 *   - the third-party backbone the reference calls as black boxes (dac.DAC 24 kHz: encoder,
 *     quantizer, decoder -- call sites Training/compare_dacvsproposal_5.py:294-296,322).  That
 *     package (`descript-audio-codec`, version unpinned by the reference, latest known 1.0.0) is NOT
 *     in /root/reference and not installed, so its published architecture is restated here from
 *     the survey: weight-normalised Conv1d / ConvTranspose1d, Snake1d,
 *     ResidualUnit, and VectorQuantize.decode_latents (L2-normalised nearest neighbour).
 *     PARITY UNPINNED for those rows: the reference holds no golden vectors for them and the
 *     upstream code could not be run here.  (The reference's own classes ARE pinned: see
 *     tests/golden/ and tests/golden/make_golden.py.)
 *
 * Arithmetic contract (what makes bit-exact comparison with the HIP path possible):
 *   every dot product is ONE fp32 fma chain in a stated order, starting from +0.0f:
 *     conv1d        : for ci ascending, for tap kk ascending        acc = fma(w, x, acc)
 *     conv_transpose: for ci ascending, for input position ascending acc = fma(w, x, acc)
 *     then  v = acc + bias ; v = v + residual ; v = snake(v) ; v = tanh(v)   (each optional)
 *   which is exactly what a k-ordered v_mfma_f32_32x32x2_f32 accumulation produces on gfx950.
 *   Elementary functions come from det_math.h.  torch's own CPU kernels (oneDNN / Sleef) use other
 *   summation orders; oracle/dac24_torch.py restates the same path with torch ops and
 *   tests/test_oracle_vs_torch.py bounds the difference (fp32 round-off only).
 *
 * Build: see oracle/c/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "det_math.h"

#define ORC_ACT_NONE 0
#define ORC_ACT_TANH 1

/* ------------------------------------------------------------------------------------------------
 * weight_norm fold (old-style torch.nn.utils.weight_norm, dim=0): w[r,:] = v[r,:] * (g[r]/||v[r,:]||)
 * rows = size of dim 0 (Cout for Conv1d, Cin for ConvTranspose1d), inner = product of other dims.
 * [upstream dac/nn/layers.py WNConv1d / WNConvTranspose1d]
 * ---------------------------------------------------------------------------------------------- */
void orc_weight_norm(const float* v, const float* g, float* w, int rows, int inner)
{
    for (int r = 0; r < rows; ++r) {
        const float* vr = v + (size_t)r * inner;
        float ss = 0.0f;
        for (int i = 0; i < inner; ++i) ss = om_fma(vr[i], vr[i], ss);
        float scale = g[r] / sqrtf(ss);
        for (int i = 0; i < inner; ++i) w[(size_t)r * inner + i] = vr[i] * scale;
    }
}

/* elementwise snake over [B,C,T] with per-channel alpha */
void orc_snake(const float* x, const float* alpha, float* y, int B, int C, int T)
{
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* xr = x + ((size_t)b * C + c) * T;
            float* yr = y + ((size_t)b * C + c) * T;
            float a = alpha[c];
            for (int t = 0; t < T; ++t) yr[t] = om_snake(xr[t], a);
        }
}

static inline float orc_epilogue(float acc, float bias, const float* res, size_t off,
                                 const float* alpha_out, int co, int act)
{
    float v = acc + bias;
    if (res) v = v + res[off];
    if (alpha_out) v = om_snake(v, alpha_out[co]);
    if (act == ORC_ACT_TANH) v = om_tanh(v);
    return v;
}

int orc_conv1d_out_len(int Tin, int ks, int stride, int dil, int pad)
{
    int span = Tin + 2 * pad - dil * (ks - 1) - 1;
    if (span < 0) return 0;
    return span / stride + 1;
}

/* ------------------------------------------------------------------------------------------------
 * conv1d, layout x[B,Cin,Tin], w[Cout,Cin,ks], y[B,Cout,Tout]
 *   alpha_in  : if non-NULL, snake(x, alpha_in[ci]) is applied to the input first (Snake1d before conv)
 *   bias      : may be NULL (treated as 0)
 *   residual  : if non-NULL, [B,Cout,Tout] added after bias (ResidualUnit skip)
 *   alpha_out : if non-NULL, snake applied to the result (the NEXT layer's Snake1d, fused)
 *   act       : ORC_ACT_TANH -> tanh on the result (decoder tail)
 * ---------------------------------------------------------------------------------------------- */
void orc_conv1d(const float* x, const float* w, const float* bias, float* y,
                int B, int Cin, int Tin, int Cout, int ks, int stride, int dil, int pad,
                const float* alpha_in, const float* residual, const float* alpha_out, int act)
{
    int Tout = orc_conv1d_out_len(Tin, ks, stride, dil, pad);
    if (Tout <= 0) return;
    /* padded, (optionally) snaked copy of one batch element: xp[ci][pad + t] */
    int need = (Tout - 1) * stride + dil * (ks - 1) + 1;   /* last index touched + 1 */
    int Tp = need > Tin + pad ? need : Tin + pad;
    Tp += 64;
    float* xp = (float*)malloc((size_t)Cin * Tp * sizeof(float));
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for
        for (int ci = 0; ci < Cin; ++ci) {
            float* row = xp + (size_t)ci * Tp;
            memset(row, 0, (size_t)Tp * sizeof(float));
            const float* xr = x + ((size_t)b * Cin + ci) * Tin;
            if (alpha_in) { float a = alpha_in[ci]; for (int t = 0; t < Tin; ++t) row[pad + t] = om_snake(xr[t], a); }
            else memcpy(row + pad, xr, (size_t)Tin * sizeof(float));
        }
#pragma omp parallel for schedule(dynamic, 1)
        for (int co = 0; co < Cout; ++co) {
            enum { TB = 64 };
            float acc[TB];
            const float* wr = w + (size_t)co * Cin * ks;
            float bv = bias ? bias[co] : 0.0f;
            for (int t0 = 0; t0 < Tout; t0 += TB) {
                int nt = Tout - t0 < TB ? Tout - t0 : TB;
                for (int i = 0; i < TB; ++i) acc[i] = 0.0f;
                if (stride == 1 && nt == TB) {
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float* row = xp + (size_t)ci * Tp + t0;
                        for (int kk = 0; kk < ks; ++kk) {
                            float wv = wr[ci * ks + kk];
                            const float* src = row + kk * dil;
                            for (int i = 0; i < TB; ++i) acc[i] = om_fma(wv, src[i], acc[i]);
                        }
                    }
                } else {
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float* row = xp + (size_t)ci * Tp;
                        for (int kk = 0; kk < ks; ++kk) {
                            float wv = wr[ci * ks + kk];
                            for (int i = 0; i < nt; ++i)
                                acc[i] = om_fma(wv, row[(size_t)(t0 + i) * stride + kk * dil], acc[i]);
                        }
                    }
                }
                size_t off = ((size_t)b * Cout + co) * Tout + t0;
                for (int i = 0; i < nt; ++i)
                    y[off + i] = orc_epilogue(acc[i], bv, residual, off + i, alpha_out, co, act);
            }
        }
    }
    free(xp);
}

int orc_conv_transpose1d_out_len(int Tin, int ks, int stride, int pad)
{
    return (Tin - 1) * stride - 2 * pad + ks;
}

/* ------------------------------------------------------------------------------------------------
 * conv_transpose1d, x[B,Cin,Tin], w[Cin,Cout,ks] (torch layout), y[B,Cout,Tout]
 *   y[co,to] = bias + sum_ci sum_{pos : kk = to + pad - pos*stride in [0,ks)} w[ci,co,kk] * xs[ci,pos]
 *   chain order: ci ascending, then pos ascending.
 * ---------------------------------------------------------------------------------------------- */
void orc_conv_transpose1d_op(const float* x, const float* w, const float* bias, float* y,
                             int B, int Cin, int Tin, int Cout, int ks, int stride, int pad, int output_padding,
                             const float* alpha_in, const float* alpha_out);

void orc_conv_transpose1d(const float* x, const float* w, const float* bias, float* y,
                          int B, int Cin, int Tin, int Cout, int ks, int stride, int pad,
                          const float* alpha_in, const float* alpha_out)
{
    orc_conv_transpose1d_op(x, w, bias, y, B, Cin, Tin, Cout, ks, stride, pad, 0, alpha_in, alpha_out);
}

/* output_padding (torch.nn.ConvTranspose1d): `output_padding` more samples at the END of every row -- the same sum evaluated at
 * those positions (only input positions that exist contribute).  The upstream DAC repository head is believed to pass
 * output_padding = stride % 2 in its DecoderBlock (release 1.0.0, which the restatement follows by default, does not). */
void orc_conv_transpose1d_op(const float* x, const float* w, const float* bias, float* y,
                             int B, int Cin, int Tin, int Cout, int ks, int stride, int pad, int output_padding,
                             const float* alpha_in, const float* alpha_out)
{
    int Tout = orc_conv_transpose1d_out_len(Tin, ks, stride, pad) + output_padding;
    if (Tout <= 0) return;
    float* xs = (float*)malloc((size_t)Cin * Tin * sizeof(float));
    for (int b = 0; b < B; ++b) {
        const float* xb = x + (size_t)b * Cin * Tin;
        if (alpha_in) {
#pragma omp parallel for
            for (int ci = 0; ci < Cin; ++ci) {
                float a = alpha_in[ci];
                for (int t = 0; t < Tin; ++t) xs[(size_t)ci * Tin + t] = om_snake(xb[(size_t)ci * Tin + t], a);
            }
        } else memcpy(xs, xb, (size_t)Cin * Tin * sizeof(float));
#pragma omp parallel for schedule(dynamic, 1)
        for (int co = 0; co < Cout; ++co) {
            float bv = bias ? bias[co] : 0.0f;
            for (int to = 0; to < Tout; ++to) {
                /* pos range: kk = to + pad - pos*stride in [0, ks) */
                int num = to + pad;
                int pos_hi = num / stride;                       /* kk >= 0 */
                int lo_num = num - (ks - 1);                     /* kk <= ks-1 */
                int pos_lo = lo_num <= 0 ? 0 : (lo_num + stride - 1) / stride;
                if (pos_hi > Tin - 1) pos_hi = Tin - 1;
                float acc = 0.0f;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float* wr = w + ((size_t)ci * Cout + co) * ks;
                    const float* xr = xs + (size_t)ci * Tin;
                    for (int pos = pos_lo; pos <= pos_hi; ++pos)
                        acc = om_fma(wr[num - pos * stride], xr[pos], acc);
                }
                float v = acc + bv;
                if (alpha_out) v = om_snake(v, alpha_out[co]);
                y[((size_t)b * Cout + co) * Tout + to] = v;
            }
        }
    }
    free(xs);
}

/* ------------------------------------------------------------------------------------------------
 * ResidualVQEMA (reference's own RVQ).
 *   _nearest_l2: argmax_k( x.e_k - 0.5*||e_k||^2 )     Training/compare_dacvsproposal_5.py:253-255
 *   forward    : per book  idx; q = emb[idx]; q_sum = q_sum + (q - residual) + residual;
 *                residual = residual - q                 ...:256-265 ; n_books_use ...6_latency.py:417-435
 * x[N,D] (row = one token), books[nb][K][D]; idx_out[nb_use][N] (int32), qsum_out[N,D].
 * Ties: lowest index (torch.argmax on CPU returns the first maximal element).
 * ---------------------------------------------------------------------------------------------- */
static void half_sq_norms(const float* emb, int K, int D, float* hn)
{
    for (int k = 0; k < K; ++k) {
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s = om_fma(emb[(size_t)k * D + d], emb[(size_t)k * D + d], s);
        hn[k] = 0.5f * s;
    }
}

static int nearest_l2(const float* x, const float* emb, const float* hn, int K, int D, float* best_out)
{
    int best = 0; float bs = -INFINITY;
    for (int k = 0; k < K; ++k) {
        float dot = 0.0f;
        const float* e = emb + (size_t)k * D;
        for (int d = 0; d < D; ++d) dot = om_fma(x[d], e[d], dot);
        float sc = dot - hn[k];
        if (sc > bs || k == 0) { bs = sc; best = k; }
    }
    if (best_out) *best_out = bs;
    return best;
}

void orc_rvq_ema_forward(const float* x, const float* books, int nb_use, int K, int D, int N,
                         int32_t* idx_out, float* qsum_out, float* residual_out)
{
    float* hn = (float*)malloc((size_t)nb_use * K * sizeof(float));
    for (int bk = 0; bk < nb_use; ++bk) half_sq_norms(books + (size_t)bk * K * D, K, D, hn + (size_t)bk * K);
#pragma omp parallel for
    for (int n = 0; n < N; ++n) {
        float res[512], qs[512];
        for (int d = 0; d < D; ++d) { res[d] = x[(size_t)n * D + d]; qs[d] = 0.0f; }
        for (int bk = 0; bk < nb_use; ++bk) {
            const float* emb = books + (size_t)bk * K * D;
            int id = nearest_l2(res, emb, hn + (size_t)bk * K, K, D, NULL);
            if (idx_out) idx_out[(size_t)bk * N + n] = id;
            for (int d = 0; d < D; ++d) {
                float q = emb[(size_t)id * D + d];
                qs[d] = (qs[d] + (q - res[d])) + res[d];
                res[d] = res[d] - q;
            }
        }
        for (int d = 0; d < D; ++d) qsum_out[(size_t)n * D + d] = qs[d];
        if (residual_out) for (int d = 0; d < D; ++d) residual_out[(size_t)n * D + d] = res[d];
    }
    free(hn);
}

/* top-1 / top-2 score margin of every query against one book (used to classify near-ties) */
void orc_rvq_margins(const float* x, const float* emb, int K, int D, int N, float* margin_out)
{
    float* hn = (float*)malloc((size_t)K * sizeof(float));
    half_sq_norms(emb, K, D, hn);
#pragma omp parallel for
    for (int n = 0; n < N; ++n) {
        float b1 = -INFINITY, b2 = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float dot = 0.0f;
            for (int d = 0; d < D; ++d) dot = om_fma(x[(size_t)n * D + d], emb[(size_t)k * D + d], dot);
            float sc = dot - hn[k];
            if (sc > b1) { b2 = b1; b1 = sc; } else if (sc > b2) b2 = sc;
        }
        margin_out[n] = b1 - b2;
    }
    free(hn);
}

/* ------------------------------------------------------------------------------------------------
 * ResidualVQEMA.ema_step                                   Training/compare_dacvsproposal_5.py:266-277
 *   every book is matched against the SAME X (the reference never subtracts the residual here);
 *   counts = bincount(idx); sums.index_add_(0, idx, X) (n ascending);
 *   for used codes: emb = decay*emb + (1-decay)*(sums/(counts+1e-9)).
 * books updated IN PLACE.
 * ---------------------------------------------------------------------------------------------- */
void orc_rvq_ema_step(const float* X, float* books, int nb, int K, int D, int N, float decay,
                      int32_t* idx_out /* [nb][N] or NULL */)
{
    float* hn = (float*)malloc((size_t)K * sizeof(float));
    float* sums = (float*)malloc((size_t)K * D * sizeof(float));
    float* counts = (float*)malloc((size_t)K * sizeof(float));
    int32_t* idx = (int32_t*)malloc((size_t)N * sizeof(int32_t));
    float omd = (float)(1.0 - (double)decay);   /* python: (1.0 - self.decay) in double, then cast */
    for (int bk = 0; bk < nb; ++bk) {
        float* emb = books + (size_t)bk * K * D;
        half_sq_norms(emb, K, D, hn);
#pragma omp parallel for
        for (int n = 0; n < N; ++n) idx[n] = nearest_l2(X + (size_t)n * D, emb, hn, K, D, NULL);
        memset(sums, 0, (size_t)K * D * sizeof(float));
        memset(counts, 0, (size_t)K * sizeof(float));
        for (int n = 0; n < N; ++n) {
            counts[idx[n]] += 1.0f;
            for (int d = 0; d < D; ++d) sums[(size_t)idx[n] * D + d] += X[(size_t)n * D + d];
            if (idx_out) idx_out[(size_t)bk * N + n] = idx[n];
        }
        for (int k = 0; k < K; ++k) {
            if (!(counts[k] > 0.0f)) continue;
            float den = counts[k] + 1e-9f;
            for (int d = 0; d < D; ++d) {
                float mean = sums[(size_t)k * D + d] / den;
                emb[(size_t)k * D + d] = decay * emb[(size_t)k * D + d] + omd * mean;
            }
        }
    }
    free(hn); free(sums); free(counts); free(idx);
}

/* ------------------------------------------------------------------------------------------------
 * DAC ResidualVectorQuantize.forward (eval mode), restated from the upstream architecture
 * [dac/nn/quantize.py VectorQuantize.forward/decode_latents, ResidualVectorQuantize.forward]:
 *   per stage i < n_q:
 *     z_e   = in_proj_i(residual)                   (1x1 conv C->Dc, weights already WN-folded; the C-long
 *                                                    sum is taken as 16 block partials of C/16 channels)
 *     e     = z_e / max(||z_e||_2, 1e-12)           (F.normalize over the Dc axis, per token)
 *     c_k   = cb_i[k] / max(||cb_i[k]||_2, 1e-12)
 *     dist  = (sum e^2 - 2*(e.c_k)) + sum c_k^2 ;  idx = argmax(-dist) (first max)
 *     z_q_i = out_proj_i( z_e + (cb_i[idx] - z_e) ) (straight-through form, RAW codebook row)
 *     z_q  += z_q_i ; residual -= z_q_i
 * z[B,C,T]; in_w[nq][Dc][C], in_b[nq][Dc], cb[nq][K][Dc], out_w[nq][C][Dc], out_b[nq][C]
 * outputs: zq[B,C,T], codes[B,nq_use,T] (int32), latents[B,nq_use*Dc,T]
 * ---------------------------------------------------------------------------------------------- */
void orc_dac_rvq(const float* z, const float* in_w, const float* in_b, const float* cb,
                 const float* out_w, const float* out_b,
                 int B, int C, int T, int nq_use, int K, int Dc,
                 float* zq, int32_t* codes, float* latents)
{
    float* cbn = (float*)malloc((size_t)nq_use * K * Dc * sizeof(float));
    float* cn2 = (float*)malloc((size_t)nq_use * K * sizeof(float));
    for (int i = 0; i < nq_use; ++i)
        for (int k = 0; k < K; ++k) {
            const float* r = cb + ((size_t)i * K + k) * Dc;
            float ss = 0.0f;
            for (int d = 0; d < Dc; ++d) ss = om_fma(r[d], r[d], ss);
            float den = fmaxf(sqrtf(ss), 1e-12f);
            float s2 = 0.0f;
            for (int d = 0; d < Dc; ++d) {
                float v = r[d] / den;
                cbn[((size_t)i * K + k) * Dc + d] = v;
                s2 = om_fma(v, v, s2);
            }
            cn2[(size_t)i * K + k] = s2;
        }
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            float* res = (float*)malloc((size_t)C * sizeof(float));
            float* acc = (float*)malloc((size_t)C * sizeof(float));
            float ze[64], e[64], pre[64];
            for (int c = 0; c < C; ++c) { res[c] = z[((size_t)b * C + c) * T + t]; acc[c] = 0.0f; }
            for (int i = 0; i < nq_use; ++i) {
                for (int d = 0; d < Dc; ++d) {
                    /* blocked order: 16 block-partials over C/16 contiguous channels each (fma chain from 0),
                     * summed in block order, then + bias */
                    const float* wr = in_w + ((size_t)i * Dc + d) * C;
                    const int cb_ = C / 16;
                    float a = 0.0f;
                    for (int g = 0; g < 16; ++g) {
                        float p = 0.0f;
                        for (int c = g * cb_; c < (g + 1) * cb_; ++c) p = om_fma(wr[c], res[c], p);
                        a = g == 0 ? p : a + p;
                    }
                    ze[d] = a + in_b[(size_t)i * Dc + d];
                    latents[((size_t)b * nq_use * Dc + (size_t)i * Dc + d) * T + t] = ze[d];
                }
                float ss = 0.0f;
                for (int d = 0; d < Dc; ++d) ss = om_fma(ze[d], ze[d], ss);
                float den = fmaxf(sqrtf(ss), 1e-12f);
                float en2 = 0.0f;
                for (int d = 0; d < Dc; ++d) { e[d] = ze[d] / den; en2 = om_fma(e[d], e[d], en2); }
                int best = 0; float bs = -INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float* ck = cbn + ((size_t)i * K + k) * Dc;
                    float dot = 0.0f;
                    for (int d = 0; d < Dc; ++d) dot = om_fma(e[d], ck[d], dot);
                    float dist = (en2 - 2.0f * dot) + cn2[(size_t)i * K + k];
                    float sc = -dist;
                    if (sc > bs || k == 0) { bs = sc; best = k; }
                }
                codes[((size_t)b * nq_use + i) * T + t] = best;
                const float* raw = cb + ((size_t)i * K + best) * Dc;
                for (int d = 0; d < Dc; ++d) pre[d] = ze[d] + (raw[d] - ze[d]);
                for (int c = 0; c < C; ++c) {
                    const float* wr = out_w + ((size_t)i * C + c) * Dc;
                    float a = 0.0f;
                    for (int d = 0; d < Dc; ++d) a = om_fma(wr[d], pre[d], a);
                    float zqi = a + out_b[(size_t)i * C + c];
                    acc[c] = acc[c] + zqi;
                    res[c] = res[c] - zqi;
                }
            }
            for (int c = 0; c < C; ++c) zq[((size_t)b * C + c) * T + t] = acc[c];
            free(res); free(acc);
        }
    free(cbn); free(cn2);
}

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis of x[B,C,T] (what TokenNorm and CrossPredictor.ln_* do after
 * their permute(0,2,1): Training/compare_dacvsproposal_5.py:223-225,237-238), eps = 1e-5, biased var.
 *   mean = (sum_c x)/C (c ascending) ; var = (sum_c (x-mean)^2)/C ; y = (x-mean)/sqrt(var+eps)*gamma+beta
 * optional fused post-ops used by the AR glue: tanh, then multiply by `post_scale` (pass 1.0 for none).
 * ---------------------------------------------------------------------------------------------- */
void orc_layernorm_c(const float* x, const float* gamma, const float* beta, float* y,
                     int B, int C, int T, float eps, int do_tanh, float post_scale)
{
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const float* xb = x + (size_t)b * C * T + t;
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s = s + xb[(size_t)c * T];
            float mean = s / (float)C;
            float v = 0.0f;
            for (int c = 0; c < C; ++c) { float d = xb[(size_t)c * T] - mean; v = om_fma(d, d, v); }
            float rstd = 1.0f / sqrtf(v / (float)C + eps);
            for (int c = 0; c < C; ++c) {
                float o = om_fma((xb[(size_t)c * T] - mean) * rstd, gamma[c], beta[c]);
                if (do_tanh) o = om_tanh(o);
                if (do_tanh || post_scale != 1.0f) o = post_scale * o;
                y[((size_t)b * C + c) * T + t] = o;
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * Multi-head cross attention core of CrossPredictor.forward
 * (Training/compare_dacvsproposal_5.py:239-242):  softmax(Q K^T / sqrt(dh)) V per head.
 * Q[B,C,Tq], K,V[B,C,Tk] channel-major (C = H*dh, head h owns channels h*dh..), ctx[B,C,Tq].
 *   s_j = (sum_d Q[d]*K_j[d], d ascending) / sqrt(dh) ; m = max_j s_j ; p_j = exp(s_j - m) ;
 *   l = sum_j p_j (j ascending) ; ctx[d] = sum_j (p_j / l) * V_j[d] (j ascending). Tk == 0 -> ctx = 0.
 * ---------------------------------------------------------------------------------------------- */
void orc_attention(const float* Q, const float* Kx, const float* V, float* ctx,
                   int B, int H, int dh, int Tq, int Tk)
{
    int C = H * dh;
    float rs = sqrtf((float)dh);
#pragma omp parallel for collapse(3)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < Tq; ++i) {
                float p[64];
                const float* q = Q + ((size_t)b * C + (size_t)h * dh) * Tq + i;
                const float* kb = Kx + ((size_t)b * C + (size_t)h * dh) * Tk;
                const float* vb = V + ((size_t)b * C + (size_t)h * dh) * Tk;
                float m = -INFINITY;
                for (int j = 0; j < Tk; ++j) {
                    float a = 0.0f;
                    for (int d = 0; d < dh; ++d) a = om_fma(q[(size_t)d * Tq], kb[(size_t)d * Tk + j], a);
                    p[j] = a / rs;
                    m = fmaxf(m, p[j]);
                }
                float l = 0.0f;
                for (int j = 0; j < Tk; ++j) { p[j] = om_exp(p[j] - m); l = l + p[j]; }
                for (int j = 0; j < Tk; ++j) p[j] = p[j] / l;
                for (int d = 0; d < dh; ++d) {
                    float a = 0.0f;
                    for (int j = 0; j < Tk; ++j) a = om_fma(p[j], vb[(size_t)d * Tk + j], a);
                    ctx[((size_t)b * C + (size_t)h * dh + d) * Tq + i] = a;
                }
            }
}

/* elementwise helpers used by the Python glue of the oracle */
void orc_gelu(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_gelu(x[i]); }
void orc_tanh(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_tanh(x[i]); }
void orc_sin(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_sin(x[i]); }
void orc_sin_turns(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_sin_turns(x[i]); }
void orc_sin2_turns(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_sin2_turns(x[i]); }
void orc_exp(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_exp(x[i]); }
void orc_erf(const float* x, float* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = om_erf(x[i]); }

/* ------------------------------------------------------------------------------------------------
 * align_by_xcorr (Evaluation/dac_vcpwq_proposed6_latency.py:164-202): integer shift s in [-max_shift, max_shift]
 * maximising c(s) = sum(r_seg * e_seg) with  s<0: r[-s:], e[:T+s] ; s>0: r[:T-s], e[s:T] ; s=0: r, e  (r, e same
 * length T, as the caller crop_match()es them first).  First maximum in ascending s (strict `c > best`), start value
 * -1e18.  Each c(s) is one fp32 fma chain in sample order.  corr_out[2*max_shift+1], returns best shift.
 * ---------------------------------------------------------------------------------------------- */
int orc_align_xcorr(const float* r, const float* e, int T, int max_shift, float* corr_out)
{
    int best_s = 0; float best = -1e18f;
    for (int s = -max_shift; s <= max_shift; ++s) {
        int n = T - (s < 0 ? -s : s);
        float c = 0.0f;
        int valid = n > 0;
        if (valid) {
            const float* rp = s < 0 ? r - s : r;
            const float* ep = s > 0 ? e + s : e;
            for (int i = 0; i < n; ++i) c = om_fma(rp[i], ep[i], c);
        }
        if (corr_out) corr_out[s + max_shift] = valid ? c : 0.0f;
        if (valid && c > best) { best = c; best_s = s; }
    }
    return best_s;
}

/* ------------------------------------------------------------------------------------------------
 * Backward pieces (decoder input-gradient, weights frozen).
 *   orc_mul_dsnake : g[b,c,t] *= d snake(x[b,c,t])/dx  (+ residual), the Snake1d backward fused behind a dgrad conv
 *   orc_mul_dtanh  : g *= (1 - y^2)                                   (decoder tail tanh)
 * The dgrad convs themselves are orc_conv1d with flipped / transposed weights (see oracle.py).
 * ---------------------------------------------------------------------------------------------- */
void orc_mul_dsnake(const float* g, const float* x, const float* alpha, const float* residual, float* out,
                    int B, int C, int T)
{
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            size_t o = ((size_t)b * C + c) * T;
            float a = alpha[c];
            for (int t = 0; t < T; ++t) {
                float v = g[o + t] * om_dsnake(x[o + t], a);
                if (residual) v = v + residual[o + t];
                out[o + t] = v;
            }
        }
}

void orc_mul_dtanh(const float* g, const float* y, float* out, size_t n)
{
    for (size_t i = 0; i < n; ++i) out[i] = g[i] * om_fma(-y[i], y[i], 1.0f);
}


/* ---- polyphase sinc resampler (row f3) ----------------------------------------------------------------------------
 * torchaudio.transforms.Resample as the reference calls it (Training/compare_dacvsproposal_5.py:110-113,
 * Evaluation/dac_vcpwq_proposed6_latency.py:151-156): y[n*newf + p] = sum_k kern[p][k] * xpad[n*orig + k], xpad = x
 * zero-padded by `width` on the left; one fp32 fma chain per output, k ascending.  kern[newf][ks] comes from
 * oracle.py:resample_kernel (float64 design, rounded to fp32). */
void orc_resample(const float* x, const float* kern, float* y, int B, int L, int Lout, int orig, int newf, int width, int ks)
{
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < Lout; ++m) {
            const int n = m / newf, p = m - n * newf;
            float acc = 0.0f;
            for (int k = 0; k < ks; ++k) {
                const int j = n * orig + k - width;
                if (j >= 0 && j < L) acc = om_fma(kern[(size_t)p * ks + k], x[(size_t)b * L + j], acc);
            }
            y[(size_t)b * Lout + m] = acc;
        }
}
