/*
 * mvq.h -- C ABI of libmvq_hip.so: the MI355X (gfx950) encode -> vector-quantise -> decode hot path of
 * aymenboudhina/Multimodal_VQVAE_compression_audio_tactile.
 *
 * The reference has no FFI: its boundary is the Python object surface its scripts touch
 * (SURVEY.md section 8b).  Each entry point below names the reference call it replaces
 * (paths relative to /root/reference).  All pointers are DEVICE pointers (HIP), all tensors fp32,
 * contiguous, channel-major [B, C, T] exactly as torch lays them out, indices int32 on this ABI
 * (the Python mirror widens to int64).  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Inputs are borrowed and never written; outputs must be caller-allocated.
 * Every function returns 0 on success or a negative MVQ_E* code; mvq_last_error() gives the text.
 * No function allocates, frees or synchronises (safe to capture into a hipGraph), except
 * mvq_device_query() and the mvq_profile_* pair.
 *
 * Arithmetic contract: every dot product is one fp32 fma chain in the order "input channel ascending,
 * then tap ascending" starting from +0.0f, followed by  + bias, + residual, snake, tanh  (each optional).
 * On gfx950 that is what a k-ordered v_mfma_f32_32x32x2_f32 accumulation yields, so results are
 * bit-reproducible and comparable bit for bit with oracle/c/oracle.c.
 */
#ifndef MVQ_H
#define MVQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVQ_OK 0
#define MVQ_EINVAL (-1)      /* bad shape / argument                     */
#define MVQ_EUNSUPPORTED (-2)/* shape outside what the kernels cover     */
#define MVQ_EHIP (-3)        /* HIP runtime error (launch failure, ...)  */

#define MVQ_ACT_NONE 0
#define MVQ_ACT_TANH 1
#define MVQ_ACT_GELU 2   /* exact-erf GELU (nn.GELU() of CrossPredictor.ffn); MFMA-tiled, non-transposed shapes only */

/* ABI version.  3 (round 5): whole-stack entry points (mvq_encoder_fwd_f32, mvq_decoder_fwd_f32, mvq_decoder_fwd_saving_f32,
 * mvq_decoder_bwd_input_f32 and the mvq_stack handle).  2 (round 4): mvq_rvq_ema_step_f32 takes the larger 16-byte-aligned scratch that
 * mvq_rvq_ema_step_scratch_bytes() reports (version 1 documented nb*B*T int32), mvq_profile_end2() reports truncation,
 * mvq_build_flags() exists. */
int mvq_abi_version(void);
const char* mvq_last_error(void);
/* How this library was built / is being driven: 0 for a product build with no A/B override in the environment.
 *   0x1 MVQ_TIMING_BUILD, 0x2 MVQ_EXP (pieces of kernels compiled out: results are WRONG by construction), 0x4 MVQ_KGROUP /
 *   MVQ_KPREFETCH, 0x8 MVQ_NO_RES_PREFETCH, 0x10 MVQ_ASM_READS > 1 -- compile-time switches of tools/conv_microbench.py's timing
 *   builds (they do not compile without -DMVQ_TIMING_BUILD); 0x100 MVQ_NO_DMA, 0x200 MVQ_ROWFAST_MAX_KB, 0x400
 *   MVQ_NO_TOKEN_RVQ, 0x800 MVQ_LN_TILE32, 0x1000 MVQ_LAT_MAX_TILES, 0x2000 MVQ_NO_DAC_RVQ_LAT, 0x4000 MVQ_NO_LN_LAT, 0x8000 MVQ_SMALL_TILE_MAX, 0x20 MVQ_F16_NO192 / MVQ_F16_NO_WIDE (opt-in f16x3 mode) -- environment
 *   overrides present (results stay correct, timings are not the product's);
 *   0x10000 (informational, not refused) the compiler-scheduled operand loop MVQ_ASM_READS=0.
 * The Python mirror refuses to load a library whose low 16 bits are non-zero unless MVQ_ALLOW_TIMING_BUILD=1;
 * bench.py prints the value in its line. */
unsigned mvq_build_flags(void);

/* Per-launch HIP-event profiler (measurement aid of bench.py; no reference counterpart).  Between mvq_profile_begin() and
 * mvq_profile_end() every conv / residual-unit kernel launch of the library is bracketed by a HIP event pair recorded on
 * the stream it is launched on, together with the launch's ALGORITHMIC FLOPs (2 * Cin * ks * valid rows * valid columns *
 * batch: zero-padded rows and tail tiles are not counted).  mvq_profile_end() waits for the events (it synchronises) and
 * returns one entry per kernel instantiation -- the name is the one rocprofv3 prints -- with the summed duration, FLOPs and
 * launch count.  Not capturable into a hipGraph while enabled; one profiling session per process at a time (global state,
 * not thread-safe). */
typedef struct mvq_profile_entry {
    char kernel[96];
    double seconds;
    double flops;
    int launches;
} mvq_profile_entry;
int mvq_profile_begin(void);
/* writes at most max_entries rows; *n_entries = the number of rows WRITTEN (instantiations beyond the buffer are dropped).
 * Returns MVQ_EHIP when an event could not be created or recorded during the session (first such error; nothing is written). */
int mvq_profile_end(mvq_profile_entry* out, int max_entries, int* n_entries);
/* same, and *n_total = the number of instantiations the session recorded: *n_total > *n_entries means the table was truncated */
int mvq_profile_end2(mvq_profile_entry* out, int max_entries, int* n_entries, int* n_total);
/* creates the event pairs of n_launches launches ahead of time: call it before a timed region so that no hipEventCreate
 * falls inside it (events are pooled and reused across sessions). */
int mvq_profile_reserve(int n_launches);
/* fills cu_count / lds_bytes / gcn arch name ("gfx950...") of the current device; synchronous. */
int mvq_device_query(int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len);

/* ---- one-off weight preparation (model load time) -------------------------------------------- */

/* torch.nn.utils.weight_norm fold, dim 0:  w[r,:] = v[r,:] * (g[r] / ||v[r,:]||_2).
 * Replaces the implicit fold inside every upstream dac WNConv1d / WNConvTranspose1d forward that
 * dac.DAC.load(...) modules perform (Training/compare_dacvsproposal_5.py:329-338). */
int mvq_weight_norm_f32(const float* v, const float* g, float* w, int rows, int inner, void* stream);

/* Packed (K-major, zero-padded) weight image used by mvq_conv1d_f32:  wp[(ci*ks + kk) * Mpad + co].
 * Query the size (in floats) first, then pack from the torch layout w[Cout, Cin, ks]. */
size_t mvq_conv1d_packed_floats(int cin, int cout, int ks);
int mvq_conv1d_pack_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream);

/* Same for ConvTranspose1d with kernel = 2*stride (every upstream DecoderBlock): torch layout
 * w[Cin, Cout, 2*stride] -> polyphase image wp[(ci*2 + j) * Mpad + (co*stride + r)]. */
size_t mvq_conv_transpose1d_packed_floats(int cin, int cout, int stride);
int mvq_conv_transpose1d_pack_f32(const float* w, float* wp, int cin, int cout, int stride, void* stream);

/* ---- conv stack primitives ---------------------------------------------------------------------- */

/* y = act( snake_out( conv1d( snake_in(x) ) + bias + residual ) ).
 * Replaces torch F.conv1d + the surrounding Snake1d / residual add / tanh inside dac Encoder / Decoder /
 * ResidualUnit, the 1x1 proj_down / proj_up convs (Training/compare_dacvsproposal_5.py:287-288) and the
 * bias-free nn.Linear layers of CrossPredictor (...:226-228) when applied on channel-major tensors.
 *   x[B,Cin,Tin], wp from mvq_conv1d_pack_f32, bias[Cout]|NULL, alpha_in[Cin]|NULL (Snake1d before
 *   the conv), residual[B,Cout,Tout]|NULL, alpha_out[Cout]|NULL (Snake1d after), act MVQ_ACT_*.
 *   Tout = (Tin + 2*pad - dil*(ks-1) - 1)/stride + 1.   y[B,Cout,Tout]. */
int mvq_conv1d_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                   const float* residual, const float* alpha_out, float* y,
                   int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act,
                   void* stream);

/* Dual-output forms.  Besides y they write  y2 = snake(v, alpha2)  where v is the value before alpha_out / act
 * (conv + bias + residual): the Snake1d that the NEXT ResidualUnit applies to its input, hoisted into the producer so
 * that a consumer with M/128 row tiles does not re-evaluate it M/128 times per element while staging.  y2[B,Cout,Tout],
 * alpha2[Cout]; pass both or neither (NULL, NULL == the plain entry points above / below). */
int mvq_conv1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                        const float* residual, const float* alpha_out, float* y, float* y2, const float* alpha2,
                        int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act,
                        void* stream);
int mvq_conv_transpose1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                  const float* alpha_out, float* y, float* y2, const float* alpha2,
                                  int batch, int cin, int tin, int cout, int stride, int pad, void* stream);

/* Name of the kernel instantiation the two conv entry points launch for a shape ("conv1d_mfma_kernel<...>" as
 * rocprofv3 prints it, or "conv1d_direct_kernel"): lets bench.py match its HIP-event timings to the trace.  Tile shape
 * depends on batch and length (64 x 64 tiles in the latency regime, 128 x 96 at the latent rate); "same"/DAC padding is
 * assumed.  For transposed convs pass the torch shape (cin, cout, ks = 2*stride, stride). */
int mvq_conv_kernel_name(int batch, int cin, int cout, int ks, int stride, int dil, int transposed, int tin,
                         char* buf, int len);

/* One upstream dac ResidualUnit:  y = snake_next?( x + conv1( snake_b( conv7_dil( snake_a(x) ) + b7 ) ) + b1 ),
 * 7-tap conv with dilation `dil` and padding 3*dil, then a 1x1 conv, both C -> C.  For C in {64, 96, 128} (the
 * high-rate ends of the stacks, where the 1x1 conv is bandwidth-bound on its own) the whole unit is ONE launch and
 * the intermediate never leaves the CU; otherwise two launches through `scratch` (B*C*T floats, see
 * mvq_residual_unit_scratch_floats; may be NULL when that returns 0).  alpha_next: the following Snake1d or NULL. */
size_t mvq_residual_unit_scratch_floats(int batch, int c, int t, int dil);
int mvq_residual_unit_kernel_name(int c, int dil, char* buf, int len);   /* "residual_unit_kernel<...>" or "(two launches)" */
int mvq_residual_unit_f32(const float* x, const float* w7p, const float* b7, const float* alpha_a,
                          const float* alpha_b, const float* w1p, const float* b1, const float* alpha_next,
                          float* y, float* scratch, int batch, int c, int t, int dil, void* stream);

/* ResidualUnit with a pre-snaked input and/or dual output: x_snaked = snake_a(x) (from a producer's y2) or NULL;
 * y2 = snake(y_raw, alpha2) for the next unit or NULL. */
int mvq_residual_unit_dual_f32(const float* x, const float* x_snaked, const float* w7p, const float* b7,
                               const float* alpha_a, const float* alpha_b, const float* w1p, const float* b1,
                               const float* alpha_next, float* y, float* y2, const float* alpha2, float* scratch,
                               int batch, int c, int t, int dil, void* stream);

/* y = snake_out( conv_transpose1d( snake_in(x) ) + bias ), kernel = 2*stride, torch `padding` = pad.
 * Replaces the Snake1d + WNConvTranspose1d at the head of every upstream DecoderBlock.
 *   Tout = (Tin-1)*stride - 2*pad + 2*stride. */
int mvq_conv_transpose1d_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                             const float* alpha_out, float* y,
                             int batch, int cin, int tin, int cout, int stride, int pad, void* stream);

/* ---- vector quantisation ------------------------------------------------------------------------ */

/* ResidualVQEMA.forward (Training/compare_dacvsproposal_5.py:256-265; eval variant with n_books_use
 * Evaluation/dac_vcpwq_proposed6_latency.py:417-435).  z[B,D,T], books[nb_use,K,D] contiguous.
 * Per token and book: idx = argmax_k(res.e_k - 0.5*||e_k||^2) (lowest index on ties), q = e[idx],
 * q_sum = (q_sum + (q - res)) + res, res = res - q.   q_out[B,D,T]; idx_out[nb_use, B*T] may be NULL. */
int mvq_rvq_ema_forward_f32(const float* z, const float* books, float* q_out, int32_t* idx_out,
                            int batch, int dim, int t, int nb_use, int k, void* stream);

/* ResidualVQEMA.ema_step (Training/compare_dacvsproposal_5.py:266-277): every book matched against the
 * SAME tokens X = z_tokens[B,D,T]; used codes move to decay*e + (1-decay)*mean(assigned tokens).
 * books[nb,K,D] updated in place.  scratch: mvq_rvq_ema_step_scratch_bytes() bytes, 16-byte aligned (the per-book
 * assignments nb*B*T int32, then the workspace of a stable counting sort of the token ids by code + the token-major copy
 * of X).  Per (code, dim) the sum runs over the code's OWN tokens in token order -- the additions index_add_ performs, in
 * its order, so the result is deterministic and bit-equal to the sequential loop -- at O(B*T*D) work per book instead of
 * O(K*D*B*T).  B*T < 2^24 (counts stay exact in fp32), K <= 16384. */
size_t mvq_rvq_ema_step_scratch_bytes(int batch, int t, int nb, int k, int dim);
int mvq_rvq_ema_step_f32(const float* z_tokens, float* books, void* scratch,
                         int batch, int dim, int t, int nb, int k, float decay, void* stream);

/* dac ResidualVectorQuantize.forward in eval mode, first nq_use stages
 * (`qa, *_ = self.A_QUANT(za)` Training/compare_dacvsproposal_5.py:295; `mdl.encode(x, n_quantizers)`
 * Evaluation/compare_dacvsproposal_5_eval.py:369).  Weights already weight-norm folded:
 *   in_w[nq,Dc,C], in_b[nq,Dc], codebook[nq,K,Dc], out_w[nq,C,Dc], out_b[nq,C].
 *   z[B,C,T] -> zq[B,C,T], codes[B,nq_use,T] (int32), latents[B,nq_use*Dc,T].   C in {256, 512, 1024}, Dc = 8.
 * Summation order of in_proj (the only C-long reduction here): 16 block partials over C/16 contiguous channels
 * (each an fma chain from +0), added in block order, then + bias; out_proj: fma chain over Dc, then + bias. */
int mvq_dac_rvq_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                    const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                    int batch, int c, int t, int nq_use, int k, int dc, void* stream);
/* The same with upstream's TRAIN-mode quantiser dropout (dac ResidualVectorQuantize.forward under `net.train()`,
 * Training/compare_dacvsproposal_5.py:401): every stage runs, item b's zq sums only stages < nq_item[b]
 * (nq_item[batch] int32 on the device, NULL = no dropout); codes / latents of all nq_use stages are produced. */
int mvq_dac_rvq_items_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                          const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                          const int32_t* nq_item, int batch, int c, int t, int nq_use, int k, int dc, void* stream);
/* One-off preparation of the quantiser's codebooks (model load time, like the weight-norm fold): cb_normalised[nq,K,Dc] =
 * F.normalize(codebook) and cb_norm2[nq,K] = its squared norms, by the same divisions / fma chains the search kernel would
 * otherwise redo in every block of every call (a fifth of its vector instructions).  mvq_dac_rvq_prepared_f32 is
 * mvq_dac_rvq_items_f32 with those two tensors handed in (both NULL = normalise in the kernel): bit-identical outputs. */
int mvq_dac_rvq_prepare_f32(const float* codebook, float* cb_normalised, float* cb_norm2, int nq, int k, int dc, void* stream);
int mvq_dac_rvq_prepared_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                             const float* cb_normalised, const float* cb_norm2,
                             const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                             const int32_t* nq_item, int batch, int c, int t, int nq_use, int k, int dc, void* stream);


/* ---- predictor / glue primitives (CrossPredictor, TokenNorm, PosEnc1D) --------------------------- */

/* y = post_scale * tanh?( LayerNorm_C(x + pe?) ), normalising over the channel axis (eps, biased variance).
 * Element (b, c, t) of x and y lives at  b*stride_b + c*stride_c + t ; pass stride_b = stride_c = 0 for a
 * contiguous [B,C,T] tensor.  (The AR loop keeps chunk tensors token-folded as [C, B*Tc]: stride_b = Tc,
 * stride_c = B*Tc, so that the predictor GEMMs see N = B*Tc contiguous columns.)
 * pe[max_len, C] (row = position) or NULL; positions are 0..T-1 (PosEnc1D restarts per call).
 * Replaces PosEnc1D + ln_q/ln_kv/ffn[0] (Training/compare_dacvsproposal_5.py:236-238,243) and
 * torch.tanh(TokenNorm(r)) * scale (...:313-315). */
int mvq_layernorm_c_f32(const float* x, const float* pe, const float* gamma, const float* beta, float* y,
                        int batch, int c, int t, size_t stride_b, size_t stride_c,
                        float eps, int do_tanh, float post_scale, void* stream);
/* The same on x - sub (element-wise, same addressing): the AR residual r = zt - z_pred followed by tanh(TokenNorm(r))*scale
 * (Training/compare_dacvsproposal_5.py:312-315) in one launch; sub may be NULL. */
int mvq_layernorm_c_sub_f32(const float* x, const float* sub, const float* pe, const float* gamma, const float* beta, float* y,
                            int batch, int c, int t, size_t stride_b, size_t stride_c,
                            float eps, int do_tanh, float post_scale, void* stream);

/* softmax(Q K^T / sqrt(dh)) V per head (Training/compare_dacvsproposal_5.py:239-242).  Q, ctx: (b,c,i) at
 * b*q_stride_b + c*q_stride_c + i ; K, V: (b,c,j) at b*k_stride_b + c*k_stride_c + j (0,0 = contiguous).
 * Tk may be 0 (ctx = 0).  Tq, Tk <= 64 and dh*(Tq+2Tk)+Tq*Tk <= 16384 floats (one head slice lives in LDS). */
int mvq_attention_f32(const float* q, const float* k, const float* v, float* ctx,
                      int batch, int heads, int dh, int tq, int tk,
                      size_t q_stride_b, size_t q_stride_c, size_t k_stride_b, size_t k_stride_c, void* stream);

/* y = gelu_erf(x) elementwise (nn.GELU() in CrossPredictor.ffn). */
int mvq_gelu_f32(const float* x, float* y, size_t n, void* stream);

/* y(b,c,t) = a(b,c,t) - b(b,c,t)   and   y(b,c,t) = a(b,c,t)   on [batch, c, n] views, each tensor with its
 * own (batch, channel) element strides and time stride 1: chunk slicing `zt[..., s:e] - z_pred`,
 * `z_run[..., s:e] = z_hat` (Training/compare_dacvsproposal_5.py:312,319) and the fold/unfold between
 * [B,C,T] and the token-folded [C, B*n] chunk layout. */
int mvq_sub3d_f32(const float* a, size_t a_sb, size_t a_sc, const float* b, size_t b_sb, size_t b_sc,
                  float* y, size_t y_sb, size_t y_sc, int batch, int c, int n, void* stream);
int mvq_copy3d_f32(const float* a, size_t a_sb, size_t a_sc, float* y, size_t y_sb, size_t y_sc,
                   int batch, int c, int n, void* stream);

/* ---- evaluation metrics next to the path ---------------------------------- */

/* align_by_xcorr (Evaluation/dac_vcpwq_proposed6_latency.py:164-202): for every integer shift s in
 * [-max_shift, max_shift] the correlation c(s) = sum(r_seg * e_seg) of two equal-length signals ref[t], est[t]
 * (s<0: ref[-s:], est[:t+s]; s>0: ref[:t-s], est[s:]), each one fp32 fma chain in sample order; best_shift = first
 * maximum in ascending s (strict >, start -1e18).  corr[2*max_shift+1], scratch[2*max_shift+1] int32, best_shift[1]
 * (device).  One launch pair instead of 2*max_shift+1 reductions with a host comparison each. */
int mvq_align_xcorr_f32(const float* ref, const float* est, int t, int max_shift, float* corr, int32_t* scratch,
                        int32_t* best_shift, void* stream);
/* The same for `batch` pairs in ONE launch pair (rows of pitch t; corr / scratch [batch][2*max_shift+1], best_shift[batch]):
 * psnr_3k_aligned_batch (Evaluation/compare_dacvsproposal_5_eval.py:212-223) aligns every item of a batch.  Per item the
 * chains are those of mvq_align_xcorr_f32, so the shifts are identical. */
int mvq_align_xcorr_batch_f32(const float* ref, const float* est, int batch, int t, int max_shift, float* corr, int32_t* scratch,
                              int32_t* best_shift, void* stream);

/* ---- backward w.r.t. the decoder input (weights are frozen in the reference) ----- */

/* Input-gradient of a conv is a conv with a transformed weight image, so it runs on the same MFMA kernels:
 *   forward Conv1d w[Cout,Cin,ks], stride 1:           wp = mvq_conv1d_pack_dgrad_f32(w)            (flip + transpose)
 *   forward ConvTranspose1d w[Cin,Cout,ks], stride s:  wp = mvq_conv_transpose1d_pack_dgrad_f32(w)  (a strided conv)
 * size query: mvq_conv1d_dgrad_packed_floats(cin, cout, ks) with cin = channels of the forward INPUT. */
size_t mvq_conv1d_dgrad_packed_floats(int cin, int cout, int ks);
int mvq_conv1d_pack_dgrad_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream);
int mvq_conv_transpose1d_pack_dgrad_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream);

/* gx[B,cin,tin] = ( dgrad-conv(gy[B,cout,tout]) ) * d snake(dsnake_src)/dx  + residual
 * (cin, tin, cout, tout, ks, stride, dil, pad describe the FORWARD layer: stride 1 for Conv1d, the up-sampling stride
 * for a ConvTranspose1d).  dsnake_src[B,cin,tin] = saved input of the Snake1d in front of the forward layer and
 * dsnake_alpha[cin] its alpha (both NULL: no Snake in front); residual[B,cin,tin] = gradient arriving over a skip
 * connection or NULL.  Chain order: forward output channel ascending, then tap ascending. */
int mvq_conv1d_dgrad_f32(const float* gy, const float* wp_dgrad, const float* dsnake_src, const float* dsnake_alpha,
                         const float* residual, float* gx,
                         int batch, int cin, int tin, int cout, int tout, int ks, int stride, int dil, int pad,
                         void* stream);

/* Backward of the reference's own trainable modules (CrossPredictor, TokenNorm, tanh*scale; Training/
 * compare_dacvsproposal_5.py:222-244,313-315 under `scaler.scale(total).backward()`, ...:393).  Same addressing as the
 * forward entry points.  Checked against torch autograd (fp32 tolerance), not part of the bit-exact forward contract.
 *   layernorm_c_bwd : gx (may be NULL) and dgamma/dbeta (ACCUMULATED into) from g; stats[2*B*T] scratch (mu, rstd)
 *   gelu_bwd        : gx = g * gelu'(x)
 *   scale_tanh      : y = scale * tanh(u);  _bwd: gu = g*scale*(1-tanh(u)^2), partial[n_partial] block sums of g*tanh(u)
 *                     (their total is d/dscale), n_partial <= 4096
 *   attention_bwd   : gq, gk, gv from g = dL/dctx (Tq, Tk <= 32)
 *   mul_scaled      : out = a*b*scale (nn.Dropout on ctx, ...:242: the keep-mask is drawn by the caller; also its backward)
 *   transpose2d     : out[c][r] = in[r][c]      rowsum: out[r] (+)= sum_c in[r][c]   (weight / bias gradients: the
 *                     weight gradient dW = g x^T over the token axis runs on mvq_conv1d_f32 with K = tokens) */
int mvq_layernorm_c_bwd_f32(const float* x, const float* pe, const float* gamma, const float* g, float* gx,
                            float* dgamma, float* dbeta, float* stats, int batch, int c, int t,
                            size_t stride_b, size_t stride_c, float eps, void* stream);
int mvq_gelu_bwd_f32(const float* x, const float* g, float* gx, size_t n, void* stream);
int mvq_scale_tanh_f32(const float* u, float scale, float* y, size_t n, void* stream);
int mvq_scale_tanh_bwd_f32(const float* u, const float* g, float scale, float* gu, float* partial, int n_partial, size_t n,
                           void* stream);
int mvq_attention_bwd_f32(const float* q, const float* k, const float* v, const float* g, float* gq, float* gk, float* gv,
                          int batch, int heads, int dh, int tq, int tk,
                          size_t q_stride_b, size_t q_stride_c, size_t k_stride_b, size_t k_stride_c, void* stream);
int mvq_mul_scaled_f32(const float* a, const float* b, float scale, float* out, size_t n, void* stream);
int mvq_transpose2d_f32(const float* in, float* out, int rows, int cols, void* stream);
int mvq_rowsum_f32(const float* in, float* out, int rows, int cols, int accumulate, void* stream);

/* Training losses: safe_l1, MultiResSTFTLoss, MelCosineLoss of
 * Training/compare_dacvsproposal_5.py:150-211 and their gradient w.r.t. the predicted waveform.  The STFT is a GEMM:
 * frames[n_fft, cols] (windowed, reflect-padded, center=True) times a real DFT basis -> spec[2*fp, cols] on
 * mvq_conv1d_f32 (k = 1), rows [0,f) = Re, rows [fp, fp+f) = Im, f = n_fft/2+1, fp = f rounded up to 32; its gradient is
 * the transposed GEMM followed by overlap_add.  Column = half*batch*nframes + b*nframes + n, half 0 = prediction,
 * half 1 = target, ncols >= 2*batch*nframes, nframes = 1 + t/hop.  These entry points are the HBM-bound glue:
 *   stft_frames       out[f][col0 + b*nframes + n] = window[f] * finite_or_zero(x[b][reflect(n*hop + f - n_fft/2)])
 *   spec_mag          mag[k][c] = max(|spec|, eps)   (rows f..fp-1 zero)
 *   spec_loss_partial partial[3][batch][p]: sums of (X-Y)^2, Y^2, |X-Y| per item (spectral convergence and log-free
 *                     magnitude L1 of MultiResSTFTLoss.forward, ...:166-170)
 *   spec_grad         g[2*fp][batch*nframes] = dL/d(Re,Im) of the prediction from coef_a[b]*(X-Y) + coef_b*sign(X-Y)
 *                     + extra[k][c] (extra: gradient arriving from the mel branch, may be NULL)
 *   overlap_add       dy[b][t] += sum of window*dframes over the frames (and reflections) covering sample t
 *   l1_loss           partial[p] block sums of |y - tgt| (finite_or_zero applied); dy += coef*sign(y - tgt) if dy
 *   mel_max / mel_cos / mel_max_grad : per-item max of the mel spectrogram (MelCosineLoss._mel_mag's amax), the per-frame
 *                     cosine of log(M/max + eps) and its gradient w.r.t. the prediction's mel magnitudes (dmel[n_mels]
 *                     [batch*nframes], dden per column; mel_max_grad routes the max's gradient to its argmax element);
 *                     use_log = 0 gives the cosine of the max-normalised LINEAR mel frames = stsim_batch
 *                     (Evaluation/compare_dacvsproposal_5_eval.py:142-177; forward only)
 * Checked against torch autograd on the restated losses and fixture G7 (fp32 tolerance). */
int mvq_stft_frames_f32(const float* x, const float* window, float* out, int batch, int t, int n_fft, int hop, int nframes,
                        size_t ncols, size_t col0, void* stream);
int mvq_spec_mag_f32(const float* spec, float* mag, int f, int fp, size_t ncols, float eps, void* stream);
int mvq_spec_loss_partial_f32(const float* mag, float* partial, int p, int f, int batch, int nframes, size_t ncols, void* stream);
int mvq_spec_grad_f32(const float* spec, const float* mag, const float* coef_a, float coef_b, const float* extra, float* g,
                      int f, int fp, int batch, int nframes, size_t ncols, float eps, void* stream);
int mvq_overlap_add_f32(const float* dframes, const float* window, float* dy, int batch, int t, int n_fft, int hop, int nframes,
                        void* stream);
int mvq_l1_loss_f32(const float* y, const float* tgt, float* partial, int p, float* dy, float coef, size_t n, void* stream);
int mvq_mel_max_f32(const float* mel, float* maxv, int* argmax, int n_mels, int batch, int nframes, size_t ncols, void* stream);
int mvq_mel_cos_f32(const float* mel, const float* maxv, float* cosv, float* dmel, float* dden, float coef, int n_mels, int batch,
                    int nframes, size_t ncols, float eps, int use_log, void* stream);
int mvq_mel_max_grad_f32(const float* dden, const float* maxv, const int* argmax, float* dmel, int batch, int nframes, float eps,
                         void* stream);

/* Zero-padded rows.  A layer whose natural length is not a multiple of 4 (the decoder's 2 999-sample block) would take
 * the 4-byte load / store paths.  Instead its tensors are allocated with rows rounded up to a multiple of 4 and the tail kept
 * at ZERO, which is exactly the zero padding a conv sees beyond the end of its input: every kernel then runs its 16-byte
 * paths and the results of the true columns are bit-identical.  `tvalid` = true length: output columns >= tvalid are
 * written as zeros (0 = all columns are data).  conv_transpose1d: `tout_rows` = row length of y (0 = natural; smaller when
 * the input carried a zero tail; up to `pad` larger with tvalid <= natural).  MFMA-tiled shapes only. */
int mvq_conv1d_padded_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                          const float* residual, const float* alpha_out, float* y, float* y2, const float* alpha2,
                          int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act, int tvalid,
                          void* stream);
int mvq_residual_unit_padded_f32(const float* x, const float* x_snaked, const float* w7p, const float* b7,
                                 const float* alpha_a, const float* alpha_b, const float* w1p, const float* b1,
                                 const float* alpha_next, float* y, float* y2, const float* alpha2, float* scratch,
                                 int batch, int c, int t, int dil, int tvalid, void* stream);
int mvq_conv_transpose1d_padded_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                    const float* alpha_out, float* y, float* y2, const float* alpha2,
                                    int batch, int cin, int tin, int cout, int stride, int pad, int tout_rows, int tvalid,
                                    void* stream);
/* The same with torch's ``output_padding`` (0 <= output_padding < stride, <= pad): the natural row length grows by that many
 * samples at the END of each row -- the same sum evaluated there.  For the DecoderBlock variant with
 * ``output_padding = stride % 2`` (believed to be upstream DAC's repository head; release 1.0.0 -- the default of this
 * package and of the oracle -- passes none): 75 tokens then decode to 24 000 samples instead of 23 992. */
int mvq_conv_transpose1d_op_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                const float* alpha_out, float* y, float* y2, const float* alpha2,
                                int batch, int cin, int tin, int cout, int stride, int pad, int output_padding, int tout_rows,
                                int tvalid, void* stream);

/* VIRTUALLY PACKED latent-rate rows (round 4): the same tiling as the packed layout -- seg_per_row items side by side in one
 * GEMM row, so a 128-column tile holds live columns only -- but with NO repacked copy of the input or the output: the LDS-DMA
 * source of every 16-byte piece is remapped (virtual column c -> item c / per_in, position c mod per_in; positions >= tin_rows
 * read zeros) and the epilogue stores column n to (item n / tout_rows, position n mod tout_rows) of the ordinary tensor.
 * Used for the encoder's last strided conv (512 -> 1024, k 16, s 8, T 600 -> 75: 78 % -> 98 % live columns) and the k3 conv
 * behind it (upstream dac Encoder tail; Training/compare_dacvsproposal_5.py:294,296).
 *   x[batch, cin, tin_rows]: tin_valid data columns then a zero tail (caller's contract, as for the zero-padded rows below);
 *   y[batch, cout, tout_rows]: conv_out_len(tin_valid) valid columns, the rest written as zeros.  Requirements: tin_rows,
 *   tout_rows, per_in multiples of 4; per_in == stride * tout_rows; per_in - tin_valid >= max(pad, right overhang of the last
 *   output); no Snake on load (the DMA cannot transform what it copies).  Same fma chains as mvq_conv1d_f32 on each item. */
int mvq_conv1d_vpacked_f32(const float* x, const float* wp, const float* bias, const float* residual, const float* alpha_out,
                           float* y, float* y2, const float* alpha2, int batch, int cin, int tin_rows, int tin_valid, int cout,
                           int ks, int stride, int dil, int pad, int act, int seg_per_row, int per_in, int tout_rows, void* stream);

/* PACKED latent-rate rows (round 3).  At the latent rate a segment is 75 columns wide: one 96- or 128-column tile per segment
 * leaves 22-41 % of the MFMAs on padding.  The decoder's first two layers therefore run on rows that hold `seg_per_row` segments
 * at a period of `seg_period` columns (a multiple of 4), `seg_valid` of them data and the rest ZEROS -- the zero padding each
 * conv would see between neighbours, so every data column is computed exactly as in the unpacked layer (same fma chains).
 *   mvq_conv1d_packed_rows_f32: a stride-1 'same' conv (2*pad == (ks-1)*dil, pad <= seg_period - seg_valid) x[rows,cin,L] ->
 *     y[rows,cout,L], L = seg_per_row * seg_period; gap columns of the output are written as zeros (also in y2).
 *   mvq_conv_transpose1d_packed_rows_f32: ConvTranspose1d(kernel 2*stride, stride, pad) reading such rows and writing the
 *     UNPACKED y[batch_out, cout, (seg_valid-1)*stride - 2*pad + 2*stride]: segment g of row r is batch item r*seg_per_row + g.
 * Replaces nothing new in the reference: it is `T_DEC`'s `model.0` / `model.1.block.1` (Training/compare_dacvsproposal_5.py:322). */
int mvq_conv1d_packed_rows_f32(const float* x, const float* wp, const float* bias, const float* residual, const float* alpha_out,
                               float* y, float* y2, const float* alpha2, int rows, int cin, int cout, int ks, int dil, int pad,
                               int act, int seg_per_row, int seg_period, int seg_valid, void* stream);
int mvq_conv_transpose1d_packed_rows_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                         const float* alpha_out, float* y, float* y2, const float* alpha2, int rows, int cin,
                                         int cout, int stride, int pad, int seg_per_row, int seg_period, int seg_valid,
                                         int batch_out, void* stream);

/* OPT-IN, NON-PARITY arithmetic mode "bf16x6" for the wide ResidualUnits' 7-tap dilated convs (upstream dac ResidualUnit,
 * called through Training/compare_dacvsproposal_5.py:294-296,322): the default path above is an exact k-ordered fp32 fma chain;
 * these three entry points are the one deliberate departure and nothing calls them unless the caller asks for the mode
 * (Python: multimodal_vqvae_compression_audio_tactile_amd.set_arith("bf16x6")).  Every fp32 operand is split into three
 * bf16 pieces (24 significant bits) and a product is six piece products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
 * fp32-ACCURATE (the dropped terms are < 2^-24 |ab|) but summed in another order than the contract, so NOT bit-identical to the
 * oracle; it earns no parity claim and is reported as its own bench line (dtype "bf16x6").
 *   mvq_bf16x3_split_f32        x[batch, c, t] fp32 -> xs (mvq_bf16x3_split_bytes bytes): [batch][c/8][3 pieces][t][8] bf16; c % 8 == 0
 *   mvq_conv1d_k7_pack_bf16x3   folded weights w[cout, cin, 7] fp32 -> wq (mvq_conv1d_k7_bf16x3_packed_bytes bytes);
 *                               cout % 128 == 0 or cout % 96 == 0, cin % 16 == 0
 *   mvq_conv1d_k7_bf16x6_f32    y[batch, cout, t] = snake_out(conv7_dil(xs) + bias), 'same' padding 3 * dil, dil in {1, 3, 9};
 *                               xs already carries the input Snake (the producer's dual output); tvalid as for the
 *                               zero-padded rows above (0 = every column is data).
 * Training config (Decoder.forward_saving / backward_input; all optional, NULL otherwise):
 *   y2 != NULL       dual output: y = conv + bias (the pre-activation the backward saves), y2 = snake_out(y) (what the 1x1 conv reads)
 *   pack ... dgrad=1 the INPUT-GRADIENT image of a forward weight w[c, m, 7] (cout = the forward Cin, cin = the forward Cout):
 *                    W'[m][c][k] = w[c][m][6 - k]; the same conv call on the output gradient then is the gradient w.r.t. the input
 *   dsn_src, dsn_alpha, residual   the input-gradient epilogue of mvq_conv1d_dgrad_f32: v = acc * d snake(dsn_src)/dx + residual */
size_t mvq_bf16x3_split_bytes(int batch, int c, int t);
int mvq_bf16x3_split_f32(const float* x, void* xs, int batch, int c, int t, void* stream);
size_t mvq_conv1d_k7_bf16x3_packed_bytes(int cout, int cin);
int mvq_conv1d_k7_pack_bf16x3(const float* w, void* wq, int cout, int cin, int dgrad, void* stream);
int mvq_conv1d_k7_bf16x6_f32(const void* xs, const void* wq, const float* bias, const float* alpha_out, float* y, float* y2,
                             const float* dsn_src, const float* dsn_alpha, const float* residual,
                             int batch, int cin, int t, int cout, int dil, int tvalid, void* stream);

/* The same opt-in departure with TWO fp16 pieces per operand and THREE piece products ("f16x3"): half the matrix work of bf16x6.
 * fp16 has 11 significant bits but a 5-bit exponent, so every ITEM of the activations and the weight tensor are first scaled by
 * the power of two that puts their largest magnitude into [2^13, 2^14) (exact); the scales are kept as the float bit patterns of
 * those maxima (xamax[batch], wamax[1], written by the split / pack calls) and undone in the conv's epilogue.  Error per product
 * <= ~3 x 2^-22 (dropped h1 g1 term + the two representation errors): about twice the rounding error of the exact fp32 chain, i.e.
 * fp32-class; the maxima are taken over the FINITE elements, so a NaN / Inf sample contaminates its own receptive field only (as in
 * the exact path) and does not disturb the item's scale; finite magnitudes above the fp32 range of a S are outside its contract.
 *   mvq_f16x2_split_f32       x[batch, c, t] -> xs (4 bytes per element: [batch][c/8][2 pieces][t][8] fp16), xamax[batch]
 *   mvq_conv1d_k7_pack_f16x2  w[cout, cin, 7] -> wq (mvq_conv1d_k7_f16x2_packed_bytes), wamax[1]
 *   mvq_conv1d_k7_f16x3_f32   as mvq_conv1d_k7_bf16x6_f32 */
int mvq_f16x2_split_f32(const float* x, void* xs, uint32_t* xamax, int batch, int c, int t, void* stream);
size_t mvq_conv1d_k7_f16x2_packed_bytes(int cout, int cin);
int mvq_conv1d_k7_pack_f16x2(const float* w, void* wq, uint32_t* wamax, int cout, int cin, int dgrad, void* stream);
int mvq_conv1d_k7_f16x3_f32(const void* xs, const uint32_t* xamax, const void* wq, const uint32_t* wamax, const float* bias,
                            const float* alpha_out, float* y, float* y2, const float* dsn_src, const float* dsn_alpha,
                            const float* residual, int batch, int cin, int t, int cout, int dil, int tvalid, void* stream);

/* Polyphase sinc resampler: torchaudio.transforms.Resample(orig, new) as the reference
 * calls it on every file (Training/compare_dacvsproposal_5.py:110-113, Evaluation/dac_vcpwq_proposed6_latency.py:151-156).
 * orig/newf are the rates divided by their gcd, kern[newf][ks] the filter bank (ks = 2*width + orig),
 * y[b][n*newf + p] = sum_k kern[p][k] * xpad[b][n*orig + k] with x zero-padded by `width` on the left,
 * len_out <= ceil(newf*len/orig).  One fp32 fma chain per output sample, k ascending (bit-exact vs the oracle). */
int mvq_resample_f32(const float* x, const float* kern, float* y, int batch, int len, int len_out, int orig, int newf,
                     int width, int ks, void* stream);
/* Ragged form (the resample step of psnr_3k_aligned_batch, ...5_eval.py:217-220, after a per-item alignment shift): item b
 * resamples x[b*pitch + off[b] : ... + len[b]]; off / len are DEVICE int32 arrays (they are computed from the device-side
 * shifts, so no host round trip sits between aligning and resampling).  y rows have pitch lout_pitch; samples at and past
 * ceil(newf*len[b]/orig) are written as zeros and that length goes to len_out[b] (may be NULL).  Same chains as above.
 * A slice is clamped to its row on the device (0 <= off, off + len <= pitch) and to lout_pitch output samples. */
int mvq_resample_ragged_f32(const float* x, const float* kern, float* y, const int32_t* off, const int32_t* len, int32_t* len_out,
                            int batch, int pitch, int lout_pitch, int orig, int newf, int width, int ks, void* stream);

/* Optimiser step of the training config (torch.optim.AdamW + clip_grad_norm_, Training/compare_dacvsproposal_5.py:367,394-395):
 *   sumsq_partial : partial[n_partial] block sums of x^2 (their total is the squared gradient norm), n_partial <= 4096
 *   adamw         : decoupled-weight-decay Adam on one tensor, torch's single-tensor operation order; clip_coef (device,
 *                   may be NULL) multiplies the gradient on the fly; `step` is the 1-based step count (bias corrections). */
int mvq_sumsq_partial_f32(const float* x, float* partial, int n_partial, size_t n, void* stream);
int mvq_adamw_f32(float* p, const float* g, float* m, float* v, const float* clip_coef, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);

/* out = g * (1 - y*y): backward of the decoder's final tanh (y = saved output). */
int mvq_mul_dtanh_f32(const float* g, const float* y, float* out, size_t n, void* stream);

/* ---- whole stacks (encoder_fwd, decoder_fwd, decoder_bwd_input) ------------------------------------------
 * One call per stack instead of one per layer: `A_ENC(a)` / `T_ENC(t)` (Training/compare_dacvsproposal_5.py:294,296), `T_DEC(z)`
 * (...:322) and the gradient of T_DEC w.r.t. its input under `scaler.scale(total).backward()` (...:393; weights frozen, ...:283-284).
 *
 * A stack handle holds the launch plan and pointers into a caller-provided WEIGHTS BLOB in which mvq_*_create folds the weight norm
 * and packs every conv once (for a decoder also the input-gradient images), next to copies of the biases and Snake alphas.  The
 * parameters are handed over as device pointers in the order mvq_stack_param_info() reports -- upstream `dac` state-dict names
 * ("block.1.block.0.block.1.weight_v", "model.3.block.1.bias", ...) with their shapes -- so any caller that can read a checkpoint
 * can bind them; they are not referenced after create returns.  create with params == NULL and weights_blob == NULL makes a
 * description-only handle (param_info / weights_bytes / out_len / workspace queries work, fwd does not).
 *
 * A forward / backward call walks the plan the Python mirror used to hold: dual outputs (a producer also emits the Snake its wide
 * consumer needs), one fused launch per ResidualUnit of width 64 / 96 / 128, zero-padded rows where a length is not a multiple of
 * 4, and at throughput batch sizes the virtually packed (encoder tail) / packed (decoder head) latent-rate rows.  Intermediates
 * live in a caller-provided WORKSPACE (mvq_*_workspace_bytes, laid out by a deterministic first-fit arena); nothing is allocated
 * or synchronised, so a call can be captured into a hipGraph.  Results are bit-identical to the per-layer entry points above
 * (tests/test_gpu_stacks.py) and to oracle/c/oracle.c.
 *
 *   mvq_encoder_fwd_f32        x[batch, 1, t]            -> z[batch, d_latent, mvq_encoder_out_len(t)]
 *   mvq_decoder_fwd_f32        z[batch, input_channel, t] -> y[batch, d_out, mvq_decoder_out_len(t)]  (tanh applied)
 *   mvq_decoder_fwd_saving_f32 the same values, keeping every Snake input and the output in `saved` (mvq_decoder_saved_bytes)
 *   mvq_decoder_bwd_input_f32  gz[batch, input_channel, t] = dL/dz from gy = dL/dy and `saved`: the same MFMA conv kernels on flipped /
 *                              transposed weight images with the Snake and tanh derivatives in their epilogues (bit-identical to
 *                              mvq_conv1d_dgrad_f32 layer by layer)
 * mvq_stack_set_plan: batch thresholds of the packed latent-rate forms (0 = keep; defaults 32 / 32) -- A/B measurements only. */
typedef struct mvq_stack mvq_stack;
typedef struct mvq_encoder_desc { int d_model; int n_strides; int strides[8]; int d_latent; } mvq_encoder_desc;     /* dac Encoder(64, [2,4,5,8], 1024) */
typedef struct mvq_decoder_desc { int input_channel; int channels; int n_rates; int rates[8]; int d_out; int output_padding; } mvq_decoder_desc;
int mvq_encoder_create(mvq_stack** out, const mvq_encoder_desc* desc, const float* const* params, void* weights_blob, size_t blob_bytes, void* stream);
int mvq_decoder_create(mvq_stack** out, const mvq_decoder_desc* desc, const float* const* params, void* weights_blob, size_t blob_bytes, void* stream);
void mvq_stack_destroy(mvq_stack* s);
int mvq_stack_param_count(const mvq_stack* s);
int mvq_stack_param_info(const mvq_stack* s, int i, char* name, int name_len, int* dims3);
size_t mvq_stack_weights_bytes(const mvq_stack* s);
int mvq_stack_set_plan(mvq_stack* s, int vpack_min_batch, int pack_min_batch);
int mvq_encoder_out_len(const mvq_stack* s, int t);
size_t mvq_encoder_workspace_bytes(const mvq_stack* s, int batch, int t);
int mvq_encoder_fwd_f32(const mvq_stack* s, const float* x, float* z, void* workspace, size_t workspace_bytes, int batch, int t, void* stream);
int mvq_decoder_out_len(const mvq_stack* s, int t);
size_t mvq_decoder_workspace_bytes(const mvq_stack* s, int batch, int t);      /* covers fwd, fwd_saving and bwd_input */
int mvq_decoder_fwd_f32(const mvq_stack* s, const float* z, float* y, void* workspace, size_t workspace_bytes, int batch, int t, void* stream);
size_t mvq_decoder_saved_bytes(const mvq_stack* s, int batch, int t);
int mvq_decoder_fwd_saving_f32(const mvq_stack* s, const float* z, float* y, void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                               int batch, int t, void* stream);
int mvq_decoder_bwd_input_f32(const mvq_stack* s, const void* saved, size_t saved_bytes, const float* gy, float* gz, void* workspace,
                              size_t workspace_bytes, int batch, int t, void* stream);

/* ---- the auto-regressive chunk loop as ONE launch ------------------------------------------------------------------------------
 * `for s in range(0, Tlat, AR_CHUNK_TOK)` of AllPredAR.forward_step (Training/compare_dacvsproposal_5.py:302-320) ==
 * ProposedEval.encode_latents (Evaluation/dac_vcpwq_proposed6_latency.py:461-477) under no_grad: per 16-token chunk the
 * CrossPredictor on the shift-by-one input, tanh(TokenNorm(zt - z_pred)) * clamp(scale), proj_down, ResidualVQEMA, proj_up, and the
 * write into z_run that the next chunk's predictor reads.  One persistent kernel (csrc/ar_fused.hip): the stages of every chunk
 * are separated by a grid-wide barrier instead of a launch boundary; each stage runs the arithmetic of the stand-alone entry points
 * above (mvq_layernorm_c_sub_f32, mvq_conv1d_f32 with k = 1, mvq_attention_f32, mvq_rvq_ema_forward_f32), so results are
 * bit-identical to the launch-per-stage sequence.  Launched cooperatively (the grid must be resident at once; the barrier gives up
 * after a bounded wait and mvq_ar_check reports it).  Shapes: the reference's (c_lat 1024, FFN 2048, 8 heads, code dim 96, chunks of
 * 16 tokens, K <= 512).  Linear weights are the packed images of mvq_conv1d_pack_f32 (k = 1); k_all / v_all are the audio keys /
 * values of ALL chunks, token-folded [c_lat][batch * t_audio] (column b * t_audio + t) as the per-chunk calls use them
 * (PosEnc restarts per chunk); pe[16][c_lat]; books [books_use][rvq_k][96].  z_run [batch, c_lat, t_lat] is written chunk by chunk;
 * r_tokens [batch, 96, t_lat] (the tokens ema_step sees) and idx_out [books_use, batch, t_lat] (int32) are optional. */
typedef struct mvq_ar_args {
    int batch, t_lat, t_audio, tactile_only, books_use, rvq_k, c_lat, c_ff, code_dim, heads, chunk;
    float ln_eps, tok_eps, scale;
    const float *zt, *k_all, *v_all, *pe;
    const float *lnq_g, *lnq_b, *wq, *wo, *lnf_g, *lnf_b, *w1, *b1, *w3, *b3;      /* CrossPredictor: ln_q, q_proj, out, ffn[0], ffn[1], ffn[3] */
    const float *tok_g, *tok_b, *wd, *bd, *wu, *bu, *books;                       /* TokenNorm, proj_down, proj_up, vq.books */
    float *z_run, *r_tokens;
    int32_t* idx_out;
} mvq_ar_args;
size_t mvq_ar_workspace_bytes(int batch, int t_lat);
int mvq_ar_latents_f32(const mvq_ar_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* the same loop as stream-ordered stand-alone launches from one host call (batch <= 8; capturable; same bits) */
int mvq_ar_latents_staged_f32(const mvq_ar_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* synchronises `stream`, then: MVQ_OK when every grid barrier of the last call on this workspace completed */
int mvq_ar_check(const void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVQ_H */
