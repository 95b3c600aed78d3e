// conv_lat.hip -- the LATENCY form of the implicit-GEMM conv: v_mfma_f32_16x16x4_f32, one wave per 16 x 16 output tile,
// operands straight from L2 into a register ring -- no LDS, no barrier.
//
// Why it exists (profiles/r05_*_before): at the reference's own operating points -- one segment (latency protocol,
// Evaluation/dac_vcpwq_proposed6_latency.py:489-525) or a batch of six (Evaluation/compare_dacvsproposal_5_eval.py:487-489,
// Training/compare_dacvsproposal_5.py:62) -- the layers at the latent rate (T = 75) and the predictor's GEMMs over one AR chunk
// (16 tokens per segment) have far fewer 32 x 32 wave tiles than the chip has SIMDs (1 024).  What such a launch waits for is
// then ONE dependent accumulation chain, and its length is fixed by the arithmetic contract (one fp32 fma chain over K in
// order; split-K would be another sum): v_mfma_f32_32x32x2_f32 advances that chain by 2 k per 64 cycles (32 cycles per k),
// v_mfma_f32_16x16x4_f32 by 4 k per 40 cycles (10 cycles per k; MI355X guide, "FP32-input MFMA") -- the same exact k-ordered
// fma chain, 3.2x shorter in time, on tiles a quarter the size (4x the waves to spread over idle SIMDs).
// K is walked exactly as conv1d_mfma.hpp walks it -- k = ci * KS + tap ascending -- so the results are the same bits
// (tests/test_gpu_parity_ops.py runs every latency-form shape against the C oracle).
//
// Lane l = (r = l & 15, q = l >> 4) feeds A[row r][k = 4 s + q] and B[k = 4 s + q][column r] of k-step s and owns
// D[rows 4 q .. 4 q + 3][column r].  A group = CG input channels = GS = CG * KS / 4 k-steps; a lane's GS activation offsets
// inside a group (channel, tap -> element) are the same for every group, so they are computed once; the weight / activation
// bases advance by a uniform stride per group.  Two groups of operand registers alternate: while group g is multiplied the
// 2 GS loads of group g + 1 are in flight (GS = 16-32 k-steps x 40 cycles of cover for the L2 / HBM latency of a lone wave).
#include "conv_dispatch.hpp"

namespace mvq {

template <int KS, int STRIDE, int DIL, int CG>
__global__ __launch_bounds__(256) void conv1d_lat_kernel(const ConvArgs a)
{
    static_assert((CG * KS) % 4 == 0, "a group is a whole number of 16x16x4 k-steps");
    constexpr int GS = CG * KS / 4;
    static_assert(GS <= 32, "offset mask is 32 bits");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / a.n_tiles, tn = blockIdx.x - b * a.n_tiles;
    const int n0 = tn * 16;
    const int m0 = (blockIdx.y * 4 + wave) * 16;
    if (m0 >= a.Mrows) return;                                   // no barrier anywhere: a wave may leave on its own

    const int n = n0 + r;
    const float* const xb = a.x + (size_t)b * a.Cin * a.Tin;
    int voff[GS];
    unsigned vmask = 0;
#pragma unroll
    for (int p = 0; p < GS; ++p) {
        const int kl = 4 * p + q, cl = kl / KS, kk = kl - cl * KS;
        const int t = n * STRIDE + kk * DIL - a.pad;
        const bool ok = n < a.Ncols && t >= 0 && t < a.Tin;
        voff[p] = cl * a.Tin + (ok ? t : 0);
        vmask |= ok ? (1u << p) : 0u;
    }
    const bool all_ok = __all(vmask == ((GS == 32) ? 0xffffffffu : ((1u << GS) - 1u)));
    const size_t aoff = (size_t)q * a.Mpad + m0 + r;             // rows up to Mpad - 1 exist in the packed image (zero rows)
    const size_t a_group = (size_t)GS * 4 * a.Mpad, x_group = (size_t)CG * a.Tin;
    const int ng = a.Cin / CG;

    struct Grp { float a[GS], b[GS]; };
    auto load = [&](Grp& G, int g) __attribute__((always_inline)) {
        const float* wg = a.wp + (size_t)g * a_group + aoff;
        const float* xg = xb + (size_t)g * x_group;
#pragma unroll
        for (int p = 0; p < GS; ++p) G.a[p] = wg[(size_t)p * 4 * a.Mpad];
#pragma unroll
        for (int p = 0; p < GS; ++p) G.b[p] = xg[voff[p]];
    };
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto compute = [&](Grp& G) __attribute__((always_inline)) {
        if (!all_ok) {                                           // conv zero padding / columns past the row: boundary tiles only
#pragma unroll
            for (int p = 0; p < GS; ++p) G.b[p] = ((vmask >> p) & 1u) ? G.b[p] : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < GS; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(G.a[p], G.b[p], acc, 0, 0, 0);
    };
    // two register groups: while one is multiplied, the loads of the next are in flight.  The prefetch index is clamped instead of
    // guarded (one wasted reload of the last group at the very end), so the loop body has no branch around its loads, and the
    // scheduling barriers keep each group's loads together in front of the other group's MFMAs.
    Grp G0, G1;
    load(G0, 0);
    int g = 0;
    for (; g + 1 < ng; g += 2) {
        load(G1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(G0);
        __builtin_amdgcn_sched_barrier(0);
        load(G0, g + 2 < ng ? g + 2 : g + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(G1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g < ng) compute(G0);

    // ---- epilogue: the operations of conv1d_mfma_body's, in its order, one element at a time
    const bool ups = a.up_s > 1;                                 // polyphase ConvTranspose1d: GEMM row = co * S + phase
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * q + i;
        if (m >= a.Mrows || n >= a.Ncols) continue;
        int ch = m, t = n;
        if (ups) {
            ch = m / a.up_s;
            t = n * a.up_s + (m - ch * a.up_s) - a.up_p;
            if (t < 0 || t >= a.Tout) continue;
        }
        const size_t off = ((size_t)b * a.Cout + ch) * a.Tout + t;
        float v = acc[i] + (a.bias ? a.bias[ch] : 0.0f);
        if (a.dsn_src) { const float ad = a.dsn_alpha[ch]; v = v * det_dsnake(a.dsn_src[off], ad, 1.0f / (ad + 1e-9f)); }
        if (a.residual) v = v + a.residual[off];
        const bool tail = a.tvalid && t >= a.tvalid;
        if (a.y2) { const float a2 = a.alpha2[ch]; a.y2[off] = tail ? 0.0f : det_snake(v, a2, 1.0f / (a2 + 1e-9f)); }
        if (a.alpha_out) { const float al = a.alpha_out[ch]; v = det_snake(v, al, 1.0f / (al + 1e-9f)); }
        if (a.act == 1) v = det_tanh(v);
        if (a.act == 2) v = det_gelu(v);
        a.y[off] = tail ? 0.0f : v;
    }
}

template <int KS, int STRIDE, int DIL, int CG>
static hipError_t launch_lat(const ConvArgs& a_in, hipStream_t s)
{
    ConvArgs a = a_in;
    if (a.Cin % CG != 0 || a.Mpad % 16 != 0) return hipErrorInvalidValue;
    if (a.name_out) {
        snprintf(a.name_out, a.name_len, "conv1d_lat_kernel<%d, %d, %d, %d>", KS, STRIDE, DIL, CG);
        return hipSuccess;
    }
    a.n_tiles = (a.Ncols + 15) / 16;
    dim3 grid((unsigned)(a.n_tiles * a.B), (unsigned)((a.Mrows + 63) / 64));
    int pi = -1;
    if (prof_enabled()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv1d_lat_kernel<%d, %d, %d, %d>", KS, STRIDE, DIL, CG);
        int lim = a.Ncols;                                        // algorithmic columns, as launch_conv1d_mfma counts them
        if (a.up_s > 1) lim = a.Ncols - 1;
        else if (a.tvalid > 0) lim = a.tvalid;
        pi = prof_begin(nm, 2.0 * a.Cin * KS * a.Mrows * (double)lim * a.B, s);
    }
    hipLaunchKernelGGL((conv1d_lat_kernel<KS, STRIDE, DIL, CG>), grid, dim3(256), 0, s, a);
    prof_end(pi, s);
    return hipGetLastError();
}

// 16 x 16 tiles the launch would have; the latency form is chosen while they are few enough to sit (about) two per SIMD -- beyond
// that the operand traffic of tiles that share nothing through LDS (512 bytes per k-step per wave out of L2) is the bound and
// the LDS-tiled kernels win.  MVQ_LAT_MAX_TILES overrides the threshold (A/B runs; 0 switches the form off).
static long lat_max_tiles()
{
    static const long v = [] {
        const char* e = getenv("MVQ_LAT_MAX_TILES");
        if (e) note_env_override(MVQ_BF_ENV_LAT_TILES);
        return e ? atol(e) : 2048L;
    }();
    return v;
}

bool conv_lat_wanted(const ConvArgs& a)
{
    if (a.alpha_in || a.tper || a.vp_seg || a.up_per_out || a.n_base || a.n_tiles_max) return false;
    if (a.B <= 0 || a.Ncols <= 0) return false;
    const long tiles = (long)a.B * ((a.Ncols + 15) / 16) * ((a.Mrows + 15) / 16);
    return tiles <= lat_max_tiles();
}

hipError_t launch_conv_lat(const ConvArgs& a, int ks, int stride, int dil, hipStream_t s)
{
    // channels per group chosen so that a group is 16-32 k-steps (the prefetch distance); every width of the model divides
    if (a.up_s > 1) return (ks == 2 && stride == 1 && dil == 1) ? launch_lat<2, 1, 1, 64>(a, s) : hipErrorInvalidValue;
    if (stride == 1) {
        if (ks == 1 && dil == 1) return a.Cin % 64 == 0 ? launch_lat<1, 1, 1, 64>(a, s) : launch_lat<1, 1, 1, 32>(a, s);
        if (ks == 3 && dil == 1) return launch_lat<3, 1, 1, 32>(a, s);
        if (ks == 7 && dil == 1) return launch_lat<7, 1, 1, 16>(a, s);
        if (ks == 7 && dil == 3) return launch_lat<7, 1, 3, 16>(a, s);
        if (ks == 7 && dil == 9) return launch_lat<7, 1, 9, 16>(a, s);
        return hipErrorInvalidValue;
    }
    if (dil != 1 || ks != 2 * stride) return hipErrorInvalidValue;
    switch (stride) {
        case 2: return launch_lat<4, 2, 1, 32>(a, s);
        case 4: return launch_lat<8, 4, 1, 16>(a, s);
        case 5: return launch_lat<10, 5, 1, 8>(a, s);
        case 8: return launch_lat<16, 8, 1, 8>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mvq
