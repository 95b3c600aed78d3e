// conv_lat.hip -- the LATENCY form of the implicit-GEMM conv: v_mfma_f32_16x16x4_f32, ONE WAVE per 16 x 16 output tile, operands
// staged through a wave-private LDS double buffer -- no block-level barrier anywhere.
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
// D[rows 4 q .. 4 q + 3][column r].  A group = CG input channels = GS = CG * KS / 4 k-steps.  Per group the wave copies its
// weight slice [4 GS][16 rows] and its activation tile [CG][XT samples] (XT = 15 STRIDE + (KS-1) DIL + 1: every tap of a channel
// reads the same staged row, a tap is an LDS offset) into LDS with 16-byte loads, then feeds the MFMAs with one ds_read_b32 per
// operand.  (The first cut read both operands straight from global memory, one dword per lane per k-step: the CU's single
// address pipeline then cost ~15 cycles per wave-load against 40 per MFMA and bounded the kernel, gpurun_out/r05lat; staged, a
// group costs 9-16 wide loads instead of 2 GS narrow ones.)  The buffers are WAVE-PRIVATE: a wave's LDS operations complete in
// order, so a store followed by a read of the same wave needs no s_barrier, and waves never wait for each other -- one-wave
// workgroups spread over every CU of the chip.  Global loads of group g + 1 are in flight while group g is multiplied.
#include "conv_dispatch.hpp"

namespace mvq {

template <int KS, int STRIDE, int DIL, int CG>
struct LatCfg {
    static constexpr int GS = CG * KS / 4;                               // k-steps per group
    static constexpr int XT = 15 * STRIDE + (KS - 1) * DIL + 1;          // input samples per channel row of a 16-column tile
    static constexpr int XV = (XT + 3 + 3) / 4;                          // 16-byte pieces per row (aligned start, shift <= 3)
    static constexpr int XTP = XV * 4 + 1;                               // LDS row pitch (odd: taps of neighbouring channels on other banks)
    static constexpr int A_FLOATS = GS * 4 * 16;
    static constexpr int X_FLOATS = CG * XTP;
    static constexpr int BUF_FLOATS = (A_FLOATS + X_FLOATS + 3) / 4 * 4;
    static constexpr int A_V = A_FLOATS / 4;                             // 16-byte weight pieces per group
    static constexpr int A_Q = (A_V + 63) / 64;                          // ... per lane
    static constexpr int X_Q = (CG * XV + 63) / 64;                      // 16-byte activation pieces per lane per group
    static constexpr int X_S = (CG * XT + 63) / 64;                      // 4-byte pieces per lane (rows that are not 16-byte aligned)
    static_assert((CG * KS) % 4 == 0, "a group is a whole number of k-steps");
};

template <int KS, int STRIDE, int DIL, int CG, bool VEC>
__device__ __forceinline__ void conv_lat_body(const ConvArgs& a)
{
    using C = LatCfg<KS, STRIDE, DIL, CG>;
    constexpr int GS = C::GS;
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / a.n_tiles, tn = blockIdx.x - b * a.n_tiles;
    const int n0 = tn * 16, m0 = blockIdx.y * 16;
    const int n = n0 + r;
    const int t_in0 = n0 * STRIDE - a.pad;                               // input sample of tile column 0
    const int g_al = VEC ? (t_in0 & ~3) : t_in0;
    const int shift = t_in0 - g_al;
    const int ng = a.Cin / CG;

    // Everything a lane needs to address is the same in every group, so it is computed ONCE: in the K loop a lane's vector
    // instructions are its loads, LDS stores / reads and the MFMAs (the first LDS-staged cut recomputed the piece -> (row,
    // column) maps per group: ~56 vector instructions per MFMA, 140 cycles per k-step instead of 40; gpurun_out/r05lat2).
    constexpr int XN = VEC ? C::X_Q : C::X_S;                            // activation pieces per lane (16-byte or 4-byte)
    constexpr int XTOT = VEC ? CG * C::XV : CG * C::XT;
    unsigned aoff[C::A_Q];                                               // global element offsets of the lane's weight pieces
    unsigned xoff[XN], xlds[XN];                                         // global element offset / LDS float index of its activation pieces
    unsigned xokm = 0;                                                   // bit u: piece u lies inside its row (else: the conv's zero padding)
#pragma unroll
    for (int u = 0; u < C::A_Q; ++u) {
        int e = lane + 64 * u;
        if (C::A_V % 64 != 0) e = e < C::A_V ? e : C::A_V - 1;
        aoff[u] = (unsigned)((e >> 2) * a.Mpad + 4 * (e & 3));
    }
#pragma unroll
    for (int u = 0; u < XN; ++u) {
        int e = lane + 64 * u;
        if (XTOT % 64 != 0) e = e < XTOT ? e : XTOT - 1;
        const int per = VEC ? C::XV : C::XT;
        const int cl = e / per, v = e - cl * per;
        const int t = g_al + (VEC ? 4 * v : v);                          // VEC: a multiple of 4, wholly inside or wholly outside the row
        const bool ok = t >= 0 && t < a.Tin;
        xoff[u] = (unsigned)(cl * a.Tin + (ok ? t : 0));
        xlds[u] = (unsigned)(C::A_FLOATS + cl * C::XTP + (VEC ? 4 * v : v));
        xokm |= ok ? (1u << u) : 0u;
    }
    static_assert(XN <= 32, "piece mask is 32 bits");
    const bool x_all_ok = __all(xokm == (XN == 32 ? 0xffffffffu : (1u << XN) - 1u));
    unsigned boff[GS];                                                   // LDS float index of the lane's B operand of every k-step
#pragma unroll
    for (int p = 0; p < GS; ++p) {
        const int kl = 4 * p + q, cl = kl / KS, kk = kl - cl * KS;
        boff[p] = (unsigned)(C::A_FLOATS + cl * C::XTP + r * STRIDE + kk * DIL + shift);
    }

    const float* wsrc = a.wp + m0;                                       // uniform bases, advanced by a uniform stride per group
    const float* xsrc = a.x + (size_t)b * a.Cin * a.Tin;
    const size_t w_step = (size_t)(GS * 4) * a.Mpad, x_step = (size_t)CG * a.Tin;
    int g_next = 0;                                                      // group the next gload fetches
    struct Stage { f32x4 w[C::A_Q]; f32x4 xq[VEC ? XN : 1]; float xs[VEC ? 1 : XN]; };
    auto gload = [&](Stage& S) __attribute__((always_inline)) {         // global -> staging registers.  Past the last group the same
#pragma unroll                                                           // addresses are fetched again (never used): no branch in the loop
        for (int u = 0; u < C::A_Q; ++u) S.w[u] = *reinterpret_cast<const f32x4*>(wsrc + aoff[u]);
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            if constexpr (VEC) S.xq[u] = *reinterpret_cast<const f32x4*>(xsrc + xoff[u]);
            else S.xs[u] = xsrc[xoff[u]];
        }
        ++g_next;
        const bool more = g_next < ng;
        wsrc += more ? w_step : 0; xsrc += more ? x_step : 0;
    };
    auto lstore = [&](const Stage& S, float* buf) __attribute__((always_inline)) {     // staging registers -> one of the wave's two LDS buffers
#pragma unroll
        for (int u = 0; u < C::A_Q; ++u)
            if (C::A_V % 64 == 0 || lane + 64 * u < C::A_V) *reinterpret_cast<f32x4*>(buf + (lane + 64 * u) * 4) = S.w[u];
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            if (XTOT % 64 != 0 && lane + 64 * u >= XTOT) continue;
            const bool ok = x_all_ok || ((xokm >> u) & 1u);
            if constexpr (VEC) {                                         // odd row pitch: four 4-byte stores
                float* d = buf + xlds[u];
                d[0] = ok ? S.xq[u].x : 0.0f; d[1] = ok ? S.xq[u].y : 0.0f; d[2] = ok ? S.xq[u].z : 0.0f; d[3] = ok ? S.xq[u].w : 0.0f;
            } else buf[xlds[u]] = ok ? S.xs[u] : 0.0f;
        }
    };
    struct Ops { float a[GS], b[GS]; };
    auto lread = [&](Ops& R, const float* buf) __attribute__((always_inline)) {        // LDS -> the operand registers of a whole group
#pragma unroll
        for (int p = 0; p < GS; ++p) { R.a[p] = buf[p * 64 + lane]; R.b[p] = buf[boff[p]]; }   // A[(4 p + q)][r] sits at p * 64 + lane
    };
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    auto mfmas = [&](const Ops& R) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < GS; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(R.a[p], R.b[p], acc, 0, 0, 0);
    };
    // Four-stage pipeline; in iteration g, all in ONE basic block and mutually independent:
    //     global loads of group g + 3 -> S[(g + 1) & 1]            (in flight for a whole iteration before they are stored)
    //     S[g & 1] (group g + 2)      -> LDS buffer g & 1
    //     LDS buffer (g + 1) & 1      -> R[(g + 1) & 1]            (group g + 1, stored one iteration ago)
    //     MFMAs of group g from R[g & 1]
    // A dependent 16x16x4 MFMA issues every 40 cycles; the ~2.5 other instructions per k-step go into those gaps (the scheduling
    // groups below), so the chain runs at its own latency.  Stages that reach past the last group move data nobody reads.
    constexpr int N_DSW = C::A_Q + (VEC ? 2 * XN : XN);                  // LDS store instructions per group (b128 / write2_b32 / b32)
    constexpr int N_VM = C::A_Q + XN;
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < GS; ++p) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);           // the two operand reads of a later k-step
            __builtin_amdgcn_sched_group_barrier(0x200, (N_DSW + GS - 1) / GS, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, (N_VM + GS - 1) / GS, 0);
        }
    };
    float* const buf0 = lat_smem;
    float* const buf1 = lat_smem + C::BUF_FLOATS;
    Stage S0, S1;
    Ops R0, R1;
    gload(S0); lstore(S0, buf0);                                         // group 0
    gload(S1);                                                           // group 1
    __builtin_amdgcn_wave_barrier();
    lread(R0, buf0);
    lstore(S1, buf1);
    gload(S0);                                                           // group 2
    for (int g = 0; g < ng; g += 2) {
        __builtin_amdgcn_wave_barrier();
        gload(S1);                                                       // group g + 3
        lstore(S0, buf0);                                                // group g + 2
        lread(R1, buf1);                                                 // group g + 1
        mfmas(R0);                                                       // group g
        interleave();
        if (g + 1 < ng) {
            __builtin_amdgcn_wave_barrier();
            gload(S0);                                                   // group g + 4
            lstore(S1, buf1);                                            // group g + 3
            lread(R0, buf0);                                             // group g + 2
            mfmas(R1);                                                   // group g + 1
            interleave();
        }
    }

    // ---- epilogue: the operations of conv1d_mfma_body's, in its order, one element at a time
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * q + i;
        if (m >= a.Mrows || n >= a.Ncols) continue;
        const int ch = m, t = n;
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
__global__ __launch_bounds__(64) void conv1d_lat_kernel(const ConvArgs a)
{
    if (a.vec4) conv_lat_body<KS, STRIDE, DIL, CG, true>(a);
    else conv_lat_body<KS, STRIDE, DIL, CG, false>(a);
}

template <int KS, int STRIDE, int DIL, int CG>
static hipError_t launch_lat(const ConvArgs& a_in, hipStream_t s)
{
    using C = LatCfg<KS, STRIDE, DIL, CG>;
    ConvArgs a = a_in;
    if (a.Cin % CG != 0 || a.Mpad % 16 != 0) return hipErrorInvalidValue;
    if (a.name_out) {
        snprintf(a.name_out, a.name_len, "conv1d_lat_kernel<%d, %d, %d, %d>", KS, STRIDE, DIL, CG);
        return hipSuccess;
    }
    a.n_tiles = (a.Ncols + 15) / 16;
    a.vec4 = (a.Tin % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    dim3 grid((unsigned)(a.n_tiles * a.B), (unsigned)((a.Mrows + 15) / 16));
    int pi = -1;
    if (prof_enabled()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv1d_lat_kernel<%d, %d, %d, %d>", KS, STRIDE, DIL, CG);
        const int lim = a.tvalid > 0 ? a.tvalid : a.Ncols;        // algorithmic columns, as launch_conv1d_mfma counts them
        pi = prof_begin(nm, 2.0 * a.Cin * KS * a.Mrows * (double)lim * a.B, s);
    }
    hipLaunchKernelGGL((conv1d_lat_kernel<KS, STRIDE, DIL, CG>), grid, dim3(64), (size_t)2 * C::BUF_FLOATS * sizeof(float), s, a);
    prof_end(pi, s);
    return hipGetLastError();
}

// When the latency form is chosen (measured layer by layer against the LDS-tiled kernels at 1 and 6 segments, device time:
// gpurun_out/r05lat4): the launch has few enough 16 x 16 tiles to sit about two per SIMD -- beyond that the LDS-tiled kernels,
// whose operands are shared by a whole block, win -- and the K chain is long enough to pay for a wave's fixed cost (its index
// setup and the element-wise epilogue): a short chain on many tiles (the 1x1 conv of a T = 600 unit at one segment: 1 824 tiles,
// 192 k-steps) stays on the 64 x 64 tiles.  A very long chain (dec.in at six segments: 2 784 tiles x 1 792 k-steps) still wins at
// up to twice the tile count.  MVQ_LAT_MAX_TILES overrides the tile threshold (A/B runs; 0 switches the form off).
static long lat_max_tiles()
{
    static const long v = [] {
        const char* e = getenv("MVQ_LAT_MAX_TILES");
        if (e) note_env_override(MVQ_BF_ENV_LAT_TILES);
        return e ? atol(e) : 2048L;
    }();
    return v;
}

bool conv_lat_wanted(const ConvArgs& a, int ks)
{
    if (a.alpha_in || a.tper || a.vp_seg || a.up_per_out || a.n_base || a.n_tiles_max) return false;
    if (a.B <= 0 || a.Ncols <= 0 || a.up_s > 1) return false;            // transposed convs: the polyphase scatter store loses (80 -> 95 us)
    const long tiles = (long)a.B * ((a.Ncols + 15) / 16) * ((a.Mrows + 15) / 16);
    const long steps = (long)a.Cin * ks / 4;
    const long cap = lat_max_tiles();
    if (tiles <= cap / 2) return true;
    if (tiles <= cap) return steps >= 512;
    return tiles <= 2 * cap && steps >= 1536;
}

hipError_t launch_conv_lat(const ConvArgs& a, int ks, int stride, int dil, hipStream_t s)
{
    // channels per group: a group of 16-32 k-steps is the prefetch distance, and the wave's two buffers stay below ~20 KB of LDS
    // (eight one-wave workgroups per CU); every width of the model divides
    if (a.up_s > 1) return hipErrorInvalidValue;                        // transposed convs keep the LDS-tiled kernels (conv_lat_wanted)
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
        case 2: return launch_lat<4, 2, 1, 16>(a, s);
        case 4: return launch_lat<8, 4, 1, 8>(a, s);
        case 5: return launch_lat<10, 5, 1, 8>(a, s);
        case 8: return launch_lat<16, 8, 1, 8>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mvq
