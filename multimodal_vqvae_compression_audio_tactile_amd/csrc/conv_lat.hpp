// conv_lat.hpp -- body of the latency-form conv (one wave per 16 x 16 tile on v_mfma_f32_16x16x4_f32; see conv_lat.hip), as a
// device function over an explicit tile index and LDS pointer so that conv_lat.hip's kernel and the fused AR-loop kernel
// (ar_fused.hip) run the SAME code.
#pragma once
#include "conv1d_mfma.hpp"

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

template <int KS, int STRIDE, int DIL, int CG, bool VEC, bool OPAQUE_TID = false>
__device__ __forceinline__ void conv_lat_body(const ConvArgs& a, int bx, int by, float* lat_smem)
{
    // (bx, by) = (item * n_tiles + column tile, row tile); lat_smem = this WAVE's 2 * BUF_FLOATS floats of LDS.  The stand-alone
    // kernel passes its blockIdx; the fused AR-loop kernel (ar_fused.hip) walks its tiles through the same function.
    using C = LatCfg<KS, STRIDE, DIL, CG>;
    constexpr int GS = C::GS;
    int lane_ = threadIdx.x & 63;
    if (OPAQUE_TID) asm volatile("" : "+v"(lane_));                   // fused callers: keep this body's index math INSIDE their stage loop (see ar_fused.hip)
    const int lane = lane_;
    const int r = lane & 15, q = lane >> 4;
    const int b = bx / a.n_tiles, tn = bx - b * a.n_tiles;
    const int n0 = tn * 16, m0 = by * 16;
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
    // Four-stage pipeline; in iteration g, all in ONE basic block:
    //     S[g & 1] (group g + 2)      -> LDS buffer g & 1
    //     global loads of group g + 4 -> S[g & 1]                  (into the registers just stored: in flight for TWO iterations)
    //     LDS buffer (g + 1) & 1      -> R[(g + 1) & 1]            (group g + 1, stored one iteration ago)
    //     MFMAs of group g from R[g & 1]
    // A dependent 16x16x4 MFMA issues every 40 cycles; the ~2.5 other instructions per k-step go into those gaps (the scheduling
    // groups below), so the chain runs at its own latency -- provided the operands have arrived: a group of 16 k-steps is ~0.4 us of
    // MFMAs, a weight slice that no L2 holds yet comes from the Infinity Cache / HBM in 0.5-0.8 us.  (The first cut fetched group
    // g + 3 one iteration before its LDS store: the 1x1 GEMMs of an AR chunk then ran at the memory latency, 13 us for K = 1024
    // against 6.4 us of MFMAs; gpurun_out/f4.)  Stages that reach past the last group move data nobody reads.
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
    // epilogue operands: requested now, they arrive while the K loop runs
    float e_bias[4], e_res[4], e_dsn[4], e_dal[4], e_a2[4], e_al[4];
    bool e_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * q + i;
        e_ok[i] = m < a.Mrows && n < a.Ncols;
        const int ch = e_ok[i] ? m : 0;
        const size_t off = e_ok[i] ? ((size_t)b * a.Cout + ch) * a.Tout + n : 0;
        e_bias[i] = a.bias ? a.bias[ch] : 0.0f;
        e_dsn[i] = a.dsn_src ? a.dsn_src[off] : 0.0f;
        e_dal[i] = a.dsn_src ? a.dsn_alpha[ch] : 0.0f;
        e_res[i] = a.residual ? a.residual[off] : 0.0f;
        e_a2[i] = a.y2 ? a.alpha2[ch] : 0.0f;
        e_al[i] = a.alpha_out ? a.alpha_out[ch] : 0.0f;
    }
    float* const buf0 = lat_smem;
    float* const buf1 = lat_smem + C::BUF_FLOATS;
    Stage S0, S1;
    Ops R0, R1;
    gload(S0);                                                           // group 0
    gload(S1);                                                           // group 1
    lstore(S0, buf0);
    gload(S0);                                                           // group 2
    __builtin_amdgcn_wave_barrier();
    lread(R0, buf0);
    lstore(S1, buf1);
    gload(S1);                                                           // group 3
    for (int g = 0; g < ng; g += 2) {
        __builtin_amdgcn_wave_barrier();
        lstore(S0, buf0);                                                // group g + 2
        gload(S0);                                                       // group g + 4
        lread(R1, buf1);                                                 // group g + 1
        mfmas(R0);                                                       // group g
        interleave();
        if (g + 1 < ng) {
            __builtin_amdgcn_wave_barrier();
            lstore(S1, buf1);                                            // group g + 3
            gload(S1);                                                   // group g + 5
            lread(R0, buf0);                                             // group g + 2
            mfmas(R1);                                                   // group g + 1
            interleave();
        }
    }

    // ---- epilogue: the operations of conv1d_mfma_body's, in its order, one element at a time.  Every operand of the lane's four
    // elements was requested BEFORE the K loop (e_* above): read here, interleaved with the stores, element i + 1's bias / residual
    // loads would wait behind element i's store (they may alias as far as the compiler knows) -- four dependent round trips to far
    // memory after the last MFMA instead of none.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!e_ok[i]) continue;
        const int ch = m0 + 4 * q + i, t = n;
        const size_t off = ((size_t)b * a.Cout + ch) * a.Tout + t;
        float v = acc[i] + e_bias[i];
        if (a.dsn_src) { const float ad = e_dal[i]; v = v * det_dsnake(e_dsn[i], ad, 1.0f / (ad + 1e-9f)); }
        if (a.residual) v = v + e_res[i];
        const bool tail = a.tvalid && t >= a.tvalid;
        if (a.y2) { const float a2 = e_a2[i]; a.y2[off] = tail ? 0.0f : det_snake(v, a2, 1.0f / (a2 + 1e-9f)); }
        if (a.alpha_out) { const float al = e_al[i]; v = det_snake(v, al, 1.0f / (al + 1e-9f)); }
        if (a.act == 1) v = det_tanh(v);
        if (a.act == 2) v = det_gelu(v);
        a.y[off] = tail ? 0.0f : v;
    }
}

}  // namespace mvq
