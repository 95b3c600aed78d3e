// conv1d_mfma.hpp -- fp32 implicit-GEMM 1-D convolution on the gfx950 matrix cores.
//
//   Out[co, t] = sum_{ci, kk} W[co, ci, kk] * snake_in(X)[ci, t*STRIDE + kk*DIL - pad]
//
// GEMM view: M = output channels (A operand = packed weights), N = output time steps (B operand =
// activations read straight out of an LDS-staged input tile -- the im2col matrix is never materialised:
// a tap is just an LDS address offset of kk*DIL), K = (ci, kk) in that order.  v_mfma_f32_32x32x2_f32 is
// an exact k-ordered fp32 fma chain, so walking K as "ci ascending, tap ascending" reproduces the
// arithmetic contract of include/mvq.h bit for bit.
//
// Block = 256 threads = 4 waves; wave tile = (32*MT) x (32*NT); block tile BM x BN.  K is walked in chunks
// of CK input channels (CK*KS a multiple of 2); weights and the input tile (with its (KS-1)*DIL halo)
// are double-buffered in LDS, register-staged: global loads for chunk c+1 are issued before the MFMAs of
// chunk c and written to the other LDS buffer after them (one barrier per chunk).
// Snake1d in front of the conv is applied while staging; bias / residual / Snake1d behind the conv /
// tanh are applied to the accumulators.  SHUFFLE=true is the polyphase form of ConvTranspose1d
// (kernel 2*S, stride S): a 2-tap conv over M = Cout*S rows whose row (co*S + r) is written to
// y[co, q*S + r - P].
#pragma once
#include <hip/hip_runtime.h>
#include "det_math.hpp"

namespace mvq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;         // [B, Cin, Tin]
    const float* wp;        // packed [(ci*KS + kk) * Mpad + m]
    const float* bias;      // [Cout] or null
    const float* alpha_in;  // [Cin] or null
    const float* residual;  // [B, Cout, Tout] or null
    const float* alpha_out; // [Cout] or null
    float* y;               // [B, Cout, Tout]
    int B, Cin, Tin, Cout, Tout;
    int pad;                // left zero padding in input samples
    int Mpad;               // padded M (row pitch of wp)
    int Mrows;              // valid GEMM rows (Cout, or Cout*S for SHUFFLE)
    int Ncols;              // GEMM columns per batch element (Tout, or Tin+1 for SHUFFLE)
    int n_tiles;            // ceil(Ncols / BN)
    int act;
    int up_s, up_p;         // SHUFFLE: stride S and torch padding P
};

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, bool SHUFFLE>
struct ConvCfg {
    static constexpr int BM = 32 * MT * WAVES_M;
    static constexpr int BN = 32 * NT * WAVES_N;
    static constexpr int KC = CK * KS;                                  // K elements per chunk
    static constexpr int XT = (BN - 1) * STRIDE + (KS - 1) * DIL + 1;    // input samples per row
    static constexpr int XTP = XT + ((XT % 2) ? 0 : 1);                  // odd pitch
    static constexpr int W_FLOATS = KC * BM;
    static constexpr int X_FLOATS = CK * XTP;
    static constexpr int LDS_BYTES = 2 * (W_FLOATS + X_FLOATS) * 4;
    static constexpr int W_VEC = W_FLOATS / 4;                           // float4 per chunk
    static constexpr int W_PER_THREAD = (W_VEC + 255) / 256;
    static constexpr int X_PER_THREAD = (CK * XT + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "block is 4 waves");
    static_assert(KC % 2 == 0, "chunk K must be even (32x32x2 MFMA)");
    static_assert(BM % 4 == 0, "float4 weight staging");
};

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, bool SHUFFLE>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(const ConvArgs a)
{
    using C = ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, SHUFFLE>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ws = smem;                          // [2][KC][BM]
    float* const Xs = smem + 2 * C::W_FLOATS;        // [2][CK][XTP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int b = blockIdx.x / a.n_tiles;
    const int tile_n = blockIdx.x - b * a.n_tiles;
    const int n0 = tile_n * C::BN;
    const int m0 = blockIdx.y * C::BM;
    const int t_in0 = n0 * STRIDE - a.pad;           // input sample of LDS column 0

    const int n_chunks = (a.Cin + CK - 1) / CK;
    const float* const xb = a.x + (size_t)b * a.Cin * a.Tin;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 wreg[C::W_PER_THREAD];
    float xreg[C::X_PER_THREAD];

    auto load_chunk = [&](int chunk) {
        const int ci0 = chunk * CK;
        const float* wsrc = a.wp + (size_t)ci0 * KS * a.Mpad + m0;
#pragma unroll
        for (int u = 0; u < C::W_PER_THREAD; ++u) {
            const int v = tid + u * 256;
            if (C::W_VEC % 256 == 0 || v < C::W_VEC) {
                const int row = v / (C::BM / 4);
                const int c4 = v - row * (C::BM / 4);
                wreg[u] = *reinterpret_cast<const float4*>(wsrc + (size_t)row * a.Mpad + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < C::X_PER_THREAD; ++u) {
            const int e = tid + u * 256;
            float val = 0.0f;
            if ((CK * C::XT) % 256 == 0 || e < CK * C::XT) {
                const int cl = e / C::XT;
                const int xi = e - cl * C::XT;
                const int ci = ci0 + cl;
                const int g = t_in0 + xi;
                if (ci < a.Cin && g >= 0 && g < a.Tin) {
                    val = xb[(size_t)ci * a.Tin + g];
                    if (a.alpha_in) {
                        const float al = a.alpha_in[ci];
                        val = det_snake(val, al, 1.0f / (al + 1e-9f));
                    }
                }
            }
            xreg[u] = val;
        }
    };

    auto store_chunk = [&](int buf) {
        float* wdst = Ws + buf * C::W_FLOATS;
        float* xdst = Xs + buf * C::X_FLOATS;
#pragma unroll
        for (int u = 0; u < C::W_PER_THREAD; ++u) {
            const int v = tid + u * 256;
            if (C::W_VEC % 256 == 0 || v < C::W_VEC) *reinterpret_cast<float4*>(wdst + v * 4) = wreg[u];
        }
#pragma unroll
        for (int u = 0; u < C::X_PER_THREAD; ++u) {
            const int e = tid + u * 256;
            if ((CK * C::XT) % 256 == 0 || e < CK * C::XT) {
                const int cl = e / C::XT;
                const int xi = e - cl * C::XT;
                xdst[cl * C::XTP + xi] = xreg[u];
            }
        }
    };

    // per-lane LDS bases
    const int a_base = h * C::BM + wm * (MT * 32) + l31;
    const int b_base = (wn * (NT * 32) + l31) * STRIDE;

    auto compute_chunk = [&](int buf) {
        const float* wsrc = Ws + buf * C::W_FLOATS + a_base;
        const float* xsrc = Xs + buf * C::X_FLOATS + b_base;
#pragma unroll
        for (int s = 0; s < C::KC / 2; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int off0 = (k0 / KS) * C::XTP + (k0 % KS) * DIL;
            const int off1 = (k1 / KS) * C::XTP + (k1 % KS) * DIL;
            const int xoff = h ? off1 : off0;
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = wsrc[2 * s * C::BM + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = xsrc[xoff + j * 32 * STRIDE];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) load_chunk(c + 1);
        compute_chunk(buf);
        if (c + 1 < n_chunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + (wm * MT + i) * 32 + row;
            if (m >= a.Mrows) continue;
            int co, rr = 0;
            if (SHUFFLE) { co = m / a.up_s; rr = m - co * a.up_s; } else { co = m; }
            const float bv = a.bias ? a.bias[co] : 0.0f;
            float al = 0.0f, inv = 0.0f;
            if (a.alpha_out) { al = a.alpha_out[co]; inv = 1.0f / (al + 1e-9f); }
            const size_t rowoff = ((size_t)b * a.Cout + co) * a.Tout;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + (wn * NT + j) * 32 + l31;
                if (n >= a.Ncols) continue;
                int t;
                if (SHUFFLE) { t = n * a.up_s + rr - a.up_p; if (t < 0 || t >= a.Tout) continue; } else { t = n; }
                float v = acc[i][j][r] + bv;
                if (!SHUFFLE && a.residual) v = v + a.residual[rowoff + t];
                if (a.alpha_out) v = det_snake(v, al, inv);
                if (a.act == 1) v = det_tanh(v);
                a.y[rowoff + t] = v;
            }
        }
    }
}

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, bool SHUFFLE>
inline hipError_t launch_conv1d_mfma(const ConvArgs& a_in, hipStream_t stream)
{
    using C = ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, SHUFFLE>;
    ConvArgs a = a_in;
    a.n_tiles = (a.Ncols + C::BN - 1) / C::BN;
    auto kern = conv1d_mfma_kernel<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, SHUFFLE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((unsigned)(a.n_tiles * a.B), (unsigned)((a.Mrows + C::BM - 1) / C::BM));
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// tile-shape selection shared by the packers and the launchers: padded M for a given number of rows
inline int conv_tile_bm(int mrows)
{
    if (mrows % 128 == 0) return 128;
    if (mrows % 96 == 0) return 96;
    if (mrows % 64 == 0) return 64;
    return mrows > 96 ? 128 : (mrows > 64 ? 96 : 64);
}
inline int conv_mpad(int mrows) { const int bm = conv_tile_bm(mrows); return (mrows + bm - 1) / bm * bm; }

}  // namespace mvq
