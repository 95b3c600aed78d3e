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
#include "conv_lat.hpp"

namespace mvq {

template <int KS, int STRIDE, int DIL, int CG>
__global__ __launch_bounds__(64) void conv1d_lat_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    if (a.vec4) conv_lat_body<KS, STRIDE, DIL, CG, true>(a, blockIdx.x, blockIdx.y, lat_smem);
    else conv_lat_body<KS, STRIDE, DIL, CG, false>(a, blockIdx.x, blockIdx.y, lat_smem);
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
