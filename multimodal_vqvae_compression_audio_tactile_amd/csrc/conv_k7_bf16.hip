// conv_k7_bf16.hip -- OPT-IN, NON-PARITY arithmetic mode "bf16x6" for the wide ResidualUnits' 7-tap dilated convs.
//
// The default path computes every conv as an exact k-ordered fp32 fma chain on v_mfma_f32_32x32x2_f32 (include/mvq.h,
// "Arithmetic contract").  This unit is the ONE deliberate departure, selected only through mvq_conv1d_k7_bf16x6_f32:
// every fp32 operand is split into three bf16 pieces a = a0 + a1 + a2 (a0 = rne(a), a1 = rne(a - a0), a2 = rne(a - a0 - a1):
// 24 significant bits, the two subtractions are exact), and a product a*b is evaluated as the six piece products
// a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms (a1b2,
// a2b1, a2b2) are below 2^-24 |ab|: the truncation error of a 1 792-term sum is 35x below the rounding error of the fp32 chain
// itself (tools/bf16x6_accuracy.py), but the summation ORDER differs from the contract, so results are fp32-accurate and NOT
// bit-identical to the oracle.  Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2 at 1/16 the time per flop: 2.67x
// the matrix rate for the same contraction.
//
// Data formats (both made on the device by the kernels below):
//   split activations  xs: bf16 [B][C/8][3 pieces][T][8 channels]      (16 bytes = one (item, channel octet, piece, t))
//   packed weights     wq: bf16 [Cout/BM][Cin/16][7 taps][3 pieces][2 octets][BM rows][8 channels], BM = 128 (or 96 where 128 does
//                      not divide Cout: C = 192) = the row tile of the kernel; one (block, tap) slice is 96 BM bytes
// GEMM view: M = output channels (A operand = weights), N = time, K walked as (16-channel block, tap); the MFMA's lane map
// (lane l: row/column l & 31, k = 8 (l >> 5) + j) makes one operand fragment of one piece the 16 bytes of one channel octet at
// one row / time step: a single conflict-free ds_read_b128 (consecutive lanes 16 bytes apart).  Both operands reach LDS by
// global -> LDS DMA only (no register staging, no VALU in the K loop): the activation tile of a channel block (all seven
// taps read it at offsets of tap * DIL) is double-buffered, the weight slices run through a ring of three.
#include "conv_dispatch.hpp"

namespace mvq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16b __attribute__((ext_vector_type(16)));

// ---- operand preparation ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float a, __bf16& p0, __bf16& p1, __bf16& p2)
{
    p0 = (__bf16)a;
    const float r1 = a - (float)p0;          // exact
    p1 = (__bf16)r1;
    const float r2 = r1 - (float)p1;         // exact
    p2 = (__bf16)r2;
}

// x[B][C][T] fp32 -> xs[B][C/8][3][T][8] bf16.  One thread per (item, octet, t): eight strided loads (each coalesced across
// the threads of a wave, which walk t), three 16-byte stores.
__global__ void bf16x3_split_kernel(const float* __restrict__ x, bf16x8* __restrict__ xs, int C, int T, size_t total)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int t = (int)(gid % (size_t)T);
    const size_t bo = gid / (size_t)T;               // item * (C/8) + octet
    const float* src = x + bo * 8 * (size_t)T + t;
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, c;
        split3(src[(size_t)j * T], a, b, c);
        q0[j] = a; q1[j] = b; q2[j] = c;
    }
    bf16x8* dst = xs + bo * 3 * (size_t)T + t;
    dst[0] = q0; dst[(size_t)T] = q1; dst[2 * (size_t)T] = q2;
}

// w[Cout][Cin][7] fp32 -> wq (layout above).  One thread per (row tile, channel block, tap, octet, row): it splits its eight weights
// once and stores the three 16-byte fragments (one per piece).  (A first form -- one thread per fragment, picking its piece out of a
// local array by a run-time index -- produced sporadically wrong fragments on the device; no run-time register indexing here.)
// flip != 0: the INPUT-GRADIENT image of a forward weight w[Cin][Cout][7] (forward Cout = this Cin): row m = forward input channel,
// k-channel = forward output channel, taps reversed -- W'[m][c][k] = w[c][m][6 - k] (stride-1 'same' conv: same dilation and padding)
__global__ void bf16x3_pack_k7_kernel(const float* __restrict__ w, bf16x8* __restrict__ wq, int Cout, int Cin, int BM, int flip, size_t total)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    // thread index -> (mt, cb, tap, h, m)
    size_t r = gid;
    const int m = (int)(r % (size_t)BM); r /= (size_t)BM;
    const int h = (int)(r % 2); r /= 2;
    const int tap = (int)(r % 7); r /= 7;
    const int ncb = Cin / 16;
    const int cb = (int)(r % (size_t)ncb);
    const int mt = (int)(r / (size_t)ncb);
    const int co = mt * BM + m;
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = cb * 16 + h * 8 + j;
        __bf16 a, b, c;
        split3(flip ? w[((size_t)ci * Cout + co) * 7 + (6 - tap)] : w[((size_t)co * Cin + ci) * 7 + tap], a, b, c);
        q0[j] = a; q1[j] = b; q2[j] = c;
    }
    // fragment (mt, cb, tap, piece, h, m)
    const size_t slice = ((size_t)mt * ncb + cb) * 7 + tap;
    bf16x8* dst = wq + slice * (size_t)(6 * BM) + (size_t)h * BM + m;
    dst[0] = q0; dst[(size_t)2 * BM] = q1; dst[(size_t)4 * BM] = q2;
}

// ---- f16x3 form: two fp16 pieces under a power-of-two scale ------------------------------------------------------------------------
// fp16 keeps 11 significant bits but only 5 exponent bits, so a tensor is first scaled by the power of two that puts its largest
// magnitude into [2^13, 2^14) (exact; well inside the fp16 range, 2^16): a = (h0 + h1) / S with h0 = rne16(a S), h1 = rne16(a S - h0),
// |a S - h0 - h1| <= max(2^-22 |a S|, 2^-25) -- 22 significant bits for every element within 2^17 of the maximum, an absolute error
// of 2^-39 of the maximum below that.  One scale per ITEM for the activations (a per-position scale would not factor out of a
// 7-tap sum), one per tensor for the weights; the conv's epilogue multiplies by the two inverse scales (exact).
__device__ __host__ __forceinline__ int f16_scale_exp(unsigned amax_bits)
{
    const int e = (int)((amax_bits >> 23) & 0xff);                 // biased exponent of the maximum (sign bit is clear)
    if (e == 0 || e == 0xff) return 0;                              // zero / denormal / non-finite maximum: no scaling
    const int k = 13 - (e - 127);                                   // S = 2^k
    return k < -126 ? -126 : (k > 126 ? 126 : k);                   // clamped ONCE, so that 2^k and 2^-k are both normal floats and the
}                                                                   // epilogue undoes exactly the scale the split applied (a maximum below
__device__ __forceinline__ float f16_pow2(int k)                    // 2^-113 keeps fewer significant bits, never a factor of two)
{
    return __uint_as_float((unsigned)(k + 127) << 23);              // k in [-126, 126] (f16_scale_exp)
}
__device__ __forceinline__ float f16_scale(unsigned amax_bits) { return f16_pow2(f16_scale_exp(amax_bits)); }
__device__ __forceinline__ float f16_inv_scale(unsigned amax_bits) { return f16_pow2(-f16_scale_exp(amax_bits)); }

// amax[item] = max |x| over the FINITE elements of the item's `per_item` elements, as float bits (non-negative floats order like
// unsigned integers); the caller zeroes amax first.  grid (blocks per item, items).  A NaN / Inf sample therefore neither decides nor
// disables the item's scale: it becomes a non-finite piece and contaminates its own receptive field only, as in the exact path.
__global__ void f16_amax_kernel(const float* __restrict__ x, unsigned* __restrict__ amax, size_t per_item)
{
    const float* src = x + (size_t)blockIdx.y * per_item;
    unsigned m = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((per_item & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {       // 16-byte loads, four in flight per thread
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        const size_t n4 = per_item >> 2;
        size_t i = i0;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const uint4 a = s4[i], b = s4[i + stride], c = s4[i + 2 * stride], d = s4[i + 3 * stride];
            unsigned q;
            q = a.x & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.y & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.z & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.w & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m;
            q = b.x & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = b.y & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = b.z & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = b.w & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m;
            q = c.x & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = c.y & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = c.z & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = c.w & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m;
            q = d.x & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = d.y & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = d.z & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = d.w & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m;
        }
        for (; i < n4; i += stride) {
            const uint4 a = s4[i];
            unsigned q;
            q = a.x & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.y & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.z & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m; q = a.w & 0x7fffffffu; q = q < 0x7f800000u ? q : 0u; m = q > m ? q : m;
        }
    } else {
        for (size_t i = i0; i < per_item; i += stride) {
            unsigned u = __float_as_uint(src[i]) & 0x7fffffffu;
            u = u < 0x7f800000u ? u : 0u;
            m = u > m ? u : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned v = __shfl_xor(m, o); m = v > m ? v : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + blockIdx.y, m);
}

__device__ __forceinline__ void split2h(float a, float S, _Float16& p0, _Float16& p1)
{
    const float as = a * S;                   // exact (power of two) unless it leaves the fp32 range
    p0 = (_Float16)as;
    p1 = (_Float16)(as - (float)p0);          // the subtraction is exact
}

// x[B][C][T] fp32 -> xs[B][C/8][2][T][8] fp16 under the item's scale
__global__ void f16x2_split_kernel(const float* __restrict__ x, f16x8* __restrict__ xs, const unsigned* __restrict__ amax, int C, int T, size_t total)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int t = (int)(gid % (size_t)T);
    const size_t bo = gid / (size_t)T;               // item * (C/8) + octet
    const float S = f16_scale(amax[bo / (size_t)(C / 8)]);
    const float* src = x + bo * 8 * (size_t)T + t;
    f16x8 q0, q1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 a, b;
        split2h(src[(size_t)j * T], S, a, b);
        q0[j] = a; q1[j] = b;
    }
    f16x8* dst = xs + bo * 2 * (size_t)T + t;
    dst[0] = q0; dst[(size_t)T] = q1;
}

// w[Cout][Cin][7] fp32 -> wq [Cout/BM][Cin/16][7][2 pieces][2 octets][BM][8] fp16 under the tensor's scale
__global__ void f16x2_pack_k7_kernel(const float* __restrict__ w, f16x8* __restrict__ wq, const unsigned* __restrict__ amax, int Cout, int Cin,
                                     int BM, int flip, size_t total)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    size_t r = gid;
    const int m = (int)(r % (size_t)BM); r /= (size_t)BM;
    const int h = (int)(r % 2); r /= 2;
    const int tap = (int)(r % 7); r /= 7;
    const int ncb = Cin / 16;
    const int cb = (int)(r % (size_t)ncb);
    const int mt = (int)(r / (size_t)ncb);
    const int co = mt * BM + m;
    const float S = f16_scale(amax[0]);
    f16x8 q0, q1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = cb * 16 + h * 8 + j;
        _Float16 a, b;
        split2h(flip ? w[((size_t)ci * Cout + co) * 7 + (6 - tap)] : w[((size_t)co * Cin + ci) * 7 + tap], S, a, b);
        q0[j] = a; q1[j] = b;
    }
    const size_t slice = ((size_t)mt * ncb + cb) * 7 + tap;
    f16x8* dst = wq + slice * (size_t)(4 * BM) + (size_t)h * BM + m;
    dst[0] = q0; dst[(size_t)2 * BM] = q1;
}

// ---- the conv -----------------------------------------------------------------------------------------------------------------
struct K7BfArgs {
    const bf16x8* xs;       // split activations (already carry the input Snake)
    const bf16x8* wq;       // packed split weights
    const float* bias;      // [Cout] or null
    const float* alpha_out; // [Cout] or null: Snake1d behind the conv
    float* y;               // [B][Cout][T] fp32
    int B, Cin, Cout, T;
    int n_tiles;            // column tiles per item
    int tvalid;             // > 0: columns >= tvalid are written as zeros (zero-padded rows, include/mvq.h)
    // f16x3 form only: |x| maximum of every item / of the weight tensor as float bit patterns (the power-of-two scales the split
    // applied follow from them: f16_scale_exp); null in the bf16x6 form
    const unsigned* xamax;
    const unsigned* wamax;
    // training config (Decoder.forward_saving / backward_input in a mode):
    float* y2;              // dual output: y = the raw value (conv + bias), y2 = snake(y, alpha_out) -- the saved pre-activation and
                            // what the 1x1 conv consumes.  null: single output (Snake applied in place when alpha_out is given)
    const float* dsn_src;   // input-gradient epilogue: v = acc * d snake(dsn_src)/dx (+ residual); [B][Cout][T], alpha dsn_alpha[Cout]
    const float* dsn_alpha;
    const float* residual;  // [B][Cout][T] added last (the gradient arriving through the skip path)
};

template <int DIL, int MT, int NT, int WM, int WN, int NST, int NP>
struct K7BfCfg {
    static constexpr int NW = WM * WN, NTHR = 64 * NW;
    static constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    static constexpr int XT = BN + 6 * DIL;                 // activation positions a tile needs
    static constexpr int XPIECES = 2 * NP * XT;             // 16-byte pieces of one channel block's tile: [piece][octet][position]
    static constexpr int XBYTES = XPIECES * 16;
    static constexpr int WPIECES = NP * 2 * BM;             // one (channel block, tap) weight slice
    static constexpr int WBYTES = WPIECES * 16;
    static constexpr int NUX = (XPIECES + NTHR - 1) / NTHR; // DMA instructions per wave for an activation tile (the last may be partial)
    static constexpr int NUW = (WPIECES + NTHR - 1) / NTHR;
    // weight ring of NST stages: slice s + NST - 1 is issued at step s, so NST - 2 slices are in flight behind the one being
    // multiplied.  The steps are short (24 MFMAs of 32 cycles per wave), so the ring has to be deep: bytes in flight per CU =
    // inflow rate x L2 latency (Little), ~20 B/clk x ~3 000 clk with 128-column tiles.
    static constexpr int LDS_BYTES = 2 * XBYTES + NST * WBYTES;
    static_assert(WPIECES % 64 == 0, "weight slice = whole wave instructions");
    static_assert(NST >= 3 && NST <= 6, "the next activation tile is issued at tap 1 and must be older than slice 7 (cb + 1)");
    static_assert((NST - 2) * NUW + NUX <= 31, "vmcnt immediates of vm_wait_dyn");
};

template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 31] (the count is an immediate of the instruction)
__device__ __forceinline__ void vm_wait_dyn(int n)
{
    switch (n) {
#define MVQ_VW(k) case k: vm_wait<k>(); break;
        MVQ_VW(0) MVQ_VW(1) MVQ_VW(2) MVQ_VW(3) MVQ_VW(4) MVQ_VW(5) MVQ_VW(6) MVQ_VW(7) MVQ_VW(8) MVQ_VW(9) MVQ_VW(10) MVQ_VW(11)
        MVQ_VW(12) MVQ_VW(13) MVQ_VW(14) MVQ_VW(15) MVQ_VW(16) MVQ_VW(17) MVQ_VW(18) MVQ_VW(19) MVQ_VW(20) MVQ_VW(21) MVQ_VW(22)
        MVQ_VW(23) MVQ_VW(24) MVQ_VW(25) MVQ_VW(26) MVQ_VW(27) MVQ_VW(28) MVQ_VW(29) MVQ_VW(30)
#undef MVQ_VW
        default: vm_wait<31>(); break;
    }
}

template <int DIL, int MT, int NT, int WM, int WN, int NST, int NP>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 64 * WM * WN), amdgpu_waves_per_eu(2)))
void conv_k7_pieces_kernel(const K7BfArgs a)
{
    using C = K7BfCfg<DIL, MT, NT, WM, WN, NST, NP>;
    static_assert(NP == 3 || NP == 2, "three bf16 pieces (six products) or two fp16 pieces (three products)");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave_u / WN, wn = wave_u % WN;   // WM x WN waves
    const int l31 = lane & 31, h = lane >> 5;

    const int bx = blockIdx.x;
    const int b = bx / a.n_tiles, tile_n = bx - b * a.n_tiles;
    const int n0 = tile_n * C::BN, mt = blockIdx.y, m0 = mt * C::BM;
    const int t0 = n0 - 3 * DIL;                                   // input position of tile position 0
    const int ncb = a.Cin / 16;
    const int n_steps = ncb * 7;
    const size_t T = (size_t)a.T;

    // ---- per-lane DMA sources
    // activation pieces: q = u * NTHR + tid -> (ph = piece * 2 + octet-of-the-block, tt); advance by two octets per channel block
    const bf16x8* xsrc[C::NUX];
    long long xstep[C::NUX];
    bool xlive[C::NUX];
#pragma unroll
    for (int u = 0; u < C::NUX; ++u) {
        const int q = u * C::NTHR + tid;
        xlive[u] = q < C::XPIECES;
        const int qq = xlive[u] ? q : C::XPIECES - 1;
        const int ph = qq / C::XT, tt = qq - ph * C::XT;
        const int p = ph >> 1, o = ph & 1;                     // piece, octet of the channel block
        const int t = t0 + tt;
        const bool ok = t >= 0 && t < a.T;
        xsrc[u] = ok ? a.xs + (((size_t)b * (a.Cin / 8) + o) * NP + p) * T + t : reinterpret_cast<const bf16x8*>(g_zero16);
        xstep[u] = ok ? (long long)(2 * NP) * (long long)T : 0;
    }
    // weight pieces: slice s of this row tile starts at wq + (mt * n_steps + s) * WPIECES
    const bf16x8* wsrc[C::NUW];
    bool wlive[C::NUW];
#pragma unroll
    for (int u = 0; u < C::NUW; ++u) {
        const int q = u * C::NTHR + tid;
        wlive[u] = q < C::WPIECES;
        wsrc[u] = a.wq + (size_t)mt * n_steps * C::WPIECES + (wlive[u] ? q : 0);
    }
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned wave_off = (unsigned)(wave_u * 64 * 16);
    // how many DMA instructions THIS wave issues for a tile / a slice (the last one only by the waves that hold live pieces)
    const int nx_issue = ((C::NUX - 1) * C::NTHR + wave_u * 64 < C::XPIECES) ? C::NUX : C::NUX - 1;
    const int nw_issue = ((C::NUW - 1) * C::NTHR + wave_u * 64 < C::WPIECES) ? C::NUW : C::NUW - 1;

    auto dma_x = [&](int buf) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (unsigned)(buf * C::XBYTES) + wave_off;
        constexpr int NFULL = C::XPIECES / C::NTHR;
        if constexpr (NFULL > 0) lds_dma_burst<NFULL, C::NTHR * 16>(reinterpret_cast<const float* const*>(xsrc), dst);
#pragma unroll
        for (int u = NFULL; u < C::NUX; ++u)
            if (xlive[u]) lds_dma_burst<1, C::NTHR * 16>(reinterpret_cast<const float* const*>(xsrc + u), dst + (unsigned)(u * C::NTHR * 16));
#pragma unroll
        for (int u = 0; u < C::NUX; ++u) xsrc[u] += xstep[u];
    };
    auto dma_w = [&](int stage) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (unsigned)(2 * C::XBYTES + stage * C::WBYTES) + wave_off;
        constexpr int NFULL = C::WPIECES / C::NTHR;
        if constexpr (NFULL > 0) lds_dma_burst<NFULL, C::NTHR * 16>(reinterpret_cast<const float* const*>(wsrc), dst);
#pragma unroll
        for (int u = NFULL; u < C::NUW; ++u)
            if (wlive[u]) lds_dma_burst<1, C::NTHR * 16>(reinterpret_cast<const float* const*>(wsrc + u), dst + (unsigned)(u * C::NTHR * 16));
#pragma unroll
        for (int u = 0; u < C::NUW; ++u) wsrc[u] += C::WPIECES;
    };

    f32x16b acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // LDS read bases (in 16-byte fragments): A = weights [piece][octet][128 rows], B = activations [piece][octet][XT positions]
    const int a_frag = h * C::BM + wm * (32 * MT) + l31;
    const int b_frag = h * C::XT + wn * (32 * NT) + l31;
    const bf16x8* const xl = reinterpret_cast<const bf16x8*>(lds);
    const bf16x8* const wl = reinterpret_cast<const bf16x8*>(lds + 2 * C::XBYTES);

    // ---- prologue: activation tile of block 0, weight slices 0 .. NST-2 (n_steps >= 7 > NST - 1)
    dma_x(0);
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) dma_w(k);

    // One K step = one (channel block, tap).  Issue order of the DMA: slice s + NST - 1 at step s; the NEXT block's activation tile at
    // tap 1 (behind that step's slice).  vmcnt counts in order, so "slice s has landed" = at most the instructions issued after it
    // are still outstanding: the slices s+1 .. s+NST-2 that exist, plus the next tile at taps 2 .. NST (it is younger than slice
    // 7 cb + NST and older than slice 7 cb + NST + 1 <= 7 (cb + 1), so it has landed when the next block starts).  Waiting for FEWER
    // outstanding than that is always safe: the tile counts with its minimum (NUX - 1).
    auto wait_slice = [&](int s, bool tile_behind) __attribute__((always_inline)) {
        int k = n_steps - 1 - s;
        k = k < NST - 2 ? k : NST - 2;
        vm_wait_dyn(k * nw_issue + (tile_behind ? C::NUX - 1 : 0));
    };
    (void)nx_issue;

    int wst = 0;                                                 // ring stage of slice s
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more_cb = cb + 1 < ncb;
        const bf16x8* const xb = xl + (cb & 1) * C::XPIECES + b_frag;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) {
            const int s = cb * 7 + tap;
            wait_slice(s, more_cb && tap >= 2 && tap <= NST);
            __syncthreads();                                     // slice s (and at tap 0 the tile) visible to every wave; the stage of slice s-1 is free
            if (s + NST - 1 < n_steps) dma_w(wst >= 1 ? wst - 1 : NST - 1);   // (s + NST - 1) % NST
            if (tap == 1 && more_cb) dma_x((cb + 1) & 1);
            const bf16x8* const wb = wl + wst * C::WPIECES + a_frag;
            bf16x8 af[MT][NP], bq[NT][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i][p] = wb[p * 2 * C::BM + i * 32];
#pragma unroll
                for (int j = 0; j < NT; ++j) bq[j][p] = xb[p * 2 * C::XT + j * 32 + tap * DIL];
            }
            // piece products per (i, j), smallest first -- bf16x6: (w2,x0) (w0,x2) (w1,x1) (w1,x0) (w0,x1) (w0,x0); f16x3: (w1,x0) (w0,x1)
            // (w0,x0).  The accumulators rotate inside each product, so a dependent MFMA on one accumulator is MT*NT issues behind its
            // producer.
            constexpr int NPROD = NP == 3 ? 6 : 3;
            constexpr int PW6[6] = {2, 0, 1, 1, 0, 0}, PX6[6] = {0, 2, 1, 0, 1, 0};
            constexpr int PW3[3] = {1, 0, 0}, PX3[3] = {0, 1, 0};
#pragma unroll
            for (int k = 0; k < NPROD; ++k)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (NP == 3)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PW6[k]], bq[j][PX6[k]], acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PW3[k]]),
                                                                               __builtin_bit_cast(f16x8, bq[j][PX3[k]]), acc[i][j], 0, 0, 0);
                    }
            wst = wst == NST - 1 ? 0 : wst + 1;
        }
    }

    // f16x3: undo the two power-of-two scales of the split (exact)
    float oscale = 1.0f;
    if constexpr (NP == 2) oscale = f16_inv_scale(a.xamax[b]) * f16_inv_scale(a.wamax[0]);
    // ---- epilogue straight from the accumulators: register r of a 32x32 tile is row (r & 3) + 8 (r >> 2) + 4 h, the lane is the
    // column, so one store instruction writes two 128-byte row segments.  bias, then the Snake (the same det_snake as the exact path).
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * (32 * MT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float bv = a.bias ? a.bias[m] : 0.0f;
            const float al = a.alpha_out ? a.alpha_out[m] : 1.0f;
            const float inv = 1.0f / (al + 1e-9f);
            const float dal = a.dsn_src ? a.dsn_alpha[m] : 1.0f;
            const float dinv = 1.0f / (dal + 1e-9f);
            const size_t rowoff = ((size_t)b * a.Cout + m) * T;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * (32 * NT) + j * 32 + l31;
                if (n >= a.T) continue;
                float v = acc[i][j][r];
                if constexpr (NP == 2) v = v * oscale;
                v = v + bv;
                if (a.dsn_src) v = v * det_dsnake(a.dsn_src[rowoff + n], dal, dinv);     // same order as the exact dgrad epilogue
                if (a.residual) v = v + a.residual[rowoff + n];
                const bool tail = a.tvalid > 0 && n >= a.tvalid;
                if (a.y2) {
                    a.y[rowoff + n] = tail ? 0.0f : v;
                    a.y2[rowoff + n] = tail ? 0.0f : det_snake(v, al, inv);
                } else {
                    if (a.alpha_out) v = det_snake(v, al, inv);
                    a.y[rowoff + n] = tail ? 0.0f : v;
                }
            }
        }
}

template <int DIL, int MT, int NT, int WM, int WN, int NST, int NP>
static hipError_t launch_k7bf(const K7BfArgs& a_in, hipStream_t s)
{
    using C = K7BfCfg<DIL, MT, NT, WM, WN, NST, NP>;
    if (a_in.Cout % C::BM != 0) return hipErrorInvalidValue;
    K7BfArgs a = a_in;
    a.n_tiles = (a.T + C::BN - 1) / C::BN;
    auto kern = conv_k7_pieces_kernel<DIL, MT, NT, WM, WN, NST, NP>;
    {
        static BigLdsOptIn opt;
        const hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern));
        if (e != hipSuccess) return e;
    }
    int pi = -1;
    if (prof_enabled()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv_k7_pieces_kernel<%d, %d, %d, %d, %d, %d, %d>", DIL, MT, NT, WM, WN, NST, NP);
        const int cols = a.tvalid > 0 ? a.tvalid : a.T;
        pi = prof_begin(nm, 2.0 * a.Cin * 7.0 * a.Cout * (double)cols * a.B, s);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.n_tiles * a.B), (unsigned)(a.Cout / C::BM)), dim3(C::NTHR), C::LDS_BYTES, s, a);
    prof_end(pi, s);
    return hipGetLastError();
}

hipError_t launch_bf16x3_split(const float* x, void* xs, int batch, int c, int t, hipStream_t s)
{
    const size_t total = (size_t)batch * (c / 8) * t;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(bf16x3_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<bf16x8*>(xs), c, t, total);
    return hipGetLastError();
}

// rows per weight slice = the row tile of the kernel that will read the image: 128 where Cout allows, else 96 (C = 192)
int bf16x6_tile_rows(int cout) { return cout % 128 == 0 ? 128 : (cout % 96 == 0 ? 96 : 0); }
// f16x3 (two pieces: smaller slices): a 192-row tile on 2 x 2 waves where it divides Cout and 128 does not (C = 192: the activation
// tile is staged once instead of twice, 10 instead of 12 operand reads per 18 MFMAs); MVQ_F16_NO192=1: A/B knob
int f16x3_tile_rows(int cout)
{
    static const bool no192 = getenv("MVQ_F16_NO192") != nullptr;
    if (cout % 128 != 0 && cout % 192 == 0 && !no192) return 192;
    return bf16x6_tile_rows(cout);
}

hipError_t launch_bf16x3_pack_k7(const float* w, void* wq, int cout, int cin, int flip, hipStream_t s)
{
    const int bm = bf16x6_tile_rows(cout);
    if (bm == 0) return hipErrorInvalidValue;
    const size_t total = (size_t)(cout / bm) * (cin / 16) * 7 * 2 * bm;
    hipLaunchKernelGGL(bf16x3_pack_k7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<bf16x8*>(wq), cout, cin, bm, flip, total);
    return hipGetLastError();
}

// Tile per row count: 128 x 128 on 2 x 2 waves, or 96 x 128 on 1 x 4 waves where 128 does not divide Cout (C = 192); two blocks per
// CU, weight ring of three.  Measured and not kept (profiles/r04_timing_experiments.json::bf16x6_forms): 128 x 256 tiles on eight
// waves with a ring of four or six slices -- the same 205-220 TFLOP/s on every layer: the loop is not waiting for its DMA, the chip
// holds its clock down under the bf16 matrix load (MI355X guide, "clock under load").
hipError_t launch_conv_k7_bf16x6(const void* xs, const void* wq, const float* bias, const float* alpha_out, float* y, int batch, int cin,
                                 int t, int cout, int dil, int tvalid, const K7Extra& ex, hipStream_t s)
{
    K7BfArgs a{};
    a.y2 = ex.y2; a.dsn_src = ex.dsn_src; a.dsn_alpha = ex.dsn_alpha; a.residual = ex.residual;
    a.xs = reinterpret_cast<const bf16x8*>(xs); a.wq = reinterpret_cast<const bf16x8*>(wq); a.bias = bias; a.alpha_out = alpha_out; a.y = y;
    a.B = batch; a.Cin = cin; a.Cout = cout; a.T = t; a.tvalid = tvalid;
    const int bm = bf16x6_tile_rows(cout);
    if (bm == 96) {
        switch (dil) {
            case 1: return launch_k7bf<1, 3, 1, 1, 4, 3, 3>(a, s);
            case 3: return launch_k7bf<3, 3, 1, 1, 4, 3, 3>(a, s);
            case 9: return launch_k7bf<9, 3, 1, 1, 4, 3, 3>(a, s);
        }
    } else if (bm == 128) {
        switch (dil) {
            case 1: return launch_k7bf<1, 2, 2, 2, 2, 3, 3>(a, s);
            case 3: return launch_k7bf<3, 2, 2, 2, 2, 3, 3>(a, s);
            case 9: return launch_k7bf<9, 2, 2, 2, 2, 3, 3>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

// ---- f16x3 form -----------------------------------------------------------------------------------------------------------------
static hipError_t launch_f16_amax(const float* x, unsigned* amax, int items, size_t per_item, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned) * (size_t)items, s);
    if (e != hipSuccess) return e;
    unsigned bx = (unsigned)((per_item + 256 * 64 - 1) / (256 * 64));        // ~64 elements per thread
    bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
    hipLaunchKernelGGL(f16_amax_kernel, dim3(bx, (unsigned)items), dim3(256), 0, s, x, amax, per_item);
    return hipGetLastError();
}
// (Measured and not kept, profiles/r04_timing_experiments.json::f16_split_groups: the two passes over groups of items that fit the
// Infinity Cache, so that the split's read would hit it -- 231 -> 246 ms per step at 128 MB groups, 254 at 64 MB: the extra, smaller
// launches cost more than the cached read saves.)
hipError_t launch_f16x2_split(const float* x, void* xs, unsigned* xamax, int batch, int c, int t, hipStream_t s)
{
    const size_t total = (size_t)batch * (c / 8) * t;
    if (total == 0) return hipSuccess;
    const hipError_t e = launch_f16_amax(x, xamax, batch, (size_t)c * t, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(f16x2_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<f16x8*>(xs), xamax, c, t, total);
    return hipGetLastError();
}
hipError_t launch_f16x2_pack_k7(const float* w, void* wq, unsigned* wamax, int cout, int cin, int flip, hipStream_t s)
{
    const int bm = f16x3_tile_rows(cout);
    if (bm == 0) return hipErrorInvalidValue;
    const hipError_t e = launch_f16_amax(w, wamax, 1, (size_t)cout * cin * 7, s);
    if (e != hipSuccess) return e;
    const size_t total = (size_t)(cout / bm) * (cin / 16) * 7 * 2 * bm;
    hipLaunchKernelGGL(f16x2_pack_k7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<f16x8*>(wq), wamax, cout, cin, bm, flip, total);
    return hipGetLastError();
}
hipError_t launch_conv_k7_f16x3(const void* xs, const unsigned* xamax, const void* wq, const unsigned* wamax, const float* bias,
                                const float* alpha_out, float* y, int batch, int cin, int t, int cout, int dil, int tvalid, const K7Extra& ex,
                                hipStream_t s)
{
    K7BfArgs a{};
    a.y2 = ex.y2; a.dsn_src = ex.dsn_src; a.dsn_alpha = ex.dsn_alpha; a.residual = ex.residual;
    a.xs = reinterpret_cast<const bf16x8*>(xs); a.wq = reinterpret_cast<const bf16x8*>(wq); a.bias = bias; a.alpha_out = alpha_out; a.y = y;
    a.B = batch; a.Cin = cin; a.Cout = cout; a.T = t; a.tvalid = tvalid; a.xamax = xamax; a.wamax = wamax;
    const int bm = f16x3_tile_rows(cout);
    if (bm == 192) {
        switch (dil) {
            case 1: return launch_k7bf<1, 3, 2, 2, 2, 3, 2>(a, s);
            case 3: return launch_k7bf<3, 3, 2, 2, 2, 3, 2>(a, s);
            case 9: return launch_k7bf<9, 3, 2, 2, 2, 3, 2>(a, s);
        }
    } else if (bm == 96) {
        switch (dil) {
            case 1: return launch_k7bf<1, 3, 1, 1, 4, 3, 2>(a, s);
            case 3: return launch_k7bf<3, 3, 1, 1, 4, 3, 2>(a, s);
            case 9: return launch_k7bf<9, 3, 1, 1, 4, 3, 2>(a, s);
        }
    } else if (bm == 128) {
        // Rows of >= 2 048 columns: 128 x 256 tile, wave tile 64 x 128 -- 0.5 instead of 0.67 operand reads per MFMA and half the weight
        // staging per MFMA.  The loop is power-bound, so less energy per MFMA is a higher clock (the guide's rule 28): 355 -> 365 and
        // 368 -> 383 TFLOP/s on the T = 3 000 layers, 233.1 -> 231.0 ms per step.  MVQ_F16_NO_WIDE=1: A/B knob.
        static const bool no_wide = getenv("MVQ_F16_NO_WIDE") != nullptr;
        if (!no_wide && t >= 2048) {
            switch (dil) {
                case 1: return launch_k7bf<1, 2, 4, 2, 2, 3, 2>(a, s);
                case 3: return launch_k7bf<3, 2, 4, 2, 2, 3, 2>(a, s);
                case 9: return launch_k7bf<9, 2, 4, 2, 2, 3, 2>(a, s);
            }
        }
        switch (dil) {
            case 1: return launch_k7bf<1, 2, 2, 2, 2, 3, 2>(a, s);
            case 3: return launch_k7bf<3, 2, 2, 2, 2, 3, 2>(a, s);
            case 9: return launch_k7bf<9, 2, 2, 2, 2, 3, 2>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace mvq
