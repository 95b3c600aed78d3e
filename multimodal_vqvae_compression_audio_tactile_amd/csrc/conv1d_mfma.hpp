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
// Block = 64*WAVES_M*WAVES_N threads (4 or 8 waves); wave tile = (32*MT) x (32*NT); block tile BM x BN.  K is walked in chunks
// of CK input channels (CK*KS even).  Pipeline per chunk c (one barrier per chunk, two LDS buffers):
//     issue the global loads of chunk c+1 into registers (raw, branch-free, 16-byte when rows are aligned)
//     MFMA over chunk c out of LDS, operands for k-step s+1 fetched before the MFMAs of step s
//     Snake1d on the loaded registers (per-channel alpha / 1/alpha from an LDS table) + LDS store of chunk c+1
// so global latency hides behind the MFMAs and the Snake VALU work of one wave overlaps the other
// co-resident wave's matrix work.  bias / residual / Snake1d behind the conv / tanh run on the accumulators.
// SHUFFLE=true is the polyphase form of ConvTranspose1d (kernel 2*S, stride S): a 2-tap conv over
// M = Cout*S rows whose row (co*S + r) is written to y[co, q*S + r - P].
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "det_math.hpp"
#include "kernels_small.hpp"

namespace mvq {

struct ConvArgs;
inline bool conv_dma_rows_ok(const ConvArgs& a);

// 16 bytes of zeros in global memory: what the LDS-DMA staging reads for pieces that lie outside their row (conv zero padding)
__device__ __attribute__((aligned(16))) static const float g_zero16[4] = {0.0f, 0.0f, 0.0f, 0.0f};

// Per-launch HIP-event profiler (api.hip; mvq_profile_begin / mvq_profile_end in include/mvq.h).  Off: prof_begin returns -1
// and nothing is recorded.  On: one event pair per kernel launch on the launch stream, with the launch's algorithmic FLOPs.
int prof_begin(const char* kernel_name, double flops, hipStream_t s);
void prof_end(int idx, hipStream_t s);
bool prof_enabled();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    int Mrows;              // valid GEMM rows (Cout, or Cout*S for UPS)
    int Ncols;              // GEMM columns per batch element (Tout, or Tin+1 for UPS)
    int n_tiles;            // column tiles per batch element of THIS launch: ceil((Ncols - n_base) / BN), at most n_tiles_max
    int n_base;             // first GEMM column of this launch (0 unless the row is split into a wide-tile launch and a tail launch)
    int n_tiles_max;        // > 0: this launch covers only that many column tiles (the rest belongs to the tail launch)
    int row_fast;           // 0: grid (column tiles, row tiles).  R > 0: 1-D grid, XCD-aware: the R row tiles of a column
                            // tile are consecutive on ONE XCD (workgroup id mod 8), so the x tile is fetched into that L2 once
    int act;
    int dma;                // 1: stage the K chunks with global -> LDS DMA (3-deep ring); set by the launcher for eligible shapes
    int up_s, up_p;         // UPS: stride S and torch padding P
    int vec4;               // input rows are 16-byte aligned (Tin % 4 == 0 and x 16-byte aligned)
    int ovec4;              // output (and residual) rows are 16-byte aligned (Tout % 4 == 0, pointers aligned)
    // fused ResidualUnit (FUSE kernels): y = x + conv1(snake_mid(conv7(snake_in(x)) + bias) ) + bias2, then alpha_out
    const float* alpha_mid; // [C] Snake between the two convs
    const float* w2p;       // packed 1x1 weights [(c) * Mpad + co]
    const float* bias2;     // [C] or null
    // dual output: y2 = snake(v, alpha2) of the value BEFORE alpha_out/act, i.e. the Snake1d the next ResidualUnit
    // applies to its input.  Hoists that Snake out of the consumer's staging loop (where a layer with M/BM row tiles
    // would evaluate it M/BM times per element); the consumer then needs no alpha_in.
    float* y2;
    const float* alpha2;
    // backward (dgrad) epilogue: v = acc * d snake(dsn_src)/dx  (+ residual) -- the Snake1d backward of the layer in
    // front, fused behind the input-gradient conv; dsn_src[B,Cout,Tout] is that Snake's saved input, dsn_alpha[Cout].
    const float* dsn_src;
    const float* dsn_alpha;
    int tvalid;             // > 0: rows are zero-padded beyond their true length (a multiple-of-4 row length keeps every
                            // load / store 16-byte aligned): output columns >= tvalid are written as zeros, so the tail stays
                            // a valid zero padding for the next conv.  0: every column is data.
    // PACKED latent-rate rows: a row holds `seg` segments at a period of `tper` columns, `tper_valid` of
    // them data and the rest zeros (the conv's own zero padding between neighbours).  Stride-1 convs: output columns with
    // (n mod tper) >= tper_valid are written as zeros.  Polyphase ConvTranspose1d reading such rows: output sample t of the
    // packed row belongs to segment t / up_per_out at position t mod up_per_out, is kept only below up_valid_out, and goes to
    // batch item b * up_seg + segment of the UNPACKED output y[up_btrue, Cout, Tout].  Magic numbers: q = umulhi(n, magic).
    int tper, tper_valid;
    unsigned tper_magic;
    int up_per_out, up_valid_out, up_seg, up_btrue;
    unsigned up_magic;
    // VIRTUALLY PACKED rows (latent-rate layers, include/mvq.h mvq_conv1d_vpacked_f32): the GEMM's B matrix is a VIRTUAL row that
    // holds vp_seg segments at a period of vp_per_in columns, but neither the input nor the output is repacked in memory -- the
    // LDS-DMA source of every 16-byte piece is remapped (virtual column c -> item b * vp_seg + c / vp_per_in, position
    // c mod vp_per_in; positions >= vp_valid_in and items >= vp_btrue read the zero block) and the epilogue maps an output column
    // n -> item b * vp_seg + n / vp_per_out, position n mod vp_per_out of the ordinary [vp_btrue, Cout, Tout] tensor (Tout =
    // the physical row length, positions >= vp_valid_out written as zeros).  LDS-DMA staging + regular epilogue only.
    int vp_seg, vp_per_in, vp_valid_in, vp_tin_phys, vp_btrue, vp_per_out, vp_valid_out;
    unsigned vp_magic_in, vp_magic_out;
    char* name_out;         // host only: when set, launchers write the kernel instantiation name here and do not launch
    int name_len;
};

// ---- timing-build fence ------------------------------------------------------------------------------------------------------
// MVQ_EXP (pieces of a kernel compiled out: WRONG RESULTS by construction, only the clock is read), MVQ_KGROUP / MVQ_KPREFETCH
// (operand-read grouping A/B), MVQ_NO_RES_PREFETCH and MVQ_ASM_READS > 1 exist for tools/conv_microbench.py's timing builds
// only (MVQ_ASM_READS=0, the compiler-scheduled operand loop, is a correct fallback for a toolchain the ISA lint rejects: it
// builds without the fence and shows up as an informational bit).  They compile only together with -DMVQ_TIMING_BUILD, and such a library reports itself through
// mvq_build_flags() (include/mvq.h) -- _lib.lib() refuses to load it unless the caller opted in, bench.py prints the value.
#ifndef MVQ_KPREFETCH
#define MVQ_KPREFETCH 1
#endif
#ifndef MVQ_ASM_READS
#define MVQ_ASM_READS 1
#endif
#if (defined(MVQ_EXP) || defined(MVQ_KGROUP) || defined(MVQ_NO_RES_PREFETCH) || MVQ_KPREFETCH != 1 || MVQ_ASM_READS > 1) && !defined(MVQ_TIMING_BUILD)
#error "MVQ_EXP / MVQ_KGROUP / MVQ_KPREFETCH / MVQ_NO_RES_PREFETCH / MVQ_ASM_READS are timing-build switches: add -DMVQ_TIMING_BUILD (the library then reports mvq_build_flags() != 0)"
#endif
// bits of mvq_build_flags() that come from the compilation (api.hip adds the environment bits)
#define MVQ_BF_TIMING_BUILD 0x1
#define MVQ_BF_EXP 0x2
#define MVQ_BF_KGROUP 0x4
#define MVQ_BF_NO_RES_PREFETCH 0x8
#define MVQ_BF_ASM_READS 0x10
#define MVQ_BF_ASM_READS_OFF 0x10000     /* informational: compiler-scheduled operand loop (correct, slower) */
#define MVQ_BF_ENV_NO_DMA 0x100
#define MVQ_BF_ENV_ROWFAST 0x200
#define MVQ_BF_ENV_NO_TOKEN_RVQ 0x400
#define MVQ_BF_ENV_LAT_TILES 0x1000
#define MVQ_BF_ENV_SMALL_TILES 0x8000
constexpr unsigned conv_compile_flags()
{
    unsigned f = 0;
#ifdef MVQ_TIMING_BUILD
    f |= MVQ_BF_TIMING_BUILD;
#endif
#ifdef MVQ_EXP
    f |= MVQ_BF_EXP;
#endif
#if defined(MVQ_KGROUP) || MVQ_KPREFETCH != 1
    f |= MVQ_BF_KGROUP;
#endif
#ifdef MVQ_NO_RES_PREFETCH
    f |= MVQ_BF_NO_RES_PREFETCH;
#endif
#if MVQ_ASM_READS > 1
    f |= MVQ_BF_ASM_READS;
#elif MVQ_ASM_READS == 0
    f |= MVQ_BF_ASM_READS_OFF;
#endif
    return f;
}
// environment overrides (A/B measurements) seen by a launcher: recorded once, reported by mvq_build_flags()
void note_env_override(unsigned bit);

// k-steps per operand-read group (see conv1d_mfma_body): 1 in a product build.  Timing builds: -DMVQ_KGROUP=n fixes the group
// size, -DMVQ_KPREFETCH=0 selects the single-buffered form (reads of a group, wait, its MFMAs).
constexpr int kgroup_steps(int ns, int regs_per_step)
{
#ifdef MVQ_KGROUP
    return MVQ_KGROUP < ns ? MVQ_KGROUP : ns;
#else
    (void)ns; (void)regs_per_step;
    return 1;
#endif
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ---- hand-placed operand reads of the LDS-DMA K loop (MVQ_ASM_READS = G k-steps per group; 1 in a product build; 0 = compiler-scheduled) ------
// Left to itself the compiler feeds the MFMAs with ds_read2_b32 pairs, whose 8-bit offsets reach 255 dwords, so it re-bases the
// LDS address with a v_add per k-step (16 per 56-MFMA chunk on the 7-tap tile) -- vector instructions the fp32 MFMAs pay for
// (DESIGN.md section 6b).  This form issues every read as ds_read_b32 with a 16-bit immediate offset from three per-chunk base
// addresses (1 VALU per chunk instead of 16), double-buffered: [reads of group g+1][s_waitcnt lgkmcnt(#reads of g+1)][MFMAs of
// group g].  Measured (same box, 256-segment step): G = 1 331.5 -> 329.0 ms, every -m gpu test bit-exact; G = 2 / 4 are 1 % / 4 %
// SLOWER than the compiler's schedule (as its own grouped forms are).
//
// What makes the counted wait sound, and what keeps the compiler out of it:
//  * LDS reads of one wave return in order, and a counted lgkmcnt(N) only ever waits LONGER when something else (a scalar load)
//    is outstanding too: with the N reads of group g+1 behind them, "at most N outstanding" implies every read of group g is done.
//  * LLVM's waitcnt pass does not see loads issued from inline asm, so the registers a ds_read asm defines are IN FLIGHT until the
//    wait.  The wait asm therefore takes the group's registers as "+v" operands: every MFMA consumes the value the WAIT defines,
//    not the one the read defines, so no pass can move a consumer above the wait, and between a read and its wait nothing but
//    further reads / the DMA issue is emitted (sched_barrier on both sides).  What this cannot exclude is a register copy or
//    spill of an in-flight register inserted by the allocator; tools/isa_lint.py (run by tests/test_isa_lint.py, CPU suite)
//    disassembles the shipped code object and fails if ANY instruction touches a ds_read destination before the s_waitcnt that
//    covers it, if a DMA-ring kernel uses scratch, or if an M0 write is not followed by s_nop + global_load_lds + restore.
template <int OFF>
__device__ __forceinline__ float lds_read_imm(unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate offset is 16 bits");
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return v;
}
// s_waitcnt lgkmcnt(P) that DEFINES the registers it makes valid (tied "+v" operands)
template <int P> __device__ __forceinline__ void lgkm_wait_tie(float& a, float& b)
{ asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(P) : "memory"); }
template <int P> __device__ __forceinline__ void lgkm_wait_tie(float& a, float& b, float& c)
{ asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(P) : "memory"); }
template <int P> __device__ __forceinline__ void lgkm_wait_tie(float& a, float& b, float& c, float& d)
{ asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(P) : "memory"); }
template <int P> __device__ __forceinline__ void lgkm_wait_tie(float& a, float& b, float& c, float& d, float& e)
{ asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(P) : "memory"); }
template <int P> __device__ __forceinline__ void lgkm_wait_tie(float& a, float& b, float& c, float& d, float& e, float& f)
{ asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(P) : "memory"); }
// one k-step's operands (MT + NT registers, 2..6)
template <int P, int MT, int NT>
__device__ __forceinline__ void lgkm_wait_step(float (&av)[MT], float (&bv)[NT])
{
    static_assert(MT >= 1 && NT >= 1 && MT + NT <= 6, "operand registers per k-step");
    if constexpr (MT == 1 && NT == 1) lgkm_wait_tie<P>(av[0], bv[0]);
    else if constexpr (MT == 2 && NT == 1) lgkm_wait_tie<P>(av[0], av[1], bv[0]);
    else if constexpr (MT == 1 && NT == 2) lgkm_wait_tie<P>(av[0], bv[0], bv[1]);
    else if constexpr (MT == 2 && NT == 2) lgkm_wait_tie<P>(av[0], av[1], bv[0], bv[1]);
    else if constexpr (MT == 3 && NT == 1) lgkm_wait_tie<P>(av[0], av[1], av[2], bv[0]);
    else if constexpr (MT == 1 && NT == 3) lgkm_wait_tie<P>(av[0], bv[0], bv[1], bv[2]);
    else if constexpr (MT == 4 && NT == 1) lgkm_wait_tie<P>(av[0], av[1], av[2], av[3], bv[0]);
    else if constexpr (MT == 1 && NT == 4) lgkm_wait_tie<P>(av[0], bv[0], bv[1], bv[2], bv[3]);
    else if constexpr (MT == 3 && NT == 2) lgkm_wait_tie<P>(av[0], av[1], av[2], bv[0], bv[1]);
    else if constexpr (MT == 2 && NT == 3) lgkm_wait_tie<P>(av[0], av[1], bv[0], bv[1], bv[2]);
    else if constexpr (MT == 4 && NT == 2) lgkm_wait_tie<P>(av[0], av[1], av[2], av[3], bv[0], bv[1]);
    else if constexpr (MT == 2 && NT == 4) lgkm_wait_tie<P>(av[0], av[1], bv[0], bv[1], bv[2], bv[3]);
    else if constexpr (MT == 3 && NT == 3) lgkm_wait_tie<P>(av[0], av[1], av[2], bv[0], bv[1], bv[2]);
    else static_assert(MT + NT <= 5 || MT == 4 || NT == 4 || (MT == 3 && NT == 3), "unsupported operand shape");
}

// ---- LDS-DMA burst: N consecutive full pieces of one wave (destinations INC bytes apart) in ONE statement --------------------
// M0 (the LDS destination base) is compiler-reserved: it is saved once, written before every piece, and restored before the
// statement ends; `s_nop 0` is the wait state gfx9 needs between an SALU write of M0 and the LDS-DMA instruction that reads it
// (the assembler pads nothing inside an asm string).  Round 4: one save / restore per BURST instead of per piece (a 7-tap chunk
// issues 4 full pieces per wave: 14 instead of 20 instructions in the MFMA wave's stream); tools/isa_lint.py R4 checks the shape.
#define MVQ_DMA_FIRST(i) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #i ", off\n\t"
#define MVQ_DMA_NEXT(i) "s_add_u32 m0, m0, %[inc]\n\t" MVQ_DMA_FIRST(i)
#define MVQ_DMA_HEAD "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %[dst]\n\t"
#define MVQ_DMA_TAIL "s_mov_b32 m0, %0"
template <int N, int INC>
__device__ __forceinline__ void lds_dma_burst(const float* const* src, unsigned dst)
{
    static_assert(N >= 1 && N <= 8, "pieces per burst");
    unsigned keep;
    if constexpr (N == 1)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_TAIL : "=&s"(keep) : "v"(src[0]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 2)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 3)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 4)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_NEXT(4) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 5)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_NEXT(4) MVQ_DMA_NEXT(5) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 6)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_NEXT(4) MVQ_DMA_NEXT(5) MVQ_DMA_NEXT(6) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else if constexpr (N == 7)
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_NEXT(4) MVQ_DMA_NEXT(5) MVQ_DMA_NEXT(6) MVQ_DMA_NEXT(7) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
    else
        asm volatile(MVQ_DMA_HEAD MVQ_DMA_FIRST(1) MVQ_DMA_NEXT(2) MVQ_DMA_NEXT(3) MVQ_DMA_NEXT(4) MVQ_DMA_NEXT(5) MVQ_DMA_NEXT(6) MVQ_DMA_NEXT(7) MVQ_DMA_NEXT(8) MVQ_DMA_TAIL
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]), [dst] "s"(dst), [inc] "n"(INC) : "memory", "scc");
}

template <int KS, int STRIDE, int DIL, int BM, int XP, int MT, int NT, int NS, int G>
struct AsmOperandLoop {
    static constexpr int NG = (NS + G - 1) / G;
    template <int S, int I>
    static __device__ __forceinline__ void load_a(float (&av)[G][MT], unsigned a_addr)
    {
        if constexpr (I < MT) {
            av[S % G][I] = lds_read_imm<(2 * S * BM + I * 32) * 4>(a_addr);
            load_a<S, I + 1>(av, a_addr);
        }
    }
    template <int S, int J>
    static __device__ __forceinline__ void load_b(float (&bv)[G][NT], unsigned b_same, unsigned b_cross)
    {
        if constexpr (J < NT) {
            constexpr int k0 = 2 * S;
            constexpr int off0 = (k0 / KS) * XP + (k0 % KS) * DIL;
            constexpr bool cross = ((k0 + 1) / KS) != (k0 / KS);
            bv[S % G][J] = lds_read_imm<(off0 + J * 32 * STRIDE) * 4>(cross ? b_cross : b_same);
            load_b<S, J + 1>(bv, b_same, b_cross);
        }
    }
    template <int GI, int U>
    static __device__ __forceinline__ void load_group(float (&av)[G][MT], float (&bv)[G][NT], unsigned a_addr, unsigned b_same, unsigned b_cross)
    {
        if constexpr (U < G && GI * G + U < NS) {
            load_a<GI * G + U, 0>(av, a_addr);
            load_b<GI * G + U, 0>(bv, b_same, b_cross);
            load_group<GI, U + 1>(av, bv, a_addr, b_same, b_cross);
        }
    }
    template <int GI>
    static constexpr int group_reads() { return ((GI * G + G <= NS) ? G : (NS - GI * G > 0 ? NS - GI * G : 0)) * (MT + NT); }
    template <int GI, int U>
    static __device__ __forceinline__ void mfma_group(f32x16_t (&acc)[MT][NT], const float (&av)[G][MT], const float (&bv)[G][NT])
    {
        if constexpr (U < G && GI * G + U < NS) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[U][i], bv[U][j], acc[i][j], 0, 0, 0);
            mfma_group<GI, U + 1>(acc, av, bv);
        }
    }
    // the counted wait of group GI: the first k-step's registers ride through the s_waitcnt itself, the others (G > 1, timing
    // builds) through empty asm statements behind it (volatile asm statements keep their order)
    template <int GI, int PW, int U>
    static __device__ __forceinline__ void wait_group(float (&av)[G][MT], float (&bv)[G][NT])
    {
        if constexpr (U < G && GI * G + U < NS) {
            if constexpr (U == 0) lgkm_wait_step<PW, MT, NT>(av[0], bv[0]);
            else {
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(av[U][i]));
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bv[U][j]));
            }
            wait_group<GI, PW, U + 1>(av, bv);
        }
    }
    // groups GI, GI+1, ... ; buffers alternate (av0/bv0 for even groups)
    template <int GI>
    static __device__ __forceinline__ void run_from(f32x16_t (&acc)[MT][NT], float (&av0)[G][MT], float (&bv0)[G][NT], float (&av1)[G][MT],
                                                    float (&bv1)[G][NT], unsigned a_addr, unsigned b_same, unsigned b_cross)
    {
        if constexpr (GI < NG) {
            if constexpr (GI + 1 < NG) {
                if constexpr ((GI + 1) % 2 == 0) load_group<GI + 1, 0>(av0, bv0, a_addr, b_same, b_cross);
                else load_group<GI + 1, 0>(av1, bv1, a_addr, b_same, b_cross);
            }
            // LDS reads return in order: group GI has landed once at most the reads of group GI+1 are outstanding.  The wait
            // re-defines group GI's registers (tied operands): their consumers depend on the wait, not on the reads.
            constexpr int pending = (GI + 1 < NG) ? group_reads<GI + 1>() : 0;
            constexpr int PW = pending > 15 ? 15 : pending;
            __builtin_amdgcn_sched_barrier(0);         // nothing (no scalar load, no VALU) between the reads and their wait
            if constexpr (GI % 2 == 0) wait_group<GI, PW, 0>(av0, bv0);
            else wait_group<GI, PW, 0>(av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (GI % 2 == 0) mfma_group<GI, 0>(acc, av0, bv0);
            else mfma_group<GI, 0>(acc, av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
            run_from<GI + 1>(acc, av0, bv0, av1, bv1, a_addr, b_same, b_cross);
        }
    }
};

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, int UPS>
struct ConvCfg {
    static constexpr int BM = 32 * MT * WAVES_M;
    static constexpr int BN = 32 * NT * WAVES_N;
    static constexpr int KC = CK * KS;                                  // K elements per chunk
    static constexpr int XT = (BN - 1) * STRIDE + (KS - 1) * DIL + 1;    // input samples per row
    // a 1x1 conv (no padding: checked by the launcher) starts every tile on a 16-byte boundary of its row, so its rows need no
    // alignment slack: exactly BN / 4 pieces -- for the 128 x 128 tile that is 4 instead of 5 LDS-DMA instructions per chunk
    static constexpr bool XALIGNED = (KS == 1 && STRIDE == 1);
    static constexpr int XV = XALIGNED ? (XT + 3) / 4 : (XT + 3 + 3) / 4; // float4 per row (aligned start, shift <= 3)
    static constexpr int XTP = XV * 4 + 4;                               // row pitch (floats), multiple of 4
    static constexpr int W_FLOATS = KC * BM;
    static constexpr int X_FLOATS = CK * XTP;
    static constexpr int STAGE_FLOATS = 2 * (W_FLOATS + X_FLOATS);
    static constexpr int BNP = BN + 4;                                   // pitch of the epilogue tile
    static constexpr int CT_FLOATS = BM * BNP;
    // plain convs with an even number of row groups per wave drain the accumulators in two passes (half the rows each)
    // when that is what keeps the block's LDS footprint down (more co-resident blocks per CU)
    static constexpr int EH = ((UPS == 0 || 32 % (UPS ? UPS : 1) == 0) && MT % 2 == 0 && CT_FLOATS > STAGE_FLOATS) ? 2 : 1;
    static constexpr int CTH_FLOATS = CT_FLOATS / EH;
    // epilogue row table Tb, filled once per block right behind the epilogue tile (inside the staging region where that is the
    // larger one, so the K-loop footprint -- blocks per CU -- does not change)
    // Regular quad mapping of the epilogue (conv1d_mfma_body): the block's NTHR threads form CQ column-quad lanes x RSTEP rows,
    // CQ = gcd(BN / 4, NTHR); a thread keeps its JN = (BN / 4) / CQ column quads (one for the 128- / 64- / 256-column tiles,
    // three for the 96- and 192-column ones) and its rows advance by RSTEP per step.  Needs RSTEP | 32 (a step never leaves
    // its 32-row accumulator group) -- true for every stride-1 / strided tile in use.
    static constexpr int QPR = BN / 4;
    static constexpr int CQ = cgcd(QPR, 64 * WAVES_M * WAVES_N);
    static constexpr int JN = QPR / CQ;
    static constexpr int RSTEP = (64 * WAVES_M * WAVES_N) / CQ;
    static constexpr bool REG_GEOM = (UPS == 0) && (32 % RSTEP == 0);
    // Tb = [bias | alpha of the epilogue Snake][BM].  LDS is allocated in 1 280-byte granules on this chip (measured: a
    // 54 272-byte block runs two per CU, 53 568 three): the 128 x 96 tiles (52 736 bytes) have room for exactly these two arrays.
    static constexpr int TB_FLOATS = REG_GEOM ? 2 * BM : 0;
    static constexpr int EPI_FLOATS = CTH_FLOATS + TB_FLOATS;
    static constexpr int EPI_FLOATS_FUSE = CT_FLOATS + TB_FLOATS;
    static constexpr int LDS_FLOATS = STAGE_FLOATS > EPI_FLOATS ? STAGE_FLOATS : EPI_FLOATS;
    // Waves per SIMD the register allocator is asked to leave room for (= blocks per CU for a 4-wave block): 3 where the
    // block's LDS footprint fits three times into 160 KB (the kernels then need 112-165 VGPRs of the 168 allowed, no
    // spills).  Four (128 VGPRs) was measured on the 7-tap 128x128 tile: 7 dwords of scratch, +1.5 % in isolation, nothing
    // on the whole step -- not kept.
    static constexpr bool FOUR_WAVES = WAVES_M * WAVES_N == 4;
    // Instantiations whose 3-stage ring can never be selected (> 64 KB: the register-staged loop only, with its staging
    // registers live across the MFMAs) spilled under the 168-register cap; they take the 2-waves-per-SIMD budget.
    static constexpr bool RING_FITS = (size_t)3 * (KC * BM + CK * XV * 4) * 4 <= 64 * 1024;
    static constexpr int MIN_WPE = (FOUR_WAVES && LDS_FLOATS * 4 <= 53 * 1024 && !(WAVES_N == 4 && NT == 2) && RING_FITS && MT * NT <= 4) ? 3
                                                                                                                : (WAVES_M * WAVES_N) / 2;
    static constexpr int LDS_FLOATS_FUSE = STAGE_FLOATS > EPI_FLOATS_FUSE ? STAGE_FLOATS : EPI_FLOATS_FUSE;   // fused unit: full tile
    static constexpr int MIN_WPE_FUSE = (FOUR_WAVES && (LDS_FLOATS_FUSE + 5 * BM) * 4 * 3 <= 160 * 1024) ? 3 : (WAVES_M * WAVES_N) / 2;
    // LDS-DMA ring: a stage is the weight chunk followed by the activation chunk with dense rows (pitch XV*4), three stages
    static constexpr int XV4 = XV * 4;
    static constexpr int DMA_STAGE_FLOATS = W_FLOATS + CK * XV4;
    static constexpr int DMA_NV = DMA_STAGE_FLOATS / 4;                  // float4 pieces per stage
    static constexpr int DMA_NU = (DMA_NV + 64 * WAVES_M * WAVES_N - 1) / (64 * WAVES_M * WAVES_N);   // DMA instructions per wave per chunk
    static constexpr int LDS_FLOATS_DMA = 3 * DMA_STAGE_FLOATS > EPI_FLOATS ? 3 * DMA_STAGE_FLOATS : EPI_FLOATS;
    // fused unit fed by LDS-DMA (its input arrives pre-snaked): the 3-stage ring, then the full intermediate / epilogue tile
    static constexpr int LDS_FLOATS_FUSE_DMA = 3 * DMA_STAGE_FLOATS > EPI_FLOATS_FUSE ? 3 * DMA_STAGE_FLOATS : EPI_FLOATS_FUSE;
    static constexpr int W_VEC = W_FLOATS / 4;                           // float4 per chunk
    static constexpr int NTHR = 64 * WAVES_M * WAVES_N;                  // threads per block
    static constexpr int W_PER_THREAD = (W_VEC + NTHR - 1) / NTHR;
    static constexpr int XV_TOTAL = CK * XV;
    static constexpr int XV_PER_THREAD = (XV_TOTAL + NTHR - 1) / NTHR;         // float4 path
    static constexpr int XS_TOTAL = CK * XT;
    static constexpr int XS_PER_THREAD = (XS_TOTAL + NTHR - 1) / NTHR;         // scalar path
    static constexpr int XREGS = (4 * XV_PER_THREAD > XS_PER_THREAD) ? 4 * XV_PER_THREAD : XS_PER_THREAD;
    static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "block is 4 or 8 waves");
    static_assert(KC % 2 == 0, "chunk K must be even (32x32x2 MFMA)");
    static_assert(BM % 4 == 0, "float4 weight staging");
};

// Per-thread view of one block's staging / MFMA work.  Plain force-inlined member functions over register
// arrays (lambdas capturing the arrays by reference kept them in scratch memory).
template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, int UPS, bool VEC>
struct ConvTile {
    using C = ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS>;
    const float* wp; const float* xb; const float* Al;
    float* Ws; float* Xs;
    int tid, Tin, Cin, Mpad, m0, t_in0, g_al;
    bool snake_in;

    // stage 1: raw global loads of one chunk into registers (no dependent math here)
    __device__ __forceinline__ void load_chunk(int chunk, f32x4 (&wreg)[C::W_PER_THREAD], f32x4 (&xv)[C::XV_PER_THREAD], float (&xs)[C::XS_PER_THREAD]) const
    {
        const int ci0 = chunk * CK;
        const float* wsrc = wp + (size_t)ci0 * KS * Mpad + m0;
#pragma unroll
        for (int u = 0; u < C::W_PER_THREAD; ++u) {
            int v = tid + u * C::NTHR;
            if (C::W_VEC % C::NTHR != 0) v = v < C::W_VEC ? v : C::W_VEC - 1;
            const int row = v / (C::BM / 4);
            const int c4 = v - row * (C::BM / 4);
            wreg[u] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)row * Mpad + c4 * 4);
        }
        if (VEC) {
#pragma unroll
            for (int u = 0; u < C::XV_PER_THREAD; ++u) {
                int e = tid + u * C::NTHR;
                if (C::XV_TOTAL % C::NTHR != 0) e = e < C::XV_TOTAL ? e : C::XV_TOTAL - 1;
                const int cl = e / C::XV;
                const int v = e - cl * C::XV;
                const int g = g_al + 4 * v;                       // multiple of 4: fully inside or fully outside
                const bool ok = g >= 0 && g < Tin;
                const int gc = ok ? g : 0;
                xv[u] = *reinterpret_cast<const f32x4*>(xb + (size_t)(ci0 + cl) * Tin + gc);
            }
        } else {
#pragma unroll
            for (int u = 0; u < C::XS_PER_THREAD; ++u) {
                int e = tid + u * C::NTHR;
                if (C::XS_TOTAL % C::NTHR != 0) e = e < C::XS_TOTAL ? e : C::XS_TOTAL - 1;
                const int cl = e / C::XT;
                const int xi = e - cl * C::XT;
                const int g = t_in0 + xi;
                const bool ok = g >= 0 && g < Tin;
                const int gc = ok ? g : 0;
                xs[u] = xb[(size_t)(ci0 + cl) * Tin + gc];      // zero-fill of the halo happens in store_chunk
            }
        }
    }

    // stage 3: Snake1d on the registers, then LDS stores
    __device__ __forceinline__ void store_chunk(int chunk, int buf, const f32x4 (&wreg)[C::W_PER_THREAD],
                                                const f32x4 (&xv)[C::XV_PER_THREAD],
                                                const float (&xs)[C::XS_PER_THREAD]) const
    {
        const int ci0 = chunk * CK;
        float* wdst = Ws + buf * C::W_FLOATS;
        float* xdst = Xs + buf * C::X_FLOATS;
#pragma unroll
        for (int u = 0; u < C::W_PER_THREAD; ++u) {
            const int v = tid + u * C::NTHR;
            if (C::W_VEC % C::NTHR == 0 || v < C::W_VEC) *reinterpret_cast<f32x4*>(wdst + v * 4) = wreg[u];
        }
        if (VEC) {
#pragma unroll
            for (int u = 0; u < C::XV_PER_THREAD; ++u) {
                const int e = tid + u * C::NTHR;
                const int ec = (C::XV_TOTAL % C::NTHR != 0 && e >= C::XV_TOTAL) ? C::XV_TOTAL - 1 : e;
                const int cl = ec / C::XV;
                const int v = ec - cl * C::XV;
                const int g = g_al + 4 * v;
                const bool ok = g >= 0 && g < Tin;
                f32x4 q = xv[u];
                if (!ok) q = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#if defined(MVQ_EXP) && (MVQ_EXP & 8)
                if (false) {
#else
                if (snake_in) {
#endif
                    const float al = Al[ci0 + cl], inv = Al[Cin + ci0 + cl];
                    q.x = det_snake(q.x, al, inv); q.y = det_snake(q.y, al, inv);
                    q.z = det_snake(q.z, al, inv); q.w = det_snake(q.w, al, inv);
                }
                if (C::XV_TOTAL % C::NTHR == 0 || e < C::XV_TOTAL)
                    *reinterpret_cast<f32x4*>(xdst + cl * C::XTP + 4 * v) = q;
            }
        } else {
#pragma unroll
            for (int u = 0; u < C::XS_PER_THREAD; ++u) {
                const int e = tid + u * C::NTHR;
                const int ec = (C::XS_TOTAL % C::NTHR != 0 && e >= C::XS_TOTAL) ? C::XS_TOTAL - 1 : e;
                const int cl = ec / C::XT;
                const int xi = ec - cl * C::XT;
                const int g = t_in0 + xi;
                float q = (g >= 0 && g < Tin) ? xs[u] : 0.0f;
                if (snake_in) q = det_snake(q, Al[ci0 + cl], Al[Cin + ci0 + cl]);
                if (C::XS_TOTAL % C::NTHR == 0 || e < C::XS_TOTAL) xdst[cl * C::XTP + xi] = q;
            }
        }
    }
};

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, int UPS, bool VEC, bool FUSE>
__device__ __forceinline__ void conv1d_mfma_body(const ConvArgs& a)
{
    using C = ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ws = smem;                          // [2][KC][BM]
    float* const Xs = smem + 2 * C::W_FLOATS;        // [2][CK][XTP]
    // behind the staging / epilogue region: Ep = per-row 1/(alpha+1e-9) of the epilogue Snakes for this block's BM rows
    // -- [0] alpha_out, [1] alpha2 (dual output), [2] dsn_alpha (dgrad) or alpha_mid (fused unit) -- computed once per
    // block (the IEEE division is ~12 VALU instructions: per element it cost more than the Snake polynomial itself, and
    // on this chip every VALU instruction is time taken from the fp32 MFMAs: DESIGN.md section 6a);
    // then Al = [2][Cin]: alpha, 1/(alpha+1e-9) of the input Snake (only with alpha_in)
    float* const Ep = smem + (FUSE ? (a.dma ? C::LDS_FLOATS_FUSE_DMA : C::LDS_FLOATS_FUSE) : (a.dma ? C::LDS_FLOATS_DMA : C::LDS_FLOATS));
    constexpr int EP_MID = 2;                                // the fused unit never has a dgrad epilogue: alpha_mid takes that slot
    float* const Al = Ep + 3 * C::BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int bx = blockIdx.x, by = blockIdx.y;
    if (a.row_fast) {
        const int xcd = bx & 7, j = bx >> 3;
        by = j % a.row_fast;
        bx = (j / a.row_fast) * 8 + xcd;
        if (bx >= a.n_tiles * a.B) return;           // padding blocks of the last group of 8 column tiles
    }
    const int b = bx / a.n_tiles;
    const int tile_n = bx - b * a.n_tiles;
    const int n0 = a.n_base + tile_n * C::BN;
    const int m0 = by * C::BM;
    const int t_in0 = n0 * STRIDE - a.pad;           // input sample of tile column 0
    const int g_al = t_in0 & ~3;                     // 16-byte aligned start (floor, also for negatives)
    const int shift = VEC ? (t_in0 - g_al) : 0;      // LDS column of tile column 0
    const int Cin = a.Cin;
    const int n_chunks = Cin / CK;
    const bool snake_in = a.alpha_in != nullptr;

    if (snake_in) {
        for (int c = tid; c < Cin; c += C::NTHR) {
            const float al = a.alpha_in[c];
            Al[c] = al;
            Al[Cin + c] = 1.0f / (al + 1e-9f);
        }
    }
    for (int r = tid; r < C::BM; r += C::NTHR) {
        int m = m0 + r;
        m = m < a.Mrows ? m : a.Mrows - 1;
        const int ch = m / (UPS ? UPS : 1);                  // ConvTranspose1d: GEMM row (co*S + phase) -> channel co
        const float* tabs[4] = {a.alpha_out, a.y2 ? a.alpha2 : nullptr, FUSE ? a.alpha_mid : (a.dsn_src ? a.dsn_alpha : nullptr), nullptr};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Ep[k * C::BM + r] = tabs[k] ? 1.0f / (tabs[k][ch] + 1e-9f) : 1.0f;
        }
    }

    ConvTile<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS, VEC> tile;
    tile.wp = a.wp; tile.xb = a.x + (size_t)b * Cin * a.Tin; tile.Al = Al; tile.Ws = Ws; tile.Xs = Xs;
    tile.tid = tid; tile.Tin = a.Tin; tile.Cin = Cin; tile.Mpad = a.Mpad; tile.m0 = m0; tile.t_in0 = t_in0;
    tile.g_al = g_al; tile.snake_in = snake_in;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 wreg[C::W_PER_THREAD];
    f32x4 xv[C::XV_PER_THREAD];
    float xs[C::XS_PER_THREAD];

    // stage 2: MFMA over one chunk.
    // K index kidx = ci_local*KS + kk; a 32x32x2 step takes kidx = 2s (lanes 0-31) and 2s+1 (lanes 32-63).
    // LDS offset of kidx: (kidx / KS) * XTP + (kidx % KS) * DIL.  The lane-half delta off(2s+1) - off(2s) takes
    // at most two values (same channel: DIL; channel boundary: XTP - (KS-1)*DIL), folded into two lane bases.
    constexpr int D_SAME = DIL;
    constexpr int D_CROSS = C::XTP - (KS - 1) * DIL;
    const int a_base = h * C::BM + wm * (MT * 32) + l31;
    const int b_base = (wn * (NT * 32) + l31) * STRIDE + shift;
    const int b_same = b_base + h * D_SAME;
    const int b_cross = b_base + h * D_CROSS;

    // one chunk of MFMAs out of LDS buffer `buf`; operands of k-step s+1 are fetched before the MFMAs of step s
    // (Skipping the MFMAs of column subtiles that lie past the end of the row was measured: +0.3 % -- the block lasts as long
    // as its busiest wave.  Rows with a mostly empty last tile are split into two launches instead: conv_tail_width.)
    // Operand reads may be issued in GROUPS of KG k-steps, one group ahead of the MFMAs that consume them ([reads of group
    // g+1][MFMAs of group g], one lgkmcnt wait per group).  Measured on the real layers (round 3, gpurun_out/r3c): KG = 4
    // double-buffered and KG = 7 single-buffered are 3-5 % SLOWER than KG = 1 on the 7-tap layers (133 vs 138.5 TFLOP/s),
    // KG = 2 equal, although a bare LDS-fed MFMA loop with batched reads reaches 151-153 TFLOP/s (tools/mfma_probe.hip): in
    // the real loop the gap to that figure is the DMA issue (4 %), the per-chunk wait + barrier (1.3 %) and the epilogue
    // (1.6 %), not the read schedule (timing builds with each piece removed, gpurun_out/r3d).
    // The default therefore stays one k-step per group (operands of step s+1 fetched before the MFMAs of step s); the
    // grouped forms remain selectable for A/B builds (kgroup_steps).
    constexpr int NS = C::KC / 2;                                       // k-steps per chunk
    constexpr int KG = kgroup_steps(NS, MT + NT);
    constexpr int NG = (NS + KG - 1) / KG;
    constexpr bool KPRE = MVQ_KPREFETCH != 0;
    auto mfma_chunk = [&](int buf) __attribute__((always_inline)) {
        const float* wsrc = Ws + buf * C::W_FLOATS + a_base;
        const float* xs_same = Xs + buf * C::X_FLOATS + b_same;
        const float* xs_cross = Xs + buf * C::X_FLOATS + b_cross;
        float av[KPRE ? 2 : 1][KG][MT], bv[KPRE ? 2 : 1][KG][NT];
        auto load_group = [&](int g, int pb) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < KG; ++u) {
                const int st = g * KG + u;
                if (st < NS) {
                    const int k0 = 2 * st;
                    const int off0 = (k0 / KS) * C::XTP + (k0 % KS) * DIL;
                    const bool cross = ((k0 + 1) / KS) != (k0 / KS);
                    const float* xsp = cross ? xs_cross : xs_same;
#pragma unroll
                    for (int i = 0; i < MT; ++i) av[pb][u][i] = wsrc[k0 * C::BM + i * 32];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[pb][u][j] = xsp[off0 + j * 32 * STRIDE];
                }
            }
        };
        if (KPRE) load_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int pb = KPRE ? (g & 1) : 0;
            if (KPRE) { if (g + 1 < NG) load_group(g + 1, (g + 1) & 1); }
            else load_group(g, 0);
#pragma unroll
            for (int u = 0; u < KG; ++u)
                if (g * KG + u < NS) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pb][u][i], bv[pb][u][j], acc[i][j], 0, 0, 0);
                }
            // the reads (of group g+1 when prefetching, of this group otherwise) first, then the MFMAs of group g
            __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT) * KG, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT * KG, 0);
        }
    };

    // the LDS-DMA form exists only in instantiations whose 3-stage ring the launcher can ever select (launch_conv1d_mfma /
    // launch_residual_unit set a.dma under the same size conditions): the others do not carry its code or its registers
    constexpr bool DMA_FITS = FUSE ? ((size_t)C::LDS_FLOATS_FUSE_DMA + 3 * C::BM) * 4 * 2 <= 160 * 1024 : (size_t)C::LDS_FLOATS_DMA * 4 <= 64 * 1024;
    bool use_dma = false;
    if constexpr (VEC && DMA_FITS) use_dma = a.dma != 0;
    if (use_dma) {
      if constexpr (VEC && DMA_FITS) {
        // ---- K loop with global -> LDS DMA staging (global_load_lds_dwordx4: 16 bytes per lane, 1 KiB per wave-instruction,
        // no register round trip, no ds_write) into a ring of THREE stages: while chunk c is multiplied out of stage c % 3 the
        // DMA of chunk c+2 is in flight into stage (c+2) % 3, which was last read during chunk c-1, i.e. before the barrier
        // every wave passed at the end of that iteration.  What is left in a wave's instruction stream besides MFMAs and
        // operand reads: DMA_NU DMA issues + pointer increments per chunk, one vmcnt wait and one barrier.
        // Out-of-row pieces (the zero padding of the conv) read a 16-byte block of zeros instead.
        constexpr int NU = C::DMA_NU;
        constexpr int XP = C::XV4;                                        // dense activation rows
        constexpr int D_CROSS_D = XP - (KS - 1) * DIL;
        const int bd_same = b_base + h * D_SAME;
        const int bd_cross = b_base + h * D_CROSS_D;
        // per-lane source pointers of this thread's NU pieces (piece p = tid + u*NTHR of the stage), advanced per chunk
        const float* src[NU];
        bool live[NU];
        long long step_b[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int p = tid + u * C::NTHR;
            live[u] = p < C::DMA_NV;
            if (p < C::W_VEC) {
                const int row = p / (C::BM / 4), c4 = p - row * (C::BM / 4);
                src[u] = a.wp + (size_t)row * a.Mpad + m0 + c4 * 4;
                step_b[u] = (long long)CK * KS * a.Mpad;
            } else {
                const int q = (p < C::DMA_NV ? p : C::DMA_NV - 1) - C::W_VEC;
                const int cl = q / C::XV, v = q - cl * C::XV;
                const int g = g_al + 4 * v;
                bool ok = g >= 0 && g < a.Tin;
                const float* sp = tile.xb + (size_t)cl * a.Tin + g;
                long long st = (long long)CK * a.Tin;
                if (a.vp_seg) {                                        // virtually packed row: remap the piece to its item / position
                    const int gg = ok ? g : 0;
                    const int seg = (int)__umulhi((unsigned)gg, a.vp_magic_in), pos = gg - seg * a.vp_per_in;
                    const int item = b * a.vp_seg + seg;
                    ok = ok && pos < a.vp_valid_in && item < a.vp_btrue;
                    sp = a.x + ((size_t)(ok ? item : 0) * a.Cin + cl) * a.vp_tin_phys + (ok ? pos : 0);
                    st = (long long)CK * a.vp_tin_phys;
                }
                src[u] = ok ? sp : g_zero16;
                step_b[u] = ok ? st : 0;
            }
        }
        const unsigned lds0 = (unsigned)(size_t)smem;                    // LDS byte address of the ring
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // DMA instructions this wave really issues per chunk (the last one is skipped by waves whose 64 pieces all lie past the end)
        const int n_issue = (NU - 1) * C::NTHR + wave_u * 64 < C::DMA_NV ? NU : NU - 1;
        // pieces every wave issues in full (the first NFULL of the NU): one burst; the last, partly empty piece on its own
        constexpr int NFULL = C::DMA_NV / C::NTHR < NU ? C::DMA_NV / C::NTHR : NU;
        auto dma_chunk = [&](int stage) __attribute__((always_inline)) {
            // wave-instruction u fills 1 KiB at stage base + (u*NTHR + wave*64) * 16 bytes
            const unsigned dst0 = lds0 + (unsigned)(stage * C::DMA_STAGE_FLOATS * 4) + (unsigned)(wave_u * 64 * 16);
            if constexpr (NFULL > 0) lds_dma_burst<NFULL, C::NTHR * 16>(src, dst0);
#pragma unroll
            for (int u = NFULL; u < NU; ++u) {
                if (live[u]) lds_dma_burst<1, C::NTHR * 16>(src + u, dst0 + (unsigned)(u * C::NTHR * 16));
            }
#if !(defined(MVQ_EXP) && (MVQ_EXP & 64))               // timing build: pointers not advanced (same chunk re-read)
#pragma unroll
            for (int u = 0; u < NU; ++u) src[u] += step_b[u];
#endif
        };
        [[maybe_unused]] auto mfma_chunk_dma = [&](int stage, bool issue_next, int next_stage) __attribute__((always_inline)) {
            const float* wsrc = smem + stage * C::DMA_STAGE_FLOATS + a_base;
            const float* xs_same = smem + stage * C::DMA_STAGE_FLOATS + C::W_FLOATS + bd_same;
            const float* xs_cross = smem + stage * C::DMA_STAGE_FLOATS + C::W_FLOATS + bd_cross;
            float av[KPRE ? 2 : 1][KG][MT], bv[KPRE ? 2 : 1][KG][NT];
            auto load_group = [&](int g, int pb) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < KG; ++u) {
                    const int st = g * KG + u;
                    if (st < NS) {
                        const int k0 = 2 * st;
                        const int off0 = (k0 / KS) * XP + (k0 % KS) * DIL;
                        const bool cross = ((k0 + 1) / KS) != (k0 / KS);
                        const float* xsp = cross ? xs_cross : xs_same;
#pragma unroll
                        for (int i = 0; i < MT; ++i) av[pb][u][i] = wsrc[k0 * C::BM + i * 32];
#pragma unroll
                        for (int j = 0; j < NT; ++j) bv[pb][u][j] = xsp[off0 + j * 32 * STRIDE];
                    }
                }
            };
            load_group(0, 0);
            if (issue_next) dma_chunk(next_stage);         // behind the first operand reads: issued while those are in flight
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int pb = KPRE ? (g & 1) : 0;
                if (KPRE) { if (g + 1 < NG) load_group(g + 1, (g + 1) & 1); }
                else if (g > 0) load_group(g, 0);
#pragma unroll
                for (int u = 0; u < KG; ++u)
                    if (g * KG + u < NS) {
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pb][u][i], bv[pb][u][j], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT) * KG, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT * KG, 0);
            }
        };
#if MVQ_ASM_READS > 0
        using AL = AsmOperandLoop<KS, STRIDE, DIL, C::BM, XP, MT, NT, NS, MVQ_ASM_READS>;
        auto mfma_chunk_asm = [&](int stage, bool issue_next, int next_stage) __attribute__((always_inline)) {
            const unsigned sb = lds0 + (unsigned)(stage * C::DMA_STAGE_FLOATS * 4);
            const unsigned a_addr = sb + (unsigned)(a_base * 4);
            const unsigned b_s = sb + (unsigned)((C::W_FLOATS + bd_same) * 4), b_c = sb + (unsigned)((C::W_FLOATS + bd_cross) * 4);
            float av0[MVQ_ASM_READS][MT], bv0[MVQ_ASM_READS][NT], av1[MVQ_ASM_READS][MT], bv1[MVQ_ASM_READS][NT];
            AL::template load_group<0, 0>(av0, bv0, a_addr, b_s, b_c);
            if (issue_next) dma_chunk(next_stage);
            AL::template run_from<0>(acc, av0, bv0, av1, bv1, a_addr, b_s, b_c);
        };
#define MVQ_CHUNK mfma_chunk_asm
#else
#define MVQ_CHUNK mfma_chunk_dma
#endif
        dma_chunk(0);
        if (n_chunks > 1) dma_chunk(1);
        int st_c = 0, st_n2 = 2;                       // stage of chunk c / of chunk c+2
#ifdef MVQ_EXP      // TIMING EXPERIMENTS ONLY (wrong results): bit 0 = no DMA in the steady state, bit 1 = no wait / barrier per chunk
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
            if (!(MVQ_EXP & 2)) {
                if (c + 1 >= n_chunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (n_issue == NU) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NU) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NU - 1) : "memory");
                __syncthreads();
            }
            MVQ_CHUNK(c & 1, (MVQ_EXP & 1) ? false : (c + 2 < n_chunks), 2);
        }
#else
        for (int c = 0; c < n_chunks; ++c) {
            // chunk c has landed when at most the DMA_NU instructions of chunk c+1 are still outstanding
            if (c + 1 >= n_chunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (n_issue == NU) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NU - 1) : "memory");
            __syncthreads();                          // ... in every wave; and stage (c+2)%3 is free (read during chunk c-1)
            MVQ_CHUNK(st_c, c + 2 < n_chunks, st_n2);
            st_c = st_c == 2 ? 0 : st_c + 1;
            st_n2 = st_n2 == 2 ? 0 : st_n2 + 1;
        }
#endif
#undef MVQ_CHUNK
      }
    } else {
    tile.load_chunk(0, wreg, xv, xs);
    __syncthreads();                                  // alpha table visible
    tile.store_chunk(0, 0, wreg, xv, xs);
    __syncthreads();
    // steady state (no conditionals around the staging registers: loads of chunk c+1 stay in flight across the
    // MFMAs of chunk c), last chunk peeled
    for (int c = 0; c + 1 < n_chunks; ++c) {
        tile.load_chunk(c + 1, wreg, xv, xs);
        __builtin_amdgcn_sched_barrier(0);            // keep the loads ABOVE the MFMAs (the scheduler sinks them otherwise)
        mfma_chunk(c & 1);
        __builtin_amdgcn_sched_barrier(0);
        tile.store_chunk(c + 1, (c + 1) & 1, wreg, xv, xs);
        __syncthreads();
    }
    mfma_chunk((n_chunks - 1) & 1);
    }

    // ---------------------------------------------------------------- fused ResidualUnit tail (FUSE)
    // The block owns ALL channels of its time tile (BM == C), so the 1x1 conv of the ResidualUnit runs here:
    // h = snake_mid(acc + bias) goes to LDS once (channel-major tile, same layout as the epilogue tile) and is the
    // B operand of a second MFMA pass whose A operand (the C x C weights, L1/L2 resident) streams straight into
    // registers.  The intermediate activation never touches HBM and the skip input is re-read L2-hot.
    const float* ep_bias = a.bias;
    if (FUSE) {
        float* const Ht = smem;                      // [BM][BNP]
        __syncthreads();                              // staging buffers are free
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float bv = a.bias ? a.bias[row] : 0.0f;
                const float al = a.alpha_mid[row], inv = Ep[EP_MID * C::BM + row];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#if defined(MVQ_EXP) && (MVQ_EXP & 16)
                    Ht[row * C::BNP + (wn * NT + j) * 32 + l31] = acc[i][j][r] + bv + al * inv;
#else
                    Ht[row * C::BNP + (wn * NT + j) * 32 + l31] = det_snake(acc[i][j][r] + bv, al, inv);
#endif
                    acc[i][j][r] = 0.0f;
                }
            }
        __syncthreads();
        constexpr int K2 = C::BM / 2;                 // k-steps of the 1x1 conv (K = C = BM)
        constexpr int G = 8;                          // k-steps per register group of A operands
        static_assert(K2 % G == 0, "BM must be a multiple of 16");
        const float* w2 = a.w2p + (size_t)h * a.Mpad + wm * (MT * 32) + l31;
        const float* hsrc = Ht + h * C::BNP + wn * (NT * 32) + l31;
        float aw[2][G][MT];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < MT; ++i) aw[0][g][i] = w2[(size_t)(2 * g) * a.Mpad + i * 32];
#if defined(MVQ_EXP) && (MVQ_EXP & 32)
        for (int grp = 0; grp < 1; ++grp) {
#else
#pragma unroll
        for (int grp = 0; grp < K2 / G; ++grp) {
#endif
            if (grp + 1 < K2 / G) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        aw[(grp + 1) & 1][g][i] = w2[(size_t)(2 * ((grp + 1) * G + g)) * a.Mpad + i * 32];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s2 = grp * G + g;
                float bvv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bvv[j] = hsrc[2 * s2 * C::BNP + j * 32];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[grp & 1][g][i], bvv[j], acc[i][j], 0, 0, 0);
            }
        }
        ep_bias = a.bias2;
    }

    // ---------------------------------------------------------------- epilogue
    // The accumulators go through LDS once (the staging buffers are free now) so that bias / residual / Snake /
    // tanh and the global stores run row-contiguous: 16-byte residual loads and stores when rows are aligned,
    // 4-byte but fully coalesced otherwise.  (ConvTranspose phases whose count does not divide BM keep the
    // direct per-lane store.)
#if defined(MVQ_EXP) && (MVQ_EXP & 4)
    {   // TIMING EXPERIMENT: no epilogue (one store keeps the accumulators alive)
        float sink = 0.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
        if (sink == 123.456f) a.y[0] = sink;
        return;
    }
#endif
    const bool has_res = (UPS == 0) && a.residual != nullptr;
    const bool snake_out = a.alpha_out != nullptr;
    const bool do_tanh = a.act == 1;
    const bool do_gelu = a.act == 2;
    const bool direct = (UPS != 0) && (C::BM % (UPS ? UPS : 1) != 0);
    if (direct) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + (wm * MT + i) * 32 + row;
                const bool mok = m < a.Mrows;
                const int mc = mok ? m : a.Mrows - 1;
                const int co = mc / (UPS ? UPS : 1), rr = mc - co * (UPS ? UPS : 1);
                const float bv = ep_bias ? ep_bias[co] : 0.0f;
                const int lr = (wm * MT + i) * 32 + row;              // row of the block tile -> Ep
                const float al = snake_out ? a.alpha_out[co] : 1.0f, inv = Ep[lr];
                const size_t rowoff = ((size_t)b * a.Cout + co) * a.Tout;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int n = n0 + (wn * NT + j) * 32 + l31;
                    int t = n * (UPS ? UPS : 1) + rr - a.up_p;
                    bool ok = mok && n < a.Ncols && t >= 0 && (a.up_per_out ? true : t < a.Tout);
                    t = ok ? t : 0;
                    size_t ro = rowoff;
                    if (a.up_per_out) {                               // packed input rows -> unpacked output
                        const int sg = (int)__umulhi((unsigned)t, a.up_magic);
                        t -= sg * a.up_per_out;
                        const int bb = b * a.up_seg + sg;
                        ok = ok && t < a.up_valid_out && bb < a.up_btrue;
                        ro = ((size_t)(ok ? bb : 0) * a.Cout + co) * a.Tout;
                        t = ok ? t : 0;
                    }
                    float v = acc[i][j][r] + bv;
                    const bool tail = a.tvalid && t >= a.tvalid;
                    if (ok && a.y2) a.y2[ro + t] = tail ? 0.0f : det_snake(v, a.alpha2[co], Ep[C::BM + lr]);
                    if (snake_out) v = det_snake(v, al, inv);
                    if (ok) a.y[ro + t] = tail ? 0.0f : v;
                }
            }
        }
        return;
    }

    float* const Ct = smem;                          // [BM / EHP][BNP]
    constexpr int EHP = FUSE ? 1 : C::EH;            // epilogue passes
    constexpr int MTH = MT / EHP;                    // row groups per wave per pass
    constexpr int BMH = C::BM / EHP;
    auto tile_row_of = [&](int lrow, int pass) __attribute__((always_inline)) {
        if (EHP == 1) return lrow;
        const int g = lrow >> 5;
        return ((g / MTH) * MT + pass * MTH + (g % MTH)) * 32 + (lrow & 31);
    };
    // One epilogue pass, with the pass index a compile-time constant (a generic lambda instantiated per pass: an unroll pragma on
    // a pass LOOP is dropped once the body outgrows the unroller's size limit, and then `acc[hp * MTH + il]` is a dynamically
    // indexed register array, i.e. scratch, and none of the regular mapping's offsets are immediates any more).
    auto epilogue_pass = [&](auto HP) __attribute__((always_inline)) {
    constexpr int hp = decltype(HP)::value;
    __syncthreads();                                  // every wave is done with the staging buffers / the previous pass
#pragma unroll
    for (int il = 0; il < MTH; ++il)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * MTH + il) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int j = 0; j < NT; ++j) Ct[row * C::BNP + (wn * NT + j) * 32 + l31] = acc[hp * MTH + il][j][r];
        }
    // Skip-path prefetch: the residual quads this thread will add are requested NOW -- all of them at once, into the registers
    // the accumulators just left -- so their HBM latency runs under the barrier and is paid once per pass, not once per group of
    // four.  (Timing builds without the epilogue: the fused C = 64 unit 108 -> 120 TFLOP/s, C = 128 126 -> 135: most of that
    // was exposed latency of these loads, not arithmetic.)
    constexpr int NVQ = BMH * C::BN / 4;
#ifdef MVQ_NO_RES_PREFETCH                              // A/B builds
    constexpr bool PRE_RES = false;
#else
    constexpr bool PRE_RES = (UPS == 0) && (NVQ % C::NTHR == 0) && (NVQ / C::NTHR <= 16);
#endif
    constexpr int RES_IT = PRE_RES ? NVQ / C::NTHR : 1;
    // REGULAR quad mapping (ConvCfg::REG_GEOM): a thread keeps its JN column quads (c4 = q_c0 + j * CQ) and its rows advance by the
    // constant RSTEP per step, so iteration it = (row step it / JN, column group it % JN).  Everything that depends only on the
    // column (bounds, the zero-tail / packed-gap mask) is per-thread state, the row of an iteration is q_r0 + a compile-time
    // constant (LDS reads of the tile and of the row tables take immediate offsets), and global addresses are a uniform 64-bit
    // base per block + a 32-bit per-thread offset that advances by a scalar -- the quad's vector work is the arithmetic itself.
    // (Before: e / (BN/4), the tile-row shuffle, a 64-bit address and three global loads of bias / alpha per quad; vector
    // instructions are additive to fp32 MFMA time on this chip)  The row table holds ONE Snake alpha:
    // a launch with both a dual output and an output Snake (no layer of the model) takes the generic path.
    constexpr int CQ = C::CQ, JN = C::JN;
    constexpr bool REG = PRE_RES && C::REG_GEOM && (BMH % C::RSTEP == 0);
    constexpr int RSTEP = REG ? C::RSTEP : 1;
    const bool reg_ok = REG && !(a.y2 && a.alpha_out);
    const int q_c0 = REG ? tid % CQ : 0, q_r0 = REG ? tid / CQ : 0;
    // row of the block tile that iteration `it` of the regular mapping handles, minus q_r0: a constant after unrolling
    auto reg_k = [&](int it, int pass) __attribute__((always_inline)) {
        const int lr = (it / JN) * RSTEP;                              // local row (of this pass) of the thread with q_r0 == 0
        if (EHP == 1) return lr;
        const int g = lr >> 5;
        return ((g / MTH) * MT + pass * MTH + (g % MTH)) * 32 + (lr & 31);
    };
    // uniform base: element (b, m0, n0) -- or, for virtually packed rows, row m0 of the block's first item (the column part of
    // the address then comes from the per-thread (item, position) of each column quad)
    const size_t blk_off = a.vp_seg ? ((size_t)b * a.vp_seg * a.Cout + m0) * a.Tout : ((size_t)b * a.Cout + m0) * a.Tout + n0;
    const unsigned q_t0 = (unsigned)(q_r0 * a.Tout);
    bool q_nok[JN];
    unsigned q_col[JN];                                               // column part of this thread's offsets, per column group
    int q_vpos[JN];                                                   // virtually packed: position inside the item's row
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int n = n0 + 4 * (q_c0 + j * CQ);
        q_nok[j] = n < a.Ncols;
        q_col[j] = 4u * (q_c0 + j * CQ);
        q_vpos[j] = 0;
        if (a.vp_seg) {
            const int nn = q_nok[j] ? n : 0;
            const int seg = (int)__umulhi((unsigned)nn, a.vp_magic_out), pos = nn - seg * a.vp_per_out;
            q_nok[j] = q_nok[j] && b * a.vp_seg + seg < a.vp_btrue && pos < a.Tout;
            q_col[j] = (unsigned)(seg * a.Cout * a.Tout + pos);
            q_vpos[j] = pos;
        }
    }
    f32x4 res_q[RES_IT];
    if (PRE_RES && has_res && a.ovec4) {
        bool pre_done = false;
        if constexpr (REG) if (reg_ok) {
            pre_done = true;
            const float* resb = a.residual + blk_off;
#pragma unroll
            for (int it = 0; it < RES_IT; ++it) {
                const int k = reg_k(it, hp), j = it % JN;
                const bool ok = q_nok[j] && m0 + q_r0 + k < a.Mrows;
                res_q[it] = *reinterpret_cast<const f32x4*>(resb + (ok ? q_t0 + (unsigned)(k * a.Tout) + q_col[j] : 0u));
            }
        }
        if (!pre_done) {
#pragma unroll
            for (int it = 0; it < RES_IT; ++it) {
                const int e = tid + it * C::NTHR;
                const int row = e / (C::BN / 4);
                const int c4 = e - row * (C::BN / 4);
                const int m = m0 + tile_row_of(row, hp), n = n0 + 4 * c4;
                const bool ok = m < a.Mrows && n < a.Ncols;
                const size_t off = ((size_t)b * a.Cout + (ok ? m : 0)) * a.Tout + (ok ? n : 0);
                res_q[it] = *reinterpret_cast<const f32x4*>(a.residual + off);
            }
        }
    }
    // epilogue row table (bias | alpha of the one epilogue Snake, for this block's rows) behind the tile: one fill per block
    float* const Tb = smem + (FUSE ? C::CT_FLOATS : C::CTH_FLOATS);
    if (reg_ok && hp == 0) {
        const float* const ax = a.y2 ? a.alpha2 : a.alpha_out;
        for (int r = tid; r < C::BM; r += C::NTHR) {
            int m = m0 + r;
            m = m < a.Mrows ? m : a.Mrows - 1;
            Tb[r] = ep_bias ? ep_bias[m] : 0.0f;
            Tb[C::BM + r] = ax ? ax[m] : 1.0f;
        }
    }
    __syncthreads();
    // local row -> row of the block tile: group g = lrow / 32 belongs to wave row g / MTH, its (hp*MTH + g % MTH)-th group
    auto tile_row = [&](int lrow) __attribute__((always_inline)) { return tile_row_of(lrow, hp); };

    if (UPS == 0) {
        if (a.ovec4) {
            constexpr int NV = BMH * C::BN / 4;
            auto quad = [&](int e, int it) __attribute__((always_inline)) {
                const int row = e / (C::BN / 4);
                const int c4 = e - row * (C::BN / 4);
                const int m = m0 + tile_row(row), n = n0 + 4 * c4;
                if (m < a.Mrows && n < a.Ncols) {
                    const float bv = ep_bias ? ep_bias[m] : 0.0f;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Ct + row * C::BNP + 4 * c4);
                    const size_t off = ((size_t)b * a.Cout + m) * a.Tout + n;
                    v.x = v.x + bv; v.y = v.y + bv; v.z = v.z + bv; v.w = v.w + bv;
                    if (a.dsn_src) {
                        const float ad = a.dsn_alpha[m], id = Ep[2 * C::BM + tile_row(row)];
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(a.dsn_src + off);
                        v.x = v.x * det_dsnake(sv.x, ad, id); v.y = v.y * det_dsnake(sv.y, ad, id);
                        v.z = v.z * det_dsnake(sv.z, ad, id); v.w = v.w * det_dsnake(sv.w, ad, id);
                    }
                    if (has_res) {
                        const f32x4 rv = PRE_RES ? res_q[PRE_RES ? it : 0] : *reinterpret_cast<const f32x4*>(a.residual + off);
                        v.x = v.x + rv.x; v.y = v.y + rv.y; v.z = v.z + rv.z; v.w = v.w + rv.w;
                    }
                    int nz = (a.tvalid && n + 4 > a.tvalid) ? n + 4 - a.tvalid : 0;           // trailing pad columns of this quad
                    if (a.tper) {                                     // packed rows: the gap columns of every period (quads never straddle one)
                        const int nn = n - (int)__umulhi((unsigned)n, a.tper_magic) * a.tper;
                        nz = nn + 4 > a.tper_valid ? (nn >= a.tper_valid ? 4 : nn + 4 - a.tper_valid) : 0;
                    }
                    if (a.y2) {
                        const float a2 = a.alpha2[m], i2 = Ep[C::BM + tile_row(row)];
                        f32x4 w = {det_snake(v.x, a2, i2), det_snake(v.y, a2, i2), det_snake(v.z, a2, i2), det_snake(v.w, a2, i2)};
                        if (nz > 0) { w.w = 0.0f; if (nz > 1) w.z = 0.0f; if (nz > 2) w.y = 0.0f; if (nz > 3) w.x = 0.0f; }
                        *reinterpret_cast<f32x4*>(a.y2 + off) = w;
                    }
                    if (snake_out) {
                        const float al = a.alpha_out[m], inv = Ep[tile_row(row)];
                        v.x = det_snake(v.x, al, inv); v.y = det_snake(v.y, al, inv);
                        v.z = det_snake(v.z, al, inv); v.w = det_snake(v.w, al, inv);
                    }
                    if (do_tanh) { v.x = det_tanh(v.x); v.y = det_tanh(v.y); v.z = det_tanh(v.z); v.w = det_tanh(v.w); }
                    if (do_gelu) { v.x = det_gelu(v.x); v.y = det_gelu(v.y); v.z = det_gelu(v.z); v.w = det_gelu(v.w); }
                    if (nz > 0) { v.w = 0.0f; if (nz > 1) v.z = 0.0f; if (nz > 2) v.y = 0.0f; if (nz > 3) v.x = 0.0f; }
                    *reinterpret_cast<f32x4*>(a.y + off) = v;
                }
            };
            bool quads_done = false;
            if constexpr (REG) if (reg_ok) {
                quads_done = true;
                int nz[JN];                                           // trailing pad / packed-gap columns of this thread's quads
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    const int n = n0 + 4 * (q_c0 + j * CQ);
                    nz[j] = (a.tvalid && n + 4 > a.tvalid) ? n + 4 - a.tvalid : 0;
                    if (a.tper) {                                     // packed rows: the gap columns of every period
                        const int nn = n - (int)__umulhi((unsigned)n, a.tper_magic) * a.tper;
                        nz[j] = nn + 4 > a.tper_valid ? (nn >= a.tper_valid ? 4 : nn + 4 - a.tper_valid) : 0;
                    }
                    if (a.vp_seg) {                                   // virtually packed rows: the zero tail of the item's own row
                        const int pp = q_vpos[j];
                        nz[j] = pp + 4 > a.vp_valid_out ? (pp >= a.vp_valid_out ? 4 : pp + 4 - a.vp_valid_out) : 0;
                    }
                }
                float* const yb = a.y + blk_off;
                float* const y2b = a.y2 ? a.y2 + blk_off : nullptr;
                const float* const dsb = a.dsn_src ? a.dsn_src + blk_off : nullptr;
                const float* const ctq = Ct + q_r0 * C::BNP + 4 * q_c0;
                const float* const tbq = Tb + q_r0;
                const float* const epq = Ep + q_r0;
                // The quad loop, instantiated per COMBINATION of the uniform epilogue options (MODE): inside a mode every option is a
                // compile-time constant, so a quad carries no scalar compare / branch per option (round 4, SQ counters: 760 scalar
                // and 80 vector instructions per quad-row of the C = 192 1x1 before; floor 52 vector).  MODE 0 keeps them all at
                // run time (backward / tanh / GELU epilogues and anything not listed).
                auto reg_quads = [&](auto MODE_) __attribute__((always_inline)) {
                    constexpr int MODE = decltype(MODE_)::value;
                    // MODE: 1 skip + dual output | 2 skip + output Snake | 3 output Snake | 4 dual output | 5 skip only | 6 bias only
                    const bool f_dsn = MODE == 0 ? (a.dsn_src != nullptr) : false;
                    const bool f_res = MODE == 0 ? has_res : (MODE == 1 || MODE == 2 || MODE == 5);
                    const bool f_y2 = MODE == 0 ? (a.y2 != nullptr) : (MODE == 1 || MODE == 4);
                    const bool f_so = MODE == 0 ? snake_out : (MODE == 2 || MODE == 3);
                    const bool f_tanh = MODE == 0 ? do_tanh : false;
                    const bool f_gelu = MODE == 0 ? do_gelu : false;
#pragma unroll
                    for (int it = 0; it < RES_IT; ++it) {
                        const int k = reg_k(it, hp), j = it % JN;          // compile-time after unrolling
                        const int trow = q_r0 + k;
                        if (q_nok[j] && m0 + trow < a.Mrows) {
                            const unsigned toff = q_t0 + (unsigned)(k * a.Tout) + q_col[j];
                            const float bv = tbq[k];
                            f32x4 v = *reinterpret_cast<const f32x4*>(ctq + (it / JN) * RSTEP * C::BNP + 4 * CQ * j);
                            v.x = v.x + bv; v.y = v.y + bv; v.z = v.z + bv; v.w = v.w + bv;
                            if (f_dsn) {
                                const float ad = a.dsn_alpha[m0 + trow], id = epq[2 * C::BM + k];
                                const f32x4 sv = *reinterpret_cast<const f32x4*>(dsb + toff);
                                v.x = v.x * det_dsnake(sv.x, ad, id); v.y = v.y * det_dsnake(sv.y, ad, id);
                                v.z = v.z * det_dsnake(sv.z, ad, id); v.w = v.w * det_dsnake(sv.w, ad, id);
                            }
                            if (f_res) {
                                const f32x4 rv = res_q[it];
                                v.x = v.x + rv.x; v.y = v.y + rv.y; v.z = v.z + rv.z; v.w = v.w + rv.w;
                            }
                            const int z = nz[j];
                            if (f_y2) {
                                const float a2 = tbq[C::BM + k], i2 = epq[C::BM + k];
                                f32x4 w = {det_snake(v.x, a2, i2), det_snake(v.y, a2, i2), det_snake(v.z, a2, i2), det_snake(v.w, a2, i2)};
                                if (z > 0) { w.w = 0.0f; if (z > 1) w.z = 0.0f; if (z > 2) w.y = 0.0f; if (z > 3) w.x = 0.0f; }
                                *reinterpret_cast<f32x4*>(y2b + toff) = w;
                            }
                            if (f_so) {
                                const float al = tbq[C::BM + k], inv = epq[k];
                                v.x = det_snake(v.x, al, inv); v.y = det_snake(v.y, al, inv);
                                v.z = det_snake(v.z, al, inv); v.w = det_snake(v.w, al, inv);
                            }
                            if (f_tanh) { v.x = det_tanh(v.x); v.y = det_tanh(v.y); v.z = det_tanh(v.z); v.w = det_tanh(v.w); }
                            if (f_gelu) { v.x = det_gelu(v.x); v.y = det_gelu(v.y); v.z = det_gelu(v.z); v.w = det_gelu(v.w); }
                            if (z > 0) { v.w = 0.0f; if (z > 1) v.z = 0.0f; if (z > 2) v.y = 0.0f; if (z > 3) v.x = 0.0f; }
                            *reinterpret_cast<f32x4*>(yb + toff) = v;
                        }
                    }
                };
                const bool plain = !a.dsn_src && !do_tanh && !do_gelu;
                const bool hy2 = a.y2 != nullptr;
                if (!plain) reg_quads(std::integral_constant<int, 0>{});
                else if (has_res && hy2 && !snake_out) reg_quads(std::integral_constant<int, 1>{});
                else if (has_res && !hy2 && snake_out) reg_quads(std::integral_constant<int, 2>{});
                else if (!has_res && !hy2 && snake_out) reg_quads(std::integral_constant<int, 3>{});
                else if (!has_res && hy2 && !snake_out) reg_quads(std::integral_constant<int, 4>{});
                else if (has_res && !hy2 && !snake_out) reg_quads(std::integral_constant<int, 5>{});
                else if (!has_res && !hy2 && !snake_out) reg_quads(std::integral_constant<int, 6>{});
                else reg_quads(std::integral_constant<int, 0>{});
            }
            if (quads_done) {
            } else if (PRE_RES) {
#pragma unroll
                for (int it = 0; it < RES_IT; ++it) quad(tid + it * C::NTHR, it);
            } else {
#pragma unroll 4
                for (int e = tid; e < NV; e += C::NTHR) quad(e, 0);
            }
        } else {
            constexpr int NE = BMH * C::BN;
#pragma unroll 4
            for (int e = tid; e < NE; e += C::NTHR) {
                const int row = e / C::BN;
                const int col = e - row * C::BN;
                const int m = m0 + tile_row(row), n = n0 + col;
                if (m < a.Mrows && n < a.Ncols) {
                    const size_t off = ((size_t)b * a.Cout + m) * a.Tout + n;
                    float v = Ct[row * C::BNP + col] + (ep_bias ? ep_bias[m] : 0.0f);
                    const int lr = tile_row(row);
                    if (a.dsn_src) v = v * det_dsnake(a.dsn_src[off], a.dsn_alpha[m], Ep[2 * C::BM + lr]);
                    if (has_res) v = v + a.residual[off];
                    bool tail = a.tvalid && n >= a.tvalid;
                    if (a.tper) tail = n - (int)__umulhi((unsigned)n, a.tper_magic) * a.tper >= a.tper_valid;
                    if (a.y2) a.y2[off] = tail ? 0.0f : det_snake(v, a.alpha2[m], Ep[C::BM + lr]);
                    if (snake_out) v = det_snake(v, a.alpha_out[m], Ep[lr]);
                    if (do_tanh) v = det_tanh(v);
                    if (do_gelu) v = det_gelu(v);
                    a.y[off] = tail ? 0.0f : v;
                }
            }
        }
    } else {
        // pixel shuffle: for each output channel of this tile the S phases interleave into one contiguous run
        // (a pass holds whole 32-row groups, i.e. 32/S whole channels each, so the two-pass form applies unchanged)
        constexpr int S = UPS ? UPS : 1;
        const int run = C::BN * S;                    // output samples per channel covered by this tile
        const int nco = BMH / S;
        const int t_base = n0 * S - a.up_p;
        const int total = nco * run;
        // One output element: (local channel col, sample tl of the channel's run) -> LDS tile [phase row][column], Snake, store.
        auto shuffle_one = [&](int col, int tl) __attribute__((always_inline)) {
            const int nl = tl / S, rr = tl - nl * S;
            const int co = (m0 + tile_row(col * S)) / S;
            int t = t_base + tl;
            bool ok = co < a.Cout && n0 + nl < a.Ncols && t >= 0 && (a.up_per_out ? true : t < a.Tout);
            int bb = b;
            if (ok && a.up_per_out) {                                 // packed input rows -> unpacked output
                const int sg = (int)__umulhi((unsigned)t, a.up_magic);
                t -= sg * a.up_per_out;
                bb = b * a.up_seg + sg;
                ok = t < a.up_valid_out && bb < a.up_btrue;
            }
            if (ok) {
                float v = Ct[(col * S + rr) * C::BNP + nl] + (ep_bias ? ep_bias[co] : 0.0f);
                const size_t off = ((size_t)bb * a.Cout + co) * a.Tout + t;
                const bool tail = a.tvalid && t >= a.tvalid;
                const int lr = tile_row(col * S);                     // any phase row of this channel: same table entry
                if (a.y2) a.y2[off] = tail ? 0.0f : det_snake(v, a.alpha2[co], Ep[C::BM + lr]);
                if (snake_out) v = det_snake(v, a.alpha_out[co], Ep[lr]);
                a.y[off] = tail ? 0.0f : v;
            }
        };
        // 16-byte form (round 4): with 16-byte output rows a channel's run splits into `lead` head samples (they complete the
        // previous tile's last quad in memory), QF aligned quads and a tail; a thread takes a whole quad -- four LDS reads at
        // compile-time (phase, column) patterns, ONE bias / alpha / reciprocal fetch, one 16-byte store per output -- and the
        // few head / tail samples go one by one.  (Before: every sample paid its own index arithmetic, its own global loads of
        // bias and alpha and a 4-byte store.)
        const bool up_quads = a.ovec4 && (C::BN * S) % 4 == 0 && (a.up_per_out == 0 || (a.up_per_out % 4 == 0 && a.up_valid_out % 4 == 0));
        if (up_quads) {
            const int lead = (4 - (t_base & 3)) & 3;                  // samples in front of the first 16-byte boundary
            const int QF = (run - lead) / 4;
            const int totq = nco * QF;
            for (int e = tid; e < totq; e += C::NTHR) {
                const int col = e / QF;
                const int tl0 = lead + 4 * (e - col * QF);
                const int co = (m0 + tile_row(col * S)) / S;
                int t = t_base + tl0;                                 // multiple of 4
                const int nl_last = (tl0 + 3) / S;
                bool ok = co < a.Cout && n0 + nl_last < a.Ncols && t >= 0 && (a.up_per_out ? true : t + 3 < a.Tout);
                int bb = b;
                if (ok && a.up_per_out) {                             // quads never straddle a segment (period, valid % 4 == 0)
                    const int sg = (int)__umulhi((unsigned)t, a.up_magic);
                    t -= sg * a.up_per_out;
                    bb = b * a.up_seg + sg;
                    ok = t < a.up_valid_out && bb < a.up_btrue;
                } else if (!ok && co < a.Cout) {                      // a quad cut by the end of the row: sample by sample
#pragma unroll
                    for (int i = 0; i < 4; ++i) shuffle_one(col, tl0 + i);
                    continue;
                }
                if (!ok) continue;
                const float bv = ep_bias ? ep_bias[co] : 0.0f;
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = tl0 + i, nl = tl / S, rr = tl - nl * S;
                    v[i] = Ct[(col * S + rr) * C::BNP + nl] + bv;
                }
                const size_t off = ((size_t)bb * a.Cout + co) * a.Tout + t;
                const int lr = tile_row(col * S);
                int nz = (a.tvalid && t + 4 > a.tvalid) ? t + 4 - a.tvalid : 0;
                nz = nz > 4 ? 4 : nz;
                if (a.y2) {
                    const float a2 = a.alpha2[co], i2 = Ep[C::BM + lr];
                    f32x4 w = {det_snake(v.x, a2, i2), det_snake(v.y, a2, i2), det_snake(v.z, a2, i2), det_snake(v.w, a2, i2)};
                    if (nz > 0) { w.w = 0.0f; if (nz > 1) w.z = 0.0f; if (nz > 2) w.y = 0.0f; if (nz > 3) w.x = 0.0f; }
                    *reinterpret_cast<f32x4*>(a.y2 + off) = w;
                }
                if (snake_out) {
                    const float al = a.alpha_out[co], inv = Ep[lr];
                    v.x = det_snake(v.x, al, inv); v.y = det_snake(v.y, al, inv); v.z = det_snake(v.z, al, inv); v.w = det_snake(v.w, al, inv);
                }
                if (nz > 0) { v.w = 0.0f; if (nz > 1) v.z = 0.0f; if (nz > 2) v.y = 0.0f; if (nz > 3) v.x = 0.0f; }
                *reinterpret_cast<f32x4*>(a.y + off) = v;
            }
            const int rest = run - 4 * QF;                            // head + tail samples of every channel
            for (int e = tid; e < nco * rest; e += C::NTHR) {
                const int col = e / rest, r = e - col * rest;
                shuffle_one(col, r < lead ? r : 4 * QF + r);
            }
        } else
        for (int e = tid; e < total; e += C::NTHR) {
            const int col = e / run;
            const int tl = e - col * run;
            const int nl = tl / S, rr = tl - nl * S;
            const int co = (m0 + tile_row(col * S)) / S;
            int t = t_base + tl;
            bool ok = co < a.Cout && n0 + nl < a.Ncols && t >= 0 && (a.up_per_out ? true : t < a.Tout);
            int bb = b;
            if (ok && a.up_per_out) {                                 // packed input rows -> unpacked output
                const int sg = (int)__umulhi((unsigned)t, a.up_magic);
                t -= sg * a.up_per_out;
                bb = b * a.up_seg + sg;
                ok = t < a.up_valid_out && bb < a.up_btrue;
            }
            if (ok) {
                float v = Ct[(col * S + rr) * C::BNP + nl] + (ep_bias ? ep_bias[co] : 0.0f);
                const size_t off = ((size_t)bb * a.Cout + co) * a.Tout + t;
                const bool tail = a.tvalid && t >= a.tvalid;
                const int lr = tile_row(col * S);                     // any phase row of this channel: same table entry
                if (a.y2) a.y2[off] = tail ? 0.0f : det_snake(v, a.alpha2[co], Ep[C::BM + lr]);
                if (snake_out) v = det_snake(v, a.alpha_out[co], Ep[lr]);
                a.y[off] = tail ? 0.0f : v;
            }
        }
    }
    };   // epilogue pass
    epilogue_pass(std::integral_constant<int, 0>{});
    if constexpr (EHP == 2) epilogue_pass(std::integral_constant<int, 1>{});
    static_assert(EHP == 1 || EHP == 2, "one or two epilogue passes");
}

// 16-byte input rows (Tin % 4 == 0) take the float4 staging path; the two paths are separate loop nests so that
// no control-flow merge sits between a chunk's global loads and the MFMAs that hide them.
template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, int UPS>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 64 * WAVES_M * WAVES_N),
                          amdgpu_waves_per_eu(ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS>::MIN_WPE)))
void conv1d_mfma_kernel(const ConvArgs a)
{
    if (a.vec4) conv1d_mfma_body<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS, true, false>(a);
    else conv1d_mfma_body<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS, false, false>(a);
}

// whole ResidualUnit (7-tap dilated conv + Snake + 1x1 conv + skip) for C == BM
template <int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 64 * WAVES_M * WAVES_N),
                          amdgpu_waves_per_eu(ConvCfg<7, 1, DIL, CK, MT, NT, WAVES_M, WAVES_N, 0>::MIN_WPE_FUSE)))
void residual_unit_kernel(const ConvArgs a)
{
    if (a.vec4) conv1d_mfma_body<7, 1, DIL, CK, MT, NT, WAVES_M, WAVES_N, 0, true, true>(a);
    else conv1d_mfma_body<7, 1, DIL, CK, MT, NT, WAVES_M, WAVES_N, 0, false, true>(a);
}

template <int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N>
inline hipError_t launch_residual_unit(const ConvArgs& a_in, hipStream_t stream)
{
    using C = ConvCfg<7, 1, DIL, CK, MT, NT, WAVES_M, WAVES_N, 0>;
    ConvArgs a = a_in;
    if (a.Cin % CK != 0 || a.Cout != C::BM || a.Cin != C::BM || a.Mpad != C::BM) return hipErrorInvalidValue;
    if (a.name_out) {
        snprintf(a.name_out, a.name_len, "residual_unit_kernel<%d, %d, %d, %d, %d, %d>", DIL, CK, MT, NT, WAVES_M, WAVES_N);
        return hipSuccess;
    }
    a.n_tiles = (a.Ncols + C::BN - 1) / C::BN;
    a.vec4 = (a.Tin % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    a.ovec4 = (a.Tout % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
              ((reinterpret_cast<uintptr_t>(a.residual) & 15) == 0);
    // input already carries its Snake (the producer's dual output) and rows are 16-byte: the 7-tap stage runs on the LDS-DMA ring
    a.dma = (conv_dma_rows_ok(a) && ((size_t)C::LDS_FLOATS_FUSE_DMA + 3 * C::BM) * 4 * 2 <= 160 * 1024) ? 1 : 0;   // == DMA_FITS of the body
    const size_t lds = a.dma ? (size_t)C::LDS_FLOATS_FUSE_DMA * 4 + (size_t)3 * C::BM * 4
                             : (size_t)C::LDS_FLOATS_FUSE * 4 + (size_t)3 * C::BM * 4 + (size_t)2 * a.Cin * 4;
    auto kern = residual_unit_kernel<DIL, CK, MT, NT, WAVES_M, WAVES_N>;
    {
        static BigLdsOptIn opt;                       // per (kernel instantiation, device)
        const hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern));
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)(a.n_tiles * a.B), 1);
    int pi = -1;
    if (prof_enabled()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "residual_unit_kernel<%d, %d, %d, %d, %d, %d>", DIL, CK, MT, NT, WAVES_M, WAVES_N);
        const int cols = a.tvalid > 0 ? a.tvalid : a.Ncols;
        pi = prof_begin(nm, 2.0 * a.Cin * a.Cout * 8.0 * cols * a.B, stream);
    }
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), lds, stream, a);
    prof_end(pi, stream);
    return hipGetLastError();
}

template <int KS, int STRIDE, int DIL, int CK, int MT, int NT, int WAVES_M, int WAVES_N, int UPS>
inline hipError_t launch_conv1d_mfma(const ConvArgs& a_in, hipStream_t stream)
{
    using C = ConvCfg<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS>;
    ConvArgs a = a_in;
    if (a.Cin % CK != 0 || a.Mpad % C::BM != 0) return hipErrorInvalidValue;
    if (C::XALIGNED && (a.pad != 0 || a.n_base % 4 != 0)) return hipErrorInvalidValue;   // the 1x1 tiling assumes 16-byte tile starts
    if (a.name_out) {
        snprintf(a.name_out, a.name_len, "conv1d_mfma_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", KS, STRIDE, DIL, CK, MT, NT,
                 WAVES_M, WAVES_N, UPS);
        return hipSuccess;
    }
    a.n_tiles = (a.Ncols - a.n_base + C::BN - 1) / C::BN;
    if (a.n_tiles_max > 0 && a.n_tiles > a.n_tiles_max) a.n_tiles = a.n_tiles_max;
    a.vec4 = (a.Tin % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    a.ovec4 = (a.Tout % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
              (!a.residual || (reinterpret_cast<uintptr_t>(a.residual) & 15) == 0) &&
              (!a.y2 || (reinterpret_cast<uintptr_t>(a.y2) & 15) == 0) &&
              (!a.dsn_src || (reinterpret_cast<uintptr_t>(a.dsn_src) & 15) == 0);
    // LDS-DMA staging whenever the rows allow it and the 3-stage ring leaves room for at least two blocks per CU
    a.dma = (conv_dma_rows_ok(a) && (size_t)C::LDS_FLOATS_DMA * 4 <= 64 * 1024) ? 1 : 0;
    const size_t lds = (a.dma ? (size_t)C::LDS_FLOATS_DMA * 4 : (size_t)C::LDS_FLOATS * 4 + (a.alpha_in ? (size_t)2 * a.Cin * 4 : 0)) + (size_t)3 * C::BM * 4;
    auto kern = conv1d_mfma_kernel<KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS>;
    {
        static BigLdsOptIn opt;                       // per (kernel instantiation, device)
        const hipError_t e = opt.ensure(reinterpret_cast<const void*>(kern));
        if (e != hipSuccess) return e;
    }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.vp_seg) {   // virtually packed rows exist in the LDS-DMA loop and the regular 16-byte epilogue only
        constexpr int EHP_ = C::EH, NVQ_ = (C::BM / EHP_) * C::BN / 4;
        constexpr bool REG_ = C::REG_GEOM && (NVQ_ % C::NTHR == 0) && (NVQ_ / C::NTHR <= 16) && ((C::BM / EHP_) % C::RSTEP == 0);
        if (!REG_ || UPS != 0 || !a.dma || !a.ovec4 || (a.y2 && a.alpha_out)) return hipErrorInvalidValue;
    }
    const unsigned gx = (unsigned)(a.n_tiles * a.B), R = (unsigned)((a.Mrows + C::BM - 1) / C::BM);
    dim3 grid(gx, R);
    // Row tiles share the input tile, column tiles share the weight rows.  Default order (row tile slowest) keeps ONE
    // weight row tile hot in L2 while x streams -- right when the weight image is larger than an XCD's 4 MB L2 (k7, big C).
    // When the whole packed weight fits L2 comfortably (k = 1 convs), put the R row tiles of a column tile back to back
    // on one XCD instead: x is then read from HBM once instead of R times.
    a.row_fast = 0;
    static const size_t row_fast_max = [] {                        // MVQ_ROWFAST_MAX_KB: A/B override of the 2.5 MB threshold
        const char* e = getenv("MVQ_ROWFAST_MAX_KB");
        if (e) note_env_override(MVQ_BF_ENV_ROWFAST);
        return e ? (size_t)atol(e) * 1024 : ((size_t)5 << 19);
    }();
    if (R > 1 && (size_t)a.Cin * KS * a.Mpad * sizeof(float) <= row_fast_max) {               // <= 2.5 MB
        a.row_fast = (int)R;
        grid = dim3(((gx + 7) / 8) * 8 * R, 1);
    }
    int pi = -1;
    if (prof_enabled()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv1d_mfma_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", KS, STRIDE, DIL, CK, MT, NT, WAVES_M, WAVES_N, UPS);
        // algorithmic FLOPs of THIS launch: its share of the true GEMM columns (zero-padded tails and, for the polyphase
        // ConvTranspose1d, the extra boundary column do not count), every valid row, the whole K
        int last = a.n_base + a.n_tiles * C::BN;
        int lim = a.Ncols;
        if (UPS > 0 && !a.up_per_out) lim = a.Ncols - 1;              // Ncols = Tin + 1 GEMM columns for Tin input samples
        else if (a.tvalid > 0) lim = a.tvalid;
        if (last > lim) last = lim;
        double cols = last > a.n_base ? last - a.n_base : 0;
        // packed rows: only the data columns of every period count
        if (a.tper) cols *= (double)a.tper_valid / a.tper;
        if (a.vp_seg) cols *= ((double)a.vp_valid_out / a.vp_per_out) * ((double)a.vp_btrue / ((double)a.B * a.vp_seg));
        if (a.up_per_out) cols *= (double)a.up_valid_out / a.up_per_out;
        pi = prof_begin(nm, 2.0 * a.Cin * KS * a.Mrows * cols * a.B, stream);
    }
    hipLaunchKernelGGL(kern, grid, dim3(C::NTHR), lds, stream, a);
    prof_end(pi, stream);
    return hipGetLastError();
}

// LDS-DMA staging (conv1d_mfma_body) needs 16-byte rows and no Snake on load (the DMA cannot transform what it copies).
// Dispatchers use this to pick the instantiation whose 3-stage ring fits three blocks per CU (smaller CK where needed).
// MVQ_NO_DMA=1 in the environment switches it off (A/B measurements).
inline bool conv_dma_rows_ok(const ConvArgs& a)
{
    static const bool off = [] { const bool o = getenv("MVQ_NO_DMA") != nullptr; if (o) note_env_override(MVQ_BF_ENV_NO_DMA); return o; }();
    return !off && !a.alpha_in && a.Tin % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
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

// Column split.  A row whose length is not a multiple of the 128-column tile ends in a tile that is mostly padding (T = 600:
// 88 of 128 columns, T = 3000: 56).  Skipping the dead MFMAs inside that tile buys nothing -- the block still lasts as long as
// its busiest wave -- so the row is split instead: the full tiles go out as one launch, the remainder as a second launch of
// a narrower tile of the SAME kernel family (128 x 96 or 128 x 64: every wave of the block again has live columns only).
// Returns the tail tile width (96 / 64) or 0 when the row is not split (no remainder, remainder > 96, or a saving < 1 %).
inline int conv_tail_width(const ConvArgs& a)
{
    if (a.n_base != 0 || a.n_tiles_max != 0) return 0;          // already one half of a split
    // The tail is its own launch behind the main one: it pays off only when it can put a block on (most of) the 256 CUs.  At the
    // reference's batch of 6 a tail launch is 18-36 blocks that each walk the whole K chain alone -- 9 such launches were 2.5 of
    // the 14.5 ms of a 6-segment step (profiles/r05_kernel_stats_B6_inference_before.csv: 4.4 TFLOP/s) -- while the unsplit grid
    // still fits the chip's resident slots, where a mostly empty last tile costs nothing extra.
    if ((long)a.B * ((a.Mrows + 127) / 128) < 128) return 0;
    const int full = a.Ncols / 128, rem = a.Ncols % 128;
    if (full == 0 || rem == 0 || rem > 96) return 0;
    const int w = rem <= 64 ? 64 : 96;
    return (128 - w) * 100 >= (full + 1) * 128 ? w : 0;
}

// Latency regime: when the 128-row tiling would leave most of the 256 CUs idle (small batch x short sequences), the
// same kernel runs with 64 x 64 tiles -- 4x the blocks, each walking the same K chain, so results are unchanged.
inline bool conv_underfilled(const ConvArgs& a)
{
    return (long)a.B * ((a.Ncols + 127) / 128) * ((a.Mrows + 127) / 128) < 160;
}
// 64 x 64 tiles instead of 128 x 128 (96): a question of how the grid divides over the 256 CUs.  Below 160 big tiles the launch
// cannot put a block on most CUs at all.  Above it the big tiles run in ceil(big / 256) rounds of which the last may be nearly
// empty -- at the reference's batch of six the 256-channel encoder level is 288 tiles (two rounds for 1.1 rounds of work: 84 of the
// 157 TFLOP/s, profiles/r05_*B6*), the 768-channel decoder level 180 -- while four times as many quarter tiles fill their last
// round, at ~0.85 of the big tile's per-block rate (less operand reuse).  The form with the better fill x rate wins, up to 600
// big tiles; beyond that the big one always.  Measured at six segments (gpurun_out/g4): 11.60 -> 11.04 ms per step.
// MVQ_SMALL_TILE_MAX=n replaces the rule by `big < n` (A/B runs).
inline bool conv_prefer_small_tiles(const ConvArgs& a)
{
    if (a.Mpad % 64 != 0) return false;
    if (a.vp_seg) return false;                       // virtually packed rows: the LDS-DMA 128-row tiles only
    const long big = (long)a.B * ((a.Ncols + 127) / 128) * ((a.Mrows + 127) / 128);
    static const long cap = [] {
        const char* e = getenv("MVQ_SMALL_TILE_MAX");               // A/B knob (reported by mvq_build_flags)
        if (e) note_env_override(MVQ_BF_ENV_SMALL_TILES);
        return e ? atol(e) : -1L;
    }();
    if (cap >= 0) return big < cap;
    if (big < 160) return true;
    if (big >= 600) return false;                     // 2.3 rounds and up: the big tiles (and their split tail launch) fill well enough
    const long small = 4 * big;
    const double fill_big = (double)big / (double)((big + 255) / 256 * 256);
    const double fill_small = (double)small / (double)((small + 255) / 256 * 256);
    return 0.85 * fill_small > fill_big;
}

}  // namespace mvq
