// ar_fused.hip -- the chunked auto-regressive loop of the proposed codec as ONE persistent kernel (mvq_ar_latents_f32).
//
// Reference: the `for s in range(0, Tlat, AR_CHUNK_TOK)` loop of AllPredAR.forward_step (Training/compare_dacvsproposal_5.py:302-320)
// == ProposedEval.encode_latents (Evaluation/dac_vcpwq_proposed6_latency.py:461-477): per 16-token chunk CrossPredictor (PosEnc +
// LayerNorm, Q projection, 8-head attention against the audio keys / values, output projection, FFN), the residual
// tanh(TokenNorm(zt - z_pred)) * scale, proj_down, the residual VQ, proj_up, and the write into z_run that the next chunk reads.
//
// Why one kernel (profiles/r05_trace_summary_B1_encode_mid.json): at the reference's operating points -- one segment, or its batch
// of six -- a chunk is 16-96 tokens, and its 17 launches are each a few microseconds of dependent chain behind ~5 us of launch
// boundary: 1.2 of the 2.6 ms of a one-segment encode.  The chains are fixed by the arithmetic contract; the boundaries are not.
// Here every stage of every chunk runs inside one grid of persistent 512-thread blocks separated by a grid-wide barrier
// (one device-scope atomic per block + an L1 invalidate), and each stage's tasks -- 16 x 16 output tiles on v_mfma_f32_16x16x4_f32
// through the SAME device function as conv_lat.hip's kernel, 4-token LayerNorm tiles, (item, head) attention tiles, one token per
// block for the codebook search -- are dealt round-robin to the blocks / waves.  Every task restates the arithmetic of the
// stand-alone kernel it replaces (kernels_small.hip: layernorm_c_tile_kernel, attention_kernel; kernels_vq.hip:
// rvq_ema_forward_rows_kernel), so the result is bit-identical to the launch-per-stage path (tests/test_gpu_ar_fused.py) and to
// the oracle.
//
// Co-residency: launched with hipLaunchCooperativeKernel (the runtime refuses a grid that cannot be resident at once).  The
// barrier spins with a bound: if it is ever exceeded the block raises the error word, every block leaves at its next barrier and
// the host call reports MVQ_EHIP -- a wave never waits for ever.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/mvq.h"
#include "conv_dispatch.hpp"
#include "conv_lat.hpp"
#include "kernels_small.hpp"
#include "det_math.hpp"
#include "ln_lat.hpp"

namespace mvq {
void set_last_error(const char* msg);
hipError_t launch_layernorm_lat_io(const LnIo& io, int B, int C, int n, hipStream_t s);      // kernels_small.hip

namespace {

constexpr int NTHR = 512;
constexpr int C_LAT = 1024, C_FF = 2048, D_CODE = 96, HEADS = 8, DH = C_LAT / HEADS, CHUNK = 16;

struct ArK {                 // kernel-side copy of mvq_ar_args + derived pointers
    mvq_ar_args a;
    // chunk-local buffers (workspace): per item rows of pitch 16 (GEMM operands) or pitch Tlat (residual partners of pitch-Tlat outputs)
    float *q16, *qT, *Q16, *ctx16, *y1T, *hdn16, *h16, *zpT, *rN16, *rD16, *qD16;
    unsigned* bar;           // [0] arrival counter, [1] error word
    unsigned long long* ts;  // MVQ_AR_TIMING: block 0's clock before / after every grid barrier (null: off)
    int nblocks;
};

// The thread index, re-read opaquely by every task: without this the compiler hoists each stage's per-thread index arithmetic out of
// the chunk loop to the kernel's entry, where the union of all stages' indices is live at once and spills (isa_lint R3).
__device__ __forceinline__ int tid_here() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }

// Activations written by other blocks in an earlier stage.  Every such read has a per-lane address, so it is a vector load through the
// L1 the barrier's acquire fence invalidated.  (It was a nontemporal load first: those took ~8 us per round trip, gpurun_out/f7.)
__device__ __forceinline__ float ld_act(const float* p) { return *p; }

// ---- grid barrier ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool grid_sync(const ArK& k, unsigned& gen, int* s_okp)
{
    int& s_ok = *s_okp;                                             // in the dynamic LDS region (the 160 KB opt-in covers dynamic LDS only)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                            // release: this block's stores are visible device-wide
        gen += 1;
        const unsigned target = gen * (unsigned)k.nblocks;
        atomicAdd(&k.bar[0], 1u);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(&k.bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&k.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > (1u << 22)) {
                __hip_atomic_store(&k.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // bounded wait: give up, everybody leaves
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __threadfence();                                            // acquire: invalidates this CU's vector L1 (other blocks' stores become readable)
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

// ---- attention, one (item, head) per task (restates attention_kernel) ------------------------------------------------------------
__device__ __forceinline__ void attn_task(const float* Q, size_t qsb, size_t qsc, const float* K, const float* V, size_t ksb, size_t ksc,
                                          float* ctx, int b, int hd, int Tq, int Tk, float* sm)
{
    constexpr int dh = DH;
    float* Qs = sm;                       // [dh][Tq]
    float* Ks = Qs + dh * Tq;             // [dh][Tk]
    float* Vs = Ks + dh * Tk;             // [dh][Tk]
    float* P = Vs + dh * Tk;              // [Tq][Tk]
    const int tid = tid_here();
    const float* q = Q + (size_t)b * qsb + (size_t)hd * dh * qsc;
    const float* kb = K + (size_t)b * ksb + (size_t)hd * dh * ksc;
    const float* vb = V + (size_t)b * ksb + (size_t)hd * dh * ksc;
    __syncthreads();
    for (int e = tid; e < dh * Tq; e += NTHR) { const int d = e / Tq, i = e - d * Tq; Qs[e] = ld_act(q + (size_t)d * qsc + i); }
    for (int e = tid; e < dh * Tk; e += NTHR) {
        const int d = e / Tk, j = e - d * Tk;
        Ks[e] = kb[(size_t)d * ksc + j];
        Vs[e] = vb[(size_t)d * ksc + j];
    }
    __syncthreads();
    const float rs = __builtin_sqrtf((float)dh);
    for (int p = tid; p < Tq * Tk; p += NTHR) {
        const int i = p / Tk, j = p - i * Tk;
        float a = 0.0f;
        for (int d = 0; d < dh; d += 8) {
            float qv[8], kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { qv[u] = Qs[(d + u) * Tq + i]; kv[u] = Ks[(d + u) * Tk + j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) a = dfma(qv[u], kv[u], a);
        }
        P[p] = a / rs;
    }
    __syncthreads();
    if (tid < Tq) {
        float* pr = P + tid * Tk;
        float m = -__builtin_inff();
        for (int j = 0; j < Tk; ++j) m = __builtin_fmaxf(m, pr[j]);
        float l = 0.0f;
        for (int j = 0; j < Tk; ++j) { const float e = det_exp(pr[j] - m); pr[j] = e; l = l + e; }
        for (int j = 0; j < Tk; ++j) pr[j] = pr[j] / l;
    }
    __syncthreads();
    float* out = ctx + (size_t)b * qsb + (size_t)hd * dh * qsc;
    for (int e = tid; e < dh * Tq; e += NTHR) {
        const int d = e / Tq, i = e - d * Tq;
        float a = 0.0f;
        for (int j = 0; j < Tk; ++j) a = dfma(P[i * Tk + j], Vs[d * Tk + j], a);
        out[(size_t)d * qsc + i] = a;
    }
}

// ---- residual VQ of one token (restates rvq_ema_forward_rows_kernel<24>: the block's 512 threads = one per code) ---------------------
__device__ __forceinline__ void amax_combine_(float& s, int& i, float os, int oi) { if (os > s || (os == s && oi < i)) { s = os; i = oi; } }
__device__ __forceinline__ void rvq_task(const ArK& k, int b, int i, int s0, int n, float* sm)
{
    constexpr int DV = 24, D = 4 * DV, HV = DV / 2, PITCH = 4 * HV + 4;
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int K = k.a.rvq_k, nb = k.a.books_use;
    float* const res = sm;
    float* const qs = res + D;
    float* const qrow = qs + D;
    float* const ws = qrow + D;
    int* const wi = reinterpret_cast<int*>(ws + 8);
    float* const rows = ws + 16;
    const int tid = tid_here(), lane = tid & 63, wave = tid >> 6;
    const bool has = tid < K;
    const int npieces = K * HV;
    const float* zin = k.rD16 + ((size_t)b * D) * CHUNK + i;                  // rD[b][d][i], pitch 16
    __syncthreads();
    if (tid < D) { res[tid] = ld_act(zin + (size_t)tid * CHUNK); qs[tid] = 0.0f; }
    v4 st[HV];
    unsigned goff[HV], loff[HV];
    unsigned livem = 0;
#pragma unroll
    for (int u = 0; u < HV; ++u) {
        const int e0 = tid + NTHR * u;
        const int ec = e0 < npieces ? e0 : npieces - 1;
        const int r = ec / HV, v = ec - r * HV;
        goff[u] = (unsigned)(r * D + 4 * v) * 4u;
        loff[u] = (unsigned)(r * PITCH + 4 * v) * 4u;
        livem |= e0 < npieces ? (1u << u) : 0u;
    }
    auto gfetch = [&](int bk, int h) __attribute__((always_inline)) {
        const uintptr_t bu = reinterpret_cast<uintptr_t>(k.a.books + (size_t)bk * K * D + h * 4 * HV);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bu), hi = __builtin_amdgcn_readfirstlane((unsigned)(bu >> 32));
        typedef const __attribute__((address_space(1))) char* gptr;          // rebuilt from two SGPRs: say it is global memory (not a flat access)
        typedef const __attribute__((address_space(1))) v4* gv4;
        const gptr base = (gptr)(((uintptr_t)hi << 32) | lo);
#pragma unroll
        for (int u = 0; u < HV; ++u) st[u] = *(gv4)(base + goff[u]);
    };
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < HV; ++u)
            if ((livem >> u) & 1u) *reinterpret_cast<v4*>(reinterpret_cast<char*>(rows) + loff[u]) = st[u];
    };
    v4 e0[HV];
    if (nb > 0) gfetch(0, 0);
    for (int bk = 0; bk < nb; ++bk) {
        const float* myrow = rows + (has ? tid : 0) * PITCH;
        float dot = 0.0f, hs = 0.0f;
        auto half = [&](auto H) __attribute__((always_inline)) {
            constexpr int h = decltype(H)::value;
            if (h == 1) __syncthreads();
            lstore();
            __syncthreads();
            if (h == 0) gfetch(bk, 1);
            else if (bk + 1 < nb) gfetch(bk + 1, 0);
            v4 ec[HV];
#pragma unroll
            for (int u = 0; u < HV; ++u) ec[u] = *reinterpret_cast<const v4*>(myrow + 4 * u);
            if (h == 0) {
#pragma unroll
                for (int u = 0; u < HV; ++u) e0[u] = ec[u];
            }
#pragma unroll
            for (int u = 0; u < HV; ++u) {
                const v4 r = *reinterpret_cast<const v4*>(res + 4 * (h * HV + u));
                const v4 c = ec[u];
                dot = dfma(r.x, c.x, dot); dot = dfma(r.y, c.y, dot); dot = dfma(r.z, c.z, dot); dot = dfma(r.w, c.w, dot);
                hs = dfma(c.x, c.x, hs); hs = dfma(c.y, c.y, hs); hs = dfma(c.z, c.z, hs); hs = dfma(c.w, c.w, hs);
            }
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
        float bs = -__builtin_inff(); int bi = 0x7fffffff;
        if (has) {
            const float sc = dot - 0.5f * hs;
            if (sc > bs) { bs = sc; bi = tid; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            amax_combine_(bs, bi, os, oi);
        }
        if (lane == 0) { ws[wave] = bs; wi[wave] = bi; }
        __syncthreads();
        float cs = ws[0]; int id = wi[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) amax_combine_(cs, id, ws[w], wi[w]);
        if (id < 0 || id >= K) id = 0;
        if (tid == id) {
#pragma unroll
            for (int u = 0; u < HV; ++u) *reinterpret_cast<v4*>(qrow + 4 * u) = e0[u];
        }
        if (tid >= 64 && tid < 64 + 4 * HV) qrow[4 * HV + tid - 64] = rows[id * PITCH + tid - 64];
        __syncthreads();
        if (tid < D) {
            const float q = qrow[tid];
            const float r = res[tid];
            qs[tid] = (qs[tid] + (q - r)) + r;
            res[tid] = r - q;
        }
        if (k.a.idx_out && tid == 0) k.a.idx_out[((size_t)bk * k.a.batch + b) * k.a.t_lat + s0 + i] = id;
    }
    __syncthreads();
    if (tid < D) k.qD16[((size_t)b * D + tid) * CHUNK + i] = (nb > 0) ? qs[tid] : 0.0f;
    (void)n;
}

// ---- stage table: a chunk is eleven stages; the HOST writes the table into the kernel's argument block, every block walks it -------
// One inlined instance of each task type (a switch over the stage's type), and the stage's operands are read from the argument
// block where the stage starts (scalar loads through an opaque pointer): nothing of one stage is live in another.
enum { ST_SKIP = 0, ST_LN, ST_GEMM64, ST_GEMM32, ST_ATTN, ST_RVQ };
enum { ADV_X = 1, ADV_RES = 2, ADV_Y = 4, ADV_LN_X = 8, ADV_LN_PREV = 16 /* also: absent in chunk 0 */, ADV_LN_SUB = 32, ADV_LN_Y1 = 64 };
struct StageDesc {           // pointers as of chunk 0; the ones named in `adv` move by s floats in the chunk that starts at token s
    int type, adv;
    // GEMM: y[b][cout][pitch y_pitch] = act(Wp . x[b][cin][16] + bias + res)
    int cin, cout, act, y_pitch;
    const float *x, *wp, *bias, *res;
    float* y;
    LnIo ln;
};
constexpr int N_STAGES = 11;
struct ArKT { ArK k; StageDesc tab[N_STAGES]; };
static_assert(sizeof(ArKT) <= 4096, "kernel argument block");

typedef const __attribute__((address_space(4))) ArKT* ArgPtr;
__device__ __forceinline__ const ArKT& args_here()
{
    ArgPtr p = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                     // not hoistable: each stage loads what it needs, where it needs it
    return *(const ArKT*)p;
}

template <int CG>
__device__ __forceinline__ void gemm_stage(const ArK& k, const StageDesc& d, int s, int n, float* wave_smem)
{
    ConvArgs a{};
    const int adv = d.adv;
    a.x = d.x + ((adv & ADV_X) ? s : 0); a.wp = d.wp; a.bias = d.bias;
    a.residual = d.res ? d.res + ((adv & ADV_RES) ? s : 0) : nullptr;
    a.y = d.y + ((adv & ADV_Y) ? s : 0);
    a.B = k.a.batch; a.Cin = d.cin; a.Tin = CHUNK; a.Cout = d.cout; a.Tout = d.y_pitch; a.pad = 0;
    a.Mpad = d.cout /* 1024, 2048 and 96 are whole row tiles: conv_mpad(cout) == cout, checked by the host */; a.Mrows = d.cout;
    a.Ncols = n; a.act = d.act; a.up_s = 1; a.n_tiles = 1; a.vec4 = 1;
    const int mt = (d.cout + 15) / 16, ntasks = k.a.batch * mt;
    const int wave = __builtin_amdgcn_readfirstlane(tid_here() >> 6), nwaves = k.nblocks * (NTHR / 64);
    for (int t = wave * k.nblocks + blockIdx.x; t < ntasks; t += nwaves)            // across the CUs first, then a CU's second wave
        conv_lat_body<1, 1, 1, CG, true, true>(a, t / mt, t % mt, wave_smem);
}

__device__ __forceinline__ void ln_stage(const ArK& k, const StageDesc& d, int s, int n, float* sm)
{
    LnIo io = d.ln;
    const int adv = d.adv;
    if (io.x && (adv & ADV_LN_X)) io.x += s;
    if (adv & ADV_LN_PREV) io.prev = s > 0 ? io.prev + (s - 1) : nullptr;
    if (io.sub && (adv & ADV_LN_SUB)) io.sub += s;
    if (io.y1 && (adv & ADV_LN_Y1)) io.y1 += s;
    const int B = k.a.batch;
    const int tid = tid_here();
    for (int t = blockIdx.x; t < (B * n + 3) / 4; t += k.nblocks) ln_lat_task(io, t, B, n, C_LAT, sm, tid, [](const float* p) { return ld_act(p); });
}

__global__ __launch_bounds__(NTHR) __attribute__((amdgpu_waves_per_eu(2, 2)))
void ar_loop_kernel(const ArKT kt_unused)
{
    (void)kt_unused;                                                // read through args_here()
    extern __shared__ __attribute__((aligned(16))) float sm_raw[];
    int* const s_ok = reinterpret_cast<int*>(sm_raw);               // first 16 bytes: the barrier's verdict; the tasks' LDS follows
    float* const sm = sm_raw + 4;
    unsigned gen = 0;
    const int Tl = args_here().k.a.t_lat;
    for (int s = 0; s < Tl; s += CHUNK) {
        const int n = Tl - s < CHUNK ? Tl - s : CHUNK;
        for (int st = 0; st < N_STAGES; ++st) {
            const ArKT& kt = args_here();
            const ArK& k = kt.k;
            const StageDesc& d = kt.tab[st];
            const int type = d.type;
            if (type == ST_SKIP) continue;
            if (type == ST_LN) {
                ln_stage(k, d, s, n, sm);
            } else if (type == ST_GEMM64 || type == ST_GEMM32) {
                float* const wave_smem = sm + (size_t)(tid_here() >> 6) * (2 * LatCfg<1, 1, 1, 64>::BUF_FLOATS);
                if (type == ST_GEMM64) gemm_stage<64>(k, d, s, n, wave_smem);
                else gemm_stage<32>(k, d, s, n, wave_smem);
            } else if (type == ST_ATTN) {
                const mvq_ar_args& a = k.a;
                const int B = a.batch;
                const int ka = (a.t_audio < s + n ? a.t_audio : s + n) - (a.t_audio < s ? a.t_audio : s);      // audio tokens under this chunk
                for (int t = blockIdx.x; t < B * HEADS; t += k.nblocks) {
                    const int b = t / HEADS, hd = t - b * HEADS;
                    attn_task(k.Q16, (size_t)C_LAT * CHUNK, CHUNK, a.k_all + (size_t)b * a.t_audio + s, a.v_all + (size_t)b * a.t_audio + s, 0,
                              (size_t)B * a.t_audio, k.ctx16, b, hd, n, ka, sm);
                }
            } else {                                                // ST_RVQ (+ the tokens the EMA update sees)
                const mvq_ar_args& a = k.a;
                const int B = a.batch;
                for (int t = blockIdx.x; t < B * n; t += k.nblocks) rvq_task(k, t / n, t % n, s, n, sm);
                if (a.r_tokens)
                    for (int e = blockIdx.x * NTHR + tid_here(); e < B * D_CODE * n; e += k.nblocks * NTHR) {
                        const int i = e % n, bd = e / n;
                        a.r_tokens[(size_t)bd * Tl + s + i] = ld_act(k.rD16 + (size_t)bd * CHUNK + i);
                    }
            }
            {
                const ArK& kk = args_here().k;
                unsigned long long* const ts = kk.ts;
                if (ts && blockIdx.x == 0 && threadIdx.x == 0 && gen < 120) { ts[2 * gen] = wall_clock64(); ts[256 + 2 * gen] = clock64(); }
                if (!grid_sync(kk, gen, s_ok)) return;
                if (ts && blockIdx.x == 0 && threadIdx.x == 0 && gen <= 120) { ts[2 * gen - 1] = wall_clock64(); ts[256 + 2 * gen - 1] = clock64(); }
            }
        }
    }
}

// zero fill as a kernel: inside a captured graph a memset NODE costs far more than the ~0.5 MB it clears (graph replay of a one-segment
// encode went 2.3 -> 2.8 ms with hipMemsetAsync here)
__global__ void ar_zero_kernel(float4* __restrict__ p, size_t n4)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
}

// host: the eleven stages of a chunk (pointers as of chunk 0)
void fill_stages(StageDesc* tab, const ArK& k)
{
    const mvq_ar_args& a = k.a;
    const int Tl = a.t_lat;
    const size_t p16 = CHUNK, cT = (size_t)C_LAT * Tl, c16 = (size_t)C_LAT * p16;
    for (int i = 0; i < N_STAGES; ++i) { tab[i] = StageDesc{}; tab[i].type = ST_SKIP; }
    const float* zpred = nullptr;
    auto gemm = [&](int i, int type, const float* x, int cin, const float* wp, const float* bias, int cout, const float* res, int act, float* y, int pitch, int adv) {
        StageDesc& d = tab[i];
        d.type = type; d.adv = adv; d.x = x; d.cin = cin; d.wp = wp; d.bias = bias; d.cout = cout; d.res = res; d.act = act; d.y = y; d.y_pitch = pitch;
    };
    if (!a.tactile_only) {
        {   // S1: q = LN(zt_prev + PE): the shift-by-one input is zero except column 0 of a chunk with s > 0 (= z_run[..., s - 1])
            LnIo& io = tab[0].ln; tab[0].type = ST_LN; tab[0].adv = ADV_LN_PREV | ADV_LN_Y1;
            io.x = nullptr; io.prev = a.z_run; io.prev_sb = cT; io.prev_sc = (size_t)Tl;
            io.pe = a.pe; io.gamma = a.lnq_g; io.beta = a.lnq_b; io.eps = a.ln_eps; io.post_scale = 1.0f;
            io.y0 = k.q16; io.y0_sb = c16; io.y0_sc = p16;
            io.y1 = k.qT; io.y1_sb = cT; io.y1_sc = (size_t)Tl;
        }
        gemm(1, ST_GEMM64, k.q16, C_LAT, a.wq, nullptr, C_LAT, nullptr, 0, k.Q16, CHUNK, 0);                       // S2: Q
        tab[2].type = ST_ATTN;                                                                                       // S3
        gemm(3, ST_GEMM64, k.ctx16, C_LAT, a.wo, nullptr, C_LAT, k.qT, 0, k.y1T, Tl, ADV_RES | ADV_Y);             // S4: y1 = out(ctx) + q
        {   // S5: hdn = LN(y1)
            LnIo& io = tab[4].ln; tab[4].type = ST_LN; tab[4].adv = ADV_LN_X;
            io.x = k.y1T; io.x_sb = cT; io.x_sc = (size_t)Tl;
            io.gamma = a.lnf_g; io.beta = a.lnf_b; io.eps = a.ln_eps; io.post_scale = 1.0f;
            io.y0 = k.hdn16; io.y0_sb = c16; io.y0_sc = p16;
        }
        gemm(5, ST_GEMM64, k.hdn16, C_LAT, a.w1, a.b1, C_FF, nullptr, MVQ_ACT_GELU, k.h16, CHUNK, 0);              // S6: GELU(ffn[1])
        gemm(6, ST_GEMM64, k.h16, C_FF, a.w3, a.b3, C_LAT, k.y1T, 0, k.zpT, Tl, ADV_RES | ADV_Y);                  // S7: z_pred = ffn[3] + y1
        zpred = k.zpT;
    }
    {   // S8: rN = tanh(TokenNorm(zt - z_pred)) * scale
        LnIo& io = tab[7].ln; tab[7].type = ST_LN; tab[7].adv = ADV_LN_X | ADV_LN_SUB;
        io.x = a.zt; io.x_sb = cT; io.x_sc = (size_t)Tl;
        io.sub = zpred; io.sub_sb = cT; io.sub_sc = (size_t)Tl;
        io.gamma = a.tok_g; io.beta = a.tok_b; io.eps = a.tok_eps; io.post_scale = a.scale; io.do_tanh = 1;
        io.y0 = k.rN16; io.y0_sb = c16; io.y0_sc = p16;
    }
    gemm(8, ST_GEMM64, k.rN16, C_LAT, a.wd, a.bd, D_CODE, nullptr, 0, k.rD16, CHUNK, 0);                           // S9: proj_down
    tab[9].type = ST_RVQ;                                                                                            // S10
    gemm(10, ST_GEMM32, k.qD16, D_CODE, a.wu, a.bu, C_LAT, zpred, 0, a.z_run, Tl, ADV_RES | ADV_Y);                // S11: z_hat -> z_run
}

}  // namespace
}  // namespace mvq

extern "C" {

size_t mvq_ar_workspace_bytes(int batch, int t_lat)
{
    if (batch <= 0 || t_lat <= 0) return 0;
    const size_t c = mvq::C_LAT, p = mvq::CHUNK;
    const size_t f16 = (size_t)batch * p, fT = (size_t)batch * t_lat;
    const size_t floats = c * f16 * 5 /* q16 Q16 ctx16 hdn16 rN16 */ + (size_t)mvq::C_FF * f16 + 2 * (size_t)mvq::D_CODE * f16 + c * fT * 3 /* qT y1T zpT */;
    return floats * sizeof(float) + 256 /* alignment */ + 256 /* barrier words */ + 8192 /* stage clocks */;
}

// argument checks + workspace carving shared by the two forms; returns MVQ_OK with `done` set when there is nothing to do
static int ar_prepare(const mvq_ar_args* args, void* workspace, size_t workspace_bytes, mvq::ArKT& kt, bool& done)
{
    using namespace mvq;
    auto fail = [](int code, const char* msg) { set_last_error(msg); return code; };
    done = false;
    if (!args) return fail(MVQ_EINVAL, "ar_latents: null argument");
    const mvq_ar_args& a = *args;
    if (a.batch < 0 || a.t_lat < 0 || a.t_audio < 0 || a.books_use < 0) return fail(MVQ_EINVAL, "ar_latents: bad shape");
    if (a.batch == 0 || a.t_lat == 0) { done = true; return MVQ_OK; }
    if (a.c_lat != C_LAT || a.c_ff != C_FF || a.code_dim != D_CODE || a.heads != HEADS || a.chunk != CHUNK)
        return fail(MVQ_EUNSUPPORTED, "ar_latents: the fused loop covers the reference's shapes (c_lat 1024, FFN 2048, 8 heads, code dim 96, chunks of 16)");
    if (a.rvq_k <= 0 || a.rvq_k > 512) return fail(MVQ_EUNSUPPORTED, "ar_latents: K <= 512 codes per book");
    if (!a.zt || !a.z_run || !a.tok_g || !a.tok_b || !a.wd || !a.bd || !a.wu || !a.bu || (a.books_use > 0 && !a.books) || !workspace)
        return fail(MVQ_EINVAL, "ar_latents: null tensor");
    if (!a.tactile_only && (!a.pe || !a.lnq_g || !a.lnq_b || !a.wq || !a.wo || !a.lnf_g || !a.lnf_b || !a.w1 || !a.b1 || !a.w3 || !a.b3 ||
                            (a.t_audio > 0 && (!a.k_all || !a.v_all))))
        return fail(MVQ_EINVAL, "ar_latents: null predictor tensor");
    if (workspace_bytes < mvq_ar_workspace_bytes(a.batch, a.t_lat)) return fail(MVQ_EINVAL, "ar_latents: workspace too small (mvq_ar_workspace_bytes)");
    if ((reinterpret_cast<uintptr_t>(a.books) & 15) != 0) return fail(MVQ_EINVAL, "ar_latents: books must be 16-byte aligned");
    if (conv_mpad(C_LAT) != C_LAT || conv_mpad(C_FF) != C_FF || conv_mpad(D_CODE) != D_CODE) return fail(MVQ_EUNSUPPORTED, "ar_latents: packed-row padding");

    kt = ArKT{};
    ArK& k = kt.k;
    k.a = a;
    char* p = reinterpret_cast<char*>(workspace);
    p += (256 - reinterpret_cast<uintptr_t>(p) % 256) % 256;
    k.bar = reinterpret_cast<unsigned*>(p); p += 256;
    static const bool timing = getenv("MVQ_AR_TIMING") != nullptr;      // debugging aid: per-stage clocks, printed by mvq_ar_check
    k.ts = timing ? reinterpret_cast<unsigned long long*>(p) : nullptr;
    p += 8192;
    auto take = [&](size_t floats) { float* q = reinterpret_cast<float*>(p); p += floats * sizeof(float); return q; };
    const size_t f16 = (size_t)a.batch * CHUNK, fT = (size_t)a.batch * a.t_lat;
    k.q16 = take(C_LAT * f16); k.Q16 = take(C_LAT * f16); k.ctx16 = take(C_LAT * f16); k.hdn16 = take(C_LAT * f16); k.rN16 = take(C_LAT * f16);
    k.h16 = take((size_t)C_FF * f16); k.rD16 = take((size_t)D_CODE * f16); k.qD16 = take((size_t)D_CODE * f16);
    k.qT = take(C_LAT * fT); k.y1T = take(C_LAT * fT); k.zpT = take(C_LAT * fT);
    fill_stages(kt.tab, k);
    return MVQ_OK;
}

int mvq_ar_latents_f32(const mvq_ar_args* args, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace mvq;
    auto fail = [](int code, const char* msg) { set_last_error(msg); return code; };
    ArKT kt;
    bool done = false;
    if (const int rc = ar_prepare(args, workspace, workspace_bytes, kt, done); rc != MVQ_OK || done) return rc;
    ArK& k = kt.k;
    const mvq_ar_args& a = k.a;

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds_gemm = (size_t)(NTHR / 64) * 2 * LatCfg<1, 1, 1, 64>::BUF_FLOATS * sizeof(float);
    const size_t lds_rvq = ((size_t)a.rvq_k * 52 + 3 * 96 + 16) * sizeof(float);
    const size_t lds_ln = ln_lat_lds_floats(C_LAT) * sizeof(float);
    const size_t lds_att = ((size_t)DH * 3 * CHUNK + CHUNK * CHUNK) * sizeof(float);
    size_t lds = lds_gemm;
    for (size_t v : {lds_rvq, lds_ln, lds_att}) lds = v > lds ? v : lds;
    lds += 16;
    static BigLdsOptIn opt;
    if (hipError_t e = opt.ensure(reinterpret_cast<const void*>(ar_loop_kernel)); e != hipSuccess) return fail(MVQ_EHIP, "ar_latents: LDS opt-in refused");
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(MVQ_EHIP, "ar_latents: device query failed");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ar_loop_kernel, NTHR, lds) != hipSuccess || per_cu < 1)
        return fail(MVQ_EHIP, "ar_latents: the persistent block does not fit a CU");
    // blocks: one per CU up to what the widest stage can use (FFN: batch x 128 tiles, one wave each; the search: one token per block)
    const int want = a.batch * (C_FF / 16);
    k.nblocks = want < cus ? want : cus;
    if (k.nblocks < 1) k.nblocks = 1;
    if (hipMemsetAsync(k.bar, 0, k.ts ? 256 + 8192 : 256, st) != hipSuccess) return fail(MVQ_EHIP, "ar_latents: memset failed");
    void* kargs[] = {&kt};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(ar_loop_kernel), dim3((unsigned)k.nblocks), dim3(NTHR), kargs, (unsigned)lds, st);
    if (e != hipSuccess) {
        static char msg[160];
        snprintf(msg, sizeof(msg), "ar_latents: cooperative launch of %d blocks x %d threads, %zu B LDS: %s", k.nblocks, NTHR, lds, hipGetErrorString(e));
        return fail(MVQ_EHIP, msg);
    }
    return MVQ_OK;
}

/* The same loop as ONE HOST CALL of stand-alone launches: the stage table of the persistent kernel above, each stage launched as the
 * kernel it restates (conv1d_lat_kernel, layernorm_c_lat_kernel, attention_kernel, rvq_ema_forward_rows_kernel), stream-ordered,
 * nothing allocated or synchronised (capturable).  Why it exists: at one segment the Python loop's ~85 foreign calls with their
 * allocations cost as much host time as the kernels take on the device (eager encode 2.5 ms against 2.3 ms replayed as a graph);
 * from C a launch costs 2-3 us.  Same kernels, same bits as the Python loop (tests/test_gpu_ar_fused.py).  Needs every GEMM in the
 * latency form's range (batch <= 8: at most 1 024 16 x 16 tiles per launch). */
int mvq_ar_latents_staged_f32(const mvq_ar_args* args, void* workspace, size_t workspace_bytes, void* stream)
{
    using namespace mvq;
    auto fail = [](int code, const char* msg) { set_last_error(msg); return code; };
    ArKT kt;
    bool done = false;
    if (const int rc = ar_prepare(args, workspace, workspace_bytes, kt, done); rc != MVQ_OK || done) return rc;
    const ArK& k = kt.k;
    const mvq_ar_args& a = k.a;
    if (a.batch > 8) return fail(MVQ_EUNSUPPORTED, "ar_latents_staged: batch <= 8 (beyond it the LDS-tiled GEMMs of the per-stage loop win)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int B = a.batch, Tl = a.t_lat;
    // chunk-local buffers start defined: a last chunk shorter than 16 tokens leaves columns nobody writes, and the search reads all 16
    const size_t chunk_bytes = (size_t)(reinterpret_cast<char*>(k.qT) - reinterpret_cast<char*>(k.q16));
    {
        const size_t n4 = chunk_bytes / 16;                        // the buffers are carved in whole 64-byte rows from a 256-byte aligned base
        size_t blocks = (n4 + 255) / 256; if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(ar_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(k.q16), n4);
        if (hipGetLastError() != hipSuccess) return fail(MVQ_EHIP, "ar_latents_staged: zero fill failed");
    }
    int32_t* const idx_tmp = reinterpret_cast<int32_t*>(k.ctx16);          // [books][B][16]: ctx16 is free between the attention and the next chunk
    if ((size_t)a.books_use * B * CHUNK > (size_t)C_LAT * B * CHUNK) return fail(MVQ_EUNSUPPORTED, "ar_latents_staged: too many books");
    hipError_t e = hipSuccess;
    for (int s = 0; s < Tl && e == hipSuccess; s += CHUNK) {
        const int n = Tl - s < CHUNK ? Tl - s : CHUNK;
        for (int i = 0; i < N_STAGES && e == hipSuccess; ++i) {
            const StageDesc& d = kt.tab[i];
            switch (d.type) {
                case ST_SKIP: break;
                case ST_LN: {
                    LnIo io = d.ln;
                    if (io.x && (d.adv & ADV_LN_X)) io.x += s;
                    if (d.adv & ADV_LN_PREV) io.prev = s > 0 ? io.prev + (s - 1) : nullptr;
                    if (io.sub && (d.adv & ADV_LN_SUB)) io.sub += s;
                    if (io.y1 && (d.adv & ADV_LN_Y1)) io.y1 += s;
                    e = launch_layernorm_lat_io(io, B, C_LAT, n, st);
                    break;
                }
                case ST_GEMM64: case ST_GEMM32: {
                    ConvArgs c{};
                    c.x = d.x + ((d.adv & ADV_X) ? s : 0); c.wp = d.wp; c.bias = d.bias;
                    c.residual = d.res ? d.res + ((d.adv & ADV_RES) ? s : 0) : nullptr;
                    c.y = d.y + ((d.adv & ADV_Y) ? s : 0);
                    c.B = B; c.Cin = d.cin; c.Tin = CHUNK; c.Cout = d.cout; c.Tout = d.y_pitch; c.pad = 0;
                    c.Mpad = d.cout; c.Mrows = d.cout; c.Ncols = n; c.act = d.act; c.up_s = 1;
                    e = launch_conv_lat(c, 1, 1, 1, st);
                    break;
                }
                case ST_ATTN: {
                    const int ka = (a.t_audio < s + n ? a.t_audio : s + n) - (a.t_audio < s ? a.t_audio : s);
                    e = launch_attention(k.Q16, a.k_all ? a.k_all + s : k.Q16, a.v_all ? a.v_all + s : k.Q16, k.ctx16, B, HEADS, DH, n, ka,
                                         (size_t)C_LAT * CHUNK, CHUNK, (size_t)a.t_audio, (size_t)B * a.t_audio, st);
                    break;
                }
                case ST_RVQ: {
                    if (a.books_use > 0)
                        e = launch_rvq_ema_forward(k.rD16, a.books, k.qD16, a.idx_out ? idx_tmp : nullptr, B, D_CODE, CHUNK, a.books_use, a.rvq_k, 1, st);
                    else {
                        hipLaunchKernelGGL(ar_zero_kernel, dim3(8), dim3(256), 0, st, reinterpret_cast<float4*>(k.qD16), (size_t)D_CODE * B * CHUNK / 4);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess && a.idx_out && a.books_use > 0)      // [books * B][16] -> idx_out[books * B][Tlat] at s (4-byte moves)
                        e = launch_strided3d(reinterpret_cast<const float*>(idx_tmp), CHUNK, 0, nullptr, 0, 0,
                                             reinterpret_cast<float*>(a.idx_out) + s, (size_t)Tl, 0, a.books_use * B, 1, n, st);
                    if (e == hipSuccess && a.r_tokens)
                        e = launch_strided3d(k.rD16, (size_t)D_CODE * CHUNK, CHUNK, nullptr, 0, 0, a.r_tokens + s, (size_t)D_CODE * Tl, (size_t)Tl, B, D_CODE, n, st);
                    break;
                }
            }
        }
    }
    if (e != hipSuccess) return fail(MVQ_EHIP, hipGetErrorString(e));
    return MVQ_OK;
}

/* after the stream has been synchronised: 0 = every grid barrier of the last call on this workspace completed */
int mvq_ar_check(const void* workspace, void* stream)
{
    if (!workspace) return MVQ_EINVAL;
    const char* p = reinterpret_cast<const char*>(workspace);
    p += (256 - reinterpret_cast<uintptr_t>(p) % 256) % 256;
    unsigned host[2] = {0, 0};
    if (hipMemcpyAsync(host, p, sizeof(host), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return MVQ_EHIP;
    if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return MVQ_EHIP;
    if (getenv("MVQ_AR_TIMING")) {
        static unsigned long long ts[512];
        if (hipMemcpy(ts, p + 256, sizeof(ts), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[ar timing] stage: work us / barrier us (block 0; 100 MHz clock)\n");
            for (int i = 1; i < 120 && ts[2 * i] != 0; ++i)
                fprintf(stderr, "  %3d: %7.2f / %6.2f   shader clock %.0f MHz\n", i, (double)(ts[2 * i] - ts[2 * i - 1]) / 100.0, (double)(ts[2 * i + 1] - ts[2 * i]) / 100.0,
                        (double)(ts[256 + 2 * i] - ts[256 + 2 * i - 1]) / ((double)(ts[2 * i] - ts[2 * i - 1]) / 100.0));
        }
    }
    if (getenv("MVQ_AR_TIMING")) {
        static unsigned long long ts[512];
        if (hipMemcpy(ts, p + 256, sizeof(ts), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[ar timing] stage: work us / barrier us (block 0; 100 MHz clock)\n");
            for (int i = 1; i < 120 && ts[2 * i] != 0; ++i)
                fprintf(stderr, "  %3d: %7.2f / %6.2f   shader clock %.0f MHz\n", i, (double)(ts[2 * i] - ts[2 * i - 1]) / 100.0, (double)(ts[2 * i + 1] - ts[2 * i]) / 100.0,
                        (double)(ts[256 + 2 * i] - ts[256 + 2 * i - 1]) / ((double)(ts[2 * i] - ts[2 * i - 1]) / 100.0));
        }
    }
    if (getenv("MVQ_AR_TIMING")) {
        unsigned long long d[8];
        if (hipMemcpy(d, p + 256 + 500 * 8, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[ln phases, cycles] load %llu sync %llu chain1 %llu d+syncs %llu chain2 %llu sync+out %llu\n", d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4], d[6] - d[5]);
    }
    if (host[1] != 0) { mvq::set_last_error("ar_latents: a grid barrier gave up (the grid was not co-resident)"); return MVQ_EHIP; }
    return MVQ_OK;
}

}  // extern "C"
