// mfma_probe.hip -- stand-alone probes of the gfx950 fp32 matrix pipe (no library code involved).
//
// Questions this answers on the box (results are quoted in DESIGN.md section 6b, raw output under profiles/):
//   1. Which clock does the chip hold under a dense fp32 MFMA stream, per MFMA shape (32x32x2 vs 16x16x4), with the
//      operands in registers and with the operands re-read from LDS every k-step (the real conv loop)?
//   2. Does VALU work of ANOTHER wave on the same SIMD overlap with fp32 MFMAs, or is it time the matrix pipe loses
//      (fp32 MFMA rate == fp32 VALU rate on this chip)?  Same question for VALU work inside the MFMA wave's own stream.
//
// build:  hipcc -O3 --offload-arch=gfx950 -o tools/_build/mfma_probe tools/mfma_probe.hip
// run:    tools/_build/mfma_probe > gpurun_out/mfma_probe.jsonl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// plain v_fma_f32 (inline asm keeps the SLP vectoriser from packing pairs into v_pk_fma_f32); -DPK_FMA: let it pack
#ifdef PK_FMA
#define VFMA(x, b) x = __builtin_fmaf(x, 0.999f, b)
#else
#define VFMA(x, b) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.999f), "v"(b))
#endif

struct Stamp { unsigned long long t0, t1, r0, r1; };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: 32x32x2, register operands          MODE 1: 16x16x4, register operands
// MODE 2: 32x32x2, operands from LDS          MODE 3: 16x16x4, operands from LDS      (same LDS bytes per flop)
// MODE 4: waves 0-3 as MODE 0, waves 4-7 run `nvalu` v_fma_f32 per MFMA-iteration (co-resident VALU work)
// MODE 5: MODE 0 with `nvalu` v_fma_f32 woven into the MFMA wave's own stream per 8 MFMAs
// MODE 6: VALU only (waves 4-7 of MODE 4 alone)
// One "iteration" = 32 768 flop per lane-group tile = 8 MFMAs 32x32x2 = 16 MFMAs 16x16x4.
template <int MODE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, Stamp* st, int iters, int nvalu)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mfma_wave = (MODE == 4) ? wave < 4 : (MODE == 6 ? false : true);
    const bool valu_wave = (MODE == 4) ? wave >= 4 : (MODE == 6);
    if (MODE == 6 && wave < 4) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = in[(tid * 8 + j) & 4095]; b[j] = in[(tid * 8 + j + 2048) & 4095]; }
    if (MODE == 2 || MODE == 3) {
        for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i & 4095];
        __syncthreads();
    }
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float sink = 0.0f;
    if (valu_wave) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a[j];
        for (int it = 0; it < iters; ++it) {
            for (int v = 0; v < nvalu; v += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) VFMA(x[j], b[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sink += x[j];
    } else if (mfma_wave) {
        if (MODE == 0 || MODE == 4 || MODE == 5) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = a[j];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s], b[2 * s + 1], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s + 1], b[2 * s], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s + 1], b[2 * s + 1], acc[3], 0, 0, 0);
                }
                if (MODE == 5) {
                    for (int v = 0; v < nvalu; v += 8) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) VFMA(x[j], b[j]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[i][r];
            if (MODE == 5)
#pragma unroll
                for (int j = 0; j < 8; ++j) sink += x[j];
        } else if (MODE == 1) {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sink += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        } else if (MODE == 2) {
            // the conv loop's shape: wave tile 64 x 64 = 2 x 2 tiles of 32 x 32; per k-step 2 A + 2 B ds_read_b32, 4 MFMAs
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
            const float* pa = lds + (lane & 31) + (lane >> 5) * 128 + wave * 64;
            const float* pb = lds + 4096 + (lane & 31) + (lane >> 5) * 160 + wave * 32;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float a0 = pa[s * 256], a1 = pa[s * 256 + 32], b0 = pb[s * 320], b1 = pb[s * 320 + 32];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
                }
                pa = lds + ((pa - lds + 512) & 2047);
                pb = lds + 4096 + ((pb - lds - 4096 + 640) & 2047);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[i][r];
        } else if (MODE == 3) {
            // same wave tile 64 x 64 = 4 x 4 tiles of 16 x 16; per k4-step 4 A + 4 B ds_read_b32, 16 MFMAs
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* pa = lds + (lane & 15) + (lane >> 4) * 128 + wave * 64;
            const float* pb = lds + 4096 + (lane & 15) + (lane >> 4) * 160 + wave * 32;
            for (int it = 0; it < iters; ++it) {
                float av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[i] = pa[i * 16]; bv[i] = pb[i * 16]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i * 4 + j], 0, 0, 0);
                pa = lds + ((pa - lds + 512) & 2047);
                pb = lds + 4096 + ((pb - lds - 4096 + 640) & 2047);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sink += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        }
    }
    if (lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        Stamp s{t0, t1, r0, r1};
        st[blockIdx.x * 8 + wave] = s;
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = sink;
}


// ---- LDS read shapes (second question of round 3): the conv loop pays ~12 % for its operand reads (mode 2 above).  Is the
// price per instruction, per LDS-array cycle, or per dword?  Same MFMAs, same operand VALUES per MFMA, different reads:
//   SHAPE 0  wave tile 64x64 (2x2), per k-step: A 2 dwords + B 2 dwords as ds_read_b32 / ds_read2_b32 (what hipcc emits today)
//   SHAPE 1  A: one ds_read_b128 per row tile per FOUR k-steps (k-interleaved weight image), B as SHAPE 0
//   SHAPE 2  A as SHAPE 1, B: one ds_read_b64 per k-step = two ADJACENT columns (column tiles interleaved even/odd), 8-byte aligned
//   SHAPE 3  SHAPE 2 with the B address at an odd dword (what odd taps / dilations give)
//   SHAPE 4  wave tile 64x128 (2x4), A as SHAPE 1, B: one ds_read_b128 per k-step = four adjacent columns, 16-byte aligned
//   SHAPE 5  SHAPE 4 with B at (16-byte + 4): unaligned ds_read_b128
//   SHAPE 6  SHAPE 4 with B as two ds_read_b64 at (8-byte + 4)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// All reads of one trip (4 k-steps) are issued by ONE asm statement that ends in s_waitcnt lgkmcnt(0) (hipcc neither counts nor
// reshapes asm loads), so every shape runs the same schedule: reads of a trip, wait, 16 (2x2) or 32 (2x4) MFMAs.
template <int SHAPE>
__global__ __launch_bounds__(256) void probe_lds(const float* __restrict__ in, float* __restrict__ out, Stamp* st, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = in[i & 4095];
    __syncthreads();
    constexpr int NT = SHAPE >= 5 ? 4 : 2;
    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    unsigned long long t0 = 0, r0 = 0;
    if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int l31 = lane & 31, h = lane >> 5;
    const int misal = (SHAPE == 3 || SHAPE == 4 || SHAPE == 6 || SHAPE == 7) ? 1 : 0;
    // byte addresses in LDS.  A image (SHAPE 0: [k][m] rows of 128 floats; else [h][m][4] per 4 k-steps)
    unsigned pa = (unsigned)(size_t)lds + (SHAPE == 0 ? (h * 128 + (wave & 1) * 64 + l31) * 4 : (h * 128 + (wave & 1) * 64 + l31) * 16);
    // B image: rows of 320 floats, row r at 16384 + r*1280 bytes; lane's first column
    unsigned pb = (unsigned)(size_t)lds + 16384 + h * 1280 + ((wave >> 1) * (NT * 32) + l31 * (SHAPE >= 2 ? NT : 1) + misal) * 4;
    for (int it = 0; it < iters; it += 2) {
        f32x4 a0, a1;                 // A of row tile 0 / 1 for k-steps 0..3
        float bv[4][NT];
        if (SHAPE == 0) {
            f32x2 p0, p1, p2, p3, q0, q1, q2, q3;
            asm volatile("ds_read2_b32 %0, %8 offset1:32\n\tds_read2_b32 %1, %8 offset0:64 offset1:96\n\t"
                         "ds_read2_b32 %2, %8 offset0:128 offset1:160\n\tds_read2_b32 %3, %8 offset0:192 offset1:224\n\t"
                         "ds_read2_b32 %4, %9 offset1:32\n\tds_read2_b32 %5, %9 offset0:80 offset1:112\n\t"
                         "ds_read2_b32 %6, %9 offset0:160 offset1:192\n\tds_read2_b32 %7, %9 offset0:208 offset1:240\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pa), "v"(pb) : "memory");
            a0 = f32x4{p0.x, p1.x, p2.x, p3.x}; a1 = f32x4{p0.y, p1.y, p2.y, p3.y};
            bv[0][0] = q0.x; bv[0][1] = q0.y; bv[1][0] = q1.x; bv[1][1] = q1.y; bv[2][0] = q2.x; bv[2][1] = q2.y; bv[3][0] = q3.x; bv[3][1] = q3.y;
        } else if (SHAPE == 1) {
            f32x2 q0, q1, q2, q3;
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:512\n\t"
                         "ds_read2_b32 %2, %7 offset1:32\n\tds_read2_b32 %3, %7 offset0:80 offset1:112\n\t"
                         "ds_read2_b32 %4, %7 offset0:160 offset1:192\n\tds_read2_b32 %5, %7 offset0:208 offset1:240\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pa), "v"(pb) : "memory");
            bv[0][0] = q0.x; bv[0][1] = q0.y; bv[1][0] = q1.x; bv[1][1] = q1.y; bv[2][0] = q2.x; bv[2][1] = q2.y; bv[3][0] = q3.x; bv[3][1] = q3.y;
        } else if (SHAPE == 2 || SHAPE == 3) {        // B: ds_read_b64 (aligned / at an odd dword)
            f32x2 q0, q1, q2, q3;
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:512\n\t"
                         "ds_read_b64 %2, %7\n\tds_read_b64 %3, %7 offset:320\n\t"
                         "ds_read_b64 %4, %7 offset:640\n\tds_read_b64 %5, %7 offset:960\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pa), "v"(pb) : "memory");
            bv[0][0] = q0.x; bv[0][1] = q0.y; bv[1][0] = q1.x; bv[1][1] = q1.y; bv[2][0] = q2.x; bv[2][1] = q2.y; bv[3][0] = q3.x; bv[3][1] = q3.y;
        } else if (SHAPE == 4) {                      // B: ds_read2_b32 of two ADJACENT dwords at an odd dword
            f32x2 q0, q1, q2, q3;
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:512\n\t"
                         "ds_read2_b32 %2, %7 offset1:1\n\tds_read2_b32 %3, %7 offset0:80 offset1:81\n\t"
                         "ds_read2_b32 %4, %7 offset0:160 offset1:161\n\tds_read2_b32 %5, %7 offset0:240 offset1:241\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pa), "v"(pb) : "memory");
            bv[0][0] = q0.x; bv[0][1] = q0.y; bv[1][0] = q1.x; bv[1][1] = q1.y; bv[2][0] = q2.x; bv[2][1] = q2.y; bv[3][0] = q3.x; bv[3][1] = q3.y;
        } else if (SHAPE == 5 || SHAPE == 6) {        // 2x4 tile, B: ds_read_b128 (aligned / at 16B + 4)
            f32x4 q0, q1, q2, q3;
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:512\n\t"
                         "ds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:320\n\t"
                         "ds_read_b128 %4, %7 offset:640\n\tds_read_b128 %5, %7 offset:960\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(pa), "v"(pb) : "memory");
            const f32x4 qq[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) { bv[sI][0] = qq[sI].x; bv[sI][1] = qq[sI].y; bv[sI][2 % NT] = qq[sI].z; bv[sI][3 % NT] = qq[sI].w; }
        } else {                                       // SHAPE 7: 2x4 tile, B: two ds_read2_b32 of adjacent dwords, odd dword
            f32x2 q0, q1, q2, q3, q4, q5, q6, q7;
            asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:512\n\t"
                         "ds_read2_b32 %2, %11 offset1:1\n\tds_read2_b32 %3, %11 offset0:2 offset1:3\n\t"
                         "ds_read2_b32 %4, %11 offset0:80 offset1:81\n\tds_read2_b32 %5, %11 offset0:82 offset1:83\n\t"
                         "ds_read2_b32 %6, %11 offset0:160 offset1:161\n\tds_read2_b32 %7, %11 offset0:162 offset1:163\n\t"
                         "ds_read2_b32 %8, %11 offset0:240 offset1:241\n\tds_read2_b32 %9, %11 offset0:242 offset1:243\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                         : "v"(pa), "v"(pb) : "memory");
            bv[0][0] = q0.x; bv[0][1] = q0.y; bv[0][2 % NT] = q1.x; bv[0][3 % NT] = q1.y;
            bv[1][0] = q2.x; bv[1][1] = q2.y; bv[1][2 % NT] = q3.x; bv[1][3 % NT] = q3.y;
            bv[2][0] = q4.x; bv[2][1] = q4.y; bv[2][2 % NT] = q5.x; bv[2][3 % NT] = q5.y;
            bv[3][0] = q6.x; bv[3][1] = q6.y; bv[3][2 % NT] = q7.x; bv[3][3 % NT] = q7.y;
        }
        const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
#pragma unroll
        for (int sI = 0; sI < 4; ++sI)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][sI], bv[sI][j], acc[i][j], 0, 0, 0);
        // walk through the images (keeps the alignment class of both addresses)
        pa ^= 4096u;
        pb ^= 8192u;
    }
    float sink = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
    if (lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        Stamp s{t0, t1, r0, r1};
        st[blockIdx.x * 8 + wave] = s;
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = sink;
}

// does the hardware return the bytes AT an unaligned address for ds_read_b64 / ds_read_b128 (or silently round it down)?
__global__ void probe_unaligned(float* out)
{
    __shared__ __attribute__((aligned(16))) float l[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = (float)i;
    __syncthreads();
    const unsigned p = (unsigned)(size_t)l + (threadIdx.x * 4 + 1) * 4;          // 16-byte + 4
    f32x2 q; f32x4 w;
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b128 %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q), "=&v"(w) : "v"(p) : "memory");
    out[threadIdx.x * 6 + 0] = q.x; out[threadIdx.x * 6 + 1] = q.y;
    out[threadIdx.x * 6 + 2] = w.x; out[threadIdx.x * 6 + 3] = w.y; out[threadIdx.x * 6 + 4] = w.z; out[threadIdx.x * 6 + 5] = w.w;
}

struct Result { double ms, tflops, clock_ghz; };

template <int MODE>
static Result run(const float* d_in, float* d_out, Stamp* d_st, int threads, int blocks_per_cu, int iters, int nvalu, double flop_per_iter_per_wave, int mfma_waves)
{
    const int grid = 256 * blocks_per_cu;
    size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 2048;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (lds < 32768 + 1024) lds = 32768 + 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm: ~1.5 s of back-to-back launches so the clock settles, then time 8 launches
    float ms = 0.0f;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), lds, 0, d_in, d_out, d_st, iters, nvalu);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int warm = (int)(1500.0 / (ms > 0.01 ? ms : 0.01));
    if (warm > 400) warm = 400;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), lds, 0, d_in, d_out, d_st, iters, nvalu);
    const int reps = 8;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), lds, 0, d_in, d_out, d_st, iters, nvalu);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamp> st((size_t)grid * 8);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    // median in-kernel clock over the waves that ran: shader cycles per 100 MHz realtime tick
    std::vector<double> clk;
    const int wpb = threads / 64;
    for (int bI = 0; bI < grid; ++bI)
        for (int w = 0; w < wpb; ++w) {
            const Stamp& s = st[(size_t)bI * 8 + w];
            if (s.r1 > s.r0 && s.t1 > s.t0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
        }
    double med = 0.0;
    if (!clk.empty()) { std::sort(clk.begin(), clk.end()); med = clk[clk.size() / 2]; }
    const double flops = flop_per_iter_per_wave * iters * mfma_waves * grid;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return Result{ms, flops / (ms * 1e-3) * 1e-12, med};
}


template <int SHAPE>
static Result run_lds(const float* d_in, float* d_out, Stamp* d_st, int blocks_per_cu, int iters)
{
    const int grid = 256 * blocks_per_cu;
    size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 2048;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_lds<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (lds < 32768 + 1024) lds = 32768 + 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.0f;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_lds<SHAPE>, dim3(grid), dim3(256), lds, 0, d_in, d_out, d_st, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    int warm = (int)(1500.0 / (ms > 0.01 ? ms : 0.01));
    if (warm > 400) warm = 400;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(probe_lds<SHAPE>, dim3(grid), dim3(256), lds, 0, d_in, d_out, d_st, iters);
    const int reps = 8;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_lds<SHAPE>, dim3(grid), dim3(256), lds, 0, d_in, d_out, d_st, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamp> st((size_t)grid * 8);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int bI = 0; bI < grid; ++bI)
        for (int w = 0; w < 4; ++w) {
            const Stamp& q = st[(size_t)bI * 8 + w];
            if (q.r1 > q.r0 && q.t1 > q.t0) clk.push_back((double)(q.t1 - q.t0) / (double)(q.r1 - q.r0) * 0.1);
        }
    double med = 0.0;
    if (!clk.empty()) { std::sort(clk.begin(), clk.end()); med = clk[clk.size() / 2]; }
    const int nt = SHAPE >= 5 ? 4 : 2;
    const double flops = 4096.0 * 2 * nt * 2.0 * iters * 4 * grid;      // per 2 iterations: 4 k-steps x (2 x NT) MFMAs
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return Result{ms, flops / (ms * 1e-3) * 1e-12, med};
}

int main(int argc, char** argv)
{
    float *d_in, *d_out; Stamp* d_st;
    std::vector<float> h(8192);
    srand(7);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    CHECK(hipMalloc(&d_in, h.size() * 4)); CHECK(hipMalloc(&d_out, (size_t)256 * 4 * 512 * 4)); CHECK(hipMalloc(&d_st, (size_t)256 * 4 * 8 * sizeof(Stamp)));
    CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_st, 0, (size_t)256 * 4 * 8 * sizeof(Stamp)));
    const bool zero = argc > 1 && !strcmp(argv[1], "0");
    if (zero) CHECK(hipMemset(d_in, 0, h.size() * 4));
    const double F = 32768.0;                 // flop per iteration per MFMA wave
    const int IT = 40000;                     // ~ 40000 * 8 MFMAs * 64 cycles = 20.5 M cycles ~ 9 ms per wave per block round
    #ifdef PK_FMA
    const char* data = zero ? "zeros,pk_fma" : "random,pk_fma";
#else
    const char* data = zero ? "zeros" : "random";
#endif

    if (argc > 1 && !strcmp(argv[1], "lds")) {
        {   // unaligned wide DS reads: value check
            float* d_u; CHECK(hipMalloc(&d_u, 64 * 6 * 4));
            hipLaunchKernelGGL(probe_unaligned, dim3(1), dim3(64), 0, 0, d_u);
            float hu[64 * 6]; CHECK(hipMemcpy(hu, d_u, sizeof(hu), hipMemcpyDeviceToHost));
            int ok64 = 1, ok128 = 1;
            for (int t = 0; t < 64; ++t) {
                const float base = (float)(t * 4 + 1);
                if (hu[t * 6] != base || hu[t * 6 + 1] != base + 1) ok64 = 0;
                for (int j = 0; j < 4; ++j) if (hu[t * 6 + 2 + j] != base + j) ok128 = 0;
            }
            printf("{\"probe\": \"unaligned_ds_read\", \"b64_at_odd_dword_exact\": %d, \"b128_at_16B_plus_4_exact\": %d, \"lane0_b128\": [%g, %g, %g, %g]}\n",
                   ok64, ok128, hu[2], hu[3], hu[4], hu[5]);
        }
        const char* names[8] = {"A 4x read2_b32, B 4x read2_b32 (cols n, n+32) per 4 k-steps (today)", "A 2x b128, B as today",
                                "A 2x b128, B 4x b64 aligned (2 adjacent columns)", "A 2x b128, B 4x b64 at an odd dword",
                                "A 2x b128, B 4x read2_b32 of adjacent dwords at an odd dword",
                                "2x4 tile: A 2x b128, B 4x b128 aligned (4 adjacent columns)", "2x4 tile: A 2x b128, B 4x b128 at 16B+4",
                                "2x4 tile: A 2x b128, B 8x read2_b32 adjacent at an odd dword"};
        for (int bpc = 1; bpc <= 3; ++bpc) {
            Result r[8];
            r[0] = run_lds<0>(d_in, d_out, d_st, bpc, IT); r[1] = run_lds<1>(d_in, d_out, d_st, bpc, IT);
            r[2] = run_lds<2>(d_in, d_out, d_st, bpc, IT); r[3] = run_lds<3>(d_in, d_out, d_st, bpc, IT);
            r[4] = run_lds<4>(d_in, d_out, d_st, bpc, IT);
            r[5] = run_lds<5>(d_in, d_out, d_st, bpc, IT / 2); r[6] = run_lds<6>(d_in, d_out, d_st, bpc, IT / 2);
            r[7] = run_lds<7>(d_in, d_out, d_st, bpc, IT / 2);
            for (int i = 0; i < 8; ++i)
                printf("{\"probe\": \"lds_read_shape\", \"shape\": %d, \"reads\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n",
                       i, names[i], bpc, r[i].ms, r[i].tflops, r[i].clock_ghz);
            fflush(stdout);
        }
        return 0;
    }
    for (int bpc = 1; bpc <= 3; ++bpc) {
        Result r;
        r = run<0>(d_in, d_out, d_st, 256, bpc, IT, 0, F, 4);
        printf("{\"probe\": \"mfma32x32x2_regs\", \"data\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, bpc, r.ms, r.tflops, r.clock_ghz);
        r = run<1>(d_in, d_out, d_st, 256, bpc, IT, 0, F, 4);
        printf("{\"probe\": \"mfma16x16x4_regs\", \"data\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, bpc, r.ms, r.tflops, r.clock_ghz);
        r = run<2>(d_in, d_out, d_st, 256, bpc, IT, 0, F, 4);
        printf("{\"probe\": \"mfma32x32x2_lds\", \"data\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, bpc, r.ms, r.tflops, r.clock_ghz);
        r = run<3>(d_in, d_out, d_st, 256, bpc, IT, 0, F, 4);
        printf("{\"probe\": \"mfma16x16x4_lds\", \"data\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, bpc, r.ms, r.tflops, r.clock_ghz);
        fflush(stdout);
    }
    // co-resident VALU: one 8-wave block per CU = 2 waves per SIMD: one MFMA wave + one VALU wave
    for (int nv : {0, 16, 32, 64, 128}) {
        Result r = run<4>(d_in, d_out, d_st, 512, 1, IT, nv, F, 4);
        printf("{\"probe\": \"mfma_wave_plus_valu_wave\", \"data\": \"%s\", \"valu_per_8_mfma\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, nv, r.ms, r.tflops, r.clock_ghz);
        fflush(stdout);
    }
    for (int nv : {16, 32, 64, 128}) {
        Result r = run<6>(d_in, d_out, d_st, 512, 1, IT, nv, 0.0, 4);
        printf("{\"probe\": \"valu_wave_alone\", \"data\": \"%s\", \"valu_per_iter\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f}\n", data, nv, r.ms, r.clock_ghz);
        fflush(stdout);
    }
    for (int nv : {0, 8, 16, 32, 64, 128}) {
        Result r = run<5>(d_in, d_out, d_st, 256, 1, IT, nv, F, 4);
        printf("{\"probe\": \"valu_in_mfma_stream\", \"data\": \"%s\", \"valu_per_8_mfma\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"clock_ghz\": %.3f}\n", data, nv, r.ms, r.tflops, r.clock_ghz);
        fflush(stdout);
    }
    return 0;
}
