// det_math.hpp -- device-side deterministic fp32 elementary functions for gfx950.
//
// Snake (sin), softmax (exp), tanh and GELU (erf) on the hot path are written as fixed sequences of
// correctly rounded IEEE-754 binary32 operations (v_fma_f32, v_mul_f32, v_add_f32, v_rndne_f32,
// IEEE division / sqrt) instead of ocml's sinf/expf/..., so that results do not depend on the math
// library build and can be reproduced exactly by a CPU checker.  Compile with -ffp-contract=off:
// every fused multiply-add below is explicit.
//
// Reference semantics these implement: upstream DAC snake() = x + (alpha+1e-9)^-1 * sin(alpha*x)^2
// (called inside every dac.DAC conv block the reference runs, Training/compare_dacvsproposal_5.py:294-296,322);
// torch.tanh(TokenNorm(r)) (...:313); softmax and nn.GELU() in CrossPredictor (...:229,241).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvq {

__device__ __forceinline__ float dfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float det_sin(float x)
{
    const float n = __builtin_rintf(x * 0.636619772367581343f);
    float r = dfma(-n, 1.5703125f, x);
    r = dfma(-n, 4.837512969970703125e-4f, r);
    r = dfma(-n, 7.549789948768648e-8f, r);
    const int q = (int)n;
    const float r2 = r * r;
    float ps = dfma(r2, 2.75573137e-06f, -1.98412698e-04f);
    ps = dfma(r2, ps, 8.33333333e-03f);
    ps = dfma(r2, ps, -1.66666667e-01f);
    const float s = dfma(r * r2, ps, r);
    float pc = dfma(r2, -2.75573144e-07f, 2.48015873e-05f);
    pc = dfma(r2, pc, -1.38888889e-03f);
    pc = dfma(r2, pc, 4.16666667e-02f);
    const float c = dfma(r2 * r2, pc, dfma(r2, -0.5f, 1.0f));
    const float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

__device__ __forceinline__ float det_exp_poly(float r)   // exp(r) - 1, |r| <= ln2/2
{
    float p = dfma(r, 1.98412698e-04f, 1.38888889e-03f);
    p = dfma(r, p, 8.33333333e-03f);
    p = dfma(r, p, 4.16666667e-02f);
    p = dfma(r, p, 1.66666667e-01f);
    p = dfma(r, p, 0.5f);
    return dfma(r * r, p, r);
}

__device__ __forceinline__ float det_exp(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = dfma(-n, 0.693145751953125f, x);
    r = dfma(-n, 1.42860682030941723e-6f, r);
    const float e = 1.0f + det_exp_poly(r);
    const int ni = (int)n;
    const float scale = __builtin_bit_cast(float, (uint32_t)(ni + 127) << 23);
    return e * scale;
}

__device__ __forceinline__ float det_tanh(float x)
{
    const float a = __builtin_fabsf(x);
    float res;
    if (a < 0.17f) {
        const float em1 = det_exp_poly(a + a);
        res = em1 / (em1 + 2.0f);
    } else if (a > 10.0f) {
        res = 1.0f;
    } else {
        const float t = det_exp(-(a + a));
        res = (1.0f - t) / (1.0f + t);
    }
    return __builtin_copysignf(res, x);
}

__device__ __forceinline__ float det_erf(float x)
{
    const float a = __builtin_fabsf(x);
    const float t = 1.0f / dfma(0.3275911f, a, 1.0f);
    float p = dfma(t, 1.061405429f, -1.453152027f);
    p = dfma(t, p, 1.421413741f);
    p = dfma(t, p, -0.284496736f);
    p = dfma(t, p, 0.254829592f);
    p = p * t;
    const float e = det_exp(-(a * a));
    const float res = dfma(-p, e, 1.0f);
    return __builtin_copysignf(res, x);
}

__device__ __forceinline__ float det_gelu(float x)
{
    return (0.5f * x) * (1.0f + det_erf(x * 0.707106781186547524f));
}

// Snake needs sin(alpha*x)^2.  Two things make it cheap (on this chip every VALU instruction is time the fp32 MFMAs do not
// get: tools/mfma_probe.hip):
//  * TURNS: t = x * (alpha/pi) is the phase in half-periods, n = rint(t), f = t - n is EXACT in fp32 (|f| <= 0.5, both
//    operands multiples of ulp(t)) -- no Cody-Waite chain;
//  * HALF ANGLE: w = sin(pi f / 2)^2 = u*(c1 + c2 u + ... + c5 u^4), u = f^2, needs 5 terms on |pi f / 2| <= pi/4, and
//    sin(pi f)^2 = 4 w (1 - w); the truncation error of w is multiplied by 4(1 - 2w), which vanishes at the end of the range.
// 11 operations per Snake (mul, rint, sub, mul, 4 fma, mul, fma, fma) instead of 18 in round 2; 1.1e-7 abs against libm over
// |t| <= 8 (round 2: 1.7e-7).  The phase rounds twice (alpha/pi, then the product with x) where torch's sin(alpha*x) rounds
// once: the same order of error (~1e-7 * |alpha x|) as the reference's own fp32 product, far inside every fixture tolerance.
__device__ __forceinline__ float det_sin2q_turns(float t)        // sin(pi t)^2 / 4
{
    const float n = __builtin_rintf(t);
    const float f = t - n;
    const float u = f * f;
    float p = dfma(u, 0.012903445400297642f, -0.11766531318426132f);
    p = dfma(u, p, 0.6676313877105713f);
    p = dfma(u, p, -2.029356002807617f);
    p = dfma(u, p, 2.4674010276794434f);
    const float w = u * p;
    return dfma(-w, w, w);                                        // w - w^2
}

__device__ __forceinline__ float det_sin2_turns(float t) { return 4.0f * det_sin2q_turns(t); }

// inv_alpha = 1.0f / (alpha + 1e-9f), precomputed once per channel (same IEEE division everywhere).  4 * inv_alpha is exact,
// so fma(4 inv, q, x) == fma(inv, 4 q, x) = x + inv * sin^2; alpha * (1/pi) and 4 * inv_alpha are per-row values the compiler
// hoists out of the per-element loops.
__device__ __forceinline__ float det_snake(float x, float alpha, float inv_alpha)
{
    return dfma(4.0f * inv_alpha, det_sin2q_turns(x * (alpha * 0.318309886183790672f)), x);
}

// sin(pi*t): n = rint(t), f = t - n exact, one odd polynomial f*(d0 + d1 v + ... + d7 v^7), sign from the parity of n
__device__ __forceinline__ float det_sin_turns(float t)
{
    const float n = __builtin_rintf(t);
    const float f = t - n;
    const float v = f * f;
    float p = dfma(v, -2.191535349993501e-05f, 0.0004663027939386666f);
    p = dfma(v, p, -0.00737043097615242f);
    p = dfma(v, p, 0.08214588463306427f);
    p = dfma(v, p, -0.5992645025253296f);
    p = dfma(v, p, 2.550163984298706f);
    p = dfma(v, p, -5.167712688446045f);
    p = dfma(v, p, 3.1415927410125732f);
    const float s = f * p;
    // parity of n without converting an unbounded float to int (that conversion saturates on the device and is undefined
    // behaviour on the host for |n| >= 2^31): n is odd exactly when n/2 is not an integer; every |n| >= 2^24 is even
    const float hn = 0.5f * n;
    return (hn != __builtin_rintf(hn)) ? -s : s;
}

// d snake(x)/dx = 1 + (alpha/(alpha+1e-9)) * sin(2*alpha*x)   (backward of Snake1d); 2*alpha*x = pi * (2t)
__device__ __forceinline__ float det_dsnake(float x, float alpha, float inv_alpha)
{
    const float t = x * (alpha * 0.318309886183790672f);
    return dfma(alpha * inv_alpha, det_sin_turns(t + t), 1.0f);
}

}  // namespace mvq
