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

// sin(y)^2 with period-pi reduction and ONE even polynomial (no quadrant select): ~17 VALU ops instead of ~35 for
// sin() then squaring -- Snake is the dominant VALU cost of the conv staging loops and epilogues.
__device__ __forceinline__ float det_sin2(float y)
{
    const float n = __builtin_rintf(y * 0.318309886183790672f);
    float r = dfma(-n, 3.140625f, y);
    r = dfma(-n, 9.67502593994140625e-4f, r);
    r = dfma(-n, 1.509957990978376e-7f, r);
    const float u = r * r;
    float p = dfma(u, 2.04724070e-11f, -1.56613913e-09f);
    p = dfma(u, p, 9.39683479e-08f);
    p = dfma(u, p, -4.27555983e-06f);
    p = dfma(u, p, 1.41093474e-04f);
    p = dfma(u, p, -3.17460317e-03f);
    p = dfma(u, p, 4.44444444e-02f);
    p = dfma(u, p, -3.33333333e-01f);
    p = dfma(u, p, 1.0f);
    return u * p;
}

// inv_alpha = 1.0f / (alpha + 1e-9f), precomputed once per channel (same IEEE division everywhere)
__device__ __forceinline__ float det_snake(float x, float alpha, float inv_alpha)
{
    return dfma(inv_alpha, det_sin2(alpha * x), x);
}

// sin(y) with the period-pi reduction of det_sin2: one odd polynomial, sign from the parity of n
__device__ __forceinline__ float det_sin_pi(float y)
{
    const float n = __builtin_rintf(y * 0.318309886183790672f);
    float r = dfma(-n, 3.140625f, y);
    r = dfma(-n, 9.67502593994140625e-4f, r);
    r = dfma(-n, 1.509957990978376e-7f, r);
    const float u = r * r;
    float p = dfma(u, -7.64716373e-13f, 1.60590438e-10f);
    p = dfma(u, p, -2.50521084e-08f);
    p = dfma(u, p, 2.75573192e-06f);
    p = dfma(u, p, -1.98412698e-04f);
    p = dfma(u, p, 8.33333333e-03f);
    p = dfma(u, p, -1.66666667e-01f);
    const float s = dfma(r * u, p, r);
    return ((int)n & 1) ? -s : s;
}

// d snake(x)/dx = 1 + (alpha/(alpha+1e-9)) * sin(2*alpha*x)   (backward of Snake1d)
__device__ __forceinline__ float det_dsnake(float x, float alpha, float inv_alpha)
{
    const float ax = alpha * x;
    return dfma(alpha * inv_alpha, det_sin_pi(ax + ax), 1.0f);
}

}  // namespace mvq
