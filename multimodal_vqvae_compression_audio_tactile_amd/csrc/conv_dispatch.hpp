// conv_dispatch.hpp -- entry points of the per-translation-unit MFMA conv instantiations.
#pragma once
#include "conv1d_mfma.hpp"

namespace mvq {
// each returns hipErrorInvalidValue when (shape, tile) has no instantiation in that unit
hipError_t launch_conv_k7(const ConvArgs& a, int dil, int bm, hipStream_t s);      // ks 7, stride 1, dil 1/3/9
hipError_t launch_conv_k1k3(const ConvArgs& a, int ks, int bm, hipStream_t s);     // ks 1 or 3, stride 1
hipError_t launch_conv_strided(const ConvArgs& a, int stride, int bm, hipStream_t s); // ks = 2*stride, stride 2/4/5/8
hipError_t launch_conv_tr(const ConvArgs& a, int bm, hipStream_t s);               // polyphase ConvTranspose1d
hipError_t launch_residual_unit_fused(const ConvArgs& a, int dil, hipStream_t s);   // C in {64, 96, 128}, dil 1/3/9
// latency form (conv_lat.hip): 16x16x4 MFMA, one wave per 16 x 16 tile; conv_lat_wanted = the launch is small enough for it
bool conv_lat_wanted(const ConvArgs& a, int ks);
hipError_t launch_conv_lat(const ConvArgs& a, int ks, int stride, int dil, hipStream_t s);          // transposed: a.up_s > 1, ks = 2
// opt-in bf16x6 / f16x3 arithmetic modes (conv_k7_bf16.hip)
struct K7Extra {              // training-config epilogues (all optional): dual output, input-gradient Snake derivative, skip gradient
    float* y2 = nullptr; const float* dsn_src = nullptr; const float* dsn_alpha = nullptr; const float* residual = nullptr;
};
hipError_t launch_bf16x3_split(const float* x, void* xs, int batch, int c, int t, hipStream_t s);
int bf16x6_tile_rows(int cout);      // 128, 96 or 0 (no tile)
hipError_t launch_bf16x3_pack_k7(const float* w, void* wq, int cout, int cin, int flip, hipStream_t s);
hipError_t launch_conv_k7_bf16x6(const void* xs, const void* wq, const float* bias, const float* alpha_out, float* y, int batch, int cin,
                                 int t, int cout, int dil, int tvalid, const K7Extra& ex, hipStream_t s);
hipError_t launch_f16x2_split(const float* x, void* xs, unsigned* xamax, int batch, int c, int t, hipStream_t s);
hipError_t launch_f16x2_pack_k7(const float* w, void* wq, unsigned* wamax, int cout, int cin, int flip, hipStream_t s);
hipError_t launch_conv_k7_f16x3(const void* xs, const unsigned* xamax, const void* wq, const unsigned* wamax, const float* bias,
                                const float* alpha_out, float* y, int batch, int cin, int t, int cout, int dil, int tvalid, const K7Extra& ex,
                                hipStream_t s);
}  // namespace mvq
