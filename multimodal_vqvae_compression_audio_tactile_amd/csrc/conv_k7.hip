// conv_k7.hip -- ResidualUnit 7-tap dilated convs (75 % of the path's FLOPs) + decoder input conv.
#include "conv_dispatch.hpp"
namespace mvq {
template <int DIL>
static hipError_t k7(const ConvArgs& a, int bm, hipStream_t s)
{
    if (DIL == 1 && bm == 128 && a.Ncols <= 96 && !conv_prefer_small_tiles(a))          // latent-rate layer (T = 75): 128 x 96 tile instead of 128 x 128
        return launch_conv1d_mfma<7, 1, 1, 4, 1, 3, 4, 1, 0>(a, s);
    if (bm != 96 && conv_prefer_small_tiles(a)) return launch_conv1d_mfma<7, 1, DIL, 8, 1, 1, 2, 2, 0>(a, s);
    if (bm == 128) {
        const int tail = a.name_out ? 0 : conv_tail_width(a);
        if (tail) {                                                  // full 128-column tiles, then the narrow tail tile
            ConvArgs m = a, t = a;
            m.n_tiles_max = a.Ncols / 128;
            t.n_base = m.n_tiles_max * 128;
            hipError_t e = launch_conv1d_mfma<7, 1, DIL, 4, 2, 2, 2, 2, 0>(m, s);
            if (e != hipSuccess) return e;
            return tail == 96 ? launch_conv1d_mfma<7, 1, DIL, 4, 1, 3, 4, 1, 0>(t, s)
                              : launch_conv1d_mfma<7, 1, DIL, 4, 2, 1, 2, 2, 0>(t, s);
        }
    }
    switch (bm) {
        case 128: return launch_conv1d_mfma<7, 1, DIL, 4, 2, 2, 2, 2, 0>(a, s);
        case 96:  return launch_conv1d_mfma<7, 1, DIL, 4, 3, 1, 1, 4, 0>(a, s);
        case 64:  return launch_conv1d_mfma<7, 1, DIL, 8, 2, 2, 1, 4, 0>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_conv_k7(const ConvArgs& a, int dil, int bm, hipStream_t s)
{
    switch (dil) {
        case 1: return k7<1>(a, bm, s);
        case 3: return k7<3>(a, bm, s);
        case 9: return k7<9>(a, bm, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
