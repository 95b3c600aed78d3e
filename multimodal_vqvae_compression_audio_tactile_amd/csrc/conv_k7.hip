// conv_k7.hip -- ResidualUnit 7-tap dilated convs (75 % of the path's FLOPs) + decoder input conv.
#include "conv_dispatch.hpp"
namespace mvq {
template <int DIL>
static hipError_t k7(const ConvArgs& a, int bm, hipStream_t s)
{
    switch (bm) {
        case 128: return launch_conv1d_mfma<7, 1, DIL, 8, 2, 2, 2, 2, false>(a, s);
        case 96:  return launch_conv1d_mfma<7, 1, DIL, 8, 3, 1, 1, 4, false>(a, s);
        case 64:  return launch_conv1d_mfma<7, 1, DIL, 8, 2, 2, 1, 4, false>(a, s);
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
