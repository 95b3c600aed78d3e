// conv_tr.hip -- DecoderBlock ConvTranspose1d (kernel 2*s, stride s) in polyphase form: a 2-tap conv
// over Cout*s GEMM rows with a pixel-shuffle store.
#include "conv_dispatch.hpp"
namespace mvq {
hipError_t launch_conv_tr(const ConvArgs& a, int bm, hipStream_t s)
{
    switch (bm) {
        case 128: return launch_conv1d_mfma<2, 1, 1, 32, 2, 2, 2, 2, true>(a, s);
        case 96:  return launch_conv1d_mfma<2, 1, 1, 32, 3, 1, 1, 4, true>(a, s);
        case 64:  return launch_conv1d_mfma<2, 1, 1, 32, 2, 2, 1, 4, true>(a, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
