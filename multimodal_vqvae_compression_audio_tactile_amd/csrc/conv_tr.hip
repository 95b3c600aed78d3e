// conv_tr.hip -- DecoderBlock ConvTranspose1d (kernel 2*s, stride s) in polyphase form: a 2-tap conv
// over Cout*s GEMM rows with a pixel-shuffle store.
#include "conv_dispatch.hpp"
// 16 channels x 2 taps per LDS stage: 49 KB of staging per block (two to three blocks per CU).  With 32-channel stages
// (98 KB, one block per CU) these layers ran at 77-101 TFLOP/s instead of 97-118.
namespace mvq {
hipError_t launch_conv_tr(const ConvArgs& a, int bm, hipStream_t s)
{
    if (bm == 128 && conv_prefer_small_tiles(a)) {
        switch (a.up_s) {
            case 8: return launch_conv1d_mfma<2, 1, 1, 32, 1, 1, 2, 2, 8>(a, s);
            case 5: return launch_conv1d_mfma<2, 1, 1, 32, 1, 1, 2, 2, 5>(a, s);
            case 4: return launch_conv1d_mfma<2, 1, 1, 32, 1, 1, 2, 2, 4>(a, s);
            case 2: return launch_conv1d_mfma<2, 1, 1, 32, 1, 1, 2, 2, 2>(a, s);
        }
    }
    // 16 channels x 2 taps per stage for the register-staged loop; with LDS-DMA staging (3-stage ring) 8 channels per stage
    // keep the ring at 37 KB, i.e. three blocks per CU
    const bool dma = conv_dma_rows_ok(a);
    const int tail = (a.name_out || bm != 128) ? 0 : conv_tail_width(a);
    if (tail && (a.up_s == 4 || a.up_s == 5)) {                     // column split (conv1d_mfma.hpp, conv_tail_width)
        ConvArgs m = a, t = a;
        m.n_tiles_max = a.Ncols / 128;
        t.n_base = m.n_tiles_max * 128;
        hipError_t e;
        if (a.up_s == 4) e = dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 4>(m, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 4>(m, s);
        else e = dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 5>(m, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 5>(m, s);
        if (e != hipSuccess) return e;
        if (a.up_s == 4) return tail == 96 ? launch_conv1d_mfma<2, 1, 1, 8, 1, 3, 4, 1, 4>(t, s) : launch_conv1d_mfma<2, 1, 1, 8, 2, 1, 2, 2, 4>(t, s);
        return tail == 96 ? launch_conv1d_mfma<2, 1, 1, 8, 1, 3, 4, 1, 5>(t, s) : launch_conv1d_mfma<2, 1, 1, 8, 2, 1, 2, 2, 5>(t, s);
    }
    const bool narrow = bm == 128 && a.Ncols <= 96;
    switch (a.up_s) {
        case 8: if (bm == 128 && narrow) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 1, 3, 4, 1, 8>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 1, 3, 4, 1, 8>(a, s);
                if (bm == 128) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 8>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 8>(a, s);
                break;
        case 5: if (bm == 128) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 5>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 5>(a, s);
                break;
        case 4: if (bm == 128) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 4>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 4>(a, s);
                break;
        case 2: if (bm == 96) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 3, 1, 1, 4, 2>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 3, 1, 1, 4, 2>(a, s);
                if (bm == 128) return dma ? launch_conv1d_mfma<2, 1, 1, 8, 2, 2, 2, 2, 2>(a, s) : launch_conv1d_mfma<2, 1, 1, 16, 2, 2, 2, 2, 2>(a, s);
                break;
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
