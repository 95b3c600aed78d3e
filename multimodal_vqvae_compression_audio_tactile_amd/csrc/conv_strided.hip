// conv_strided.hip -- EncoderBlock down-sampling convs: kernel 2*s, stride s, s in {2,4,5,8}.
#include "conv_dispatch.hpp"
// Chunk sizes (channels per LDS stage) are chosen so that a block's double-buffered stage stays near 50 KB: two to three
// blocks share a CU's 160 KB.  With K chunks of 64 (98 KB per block, ONE block per CU) these layers ran at 84-100 TFLOP/s.
namespace mvq {
hipError_t launch_conv_strided(const ConvArgs& a, int stride, int bm, hipStream_t s)
{
    // With LDS-DMA staging (3-stage ring) the stages are halved where needed so that the ring still fits three blocks per CU.
    const bool dma = conv_dma_rows_ok(a);
    // 96-row tiles: only the input-gradient of the last DecoderBlock's ConvTranspose1d (96 <- 192 channels, stride 2)
    if (bm == 96 && stride == 2) return dma ? launch_conv1d_mfma<4, 2, 1, 4, 3, 1, 1, 4, 0>(a, s) : launch_conv1d_mfma<4, 2, 1, 8, 3, 1, 1, 4, 0>(a, s);
    if (bm != 128) return hipErrorInvalidValue;
    if (conv_prefer_small_tiles(a)) {
        switch (stride) {
            case 2: return launch_conv1d_mfma<4, 2, 1, 16, 1, 1, 2, 2, 0>(a, s);
            case 4: return launch_conv1d_mfma<8, 4, 1, 8, 1, 1, 2, 2, 0>(a, s);
            case 5: return launch_conv1d_mfma<10, 5, 1, 4, 1, 1, 2, 2, 0>(a, s);
            case 8: return launch_conv1d_mfma<16, 8, 1, 4, 1, 1, 2, 2, 0>(a, s);
        }
    }
    const int tail = a.name_out ? 0 : conv_tail_width(a);
    if (tail && (stride == 4 || stride == 5)) {                     // column split (conv1d_mfma.hpp, conv_tail_width)
        ConvArgs m = a, t = a;
        m.n_tiles_max = a.Ncols / 128;
        t.n_base = m.n_tiles_max * 128;
        hipError_t e = stride == 4 ? (dma ? launch_conv1d_mfma<8, 4, 1, 2, 2, 2, 2, 2, 0>(m, s) : launch_conv1d_mfma<8, 4, 1, 4, 2, 2, 2, 2, 0>(m, s))
                                   : launch_conv1d_mfma<10, 5, 1, 2, 2, 2, 2, 2, 0>(m, s);
        if (e != hipSuccess) return e;
        if (stride == 4) return tail == 96 ? launch_conv1d_mfma<8, 4, 1, 2, 1, 3, 4, 1, 0>(t, s) : launch_conv1d_mfma<8, 4, 1, 2, 2, 1, 2, 2, 0>(t, s);
        return tail == 96 ? launch_conv1d_mfma<10, 5, 1, 2, 1, 3, 4, 1, 0>(t, s) : launch_conv1d_mfma<10, 5, 1, 2, 2, 1, 2, 2, 0>(t, s);
    }
    switch (stride) {
        case 2: return dma ? launch_conv1d_mfma<4, 2, 1, 4, 2, 2, 2, 2, 0>(a, s) : launch_conv1d_mfma<4, 2, 1, 8, 2, 2, 2, 2, 0>(a, s);
        case 4: return dma ? launch_conv1d_mfma<8, 4, 1, 2, 2, 2, 2, 2, 0>(a, s) : launch_conv1d_mfma<8, 4, 1, 4, 2, 2, 2, 2, 0>(a, s);
        case 5: return launch_conv1d_mfma<10, 5, 1, 2, 2, 2, 2, 2, 0>(a, s);
        case 8: if (a.Ncols <= 96) return dma ? launch_conv1d_mfma<16, 8, 1, 1, 1, 3, 4, 1, 0>(a, s) : launch_conv1d_mfma<16, 8, 1, 2, 1, 3, 4, 1, 0>(a, s);
                return dma ? launch_conv1d_mfma<16, 8, 1, 1, 2, 2, 2, 2, 0>(a, s) : launch_conv1d_mfma<16, 8, 1, 2, 2, 2, 2, 2, 0>(a, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
