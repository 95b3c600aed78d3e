// conv_k1k3.hip -- 1x1 convs (ResidualUnit second conv, predictor linears, proj_down/up) and the k3 encoder tail.
#include "conv_dispatch.hpp"
namespace mvq {
hipError_t launch_conv_k1k3(const ConvArgs& a, int ks, int bm, hipStream_t s)
{
    if (ks == 1) {
        if (bm != 96 && conv_prefer_small_tiles(a)) return launch_conv1d_mfma<1, 1, 1, 32, 1, 1, 2, 2, 0>(a, s);
        if (bm == 128) {
            const int tail = a.name_out ? 0 : conv_tail_width(a);
            if (tail) {                                              // full 128-column tiles, then the narrow tail tile
                ConvArgs m = a, t = a;
                m.n_tiles_max = a.Ncols / 128;
                t.n_base = m.n_tiles_max * 128;
                hipError_t e = launch_conv1d_mfma<1, 1, 1, 16, 2, 2, 2, 2, 0>(m, s);
                if (e != hipSuccess) return e;
                return tail == 96 ? launch_conv1d_mfma<1, 1, 1, 16, 1, 3, 4, 1, 0>(t, s)
                                  : launch_conv1d_mfma<1, 1, 1, 16, 2, 1, 2, 2, 0>(t, s);
            }
        }
        switch (bm) {
            case 128: return launch_conv1d_mfma<1, 1, 1, 16, 2, 2, 2, 2, 0>(a, s);
            case 96:  // 16-channel stages: 3 blocks per CU (+11 % on the C = 192 layer); a grid that cannot fill the CUs anyway
                      // (proj_down on a handful of tokens) takes 32-channel stages: half the trips through the K loop
                      return (long)a.B * ((a.Ncols + 127) / 128) * ((a.Mrows + 95) / 96) < 200
                                 ? launch_conv1d_mfma<1, 1, 1, 32, 3, 1, 1, 4, 0>(a, s)
                                 : launch_conv1d_mfma<1, 1, 1, 16, 3, 1, 1, 4, 0>(a, s);
            case 64:  return launch_conv1d_mfma<1, 1, 1, 16, 2, 2, 1, 4, 0>(a, s);   // 16-channel stages: 61 KB ring (LDS-DMA capable), no spills
        }
    } else if (ks == 3) {
        if (bm != 96 && conv_prefer_small_tiles(a)) return launch_conv1d_mfma<3, 1, 1, 16, 1, 1, 2, 2, 0>(a, s);
        if (bm == 128 && a.Ncols <= 96) return launch_conv1d_mfma<3, 1, 1, 8, 1, 3, 4, 1, 0>(a, s);
        switch (bm) {
            // 8-channel stages keep the 3-stage LDS-DMA ring at 50 KB (16-channel stages: 101 KB, register-staged loop only)
            case 128: return conv_dma_rows_ok(a) ? launch_conv1d_mfma<3, 1, 1, 8, 2, 2, 2, 2, 0>(a, s)
                                                 : launch_conv1d_mfma<3, 1, 1, 16, 2, 2, 2, 2, 0>(a, s);
        }
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
