// conv_ru.hip -- whole ResidualUnit in one launch for the narrow ends of the stacks (C = 64, 96, 128), where the
// block tile covers every channel: conv7(dil) -> Snake -> conv1 -> + skip without the intermediate leaving the CU.
#include "conv_dispatch.hpp"
namespace mvq {
template <int DIL>
static hipError_t ru(const ConvArgs& a, int c, hipStream_t s)
{
    // Latency regime (the wide-tile grid would leave most CUs idle: one segment, or a few): half-width time tiles -- twice the
    // blocks, each with half the MFMA chain.  (At full batch the wide tiles win: more operand reuse per LDS read.)
    const long wide_blocks = (long)a.B * ((a.Ncols + (c == 64 ? 191 : 127)) / (c == 64 ? 192 : 128));     // in 128/192-column units
    const bool narrow = wide_blocks < 200 && !a.name_out;
    switch (c) {
        // C = 128: 128 x 96 tiles -- the 51 KB intermediate tile lets three blocks share a CU (128 x 128: 68 KB, two blocks);
        // measured 121.4 vs 119.3 TFLOP/s, and T = 12 000 is 125 such tiles exactly
        case 128: return narrow ? launch_residual_unit<DIL, 4, 2, 1, 2, 2>(a, s) : launch_residual_unit<DIL, 4, 1, 3, 4, 1>(a, s);
        case 96:  return launch_residual_unit<DIL, 4, 3, 1, 1, 4>(a, s);
        // C = 64: 64 x 192 tiles (wave tile 32 x 96, four-channel stages): 51.5 KB per block -> three blocks per CU, and T = 24 000
        // is 125 such tiles exactly.  Round 4, same box, Snake-on-load form: 119.5 vs 114.5 TFLOP/s for the 64 x 256 tile
        // (66 KB intermediate tile, two blocks per CU); profiles/r04_timing_experiments.json
        case 64:  return narrow ? launch_residual_unit<DIL, 8, 2, 1, 1, 4>(a, s) : launch_residual_unit<DIL, 4, 1, 3, 2, 2>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_residual_unit_fused(const ConvArgs& a, int dil, hipStream_t s)
{
    switch (dil) {
        case 1: return ru<1>(a, a.Cout, s);
        case 3: return ru<3>(a, a.Cout, s);
        case 9: return ru<9>(a, a.Cout, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace mvq
