#!/bin/bash
# lat_sweep.sh -- A/B of the latency form (conv_lat.hip) against the LDS-tiled kernels, layer by layer, device time only.
set -o pipefail
OUT=gpurun_out/${1:-r05lat}
mkdir -p "$OUT"
export MVQ_MB_EVENTS=1 MVQ_ALLOW_TIMING_BUILD=1
for B in 1 6; do
  for L in 0 2048 8192; do
    MVQ_LAT_MAX_TILES=$L python3 tools/conv_microbench.py $B > "$OUT/mb_B${B}_lat${L}.txt" 2>&1 || exit 1
  done
done
tail -1 "$OUT"/mb_*.txt
