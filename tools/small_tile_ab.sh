#!/bin/bash
# small_tile_ab.sh -- A/B of the 64 x 64 / 128 x 128 tile threshold (conv_prefer_small_tiles) at the reference's batch of six.
#   usage (GPU box): bash tools/small_tile_ab.sh <outdir> [caps...]
OUT=${1:-gpurun_out/stab}; shift
mkdir -p "$OUT"
for cap in ${@:-160 320 480 700}; do
  MVQ_ALLOW_TIMING_BUILD=1 MVQ_SMALL_TILE_MAX=$cap timeout -k 10 200 python3 bench.py --batch 6 --steps 30 --warmup 5 --no-cpu-baseline --no-latency --no-sweep --allow-overrides > "$OUT/inf_$cap.json" 2> "$OUT/inf_$cap.err" || exit 1
  python3 -c "
import json,sys; d=json.loads(open('$OUT/inf_$cap.json').read().strip().splitlines()[-1]); print('cap $cap inference B6 ms', round(d['ms_per_step'],3), 'conv stack TF', round(d['conv_stack']['tflops'],1))"
done
