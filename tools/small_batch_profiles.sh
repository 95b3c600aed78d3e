#!/bin/bash
# small_batch_profiles.sh -- the reference's own operating points: B = 1 encode (latency protocol),
# B = 6 inference, B = 6 training step; one rocprofv3 --kernel-trace --stats pass each + the plain bench lines.
#   usage (repo root, on the GPU box): bash tools/small_batch_profiles.sh <tag>
set -o pipefail
TAG=${1:-r05}
OUT=gpurun_out/${TAG}small
mkdir -p "$OUT"
echo "[1/6] B=1 encode trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b1enc" -- python3 tools/latency_trace.py enc 8 512 > "$OUT/b1enc.out" 2> "$OUT/b1enc.err" || exit 11
cp "$(ls $OUT/b1enc/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_B1_encode.csv" || exit 12
python3 tools/trace_gaps.py "$(ls $OUT/b1enc/*/*kernel_trace.csv | tail -1)" > "$OUT/gaps_B1_encode.json" || exit 13
echo "[2/6] B=1 decode trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b1dec" -- python3 tools/latency_trace.py dec 8 512 > "$OUT/b1dec.out" 2> "$OUT/b1dec.err" || exit 14
cp "$(ls $OUT/b1dec/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_B1_decode.csv" || exit 15
python3 tools/trace_gaps.py "$(ls $OUT/b1dec/*/*kernel_trace.csv | tail -1)" > "$OUT/gaps_B1_decode.json" || exit 16
echo "[3/6] B=6 inference trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b6inf" -- python3 bench.py --batch 6 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-kernel-events --no-sweep > "$OUT/bench_B6_under_rocprof.json" 2> "$OUT/b6inf.err" || exit 21
cp "$(ls $OUT/b6inf/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_B6_inference.csv" || exit 22
python3 tools/trace_gaps.py "$(ls $OUT/b6inf/*/*kernel_trace.csv | tail -1)" > "$OUT/gaps_B6_inference.json" || exit 23
echo "[4/6] B=6 training-step trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/b6train" -- python3 bench.py --workload train --batch 6 --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-kernel-events > "$OUT/bench_train_B6_under_rocprof.json" 2> "$OUT/b6train.err" || exit 31
cp "$(ls $OUT/b6train/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_B6_train.csv" || exit 32
python3 tools/trace_gaps.py "$(ls $OUT/b6train/*/*kernel_trace.csv | tail -1)" > "$OUT/gaps_B6_train.json" || exit 33
echo "[5/6] plain lines B = 1, 6, 64"
for b in 1 6 64; do
  python3 bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-latency > "$OUT/bench_B${b}.json" 2> "$OUT/bench_B${b}.err" || exit 41
done
echo "[6/6] plain B = 6 training line"
python3 bench.py --workload train --batch 6 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_train_B6.json" 2> "$OUT/bench_train_B6.err" || exit 51
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
echo done
