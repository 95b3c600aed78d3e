#!/bin/bash
# collect_profiles.sh -- every rocprofv3 pass the round's evidence under profiles/ comes from, in one go on the GPU box.
#   usage (from the repo root, on the box):  bash tools/collect_profiles.sh <round-tag> <commit>
# Writes everything under gpurun_out/<round-tag>prof/; the summaries to keep are copied into profiles/ by hand afterwards.
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains), the program itself directly after "--".
set -o pipefail
TAG=${1:-r04}
COMMIT=${2:-unknown}
OUT=gpurun_out/${TAG}prof
mkdir -p "$OUT"
BENCH_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-sweep"    # the trace is of the 256-segment steps only: the batch sweep launches the same
                                                                                   # instantiations at 1-64 segments and would pull their per-kernel averages down

echo "[1/8] kernel trace of the default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -- python3 bench.py $BENCH_ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/ks.err" || exit 11
cp "$(ls $OUT/ks/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_bench_default_B256.csv" || exit 12

echo "[2/8] FETCH_SIZE pass"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-latency > /dev/null 2> "$OUT/fetch.err" || exit 21
echo "[3/8] WRITE_SIZE pass"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-latency > /dev/null 2> "$OUT/write.err" || exit 31
python3 tools/pmc_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "$COMMIT" \
    rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-latency > "$OUT/pmc_traffic.txt" || exit 32
# the bench reads profiles/pmc_traffic.json: install the fresh table BEFORE the stored bench lines are produced
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json

echo "[3b] SQ counters of the codebook-search kernels (LDS instructions per global read, bank conflicts, LDS busy share)"
VQ_CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv \
    -d "$OUT/vq" -- $VQ_CMD > /dev/null 2> "$OUT/vq.err" || exit 33
python3 tools/pmc_vq.py "$OUT/vq" "$OUT/pmc_vq_lds.json" "$COMMIT" rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE -- $VQ_CMD > "$OUT/pmc_vq_lds.txt" || exit 34
cp "$OUT/pmc_vq_lds.json" profiles/pmc_vq_lds.json

echo "[4/8] SQ counters: dominant k7 kernel (dec.b0, d = 3) and one fused unit (enc.b1, d = 3)"
for L in dec.b0.k7d3 enc.b1.RUd3; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv \
        -d "$OUT/sq_a_$L" -- python3 tools/conv_microbench.py 256 $L > "$OUT/sq_a_$L.txt" 2> "$OUT/sq_a_$L.err" || exit 41
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv \
        -d "$OUT/sq_b_$L" -- python3 tools/conv_microbench.py 256 $L > "$OUT/sq_b_$L.txt" 2> "$OUT/sq_b_$L.err" || exit 42
done
python3 tools/pmc_sum.py "$OUT/pmc_k7_dec_b0_d3.json" "$OUT/sq_a_dec.b0.k7d3" "$OUT/sq_b_dec.b0.k7d3" || exit 43
python3 tools/pmc_sum.py "$OUT/pmc_ru_enc_b1_d3.json" "$OUT/sq_a_enc.b1.RUd3" "$OUT/sq_b_enc.b1.RUd3" || exit 44
python3 tools/pmc_wave.py "$OUT/sq_a_dec.b0.k7d3" "$OUT/sq_b_dec.b0.k7d3" "$OUT/sq_a_enc.b1.RUd3" "$OUT/sq_b_enc.b1.RUd3" > "$OUT/pmc_wave.txt" || exit 45

echo "[5/8] kernel trace of the training step (B = 256)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_train" -- python3 bench.py --workload train $BENCH_ARGS > "$OUT/bench_train_under_rocprof.json" 2> "$OUT/ks_train.err" || exit 51
cp "$(ls $OUT/ks_train/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_train_B256.csv" || exit 52

echo "[6/8] stored bench lines (after the traffic table: kernels_unchanged_since_profile must read true)"
python3 bench.py --steps 5 --warmup 2 > "$OUT/bench_default_B256.json" 2> "$OUT/bench_default.err" || exit 61
python3 bench.py --workload train --steps 5 --warmup 3 > "$OUT/bench_train_B256.json" 2> "$OUT/bench_train.err" || exit 62
# the reference's own training batch (Training/compare_dacvsproposal_5.py:62), with its CPU baseline
python3 bench.py --workload train --batch 6 --steps 10 --warmup 3 > "$OUT/bench_train_B6.json" 2> "$OUT/bench_train_B6.err" || exit 63

echo "[7/8] latency + corpus"
python3 tools/latency.py > "$OUT/latency_B1.json" 2> "$OUT/latency.err" || exit 71
python3 tools/corpus_eval.py > "$OUT/corpus_eval_1gpu.json" 2> "$OUT/corpus.err" || exit 72

echo "[8/8] RCCL path in a one-rank group (probe all-reduce, gradient all-reduce, token all-gather, metric gather)"
python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 1 bench.py --gpus 1 --batch 32 --workload train \
    --force-collectives --no-cpu-baseline --steps 3 --warmup 2 > "$OUT/bench_train_1rank_rccl.json" 2> "$OUT/rccl_train.err" || exit 81
python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 1 tools/corpus_eval.py --gpus 1 --clips 100 \
    --force-collectives > "$OUT/corpus_eval_1rank_rccl.json" 2> "$OUT/rccl_corpus.err" || exit 82
echo "[9] multi-rank plumbing on one device: two-rank gloo rehearsal through the bare self-launch path, and the fatal RCCL path"
MVQ_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_2rank_one_device_rehearsal.json" 2> "$OUT/rehearsal.err" || exit 91
MVQ_BENCH_ONE_DEVICE=1 MVQ_BENCH_REHEARSE_RCCL_FAILURE=1 timeout -k 10 120 python3 bench.py --gpus 2 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/rccl_failure_rehearsal.out" 2> "$OUT/rccl_failure_rehearsal.err"
echo "failure rehearsal exit code: $?" >> "$OUT/rccl_failure_rehearsal.err"
echo "[10] opt-in arithmetic modes (NON-PARITY; DESIGN.md section 6d): their own bench lines next to the default one of step 6 (same box), kernel trace of f16x3"
for M in bf16x6 f16x3; do
    python3 bench.py --arith $M --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_arith_$M.json" 2> "$OUT/bench_arith_$M.err" || exit 101
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks_f16x3" -- python3 bench.py --arith f16x3 $BENCH_ARGS > "$OUT/bench_arith_f16x3_under_rocprof.json" 2> "$OUT/ks_f16x3.err" || exit 102
cp "$(ls $OUT/ks_f16x3/*/*kernel_stats.csv | tail -1)" "$OUT/kernel_stats_bench_arith_f16x3_B256.csv" || exit 103
python3 tools/bf16x6_bench.py 256 > "$OUT/arith_layers_B256.txt" 2> "$OUT/arith_layers.err" || exit 104
for M in bf16x6 f16x3; do
    python3 bench.py --workload train --arith $M --steps 5 --warmup 3 --no-cpu-baseline > "$OUT/bench_train_arith_$M.json" 2> "$OUT/bench_train_arith_$M.err" || exit 105
done
for M in bf16x6 f16x3; do
    python3 tools/corpus_eval.py --compare-arith $M > "$OUT/corpus_arith_comparison_$M.json" 2> "$OUT/corpus_cmp_$M.err" || exit 106
done
python3 tools/corpus_eval.py --arith f16x3 > "$OUT/corpus_eval_1gpu_f16x3.json" 2> "$OUT/corpus_f16x3.err" || exit 107
echo done
