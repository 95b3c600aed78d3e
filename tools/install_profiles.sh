#!/bin/bash
# install_profiles.sh -- copy the summaries of one tools/collect_profiles.sh run (gpurun_out/<tag>prof/) into profiles/ under the
# round's names.  usage: bash tools/install_profiles.sh [tag]
set -e
TAG=${1:-r04}
O=gpurun_out/${TAG}prof
for f in bench_default_B256.json bench_train_B256.json latency_B1.json corpus_eval_1gpu.json corpus_eval_1rank_rccl.json bench_train_1rank_rccl.json \
         bench_2rank_one_device_rehearsal.json kernel_stats_bench_default_B256.csv kernel_stats_train_B256.csv pmc_k7_dec_b0_d3.json \
         pmc_ru_enc_b1_d3.json pmc_wave.txt pmc_vq_lds.json rccl_failure_rehearsal.err bench_arith_bf16x6.json bench_arith_f16x3.json \
         bench_arith_f16x3_under_rocprof.json kernel_stats_bench_arith_f16x3_B256.csv bench_train_arith_bf16x6.json bench_train_arith_f16x3.json \
         corpus_arith_comparison_bf16x6.json corpus_arith_comparison_f16x3.json corpus_eval_1gpu_f16x3.json; do
    cp "$O/$f" "profiles/${TAG}_$f"
done
cp "$O/bench_under_rocprof.json" "profiles/${TAG}_bench_default_B256_under_rocprof.json"
cp "$O/bench_train_under_rocprof.json" "profiles/${TAG}_bench_train_B256_under_rocprof.json"
cp "$O/arith_layers_B256.txt" "profiles/${TAG}_arith_layers_B256.txt"
cp "$O/pmc_traffic.json" profiles/pmc_traffic.json
cp "$O/pmc_vq_lds.json" profiles/pmc_vq_lds.json
python3 - "$TAG" <<'PY'
import json, csv, sys
tag = sys.argv[1]
def L(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = L(f'profiles/{tag}_bench_default_B256.json')
print("default", round(d['ms_per_step'], 2), "ms", round(d['value']), d['unit'], "path", round(d['path_tflops'], 2), "TFLOP/s; dominant", round(d['roofline']['achieved'], 1),
      round(d['roofline']['frac'], 3), "HIP us", round(d['roofline']['avg_launch_us']), "conv stack", round(d['conv_stack']['frac_of_fp32_mfma_peak'], 3),
      "no-events", round(d.get('ms_per_step_without_kernel_events'), 2), "spot", d['parity_spot_check'], "flags", d['build_flags'],
      "stamp ok", d['conv_stack']['hbm']['source']['kernels_unchanged_since_profile'], "cpu seg/s", round(d['cpu_baseline']['segments_per_s'], 2))
rows = list(csv.DictReader(open(f'profiles/{tag}_kernel_stats_bench_default_B256.csv'))); print("rocprof dominant avg us", round(float(rows[0]['AverageNs']) / 1e3), rows[0]['Name'][:64])
for m in ('bf16x6', 'f16x3'):
    a = L(f'profiles/{tag}_bench_arith_{m}.json'); t = L(f'profiles/{tag}_bench_train_arith_{m}.json'); c = L(f'profiles/{tag}_corpus_arith_comparison_{m}.json')['arith_comparison']
    print(m, round(a['ms_per_step'], 2), "ms", round(a['value']), "codes/idx equal", a['arith_check']['audio_codes_equal_fraction'], a['arith_check']['rvq_indices_equal_fraction'],
          "psnr delta", a['arith_check']['recon_psnr_delta_db_max'], "| train", round(t['ms_per_step'], 1), "| corpus equal", round(c['index_equal_fraction'], 7), "flip segs", c['segments_with_a_flip'],
          "max margin/scale", c['first_flip_margin_over_scale_max'])
t = L(f'profiles/{tag}_bench_train_B256.json'); print("train", round(t['ms_per_step'], 1), "ms peak GB", round(t['device_memory']['peak_allocated_GB'], 1), "conv stack", round(t['conv_stack']['frac_of_fp32_mfma_peak'], 3))
c = L(f'profiles/{tag}_corpus_eval_1gpu.json'); c2 = L(f'profiles/{tag}_corpus_eval_1gpu_f16x3.json')
print("corpus seg/s", round(c['segments_per_s_forward']), round(c['segments_per_s_with_metrics']), "| f16x3", round(c2['segments_per_s_forward']), round(c2['segments_per_s_with_metrics']))
print("traffic table commit", json.load(open('profiles/pmc_traffic.json'))['_meta']['commit'])
PY
