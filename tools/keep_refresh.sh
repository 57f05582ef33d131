#!/bin/bash
# keep_refresh.sh PREFIX: copies the summaries of gpurun_out/refresh/ (tools/refresh_profiles.sh) that
# are kept under version control into profiles/ as PREFIX_*
set -e
R=${1:?usage: keep_refresh.sh PREFIX}; O=gpurun_out/refresh; P=profiles
cp $O/bench.json $P/${R}_bench_line.json
cp $O/prof/b_kernel_stats.csv $P/${R}_bench_kernel_stats.csv
cp $O/traffic.json $P/${R}_gemv_traffic.json
cp $O/microbench.txt $P/${R}_microbench.txt
cp $O/prof_prefill/p_kernel_stats.csv $P/${R}_prefill_kernel_stats.csv
cp $O/prefill_mfma.json $P/${R}_prefill_mfma_counters.json
cp $O/tp.json $P/${R}_tp_rank_compute.json
cp $O/summary.txt $P/${R}_refresh_summary.txt
cp $O/short_prompts.txt $P/${R}_short_prompts.txt
cp $O/few_rows_8b.txt $P/${R}_few_rows_kernel_llama-3-8b.txt
: > $P/${R}_other_configs.jsonl
for f in b70 b1 bq bs bkv16 b8k bkv16_8k; do tail -1 $O/$f.json >> $P/${R}_other_configs.jsonl; done
