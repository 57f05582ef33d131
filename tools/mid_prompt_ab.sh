#!/bin/bash
# Prompt-pass time of Llama-3-8B at mid-size lengths with the few-rows kernel capped at 64 rows
# (KX_PREFILL_ROWS_MAX=64: the tiled fp16 kernels take over above) and at 128 rows.
for n in ${ROWS:-64 72 80 96 112 128}; do
  for cap in ${CAPS:-64 128}; do
    echo -n "rows_max=$cap  "
    KX_PREFILL_ROWS_MAX=$cap timeout -k 10 100 python tools/bench_prefill_model.py llama-3-8b $n 3 | tail -1 || exit 1
  done
done
