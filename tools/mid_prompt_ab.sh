#!/bin/bash
# Prompt-pass time of Llama-3-8B at short and mid-size lengths under A/B switches of the prompt pass.
# Default: the few-rows kernel capped at 64 rows (KX_PREFILL_ROWS_MAX=64: the tiled fp16 kernels take
# over above) against 128.  FUSED=1: separate SwiGLU / RoPE launches (KX_PREFILL_ROWS_UNFUSED=1)
# against the products' fused epilogues.
for n in ${ROWS:-64 72 80 96 112 128}; do
  if [ -n "$FUSED" ]; then
    echo -n "unfused  "; KX_PREFILL_ROWS_UNFUSED=1 timeout -k 10 100 python tools/bench_prefill_model.py llama-3-8b $n 3 | tail -1 || exit 1
    echo -n "fused    "; timeout -k 10 100 python tools/bench_prefill_model.py llama-3-8b $n 3 | tail -1 || exit 1
    continue
  fi
  for cap in ${CAPS:-64 128}; do
    echo -n "rows_max=$cap  "
    KX_PREFILL_ROWS_MAX=$cap timeout -k 10 100 python tools/bench_prefill_model.py llama-3-8b $n 3 | tail -1 || exit 1
  done
done
