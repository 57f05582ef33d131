#!/bin/bash
# ab_rows.sh ROWS NAME...: the few-rows kernel microbench with lib/ab/libkx_hip_NAME.so, per variant
L=$PWD/llm-inference-engine_amd/lib/ab
R=$1; shift
for n in "$@"; do
  echo "== $n"
  KX_HIP_LIB=$L/libkx_hip_$n.so timeout -k 10 200 python tools/bench_rows_kernel.py llama-3-8b 60 $R || exit 1
done
