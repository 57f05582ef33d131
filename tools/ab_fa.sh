#!/bin/bash
# ab_fa.sh NAME...: times tools/bench_fa.py on lib/ab/libkx_hip_NAME.so for each name, two rounds
# (interleaved, so clock drift between boxes and over time shows up as spread, not as a winner)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for n in "$@"; do
    KX_HIP_LIB=$PWD/llm-inference-engine_amd/lib/ab/libkx_hip_$n.so timeout -k 10 100 python tools/bench_fa.py || exit 1
  done
done
