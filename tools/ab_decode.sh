#!/bin/bash
# ab_decode.sh NAME...: decode tok/s of the headline command with lib/ab/libkx_hip_NAME.so, the
# variants interleaved three times on one box (clock drift between runs shows as spread)
B="--no-cpu-baseline --no-prefill-kernels --steps 256"
L=$PWD/llm-inference-engine_amd/lib/ab
for rep in 1 2 3; do
  for n in "$@"; do
    echo -n "$n: "
    KX_HIP_LIB=$L/libkx_hip_$n.so timeout -k 10 120 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['roofline']['avg_launch_us'],2), 'us/gemv')" || exit 1
  done
done
