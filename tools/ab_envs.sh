#!/bin/bash
# ab_envs.sh "VAR=a VAR2=b" "VAR=c" ...: decode tok/s of the headline command under each set of
# environment settings ("-" = none), interleaved three times on one box; KX_AB_BENCH_ARGS: more
# bench.py arguments (e.g. --model llama-3.2-1b)
B="--no-cpu-baseline --no-prefill-kernels --steps 256 $KX_AB_BENCH_ARGS"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "tok/s", round(d["roofline"]["avg_launch_us"],2), "us/gemv")'
for rep in 1 2 3; do
  for e in "$@"; do
    echo -n "$e: "
    if [ "$e" = "-" ]; then timeout -k 10 150 python bench.py $B 2>/dev/null | python -c "$show" || exit 1
    else env $e timeout -k 10 150 python bench.py $B 2>/dev/null | python -c "$show" || exit 1; fi
  done
done
