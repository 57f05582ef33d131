#!/bin/bash
# ab_dirs.sh NAME...: decode tok/s of the headline command with the library pair in tools/_ab/NAME/
# (libkx_hip.so + libkx_host.so; the host library loads the kernel library from its own directory),
# the variants interleaved three times on one box; then the GEMV micro-benchmark of each
B="--no-cpu-baseline --no-prefill-kernels --steps 256 $KX_AB_BENCH_ARGS"
show='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "tok/s", round(d["roofline"]["avg_launch_us"],2), "us/gemv")'
for rep in 1 2 3; do
  for n in "$@"; do
    echo -n "$n: "
    KX_LIB_DIR=$PWD/tools/_ab/$n timeout -k 10 150 python bench.py $B 2>/dev/null | python -c "$show" || exit 1
  done
done
for n in "$@"; do
  echo "== $n"
  KX_LIB_DIR=$PWD/tools/_ab/$n timeout -k 10 100 python -u tools/bench_gemv.py 8b 2>&1 | tail -7
done
