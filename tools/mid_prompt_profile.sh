#!/bin/bash
# Per-kernel view of mid-size prompt passes (Llama-3-8B): prints the pass times for a few prompt
# lengths, then rocprofv3 kernel statistics for the length given as $1 (default 128).
set -e
R=$(pwd); O=$R/gpurun_out/mid; mkdir -p $O
N=${1:-128}
for n in 64 96 128 192 256 384 512; do
  timeout -k 10 100 python tools/bench_prefill_model.py llama-3-8b $n 2 | tail -1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/tools/bench_prefill_model.py llama-3-8b $N 4 > $O/run.txt 2> $O/run.err
rm -f $O/prof/p_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof/p_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
