#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/smallp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 8 32; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$N -o p -- python3 $R/tools/bench_prefill_model.py llama-3-8b $N 6 > $O/run$N.txt 2> $O/run$N.err || { tail -3 $O/run$N.err; exit 1; }
tail -1 $O/run$N.txt
rm -f $O/prof$N/p_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof$N/p_kernel_stats.csv")))
for r in rows[:16]:
    if "fill_hash" in r["Name"]: continue
    print("%-80s calls %6s avg %9.1f us  %5s %%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
