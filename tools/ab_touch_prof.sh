#!/bin/bash
# per-kernel averages of the token loop with and without the next-launch touch of the attention
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in on off; do
  rm -rf /tmp/prof_$v
  if [ $v = off ]; then export KX_ATTN_NO_TOUCH=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o run -- python3 $R/bench.py --no-cpu-baseline --no-prefill-kernels --steps 128 > /tmp/prof_$v.log 2> /tmp/prof_$v.err || { tail -5 /tmp/prof_$v.err; exit 1; }
  echo "== touch $v"
  python3 -c "
import csv,sys
for r in csv.DictReader(open('/tmp/prof_$v/run_kernel_stats.csv')):
    if 'gemv_w8_kernel' in r['Name'] or 'attn_decode' in r['Name']:
        print('%-60s calls %6s avg %9.1f ns' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])))
"
done
