#!/bin/bash
# decode tok/s of the headline command under a few runtime settings (A/B on one box)
B="--no-cpu-baseline --no-prefill-kernels --steps 256"
run() { echo "== $1"; env $1 timeout -k 10 120 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['roofline']['avg_launch_us'],2), 'us/gemv', d['config']['graph_replay'])"; }
run "KX_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "KX_NOP=2"
