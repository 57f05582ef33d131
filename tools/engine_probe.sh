#!/bin/bash
# The chained-GEMV launch against the single launches it replaces, with the probes that say where
# its time goes: trace 1 = wait counters, trace 2 = loader alone (consumers read nothing),
# trace 3 = consumers alone (nothing loaded), variant 4 = DMA issue pace alone (no counted wait).
# Writes gpurun_out/engine_probe.txt.  Every step under its own time limit; stops at the first failure.
set -e -o pipefail
out=gpurun_out/engine_probe.txt
mkdir -p gpurun_out
: > $out
run() {  # label, env...
  echo "== $1" >> $out
  shift
  env "$@" timeout -k 10 100 python -u tools/bench_engine.py ${MODEL:-8b} 20 >> $out 2>&1
}
run "1 loader x 40 in flight, 4 consumers (the default form)"
run "the same, wait counters (timings of this block are disturbed by the counters' atomics)" KX_CHAIN_TRACE=1
run "loader alone" KX_CHAIN_TRACE=2
run "consumers alone" KX_CHAIN_TRACE=3
run "DMA issue pace alone" KX_CHAIN_TRACE=2 KX_CHAIN_VARIANT=4
run "2 loaders x 24 in flight" KX_CHAIN_FORM=4,24,2
run "2 loaders x 24 in flight, loaders alone" KX_CHAIN_FORM=4,24,2 KX_CHAIN_TRACE=2
run "1 loader x 24 in flight" KX_CHAIN_FORM=4,24,1
run "1 loader x 56 in flight" KX_CHAIN_FORM=4,56,1
