#!/bin/bash
# Runs on the GPU box, from the repository root: the GPU test suite and both randomised sweeps
# against the bounds-checking build (make -C llm-inference-engine_amd BOUNDS=1, built beforehand).
# Every run must end with an empty violation record; the reports land in gpurun_out/bounds/.
set -e
R=$(pwd); O=$R/gpurun_out/bounds; mkdir -p $O
export KX_LIB_DIR=$R/llm-inference-engine_amd/lib_bounds
timeout -k 10 ${1:-900} python -u -m pytest tests -m gpu -q --timeout 600 --timeout-method thread \
  --deselect tests/test_zy_gpu_peer.py -p no:cacheprovider > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
cp gpurun_out/bounds_report.json $O/pytest_bounds_report.json
timeout -k 10 300 python -u tools/bounds_sweeps.py > $O/sweeps.log 2>&1 || true
tail -4 $O/sweeps.log
