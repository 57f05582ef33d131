"""Times the two MFMA prefill kernels standalone (the same measurement bench.py reports under
prefill_kernels): causal flash attention over T rows and the two-pass W8A16 GEMM."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402
from kx import host  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    print(json.dumps(bench.prefill_kernels(kx, host.MODEL_SHAPES[model], 1, T), indent=1))
