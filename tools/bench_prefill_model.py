"""Runs only the prompt pass of a model (random-init weights) a few times and prints tok/s; the
command to put behind rocprofv3 --kernel-trace --stats for a per-kernel view of the prompt pass.
usage: bench_prefill_model.py [model] [tokens] [passes] [warm_tokens]
warm_tokens > 0: a prompt of that many rows goes first (512+ rows leave the fp16 weight images behind)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402
from kx import host  # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    shape = host.MODEL_SHAPES[name]
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    model = host.Model(seq_len=(max(n, warm) + 127) // 64 * 64, group_size=64, **shape)
    model.init_synthetic(1234)
    prompt = np.random.default_rng(42).integers(0, shape["vocab_size"], n).astype(np.int32)
    if warm:
        model.prefill(np.random.default_rng(43).integers(0, shape["vocab_size"], warm).astype(np.int32))
    model.prefill(prompt)
    model.sync()
    for _ in range(passes):
        t0 = time.perf_counter()
        model.prefill(prompt)
        model.sync()
        dt = time.perf_counter() - t0
        print(f"{name} {n} tokens: {dt * 1e3:.2f} ms, {n / dt:.0f} tok/s", flush=True)
