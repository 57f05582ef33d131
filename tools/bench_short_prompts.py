"""Prompt pass of Llama-3-8B shapes at 32..768 rows with the fp16 weight images resident (a 1024-row
prompt goes first), once with the int8 single-pass kernel below 512 rows (KX_PREFILL_F16_MIN_ROWS=512)
and once with the fp16 path from 32 rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-inference-engine_amd"))
import kx
from kx import host
shape = host.MODEL_SHAPES["llama-3-8b"]
model = host.Model(seq_len=4608, group_size=64, **shape)
model.init_synthetic(1234)
rng = np.random.default_rng(1)
model.prefill(rng.integers(0, shape["vocab_size"], 1024).astype(np.int32))  # makes the fp16 images
model.sync()
for T in (32, 64, 128, 192, 256, 384, 512, 768):
    prompt = rng.integers(0, shape["vocab_size"], T).astype(np.int32)
    line = f"T={T:4d}"
    for thr in ("512", "32"):
        os.environ["KX_PREFILL_F16_MIN_ROWS"] = thr
        model.prefill(prompt); model.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); model.prefill(prompt); model.sync(); best = min(best, time.perf_counter() - t0)
        line += f"  min_rows {thr:>3s}: {best*1e3:7.2f} ms {T/best:8.0f} tok/s"
    print(line, flush=True)
