"""Prompt pass of Llama-3-8B shapes at 4..768 rows, fp16 weight images resident (a 1024-row prompt
goes first): the form the engine picks (the few-rows kernel on the int8 weights up to 192 rows, the
fp16 GEMMs above), and beside it the tiled kernels with the few-rows kernel switched off
(KX_PREFILL_NO_ROWS_KERNEL=1: int8 single-pass below 32 rows, fp16 GEMMs from 32 rows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-inference-engine_amd"))
import kx
from kx import host
shape = host.MODEL_SHAPES["llama-3-8b"]
model = host.Model(seq_len=4608, group_size=64, prefill_weight_cache=1, **shape)
model.init_synthetic(1234)
rng = np.random.default_rng(1)
model.prefill(rng.integers(0, shape["vocab_size"], 1024).astype(np.int32))  # makes the fp16 images
model.sync()
for T in (4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256, 384, 512, 768):
    prompt = rng.integers(0, shape["vocab_size"], T).astype(np.int32)
    line = f"T={T:4d}"
    for tag, off in (("engine", None), ("tiled kernels", "1")):
        if T > 192 and off:
            continue
        if off:
            os.environ["KX_PREFILL_NO_ROWS_KERNEL"] = off
        else:
            os.environ.pop("KX_PREFILL_NO_ROWS_KERNEL", None)
        model.prefill(prompt); model.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); model.prefill(prompt); model.sync(); best = min(best, time.perf_counter() - t0)
        line += f"  {tag}: {best*1e3:7.2f} ms {T/best:8.0f} tok/s"
    print(line, flush=True)
os.environ.pop("KX_PREFILL_NO_ROWS_KERNEL", None)
# the same lengths on the int8 weights alone (policy 0: no fp16 image kept): 193..304 rows run as
# pieces of at most 192 rows through the few-rows kernel, longer ones convert the weights per pass
# and run the fp16 GEMMs; KX_PREFILL_NO_PIECES=1 = the int8 single-pass kernel in place of the pieces
model.set_prefill_weight_cache(0)
print("no fp16 weight images (prefill_weight_cache = 0):", flush=True)
for T in (192, 224, 256, 304, 320, 384, 400, 448, 512, 768):
    prompt = rng.integers(0, shape["vocab_size"], T).astype(np.int32)
    line = f"T={T:4d}"
    for tag, off in (("engine", None), ("whole prompt in one pass", "1")):
        if off:
            os.environ["KX_PREFILL_NO_PIECES"] = off
        else:
            os.environ.pop("KX_PREFILL_NO_PIECES", None)
        model.prefill(prompt); model.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); model.prefill(prompt); model.sync(); best = min(best, time.perf_counter() - t0)
        line += f"  {tag}: {best*1e3:7.2f} ms {T/best:8.0f} tok/s"
    print(line, flush=True)
os.environ.pop("KX_PREFILL_NO_PIECES", None)
