import os, sys, time
import numpy as np
sys.path.insert(0, "llm-inference-engine_amd")
import kx
from kx import host
shape = host.MODEL_SHAPES["llama-3-8b"]
model = host.Model(seq_len=4608, group_size=64, prefill_weight_cache=0, **shape)
model.init_synthetic(1234)
rng = np.random.default_rng(1)
for T in (224, 256, 288, 320, 352, 384, 400):
    prompt = rng.integers(0, shape["vocab_size"], T).astype(np.int32)
    line = f"T={T:4d}"
    for tag, env in (("pieces", {}), ("convert + fp16 GEMMs", {"KX_PREFILL_NO_PIECES": "1", "KX_PREFILL_F16_MIN_ROWS": "193"})):
        for k in ("KX_PREFILL_NO_PIECES", "KX_PREFILL_F16_MIN_ROWS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        model.prefill(prompt); model.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); model.prefill(prompt); model.sync(); best = min(best, time.perf_counter() - t0)
        line += f"  {tag}: {best*1e3:7.2f} ms"
    print(line, flush=True)
