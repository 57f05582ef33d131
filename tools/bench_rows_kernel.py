"""The few-rows W8A16 product (kx_gemm_w8_rows_f16act) on the four layer matrices of a model, 16 / 32 / 64 /
128 rows: launches back to back on one stream over rotating copies of the weights (more bytes than the
256 MB cache holds), wall time per launch and the weight bytes per second it amounts to.
Usage: bench_rows_kernel.py [model] [launches] [rows,rows,...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-inference-engine_amd"))
import kx
from kx import host
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sh = host.MODEL_SHAPES[name]
dim, hid = sh["dim"], sh["hidden_dim"]
hs = dim // sh["head_num"]
kv = sh["kv_head_num"] * hs
shapes = {"wq|wk|wv": (dim + 2 * kv, dim), "wo": (dim, dim), "w1|w3": (2 * hid, dim), "w2": (dim, hid)}
GS = 64
lib = kx.lib()
sp = kx.Buf.zeros(lib.kx_gemm_split_workspace_bytes(), np.uint8)
rng = np.random.default_rng(1)
row_counts = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 32, 64, 128]
for T in row_counts:
    total = 0.0
    for label, (K, M) in shapes.items():
        ncopy = max(2, int(600e6 // (K * M)) + 1)
        ws = [kx.Buf.from_numpy(rng.integers(-127, 128, (K, M), dtype=np.int8)) for _ in range(ncopy)]
        ss = kx.Buf.from_numpy((rng.random(K * M // GS, dtype=np.float32) * 0.02).astype(np.float32))
        X = kx.Buf.from_numpy(rng.standard_normal((T, M)).astype(np.float16))
        Y = kx.Buf.zeros((T, K))
        def run(n):
            for i in range(n):
                rc = lib.kx_gemm_w8_rows_f16act(X.ptr, M, ws[i % ncopy].ptr, ss.ptr, GS, Y.ptr, K, None, 0, None, 0,
                                                K, K, None, 0, T, M, K, sp.ptr, sp.nbytes, None)
                assert rc == 0, rc
        run(5); kx.sync()
        t0 = time.perf_counter(); run(launches); kx.sync()
        dt = (time.perf_counter() - t0) / launches
        total += dt
        print(f"{name} T={T:3d} {label:9s} {K:6d} x {M:6d}: {dt * 1e6:7.1f} us  {K * M * 1.0625 / dt / 1e12:5.2f} TB/s", flush=True)
        del ws
    print(f"{name} T={T:3d} layer: {total * 1e6:7.1f} us", flush=True)
