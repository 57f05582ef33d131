"""Times the four fp16-image GEMMs of one layer (W1|W3 with SwiGLU, W2, Wq|Wk|Wv, Wo; weights kept,
residual on the plain ones) at several prompt lengths, under each tile kernel and each way of cutting
the reductions of a partial last round of tiles (KX_GEMM_F16_KERNEL, KX_GEMM_SPLIT), and as the
library picks by itself ("auto").  MODEL = 8b | 1b | q7 | 70b-tp8, TS = comma-separated row counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-inference-engine_amd"))
import kx
L = kx.lib()
GS = 64
rng = np.random.default_rng(0)
stream = kx.Stream()
sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
dims = {"70b-tp8": (8192, 3584, 128), "8b": (4096, 14336, 1024), "1b": (2048, 8192, 512), "q7": (3584, 18944, 512)}
model = os.environ.get("MODEL", "8b")
dim, hid, kvd = dims[model]
for T in [int(t) for t in os.environ.get("TS", "512,1024,2048,4096").split(",")]:
    shapes = [(T, 2 * hid, dim, 1), (T, dim, hid, 0), (T, dim + 2 * kvd, dim, 0), (T, dim, dim, 0)]
    if model == "70b-tp8":
        shapes = [(T, 7168, 8192, 1), (T, 8192, 3584, 0), (T, 1280, 8192, 0), (T, 8192, 1024, 0)]
    tot = {}
    for (T, K, M, swiglu) in shapes:
        x = kx.Buf.from_numpy(rng.standard_normal((T, M)).astype(np.float16))
        w = kx.Buf.from_numpy(rng.integers(-127, 128, (K, M)).astype(np.int8))
        s = kx.Buf.from_numpy((rng.random(K * M // GS).astype(np.float32) + 0.5) * 1e-3)
        y = kx.Buf.zeros((T, K)); r = kx.Buf.zeros((T, K))
        h = kx.Buf.zeros((T, K // 2), np.float16)
        wsb = L.kx_gemm_w8_f16act_workspace_bytes(M, K)
        ws = kx.Buf(wsb)
        def rec(flag, st=stream.ptr):
            if swiglu:
                kx.check(L.kx_gemm_w8_f16act_split_ws(x.ptr, w.ptr, s.ptr, GS, 1 | flag, h.ptr, None, 0, None, 0, None, 0, 0, 0, None, 0, T, M, K, ws.ptr, wsb, sp.ptr, sp.nbytes, st))
            else:
                kx.check(L.kx_gemm_w8_f16act_split_ws(x.ptr, w.ptr, s.ptr, GS, 0 | flag, None, y.ptr, K, None, 0, None, 0, K, K, r.ptr, K, T, M, K, ws.ptr, wsb, sp.ptr, sp.nbytes, st))
        rec(0); stream.sync()
        line = f"T={T:5d} K={K:6d} M={M:6d} {'swiglu' if swiglu else 'plain '}"
        for name, env in (("k1", {"KX_GEMM_F16_KERNEL": "1"}), ("k3s1", {"KX_GEMM_F16_KERNEL": "3", "KX_GEMM_SPLIT": "1"}),
                          ("k3s2", {"KX_GEMM_F16_KERNEL": "3", "KX_GEMM_SPLIT": "2"}), ("k3s4", {"KX_GEMM_F16_KERNEL": "3", "KX_GEMM_SPLIT": "4"}),
                          ("k3s8", {"KX_GEMM_F16_KERNEL": "3", "KX_GEMM_SPLIT": "8"}), ("k3s16", {"KX_GEMM_F16_KERNEL": "3", "KX_GEMM_SPLIT": "16"}),
                          ("k3auto", {"KX_GEMM_F16_KERNEL": "3"}), ("auto", {})):
            for k in ("KX_GEMM_F16_KERNEL", "KX_GEMM_SPLIT"):
                os.environ.pop(k, None)
            os.environ.update(env)
            us = kx.time_graph(stream, lambda: rec(0x100), 6)
            tot[name] = tot.get(name, 0) + us
            line += f" {name} {us:7.1f}"
        print(line, flush=True)
    print(f"T={T:5d} layer total " + " ".join(f"{k} {v:7.1f}" for k, v in tot.items()), flush=True)
