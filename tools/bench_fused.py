"""Times the fused W8 GEMV modes on the Llama-3-8B layer shapes (rotating weight buffers)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GS = 64


Desc = kx.GemvDesc


def main():
    L = kx.lib()
    st = kx.Stream()
    rng = np.random.default_rng(0)
    dim, kvd, hid, hs, S = 4096, 1024, 14336, 128, 4608
    sin = kx.Buf.zeros((S, hs)); cos = kx.Buf.zeros((S, hs))
    kc = kx.Buf.zeros((S, kvd)); vc = kx.Buf.zeros((S, kvd))
    pos = kx.Buf.from_numpy(np.array([4000], np.int32))
    cases = [("plain", 0, dim, dim, False, False), ("plain+res", 0, dim, dim, False, True),
             ("plain+norm", 0, dim, dim, True, False),
             ("qkv_rope+norm", 2, dim + 2 * kvd, dim, True, False),
             ("gate_up+norm", 1, 2 * hid, dim, True, False),
             ("w2 plain+res", 0, dim, hid, False, True),
             ("w13 as plain", 0, 2 * hid, dim, False, False)]
    for name, mode, K, M, norm, res in cases:
        per = K * M * (1 + 4 / GS)
        nbuf = max(2, int(600e6 // per) + 1)
        w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
        s = rng.uniform(1e-4, 3e-4, size=K * M // GS).astype(np.float32)
        ws = [kx.Buf.from_numpy(w) for _ in range(nbuf)]
        ss = [kx.Buf.from_numpy(s) for _ in range(nbuf)]
        x = kx.Buf.from_numpy(rng.standard_normal(M).astype(np.float32))
        nw = kx.Buf.from_numpy(np.ones(M, np.float32))
        y = kx.Buf.zeros(K)
        descs = []
        for j in range(nbuf):
            d = Desc(mode=mode, group_size=GS, M=M, K=K, x=x.ptr, w=ws[j].ptr, scales=ss[j].ptr,
                     norm_w=nw.ptr if norm else None, norm_eps=1e-5, y0=y.ptr,
                     residual=y.ptr if res else None)
            if mode == 2:
                d.y1, d.y2, d.k0, d.k1 = kc.ptr, vc.ptr, dim, dim + kvd
                d.sin_cache, d.cos_cache, d.pos_dev = sin.ptr, cos.ptr, pos.ptr
                d.rope_variant, d.head_size, d.kv_pos_stride = 1, hs, kvd
            descs.append(d)
        reps = max(1, 48 // nbuf)

        def record():
            for _ in range(reps):
                for d in descs:
                    kx.check(L.kx_gemv_w8_fused_f32(C.byref(d), st.ptr), name)
        try:
            us = kx.time_graph(st, record) / (reps * nbuf)
        except kx.KxError as e:
            print(f"{name:16s} skipped ({e})", flush=True)
            continue
        b = K * M + 4 * K * M // GS + 4 * (M + K)
        print(f"{name:16s} K={K:6d} M={M:6d} {us:8.2f} us {b / us / 1e3:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
