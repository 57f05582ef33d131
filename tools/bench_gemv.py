"""Micro-benchmark of the W8 GEMV entry on the linear shapes of the model configs.
Rotates over enough distinct weight buffers to defeat the 256 MiB Infinity Cache."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GS = 64
SHAPES = {
    "8b": [(4096, 4096), (1024, 4096), (6144, 4096), (14336, 4096), (28672, 4096), (4096, 14336),
           (128256, 4096)],
    "1b": [(2048, 2048), (512, 2048), (3072, 2048), (8192, 2048), (16384, 2048), (2048, 8192),
           (128256, 2048)],
    "q7": [(3584, 3584), (512, 3584), (18944, 3584), (3584, 18944), (152064, 3584)],
    "70b": [(8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672)],
}


def algo_bytes(K, M):
    return K * M + 4 * K * M // GS + 4 * (M + K)


def main():
    which = sys.argv[1:] or ["8b", "1b"]
    # bytes of distinct weights cycled through: > 256 MiB keeps every read cold (HBM); ~120e6 leaves
    # the weights resident in the Infinity Cache but not in the 8 x 4 MiB L2s
    footprint = float(os.environ.get("KX_BENCH_FOOTPRINT", "600e6"))
    L = kx.lib()
    st = kx.Stream()
    rng = np.random.default_rng(0)
    print(kx.device_name())
    for name in which:
        for K, M in SHAPES[name]:
            per = K * M + 4 * K * M // GS
            nbuf = max(1, min(64, int(footprint // per) + 1))
            w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
            s = rng.uniform(1e-4, 3e-4, size=K * M // GS).astype(np.float32)
            ws = [kx.Buf.from_numpy(w) for _ in range(nbuf)]
            ss = [kx.Buf.from_numpy(s) for _ in range(nbuf)]
            x = kx.Buf.from_numpy(rng.standard_normal(M).astype(np.float32))
            y = kx.Buf.zeros(K)
            reps = max(1, 64 // nbuf)

            def record():
                for _ in range(reps):
                    for j in range(nbuf):
                        kx.check(L.kx_gemv_w8_f32(x.ptr, ws[j].ptr, ss[j].ptr, GS, y.ptr, M, K, st.ptr))
            us = kx.time_graph(st, record) / (reps * nbuf)
            gbs = algo_bytes(K, M) / us / 1e3
            print(f"{name} K={K:6d} M={M:6d} nbuf={nbuf:3d} {us:9.2f} us  {gbs:8.1f} GB/s  "
                  f"{gbs / 8000 * 100:5.1f}% of 8TB/s", flush=True)
            del ws, ss


if __name__ == "__main__":
    main()
