"""Gate/up GEMV (kx_gemv_w8_swiglu_f32) at hidden sizes that give every wave the same number of
gate/up pairs and at Llama-3-8B's 14336 (1024 workgroups of 14 pairs: waves take 4, 4, 3, 3):
does the uneven last quarter cost bandwidth?  Cold weights (buffers in rotation), graph replay."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GS = 64


def main():
    L = kx.lib()
    st = kx.Stream()
    rng = np.random.default_rng(0)
    M = 4096
    print(kx.device_name())
    for H in (12288, 14336, 16384, 13312, 15360):
        K = 2 * H
        per = K * M + 4 * K * M // GS
        nbuf = int(700e6 // per) + 1
        w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
        s = rng.uniform(1e-4, 3e-4, size=K * M // GS).astype(np.float32)
        ws = [kx.Buf.from_numpy(w) for _ in range(nbuf)]
        ss = [kx.Buf.from_numpy(s) for _ in range(nbuf)]
        x = kx.Buf.from_numpy(rng.standard_normal(M).astype(np.float32))
        y = kx.Buf.zeros(H)

        def record():
            for _ in range(4):
                for j in range(nbuf):
                    kx.check(L.kx_gemv_w8_swiglu_f32(x.ptr, ws[j].ptr, ss[j].ptr, GS, y.ptr, M, H, st.ptr))
        us = kx.time_graph(st, record) / (4 * nbuf)
        bytes_ = per + 4 * (M + H)
        print(f"H={H:6d} pairs/workgroup={H / 1024:5.2f}  {us:7.2f} us  {bytes_ / us / 1e3:7.1f} GB/s  "
              f"model 2.7 + bytes/6.9: {2.7 + bytes_ / 6.9e6:6.2f} us", flush=True)
        del ws, ss


if __name__ == "__main__":
    main()
