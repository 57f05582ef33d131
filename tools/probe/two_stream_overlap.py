"""Upper bound for overlapping a GEMV launch's head with its predecessor's tail: the four W8 GEMVs
of a Llama-3-8B layer (q|k|v as a plain 6144 x 4096 product), cold weights (12 layers in rotation),
replayed from a graph (a) on one stream and (b) alternating between two streams, where launch k+1
may start while launch k still runs (its input would not be ready: timing only)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GS = 64
LAYERS = 12


def main():
    L = kx.lib()
    st, st2 = kx.Stream(), kx.Stream()
    rng = np.random.default_rng(0)
    print(kx.device_name())
    shapes = {"qkv": (6144, 4096), "wo": (4096, 4096), "w13": (28672, 4096), "w2": (4096, 14336)}
    host = {n: (rng.integers(-127, 128, size=s, dtype=np.int8),
                rng.uniform(1e-4, 3e-4, size=s[0] * s[1] // GS).astype(np.float32)) for n, s in shapes.items()}
    dev = [{n: (kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)) for n, (w, s) in host.items()} for _ in range(LAYERS)]
    x = kx.Buf.from_numpy(rng.standard_normal(14336).astype(np.float32) * 0.01)
    q, o, h, y = kx.Buf.zeros(6144), kx.Buf.zeros(4096), kx.Buf.zeros(14336), kx.Buf.zeros(4096)
    ev0, ev1 = kx.Event(), kx.Event()

    def layer(d, sa, sb):
        kx.check(L.kx_gemv_w8_f32(x.ptr, d["qkv"][0].ptr, d["qkv"][1].ptr, GS, q.ptr, 4096, 6144, sa.ptr))
        kx.check(L.kx_gemv_w8_f32(x.ptr, d["wo"][0].ptr, d["wo"][1].ptr, GS, o.ptr, 4096, 4096, sb.ptr))
        kx.check(L.kx_gemv_w8_swiglu_f32(x.ptr, d["w13"][0].ptr, d["w13"][1].ptr, GS, h.ptr, 4096, 14336, sa.ptr))
        kx.check(L.kx_gemv_w8_f32(x.ptr, d["w2"][0].ptr, d["w2"][1].ptr, GS, y.ptr, 14336, 4096, sb.ptr))

    def one():
        for _ in range(2):
            for d in dev:
                layer(d, st, st)

    def two():
        ev0.record(st.ptr)
        kx.check(L.kx_stream_wait_event(st2.ptr, ev0.ptr))
        for _ in range(2):
            for d in dev:
                layer(d, st, st2)
        ev1.record(st2.ptr)
        kx.check(L.kx_stream_wait_event(st.ptr, ev1.ptr))

    per = sum(s[0] * s[1] * (1 + 4 / GS) for s in shapes.values())
    for name, rec in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
        us = kx.time_graph(st, rec) / (2 * LAYERS)
        print(f"{name:12s} {us:7.2f} us per layer's four GEMVs  {per / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
