"""What would two half-width chains on two streams buy?  The token loop's GEMVs split as tensor
parallelism splits them (q|k|v, W1|W3 by rows; Wo, W2 by columns), both halves on ONE GPU, one per
stream, joined where a layer needs the whole vector (behind Wo and behind W2): launches of the two
chains fill each other's heads and tails.  Timing only (the halves' partial sums are not added),
Llama-3-8B shapes, cold weights (12 layers in rotation), graph replay:
  a) the four full GEMVs of a layer on one stream (what the token loop does)
  b) eight half GEMVs, two streams, joins behind Wo and W2 (two per layer)
  c) as b) without the joins (upper bound)"""
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
    full = {"qkv": (6144, 4096), "wo": (4096, 4096), "w13": (28672, 4096), "w2": (4096, 14336)}
    half = {"qkv": (3072, 4096), "wo": (4096, 2048), "w13": (14336, 4096), "w2": (4096, 7168)}

    def make(shapes, copies):
        host = {n: (rng.integers(-127, 128, size=s, dtype=np.int8),
                    rng.uniform(1e-4, 3e-4, size=s[0] * s[1] // GS).astype(np.float32)) for n, s in shapes.items()}
        return [{n: (kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)) for n, (w, s) in host.items()} for _ in range(copies)]

    dev_full = make(full, LAYERS)
    dev_half = make(half, 2 * LAYERS)
    x = kx.Buf.from_numpy(rng.standard_normal(14336).astype(np.float32) * 0.01)
    outs = [[kx.Buf.zeros(6144), kx.Buf.zeros(4096), kx.Buf.zeros(14336), kx.Buf.zeros(4096)] for _ in range(2)]
    evs = [kx.Event() for _ in range(4)]

    def gemv(d, name, out, s, M, K):
        kx.check(L.kx_gemv_w8_f32(x.ptr, d[name][0].ptr, d[name][1].ptr, GS, out.ptr, M, K, s.ptr))

    def swiglu(d, out, s, M, H):
        kx.check(L.kx_gemv_w8_swiglu_f32(x.ptr, d["w13"][0].ptr, d["w13"][1].ptr, GS, out.ptr, M, H, s.ptr))

    def one():
        o = outs[0]
        for _ in range(2):
            for d in dev_full:
                gemv(d, "qkv", o[0], st, 4096, 6144)
                gemv(d, "wo", o[1], st, 4096, 4096)
                swiglu(d, o[2], st, 4096, 14336)
                gemv(d, "w2", o[3], st, 14336, 4096)

    def join():
        evs[0].record(st.ptr)
        evs[1].record(st2.ptr)
        kx.check(L.kx_stream_wait_event(st.ptr, evs[1].ptr))
        kx.check(L.kx_stream_wait_event(st2.ptr, evs[0].ptr))

    def dual(joins):
        evs[2].record(st.ptr)
        kx.check(L.kx_stream_wait_event(st2.ptr, evs[2].ptr))
        for _ in range(2):
            for i in range(LAYERS):
                for c, s in ((0, st), (1, st2)):
                    d, o = dev_half[2 * i + c], outs[c]
                    gemv(d, "qkv", o[0], s, 4096, 3072)
                    gemv(d, "wo", o[1], s, 2048, 4096)
                if joins:
                    join()
                for c, s in ((0, st), (1, st2)):
                    d, o = dev_half[2 * i + c], outs[c]
                    swiglu(d, o[2], s, 4096, 7168)
                    gemv(d, "w2", o[3], s, 7168, 4096)
                if joins:
                    join()
        evs[3].record(st2.ptr)
        kx.check(L.kx_stream_wait_event(st.ptr, evs[3].ptr))

    per = sum(s[0] * s[1] * (1 + 4 / GS) for s in full.values())
    for name, rec in (("a) one stream, full GEMVs", one), ("b) two chains, two joins per layer", lambda: dual(True)),
                      ("c) two chains, no joins", lambda: dual(False)), ("a) one stream, full GEMVs", one),
                      ("b) two chains, two joins per layer", lambda: dual(True))):
        us = kx.time_graph(st, rec) / (2 * LAYERS)
        print(f"{name:38s} {us:7.2f} us per layer  {per / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
