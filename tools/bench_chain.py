"""Wo -> W1|W3 -> W2 -> next Wq|Wk|Wv as four launches vs one chained launch (kx_gemv_w8_chain_f32),
on `layers` distinct sets of Llama-shaped weights (cold: a pass exceeds the 256 MiB Infinity Cache),
graph-captured, microseconds per layer."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GS = 64
SHAPES = {"8b": (4096, 14336, 32, 8, 128), "1b": (2048, 8192, 32, 8, 64),
          "70b": (8192, 28672, 64, 8, 128), "qwen7b": (3584, 18944, 28, 4, 128)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "8b"
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dim, hid, H, KVH, hs = SHAPES[name]
    kvd, S = KVH * hs, 512
    L = kx.lib()
    st = kx.Stream()

    def w8(K, M, seed):
        w = kx.Buf(K * M)
        s = kx.Buf(K * M // GS * 4)
        kx.check(L.kx_fill_hash_i8(w.ptr, K, M, M, 0, 0, seed, None))
        kx.check(L.kx_fill_hash_f32(s.ptr, K, M // GS, M // GS, 0, 0, seed + 1, 1e-4, 3e-4, None))
        return w, s
    rng = np.random.default_rng(0)
    x = kx.Buf.from_numpy(rng.standard_normal(dim).astype(np.float32))
    att = kx.Buf.from_numpy(rng.standard_normal(dim).astype(np.float32))
    h = kx.Buf.zeros(hid)
    q = kx.Buf.zeros(dim)
    kc, vc = kx.Buf.zeros((S, kvd)), kx.Buf.zeros((S, kvd))
    sin, cos = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    kx.check(L.kx_sin_cos_cache_f32(hs, S, 5e5, sin.ptr, cos.ptr, None))
    pos = kx.Buf.from_numpy(np.array([100], np.int32))
    nw = kx.Buf.from_numpy(np.ones(dim, np.float32))
    sb = L.kx_gemv_chain_state_bytes()
    state = kx.Buf.zeros(sb + 256 * 16 * 8, np.uint8)
    per_layer = []
    nbytes = 0
    for l in range(layers):
        wo, w13, w2, wqkv = w8(dim, dim, 10 * l), w8(2 * hid, dim, 10 * l + 2), w8(dim, hid, 10 * l + 4), \
            w8(dim + 2 * kvd, dim, 10 * l + 6)
        D = kx.GemvDesc
        descs = (D * 4)(
            D(mode=0, group_size=GS, M=dim, K=dim, x=att.ptr, w=wo[0].ptr, scales=wo[1].ptr, y0=x.ptr,
              residual=x.ptr),
            D(mode=1, group_size=GS, M=dim, K=2 * hid, x=x.ptr, w=w13[0].ptr, scales=w13[1].ptr,
              norm_w=nw.ptr, norm_eps=1e-5, y0=h.ptr),
            D(mode=0, group_size=GS, M=hid, K=dim, x=h.ptr, w=w2[0].ptr, scales=w2[1].ptr, y0=x.ptr,
              residual=x.ptr),
            D(mode=2, group_size=GS, M=dim, K=dim + 2 * kvd, x=x.ptr, w=wqkv[0].ptr, scales=wqkv[1].ptr,
              norm_w=nw.ptr, norm_eps=1e-5, y0=q.ptr, y1=kc.ptr, y2=vc.ptr, k0=dim, k1=dim + kvd,
              sin_cache=sin.ptr, cos_cache=cos.ptr, pos_dev=pos.ptr, rope_variant=1, head_size=hs,
              kv_pos_stride=kvd))
        per_layer.append((descs, (wo, w13, w2, wqkv)))
    for K, M in ((dim, dim), (2 * hid, dim), (dim, hid), (dim + 2 * kvd, dim)):
        nbytes += K * M + 4 * K * M // GS + 4 * (K + M)
    kx.sync()

    def separate():
        for descs, _ in per_layer:
            for i in range(4):
                kx.check(L.kx_gemv_w8_fused_f32(C.byref(descs[i]), st.ptr))

    def chained(n):
        def rec():
            for descs, _ in per_layer:
                kx.check(L.kx_gemv_w8_chain_f32(descs, n, state.ptr, state.nbytes, st.ptr))
                for i in range(n, 4):
                    kx.check(L.kx_gemv_w8_fused_f32(C.byref(descs[i]), st.ptr))
        return rec
    for label, fn in (("4 launches", separate), ("chain of 4", chained(4)), ("chain of 3 + 1", chained(3)),
                      ("chain of 2 + 2", chained(2)), ("4 launches", separate)):
        us = kx.time_graph(st, fn, replays=10) / layers
        print(f"{name} {label:16s} {us:8.2f} us/layer  {nbytes / us / 1e3:8.1f} GB/s "
              f"({nbytes / us / 8e6 * 100:.1f} % of 8 TB/s)", flush=True)
    err = state.numpy(np.uint32, (state.nbytes // 4,))[sb // 4 - 64]
    print("barrier error word:", err)
    if os.environ.get("KX_CHAIN_TRACE"):
        # timeline of the LAST chained launch, per workgroup: stamps 3i (weights of stage i requested),
        # 3i+1 (barrier passed), 3i+2 (x staged), 12 (done); 100 MHz wall clock -> microseconds
        kx.check(L.kx_gemv_w8_chain_f32(per_layer[0][0], 4, state.ptr, state.nbytes, st.ptr))
        st.sync()
        t = state.numpy(np.uint8, (state.nbytes,))[sb:].view(np.uint64).reshape(256, 16).astype(np.float64)
        t = (t - t[:, 0].min()) / 100.0
        names = ["s0 req", "s0 bar", "s0 x", "s1 req", "s1 bar", "s1 x", "s2 req", "s2 bar", "s2 x",
                 "s3 req", "s3 bar", "s3 x", "end"]
        for i, nme in enumerate(names):
            print(f"{nme:8s} min {t[:, i].min():7.2f} median {np.median(t[:, i]):7.2f} max {t[:, i].max():7.2f}")


if __name__ == "__main__":
    main()
