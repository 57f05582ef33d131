"""Does a side stream that reads the NEXT launches' weights into the die-level cache, while the
latency-bound attention launch runs, shorten the decode layer?  Llama-3-8B layer as its five
launches (QKV GEMV, attention at `pos`, Wo, W1|W3, W2; plain int8 GEMVs), layer sets rotated so
every launch reads cold bytes, HIP-graph replay.  Variants:
  none     the five launches on one stream
  attn:N   beside the attention launch the side stream touches Wo and the first N MB of W1|W3
  ahead    beside every launch the side stream touches the whole matrix of the launch after it
Usage: probe_prefetch.py [pos] [workgroups of the touch kernel]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 4367
twg = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim, hid, H, KVH, hs = 4096, 14336, 32, 8, 128
ql, kl = H * hs, KVH * hs
L = kx.lib()
st, side = kx.Stream(), kx.Stream()
GS = 64
seq = pos + 1


def mat(rows, cols, seed):
    w, s = kx.Buf(rows * cols), kx.Buf(rows * cols // GS * 4)
    kx.check(L.kx_fill_hash_i8(w.ptr, rows, cols, cols, 0, 0, seed, st.ptr), "fill")
    kx.check(L.kx_fill_hash_f32(s.ptr, rows, cols // GS, cols // GS, 0, 0, seed + 1, 1e-4, 3e-4, st.ptr), "fill")
    return w, s, rows, cols


def vec(n, seed):
    v = kx.Buf(n * 4)
    kx.check(L.kx_fill_hash_f32(v.ptr, 1, n, n, 0, 0, seed, -1.0, 1.0, st.ptr), "fill")
    return v


nl = 4
layers = [dict(qkv=mat(ql + 2 * kl, dim, 10 * l + 1), wo=mat(dim, ql, 10 * l + 3), w13=mat(2 * hid, dim, 10 * l + 5),
               w2=mat(dim, hid, 10 * l + 7)) for l in range(nl)]
kc, vc = kx.Buf(nl * seq * kl * 4), kx.Buf(nl * seq * kl * 4)
kx.check(L.kx_fill_hash_f32(kc.ptr, nl * seq, kl, kl, 0, 0, 1, -1.0, 1.0, st.ptr), "fill")
kx.check(L.kx_fill_hash_f32(vc.ptr, nl * seq, kl, kl, 0, 0, 2, -1.0, 1.0, st.ptr), "fill")
x, qkv, att, h2 = vec(dim, 1), vec(ql + 2 * kl, 2), vec(ql, 3), vec(2 * hid, 4)
ws = kx.Buf.zeros(L.kx_attn_workspace_bytes(H, hs), np.uint8)
sink = kx.Buf.zeros(4, np.int32)
st.sync()


def gemv(m, xin, yout):
    w, s, K, M = m
    kx.check(L.kx_gemv_w8_f32(xin.ptr, w.ptr, s.ptr, GS, yout.ptr, M, K, st.ptr), "gemv")


def touch(m, mb=None):
    w = m[0]
    n = w.nbytes if mb is None else min(w.nbytes, int(mb * 1e6))
    kx.check(L.kx_touch_bytes(w.ptr, n, twg, sink.ptr, side.ptr), "touch")


events = [kx.Event() for _ in range(64)]


def record(variant):
    ev = iter(events)

    def fork():   # the side stream continues from this point of the main stream
        e = next(ev)
        e.record(st.ptr)
        kx.check(L.kx_stream_wait_event(side.ptr, e.ptr), "wait")

    for l in range(nl):
        lw = layers[l]
        nxt = layers[(l + 1) % nl]
        if variant == "ahead":
            fork(); touch(lw["wo"])
        gemv(lw["qkv"], x, qkv)
        if variant.startswith("attn"):
            fork(); touch(lw["wo"]); touch(lw["w13"], float(variant.split(":")[1]))
        if variant == "ahead":
            fork(); touch(lw["w13"], 60)
        kx.check(L.kx_mha_decode_f32(qkv.ptr, kc.ptr, vc.ptr, att.ptr, pos, H, H // KVH, hs, kl, seq, l, ws.ptr,
                                     ws.nbytes, st.ptr), "attn")
        gemv(lw["wo"], att, x)
        if variant == "ahead":
            fork(); touch(lw["w2"])
        gemv(lw["w13"], x, h2)
        if variant == "ahead":
            fork(); touch(nxt["qkv"])
        gemv(lw["w2"], h2, x)
    if variant != "none":   # the side stream joins before the capture ends
        e = next(ev)
        e.record(side.ptr)
        kx.check(L.kx_stream_wait_event(st.ptr, e.ptr), "join")


print(f"pos {pos}, touch kernel {twg} workgroups, {kx.device_name()}")
for variant in ("none", "attn:0", "attn:30", "attn:60", "ahead", "none"):
    us = kx.time_graph(st, lambda: record(variant), 20) / nl
    print(f"{variant:8s} {us:8.2f} us per layer", flush=True)
