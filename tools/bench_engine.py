"""Times the chained-GEMV launch (kx_gemv_w8_chain_f32) against the single launches it replaces,
on one model's layer shapes, cold weights (buffers rotated over > 2 x the last-level cache), graph
replay.  Usage: bench_engine.py [8b|1b|70b] [replays]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

SHAPES = {"8b": (4096, 14336, 32, 8, 128), "1b": (2048, 8192, 32, 8, 64), "70b": (8192, 28672, 64, 8, 128)}
name = sys.argv[1] if len(sys.argv) > 1 else "8b"
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dim, hid, H, KVH, hs = SHAPES[name]
ql, kl = H * hs, KVH * hs
L = kx.lib()
st = kx.Stream()
GS = 64


def mat(rows, cols, seed):
    w, s = kx.Buf(rows * cols), kx.Buf(rows * cols // GS * 4)
    kx.check(L.kx_fill_hash_i8(w.ptr, rows, cols, cols, 0, 0, seed, st.ptr), "fill")
    kx.check(L.kx_fill_hash_f32(s.ptr, rows, cols // GS, cols // GS, 0, 0, seed + 1, 1e-4, 3e-4, st.ptr), "fill")
    return w, s


def vec(n, seed, lo=-1.0, hi=1.0):
    v = kx.Buf(n * 4)
    kx.check(L.kx_fill_hash_f32(v.ptr, 1, n, n, 0, 0, seed, lo, hi, st.ptr), "fill")
    return v


per_layer = ((ql + 2 * kl) * dim + dim * ql + 2 * hid * dim + dim * hid) * 1.0625
nlayers = max(3, int(600e6 / per_layer) + 1)
layers = []
for l in range(nlayers):
    layers.append(dict(qkv=mat(ql + 2 * kl, dim, 10 * l + 1), wo=mat(dim, ql, 10 * l + 3), w13=mat(2 * hid, dim, 10 * l + 5),
                       w2=mat(dim, hid, 10 * l + 7)))
att, x, h, q = vec(ql, 1), vec(dim, 2), vec(hid, 3), vec(ql, 4)
norm = vec(dim, 5, 0.5, 1.5)
S = 64
kc, vc = kx.Buf.zeros(S * kl), kx.Buf.zeros(S * kl)
sin_t, cos_t = kx.Buf(S * hs * 4), kx.Buf(S * hs * 4)
kx.check(L.kx_sin_cos_cache_f32(hs, S, 500000.0, sin_t.ptr, cos_t.ptr, st.ptr), "sincos")
pos_dev = kx.Buf.from_numpy(np.array([5], np.int32))
ws = kx.Buf(L.kx_gemv_w8_chain_workspace_bytes(max(dim, hid)))
kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, st.ptr), "reset")
st.sync()


def descs(lw):
    wo = kx.make_gemv_desc(kx.GEMV_PLAIN, att.ptr, lw["wo"][0].ptr, lw["wo"][1].ptr, x.ptr, ql, dim, residual=x.ptr)
    w13 = kx.make_gemv_desc(kx.GEMV_GATE_UP, x.ptr, lw["w13"][0].ptr, lw["w13"][1].ptr, h.ptr, dim, 2 * hid,
                            norm_w=norm.ptr)
    w2 = kx.make_gemv_desc(kx.GEMV_PLAIN, h.ptr, lw["w2"][0].ptr, lw["w2"][1].ptr, x.ptr, hid, dim, residual=x.ptr)
    qkv = kx.make_gemv_desc(kx.GEMV_QKV_ROPE, x.ptr, lw["qkv"][0].ptr, lw["qkv"][1].ptr, q.ptr, dim, ql + 2 * kl,
                            norm_w=norm.ptr, y1=kc.ptr, y2=vc.ptr, k0=ql, k1=ql + kl, sin=sin_t.ptr, cos=cos_t.ptr,
                            pos_dev=pos_dev.ptr, rope_variant=1, head_size=hs, kv_pos_stride=kl)
    return wo, w13, w2, qkv


def single(d):
    kx.check(L.kx_gemv_w8_fused_f32(C.byref(d), st.ptr), "fused")


def run_single(which):
    for lw in layers:
        wo, w13, w2, qkv = descs(lw)
        for d in {"A": (wo, w13), "B": (w2, qkv), "wo": (wo,), "w13": (w13,), "w2": (w2,), "qkv": (qkv,)}[which]:
            single(d)


def run_chain(which):
    for i, lw in enumerate(layers):
        wo, w13, w2, qkv = descs(lw)
        ops = {"A": (wo, w13), "B": (w2, qkv), "wo": (wo,), "w13": (w13,), "w2": (w2,), "qkv": (qkv,)}[which]
        kx.gemv_chain(list(ops), ws, chain_id=i % 128, pos_dev=pos_dev.ptr, stream=st.ptr)


def bytes_of(which):
    b = {"wo": dim * ql, "w13": 2 * hid * dim, "w2": dim * hid, "qkv": (ql + 2 * kl) * dim}
    b["A"], b["B"] = b["wo"] + b["w13"], b["w2"] + b["qkv"]
    return b[which] * 1.0625


print(f"{name}: {nlayers} layer sets rotated ({nlayers * per_layer / 1e6:.0f} MB), {kx.device_name()}")
for which in ("wo", "w13", "w2", "qkv", "A", "B"):
    t_s = kx.time_graph(st, lambda: run_single(which), replays) / nlayers
    t_c = kx.time_graph(st, lambda: run_chain(which), replays) / nlayers
    gave = C.c_int32(0)
    kx.check(L.kx_gemv_w8_chain_status(ws.ptr, C.byref(gave)), "status")
    if os.environ.get("KX_CHAIN_TRACE"):
        kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, st.ptr), "reset")
        run_chain(which)
        st.sync()
        w = ws.numpy(np.int32, (16,))
        ncu, n = 256, nlayers
        print(f"     per CU per launch: loader space polls {w[1] / ncu / n:.0f}, steps blocked on space {w[2] / ncu / n:.1f}, "
              f"loader time {w[3] / ncu / n / 100:.2f} us | consumer 0 fill polls {w[4] / ncu / n:.0f}, steps starved "
              f"{w[5] / ncu / n:.1f}, staging {w[6] / ncu / n / 100:.2f} us, run {w[7] / ncu / n / 100:.2f} us, "
              f"steps {w[8] / ncu / n:.1f}")
    print(f"{which:4s} single launches {t_s:7.2f} us ({bytes_of(which) / t_s / 1e6:6.2f} TB/s)   "
          f"chain launch {t_c:7.2f} us ({bytes_of(which) / t_c / 1e6:6.2f} TB/s)   ratio {t_c / t_s:5.3f}"
          f"{'   A WAIT GAVE UP' if gave.value else ''}", flush=True)
