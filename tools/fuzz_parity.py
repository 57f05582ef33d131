"""Randomised parity sweep of the hot-path kernels against the CPU oracle (a tool, not a test: run it
on the GPU box after a kernel change; the fixed cases of tests/ are the gate).  Shapes, positions and
value scales are drawn from a seeded generator; every mismatch is printed with its case.
Usage: fuzz_parity.py [cases per kernel] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402
from oracle import kxo  # noqa: E402

def run(N=40, seed=1):
    """N cases per kernel from the generator seeded with `seed`; returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    GS = 64
    bad = 0


    def w8(K, M, scale_lo=1e-4, scale_hi=2e-2):
        w = rng.integers(-127, 128, (K, M), dtype=np.int8)
        s = rng.uniform(scale_lo, scale_hi, K * M // GS).astype(np.float32)
        return w, s


    def report(name, case, err, tol):
        nonlocal bad
        if not np.all(err <= tol):
            bad += 1
            print(f"MISMATCH {name} {case}: worst {float((err - tol).max()):.3e} over the bound", flush=True)


    # ---- decode GEMV: any K, M a multiple of 64 ----
    for _ in range(N):
        K = int(rng.integers(1, 700))
        M = int(rng.integers(1, 40)) * 64
        w, s = w8(K, M)
        x = (rng.standard_normal(M) * 10.0 ** rng.uniform(-2, 2)).astype(np.float32)
        got = kx.gemv_w8(x, w, s, GS)
        want = kxo.gemv_w8(x, w, s, GS)
        absum = np.abs(x.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w)).T
        report("gemv_w8", (K, M), np.abs(got - want), 1e-5 * absum + 1e-30)

    # ---- few-rows product: T <= 192, M a multiple of 256 ----
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    for _ in range(N):
        T = int(rng.integers(2, 193))
        K = int(rng.integers(1, 900))
        M = int(rng.integers(1, 9)) * 256
        w, s = w8(K, M)
        X = (rng.standard_normal((T, M)) * 10.0 ** rng.uniform(-1, 1)).astype(np.float16).astype(np.float32)
        got = kx.gemm_w8_rows(X, w, s, GS, split_ws=sp, half=True)
        want = kxo.gemm_w8(X, w, s, GS, threads=8)
        sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
        tol = 2.0 ** -11 * (np.abs(X.astype(np.float64)) @ sd.T) + 1e-7 * np.abs(want).max()
        report("gemm_w8_rows", (T, K, M), np.abs(got - want), tol)
    assert not sp.numpy()[:4096].any(), "arrival tickets not back at zero"

    # ---- tiled prompt GEMMs: int8 single pass, two-pass, and the fp16-activation form (fp16 weight
    # images: one more rounding per weight, scales spread over three decades) ----
    for _ in range(N):
        T = int(rng.integers(1, 600))
        K = int(rng.integers(1, 10)) * 64
        M = int(rng.integers(1, 12)) * 64
        w, s = w8(K, M, 1e-5, 1e-2)
        X = (rng.standard_normal((T, M)) * 10.0 ** rng.uniform(-1, 1)).astype(np.float16).astype(np.float32)
        want = kxo.gemm_w8(X, w, s, GS, threads=8)
        sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
        base = np.abs(X.astype(np.float64)) @ sd.T
        for name, got, ulps in (("gemm_w8", kx.gemm_w8(X, w, s, GS), 1.0),
                                ("gemm_w8 two-pass", kx.gemm_w8(X, w, s, GS, two_pass=True), 2.0),
                                ("gemm_w8_f16act", kx.gemm_w8_f16act(X.astype(np.float16), w, s, GS, split_ws=sp), 2.0)):
            report(name, (T, K, M), np.abs(got - want), ulps * 2.0 ** -11 * base + 1e-7 * np.abs(want).max())
    assert not sp.numpy()[:4096].any(), "arrival tickets not back at zero"

    # ---- attention: prompt pass and decode step over the same cache ----
    for _ in range(N):
        hs = int(rng.choice([32, 64, 128]))
        KVH = int(rng.choice([1, 2, 4]))
        H = KVH * int(rng.choice([1, 2, 4, 7]))
        Tq = int(rng.integers(1, 400))
        pos = Tq - 1 + int(rng.integers(0, 100))
        S = pos + 1 + int(rng.integers(0, 50))
        spread = 10.0 ** rng.uniform(-0.5, 0.7)          # flat to sharply peaked score rows
        kc = (rng.standard_normal((1, S, KVH * hs)) * spread).astype(np.float32)
        vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
        q = (rng.standard_normal((Tq, H, hs)) * spread).astype(np.float32)
        want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, 0, threads=8)
        for f16_copies in (True, False):
            got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0, f16_copies=f16_copies)
            case = (H, KVH, hs, Tq, pos, round(float(spread), 2), f16_copies)
            if not np.isfinite(got).all():
                bad += 1
                print(f"NON-FINITE flash_attention {case}", flush=True)
                continue
            # fp16 rounding of q, k shifts a score by up to 2^-10 |q||k|: the bound follows the spread
            report("flash_attention", case, np.abs(got - want), 3e-3 * max(1.0, spread * spread) * np.ones_like(want))
        p_ = int(rng.integers(0, pos + 1))
        qd = q[min(p_, Tq - 1)].reshape(-1)
        one = kx.mha_decode(qd, kc, vc, p_, H, H // KVH, hs, 0)
        ref = kxo.mha(p_, H, 0, S, KVH * hs, H // KVH, hs, qd, kc, vc).reshape(H, hs)
        report("mha_decode", (H, KVH, hs, p_), np.abs(one - ref), 1e-4 * np.ones_like(ref))

    # ---- two GEMVs in one launch (kx_gemv_w8_chain_f32): [plain + residual -> RMSNorm -> gate|up +
    # SwiGLU] and [plain + residual -> RMSNorm -> q|k|v + bias + RoPE + K/V placement] ----
    L = kx.lib()
    for _ in range(max(1, N // 4)):
        hs = int(rng.choice([32, 64, 128]))
        KVH = int(rng.choice([1, 2, 4]))
        H = KVH * int(rng.choice([1, 2, 4]))
        dim = 1024 * int(rng.integers(1, 4))
        hid = 1024 * int(rng.integers(1, 4))
        ql, kl = H * hs, KVH * hs
        if ql % 1024 or (ql + 2 * kl) % 8:
            ql = 1024                           # the first op's input width must be whole 1024-column steps
            H = ql // hs
            KVH = min(KVH, H)
            H = H // KVH * KVH
            ql, kl = H * hs, KVH * hs
            if ql % 1024:
                continue
        variant, pos = int(rng.integers(0, 2)), int(rng.integers(0, 50))
        S = pos + 2
        mats = dict(wo=w8(dim, ql, 1e-4, 3e-3), w13=w8(2 * hid, dim, 1e-4, 3e-3), w2=w8(dim, hid, 1e-4, 3e-3),
                    qkv=w8(ql + 2 * kl, dim, 1e-4, 3e-3))
        att, x0 = rng.standard_normal(ql).astype(np.float32), rng.standard_normal(dim).astype(np.float32)
        g1, g2 = (rng.uniform(0.5, 1.5, dim).astype(np.float32) for _ in range(2))
        bias = (rng.standard_normal(ql + 2 * kl) * 0.1).astype(np.float32)
        sin, cos = kxo.sin_cos_cache(hs, S, 1e4)
        dev = {k: (kx.Buf.from_numpy(a_), kx.Buf.from_numpy(b_)) for k, (a_, b_) in mats.items()}
        datt, dg1, dg2, dbias, x = (kx.Buf.from_numpy(a_) for a_ in (att, g1, g2, bias, x0))
        h, q = kx.Buf.zeros(hid), kx.Buf.zeros(ql)
        kc, vc = kx.Buf.zeros(S * kl), kx.Buf.zeros(S * kl)
        dsin, dcos = kx.Buf.from_numpy(sin), kx.Buf.from_numpy(cos)
        ws = kx.Buf(L.kx_gemv_w8_chain_workspace_bytes(max(dim, hid)))
        kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, None), "reset")
        d_wo = kx.make_gemv_desc(kx.GEMV_PLAIN, datt.ptr, dev["wo"][0].ptr, dev["wo"][1].ptr, x.ptr, ql, dim, residual=x.ptr)
        d_w13 = kx.make_gemv_desc(kx.GEMV_GATE_UP, None, dev["w13"][0].ptr, dev["w13"][1].ptr, h.ptr, dim, 2 * hid,
                                  norm_w=dg1.ptr, norm_eps=1e-5)
        d_w2 = kx.make_gemv_desc(kx.GEMV_PLAIN, h.ptr, dev["w2"][0].ptr, dev["w2"][1].ptr, x.ptr, hid, dim, residual=x.ptr)
        d_qkv = kx.make_gemv_desc(kx.GEMV_QKV_ROPE, None, dev["qkv"][0].ptr, dev["qkv"][1].ptr, q.ptr, dim, ql + 2 * kl,
                                  norm_w=dg2.ptr, norm_eps=1e-5, y1=kc.ptr, y2=vc.ptr, k0=ql, k1=ql + kl, bias=dbias.ptr,
                                  sin=dsin.ptr, cos=dcos.ptr, pos=pos, rope_variant=variant, head_size=hs, kv_pos_stride=kl)
        x1 = kxo.gemv_w8(att, *mats["wo"], GS) + x0
        h13 = kxo.gemv_w8(kxo.rmsnorm(x1, g1, 1e-5), *mats["w13"], GS)
        hh = kxo.swiglu(h13[:hid], h13[hid:])
        x2 = kxo.gemv_w8(hh, *mats["w2"], GS) + x1
        qkv_o = kxo.gemv_w8(kxo.rmsnorm(x2, g2, 1e-5), *mats["qkv"], GS) + bias
        qo, ko = kxo.rope(variant, ql, kl, hs, qkv_o[:ql].copy(), qkv_o[ql:ql + kl].copy(), pos, sin, cos)
        case = (dim, hid, H, KVH, hs, variant, pos)
        # every seam of a token needs its own chain_id (the tag of its granules): two different ones
        id_a = int(rng.integers(0, 128))
        kx.gemv_chain([d_wo, d_w13], ws, chain_id=id_a, pos=pos)
        report("chain Wo", case, np.abs(x.numpy() - x1), 3e-5 * max(1.0, np.abs(x1).max()) * np.ones_like(x1))
        report("chain W1|W3", case, np.abs(h.numpy() - hh), 3e-5 * max(1.0, np.abs(hh).max()) * np.ones_like(hh))
        id_b = (id_a + 1 + int(rng.integers(0, 127))) % 128
        kx.gemv_chain([d_w2, d_qkv], ws, chain_id=id_b, pos=pos)
        report("chain W2", case, np.abs(x.numpy() - x2), 3e-5 * max(1.0, np.abs(x2).max()) * np.ones_like(x2))
        report("chain q", case, np.abs(q.numpy() - qo), 3e-5 * max(1.0, np.abs(qo).max()) * np.ones_like(qo))
        report("chain k", case, np.abs(kc.numpy().reshape(S, kl)[pos] - ko), 3e-5 * max(1.0, np.abs(ko).max()) * np.ones_like(ko))
        report("chain v", case, np.abs(vc.numpy().reshape(S, kl)[pos] - qkv_o[ql + kl:]),
               3e-5 * max(1.0, np.abs(qkv_o).max()) * np.ones(kl))

    print(f"{7 * N} kernel runs (+ {max(1, N // 4)} chained launches) checked, {bad} mismatches", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
