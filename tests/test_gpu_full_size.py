"""GPU parity at the full sizes of the benchmark configurations: every GEMV dispatch form the
model shapes select (rows sampled against the oracle), the fused RMSNorm + Wq|Wk|Wv + RoPE + cache
write at model size, flash attention at 4096 / 8192 query rows (sampled rows against the oracle,
poisoned rows past the position, last row against the decode kernel), and prompts longer than one
4096-row pass.  Tolerances as in test_gpu_kernels.py: fp32 paths 1e-5 * sum|terms|; MFMA prompt
paths (fp16 operands) 3e-3 absolute on N(0,1) attention data, 4e-3 of the tensor's range on model
rows."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kx  # noqa: E402
from kx import host  # noqa: E402
from oracle import kxo  # noqa: E402

GS = 64


def _w8(rng, K, M):
    w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
    s = rng.uniform(1e-4, 3e-4, size=(K * M) // GS).astype(np.float32)
    return w, s


def _abs_sum(x, w, s):
    K, M = w.shape
    sd = np.repeat(s.astype(np.float64), GS).reshape(K, M)
    return (np.abs(x.astype(np.float64))[None, :] * sd * np.abs(w.astype(np.float64))).sum(1)


def _rows_of(w, s, rows):
    M = w.shape[1]
    g = M // GS
    return w[rows], s.reshape(-1, g)[rows].reshape(-1)


# ---- W8 GEMV: the (rows per wave, loads in flight, block size) form each model shape selects ----
@pytest.mark.parametrize("K,M,what", [
    (8192, 2048, "Llama-3.2-1B W1 or W3 alone"), (2048, 8192, "1B W2"),
    (128256, 2048, "1B classifier"), (128256, 4096, "8B classifier"), (4096, 14336, "8B W2"),
    (8192, 3584, "70B W2 per rank at TP=8"), (8192, 1024, "70B Wo per rank at TP=8"),
    (8192, 8192, "70B Wo"), (8192, 28672, "70B W2"), (3584, 18944, "Qwen2.5-7B W2"),
    (152064, 3584, "Qwen2.5-7B classifier"), (3584, 9472, "Qwen W2 per rank at TP=2")])
def test_gemv_w8_model_shapes_sampled_rows(K, M, what):
    rng = np.random.default_rng(K + M)
    w, s = _w8(rng, K, M)
    x = rng.standard_normal(M).astype(np.float32)
    got = kx.gemv_w8(x, w, s, GS)
    rows = np.unique(np.concatenate([[0, 1, K - 2, K - 1], rng.choice(K, 96, replace=False)]))
    wr, sr = _rows_of(w, s, rows)
    want = kxo.gemv_w8(x, wr, sr, GS)
    assert np.all(np.abs(got[rows] - want) <= 1e-5 * _abs_sum(x, wr, sr)), what
    assert np.isfinite(got).all()


@pytest.mark.parametrize("H,M,what", [
    (8192, 2048, "Llama-3.2-1B W1|W3"), (18944, 3584, "Qwen2.5-7B W1|W3"),
    (3584, 8192, "70B W1|W3 per rank at TP=8"), (28672, 8192, "70B W1|W3")])
def test_gemv_w8_gate_up_model_shapes_sampled_rows(H, M, what):
    rng = np.random.default_rng(H + M)
    w, s = _w8(rng, 2 * H, M)
    s *= 20
    x = rng.standard_normal(M).astype(np.float32)
    dx, dw, ds = map(kx.Buf.from_numpy, (x, w, s))
    out = kx.Buf.zeros(H)
    kx.check(kx.lib().kx_gemv_w8_swiglu_f32(dx.ptr, dw.ptr, ds.ptr, GS, out.ptr, M, H, None))
    got = out.numpy()
    rows = np.unique(np.concatenate([[0, H - 1], rng.choice(H, 64, replace=False)]))
    wg, sg = _rows_of(w, s, rows)
    wu, su = _rows_of(w, s, rows + H)
    g, u = kxo.gemv_w8(x, wg, sg, GS), kxo.gemv_w8(x, wu, su, GS)
    want = kxo.swiglu(g, u)
    tol = 1e-5 * (1.1 * np.abs(u) * _abs_sum(x, wg, sg) + np.abs(g) * _abs_sum(x, wu, su)) + 1e-7
    assert np.all(np.abs(got[rows] - want) <= tol), what


# ---- fused RMSNorm + Wq|Wk|Wv + bias + RoPE + cache-row write at model size --------------------------
@pytest.mark.parametrize("dim,H,KVH,hs,theta,eps,bias,what", [
    (4096, 32, 8, 128, 5e5, 1e-5, False, "Llama-3-8B"), (2048, 32, 8, 64, 5e5, 1e-5, False, "Llama-3.2-1B"),
    (8192, 8, 1, 128, 5e5, 1e-5, False, "70B rank at TP=8"), (3584, 28, 4, 128, 1e6, 1e-6, True, "Qwen2.5-7B"),
    (3584, 7, 1, 128, 1e6, 1e-6, True, "Qwen2.5-7B rank at TP=4")])
def test_fused_qkv_rope_model_size_vs_oracle(dim, H, KVH, hs, theta, eps, bias, what):
    rng = np.random.default_rng(dim + H)
    q_dim, kvd = H * hs, KVH * hs
    K = q_dim + 2 * kvd
    w, s = _w8(rng, K, dim)
    s *= 10
    x = rng.standard_normal(dim).astype(np.float32) * 2
    g = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float32)
    b = (0.3 * rng.standard_normal(K)).astype(np.float32) if bias else None
    S, pos = 64, 37
    sin, cos = kxo.sin_cos_cache(hs, S, theta)
    xn = kxo.rmsnorm(x, g, eps)
    full = kxo.gemv_w8(xn, w, s, GS, threads=8)
    if bias:
        full = full + b
    wq, wk = kxo.rope(kxo.ROPE_HALF_SPLIT, q_dim, kvd, hs, full[:q_dim], full[q_dim:q_dim + kvd],
                      pos, sin, cos)
    wv = full[q_dim + kvd:]
    bufs = [kx.Buf.from_numpy(a) for a in (x, w, s, g, sin, cos)]
    dx, dw, ds, dg, dsin, dcos = bufs
    db = kx.Buf.from_numpy(b) if bias else None
    dq, dk, dv = kx.Buf.zeros(q_dim), kx.Buf.zeros((S, kvd)), kx.Buf.zeros((S, kvd))
    dpos = kx.Buf.from_numpy(np.array([pos], np.int32))
    d = kx.GemvDesc(mode=kx.GEMV_QKV_ROPE, group_size=GS, M=dim, K=K, x=dx.ptr, w=dw.ptr,
                    scales=ds.ptr, norm_w=dg.ptr, norm_eps=eps, y0=dq.ptr, y1=dk.ptr, y2=dv.ptr,
                    k0=q_dim, k1=q_dim + kvd, bias=db.ptr if bias else None, sin_cache=dsin.ptr,
                    cos_cache=dcos.ptr, pos_dev=dpos.ptr, rope_variant=kx.ROPE_HALF_SPLIT,
                    head_size=hs, kv_pos_stride=kvd)
    kx.check(kx.lib().kx_gemv_w8_fused_f32(C.byref(d), None), what)
    a = _abs_sum(xn, w, s) + (np.abs(b) if bias else 0)
    # a rotated element mixes two rows: bound by the pair's summed magnitudes
    half = hs // 2

    def pair_tol(t):
        t = t.reshape(-1, 2, half)
        return (t.sum(1, keepdims=True) * np.ones((1, 2, 1))).reshape(-1)
    tq = 2e-5 * pair_tol(a[:q_dim]) + 1e-6
    tk = 2e-5 * pair_tol(a[q_dim:q_dim + kvd]) + 1e-6
    assert np.all(np.abs(dq.numpy() - wq) <= tq), what
    kc, vc = dk.numpy(), dv.numpy()
    assert np.all(np.abs(kc[pos] - wk) <= tk), what
    assert np.all(np.abs(vc[pos] - wv) <= 2e-5 * a[q_dim + kvd:] + 1e-6), what
    kc[pos] = 0
    vc[pos] = 0
    assert not kc.any() and not vc.any()   # only the position's cache row is written


# ---- flash attention at full prompt sizes ----------------------------------------------------------
def test_gemv_chain_llama3_8b_layer_shapes():
    # the two chains of a decode step as ONE launch each, at Llama-3-8B layer shapes (the opt-in
    # kx_gemv_w8_chain_f32): [Wo + residual -> RMSNorm -> W1|W3 + SwiGLU] and [W2 + residual ->
    # RMSNorm -> Wq|Wk|Wv + RoPE + K/V placement], every output against the oracle's step-by-step
    # formulas (fp32 FMA chains: 3e-5 of the vector's range)
    rng = np.random.default_rng(88)
    dim, hid, H, KVH, hs, pos = 4096, 14336, 32, 8, 128, 4100
    ql, kl, S = H * hs, KVH * hs, pos + 2
    L = kx.lib()

    def w8(K, M):
        return rng.integers(-127, 128, (K, M), dtype=np.int8), rng.uniform(1e-4, 3e-4, K * M // GS).astype(np.float32)
    m = dict(wo=w8(dim, ql), w13=w8(2 * hid, dim), w2=w8(dim, hid), qkv=w8(ql + 2 * kl, dim))
    att, x0 = rng.standard_normal(ql).astype(np.float32), rng.standard_normal(dim).astype(np.float32)
    g1, g2 = (rng.uniform(0.5, 1.5, dim).astype(np.float32) for _ in range(2))
    sin, cos = kxo.sin_cos_cache(hs, S, 5e5)
    dev = {k: (kx.Buf.from_numpy(a), kx.Buf.from_numpy(b)) for k, (a, b) in m.items()}
    datt, dg1, dg2, x = (kx.Buf.from_numpy(a) for a in (att, g1, g2, x0))
    h, q = kx.Buf.zeros(hid), kx.Buf.zeros(ql)
    kc, vc = kx.Buf.zeros(S * kl), kx.Buf.zeros(S * kl)
    dsin, dcos = kx.Buf.from_numpy(sin), kx.Buf.from_numpy(cos)
    pos_dev = kx.Buf.from_numpy(np.array([pos], np.int32))
    ws = kx.Buf(L.kx_gemv_w8_chain_workspace_bytes(hid))
    kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, None), "reset")
    wo = kx.make_gemv_desc(kx.GEMV_PLAIN, datt.ptr, dev["wo"][0].ptr, dev["wo"][1].ptr, x.ptr, ql, dim, residual=x.ptr)
    w13 = kx.make_gemv_desc(kx.GEMV_GATE_UP, None, dev["w13"][0].ptr, dev["w13"][1].ptr, h.ptr, dim, 2 * hid,
                            norm_w=dg1.ptr, norm_eps=1e-5)
    w2 = kx.make_gemv_desc(kx.GEMV_PLAIN, h.ptr, dev["w2"][0].ptr, dev["w2"][1].ptr, x.ptr, hid, dim, residual=x.ptr)
    qkv = kx.make_gemv_desc(kx.GEMV_QKV_ROPE, None, dev["qkv"][0].ptr, dev["qkv"][1].ptr, q.ptr, dim, ql + 2 * kl,
                            norm_w=dg2.ptr, norm_eps=1e-5, y1=kc.ptr, y2=vc.ptr, k0=ql, k1=ql + kl, sin=dsin.ptr,
                            cos=dcos.ptr, pos_dev=pos_dev.ptr, rope_variant=1, head_size=hs, kv_pos_stride=kl)
    x1 = kxo.gemv_w8(att, *m["wo"], GS, threads=8) + x0
    h13 = kxo.gemv_w8(kxo.rmsnorm(x1, g1, 1e-5), *m["w13"], GS, threads=8)
    hh = kxo.swiglu(h13[:hid], h13[hid:])
    x2 = kxo.gemv_w8(hh, *m["w2"], GS, threads=8) + x1
    qkv_o = kxo.gemv_w8(kxo.rmsnorm(x2, g2, 1e-5), *m["qkv"], GS, threads=8)
    qo, ko = kxo.rope(1, ql, kl, hs, qkv_o[:ql].copy(), qkv_o[ql:ql + kl].copy(), pos, sin, cos)

    def close(got, want, what):
        assert np.abs(got - want).max() <= 3e-5 * max(1.0, np.abs(want).max()), what
    kx.gemv_chain([wo, w13], ws, chain_id=1, pos_dev=pos_dev.ptr)
    close(x.numpy(), x1, "Wo + residual")
    close(h.numpy(), hh, "W1|W3 + SwiGLU behind the seam")
    kx.gemv_chain([w2, qkv], ws, chain_id=2, pos_dev=pos_dev.ptr)
    close(x.numpy(), x2, "W2 + residual")
    close(q.numpy(), qo, "q behind the seam")
    close(kc.numpy().reshape(S, kl)[pos], ko, "k row")
    close(vc.numpy().reshape(S, kl)[pos], qkv_o[ql + kl:], "v row")
    assert not kc.numpy().reshape(S, kl)[:pos].any() and not vc.numpy().reshape(S, kl)[pos + 1:].any()
    gave = C.c_int32(1)
    kx.check(L.kx_gemv_w8_chain_status(ws.ptr, C.byref(gave)), "status")
    assert gave.value == 0


@pytest.mark.parametrize("H,KVH,hs,Tq,what", [
    (32, 8, 128, 4096, "Llama-3-8B, 4096-row prompt"), (8, 1, 128, 8192, "70B rank at TP=8, 8192 rows"),
    (32, 8, 64, 4096, "Llama-3.2-1B"), (28, 4, 128, 4096, "Qwen2.5-7B"),
    (14, 2, 128, 4096, "Qwen2.5-7B rank at TP=2"), (7, 1, 128, 4096, "Qwen2.5-7B rank at TP=4"),
    (64, 8, 128, 2048, "70B unsharded heads")])
def test_flash_attention_full_size_sampled_rows(H, KVH, hs, Tq, what):
    rng = np.random.default_rng(H * 1000 + Tq)
    S, pos = Tq + 64, Tq - 1
    kvd = KVH * hs
    kc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    vc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    q = rng.standard_normal((Tq, H, hs)).astype(np.float32)
    kc[0, pos + 1:] = np.nan   # rows past the prompt are poison: never read
    vc[0, pos + 1:] = np.nan
    got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0)
    assert np.isfinite(got).all(), what
    kz, vz = np.nan_to_num(kc), np.nan_to_num(vc)
    rows = np.unique(np.concatenate([[0, 1, 15, 16, 63, 64, Tq // 2, Tq - 2, Tq - 1],
                                     rng.choice(Tq, 12, replace=False)]))
    err2, n = 0.0, 0
    for i in rows:
        want = kxo.mha(int(i), H, 0, S, kvd, H // KVH, hs, q[i].reshape(-1), kz, vz).reshape(H, hs)
        np.testing.assert_allclose(got[i], want, rtol=0, atol=3e-3, err_msg=f"{what}: row {i}")
        err2 += ((got[i] - want) ** 2).sum()
        n += want.size
    assert np.sqrt(err2 / n) <= 4e-4
    # the last row is what the decode kernel computes at that position
    last = kx.mha_decode(q[-1].reshape(-1), kz, vz, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got[-1], last, rtol=0, atol=3e-3)
    # fp16 output rows (the form the long-prompt pipeline uses): the same numbers rounded once
    got16 = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0, f16_out=True)
    np.testing.assert_allclose(got16[rows].astype(np.float32), got[rows], rtol=2.0 ** -10, atol=1e-6)


# ---- prompts longer than one 4096-row pass -------------------------------------------------------------
def _layer0_kv_rows_oracle(m, shape, toks, rows, eps, theta):
    """K (post-RoPE) and V cache rows of layer 0 for the sampled prompt positions: they depend on
    the token's embedding only, so the oracle can produce them without running the prompt."""
    dim, hs = shape["dim"], shape["dim"] // shape["head_num"]
    kvd = shape["kv_head_num"] * hs
    emb, g = m.read_weight("emb"), m.read_weight("attn_norm", 0)
    wk, sk = m.read_weight("wk", 0), m.read_weight("wk_s", 0)
    wv, sv = m.read_weight("wv", 0), m.read_weight("wv_s", 0)
    sin, cos = kxo.sin_cos_cache(hs, int(max(rows)) + 1, theta)
    ks, vs = [], []
    for p in rows:
        xn = kxo.rmsnorm(emb[toks[p]], g, eps)
        k = kxo.gemv_w8(xn, wk, sk, GS, threads=8)
        _, k = kxo.rope(kxo.ROPE_HALF_SPLIT, kvd, kvd, hs, np.zeros(kvd, np.float32), k, int(p), sin, cos)
        ks.append(k)
        vs.append(kxo.gemv_w8(xn, wv, sv, GS, threads=8))
    return np.stack(ks), np.stack(vs)


@pytest.mark.parametrize("shape,n,what", [
    (dict(dim=4096, hidden_dim=14336, layer_num=1, head_num=32, kv_head_num=8, vocab_size=512,
          rope_variant=1, rope_theta=5e5, rms_eps=1e-5), 4200, "one Llama-3-8B layer, 4200-row prompt"),
    (dict(dim=2048, hidden_dim=8192, layer_num=2, head_num=32, kv_head_num=8, vocab_size=512,
          rope_variant=1, rope_theta=5e5, rms_eps=1e-5), 4500, "two Llama-3.2-1B layers, 4500 rows"),
])
def test_two_pass_prompt_vs_sequential_decode_and_oracle_rows(shape, n, what):
    seq_len = n + 8
    rng = np.random.default_rng(n)
    toks = rng.integers(0, shape["vocab_size"], n).astype(np.int32)
    m = host.Model(seq_len=seq_len, group_size=GS, **shape)
    m.init_synthetic(17)
    last = m.prefill(toks)                       # 4096 rows, then the rest
    ref = host.Model(seq_len=seq_len, group_size=GS, **shape)
    ref.init_synthetic(17)
    want_last = ref.prefill(toks, exact=True)    # one fp32 decode step per row
    assert np.abs(last - want_last).max() <= 4e-3 * np.abs(want_last).max(), what
    for layer in range(shape["layer_num"]):
        k, v = m.read_kv(layer, n)
        kr, vr = ref.read_kv(layer, n)
        assert np.abs(k - kr).max() <= 4e-3 * np.abs(kr).max(), (what, layer)
        assert np.abs(v - vr).max() <= 4e-3 * np.abs(vr).max(), (what, layer)
    # layer 0 rows on both sides of the pass boundary against the oracle itself
    rows = np.array([0, 1, 4095, 4096, 4097, n - 1])
    ko, vo = _layer0_kv_rows_oracle(ref, shape, toks, rows, shape["rms_eps"], shape["rope_theta"])
    kr, vr = ref.read_kv(0, n)
    k, v = m.read_kv(0, n)
    # K carries the rotation: the device's sin/cos table differs from libm's by up to 2e-4 at
    # angles in the thousands of radians (an ulp of the fp32 angle pos * freq at pos ~ 4000 is
    # 2.4e-4 rad; the reference's own RoPE test allows 1e-3, test_cu_rope.cpp:43)
    assert np.abs(kr[rows] - ko).max() <= 2e-4 * np.abs(ko).max()
    assert np.abs(vr[rows] - vo).max() <= 3e-5 * np.abs(vo).max()
    assert np.abs(k[rows] - ko).max() <= 4e-3 * np.abs(ko).max()
    assert np.abs(v[rows] - vo).max() <= 4e-3 * np.abs(vo).max()
    assert m.state()[1] == n and ref.state()[1] == n
    m.close()
    ref.close()


# ---- Llama-3-70B at TP=8: an 8192-row prompt over the per-rank shapes ----------------------------------
def _run_ranks(models, fn):
    """fn(rank, model) on one host thread per rank (ctypes releases the GIL); the ranks meet inside
    the loopback all-reduce."""
    import threading
    out, err = [None] * len(models), []

    def work(r):
        try:
            out[r] = fn(r, models[r])
        except Exception as e:  # noqa: BLE001
            err.append(e)
    threads = [threading.Thread(target=work, args=(r,)) for r in range(len(models))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(170)
    assert not err, err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


def test_70b_layer_tp8_8192_row_prompt_and_decode():
    # One Llama-3-70B layer.  Every rank of an 8-way split holds exactly the per-GPU shapes of the
    # TP=8 configuration (QKV 1280 x 8192, Wo 8192 x 1024, W1|W3 7168 x 8192, W2 8192 x 3584, one kv
    # head of 128, 8 query heads); the ranks live in this process and sum through a host rendezvous.
    # The 8192-row prompt takes two 4096-row passes; then greedy steps at a context past 8192.
    shape = dict(dim=8192, hidden_dim=28672, layer_num=1, head_num=64, kv_head_num=8, vocab_size=512,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    n, tp = 8192, 8
    seq_len = n + 16
    toks = np.random.default_rng(70).integers(0, 512, n).astype(np.int32)
    one = host.Model(seq_len=seq_len, group_size=GS, **shape)
    one.init_synthetic(23)
    want_last = one.prefill(toks)
    # the unsharded batched pass against one fp32 decode step per row, and against the oracle
    ref = host.Model(seq_len=seq_len, group_size=GS, **shape)
    ref.init_synthetic(23)
    exact_last = ref.prefill(toks, exact=True)
    assert np.abs(want_last - exact_last).max() <= 4e-3 * np.abs(exact_last).max()
    kf, vf = one.read_kv(0, n)
    kr, vr = ref.read_kv(0, n)
    assert np.abs(kf - kr).max() <= 4e-3 * np.abs(kr).max()
    assert np.abs(vf - vr).max() <= 4e-3 * np.abs(vr).max()
    rows = np.array([0, 4095, 4096, n - 1])
    ko, vo = _layer0_kv_rows_oracle(ref, shape, toks, rows, 1e-5, 5e5)
    assert np.abs(kr[rows] - ko).max() <= 4e-4 * np.abs(ko).max()   # sin/cos table at pos ~ 8000
    assert np.abs(vr[rows] - vo).max() <= 3e-5 * np.abs(vo).max()
    ref.close()
    group = host.loopback_create(tp)
    ranks = [host.Model(seq_len=seq_len, group_size=GS, tp_size=tp, tp_rank=r, **shape)
             for r in range(tp)]
    for m in ranks:
        m.init_synthetic(23)
        m.join_loopback(group)
    outs = _run_ranks(ranks, lambda r, m: (m.prefill(toks), m.decode(4, use_graph=False),
                                           m.history(n + 5)))
    one.decode(4, use_graph=False)
    for r in range(tp):
        np.testing.assert_array_equal(outs[r][0], outs[0][0])
        np.testing.assert_array_equal(outs[r][2], outs[0][2])
    assert np.abs(outs[0][0] - want_last).max() <= 2e-3 * np.abs(want_last).max()
    kvl = ranks[0].kv_dim_local
    assert kvl == 128 and ranks[0].q_dim_local == 1024 and ranks[0].hidden_local == 3584
    for r, m in enumerate(ranks):
        k, v = m.read_kv(0, n)
        # same inputs, same kernels on the rank's own columns: layer 0 rows are the column slice
        np.testing.assert_allclose(k, kf[:, r * kvl:(r + 1) * kvl], rtol=0, atol=1e-4 * np.abs(kf).max())
        np.testing.assert_allclose(v, vf[:, r * kvl:(r + 1) * kvl], rtol=0, atol=1e-4 * np.abs(vf).max())
    order = np.argsort(want_last)
    if want_last[order[-1]] - want_last[order[-2]] > 1e-2 * np.abs(want_last).max():
        assert outs[0][2][n] == one.history(n + 5)[n]
    for m in ranks:
        m.close()
    one.close()
    host.loopback_destroy(group)
