"""GPU parity tests at the layer and model level: op::MatmulLayer / MultiHeadAttention through the
kernel registry on DeviceType::kDeviceHIP, the reference-structured token loop, and the fused
graph-replayed decode loop -- against the golden fixtures generated from the reference's Python
and against the CPU oracle on seeded synthetic models."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kx  # noqa: E402
from kx import host  # noqa: E402
from oracle import kxo  # noqa: E402

GS = 64
TINY = dict(dim=128, hidden_dim=256, layer_num=2, head_num=4, kv_head_num=2, vocab_size=128,
            rope_variant=0, rope_theta=1e4, rms_eps=1e-5)


# ---------------- layer level (reads like test/test_op/test_cu_matmul.cpp, test_load.cpp) --------
def test_registry_has_every_hip_kernel():
    assert host.hlib().kxm_test_registry() == 1


def test_matmul_layer_reference_known_answers():
    # test_cu_matmul.cpp:48-106 through op::MatmulLayer on kDeviceHIP
    w = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    assert host.matmul_layer(np.array([1, 1, -1], np.float32), w).tolist() == [0.0, 3.0, 6.0]
    # test_load.cpp:75-105
    w = np.arange(2048, dtype=np.float32).reshape(16, 128)
    y = host.matmul_layer(np.ones(128, np.float32), w)
    assert y[0] == 8128 and y[1] == 24512 and y[14] == 237504 and y[15] == 253888


def test_matmul_layer_quant_blob_contract_and_bias(golden):
    # int8 weights with the fp32 scales right behind them (layer.cpp:203-216) + has_bias
    g = golden("g1_quant_gemv.npz")
    x, q, s = g["c_x"], g["c_q"], g["c_s"]
    bias = np.linspace(-1, 1, q.shape[0]).astype(np.float32)
    y = host.matmul_layer(x, q, s, GS, bias)
    want = g["c_y64"] + bias
    K, M = q.shape
    absum = (np.abs(x.astype(np.float64))[None, :] * np.repeat(s.astype(np.float64), GS).reshape(K, M)
             * np.abs(q.astype(np.float64))).sum(1)
    assert np.all(np.abs(y - want) <= 1e-5 * absum + 1e-6)


def test_mha_layer_golden(golden):
    g = golden("g5_attention.npz")
    meta, q, want = g["c3_meta"], g["c3_q"], g["c3_out"]
    H, KVH, hs, S, pos, Tq, L, layer, cid = [int(v) for v in meta]
    got = host.mha_layer(q.reshape(-1), g[f"cache{cid}_k"], g[f"cache{cid}_v"], H, H // KVH, hs,
                         layer, pos)
    np.testing.assert_allclose(got.reshape(1, H, hs), want, rtol=0, atol=1e-5)


def test_flash_layer_single_row_equals_mha(golden):
    g = golden("g5_attention.npz")
    meta, q, want = g["c2_meta"], g["c2_q"], g["c2_out"]
    H, KVH, hs, S, pos, Tq, L, layer, cid = [int(v) for v in meta]
    got = host.flash_layer(q, g[f"cache{cid}_k"], g[f"cache{cid}_v"], H, H // KVH, hs, layer, pos)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)


def test_kv_cache_manager_layout():
    # each block [layer][seq_len][kv_dim]; V block right after the K block
    L, S, KVH, hs = 4, 96, 2, 64
    kvd = KVH * hs
    for layer in range(L):
        k, v, total = host.kv_layout(L, S, KVH, hs, layer)
        assert k == layer * S * kvd * 4
        assert v == (L + layer) * S * kvd * 4
        assert total == 2 * L * S * kvd * 4


# ---------------- golden tiny models ----------------------------------------------------------------
def _stream(step, tokens, n_greedy):
    logits = [step(int(t), pos) for pos, t in enumerate(tokens)]
    greedy, pos = [], len(tokens)
    nxt = int(np.argmax(logits[-1]))
    for _ in range(n_greedy):
        greedy.append(nxt)
        lg = step(nxt, pos)
        pos += 1
        nxt = int(np.argmax(lg))
    return np.stack(logits), greedy


def test_tiny_fp32_reference_structured_path(golden, golden_bytes):
    # configs[0] semantics (stories-style fp32, interleaved RoPE) through op::Layer + registry
    g = golden("g3_tiny_fp32.npz")
    m = host.Model(seq_len=32, group_size=0, **TINY)
    m.load_blob(golden_bytes("g3_tiny_fp32.bin"))
    logits, greedy = _stream(m.forward_layers, g["tokens"], 16)
    scale = np.abs(g["logits"]).max()
    assert np.abs(logits - g["logits"]).max() <= 2e-5 * scale
    assert greedy == g["greedy"].tolist()


def test_tiny_q8_both_paths_vs_golden(golden, golden_bytes):
    g = golden("g4_tiny_q8.npz")
    scale = np.abs(g["logits"]).max()
    for path in ("forward_layers", "step_logits"):
        m = host.Model(seq_len=32, group_size=GS, **TINY)
        m.load_blob(golden_bytes("g4_tiny_q8.bin"))
        logits, greedy = _stream(getattr(m, path), g["tokens"], 16)
        assert np.abs(logits - g["logits"]).max() <= 2e-5 * scale, path
        assert greedy == g["greedy"].tolist(), path


@pytest.mark.parametrize("exact", [True, False])
def test_tiny_q8_prefill_then_graph_decode_token_exact(golden, golden_bytes, exact):
    # exact: one fp32 decode step per prompt row (tolerance of the decode path).  Batched: MFMA
    # GEMMs on fp16-rounded activations + fp16 flash attention, tolerance 3e-3 of the logit range;
    # the golden greedy stream has top-1 margins far above that.
    g = golden("g4_tiny_q8.npz")
    toks = g["tokens"]
    tol = 2e-5 if exact else 3e-3
    oracle = kxo.Model(kxo.Config(rope_variant=0, rope_theta=1e4, rms_eps=1e-5),
                       blob=golden_bytes("g4_tiny_q8.bin"), quant=True)
    for use_graph in (False, True):
        m = host.Model(seq_len=32, group_size=GS, **TINY)
        m.load_blob(golden_bytes("g4_tiny_q8.bin"))
        last = m.prefill(toks, exact=exact)
        assert np.abs(last - g["logits"][-1]).max() <= tol * np.abs(g["logits"]).max()
        m.decode(15, use_graph=use_graph)
        hist = m.history(len(toks) + 16)
        assert hist[:len(toks)].tolist() == toks.tolist()
        assert hist[len(toks):].tolist() == g["greedy"].tolist()
        tok, pos = m.state()
        assert pos == len(toks) + 15
    # KV cache after the run equals the oracle's cache row for row
    for pos, t in enumerate(hist[:-1]):
        oracle.forward(int(t), pos)
    ok_, ov_ = oracle.caches()
    for layer in range(2):
        k, v = m.read_kv(layer, len(hist) - 1)
        n = len(hist) - 1
        np.testing.assert_allclose(k, ok_[layer, :n], rtol=0, atol=tol * np.abs(ok_[layer, :n]).max())
        np.testing.assert_allclose(v, ov_[layer, :n], rtol=0, atol=tol * np.abs(ov_[layer, :n]).max())


def test_captured_step_is_renewed_past_its_attention_bound(monkeypatch):
    # the captured step's attention launches are cut for an upper bound of the position (the
    # position itself only masks) and serve a span of positions; past the bound, or when the
    # context has become much shorter, the step is captured again.  Spans of 8 and 1 position,
    # several decode calls, a reset to a shorter context and one to a longer one: the same tokens
    # as steps issued one by one, whose launches are cut for their exact position.  Split contexts
    # (160+ keys: several workgroups per kv head) included.
    shape = dict(dim=256, hidden_dim=512, layer_num=2, head_num=4, kv_head_num=2, vocab_size=500,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(5)
    long_prompt = rng.integers(0, 500, 150).astype(np.int32)
    short_prompt = rng.integers(0, 500, 9).astype(np.int32)
    eager = host.Model(seq_len=400, group_size=GS, **shape)
    eager.init_synthetic(3)
    for span in ("8", "1", "64"):
        monkeypatch.setenv("KX_ATTN_SPAN", span)
        m = host.Model(seq_len=400, group_size=GS, **shape)
        m.init_synthetic(3)
        for prompt, calls in ((long_prompt, (3, 30, 1, 40)), (short_prompt, (25, 2)), (long_prompt, (100,))):
            m.prefill(prompt, exact=True)
            eager.prefill(prompt, exact=True)
            for n in calls:
                m.decode(n, use_graph=True)
                eager.decode(n, use_graph=False)
                assert m.state() == eager.state(), (span, len(prompt), n)
            total = len(prompt) + sum(calls)
            assert m.history(total + 1).tolist() == eager.history(total + 1).tolist(), (span, len(prompt))
        m.close()
    eager.close()


# ---------------- half-split RoPE families: fixtures written by the reference's HF exporters -------
# g6 / g7 (tests/golden/make_golden_hf.py): files from tools/export_llama3.py / export_qwen2.py,
# expected logits and greedy streams computed by the Hugging Face models themselves.
LLAMA3_TINY = dict(dim=256, hidden_dim=512, layer_num=2, head_num=4, kv_head_num=2, vocab_size=128,
                   rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
QWEN2_TINY = dict(dim=128, hidden_dim=256, layer_num=2, head_num=2, kv_head_num=1, vocab_size=128,
                  rope_variant=1, rope_theta=1e6, rms_eps=1e-6)


def _prefill_decode_exact(make_model, g, logits_key="logits", greedy_key="greedy"):
    toks = g["tokens"]
    scale = np.abs(g[logits_key]).max()
    for exact, tol in ((True, 2e-5), (False, 3e-3)):
        for use_graph in (False, True):
            m = make_model()
            last = m.prefill(toks, exact=exact)
            assert np.abs(last - g[logits_key][-1]).max() <= tol * scale, (exact, use_graph)
            m.decode(15, use_graph=use_graph)
            hist = m.history(len(toks) + 16)
            assert hist[:len(toks)].tolist() == toks.tolist()
            assert hist[len(toks):].tolist() == g[greedy_key].tolist(), (exact, use_graph)
            m.close()


def test_llama3_q8_hf_fixture_every_path(golden, golden_bytes):
    g = golden("g6_llama3_q8.npz")
    blob = golden_bytes("g6_llama3_q8.bin")
    scale = np.abs(g["logits"]).max()

    def make():
        m = host.Model(seq_len=48, group_size=GS, **LLAMA3_TINY)
        m.load_blob(blob)
        return m
    for path in ("forward_layers", "step_logits"):
        m = make()
        logits, greedy = _stream(getattr(m, path), g["tokens"], 16)
        assert np.abs(logits - g["logits"]).max() <= 2e-5 * scale, path
        assert greedy == g["greedy"].tolist(), path
        m.close()
    _prefill_decode_exact(make, g)


LLAMA3_TIED_TINY = dict(dim=128, hidden_dim=384, layer_num=2, head_num=4, kv_head_num=2, vocab_size=192,
                        rope_variant=1, rope_theta=5e5, rms_eps=1e-5)


def test_llama3_q8_tied_classifier_fixture_every_path(golden, golden_bytes):
    # the exporter writes no int8 classifier when it is tied to the embedding (Llama-3.2-1B is);
    # load_blob quantises the fp32 table on the device with the exporter's rule: same bytes as the
    # exporter computes, logits and greedy stream of the HF model on every path
    g = golden("g8_llama3_tied_q8.npz")
    blob = golden_bytes("g8_llama3_tied_q8.bin")
    scale = np.abs(g["logits"]).max()

    def make():
        m = host.Model(seq_len=48, group_size=GS, **LLAMA3_TIED_TINY)
        m.load_blob(blob)
        return m
    m = make()
    np.testing.assert_array_equal(m.read_weight("cls"), g["cls_q"])
    np.testing.assert_array_equal(m.read_weight("cls_s"), g["cls_s"])
    m.close()
    for path in ("forward_layers", "step_logits"):
        m = make()
        logits, greedy = _stream(getattr(m, path), g["tokens"], 16)
        assert np.abs(logits - g["logits"]).max() <= 2e-5 * scale, path
        assert greedy == g["greedy"].tolist(), path
        m.close()
    _prefill_decode_exact(make, g)
    # over two tensor-parallel ranks each rank quantises its half of the table's rows
    halves = []
    for r in range(2):
        m = host.Model(seq_len=48, group_size=GS, tp_size=2, tp_rank=r, **LLAMA3_TIED_TINY)
        m.load_blob(blob)
        halves.append((m.read_weight("cls"), m.read_weight("cls_s")))
        m.close()
    np.testing.assert_array_equal(np.concatenate([h[0] for h in halves]), g["cls_q"])
    np.testing.assert_array_equal(np.concatenate([h[1] for h in halves]), g["cls_s"])


def test_quantize_q80_on_the_device_equals_the_oracle():
    rng = np.random.default_rng(80)
    w = (rng.standard_normal((300, 192)) * 10.0 ** rng.uniform(-3, 2, (300, 1))).astype(np.float32)
    w[7, :64] = 0                         # an all-zero group
    w[9, 64:128] = np.arange(64) * 0.5    # exact halves: ties go to even
    for gs in (64, 32, 128):
        q, s = kx.quantize_q80(w, gs)
        qo, so = kxo.quantize_q80(w, gs)
        np.testing.assert_array_equal(s, so)
        np.testing.assert_array_equal(q, qo)


def test_attention_assembled_from_the_registry_pieces(golden):
    # matmul-with-scale per key, softmax, scale-sum: the reference's CPU mha_kernel flow
    # (cpu/mha_kernel.cpp:10-60) run with kDeviceHIP kernels, against the oracle's attention
    rng = np.random.default_rng(12)
    H, KVH, hs, S, pos, layer = 6, 2, 48, 40, 33, 1
    kc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    got, twice = host.mha_from_registry_pieces(q, kc, vc, H, H // KVH, hs, layer, pos)
    want = kxo.mha(pos, H, layer, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    assert np.abs(got - want).max() <= 2e-5
    np.testing.assert_array_equal(twice, got * np.float32(2.0))


def test_qwen2_fp32_hf_fixture_with_biases_layer_path(golden, golden_bytes):
    # fp32 file with q/k/v biases (qwen2.cpp:306-333) through op::MatmulLayer(has_bias) + registry
    g = golden("g7_qwen2_fp32.npz")
    m = host.Model(seq_len=48, group_size=0, qkv_bias=True, **QWEN2_TINY)
    m.load_blob(golden_bytes("g7_qwen2_fp32.bin"))
    logits, greedy = _stream(m.forward_layers, g["tokens"], 16)
    assert np.abs(logits - g["logits"]).max() <= 2e-5 * np.abs(g["logits"]).max()
    assert greedy == g["greedy"].tolist()


@pytest.mark.parametrize("with_bias", [False, True])
def test_qwen2_q8_hf_fixture_every_path(golden, golden_bytes, with_bias):
    # the quant file carries no biases (qwen2.cpp:192-217); attached from the fp32 file's values
    # they go through the fused GEMV epilogue (decode) and the RoPE pass of the prompt path
    g = golden("g7_qwen2_q8.npz")
    blob = golden_bytes("g7_qwen2_q8.bin")
    lk, gk = ("logits_bias", "greedy_bias") if with_bias else ("logits", "greedy")
    scale = np.abs(g[lk]).max()

    def make():
        m = host.Model(seq_len=48, group_size=GS, qkv_bias=with_bias, **QWEN2_TINY)
        m.load_blob(blob)
        if with_bias:
            for layer in range(2):
                m.set_qkv_bias(layer, g["bq"][layer], g["bk"][layer], g["bv"][layer])
                np.testing.assert_array_equal(m.read_weight("bk", layer), g["bk"][layer])
        return m
    for path in ("forward_layers", "step_logits"):
        m = make()
        logits, greedy = _stream(getattr(m, path), g["tokens"], 16)
        assert np.abs(logits - g[lk]).max() <= 2e-5 * scale, path
        assert greedy == g[gk].tolist(), path
        m.close()
    _prefill_decode_exact(make, g, lk, gk)


def test_two_model_families_in_one_process(golden, golden_bytes):
    # RoPE pairing and RMS epsilon travel with each model's layers: an interleaved / 1e-5 model and
    # a half-split / 1e-6 model stepped alternately both keep matching their fixtures
    g4, g7 = golden("g4_tiny_q8.npz"), golden("g7_qwen2_q8.npz")
    a = host.Model(seq_len=32, group_size=GS, **TINY)
    a.load_blob(golden_bytes("g4_tiny_q8.bin"))
    b = host.Model(seq_len=48, group_size=GS, **QWEN2_TINY)
    b.load_blob(golden_bytes("g7_qwen2_q8.bin"))
    for pos in range(6):
        la = a.forward_layers(int(g4["tokens"][pos]), pos)
        lb = b.forward_layers(int(g7["tokens"][pos]), pos)
        assert np.abs(la - g4["logits"][pos]).max() <= 2e-5 * np.abs(g4["logits"]).max()
        assert np.abs(lb - g7["logits"][pos]).max() <= 2e-5 * np.abs(g7["logits"]).max()


def test_synthetic_qkv_bias_fused_and_layer_paths_vs_oracle():
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=600,
                 rope_variant=1, rope_theta=1e6, rms_eps=1e-6)
    seq_len = 600
    m = host.Model(seq_len=seq_len, group_size=GS, qkv_bias=True, **shape)
    m.init_synthetic(13)
    oracle = _oracle_from(m, shape, seq_len)
    oracle.set_bias(*[np.stack([m.read_weight(n, l) for l in range(2)]) for n in ("bq", "bk", "bv")])
    m2 = host.Model(seq_len=seq_len, group_size=GS, qkv_bias=True, **shape)
    m2.init_synthetic(13)
    toks = np.random.default_rng(2).integers(0, 600, 530).astype(np.int32)
    for pos, t in enumerate(toks):
        want = oracle.forward(int(t), pos)
        if pos < 5:
            tol = 3e-5 * max(1.0, np.abs(want).max())
            assert np.abs(m.step_logits(int(t), pos) - want).max() <= tol
            assert np.abs(m2.forward_layers(int(t), pos) - want).max() <= tol
    # batched prompt pass (fp16-activation GEMMs from 512 rows on): biases enter ahead of RoPE
    last = m.prefill(toks)
    assert np.abs(last - want).max() <= 4e-3 * np.abs(want).max()
    ok_, ov_ = oracle.caches()
    for layer in range(2):
        k, v = m.read_kv(layer, len(toks))
        assert np.abs(k - ok_[layer, :len(toks)]).max() <= 4e-3 * np.abs(ok_[layer]).max()
        assert np.abs(v - ov_[layer, :len(toks)]).max() <= 4e-3 * np.abs(ov_[layer]).max()
    # tensor-parallel ranks keep their heads' bias rows
    for r in range(2):
        mr = host.Model(seq_len=16, group_size=GS, tp_size=2, tp_rank=r, qkv_bias=True, **shape)
        mr.init_synthetic(13)
        np.testing.assert_array_equal(mr.read_weight("bq"), m.read_weight("bq")[r * 256:(r + 1) * 256])
        np.testing.assert_array_equal(mr.read_weight("bv"), m.read_weight("bv")[r * 64:(r + 1) * 64])
        mr.close()


# ---------------- drop-in hazards ---------------------------------------------------------------------
@pytest.mark.parametrize("H,KVH,hs,S,pos", [
    (32, 8, 128, 4608, 4200),   # score tensor large enough to serve as the split workspace
    (8, 2, 64, 128, 100),       # score tensor smaller than the ticket words: one split per kv head
    (4, 2, 48, 96, 70),         # head size outside the tiled kernel: score rows used as such
])
def test_mha_layer_without_reserved_workspace_and_garbage_scores(H, KVH, hs, S, pos):
    # a caller that never reserves the attention workspace hands over the reference's score tensor,
    # un-zeroed (llama3.cpp:523-525): the layer must give the right answer anyway, twice in a row
    rng = np.random.default_rng(pos)
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    got = host.mha_layer_no_workspace(q, kc, vc, H, H // KVH, hs, 0, pos)
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    assert np.abs(got - want).max() <= 2e-5


def test_row_parallel_layer_without_communicator_fails():
    assert host.hlib().kxm_test_matmul_tp_without_comm() == 1
    shape = dict(dim=512, hidden_dim=1024, layer_num=1, head_num=8, kv_head_num=2, vocab_size=256,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    m = host.Model(seq_len=16, group_size=GS, tp_size=2, tp_rank=0, **shape)
    m.init_synthetic(5)
    for step in (m.forward_layers, m.step_logits):   # neither init_comm nor a loopback group
        with pytest.raises(kx.KxError):
            step(1, 0)


# ---------------- synthetic models vs the oracle ---------------------------------------------------
def _oracle_from(m, cfg_kwargs, seq_len):
    c = kxo.Config(seq_len=seq_len, group_size=GS, **cfg_kwargs)
    L = c.layer_num
    arrays = {}
    for name in ("wq", "wk", "wv", "wo", "w1", "w2", "w3"):
        arrays[name + "_q"] = np.stack([m.read_weight(name, l) for l in range(L)])
        arrays[name + "_s"] = np.stack([m.read_weight(name + "_s", l) for l in range(L)])
    arrays["cls_q"] = m.read_weight("cls")
    arrays["cls_s"] = m.read_weight("cls_s")
    arrays["emb"] = m.read_weight("emb")
    arrays["attn_norm"] = np.stack([m.read_weight("attn_norm", l) for l in range(L)])
    arrays["ffn_norm"] = np.stack([m.read_weight("ffn_norm", l) for l in range(L)])
    arrays["final_norm"] = m.read_weight("final_norm")
    return kxo.Model(c, arrays=arrays)


@pytest.mark.parametrize("shape", [
    dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
         rope_variant=1, rope_theta=5e5, rms_eps=1e-5),          # Llama-3 style, head 64, GQA 4
    dict(dim=512, hidden_dim=1536, layer_num=2, head_num=4, kv_head_num=4, vocab_size=640,
         rope_variant=1, rope_theta=1e6, rms_eps=1e-6),          # Qwen2 style, head 128
    dict(dim=768, hidden_dim=2048, layer_num=1, head_num=12, kv_head_num=12, vocab_size=512,
         rope_variant=0, rope_theta=1e4, rms_eps=1e-5),          # stories110M layer shape
])
def test_synthetic_model_fused_and_layer_paths_vs_oracle(shape):
    seq_len = 40
    m = host.Model(seq_len=seq_len, group_size=GS, **shape)
    m.init_synthetic(7)
    oracle = _oracle_from(m, shape, seq_len)
    rng = np.random.default_rng(0)
    toks = rng.integers(0, shape["vocab_size"], 6)
    m2 = host.Model(seq_len=seq_len, group_size=GS, **shape)
    m2.init_synthetic(7)
    for pos, t in enumerate(toks):
        want = oracle.forward(int(t), pos)
        tol = 3e-5 * max(1.0, np.abs(want).max())
        assert np.abs(m.step_logits(int(t), pos) - want).max() <= tol
        assert np.abs(m2.forward_layers(int(t), pos) - want).max() <= tol
    # greedy continuation: token-exact wherever the oracle's top-1 margin exceeds the tolerance
    m.set_state(int(toks[-1]), len(toks) - 1)
    m.decode(20, use_graph=True)
    hist = m.history(len(toks) + 20)
    pos = len(toks) - 1
    tok = int(toks[-1])
    for i in range(20):
        lg = oracle.forward(tok, pos)
        order = np.argsort(lg)
        margin = lg[order[-1]] - lg[order[-2]]
        got = int(hist[pos + 1])
        if margin > 1e-3 * max(1.0, np.abs(lg).max()):
            assert got == int(order[-1]), (i, margin)
        else:  # near tie: either of the tied candidates is a correct greedy pick
            assert lg[got] >= lg[order[-1]] - 1e-3 * max(1.0, np.abs(lg).max())
        tok, pos = got, pos + 1


@pytest.mark.parametrize("shape,n", [
    (dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
          rope_variant=1, rope_theta=5e5, rms_eps=1e-5), 37),
    (dict(dim=512, hidden_dim=1536, layer_num=2, head_num=4, kv_head_num=4, vocab_size=640,
          rope_variant=1, rope_theta=1e6, rms_eps=1e-6), 150),
    (dict(dim=768, hidden_dim=2048, layer_num=1, head_num=12, kv_head_num=12, vocab_size=512,
          rope_variant=0, rope_theta=1e4, rms_eps=1e-5), 64),
    # 512 rows and more per call: activations stay in fp16 between the kernels of a layer
    (dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
          rope_variant=1, rope_theta=5e5, rms_eps=1e-5), 530),
    (dict(dim=512, hidden_dim=1536, layer_num=1, head_num=4, kv_head_num=4, vocab_size=640,
          rope_variant=0, rope_theta=1e4, rms_eps=1e-6), 257),
])
def test_batched_prefill_vs_oracle_rows(shape, n):
    # The batched prompt pass must leave the KV rows and last-row logits that n single-token steps
    # leave (fp16 operand rounding: 4e-3 of each tensor's range), also when split into two calls.
    seq_len = max(192, n + 12)
    rng = np.random.default_rng(3)
    toks = rng.integers(0, shape["vocab_size"], n).astype(np.int32)
    m = host.Model(seq_len=seq_len, group_size=GS, **shape)
    m.init_synthetic(11)
    oracle = _oracle_from(m, shape, seq_len)
    for pos, t in enumerate(toks):
        want = oracle.forward(int(t), pos)
    ok_, ov_ = oracle.caches()
    split = n // 3
    m2 = host.Model(seq_len=seq_len, group_size=GS, **shape)
    m2.init_synthetic(11)
    m2.prefill(toks[:split])
    for mm, last in ((m, m.prefill(toks)), (m2, m2.prefill(toks[split:], pos0=split))):
        assert np.abs(last - want).max() <= 4e-3 * np.abs(want).max()
        for layer in range(shape["layer_num"]):
            k, v = mm.read_kv(layer, n)
            assert np.abs(k - ok_[layer, :n]).max() <= 4e-3 * np.abs(ok_[layer, :n]).max()
            assert np.abs(v - ov_[layer, :n]).max() <= 4e-3 * np.abs(ov_[layer, :n]).max()
        tok, pos = mm.state()
        assert pos == n
        order = np.argsort(want)
        if want[order[-1]] - want[order[-2]] > 1e-2 * np.abs(want).max():
            assert tok == int(order[-1])
        assert mm.history(n + 1)[n] == tok


def test_kept_fp16_weight_images_give_the_same_prompt_results():
    # prefill_weight_cache: the fp16 images of the int8 weights are made per pass (0) or kept after
    # the first pass (1); same kernels on the same images, so the results are bit-identical, also on
    # the second prompt (which skips the dequantisation) and after the weights are replaced
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(8)
    p1 = rng.integers(0, 1000, 600).astype(np.int32)
    p2 = rng.integers(0, 1000, 520).astype(np.int32)
    outs = {}
    for cache in (0, 1):
        m = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=cache, **shape)
        m.init_synthetic(31)
        a = m.prefill(p1)
        b = m.prefill(p2)           # images ready: no dequantisation pass
        ka, va = m.read_kv(1, 520)
        m.init_synthetic(32)        # new weights: kept images must not be reused
        c = m.prefill(p2)
        outs[cache] = (a, b, ka, va, c)
        m.close()
    for x, y in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(x, y)
    assert np.abs(outs[1][1] - outs[1][4]).max() > 0


def test_weight_image_policy_changed_on_a_live_model():
    # set_prefill_weight_cache on a model that already holds images: policy 0 gives them back
    # (weight_image_bytes() == 0) and the next prompt de-quantises per pass; back to "kept" the
    # images are made again; every pass gives the bit-identical result (same kernels, same images)
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(9)
    p = rng.integers(0, 1000, 600).astype(np.int32)
    m = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=1, **shape)
    m.init_synthetic(31)
    assert m.weight_image_bytes() == 0          # nothing before the first long prompt
    kept = m.prefill(p)
    per_layer = 2 * (512 * (512 + 2 * 128) + 512 * 512 + 2 * 1024 * 512 + 512 * 1024)
    assert m.weight_image_bytes() >= 2 * per_layer
    m.set_prefill_weight_cache(0)
    assert m.weight_image_bytes() == 0
    np.testing.assert_array_equal(m.prefill(p), kept)
    assert m.weight_image_bytes() == 0
    m.set_prefill_weight_cache(-1)              # kept if they fit: they do
    np.testing.assert_array_equal(m.prefill(p), kept)
    assert m.weight_image_bytes() >= 2 * per_layer
    np.testing.assert_array_equal(m.prefill(p), kept)   # second pass on the kept images
    m.close()


def test_prompts_up_to_192_rows_stream_the_int8_weights(monkeypatch):
    # up to 192 rows the prompt pass runs the few-rows kernel on the int8 weights (no fp16 image, with
    # or without prefill_weight_cache, whatever came before): within the batched-prompt tolerance of
    # one fp32 decode step per row, K/V rows included, same bits when repeated and after a long
    # prompt; the switch back to the tiled kernels stays within the same tolerance
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(4)
    m = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=1, **shape)
    m.init_synthetic(31)
    plain = host.Model(seq_len=640, group_size=GS, **shape)
    plain.init_synthetic(31)
    ref = host.Model(seq_len=640, group_size=GS, **shape)
    ref.init_synthetic(31)
    prompts = {n: rng.integers(0, 1000, n).astype(np.int32) for n in (2, 7, 16, 17, 40, 64, 65, 100, 128, 129, 192)}
    first = {n: plain.prefill(p) for n, p in prompts.items()}
    m.prefill(rng.integers(0, 1000, 520).astype(np.int32))      # a long prompt first: images exist now
    for n, p in prompts.items():
        got = m.prefill(p)
        np.testing.assert_array_equal(got, first[n])             # same form either way: same bits
        np.testing.assert_array_equal(got, plain.prefill(p))
        want = ref.prefill(p, exact=True)
        assert np.abs(got - want).max() <= 4e-3 * np.abs(want).max(), n
        assert int(np.argmax(got)) == int(np.argmax(want)) or np.sort(want)[-1] - np.sort(want)[-2] < 1e-2
        for layer in range(2):
            k, v = m.read_kv(layer, n)
            kr, vr = ref.read_kv(layer, n)
            assert np.abs(k - kr).max() <= 4e-3 * np.abs(kr).max()
            assert np.abs(v - vr).max() <= 4e-3 * np.abs(vr).max()
        assert m.state()[1] == n
        m.decode(3, use_graph=False)
        ref.decode(3, use_graph=False)
    # SwiGLU and bias + RoPE as launches of their own instead of the products' epilogues: the same
    # formulas on the same totals
    monkeypatch.setenv("KX_PREFILL_ROWS_UNFUSED", "1")
    for n in (7, 40, 100, 192):
        got = plain.prefill(prompts[n])
        assert np.abs(got - first[n]).max() <= 1e-5 * np.abs(first[n]).max(), n
    monkeypatch.delenv("KX_PREFILL_ROWS_UNFUSED")
    monkeypatch.setenv("KX_PREFILL_NO_ROWS_KERNEL", "1")
    for n in (16, 40):
        got = plain.prefill(prompts[n])
        assert np.abs(got - first[n]).max() <= 4e-3 * np.abs(first[n]).max()
    for x in (m, plain, ref):
        x.close()


def test_prompts_of_193_to_304_rows_without_weight_images_run_as_pieces(monkeypatch):
    # no fp16 images (policy 0): a prompt of 193..304 rows is cut into two pieces of at most
    # 192 rows, each through the few-rows kernel on the int8 weights, later pieces attending the
    # earlier ones' cache rows.  Within the batched-prompt tolerance of one fp32 decode step per row,
    # K/V rows included; the single-pass int8 kernel (pieces switched off) stays within it too; with
    # kept images the same prompts take the fp16 GEMMs and the policy setter moves between the two
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(14)
    m = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=0, **shape)
    m.init_synthetic(31)
    ref = host.Model(seq_len=640, group_size=GS, **shape)
    ref.init_synthetic(31)
    for n in (193, 256, 300, 304, 305, 384, 448):  # 305+: weights converted per pass, fp16 GEMMs
        p = rng.integers(0, 1000, n).astype(np.int32)
        got = m.prefill(p)
        assert m.weight_image_bytes() == 0
        want = ref.prefill(p, exact=True)
        assert np.abs(got - want).max() <= 4e-3 * np.abs(want).max(), n
        for layer in range(2):
            k, v = m.read_kv(layer, n)
            kr, vr = ref.read_kv(layer, n)
            assert np.abs(k - kr).max() <= 4e-3 * np.abs(kr).max(), (n, layer)
            assert np.abs(v - vr).max() <= 4e-3 * np.abs(vr).max(), (n, layer)
        assert m.state()[1] == n
        np.testing.assert_array_equal(m.prefill(p), got)           # repeatable
        monkeypatch.setenv("KX_PREFILL_NO_PIECES", "1")
        whole = m.prefill(p)                                        # the int8 single-pass kernel on all rows
        monkeypatch.delenv("KX_PREFILL_NO_PIECES")
        assert np.abs(whole - want).max() <= 4e-3 * np.abs(want).max(), n
    for x in (m, ref):
        x.close()


def test_short_prompts_with_qkv_biases_and_interleaved_rope():
    # the few-rows prompt pass with q / k / v biases (Qwen2's layout) ahead of the rotation, for both
    # RoPE variants: against one fp32 decode step per row, K/V rows included
    rng = np.random.default_rng(21)
    for variant in (0, 1):
        shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                     rope_variant=variant, rope_theta=1e6, rms_eps=1e-6)
        m = host.Model(seq_len=256, group_size=GS, qkv_bias=True, **shape)
        m.init_synthetic(77)
        ref = host.Model(seq_len=256, group_size=GS, qkv_bias=True, **shape)
        ref.init_synthetic(77)
        for layer in range(2):
            bq = (rng.standard_normal(512) * 0.3).astype(np.float32)
            bk = (rng.standard_normal(128) * 0.3).astype(np.float32)
            bv = (rng.standard_normal(128) * 0.3).astype(np.float32)
            m.set_qkv_bias(layer, bq, bk, bv)
            ref.set_qkv_bias(layer, bq, bk, bv)
        for n in (9, 70, 150):
            p = rng.integers(0, 1000, n).astype(np.int32)
            got = m.prefill(p)
            want = ref.prefill(p, exact=True)
            assert np.abs(got - want).max() <= 4e-3 * np.abs(want).max(), (variant, n)
            for layer in range(2):
                k, v = m.read_kv(layer, n)
                kr, vr = ref.read_kv(layer, n)
                assert np.abs(k - kr).max() <= 4e-3 * np.abs(kr).max()
                assert np.abs(v - vr).max() <= 4e-3 * np.abs(vr).max()
        m.close()
        ref.close()


def test_short_prompts_use_the_kept_weight_images():
    # a model that keeps fp16 images of its weights runs prompts of 193..511 rows on the fp16 GEMMs
    # (with cut reductions: few tiles, many idle CUs) instead of the int8 single-pass kernel, whether
    # or not a long prompt came first; same tolerance against one fp32 decode step per row as the
    # long-prompt form
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(9)
    m = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=1, **shape)
    m.init_synthetic(31)
    ref = host.Model(seq_len=640, group_size=GS, **shape)
    ref.init_synthetic(31)
    fresh = host.Model(seq_len=640, group_size=GS, prefill_weight_cache=1, **shape)
    fresh.init_synthetic(31)
    p0 = rng.integers(0, 1000, 250).astype(np.int32)
    first = fresh.prefill(p0)                                    # makes the images on the way
    m.prefill(rng.integers(0, 1000, 520).astype(np.int32))      # images made by a long prompt
    np.testing.assert_array_equal(first, m.prefill(p0))         # same form either way: same bits
    fresh.close()
    for n in (193, 250, 300):
        p = rng.integers(0, 1000, n).astype(np.int32)
        got = m.prefill(p)
        want = ref.prefill(p, exact=True)
        # fp16 weight images and fp16 rows: 2^-11 per factor, through two layers and the classifier
        assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max(), n
        assert int(np.argmax(got)) == int(np.argmax(want)) or np.sort(want)[-1] - np.sort(want)[-2] < 1e-2
        for layer in range(2):
            k, v = m.read_kv(layer, n)
            kr, vr = ref.read_kv(layer, n)
            assert np.abs(k - kr).max() <= 4e-3 * np.abs(kr).max()
            assert np.abs(v - vr).max() <= 4e-3 * np.abs(vr).max()
        assert m.state()[1] == n
        m.decode(3, use_graph=False)
        ref.decode(3, use_graph=False)
    m.close()
    ref.close()


def test_tp_shards_are_slices_of_the_unsharded_weights():
    # column-parallel rows / row-parallel columns as the shard plan says, for synthetic weights
    shape = dict(dim=512, hidden_dim=1024, layer_num=1, head_num=8, kv_head_num=2, vocab_size=256,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    full = host.Model(seq_len=16, group_size=GS, **shape)
    full.init_synthetic(5)
    for r in range(2):
        m = host.Model(seq_len=16, group_size=GS, tp_size=2, tp_rank=r, **shape)
        m.init_synthetic(5)
        plan, _ = host.shard_plan(512, 1024, 8, 2, 2, r)
        for name, (r0, rows, c0, cols) in plan.items():
            np.testing.assert_array_equal(m.read_weight(name),
                                          full.read_weight(name)[r0:r0 + rows, c0:c0 + cols])
            sf = full.read_weight(name + "_s")
            sf = sf.reshape(full.read_weight(name).shape[0], -1)
            np.testing.assert_array_equal(m.read_weight(name + "_s").reshape(rows, -1),
                                          sf[r0:r0 + rows, c0 // GS:(c0 + cols) // GS])


def test_single_rank_communicator_runs_the_allreduce_path():
    # A one-rank RCCL communicator: the all-reduces after Wo and W2 (side stream, two events) run
    # as identities, eagerly, under graph capture/replay and in the batched prompt pass.
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    toks = np.random.default_rng(1).integers(0, 1000, 24).astype(np.int32)
    runs = []
    for with_comm in (False, True):
        m = host.Model(seq_len=64, group_size=GS, **shape)
        m.init_synthetic(9)
        if with_comm:
            m.init_comm(host.comm_unique_id())
        last = m.prefill(toks)
        m.decode(12, use_graph=True)
        runs.append((last, m.history(len(toks) + 13)))
        if with_comm:
            assert m.profile_allreduce(512, 5) > 0.0    # ms per call on the side-stream path
        else:
            with pytest.raises(kx.KxError):
                m.profile_allreduce(512, 5)             # no communicator
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])


def test_prompt_pass_overlaps_reduces_with_the_next_sub_block():
    # Under tensor parallelism a long prompt pass sends Wo and W2 through the GEMM in row
    # sub-blocks; the reduce of one sub-block is queued on the side stream while the next
    # sub-block's GEMM runs.  With a one-rank RCCL communicator (the collectives are identities)
    # the mechanism runs for real -- two streams, two events, several collectives in flight per
    # GEMM -- and must give exactly what the serial form and the communicator-free model give.
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    toks = np.random.default_rng(4).integers(0, 1000, 2300).astype(np.int32)
    outs = []
    for comm, nsub in ((False, 4), (True, 1), (True, 4), (True, 3)):
        m = host.Model(seq_len=2400, group_size=GS, **shape)
        m.init_synthetic(9)
        if comm:
            m.init_comm(host.comm_unique_id())
        m.set_prefill_subblocks(nsub)
        last = m.prefill(toks)
        k, v = m.read_kv(1, len(toks))
        m.decode(4, use_graph=True)
        outs.append((last, k, v, m.history(len(toks) + 5)))
        m.close()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            np.testing.assert_array_equal(a, b)


def test_tensor_parallel_prompt_sub_blocks_equal_single_rank():
    # the same with real partial sums: two ranks in this process (host rendezvous), 2300 rows in
    # sub-blocks of 768/1024 rows, against the unsharded model
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=4, vocab_size=700,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    toks = np.random.default_rng(6).integers(0, 700, 2300).astype(np.int32)
    one = host.Model(seq_len=2400, group_size=GS, **shape)
    one.init_synthetic(21)
    want = one.prefill(toks)
    for nsub in (1, 3):
        group = host.loopback_create(2)
        ranks = [host.Model(seq_len=2400, group_size=GS, tp_size=2, tp_rank=r, **shape) for r in range(2)]
        for m in ranks:
            m.init_synthetic(21)
            m.join_loopback(group)
            m.set_prefill_subblocks(nsub)
        got = _run_ranks(ranks, lambda r, m: m.prefill(toks))
        np.testing.assert_array_equal(got[0], got[1])
        assert np.abs(got[0] - want).max() <= 2e-3 * np.abs(want).max()
        for m in ranks:
            m.close()
        host.loopback_destroy(group)
    one.close()


def test_tensor_parallel_prompts_under_both_weight_image_policies():
    # the sequence bench.py runs on every rank at N > 1: policy 0 (no fp16 image kept), prompt,
    # policy -1, prompt, prompt -- on two ranks of this process; every pass agrees across the ranks
    # and with the unsharded model, for a long prompt (per-pass conversion / kept images) and for one
    # of 300 rows (two few-rows pieces without images, fp16 GEMMs with them)
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=4, vocab_size=700,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    rng = np.random.default_rng(16)
    one = host.Model(seq_len=1200, group_size=GS, **shape)
    one.init_synthetic(21)
    group = host.loopback_create(2)
    ranks = [host.Model(seq_len=1200, group_size=GS, tp_size=2, tp_rank=r, **shape) for r in range(2)]
    for m in ranks:
        m.init_synthetic(21)
        m.join_loopback(group)
    for n in (1100, 300):
        toks = rng.integers(0, 700, n).astype(np.int32)
        want = one.prefill(toks)

        def passes(r, m):
            out = []
            m.set_prefill_weight_cache(0)
            out.append(m.prefill(toks))
            assert m.weight_image_bytes() == 0
            m.set_prefill_weight_cache(-1)
            out.append(m.prefill(toks))
            out.append(m.prefill(toks))
            assert m.weight_image_bytes() > 0
            return out
        got = _run_ranks(ranks, passes)
        for a_, b_ in zip(got[0], got[1]):
            np.testing.assert_array_equal(a_, b_)
            assert np.abs(a_ - want).max() <= 4e-3 * np.abs(want).max(), n
    for m in ranks:
        m.close()
    host.loopback_destroy(group)
    one.close()


# ---------------- 16-bit KV cache -------------------------------------------------------------------
@pytest.mark.parametrize("n", [40, 530])
def test_kv16_model_tracks_the_fp32_cache_model(n):
    # Same weights, fp32 cache vs IEEE-half cache: layer 0's rows are the fp32 rows rounded once
    # (the same values go in), deeper layers and the logits stay within the rounding's reach, and
    # the batched prompt pass, eager steps and graph replay agree with each other.
    shape = dict(dim=512, hidden_dim=1024, layer_num=3, head_num=8, kv_head_num=2, vocab_size=1000,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    seq_len = n + 40
    toks = np.random.default_rng(n).integers(0, 1000, n).astype(np.int32)
    ref = host.Model(seq_len=seq_len, group_size=GS, **shape)
    ref.init_synthetic(4)
    m = host.Model(seq_len=seq_len, group_size=GS, kv_f16=True, **shape)
    m.init_synthetic(4)
    want_last = ref.prefill(toks)
    got_last = m.prefill(toks)
    assert np.abs(got_last - want_last).max() <= 1e-2 * np.abs(want_last).max()
    for layer in range(3):
        kf, vf = ref.read_kv(layer, n)
        k, v = m.read_kv(layer, n)
        if layer == 0:
            np.testing.assert_array_equal(k, kf.astype(np.float16).astype(np.float32))
            np.testing.assert_array_equal(v, vf.astype(np.float16).astype(np.float32))
        else:
            assert np.abs(k - kf).max() <= 1e-2 * np.abs(kf).max()
            assert np.abs(v - vf).max() <= 1e-2 * np.abs(vf).max()
    # the sequential prompt pass (fused decode steps) files the same kind of rows
    m2 = host.Model(seq_len=seq_len, group_size=GS, kv_f16=True, **shape)
    m2.init_synthetic(4)
    seq_last = m2.prefill(toks, exact=True)
    assert np.abs(seq_last - want_last).max() <= 1e-2 * np.abs(want_last).max()
    k0, v0 = m2.read_kv(0, n)
    kf, vf = ref.read_kv(0, n)
    # layer 0 again, now from the GEMV epilogue: fp32 GEMV vs the MFMA prompt GEMM differ by the
    # GEMM's own operand rounding, so compare with that tolerance
    assert np.abs(k0 - kf).max() <= 4e-3 * np.abs(kf).max()
    # eager and graph-replayed decode from the same state give the same tokens
    m.decode(8, use_graph=False)
    eager = m.history(n + 9)
    m3 = host.Model(seq_len=seq_len, group_size=GS, kv_f16=True, **shape)
    m3.init_synthetic(4)
    m3.prefill(toks)
    m3.decode(8, use_graph=True)
    np.testing.assert_array_equal(m3.history(n + 9), eager)
    with pytest.raises(kx.KxError):
        m.forward_layers(1, 0)   # the layer-by-layer loop reads fp32 cache views
    for mm in (ref, m, m2, m3):
        mm.close()


def test_kv16_under_tensor_parallel_shards():
    # every rank rounds its own kv heads' rows: the sharded half cache is the column slice of the
    # unsharded one, and the logits differ only by the re-association of the row-parallel sums
    shape = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=4, vocab_size=700,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
    seq_len, tp = 576, 2
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, 700, 520).astype(np.int32)
    one = host.Model(seq_len=seq_len, group_size=GS, kv_f16=True, **shape)
    one.init_synthetic(21)
    ref_last = one.prefill(prompt)
    one.decode(4, use_graph=False)
    group = host.loopback_create(tp)
    ranks = [host.Model(seq_len=seq_len, group_size=GS, tp_size=tp, tp_rank=r, kv_f16=True, **shape)
             for r in range(tp)]
    for m in ranks:
        m.init_synthetic(21)
        m.join_loopback(group)
    outs = _run_ranks(ranks, lambda r, m: (m.prefill(prompt), m.decode(4, use_graph=False),
                                           m.history(len(prompt) + 5)))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    assert np.abs(outs[0][0] - ref_last).max() <= 2e-3 * np.abs(ref_last).max()
    kvl = ranks[0].kv_dim_local
    kf, vf = one.read_kv(0, len(prompt))
    for r, m in enumerate(ranks):
        k, v = m.read_kv(0, len(prompt))
        np.testing.assert_array_equal(k, kf[:, r * kvl:(r + 1) * kvl])   # layer 0: same inputs, same rounding
        np.testing.assert_array_equal(v, vf[:, r * kvl:(r + 1) * kvl])
    for m in ranks:
        m.close()
    one.close()
    host.loopback_destroy(group)


def test_fused_path_refuses_fp32_weights():
    m = host.Model(seq_len=8, group_size=0, **TINY)
    m.init_synthetic(1)
    with pytest.raises(kx.KxError):
        m.step_logits(1, 0)


# ---------------- tensor-parallel token loop on one GPU ---------------------------------------------
def _run_ranks(models, fn):
    """Runs fn(rank, model) on one host thread per rank (ctypes releases the GIL) and returns the
    results in rank order; the ranks meet inside the loopback all-reduce."""
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
        t.join(120)
    assert not err, err
    assert all(o is not None for o in out), "a rank did not finish"
    return out


TP_SMALL = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=4, vocab_size=700,
                rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
# one Llama-3-8B layer: at tp = 8 every rank holds exactly the per-GPU shapes of the 8-GPU
# benchmark run (QKV 768 x 4096, Wo 4096 x 512, W1|W3 3584 x 4096, W2 4096 x 1792, one kv head)
TP_8B_LAYER = dict(dim=4096, hidden_dim=14336, layer_num=1, head_num=32, kv_head_num=8,
                   vocab_size=700, rope_variant=1, rope_theta=5e5, rms_eps=1e-5)


@pytest.mark.parametrize("tp,shape", [(2, TP_SMALL), (4, TP_SMALL), (8, TP_8B_LAYER)],
                         ids=["tp2", "tp4", "tp8-8b-layer"])
def test_tensor_parallel_ranks_equal_single_rank(tp, shape):
    # Every rank of a TP model as an object of this process, the all-reduces through a host
    # rendezvous: column/row shards, the residual owner and the two reduces per layer must
    # reproduce the unsharded model (row-parallel sums are re-associated: 1e-4 of the logit range).
    seq_len = 576
    rng = np.random.default_rng(5)
    toks = rng.integers(0, 700, 10).astype(np.int32)
    one = host.Model(seq_len=seq_len, group_size=GS, **shape)
    one.init_synthetic(21)
    want = np.stack([one.step_logits(int(t), p) for p, t in enumerate(toks)])
    group = host.loopback_create(tp)
    ranks = [host.Model(seq_len=seq_len, group_size=GS, tp_size=tp, tp_rank=r, **shape)
             for r in range(tp)]
    for m in ranks:
        m.init_synthetic(21)
        m.join_loopback(group)
    got = _run_ranks(ranks, lambda r, m: np.stack([m.step_logits(int(t), p)
                                                   for p, t in enumerate(toks)]))
    for r in range(tp):
        np.testing.assert_array_equal(got[r], got[0])   # every rank ends with the same logits
    assert np.abs(got[0] - want).max() <= 1e-4 * np.abs(want).max()
    # the classifier's rows are split over the ranks when the vocabulary divides by tp
    assert ranks[0].read_weight("cls").shape[0] == (700 // tp if 700 % tp == 0 else 700)
    # the reference-structured loop (op::MatmulLayer with its reducer, registry kernels) under the
    # same sharding: row-parallel layers reduce through the model, the classifier slices are gathered
    ref_layers = np.stack([one.forward_layers(int(t), p) for p, t in enumerate(toks[:3])])
    got_layers = _run_ranks(ranks, lambda r, m: np.stack([m.forward_layers(int(t), p)
                                                          for p, t in enumerate(toks[:3])]))
    for r in range(tp):
        assert np.abs(got_layers[r] - ref_layers).max() <= 1e-4 * np.abs(ref_layers).max()
    # each rank's cache rows are its kv heads' columns of the unsharded cache
    kvl = ranks[0].kv_dim_local
    for layer in range(shape["layer_num"]):
        kf, vf = one.read_kv(layer, len(toks))
        for r, m in enumerate(ranks):
            k, v = m.read_kv(layer, len(toks))
            np.testing.assert_allclose(k, kf[:, r * kvl:(r + 1) * kvl], rtol=0, atol=1e-4 * np.abs(kf).max())
            np.testing.assert_allclose(v, vf[:, r * kvl:(r + 1) * kvl], rtol=0, atol=1e-4 * np.abs(vf).max())
    # batched prompt pass under TP (GEMMs, flash attention on the local heads, reduces of whole
    # row blocks), long enough for the two-pass GEMM, then eager greedy steps
    prompt = rng.integers(0, 700, 530).astype(np.int32)
    ref_last = one.prefill(prompt)
    one.decode(6, use_graph=False)
    ref_hist = one.history(len(prompt) + 7)
    outs = _run_ranks(ranks, lambda r, m: (m.prefill(prompt), m.decode(6, use_graph=False),
                                           m.history(len(prompt) + 7)))
    for r in range(tp):
        assert np.abs(outs[r][0] - ref_last).max() <= 2e-3 * np.abs(ref_last).max()
        np.testing.assert_array_equal(outs[r][2], outs[0][2])
    order = np.argsort(ref_last)
    if ref_last[order[-1]] - ref_last[order[-2]] > 1e-2 * np.abs(ref_last).max():
        assert outs[0][2][len(prompt)] == ref_hist[len(prompt)]
    # short prompts under TP: the few-rows products on each rank's shard (where its widths allow)
    for n in (9, 100):
        short = rng.integers(0, 700, n).astype(np.int32)
        ref_short = one.prefill(short)
        outs = _run_ranks(ranks, lambda r, m: m.prefill(short))
        for r in range(tp):
            assert np.abs(outs[r] - ref_short).max() <= 2e-3 * np.abs(ref_short).max(), (n, r)
            np.testing.assert_array_equal(outs[r], outs[0])
    with pytest.raises(kx.KxError):
        ranks[0].decode(1, use_graph=True)   # a host rendezvous cannot be captured
    for m in ranks:
        m.close()
    host.loopback_destroy(group)


# ---------------- the text front-end (demo/main.cpp:5-75) on the device -------------------------------
def test_generate_front_end_matches_one_token_per_step(tmp_path):
    # checkpoint file + tokenizer file -> words, as the demo: kx.generate opens the golden int8 export
    # by its header, a SentencePiece model of the same vocabulary size (trained here; the reference
    # ships none) tokenises, and the chunked graph-replayed loop must give the words of a host loop
    # that asks for one token per step and tests each for the sentence end (demo/main.cpp:18-41).
    import io
    import os
    import sentencepiece as spm
    from kx import generate as gen
    corpus = ["hello world this is a tiny corpus", "the quick brown fox jumps over the lazy dog",
              "once upon a time there was a little model", "hello hello world"] * 20
    w = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(corpus), model_writer=w, vocab_size=128,
                                   model_type="bpe", bos_id=1, eos_id=2, unk_id=0, pad_id=-1, minloglevel=2)
    tok_path = tmp_path / "tokenizer.model"
    tok_path.write_bytes(w.getvalue())
    ckpt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4_tiny_q8.bin")
    tok = gen.open_tokenizer(str(tok_path), "llama2")
    assert tok.vocab_size() == 128

    class EndsAt:  # a tokenizer whose sentence end is a token the stream is known to reach
        def __init__(self, inner, end):
            self.inner, self.end = inner, end
        def encode(self, s): return self.inner.encode(s)
        def decode(self, ids): return self.inner.decode(ids)
        def is_sentence_ending(self, t): return int(t) == self.end

    sentence = "once upon a time there was"
    tokens = tok.encode(sentence)
    n = len(tokens)
    # the one-token-per-step loop: same prompt pass, then single eager steps read back one by one
    ref = gen.open_model(ckpt, "llama2", seq_len=32)
    ref.prefill(np.asarray(tokens, np.int32))
    stream = [ref.state()[0]]
    for _ in range(32 - n):
        ref.decode(1, use_graph=False)
        stream.append(ref.state()[0])
    ref.close()
    m = gen.open_model(ckpt, "llama2", seq_len=32)
    for total, chunk in ((n - 1, 16), (n + 5, 4), (1000, 16), (1000, 5)):
        steps, words = gen.generate(m, tok, sentence, total, chunk=chunk)
        sampled = min(max(total - (n - 1), 0), 32 - n + 1)
        want = tokens[1:] + [t for t in stream[:sampled]]
        if tok.is_sentence_ending(2) and 2 in stream[:sampled]:
            want = tokens[1:] + stream[:stream.index(2)]
        assert words == want and steps == len(want), (total, chunk)
    # a sentence end in the middle of a chunk: the words stop in front of it
    end = stream[6]
    first = stream.index(end)
    steps, words = gen.generate(m, EndsAt(tok, end), sentence, 1000, chunk=4)
    assert words == tokens[1:] + stream[:first] and steps == n - 1 + first
    out = io.StringIO()
    gen.generate(m, tok, sentence, n + 3, need_output=True, out=out)
    assert out.getvalue().startswith(sentence[:4])
    m.close()
    # and the command line (demo/main.cpp:50-75): checkpoint path, tokenizer path -> text and steps/s
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = gen.main([ckpt, str(tok_path), "--family", "llama2", "--prompt", sentence, "--steps", "12",
                       "--seq-len", "32"])
    assert rc == 0 and buf.getvalue().startswith("Generating...") and "steps/s:" in buf.getvalue()


def test_generate_front_end_reproduces_the_golden_streams(golden):
    # the fixtures' prompt ids in, the reference implementation's greedy streams out: the fp32 export
    # one reference-structured step per position (the demo's default model kind), the int8 export
    # through prefill + chunked graph-replayed decode
    import os
    from kx import generate as gen

    class Fixed:  # stands in for the tokenizer: the sentence IS the fixture's prompt
        def __init__(self, ids):
            self.ids = ids
        def encode(self, s): return list(self.ids)
        def decode(self, ids): return " ".join(str(i) for i in ids)
        def is_sentence_ending(self, t): return False

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, quant in (("g3_tiny_fp32", False), ("g4_tiny_q8", True)):
        g = golden(name + ".npz")
        toks, greedy = g["tokens"].tolist(), g["greedy"].tolist()
        m = gen.open_model(os.path.join(here, name + ".bin"), "llama2", quant=quant, seq_len=32)
        total = len(toks) - 1 + len(greedy)
        steps, words = gen.generate(m, Fixed(toks), "the fixture's prompt", total, chunk=5)
        assert words == toks[1:] + greedy and steps == total, name
        m.close()
