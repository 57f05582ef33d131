"""Cases at the END of an allocation, and fixed-seed slices of the randomised sweeps.

A read a few bytes past an array is invisible to a parity check (nothing uses the value) until the
array ends on the last mapped page: then the device faults.  Two such reads were found by the
sweeps in tools/ (the few-rows product's scale fetch on the matrix's last row when M % 512 == 256;
the fused Wq|Wk|Wv GEMV indexing the half-split sin/cos row with a v row's index at the last cache
position).  These tests pin both shapes with buffers allocated at their exact size, and run a
slice of each sweep so that every GPU test run draws shapes nobody wrote down by hand."""
import os
import sys

import numpy as np
import pytest

import kx
from kx import host
from oracle import kxo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
GS = 64


@pytest.mark.parametrize("hs,H,KVH", [(64, 8, 2), (128, 4, 2), (32, 8, 4)])
@pytest.mark.parametrize("variant", [1, 0])
def test_fused_decode_step_at_the_last_cache_row(hs, H, KVH, variant):
    # 448-row cache; a prompt of seq - 3 rows, then three decode steps: the third one sits at
    # position seq - 1, where the rotary tables and the K/V blocks of the last layer end.  Eager
    # and graph replay; logits of every step against the reference-structured layer loop.
    seq = 448
    shape = dict(dim=H * hs, hidden_dim=832, layer_num=2, head_num=H, kv_head_num=KVH, vocab_size=1141,
                 rope_variant=variant, rope_theta=1e4, rms_eps=1e-6)
    rng = np.random.default_rng(1)
    prompt = rng.integers(0, shape["vocab_size"], seq - 3).astype(np.int32)
    ref = host.Model(seq_len=seq, group_size=GS, **shape)
    ref.init_synthetic(99)
    streams = []
    for use_graph in (False, True):
        m = host.Model(seq_len=seq, group_size=GS, **shape)
        m.init_synthetic(99)
        m.prefill(prompt, exact=True)
        ref.prefill(prompt, exact=True)
        for step in range(3):
            tok, pos = m.state()
            assert pos == seq - 3 + step
            want = ref.forward_layers(tok, pos)
            got = m.step_logits(tok, pos)
            assert np.isfinite(got).all()
            assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), (use_graph, step)
            m.set_state(tok, pos)
            m.decode(1, use_graph=use_graph)
            m.sync()
        assert m.state()[1] == seq
        streams.append(m.history(seq + 1)[seq - 3:])
        m.close()
    np.testing.assert_array_equal(streams[0], streams[1])
    ref.close()


@pytest.mark.parametrize("K,M", [(2048, 256), (768, 768), (130, 1280), (128, 256)])
@pytest.mark.parametrize("T", [2, 40, 130])
def test_few_rows_product_whose_scale_array_ends_the_allocation(K, M, T):
    # M % 512 == 256: the last 128-element unit pair of a row has one half only.  Weights, scales
    # and activations are uploaded into allocations of exactly their size, matrix last row included.
    rng = np.random.default_rng(K + M + T)
    w = rng.integers(-127, 128, (K, M), dtype=np.int8)
    s = rng.uniform(1e-4, 2e-2, K * M // GS).astype(np.float32)
    X = rng.standard_normal((T, M)).astype(np.float16).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
    tol = 2.0 ** -11 * (np.abs(X.astype(np.float64)) @ sd.T) + 1e-7 * np.abs(want).max()
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    for half in (True, False):
        got = kx.gemm_w8_rows(X, w, s, GS, split_ws=sp, half=half)
        assert np.all(np.abs(got - want) <= tol), (half, float((np.abs(got - want) - tol).max()))
    assert not sp.numpy()[:4096].any()


def test_prompt_pass_with_a_head_size_the_fp16_attention_has_no_form_for():
    # head sizes other than 32 / 64 / 128 (here 96 and 48) have no fp16-output attention kernel: a
    # prompt of 4..192 rows must take the fp32 route instead of failing
    for hs, H, KVH in ((96, 8, 2), (48, 16, 4)):
        shape = dict(dim=H * hs, hidden_dim=1024, layer_num=2, head_num=H, kv_head_num=KVH, vocab_size=777,
                     rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
        rng = np.random.default_rng(hs)
        m = host.Model(seq_len=256, group_size=GS, **shape)
        ref = host.Model(seq_len=256, group_size=GS, **shape)
        m.init_synthetic(5)
        ref.init_synthetic(5)
        for n in (4, 16, 100, 192, 230):
            p = rng.integers(0, 777, n).astype(np.int32)
            got = m.prefill(p)
            # the fused token loop has no attention form for these head sizes either: the
            # reference-structured layer loop, one row at a time, is the comparison
            for pos, t in enumerate(p):
                want = ref.forward_layers(int(t), pos)
            assert np.isfinite(got).all()
            assert np.abs(got - want).max() <= 6e-3 * np.abs(want).max(), (hs, n)
        m.close()
        ref.close()


def test_model_created_while_the_null_stream_is_busy():
    # The model's buffers are zero-filled without a stream and first written by kernels on the
    # model's own NON-BLOCKING stream (rotary tables, synthetic weights).  A fill still pending on
    # the null stream when those kernels ran zeroed their output afterwards: 3 of 11 runs of this
    # suite against the bounds-checking build had one model whose rotary table began with zeros (K
    # row 0 all zero in the fused loop, logits off by 1-2 % from step 1 on).  kx_memset without a
    # stream now completes before it returns.  This case creates and fills a model while the null
    # stream is busy; it guards the ordering, though the pending fill of the original failure came
    # from the runtime's own batching and did not need a busy stream.
    shape = dict(dim=512, hidden_dim=832, layer_num=2, head_num=4, kv_head_num=2, vocab_size=1243,
                 rope_variant=1, rope_theta=5e5, rms_eps=1e-6)
    L = kx.lib()
    ref = host.Model(seq_len=256, group_size=GS, **shape)
    ref.init_synthetic(77)
    big = kx.Buf(1 << 30)
    for rep in range(200):    # ~0.3 ms each at 1 GiB: the null stream stays busy through the creation below
        kx.check(L.kx_fill_hash_f32(big.ptr, 1, 1 << 28, 1 << 28, 0, 0, rep, -1.0, 1.0, None), "fill")
    m = host.Model(seq_len=256, group_size=GS, **shape)
    for rep in range(200):
        kx.check(L.kx_fill_hash_f32(big.ptr, 1, 1 << 28, 1 << 28, 0, 0, rep, -1.0, 1.0, None), "fill")
    m.init_synthetic(77)
    rng = np.random.default_rng(3)
    for p, t in enumerate(rng.integers(0, 1243, 4)):
        a, b = m.step_logits(int(t), p), ref.forward_layers(int(t), p)
        assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max(), p
    prompt = rng.integers(0, 1243, 14).astype(np.int32)
    got, want = m.prefill(prompt), ref.prefill(prompt, exact=True)
    assert np.abs(got - want).max() <= 6e-3 * np.abs(want).max()
    m.close()
    ref.close()
    big.free()


def test_randomised_kernel_sweep_fixed_seed():
    import fuzz_parity
    assert fuzz_parity.run(40, 20261) == 0


def test_randomised_model_sweep_fixed_seed(capfd):
    import fuzz_model
    assert fuzz_model.run(60, 42) == 0     # the sweep that found the rotary-table read
    assert fuzz_model.run(20, 20262, head_sizes=(64, 128)) == 0


def test_bounds_build_sees_a_read_past_a_scale_array():
    # the self-test of the bounds-checking build: a scale array one group short makes the last row's
    # last load reach 4 bytes past its allocation -- invisible to parity, recorded by the checker
    # (and then cleared, so the run's final report stays about the library, not about this test)
    if not kx.bounds_enabled():
        pytest.skip("not the bounds-checking build (make BOUNDS=1, KX_LIB_DIR=.../lib_bounds)")
    before, _ = kx.bounds_report()
    K, M = 64, 1024
    rng = np.random.default_rng(0)
    dx = kx.Buf.from_numpy(rng.standard_normal(M).astype(np.float32))
    dw = kx.Buf.from_numpy(rng.integers(-127, 128, (K, M), dtype=np.int8))
    ds = kx.Buf.from_numpy(np.ones(K * M // GS - 1, np.float32))      # one scale short
    dy = kx.Buf.zeros(K)
    kx.check(kx.lib().kx_gemv_w8_f32(dx.ptr, dw.ptr, ds.ptr, GS, dy.ptr, M, K, None), "kx_gemv_w8_f32")
    total, recs = kx.bounds_report()
    new = recs[before:] if before < 1024 else []
    assert total > before and new
    assert all(r[0] == "gemv_w8.hip" and r[2] == 2 and r[3] == r[4] for r in new), new[:4]
    if before == 0:
        kx.check(kx.lib().kx_bounds_reset(), "kx_bounds_reset")
