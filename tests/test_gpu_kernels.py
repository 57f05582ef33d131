"""GPU parity tests: every kernel entry of the C ABI against the CPU oracle on identical seeded
inputs, against the committed golden fixtures, and through size-independent properties at
full model sizes.  Tolerances are stated per test; fp32 FMA paths use
|got - oracle| <= 1e-5 * sum|terms| (the reference's own CPU-vs-CUDA tests use 1e-5:
test/test_op/test_cu_rmsnorm.cpp:37)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kx  # noqa: E402
from oracle import kxo  # noqa: E402

GS = 64


def rand_w8(rng, K, M, gs=GS):
    w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
    s = rng.uniform(1e-4, 3e-4, size=(K * M) // gs).astype(np.float32)
    x = rng.standard_normal(M).astype(np.float32)
    return x, w, s


def w8_abs_sum(x, w, s, gs=GS):
    K, M = w.shape
    sd = np.repeat(s.astype(np.float64), gs).reshape(K, M)
    return (np.abs(x.astype(np.float64))[None, :] * sd * np.abs(w.astype(np.float64))).sum(1)


# ------------------------------- W8 GEMV -----------------------------------------------------
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_gemv_w8_golden(golden, name):
    g = golden("g1_quant_gemv.npz")
    x, q, s = g[f"{name}_x"], g[f"{name}_q"], g[f"{name}_s"]
    y = kx.gemv_w8(x, q, s, GS)
    tol = 1e-5 * w8_abs_sum(x, q, s)
    assert np.all(np.abs(y - g[f"{name}_y64"]) <= tol)


@pytest.mark.parametrize("K,M", [(1, 64), (3, 64), (7, 192), (64, 1024), (130, 2048), (512, 2048),
                                 (33, 3584), (256, 4096), (17, 14336), (8, 18944), (5, 28672),
                                 (2048, 2048),
                                 # rows per block that are not a whole round of waves (10 and 18
                                 # rows over 4 waves of 2): the last waves run one row set fewer
                                 (9000, 4096), (18000, 2048),
                                 # per-rank shapes of the tensor-parallel configs: 8B at TP=8
                                 # (Wo, W2), Qwen2.5-7B at TP=4 (Wo, W2), 70B at TP=8 (W2)
                                 (4096, 512), (4096, 1792), (3584, 896), (3584, 4736), (1024, 3584)])
def test_gemv_w8_vs_oracle(K, M):
    rng = np.random.default_rng(K * 100003 + M)
    x, w, s = rand_w8(rng, K, M)
    want = kxo.gemv_w8(x, w, s, GS)
    got = kx.gemv_w8(x, w, s, GS)
    assert np.all(np.abs(got - want) <= 1e-5 * w8_abs_sum(x, w, s))


@pytest.mark.parametrize("K,M,gs", [(6, 100, 50), (4, 48, 32), (9, 36, 12), (4, 200, 16)])
def test_gemv_w8_generic_groups(K, M, gs):
    # group sizes / row lengths the fast path does not take; groups may straddle rows
    rng = np.random.default_rng(gs)
    x, w, s = rand_w8(rng, K, M, gs)
    want = kxo.gemv_w8(x, w, s, gs)
    got = kx.gemv_w8(x, w, s, gs)
    assert np.all(np.abs(got - want) <= 1e-5 * w8_abs_sum(x, w, s, gs))


def test_gemv_w8_preconditions():
    L = kx.lib()
    b = kx.Buf(4096)
    assert L.kx_gemv_w8_f32(b.ptr, b.ptr, b.ptr, 64, b.ptr, 6, 4, None) == -1  # M % 4 != 0
    assert L.kx_gemv_w8_f32(None, b.ptr, b.ptr, 64, b.ptr, 64, 4, None) == -1
    assert L.kx_gemv_w8_f32(b.ptr, b.ptr, b.ptr, 0, b.ptr, 64, 4, None) == -1


def test_gemv_w8_ex_routing_bias_residual():
    rng = np.random.default_rng(5)
    dim, kvd = 512, 128
    K = dim + 2 * kvd
    x, w, s = rand_w8(rng, K, dim)
    bias = rng.standard_normal(K).astype(np.float32) * 1e-2
    want = kxo.gemv_w8(x, w, s, GS) + bias
    dx, dw, ds, db = map(kx.Buf.from_numpy, (x, w, s, bias))
    y0, y1, y2 = kx.Buf.zeros(dim), kx.Buf.zeros(kvd), kx.Buf.zeros(kvd)
    kx.check(kx.lib().kx_gemv_w8_f32_ex(dx.ptr, dw.ptr, ds.ptr, GS, y0.ptr, y1.ptr, y2.ptr, dim,
                                        dim + kvd, db.ptr, None, dim, K, None))
    got = np.concatenate([y0.numpy(), y1.numpy(), y2.numpy()])
    assert np.all(np.abs(got - want) <= 1e-5 * w8_abs_sum(x, w, s) + 1e-7)
    # residual add in place: y = W x + y
    res = rng.standard_normal(dim).astype(np.float32)
    dy = kx.Buf.from_numpy(res)
    kx.check(kx.lib().kx_gemv_w8_f32_ex(dx.ptr, dw.ptr, ds.ptr, GS, dy.ptr, None, None, dim, dim,
                                        None, dy.ptr, dim, dim, None))
    want2 = kxo.gemv_w8(x, w[:dim], s[:dim * dim // GS], GS) + res
    assert np.all(np.abs(dy.numpy() - want2) <= 1e-5 * w8_abs_sum(x, w[:dim], s[:dim * dim // GS])
                  + 1e-6)


@pytest.mark.parametrize("H,M", [(256, 512), (1000, 2048), (344, 4096), (1792, 4096), (4736, 3584),
                                 (14336, 4096)])  # the last: 14 pairs per block, waves take 4, 4, 3, 3
def test_gemv_w8_swiglu(H, M):
    rng = np.random.default_rng(H)
    x, w, s = rand_w8(rng, 2 * H, M)
    s *= 20  # push gate values into the non-linear part of the sigmoid
    full = kxo.gemv_w8(x, w, s, GS)
    want = kxo.swiglu(full[:H], full[H:])
    dx, dw, ds = map(kx.Buf.from_numpy, (x, w, s))
    out = kx.Buf.zeros(H)
    kx.check(kx.lib().kx_gemv_w8_swiglu_f32(dx.ptr, dw.ptr, ds.ptr, GS, out.ptr, M, H, None))
    a = w8_abs_sum(x, w, s)
    # d(silu(g)*u) <= |u| dg * 1.1 + |silu(g)| du
    tol = 1e-5 * (1.1 * np.abs(full[H:]) * a[:H] + np.abs(full[:H]) * a[H:]) + 1e-7
    assert np.all(np.abs(out.numpy() - want) <= tol)


def test_gemv_w8_full_size_linearity():
    # Llama-3-8B w1 shape: W(x1 + 2*x2) == W x1 + 2 W x2 and a row-subset check against the oracle
    rng = np.random.default_rng(11)
    K, M = 14336, 4096
    x1, w, s = rand_w8(rng, K, M)
    x2 = rng.standard_normal(M).astype(np.float32)
    dw, ds = kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)

    def run(x):
        dx, dy = kx.Buf.from_numpy(x), kx.Buf.zeros(K)
        kx.check(kx.lib().kx_gemv_w8_f32(dx.ptr, dw.ptr, ds.ptr, GS, dy.ptr, M, K, None))
        return dy.numpy()
    y1, y2, y12 = run(x1), run(x2), run(x1 + 2 * x2)
    a = w8_abs_sum(np.abs(x1) + 2 * np.abs(x2), w, s)
    assert np.all(np.abs(y12 - (y1 + 2 * y2)) <= 3e-5 * a)
    rows = rng.choice(K, 64, replace=False)
    for r in rows:
        want = kxo.gemv_w8(x1, w[r:r + 1], s[r * M // GS:(r + 1) * M // GS], GS)[0]
        assert abs(y1[r] - want) <= 1e-5 * w8_abs_sum(x1, w[r:r + 1], s[r * M // GS:(r + 1) * M // GS])


# ------------------------------- fp32 GEMV ---------------------------------------------------
def test_gemv_f32_reference_known_answers():
    # test/test_op/test_cu_matmul.cpp:48-106
    w = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    assert kx.gemv_f32(np.array([1, 1, -1], np.float32), w).tolist() == [0.0, 3.0, 6.0]
    # test_cu_matmul.cpp:10-46
    w = np.arange(16, dtype=np.float32).reshape(4, 4)
    x = np.arange(4, dtype=np.float32)
    assert kx.gemv_f32(x, w).tolist() == (w @ x).tolist()
    # test/test_op/test_load.cpp:97-105
    w = np.arange(2048, dtype=np.float32).reshape(16, 128)
    y = kx.gemv_f32(np.ones(128, np.float32), w)
    assert y[0] == 8128 and y[1] == 24512 and y[14] == 237504 and y[15] == 253888


@pytest.mark.parametrize("K,M", [(5, 7), (768, 768), (2048, 768), (768, 2048), (1000, 4100),
                                 (3, 16384)])
def test_gemv_f32_vs_oracle(K, M):
    rng = np.random.default_rng(K + M)
    w = rng.standard_normal((K, M)).astype(np.float32)
    x = rng.standard_normal(M).astype(np.float32)
    want = kxo.gemv_f32(x, w, 0.5)
    got = kx.gemv_f32(x, w, 0.5)
    tol = 1e-5 * (np.abs(w.astype(np.float64)) @ np.abs(x.astype(np.float64))) * 0.5
    assert np.all(np.abs(got - want) <= tol)


# ------------------------------- decode attention ----------------------------------------------
def _attn_cases(g):
    for i in range(int(g["n_cases"])):
        meta = g[f"c{i}_meta"]
        cid = int(meta[8])
        yield f"c{i}", meta, g[f"c{i}_q"], g[f"cache{cid}_k"], g[f"cache{cid}_v"], g[f"c{i}_out"]
    yield "ref", g["ref_meta"], g["ref_q"], g["ref_kc"], g["ref_vc"], g["ref_out"]


def test_mha_decode_golden(golden):
    g = golden("g5_attention.npz")
    ran = 0
    for name, meta, q, kc, vc, want in _attn_cases(g):
        H, KVH, hs, S, pos, Tq, L, layer = [int(v) for v in meta[:8]]
        if Tq != 1:
            continue
        got = kx.mha_decode(q.reshape(-1), kc, vc, pos, H, H // KVH, hs, layer)
        np.testing.assert_allclose(got.reshape(1, H, hs), want, rtol=0, atol=1e-5, err_msg=name)
        ran += 1
    assert ran >= 5


def test_mha_decode_reference_test_inputs_row_by_row(golden):
    # test/test_op/test_flash_attention.cpp:38-81 geometry (head_size 8 -> generic kernel)
    g = golden("g5_attention.npz")
    H, KVH, hs, S, pos, Tq, L, layer = [int(v) for v in g["ref_meta"][:8]]
    for i in range(Tq):
        got = kx.mha_decode(g["ref_q"][i].reshape(-1), g["ref_kc"], g["ref_vc"], i, H, 1, hs, 0)
        np.testing.assert_allclose(got, g["ref_out"][i], rtol=0, atol=1e-6)


@pytest.mark.parametrize("H,KVH,hs,S,pos", [
    (12, 12, 64, 256, 0), (12, 12, 64, 256, 15), (12, 12, 64, 256, 16), (12, 12, 64, 256, 255),
    (32, 8, 64, 600, 599), (32, 8, 128, 2100, 2047), (28, 4, 128, 700, 650), (64, 8, 128, 300, 17),
    (4, 2, 32, 32, 31), (8, 1, 64, 130, 129), (40, 2, 64, 64, 63)])
def test_mha_decode_vs_oracle(H, KVH, hs, S, pos):
    rng = np.random.default_rng(H * 7 + pos)
    L, layer = 2, 1
    kc = rng.standard_normal((L, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((L, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    want = kxo.mha(pos, H, layer, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, layer)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_mha_decode_long_context_several_tiles_per_wave():
    # past 8192 keys per kv head a wave walks several tiles with the next one prefetched
    rng = np.random.default_rng(90)
    H, KVH, hs, S, pos = 8, 1, 128, 9100, 9000
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("H,KVH,hs,S,pos", [
    (32, 8, 128, 4700, 4096), (32, 8, 128, 4700, 4367), (32, 8, 128, 6200, 6143),
    (12, 12, 64, 3100, 3000), (16, 4, 128, 8300, 8200),
    # 128-wide heads with 1..4 query heads per kv head (the 4x4x1 matrix-instruction form): one kv
    # head as on a rank of an 8-way split, several tiles per wave, fewer than four heads
    (4, 1, 128, 4700, 4367), (4, 1, 128, 9100, 9000), (6, 2, 128, 2100, 2047), (2, 2, 128, 300, 257),
    (3, 1, 128, 40, 33),
    # five to eight query heads per kv head: two groups of four on the same tile (Qwen2.5-7B: 28 / 4,
    # Llama-3-70B: 64 / 8), single tiles, several tiles per wave, a last group that is not full
    (28, 4, 128, 4700, 4367), (64, 8, 128, 4700, 4367), (8, 1, 128, 9100, 9000), (5, 1, 128, 300, 257),
    (12, 2, 128, 2100, 2047), (7, 1, 128, 40, 15), (14, 2, 128, 8300, 8200)])
def test_mha_decode_uneven_splits_past_two_workgroups_per_cu(H, KVH, hs, S, pos):
    # once one tile per wave would need more than two workgroups per CU the splits stop growing in
    # number: runs of near-equal length, the first `rem` one tile longer, some waves with two tiles
    rng = np.random.default_rng(pos)
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("H,KVH,hs,S,pos", [(28, 4, 128, 4700, 4367), (5, 1, 128, 300, 257), (12, 2, 128, 2100, 2047),
                                            (7, 1, 128, 40, 15), (14, 2, 128, 8300, 8200), (8, 1, 128, 700, 640)])
def test_mha_decode_two_head_groups_forced(H, KVH, hs, S, pos, monkeypatch):
    # the two-groups-of-four form is picked by itself only for eight heads per kv head on long
    # contexts; forced here for five to eight heads, short and long, last group not full
    monkeypatch.setenv("KX_ATTN_M4X2", "1")
    rng = np.random.default_rng(pos + H)
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    monkeypatch.setenv("KX_ATTN_M4X2", "0")
    same = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(same, want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("H,KVH,hs,S,pos", [(32, 8, 128, 4700, 4367), (4, 1, 128, 300, 257), (6, 2, 128, 2100, 2047),
                                            (3, 1, 128, 40, 15), (8, 1, 128, 9100, 9000), (16, 4, 128, 700, 640),
                                            (32, 8, 64, 4700, 4367), (8, 2, 64, 300, 257), (3, 1, 64, 40, 15),
                                            (4, 4, 64, 2100, 2047)])
def test_mha_decode_eight_wave_workgroups_forced(H, KVH, hs, S, pos, monkeypatch):
    # eight waves per workgroup (one workgroup per CU) is picked by itself only past the finest cut
    # of the four-wave form; forced here on short and long contexts, one and two head groups, and
    # through the device-position entry with a grid sized for a longer context
    monkeypatch.setenv("KX_ATTN_NW", "8")
    monkeypatch.setenv("KX_ATTN_M4X2", "1")
    rng = np.random.default_rng(pos * 3 + H)
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    L = kx.lib()
    dq, dk, dv = (kx.Buf.from_numpy(a) for a in (q, kc, vc))
    out = kx.Buf.zeros(H * hs)
    wsb = L.kx_attn_workspace_bytes(H, hs)
    ws = kx.Buf.zeros(wsb, np.uint8)
    pos_dev = kx.Buf.from_numpy(np.array([pos], np.int32))
    for _ in range(2):   # the tickets are back at zero after a launch
        kx.check(L.kx_mha_decode_dpos_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, S - 1, H, H // KVH, hs,
                                          KVH * hs, S, 0, ws.ptr, wsb, None), "dpos")
        np.testing.assert_allclose(out.numpy().reshape(H, hs), want, rtol=0, atol=2e-5)
    assert not ws.numpy()[:16384].any()


def test_mha_decode_device_position_grid_for_a_longer_context():
    # graph-replay form with the grid cut for 4736 rows (64 splits of 4..5 tiles at the end) and
    # positions on both sides of the point where the first split gets its fifth tile
    rng = np.random.default_rng(92)
    H, KVH, hs, S = 32, 8, 128, 4736
    kvd = KVH * hs
    kc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    vc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    L = kx.lib()
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(kc), kx.Buf.from_numpy(vc)
    out = kx.Buf.zeros(H * hs)
    wsb = L.kx_attn_workspace_bytes(H, hs)
    ws = kx.Buf.zeros(wsb, np.uint8)
    pos_dev = kx.Buf.zeros(1, np.int32)
    for pos in (4095, 4096, 4111, 3, 4367, 1000, 4623, 4735):
        pos_dev.upload(np.array([pos], np.int32))
        kx.check(L.kx_mha_decode_dpos_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, S - 1, H,
                                          H // KVH, hs, kvd, S, 0, ws.ptr, wsb, None))
        kx.sync()
        want = kxo.mha(pos, H, 0, S, kvd, H // KVH, hs, q, kc, vc).reshape(H, hs)
        np.testing.assert_allclose(out.numpy(np.float32, (H, hs)), want, rtol=0, atol=2e-5,
                                   err_msg=f"pos={pos}")
    assert not ws.numpy(np.uint8, (wsb,))[:16384].any()  # tickets are back at zero


def test_mha_decode_device_position_one_workspace_many_launches():
    # the form used under graph replay: the grid is sized for the longest context, the position
    # is read from device memory, and every launch shares one workspace whose arrival counters
    # must come back to zero whatever the number of splits was
    rng = np.random.default_rng(91)
    H, KVH, hs, S = 32, 8, 128, 4096
    kvd = KVH * hs
    kc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    vc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    L = kx.lib()
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(kc), kx.Buf.from_numpy(vc)
    out = kx.Buf.zeros(H * hs)
    wsb = L.kx_attn_workspace_bytes(H, hs)
    ws = kx.Buf.zeros(wsb, np.uint8)
    pos_dev = kx.Buf.zeros(1, np.int32)
    for pos in (5, 4000, 63, 64, 2047, 100, 4095, 0):
        pos_dev.upload(np.array([pos], np.int32))
        kx.check(L.kx_mha_decode_dpos_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, S - 1, H,
                                          H // KVH, hs, kvd, S, 0, ws.ptr, wsb, None))
        kx.sync()
        want = kxo.mha(pos, H, 0, S, kvd, H // KVH, hs, q, kc, vc).reshape(H, hs)
        np.testing.assert_allclose(out.numpy(np.float32, (H, hs)), want, rtol=0, atol=2e-5,
                                   err_msg=f"pos={pos}")
    assert not ws.numpy(np.uint8, (wsb,))[:16384].any()  # tickets are back at zero
    # a workspace left dirty (as by a launch aborted midway) is put right by the reset entry
    dirty = np.zeros(wsb, np.uint8)
    dirty[:16384] = rng.integers(1, 255, 16384)
    ws.upload(dirty)
    kx.check(L.kx_attn_workspace_reset(ws.ptr, wsb, None))
    pos_dev.upload(np.array([3000], np.int32))
    kx.check(L.kx_mha_decode_dpos_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, S - 1, H, H // KVH, hs, kvd,
                                      S, 0, ws.ptr, wsb, None))
    kx.sync()
    want = kxo.mha(3000, H, 0, S, kvd, H // KVH, hs, q, kc, vc).reshape(H, hs)
    np.testing.assert_allclose(out.numpy(np.float32, (H, hs)), want, rtol=0, atol=2e-5)
    assert L.kx_attn_workspace_reset(ws.ptr, 100, None) == -1


@pytest.mark.parametrize("H,KVH,hs,S,K,M", [(32, 8, 128, 4736, 4096, 4096), (32, 8, 128, 2304, 4000, 4096),
                                            (6, 2, 64, 1200, 1536, 384), (16, 2, 128, 4500, 2048, 2048)])
def test_mha_decode_with_hints(H, KVH, hs, S, K, M, monkeypatch):
    # the hinted entries (1) read the head of the granules of a following K x M GEMV (int8 rows and
    # scales, buffers of exactly that size: the bounds-checking build sees every touched address):
    # same bits as the plain entry, workspace tickets at zero, the touched buffers untouched; a
    # ragged last granule, a range that is not whole lines, a null range, a share of 100 %, a single
    # split and the 16-bit cache form included; (2) cut the launch for an upper bound of the
    # position and use the position as a mask only: the rows between the two are read with zero
    # weight (here they hold other random rows), bounds right at the position, a tile and many
    # tiles above it, whole splits masked out -- against the plain entry's result
    rng = np.random.default_rng(S + K)
    kvd = KVH * hs
    kc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    vc = rng.standard_normal((1, S, kvd)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    L = kx.lib()
    rpb = L.kx_gemv_w8_block_rows(M, K, GS)
    assert rpb > 0
    w = rng.integers(-127, 128, size=(K, M), dtype=np.int8)
    sc = rng.uniform(1e-4, 3e-4, size=K * M // GS).astype(np.float32)
    dw, dsc = kx.Buf.from_numpy(w), kx.Buf.from_numpy(sc)
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(kc), kx.Buf.from_numpy(vc)
    dk16, dv16 = kx.Buf.from_numpy(kc.astype(np.float16)), kx.Buf.from_numpy(vc.astype(np.float16))
    wsb = L.kx_attn_workspace_bytes(H, hs)
    ws = kx.Buf.zeros(wsb, np.uint8)
    pos_dev = kx.Buf.zeros(1, np.int32)
    g0, g1 = rpb * M, rpb * M // GS * 4

    def plain(pos, half):
        out = kx.Buf.zeros(H * hs)
        if half:
            kx.check(L.kx_mha_decode_kv16(dq.ptr, dk16.ptr, dv16.ptr, out.ptr, S - 1, pos_dev.ptr, H, H // KVH, hs,
                                          kvd, S, 0, ws.ptr, wsb, None))
        else:
            kx.check(L.kx_mha_decode_dpos_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, S - 1, H, H // KVH, hs,
                                              kvd, S, 0, ws.ptr, wsb, None))
        kx.sync()
        return out.numpy()

    def touching(pos, half, n0, b0, gr0, n1, b1, gr1, bound=None):
        out = kx.Buf.zeros(H * hs)
        top, fixed = (S - 1, 0) if bound is None else (bound, 1)
        if half:
            kx.check(L.kx_mha_decode_kv16_ex(dq.ptr, dk16.ptr, dv16.ptr, out.ptr, top, pos_dev.ptr, fixed, H,
                                             H // KVH, hs, kvd, S, 0, ws.ptr, wsb, n0, b0, gr0, n1, b1, gr1, None))
        else:
            kx.check(L.kx_mha_decode_dpos_ex_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos_dev.ptr, top, fixed, H,
                                                 H // KVH, hs, kvd, S, 0, ws.ptr, wsb, n0, b0, gr0, n1, b1, gr1, None))
        kx.sync()
        return out.numpy()

    for pos in (S - 1, S // 2, 40, 3):
        pos_dev.upload(np.array([pos], np.int32))
        for half in (False, True):
            base = plain(pos, half)
            if not half:
                want = kxo.mha(pos, H, 0, S, kvd, H // KVH, hs, q, kc, vc).reshape(-1)
                np.testing.assert_allclose(base, want, rtol=0, atol=2e-5)
            for pct in ("", "100", "3"):
                if pct:
                    monkeypatch.setenv("KX_ATTN_TOUCH_PCT", pct)
                else:
                    monkeypatch.delenv("KX_ATTN_TOUCH_PCT", raising=False)
                assert np.array_equal(touching(pos, half, dw.ptr, K * M, g0, dsc.ptr, sc.nbytes, g1), base)
            monkeypatch.delenv("KX_ATTN_TOUCH_PCT", raising=False)
            for bound in sorted({pos, min(pos + 1, S - 1), min(pos + 16, S - 1), min(pos + 255, S - 1), S - 1}):
                got = touching(pos, half, dw.ptr, K * M, g0, dsc.ptr, sc.nbytes, g1, bound)
                np.testing.assert_allclose(got, base, rtol=0, atol=2e-5, err_msg=f"pos={pos} half={half} bound={bound}")
            # one range only; a range that is not whole lines (ignored); granules that do not divide the range
            assert np.array_equal(touching(pos, half, None, 0, 0, dsc.ptr, sc.nbytes, g1), base)
            assert np.array_equal(touching(pos, half, dw.ptr, K * M - 64, g0, None, 0, 0), base)
            assert np.array_equal(touching(pos, half, dw.ptr, K * M, 3 * 128, dsc.ptr, sc.nbytes, 5 * 128), base)
    assert not ws.numpy(np.uint8, (wsb,))[:16384].any()
    assert np.array_equal(dw.numpy(np.int8, (K, M)), w) and np.array_equal(dsc.numpy(), sc)


def test_gemv_w8_block_rows_is_what_the_kernel_uses():
    # rows per workgroup of the plain GEMV; 0 for shapes that take the generic kernel
    L = kx.lib()
    assert L.kx_gemv_w8_block_rows(4096, 4096, 64) == 8
    assert L.kx_gemv_w8_block_rows(14336, 4096, 64) == 16
    assert L.kx_gemv_w8_block_rows(4096, 128256, 64) == 126
    assert L.kx_gemv_w8_block_rows(2048, 2048, 64) == 8
    assert L.kx_gemv_w8_block_rows(100, 64, 4) == 0
    assert L.kx_gemv_w8_block_rows(0, 64, 64) == 0


def test_mha_decode_forced_rescale_and_no_workspace():
    # one key far above the rest late in the sequence forces the online-softmax rescale branch;
    # with no workspace the kernel must fall back to a single split and still be right
    rng = np.random.default_rng(3)
    H, KVH, hs, S, pos = 8, 2, 128, 512, 500
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32) * 0.1
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    kc[0, 333, :hs] = q[:hs] * 3.0   # spike for heads sharing kv head 0
    want = kxo.mha(pos, H, 0, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    for wsb in (None, 0):
        got = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0, workspace_bytes=wsb)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_mha_decode_ignores_rows_past_pos():
    rng = np.random.default_rng(4)
    H, KVH, hs, S, pos = 8, 2, 64, 128, 37
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    a = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    kc[0, pos + 1:] = np.nan
    vc[0, pos + 1:] = np.inf
    b = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0)
    assert np.array_equal(a, b)


def test_attention_and_gemm_preconditions():
    # every entry checks its operands on the host and returns a code; nothing is launched
    L = kx.lib()
    b = kx.Buf(1 << 16)
    ws = kx.Buf.zeros(1 << 16, np.uint8)
    EINVAL, EUNSUP, EWS = -1, -2, -3
    # position outside the cache, heads not divisible by kv_mul, kv_dim inconsistent
    assert L.kx_mha_decode_f32(b.ptr, b.ptr, b.ptr, b.ptr, 16, 4, 2, 32, 64, 16, 0, ws.ptr, 1 << 16, None) == EINVAL
    assert L.kx_mha_decode_f32(b.ptr, b.ptr, b.ptr, b.ptr, 3, 5, 2, 32, 64, 16, 0, ws.ptr, 1 << 16, None) == EINVAL
    assert L.kx_mha_decode_f32(b.ptr, b.ptr, b.ptr, b.ptr, 3, 4, 2, 32, 96, 16, 0, ws.ptr, 1 << 16, None) == EINVAL
    assert L.kx_mha_decode_f32(None, b.ptr, b.ptr, b.ptr, 3, 4, 2, 32, 64, 16, 0, ws.ptr, 1 << 16, None) == EINVAL
    # a head size the tiled kernel is not built for needs the [head][seq_len] score scratch
    assert L.kx_mha_decode_f32(b.ptr, b.ptr, b.ptr, b.ptr, 3, 4, 2, 8, 16, 16, 0, None, 0, None) == EWS
    # ... and cannot read the position from device memory
    assert L.kx_mha_decode_dpos_f32(b.ptr, b.ptr, b.ptr, b.ptr, b.ptr, 15, 4, 2, 8, 16, 16, 0, ws.ptr,
                                    1 << 16, None) == EUNSUP
    assert L.kx_mha_decode_dpos_f32(b.ptr, b.ptr, b.ptr, b.ptr, None, 15, 4, 2, 32, 64, 16, 0, ws.ptr,
                                    1 << 16, None) == EINVAL
    # flash attention: more query rows than positions, bad head grouping
    scale = float(1 / np.sqrt(32))
    assert L.kx_flash_attention_f32(b.ptr, b.ptr, b.ptr, b.ptr, 8, 4, 2, 32, 16, 3, 0, scale, 1, ws.ptr,
                                    1 << 16, None) == EINVAL
    assert L.kx_flash_attention_f32(b.ptr, b.ptr, b.ptr, b.ptr, 2, 4, 3, 32, 16, 3, 0, scale, 1, ws.ptr,
                                    1 << 16, None) == EINVAL
    # single rows go through the decode kernel, which takes the default scale only
    assert L.kx_flash_attention_f32(b.ptr, b.ptr, b.ptr, b.ptr, 1, 4, 2, 32, 16, 3, 0, 0.5, 1, ws.ptr,
                                    1 << 16, None) == EUNSUP
    # GEMM: a routed column range without its destination, split points out of order
    assert L.kx_gemm_w8_f32_ex(b.ptr, 64, b.ptr, b.ptr, 64, b.ptr, 64, b.ptr, 64, None, 0, 8, 16, None, 0,
                               4, 64, 32, None) == EINVAL
    assert L.kx_gemm_w8_f32_ex(b.ptr, 64, b.ptr, b.ptr, 64, b.ptr, 64, b.ptr, 64, b.ptr, 64, 16, 8, None,
                               0, 4, 64, 32, None) == EINVAL
    assert L.kx_gemm_w8_f32(None, b.ptr, b.ptr, 64, b.ptr, 4, 64, 32, None) == EINVAL
    assert L.kx_gemm_w8_workspace_bytes(0, 64, 64) == 0
    assert L.kx_flash_attention_workspace_bytes(-1, 2, 64) == 0
    kx.sync()


# ------------------------------- support kernels -------------------------------------------------
def _run(fn, *args):
    kx.check(fn(*args))
    kx.sync()


@pytest.mark.parametrize("dim,rows,eps", [(768, 1, 1e-5), (4096, 3, 1e-5), (3584, 2, 1e-6),
                                          (130, 1, 1e-5)])
def test_rmsnorm(dim, rows, eps):
    rng = np.random.default_rng(dim)
    x = rng.standard_normal((rows, dim)).astype(np.float32)
    w = rng.standard_normal(dim).astype(np.float32)
    dx, dw, do = kx.Buf.from_numpy(x), kx.Buf.from_numpy(w), kx.Buf.zeros((rows, dim))
    _run(kx.lib().kx_rmsnorm_f32, dx.ptr, dw.ptr, do.ptr, dim, rows, eps, None)
    want = np.stack([kxo.rmsnorm(x[r], w, eps) for r in range(rows)])
    np.testing.assert_allclose(do.numpy(), want, rtol=1e-5, atol=1e-6)  # test_cu_rmsnorm.cpp:37


@pytest.mark.parametrize("variant,theta,hs", [(0, 1e4, 64), (1, 5e5, 64), (1, 5e5, 128),
                                              (1, 1e6, 128), (0, 1e4, 32)])
def test_rope_and_table(variant, theta, hs):
    rng = np.random.default_rng(int(theta) + hs)
    H, KVH, S, rows, pos = 8, 2, 64, 3, 20
    dim, kvd = H * hs, KVH * hs
    ds_, dc_ = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    _run(kx.lib().kx_sin_cos_cache_f32, hs, S, theta, ds_.ptr, dc_.ptr, None)
    s_want, c_want = kxo.sin_cos_cache(hs, S, theta)
    # device powf/sinf/cosf vs libm: the reference's own rope test uses 1e-3 (test_cu_rope.cpp:43)
    np.testing.assert_allclose(ds_.numpy(), s_want, atol=2e-4)
    np.testing.assert_allclose(dc_.numpy(), c_want, atol=2e-4)
    # rotate with the ORACLE's tables so the rotation itself is checked tightly
    ds_.upload(s_want)
    dc_.upload(c_want)
    q = rng.standard_normal((rows, dim)).astype(np.float32)
    k_stride = kvd + 16
    k = rng.standard_normal((rows, k_stride)).astype(np.float32)
    dq, dk = kx.Buf.from_numpy(q), kx.Buf.from_numpy(k)
    _run(kx.lib().kx_rope_f32, variant, dim, kvd, hs, dq.ptr, dk.ptr, k_stride, pos, rows, ds_.ptr,
         dc_.ptr, None)
    gq, gk = dq.numpy(), dk.numpy()
    for r in range(rows):
        wq, wk = kxo.rope(variant, dim, kvd, hs, q[r], k[r, :kvd], pos + r, s_want, c_want)
        np.testing.assert_allclose(gq[r], wq, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(gk[r, :kvd], wk, rtol=1e-6, atol=1e-6)
        assert np.array_equal(gk[r, kvd:], k[r, kvd:])  # padding between rows untouched


def test_swiglu_add_embedding_argmax():
    rng = np.random.default_rng(9)
    n = 14336
    a = rng.standard_normal(n).astype(np.float32) * 3
    b = rng.standard_normal(n).astype(np.float32)
    da, db, do = kx.Buf.from_numpy(a), kx.Buf.from_numpy(b), kx.Buf.zeros(n)
    _run(kx.lib().kx_swiglu_f32, da.ptr, db.ptr, do.ptr, n, None)
    np.testing.assert_allclose(do.numpy(), kxo.swiglu(a, b), rtol=1e-5, atol=1e-6)  # test_cu_swiglu.cpp:39
    _run(kx.lib().kx_add_f32, da.ptr, db.ptr, do.ptr, n, None)
    assert np.array_equal(do.numpy(), a + b)
    table = rng.standard_normal((1000, 768)).astype(np.float32)
    toks = np.array([5, 999, 0, 5], np.int32)
    dt, dtok, dout = kx.Buf.from_numpy(table), kx.Buf.from_numpy(toks), kx.Buf.zeros((4, 768))
    _run(kx.lib().kx_embedding_f32, dtok.ptr, 4, dt.ptr, 768, 1000, dout.ptr, None)
    assert np.array_equal(dout.numpy(), table[toks])
    for nn in (1, 63, 1024, 128256):
        lg = rng.standard_normal(nn).astype(np.float32)
        if nn > 100:
            lg[77] = lg[90] = 50.0  # tie: first maximum wins
        dl, di = kx.Buf.from_numpy(lg), kx.Buf.zeros(1, np.int32)
        _run(kx.lib().kx_argmax_f32, dl.ptr, nn, di.ptr, None)
        assert int(di.numpy()[0]) == kxo.argmax(lg)


def test_split_classifier_candidates_pick_the_first_maximum():
    # four "ranks" each reduce their slice to a {value, global index} candidate; the pick over the
    # gathered candidates is the first maximum of the whole vector, ties across slices included
    rng = np.random.default_rng(12)
    L = kx.lib()
    n, ranks = 128256, 4
    per = n // ranks
    for case in range(3):
        lg = rng.standard_normal(n).astype(np.float32)
        if case == 1:
            lg[[per - 1, per, 3 * per + 7]] = 9.0      # tie across three slices: lowest index wins
        if case == 2:
            lg[:] = -np.inf                            # nothing finite anywhere: token 0
        dl = kx.Buf.from_numpy(lg)
        cands = kx.Buf.zeros((ranks, 2))
        ws = kx.Buf.zeros(L.kx_step_workspace_bytes(), np.uint8)
        for r in range(ranks):
            kx.check(L.kx_argmax_candidate_f32(dl.at(r * per * 4), per, r * per, cands.at(r * 8), ws.ptr,
                                               ws.nbytes, None))
        tok, pos = kx.Buf.zeros(1, np.int32), kx.Buf.from_numpy(np.array([5], np.int32))
        hist = kx.Buf.zeros(16, np.int32)
        kx.check(L.kx_decode_step_end_candidates(cands.ptr, ranks, tok.ptr, pos.ptr, hist.ptr, 16, None))
        kx.sync()
        want = kxo.argmax(lg) if case != 2 else 0
        assert int(tok.numpy()[0]) == want and int(pos.numpy()[0]) == 6 and int(hist.numpy()[6]) == want
    assert L.kx_argmax_candidate_f32(dl.ptr, per, 1 << 24, cands.ptr, ws.ptr, ws.nbytes, None) == -2


def test_step_end_in_one_launch_with_the_next_embedding():
    # greedy pick + bookkeeping + the next step's embedding row, one launch, the same workspace
    # reused call after call (its arrival counter must come back to zero every time)
    rng = np.random.default_rng(21)
    L = kx.lib()
    dim, vocab = 2048, 128256
    table = rng.standard_normal((vocab, 8)).astype(np.float32)  # narrow table, wide enough to tell rows apart
    table = np.ascontiguousarray(np.tile(table, (1, dim // 8)))
    dt = kx.Buf.from_numpy(table)
    ws = kx.Buf.zeros(L.kx_step_workspace_bytes(), np.uint8)
    hist_len = 32
    for case, n in enumerate((vocab, vocab, 1000, 200, vocab)):
        lg = rng.standard_normal(n).astype(np.float32)
        if case == 1:
            lg[[300, 70000, 128000]] = 11.0          # a tie across workgroups: the lowest index wins
        if case == 4:
            lg[:] = -np.inf                          # nothing finite: token 0
        want = kxo.argmax(lg) if case != 4 else 0
        dl = kx.Buf.from_numpy(lg)
        tok, pos = kx.Buf.zeros(1, np.int32), kx.Buf.from_numpy(np.array([7 + case], np.int32))
        h0 = rng.integers(0, n, hist_len).astype(np.int32)
        hist = kx.Buf.from_numpy(h0)
        x = kx.Buf.zeros(dim)
        write = 0 if case == 2 else 1                # case 2: a prompt row, the history keeps its token
        kx.check(L.kx_decode_step_end_embed(dl.ptr, n, tok.ptr, pos.ptr, hist.ptr, hist_len, write, dt.ptr, dim,
                                            vocab, x.ptr, ws.ptr, ws.nbytes, None))
        kx.sync()
        p1 = 8 + case
        assert int(tok.numpy()[0]) == want and int(pos.numpy()[0]) == p1
        h = hist.numpy()
        nxt = want if write else int(h0[p1])
        assert int(h[p1]) == nxt and np.array_equal(np.delete(h, p1), np.delete(h0, p1))
        assert np.array_equal(x.numpy(), table[nxt])
        assert not ws.numpy(np.uint8, (ws.nbytes,))[-256:].any()      # counter back at zero
        # the plain form: same pick, no embedding, NULL history allowed
        tok2, pos2 = kx.Buf.zeros(1, np.int32), kx.Buf.from_numpy(np.array([3], np.int32))
        kx.check(L.kx_decode_step_end(dl.ptr, n, tok2.ptr, pos2.ptr, None, 0, ws.ptr, ws.nbytes, None))
        kx.sync()
        assert int(tok2.numpy()[0]) == want and int(pos2.numpy()[0]) == 4
    # past the end of the history nothing is gathered or stored
    pos = kx.Buf.from_numpy(np.array([hist_len - 1], np.int32))
    x = kx.Buf.zeros(dim)
    kx.check(L.kx_decode_step_end_embed(dl.ptr, n, tok.ptr, pos.ptr, hist.ptr, hist_len, 1, dt.ptr, dim, vocab,
                                        x.ptr, ws.ptr, ws.nbytes, None))
    kx.sync()
    assert int(pos.numpy()[0]) == hist_len and not x.numpy().any() and np.array_equal(hist.numpy(), h)
    assert L.kx_decode_step_end_embed(dl.ptr, n, tok.ptr, pos.ptr, None, 0, 1, dt.ptr, dim, vocab, x.ptr,
                                      ws.ptr, ws.nbytes, None) == -1
    assert L.kx_decode_step_end(dl.ptr, n, tok.ptr, pos.ptr, None, 0, ws.ptr, 64, None) == -3


@pytest.mark.parametrize("K,M,tie", [(128256, 512, False), (5000, 2048, True), (3001 * 2, 4096, False), (640, 1024, True)])
def test_classifier_gemv_that_ends_the_step(K, M, tie):
    # one launch = classifier GEMV (RMS norm of x included) + greedy pick + bookkeeping + next
    # embedding row, x_out being the GEMV's own input: logits, token, position, history and x_out
    # equal to those of the two separate launches bit for bit; all three kernel forms (4 / 2 / 1 rows
    # per wave), ties across workgroups (lowest row wins), a prompt row (history kept), and a shape
    # the fast GEMV form does not take (refused: the caller runs the two launches)
    rng = np.random.default_rng(K + M)
    L = kx.lib()
    _, w, s = rand_w8(rng, K, M)
    if tie:
        w[K - 3] = w[7]
        s.reshape(K, M // GS)[K - 3] = s.reshape(K, M // GS)[7]
    x0 = rng.standard_normal(M).astype(np.float32)
    nw = rng.uniform(0.5, 1.5, M).astype(np.float32)
    table = rng.standard_normal((K, M)).astype(np.float32) if K * M <= 64e6 else None
    dw, dsc, dn = kx.Buf.from_numpy(w), kx.Buf.from_numpy(s), kx.Buf.from_numpy(nw)
    dt = kx.Buf.from_numpy(table) if table is not None else None
    ws = kx.Buf.zeros(L.kx_step_workspace_bytes(), np.uint8)
    hist_len = 16

    def run(fused, write):
        x = kx.Buf.from_numpy(x0)
        y = kx.Buf.zeros(K)
        tok, pos = kx.Buf.zeros(1, np.int32), kx.Buf.from_numpy(np.array([4], np.int32))
        hist = kx.Buf.from_numpy((np.arange(hist_len) * 37 % K).astype(np.int32))
        d = kx.make_gemv_desc(0, x.ptr, dw.ptr, dsc.ptr, y.ptr, M, K, GS, norm_w=dn.ptr, k0=K, k1=K)
        tp = dt.ptr if dt else None
        if fused:
            rc = L.kx_gemv_w8_classify_pick_f32(C.byref(d), tok.ptr, pos.ptr, hist.ptr, hist_len, write, tp, M, K,
                                                x.ptr if dt else None, ws.ptr, ws.nbytes, None)
            if rc != 0:
                return rc
        else:
            kx.check(L.kx_gemv_w8_fused_f32(C.byref(d), None))
            kx.check(L.kx_decode_step_end_embed(y.ptr, K, tok.ptr, pos.ptr, hist.ptr, hist_len, write, tp, M, K,
                                                x.ptr if dt else None, ws.ptr, ws.nbytes, None))
        kx.sync()
        assert not ws.numpy(np.uint8, (ws.nbytes,))[-256:].any()
        return y.numpy(), int(tok.numpy()[0]), int(pos.numpy()[0]), hist.numpy(), x.numpy()

    for write in (1, 0, 1):
        a, b = run(True, write), run(False, write)
        assert np.array_equal(a[0], b[0]) and a[1:3] == b[1:3]
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        assert a[1] == kxo.argmax(a[0]) and a[2] == 5
        if tie:
            assert a[0][7] == a[0][K - 3]
    # preconditions: routed rows and a residual are refused
    x, y = kx.Buf.from_numpy(x0), kx.Buf.zeros(K)
    tok, pos = kx.Buf.zeros(1, np.int32), kx.Buf.zeros(1, np.int32)
    d = kx.make_gemv_desc(0, x.ptr, dw.ptr, dsc.ptr, y.ptr, M, K, GS, residual=y.ptr, k0=K, k1=K)
    assert L.kx_gemv_w8_classify_pick_f32(C.byref(d), tok.ptr, pos.ptr, None, 0, 1, None, 0, 0, None, ws.ptr,
                                          ws.nbytes, None) == -1
    d = kx.make_gemv_desc(0, x.ptr, dw.ptr, dsc.ptr, y.ptr, M, K, GS, k0=K, k1=K)
    assert L.kx_gemv_w8_classify_pick_f32(C.byref(d), tok.ptr, pos.ptr, None, 0, 1, None, 0, 0, None, ws.ptr, 64,
                                          None) == -3
    d = kx.make_gemv_desc(0, x.ptr, dw.ptr, dsc.ptr, y.ptr, 80, 4, GS, k0=4, k1=4)   # rows of 80: the generic kernel
    assert L.kx_gemv_w8_classify_pick_f32(C.byref(d), tok.ptr, pos.ptr, None, 0, 1, None, 0, 0, None, ws.ptr,
                                          ws.nbytes, None) == -2


def test_streams_events_graph_replay():
    rng = np.random.default_rng(1)
    K, M = 256, 1024
    x, w, s = rand_w8(rng, K, M)
    dx, dw, ds, dy = kx.Buf.from_numpy(x), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s), kx.Buf.zeros(K)
    st = kx.Stream()
    L = kx.lib()
    kx.check(L.kx_graph_begin_capture(st.ptr))
    kx.check(L.kx_gemv_w8_f32(dx.ptr, dw.ptr, ds.ptr, GS, dy.ptr, M, K, st.ptr))
    ge = C.c_void_p()
    kx.check(L.kx_graph_end_capture(st.ptr, C.byref(ge)))
    assert np.all(dy.numpy() == 0)  # capture records, it does not run
    e0, e1 = kx.Event(), kx.Event()
    e0.record(st.ptr)
    kx.check(L.kx_graph_launch(ge, st.ptr))
    e1.record(st.ptr)
    e1.sync()
    assert e0.elapsed_ms(e1) >= 0
    assert np.all(np.abs(dy.numpy() - kxo.gemv_w8(x, w, s, GS)) <= 1e-5 * w8_abs_sum(x, w, s))
    kx.check(L.kx_graph_destroy(ge))


# ------------------------------- prefill: W8A16 MFMA GEMM ----------------------------------------
# Tolerance: the only deviation from the fp32 formula is the fp16 rounding of X (relative 2^-11
# per element), so |err| <= 2^-11 * sum|terms| is a hard bound; random signs land far below it.
@pytest.mark.parametrize("T,K,M", [(4, 128, 64), (37, 200, 128), (128, 128, 256), (130, 384, 512),
                                   (300, 1000, 1024), (64, 6144, 4096)])
def test_gemm_w8_vs_oracle(T, K, M):
    rng = np.random.default_rng(T * 7 + K)
    _, w, s = rand_w8(rng, K, M)
    X = rng.standard_normal((T, M)).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    got = kx.gemm_w8(X, w, s, GS)
    sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
    absum = np.abs(X.astype(np.float64)) @ sd.T
    err = np.abs(got - want)
    assert np.all(err <= 2.0 ** -11 * absum + 1e-7)
    assert np.sqrt((err ** 2).mean()) <= 1e-4 * np.sqrt((absum ** 2).mean())


def _w8_tol(X, w, s, want):
    K, M = w.shape
    sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
    return 2.0 ** -11 * (np.abs(X.astype(np.float64)) @ sd.T) + 1e-7 * np.abs(want).max()


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("T,K,M", [(2, 384, 1024), (16, 200, 1536), (17, 130, 2304), (33, 640, 512),
                                   (48, 96, 256), (64, 1000, 1280), (65, 300, 768), (96, 257, 1024),
                                   (100, 130, 2048), (128, 520, 1280), (130, 200, 512), (160, 130, 1024),
                                   (192, 300, 768)])
def test_gemm_w8_few_rows_vs_oracle(monkeypatch, half, T, K, M):
    # 2..192 rows against the int8 weight stream: same arithmetic as the tiled int8 kernel (rows rounded
    # to fp16, fp32 group sums times fp32 scales); every cut of the reduction gives the bound, the
    # same bits when repeated, and leaves the arrival tickets at zero
    rng = np.random.default_rng(T * 1000 + K + M)
    _, w, s = rand_w8(rng, K, M)
    X = rng.standard_normal((T, M)).astype(np.float32)
    if half:
        X = X.astype(np.float16).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    tol = _w8_tol(X, w, s, want)
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    for parts in ("", "1", "2", "4", "8"):
        if parts and M // 128 < int(parts):
            continue
        if parts:
            monkeypatch.setenv("KX_ROWS_SPLIT", parts)
        else:
            monkeypatch.delenv("KX_ROWS_SPLIT", raising=False)
        got = kx.gemm_w8_rows(X, w, s, GS, split_ws=sp, half=half)
        assert np.all(np.abs(got - want) <= tol), parts
        np.testing.assert_array_equal(got, kx.gemm_w8_rows(X, w, s, GS, split_ws=sp, half=half))
    assert not sp.numpy()[:4096].any()


def test_gemm_w8_few_rows_routing_residual_and_refusals():
    # weight-row ranges to three destinations (q | k | v) and a residual on the first, as the tiled
    # kernels take them; shapes the kernel is not built for are refused, not approximated
    rng = np.random.default_rng(11)
    T, K, M, k0, k1 = 40, 640, 768, 384, 512
    _, w, s = rand_w8(rng, K, M)
    X = rng.standard_normal((T, M)).astype(np.float16).astype(np.float32)
    res = rng.standard_normal((T, k0)).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    tol = _w8_tol(X, w, s, want)
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    for half in (False, True):
        y0, y1, y2 = kx.gemm_w8_rows(X, w, s, GS, split_ws=sp, half=half, k0=k0, k1=k1, residual=res)
        assert np.all(np.abs(y0 - (want[:, :k0] + res)) <= tol[:, :k0] + 1e-6)
        assert np.all(np.abs(y1 - want[:, k0:k1]) <= tol[:, k0:k1])
        assert np.all(np.abs(y2 - want[:, k1:]) <= tol[:, k1:])
    dx, dw, ds = kx.Buf.from_numpy(X), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)
    dy = kx.Buf.zeros((80, K))
    call = lambda T_, M_, gs: kx.lib().kx_gemm_w8_rows_f32(dx.ptr, M_, dw.ptr, ds.ptr, gs, dy.ptr, K, None, 0, None,
                                                           0, K, K, None, 0, T_, M_, K, sp.ptr, sp.nbytes, None)
    # -2 = KX_EUNSUPPORTED
    assert call(193, M, GS) == -2         # more than 192 rows
    assert call(T, 192, GS) == -2         # reduction not a multiple of 256
    assert call(T, M, 128) == -2          # other group sizes
    assert kx.lib().kx_gemm_w8_rows_f32(dx.ptr, M, dw.ptr, ds.ptr, GS, dy.ptr, K, None, 0, None, 0, K, K, None,
                                        0, T, M, K, None, 0, None) == -2  # fp32 rows need the workspace


@pytest.mark.parametrize("T,K,M,k0,k1", [(48, 6144, 4096, 4096, 5120), (64, 4096, 14336, 4096, 4096),
                                          (24, 28672, 4096, 28672, 28672), (32, 3072, 2048, 2048, 2560),
                                          (128, 4096, 14336, 4096, 4096), (90, 6144, 4096, 4096, 5120),
                                          (128, 28672, 4096, 28672, 28672), (192, 4096, 14336, 4096, 4096),
                                          (150, 6144, 4096, 4096, 5120)])
def test_gemm_w8_few_rows_full_size(T, K, M, k0, k1):
    # Llama-3-8B (and one Llama-3.2-1B) matrices at full size, checked on a sample of weight rows
    rng = np.random.default_rng(K + M)
    _, w, s = rand_w8(rng, K, M)
    X = rng.standard_normal((T, M)).astype(np.float16)
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    if k0 < K:
        y = np.concatenate(kx.gemm_w8_rows(X.astype(np.float32), w, s, GS, split_ws=sp, half=True, k0=k0, k1=k1),
                           axis=1)
    else:
        y = kx.gemm_w8_rows(X.astype(np.float32), w, s, GS, split_ws=sp, half=True)
    rows = np.unique(np.r_[0, K - 1, rng.integers(0, K, 200)])
    ws, ss = w[rows], s.reshape(K, M // GS)[rows].reshape(-1)
    want = kxo.gemm_w8(X.astype(np.float32), ws, ss, GS, threads=8)
    assert np.all(np.abs(y[:, rows] - want) <= _w8_tol(X.astype(np.float32), ws, ss, want))
    assert not sp.numpy()[:4096].any()


@pytest.mark.parametrize("T,M,H", [(5, 512, 128), (40, 768, 320), (64, 256, 64), (100, 1024, 192), (192, 512, 256)])
def test_gemm_w8_few_rows_swiglu_in_the_epilogue(monkeypatch, T, M, H):
    # [W1; W3] product and SwiGLU in one launch: fp16(silu(gate) * up) of the plain few-rows product's
    # totals -- the same bits, whichever way the reduction is cut
    rng = np.random.default_rng(T + M + H)
    L = kx.lib()
    _, w, s = rand_w8(rng, 2 * H, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
    dx, dw, dsc = kx.Buf.from_numpy(X16), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)
    for parts in ("", "1", "2", "4"):
        if parts and M // 128 < int(parts):
            continue
        if parts:
            monkeypatch.setenv("KX_ROWS_SPLIT", parts)
        else:
            monkeypatch.delenv("KX_ROWS_SPLIT", raising=False)
        plain = kx.gemm_w8_rows(X16.astype(np.float32), w, s, GS, split_ws=sp, half=True)
        g, u = plain[:, :H], plain[:, H:]
        with np.errstate(over="ignore"):
            want = (g * (np.float32(1.0) / (np.float32(1.0) + np.exp(-g))) * u).astype(np.float32)
        out = kx.Buf.from_numpy(np.full((T, H + 8), 3.0, np.float16))   # rows wider than H: padding untouched
        kx.check(L.kx_gemm_w8_rows_swiglu_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, out.ptr, H + 8, T, M, H, sp.ptr,
                                                 sp.nbytes, None))
        kx.sync()
        got = out.numpy(np.float16, (T, H + 8))
        assert np.all(got[:, H:] == np.float16(3.0))
        # expf on the device against numpy's exp: an ulp or two in fp32, then one fp16 rounding
        np.testing.assert_allclose(got[:, :H].astype(np.float32), want, rtol=2.0 ** -10, atol=1e-6, err_msg=parts)
    assert not sp.numpy()[:4096].any()
    assert L.kx_gemm_w8_rows_swiglu_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, out.ptr, 96, T, M, 96, sp.ptr,
                                           sp.nbytes, None) == -2   # H not a multiple of 64


@pytest.mark.parametrize("hs,variant,T,pos0,with_bias", [(128, 1, 100, 5, False), (64, 1, 192, 0, True),
                                                         (64, 0, 33, 17, True), (128, 0, 16, 3, False),
                                                         (32, 1, 70, 2, True), (32, 0, 7, 0, False)])
def test_gemm_w8_few_rows_qkv_with_bias_and_rope_in_the_epilogue(monkeypatch, hs, variant, T, pos0, with_bias):
    # one launch = few-rows product + bias + rotary embedding + q / K-row / V-row placement, against the
    # plain few-rows product followed by the oracle's bias-then-rope
    rng = np.random.default_rng(hs + T)
    L = kx.lib()
    M, H, KVH = 512, 8, 4
    ql, kl = H * hs, KVH * hs
    K = ql + 2 * kl
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    bias = rng.standard_normal(K).astype(np.float32) * 0.1 if with_bias else None
    S = pos0 + T + 3
    ds_, dc_ = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    kx.check(L.kx_sin_cos_cache_f32(hs, S, 10000.0, ds_.ptr, dc_.ptr, None))
    s_tab, c_tab = ds_.numpy(), dc_.numpy()
    sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
    plain = kx.gemm_w8_rows(X16.astype(np.float32), w, s, GS, split_ws=sp, half=True)
    if with_bias:
        plain = plain + bias
    dx, dw, dsc = kx.Buf.from_numpy(X16), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)
    db = kx.Buf.from_numpy(bias) if with_bias else None
    kst = kl + 64                                       # K / V rows live in a wider cache
    for parts in ("", "2"):
        if parts:
            monkeypatch.setenv("KX_ROWS_SPLIT", parts)
        q = kx.Buf.zeros((T, ql))
        kc = kx.Buf.from_numpy(np.full((S, kst), 7.0, np.float32))
        vc = kx.Buf.from_numpy(np.full((S, kst), 9.0, np.float32))
        kx.check(L.kx_gemm_w8_rows_qkv_rope_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, q.ptr, ql, kc.at(pos0 * kst * 4),
                                                   kst, vc.at(pos0 * kst * 4), kst, ql, kl, hs, variant, ds_.ptr,
                                                   dc_.ptr, pos0, db.ptr if db else None, T, M, sp.ptr, sp.nbytes,
                                                   None))
        kx.sync()
        gq, gk, gv = q.numpy(), kc.numpy(), vc.numpy()
        tol = 2e-5 * np.abs(plain).max()
        for r in range(T):
            wq, wk = kxo.rope(variant, ql, kl, hs, plain[r, :ql].copy(), plain[r, ql:ql + kl].copy(), pos0 + r,
                              s_tab, c_tab)
            np.testing.assert_allclose(gq[r], wq, rtol=1e-5, atol=tol)
            np.testing.assert_allclose(gk[pos0 + r, :kl], wk, rtol=1e-5, atol=tol)
            np.testing.assert_allclose(gv[pos0 + r, :kl], plain[r, ql + kl:], rtol=1e-6, atol=tol)
        # rows outside the chunk and the padding beside the rows are untouched
        assert np.all(gk[:pos0] == 7.0) and np.all(gk[pos0 + T:] == 7.0) and np.all(gk[:, kl:] == 7.0)
        assert np.all(gv[:pos0] == 9.0) and np.all(gv[pos0 + T:] == 9.0) and np.all(gv[:, kl:] == 9.0)
    assert not sp.numpy()[:4096].any()
    # dimensions that are not whole 128-row tiles are refused, not mangled
    assert L.kx_gemm_w8_rows_qkv_rope_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, q.ptr, ql, kc.ptr, kst, vc.ptr, kst,
                                             ql - 64, kl + 32, hs, variant, ds_.ptr, dc_.ptr, 0, None, T, M, sp.ptr,
                                             sp.nbytes, None) == -2


def test_gemm_w8_few_rows_fused_forms_full_size():
    # Llama-3-8B's W1|W3 and Wq|Wk|Wv at full size, 128 rows: the fused epilogues on the totals of the
    # plain few-rows product (itself checked against the oracle at these sizes above)
    rng = np.random.default_rng(5)
    L = kx.lib()
    sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
    T, M, H = 128, 4096, 14336
    _, w, s = rand_w8(rng, 2 * H, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    dx, dw, dsc = kx.Buf.from_numpy(X16), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)
    plain = kx.gemm_w8_rows(X16.astype(np.float32), w, s, GS, split_ws=sp, half=True)
    g, u = plain[:, :H], plain[:, H:]
    with np.errstate(over="ignore"):
        want = g * (np.float32(1.0) / (np.float32(1.0) + np.exp(-g))) * u
    out = kx.Buf.zeros((T, H), np.float16)
    kx.check(L.kx_gemm_w8_rows_swiglu_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, out.ptr, H, T, M, H, sp.ptr, sp.nbytes,
                                             None))
    kx.sync()
    np.testing.assert_allclose(out.numpy(np.float16, (T, H)).astype(np.float32), want, rtol=2.0 ** -10, atol=1e-6)
    del dw, dsc, w, s
    hs, ql, kl, pos0 = 128, 4096, 1024, 4000
    K = ql + 2 * kl
    _, w, s = rand_w8(rng, K, M)
    bias = (rng.standard_normal(K) * 0.1).astype(np.float32)
    S = pos0 + T
    ds_, dc_ = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    kx.check(L.kx_sin_cos_cache_f32(hs, S, 500000.0, ds_.ptr, dc_.ptr, None))
    s_tab, c_tab = ds_.numpy(), dc_.numpy()
    plain = kx.gemm_w8_rows(X16.astype(np.float32), w, s, GS, split_ws=sp, half=True) + bias
    dw, dsc, db = kx.Buf.from_numpy(w), kx.Buf.from_numpy(s), kx.Buf.from_numpy(bias)
    q, kc, vc = kx.Buf.zeros((T, ql)), kx.Buf.zeros((T, kl)), kx.Buf.zeros((T, kl))
    kx.check(L.kx_gemm_w8_rows_qkv_rope_f16act(dx.ptr, M, dw.ptr, dsc.ptr, GS, q.ptr, ql, kc.ptr, kl, vc.ptr, kl, ql,
                                               kl, hs, 1, ds_.ptr, dc_.ptr, pos0, db.ptr, T, M, sp.ptr, sp.nbytes,
                                               None))
    kx.sync()
    gq, gk, gv = q.numpy(), kc.numpy(), vc.numpy()
    tol = 2e-5 * np.abs(plain).max()
    for r in (0, 1, 17, 64, 127):
        wq, wk = kxo.rope(1, ql, kl, hs, plain[r, :ql].copy(), plain[r, ql:ql + kl].copy(), pos0 + r, s_tab, c_tab)
        np.testing.assert_allclose(gq[r], wq, rtol=1e-5, atol=tol)
        np.testing.assert_allclose(gk[r], wk, rtol=1e-5, atol=tol)
        np.testing.assert_allclose(gv[r], plain[r, ql + kl:], rtol=1e-6, atol=tol)
    assert not sp.numpy()[:4096].any()


def test_swiglu_rows_fp16_result():
    rng = np.random.default_rng(3)
    T, half = 37, 1024
    H = (rng.standard_normal((T, 2 * half)) * 3).astype(np.float32)
    H[0, 0], H[0, half] = 300.0, 400.0                    # 120000 in fp32: saturates, no infinity
    din, dout = kx.Buf.from_numpy(H), kx.Buf.zeros((T, half), np.float16)
    kx.check(kx.lib().kx_swiglu_rows_f16out(din.ptr, 2 * half, half, dout.ptr, half, T, None), "swiglu16")
    kx.sync()
    g, u = H[:, :half].astype(np.float64), H[:, half:].astype(np.float64)
    ref = g / (1.0 + np.exp(-g)) * u
    got = dout.numpy().astype(np.float64)
    assert got[0, 0] == 65504.0
    ref[0, 0] = 65504.0
    assert np.all(np.abs(got - ref) <= 2.0 ** -10 * np.abs(ref) + 1e-6)


@pytest.mark.parametrize("T,K,M", [(512, 128, 128), (533, 200, 256), (700, 384, 1024), (1100, 200, 128)])
def test_gemm_w8_two_pass_vs_oracle(T, K, M):
    # long-prompt form: X and scale*w each rounded once to fp16 (2^-11 relative, per element)
    rng = np.random.default_rng(T + K + M)
    _, w, s = rand_w8(rng, K, M)
    s = s.reshape(K, M // GS)
    s[1] *= 1e-4          # a row of tiny scales: the per-row power-of-two factor keeps it normal
    s[2, 1:] *= 1e-3      # small groups beside a large one in the same row
    w[3] = 0              # an all-zero row
    s = s.reshape(-1)
    X = rng.standard_normal((T, M)).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    got = kx.gemm_w8(X, w, s, GS, two_pass=True)
    sd = np.repeat(s.astype(np.float64), GS).reshape(K, M) * np.abs(w.astype(np.float64))
    absum = np.abs(X.astype(np.float64)) @ sd.T
    err = np.abs(got - want)
    assert np.all(err <= 2.0 ** -10 * absum + 1e-7 * np.abs(want).max())
    assert np.sqrt((err ** 2).mean()) <= 2e-4 * np.sqrt((absum ** 2).mean())
    assert np.all(got[:, 3] == 0)


@pytest.mark.parametrize("T,K,M", [(256, 256, 256), (300, 448, 128), (1100, 512, 192)])
def test_gemm_w8_f16_activations_and_fused_swiglu(T, K, M):
    # fp16 rows in; fp32 rows out, or silu(gate)*up in fp16 from the [gate | up] matrix
    rng = np.random.default_rng(T * 3 + K)
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    X = X16.astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    absum = np.abs(X.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(K, M) *
                                             np.abs(w.astype(np.float64))).T
    tol = 2.0 ** -11 * absum + 1e-7 * np.abs(want).max()   # only scale*w is rounded here
    got = kx.gemm_w8_f16act(X16, w, s, GS)
    assert np.all(np.abs(got - want) <= tol)
    h = kx.gemm_w8_f16act(X16, w, s, GS, swiglu=True).astype(np.float64)
    g, u = want[:, :K // 2].astype(np.float64), want[:, K // 2:].astype(np.float64)
    ref = g / (1.0 + np.exp(-g)) * u
    tg, tu = tol[:, :K // 2], tol[:, K // 2:]
    # d silu/dg is within [-0.1, 1.1]; plus the fp16 rounding of the result
    bound = 1.1 * tg * np.abs(u) + tu * np.abs(g) + tg * tu + 2.0 ** -10 * np.abs(ref) + 1e-6
    assert h.shape == (T, K // 2)
    assert np.all(np.abs(h - ref) <= bound)


@pytest.mark.parametrize("choice", ["1", "3"])
@pytest.mark.parametrize("T,K,M", [(1100, 600, 192), (520, 256, 64), (1024, 512, 1024)])
def test_gemm_w8_f16_every_tile_kernel(monkeypatch, choice, T, K, M):
    # the 128 x 128 and the pipelined 256 x 256 kernel on ragged tiles, 1 / 3 / 16 K-tiles
    monkeypatch.setenv("KX_GEMM_F16_KERNEL", choice)
    rng = np.random.default_rng(int(choice) * 7 + T)
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    want = kxo.gemm_w8(X16.astype(np.float32), w, s, GS, threads=8)
    absum = np.abs(X16.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(K, M) *
                                               np.abs(w.astype(np.float64))).T
    got = kx.gemm_w8_f16act(X16, w, s, GS)
    assert np.all(np.abs(got - want) <= 2.0 ** -11 * absum + 1e-7 * np.abs(want).max())


@pytest.mark.parametrize("T,K,M", [(520, 600, 1024), (300, 256, 512), (4352, 4096, 512), (40, 512, 2048)])
def test_gemm_w8_f16_reduction_cut_over_idle_cus(monkeypatch, T, K, M):
    # a last, partial round of 256 x 256 tiles has its reduction cut into 2 or 4 parts that meet
    # in fp32 slabs: same result as the uncut kernel up to fp32 re-association, bitwise repeatable,
    # tickets back at zero; (4352, 4096): 16 cut tiles behind 256 whole ones
    monkeypatch.setenv("KX_GEMM_F16_KERNEL", "3")
    rng = np.random.default_rng(T + M)
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    absum = np.abs(X16.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(K, M) *
                                               np.abs(w.astype(np.float64))).T
    monkeypatch.setenv("KX_GEMM_SPLIT", "1")
    base = kx.gemm_w8_f16act(X16, w, s, GS)
    rows = np.unique(np.r_[0, T - 1, rng.integers(0, T, 6)])
    want = kxo.gemm_w8(X16[rows].astype(np.float32), w, s, GS, threads=8)
    assert np.all(np.abs(base[rows] - want) <= 2.0 ** -11 * absum[rows] + 1e-7 * np.abs(want).max())
    sp = kx.Buf.zeros(kx.lib().kx_gemm_split_workspace_bytes(), np.uint8)
    for parts in ("2", "4", "8"):
        if M // 64 < 2 * int(parts):
            continue                                 # too few K-tiles to cut this finely
        monkeypatch.setenv("KX_GEMM_SPLIT", parts)
        got = kx.gemm_w8_f16act(X16, w, s, GS, split_ws=sp)
        assert np.all(np.abs(got - base) <= 1e-6 * absum + 1e-7 * np.abs(base).max())
        assert not np.array_equal(got, base)   # the cut really happened
        again = kx.gemm_w8_f16act(X16, w, s, GS, split_ws=sp)
        assert np.array_equal(got, again)
        assert not sp.numpy(np.uint8, (4096,)).any()
        if K % 64 == 0:
            h = kx.gemm_w8_f16act(X16, w, s, GS, swiglu=True, split_ws=sp).astype(np.float32)
            monkeypatch.setenv("KX_GEMM_SPLIT", "1")
            h1 = kx.gemm_w8_f16act(X16, w, s, GS, swiglu=True).astype(np.float32)
            assert np.all(np.abs(h - h1) <= 2.0 ** -9 * np.abs(h1) + 1e-5)
    # no split workspace: the request is ignored, the plain kernel runs
    monkeypatch.setenv("KX_GEMM_SPLIT", "2")
    assert np.array_equal(kx.gemm_w8_f16act(X16, w, s, GS), base)


@pytest.mark.parametrize("hs,variant,T,pos0,with_bias", [(128, 1, 300, 5, False), (64, 1, 530, 0, True),
                                                         (64, 0, 260, 17, True), (128, 0, 256, 3, False)])
def test_gemm_w8_f16_qkv_with_bias_and_rope_in_the_epilogue(hs, variant, T, pos0, with_bias):
    # one launch = product + bias + rotary embedding + q / K-row / V-row placement; checked against
    # the plain product (same kernel, plain epilogue) followed by the oracle's bias-then-rope
    rng = np.random.default_rng(hs + T)
    L = kx.lib()
    M, H, KVH = 256, 4, 2
    ql, kl = H * hs, KVH * hs
    K = ql + 2 * kl
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    bias = rng.standard_normal(K).astype(np.float32) * 0.1 if with_bias else None
    S = pos0 + T + 3
    ds_, dc_ = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    kx.check(L.kx_sin_cos_cache_f32(hs, S, 10000.0, ds_.ptr, dc_.ptr, None))
    s_tab, c_tab = ds_.numpy(), dc_.numpy()
    plain = kx.gemm_w8_f16act(X16, w, s, GS)
    if with_bias:
        plain = plain + bias
    dx, dw, dsc = kx.Buf.from_numpy(X16), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s)
    q = kx.Buf.zeros((T, ql))
    kst = kl + 64                                       # K / V rows live in a wider cache
    kc, vc = kx.Buf.from_numpy(np.full((S, kst), 7.0, np.float32)), kx.Buf.from_numpy(np.full((S, kst), 9.0, np.float32))
    db = kx.Buf.from_numpy(bias) if with_bias else None
    wsb = L.kx_gemm_w8_f16act_workspace_bytes(M, K)
    ws = kx.Buf(wsb)
    sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
    kx.check(L.kx_gemm_w8_f16act_qkv_rope(dx.ptr, dw.ptr, dsc.ptr, GS, 0, q.ptr, ql, kc.at(pos0 * kst * 4), kst,
                                          vc.at(pos0 * kst * 4), kst, ql, kl, hs, variant, ds_.ptr, dc_.ptr, pos0,
                                          db.ptr if db else None, T, M, ws.ptr, wsb, sp.ptr, sp.nbytes, None))
    gq, gk, gv = q.numpy(), kc.numpy(), vc.numpy()
    tol = 2e-5 * np.abs(plain).max()
    for r in range(T):
        wq, wk = kxo.rope(variant, ql, kl, hs, plain[r, :ql].copy(), plain[r, ql:ql + kl].copy(), pos0 + r,
                          s_tab, c_tab)
        np.testing.assert_allclose(gq[r], wq, rtol=1e-5, atol=tol)
        np.testing.assert_allclose(gk[pos0 + r, :kl], wk, rtol=1e-5, atol=tol)
        np.testing.assert_allclose(gv[pos0 + r, :kl], plain[r, ql + kl:], rtol=1e-6, atol=tol)
    # rows outside the chunk and the padding beside the rows are untouched
    assert np.all(gk[:pos0] == 7.0) and np.all(gk[pos0 + T:] == 7.0) and np.all(gk[:, kl:] == 7.0)
    assert np.all(gv[:pos0] == 9.0) and np.all(gv[pos0 + T:] == 9.0) and np.all(gv[:, kl:] == 9.0)
    # head sizes the epilogue is not laid out for are refused, not mangled
    assert L.kx_gemm_w8_f16act_qkv_rope(dx.ptr, dw.ptr, dsc.ptr, GS, 0, q.ptr, ql, kc.ptr, kst, vc.ptr, kst, ql, kl,
                                        32, variant, ds_.ptr, dc_.ptr, 0, None, T, M, ws.ptr, wsb, None, 0,
                                        None) == -2


@pytest.mark.parametrize("hs,variant,T,pos0,parts", [(128, 1, 300, 5, "4"), (64, 1, 530, 0, "8"), (64, 0, 260, 17, "4"),
                                                     (128, 0, 256, 3, "16"), (128, 1, 40, 2, "8")])
def test_gemm_w8_f16_cut_reduction_summed_by_a_launch_of_its_own(monkeypatch, hs, variant, T, pos0, parts):
    # four parts or more: the slabs are summed by a second launch spread over the chip, in part
    # order like the last arriver's sum, so both give the same bits -- plain, residual and the
    # Wq|Wk|Wv form (bias, rotary embedding, K / V rows in a wider cache)
    monkeypatch.setenv("KX_GEMM_F16_KERNEL", "3")
    monkeypatch.setenv("KX_GEMM_SPLIT", parts)
    rng = np.random.default_rng(hs + T)
    L = kx.lib()
    M, H, KVH = 2048, 4, 2
    ql, kl = H * hs, KVH * hs
    K = ql + 2 * kl
    _, w, s = rand_w8(rng, K, M)
    X16 = rng.standard_normal((T, M)).astype(np.float16)
    bias = rng.standard_normal(K).astype(np.float32) * 0.1
    res = rng.standard_normal((T, K)).astype(np.float32)
    S = pos0 + T + 3
    ds_, dc_ = kx.Buf.zeros((S, hs)), kx.Buf.zeros((S, hs))
    kx.check(L.kx_sin_cos_cache_f32(hs, S, 10000.0, ds_.ptr, dc_.ptr, None))
    dx, dw, dsc, db = kx.Buf.from_numpy(X16), kx.Buf.from_numpy(w), kx.Buf.from_numpy(s), kx.Buf.from_numpy(bias)
    wsb = L.kx_gemm_w8_f16act_workspace_bytes(M, K)
    ws = kx.Buf(wsb)
    sp = kx.Buf.zeros(L.kx_gemm_split_workspace_bytes(), np.uint8)
    kst = kl + 64

    def run():
        plain = kx.gemm_w8_f16act(X16, w, s, GS, split_ws=sp)
        resid = kx.gemm_w8_f16act(X16, w, s, GS, split_ws=sp, residual=res)
        q = kx.Buf.zeros((T, ql))
        kc = kx.Buf.from_numpy(np.full((S, kst), 7.0, np.float32))
        vc = kx.Buf.from_numpy(np.full((S, kst), 9.0, np.float32))
        kx.check(L.kx_gemm_w8_f16act_qkv_rope(dx.ptr, dw.ptr, dsc.ptr, GS, 0, q.ptr, ql, kc.at(pos0 * kst * 4), kst,
                                              vc.at(pos0 * kst * 4), kst, ql, kl, hs, variant, ds_.ptr, dc_.ptr,
                                              pos0, db.ptr, T, M, ws.ptr, wsb, sp.ptr, sp.nbytes, None))
        kx.sync()
        assert not sp.numpy(np.uint8, (4096,)).any()        # tickets back at zero
        return plain, resid, q.numpy(), kc.numpy(), vc.numpy()

    later = run()
    monkeypatch.setenv("KX_GEMM_COMBINE", "0")
    arriver = run()
    for a, b in zip(later, arriver):
        assert np.array_equal(a, b)
    monkeypatch.setenv("KX_GEMM_SPLIT", "1")
    whole = run()
    assert not np.array_equal(whole[0], later[0])            # the cut really happened
    scale = np.abs(whole[0]).max()
    for a, b in zip(later, whole):
        assert np.all(np.abs(a - b) <= 2e-5 * scale)
    plain, _, gq, gk, gv = later
    pb = plain + bias
    for r in np.unique(np.r_[0, T - 1, rng.integers(0, T, 6)]):
        wq, wk = kxo.rope(variant, ql, kl, hs, pb[r, :ql].copy(), pb[r, ql:ql + kl].copy(), pos0 + r,
                          ds_.numpy(), dc_.numpy())
        np.testing.assert_allclose(gq[r], wq, rtol=1e-5, atol=2e-5 * scale)
        np.testing.assert_allclose(gk[pos0 + r, :kl], wk, rtol=1e-5, atol=2e-5 * scale)
        np.testing.assert_allclose(gv[pos0 + r, :kl], pb[r, ql + kl:], rtol=1e-6, atol=2e-5 * scale)
    assert np.all(gk[:pos0] == 7.0) and np.all(gk[pos0 + T:] == 7.0) and np.all(gk[:, kl:] == 7.0)
    assert np.all(gv[:pos0] == 9.0) and np.all(gv[pos0 + T:] == 9.0) and np.all(gv[:, kl:] == 9.0)


def test_fp16_activations_saturate_instead_of_overflowing():
    # silu(g) * u and the normalised rows leave as fp16: a value past 65504 must stay finite
    T, K, M = 256, 128, 64
    w = np.full((K, M), 127, np.int8)
    s = np.ones(K * M // GS, np.float32)
    X16 = np.ones((T, M), np.float16)
    h = kx.gemm_w8_f16act(X16, w, s, GS, swiglu=True)       # gate = up = 8128 -> 6.6e7
    assert np.all(h == np.float16(65504.0))
    x = np.full((4, 128), 3.0, np.float32)
    out = kx.rmsnorm_f16out(x, np.full(128, 1e6, np.float32))
    assert np.all(out == np.float16(65504.0))
    out = kx.rmsnorm_f16out(-x, np.full(128, 1e6, np.float32))
    assert np.all(out == np.float16(-65504.0))


def test_gemm_w8_f16_activations_preconditions():
    L = kx.lib()
    T, K, M = 256, 128, 128
    x, w, s = kx.Buf.zeros((T, M), np.float16), kx.Buf.zeros((K, M), np.int8), kx.Buf.zeros(K * M // GS)
    y, h = kx.Buf.zeros((T, K)), kx.Buf.zeros((T, K // 2), np.float16)
    wsb = L.kx_gemm_w8_f16act_workspace_bytes(M, K)
    assert wsb >= 2 * K * M + 4 * K
    ws = kx.Buf(wsb)
    call = lambda mode, hp, yp, k, wb, res=None: L.kx_gemm_w8_f16act_ws(
        x.ptr, w.ptr, s.ptr, GS, mode, hp, yp, K, None, 0, None, 0, K, K, res, 0, T, M, k, ws.ptr, wb,
        None)
    assert call(0, None, y.ptr, K, wsb) == 0
    assert call(1, h.ptr, None, K, wsb) == 0
    assert call(0, None, None, K, wsb) == -1          # no fp32 output
    assert call(1, None, y.ptr, K, wsb) == -1         # no fp16 output
    assert call(1, h.ptr, None, 96, wsb) == -1        # not whole 32-unit blocks
    assert call(1, h.ptr, None, K, wsb, y.ptr) == -1  # no residual in the SwiGLU form
    assert call(2, h.ptr, y.ptr, K, wsb) == -1
    assert call(0, None, y.ptr, K, wsb - 1) == -3
    kx.sync()


def test_rmsnorm_f16out_vs_oracle():
    rng = np.random.default_rng(5)
    for rows, dim in ((3, 256), (260, 4096), (17, 896)):
        x = rng.standard_normal((rows, dim)).astype(np.float32) * 3
        g = rng.standard_normal(dim).astype(np.float32)
        want = np.stack([kxo.rmsnorm(r, g, 1e-5) for r in x])
        got = kx.rmsnorm_f16out(x, g, 1e-5)
        assert got.dtype == np.float16
        np.testing.assert_allclose(got.astype(np.float32), want, rtol=2.0 ** -10, atol=1e-6)


def test_gemm_w8_two_pass_routing_residual_and_fallback():
    rng = np.random.default_rng(31)
    T, dim, kvd, S = 530, 256, 128, 560
    K = dim + 2 * kvd
    _, w, s = rand_w8(rng, K, dim)
    X = rng.standard_normal((T, dim)).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    dx, dw, ds = map(kx.Buf.from_numpy, (X, w, s))
    L = kx.lib()
    wsb = L.kx_gemm_w8_workspace_bytes(T, dim, K)
    assert wsb >= 2 * T * dim + 2 * K * dim + 4 * K
    ws = kx.Buf(wsb)
    q = kx.Buf.zeros((T, dim))
    kc, vc = kx.Buf.zeros((S, kvd)), kx.Buf.zeros((S, kvd))
    pos0 = 7
    kx.check(L.kx_gemm_w8_f32_ws(dx.ptr, dim, dw.ptr, ds.ptr, GS, q.ptr, dim, kc.at(pos0 * kvd * 4),
                                 kvd, vc.at(pos0 * kvd * 4), kvd, dim, dim + kvd, None, 0, T, dim, K,
                                 ws.ptr, wsb, None))
    tol = 2.0 ** -10 * (np.abs(X.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(
        K, dim) * np.abs(w.astype(np.float64))).T) + 1e-7
    assert np.all(np.abs(q.numpy() - want[:, :dim]) <= tol[:, :dim])
    kcn, vcn = kc.numpy(), vc.numpy()
    assert np.all(np.abs(kcn[pos0:pos0 + T] - want[:, dim:dim + kvd]) <= tol[:, dim:dim + kvd])
    assert np.all(np.abs(vcn[pos0:pos0 + T] - want[:, dim + kvd:]) <= tol[:, dim + kvd:])
    assert np.all(kcn[:pos0] == 0) and np.all(kcn[pos0 + T:] == 0)
    R = rng.standard_normal((T, dim)).astype(np.float32)
    y = kx.Buf.from_numpy(R)
    kx.check(L.kx_gemm_w8_f32_ws(dx.ptr, dim, dw.ptr, ds.ptr, GS, y.ptr, dim, None, 0, None, 0, dim,
                                 dim, y.ptr, dim, T, dim, dim, ws.ptr, wsb, None))
    assert np.all(np.abs(y.numpy() - (want[:, :dim] + R)) <= tol[:, :dim] + 1e-6)
    # no workspace (or a short prompt): same entry point, single-pass kernel
    y2 = kx.Buf.zeros((T, dim))
    kx.check(L.kx_gemm_w8_f32_ws(dx.ptr, dim, dw.ptr, ds.ptr, GS, y2.ptr, dim, None, 0, None, 0, dim,
                                 dim, None, 0, T, dim, dim, None, 0, None))
    assert np.all(np.abs(y2.numpy() - want[:, :dim]) <= 2.0 ** -11 * tol[:, :dim] * 2 ** 10 + 1e-7)


def test_gemm_w8_small_T_and_odd_group_fall_back_to_gemv_rows():
    rng = np.random.default_rng(2)
    x, w, s = rand_w8(rng, 96, 192)
    X = rng.standard_normal((3, 192)).astype(np.float32)
    got = kx.gemm_w8(X, w, s, GS)  # T < 4: row-wise GEMV, fp32 exact path
    want = kxo.gemm_w8(X, w, s, GS)
    for t in range(3):
        assert np.all(np.abs(got[t] - want[t]) <= 1e-5 * w8_abs_sum(X[t], w, s))


def test_gemm_w8_ex_routing_and_residual():
    rng = np.random.default_rng(3)
    T, dim, kvd, S = 70, 256, 128, 96
    K = dim + 2 * kvd
    _, w, s = rand_w8(rng, K, dim)
    X = rng.standard_normal((T, dim)).astype(np.float32)
    want = kxo.gemm_w8(X, w, s, GS, threads=8)
    dx, dw, ds = map(kx.Buf.from_numpy, (X, w, s))
    q = kx.Buf.zeros((T, dim))
    kc, vc = kx.Buf.zeros((S, kvd)), kx.Buf.zeros((S, kvd))
    pos0 = 11  # k/v rows land in cache rows pos0 .. pos0+T-1
    kx.check(kx.lib().kx_gemm_w8_f32_ex(dx.ptr, dim, dw.ptr, ds.ptr, GS, q.ptr, dim,
                                        kc.at(pos0 * kvd * 4), kvd, vc.at(pos0 * kvd * 4), kvd, dim,
                                        dim + kvd, None, 0, T, dim, K, None))
    tol = 2.0 ** -11 * (np.abs(X.astype(np.float64)) @ (np.repeat(s.astype(np.float64), GS).reshape(
        K, dim) * np.abs(w.astype(np.float64))).T) + 1e-7
    assert np.all(np.abs(q.numpy() - want[:, :dim]) <= tol[:, :dim])
    kcn, vcn = kc.numpy(), vc.numpy()
    assert np.all(np.abs(kcn[pos0:pos0 + T] - want[:, dim:dim + kvd]) <= tol[:, dim:dim + kvd])
    assert np.all(np.abs(vcn[pos0:pos0 + T] - want[:, dim + kvd:]) <= tol[:, dim + kvd:])
    assert np.all(kcn[:pos0] == 0) and np.all(kcn[pos0 + T:] == 0)
    # residual in place
    R = rng.standard_normal((T, dim)).astype(np.float32)
    y = kx.Buf.from_numpy(R)
    kx.check(kx.lib().kx_gemm_w8_f32_ex(dx.ptr, dim, dw.ptr, ds.ptr, GS, y.ptr, dim, None, 0, None,
                                        0, dim, dim, y.ptr, dim, T, dim, dim, None))
    assert np.all(np.abs(y.numpy() - (want[:, :dim] + R)) <= tol[:, :dim] + 1e-6)


# ------------------------------- prefill: MFMA flash attention -------------------------------------
# Q*scale, K, V and the probabilities are rounded to fp16 for the matrix instructions
# (softmax, accumulation and the output are fp32): stated tolerance 3e-3 absolute on N(0,1) data.
def test_flash_attention_golden(golden):
    g = golden("g5_attention.npz")
    ran = 0
    for name, meta, q, kc, vc, want in _attn_cases(g):
        H, KVH, hs, S, pos, Tq, L, layer = [int(v) for v in meta[:8]]
        if Tq == 1:
            continue
        got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, layer)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-3, err_msg=name)
        ran += 1
    assert ran >= 4  # includes the reference test's own inputs (head size 8: row-wise path)


@pytest.mark.parametrize("H,KVH,hs,S,pos,Tq,causal", [
    (8, 2, 64, 300, 255, 256, True), (8, 2, 128, 300, 299, 130, True), (4, 4, 64, 200, 150, 33, True),
    (14, 2, 128, 160, 159, 160, True), (16, 2, 128, 128, 100, 64, True), (6, 2, 64, 100, 99, 50, False),
    (4, 2, 32, 70, 69, 70, True),
    # caches of more than three tiles take the LDS-DMA form when fp16 copies exist: workgroups of 7,
    # 6 and 5-as-4 waves (only waves 0..3 issue the pieces), 8 waves from 2 heads x 4 sub-blocks,
    # 32-wide heads, one head per kv head, a non-causal pass, rows that end inside a tile
    (14, 2, 128, 280, 279, 200, True), (3, 1, 128, 330, 329, 150, True), (5, 1, 64, 260, 259, 100, True),
    (4, 2, 128, 300, 299, 300, True), (4, 2, 32, 400, 399, 333, True), (2, 2, 64, 270, 269, 77, True),
    (8, 1, 128, 256, 230, 40, False)])
def test_flash_attention_vs_oracle(H, KVH, hs, S, pos, Tq, causal):
    rng = np.random.default_rng(H * 31 + Tq)
    L, layer = 2, 1
    kc = rng.standard_normal((L, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((L, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal((Tq, H, hs)).astype(np.float32)
    want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, layer, is_causal=causal, threads=8)
    # both staging forms: fp16 copies of K/V made once in the workspace, and per-tile conversion
    for f16_copies in (True, False):
        got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, layer, is_causal=causal,
                                 f16_copies=f16_copies)
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
        assert np.sqrt(((got - want) ** 2).mean()) <= 4e-4
    # fp16 output rows: the same numbers rounded once more
    got16 = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, layer, is_causal=causal, f16_out=True)
    assert got16.dtype == np.float16
    np.testing.assert_allclose(got16.astype(np.float32), got, rtol=2.0 ** -10, atol=1e-6)


def test_flash_attention_nan_past_pos_is_never_read():
    # cache rows after `pos` are poison: neither the fp16 pre-pass nor the tiles may touch them
    rng = np.random.default_rng(77)
    H, KVH, hs, S, pos, Tq = 8, 2, 128, 256, 200, 140
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal((Tq, H, hs)).astype(np.float32)
    want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, 0, threads=8)
    kc[0, pos + 1:] = np.nan
    vc[0, pos + 1:] = np.nan
    for f16_copies in (True, False):
        got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0, f16_copies=f16_copies)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)


def test_flash_attention_forced_rescale_and_last_row_equals_decode():
    rng = np.random.default_rng(8)
    H, KVH, hs, S, pos, Tq = 8, 2, 128, 320, 319, 320
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32) * 0.2
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal((Tq, H, hs)).astype(np.float32)
    kc[0, 200, :hs] = q[250, 0] * 2.0  # late spike: the running max jumps at key tile 3
    want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, 0, threads=8)
    got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
    last = kx.mha_decode(q[-1].reshape(-1), kc, vc, pos, H, H // KVH, hs, 0)
    np.testing.assert_allclose(got[-1], last, rtol=0, atol=3e-3)


@pytest.mark.parametrize("hs", [128, 64, 32])
def test_flash_attention_one_dominant_key_per_tile(hs):
    # Every 64-key tile holds one key whose score beats everything before it by ~17 (natural log
    # units), at a different position inside each tile.  The tile's running maximum has to be the
    # maximum over ALL of the tile's keys, whichever lanes hold them: taken over a part of them, the
    # probabilities of the rest leave fp16's range on their way into the PV product.
    rng = np.random.default_rng(hs)
    H, KVH, S = 4, 2, 512
    pos, Tq = S - 1, S
    u = rng.standard_normal(hs).astype(np.float32)
    u *= np.sqrt(hs) / np.linalg.norm(u)
    q = (u + 0.05 * rng.standard_normal((Tq, H, hs))).astype(np.float32)
    kc = (0.02 * rng.standard_normal((1, S, KVH * hs))).astype(np.float32)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float32)
    for t in range(S // 64):
        key = 64 * t + (5 * t + 2) % 64
        kc[0, key] = np.tile(u, KVH) * (1.5 * (t + 1) * np.sqrt(128.0 / hs))
    want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, 0, threads=8)
    for f16_copies in (True, False):
        got = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0, f16_copies=f16_copies)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
    got16 = kx.flash_attention(q, kc, vc, H, KVH, hs, pos, 0, f16_out=True)
    np.testing.assert_allclose(got16.astype(np.float32), want, rtol=2.0 ** -10, atol=3e-3)
    # a 16-bit cache: same data rounded once
    k16, v16 = kc.astype(np.float16), vc.astype(np.float16)
    want16 = kxo.flash_attention(q, k16.astype(np.float32), v16.astype(np.float32), H, KVH, hs, S, pos, 0,
                                 threads=8)
    got = kx.flash_attention(q, k16, v16, H, KVH, hs, pos, 0, kv16=True)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want16, rtol=0, atol=3e-3)
    # the decode kernels on the same cache, at positions that end inside different tiles
    for p_ in (70, 200, 330, pos):
        one = kx.mha_decode(q[p_].reshape(-1), kc, vc, p_, H, H // KVH, hs, 0)
        np.testing.assert_allclose(one, want[p_], rtol=0, atol=3e-3)
        one16 = kx.mha_decode(q[p_].reshape(-1), k16, v16, p_, H, H // KVH, hs, 0, kv16=True)
        np.testing.assert_allclose(one16, want16[p_], rtol=0, atol=3e-3)


# ---------------- 16-bit KV cache ---------------------------------------------------------------------
@pytest.mark.parametrize("H,KVH,hs,S,pos", [(32, 8, 128, 600, 599), (8, 2, 64, 100, 37), (4, 4, 32, 40, 0),
                                            (28, 4, 128, 1100, 1099)])
def test_mha_decode_kv16_equals_fp32_path_on_rounded_cache(H, KVH, hs, S, pos):
    # K and V are widened exactly, so a half cache must give what the fp32 kernel gives on the
    # same values; and it stays within fp16 rounding of the unrounded cache (oracle)
    rng = np.random.default_rng(pos + H)
    kc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    k16, v16 = kc.astype(np.float16), vc.astype(np.float16)
    got = kx.mha_decode(q, k16, v16, pos, H, H // KVH, hs, 1, kv16=True)
    same = kx.mha_decode(q, k16.astype(np.float32), v16.astype(np.float32), pos, H, H // KVH, hs, 1)
    np.testing.assert_allclose(got, same, rtol=0, atol=1e-6)
    want = kxo.mha(pos, H, 1, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
    np.testing.assert_allclose(got, want, rtol=0, atol=4e-3)


def test_mha_decode_kv16_device_position_and_preconditions():
    L = kx.lib()
    rng = np.random.default_rng(2)
    H, KVH, hs, S = 8, 2, 128, 300
    kc = rng.standard_normal((1, S, KVH * hs)).astype(np.float16)
    vc = rng.standard_normal((1, S, KVH * hs)).astype(np.float16)
    q = rng.standard_normal(H * hs).astype(np.float32)
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(kc), kx.Buf.from_numpy(vc)
    out = kx.Buf.zeros(H * hs)
    wsb = L.kx_attn_workspace_bytes(H, hs)
    ws = kx.Buf.zeros(wsb, np.uint8)
    for pos in (0, 17, 299):
        pd = kx.Buf.from_numpy(np.array([pos], np.int32))
        kx.check(L.kx_mha_decode_kv16(dq.ptr, dk.ptr, dv.ptr, out.ptr, S - 1, pd.ptr, H, H // KVH, hs,
                                      KVH * hs, S, 0, ws.ptr, wsb, None))
        want = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 0, kv16=True)
        np.testing.assert_allclose(out.numpy(np.float32, (H, hs)), want, rtol=0, atol=1e-6)
    # head size the tiled kernel is not built for: the generic fp32 kernel does not read half rows
    assert L.kx_mha_decode_kv16(dq.ptr, dk.ptr, dv.ptr, out.ptr, 3, None, 8, 4, 16, 32, 64, 0, ws.ptr,
                                wsb, None) == -2
    assert L.kx_mha_decode_kv16(dq.ptr, None, dv.ptr, out.ptr, 3, None, H, H // KVH, hs, KVH * hs, S, 0,
                                ws.ptr, wsb, None) == -1


@pytest.mark.parametrize("H,KVH,hs,S,pos,Tq", [(8, 2, 128, 300, 299, 300), (8, 2, 64, 200, 150, 33),
                                               (4, 2, 32, 70, 69, 70)])
def test_flash_attention_kv16(H, KVH, hs, S, pos, Tq):
    rng = np.random.default_rng(Tq)
    kc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal((Tq, H, hs)).astype(np.float32)
    k16, v16 = kc.astype(np.float16), vc.astype(np.float16)
    got = kx.flash_attention(q, k16, v16, H, KVH, hs, pos, 1, kv16=True)
    # the prefill kernel rounds K and V to fp16 anyway: identical to the fp32-cache path on the
    # rounded values
    same = kx.flash_attention(q, k16.astype(np.float32), v16.astype(np.float32), H, KVH, hs, pos, 1)
    np.testing.assert_array_equal(got, same)
    want = kxo.flash_attention(q, kc, vc, H, KVH, hs, S, pos, 1, threads=8)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
    got16 = kx.flash_attention(q, k16, v16, H, KVH, hs, pos, 1, kv16=True, f16_out=True)
    np.testing.assert_allclose(got16.astype(np.float32), got, rtol=2.0 ** -10, atol=1e-6)
    # without the workspace for the per-head copies the half cache cannot be read
    L = kx.lib()
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(k16), kx.Buf.from_numpy(v16)
    out = kx.Buf.zeros(q.size)
    ws = kx.Buf.zeros(32768, np.uint8)
    assert L.kx_flash_attention_kv16(dq.ptr, dk.ptr, dv.ptr, out.ptr, None, Tq, H, KVH, hs, S, pos, 1,
                                     float(1 / np.sqrt(hs)), 1, ws.ptr, 20000, None) == -3
    assert L.kx_flash_attention_kv16(dq.ptr, dk.ptr, dv.ptr, None, None, Tq, H, KVH, hs, S, pos, 1,
                                     float(1 / np.sqrt(hs)), 1, ws.ptr, 32768, None) == -1


def test_f32_to_f16_rows_strided():
    rng = np.random.default_rng(0)
    rows, cols, lds, ldd = 37, 64, 80, 96
    src = rng.standard_normal((rows, lds)).astype(np.float32)
    d_src = kx.Buf.from_numpy(src)
    d_dst = kx.Buf.from_numpy(np.full((rows, ldd), 7, np.float16))
    kx.check(kx.lib().kx_f32_to_f16_rows(d_src.ptr, lds, d_dst.ptr, ldd, cols, rows, None))
    got = d_dst.numpy(np.float16, (rows, ldd))
    np.testing.assert_array_equal(got[:, :cols], src[:, :cols].astype(np.float16))
    assert np.all(got[:, cols:] == 7)
    assert kx.lib().kx_f32_to_f16_rows(d_src.ptr, lds, d_dst.ptr, ldd, 62, rows, None) == -1


# ---------------- chained GEMVs in one persistent launch ---------------------------------------------
def _chain_case(rng, dim, hid, H, KVH, hs, variant, pos):
    ql, kl = H * hs, KVH * hs
    S = pos + 3

    def w8(K, M):
        return rng.integers(-127, 128, (K, M), dtype=np.int8), rng.uniform(1e-3, 3e-3, K * M // GS).astype(np.float32)
    m = dict(wo=w8(dim, ql), w13=w8(2 * hid, dim), w2=w8(dim, hid), qkv=w8(ql + 2 * kl, dim))
    v = dict(att=rng.standard_normal(ql).astype(np.float32), x=rng.standard_normal(dim).astype(np.float32),
             g1=rng.uniform(0.5, 1.5, dim).astype(np.float32), g2=rng.uniform(0.5, 1.5, dim).astype(np.float32),
             bias=(rng.standard_normal(ql + 2 * kl) * 0.1).astype(np.float32))
    sin, cos = kxo.sin_cos_cache(hs, S, 1e4)
    return m, v, sin, cos, ql, kl, S


@pytest.mark.parametrize("dim,hid,H,KVH,hs,variant,pos", [
    (1024, 2048, 8, 2, 128, 1, 7), (1024, 3072, 16, 4, 64, 0, 0), (2048, 1024, 16, 16, 128, 1, 33),
    (1024, 1024, 32, 8, 32, 1, 2)])
def test_gemv_chain_vs_oracle(dim, hid, H, KVH, hs, variant, pos):
    # [Wo + residual -> RMSNorm -> W1|W3 + SwiGLU] and [W2 + residual -> RMSNorm -> Wq|Wk|Wv + bias + RoPE +
    # K/V placement], each as ONE launch, against the oracle's formulas step by step; and each single
    # op through the same kernel
    rng = np.random.default_rng(dim + hid + hs + pos)
    m, v, sin, cos, ql, kl, S = _chain_case(rng, dim, hid, H, KVH, hs, variant, pos)
    L = kx.lib()
    dev = {k: (kx.Buf.from_numpy(a), kx.Buf.from_numpy(b)) for k, (a, b) in m.items()}
    att, g1, g2, bias = (kx.Buf.from_numpy(v[k]) for k in ("att", "g1", "g2", "bias"))
    x = kx.Buf.from_numpy(v["x"])
    h, q = kx.Buf.zeros(hid), kx.Buf.zeros(ql)
    kc, vc = kx.Buf.zeros(S * kl), kx.Buf.zeros(S * kl)
    dsin, dcos = kx.Buf.from_numpy(sin), kx.Buf.from_numpy(cos)
    pos_dev = kx.Buf.from_numpy(np.array([pos], np.int32))
    ws = kx.Buf(L.kx_gemv_w8_chain_workspace_bytes(max(dim, hid)))
    kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, None), "reset")
    wo = kx.make_gemv_desc(kx.GEMV_PLAIN, att.ptr, dev["wo"][0].ptr, dev["wo"][1].ptr, x.ptr, ql, dim, residual=x.ptr)
    w13 = kx.make_gemv_desc(kx.GEMV_GATE_UP, None, dev["w13"][0].ptr, dev["w13"][1].ptr, h.ptr, dim, 2 * hid,
                            norm_w=g1.ptr, norm_eps=1e-5)
    w2 = kx.make_gemv_desc(kx.GEMV_PLAIN, h.ptr, dev["w2"][0].ptr, dev["w2"][1].ptr, x.ptr, hid, dim, residual=x.ptr)
    qkv = kx.make_gemv_desc(kx.GEMV_QKV_ROPE, x.ptr, dev["qkv"][0].ptr, dev["qkv"][1].ptr, q.ptr, dim, ql + 2 * kl,
                            norm_w=g2.ptr, norm_eps=1e-5, y1=kc.ptr, y2=vc.ptr, k0=ql, k1=ql + kl, bias=bias.ptr,
                            sin=dsin.ptr, cos=dcos.ptr, pos_dev=pos_dev.ptr, rope_variant=variant, head_size=hs,
                            kv_pos_stride=kl)
    # oracle, in the reference's op order
    x1 = kxo.gemv_w8(v["att"], *m["wo"], GS) + v["x"]
    xn = kxo.rmsnorm(x1, v["g1"], 1e-5)
    h13 = kxo.gemv_w8(xn, *m["w13"], GS)
    hh = kxo.swiglu(h13[:hid], h13[hid:])
    x2 = kxo.gemv_w8(hh, *m["w2"], GS) + x1
    xn2 = kxo.rmsnorm(x2, v["g2"], 1e-5)
    qkv_o = kxo.gemv_w8(xn2, *m["qkv"], GS) + v["bias"]
    qo, ko, vo = qkv_o[:ql].copy(), qkv_o[ql:ql + kl].copy(), qkv_o[ql + kl:]
    qo, ko = kxo.rope(variant, ql, kl, hs, qo, ko, pos, sin, cos)

    def close(got, want, what):
        assert np.abs(got - want).max() <= 3e-5 * max(1.0, np.abs(want).max()), what

    for rep in range(2):   # twice: the second launch at the next position (new tags over the old granules)
        kx.check(L.kx_memcpy_h2d(x.ptr, v["x"].ctypes.data, dim * 4, None), "h2d")
        kx.check(L.kx_memcpy_h2d(pos_dev.ptr, np.array([pos + rep], np.int32).ctypes.data, 4, None), "h2d")
        kx.gemv_chain([wo, w13], ws, chain_id=3, pos_dev=pos_dev.ptr)
        close(x.numpy(), x1, "Wo + residual")
        close(h.numpy(), hh, "W1|W3 + SwiGLU behind the seam")
        if rep == 0:
            kx.gemv_chain([w2, qkv], ws, chain_id=4, pos_dev=pos_dev.ptr)
            close(x.numpy(), x2, "W2 + residual")
            close(q.numpy(), qo, "q behind the seam")
            close(kc.numpy().reshape(S, kl)[pos], ko, "k row")
            close(vc.numpy().reshape(S, kl)[pos], vo, "v row")
    gave = C.c_int32(0)
    kx.check(L.kx_gemv_w8_chain_status(ws.ptr, C.byref(gave)), "status")
    assert gave.value == 0
    # one op alone through the same kernel (no seam)
    kx.check(L.kx_memcpy_h2d(x.ptr, v["x"].ctypes.data, dim * 4, None), "h2d")
    kx.gemv_chain([wo], ws, chain_id=9, pos=pos)
    close(x.numpy(), x1, "single op")


def test_gemv_chain_refuses_what_it_cannot_run():
    # argument checks of the chained launch: every refusal is a return code, nothing is launched
    L = kx.lib()
    rng = np.random.default_rng(1)
    M, K = 1024, 64
    w = kx.Buf.from_numpy(rng.integers(-127, 128, (K, M), dtype=np.int8))
    s_ = kx.Buf.from_numpy(np.ones(K * M // GS, np.float32))
    x, y = kx.Buf.zeros(M), kx.Buf.zeros(K)
    ws = kx.Buf(L.kx_gemv_w8_chain_workspace_bytes(1024))
    kx.check(L.kx_gemv_w8_chain_reset(ws.ptr, ws.nbytes, None), "reset")

    def run(descs, chain_id=0, pos=0, wsb=None):
        arr = (kx.GemvDesc * len(descs))(*descs)
        return L.kx_gemv_w8_chain_f32(arr, len(descs), ws.ptr, ws.nbytes if wsb is None else wsb, chain_id, None, pos, None)

    good = kx.make_gemv_desc(kx.GEMV_PLAIN, x.ptr, w.ptr, s_.ptr, y.ptr, M, K)
    assert run([good]) == 0
    kx.sync()
    assert run([good], chain_id=128) != 0 and run([good], pos=-1) != 0
    assert run([kx.make_gemv_desc(kx.GEMV_PLAIN, x.ptr, w.ptr, s_.ptr, y.ptr, M, K, group_size=32)]) != 0
    assert run([kx.make_gemv_desc(kx.GEMV_PLAIN, x.ptr, w.ptr, s_.ptr, y.ptr, 512, K)]) != 0       # M % 1024
    assert run([kx.make_gemv_desc(kx.GEMV_PLAIN, x.ptr, w.ptr, s_.ptr, y.ptr, M, 60)]) != 0        # K % 8
    assert run([kx.make_gemv_desc(kx.GEMV_PLAIN, None, w.ptr, s_.ptr, y.ptr, M, K)]) != 0          # no input
    # a second op whose width is not the first op's row count; a workspace too small for the seam
    y2 = kx.Buf.zeros(K)
    second = kx.make_gemv_desc(kx.GEMV_PLAIN, None, w.ptr, s_.ptr, y2.ptr, M, K)
    assert run([good, second]) != 0                                                              # 64 outputs, 1024 wide
    w2 = kx.Buf.from_numpy(rng.integers(-127, 128, (1024, M), dtype=np.int8))
    s2 = kx.Buf.from_numpy(np.ones(1024 * M // GS, np.float32))
    y1k = kx.Buf.zeros(1024)
    first = kx.make_gemv_desc(kx.GEMV_PLAIN, x.ptr, w2.ptr, s2.ptr, y1k.ptr, M, 1024)
    assert run([first, second], wsb=256 + 8 * 512) != 0
    assert run([first, second]) == 0
    kx.sync()
    gave = C.c_int32(1)
    kx.check(L.kx_gemv_w8_chain_status(ws.ptr, C.byref(gave)), "status")
    assert gave.value == 0


def test_touch_bytes_reads_and_keeps_nothing():
    L = kx.lib()
    a = np.arange(1 << 20, dtype=np.int32)
    b = kx.Buf.from_numpy(a)
    sink = kx.Buf.zeros(1, np.int32)
    for n in (0, 15, 16, 4096 + 48, a.nbytes):
        kx.check(L.kx_touch_bytes(b.ptr, n, 64, sink.ptr, None), "kx_touch_bytes")
    kx.sync()
    np.testing.assert_array_equal(b.numpy(), a)
    assert sink.numpy()[0] == 0
    assert L.kx_touch_bytes(b.ptr + 4, 64, 64, sink.ptr, None) != 0      # 16-byte alignment
    assert L.kx_touch_bytes(b.ptr, 64, 0, sink.ptr, None) != 0
