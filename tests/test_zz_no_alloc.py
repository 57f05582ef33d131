"""The entry point that allocates nothing (kx_mha_decode_static_f32): its split workspace is a
fixed array inside the library.  Checked through the C ABI against kx_mha_decode_f32 (bit for bit:
same kernels, same split count) and against the CPU oracle, on two calls in a row, the second with
a smaller input, so leftovers of the first call in the shared array must not leak into it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kx  # noqa: E402
from oracle import kxo  # noqa: E402


def _case(H, KVH, hs, S, pos, seed):
    rng = np.random.default_rng(seed)
    kc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    vc = rng.standard_normal((2, S, KVH * hs)).astype(np.float32)
    q = rng.standard_normal(H * hs).astype(np.float32)
    return q, kc, vc


def _static(q, kc, vc, pos, H, kv_mul, hs, layer):
    L, S, kv_dim = kc.shape
    dq, dk, dv = kx.Buf.from_numpy(q), kx.Buf.from_numpy(kc), kx.Buf.from_numpy(vc)
    out = kx.Buf.zeros(H * hs)
    kx.check(kx.lib().kx_mha_decode_static_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos, H, kv_mul, hs,
                                               kv_dim, S, layer, None), "kx_mha_decode_static_f32")
    return out.numpy(np.float32, (H, hs))


def test_static_scratch_attention_two_calls_second_smaller():
    # first: Llama-3-8B geometry at 4.3k keys (64 splits per kv head); second: fewer heads, a
    # 64-wide head and a short context (few splits), right behind it on the same GPU
    for (H, KVH, hs, S, pos) in ((32, 8, 128, 4608, 4300), (8, 2, 64, 512, 333), (4, 2, 48, 96, 70)):
        q, kc, vc = _case(H, KVH, hs, S, pos, seed=pos)
        got = _static(q, kc, vc, pos, H, H // KVH, hs, 1)
        same = kx.mha_decode(q, kc, vc, pos, H, H // KVH, hs, 1)
        np.testing.assert_array_equal(got, same)
        want = kxo.mha(pos, H, 1, S, KVH * hs, H // KVH, hs, q, kc, vc).reshape(H, hs)
        assert np.abs(got - want).max() <= 2e-5


def test_static_scratch_attention_rejects_what_does_not_fit():
    limit = kx.lib().kx_mha_decode_static_limit_bytes()
    assert limit == 16384 + 128 * 64 * 132 * 4
    H, KVH, hs, S = 128, 16, 128, 64   # 128 heads of 128: twice the capacity
    assert kx.lib().kx_attn_workspace_bytes(H, hs) > limit
    q, kc, vc = _case(H, KVH, hs, S, 10, seed=1)
    with pytest.raises(kx.KxError, match="fixed scratch capacity"):
        _static(q, kc, vc, 10, H, H // KVH, hs, 0)
