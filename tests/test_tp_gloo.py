"""Tensor-parallel decode path on CPU: two processes over gloo.

Each rank takes its slice of every layer matrix from the library's own shard plan
(kxm_shard_plan, the host logic LlamaModel uses to load and to generate weights), runs the rank's
part of the token loop with the oracle kernels, and sums the row-parallel partial outputs with an
all-reduce at the two places the device path does (after Wo and after W2, the hook of
kuiper/source/op/matmul.cpp:81-127).  Result must equal the unsharded oracle token loop.
"""
import os
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

GS = 64
SHAPE = dict(dim=128, hidden_dim=256, layer_num=2, head_num=4, kv_head_num=2, vocab_size=96)
SEQ = 12
TOKENS = [5, 17, 90, 3, 44, 61]


def _weights(rope_variant):
    rng = np.random.default_rng(100 + rope_variant)
    d, h, L, V = SHAPE["dim"], SHAPE["hidden_dim"], SHAPE["layer_num"], SHAPE["vocab_size"]
    kvd = d * SHAPE["kv_head_num"] // SHAPE["head_num"]
    a = {}
    for name, (K, M) in dict(wq=(d, d), wk=(kvd, d), wv=(kvd, d), wo=(d, d), w1=(h, d), w2=(d, h),
                             w3=(h, d)).items():
        a[name + "_q"] = rng.integers(-127, 128, size=(L, K, M), dtype=np.int8)
        a[name + "_s"] = rng.uniform(5e-4, 2e-3, size=(L, K, M // GS)).astype(np.float32)
    a["cls_q"] = rng.integers(-127, 128, size=(V, d), dtype=np.int8)
    a["cls_s"] = rng.uniform(5e-4, 2e-3, size=(V, d // GS)).astype(np.float32)
    a["emb"] = rng.standard_normal((V, d)).astype(np.float32)
    a["attn_norm"] = rng.uniform(0.5, 1.5, (L, d)).astype(np.float32)
    a["ffn_norm"] = rng.uniform(0.5, 1.5, (L, d)).astype(np.float32)
    a["final_norm"] = rng.uniform(0.5, 1.5, d).astype(np.float32)
    return a


def _rank_main(rank, world, rdzv, rope_variant, theta, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "llm-inference-engine_amd"))
    import torch
    import torch.distributed as dist
    from kx import host
    from oracle import kxo

    dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
    a = _weights(rope_variant)
    d, L, H, KVH = SHAPE["dim"], SHAPE["layer_num"], SHAPE["head_num"], SHAPE["kv_head_num"]
    hs = d // H
    plan, carries = host.shard_plan(d, SHAPE["hidden_dim"], H, KVH, world, rank)

    def shard(name, l):
        r0, rows, c0, cols = plan[name]
        q = a[name + "_q"][l][r0:r0 + rows, c0:c0 + cols]
        s = a[name + "_s"][l][r0:r0 + rows, c0 // GS:(c0 + cols) // GS]
        return np.ascontiguousarray(q), np.ascontiguousarray(s).reshape(-1)

    def all_reduce(x):
        t = torch.from_numpy(x)
        dist.all_reduce(t)
        return t.numpy()

    hl, kvl = H // world, KVH // world
    ql, kl = hl * hs, kvl * hs
    kc = np.zeros((L, SEQ, kl), np.float32)
    vc = np.zeros((L, SEQ, kl), np.float32)
    sin, cos = kxo.sin_cos_cache(hs, SEQ, theta)
    logits = []
    for pos, tok in enumerate(TOKENS):
        x = a["emb"][tok].copy()
        for l in range(L):
            xn = kxo.rmsnorm(x, a["attn_norm"][l], 1e-5)
            q = kxo.gemv_w8(xn, *shard("wq", l), GS)
            k = kxo.gemv_w8(xn, *shard("wk", l), GS)
            v = kxo.gemv_w8(xn, *shard("wv", l), GS)
            q, k = kxo.rope(rope_variant, ql, kl, hs, q, k, pos, sin, cos)
            kc[l, pos], vc[l, pos] = k, v
            att = kxo.mha(pos, hl, l, SEQ, kl, hl // kvl, hs, q, kc, vc)
            o = kxo.gemv_w8(att, *shard("wo", l), GS)
            x = all_reduce(o + x if carries else o)        # rank 0 carries the residual
            xn = kxo.rmsnorm(x, a["ffn_norm"][l], 1e-5)
            h = kxo.swiglu(kxo.gemv_w8(xn, *shard("w1", l), GS), kxo.gemv_w8(xn, *shard("w3", l), GS))
            o = kxo.gemv_w8(h, *shard("w2", l), GS)
            x = all_reduce(o + x if carries else o)
        xn = kxo.rmsnorm(x, a["final_norm"], 1e-5)
        logits.append(kxo.gemv_w8(xn, a["cls_q"], a["cls_s"].reshape(-1), GS))  # replicated
    np.save(os.path.join(out_dir, f"logits{rank}.npy"), np.stack(logits))
    np.save(os.path.join(out_dir, f"k{rank}.npy"), kc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rope_variant,theta", [(0, 1e4), (1, 5e5)])
def test_two_rank_token_loop_equals_unsharded(rope_variant, theta):
    from oracle import kxo
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_rank_main,
                             args=(r, world, os.path.join(tmp, "rdzv"), rope_variant, theta, tmp))
                 for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        got = [np.load(os.path.join(tmp, f"logits{r}.npy")) for r in range(world)]
        ks = [np.load(os.path.join(tmp, f"k{r}.npy")) for r in range(world)]
    cfg = kxo.Config(seq_len=SEQ, group_size=GS, rope_variant=rope_variant, rope_theta=theta,
                     rms_eps=1e-5, **SHAPE)
    full = kxo.Model(cfg, arrays=_weights(rope_variant))
    want = np.stack([full.forward(t, pos) for pos, t in enumerate(TOKENS)])
    np.testing.assert_array_equal(got[0], got[1])  # every rank ends with the same logits
    assert np.abs(got[0] - want).max() <= 2e-5 * np.abs(want).max()
    # each rank's key cache holds its kv heads' columns of the unsharded cache
    kfull, _ = full.caches()
    n = len(TOKENS)
    kl = ks[0].shape[2]
    for r in range(world):
        np.testing.assert_allclose(ks[r][:, :n], kfull[:, :n, r * kl:(r + 1) * kl], rtol=0, atol=2e-5)


def test_shard_plan_partitions_every_matrix():
    import sys
    from kx import host
    d, hdim, H, KVH = 4096, 14336, 32, 8
    hs = d // H
    kvd = KVH * hs
    full = dict(wq=(d, d), wk=(kvd, d), wv=(kvd, d), wo=(d, d), w1=(hdim, d), w2=(d, hdim),
                w3=(hdim, d))
    for world in (1, 2, 4, 8):
        cover = {n: np.zeros(full[n], np.int8) for n in full}
        owners = 0
        for r in range(world):
            plan, carries = host.shard_plan(d, hdim, H, KVH, world, r)
            owners += carries
            for n, (r0, rows, c0, cols) in plan.items():
                cover[n][r0:r0 + rows, c0:c0 + cols] += 1
                assert c0 % GS == 0 and cols % GS == 0  # quant groups are never cut
        assert owners == 1
        for n in full:
            assert (cover[n] == 1).all(), (world, n)  # disjoint and complete
    with pytest.raises(ValueError):
        host.shard_plan(d, hdim, H, KVH, 3, 0)
    with pytest.raises(ValueError):
        host.shard_plan(d, hdim, H, KVH, 16, 0)  # more ranks than kv heads
