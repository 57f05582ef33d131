"""One rank of the peer-mailbox collectives test (tests/test_zy_gpu_peer.py starts world_size of
these as separate processes, all on GPU 0, so the mailboxes really are mapped across processes
through IPC handles).  Ranks meet over gloo on 127.0.0.1 to exchange the 64-byte handles.
Prints one JSON line; exit status 0 only when every check of this rank passed."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))


def all_ok(dist, ok):
    """True when every rank's set-up step succeeded (a rank that cannot map the others' mailboxes
    must not leave them waiting for it)."""
    import torch
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0)


def leave_skipped(dist, rank, why):
    print(json.dumps({"rank": rank, "skip": why}), flush=True)
    dist.destroy_process_group()
    sys.exit(0)


def connect_group(dist, L, kx, world, rank, count):
    """Creates, exchanges and maps the mailboxes; returns (group, error text or None) with the same
    verdict on every rank."""
    g = C.c_void_p()
    err = None
    try:
        kx.check(L.kx_peer_create(C.byref(g), world, rank, count), "peer_create")
        h = C.create_string_buffer(64)
        kx.check(L.kx_peer_handle(g, h), "peer_handle")
        raw = h.raw
    except kx.KxError as e:
        err, raw = str(e), b"\0" * 64
    handles = [None] * world
    dist.all_gather_object(handles, raw)
    if err is None:
        try:
            kx.check(L.kx_peer_connect(g, C.create_string_buffer(b"".join(handles), 64 * world)), "peer_connect")
        except kx.KxError as e:
            err = str(e)
    if not all_ok(dist, err is None):
        return g, err or "another rank could not set its mailbox up"
    dist.barrier()  # every mailbox is zeroed and mapped before anybody writes
    return g, None


TP_SMALL = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=4, vocab_size=700,
                rope_variant=1, rope_theta=5e5, rms_eps=1e-5)
# Qwen2 traits: q/k/v biases, theta 1e6, eps 1e-6, four query heads per kv head
TP_QWEN = dict(dim=512, hidden_dim=1024, layer_num=2, head_num=8, kv_head_num=2, vocab_size=700,
               rope_variant=1, rope_theta=1e6, rms_eps=1e-6)
MODEL_SHAPES = {"llama": (TP_SMALL, False), "qwen": (TP_QWEN, True)}


def model_rank(rank, world, dist, out_path, family="llama"):
    """One tensor-parallel rank of a small model whose collectives all go through the peer
    mailboxes: ten single steps, a 530-row batched prompt, six greedy steps under graph replay."""
    import kx
    from kx import host
    kx.check(kx.lib().kx_set_device(0), "set_device")
    seq_len = 576
    rng = np.random.default_rng(5)
    toks = rng.integers(0, 700, 10).astype(np.int32)
    prompt = rng.integers(0, 700, 530).astype(np.int32)
    shape, bias = MODEL_SHAPES[family]
    m = host.Model(seq_len=seq_len, group_size=64, tp_size=world, tp_rank=rank, qkv_bias=bias, **shape)
    m.init_synthetic(21)
    err = None
    try:
        raw = m.peer_create(len(prompt) * shape["dim"])
    except kx.KxError as e:
        err, raw = str(e), b"\0" * 64
    handles = [None] * world
    dist.all_gather_object(handles, raw)
    if err is None:
        try:
            m.peer_connect(handles)
        except kx.KxError as e:
            err = str(e)
    if not all_ok(dist, err is None):
        m.close()
        leave_skipped(dist, rank, err or "another rank could not set its mailbox up")
    dist.barrier()
    steps = np.stack([m.step_logits(int(t), p) for p, t in enumerate(toks)])
    last = m.prefill(prompt)
    m.decode(6, use_graph=True)
    hist = m.history(len(prompt) + 7)
    timed_out = m.peer_timed_out()
    dist.barrier()
    np.savez(out_path, steps=steps, last=last, hist=hist, timed_out=np.array([int(timed_out)]))
    m.close()
    print(json.dumps({"rank": rank, "ok": not timed_out, "timed_out": int(timed_out)}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if not timed_out else 1)


def absent_rank(rank, world, dist, count):
    """Rank 0 reduces, the other ranks never do: rank 0's wait must give up (4 s), raise the flag and
    let the launch finish; a second call must then return without waiting again."""
    import time
    import kx
    L = kx.lib()
    kx.check(L.kx_set_device(0), "set_device")
    st = kx.Stream()
    g, err = connect_group(dist, L, kx, world, rank, count)
    if err:
        leave_skipped(dist, rank, err)
    ok, first_s, second_s, flag = True, 0.0, 0.0, C.c_int32(0)
    if rank == 0:
        buf = kx.Buf.from_numpy(np.ones(count, np.float32))
        t0 = time.perf_counter()
        kx.check(L.kx_peer_allreduce_sum_f32(g, buf.ptr, count, st.ptr), "peer_allreduce")
        st.sync()
        first_s = time.perf_counter() - t0
        kx.check(L.kx_peer_timed_out(g, C.byref(flag)), "peer_timed_out")
        t0 = time.perf_counter()
        kx.check(L.kx_peer_allreduce_sum_f32(g, buf.ptr, count, st.ptr), "peer_allreduce")
        st.sync()
        second_s = time.perf_counter() - t0
        ok = flag.value == 1 and 2.0 < first_s < 20.0 and second_s < 1.0
    dist.barrier()
    if rank != 0:
        # rank 0 has left the protocol and said so in every mailbox: a rank that reduces now must
        # not trust the sum -- its own flag goes up within this one call, without a 4 s wait
        buf = kx.Buf.from_numpy(np.ones(count, np.float32))
        t0 = time.perf_counter()
        kx.check(L.kx_peer_allreduce_sum_f32(g, buf.ptr, count, st.ptr), "peer_allreduce")
        st.sync()
        first_s = time.perf_counter() - t0
        kx.check(L.kx_peer_timed_out(g, C.byref(flag)), "peer_timed_out")
        ok = flag.value == 1 and first_s < 1.5
    dist.barrier()
    kx.check(L.kx_peer_destroy(g), "peer_destroy")
    print(json.dumps({"rank": rank, "ok": bool(ok), "timed_out": 0, "flag": flag.value, "first_s": first_s,
                      "second_s": second_s}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    rank, world, port, count = (int(v) for v in sys.argv[1:5])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if len(sys.argv) > 5 and sys.argv[5] == "model":
        return model_rank(rank, world, dist, sys.argv[6], sys.argv[7] if len(sys.argv) > 7 else "llama")
    if len(sys.argv) > 5 and sys.argv[5] == "absent":
        return absent_rank(rank, world, dist, count)
    import kx
    L = kx.lib()
    kx.check(L.kx_set_device(0), "set_device")
    st = kx.Stream()
    g, err = connect_group(dist, L, kx, world, rank, count)
    if err:
        leave_skipped(dist, rank, err)

    def vec(r, it, n):
        return np.random.default_rng(1000 * it + r).standard_normal(n).astype(np.float32)

    def expect_sum(it, n):
        acc = vec(0, it, n).copy()
        for r in range(1, world):
            acc = acc + vec(r, it, n)  # rank order, fp32: what the kernel does
        return acc

    ok = True
    worst = 0.0
    # eager calls, alternating sizes (both parities of the slot pair, odd tails)
    sizes = [count, 4096 if count >= 4096 else count, 1000, 6, count]
    for it, n in enumerate(sizes):
        n = min(n, count)
        buf = kx.Buf.from_numpy(np.concatenate([vec(rank, it, n), np.zeros(4, np.float32)]))
        kx.check(L.kx_peer_allreduce_sum_f32(g, buf.ptr, n, st.ptr), "peer_allreduce")
        st.sync()
        got = buf.numpy(np.float32, (n + 4,))
        want = expect_sum(it, n)
        ok &= bool(np.array_equal(got[:n], want)) and bool(np.all(got[n:] == 0))
        worst = max(worst, float(np.max(np.abs(got[:n] - want))))
    # all-gather of two floats per rank (the argmax candidates)
    parts = np.zeros((world, 2), np.float32)
    parts[rank] = [rank + 0.5, 100 + rank]
    gb = kx.Buf.from_numpy(parts)
    kx.check(L.kx_peer_allgather_f32(g, gb.ptr, 2, st.ptr), "peer_allgather")
    st.sync()
    gathered = gb.numpy(np.float32, (world, 2))
    ok &= bool(np.array_equal(gathered, np.array([[r + 0.5, 100 + r] for r in range(world)], np.float32)))
    # parts of six floats: an odd rank's part starts 8 bytes off a 16-byte boundary
    parts6 = np.zeros((world, 6), np.float32)
    parts6[rank] = np.arange(6) + 10 * rank
    gb6 = kx.Buf.from_numpy(parts6)
    kx.check(L.kx_peer_allgather_f32(g, gb6.ptr, 6, st.ptr), "peer_allgather")
    st.sync()
    ok &= bool(np.array_equal(gb6.numpy(np.float32, (world, 6)),
                              np.array([np.arange(6) + 10 * r for r in range(world)], np.float32)))
    # a captured launch replayed: x <- sum over ranks of x, three times: x0 * world^3 (exact in fp32
    # for these small integers)
    n = min(count, 2048)
    xb = kx.Buf.from_numpy(np.full(n, float(rank + 1), np.float32))
    gr = C.c_void_p()
    kx.check(L.kx_graph_begin_capture(st.ptr), "begin_capture")
    kx.check(L.kx_peer_allreduce_sum_f32(g, xb.ptr, n, st.ptr), "peer_allreduce (captured)")
    kx.check(L.kx_graph_end_capture(st.ptr, C.byref(gr)), "end_capture")
    for _ in range(3):
        kx.check(L.kx_graph_launch(gr, st.ptr), "graph_launch")
    st.sync()
    s1 = world * (world + 1) / 2
    ok &= bool(np.all(xb.numpy(np.float32, (n,)) == np.float32(s1 * world * world)))
    # timing of back-to-back replays (same-GPU processes: no xGMI in it)
    e0, e1 = kx.Event(), kx.Event()
    dist.barrier()
    e0.record(st.ptr)
    reps = 50
    for _ in range(reps):
        L.kx_graph_launch(gr, st.ptr)
    e1.record(st.ptr)
    e1.sync()
    us = e0.elapsed_ms(e1) * 1e3 / reps
    flag = C.c_int32(0)
    kx.check(L.kx_peer_timed_out(g, C.byref(flag)), "peer_timed_out")
    ok &= flag.value == 0
    dist.barrier()
    L.kx_graph_destroy(gr)
    kx.check(L.kx_peer_destroy(g), "peer_destroy")
    print(json.dumps({"rank": rank, "ok": bool(ok), "worst_abs_diff": worst, "timed_out": flag.value,
                      "us_per_allreduce": us}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
