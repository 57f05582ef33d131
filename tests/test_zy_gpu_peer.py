"""(Named to be collected near the end: these tests start several GPU processes.)
Peer-mailbox collectives across PROCESSES: world_size ranks, each its own process, all on the
one GPU of the box; mailboxes mapped through IPC handles.  Sums must equal the fp32 sum taken in
rank order bit for bit, on every rank, eagerly and under graph replay."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "llm-inference-engine_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, count, extra=()):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_worker.py"), str(r), str(world),
                               str(port), str(count)] + [str(e).replace("%r", str(r)) for e in extra],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=100)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    recs = []
    for rc, out, err in outs:
        assert rc == 0, f"rank failed (rc {rc}):\n{out}\n{err[-2000:]}"
        recs.append(json.loads(out.strip().splitlines()[-1]))
    # a box whose runtime cannot share device memory between processes (no IPC handles, no
    # fine-grained allocations): every rank reports the set-up error and leaves; nothing was computed
    skipped = [r["skip"] for r in recs if "skip" in r]
    if skipped:
        pytest.skip("peer mailboxes could not be set up on this box: " + skipped[0])
    for rec in recs:
        assert rec["ok"] and rec["timed_out"] == 0, rec


@pytest.mark.gpu
@pytest.mark.parametrize("world,count", [(2, 4096), (4, 8192), (8, 4096)])
def test_peer_collectives_across_processes(world, count):
    _run_ranks(world, count)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_peer_wait_gives_up_when_a_rank_is_absent(world):
    # the bounded wait: a reduce whose partners never show up ends after 4 s with the flag raised,
    # and later calls on that group do not wait again; the rank that gave up marks itself dead in
    # every mailbox, so the other ranks' next call raises THEIR flags at once (no healthy rank is
    # left adding the slots of a rank that has stopped taking part)
    _run_ranks(world, 1024, ("absent",))


@pytest.mark.gpu
@pytest.mark.parametrize("world,family", [(2, "llama"), (4, "llama"), (2, "qwen")])
def test_tensor_parallel_model_across_processes(world, family, tmp_path):
    # The tensor-parallel model with one PROCESS per rank (all on this box's one GPU), every
    # collective of the token loop and of the prompt pass through the peer mailboxes, the greedy
    # steps under graph replay: all ranks agree bit for bit, and match the unsharded model run here
    # to the bound of the in-process rig (row-parallel sums are re-associated).
    import peer_worker
    from kx import host
    shape, bias = peer_worker.MODEL_SHAPES[family]
    rng = np.random.default_rng(5)
    toks = rng.integers(0, 700, 10).astype(np.int32)
    prompt = rng.integers(0, 700, 530).astype(np.int32)
    one = host.Model(seq_len=576, group_size=64, qkv_bias=bias, **shape)
    one.init_synthetic(21)
    want = np.stack([one.step_logits(int(t), p) for p, t in enumerate(toks)])
    ref_last = one.prefill(prompt)
    one.decode(6, use_graph=True)
    ref_hist = one.history(len(prompt) + 7)
    one.close()
    _run_ranks(world, 4096, ("model", str(tmp_path / "rank%r.npz"), family))
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in range(world):
        assert int(got[r]["timed_out"][0]) == 0
        np.testing.assert_array_equal(got[r]["steps"], got[0]["steps"])
        np.testing.assert_array_equal(got[r]["hist"], got[0]["hist"])
        assert np.abs(got[r]["last"] - ref_last).max() <= 2e-3 * np.abs(ref_last).max()
    assert np.abs(got[0]["steps"] - want).max() <= 1e-4 * np.abs(want).max()
    order = np.argsort(ref_last)
    if ref_last[order[-1]] - ref_last[order[-2]] > 1e-2 * np.abs(ref_last).max():
        assert got[0]["hist"][len(prompt)] == ref_hist[len(prompt)]
