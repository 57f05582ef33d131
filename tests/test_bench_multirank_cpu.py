"""bench.py's whole N > 1 control flow on two CPU processes over gloo, against a GPU-less stand-in
for the kx package (tests/stubs): rendezvous, RCCL-id broadcast, the timed loops and their
max-over-ranks, the communication profile, the peer-mailbox stages with their rank agreement, and
the one JSON line from rank 0 -- so that the first run on a real multi-GPU node cannot die on
plumbing.  Nothing here is a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra_env=None, extra_args=()):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   KX_BENCH_PACKAGE_DIR=os.path.join(ROOT, "tests", "stubs"), **(extra_env or {}))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
             "--context", "2048", "--no-prefill-kernels", *extra_args],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][0]
    assert not any(ln.startswith("{") for o, _ in outs[1:] for ln in o.splitlines())   # rank 0 alone prints
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_bench_line(world):
    line = _run(world)
    assert line["n_gpus"] == world and line["steps"] == 6 and line["warmup"] == 2
    assert line["scaling"] == "strong" and line["unit"] == "tok/s" and line["value"] > 0
    assert line["config"]["parallelism"] == f"tp{world}"
    comm = line["tensor_parallel_comm"]
    assert comm["allreduces_per_token"] == 64 and comm["decode_allreduce_bytes"] == 4096 * 4
    assert set(comm["prefill_overlap"]) >= {"serial_tok_per_s", "overlapped_tok_per_s", "overlap_fraction"}
    pm = comm["peer_mailbox"]
    assert pm["timed_out"] is False and pm["tok_per_s"] > 0 and pm["rccl_tok_per_s"] > 0
    assert "cpu_baseline" not in line and "config1" not in line        # timed at N = 1 only
    assert line["roofline"]["traffic"] is None


def test_a_rank_that_cannot_map_a_mailbox_keeps_the_rccl_figure():
    line = _run(2, extra_env={"KX_STUB_FAIL_PEER_RANK": "1"})
    pm = line["tensor_parallel_comm"]["peer_mailbox"]
    assert "error" in pm and "tok_per_s" not in pm
    assert line["config"]["token_loop_collective"] == "rccl all-reduce on a side stream"
    assert line["value"] > 0


def test_rehearsal_mode_takes_the_mailbox_route_only():
    line = _run(2, extra_args=("--one-gpu-rehearsal",))
    assert "REHEARSAL" in line["config"]["token_loop_collective"]
    assert line["config"]["context"] == 0
