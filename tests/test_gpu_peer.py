"""Peer-mailbox collectives across PROCESSES: world_size ranks, each its own process, all on the
one GPU of the box; mailboxes mapped through IPC handles.  Sums must equal the fp32 sum taken in
rank order bit for bit, on every rank, eagerly and under graph replay."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world,count", [(2, 4096), (4, 8192), (8, 4096)])
def test_peer_collectives_across_processes(world, count):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_worker.py"), str(r), str(world),
                               str(port), str(count)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=100)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, out, err in outs:
        assert rc == 0, f"rank failed (rc {rc}):\n{out}\n{err[-2000:]}"
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["ok"] and rec["timed_out"] == 0, rec
