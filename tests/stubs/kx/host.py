"""Stand-in for kx.host (see the package docstring): a Model whose calls take a little time and
keep the bookkeeping bench.py reads back (state, peer flags), nothing else."""
import time

import numpy as np

from . import FAIL_PEER_RANK, KxError

MODEL_SHAPES = {
    "llama-3-8b": dict(dim=4096, hidden_dim=14336, layer_num=32, head_num=32, kv_head_num=8,
                       vocab_size=128256, rope_variant=1, rope_theta=5e5, rms_eps=1e-5),
}


def comm_unique_id():
    return bytes(128)


class Model:
    def __init__(self, seq_len, group_size=64, tp_size=1, tp_rank=0, kv_f16=False, **shape):
        self.seq_len, self.tp_size, self.tp_rank, self.shape = seq_len, tp_size, tp_rank, shape
        self.tok, self.pos = 0, 0
        self.comm = self.peer = False
        self.calls = []

    def init_synthetic(self, seed=1234):
        self.calls.append("init_synthetic")

    def init_comm(self, id128):
        assert len(id128) == 128
        self.comm = True

    def peer_create(self, max_floats):
        return bytes([self.tp_rank]) * 64

    def peer_connect(self, handles):
        assert len(handles) == self.tp_size and all(len(h) == 64 for h in handles)
        if self.tp_rank == FAIL_PEER_RANK:
            raise KxError("stub: this rank cannot map a mailbox")
        self.peer = True

    def peer_disable(self):
        self.peer = False

    def peer_timed_out(self):
        return False

    def set_prefill_subblocks(self, n):
        pass

    def set_prefill_weight_cache(self, policy):
        pass

    def weight_image_bytes(self):
        return 0

    def prefill(self, tokens, pos0=0, exact=False):
        time.sleep(0.002)
        self.tok, self.pos = 1, pos0 + len(tokens)
        return np.zeros(self.shape["vocab_size"], np.float32)

    def set_state(self, token, pos):
        self.tok, self.pos = token, pos

    def state(self):
        return self.tok, self.pos

    def decode(self, steps, use_graph=True):
        if self.pos + steps > self.seq_len:
            raise KxError("decode would run past seq_len")
        time.sleep(0.0005 * steps)
        self.pos += steps

    def profile_gemv(self, steps):
        return 129 * steps, 7.98e9 * steps / self.tp_size, 1.6 * steps

    def profile_allreduce(self, count, iters):
        if self.tp_size > 1 and not (self.comm or self.peer):
            raise KxError("no communicator")
        return 0.01

    def linear_bytes_per_token(self):
        return 7.973e9 / self.tp_size
