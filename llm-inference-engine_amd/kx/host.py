"""ctypes binding of libkx_host.so: the C++17 host layer's model driver and layer probes
(host/include/kx_host.h).  FFI plumbing only; no compute happens in Python."""
import ctypes as C
import os

import numpy as np

from . import KxError, LIB_DIR, lib as _kxlib

HOST_LIB_PATH = os.path.join(LIB_DIR, "libkx_host.so")

ROPE_INTERLEAVED, ROPE_HALF_SPLIT = 0, 1


class Config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("hidden_dim", C.c_int32), ("layer_num", C.c_int32),
                ("head_num", C.c_int32), ("kv_head_num", C.c_int32), ("vocab_size", C.c_int32),
                ("seq_len", C.c_int32), ("group_size", C.c_int32), ("rope_variant", C.c_int32),
                ("rope_theta", C.c_float), ("rms_eps", C.c_float), ("tp_size", C.c_int32),
                ("tp_rank", C.c_int32), ("kv_f16", C.c_int32), ("qkv_bias", C.c_int32),
                ("prefill_weight_cache", C.c_int32)]


# the model shapes named by the benchmark configs
MODEL_SHAPES = {
    "stories110M": dict(dim=768, hidden_dim=2048, layer_num=12, head_num=12, kv_head_num=12,
                        vocab_size=32000, rope_variant=0, rope_theta=1e4, rms_eps=1e-5),
    "llama-3.2-1b": dict(dim=2048, hidden_dim=8192, layer_num=16, head_num=32, kv_head_num=8,
                         vocab_size=128256, rope_variant=1, rope_theta=5e5, rms_eps=1e-5),
    "llama-3-8b": dict(dim=4096, hidden_dim=14336, layer_num=32, head_num=32, kv_head_num=8,
                       vocab_size=128256, rope_variant=1, rope_theta=5e5, rms_eps=1e-5),
    "qwen2.5-7b": dict(dim=3584, hidden_dim=18944, layer_num=28, head_num=28, kv_head_num=4,
                       vocab_size=152064, rope_variant=1, rope_theta=1e6, rms_eps=1e-6),
    "llama-3-70b": dict(dim=8192, hidden_dim=28672, layer_num=80, head_num=64, kv_head_num=8,
                        vocab_size=128256, rope_variant=1, rope_theta=5e5, rms_eps=1e-5),
}

_hlib = None
_vp, _i32 = C.c_void_p, C.c_int32


def hlib():
    global _hlib
    if _hlib is None:
        _kxlib()  # libkx_hip.so first (also the loud failure when it is missing)
        if not os.path.exists(HOST_LIB_PATH):
            raise KxError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(HOST_LIB_PATH)
        L.kxm_last_error.restype = C.c_char_p
        L.kxm_create.restype = _vp
        L.kxm_create.argtypes = [C.POINTER(Config)]
        L.kxm_destroy.argtypes = [_vp]
        L.kxm_init_synthetic.argtypes = [_vp, C.c_uint64]
        L.kxm_load_blob.argtypes = [_vp, _vp, C.c_size_t]
        L.kxm_set_qkv_bias.argtypes = [_vp, _i32, _vp, _vp, _vp]
        L.kxm_read_weight.argtypes = [_vp, C.c_char_p, _i32, _vp, C.c_size_t]
        L.kxm_comm_unique_id.argtypes = [_vp]
        L.kxm_init_comm.argtypes = [_vp, _vp]
        L.kxm_peer_create.argtypes = [_vp, C.c_uint64, _vp]
        L.kxm_peer_connect.argtypes = [_vp, _vp]
        L.kxm_peer_timed_out.argtypes = [_vp, C.POINTER(_i32)]
        L.kxm_peer_disable.argtypes = [_vp]
        L.kxm_forward_layers.argtypes = [_vp, _i32, _i32, _vp]
        L.kxm_set_state.argtypes = [_vp, _i32, _i32]
        L.kxm_step_logits.argtypes = [_vp, _i32, _i32, _vp]
        L.kxm_decode.argtypes = [_vp, _i32, _i32]
        L.kxm_prefill.argtypes = [_vp, _vp, _i32, _i32, _vp, _i32]
        L.kxm_read_history.argtypes = [_vp, _vp, _i32]
        L.kxm_read_state.argtypes = [_vp, C.POINTER(_i32), C.POINTER(_i32)]
        L.kxm_read_kv.argtypes = [_vp, _i32, _i32, _vp, _vp]
        L.kxm_sync.argtypes = [_vp]
        L.kxm_profile_gemv.argtypes = [_vp, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]
        L.kxm_profile_allreduce.argtypes = [_vp, C.c_uint64, _i32, C.POINTER(C.c_double)]
        L.kxm_linear_bytes_per_token.restype = C.c_double
        L.kxm_linear_bytes_per_token.argtypes = [_vp]
        L.kxm_stream.restype = _vp
        L.kxm_stream.argtypes = [_vp]
        L.kxm_test_matmul_layer.argtypes = [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp]
        L.kxm_test_mha_layer.argtypes = [_vp, _vp, _vp] + [_i32] * 7 + [_vp]
        L.kxm_test_mha_layer_no_workspace.argtypes = [_vp, _vp, _vp] + [_i32] * 7 + [_vp]
        L.kxm_test_flash_layer.argtypes = [_vp, _vp, _vp] + [_i32] * 8 + [_vp]
        L.kxm_test_mha_from_registry_pieces.argtypes = [_vp, _vp, _vp] + [_i32] * 7 + [_vp]
        L.kxm_test_kv_layout.argtypes = [_i32] * 5 + [C.POINTER(C.c_int64)] * 3
        _hlib = L
    return _hlib


def hcheck(rc, what=""):
    if rc != 0:
        raise KxError(f"{what} failed: rc={rc}: {hlib().kxm_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Model:
    def __init__(self, seq_len, group_size=64, tp_size=1, tp_rank=0, kv_f16=False, qkv_bias=False,
                 prefill_weight_cache=-1, **shape):
        self.cfg = Config(seq_len=seq_len, group_size=group_size, tp_size=tp_size, tp_rank=tp_rank,
                          kv_f16=1 if kv_f16 else 0, qkv_bias=1 if qkv_bias else 0,
                          prefill_weight_cache=prefill_weight_cache, **shape)
        self.h = hlib().kxm_create(C.byref(self.cfg))
        if not self.h:
            raise KxError("kxm_create failed: " + hlib().kxm_last_error().decode())
        c = self.cfg
        self.head_size = c.dim // c.head_num
        self.kv_dim_local = c.kv_head_num // tp_size * self.head_size
        self.q_dim_local = c.head_num // tp_size * self.head_size
        self.hidden_local = c.hidden_dim // tp_size
        # the classifier's rows are split over the ranks when the vocabulary divides
        self.vocab_local = c.vocab_size // tp_size if c.vocab_size % tp_size == 0 else c.vocab_size

    def close(self):
        if self.h:
            hlib().kxm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_synthetic(self, seed=1234):
        hcheck(hlib().kxm_init_synthetic(self.h, seed), "kxm_init_synthetic")

    def load_blob(self, blob: bytes):
        buf = np.frombuffer(blob, np.uint8)
        hcheck(hlib().kxm_load_blob(self.h, _p(buf), buf.size), "kxm_load_blob")

    def set_qkv_bias(self, layer, bq, bk, bv):
        a = [np.ascontiguousarray(b, np.float32) for b in (bq, bk, bv)]
        hcheck(hlib().kxm_set_qkv_bias(self.h, layer, _p(a[0]), _p(a[1]), _p(a[2])),
               "kxm_set_qkv_bias")

    def weight_shape(self, name):
        c = self.cfg
        gs = c.group_size
        shapes = {"wq": (self.q_dim_local, c.dim), "wk": (self.kv_dim_local, c.dim),
                  "wv": (self.kv_dim_local, c.dim), "wo": (c.dim, self.q_dim_local),
                  "w1": (self.hidden_local, c.dim), "w3": (self.hidden_local, c.dim),
                  "w2": (c.dim, self.hidden_local), "cls": (self.vocab_local, c.dim)}
        if name in shapes:
            return shapes[name], (np.int8 if gs else np.float32)
        if name.endswith("_s"):
            r, k = shapes[name[:-2]]
            return (r * k // gs,), np.float32
        if name in ("bq", "bk", "bv"):
            return (self.q_dim_local if name == "bq" else self.kv_dim_local,), np.float32
        if name == "emb":
            return (c.vocab_size, c.dim), np.float32
        return (c.dim,), np.float32

    def read_weight(self, name, layer=0):
        shape, dt = self.weight_shape(name)
        out = np.empty(shape, dt)
        hcheck(hlib().kxm_read_weight(self.h, name.encode(), layer, _p(out), out.nbytes),
               f"kxm_read_weight({name})")
        return out

    def init_comm(self, id128: bytes):
        buf = np.frombuffer(id128, np.uint8).copy() if id128 is not None else None
        hcheck(hlib().kxm_init_comm(self.h, _p(buf)), "kxm_init_comm")

    def peer_create(self, max_floats) -> bytes:
        """Creates this rank's peer mailbox; returns its 64-byte IPC handle for the other ranks."""
        h = C.create_string_buffer(64)
        hcheck(hlib().kxm_peer_create(self.h, C.c_uint64(max_floats), h), "kxm_peer_create")
        return h.raw

    def peer_connect(self, handles):
        """handles: every rank's handle, ordered by rank.  Follow with a barrier between the ranks."""
        blob = C.create_string_buffer(b"".join(handles), 64 * len(handles))
        hcheck(hlib().kxm_peer_connect(self.h, blob), "kxm_peer_connect")

    def peer_disable(self):
        hcheck(hlib().kxm_peer_disable(self.h), "kxm_peer_disable")

    def peer_timed_out(self) -> bool:
        f = _i32(0)
        hcheck(hlib().kxm_peer_timed_out(self.h, C.byref(f)), "kxm_peer_timed_out")
        return bool(f.value)

    def set_prefill_weight_cache(self, policy):
        L = hlib()
        L.kxm_set_prefill_weight_cache.argtypes = [_vp, _i32]
        hcheck(L.kxm_set_prefill_weight_cache(self.h, int(policy)), "kxm_set_prefill_weight_cache")

    def weight_image_bytes(self):
        L = hlib()
        L.kxm_weight_image_bytes.argtypes = [_vp, C.POINTER(C.c_uint64)]
        n = C.c_uint64(0)
        hcheck(L.kxm_weight_image_bytes(self.h, C.byref(n)), "kxm_weight_image_bytes")
        return n.value

    def set_prefill_subblocks(self, n):
        L = hlib()
        L.kxm_set_prefill_subblocks.argtypes = [_vp, _i32]
        hcheck(L.kxm_set_prefill_subblocks(self.h, int(n)), "kxm_set_prefill_subblocks")

    def join_loopback(self, group):
        hcheck(hlib().kxm_join_loopback(self.h, group), "kxm_join_loopback")

    def forward_layers(self, token, pos):
        out = np.empty(self.cfg.vocab_size, np.float32)
        hcheck(hlib().kxm_forward_layers(self.h, int(token), int(pos), _p(out)), "kxm_forward_layers")
        return out

    def step_logits(self, token, pos):
        out = np.empty(self.cfg.vocab_size, np.float32)
        hcheck(hlib().kxm_step_logits(self.h, int(token), int(pos), _p(out)), "kxm_step_logits")
        return out

    def set_state(self, token, pos):
        hcheck(hlib().kxm_set_state(self.h, int(token), int(pos)), "kxm_set_state")

    def decode(self, steps, use_graph=True):
        hcheck(hlib().kxm_decode(self.h, int(steps), 1 if use_graph else 0), "kxm_decode")

    def prefill(self, tokens, pos0=0, exact=False):
        t = np.ascontiguousarray(tokens, np.int32)
        out = np.empty(self.cfg.vocab_size, np.float32)
        hcheck(hlib().kxm_prefill(self.h, _p(t), t.size, int(pos0), _p(out), 1 if exact else 0),
               "kxm_prefill")
        return out

    def history(self, n):
        out = np.empty(n, np.int32)
        hcheck(hlib().kxm_read_history(self.h, _p(out), n), "kxm_read_history")
        return out

    def state(self):
        t, p = _i32(), _i32()
        hcheck(hlib().kxm_read_state(self.h, C.byref(t), C.byref(p)), "kxm_read_state")
        return t.value, p.value

    def read_kv(self, layer, rows):
        k = np.empty((rows, self.kv_dim_local), np.float32)
        v = np.empty((rows, self.kv_dim_local), np.float32)
        hcheck(hlib().kxm_read_kv(self.h, layer, rows, _p(k), _p(v)), "kxm_read_kv")
        return k, v

    def sync(self):
        hcheck(hlib().kxm_sync(self.h), "kxm_sync")

    def profile_gemv(self, steps):
        n, b, ms = C.c_int64(), C.c_double(), C.c_double()
        hcheck(hlib().kxm_profile_gemv(self.h, steps, C.byref(n), C.byref(b), C.byref(ms)),
               "kxm_profile_gemv")
        return n.value, b.value, ms.value

    def profile_allreduce(self, count, iters):
        ms = C.c_double()
        hcheck(hlib().kxm_profile_allreduce(self.h, C.c_uint64(count), iters, C.byref(ms)),
               "kxm_profile_allreduce")
        return ms.value

    def linear_bytes_per_token(self):
        return hlib().kxm_linear_bytes_per_token(self.h)

    def stream(self):
        return hlib().kxm_stream(self.h)


SHARD_NAMES = ("wq", "wk", "wv", "wo", "w1", "w2", "w3")


def shard_plan(dim, hidden_dim, head_num, kv_head_num, tp_size, tp_rank):
    """The rank's slice of every layer matrix as {name: (row0, rows, col0, cols)} plus whether
    the rank carries the residual.  Host logic of the library itself; needs no GPU."""
    out = (C.c_int32 * 28)()
    res = C.c_int32(0)
    L = hlib()
    L.kxm_shard_plan.restype = C.c_int
    L.kxm_shard_plan.argtypes = [_i32] * 6 + [_vp, _vp]
    rc = L.kxm_shard_plan(dim, hidden_dim, head_num, kv_head_num, tp_size, tp_rank, out,
                          C.byref(res))
    if rc != 0:
        raise ValueError("geometry does not divide over tp_size ranks")
    plan = {n: tuple(out[4 * i:4 * i + 4]) for i, n in enumerate(SHARD_NAMES)}
    return plan, bool(res.value)


def loopback_create(ranks):
    L = hlib()
    L.kxm_loopback_create.restype = C.c_void_p
    L.kxm_loopback_create.argtypes = [_i32]
    L.kxm_join_loopback.argtypes = [_vp, _vp]
    L.kxm_loopback_destroy.argtypes = [_vp]
    L.kxm_loopback_destroy.restype = None
    return L.kxm_loopback_create(ranks)


def loopback_destroy(group):
    hlib().kxm_loopback_destroy(group)


def comm_unique_id():
    buf = np.zeros(128, np.uint8)
    hcheck(hlib().kxm_comm_unique_id(_p(buf)), "kxm_comm_unique_id")
    return buf.tobytes()


# ---- layer probes ----
def matmul_layer(x, w, scales=None, group_size=0, bias=None):
    x = np.ascontiguousarray(x, np.float32)
    rows = x.shape[0] if x.ndim == 2 else 1
    dim0, dim1 = w.shape
    w = np.ascontiguousarray(w, np.int8 if group_size else np.float32)
    s = np.ascontiguousarray(scales, np.float32) if scales is not None else None
    b = np.ascontiguousarray(bias, np.float32) if bias is not None else None
    y = np.empty((rows, dim0) if x.ndim == 2 else (dim0,), np.float32)
    hcheck(hlib().kxm_test_matmul_layer(_p(x), _p(w), _p(s), group_size, _p(b), dim0, dim1, rows,
                                        _p(y)), "MatmulLayer::forward")
    return y


def mha_layer(q, kc, vc, head_num, kv_mul, head_size, layer_idx, pos):
    L, S, _ = kc.shape
    out = np.empty(head_num * head_size, np.float32)
    hcheck(hlib().kxm_test_mha_layer(_p(np.ascontiguousarray(q, np.float32)),
                                     _p(np.ascontiguousarray(kc, np.float32)),
                                     _p(np.ascontiguousarray(vc, np.float32)), L, S, head_num,
                                     kv_mul, head_size, layer_idx, pos, _p(out)),
           "MultiHeadAttention::forward")
    return out.reshape(head_num, head_size)


def mha_layer_no_workspace(q, kc, vc, head_num, kv_mul, head_size, layer_idx, pos):
    L, S, _ = kc.shape
    out = np.empty(head_num * head_size, np.float32)
    hcheck(hlib().kxm_test_mha_layer_no_workspace(_p(np.ascontiguousarray(q, np.float32)),
                                                  _p(np.ascontiguousarray(kc, np.float32)),
                                                  _p(np.ascontiguousarray(vc, np.float32)), L, S,
                                                  head_num, kv_mul, head_size, layer_idx, pos,
                                                  _p(out)),
           "MultiHeadAttention::forward (no workspace)")
    return out.reshape(head_num, head_size)


def mha_from_registry_pieces(q, kc, vc, head_num, kv_mul, head_size, layer_idx, pos):
    """(attention rows, the same rows through the scale kernel with factor 2)."""
    L, S, _ = kc.shape
    out = np.empty(2 * head_num * head_size, np.float32)
    hcheck(hlib().kxm_test_mha_from_registry_pieces(_p(np.ascontiguousarray(q, np.float32)),
                                                    _p(np.ascontiguousarray(kc, np.float32)),
                                                    _p(np.ascontiguousarray(vc, np.float32)), L, S,
                                                    head_num, kv_mul, head_size, layer_idx, pos, _p(out)),
           "attention from registry pieces")
    a, b = out.reshape(2, head_num, head_size)
    return a, b


def flash_layer(q, kc, vc, head_num, kv_mul, head_size, layer_idx, pos):
    L, S, _ = kc.shape
    q = np.ascontiguousarray(q, np.float32).reshape(-1, head_num, head_size)
    out = np.empty_like(q)
    hcheck(hlib().kxm_test_flash_layer(_p(q), _p(np.ascontiguousarray(kc, np.float32)),
                                       _p(np.ascontiguousarray(vc, np.float32)), L, S, head_num,
                                       kv_mul, head_size, layer_idx, pos, q.shape[0], _p(out)),
           "FlashAttention::forward")
    return out


def kv_layout(layers, seq_len, kv_head_num, head_size, layer_idx):
    k, v, t = C.c_int64(), C.c_int64(), C.c_int64()
    hcheck(hlib().kxm_test_kv_layout(layers, seq_len, kv_head_num, head_size, layer_idx,
                                     C.byref(k), C.byref(v), C.byref(t)), "KVCacheManager")
    return k.value, v.value, t.value
