"""ctypes view of the CPU oracle (oracle/kx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkx_oracle.so")

ROPE_INTERLEAVED = 0
ROPE_HALF_SPLIT = 1


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "kx_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("hidden_dim", C.c_int32), ("layer_num", C.c_int32),
                ("head_num", C.c_int32), ("kv_head_num", C.c_int32), ("vocab_size", C.c_int32),
                ("seq_len", C.c_int32), ("group_size", C.c_int32), ("rope_variant", C.c_int32),
                ("rope_theta", C.c_float), ("rms_eps", C.c_float)]


_MATS = ["wq", "wk", "wv", "wo", "w1", "w2", "w3", "cls"]


class Weights(C.Structure):
    _fields_ = ([("cfg", Config)] + [(m + "_f", C.c_void_p) for m in _MATS] +
                [(m + "_q", C.c_void_p) for m in _MATS] + [(m + "_s", C.c_void_p) for m in _MATS] +
                [("emb", C.c_void_p), ("attn_norm", C.c_void_p), ("ffn_norm", C.c_void_p),
                 ("final_norm", C.c_void_p), ("bq", C.c_void_p), ("bk", C.c_void_p),
                 ("bv", C.c_void_p)])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.kxo_state_new.restype = C.c_void_p
        L.kxo_state_new.argtypes = [C.POINTER(Config)]
        L.kxo_state_free.argtypes = [C.c_void_p]
        for f in ("kxo_state_key_cache", "kxo_state_value_cache", "kxo_state_logits"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [C.c_void_p]
        L.kxo_forward_token.argtypes = [C.POINTER(Weights), C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32]
        L.kxo_argmax.restype = C.c_int32
        L.kxo_parse_legacy_f32.restype = C.c_int
        L.kxo_parse_legacy_q8.restype = C.c_int
        L.kxo_parse_qwen2_f32.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def quantize_q80(w, group_size=64):
    w = _f32(w)
    q = np.empty(w.shape, np.int8)
    s = np.empty(w.size // group_size, np.float32)
    lib().kxo_quantize_q80(_p(w), C.c_size_t(w.size), C.c_int32(group_size), _p(q), _p(s))
    return q, s


def gemv_w8(x, w, s, group_size=64, threads=1):
    x = _f32(x)
    w = np.ascontiguousarray(w, np.int8)
    s = _f32(s)
    K, M = w.shape
    y = np.empty(K, np.float32)
    lib().kxo_gemv_w8(_p(x), _p(w), _p(s), C.c_int32(group_size), _p(y), C.c_int32(M), C.c_int32(K),
                      C.c_int32(threads))
    return y


def gemv_f32(x, w, scale=1.0, threads=1):
    x = _f32(x)
    w = _f32(w)
    K, M = w.shape
    y = np.empty(K, np.float32)
    lib().kxo_gemv_f32(_p(x), _p(w), _p(y), C.c_int32(M), C.c_int32(K), C.c_float(scale),
                       C.c_int32(threads))
    return y


def gemm_w8(X, w, s, group_size=64, threads=1):
    X = _f32(X)
    w = np.ascontiguousarray(w, np.int8)
    s = _f32(s)
    K, M = w.shape
    T = X.shape[0]
    Y = np.empty((T, K), np.float32)
    lib().kxo_gemm_w8(_p(X), _p(w), _p(s), C.c_int32(group_size), _p(Y), C.c_int32(T), C.c_int32(M),
                      C.c_int32(K), C.c_int32(threads))
    return Y


def gemm_f32(X, w, threads=1):
    X = _f32(X)
    w = _f32(w)
    K, M = w.shape
    T = X.shape[0]
    Y = np.empty((T, K), np.float32)
    lib().kxo_gemm_f32(_p(X), _p(w), _p(Y), C.c_int32(T), C.c_int32(M), C.c_int32(K),
                       C.c_int32(threads))
    return Y


def mha(pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size, query, key_cache,
        value_cache, want_score=False):
    """key_cache/value_cache: whole [L][seq_len][kv_dim] arrays."""
    q = _f32(query)
    kc = _f32(key_cache)
    vc = _f32(value_cache)
    out = np.empty(head_num * head_size, np.float32)
    score = np.zeros((head_num, seq_len), np.float32) if want_score else None
    lib().kxo_mha(C.c_int32(pos), C.c_int32(head_num), C.c_int32(layer_index), C.c_int32(seq_len),
                  C.c_int32(kv_dim), C.c_int32(kv_mul), C.c_int32(head_size), _p(out), _p(q),
                  _p(score), _p(kc), _p(vc))
    return (out, score) if want_score else out


def flash_attention(query, key_cache, value_cache, head_num, kv_head_num, head_size, seq_len, pos,
                    layer_idx, softmax_scale=None, is_causal=True, threads=1):
    q = _f32(query).reshape(-1, head_num, head_size)
    Tq = q.shape[0]
    kc = _f32(key_cache)
    vc = _f32(value_cache)
    if softmax_scale is None:
        softmax_scale = 1.0 / np.sqrt(np.float32(head_size))
    out = np.empty_like(q)
    lib().kxo_flash_attention(_p(q), _p(kc), _p(vc), _p(out), C.c_int32(Tq), C.c_int32(head_num),
                              C.c_int32(kv_head_num), C.c_int32(head_size), C.c_int32(seq_len),
                              C.c_int32(pos), C.c_int32(layer_idx), C.c_float(softmax_scale),
                              C.c_int32(1 if is_causal else 0), C.c_int32(threads))
    return out


def rmsnorm(x, w, eps):
    x = _f32(x)
    w = _f32(w)
    out = np.empty_like(x)
    lib().kxo_rmsnorm(_p(x), _p(w), _p(out), C.c_int32(x.size), C.c_float(eps))
    return out


def sin_cos_cache(head_size, max_seq_len, theta):
    s = np.empty((max_seq_len, head_size), np.float32)
    c = np.empty((max_seq_len, head_size), np.float32)
    lib().kxo_sin_cos_cache(C.c_int32(head_size), C.c_int32(max_seq_len), C.c_float(theta), _p(s),
                            _p(c))
    return s, c


def rope(variant, dim, kv_dim, head_size, q, k, pos, sin_cache, cos_cache):
    q = _f32(q).copy()
    k = _f32(k).copy()
    lib().kxo_rope(C.c_int32(variant), C.c_int32(dim), C.c_int32(kv_dim), C.c_int32(head_size),
                   _p(q), _p(k), C.c_int32(pos), _p(_f32(sin_cache)), _p(_f32(cos_cache)))
    return q, k


def swiglu(a, b):
    a = _f32(a)
    b = _f32(b)
    out = np.empty_like(a)
    lib().kxo_swiglu(_p(a), _p(b), _p(out), C.c_int32(a.size))
    return out


def add(a, b):
    a = _f32(a)
    b = _f32(b)
    out = np.empty_like(a)
    lib().kxo_add(_p(a), _p(b), _p(out), C.c_int32(a.size))
    return out


def embedding(tokens, table):
    t = np.ascontiguousarray(tokens, np.int32)
    table = _f32(table)
    out = np.empty((t.size, table.shape[1]), np.float32)
    lib().kxo_embedding(_p(t), C.c_int32(t.size), _p(table), C.c_int32(table.shape[1]), _p(out))
    return out


def argmax(logits):
    l = _f32(logits)
    return int(lib().kxo_argmax(_p(l), C.c_int32(l.size)))


class Model:
    """Reference token loop over oracle kernels.  `arrays` maps names to numpy arrays:
    fp32: wq..w3 [L,K,M], cls [V,dim]; quant: wq_q/wq_s ...; emb, attn_norm, ffn_norm, final_norm."""

    def __init__(self, cfg: Config, arrays=None, blob=None, quant=False, qkv_bias=False):
        self.cfg = cfg
        self._keep = []
        self.w = Weights()
        if blob is not None:
            self._blob = np.frombuffer(blob, np.uint8).copy()
            if qkv_bias and quant:
                raise ValueError("the quant layout carries no biases; attach them with set_bias()")
            fn = (lib().kxo_parse_legacy_q8 if quant else
                  lib().kxo_parse_qwen2_f32 if qkv_bias else lib().kxo_parse_legacy_f32)
            rc = fn(_p(self._blob), C.c_size_t(self._blob.size), C.byref(self.w))
            if rc != 0:
                raise ValueError(f"oracle model parse failed rc={rc}")
            # runtime knobs that the file does not carry
            self.w.cfg.rope_variant = cfg.rope_variant
            self.w.cfg.rope_theta = cfg.rope_theta
            self.w.cfg.rms_eps = cfg.rms_eps
            self.cfg = self.w.cfg
        else:
            self.w.cfg = cfg
            for name, arr in arrays.items():
                dt = np.int8 if name.endswith("_q") else np.float32
                a = np.ascontiguousarray(arr, dt)
                self._keep.append(a)
                field = name
                if field in _MATS:  # bare name = fp32 matrix
                    field = field + "_f"
                setattr(self.w, field, a.ctypes.data)
        self.st = lib().kxo_state_new(C.byref(self.w.cfg))

    def __del__(self):
        try:
            lib().kxo_state_free(self.st)
        except Exception:
            pass

    def set_bias(self, bq, bk, bv):
        """q/k/v biases [L][rows] added after the matmuls (the quant files carry none)."""
        for name, arr in (("bq", bq), ("bk", bk), ("bv", bv)):
            a = np.ascontiguousarray(arr, np.float32)
            self._keep.append(a)
            setattr(self.w, name, a.ctypes.data)

    def forward(self, token, pos, threads=1):
        lib().kxo_forward_token(C.byref(self.w), self.st, C.c_int32(int(token)), C.c_int32(int(pos)),
                                C.c_int32(threads))
        ptr = lib().kxo_state_logits(self.st)
        return np.ctypeslib.as_array(ptr, shape=(self.w.cfg.vocab_size,)).copy()

    def caches(self):
        c = self.w.cfg
        kvd = c.dim * c.kv_head_num // c.head_num
        shape = (c.layer_num, c.seq_len, kvd)
        k = np.ctypeslib.as_array(lib().kxo_state_key_cache(self.st), shape=shape)
        v = np.ctypeslib.as_array(lib().kxo_state_value_cache(self.st), shape=shape)
        return k, v
