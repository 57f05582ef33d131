"""ctypes binding of libkx_hip.so (the C ABI declared in include/kx_hip.h).

This is FFI plumbing for the tests and the benchmark driver: numpy arrays in, device buffers
handled through the library's own allocator.  There is no CPU fallback anywhere in this
package: if the shared library is missing or a call fails, a KxError is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# KX_LIB_DIR selects another build of both libraries (lib_bounds: the bounds-checking build)
LIB_DIR = os.environ.get("KX_LIB_DIR") or os.path.join(os.path.dirname(_PKG), "lib")
LIB_PATH = os.environ.get("KX_HIP_LIB") or os.path.join(LIB_DIR, "libkx_hip.so")

ROPE_INTERLEAVED = 0
ROPE_HALF_SPLIT = 1


class KxError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); mirrors include/kx_hip.h one to one
_vp, _i32, _sz, _f = C.c_void_p, C.c_int32, C.c_size_t, C.c_float
SIGNATURES = {
    "kx_error_string": (C.c_char_p, [C.c_int]),
    "kx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kx_set_device": (C.c_int, [C.c_int]),
    "kx_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "kx_device_name": (C.c_int, [C.c_char_p, _sz]),
    "kx_malloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "kx_free": (C.c_int, [_vp]),
    "kx_memset": (C.c_int, [_vp, C.c_int, _sz, _vp]),
    "kx_memcpy_h2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "kx_memcpy_d2h": (C.c_int, [_vp, _vp, _sz, _vp]),
    "kx_memcpy_d2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "kx_mem_info": (C.c_int, [C.POINTER(_sz), C.POINTER(_sz)]),
    "kx_host_alloc_pinned": (C.c_int, [C.POINTER(_vp), _sz]),
    "kx_host_free_pinned": (C.c_int, [_vp]),
    "kx_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "kx_stream_destroy": (C.c_int, [_vp]),
    "kx_stream_sync": (C.c_int, [_vp]),
    "kx_device_sync": (C.c_int, []),
    "kx_event_create": (C.c_int, [C.POINTER(_vp)]),
    "kx_event_destroy": (C.c_int, [_vp]),
    "kx_event_record": (C.c_int, [_vp, _vp]),
    "kx_event_sync": (C.c_int, [_vp]),
    "kx_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(_f)]),
    "kx_stream_wait_event": (C.c_int, [_vp, _vp]),
    "kx_graph_begin_capture": (C.c_int, [_vp]),
    "kx_graph_end_capture": (C.c_int, [_vp, C.POINTER(_vp)]),
    "kx_graph_launch": (C.c_int, [_vp, _vp]),
    "kx_graph_destroy": (C.c_int, [_vp]),
    "kx_bounds_enabled": (C.c_int, []),
    "kx_bounds_report": (C.c_int, [_vp, _i32, C.POINTER(C.c_uint64)]),
    "kx_bounds_reset": (C.c_int, []),
    "kx_gemv_w8_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "kx_gemv_w8_f32_ex": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32,
                                    _i32, _vp]),
    "kx_gemv_w8_swiglu_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp]),
    "kx_gemv_w8_chain_workspace_bytes": (_sz, [_i32]),
    "kx_gemv_w8_chain_reset": (C.c_int, [_vp, _sz, _vp]),
    "kx_gemv_w8_chain_status": (C.c_int, [_vp, C.POINTER(_i32)]),
    "kx_gemv_w8_chain_f32": (C.c_int, [_vp, _i32, _vp, _sz, _i32, _vp, _i32, _vp]),
    "kx_gemv_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _vp]),
    "kx_gemm_w8_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "kx_gemm_w8_f32_ex": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _vp, C.c_int64, _vp,
                                    C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32, _vp]),
    "kx_gemm_w8_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "kx_gemm_w8_f32_ws": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _vp, C.c_int64, _vp,
                                    C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32, _vp, _sz,
                                    _vp]),
    "kx_gemm_w8_f16act_workspace_bytes": (_sz, [_i32, _i32]),
    "kx_gemm_w8_f16act_ws": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, C.c_int64, _vp, C.c_int64,
                                       _vp, C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32,
                                       _vp, _sz, _vp]),
    "kx_gemm_split_workspace_bytes": (_sz, []),
    "kx_gemm_w8_f16act_qkv_rope": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, C.c_int64, _vp, C.c_int64, _vp,
                                             C.c_int64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32,
                                             _i32, _vp, _sz, _vp, _sz, _vp]),
    "kx_gemm_w8_f16act_split_ws": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, C.c_int64, _vp, C.c_int64,
                                             _vp, C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32,
                                             _vp, _sz, _vp, _sz, _vp]),
    "kx_gemm_w8_rows_f32": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _vp, C.c_int64, _vp,
                                      C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "kx_gemm_w8_rows_f16act": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _vp, C.c_int64, _vp,
                                         C.c_int64, _i32, _i32, _vp, C.c_int64, _i32, _i32, _i32, _vp, _sz,
                                         _vp]),
    "kx_gemm_w8_rows_swiglu_f16act": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _i32, _i32, _i32,
                                                _vp, _sz, _vp]),
    "kx_gemm_w8_rows_qkv_rope_f16act": (C.c_int, [_vp, C.c_int64, _vp, _vp, _i32, _vp, C.c_int64, _vp, C.c_int64,
                                                  _vp, C.c_int64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp,
                                                  _i32, _i32, _vp, _sz, _vp]),
    "kx_rmsnorm_f16out": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _vp]),
    "kx_flash_attention_f16out": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                            _i32, _f, _i32, _vp, _sz, _vp]),
    "kx_gemm_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "kx_attn_workspace_bytes": (_sz, [_i32, _i32]),
    "kx_attn_workspace_reset": (C.c_int, [_vp, _sz, _vp]),
    "kx_flash_attention_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "kx_mha_decode_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                    _vp, _sz, _vp]),
    "kx_mha_decode_scores_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                           _i32, _vp, _sz, _vp]),
    "kx_mha_decode_static_limit_bytes": (_sz, []),
    "kx_mha_decode_static_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                           _i32, _vp]),
    "kx_mha_decode_dpos_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                         _i32, _i32, _vp, _sz, _vp]),
    "kx_mha_decode_dpos_ex_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                               _i32, _i32, _vp, _sz, _vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    "kx_gemv_w8_block_rows": (_i32, [_i32, _i32, _i32]),
    "kx_mha_decode_kv16_ex": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                           _sz, _vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    "kx_flash_attention_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                         _i32, _f, _i32, _vp, _sz, _vp]),
    "kx_mha_decode_kv16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                     _sz, _vp]),
    "kx_flash_attention_kv16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                          _f, _i32, _vp, _sz, _vp]),
    "kx_f32_to_f16_rows": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _i32, _i32, _vp]),
    "kx_rmsnorm_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f, _vp]),
    "kx_sin_cos_cache_f32": (C.c_int, [_i32, _i32, _f, _vp, _vp, _vp]),
    "kx_rope_f32": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "kx_rope_bias_f32": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp,
                                   _i32, _i32, _vp, _vp, _vp]),
    "kx_swiglu_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "kx_swiglu_rows_f32": (C.c_int, [_vp, C.c_int64, _i32, _vp, C.c_int64, _i32, _vp]),
    "kx_swiglu_rows_f16out": (C.c_int, [_vp, C.c_int64, _i32, _vp, C.c_int64, _i32, _vp]),
    "kx_add_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "kx_embedding_f32": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "kx_embedding_dpos_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "kx_argmax_f32": (C.c_int, [_vp, _i32, _vp, _vp]),
    "kx_scale_f32": (C.c_int, [_f, _vp, _i32, _vp]),
    "kx_softmax_f32": (C.c_int, [_vp, _i32, _vp]),
    "kx_scale_sum_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "kx_quantize_q80_f32": (C.c_int, [_vp, _sz, _i32, _vp, _vp, _vp]),
    "kx_step_workspace_bytes": (_sz, []),
    "kx_decode_step_end": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "kx_gemv_w8_fused_f32": (C.c_int, [_vp, _vp]),
    "kx_gemv_w8_classify_pick_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "kx_decode_step_end_embed": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "kx_argmax_candidate_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "kx_decode_step_end_candidates_embed": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "kx_decode_step_end_candidates": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "kx_touch_bytes": (C.c_int, [_vp, _sz, _i32, _vp, _vp]),
    "kx_fill_hash_i8": (C.c_int, [_vp, _i32, _i32, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, _vp]),
    "kx_fill_hash_f32": (C.c_int, [_vp, _i32, _i32, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, _f,
                                   _f, _vp]),
    "kx_comm_unique_id": (C.c_int, [_vp]),
    "kx_comm_init_rank": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32]),
    "kx_comm_destroy": (C.c_int, [_vp]),
    "kx_comm_allreduce_sum_f32": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "kx_comm_broadcast_f32": (C.c_int, [_vp, _vp, _vp, _sz, _i32, _vp]),
    "kx_comm_allgather_f32": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "kx_peer_create": (C.c_int, [C.POINTER(_vp), _i32, _i32, _sz]),
    "kx_peer_handle": (C.c_int, [_vp, _vp]),
    "kx_peer_connect": (C.c_int, [_vp, _vp]),
    "kx_peer_allreduce_sum_f32": (C.c_int, [_vp, _vp, _sz, _vp]),
    "kx_peer_allgather_f32": (C.c_int, [_vp, _vp, _sz, _vp]),
    "kx_peer_timed_out": (C.c_int, [_vp, C.POINTER(_i32)]),
    "kx_peer_destroy": (C.c_int, [_vp]),
}


class GemvDesc(C.Structure):
    """kx_gemv_w8_desc of include/kx_hip.h, field for field."""
    _fields_ = [("mode", C.c_int32), ("group_size", C.c_int32), ("M", C.c_int32), ("K", C.c_int32),
                ("x", C.c_void_p), ("w", C.c_void_p), ("scales", C.c_void_p), ("norm_w", C.c_void_p),
                ("norm_eps", C.c_float), ("y0", C.c_void_p), ("y1", C.c_void_p), ("y2", C.c_void_p),
                ("k0", C.c_int32), ("k1", C.c_int32), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("sin_cache", C.c_void_p), ("cos_cache", C.c_void_p), ("pos_dev", C.c_void_p),
                ("pos", C.c_int32), ("rope_variant", C.c_int32), ("head_size", C.c_int32),
                ("kv_pos_stride", C.c_int32), ("kv_f16", C.c_int32)]


GEMV_PLAIN, GEMV_GATE_UP, GEMV_QKV_ROPE = 0, 1, 2


class BoundsRecord(C.Structure):
    """kx_bounds_record of include/kx_hip.h."""
    _fields_ = [("file_id", C.c_int32), ("line", C.c_int32), ("operand", C.c_int32), ("pad_", C.c_int32),
                ("offset", C.c_int64), ("extent", C.c_int64)]


BOUNDS_FILES = {1: "gemv_w8.hip", 2: "gemm_w8_rows.hip", 3: "gemm_w8.hip", 4: "gemm_f16.hip",
                5: "attn_decode.hip", 6: "attn_prefill.hip", 7: "support.hip", 8: "gemv_f32.hip",
                9: "gemv_engine.hip"}


def bounds_enabled():
    return bool(lib().kx_bounds_enabled())


def bounds_report(reset=False):
    """(total violations, [(file, line, operand, offset, extent), ...]) of a bounds-checking build."""
    recs = (BoundsRecord * 1024)()
    total = C.c_uint64(0)
    check(lib().kx_bounds_report(recs, 1024, C.byref(total)), "kx_bounds_report")
    out = [(BOUNDS_FILES.get(r.file_id, str(r.file_id)), r.line, r.operand, r.offset, r.extent)
           for r in recs[:min(total.value, 1024)]]
    if reset:
        check(lib().kx_bounds_reset(), "kx_bounds_reset")
    return total.value, out


def lib():
    """Loads libkx_hip.so; raises KxError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KxError(f"{LIB_PATH} is missing: build it with `make -C llm-inference-engine_amd` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().kx_error_string(rc).decode()
        raise KxError(f"{what} failed: rc={rc} ({msg})")


class Buf:
    """Owning device buffer from kx_malloc."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().kx_malloc(C.byref(p), max(self.nbytes, 1)), "kx_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.dtype, b.shape = a.dtype, a.shape
        if a.nbytes:
            check(lib().kx_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, None), "h2d")
        return b

    @classmethod
    def zeros(cls, shape, dtype=np.float32):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        b = cls(n)
        b.dtype, b.shape = np.dtype(dtype), tuple(np.atleast_1d(shape))
        check(lib().kx_memset(b.ptr, 0, n, None), "memset")
        check(lib().kx_device_sync(), "sync")
        return b

    def upload(self, a, byte_offset=0):
        a = np.ascontiguousarray(a)
        assert byte_offset + a.nbytes <= self.nbytes
        check(lib().kx_memcpy_h2d(self.ptr + byte_offset, a.ctypes.data, a.nbytes, None), "h2d")

    def numpy(self, dtype=None, shape=None):
        dtype = np.dtype(dtype or getattr(self, "dtype", np.uint8))
        shape = shape if shape is not None else getattr(self, "shape", (self.nbytes // dtype.itemsize,))
        out = np.empty(shape, dtype)
        check(lib().kx_device_sync(), "sync")
        if out.nbytes:
            check(lib().kx_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "d2h")
        return out

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def free(self):
        if self.ptr:
            lib().kx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_name():
    b = C.create_string_buffer(256)
    check(lib().kx_device_name(b, 256), "kx_device_name")
    return b.value.decode()


def sync():
    check(lib().kx_device_sync(), "kx_device_sync")


class Stream:
    def __init__(self):
        p = C.c_void_p()
        check(lib().kx_stream_create(C.byref(p)), "kx_stream_create")
        self.ptr = p.value

    def sync(self):
        check(lib().kx_stream_sync(self.ptr), "kx_stream_sync")

    def __del__(self):
        try:
            lib().kx_stream_destroy(self.ptr)
        except Exception:
            pass


def time_graph(stream, record, replays=20):
    """Captures the launches that `record()` issues on `stream` into a HIP graph, replays the graph
    `replays` times and returns microseconds per replay (HIP events; no host launch cost inside)."""
    g = C.c_void_p()
    record()  # eager once: lazy per-kernel setup must not happen during capture
    stream.sync()
    check(lib().kx_graph_begin_capture(stream.ptr), "begin_capture")
    record()
    check(lib().kx_graph_end_capture(stream.ptr, C.byref(g)), "end_capture")
    check(lib().kx_graph_launch(g, stream.ptr), "graph_launch")
    stream.sync()
    e0, e1 = Event(), Event()
    e0.record(stream.ptr)
    for _ in range(replays):
        lib().kx_graph_launch(g, stream.ptr)
    e1.record(stream.ptr)
    e1.sync()
    us = e0.elapsed_ms(e1) * 1e3 / replays
    lib().kx_graph_destroy(g)
    return us


class Event:
    def __init__(self):
        p = C.c_void_p()
        check(lib().kx_event_create(C.byref(p)), "kx_event_create")
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().kx_event_record(self.ptr, stream), "kx_event_record")

    def sync(self):
        check(lib().kx_event_sync(self.ptr), "kx_event_sync")

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().kx_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)), "kx_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib().kx_event_destroy(self.ptr)
        except Exception:
            pass


# ---------------- numpy-in / numpy-out conveniences used by the parity tests ----------------
def gemv_w8(x, w, s, group_size=64):
    K, M = w.shape
    dx, dw, ds = Buf.from_numpy(x.astype(np.float32)), Buf.from_numpy(w.astype(np.int8)), \
        Buf.from_numpy(s.astype(np.float32))
    dy = Buf.zeros(K)
    check(lib().kx_gemv_w8_f32(dx.ptr, dw.ptr, ds.ptr, group_size, dy.ptr, M, K, None),
          "kx_gemv_w8_f32")
    return dy.numpy(np.float32, (K,))


def gemv_f32(x, w, scale=1.0):
    K, M = w.shape
    dx, dw = Buf.from_numpy(x.astype(np.float32)), Buf.from_numpy(w.astype(np.float32))
    dy = Buf.zeros(K)
    check(lib().kx_gemv_f32(dx.ptr, dw.ptr, dy.ptr, M, K, scale, None), "kx_gemv_f32")
    return dy.numpy(np.float32, (K,))


def mha_decode(q, kc, vc, pos, head_num, kv_mul, head_size, layer_index, workspace_bytes=None,
               kv16=False):
    """kv16: the caches are uploaded as IEEE half (the 16-bit KV cache form)."""
    L, S, kv_dim = kc.shape
    kvt = np.float16 if kv16 else np.float32
    dq, dk, dv = Buf.from_numpy(q.astype(np.float32)), Buf.from_numpy(kc.astype(kvt)), \
        Buf.from_numpy(vc.astype(kvt))
    out = Buf.zeros(head_num * head_size)
    if workspace_bytes is None:
        workspace_bytes = max(lib().kx_attn_workspace_bytes(head_num, head_size),
                              16384 + head_num * S * 4)
    ws = Buf.zeros(workspace_bytes, np.uint8) if workspace_bytes else None
    if kv16:
        check(lib().kx_mha_decode_kv16(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos, None, head_num, kv_mul,
                                       head_size, kv_dim, S, layer_index, ws.ptr if ws else None,
                                       workspace_bytes, None), "kx_mha_decode_kv16")
        return out.numpy(np.float32, (head_num, head_size))
    check(lib().kx_mha_decode_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, pos, head_num, kv_mul, head_size,
                                  kv_dim, S, layer_index, ws.ptr if ws else None, workspace_bytes,
                                  None), "kx_mha_decode_f32")
    return out.numpy(np.float32, (head_num, head_size))


def flash_attention(q, kc, vc, head_num, kv_head_num, head_size, pos, layer_idx, softmax_scale=None,
                    is_causal=True, f16_copies=True, f16_out=False, kv16=False):
    L, S, kv_dim = kc.shape
    q = np.ascontiguousarray(q, np.float32).reshape(-1, head_num, head_size)
    Tq = q.shape[0]
    if softmax_scale is None:
        softmax_scale = float(1.0 / np.sqrt(np.float32(head_size)))
    kvt = np.float16 if kv16 else np.float32
    dq, dk, dv = Buf.from_numpy(q), Buf.from_numpy(kc.astype(kvt)), Buf.from_numpy(vc.astype(kvt))
    out = Buf.zeros(q.size)
    wsb = max(lib().kx_attn_workspace_bytes(head_num, head_size), 16384 + head_num * S * 4)
    if f16_copies:  # large enough for the fp16 K/V copies of the prefill path
        wsb = max(wsb, lib().kx_flash_attention_workspace_bytes(pos, kv_head_num, head_size))
    else:
        wsb = min(wsb, lib().kx_flash_attention_workspace_bytes(pos, kv_head_num, head_size) - 1)
    ws = Buf.zeros(wsb, np.uint8)
    if kv16:
        out16 = Buf.zeros(q.size, np.float16) if f16_out else None
        check(lib().kx_flash_attention_kv16(dq.ptr, dk.ptr, dv.ptr, None if f16_out else out.ptr,
                                            out16.ptr if f16_out else None, Tq, head_num, kv_head_num,
                                            head_size, S, pos, layer_idx, softmax_scale,
                                            1 if is_causal else 0, ws.ptr, wsb, None),
              "kx_flash_attention_kv16")
        return out16.numpy(np.float16, q.shape) if f16_out else out.numpy(np.float32, q.shape)
    if f16_out:
        out16 = Buf.zeros(q.size, np.float16)
        check(lib().kx_flash_attention_f16out(dq.ptr, dk.ptr, dv.ptr, out16.ptr, Tq, head_num,
                                              kv_head_num, head_size, S, pos, layer_idx, softmax_scale,
                                              1 if is_causal else 0, ws.ptr, wsb, None),
              "kx_flash_attention_f16out")
        return out16.numpy(np.float16, q.shape)
    check(lib().kx_flash_attention_f32(dq.ptr, dk.ptr, dv.ptr, out.ptr, Tq, head_num, kv_head_num,
                                       head_size, S, pos, layer_idx, softmax_scale,
                                       1 if is_causal else 0, ws.ptr, wsb, None),
          "kx_flash_attention_f32")
    return out.numpy(np.float32, q.shape)


def gemm_w8_f16act(X16, w, s, group_size=64, swiglu=False, split_ws=None, residual=None):
    """fp16 activations [T][M]; fp32 [T][K] out (+ residual [T][K]), or fp16 silu(gate)*up [T][K/2]
    with swiglu.
    split_ws: a zero-filled Buf of kx_gemm_split_workspace_bytes() lets the kernel cut reductions."""
    K, M = w.shape
    T = X16.shape[0]
    dx, dw, ds = Buf.from_numpy(X16.astype(np.float16)), Buf.from_numpy(w.astype(np.int8)), \
        Buf.from_numpy(s.astype(np.float32))
    wsb = lib().kx_gemm_w8_f16act_workspace_bytes(M, K)
    ws = Buf(wsb)
    sp, spb = (split_ws.ptr, split_ws.nbytes) if split_ws is not None else (None, 0)
    if swiglu:
        h = Buf.zeros((T, K // 2), np.float16)
        check(lib().kx_gemm_w8_f16act_split_ws(dx.ptr, dw.ptr, ds.ptr, group_size, 1, h.ptr, None, 0, None,
                                               0, None, 0, 0, 0, None, 0, T, M, K, ws.ptr, wsb, sp, spb, None),
              "kx_gemm_w8_f16act_split_ws")
        return h.numpy(np.float16, (T, K // 2))
    dy = Buf.zeros((T, K))
    dr = Buf.from_numpy(np.ascontiguousarray(residual, np.float32)) if residual is not None else None
    check(lib().kx_gemm_w8_f16act_split_ws(dx.ptr, dw.ptr, ds.ptr, group_size, 0, None, dy.ptr, K, None, 0,
                                           None, 0, K, K, dr.ptr if dr else None, K if dr else 0, T, M, K,
                                           ws.ptr, wsb, sp, spb, None),
          "kx_gemm_w8_f16act_split_ws")
    return dy.numpy(np.float32, (T, K))


def rmsnorm_f16out(x, weight, eps=1e-5):
    x = np.ascontiguousarray(x, np.float32)
    rows, dim = x.shape
    dx, dw = Buf.from_numpy(x), Buf.from_numpy(weight.astype(np.float32))
    out = Buf.zeros((rows, dim), np.float16)
    check(lib().kx_rmsnorm_f16out(dx.ptr, dw.ptr, out.ptr, dim, rows, eps, None), "kx_rmsnorm_f16out")
    return out.numpy(np.float16, (rows, dim))


def gemm_w8_rows(X, w, s, group_size=64, split_ws=None, half=False, k0=None, k1=None, residual=None):
    """Few-row product (T <= 192): returns Y[T][K], or (y0, y1, y2) when k0/k1 route weight-row ranges.
    half: hand the rows over as fp16.  split_ws: a zero-filled Buf of kx_gemm_split_workspace_bytes()."""
    K, M = w.shape
    T = X.shape[0]
    dx = Buf.from_numpy(X.astype(np.float16 if half else np.float32))
    dw, ds = Buf.from_numpy(w.astype(np.int8)), Buf.from_numpy(s.astype(np.float32))
    sp, spb = (split_ws.ptr, split_ws.nbytes) if split_ws is not None else (None, 0)
    fn = lib().kx_gemm_w8_rows_f16act if half else lib().kx_gemm_w8_rows_f32
    name = "kx_gemm_w8_rows_f16act" if half else "kx_gemm_w8_rows_f32"
    dr = Buf.from_numpy(residual.astype(np.float32)) if residual is not None else None
    if k0 is None:
        dy = Buf.zeros((T, K))
        check(fn(dx.ptr, M, dw.ptr, ds.ptr, group_size, dy.ptr, K, None, 0, None, 0, K, K,
                 dr.ptr if dr else None, K if dr else 0, T, M, K, sp, spb, None), name)
        sync()
        return dy.numpy()
    y0, y1, y2 = Buf.zeros((T, k0)), Buf.zeros((T, k1 - k0)), Buf.zeros((T, K - k1))
    check(fn(dx.ptr, M, dw.ptr, ds.ptr, group_size, y0.ptr, k0, y1.ptr, k1 - k0, y2.ptr, K - k1, k0, k1,
             dr.ptr if dr else None, k0 if dr else 0, T, M, K, sp, spb, None), name)
    sync()
    return y0.numpy(), y1.numpy(), y2.numpy()


def gemm_w8(X, w, s, group_size=64, two_pass=False):
    K, M = w.shape
    T = X.shape[0]
    dx, dw, ds = Buf.from_numpy(X.astype(np.float32)), Buf.from_numpy(w.astype(np.int8)), \
        Buf.from_numpy(s.astype(np.float32))
    dy = Buf.zeros((T, K))
    if two_pass:
        wsb = lib().kx_gemm_w8_workspace_bytes(T, M, K)
        ws = Buf(wsb)
        check(lib().kx_gemm_w8_f32_ws(dx.ptr, M, dw.ptr, ds.ptr, group_size, dy.ptr, K, None, 0, None,
                                      0, K, K, None, 0, T, M, K, ws.ptr, wsb, None),
              "kx_gemm_w8_f32_ws")
    else:
        check(lib().kx_gemm_w8_f32(dx.ptr, dw.ptr, ds.ptr, group_size, dy.ptr, T, M, K, None),
              "kx_gemm_w8_f32")
    return dy.numpy(np.float32, (T, K))


def quantize_q80(w, group_size=64):
    """quantize_q80 on the device: (int8 values shaped like w, fp32 scales [w.size / group_size])."""
    w = np.ascontiguousarray(w, np.float32)
    dw = Buf.from_numpy(w)
    dq, ds = Buf.zeros(w.size, np.int8), Buf.zeros(w.size // group_size)
    check(lib().kx_quantize_q80_f32(dw.ptr, w.size, group_size, dq.ptr, ds.ptr, None), "kx_quantize_q80_f32")
    return dq.numpy(np.int8, w.shape), ds.numpy(np.float32, (w.size // group_size,))


def make_gemv_desc(mode, x, w, s, y0, M, K, group_size=64, norm_w=None, norm_eps=1e-5, y1=None, y2=None, k0=0,
                   k1=0, bias=None, residual=None, sin=None, cos=None, pos_dev=None, pos=0, rope_variant=0,
                   head_size=0, kv_pos_stride=0, kv_f16=0):
    """A kx_gemv_w8_desc from device pointers (ints / None)."""
    d = GemvDesc()
    d.mode, d.group_size, d.M, d.K = mode, group_size, M, K
    d.x, d.w, d.scales, d.norm_w, d.norm_eps = x, w, s, norm_w, norm_eps
    d.y0, d.y1, d.y2, d.k0, d.k1 = y0, y1, y2, k0, k1
    d.bias, d.residual = bias, residual
    d.sin_cache, d.cos_cache, d.pos_dev, d.pos = sin, cos, pos_dev, pos
    d.rope_variant, d.head_size, d.kv_pos_stride, d.kv_f16 = rope_variant, head_size, kv_pos_stride, kv_f16
    return d


def gemv_chain(descs, workspace, chain_id=0, pos_dev=None, pos=0, stream=None):
    """kx_gemv_w8_chain_f32 over one or two descriptions; workspace: a Buf sized by
    kx_gemv_w8_chain_workspace_bytes and reset with kx_gemv_w8_chain_reset."""
    arr = (GemvDesc * len(descs))(*descs)
    check(lib().kx_gemv_w8_chain_f32(arr, len(descs), workspace.ptr, workspace.nbytes, chain_id, pos_dev, pos,
                                     stream), "kx_gemv_w8_chain_f32")
