#!/usr/bin/env python3
"""Per-rank compute time of a decode step at tensor-parallel degrees 1/2/4/8, measured on ONE
GPU: rank 0's shard of the model is built for each degree and its GEMV chain (graph replay of the
GEMV launches only, no collective) and its decode-attention launch are timed.  What a rank of an
N-GPU run spends outside the all-reduces."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402
from kx import host  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    shape = host.MODEL_SHAPES[name]
    L = kx.lib()
    out = {}
    for tp in (1, 2, 4, 8):
        if shape["kv_head_num"] % tp:
            continue
        m = host.Model(seq_len=ctx + 64, group_size=64, tp_size=tp, tp_rank=0, **shape)
        m.init_synthetic(1234)
        n, nbytes, ms = m.profile_gemv(4)
        H, KVH = shape["head_num"] // tp, shape["kv_head_num"] // tp
        hs = shape["dim"] // shape["head_num"]
        kvd = KVH * hs
        st = kx.Stream()
        q = kx.Buf(H * hs * 4)
        NL = 16  # cache layers walked by one timed replay (a one-kernel graph is launch-bound)
        kc, vc = kx.Buf(NL * (ctx + 64) * kvd * 4), kx.Buf(NL * (ctx + 64) * kvd * 4)
        o = kx.Buf(H * hs * 4)
        for buf, cnt, seed in ((q, H * hs, 3), (kc, NL * (ctx + 64) * kvd, 4),
                               (vc, NL * (ctx + 64) * kvd, 5)):
            kx.check(L.kx_fill_hash_f32(buf.ptr, 1, cnt, cnt, 0, 0, seed, -1.0, 1.0, st.ptr), "fill")
        wsb = L.kx_attn_workspace_bytes(H, hs)
        ws = kx.Buf.zeros(wsb, np.uint8)

        def attn():
            for layer in range(NL):
                kx.check(L.kx_mha_decode_f32(q.ptr, kc.ptr, vc.ptr, o.ptr, ctx - 1, H, H // KVH, hs,
                                             kvd, ctx + 64, layer, ws.ptr, wsb, st.ptr), "mha")
        attn_us = kx.time_graph(st, attn, replays=20) / NL
        gemv_us = ms * 1e3 / n
        step_us = ms * 1e3 / 4 + attn_us * shape["layer_num"]
        out[f"tp{tp}"] = {"gemv_launches": n // 4, "gemv_us_per_launch": gemv_us,
                          "gemv_gbps": nbytes / (ms * 1e-3) / 1e9, "attn_us": attn_us,
                          "compute_us_per_token": step_us,
                          "tok_per_s_without_collectives": 1e6 / step_us}
        m.close()
    print(json.dumps({"model": name, "context": ctx, "per_rank": out}))


if __name__ == "__main__":
    main()
