"""Micro-benchmark of single-token attention over the fp32 KV cache (kx_mha_decode_f32 and the
device-position form used under graph replay).  Walks the layers of a cache larger than the
Infinity Cache so every launch streams cold rows."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402

GEOM = {  # head_num, kv_head_num, head_size
    "8b": (32, 8, 128),
    "1b": (32, 8, 64),
    "q7": (28, 4, 128),
    "70b-tp8": (8, 1, 128),
}


def main():
    names = sys.argv[1:] or ["8b", "1b"]
    L = kx.lib()
    st = kx.Stream()
    rng = np.random.default_rng(0)
    print(kx.device_name())
    for name in names:
        H, KVH, hs = GEOM[name]
        kvd = KVH * hs
        for pos in [int(p) for p in os.environ.get('KX_BENCH_POS', '255,1023,4095,8191').split(',')]:
            seq = pos + 1
            layers = max(2, min(64, int(700e6 // (2 * seq * kvd * 4)) + 1))
            kc = kx.Buf(layers * seq * kvd * 4)
            vc = kx.Buf(layers * seq * kvd * 4)
            kx.check(L.kx_fill_hash_f32(kc.ptr, layers * seq, kvd, kvd, 0, 0, 1, -1.0, 1.0, st.ptr))
            kx.check(L.kx_fill_hash_f32(vc.ptr, layers * seq, kvd, kvd, 0, 0, 2, -1.0, 1.0, st.ptr))
            q = kx.Buf.from_numpy(rng.standard_normal(H * hs).astype(np.float32))
            out = kx.Buf.zeros(H * hs)
            wsb = L.kx_attn_workspace_bytes(H, hs)
            if os.environ.get("KX_BENCH_NO_SPLIT"):
                wsb = 16384 + 64  # tickets only: forces a single split per kv head
            trace = bool(os.environ.get("KX_ATTN_TRACE"))
            used = wsb
            if trace:
                wsb += 64 * 1024
            ws = kx.Buf.zeros(wsb, np.uint8)
            pos_dev = kx.Buf.from_numpy(np.array([pos], np.int32))
            for mode in ("value", "device"):
                def launch(layer):
                    if mode == "value":
                        return L.kx_mha_decode_f32(q.ptr, kc.ptr, vc.ptr, out.ptr, pos, H, H // KVH,
                                                   hs, kvd, seq, layer, ws.ptr, wsb, st.ptr)
                    return L.kx_mha_decode_dpos_f32(q.ptr, kc.ptr, vc.ptr, out.ptr, pos_dev.ptr,
                                                    seq - 1, H, H // KVH, hs, kvd, seq, layer,
                                                    ws.ptr, wsb, st.ptr)
                reps = max(1, 96 // layers)

                def record():
                    for _ in range(reps):
                        for l in range(layers):
                            kx.check(launch(l))
                us = kx.time_graph(st, record) / (reps * layers)
                byts = 2 * seq * kvd * 4
                print(f"{name} pos={pos:5d} {mode:6s} layers={layers:3d} {us:8.2f} us "
                      f"{byts / us / 1e3:8.1f} GB/s", flush=True)
                if trace and mode == "value":
                    # timeline of ONE launch on cold rows (the graph above left layer `layers-1` last;
                    # layer 0 has long been evicted): per-workgroup wall-clock stamps, microseconds
                    # from the earliest workgroup's start
                    kx.check(launch(0))
                    st.sync()
                    raw = ws.numpy(np.uint8, (wsb,))[used:].view(np.uint64)
                    nwg = int((raw[0::8] != 0).sum())
                    t = raw[:nwg * 8].reshape(nwg, 8).astype(np.float64)
                    t0 = t[:, 0].min()
                    labels = ["start", "wave 0 tiles done", "all waves done", "partial acked",
                              "ticket drawn", "merged (last arrivers)"]
                    print(f"  timeline of one launch, {nwg} workgroups:")
                    for i, lab in enumerate(labels):
                        col = t[:, i]
                        col = col[col > 0]
                        if col.size:
                            col = (col - t0) / 100.0
                            print(f"    {lab:24s} n={col.size:4d} min {col.min():6.2f} median "
                                  f"{np.median(col):6.2f} max {col.max():6.2f}")


if __name__ == "__main__":
    main()
