"""Times the flash-attention prompt kernel alone.  FA_GEOM = rows,heads,kv_heads,head_size
(default 4096,32,8,128); KX_HIP_LIB selects another build of the library for A/B runs."""
# times the flash-attention prefill kernel (8B geometry, 4096 rows) of whichever library KX_HIP_LIB names
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llm-inference-engine_amd"))
import kx
L = kx.lib()
st = kx.Stream()
T, H, KVH, hs = [int(v) for v in os.environ.get("FA_GEOM", "4096,32,8,128").split(",")]
kvd = KVH * hs
q = kx.Buf(T * H * hs * 4); kc = kx.Buf(T * kvd * 4); vc = kx.Buf(T * kvd * 4); out = kx.Buf(T * H * hs * 4)
for buf, n, seed in ((q, T * H * hs, 3), (kc, T * kvd, 4), (vc, T * kvd, 5)):
    kx.check(L.kx_fill_hash_f32(buf.ptr, 1, n, n, 0, 0, seed, -1.0, 1.0, st.ptr))
wsb = max(L.kx_attn_workspace_bytes(H, hs), L.kx_flash_attention_workspace_bytes(T - 1, KVH, hs))
ws = kx.Buf.zeros(wsb, np.uint8)
def fa():
    kx.check(L.kx_flash_attention_f16out(q.ptr, kc.ptr, vc.ptr, out.ptr, T, H, KVH, hs, T, T - 1, 0, float(1 / np.sqrt(hs)), 1, ws.ptr, wsb, st.ptr) if os.environ.get("FA_F16OUT") else L.kx_flash_attention_f32(q.ptr, kc.ptr, vc.ptr, out.ptr, T, H, KVH, hs, T, T - 1, 0, float(1 / np.sqrt(hs)), 1, ws.ptr, wsb, st.ptr))
us = min(kx.time_graph(st, fa, replays=10) for _ in range(3))
o = out.numpy(np.float32, (T * H * hs,))
print(os.path.basename(os.environ.get("KX_HIP_LIB", "default")), f"{us:.1f} us  {4.0*hs*T*(T+1)/2*H/us*1e-6:.0f} TF  chk {float(np.nansum(np.abs(o[:100000]))):.4e}", flush=True)
