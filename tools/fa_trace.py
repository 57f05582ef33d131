"""Per-phase cycle counts of the flash-attention prompt kernel's heaviest workgroup.  Needs a library
built with -DFA_TRACE (KX_HIP_LIB names it); prints, per wave, the cycles spent between the stamps of
the tile loop, the loop total, and the clock (cycles per 10 ns tick of the 100 MHz wall counter)."""
import ctypes, os, sys
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
    kx.check(L.kx_flash_attention_f32(q.ptr, kc.ptr, vc.ptr, out.ptr, T, H, KVH, hs, T, T - 1, 0, float(1 / np.sqrt(hs)), 1, ws.ptr, wsb, st.ptr))
us = min(kx.time_graph(st, fa, replays=20) for _ in range(3))
st.sync()
raw = (ctypes.c_ulonglong * 128)()
raw_lib = ctypes.CDLL(os.environ["KX_HIP_LIB"])
assert raw_lib.kx_fa_trace_read(raw) == 0
a = np.array(raw[:], dtype=np.float64).reshape(8, 16)
names = ["loop-top", "load-issue", "S mfma", "max+rescale", "exp+PV", "ds_write", "barrier"]
print(f"{us:.1f} us per launch; tiles {int(a[0,10])}")
for w in range(8):
    nt = a[w, 10]
    print(f"wave {w}: " + "  ".join(f"{n} {a[w,i]/nt:.0f}" for i, n in enumerate(names)) +
          f"  | per tile {a[w,8]/nt:.0f} cyc, clock {a[w,8]/a[w,9]/10:.2f} GHz"
          f"  | prologue {a[w,11]:.0f} cyc, epilogue (stores acknowledged) {a[w,12]:.0f} cyc")
