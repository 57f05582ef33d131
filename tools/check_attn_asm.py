"""Audits the compiled decode-attention kernels for the cache policy the split hand-off rests on:
every store of a partial is a device-scope (sc1, write-through) store, every load in the merge is
an sc1 buffer load, and no cache write-back / invalidate instruction is left in the kernel.
usage: check_attn_asm.py <attn_decode.s>   (hipcc -S --cuda-device-only output); exit 1 on a miss"""
import re
import sys


def audit(text):
    rows = []
    for m in re.finditer(r"^(_ZN2kx18attn_decode_kernel\w+):\s*(?:;.*)?$", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        lines = text[m.end():end].split("\n")
        stores = [l for l in lines if re.search(r"\b(global|buffer|flat)_store", l)]
        bl = [l for l in lines if "buffer_load" in l]
        rows.append({
            "kernel": m.group(1),
            "stores": len(stores),
            "stores_sc1": sum(" sc1" in l for l in stores),
            "merge_loads": len(bl),
            "merge_loads_sc1": sum(" sc1" in l for l in bl),
            "buffer_inv": sum("buffer_inv" in l for l in lines),
            "buffer_wbl2": sum("buffer_wbl2" in l for l in lines),
        })
    return rows


if __name__ == "__main__":
    rows = audit(open(sys.argv[1]).read())
    bad = False
    for r in rows:
        # the plain stores are the final output rows (single split / merged result) and the
        # optional timeline stamps; the partials are the sc1 ones
        ok = (r["stores_sc1"] >= 3 and r["merge_loads"] > 0 and
              r["merge_loads"] == r["merge_loads_sc1"] and r["buffer_inv"] == 0 and r["buffer_wbl2"] == 0)
        bad |= not ok
        print(("ok  " if ok else "MISS"), r)
    sys.exit(1 if bad or not rows else 0)
