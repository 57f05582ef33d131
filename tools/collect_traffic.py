"""Turns rocprofv3 --pmc counter CSVs of a bench.py run into the per-launch HBM traffic of the
W8 GEMV kernel, the figure bench.py reports as roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/collect_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv \
        gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_gemv_traffic.json

Corrections (MI355X / gfx950): both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of
a wide coalesced streaming read at 64 bytes, so it is doubled; WRITE_SIZE is exact.  The two
counters need separate passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gemv_source_digest():
    """Digest of the GEMV kernel source the counters were collected from; bench.py reports the
    stored traffic only while the tree still holds that source."""
    h = hashlib.sha256()
    for name in ("gemv_w8.hip", "gemv_w8.hpp"):
        with open(os.path.join(ROOT, "llm-inference-engine_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            acc[name][0] += 1
            acc[name][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(name, [0, 0.0])
        nw, sw = write.get(name, [0, 0.0])
        kernels[name] = {
            "launches_fetch_pass": nf,
            "launches_write_pass": nw,
            "fetch_bytes_per_launch": 2.0 * 1024.0 * sf / nf if nf else None,
            "write_bytes_per_launch": 1024.0 * sw / nw if nw else None,
        }
    # gemv_w8_pick_kernel: the classifier launch of the token loop (the GEMV that also ends the step)
    gemv = {k: v for k, v in kernels.items() if "gemv_w8_kernel" in k or "gemv_w8_pick_kernel" in k}
    n = sum(v["launches_fetch_pass"] for v in gemv.values())
    fetch_total = sum(v["fetch_bytes_per_launch"] * v["launches_fetch_pass"] for v in gemv.values())
    nw = sum(v["launches_write_pass"] for v in gemv.values())
    write_total = sum((v["write_bytes_per_launch"] or 0.0) * v["launches_write_pass"] for v in gemv.values())
    doc = {
        "kernel": "gemv_w8_kernel (all instantiations)",
        "kernel_source_sha256": gemv_source_digest(),
        "launches": n,
        "hbm_bytes_per_launch": fetch_total / n + (write_total / nw if nw else 0.0),
        "fetch_bytes_per_launch": fetch_total / n,
        "write_bytes_per_launch": write_total / nw if nw else None,
        "corrections": "FETCH_SIZE, WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (128-B requests "
                       "tallied at 64 B); separate --pmc passes",
        "per_kernel": kernels,
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("launches", "hbm_bytes_per_launch", "fetch_bytes_per_launch",
                                          "write_bytes_per_launch")}))


if __name__ == "__main__":
    main()
