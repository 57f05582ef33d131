"""Summarises a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
GRBM_GUI_ACTIVE) over tools/bench_prefill.py into per-kernel MFMA figures:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * SIMDs)   (the tool's MfmaUtil formula;
                the CSV holds GRBM_GUI_ACTIVE summed over the 8 XCDs, the formula wants one XCD's)
    flops     = SQ_INSTS_VALU_MFMA_MOPS_F16 * 512
usage: collect_mfma.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
XCDS = 8

acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "attn_prefill_kernel" in name or "gemm_f16_" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, c in acc.items():
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    gui = avg.get("GRBM_GUI_ACTIVE")
    out[name] = {
        "launches": len(next(iter(c.values()))),
        "counters_avg_per_launch": avg,
        "mfma_busy_fraction": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / XCDS * SIMDS) if gui else None,
        "active_cycles_per_xcd": gui / XCDS if gui else None,
        "mfma_flops_per_launch": avg.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512,
    }
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
