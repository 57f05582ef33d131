"""Both randomised sweeps (tools/fuzz_parity.py, tools/fuzz_model.py) against whichever library
KX_LIB_DIR selects, followed by the bounds-checking build's violation record (empty = good)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "llm-inference-engine_amd"))
import kx  # noqa: E402
import fuzz_model  # noqa: E402
import fuzz_parity  # noqa: E402

bad = fuzz_parity.run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, 1)
bad += fuzz_model.run(int(sys.argv[2]) if len(sys.argv) > 2 else 60, 42)
if kx.bounds_enabled():
    total, recs = kx.bounds_report()
    doc = {"violations": total, "records": [dict(zip(("file", "line", "operand", "offset", "extent"), r)) for r in recs]}
    os.makedirs(os.path.join(ROOT, "gpurun_out", "bounds"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bounds", "sweeps_bounds_report.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(f"[bounds] {total} violation(s) over both sweeps", recs[:10])
    bad += 1 if total else 0
else:
    print("[bounds] not a bounds-checking build")
print("problems:", bad)
sys.exit(1 if bad else 0)
