import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "llm-inference-engine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def golden_bytes():
    def load(name):
        with open(os.path.join(GOLDEN, name), "rb") as f:
            return f.read()
    return load


def pytest_sessionfinish(session, exitstatus):
    """Under the bounds-checking build (KX_LIB_DIR=.../lib_bounds) the run ends with the library's
    violation record: written to gpurun_out/bounds_report.json, and any entry fails the run."""
    import json
    try:
        import kx
        if kx._lib is None or not kx.bounds_enabled():
            return
        total, recs = kx.bounds_report()
    except Exception as e:  # no library / no GPU: nothing to report
        print(f"\n[bounds] no report: {e}")
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    doc = {"violations": total, "records": [dict(zip(("file", "line", "operand", "offset", "extent"), r))
                                             for r in recs],
           "tests_collected": session.testscollected, "tests_failed": session.testsfailed}
    with open(os.path.join(out, "bounds_report.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(f"\n[bounds] {total} violation(s) recorded over {session.testscollected} tests")
    for r in recs[:20]:
        print("[bounds]  ", r)
    if total and session.exitstatus == 0:
        session.exitstatus = 1
