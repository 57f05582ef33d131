"""The HBM-traffic record bench.py reports (profiles/r*_gemv_traffic.json) carries a digest of the
GEMV kernel source it was collected from; bench.py reports `traffic: null` when the source has
changed since.  This test fails in that case, so an edit of the kernel is followed by a new
collection (tools/refresh_profiles.sh on the GPU, tools/keep_refresh.sh here) before it is committed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_record_was_collected_from_the_gemv_source_in_the_tree():
    sys.path.insert(0, ROOT)
    import bench
    value, source = bench.measured_traffic()
    assert value is not None, source
    assert 0.9 < value / 61.88e6 < 1.2, (value, source)   # bytes per launch against the algorithmic figure
