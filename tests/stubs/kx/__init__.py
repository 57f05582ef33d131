"""A stand-in for the kx package WITHOUT a GPU behind it: just enough of the binding's surface for
bench.py to run its whole control flow (N > 1 included) on CPU processes over gloo.  Used only by
tests/test_bench_multirank_cpu.py; it computes nothing and measures nothing."""
import os


class KxError(RuntimeError):
    pass


class _Lib:
    def kx_set_device(self, dev):
        return 0


def lib():
    return _Lib()


def check(rc, what=""):
    if rc != 0:
        raise KxError(f"{what} failed: rc={rc}")


def sync():
    pass


STUB = True
FAIL_PEER_RANK = int(os.environ.get("KX_STUB_FAIL_PEER_RANK", "-1"))
