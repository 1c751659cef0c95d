import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU runs: import torch (and scipy) once, BEFORE the first test and outside every per-test timeout.  On a fresh
    box the first `import torch` pages in several GB and has been seen to take more than five minutes; the two tests
    that use torch for device buffers would otherwise spend their whole time budget on the import."""
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        return
    if any(item.get_closest_marker("gpu") for item in items):
        try:
            import torch  # noqa: F401
            import scipy.optimize  # noqa: F401
        except Exception:  # the tests that need them report it themselves
            pass
