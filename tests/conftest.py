import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as `import mmtg_amd` sets it (before the HIP runtime initialises)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a plain
    `pytest tests/` works in the CPU-only build container."""
    import torch
    # The oracle-vs-golden tests run first: after test_host_cpu.py's multi-process tests (gloo workers, bench.py self-launches)
    # the oracle's 220-position greedy loop was measured 25x slower in the same pytest process in this 8-CPU container
    # (230 s instead of 8 s; tiny CPU ops, timing-dependent) -- order only, nothing is skipped.
    items.sort(key=lambda it: 0 if "test_oracle_golden" in it.nodeid else 1)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
