import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU thread pools within the cgroup quota: the oracle's CPU runs are no faster with 128 threads, and the spinning
    # workers throttle the whole container (nsdp_amd/cpu_budget.py)
    from nsdp_amd.cpu_budget import cap_thread_pools
    cap_thread_pools(16)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
