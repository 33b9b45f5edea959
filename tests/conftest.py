import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # A/B of product switches over the whole suite (tools/gpu.sh: CC_NET_STREAMS=1 bash tools/gpu.sh TAG tests)
    if os.environ.get("CC_NET_STREAMS") is not None:
        from cc_amd import config as _cfg
        _cfg.net_streams = int(os.environ["CC_NET_STREAMS"])
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
