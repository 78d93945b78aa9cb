import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import rtmodt_amd  # noqa: F401
    return sys.modules["rtmodt_amd"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def unpack_frames(z):
    """Inverse of oracle/gen_golden_tracker.py:pack_inputs."""
    off = z["in_offsets"]
    return [(z["in_xyxy"][off[i]:off[i + 1]], z["in_conf"][off[i]:off[i + 1]], z["in_cls"][off[i]:off[i + 1]])
            for i in range(len(off) - 1)]
