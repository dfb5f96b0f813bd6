import os
import sys
from pathlib import Path

import pytest

try:  # load torch's HIP runtime before libvsf_hip.so pulls in /opt/rocm's: the other order leaves torch without GPUs
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; PARITY UNPINNED, see oracle/vsf_oracle.h)."""
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def stereo640():
    from vision_slam_frontend_amd import synth
    return synth.stereo_pair(640, 480, 0)
