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


def _gpu_unavailable_reason():
    """None when gpu-marked tests can run here; otherwise why not.  With a GPU present a missing libvsf_hip.so is NOT a
    reason to skip: those tests must then fail loudly (the product has no fallback path)."""
    try:
        if torch is None or not torch.cuda.is_available():
            return "no GPU visible (gpu-marked tests run on the MI355X box: pytest -m gpu)"
    except Exception as e:  # pragma: no cover
        return "torch cannot query the GPU: %r" % (e,)
    return None


def pytest_collection_modifyitems(config, items):
    reason = _gpu_unavailable_reason()
    if reason is None:
        return
    skip = pytest.mark.skip(reason=reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
