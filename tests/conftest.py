import os
import sys

import numpy as np
import pytest

# torch ships its own ROCm runtime: load it BEFORE libsfmhip.so pulls in the system libamdhip64,
# or a later `import torch` in the same process no longer finds the GPU ("No HIP GPUs are
# available").  bench.py imports torch first for the same reason.
try:
    import torch  # noqa: F401
except ImportError:  # the C ABI itself does not need torch
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def golden():
    return {n[:-4]: np.load(os.path.join(GOLDEN, n)) for n in os.listdir(GOLDEN) if n.endswith(".npz")}


@pytest.fixture(scope="session")
def ctx():
    """sfmhip context on GPU 0; only gpu-marked tests may request it."""
    from sfm_danpipeline_amd import _lib, build
    build.build()
    return _lib.default_context()
