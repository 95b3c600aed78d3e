import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """libmvq_hip.so is built in-tree by __graft_entry__.build(); if a checkout has not been built yet, build it once here
    (hipcc cross-compiles without a GPU) so that the ABI / host tests have something to load."""
    from multimodal_vqvae_compression_audio_tactile_amd import _lib
    if not _lib.SO_PATH.exists():
        import shutil
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            _lib.build()
    yield


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    if os.environ.get("MVQ_TEST_POISON") == "1":
        # every torch.empty() comes back filled with NaN: a kernel that reads an element its producer never wrote (a tile tail, a
        # padded column) then shows up as a NaN / a mismatch instead of depending on what the caching allocator last kept there
        torch.utils.deterministic.fill_uninitialized_memory = True
        torch.use_deterministic_algorithms(True, warn_only=True)
    return torch.device("cuda:0")
