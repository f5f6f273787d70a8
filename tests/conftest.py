import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_finish(session):
    """GPU sessions: bring torch's HIP runtime up BEFORE libptamd.so initialises its own (the order bench.py uses: torch owns the
    accumulator tensor, then the library is created).  torch's wheel bundles its own libamdhip64 / libhsa-runtime64; a torch that
    first touches the GPU after the library has done so can fail with "No HIP GPUs are available" depending on what ran in between
    (seen once when the test files were run in a non-alphabetical order)."""
    if not any(item.get_closest_marker("gpu") for item in session.items):
        return
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass


@pytest.fixture(scope="session")
def gpu_renderer():
    """One Renderer (one HIP context) shared by every GPU test in the session."""
    from platinum_amd import Renderer
    r = Renderer(device=0)
    yield r
    r.close()
