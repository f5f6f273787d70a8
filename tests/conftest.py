import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# (r4) The former torch-first hook is gone.  It hid a load-order fault: torch's wheel bundles its own libamdhip64 / libhsa-runtime64,
# and a libptamd.so loaded BEFORE torch mapped /opt/rocm's copies beside them - two HIP runtimes, the second of which never sees the
# GPU ("No HIP GPUs are available").  platinum_amd.abi.load_library() now settles the process on ONE runtime in either order and
# pt_create refuses a process that holds two (tests/test_runtime_identity.py).

STRUCTURE_ENV = ("PTAMD_BVH4", "PTAMD_RADIX_TREE", "PTAMD_BVH_LEGACY", "PTAMD_TWO_LEVEL", "PTAMD_TEST_W6_LEVELS")


def skip_if_structure_env_preset():
    """Tests that switch the acceleration-structure variables themselves compare "default" against "forced": they mean nothing when the
    whole session already runs with one of them exported (the sweeps of DESIGN section 2 do that), so they skip."""
    preset = [v for v in STRUCTURE_ENV if v in os.environ]
    if preset:
        pytest.skip("preset for the whole session: " + ", ".join(preset))


@pytest.fixture(scope="session")
def gpu_renderer():
    """One Renderer (one HIP context) shared by every GPU test in the session."""
    from platinum_amd import Renderer
    r = Renderer(device=0)
    yield r
    r.close()
