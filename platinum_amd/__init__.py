"""platinum_amd — MI355X-native wavefront path tracer behind teofum/platinum's `renderer_pt` surface.

Layout: csrc/ (HIP kernels + C-ABI, libptamd.so), abi.py (ctypes binding of include/ptamd.h),
renderer.py (the `Renderer`-shaped host mirror), scenes.py (Scene/primitives/camera restated for headless use).
"""
from . import abi, scenes  # noqa: F401
from .renderer import Renderer, make_params  # noqa: F401
