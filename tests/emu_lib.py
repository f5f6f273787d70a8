"""ctypes loader for tests/_build/libwavefront_emu.so — the host build of the product's per-path stage functions
(TEST HARNESS, see tests/emu/wavefront_emu.cpp). Never imported by platinum_amd."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from platinum_amd import abi  # noqa: E402

SRC = os.path.join(_ROOT, "tests", "emu", "wavefront_emu.cpp")
LIB = os.path.join(_ROOT, "tests", "_build", "libwavefront_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    csrc = os.path.join(_ROOT, "platinum_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-o", LIB, SRC])
    L = C.CDLL(LIB)
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.POINTER(abi.SceneSnapshot), C.POINTER(abi.RenderParams), C.c_void_p, C.c_uint64]
    L.emu_destroy.argtypes = [C.c_void_p]
    L.emu_get_constants.argtypes = [C.c_void_p, C.POINTER(abi.Constants)]
    L.emu_get_lights.restype = C.c_uint32
    L.emu_get_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.emu_debug_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.emu_trace_primary.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.emu_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.emu_halton.restype = C.c_float
    L.emu_halton.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    _lib = L
    return L


class EmuScene:
    def __init__(self, scene, params):
        self.L = lib()
        self.params = params
        self.snapshot = scene.snapshot()
        blob = open(abi.LUT_PATH, "rb").read()
        self._blob = C.create_string_buffer(blob, len(blob))
        self.h = self.L.emu_create(C.byref(self.snapshot.struct), C.byref(params), self._blob, len(blob))
        if not self.h:
            raise RuntimeError("emu_create failed")
        self.W, self.H = params.width, params.height

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_destroy(self.h)
            self.h = None

    def constants(self):
        c = abi.Constants()
        self.L.emu_get_constants(self.h, C.byref(c))
        return c

    def lights(self):
        arr = (abi.AreaLight * 65536)()
        n = self.L.emu_get_lights(self.h, arr, 65536)
        return list(arr)[:n]

    def debug_sample(self, sample_idx):
        B = self.params.max_bounces
        rad = np.zeros((self.H, self.W, 4), dtype=np.float32)
        hits = np.zeros((B, self.H, self.W, 2), dtype=np.int32)
        self.L.emu_debug_sample(self.h, sample_idx, rad.ctypes.data, hits.ctypes.data)
        return rad, hits

    def trace_primary(self, sample_idx=0):
        out = np.zeros(self.W * self.H, dtype=[("t", "f4"), ("u", "f4"), ("v", "f4"), ("instance", "i4"), ("primitive", "i4")])
        self.L.emu_trace_primary(self.h, sample_idx, out.ctypes.data)
        return out.reshape(self.H, self.W)

    def render(self, first, ns, acc=None, acc_n0=0, threads=1):
        """Samples [first, first + ns) of every pixel folded into the running mean `acc` (acc_n0 samples already in it): the product's
        stage functions on `threads` host threads (bench.py's cpu_baseline, kind "same-kernels-host")."""
        if acc is None:
            acc = np.zeros((self.H, self.W, 4), dtype=np.float32)
        assert acc.dtype == np.float32 and acc.flags["C_CONTIGUOUS"] and acc.shape == (self.H, self.W, 4)
        self.L.emu_render(self.h, first, ns, acc.ctypes.data, acc_n0, threads)
        return acc

    def halton(self, i, d):
        return self.L.emu_halton(self.h, i, d)
