"""ctypes loader for the CPU oracle (oracle/_build/libptoracle.so) — TEST INFRASTRUCTURE.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from platinum_amd import abi  # noqa: E402  (struct layouts only)

ORACLE_DIR = os.path.join(_ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libptoracle.so")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "triangles", "bvh_nodes", "closest_rays", "shadow_rays", "shaded_hits", "paths",
        "nodes_closest", "tris_closest", "nodes_shadow", "tris_shadow", "nonfinite")]


_lib = None


def host_threads():
    """Worker threads for the oracle: the affinity mask cut down to the cgroup's CPU quota (a one-GPU box of the pool shows 256 logical
    CPUs and grants 16; 256 threads on 16 cores run the oracle at half its speed)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(per)))))
    except (OSError, ValueError, IndexError):
        pass
    return max(1, n)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pt_oracle.cpp", "pt_oracle.h", "oracle_math.h")]
    if not os.path.exists(ORACLE_LIB) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs):
        build()
    L = C.CDLL(ORACLE_LIB)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.POINTER(abi.SceneSnapshot), C.POINTER(abi.RenderParams), C.c_void_p, C.c_uint64, C.c_int]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_get_constants.argtypes = [C.c_void_p, C.POINTER(abi.Constants)]
    L.orc_get_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_atan2.restype = C.c_float; L.orc_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_acos.restype = C.c_float; L.orc_acos.argtypes = [C.c_float]
    L.orc_tex_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.orc_ray_dir_to_uv.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_uv_to_ray_dir.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_get_env_alias.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    L.orc_gmon_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
    L.orc_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.PostOptions), C.POINTER(abi.TonemapOptions), C.c_void_p, C.c_void_p]
    L.orc_trace_primary.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_debug_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_render_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
    L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(OrcStats)]
    L.orc_halton_offset.restype = C.c_uint32
    L.orc_halton_offset.argtypes = [C.c_uint32] * 3
    L.orc_pcg4d.argtypes = [C.c_uint32 * 4, C.c_uint32 * 4]
    L.orc_halton.restype = C.c_float
    L.orc_halton.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_prime.restype = C.c_uint32
    L.orc_prime.argtypes = [C.c_uint32]
    L.orc_fresnel.restype = C.c_float
    L.orc_fresnel.argtypes = [C.c_float, C.c_float]
    L.orc_avg_dielectric_fresnel_fit.restype = C.c_float
    L.orc_avg_dielectric_fresnel_fit.argtypes = [C.c_float]
    L.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_log2.restype = C.c_float
    L.orc_log2.argtypes = [C.c_float]
    L.orc_exp2.restype = C.c_float
    L.orc_exp2.argtypes = [C.c_float]
    L.orc_sample_cosine_hemisphere.argtypes = [C.c_float, C.c_float, C.c_float * 3]
    L.orc_sample_tri_uniform.argtypes = [C.c_float, C.c_float, C.c_float * 2]
    L.orc_lut_sample.restype = C.c_float
    L.orc_lut_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
    L.orc_lut_regen.restype = C.c_double
    L.orc_lut_regen.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_lut_regen_texel.restype = C.c_double
    L.orc_lut_regen_texel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_lut_texel.restype = C.c_float
    L.orc_lut_texel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_bsdf_sample.argtypes = [C.c_void_p, C.POINTER(abi.MaterialGPU), C.c_float * 3, C.c_float * 4, C.c_float * 2, C.c_float * 11]
    L.orc_bsdf_eval.argtypes = [C.c_void_p, C.POINTER(abi.MaterialGPU), C.c_float * 3, C.c_float * 3, C.c_float * 4]
    _lib = L
    return L


def lut_blob():
    return open(abi.LUT_PATH, "rb").read()


class OracleScene:
    """One flattened scene + render params inside the oracle."""

    def __init__(self, scene, params, use_bvh=True):
        self.L = lib()
        self.params = params
        self.snapshot = scene.snapshot()
        blob = lut_blob()
        self._blob = C.create_string_buffer(blob, len(blob))
        self.h = self.L.orc_scene_create(C.byref(self.snapshot.struct), C.byref(params), self._blob, len(blob), int(use_bvh))
        if not self.h:
            raise RuntimeError("orc_scene_create failed")
        self.W, self.H = params.width, params.height

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def constants(self):
        c = abi.Constants()
        self.L.orc_get_constants(self.h, C.byref(c))
        return c

    def lights(self):
        n = C.c_uint32()
        self.L.orc_get_lights(self.h, None, 0, C.byref(n))
        arr = (abi.AreaLight * max(1, n.value))()
        self.L.orc_get_lights(self.h, arr, n.value, C.byref(n))
        return list(arr)[: n.value]

    def tex_sample(self, texture, u, v):
        out = np.zeros(4, dtype=np.float32)
        self.L.orc_tex_sample(self.h, texture, u, v, out.ctypes.data)
        return out

    def envAlias(self):
        n = C.c_uint64()
        self.L.orc_get_env_alias(self.h, None, 0, C.byref(n))
        arr = np.zeros(n.value, dtype=abi.ALIAS_DTYPE)
        if n.value:
            self.L.orc_get_env_alias(self.h, arr.ctypes.data, n.value, C.byref(n))
        return arr

    def render(self, first_sample, nsamples, acc=None, acc_n0=0, threads=None, count_traversal=False):
        if acc is None:
            acc = np.zeros((self.H, self.W, 4), dtype=np.float32)
        threads = threads or host_threads()
        self.L.orc_render(self.h, first_sample, nsamples, acc.ctypes.data, acc_n0, threads, int(count_traversal))
        return acc

    def render_pixels(self, xy, first_sample, nsamples, acc_n0=0, threads=None):
        """The running mean of samples [first_sample, first_sample + nsamples) for the listed (x, y) pixels only: [len(xy), 4]."""
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros((len(xy), 4), dtype=np.float32)
        self.L.orc_render_pixels(self.h, xy.ctypes.data, len(xy), first_sample, nsamples, out.ctypes.data, acc_n0, threads or host_threads())
        return out

    def render_gmon(self, nsamples, threads=None):
        """All `nsamples` (= params.spp) samples into the GMoN buckets; returns (buckets[B,H,W,4], resolved[H,W,4])."""
        B = self.params.gmon_buckets
        buckets = np.zeros((B, self.H, self.W, 4), dtype=np.float32)
        self.L.orc_render(self.h, 0, nsamples, buckets.ctypes.data, 0, threads or host_threads(), 0)
        spb = (self.params.spp + B - 1) // B
        full = (nsamples - 1) // spb + 1
        return buckets, self.gmon_resolve(buckets, full)

    def gmon_resolve(self, buckets, full_buckets, cap=1.0):
        out = np.zeros((self.H, self.W, 4), dtype=np.float32)
        b = np.ascontiguousarray(buckets, dtype=np.float32)
        self.L.orc_gmon_resolve(self.h, b.ctypes.data, full_buckets, cap, out.ctypes.data)
        return out

    def postprocess(self, acc, post, tonemap, want_float=False):
        acc = np.ascontiguousarray(acc, dtype=np.float32)
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        fl = np.zeros((self.H, self.W, 3), dtype=np.float32) if want_float else None
        self.L.orc_postprocess(self.h, acc.ctypes.data, C.byref(post), C.byref(tonemap), out.ctypes.data, fl.ctypes.data if want_float else None)
        return (out, fl) if want_float else out

    def trace_primary(self, sample_idx=0):
        out = np.zeros(self.W * self.H, dtype=[("t", "f4"), ("u", "f4"), ("v", "f4"), ("instance", "i4"), ("primitive", "i4")])
        self.L.orc_trace_primary(self.h, sample_idx, out.ctypes.data)
        return out.reshape(self.H, self.W)

    def debug_sample(self, sample_idx, threads=None):
        B = self.params.max_bounces
        rad = np.zeros((self.H, self.W, 4), dtype=np.float32)
        hits = np.zeros((B, self.H, self.W, 2), dtype=np.int32)
        self.L.orc_debug_sample(self.h, sample_idx, rad.ctypes.data, hits.ctypes.data, threads or host_threads())
        return rad, hits

    def stats(self):
        s = OrcStats()
        self.L.orc_get_stats(self.h, C.byref(s))
        return s
