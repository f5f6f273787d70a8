"""Host-side mirror of `pt::renderer_pt::Renderer` (src/renderer_pt/renderer_pt.hpp:28-73) over the C ABI.

Same method names and argument meaning as the reference class, so that parity tests read like calls into the
reference: startRender(...) / render() / status() / renderProgress() / renderTime() / selectKernel(...).
The float accumulator (never exported by the reference) is the parity surface.
"""
import ctypes as C

import numpy as np

from . import abi, scenes


def make_params(width, height, spp, max_bounces, flags=abi.FLAG_MULTISCATTER_GGX, integrator=abi.INTEGRATOR_MIS,
                working_space=scenes.BT2020, gmon_buckets=1, first_sample=0, samples_in_flight=0,
                external_accumulator=None, stream=None, nonfinite_policy=abi.NONFINITE_PROPAGATE, accel_structure=abi.ACCEL_AUTO):
    p = abi.RenderParams()
    p.accel_structure = accel_structure
    p.width, p.height, p.spp, p.gmon_buckets = width, height, spp, gmon_buckets
    p.flags, p.integrator = flags, integrator
    p.working_space = scenes.colorspace(working_space)
    p.max_bounces, p.first_sample, p.samples_in_flight = max_bounces, first_sample, samples_in_flight
    p.nonfinite_policy = nonfinite_policy
    p.external_accumulator = external_accumulator
    p.stream = stream
    return p


class Renderer:
    # renderer_pt.hpp:16-26
    Integrator_Simple, Integrator_MIS = abi.INTEGRATOR_SIMPLE, abi.INTEGRATOR_MIS
    Status_Blocked, Status_Ready, Status_Busy, Status_Done = 0, 1, 4, 8

    def __init__(self, device=0, lut_path=None, devices=None):
        """Renderer(device, queue, store) (renderer_pt.cpp:18-60). Raises if the HIP library or a GPU is missing.
        `devices` = a list of HIP device ordinals: a device group that shards every render's samples (include/ptamd.h)."""
        self._lib = abi.load_library()
        info = abi.CreateInfo()
        info.abi_version = abi.PT_ABI_VERSION
        info.device_ordinal = device
        if devices is not None:
            self._devices = (C.c_int32 * len(devices))(*devices)
            info.device_ordinals = self._devices
            info.device_count = len(devices)
        self._lut = open(lut_path or abi.LUT_PATH, "rb").read()
        self._lut_buf = C.create_string_buffer(self._lut, len(self._lut))
        info.lut_blob = C.addressof(self._lut_buf)
        info.lut_blob_size = len(self._lut)
        info.lut_path = None
        h = C.c_void_p()
        abi.check(self._lib, self._lib.pt_create(C.byref(info), C.byref(h)))
        self._h = h
        self._integrator = abi.INTEGRATOR_MIS  # renderer_pt.hpp:98
        self._params = None
        self.size = (0, 0)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # renderer_pt.hpp:47-53
    def selectedKernel(self):
        return self._integrator

    def selectKernel(self, k):
        self._integrator = k

    def startRender(self, scene, size, spp, gmonBuckets=1, workingSpace=scenes.BT2020, flags=abi.FLAG_MULTISCATTER_GGX,
                    max_bounces=50, first_sample=0, samples_in_flight=0, external_accumulator=None, stream=None,
                    nonfinite_policy=abi.NONFINITE_PROPAGATE, accel_structure=abi.ACCEL_AUTO):
        """startRender(camera, size, spp, gmonBuckets, workingSpace, flags) (renderer_pt.hpp:38-45).
        `scene` (a scenes.Scene holding the camera node) replaces the NodeID into Store."""
        p = make_params(int(size[0]), int(size[1]), spp, max_bounces, flags, self._integrator, workingSpace, gmonBuckets,
                        first_sample, samples_in_flight, external_accumulator, stream, nonfinite_policy, accel_structure)
        snap = scene.snapshot()
        abi.check(self._lib, self._lib.pt_start_render(self._h, C.byref(snap.struct), C.byref(p)))
        self._params = p
        self.size = (p.width, p.height)

    def render(self, max_spp=1):
        """render() (renderer_pt.cpp:113-197) encodes 1 spp per call; max_spp=0 enqueues all remaining."""
        abi.check(self._lib, self._lib.pt_render_step(self._h, max_spp))

    def wait(self):
        abi.check(self._lib, self._lib.pt_wait(self._h))

    def status(self):
        return self._lib.pt_status(self._h)

    def renderProgress(self):
        a, t = C.c_uint64(), C.c_uint64()
        abi.check(self._lib, self._lib.pt_progress(self._h, C.byref(a), C.byref(t)))
        return a.value, t.value

    def renderTime(self):
        return self._lib.pt_render_time_ms(self._h)

    def readbackAccumulator(self):
        w, h = self.size
        out = np.empty((h, w, 4), dtype=np.float32)
        abi.check(self._lib, self._lib.pt_read_accumulator(self._h, out.ctypes.data))
        return out

    # renderer_pt.hpp:65-73: the option structs the UI edits every frame
    def postProcessOptions(self):
        o = abi.PostOptions()
        self._lib.pt_default_post_options(C.byref(o))
        return o

    def tonemapOptions(self):
        o = abi.TonemapOptions()
        self._lib.pt_default_tonemap_options(C.byref(o))
        return o

    def setPostProcessOptions(self, o):
        abi.check(self._lib, self._lib.pt_set_post_options(self._h, C.byref(o)))

    def setTonemapOptions(self, o):
        abi.check(self._lib, self._lib.pt_set_tonemap_options(self._h, C.byref(o)))

    def readbackRenderTarget(self):
        """readbackRenderTarget() (renderer_pt.cpp:1039-1059): (H, W, 4) uint8, post-processed + tonemapped."""
        w, h = self.size
        out = np.empty((h, w, 4), dtype=np.uint8)
        abi.check(self._lib, self._lib.pt_read_render_target(self._h, out.ctypes.data))
        return out

    def presentRenderTarget(self):
        """presentRenderTarget() (renderer_pt.hpp:55): (device address of the RGBA8 image, hipStream_t it is produced on)."""
        ptr, stream = C.c_void_p(), C.c_void_p()
        abi.check(self._lib, self._lib.pt_present_render_target(self._h, C.byref(ptr), C.byref(stream)))
        return ptr.value, stream.value

    def setGmonOptions(self, cap=1.0):
        """gmonOptions().cap (renderer_pt.hpp:71)."""
        o = abi.GmonOptions(cap)
        abi.check(self._lib, self._lib.pt_set_gmon_options(self._h, C.byref(o)))

    def readGmonBucket(self, bucket):
        w, h = self.size
        out = np.empty((h, w, 4), dtype=np.float32)
        abi.check(self._lib, self._lib.pt_read_gmon_bucket(self._h, bucket, out.ctypes.data))
        return out

    def accumulatorDevicePtr(self):
        return self._lib.pt_accumulator_device_ptr(self._h)

    # ---- parity / measurement surface ----
    def constants(self):
        c = abi.Constants()
        abi.check(self._lib, self._lib.pt_get_constants(self._h, C.byref(c)))
        return c

    def lights(self):
        n = C.c_uint32()
        abi.check(self._lib, self._lib.pt_get_lights(self._h, None, 0, C.byref(n)))
        arr = (abi.AreaLight * max(1, n.value))()
        abi.check(self._lib, self._lib.pt_get_lights(self._h, arr, n.value, C.byref(n)))
        return list(arr)[: n.value]

    def envAlias(self):
        """The environment alias table in use (Environment::rebuildAliasTable, core/environment.cpp:5-91)."""
        n = C.c_uint64()
        abi.check(self._lib, self._lib.pt_get_env_alias(self._h, None, 0, C.byref(n)))
        arr = np.zeros(n.value, dtype=abi.ALIAS_DTYPE)
        if n.value:
            abi.check(self._lib, self._lib.pt_get_env_alias(self._h, arr.ctypes.data, n.value, C.byref(n)))
        return arr

    def tracePrimary(self, sample_idx=0):
        w, h = self.size
        out = np.zeros(w * h, dtype=[("t", "f4"), ("u", "f4"), ("v", "f4"), ("instance", "i4"), ("primitive", "i4")])
        abi.check(self._lib, self._lib.pt_trace_primary(self._h, sample_idx, out.ctypes.data))
        return out.reshape(h, w)

    def debugSample(self, sample_idx):
        w, h = self.size
        B = self._params.max_bounces
        rad = np.zeros((h, w, 4), dtype=np.float32)
        hits = np.zeros((B, h, w, 2), dtype=np.int32)
        abi.check(self._lib, self._lib.pt_debug_sample(self._h, sample_idx, rad.ctypes.data, hits.ctypes.data))
        return rad, hits

    def stats(self):
        s = abi.Stats()
        abi.check(self._lib, self._lib.pt_get_stats(self._h, C.byref(s)))
        return s

    def setProfiling(self, enabled):
        abi.check(self._lib, self._lib.pt_set_profiling(self._h, int(enabled)))

    def measureTraversal(self, sample_idx=0):
        abi.check(self._lib, self._lib.pt_measure_traversal(self._h, sample_idx))
