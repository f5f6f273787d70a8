"""ctypes mirror of include/ptamd.h and the loader of libptamd.so.

This is the reference-side binding a maintainer would write for `pt::renderer_pt::Renderer`
(src/renderer_pt/renderer_pt.hpp:28-73) — see INTEGRATION.md.  There is no CPU fallback: if the HIP
library is missing or cannot find a device, loading / pt_create raises.
"""
import ctypes as C
import os

PT_ABI_VERSION = 4

# renderer_pt.hpp:21-26
STATUS_BLOCKED, STATUS_READY, STATUS_BUSY, STATUS_DONE = 0, 1, 4, 8
# renderer_pt.hpp:16-19
INTEGRATOR_SIMPLE, INTEGRATOR_MIS = 0, 1
# pt_shader_defs.hpp:75-79
FLAG_NONE, FLAG_MULTISCATTER_GGX, FLAG_GMON = 0, 1, 2
# pt_shader_defs.hpp:85-90
MATERIAL_THIN_DIELECTRIC, MATERIAL_USE_ALPHA, MATERIAL_EMISSIVE, MATERIAL_ANISOTROPIC = 1, 2, 4, 8
NONFINITE_PROPAGATE, NONFINITE_ZERO = 0, 1
ACCEL_AUTO, ACCEL_ONE_BVH, ACCEL_TWO_LEVEL = 0, 1, 2


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("_pad", C.c_float)]


class VertexData(C.Structure):
    _fields_ = [("normal", Float3), ("tangent", C.c_float * 4), ("texCoords", C.c_float * 2), ("_pad", C.c_float * 2)]


class MaterialGPU(C.Structure):
    _fields_ = [
        ("baseColor", C.c_float * 4), ("emission", Float3), ("emissionStrength", C.c_float),
        ("roughness", C.c_float), ("metallic", C.c_float), ("transmission", C.c_float), ("ior", C.c_float),
        ("anisotropy", C.c_float), ("anisotropyRotation", C.c_float), ("clearcoat", C.c_float),
        ("clearcoatRoughness", C.c_float), ("flags", C.c_int32), ("baseTextureId", C.c_int32),
        ("rmTextureId", C.c_int32), ("transmissionTextureId", C.c_int32), ("clearcoatTextureId", C.c_int32),
        ("emissionTextureId", C.c_int32), ("normalTextureId", C.c_int32),
    ]


class Mesh(C.Structure):
    _fields_ = [
        ("positions", C.c_void_p), ("vertex_data", C.c_void_p), ("indices", C.c_void_p),
        ("material_slots", C.c_void_p), ("vertex_count", C.c_uint32), ("triangle_count", C.c_uint32),
    ]


class Instance(C.Structure):
    _fields_ = [
        ("transform", (C.c_float * 3) * 4), ("options", C.c_uint32), ("mask", C.c_uint32),
        ("intersectionFunctionTableOffset", C.c_uint32), ("accelerationStructureIndex", C.c_uint32),
    ]


class InstanceMaterials(C.Structure):
    _fields_ = [("materials", C.c_void_p), ("material_count", C.c_uint32), ("_pad", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [
        ("world", (C.c_float * 4) * 4), ("sensor_size", C.c_float * 2), ("focal_length", C.c_float),
        ("aperture", C.c_float), ("aperture_blades", C.c_uint32), ("roundness", C.c_float),
        ("bokeh_power", C.c_float), ("focus_distance", C.c_float),
    ]


class Colorspace(C.Structure):
    _fields_ = [("r", C.c_float * 2), ("g", C.c_float * 2), ("b", C.c_float * 2), ("w", C.c_float * 2)]


TEX_RGBA8_SRGB, TEX_RGBA8, TEX_RG8, TEX_R8, TEX_RGBA32F = 0, 1, 2, 3, 4


class Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32), ("_pad", C.c_uint32)]


class AliasEntry(C.Structure):
    _fields_ = [("pdf", C.c_float), ("p", C.c_float), ("aliasIdx", C.c_uint32)]


ALIAS_DTYPE = [("pdf", "f4"), ("p", "f4"), ("aliasIdx", "u4")]  # numpy view of pt_alias_entry


class SceneSnapshot(C.Structure):
    _fields_ = [
        ("meshes", C.c_void_p), ("mesh_count", C.c_uint32), ("instance_count", C.c_uint32),
        ("instances", C.c_void_p), ("instance_materials", C.c_void_p), ("camera", Camera),
        ("textures", C.c_void_p), ("texture_count", C.c_uint32), ("env_texture", C.c_int32), ("env_alias", C.c_void_p),
    ]


class CameraData(C.Structure):
    _fields_ = [
        ("position", Float3), ("topLeft", Float3), ("pixelDeltaU", Float3), ("pixelDeltaV", Float3),
        ("apertureRadius", C.c_float), ("apertureBlades", C.c_uint32), ("apertureRoundness", C.c_float),
        ("bokehPower", C.c_float),
    ]


class Constants(C.Structure):
    _fields_ = [
        ("frameIdx", C.c_uint32), ("spp", C.c_uint32), ("gmonBuckets", C.c_uint32), ("lightCount", C.c_uint32),
        ("envLightCount", C.c_uint32), ("lutSizeE", C.c_uint32), ("lutSizeEavg", C.c_uint32), ("flags", C.c_int32),
        ("totalLightPower", C.c_float), ("_pad0", C.c_uint32), ("size", C.c_uint32 * 2), ("idt", Float3 * 3),
        ("camera", CameraData),
    ]


class AreaLight(C.Structure):
    _fields_ = [
        ("instanceIdx", C.c_uint32), ("indices", C.c_uint32 * 3), ("area", C.c_float), ("power", C.c_float),
        ("cumulativePower", C.c_float), ("_pad", C.c_float), ("emission", Float3),
    ]


class CreateInfo(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device_ordinal", C.c_int32), ("lut_blob", C.c_void_p),
        ("lut_blob_size", C.c_uint64), ("lut_path", C.c_char_p),
        ("device_ordinals", C.POINTER(C.c_int32)), ("device_count", C.c_uint32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("gmon_buckets", C.c_uint32),
        ("flags", C.c_int32), ("integrator", C.c_uint32), ("working_space", Colorspace),
        ("max_bounces", C.c_uint32), ("first_sample", C.c_uint32), ("samples_in_flight", C.c_uint32),
        ("nonfinite_policy", C.c_uint32), ("external_accumulator", C.c_void_p), ("stream", C.c_void_p),
        ("accel_structure", C.c_uint32), ("_reserved", C.c_uint32),
    ]


class QueuePlan(C.Structure):
    _fields_ = [("samples_in_flight", C.c_uint32), ("tiles_per_seg", C.c_uint32), ("nseg", C.c_uint32), ("seg_cap", C.c_uint32),
                ("capacity", C.c_uint64), ("lbuf_entries", C.c_uint64)]


class GmonOptions(C.Structure):
    _fields_ = [("cap", C.c_float)]


class PostOptions(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "exposure", "ca_amount", "ca_green_shift", "contrast", "saturation", "blacks", "shadows", "highlights", "whites",
        "vig_amount", "vig_midpoint", "vig_feather", "vig_power", "vig_roundness")]


class TonemapOptions(C.Structure):
    _fields_ = [
        ("tonemapper", C.c_uint32), ("agx_offset", C.c_float * 3), ("agx_slope", C.c_float * 3), ("agx_power", C.c_float * 3),
        ("agx_saturation", C.c_float), ("khr_compression_start", C.c_float), ("khr_desaturation", C.c_float),
        ("flim_pre_exposure", C.c_float), ("flim_pre_formation_filter", C.c_float * 3), ("flim_pre_formation_filter_strength", C.c_float),
        ("flim_extended_gamut_scale", C.c_float * 3), ("flim_extended_gamut_rotation", C.c_float * 3), ("flim_extended_gamut_mul", C.c_float * 3),
        ("flim_sigmoid_log2_min", C.c_float), ("flim_sigmoid_log2_max", C.c_float), ("flim_sigmoid_toe", C.c_float * 2),
        ("flim_sigmoid_shoulder", C.c_float * 2), ("flim_negative_exposure", C.c_float), ("flim_negative_density", C.c_float),
        ("flim_print_backlight", C.c_float * 3), ("flim_print_exposure", C.c_float), ("flim_print_density", C.c_float),
        ("flim_black_point", C.c_float), ("flim_auto_black_point", C.c_uint32), ("flim_post_formation_filter", C.c_float * 3),
        ("flim_post_formation_filter_strength", C.c_float), ("flim_midtone_saturation", C.c_float),
        ("shadow_color", C.c_float * 3), ("midtone_color", C.c_float * 3), ("highlight_color", C.c_float * 3),
        ("shadow_offset", C.c_float), ("midtone_offset", C.c_float), ("highlight_offset", C.c_float), ("output_space", Colorspace),
    ]


TONEMAP_NONE, TONEMAP_AGX, TONEMAP_KHRONOS_PBR, TONEMAP_FLIM = 0, 1, 2, 3


class HitRecord(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("instance", C.c_int32), ("primitive", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        ("triangles", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_max_depth", C.c_uint32),
        ("samples_in_flight", C.c_uint32), ("upload_ms", C.c_double), ("bvh_build_ms", C.c_double),
        ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shaded_hits", C.c_uint64), ("paths", C.c_uint64),
        ("nonfinite_samples", C.c_uint64), ("ms_raygen", C.c_double), ("ms_closest", C.c_double), ("ms_shade", C.c_double), ("ms_shadow", C.c_double),
        ("ms_accumulate", C.c_double), ("launches_closest", C.c_uint64), ("launches_shadow", C.c_uint64),
        ("nodes_per_closest_ray", C.c_double), ("tris_per_closest_ray", C.c_double),
        ("nodes_per_shadow_ray", C.c_double), ("tris_per_shadow_ray", C.c_double),
        ("accel_two_level", C.c_uint32), ("batches", C.c_uint32), ("leaf_slots", C.c_uint64),
    ]


class RuntimeInfo(C.Structure):
    _fields_ = [
        ("hip_runtime_path", C.c_char * 512), ("hsa_runtime_path", C.c_char * 512), ("rccl_path", C.c_char * 512),
        ("hip_runtimes_mapped", C.c_uint32), ("hsa_runtimes_mapped", C.c_uint32), ("rccl_mapped", C.c_uint32),
        ("hip_runtime_version", C.c_int32), ("all_mapped", C.c_char * 2048),
    ]


assert C.sizeof(Float3) == 16 and C.sizeof(VertexData) == 48 and C.sizeof(MaterialGPU) == 96
assert C.sizeof(Instance) == 64 and C.sizeof(CameraData) == 80 and C.sizeof(Constants) == 176
assert C.sizeof(AreaLight) == 48

# every symbol include/ptamd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("pt_create", C.c_int, [C.POINTER(CreateInfo), C.POINTER(C.c_void_p)]),
    ("pt_group_partition", C.c_int, [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("pt_get_runtime_info", C.c_int, [C.POINTER(RuntimeInfo)]),
    ("pt_rccl_probe", C.c_int, []),
    ("pt_rccl_selftest", C.c_int, [C.c_int32]),
    ("pt_plan_queues", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(QueuePlan)]),
    ("pt_destroy", None, [C.c_void_p]),
    ("pt_start_render", C.c_int, [C.c_void_p, C.POINTER(SceneSnapshot), C.POINTER(RenderParams)]),
    ("pt_render_step", C.c_int, [C.c_void_p, C.c_uint32]),
    ("pt_wait", C.c_int, [C.c_void_p]),
    ("pt_status", C.c_int, [C.c_void_p]),
    ("pt_progress", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pt_render_time_ms", C.c_uint64, [C.c_void_p]),
    ("pt_read_accumulator", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pt_accumulator_device_ptr", C.c_void_p, [C.c_void_p]),
    ("pt_set_gmon_options", C.c_int, [C.c_void_p, C.POINTER(GmonOptions)]),
    ("pt_default_post_options", None, [C.POINTER(PostOptions)]),
    ("pt_default_tonemap_options", None, [C.POINTER(TonemapOptions)]),
    ("pt_set_post_options", C.c_int, [C.c_void_p, C.POINTER(PostOptions)]),
    ("pt_set_tonemap_options", C.c_int, [C.c_void_p, C.POINTER(TonemapOptions)]),
    ("pt_read_render_target", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pt_present_render_target", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("pt_read_gmon_bucket", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    ("pt_last_error", C.c_char_p, []),
    ("pt_get_constants", C.c_int, [C.c_void_p, C.POINTER(Constants)]),
    ("pt_get_lights", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("pt_get_env_alias", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("pt_trace_primary", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    ("pt_debug_sample", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("pt_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("pt_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("pt_measure_traversal", C.c_int, [C.c_void_p, C.c_uint32]),
]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "csrc", "libptamd.so")
LUT_PATH = os.path.join(_PKG_DIR, "data", "ggx_luts.bin")

_lib = None


class PtamdError(RuntimeError):
    pass


def library_path():
    """The libptamd.so that load_library() binds ($PTAMD_LIB selects a variant build)."""
    return os.environ.get("PTAMD_LIB", LIB_PATH)


def _mapped_objects(prefix):
    """Shared objects of this process whose file name starts with `prefix` (from /proc/self/maps)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                parts = line.split(None, 5)
                if len(parts) == 6 and os.path.basename(parts[5].strip()).startswith(prefix) and parts[5].strip() not in found:
                    found.append(parts[5].strip())
    except OSError:
        pass
    return found


def _torch_bundled_hip():
    """Path of the libamdhip64.so PyTorch's ROCm wheel bundles (torch/lib), found WITHOUT importing torch; None if there is none."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    import glob
    for d in spec.submodule_search_locations:
        p = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(p):
            return p
        versioned = sorted(glob.glob(p + ".*"))   # a wheel that ships only the versioned file name
        if versioned:
            return versioned[0]
    return None


def _elf_dynamic_strings(path, tag):
    """DT_SONAME (tag 14) / DT_NEEDED (tag 1) strings of an ELF64 little-endian shared object, read without any tool; [] on any surprise.
    Only the ELF header, the section headers, .dynamic and .dynstr are read (seek): libamdhip64.so is tens of megabytes."""
    import struct
    try:
        with open(path, "rb") as f:
            hdr = f.read(0x40)
            if hdr[:4] != b"\x7fELF" or hdr[4] != 2 or hdr[5] != 1:
                return []
            shoff, = struct.unpack_from("<Q", hdr, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 0x3A)
            if shentsize < 64 or shnum == 0 or shnum > 4096:
                return []
            f.seek(shoff)
            sh = f.read(shentsize * shnum)
            secs = [struct.unpack_from("<IIQQQQIIQQ", sh, i * shentsize) for i in range(shnum)]
            out = []
            for sec in secs:
                if sec[1] != 6:      # SHT_DYNAMIC
                    continue
                strsec = secs[sec[6]]   # sh_link -> .dynstr
                f.seek(sec[4]); dyn = f.read(sec[5])
                f.seek(strsec[4]); strtab = f.read(strsec[5])
                for off in range(0, len(dyn) - 15, 16):
                    t, v = struct.unpack_from("<qQ", dyn, off)
                    if t == 0:
                        break
                    if t == tag:
                        out.append(strtab[v:strtab.index(b"\0", v)].decode())
            return out
    except Exception:
        return []


def _settle_hip_runtime(lib_file=None):
    """Make sure the process ends up with ONE HIP runtime whichever of {libptamd.so, torch} arrives first.

    Root cause (round 4): torch's wheel bundles libamdhip64.so / libhsa-runtime64.so / librccl.so with the same SONAMEs
    as /opt/rocm's but links them by their unversioned file names through RPATH $ORIGIN.  torch first: libptamd.so's DT_NEEDED
    `libamdhip64.so.7` matches the soname of torch's loaded copy - one runtime.  libptamd.so first: /opt/rocm's copy is mapped, and a
    later `import torch` maps ITS copy beside it (DT_NEEDED `libamdhip64.so` matches no loaded soname and $ORIGIN leads to a different
    file): two HIP + two HSA runtimes, and the one that initialises second cannot acquire the GPU VM - torch reports "No HIP GPUs are
    available".  So: when no HIP runtime is mapped yet and this interpreter HAS a torch with a bundled runtime, map that one first
    (RTLD_GLOBAL); libptamd.so then binds to it by soname and a later `import torch` finds the very file already loaded.
    $PTAMD_HIP_RUNTIME=system keeps /opt/rocm's runtime instead (then torch must not be imported in this process);
    the default `auto` does the above.  Returns a short description of what was done."""
    mapped = _mapped_objects("libamdhip64.so")
    if mapped:
        return "already mapped: " + mapped[0]
    mode = os.environ.get("PTAMD_HIP_RUNTIME", "auto")
    if mode == "system":
        return "system ($PTAMD_HIP_RUNTIME)"
    if mode != "auto":
        raise PtamdError(f"$PTAMD_HIP_RUNTIME={mode!r}: expected 'auto' or 'system'")
    bundled = _torch_bundled_hip()
    if bundled is None:
        return "system (no torch with a bundled runtime in this interpreter)"
    # The preload only helps if libptamd.so's DT_NEEDED names the SONAME of torch's copy (libamdhip64.so.7 on both sides in this image).  A
    # torch wheel built for another ROCm major carries another soname: preloading it would put a runtime into the process that libptamd.so
    # does not bind to, /opt/rocm's would be mapped beside it, and every pt_create would fail with "two GPU runtimes" (ADVICE r4).
    soname = _elf_dynamic_strings(bundled, 14)
    needed = [n for n in _elf_dynamic_strings(lib_file or library_path(), 1) if n.startswith("libamdhip64.so")]
    if soname and needed and soname[0] not in needed:
        return ("system (torch bundles %s, libptamd.so needs %s: a different ROCm major - not preloaded; do not import torch in this process, "
                "or build libptamd.so against torch's ROCm)" % (soname[0], needed[0]))
    C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    return "preloaded torch's bundled runtime: " + bundled


_runtime_note = None


def runtime_info(lib=None):
    """pt_get_runtime_info as a dict (paths decoded) + how load_library() settled the HIP runtime."""
    lib = lib or load_library()
    ri = RuntimeInfo()
    check(lib, lib.pt_get_runtime_info(C.byref(ri)))
    d = {n: getattr(ri, n) for n, _ in RuntimeInfo._fields_}
    for k, v in d.items():
        if isinstance(v, bytes):
            d[k] = v.decode()
    d["settled"] = _runtime_note
    return d


def load_library(path=None):
    """Load libptamd.so and bind every declared symbol.  Raises if the library is not built - there is
    deliberately no fallback implementation - or if the process holds two HIP runtimes (see _settle_hip_runtime)."""
    global _lib, _runtime_note
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("PTAMD_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise PtamdError(f"{p} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')")
    note = _settle_hip_runtime(p)   # (the SONAME / DT_NEEDED check reads the library actually being loaded)
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    ri = RuntimeInfo()
    lib.pt_get_runtime_info(C.byref(ri))
    if ri.hip_runtimes_mapped > 1:   # (two libhsa-runtime64 under ONE HIP runtime is what rocprofv3's tool library gives: legitimate)
        raise PtamdError("two GPU runtimes are mapped into this process (" + ri.all_mapped.decode() + "): only the one that initialises "
                         "first would see the GPU.  Import torch, or platinum_amd, before anything else that links /opt/rocm's libamdhip64; "
                         "in a process that never uses torch, $PTAMD_HIP_RUNTIME=system keeps /opt/rocm's runtime alone (how the HIP runtime was settled: "
                         + str(note) + ")")
    if path is None:
        _lib = lib
        _runtime_note = note
    return lib


def check(lib, code):
    if code != 0:
        msg = lib.pt_last_error()
        raise PtamdError(f"ptamd error {code}: {msg.decode() if msg else '?'}")
