"""ctypes mirror of include/ptamd_scene.h — scene ingestion: the reference's scene.json + _data.bin
(core/scene.cpp:30-84, 536-903), a minimal glTF importer (loaders/gltf.cpp) and the flattening into the
`pt_scene_snapshot` that `Renderer.startRender` consumes.  All the work happens in libptamd.so (scene_io.cpp,
scene_gltf.cpp); this file only binds it."""
import ctypes as C

import numpy as np

from . import abi

GLTF_NONE, GLTF_SKIP_EMPTY_NODES, GLTF_CREATE_SCENE_NODES = 0, 1, 2
MTL_R8UNORM, MTL_RG8UNORM, MTL_RGBA8UNORM, MTL_RGBA8UNORM_SRGB, MTL_RGBA32FLOAT = 10, 30, 70, 71, 125


class SceneCounts(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("meshes", C.c_uint32), ("textures", C.c_uint32), ("materials", C.c_uint32),
                ("cameras", C.c_uint32), ("instances", C.c_uint32), ("triangles", C.c_uint64)]


SCENE_SYMBOLS = [
    ("pt_scene_create", C.c_int, [C.POINTER(C.c_void_p)]),
    ("pt_scene_destroy", None, [C.c_void_p]),
    ("pt_scene_load_json", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("pt_scene_save_json", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pt_scene_import_gltf", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("pt_scene_set_environment", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]),
    ("pt_scene_load_environment", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pt_scene_get_counts", C.c_int, [C.c_void_p, C.POINTER(SceneCounts)]),
    ("pt_scene_get_camera", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint32]),
    ("pt_scene_add_camera", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_uint64)]),
    ("pt_scene_build_snapshot", C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(abi.SceneSnapshot))]),
    ("pt_generate_tangents", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    ("pt_decode_image_rgba8", C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]),
    ("pt_scene_last_error", C.c_char_p, []),
]

_bound = None


def _lib():
    global _bound
    if _bound is None:
        lib = abi.load_library()
        for name, restype, argtypes in SCENE_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = lib
    return _bound


def _check(code):
    if code != 0:
        raise abi.PtamdError(f"ptamd scene error {code}: {_lib().pt_scene_last_error().decode()}")


class _SnapshotView:
    """What Renderer.startRender / the test oracle expect from `scene.snapshot()`: an object with `.struct`."""

    def __init__(self, owner, ptr):
        self._owner = owner            # keeps the pt_scene (which owns every array) alive
        self.struct = ptr.contents


class SceneFile:
    """A loaded / imported scene (pt_scene).  `camera` selects the camera node used by snapshot()."""

    def __init__(self, handle):
        self._h = handle
        self.camera = None

    @staticmethod
    def empty():
        h = C.c_void_p()
        _check(_lib().pt_scene_create(C.byref(h)))
        return SceneFile(h)

    @staticmethod
    def load(json_path):                                   # Scene::Scene(path, device), core/scene.cpp:30-84
        h = C.c_void_p()
        _check(_lib().pt_scene_load_json(str(json_path).encode(), C.byref(h)))
        return SceneFile(h)

    def save(self, json_path):                             # Scene::saveToFile, core/scene.cpp:536-631
        _check(_lib().pt_scene_save_json(self._h, str(json_path).encode()))

    def import_gltf(self, path, options=GLTF_NONE):        # GltfLoader::load, loaders/gltf.cpp:28-113
        _check(_lib().pt_scene_import_gltf(self._h, str(path).encode(), options))
        return self

    def set_environment(self, rgba, name="environment"):
        px = np.ascontiguousarray(rgba, dtype=np.float32)
        assert px.ndim == 3 and px.shape[2] == 4
        _check(_lib().pt_scene_set_environment(self._h, px.ctypes.data, px.shape[1], px.shape[0], name.encode()))

    def load_environment(self, path):
        """An .exr (as tinyexr LoadEXR) or Radiance .hdr (as stbi_loadf) file becomes the scene environment."""
        _check(_lib().pt_scene_load_environment(self._h, str(path).encode()))

    def counts(self):
        c = SceneCounts()
        _check(_lib().pt_scene_get_counts(self._h, C.byref(c)))
        return c

    def cameras(self):
        out = []
        for i in range(self.counts().cameras):
            nid, name = C.c_uint64(), C.create_string_buffer(256)
            _check(_lib().pt_scene_get_camera(self._h, i, C.byref(nid), name, 256))
            out.append((nid.value, name.value.decode()))
        return out

    def add_camera(self, position, target, focal_length=28.0, name="Camera"):
        nid = C.c_uint64()
        p, t = (C.c_float * 3)(*position), (C.c_float * 3)(*target)
        _check(_lib().pt_scene_add_camera(self._h, name.encode(), p, t, focal_length, C.byref(nid)))
        self.camera = nid.value
        return nid.value

    def snapshot(self):
        cam = self.camera
        if cam is None:
            cams = self.cameras()
            if not cams:
                raise abi.PtamdError("scene has no camera node: add_camera() first")
            cam = cams[0][0]
        ptr = C.POINTER(abi.SceneSnapshot)()
        _check(_lib().pt_scene_build_snapshot(self._h, cam, C.byref(ptr)))
        return _SnapshotView(self, ptr)

    def close(self):
        if self._h:
            _lib().pt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_tangents(positions, vertex_data, indices):
    """MikkTSpace tangents written into vertex_data[:, 4:8] (in place), as core/mesh.cpp:135-157."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    assert vertex_data.dtype == np.float32 and vertex_data.flags["C_CONTIGUOUS"] and vertex_data.shape[1] == 12
    _check(_lib().pt_generate_tangents(pos.ctypes.data, vertex_data.ctypes.data, len(pos), idx.ctypes.data, len(idx) // 3))
    return vertex_data


def decode_image_rgba8(data):
    """bytes of a PNG / JPEG file -> (H, W, 4) uint8, as stbi_load_from_memory(..., 4) (loaders/texture.cpp:111-119)."""
    w, h = C.c_uint32(), C.c_uint32()
    _check(_lib().pt_decode_image_rgba8(data, len(data), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    _check(_lib().pt_decode_image_rgba8(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes))
    return out
