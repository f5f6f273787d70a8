"""Host-side scene model: the part of the reference's `Store`/`Scene` the path tracer consumes, restated in
numpy so that headless drivers (tests, bench) can hand the renderer the same flattened snapshot the
SDL2/ImGui frontend would.

Mirrors (by behaviour, fp32 throughout):
  core/primitives.cpp:7-190        plane / cube / sphere / cornellBox
  core/transform.hpp:36-51         Transform::matrix()  (T*Ry*Rx*Rz*S, or inverse(lookAt)*S when tracking)
  utils/matrices.cpp:9-145         translation / rotation_* / scaling / lookAt
  core/camera.hpp:10-51            Camera
  frontend/windows/scene_explorer.cpp:50-90   the "Cornell Box" and "Camera" menu entries
  core/scene.cpp:479-534           getInstances(): depth-first hierarchy walk -> Instance{mesh, worldMatrix}
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi

f32 = np.float32

# core/colorspace.cpp:5-7, colorspace.hpp:47
WHITEPOINT_D65 = (0.3127, 0.3290)
BT709 = ((0.640, 0.330), (0.300, 0.600), (0.150, 0.060), WHITEPOINT_D65)
DISPLAY_P3 = ((0.680, 0.320), (0.265, 0.690), (0.150, 0.060), WHITEPOINT_D65)
BT2020 = ((0.708, 0.292), (0.170, 0.797), (0.131, 0.046), WHITEPOINT_D65)


def colorspace(cs=BT2020):
    out = abi.Colorspace()
    for name, v in zip("rgbw", cs):
        getattr(out, name)[0] = v[0]
        getattr(out, name)[1] = v[1]
    return out


# ---------------------------------------------------------------------------------------------------------------
# matrices (column-major 4x4 like simd: M[c] is column c) — utils/matrices.cpp
# ---------------------------------------------------------------------------------------------------------------
def _cols(*cols):
    return np.array(cols, dtype=f32)  # shape (4,4): [column][row]


def mat_identity():
    return np.eye(4, dtype=f32)


def mat_translation(t):
    return _cols([1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [t[0], t[1], t[2], 1])


def mat_scaling(s):
    s = np.broadcast_to(np.asarray(s, dtype=f32), (3,))
    return _cols([s[0], 0, 0, 0], [0, s[1], 0, 0], [0, 0, s[2], 0], [0, 0, 0, 1])


def mat_rotation_x(a):
    c, s = f32(np.cos(f32(a))), f32(np.sin(f32(a)))
    return _cols([1, 0, 0, 0], [0, c, s, 0], [0, -s, c, 0], [0, 0, 0, 1])


def mat_rotation_y(a):
    c, s = f32(np.cos(f32(a))), f32(np.sin(f32(a)))
    return _cols([c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1])


def mat_rotation_z(a):
    c, s = f32(np.cos(f32(a))), f32(np.sin(f32(a)))
    return _cols([c, s, 0, 0], [-s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1])


def mat_mul(a, b):
    """simd a * b for column-major storage [col][row]."""
    # (a*b)[c][r] = sum_k a[k][r] * b[c][k]
    return np.einsum("kr,ck->cr", a.astype(f32), b.astype(f32)).astype(f32)


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    return (v / f32(np.sqrt(np.dot(v, v)))).astype(f32)


def mat_look_at(position, target, up):  # utils/matrices.cpp:132-145
    position = np.asarray(position, dtype=f32)
    target = np.asarray(target, dtype=f32)
    if np.array_equal(position, target):
        return mat_identity()
    f = _normalize(position - target)
    s = _normalize(np.cross(np.asarray(up, dtype=f32), f))
    u = np.cross(f, s).astype(f32)
    return _cols([s[0], u[0], f[0], 0], [s[1], u[1], f[1], 0], [s[2], u[2], f[2], 0],
                 [-np.dot(s, position), -np.dot(u, position), -np.dot(f, position), 1])


def mat_inverse(m):
    # [col][row] storage is the transpose of the usual row-major matrix; inverse commutes with transpose.
    return np.linalg.inv(m.astype(np.float64)).astype(f32)


@dataclass
class Transform:  # core/transform.hpp:19-51
    translation: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (0.0, 0.0, 0.0)
    scale: tuple = (1.0, 1.0, 1.0)
    target: tuple = (0.0, 0.0, 0.0)
    track: bool = False

    def matrix(self):
        S = mat_scaling(self.scale)
        if self.track:
            t, g = self.translation, self.target
            up = (0, 0, 1) if (t[0] == g[0] and t[2] == g[2]) else (0, 1, 0)
            L = mat_inverse(mat_look_at(t, g, up))
            return mat_mul(L, S)
        T = mat_translation(self.translation)
        Rx, Ry, Rz = mat_rotation_x(self.rotation[0]), mat_rotation_y(self.rotation[1]), mat_rotation_z(self.rotation[2])
        return mat_mul(mat_mul(mat_mul(mat_mul(T, Ry), Rx), Rz), S)


@dataclass
class Camera:  # core/camera.hpp:10-18
    sensor_size: tuple = (36.0, 24.0)
    focal_length: float = 50.0
    aperture: float = 0.0
    aperture_blades: int = 7
    roundness: float = 1.0
    bokeh_power: float = 0.0
    focus_distance: float = 1.0

    @staticmethod
    def with_focal_length(f, sensor_size=(36.0, 24.0), aperture=0.0):  # camera.hpp:20-30
        return Camera(sensor_size=sensor_size, focal_length=f, aperture=aperture)


@dataclass
class Material:  # core/material.hpp:15-49 (texture slots are a "next" row)
    name: str = ""
    base_color: tuple = (0.8, 0.8, 0.8, 1.0)
    emission: tuple = (0.0, 0.0, 0.0)
    emission_strength: float = 0.0
    roughness: float = 1.0
    metallic: float = 0.0
    transmission: float = 0.0
    ior: float = 1.5
    anisotropy: float = 0.0
    anisotropy_rotation: float = 0.0
    clearcoat: float = 0.0
    clearcoat_roughness: float = 0.05
    thin_transmission: bool = False
    # texture slots (material.hpp:16-23): indices into Scene.textures, -1 = none
    base_texture: int = -1
    rm_texture: int = -1
    transmission_texture: int = -1
    clearcoat_texture: int = -1
    emission_texture: int = -1
    normal_texture: int = -1
    base_texture_has_alpha: bool = False

    def is_emissive(self):  # material.hpp:44-47
        e = np.asarray(self.emission, dtype=f32) * f32(self.emission_strength)
        return float(np.dot(e, e)) > 0.0 or self.emission_texture >= 0

    def to_gpu(self):  # renderer_pt.cpp:583-633
        m = abi.MaterialGPU()
        for i in range(4):
            m.baseColor[i] = self.base_color[i]
        m.emission = abi.Float3(self.emission[0], self.emission[1], self.emission[2], 0.0)
        m.emissionStrength = self.emission_strength
        m.roughness, m.metallic, m.transmission, m.ior = self.roughness, self.metallic, self.transmission, self.ior
        m.anisotropy, m.anisotropyRotation = self.anisotropy, self.anisotropy_rotation
        m.clearcoat, m.clearcoatRoughness = self.clearcoat, self.clearcoat_roughness
        flags = 0
        if self.thin_transmission:
            flags |= abi.MATERIAL_THIN_DIELECTRIC
        if self.base_color[3] < 1.0 or (self.base_texture >= 0 and self.base_texture_has_alpha):  # renderer_pt.cpp:628-630
            flags |= abi.MATERIAL_USE_ALPHA
        if self.anisotropy != 0.0:
            flags |= abi.MATERIAL_ANISOTROPIC
        if self.is_emissive():
            flags |= abi.MATERIAL_EMISSIVE
        m.flags = flags
        m.baseTextureId, m.rmTextureId, m.transmissionTextureId = self.base_texture, self.rm_texture, self.transmission_texture
        m.clearcoatTextureId, m.emissionTextureId, m.normalTextureId = self.clearcoat_texture, self.emission_texture, self.normal_texture
        return m


@dataclass
class MeshData:  # core/mesh.hpp:23-60 — the four shared buffers
    positions: np.ndarray      # (V,4) f32, 16-byte float3
    vertex_data: np.ndarray    # (V,12) f32, 48-byte VertexData
    indices: np.ndarray        # (3T,) u32
    material_slots: np.ndarray # (T,) u32

    @property
    def triangle_count(self):
        return len(self.indices) // 3


def _make_mesh(vertices, normals, tangents, uvs, indices, mat_indices):
    v = np.zeros((len(vertices), 4), dtype=f32)
    v[:, :3] = np.asarray(vertices, dtype=f32)
    vd = np.zeros((len(vertices), 12), dtype=f32)
    vd[:, 0:3] = np.asarray(normals, dtype=f32)
    vd[:, 4:8] = np.asarray(tangents, dtype=f32)
    vd[:, 8:10] = np.asarray(uvs, dtype=f32)
    return MeshData(v, vd, np.asarray(indices, dtype=np.uint32), np.asarray(mat_indices, dtype=np.uint32))


_FACE_POS = np.array([[1, -1], [1, 1], [-1, -1], [-1, 1]], dtype=f32)


def plane(side):  # primitives.cpp:7-29
    h = f32(side) * f32(0.5)
    verts = np.array([[-h, 0, -h], [h, 0, -h], [-h, 0, h], [h, 0, h]], dtype=f32)
    uvs = (verts[:, [0, 2]] + h) / (f32(2.0) * h)
    return _make_mesh(verts, [[0, 1, 0]] * 4, [[1, 0, 0, 0]] * 4, uvs, [0, 2, 1, 1, 2, 3], [0, 0])


def _box_faces(face_normals, h, sign, offset):
    verts, normals, tangents, uvs, indices = [], [], [], [], []
    for i, fn in enumerate(face_normals):
        fn = np.asarray(fn, dtype=f32)
        up = np.array([1, 0, 0], dtype=f32) if abs(fn[1]) == 1.0 else np.array([0, 1, 0], dtype=f32)
        right = np.cross(up, fn).astype(f32)
        for fp in _FACE_POS:
            verts.append((sign * fn + up * fp[0] + right * fp[1]) * f32(h) + np.asarray(offset, dtype=f32))
            normals.append(fn)
            tangents.append([right[0], right[1], right[2], 1.0])
            uvs.append(fp)
        indices += [4 * i + 0, 4 * i + 2, 4 * i + 1, 4 * i + 1, 4 * i + 2, 4 * i + 3]
    return verts, normals, tangents, uvs, indices


def cube(side):  # primitives.cpp:31-77
    fns = [[0, 0, 1], [1, 0, 0], [0, 0, -1], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
    v, n, t, uv, idx = _box_faces(fns, f32(side) * f32(0.5), f32(1.0), [0, 0, 0])
    return _make_mesh(v, n, t, uv, idx, [0] * 12)


def sphere(radius, lat, lng):  # primitives.cpp:79-130
    pi = f32(np.pi)
    d_lat = pi / f32(lat)
    d_lng = pi / f32(lng) * f32(2.0)
    i = np.arange(lat + 1, dtype=f32)
    j = np.arange(lng + 1, dtype=f32)
    phi = (f32(0.5) * pi - i * d_lat).astype(f32)
    theta = (j * d_lng).astype(f32)
    c = np.cos(phi).astype(f32)
    pos = np.stack([np.outer(c, np.cos(theta).astype(f32)), np.repeat(np.sin(phi).astype(f32)[:, None], lng + 1, 1),
                    np.outer(c, np.sin(theta).astype(f32))], axis=-1).astype(f32).reshape(-1, 3)
    tang = np.stack([np.tile(-np.sin(theta), lat + 1), np.zeros((lat + 1) * (lng + 1)), np.tile(np.cos(theta), lat + 1),
                     np.ones((lat + 1) * (lng + 1))], axis=-1).astype(f32)
    uv = np.stack([np.tile(j / f32(lng), lat + 1), np.repeat(i / f32(lat), lng + 1)], axis=-1).astype(f32)
    ii, jj = np.meshgrid(np.arange(1, lat + 1), np.arange(1, lng + 1), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    v0 = (ii - 1) * (lng + 1) + (jj - 1)
    v1 = (ii - 1) * (lng + 1) + jj
    v2 = ii * (lng + 1) + (jj - 1)
    v3 = ii * (lng + 1) + jj
    indices = np.stack([v0, v1, v2, v1, v3, v2], axis=-1).ravel()
    return _make_mesh(pos * f32(radius), pos, tang, uv, indices, np.zeros(lat * lng * 2))


def cornell_box():  # primitives.cpp:133-190
    h = f32(5.0)
    fns = [[0, 0, 1], [0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0]]
    v, n, t, uv, idx = _box_faces(fns, h, f32(-1.0), [0, h, 0])
    mat = []
    for i in range(5):
        mat += [0 if i < 3 else i - 2] * 2
    for fp in _FACE_POS:
        v.append(np.array([fp[0], f32(2) * h - f32(0.01), fp[1]], dtype=f32))
        n.append([0, -1, 0])
        t.append([0, 0, 1, 1])
        uv.append(fp)
    idx += [20, 22, 21, 21, 22, 23]
    mat += [3, 3]
    return _make_mesh(v, n, t, uv, idx, mat)


def cornell_materials():  # scene_explorer.cpp:52-67
    return [
        Material(name="cornell_base", base_color=(1, 1, 1, 1)),
        Material(name="cornell_wall_l", base_color=(0.704, 0.016, 0.020, 1)),
        Material(name="cornell_wall_r", base_color=(0.009, 0.591, 0.006, 1)),
        Material(name="cornell_base", base_color=(0, 0, 0, 1), emission=(1, 1, 1), emission_strength=50.0),
    ]


# ---------------------------------------------------------------------------------------------------------------
# Scene container -> pt_scene_snapshot
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class Node:
    mesh: int                    # index into Scene.meshes
    world: np.ndarray            # 4x4 [col][row] world matrix (scene.cpp:527)
    materials: list              # one Material per slot of the mesh


@dataclass
class TextureData:  # core/texture.hpp + loaders/texture.cpp:30-48 (pixel format per TextureType)
    pixels: np.ndarray  # (H, W, C) uint8 or float32
    format: int         # abi.TEX_*


@dataclass
class Scene:
    meshes: list = field(default_factory=list)
    nodes: list = field(default_factory=list)
    textures: list = field(default_factory=list)
    env_texture: int = -1
    env_alias: object = None     # optional host-built alias table, numpy array of abi.ALIAS_DTYPE (width*height entries)
    camera: Camera = field(default_factory=Camera)
    camera_world: np.ndarray = field(default_factory=mat_identity)
    name: str = "scene"

    def add_texture(self, pixels, fmt):
        chans = {abi.TEX_RGBA8_SRGB: 4, abi.TEX_RGBA8: 4, abi.TEX_RG8: 2, abi.TEX_R8: 1, abi.TEX_RGBA32F: 4}[fmt]
        dt = np.float32 if fmt == abi.TEX_RGBA32F else np.uint8
        px = np.ascontiguousarray(pixels, dtype=dt)
        if px.ndim == 2:
            px = px[..., None]
        assert px.ndim == 3 and px.shape[2] == chans, (px.shape, chans)
        self.textures.append(TextureData(px, fmt))
        return len(self.textures) - 1

    def add_mesh(self, mesh):
        self.meshes.append(mesh)
        return len(self.meshes) - 1

    def add_instance(self, mesh_idx, transform, materials):
        world = transform.matrix() if isinstance(transform, Transform) else np.asarray(transform, dtype=f32)
        self.nodes.append(Node(mesh_idx, world, list(materials)))

    def set_camera(self, camera, transform):
        self.camera = camera
        self.camera_world = transform.matrix() if isinstance(transform, Transform) else np.asarray(transform, dtype=f32)

    @property
    def triangle_count(self):
        return sum(self.meshes[n.mesh].triangle_count for n in self.nodes)

    def snapshot(self):
        """Build the ctypes pt_scene_snapshot. The returned object keeps every buffer alive."""
        return Snapshot(self)


class Snapshot:
    def __init__(self, scene):
        self._keep = []
        nm, ni = len(scene.meshes), len(scene.nodes)
        self.meshes = (abi.Mesh * nm)()
        for k, m in enumerate(scene.meshes):
            arrs = [np.ascontiguousarray(m.positions, dtype=f32), np.ascontiguousarray(m.vertex_data, dtype=f32),
                    np.ascontiguousarray(m.indices, dtype=np.uint32), np.ascontiguousarray(m.material_slots, dtype=np.uint32)]
            self._keep += arrs
            self.meshes[k].positions = arrs[0].ctypes.data
            self.meshes[k].vertex_data = arrs[1].ctypes.data
            self.meshes[k].indices = arrs[2].ctypes.data
            self.meshes[k].material_slots = arrs[3].ctypes.data
            self.meshes[k].vertex_count = len(arrs[0])
            self.meshes[k].triangle_count = len(arrs[2]) // 3
        self.instances = (abi.Instance * ni)()
        self.instance_materials = (abi.InstanceMaterials * ni)()
        for k, n in enumerate(scene.nodes):
            for c in range(4):
                for r in range(3):
                    self.instances[k].transform[c][r] = n.world[c][r]
            self.instances[k].options = 0
            self.instances[k].mask = 0xFF
            self.instances[k].intersectionFunctionTableOffset = 0
            self.instances[k].accelerationStructureIndex = n.mesh
            mats = (abi.MaterialGPU * len(n.materials))(*[m.to_gpu() for m in n.materials])
            self._keep.append(mats)
            self.instance_materials[k].materials = C.addressof(mats)
            self.instance_materials[k].material_count = len(n.materials)
        nt = len(scene.textures)
        self.textures = (abi.Texture * max(1, nt))()
        for k, t in enumerate(scene.textures):
            self._keep.append(t.pixels)
            self.textures[k].pixels = t.pixels.ctypes.data
            self.textures[k].height, self.textures[k].width = t.pixels.shape[0], t.pixels.shape[1]
            self.textures[k].format = t.format
        self.struct = abi.SceneSnapshot()
        self.struct.textures = C.addressof(self.textures) if nt else None
        self.struct.texture_count = nt
        self.struct.env_texture = scene.env_texture
        self.struct.env_alias = None
        if scene.env_alias is not None:
            al = np.ascontiguousarray(scene.env_alias, dtype=abi.ALIAS_DTYPE)
            self._keep.append(al)
            self.struct.env_alias = al.ctypes.data
        self.struct.meshes = C.addressof(self.meshes)
        self.struct.mesh_count = nm
        self.struct.instance_count = ni
        self.struct.instances = C.addressof(self.instances)
        self.struct.instance_materials = C.addressof(self.instance_materials)
        cam = self.struct.camera
        for c in range(4):
            for r in range(4):
                cam.world[c][r] = scene.camera_world[c][r]
        cam.sensor_size[0], cam.sensor_size[1] = scene.camera.sensor_size
        cam.focal_length = scene.camera.focal_length
        cam.aperture = scene.camera.aperture
        cam.aperture_blades = scene.camera.aperture_blades
        cam.roundness = scene.camera.roundness
        cam.bokeh_power = scene.camera.bokeh_power
        cam.focus_distance = scene.camera.focus_distance


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs (all procedural; integer-hash seeded; no external files)
# ---------------------------------------------------------------------------------------------------------------
def _pcg4d(v):
    """samplers.metal:16-23 on a python tuple of 4 u32 (used to seed procedural scenes)."""
    M = 0xFFFFFFFF
    x, y, z, w = [(a * 1664525 + 1013904223) & M for a in v]
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16
    x = (x + y * w) & M; y = (y + z * x) & M; z = (z + x * y) & M; w = (w + y * z) & M
    return x, y, z, w


def cornell_scene(camera="bench"):
    """C1: the Cornell box of scene_explorer.cpp:50-73.
    camera="default": the reference's Camera menu entry (translation (-5,5,5), track origin, f=28mm)
    (scene_explorer.cpp:84-90) — it sits outside the open +z face, off to the side.
    camera="bench":   (0,5,15) looking at (0,5,0), the framing stated in BASELINE.md §3."""
    sc = Scene(name="cornell")
    box = sc.add_mesh(cornell_box())
    sc.add_instance(box, Transform(), cornell_materials())
    if camera == "default":
        sc.set_camera(Camera.with_focal_length(28.0), Transform(translation=(-5, 5, 5), track=True))
    else:
        sc.set_camera(Camera.with_focal_length(28.0), Transform(translation=(0, 5, 15), target=(0, 5, 0), track=True))
    return sc


def cornell_sphere_scene(transmission=1.0, roughness=0.3):
    """C2: Cornell box + one GGX dielectric object.  Suzanne is not in the reference tree; the stand-in is the
    reference's own sphere primitive, primitives::sphere(1, 48, 64) = 6144 triangles (scene_explorer.cpp:46-47),
    scaled x2 and resting on the floor; roughness 0.3, ior 1.5."""
    sc = cornell_scene("bench")
    sc.name = "cornell_sphere"
    sph = sc.add_mesh(sphere(1.0, 48, 64))
    sc.add_instance(sph, Transform(translation=(0.5, 2.0, 0.0), scale=(2, 2, 2)),
                    [Material(name="glass", base_color=(1, 1, 1, 1), roughness=roughness, ior=1.5, transmission=transmission)])
    return sc


def field_scene(grid=32):
    """C3/C4: the 1.04 M-triangle instanced mesh field: sphere(0.25, 22, 23) = 1012 triangles x grid^2 instances
    on a 64x64 floor inside a Cornell shell enlarged x6.4; per-instance material from pcg4d(i,0,0,0)."""
    sc = Scene(name=f"field{grid}")
    shell = sc.add_mesh(cornell_box())
    sc.add_instance(shell, Transform(scale=(6.4, 6.4, 6.4)), cornell_materials())
    ball = sc.add_mesh(sphere(0.25, 22, 23))
    span = 64.0
    step = span / grid
    for i in range(grid * grid):
        gx, gz = i % grid, i // grid
        h = _pcg4d((i, 0, 0, 0))
        u = [(c >> 8) / float(1 << 24) for c in h]
        kind = h[3] % 8
        color = (0.25 + 0.7 * u[0], 0.25 + 0.7 * u[1], 0.25 + 0.7 * u[2], 1.0)
        if kind < 4:
            mat = Material(base_color=color, roughness=1.0)
        elif kind < 6:
            mat = Material(base_color=color, roughness=0.15 + 0.5 * u[0], metallic=1.0)
        elif kind == 6:
            mat = Material(base_color=color, roughness=0.25, ior=1.5)
        else:
            mat = Material(base_color=(1, 1, 1, 1), roughness=0.2, ior=1.5, transmission=1.0)
        s = 2.0 + 1.5 * u[1]                      # radius 0.5 .. 0.875
        x = -span / 2 + (gx + 0.5) * step
        z = -span / 2 + (gz + 0.5) * step
        y = 0.25 * s + 3.0 * u[2] * (1 if kind >= 4 else 0)
        sc.add_instance(ball, Transform(translation=(x, y, z), scale=(s, s, s)), [mat])
    sc.set_camera(Camera.with_focal_length(28.0), Transform(translation=(0, 14, 31.5), target=(0, 2, 0), track=True))
    return sc


def _hash_bytes(shape, seed):
    """Deterministic u8 noise from pcg4d (no numpy RNG: the fixtures must not depend on a library's generator)."""
    n = int(np.prod(shape))
    out = np.zeros(n, dtype=np.uint8)
    for i in range(0, n, 16):
        h = _pcg4d((i, seed, 0, 0))
        b = b"".join(int(c).to_bytes(4, "little") for c in h)
        out[i:i + 16] = np.frombuffer(b, dtype=np.uint8)[: n - i]
    return out.reshape(shape)


def sky_environment(width=32, height=16, sun=(9, 4), sun_radiance=60.0):
    """A procedural RGBA32F lat-long environment: blue-to-white sky gradient, dim ground, one bright sun texel block."""
    y = (np.arange(height, dtype=f32) + f32(0.5)) / f32(height)
    x = (np.arange(width, dtype=f32) + f32(0.5)) / f32(width)
    t = np.clip((f32(0.5) - y) * f32(2.0), 0, 1).astype(f32)[:, None]
    px = np.zeros((height, width, 4), dtype=f32)
    px[..., 0] = f32(0.9) - f32(0.6) * t
    px[..., 1] = f32(0.9) - f32(0.35) * t
    px[..., 2] = f32(1.0) - f32(0.05) * t
    px[..., :3] *= (f32(0.6) + f32(0.4) * np.cos(x * f32(6.2831853))[None, :, None]).astype(f32)
    px[y > 0.5] *= f32(0.15)
    sx, sy = sun
    px[sy:sy + 2, sx:sx + 2, :3] = f32(sun_radiance)
    px[..., 3] = 1.0
    return px


def textured_scene(env=True, area_light=True, alpha=True):
    """N3 test scene: every texture slot of core/material.hpp:16-23, a normal map, cut-out (alpha-tested) geometry,
    an emission-textured area light and (optionally) an importance-sampled environment map."""
    sc = Scene(name="textured")
    # --- textures ---
    chk = np.zeros((16, 16, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:16, 0:16]
    on = ((xx // 4 + yy // 4) % 2).astype(bool)
    chk[..., 0] = np.where(on, 220, 40); chk[..., 1] = np.where(on, 180, 60); chk[..., 2] = np.where(on, 60, 200); chk[..., 3] = 255
    chk[..., :3] = (chk[..., :3].astype(np.int32) + (_hash_bytes((16, 16, 3), 1) >> 3)).clip(0, 255).astype(np.uint8)
    t_base = sc.add_texture(chk, abi.TEX_RGBA8_SRGB)
    t_rm = sc.add_texture(np.maximum(_hash_bytes((8, 8, 2), 2), 24), abi.TEX_RG8)
    nrm = np.zeros((16, 16, 4), dtype=np.uint8)
    nrm[..., 0] = 128 + (np.sin(xx * 0.8) * 50).astype(np.int32)
    nrm[..., 1] = 128 + (np.cos(yy * 0.6) * 50).astype(np.int32)
    nrm[..., 2] = 230; nrm[..., 3] = 255
    t_nrm = sc.add_texture(nrm, abi.TEX_RGBA8)
    cut = np.zeros((32, 32, 4), dtype=np.uint8)
    yy2, xx2 = np.mgrid[0:32, 0:32]
    r2 = (xx2 % 16 - 7.5) ** 2 + (yy2 % 16 - 7.5) ** 2
    cut[..., 0] = 200; cut[..., 1] = 90; cut[..., 2] = 50
    cut[..., 3] = np.clip(255 - r2 * 6.0, 0, 255).astype(np.uint8)  # soft holes: exercises the stochastic test
    t_cut = sc.add_texture(cut, abi.TEX_RGBA8_SRGB)
    t_tr = sc.add_texture(_hash_bytes((8, 8), 3), abi.TEX_R8)
    t_cc = sc.add_texture(_hash_bytes((4, 4), 4), abi.TEX_R8)
    em = np.zeros((4, 4, 4), dtype=np.uint8)
    em[..., :3] = 64 + (_hash_bytes((4, 4, 3), 5) // 2); em[..., 3] = 255
    t_em = sc.add_texture(em, abi.TEX_RGBA8_SRGB)
    # --- geometry ---
    quad = sc.add_mesh(plane(1.0))
    ball = sc.add_mesh(sphere(1.0, 16, 24))
    sc.add_instance(quad, Transform(scale=(12, 1, 12)),
                    [Material(name="floor", base_texture=t_base, rm_texture=t_rm, normal_texture=t_nrm, roughness=0.9, metallic=0.6)])
    if alpha:
        sc.add_instance(quad, Transform(translation=(-1.0, 2.0, 1.5), rotation=(np.pi / 2, 0, 0), scale=(4, 1, 4)),
                        [Material(name="cutout", base_texture=t_cut, base_texture_has_alpha=True, roughness=0.6)])
        sc.add_instance(ball, Transform(translation=(2.6, 0.8, 2.0), scale=(0.8, 0.8, 0.8)),
                        [Material(name="ghost", base_color=(0.2, 0.8, 0.3, 0.5), roughness=0.4)])
    sc.add_instance(ball, Transform(translation=(-0.5, 1.3, -1.0), scale=(1.3, 1.3, 1.3)),
                    [Material(name="frosted", base_color=(1, 1, 1, 1), roughness=0.25, ior=1.45, transmission=1.0,
                              transmission_texture=t_tr, clearcoat=1.0, clearcoat_texture=t_cc, rm_texture=t_rm)])
    sc.add_instance(ball, Transform(translation=(2.4, 1.0, -1.5)),
                    [Material(name="bumpy metal", base_texture=t_base, normal_texture=t_nrm, roughness=0.35, metallic=1.0)])
    if area_light:
        sc.add_instance(quad, Transform(translation=(0, 5.5, 0), rotation=(np.pi, 0, 0), scale=(3, 1, 3)),
                        [Material(name="panel", base_color=(0, 0, 0, 1), emission=(1.0, 0.9, 0.8), emission_strength=6.0,
                                  emission_texture=t_em)])
    if env:
        sc.env_texture = sc.add_texture(sky_environment(), abi.TEX_RGBA32F)
    sc.set_camera(Camera.with_focal_length(28.0), Transform(translation=(0.5, 3.0, 8.0), target=(0, 1.2, 0), track=True))
    return sc


def random_scene(seed, textures=True, extras=False):
    """A seeded random scene for parity fuzzing: 2-10 instances of the reference primitives under arbitrary TRS (non-uniform and
    mirrored scales included), materials drawn over the whole parameter space of core/material.hpp (exact 0 / 1 corner values
    over-represented: smooth surfaces, pure metal, pure glass), random light panels, optional textures and environment.
    `extras` adds, from a second generator (the base scene of a seed stays what it was): the transmission / clearcoat / emission
    texture slots, anisotropy rotation, emission-textured lights, degenerate material corners (ior 1, black base colour)."""
    rng = np.random.default_rng(seed)
    rx = np.random.default_rng(seed + 1000003) if extras else None
    sc = Scene(name=f"random{seed}")
    meshes = [sc.add_mesh(plane(2.0)), sc.add_mesh(cube(1.0)), sc.add_mesh(sphere(0.6, 8, 12))]
    tex = []
    if textures and rng.random() < 0.7:
        tex.append(sc.add_texture(rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), abi.TEX_RGBA8_SRGB))
        tex.append(sc.add_texture(rng.integers(8, 256, (4, 8, 2), dtype=np.uint8), abi.TEX_RG8))
        nm = rng.integers(96, 160, (8, 8, 4), dtype=np.uint8); nm[..., 2] = 230
        tex.append(sc.add_texture(nm, abi.TEX_RGBA8))
    xtex = []
    if rx is not None and rx.random() < 0.7:
        xtex.append(sc.add_texture(rx.integers(0, 256, (8, 4), dtype=np.uint8), abi.TEX_R8))        # transmission
        xtex.append(sc.add_texture(rx.integers(0, 256, (4, 4), dtype=np.uint8), abi.TEX_R8))        # clearcoat
        em = rx.integers(0, 256, (4, 8, 4), dtype=np.uint8); em[..., 3] = 255
        xtex.append(sc.add_texture(em, abi.TEX_RGBA8_SRGB))                                         # emission

    def corner(lo=0.0, hi=1.0):
        u = rng.random()
        return lo if u < 0.25 else (hi if u < 0.45 else float(rng.uniform(lo, hi)))

    def material():
        m = Material(base_color=(float(rng.uniform(0.05, 1)), float(rng.uniform(0.05, 1)), float(rng.uniform(0.05, 1)),
                                 1.0 if rng.random() < 0.8 else float(rng.uniform(0.2, 0.9))),
                     roughness=corner(), metallic=corner(), transmission=corner(), ior=float(rng.uniform(1.05, 2.2)),
                     anisotropy=0.0 if rng.random() < 0.6 else float(rng.uniform(-0.9, 0.9)), clearcoat=0.0 if rng.random() < 0.6 else float(rng.uniform(0.1, 1)),
                     clearcoat_roughness=corner(0.0, 0.6), thin_transmission=bool(rng.random() < 0.2))
        if rng.random() < 0.2:
            m.emission = (float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1))); m.emission_strength = float(rng.uniform(0.5, 8))
        if tex and rng.random() < 0.5:
            m.base_texture = tex[0]; m.base_texture_has_alpha = bool(rng.random() < 0.3)
            if rng.random() < 0.5: m.rm_texture = tex[1]
            if rng.random() < 0.5: m.normal_texture = tex[2]
        if rx is not None:
            if m.anisotropy != 0.0 and rx.random() < 0.7: m.anisotropy_rotation = float(rx.uniform(0, 1))
            if xtex and rx.random() < 0.3: m.transmission_texture = xtex[0]
            if xtex and rx.random() < 0.3: m.clearcoat_texture = xtex[1]; m.clearcoat = max(m.clearcoat, 0.5)
            if xtex and rx.random() < 0.15:
                m.emission_texture = xtex[2]
                if m.emission_strength == 0.0: m.emission = (1.0, 1.0, 1.0); m.emission_strength = float(rx.uniform(0.5, 4))
            u = rx.random()
            if u < 0.05: m.ior = 1.0
            elif u < 0.10: m.base_color = (0.0, 0.0, 0.0, m.base_color[3])
            elif u < 0.15: m.base_color = (1.0, 1.0, 1.0, m.base_color[3])
        return m

    sc.add_instance(meshes[0], Transform(scale=(6, 1, 6)), [material()])   # a floor so that most paths bounce
    for _ in range(int(rng.integers(2, 10))):
        s = rng.uniform(0.3, 1.8, 3) * np.where(rng.random(3) < 0.15, -1.0, 1.0)
        sc.add_instance(meshes[int(rng.integers(0, 3))],
                        Transform(translation=tuple(rng.uniform((-3, 0.2, -3), (3, 3, 3))), rotation=tuple(rng.uniform(-3.2, 3.2, 3)), scale=tuple(s)),
                        [material()])
    if rng.random() < 0.8:  # a light panel (otherwise: emissive objects or the environment, or darkness)
        sc.add_instance(meshes[0], Transform(translation=(float(rng.uniform(-1, 1)), 4.5, float(rng.uniform(-1, 1))), rotation=(np.pi, 0, 0)),
                        [Material(base_color=(0, 0, 0, 1), emission=(1, 1, 1), emission_strength=float(rng.uniform(2, 12)))])
    if rng.random() < 0.5:
        sc.env_texture = sc.add_texture(sky_environment(16, 8, sun=(int(rng.integers(0, 14)), int(rng.integers(0, 4)))), abi.TEX_RGBA32F)
    cam = Camera.with_focal_length(float(rng.uniform(18, 50)))
    if rng.random() < 0.3:
        cam.aperture = 2.0; cam.focus_distance = 6.0; cam.roundness = float(rng.uniform(0, 1)); cam.bokeh_power = float(rng.uniform(-1, 1))
    sc.set_camera(cam, Transform(translation=tuple(rng.uniform((-4, 1, 4), (4, 4, 7))), target=(0, 1, 0), track=True))
    return sc


def atrium_scene(env_size=(4096, 2048), columns=20):
    """C5 stand-in (BASELINE.json configs[4]: "glTF Sponza-class scene with mixed Lambert/GGX/emissive + EXR envmap"): no
    Sponza asset exists offline, so this is a procedural colonnade of the same class — ~260 k triangles, textured
    floor/walls with normal + roughness-metallic maps, GGX metal and glass objects, alpha-tested banners, emissive lamps
    and a 4096x2048 RGBA32F environment with a sun (8.4 M alias entries, ~100 MB like the 4K EXR the survey names)."""
    sc = Scene(name="atrium")
    yy, xx = np.mgrid[0:64, 0:64]
    tiles = np.zeros((64, 64, 4), dtype=np.uint8)
    grout = ((xx % 16) < 1) | ((yy % 16) < 1)
    tiles[..., 0] = np.where(grout, 60, 170 + (_hash_bytes((64, 64), 11) >> 3))
    tiles[..., 1] = np.where(grout, 55, 150 + (_hash_bytes((64, 64), 12) >> 3))
    tiles[..., 2] = np.where(grout, 50, 120 + (_hash_bytes((64, 64), 13) >> 3))
    tiles[..., 3] = 255
    t_tiles = sc.add_texture(tiles, abi.TEX_RGBA8_SRGB)
    rm = np.zeros((64, 64, 2), dtype=np.uint8)
    rm[..., 0] = np.where(grout, 230, 60 + (_hash_bytes((64, 64), 14) >> 2))
    rm[..., 1] = 0
    t_rm = sc.add_texture(rm, abi.TEX_RG8)
    nrm = np.zeros((64, 64, 4), dtype=np.uint8)
    nrm[..., 0] = np.where((xx % 16) < 1, 90, np.where((xx % 16) == 1, 166, 128))
    nrm[..., 1] = np.where((yy % 16) < 1, 90, np.where((yy % 16) == 1, 166, 128))
    nrm[..., 2] = 240; nrm[..., 3] = 255
    t_nrm = sc.add_texture(nrm, abi.TEX_RGBA8)
    banner = np.zeros((64, 32, 4), dtype=np.uint8)
    by, bx = np.mgrid[0:64, 0:32]
    banner[..., 0] = 150 + 60 * ((by // 8) % 2); banner[..., 1] = 30; banner[..., 2] = 40
    banner[..., 3] = np.where((by > 52) & (((bx // 4) % 2) == 0), 0, 255)   # a fringed lower edge: alpha cut-outs
    t_banner = sc.add_texture(banner, abi.TEX_RGBA8_SRGB)
    # --- geometry ---
    quad = sc.add_mesh(plane(1.0))
    box = sc.add_mesh(cube(1.0))
    col = sc.add_mesh(sphere(0.5, 48, 64))      # 6144 triangles, stretched into a column
    orb = sc.add_mesh(sphere(0.5, 24, 32))
    L, Wd, Hh = 40.0, 14.0, 10.0
    stone = Material(name="stone", base_texture=t_tiles, rm_texture=t_rm, normal_texture=t_nrm, roughness=1.0, metallic=1.0)
    plaster = Material(name="plaster", base_color=(0.75, 0.7, 0.62, 1.0), roughness=0.9)
    sc.add_instance(quad, Transform(scale=(Wd, 1, L)), [stone])                                                   # floor
    sc.add_instance(box, Transform(translation=(-Wd / 2 - 0.25, Hh / 2, 0), scale=(0.5, Hh, L)), [plaster])        # side walls
    sc.add_instance(box, Transform(translation=(Wd / 2 + 0.25, Hh / 2, 0), scale=(0.5, Hh, L)), [plaster])
    sc.add_instance(box, Transform(translation=(0, Hh / 2, -L / 2 - 0.25), scale=(Wd + 1, Hh, 0.5)), [plaster])    # back wall
    for side in (-1, 1):                                                                                           # roof beams: the sky shows between them
        for k in range(8):
            sc.add_instance(box, Transform(translation=(side * Wd / 4, Hh + 0.2, -L / 2 + (k + 0.5) * L / 8), scale=(Wd / 2 - 1.5, 0.4, 1.2)), [plaster])
    for i in range(columns):
        for side in (-1, 1):
            h = _pcg4d((i, side + 2, 5, 0))
            u = [(c >> 8) / float(1 << 24) for c in h]
            z = -L / 2 + (i + 0.5) * L / columns
            kind = h[3] % 4
            if kind == 0:
                m = Material(name="marble", base_color=(0.85, 0.83, 0.8, 1.0), roughness=0.35, clearcoat=0.6)
            elif kind == 1:
                m = Material(name="bronze", base_color=(0.7, 0.45, 0.2, 1.0), roughness=0.25 + 0.3 * u[0], metallic=1.0, anisotropy=0.4)
            elif kind == 2:
                m = Material(name="painted", base_texture=t_tiles, roughness=0.6)
            else:
                m = Material(name="stone column", base_color=(0.6, 0.6, 0.58, 1.0), roughness=0.8)
            sc.add_instance(col, Transform(translation=(side * (Wd / 2 - 2.0), Hh / 2 - 0.5, z), scale=(1.2, Hh - 1.0, 1.2)), [m])
        if i % 4 == 1:
            sc.add_instance(quad, Transform(translation=(0, Hh - 2.5, -L / 2 + (i + 0.5) * L / columns), rotation=(np.pi / 2, 0, 0), scale=(3.0, 1, 4.0)),
                            [Material(name="banner", base_texture=t_banner, base_texture_has_alpha=True, roughness=0.8)])
        if i % 5 == 2:
            sc.add_instance(orb, Transform(translation=(0, 1.0, -L / 2 + (i + 0.5) * L / columns), scale=(2, 2, 2)),
                            [Material(name="glass orb", base_color=(1, 1, 1, 1), roughness=0.05 + 0.1 * (i % 3), ior=1.5, transmission=1.0)])
        if i % 5 == 4:
            sc.add_instance(orb, Transform(translation=(2.5 * (1 if i % 2 else -1), Hh - 1.5, -L / 2 + (i + 0.5) * L / columns), scale=(0.8, 0.8, 0.8)),
                            [Material(name="lamp", base_color=(0, 0, 0, 1), emission=(1.0, 0.75, 0.45), emission_strength=25.0)])
    env = sky_environment(env_size[0], env_size[1], sun=(int(env_size[0] * 0.30), int(env_size[1] * 0.18)), sun_radiance=400.0)
    sy, sx = int(env_size[1] * 0.18), int(env_size[0] * 0.30)
    r = max(2, env_size[0] // 256)
    env[sy:sy + r, sx:sx + r, :3] = 400.0       # a sun disc that scales with the map resolution
    sc.env_texture = sc.add_texture(env, abi.TEX_RGBA32F)
    sc.set_camera(Camera.with_focal_length(24.0), Transform(translation=(0.5, 2.2, L / 2 - 2.0), target=(0, 3.0, -L / 2), track=True))
    return sc


def pairing_edge_cases_scene():
    """One INDEXED mesh whose consecutive triangles exercise every branch of the leaf-slot pairing (host_scene.h pair_mesh_triangles, lbvh.hip
    k_flatten): a plain quad; a fan of three (the third stays single); a partner that repeats a vertex (zero area, still a pair); a degenerate
    first triangle with a proper partner; a partner that shares all three indices (not a pair); triangles that share only one vertex; a
    mirrored duplicate (same three indices in another order); two materials across one pair.  Two instances, one mirrored."""
    import numpy as _np
    P = _np.array([[-2, 0, -2], [-2, 0, 2], [2, 0, -2], [2, 0, 2],            # 0-3 floor quad
                   [-3, 0.5, 0], [-2.5, 1.5, 0.3], [-2, 0.6, 0.5], [-1.5, 1.4, 0.2], [-1, 0.5, 0],   # 4-8 fan
                   [0, 1, -1], [1, 1, -1], [0.5, 2, -1.2], [1.5, 2.1, -0.8],      # 9-12
                   [3, 0.2, 1], [3.5, 1.2, 1], [4, 0.3, 1.2], [2.5, 1.0, 0.5]], dtype=_np.float32)
    I = _np.array([0, 1, 2,  2, 1, 3,           # quad: pair
                   4, 5, 6,  4, 6, 7,  4, 7, 8,  # fan: pair + single
                   9, 10, 11,  9, 9, 12,         # partner repeats a vertex: shared = 2 (9, 9), pair with a zero-area B
                   10, 10, 11,  10, 11, 12,      # degenerate A, proper B
                   13, 14, 15,  15, 13, 14,      # the same triangle twice (shared = 3): not a pair; the tie-break decides
                   13, 15, 16,  0, 16, 12,       # shares one vertex with its predecessor: single, single
                   5, 6, 7,  7, 6, 5], dtype=_np.uint32)   # a mirrored duplicate (shared = 3)
    n = _np.tile(_np.array([[0, 1, 0]], _np.float32), (len(P), 1))
    tg = _np.tile(_np.array([[1, 0, 0, 1]], _np.float32), (len(P), 1))
    uv = (P[:, [0, 2]] * 0.25).astype(_np.float32)
    slots = _np.zeros(len(I) // 3, _np.uint32)
    slots[1] = 1                                   # the quad's halves carry different materials (different shading classes in one slot)
    sc = Scene(name="pairing")
    m = sc.add_mesh(_make_mesh(P, n, tg, uv, I, slots))
    mats = [Material(base_color=(0.7, 0.6, 0.5, 1.0), roughness=0.8), Material(base_color=(0.9, 0.9, 0.9, 1.0), roughness=0.2, metallic=1.0)]
    sc.add_instance(m, Transform(translation=(0, 0, 0)), mats)
    sc.add_instance(m, Transform(translation=(0.3, 2.5, -1.0), scale=(-0.8, 0.9, 1.1)), mats)
    sc.env_texture = sc.add_texture(sky_environment(16, 8), 4)
    sc.set_camera(Camera.with_focal_length(24.0), Transform(translation=(0.5, 3.5, 7), target=(0.3, 1, 0), track=True))
    return sc


CONFIGS = {
    # name: (scene factory, width, height, spp, bounces)
    "c1": (lambda: cornell_scene("bench"), 512, 512, 64, 4),
    "c2": (cornell_sphere_scene, 1920, 1080, 256, 8),
    "c3": (field_scene, 1920, 1080, 256, 8),
    # context workload, not a BASELINE.json config: the same field with 128 x 128 instances = 16.6 M triangles — a structure
    # (~0.8 GB) that exceeds the 256 MB Infinity Cache, i.e. the regime in which HBM really is what the traversal kernels fetch from.  (The
    # spheres keep their size while their spacing drops to 0.5: they interpenetrate, most triangles lie inside a neighbour.)
    "c3xl": (lambda: field_scene(128), 1920, 1080, 256, 8),
    "c5": (atrium_scene, 3840, 2160, 512, 12),
}
