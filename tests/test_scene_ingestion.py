"""CPU: SURVEY §8f row N4 — scene ingestion (include/ptamd_scene.h; platinum_amd/csrc/scene_io.cpp, scene_gltf.cpp).

  * the reference's scene.json + _data.bin (core/scene.cpp:30-84, 536-903): files are written here by an independent
    Python restatement of Scene::saveToFile (tests/scene_formats.py) and by the committed fixture
    tests/golden/scene_fixture/mini.{json,_data.bin}; the C++ reader's snapshot is compared field by field with what
    rebuildResourceBuffers / getInstances would hand the renderer (index assignment in asset order, LIFO traversal, pruned
    invisible subtrees, default materials, per-instance MaterialGPU flags, camera world transform, environment)
  * MikkTSpace tangents against the reference's own deps/mikkt/mikktspace.c (oracle/_ref/libmikkt.so), bit-exact, and
    against the committed fixture tests/golden/mikkt_tangents.npz minted from it
  * the glTF importer (loaders/gltf.cpp): accessors, interleaving, normalized integers, .gltf/.glb/data URIs, TRS and matrix
    nodes (fastgltf decomposition + the reference's eulerFromQuat), cameras, KHR material extensions, PNG textures
    converted per TextureType, tangent generation when the file has none
The loaded scenes render through the oracle here and through the HIP path in test_gpu_parity.py."""
import ctypes as C
import json
import os
import shutil

import numpy as np
import pytest

import oracle_lib
import scene_formats as sf
from platinum_amd import abi, scene_io, scenes
from platinum_amd.renderer import make_params

G = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


# ---- helpers: read a pt_scene_snapshot back into numpy --------------------------------------------------------------
def snap_meshes(s):
    arr = C.cast(s.meshes, C.POINTER(abi.Mesh))
    out = []
    for i in range(s.mesh_count):
        m = arr[i]
        pos = np.ctypeslib.as_array(C.cast(m.positions, C.POINTER(C.c_float)), (m.vertex_count, 4)).copy()
        vd = np.ctypeslib.as_array(C.cast(m.vertex_data, C.POINTER(C.c_float)), (m.vertex_count, 12)).copy()
        idx = np.ctypeslib.as_array(C.cast(m.indices, C.POINTER(C.c_uint32)), (3 * m.triangle_count,)).copy()
        sl = np.ctypeslib.as_array(C.cast(m.material_slots, C.POINTER(C.c_uint32)), (m.triangle_count,)).copy()
        out.append((pos, vd, idx, sl))
    return out


def snap_instances(s):
    inst = C.cast(s.instances, C.POINTER(abi.Instance))
    mats = C.cast(s.instance_materials, C.POINTER(abi.InstanceMaterials))
    out = []
    for i in range(s.instance_count):
        tr = np.array([[inst[i].transform[c][r] for r in range(3)] for c in range(4)], dtype=f32)
        ml = C.cast(mats[i].materials, C.POINTER(abi.MaterialGPU))
        out.append((inst[i].accelerationStructureIndex, tr, [ml[k] for k in range(mats[i].material_count)], inst[i].mask))
    return out


def snap_textures(s):
    tex = C.cast(s.textures, C.POINTER(abi.Texture))
    bpp = {abi.TEX_RGBA8_SRGB: 4, abi.TEX_RGBA8: 4, abi.TEX_RG8: 2, abi.TEX_R8: 1, abi.TEX_RGBA32F: 16}
    out = []
    for i in range(s.texture_count):
        n = tex[i].width * tex[i].height * bpp[tex[i].format]
        out.append((tex[i].width, tex[i].height, tex[i].format, C.string_at(tex[i].pixels, n)))
    return out


def world(*chain):
    """Product of Transform matrices down a hierarchy, root first (core/scene.cpp:524)."""
    m = scenes.mat_identity()
    for t in chain:
        m = scenes.mat_mul(m, t.matrix())
    return m


# ---- the "mini" scene of the committed fixture ----------------------------------------------------------------------
def mini_scene_spec():
    rng = np.random.default_rng(17)
    base = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    base[..., 3] = np.where(np.arange(8)[None, :] < 4, 255, 90)
    rm = rng.integers(16, 256, (4, 4, 2), dtype=np.uint8)
    env = scenes.sky_environment(16, 8, sun=(5, 2), sun_radiance=40.0)
    plane, ball = scenes.plane(1.0), scenes.sphere(0.5, 6, 8)
    # alias table: what the reference stores is Environment's own table; take the library-independent oracle's
    tmp = scenes.Scene()
    tmp.add_instance(tmp.add_mesh(plane), scenes.Transform(), [scenes.Material()])
    tmp.env_texture = tmp.add_texture(env, abi.TEX_RGBA32F)
    alias = oracle_lib.OracleScene(tmp, make_params(8, 8, 1, 1)).envAlias()
    assets = [
        {"id": 0, "type": "texture", "name": "base", "alpha": True, "format": "srgb8", "pixels": base, "rc": 1},
        {"id": 2, "type": "material", "rc": 1, "data": sf.material_json("textured", (1, 1, 1, 1), roughness=0.7, metallic=0.2,
                                                                         textures=[("base", 0), ("rm", 1)])},
        {"id": 4, "type": "mesh", "mesh": plane, "rc": 3},
        {"id": 1, "type": "texture", "name": "rm", "alpha": False, "format": "rg8", "pixels": rm, "rc": 1},
        {"id": 3, "type": "material", "rc": 1, "data": sf.material_json("lamp", (0, 0, 0, 1), emission=(1, 0.8, 0.6), emission_strength=9.0)},
        {"id": 5, "type": "mesh", "mesh": ball, "rc": 3, "retain": False},
        {"id": 6, "type": "material", "rc": 1, "data": sf.material_json("glass", (1, 1, 1, 1), roughness=0.1, transmission=1.0, ior=1.45,
                                                                         aniso=0.3, thin=True)},
        {"id": 7, "type": "texture", "name": "sky", "alpha": False, "format": "rgba32f", "pixels": env, "rc": 1},
    ]
    T = sf.transform_json
    root = sf.node_json(0, "Scene", children=[
        sf.node_json(1, "floor", T(s=(8, 1, 8)), mesh=4, materials=[2]),
        sf.node_json(2, "group", T(t=(1, 0.5, 0), r=(0.1, 0.7, -0.2), s=(1.5, 1.5, 1.5)), children=[
            sf.node_json(3, "ball", T(t=(0, 1, 0)), mesh=5, materials=[6]),
            sf.node_json(4, "hidden", T(t=(0, 2, 0)), visible=False, mesh=5, materials=[6], children=[
                sf.node_json(5, "hidden child", T(), mesh=4, materials=[2])]),
            sf.node_json(6, "lamp", T(t=(0, 4, 0), r=(np.pi, 0, 0), s=(2, 1, 2)), mesh=4, materials=[3]),
        ]),
        sf.node_json(7, "cam rig", T(t=(0, 1, 0)), children=[
            sf.node_json(8, "Camera", T(t=(0, 2, 9), tgt=(0, 1, 0), track=True), camera={"f": 35.0, "aperture": 0.0, "sensor": (36.0, 24.0)})]),
        sf.node_json(9, "plain ball", T(t=(-2, 0.5, 1)), mesh=5, materials=[None]),
    ])
    return assets, root, {"texture": 7, "alias": alias}, dict(base=base, rm=rm, env=env, plane=plane, ball=ball, alias=alias)


@pytest.fixture(scope="module")
def mini(tmp_path_factory):
    d = tmp_path_factory.mktemp("mini")
    assets, root, envmap, data = mini_scene_spec()
    path = str(d / "mini.json")
    sf.write_reference_scene(path, assets, root, envmap)
    return path, data


def check_mini_snapshot(sc, data):
    Tr = scenes.Transform
    c = sc.counts()
    assert (c.nodes, c.meshes, c.textures, c.materials, c.cameras, c.instances) == (10, 2, 3, 3, 1, 4)
    assert c.triangles == 2 + 2 + 2 * 6 * 8 * 2
    assert sc.cameras() == [(8, "Camera")]
    s = sc.snapshot().struct
    # meshes and textures are indexed in asset (file) order per type (renderer_pt.cpp:485, 525)
    meshes = snap_meshes(s)
    for got, want in zip(meshes, (data["plane"], data["ball"])):
        assert got[0].tobytes() == np.asarray(want.positions, f32).tobytes() and got[1].tobytes() == np.asarray(want.vertex_data, f32).tobytes()
        assert np.array_equal(got[2], want.indices) and np.array_equal(got[3], want.material_slots)
    tex = snap_textures(s)
    assert [(t[0], t[1], t[2]) for t in tex] == [(8, 8, abi.TEX_RGBA8_SRGB), (4, 4, abi.TEX_RG8), (16, 8, abi.TEX_RGBA32F)]
    assert tex[0][3] == data["base"].tobytes() and tex[1][3] == data["rm"].tobytes() and tex[2][3] == data["env"].tobytes()
    assert s.env_texture == 2
    al = np.ctypeslib.as_array(C.cast(s.env_alias, C.POINTER(C.c_uint8)), (16 * 8 * 12,)).tobytes()
    assert al == data["alias"].tobytes()
    # instances: LIFO traversal (core/scene.cpp:514-534) => last child first; the invisible node prunes its subtree
    inst = snap_instances(s)
    group = Tr(translation=(1, 0.5, 0), rotation=(0.1, 0.7, -0.2), scale=(1.5, 1.5, 1.5))
    expect = [
        (1, world(Tr(), Tr(translation=(-2, 0.5, 1)))),
        (0, world(Tr(), group, Tr(translation=(0, 4, 0), rotation=(np.pi, 0, 0), scale=(2, 1, 2)))),
        (1, world(Tr(), group, Tr(translation=(0, 1, 0)))),
        (0, world(Tr(), Tr(scale=(8, 1, 8)))),
    ]
    assert [i[0] for i in inst] == [e[0] for e in expect]
    for (mi, tr, mats, mask), (_, w) in zip(inst, expect):
        np.testing.assert_allclose(tr, w[:, :3], rtol=2e-6, atol=2e-6)
        assert mask == 0xFF
    # materials: per-instance MaterialGPU arrays (renderer_pt.cpp:560-640)
    default = bytes(scenes.Material().to_gpu())
    got_default = inst[0][2][0]
    assert bytes(got_default)[:68] == default[:68] and got_default.flags == 0 and got_default.baseTextureId == -1
    lamp = inst[1][2][0]
    assert lamp.flags == abi.MATERIAL_EMISSIVE and lamp.emissionStrength == 9.0 and tuple(lamp.baseColor) == (0, 0, 0, 1)
    glass = inst[2][2][0]
    assert glass.flags == abi.MATERIAL_THIN_DIELECTRIC | abi.MATERIAL_ANISOTROPIC and glass.transmission == 1.0 and glass.ior == f32(1.45)
    floor = inst[3][2][0]
    assert (floor.baseTextureId, floor.rmTextureId, floor.normalTextureId, floor.emissionTextureId) == (0, 1, -1, -1)
    assert floor.flags == abi.MATERIAL_USE_ALPHA  # the base texture has alpha (renderer_pt.cpp:628-630)
    assert floor.roughness == f32(0.7) and floor.metallic == f32(0.2)
    # camera: world transform up the parent chain (core/scene.cpp:463-474) + Camera::withFocalLength defaults
    cw = np.array([[s.camera.world[c_][r] for r in range(4)] for c_ in range(4)], dtype=f32)
    np.testing.assert_allclose(cw, world(Tr(), Tr(translation=(0, 1, 0)), Tr(translation=(0, 2, 9), target=(0, 1, 0), track=True)), rtol=2e-6, atol=2e-6)
    assert (s.camera.focal_length, s.camera.aperture, s.camera.aperture_blades, s.camera.roundness, s.camera.focus_distance) == (35.0, 0.0, 7, 1.0, 1.0)
    assert tuple(s.camera.sensor_size) == (36.0, 24.0)
    return s


def test_reference_scene_json_reader(mini):
    path, data = mini
    sc = scene_io.SceneFile.load(path)
    check_mini_snapshot(sc, data)


def test_committed_scene_fixture_matches_the_writer(mini, tmp_path):
    """tests/golden/scene_fixture/mini.json + mini_data.bin (tools/make_golden.py): the reader on committed bytes."""
    path, data = mini
    gj, gb = os.path.join(G, "scene_fixture", "mini.json"), os.path.join(G, "scene_fixture", "mini_data.bin")
    assert open(gb, "rb").read() == open(path.replace(".json", "_data.bin"), "rb").read()
    assert json.load(open(gj)) == json.load(open(path))
    check_mini_snapshot(scene_io.SceneFile.load(gj), data)


def test_scene_json_save_round_trip(mini, tmp_path):
    """Scene::saveToFile restated in C++ (pt_scene_save_json): load -> save -> load reproduces the snapshot bytes, and the
    written json has the reference's structure (same keys, [offset, length] pairs that tile the .bin)."""
    path, data = mini
    a = scene_io.SceneFile.load(path)
    out = str(tmp_path / "again.json")
    a.save(out)
    b = scene_io.SceneFile.load(out)
    check_mini_snapshot(b, data)
    ja, jb = json.load(open(path)), json.load(open(out))
    strip_rc = lambda j: [{k: v for k, v in x.items() if k != "rc"} for x in j["assets"]["assets"]]
    assert strip_rc(ja) == strip_rc(jb) and ja["root"] == jb["root"] and ja["envmap"] == jb["envmap"]
    # the node pass retains assets again on load (core/scene.cpp:880 -> setMesh -> retainAsset): rc grows exactly like the reference's
    rc = {x["id"]: x["rc"] for x in jb["assets"]["assets"]}
    assert rc[4] == 3 + 3 and rc[5] == 3 + 3 and rc[2] == 1 + 2 and rc[0] == 1
    assert open(out.replace(".json", "_data.bin"), "rb").read() == open(path.replace(".json", "_data.bin"), "rb").read()


def test_loaded_scene_renders_through_the_oracle(mini):
    path, _ = mini
    sc = scene_io.SceneFile.load(path)
    p = make_params(64, 36, 1, 5)
    o = oracle_lib.OracleScene(sc, p)
    rad, hits = o.debug_sample(0)
    assert np.isfinite(rad).all() and rad[..., :3].mean() > 1e-3
    assert set(np.unique(hits[0, ..., 0])) >= {0, 2, 3}  # plain ball, glass ball and floor are in view (the lamp is above the frame)
    assert (hits[1:, ..., 0] == 1).any()                  # ... and the lamp is reached by secondary rays
    assert o.constants().envLightCount == 1 and len(o.lights()) == 2


def test_scene_reader_errors(tmp_path, mini):
    path, _ = mini
    with pytest.raises(abi.PtamdError, match="cannot open"):
        scene_io.SceneFile.load(str(tmp_path / "missing.json"))
    d = tmp_path / "trunc"
    d.mkdir()
    shutil.copy(path, d / "mini.json")
    blob = open(path.replace(".json", "_data.bin"), "rb").read()
    open(d / "mini_data.bin", "wb").write(blob[: len(blob) // 2])
    with pytest.raises(abi.PtamdError, match="shorter than the json says"):
        scene_io.SceneFile.load(str(d / "mini.json"))
    open(d / "bad.json", "w").write('{"assets": {"nextId": 0, "assets": []}, "root": {"id": 0, "name": "x"')
    open(d / "bad_data.bin", "wb").write(b"")
    with pytest.raises(abi.PtamdError, match="json:"):
        scene_io.SceneFile.load(str(d / "bad.json"))
    sc = scene_io.SceneFile.load(path)
    sc.camera = 3  # not a camera node
    with pytest.raises(abi.PtamdError, match="not a camera node"):
        sc.snapshot()


# ---- tangents ----------------------------------------------------------------------------------------------------------
def tangent_cases():
    rng = np.random.default_rng(3)
    n = 9
    xs, zs = np.meshgrid(np.arange(n), np.arange(n))
    P = np.stack([xs.ravel(), rng.normal(0, 0.2, n * n), zs.ravel()], -1).astype(f32)
    N = rng.normal(size=(n * n, 3))
    N[:, 1] += 3
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    UV = np.stack([np.abs(xs.ravel() / (n - 1) - 0.5) * 2, zs.ravel() / (n - 1)], -1).astype(f32)  # mirrored in u: two orientations
    idx = []
    for z in range(n - 1):
        for x in range(n - 1):
            a = z * n + x
            idx += [a, a + n, a + 1, a + 1, a + n, a + n + 1]
    idx += [0, 0, 5, 3, 4, 4]      # degenerate triangles (repeated vertex)
    UV[40] = UV[41]                # zero-area uv triangles: GROUP_WITH_ANY
    grid = scenes._make_mesh(P, N, np.zeros((n * n, 4)), UV, idx, np.zeros(len(idx) // 3))
    return {"plane": scenes.plane(2.0), "cube": scenes.cube(2.0), "sphere": scenes.sphere(1.0, 12, 16), "cornell": scenes.cornell_box(),
            "grid": grid}


@pytest.mark.parametrize("name", ["plane", "cube", "sphere", "cornell", "grid"])
def test_tangents_match_reference_mikktspace_and_fixture(name):
    m = tangent_cases()[name]
    vd = np.ascontiguousarray(m.vertex_data, dtype=f32).copy()
    vd[:, 4:8] = 0
    mine = scene_io.generate_tangents(m.positions, vd.copy(), m.indices)[:, 4:8]
    g = np.load(os.path.join(G, "mikkt_tangents.npz"))
    assert mine.tobytes() == g[name].tobytes()             # committed output of the reference's mikktspace.c
    if sf.mikkt_available():                                # and the library itself, where oracle/_ref was built
        ref = sf.mikkt_reference_tangents(m.positions, vd, m.indices)
        assert ref.tobytes() == mine.tobytes()
    assert np.allclose(np.linalg.norm(mine[:, :3], axis=1), 1.0, atol=1e-5) and set(np.unique(mine[:, 3])) <= {-1.0, 1.0}


# ---- PNG -----------------------------------------------------------------------------------------------------------------
def test_png_adam7_interlaced_matches_plain_and_the_references_stb_image(tmp_path):
    """Adam7 (stb_image decodes it, so loaders/texture.cpp does): every pass geometry incl. images narrower / shorter than the
    8x8 pattern (empty passes), 8- and 16-bit, palette, grey; checked against the array that was encoded, against the same picture
    written without interlacing, and — where oracle/_ref was built — against the reference's own stb_image v2.30."""
    rng = np.random.default_rng(8)
    have_stb = os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "stbi2raw"))
    pal = rng.integers(0, 256, (7, 3), dtype=np.uint8)
    for (h, w) in [(1, 1), (1, 9), (2, 3), (3, 5), (5, 2), (8, 8), (9, 9), (13, 7), (17, 33)]:
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        rgba16 = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
        grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
        pidx = rng.integers(0, 7, (h, w), dtype=np.uint8)
        cases = [(rgba, {}, rgba), (rgba16, {}, (rgba16 >> 8).astype(np.uint8)),
                 (grey, {}, np.stack([grey, grey, grey, np.full_like(grey, 255)], -1)),
                 (rgba[..., :3], {}, np.concatenate([rgba[..., :3], np.full((h, w, 1), 255, np.uint8)], -1)),
                 (pidx, {"palette": pal}, np.concatenate([pal[pidx], np.full((h, w, 1), 255, np.uint8)], -1))]
        for k, (img, kw, want) in enumerate(cases):
            png = sf.png_bytes(img, interlace=True, **kw)
            got = scene_io.decode_image_rgba8(png)
            assert got.shape == (h, w, 4) and np.array_equal(got, want), (h, w, k)
            assert np.array_equal(got, scene_io.decode_image_rgba8(sf.png_bytes(img, **kw))), (h, w, k)
            if have_stb:
                f = tmp_path / f"a7_{h}x{w}_{k}.png"
                f.write_bytes(png)
                assert np.array_equal(sf.stbi_reference_rgba(f), got), (h, w, k)
    bad = bytearray(sf.png_bytes(rng.integers(0, 256, (4, 4, 4), dtype=np.uint8)))
    bad[8 + 8 + 12] = 2                                    # IHDR interlace method 2 does not exist
    with pytest.raises(abi.PtamdError, match="png: "):
        scene_io.decode_image_rgba8(bytes(bad))


def test_png_decoder_all_colour_types(tmp_path):
    """Through the importer: a glTF whose base-colour texture is the PNG under test (sRGB type = RGBA8 bytes verbatim)."""
    rng = np.random.default_rng(2)
    rgba = rng.integers(0, 256, (13, 7, 4), dtype=np.uint8)
    rgb, grey, ga = rgba[..., :3], rgba[..., 0], rgba[..., [0, 3]]
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    pidx = rng.integers(0, 16, (13, 7), dtype=np.uint8)
    trns = bytes(rng.integers(0, 256, 10, dtype=np.uint8))
    rgba16 = rng.integers(0, 65536, (5, 9, 4), dtype=np.uint16)
    cases = {
        "rgba": (sf.png_bytes(rgba), rgba),
        "rgb": (sf.png_bytes(rgb), np.concatenate([rgb, np.full((13, 7, 1), 255, np.uint8)], -1)),
        "grey": (sf.png_bytes(grey), np.stack([grey, grey, grey, np.full_like(grey, 255)], -1)),
        "grey_alpha": (sf.png_bytes(ga), np.stack([ga[..., 0]] * 3 + [ga[..., 1]], -1)),
        "palette_trns": (sf.png_bytes(pidx, palette=pal, trns=trns),
                         np.concatenate([pal[pidx], np.array([trns[i] if i < 10 else 255 for i in pidx.ravel()], np.uint8).reshape(13, 7, 1)], -1)),
        "rgba16": (sf.png_bytes(rgba16), (rgba16 >> 8).astype(np.uint8)),
        "filter_paeth_only": (sf.png_bytes(rgba, filter_type=4), rgba),
    }
    for name, (png, want) in cases.items():
        b = sf.GltfBuilder()
        t = b.image_png(png)
        b.doc["materials"].append({"pbrMetallicRoughness": {"baseColorTexture": {"index": t}}})
        pos = b.accessor(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], f32), "VEC3")
        b.doc["meshes"].append({"primitives": [{"attributes": {"POSITION": pos}, "material": 0}]})
        b.doc["nodes"].append({"mesh": 0})
        b.doc["scenes"].append({"nodes": [0]})
        path = str(tmp_path / f"{name}.glb")
        b.write(path, glb=True)
        sc = scene_io.SceneFile.empty().import_gltf(path)
        sc.add_camera((0, 0, 3), (0, 0, 0))
        tex = snap_textures(sc.snapshot().struct)
        assert (tex[0][0], tex[0][1], tex[0][2]) == (want.shape[1], want.shape[0], abi.TEX_RGBA8_SRGB), name
        assert tex[0][3] == want.tobytes(), name
    b = sf.GltfBuilder()
    b.image_png(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    b.doc["materials"].append({"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}})
    b.write(str(tmp_path / "jpeg.glb"), glb=True)
    with pytest.raises(abi.PtamdError, match="jpeg: "):  # a JPEG signature followed by garbage
        scene_io.SceneFile.empty().import_gltf(str(tmp_path / "jpeg.glb"))


# ---- glTF ----------------------------------------------------------------------------------------------------------------
def quat_from_euler_yxz(rx, ry, rz):
    """A quaternion for R = Ry * Rx * Rz (the order Transform::matrix composes, core/transform.hpp:47-50)."""
    def q(axis, a):
        v = np.zeros(4)
        v[axis] = np.sin(a / 2)
        v[3] = np.cos(a / 2)
        return v

    def mul(a, b):
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
    return mul(mul(q(1, ry), q(0, rx)), q(2, rz))


def build_gltf(tmp_path, kind):
    ball = scenes.sphere(0.5, 6, 8)
    quad = scenes.plane(2.0)
    rng = np.random.default_rng(23)
    base = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)
    base[..., 3] = 255
    orm = rng.integers(0, 256, (4, 4, 3), dtype=np.uint8)
    b = sf.GltfBuilder()
    t_base = b.image_png(sf.png_bytes(base), embed="view" if kind == "glb" else ("data" if kind == "embedded" else "file"),
                         dirpath=str(tmp_path), name="base colour.png")
    t_orm = b.image_png(sf.png_bytes(orm), embed="view" if kind == "glb" else "data", name="orm")
    b.doc["materials"] += [
        {"name": "painted", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.8, 0.7, 1.0], "roughnessFactor": 0.6, "metallicFactor": 0.3,
                                                       "baseColorTexture": {"index": t_base}, "metallicRoughnessTexture": {"index": t_orm}},
         "normalTexture": {"index": t_base}, "emissiveFactor": [0.0, 0.0, 0.0],
         "extensions": {"KHR_materials_clearcoat": {"clearcoatFactor": 0.5, "clearcoatRoughnessFactor": 0.1, "clearcoatTexture": {"index": t_orm}},
                        "KHR_materials_ior": {"ior": 1.33}}},
        {"name": "glow", "emissiveFactor": [1.0, 0.5, 0.25], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0},
                                                                              "KHR_materials_transmission": {"transmissionFactor": 0.25},
                                                                              "KHR_materials_anisotropy": {"anisotropyStrength": 0.4, "anisotropyRotation": 1.1}}},
    ]
    # mesh 0: the sphere, interleaved POSITION+NORMAL (stride 24), uv as normalized ushort, ushort indices, NO tangents
    inter = np.concatenate([ball.positions[:, :3], ball.vertex_data[:, 0:3]], 1).astype(f32)
    v = b.view(inter.tobytes(), stride=24)
    a_pos = b.accessor(inter, "VEC3", view=v, offset=0)
    a_nrm = b.accessor(inter, "VEC3", view=v, offset=12)
    uv16 = np.round(ball.vertex_data[:, 8:10] * 65535.0).astype(np.uint16)
    a_uv = b.accessor(uv16, "VEC2", component=5123, normalized=True)
    a_idx = b.accessor(ball.indices.astype(np.uint16), "SCALAR", component=5123)
    # mesh 1: two primitives (two material slots): the quad with float data + tangents, and a single unindexed triangle without material
    a_qpos = b.accessor(quad.positions[:, :3].astype(f32), "VEC3")
    a_qnrm = b.accessor(quad.vertex_data[:, 0:3].astype(f32), "VEC3")
    a_quv = b.accessor(quad.vertex_data[:, 8:10].astype(f32), "VEC2")
    a_qtan = b.accessor(np.tile(np.array([[1, 0, 0, 1]], f32), (4, 1)), "VEC4")
    a_qidx = b.accessor(quad.indices.astype(np.uint32), "SCALAR", component=5125)
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 0, -1]], f32)
    a_tpos = b.accessor(tri, "VEC3")
    b.doc["meshes"] += [
        {"name": "ball", "primitives": [{"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv}, "indices": a_idx, "material": 0}]},
        {"name": "quad+tri", "primitives": [{"attributes": {"POSITION": a_qpos, "NORMAL": a_qnrm, "TEXCOORD_0": a_quv, "TANGENT": a_qtan},
                                             "indices": a_qidx, "material": 1},
                                            {"attributes": {"POSITION": a_tpos}, "mode": 4},
                                            {"attributes": {"POSITION": a_tpos}, "mode": 1}]},  # lines: skipped with a warning
    ]
    b.doc["cameras"] += [{"type": "perspective", "perspective": {"yfov": 0.6, "aspectRatio": 1.6, "znear": 0.1}},
                         {"type": "orthographic", "orthographic": {"xmag": 1, "ymag": 1, "znear": 0.1, "zfar": 10}}]
    e = (0.3, -0.8, 0.2)
    q = quat_from_euler_yxz(*e)
    M = scenes.mat_mul(scenes.mat_mul(scenes.mat_translation((1, 2, 3)), scenes.mat_rotation_y(0.5)), scenes.mat_scaling((2, 2, 2)))
    b.doc["nodes"] += [
        {"name": "trs ball", "mesh": 0, "translation": [0.5, 1.0, -0.5], "rotation": [float(x) for x in q], "scale": [1.0, 2.0, 1.0], "children": [1, 2]},
        {"name": "matrix child", "mesh": 1, "matrix": [float(x) for x in M.reshape(-1)]},
        {"name": "empty leaf"},
        {"name": "eye", "camera": 0, "translation": [0.0, 1.0, 6.0]},
    ]
    b.doc["scenes"].append({"nodes": [0, 3]})
    path = str(tmp_path / ("scene.glb" if kind == "glb" else "scene.gltf"))
    b.write(path, glb=(kind == "glb"), embed_buffer=(kind == "embedded"))
    return path, dict(ball=ball, quad=quad, base=base, orm=orm, euler=e, uv16=uv16, tri=tri)


@pytest.mark.parametrize("kind", ["glb", "embedded", "external"])
def test_gltf_import(tmp_path, kind):
    path, d = build_gltf(tmp_path, kind)
    sc = scene_io.SceneFile.empty().import_gltf(path, scene_io.GLTF_SKIP_EMPTY_NODES)
    c = sc.counts()
    assert (c.meshes, c.textures, c.materials, c.cameras, c.instances) == (2, 2, 2, 1, 2)
    assert c.nodes == 1 + 3  # root + three nodes; the empty leaf is skipped (LoadOptions_SkipEmptyNodes, gltf.cpp:259-263)
    assert sc.cameras() == [(3, "eye")]
    s = sc.snapshot().struct
    meshes = snap_meshes(s)
    ball, quad = d["ball"], d["quad"]
    # mesh 0: positions / normals verbatim, uv = ushort / 65535, tangents generated by MikkTSpace (no TANGENT attribute)
    pos, vd, idx, sl = meshes[0]
    assert np.array_equal(pos[:, :3], ball.positions[:, :3]) and np.array_equal(vd[:, 0:3], ball.vertex_data[:, 0:3])
    assert np.array_equal(vd[:, 8:10], d["uv16"].astype(f32) / f32(65535.0))
    assert np.array_equal(idx, ball.indices) and np.all(sl == 0)
    want_vd = vd.copy()
    want_vd[:, 4:8] = 0
    assert np.array_equal(vd[:, 4:8], scene_io.generate_tangents(pos, want_vd, idx)[:, 4:8])
    assert np.allclose(np.linalg.norm(vd[:, 4:7], axis=1), 1.0, atol=1e-5)
    # mesh 1: primitives concatenated with index offsets and slot numbers (gltf.cpp:217-230); TANGENT kept; lines skipped
    pos, vd, idx, sl = meshes[1]
    assert len(pos) == 4 + 3 and np.array_equal(pos[:4, :3], quad.positions[:, :3]) and np.array_equal(pos[4:, :3], d["tri"])
    assert np.array_equal(idx, np.concatenate([quad.indices, [4, 5, 6]])) and list(sl) == [0, 0, 1]
    assert np.array_equal(vd[:4, 4:8], np.tile(np.array([[1, 0, 0, 1]], f32), (4, 1))) and np.all(vd[4:, 0:3] == 0)
    # textures in first-use order with their TextureType conversion (texture.cpp:30-48): base sRGB RGBA8, ORM -> RG8 = (G, B)
    # NB the same glTF texture used for base colour AND normal keeps the LAST registered type (gltf.cpp:343-366): LinearRGB
    tex = snap_textures(s)
    assert (tex[0][0], tex[0][1], tex[0][2]) == (5, 6, abi.TEX_RGBA8) and tex[0][3] == d["base"].tobytes()
    assert (tex[1][0], tex[1][1], tex[1][2]) == (4, 4, abi.TEX_R8) and tex[1][3] == d["orm"][..., 0].tobytes()  # clearcoat (Mono) registered last
    inst = snap_instances(s)
    # traversal is LIFO: scene nodes [trs ball, eye]; children of "trs ball" after it
    assert [i[0] for i in inst] == [0, 1]
    painted = inst[0][2][0]
    assert np.allclose(tuple(painted.baseColor), (0.9, 0.8, 0.7, 1.0)) and painted.roughness == f32(0.6) and painted.metallic == f32(0.3)
    assert (painted.baseTextureId, painted.normalTextureId, painted.rmTextureId, painted.clearcoatTextureId) == (0, 0, 1, 1)
    assert painted.clearcoat == 0.5 and painted.clearcoatRoughness == f32(0.1) and painted.ior == f32(1.33)
    assert painted.emissionStrength == 1.0 and painted.flags == 0  # fastgltf default strength 1, factor 0 => not emissive
    glow, none = inst[1][2]
    assert glow.flags == abi.MATERIAL_EMISSIVE | abi.MATERIAL_ANISOTROPIC and glow.emissionStrength == 4.0 and glow.transmission == 0.25
    assert glow.metallic == 1.0 and glow.roughness == 1.0 and glow.anisotropyRotation == f32(1.1)
    # a primitive without material gets asset id 0 (gltf.cpp:234-235) — here that IS the first material ("painted")
    assert bytes(none) == bytes(painted)
    # node transforms: quaternion -> eulerFromQuat (gltf.cpp:9-17) -> Transform::matrix = T*Ry*Rx*Rz*S.  The reference's
    # formula is the heading/attitude/bank extraction (R = Ry*Rz*Rx) with the asin argument clamped to +-0.5, which is NOT
    # the inverse of its own Ry*Rx*Rz composition for a general rotation: a drop-in reproduces that, it does not fix it.
    qx, qy, qz, qw = [f32(v) for v in quat_from_euler_yxz(*d["euler"])]
    two, one = f32(2), f32(1)
    e = (np.arctan2(two * (qw * qx - qy * qz), one - two * (qx * qx + qz * qz)),
         np.arctan2(two * (qw * qy - qx * qz), one - two * (qy * qy + qz * qz)),
         np.arcsin(two * np.clip(qx * qy + qw * qz, f32(-0.5), f32(0.5))))
    Tr = scenes.Transform
    w0 = Tr(translation=(0.5, 1.0, -0.5), rotation=e, scale=(1, 2, 1)).matrix()
    np.testing.assert_allclose(inst[0][1], w0[:, :3], rtol=1e-5, atol=2e-6)
    M = scenes.mat_mul(scenes.mat_mul(scenes.mat_translation((1, 2, 3)), scenes.mat_rotation_y(0.5)), scenes.mat_scaling((2, 2, 2)))
    np.testing.assert_allclose(inst[1][1], scenes.mat_mul(w0, M)[:, :3], rtol=1e-5, atol=5e-6)  # matrix decomposed (fastgltf math.hpp:854-891)
    # camera: Camera::withFov(yfov, {24 * aspect, 24}) (gltf.cpp:83-91, core/camera.hpp:32-42); orthographic ones are dropped
    assert np.isclose(s.camera.sensor_size[0], 24 * 1.6) and s.camera.sensor_size[1] == 24.0
    assert np.isclose(s.camera.focal_length, 24.0 / (2 * np.tan(0.3)), rtol=1e-6)
    # renders
    o = oracle_lib.OracleScene(sc, make_params(48, 30, 1, 4))
    rad, hits = o.debug_sample(0)
    assert np.isfinite(rad).all() and (hits[0, ..., 0] >= 0).any()


def test_gltf_create_scene_nodes_option_and_errors(tmp_path):
    path, _ = build_gltf(tmp_path, "glb")
    sc = scene_io.SceneFile.empty().import_gltf(path, scene_io.GLTF_CREATE_SCENE_NODES)
    assert sc.counts().nodes == 1 + 1 + 4  # root, the "scene" node named after the file, all four nodes (empty leaf kept)
    with pytest.raises(abi.PtamdError, match="cannot open"):
        scene_io.SceneFile.empty().import_gltf(str(tmp_path / "nope.gltf"))
    bad = tmp_path / "bad.gltf"
    bad.write_text(json.dumps({"asset": {"version": "2.0"}, "buffers": [{"byteLength": 4, "uri": "data:application/octet-stream;base64,AAAAAA=="}],
                               "bufferViews": [{"buffer": 0, "byteLength": 4}],
                               "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
                               "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}]}))
    with pytest.raises(abi.PtamdError, match="accessor exceeds its bufferView"):
        scene_io.SceneFile.empty().import_gltf(str(bad))


def test_gltf_sparse_accessors(tmp_path):
    """accessor.sparse (fastgltf::iterateAccessor substitutes the listed elements, tools.hpp:529-548): positions with two displaced
    vertices, u8 / u16 index types, a texcoord accessor without a bufferView (zeros + sparse), sparse triangle indices."""
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [2, 0, 0], [2, 1, 0]], f32)
    b = sf.GltfBuilder()
    pos = b.accessor(base, "VEC3")
    sp_idx = b.view(np.array([1, 4], np.uint8).tobytes())
    sp_val = b.view(np.array([[1, 0, 5], [2, 0, 7]], f32).tobytes())
    b.doc["accessors"][pos]["sparse"] = {"count": 2, "indices": {"bufferView": sp_idx, "componentType": 5121},
                                          "values": {"bufferView": sp_val}}
    uv_idx = b.view(b"\0\0" + np.array([2, 5], np.uint16).tobytes())        # (byteOffset 2 into the view)
    uv_val = b.view(np.array([[0.25, 0.5], [0.75, 1.0]], f32).tobytes())
    b.doc["accessors"].append({"componentType": 5126, "count": 6, "type": "VEC2",
                               "sparse": {"count": 2, "indices": {"bufferView": uv_idx, "byteOffset": 2, "componentType": 5123},
                                          "values": {"bufferView": uv_val}}})
    uv = len(b.doc["accessors"]) - 1
    ind = b.accessor(np.array([0, 1, 2, 1, 3, 2, 0, 0, 0], np.uint16), "SCALAR", component=5123)
    ii = b.view(np.array([6, 7, 8], np.uint32).tobytes())
    iv = b.view(np.array([1, 4, 5], np.uint16).tobytes())
    b.doc["accessors"][ind]["sparse"] = {"count": 3, "indices": {"bufferView": ii, "componentType": 5125}, "values": {"bufferView": iv}}
    b.doc["meshes"].append({"primitives": [{"attributes": {"POSITION": pos, "TEXCOORD_0": uv}, "indices": ind}]})
    b.doc["nodes"].append({"mesh": 0})
    b.doc["scenes"].append({"nodes": [0]})
    path = str(tmp_path / "sparse.glb")
    b.write(path, glb=True)
    sc = scene_io.SceneFile.empty().import_gltf(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    (p, vd, idx, _), = snap_meshes(sc.snapshot().struct)
    want = base.copy(); want[1] = (1, 0, 5); want[4] = (2, 0, 7)
    assert np.array_equal(p[:, :3], want)
    wuv = np.zeros((6, 2), f32); wuv[2] = (0.25, 0.5); wuv[5] = (0.75, 1.0)
    assert np.array_equal(vd[:, 8:10], wuv)
    assert idx.tolist() == [0, 1, 2, 1, 3, 2, 1, 4, 5]
    # indices out of order / beyond the accessor are rejected
    b.doc["accessors"][pos]["sparse"]["indices"]["bufferView"] = b.view(np.array([4, 1], np.uint8).tobytes())
    b.write(path, glb=True)
    with pytest.raises(abi.PtamdError, match="sparse indices must be strictly increasing"):
        scene_io.SceneFile.empty().import_gltf(path)
    b.doc["accessors"][pos]["sparse"]["indices"]["bufferView"] = b.view(np.array([1, 6], np.uint8).tobytes())
    b.write(path, glb=True)
    with pytest.raises(abi.PtamdError, match="sparse indices must be strictly increasing and below"):
        scene_io.SceneFile.empty().import_gltf(path)


# ---- environment files: OpenEXR and Radiance HDR ------------------------------------------------------------------------
def _env_pixels(sc):
    s = sc.snapshot().struct
    tex = snap_textures(s)[s.env_texture]
    return np.frombuffer(tex[3], dtype=f32).reshape(tex[1], tex[0], 4)


@pytest.mark.parametrize("compression,ptype", [("none", "float"), ("zip", "float"), ("zips", "half"), ("zip", "half"), ("rle", "half"), ("rle", "float")])
def test_exr_environment_reader(tmp_path, compression, ptype):
    rng = np.random.default_rng(31)
    h, w = 37, 53  # not a multiple of the 16-line ZIP block
    dt = np.float16 if ptype == "half" else f32
    ch = {n: (rng.random((h, w)) * (50.0 if n != "A" else 1.0)).astype(dt).astype(f32) for n in "RGBA"}
    ch["R"][5:9, 7:30] = 3.25  # flat runs for the RLE path
    want = np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1)
    path = str(tmp_path / "env.exr")
    sf.write_exr(path, ch, compression, ptype)
    sc = scene_io.SceneFile.empty()
    sc.load_environment(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    got = _env_pixels(sc)
    assert got.shape == (h, w, 4) and got.tobytes() == want.tobytes()
    if os.path.exists(sf.EXR2RAW_PATH):  # the reference's tinyexr reads the same file to the same bits
        assert sf.tinyexr_reference_rgba(path, str(tmp_path)).tobytes() == got.tobytes()


@pytest.mark.parametrize("compression,ptype,mipmap", [("zip", "float", False), ("none", "half", False), ("zip", "half", True), ("none", "float", True)])
def test_exr_tiled_reader(tmp_path, compression, ptype, mipmap):
    """Tiled files (LoadEXR assembles the tiles of the full-size level, tinyexr.h:6374-6440): edge tiles narrower / lower than the tile
    size, a multi-resolution file whose further levels must be ignored; the same bits as the reference's tinyexr where it was built."""
    rng = np.random.default_rng(41)
    h, w = 45, 70
    dt = np.float16 if ptype == "half" else f32
    ch = {n: (rng.random((h, w)) * 30.0).astype(dt).astype(f32) for n in "RGB"}
    want = np.stack([ch["R"], ch["G"], ch["B"], np.ones((h, w), f32)], -1)
    path = str(tmp_path / "tiled.exr")
    sf.write_exr_tiled(path, ch, (32, 16), compression, ptype, mipmap)
    sc = scene_io.SceneFile.empty()
    sc.load_environment(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    got = _env_pixels(sc)
    assert got.shape == (h, w, 4) and got.tobytes() == want.tobytes()
    if os.path.exists(sf.EXR2RAW_PATH):
        assert sf.tinyexr_reference_rgba(path, str(tmp_path)).tobytes() == got.tobytes()


def test_exr_tiled_piz_fixture_written_and_read_by_the_references_tinyexr():
    """tests/golden/exr_piz_tiled_fixture.exr: a 70x45 RGB half file in 32x16 PIZ tiles ENCODED by the reference's tinyexr
    (oracle/_ref/exrwrite … 32 16); the .npz holds what its LoadEXR reads back (tools/make_golden.py)."""
    g = np.load(os.path.join(G, "exr_piz_tiled_fixture.npz"))
    sc = scene_io.SceneFile.empty()
    sc.load_environment(os.path.join(G, "exr_piz_tiled_fixture.exr"))
    sc.add_camera((0, 0, 3), (0, 0, 0))
    got = _env_pixels(sc)
    assert got.shape == g["rgba"].shape and got.tobytes() == g["rgba"].tobytes()


def test_exr_tiled_errors(tmp_path):
    rng = np.random.default_rng(42)
    ch = {n: rng.random((20, 24)).astype(f32) for n in "RGB"}
    path = str(tmp_path / "t.exr")
    sf.write_exr_tiled(path, ch, (16, 16), "none", "float")
    blob = bytearray(open(path, "rb").read())
    first = int.from_bytes(blob[blob.index(b"tiledesc") + 9 + 4 + 9 + 1:][:8], "little")   # offset-table entry 0 (after the header's final NUL)
    bad = bytearray(blob); bad[first + 8:first + 12] = (1).to_bytes(4, "little")              # level_x = 1
    open(path, "wb").write(bad)
    sc = scene_io.SceneFile.empty()
    with pytest.raises(abi.PtamdError, match="does not start with the full-resolution tiles"):
        sc.load_environment(path)
    bad = bytearray(blob); bad[first:first + 4] = (9).to_bytes(4, "little")                   # tile_x beyond the window
    open(path, "wb").write(bad)
    with pytest.raises(abi.PtamdError, match="tile outside the data window"):
        sc.load_environment(path)
    bad = bytearray(blob); i = bad.index(b"tiledesc") + 9 + 4
    bad[i:i + 4] = (0).to_bytes(4, "little")                                                 # tile width 0
    open(path, "wb").write(bad)
    with pytest.raises(abi.PtamdError, match="valid tile description"):
        sc.load_environment(path)


def test_exr_channel_assembly_window_and_line_order(tmp_path):
    """LoadEXR semantics: RGB without A gets alpha 1, one channel alone fills all four, other channels are ignored; data windows
    with an origin and decreasing-y files are placed correctly."""
    rng = np.random.default_rng(32)
    h, w = 20, 9
    r, g, b, z = [rng.random((h, w)).astype(f32) for _ in range(4)]
    cases = {
        "rgb": ({"R": r, "G": g, "B": b}, np.stack([r, g, b, np.ones_like(r)], -1), dict()),
        "rgbz": ({"R": r, "G": g, "B": b, "Z": z}, np.stack([r, g, b, np.ones_like(r)], -1), dict(compression="zip")),
        "single": ({"Y": g}, np.stack([g, g, g, g], -1), dict(compression="zips")),
        "window": ({"R": r, "G": g, "B": b}, np.stack([r, g, b, np.ones_like(r)], -1), dict(data_window_origin=(-7, 12), compression="zip")),
        "decreasing_y": ({"R": r, "G": g, "B": b}, np.stack([r, g, b, np.ones_like(r)], -1), dict(line_order=1, compression="zip")),
    }
    for name, (ch, want, kw) in cases.items():
        path = str(tmp_path / f"{name}.exr")
        sf.write_exr(path, ch, **kw)
        sc = scene_io.SceneFile.empty()
        sc.load_environment(path)
        sc.add_camera((0, 0, 3), (0, 0, 0))
        assert _env_pixels(sc).tobytes() == want.tobytes(), name
        if os.path.exists(sf.EXR2RAW_PATH):
            assert sf.tinyexr_reference_rgba(path, str(tmp_path)).tobytes() == want.tobytes(), name
    open(tmp_path / "bad.exr", "wb").write(b"not an exr at all")
    with pytest.raises(abi.PtamdError, match="exr: bad magic"):
        scene_io.SceneFile.empty().load_environment(str(tmp_path / "bad.exr"))


@pytest.mark.parametrize("rle", [True, False])
def test_radiance_hdr_environment_reader(tmp_path, rle):
    rng = np.random.default_rng(33)
    h, w = 11, 40
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(120, 140, (h, w))
    rgbe[3, 5:25] = (10, 20, 30, 129)   # a run
    rgbe[7, :, 3] = 0                   # zero exponent = black
    path = str(tmp_path / "env.hdr")
    sf.write_radiance_hdr(path, rgbe, rle)
    sc = scene_io.SceneFile.empty()
    sc.load_environment(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    got = _env_pixels(sc)
    scale = np.ldexp(f32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(f32)   # stbi__hdr_convert
    want = np.where(rgbe[..., 3:4] != 0, rgbe[..., :3].astype(f32) * scale[..., None], f32(0.0))
    assert got.shape == (h, w, 4) and np.array_equal(got[..., :3], want) and np.all(got[..., 3] == 1.0)
    # and it renders: a scene lit by the file
    sc2 = scene_io.SceneFile.load(os.path.join(G, "scene_fixture", "mini.json"))
    sc2.load_environment(path)
    o = oracle_lib.OracleScene(sc2, make_params(32, 18, 1, 3))
    assert o.constants().envLightCount == 1 and np.isfinite(o.debug_sample(0)[0]).all()


EXRWRITE_PATH = os.path.join(sf.ROOT, "oracle", "_ref", "exrwrite")


@pytest.mark.skipif(not (os.path.exists(EXRWRITE_PATH) and os.path.exists(sf.EXR2RAW_PATH)), reason="oracle/_ref (reference tinyexr) not built")
@pytest.mark.parametrize("comp,ptype,chans,shape", [("piz", "half", 3, (70, 51)), ("piz", "float", 4, (33, 64)), ("piz", "half", 1, (40, 40)),
                                                     ("piz", "half", 4, (5, 3)), ("zip", "half", 3, (70, 51)), ("rle", "float", 3, (20, 31))])
def test_exr_files_written_by_the_reference_tinyexr(tmp_path, comp, ptype, chans, shape):
    """Files ENCODED by the reference's own tinyexr (oracle/_ref/exrwrite) — the only PIZ encoder here — read by the product and by
    the reference's LoadEXR to the same bits.  Smooth + noisy + constant content: both wavelet modes (14 / 16 bit), Huffman
    run-length codes and long codes get exercised."""
    import subprocess
    h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([np.sin(xx * 0.2) * 3 + 4, (yy * xx) % 7 * 0.5, rng.random((h, w)) * (60000.0 if ptype == "half" else 1e6), np.full((h, w), 0.75)], -1)[..., :chans]
    img[: h // 3, : w // 2] = 1.5   # a constant region -> run-length codes
    img = np.ascontiguousarray(img, dtype=f32)
    raw = str(tmp_path / "in.f32"); path = str(tmp_path / "t.exr")
    img.tofile(raw)
    subprocess.check_call([EXRWRITE_PATH, raw, str(w), str(h), str(chans), ptype, comp, path])
    sc = scene_io.SceneFile.empty()
    sc.load_environment(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    got = _env_pixels(sc)
    ref = sf.tinyexr_reference_rgba(path, str(tmp_path))
    assert got.shape == (h, w, 4) and got.tobytes() == ref.tobytes()
    want = img.astype(np.float16).astype(f32) if ptype == "half" else img
    if chans >= 3:
        assert np.array_equal(got[..., :3], want[..., :3])
    else:
        assert np.array_equal(got[..., 0], want[..., 0]) and np.array_equal(got[..., 3], want[..., 0])


def test_committed_piz_fixture():
    """tests/golden/exr_piz_fixture.exr (minted by tools/make_golden.py through the reference's tinyexr encoder) + its pixels."""
    g = np.load(os.path.join(G, "exr_piz_fixture.npz"))
    sc = scene_io.SceneFile.empty()
    sc.load_environment(os.path.join(G, "exr_piz_fixture.exr"))
    sc.add_camera((0, 0, 3), (0, 0, 0))
    assert _env_pixels(sc).tobytes() == g["rgba"].tobytes()


def _malformed_piz_files(tmp_path):
    """PIZ blocks whose Huffman length tables are not prefix codes (over-subscribed), which tinyexr's hufBuildDecTable rejects."""
    out = []
    for k, lengths in enumerate([[1, 1, 1, 1], [1, 1, 1], [2, 2, 2, 2, 2], [12] * 4097 + [0], [13, 1, 1, 1]]):
        path = str(tmp_path / f"bad_piz_{k}.exr")
        sf.write_exr_blocks(path, ["Y"], 8, 1, "half", 4, 32, [sf.piz_block_with_code_lengths(lengths)])
        out.append(path)
    return out


def test_piz_huffman_rejects_tables_that_are_not_prefix_codes(tmp_path):
    """an over-subscribed code-length table (three symbols of length 1 ...) used to index past the 4096-entry decode
    table.  The environment loader must fail cleanly, as the reference's tinyexr does."""
    for path in _malformed_piz_files(tmp_path):
        sc = scene_io.SceneFile.empty()
        with pytest.raises(abi.PtamdError, match="PIZ"):
            sc.load_environment(path)


def test_piz_huffman_malformed_tables_under_address_sanitizer(tmp_path):
    """The same inputs through a CPU AddressSanitizer build of scene_image.cpp (sanitizers run on the CPU build only)."""
    src = os.path.join(os.path.dirname(__file__), "..", "platinum_amd", "csrc")
    main = tmp_path / "asan_main.cpp"
    main.write_text(
        '#include "scene_io.h"\n#include <cstdio>\n#include <stdexcept>\n'
        'int main(int argc, char** argv) { int bad = 0; for (int i = 1; i < argc; i++) { uint32_t w, h; '
        'try { (void)ptio::read_exr_rgba(argv[i], &w, &h); printf("decoded %s\\n", argv[i]); } '
        'catch (const std::runtime_error& e) { printf("rejected: %s\\n", e.what()); bad++; } } return bad == argc - 1 ? 0 : 3; }\n')
    exe = str(tmp_path / "asan_exr")
    import subprocess
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", src,
                         "-I", os.path.join(src, "..", "..", "include"), str(main), os.path.join(src, "scene_image.cpp"), "-lz", "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "asan" in cc.stderr.lower():
        pytest.skip("libasan not installed")
    assert cc.returncode == 0, cc.stderr
    files = _malformed_piz_files(tmp_path) + [os.path.join(G, "exr_piz_fixture.exr")]
    run = subprocess.run([exe] + files[:-1], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr       # every malformed file rejected, no sanitizer report
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
    good = subprocess.run([exe, files[-1]], capture_output=True, text=True)
    assert "decoded" in good.stdout and "AddressSanitizer" not in good.stderr, good.stdout + good.stderr


# ---- JPEG textures (stbi_load semantics, loaders/texture.cpp:101-119) -----------------------------------------------------
def jpeg_test_image(w, h, mode):
    """A deterministic gradient + noise image as a PIL image (PIL / libjpeg is the ENCODER here: test infrastructure)."""
    from PIL import Image
    rng = np.random.default_rng(w * 1000 + h)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([x * 255 // max(1, w - 1), y * 255 // max(1, h - 1), (x + y) * 7 % 256], -1)
    a = np.clip(base + rng.integers(0, 60, (h, w, 3)) - 30, 0, 255).astype(np.uint8)
    if mode == "L":
        return Image.fromarray(a[..., 0], "L")
    if mode == "CMYK":
        return Image.fromarray(np.concatenate([a, a[..., :1]], -1), "CMYK")
    return Image.fromarray(a, "RGB")


def jpeg_fixture_cases():
    return [(33, 17, "RGB", dict(quality=90, subsampling=0)), (64, 48, "RGB", dict(quality=75, subsampling=1)),
            (129, 97, "RGB", dict(quality=60, subsampling=2)), (65, 33, "RGB", dict(quality=85, subsampling=2, progressive=True)),
            (40, 40, "RGB", dict(quality=95, subsampling=0, progressive=True, optimize=True)), (31, 29, "L", dict(quality=80)),
            (50, 20, "L", dict(quality=70, progressive=True)), (24, 24, "CMYK", dict(quality=85)),
            (100, 37, "RGB", dict(quality=80, subsampling=2, restart_marker_blocks=3))]


def test_jpeg_decoder_reproduces_the_reference_stb_image_on_the_committed_files():
    """tests/golden/jpeg_stb_fixture.npz: file bytes + the RGBA8 decoded by the reference's own stb_image (tools/make_golden.py)."""
    g = np.load(os.path.join(G, "jpeg_stb_fixture.npz"))
    n = len([k for k in g.files if k.startswith("jpg_")])
    assert n == len(jpeg_fixture_cases())
    for k in range(n):
        got = scene_io.decode_image_rgba8(g[f"jpg_{k}"].tobytes())
        assert got.shape == g[f"rgba_{k}"].shape and np.array_equal(got, g[f"rgba_{k}"]), k


def test_jpeg_decoder_vs_the_reference_stb_image_over_many_variants(tmp_path):
    """Every sampling mode x baseline / progressive (libjpeg's scripts use successive approximation) / optimised tables /
    restart intervals x odd sizes down to 1x1: byte-identical to stb_image v2.30 compiled from the reference tree."""
    pytest.importorskip("PIL")
    if not sf.stbi_available():
        pytest.skip("oracle/_ref/stbi2raw not built (reference tree absent)")
    import io
    n = 0
    for (w, h) in [(1, 1), (7, 5), (16, 16), (33, 17), (129, 97), (250, 3)]:
        for mode in ("RGB", "L", "CMYK"):
            for kw in (dict(quality=90, subsampling=0), dict(quality=75, subsampling=1), dict(quality=60, subsampling=2),
                       dict(quality=85, subsampling=2, progressive=True), dict(quality=30, subsampling=1, optimize=True),
                       dict(quality=80, subsampling=2, restart_marker_blocks=3)):
                if mode != "RGB":
                    kw = {k: v for k, v in kw.items() if k != "subsampling"}
                buf = io.BytesIO()
                jpeg_test_image(w, h, mode).save(buf, "JPEG", **kw)
                path = tmp_path / "t.jpg"
                path.write_bytes(buf.getvalue())
                assert np.array_equal(scene_io.decode_image_rgba8(buf.getvalue()), sf.stbi_reference_rgba(path)), (w, h, mode, kw)
                n += 1
    assert n == 108


def test_jpeg_and_image_decoder_errors():
    g = np.load(os.path.join(G, "jpeg_stb_fixture.npz"))
    data = g["jpg_0"].tobytes()
    for bad in (data[:2], data[:40], b"\xff\xd8\xff\xc9" + data[4:], b"GIF89a" + bytes(32), data[:-(len(data) // 2)].replace(b"\xff\xc4", b"\xff\xcc", 1)):
        try:
            img = scene_io.decode_image_rgba8(bad)
        except abi.PtamdError:
            continue
        assert img.shape[2] == 4  # (a scan cut short decodes to the rows that were present, as stb_image does)


def test_gltf_import_decodes_jpeg_textures(tmp_path):
    """A .glb whose base-colour texture is a JPEG in a bufferView (what Sponza-class assets ship)."""
    g = np.load(os.path.join(G, "jpeg_stb_fixture.npz"))
    b = sf.GltfBuilder()
    tex = b.image_png(g["jpg_2"].tobytes(), embed="view")     # (the builder stores the bytes as given)
    b.doc["images"][-1]["mimeType"] = "image/jpeg"
    b.doc["materials"].append({"pbrMetallicRoughness": {"baseColorTexture": {"index": tex}}})
    path = str(tmp_path / "jpeg_tex.glb")
    b.write(path, glb=True)
    sc = scene_io.SceneFile.empty().import_gltf(path)
    sc.add_camera((0, 0, 3), (0, 0, 0))
    t = snap_textures(sc.snapshot().struct)
    want = g["rgba_2"]
    assert (t[0][0], t[0][1], t[0][2]) == (want.shape[1], want.shape[0], abi.TEX_RGBA8_SRGB)
    assert t[0][3] == want.tobytes()
