"""CPU: SURVEY §8f row N3 — scene textures (every slot of core/material.hpp:16-23), normal maps, the alpha-test
intersection function (intersections.metal:8-39) and the environment light (kernel.metal:20-34, 440-467, 517-539;
core/environment.cpp:5-91).  The oracle is checked against independent numpy restatements (double precision, stated
tolerance) and structural properties; the product's stage functions (host build, tests/emu) against the oracle, bit-exact.
The GPU parity tests of the same scenes are in test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

import emu_lib
import oracle_lib
from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

G = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def _srgb_eotf(i):
    c = i / 255.0
    return c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4


def _bilinear_repeat(px, u, v):
    """Independent double-precision restatement of the filtering contract (oracle/pt_oracle.cpp tex_sample)."""
    h, w, _ = px.shape
    x, y = float(f32(u)) * w - 0.5, float(f32(v)) * h - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    wx, wy = x - x0, y - y0
    t = lambda xx, yy: px[yy % h, xx % w].astype(np.float64)
    a = t(x0, y0) * (1 - wx) + t(x0 + 1, y0) * wx
    b = t(x0, y0 + 1) * (1 - wx) + t(x0 + 1, y0 + 1) * wx
    return a * (1 - wy) + b * wy


def _one_texture_scene(pixels, fmt):
    sc = scenes.cornell_scene()
    t = sc.add_texture(pixels, fmt)
    return sc, t


def test_texel_decode_per_format():
    p = make_params(8, 8, 1, 2)
    rgba = np.array([[[0, 128, 255, 64], [10, 11, 12, 13]]], dtype=np.uint8)  # 1 x 2
    for fmt in (abi.TEX_RGBA8_SRGB, abi.TEX_RGBA8):
        sc, t = _one_texture_scene(rgba, fmt)
        o = oracle_lib.OracleScene(sc, p)
        got = o.tex_sample(t, 0.25, 0.5)  # texel centre of (0, 0)
        if fmt == abi.TEX_RGBA8_SRGB:
            want = [f32(_srgb_eotf(0)), f32(_srgb_eotf(128)), f32(_srgb_eotf(255)), f32(64) / f32(255)]
            assert abs(float(want[1]) - 0.21586050) < 1e-7  # the well-known value of sRGB 128
        else:
            want = [f32(v) / f32(255) for v in (0, 128, 255, 64)]
        assert [float(a) for a in got] == [float(b) for b in want], fmt
    sc, t = _one_texture_scene(np.array([[[51, 204]]], dtype=np.uint8), abi.TEX_RG8)
    assert list(oracle_lib.OracleScene(sc, p).tex_sample(t, 0.5, 0.5)) == [f32(51) / f32(255), f32(204) / f32(255), 0.0, 1.0]
    sc, t = _one_texture_scene(np.array([[77]], dtype=np.uint8), abi.TEX_R8)
    assert list(oracle_lib.OracleScene(sc, p).tex_sample(t, 0.1, 0.9)) == [f32(77) / f32(255), 0.0, 0.0, 1.0]
    hdr = np.array([[[1.5, 2.5, 1e4, -3.0]]], dtype=f32)
    sc, t = _one_texture_scene(hdr, abi.TEX_RGBA32F)
    assert list(oracle_lib.OracleScene(sc, p).tex_sample(t, 0.3, 0.3)) == [1.5, 2.5, 1e4, -3.0]


def test_bilinear_repeat_filter_vs_numpy():
    rng = np.random.default_rng(5)
    px = rng.random((5, 7, 4), dtype=np.float32)
    sc, t = _one_texture_scene(px, abi.TEX_RGBA32F)
    o = oracle_lib.OracleScene(sc, make_params(8, 8, 1, 2))
    uv = np.concatenate([rng.uniform(-2.5, 3.5, (300, 2)), [[0, 0], [1, 1], [0.5 / 7, 0.5 / 5], [-1e-7, 1 + 1e-7], [0.999999, 0.0]]]).astype(f32)
    for u, v in uv:
        got = o.tex_sample(t, float(u), float(v)).astype(np.float64)
        want = _bilinear_repeat(px, u, v)
        assert np.max(np.abs(got - want)) < 4e-6, (u, v, got, want)  # fp32 lerp vs double: a few ulp of O(1) values + |uv|*W ulp
    # a texel centre returns the texel exactly; the seam wraps
    assert list(o.tex_sample(t, (3 + 0.5) / 7, (2 + 0.5) / 5)) == list(px[2, 3])
    assert np.allclose(o.tex_sample(t, 0.0, 0.5), 0.5 * (px[2, 6] + px[2, 0]), atol=1e-6)


def test_atan2_acos_and_direction_mapping():
    L = oracle_lib.lib()
    rng = np.random.default_rng(9)
    xy = rng.normal(size=(2000, 2)).astype(f32)
    got = np.array([L.orc_atan2(float(y), float(x)) for x, y in xy])
    assert np.max(np.abs(got - np.arctan2(xy[:, 1].astype(np.float64), xy[:, 0].astype(np.float64)))) < 5e-7  # ~2 ulp at pi
    for y, x, want in [(0, 1, 0.0), (1, 0, np.pi / 2), (-1, 0, -np.pi / 2), (0, -1, np.pi), (0, 0, 0.0), (1, 1, np.pi / 4), (-1, -1, -3 * np.pi / 4)]:
        assert abs(L.orc_atan2(y, x) - want) < 3e-7
    c = np.concatenate([rng.uniform(-1, 1, 1000), [-1, 1, 0, 1.5, -1.5]]).astype(f32)
    got = np.array([L.orc_acos(float(v)) for v in c])
    assert np.max(np.abs(got - np.arccos(np.clip(c.astype(np.float64), -1, 1)))) < 6e-7
    # kernel.metal:20-34: uvToRayDir(rayDirToUv(d)) = d; uv = (phi / 2pi, theta / pi) with phi = atan2(-z, -x)
    d = rng.normal(size=(500, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    for v in d:
        uv = np.zeros(2, f32); back = np.zeros(3, f32)
        L.orc_ray_dir_to_uv(v.ctypes.data, uv.ctypes.data)
        assert abs(uv[0] - np.arctan2(-float(v[2]), -float(v[0])) / (2 * np.pi)) < 2e-7 and abs(uv[1] - np.arccos(float(v[1])) / np.pi) < 2e-7
        assert -0.5 <= uv[0] <= 0.5 and 0 <= uv[1] <= 1
        L.orc_uv_to_ray_dir(uv.ctypes.data, back.ctypes.data)
        assert np.max(np.abs(back - v)) < 2e-6


def test_alias_table_is_a_valid_vose_table():
    """core/environment.cpp:5-91: pdf = luma * n / sum(luma); the table must reproduce pdf/n as sampling probability."""
    sc = scenes.textured_scene()
    o = oracle_lib.OracleScene(sc, make_params(16, 16, 1, 2))
    al = o.envAlias()
    env = sc.textures[sc.env_texture].pixels
    n = env.shape[0] * env.shape[1]
    assert len(al) == n and o.constants().envLightCount == 1
    luma = (env[..., :3].astype(np.float64) @ np.array([0.2126, 0.7152, 0.0722])).ravel()
    assert np.allclose(al["pdf"], luma * n / luma.sum(), rtol=2e-5)
    assert np.all((al["p"] >= 0) & (al["p"] <= 1)) and np.all(al["aliasIdx"] < n)
    prob = al["p"].astype(np.float64) / n
    np.add.at(prob, al["aliasIdx"], (1.0 - al["p"].astype(np.float64)) / n)
    assert np.allclose(prob, al["pdf"].astype(np.float64) / n, rtol=0, atol=2e-5 / n * al["pdf"].max())
    assert abs(prob.sum() - 1.0) < 1e-5
    # an entry that redirects does so to an above-average texel
    redir = al["p"] < 1.0
    assert np.all(al["pdf"][al["aliasIdx"][redir]] >= 1.0)


def test_environment_only_scene_primary_radiance_is_the_texture_lookup():
    """No geometry hit on bounce 0: L = environment(rayDirToUv(d)) (kernel.metal:517-526)."""
    sc = scenes.Scene(name="sky")
    q = sc.add_mesh(scenes.plane(1.0))
    sc.add_instance(q, scenes.Transform(translation=(0, -50, 0)), [scenes.Material()])  # far below, outside the view
    sc.env_texture = sc.add_texture(scenes.sky_environment(), abi.TEX_RGBA32F)
    sc.set_camera(scenes.Camera.with_focal_length(20.0), scenes.Transform(translation=(0, 1, 0), target=(3, 2.5, -1), track=True))
    p = make_params(48, 32, 1, 3)
    o = oracle_lib.OracleScene(sc, p)
    rad, hits = o.debug_sample(0)
    assert np.all(hits[0, ..., 0] == -1)
    C_ = o.constants().camera
    L = oracle_lib.lib()
    env = sc.textures[sc.env_texture].pixels
    pos = np.array([C_.position.x, C_.position.y, C_.position.z], np.float64)
    tl = np.array([C_.topLeft.x, C_.topLeft.y, C_.topLeft.z], np.float64)
    du = np.array([C_.pixelDeltaU.x, C_.pixelDeltaU.y, C_.pixelDeltaU.z], np.float64)
    dv = np.array([C_.pixelDeltaV.x, C_.pixelDeltaV.y, C_.pixelDeltaV.z], np.float64)
    for (x, y) in [(0, 0), (47, 31), (20, 10), (5, 29), (33, 3)]:
        off = L.orc_halton_offset(x, y, 0)
        d = tl + (x + L.orc_halton(off, 0)) * du + (y + L.orc_halton(off, 1)) * dv - pos
        d /= np.linalg.norm(d)
        u, v = np.arctan2(-d[2], -d[0]) / (2 * np.pi), np.arccos(d[1]) / np.pi
        want = _bilinear_repeat(env, u, v)[:3]
        assert np.allclose(rad[y, x, :3], want, rtol=2e-4, atol=1e-5), (x, y, rad[y, x], want)


def test_stochastic_alpha_test_statistics():
    """intersections.metal:8-39: a candidate hit counts iff alpha > r, r one Halton dimension per ray: a constant-alpha
    0.5 sphere stops about half of the primary rays that cross it and none of the shading normals change."""
    sc = scenes.textured_scene(env=True, area_light=False)
    ghost = [i for i, n in enumerate(sc.nodes) if n.materials[0].name == "ghost"][0]
    p = make_params(96, 54, 32, 2)
    o = oracle_lib.OracleScene(sc, p)
    sc_opaque = scenes.textured_scene(env=True, area_light=False)
    sc_opaque.nodes[ghost].materials[0].base_color = (0.2, 0.8, 0.3, 1.0)
    oo = oracle_lib.OracleScene(sc_opaque, p)
    hit = tot = 0
    for s in range(32):
        a, b = o.trace_primary(s), oo.trace_primary(s)
        crosses = b["instance"] == ghost
        tot += int(crosses.sum())
        hit += int((a["instance"][crosses] == ghost).sum())
        # accepted hits are the same candidates: identical (t, u, v)
        same = crosses & (a["instance"] == ghost) & (a["primitive"] == b["primitive"])
        assert np.array_equal(a["t"][same], b["t"][same])
    assert tot > 3000
    # P(accept first surface) = 0.5; a ray refused at the front face is tested again at the back face with the same r
    # and is refused again, so the fraction is 0.5 (binomial sigma ~ 0.009 here)
    assert abs(hit / tot - 0.5) < 0.04, hit / tot


@pytest.mark.parametrize("kw,integrator", [
    (dict(), abi.INTEGRATOR_MIS),
    (dict(), abi.INTEGRATOR_SIMPLE),
    (dict(area_light=False), abi.INTEGRATOR_MIS),   # lightCount == 0: pInfinite = 1 (kernel.metal:593-596)
    (dict(env=False), abi.INTEGRATOR_MIS),          # textures + cut-outs with area lights only
    (dict(alpha=False), abi.INTEGRATOR_MIS),        # no non-opaque instance: payload never evaluated
])
def test_stage_functions_bit_exact_vs_oracle_textured(kw, integrator):
    sc = scenes.textured_scene(**kw)
    p = make_params(88, 50, 2, 6, integrator=integrator)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert bytes(o.constants()) == bytes(e.constants())
    assert [bytes(a) for a in o.lights()] == [bytes(b) for b in e.lights()]
    assert o.trace_primary(1).tobytes() == e.trace_primary(1).tobytes()
    for s in (0, 1):
        ro, ho = o.debug_sample(s)
        re_, he = e.debug_sample(s)
        assert np.array_equal(ho, he)
        assert ro.tobytes() == re_.tobytes()
    assert np.isfinite(ro).all() and ro[..., :3].mean() > 1e-3


@pytest.mark.parametrize("native", ["0", "1"])
def test_eight_bit_textures_decoded_on_fetch_give_the_same_bits(native, monkeypatch):
    """r4 texture storage (host_scene.h decode_textures, pt_bsdf.h tex_fetch): 8-bit textures either decoded once to float4 or kept 8-bit and
    decoded per tap through the unorm / sRGB tables.  Both forms ($PTAMD_TEX_NATIVE forces one) must give the oracle's radiance bit for
    bit — every format is in this scene: sRGB8 base colour, RG8 roughness / metallic, RGBA8 normals, R8 transmission, RGBA32F environment."""
    monkeypatch.setenv("PTAMD_TEX_NATIVE", native)
    sc = scenes.random_scene(1003, extras=True)
    for sc in (scenes.textured_scene(), scenes.random_scene(1003, extras=True)):
        p = make_params(72, 40, 1, 6)
        o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
        ro, ho = o.debug_sample(0)
        re_, he = e.debug_sample(0)
        nan = np.isnan(ro)
        assert np.array_equal(ho, he) and np.array_equal(nan, np.isnan(re_)) and np.array_equal(ro.view(np.uint32)[~nan], re_.view(np.uint32)[~nan])


def test_atrium_c5_class_scene_stage_functions_bit_exact():
    """BASELINE configs[4] stand-in (scenes.atrium_scene) at toy size: every N3 feature in one scene, 12 bounces."""
    sc = scenes.atrium_scene(env_size=(64, 32), columns=3)
    p = make_params(80, 45, 1, 12)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert bytes(o.constants()) == bytes(e.constants()) and o.constants().envLightCount == 1
    assert [bytes(a) for a in o.lights()] == [bytes(b) for b in e.lights()]
    ro, ho = o.debug_sample(0)
    re_, he = e.debug_sample(0)
    assert np.array_equal(ho, he) and ro.tobytes() == re_.tobytes()
    assert (ho[0, ..., 0] >= 0).mean() > 0.8 and (ho[6:, ..., 0] >= 0).any()   # enclosed hall: long paths exist


def test_bvh_and_brute_force_agree_with_alpha_test():
    """The alpha test is applied to every candidate, so closest-hit selection stays BVH-independent."""
    sc = scenes.textured_scene()
    p = make_params(64, 36, 1, 5)
    a = oracle_lib.OracleScene(sc, p, use_bvh=True)
    b = oracle_lib.OracleScene(sc, p, use_bvh=False)
    ra, ha = a.debug_sample(0)
    rb, hb = b.debug_sample(0)
    assert np.array_equal(ha, hb) and ra.tobytes() == rb.tobytes()


def test_emission_texture_makes_area_lights_and_modulates_le():
    sc = scenes.textured_scene(env=False)
    o = oracle_lib.OracleScene(sc, make_params(32, 18, 1, 2))
    lights = o.lights()
    assert len(lights) == 2  # the panel's two triangles (material.hpp:44-47: emissive through its emission texture too)
    sc2 = scenes.textured_scene(env=False)
    panel = [n for n in sc2.nodes if n.materials[0].name == "panel"][0]
    panel.materials[0].emission = (0.0, 0.0, 0.0)  # still "emissive" because of the texture slot; zero power
    o2 = oracle_lib.OracleScene(sc2, make_params(32, 18, 1, 2))
    assert len(o2.lights()) == 2 and o2.constants().totalLightPower == 0.0


def test_textured_golden_fixture():
    """tests/golden/n3_textured_golden.npz (tools/make_golden.py): pins the oracle's N3 arithmetic against regressions."""
    g = np.load(os.path.join(G, "n3_textured_golden.npz"))
    sc = scenes.textured_scene()
    o = oracle_lib.OracleScene(sc, make_params(96, 54, 2, 6))
    acc = o.render(0, 2)
    rad0, hits0 = o.debug_sample(0)
    assert np.array_equal(hits0.astype(np.int16), g["hits0"])
    assert acc.tobytes() == g["acc2"].tobytes()
    al = o.envAlias()
    assert al["pdf"].tobytes() == g["alias_pdf"].tobytes() and al["p"].tobytes() == g["alias_p"].tobytes()
    assert np.array_equal(al["aliasIdx"], g["alias_idx"])
