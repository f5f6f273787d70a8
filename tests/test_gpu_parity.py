"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical (pixel, sample) streams.

Bars: hit (instance, primitive) ids bit-exact at every bounce; radiance bit-exact per sample for the
configurations below (both sides implement the same deterministic fp32 contract independently), with a stated
fallback tolerance of 1e-5 relative on per-sample radiance should a future compiler change one rounding.
"""
import os

import numpy as np
import pytest

from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

import oracle_lib
from conftest import skip_if_structure_env_preset

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # stated float tolerance for per-sample radiance (we currently observe exact equality)


def _scene(name):
    if name == "cornell":
        return scenes.cornell_scene("bench")
    if name == "cornell_default_cam":
        return scenes.cornell_scene("default")
    if name == "cornell_sphere":
        return scenes.cornell_sphere_scene()
    if name == "cornell_sphere_opaque":
        return scenes.cornell_sphere_scene(transmission=0.0)
    if name == "field8":
        return scenes.field_scene(8)
    raise KeyError(name)


def _start(r, scene, w, h, spp, bounces, integrator=abi.INTEGRATOR_MIS, **kw):
    r.selectKernel(integrator)
    kw = dict(kw)
    space = kw.pop("working_space", scenes.BT2020)
    r.startRender(scene, (w, h), spp, workingSpace=space, max_bounces=bounces, **kw)
    # (the oracle's parameters: what changes the arithmetic; first_sample is passed to OracleScene.render by the caller)
    return make_params(w, h, spp, bounces, flags=kw.get("flags", abi.FLAG_MULTISCATTER_GGX), integrator=integrator, working_space=space)


@pytest.mark.parametrize("name", ["cornell", "cornell_default_cam", "cornell_sphere", "field8"])
def test_constants_and_lights_match_oracle(gpu_renderer, name):
    sc = _scene(name)
    p = _start(gpu_renderer, sc, 64, 48, 1, 4)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    lg, lo = gpu_renderer.lights(), o.lights()
    assert len(lg) == len(lo) and all(bytes(a) == bytes(b) for a, b in zip(lg, lo))


@pytest.mark.parametrize("name,w,h", [("cornell", 512, 512), ("cornell_sphere", 320, 180), ("field8", 320, 180)])
def test_primary_hits_bit_exact(gpu_renderer, name, w, h):
    sc = _scene(name)
    p = _start(gpu_renderer, sc, w, h, 1, 1)
    o = oracle_lib.OracleScene(sc, p)
    for s in (0, 3):
        g, c = gpu_renderer.tracePrimary(s), o.trace_primary(s)
        assert np.array_equal(g["instance"], c["instance"])
        assert np.array_equal(g["primitive"], c["primitive"])
        for k in "tuv":
            assert np.array_equal(g[k].view(np.uint32), c[k].view(np.uint32)), k
    assert (g["instance"] >= 0).mean() > 0.5


@pytest.mark.parametrize("name,integrator,bounces", [
    ("cornell", abi.INTEGRATOR_MIS, 4),
    ("cornell", abi.INTEGRATOR_SIMPLE, 6),
    ("cornell_sphere", abi.INTEGRATOR_MIS, 8),
    ("cornell_sphere_opaque", abi.INTEGRATOR_MIS, 5),
    ("field8", abi.INTEGRATOR_MIS, 8),
])
def test_per_bounce_hit_ids_and_radiance(gpu_renderer, name, integrator, bounces):
    sc = _scene(name)
    w, h = 160, 96
    p = _start(gpu_renderer, sc, w, h, 4, bounces, integrator=integrator)
    o = oracle_lib.OracleScene(sc, p)
    for s in (0, 2):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc), f"hit ids differ at {np.argwhere((hg != hc).any(-1))[:5]}"
        np.testing.assert_allclose(rg, rc, rtol=REL_TOL, atol=1e-7)
        assert np.array_equal(rg.view(np.uint32), rc.view(np.uint32)), "radiance not bit-identical"


@pytest.mark.parametrize("integrator", [abi.INTEGRATOR_MIS, abi.INTEGRATOR_SIMPLE])
def test_fifty_bounces_reach_the_end_of_the_sampler_table(gpu_renderer, integrator):
    """max_bounces = 50 (kernel.metal:5) inside a CLOSED glass-sphere Cornell box: the white walls have albedo 1, so roulette lets
    paths live; the last bounces draw Halton dimensions around 600 of the 620 — far beyond the window k_shade stages in LDS — and the
    per-bounce counters are used up to index 49."""
    sc = _scene("cornell_sphere")
    # close the open +z face with a white wall and put the camera inside: nothing escapes
    wall = sc.add_mesh(scenes.plane(10.0))
    sc.add_instance(wall, scenes.Transform(translation=(0, 5, 5), rotation=(-np.pi / 2, 0, 0)), [scenes.Material(base_color=(1, 1, 1, 1))])
    sc.set_camera(scenes.Camera.with_focal_length(20.0), scenes.Transform(translation=(0, 5, 4.5), target=(0, 4, 0), track=True))
    w, h = 64, 36
    p = _start(gpu_renderer, sc, w, h, 2, 50, integrator=integrator)
    o = oracle_lib.OracleScene(sc, p)
    rg, hg = gpu_renderer.debugSample(1)
    rc, hc = o.debug_sample(1)
    assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    alive = (hg.reshape(50, h * w, 2)[:, :, 0] >= 0).sum(1)
    assert alive[49] > 0, alive            # some paths really are that long
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))
    with pytest.raises(abi.PtamdError, match="max_bounces must be 1..50"):
        gpu_renderer.startRender(sc, (w, h), 1, max_bounces=51)


def test_accumulator_matches_oracle_c1_small(gpu_renderer):
    """C1 (Cornell, 4 bounces) at reduced size/spp: running-mean accumulator, several batches."""
    sc = _scene("cornell")
    w, h, spp = 128, 128, 12
    p = _start(gpu_renderer, sc, w, h, spp, 4, samples_in_flight=5)  # 5 + 5 + 2: exercises batching
    while gpu_renderer.status() & abi.STATUS_BUSY:
        gpu_renderer.render(1)
    acc = gpu_renderer.readbackAccumulator()
    assert gpu_renderer.status() == abi.STATUS_READY | abi.STATUS_DONE
    assert gpu_renderer.renderProgress() == (spp, spp)
    o = oracle_lib.OracleScene(sc, p)
    ref = o.render(0, spp)
    assert np.array_equal(acc[..., 3], np.ones((h, w), np.float32))
    np.testing.assert_allclose(acc, ref, rtol=REL_TOL, atol=1e-7)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    # counters agree with the oracle's ray counts
    st, so = gpu_renderer.stats(), o.stats()
    assert (st.closest_rays, st.shadow_rays, st.shaded_hits, st.paths) == (so.closest_rays, so.shadow_rays, so.shaded_hits, so.paths)


def test_one_sample_render_calls_are_merged_into_batches_and_give_the_same_image(gpu_renderer):
    """The reference's frontend calls render() once per UI frame for ONE sample (renderer_pt.cpp:131-153, frontend.cpp:209-210).
    Calls that arrive while the GPU is busy are merged into batches (renderer.hip flush_pending); the accumulator is
    bit-identical to one big step, progress counts accepted samples, and nothing stays pending once the render is Done."""
    sc = _scene("cornell_sphere")
    w, h, spp, B = 1280, 720, 48, 6   # (a one-sample batch of this size keeps the GPU busy for milliseconds: many host calls long)
    r = gpu_renderer
    _start(r, sc, w, h, spp, B, samples_in_flight=16)
    r.render(0)
    ref = r.readbackAccumulator()
    ref_batches = r.stats().batches
    assert ref_batches == 3
    _start(r, sc, w, h, spp, B, samples_in_flight=16)
    n = 0
    while r.status() & abi.STATUS_BUSY:      # the frontend's loop, as fast as the host can call
        r.render(1)
        n += 1
        assert r.renderProgress() == (n, spp)
    assert n == spp and r.status() == abi.STATUS_READY | abi.STATUS_DONE
    got = r.readbackAccumulator()
    st = r.stats()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert 3 <= st.batches < spp, st.batches   # merged (48 one-sample batches would be the old behaviour)
    # a caller slower than the GPU still gets one batch per call
    _start(r, sc, w, h, 3, B, samples_in_flight=16)
    for _ in range(3):
        r.render(1)
        r.wait()
    assert r.stats().batches == 3


def test_merged_render_calls_with_gmon_resolve_like_one_step(gpu_renderer):
    """The merged batches of a render(1) loop cross GMoN bucket boundaries wherever they happen to fall; buckets and the resolved image
    must not depend on it (the reference resolves after every frame: only the state after the last one is observable)."""
    sc = _scene("cornell_sphere")
    w, h, spp, B, buckets = 160, 96, 30, 5, 15
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    r = gpu_renderer
    r.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=B, samples_in_flight=8)
    r.render(0)
    ref = r.readbackAccumulator()
    ref_b = [r.readGmonBucket(b) for b in range(buckets)]
    r.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=B, samples_in_flight=8)
    while r.status() & abi.STATUS_BUSY:
        r.render(1)
    got = r.readbackAccumulator()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    for b in range(buckets):
        assert r.readGmonBucket(b).tobytes() == ref_b[b].tobytes(), b
    # a progressive read in the middle of the loop shows every accepted sample (presenting / reading flushes what is pending)
    r.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=B, samples_in_flight=8)
    for _ in range(11):
        r.render(1)
    mid = r.readbackAccumulator()
    assert r.renderProgress() == (11, spp)
    # (11 samples touch 6 buckets: an EVEN count, for which the reference's resolve divides 0 by 0 on all-black pixels — G = NaN,
    #  min(NaN, cap) = cap, c = n / 2, gmon.metal:44-53 — reproduced as is: the images must agree bit for bit, NaNs included)
    assert np.isnan(mid[..., 0]).any() and np.isfinite(mid[..., 0]).any()
    r.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=B, samples_in_flight=8)
    r.render(11)
    assert np.array_equal(r.readbackAccumulator().view(np.uint32), mid.view(np.uint32))


def test_restarts_reuse_device_arrays_across_sizes_scenes_and_modes():
    """r03: device arrays are kept across pt_start_render (a restart took 1.3-1.5 s when they were released and re-allocated) and
    only re-allocated to grow.  A renderer that has been through bigger / smaller images, other scenes, GMoN and the two-level
    structure must give the same bits as a fresh one — nothing stale may leak from a longer array into a shorter use."""
    from platinum_amd import Renderer
    seq = [
        ("cornell", 64, 64, 3, 4, {}),
        ("field8", 256, 128, 4, 6, {"samples_in_flight": 4}),
        ("cornell_sphere", 96, 54, 5, 8, {"gmonBuckets": 5, "flags": abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON}),
        ("cornell", 64, 64, 3, 4, {}),
        ("field8", 40, 24, 2, 5, {"accel_structure": abi.ACCEL_TWO_LEVEL}),
        ("field8", 256, 128, 4, 6, {"samples_in_flight": 2}),
    ]
    def render(r, name, w, h, spp, B, kw):
        r.startRender(_scene(name), (w, h), spp, max_bounces=B, **kw)
        r.render(0)
        return r.readbackAccumulator().copy(), r.readbackRenderTarget().copy()
    veteran = Renderer(device=0)
    try:
        got = [render(veteran, *c) for c in seq]
    finally:
        veteran.close()
    for c, (acc, img) in zip(seq, got):
        fresh = Renderer(device=0)
        try:
            ref_acc, ref_img = render(fresh, *c)
        finally:
            fresh.close()
        assert np.array_equal(acc.view(np.uint32), ref_acc.view(np.uint32)), c[:5]
        assert np.array_equal(img, ref_img), c[:5]
    assert np.array_equal(got[0][0].view(np.uint32), got[3][0].view(np.uint32))   # the same render before and after the others


@pytest.mark.parametrize("seed", [24, 648, 996])
def test_escaping_path_with_a_non_finite_throughput_gets_the_references_nan(gpu_renderer, seed):
    """Found by the r03 fuzz sweep (tests/fuzz_parity_sweep.py: 3 pixels in 1 000 random scenes, then 0 in 2 000 after the fix).  The
    reference adds `attenuation * backgroundColor` (= attenuation * 0, kernel.metal:311 / :541, defs.metal:21) to every path that
    escapes.  When the BSDF has driven the throughput to inf or NaN (a zero pdf) that term is NaN; the wavefront used to skip the
    misses of scenes without an environment (radiance 0 instead of NaN) and to leave out the term after the environment's radiance
    (inf instead of NaN).  Now a path whose throughput is not finite carries its miss processed by k_shade (whose scan of a scene without an environment reads the throughput of the escaping paths)."""
    sc = scenes.random_scene(seed)
    w, h, B, spp = 71, 45, 3 + seed % 7, 2 + seed % 2
    p = _start(gpu_renderer, sc, w, h, spp, B, integrator=abi.INTEGRATOR_SIMPLE, samples_in_flight=1 + seed % 3)
    o = oracle_lib.OracleScene(sc, p)
    saw_nan = False
    for s in range(spp):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc), s
        saw_nan = saw_nan or bool(np.isnan(rc).any())
    assert saw_nan   # (the scene still produces the case)
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, spp))


def test_sample_sharding_is_the_same_sample_set(gpu_renderer):
    """§8e: renderers with disjoint first_sample ranges together trace exactly the samples of one big render."""
    sc = _scene("cornell_sphere")
    w, h = 96, 64
    _start(gpu_renderer, sc, w, h, 8, 6)
    gpu_renderer.render(0)
    full = gpu_renderer.readbackAccumulator().astype(np.float64)
    parts = []
    for g in range(2):
        _start(gpu_renderer, sc, w, h, 4, 6, first_sample=4 * g)
        gpu_renderer.render(0)
        parts.append(gpu_renderer.readbackAccumulator().astype(np.float64))
    merged = (parts[0] + parts[1]) / 2
    np.testing.assert_allclose(merged[..., :3], full[..., :3], rtol=1e-5, atol=1e-6)


def test_full_size_properties_c2(gpu_renderer):
    """BASELINE size (1920x1080, 8 bounces): size-independent properties — finite, alpha 1, progressive mean is
    consistent (mean of two halves == whole), and energy is within a sane range."""
    sc = _scene("cornell_sphere")
    _start(gpu_renderer, sc, 1920, 1080, 4, 8)
    gpu_renderer.render(0)
    a = gpu_renderer.readbackAccumulator()
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and (a[..., :3] >= 0).all()
    m = a[..., :3].mean()
    assert 0.05 < m < 5.0
    st = gpu_renderer.stats()
    assert st.paths == 1920 * 1080 * 4 and st.closest_rays >= st.paths and st.triangles == 12 + 6144


def test_million_triangle_field_parity(gpu_renderer):
    """C3 scene at full triangle count (1 036 300 triangles: exercises the HBM spill part of the traversal stack and the chunk
    tables over thousands of segments) at reduced resolution."""
    sc = scenes.field_scene(32)
    w, h, bounces = 256, 144, 6
    p = _start(gpu_renderer, sc, w, h, 2, bounces)
    st = gpu_renderer.stats()
    assert st.triangles == 1036300 and st.bvh_max_depth >= 8 and 3 * st.bvh_max_depth > 13   # deeper than the LDS part of the stack
    o = oracle_lib.OracleScene(sc, p)
    g, c = gpu_renderer.tracePrimary(0), o.trace_primary(0)
    assert g.tobytes() == c.tobytes()
    rg, hg = gpu_renderer.debugSample(1)
    rc, hc = o.debug_sample(1)
    assert np.array_equal(hg, hc)
    assert np.array_equal(rg.view(np.uint32), rc.view(np.uint32))
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    ref = o.render(0, 2)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    so = o.stats()
    st = gpu_renderer.stats()
    assert (st.closest_rays, st.shadow_rays, st.shaded_hits) == (so.closest_rays, so.shadow_rays, so.shaded_hits)


def test_odd_image_size_and_many_batches(gpu_renderer):
    """Image size that is not a multiple of the 8x8 raygen tile, samples_in_flight that does not divide spp."""
    sc = _scene("cornell_sphere")
    w, h, spp = 203, 117, 7
    p = _start(gpu_renderer, sc, w, h, spp, 5, samples_in_flight=3)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    ref = oracle_lib.OracleScene(sc, p).render(0, spp)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("spp,buckets,sif", [(30, 15, 8), (17, 5, 3)])
def test_gmon_matches_oracle(gpu_renderer, spp, buckets, sif):
    """bucketed accumulation + Gini-weighted median of means, bit-identical to the oracle."""
    sc = _scene("cornell_sphere")
    w, h, bounces = 96, 64, 5
    flags = abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON
    gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
    gpu_renderer.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=bounces, samples_in_flight=sif)
    assert gpu_renderer.constants().gmonBuckets == buckets
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    o = oracle_lib.OracleScene(sc, make_params(w, h, spp, bounces, flags=flags, gmon_buckets=buckets))
    ob, oresolved = o.render_gmon(spp)
    for b in range(buckets):
        assert gpu_renderer.readGmonBucket(b).tobytes() == ob[b].tobytes(), b
    assert np.array_equal(acc.view(np.uint32), oresolved.view(np.uint32))
    gpu_renderer.setGmonOptions(cap=0.25)
    gpu_renderer.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=bounces, samples_in_flight=sif)
    gpu_renderer.render(0)
    spb = (spp + buckets - 1) // buckets
    assert np.array_equal(gpu_renderer.readbackAccumulator().view(np.uint32), o.gmon_resolve(ob, (spp - 1) // spb + 1, cap=0.25).view(np.uint32))
    gpu_renderer.setGmonOptions(cap=1.0)


def test_present_render_target_is_the_readback_image_on_the_device(gpu_renderer):
    """presentRenderTarget() (renderer_pt.hpp:55): the RGBA8 image stays in HBM; its bytes equal readbackRenderTarget()'s."""
    torch = pytest.importorskip("torch")
    sc = _scene("cornell_sphere")
    w, h = 96, 64
    _start(gpu_renderer, sc, w, h, 3, 5)
    gpu_renderer.render(0)
    gpu_renderer.wait()
    ptr, stream = gpu_renderer.presentRenderTarget()
    assert ptr and stream
    want = gpu_renderer.readbackRenderTarget()      # (synchronises the renderer's stream)
    import ctypes as C
    from platinum_amd import abi
    hip = abi.load_library()   # dlsym on the library's handle also searches its dependencies: the HIP runtime it renders with
    got = np.empty((h, w, 4), np.uint8)
    assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), C.c_void_p(ptr), C.c_size_t(got.nbytes), 2) == 0  # hipMemcpyDeviceToHost
    assert np.array_equal(got, want) and want[..., 3].min() == 255


def test_postprocess_tonemap_rgba8_matches_oracle(gpu_renderer):
    """exposure / CA / contrast-saturation / tone curve / vignette / tonemap / lift-gamma-gain / odt / sRGB
    -> RGBA8, identical to the oracle for the three tonemappers (at most 1 LSB on a handful of pixels is tolerated in
    case a future compiler changes one rounding; we currently see exact equality)."""
    sc = _scene("cornell_sphere")
    w, h = 160, 96
    _start(gpu_renderer, sc, w, h, 4, 5)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    o = oracle_lib.OracleScene(sc, make_params(w, h, 4, 5))
    for tm in (abi.TONEMAP_AGX, abi.TONEMAP_KHRONOS_PBR, abi.TONEMAP_FLIM, abi.TONEMAP_NONE):
        po, to = gpu_renderer.postProcessOptions(), gpu_renderer.tonemapOptions()
        to.tonemapper = tm
        if tm != abi.TONEMAP_AGX:                       # non-default grading on the other three
            po.exposure, po.ca_amount, po.contrast, po.saturation = 0.7, 40.0, 12.0, -8.0
            po.blacks, po.shadows, po.highlights, po.whites = 5.0, -10.0, 8.0, -4.0
            po.vig_amount, po.vig_midpoint = -1.5, 10.0
            to.shadow_color[0], to.highlight_color[2], to.midtone_offset = 0.55, 0.6, 4.0
            to.agx_power[0] = 1.35
        gpu_renderer.setPostProcessOptions(po)
        gpu_renderer.setTonemapOptions(to)
        g = gpu_renderer.readbackRenderTarget()
        c = o.postprocess(acc, po, to)
        diff = np.abs(g.astype(np.int16) - c.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (tm, diff.max(), (diff > 0).mean())
        assert np.array_equal(g, c), tm
        assert g[..., :3].std() > 5
    gpu_renderer.setPostProcessOptions(gpu_renderer.postProcessOptions())
    gpu_renderer.setTonemapOptions(gpu_renderer.tonemapOptions())


# ---- textures, normal maps, alpha-tested cut-outs, environment light ------------------------------
N3_VARIANTS = {
    "full": dict(),
    "env_only": dict(area_light=False),     # lightCount == 0: pInfinite = 1
    "no_env": dict(env=False),
    "opaque": dict(alpha=False),            # has_alpha == 0: the payload is never evaluated
}


@pytest.mark.parametrize("variant,integrator", [("full", abi.INTEGRATOR_MIS), ("full", abi.INTEGRATOR_SIMPLE),
                                                ("env_only", abi.INTEGRATOR_MIS), ("no_env", abi.INTEGRATOR_MIS),
                                                ("opaque", abi.INTEGRATOR_MIS)])
def test_textured_scene_hit_ids_and_radiance(gpu_renderer, variant, integrator):
    sc = scenes.textured_scene(**N3_VARIANTS[variant])
    w, h = 200, 112
    p = _start(gpu_renderer, sc, w, h, 4, 7, integrator=integrator)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    lg, lo = gpu_renderer.lights(), o.lights()
    assert len(lg) == len(lo) and all(bytes(a) == bytes(b) for a, b in zip(lg, lo))
    ag, ao = gpu_renderer.envAlias(), o.envAlias()
    assert ag.tobytes() == ao.tobytes() and (len(ag) > 0) == (variant != "no_env")
    for s in (0, 3):
        g, c = gpu_renderer.tracePrimary(s), o.trace_primary(s)
        assert g.tobytes() == c.tobytes()
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc), f"hit ids differ at {np.argwhere((hg != hc).any(-1))[:5]}"
        np.testing.assert_allclose(rg, rc, rtol=REL_TOL, atol=1e-7)
        assert rg.tobytes() == rc.tobytes()
    assert rc[..., :3].mean() > 1e-3


@pytest.mark.parametrize("native", ["0", "1"])
def test_textures_kept_eight_bit_in_hbm_give_the_same_image(gpu_renderer, native, monkeypatch):
    """r4: 8-bit textures stored decoded (small texture sets) or kept 8-bit and decoded per tap through the host's tables (large ones):
    $PTAMD_TEX_NATIVE forces the form; both must be the oracle's image bit for bit (base colour, roughness / metallic, normal map, cut-out
    alpha in the trace kernels' alpha test, emission texture, float environment)."""
    if "PTAMD_TEX_NATIVE" in os.environ:
        pytest.skip("$PTAMD_TEX_NATIVE is preset for this session")
    monkeypatch.setenv("PTAMD_TEX_NATIVE", native)
    for sc in (scenes.textured_scene(), scenes.random_scene(1005, extras=True)):
        p = _start(gpu_renderer, sc, 160, 90, 3, 6)
        gpu_renderer.render(0)
        assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), oracle_lib.OracleScene(sc, p).render(0, 3))
    monkeypatch.delenv("PTAMD_TEX_NATIVE")


def test_textured_scene_accumulator_and_golden(gpu_renderer):
    """The N3 golden fixture (tests/golden/n3_textured_golden.npz, minted from the oracle) through the product path."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "n3_textured_golden.npz"))
    sc = scenes.textured_scene()
    _start(gpu_renderer, sc, 96, 54, 2, 6)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    assert acc.tobytes() == g["acc2"].tobytes()
    _, hits = gpu_renderer.debugSample(0)
    assert np.array_equal(hits.astype(np.int16), g["hits0"])
    # multi-batch accumulation (samples_in_flight 1) reproduces the same bits
    _start(gpu_renderer, sc, 96, 54, 2, 6, samples_in_flight=1)
    gpu_renderer.render(0)
    assert gpu_renderer.readbackAccumulator().tobytes() == g["acc2"].tobytes()


def test_host_supplied_alias_table_is_used_verbatim(gpu_renderer):
    sc = scenes.textured_scene(area_light=False)
    p = make_params(64, 36, 1, 4)
    o = oracle_lib.OracleScene(sc, p)
    al = o.envAlias().copy()
    al["p"] = 1.0            # a (wrong but legal) table: never redirect => uniform texel sampling with the luma pdf
    sc.env_alias = al
    _start(gpu_renderer, sc, 64, 36, 1, 4)
    assert gpu_renderer.envAlias().tobytes() == al.tobytes()
    o2 = oracle_lib.OracleScene(sc, p)
    rg, hg = gpu_renderer.debugSample(0)
    rc, hc = o2.debug_sample(0)
    assert np.array_equal(hg, hc) and rg.tobytes() == rc.tobytes()
    r0, _ = o.debug_sample(0)
    assert r0.tobytes() != rc.tobytes()  # and it matters


def test_texture_argument_validation(gpu_renderer):
    sc = scenes.textured_scene(env=False)
    sc.nodes[0].materials[0].base_texture = 99
    with pytest.raises(abi.PtamdError, match="texture id out of range"):
        _start(gpu_renderer, sc, 32, 18, 1, 2)
    sc = scenes.textured_scene(env=False)
    sc.env_texture = 42
    with pytest.raises(abi.PtamdError, match="env_texture out of range"):
        _start(gpu_renderer, sc, 32, 18, 1, 2)


def test_textured_scene_large_image_properties(gpu_renderer):
    """1920x1080, 8 spp, textures + cut-outs + environment: finite accumulator, alpha = 1, and the accumulator of a
    two-way sample split merges to the same running mean within fp32 reassociation."""
    sc = scenes.textured_scene()
    _start(gpu_renderer, sc, 1920, 1080, 8, 6, nonfinite_policy=abi.NONFINITE_ZERO)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    assert np.isfinite(acc).all() and np.all(acc[..., 3] == 1.0)
    st = gpu_renderer.stats()
    assert st.paths == 1920 * 1080 * 8 and st.shadow_rays > 0
    halves = []
    for first in (0, 4):
        _start(gpu_renderer, sc, 1920, 1080, 4, 6, first_sample=first, nonfinite_policy=abi.NONFINITE_ZERO)
        gpu_renderer.render(0)
        halves.append(gpu_renderer.readbackAccumulator())
    merged = 0.5 * (halves[0][..., :3].astype(np.float64) + halves[1][..., :3])
    np.testing.assert_allclose(acc[..., :3], merged, rtol=2e-5, atol=1e-6)


# ---- ingested scenes (scene.json fixture, glTF) through the HIP path -----------------------------------
def test_ingested_scene_json_fixture_parity(gpu_renderer):
    """The committed scene.json + _data.bin fixture, read by libptamd's own loader (pt_scene_load_json), rendered on the GPU and
    by the oracle from the same snapshot: textures, cut-out floor, thin anisotropic glass, emissive quad, stored alias table."""
    import os
    from platinum_amd import scene_io
    sc = scene_io.SceneFile.load(os.path.join(os.path.dirname(__file__), "golden", "scene_fixture", "mini.json"))
    w, h = 160, 90
    p = _start(gpu_renderer, sc, w, h, 3, 6)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    assert gpu_renderer.envAlias().tobytes() == o.envAlias().tobytes()
    for s in (0, 2):
        assert gpu_renderer.tracePrimary(s).tobytes() == o.trace_primary(s).tobytes()
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and rg.tobytes() == rc.tobytes()
    gpu_renderer.render(0)
    assert gpu_renderer.readbackAccumulator().tobytes() == o.render(0, 3).tobytes()


@pytest.mark.parametrize("which", ["field", "fuzz", "textured", "degenerate", "c2"])
def test_two_level_structure_gives_the_same_hits_and_radiance(gpu_renderer, which):
    """PT_ACCEL_TWO_LEVEL (TLAS over instances + one object-space BLAS per mesh, what renderer_pt.cpp:653-749 builds): the
    triangle test stays the world-space one, the object-space slab tests carry an error-bound slack, so closest hits and
    radiance are the oracle's bit for bit — instanced field (rotation-free), fuzz scenes (mirrored / non-uniform / rotated
    instances, thin lens), cut-outs + environment, one-triangle meshes and coincident instances."""
    cases = {"field": [scenes.field_scene(8)], "fuzz": [scenes.random_scene(s) for s in (3, 7, 11, 19)], "textured": [scenes.textured_scene()],
             "degenerate": [], "c2": [scenes.cornell_sphere_scene()]}[which]
    if which == "degenerate":
        quad = [[[-2, 0, -2], [-2, 0, 2], [2, 0, -2]], [[2, 0, -2], [-2, 0, 2], [2, 0, 2]]]
        cases = [_tiny_scene(quad[:1]), _tiny_scene(quad), _tiny_scene(quad, extra_instances=2), _tiny_scene(quad * 40)]
    for sc in cases:
        w, h, bounces = 128, 72, 6
        gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
        gpu_renderer.startRender(sc, (w, h), 2, max_bounces=bounces, accel_structure=abi.ACCEL_TWO_LEVEL)
        st = gpu_renderer.stats()
        assert st.accel_two_level == 1
        o = oracle_lib.OracleScene(sc, make_params(w, h, 2, bounces))
        assert gpu_renderer.tracePrimary(0).tobytes() == o.trace_primary(0).tobytes()
        for s_ in (0, 1):
            rg, hg = gpu_renderer.debugSample(s_)
            rc, hc = o.debug_sample(s_)
            assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
        gpu_renderer.render(0)
        assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))
    if which == "field":   # the structure is the small one: 64 instances + two meshes, not 64 k triangles' worth of nodes
        assert gpu_renderer.stats().bvh_nodes < 700


@pytest.mark.parametrize("which", ["field", "fuzz", "textured"])
def test_four_wide_fallback_form_gives_the_same_hits_and_radiance(gpu_renderer, which, monkeypatch):
    """r03: the device build emits 6-wide nodes (BvhNode6) by default — every other test in this file runs on them.  $PTAMD_BVH4 keeps the
    4-wide form (what the fallback builders — radix tree, per-pass launches, trees deeper than the stack at five pushes per level — and the
    two-level structure use); it must give the oracle's answers too, from a tree with more nodes."""
    skip_if_structure_env_preset()
    cases = {"field": [scenes.field_scene(8)], "fuzz": [scenes.random_scene(s) for s in (3, 7, 11, 19)], "textured": [scenes.textured_scene()]}[which]
    for sc in cases:
        w, h, bounces = 128, 72, 6
        gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
        gpu_renderer.startRender(sc, (w, h), 2, max_bounces=bounces)
        nodes6 = gpu_renderer.stats().bvh_nodes
        monkeypatch.setenv("PTAMD_BVH4", "1")
        gpu_renderer.startRender(sc, (w, h), 2, max_bounces=bounces)
        monkeypatch.delenv("PTAMD_BVH4")
        assert gpu_renderer.stats().bvh_nodes > nodes6
        o = oracle_lib.OracleScene(sc, make_params(w, h, 2, bounces))
        assert gpu_renderer.tracePrimary(0).tobytes() == o.trace_primary(0).tobytes()
        rg, hg = gpu_renderer.debugSample(0)
        rc, hc = o.debug_sample(0)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
        gpu_renderer.render(0)
        assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))


def test_six_wide_tree_deeper_than_the_stack_is_rebuilt_four_wide(gpu_renderer, monkeypatch):
    """A tree deeper than the traversal stack holds at five pushes per level must come back in the 4-wide form (three per level) from the
    same build call.  No test scene is that deep, so $PTAMD_TEST_W6_LEVELS lowers the 6-wide limit: the retry path runs for real."""
    skip_if_structure_env_preset()
    sc = scenes.field_scene(8)
    w, h, bounces = 96, 54, 5
    gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
    monkeypatch.setenv("PTAMD_BVH4", "1")
    gpu_renderer.startRender(sc, (w, h), 1, max_bounces=bounces)
    nodes4 = gpu_renderer.stats().bvh_nodes
    monkeypatch.delenv("PTAMD_BVH4")
    monkeypatch.setenv("PTAMD_TEST_W6_LEVELS", "3")
    gpu_renderer.startRender(sc, (w, h), 1, max_bounces=bounces)
    monkeypatch.delenv("PTAMD_TEST_W6_LEVELS")
    assert gpu_renderer.stats().bvh_nodes == nodes4          # the 4-wide tree, not a truncated 6-wide one
    o = oracle_lib.OracleScene(sc, make_params(w, h, 1, bounces))
    rg, hg = gpu_renderer.debugSample(0)
    rc, hc = o.debug_sample(0)
    assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)


def _same_bits_or_both_nan(a, b):
    """Bitwise equality, except that NaNs only have to coincide: x86 and gfx950 produce default NaNs of opposite sign."""
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])


def test_ingested_gltf_parity(gpu_renderer, tmp_path):
    """(The glTF has one triangle without normals — zero-length shading normal — so a few paths are NaN on both sides.)"""
    from platinum_amd import scene_io
    import test_scene_ingestion as tsi
    path, _ = tsi.build_gltf(tmp_path, "glb")
    sc = scene_io.SceneFile.empty().import_gltf(path)
    sc.set_environment(scenes.sky_environment(32, 16))
    p = _start(gpu_renderer, sc, 128, 80, 2, 6)
    o = oracle_lib.OracleScene(sc, p)
    for s in (0, 1):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    assert (hc[0, ..., 0] >= 0).mean() > 0.05 and np.nanmean(rc[..., :3]) > 1e-3 and np.isnan(rc).mean() < 0.01


# ---- BASELINE configs[4] (C5) stand-in: the procedural atrium ----------------------------------------------------------
def test_atrium_c5_small_parity(gpu_renderer):
    sc = scenes.atrium_scene(env_size=(256, 128), columns=6)
    p = _start(gpu_renderer, sc, 192, 108, 2, 12)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    assert gpu_renderer.envAlias().tobytes() == o.envAlias().tobytes()
    for s in (0, 1):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))


def test_atrium_c5_through_gltf_and_exr_ingestion_parity(gpu_renderer, tmp_path):
    """the C5-class scene written as FILES (tools/export_gltf.py: .glb with JPEG + PNG textures, ZIP OpenEXR
    environment), read by the product's loaders (pt_scene_import_gltf: JPEG decode, MikkTSpace tangents, KHR material
    extensions; pt_scene_load_environment) and rendered on the GPU and by the oracle from that same imported snapshot."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import export_gltf
    sc = export_gltf.atrium_through_ingestion(str(tmp_path), env_size=(256, 128), columns=6, jpeg=True)
    c = sc.counts()
    assert c.triangles == scenes.atrium_scene(env_size=(8, 4), columns=6).triangle_count and c.textures == 5 and c.cameras == 1
    p = _start(gpu_renderer, sc, 192, 108, 2, 12)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    assert gpu_renderer.envAlias().tobytes() == o.envAlias().tobytes()
    for s in (0, 1):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    assert _same_bits_or_both_nan(acc, o.render(0, 2))
    # the files describe the scene the procedural snapshot describes (JPEG quantisation and regenerated tangents aside)
    ref = oracle_lib.OracleScene(scenes.atrium_scene(env_size=(256, 128), columns=6), p).render(0, 2)
    assert abs(np.nanmean(acc[..., :3]) - np.nanmean(ref[..., :3])) < 0.05 * np.nanmean(ref[..., :3])


def test_atrium_c5_full_size_properties(gpu_renderer):
    """3840x2160, 12 bounces, 4096x2048 environment (8.4 M alias entries): size-independent properties."""
    sc = scenes.atrium_scene()
    assert sc.triangle_count > 250_000
    _start(gpu_renderer, sc, 3840, 2160, 2, 12, nonfinite_policy=abi.NONFINITE_ZERO)
    gpu_renderer.render(0)
    a = gpu_renderer.readbackAccumulator()
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and (a[..., :3] >= 0).all()
    st = gpu_renderer.stats()
    assert st.paths == 3840 * 2160 * 2 and st.shadow_rays > st.paths and st.nonfinite_samples < 100
    al = gpu_renderer.envAlias()
    # (the mean of pdf is 1.03, not 1: the reference sums the 8.4 M importances in fp32, core/environment.cpp:24-31 — kept)
    assert len(al) == 4096 * 2048 and abs(float(al["pdf"].astype(np.float64).mean()) - 1.0) < 0.05
    # halves merge to the whole
    parts = []
    for first in (0, 1):
        _start(gpu_renderer, sc, 3840, 2160, 1, 12, first_sample=first, nonfinite_policy=abi.NONFINITE_ZERO)
        gpu_renderer.render(0)
        parts.append(gpu_renderer.readbackAccumulator()[..., :3].astype(np.float64))
    np.testing.assert_allclose(a[..., :3], 0.5 * (parts[0] + parts[1]), rtol=2e-5, atol=1e-6)


def test_axis_parallel_rays_do_not_walk_the_whole_tree(gpu_renderer):
    """Regression: a direction component of exactly 0 used to give inv = inf and NaN slabs (constraints dropped), so rays
    towards the top row of a lat-long environment — direction exactly (0, 1, 0) — visited thousands of nodes each."""
    sc = scenes.atrium_scene(env_size=(64, 32), columns=6)
    env = sc.textures[sc.env_texture].pixels
    env[...] = 1e-4
    env[0, :, :3] = 1000.0   # all the importance in the top row: every environment NEE ray points straight up
    env[..., 3] = 1.0
    p = _start(gpu_renderer, sc, 160, 90, 1, 4)
    gpu_renderer.measureTraversal(0)
    st = gpu_renderer.stats()
    assert 1.0 < st.nodes_per_shadow_ray < 60, st.nodes_per_shadow_ray   # (thousands before the fix)
    o = oracle_lib.OracleScene(sc, p)
    rg, hg = gpu_renderer.debugSample(0)
    rc, hc = o.debug_sample(0)
    assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)


# ---- degenerate inputs (the reference's tests have none; these pin OUR defined behaviour and guard against GPU faults) ----
def _tiny_scene(tris, env=True, extra_instances=0):
    """tris: list of 3x3 vertex arrays (one mesh, one material)."""
    sc = scenes.Scene(name="tiny")
    if tris:
        v = np.concatenate([np.asarray(t, np.float32) for t in tris])
        n = np.tile(np.array([[0, 1, 0]], np.float32), (len(v), 1))
        tg = np.tile(np.array([[1, 0, 0, 1]], np.float32), (len(v), 1))
        uv = np.zeros((len(v), 2), np.float32)
        m = sc.add_mesh(scenes._make_mesh(v, n, tg, uv, np.arange(len(v)), np.zeros(len(tris))))
        for k in range(1 + extra_instances):
            sc.add_instance(m, scenes.Transform(translation=(0, 0.0, 0)), [scenes.Material(base_color=(0.6, 0.5, 0.4, 1.0))])
    if env:
        sc.env_texture = sc.add_texture(scenes.sky_environment(16, 8), abi.TEX_RGBA32F)
    sc.set_camera(scenes.Camera.with_focal_length(28.0), scenes.Transform(translation=(0, 2, 5), target=(0, 0, 0), track=True))
    return sc


@pytest.mark.parametrize("case", ["empty", "one_triangle", "two_triangles", "zero_area", "coincident_copies", "three_instances_same_place"])
def test_degenerate_scenes_match_oracle(gpu_renderer, case):
    quad = [[[-2, 0, -2], [-2, 0, 2], [2, 0, -2]], [[2, 0, -2], [-2, 0, 2], [2, 0, 2]]]
    if case == "empty":
        sc = _tiny_scene([])
    elif case == "one_triangle":
        sc = _tiny_scene(quad[:1])
    elif case == "two_triangles":
        sc = _tiny_scene(quad)
    elif case == "zero_area":
        sc = _tiny_scene(quad + [[[0, 1, 0], [0, 1, 0], [0, 1, 0]], [[1, 1, 1], [2, 2, 2], [3, 3, 3]]])  # a point and a line
    elif case == "coincident_copies":
        sc = _tiny_scene(quad * 40)               # 80 triangles, 40 exact copies of each: the tie-break decides
    else:
        sc = _tiny_scene(quad, extra_instances=2)  # three instances of the same mesh at the same place
    p = _start(gpu_renderer, sc, 64, 36, 2, 4)
    o = oracle_lib.OracleScene(sc, p)
    g, c = gpu_renderer.tracePrimary(0), o.trace_primary(0)
    assert g.tobytes() == c.tobytes()
    if case in ("coincident_copies", "three_instances_same_place", "two_triangles"):
        hit = g["instance"] >= 0
        assert hit.any() and np.all(g["instance"][hit] == 0) and np.all(g["primitive"][hit] <= 1)  # lowest (instance, primitive) wins ties
    for s in (0, 1):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))


def test_render_scene_tool_end_to_end(tmp_path):
    """tools/render_scene.py: scene.json -> ingestion -> GMoN render -> fused post-process -> PNG, in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "mini.png"
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "render_scene.py"), os.path.join(root, "tests", "golden", "scene_fixture", "mini.json"),
                        str(out), "--size", "96", "54", "--spp", "16", "--gmon", "4"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    data = out.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 1000 and "Msamples/s" in p.stdout


def test_radix_tree_fallback_builder_gives_the_same_image(gpu_renderer, monkeypatch):
    """The Karras radix tree (fallback of the PLOC builder, PTAMD_RADIX_TREE=1) and the PLOC tree answer every ray alike —
    the intersection contract does not depend on the acceleration structure."""
    skip_if_structure_env_preset()
    sc = scenes.field_scene(8)
    p = _start(gpu_renderer, sc, 160, 90, 2, 6)
    nodes_ploc = gpu_renderer.stats().bvh_nodes
    gpu_renderer.render(0)
    a = gpu_renderer.readbackAccumulator()
    monkeypatch.setenv("PTAMD_RADIX_TREE", "1")
    _start(gpu_renderer, sc, 160, 90, 2, 6)
    nodes_radix = gpu_renderer.stats().bvh_nodes
    gpu_renderer.render(0)
    b = gpu_renderer.readbackAccumulator()
    monkeypatch.delenv("PTAMD_RADIX_TREE")
    assert a.tobytes() == b.tobytes() and nodes_ploc != nodes_radix
    assert a.tobytes() == oracle_lib.OracleScene(sc, p).render(0, 2).tobytes()


def test_leaf_slots_hold_two_triangles_and_the_one_triangle_form_gives_the_same_image(gpu_renderer, monkeypatch):
    """r4: a 64-byte leaf slot holds two consecutive triangles of a mesh that share an edge (pt_device.h TriRec, host_scene.h pair_mesh_triangles):
    the field's spheres and the Cornell shell pair up completely (slots = triangles / 2), $PTAMD_NO_PAIRS keeps one triangle per slot, and both forms
    give the oracle's hits, per-sample radiance and image bit for bit — which triangles share a slot cannot change an answer."""
    if "PTAMD_NO_PAIRS" in os.environ:
        pytest.skip("$PTAMD_NO_PAIRS is preset for this session")
    skip_if_structure_env_preset()
    for sc in (scenes.field_scene(6), scenes.random_scene(9), scenes.textured_scene()):
        w, h, B = 120, 68, 6
        o = None
        for pairs in (True, False):
            if pairs:
                monkeypatch.delenv("PTAMD_NO_PAIRS", raising=False)
            else:
                monkeypatch.setenv("PTAMD_NO_PAIRS", "1")
            p = _start(gpu_renderer, sc, w, h, 3, B)
            st = gpu_renderer.stats()
            assert st.leaf_slots == st.triangles if not pairs else st.leaf_slots < st.triangles
            o = o or oracle_lib.OracleScene(sc, p)
            assert gpu_renderer.tracePrimary(1).tobytes() == o.trace_primary(1).tobytes()
            rg, hg = gpu_renderer.debugSample(0)
            rc, hc = o.debug_sample(0)
            assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
            gpu_renderer.render(0)
            assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 3))
    monkeypatch.delenv("PTAMD_NO_PAIRS", raising=False)


@pytest.mark.parametrize("accel", [abi.ACCEL_ONE_BVH, abi.ACCEL_TWO_LEVEL])
def test_leaf_slot_pairing_edge_cases(gpu_renderer, accel):
    """scenes.pairing_edge_cases_scene on the device builder (host_scene.h pair_mesh_triangles -> lbvh.hip k_flatten): plain quad, fan of three, a
    partner with a repeated vertex, a degenerate first triangle, duplicated triangles, triangles sharing one vertex, two materials (two shading
    classes) inside one slot, a mirrored instance — hits, per-sample radiance and image against the oracle; the pairs actually form."""
    sc = scenes.pairing_edge_cases_scene()
    gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
    gpu_renderer.startRender(sc, (128, 80), 3, max_bounces=5, accel_structure=accel)
    st = gpu_renderer.stats()
    if "PTAMD_NO_PAIRS" not in os.environ and "PTAMD_TWO_LEVEL" not in os.environ:   # (a preset switch changes the slot count, not the image)
        assert st.triangles == 30 and st.leaf_slots == (20 if accel == abi.ACCEL_ONE_BVH else 30)   # 2 instances x (15 triangles -> 10 slots: 5 pairs + 5 singles)
    p = make_params(128, 80, 3, 5)
    o = oracle_lib.OracleScene(sc, p)
    assert gpu_renderer.tracePrimary(1).tobytes() == o.trace_primary(1).tobytes()
    for s in (0, 2):
        rg, hg = gpu_renderer.debugSample(s)
        rc, hc = o.debug_sample(s)
        assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 3))


@pytest.mark.parametrize("seed,extras,two_level", [(20341, False, False), (104187, False, False), (50035, True, False), (310601, True, False), (310601, True, True)])
def test_rays_in_a_triangles_plane_miss_it_whatever_structure_is_walked(gpu_renderer, seed, extras, two_level):
    """r4, found by extending the fuzz to 22 000 new seeds: a shadow ray that leaves one half of a flat quad INSIDE the quad's plane (a light
    flush with the wall).  Against a coplanar triangle Moeller-Trumbore's determinant is rounding noise (1.4e-6 where |e1||e2| = 2.8; exact
    arithmetic says miss, fp32 said hit with u = v = 0.5), so with `det == 0` as the only degenerate case the ray counted as occluded exactly
    when that triangle got TESTED: two-triangle leaf slots always test the partner (seeds 20341, 104187, 50035: HIP path = brute force != the
    oracle's BVH).  The contract now calls |det| <= 1e-6 |e1|_1 |d x e2|_1 a miss (pt_bvh.h intersect_triangle, the oracle alike): the HIP path,
    the oracle's BVH and the oracle's brute-force traversal agree again, bit for bit.
    What no determinant rule removes (seed 310601): a ray 3e-6 off the plane of a triangle 0.1 units small, 5 units away — the determinant is
    accurate, the barycentric numerator s . p is not (its terms are 50x the result), fp32 says u = -0 where exact arithmetic says -0.14.  Every
    structure that tests that triangle reports the same false hit (one BVH, the oracle's tree, brute force); the two-level structure's tighter
    object-space boxes do not reach it.  One pixel-sample in 1 500 two-level scenes; documented, not hidden: the assertion below allows it."""
    sc = scenes.random_scene(seed, extras=extras)
    w, h, B, spp, first, flags, space = 96, 54, 3 + seed % 7, 2 + seed % 2, 0, abi.FLAG_MULTISCATTER_GGX, scenes.BT2020
    if seed % 3 == 0:
        w, h = 71, 45
    if extras:
        flags = abi.FLAG_MULTISCATTER_GGX if seed % 5 else 0
        space = scenes.BT2020 if seed % 7 else scenes.BT709
        first, spp = (seed % 13) * 3, 2 + seed % 4
    if seed % 6 == 5:
        spp, (w, h) = 33 + seed % 41, ((40, 27) if seed % 3 else (33, 18))
    integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
    gpu_renderer.selectKernel(integ)
    gpu_renderer.startRender(sc, (w, h), spp, workingSpace=space, flags=flags, max_bounces=B, first_sample=first,
                             accel_structure=abi.ACCEL_TWO_LEVEL if two_level else abi.ACCEL_ONE_BVH)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
    p = make_params(w, h, spp, B, flags=flags, integrator=integ, working_space=space, first_sample=first)
    tree = oracle_lib.OracleScene(sc, p, use_bvh=True).render(first, spp)
    brute = oracle_lib.OracleScene(sc, p, use_bvh=False).render(first, spp)
    assert _same_bits_or_both_nan(tree, brute)
    if two_level and seed == 310601:   # THE named case, not a tolerance of the two-level path: every other two-level test demands the same bits
        differing = np.argwhere((acc.view(np.uint32) != tree.view(np.uint32)).any(-1))
        assert len(differing) <= 1 and np.nanmax(np.abs(acc - tree)) < 1e-5 and np.array_equal(np.isnan(acc), np.isnan(tree))
    else:
        assert _same_bits_or_both_nan(acc, tree)


def test_where_the_oracles_tree_and_brute_force_part_the_product_follows_brute_force(gpu_renderer):
    """r5, found by 60 000 new fuzz seeds after the cull margin was widened (seed 800642, extras): ONE shadow ray whose any-hit answer differs between
    the oracle's own BVH and the oracle's brute-force loop — the contract's definition — because fp32 Moeller-Trumbore accepts a triangle the ray
    passes outside of (a false hit, DESIGN.md section 2 case 2): brute force tests every triangle, the oracle's tight per-triangle boxes never reach
    that one, the product's pair-slot boxes do.  The product equals BRUTE FORCE bit for bit; the oracle's tree differs from both in one pixel-sample
    by 3e-7.  Kept as what it is: evidence that "equal to the oracle" means "equal to the contract", and that the oracle's tree is only a fast way
    to evaluate it that is right all but once in ~1e9 rays."""
    seed = 800642
    sc = scenes.random_scene(seed, extras=True)
    w, h, B = 96, 54, 3 + seed % 7
    flags = abi.FLAG_MULTISCATTER_GGX if seed % 5 else 0
    space = scenes.BT2020 if seed % 7 else scenes.BT709
    first, spp, sif = (seed % 13) * 3, 2 + seed % 4, 1 + seed % 5
    integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
    gpu_renderer.selectKernel(integ)
    gpu_renderer.startRender(sc, (w, h), spp, workingSpace=space, flags=flags, max_bounces=B, first_sample=first, samples_in_flight=sif)
    gpu_renderer.render(0)
    acc = gpu_renderer.readbackAccumulator()
    gpu_renderer.selectKernel(abi.INTEGRATOR_MIS)
    p = make_params(w, h, spp, B, flags=flags, integrator=integ, working_space=space, first_sample=first)
    brute = oracle_lib.OracleScene(sc, p, use_bvh=False).render(first, spp)
    tree = oracle_lib.OracleScene(sc, p, use_bvh=True).render(first, spp)
    differing = np.argwhere((tree.view(np.uint32) != brute.view(np.uint32)).any(-1))
    assert len(differing) <= 1 and np.nanmax(np.abs(tree - brute)) < 1e-5
    # the default structure (pair slots: a slot's box spans two triangles and reaches the false hit) follows brute force; with one triangle per slot
    # ($PTAMD_NO_PAIRS: boxes as tight as the oracle's) the product follows the oracle's tree instead — an any-hit answer depends on whether a
    # structure's boxes reach a triangle fp32 wrongly accepts, and this seed shows both sides of it.  Exactly ONE side is expected per structure
    # (ADVICE r5); the session-wide structure presets of the sweeps (4-wide, radix tree, two-level ...) have boxes of their own: not claimed here.
    skip_if_structure_env_preset()
    if "PTAMD_NO_PAIRS" in os.environ:
        assert _same_bits_or_both_nan(acc, tree)
    else:
        assert _same_bits_or_both_nan(acc, brute)


@pytest.mark.parametrize("seed", list(range(24)) + [1000 + i for i in range(12)])
def test_random_scene_fuzz_parity(gpu_renderer, seed):
    """36 seeded random scenes (scenes.random_scene) through the HIP path against the oracle: hit ids, per-sample radiance and
    the accumulated image, bit for bit (NaNs — the reference's BSDF produces a few — only have to coincide).  Seeds >= 1000 draw the
    `extras` too (every texture slot, anisotropy rotation, degenerate material corners) and switch the multiscatter flag / working space.
    (tests/fuzz_parity_sweep.py is the same check over thousands of seeds: r03 ran 3 000 plain + 3 000 with extras, all identical.)"""
    extras = seed >= 1000
    sc = scenes.random_scene(seed, extras=extras)
    integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
    kw = dict(flags=abi.FLAG_MULTISCATTER_GGX if seed % 5 else 0, working_space=scenes.BT2020 if seed % 3 else scenes.BT709) if extras else {}
    p = _start(gpu_renderer, sc, 96, 54, 2, 6, integrator=integ, **kw)
    o = oracle_lib.OracleScene(sc, p)
    assert bytes(gpu_renderer.constants()) == bytes(o.constants())
    assert gpu_renderer.tracePrimary(1).tobytes() == o.trace_primary(1).tobytes()
    rg, hg = gpu_renderer.debugSample(0)
    rc, hc = o.debug_sample(0)
    assert np.array_equal(hg, hc) and _same_bits_or_both_nan(rg, rc)
    gpu_renderer.render(0)
    assert _same_bits_or_both_nan(gpu_renderer.readbackAccumulator(), o.render(0, 2))
