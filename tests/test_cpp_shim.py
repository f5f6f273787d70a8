"""The C++ host side above the C ABI: include/ptamd_renderer.hpp carries the public members of the reference's
pt::renderer_pt::Renderer (renderer_pt.hpp:14-73); tests/cpp/shim_render.cpp drives it the way the reference's frontend drives
that class (startRender, one render() per frame while status() says busy, renderProgress(), readback / present)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "shim_render.cpp")
HDRS = [os.path.join(ROOT, "include", h) for h in ("ptamd_renderer.hpp", "ptamd_postprocess.hpp", "ptamd.h", "ptamd_scene.h")]
WALK_SRC = os.path.join(ROOT, "tests", "cpp", "options_walk.cpp")
WALK_EXE = os.path.join(ROOT, "tests", "_build", "options_walk")
EXE = os.path.join(ROOT, "tests", "_build", "shim_render")
LIBDIR = os.path.join(ROOT, "platinum_amd", "csrc")
FIXTURE = os.path.join(ROOT, "tests", "golden", "scene_fixture", "mini.json")
LUT = os.path.join(ROOT, "platinum_amd", "data", "ggx_luts.bin")


def _build(src, exe):
    lib = os.path.join(LIBDIR, "libptamd.so")
    if not os.path.exists(lib):
        pytest.fail("libptamd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    newest = max(os.path.getmtime(p) for p in [src, lib] + HDRS)
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L" + LIBDIR, "-lptamd", "-ldl", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def build_shim():
    _build(WALK_SRC, WALK_EXE)
    return _build(SRC, EXE)


def run_shim(args, **kw):
    env = dict(os.environ, PTAMD_LUT_PATH=LUT)
    return subprocess.run([build_shim()] + [str(a) for a in args], env=env, capture_output=True, text=True, **kw)


def test_header_is_clean_cxx17_and_needs_nothing_but_the_c_abi(tmp_path):
    """-pedantic -Werror, and the header alone (no torch, no HIP headers, no reference headers) is enough to use the class."""
    tu = tmp_path / "only_header.cpp"
    tu.write_text('#include "ptamd_renderer.hpp"\nint main() { ptamd::renderer_pt::Renderer* r = nullptr; (void)r; return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(tu)])
    text = open(HDRS[0]).read()
    for member in ("void render()", "void startRender(", "selectedKernel()", "selectKernel(uint32_t", "presentRenderTarget()", "readbackRenderTarget(uint2*",
                   "int status()", "renderProgress()", "renderTime()", "std::vector<postprocess::PostProcessPass::Options> postProcessOptions()",
                   "postprocess::Tonemap::Options* tonemapOptions()", "gmonOptions()", "outputColorspace()",
                   "Status_Blocked = 0", "Status_Ready = 1 << 0", "Status_Busy = 1 << 2", "Status_Done = 1 << 3", "enum class Integrators { Simple = 0, MIS }"):
        assert member in text, member          # the reference's names (renderer_pt.hpp:14-73)


def test_cpp_host_without_a_gpu_reports_and_stays_blocked():
    """Error behaviour of the reference's class: nothing throws; the failure is printed and the object is Status_Blocked."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device path cannot be shown")
    r = run_shim([FIXTURE, 64, 36, 2, 4, "/tmp/ptamd_shim_nogpu"])
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "renderer_pt: pt_create: no HIP device available" in r.stderr
    assert run_shim([]).returncode == 2


@pytest.mark.gpu
def test_cpp_host_device_group_constructor(tmp_path):
    """Renderer(std::vector<int>{0, 0}): the C++ face of the device group (two logical shards on one GPU) gives the single-device
    image up to the order of the final sum."""
    w, h, spp, bounces = 96, 54, 8, 5
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r1 = run_shim([FIXTURE, w, h, spp, bounces, one], timeout=300)
    r2 = run_shim([FIXTURE, w, h, spp, bounces, two, 0, "0,0"], timeout=300)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
    a, b = np.fromfile(one + ".acc", np.float32), np.fromfile(two + ".acc", np.float32)
    np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-7)
    assert np.abs(np.fromfile(one + ".rgba", np.uint8).astype(int) - np.fromfile(two + ".rgba", np.uint8).astype(int)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("buckets", [0, 3])
def test_cpp_host_renders_the_fixture_like_the_python_host_and_the_oracle(tmp_path, buckets):
    """One render() per frame from C++ gives the accumulator, the RGBA8 readback and the presented device image that the Python host
    gets from one batched render of the same scene file — and the accumulator the oracle computes, bit for bit."""
    import oracle_lib
    from platinum_amd import Renderer, abi, scene_io
    from platinum_amd.renderer import make_params
    w, h, spp, bounces = 96, 54, 6, 5
    out = str(tmp_path / "shim")
    r = run_shim([FIXTURE, w, h, spp, bounces, out] + ([buckets] if buckets else []), timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert f"{spp} spp in {spp} render() calls" in r.stdout
    acc = np.fromfile(out + ".acc", np.float32).reshape(h, w, 4)
    rgba = np.fromfile(out + ".rgba", np.uint8).reshape(h, w, 4)
    presented = np.fromfile(out + ".present", np.uint8).reshape(h, w, 4)
    assert np.array_equal(rgba, presented) and rgba[..., 3].min() == 255

    sc = scene_io.SceneFile.load(FIXTURE)
    flags = abi.FLAG_MULTISCATTER_GGX | (abi.FLAG_GMON if buckets else 0)
    py = Renderer(device=0)
    py.startRender(sc, (w, h), spp, gmonBuckets=buckets, flags=flags, max_bounces=bounces)
    py.render(0)
    assert py.readbackAccumulator().tobytes() == acc.tobytes()
    assert np.array_equal(py.readbackRenderTarget(), rgba)
    o = oracle_lib.OracleScene(sc, make_params(w, h, spp, bounces, flags=flags, gmon_buckets=buckets) if buckets else make_params(w, h, spp, bounces, flags=flags))
    want = o.render_gmon(spp)[1] if buckets else o.render(0, spp)
    assert want.tobytes() == acc.tobytes()


def test_option_types_carry_the_references_shapes():
    """ptamd_postprocess.hpp: the names the untouched UI code dereferences (core/postprocessing.hpp:29-252; pt_viewport.cpp:259-535)."""
    text = open(os.path.join(ROOT, "include", "ptamd_postprocess.hpp")).read()
    for name in ("struct Look { float3 offset, slope, power; float saturation; }", "namespace looks", "golden", "punchy", "namespace khronos_pbr",
                 "compressionStart", "desaturation", "namespace presets", "silver", "enum class Tonemapper { None, AgX, KhronosPBR, flim }",
                 "struct ExposureOptions", "struct ToneCurveOptions", "struct VignetteOptions", "struct ChromaticAberrationOptions",
                 "struct ContrastSaturationOptions", "struct LiftGammaGain", "agx::Options agxOptions", "khronos_pbr::Options khrOptions",
                 "flim::Options flimOptions = flim::presets::flim", "LiftGammaGain postTonemap",
                 "enum class Type { Exposure, ToneCurve, Vignette, ChromaticAberration, ContrastSaturation, Tonemap }",
                 "ExposureOptions* exposure = nullptr", "ToneCurveOptions* toneCurve", "VignetteOptions* vignette",
                 "ChromaticAberrationOptions* chromaticAberration", "ContrastSaturationOptions* contrastSaturation", "TonemapOptions* tonemap"):
        assert name in text, name


@pytest.mark.gpu
def test_the_uis_walk_over_the_option_structs_changes_the_image_like_the_oracle(tmp_path):
    """tests/cpp/options_walk.cpp edits the options the way RenderViewport does — the switch over postProcessOptions()' tagged pointers,
    tonemapOptions()->agxOptions.look = looks::golden, flimOptions = presets::silver, postTonemap's colours — in four frames.  Each frame's
    RGBA8 readback must equal, byte for byte, what the oracle's post-process makes of the same accumulator with the same values put
    into the flat C structs (and what the Python host reads back)."""
    import oracle_lib
    from platinum_amd import Renderer, abi, scene_io
    from platinum_amd.renderer import make_params
    build_shim()
    w, h, spp, bounces = 96, 54, 6, 5
    out = str(tmp_path / "walk")
    r = subprocess.run([WALK_EXE, FIXTURE, str(w), str(h), str(spp), str(bounces), out], env=dict(os.environ, PTAMD_LUT_PATH=LUT),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)

    sc = scene_io.SceneFile.load(FIXTURE)
    py = Renderer(device=0)
    py.startRender(sc, (w, h), spp, flags=abi.FLAG_MULTISCATTER_GGX, max_bounces=bounces)
    py.render(0)
    acc = py.readbackAccumulator()
    o = oracle_lib.OracleScene(sc, make_params(w, h, spp, bounces, flags=abi.FLAG_MULTISCATTER_GGX))
    lib = abi.load_library()
    tm = abi.TonemapOptions()
    lib.pt_default_tonemap_options(tm)          # the tonemap struct persists across the UI's frames, like the C++ object's
    post_scripts = [
        [0.5, 10.0, 60.0, 12.0, -8.0, 5.0, -10.0, 15.0, -5.0, 0.8, 10.0, 40.0, 25.0, 90.0],
        [-0.4, 0.0, 70.0, -15.0, 20.0, 0.0, 0.0, 0.0, 0.0, -1.0, -20.0, 60.0, 30.0, 70.0],
        [0.0, 25.0, 50.0, 0.0, 0.0, 10.0, 10.0, -10.0, -10.0, 0.0, 0.0, 50.0, 20.0, 100.0],
        [1.0, -30.0, 80.0, 30.0, -30.0, -20.0, 20.0, 20.0, -20.0, 1.5, 30.0, 20.0, 10.0, 50.0],
    ]
    post_fields = ["exposure", "ca_amount", "ca_green_shift", "contrast", "saturation", "blacks", "shadows", "highlights", "whites",
                   "vig_amount", "vig_midpoint", "vig_feather", "vig_power", "vig_roundness"]   # the UI's pass order: exposure, CA, contrast/saturation, curve, vignette

    def grade(vals):
        tm.shadow_color[:] = vals[0:3]; tm.midtone_color[:] = vals[3:6]; tm.highlight_color[:] = vals[6:9]
        tm.shadow_offset, tm.midtone_offset, tm.highlight_offset = vals[9:12]

    def flim(preset):
        d = abi.TonemapOptions()
        lib.pt_default_tonemap_options(d)
        for name, _ in abi.TonemapOptions._fields_:
            if name.startswith("flim_"):
                setattr(tm, name, getattr(d, name))
        if preset == "silver":                   # postprocessing.hpp:135-163
            tm.flim_pre_exposure = 3.9; tm.flim_pre_formation_filter[:] = [0.0, 0.5, 1.0]; tm.flim_pre_formation_filter_strength = 0.05
            tm.flim_extended_gamut_mul[:] = [1.0, 1.0, 1.06]; tm.flim_negative_exposure = 4.7; tm.flim_negative_density = 7.0
            tm.flim_print_backlight[:] = [0.9992, 0.99, 1.0]; tm.flim_print_exposure = 4.7; tm.flim_print_density = 30.0
            tm.flim_black_point = 0.5; tm.flim_auto_black_point = 0; tm.flim_post_formation_filter[:] = [1.0, 1.0, 0.0]
            tm.flim_post_formation_filter_strength = 0.04; tm.flim_midtone_saturation = 1.0

    images = []
    for f in range(4):
        post = abi.PostOptions()
        for name, v in zip(post_fields, post_scripts[f]):
            setattr(post, name, v)
        if f == 0:
            tm.tonemapper = abi.TONEMAP_AGX       # looks::golden (postprocessing.hpp:45-50)
            tm.agx_offset[:] = [0, 0, 0]; tm.agx_slope[:] = [1.0, 0.9, 0.5]; tm.agx_power[:] = [0.8, 0.8, 0.8]; tm.agx_saturation = 0.8
            grade([0.52, 0.5, 0.48, 0.5, 0.51, 0.5, 0.49, 0.5, 0.53, 3.0, -2.0, 4.0])
        elif f == 1:
            tm.tonemapper = abi.TONEMAP_KHRONOS_PBR
            tm.khr_compression_start, tm.khr_desaturation = 0.7, 0.2
            grade([0.5] * 9 + [0.0, 0.0, 0.0])
        elif f == 2:
            tm.tonemapper = abi.TONEMAP_FLIM
            flim("silver")
            grade([0.5] * 9 + [0.0, 0.0, 0.0])
        else:
            tm.tonemapper = abi.TONEMAP_FLIM
            flim("default")
            tm.flim_pre_exposure = 4.0; tm.flim_sigmoid_toe[:] = [0.4, 0.3]; tm.flim_pre_formation_filter[:] = [0.9, 0.8, 0.7]
            tm.flim_pre_formation_filter_strength = 0.1; tm.flim_auto_black_point = 0; tm.flim_black_point = 0.02
            grade([0.45, 0.5, 0.55, 0.5, 0.5, 0.5, 0.55, 0.5, 0.45, -3.0, 2.0, 1.0])
        want = o.postprocess(acc, post, tm)
        got = np.fromfile(out + "_%d.rgba" % f, np.uint8).reshape(h, w, 4)
        assert np.array_equal(got, want), "frame %d: %d bytes differ" % (f, int((got != want).sum()))
        py.setPostProcessOptions(post)
        py.setTonemapOptions(tm)
        assert np.array_equal(py.readbackRenderTarget(), got)   # the Python host over the same C ABI
        images.append(got)
    assert all(not np.array_equal(images[i], images[j]) for i in range(4) for j in range(i))   # the edits do reach the image
