"""The C++ host side above the C ABI: include/ptamd_renderer.hpp carries the public members of the reference's
pt::renderer_pt::Renderer (renderer_pt.hpp:14-73); tests/cpp/shim_render.cpp drives it the way the reference's frontend drives
that class (startRender, one render() per frame while status() says busy, renderProgress(), readback / present)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "shim_render.cpp")
HDRS = [os.path.join(ROOT, "include", h) for h in ("ptamd_renderer.hpp", "ptamd.h", "ptamd_scene.h")]
EXE = os.path.join(ROOT, "tests", "_build", "shim_render")
LIBDIR = os.path.join(ROOT, "platinum_amd", "csrc")
FIXTURE = os.path.join(ROOT, "tests", "golden", "scene_fixture", "mini.json")
LUT = os.path.join(ROOT, "platinum_amd", "data", "ggx_luts.bin")


def build_shim():
    lib = os.path.join(LIBDIR, "libptamd.so")
    if not os.path.exists(lib):
        pytest.fail("libptamd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    newest = max(os.path.getmtime(p) for p in [SRC, lib] + HDRS)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
                               "-L" + LIBDIR, "-lptamd", "-ldl", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


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
                   "int status()", "renderProgress()", "renderTime()", "postProcessOptions()", "tonemapOptions()", "gmonOptions()", "outputColorspace()",
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
