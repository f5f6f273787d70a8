"""CPU: the C-ABI library loads and exports every symbol include/ptamd.h declares; struct layouts equal the reference's
(SURVEY §8b); the library refuses to run without a GPU (no CPU fallback); host scene model."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    hdr = open(os.path.join(ROOT, "include", "ptamd.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"pt_error"}
    assert declared == {name for name, _, _ in abi.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name)
    # include/ptamd_scene.h (scene ingestion)
    from platinum_amd import scene_io
    hdr2 = open(os.path.join(ROOT, "include", "ptamd_scene.h")).read()
    declared2 = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr2))
    assert declared2 == {name for name, _, _ in scene_io.SCENE_SYMBOLS}
    for name in declared2:
        assert hasattr(lib, name)


def test_struct_layouts_match_reference_abi():
    """Compile include/ptamd.h as C and check sizes/offsets against SURVEY §8b with the compiler's own offsetof."""
    src = r'''
#include <stddef.h>
#include "ptamd.h"
#define SA(c) _Static_assert(c, #c)
SA(sizeof(pt_float3) == 16); SA(sizeof(pt_vertex_data) == 48); SA(offsetof(pt_vertex_data, tangent) == 16); SA(offsetof(pt_vertex_data, texCoords) == 32);
SA(sizeof(pt_material_gpu) == 96); SA(offsetof(pt_material_gpu, emission) == 16); SA(offsetof(pt_material_gpu, emissionStrength) == 32);
SA(offsetof(pt_material_gpu, roughness) == 36); SA(offsetof(pt_material_gpu, ior) == 48); SA(offsetof(pt_material_gpu, clearcoatRoughness) == 64);
SA(offsetof(pt_material_gpu, flags) == 68); SA(offsetof(pt_material_gpu, baseTextureId) == 72); SA(offsetof(pt_material_gpu, normalTextureId) == 92);
SA(sizeof(pt_instance) == 64); SA(offsetof(pt_instance, options) == 48); SA(offsetof(pt_instance, accelerationStructureIndex) == 60);
SA(sizeof(pt_area_light) == 48); SA(offsetof(pt_area_light, area) == 16); SA(offsetof(pt_area_light, cumulativePower) == 24); SA(offsetof(pt_area_light, emission) == 32);
SA(sizeof(pt_camera_data) == 80); SA(offsetof(pt_camera_data, apertureRadius) == 64); SA(offsetof(pt_camera_data, bokehPower) == 76);
SA(sizeof(pt_constants) == 176); SA(offsetof(pt_constants, totalLightPower) == 32); SA(offsetof(pt_constants, size) == 40);
SA(offsetof(pt_constants, idt) == 48); SA(offsetof(pt_constants, camera) == 96);
#include <stdio.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pt_render_params), sizeof(pt_scene_snapshot), sizeof(pt_stats), sizeof(pt_create_info),
         sizeof(pt_hit_record), sizeof(pt_camera), sizeof(pt_mesh), sizeof(pt_post_options), sizeof(pt_tonemap_options));
  return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "t.c")
        open(f, "w").write(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), f, "-o", os.path.join(d, "t")])
        c_sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    # the ctypes mirror (platinum_amd/abi.py) must agree with the C compiler on every by-value struct
    py_sizes = [C.sizeof(t) for t in (abi.RenderParams, abi.SceneSnapshot, abi.Stats, abi.CreateInfo, abi.HitRecord, abi.Camera, abi.Mesh, abi.PostOptions, abi.TonemapOptions)]
    assert c_sizes == py_sizes


def test_no_gpu_means_no_renderer():
    """The product path fails loudly without a HIP device — there is no CPU fallback."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    lib = abi.load_library()
    info = abi.CreateInfo(abi.PT_ABI_VERSION, 0, None, 0, abi.LUT_PATH.encode())
    h = C.c_void_p()
    rc = lib.pt_create(C.byref(info), C.byref(h))
    assert rc == -2 and not h.value                          # PT_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.pt_last_error()
    from platinum_amd import Renderer
    with pytest.raises(abi.PtamdError):
        Renderer()


def test_missing_library_raises():
    with pytest.raises(abi.PtamdError):
        abi.load_library("/nonexistent/libptamd.so")


def test_primitives_match_reference_counts():
    assert scenes.cornell_box().triangle_count == 12 and len(scenes.cornell_box().positions) == 24   # primitives.cpp:133-190
    assert scenes.sphere(1.0, 48, 64).triangle_count == 6144                                             # scene_explorer.cpp:47
    assert scenes.sphere(0.25, 22, 23).triangle_count == 1012
    assert scenes.cube(2.0).triangle_count == 12 and scenes.plane(2.0).triangle_count == 2
    cb = scenes.cornell_box()
    assert cb.material_slots.tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3]
    assert np.allclose(cb.positions[20:24, 1], 9.99)                                                      # light quad 0.01 below the ceiling
    assert cb.positions[:, :3].min() == -5 and cb.positions[:20, 1].max() == 10
    f = scenes.field_scene(32)
    assert f.triangle_count == 12 + 1024 * 1012 == 1036300


def test_transform_and_look_at():
    t = scenes.Transform(translation=(1, 2, 3), scale=(2, 2, 2)).matrix()
    assert t[3].tolist() == [1, 2, 3, 1] and t[0][0] == 2
    cam = scenes.Transform(translation=(0, 5, 15), target=(0, 5, 0), track=True).matrix()
    assert np.allclose(cam[3][:3], [0, 5, 15]) and np.allclose(cam[2][:3], [0, 0, 1], atol=1e-6)  # camera looks down -z
    m = scenes.Material(emission=(1, 1, 1), emission_strength=50.0).to_gpu()
    assert m.flags & abi.MATERIAL_EMISSIVE and m.baseTextureId == -1
