#!/usr/bin/env python3
"""Mint the self-made pins of SURVEY §8c (the reference ships no golden vectors) from the CPU oracle and commit them
as small fixtures under tests/golden/.  Re-running must reproduce the files bit-for-bit.

    python tools/make_golden.py

  sampler_kat.json        integer known-answers of pcg4d / Halton offset (computed from
                          samplers.metal:16-23,154-156 by the surveyor) + Halton values of the first dims
  c1_cornell_golden.npz   C1 (Cornell 512x512, 4 bounces, MIS): 64x64 centre crop of the accumulator at 1, 4 and 64 spp (64 = the
                          full BASELINE.json configs[0]), per-channel means and a sha256 of the full images; primary-ray
                          (instance, primitive) map checksum and a 64x64 crop of (t,u,v)
  c2_small_golden.npz     Cornell + glass sphere, 160x90, 8 bounces: full accumulator at 2 spp + per-bounce hit ids of sample 0
  mikkt_tangents.npz      tangents of five meshes computed by the REFERENCE's deps/mikkt/mikktspace.c (compiled where it lies into
                          oracle/_ref/libmikkt.so) through the callbacks of core/mesh.cpp:11-57
  scene_fixture/mini.*    a scene.json + _data.bin pair in the reference's format (tests/scene_formats.py restates saveToFile)
  exr_piz_fixture.exr/npz a 96x48 RGB half PIZ OpenEXR ENCODED by the reference's tinyexr (oracle/_ref/exrwrite) and the RGBA floats its
                          LoadEXR decodes from it (oracle/_ref/exr2raw)
  exr_piz_tiled_fixture.* the same for a 70x45 file in 32x16 PIZ tiles (oracle/_ref/exrwrite … 32 16)
  jpeg_stb_fixture.npz    nine small JPEG files (baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, grey, CMYK, restart markers) and the RGBA8
                          the reference's vendored stb_image (oracle/_ref/stbi2raw) decodes from each
  n3_textured_golden.npz  scenes.textured_scene() (textures, normal map, cut-outs, environment), 96x54, 6 bounces: accumulator
                          at 2 spp, per-bounce hit ids of sample 0, the environment alias table
"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from platinum_amd import scenes
from platinum_amd.renderer import make_params

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
L = oracle_lib.lib()

kat = {
    "source": "SURVEY.md §8a (integer arithmetic of samplers.metal:16-23,154-156)",
    "offsets": [[0, 0, 0, 0x0f02f829], [1, 0, 0, 0x87708cf9], [0, 1, 0, 0xd42dba74], [17, 5, 3, 0x5ca03064], [1919, 1079, 255, 0xe596edc1]],
    "halton_exact": [[1, 0, 0.5], [2, 0, 0.25], [1, 1, 1.0 / 3.0]],
    "primes": {"count": 620, "first": 2, "last": 4583},
    "halton_bits": [[i, d, int(np.float32(L.orc_halton(i, d)).view(np.uint32))] for i in (1, 12345, 0x0f02f829, 0xffffffff) for d in (0, 1, 2, 3, 7, 100, 619)],
}
json.dump(kat, open(os.path.join(G, "sampler_kat.json"), "w"), indent=1)

sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

# ---- C1 ----
sc = scenes.cornell_scene("bench")
W = H = 512
o = oracle_lib.OracleScene(sc, make_params(W, H, 4, 4))
acc1 = o.render(0, 1)
acc4 = o.render(1, 3, acc=acc1.copy(), acc_n0=1)
acc64 = o.render(4, 60, acc=acc4.copy(), acc_n0=4)   # the parity gate is C1 at 1 / 4 / 64 spp
prim = o.trace_primary(0)
c = slice(224, 288)
np.savez_compressed(os.path.join(G, "c1_cornell_golden.npz"),
                    acc1_crop=acc1[c, c], acc4_crop=acc4[c, c], acc1_mean=acc1[..., :3].mean((0, 1)), acc4_mean=acc4[..., :3].mean((0, 1)),
                    acc64_crop=acc64[c, c], acc64_mean=acc64[..., :3].mean((0, 1)), acc64_sha=sha(acc64),
                    acc1_sha=sha(acc1), acc4_sha=sha(acc4), prim_ids_sha=sha(np.stack([prim["instance"], prim["primitive"]], -1)),
                    prim_tuv_crop=np.stack([prim["t"], prim["u"], prim["v"]], -1)[c, c], prim_ids_crop=np.stack([prim["instance"], prim["primitive"]], -1)[c, c])
# ---- small C2 ----
sc2 = scenes.cornell_sphere_scene()
o2 = oracle_lib.OracleScene(sc2, make_params(160, 90, 2, 8))
acc2 = o2.render(0, 2)
rad0, hits0 = o2.debug_sample(0)
np.savez_compressed(os.path.join(G, "c2_small_golden.npz"), acc2=acc2, hits0=hits0.astype(np.int16), rad0=rad0)
# ---- N3 textured scene ----
o3 = oracle_lib.OracleScene(scenes.textured_scene(), make_params(96, 54, 2, 6))
acc3 = o3.render(0, 2)
rad3, hits3 = o3.debug_sample(0)
al = o3.envAlias()
np.savez_compressed(os.path.join(G, "n3_textured_golden.npz"), acc2=acc3, hits0=hits3.astype(np.int16),
                    alias_pdf=al["pdf"], alias_p=al["p"], alias_idx=al["aliasIdx"])
# ---- N4: tangents from the reference's own mikktspace.c (oracle/_ref/libmikkt.so) and the scene.json fixture ----
import scene_formats as sf
import test_scene_ingestion as tsi
if sf.mikkt_available():
    tang = {}
    for name, m in tsi.tangent_cases().items():
        vd = np.ascontiguousarray(m.vertex_data, dtype=np.float32).copy()
        vd[:, 4:8] = 0
        tang[name] = sf.mikkt_reference_tangents(m.positions, vd, m.indices)
    np.savez_compressed(os.path.join(G, "mikkt_tangents.npz"), **tang)
else:
    print("oracle/_ref/libmikkt.so not built (make -C oracle ref): mikkt_tangents.npz left as is")
os.makedirs(os.path.join(G, "scene_fixture"), exist_ok=True)
assets, root, envmap, _ = tsi.mini_scene_spec()
sf.write_reference_scene(os.path.join(G, "scene_fixture", "mini.json"), assets, root, envmap)
# ---- N4: a PIZ-compressed OpenEXR written by the reference's tinyexr (oracle/_ref/exrwrite) and what its LoadEXR reads back ----
exrwrite, exr2raw = os.path.join(ROOT, "oracle", "_ref", "exrwrite"), os.path.join(ROOT, "oracle", "_ref", "exr2raw")
if os.path.exists(exrwrite) and os.path.exists(exr2raw):
    import subprocess, tempfile
    env = scenes.sky_environment(96, 48, sun=(30, 9), sun_radiance=300.0)[..., :3].copy()
    with tempfile.TemporaryDirectory() as td:
        env.astype(np.float32).tofile(os.path.join(td, "in.f32"))
        subprocess.check_call([exrwrite, os.path.join(td, "in.f32"), "96", "48", "3", "half", "piz", os.path.join(G, "exr_piz_fixture.exr")])
        np.savez_compressed(os.path.join(G, "exr_piz_fixture.npz"), rgba=sf.tinyexr_reference_rgba(os.path.join(G, "exr_piz_fixture.exr"), td))
        # ... and the same encoder writing 32x16 TILES (edge tiles 6 wide / 13 high)
        env2 = (scenes.sky_environment(70, 45, sun=(20, 8), sun_radiance=150.0)[..., :3] + np.random.default_rng(77).random((45, 70, 3)) * 0.25).astype(np.float32)
        env2.tofile(os.path.join(td, "in2.f32"))
        subprocess.check_call([exrwrite, os.path.join(td, "in2.f32"), "70", "45", "3", "half", "piz", os.path.join(G, "exr_piz_tiled_fixture.exr"), "32", "16"])
        np.savez_compressed(os.path.join(G, "exr_piz_tiled_fixture.npz"), rgba=sf.tinyexr_reference_rgba(os.path.join(G, "exr_piz_tiled_fixture.exr"), td))
else:
    print("oracle/_ref/exrwrite not built: exr_piz_fixture left as is")
# ---- N4: JPEG files (encoded by PIL / libjpeg: test infrastructure) and the RGBA8 the REFERENCE's stb_image decodes from them ----
stbi2raw = os.path.join(ROOT, "oracle", "_ref", "stbi2raw")
try:
    from PIL import Image
except Exception:
    Image = None
if os.path.exists(stbi2raw) and Image is not None:
    import io, subprocess, tempfile
    fx = {}
    for k, (w, h, mode, kw) in enumerate(tsi.jpeg_fixture_cases()):
        buf = io.BytesIO()
        tsi.jpeg_test_image(w, h, mode).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "t.jpg")
            open(path, "wb").write(data)
            fx[f"jpg_{k}"] = np.frombuffer(data, np.uint8)
            fx[f"rgba_{k}"] = sf.stbi_reference_rgba(path)
    np.savez_compressed(os.path.join(G, "jpeg_stb_fixture.npz"), **fx)
else:
    print("oracle/_ref/stbi2raw or PIL missing: jpeg_stb_fixture.npz left as is")
for f in sorted(os.listdir(G)):
    print(f, os.path.getsize(os.path.join(G, f)))
