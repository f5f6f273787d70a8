#!/usr/bin/env python3
"""Render a scene file on the MI355X without the reference's frontend: the reference's own scene.json (+ _data.bin), a
glTF / GLB, or one of the built-in procedural workloads, through the whole chain — ingestion (libptamd scene_io), path
tracing (GMoN optional), fused post-process + tonemap — to an 8-bit PNG.

    python tools/render_scene.py tests/golden/scene_fixture/mini.json out.png --size 640 360 --spp 64
    python tools/render_scene.py model.glb out.png --camera-pos 0 1.5 6 --camera-target 0 1 0 --env sky
    python tools/render_scene.py builtin:c5 out.png --size 960 540 --spp 32 --bounces 12
"""
import argparse, os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from platinum_amd import Renderer, abi, scene_io, scenes


def write_png_rgba8(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene"); ap.add_argument("output")
    ap.add_argument("--size", type=int, nargs=2, default=(960, 540))
    ap.add_argument("--spp", type=int, default=64); ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--gmon", type=int, default=0, help="GMoN bucket count (0 = plain mean)")
    ap.add_argument("--camera-pos", type=float, nargs=3); ap.add_argument("--camera-target", type=float, nargs=3, default=(0, 0, 0))
    ap.add_argument("--focal", type=float, default=28.0)
    ap.add_argument("--env", default="none", help="'sky' = procedural sky, or the path of an .exr / Radiance .hdr environment map")
    ap.add_argument("--exposure", type=float, default=0.0)
    a = ap.parse_args()
    t0 = time.time()
    if a.scene.startswith("builtin:"):
        factory, *_ = scenes.CONFIGS[a.scene.split(":", 1)[1]]
        sc = factory()
    else:
        sc = scene_io.SceneFile.load(a.scene) if a.scene.endswith(".json") else scene_io.SceneFile.empty().import_gltf(a.scene, scene_io.GLTF_SKIP_EMPTY_NODES)
        if a.env == "sky":
            sc.set_environment(scenes.sky_environment(1024, 512))
        elif a.env != "none":
            sc.load_environment(a.env)
        if a.camera_pos is not None or not sc.cameras():
            sc.add_camera(a.camera_pos or (0.0, 1.5, 6.0), a.camera_target, a.focal)
        c = sc.counts()
        print(f"scene: {c.instances} instances, {c.triangles} triangles, {c.textures} textures, {c.materials} materials, cameras {sc.cameras()}")
    r = Renderer(device=0)
    flags = abi.FLAG_MULTISCATTER_GGX | (abi.FLAG_GMON if a.gmon > 1 else 0)
    r.startRender(sc, tuple(a.size), a.spp, gmonBuckets=max(1, a.gmon), flags=flags, max_bounces=a.bounces, nonfinite_policy=abi.NONFINITE_ZERO)
    t1 = time.time()
    r.render(0); r.wait()
    t2 = time.time()
    po = r.postProcessOptions(); po.exposure = a.exposure
    r.setPostProcessOptions(po)
    img = r.readbackRenderTarget()
    write_png_rgba8(a.output, img)
    st = r.stats()
    print(f"setup {t1 - t0:.2f} s (BVH {st.bvh_build_ms:.1f} ms), render {t2 - t1:.3f} s = {a.size[0] * a.size[1] * a.spp * a.bounces / (t2 - t1) / 1e6:.0f} Msamples/s, wrote {a.output}")


if __name__ == "__main__":
    main()
