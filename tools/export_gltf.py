#!/usr/bin/env python3
"""Write a procedural platinum_amd.scenes.Scene as the FILES the reference's loaders read — a binary glTF 2.0 (.glb: meshes,
node matrices, PBR materials with the KHR_materials_{emissive_strength, transmission, ior, clearcoat, anisotropy} extensions,
PNG or JPEG textures in bufferViews) and an OpenEXR environment (.exr: scanline, ZIP, FLOAT RGB) — so that the C5-class
workload enters the renderer through scene ingestion (pt_scene_import_gltf + pt_scene_load_environment) like
a real asset would, instead of through an in-memory snapshot.

    python tools/export_gltf.py atrium out_dir [--jpeg] [--env 4096x2048] [--columns 20]

`load_exported(glb, exr, camera)` (used by bench.py --workload c5 and tests) imports the pair again and returns a
scene_io.SceneFile whose snapshot() feeds Renderer.startRender and the oracle alike."""
import io
import json
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platinum_amd import abi, scenes  # noqa: E402


def png_bytes(img):
    """(H, W, C) uint8, C in 1 (grey), 3 (RGB), 4 (RGBA) -> PNG file bytes (filter 0, zlib)."""
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + np.ascontiguousarray(img[y]).tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def jpeg_bytes(rgb, quality=92):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb[..., :3]), "RGB").save(buf, "JPEG", quality=quality, subsampling=2)
    return buf.getvalue()


def write_exr_rgb(path, img):
    """(H, W, >=3) float32 -> single-part scanline OpenEXR 2.0, ZIP (16 lines per block), FLOAT channels B, G, R."""
    h, w = img.shape[:2]
    chans = {"B": img[..., 2], "G": img[..., 1], "R": img[..., 0]}

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IB3xII", 2, 0, 1, 1) for n in sorted(chans)) + b"\0"
    win = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([3]))
    hdr += attr("dataWindow", "box2i", win) + attr("displayWindow", "box2i", win) + attr("lineOrder", "lineOrder", bytes([0]))
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    starts = list(range(0, h, 16))
    blocks = []
    for y0 in starts:
        raw = b"".join(np.ascontiguousarray(chans[n][y], dtype=np.float32).tobytes() for y in range(y0, min(h, y0 + 16)) for n in sorted(chans))
        a = np.frombuffer(raw, dtype=np.uint8)
        inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)
        pred = np.empty_like(inter)
        pred[0] = inter[0]
        pred[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256
        cb = zlib.compress(pred.astype(np.uint8).tobytes(), 4)
        blocks.append(cb if len(cb) < len(raw) else raw)
    pos = len(hdr) + 8 * len(starts)
    table, body = b"", []
    for y0, blk in zip(starts, blocks):
        table += struct.pack("<Q", pos)
        body.append(struct.pack("<iI", y0, len(blk)) + blk)
        pos += 8 + len(blk)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(body))


class _Glb:
    def __init__(self):
        self.doc = {"asset": {"version": "2.0", "generator": "platinum_amd tools/export_gltf.py"}, "buffers": [], "bufferViews": [], "accessors": [],
                    "meshes": [], "materials": [], "nodes": [], "scenes": [{"nodes": []}], "scene": 0, "textures": [], "images": [],
                    "extensionsUsed": []}
        self.bin = bytearray()

    def view(self, data):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin.extend(data)
        self.doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, type_, component):
        a = np.ascontiguousarray(arr)
        acc = {"bufferView": self.view(a.tobytes()), "componentType": component, "count": int(len(a)), "type": type_}
        if type_ == "VEC3" and component == 5126:
            acc["min"] = [float(x) for x in a.min(0)]
            acc["max"] = [float(x) for x in a.max(0)]
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def use(self, ext):
        if ext not in self.doc["extensionsUsed"]:
            self.doc["extensionsUsed"].append(ext)

    def write(self, path):
        doc = {k: v for k, v in self.doc.items() if v or k in ("asset", "scene")}
        doc["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        b = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(b), 0x004E4942) + b
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)


def export_scene(scene, glb_path, exr_path=None, jpeg=False):
    """Writes `scene` (platinum_amd.scenes.Scene).  One glTF mesh per (mesh, material list) pair that occurs — glTF binds
    materials to primitives, the reference's scenes bind them to nodes — sharing the vertex accessors of the mesh."""
    g = _Glb()
    # textures: glTF stores roughness-metallic as G / B of an RGB image (loaders/gltf.cpp converts to RG8), everything else RGBA
    tex_index = {}

    def texture(i):
        if i < 0:
            return None
        if i not in tex_index:
            t = scene.textures[i]
            px = t.pixels
            if t.format == abi.TEX_RG8:
                rgb = np.zeros(px.shape[:2] + (3,), np.uint8)
                rgb[..., 1], rgb[..., 2] = px[..., 0], px[..., 1]
                data, mime = png_bytes(rgb), "image/png"
            elif t.format == abi.TEX_R8:
                data, mime = png_bytes(np.repeat(px, 3, axis=2)), "image/png"
            elif jpeg and t.format in (abi.TEX_RGBA8_SRGB, abi.TEX_RGBA8) and px[..., 3].min() == 255:
                data, mime = jpeg_bytes(px), "image/jpeg"
            else:
                data, mime = png_bytes(px), "image/png"
            g.doc["images"].append({"bufferView": g.view(data), "mimeType": mime})
            g.doc["textures"].append({"source": len(g.doc["images"]) - 1})
            tex_index[i] = len(g.doc["textures"]) - 1
        return {"index": tex_index[i]}

    mat_index = {}

    def material(m):
        key = repr(m)
        if key in mat_index:
            return mat_index[key]
        pbr = {"baseColorFactor": [float(x) for x in m.base_color], "metallicFactor": float(m.metallic), "roughnessFactor": float(m.roughness)}
        d = {"name": m.name, "pbrMetallicRoughness": pbr, "extensions": {}}
        if texture(m.base_texture):
            pbr["baseColorTexture"] = texture(m.base_texture)
        if texture(m.rm_texture):
            pbr["metallicRoughnessTexture"] = texture(m.rm_texture)
        if texture(m.normal_texture):
            d["normalTexture"] = texture(m.normal_texture)
        if m.emission_strength > 0 or m.emission_texture >= 0:
            d["emissiveFactor"] = [float(x) for x in m.emission]
            d["extensions"]["KHR_materials_emissive_strength"] = {"emissiveStrength": float(m.emission_strength)}
            if texture(m.emission_texture):
                d["emissiveTexture"] = texture(m.emission_texture)
        if m.transmission > 0 or m.transmission_texture >= 0:
            d["extensions"]["KHR_materials_transmission"] = {"transmissionFactor": float(m.transmission)}
            if texture(m.transmission_texture):
                d["extensions"]["KHR_materials_transmission"]["transmissionTexture"] = texture(m.transmission_texture)
        if m.ior != 1.5:
            d["extensions"]["KHR_materials_ior"] = {"ior": float(m.ior)}
        if m.clearcoat > 0 or m.clearcoat_texture >= 0:
            d["extensions"]["KHR_materials_clearcoat"] = {"clearcoatFactor": float(m.clearcoat), "clearcoatRoughnessFactor": float(m.clearcoat_roughness)}
        if m.anisotropy != 0:
            d["extensions"]["KHR_materials_anisotropy"] = {"anisotropyStrength": float(m.anisotropy), "anisotropyRotation": float(m.anisotropy_rotation)}
        for e in d["extensions"]:
            g.use(e)
        if not d["extensions"]:
            del d["extensions"]
        g.doc["materials"].append(d)
        mat_index[key] = len(g.doc["materials"]) - 1
        return mat_index[key]

    attr_cache, mesh_cache = {}, {}
    for node in scene.nodes:
        mats = tuple(material(m) for m in node.materials)
        key = (node.mesh, mats)
        if key not in mesh_cache:
            md = scene.meshes[node.mesh]
            if node.mesh not in attr_cache:
                vd = md.vertex_data.reshape(-1, 12)
                attr_cache[node.mesh] = {"POSITION": g.accessor(md.positions[:, :3].astype(np.float32), "VEC3", 5126),
                                         "NORMAL": g.accessor(vd[:, 0:3].astype(np.float32), "VEC3", 5126),
                                         "TEXCOORD_0": g.accessor(vd[:, 8:10].astype(np.float32), "VEC2", 5126)}
            tris = md.indices.reshape(-1, 3)
            prims = []
            for slot in sorted(set(int(s) for s in md.material_slots)):
                idx = tris[md.material_slots == slot].reshape(-1).astype(np.uint32)
                prims.append({"attributes": attr_cache[node.mesh], "indices": g.accessor(idx, "SCALAR", 5125), "material": mats[slot], "mode": 4})
            g.doc["meshes"].append({"name": f"mesh{node.mesh}", "primitives": prims})
            mesh_cache[key] = len(g.doc["meshes"]) - 1
        w = np.asarray(node.world, dtype=np.float32)  # [col][row]
        g.doc["nodes"].append({"name": node.materials[0].name or f"node{len(g.doc['nodes'])}", "mesh": mesh_cache[key],
                               "matrix": [float(w[c][r]) for c in range(4) for r in range(4)]})
        g.doc["scenes"][0]["nodes"].append(len(g.doc["nodes"]) - 1)
    g.write(glb_path)
    if exr_path is not None and scene.env_texture >= 0:
        write_exr_rgb(exr_path, scene.textures[scene.env_texture].pixels)


def load_exported(glb_path, exr_path, camera_position, camera_target, focal_length):
    """pt_scene_import_gltf + pt_scene_load_environment + the camera the frontend would add (scene_explorer.cpp:84-90)."""
    from platinum_amd import scene_io
    sc = scene_io.SceneFile.empty().import_gltf(glb_path)
    if exr_path is not None:
        sc.load_environment(exr_path)
    sc.add_camera(camera_position, camera_target, focal_length)
    return sc


ATRIUM_CAMERA = ((0.5, 2.2, 40.0 / 2 - 2.0), (0.0, 3.0, -40.0 / 2), 24.0)  # scenes.atrium_scene's tracking camera


def atrium_through_ingestion(cache_dir, env_size=(4096, 2048), columns=20, jpeg=True):
    """The C5-class workload as FILES: writes atrium.glb (JPEG + PNG textures) + atrium.exr once per parameter set, imports them."""
    os.makedirs(cache_dir, exist_ok=True)
    tag = f"atrium_{env_size[0]}x{env_size[1]}_{columns}{'_jpg' if jpeg else ''}"
    glb, exr = os.path.join(cache_dir, tag + ".glb"), os.path.join(cache_dir, tag + ".exr")
    if not (os.path.exists(glb) and os.path.exists(exr)):
        sc = scenes.atrium_scene(env_size=env_size, columns=columns)
        export_scene(sc, glb + ".tmp", exr + ".tmp", jpeg=jpeg)
        os.replace(glb + ".tmp", glb)
        os.replace(exr + ".tmp", exr)
    return load_exported(glb, exr, *ATRIUM_CAMERA)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", choices=["atrium", "textured", "c2", "c3"])
    ap.add_argument("out_dir")
    ap.add_argument("--jpeg", action="store_true")
    ap.add_argument("--env", default="4096x2048")
    ap.add_argument("--columns", type=int, default=20)
    a = ap.parse_args()
    ew, eh = (int(x) for x in a.env.split("x"))
    sc = {"atrium": lambda: scenes.atrium_scene(env_size=(ew, eh), columns=a.columns), "textured": scenes.textured_scene,
          "c2": scenes.cornell_sphere_scene, "c3": scenes.field_scene}[a.scene]()
    os.makedirs(a.out_dir, exist_ok=True)
    glb, exr = os.path.join(a.out_dir, a.scene + ".glb"), os.path.join(a.out_dir, a.scene + ".exr")
    export_scene(sc, glb, exr if sc.env_texture >= 0 else None, jpeg=a.jpeg)
    print(glb, os.path.getsize(glb), "bytes;", sc.triangle_count, "triangles")
