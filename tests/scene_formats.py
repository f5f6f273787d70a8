"""Test-side writers for the file formats the scene-ingestion component reads, written independently of
the C++ reader from the reference's own serializer, plus a ctypes binding of the reference's MikkTSpace
(oracle/_ref/libmikkt.so, built by `make -C oracle ref` from /root/reference/deps/mikkt/mikktspace.c where it lies).

  write_reference_scene   Scene::saveToFile                /root/reference/src/core/scene.cpp:536-787, utils/json.hpp:27-40
  write_png               PNG (zlib) encoder for texture fixtures
  write_gltf              a .gltf + .bin (or .glb) with the constructs loaders/gltf.cpp consumes
"""
import base64
import ctypes as C
import json
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32

MTL_FORMAT = {"r8": 10, "rg8": 30, "rgba8": 70, "srgb8": 71, "rgba32f": 125}
SLOT = {"base": 0, "rm": 1, "transmission": 2, "clearcoat": 3, "emission": 4, "normal": 5}  # material.hpp:16-23


def _fl(v):
    """json_utils::vec (utils/json.hpp:27-29): floats are stored as the double value of the float."""
    return [float(f32(x)) for x in v]


def transform_json(t=(0, 0, 0), r=(0, 0, 0), s=(1, 1, 1), tgt=(0, 0, 0), track=False):
    return {"t": _fl(t), "r": _fl(r), "s": _fl(s), "tgt": _fl(tgt), "track": bool(track)}


def node_json(node_id, name, transform=None, visible=True, mesh=None, materials=None, camera=None, children=()):
    """nodeToJson (core/scene.cpp:633-681). materials: list of asset ids or None (-> "default")."""
    j = {"id": int(node_id), "name": name, "visible": bool(visible), "transform": transform or transform_json(), "children": list(children)}
    if mesh is not None:
        j["mesh"] = {"id": int(mesh), "materials": ["default" if m is None else int(m) for m in (materials or [])]}
    if camera is not None:
        j["camera"] = {"f": float(f32(camera["f"])), "aperture": float(f32(camera.get("aperture", 0.0))),
                       "sensor": _fl(camera.get("sensor", (36.0, 24.0)))}
    return j


def material_json(name="", base_color=(0.8, 0.8, 0.8, 1.0), roughness=1.0, metallic=0.0, transmission=0.0, ior=1.5, aniso=0.0,
                  aniso_rotation=0.0, clearcoat=0.0, clearcoat_roughness=0.05, emission=(0, 0, 0), emission_strength=0.0,
                  thin=False, textures=()):
    """toJson(AssetData<Material>) (core/scene.cpp:738-763). textures: [(slot name, texture asset id)]."""
    g = lambda x: float(f32(x))
    return {"name": name, "baseColor": _fl(base_color), "roughness": g(roughness), "metallic": g(metallic), "transmission": g(transmission),
            "ior": g(ior), "aniso": g(aniso), "anisoRotation": g(aniso_rotation), "clearcoat": g(clearcoat),
            "clearcoatRoughness": g(clearcoat_roughness), "emission": _fl(emission), "emissionStrength": g(emission_strength),
            "thinTransmission": bool(thin), "textures": [[SLOT[s], int(t)] for s, t in textures]}


def write_reference_scene(json_path, assets, root, envmap=None, next_id=None):
    """assets: list of dicts in file order:
         {"id", "type": "texture", "name", "alpha", "format": key of MTL_FORMAT, "pixels": ndarray (H, W, C)}
         {"id", "type": "mesh", "mesh": scenes.MeshData}
         {"id", "type": "material", "data": material_json(...)}
       root: node_json(...) tree.  envmap: {"texture": id, "alias": structured ndarray(pdf, p, aliasIdx)} or None.
       Writes <stem>.json and <stem>_data.bin exactly as Scene::saveToFile lays them out."""
    stem = os.path.splitext(os.path.basename(json_path))[0]
    bin_path = os.path.join(os.path.dirname(json_path), stem + "_data.bin")
    blob = bytearray()

    def dump(b):
        off = len(blob)
        blob.extend(b)
        return [off, len(b)]

    out_assets = []
    for a in assets:
        j = {"id": int(a["id"]), "retain": bool(a.get("retain", True)), "rc": int(a.get("rc", 1)), "type": a["type"]}
        if a["type"] == "texture":
            px = np.ascontiguousarray(a["pixels"])
            j["data"] = {"name": a.get("name", ""), "alpha": bool(a.get("alpha", False)), "size": [int(px.shape[1]), int(px.shape[0])],
                         "format": MTL_FORMAT[a["format"]], "data": dump(px.tobytes())}
        elif a["type"] == "mesh":
            m = a["mesh"]
            pos = np.ascontiguousarray(m.positions, dtype=f32)
            vd = np.ascontiguousarray(m.vertex_data, dtype=f32)
            idx = np.ascontiguousarray(m.indices, dtype=np.uint32)
            slots = np.ascontiguousarray(m.material_slots, dtype=np.uint32)
            j["data"] = {"indexCount": int(len(idx)), "vertexCount": int(len(pos)), "positions": dump(pos.tobytes()),
                         "vertexData": dump(vd.tobytes()), "indices": dump(idx.tobytes()), "materials": dump(slots.tobytes())}
        else:
            j["data"] = a["data"]
        out_assets.append(j)
    doc = {"root": root, "assets": {"nextId": int(next_id if next_id is not None else max([a["id"] for a in assets] + [-1]) + 1),
                                    "assets": out_assets}}
    if envmap is not None:
        doc["envmap"] = {"texture": int(envmap["texture"]), "aliasTable": dump(np.ascontiguousarray(envmap["alias"]).tobytes())}
    with open(bin_path, "wb") as f:
        f.write(bytes(blob))
    with open(json_path, "w") as f:
        json.dump(doc, f)
    return doc


# ---- PNG --------------------------------------------------------------------------------------------------------
def _png_filtered(a, depth, filter_type):
    """Filtered scanlines (filter byte + bytes) of one (H, W, C) image — the whole picture, or one Adam7 pass."""
    h, w, c = a.shape
    raw_rows = a.astype(">u2").tobytes() if depth == 16 else a.astype(np.uint8).tobytes()
    stride = w * c * depth // 8
    bpp = max(1, c * depth // 8)
    out = bytearray()
    prev = bytearray(stride)
    for y in range(h):
        row = bytearray(raw_rows[y * stride:(y + 1) * stride])
        ft = (y % 5) if filter_type is None else filter_type
        enc = bytearray(stride)
        for i in range(stride):
            A = row[i - bpp] if i >= bpp else 0
            B = prev[i]
            Cc = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = A
            elif ft == 2: pred = B
            elif ft == 3: pred = (A + B) >> 1
            else:
                p = A + B - Cc
                pa, pb, pc = abs(p - A), abs(p - B), abs(p - Cc)
                pred = A if (pa <= pb and pa <= pc) else (B if pb <= pc else Cc)
            enc[i] = (row[i] - pred) & 0xFF
        out.append(ft)
        out.extend(enc)
        prev = row
    return out


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))  # x0, y0, dx, dy


def png_bytes(img, palette=None, trns=None, filter_type=None, interlace=False):
    """Encode an (H, W[, C]) uint8/uint16 array: C = 1 grey, 2 grey+alpha, 3 RGB, 4 RGBA; with `palette` (N,3) the array
    holds indices (colour type 3).  Filters are cycled per row unless fixed (exercises every unfilter path).
    interlace=True writes the seven Adam7 reduced images."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[..., None]
    h, w, c = a.shape
    depth = 16 if a.dtype == np.uint16 else 8
    ctype = 3 if palette is not None else {1: 0, 2: 4, 3: 2, 4: 6}[c]
    if interlace:
        out = bytearray()
        for x0, y0, dx, dy in ADAM7:
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                out.extend(_png_filtered(np.ascontiguousarray(sub), depth, filter_type))
    else:
        out = _png_filtered(a, depth, filter_type)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        png += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    if trns is not None:
        png += chunk(b"tRNS", bytes(trns))
    data = zlib.compress(bytes(out), 6)
    half = len(data) // 2
    png += chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:]) + chunk(b"IEND", b"")  # split IDAT on purpose
    return png


# ---- glTF -------------------------------------------------------------------------------------------------------
class GltfBuilder:
    """Just enough of a glTF 2.0 writer for the importer tests: float/short/byte accessors, interleaved views, embedded or
    external buffers, GLB, TRS and matrix nodes, cameras, the KHR material extensions gltf.cpp enables."""

    def __init__(self):
        self.doc = {"asset": {"version": "2.0"}, "buffers": [], "bufferViews": [], "accessors": [], "meshes": [], "materials": [],
                    "nodes": [], "scenes": [], "textures": [], "images": [], "cameras": []}
        self.bin = bytearray()

    def view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin.extend(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, type_, component=5126, normalized=False, view=None, offset=0, count=None):
        a = np.ascontiguousarray(arr)
        if view is None:
            view = self.view(a.tobytes())
        acc = {"bufferView": view, "componentType": component, "count": int(count if count is not None else len(a)), "type": type_}
        if offset:
            acc["byteOffset"] = offset
        if normalized:
            acc["normalized"] = True
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def image_png(self, png, embed="view", dirpath=None, name="tex.png"):
        if embed == "view":
            self.doc["images"].append({"bufferView": self.view(png), "mimeType": "image/png"})
        elif embed == "data":
            self.doc["images"].append({"uri": "data:image/png;base64," + base64.b64encode(png).decode()})
        else:
            with open(os.path.join(dirpath, name), "wb") as f:
                f.write(png)
            self.doc["images"].append({"uri": name})
        self.doc["textures"].append({"source": len(self.doc["images"]) - 1, "name": name})
        return len(self.doc["textures"]) - 1

    def write(self, path, glb=False, embed_buffer=False):
        doc = {k: v for k, v in self.doc.items() if v or k == "asset"}
        if glb:
            doc["buffers"] = [{"byteLength": len(self.bin)}]
            js = json.dumps(doc).encode()
            js += b" " * (-len(js) % 4)
            b = bytes(self.bin) + b"\0" * (-len(self.bin) % 4)
            body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(b), 0x004E4942) + b
            with open(path, "wb") as f:
                f.write(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)
            return
        if embed_buffer:
            doc["buffers"] = [{"byteLength": len(self.bin), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}]
        else:
            name = os.path.splitext(os.path.basename(path))[0] + " data.bin"  # a space: exercises URI decoding
            with open(os.path.join(os.path.dirname(path), name), "wb") as f:
                f.write(bytes(self.bin))
            doc["buffers"] = [{"byteLength": len(self.bin), "uri": name.replace(" ", "%20")}]
        with open(path, "w") as f:
            json.dump(doc, f)


# ---- the reference's MikkTSpace -----------------------------------------------------------------------------------
MIKKT_PATH = os.path.join(ROOT, "oracle", "_ref", "libmikkt.so")


def mikkt_available():
    return os.path.exists(MIKKT_PATH)


def mikkt_reference_tangents(positions, vertex_data, indices):
    """Runs genTangSpaceDefault with the callbacks of core/mesh.cpp:11-57 (per face-vertex results written onto the shared
    vertex, last write wins) and returns the (V, 4) tangents."""
    lib = C.CDLL(MIKKT_PATH)
    pos = np.ascontiguousarray(positions, dtype=f32)
    vd = np.ascontiguousarray(vertex_data, dtype=f32)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    out = np.zeros((len(pos), 4), dtype=f32)
    FP = C.POINTER(C.c_float)
    GET_N = C.CFUNCTYPE(C.c_int, C.c_void_p)
    GET_NV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
    GET3 = C.CFUNCTYPE(None, C.c_void_p, FP, C.c_int, C.c_int)
    SET_BASIC = C.CFUNCTYPE(None, C.c_void_p, FP, C.c_float, C.c_int, C.c_int)
    SET_FULL = C.CFUNCTYPE(None, C.c_void_p, FP, FP, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int)

    class Interface(C.Structure):
        _fields_ = [("getNumFaces", GET_N), ("getNumVerticesOfFace", GET_NV), ("getPosition", GET3), ("getNormal", GET3),
                    ("getTexCoord", GET3), ("setTSpaceBasic", SET_BASIC), ("setTSpace", SET_FULL)]

    class Context(C.Structure):
        _fields_ = [("iface", C.POINTER(Interface)), ("user", C.c_void_p)]

    def get_pos(ctx, o, f, v):
        p = pos[idx[3 * f + v]]
        o[0], o[1], o[2] = p[0], p[1], p[2]

    def get_nrm(ctx, o, f, v):
        n = vd[idx[3 * f + v]]
        o[0], o[1], o[2] = n[0], n[1], n[2]

    def get_uv(ctx, o, f, v):
        t = vd[idx[3 * f + v]]
        o[0], o[1] = t[8], t[9]

    def set_basic(ctx, t, sign, f, v):
        out[idx[3 * f + v]] = (t[0], t[1], t[2], sign)

    iface = Interface(GET_N(lambda ctx: len(idx) // 3), GET_NV(lambda ctx, f: 3), GET3(get_pos), GET3(get_nrm), GET3(get_uv),
                      SET_BASIC(set_basic), SET_FULL())
    ctx = Context(C.pointer(iface), None)
    lib.genTangSpaceDefault.restype = C.c_int
    lib.genTangSpaceDefault.argtypes = [C.POINTER(Context)]
    lib.genTangSpaceDefault(C.byref(ctx))
    return out


# ---- OpenEXR / Radiance HDR writers (environment-map fixtures) --------------------------------------------------------
def write_exr(path, channels, compression="zip", pixel_type="float", data_window_origin=(0, 0), line_order=0):
    """channels: dict name -> (H, W) float array.  Single-part scanline OpenEXR 2.0: NONE / RLE / ZIPS / ZIP, HALF or FLOAT."""
    names = sorted(channels)  # the file stores channels in alphabetical order
    h, w = channels[names[0]].shape
    comp = {"none": 0, "rle": 1, "zips": 2, "zip": 3}[compression]
    ptype = {"half": 1, "float": 2}[pixel_type]
    dt = np.float16 if ptype == 1 else np.float32

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IB3xII", ptype, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = data_window_origin
    win = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = struct.pack("<II", 20000630, 2)
    hdr += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", win) + attr("displayWindow", "box2i", win) + attr("lineOrder", "lineOrder", bytes([line_order]))
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines_per_block = 16 if comp == 3 else 1
    starts = list(range(0, h, lines_per_block))
    if line_order == 1:
        starts = starts[::-1]  # decreasing y: blocks are stored bottom-up, the offset table is still indexed by increasing y

    def pack_block(y):
        raw = b"".join(np.ascontiguousarray(channels[n][yy], dtype=dt).tobytes() for yy in range(y, min(h, y + lines_per_block)) for n in names)
        if comp == 0:
            return raw
        a = np.frombuffer(raw, dtype=np.uint8)
        inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)      # de-interleave into two halves
        pred = np.empty_like(inter)
        pred[0] = inter[0]
        pred[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256            # delta predictor
        pb = pred.astype(np.uint8).tobytes()
        if comp == 1:
            out = bytearray()
            i = 0
            while i < len(pb):  # simple RLE: runs of >= 3 equal bytes, else literals
                j = i
                while j + 1 < len(pb) and pb[j + 1] == pb[i] and j - i < 126:
                    j += 1
                if j - i >= 2:
                    out += bytes([j - i]) + pb[i:i + 1]
                    i = j + 1
                else:
                    k = i
                    while k < len(pb) and k - i < 127 and not (k + 2 < len(pb) and pb[k] == pb[k + 1] == pb[k + 2]):
                        k += 1
                    out += bytes([(-(k - i)) & 0xFF]) + pb[i:k]
                    i = k
            cb = bytes(out)
        else:
            cb = zlib.compress(pb, 6)
        return cb if len(cb) < len(raw) else raw  # OpenEXR stores a block raw when compression does not pay

    blocks = {y: pack_block(y) for y in starts}
    table_pos = len(hdr)
    pos = table_pos + 8 * len(starts)
    offsets = {}
    body = b""
    for y in starts:
        offsets[y] = pos
        chunk = struct.pack("<iI", y + y0, len(blocks[y])) + blocks[y]
        body += chunk
        pos += len(chunk)
    table = b"".join(struct.pack("<Q", offsets[y]) for y in sorted(starts))
    with open(path, "wb") as f:
        f.write(hdr + table + body)


def write_exr_tiled(path, channels, tile=(32, 16), compression="zip", pixel_type="float", mipmap=False, line_order=0):
    """Single-part TILED OpenEXR 2.0 (NONE / ZIP), one level or MIPMAP_LEVELS + ROUND_DOWN.  The further levels of a mipmap file hold
    the constant 7 (a reader that assembled anything but level 0 would show it).  Offset table: level by level, tiles row-major."""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    comp = {"none": 0, "zip": 3}[compression]
    ptype = {"half": 1, "float": 2}[pixel_type]
    dt = np.float16 if ptype == 1 else np.float32
    tx, ty = tile

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IB3xII", ptype, 0, 1, 1) for n in names) + b"\0"
    win = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2 | 0x200)
    hdr += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp]))
    hdr += attr("dataWindow", "box2i", win) + attr("displayWindow", "box2i", win) + attr("lineOrder", "lineOrder", bytes([line_order]))
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tx, ty, 1 if mipmap else 0)) + b"\0"
    levels = [(w, h)]
    if mipmap:
        n = int(np.floor(np.log2(max(w, h)))) + 1
        levels = [(max(1, w >> l), max(1, h >> l)) for l in range(n)]

    def pack(raw):
        if comp == 0:
            return raw
        a = np.frombuffer(raw, dtype=np.uint8)
        inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
        pred = np.empty_like(inter)
        pred[0] = inter[0]
        pred[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256
        cb = zlib.compress(pred.astype(np.uint8).tobytes(), 6)
        return cb if len(cb) < len(raw) else raw

    chunks = []
    for l, (lw, lh) in enumerate(levels):
        for j in range((lh + ty - 1) // ty):
            for i in range((lw + tx - 1) // tx):
                x0, y0 = i * tx, j * ty
                x1, y1 = min(lw, x0 + tx), min(lh, y0 + ty)
                if l == 0:
                    raw = b"".join(np.ascontiguousarray(channels[n][yy, x0:x1], dtype=dt).tobytes() for yy in range(y0, y1) for n in names)
                else:
                    raw = np.full((y1 - y0) * len(names) * (x1 - x0), 7.0, dt).tobytes()
                data = pack(raw)
                chunks.append(struct.pack("<iiiiI", i, j, l, l, len(data)) + data)
    pos = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(chunks))


def write_exr_blocks(path, names, w, h, pixel_type, comp_code, lines_per_block, blocks):
    """A single-part scanline OpenEXR whose (already encoded) block payloads are given by the caller — for crafted / malformed inputs."""
    ptype = {"half": 1, "float": 2}[pixel_type]

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<IB3xII", ptype, 0, 1, 1) for n in sorted(names)) + b"\0"
    win = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2)
    hdr += attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp_code]))
    hdr += attr("dataWindow", "box2i", win) + attr("displayWindow", "box2i", win) + attr("lineOrder", "lineOrder", bytes([0]))
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    starts = list(range(0, h, lines_per_block))
    assert len(starts) == len(blocks)
    pos = len(hdr) + 8 * len(starts)
    table, body = b"", b""
    for y, blk in zip(starts, blocks):
        table += struct.pack("<Q", pos)
        chunk = struct.pack("<iI", y, len(blk)) + blk
        body += chunk
        pos += len(chunk)
    with open(path, "wb") as f:
        f.write(hdr + table + body)


def piz_block_with_code_lengths(lengths, nbits=8, payload=b"\0"):
    """A PIZ block whose Huffman code-length table holds `lengths` for symbols 0..len-1 (6 bits each, hufUnpackEncTable layout)."""
    bits = "".join(format(l, "06b") for l in lengths)
    bits += "0" * (-len(bits) % 8)
    table = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    huf = struct.pack("<IIIII", 0, len(lengths) - 1, len(table), nbits, 0) + table + payload
    return struct.pack("<HH", 0, 0) + b"\x01" + struct.pack("<i", len(huf)) + huf


def write_radiance_hdr(path, rgbe, rle=True):
    """rgbe: (H, W, 4) uint8.  New-style RLE scanlines (per channel) or flat."""
    h, w, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\n# test\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
    for y in range(h):
        if not rle or w < 8 or w >= 32768:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c].tobytes()
            i = 0
            while i < w:
                j = i
                while j + 1 < w and row[j + 1] == row[i] and j - i < 126:
                    j += 1
                if j - i >= 2:
                    out += bytes([128 + (j - i + 1), row[i]])
                    i = j + 1
                else:
                    k = i
                    while k < w and k - i < 128 and not (k + 2 < w and row[k] == row[k + 1] == row[k + 2]):
                        k += 1
                    out += bytes([k - i]) + row[i:k]
                    i = k
    with open(path, "wb") as f:
        f.write(bytes(out))


EXR2RAW_PATH = os.path.join(ROOT, "oracle", "_ref", "exr2raw")


def tinyexr_reference_rgba(exr_path, tmp_dir):
    """The reference's own tinyexr LoadEXR (oracle/_ref/exr2raw, compiled from /root/reference/deps/tinyexr where it lies)."""
    import subprocess
    out = os.path.join(tmp_dir, "ref.f32")
    subprocess.check_call([EXR2RAW_PATH, exr_path, out, "rgba"])
    raw = open(out, "rb").read()
    w, h = struct.unpack_from("<ii", raw, 0)
    return np.frombuffer(raw, dtype=np.float32, offset=8).reshape(h, w, 4)


def stbi_available():
    return os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "stbi2raw"))


def stbi_reference_rgba(path):
    """(H, W, 4) uint8 decoded by the reference's own stb_image v2.30 exactly as loaders/texture.cpp:111-119 calls it."""
    exe = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "stbi2raw")
    out = subprocess.run([exe, str(path)], capture_output=True, check=True).stdout
    hdr, raw = out.split(b"\n", 1)
    w, h = map(int, hdr.split())
    return np.frombuffer(raw, np.uint8).reshape(h, w, 4).copy()
