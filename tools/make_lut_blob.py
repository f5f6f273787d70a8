#!/usr/bin/env python3
"""Convert the reference's GGX energy LUT data files (resource/lut/*.exr, 101 single-channel
FLOAT/ZIP OpenEXR images) into ONE raw little-endian fp32 blob the MI355X renderer loads.

The reference loads the same files at start-up (renderer_pt.cpp:385-446, table renderer_pt.hpp:154-165):
8 LUTs, the 3-D ones as 32 slices `<name>_<z>.exr`.  The blob keeps exactly that order so
`Luts` (pt_shader_defs.hpp:130-139) maps 1:1.  Run in the authoring container only (the GPU box
has no /root/reference); the output is committed as data:

    python tools/make_lut_blob.py [--ref /root/reference] [--check-tinyexr oracle/_ref/exr2raw]

Blob layout (all little-endian):
    char[8]  magic  "PTLUT01\\0"
    u32      count  (= 8)
    count x { u32 width, u32 height, u32 depth, u32 offset_in_floats }
    f32[]    texels, per LUT [z][y][x]  (x fastest)

The EXR decode here is an independent pure-python implementation (zlib + OpenEXR predictor +
byte de-interleave); with --check-tinyexr every image is also decoded by the reference's own
vendored tinyexr (built by `make -C oracle ref`) and compared bit-for-bit.
"""
import argparse
import hashlib
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

# (file stem, depth) in the order of renderer_pt.hpp:154-165
LUTS = [
    ("ggx_E", 1),
    ("ggx_E_avg", 1),
    ("ggx_ms_E", 32),
    ("ggx_ms_E_avg", 1),
    ("ggx_E_trans_in", 32),
    ("ggx_E_trans_out", 32),
    ("ggx_E_trans_in_avg", 1),
    ("ggx_E_trans_out_avg", 1),
]
MAGIC = b"PTLUT01\0"


def _read_cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("ascii"), end + 1


def read_exr_y_float(path):
    """Decode a single-part scanline OpenEXR with one FLOAT channel, NONE/ZIPS/ZIP compression."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", buf, 0)
    assert magic == 20000630, "not an EXR"
    assert (version & 0x200) == 0, "tiled EXR not supported"
    pos = 8
    attrs = {}
    while True:
        name, pos = _read_cstr(buf, pos)
        if name == "":
            break
        typ, pos = _read_cstr(buf, pos)
        (size,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    # channels
    ch = attrs["channels"][1]
    cpos = 0
    chans = []
    while ch[cpos] != 0:
        cname, cpos = _read_cstr(ch, cpos)
        ptype, _plin, xs, ys = struct.unpack_from("<IB3xII", ch, cpos)
        cpos += 16
        chans.append((cname, ptype, xs, ys))
    assert len(chans) == 1 and chans[0][1] == 2, f"expected one FLOAT channel, got {chans}"
    comp = attrs["compression"][1][0]
    assert comp in (0, 2, 3), f"unsupported compression {comp}"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines_per_chunk = {0: 1, 2: 1, 3: 16}[comp]
    nchunks = (h + lines_per_chunk - 1) // lines_per_chunk
    offsets = struct.unpack_from(f"<{nchunks}Q", buf, pos)
    out = np.zeros((h, w), dtype="<f4")
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        nlines = min(lines_per_chunk, y0 + h - y)
        raw_size = nlines * w * 4
        if comp != 0 and size < raw_size:
            tmp = np.frombuffer(zlib.decompress(data), dtype=np.uint8)
            assert tmp.size == raw_size
            t = _undo_predictor(tmp)
            half = (raw_size + 1) // 2
            res = np.empty(raw_size, dtype=np.uint8)
            res[0::2] = t[:half]
            res[1::2] = t[half:]
            data = res.tobytes()
        arr = np.frombuffer(data, dtype="<f4").reshape(nlines, w)
        out[y - y0:y - y0 + nlines] = arr
    return out


def _undo_predictor(tmp):
    # OpenEXR zip predictor: d[i] = d[i-1] + d[i] - 128, bytewise, first byte untouched
    acc = np.cumsum(tmp.astype(np.int64))
    idx = np.arange(tmp.size, dtype=np.int64)
    return ((acc - 128 * idx) % 256).astype(np.uint8)


def load_lut(ref, stem, depth):
    lut_dir = os.path.join(ref, "resource", "lut")
    if depth == 1:
        files = [os.path.join(lut_dir, f"{stem}.exr")]
    else:
        files = [os.path.join(lut_dir, f"{stem}_{z}.exr") for z in range(depth)]
    slices = [read_exr_y_float(f) for f in files]
    return np.stack(slices, axis=0), files


def tinyexr_decode(exe, path):
    with tempfile.NamedTemporaryFile(suffix=".f32") as tf:
        subprocess.check_call([exe, path, tf.name])
        d = open(tf.name, "rb").read()
    w, h = struct.unpack("<ii", d[:8])
    return np.frombuffer(d[8:], dtype="<f4").reshape(h, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "platinum_amd", "data", "ggx_luts.bin"))
    ap.add_argument("--check-tinyexr", default=None, help="path to oracle/_ref/exr2raw")
    args = ap.parse_args()

    header = [MAGIC, struct.pack("<I", len(LUTS))]
    blobs = []
    off = 0
    nfiles = 0
    for stem, depth in LUTS:
        vol, files = load_lut(args.ref, stem, depth)
        nfiles += len(files)
        if args.check_tinyexr:
            for z, f in enumerate(files):
                ref_img = tinyexr_decode(args.check_tinyexr, f)
                assert ref_img.shape == vol[z].shape
                assert np.array_equal(ref_img.view(np.uint32), vol[z].view(np.uint32)), f"mismatch vs tinyexr: {f}"
        d, h, w = vol.shape
        header.append(struct.pack("<4I", w, h, d, off))
        blobs.append(np.ascontiguousarray(vol, dtype="<f4").tobytes())
        off += vol.size
        print(f"{stem:22s} {w}x{h}x{d}  min {vol.min():.6f} max {vol.max():.6f} mean {vol.mean():.6f} first {vol.flat[0]:.6f}")
    blob = b"".join(header) + b"".join(blobs)
    out = os.path.abspath(args.out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "wb").write(blob)
    print(f"{nfiles} EXR files -> {out}: {len(blob)} bytes, sha256 {hashlib.sha256(blob).hexdigest()}"
          + ("  (bit-identical to tinyexr decode)" if args.check_tinyexr else ""))


if __name__ == "__main__":
    sys.exit(main())
