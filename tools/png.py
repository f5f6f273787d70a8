"""Tiny PNG writer (zlib only) for eyeballing accumulators: sRGB-ish display of a float RGBA image."""
import struct, zlib
import numpy as np

def write_png(path, img, exposure=1.0):
    rgb = np.clip(np.asarray(img)[..., :3] * exposure, 0, None)
    rgb = rgb / (1.0 + rgb)                      # Reinhard, just for viewing
    rgb = (np.clip(rgb, 0, 1) ** (1 / 2.2) * 255 + 0.5).astype(np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                           + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
