// scene_image.cpp — HDR image files for the scene environment: what the reference's TextureLoader reads for
// TextureType::HDR (/root/reference/src/loaders/texture.cpp:86-103): OpenEXR through tinyexr's LoadEXR, anything else
// through stb_image's stbi_loadf (Radiance .hdr).  Both are third-party code vendored by the reference (deps/tinyexr,
// deps/stb_image); this file restates the parts of their behaviour an environment map needs:
//   EXR   single-part scanline files, NONE / RLE / ZIPS / ZIP compression, HALF / FLOAT / UINT channels, data-window offsets,
//         either line order; RGBA assembled as LoadEXR does (missing A = 1, a single channel is replicated to all four).
//         PIZ / PXR24 / B44 / DWA and tiled or multi-part files are a loud error.
//   HDR   "#?RADIANCE" / "#?RGBE", -Y H +X W, flat and new-style RLE scanlines; float = mantissa * 2^(e - 136), alpha 1.
// Pinned in tests/test_scene_ingestion.py against files written by an independent Python encoder and, where oracle/_ref was
// built, against the reference's own tinyexr (oracle/_ref/exr2raw).
#include "scene_io.h"

#include <zlib.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace ptio {

[[noreturn]] static void fail(const std::string& m) { throw std::runtime_error(m); }

static std::string read_all(const std::string& path) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) fail("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {  // subnormal: normalise
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
  else bits = sign | (exp + 112u) << 23 | man << 13;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

namespace {
struct Channel { std::string name; int type; int xs, ys; };  // type: 0 UINT, 1 HALF, 2 FLOAT

// OpenEXR's post-inflate reconstruction: running-sum predictor, then the two interleaved halves
void exr_unpredict(std::vector<uint8_t>& buf) {
  for (size_t i = 1; i < buf.size(); i++) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
  std::vector<uint8_t> out(buf.size());
  const size_t half = (buf.size() + 1) / 2;
  for (size_t i = 0, a = 0, b = half; i < buf.size();) {
    out[i++] = buf[a++];
    if (i < buf.size()) out[i++] = buf[b++];
  }
  buf.swap(out);
}
bool exr_unrle(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  size_t o = 0, i = 0;
  while (i < n) {
    const int8_t c = (int8_t)in[i++];
    if (c < 0) {  // -c literal bytes
      const size_t k = (size_t)(-(int)c);
      if (i + k > n || o + k > out.size()) return false;
      memcpy(&out[o], &in[i], k); i += k; o += k;
    } else {  // c + 1 copies of the next byte
      const size_t k = (size_t)c + 1;
      if (i >= n || o + k > out.size()) return false;
      memset(&out[o], in[i++], k); o += k;
    }
  }
  return o == out.size();
}
}  // namespace

std::vector<float> read_exr_rgba(const std::string& path, uint32_t* w_out, uint32_t* h_out) {
  const std::string file = read_all(path);
  const uint8_t* b = (const uint8_t*)file.data();
  const size_t n = file.size();
  auto u32 = [&](size_t p) { if (p + 4 > n) fail("exr: truncated"); uint32_t v; memcpy(&v, b + p, 4); return v; };
  auto i32 = [&](size_t p) { return (int32_t)u32(p); };
  if (n < 8 || u32(0) != 20000630u) fail("exr: bad magic");
  const uint32_t version = u32(4);
  if ((version & 0xffu) != 2) fail("exr: unsupported version");
  if (version & 0x200u) fail("exr: tiled files are not supported");
  if (version & 0x1800u) fail("exr: deep / multi-part files are not supported");
  size_t p = 8;
  auto cstr = [&](size_t& q) { std::string s; while (q < n && b[q]) s += (char)b[q++]; if (q >= n) fail("exr: truncated header"); q++; return s; };
  std::vector<Channel> channels;
  int compression = -1, line_order = 0;
  int dw[4] = {0, 0, -1, -1};
  for (;;) {
    const std::string name = cstr(p);
    if (name.empty()) break;
    const std::string type = cstr(p);
    const uint32_t size = u32(p);
    p += 4;
    if (p + size > n) fail("exr: truncated attribute");
    if (name == "channels") {
      size_t q = p;
      while (q < p + size && b[q]) {
        Channel c;
        c.name = cstr(q);
        c.type = i32(q); c.xs = i32(q + 8); c.ys = i32(q + 12);
        q += 16;
        if (c.xs != 1 || c.ys != 1) fail("exr: subsampled channels are not supported");
        if (c.type < 0 || c.type > 2) fail("exr: bad pixel type");
        channels.push_back(c);
      }
    } else if (name == "compression") compression = b[p];
    else if (name == "dataWindow") for (int k = 0; k < 4; k++) dw[k] = i32(p + 4 * (size_t)k);
    else if (name == "lineOrder") line_order = b[p];
    p += size;
  }
  if (channels.empty() || compression < 0 || dw[2] < dw[0] || dw[3] < dw[1]) fail("exr: incomplete header");
  if (compression > 3) fail("exr: compression " + std::to_string(compression) + " (PIZ / PXR24 / B44 / DWA) is not supported: re-save as ZIP");
  if (line_order > 1) fail("exr: random-y line order is not supported");
  const int64_t W = (int64_t)dw[2] - dw[0] + 1, H = (int64_t)dw[3] - dw[1] + 1;
  if (W <= 0 || H <= 0 || W > 65536 || H > 65536) fail("exr: bad data window");
  const int lines_per_block = compression == 3 ? 16 : 1;
  const size_t nblocks = (size_t)((H + lines_per_block - 1) / lines_per_block);
  size_t bytes_per_line = 0;
  std::vector<size_t> ch_off(channels.size());
  for (size_t c = 0; c < channels.size(); c++) { ch_off[c] = bytes_per_line; bytes_per_line += (size_t)W * (channels[c].type == 1 ? 2 : 4); }
  // RGBA assembly as tinyexr's LoadEXR: channels by name; one channel alone fills all four
  int idx[4] = {-1, -1, -1, -1};
  for (size_t c = 0; c < channels.size(); c++) {
    const std::string& nm = channels[c].name;
    if (nm == "R") idx[0] = (int)c; else if (nm == "G") idx[1] = (int)c; else if (nm == "B") idx[2] = (int)c; else if (nm == "A") idx[3] = (int)c;
  }
  const bool single = channels.size() == 1;
  if (!single && (idx[0] < 0 || idx[1] < 0 || idx[2] < 0)) fail("exr: R, G or B channel not found");
  std::vector<float> out((size_t)W * H * 4);
  const size_t table = p;
  if (table + 8 * nblocks > n) fail("exr: truncated offset table");
  std::vector<uint8_t> raw;
  for (size_t blk = 0; blk < nblocks; blk++) {
    uint64_t off;
    memcpy(&off, b + table + 8 * blk, 8);
    if (off + 8 > n) fail("exr: block offset out of range");
    const int y0 = i32((size_t)off) - dw[1];
    const uint32_t csize = u32((size_t)off + 4);
    if (off + 8 + (uint64_t)csize > n) fail("exr: truncated block");
    if (y0 < 0 || y0 >= H) fail("exr: block outside the data window");
    const int lines = (int)std::min<int64_t>(lines_per_block, H - y0);
    raw.assign(bytes_per_line * (size_t)lines, 0);
    const uint8_t* src = b + off + 8;
    if (csize == raw.size()) memcpy(raw.data(), src, raw.size());  // stored uncompressed (NONE, or when compression did not pay)
    else if (compression == 1) { if (!exr_unrle(src, csize, raw)) fail("exr: bad RLE data"); exr_unpredict(raw); }
    else if (compression == 2 || compression == 3) {
      uLongf len = (uLongf)raw.size();
      if (uncompress(raw.data(), &len, src, csize) != Z_OK || len != raw.size()) fail("exr: bad zlib data");
      exr_unpredict(raw);
    } else fail("exr: block size does not match an uncompressed file");
    for (int l = 0; l < lines; l++) {
      const uint8_t* line = &raw[bytes_per_line * (size_t)l];
      float* dst = &out[(size_t)(y0 + l) * W * 4];
      auto value = [&](int c, int64_t x) -> float {
        const uint8_t* q = line + ch_off[(size_t)c];
        switch (channels[(size_t)c].type) {
          case 1: { uint16_t h; memcpy(&h, q + 2 * x, 2); return half_to_float(h); }
          case 2: { float f; memcpy(&f, q + 4 * x, 4); return f; }
          default: { uint32_t u; memcpy(&u, q + 4 * x, 4); return (float)u; }
        }
      };
      for (int64_t x = 0; x < W; x++) {
        if (single) { const float v = value(0, x); dst[4 * x] = dst[4 * x + 1] = dst[4 * x + 2] = dst[4 * x + 3] = v; }
        else {
          dst[4 * x] = value(idx[0], x); dst[4 * x + 1] = value(idx[1], x); dst[4 * x + 2] = value(idx[2], x);
          dst[4 * x + 3] = idx[3] >= 0 ? value(idx[3], x) : 1.0f;
        }
      }
    }
  }
  *w_out = (uint32_t)W; *h_out = (uint32_t)H;
  return out;
}

std::vector<float> read_radiance_hdr_rgba(const std::string& path, uint32_t* w_out, uint32_t* h_out) {
  const std::string file = read_all(path);
  const uint8_t* b = (const uint8_t*)file.data();
  const size_t n = file.size();
  size_t p = 0;
  auto line = [&]() { std::string s; while (p < n && b[p] != '\n') s += (char)b[p++]; p++; return s; };
  const std::string magic = line();
  if (magic != "#?RADIANCE" && magic != "#?RGBE") fail("hdr: not a Radiance file");
  bool format_ok = false;
  for (;;) {
    if (p >= n) fail("hdr: truncated header");
    const std::string l = line();
    if (l.empty()) break;
    if (l == "FORMAT=32-bit_rle_rgbe") format_ok = true;
  }
  if (!format_ok) fail("hdr: unsupported format (32-bit_rle_rgbe only)");
  const std::string res = line();
  int H = 0, W = 0;
  if (sscanf(res.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0 || W > 65536 || H > 65536) fail("hdr: unsupported resolution line '" + res + "'");
  std::vector<float> out((size_t)W * H * 4);
  std::vector<uint8_t> scan((size_t)W * 4);
  auto convert = [&](const uint8_t* rgbe, float* dst) {  // stbi__hdr_convert with req_comp = 4
    if (rgbe[3] != 0) {
      const float f1 = ldexpf(1.0f, (int)rgbe[3] - (128 + 8));
      dst[0] = rgbe[0] * f1; dst[1] = rgbe[1] * f1; dst[2] = rgbe[2] * f1;
    } else dst[0] = dst[1] = dst[2] = 0.0f;
    dst[3] = 1.0f;
  };
  for (int y = 0; y < H; y++) {
    if (p + 4 > n) fail("hdr: truncated data");
    const bool rle = W >= 8 && W < 32768 && b[p] == 2 && b[p + 1] == 2 && !(b[p + 2] & 0x80) && ((int)b[p + 2] << 8 | b[p + 3]) == W;
    if (!rle) {  // flat scanline
      if (p + (size_t)W * 4 > n) fail("hdr: truncated data");
      memcpy(scan.data(), b + p, (size_t)W * 4);
      p += (size_t)W * 4;
    } else {
      p += 4;
      for (int c = 0; c < 4; c++) {
        int x = 0;
        while (x < W) {
          if (p >= n) fail("hdr: truncated data");
          int count = b[p++];
          if (count > 128) {  // run
            count -= 128;
            if (p >= n || x + count > W) fail("hdr: corrupt RLE data");
            const uint8_t v = b[p++];
            for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = v;
          } else {  // literal
            if (count == 0 || p + (size_t)count > n || x + count > W) fail("hdr: corrupt RLE data");
            for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = b[p++];
          }
        }
      }
    }
    for (int x = 0; x < W; x++) convert(&scan[(size_t)x * 4], &out[((size_t)y * W + x) * 4]);
  }
  *w_out = (uint32_t)W; *h_out = (uint32_t)H;
  return out;
}

}  // namespace ptio
