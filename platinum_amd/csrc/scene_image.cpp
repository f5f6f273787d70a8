// scene_image.cpp — HDR image files for the scene environment: what the reference's TextureLoader reads for
// TextureType::HDR (/root/reference/src/loaders/texture.cpp:86-103): OpenEXR through tinyexr's LoadEXR, anything else
// through stb_image's stbi_loadf (Radiance .hdr).  Both are third-party code vendored by the reference (deps/tinyexr,
// deps/stb_image); this file restates the parts of their behaviour an environment map needs:
//   EXR   single-part scanline and tiled files (of a multi-resolution file the full-size level, the one LoadEXR assembles),
//         NONE / RLE / ZIPS / ZIP / PIZ compression, HALF / FLOAT / UINT channels, data-window offsets, either line order;
//         RGBA assembled as LoadEXR does (missing A = 1, a single channel is replicated to all four).
//         PXR24 / B44 / DWA and multi-part or deep files are a loud error.
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

// ---- PIZ (OpenEXR ImfPizCompressor; tinyexr.h DecompressPiz): bitmap of the 16-bit values in use, a Huffman-coded stream of the
// range-compressed values (canonical codes of <= 58 bits, a run-length symbol), and a 2-D Haar-like wavelet per channel ----
struct BitReader {
  const uint8_t* p; size_t nbits, pos = 0;
  bool ok = true;
  uint32_t bit() { if (pos >= nbits) { ok = false; return 0; } const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u; pos++; return b; }
  uint32_t bits(int n) { uint32_t v = 0; for (int i = 0; i < n; i++) v = v << 1 | bit(); return v; }
};

// hufUncompress: 20-byte header (im, iM, table length, nBits, reserved), the packed code-length table, the bit stream.
bool piz_huf_uncompress(const uint8_t* in, size_t n, std::vector<uint16_t>& out) {
  if (n < 20) return false;
  auto rd = [&](size_t o) { uint32_t v; memcpy(&v, in + o, 4); return v; };
  const uint32_t im = rd(0), iM = rd(4), nbits = rd(12);
  constexpr uint32_t kEnc = (1u << 16) + 1;
  if (im >= kEnc || iM >= kEnc || im > iM) return false;
  // code lengths: 6 bits each; 59..62 = a run of 2..5 zero lengths, 63 = 8 more bits: a run of 6..261
  std::vector<uint8_t> len(kEnc, 0);
  BitReader tb{in + 20, (n - 20) * 8};
  for (uint32_t i = im; i <= iM; i++) {
    const uint32_t l = tb.bits(6);
    if (!tb.ok) return false;
    if (l == 63) { const uint32_t run = tb.bits(8) + 6; if (i + run > iM + 1) return false; i += run - 1; }
    else if (l >= 59) { const uint32_t run = l - 59 + 2; if (i + run > iM + 1) return false; i += run - 1; }
    else len[i] = (uint8_t)l;
  }
  if (!tb.ok) return false;
  const size_t table_bytes = (tb.pos + 7) / 8;
  const uint8_t* data = in + 20 + table_bytes;
  if ((size_t)nbits > (n - 20 - table_bytes) * 8) return false;
  // canonical codes (hufCanonicalCodeTable): first code of each length from the longest up; within a length, by symbol value
  uint64_t count[59] = {0}, first[59] = {0};
  for (uint32_t i = 0; i < kEnc; i++) count[len[i]]++;
  {
    uint64_t c = 0;
    for (int l = 58; l > 0; l--) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; }
  }
  // A length table that is not a prefix code (over-subscribed: more codes of length l than l bits can hold) must be rejected
  // here — tinyexr's hufBuildDecTable does ("code needs more than l bits") — or the table fill below would leave its bounds.
  for (int l = 1; l <= 58; l++)
    if (count[l] && ((first[l] + count[l] - 1) >> l) != 0) return false;
  std::vector<uint32_t> offset(60, 0), sorted;
  sorted.reserve(kEnc);
  for (int l = 1; l <= 58; l++) {
    offset[l] = (uint32_t)sorted.size();
    if (count[l]) for (uint32_t i = im; i <= iM; i++) if (len[i] == l) sorted.push_back(i);
  }
  // a 12-bit table for the short codes, bit-serial continuation for the long ones
  constexpr int kFast = 12;
  std::vector<uint32_t> fast(1u << kFast, 0);  // (symbol << 6) | length, 0 = none
  for (int l = 1; l <= kFast; l++)
    for (uint64_t k = 0; k < count[l]; k++) {
      const uint64_t code = first[l] + k;
      const uint32_t sym = sorted[offset[l] + (uint32_t)k];
      const size_t base = (size_t)(code << (kFast - l));
      if (base + (1u << (kFast - l)) > fast.size()) return false;  // cannot happen after the prefix-code check; kept as a guard
      for (uint32_t fill = 0; fill < (1u << (kFast - l)); fill++) fast[base | fill] = sym << 6 | (uint32_t)l;
    }
  BitReader br{data, nbits};
  size_t o = 0;
  const uint32_t rlc = iM;
  while (br.pos < br.nbits) {
    uint32_t sym = ~0u;
    if (br.nbits - br.pos >= (size_t)kFast) {
      uint32_t peek = 0;
      for (int i = 0; i < kFast; i++) { const size_t q = br.pos + i; peek = peek << 1 | ((br.p[q >> 3] >> (7 - (q & 7))) & 1u); }
      const uint32_t e = fast[peek];
      if (e) { sym = e >> 6; br.pos += e & 63u; }
    }
    if (sym == ~0u) {
      uint64_t code = 0;
      for (int l = 1; l <= 58 && sym == ~0u; l++) {
        code = code << 1 | br.bit();
        if (!br.ok) return false;
        if (count[l] && code >= first[l] && code - first[l] < count[l]) sym = sorted[offset[l] + (uint32_t)(code - first[l])];
      }
      if (sym == ~0u) return false;
    }
    if (sym == rlc) {
      const uint32_t run = br.bits(8);
      if (!br.ok || o == 0 || o + run > out.size()) return false;
      for (uint32_t k = 0; k < run; k++, o++) out[o] = out[o - 1];
    } else {
      if (o >= out.size()) return false;
      out[o++] = (uint16_t)sym;
    }
  }
  return o == out.size();
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int ls = (int16_t)l, hs = (int16_t)h;
  const int ai = ls + (hs & 1) + (hs >> 1);
  a = (uint16_t)(int16_t)ai;
  b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - 0x8000) & 0xffff;
  b = (uint16_t)bb;
  a = (uint16_t)aa;
}
// wav2Decode: in-place inverse transform of an nx x ny plane with element strides ox / oy
void piz_wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1 << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1;
  while (p <= n) p <<= 1;
  p >>= 1;
  int p2 = p;
  p >>= 1;
  auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
  while (p >= 1) {
    uint16_t* py = in;
    uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
    const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
    uint16_t i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      uint16_t* px = py;
      uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
        dec(*px, *p10, i00, i10);
        dec(*p01, *p11, i01, i11);
        dec(i00, i01, *px, *p01);
        dec(i10, i11, *p10, *p11);
      }
      if (nx & p) {  // odd column
        uint16_t* p10 = px + oy1;
        dec(*px, *p10, i00, *p10);
        *px = i00;
      }
    }
    if (ny & p) {  // odd line
      uint16_t* px = py;
      uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        dec(*px, *p01, i00, *p01);
        *px = i00;
      }
    }
    p2 = p;
    p >>= 1;
  }
}

// One PIZ block -> the raw block layout (per scanline, channels in file order, planar)
bool piz_decompress(const uint8_t* in, size_t n, const std::vector<Channel>& channels, int width, int lines, std::vector<uint8_t>& raw) {
  if (n < 4) return false;
  uint16_t min_nz, max_nz;
  memcpy(&min_nz, in, 2); memcpy(&max_nz, in + 2, 2);
  constexpr size_t kBitmap = 65536 / 8;
  std::vector<uint8_t> bitmap(kBitmap, 0);
  size_t p = 4;
  if (max_nz >= kBitmap) return false;
  if (min_nz <= max_nz) {
    const size_t len = (size_t)max_nz - min_nz + 1;
    if (p + len > n) return false;
    memcpy(&bitmap[min_nz], in + p, len);
    p += len;
  } else if (!(min_nz == kBitmap - 1 && max_nz == 0)) return false;  // (all-zero block)
  std::vector<uint16_t> lut(65536, 0);
  uint32_t k = 0;
  for (uint32_t i = 0; i < 65536; i++) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
  const uint16_t max_value = (uint16_t)(k - 1);
  if (p + 4 > n) return false;
  int32_t length;
  memcpy(&length, in + p, 4);
  p += 4;
  if (length < 0 || p + (size_t)length > n) return false;
  std::vector<uint16_t> tmp(raw.size() / 2);
  if (!piz_huf_uncompress(in + p, (size_t)length, tmp)) return false;
  std::vector<size_t> start(channels.size());
  size_t off = 0;
  for (size_t c = 0; c < channels.size(); c++) {
    const int size = channels[c].type == 1 ? 1 : 2;  // 16-bit words per pixel
    start[c] = off;
    for (int j = 0; j < size; j++) piz_wav2_decode(&tmp[off + (size_t)j], width, size, lines, width * size, max_value);
    off += (size_t)width * lines * size;
  }
  for (auto& v : tmp) v = lut[v];
  uint8_t* dst = raw.data();
  for (int y = 0; y < lines; y++)
    for (size_t c = 0; c < channels.size(); c++) {
      const size_t words = (size_t)width * (channels[c].type == 1 ? 1 : 2);
      memcpy(dst, &tmp[start[c] + (size_t)y * words], words * 2);
      dst += words * 2;
    }
  return true;
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
  const bool tiled = (version & 0x200u) != 0;
  if (version & 0x1800u) fail("exr: deep / multi-part files are not supported");
  size_t p = 8;
  auto cstr = [&](size_t& q) { std::string s; while (q < n && b[q]) s += (char)b[q++]; if (q >= n) fail("exr: truncated header"); q++; return s; };
  std::vector<Channel> channels;
  int compression = -1, line_order = 0;
  uint32_t tile_w = 0, tile_h = 0, tile_mode = 0;
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
    else if (name == "tiles" && size >= 9) { tile_w = u32(p); tile_h = u32(p + 4); tile_mode = b[p + 8]; }
    p += size;
  }
  if (channels.empty() || compression < 0 || dw[2] < dw[0] || dw[3] < dw[1]) fail("exr: incomplete header");
  if (compression > 4) fail("exr: compression " + std::to_string(compression) + " (PXR24 / B44 / DWA) is not supported: re-save as ZIP or PIZ");
  if (line_order > 1) fail("exr: random-y line order is not supported");
  const int64_t W = (int64_t)dw[2] - dw[0] + 1, H = (int64_t)dw[3] - dw[1] + 1;
  if (W <= 0 || H <= 0 || W > 65536 || H > 65536) fail("exr: bad data window");
  if (tiled && (!tile_w || !tile_h || tile_w > 65536 || tile_h > 65536 || (tile_mode & 0xf) > 2)) fail("exr: tiled file without a valid tile description");
  const int lines_per_block = compression == 3 ? 16 : (compression == 4 ? 32 : 1);
  // blocks: scan-line files = groups of lines over the full width; tiled files = the tiles of level (0, 0), which come first in
  // the offset table of increasing- and decreasing-y files (the further levels of MIPMAP / RIPMAP files are not read)
  const size_t tiles_x = tiled ? (size_t)((W + tile_w - 1) / tile_w) : 1, tiles_y = tiled ? (size_t)((H + tile_h - 1) / tile_h) : 0;
  const size_t nblocks = tiled ? tiles_x * tiles_y : (size_t)((H + lines_per_block - 1) / lines_per_block);
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
    int y0, lines;
    int64_t x0 = 0, bw = W;  // the block's columns
    size_t hdr = 8;
    if (tiled) {
      hdr = 20;
      if (off + hdr > n) fail("exr: block offset out of range");
      const int tx = i32((size_t)off), ty = i32((size_t)off + 4);
      if (i32((size_t)off + 8) != 0 || i32((size_t)off + 12) != 0) fail("exr: the offset table does not start with the full-resolution tiles");
      if (tx < 0 || ty < 0 || (size_t)tx >= tiles_x || (size_t)ty >= tiles_y) fail("exr: tile outside the data window");
      x0 = (int64_t)tx * tile_w; bw = std::min<int64_t>(tile_w, W - x0);
      y0 = (int)((int64_t)ty * tile_h); lines = (int)std::min<int64_t>(tile_h, H - y0);
    } else {
      y0 = i32((size_t)off) - dw[1];
      if (y0 < 0 || y0 >= H) fail("exr: block outside the data window");
      lines = (int)std::min<int64_t>(lines_per_block, H - y0);
    }
    const uint32_t csize = u32((size_t)off + hdr - 4);
    if (off + hdr + (uint64_t)csize > n) fail("exr: truncated block");
    size_t bytes_per_line = 0;
    std::vector<size_t> ch_off(channels.size());
    for (size_t c = 0; c < channels.size(); c++) { ch_off[c] = bytes_per_line; bytes_per_line += (size_t)bw * (channels[c].type == 1 ? 2 : 4); }
    raw.assign(bytes_per_line * (size_t)lines, 0);
    const uint8_t* src = b + off + hdr;
    if (csize == raw.size()) memcpy(raw.data(), src, raw.size());  // stored uncompressed (NONE, or when compression did not pay)
    else if (compression == 1) { if (!exr_unrle(src, csize, raw)) fail("exr: bad RLE data"); exr_unpredict(raw); }
    else if (compression == 2 || compression == 3) {
      uLongf len = (uLongf)raw.size();
      if (uncompress(raw.data(), &len, src, csize) != Z_OK || len != raw.size()) fail("exr: bad zlib data");
      exr_unpredict(raw);
    } else if (compression == 4) {
      if (!piz_decompress(src, csize, channels, (int)bw, lines, raw)) fail("exr: bad PIZ data");
    } else fail("exr: block size does not match an uncompressed file");
    for (int l = 0; l < lines; l++) {
      const uint8_t* line = &raw[bytes_per_line * (size_t)l];
      float* dst = &out[((size_t)(y0 + l) * W + (size_t)x0) * 4];
      auto value = [&](int c, int64_t x) -> float {
        const uint8_t* q = line + ch_off[(size_t)c];
        switch (channels[(size_t)c].type) {
          case 1: { uint16_t h; memcpy(&h, q + 2 * x, 2); return half_to_float(h); }
          case 2: { float f; memcpy(&f, q + 4 * x, 4); return f; }
          default: { uint32_t u; memcpy(&u, q + 4 * x, 4); return (float)u; }
        }
      };
      for (int64_t x = 0; x < bw; x++) {
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
