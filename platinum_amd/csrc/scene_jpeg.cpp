// scene_jpeg.cpp — JPEG textures for the glTF importer: what the reference gets from
// stbi_load / stbi_load_from_memory(..., 4) (/root/reference/src/loaders/texture.cpp:101-119, deps/stb_image v2.30) for a
// .jpg image: baseline and progressive Huffman JPEG, 8-bit, 1 / 3 / 4 components, any sampling factors, restart intervals,
// expanded to RGBA8 with alpha 255.
//
// The entropy decoding and the frame structure follow ITU-T T.81.  What makes two decoders produce the same BYTES is the
// arithmetic after the coefficients, and that is stb_image's (restated here, checked bit-for-bit against the reference's
// own copy compiled from where it lies — oracle/_ref/stbi2raw, tests/test_scene_ingestion.py):
//   * coefficients are dequantised as 16-bit values; the inverse DCT is the 12-bit fixed-point "islow" one of the IJG code
//     with two extra fraction bits kept between the column and the row pass, a DC-only column short-cut, and rounding
//     constants 512 and 65536 + (128 << 17) (stb_image.h:2424-2519);
//   * chroma upsampling: h2v1 and h2v2 triangle filters ((3a + b + 2) >> 2, (3·(3a + b) + (3c + d) + 8) >> 4), h1v2
//     (3a + b + 2) >> 2, anything else nearest (stb_image.h:3465-3657);
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits first and the Cb term of green masked to
//     its upper 16 bits (stb_image.h:3660-3684);
//   * RGB when the component ids are 'R','G','B' or an Adobe APP14 marker says transform 0 and there is no JFIF header;
//     CMYK / YCCK through (x · k + 128 + ((x · k + 128) >> 8)) >> 8 (stb_image.h:3859-3975).
// Arithmetic-coded, lossless and 12-bit files are a loud error (stb_image rejects them too).
#include "scene_io.h"

#include <cstring>
#include <stdexcept>

namespace ptio {

namespace {

[[noreturn]] void bad(const std::string& m) { throw std::runtime_error("jpeg: " + m); }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {  // canonical code of T.81 Annex C, decoded through a 9-bit table + a bit-serial tail
  bool present = false;
  uint8_t counts[17] = {0};
  uint8_t symbols[256] = {0};
  int32_t mincode[18], maxcode[18], valptr[18];  // per code length
  uint16_t fast[512];                            // (length << 8) | symbol, 0 = longer than 9 bits
  void build() {
    int code = 0, k = 0;
    memset(fast, 0, sizeof(fast));
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      for (int i = 0; i < counts[l]; i++, k++, code++) {
        if (code >= (1 << l)) bad("bad code lengths");
        if (l <= 9) {
          const int lo = code << (9 - l);
          for (int f = 0; f < (1 << (9 - l)); f++) fast[lo + f] = (uint16_t)((l << 8) | symbols[k]);
        }
      }
      maxcode[l] = counts[l] ? code - 1 : -1;
      code <<= 1;
    }
    present = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int hd = 0, ha = 0;           // DC / AC table selectors of the current scan
  int dc_pred = 0;
  int x = 0, y = 0;             // size in samples
  int w2 = 0, h2 = 0;           // size padded to whole MCUs
  std::vector<uint8_t> data;    // w2 * h2 samples after the IDCT
  std::vector<int16_t> coeff;   // progressive: (w2 / 8) * (h2 / 8) blocks of 64
};

struct Decoder {
  const uint8_t* p;
  const uint8_t* end;
  // entropy-coded segment reader
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  int marker = -1;  // a marker met inside the entropy-coded data (0xFFxx), -1 = none
  bool nomore = false;

  uint16_t dequant[4][64];
  bool have_dequant[4] = {false, false, false, false};
  Huffman dc[4], ac[4];
  std::vector<Component> comps;
  int width = 0, height = 0, hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
  bool progressive = false, jfif = false;
  int app14_transform = -1, rgb_ids = 0;
  int restart_interval = 0, todo = 0;
  int scan_n = 0, order[4] = {0, 0, 0, 0};
  int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;

  int get8() { if (p >= end) return 0; return *p++; }
  int get16() { const int a = get8(); return (a << 8) | get8(); }
  void skip(int n) { if (n < 0 || n > end - p) { p = end; return; } p += n; }

  // ---- bit reader: bytes are consumed until a marker shows up; from then on zeros are fed (as stb_image does) ----
  void fill() {
    while (bitcnt <= 24) {
      int b = nomore ? 0 : get8();
      if (b == 0xff && !nomore) {
        int c = get8();
        while (c == 0xff) c = get8();  // fill bytes
        if (c != 0) { marker = c; nomore = true; b = 0; }
      }
      bitbuf |= (uint32_t)b << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  int bits(int n) {  // n <= 16
    if (n == 0) return 0;
    if (bitcnt < n) fill();
    const int v = (int)(bitbuf >> (32 - n));
    bitbuf <<= n;
    bitcnt -= n;
    return v;
  }
  int bit() { return bits(1); }
  int receive_extend(int n) {  // T.81 F.2.2.1
    if (n == 0) return 0;
    const int v = bits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  int decode(const Huffman& h) {
    if (bitcnt < 16) fill();
    const uint16_t f = h.fast[bitbuf >> 23];
    if (f) {
      const int l = f >> 8;
      bitbuf <<= l;
      bitcnt -= l;
      return f & 0xff;
    }
    int code = (int)(bitbuf >> 22);  // 10 bits
    for (int l = 10; l <= 16; l++) {
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) {
        bitbuf <<= l;
        bitcnt -= l;
        return h.symbols[h.valptr[l] + code - h.mincode[l]];
      }
      code = (int)(bitbuf >> (32 - (l + 1)));
    }
    bad("bad huffman code");
  }
  void reset_entropy() {
    bitbuf = 0; bitcnt = 0; nomore = false; marker = -1;
    for (auto& c : comps) c.dc_pred = 0;
    eob_run = 0;
    todo = restart_interval ? restart_interval : 0x7fffffff;
  }

  // ---- markers ----
  void read_dqt() {
    int len = get16() - 2;
    while (len > 0) {
      const int q = get8(), prec = q >> 4, t = q & 15;
      if (prec > 1 || t > 3) bad("bad DQT");
      for (int i = 0; i < 64; i++) dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
      have_dequant[t] = true;
      len -= prec ? 129 : 65;
    }
    if (len != 0) bad("bad DQT length");
  }
  void read_dht() {
    int len = get16() - 2;
    while (len > 0) {
      const int q = get8(), tc = q >> 4, th = q & 15;
      if (tc > 1 || th > 3) bad("bad DHT");
      Huffman& h = tc ? ac[th] : dc[th];
      int n = 0;
      for (int l = 1; l <= 16; l++) { h.counts[l] = (uint8_t)get8(); n += h.counts[l]; }
      if (n > 256) bad("bad DHT");
      for (int i = 0; i < n; i++) h.symbols[i] = (uint8_t)get8();
      h.build();
      len -= 17 + n;
    }
    if (len != 0) bad("bad DHT length");
  }
  void read_sof(int m) {
    progressive = m == 0xc2;
    const int len = get16();
    if (get8() != 8) bad("only 8-bit samples are supported");
    height = get16(); width = get16();
    const int n = get8();
    if (height == 0 || width == 0) bad("empty image");
    if (n != 1 && n != 3 && n != 4) bad("bad component count");
    if (len != 8 + 3 * n) bad("bad SOF length");
    comps.assign((size_t)n, Component());
    static const char rgb[3] = {'R', 'G', 'B'};
    rgb_ids = 0;
    for (int i = 0; i < n; i++) {
      Component& c = comps[(size_t)i];
      c.id = get8();
      if (n == 3 && c.id == rgb[i]) rgb_ids++;
      const int q = get8();
      c.h = q >> 4; c.v = q & 15; c.tq = get8();
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad sampling factors");
      hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
    }
    for (auto& c : comps) if (hmax % c.h || vmax % c.v) bad("bad sampling factors");
    mcu_w = hmax * 8; mcu_h = vmax * 8;
    mcu_x = (width + mcu_w - 1) / mcu_w; mcu_y = (height + mcu_h - 1) / mcu_h;
    for (auto& c : comps) {
      c.x = (width * c.h + hmax - 1) / hmax;
      c.y = (height * c.v + vmax - 1) / vmax;
      c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
      c.data.assign((size_t)c.w2 * c.h2, 0);
      if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
    }
  }
  void read_sos() {
    const int len = get16();
    scan_n = get8();
    if (scan_n < 1 || scan_n > 4 || scan_n > (int)comps.size()) bad("bad SOS component count");
    if (len != 6 + 2 * scan_n) bad("bad SOS length");
    for (int i = 0; i < scan_n; i++) {
      const int id = get8(), q = get8();
      int which = -1;
      for (size_t k = 0; k < comps.size(); k++) if (comps[k].id == id) which = (int)k;
      if (which < 0) bad("SOS names an unknown component");
      comps[(size_t)which].hd = q >> 4; comps[(size_t)which].ha = q & 15;
      if (comps[(size_t)which].hd > 3 || comps[(size_t)which].ha > 3) bad("bad table selector");
      order[i] = which;
    }
    spec_start = get8(); spec_end = get8();
    const int a = get8();
    succ_high = a >> 4; succ_low = a & 15;
    if (progressive) {
      if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) bad("bad progressive scan");
    } else {
      if (spec_start != 0 || succ_high != 0 || succ_low != 0) bad("bad baseline scan");
      spec_end = 63;
    }
  }

  // ---- blocks ----
  void block_baseline(int16_t out[64], Component& c) {
    const Huffman &hd = dc[c.hd], &ha = ac[c.ha];
    if (!hd.present || !ha.present || !have_dequant[c.tq]) bad("scan uses a table that was never defined");
    const uint16_t* dq = dequant[c.tq];
    memset(out, 0, 64 * sizeof(int16_t));
    const int t = decode(hd);
    if (t > 15) bad("bad DC size");
    c.dc_pred += receive_extend(t);
    out[0] = (int16_t)(c.dc_pred * dq[0]);
    for (int k = 1; k < 64;) {
      const int rs = decode(ha), r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (rs != 0xf0) break;  // end of block
        k += 16;
      } else {
        k += r;
        if (k > 63) bad("bad AC run");
        const int z = kZigzag[k++];
        out[z] = (int16_t)(receive_extend(s) * dq[z]);
      }
    }
  }
  void block_prog_dc(int16_t* data, Component& c) {
    if (spec_end != 0) bad("DC scan with AC coefficients");
    if (succ_high == 0) {
      const Huffman& hd = dc[c.hd];
      if (!hd.present) bad("scan uses a table that was never defined");
      memset(data, 0, 64 * sizeof(int16_t));
      const int t = decode(hd);
      if (t > 15) bad("bad DC size");
      c.dc_pred += receive_extend(t);
      data[0] = (int16_t)(c.dc_pred * (1 << succ_low));
    } else if (bit()) {
      data[0] = (int16_t)(data[0] + (1 << succ_low));
    }
  }
  void block_prog_ac(int16_t* data, Component& c) {
    if (spec_start == 0) bad("AC scan that starts at the DC coefficient");
    const Huffman& ha = ac[c.ha];
    if (!ha.present) bad("scan uses a table that was never defined");
    if (succ_high == 0) {  // first pass of this band (T.81 G.1.2.2)
      const int shift = succ_low;
      if (eob_run) { eob_run--; return; }
      for (int k = spec_start; k <= spec_end;) {
        const int rs = decode(ha), r = rs >> 4, s = rs & 15;
        if (s == 0) {
          if (r < 15) {
            eob_run = (1 << r);
            if (r) eob_run += bits(r);
            eob_run--;
            break;
          }
          k += 16;
        } else {
          k += r;
          if (k > 63) bad("bad AC run");
          data[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << shift));
        }
      }
    } else {  // refinement (G.1.2.3)
      const int16_t bitv = (int16_t)(1 << succ_low);
      auto refine = [&](int16_t* q) {
        if (*q != 0 && bit() && (*q & bitv) == 0) *q = (int16_t)(*q > 0 ? *q + bitv : *q - bitv);
      };
      if (eob_run) {
        eob_run--;
        for (int k = spec_start; k <= spec_end; k++) refine(&data[kZigzag[k]]);
        return;
      }
      int k = spec_start;
      do {
        const int rs = decode(ha);
        int r = rs >> 4, s = rs & 15;
        if (s == 0) {
          if (r < 15) {
            eob_run = (1 << r) - 1;
            if (r) eob_run += bits(r);
            r = 64;  // run to the end of the band, refining what is already non-zero
          }
        } else {
          if (s != 1) bad("bad refinement code");
          s = bit() ? bitv : -bitv;
        }
        while (k <= spec_end) {
          int16_t* q = &data[kZigzag[k++]];
          if (*q != 0) {
            if (bit() && (*q & bitv) == 0) *q = (int16_t)(*q > 0 ? *q + bitv : *q - bitv);
          } else {
            if (r == 0) { *q = (int16_t)s; break; }
            r--;
          }
        }
      } while (k <= spec_end);
    }
  }

  // ---- inverse DCT, stb_image.h:2424-2519 ----
  static uint8_t clamp8(int x) { return (uint8_t)((unsigned)x > 255 ? (x < 0 ? 0 : 255) : x); }
  static void idct(uint8_t* out, int stride, const int16_t d[64]) {
    auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
    static const int c0541 = f2f(0.5411961), cm1847 = f2f(-1.847759065), c0765 = f2f(0.765366865), c1175 = f2f(1.175875602),
                     c0298 = f2f(0.298631336), c2053 = f2f(2.053119869), c3072 = f2f(3.072711026), c1501 = f2f(1.501321110),
                     cm0899 = f2f(-0.899976223), cm2562 = f2f(-2.562915447), cm1961 = f2f(-1.961570560), cm0390 = f2f(-0.390180644);
    struct Odd { int x0, x1, x2, x3, t0, t1, t2, t3; };
    auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
      Odd r;
      int p2 = s2, p3 = s6;
      int p1 = (p2 + p3) * c0541;
      int t2 = p1 + p3 * cm1847, t3 = p1 + p2 * c0765;
      p2 = s0; p3 = s4;
      int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
      r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
      t0 = s7; t1 = s5; t2 = s3; t3 = s1;
      p3 = t0 + t2;
      int p4 = t1 + t3;
      p1 = t0 + t3; p2 = t1 + t2;
      const int p5 = (p3 + p4) * c1175;
      t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;
      p1 = p5 + p1 * cm0899; p2 = p5 + p2 * cm2562; p3 = p3 * cm1961; p4 = p4 * cm0390;
      r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
      return r;
    };
    int val[64];
    for (int i = 0; i < 8; i++) {  // columns
      const int16_t* c = d + i;
      int* v = val + i;
      if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
        const int dcterm = c[0] * 4;
        v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
      } else {
        Odd r = pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
        r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
        v[0] = (r.x0 + r.t3) >> 10; v[56] = (r.x0 - r.t3) >> 10;
        v[8] = (r.x1 + r.t2) >> 10; v[48] = (r.x1 - r.t2) >> 10;
        v[16] = (r.x2 + r.t1) >> 10; v[40] = (r.x2 - r.t1) >> 10;
        v[24] = (r.x3 + r.t0) >> 10; v[32] = (r.x3 - r.t0) >> 10;
      }
    }
    for (int i = 0; i < 8; i++) {  // rows
      const int* v = val + 8 * i;
      uint8_t* o = out + (size_t)stride * i;
      Odd r = pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      const int k = 65536 + (128 << 17);
      r.x0 += k; r.x1 += k; r.x2 += k; r.x3 += k;
      o[0] = clamp8((r.x0 + r.t3) >> 17); o[7] = clamp8((r.x0 - r.t3) >> 17);
      o[1] = clamp8((r.x1 + r.t2) >> 17); o[6] = clamp8((r.x1 - r.t2) >> 17);
      o[2] = clamp8((r.x2 + r.t1) >> 17); o[5] = clamp8((r.x2 - r.t1) >> 17);
      o[3] = clamp8((r.x3 + r.t0) >> 17); o[4] = clamp8((r.x3 - r.t0) >> 17);
    }
  }

  // ---- one scan ----
  bool restart_due() {  // returns false when the data ended without the expected RSTn (the rest of the scan stays as it is)
    if (--todo > 0) return true;
    if (bitcnt < 24) fill();
    if (marker < 0xd0 || marker > 0xd7) return false;
    reset_entropy();
    return true;
  }
  void scan() {
    reset_entropy();
    int16_t block[64];
    if (scan_n == 1) {  // non-interleaved: the component's own blocks, row by row
      Component& c = comps[(size_t)order[0]];
      const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
      for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++) {
          if (progressive) {
            int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * (c.w2 / 8))];
            if (spec_start == 0) block_prog_dc(d, c); else block_prog_ac(d, c);
          } else {
            block_baseline(block, c);
            idct(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, block);
          }
          if (!restart_due()) return;
        }
      return;
    }
    for (int j = 0; j < mcu_y; j++)  // interleaved MCUs
      for (int i = 0; i < mcu_x; i++) {
        for (int k = 0; k < scan_n; k++) {
          Component& c = comps[(size_t)order[k]];
          for (int y = 0; y < c.v; y++)
            for (int x = 0; x < c.h; x++) {
              const int bx = i * c.h + x, by = j * c.v + y;
              if (progressive) {
                if (spec_start != 0) bad("interleaved AC scan");
                block_prog_dc(&c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 / 8))], c);
              } else {
                block_baseline(block, c);
                idct(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
              }
            }
        }
        if (!restart_due()) return;
      }
  }
  void finish_progressive() {
    for (auto& c : comps) {
      if (!have_dequant[c.tq]) bad("component uses a quantisation table that was never defined");
      const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
      for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++) {
          int16_t* d = &c.coeff[64 * ((size_t)i + (size_t)j * (c.w2 / 8))];
          for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * dequant[c.tq][k]);
          idct(&c.data[(size_t)c.w2 * j * 8 + (size_t)i * 8], c.w2, d);
        }
    }
  }

  void decode_image() {
    if (get8() != 0xff || get8() != 0xd8) bad("not a JPEG (no SOI)");
    bool have_frame = false, done = false;
    int m = -1;
    auto next_marker = [&]() {
      if (marker >= 0) { const int k = marker; marker = -1; return k; }
      int x = get8();
      while (x != 0xff) { if (p >= end) return 0xd9; x = get8(); }  // (junk between segments is skipped)
      while (x == 0xff) x = get8();
      return x;
    };
    while (!done) {
      m = next_marker();
      switch (m) {
        case 0xd9: done = true; break;
        case 0xdb: read_dqt(); break;
        case 0xc4: read_dht(); break;
        case 0xdd: if (get16() != 4) bad("bad DRI"); restart_interval = get16(); break;
        case 0xc0: case 0xc1: case 0xc2:
          if (have_frame) bad("second frame header");
          read_sof(m); have_frame = true; break;
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
          bad("lossless, hierarchical and arithmetic-coded JPEG are not supported");
        case 0xda:
          if (!have_frame) bad("scan before the frame header");
          read_sos();
          scan();
          if (marker < 0) {  // the scan ended at the end of the data or before a marker we have not read yet
            bitcnt = 0;
          }
          break;
        case 0xe0: {  // JFIF?
          const int len = get16();
          if (len >= 7 && end - p >= 5 && memcmp(p, "JFIF\0", 5) == 0) jfif = true;
          skip(len - 2);
          break;
        }
        case 0xee: {  // Adobe: colour transform flag
          const int len = get16();
          if (len >= 14 && end - p >= 12 && memcmp(p, "Adobe\0", 6) == 0) app14_transform = p[11];
          skip(len - 2);
          break;
        }
        default:
          if ((m >= 0xe0 && m <= 0xef) || m == 0xfe || m == 0xdc) { const int len = get16(); if (len < 2) bad("bad segment length"); skip(len - 2); }
          else if (m >= 0xd0 && m <= 0xd7) { /* stray RSTn */ }
          else if (m == 0x01 || m == 0xff || m == 0x00) { /* TEM / fill */ }
          else bad("unknown marker");
      }
      if (p >= end && marker < 0) done = true;
    }
    if (!have_frame) bad("no frame header");
    if (progressive) finish_progressive();
  }

  // ---- upsampling + colour (stb_image.h:3465-3684, 3865-3975) ----
  static uint8_t blinn(uint8_t x, uint8_t y) { const unsigned t = x * y + 128; return (uint8_t)((t + (t >> 8)) >> 8); }
  std::vector<uint8_t> to_rgba() {
    const int n = (int)comps.size();
    const bool is_rgb = n == 3 && (rgb_ids == 3 || (app14_transform == 0 && !jfif));
    std::vector<uint8_t> out((size_t)width * height * 4);
    struct Up { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    std::vector<Up> up((size_t)n);
    for (int k = 0; k < n; k++) {
      Up& r = up[(size_t)k];
      const Component& c = comps[(size_t)k];
      r.hs = hmax / c.h; r.vs = vmax / c.v; r.ystep = r.vs >> 1;
      r.w_lores = (width + r.hs - 1) / r.hs; r.ypos = 0;
      r.line0 = r.line1 = c.data.data();
      r.buf.assign((size_t)width + 8, 0);
    }
    std::vector<const uint8_t*> row((size_t)n);
    for (int j = 0; j < height; j++) {
      for (int k = 0; k < n; k++) {
        Up& r = up[(size_t)k];
        const Component& c = comps[(size_t)k];
        const bool y_bot = r.ystep >= (r.vs >> 1);
        const uint8_t* nr = y_bot ? r.line1 : r.line0;
        const uint8_t* fr = y_bot ? r.line0 : r.line1;
        uint8_t* o = r.buf.data();
        const int w = r.w_lores;
        if (r.hs == 1 && r.vs == 1) {
          row[(size_t)k] = nr;
        } else {
          if (r.hs == 1 && r.vs == 2) {
            for (int i = 0; i < w; i++) o[i] = (uint8_t)((3 * nr[i] + fr[i] + 2) >> 2);
          } else if (r.hs == 2 && r.vs == 1) {
            if (w == 1) o[0] = o[1] = nr[0];
            else {
              o[0] = nr[0];
              o[1] = (uint8_t)((nr[0] * 3 + nr[1] + 2) >> 2);
              int i;
              for (i = 1; i < w - 1; i++) {
                const int t = 3 * nr[i] + 2;
                o[i * 2] = (uint8_t)((t + nr[i - 1]) >> 2);
                o[i * 2 + 1] = (uint8_t)((t + nr[i + 1]) >> 2);
              }
              o[i * 2] = (uint8_t)((nr[w - 2] * 3 + nr[w - 1] + 2) >> 2);
              o[i * 2 + 1] = nr[w - 1];
            }
          } else if (r.hs == 2 && r.vs == 2) {
            if (w == 1) o[0] = o[1] = (uint8_t)((3 * nr[0] + fr[0] + 2) >> 2);
            else {
              int t1 = 3 * nr[0] + fr[0];
              o[0] = (uint8_t)((t1 + 2) >> 2);
              for (int i = 1; i < w; i++) {
                const int t0 = t1;
                t1 = 3 * nr[i] + fr[i];
                o[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                o[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
              }
              o[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
            }
          } else {
            for (int i = 0; i < w; i++)
              for (int q = 0; q < r.hs; q++)
                if (i * r.hs + q < width + 8) o[i * r.hs + q] = nr[i];
          }
          row[(size_t)k] = o;
        }
        if (++r.ystep >= r.vs) {
          r.ystep = 0;
          r.line0 = r.line1;
          if (++r.ypos < c.y) r.line1 += c.w2;
        }
      }
      uint8_t* px = &out[(size_t)j * width * 4];
      auto ycc = [&](uint8_t* o, int i) {
        const int yf = (row[0][i] << 20) + (1 << 19);
        const int cr = row[2][i] - 128, cb = row[1][i] - 128;
        auto fx = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
        int r = yf + cr * fx(1.40200f);
        int g = yf + (cr * -fx(0.71414f)) + ((cb * -fx(0.34414f)) & 0xffff0000);
        int b = yf + cb * fx(1.77200f);
        r >>= 20; g >>= 20; b >>= 20;
        o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b); o[3] = 255;
      };
      for (int i = 0; i < width; i++, px += 4) {
        if (n == 3) {
          if (is_rgb) { px[0] = row[0][i]; px[1] = row[1][i]; px[2] = row[2][i]; px[3] = 255; }
          else ycc(px, i);
        } else if (n == 4) {
          const uint8_t m = row[3][i];
          if (app14_transform == 0) {  // CMYK
            px[0] = blinn(row[0][i], m); px[1] = blinn(row[1][i], m); px[2] = blinn(row[2][i], m); px[3] = 255;
          } else if (app14_transform == 2) {  // YCCK
            ycc(px, i);
            px[0] = blinn((uint8_t)(255 - px[0]), m); px[1] = blinn((uint8_t)(255 - px[1]), m); px[2] = blinn((uint8_t)(255 - px[2]), m);
          } else ycc(px, i);  // YCbCr + a fourth channel that is ignored
        } else {
          px[0] = px[1] = px[2] = row[0][i]; px[3] = 255;
        }
      }
    }
    return out;
  }
};

}  // namespace

bool is_jpeg(const uint8_t* data, size_t len) { return len >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff; }

std::vector<uint8_t> decode_jpeg_rgba8(const uint8_t* data, size_t len, uint32_t* w, uint32_t* h) {
  Decoder d;
  d.p = data;
  d.end = data + len;
  d.decode_image();
  *w = (uint32_t)d.width;
  *h = (uint32_t)d.height;
  return d.to_rgba();
}

}  // namespace ptio
