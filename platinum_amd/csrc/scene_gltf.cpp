// scene_gltf.cpp — the import half of scene ingestion: a minimal glTF 2.0 reader that produces what
// loaders::gltf::GltfLoader (/root/reference/src/loaders/gltf.cpp) produces, PNG decoding for its textures, and
// MikkTSpace-compatible tangents (core/mesh.cpp:135-157 runs the reference's deps/mikkt on the indexed vertices).
// Host-only C++17.  Third-party pieces of the reference that are restated from their published behaviour:
//   fastgltf (deps/fastgltf, headers only in the tree): accessor iteration with component conversion, material defaults
//     (types.hpp:1873-2014), decomposeTransformMatrix (math.hpp:854-891)
//   stb_image (deps/stb_image): PNG -> 8-bit RGBA here, JPEG -> 8-bit RGBA in scene_jpeg.cpp.
//   mikktspace.c (deps/mikkt): the tangent-space algorithm, restated for triangle lists; pinned by tests against the
//     reference's own mikktspace.c compiled into oracle/_ref (tests/test_scene_ingestion.py).
#include "scene_io.h"

#include <zlib.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace ptio {

[[noreturn]] static void fail(const std::string& m) { throw std::runtime_error(m); }

// ---------------------------------------------------------------------------------------------------------------
// PNG -> RGBA8 (what stbi_load_from_memory(..., 4) returns for a PNG): all colour types, bit depths 1-16, tRNS, Adam7
// ---------------------------------------------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

std::vector<uint8_t> decode_png_rgba8(const uint8_t* data, size_t len, uint32_t* w_out, uint32_t* h_out) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (len < 8 || memcmp(data, sig, 8)) {
    fail("image: not a PNG");
  }
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t p = 8;
  bool end = false;
  while (!end && p + 12 <= len) {
    const uint32_t clen = be32(data + p);
    const uint8_t* type = data + p + 4;
    const uint8_t* body = data + p + 8;
    if (p + 12 + (size_t)clen > len) fail("png: truncated chunk");
    if (!memcmp(type, "IHDR", 4)) {
      if (clen < 13) fail("png: bad IHDR");
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + clen);
    else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + clen);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + clen);
    else if (!memcmp(type, "IEND", 4)) end = true;
    p += 12 + (size_t)clen;
  }
  if (!w || !h || w > 32768 || h > 32768) fail("png: bad size");
  if (interlace > 1) fail("png: unknown interlace method");
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels || !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) fail("png: unsupported colour type / bit depth");
  const size_t bpp_bits = (size_t)channels * depth;
  const size_t fbpp = std::max<size_t>(1, bpp_bits / 8);  // filter byte distance
  // The image as one pass, or Adam7's seven reduced images (each filtered on its own, stored one after the other)
  struct Pass { uint32_t x0, dx, y0, dy, pw, ph; };
  std::vector<Pass> passes;
  if (!interlace) passes.push_back({0, 1, 0, 1, w, h});
  else {
    static const uint32_t xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    for (int k = 0; k < 7; k++) {
      const uint32_t pw = (w - xo[k] + xs[k] - 1) / xs[k], ph = (h - yo[k] + ys[k] - 1) / ys[k];
      if (w > xo[k] && h > yo[k] && pw && ph) passes.push_back({xo[k], xs[k], yo[k], ys[k], pw, ph});
    }
  }
  size_t raw_size = 0;
  for (const Pass& ps : passes) raw_size += ((ps.pw * bpp_bits + 7) / 8 + 1) * ps.ph;
  std::vector<uint8_t> raw(raw_size);
  {
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) fail("png: zlib stream is corrupt");
  }
  std::vector<uint8_t> out((size_t)w * h * 4);
  size_t raw_at = 0;
  for (const Pass& ps : passes) {
  const size_t stride = (ps.pw * bpp_bits + 7) / 8;
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  for (uint32_t py = 0; py < ps.ph; py++) {
    const uint32_t y = ps.y0 + py * ps.dy;
    const uint8_t ft = raw[raw_at];
    const uint8_t* src = &raw[raw_at + 1];
    raw_at += stride + 1;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= fbpp ? cur[i - fbpp] : 0, b = prev[i], c = i >= fbpp ? prev[i - fbpp] : 0;
      int v = src[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: { const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: fail("png: bad filter type");
      }
      cur[i] = (uint8_t)v;
    }
    auto sample = [&](size_t idx) -> uint32_t {  // idx-th sample of the row, raw value
      if (depth == 8) return cur[idx];
      if (depth == 16) return (uint32_t)cur[2 * idx] << 8 | cur[2 * idx + 1];
      const size_t bit = idx * depth;
      return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    };
    auto to8 = [&](uint32_t v) -> uint8_t {  // stb: 16-bit -> high byte; 1/2/4-bit grey scaled to 0..255
      if (depth == 16) return (uint8_t)(v >> 8);
      if (depth == 8) return (uint8_t)v;
      return (uint8_t)(v * (depth == 1 ? 255 : depth == 2 ? 85 : 17));
    };
    for (uint32_t x = 0; x < ps.pw; x++) {
      uint8_t* o = &out[((size_t)y * w + ps.x0 + (size_t)x * ps.dx) * 4];
      switch (ctype) {
        case 0: {
          const uint32_t g = sample(x);
          o[0] = o[1] = o[2] = to8(g);
          o[3] = (trns.size() >= 2 && g == ((uint32_t)trns[0] << 8 | trns[1])) ? 0 : 255;
          break;
        }
        case 2: {
          const uint32_t r = sample(3 * x), g = sample(3 * x + 1), b = sample(3 * x + 2);
          o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
          o[3] = (trns.size() >= 6 && r == ((uint32_t)trns[0] << 8 | trns[1]) && g == ((uint32_t)trns[2] << 8 | trns[3]) &&
                  b == ((uint32_t)trns[4] << 8 | trns[5])) ? 0 : 255;
          break;
        }
        case 3: {
          const uint32_t i = sample(x);
          if (3 * (size_t)i + 2 >= plte.size()) fail("png: palette index out of range");
          o[0] = plte[3 * i]; o[1] = plte[3 * i + 1]; o[2] = plte[3 * i + 2];
          o[3] = i < trns.size() ? trns[i] : 255;
          break;
        }
        case 4: o[0] = o[1] = o[2] = to8(sample(2 * x)); o[3] = to8(sample(2 * x + 1)); break;
        default: o[0] = to8(sample(4 * x)); o[1] = to8(sample(4 * x + 1)); o[2] = to8(sample(4 * x + 2)); o[3] = to8(sample(4 * x + 3)); break;
      }
    }
    prev.swap(cur);
  }
  }
  *w_out = w; *h_out = h;
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Tangents: MikkTSpace (Mikkelsen 2008; deps/mikkt/mikktspace.c genTangSpaceDefault) for a triangle list.
// Same arithmetic in the same order as the reference's library (SVec3 helpers, angle weights through double acos), so
// that with contraction off the result is the same bits; the data structures are our own.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct V3 { float x, y, z; };
inline V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 vscale(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float vlen(V3 v) { return sqrtf(vdot(v, v)); }
inline V3 vnormalize(V3 v) { return vscale(1.0f / vlen(v), v); }
inline bool not_zero(float x) { return fabsf(x) > FLT_MIN; }
inline bool vnot_zero(V3 v) { return not_zero(v.x) || not_zero(v.y) || not_zero(v.z); }
inline bool veq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline V3 project_unit(V3 v, V3 n) { V3 r = vsub(v, vscale(vdot(n, v), n)); return vnot_zero(r) ? vnormalize(r) : r; }

struct TSpace { V3 os{1.0f, 0.0f, 0.0f}; bool orient = false; };
struct Tri {
  V3 os{0, 0, 0}, ot{0, 0, 0};
  float mag_s = 0, mag_t = 0;
  bool orient = false, any = true;  // ORIENT_PRESERVING, GROUP_WITH_ANY
  int neighbor[3] = {-1, -1, -1};
  int group[3] = {-1, -1, -1};
  uint32_t face = 0;  // original triangle number
};
struct Group { int vertex; bool orient; std::vector<int> faces; };

struct VertKey {
  float v[8];
  bool operator==(const VertKey& o) const { for (int i = 0; i < 8; i++) if (!(v[i] == o.v[i])) return false; return true; }
};
struct VertKeyHash {
  size_t operator()(const VertKey& k) const {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 8; i++) { float f = k.v[i] == 0.0f ? 0.0f : k.v[i]; uint32_t b; memcpy(&b, &f, 4); h = (h ^ b) * 1099511628211ull; }
    return (size_t)h;
  }
};
}  // namespace

void generate_tangents(const pt_float3* positions, pt_vertex_data* vdata, uint32_t vertex_count, const uint32_t* indices,
                       uint32_t triangle_count) {
  (void)vertex_count;
  if (triangle_count == 0) return;  // genTangSpace returns false before touching anything
  const size_t ncorner = 3 * (size_t)triangle_count;
  auto pos = [&](int c) { const pt_float3& p = positions[indices[c]]; return V3{p.x, p.y, p.z}; };
  auto nrm = [&](int c) { const pt_float3& p = vdata[indices[c]].normal; return V3{p.x, p.y, p.z}; };
  auto uv = [&](int c) { const float* t = vdata[indices[c]].texCoords; return V3{t[0], t[1], 1.0f}; };

  // welded index list: corners with identical (position, normal, texcoord) share the lowest such corner
  std::vector<int> weld(ncorner);
  {
    std::unordered_map<VertKey, int, VertKeyHash> seen;
    seen.reserve(ncorner);
    for (size_t c = 0; c < ncorner; c++) {
      const V3 p = pos((int)c), n = nrm((int)c), t = uv((int)c);
      const VertKey k{{p.x, p.y, p.z, n.x, n.y, n.z, t.x, t.y}};
      auto it = seen.find(k);
      if (it == seen.end()) { seen.emplace(k, (int)c); weld[c] = (int)c; } else weld[c] = it->second;
    }
  }
  // good triangles first (in order), degenerate ones (two equal positions) after
  std::vector<uint32_t> good, degenerate;
  for (uint32_t f = 0; f < triangle_count; f++) {
    const V3 p0 = pos(weld[3 * f]), p1 = pos(weld[3 * f + 1]), p2 = pos(weld[3 * f + 2]);
    (veq(p0, p1) || veq(p0, p2) || veq(p1, p2) ? degenerate : good).push_back(f);
  }
  const int ng = (int)good.size();
  std::vector<Tri> tris(ng);
  std::vector<int> tl(3 * (size_t)ng);  // welded corner indices of the good triangles
  for (int t = 0; t < ng; t++) {
    tris[t].face = good[t];
    for (int i = 0; i < 3; i++) tl[3 * t + i] = weld[3 * (size_t)good[t] + i];
  }
  // first-order derivatives (InitTriInfo)
  for (int f = 0; f < ng; f++) {
    const V3 v1 = pos(tl[3 * f]), v2 = pos(tl[3 * f + 1]), v3 = pos(tl[3 * f + 2]);
    const V3 t1 = uv(tl[3 * f]), t2 = uv(tl[3 * f + 1]), t3 = uv(tl[3 * f + 2]);
    const float t21x = t2.x - t1.x, t21y = t2.y - t1.y, t31x = t3.x - t1.x, t31y = t3.y - t1.y;
    const V3 d1 = vsub(v2, v1), d2 = vsub(v3, v1);
    const float area2 = t21x * t31y - t21y * t31x;
    const V3 os = vsub(vscale(t31y, d1), vscale(t21y, d2));
    const V3 ot = vadd(vscale(-t31x, d1), vscale(t21x, d2));
    Tri& T = tris[f];
    T.orient = area2 > 0;
    if (not_zero(area2)) {
      const float abs_area = fabsf(area2), len_os = vlen(os), len_ot = vlen(ot);
      const float s = T.orient ? 1.0f : -1.0f;
      if (not_zero(len_os)) T.os = vscale(s / len_os, os);
      if (not_zero(len_ot)) T.ot = vscale(s / len_ot, ot);
      T.mag_s = len_os / abs_area;
      T.mag_t = len_ot / abs_area;
      if (not_zero(T.mag_s) && not_zero(T.mag_t)) T.any = false;
    }
  }
  // neighbours: edges sorted by (min index, max index, triangle); an unassigned edge pairs with the first later entry of
  // the same undirected edge that runs the opposite way and is itself unassigned (BuildNeighborsFast)
  {
    struct Edge { int i0, i1, f; };
    std::vector<Edge> edges(3 * (size_t)ng);
    for (int f = 0; f < ng; f++)
      for (int i = 0; i < 3; i++) {
        const int a = tl[3 * f + i], b = tl[3 * f + (i < 2 ? i + 1 : 0)];
        edges[3 * (size_t)f + i] = {std::min(a, b), std::max(a, b), f};
      }
    std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) { return a.i0 != b.i0 ? a.i0 < b.i0 : a.i1 != b.i1 ? a.i1 < b.i1 : a.f < b.f; });
    auto get_edge = [&](int f, int i0, int i1, int* a, int* b) -> int {  // GetEdge: directed ends and edge number
      const int* ix = &tl[3 * f];
      if (ix[0] == i0 || ix[0] == i1) {
        if (ix[1] == i0 || ix[1] == i1) { *a = ix[0]; *b = ix[1]; return 0; }
        *a = ix[2]; *b = ix[0]; return 2;
      }
      *a = ix[1]; *b = ix[2]; return 1;
    };
    for (size_t i = 0; i < edges.size(); i++) {
      const Edge& e = edges[i];
      int a0, a1;
      const int ea = get_edge(e.f, e.i0, e.i1, &a0, &a1);
      if (tris[e.f].neighbor[ea] != -1) continue;
      for (size_t j = i + 1; j < edges.size() && edges[j].i0 == e.i0 && edges[j].i1 == e.i1; j++) {
        int b1, b0;
        const int eb = get_edge(edges[j].f, edges[j].i0, edges[j].i1, &b1, &b0);  // flipped on purpose
        if (a0 == b0 && a1 == b1 && tris[edges[j].f].neighbor[eb] == -1) {
          tris[e.f].neighbor[ea] = edges[j].f;
          tris[edges[j].f].neighbor[eb] = e.f;
          break;
        }
      }
    }
  }
  // groups (Build4RuleGroups / AssignRecur; the recursion is an explicit stack visiting L before R)
  std::vector<Group> groups;
  for (int f = 0; f < ng; f++)
    for (int i = 0; i < 3; i++) {
      if (tris[f].any || tris[f].group[i] != -1) continue;
      const int g = (int)groups.size();
      groups.push_back({tl[3 * f + i], tris[f].orient, {}});
      tris[f].group[i] = g;
      groups[g].faces.push_back(f);
      struct Frame { int tri; int stage; int corner; };
      std::vector<Frame> stack;
      auto enter = [&](int t) {  // AssignRecur prologue; pushes a frame when the triangle joins
        if (t < 0) return;
        Tri& T = tris[t];
        const int rep = groups[g].vertex;
        const int c = tl[3 * t] == rep ? 0 : tl[3 * t + 1] == rep ? 1 : 2;
        if (T.group[c] != -1) return;  // this group already, or another one
        if (T.any && T.group[0] == -1 && T.group[1] == -1 && T.group[2] == -1) T.orient = groups[g].orient;
        if (T.orient != groups[g].orient) return;
        groups[g].faces.push_back(t);
        T.group[c] = g;
        stack.push_back({t, 0, c});
      };
      // the seed triangle's two neighbours, L then R; depth-first like the recursion
      const int seedL = tris[f].neighbor[i], seedR = tris[f].neighbor[i > 0 ? i - 1 : 2];
      for (int seed : {seedL, seedR}) {
        enter(seed);
        while (!stack.empty()) {
          Frame& fr = stack.back();
          const int t = fr.tri, c = fr.corner;
          if (fr.stage == 0) { fr.stage = 1; enter(tris[t].neighbor[c]); }
          else if (fr.stage == 1) { fr.stage = 2; enter(tris[t].neighbor[c > 0 ? c - 1 : 2]); }
          else stack.pop_back();
        }
      }
    }
  // tangent spaces per group / sub-group (GenerateTSpaces, EvalTspace); threshold 180 degrees
  const float thres_cos = (float)cos((180.0f * (float)M_PI) / 180.0f);
  std::vector<TSpace> ts(ncorner);
  std::vector<int> members;
  for (size_t g = 0; g < groups.size(); g++) {
    const Group& G = groups[g];
    std::vector<std::vector<int>> sub_members;
    std::vector<V3> sub_os;
    for (int f : G.faces) {
      const int index = tris[f].group[0] == (int)g ? 0 : tris[f].group[1] == (int)g ? 1 : 2;
      const V3 n = nrm(tl[3 * f + index]);
      const V3 os = project_unit(tris[f].os, n), ot = project_unit(tris[f].ot, n);
      members.clear();
      for (int t : G.faces) {
        const V3 os2 = project_unit(tris[t].os, n), ot2 = project_unit(tris[t].ot, n);
        const bool any = tris[f].any || tris[t].any;
        const float cs = vdot(os, os2), ct = vdot(ot, ot2);
        if (any || f == t || (cs > thres_cos && ct > thres_cos)) members.push_back(t);
      }
      std::sort(members.begin(), members.end());
      size_t l = 0;
      while (l < sub_members.size() && sub_members[l] != members) l++;
      if (l == sub_members.size()) {
        V3 res{0, 0, 0};
        for (int t : members) {
          if (tris[t].any) continue;
          const int i = tl[3 * t] == G.vertex ? 0 : tl[3 * t + 1] == G.vertex ? 1 : 2;
          const V3 nn = nrm(tl[3 * t + i]);
          const V3 vos = project_unit(tris[t].os, nn);
          const V3 p0 = pos(tl[3 * t + (i > 0 ? i - 1 : 2)]), p1 = pos(tl[3 * t + i]), p2 = pos(tl[3 * t + (i < 2 ? i + 1 : 0)]);
          const V3 v1 = project_unit(vsub(p0, p1), nn), v2 = project_unit(vsub(p2, p1), nn);
          float c = vdot(v1, v2);
          c = c > 1 ? 1 : (c < -1 ? -1 : c);
          const float angle = (float)acos((double)c);
          res = vadd(res, vscale(angle, vos));
        }
        if (vnot_zero(res)) res = vnormalize(res);
        sub_members.push_back(members);
        sub_os.push_back(res);
      }
      TSpace& out = ts[3 * (size_t)tris[f].face + index];
      out.os = sub_os[l];
      out.orient = G.orient;
    }
  }
  // degenerate triangles copy from the first good corner with the same welded index (DegenEpilogue)
  if (!degenerate.empty()) {
    std::unordered_map<int, size_t> first_corner;  // welded index -> corner (3*face+i) of the first good triangle using it
    for (int t = 0; t < ng; t++)
      for (int i = 0; i < 3; i++) first_corner.emplace(tl[3 * t + i], 3 * (size_t)tris[t].face + i);
    for (uint32_t f : degenerate)
      for (int i = 0; i < 3; i++) {
        auto it = first_corner.find(weld[3 * (size_t)f + i]);
        if (it != first_corner.end()) ts[3 * (size_t)f + i] = ts[it->second];
      }
  }
  // setTSpaceBasic per face-vertex in face order onto the SHARED vertex: the last face to touch a vertex wins
  for (size_t c = 0; c < ncorner; c++) {
    float* t = vdata[indices[c]].tangent;
    t[0] = ts[c].os.x; t[1] = ts[c].os.y; t[2] = ts[c].os.z; t[3] = ts[c].orient ? 1.0f : -1.0f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// glTF 2.0
// ---------------------------------------------------------------------------------------------------------------
namespace {

std::string read_all(const std::string& path) {
  std::ifstream f(path, std::ios::in | std::ios::binary);
  if (!f) fail("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
std::string base64_decode(const std::string& s, size_t from) {
  std::string o;
  uint32_t acc = 0; int bits = 0;
  for (size_t i = from; i < s.size(); i++) {
    const char c = s[i];
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else if (c == '=') break; else continue;
    acc = acc << 6 | (uint32_t)v; bits += 6;
    if (bits >= 8) { bits -= 8; o += (char)((acc >> bits) & 0xFF); }
  }
  return o;
}
std::string uri_decode(const std::string& s) {
  std::string o;
  for (size_t i = 0; i < s.size(); i++) {
    if (s[i] == '%' && i + 2 < s.size()) { o += (char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; } else o += s[i];
  }
  return o;
}

struct Gltf {
  JV doc;
  std::string dir;
  std::vector<std::string> buffers;
  const JV* arr(const char* k) const { const JV* v = doc.find(k); return v && v->t == JV::ARR ? v : nullptr; }
  size_t count(const char* k) const { const JV* v = arr(k); return v ? v->a.size() : 0; }
  const JV& item(const char* k, size_t i) const {
    const JV* v = arr(k);
    if (!v || i >= v->a.size()) fail(std::string("gltf: index out of range in '") + k + "'");
    return v->a[i];
  }
};
float fnum(const JV& o, const char* k, float def) { const JV* v = o.find(k); return v ? (float)v->num() : def; }
std::string sname(const JV& o) { const JV* v = o.find("name"); return v && v->t == JV::STR ? v->s : std::string(); }

struct View { const uint8_t* p; size_t len; };
View buffer_view(const Gltf& g, size_t idx) {
  const JV& bv = g.item("bufferViews", idx);
  const size_t b = (size_t)bv.at("buffer").u64();
  if (b >= g.buffers.size()) fail("gltf: bufferView.buffer out of range");
  const size_t off = bv.find("byteOffset") ? (size_t)bv.at("byteOffset").u64() : 0, len = (size_t)bv.at("byteLength").u64();
  if (off + len > g.buffers[b].size()) fail("gltf: bufferView exceeds its buffer");
  return {(const uint8_t*)g.buffers[b].data() + off, len};
}

// fastgltf::iterateAccessor<T>: element i, component c of an accessor converted to float (normalized integers mapped
// to [0,1] / [-1,1] as the glTF spec prescribes) or to uint32 for indices.
struct Accessor {
  const uint8_t* base = nullptr;
  size_t count = 0, stride = 0;
  int comps = 0, ctype = 0;
  bool normalized = false;
  // accessor.sparse (fastgltf tools.hpp:529-548): element sp_index[k] is replaced by the k-th packed element of sp_values
  std::vector<uint32_t> sp_index;
  const uint8_t* sp_values = nullptr;
  bool zeros = false;  // no bufferView: the elements that sparse does not replace are zero (glTF 2.0 §3.6.2.3)
  size_t csize() const { return ctype == 5120 || ctype == 5121 ? 1 : ctype == 5122 || ctype == 5123 ? 2 : 4; }
  const uint8_t* element(size_t i) const {
    static const uint8_t zero[16] = {0};
    if (!sp_index.empty()) {
      const auto it = std::lower_bound(sp_index.begin(), sp_index.end(), (uint32_t)i);
      if (it != sp_index.end() && *it == i) return sp_values + (size_t)(it - sp_index.begin()) * csize() * (size_t)comps;
    }
    return zeros ? zero : base + i * stride;
  }
  float f(size_t i, int c) const {
    const uint8_t* p = element(i) + (size_t)c * csize();
    switch (ctype) {
      case 5126: { float v; memcpy(&v, p, 4); return v; }
      case 5121: return normalized ? (float)p[0] / 255.0f : (float)p[0];
      case 5120: { const int8_t v = (int8_t)p[0]; return normalized ? std::max((float)v / 127.0f, -1.0f) : (float)v; }
      case 5123: { uint16_t v; memcpy(&v, p, 2); return normalized ? (float)v / 65535.0f : (float)v; }
      case 5122: { int16_t v; memcpy(&v, p, 2); return normalized ? std::max((float)v / 32767.0f, -1.0f) : (float)v; }
      case 5125: { uint32_t v; memcpy(&v, p, 4); return (float)v; }
      default: fail("gltf: bad componentType");
    }
  }
  uint32_t u(size_t i) const {
    const uint8_t* p = element(i);
    switch (ctype) {
      case 5121: return p[0];
      case 5123: { uint16_t v; memcpy(&v, p, 2); return v; }
      case 5125: { uint32_t v; memcpy(&v, p, 4); return v; }
      default: fail("gltf: index accessor must be an unsigned integer type");
    }
  }
};
Accessor accessor(const Gltf& g, size_t idx, int want_comps) {
  const JV& a = g.item("accessors", idx);
  Accessor r;
  r.count = (size_t)a.at("count").u64();
  r.ctype = (int)a.at("componentType").u64();
  const std::string& type = a.at("type").str();
  r.comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
  if (r.comps != want_comps) fail("gltf: accessor has type " + type + ", expected " + std::to_string(want_comps) + " components");
  r.normalized = a.find("normalized") && a.at("normalized").boolean();
  const size_t elem = r.csize() * (size_t)r.comps;
  if (const JV* sp = a.find("sparse")) {
    const size_t n = (size_t)sp->at("count").u64();
    const JV& si = sp->at("indices");
    const JV& sv = sp->at("values");
    const View iv = buffer_view(g, (size_t)si.at("bufferView").u64()), vv = buffer_view(g, (size_t)sv.at("bufferView").u64());
    const size_t ioff = si.find("byteOffset") ? (size_t)si.at("byteOffset").u64() : 0, voff = sv.find("byteOffset") ? (size_t)sv.at("byteOffset").u64() : 0;
    const int ict = (int)si.at("componentType").u64();
    const size_t isz = ict == 5121 ? 1 : ict == 5123 ? 2 : ict == 5125 ? 4 : 0;
    if (!isz) fail("gltf: sparse indices must be an unsigned integer type");
    if (n > r.count || ioff + n * isz > iv.len || voff + n * elem > vv.len) fail("gltf: sparse accessor exceeds its bufferViews");
    r.sp_index.resize(n);
    for (size_t k = 0; k < n; k++) {
      const uint8_t* p = iv.p + ioff + k * isz;
      uint32_t x = 0;
      if (isz == 1) x = p[0]; else if (isz == 2) { uint16_t t; memcpy(&t, p, 2); x = t; } else memcpy(&x, p, 4);
      if (x >= r.count || (k && x <= r.sp_index[k - 1])) fail("gltf: sparse indices must be strictly increasing and below accessor.count");
      r.sp_index[k] = x;
    }
    r.sp_values = vv.p + voff;
  }
  if (!a.find("bufferView")) {
    if (r.sp_index.empty() && r.count) fail("gltf: accessor without a bufferView is not supported");
    r.zeros = true;
    return r;
  }
  const size_t bvi = (size_t)a.at("bufferView").u64();
  const View v = buffer_view(g, bvi);
  const JV& bv = g.item("bufferViews", bvi);
  const size_t off = a.find("byteOffset") ? (size_t)a.at("byteOffset").u64() : 0;
  r.stride = bv.find("byteStride") ? (size_t)bv.at("byteStride").u64() : elem;
  if (r.count && off + (r.count - 1) * r.stride + elem > v.len) fail("gltf: accessor exceeds its bufferView");
  r.base = v.p + off;
  return r;
}

// loaders/gltf.cpp:9-17
void euler_from_quat(const float q[4] /*x,y,z,w*/, float out[3]) {
  const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  out[0] = atan2f(2.0f * (qw * qx - qy * qz), 1.0f - 2.0f * (qx * qx + qz * qz));
  out[1] = atan2f(2.0f * (qw * qy - qx * qz), 1.0f - 2.0f * (qy * qy + qz * qz));
  out[2] = asinf(2.0f * std::min(std::max(qx * qy + qw * qz, -0.5f), 0.5f));
}
// fastgltf math.hpp:854-891 (Options::DecomposeNodeMatrices)
void decompose(const float m[16] /*column-major*/, float t[3], float q[4], float s[3]) {
  float c[3][3];
  for (int k = 0; k < 3; k++) {
    t[k] = m[12 + k];
    s[k] = sqrtf(m[4 * k] * m[4 * k] + m[4 * k + 1] * m[4 * k + 1] + m[4 * k + 2] * m[4 * k + 2] + m[4 * k + 3] * m[4 * k + 3]);
    for (int r = 0; r < 3; r++) c[k][r] = m[4 * k + r] / s[k];
  }
  q[0] = std::max(0.0f, 1.0f + c[0][0] - c[1][1] - c[2][2]);
  q[1] = std::max(0.0f, 1.0f - c[0][0] + c[1][1] - c[2][2]);
  q[2] = std::max(0.0f, 1.0f - c[0][0] - c[1][1] + c[2][2]);
  q[3] = std::max(0.0f, 1.0f + c[0][0] + c[1][1] + c[2][2]);
  for (int k = 0; k < 4; k++) q[k] = (float)sqrt((double)q[k]) / 2;
  q[0] = copysignf(q[0], c[1][2] - c[2][1]);
  q[1] = copysignf(q[1], c[2][0] - c[0][2]);
  q[2] = copysignf(q[2], c[0][1] - c[1][0]);
}

enum TexType { TT_SRGB, TT_LINEAR, TT_MONO, TT_RM };  // loaders/texture.hpp TextureType (HDR comes through set_environment)

struct Importer {
  Scene& scene;
  Gltf g;
  int options = 0;
  std::vector<uint64_t> material_ids, mesh_ids;
  std::map<uint64_t, std::vector<uint64_t>> mesh_materials;
  std::vector<Camera> cameras;
  struct TexUse { int type = TT_SRGB; std::vector<std::pair<uint64_t, int>> users; };
  std::vector<std::pair<size_t, TexUse>> textures_to_load;  // insertion-ordered map keyed by glTF texture index
  TexUse& tex_use(size_t idx) {
    for (auto& t : textures_to_load) if (t.first == idx) return t.second;
    textures_to_load.emplace_back(idx, TexUse{});
    return textures_to_load.back().second;
  }

  void load_material(const JV& m) {  // loaders/gltf.cpp:304-394
    Asset a;
    a.type = Asset::MATERIAL;
    Material& mat = a.mat;
    mat.name = sname(m);
    const JV* pbr = m.find("pbrMetallicRoughness");
    const JV empty;
    const JV& P = pbr ? *pbr : empty;
    for (int i = 0; i < 4; i++) mat.base_color[i] = 1.0f;
    if (const JV* f = P.find("baseColorFactor")) for (int i = 0; i < 4; i++) mat.base_color[i] = (float)f->at((size_t)i).num();
    mat.roughness = fnum(P, "roughnessFactor", 1.0f);
    mat.metallic = fnum(P, "metallicFactor", 1.0f);
    const JV* ext = m.find("extensions");
    auto extension = [&](const char* name) -> const JV* { return ext ? ext->find(name) : nullptr; };
    const JV* tr = extension("KHR_materials_transmission");
    if (tr) mat.transmission = fnum(*tr, "transmissionFactor", 0.0f);
    const JV* es = extension("KHR_materials_emissive_strength");
    mat.emission_strength = es ? fnum(*es, "emissiveStrength", 1.0f) : 1.0f;  // fastgltf types.hpp:2009
    if (const JV* e = m.find("emissiveFactor")) for (int i = 0; i < 3; i++) mat.emission[i] = (float)e->at((size_t)i).num();
    const JV* ior = extension("KHR_materials_ior");
    mat.ior = ior ? fnum(*ior, "ior", 1.5f) : 1.5f;
    const JV* an = extension("KHR_materials_anisotropy");
    if (an) { mat.anisotropy = fnum(*an, "anisotropyStrength", 0.0f); mat.anisotropy_rotation = fnum(*an, "anisotropyRotation", 0.0f); }
    const JV* cc = extension("KHR_materials_clearcoat");
    if (cc) { mat.clearcoat = fnum(*cc, "clearcoatFactor", 0.0f); mat.clearcoat_roughness = fnum(*cc, "clearcoatRoughnessFactor", 0.0f); }
    const uint64_t id = scene.create_asset(std::move(a), false);
    material_ids.push_back(id);
    auto use = [&](const JV* info, int type, int slot) {
      if (!info) return;
      const size_t ti = (size_t)info->at("index").u64();
      TexUse& u = tex_use(ti);
      u.type = type;
      u.users.emplace_back(id, slot);
    };
    use(P.find("baseColorTexture"), TT_SRGB, 0);
    use(P.find("metallicRoughnessTexture"), TT_RM, 1);
    use(m.find("normalTexture"), TT_LINEAR, 5);
    use(m.find("emissiveTexture"), TT_SRGB, 4);
    if (tr) use(tr->find("transmissionTexture"), TT_MONO, 2);
    if (cc) use(cc->find("clearcoatTexture"), TT_MONO, 3);
  }

  uint64_t load_texture(size_t tex_index, int type) {  // loaders/gltf.cpp:399-420 + loaders/texture.cpp:113-218
    const JV& tex = g.item("textures", tex_index);
    const JV& img = g.item("images", (size_t)tex.at("source").u64());
    std::string storage;
    View v{nullptr, 0};
    if (const JV* bv = img.find("bufferView")) v = buffer_view(g, (size_t)bv->u64());
    else if (const JV* uri = img.find("uri")) {  // (the reference only handles bufferView images; URIs are a superset)
      const std::string& u = uri->str();
      if (u.rfind("data:", 0) == 0) { const size_t c = u.find(','); if (c == std::string::npos) fail("gltf: bad data URI"); storage = base64_decode(u, c + 1); }
      else storage = read_all(g.dir + uri_decode(u));
      v = {(const uint8_t*)storage.data(), storage.size()};
    } else fail("gltf: image has neither bufferView nor uri");
    uint32_t w = 0, h = 0;
    // stbi_load_from_memory(data, len, &w, &h, nullptr, 4) (texture.cpp:111-119): the format is sniffed from the bytes
    const std::vector<uint8_t> rgba = is_jpeg(v.p, v.len) ? decode_jpeg_rgba8(v.p, v.len, &w, &h) : decode_png_rgba8(v.p, v.len, &w, &h);
    Asset a;
    a.type = Asset::TEXTURE;
    a.tex.name = sname(tex);
    a.tex.width = w; a.tex.height = h;
    a.tex.alpha = false;
    for (size_t i = 0; i < (size_t)w * h; i++) if (rgba[4 * i + 3] < 255) { a.tex.alpha = true; break; }  // texture.cpp:135-143
    const size_t n = (size_t)w * h;
    switch (type) {  // getAttributesForTexture (texture.cpp:30-48) + convertTexture channel map (texture_converter.metal)
      case TT_SRGB: a.tex.mtl_format = PT_MTL_RGBA8UNORM_SRGB; a.tex.bytes = rgba; break;
      case TT_LINEAR: a.tex.mtl_format = PT_MTL_RGBA8UNORM; a.tex.bytes = rgba; break;
      case TT_MONO: a.tex.mtl_format = PT_MTL_R8UNORM; a.tex.bytes.resize(n); for (size_t i = 0; i < n; i++) a.tex.bytes[i] = rgba[4 * i]; break;
      default: a.tex.mtl_format = PT_MTL_RG8UNORM; a.tex.bytes.resize(2 * n);
        for (size_t i = 0; i < n; i++) { a.tex.bytes[2 * i] = rgba[4 * i + 1]; a.tex.bytes[2 * i + 1] = rgba[4 * i + 2]; }
        break;
    }
    return scene.create_asset(std::move(a), false);
  }

  void load_mesh(const JV& mesh) {  // loaders/gltf.cpp:115-248
    Asset a;
    a.type = Asset::MESH;
    Mesh& M = a.mesh;
    std::vector<uint64_t> slots;
    bool loaded_tangents = false;
    uint32_t slot_idx = 0;
    for (const JV& prim : mesh.at("primitives").a) {
      const int mode = prim.find("mode") ? (int)prim.at("mode").u64() : 4;
      if (mode != 4) { fprintf(stderr, "[Warn] gltf: Unsupported primitive type\n"); continue; }
      const JV& attrs = prim.at("attributes");
      const size_t offset = M.positions.size();
      const Accessor pos = accessor(g, (size_t)attrs.at("POSITION").u64(), 3);
      M.positions.resize(offset + pos.count);
      M.vdata.resize(offset + pos.count);  // VertexData{}: zero normal / tangent / texcoords
      for (size_t i = 0; i < pos.count; i++) M.positions[offset + i] = {pos.f(i, 0), pos.f(i, 1), pos.f(i, 2), 0.0f};
      if (const JV* n = attrs.find("NORMAL")) {
        const Accessor acc = accessor(g, (size_t)n->u64(), 3);
        for (size_t i = 0; i < std::min(acc.count, pos.count); i++) M.vdata[offset + i].normal = {acc.f(i, 0), acc.f(i, 1), acc.f(i, 2), 0.0f};
      }
      if (const JV* t = attrs.find("TEXCOORD_0")) {
        const Accessor acc = accessor(g, (size_t)t->u64(), 2);
        for (size_t i = 0; i < std::min(acc.count, pos.count); i++) { M.vdata[offset + i].texCoords[0] = acc.f(i, 0); M.vdata[offset + i].texCoords[1] = acc.f(i, 1); }
      }
      if (const JV* t = attrs.find("TANGENT")) {
        const Accessor acc = accessor(g, (size_t)t->u64(), 4);
        for (size_t i = 0; i < std::min(acc.count, pos.count); i++) for (int c = 0; c < 4; c++) M.vdata[offset + i].tangent[c] = acc.f(i, c);
        loaded_tangents = true;
      }
      size_t nidx;
      if (const JV* ind = prim.find("indices")) {
        const Accessor acc = accessor(g, (size_t)ind->u64(), 1);
        nidx = acc.count;
        for (size_t i = 0; i < nidx; i++) {
          const uint32_t v = acc.u(i);
          if (v >= pos.count) fail("gltf: vertex index out of range");
          M.indices.push_back(v + (uint32_t)offset);
        }
      } else {  // Options::GenerateMeshIndices
        nidx = pos.count;
        for (size_t i = 0; i < nidx; i++) M.indices.push_back((uint32_t)(offset + i));
      }
      if (nidx % 3) fail("gltf: triangle primitive whose index count is not a multiple of 3");
      M.slots.insert(M.slots.end(), nidx / 3, slot_idx++);
      const JV* mi = prim.find("material");
      if (mi && (size_t)mi->u64() >= material_ids.size()) fail("gltf: primitive.material out of range");
      slots.push_back(mi ? material_ids[(size_t)mi->u64()] : 0);  // (sic) asset id 0 when the primitive has no material
    }
    if (!loaded_tangents && !M.indices.empty())
      generate_tangents(M.positions.data(), M.vdata.data(), (uint32_t)M.positions.size(), M.indices.data(), (uint32_t)(M.indices.size() / 3));
    const uint64_t id = scene.create_asset(std::move(a), false);
    mesh_ids.push_back(id);
    mesh_materials[id] = slots;
  }

  void load_node(size_t node_index, size_t parent, int depth) {  // loaders/gltf.cpp:253-293
    if (depth > 512) fail("gltf: node hierarchy too deep (cycle?)");
    const JV& n = g.item("nodes", node_index);
    const JV* mesh = n.find("mesh");
    const JV* cam = n.find("camera");
    const JV* children = n.find("children");
    const bool no_children = !children || children->a.empty();
    if ((options & PT_GLTF_SKIP_EMPTY_NODES) && !mesh && !cam && no_children) return;
    const size_t idx = scene.create_node(sname(n), parent, kNoId);
    if (cam) {
      const size_t ci = (size_t)cam->u64();
      if (ci >= cameras.size()) fail("gltf: node.camera does not name a perspective camera");
      scene.nodes[idx].camera = cameras[ci]; scene.nodes[idx].has_camera = true;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    if (const JV* m = n.find("matrix")) {
      float mm[16];
      for (int i = 0; i < 16; i++) mm[i] = (float)m->at((size_t)i).num();
      decompose(mm, t, q, s);
    } else {
      if (const JV* v = n.find("translation")) for (int i = 0; i < 3; i++) t[i] = (float)v->at((size_t)i).num();
      if (const JV* v = n.find("rotation")) for (int i = 0; i < 4; i++) q[i] = (float)v->at((size_t)i).num();
      if (const JV* v = n.find("scale")) for (int i = 0; i < 3; i++) s[i] = (float)v->at((size_t)i).num();
    }
    Transform& tr = scene.nodes[idx].transform;
    memcpy(tr.translation, t, 12); memcpy(tr.scale, s, 12);
    euler_from_quat(q, tr.rotation);
    if (mesh) {
      const size_t mi = (size_t)mesh->u64();
      if (mi >= mesh_ids.size()) fail("gltf: node.mesh out of range");
      scene.set_mesh(idx, mesh_ids[mi]);
      const auto& mats = mesh_materials[mesh_ids[mi]];
      for (size_t i = 0; i < mats.size(); i++) scene.set_material(idx, i, mats[i]);
    }
    if (children) for (const JV& c : children->a) load_node((size_t)c.u64(), idx, depth + 1);
  }
};

}  // namespace

void import_gltf(Scene& scene, const std::string& path, int options) {  // loaders/gltf.cpp:28-113
  Importer im{scene};
  im.options = options;
  Gltf& g = im.g;
  const size_t slash = path.find_last_of('/');
  g.dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
  std::string stem = slash == std::string::npos ? path : path.substr(slash + 1);
  if (stem.find_last_of('.') != std::string::npos && stem.find_last_of('.') > 0) stem = stem.substr(0, stem.find_last_of('.'));
  const std::string file = read_all(path);
  std::string glb_bin;
  bool has_glb_bin = false;
  if (file.size() >= 12 && !memcmp(file.data(), "glTF", 4)) {  // GLB container: 12-byte header, JSON chunk, optional BIN chunk
    size_t p = 12;
    std::string json;
    while (p + 8 <= file.size()) {
      uint32_t clen, ctype;
      memcpy(&clen, file.data() + p, 4); memcpy(&ctype, file.data() + p + 4, 4);
      if (p + 8 + (size_t)clen > file.size()) fail("glb: truncated chunk");
      if (ctype == 0x4E4F534A) json.assign(file.data() + p + 8, clen);
      else if (ctype == 0x004E4942 && !has_glb_bin) { glb_bin.assign(file.data() + p + 8, clen); has_glb_bin = true; }
      p += 8 + (size_t)clen;
    }
    if (json.empty()) fail("glb: no JSON chunk");
    g.doc = json_parse(json);
  } else g.doc = json_parse(file);

  for (size_t b = 0; b < g.count("buffers"); b++) {  // Options::LoadExternalBuffers
    const JV& buf = g.item("buffers", b);
    if (const JV* uri = buf.find("uri")) {
      const std::string& u = uri->str();
      if (u.rfind("data:", 0) == 0) { const size_t c = u.find(','); if (c == std::string::npos) fail("gltf: bad data URI"); g.buffers.push_back(base64_decode(u, c + 1)); }
      else g.buffers.push_back(read_all(g.dir + uri_decode(u)));
    } else if (b == 0 && has_glb_bin) g.buffers.push_back(glb_bin);
    else fail("gltf: buffer without uri");
    if (g.buffers.back().size() < (size_t)buf.at("byteLength").u64()) fail("gltf: buffer shorter than byteLength");
  }
  for (size_t i = 0; i < g.count("materials"); i++) im.load_material(g.item("materials", i));
  for (auto& tu : im.textures_to_load) {
    const uint64_t tid = im.load_texture(tu.first, tu.second.type);
    for (auto& user : tu.second.users) {  // Scene::updateMaterialTexture (core/scene.cpp:145-159)
      Asset* m = scene.find_asset(user.first);
      if (m->mat.get_texture(user.second) == tid) continue;
      scene.retain(tid);
      m->mat.set_texture(user.second, tid);
    }
  }
  for (size_t i = 0; i < g.count("meshes"); i++) im.load_mesh(g.item("meshes", i));
  for (size_t i = 0; i < g.count("cameras"); i++) {  // :83-91: perspective cameras only, Camera::withFov(yfov, {24*aspect, 24})
    const JV& c = g.item("cameras", i);
    const JV* p = c.find("perspective");
    if (!p) continue;
    Camera cam;
    const float aspect = p->find("aspectRatio") ? (float)p->at("aspectRatio").num() : 1.5f;
    cam.sensor_size[0] = 24.0f * aspect; cam.sensor_size[1] = 24.0f;
    cam.focal_length = cam.sensor_size[1] / (2.0f * tanf((float)p->at("yfov").num() * 0.5f));  // core/camera.hpp:32-42
    im.cameras.push_back(cam);
  }
  size_t local_root = scene.root_index;
  const size_t nscenes = g.count("scenes");
  uint32_t scene_idx = 0;
  for (size_t s = 0; s < nscenes; s++) {
    const JV& sc = g.item("scenes", s);
    if (options & PT_GLTF_CREATE_SCENE_NODES) {
      std::string name = stem;
      if (nscenes > 1) { char b[32]; snprintf(b, sizeof b, ".%3u", scene_idx++); name += b; }  // std::format("{}.{:3}", ...)
      local_root = scene.create_node(name, scene.root_index, kNoId);
    }
    if (const JV* nodes = sc.find("nodes")) for (const JV& n : nodes->a) im.load_node((size_t)n.u64(), local_root, 0);
  }
}

}  // namespace ptio
