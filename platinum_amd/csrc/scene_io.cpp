// scene_io.cpp — scene ingestion: the reference's scene.json + _data.bin format, a minimal glTF 2.0
// importer, MikkTSpace-compatible tangents, and the flattening of a node hierarchy into `pt_scene_snapshot`.
// Host-only C++17 (compiled by g++ with -ffp-contract=off; no HIP).  Interface and citations: include/ptamd_scene.h.
//
// What is a definition here rather than a reproduction ("parity unpinned"): simd float4x4 products and
// simd::inverse are Apple-closed — products are evaluated in fp32 as ((a0*b0 + a1*b1) + a2*b2) + a3*b3, the lookAt
// inverse in double and rounded once; sin/cos are the C library's.
#include "scene_io.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace ptio {

thread_local std::string g_error;

[[noreturn]] static void fail(const std::string& m) { throw std::runtime_error(m); }

// ---------------------------------------------------------------------------------------------------------------
// JSON (the subset nlohmann::json accepts for these files: no comments, UTF-8 passed through, \uXXXX decoded)
// ---------------------------------------------------------------------------------------------------------------
const JV* JV::find(const std::string& k) const {
  if (t != OBJ) return nullptr;
  for (const auto& kv : o) if (kv.first == k) return &kv.second;
  return nullptr;
}
const JV& JV::at(const std::string& k) const {
  const JV* v = find(k);
  if (!v) fail("json: key '" + k + "' not found");
  return *v;
}
const JV& JV::at(size_t i) const {
  if (t != ARR || i >= a.size()) fail("json: array index out of range");
  return a[i];
}
double JV::num() const { if (t != NUM) fail("json: number expected"); return d; }
uint64_t JV::u64() const {
  if (t != NUM) fail("json: number expected");
  if (is_uint) return u;
  if (d < 0) fail("json: unsigned integer expected");
  return (uint64_t)d;
}
bool JV::boolean() const { if (t != BOOL) fail("json: boolean expected"); return b; }
const std::string& JV::str() const { if (t != STR) fail("json: string expected"); return s; }

namespace {
struct Parser {
  const char* p; const char* e;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  [[noreturn]] void err(const char* m) { fail(std::string("json: ") + m + " at byte " + std::to_string((long)(p - (e - len)))); }
  size_t len = 0;
  JV value(int depth) {
    if (depth > 256) err("nesting too deep");
    ws();
    if (p >= e) err("unexpected end");
    JV v;
    switch (*p) {
      case '{': {
        v.t = JV::OBJ; p++; ws();
        if (p < e && *p == '}') { p++; return v; }
        for (;;) {
          ws();
          if (p >= e || *p != '"') err("string key expected");
          std::string k = string();
          ws();
          if (p >= e || *p != ':') err("':' expected");
          p++;
          v.o.emplace_back(std::move(k), value(depth + 1));
          ws();
          if (p < e && *p == ',') { p++; continue; }
          if (p < e && *p == '}') { p++; return v; }
          err("',' or '}' expected");
        }
      }
      case '[': {
        v.t = JV::ARR; p++; ws();
        if (p < e && *p == ']') { p++; return v; }
        for (;;) {
          v.a.push_back(value(depth + 1));
          ws();
          if (p < e && *p == ',') { p++; continue; }
          if (p < e && *p == ']') { p++; return v; }
          err("',' or ']' expected");
        }
      }
      case '"': v.t = JV::STR; v.s = string(); return v;
      case 't': if (e - p >= 4 && !memcmp(p, "true", 4)) { p += 4; v.t = JV::BOOL; v.b = true; return v; } err("bad literal");
      case 'f': if (e - p >= 5 && !memcmp(p, "false", 5)) { p += 5; v.t = JV::BOOL; v.b = false; return v; } err("bad literal");
      case 'n': if (e - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return v; } err("bad literal");
      default: return number();
    }
  }
  JV number() {
    const char* s = p;
    bool integral = true;
    if (p < e && *p == '-') p++;
    if (p >= e || !(*p >= '0' && *p <= '9')) err("value expected");
    while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
      if (*p == '.' || *p == 'e' || *p == 'E') integral = false;
      p++;
    }
    std::string tok(s, p);
    JV v; v.t = JV::NUM;
    v.d = strtod(tok.c_str(), nullptr);
    if (integral && tok[0] != '-' && tok.size() <= 20) { v.is_uint = true; v.u = strtoull(tok.c_str(), nullptr, 10); }
    return v;
  }
  static void utf8(std::string& o, uint32_t c) {
    if (c < 0x80) o += (char)c;
    else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
  }
  uint32_t hex4() {
    if (e - p < 4) err("bad \\u escape");
    uint32_t c = 0;
    for (int i = 0; i < 4; i++) {
      const char h = *p++;
      c = c * 16 + (h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : (err("bad hex"), 0));
    }
    return c;
  }
  std::string string() {
    std::string o;
    p++;  // opening quote
    while (p < e && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= e) err("bad escape");
        const char c = *p++;
        switch (c) {
          case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
          case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break;
          case 'r': o += '\r'; break; case 't': o += '\t'; break;
          case 'u': {
            uint32_t c1 = hex4();
            if (c1 >= 0xD800 && c1 < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') { p += 2; const uint32_t c2 = hex4(); c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00); }
            utf8(o, c1);
            break;
          }
          default: err("bad escape");
        }
      } else o += *p++;
    }
    if (p >= e) err("unterminated string");
    p++;
    return o;
  }
};

void dump_string(std::string& o, const std::string& s) {
  o += '"';
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break; case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
      default: if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o += (char)c;
    }
  }
  o += '"';
}
}  // namespace

JV json_parse(const std::string& text) {
  Parser ps{text.data(), text.data() + text.size()};
  ps.len = text.size();
  JV v = ps.value(0);
  ps.ws();
  if (ps.p != ps.e) ps.err("trailing characters");
  return v;
}

void json_dump(const JV& v, std::string& o) {
  switch (v.t) {
    case JV::NUL: o += "null"; break;
    case JV::BOOL: o += v.b ? "true" : "false"; break;
    case JV::NUM: {
      char b[40];
      if (v.is_uint) snprintf(b, sizeof b, "%llu", (unsigned long long)v.u);
      else if (std::isfinite(v.d)) {
        snprintf(b, sizeof b, "%.17g", v.d);
        if (!strpbrk(b, ".eE")) strcat(b, ".0");  // keep floats floats, as nlohmann does
      } else snprintf(b, sizeof b, "null");
      o += b;
      break;
    }
    case JV::STR: dump_string(o, v.s); break;
    case JV::ARR: o += '['; for (size_t i = 0; i < v.a.size(); i++) { if (i) o += ','; json_dump(v.a[i], o); } o += ']'; break;
    case JV::OBJ: {
      // nlohmann::json's default object is a sorted map: keys come out in lexicographic order
      std::vector<const std::pair<std::string, JV>*> kv;
      for (const auto& e : v.o) kv.push_back(&e);
      std::stable_sort(kv.begin(), kv.end(), [](auto* a, auto* b) { return a->first < b->first; });
      o += '{';
      for (size_t i = 0; i < kv.size(); i++) { if (i) o += ','; dump_string(o, kv[i]->first); o += ':'; json_dump(kv[i]->second, o); }
      o += '}';
      break;
    }
  }
}

static JV jnum(double d) { JV v; v.t = JV::NUM; v.d = d; return v; }
static JV juint(uint64_t u) { JV v; v.t = JV::NUM; v.is_uint = true; v.u = u; v.d = (double)u; return v; }
static JV jbool(bool b) { JV v; v.t = JV::BOOL; v.b = b; return v; }
static JV jstr(const std::string& s) { JV v; v.t = JV::STR; v.s = s; return v; }
static JV jarr(std::initializer_list<JV> l) { JV v; v.t = JV::ARR; v.a = l; return v; }
static JV jobj() { JV v; v.t = JV::OBJ; return v; }
static JV jvec(const float* f, int n) { JV v; v.t = JV::ARR; for (int i = 0; i < n; i++) v.a.push_back(jnum((double)f[i])); return v; }  // utils/json.hpp:27-29
static void parse_floats(const JV& j, float* out, int n) { for (int i = 0; i < n; i++) out[i] = (float)j.at((size_t)i).num(); }  // :41-43

// ---------------------------------------------------------------------------------------------------------------
// matrices (utils/matrices.cpp) and Transform::matrix (core/transform.hpp:36-51); [col][row] storage like simd
// ---------------------------------------------------------------------------------------------------------------
Mat4 mat_identity() { Mat4 m{}; for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f; return m; }
Mat4 mat_mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int c = 0; c < 4; c++)
    for (int row = 0; row < 4; row++)
      r.c[c][row] = ((a.c[0][row] * b.c[c][0] + a.c[1][row] * b.c[c][1]) + a.c[2][row] * b.c[c][2]) + a.c[3][row] * b.c[c][3];
  return r;
}
static Mat4 mat_translation(const float t[3]) { Mat4 m = mat_identity(); m.c[3][0] = t[0]; m.c[3][1] = t[1]; m.c[3][2] = t[2]; return m; }
static Mat4 mat_scaling(const float s[3]) { Mat4 m = mat_identity(); m.c[0][0] = s[0]; m.c[1][1] = s[1]; m.c[2][2] = s[2]; return m; }
static Mat4 mat_rotation_x(float a) { const float c = cosf(a), s = sinf(a); Mat4 m = mat_identity(); m.c[1][1] = c; m.c[1][2] = s; m.c[2][1] = -s; m.c[2][2] = c; return m; }   // :50-61
static Mat4 mat_rotation_y(float a) { const float c = cosf(a), s = sinf(a); Mat4 m = mat_identity(); m.c[0][0] = c; m.c[0][2] = -s; m.c[2][0] = s; m.c[2][2] = c; return m; }   // :63-74
static Mat4 mat_rotation_z(float a) { const float c = cosf(a), s = sinf(a); Mat4 m = mat_identity(); m.c[0][0] = c; m.c[0][1] = s; m.c[1][0] = -s; m.c[1][1] = c; return m; }   // :76-87
static void normalize3(float v[3]) { const float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }
static void cross3(const float a[3], const float b[3], float o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static Mat4 mat_look_at(const float pos[3], const float tgt[3], const float up[3]) {  // :132-145
  if (pos[0] == tgt[0] && pos[1] == tgt[1] && pos[2] == tgt[2]) return mat_identity();
  float f[3] = {pos[0] - tgt[0], pos[1] - tgt[1], pos[2] - tgt[2]}, s[3], u[3];
  normalize3(f);
  cross3(up, f, s); normalize3(s);
  cross3(f, s, u);
  Mat4 m{};
  m.c[0][0] = s[0]; m.c[0][1] = u[0]; m.c[0][2] = f[0];
  m.c[1][0] = s[1]; m.c[1][1] = u[1]; m.c[1][2] = f[1];
  m.c[2][0] = s[2]; m.c[2][1] = u[2]; m.c[2][2] = f[2];
  m.c[3][0] = -dot3(s, pos); m.c[3][1] = -dot3(u, pos); m.c[3][2] = -dot3(f, pos); m.c[3][3] = 1.0f;
  return m;
}
static Mat4 mat_inverse(const Mat4& m) {  // Gauss-Jordan with partial pivoting in double, rounded once
  double a[4][8];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = m.c[c][r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
  for (int i = 0; i < 4; i++) {
    int piv = i;
    for (int r = i + 1; r < 4; r++) if (fabs(a[r][i]) > fabs(a[piv][i])) piv = r;
    if (piv != i) for (int c = 0; c < 8; c++) std::swap(a[i][c], a[piv][c]);
    const double d = a[i][i];
    for (int c = 0; c < 8; c++) a[i][c] /= d;
    for (int r = 0; r < 4; r++) if (r != i) { const double f = a[r][i]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
  }
  Mat4 o;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o.c[c][r] = (float)a[r][4 + c];
  return o;
}
Mat4 Transform::matrix() const {  // core/transform.hpp:36-51
  const Mat4 S = mat_scaling(scale);
  if (track) {
    const float upz[3] = {0, 0, 1}, upy[3] = {0, 1, 0};
    const bool same_xz = translation[0] == target[0] && translation[2] == target[2];
    return mat_mul(mat_inverse(mat_look_at(translation, target, same_xz ? upz : upy)), S);
  }
  return mat_mul(mat_mul(mat_mul(mat_mul(mat_translation(translation), mat_rotation_y(rotation[1])), mat_rotation_x(rotation[0])),
                         mat_rotation_z(rotation[2])), S);
}

// ---------------------------------------------------------------------------------------------------------------
// Scene container
// ---------------------------------------------------------------------------------------------------------------
size_t mtl_bytes_per_pixel(uint32_t f) {
  switch (f) {
    case PT_MTL_R8UNORM: return 1; case PT_MTL_RG8UNORM: return 2; case PT_MTL_RGBA8UNORM: case PT_MTL_RGBA8UNORM_SRGB: return 4;
    case PT_MTL_RGBA32FLOAT: return 16;
    default: fail("texture: unsupported MTLPixelFormat " + std::to_string(f));
  }
}
static uint32_t mtl_to_pt_format(uint32_t f) {
  switch (f) {
    case PT_MTL_R8UNORM: return PT_TEX_R8; case PT_MTL_RG8UNORM: return PT_TEX_RG8; case PT_MTL_RGBA8UNORM: return PT_TEX_RGBA8;
    case PT_MTL_RGBA8UNORM_SRGB: return PT_TEX_RGBA8_SRGB; default: return PT_TEX_RGBA32F;
  }
}

Scene::Scene() {  // core/scene.cpp:22-28
  Node root;
  root.id = 0; root.name = "Scene"; root.parent = kNoNode;
  nodes.push_back(root);
  root_index = 0;
  next_node_id = 1;
}
Asset* Scene::find_asset(uint64_t id) { for (auto& a : assets) if (a.id == id) return &a; return nullptr; }
const Asset* Scene::find_asset(uint64_t id) const { for (auto& a : assets) if (a.id == id) return &a; return nullptr; }
uint64_t Scene::create_asset(Asset&& a, bool retain) {  // core/scene.hpp:196-207
  a.id = next_asset_id++;
  a.retain = retain;
  a.rc = 0;
  assets.push_back(std::move(a));
  return assets.back().id;
}
size_t Scene::create_node(const std::string& name, size_t parent, uint64_t id) {  // core/scene.cpp:335-351
  Node n;
  if (id == kNoId) id = next_node_id;
  next_node_id = std::max(next_node_id, id + 1);
  n.id = id; n.name = name; n.parent = parent;
  nodes.push_back(n);
  const size_t idx = nodes.size() - 1;
  if (parent != kNoNode) nodes[parent].children.push_back(idx);
  return idx;
}
void Scene::retain(uint64_t id) { if (Asset* a = find_asset(id)) a->rc++; }
void Scene::set_mesh(size_t node, uint64_t mesh_id) {  // core/scene.cpp:215-236 (MeshComponent: one empty slot)
  retain(mesh_id);
  nodes[node].mesh = mesh_id; nodes[node].has_mesh = true;
  nodes[node].materials.assign(1, kNoId);
}
void Scene::set_material(size_t node, size_t idx, uint64_t id) {  // :264-280
  if (!nodes[node].has_mesh) return;
  auto& m = nodes[node].materials;
  if (idx >= m.size()) m.resize(idx + 1, kNoId);
  if (id != kNoId) retain(id);
  m[idx] = id;
}

// traverseHierarchy (core/scene.cpp:514-534): LIFO stack, invisible nodes prune their subtree, world = parent * local
template <typename F>
static void traverse(const Scene& s, F&& cb) {
  std::vector<std::pair<size_t, Mat4>> stack;
  stack.emplace_back(s.root_index, mat_identity());
  while (!stack.empty()) {
    const auto [cur, parent] = stack.back();
    stack.pop_back();
    const Node& n = s.nodes[cur];
    if (!n.visible) continue;
    const Mat4 world = mat_mul(parent, n.transform.matrix());
    cb(cur, world);
    for (size_t child : n.children) stack.emplace_back(child, world);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// scene.json + _data.bin
// ---------------------------------------------------------------------------------------------------------------
static std::string bin_path_for(const std::string& json_path) {  // "{stem}_data.bin" next to the json (core/scene.cpp:33-34)
  const size_t slash = json_path.find_last_of('/');
  const std::string dir = slash == std::string::npos ? "" : json_path.substr(0, slash + 1);
  std::string file = slash == std::string::npos ? json_path : json_path.substr(slash + 1);
  const size_t dot = file.find_last_of('.');
  if (dot != std::string::npos && dot > 0) file = file.substr(0, dot);
  return dir + file + "_data.bin";
}
static std::string read_file(const std::string& path, bool binary = false) {
  std::ifstream f(path, binary ? std::ios::in | std::ios::binary : std::ios::in);
  if (!f) fail("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static size_t node_from_json(Scene& s, const JV& j, size_t parent) {  // core/scene.cpp:865-902
  const uint64_t id = j.at("id").u64();
  const size_t idx = parent == kNoNode ? s.root_index : s.create_node(j.at("name").str(), parent, id);
  if (parent == kNoNode) { s.nodes[idx].id = id; s.nodes[idx].name = j.at("name").str(); s.next_node_id = std::max(s.next_node_id, id + 1); }
  s.nodes[idx].visible = j.at("visible").boolean();
  {
    const JV& t = j.at("transform");  // utils/json.hpp:45-55
    Transform& tr = s.nodes[idx].transform;
    parse_floats(t.at("t"), tr.translation, 3); parse_floats(t.at("r"), tr.rotation, 3); parse_floats(t.at("s"), tr.scale, 3);
    parse_floats(t.at("tgt"), tr.target, 3);
    tr.track = t.at("track").boolean();
  }
  if (const JV* mesh = j.find("mesh")) {
    s.set_mesh(idx, mesh->at("id").u64());
    const JV& mats = mesh->at("materials");
    for (size_t i = 0; i < mats.a.size(); i++)
      if (!(mats.a[i].t == JV::STR && mats.a[i].s == "default")) s.set_material(idx, i, mats.a[i].u64());
  }
  if (const JV* cam = j.find("camera")) {  // Camera::withFocalLength(f, sensor, aperture): the other fields keep their defaults
    Camera c;
    c.focal_length = (float)cam->at("f").num();
    c.aperture = (float)cam->at("aperture").num();
    parse_floats(cam->at("sensor"), c.sensor_size, 2);
    s.nodes[idx].camera = c; s.nodes[idx].has_camera = true;
  }
  for (const JV& child : j.at("children").a) node_from_json(s, child, idx);
  return idx;
}

std::unique_ptr<Scene> load_json(const std::string& path) {  // core/scene.cpp:30-84
  auto sp = std::make_unique<Scene>();
  Scene& s = *sp;
  const JV data = json_parse(read_file(path));
  const std::string bin = read_file(bin_path_for(path), true);
  size_t cursor = 0;  // the reference reads the binary file strictly sequentially, in asset order (offsets are informative)
  auto take = [&](size_t len, void* dst) {
    if (cursor + len > bin.size()) fail("scene: " + bin_path_for(path) + " is shorter than the json says");
    memcpy(dst, bin.data() + cursor, len);
    cursor += len;
  };
  const JV& asset_data = data.at("assets");
  s.next_asset_id = asset_data.at("nextId").u64();
  for (const JV& aj : asset_data.at("assets").a) {
    Asset a;
    a.id = aj.at("id").u64();
    const std::string& type = aj.at("type").str();
    const JV& d = aj.at("data");
    if (type == "texture") {  // textureFromJson :789-817
      a.type = Asset::TEXTURE;
      const size_t len = (size_t)d.at("data").at(1).u64();
      a.tex.width = (uint32_t)d.at("size").at(0).u64();
      a.tex.height = (uint32_t)d.at("size").at(1).u64();
      a.tex.mtl_format = (uint32_t)d.at("format").u64();
      if (len != mtl_bytes_per_pixel(a.tex.mtl_format) * a.tex.width * a.tex.height) fail("scene: texture byte length does not match size x format");
      a.tex.bytes.resize(len);
      take(len, a.tex.bytes.data());
      a.tex.name = d.at("name").str();
      a.tex.alpha = d.at("alpha").boolean();
    } else if (type == "mesh") {  // meshFromJson :819-841
      a.type = Asset::MESH;
      const size_t lp = (size_t)d.at("positions").at(1).u64(), lv = (size_t)d.at("vertexData").at(1).u64();
      const size_t li = (size_t)d.at("indices").at(1).u64(), lm = (size_t)d.at("materials").at(1).u64();
      const size_t vc = (size_t)d.at("vertexCount").u64(), ic = (size_t)d.at("indexCount").u64();
      if (lp != vc * sizeof(pt_float3) || lv != vc * sizeof(pt_vertex_data) || li != ic * 4 || lm != ic / 3 * 4 || ic % 3)
        fail("scene: mesh buffer lengths do not match vertexCount/indexCount");
      a.mesh.positions.resize(vc); a.mesh.vdata.resize(vc); a.mesh.indices.resize(ic); a.mesh.slots.resize(ic / 3);
      take(lp, a.mesh.positions.data()); take(lv, a.mesh.vdata.data()); take(li, a.mesh.indices.data()); take(lm, a.mesh.slots.data());
    } else {  // materialFromJson :843-863
      a.type = Asset::MATERIAL;
      Material& m = a.mat;
      m.name = d.at("name").str();
      parse_floats(d.at("baseColor"), m.base_color, 4);
      parse_floats(d.at("emission"), m.emission, 3);
      m.emission_strength = (float)d.at("emissionStrength").num();
      m.roughness = (float)d.at("roughness").num(); m.metallic = (float)d.at("metallic").num();
      m.transmission = (float)d.at("transmission").num(); m.ior = (float)d.at("ior").num();
      m.anisotropy = (float)d.at("aniso").num(); m.anisotropy_rotation = (float)d.at("anisoRotation").num();
      m.clearcoat = (float)d.at("clearcoat").num(); m.clearcoat_roughness = (float)d.at("clearcoatRoughness").num();
      m.thin_transmission = d.at("thinTransmission").boolean();
      for (const JV& t : d.at("textures").a) m.set_texture((int)t.at(0).u64(), t.at(1).u64());
    }
    a.retain = aj.at("retain").boolean();
    a.rc = (uint32_t)aj.at("rc").u64();  // :58 takes the stored count; the node pass below retains again, as the reference does
    const uint64_t id = a.id;
    if (Asset* old = s.find_asset(id)) *old = std::move(a); else s.assets.push_back(std::move(a));
    s.next_asset_id = std::max(s.next_asset_id, id + 1);
  }
  node_from_json(s, data.at("root"), kNoNode);
  if (const JV* env = data.find("envmap")) {  // :70-79
    s.env_texture = env->at("texture").u64(); s.has_env = true;
    const size_t len = (size_t)env->at("aliasTable").at(0).u64();  // (sic) element 0 — the OFFSET — is what the reference reads as length
    const size_t real_len = (size_t)env->at("aliasTable").at(1).u64();
    // The reference allocates and reads `offset` bytes (core/scene.cpp:74-77), which only equals the table when offset ==
    // length.  We read the table that was written: `length` bytes at the sequential cursor.
    (void)len;
    if (real_len % sizeof(pt_alias_entry)) fail("scene: alias table length is not a multiple of 12");
    s.env_alias.resize(real_len / sizeof(pt_alias_entry));
    take(real_len, s.env_alias.data());
  }
  return sp;
}

static JV node_to_json(const Scene& s, size_t idx) {  // core/scene.cpp:633-681
  const Node& n = s.nodes[idx];
  JV j = jobj();
  j.o.emplace_back("id", juint(n.id));
  j.o.emplace_back("name", jstr(n.name));
  j.o.emplace_back("visible", jbool(n.visible));
  JV t = jobj();
  t.o.emplace_back("t", jvec(n.transform.translation, 3)); t.o.emplace_back("r", jvec(n.transform.rotation, 3));
  t.o.emplace_back("s", jvec(n.transform.scale, 3)); t.o.emplace_back("tgt", jvec(n.transform.target, 3));
  t.o.emplace_back("track", jbool(n.transform.track));
  j.o.emplace_back("transform", t);
  if (n.has_mesh) {
    JV m = jobj(), mats; mats.t = JV::ARR;
    for (uint64_t id : n.materials) mats.a.push_back(id == kNoId ? jstr("default") : juint(id));
    m.o.emplace_back("id", juint(n.mesh)); m.o.emplace_back("materials", mats);
    j.o.emplace_back("mesh", m);
  }
  if (n.has_camera) {
    JV c = jobj();
    c.o.emplace_back("f", jnum(n.camera.focal_length)); c.o.emplace_back("aperture", jnum(n.camera.aperture));
    c.o.emplace_back("sensor", jvec(n.camera.sensor_size, 2));
    j.o.emplace_back("camera", c);
  }
  JV ch; ch.t = JV::ARR;
  for (size_t c : n.children) ch.a.push_back(node_to_json(s, c));
  j.o.emplace_back("children", ch);
  return j;
}

void save_json(const Scene& s, const std::string& path) {  // core/scene.cpp:536-631
  std::ofstream bin(bin_path_for(path), std::ios::out | std::ios::binary);
  if (!bin) fail("cannot write " + bin_path_for(path));
  size_t off = 0;
  auto dump = [&](const void* p, size_t len) { bin.write((const char*)p, (std::streamsize)len); JV r = jarr({juint(off), juint(len)}); off += len; return r; };
  JV assets; assets.t = JV::ARR;
  for (const Asset& a : s.assets) {
    JV aj = jobj(), d = jobj();
    aj.o.emplace_back("id", juint(a.id)); aj.o.emplace_back("retain", jbool(a.retain)); aj.o.emplace_back("rc", juint(a.rc));
    if (a.type == Asset::TEXTURE) {
      aj.o.emplace_back("type", jstr("texture"));
      d.o.emplace_back("name", jstr(a.tex.name)); d.o.emplace_back("alpha", jbool(a.tex.alpha));
      d.o.emplace_back("size", jarr({juint(a.tex.width), juint(a.tex.height)})); d.o.emplace_back("format", juint(a.tex.mtl_format));
      d.o.emplace_back("data", dump(a.tex.bytes.data(), a.tex.bytes.size()));
    } else if (a.type == Asset::MESH) {
      aj.o.emplace_back("type", jstr("mesh"));
      d.o.emplace_back("indexCount", juint(a.mesh.indices.size())); d.o.emplace_back("vertexCount", juint(a.mesh.positions.size()));
      d.o.emplace_back("positions", dump(a.mesh.positions.data(), a.mesh.positions.size() * sizeof(pt_float3)));
      d.o.emplace_back("vertexData", dump(a.mesh.vdata.data(), a.mesh.vdata.size() * sizeof(pt_vertex_data)));
      d.o.emplace_back("indices", dump(a.mesh.indices.data(), a.mesh.indices.size() * 4));
      d.o.emplace_back("materials", dump(a.mesh.slots.data(), a.mesh.slots.size() * 4));
    } else {
      aj.o.emplace_back("type", jstr("material"));
      const Material& m = a.mat;
      d.o.emplace_back("name", jstr(m.name)); d.o.emplace_back("baseColor", jvec(m.base_color, 4));
      d.o.emplace_back("roughness", jnum(m.roughness)); d.o.emplace_back("metallic", jnum(m.metallic));
      d.o.emplace_back("transmission", jnum(m.transmission)); d.o.emplace_back("ior", jnum(m.ior));
      d.o.emplace_back("aniso", jnum(m.anisotropy)); d.o.emplace_back("anisoRotation", jnum(m.anisotropy_rotation));
      d.o.emplace_back("clearcoat", jnum(m.clearcoat)); d.o.emplace_back("clearcoatRoughness", jnum(m.clearcoat_roughness));
      d.o.emplace_back("emission", jvec(m.emission, 3)); d.o.emplace_back("emissionStrength", jnum(m.emission_strength));
      d.o.emplace_back("thinTransmission", jbool(m.thin_transmission));
      JV tx; tx.t = JV::ARR;
      for (const auto& st : m.textures) tx.a.push_back(jarr({juint((uint64_t)st.first), juint(st.second)}));
      d.o.emplace_back("textures", tx);
    }
    aj.o.emplace_back("data", d);
    assets.a.push_back(aj);
  }
  JV aroot = jobj();
  aroot.o.emplace_back("nextId", juint(s.next_asset_id)); aroot.o.emplace_back("assets", assets);
  JV root = jobj();
  root.o.emplace_back("root", node_to_json(s, s.root_index)); root.o.emplace_back("assets", aroot);
  if (s.has_env) {
    JV e = jobj();
    e.o.emplace_back("texture", juint(s.env_texture));
    e.o.emplace_back("aliasTable", dump(s.env_alias.data(), s.env_alias.size() * sizeof(pt_alias_entry)));
    root.o.emplace_back("envmap", e);
  }
  std::string text;
  json_dump(root, text);
  std::ofstream f(path);
  if (!f) fail("cannot write " + path);
  f << text;
}

// ---------------------------------------------------------------------------------------------------------------
// snapshot: rebuildResourceBuffers (renderer_pt.cpp:448-651) + getInstances + camera world transform
// ---------------------------------------------------------------------------------------------------------------
static pt_material_gpu to_material_gpu(const Scene& s, const Material& m, const std::map<uint64_t, int32_t>& tex_index) {  // :583-633
  pt_material_gpu g{};
  memcpy(g.baseColor, m.base_color, 16);
  g.emission = {m.emission[0], m.emission[1], m.emission[2], 0.0f};
  g.emissionStrength = m.emission_strength;
  g.roughness = m.roughness; g.metallic = m.metallic; g.transmission = m.transmission; g.ior = m.ior;
  g.anisotropy = m.anisotropy; g.anisotropyRotation = m.anisotropy_rotation;
  g.clearcoat = m.clearcoat; g.clearcoatRoughness = m.clearcoat_roughness;
  auto tid = [&](int slot) -> int32_t {
    const uint64_t id = m.get_texture(slot);
    if (id == kNoId) return -1;
    auto it = tex_index.find(id);
    return it == tex_index.end() ? 0 : it->second;  // m_textureIndices[id] default-inserts 0 for an unknown id
  };
  g.baseTextureId = tid(0); g.rmTextureId = tid(1); g.transmissionTextureId = tid(2); g.clearcoatTextureId = tid(3);
  g.emissionTextureId = tid(4); g.normalTextureId = tid(5);
  const uint64_t base = m.get_texture(0);
  const Asset* bt = base == kNoId ? nullptr : s.find_asset(base);
  int flags = 0;
  if (m.thin_transmission) flags |= PT_MATERIAL_THIN_DIELECTRIC;
  if (m.base_color[3] < 1.0f || (bt && bt->type == Asset::TEXTURE && bt->tex.alpha)) flags |= PT_MATERIAL_USE_ALPHA;
  if (m.anisotropy != 0.0f) flags |= PT_MATERIAL_ANISOTROPIC;
  const float e[3] = {m.emission[0] * m.emission_strength, m.emission[1] * m.emission_strength, m.emission[2] * m.emission_strength};
  if (dot3(e, e) > 0.0f || m.get_texture(4) != kNoId) flags |= PT_MATERIAL_EMISSIVE;
  g.flags = flags;
  return g;
}

const pt_scene_snapshot* Scene::build_snapshot(uint64_t camera_node) {
  Snapshot& S = snap;
  S = Snapshot{};
  std::map<uint64_t, uint32_t> mesh_index;
  std::map<uint64_t, int32_t> tex_index;
  for (const Asset& a : assets) {  // getAll<Mesh>() / getAll<Texture>(): asset-map (= insertion = file) order
    if (a.type == Asset::MESH) {
      mesh_index[a.id] = (uint32_t)S.meshes.size();
      S.meshes.push_back({a.mesh.positions.data(), a.mesh.vdata.data(), a.mesh.indices.data(), a.mesh.slots.data(),
                          (uint32_t)a.mesh.positions.size(), (uint32_t)(a.mesh.indices.size() / 3)});
    } else if (a.type == Asset::TEXTURE) {
      tex_index[a.id] = (int32_t)S.textures.size();
      S.textures.push_back({a.tex.bytes.data(), a.tex.width, a.tex.height, mtl_to_pt_format(a.tex.mtl_format), 0});
    }
  }
  const Material default_material{};
  const Node* cam_node = nullptr;
  Mat4 cam_world = mat_identity();
  std::vector<size_t> inst_nodes;
  traverse(*this, [&](size_t idx, const Mat4& world) {
    const Node& n = nodes[idx];
    if (!n.has_mesh) return;
    auto mi = mesh_index.find(n.mesh);
    if (mi == mesh_index.end()) fail("scene: node '" + n.name + "' references a missing mesh asset");
    pt_instance d{};
    for (int c = 0; c < 4; c++) for (int r = 0; r < 3; r++) d.transform[c][r] = world.c[c][r];
    d.mask = 0xFF;
    d.accelerationStructureIndex = mi->second;
    S.instances.push_back(d);
    inst_nodes.push_back(idx);
  });
  S.material_storage.resize(S.instances.size());
  for (size_t i = 0; i < S.instances.size(); i++) {
    const Node& n = nodes[inst_nodes[i]];
    const pt_mesh& pm = S.meshes[S.instances[i].accelerationStructureIndex];
    uint32_t max_slot = 0;
    for (uint32_t t = 0; t < pm.triangle_count; t++) max_slot = std::max(max_slot, pm.material_slots[t]);
    // The reference sizes the buffer by materialIds.size() and would read past it for a slot without an entry
    // (a trailing "default" is not stored, core/scene.cpp:881-883): such slots get the default material here.
    const size_t count = std::max<size_t>(n.materials.size(), (size_t)max_slot + 1);
    for (size_t k = 0; k < count; k++) {
      const uint64_t id = k < n.materials.size() ? n.materials[k] : kNoId;
      const Asset* a = id == kNoId ? nullptr : find_asset(id);
      const Material& m = (a && a->type == Asset::MATERIAL) ? a->mat : default_material;  // getMaterialOrDefault :1061-1069
      S.material_storage[i].push_back(to_material_gpu(*this, m, tex_index));
    }
  }
  for (size_t i = 0; i < S.instances.size(); i++)
    S.instance_materials.push_back({S.material_storage[i].data(), (uint32_t)S.material_storage[i].size(), 0});
  // camera: Scene::worldTransform(camera) (core/scene.cpp:463-474) — the product up the parent chain, visibility ignored
  for (const Node& n : nodes) if (n.id == camera_node && n.has_camera) cam_node = &n;
  if (!cam_node) fail("scene: node " + std::to_string(camera_node) + " is not a camera node");
  cam_world = cam_node->transform.matrix();
  for (size_t p = cam_node->parent; p != kNoNode; p = nodes[p].parent) cam_world = mat_mul(nodes[p].transform.matrix(), cam_world);
  pt_scene_snapshot& o = S.snapshot;
  memset(&o, 0, sizeof(o));
  o.meshes = S.meshes.data(); o.mesh_count = (uint32_t)S.meshes.size();
  o.instances = S.instances.data(); o.instance_count = (uint32_t)S.instances.size();
  o.instance_materials = S.instance_materials.data();
  memcpy(o.camera.world, cam_world.c, sizeof(float) * 16);
  const Camera& c = cam_node->camera;
  o.camera.sensor_size[0] = c.sensor_size[0]; o.camera.sensor_size[1] = c.sensor_size[1];
  o.camera.focal_length = c.focal_length; o.camera.aperture = c.aperture; o.camera.aperture_blades = c.aperture_blades;
  o.camera.roundness = c.roundness; o.camera.bokeh_power = c.bokeh_power; o.camera.focus_distance = c.focus_distance;
  o.textures = S.textures.empty() ? nullptr : S.textures.data();
  o.texture_count = (uint32_t)S.textures.size();
  o.env_texture = -1;
  o.env_alias = nullptr;
  if (has_env) {
    auto it = tex_index.find(env_texture);
    if (it == tex_index.end()) fail("scene: envmap references a missing texture asset");
    o.env_texture = it->second;
    const pt_texture& t = S.textures[it->second];
    if (!env_alias.empty()) {
      if (env_alias.size() != (size_t)t.width * t.height) fail("scene: alias table size does not match the environment texture");
      o.env_alias = env_alias.data();
    }
  }
  return &o;
}

void Scene::counts(pt_scene_counts* out) const {
  memset(out, 0, sizeof(*out));
  out->nodes = (uint32_t)nodes.size();
  for (const Asset& a : assets) { if (a.type == Asset::MESH) out->meshes++; else if (a.type == Asset::TEXTURE) out->textures++; else out->materials++; }
  traverse(*this, [&](size_t idx, const Mat4&) {
    const Node& n = nodes[idx];
    if (n.has_camera) out->cameras++;
    if (n.has_mesh) {
      out->instances++;
      if (const Asset* a = find_asset(n.mesh)) out->triangles += a->mesh.indices.size() / 3;
    }
  });
}
std::vector<size_t> Scene::cameras() const {
  std::vector<size_t> r;
  traverse(*this, [&](size_t idx, const Mat4&) { if (nodes[idx].has_camera) r.push_back(idx); });
  return r;
}

}  // namespace ptio

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
struct pt_scene { std::unique_ptr<ptio::Scene> s; };

namespace {
template <typename F>
int guarded(F&& f) {
  try { f(); return PT_OK; }
  catch (const std::bad_alloc&) { ptio::g_error = "out of memory"; return PT_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { ptio::g_error = e.what(); return PT_ERR_INVALID_ARGUMENT; }
}
}  // namespace

extern "C" {

const char* pt_scene_last_error(void) { return ptio::g_error.c_str(); }

int pt_scene_create(pt_scene** out) {
  if (!out) return PT_ERR_INVALID_ARGUMENT;
  return guarded([&] { *out = new pt_scene{std::make_unique<ptio::Scene>()}; });
}
void pt_scene_destroy(pt_scene* s) { delete s; }

int pt_scene_load_json(const char* path, pt_scene** out) {
  if (!path || !out) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  return guarded([&] { *out = new pt_scene{ptio::load_json(path)}; });
}
int pt_scene_save_json(const pt_scene* s, const char* path) {
  if (!s || !path) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] { ptio::save_json(*s->s, path); });
}
int pt_scene_import_gltf(pt_scene* s, const char* path, int options) {
  if (!s || !path) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] { ptio::import_gltf(*s->s, path, options); });
}
int pt_scene_set_environment(pt_scene* s, const float* rgba, uint32_t w, uint32_t h, const char* name) {
  if (!s || !rgba || !w || !h) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] {
    ptio::Asset a;
    a.type = ptio::Asset::TEXTURE;
    a.tex.name = name ? name : "environment"; a.tex.alpha = false; a.tex.width = w; a.tex.height = h; a.tex.mtl_format = PT_MTL_RGBA32FLOAT;
    a.tex.bytes.resize((size_t)w * h * 16);
    memcpy(a.tex.bytes.data(), rgba, a.tex.bytes.size());
    const uint64_t id = s->s->create_asset(std::move(a), true);
    s->s->retain(id);
    s->s->env_texture = id; s->s->has_env = true; s->s->env_alias.clear();  // rebuilt by pt_start_render (core/environment.cpp:5-91)
  });
}
int pt_scene_load_environment(pt_scene* s, const char* path) {
  if (!s || !path) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  uint32_t w = 0, h = 0;
  std::vector<float> px;
  const int rc = guarded([&] {
    const std::string p(path);
    const bool exr = p.size() >= 4 && p.compare(p.size() - 4, 4, ".exr") == 0;
    px = exr ? ptio::read_exr_rgba(p, &w, &h) : ptio::read_radiance_hdr_rgba(p, &w, &h);
  });
  if (rc != PT_OK) return rc;
  const char* slash = strrchr(path, '/');
  return pt_scene_set_environment(s, px.data(), w, h, slash ? slash + 1 : path);
}
int pt_scene_get_counts(const pt_scene* s, pt_scene_counts* out) {
  if (!s || !out) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] { s->s->counts(out); });
}
int pt_scene_get_camera(const pt_scene* s, uint32_t i, uint64_t* node_id, char* name, uint32_t cap) {
  if (!s || !node_id) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] {
    const auto cams = s->s->cameras();
    if (i >= cams.size()) throw std::runtime_error("camera index out of range");
    *node_id = s->s->nodes[cams[i]].id;
    if (name && cap) { strncpy(name, s->s->nodes[cams[i]].name.c_str(), cap - 1); name[cap - 1] = 0; }
  });
}
int pt_scene_add_camera(pt_scene* s, const char* name, const float position[3], const float target[3], float focal_mm, uint64_t* node_id) {
  if (!s || !position || !target) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] {
    const size_t idx = s->s->create_node(name ? name : "Camera", s->s->root_index, ptio::kNoId);
    ptio::Node& n = s->s->nodes[idx];
    memcpy(n.transform.translation, position, 12); memcpy(n.transform.target, target, 12);
    n.transform.track = true;
    n.camera = ptio::Camera{}; n.camera.focal_length = focal_mm; n.has_camera = true;
    if (node_id) *node_id = n.id;
  });
}
int pt_scene_build_snapshot(pt_scene* s, uint64_t camera_node, const pt_scene_snapshot** out) {
  if (!s || !out) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] { *out = s->s->build_snapshot(camera_node); });
}
int pt_generate_tangents(const pt_float3* positions, pt_vertex_data* vdata, uint32_t vertex_count, const uint32_t* indices, uint32_t triangle_count) {
  if (!positions || !vdata || !indices) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] {
    for (size_t k = 0; k < 3 * (size_t)triangle_count; k++) if (indices[k] >= vertex_count) throw std::runtime_error("tangents: vertex index out of range");
    ptio::generate_tangents(positions, vdata, vertex_count, indices, triangle_count);
  });
}

int pt_decode_image_rgba8(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba_out, uint64_t capacity) {
  if (!data || !width || !height) { ptio::g_error = "null argument"; return PT_ERR_INVALID_ARGUMENT; }
  return guarded([&] {
    uint32_t w = 0, h = 0;
    const std::vector<uint8_t> px = ptio::is_jpeg(data, (size_t)len) ? ptio::decode_jpeg_rgba8(data, (size_t)len, &w, &h)
                                                                      : ptio::decode_png_rgba8(data, (size_t)len, &w, &h);
    *width = w; *height = h;
    if (rgba_out) {
      if (capacity < px.size()) throw std::runtime_error("image: output buffer too small");
      memcpy(rgba_out, px.data(), px.size());
    }
  });
}

}  // extern "C"
