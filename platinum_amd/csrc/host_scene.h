// host_scene.h — host-side, once-per-render table building used by the HIP driver (renderer.hip): flattening of
// the snapshot and the reference's CPU-side derivations, restated in fp32 with a fixed operation order.
//   rebuildResourceBuffers   renderer_pt.cpp:448-651   (per-mesh / per-instance tables, MaterialGPU flags)
//   updateConstants          renderer_pt.cpp:965-1021  (camera frame, idt)
//   rebuildLightData         renderer_pt.cpp:838-917   (area-light table, cumulative power)
//   texture decode           (the MTLTexture pixel formats the reference samples: DESIGN.md "Texture contract")
//   rebuildAliasTable        core/environment.cpp:5-91 (environment importance table)
// Plain C++ (no HIP) so the tests/emu debugging harness can build it for the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_device.h"

namespace pt {

struct HostScene {
  std::vector<pt_float3> positions;
  std::vector<pt_vertex_data> vdata;
  std::vector<uint32_t> indices, slots;
  std::vector<MeshInfo> meshes;
  std::vector<InstanceInfo> instances;
  std::vector<pt_material_gpu> materials;
  std::vector<pt_area_light> lights;
  std::vector<uint8_t> tex_data;         // every texture in its own format, each at a multiple of 16 bytes (pt_device.h TexInfo)
  std::vector<float> tex_decode;         // unorm[256], srgb[256]: the floats 8-bit texels decode to
  std::vector<TexInfo> textures;
  uint32_t tex_native = 0;               // some texture is stored in an 8-bit form (DeviceScene::tex_native)
  std::vector<pt_alias_entry> env_alias;
  int32_t env_texture = -1;
  bool has_alpha = false;
  pt_constants constants{};
  Mat3 idt{};
  uint32_t tri_count = 0;
  // leaf slots of the one-BVH structure (pair_mesh_triangles): per unique mesh a list of primitives = one triangle or two consecutive
  // triangles that share an edge; prim_tri[mesh_prim_base[m] + k] = first triangle of primitive k of mesh m (local index) | kPrimPairBit
  std::vector<uint32_t> prim_tri, mesh_prim_base, inst_prim_base;
  uint32_t prim_count = 0;  // primitives of the flattened scene = inst_prim_base.back()
};
constexpr uint32_t kPrimPairBit = 0x80000000u;

// Groups the triangles of ONE mesh into leaf primitives: triangles t and t + 1 become one primitive when exactly two of t + 1's vertex
// indices occur in t (a shared edge: the two halves of a quad, as every generator and exporter emits them), greedily from the front.
// Same index = same object-space vertex = the same world-space point bit for bit after transformPoint, which is what lets the slot store
// four corners for two triangles (pt_device.h TriRec).  `pairs` = false: one triangle per primitive.
inline void pair_mesh_triangles(const uint32_t* idx, uint32_t tri_count, bool pairs, std::vector<uint32_t>* prim_tri) {
  for (uint32_t t = 0; t < tri_count;) {
    bool pair = false;
    if (pairs && t + 1 < tri_count) {
      const uint32_t* a = idx + 3 * (size_t)t;
      const uint32_t* b = a + 3;
      int shared = 0;
      for (int k = 0; k < 3; k++) shared += (b[k] == a[0] || b[k] == a[1] || b[k] == a[2]) ? 1 : 0;
      pair = shared == 2;
    }
    prim_tri->push_back(pair ? (t | kPrimPairBit) : t);
    t += pair ? 2u : 1u;
  }
}
inline void build_primitives(HostScene* hs, bool pairs) {
  hs->prim_tri.clear();
  hs->mesh_prim_base.assign(hs->meshes.size() + 1, 0u);
  for (size_t m = 0; m < hs->meshes.size(); m++) {
    hs->mesh_prim_base[m] = (uint32_t)hs->prim_tri.size();
    pair_mesh_triangles(hs->indices.data() + 3 * (size_t)hs->meshes[m].tri_base, hs->meshes[m].tri_count, pairs, &hs->prim_tri);
  }
  hs->mesh_prim_base[hs->meshes.size()] = (uint32_t)hs->prim_tri.size();
  hs->inst_prim_base.assign(hs->instances.size() + 1, 0u);
  uint32_t total = 0;
  for (size_t i = 0; i < hs->instances.size(); i++) {
    hs->inst_prim_base[i] = total;
    total += hs->mesh_prim_base[hs->instances[i].mesh + 1] - hs->mesh_prim_base[hs->instances[i].mesh];
  }
  hs->inst_prim_base[hs->instances.size()] = total;
  hs->prim_count = total;
}

inline int hs_fail(std::string* err, int code, const char* msg) {
  if (err) *err = msg;
  return code;
}

struct M3d { double m[3][3]; };  // [col][row]
inline M3d m3_mul(const M3d& a, const M3d& b) {
  M3d r{};
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a.m[k][rr] * b.m[c][k];
      r.m[c][rr] = s;
    }
  return r;
}
inline M3d m3_inv(const M3d& a) {
  const double(*m)[3] = a.m;
  const double c00 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  const double c01 = m[2][1] * m[0][2] - m[0][1] * m[2][2];
  const double c02 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
  const double id = 1.0 / (m[0][0] * c00 + m[1][0] * c01 + m[2][0] * c02);
  M3d r{};
  r.m[0][0] = c00 * id; r.m[0][1] = c01 * id; r.m[0][2] = c02 * id;
  r.m[1][0] = (m[2][0] * m[1][2] - m[1][0] * m[2][2]) * id;
  r.m[1][1] = (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * id;
  r.m[1][2] = (m[1][0] * m[0][2] - m[0][0] * m[1][2]) * id;
  r.m[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * id;
  r.m[2][1] = (m[2][0] * m[0][1] - m[0][0] * m[2][1]) * id;
  r.m[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * id;
  return r;
}
// core/colorspace.cpp:13-33: primaries + white point -> RGB->XYZ. Evaluated in double, rounded once (Apple's
// simd::inverse is closed, so the last bit of `idt` is unpinned either way).
inline M3d colorspace_to_xyz(const float r[2], const float g[2], const float b[2], const float w[2]) {
  const double prim[3][3] = {{r[0], r[1], 1.0 - (double)r[0] - (double)r[1]},
                             {g[0], g[1], 1.0 - (double)g[0] - (double)g[1]},
                             {b[0], b[1], 1.0 - (double)b[0] - (double)b[1]}};
  const double wx = w[0], wy = w[1], wz = 1.0 - wx - wy;
  const double W[3] = {wx / wy, 1.0, wz / wy};
  M3d mx{};
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) mx.m[c][rr] = prim[c][rr];
  const M3d inv = m3_inv(mx);
  double scale[3];
  for (int rr = 0; rr < 3; rr++) scale[rr] = inv.m[0][rr] * W[0] + inv.m[1][rr] * W[1] + inv.m[2][rr] * W[2];
  M3d out{};
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) out.m[c][rr] = mx.m[c][rr] * scale[c];
  return out;
}
// color::transform(BT709, working) (colorspace.hpp:61-63; renderer_pt.cpp:895, 1000-1002)
inline Mat3 compute_idt(const pt_colorspace& ws) {
  const float r709[2] = {0.640f, 0.330f}, g709[2] = {0.300f, 0.600f}, b709[2] = {0.150f, 0.060f}, d65[2] = {0.3127f, 0.3290f};
  const M3d src = colorspace_to_xyz(r709, g709, b709, d65);
  const M3d dst = colorspace_to_xyz(ws.r, ws.g, ws.b, ws.w);
  const M3d t = m3_mul(m3_inv(dst), src);
  Mat3 o;
  o.c0 = v3((float)t.m[0][0], (float)t.m[0][1], (float)t.m[0][2]);
  o.c1 = v3((float)t.m[1][0], (float)t.m[1][1], (float)t.m[1][2]);
  o.c2 = v3((float)t.m[2][0], (float)t.m[2][1], (float)t.m[2][2]);
  return o;
}
// color::transform(src, dst) for arbitrary colour spaces (the tonemap pass's odt = working -> output, renderer_pt.cpp:190)
inline Mat3 compute_transform(const pt_colorspace& src_cs, const pt_colorspace& dst_cs) {
  const M3d src = colorspace_to_xyz(src_cs.r, src_cs.g, src_cs.b, src_cs.w);
  const M3d dst = colorspace_to_xyz(dst_cs.r, dst_cs.g, dst_cs.b, dst_cs.w);
  const M3d t = m3_mul(m3_inv(dst), src);
  Mat3 o;
  o.c0 = v3((float)t.m[0][0], (float)t.m[0][1], (float)t.m[0][2]);
  o.c1 = v3((float)t.m[1][0], (float)t.m[1][1], (float)t.m[1][2]);
  o.c2 = v3((float)t.m[2][0], (float)t.m[2][1], (float)t.m[2][2]);
  return o;
}
inline pt_float3 to_pt(vec3 v) { return {v.x, v.y, v.z, 0.0f}; }
inline vec3 from_pt(const pt_float3& v) { return v3(v.x, v.y, v.z); }


// Texel decode (DESIGN.md "Texture contract"): UNORM8 -> i/255; sRGB8 colour channels through the piecewise sRGB EOTF
// evaluated in double and rounded once; R8 -> (r,0,0,1); RG8 -> (r,g,0,1); RGBA32F verbatim.  r4: the two 256-entry tables ARE the decode;
// the texels stay in their format and go through them on fetch (pt_bsdf.h tex_fetch), on the device and here alike.
constexpr size_t kTexDecodedBudget = 64u << 20;
inline vec4 decode_texel(const float* tab, uint32_t format, const uint8_t* b, size_t i) {
  const float* un = tab + kTexDecodeUnorm;
  if (format == PT_TEX_RGBA32F) { vec4 v; memcpy(&v, b + 16 * i, 16); return v; }
  if (format == PT_TEX_RGBA8_SRGB || format == PT_TEX_RGBA8) {
    const float* col = tab + (format == PT_TEX_RGBA8_SRGB ? kTexDecodeSrgb : kTexDecodeUnorm);
    return vec4{col[b[4 * i]], col[b[4 * i + 1]], col[b[4 * i + 2]], un[b[4 * i + 3]]};
  }
  if (format == PT_TEX_RG8) return vec4{un[b[2 * i]], un[b[2 * i + 1]], 0.0f, 1.0f};
  return vec4{un[b[i]], 0.0f, 0.0f, 1.0f};
}
inline uint32_t tex_bytes_per_texel(uint32_t format) {
  return format == PT_TEX_RGBA32F ? 16u : format == PT_TEX_RG8 ? 2u : format == PT_TEX_R8 ? 1u : 4u;
}
inline int decode_textures(const pt_scene_snapshot* scene, HostScene* out, std::string* err) {
  out->tex_decode.resize(kTexDecodeEntries);
  for (int i = 0; i < 256; i++) {
    const double c = (double)i / 255.0;
    out->tex_decode[kTexDecodeUnorm + i] = (float)i / 255.0f;
    out->tex_decode[kTexDecodeSrgb + i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
  }
  size_t texels = 0, texels8 = 0;
  for (uint32_t t = 0; t < scene->texture_count; t++) {
    const pt_texture& tx = scene->textures[t];
    if (!tx.pixels || tx.width == 0 || tx.height == 0 || tx.width > 32768 || tx.height > 32768)
      return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "texture: null pixels or bad size");
    if (tx.format < PT_TEX_RGBA8_SRGB || tx.format > PT_TEX_RGBA32F) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "texture: unknown format");
    texels += (size_t)tx.width * tx.height;
    if (tx.format != PT_TEX_RGBA32F) texels8 += (size_t)tx.width * tx.height;
  }
  if (texels >= (1ull << 32)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "textures: more than 2^32 texels");
  // Storage policy.  Decoded float4 texels cost one dependent load less per tap (the table look-up) and 4-16x the bytes.  While the 8-bit
  // textures of a scene decode into less than kTexDecodedBudget they are stored decoded (measured r4 on the atrium, whose four 64x64
  // material textures are cache-resident either way: k_shade +7.6 %, k_trace_shadow +5 % with the tables in the path); a scene with real
  // texture sets (a 2k x 2k RGBA8 map is 16 MB, 64 MB decoded) keeps them 8-bit so that they share the 256 MB Infinity Cache with the
  // acceleration structure instead of evicting it.  $PTAMD_TEX_NATIVE=0 / 1 forces the choice (the parity tests run both).
  bool native = texels8 * sizeof(vec4) > kTexDecodedBudget;
  if (const char* e = getenv("PTAMD_TEX_NATIVE")) native = atoi(e) != 0;
  size_t total = 0;
  for (uint32_t t = 0; t < scene->texture_count; t++) {
    const pt_texture& tx = scene->textures[t];
    total = (total + 15u) / 16u * 16u + (size_t)tx.width * tx.height * (native ? tex_bytes_per_texel(tx.format) : 16u);
  }
  if (total >= (1ull << 36)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "textures: more than 64 GB");
  out->tex_data.assign((total + 15u) / 16u * 16u, 0);
  out->textures.resize(scene->texture_count);
  size_t base = 0;
  for (uint32_t t = 0; t < scene->texture_count; t++) {
    const pt_texture& tx = scene->textures[t];
    const size_t n = (size_t)tx.width * tx.height;
    base = (base + 15u) / 16u * 16u;
    if (native || tx.format == PT_TEX_RGBA32F) {
      out->textures[t] = {(uint32_t)(base / 16u), tx.width, tx.height, tx.format};
      if (tx.format != PT_TEX_RGBA32F) out->tex_native = 1;
      memcpy(&out->tex_data[base], tx.pixels, n * tex_bytes_per_texel(tx.format));
      base += n * tex_bytes_per_texel(tx.format);
    } else {  // decoded once, through the same tables a fetch of the 8-bit form goes through
      const TexInfo as_given{0u, tx.width, tx.height, tx.format};
      out->textures[t] = {(uint32_t)(base / 16u), tx.width, tx.height, (uint32_t)PT_TEX_RGBA32F};
      vec4* dst = reinterpret_cast<vec4*>(&out->tex_data[base]);
      for (size_t i = 0; i < n; i++) dst[i] = decode_texel(out->tex_decode.data(), as_given.format, (const uint8_t*)tx.pixels, i);
      base += n * sizeof(vec4);
    }
  }
  return PT_OK;
}
// texel i of texture t as the linear float4 the sampler filters (the host twin of pt_bsdf.h tex_fetch)
inline vec4 host_texel(const HostScene& hs, const TexInfo& t, size_t i) {
  return decode_texel(hs.tex_decode.data(), t.format, &hs.tex_data[(size_t)t.offset16 * 16u], i);
}

// Environment::rebuildAliasTable (core/environment.cpp:5-91): importance = BT.709 luma of each texel, normalised to mean 1
// (that is EnvironmentLight::alias[i].pdf), then Vose's alias method with the reference's two LIFO work lists.
inline void build_env_alias(const HostScene& hs, const TexInfo& et, size_t n, std::vector<pt_alias_entry>* table) {
  table->assign(n, pt_alias_entry{0.0f, 0.0f, 0u});
  std::vector<float> q(n);
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) {
    const vec4 px = host_texel(hs, et, i);
    q[i] = dot(v3(px.x, px.y, px.z), v3(0.2126f, 0.7152f, 0.0722f));
    sum += q[i];
  }
  const float scale = (float)n / sum;
  std::vector<size_t> under, over;
  for (size_t i = 0; i < n; i++) {
    q[i] *= scale;
    (*table)[i].pdf = q[i];
    (q[i] < 1.0f ? under : over).push_back(i);
  }
  while (!under.empty() && !over.empty()) {
    const size_t lo = under.back(), hi = over.back();
    under.pop_back(); over.pop_back();
    (*table)[lo].p = q[lo];
    (*table)[lo].aliasIdx = (uint32_t)hi;
    q[hi] = (q[hi] + q[lo]) - 1.0f;
    (q[hi] < 1.0f ? under : over).push_back(hi);
  }
  for (size_t i : over) (*table)[i].p = 1.0f;
  for (size_t i : under) (*table)[i].p = 1.0f;
}

inline int build_host_scene(const pt_scene_snapshot* scene, const pt_render_params* p, uint32_t lut_w_E, uint32_t lut_w_Eavg,
                            HostScene* out, std::string* err) {
  // ---- flatten the snapshot (rebuildResourceBuffers, renderer_pt.cpp:448-651) ----
  std::vector<MeshInfo>& meshes = out->meshes;
  meshes.assign(scene->mesh_count, MeshInfo{});
  size_t nv = 0, nt = 0;
  for (uint32_t m = 0; m < scene->mesh_count; m++) {
    const pt_mesh& pm = scene->meshes[m];
    if (pm.vertex_count && (!pm.positions || !pm.vertex_data)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "mesh: null vertex arrays");
    if (pm.triangle_count && (!pm.indices || !pm.material_slots)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "mesh: null index arrays");
    meshes[m] = {(uint32_t)nv, (uint32_t)nt, pm.triangle_count, 0};
    nv += pm.vertex_count;
    nt += pm.triangle_count;
  }
  if (nv >= (1ull << 31) || nt >= (1ull << 30)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "scene too large for 32-bit indices");
  std::vector<pt_float3>& positions = out->positions;
  std::vector<pt_vertex_data>& vdata = out->vdata;
  std::vector<uint32_t>&indices = out->indices, &slots = out->slots;
  positions.resize(nv); vdata.resize(nv); indices.resize(3 * nt); slots.resize(nt);
  for (uint32_t m = 0; m < scene->mesh_count; m++) {
    const pt_mesh& pm = scene->meshes[m];
    if (pm.vertex_count) {
      memcpy(&positions[meshes[m].vertex_base], pm.positions, sizeof(pt_float3) * pm.vertex_count);
      memcpy(&vdata[meshes[m].vertex_base], pm.vertex_data, sizeof(pt_vertex_data) * pm.vertex_count);
    }
    if (pm.triangle_count) {
      memcpy(&indices[3 * (size_t)meshes[m].tri_base], pm.indices, sizeof(uint32_t) * 3 * pm.triangle_count);
      memcpy(&slots[meshes[m].tri_base], pm.material_slots, sizeof(uint32_t) * pm.triangle_count);
      for (size_t k = 0; k < 3 * (size_t)pm.triangle_count; k++)
        if (pm.indices[k] >= pm.vertex_count) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "mesh: vertex index out of range");
    }
  }
  // ---- textures and the environment light (N3) ----
  if (scene->texture_count && !scene->textures) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "scene: null texture array");
  if (const int rc = decode_textures(scene, out, err)) return rc;
  out->env_texture = -1;
  out->env_alias.clear();
  if (scene->env_texture >= 0) {
    if ((uint32_t)scene->env_texture >= scene->texture_count) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "scene: env_texture out of range");
    out->env_texture = scene->env_texture;
    const TexInfo& et = out->textures[out->env_texture];
    const size_t n = (size_t)et.w * et.h;
    if (scene->env_alias) out->env_alias.assign(scene->env_alias, scene->env_alias + n);  // the host's own table, verbatim
    else build_env_alias(*out, et, n, &out->env_alias);
  }
  out->has_alpha = false;

  std::vector<InstanceInfo>& instances = out->instances;
  instances.assign(scene->instance_count, InstanceInfo{});
  std::vector<pt_material_gpu>& materials = out->materials;
  materials.clear();
  uint64_t tri_total = 0;
  for (uint32_t i = 0; i < scene->instance_count; i++) {
    const pt_instance& in = scene->instances[i];
    if (in.accelerationStructureIndex >= scene->mesh_count) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "instance: mesh index out of range");
    const pt_instance_materials& im = scene->instance_materials[i];
    if (im.material_count && !im.materials) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "instance: null material array");
    InstanceInfo ii{};
    for (int k = 0; k < 3; k++) { ii.c0[k] = in.transform[0][k]; ii.c1[k] = in.transform[1][k]; ii.c2[k] = in.transform[2][k]; ii.c3[k] = in.transform[3][k]; }
    ii.mesh = in.accelerationStructureIndex;
    ii.material_base = (uint32_t)materials.size();
    ii.tri_global_base = (uint32_t)tri_total;
    instances[i] = ii;
    const pt_mesh& pm = scene->meshes[ii.mesh];
    for (uint32_t t = 0; t < pm.triangle_count; t++)
      if (pm.material_slots[t] >= im.material_count) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "instance: material slot out of range");
    for (uint32_t k = 0; k < im.material_count; k++) {
      pt_material_gpu mat = im.materials[k];
      const int32_t ids[6] = {mat.baseTextureId, mat.rmTextureId, mat.transmissionTextureId, mat.clearcoatTextureId,
                              mat.emissionTextureId, mat.normalTextureId};
      for (int32_t id : ids)
        if (id >= 0 && (uint32_t)id >= scene->texture_count) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "material: texture id out of range");
      // renderer_pt.cpp:626-633: the Renderer derives these two flags when it fills MaterialGPU
      // (Material::isEmissive, core/material.hpp:44-47: non-zero emission or an emission texture)
      const vec3 e = from_pt(mat.emission) * mat.emissionStrength;
      if (length_squared(e) > 0.0f || mat.emissionTextureId >= 0) mat.flags |= PT_MATERIAL_EMISSIVE;
      if (mat.anisotropy != 0.0f) mat.flags |= PT_MATERIAL_ANISOTROPIC;
      // renderer_pt.cpp:714-729: an instance with any alpha-tested material is built non-opaque
      if (mat.flags & PT_MATERIAL_USE_ALPHA) { instances[i].flags |= kInstanceNonOpaque; out->has_alpha = true; }
      materials.push_back(mat);
    }
    tri_total += pm.triangle_count;
  }
  if (tri_total >= (1ull << 31)) return hs_fail(err, PT_ERR_INVALID_ARGUMENT, "too many flattened triangles");
  out->tri_count = (uint32_t)tri_total;

  // ---- constants (updateConstants, renderer_pt.cpp:965-1021) ----
  const Mat3 idt = compute_idt(p->working_space);
  out->idt = idt;
  const pt_camera& cam = scene->camera;
  pt_constants& C = out->constants;
  memset(&C, 0, sizeof(C));
  {
    vec3 col[4];
    for (int i = 0; i < 4; i++) col[i] = v3(cam.world[i][0], cam.world[i][1], cam.world[i][2]);
    const vec3 u = col[0] / length(col[0]);  // "rescale the camera transform to ignore any scaling" (:970-976)
    const vec3 v = col[1] / length(col[1]);
    const vec3 w = col[2] / length(col[2]);
    const vec3 pos = col[3];
    const float sx = (float)p->width, sy = (float)p->height;
    const float aspect = sx / sy;
    const float sensorAspect = cam.sensor_size[0] / cam.sensor_size[1];
    const float cropped = cam.sensor_size[0] / fmaxf(sensorAspect, aspect);  // camera.hpp:47-50
    const float vh = cam.focus_distance * cropped / cam.focal_length;
    const float vw = vh * aspect;
    const vec3 vu = u * vw;
    const vec3 vv = -v * vh;
    C.spp = p->spp;
    C.gmonBuckets = (p->flags & PT_FLAG_GMON) ? p->gmon_buckets : 1;  // renderer_pt.cpp:993-994
    C.lutSizeE = lut_w_E;
    C.lutSizeEavg = lut_w_Eavg;
    C.flags = p->flags;
    C.size[0] = p->width;
    C.size[1] = p->height;
    C.idt[0] = to_pt(idt.c0); C.idt[1] = to_pt(idt.c1); C.idt[2] = to_pt(idt.c2);
    C.camera.position = to_pt(pos);
    C.camera.topLeft = to_pt((pos - cam.focus_distance * w) - (vu + vv) * 0.5f);
    C.camera.pixelDeltaU = to_pt(vu / sx);
    C.camera.pixelDeltaV = to_pt(vv / sy);
    C.camera.apertureRadius = cam.aperture > 0.0f ? (cam.focal_length / 2000.0f) / cam.aperture : 0.0f;
    C.camera.apertureBlades = cam.aperture_blades;
    C.camera.apertureRoundness = cam.roundness;
    C.camera.bokehPower = cam.bokeh_power;
  }

  // ---- area lights (rebuildLightData, renderer_pt.cpp:838-917) ----
  out->lights.clear();
  float totalPower = 0.0f;
  for (uint32_t i = 0; i < scene->instance_count; i++) {
    const InstanceInfo& ii = instances[i];
    const pt_mesh& pm = scene->meshes[ii.mesh];
    const pt_instance_materials& im = scene->instance_materials[i];
    bool any = false;
    for (uint32_t k = 0; k < im.material_count; k++) any = any || (materials[ii.material_base + k].flags & PT_MATERIAL_EMISSIVE);
    if (!any) continue;
    const vec3 c0 = v3(ii.c0[0], ii.c0[1], ii.c0[2]), c1 = v3(ii.c1[0], ii.c1[1], ii.c1[2]);
    const vec3 c2 = v3(ii.c2[0], ii.c2[1], ii.c2[2]), c3 = v3(ii.c3[0], ii.c3[1], ii.c3[2]);
    auto xf = [&](const pt_float3& q) { return ((c0 * q.x + c1 * q.y) + c2 * q.z) + c3; };
    for (uint32_t t = 0; t < pm.triangle_count; t++) {
      const pt_material_gpu& mat = materials[ii.material_base + pm.material_slots[t]];
      if (!(mat.flags & PT_MATERIAL_EMISSIVE)) continue;
      const uint32_t i0 = pm.indices[3 * t], i1 = pm.indices[3 * t + 1], i2 = pm.indices[3 * t + 2];
      const vec3 v0 = xf(pm.positions[i0]), v1 = xf(pm.positions[i1]), v2 = xf(pm.positions[i2]);
      const float area = length(cross(v1 - v0, v2 - v0)) * 0.5f;
      const vec3 emission = mul(idt, from_pt(mat.emission)) * mat.emissionStrength;
      const float power = emission.y * area * kPi;  // dot(emission, (0,1,0)) * area * pi
      totalPower += power;
      pt_area_light L{};
      L.instanceIdx = i; L.indices[0] = i0; L.indices[1] = i1; L.indices[2] = i2;
      L.area = area; L.power = power; L.cumulativePower = totalPower; L.emission = to_pt(emission);
      out->lights.push_back(L);
    }
  }
  C.lightCount = (uint32_t)out->lights.size();
  C.envLightCount = out->env_texture >= 0 ? 1 : 0;  // one EnvironmentLight per scene environment (renderer_pt.cpp:919-938)
  C.totalLightPower = totalPower;

  return PT_OK;
}

}  // namespace pt
