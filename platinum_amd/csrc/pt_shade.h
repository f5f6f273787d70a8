// pt_shade.h — per-path stage functions of the wavefront integrator: everything `misKernel`
// (kernel.metal:473-686) and `pathtracingKernel` (:256-372) do between two `intersect` calls.
//
//   stage_raygen  : HaltonSampler ctor + spawnRayFromCamera                     kernel.metal:195-238, 491-498
//   stage_shade   : getIntersectionData, BSDF sample, emitted + MIS, NEE set-up, kernel.metal:118-188, 545-669
//                   throughput, Russian roulette, next ray
// The callers (kernels.hip) own queues, compaction and memory traffic; these functions are pure per-path math so
// that tests/emu can run them on the host for debugging.
#pragma once
#include "pt_bvh.h"

namespace pt {

// kernel.metal:40-69
struct Frame {
  vec3 x, y, z;
  PT_HD vec3 worldToLocal(vec3 w) const { return {dot(w, x), dot(w, y), dot(w, z)}; }
  PT_HD vec3 localToWorld(vec3 l) const { return (x * l.x + y * l.y) + z * l.z; }
};
PT_HD Frame frame_from_normal(vec3 n) {  // :43-50
  const vec3 a = fabsf(n.x) > 0.5f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
  const vec3 b = normalize(cross(n, a));
  const vec3 t = cross(n, b);
  return {t, b, n};
}
PT_HD Frame frame_from_nt(vec3 n, vec3 t, float sgn) {  // :52-60
  if (fabsf(dot(n, t)) > 0.9f) return frame_from_normal(n);
  const vec3 b = normalize(cross(n, t)) * sgn;
  t = cross(b, n);
  return {t, b, n};
}

struct Xform { vec3 c0, c1, c2, c3; };
PT_HD Xform load_xform(const InstanceInfo& in) {
  return {v3(in.c0[0], in.c0[1], in.c0[2]), v3(in.c1[0], in.c1[1], in.c1[2]), v3(in.c2[0], in.c2[1], in.c2[2]),
          v3(in.c3[0], in.c3[1], in.c3[2])};
}
PT_HD vec3 transformVec(vec3 p, const Xform& m) { return (m.c0 * p.x + m.c1 * p.y) + m.c2 * p.z; }             // :10-13
PT_HD vec3 transformPoint(vec3 p, const Xform& m) { return ((m.c0 * p.x + m.c1 * p.y) + m.c2 * p.z) + m.c3; }  // :15-18
PT_HD vec3 ld3(const pt_float3& v) { return v3(v.x, v.y, v.z); }

// defs.metal:27-30
PT_HD vec3 interpolate3(vec3 a0, vec3 a1, vec3 a2, float u, float v) { return ((1.0f - u - v) * a0 + u * a1) + v * a2; }

// Which lobes of BSDF::sample / eval (bsdf.metal:199-252) a material can take — the divergence class of a hit:
//   0 opaque dielectric only, 1 + metallic, 2 + transmission, 3 clearcoat or anything textured that may switch lobes per texel.
PT_HD uint32_t material_class(const pt_material_gpu& m) {
  if (m.clearcoat > 0.0f || m.clearcoatTextureId >= 0 || m.rmTextureId >= 0 || m.transmissionTextureId >= 0) return 3u;
  if (m.transmission > 0.0f) return 2u;
  if (m.metallic > 0.0f) return 1u;
  return 0u;
}

// getIntersectionData's table walk (kernel.metal:118-141) and its geometric normal (:150-162), done once per flattened
// triangle instead of once per hit
PT_HD ShadeRec make_shade_rec(const DeviceScene& S, uint32_t instanceIdx, uint32_t prim) {
  const InstanceInfo& inst = S.instances[instanceIdx];
  const MeshInfo mesh = S.meshes[inst.mesh];
  const uint32_t* __restrict__ idx = &S.indices[3 * (size_t)(mesh.tri_base + prim)];
  ShadeRec r;
  r.v[0] = mesh.vertex_base + idx[0]; r.v[1] = mesh.vertex_base + idx[1]; r.v[2] = mesh.vertex_base + idx[2];
  r.material = inst.material_base + S.slots[mesh.tri_base + prim];
  r.inst = instanceIdx;
  const vec3 p0 = ld3(S.positions[r.v[0]]), p1 = ld3(S.positions[r.v[1]]), p2 = ld3(S.positions[r.v[2]]);
  const vec3 geometricNormal = normalize(cross(p1 - p0, p2 - p0));
  const vec3 ws = normalize(transformVec(geometricNormal, load_xform(inst)));
  r.ng[0] = ws.x; r.ng[1] = ws.y; r.ng[2] = ws.z;
  return r;
}

PT_HD LightRec make_light_rec(const DeviceScene& S, const pt_area_light& light) {
  const InstanceInfo& linst = S.instances[light.instanceIdx];
  const uint32_t vb = S.meshes[linst.mesh].vertex_base;
  const vec3 q0 = ld3(S.positions[vb + light.indices[0]]);
  const vec3 q1 = ld3(S.positions[vb + light.indices[1]]);
  const vec3 q2 = ld3(S.positions[vb + light.indices[2]]);
  const vec3 osNormal = cross(q1 - q0, q2 - q0);
  const vec3 n = normalize(transformVec(osNormal, load_xform(linst)));  // kernel.metal:420-423
  LightRec r;
  r.q0[0] = q0.x; r.q0[1] = q0.y; r.q0[2] = q0.z; r.area = light.area;
  r.q1[0] = q1.x; r.q1[1] = q1.y; r.q1[2] = q1.z; r.power = light.power;
  r.q2[0] = q2.x; r.q2[1] = q2.y; r.q2[2] = q2.z; r.cumulativePower = light.cumulativePower;
  for (int k = 0; k < 3; k++) { r.c0[k] = linst.c0[k]; r.c1[k] = linst.c1[k]; r.c2[k] = linst.c2[k]; r.c3[k] = linst.c3[k]; }
  r.e0 = light.emission.x; r.e1 = light.emission.y; r.e2 = light.emission.z;
  r.n[0] = n.x; r.n[1] = n.y; r.n[2] = n.z;
  r._pad0 = 0.0f; r._pad1 = 0.0f;
  return r;
}

// ---- raygen -----------------------------------------------------------------------------------------------------
struct RayGenOut { vec3 o, d; uint32_t offset, dim; };

PT_HD RayGenOut stage_raygen(const DeviceScene& S, uint32_t px, uint32_t py, uint32_t sample) {
  Halton h{halton_table(S.halton), halton_offset(px, py, sample), 0};
  const vec2 pixelSample = h.sample2d();  // dims 0-1 (kernel.metal:497)
  vec2 lensSample = {0.0f, 0.0f};         // dims 2-3: consumed always, evaluated only by a thin-lens camera
  if (S.camera.apertureRadius > 0.0f) lensSample = h.sample2d(); else h.dim += 2;
  const pt_camera_data& cam = S.camera;
  vec3 origin = ld3(cam.position);
  if (cam.apertureRadius > 0.0f) {  // kernel.metal:205-226
    vec2 lensPos = sampleDiskPolar(lensSample);
    lensPos.x = powr_det(lensPos.x, exp2_det(cam.bokehPower));
    if (cam.apertureRoundness < 1.0f) {
      const float n = (float)cam.apertureBlades;
      const float rPolygon = cos_det(kPi / n) / cos_det(fmodf(lensPos.y + 1.5f * kPi, 2.0f * kPi / n) - kPi / n);
      const float r = mix(rPolygon, 1.0f, cam.apertureRoundness);
      lensPos.x = lensPos.x * r;
    }
    float s, c;
    sincos_det(lensPos.y, &s, &c);
    const float lx = lensPos.x * c * cam.apertureRadius;
    const float ly = lensPos.x * s * cam.apertureRadius;
    origin = origin + (lx * normalize(ld3(cam.pixelDeltaU)) + ly * normalize(ld3(cam.pixelDeltaV)));
  }
  const float fx = (float)px + pixelSample.x;
  const float fy = (float)py + pixelSample.y;
  RayGenOut out;
  out.o = origin;
  out.d = normalize(((ld3(cam.topLeft) + fx * ld3(cam.pixelDeltaU)) + fy * ld3(cam.pixelDeltaV)) - origin);
  out.offset = h.offset;
  out.dim = h.dim;
  return out;
}

// ---- shade ------------------------------------------------------------------------------------------------------
struct ShadeIn {
  vec3 o, d;          // the ray that produced the hit
  vec3 att;           // attenuation (throughput) before this bounce
  const vec4* rayO;   // &state.rayO[slot], &state.rayD[slot]: the MIS weight of a light hit (rare) re-reads the ray origin,
  const vec4* rayD;   //   direction and lastSample.pdf from the queue instead of holding 7 registers across the whole stage
  bool lastSpecular;  // lastSample.flags & Sample_Specular
  uint32_t offset, dim;
  uint32_t bounce;
  float t, u, v;      // hit
  uint32_t tri;       // 2 * leaf slot + half: index into S.shade_recs
};
struct ShadeOut {
  vec3 emitted;        // to add to the path radiance now (already multiplied by attenuation and MIS weight)
  bool has_emitted;
  bool shadow;         // NEE shadow ray requested
  vec3 shadow_o, shadow_d;
  float shadow_tmax;
  vec3 shadow_contrib; // attenuation * Ld, to add if the shadow ray is unoccluded
  float shadow_payload;
  bool alive;          // path continues to bounce + 1
  vec3 next_o, next_d, next_att;
  float next_pdf;
  bool next_specular;
  uint32_t dim;
};

// The read-mostly tables of the shading stage.  k_shade stages the small, hot ones in LDS once per block (the Halton
// entries one bounce can touch, the light table, the E / Eavg / EavgMs energy tables); everywhere else they are the
// DeviceScene's own HBM copies.
struct ShadeTables {
  LutSet luts;
  HaltonTab halton;
  const LightRec* lights;
  const float* light_cdf;  // table in HBM (more than the staged 64 lights): the search reads this 4-byte-stride copy, not the 128-byte records
  int lights_lds;
};
PT_HD ShadeTables shade_tables(const DeviceScene& S) { return {S.luts, halton_table(S.halton), S.light_recs, S.light_cdf, 0}; }
PT_HD LightRec load_light(const ShadeTables& T, uint32_t i) { return T.lights_lds ? T.lights[i] : ldg(&T.lights[i]); }
PT_HD float light_cumulative_power(const ShadeTables& T, uint32_t i) {
  return T.lights_lds ? T.lights[i].cumulativePower : ldg(&T.light_cdf[i]);
}

// kernel.metal:20-25 rayDirToUv, :27-34 uvToRayDir
PT_HD vec2 rayDirToUv(vec3 dir) {
  const float phi = atan2_det(-dir.z, -dir.x);
  const float theta = acos_det(dir.y);
  return {phi / (2.0f * kPi), theta / kPi};
}
PT_HD vec3 uvToRayDir(vec2 uv) {
  float y, r, cosPhi, sinPhi;
  sincos_det(uv.y * kPi, &r, &y);
  sincos_det(uv.x * 2.0f * kPi, &sinPhi, &cosPhi);
  return normalize(v3(-cosPhi * r, y, -sinPhi * r));
}

struct EnvSample { vec3 Li, wi; float pdf; };
PT_HD EnvSample sample_environment(const DeviceScene& S, vec2 r) {  // kernel.metal:440-467
  const TexInfo t = ldg(&S.textures[S.env_texture]);
  const uint64_t w = t.w, h = t.h, n = w * h;
  uint64_t i = (uint64_t)(r.x * (float)n);
  if (i > n - 1) i = n - 1;
  if (r.y >= ldg(&S.env_alias[i].p)) i = ldg(&S.env_alias[i].aliasIdx);
  const uint64_t x = i % w, y = i / w;
  const vec2 uv = {(float)x / (float)w, (float)y / (float)h};
  const vec4 Le = tex_sample(S, S.env_texture, uv);
  EnvSample es;
  es.Li = v3(Le.x, Le.y, Le.z);
  es.wi = uvToRayDir(uv);
  es.pdf = ldg(&S.env_alias[i].pdf) / (4.0f * kPi);
  return es;
}

// Ray miss with an environment light (kernel.metal:517-539; :299-311 for the simple integrator): returns the radiance
// to add (already multiplied by the attenuation and the MIS weight).
PT_HD vec3 stage_miss(const DeviceScene& S, vec3 d, vec3 att, uint32_t bounce, float lastPdf, bool lastSpecular) {
  const TexInfo t = ldg(&S.textures[S.env_texture]);
  const vec2 uv = rayDirToUv(d);
  const vec4 s = tex_sample(S, S.env_texture, uv);
  const vec3 Le = v3(s.x, s.y, s.z);
  if (S.integrator != PT_INTEGRATOR_MIS || bounce == 0 || lastSpecular) return att * Le;
  // uint32_t x = w * uv.x: a negative product (half the sphere, atan2 range) converts to 0; indices are clamped
  const float fxw = (float)t.w * uv.x, fyh = (float)t.h * uv.y;
  uint32_t x = fxw > 0.0f ? (uint32_t)fxw : 0u, y = fyh > 0.0f ? (uint32_t)fyh : 0u;
  x = x < t.w - 1 ? x : t.w - 1;
  y = y < t.h - 1 ? y : t.h - 1;
  const float lightPdf = ldg(&S.env_alias[(size_t)y * t.w + x].pdf) * 0.25f * 0.318309886183790671538f;
  const float bsdfWeight = lastPdf / (lastPdf + lightPdf);
  return att * bsdfWeight * Le;
}

// kernel.metal:379-394
PT_HD uint32_t sampleLightPower(const DeviceScene& S, const ShadeTables& T, float r) {
  r = r * S.totalLightPower;
  uint32_t sz = S.lightCount - 1, idx = 0u;
  while (sz > 0) {
    const uint32_t h = sz >> 1, middle = idx + h;
    const bool res = light_cumulative_power(T, middle) < r;
    idx = res ? (middle + 1) : idx;
    sz = res ? sz - (h + 1) : h;
  }
  return idx > S.lightCount - 1 ? S.lightCount - 1 : idx;
}

// ---- the shading stage in three parts (k_shade runs them with a queue flush in between, which keeps the NEE results
//      out of the registers while the BSDF sample is computed; stage_shade() below chains them for the host harness) ----

// Resources::getIntersectionData (kernel.metal:118-188) + ShadingContext (bsdf.metal:12-43)
struct ShadeGeom {
  vec3 hitPos;
  Frame frame;
  vec3 wo;
};
PT_HD void shade_geometry(const DeviceScene& S, const ShadeIn& in, ShadeGeom& g, ShadingContext& ctx) {
  // (vertexResources / primitiveResources / instanceResources lookups, resolved per render into ShadeRec)
  const ShadeRec rec = ldg(&S.shade_recs[in.tri]);
  const InstanceInfo inst = ldg(&S.instances[rec.inst]);
  const pt_material_gpu material = ldg(&S.materials[rec.material]);
  const pt_vertex_data vd0 = ldg(&S.vdata[rec.v[0]]), vd1 = ldg(&S.vdata[rec.v[1]]), vd2 = ldg(&S.vdata[rec.v[2]]);
  const float tangentSign = vd0.tangent[3];

  const vec3 surfaceNormal = interpolate3(ld3(vd0.normal), ld3(vd1.normal), ld3(vd2.normal), in.u, in.v);
  const vec3 surfaceTangent = interpolate3(v3(vd0.tangent[0], vd0.tangent[1], vd0.tangent[2]),
                                           v3(vd1.tangent[0], vd1.tangent[1], vd1.tangent[2]),
                                           v3(vd2.tangent[0], vd2.tangent[1], vd2.tangent[2]), in.u, in.v);
  const float wb = 1.0f - in.u - in.v;  // interpolate(vertexTexCoords, bary), kernel.metal:143
  const vec2 surfaceUV = {(wb * vd0.texCoords[0] + in.u * vd1.texCoords[0]) + in.v * vd2.texCoords[0],
                          (wb * vd0.texCoords[1] + in.u * vd1.texCoords[1]) + in.v * vd2.texCoords[1]};

  const Xform objectToWorld = load_xform(inst);
  g.hitPos = in.o + in.d * in.t;
  const vec3 wsSurfaceNormal = normalize(transformVec(surfaceNormal, objectToWorld));
  const vec3 wsSurfaceTangent = normalize(transformVec(surfaceTangent, objectToWorld));
  g.frame = frame_from_nt(wsSurfaceNormal, wsSurfaceTangent, tangentSign);
  if (material.normalTextureId >= 0) {  // kernel.metal:166-175
    const vec4 t = tex_sample(S, material.normalTextureId, surfaceUV);
    const vec3 sampledNormal = v3(t.x, t.y, t.z) * 2.0f - v3(1.0f);
    g.frame = frame_from_normal(g.frame.localToWorld(sampledNormal));
  }
  g.wo = g.frame.worldToLocal(-in.d);
  ctx = make_shading_context(S, material, surfaceUV);
}

// Next-event estimation (kernel.metal:587-639).  `dim` = the first NEE dimension (the six BSDF-sample dimensions come
// before it in the schedule); returns the cursor after NEE in `dim`.
struct NeeOut {
  bool shadow;
  vec3 d;
  float tmax;
  vec3 contrib;
  float payload;
  uint32_t dim;
};
PT_HD NeeOut shade_nee(const DeviceScene& S, const ShadeTables& T, const ShadeIn& in, const ShadeGeom& g, const ShadingContext& ctx,
                       const BSDF& bsdf, uint32_t dim) {
  NeeOut out;
  out.shadow = false;
  out.dim = dim;
  if (!(S.integrator == PT_INTEGRATOR_MIS && (ctx.roughness > 0.0f || ctx.metallic + ctx.transmission < 1.0f))) return out;
  Halton halton{T.halton, in.offset, dim};
  const vec2 rl = halton.sample2d();
  float rz = halton.sample1d();
  out.dim = halton.dim;
  // kernel.metal:590-616. Without any light the reference indexes envLights[0] out of bounds (UB): NEE is skipped then,
  // the three dimensions are still consumed.
  const uint32_t envCount = S.envLightCount;
  if (!(S.lightCount > 0 || envCount > 0)) return out;
  const float pInfinite = S.lightCount == 0 ? 1.0f : (float)envCount / (float)(envCount + 1);
  vec3 Li, lpos, lwi;
  float lpdf, pLight;
  if (rz < pInfinite) {
    rz = rz / pInfinite;
    pLight = pInfinite / (float)envCount;
    const EnvSample es = sample_environment(S, rl);  // sampleEnvironmentLight, kernel.metal:440-467
    Li = es.Li; lpos = es.wi * 100.0f; lwi = es.wi; lpdf = es.pdf;
  } else {
    rz = (rz - pInfinite) / (1.0f - pInfinite);
    const LightRec light = load_light(T, sampleLightPower(S, T, rz));
    pLight = (1.0f - pInfinite) * light.power / S.totalLightPower;
    // sampleAreaLight (kernel.metal:407-435); vertices, transform and world-space normal come resolved in the record
    const vec3 q0 = v3(light.q0[0], light.q0[1], light.q0[2]), q1 = v3(light.q1[0], light.q1[1], light.q1[2]),
               q2 = v3(light.q2[0], light.q2[1], light.q2[2]);
    const vec2 sc = sampleTriUniform(rl);
    const Xform lx = {v3(light.c0[0], light.c0[1], light.c0[2]), v3(light.c1[0], light.c1[1], light.c1[2]),
                      v3(light.c2[0], light.c2[1], light.c2[2]), v3(light.c3[0], light.c3[1], light.c3[2])};
    lpos = transformPoint(interpolate3(q0, q1, q2, sc.x, sc.y), lx);
    const vec3 lnormal = v3(light.n[0], light.n[1], light.n[2]);
    lwi = normalize(lpos - g.hitPos);
    lpdf = length_squared(lpos - g.hitPos) / (fabsf(dot(lnormal, lwi)) * light.area);
    Li = v3(light.e0, light.e1, light.e2);
  }

  const vec3 wi = g.frame.worldToLocal(lwi);
  const BsdfEval ev = bsdf.eval(g.wo, wi);
  if (length_squared(ev.f) > 0.0f) {
    // `ir` payload of the shadow ray (kernel.metal:625): only evaluated when an alpha test can consume it
    out.payload = S.has_alpha ? halton.sample1d() : (halton.dim++, 0.0f);
    out.dim = halton.dim;
    const float pdfLight = pLight * lpdf;
    const vec3 Ld = Li * ev.f * fabsf(wi.z) / (pdfLight + ev.pdf);
    out.shadow = true;
    out.d = lwi;
    out.tmax = length(lpos - g.hitPos) - 1e-3f;
    out.contrib = in.att * Ld;
  }
  return out;
}

// BSDF sample (kernel.metal:550-556), light hit (:560-576), throughput, Russian roulette, next ray (:644-669).
// The sample's six dimensions start at in.dim; `dim_rr` = the cursor after NEE (where the roulette dimension sits).
struct BounceOut {
  vec3 emitted;
  bool has_emitted;
  bool alive;
  vec3 next_d, next_att;
  float next_pdf;
  bool next_specular;
  uint32_t dim;
};
// a throughput with a NaN or an infinity in it (conservative: also the last finite binade) — see k_shade's scan of the misses
PT_HD bool att_poisoned(vec3 a) { return !(fabsf(a.x) <= 3.0e38f && fabsf(a.y) <= 3.0e38f && fabsf(a.z) <= 3.0e38f); }
PT_HD BounceOut shade_bounce(const DeviceScene& S, const ShadeTables& T, const ShadeIn& in, const ShadeGeom& g, const ShadingContext& ctx,
                             const BSDF& bsdf, uint32_t dim_rr) {
  BounceOut out;
  out.has_emitted = false;
  out.alive = false;
  out.emitted = v3(0.0f);
  const bool mis = S.integrator == PT_INTEGRATOR_MIS;
  Halton halton{T.halton, in.offset, in.dim};
  const vec2 r01 = halton.sample2d();
  const float r2 = halton.sample1d();
  // r3 picks the lobe (bsdf.metal:228-252: r.w < pClearcoat / pMetallic / pTransparent).  Without a coat the thresholds
  // are m and m + (1-m)t; a threshold <= 0 is never passed and one >= 1 always is (r.w < 1), so r3 only matters when one
  // of them lies strictly inside (0,1) — not for a pure dielectric, a pure metal or pure glass.  The dimension is consumed
  // either way.
  float r3 = 0.0f;
  {
    const float m = ctx.metallic, pT = m + (1.0f - m) * ctx.transmission;
    if (ctx.clearcoat > 0.0f || (m > 0.0f && m < 1.0f) || (pT > 0.0f && pT < 1.0f)) r3 = halton.sample1d(); else halton.dim += 1;
  }
  // rc only feeds the clearcoat micro-normal (bsdf.metal:231-236): the two dimensions are always consumed, the radical
  // inverses only computed when a coat exists
  vec2 rc = {0.0f, 0.0f};
  if (ctx.clearcoat > 0.0f) rc = halton.sample2d(); else halton.dim += 2;
  const BsdfSample sample = bsdf.sample(g.wo, vec4{r01.x, r01.y, r2, r3}, rc);

  // ---- light hit (kernel.metal:560-576; :325-327 for the simple integrator) ----
  if (sample.flags & Sample_Emitted) {
    out.has_emitted = true;
    if (!mis || in.bounce == 0 || in.lastSpecular) {
      out.emitted = in.att * sample.Le;
    } else {
      // lastHit.pos is the origin of the current ray (kernel.metal:582, 666-669); the geometric normal comes from the
      // triangle's ShadeRec (re-read here: this branch is rare and three registers over the whole kernel are not)
      const ShadeRec rec = ldg(&S.shade_recs[in.tri]);
      const vec3 wsGeometricNormal = v3(rec.ng[0], rec.ng[1], rec.ng[2]);
      const vec4 o4 = ldg(in.rayO), d4 = ldg(in.rayD);
      const vec3 ro = v3(o4.x, o4.y, o4.z), rd = v3(d4.x, d4.y, d4.z);
      const float lastPdf = o4.w;
      const float lightPdf = (sample.Le.y * kPi / S.totalLightPower) * length_squared(ro - g.hitPos) /
                             fabsf(dot(rd, wsGeometricNormal));
      const float bsdfWeight = lastPdf / (lastPdf + lightPdf);
      out.emitted = in.att * bsdfWeight * sample.Le;
    }
  }

  halton.dim = dim_rr;
  out.dim = halton.dim;
  if (!(sample.flags & (Sample_Reflected | Sample_Transmitted))) return out;  // kernel.metal:644-645

  vec3 att = in.att * (sample.f * fabsf(sample.wi.z) / sample.pdf);  // :650
  if (in.bounce > 0) {                                               // :655-661
    const float q = fmaxf(0.0f, 1.0f - fmaxf(att.x, fmaxf(att.y, att.z)));
    if (q > 0.0f) {  // (q == 0: no sample is below it; the dimension is consumed without evaluating it)
      if (halton.sample1d() < q) { out.dim = halton.dim; return out; }
    } else halton.dim += 1;
    att = att / (1.0f - q);
  }
  out.dim = halton.dim;
  if (in.bounce + 1 >= S.max_bounces) return out;  // the `bounce < MAX_BOUNCES` loop bound (kernel.metal:509)

  out.alive = true;
  out.next_d = normalize(g.frame.localToWorld(sample.wi));  // :667
  out.next_att = att;
  out.next_pdf = sample.pdf;
  out.next_specular = (sample.flags & Sample_Specular) != 0;
  return out;
}

// The whole stage for one hit (host harness tests/emu; the kernel calls the three parts itself).  NEE is evaluated before
// the BSDF sample: every random number is addressed by its dimension, so the order of evaluation does not change a bit.
PT_HD ShadeOut stage_shade(const DeviceScene& S, const ShadeIn& in) {
  const ShadeTables T = shade_tables(S);
  ShadeGeom g;
  ShadingContext ctx;
  shade_geometry(S, in, g, ctx);
  const BSDF bsdf(ctx, S.flags, T.luts, g.wo);
  const NeeOut nee = shade_nee(S, T, in, g, ctx, bsdf, in.dim + 6);
  const BounceOut bo = shade_bounce(S, T, in, g, ctx, bsdf, nee.dim);
  ShadeOut out;
  out.emitted = bo.emitted;
  out.has_emitted = bo.has_emitted;
  out.shadow = nee.shadow;
  out.shadow_o = g.hitPos;
  out.shadow_d = nee.shadow ? nee.d : v3(0.0f);
  out.shadow_tmax = nee.shadow ? nee.tmax : 0.0f;
  out.shadow_contrib = nee.shadow ? nee.contrib : v3(0.0f);
  out.shadow_payload = nee.shadow ? nee.payload : 0.0f;
  out.alive = bo.alive;
  out.next_o = g.hitPos;
  // (shade_bounce leaves the next-ray fields unwritten for a path that ends here, NEE likewise: the kernel never reads them then; this chain
  //  must not copy indeterminate values either — UBSan on the host build, r4)
  out.next_d = bo.alive ? bo.next_d : v3(0.0f);
  out.next_att = bo.alive ? bo.next_att : v3(0.0f);
  out.next_pdf = bo.alive ? bo.next_pdf : 0.0f;
  out.next_specular = bo.alive ? bo.next_specular : false;
  out.dim = bo.dim;
  return out;
}

}  // namespace pt
