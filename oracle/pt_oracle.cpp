// TEST INFRASTRUCTURE — oracle/pt_oracle.cpp
// CPU restatement (scalar, one path at a time, megakernel-shaped like the reference) of the hot path of
// teofum/platinum's progressive path tracer: src/renderer_pt/shaders/{kernel,bsdf,samplers,defs}.metal and the
// host-side table builders of src/renderer_pt/renderer_pt.cpp.  Every function cites the lines it follows
// (paths relative to /root/reference/src/renderer_pt unless stated).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/call this library.  It is the
// checker, never the thing shipped: the product (platinum_amd/csrc) shares no code with it.
//
// PARITY PINNING: the reference has no tests, golden vectors or fixtures and cannot be built or run
// in this pipeline (Metal/macOS only).  What IS pinned: the integer sampler known-answers of SURVEY §8a
// (tests/golden/sampler_kat.json), the LUT data files themselves (decoded by the reference's vendored tinyexr,
// tools/make_lut_blob.py --check-tinyexr), closed-form checks (fresnel(1,1.5)=0.04, white-furnace of E) and — round 2 — the BSDF
// pieces (GGX D / G / VNDF sampling, Fresnel, the dielectric lobes) against the reference's committed energy tables: namespace lutgen
// below restates the reference's generator (ms_lut_gen.metal:337-743) on top of THIS file's BSDF functions and re-integrates all
// eight tables (tests/test_lut_pin.py, tools/lut_pin.py: agreement to ~3e-4).
// Ray/triangle intersection and texture filtering are Apple-closed in the reference and are DEFINED
// here: => "parity unpinned" for hit selection, LUT interpolation rounding and the float radiance.
// (Round 3: tests/test_gpu_physics.py anchors the integrator as a whole in physics — analytic irradiance under a panel light to 0.3 %, white
// furnace, MIS against SIMPLE — through the HIP path, which is bit-identical to this file on every small case.  A bound, not a reference pin.)
//
// Intersection contract (ours): triangles are flattened to world space in fp32 (transformPoint below), tested
// with the Moeller-Trumbore sequence in intersect_triangle(); closest hit = minimum t in [tmin, tmax], ties
// broken by the lowest (instance, primitive); any-hit = exists t in [tmin, tmax].  Both are independent of the
// acceleration structure, so brute force and any conservative BVH give identical answers.
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/ptamd.h"
#include "oracle_math.h"
#include "pt_oracle.h"

using namespace orc;

namespace {

// ------------------------------------------------------------------------------------------------------------
// samplers.metal / defs.metal
// ------------------------------------------------------------------------------------------------------------

constexpr float oneMinusEpsilon = 0x1.fffffep-1;  // defs.metal:22
constexpr int kNumPrimes = 620;                   // defs.metal:115-194 (620 entries, last 4583)

struct Primes {
  uint32_t p[kNumPrimes];
  Primes() {
    int n = 0;
    for (uint32_t c = 2; n < kNumPrimes; c++) {
      bool prime = true;
      for (uint32_t d = 2; d * d <= c; d++)
        if (c % d == 0) { prime = false; break; }
      if (prime) p[n++] = c;
    }
  }
};
const Primes g_primes;

struct uint4_ { uint32_t x, y, z, w; };

// samplers.metal:16-23
inline uint4_ pcg4d(uint4_ v) {
  v.x = v.x * 1664525u + 1013904223u;
  v.y = v.y * 1664525u + 1013904223u;
  v.z = v.z * 1664525u + 1013904223u;
  v.w = v.w * 1664525u + 1013904223u;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  v.x ^= v.x >> 16u; v.y ^= v.y >> 16u; v.z ^= v.z >> 16u; v.w ^= v.w >> 16u;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  return v;
}

// samplers.metal:154-184 (HaltonSampler), decl defs.metal:107-200
struct HaltonSampler {
  uint32_t m_offset;
  uint32_t m_dim = 0;
  HaltonSampler(uint32_t tx, uint32_t ty, uint32_t sample) {
    m_offset = pcg4d({tx, ty, sample, tx + ty}).x;  // samplers.metal:154-156
  }
  static float halton(uint32_t i, uint32_t d) {  // samplers.metal:168-184
    uint32_t b = g_primes.p[d];
    float f = 1.0f;
    float invB = 1.0f / (float)b;
    float r = 0;
    while (i > 0) {
      f = f * invB;
      r = r + f * (float)(i % b);
      i = i / b;
    }
    return fminf(r, oneMinusEpsilon);
  }
  float sample1d() { return halton(m_offset, m_dim++); }
  float2 sample2d() {
    float x = halton(m_offset, m_dim++);
    float y = halton(m_offset, m_dim++);
    return {x, y};
  }
};

// samplers.metal:200-207
inline float2 sampleDisk(float2 u) {
  const float r = sqrtf(u.x);
  const float theta = 2.0f * PI_F * u.y;
  float c, s;
  sincos_det(theta, &s, &c);
  return {r * c, r * s};
}
// samplers.metal:209-214
inline float2 sampleDiskPolar(float2 u) {
  const float r = sqrtf(u.x);
  const float theta = 2.0f * PI_F * u.y;
  return {r, theta};
}
// samplers.metal:216-225
inline float3 sampleCosineHemisphere(float2 u) {
  const float phi = u.x * 2.0f * PI_F;
  const float sinTheta = sqrtf(u.y);
  const float cosTheta = sqrtf(1.0f - u.y);
  float cosPhi, sinPhi;
  sincos_det(phi, &sinPhi, &cosPhi);
  return {cosPhi * sinTheta, sinPhi * sinTheta, cosTheta};
}
// samplers.metal:227-238
inline float2 sampleTriUniform(float2 u) {
  float b0, b1;
  if (u.x < u.y) {
    b0 = u.x * 0.5f;
    b1 = u.y - b0;
  } else {
    b1 = u.y * 0.5f;
    b0 = u.x - b1;
  }
  return {b0, b1};
}

// ------------------------------------------------------------------------------------------------------------
// kernel.metal:40-69  Frame
// ------------------------------------------------------------------------------------------------------------
struct Frame {
  float3 x, y, z;
  static Frame fromNormal(float3 n) {  // :43-50
    float3 a = fabsf(n.x) > 0.5f ? f3(0, 0, 1) : f3(1, 0, 0);
    float3 b = normalize(cross(n, a));
    float3 t = cross(n, b);
    return {t, b, n};
  }
  static Frame fromNT(float3 n, float3 t, float sign = 1.0f) {  // :52-60
    if (fabsf(dot(n, t)) > 0.9f) return fromNormal(n);
    float3 b = normalize(cross(n, t)) * sign;
    t = cross(b, n);
    return {t, b, n};
  }
  float3 worldToLocal(float3 w) const { return {dot(w, x), dot(w, y), dot(w, z)}; }  // :62-64
  float3 localToWorld(float3 l) const { return (x * l.x + y * l.y) + z * l.z; }       // :66-68
};

// defs.metal:27-30 interpolate: (1 - u - v) * a0 + u * a1 + v * a2
inline float3 interpolate(const float3* att, float2 uv) {
  return ((1.0f - uv.x - uv.y) * att[0] + uv.x * att[1]) + uv.y * att[2];
}
inline float2 interpolate(const float2* att, float2 uv) {
  float w = 1.0f - uv.x - uv.y;
  return {(w * att[0].x + uv.x * att[1].x) + uv.y * att[2].x, (w * att[0].y + uv.x * att[1].y) + uv.y * att[2].y};
}

// A 4x3 object->world transform as four columns (kernel.metal:103-110 getTransform rebuilds the 4x4).
struct Xform { float3 c0, c1, c2, c3; };
// kernel.metal:10-13 transformVec: M * (p, 0)
inline float3 transformVec(float3 p, const Xform& m) { return (m.c0 * p.x + m.c1 * p.y) + m.c2 * p.z; }
// kernel.metal:15-18 transformPoint: M * (p, 1)
inline float3 transformPoint(float3 p, const Xform& m) { return ((m.c0 * p.x + m.c1 * p.y) + m.c2 * p.z) + m.c3; }

// ------------------------------------------------------------------------------------------------------------
// LUTs (pt_shader_defs.hpp:130-139; sampled with clamp_to_edge + linear, defs.metal:352, bsdf.metal:264,297,316)
// Software filtering contract (ours; Apple's texture unit is closed): unnormalised coordinate x = c*N - 0.5,
// i0 = floor(x), w = x - i0, both taps clamped to [0, N-1], lerp a + (b - a) * w, x first, then y, then z.
// ------------------------------------------------------------------------------------------------------------
struct Lut {
  const float* d = nullptr;
  int w = 0, h = 0, depth = 0;
};
struct LutSet { Lut E, Eavg, EMs, EavgMs, ETransIn, ETransOut, EavgTransIn, EavgTransOut; };

inline void lut_axis(float c, int n, int* i0, int* i1, float* w) {
  float x = c * (float)n - 0.5f;
  float fl = floorf(x);
  *w = x - fl;
  int i = (int)fl;
  int a = i, b = i + 1;
  a = a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
  b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
  *i0 = a; *i1 = b;
}
inline float lut1(const Lut& l, float cx) {
  int x0, x1; float wx;
  lut_axis(cx, l.w, &x0, &x1, &wx);
  return l.d[x0] + (l.d[x1] - l.d[x0]) * wx;
}
inline float lut2_slice(const float* d, int W, int H, float cx, float cy) {
  int x0, x1, y0, y1; float wx, wy;
  lut_axis(cx, W, &x0, &x1, &wx);
  lut_axis(cy, H, &y0, &y1, &wy);
  float a = d[y0 * W + x0] + (d[y0 * W + x1] - d[y0 * W + x0]) * wx;
  float b = d[y1 * W + x0] + (d[y1 * W + x1] - d[y1 * W + x0]) * wx;
  return a + (b - a) * wy;
}
inline float lut2(const Lut& l, float cx, float cy) { return lut2_slice(l.d, l.w, l.h, cx, cy); }
inline float lut3(const Lut& l, float cx, float cy, float cz) {
  int z0, z1; float wz;
  lut_axis(cz, l.depth, &z0, &z1, &wz);
  float a = lut2_slice(l.d + (size_t)z0 * l.w * l.h, l.w, l.h, cx, cy);
  float b = lut2_slice(l.d + (size_t)z1 * l.w * l.h, l.w, l.h, cx, cy);
  return a + (b - a) * wz;
}

// ------------------------------------------------------------------------------------------------------------
// Scene textures.  Filtering contract (ours; Apple's texture unit is closed): sampler(address::repeat,
// filter::linear) with normalised coordinates: x = u*W - 0.5, i0 = floor(x), w = x - i0, both taps wrapped modulo W,
// lerp a + (b - a) * w along x then y.  Texels are decoded to linear float4 up front: UNORM8 = i / 255, sRGB8 through
// the exact piecewise curve evaluated in double and rounded once (alpha stays linear), R8 -> (r,0,0,1), RG8 -> (r,g,0,1).
// ------------------------------------------------------------------------------------------------------------
struct Tex { int w = 0, h = 0; std::vector<float> px; };  // px: h*w*4

inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

float4 tex_sample(const Tex& t, float2 uv) {
  const float fx = uv.x * (float)t.w - 0.5f, fy = uv.y * (float)t.h - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx = fx - x0f, wy = fy - y0f;
  const int x0 = wrapi((int)x0f, t.w), x1 = wrapi((int)x0f + 1, t.w);
  const int y0 = wrapi((int)y0f, t.h), y1 = wrapi((int)y0f + 1, t.h);
  const float* p00 = &t.px[4 * ((size_t)y0 * t.w + x0)];
  const float* p01 = &t.px[4 * ((size_t)y0 * t.w + x1)];
  const float* p10 = &t.px[4 * ((size_t)y1 * t.w + x0)];
  const float* p11 = &t.px[4 * ((size_t)y1 * t.w + x1)];
  float o[4];
  for (int c = 0; c < 4; c++) {
    const float a = p00[c] + (p01[c] - p00[c]) * wx;
    const float b = p10[c] + (p11[c] - p10[c]) * wx;
    o[c] = a + (b - a) * wy;
  }
  return {o[0], o[1], o[2], o[3]};
}

// ------------------------------------------------------------------------------------------------------------
// bsdf.metal
// ------------------------------------------------------------------------------------------------------------
enum SampleFlags {  // defs.metal:264-272
  Sample_Absorbed = 0, Sample_Emitted = 1 << 0, Sample_Reflected = 1 << 1, Sample_Transmitted = 1 << 2,
  Sample_Diffuse = 1 << 3, Sample_Glossy = 1 << 4, Sample_Specular = 1 << 5,
};

struct Mat3 { float3 c0, c1, c2; };
inline float3 mul(const Mat3& m, float3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }

// defs.metal:283-299 + bsdf.metal:12-43
struct ShadingContext {
  float3 albedo; float roughness, metallic, transmission, clearcoat, clearcoatRoughness, anisotropy, ior;
  int flags; float3 emission;
  ShadingContext(const pt_material_gpu& mat, float2 uv, const Mat3& idt, const std::vector<Tex>& textures) {
    albedo = f3(mat.baseColor[0], mat.baseColor[1], mat.baseColor[2]);
    emission = f3(mat.emission.x, mat.emission.y, mat.emission.z);
    roughness = mat.roughness; metallic = mat.metallic; transmission = mat.transmission;
    clearcoat = mat.clearcoat; clearcoatRoughness = mat.clearcoatRoughness; anisotropy = mat.anisotropy;
    ior = mat.ior; flags = mat.flags;
    if (mat.baseTextureId >= 0) { float4 t = tex_sample(textures[mat.baseTextureId], uv); albedo = f3(t.x, t.y, t.z); }          // :25-26
    if (mat.emissionTextureId >= 0) { float4 t = tex_sample(textures[mat.emissionTextureId], uv); emission *= f3(t.x, t.y, t.z); }  // :27-28
    if (mat.transmissionTextureId >= 0) transmission = tex_sample(textures[mat.transmissionTextureId], uv).x;                     // :29-30
    if (mat.clearcoatTextureId >= 0) clearcoat = tex_sample(textures[mat.clearcoatTextureId], uv).x;                              // :31-32
    if (mat.rmTextureId >= 0) {                                                                                                    // :33-37
      float4 rm = tex_sample(textures[mat.rmTextureId], uv);
      roughness *= rm.x;
      metallic *= rm.y;
    }
    albedo = mul(idt, albedo);      // :40
    emission = mul(idt, emission);  // :41
    emission *= mat.emissionStrength;  // :42
  }
};

struct Sample { float3 wi = f3(0); float3 f = f3(0); float3 Le = f3(0); float pdf = 0.0f; int flags = 0; };  // defs.metal:301-307
struct Eval {  // defs.metal:309-328 — note the default pdf = 1 (quirk preserved: `return {}` leaks pdf 1)
  float3 f = f3(0); float3 Le = f3(0); float pdf = 1.0f;
  Eval& operator+=(const Eval& e) { f += e.f; Le += e.Le; pdf += e.pdf; return *this; }
  Eval operator+(const Eval& e) const { return {f + e.f, Le + e.Le, pdf + e.pdf}; }
  Eval operator*(float c) const { return {f * c, Le * c, pdf * c}; }
};

inline float3 schlick(float3 f0, float cosTheta) {  // bsdf.metal:49-53
  const float k = 1.0f - cosTheta;
  const float k2 = k * k;
  return f0 + (f3(1.0f) - f0) * (k2 * k2 * k);
}
inline float fresnel(float cosTheta, float ior) {  // bsdf.metal:72-85
  cosTheta = saturate(cosTheta);
  const float sin2Theta_t = (1.0f - cosTheta * cosTheta) / (ior * ior);
  if (sin2Theta_t >= 1.0f) return 1.0f;
  const float cosTheta_t = sqrtf(1.0f - sin2Theta_t);
  const float parallel = (ior * cosTheta - cosTheta_t) / (ior * cosTheta + cosTheta_t);
  const float perpendicular = (cosTheta - ior * cosTheta_t) / (cosTheta + ior * cosTheta_t);
  return (parallel * parallel + perpendicular * perpendicular) * 0.5f;
}
inline float avgDielectricFresnelFit(float ior) {  // bsdf.metal:92-96
  return ior >= 1.0f ? (ior - 1.0f) / (4.08567f + 1.00071f * ior)
                     : 0.997118f + 0.1014f * ior - 0.965241f * ior * ior - 0.130607f * ior * ior * ior;
}

struct GGX {  // bsdf.metal:102-183
  float ax, ay;
  explicit GGX(float roughness) { ax = ay = roughness * roughness; }  // :103
  GGX(float roughness, float anisotropic) {                           // :105-109
    const float alpha = roughness * roughness;
    const float aspect = sqrtf(1.0f - 0.9f * anisotropic);
    ax = alpha / aspect; ay = alpha * aspect;
  }
  float lambda(float3 w) const {  // :173-182
    const float cos2Theta = w.z * w.z;
    float alpha2 = ax * ax;
    if (ax != ay) alpha2 = alpha2 * w.x * w.x + ay * ay * w.y * w.y;
    return (sqrtf(1.0f + alpha2 / cos2Theta) - 1.0f) * 0.5f;
  }
  float mdf(float3 w) const {  // :112-122
    const float cos2Theta = w.z * w.z;
    const float cos4Theta = cos2Theta * cos2Theta;
    float k = 1.0f / cos2Theta * (w.x * w.x / (ax * ax) + w.y * w.y / (ay * ay));
    k = (1.0f + k) * (1.0f + k);
    return 1.0f / (PI_F * ax * ay * cos4Theta * k);
  }
  float g1(float3 w) const { return 1.0f / (1.0f + lambda(w)); }                          // :125
  float g(float3 wo, float3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }  // :128-130
  float vmdf(float3 w, float3 wm) const { return g1(w) / fabsf(w.z) * mdf(wm) * fabsf(dot(w, wm)); }  // :133-135
  float3 sampleVmdf(float3 w, float2 u) const {  // :138-157
    float3 wh = normalize(w * f3(ax, ay, 1.0f));
    if (wh.z < 0) wh *= -1.0f;
    const float3 b = (wh.z < 0.9999f) ? normalize(cross(f3(0.0f, 0.0f, 1.0f), wh)) : f3(1.0f, 0.0f, 0.0f);
    const float3 t = cross(wh, b);
    float2 p = sampleDisk(u);
    const float h = sqrtf(1.0f - p.x * p.x);
    p.y = mix(h, p.y, 0.5f * wh.z + 0.5f);
    const float pz = sqrtf(fmaxf(0.0f, 1.0f - length_squared(p)));
    const float3 nh = (p.x * b + p.y * t) + pz * wh;
    return normalize(f3(ax * nh.x, ay * nh.y, fmaxf(1e-6f, nh.z)));
  }
  float singleScatterBRDF(float3 wo, float3 wi, float3 wm) const {  // :159-162
    return mdf(wm) * g(wo, wi) / (4 * fabsf(wo.z) * fabsf(wi.z));
  }
  float pdf(float3 wo, float3 wm) const { return vmdf(wo, wm) / (4.0f * fabsf(dot(wo, wm))); }  // :164-166
  bool isSmooth() const { return ax < 1e-3f && ay < 1e-3f; }                                     // :171-173
};

struct BSDF {  // defs.metal:330-394, bsdf.metal:190-715
  ShadingContext& m_ctx;
  GGX m_ggx, m_ggxCoat;
  int m_flags;  // constants.flags
  const LutSet& m_luts;
  static constexpr float m_clearcoatIor = 1.5f;

  BSDF(ShadingContext& ctx, int rendererFlags, const LutSet& luts)  // bsdf.metal:190-193
      : m_ctx(ctx), m_ggx(ctx.roughness, ctx.anisotropy), m_ggxCoat(ctx.clearcoatRoughness), m_flags(rendererFlags), m_luts(luts) {}

  // defs.metal:349-361 multiscatter<T>
  float3 multiscatter(float3 wo, float3 wi, float3 F_avg) const {
    const float E_wo = lut2(m_luts.E, wo.z, m_ctx.roughness);
    const float E_wi = lut2(m_luts.E, wi.z, m_ctx.roughness);
    const float E_avg = lut1(m_luts.Eavg, m_ctx.roughness);
    const float brdf_ms = (1.0f - E_wo) * (1.0f - E_wi) / (PI_F * (1.0f - E_avg));
    const float3 fresnel_ms = F_avg * F_avg * E_avg / (f3(1.0f) - F_avg * (1.0f - E_avg));
    return fresnel_ms * brdf_ms;
  }
  float multiscatter(float3 wo, float3 wi, float F_avg) const {
    const float E_wo = lut2(m_luts.E, wo.z, m_ctx.roughness);
    const float E_wi = lut2(m_luts.E, wi.z, m_ctx.roughness);
    const float E_avg = lut1(m_luts.Eavg, m_ctx.roughness);
    const float brdf_ms = (1.0f - E_wo) * (1.0f - E_wi) / (PI_F * (1.0f - E_avg));
    const float fresnel_ms = F_avg * F_avg * E_avg / (1.0f - F_avg * (1.0f - E_avg));
    return fresnel_ms * brdf_ms;
  }
  // bsdf.metal:262-284
  float transparentMultiscatter(float3 wo, float3 /*wi*/, float ior) const {
    if (ior < 1.0f) {
      const float iorParam = 1.0f - ior;
      const float E_wo = lut3(m_luts.ETransOut, fabsf(wo.z), m_ctx.roughness, iorParam);
      return 1.0f / E_wo;
    } else {
      const float iorParam = (ior - 1.0f) / ior;
      const float E_wo = lut3(m_luts.ETransIn, fabsf(wo.z), m_ctx.roughness, iorParam);
      return 1.0f / E_wo;
    }
  }
  // bsdf.metal:291-305
  float diffuseFactor(float3 wo, float3 wi) const {
    const float iorParam = (m_ctx.ior - 1.0f) / m_ctx.ior;
    const float E_ms_wo = lut3(m_luts.EMs, wo.z, m_ctx.roughness, iorParam);
    const float E_ms_wi = lut3(m_luts.EMs, wi.z, m_ctx.roughness, iorParam);
    const float E_ms_avg = lut2(m_luts.EavgMs, iorParam, m_ctx.roughness);
    return (1.0f - E_ms_wo) * (1.0f - E_ms_wi) / (PI_F * (1.0f - E_ms_avg));
  }
  // bsdf.metal:311-326
  float opaqueDielectricFactor(float3 wo, float F_avg) const {
    const float iorParam = (m_ctx.ior - 1.0f) / m_ctx.ior;
    const float E_wo = lut2(m_luts.E, wo.z, m_ctx.roughness);
    const float E_ms_wo = lut3(m_luts.EMs, wo.z, m_ctx.roughness, iorParam);
    const float fresnel_ms = F_avg * F_avg * E_wo / (1.0f - F_avg * (1.0f - E_wo));
    const float dielectricFactor = F_avg * E_ms_wo + fresnel_ms * (1.0f - E_ms_wo);
    return dielectricFactor;
  }

  // ---- eval ------------------------------------------------------------------------------------------------
  Eval evalMetallic(float3 wo, float3 wi, float3 wm) const {  // bsdf.metal:339-355
    const float3 fresnel_ss = schlick(m_ctx.albedo, fabsf(dot(wo, wm)));
    float3 brdf = fresnel_ss * m_ggx.singleScatterBRDF(wo, wi, wm);
    if (m_flags & PT_FLAG_MULTISCATTER_GGX) {
      const float3 F_avg = (20.0f * m_ctx.albedo + f3(1.0f)) / 21.0f;
      brdf += multiscatter(wo, wi, F_avg);
    }
    Eval e; e.f = brdf; e.Le = f3(0); e.pdf = m_ggx.pdf(wo, wm);
    return e;
  }
  Eval evalMetallic(float3 wo, float3 wi) const {  // bsdf.metal:360-370
    if (m_ggx.isSmooth()) return {};
    float3 wm = normalize(wo + wi);
    if (length_squared(wm) == 0.0f) return {};
    wm *= sign(wm.z);
    return evalMetallic(wo, wi, wm);
  }
  Eval evalTransparentDielectric(float3 wo, float3 wi, float3 wm, float fresnel_ss, float ior) const {  // :377-419
    const bool thin = m_ctx.flags & PT_MATERIAL_THIN_DIELECTRIC;
    const bool isReflection = wo.z * wi.z > 0.0f;
    float3 bsdf;
    float pdf, k = fresnel_ss;
    if (isReflection) {
      bsdf = f3(m_ggx.singleScatterBRDF(wo, wi, wm));
      pdf = m_ggx.pdf(wo, wm);
    } else {
      k = 1.0f - fresnel_ss;
      float btdf_ss;
      if (thin) {
        btdf_ss = m_ggx.singleScatterBRDF(wo, wi, wm);
        pdf = m_ggx.pdf(wo, wm);
      } else {
        float denom = dot(wi, wm) * ior + dot(wo, wm);
        denom *= denom;
        const float dwm_dwi = fabsf(dot(wi, wm)) / denom;
        btdf_ss = m_ggx.mdf(wm) * m_ggx.g(wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * denom));
        pdf = m_ggx.vmdf(wo, wm) * dwm_dwi;
      }
      bsdf = m_ctx.albedo * btdf_ss;
    }
    if (m_flags & PT_FLAG_MULTISCATTER_GGX) bsdf *= transparentMultiscatter(wo, wi, ior);
    Eval e; e.f = k * bsdf; e.Le = f3(0); e.pdf = k * pdf;
    return e;
  }
  Eval evalTransparentDielectric(float3 wo, float3 wi) const {  // bsdf.metal:424-446
    if (m_ggx.isSmooth()) return {};
    const bool thin = m_ctx.flags & PT_MATERIAL_THIN_DIELECTRIC;
    const float ior = (!thin && wo.z < 0.0f && wi.z < 0.0f) ? 1.0f / m_ctx.ior : m_ctx.ior;
    float3 wm = ior * wi + wo;
    if (wi.z == 0 || wo.z == 0 || wm.z == 0) return {};
    wm = normalize(wm * sign(wm.z));
    if (dot(wi, wm) * wi.z < 0.0f || dot(wo, wm) * wo.z < 0.0f) return {};
    if (thin) {
      wi = reflect(wi, f3(0.0f, 0.0f, 1.0f));
      wm = normalize(wi + wo);
    }
    const float fresnel_ss = fresnel(dot(wo, wm), ior);
    return evalTransparentDielectric(wo, wi, wm, fresnel_ss, ior);
  }
  Eval evalOpaqueDielectric(float3 wo, float3 wi) const {  // bsdf.metal:451-486
    const float F_avg = avgDielectricFresnelFit(m_ctx.ior);
    const float blendingFactor = opaqueDielectricFactor(wo, F_avg);
    const float cDiffuse = diffuseFactor(wo, wi);
    const float diffusePdf = fabsf(wi.z) / PI_F;
    if (m_ggx.isSmooth()) {
      Eval e; e.f = m_ctx.albedo * cDiffuse; e.Le = f3(0); e.pdf = diffusePdf * (1.0f - blendingFactor);
      return e;
    }
    float3 wm = normalize(wo + wi);
    if (length_squared(wm) == 0.0f) return {};
    wm *= sign(wm.z);
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), m_ctx.ior);
    float dielectricBrdf = fresnel_ss * m_ggx.singleScatterBRDF(wo, wi, wm);
    if (m_flags & PT_FLAG_MULTISCATTER_GGX) dielectricBrdf += multiscatter(wo, wi, F_avg);
    Eval e;
    e.f = f3(dielectricBrdf) + m_ctx.albedo * cDiffuse;
    e.Le = f3(0);
    e.pdf = m_ggx.pdf(wo, wm) * blendingFactor + diffusePdf * (1.0f - blendingFactor);
    return e;
  }
  // bsdf.metal:488-503. `fresnel_ss` is left unwritten by the reference on the early returns (UB there);
  // the caller below initialises it to 0.
  Eval evalClearcoat(float3 wo, float3 wi, float& fresnel_ss) const {
    if (m_ggxCoat.isSmooth()) return {};
    float3 wm = wo + wi;
    wm = normalize(wm * sign(wm.z));
    if (length_squared(wm) == 0.0f) return {};
    fresnel_ss = fresnel(dot(wo, wm), m_clearcoatIor);
    Eval e; e.f = f3(m_ggxCoat.singleScatterBRDF(wo, wi, wm)); e.Le = f3(0); e.pdf = m_ggxCoat.pdf(wo, wm);
    return e;
  }
  Eval eval(float3 wo, float3 wi) const {  // bsdf.metal:199-223
    if (wo.z < 1.5e-3f || wi.z < 1.5e-3f) return {};
    float metallic = m_ctx.metallic;
    float transparent = (1.0f - metallic) * m_ctx.transmission;
    float opaque = (1.0f - metallic) * (1.0f - transparent);
    Eval result; result.f = f3(0); result.Le = f3(0); result.pdf = 0.0f;
    if (metallic > 0.0f) result += evalMetallic(wo, wi) * metallic;
    if (transparent > 0.0f) result += evalTransparentDielectric(wo, wi) * transparent;
    if (opaque > 0.0f) result += evalOpaqueDielectric(wo, wi) * opaque;
    float coat = m_ctx.clearcoat;
    if (coat > 0.0f) {
      float coatFresnel_ss = 0.0f;
      Eval coatResult = evalClearcoat(wo, wi, coatFresnel_ss);
      coat *= coatFresnel_ss;
      result = result * (1.0f - coat) + coatResult * coat;
    }
    return result;
  }

  // ---- sample ----------------------------------------------------------------------------------------------
  Sample sampleMetallic(float3 wo, float3 r) const {  // bsdf.metal:513-543
    if (m_ggx.isSmooth()) {
      const float3 fresnel_ss = schlick(m_ctx.albedo, wo.z);
      Sample s; s.flags = Sample_Reflected | Sample_Specular; s.f = fresnel_ss / fabsf(wo.z);
      s.wi = f3(-wo.x, -wo.y, wo.z); s.pdf = 1.0f;
      return s;
    }
    float3 wm = m_ggx.sampleVmdf(wo, {r.x, r.y});
    float3 wi = reflect(-wo, wm);
    if (wo.z * wi.z < 0.0f) return {};
    const Eval eval = evalMetallic(wo, wi, wm);
    Sample s; s.flags = Sample_Reflected | Sample_Glossy; s.wi = wi; s.f = eval.f; s.pdf = eval.pdf;
    return s;
  }
  Sample sampleTransparentDielectric(float3 wo, float3 r) const {  // bsdf.metal:550-619
    const bool thin = m_ctx.flags & PT_MATERIAL_THIN_DIELECTRIC;
    float ior = (wo.z < 0.0f && !thin) ? 1.0f / m_ctx.ior : m_ctx.ior;
    if (m_ggx.isSmooth()) {
      const float fresnel_ss = fresnel(fabsf(wo.z), ior);
      float3 wi, color = f3(1.0f);
      float pdf = fresnel_ss;
      int flags = Sample_Specular;
      if (r.z < fresnel_ss) {
        wi = f3(-wo.x, -wo.y, wo.z);
        flags |= Sample_Reflected;
      } else {
        wi = thin ? -wo : refract(-wo, f3(0.0f, 0.0f, sign(wo.z)), 1.0f / ior);
        if (wi.z == 0.0f) return {};
        pdf = (1.0f - fresnel_ss);
        color = m_ctx.albedo;
        flags |= Sample_Transmitted;
      }
      Sample s; s.flags = flags; s.wi = wi; s.f = pdf * color / fabsf(wi.z); s.Le = f3(0.0f); s.pdf = pdf;
      return s;
    }
    const float3 wm = m_ggx.sampleVmdf(wo, {r.x, r.y});
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ior);
    float3 wi;
    int flags = Sample_Glossy;
    if (r.z < fresnel_ss) {
      wi = reflect(-wo, wm);
      if (wo.z * wi.z < 0.0f) return {};
      flags |= Sample_Reflected;
    } else if (thin) {
      wi = reflect(-wo, wm) * f3(1.0f, 1.0f, -1.0f);
      flags |= Sample_Transmitted;
    } else {
      wi = refract(-wo, wm * sign(dot(wo, wm)), 1.0f / ior);
      if (wo.z * wi.z >= 0.0f) return {};
      flags |= Sample_Transmitted;
    }
    const Eval eval = evalTransparentDielectric(wo, wi, wm, fresnel_ss, ior);
    Sample s; s.flags = flags; s.wi = wi; s.f = eval.f; s.Le = eval.Le; s.pdf = eval.pdf;
    return s;
  }
  Sample sampleOpaqueDielectric(float3 wo, float3 r) const {  // bsdf.metal:626-684
    const float F_avg = avgDielectricFresnelFit(m_ctx.ior);
    const float blendingFactor = opaqueDielectricFactor(wo, F_avg);
    if (r.z < blendingFactor) {
      if (m_ggx.isSmooth()) {
        const float fresnel_ss = fresnel(fabsf(wo.z), m_ctx.ior);
        const float3 wi = f3(-wo.x, -wo.y, wo.z);
        Sample s; s.flags = Sample_Reflected | Sample_Specular; s.wi = wi; s.f = f3(fresnel_ss / fabsf(wi.z));
        s.pdf = blendingFactor;
        return s;
      }
      const float3 wm = m_ggx.sampleVmdf(wo, {r.x, r.y});
      if (length_squared(wm) == 0.0f) return {};
      const float3 wi = reflect(-wo, wm);
      const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), m_ctx.ior);
      float dielectricBrdf = fresnel_ss * m_ggx.singleScatterBRDF(wo, wi, wm);
      if (m_flags & PT_FLAG_MULTISCATTER_GGX) dielectricBrdf += multiscatter(wo, wi, F_avg);
      Sample s; s.flags = Sample_Reflected | Sample_Glossy; s.wi = wi; s.f = f3(dielectricBrdf);
      s.pdf = m_ggx.pdf(wo, wm) * blendingFactor;
      return s;
    } else {
      float3 wi = sampleCosineHemisphere({r.x, r.y});
      if (wo.z < 0.0f) wi *= -1.0f;
      const float cDiffuse = diffuseFactor(wo, wi);
      int flags = Sample_Reflected | Sample_Diffuse;
      if (m_ctx.flags & PT_MATERIAL_EMISSIVE) flags |= Sample_Emitted;
      Sample s; s.flags = flags; s.wi = wi; s.f = m_ctx.albedo * cDiffuse;
      s.Le = m_ctx.emission / (1.0f - blendingFactor);
      s.pdf = fabsf(wi.z) / PI_F * (1.0f - blendingFactor);
      return s;
    }
  }
  Sample sampleClearcoat(float3 wo, float3 r) const {  // bsdf.metal:686-714
    if (m_ggxCoat.isSmooth()) {
      const float fresnel_ss = fresnel(wo.z, m_clearcoatIor);
      const float3 wi = f3(-wo.x, -wo.y, wo.z);
      Sample s; s.flags = Sample_Reflected | Sample_Specular; s.wi = wi; s.f = f3(fresnel_ss / fabsf(wi.z));
      s.pdf = fresnel_ss;
      return s;
    }
    const float3 wm = m_ggxCoat.sampleVmdf(wo, {r.x, r.y});
    const float3 wi = reflect(-wo, wm);
    if (wo.z * wi.z < 0.0f) return {};
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), m_clearcoatIor);
    Sample s; s.flags = Sample_Reflected | Sample_Glossy; s.wi = wi;
    s.f = f3(fresnel_ss * m_ggxCoat.singleScatterBRDF(wo, wi, wm));
    s.pdf = fresnel_ss * m_ggxCoat.pdf(wo, wm);
    return s;
  }
  Sample sample(float3 wo, float4 r, float2 rc) const {  // bsdf.metal:228-252
    float c = m_ctx.clearcoat;
    float m = m_ctx.metallic;
    float t = m_ctx.transmission;
    float pClearcoat = c;
    if (pClearcoat > 0.0f) {
      const float3 wmCoat = m_ggxCoat.isSmooth() ? f3(0, 0, 1) : m_ggxCoat.sampleVmdf(wo, rc);
      pClearcoat *= fresnel(fabsf(dot(wo, wmCoat)), m_clearcoatIor);
    }
    const float pMetallic = pClearcoat + (1.0f - pClearcoat) * m;
    const float pTransparent = pClearcoat + (1.0f - pClearcoat) * (m + (1.0f - m) * t);
    const float3 rxyz = f3(r.x, r.y, r.z);
    if (r.w < pClearcoat) return sampleClearcoat(wo, rxyz);
    if (r.w < pMetallic) return sampleMetallic(wo, rxyz);
    if (r.w < pTransparent) return sampleTransparentDielectric(wo, rxyz);
    return sampleOpaqueDielectric(wo, rxyz);
  }
};

// ------------------------------------------------------------------------------------------------------------
// Scene (flattened snapshot) + intersection
// ------------------------------------------------------------------------------------------------------------
struct WorldTri {
  float3 v0, e1, e2;
  uint32_t inst, prim;
};

struct Ray { float3 origin, direction; float min_distance, max_distance; };
struct Intersection { bool hit = false; float distance = 0; float u = 0, v = 0; uint32_t instance_id = 0, primitive_id = 0; };

// Moeller-Trumbore, fixed operation order (the intersection contract in the header comment).
inline bool intersect_triangle(const Ray& ray, const WorldTri& tri, float* t_out, float* u_out, float* v_out) {
  const float3 p = cross(ray.direction, tri.e2);
  const float det = dot(tri.e1, p);
  // r4 (intersection contract): a determinant that is rounding noise — a ray lying IN the triangle's plane, the sum e1 . p
  // cancelling to below 1e-6 of its terms' magnitude — is a miss, not a coin toss: with `det == 0` alone such rays were accepted or rejected by
  // noise, and the answer depended on which coplanar triangles a traversal happened to test (found by the extended fuzz: seeds 20341, 310601)
  if (!(fabsf(det) > (1e-6f * ((fabsf(tri.e1.x) + fabsf(tri.e1.y)) + fabsf(tri.e1.z))) * ((fabsf(p.x) + fabsf(p.y)) + fabsf(p.z)))) return false;
  const float inv = 1.0f / det;
  const float3 s = ray.origin - tri.v0;
  const float u = dot(s, p) * inv;
  if (!(u >= 0.0f && u <= 1.0f)) return false;
  const float3 q = cross(s, tri.e1);
  const float v = dot(ray.direction, q) * inv;
  if (!(v >= 0.0f && u + v <= 1.0f)) return false;
  const float t = dot(tri.e2, q) * inv;
  if (!(t >= ray.min_distance && t <= ray.max_distance)) return false;
  *t_out = t; *u_out = u; *v_out = v;
  return true;
}

struct BvhNode {
  float lo[3], hi[3];
  uint32_t left, right;   // internal: children; leaf: left = first, right = count | 0x80000000
};

struct MeshData {
  std::vector<pt_float3> positions;
  std::vector<pt_vertex_data> vdata;
  std::vector<uint32_t> indices, slots;
};

struct TraversalCounters { uint64_t nodes = 0, tris = 0; };

}  // namespace

struct orc_scene {
  std::vector<MeshData> meshes;
  std::vector<pt_instance> instances;
  std::vector<Xform> xforms;
  std::vector<std::vector<pt_material_gpu>> inst_materials;
  std::vector<uint8_t> inst_nonopaque;  // MTL::AccelerationStructureInstanceOptionNonOpaque (renderer_pt.cpp:714-729)
  std::vector<Tex> textures;
  int env_texture = -1;
  std::vector<pt_alias_entry> env_alias;
  std::vector<WorldTri> tris;       // instance-major: global id order == (instance, primitive) order
  std::vector<uint32_t> bvh_order;  // permutation of triangle ids used by leaves
  std::vector<BvhNode> bvh;
  bool use_bvh = true;
  std::vector<float> lut_storage;
  LutSet luts;
  pt_render_params params{};
  pt_constants constants{};
  Mat3 idt{};
  std::vector<pt_area_light> lights;
  std::atomic<uint64_t> n_closest{0}, n_shadow{0}, n_shaded{0}, n_paths{0}, n_nodes_closest{0}, n_tris_closest{0},
      n_nodes_shadow{0}, n_tris_shadow{0}, n_nonfinite{0};
};

namespace {

// ---- BVH (oracle-private; median split; conservative slab test on inflated boxes) ---------------------------
void tri_bounds(const WorldTri& t, float lo[3], float hi[3]) {
  float3 a = t.v0, b = t.v0 + t.e1, c = t.v0 + t.e2;
  const float* pa = &a.x; const float* pb = &b.x; const float* pc = &c.x;
  for (int k = 0; k < 3; k++) {
    lo[k] = std::min(pa[k], std::min(pb[k], pc[k]));
    hi[k] = std::max(pa[k], std::max(pb[k], pc[k]));
    // v0 + e1 re-rounds; widen by a few ulps so the original vertex is inside for sure
    float m = std::max(std::fabs(lo[k]), std::fabs(hi[k]));
    float eps = m * 8e-6f + 1e-30f;
    lo[k] -= eps; hi[k] += eps;
  }
}

uint32_t build_node(orc_scene& sc, std::vector<float>& cent, uint32_t first, uint32_t count) {
  uint32_t idx = (uint32_t)sc.bvh.size();
  sc.bvh.push_back({});
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
  for (uint32_t i = first; i < first + count; i++) {
    float l[3], h[3];
    tri_bounds(sc.tris[sc.bvh_order[i]], l, h);
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]);
      float c = cent[3 * (size_t)sc.bvh_order[i] + k];
      clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
    }
  }
  BvhNode n{};
  for (int k = 0; k < 3; k++) { n.lo[k] = lo[k]; n.hi[k] = hi[k]; }
  int axis = 0;
  float ext = chi[0] - clo[0];
  for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > ext) { ext = chi[k] - clo[k]; axis = k; }
  if (count <= 4 || ext <= 0.0f) {
    n.left = first; n.right = count | 0x80000000u;
    sc.bvh[idx] = n;
    return idx;
  }
  uint32_t mid = first + count / 2;
  std::nth_element(sc.bvh_order.begin() + first, sc.bvh_order.begin() + mid, sc.bvh_order.begin() + first + count,
                   [&](uint32_t a, uint32_t b) { return cent[3 * (size_t)a + axis] < cent[3 * (size_t)b + axis]; });
  uint32_t l = build_node(sc, cent, first, mid - first);
  uint32_t r = build_node(sc, cent, mid, first + count - mid);
  n.left = l; n.right = r;
  sc.bvh[idx] = n;
  return idx;
}

void build_bvh(orc_scene& sc) {
  size_t n = sc.tris.size();
  sc.bvh_order.resize(n);
  std::vector<float> cent(3 * n);
  for (size_t i = 0; i < n; i++) {
    sc.bvh_order[i] = (uint32_t)i;
    float lo[3], hi[3];
    tri_bounds(sc.tris[i], lo, hi);
    for (int k = 0; k < 3; k++) cent[3 * i + k] = 0.5f * (lo[k] + hi[k]);
  }
  sc.bvh.clear();
  sc.bvh.reserve(n);
  if (n) build_node(sc, cent, 0, (uint32_t)n);
}

// r5: boxes are culled against tbest * (1 + 1e-4) (r1-r4: tbest) — fp32 Moeller-Trumbore reports t with an error of ~ulp(|o - v0|) /
// cos(incidence), at grazing incidence up to ~1e-5 t BEFORE the ray enters the triangle's accurately tested box; with the old margin such a
// triangle was tested or culled depending on the order the candidates came up in, and this tree, the product's tree and brute force could
// each answer differently (found by the full-size C5 test).  Same constant as pt_bvh.h kCullSlack.
inline bool slab(const BvhNode& n, const Ray& r, const float inv[3], float tbest) {
  const float* o = &r.origin.x;
  float tn = r.min_distance, tf = tbest * 1.0001f;
  for (int k = 0; k < 3; k++) {
    float t0 = (n.lo[k] - o[k]) * inv[k];
    float t1 = (n.hi[k] - o[k]) * inv[k];
    float a = fminf(t0, t1), b = fmaxf(t0, t1);  // NaN (0*inf) operands are ignored by fmin/fmax
    tn = fmaxf(tn, a);
    tf = fminf(tf, b);
  }
  return tn <= tf * 1.0000005f + 1e-30f;
}

// intersections.metal:8-39 alphaTestIntersectionFunction: called for every candidate on a non-opaque instance
bool alpha_test(const orc_scene& sc, uint32_t inst, uint32_t prim, float u, float v, float r) {
  const MeshData& mesh = sc.meshes[sc.instances[inst].accelerationStructureIndex];
  const pt_material_gpu& material = sc.inst_materials[inst][mesh.slots[prim]];
  float alpha = material.baseColor[3];
  if (material.baseTextureId >= 0) {
    const uint32_t* idx = &mesh.indices[3 * (size_t)prim];
    float2 tc[3];
    for (int i = 0; i < 3; i++) tc[i] = {mesh.vdata[idx[i]].texCoords[0], mesh.vdata[idx[i]].texCoords[1]};
    float2 surfaceUV = interpolate(tc, {u, v});
    alpha *= tex_sample(sc.textures[material.baseTextureId], surfaceUV).w;
  }
  return alpha > r;
}

template <bool ANY>
Intersection intersect_scene(const orc_scene& sc, const Ray& ray, float payload_r, TraversalCounters* tc) {
  Intersection best;
  float tbest = ray.max_distance;
  uint32_t best_id = 0xffffffffu;
  auto test = [&](uint32_t id) -> bool {
    float t, u, v;
    if (tc) tc->tris++;
    if (!intersect_triangle(ray, sc.tris[id], &t, &u, &v)) return false;
    if (sc.inst_nonopaque[sc.tris[id].inst] && !alpha_test(sc, sc.tris[id].inst, sc.tris[id].prim, u, v, payload_r)) return false;
    if (ANY) { best.hit = true; return true; }
    if (!best.hit || t < tbest || (t == tbest && id < best_id)) {
      best.hit = true; tbest = t; best_id = id;
      best.distance = t; best.u = u; best.v = v;
      best.instance_id = sc.tris[id].inst; best.primitive_id = sc.tris[id].prim;
    }
    return false;
  };
  if (!sc.use_bvh) {
    for (uint32_t id = 0; id < (uint32_t)sc.tris.size(); id++)
      if (test(id)) return best;
    return best;
  }
  if (sc.bvh.empty()) return best;
  float inv[3] = {1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const BvhNode& n = sc.bvh[stack[--sp]];
    if (tc) tc->nodes++;
    if (!slab(n, ray, inv, tbest)) continue;
    if (n.right & 0x80000000u) {
      uint32_t cnt = n.right & 0x7fffffffu;
      for (uint32_t i = 0; i < cnt; i++)
        if (test(sc.bvh_order[n.left + i])) return best;
    } else {
      assert(sp + 2 <= 128);
      stack[sp++] = n.left;
      stack[sp++] = n.right;
    }
  }
  return best;
}

// ---- host-side table builders --------------------------------------------------------------------------------

// core/colorspace.cpp:13-33 (Colorspace ctor) and colorspace.hpp:61-63 (transform = dst.fromXYZ * src.toXYZ).
// Evaluated in double and rounded once: Apple's simd inverse() is closed, so last-bit parity is unpinned anyway.
struct M3d { double m[3][3]; };  // m[col][row]
M3d m3_mul(const M3d& a, const M3d& b) {
  M3d r{};
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a.m[k][rr] * b.m[c][k];
      r.m[c][rr] = s;
    }
  return r;
}
M3d m3_inv(const M3d& a) {
  const double (*m)[3] = a.m;
  double c00 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  double c01 = m[2][1] * m[0][2] - m[0][1] * m[2][2];
  double c02 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
  double det = m[0][0] * c00 + m[1][0] * c01 + m[2][0] * c02;
  double id = 1.0 / det;
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
M3d colorspace_toXYZ(const float r[2], const float g[2], const float b[2], const float w[2]) {
  double prim[3][3] = {{r[0], r[1], 1.0 - (double)r[0] - (double)r[1]},
                       {g[0], g[1], 1.0 - (double)g[0] - (double)g[1]},
                       {b[0], b[1], 1.0 - (double)b[0] - (double)b[1]}};
  double wx = w[0], wy = w[1], wz = 1.0 - wx - wy;
  double W[3] = {wx / wy, 1.0, wz / wy};
  M3d mx{};
  for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) mx.m[c][rr] = prim[c][rr];
  M3d inv = m3_inv(mx);
  double scale[3];
  for (int rr = 0; rr < 3; rr++) scale[rr] = inv.m[0][rr] * W[0] + inv.m[1][rr] * W[1] + inv.m[2][rr] * W[2];
  M3d out{};
  for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) out.m[c][rr] = mx.m[c][rr] * scale[c];
  return out;
}
const float BT709_r[2] = {0.640f, 0.330f}, BT709_g[2] = {0.300f, 0.600f}, BT709_b[2] = {0.150f, 0.060f},
            D65[2] = {0.3127f, 0.3290f};  // core/colorspace.cpp:5, colorspace.hpp:47

Mat3 compute_transform(const pt_colorspace& a, const pt_colorspace& b) {  // colorspace.hpp:61-63: dst.fromXYZ * src.toXYZ
  M3d src = colorspace_toXYZ(a.r, a.g, a.b, a.w);
  M3d dst = colorspace_toXYZ(b.r, b.g, b.b, b.w);
  M3d t = m3_mul(m3_inv(dst), src);
  Mat3 o;
  o.c0 = f3((float)t.m[0][0], (float)t.m[0][1], (float)t.m[0][2]);
  o.c1 = f3((float)t.m[1][0], (float)t.m[1][1], (float)t.m[1][2]);
  o.c2 = f3((float)t.m[2][0], (float)t.m[2][1], (float)t.m[2][2]);
  return o;
}

Mat3 compute_idt(const pt_colorspace& ws) {
  M3d src = colorspace_toXYZ(BT709_r, BT709_g, BT709_b, D65);
  M3d dst = colorspace_toXYZ(ws.r, ws.g, ws.b, ws.w);
  M3d t = m3_mul(m3_inv(dst), src);
  Mat3 o;
  o.c0 = f3((float)t.m[0][0], (float)t.m[0][1], (float)t.m[0][2]);
  o.c1 = f3((float)t.m[1][0], (float)t.m[1][1], (float)t.m[1][2]);
  o.c2 = f3((float)t.m[2][0], (float)t.m[2][1], (float)t.m[2][2]);
  return o;
}

inline pt_float3 to_pt(float3 v) { return {v.x, v.y, v.z, 0.0f}; }
inline float3 from_pt(const pt_float3& v) { return f3(v.x, v.y, v.z); }

// renderer_pt.cpp:965-1021 updateConstants (camera part), core/camera.hpp:47-50 croppedSensorHeight
void update_constants(orc_scene& sc, const pt_camera& cam) {
  const pt_render_params& p = sc.params;
  float3 col[4];
  for (int i = 0; i < 4; i++) col[i] = f3(cam.world[i][0], cam.world[i][1], cam.world[i][2]);
  float3 u = col[0] / length(col[0]);
  float3 v = col[1] / length(col[1]);
  float3 w = col[2] / length(col[2]);
  float3 pos = col[3];
  float sizex = (float)p.width, sizey = (float)p.height;
  float aspect = sizex / sizey;  // renderer_pt.cpp:203 m_aspect = size.x / size.y
  float sensorAspect = cam.sensor_size[0] / cam.sensor_size[1];
  float cropped = cam.sensor_size[0] / fmaxf(sensorAspect, aspect);
  float vh = cam.focus_distance * cropped / cam.focal_length;
  float vw = vh * aspect;
  float3 vu = u * vw;
  float3 vv = -v * vh;
  pt_constants& c = sc.constants;
  memset(&c, 0, sizeof(c));
  c.frameIdx = 0;
  c.spp = p.spp;
  c.gmonBuckets = (p.flags & PT_FLAG_GMON) ? p.gmon_buckets : 1;  // renderer_pt.cpp:993-994
  c.lutSizeE = (uint32_t)sc.luts.E.w;
  c.lutSizeEavg = (uint32_t)sc.luts.Eavg.w;
  c.flags = p.flags;
  c.size[0] = p.width; c.size[1] = p.height;
  c.idt[0] = to_pt(sc.idt.c0); c.idt[1] = to_pt(sc.idt.c1); c.idt[2] = to_pt(sc.idt.c2);
  c.camera.position = to_pt(pos);
  c.camera.topLeft = to_pt((pos - cam.focus_distance * w) - (vu + vv) * 0.5f);
  c.camera.pixelDeltaU = to_pt(vu / sizex);
  c.camera.pixelDeltaV = to_pt(vv / sizey);
  c.camera.apertureRadius = cam.aperture > 0.0f ? (cam.focal_length / 2000.0f) / cam.aperture : 0.0f;
  c.camera.apertureBlades = cam.aperture_blades;
  c.camera.apertureRoundness = cam.roundness;
  c.camera.bokehPower = cam.bokeh_power;
}

// core/material.hpp:44-47 isEmissive (textures are a "next" row)
inline bool is_emissive(const pt_material_gpu& m) {
  float3 e = from_pt(m.emission) * m.emissionStrength;
  return length_squared(e) > 0.0f || m.emissionTextureId >= 0;  // textures.contains(TextureSlot::Emission)
}

// Decode one snapshot texture to linear float4 (see the filtering contract above).
Tex decode_texture(const pt_texture& t) {
  Tex o;
  o.w = (int)t.width; o.h = (int)t.height;
  const size_t n = (size_t)t.width * t.height;
  o.px.resize(4 * n);
  static float srgb_lut[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) {
      double c = i / 255.0;
      srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    init = true;
  }
  const uint8_t* b = (const uint8_t*)t.pixels;
  const float* f = (const float*)t.pixels;
  for (size_t i = 0; i < n; i++) {
    float* d = &o.px[4 * i];
    switch (t.format) {
      case PT_TEX_RGBA8_SRGB: d[0] = srgb_lut[b[4 * i]]; d[1] = srgb_lut[b[4 * i + 1]]; d[2] = srgb_lut[b[4 * i + 2]]; d[3] = (float)b[4 * i + 3] / 255.0f; break;
      case PT_TEX_RGBA8: for (int c = 0; c < 4; c++) d[c] = (float)b[4 * i + c] / 255.0f; break;
      case PT_TEX_RG8: d[0] = (float)b[2 * i] / 255.0f; d[1] = (float)b[2 * i + 1] / 255.0f; d[2] = 0.0f; d[3] = 1.0f; break;
      case PT_TEX_R8: d[0] = (float)b[i] / 255.0f; d[1] = 0.0f; d[2] = 0.0f; d[3] = 1.0f; break;
      default: for (int c = 0; c < 4; c++) d[c] = f[4 * i + c]; break;  // PT_TEX_RGBA32F
    }
  }
  return o;
}

// core/environment.cpp:5-91 Environment::rebuildAliasTable (Vose's method on luma-proportional importance)
std::vector<pt_alias_entry> rebuild_alias_table(const Tex& texture) {
  const uint64_t n = (uint64_t)texture.w * texture.h;
  std::vector<pt_alias_entry> table(n, pt_alias_entry{0.0f, 0.0f, 0u});
  float totalImportance = 0.0f;
  std::vector<float> importance;
  importance.reserve(n);
  const float3 lumaCoeffs = f3(0.2126f, 0.7152f, 0.0722f);
  for (uint64_t i = 0; i < n; i++) {
    float luma = dot(f3(texture.px[4 * i], texture.px[4 * i + 1], texture.px[4 * i + 2]), lumaCoeffs);
    importance.push_back(luma);
    totalImportance += luma;
  }
  float scale = (float)n / totalImportance;
  for (uint64_t i = 0; i < n; i++) {
    importance[i] *= scale;
    table[i].pdf = importance[i];
  }
  std::vector<size_t> small, large;
  for (uint64_t i = 0; i < n; i++) {
    if (importance[i] < 1.0f) small.push_back(i);
    else large.push_back(i);
  }
  while (!small.empty() && !large.empty()) {
    size_t l = small.back(); small.pop_back();
    size_t g = large.back(); large.pop_back();
    table[l].p = importance[l];
    table[l].aliasIdx = (uint32_t)g;
    importance[g] = (importance[g] + importance[l]) - 1.0f;
    if (importance[g] < 1.0f) small.push_back(g);
    else large.push_back(g);
  }
  while (!large.empty()) { size_t g = large.back(); large.pop_back(); table[g].p = 1.0f; }
  while (!small.empty()) { size_t l = small.back(); small.pop_back(); table[l].p = 1.0f; }
  return table;
}

// renderer_pt.cpp:838-917 rebuildLightData (area lights)
void rebuild_light_data(orc_scene& sc) {
  sc.lights.clear();
  float total = 0.0f;
  // color::transform(BT709, working) * emission * strength (renderer_pt.cpp:895-897)
  for (uint32_t ii = 0; ii < sc.instances.size(); ii++) {
    const MeshData& mesh = sc.meshes[sc.instances[ii].accelerationStructureIndex];
    const auto& mats = sc.inst_materials[ii];
    bool any = false;
    for (const auto& m : mats) any = any || is_emissive(m);
    if (!any) continue;
    const Xform& X = sc.xforms[ii];
    size_t ntri = mesh.indices.size() / 3;
    for (size_t i = 0; i < ntri; i++) {
      const pt_material_gpu& m = mats[mesh.slots[i]];
      if (!is_emissive(m)) continue;
      uint32_t i0 = mesh.indices[3 * i], i1 = mesh.indices[3 * i + 1], i2 = mesh.indices[3 * i + 2];
      float3 v0 = transformPoint(from_pt(mesh.positions[i0]), X);
      float3 v1 = transformPoint(from_pt(mesh.positions[i1]), X);
      float3 v2 = transformPoint(from_pt(mesh.positions[i2]), X);
      float3 edge1 = v1 - v0, edge2 = v2 - v0;
      float area = length(cross(edge1, edge2)) * 0.5f;
      float3 emission = mul(sc.idt, from_pt(m.emission)) * m.emissionStrength;
      float lightPower = emission.y * area * PI_F;  // dot(emission, (0,1,0)) * area * pi
      total += lightPower;
      pt_area_light L{};
      L.instanceIdx = ii; L.indices[0] = i0; L.indices[1] = i1; L.indices[2] = i2;
      L.area = area; L.power = lightPower; L.cumulativePower = total; L.emission = to_pt(emission);
      sc.lights.push_back(L);
    }
  }
  sc.constants.lightCount = (uint32_t)sc.lights.size();
  sc.constants.envLightCount = sc.env_texture >= 0 ? 1 : 0;
  sc.constants.totalLightPower = total;
}

// ------------------------------------------------------------------------------------------------------------
// kernel.metal device functions
// ------------------------------------------------------------------------------------------------------------

// kernel.metal:195-238 spawnRayFromCamera
Ray spawnRayFromCamera(const pt_camera_data& camera, uint32_t px, uint32_t py, float2 pixelSample, float2 lensSample) {
  Ray ray;
  ray.origin = from_pt(camera.position);
  if (camera.apertureRadius > 0.0f) {
    float2 lensPos = sampleDiskPolar(lensSample);
    lensPos.x = powr_det(lensPos.x, exp2_det(camera.bokehPower));
    if (camera.apertureRoundness < 1.0f) {
      float n = (float)camera.apertureBlades;
      float rPolygon = cos_det(PI_F / n) / cos_det(fmodf(lensPos.y + 1.5f * PI_F, 2.0f * PI_F / n) - PI_F / n);
      float r = mix(rPolygon, 1.0f, camera.apertureRoundness);
      lensPos.x *= r;
    }
    float c, s;
    sincos_det(lensPos.y, &s, &c);
    float2 lp = {lensPos.x * c * camera.apertureRadius, lensPos.x * s * camera.apertureRadius};
    ray.origin += lp.x * normalize(from_pt(camera.pixelDeltaU)) + lp.y * normalize(from_pt(camera.pixelDeltaV));
  }
  float2 filmPos = {(float)px + pixelSample.x, (float)py + pixelSample.y};
  ray.direction = normalize(((from_pt(camera.topLeft) + filmPos.x * from_pt(camera.pixelDeltaU)) +
                             filmPos.y * from_pt(camera.pixelDeltaV)) - ray.origin);
  ray.max_distance = std::numeric_limits<float>::infinity();
  ray.min_distance = 1e-3f;
  return ray;
}

struct Hit {  // kernel.metal:74-83
  float3 pos, normal, geometricNormal; float2 uv; float3 wo; Frame frame; const pt_material_gpu* material;
};

// kernel.metal:118-188 Resources::getIntersectionData
Hit getIntersectionData(const orc_scene& sc, const Ray& ray, const Intersection& isect) {
  uint32_t instanceIdx = isect.instance_id;
  uint32_t geometryIdx = sc.instances[instanceIdx].accelerationStructureIndex;
  const MeshData& mesh = sc.meshes[geometryIdx];
  const uint32_t* idx = &mesh.indices[3 * (size_t)isect.primitive_id];
  uint32_t materialSlot = mesh.slots[isect.primitive_id];
  const pt_material_gpu& material = sc.inst_materials[instanceIdx][materialSlot];

  float3 vertexPositions[3], vertexNormals[3], vertexTangents[3];
  float2 vertexTexCoords[3];
  for (int i = 0; i < 3; i++) {
    vertexPositions[i] = from_pt(mesh.positions[idx[i]]);
    vertexNormals[i] = from_pt(mesh.vdata[idx[i]].normal);
    vertexTangents[i] = f3(mesh.vdata[idx[i]].tangent[0], mesh.vdata[idx[i]].tangent[1], mesh.vdata[idx[i]].tangent[2]);
    vertexTexCoords[i] = {mesh.vdata[idx[i]].texCoords[0], mesh.vdata[idx[i]].texCoords[1]};
  }
  float tangentSign = mesh.vdata[idx[0]].tangent[3];

  float2 bary = {isect.u, isect.v};
  float3 surfaceNormal = interpolate(vertexNormals, bary);
  float3 surfaceTangent = interpolate(vertexTangents, bary);
  float2 surfaceUV = interpolate(vertexTexCoords, bary);
  float3 geometricNormal = normalize(cross(vertexPositions[1] - vertexPositions[0], vertexPositions[2] - vertexPositions[0]));

  const Xform& objectToWorld = sc.xforms[instanceIdx];
  float3 wsHitPoint = ray.origin + ray.direction * isect.distance;
  float3 wsSurfaceNormal = normalize(transformVec(surfaceNormal, objectToWorld));
  float3 wsSurfaceTangent = normalize(transformVec(surfaceTangent, objectToWorld));
  float3 wsGeometricNormal = normalize(transformVec(geometricNormal, objectToWorld));

  Frame frame = Frame::fromNT(wsSurfaceNormal, wsSurfaceTangent, tangentSign);
  if (material.normalTextureId >= 0) {  // kernel.metal:166-175
    float4 t = tex_sample(sc.textures[material.normalTextureId], surfaceUV);
    float3 sampledNormal = f3(t.x, t.y, t.z) * 2.0f - f3(1.0f);
    wsSurfaceNormal = frame.localToWorld(sampledNormal);
    frame = Frame::fromNormal(wsSurfaceNormal);
  }
  float3 wo = frame.worldToLocal(-ray.direction);
  return {wsHitPoint, wsSurfaceNormal, wsGeometricNormal, surfaceUV, wo, frame, &material};
}

// kernel.metal:379-394 sampleLightPower
const pt_area_light& sampleLightPower(const orc_scene& sc, float r) {
  const pt_constants& constants = sc.constants;
  r *= constants.totalLightPower;
  uint32_t sz = constants.lightCount - 1, idx = 0u;
  while (sz > 0) {
    uint32_t h = sz >> 1, middle = idx + h;
    bool res = sc.lights[middle].cumulativePower < r;
    idx = res ? (middle + 1) : idx;
    sz = res ? sz - (h + 1) : h;
  }
  idx = std::min(std::max(idx, 0u), constants.lightCount - 1);
  return sc.lights[idx];
}

struct LightSample { float3 Li, pos, normal, wi; float pdf; };  // kernel.metal:396-402

// kernel.metal:407-435 sampleAreaLight
LightSample sampleAreaLight(const orc_scene& sc, const Hit& hit, const pt_area_light& light, float2 r) {
  const MeshData& mesh = sc.meshes[sc.instances[light.instanceIdx].accelerationStructureIndex];
  float3 vertexPositions[3];
  for (int i = 0; i < 3; i++) vertexPositions[i] = from_pt(mesh.positions[light.indices[i]]);
  const float2 sampledCoords = sampleTriUniform(r);
  const Xform& transform = sc.xforms[light.instanceIdx];
  const float3 osNormal = cross(vertexPositions[1] - vertexPositions[0], vertexPositions[2] - vertexPositions[0]);
  const float3 pos = transformPoint(interpolate(vertexPositions, sampledCoords), transform);
  const float3 normal = normalize(transformVec(osNormal, transform));
  const float3 wi = normalize(pos - hit.pos);
  LightSample ls;
  ls.Li = from_pt(light.emission);
  ls.pos = pos; ls.normal = normal; ls.wi = wi;
  ls.pdf = length_squared(pos - hit.pos) / (fabsf(dot(normal, wi)) * light.area);
  return ls;
}

// kernel.metal:20-25 rayDirToUv, :27-34 uvToRayDir
inline float2 rayDirToUv(float3 dir) {
  float phi = atan2_det(-dir.z, -dir.x);
  float theta = acos_det(dir.y);
  return {phi / (2.0f * PI_F), theta / PI_F};
}
inline float3 uvToRayDir(float2 uv) {
  float y, r, cosPhi, sinPhi;
  sincos_det(uv.y * PI_F, &r, &y);
  sincos_det(uv.x * 2.0f * PI_F, &sinPhi, &cosPhi);
  return normalize(f3(-cosPhi * r, y, -sinPhi * r));
}

// kernel.metal:440-467 sampleEnvironmentLight
LightSample sampleEnvironmentLight(const orc_scene& sc, float2 r) {
  const Tex& texture = sc.textures[sc.env_texture];
  uint64_t w = (uint64_t)texture.w, h = (uint64_t)texture.h;
  uint64_t n = w * h;
  uint64_t i = std::min<uint64_t>(n - 1, (uint64_t)(r.x * (float)n));
  if (r.y >= sc.env_alias[i].p) i = sc.env_alias[i].aliasIdx;
  uint64_t x = i % w, y = i / w;
  float2 uv = {(float)x / (float)w, (float)y / (float)h};
  float4 Le = tex_sample(texture, uv);
  float3 wi = uvToRayDir(uv);
  LightSample ls;
  ls.Li = f3(Le.x, Le.y, Le.z);
  ls.pos = wi * 100.0f;
  ls.normal = -wi;
  ls.wi = wi;
  ls.pdf = sc.env_alias[i].pdf / (4.0f * PI_F);
  return ls;
}

struct PathLog { int32_t* hits; uint32_t stride; uint32_t pixel; };  // hits[(bounce*stride + pixel)*2 + {0,1}]

struct ThreadStats { uint64_t closest = 0, shadow = 0, shaded = 0, nonfinite = 0; TraversalCounters tc_closest, tc_shadow; bool verbose = false; };

// kernel.metal:473-686 misKernel (integrator == MIS) and :256-372 pathtracingKernel (SIMPLE), one pixel, one sample.
// `max_bounces` replaces the compile-time MAX_BOUNCES 50 (kernel.metal:5).
float3 trace_path(const orc_scene& sc, uint32_t px, uint32_t py, uint32_t frameIdx, PathLog* log, ThreadStats& st,
                  bool count_traversal) {
  const pt_constants& C = sc.constants;
  const bool mis = sc.params.integrator == PT_INTEGRATOR_MIS;
  HaltonSampler halton(px, py, frameIdx);
  float2 s_pixel = halton.sample2d();
  float2 s_lens = halton.sample2d();
  Ray ray = spawnRayFromCamera(C.camera, px, py, s_pixel, s_lens);

  float3 attenuation = f3(1.0f);
  float3 L = f3(0.0f);
  float3 lastHitPos = f3(0.0f);
  Sample lastSample;
  const float3 backgroundColor = f3(0.0f);  // defs.metal:21
  TraversalCounters* tcc = count_traversal ? &st.tc_closest : nullptr;
  TraversalCounters* tcs = count_traversal ? &st.tc_shadow : nullptr;

  for (uint32_t bounce = 0; bounce < sc.params.max_bounces; bounce++) {
    float ir = halton.sample1d();  // alpha-test payload (kernel.metal:510)
    st.closest++;
    Intersection isect = intersect_scene<false>(sc, ray, ir, tcc);
    if (log) {
      int32_t* h = &log->hits[((size_t)bounce * log->stride + log->pixel) * 2];
      h[0] = isect.hit ? (int32_t)isect.instance_id : -1;
      h[1] = isect.hit ? (int32_t)isect.primitive_id : -1;
    }
    if (!isect.hit) {  // kernel.metal:517-543 (simple integrator :299-311: no MIS weight)
      if (sc.env_texture >= 0) {
        const Tex& texture = sc.textures[sc.env_texture];
        float2 uv = rayDirToUv(ray.direction);
        float4 t = tex_sample(texture, uv);
        const float3 Le = f3(t.x, t.y, t.z);
        if (!mis || bounce == 0 || (lastSample.flags & Sample_Specular)) {
          L += attenuation * Le;
        } else {
          // uint32_t x = w * uv.x (kernel.metal:530-531): uv.x is negative for half the sphere (atan2 range); the
          // float->uint conversion of a negative value is defined here as 0, and indices are clamped into the table
          uint32_t w = (uint32_t)texture.w, h = (uint32_t)texture.h;
          float fxw = (float)w * uv.x, fyh = (float)h * uv.y;
          uint32_t x = fxw > 0.0f ? (uint32_t)fxw : 0u, y = fyh > 0.0f ? (uint32_t)fyh : 0u;
          x = std::min(x, w - 1); y = std::min(y, h - 1);
          float lightPdf = sc.env_alias[(size_t)y * w + x].pdf * 0.25f * 0.318309886183790671538f;  // M_1_PI_F
          float bsdfWeight = lastSample.pdf / (lastSample.pdf + lightPdf);
          L += attenuation * bsdfWeight * Le;
        }
      }
      L += attenuation * backgroundColor;
      break;
    }
    st.shaded++;
    if (st.verbose)
      fprintf(stderr, "  bounce %u hit inst %u prim %u t %.9g u %.9g v %.9g o (%.9g %.9g %.9g) d (%.9g %.9g %.9g)\n", bounce, isect.instance_id,
              isect.primitive_id, isect.distance, isect.u, isect.v, ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z);
    const Hit hit = getIntersectionData(sc, ray, isect);

    float2 r01 = halton.sample2d();
    float r2 = halton.sample1d();
    float r3 = halton.sample1d();
    float4 r = {r01.x, r01.y, r2, r3};
    float2 rc = halton.sample2d();

    ShadingContext ctx(*hit.material, hit.uv, sc.idt, sc.textures);
    BSDF bsdf(ctx, C.flags, sc.luts);
    Sample sample = bsdf.sample(hit.wo, r, rc);

    if (sample.flags & Sample_Emitted) {  // kernel.metal:560-576 / :325-327
      if (!mis || bounce == 0 || (lastSample.flags & Sample_Specular)) {
        L += attenuation * sample.Le;
      } else {
        const float lightPdf = (sample.Le.y * PI_F / C.totalLightPower) * length_squared(lastHitPos - hit.pos) /
                               fabsf(dot(ray.direction, hit.geometricNormal));
        const float bsdfWeight = lastSample.pdf / (lastSample.pdf + lightPdf);
        L += attenuation * bsdfWeight * sample.Le;
      }
    }

    ray.origin = hit.pos;  // kernel.metal:582

    if (mis && (ctx.roughness > 0.0f || ctx.metallic + ctx.transmission < 1.0f)) {  // kernel.metal:587-639
      float2 rl = halton.sample2d();
      float rz = halton.sample1d();
      // kernel.metal:590-616. With neither area lights nor an environment the reference indexes envLights[0] out of
      // bounds (UB); we skip NEE then but keep the dimension schedule.
      const uint32_t envCount = C.envLightCount;
      if (C.lightCount > 0 || envCount > 0) {
        const float pInfinite = C.lightCount == 0 ? 1.0f : (float)envCount / (float)(envCount + 1);
        LightSample lightSample;
        float pLight = 0;
        if (rz < pInfinite) {
          rz /= pInfinite;  // (selects among envCount lights; there is one)
          pLight = pInfinite / (float)envCount;
          lightSample = sampleEnvironmentLight(sc, rl);
        } else {
          rz = (rz - pInfinite) / (1.0f - pInfinite);
          const pt_area_light& light = sampleLightPower(sc, rz);
          pLight = (1.0f - pInfinite) * light.power / C.totalLightPower;
          lightSample = sampleAreaLight(sc, hit, light, rl);
        }

        const float3 wi = hit.frame.worldToLocal(lightSample.wi);
        const Eval bsdfEval = bsdf.eval(hit.wo, wi);
        if (length_squared(bsdfEval.f) > 0.0f) {
          Ray shadow;
          shadow.origin = ray.origin;
          shadow.direction = lightSample.wi;
          shadow.min_distance = 1e-3f;
          shadow.max_distance = length(lightSample.pos - hit.pos) - 1e-3f;
          float ir2 = halton.sample1d();
          st.shadow++;
          bool occluded = intersect_scene<true>(sc, shadow, ir2, tcs).hit;
          if (st.verbose) fprintf(stderr, "  bounce %u shadow ray occluded %d tmax %.9g d (%.9g %.9g %.9g)\n", bounce, (int)occluded, shadow.max_distance, shadow.direction.x, shadow.direction.y, shadow.direction.z);
          if (!occluded) {
            float pdfLight = pLight * lightSample.pdf;
            float3 Ld = lightSample.Li * bsdfEval.f * fabsf(wi.z) / (pdfLight + bsdfEval.pdf);
            L += attenuation * Ld;
          }
        }
      }
    }

    if (!(sample.flags & (Sample_Reflected | Sample_Transmitted))) break;  // kernel.metal:644-645

    if (st.verbose)
      fprintf(stderr, "bounce %u inst %u prim %u t %g wo (%g %g %g) flags %d wi (%g %g %g) f (%g %g %g) pdf %g att (%g %g %g) L (%g %g %g) rough %g metal %g trans %g\n",
              bounce, isect.instance_id, isect.primitive_id, isect.distance, hit.wo.x, hit.wo.y, hit.wo.z, sample.flags, sample.wi.x,
              sample.wi.y, sample.wi.z, sample.f.x, sample.f.y, sample.f.z, sample.pdf, attenuation.x, attenuation.y, attenuation.z,
              L.x, L.y, L.z, ctx.roughness, ctx.metallic, ctx.transmission);
    attenuation *= sample.f * fabsf(sample.wi.z) / sample.pdf;  // :650

    if (bounce > 0) {  // :655-661
      float q = fmaxf(0.0f, 1.0f - fmaxf(attenuation.x, fmaxf(attenuation.y, attenuation.z)));
      if (halton.sample1d() < q) break;
      attenuation /= 1.0f - q;
    }

    ray.max_distance = std::numeric_limits<float>::infinity();
    ray.direction = normalize(hit.frame.localToWorld(sample.wi));  // :667
    lastHitPos = hit.pos;
    lastSample = sample;
  }
  return L;
}

#include "post_oracle.inc"

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------------------
extern "C" {

orc_scene* orc_scene_create(const pt_scene_snapshot* snap, const pt_render_params* params, const void* lut_blob,
                            uint64_t lut_size, int use_bvh) {
  if (!snap || !params || !lut_blob) return nullptr;
  auto* sc = new orc_scene();
  sc->params = *params;
  sc->use_bvh = use_bvh != 0;
  // LUT blob (tools/make_lut_blob.py)
  const uint8_t* b = (const uint8_t*)lut_blob;
  if (lut_size < 12 || memcmp(b, "PTLUT01\0", 8) != 0) { delete sc; return nullptr; }
  uint32_t count; memcpy(&count, b + 8, 4);
  if (count != 8) { delete sc; return nullptr; }
  const uint32_t* hdr = (const uint32_t*)(b + 12);
  size_t data_off = 12 + 16 * (size_t)count;
  size_t nfloats = (lut_size - data_off) / 4;
  sc->lut_storage.resize(nfloats);
  memcpy(sc->lut_storage.data(), b + data_off, nfloats * 4);
  Lut* ls[8] = {&sc->luts.E, &sc->luts.Eavg, &sc->luts.EMs, &sc->luts.EavgMs, &sc->luts.ETransIn, &sc->luts.ETransOut,
                &sc->luts.EavgTransIn, &sc->luts.EavgTransOut};
  for (int i = 0; i < 8; i++) {
    ls[i]->w = (int)hdr[4 * i]; ls[i]->h = (int)hdr[4 * i + 1]; ls[i]->depth = (int)hdr[4 * i + 2];
    ls[i]->d = sc->lut_storage.data() + hdr[4 * i + 3];
  }
  // meshes / instances
  sc->meshes.resize(snap->mesh_count);
  for (uint32_t m = 0; m < snap->mesh_count; m++) {
    const pt_mesh& pm = snap->meshes[m];
    MeshData& md = sc->meshes[m];
    md.positions.assign(pm.positions, pm.positions + pm.vertex_count);
    md.vdata.assign(pm.vertex_data, pm.vertex_data + pm.vertex_count);
    md.indices.assign(pm.indices, pm.indices + 3 * (size_t)pm.triangle_count);
    md.slots.assign(pm.material_slots, pm.material_slots + pm.triangle_count);
  }
  for (uint32_t t = 0; t < snap->texture_count; t++) sc->textures.push_back(decode_texture(snap->textures[t]));
  sc->env_texture = (snap->env_texture >= 0 && (uint32_t)snap->env_texture < snap->texture_count) ? snap->env_texture : -1;
  if (sc->env_texture >= 0) {
    const Tex& et = sc->textures[sc->env_texture];
    if (snap->env_alias) sc->env_alias.assign(snap->env_alias, snap->env_alias + (size_t)et.w * et.h);
    else sc->env_alias = rebuild_alias_table(et);
  }
  sc->instances.assign(snap->instances, snap->instances + snap->instance_count);
  sc->xforms.resize(snap->instance_count);
  sc->inst_nonopaque.assign(snap->instance_count, 0);
  sc->inst_materials.resize(snap->instance_count);
  for (uint32_t i = 0; i < snap->instance_count; i++) {
    const pt_instance& in = snap->instances[i];
    sc->xforms[i] = {f3(in.transform[0][0], in.transform[0][1], in.transform[0][2]),
                     f3(in.transform[1][0], in.transform[1][1], in.transform[1][2]),
                     f3(in.transform[2][0], in.transform[2][1], in.transform[2][2]),
                     f3(in.transform[3][0], in.transform[3][1], in.transform[3][2])};
    const pt_instance_materials& im = snap->instance_materials[i];
    sc->inst_materials[i].assign(im.materials, im.materials + im.material_count);
    // renderer_pt.cpp:626-633: the Renderer derives the Emissive / Anisotropic flags when it fills MaterialGPU
    for (auto& m : sc->inst_materials[i]) {
      if (is_emissive(m)) m.flags |= PT_MATERIAL_EMISSIVE;
      if (m.anisotropy != 0.0f) m.flags |= PT_MATERIAL_ANISOTROPIC;
      if (m.flags & PT_MATERIAL_USE_ALPHA) sc->inst_nonopaque[i] = 1;  // renderer_pt.cpp:714-729
    }
    const MeshData& md = sc->meshes[in.accelerationStructureIndex];
    size_t ntri = md.indices.size() / 3;
    for (size_t t = 0; t < ntri; t++) {
      float3 v0 = transformPoint(from_pt(md.positions[md.indices[3 * t]]), sc->xforms[i]);
      float3 v1 = transformPoint(from_pt(md.positions[md.indices[3 * t + 1]]), sc->xforms[i]);
      float3 v2 = transformPoint(from_pt(md.positions[md.indices[3 * t + 2]]), sc->xforms[i]);
      sc->tris.push_back({v0, v1 - v0, v2 - v0, i, (uint32_t)t});
    }
  }
  sc->idt = compute_idt(params->working_space);
  update_constants(*sc, snap->camera);
  rebuild_light_data(*sc);
  if (sc->use_bvh) build_bvh(*sc);
  return sc;
}

void orc_scene_destroy(orc_scene* sc) { delete sc; }

int orc_get_constants(const orc_scene* sc, pt_constants* out) { *out = sc->constants; return 0; }

float orc_atan2(float y, float x) { return atan2_det(y, x); }
float orc_acos(float x) { return acos_det(x); }
void orc_tex_sample(const orc_scene* sc, int texture, float u, float v, float out[4]) {
  float4 t = tex_sample(sc->textures[texture], float2{u, v});
  out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
}
void orc_ray_dir_to_uv(const float dir[3], float out[2]) { float2 uv = rayDirToUv(f3(dir[0], dir[1], dir[2])); out[0] = uv.x; out[1] = uv.y; }
void orc_uv_to_ray_dir(const float uv[2], float out[3]) { float3 d = uvToRayDir(float2{uv[0], uv[1]}); out[0] = d.x; out[1] = d.y; out[2] = d.z; }

int orc_get_env_alias(const orc_scene* sc, pt_alias_entry* out, uint64_t capacity, uint64_t* count) {
  *count = sc->env_alias.size();
  for (uint64_t i = 0; i < std::min<uint64_t>(capacity, *count); i++) out[i] = sc->env_alias[i];
  return 0;
}

int orc_get_lights(const orc_scene* sc, pt_area_light* out, uint32_t capacity, uint32_t* count) {
  *count = (uint32_t)sc->lights.size();
  for (uint32_t i = 0; i < std::min<uint32_t>(capacity, *count); i++) out[i] = sc->lights[i];
  return 0;
}

// Render samples [first_sample, first_sample + nsamples) of every pixel into the running-mean accumulator
// (kernel.metal:672-684): n = number of samples already in `acc` for the first one.
int orc_render(orc_scene* sc, uint32_t first_sample, uint32_t nsamples, float* acc, uint32_t acc_n0, int threads,
               int count_traversal) {
  const uint32_t W = sc->params.width, H = sc->params.height;
  const bool gmon = (sc->params.flags & PT_FLAG_GMON) != 0;
  const uint32_t gbuckets = gmon ? sc->params.gmon_buckets : 1;
  const uint32_t spb = (sc->params.spp + gbuckets - 1) / gbuckets;
  if (threads < 1) threads = 1;
  std::atomic<uint32_t> next_row{0};
  auto worker = [&]() {
    ThreadStats st;
    for (;;) {
      uint32_t y = next_row.fetch_add(1);
      if (y >= H) break;
      for (uint32_t x = 0; x < W; x++) {
        float* px = &acc[4 * ((size_t)y * W + x)];
        for (uint32_t s = 0; s < nsamples; s++) {
          float3 L = trace_path(*sc, x, y, first_sample + s, nullptr, st, count_traversal != 0);
          if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {
            st.nonfinite++;
            if (sc->params.nonfinite_policy == PT_NONFINITE_ZERO) L = f3(0.0f);  // build extension, see ptamd.h
          }
          // kernel.metal:675-681: localFrameIdx = frameIdx / gmonBuckets (gmonBuckets is 1 without GMoN, renderer_pt.cpp:993).
          // With RendererFlags_GMoN the sample goes to bucket frameIdx / ceil(spp / buckets) (renderer_pt.cpp:124-139); `acc`
          // then holds `gmon_buckets` images back to back.
          const uint32_t f = acc_n0 + s;
          uint32_t localFrameIdx = f;
          if (gmon) {
            px = &acc[4 * ((size_t)(f / spb) * W * H + (size_t)y * W + x)];
            localFrameIdx = f / gbuckets;
          }
          if (localFrameIdx > 0) {
            float3 L_prev = f3(px[0], px[1], px[2]);
            L += L_prev * (float)localFrameIdx;
            L /= (float)(localFrameIdx + 1);
          }
          px[0] = L.x; px[1] = L.y; px[2] = L.z; px[3] = 1.0f;
        }
      }
    }
    sc->n_closest += st.closest; sc->n_shadow += st.shadow; sc->n_shaded += st.shaded; sc->n_nonfinite += st.nonfinite;
    sc->n_nodes_closest += st.tc_closest.nodes; sc->n_tris_closest += st.tc_closest.tris;
    sc->n_nodes_shadow += st.tc_shadow.nodes; sc->n_tris_shadow += st.tc_shadow.tris;
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  sc->n_paths += (uint64_t)W * H * nsamples;
  return 0;
}

// shaders/gmon.metal:14-55 — resolve `nBuckets` bucket images (back to back) into `out` (W*H*4).
int orc_gmon_resolve(const orc_scene* sc, const float* buckets, uint32_t nBuckets, float cap, float* out) {
  const size_t npix = (size_t)sc->params.width * sc->params.height;
  const float3 lw = f3(0.2126f, 0.7152f, 0.0722f);  // gmon.metal:10
  for (size_t p = 0; p < npix; p++) {
    float3 values[32];  // maxBuckets, gmon.metal:12
    for (uint32_t i = 0; i < nBuckets; i++) {
      const float* b = &buckets[4 * ((size_t)i * npix + p)];
      values[i] = f3(b[0], b[1], b[2]);
    }
    for (uint32_t i = nBuckets; i > 1; i--)
      for (uint32_t j = 1; j < i; j++)
        if (dot(values[j], lw) < dot(values[j - 1], lw)) {
          float3 temp = values[j - 1];
          values[j - 1] = values[j];
          values[j] = temp;
        }
    float3 sum = f3(0.0f), weightedSum = f3(0.0f);
    for (uint32_t i = 0; i < nBuckets; i++) {
      sum += values[i];
      weightedSum += (float)(i + 1) * values[i];
    }
    float G = (2.0f * dot(weightedSum, lw)) / ((float)nBuckets * dot(sum, lw)) - (float)(nBuckets + 1) / (float)nBuckets;
    G = fminf(G, cap);
    // int(G * float(nBuckets / 2)): a NaN / negative product (black pixel) is defined as 0 here (UB in MSL)
    const float cf = G * (float)(nBuckets / 2);
    const int c = cf > 0.0f ? (int)cf : 0;
    sum = f3(0.0f);
    for (int i = c; i < (int)nBuckets - c; i++) sum += values[i];
    float3 color = sum / (float)((int)nBuckets - 2 * c);
    out[4 * p] = color.x; out[4 * p + 1] = color.y; out[4 * p + 2] = color.z; out[4 * p + 3] = 1.0f;
  }
  return 0;
}

// Post-process + tonemap the W*H*4 float image `acc` into RGBA8 (readbackRenderTarget, renderer_pt.cpp:1039-1059).
int orc_postprocess(const orc_scene* sc, const float* acc, const pt_post_options* po, const pt_tonemap_options* to, uint8_t* rgba8_out,
                    float* float_out /* optional W*H*3 pre-quantisation */) {
  const uint32_t W = sc->params.width, H = sc->params.height;
  const Mat3 odt = compute_transform(sc->params.working_space, to->output_space);
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      float3 c = post::pixel(acc, W, H, x, y, *po, *to, odt);
      uint32_t v = post::pack_rgba8(c);
      memcpy(&rgba8_out[4 * ((size_t)y * W + x)], &v, 4);
      if (float_out) { float* f = &float_out[3 * ((size_t)y * W + x)]; f[0] = c.x; f[1] = c.y; f[2] = c.z; }
    }
  return 0;
}

int orc_trace_primary(orc_scene* sc, uint32_t sample_idx, pt_hit_record* out) {
  const uint32_t W = sc->params.width, H = sc->params.height;
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      HaltonSampler halton(x, y, sample_idx);
      float2 a = halton.sample2d();
      float2 b = halton.sample2d();
      Ray ray = spawnRayFromCamera(sc->constants.camera, x, y, a, b);
      float ir = halton.sample1d();
      Intersection is = intersect_scene<false>(*sc, ray, ir, nullptr);
      pt_hit_record& h = out[(size_t)y * W + x];
      h.t = is.hit ? is.distance : 0.0f; h.u = is.hit ? is.u : 0.0f; h.v = is.hit ? is.v : 0.0f;
      h.instance = is.hit ? (int32_t)is.instance_id : -1;
      h.primitive = is.hit ? (int32_t)is.primitive_id : -1;
    }
  return 0;
}

int orc_debug_sample(orc_scene* sc, uint32_t sample_idx, float* radiance_out, int32_t* hits_out, int threads) {
  const uint32_t W = sc->params.width, H = sc->params.height;
  const uint32_t B = sc->params.max_bounces;
  if (hits_out)
    for (size_t i = 0; i < (size_t)B * W * H * 2; i++) hits_out[i] = -1;
  if (threads < 1) threads = 1;
  std::atomic<uint32_t> next_row{0};
  auto worker = [&]() {
    ThreadStats st;
    for (;;) {
      uint32_t y = next_row.fetch_add(1);
      if (y >= H) break;
      for (uint32_t x = 0; x < W; x++) {
        PathLog log{hits_out, W * H, y * W + x};
        float3 L = trace_path(*sc, x, y, sample_idx, hits_out ? &log : nullptr, st, false);
        if (radiance_out) {
          float* px = &radiance_out[4 * ((size_t)y * W + x)];
          px[0] = L.x; px[1] = L.y; px[2] = L.z; px[3] = 1.0f;
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  return 0;
}

// A list of pixels through the very loop of orc_render: samples [first_sample, first_sample + nsamples) of pixel (xy[2i], xy[2i+1]) folded
// into out[4i..4i+3] with the running-mean rule of kernel.metal:672-684 (acc_n0 samples already in it; no GMoN).  What the full-size
// parity tests compare the HIP accumulator with at BASELINE.json's sizes, where a whole oracle frame would take minutes.
int orc_render_pixels(orc_scene* sc, const uint32_t* xy, uint32_t npixels, uint32_t first_sample, uint32_t nsamples, float* out,
                      uint32_t acc_n0, int threads) {
  if (threads < 1) threads = 1;
  std::atomic<uint32_t> next{0};
  auto worker = [&]() {
    ThreadStats st;
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= npixels) break;
      float* px = &out[4 * (size_t)i];
      for (uint32_t s = 0; s < nsamples; s++) {
        float3 L = trace_path(*sc, xy[2 * i], xy[2 * i + 1], first_sample + s, nullptr, st, false);
        if (!(fabsf(L.x) <= 3.0e38f && fabsf(L.y) <= 3.0e38f && fabsf(L.z) <= 3.0e38f)) {
          if (sc->params.nonfinite_policy == PT_NONFINITE_ZERO) L = f3(0.0f);
        }
        const uint32_t f = acc_n0 + s;
        if (f > 0) {
          float3 L_prev = f3(px[0], px[1], px[2]);
          L += L_prev * (float)f;
          L /= (float)(f + 1);
        }
        px[0] = L.x; px[1] = L.y; px[2] = L.z; px[3] = 1.0f;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  return 0;
}

// Verbose single-path trace to stderr (debugging aid).
int orc_debug_pixel(orc_scene* sc, uint32_t x, uint32_t y, uint32_t sample_idx, float* L_out) {
  ThreadStats st;
  st.verbose = true;
  float3 L = trace_path(*sc, x, y, sample_idx, nullptr, st, false);
  L_out[0] = L.x; L_out[1] = L.y; L_out[2] = L.z;
  return 0;
}

int orc_get_stats(const orc_scene* sc, orc_stats* out) {
  out->triangles = sc->tris.size();
  out->bvh_nodes = sc->bvh.size();
  out->closest_rays = sc->n_closest; out->shadow_rays = sc->n_shadow; out->shaded_hits = sc->n_shaded;
  out->paths = sc->n_paths;
  out->nodes_closest = sc->n_nodes_closest; out->tris_closest = sc->n_tris_closest;
  out->nodes_shadow = sc->n_nodes_shadow; out->tris_shadow = sc->n_tris_shadow;
  out->nonfinite = sc->n_nonfinite;
  return 0;
}

// ---- unit-level entry points for known-answer tests ----------------------------------------------------------
uint32_t orc_halton_offset(uint32_t x, uint32_t y, uint32_t sample) { return HaltonSampler(x, y, sample).m_offset; }
void orc_pcg4d(const uint32_t in[4], uint32_t out[4]) {
  uint4_ v = pcg4d({in[0], in[1], in[2], in[3]});
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
float orc_halton(uint32_t i, uint32_t d) { return HaltonSampler::halton(i, d); }
uint32_t orc_prime(uint32_t d) { return d < kNumPrimes ? g_primes.p[d] : 0; }
float orc_fresnel(float cosTheta, float ior) { return fresnel(cosTheta, ior); }
float orc_avg_dielectric_fresnel_fit(float ior) { return avgDielectricFresnelFit(ior); }
void orc_sincos(float x, float* s, float* c) { sincos_det(x, s, c); }
float orc_log2(float x) { return log2_det(x); }
float orc_exp2(float x) { return exp2_det(x); }
void orc_sample_cosine_hemisphere(float u0, float u1, float out[3]) {
  float3 w = sampleCosineHemisphere({u0, u1}); out[0] = w.x; out[1] = w.y; out[2] = w.z;
}
void orc_sample_tri_uniform(float u0, float u1, float out[2]) { float2 b = sampleTriUniform({u0, u1}); out[0] = b.x; out[1] = b.y; }
// ---- LUT generator restatement (pins A7-A9 against reference-held data) -------------------------------------------------
// The eight energy tables the renderer loads (resource/lut/*.exr -> platinum_amd/data/ggx_luts.bin) are Monte-Carlo integrals
// of the reference's own lobes, produced by its tool /root/reference/src/frontend/windows/tools/shaders/ms_lut_gen.metal:
//   :337-389 E            generateDirectionalAlbedoLookup          (cosTheta x, roughness y; the 0.961 "funny hack")
//   :395-441 E_avg        generateHemisphericalAlbedoLookup
//   :447-505 E_ms         generateMultiscatterDirectionalAlbedoLookup   (samples E and E_avg)
//   :511-571 E_ms_avg     generateMultiscatterHemisphericalAlbedoLookup
//   :576-650 E_trans_in / out   generateTransparentDirectionalAlbedoLookup
//   :656-743 E_trans_in_avg / out_avg   generateTransparentHemisphericalAlbedoLookup  (r.w == r.z: the same Halton dimension twice)
// Recomputing a texel with the ORACLE's fresnel / avgDielectricFresnelFit / sampleVmdf / refract / LUT sampling and the tool's
// integrand, and finding the value the reference committed, pins those functions against data the reference holds.
// The tool's GGX differs from the renderer's in two places, both restated here: lambda() is the textbook
// alpha^2 tan^2(theta) (:203-217; the renderer's isotropic lambda has no sin^2, bsdf.metal:173-182) and mdf() is written with
// tan^2 (:121-141).  `lambda_mode` bit 0 swaps in the renderer's lambda (to show that the tables tell the two apart), bit 1
// drops the multiscatter term of the E_ms integrand, bit 2 the 0.961 corner factor of E (see the FINDINGs below).
namespace lutgen {
struct GGXGen {  // ms_lut_gen.metal:110-218
  float ax, ay;
  int lambda_mode;
  GGXGen(float roughness, int mode) : lambda_mode(mode) { ax = ay = roughness * roughness; }
  float lambda(float3 w) const {
    if (lambda_mode == 1) return GGX(sqrtf(ax)).lambda(w);  // (never used for the pin itself)
    const float cos2Theta = w.z * w.z;
    const float sin2Theta = 1.0f - cos2Theta;
    const float tan2Theta = sin2Theta / cos2Theta;
    const float alpha2 = ax * ax;  // isotropic tables only
    return (sqrtf(1.0f + alpha2 * tan2Theta) - 1.0f) * 0.5f;
  }
  float mdf(float3 w) const {
    const float cos2Theta = w.z * w.z;
    const float sin2Theta = fmaxf(0.0f, 1.0f - cos2Theta);
    const float tan2Theta = sin2Theta / cos2Theta;
    const float cos4Theta = cos2Theta * cos2Theta;
    float k = tan2Theta;
    k /= (ax * ax);
    k = (1.0f + k) * (1.0f + k);
    return 1.0f / (PI_F * ax * ay * cos4Theta * k);
  }
  float g1(float3 w) const { return 1.0f / (1.0f + lambda(w)); }
  float g(float3 wo, float3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
  float vmdf(float3 w, float3 wm) const { return g1(w) / fabsf(w.z) * mdf(wm) * fabsf(dot(w, wm)); }
  float3 sampleVmdf(float3 w, float2 u) const { GGX r(0.0f); r.ax = ax; r.ay = ay; return r.sampleVmdf(w, u); }  // identical text (:166-181)
  float singleScatterBRDF(float3 wo, float3 wi, float3 wm) const { return mdf(wm) * g(wo, wi) / (4 * fabsf(wo.z) * fabsf(wi.z)); }
  float pdf(float3 wo, float3 wm) const { return vmdf(wo, wm) / (4.0f * fabsf(dot(wo, wm))); }
};
struct GenSample { float3 wi; float f, pdf; };

GenSample sampleSingleScatterGGX(float3 wo, const GGXGen& ggx, float2 r) {  // :232-247
  const float3 wm = ggx.sampleVmdf(wo, r);
  const float3 wi = reflect(-wo, wm);
  if (wm.z <= 0.0f || wo.z * wi.z < 0.0f) return {wi, 0.0f, 1.0f};
  return {wi, ggx.singleScatterBRDF(wo, wi, wm), ggx.pdf(wo, wm)};
}
GenSample sampleMultiscatterDielectricGGX(float3 wo, float ior, float roughness, const GGXGen& ggx, float2 r, const LutSet& luts,
                                          bool with_ms_term) {  // :252-282
  const float3 wm = ggx.sampleVmdf(wo, r);
  const float3 wi = reflect(-wo, wm);
  if (wo.z * wi.z < 0.0f) return {wi, 0.0f, 1.0f};
  const float cosTheta_o = fabsf(wo.z), cosTheta_i = fabsf(wi.z);
  const float brdf_ss = ggx.mdf(wm) * ggx.g(wo, wi) / (4 * cosTheta_o * cosTheta_i);
  const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ior);
  const float E_wo = lut2(luts.E, wo.z, roughness);
  const float E_wi = lut2(luts.E, wi.z, roughness);
  const float E_avg = lut1(luts.Eavg, roughness);
  const float F_avg = avgDielectricFresnelFit(ior);
  const float brdf_ms = (1.0f - E_wo) * (1.0f - E_wi) / (PI_F * (1.0f - E_avg));
  const float fresnel_ms = F_avg * F_avg * E_avg / (1.0f - F_avg * (1.0f - E_avg));
  // FINDING (tools/lut_pin.py): the committed ggx_ms_E_*.exr / ggx_ms_E_avg.exr equal this integral WITHOUT the
  // fresnel_ms * brdf_ms term to ~1e-4 at every texel (and differ from the integral as written by up to 0.44 at high ior and
  // roughness): the reference's data files hold the single-scatter, Fresnel-weighted albedo.  with_ms_term = false reproduces them.
  const float f = with_ms_term ? fresnel_ss * brdf_ss + fresnel_ms * brdf_ms : fresnel_ss * brdf_ss;
  return {wi, f, ggx.vmdf(wo, wm) / (4.0f * fabsf(dot(wo, wm)))};
}
GenSample sampleTransparentDielectricGGX(float3 wo, const GGXGen& ggx, float ior, float3 r) {  // :287-331 (thin = false)
  const float3 wm = ggx.sampleVmdf(wo, {r.x, r.y});
  const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ior);
  float3 wi;
  if (r.z < fresnel_ss) {
    wi = reflect(-wo, wm);
    if (wo.z * wi.z < 0.0f) return {wi, 0.0f, 1.0f};
  } else {
    wi = refract(-wo, wm * sign(dot(wo, wm)), 1.0f / ior);
    if (wo.z * wi.z >= 0.0f) return {wi, 0.0f, 1.0f};
  }
  const bool isReflection = wo.z * wi.z > 0.0f;
  float bsdf, pdf;
  if (isReflection) {
    bsdf = ggx.singleScatterBRDF(wo, wi, wm);
    pdf = ggx.pdf(wo, wm);
  } else {
    float denom = dot(wi, wm) * ior + dot(wo, wm);
    denom *= denom;
    const float dwm_dwi = fabsf(dot(wi, wm)) / denom;
    bsdf = ggx.mdf(wm) * ggx.g(wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * denom));
    pdf = ggx.vmdf(wo, wm) * dwm_dwi;
  }
  const float k = isReflection ? fresnel_ss : 1.0f - fresnel_ss;
  return {wi, k * bsdf, k * pdf};
}
}  // namespace lutgen

// The generator's integral for table `which` (0 E, 1 E_avg, 2 E_ms, 3 E_ms_avg, 4 E_trans_in, 5 E_trans_out, 6 E_trans_in_avg,
// 7 E_trans_out_avg) at explicit parameters, `nsamples` Halton samples from index `offset` (the tool uses a random per-texel
// offset, ms_lut_gen.cpp:235-237).  cosTheta is ignored by the hemispherical tables (they draw it from dimension 2).
double orc_lut_regen(const orc_scene* sc, int which, float cosTheta, float roughness, float ior, uint32_t nsamples, uint32_t offset,
                     int lambda_mode) {
  using namespace lutgen;
  auto H = [&](uint32_t i, uint32_t d) {  // the tool's own halton (no clamp): ms_lut_gen.metal:30-45
    uint32_t b = g_primes.p[d];
    float f = 1.0f, invB = 1.0f / (float)b, r = 0;
    while (i > 0) { f = f * invB; r = r + f * (float)(i % b); i = i / b; }
    return r;
  };
  const GGXGen ggx(roughness, lambda_mode & 1);
  const bool with_ms = (lambda_mode & 2) == 0;    // mode bit 0: the renderer's lambda; bit 1: E_ms tables without the multiscatter term
  const bool with_hack = (lambda_mode & 4) == 0;  // bit 2: E without the 0.961 corner factor
  double sum = 0.0;
  for (uint32_t s = 0; s < nsamples; s++) {
    const uint32_t idx = offset + s;
    const float r0 = H(idx, 0), r1 = H(idx, 1), r2 = H(idx, 2);
    float v = 0.0f;
    switch (which) {
      case 0: {
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const GenSample sm = sampleSingleScatterGGX(f3(sinTheta, 0.0f, cosTheta), ggx, {r0, r1});
        v = sm.f * fabsf(sm.wi.z) / sm.pdf;
        // "Funny hack" (:371-374).  FINDING: the committed ggx_E.exr does not carry it (its 4 x 8 corner texels are ~1.0 where
        // the hack gives 0.961): mode bit 2 leaves it out
        if (with_hack && roughness < 2.0f / 32.0f && cosTheta < 1.0f / 32.0f) v *= 0.961f;
        break;
      }
      case 1: {
        const float ct = r2, sinTheta = sqrtf(1.0f - ct * ct);
        const float3 wo = f3(sinTheta, 0.0f, ct);
        const GenSample sm = sampleSingleScatterGGX(wo, ggx, {r0, r1});
        v = 2.0f * sm.f * fabsf(sm.wi.z) * wo.z / sm.pdf;
        break;
      }
      case 2: {
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const GenSample sm = sampleMultiscatterDielectricGGX(f3(sinTheta, 0.0f, cosTheta), ior, roughness, ggx, {r0, r1}, sc->luts, with_ms);
        v = sm.f * fabsf(sm.wi.z) / sm.pdf;
        break;
      }
      case 3: {
        const float ct = r2, sinTheta = sqrtf(1.0f - ct * ct);
        const float3 wo = f3(sinTheta, 0.0f, ct);
        const GenSample sm = sampleMultiscatterDielectricGGX(wo, ior, roughness, ggx, {r0, r1}, sc->luts, with_ms);
        v = 2.0f * sm.f * fabsf(sm.wi.z) * fabsf(wo.z) / sm.pdf;
        break;
      }
      case 4: case 5: {
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float3 wo = f3(sinTheta, 0.0f, cosTheta * (which == 5 ? -1.0f : 1.0f));
        const GenSample sm = sampleTransparentDielectricGGX(wo, ggx, ior, f3(r0, r1, r2));
        v = sm.f * fabsf(sm.wi.z) / sm.pdf;
        break;
      }
      default: {
        const float ct = r2 * 2.0f - 1.0f;  // r.w: halton dimension 2 again (:709-712)
        const float sinTheta = sqrtf(1.0f - ct * ct);
        const GenSample sm = sampleTransparentDielectricGGX(f3(sinTheta, 0.0f, ct), ggx, ior, f3(r0, r1, r2));
        v = sm.f * fabsf(sm.wi.z) / sm.pdf;
        break;
      }
    }
    if (v == v) sum += (double)v;  // (a 0/0 sample at a grazing texel contributes nothing)
  }
  return sum / (double)nsamples;
}

// One texel of table `which`, parameters from the texel centre exactly as the tool computes them (:351-352, :411, :465-472,
// :527-533, :594-601, :674-680)
double orc_lut_regen_texel(const orc_scene* sc, int which, int x, int y, int z, uint32_t nsamples, uint32_t seed, int lambda_mode) {
  const Lut* ls[8] = {&sc->luts.E, &sc->luts.Eavg, &sc->luts.EMs, &sc->luts.EavgMs, &sc->luts.ETransIn, &sc->luts.ETransOut,
                      &sc->luts.EavgTransIn, &sc->luts.EavgTransOut};
  const float size = (float)ls[which]->w;
  const uint32_t offset = pcg4d({(uint32_t)x, (uint32_t)y, (uint32_t)z + 977u * (uint32_t)which, seed}).x % (1024u * 1024u);
  float cosTheta = 0.0f, roughness = 0.0f, iorParam = 0.0f;
  bool out = false;
  switch (which) {
    case 0: roughness = ((float)y + 0.5f) / size; cosTheta = ((float)x + 0.5f) / size; break;
    case 1: roughness = ((float)x + 0.5f) / size; break;
    case 2: case 4: case 5:
      iorParam = ((float)z + 0.5f) / size; roughness = ((float)y + 0.5f) / size; cosTheta = ((float)x + 0.5f) / size; out = which == 5; break;
    default: roughness = ((float)y + 0.5f) / size; iorParam = ((float)x + 0.5f) / size; out = which == 7; break;
  }
  const float ior = out ? 1.0f - iorParam : 1.0f / (1.0f - iorParam);
  return orc_lut_regen(sc, which, cosTheta, roughness, ior, nsamples, offset, lambda_mode);
}

// the committed value of that texel
float orc_lut_texel(const orc_scene* sc, int which, int x, int y, int z) {
  const Lut* ls[8] = {&sc->luts.E, &sc->luts.Eavg, &sc->luts.EMs, &sc->luts.EavgMs, &sc->luts.ETransIn, &sc->luts.ETransOut,
                      &sc->luts.EavgTransIn, &sc->luts.EavgTransOut};
  const Lut& l = *ls[which];
  return l.d[((size_t)z * l.h + y) * l.w + x];
}

float orc_lut_sample(const orc_scene* sc, int which, float cx, float cy, float cz) {
  const Lut* ls[8] = {&sc->luts.E, &sc->luts.Eavg, &sc->luts.EMs, &sc->luts.EavgMs, &sc->luts.ETransIn, &sc->luts.ETransOut,
                      &sc->luts.EavgTransIn, &sc->luts.EavgTransOut};
  const Lut& l = *ls[which];
  if (l.depth > 1) return lut3(l, cx, cy, cz);
  if (l.h > 1) return lut2(l, cx, cy);
  return lut1(l, cx);
}

// BSDF sample/eval on a bare material (tangent space), for furnace / reciprocity style checks.
//   out_sample: wi[3], f[3], Le[3], pdf, flags(as float)  = 11 floats
void orc_bsdf_sample(const orc_scene* sc, const pt_material_gpu* mat, const float wo[3], const float r[4], const float rc[2],
                     float out_sample[11]) {
  ShadingContext ctx(*mat, float2{0.0f, 0.0f}, sc->idt, sc->textures);
  BSDF bsdf(ctx, sc->constants.flags, sc->luts);
  Sample s = bsdf.sample(f3(wo[0], wo[1], wo[2]), {r[0], r[1], r[2], r[3]}, {rc[0], rc[1]});
  float o[11] = {s.wi.x, s.wi.y, s.wi.z, s.f.x, s.f.y, s.f.z, s.Le.x, s.Le.y, s.Le.z, s.pdf, (float)s.flags};
  memcpy(out_sample, o, sizeof(o));
}
//   out_eval: f[3], pdf = 4 floats
void orc_bsdf_eval(const orc_scene* sc, const pt_material_gpu* mat, const float wo[3], const float wi[3], float out_eval[4]) {
  ShadingContext ctx(*mat, float2{0.0f, 0.0f}, sc->idt, sc->textures);
  BSDF bsdf(ctx, sc->constants.flags, sc->luts);
  Eval e = bsdf.eval(f3(wo[0], wo[1], wo[2]), f3(wi[0], wi[1], wi[2]));
  out_eval[0] = e.f.x; out_eval[1] = e.f.y; out_eval[2] = e.f.z; out_eval[3] = e.pdf;
}

}  // extern "C"
