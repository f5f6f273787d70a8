// pt_bsdf.h — the reference's principled GGX BSDF on the device (src/renderer_pt/shaders/bsdf.metal, decls
// defs.metal:225-394).  Lobes: clearcoat / metallic / transparent dielectric (thin & refractive) / opaque
// dielectric (GGX + energy-compensated diffuse); Kulla-Conty & Turquin multiple scattering through the LUTs.
// Reference quirks are kept on purpose: `Eval{}` defaults to pdf = 1, emission is returned only
// by the diffuse branch as Le / (1 - blend), eval() is 0 below z = 1.5e-3, the isotropic lambda() has no sin^2.
//
// LUT filtering (Apple's texture unit is closed; this is the build's definition): clamp-to-edge,
// x = c*N - 0.5, i0 = floor(x), taps clamped to [0, N-1], lerp a + (b - a) * w; x, then y, then z.
#pragma once
#include "pt_sampler.h"

namespace pt {

enum SampleFlags {  // defs.metal:264-272
  Sample_Absorbed = 0, Sample_Emitted = 1 << 0, Sample_Reflected = 1 << 1, Sample_Transmitted = 1 << 2,
  Sample_Diffuse = 1 << 3, Sample_Glossy = 1 << 4, Sample_Specular = 1 << 5,
};

struct LutAxis { int i0, i1; float w; };
PT_HD LutAxis lut_axis(float c, int n) {
  float x = c * (float)n - 0.5f;
  float fl = floorf(x);
  LutAxis a;
  a.w = x - fl;
  int i = (int)fl;
  a.i0 = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
  int j = i + 1;
  a.i1 = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
  return a;
}
// one texel: from the block's LDS copy when the table was staged there, from HBM (global_load) otherwise
PT_HD float lut_tap(const Lut& l, int i) { return l.lds ? l.d[i] : ldg(&l.d[i]); }
PT_HD float lut1(const Lut& l, float cx) {
  LutAxis ax = lut_axis(cx, l.w);
  float a = lut_tap(l, ax.i0), b = lut_tap(l, ax.i1);
  return a + (b - a) * ax.w;
}
PT_HD float lut2_slice(const Lut& l, int base, int W, const LutAxis& ax, const LutAxis& ay) {
  float t00 = lut_tap(l, base + ay.i0 * W + ax.i0), t01 = lut_tap(l, base + ay.i0 * W + ax.i1);
  float t10 = lut_tap(l, base + ay.i1 * W + ax.i0), t11 = lut_tap(l, base + ay.i1 * W + ax.i1);
  float a = t00 + (t01 - t00) * ax.w;
  float b = t10 + (t11 - t10) * ax.w;
  return a + (b - a) * ay.w;
}
PT_HD float lut2(const Lut& l, float cx, float cy) {
  LutAxis ax = lut_axis(cx, l.w), ay = lut_axis(cy, l.h);
  return lut2_slice(l, 0, l.w, ax, ay);
}
PT_HD float lut3(const Lut& l, float cx, float cy, float cz) {
  LutAxis ax = lut_axis(cx, l.w), ay = lut_axis(cy, l.h), az = lut_axis(cz, l.depth);
  float a = lut2_slice(l, az.i0 * (l.w * l.h), l.w, ax, ay);
  float b = lut2_slice(l, az.i1 * (l.w * l.h), l.w, ax, ay);
  return a + (b - a) * az.w;
}

// defs.metal:283-299, bsdf.metal:12-43
struct ShadingContext {
  vec3 albedo;
  float roughness, metallic, transmission, clearcoat, clearcoatRoughness, anisotropy, ior;
  int flags;
  vec3 emission;
};
// sampler(address::repeat, filter::linear) on a decoded float4 texture (filtering contract: DESIGN.md §2)
PT_HD int tex_wrap(int i, int n) { const int m = i % n; return m < 0 ? m + n : m; }
// texel i of a texture as linear float4 (TexInfo: the stored format, decoded through the host's own tables)
PT_HD vec4 tex_fetch(const DeviceScene& S, const TexInfo& t, size_t i) {
  const uint8_t* base = S.tex_data + (size_t)t.offset16 * 16u;
  if (t.format == PT_TEX_RGBA32F) return ldg(reinterpret_cast<const vec4*>(base) + i);
  const float* unorm = S.tex_decode + kTexDecodeUnorm;
  if (t.format == PT_TEX_RGBA8_SRGB || t.format == PT_TEX_RGBA8) {
    const uint32_t v = ldg(reinterpret_cast<const uint32_t*>(base) + i);
    const float* colour = S.tex_decode + (t.format == PT_TEX_RGBA8_SRGB ? kTexDecodeSrgb : kTexDecodeUnorm);
    return vec4{ldg(&colour[v & 0xffu]), ldg(&colour[(v >> 8) & 0xffu]), ldg(&colour[(v >> 16) & 0xffu]), ldg(&unorm[v >> 24])};
  }
  if (t.format == PT_TEX_RG8) {
    const uint16_t v = ldg(reinterpret_cast<const uint16_t*>(base) + i);
    return vec4{ldg(&unorm[v & 0xffu]), ldg(&unorm[v >> 8]), 0.0f, 1.0f};
  }
  return vec4{ldg(&unorm[ldg(base + i)]), 0.0f, 0.0f, 1.0f};  // PT_TEX_R8
}
// The storage form is a property of the SCENE (host_scene.h decode_textures): when no texture kept an 8-bit form (S.tex_native == 0, a
// wave-uniform scalar) every texel is a float4 and the four taps are four plain loads — the format chain of tex_fetch is not even
// entered.  r4 ran that chain inside each tap of every sample (C5 k_shade +8.5 %).
PT_HD vec4 tex_bilerp(const vec4& p00, const vec4& p01, const vec4& p10, const vec4& p11, float wx, float wy) {
  vec4 o;
  { const float a = p00.x + (p01.x - p00.x) * wx, b = p10.x + (p11.x - p10.x) * wx; o.x = a + (b - a) * wy; }
  { const float a = p00.y + (p01.y - p00.y) * wx, b = p10.y + (p11.y - p10.y) * wx; o.y = a + (b - a) * wy; }
  { const float a = p00.z + (p01.z - p00.z) * wx, b = p10.z + (p11.z - p10.z) * wx; o.z = a + (b - a) * wy; }
  { const float a = p00.w + (p01.w - p00.w) * wx, b = p10.w + (p11.w - p10.w) * wx; o.w = a + (b - a) * wy; }
  return o;
}
PT_HD vec4 tex_sample(const DeviceScene& S, int id, vec2 uv) {
  const TexInfo t = ldg(&S.textures[id]);
  const float fx = uv.x * (float)t.w - 0.5f, fy = uv.y * (float)t.h - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx = fx - x0f, wy = fy - y0f;
  const int x0 = tex_wrap((int)x0f, (int)t.w), x1 = tex_wrap((int)x0f + 1, (int)t.w);
  const int y0 = tex_wrap((int)y0f, (int)t.h), y1 = tex_wrap((int)y0f + 1, (int)t.h);
  const size_t i00 = (size_t)y0 * t.w + x0, i01 = (size_t)y0 * t.w + x1, i10 = (size_t)y1 * t.w + x0, i11 = (size_t)y1 * t.w + x1;
  if (!S.tex_native) {
    const vec4* texels = reinterpret_cast<const vec4*>(S.tex_data + (size_t)t.offset16 * 16u);
    const vec4 p00 = ldg(texels + i00), p01 = ldg(texels + i01), p10 = ldg(texels + i10), p11 = ldg(texels + i11);
    return tex_bilerp(p00, p01, p10, p11, wx, wy);
  }
  const vec4 p00 = tex_fetch(S, t, i00), p01 = tex_fetch(S, t, i01), p10 = tex_fetch(S, t, i10), p11 = tex_fetch(S, t, i11);
  return tex_bilerp(p00, p01, p10, p11, wx, wy);
}

PT_HD ShadingContext make_shading_context(const DeviceScene& S, const pt_material_gpu& mat, vec2 uv) {
  const Mat3& idt = S.idt;
  ShadingContext c;
  c.albedo = v3(mat.baseColor[0], mat.baseColor[1], mat.baseColor[2]);
  c.emission = v3(mat.emission.x, mat.emission.y, mat.emission.z);
  c.roughness = mat.roughness;
  c.metallic = mat.metallic;
  c.transmission = mat.transmission;
  c.clearcoat = mat.clearcoat;
  c.clearcoatRoughness = mat.clearcoatRoughness;
  c.anisotropy = mat.anisotropy;
  c.ior = mat.ior;
  c.flags = mat.flags;
  if (mat.baseTextureId >= 0) { const vec4 t = tex_sample(S, mat.baseTextureId, uv); c.albedo = v3(t.x, t.y, t.z); }  // bsdf.metal:25-26
  if (mat.emissionTextureId >= 0) { const vec4 t = tex_sample(S, mat.emissionTextureId, uv); c.emission = c.emission * v3(t.x, t.y, t.z); }
  if (mat.transmissionTextureId >= 0) c.transmission = tex_sample(S, mat.transmissionTextureId, uv).x;
  if (mat.clearcoatTextureId >= 0) c.clearcoat = tex_sample(S, mat.clearcoatTextureId, uv).x;
  if (mat.rmTextureId >= 0) {  // bsdf.metal:33-37
    const vec4 rm = tex_sample(S, mat.rmTextureId, uv);
    c.roughness = c.roughness * rm.x;
    c.metallic = c.metallic * rm.y;
  }
  c.albedo = mul(idt, c.albedo);
  c.emission = mul(idt, c.emission);
  c.emission = c.emission * mat.emissionStrength;
  return c;
}

struct BsdfSample { vec3 wi; vec3 f; vec3 Le; float pdf; int flags; };  // defs.metal:301-307
PT_HD BsdfSample sample_none() { return {v3(0.0f), v3(0.0f), v3(0.0f), 0.0f, 0}; }
struct BsdfEval { vec3 f; float pdf; };                                    // defs.metal:309-328 (Le is always 0)
PT_HD BsdfEval eval_default() { return {v3(0.0f), 1.0f}; }                 // `return {}` : pdf defaults to 1

PT_HD vec3 schlick(vec3 f0, float cosTheta) {  // bsdf.metal:49-53
  const float k = 1.0f - cosTheta;
  const float k2 = k * k;
  return f0 + (v3(1.0f) - f0) * (k2 * k2 * k);
}
PT_HD float fresnel(float cosTheta, float ior) {  // bsdf.metal:72-85
  cosTheta = saturate(cosTheta);
  const float sin2Theta_t = (1.0f - cosTheta * cosTheta) / (ior * ior);
  if (sin2Theta_t >= 1.0f) return 1.0f;
  const float cosTheta_t = sqrtf(1.0f - sin2Theta_t);
  const float parallel = (ior * cosTheta - cosTheta_t) / (ior * cosTheta + cosTheta_t);
  const float perpendicular = (cosTheta - ior * cosTheta_t) / (cosTheta + ior * cosTheta_t);
  return (parallel * parallel + perpendicular * perpendicular) * 0.5f;
}
PT_HD float avgDielectricFresnelFit(float ior) {  // bsdf.metal:92-96
  return ior >= 1.0f ? (ior - 1.0f) / (4.08567f + 1.00071f * ior)
                     : 0.997118f + 0.1014f * ior - 0.965241f * ior * ior - 0.130607f * ior * ior * ior;
}

struct GGX {  // bsdf.metal:102-183
  float ax, ay;
  PT_HD float lambda(vec3 w) const {
    const float cos2Theta = w.z * w.z;
    float alpha2 = ax * ax;
    if (ax != ay) alpha2 = alpha2 * w.x * w.x + ay * ay * w.y * w.y;
    return (sqrtf(1.0f + alpha2 / cos2Theta) - 1.0f) * 0.5f;
  }
  PT_HD float mdf(vec3 w) const {
    const float cos2Theta = w.z * w.z;
    const float cos4Theta = cos2Theta * cos2Theta;
    float k = 1.0f / cos2Theta * (w.x * w.x / (ax * ax) + w.y * w.y / (ay * ay));
    k = (1.0f + k) * (1.0f + k);
    return 1.0f / (kPi * ax * ay * cos4Theta * k);
  }
  PT_HD float g1(vec3 w) const { return 1.0f / (1.0f + lambda(w)); }
  PT_HD float g(vec3 wo, vec3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
  PT_HD float vmdf(vec3 w, vec3 wm) const { return g1(w) / fabsf(w.z) * mdf(wm) * fabsf(dot(w, wm)); }
  PT_HD vec3 sampleVmdf(vec3 w, vec2 u) const {
    vec3 wh = normalize(w * v3(ax, ay, 1.0f));
    if (wh.z < 0) wh = wh * -1.0f;
    const vec3 b = (wh.z < 0.9999f) ? normalize(cross(v3(0.0f, 0.0f, 1.0f), wh)) : v3(1.0f, 0.0f, 0.0f);
    const vec3 t = cross(wh, b);
    vec2 p = sampleDisk(u);
    const float h = sqrtf(1.0f - p.x * p.x);
    p.y = mix(h, p.y, 0.5f * wh.z + 0.5f);
    const float pz = sqrtf(fmaxf(0.0f, 1.0f - (p.x * p.x + p.y * p.y)));
    const vec3 nh = (p.x * b + p.y * t) + pz * wh;
    return normalize(v3(ax * nh.x, ay * nh.y, fmaxf(1e-6f, nh.z)));
  }
  PT_HD float singleScatterBRDF(vec3 wo, vec3 wi, vec3 wm) const {
    return mdf(wm) * g(wo, wi) / (4 * fabsf(wo.z) * fabsf(wi.z));
  }
  PT_HD float pdf(vec3 wo, vec3 wm) const { return vmdf(wo, wm) / (4.0f * fabsf(dot(wo, wm))); }
  PT_HD bool isSmooth() const { return ax < 1e-3f && ay < 1e-3f; }
};
PT_HD GGX make_ggx(float roughness) { float a = roughness * roughness; return {a, a}; }          // bsdf.metal:103
PT_HD GGX make_ggx(float roughness, float anisotropic) {                                          // bsdf.metal:105-109
  const float alpha = roughness * roughness;
  const float aspect = sqrtf(1.0f - 0.9f * anisotropic);
  return {alpha / aspect, alpha * aspect};
}

struct BSDF {
  const ShadingContext& ctx;
  const LutSet& luts;
  GGX ggx, ggxCoat;
  bool ms;  // RendererFlags_MultiscatterGGX
  static constexpr float kClearcoatIor = 1.5f;  // defs.metal:345

  // The energy-table terms that depend only on the outgoing direction and the material: the reference evaluates them
  // again in every lobe of sample() and of the NEE eval() (bsdf.metal:291-326, defs.metal:349-361); a shaded hit uses ONE
  // wo for both calls, so they are looked up once here — same functions, same arguments, same bits.
  float cE_wo = 0.0f, cE_avg = 0.0f, cEms_wo = 0.0f, cEms_avg = 0.0f;

  PT_HD BSDF(const ShadingContext& c, int rendererFlags, const LutSet& l, vec3 wo)
      : ctx(c), luts(l), ggx(make_ggx(c.roughness, c.anisotropy)), ggxCoat(make_ggx(c.clearcoatRoughness)),
        ms((rendererFlags & PT_FLAG_MULTISCATTER_GGX) != 0) {
    const float m = c.metallic, t = c.transmission;
    // the opaque lobe can be SAMPLED with probability (1-m)(1-t) and is EVALUATED with weight (1-m)(1-(1-m)t)
    const bool opaque_lobe = (1.0f - m) * (1.0f - t) > 0.0f || (1.0f - m) * (1.0f - (1.0f - m) * t) > 0.0f;
    if (opaque_lobe || (ms && m > 0.0f)) cE_wo = lut2(luts.E, wo.z, ctx.roughness);
    if (ms && (opaque_lobe || m > 0.0f)) cE_avg = lut1(luts.Eavg, ctx.roughness);
    if (opaque_lobe) {
      const float iorParam = (ctx.ior - 1.0f) / ctx.ior;
      cEms_wo = lut3(luts.EMs, wo.z, ctx.roughness, iorParam);
      cEms_avg = lut2(luts.EavgMs, iorParam, ctx.roughness);
    }
  }

  // defs.metal:349-361 — the E / Eavg part is shared by the float and float3 instantiations
  PT_HD void multiscatter_terms(vec3 wo, vec3 wi, float* brdf_ms, float* E_avg) const {
    const float E_wo = cE_wo;
    const float E_wi = lut2(luts.E, wi.z, ctx.roughness);
    *E_avg = cE_avg;
    *brdf_ms = (1.0f - E_wo) * (1.0f - E_wi) / (kPi * (1.0f - *E_avg));
  }
  PT_HD float multiscatter(vec3 wo, vec3 wi, float F_avg) const {
    float brdf_ms, E_avg;
    multiscatter_terms(wo, wi, &brdf_ms, &E_avg);
    const float fresnel_ms = F_avg * F_avg * E_avg / (1.0f - F_avg * (1.0f - E_avg));
    return fresnel_ms * brdf_ms;
  }
  PT_HD vec3 multiscatter(vec3 wo, vec3 wi, vec3 F_avg) const {
    float brdf_ms, E_avg;
    multiscatter_terms(wo, wi, &brdf_ms, &E_avg);
    const vec3 fresnel_ms = F_avg * F_avg * E_avg / (v3(1.0f) - F_avg * (1.0f - E_avg));
    return fresnel_ms * brdf_ms;
  }
  PT_HD float transparentMultiscatter(vec3 wo, float ior) const {  // bsdf.metal:262-284
    if (ior < 1.0f) {
      const float E_wo = lut3(luts.ETransOut, fabsf(wo.z), ctx.roughness, 1.0f - ior);
      return 1.0f / E_wo;
    }
    const float E_wo = lut3(luts.ETransIn, fabsf(wo.z), ctx.roughness, (ior - 1.0f) / ior);
    return 1.0f / E_wo;
  }
  PT_HD float diffuseFactor(vec3 wo, vec3 wi) const {  // bsdf.metal:291-305
    const float iorParam = (ctx.ior - 1.0f) / ctx.ior;
    const float E_ms_wo = cEms_wo;
    const float E_ms_wi = lut3(luts.EMs, wi.z, ctx.roughness, iorParam);
    const float E_ms_avg = cEms_avg;
    return (1.0f - E_ms_wo) * (1.0f - E_ms_wi) / (kPi * (1.0f - E_ms_avg));
  }
  PT_HD float opaqueDielectricFactor(vec3 wo, float F_avg) const {  // bsdf.metal:311-326
    const float E_wo = cE_wo;
    const float E_ms_wo = cEms_wo;
    const float fresnel_ms = F_avg * F_avg * E_wo / (1.0f - F_avg * (1.0f - E_wo));
    return F_avg * E_ms_wo + fresnel_ms * (1.0f - E_ms_wo);
  }

  // ---- evaluation ----------------------------------------------------------------------------------------------
  PT_HD BsdfEval evalMetallicWm(vec3 wo, vec3 wi, vec3 wm) const {  // bsdf.metal:339-355
    const vec3 fresnel_ss = schlick(ctx.albedo, fabsf(dot(wo, wm)));
    vec3 brdf = fresnel_ss * ggx.singleScatterBRDF(wo, wi, wm);
    if (ms) {
      const vec3 F_avg = (20.0f * ctx.albedo + v3(1.0f)) / 21.0f;
      brdf = brdf + multiscatter(wo, wi, F_avg);
    }
    return {brdf, ggx.pdf(wo, wm)};
  }
  PT_HD BsdfEval evalMetallic(vec3 wo, vec3 wi) const {  // bsdf.metal:360-370
    if (ggx.isSmooth()) return eval_default();
    vec3 wm = normalize(wo + wi);
    if (length_squared(wm) == 0.0f) return eval_default();
    wm = wm * sign(wm.z);
    return evalMetallicWm(wo, wi, wm);
  }
  PT_HD BsdfEval evalTransparentWm(vec3 wo, vec3 wi, vec3 wm, float fresnel_ss, float ior) const {  // :377-419
    const bool thin = (ctx.flags & PT_MATERIAL_THIN_DIELECTRIC) != 0;
    const bool isReflection = wo.z * wi.z > 0.0f;
    vec3 bsdf;
    float pdf, k = fresnel_ss;
    if (isReflection) {
      bsdf = v3(ggx.singleScatterBRDF(wo, wi, wm));
      pdf = ggx.pdf(wo, wm);
    } else {
      k = 1.0f - fresnel_ss;
      float btdf_ss;
      if (thin) {
        btdf_ss = ggx.singleScatterBRDF(wo, wi, wm);
        pdf = ggx.pdf(wo, wm);
      } else {
        float denom = dot(wi, wm) * ior + dot(wo, wm);
        denom = denom * denom;
        const float dwm_dwi = fabsf(dot(wi, wm)) / denom;
        btdf_ss = ggx.mdf(wm) * ggx.g(wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm) / (wi.z * wo.z * denom));
        pdf = ggx.vmdf(wo, wm) * dwm_dwi;
      }
      bsdf = ctx.albedo * btdf_ss;
    }
    if (ms) bsdf = bsdf * transparentMultiscatter(wo, ior);
    return {k * bsdf, k * pdf};
  }
  PT_HD BsdfEval evalTransparent(vec3 wo, vec3 wi) const {  // bsdf.metal:424-446
    if (ggx.isSmooth()) return eval_default();
    const bool thin = (ctx.flags & PT_MATERIAL_THIN_DIELECTRIC) != 0;
    const float ior = (!thin && wo.z < 0.0f && wi.z < 0.0f) ? 1.0f / ctx.ior : ctx.ior;
    vec3 wm = ior * wi + wo;
    if (wi.z == 0 || wo.z == 0 || wm.z == 0) return eval_default();
    wm = normalize(wm * sign(wm.z));
    if (dot(wi, wm) * wi.z < 0.0f || dot(wo, wm) * wo.z < 0.0f) return eval_default();
    if (thin) {
      wi = reflect(wi, v3(0.0f, 0.0f, 1.0f));
      wm = normalize(wi + wo);
    }
    const float fresnel_ss = fresnel(dot(wo, wm), ior);
    return evalTransparentWm(wo, wi, wm, fresnel_ss, ior);
  }
  PT_HD BsdfEval evalOpaque(vec3 wo, vec3 wi) const {  // bsdf.metal:451-486
    const float F_avg = avgDielectricFresnelFit(ctx.ior);
    const float blendingFactor = opaqueDielectricFactor(wo, F_avg);
    const float cDiffuse = diffuseFactor(wo, wi);
    const float diffusePdf = fabsf(wi.z) / kPi;
    if (ggx.isSmooth()) return {ctx.albedo * cDiffuse, diffusePdf * (1.0f - blendingFactor)};
    vec3 wm = normalize(wo + wi);
    if (length_squared(wm) == 0.0f) return eval_default();
    wm = wm * sign(wm.z);
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ctx.ior);
    float dielectricBrdf = fresnel_ss * ggx.singleScatterBRDF(wo, wi, wm);
    if (ms) dielectricBrdf += multiscatter(wo, wi, F_avg);
    return {v3(dielectricBrdf) + ctx.albedo * cDiffuse,
            ggx.pdf(wo, wm) * blendingFactor + diffusePdf * (1.0f - blendingFactor)};
  }
  // bsdf.metal:488-503; *fresnel_ss is only written on the full path (the reference leaves it undefined on the
  // early returns; the caller zero-initialises it)
  PT_HD BsdfEval evalClearcoat(vec3 wo, vec3 wi, float* fresnel_ss) const {
    if (ggxCoat.isSmooth()) return eval_default();
    vec3 wm = wo + wi;
    wm = normalize(wm * sign(wm.z));
    if (length_squared(wm) == 0.0f) return eval_default();
    *fresnel_ss = fresnel(dot(wo, wm), kClearcoatIor);
    return {v3(ggxCoat.singleScatterBRDF(wo, wi, wm)), ggxCoat.pdf(wo, wm)};
  }
  PT_HD BsdfEval eval(vec3 wo, vec3 wi) const {  // bsdf.metal:199-223
    if (wo.z < 1.5e-3f || wi.z < 1.5e-3f) return eval_default();
    const float metallic = ctx.metallic;
    const float transparent = (1.0f - metallic) * ctx.transmission;
    const float opaque = (1.0f - metallic) * (1.0f - transparent);
    BsdfEval result = {v3(0.0f), 0.0f};
    if (metallic > 0.0f) {
      BsdfEval e = evalMetallic(wo, wi);
      result.f = result.f + e.f * metallic;
      result.pdf = result.pdf + e.pdf * metallic;
    }
    if (transparent > 0.0f) {
      BsdfEval e = evalTransparent(wo, wi);
      result.f = result.f + e.f * transparent;
      result.pdf = result.pdf + e.pdf * transparent;
    }
    if (opaque > 0.0f) {
      BsdfEval e = evalOpaque(wo, wi);
      result.f = result.f + e.f * opaque;
      result.pdf = result.pdf + e.pdf * opaque;
    }
    float coat = ctx.clearcoat;
    if (coat > 0.0f) {
      float coatFresnel_ss = 0.0f;
      BsdfEval c = evalClearcoat(wo, wi, &coatFresnel_ss);
      coat = coat * coatFresnel_ss;
      result.f = result.f * (1.0f - coat) + c.f * coat;
      result.pdf = result.pdf * (1.0f - coat) + c.pdf * coat;
    }
    return result;
  }

  // ---- sampling ------------------------------------------------------------------------------------------------
  PT_HD BsdfSample sampleMetallic(vec3 wo, vec3 r) const {  // bsdf.metal:513-543
    if (ggx.isSmooth()) {
      const vec3 fresnel_ss = schlick(ctx.albedo, wo.z);
      return {v3(-wo.x, -wo.y, wo.z), fresnel_ss / fabsf(wo.z), v3(0.0f), 1.0f, Sample_Reflected | Sample_Specular};
    }
    const vec3 wm = ggx.sampleVmdf(wo, {r.x, r.y});
    const vec3 wi = reflect(-wo, wm);
    if (wo.z * wi.z < 0.0f) return sample_none();
    const BsdfEval e = evalMetallicWm(wo, wi, wm);
    return {wi, e.f, v3(0.0f), e.pdf, Sample_Reflected | Sample_Glossy};
  }
  PT_HD BsdfSample sampleTransparent(vec3 wo, vec3 r) const {  // bsdf.metal:550-619
    const bool thin = (ctx.flags & PT_MATERIAL_THIN_DIELECTRIC) != 0;
    const float ior = (wo.z < 0.0f && !thin) ? 1.0f / ctx.ior : ctx.ior;
    if (ggx.isSmooth()) {
      const float fresnel_ss = fresnel(fabsf(wo.z), ior);
      vec3 wi, color = v3(1.0f);
      float pdf = fresnel_ss;
      int flags = Sample_Specular;
      if (r.z < fresnel_ss) {
        wi = v3(-wo.x, -wo.y, wo.z);
        flags |= Sample_Reflected;
      } else {
        wi = thin ? -wo : refract(-wo, v3(0.0f, 0.0f, sign(wo.z)), 1.0f / ior);
        if (wi.z == 0.0f) return sample_none();
        pdf = (1.0f - fresnel_ss);
        color = ctx.albedo;
        flags |= Sample_Transmitted;
      }
      return {wi, pdf * color / fabsf(wi.z), v3(0.0f), pdf, flags};
    }
    const vec3 wm = ggx.sampleVmdf(wo, {r.x, r.y});
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ior);
    vec3 wi;
    int flags = Sample_Glossy;
    if (r.z < fresnel_ss) {
      wi = reflect(-wo, wm);
      if (wo.z * wi.z < 0.0f) return sample_none();
      flags |= Sample_Reflected;
    } else if (thin) {
      wi = reflect(-wo, wm) * v3(1.0f, 1.0f, -1.0f);
      flags |= Sample_Transmitted;
    } else {
      wi = refract(-wo, wm * sign(dot(wo, wm)), 1.0f / ior);
      if (wo.z * wi.z >= 0.0f) return sample_none();
      flags |= Sample_Transmitted;
    }
    const BsdfEval e = evalTransparentWm(wo, wi, wm, fresnel_ss, ior);
    return {wi, e.f, v3(0.0f), e.pdf, flags};
  }
  PT_HD BsdfSample sampleOpaque(vec3 wo, vec3 r) const {  // bsdf.metal:626-684
    const float F_avg = avgDielectricFresnelFit(ctx.ior);
    const float blendingFactor = opaqueDielectricFactor(wo, F_avg);
    if (r.z < blendingFactor) {
      if (ggx.isSmooth()) {
        const float fresnel_ss = fresnel(fabsf(wo.z), ctx.ior);
        const vec3 wi = v3(-wo.x, -wo.y, wo.z);
        return {wi, v3(fresnel_ss / fabsf(wi.z)), v3(0.0f), blendingFactor, Sample_Reflected | Sample_Specular};
      }
      const vec3 wm = ggx.sampleVmdf(wo, {r.x, r.y});
      if (length_squared(wm) == 0.0f) return sample_none();
      const vec3 wi = reflect(-wo, wm);
      const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), ctx.ior);
      float dielectricBrdf = fresnel_ss * ggx.singleScatterBRDF(wo, wi, wm);
      if (ms) dielectricBrdf += multiscatter(wo, wi, F_avg);
      return {wi, v3(dielectricBrdf), v3(0.0f), ggx.pdf(wo, wm) * blendingFactor, Sample_Reflected | Sample_Glossy};
    }
    vec3 wi = sampleCosineHemisphere({r.x, r.y});
    if (wo.z < 0.0f) wi = wi * -1.0f;
    const float cDiffuse = diffuseFactor(wo, wi);
    int flags = Sample_Reflected | Sample_Diffuse;
    if (ctx.flags & PT_MATERIAL_EMISSIVE) flags |= Sample_Emitted;
    return {wi, ctx.albedo * cDiffuse, ctx.emission / (1.0f - blendingFactor),
            fabsf(wi.z) / kPi * (1.0f - blendingFactor), flags};
  }
  PT_HD BsdfSample sampleClearcoat(vec3 wo, vec3 r) const {  // bsdf.metal:686-714
    if (ggxCoat.isSmooth()) {
      const float fresnel_ss = fresnel(wo.z, kClearcoatIor);
      const vec3 wi = v3(-wo.x, -wo.y, wo.z);
      return {wi, v3(fresnel_ss / fabsf(wi.z)), v3(0.0f), fresnel_ss, Sample_Reflected | Sample_Specular};
    }
    const vec3 wm = ggxCoat.sampleVmdf(wo, {r.x, r.y});
    const vec3 wi = reflect(-wo, wm);
    if (wo.z * wi.z < 0.0f) return sample_none();
    const float fresnel_ss = fresnel(fabsf(dot(wo, wm)), kClearcoatIor);
    return {wi, v3(fresnel_ss * ggxCoat.singleScatterBRDF(wo, wi, wm)), v3(0.0f), fresnel_ss * ggxCoat.pdf(wo, wm),
            Sample_Reflected | Sample_Glossy};
  }
  PT_HD BsdfSample sample(vec3 wo, vec4 r, vec2 rc) const {  // bsdf.metal:228-252
    const float m = ctx.metallic;
    const float t = ctx.transmission;
    float pClearcoat = ctx.clearcoat;
    if (pClearcoat > 0.0f) {
      const vec3 wmCoat = ggxCoat.isSmooth() ? v3(0.0f, 0.0f, 1.0f) : ggxCoat.sampleVmdf(wo, rc);
      pClearcoat = pClearcoat * fresnel(fabsf(dot(wo, wmCoat)), kClearcoatIor);
    }
    const float pMetallic = pClearcoat + (1.0f - pClearcoat) * m;
    const float pTransparent = pClearcoat + (1.0f - pClearcoat) * (m + (1.0f - m) * t);
    const vec3 rxyz = v3(r.x, r.y, r.z);
    if (r.w < pClearcoat) return sampleClearcoat(wo, rxyz);
    if (r.w < pMetallic) return sampleMetallic(wo, rxyz);
    if (r.w < pTransparent) return sampleTransparent(wo, rxyz);
    return sampleOpaque(wo, rxyz);
  }
};

}  // namespace pt
