// pt_post.h — SURVEY §8f row N2: the reference's post-process chain and tonemap, fused per pixel.
//
// Reference: five full-screen fragment passes over RGBA32F ping-pong targets followed by the tonemap pass into RGBA8
// (renderer_pt.cpp:184-194, pass order :343-353 = exposure, chromaticAberration, contrastSaturation, toneCurve,
// vignette, tonemap; shaders/postprocess.metal:415-600; options core/postprocessing.hpp:29-260).
// Every pass samples its source at the pixel centre with a linear filter — which returns the texel itself — except
// chromatic aberration, which reads the (exposure-scaled) image at three shifted positions.  One kernel therefore
// reproduces the chain: exposure is applied on the fly to each tap, the rest is pointwise.
// Transcendentals are the deterministic ones of pt_math.h (log2/exp2/powr); MSL uses fast-math intrinsics there, so
// the last bits differ from Metal by construction (float tolerance).
#pragma once
#include "pt_device.h"

namespace pt {

PT_HD float pp_luma(vec3 c) { return (c.x * 0.2126f + c.y * 0.7152f) + c.z * 0.0722f; }  // lw, postprocess.metal:19
PT_HD float pp_rgbSum(vec3 c) { return c.x + c.y + c.z; }
PT_HD float pp_rgbAvg(vec3 c) { return (c.x + c.y + c.z) / 3.0f; }
PT_HD float pp_rgbMax(vec3 c) { return fmaxf(fmaxf(c.x, c.y), c.z); }
PT_HD float pp_rgbMin(vec3 c) { return fminf(fminf(c.x, c.y), c.z); }
PT_HD float pp_invLerp(float x, float s, float e) { return saturate((x - s) / (e - s)); }
PT_HD float pp_smoothstep(float e0, float e1, float x) {
  const float t = saturate((x - e0) / (e1 - e0));
  return t * t * (3.0f - 2.0f * t);
}
PT_HD vec3 pp_mix3(vec3 a, vec3 b, float t) { return a + (b - a) * t; }
PT_HD float pp_log2(float x) { return x > 0.0f ? log2_det(x) : -kInf; }
PT_HD vec3 pp_log2v(vec3 v) { return {pp_log2(v.x), pp_log2(v.y), pp_log2(v.z)}; }
PT_HD vec3 pp_exp2v(vec3 v) { return {exp2_det(v.x), exp2_det(v.y), exp2_det(v.z)}; }
PT_HD vec3 pp_powrv(vec3 v, float p) { return {powr_det(v.x, p), powr_det(v.y, p), powr_det(v.z, p)}; }
PT_HD vec3 pp_saturate3(vec3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
PT_HD vec3 pp_clamp3(vec3 v, float lo, float hi) { return {fminf(fmaxf(v.x, lo), hi), fminf(fmaxf(v.y, lo), hi), fminf(fmaxf(v.z, lo), hi)}; }
PT_HD float pp_exp2s(float x) { return x < -125.0f ? 0.0f : (x > 125.0f ? kInf : exp2_det(x)); }  // range guard around exp2_det

struct PPMat3 { vec3 c0, c1, c2; };  // columns, like MSL float3x3
PT_HD vec3 pp_mul(const PPMat3& m, vec3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }            // M * v
PT_HD vec3 pp_vmul(vec3 v, const PPMat3& m) { return {dot(v, m.c0), dot(v, m.c1), dot(v, m.c2)}; }       // v * M
PT_HD float pp_m(const PPMat3& m, int c, int r) { const vec3& col = c == 0 ? m.c0 : (c == 1 ? m.c1 : m.c2); return r == 0 ? col.x : (r == 1 ? col.y : col.z); }
PT_HD PPMat3 pp_inverse(const PPMat3& m) {  // postprocess.metal:42-63, m[c][r]
  const float a = pp_m(m, 1, 1) * pp_m(m, 2, 2) - pp_m(m, 2, 1) * pp_m(m, 1, 2);
  const float b = pp_m(m, 1, 2) * pp_m(m, 2, 0) - pp_m(m, 1, 0) * pp_m(m, 2, 2);
  const float c = pp_m(m, 1, 0) * pp_m(m, 2, 1) - pp_m(m, 1, 1) * pp_m(m, 2, 0);
  const float det = (pp_m(m, 0, 0) * a + pp_m(m, 0, 1) * b) + pp_m(m, 0, 2) * c;
  const float invdet = 1.0f / det;
  PPMat3 inv;
  inv.c0 = v3(a * invdet, (pp_m(m, 0, 2) * pp_m(m, 2, 1) - pp_m(m, 0, 1) * pp_m(m, 2, 2)) * invdet,
              (pp_m(m, 0, 1) * pp_m(m, 1, 2) - pp_m(m, 0, 2) * pp_m(m, 1, 1)) * invdet);
  inv.c1 = v3(b * invdet, (pp_m(m, 0, 0) * pp_m(m, 2, 2) - pp_m(m, 0, 2) * pp_m(m, 2, 0)) * invdet,
              (pp_m(m, 1, 0) * pp_m(m, 0, 2) - pp_m(m, 0, 0) * pp_m(m, 1, 2)) * invdet);
  inv.c2 = v3(c * invdet, (pp_m(m, 2, 0) * pp_m(m, 0, 1) - pp_m(m, 0, 0) * pp_m(m, 2, 1)) * invdet,
              (pp_m(m, 0, 0) * pp_m(m, 1, 1) - pp_m(m, 1, 0) * pp_m(m, 0, 1)) * invdet);
  return inv;
}

// ---- AgX (postprocess.metal:89-147) ----------------------------------------------------------------------------------
PT_HD vec3 agx_contrast(vec3 x) {
  const vec3 x2 = x * x, x4 = x2 * x2;
  return (((((15.5f * x4 * x2 - 40.14f * x4 * x) + 31.96f * x4) - 6.868f * x2 * x) + 0.4298f * x2) + 0.1191f * x) - v3(0.00232f);
}
PT_HD vec3 agx_start(vec3 val) {
  const PPMat3 M = {v3(0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f),
                    v3(0.0784335999999992f, 0.878468636469772f, 0.0784336f),
                    v3(0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f)};
  const float minEv = -12.47393f, maxEv = 4.026069f;
  val = pp_mul(M, val);
  val = pp_clamp3(pp_log2v(val), minEv, maxEv);
  val = (val - v3(minEv)) / (maxEv - minEv);
  return agx_contrast(val);
}
PT_HD vec3 agx_end(vec3 val) {
  const PPMat3 Mi = {v3(1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f),
                     v3(-0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f),
                     v3(-0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f)};
  return pp_saturate3(pp_mul(Mi, val));
}
PT_HD vec3 agx_apply(vec3 val, const pt_tonemap_options& o) {
  val = agx_start(val);
  const float luma = pp_luma(val);  // applyLook, :131-136
  vec3 t = val * v3(o.agx_slope[0], o.agx_slope[1], o.agx_slope[2]) + v3(o.agx_offset[0], o.agx_offset[1], o.agx_offset[2]);
  t = v3(powr_det(t.x, o.agx_power[0]), powr_det(t.y, o.agx_power[1]), powr_det(t.z, o.agx_power[2]));
  val = pp_mix3(v3(luma), t, o.agx_saturation);
  return agx_end(val);
}

// ---- Khronos PBR neutral (postprocess.metal:155-175) -------------------------------------------------------------------
PT_HD vec3 khronos_apply(vec3 val, const pt_tonemap_options& o) {
  const float compressionStart = o.khr_compression_start - 0.04f;
  const float x = fminf(val.x, fminf(val.y, val.z));
  const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
  val = val - v3(offset);
  const float peak = fmaxf(val.x, fmaxf(val.y, val.z));
  if (peak < compressionStart) return val;
  const float d = 1.0f - compressionStart;
  const float newPeak = 1.0f - d * d / (peak + d - compressionStart);
  val = val * (newPeak / peak);
  const float g = 1.0f - 1.0f / (o.khr_desaturation * (peak - newPeak) + 1.0f);
  return pp_mix3(val, v3(newPeak), g);
}

// ---- flim (postprocess.metal:181-413) ------------------------------------------------------------------------------------
PT_HD float flim_wrap(float x, float s, float e) { return s + fmodf(x - s, e - s); }
PT_HD vec3 flim_rgbUniformOffset(vec3 color, float blackPoint, float whitePoint) {
  const float mono = pp_rgbAvg(color);
  const float mono2 = pp_invLerp(mono, blackPoint / 1000.0f, 1.0f - whitePoint / 1000.0f);
  return color * (mono2 / mono);
}
PT_HD vec3 flim_rgbToHsv(vec3 rgb) {
  const float cmax = pp_rgbMax(rgb), cmin = pp_rgbMin(rgb), cdelta = cmax - cmin;
  float h = 0.0f, s, v = cmax;
  if (cmax != 0.0f) s = cdelta / cmax; else s = 0.0f;
  if (s != 0.0f) {
    const vec3 c = (v3(cmax) - rgb) / cdelta;
    if (rgb.x == cmax) h = c.z - c.y;
    else if (rgb.y == cmax) h = 2.0f + c.x - c.z;
    else h = 4.0f + c.y - c.x;
    h = h / 6.0f;
    if (h < 0.0f) h += 1.0f;
  }
  return v3(h, s, v);
}
PT_HD vec3 flim_hsvToRgb(vec3 hsv) {
  float h = hsv.x;
  const float s = hsv.y, v = hsv.z;
  if (s == 0.0f) return v3(v);
  if (h == 1.0f) h = 0.0f;
  h = h * 6.0f;
  const int i = (int)floorf(h);
  const float f = h - (float)i;
  const float p = v * (1.0f - s), q = v * (1.0f - (s * f)), t = v * (1.0f - (s * (1.0f - f)));
  switch (i) {
    case 0: return v3(v, t, p);
    case 1: return v3(q, v, p);
    case 2: return v3(p, v, t);
    case 3: return v3(p, q, v);
    case 4: return v3(t, p, v);
    default: return v3(v, p, q);
  }
}
PT_HD vec3 flim_hueSat(vec3 color, float hue, float sat, float value) {
  vec3 hsv = flim_rgbToHsv(color);
  const float hh = hsv.x + hue + 0.5f;
  hsv.x = hh - floorf(hh);  // fract
  hsv.y = saturate(hsv.y * sat);
  hsv.z = hsv.z * value;
  return flim_hsvToRgb(hsv);
}
PT_HD vec3 flim_gamutRow(float primaryHue, float scale, float rotate, float mul) {
  vec3 r = flim_hsvToRgb(v3(flim_wrap(primaryHue + (rotate / 360.0f), 0.0f, 1.0f), 1.0f / scale, 1.0f));
  r = r / pp_rgbSum(r);
  return r * mul;
}
PT_HD float flim_superSigmoid(float x, vec2 toe, vec2 shoulder) {
  x = saturate(x);
  toe = {saturate(toe.x), saturate(toe.y)};
  shoulder = {saturate(shoulder.x), saturate(shoulder.y)};
  const float slope = (shoulder.y - toe.y) / (shoulder.x - toe.x);
  if (x < toe.x) return toe.y * powr_det(x / toe.x, slope * toe.x / toe.y);
  if (x < shoulder.x) return slope * x + toe.y - (slope * toe.x);
  const float shoulderPow = -slope / ((shoulder.x - 1.0f) / powr_det(1.0f - shoulder.x, 2.0f) * (1.0f - shoulder.y));
  return (1.0f - powr_det(1.0f - (x - shoulder.x) / (1.0f - shoulder.x), shoulderPow)) * (1.0f - shoulder.y) + shoulder.y;
}
PT_HD float flim_dyeMixFactor(float mono, float maxDensity, const pt_tonemap_options& o) {
  const float offset = pp_exp2s(o.flim_sigmoid_log2_min);
  float fac = pp_invLerp(pp_log2(mono + offset), o.flim_sigmoid_log2_min, o.flim_sigmoid_log2_max);
  fac = flim_superSigmoid(fac, {o.flim_sigmoid_toe[0], o.flim_sigmoid_toe[1]}, {o.flim_sigmoid_shoulder[0], o.flim_sigmoid_shoulder[1]});
  fac = fac * maxDensity;
  fac = pp_exp2s(-fac);
  return saturate(fac);
}
PT_HD vec3 flim_colorLayer(vec3 color, vec3 sensitivityTone, vec3 dyeTone, float maxDensity, const pt_tonemap_options& o) {
  sensitivityTone = sensitivityTone / pp_rgbSum(sensitivityTone);
  dyeTone = dyeTone / pp_rgbMax(dyeTone);
  const float mono = dot(color, sensitivityTone);
  const float mixFactor = flim_dyeMixFactor(mono, maxDensity, o);
  return pp_mix3(dyeTone, v3(1.0f), mixFactor);
}
PT_HD vec3 flim_develop(vec3 color, float exposure, float maxDensity, const pt_tonemap_options& o) {
  color = color * pp_exp2s(exposure);
  vec3 result = flim_colorLayer(color, v3(0, 0, 1), v3(1, 1, 0), maxDensity, o);
  result = result * flim_colorLayer(color, v3(0, 1, 0), v3(1, 0, 1), maxDensity, o);
  result = result * flim_colorLayer(color, v3(1, 0, 0), v3(0, 1, 1), maxDensity, o);
  return result;
}
PT_HD vec3 flim_negativeAndPrint(vec3 color, vec3 backlight, const pt_tonemap_options& o) {
  color = flim_develop(color, o.flim_negative_exposure, o.flim_negative_density, o);
  color = color * backlight;
  return flim_develop(color, o.flim_print_exposure, o.flim_print_density, o);
}
PT_HD vec3 flim_apply(vec3 val, const pt_tonemap_options& o) {
  val = val * pp_exp2s(o.flim_pre_exposure);
  PPMat3 ext;
  ext.c0 = flim_gamutRow(0.0f / 3.0f, o.flim_extended_gamut_scale[0], o.flim_extended_gamut_rotation[0], o.flim_extended_gamut_mul[0]);
  ext.c1 = flim_gamutRow(1.0f / 3.0f, o.flim_extended_gamut_scale[1], o.flim_extended_gamut_rotation[1], o.flim_extended_gamut_mul[1]);
  ext.c2 = flim_gamutRow(2.0f / 3.0f, o.flim_extended_gamut_scale[2], o.flim_extended_gamut_rotation[2], o.flim_extended_gamut_mul[2]);
  const PPMat3 extInv = pp_inverse(ext);
  const vec3 backlight = pp_vmul(v3(o.flim_print_backlight[0], o.flim_print_backlight[1], o.flim_print_backlight[2]), ext);
  const vec3 whiteCap = flim_negativeAndPrint(v3(1e7f), backlight, o);
  const vec3 pre = v3(o.flim_pre_formation_filter[0], o.flim_pre_formation_filter[1], o.flim_pre_formation_filter[2]);
  val = pp_mix3(val, val * pre, o.flim_pre_formation_filter_strength);
  val = pp_vmul(val, ext);
  val = flim_negativeAndPrint(val, backlight, o);
  val = pp_vmul(val, extInv);
  val = v3(fmaxf(val.x, 0.0f), fmaxf(val.y, 0.0f), fmaxf(val.z, 0.0f));
  val = val / whiteCap;
  if (o.flim_auto_black_point) {
    vec3 blackCap = flim_negativeAndPrint(v3(0.0f), backlight, o);
    blackCap = blackCap / whiteCap;
    val = flim_rgbUniformOffset(val, pp_rgbAvg(blackCap) * 1000.0f, 0.0f);
  } else {
    val = flim_rgbUniformOffset(val, o.flim_black_point, 0.0f);
  }
  const vec3 post = v3(o.flim_post_formation_filter[0], o.flim_post_formation_filter[1], o.flim_post_formation_filter[2]);
  val = pp_mix3(val, val * post, o.flim_post_formation_filter_strength);
  val = pp_saturate3(val);
  const float mono = pp_rgbAvg(val);
  const float mixFactor = (mono < 0.5f) ? pp_invLerp(mono, 0.05f, 0.5f) : pp_invLerp(mono, 0.95f, 0.5f);
  val = pp_mix3(val, flim_hueSat(val, 0.5f, o.flim_midtone_saturation, 1.0f), mixFactor);
  return pp_saturate3(val);
}

// ---- the chain -------------------------------------------------------------------------------------------------------------
PT_HD vec2 pp_aspectUv(vec2 uv, float aspect) {  // :481-486
  if (aspect > 1.0f) uv.y = (uv.y - 0.5f) / aspect + 0.5f; else uv.x = (uv.x - 0.5f) * aspect + 0.5f;
  return uv;
}
PT_HD vec2 pp_aspectUvInv(vec2 uv, float aspect) {  // :488-493
  if (aspect > 1.0f) uv.y = (uv.y - 0.5f) * aspect + 0.5f; else uv.x = (uv.x - 0.5f) / aspect + 0.5f;
  return uv;
}

// One channel of the exposure-pass image, sampled bilinearly (clamp to edge) at normalised uv.
PT_HD float pp_sample_exposed(const vec4* __restrict__ acc, uint32_t W, uint32_t H, float expScale, vec2 uv, int ch) {
  const float fx = uv.x * (float)W - 0.5f, fy = uv.y * (float)H - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx = fx - x0f, wy = fy - y0f;
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const int Wi = (int)W - 1, Hi = (int)H - 1;
  x0 = x0 < 0 ? 0 : (x0 > Wi ? Wi : x0); x1 = x1 < 0 ? 0 : (x1 > Wi ? Wi : x1);
  y0 = y0 < 0 ? 0 : (y0 > Hi ? Hi : y0); y1 = y1 < 0 ? 0 : (y1 > Hi ? Hi : y1);
  auto tap = [&](int x, int y) {
    const vec4 a = acc[(size_t)y * W + x];
    return (ch == 0 ? a.x : (ch == 1 ? a.y : a.z)) * expScale;
  };
  const float t00 = tap(x0, y0), t01 = tap(x1, y0), t10 = tap(x0, y1), t11 = tap(x1, y1);
  const float a = t00 + (t01 - t00) * wx, b = t10 + (t11 - t10) * wx;
  return a + (b - a) * wy;
}

struct PostConstants {  // precomputed on the host once per resolve
  pt_post_options post;
  pt_tonemap_options tm;
  PPMat3 odt;  // color::transform(working, output), renderer_pt.cpp:190-191
};

// Returns the display-referred colour in [0,1] (before the RGBA8 quantisation) of pixel (px, py).
PT_HD vec3 postprocess_pixel(const vec4* __restrict__ acc, uint32_t W, uint32_t H, uint32_t px, uint32_t py, const PostConstants& pc) {
  const pt_post_options& o = pc.post;
  const vec2 uv = {((float)px + 0.5f) / (float)W, ((float)py + 0.5f) / (float)H};
  const float expScale = pp_exp2s(o.exposure);  // exposure pass, :425-437
  const vec4 a = acc[(size_t)py * W + px];
  vec3 color = v3(a.x, a.y, a.z) * expScale;

  if (o.ca_amount != 0.0f) {  // chromaticAberration, :526-551
    const float aspect = (float)W / (float)H;
    const vec2 uvM = pp_aspectUv(uv, aspect);
    const float amount = o.ca_amount * 0.005f * 0.01f;
    const float sr = 1.0f + amount, sg = 1.0f - amount * o.ca_green_shift * 0.01f, sb = 1.0f - amount;
    const vec2 uvR = pp_aspectUvInv({(uvM.x - 0.5f) * sr + 0.5f, (uvM.y - 0.5f) * sr + 0.5f}, aspect);
    const vec2 uvG = pp_aspectUvInv({(uvM.x - 0.5f) * sg + 0.5f, (uvM.y - 0.5f) * sg + 0.5f}, aspect);
    const vec2 uvB = pp_aspectUvInv({(uvM.x - 0.5f) * sb + 0.5f, (uvM.y - 0.5f) * sb + 0.5f}, aspect);
    color.x = pp_sample_exposed(acc, W, H, expScale, uvR, 0);
    color.y = pp_sample_exposed(acc, W, H, expScale, uvG, 1);
    color.z = pp_sample_exposed(acc, W, H, expScale, uvB, 2);
  }
  {  // contrastSaturation, :439-463
    const float eps = 1e-6f;
    const vec3 logColor = pp_log2v(color + v3(eps));
    const vec3 adj = pp_mix3(v3(0.18f), logColor, 1.0f + o.contrast * 0.01f);
    const vec3 e = pp_exp2v(adj) - v3(eps);
    color = v3(fmaxf(0.0f, e.x), fmaxf(0.0f, e.y), fmaxf(0.0f, e.z));
    const vec3 gray = v3(pp_luma(color));
    color = pp_mix3(gray, color, 1.0f + o.saturation * 0.01f);
  }
  {  // toneCurve, :465-479
    const float luma = pp_luma(color);
    const float blacks = pp_smoothstep(0.04f, 0.0f, luma), shadows = pp_smoothstep(0.18f, 0.0f, luma);
    const float highlights = pp_smoothstep(0.18f, 1.0f, luma), whites = pp_smoothstep(0.75f, 1.0f, luma);
    color = color * pp_exp2s(0.01f * o.blacks * blacks);
    color = color * pp_exp2s(0.01f * o.shadows * shadows);
    color = color * pp_exp2s(0.01f * o.highlights * highlights);
    color = color * pp_exp2s(0.01f * o.whites * whites);
  }
  {  // vignette, :495-524
    float aspect = (float)W / (float)H;
    aspect = mix(1.0f, aspect, o.vig_roundness * 0.01f);
    const vec2 uvM = pp_aspectUv(uv, aspect);
    const float cornerToCenter = sqrtf(0.5f * 0.5f + 0.5f * 0.5f);
    const float dx = uvM.x - 0.5f, dy = uvM.y - 0.5f;
    const float distanceNorm = sqrtf(dx * dx + dy * dy) / cornerToCenter;
    const float end = 1.0f - o.vig_midpoint * 0.01f;
    const float start = end * (1.0f - o.vig_feather * 0.01f);
    const float power = o.vig_power * 0.05f;
    const float d = pp_invLerp(distanceNorm, start, end);
    const float vignetting = (d == 0.0f ? 0.0f : powr_det(d, power)) * pp_smoothstep(start, end, distanceNorm);
    color = color * pp_exp2s(o.vig_amount * vignetting);
  }
  // tonemap, :553-600
  const pt_tonemap_options& t = pc.tm;
  switch (t.tonemapper) {
    case 1: color = pp_powrv(agx_apply(color, t), 2.2f); break;  // "Linearize AgX output"
    case 2: color = khronos_apply(color, t); break;
    case 3: color = flim_apply(color, t); break;
    default: break;
  }
  vec3 liftColor = v3(t.shadow_color[0], t.shadow_color[1], t.shadow_color[2]);
  liftColor = liftColor - v3(pp_rgbAvg(liftColor));
  vec3 gammaColor = v3(t.midtone_color[0], t.midtone_color[1], t.midtone_color[2]);
  gammaColor = gammaColor - v3(pp_rgbAvg(gammaColor));
  vec3 gainColor = v3(t.highlight_color[0], t.highlight_color[1], t.highlight_color[2]);
  gainColor = gainColor - v3(pp_rgbAvg(gainColor));
  const vec3 lift = liftColor + v3(t.shadow_offset * 0.01f);
  const vec3 gain = (v3(1.0f) + gainColor) + v3(t.highlight_offset * 0.01f);
  const vec3 midGray = (v3(0.5f) + gammaColor) + v3(t.midtone_offset * 0.01f);
  const vec3 num = pp_log2v((v3(0.5f) - lift) / (gain - lift));  // log10(a) / log10(b) == log2(a) / log2(b)
  const vec3 den = pp_log2v(midGray);
  const vec3 gamma = num / den;
  const vec3 tt = pp_saturate3(v3(powr_det(color.x, 1.0f / gamma.x), powr_det(color.y, 1.0f / gamma.y), powr_det(color.z, 1.0f / gamma.z)));
  color = lift + (gain - lift) * tt;  // mix(lift, gain, t)
  color = pp_mul(pc.odt, color);
  auto srgb = [](float c) { return c < 0.0031308f ? 12.92f * c : 1.055f * powr_det(c, 1.0f / 2.4f) - 0.055f; };  // :29-36
  return v3(srgb(color.x), srgb(color.y), srgb(color.z));
}

// RGBA8Unorm store: clamp, scale by 255, round to nearest
PT_HD uint32_t pp_pack_rgba8(vec3 c) {
  auto q = [](float v) -> uint32_t {
    if (!(v > 0.0f)) return 0u;  // also NaN
    if (v >= 1.0f) return 255u;
    return (uint32_t)(v * 255.0f + 0.5f);
  };
  return q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (255u << 24);
}

}  // namespace pt
