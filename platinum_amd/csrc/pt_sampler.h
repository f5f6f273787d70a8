// pt_sampler.h — the reference's sample streams on the device.
//   pcg4d                     samplers.metal:16-23
//   HaltonSampler             samplers.metal:154-184 (ctor, sample1d, sample2d, halton)
//   sampleDisk / Polar / CosineHemisphere / TriUniform   samplers.metal:200-238
// The radical inverse keeps the reference's float sequence (f *= 1/b; r += f * digit) exactly; only the integer
// i / b and i % b are strength-reduced: one 32-bit multiply-high per chunk of digits (HaltonEntry) + exact fp32 digit
// splitting, identical digits for every 32-bit i.
#pragma once
#include "pt_device.h"

namespace pt {

constexpr float kOneMinusEpsilon = 0x1.fffffep-1;  // defs.metal:22

PT_HD uint32_t pcg4d_x(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  x = x * 1664525u + 1013904223u;
  y = y * 1664525u + 1013904223u;
  z = z * 1664525u + 1013904223u;
  w = w * 1664525u + 1013904223u;
  x += y * w; y += z * x; z += x * y; w += y * z;
  x ^= x >> 16u; y ^= y >> 16u; z ^= z >> 16u; w ^= w >> 16u;
  x += y * w;  // only .x is consumed by HaltonSampler (samplers.metal:155)
  return x;
}

PT_HD uint32_t mulhi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
// a * b for a, b < 2^24 with a product < 2^32 (v_mul_u32_u24: full rate, where v_mul_lo_u32 is quarter rate)
PT_HD uint32_t mul_u24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}

PT_HD uint32_t halton_offset(uint32_t px, uint32_t py, uint32_t sample) { return pcg4d_x(px, py, sample, px + py); }

// The table entry of one prime (host side: renderer.hip builds the 620-entry table once per renderer).
inline HaltonEntry make_halton_entry(uint32_t prime) {
  uint32_t digits = 1;
  uint64_t chunk = prime;
  while (chunk * prime < (1ull << 22)) { chunk *= prime; digits++; }  // chunk < 2^22: see HaltonEntry
  uint32_t l = 0;
  while ((1ull << l) < chunk) l++;  // l = ceil(log2 chunk)
  const uint64_t magic = ((1ull << 32) * ((1ull << l) - chunk)) / chunk + 1;  // < 2^32
  const float inv = 1.0f / (float)prime;
  return {(uint32_t)chunk, (uint32_t)magic, l - 1, inv, (float)prime, digits, prime, 0.5f * inv};
}

// Where the per-dimension entries live: the HBM table (all 620 dimensions) or a window [base, base + count) of it that a
// block staged in LDS (k_shade: the dimensions one bounce can touch); dimensions outside the window fall back to HBM.
struct HaltonTab {
  const HaltonEntry* global;
  const HaltonEntry* lds;  // nullptr: no staged window
  uint32_t base, count;
};
PT_HD HaltonTab halton_table(const HaltonEntry* global) { return {global, nullptr, 0u, 0u}; }
PT_HD HaltonEntry halton_entry(const HaltonTab& t, uint32_t d) {
  if (t.lds != nullptr && d - t.base < t.count) return t.lds[d - t.base];
  return ldg(&t.global[d]);
}

// The `digits` base-prime digits of rem < 2^22 (leading zeros included), least significant first, through the reference's
// float sequence f *= 1/b; r += f * digit (samplers.metal:172-180).
//   rem / prime: floor(fma(rem, inv, 0.5 inv)) is exact — (rem + 0.5) / prime is at least 0.5 / prime away from an integer,
//   and the two roundings (inv, the fma) move it by less than ((rem + 0.5) / prime) * 2^-23 < 0.5 / prime.
//   digit = rem - qf * prime: every term is an integer below 2^22, so the explicit fma is exact (it is not a contraction of
//   reference arithmetic: the reference computes i % b in integers).
// Zero digits above the index's leading digit add f * 0 = 0 to r, exactly like not visiting them.
// TOP: `rem` holds the index's LEADING digits (the last call of a draw): the loop ends with them — the reference's `while (i > 0)` — instead of
// running all `digits` positions (r6: a top part is below 2^32 / chunk and mostly has fewer; 12 % of the digit steps of a draw on average).
template <bool TOP = false>
PT_HD void halton_digits(const HaltonEntry& e, float rem, float& f, float& r) {
  if (e.digits == 1) {
    f = f * e.inv;
    r = r + f * rem;
    return;
  }
  for (uint32_t j = 0; j < e.digits; j++) {
    const float qf = floorf(__builtin_fmaf(rem, e.inv, e.hinv));
    const float digit = __builtin_fmaf(-qf, e.primef, rem);
    f = f * e.inv;
    r = r + f * digit;
    rem = qf;
#ifndef PT_HALTON_FULL_TOP   // (A/B switch: the r1-r5 form)
    if (TOP && !(rem > 0.0f)) break;
#endif
  }
}

PT_HD float halton(const HaltonTab& tab, uint32_t i, uint32_t d) {
  const HaltonEntry e = halton_entry(tab, d);
  float f = 1.0f;
  float r = 0.0f;
  for (;;) {
    // q = i / chunk (HaltonEntry); rem = i % chunk holds `digits` base-prime digits
    const uint32_t t = mulhi_u32(e.magic, i);
    const uint32_t q = (t + ((i - t) >> 1)) >> e.shift;
    halton_digits(e, (float)(i - mul_u24(q, e.chunk)), f, r);  // the remainder is below 2^22: exact
    i = q;
    if (q < e.chunk) break;  // q is its own remainder: no further division
  }
  if (i > 0) halton_digits<true>(e, (float)i, f, r);
  return fminf(r, kOneMinusEpsilon);
}

// A sampler cursor: (offset, dim) live in the path state between kernels.
struct Halton {
  HaltonTab tab;
  uint32_t offset;
  uint32_t dim;
  PT_HD float sample1d() { return halton(tab, offset, dim++); }
  PT_HD vec2 sample2d() {
    float x = halton(tab, offset, dim++);
    float y = halton(tab, offset, dim++);
    return {x, y};
  }
};

PT_HD vec2 sampleDisk(vec2 u) {
  const float r = sqrtf(u.x);
  const float theta = 2.0f * kPi * u.y;
  float s, c;
  sincos_det(theta, &s, &c);
  return {r * c, r * s};
}
PT_HD vec2 sampleDiskPolar(vec2 u) { return {sqrtf(u.x), 2.0f * kPi * u.y}; }
PT_HD vec3 sampleCosineHemisphere(vec2 u) {
  const float phi = u.x * 2.0f * kPi;
  const float sinTheta = sqrtf(u.y);
  const float cosTheta = sqrtf(1.0f - u.y);
  float sinPhi, cosPhi;
  sincos_det(phi, &sinPhi, &cosPhi);
  return {cosPhi * sinTheta, sinPhi * sinTheta, cosTheta};
}
PT_HD vec2 sampleTriUniform(vec2 u) {
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

}  // namespace pt
