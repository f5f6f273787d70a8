// pt_math.h — fp32 vector math and deterministic transcendentals for the gfx950 kernels.
//
// Deterministic fp32 contract: every kernel TU is compiled with -ffp-contract=off and no fast-math,
// divide and sqrt are IEEE (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), so each expression below is
// the operation sequence it spells.  The MSL built-ins used by the reference shaders
// (src/renderer_pt/shaders/*.metal: dot, cross, normalize, reflect, refract, mix, saturate, sign, sincos, powr,
// exp2, cos, fmod) are implemented here once for the device; the CPU oracle implements the same contract
// separately.  PT_HD functions are plain C++ so that tests/emu can also compile them for the host to debug
// stage logic without a GPU (test harness only — the library has no CPU path).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline
#else
#define PT_HD inline
#endif

namespace pt {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct alignas(16) vec4 { float x, y, z, w; };

PT_HD vec3 v3(float x, float y, float z) { return {x, y, z}; }
PT_HD vec3 v3(float s) { return {s, s, s}; }
PT_HD vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_HD vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_HD vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_HD vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PT_HD vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }

PT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
PT_HD vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_HD float length_squared(vec3 v) { return dot(v, v); }
PT_HD float length(vec3 v) { return sqrtf(dot(v, v)); }
PT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
PT_HD float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
PT_HD float mix(float a, float b, float t) { return a + (b - a) * t; }
PT_HD float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
PT_HD vec3 reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
PT_HD vec3 refract(vec3 I, vec3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - (eta * eta) * (1.0f - d * d);
  if (k < 0.0f) return v3(0.0f);
  return eta * I - (eta * d + sqrtf(k)) * N;
}

PT_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PT_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInf = __builtin_huge_valf();

// sincos: quadrant k = rint(x * 2/pi); r = ((x - k*DP1) - k*DP2) - k*DP3; degree-7/6 polynomials in r.
PT_HD void sincos_det(float x, float* s_out, float* c_out) {
  float kf = rintf(x * 0.63661977236758134308f);
  int k = (int)kf;
  float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  int q = k & 3;
  float s = (q & 1) ? cp : sp;
  float c = (q & 1) ? sp : cp;
  if (q == 2 || q == 3) s = -s;
  if (q == 1 || q == 2) c = -c;
  *s_out = s;
  *c_out = c;
}
PT_HD float cos_det(float x) { float s, c; sincos_det(x, &s, &c); return c; }

// atan: cephes atanf reduction (tan(3pi/8), tan(pi/8)) + degree-9 odd polynomial; atan2 by quadrant; acos through atan2.
PT_HD float atan_det(float xx) {
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  const float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
  return xx < 0.0f ? -y : y;
}
PT_HD float atan2_det(float y, float x) {
  if (x > 0.0f) return atan_det(y / x);
  if (x < 0.0f) return y >= 0.0f ? atan_det(y / x) + kPi : atan_det(y / x) - kPi;
  if (y > 0.0f) return 1.5707963267948966f;
  if (y < 0.0f) return -1.5707963267948966f;
  return 0.0f;
}
PT_HD float acos_det(float x) {
  x = fminf(fmaxf(x, -1.0f), 1.0f);
  return atan2_det(sqrtf((1.0f - x) * (1.0f + x)), x);
}

PT_HD float log2_det(float x) {
  uint32_t bits = f2u(x);
  int e = (int)((bits >> 23) & 0xff) - 126;
  float m = u2f((bits & 0x007fffffu) | 0x3f000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }
  float t = m - 1.0f;
  float z = t * t;
  float y = ((((((((7.0376836292e-2f * t - 1.1514610310e-1f) * t + 1.1676998740e-1f) * t - 1.2420140846e-1f) * t
               + 1.4249322787e-1f) * t - 1.6668057665e-1f) * t + 2.0000714765e-1f) * t - 2.4999993993e-1f) * t
             + 3.3333331174e-1f) * t * z;
  y = y - 0.5f * z;
  float ln_m = t + y;
  return ln_m * 1.44269504088896340736f + (float)e;
}
PT_HD float exp2_det(float y) {
  float nf = rintf(y);
  int n = (int)nf;
  float f = y - nf;
  float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
              + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
  return p * u2f((uint32_t)(n + 127) << 23);
}
PT_HD float powr_det(float x, float y) { return x <= 0.0f ? 0.0f : exp2_det(y * log2_det(x)); }

}  // namespace pt
