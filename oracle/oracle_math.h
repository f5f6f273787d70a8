// TEST INFRASTRUCTURE — part of oracle/: the CPU restatement of teofum/platinum's path-tracing arithmetic.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call this.
// The product (platinum_amd/) never includes, links or executes anything in this directory.
//
// oracle_math.h: the MSL built-ins the shaders use, restated as explicit fp32 operation sequences.
// MSL is compiled with fast-math (CMakeLists.txt:20-24), so the reference's radiance is not
// bit-reproducible across compilers; this file fixes ONE operation order and ONE set of
// transcendental approximations.  DESIGN.md §"Deterministic fp32 contract" states the same contract in
// prose; the HIP kernels implement it independently (platinum_amd/csrc/pt_math.h).
// Build flags required: -ffp-contract=off -fno-fast-math (see oracle/Makefile).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static inline float3 f3(float x, float y, float z) { return {x, y, z}; }
static inline float3 f3(float s) { return {s, s, s}; }
static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
static inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
static inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
static inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
static inline float3& operator/=(float3& a, float s) { a = a / s; return a; }

// dot: (x*x' + y*y') + z*z'   (left to right, no contraction)
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length_squared(float3 v) { return dot(v, v); }
static inline float length(float3 v) { return sqrtf(dot(v, v)); }
// normalize: v * (1 / sqrt(dot(v, v)))   [MSL fast-math uses rsqrt; we fix an IEEE sequence]
static inline float3 normalize(float3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
static inline float length_squared(float2 v) { return v.x * v.x + v.y * v.y; }

static inline float fmin_(float a, float b) { return fminf(a, b); }
static inline float fmax_(float a, float b) { return fmaxf(a, b); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float mix(float a, float b, float t) { return a + (b - a) * t; }
static inline float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
// Metal spec: reflect(I, N) = I - 2 * dot(N, I) * N
static constexpr float PI_F = 3.14159265358979323846f;  // M_PI_F

static inline float3 reflect(float3 I, float3 N) { return I - (2.0f * dot(N, I)) * N; }
// Metal spec: refract(I, N, eta): k = 1 - eta^2 (1 - dot(N,I)^2); k < 0 ? 0 : eta*I - (eta*dot(N,I) + sqrt(k)) * N
static inline float3 refract(float3 I, float3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - (eta * eta) * (1.0f - d * d);
  if (k < 0.0f) return f3(0.0f);
  return eta * I - (eta * d + sqrtf(k)) * N;
}


// ---- deterministic transcendentals (contract shared, in prose, with the HIP kernels) ------------------------

// sincos(x): Cody-Waite reduction by pi/2 with three constants, cephes-style minimax polynomials on
// [-pi/4, pi/4].  |x| < 2^16 * pi/2.  No fma anywhere.
static inline void sincos_det(float x, float* s_out, float* c_out) {
  const float TWO_OVER_PI = 0.63661977236758134308f;
  const float DP1 = 1.5703125f;                 // 8 significant bits: k*DP1 exact for |k| < 2^16
  const float DP2 = 4.837512969970703125e-4f;
  const float DP3 = 7.54978995489188216e-8f;
  float kf = rintf(x * TWO_OVER_PI);
  int k = (int)kf;
  float r = ((x - kf * DP1) - kf * DP2) - kf * DP3;
  float z = r * r;
  float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  float s, c;
  switch (k & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
  }
  *s_out = s;
  *c_out = c;
}
static inline float cos_det(float x) { float s, c; sincos_det(x, &s, &c); return c; }

// atan(x): cephes atanf range reduction (tan(3pi/8), tan(pi/8)) + degree-9 odd polynomial. No fma.
static inline float atan_det(float xx) {
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
  return xx < 0.0f ? -y : y;
}
// atan2(y, x) with the usual quadrant rules; atan2(0, 0) = 0.
static inline float atan2_det(float y, float x) {
  if (x > 0.0f) return atan_det(y / x);
  if (x < 0.0f) return y >= 0.0f ? atan_det(y / x) + PI_F : atan_det(y / x) - PI_F;
  if (y > 0.0f) return 1.5707963267948966f;
  if (y < 0.0f) return -1.5707963267948966f;
  return 0.0f;
}
// acos(x) = atan2(sqrt((1 - x) * (1 + x)), x), x clamped to [-1, 1]
static inline float acos_det(float x) {
  x = fminf(fmaxf(x, -1.0f), 1.0f);
  return atan2_det(sqrtf((1.0f - x) * (1.0f + x)), x);
}

// log2(x), x > 0 finite normal: x = m * 2^e with m in [sqrt(1/2), sqrt(2)); cephes logf polynomial.
static inline float log2_det(float x) {
  uint32_t bits; memcpy(&bits, &x, 4);
  int e = (int)((bits >> 23) & 0xff) - 126;                 // x = m * 2^e, m in [0.5, 1)
  bits = (bits & 0x007fffffu) | 0x3f000000u;
  float m; memcpy(&m, &bits, 4);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m; }  // m in [sqrt(.5), sqrt(2))
  float t = m - 1.0f;
  float z = t * t;
  float y = ((((((((7.0376836292e-2f * t - 1.1514610310e-1f) * t + 1.1676998740e-1f) * t - 1.2420140846e-1f) * t
               + 1.4249322787e-1f) * t - 1.6668057665e-1f) * t + 2.0000714765e-1f) * t - 2.4999993993e-1f) * t
             + 3.3333331174e-1f) * t * z;
  y = y - 0.5f * z;
  float ln_m = t + y;
  return ln_m * 1.44269504088896340736f + (float)e;
}

// exp2(y), |y| < 126: n = rint(y), f = y - n in [-0.5, 0.5]; cephes exp2f polynomial; scale by 2^n.
static inline float exp2_det(float y) {
  float nf = rintf(y);
  int n = (int)nf;
  float f = y - nf;
  float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
              + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
  uint32_t sb = (uint32_t)(n + 127) << 23;
  float scale; memcpy(&scale, &sb, 4);
  return p * scale;
}

// powr(x, y) = exp2(y * log2(x)), x >= 0 (Metal: undefined for x < 0). powr(0, y>0) = 0.
static inline float powr_det(float x, float y) {
  if (x <= 0.0f) return 0.0f;
  return exp2_det(y * log2_det(x));
}

}  // namespace orc
