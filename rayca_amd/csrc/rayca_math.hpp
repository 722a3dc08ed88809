// rayca_math.hpp -- host + device arithmetic contract of the MI355X path-tracing core.
//
// Restates the f32 semantics of rayca-math that decide hit/miss and pixel values:
//   * Vec3 / Point3 are four f32 lanes with w = 0 / w = 1   (rayca-math/src/vec3.rs:25-29,68-72;
//     point3.rs:12-29)
//   * dot = ordered left-to-right lane sum, no FMA          (vec3.rs:240-244; core::simd reduce_sum)
//   * cross = rounded products, then subtraction             (vec3.rs:134-142)
//   * rotate = 2(u.v)u + (s^2 - u.u)v + 2s(u x v)            (vec3.rs:148-159, point3.rs:65-76)
//   * Point3::scale is the one true fused multiply-add       (point3.rs:59-63)
//   * zero-safe reciprocal                                   (vec3.rs:195-216)
//   * normalize only when len > EPS = 2^-10                  (vec3.rs:183-188, lib.rs:33)
//   * Color: `+` multiplies the right operand by its alpha   (color/mod.rs:239-286)
// The translation unit MUST be compiled with -ffp-contract=off (both host and gfx950 passes): any
// contraction or re-association flips sign tests at silhouettes.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#include "libm_exact.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RC_FN __host__ __device__ __forceinline__
#else
#define RC_FN inline
#endif

namespace rayca {

constexpr float kEps = FLT_EPSILON * 8192.0f;  // rayca-math/src/lib.rs:33
constexpr float kRayBias = 1e-4f;              // rayca-math/src/ray.rs:57
constexpr float kPi = 3.14159274101257324219f;
constexpr float kFrac1Pi = 0.318309873342514038086f;
constexpr float kFrac2Pi = 0.636619746685028076172f;

struct F4 {
  float x, y, z, w;
};
struct F2 {
  float x, y;
};
struct Color {
  float r, g, b, a;
};
struct Trs {
  F4 translation, rotation, scale;
};
struct Mat3 {
  float m[3][3];
};

RC_FN F4 f4(float x, float y, float z, float w) { return F4{x, y, z, w}; }
RC_FN F4 vec3(float x, float y, float z) { return F4{x, y, z, 0.0f}; }
RC_FN F4 point3(float x, float y, float z) { return F4{x, y, z, 1.0f}; }
RC_FN F4 as_vec(F4 s) {  // Vec3::simd: w := 0
  s.w = 0.0f;
  return s;
}
RC_FN F4 operator+(F4 a, F4 b) { return F4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
RC_FN F4 operator-(F4 a, F4 b) { return F4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
RC_FN F4 operator*(F4 a, F4 b) { return F4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
RC_FN F4 operator*(F4 a, float f) { return F4{a.x * f, a.y * f, a.z * f, a.w * f}; }
RC_FN F4 operator*(float f, F4 a) { return F4{a.x * f, a.y * f, a.z * f, a.w * f}; }
RC_FN F4 operator/(F4 a, float f) { return F4{a.x / f, a.y / f, a.z / f, a.w / f}; }
RC_FN F4 operator-(F4 a) { return F4{-a.x, -a.y, -a.z, -a.w}; }

// ordered f32x4::reduce_sum (seed -0.0)
RC_FN float hsum(F4 a) { return (((-0.0f + a.x) + a.y) + a.z) + a.w; }
RC_FN float dot(F4 a, F4 b) { return hsum(a * b); }
RC_FN F4 cross(F4 a, F4 b) {
  const F4 t0{a.y, a.z, a.x, a.w};
  const F4 t1{b.z, b.x, b.y, b.w};
  const F4 t2 = t0 * b;
  const F4 t3 = t0 * t1;
  const F4 t4{t2.y, t2.z, t2.x, t2.w};
  return as_vec(t3 - t4);
}
RC_FN float norm2(F4 a) { return dot(a, a); }       // Vec3::norm (squared length) vec3.rs:171-173
RC_FN float length(F4 a) { return sqrtf(norm2(a)); }
RC_FN F4 normalized(F4 a) {
  const float len = length(a);
  if (len > kEps) return F4{a.x / len, a.y / len, a.z / len, a.w / 1.0f};
  return a;
}
RC_FN F4 reciprocal(F4 a) {
  F4 num{1.0f, 1.0f, 1.0f, 0.0f};
  F4 den{a.x, a.y, a.z, a.w + 1.0f};
  if (a.x == 0.0f) { num.x -= 1.0f; den.x += 1.0f; }
  if (a.y == 0.0f) { num.y -= 1.0f; den.y += 1.0f; }
  if (a.z == 0.0f) { num.z -= 1.0f; den.z += 1.0f; }
  return as_vec(F4{num.x / den.x, num.y / den.y, num.z / den.z, num.w / den.w});
}
RC_FN F4 reflect(F4 a, F4 n) { return a - (2.0f * dot(a, n)) * n; }
// Vec3::close: lexicographic `<` over the four lanes (vec3.rs:26,108-111)
RC_FN bool close(F4 a, F4 b) {
  const F4 d = a - b;
  const float l[4] = {fabsf(d.x), fabsf(d.y), fabsf(d.z), fabsf(d.w)};
  const float r[4] = {kEps, kEps, kEps, 0.0f};
  for (int i = 0; i < 4; ++i) {
    if (l[i] < r[i]) return true;
    if (l[i] > r[i]) return false;
    if (!(l[i] == r[i])) return false;
  }
  return false;
}
RC_FN F4 rotate(F4 v, F4 q) {
  const F4 u = as_vec(q * F4{1.0f, 1.0f, 1.0f, 0.0f});
  const float s = q.w;
  return ((2.0f * dot(u, v)) * u + (s * s - dot(u, u)) * v) + (2.0f * s) * cross(u, v);
}
RC_FN F4 to_vec(F4 p) { return as_vec(p - F4{0, 0, 0, 1.0f}); }    // Vec3::from(Point3)
RC_FN F4 to_point(F4 v) { return v + F4{0, 0, 0, 1.0f}; }          // Point3::from(Vec3)
RC_FN F4 point_scale(F4 p, F4 s) {                                   // Point3::scale (FMA)
  return F4{fmaf(p.x, s.x, 0.0f), fmaf(p.y, s.y, 0.0f), fmaf(p.z, s.z, 0.0f), fmaf(p.w, s.w, 1.0f)};
}
RC_FN F4 point_rotate(F4 p, F4 q) { return to_point(rotate(to_vec(p), q)); }
RC_FN F4 vmin(F4 a, F4 b) { return F4{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)}; }
RC_FN F4 vmax(F4 a, F4 b) { return F4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }

// ---- Quat / Trs (rayca-math/src/quat.rs:236-258, trs.rs:211-221,253-284) ----------------------
RC_FN F4 quat_mul(F4 a, F4 b) {
  return F4{a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x, -a.x * b.z + a.y * b.w + a.z * b.x + a.w * b.y,
            a.x * b.y - a.y * b.x + a.z * b.w + a.w * b.z, -a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w};
}
RC_FN F4 quat_conj(F4 q) { return q * F4{-1.0f, -1.0f, -1.0f, 1.0f}; }
RC_FN Trs trs_compose(const Trs& a, const Trs& b) {  // &a * &b
  Trs r;
  r.translation = a.translation + rotate(a.scale * b.translation, a.rotation);
  r.rotation = quat_mul(a.rotation, b.rotation);
  r.scale = rotate(a.scale * rotate(b.scale, b.rotation), quat_conj(b.rotation));
  return r;
}
RC_FN F4 trs_apply_point(const Trs& t, F4 p) { return point_rotate(point_scale(p, t.scale), t.rotation) + t.translation; }
RC_FN F4 trs_apply_vec(const Trs& t, F4 v) { return rotate(v * t.scale, t.rotation) + t.translation; }
RC_FN F4 trs_world_translation(const Trs& t) { return rotate(t.translation, t.rotation); }  // trs.rs:126-128

// ---- Mat3 (rayca-math/src/mat3.rs) ------------------------------------------------------------
RC_FN Mat3 mat3_identity() { return Mat3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
RC_FN Mat3 mat3_from_quat(F4 q) {  // mat3.rs:190-214
  const F4 xq = f4(q.x, q.x, q.x, q.x) * q, yq = f4(q.y, q.y, q.y, q.y) * q, zq = f4(q.z, q.z, q.z, q.z) * q;
  return Mat3{{{1.0f - 2.0f * (yq.y + zq.z), 2.0f * (xq.y - zq.w), 2.0f * (xq.z + yq.w)},
               {2.0f * (xq.y + zq.w), 1.0f - 2.0f * (xq.x + zq.z), 2.0f * (yq.z - xq.w)},
               {2.0f * (xq.z - yq.w), 2.0f * (yq.z + xq.w), 1.0f - 2.0f * (xq.x + yq.y)}}};
}
RC_FN Mat3 mat3_mul(const Mat3& a, const Mat3& b) {  // mat3.rs:159-180
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const float e = a.m[i][0] * b.m[0][j], f = a.m[i][1] * b.m[1][j], g = a.m[i][2] * b.m[2][j];
      r.m[i][j] = e + f + g;
    }
  return r;
}
RC_FN Mat3 mat3_from_scale(F4 s) {
  Mat3 r = mat3_identity();
  r.m[0][0] *= s.x;
  r.m[1][1] *= s.y;
  r.m[2][2] *= s.z;
  return r;
}
RC_FN Mat3 mat3_from_trs(const Trs& t) { return mat3_mul(mat3_from_quat(t.rotation), mat3_from_scale(t.scale)); }  // mat3.rs:126-132
RC_FN Mat3 mat3_from_inverse_trs(const Trs& t) {  // mat3.rs:140-144
  return mat3_mul(mat3_from_scale(reciprocal(t.scale)), mat3_mul(mat3_from_quat(quat_conj(t.rotation)), mat3_identity()));
}
RC_FN Mat3 mat3_transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
RC_FN F4 mat3_apply(const Mat3& m, F4 v) {  // mat3.rs:216-232: accumulates from 0.0
  const float in[3] = {v.x, v.y, v.z};
  float out[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[i] += m.m[i][j] * in[j];
  return vec3(out[0], out[1], out[2]);
}
RC_FN Mat3 mat3_tbn(F4 t, F4 b, F4 n) { return Mat3{{{t.x, b.x, n.x}, {t.y, b.y, n.y}, {t.z, b.z, n.z}}}; }

// ---- Mat4, only what the sphere normal path needs (rayca-math/src/mat4.rs) -----------------------
struct Mat4 {
  float m[4][4];
};
RC_FN Mat4 mat4_identity() { return Mat4{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
RC_FN Mat4 mat4_mul(const Mat4& a, const Mat4& b) {  // mat4.rs:160-188: ordered 16-lane sum, 12 lanes zero
  Mat4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = -0.0f;
      for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
      for (int k = 4; k < 16; ++k) s += 0.0f;
      r.m[i][j] = s;
    }
  return r;
}
RC_FN Mat4 mat4_from_quat(F4 q) {  // mat4.rs:205-236
  const Mat3 r3 = mat3_from_quat(q);
  Mat4 r = mat4_identity();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = r3.m[i][j];
  return r;
}
RC_FN Mat4 mat4_from_inverse_trs(const Trs& t) {  // trs.rs:372-381: S^-1 * (R^-1 * T^-1)
  Mat4 s = mat4_identity();
  const F4 is = reciprocal(t.scale);
  s.m[0][0] *= is.x;
  s.m[1][1] *= is.y;
  s.m[2][2] *= is.z;
  const Mat4 r = mat4_mul(mat4_from_quat(quat_conj(t.rotation)), mat4_identity());
  Mat4 tr = mat4_identity();
  tr.m[0][3] += -t.translation.x;
  tr.m[1][3] += -t.translation.y;
  tr.m[2][3] += -t.translation.z;
  return mat4_mul(s, mat4_mul(r, tr));
}
RC_FN F4 mat4_apply_point(const float (*m)[4], F4 p) {  // impl_mul3!(Point3, Mat4)  mat4.rs:296-320
  float ret[4];
  for (int i = 0; i < 4; ++i) ret[i] = hsum(f4(m[i][0], m[i][1], m[i][2], m[i][3]) * p);
  const float den = ret[3] != 0.0f ? ret[3] : 1.0f;
  return point3(ret[0] / den, ret[1] / den, ret[2] / den);
}

// ---- Color (rayca-math/src/color/mod.rs) -------------------------------------------------------
RC_FN Color rgba(float r, float g, float b, float a) { return Color{r, g, b, a}; }
RC_FN Color black() { return Color{0.0f, 0.0f, 0.0f, 1.0f}; }
RC_FN Color white() { return Color{1.0f, 1.0f, 1.0f, 1.0f}; }
RC_FN Color operator+(Color a, Color b) { return Color{a.r + b.r * b.a, a.g + b.g * b.a, a.b + b.b * b.a, a.a}; }
RC_FN Color operator-(Color a, Color b) { return Color{a.r - b.r * b.a, a.g - b.g * b.a, a.b - b.b * b.a, a.a}; }
RC_FN Color operator*(Color a, float f) { return Color{a.r * f, a.g * f, a.b * f, a.a}; }
RC_FN Color operator*(float f, Color a) { return Color{f * a.r, f * a.g, f * a.b, a.a}; }
RC_FN Color operator*(Color a, Color b) { return Color{a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }
RC_FN Color operator/(Color a, float f) { return Color{a.r / f, a.g / f, a.b / f, a.a}; }
RC_FN F4 premultiplied(Color c) { return vec3(c.r * c.a, c.g * c.a, c.b * c.a); }  // Vec3::from(&Color) vec3.rs:407-411
RC_FN bool is_transparent(Color c) { return c.a < 1.0f - FLT_EPSILON; }
RC_FN float clampf(float v, float lo, float hi) {  // f32::clamp (NaN stays NaN)
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}
RC_FN uint8_t quantize(float c) {  // RGBA8::from(Color)  color/rgba8.rs:75-84
  float v = c * 255.0f;
  if (v != v) return 0;
  if (v < 0.0f) v = 0.0f;
  if (v > 255.0f) v = 255.0f;
  return (uint8_t)v;
}

// ---- counter-based RNG (no reference counterpart; see include/rayca_hip.h RaycaConfig::seed) ----
RC_FN uint32_t rng_mix(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
RC_FN uint32_t rng_hash2(uint32_t a, uint32_t b) { return rng_mix(a * 0x9E3779B1u + rng_mix(b + 0x7F4A7C15u)); }
RC_FN uint32_t rng_root(uint32_t seed, uint32_t pixel, uint32_t sample) { return rng_hash2(rng_hash2(seed, pixel), sample); }
RC_FN uint32_t rng_child(uint32_t key, uint32_t k) { return rng_hash2(key, k + 1u); }
RC_FN float rng_f32(uint32_t key, uint32_t dim) {
  const uint32_t u = rng_hash2(key ^ 0xA511E9B3u, dim);
  const uint32_t bits = 0x3F800000u | (u >> 9);
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(bits) - 1.0f;
#else
  float f;
  __builtin_memcpy(&f, &bits, 4);
  return f - 1.0f;
#endif
}

}  // namespace rayca
