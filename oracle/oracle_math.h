/*
 * oracle_math.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic contract of rayca-math (Rust, f32x4 portable_simd) that the
 * hot path relies on.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/.
 *
 * Parity status: pinned by the reference's own known-answer unit tests restated in
 * tests/test_oracle_kat.py (rayca-math/src/vec3.rs:520-604, ray.rs:158-194, trs.rs:431-553,
 * quat.rs:301-396, color/rgba8.rs:110-132, rayca-geometry/src/triangle.rs:569-593,
 * sphere.rs:185-213, rayca-soft/src/bvh/blas.rs:403-490).  The reference ships no golden images
 * and cannot be built here (Rust nightly, no toolchain), so beyond those KATs the pixel-level
 * parity is "unpinned": it rests on this line-by-line restatement.
 *
 * Rules: plain f32, every operation rounded separately (build with -ffp-contract=off), operation
 * ORDER exactly as in the Rust source.  f32x4 lanes are kept as 4 scalars; `reduce_sum` of
 * core::simd is an ordered left-to-right add starting from -0.0
 * (library/core/src/../portable-simd: simd_reduce_add_ordered(self, -0.)).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <float.h>
#include <math.h>
#include <stdint.h>

typedef struct { float x, y, z, w; } V4; /* f32x4: Vec3 (w=0), Point3 (w=1), Quat (xyzw) */
typedef struct { float r, g, b, a; } Col; /* rayca-math/src/color/mod.rs:59-66 */
typedef struct { float x, y; } V2;        /* rayca-math/src/vec2.rs:9-14 */
typedef struct { V4 translation, rotation, scale; } Trs; /* rayca-math/src/trs.rs:75-86 */
typedef struct { float m[3][3]; } M3;     /* row-major, rayca-math/src/mat3.rs:12-17 */
typedef struct { float m[4][4]; } M4;     /* row-major, rayca-math/src/mat4.rs:12-17 */

/* rayca-math/src/lib.rs:33  const EPS: f32 = f32::EPSILON * 8192.0 */
#define ORC_EPS (FLT_EPSILON * 8192.0f)
/* rayca-math/src/ray.rs:57 */
#define ORC_RAY_BIAS 1e-4f

static inline V4 v4(float x, float y, float z, float w) { V4 r = {x, y, z, w}; return r; }
/* Vec3::new  vec3.rs:68-72 ; Point3::new point3.rs:25-29 */
static inline V4 vec3(float x, float y, float z) { return v4(x, y, z, 0.0f); }
static inline V4 point3(float x, float y, float z) { return v4(x, y, z, 1.0f); }
/* Vec3::simd forces w = 0  vec3.rs:78-81 */
static inline V4 vec3_simd(V4 s) { s.w = 0.0f; return s; }

/* f32x4::reduce_sum -- ordered, seeded with -0.0 */
static inline float reduce_sum4(V4 a) { return (((-0.0f + a.x) + a.y) + a.z) + a.w; }
static inline V4 mul4(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline V4 add4(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline V4 sub4(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline V4 splat4(float f) { return v4(f, f, f, f); }

/* Dot for Vec3 / Point3: (self.simd * rhs.simd).reduce_sum()  vec3.rs:240-244, point3.rs:124-136 */
static inline float dot4(V4 a, V4 b) { return reduce_sum4(mul4(a, b)); }

/* Vec3 (+,-,*) keep all four lanes (w stays 0 for Vec3 operands)  vec3.rs:246-379 */
static inline V4 vadd(V4 a, V4 b) { return add4(a, b); }
static inline V4 vsub(V4 a, V4 b) { return sub4(a, b); }
static inline V4 vmul(V4 a, V4 b) { return mul4(a, b); }
/* Mul<f32> for Vec3: simd *= splat(rhs)  vec3.rs:356-379 */
static inline V4 vscale(V4 a, float f) { return mul4(a, splat4(f)); }
static inline V4 vneg(V4 a) { return v4(-a.x, -a.y, -a.z, -a.w); }
/* Div<f32> for Vec3  vec3.rs:448-455 */
static inline V4 vdivf(V4 a, float f) { return v4(a.x / f, a.y / f, a.z / f, a.w / f); }

/* Vec3::cross  vec3.rs:134-142 (swizzle form: products rounded, then subtracted) */
static inline V4 vcross(V4 a, V4 b) {
  /* tmp0 = a.yzxw ; tmp1 = b.zxyw ; tmp2 = tmp0*b ; tmp3 = tmp0*tmp1 ; tmp4 = tmp2.yzxw */
  V4 tmp0 = v4(a.y, a.z, a.x, a.w);
  V4 tmp1 = v4(b.z, b.x, b.y, b.w);
  V4 tmp2 = mul4(tmp0, b);
  V4 tmp3 = mul4(tmp0, tmp1);
  V4 tmp4 = v4(tmp2.y, tmp2.z, tmp2.x, tmp2.w);
  return vec3_simd(sub4(tmp3, tmp4));
}

/* Vec3::norm = dot(self,self); len = sqrt  vec3.rs:171-177 */
static inline float vnorm(V4 a) { return dot4(a, a); }
static inline float vlen(V4 a) { return sqrtf(vnorm(a)); }
/* Vec3::normalize  vec3.rs:183-188: divides by [len,len,len,1] only if len > EPS */
static inline V4 vnormalize(V4 a) {
  float len = vlen(a);
  if (len > ORC_EPS) return v4(a.x / len, a.y / len, a.z / len, a.w / 1.0f);
  return a;
}
/* Vec3::get_reciprocal  vec3.rs:195-216: zero components give 0, not inf */
static inline V4 vreciprocal(V4 a) {
  V4 num = v4(1.0f, 1.0f, 1.0f, 0.0f);
  V4 den = add4(a, v4(0.0f, 0.0f, 0.0f, 1.0f));
  if (a.x == 0.0f) { num.x -= 1.0f; den.x += 1.0f; }
  if (a.y == 0.0f) { num.y -= 1.0f; den.y += 1.0f; }
  if (a.z == 0.0f) { num.z -= 1.0f; den.z += 1.0f; }
  return vec3_simd(v4(num.x / den.x, num.y / den.y, num.z / den.z, num.w / den.w));
}
/* Vec3::reflect  vec3.rs:219-221:  self - 2.0 * self.dot(normal) * normal */
static inline V4 vreflect(V4 a, V4 n) { return vsub(a, vscale(n, 2.0f * dot4(a, n))); }
/* Vec3::close  vec3.rs:108-111: derived PartialOrd on f32x4 => LEXICOGRAPHIC `<` over lanes */
static inline int lex_lt4(V4 a, V4 b) {
  const float l[4] = {a.x, a.y, a.z, a.w}, r[4] = {b.x, b.y, b.z, b.w};
  for (int i = 0; i < 4; ++i) {
    if (l[i] < r[i]) return 1;  /* Some(Less) */
    if (l[i] > r[i]) return 0;  /* Some(Greater) */
    if (!(l[i] == r[i])) return 0; /* None (NaN) */
  }
  return 0; /* Some(Equal) */
}
static inline int vclose(V4 a, V4 b) {
  V4 d = sub4(a, b);
  d = v4(fabsf(d.x), fabsf(d.y), fabsf(d.z), fabsf(d.w));
  return lex_lt4(d, vec3(ORC_EPS, ORC_EPS, ORC_EPS));
}

/* Vec3::rotate  vec3.rs:148-159 and Point3::rotate  point3.rs:65-76 (same formula):
 *   2.0 * u.dot(v) * u + (s*s - u.dot(u)) * v + 2.0 * s * u.cross(v)          */
static inline V4 vrotate(V4 v, V4 q) {
  V4 u = vec3_simd(mul4(q, v4(1.0f, 1.0f, 1.0f, 0.0f)));
  float s = q.w;
  V4 t0 = vscale(u, 2.0f * dot4(u, v));
  V4 t1 = vscale(v, s * s - dot4(u, u));
  V4 t2 = vscale(vcross(u, v), 2.0f * s);
  return vadd(vadd(t0, t1), t2);
}
/* Vec3::from(Point3): w -= 1  vec3.rs:394-399 ;  Point3::from(Vec3): w += 1  point3.rs:117-122 */
static inline V4 vec_from_point(V4 p) { return vec3_simd(sub4(p, v4(0, 0, 0, 1.0f))); }
static inline V4 point_from_vec(V4 v) { return add4(v, v4(0, 0, 0, 1.0f)); }
/* Point3::scale  point3.rs:59-63: simd.mul_add(scale, [0,0,0,1]) -- a true fused multiply-add */
static inline V4 pscale(V4 p, V4 s) {
  return v4(fmaf(p.x, s.x, 0.0f), fmaf(p.y, s.y, 0.0f), fmaf(p.z, s.z, 0.0f), fmaf(p.w, s.w, 1.0f));
}
static inline V4 protate(V4 p, V4 q) { return point_from_vec(vrotate(vec_from_point(p), q)); }
static inline V4 ptranslate(V4 p, V4 t) { return add4(p, t); }
/* simd_min / simd_max: fmin/fmax semantics lane-wise  point3.rs:83-97 */
static inline V4 min4(V4 a, V4 b) { return v4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)); }
static inline V4 max4(V4 a, V4 b) { return v4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

/* ---- Quat  rayca-math/src/quat.rs ----------------------------------------------------------- */
/* Mul<Quat> for Quat  quat.rs:236-258 (Hamilton product, term order as written) */
static inline V4 qmul(V4 a, V4 b) {
  return v4(a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x,
            -a.x * b.z + a.y * b.w + a.z * b.x + a.w * b.y,
            a.x * b.y - a.y * b.x + a.z * b.w + a.w * b.z,
            -a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w);
}
/* get_conjugate  quat.rs:95-97 ; get_inverse asserts normalised then conjugates  quat.rs:99-103 */
static inline V4 qconj(V4 q) { return mul4(q, v4(-1.0f, -1.0f, -1.0f, 1.0f)); }
static inline float qlen(V4 q) { return sqrtf(dot4(q, q)); }
static inline int q_is_normalized(V4 q) { return fabsf(qlen(q) - 1.0f) < 0.001f; }
static inline V4 qnormalize(V4 q) { float l = qlen(q); return v4(q.x / l, q.y / l, q.z / l, q.w / l); }
/* Quat::axis_angle  quat.rs:67-77 */
static inline V4 q_axis_angle(V4 axis, float angle) {
  float factor = sinf(angle / 2.0f);
  V4 s = add4(mul4(axis, splat4(factor)), v4(0.0f, 0.0f, 0.0f, cosf(angle / 2.0f)));
  return qnormalize(s);
}

/* ---- Trs  rayca-math/src/trs.rs ------------------------------------------------------------- */
static inline Trs trs_identity(void) {
  Trs t = {vec3(0, 0, 0), v4(0, 0, 0, 1.0f), vec3(1.0f, 1.0f, 1.0f)};
  return t;
}
/* Mul<&Trs> for &Trs  trs.rs:211-221 */
static inline Trs trs_mul(const Trs* a, const Trs* b) {
  Trs r;
  r.translation = vadd(a->translation, vrotate(vmul(a->scale, b->translation), a->rotation));
  r.rotation = qmul(a->rotation, b->rotation);
  /* rhs.rotation.get_inverse() * (self.scale * (rhs.rotation * rhs.scale)) */
  r.scale = vrotate(vmul(a->scale, vrotate(b->scale, b->rotation)), qconj(b->rotation));
  return r;
}
/* Mul<Point3> for &Trs  trs.rs:264-273 */
static inline V4 trs_point(const Trs* t, V4 p) {
  return ptranslate(protate(pscale(p, t->scale), t->rotation), t->translation);
}
/* Mul<Vec3> for &Trs  trs.rs:253-262 (Vec3::scale is a plain multiply, vec3.rs:144-146) */
static inline V4 trs_vec(const Trs* t, V4 v) {
  return vadd(vrotate(vmul(v, t->scale), t->rotation), t->translation);
}
/* Trs::get_translation = rotation * translation  trs.rs:126-128 */
static inline V4 trs_get_translation(const Trs* t) { return vrotate(t->translation, t->rotation); }

/* ---- Mat3  rayca-math/src/mat3.rs ----------------------------------------------------------- */
static inline M3 m3_identity(void) { M3 r = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; return r; }
/* From<&Quat> for Mat3  mat3.rs:190-214 */
static inline M3 m3_from_quat(V4 q) {
  V4 xq = mul4(splat4(q.x), q), yq = mul4(splat4(q.y), q), zq = mul4(splat4(q.z), q);
  M3 r = {{{1.0f - 2.0f * (yq.y + zq.z), 2.0f * (xq.y - zq.w), 2.0f * (xq.z + yq.w)},
           {2.0f * (xq.y + zq.w), 1.0f - 2.0f * (xq.x + zq.z), 2.0f * (yq.z - xq.w)},
           {2.0f * (xq.z - yq.w), 2.0f * (yq.z + xq.w), 1.0f - 2.0f * (xq.x + yq.y)}}};
  return r;
}
/* Mul<&Mat3> for Mat3  mat3.rs:159-180:  ret[i][j] = e + f + g */
static inline M3 m3_mul(const M3* a, const M3* b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float e = a->m[i][0] * b->m[0][j], f = a->m[i][1] * b->m[1][j], g = a->m[i][2] * b->m[2][j];
      r.m[i][j] = e + f + g;
    }
  return r;
}
/* Mat3::from_scale / scale  mat3.rs:44-48,67-71 */
static inline M3 m3_from_scale(V4 s) {
  M3 r = m3_identity();
  r.m[0][0] *= s.x; r.m[1][1] *= s.y; r.m[2][2] *= s.z;
  return r;
}
/* Mat3::rotate: *self = Mat3::from(q) * self  mat3.rs:73-75 */
static inline M3 m3_rotate(const M3* m, V4 q) { M3 rq = m3_from_quat(q); return m3_mul(&rq, m); }
/* From<&Trs> for Mat3  mat3.rs:126-132 */
static inline M3 m3_from_trs(const Trs* t) { M3 s = m3_from_scale(t->scale); return m3_rotate(&s, t->rotation); }
/* From<&Inversed<&Trs>> for Mat3  mat3.rs:140-144: from_scale(1/s) * from_rotation(R^-1) */
static inline M3 m3_from_inv_trs(const Trs* t) {
  M3 s = m3_from_scale(vreciprocal(t->scale));
  M3 id = m3_identity();
  M3 r = m3_rotate(&id, qconj(t->rotation));
  return m3_mul(&s, &r);
}
static inline M3 m3_transpose(const M3* m) {
  M3 r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m->m[j][i];
  return r;
}
/* Mul<Vec3> for &Mat3  mat3.rs:216-232: ret[i] starts at 0.0 and accumulates j = 0,1,2 */
static inline V4 m3_vec(const M3* m, V4 v) {
  float in[3] = {v.x, v.y, v.z}, out[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[i] += m->m[i][j] * in[j];
  return vec3(out[0], out[1], out[2]);
}
/* Mat3::tbn  mat3.rs:57-65 */
static inline M3 m3_tbn(V4 t, V4 b, V4 n) {
  M3 r = {{{t.x, b.x, n.x}, {t.y, b.y, n.y}, {t.z, b.z, n.z}}};
  return r;
}

/* ---- Mat4 (only what the sphere normal path needs)  rayca-math/src/mat4.rs ------------------- */
static inline M4 m4_identity(void) { M4 r = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; return r; }
/* Mul<&Mat4> for Mat4  mat4.rs:160-188: each element = ordered sum of 4 products (12 zero lanes
 * of the f32x16 reduce_sum add +0.0 afterwards, which cannot change a finite value) */
static inline M4 m4_mul(const M4* a, const M4* b) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = -0.0f;
      for (int k = 0; k < 4; ++k) s += a->m[i][k] * b->m[k][j];
      for (int k = 4; k < 16; ++k) s += 0.0f;
      r.m[i][j] = s;
    }
  return r;
}
/* From<&Quat> for Mat4  mat4.rs:205-236 */
static inline M4 m4_from_quat(V4 q) {
  M3 r3 = m3_from_quat(q);
  M4 r = m4_identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = r3.m[i][j];
  return r;
}
/* from_translation / from_rotation / from_scale  mat4.rs:62-78,97-111 */
static inline M4 m4_from_translation(V4 t) { M4 r = m4_identity(); r.m[0][3] += t.x; r.m[1][3] += t.y; r.m[2][3] += t.z; return r; }
static inline M4 m4_from_rotation(V4 q) { M4 rq = m4_from_quat(q), id = m4_identity(); return m4_mul(&rq, &id); }
static inline M4 m4_from_scale(V4 s) { M4 r = m4_identity(); r.m[0][0] *= s.x; r.m[1][1] *= s.y; r.m[2][2] *= s.z; return r; }
/* impl_mul3!(Point3, Mat4)  mat4.rs:296-320 */
static inline V4 m4_point(const M4* m, V4 p) {
  float ret[4];
  for (int i = 0; i < 4; ++i) ret[i] = reduce_sum4(mul4(v4(m->m[i][0], m->m[i][1], m->m[i][2], m->m[i][3]), p));
  float den = ret[3] != 0.0f ? ret[3] : 1.0f;
  return point3(ret[0] / den, ret[1] / den, ret[2] / den);
}
/* impl_mul3!(Vec3, Mat4)  mat4.rs:296-320: the same macro with a w = 0 operand */
static inline V4 m4_vec(const M4* m, V4 v) {
  float ret[4];
  for (int i = 0; i < 4; ++i) ret[i] = reduce_sum4(mul4(v4(m->m[i][0], m->m[i][1], m->m[i][2], m->m[i][3]), v));
  float den = ret[3] != 0.0f ? ret[3] : 1.0f;
  return vec3(ret[0] / den, ret[1] / den, ret[2] / den);
}
/* get_transpose  mat4.rs:131-137 */
static inline M4 m4_transpose(const M4* m) {
  M4 r;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = m->m[j][i];
  return r;
}
/* Mat4::look_at  mat4.rs:81-95 (the Z axis points towards the eye) */
static inline M4 m4_look_at(V4 target, V4 eye, V4 up) {
  V4 z = vnormalize(vsub(eye, target));
  V4 x = vnormalize(vcross(up, z));
  V4 y = vcross(z, x);
  V4 ne = vneg(eye);
  M4 r = {{{x.x, x.y, x.z, x.w + dot4(x, ne)}, {y.x, y.y, y.z, y.w + dot4(y, ne)}, {z.x, z.y, z.z, z.w + dot4(z, ne)}, {0, 0, 0, 1}}};
  return r;
}
/* From<&Mat4> for Quat  quat.rs:184-226 (Mat4::get_rotation  mat4.rs:117-119) */
static inline V4 q_from_m4(const M4* m) {
  V4 r;
  float t = m->m[0][0] + m->m[1][1] + m->m[2][2];
  if (t > 0.0f) {
    float s = 0.5f / sqrtf(t + 1.0f);
    r = v4((m->m[2][1] - m->m[1][2]) * s, (m->m[0][2] - m->m[2][0]) * s, (m->m[1][0] - m->m[0][1]) * s, 0.25f / s);
  } else if (m->m[0][0] > m->m[1][1] && m->m[0][0] > m->m[2][2]) {
    float s = 2.0f * sqrtf(1.0f + m->m[0][0] - m->m[1][1] - m->m[2][2]);
    r = v4(0.25f * s, (m->m[0][1] + m->m[1][0]) / s, (m->m[0][2] + m->m[2][0]) / s, (m->m[2][1] - m->m[1][2]) / s);
  } else if (m->m[1][1] > m->m[2][2]) {
    float s = 2.0f * sqrtf(1.0f + m->m[1][1] - m->m[0][0] - m->m[2][2]);
    r = v4((m->m[0][1] + m->m[1][0]) / s, 0.25f * s, (m->m[1][2] + m->m[2][1]) / s, (m->m[0][2] - m->m[2][0]) / s);
  } else {
    float s = 2.0f * sqrtf(1.0f + m->m[2][2] - m->m[0][0] - m->m[1][1]);
    r = v4((m->m[0][2] + m->m[2][0]) / s, (m->m[1][2] + m->m[2][1]) / s, 0.25f * s, (m->m[1][0] - m->m[0][1]) / s);
  }
  return qnormalize(r);
}
/* Mul<Point3> for &Inversed<Trs>  trs.rs:372-381 */
static inline V4 inv_trs_point(const Trs* t, V4 p) {
  M4 s = m4_from_scale(vreciprocal(t->scale));
  M4 r = m4_from_rotation(qconj(t->rotation));
  M4 tr = m4_from_translation(vneg(t->translation));
  M4 rt = m4_mul(&r, &tr);
  M4 m = m4_mul(&s, &rt);
  return m4_point(&m, p);
}

/* ---- Color  rayca-math/src/color/mod.rs ----------------------------------------------------- */
static inline Col col(float r, float g, float b, float a) { Col c = {r, g, b, a}; return c; }
#define COL_BLACK col(0.0f, 0.0f, 0.0f, 1.0f)
#define COL_WHITE col(1.0f, 1.0f, 1.0f, 1.0f)
/* Add: self.rgb += rhs.rgb * rhs.a ; alpha of lhs kept  color/mod.rs:239-286 */
static inline Col cadd(Col a, Col b) { return col(a.r + b.r * b.a, a.g + b.g * b.a, a.b + b.b * b.a, a.a); }
/* Sub  color/mod.rs:288-298 */
static inline Col csub(Col a, Col b) { return col(a.r - b.r * b.a, a.g - b.g * b.a, a.b - b.b * b.a, a.a); }
/* Mul<f32> (either side): rgb scaled, alpha kept  color/mod.rs:300-360 */
static inline Col cmulf(Col a, float f) { return col(a.r * f, a.g * f, a.b * f, a.a); }
static inline Col fmulc(float f, Col a) { return col(f * a.r, f * a.g, f * a.b, a.a); }
/* Mul<Color>: all four channels  color/mod.rs:362-408 */
static inline Col cmul(Col a, Col b) { return col(a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a); }
/* Div<f32>  color/mod.rs:410-424 */
static inline Col cdivf(Col a, float f) { return col(a.r / f, a.g / f, a.b / f, a.a); }
/* From<Vec3> for Color (alpha 1)  color/mod.rs:199-203 ; From<&Color> for Vec3 premultiplies
 * by alpha  vec3.rs:407-411 */
static inline Col col_from_vec(V4 v) { return col(v.x, v.y, v.z, 1.0f); }
static inline V4 vec_from_col(Col c) { return vec3(c.r * c.a, c.g * c.a, c.b * c.a); }
/* Color::close  color/mod.rs:160-169 */
static inline int cclose(Col a, Col b) {
  return fabsf(a.r - b.r) < FLT_EPSILON && fabsf(a.g - b.g) < FLT_EPSILON &&
         fabsf(a.b - b.b) < FLT_EPSILON && fabsf(a.a - b.a) < FLT_EPSILON;
}
/* is_transparent  color/mod.rs:156-158 */
static inline int c_is_transparent(Col c) { return c.a < 1.0f - FLT_EPSILON; }
/* Color::over  color/mod.rs:149-154 */
static inline Col cover(Col s, Col top) {
  return col(top.r * top.a + s.r * (1.0f - top.a), top.g * top.a + s.g * (1.0f - top.a),
             top.b * top.a + s.b * (1.0f - top.a), 1.0f);
}
/* correct_gamma  color/mod.rs:175-180 */
static inline Col c_gamma(Col c, float gamma) {
  float f = 1.0f / gamma;
  return col(powf(c.r, f), powf(c.g, f), powf(c.b, f), c.a);
}
/* f32::max semantics (NaN-ignoring)  color/mod.rs:183-185 */
static inline float c_max_rgb(Col c) { return fmaxf(fmaxf(c.r, c.g), c.b); }
/* RGBA8::from(Color)  color/rgba8.rs:75-84: (c*255).clamp(0,255) as u8 -- `as` truncates toward
 * zero and maps NaN to 0; f32::clamp keeps NaN */
static inline uint8_t to_u8(float c) {
  float v = c * 255.0f;
  if (v != v) return 0;
  if (v < 0.0f) v = 0.0f;
  if (v > 255.0f) v = 255.0f;
  return (uint8_t)v;
}
/* f32::clamp(min,max): NaN stays NaN */
static inline float clampf(float v, float lo, float hi) {
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}

/* ---- V2  rayca-math/src/vec2.rs -------------------------------------------------------------- */
static inline V2 v2(float x, float y) { V2 r = {x, y}; return r; }
static inline V2 v2add(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
static inline V2 v2scale(V2 a, float f) { return v2(a.x * f, a.y * f); }

#endif /* ORACLE_MATH_H */
