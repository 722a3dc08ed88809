/*
 * rayca_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C API of liborayca_oracle.so: a CPU restatement of rayca-soft's SoftRenderer::draw
 * (rayca-soft/src/scene.rs:88-154) and everything below it, taking the same flat scene description
 * as the product (include/rayca_hip.h is the shared *interface* spec; no product code is used).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#ifndef RAYCA_ORACLE_H
#define RAYCA_ORACLE_H

#include "../include/rayca_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene OracleScene;

enum {
  /* literal Blas::evaluate_sah: 63 planes x 3 axes, one pass over the node's primitives each
   * (bvh/blas.rs:64-123) */
  ORACLE_BUILD_LITERAL = 0,
  /* same cost values from one binned sweep per axis (exactly equal results; checked against
   * LITERAL in tests/test_oracle_bvh.py) -- lets the oracle build large scenes */
  ORACLE_BUILD_BINNED = 1
};
enum {
  /* per ray per triangle world transform of the three vertices, as Triangle::intersects does
   * (rayca-geometry/src/triangle.rs:85-87): the reference's cost profile, used for CPU timing */
  ORACLE_XFORM_PER_TEST = 0,
  /* world-space vertices computed once with the same operations (bit-identical results) */
  ORACLE_XFORM_CACHED = 1
};

typedef struct OracleOptions {
  uint32_t build;   /* ORACLE_BUILD_* */
  uint32_t xform;   /* ORACLE_XFORM_* */
  uint32_t threads; /* 0 => all online cores */
  uint32_t reserved;
} OracleOptions;

typedef struct OracleBvhNode { /* BvhNode, bvh/blas.rs:11-15 */
  float a[4], b[4];
  uint32_t offset, count;
} OracleBvhNode;

const char* oracle_last_error(void);
int32_t oracle_scene_create(const RaycaSceneDesc* desc, const RaycaConfig* cfg,
                            const OracleOptions* opts, OracleScene** out);
void oracle_scene_destroy(OracleScene* s);

/* counts after the build */
uint32_t oracle_scene_blas_count(const OracleScene* s);
uint32_t oracle_scene_primitive_count(const OracleScene* s);
uint32_t oracle_blas_node_count(const OracleScene* s, uint32_t blas);
uint32_t oracle_blas_primitive_count(const OracleScene* s, uint32_t blas);
/* blas = index in TLAS blas_nodes order (post-build); nodes in the reference's layout (root 0,
 * slot 1 unused, children adjacent) */
int32_t oracle_blas_nodes(const OracleScene* s, uint32_t blas, OracleBvhNode* out, uint32_t cap);
/* global post-build primitive order: concatenation over blas_nodes order of each BLAS's
 * primitives; value = index of that primitive in flatten order */
int32_t oracle_scene_primitive_order(const OracleScene* s, uint32_t* out, uint32_t cap);
/* world-space vertices (9 floats per triangle, flatten order); spheres give zeros */
int32_t oracle_scene_world_triangles(const OracleScene* s, float* out, uint32_t cap_tris);

/* SoftRenderer::draw pixel loop.  tile may be NULL.  seconds_out = wall time of the pixel loop
 * only, like the reference's timer (scene.rs:101,152). */
int32_t oracle_render(OracleScene* s, const RaycaConfig* cfg, uint32_t width, uint32_t height,
                      const RaycaTile* tile, uint8_t* rgba8_out, float* rgba32f_out,
                      RaycaStats* stats_out, double* seconds_out);
/* Render only rows [row_begin, row_end) -- the bounded CPU-baseline sample. */
int32_t oracle_render_rows(OracleScene* s, const RaycaConfig* cfg, uint32_t width, uint32_t height,
                           uint32_t row_begin, uint32_t row_end, uint8_t* rgba8_out,
                           float* rgba32f_out, RaycaStats* stats_out, double* seconds_out);
/* Tlas::intersects for caller-supplied rays; prim_out in global post-build order */
int32_t oracle_trace_rays(OracleScene* s, uint32_t count, const float* rays, float* t_out,
                          uint32_t* prim_out, float* uv_out, RaycaStats* stats_out);

/* ---- known-answer hooks for the restated unit tests (tests/test_oracle_kat.py) -------------- */
void oracle_vec3_rotate(const float v[3], const float q[4], float out[3]);
void oracle_vec3_normalize(const float v[3], float out[3]);
void oracle_vec3_reciprocal(const float v[3], float out[3]);
void oracle_vec3_reflect(const float v[3], const float n[3], float out[3]);
void oracle_vec3_cross(const float a[3], const float b[3], float out[3]);
float oracle_vec3_dot(const float a[3], const float b[3]);
int32_t oracle_vec3_close(const float a[3], const float b[3]);
void oracle_quat_mul(const float a[4], const float b[4], float out[4]);
void oracle_trs_mul(const RaycaTrs* a, const RaycaTrs* b, RaycaTrs* out);
void oracle_trs_point(const RaycaTrs* t, const float p[3], float out[3]);
void oracle_trs_vec(const RaycaTrs* t, const float v[3], float out[3]);
void oracle_inv_trs_vec(const RaycaTrs* t, const float v[3], float out[3]);
/* &trs * Ray::new(origin, dir): out = origin xyz, dir xyz, rdir xyz */
void oracle_trs_ray(const RaycaTrs* t, const float origin[3], const float dir[3], float out[9]);
void oracle_rgba8_from_color(const float c[4], uint8_t out[4]);
void oracle_color_add(const float a[4], const float b[4], float out[4]);
/* Triangle::intersects(trs, ray): returns 1 on hit and fills t,u,v,point */
int32_t oracle_triangle_intersects(const float verts[9], const RaycaTrs* trs, const float origin[3],
                                   const float dir[3], float* t, float uv[2], float point[3]);
int32_t oracle_sphere_intersects(const float center[3], float radius, const RaycaTrs* trs,
                                 const float origin[3], const float dir[3], float* t,
                                 float point[3]);
/* Triangle / Sphere ::get_centroid, min, max under `trs` (triangle.rs:160-177, sphere.rs:164-178) */
void oracle_triangle_bounds(const float verts[9], const RaycaTrs* trs, float centroid[3], float mn[3], float mx[3]);
void oracle_sphere_bounds(const float center[3], float radius, const RaycaTrs* trs, float centroid[3], float mn[3], float mx[3]);
/* BvhPrimitive::intersects(scene, ray) of the primitive with flatten-order index `src`, without any BVH
 * (bvh/triangle.rs:83-114): 1 hit, 0 miss, -1 no such primitive */
int32_t oracle_scene_primitive_intersects(const OracleScene* s, uint32_t src, const float origin[3], const float dir[3], float* t, float uv[2]);
/* Mat4 (row-major, 16 floats)  rayca-math/src/mat4.rs */
void oracle_mat4_identity(float out[16]);
void oracle_mat4_mul(const float a[16], const float b[16], float out[16]);
void oracle_mat4_from_scale(const float s[3], float out[16]);
void oracle_mat4_from_translation(const float t[3], float out[16]);
void oracle_mat4_transpose(const float m[16], float out[16]);
void oracle_mat4_look_at(const float target[3], const float eye[3], const float up[3], float out[16]);
void oracle_mat4_get_rotation(const float m[16], float out[4]);
void oracle_mat4_mul_vec3(const float m[16], const float v[3], float out[3]);
void oracle_mat4_mul_point3(const float m[16], const float p[3], float out[3]);
/* Quat  rayca-math/src/quat.rs */
void oracle_quat_axis_angle(const float axis[3], float angle, float out[4]);
void oracle_quat_conjugate(const float q[4], float out[4]);
void oracle_quat_normalize(const float q[4], float out[4]);
int32_t oracle_quat_is_normalized(const float q[4]);
float oracle_quat_dot(const float a[4], const float b[4]);
float oracle_quat_len(const float q[4]);
/* Vec3 arithmetic: a+b, b-a, a*s, b/s, -a; lane-wise min / max  (vec3.rs:556-573) */
void oracle_vec3_arith(const float a[3], const float b[3], float s, float add[3], float sub[3], float mul[3], float div[3], float neg[3]);
void oracle_vec3_min_max(const float a[3], const float b[3], float mn[3], float mx[3]);
/* AABB::intersects: returns tmin or f32::MAX */
float oracle_aabb_intersects(const float a[3], const float b[3], const float origin[3],
                             const float dir[3]);
/* Sampler::sample wrap rule: returns texel x for coordinate u on an image `size` wide */
uint32_t oracle_sampler_wrap(float u, uint32_t size);
/* counter-based RNG: root key, child key, draw */
uint32_t oracle_rng_root(uint32_t seed, uint32_t pixel, uint32_t sample);
uint32_t oracle_rng_child(uint32_t key, uint32_t k);
float oracle_rng_f32(uint32_t key, uint32_t dim);

#ifdef __cplusplus
}
#endif
#endif
