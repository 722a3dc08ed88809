/*
 * rayca_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the rayca-soft hot path, following the Rust source function by function
 * (citations are paths relative to /root/reference).  It is the parity oracle for the HIP kernels
 * and, in ORACLE_XFORM_PER_TEST mode with all host cores, the "reference CPU path" that bench.py
 * times next to the GPU.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use it; the product (rayca_amd/) never links, loads or calls anything in this directory.
 *
 * Parity status: see oracle_math.h.  The real rayca-soft cannot be compiled in this environment
 * (Rust nightly + portable_simd, no cargo/rustc, no network), and its tests/model assets are an
 * un-vendored submodule, so this file is pinned by the reference's known-answer unit tests only
 * ("parity unpinned" beyond them).
 *
 * Declared deviations from the reference (none changes a result):
 *  - world transforms are looked up by array index instead of a SipHash HashMap
 *    (rayca-soft/src/scene.rs:166,284-286);
 *  - BLAS creation order is ascending model id; the reference iterates HashMap::values()
 *    (bvh/primitive.rs:388), i.e. leaves the order unspecified;
 *  - random numbers are a counter-based function of (seed, pixel, sample, path vertex, dimension)
 *    instead of a thread-local OS-seeded fastrand (sampler/cosine.rs:66-67), whose output is not
 *    reproducible even between two runs of the reference;
 *  - pixels are distributed over pthreads by rows instead of rayon tasks (scene.rs:117-122);
 *  - debug asserts of the reference (quaternion normalised, origin.w == 1) are not evaluated.
 */
#define _GNU_SOURCE
#include "rayca_oracle.h"

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "oracle_math.h"

/* ============================================================================================ */
/* errors                                                                                        */
/* ============================================================================================ */
static __thread char g_err[512];
static int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
const char* oracle_last_error(void) { return g_err; }

/* ============================================================================================ */
/* counter-based RNG (the one declared deviation that is visible in bounced images)              */
/* ============================================================================================ */
static inline uint32_t rng_mix(uint32_t h) { /* murmur3 fmix32 */
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
static inline uint32_t rng_hash2(uint32_t a, uint32_t b) {
  return rng_mix(a * 0x9E3779B1u + rng_mix(b + 0x7F4A7C15u));
}
uint32_t oracle_rng_root(uint32_t seed, uint32_t pixel, uint32_t sample) {
  return rng_hash2(rng_hash2(seed, pixel), sample);
}
uint32_t oracle_rng_child(uint32_t key, uint32_t k) { return rng_hash2(key, k + 1u); }
/* same bit recipe as fastrand::f32 (fastrand 2.3.0, Cargo.lock:769): 23 random mantissa bits in
 * [1,2) minus 1 */
float oracle_rng_f32(uint32_t key, uint32_t dim) {
  uint32_t u = rng_hash2(key ^ 0xA511E9B3u, dim);
  uint32_t bits = 0x3F800000u | (u >> 9);
  float f;
  memcpy(&f, &bits, 4);
  return f - 1.0f;
}
typedef struct { uint32_t key, dim; } Rng;
static inline float rng_next(Rng* r) { return oracle_rng_f32(r->key, r->dim++); }

/* ============================================================================================ */
/* scene types                                                                                   */
/* ============================================================================================ */
typedef struct { /* VertexExt  rayca-geometry/src/vertex.rs:64-72 */
  Col color;
  V4 normal, tangent, bitangent;
  V2 uv;
} VertexExt;

typedef struct { /* BvhPrimitive  bvh/primitive.rs:52-60 (+ BvhTriangle / Sphere payload) */
  uint32_t kind;     /* RAYCA_GEOMETRY_* */
  uint32_t node;     /* NodeDrawInfo -> index of the world Trs */
  uint32_t material; /* global material index or RAYCA_NONE */
  uint32_t src;      /* index in flatten order */
  V4 p[3];           /* Triangle.vertices, MODEL space (primitive.rs:211 "a.pos = a.pos") */
  V4 centroid;       /* Triangle.centroid (model space)  triangle.rs:59-63 */
  VertexExt ext[3];
  V4 center;         /* Sphere */
  float radius, radius2;
  V4 wp[3];          /* cached world-space vertices (ORACLE_XFORM_CACHED / binned build) */
  V4 wcentroid;      /* cached get_centroid(scene) */
  V4 wmin, wmax;     /* cached min/max(scene) */
} Prim;

typedef struct { V4 a, b; } AABB; /* bvh/aabb.rs:9-13 */

typedef struct { /* BvhNode  bvh/blas.rs:11-15 */
  AABB bounds;
  uint32_t offset, count; /* BvhRange<BvhPrimitive> */
} BNode;

typedef struct { /* Blas  bvh/blas.rs:213-222 */
  uint8_t max_depth;
  BNode* nodes;
  uint32_t node_count, node_cap;
  Prim* prims;
  uint32_t prim_count;
  uint32_t model;
} Blas;

typedef struct TNode { /* TlasNode  bvh/tlas.rs:41-49 */
  AABB bounds;
  int32_t left, right; /* Handle<TlasNode>, -1 = NONE */
  uint32_t blas_offset, blas_count;
} TNode;

typedef struct { uint32_t blas; uint32_t model; } BlasNode; /* bvh/tlas.rs:185-189 */

typedef struct {
  uint32_t kind;
  uint32_t node;  /* node index: node-LOCAL trs is used for sampling (nee.rs:85,133) */
  RaycaLight l;
} LightInfo;

struct OracleScene {
  /* SceneDrawInfo  scene.rs:162-182 */
  uint32_t node_count;
  Trs* local_trs;
  Trs* world_trs;
  uint8_t* has_world;
  uint32_t camera_node; /* camera_draw_infos[0] */
  float camera_yfov;
  int has_camera;
  LightInfo* lights; /* light_draw_infos (traversal order) */
  uint32_t light_count;
  /* materials / textures */
  RaycaMaterial* materials;
  uint32_t material_count;
  RaycaTexture* textures;
  uint32_t texture_count;
  RaycaImage* images;
  uint32_t image_count;
  uint8_t* image_bytes;
  uint64_t image_byte_count;
  /* Tlas  bvh/tlas.rs:229-241 */
  TNode root;
  TNode* tnodes;
  uint32_t tnode_count, tnode_cap;
  BlasNode* blas_nodes;
  Blas* blass;
  uint32_t blas_count;
  uint32_t* blas_prim_base; /* global primitive index base per blas_nodes slot */
  uint32_t prim_total;
  OracleOptions opts;
  uint32_t flat_prim_count;
  float* world_tris; /* flatten-order world vertices for export */
};

static inline Trs trs_from_abi(const RaycaTrs* t) {
  Trs r;
  r.translation = vec3(t->translation[0], t->translation[1], t->translation[2]);
  r.rotation = v4(t->rotation[0], t->rotation[1], t->rotation[2], t->rotation[3]);
  r.scale = vec3(t->scale[0], t->scale[1], t->scale[2]);
  return r;
}
static inline void trs_to_abi(const Trs* t, RaycaTrs* o) {
  o->translation[0] = t->translation.x; o->translation[1] = t->translation.y; o->translation[2] = t->translation.z;
  o->rotation[0] = t->rotation.x; o->rotation[1] = t->rotation.y; o->rotation[2] = t->rotation.z; o->rotation[3] = t->rotation.w;
  o->scale[0] = t->scale.x; o->scale[1] = t->scale.y; o->scale[2] = t->scale.z;
}

/* ============================================================================================ */
/* Ray / Hit  rayca-math/src/ray.rs                                                              */
/* ============================================================================================ */
typedef struct { V4 origin, dir, rdir; Col throughput; } Ray; /* ray.rs:42-54 */
typedef struct { /* ray.rs:115-134 */
  Ray ray;
  V4 point;
  uint32_t blas, primitive;
  float depth;
  V2 uv;
} Hit;

/* Ray::new  ray.rs:63-72 */
static inline Ray ray_new(V4 origin, V4 dir) {
  Ray r;
  r.rdir = vreciprocal(dir);
  origin.w = 1.0f;
  r.origin = origin;
  r.dir = dir;
  r.throughput = COL_WHITE;
  return r;
}
/* Ray::scale / rotate / translate  ray.rs:74-91 (Vec3::scale is a plain multiply) */
static inline void ray_scale(Ray* r, V4 s) {
  r->dir = vmul(r->dir, s);
  r->rdir = vreciprocal(r->dir);
  r->origin = pscale(r->origin, s);
}
static inline void ray_rotate(Ray* r, V4 q) {
  r->dir = vrotate(r->dir, q);
  r->rdir = vreciprocal(r->dir);
  r->origin = protate(r->origin, q);
  r->origin.w = 1.0f;
}
static inline void ray_translate(Ray* r, V4 t) { r->origin = add4(r->origin, t); }
/* Mul<Ray> for &Trs  trs.rs:275-284 */
static inline Ray trs_ray(const Trs* t, Ray r) {
  ray_scale(&r, t->scale);
  ray_rotate(&r, t->rotation);
  ray_translate(&r, t->translation);
  return r;
}
/* Mul<Ray> for &Inversed<&Trs>  trs.rs:405-414: translate(-T), rotate(R^-1), scale(1/S) */
static inline Ray inv_trs_ray(const Trs* t, Ray r) {
  ray_translate(&r, vneg(t->translation));
  ray_rotate(&r, qconj(t->rotation));
  ray_scale(&r, vreciprocal(t->scale));
  return r;
}

/* ============================================================================================ */
/* geometry tests                                                                                */
/* ============================================================================================ */
/* AABB::intersects  rayca-soft/src/bvh/aabb.rs:74-93 */
static inline float aabb_intersects(const AABB* bx, const Ray* ray) {
  V4 origin_vec = vec_from_point(ray->origin);
  V4 t1 = pscale(sub4(bx->a, origin_vec), ray->rdir);
  V4 t2 = pscale(sub4(bx->b, origin_vec), ray->rdir);
  V4 vmax = mul4(max4(t1, t2), v4(1.0f, 1.0f, 1.0f, FLT_MAX));
  V4 vmin = mul4(min4(t1, t2), v4(1.0f, 1.0f, 1.0f, -FLT_MAX));
  float tmax = fminf(fminf(vmax.x, vmax.y), fminf(vmax.z, vmax.w));
  float tmin = fmaxf(fmaxf(vmin.x, vmin.y), fmaxf(vmin.z, vmin.w));
  if (tmax >= tmin && tmax > 0.0f) return tmin;
  return FLT_MAX;
}
/* AABB::area  aabb.rs:20-23 */
static inline float aabb_area(const AABB* bx) {
  V4 e = vec3_simd(sub4(bx->b, bx->a));
  return e.x * e.y + e.y * e.z + e.z * e.x;
}
static inline void aabb_grow(AABB* bx, V4 p) { bx->a = min4(bx->a, p); bx->b = max4(bx->b, p); }

/* Triangle::intersects  rayca-geometry/src/triangle.rs:84-159 with v0,v1,v2 already = trs * vertex */
static inline int triangle_intersects_world(V4 w0, V4 w1, V4 w2, const Ray* ray, Hit* hit) {
  V4 v0 = vec_from_point(w0), v1 = vec_from_point(w1), vv2 = vec_from_point(w2);
  V4 v0v1 = vsub(v1, v0);
  V4 v0v2 = vsub(vv2, v0);
  V4 n = vcross(v0v1, v0v2);
  if (dot4(ray->dir, n) > 0.0f) return 0; /* back-face test :96 */
  float denom = dot4(n, n);
  float n_dot_ray_dir = dot4(n, ray->dir);
  if (fabsf(n_dot_ray_dir) < FLT_EPSILON) return 0; /* :106 */
  float d = -dot4(n, v0);
  float t = -(dot4(n, vec_from_point(ray->origin)) + d) / n_dot_ray_dir;
  if (t < 0.0f) return 0; /* :117 */
  V4 p = add4(ray->origin, vscale(ray->dir, t)); /* Point3 + Vec3 :122 */
  V4 edge0 = vsub(v1, v0);
  V4 vp0 = vec_from_point(sub4(p, v0));
  V4 c = vcross(edge0, vp0);
  if (dot4(n, c) < 0.0f) return 0;
  V4 edge1 = vsub(vv2, v1);
  V4 vp1 = vec_from_point(sub4(p, v1));
  c = vcross(edge1, vp1);
  float u = dot4(n, c);
  if (u < 0.0f) return 0;
  V4 edge2 = vsub(v0, vv2);
  V4 vp2 = vec_from_point(sub4(p, vv2));
  c = vcross(edge2, vp2);
  float v = dot4(n, c);
  if (v < 0.0f) return 0;
  hit->ray = *ray; /* ray.clone() :157 */
  hit->blas = RAYCA_NONE;
  hit->primitive = RAYCA_NONE;
  hit->depth = t;
  hit->point = p;
  hit->uv = v2(u / denom, v / denom);
  return 1;
}
static inline int triangle_intersects(const V4 p[3], const Trs* trs, const Ray* ray, Hit* hit) {
  /* get_vertex(i, trs) = trs * self.vertices[i]  triangle.rs:67-69,85-87 */
  return triangle_intersects_world(trs_point(trs, p[0]), trs_point(trs, p[1]), trs_point(trs, p[2]), ray, hit);
}

/* Sphere::intersects_impl  rayca-geometry/src/sphere.rs:101-140 (ray in model space) */
static inline int sphere_intersects_impl(V4 center, float radius2, const Ray* ray, Hit* hit) {
  float a = dot4(ray->dir, ray->dir);
  V4 c_to_r = vec3_simd(sub4(ray->origin, center)); /* Point3 - Point3 -> Vec3 */
  float b = dot4(c_to_r, ray->dir);
  float c = dot4(c_to_r, c_to_r) - radius2;
  float det = b * b - a * c;
  if (det < 0.0f) return 0;
  float det_sqrt = sqrtf(det);
  float t0 = (-b + det_sqrt) / a;
  float t1 = (-b - det_sqrt) / a;
  if (t0 < 0.0f && t1 < 0.0f) return 0;
  float t;
  if (t0 >= 0.0f && t1 >= 0.0f) t = fminf(t0, t1);
  else if (t0 >= 0.0f) t = t0;
  else t = t1;
  hit->ray = *ray;
  hit->blas = RAYCA_NONE;
  hit->primitive = RAYCA_NONE;
  hit->depth = t;
  hit->point = add4(ray->origin, vscale(ray->dir, t));
  hit->uv = v2(0.0f, 0.0f);
  return 1;
}
/* Sphere::intersects  sphere.rs:155-163 */
static inline int sphere_intersects(V4 center, float radius2, const Trs* trs, const Ray* ray, Hit* hit) {
  Ray inv = inv_trs_ray(trs, *ray);
  if (!sphere_intersects_impl(center, radius2, &inv, hit)) return 0;
  hit->point = trs_point(trs, hit->point);
  return 1;
}

/* ============================================================================================ */
/* BvhPrimitive helpers  rayca-soft/src/bvh/primitive.rs:71-101                                   */
/* ============================================================================================ */
/* Triangle::new  triangle.rs:58-63: centroid = (v0 + v1 + v2) * 0.3333, in model space */
static inline V4 tri_model_centroid(const V4 p[3]) {
  return vscale(vadd(vadd(vec_from_point(p[0]), vec_from_point(p[1])), vec_from_point(p[2])), 0.3333f);
}
static inline V4 tri_min(const V4 p[3], const Trs* t) { /* triangle.rs:165-170 */
  V4 m = point3(FLT_MAX, FLT_MAX, FLT_MAX);
  m = min4(m, trs_point(t, p[0])); m = min4(m, trs_point(t, p[1])); m = min4(m, trs_point(t, p[2]));
  return m;
}
static inline V4 tri_max(const V4 p[3], const Trs* t) { /* triangle.rs:172-177 */
  V4 m = point3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  m = max4(m, trs_point(t, p[0])); m = max4(m, trs_point(t, p[1])); m = max4(m, trs_point(t, p[2]));
  return m;
}
static inline float sphere_world_radius(const Prim* pr, const Trs* t) { /* sphere.rs:90-92 */
  return pr->radius * fmaxf(fmaxf(t->scale.x, t->scale.y), fmaxf(t->scale.z, t->scale.w));
}
static V4 prim_centroid(const OracleScene* s, const Prim* pr) { /* primitive.rs:71-77 */
  if (s->opts.xform == ORACLE_XFORM_CACHED) return pr->wcentroid;
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) return point_from_vec(trs_vec(t, pr->centroid));
  return trs_point(t, pr->center);
}
static V4 prim_min(const OracleScene* s, const Prim* pr) { /* primitive.rs:79-85 */
  if (s->opts.xform == ORACLE_XFORM_CACHED) return pr->wmin;
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) return tri_min(pr->p, t);
  float r = sphere_world_radius(pr, t); /* sphere.rs:170-173 */
  return sub4(trs_point(t, pr->center), vec3(r, r, r));
}
static V4 prim_max(const OracleScene* s, const Prim* pr) { /* primitive.rs:87-93 */
  if (s->opts.xform == ORACLE_XFORM_CACHED) return pr->wmax;
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) return tri_max(pr->p, t);
  float r = sphere_world_radius(pr, t); /* sphere.rs:175-178 */
  return add4(trs_point(t, pr->center), vec3(r, r, r));
}
/* AABB::grow_primitive  aabb.rs:62-72 (grow_triangle :30-34, grow_sphere :36-45) */
static void aabb_grow_primitive(const OracleScene* s, AABB* bx, const Prim* pr) {
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
    if (s->opts.xform == ORACLE_XFORM_CACHED) {
      aabb_grow(bx, pr->wp[0]); aabb_grow(bx, pr->wp[1]); aabb_grow(bx, pr->wp[2]);
    } else {
      aabb_grow(bx, trs_point(t, pr->p[0])); aabb_grow(bx, trs_point(t, pr->p[1])); aabb_grow(bx, trs_point(t, pr->p[2]));
    }
  } else {
    float r = sphere_world_radius(pr, t);
    V4 c = trs_point(t, pr->center);
    aabb_grow(bx, add4(c, vec3(-r, 0, 0))); aabb_grow(bx, add4(c, vec3(r, 0, 0)));
    aabb_grow(bx, add4(c, vec3(0, -r, 0))); aabb_grow(bx, add4(c, vec3(0, r, 0)));
    aabb_grow(bx, add4(c, vec3(0, 0, -r))); aabb_grow(bx, add4(c, vec3(0, 0, r)));
  }
}
/* BvhPrimitive::intersects  primitive.rs:95-101 */
static inline int prim_intersects(const OracleScene* s, const Prim* pr, const Ray* ray, Hit* hit) {
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
    if (s->opts.xform == ORACLE_XFORM_CACHED) return triangle_intersects_world(pr->wp[0], pr->wp[1], pr->wp[2], ray, hit);
    return triangle_intersects(pr->p, t, ray, hit);
  }
  return sphere_intersects(pr->center, pr->radius2, t, ray, hit);
}
static void prim_cache_world(const OracleScene* s, Prim* pr) {
  const Trs* t = &s->world_trs[pr->node];
  if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
    for (int i = 0; i < 3; ++i) pr->wp[i] = trs_point(t, pr->p[i]);
    pr->wcentroid = point_from_vec(trs_vec(t, pr->centroid));
    pr->wmin = tri_min(pr->p, t);
    pr->wmax = tri_max(pr->p, t);
  } else {
    float r = sphere_world_radius(pr, t);
    pr->wcentroid = trs_point(t, pr->center);
    pr->wmin = sub4(trs_point(t, pr->center), vec3(r, r, r));
    pr->wmax = add4(trs_point(t, pr->center), vec3(r, r, r));
    pr->wp[0] = pr->wp[1] = pr->wp[2] = point3(0, 0, 0);
  }
}

/* ============================================================================================ */
/* BLAS build  rayca-soft/src/bvh/blas.rs                                                         */
/* ============================================================================================ */
static inline float axis_of(V4 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

/* BvhNode::new  blas.rs:27-36 + AABB::grow_range aabb.rs:47-60 */
static BNode bnode_new(const OracleScene* s, const Blas* bl, uint32_t offset, uint32_t count) {
  BNode n;
  n.bounds.a = point3(FLT_MAX, FLT_MAX, FLT_MAX);
  n.bounds.b = point3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (uint32_t i = offset; i < offset + count; ++i) {
    n.bounds.a = min4(n.bounds.a, prim_min(s, &bl->prims[i]));
    n.bounds.b = max4(n.bounds.b, prim_max(s, &bl->prims[i]));
  }
  n.offset = offset;
  n.count = count;
  return n;
}
/* BvhNode::evaluate_sah  blas.rs:64-89 */
static float evaluate_sah(const OracleScene* s, const Blas* bl, const BNode* node, int axis, float pos) {
  AABB left_box = {point3(0, 0, 0), point3(0, 0, 0)}; /* AABB::default(): both corners at origin */
  AABB right_box = left_box;
  uint32_t left_count = 0, right_count = 0;
  for (uint32_t i = node->offset; i < node->offset + node->count; ++i) {
    const Prim* pr = &bl->prims[i];
    V4 c = prim_centroid(s, pr);
    if (axis_of(c, axis) < pos) { left_count++; aabb_grow_primitive(s, &left_box, pr); }
    else { right_count++; aabb_grow_primitive(s, &right_box, pr); }
  }
  float cost = (float)left_count * aabb_area(&left_box) + (float)right_count * aabb_area(&right_box);
  return cost > 0.0f ? cost : FLT_MAX;
}
/* find_best_split_plane  blas.rs:93-123, literal */
static void find_best_split_literal(const OracleScene* s, const Blas* bl, const BNode* node, int* best_axis, float* split_pos, float* best_cost) {
  *best_cost = FLT_MAX; *best_axis = 0; *split_pos = 0.0f;
  for (int axis = 0; axis < 3; ++axis) {
    float bounds_min = axis_of(node->bounds.a, axis), bounds_max = axis_of(node->bounds.b, axis);
    if (bounds_min == bounds_max) continue;
    float scale = (bounds_max - bounds_min) / (float)64;
    for (int i = 1; i < 64; ++i) {
      float candidate_pos = bounds_min + (float)i * scale;
      float cost = evaluate_sah(s, bl, node, axis, candidate_pos);
      if (cost < *best_cost) { *best_cost = cost; *best_axis = axis; *split_pos = candidate_pos; }
    }
  }
}
/* same values from one sweep per axis: prim p is on the left of plane i iff centroid < pos_i;
 * pos_i is non-decreasing in i, so p is left for exactly the planes i > k(p) with
 * k(p) = #{i : !(centroid < pos_i)}.  min/max unions are exact, so every box and count (hence
 * every cost) equals the literal evaluation. */
static void find_best_split_binned(const OracleScene* s, const Blas* bl, const BNode* node, int* best_axis, float* split_pos, float* best_cost) {
  *best_cost = FLT_MAX; *best_axis = 0; *split_pos = 0.0f;
  for (int axis = 0; axis < 3; ++axis) {
    float bounds_min = axis_of(node->bounds.a, axis), bounds_max = axis_of(node->bounds.b, axis);
    if (bounds_min == bounds_max) continue;
    float scale = (bounds_max - bounds_min) / (float)64;
    float pos[64];
    for (int i = 1; i < 64; ++i) pos[i] = bounds_min + (float)i * scale;
    AABB bin_box[64];
    uint32_t bin_cnt[64];
    for (int b = 0; b < 64; ++b) { bin_box[b].a = point3(FLT_MAX, FLT_MAX, FLT_MAX); bin_box[b].b = point3(-FLT_MAX, -FLT_MAX, -FLT_MAX); bin_cnt[b] = 0; }
    for (uint32_t i = node->offset; i < node->offset + node->count; ++i) {
      const Prim* pr = &bl->prims[i];
      float c = axis_of(prim_centroid(s, pr), axis);
      int k = 0;
      for (int j = 1; j < 64; ++j) if (!(c < pos[j])) k++;  /* not assumed contiguous */
      /* k(p) as a count is only a valid bin if the predicate is monotone in j; verify */
      int mono = 1;
      for (int j = 1; j < 64; ++j) { int left = c < pos[j]; if (left != (j > k)) { mono = 0; break; } }
      if (!mono) { /* cannot happen for non-decreasing pos; fall back to the literal evaluation */
        find_best_split_literal(s, bl, node, best_axis, split_pos, best_cost);
        return;
      }
      bin_cnt[k]++;
      AABB* bb = &bin_box[k];
      if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) { aabb_grow(bb, pr->wp[0]); aabb_grow(bb, pr->wp[1]); aabb_grow(bb, pr->wp[2]); }
      else { AABB tmp = *bb; aabb_grow_primitive(s, &tmp, pr); *bb = tmp; }
    }
    /* prefix (left) and suffix (right) unions, each seeded with the origin box */
    AABB lbox[65], rbox[65];
    uint32_t lcnt[65], rcnt[65];
    AABB origin = {point3(0, 0, 0), point3(0, 0, 0)};
    lbox[0] = origin; lcnt[0] = 0;
    for (int b = 0; b < 64; ++b) {
      lbox[b + 1] = lbox[b]; lcnt[b + 1] = lcnt[b] + bin_cnt[b];
      if (bin_cnt[b]) { lbox[b + 1].a = min4(lbox[b + 1].a, bin_box[b].a); lbox[b + 1].b = max4(lbox[b + 1].b, bin_box[b].b); }
    }
    rbox[64] = origin; rcnt[64] = 0;
    for (int b = 63; b >= 0; --b) {
      rbox[b] = rbox[b + 1]; rcnt[b] = rcnt[b + 1] + bin_cnt[b];
      if (bin_cnt[b]) { rbox[b].a = min4(rbox[b].a, bin_box[b].a); rbox[b].b = max4(rbox[b].b, bin_box[b].b); }
    }
    for (int i = 1; i < 64; ++i) {
      /* left = bins 0..i-1, right = bins i..63 */
      float cost = (float)lcnt[i] * aabb_area(&lbox[i]) + (float)rcnt[i] * aabb_area(&rbox[i]);
      cost = cost > 0.0f ? cost : FLT_MAX;
      if (cost < *best_cost) { *best_cost = cost; *best_axis = axis; *split_pos = pos[i]; }
    }
  }
}
static void blas_push_node(Blas* bl, BNode n) {
  if (bl->node_count == bl->node_cap) {
    bl->node_cap = bl->node_cap ? bl->node_cap * 2 : 64;
    bl->nodes = (BNode*)realloc(bl->nodes, sizeof(BNode) * bl->node_cap);
  }
  bl->nodes[bl->node_count++] = n;
}
/* Blas::set_primitives_recursive  blas.rs:261-316 */
static void blas_split(const OracleScene* s, Blas* bl, uint32_t node_index, uint32_t level) {
  if (level >= bl->max_depth) return;
  BNode node = bl->nodes[node_index];
  int split_axis; float split_pos, split_cost;
  if (s->opts.build == ORACLE_BUILD_BINNED) find_best_split_binned(s, bl, &node, &split_axis, &split_pos, &split_cost);
  else find_best_split_literal(s, bl, &node, &split_axis, &split_pos, &split_cost);
  float no_split_cost = (float)node.count * aabb_area(&node.bounds); /* calculate_cost :125-127 */
  if (split_cost > no_split_cost) return;
  uint32_t i_tri = node.offset, j_tri = node.offset + node.count;
  while (i_tri < j_tri) {
    V4 c = prim_centroid(s, &bl->prims[i_tri]);
    if (axis_of(c, split_axis) < split_pos) i_tri++;
    else { Prim tmp = bl->prims[i_tri]; bl->prims[i_tri] = bl->prims[j_tri - 1]; bl->prims[j_tri - 1] = tmp; j_tri--; }
  }
  uint32_t left_count = i_tri - node.offset;
  uint32_t right_count = node.count - left_count;
  if (left_count > 0 && right_count > 0) {
    uint32_t left_index = bl->node_count;
    BNode lc = bnode_new(s, bl, node.offset, left_count);
    BNode rc = bnode_new(s, bl, node.offset + left_count, right_count);
    node.offset = left_index; /* set_left_child_index :47-50 */
    node.count = 0;
    blas_push_node(bl, lc);
    blas_push_node(bl, rc);
    blas_split(s, bl, left_index, level + 1);
    blas_split(s, bl, left_index + 1, level + 1);
  }
  bl->nodes[node_index] = node;
}
/* Blas::set_primitives  blas.rs:245-259 */
static void blas_build(const OracleScene* s, Blas* bl) {
  bl->node_count = 0;
  blas_push_node(bl, bnode_new(s, bl, 0, bl->prim_count));
  BNode dummy;
  memset(&dummy, 0, sizeof dummy);
  dummy.bounds.a = point3(0, 0, 0); dummy.bounds.b = point3(0, 0, 0);
  blas_push_node(bl, dummy);
  if (bl->prim_count > 0) blas_split(s, bl, 0, 0);
}

/* ============================================================================================ */
/* TLAS build  rayca-soft/src/bvh/tlas.rs:74-134                                                  */
/* ============================================================================================ */
static int32_t tlas_push(OracleScene* s, TNode n) {
  if (s->tnode_count == s->tnode_cap) {
    s->tnode_cap = s->tnode_cap ? s->tnode_cap * 2 : 16;
    s->tnodes = (TNode*)realloc(s->tnodes, sizeof(TNode) * s->tnode_cap);
  }
  s->tnodes[s->tnode_count] = n;
  return (int32_t)s->tnode_count++;
}
static TNode tnode_default(void) {
  TNode n;
  n.bounds.a = point3(0, 0, 0); n.bounds.b = point3(0, 0, 0);
  n.left = n.right = -1; n.blas_offset = n.blas_count = 0;
  return n;
}
static void tlas_replace_models_recursive(OracleScene* s, TNode* self, uint32_t offset, uint32_t count) {
  self->blas_offset = offset; self->blas_count = count;
  self->bounds.a = point3(FLT_MAX, FLT_MAX, FLT_MAX);
  self->bounds.b = point3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (uint32_t i = offset; i < offset + count; ++i) {
    const Blas* bl = &s->blass[s->blas_nodes[i].blas];
    self->bounds.a = min4(self->bounds.a, bl->nodes[0].bounds.a);
    self->bounds.b = max4(self->bounds.b, bl->nodes[0].bounds.b);
  }
  V4 extent = vec3_simd(sub4(self->bounds.b, self->bounds.a));
  int axis = 0;
  if (extent.y > extent.x) axis = 1;
  if (extent.z > axis_of(extent, axis)) axis = 2;
  float split_pos = axis_of(self->bounds.a, axis) + axis_of(extent, axis) * 0.5f;
  uint32_t i = offset, j = offset + count;
  while (i < j) {
    const Blas* bl = &s->blass[s->blas_nodes[i].blas];
    /* AABB::get_centroid = (b - a) / 2.0 -- the half-extent, not the centre  aabb.rs:95-97 */
    V4 cen = point_from_vec(vdivf(vec3_simd(sub4(bl->nodes[0].bounds.b, bl->nodes[0].bounds.a)), 2.0f));
    if (axis_of(cen, axis) < split_pos) i++;
    else { BlasNode t = s->blas_nodes[i]; s->blas_nodes[i] = s->blas_nodes[j - 1]; s->blas_nodes[j - 1] = t; j--; }
  }
  uint32_t left_count = i - offset, right_count = count - left_count;
  if (left_count > 0 && right_count > 0) {
    TNode l = tnode_default();
    tlas_replace_models_recursive(s, &l, offset, left_count);
    TNode r = tnode_default();
    tlas_replace_models_recursive(s, &r, offset + left_count, right_count);
    self->left = tlas_push(s, l);
    self->right = tlas_push(s, r);
    self->blas_count = 0;
  }
}

/* ============================================================================================ */
/* traversal                                                                                     */
/* ============================================================================================ */
typedef struct { uint64_t boxes, tris; } TraceCount;

/* BvhNode::intersects  blas.rs:129-177 */
static int bnode_intersects(const OracleScene* s, const Blas* bl, const BNode* self, const Ray* ray, Hit* out, TraceCount* tc) {
  tc->boxes++;
  float d = aabb_intersects(&self->bounds, ray);
  if (d == FLT_MAX) return 0;
  int have = 0;
  float depth = INFINITY;
  if (self->count != 0) {
    tc->tris += self->count;
    for (uint32_t pri = self->offset; pri < self->offset + self->count; ++pri) {
      Hit h;
      if (prim_intersects(s, &bl->prims[pri], ray, &h)) {
        if (h.depth < depth) { depth = h.depth; h.primitive = pri; *out = h; have = 1; }
      }
    }
  } else {
    Hit h;
    if (bnode_intersects(s, bl, &bl->nodes[self->offset], ray, &h, tc)) {
      if (h.depth < depth) { depth = h.depth; *out = h; have = 1; }
    }
    if (bnode_intersects(s, bl, &bl->nodes[self->offset + 1], ray, &h, tc)) {
      if (h.depth < depth) { *out = h; have = 1; }
    }
  }
  return have;
}
/* TlasNode::intersects  tlas.rs:136-180 */
static int tnode_intersects(const OracleScene* s, const TNode* self, const Ray* ray, Hit* out, TraceCount* tc) {
  tc->boxes++;
  if (aabb_intersects(&self->bounds, ray) == FLT_MAX) return 0;
  int have = 0;
  float depth = INFINITY;
  if (self->left < 0 && self->right < 0) {
    for (uint32_t i = self->blas_offset; i < self->blas_offset + self->blas_count; ++i) {
      const Blas* bl = &s->blass[s->blas_nodes[i].blas];
      Hit h;
      if (bnode_intersects(s, bl, &bl->nodes[0], ray, &h, tc)) {
        if (h.depth < depth) { depth = h.depth; h.blas = i; *out = h; have = 1; }
      }
    }
  } else {
    Hit h;
    if (self->left >= 0 && tnode_intersects(s, &s->tnodes[self->left], ray, &h, tc)) {
      if (h.depth < depth) { depth = h.depth; *out = h; have = 1; }
    }
    if (self->right >= 0 && tnode_intersects(s, &s->tnodes[self->right], ray, &h, tc)) {
      if (h.depth < depth) { *out = h; have = 1; }
    }
  }
  return have;
}

/* ============================================================================================ */
/* HitInfo  rayca-soft/src/hit.rs                                                                 */
/* ============================================================================================ */
typedef struct {
  const OracleScene* scene;
  Hit hit;
  const Prim* primitive;
  int has_color, has_normal, has_uv, has_reflection, has_next_origin, has_diffuse;
  Col color, diffuse;
  V4 normal, reflection, next_ray_origin;
  V2 uv;
} HitInfo;

typedef struct { /* per-thread render context */
  const OracleScene* scene;
  const RaycaConfig* cfg;
  TraceCount tc;
  uint64_t rays_shadow, rays_bounce, hits_shaded;
  int unsupported; /* set when a todo!()/unimplemented!() arm of the reference is reached */
} Ctx;

/* Tlas::intersects  tlas.rs:271-275 */
static int tlas_intersects(Ctx* cx, Ray ray, HitInfo* hi) {
  Hit h;
  if (!tnode_intersects(cx->scene, &cx->scene->root, &ray, &h, &cx->tc)) return 0;
  memset(hi, 0, sizeof *hi);
  hi->scene = cx->scene;
  hi->hit = h;
  return 1;
}
/* Tlas::get_primitive  tlas.rs:282-285 */
static const Prim* hi_primitive(HitInfo* hi) {
  if (!hi->primitive) {
    const Blas* bl = &hi->scene->blass[hi->scene->blas_nodes[hi->hit.blas].blas];
    hi->primitive = &bl->prims[hi->hit.primitive];
  }
  return hi->primitive;
}
static const RaycaMaterial MATERIAL_DEFAULT = { /* Material::DEFAULT = Pbr(NONE) -> PbrMaterial::WHITE  material/pbr.rs:69-76 */
  RAYCA_MATERIAL_PBR, RAYCA_NONE, RAYCA_NONE, RAYCA_NONE, {1, 1, 1, 1}, 0.0f, 1.0f, 0.0f, 0.0f,
  {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0, 1}};
/* BvhPrimitive::get_material  primitive.rs:103-110 */
static const RaycaMaterial* prim_material(const OracleScene* s, const Prim* pr) {
  if (pr->material == RAYCA_NONE || pr->material >= s->material_count) return &MATERIAL_DEFAULT;
  return &s->materials[pr->material];
}
static inline Col col4(const float c[4]) { return col(c[0], c[1], c[2], c[3]); }

/* Sampler::sample  rayca-model/src/sampler.rs:11-30 */
static inline uint32_t f32_as_u32(float v) { /* Rust `as u32`: saturating, NaN -> 0 */
  if (!(v == v)) return 0;
  if (v <= 0.0f) return 0;
  if (v >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)v;
}
uint32_t oracle_sampler_wrap(float u, uint32_t size) {
  float x = (u - floorf(u) + 1.0f) * (float)size;
  return f32_as_u32(x) % size;
}
static Col sample_texture(const OracleScene* s, uint32_t texture, V2 uv) {
  const RaycaImage* im = &s->images[s->textures[texture].image];
  uint32_t x = oracle_sampler_wrap(uv.x, im->width), y = oracle_sampler_wrap(uv.y, im->height);
  size_t idx = (size_t)y * im->width + x;
  const uint8_t* base = s->image_bytes + im->byte_offset;
  if (im->color_type == RAYCA_COLOR_RGBA32F) { /* From<RGBA32F> for Color divides by 255  color/mod.rs:222-231 */
    const float* f = (const float*)base + idx * 4;
    return col(f[0] / 255.0f, f[1] / 255.0f, f[2] / 255.0f, f[3] / 255.0f);
  }
  if (im->color_type == RAYCA_COLOR_RGBA8) { /* color/mod.rs:211-220 */
    const uint8_t* p = base + idx * 4;
    return col((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
  }
  const uint8_t* p = base + idx * 3; /* RGB8 -> RGBA8(a=255) -> Color */
  return col((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, 255.0f / 255.0f);
}
static int has_texture(const OracleScene* s, uint32_t t) {
  return t != RAYCA_NONE && t < s->texture_count && s->textures[t].image < s->image_count;
}
/* PbrMaterial::get_color  material/pbr.rs:94-102 */
static Col pbr_get_color(const OracleScene* s, const RaycaMaterial* m, V2 uv) {
  if (has_texture(s, m->albedo_texture)) return cmul(col4(m->color), sample_texture(s, m->albedo_texture, uv));
  return col4(m->color);
}
/* PbrMaterial::get_metallic_roughness  material/pbr.rs:125-137: (color.b, color.r) */
static void pbr_metallic_roughness(const OracleScene* s, const RaycaMaterial* m, V2 uv, float* metallic, float* roughness) {
  if (has_texture(s, m->metallic_roughness_texture)) {
    Col c = sample_texture(s, m->metallic_roughness_texture, uv);
    *metallic = c.b; *roughness = c.r;
  } else { *metallic = m->metallic_factor; *roughness = m->roughness_factor; }
}
/* Material::get_color  material/mod.rs:107-113 */
static Col material_get_color(const OracleScene* s, const RaycaMaterial* m, V2 uv) {
  if (m->kind == RAYCA_MATERIAL_PBR) return pbr_get_color(s, m, uv);
  if (m->kind == RAYCA_MATERIAL_PHONG) return cadd(col4(m->ambient), col4(m->emission)); /* phong.rs:62-64 */
  return col4(m->diffuse);
}
/* Material::get_diffuse  material/mod.rs:116-122 */
static Col material_get_diffuse(const OracleScene* s, const RaycaMaterial* m, V2 uv) {
  if (m->kind == RAYCA_MATERIAL_PBR) return pbr_get_color(s, m, uv);
  return col4(m->diffuse);
}
/* is_emissive  primitive.rs:122-130, phong.rs:54-56 */
static int material_is_emissive(const RaycaMaterial* m) {
  if (m->kind == RAYCA_MATERIAL_PHONG) return !cclose(col4(m->emission), COL_BLACK);
  return 0;
}
static Col material_get_emission(const RaycaMaterial* m) {
  if (m->kind == RAYCA_MATERIAL_PHONG) return col4(m->emission);
  return COL_BLACK;
}
/* get_t  material/mod.rs:165-171; phong.rs:67-74; ggx.rs:53-60.  reduce_avg = reduce_sum / 3.0 */
static float material_get_t(Ctx* cx, const RaycaMaterial* m) {
  if (m->kind == RAYCA_MATERIAL_PBR) { cx->unsupported = 1; return 0.0f; } /* todo!() */
  float kd_avg = reduce_sum4(vec3(m->diffuse[0], m->diffuse[1], m->diffuse[2])) / 3.0f;
  float ks_avg = reduce_sum4(vec3(m->specular[0], m->specular[1], m->specular[2])) / 3.0f;
  if (ks_avg == 0.0f && kd_avg == 0.0f) return 1.0f;
  float t = ks_avg / (ks_avg + kd_avg);
  if (m->kind == RAYCA_MATERIAL_GGX) return fmaxf(t, 0.25f);
  return t;
}

/* BvhTriangle interpolation  bvh/triangle.rs:33-77: weights (1-u-v) on vertex 2, u on 0, v on 1 */
static inline float w2_of(V2 uv) { return 1.0f - uv.x - uv.y; }
static Col geom_color(const Prim* pr, const Hit* h) {
  if (pr->kind != RAYCA_GEOMETRY_TRIANGLE_MESH) return COL_WHITE;
  return cadd(cadd(cmulf(pr->ext[2].color, w2_of(h->uv)), cmulf(pr->ext[0].color, h->uv.x)), cmulf(pr->ext[1].color, h->uv.y));
}
static V2 geom_uv(const Prim* pr, const Hit* h) {
  if (pr->kind != RAYCA_GEOMETRY_TRIANGLE_MESH) return v2(0, 0);
  return v2add(v2add(v2scale(pr->ext[2].uv, w2_of(h->uv)), v2scale(pr->ext[0].uv, h->uv.x)), v2scale(pr->ext[1].uv, h->uv.y));
}
static V4 interp_vec(V4 a2, V4 a0, V4 a1, V2 uv) {
  return vadd(vadd(vscale(a2, w2_of(uv)), vscale(a0, uv.x)), vscale(a1, uv.y));
}
static V2 hi_uv(HitInfo* hi) { /* hit.rs:85-91 */
  if (!hi->has_uv) { hi->uv = geom_uv(hi_primitive(hi), &hi->hit); hi->has_uv = 1; }
  return hi->uv;
}
static Col hi_color(HitInfo* hi) { /* hit.rs:65-71 -> primitive.rs:142-148 */
  if (!hi->has_color) {
    const Prim* pr = hi_primitive(hi);
    Col gc = geom_color(pr, &hi->hit);
    V2 uv = geom_uv(pr, &hi->hit);
    Col mc = material_get_color(hi->scene, prim_material(hi->scene, pr), uv);
    hi->color = cmul(gc, mc);
    hi->has_color = 1;
  }
  return hi->color;
}
static V4 hi_normal(HitInfo* hi) { /* hit.rs:77-83 -> primitive.rs:172-192 */
  if (!hi->has_normal) {
    const Prim* pr = hi_primitive(hi);
    const OracleScene* s = hi->scene;
    if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
      V2 uv = geom_uv(pr, &hi->hit);
      V4 normal = vnormalize(interp_vec(pr->ext[2].normal, pr->ext[0].normal, pr->ext[1].normal, hi->hit.uv));
      V4 tangent = vnormalize(interp_vec(pr->ext[2].tangent, pr->ext[0].tangent, pr->ext[1].tangent, hi->hit.uv));
      V4 bitangent = vnormalize(interp_vec(pr->ext[2].bitangent, pr->ext[0].bitangent, pr->ext[1].bitangent, hi->hit.uv));
      const RaycaMaterial* m = prim_material(s, pr);
      if (m->kind == RAYCA_MATERIAL_PBR && has_texture(s, m->normal_texture)) { /* pbr.rs:104-123 */
        V4 sn = vec_from_col(sample_texture(s, m->normal_texture, uv));
        sn = sub4(vscale(sn, 2.0f), splat4(1.0f)); /* Sub<f32> for Vec3 touches w too; w is unused below */
        M3 tbn = m3_tbn(tangent, bitangent, normal);
        hi->normal = vnormalize(m3_vec(&tbn, sn));
      } else hi->normal = normal;
    } else {
      const Trs* t = &s->world_trs[pr->node];
      V4 hit_point = inv_trs_point(t, hi->hit.point);
      V4 normal = vnormalize(vec3_simd(sub4(hit_point, pr->center))); /* sphere.rs:95-97 */
      M3 inv = m3_from_inv_trs(t);
      M3 nm = m3_transpose(&inv);
      hi->normal = vnormalize(m3_vec(&nm, normal));
    }
    hi->has_normal = 1;
  }
  return hi->normal;
}
static V4 hi_view(const HitInfo* hi) { return vneg(hi->hit.ray.dir); } /* ray.rs:149-151 */
static V4 hi_reflection(HitInfo* hi) { /* hit.rs:93-101 */
  if (!hi->has_reflection) { hi->reflection = vnormalize(vreflect(hi->hit.ray.dir, hi_normal(hi))); hi->has_reflection = 1; }
  return hi->reflection;
}
static Col hi_diffuse(HitInfo* hi) { /* hit.rs:130-137 -> primitive.rs:150-155 */
  if (!hi->has_diffuse) {
    const Prim* pr = hi_primitive(hi);
    V2 uv = hi_uv(hi);
    hi->diffuse = cmul(geom_color(pr, &hi->hit), material_get_diffuse(hi->scene, prim_material(hi->scene, pr), uv));
    hi->has_diffuse = 1;
  }
  return hi->diffuse;
}
static Col hi_specular(HitInfo* hi) { /* hit.rs:139-142 -> material/mod.rs:141-151 */
  const OracleScene* s = hi->scene;
  const RaycaMaterial* m = prim_material(s, hi_primitive(hi));
  if (m->kind == RAYCA_MATERIAL_PBR) {
    V2 uv = hi_uv(hi);
    Col base = pbr_get_color(s, m, uv);
    float metallic, roughness;
    pbr_metallic_roughness(s, m, uv, &metallic, &roughness);
    return fmulc(metallic, base);
  }
  return col4(m->specular);
}
static float hi_shininess(Ctx* cx, HitInfo* hi) { /* material/mod.rs:153-159 */
  const RaycaMaterial* m = prim_material(hi->scene, hi_primitive(hi));
  if (m->kind != RAYCA_MATERIAL_PHONG) { cx->unsupported = 1; return 0.0f; }
  return m->shininess;
}
static float hi_roughness(HitInfo* hi) { /* material/mod.rs:173-185 */
  const RaycaMaterial* m = prim_material(hi->scene, hi_primitive(hi));
  if (m->kind == RAYCA_MATERIAL_PHONG) return clampf(sqrtf(2.0f / (m->shininess + 2.0f)), 0.0f, 1.0f);
  if (m->kind == RAYCA_MATERIAL_PBR) { float me, ro; pbr_metallic_roughness(hi->scene, m, hi_uv(hi), &me, &ro); return ro; }
  return m->roughness_factor; /* Ggx: roughness stored in roughness_factor */
}
static V4 hi_next_ray_origin(HitInfo* hi) { /* hit.rs:164-171 */
  if (!hi->has_next_origin) { hi->next_ray_origin = add4(hi->hit.point, vscale(hi_normal(hi), ORC_RAY_BIAS)); hi->has_next_origin = 1; }
  return hi->next_ray_origin;
}
static Ray hi_next_ray(HitInfo* hi, V4 dir) { return ray_new(hi_next_ray_origin(hi), dir); } /* hit.rs:173-176 */
static int hi_is_emissive(HitInfo* hi) { return material_is_emissive(prim_material(hi->scene, hi_primitive(hi))); }
static Col hi_emission(HitInfo* hi) { return material_get_emission(prim_material(hi->scene, hi_primitive(hi))); }

/* ============================================================================================ */
/* BRDFs  rayca-soft/src/brdf/{ggx,lambertian}.rs                                                 */
/* ============================================================================================ */
#define F_PI 3.14159274101257324219f       /* std::f32::consts::PI */
#define F_1_PI 0.318309873342514038086f    /* FRAC_1_PI */
#define F_2_PI 0.636619746685028076172f    /* FRAC_2_PI */

static float ggx_get_d(float a, V4 h, V4 n) { /* ggx.rs:58-67 */
  float a_squared = a * a;
  float cos_theta = clampf(dot4(h, n), 0.0f, 1.0f);
  float theta = acosf(cos_theta);
  float denominator = powf(cos_theta, 4.0f) * powf(a_squared + powf(tanf(theta), 2.0f), 2.0f);
  if (denominator == 0.0f) return 0.0f;
  return a_squared * F_1_PI / denominator;
}
static float ggx_get_g1(float a, V4 omega, V4 n) { /* ggx.rs:74-83 */
  float cos_theta = dot4(omega, n);
  if (cos_theta <= 0.0f) return 0.0f;
  float theta = acosf(cos_theta);
  float denominator = 1.0f + sqrtf(1.0f + a * a * powf(tanf(theta), 2.0f));
  return 2.0f / denominator;
}
static Col ggx_get_f(Col ks, V4 omega_i, V4 h) { /* ggx.rs:100-103 */
  float omega_i_dot_h = fabsf(dot4(omega_i, h));
  return cadd(ks, cmulf(csub(COL_WHITE, ks), powf(1.0f - omega_i_dot_h, 5.0f)));
}
static Col ggx_get_bsdf(HitInfo* hi, V4 omega_i) { /* ggx.rs:106-124 */
  V4 omega_o = hi_view(hi);
  V4 n = hi_normal(hi);
  float omega_i_dot_n = clampf(dot4(omega_i, n), 0.0f, 1.0f);
  float omega_o_dot_n = clampf(dot4(omega_o, n), 0.0f, 1.0f);
  if (omega_i_dot_n == 0.0f || omega_o_dot_n == 0.0f) return COL_BLACK;
  Col ks = hi_specular(hi);
  float a = hi_roughness(hi);
  V4 h = vnormalize(vadd(omega_i, omega_o));
  Col f = ggx_get_f(ks, omega_i, h);
  float g = ggx_get_g1(a, omega_i, n) * ggx_get_g1(a, omega_o, n); /* get_g :90-92 */
  float d = ggx_get_d(a, h, n);
  float denominator = 4.0f * omega_i_dot_n * omega_o_dot_n;
  return cdivf(cmulf(cmulf(f, g), d), denominator);
}
static Col ggx_get_brdf(HitInfo* hi, V4 omega_i) { /* ggx.rs:126-129 */
  Col kd = hi_diffuse(hi);
  return cadd(cmulf(kd, F_1_PI), ggx_get_bsdf(hi, omega_i));
}
static float ggx_get_pdf(Ctx* cx, HitInfo* hi, V4 omega_i) { /* ggx.rs:131-146 */
  V4 omega_o = hi_view(hi);
  V4 h = vnormalize(vadd(omega_o, omega_i));
  float h_dot_omega_i = clampf(dot4(h, omega_i), 0.0f, 1.0f);
  if (h_dot_omega_i == 0.0f) return 0.0f;
  V4 n = hi_normal(hi);
  float n_dot_h = clampf(dot4(n, h), 0.0f, 1.0f);
  float spec = ggx_get_d(hi_roughness(hi), h, n) * n_dot_h / (4.0f * h_dot_omega_i);
  float dif = clampf(dot4(n, omega_i), 0.0f, 1.0f) * F_1_PI;
  float t = material_get_t(cx, prim_material(hi->scene, hi_primitive(hi)));
  return (1.0f - t) * dif + t * spec;
}
/* rotate a local sample s around w: shared tail of cosine.rs:77-87, hemisphere.rs:29-39,
 * ggx.rs:168-178, lambertian.rs:62-71 */
static V4 orient_sample(V4 s, V4 w, int normalize_v) {
  V4 a = vclose(w, vec3(0, 1, 0)) ? vec3(1, 0, 0) : vec3(0, 1, 0);
  V4 u = vnormalize(vcross(a, w));
  V4 v = vcross(w, u);
  if (normalize_v) v = vnormalize(v);
  return vadd(vadd(vscale(u, s.x), vscale(v, s.y)), vscale(w, s.z));
}
static V4 spherical(float theta, float omega) {
  return vec3(cosf(omega) * sinf(theta), sinf(omega) * sinf(theta), cosf(theta));
}
static V4 ggx_get_random_dir(Ctx* cx, HitInfo* hi, Rng* rng) { /* ggx.rs:148-187 */
  float e0 = rng_next(rng), e1 = rng_next(rng), e2 = rng_next(rng);
  float t = material_get_t(cx, prim_material(hi->scene, hi_primitive(hi)));
  float a = hi_roughness(hi);
  float theta = (e0 <= t) ? atanf((a * sqrtf(e1)) / sqrtf(1.0f - e1)) : acosf(clampf(sqrtf(e1), -1.0f, 1.0f));
  float omega = 2.0f * F_PI * e2;
  V4 s = orient_sample(spherical(theta, omega), hi_normal(hi), 0);
  if (e0 <= t) return vreflect(vneg(hi_view(hi)), s);
  return s;
}
static Col ggx_get_specular_component(HitInfo* hi, V4 omega_i) { /* ggx.rs:189-203 */
  V4 omega_o = hi_view(hi);
  V4 n = hi_normal(hi);
  float omega_o_dot_n = clampf(dot4(omega_o, n), 0.0f, 1.0f);
  V4 h = vnormalize(vadd(omega_i, omega_o));
  float n_dot_h = clampf(dot4(n, h), 0.0f, 1.0f);
  if (omega_o_dot_n == 0.0f || n_dot_h == 0.0f) return COL_BLACK;
  Col ks = hi_specular(hi);
  float a = hi_roughness(hi);
  float h_dot_omega_i = clampf(dot4(h, omega_i), 0.0f, 1.0f);
  float g = ggx_get_g1(a, omega_i, n) * ggx_get_g1(a, omega_o, n);
  return cdivf(cmulf(cmulf(ggx_get_f(ks, omega_i, h), g), h_dot_omega_i), omega_o_dot_n * n_dot_h);
}
static Col lam_get_brdf(Ctx* cx, HitInfo* hi, V4 omega_i) { /* lambertian.rs:7-16 */
  Col lambertian = cmulf(hi_diffuse(hi), F_1_PI);
  float s = hi_shininess(cx, hi);
  Col specular = cdivf(cmulf(cmulf(cmulf(hi_specular(hi), s + 2.0f), powf(dot4(hi_reflection(hi), omega_i), s)), F_1_PI), 2.0f);
  return cadd(lambertian, specular);
}
static float lam_get_pdf(Ctx* cx, HitInfo* hi, V4 omega) { /* lambertian.rs:18-27 */
  float r_dot_omega = clampf(dot4(hi_reflection(hi), omega), 0.0f, 1.0f);
  float s = hi_shininess(cx, hi);
  float spec = (s + 1.0f) * F_2_PI * powf(r_dot_omega, s);
  float diff = clampf(dot4(hi_normal(hi), omega), 0.0f, 1.0f) * F_1_PI;
  float t = material_get_t(cx, prim_material(hi->scene, hi_primitive(hi)));
  return (1.0f - t) * diff + t * spec;
}
static V4 lam_get_random_dir(Ctx* cx, HitInfo* hi, Rng* rng) { /* lambertian.rs:32-72 */
  float e0 = rng_next(rng), e1 = rng_next(rng), e2 = rng_next(rng);
  float t = material_get_t(cx, prim_material(hi->scene, hi_primitive(hi)));
  float s = hi_shininess(cx, hi);
  float theta = (e0 <= t) ? acosf(clampf(powf(e1, 1.0f / (s + 1.0f)), -1.0f, 1.0f)) : acosf(clampf(sqrtf(e1), -1.0f, 1.0f));
  float omega = 2.0f * F_PI * e2;
  V4 w = (e0 <= t) ? hi_reflection(hi) : hi_normal(hi);
  return orient_sample(spherical(theta, omega), w, 0);
}
static Col lam_get_specular_component(Ctx* cx, HitInfo* hi, V4 omega) { /* lambertian.rs:74-80 */
  float s = hi_shininess(cx, hi);
  float n_dot_omega = clampf(dot4(hi_normal(hi), omega), 0.0f, 1.0f);
  return cdivf(cmulf(cmulf(hi_specular(hi), n_dot_omega), s + 2.0f), s + 1.0f);
}
static int hi_mat_kind(HitInfo* hi) { return (int)prim_material(hi->scene, hi_primitive(hi))->kind; }
/* HitInfo::get_brdf / get_pdf / get_random_dir / get_specular_component  hit.rs:211-245 */
static Col hi_get_brdf(Ctx* cx, HitInfo* hi, V4 omega_i) {
  return hi_mat_kind(hi) == RAYCA_MATERIAL_PHONG ? lam_get_brdf(cx, hi, omega_i) : ggx_get_brdf(hi, omega_i);
}
static float hi_get_pdf(Ctx* cx, HitInfo* hi, V4 omega) {
  return hi_mat_kind(hi) == RAYCA_MATERIAL_PHONG ? lam_get_pdf(cx, hi, omega) : ggx_get_pdf(cx, hi, omega);
}
static V4 hi_get_random_dir(Ctx* cx, HitInfo* hi, Rng* rng) {
  return hi_mat_kind(hi) == RAYCA_MATERIAL_PHONG ? lam_get_random_dir(cx, hi, rng) : ggx_get_random_dir(cx, hi, rng);
}
static Col hi_get_specular_component(Ctx* cx, HitInfo* hi, V4 omega) {
  return hi_mat_kind(hi) == RAYCA_MATERIAL_PHONG ? lam_get_specular_component(cx, hi, omega) : ggx_get_specular_component(hi, omega);
}

/* Irradiance::new  hit.rs:263-287 and the Whitted-style get_radiance  ggx.rs:33-52,
 * lambertian.rs:76-81, used by the Raytracer / Scratcher integrators */
typedef struct { Col intensity; float n_dot_v, n_dot_l; V4 h; float n_dot_h, l_dot_h; } Irradiance;
static Irradiance irradiance_new(Col intensity, HitInfo* hi, V4 light_dir) {
  Irradiance ir;
  V4 l = light_dir, n = hi_normal(hi), v = vneg(hi->hit.ray.dir);
  ir.intensity = intensity;
  ir.n_dot_v = clampf(dot4(n, v), 0.0f, 1.0f) + 1e-5f;
  ir.n_dot_l = clampf(dot4(n, l), 0.0f, 1.0f);
  ir.h = vnormalize(vadd(v, l));
  ir.n_dot_h = clampf(dot4(n, ir.h), 0.0f, 1.0f);
  ir.l_dot_h = clampf(dot4(l, ir.h), 0.0f, 1.0f);
  return ir;
}
static Col ggx_get_radiance(Ctx* cx, HitInfo* hi, Irradiance ir) { /* ggx.rs:33-52 */
  float metallic = 0.0f, roughness = 1.0f;
  {
    const RaycaMaterial* m = prim_material(hi->scene, hi_primitive(hi));
    /* get_metallic_roughness -> get_pbr_material panics for a Ggx material (material/mod.rs:54-62) */
    if (m->kind == RAYCA_MATERIAL_PBR) pbr_metallic_roughness(hi->scene, m, hi_uv(hi), &metallic, &roughness);
    else cx->unsupported = 1;
  }
  float a = ir.n_dot_h * roughness; /* distribution_ggx :11-15 */
  float k = roughness / (1.0f - ir.n_dot_h * ir.n_dot_h + a * a);
  float d = k * k * F_1_PI;
  Col albedo = hi_color(hi);
  V4 f0 = vadd(vscale(splat4(0.04f), 1.0f - metallic), vscale(vec_from_col(albedo), metallic));
  f0.w = 0.0f; /* Vec3::splat keeps w = 0 */
  float fpow = powf(1.0f - ir.l_dot_h, 5.0f); /* fresnel_schlick :19-22 */
  V4 f = vadd(f0, vscale(vsub(vec3(1, 1, 1), f0), fpow));
  V4 kd = vscale(vsub(vec3(1, 1, 1), f), 1.0f - metallic);
  float ggxv = ir.n_dot_l * (ir.n_dot_v * (1.0f - roughness) + roughness); /* geometry_smith_ggx :25-30 */
  float ggxl = ir.n_dot_v * (ir.n_dot_l * (1.0f - roughness) + roughness);
  float g = 0.5f / (ggxv + ggxl);
  Col fr = fmulc(d * g, col_from_vec(f));
  Col fd = cmulf(cmul(col_from_vec(kd), albedo), F_1_PI);
  return cmulf(cmul(cadd(fd, fr), ir.intensity), ir.n_dot_l);
}
static Col lam_get_radiance(Ctx* cx, HitInfo* hi, Irradiance ir) { /* lambertian.rs:76-81 */
  Col diffuse = cmulf(hi_diffuse(hi), ir.n_dot_l);
  Col specular = cmulf(hi_specular(hi), powf(ir.n_dot_h, hi_shininess(cx, hi)));
  return cmul(cadd(diffuse, specular), ir.intensity);
}
static Col hi_get_radiance(Ctx* cx, HitInfo* hi, Irradiance ir) { /* hit.rs:202-209 */
  return hi_mat_kind(hi) == RAYCA_MATERIAL_PHONG ? lam_get_radiance(cx, hi, ir) : ggx_get_radiance(cx, hi, ir);
}

/* ============================================================================================ */
/* lights  rayca-model/src/light/ (point, quad, directional)                                                             */
/* ============================================================================================ */
static float point_get_fallof(const RaycaLight* l, const Trs* light_trs, V4 frag_pos) { /* point.rs:41-49 */
  V4 dist = vsub(vec_from_point(frag_pos), trs_get_translation(light_trs));
  float r2 = vnorm(dist);
  float r = sqrtf(r2);
  return reduce_sum4(mul4(vec3(l->attenuation[0], l->attenuation[1], l->attenuation[2]), vec3(1.0f, r, r2)));
}
static Col point_get_intensity(const RaycaLight* l, const Trs* light_trs, V4 frag_pos) { /* point.rs:37-39 */
  return cdivf(fmulc(l->intensity, col4(l->color)), point_get_fallof(l, light_trs, frag_pos));
}
static V4 quad_normal(const RaycaLight* l) { /* quad.rs:36-38 */
  return vnormalize(vcross(vec3(l->ab[0], l->ab[1], l->ab[2]), vec3(l->ac[0], l->ac[1], l->ac[2])));
}
static float quad_area(const RaycaLight* l) { /* quad.rs:40-46 (sin_theta = 1 - cos_theta, as written) */
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  float ab_len = vlen(ab), ac_len = vlen(ac);
  float cos_theta = dot4(ab, ac) / (ab_len * ac_len);
  float sin_theta = 1.0f - cos_theta;
  return sin_theta * ab_len * ac_len;
}
static V4 quad_get_a(const RaycaLight* l, const Trs* trs, uint32_t edge) { /* quad.rs:48-57 */
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  V4 a = point_from_vec(trs_get_translation(trs));
  switch (edge) { case 0: return a; case 1: return add4(a, ab); case 2: return add4(add4(a, ab), ac); default: return add4(a, ac); }
}
static V4 quad_get_b(const RaycaLight* l, const Trs* trs, uint32_t edge) { /* quad.rs:59-68 */
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  V4 a = point_from_vec(trs_get_translation(trs));
  switch (edge) { case 0: return add4(a, ab); case 1: return add4(add4(a, ab), ac); case 2: return add4(a, ac); default: return a; }
}
static V4 quad_get_radiance(const RaycaLight* l, const Trs* trs, V4 frag_pos) { /* quad.rs:70-101 */
  V4 ret = vec3(0, 0, 0);
  for (uint32_t e = 0; e < 4; ++e) {
    V4 ra = vec3_simd(sub4(quad_get_a(l, trs, e), frag_pos)), rb = vec3_simd(sub4(quad_get_b(l, trs, e), frag_pos));
    float theta = acosf(dot4(vnormalize(ra), vnormalize(rb)));
    V4 gamma = vnormalize(vcross(ra, rb));
    ret = vadd(ret, vscale(gamma, theta));
  }
  return vdivf(ret, 2.0f);
}
static Col quad_get_intensity(const RaycaLight* l, const Trs* trs, V4 frag_pos, V4 frag_n) { /* quad.rs:88-92 */
  float irradiance = dot4(quad_get_radiance(l, trs, frag_pos), frag_n);
  return cmulf(fmulc(l->intensity, col4(l->color)), irradiance);
}
static V4 quad_get_random_point(const RaycaLight* l, const Trs* trs, int stratify, uint32_t strate_count, uint32_t i, Rng* rng) { /* quad.rs:112-135 */
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  V4 step1 = vdivf(ab, (float)strate_count), step2 = vdivf(ac, (float)strate_count);
  float u1 = rng_next(rng) / (float)strate_count;
  float u2 = rng_next(rng) / (float)strate_count;
  V4 a = quad_get_a(l, trs, 0);
  V4 x1 = add4(add4(a, vscale(ab, u1)), vscale(ac, u2));
  if (stratify) {
    float i1 = (float)(i % strate_count), i2 = (float)(i / strate_count);
    x1 = add4(x1, vadd(vscale(step1, i1), vscale(step2, i2)));
  }
  return x1;
}
/* QuadLight::intersects with the widened triangles  quad.rs:138-159 */
static int quad_intersects(const RaycaLight* l, const Trs* trs, const Ray* ray, Hit* hit) {
  const float SMALL_BIAS = 1e-2f;
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  V4 a = point_from_vec(vscale(vneg(vnormalize(vadd(ab, ac))), SMALL_BIAS));
  V4 b = point_from_vec(vadd(ab, vscale(vnormalize(vsub(ab, ac)), SMALL_BIAS)));
  V4 c = point_from_vec(vadd(vadd(ab, ac), vscale(vnormalize(vadd(ab, ab)), SMALL_BIAS)));
  V4 d = point_from_vec(vadd(ac, vscale(vnormalize(vsub(ac, ab)), SMALL_BIAS)));
  V4 t1[3] = {a, c, b}, t2[3] = {a, d, c};
  if (triangle_intersects(t1, trs, ray, hit)) return 1;
  if (triangle_intersects(t2, trs, ray, hit)) return 1;
  return 0;
}
/* Light::get_direction  light/mod.rs:83-89 */
static V4 light_get_direction(Ctx* cx, const LightInfo* li, const Trs* trs, V4 frag_pos) {
  if (li->kind == RAYCA_LIGHT_DIRECTIONAL) return vneg(vrotate(vec3(1, 0, 0), trs->rotation)); /* directional.rs:47-51 */
  if (li->kind == RAYCA_LIGHT_POINT) { /* point.rs:51-55 */
    V4 dist = vnormalize(vsub(vec_from_point(frag_pos), trs_get_translation(trs)));
    return vneg(dist);
  }
  cx->unsupported = 1; /* Quad: todo!() */
  return vec3(0, 0, 0);
}
static float light_get_distance(Ctx* cx, const LightInfo* li, const Trs* trs, V4 frag_pos) { /* light/mod.rs:45-51 */
  if (li->kind == RAYCA_LIGHT_QUAD) { cx->unsupported = 1; return 0.0f; }
  /* point.rs:32-35: frag_pos - translation is Point3 - Vec3 -> Point3, then Vec3::from */
  return vlen(vec_from_point(sub4(frag_pos, trs_get_translation(trs))));
}
static Col light_get_intensity(const LightInfo* li, const Trs* trs, V4 frag_pos, V4 frag_n) { /* light/mod.rs:53-59 */
  if (li->kind == RAYCA_LIGHT_DIRECTIONAL) return fmulc(li->l.intensity, col4(li->l.color));
  if (li->kind == RAYCA_LIGHT_POINT) return point_get_intensity(&li->l, trs, frag_pos);
  return quad_get_intensity(&li->l, trs, frag_pos, frag_n);
}

/* ============================================================================================ */
/* samplers  rayca-soft/src/sampler/                                                              */
/* ============================================================================================ */
typedef struct { uint32_t light; V4 omega; Col x; float pdf; int is_nee; } Sample;

static uint32_t cfg_strate_count(const RaycaConfig* c) { /* config.rs:73-79 */
  return c->light_stratify ? (uint32_t)sqrtf((float)c->light_samples) : 1u;
}
/* NextEventEstimationSample::get_pdf  nee.rs:44-66 */
static float nee_get_pdf(Ctx* cx, uint32_t light, HitInfo* hi, V4 omega) {
  const OracleScene* s = cx->scene;
  const LightInfo* li = &s->lights[light];
  Ray ray = hi_next_ray(hi, omega);
  Hit lh;
  if (li->kind != RAYCA_LIGHT_QUAD) return 0.0f; /* Light::intersects -> None  light/mod.rs:107-113 */
  if (!quad_intersects(&li->l, &s->local_trs[li->node], &ray, &lh)) return 0.0f;
  float area = quad_area(&li->l);
  if (area == 0.0f) return 0.0f;
  float nl_dot_omega = clampf(dot4(quad_normal(&li->l), omega), 0.0f, 1.0f);
  if (nl_dot_omega == 0.0f) return 0.0f;
  float r_squared = vnorm(vec3_simd(sub4(lh.point, hi->hit.point)));
  return r_squared / (area * nl_dot_omega);
}
/* get_quad_light_sample  nee.rs:72-125 */
static Sample nee_quad_sample(Ctx* cx, HitInfo* hi, uint32_t light_sample_index, uint32_t light, Rng* rng) {
  const OracleScene* s = cx->scene;
  const LightInfo* li = &s->lights[light];
  Col ld = COL_BLACK;
  float area = quad_area(&li->l);
  uint32_t strate_count = cfg_strate_count(cx->cfg);
  V4 x1 = quad_get_random_point(&li->l, &s->local_trs[li->node], (int)cx->cfg->light_stratify, strate_count, light_sample_index, rng);
  V4 x = hi->hit.point;
  V4 x_to_x1 = vec3_simd(sub4(x1, x));
  V4 omega = vnormalize(x_to_x1);
  Ray shadow_ray = hi_next_ray(hi, omega);
  float pdf = 0.0f;
  HitInfo sh;
  cx->rays_shadow++;
  if (tlas_intersects(cx, shadow_ray, &sh)) {
    if (hi_is_emissive(&sh)) {
      Col le = fmulc(li->l.intensity, col4(li->l.color));
      Col brdf = hi_get_brdf(cx, hi, omega);
      float r_squared = vnorm(x_to_x1);
      float d_omega = dot4(quad_normal(&li->l), omega) / r_squared;
      float n_dot_omega = clampf(dot4(hi_normal(hi), omega), 0.0f, 1.0f);
      ld = cmulf(cmulf(cmul(cmulf(le, area), brdf), n_dot_omega), d_omega);
      pdf = nee_get_pdf(cx, light, hi, omega);
    }
  }
  Sample sm = {light, omega, ld, pdf, 1};
  return sm;
}
/* get_point_light_sample  nee.rs:127-166 */
static Sample nee_point_sample(Ctx* cx, HitInfo* hi, uint32_t light) {
  const OracleScene* s = cx->scene;
  const LightInfo* li = &s->lights[light];
  const Trs* ltrs = &s->local_trs[li->node];
  V4 x1 = point_from_vec(trs_get_translation(ltrs));
  V4 x = hi->hit.point;
  V4 x_to_x1 = vec3_simd(sub4(x1, x));
  float point_light_distance = vlen(x_to_x1);
  V4 omega = vnormalize(x_to_x1);
  Sample ret = {light, omega, COL_BLACK, 0.0f, 1};
  Ray shadow_ray = hi_next_ray(hi, omega);
  HitInfo sh;
  cx->rays_shadow++;
  if (tlas_intersects(cx, shadow_ray, &sh)) {
    if (sh.hit.depth < point_light_distance) return ret;
  }
  Col le = point_get_intensity(&li->l, ltrs, x);
  Col brdf = hi_get_brdf(cx, hi, omega);
  float r_squared = vnorm(x_to_x1);
  float d_omega = 1.0f / r_squared;
  float n_dot_omega = clampf(dot4(hi_normal(hi), omega), 0.0f, 1.0f);
  ret.x = cmulf(cmulf(cmul(le, brdf), n_dot_omega), d_omega);
  ret.pdf = nee_get_pdf(cx, light, hi, omega);
  return ret;
}
/* get_samples  nee.rs:168-206: light-major, light_samples each */
static uint32_t nee_get_samples(Ctx* cx, HitInfo* hi, Rng* rng, Sample* out, uint32_t cap) {
  uint32_t n = 0;
  for (uint32_t light = 0; light < cx->scene->light_count; ++light)
    for (uint32_t i = 0; i < cx->cfg->light_samples; ++i) {
      Sample sm;
      uint32_t kind = cx->scene->lights[light].kind;
      if (kind == RAYCA_LIGHT_POINT) sm = nee_point_sample(cx, hi, light);
      else if (kind == RAYCA_LIGHT_QUAD) sm = nee_quad_sample(cx, hi, i, light, rng);
      else { cx->unsupported = 1; memset(&sm, 0, sizeof sm); sm.x = COL_BLACK; } /* todo!() nee.rs:178 */
      if (n < cap) out[n] = sm;
      n++;
    }
  return n;
}
/* CosineSampler::get_random_dir  cosine.rs:65-88 ; HemisphereSampler  hemisphere.rs:17-40 */
static V4 cosine_random_dir(HitInfo* hi, Rng* rng) {
  float e1 = rng_next(rng), e2 = rng_next(rng);
  float theta = acosf(sqrtf(e1));
  float omega = 2.0f * F_PI * e2;
  return orient_sample(spherical(theta, omega), hi_normal(hi), 0);
}
static V4 hemisphere_random_dir(HitInfo* hi, Rng* rng) {
  float e1 = rng_next(rng), e2 = rng_next(rng);
  float theta = acosf(e1);
  float omega = 2.0f * F_PI * e2;
  return orient_sample(spherical(theta, omega), hi_normal(hi), 1);
}
static V4 indirect_random_dir(Ctx* cx, HitInfo* hi, Rng* rng) {
  switch (cx->cfg->indirect_sampler) {
    case RAYCA_SAMPLER_HEMISPHERE: return hemisphere_random_dir(hi, rng);
    case RAYCA_SAMPLER_COSINE: return cosine_random_dir(hi, rng);
    case RAYCA_SAMPLER_BRDF: return hi_get_random_dir(cx, hi, rng);
    default: cx->unsupported = 1; return vec3(0, 0, 1);
  }
}
/* SoftSampler::get_radiance  cosine.rs:90-99, hemisphere.rs:42-52, brdf.rs:77-90 */
static Col indirect_get_radiance(Ctx* cx, HitInfo* hi, V4 omega_i, Col indirect_sample, float weight) {
  switch (cx->cfg->indirect_sampler) {
    case RAYCA_SAMPLER_COSINE: return cmulf(cmul(fmulc(F_PI, hi_get_brdf(cx, hi, omega_i)), indirect_sample), weight);
    case RAYCA_SAMPLER_HEMISPHERE: {
      Col brdf = hi_get_brdf(cx, hi, omega_i);
      float cosine_law = clampf(dot4(hi_normal(hi), omega_i), 0.0f, 1.0f);
      return cmulf(cmul(cmulf(fmulc(2.0f * F_PI, brdf), cosine_law), indirect_sample), weight);
    }
    case RAYCA_SAMPLER_BRDF: {
      Col cd = hi_diffuse(hi);
      Col cs = hi_get_specular_component(cx, hi, omega_i);
      return cmulf(cmul(indirect_sample, cadd(cd, cs)), weight);
    }
    default: cx->unsupported = 1; return COL_BLACK;
  }
}
/* BrdfSampler::sample_direct  brdf.rs:47-66 */
static Sample brdf_sample_direct(Ctx* cx, HitInfo* hi, Rng* rng) {
  V4 omega = hi_get_random_dir(cx, hi, rng);
  float pdf = hi_get_pdf(cx, hi, omega);
  Col x = COL_BLACK;
  Ray shadow_ray = hi_next_ray(hi, omega);
  HitInfo sh;
  cx->rays_shadow++;
  if (tlas_intersects(cx, shadow_ray, &sh)) {
    if (hi_is_emissive(&sh)) {
      Col li = hi_emission(&sh);
      x = cmul(li, cadd(hi_diffuse(hi), hi_get_specular_component(cx, hi, omega)));
    }
  }
  Sample sm = {RAYCA_NONE, omega, x, pdf, 0};
  return sm;
}
#define MAX_SAMPLES 256
/* MultipleImportanceSampling::get_direct_lighting  mis.rs:40-72 */
static float mis_pdf_nee(Ctx* cx, HitInfo* hi, const Sample* sample, const Sample* ls, uint32_t n) { /* mis.rs:25-37 */
  float ret = 0.0f;
  for (uint32_t i = 0; i < n; ++i) ret += nee_get_pdf(cx, ls[i].light, hi, sample->omega);
  return ret / (float)n;
}
static Col direct_lighting(Ctx* cx, HitInfo* hi, Rng* rng) {
  Sample ls[MAX_SAMPLES];
  switch (cx->cfg->direct_sampler) {
    case RAYCA_SAMPLER_NONE: return COL_BLACK; /* NoSampler  sampler/mod.rs:110-115 */
    case RAYCA_SAMPLER_NEE: { /* nee.rs:209-216 */
      uint32_t n = nee_get_samples(cx, hi, rng, ls, MAX_SAMPLES);
      if (n > MAX_SAMPLES) { cx->unsupported = 1; n = MAX_SAMPLES; }
      Col ret = COL_BLACK;
      for (uint32_t i = 0; i < n; ++i) ret = cadd(ret, ls[i].x);
      return ret;
    }
    case RAYCA_SAMPLER_MIS: {
      uint32_t n = nee_get_samples(cx, hi, rng, ls, MAX_SAMPLES);
      if (n > MAX_SAMPLES) { cx->unsupported = 1; n = MAX_SAMPLES; }
      Sample bs = brdf_sample_direct(cx, hi, rng);
      Col ret = COL_BLACK;
      for (uint32_t i = 0; i < n; ++i) {
        float pdf_nee = mis_pdf_nee(cx, hi, &ls[i], ls, n);
        float pdf_brdf = hi_get_pdf(cx, hi, ls[i].omega); /* BrdfSample::get_pdf_for  brdf.rs:35-37 */
        float pdf_den = powf(pdf_nee, 2.0f) + powf(pdf_brdf, 2.0f);
        float w = pdf_den == 0.0f ? 0.0f : powf(pdf_nee, 2.0f) / pdf_den;
        ret = cadd(ret, fmulc(w, ls[i].x));
      }
      float pdf_nee = mis_pdf_nee(cx, hi, &bs, ls, n);
      float pdf_brdf = bs.pdf;
      float pdf_den = powf(pdf_nee, 2.0f) + powf(pdf_brdf, 2.0f);
      float w = pdf_den == 0.0f ? 0.0f : powf(pdf_brdf, 2.0f) / pdf_den;
      ret = cadd(ret, fmulc(w, bs.x));
      return ret;
    }
    default: cx->unsupported = 1; return COL_BLACK; /* panic!("Unsupported direct sampler") */
  }
}

/* ============================================================================================ */
/* integrators  rayca-soft/src/integrator/                                                        */
/* ============================================================================================ */
static int trace(Ctx* cx, Ray ray, uint32_t depth, uint32_t key, Col* out);

/* Ray::next_russian_roulette  ray.rs:95-105 */
static int next_russian_roulette(Col next_throughput, Rng* rng, float* weight) {
  float q = 1.0f - fminf(c_max_rgb(next_throughput), 1.0f);
  if (q < clampf(rng_next(rng), 0.0f, 1.0f - FLT_EPSILON)) { *weight = 1.0f / (1.0f - q); return 1; }
  return 0;
}
/* Pathtracer::trace_impl  pathtracer.rs:68-106 and get_indirect_lighting :23-66.
 * `key` identifies this path vertex for the counter-based RNG: draws at the vertex use dimensions
 * 0,1,2,... in program order; the k-th indirect child vertex gets oracle_rng_child(key, k). */
static int pathtracer_trace_impl(Ctx* cx, Ray ray, uint32_t depth, int collect_emissive, uint32_t key, Col* out) {
  const RaycaConfig* cfg = cx->cfg;
  if (!cfg->russian_roulette && depth >= cfg->max_depth) return 0;
  if (depth > 0) cx->rays_bounce++; /* statistics: secondary rays that are actually traced */
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  Col ambient_and_emissive = hi_color(&hit);
  if (collect_emissive && hi_is_emissive(&hit)) { *out = ambient_and_emissive; return 1; }
  Rng rng = {key, 0};
  Col direct = direct_lighting(cx, &hit, &rng);
  uint32_t indirect_depth_limit = cfg->direct_sampler != RAYCA_SAMPLER_NONE ? cfg->max_depth - 1 : cfg->max_depth;
  Col indirect = COL_BLACK;
  if (cfg->russian_roulette || depth < indirect_depth_limit) {
    Col li = COL_BLACK;
    int child_collect = cfg->direct_sampler == RAYCA_SAMPLER_NONE;
    for (uint32_t k = 0; k < cfg->light_samples; ++k) {
      V4 omega_i = indirect_random_dir(cx, &hit, &rng);
      Col brdf = hi_get_brdf(cx, &hit, omega_i);
      Ray next_ray = hi_next_ray(&hit, omega_i);
      float weight = 1.0f;
      if (cfg->russian_roulette) {
        Col next_throughput = cmulf(cmul(fmulc(2.0f * F_PI, hit.hit.ray.throughput), brdf), clampf(dot4(hi_normal(&hit), omega_i), 0.0f, 1.0f));
        float boost;
        if (next_russian_roulette(next_throughput, &rng, &boost)) { weight = boost; next_ray.throughput = cmulf(next_throughput, boost); }
        else continue;
      }
      Col indirect_sample;
      if (pathtracer_trace_impl(cx, next_ray, depth + 1, child_collect, oracle_rng_child(key, k), &indirect_sample))
        li = cadd(li, indirect_get_radiance(cx, &hit, omega_i, indirect_sample, weight));
    }
    indirect = cadd(indirect, cdivf(li, (float)cfg->light_samples));
  }
  *out = cadd(direct, indirect);
  return 1;
}
/* Flat::trace  flat.rs:16-28 */
static int flat_trace(Ctx* cx, Ray ray, Col* out) {
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  *out = hi_color(&hit);
  return 1;
}
/* shared light loop of Raytracer (raytracer.rs:34-62) and Scratcher (scratcher.rs:45-75) */
static Col whitted_lights(Ctx* cx, HitInfo* hit, Col acc) {
  const OracleScene* s = cx->scene;
  for (uint32_t li = 0; li < s->light_count; ++li) {
    const LightInfo* l = &s->lights[li];
    const Trs* ltrs = &s->local_trs[l->node];
    V4 light_dir = light_get_direction(cx, l, ltrs, hit->hit.point);
    Ray shadow_ray = hi_next_ray(hit, light_dir); /* get_shadow_ray  hit.rs:195-198 */
    HitInfo sh;
    int is_light;
    cx->rays_shadow++;
    if (!tlas_intersects(cx, shadow_ray, &sh)) is_light = 1;
    else {
      float light_distance = light_get_distance(cx, l, ltrs, hit->hit.point);
      if (sh.hit.depth > light_distance) is_light = 1;
      else is_light = c_is_transparent(hi_color(&sh));
    }
    if (is_light) {
      Col intensity = light_get_intensity(l, ltrs, hit->hit.point, hi_normal(hit));
      Irradiance ir = irradiance_new(intensity, hit, light_dir);
      acc = cadd(acc, hi_get_radiance(cx, hit, ir));
    }
  }
  return acc;
}
static int raytracer_trace(Ctx* cx, Ray ray, uint32_t depth, Col* out) { /* raytracer.rs:16-76 */
  if (depth > cx->cfg->max_depth) return 0;
  if (depth > 0) cx->rays_bounce++;
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  Col ambient_emissive = hi_color(&hit);
  Col light_contribution = whitted_lights(cx, &hit, COL_BLACK);
  Ray reflection_ray = ray_new(hi_next_ray_origin(&hit), hi_reflection(&hit));
  Col reflection_color;
  if (raytracer_trace(cx, reflection_ray, depth + 1, &reflection_color))
    light_contribution = cadd(light_contribution, cmulf(cmul(reflection_color, hi_specular(&hit)), 1.0f));
  *out = cadd(ambient_emissive, light_contribution);
  return 1;
}
static int scratcher_trace(Ctx* cx, Ray ray, uint32_t depth, Col* out) { /* scratcher.rs:16-89 */
  if (depth > cx->cfg->max_depth) return 0;
  if (depth > 0) cx->rays_bounce++;
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  Col pixel_color = COL_BLACK;
  if (c_is_transparent(hi_color(&hit))) {
    V4 torigin = add4(hit.hit.point, vscale(vneg(hi_normal(&hit)), ORC_RAY_BIAS)); /* hit.rs:178-187 */
    Col transmit;
    if (scratcher_trace(cx, ray_new(torigin, hit.hit.ray.dir), depth + 1, &transmit)) {
      transmit = cover(transmit, hi_color(&hit));
      pixel_color = cadd(pixel_color, transmit);
    }
  }
  pixel_color = whitted_lights(cx, &hit, pixel_color);
  V4 reflection = hi_reflection(&hit);
  Ray reflection_ray = ray_new(hi_next_ray_origin(&hit), reflection);
  Col refl;
  if (scratcher_trace(cx, reflection_ray, depth + 1, &refl)) {
    Irradiance ir = irradiance_new(refl, &hit, reflection);
    pixel_color = cadd(pixel_color, hi_get_radiance(cx, &hit, ir));
  }
  *out = pixel_color;
  return 1;
}
static int direct_trace(Ctx* cx, Ray ray, uint32_t depth, uint32_t key, Col* out) { /* direct.rs:16-86 */
  const OracleScene* s = cx->scene;
  const RaycaConfig* cfg = cx->cfg;
  if (depth >= cfg->max_depth) return 0;
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  Col ambient_and_emissive = hi_color(&hit);
  if (hi_is_emissive(&hit)) { *out = ambient_and_emissive; return 1; }
  Rng rng = {key, 0};
  Col light_contribution = COL_BLACK;
  uint32_t strate_count = cfg_strate_count(cfg);
  for (uint32_t li = 0; li < s->light_count; ++li) {
    const LightInfo* l = &s->lights[li];
    if (l->kind != RAYCA_LIGHT_QUAD) continue;
    Col ld = COL_BLACK;
    for (uint32_t i = 0; i < cfg->light_samples; ++i) {
      V4 x1 = quad_get_random_point(&l->l, &s->local_trs[l->node], (int)cfg->light_stratify, strate_count, i, &rng);
      V4 x1_to_hit = vec3_simd(sub4(x1, hit.hit.point));
      V4 omega_i = vnormalize(x1_to_hit);
      Ray shadow_ray = ray_new(hi_next_ray_origin(&hit), omega_i);
      HitInfo sh;
      cx->rays_shadow++;
      if (tlas_intersects(cx, shadow_ray, &sh)) { if (!hi_is_emissive(&sh)) continue; }
      Col brdf = hi_get_brdf(cx, &hit, omega_i);
      float r_squared = powf(vlen(x1_to_hit), 2.0f);
      float d_omega_i = dot4(quad_normal(&l->l), omega_i) / r_squared;
      ld = cadd(ld, cmulf(cmulf(brdf, dot4(hi_normal(&hit), omega_i)), d_omega_i));
    }
    Col lcol = fmulc(l->l.intensity, col4(l->l.color));
    float area = quad_area(&l->l);
    ld = cdivf(cmul(cmulf(lcol, area), ld), (float)cfg->light_samples);
    light_contribution = cadd(light_contribution, ld);
  }
  *out = light_contribution;
  return 1;
}
static int analytic_direct_trace(Ctx* cx, Ray ray, uint32_t depth, Col* out) { /* analyticdirect.rs:16-48 */
  const OracleScene* s = cx->scene;
  if (depth >= cx->cfg->max_depth) return 0;
  HitInfo hit;
  if (!tlas_intersects(cx, ray, &hit)) return 0;
  cx->hits_shaded++;
  Col ambient_and_emission = hi_color(&hit);
  if (hi_is_emissive(&hit)) { *out = ambient_and_emission; return 1; }
  V4 n = hi_normal(&hit);
  Col light_contribution = COL_BLACK;
  for (uint32_t li = 0; li < s->light_count; ++li) {
    const LightInfo* l = &s->lights[li];
    light_contribution = cadd(light_contribution, light_get_intensity(l, &s->local_trs[l->node], hit.hit.point, n));
  }
  Col f = cmulf(hi_diffuse(&hit), F_1_PI);
  *out = cmul(f, light_contribution);
  return 1;
}
/* IntegratorStrategy::get_integrator().trace  integrator/mod.rs:43-72 */
static int trace(Ctx* cx, Ray ray, uint32_t depth, uint32_t key, Col* out) {
  switch (cx->cfg->integrator) {
    case RAYCA_INTEGRATOR_SCRATCHER: return scratcher_trace(cx, ray, depth, out);
    case RAYCA_INTEGRATOR_RAYTRACER: return raytracer_trace(cx, ray, depth, out);
    case RAYCA_INTEGRATOR_FLAT: return flat_trace(cx, ray, out);
    case RAYCA_INTEGRATOR_ANALYTIC_DIRECT: return analytic_direct_trace(cx, ray, depth, out);
    case RAYCA_INTEGRATOR_DIRECT: return direct_trace(cx, ray, depth, key, out);
    case RAYCA_INTEGRATOR_PATHTRACER: return pathtracer_trace_impl(cx, ray, depth, 1, key, out);
    default: cx->unsupported = 1; return 0;
  }
}

/* ============================================================================================ */
/* scene flattening  rayca-soft/src/scene.rs:190-282 and bvh/primitive.rs:194-395                 */
/* ============================================================================================ */
static Col vcolor(const RaycaSceneDesc* d, uint32_t v) { return d->colors ? col(d->colors[4 * v], d->colors[4 * v + 1], d->colors[4 * v + 2], d->colors[4 * v + 3]) : COL_WHITE; }
static V4 vattr3(const float* arr, uint32_t v, V4 dflt) { return arr ? vec3(arr[3 * v], arr[3 * v + 1], arr[3 * v + 2]) : dflt; }

static int fetch_index(const RaycaSceneDesc* d, const RaycaPrimitive* p, uint32_t i, uint32_t* out) {
  uint64_t off = p->index_byte_offset;
  switch (p->index_type) {
    case RAYCA_INDEX_U8: if (off + i >= d->index_byte_count) return 0; *out = d->index_bytes[off + i]; return 1;
    case RAYCA_INDEX_U16: { if (off + 2ull * i + 2 > d->index_byte_count) return 0; uint16_t v; memcpy(&v, d->index_bytes + off + 2ull * i, 2); *out = v; return 1; }
    case RAYCA_INDEX_U32: { if (off + 4ull * i + 4 > d->index_byte_count) return 0; uint32_t v; memcpy(&v, d->index_bytes + off + 4ull * i, 4); *out = v; return 1; }
    default: return 0; /* panic!("Index type not supported")  primitive.rs:258 */
  }
}

typedef struct { Prim* v; uint32_t n, cap; } PrimVec;
static void pv_push(PrimVec* pv, const Prim* p) {
  if (pv->n == pv->cap) { pv->cap = pv->cap ? pv->cap * 2 : 256; pv->v = (Prim*)realloc(pv->v, sizeof(Prim) * pv->cap); }
  pv->v[pv->n++] = *p;
}
/* BvhPrimitive::from_triangle_mesh_impl  primitive.rs:194-237 */
static int32_t prims_from_triangle_mesh(const OracleScene* s, const RaycaSceneDesc* d, const RaycaPrimitive* p, uint32_t node, PrimVec* out, uint32_t* flat_counter) {
  const Trs* trs = &s->world_trs[node];
  M3 tangent_matrix = m3_from_trs(trs);
  M3 inv = m3_from_inv_trs(trs);
  M3 normal_matrix = m3_transpose(&inv);
  for (uint32_t i = 0; i < p->index_count / 3; ++i) {
    Prim pr;
    memset(&pr, 0, sizeof pr);
    pr.kind = RAYCA_GEOMETRY_TRIANGLE_MESH;
    pr.node = node;
    pr.material = p->material;
    for (int k = 0; k < 3; ++k) {
      uint32_t idx;
      if (!fetch_index(d, p, i * 3 + (uint32_t)k, &idx)) return fail(RAYCA_ERR_BAD_ARG, "index fetch out of range / bad index type");
      if (idx >= p->vertex_count) return fail(RAYCA_ERR_BAD_ARG, "vertex index %u out of range", idx);
      uint32_t v = p->first_vertex + idx;
      if (v >= d->vertex_count) return fail(RAYCA_ERR_BAD_ARG, "vertex %u out of range", v);
      pr.p[k] = point3(d->positions[3 * v], d->positions[3 * v + 1], d->positions[3 * v + 2]);
      pr.ext[k].color = vcolor(d, v);
      pr.ext[k].normal = m3_vec(&normal_matrix, vattr3(d->normals, v, vec3(0, 0, 1)));
      pr.ext[k].tangent = m3_vec(&tangent_matrix, vattr3(d->tangents, v, vec3(0, 0, 0)));
      pr.ext[k].bitangent = m3_vec(&tangent_matrix, vattr3(d->bitangents, v, vec3(0, 0, 0)));
      pr.ext[k].uv = d->uvs ? v2(d->uvs[2 * v], d->uvs[2 * v + 1]) : v2(0, 0);
    }
    /* Triangle::new centroid  triangle.rs:59-63 */
    pr.centroid = tri_model_centroid(pr.p);
    pr.src = (*flat_counter)++;
    pv_push(out, &pr);
  }
  return RAYCA_OK;
}
/* BvhPrimitive::from_quad_light  primitive.rs:310-346 */
static void prims_from_quad_light(const RaycaLight* l, uint32_t node, PrimVec* out, uint32_t* flat_counter) {
  V4 ab = vec3(l->ab[0], l->ab[1], l->ab[2]), ac = vec3(l->ac[0], l->ac[1], l->ac[2]);
  V4 normal = quad_normal(l);
  V4 qa = point3(0, 0, 0);
  V4 a = qa, b = add4(qa, ab), dd = add4(add4(qa, ab), ac), c = add4(qa, ac);
  V4 tri[2][3] = {{a, dd, b}, {a, c, dd}};
  for (int t = 0; t < 2; ++t) {
    Prim pr;
    memset(&pr, 0, sizeof pr);
    pr.kind = RAYCA_GEOMETRY_TRIANGLE_MESH;
    pr.node = node;
    pr.material = l->material;
    for (int k = 0; k < 3; ++k) {
      pr.p[k] = tri[t][k];
      pr.ext[k].color = COL_WHITE;
      pr.ext[k].normal = normal;
      pr.ext[k].tangent = vec3(0, 0, 0);
      pr.ext[k].bitangent = vec3(0, 0, 0);
      pr.ext[k].uv = v2(0, 0);
    }
    pr.centroid = tri_model_centroid(pr.p);
    pr.src = (*flat_counter)++;
    pv_push(out, &pr);
  }
}

typedef struct { uint32_t* v; uint32_t n, cap; } U32Vec;
static void uv_push(U32Vec* a, uint32_t x) {
  if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 16; a->v = (uint32_t*)realloc(a->v, 4 * a->cap); }
  a->v[a->n++] = x;
}
static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

static uint32_t online_cores(void) { long n = sysconf(_SC_NPROCESSORS_ONLN); return n > 0 ? (uint32_t)n : 1u; }

int32_t oracle_scene_create(const RaycaSceneDesc* d, const RaycaConfig* cfg, const OracleOptions* opts, OracleScene** out) {
  if (!d || !out) return fail(RAYCA_ERR_BAD_ARG, "null argument");
  if (d->abi_version != RAYCA_ABI_VERSION) return fail(RAYCA_ERR_BAD_ARG, "abi version mismatch");
  if (d->node_count && !d->nodes) return fail(RAYCA_ERR_BAD_ARG, "nodes is null");
  if (d->vertex_count && !d->positions) return fail(RAYCA_ERR_BAD_ARG, "positions is null");
  OracleScene* s = (OracleScene*)calloc(1, sizeof *s);
  if (opts) s->opts = *opts;
  if (s->opts.threads == 0) s->opts.threads = online_cores();
  const uint32_t use_bvh = cfg ? cfg->bvh : 1u; /* Config::default(): bvh = true  config.rs:12-13 */
  int32_t rc = RAYCA_OK;

  /* --- SceneDrawInfo::new: world transforms in traversal order  scene.rs:206-282 ------------- */
  uint32_t N = d->node_count;
  s->node_count = N;
  s->local_trs = (Trs*)calloc(N ? N : 1, sizeof(Trs));
  s->world_trs = (Trs*)calloc(N ? N : 1, sizeof(Trs));
  for (uint32_t i = 0; i < N; ++i) {
    const RaycaNode* n = &d->nodes[i];
    if (n->parent >= (int32_t)i) { rc = fail(RAYCA_ERR_BAD_ARG, "node %u: parent must precede child", i); goto done; }
    s->local_trs[i] = trs_from_abi(&n->trs);
    if (n->parent < 0) s->world_trs[i] = s->local_trs[i];
    else s->world_trs[i] = trs_mul(&s->world_trs[n->parent], &s->local_trs[i]);
  }
  /* DFS pre-order over children lists (children in ascending index order) */
  uint32_t* order = (uint32_t*)malloc(4 * (N ? N : 1));
  uint32_t order_n = 0;
  {
    uint32_t* child_count = (uint32_t*)calloc(N + 1, 4);
    uint32_t* child_start = (uint32_t*)calloc(N + 2, 4);
    uint32_t* children = (uint32_t*)malloc(4 * (N ? N : 1));
    uint32_t tops = 0;
    for (uint32_t i = 0; i < N; ++i) { if (d->nodes[i].parent >= 0) child_count[d->nodes[i].parent]++; else tops++; }
    for (uint32_t i = 0; i < N; ++i) child_start[i + 1] = child_start[i] + child_count[i];
    uint32_t* fill = (uint32_t*)calloc(N + 1, 4);
    for (uint32_t i = 0; i < N; ++i) if (d->nodes[i].parent >= 0) { uint32_t p = (uint32_t)d->nodes[i].parent; children[child_start[p] + fill[p]++] = i; }
    uint32_t* stack = (uint32_t*)malloc(4 * (N ? N : 1));
    uint32_t sp = 0;
    for (int64_t i = (int64_t)N - 1; i >= 0; --i) if (d->nodes[i].parent < 0) stack[sp++] = (uint32_t)i;
    while (sp) {
      uint32_t n = stack[--sp];
      order[order_n++] = n;
      for (int64_t c = (int64_t)child_count[n] - 1; c >= 0; --c) stack[sp++] = children[child_start[n] + (uint32_t)c];
    }
    free(child_count); free(child_start); free(children); free(fill); free(stack);
    (void)tops;
  }
  /* collect draw infos */
  U32Vec mesh_nodes = {0}, light_nodes = {0}, models = {0};
  s->has_camera = 0;
  for (uint32_t k = 0; k < order_n; ++k) {
    uint32_t i = order[k];
    const RaycaNode* n = &d->nodes[i];
    if (n->mesh != RAYCA_NONE) {
      if (n->mesh >= d->mesh_count) { rc = fail(RAYCA_ERR_BAD_ARG, "node %u: mesh out of range", i); goto done_lists; }
      uv_push(&mesh_nodes, i);
    }
    if (n->light != RAYCA_NONE) {
      if (n->light >= d->light_count) { rc = fail(RAYCA_ERR_BAD_ARG, "node %u: light out of range", i); goto done_lists; }
      uv_push(&light_nodes, i);
    }
    if (n->camera != RAYCA_NONE && !s->has_camera) {
      if (n->camera >= d->camera_count) { rc = fail(RAYCA_ERR_BAD_ARG, "node %u: camera out of range", i); goto done_lists; }
      s->has_camera = 1; s->camera_node = i; s->camera_yfov = d->cameras[n->camera].yfov_radians;
    }
  }
  s->light_count = light_nodes.n;
  s->lights = (LightInfo*)calloc(light_nodes.n ? light_nodes.n : 1, sizeof(LightInfo));
  for (uint32_t k = 0; k < light_nodes.n; ++k) {
    uint32_t node = light_nodes.v[k];
    s->lights[k].node = node;
    s->lights[k].l = d->lights[d->nodes[node].light];
    s->lights[k].kind = s->lights[k].l.kind;
  }
  /* materials / textures / images */
  s->material_count = d->material_count;
  s->materials = (RaycaMaterial*)malloc(sizeof(RaycaMaterial) * (d->material_count ? d->material_count : 1));
  if (d->material_count) memcpy(s->materials, d->materials, sizeof(RaycaMaterial) * d->material_count);
  s->texture_count = d->texture_count;
  s->textures = (RaycaTexture*)malloc(sizeof(RaycaTexture) * (d->texture_count ? d->texture_count : 1));
  if (d->texture_count) memcpy(s->textures, d->textures, sizeof(RaycaTexture) * d->texture_count);
  s->image_count = d->image_count;
  s->images = (RaycaImage*)malloc(sizeof(RaycaImage) * (d->image_count ? d->image_count : 1));
  if (d->image_count) memcpy(s->images, d->images, sizeof(RaycaImage) * d->image_count);
  s->image_byte_count = d->image_byte_count;
  s->image_bytes = (uint8_t*)malloc(d->image_byte_count ? d->image_byte_count : 1);
  if (d->image_byte_count) memcpy(s->image_bytes, d->image_bytes, d->image_byte_count);

  /* --- BvhScene::from_scene: one BvhModel per model that owns a mesh or a quad light --------- */
  for (uint32_t k = 0; k < mesh_nodes.n; ++k) uv_push(&models, d->nodes[mesh_nodes.v[k]].model);
  for (uint32_t k = 0; k < light_nodes.n; ++k) if (d->lights[d->nodes[light_nodes.v[k]].light].kind == RAYCA_LIGHT_QUAD) uv_push(&models, d->nodes[light_nodes.v[k]].model);
  qsort(models.v, models.n, 4, cmp_u32);
  { uint32_t w = 0; for (uint32_t k = 0; k < models.n; ++k) if (k == 0 || models.v[k] != models.v[k - 1]) models.v[w++] = models.v[k]; models.n = w; }
  s->blas_count = models.n;
  s->blass = (Blas*)calloc(models.n ? models.n : 1, sizeof(Blas));
  s->blas_nodes = (BlasNode*)calloc(models.n ? models.n : 1, sizeof(BlasNode));
  uint32_t flat_counter = 0;
  for (uint32_t m = 0; m < models.n; ++m) {
    PrimVec pv = {0};
    /* BvhModel::from_model  primitive.rs:355-370: meshes first, then quad lights */
    for (uint32_t k = 0; k < mesh_nodes.n && rc == RAYCA_OK; ++k) {
      uint32_t node = mesh_nodes.v[k];
      if (d->nodes[node].model != models.v[m]) continue;
      const RaycaMesh* mesh = &d->meshes[d->nodes[node].mesh];
      for (uint32_t pi = mesh->first_primitive; pi < mesh->first_primitive + mesh->primitive_count && rc == RAYCA_OK; ++pi) {
        if (pi >= d->primitive_count) { rc = fail(RAYCA_ERR_BAD_ARG, "primitive out of range"); break; }
        const RaycaPrimitive* p = &d->primitives[pi];
        if (p->geometry == RAYCA_GEOMETRY_TRIANGLE_MESH) rc = prims_from_triangle_mesh(s, d, p, node, &pv, &flat_counter);
        else { /* from_sphere  primitive.rs:262-274 */
          Prim pr;
          memset(&pr, 0, sizeof pr);
          pr.kind = RAYCA_GEOMETRY_SPHERE; pr.node = node; pr.material = p->material;
          pr.center = point3(p->sphere_center[0], p->sphere_center[1], p->sphere_center[2]);
          pr.radius = p->sphere_radius; pr.radius2 = p->sphere_radius * p->sphere_radius;
          pr.src = flat_counter++;
          pv_push(&pv, &pr);
        }
      }
    }
    for (uint32_t k = 0; k < light_nodes.n; ++k) {
      uint32_t node = light_nodes.v[k];
      const RaycaLight* l = &d->lights[d->nodes[node].light];
      if (d->nodes[node].model != models.v[m] || l->kind != RAYCA_LIGHT_QUAD) continue;
      prims_from_quad_light(l, node, &pv, &flat_counter);
    }
    Blas* bl = &s->blass[m];
    bl->prims = pv.v; bl->prim_count = pv.n; bl->model = models.v[m];
    bl->max_depth = use_bvh ? 255 : 0; /* scene.rs:95-98, blas.rs:194 */
    s->blas_nodes[m].blas = m; s->blas_nodes[m].model = m;
  }
  if (rc != RAYCA_OK) goto done_lists;
  s->flat_prim_count = flat_counter;
  /* export copy of world triangles (flatten order) + caches */
  s->world_tris = (float*)calloc((size_t)(flat_counter ? flat_counter : 1) * 9, 4);
  for (uint32_t m = 0; m < s->blas_count; ++m)
    for (uint32_t i = 0; i < s->blass[m].prim_count; ++i) {
      Prim* pr = &s->blass[m].prims[i];
      prim_cache_world(s, pr);
      if (pr->kind == RAYCA_GEOMETRY_TRIANGLE_MESH)
        for (int k = 0; k < 3; ++k) { float* o = s->world_tris + 9 * (size_t)pr->src + 3 * k; o[0] = pr->wp[k].x; o[1] = pr->wp[k].y; o[2] = pr->wp[k].z; }
    }
  /* --- Tlas::new  tlas.rs:248-269 ------------------------------------------------------------ */
  {
    OracleScene build_view = *s; /* the build honours opts.xform only through prim_* helpers */
    if (s->opts.build == ORACLE_BUILD_BINNED) build_view.opts.xform = ORACLE_XFORM_CACHED;
    for (uint32_t m = 0; m < s->blas_count; ++m) blas_build(&build_view, &s->blass[m]);
  }
  s->root = tnode_default();
  if (s->blas_count > 0) tlas_replace_models_recursive(s, &s->root, 0, s->blas_count);
  s->blas_prim_base = (uint32_t*)calloc(s->blas_count ? s->blas_count : 1, 4);
  { uint32_t base = 0; for (uint32_t i = 0; i < s->blas_count; ++i) { s->blas_prim_base[i] = base; base += s->blass[s->blas_nodes[i].blas].prim_count; } s->prim_total = base; }

done_lists:
  free(mesh_nodes.v); free(light_nodes.v); free(models.v);
  free(order);
done:
  if (rc != RAYCA_OK) { oracle_scene_destroy(s); return rc; }
  *out = s;
  return RAYCA_OK;
}

void oracle_scene_destroy(OracleScene* s) {
  if (!s) return;
  for (uint32_t m = 0; m < s->blas_count; ++m) { free(s->blass[m].prims); free(s->blass[m].nodes); }
  free(s->blass); free(s->blas_nodes); free(s->tnodes); free(s->blas_prim_base);
  free(s->local_trs); free(s->world_trs); free(s->has_world); free(s->lights);
  free(s->materials); free(s->textures); free(s->images); free(s->image_bytes); free(s->world_tris);
  free(s);
}

uint32_t oracle_scene_blas_count(const OracleScene* s) { return s->blas_count; }
uint32_t oracle_scene_primitive_count(const OracleScene* s) { return s->prim_total; }
uint32_t oracle_blas_node_count(const OracleScene* s, uint32_t b) { return b < s->blas_count ? s->blass[s->blas_nodes[b].blas].node_count : 0; }
uint32_t oracle_blas_primitive_count(const OracleScene* s, uint32_t b) { return b < s->blas_count ? s->blass[s->blas_nodes[b].blas].prim_count : 0; }
int32_t oracle_blas_nodes(const OracleScene* s, uint32_t b, OracleBvhNode* out, uint32_t cap) {
  if (b >= s->blas_count) return fail(RAYCA_ERR_BAD_ARG, "blas out of range");
  const Blas* bl = &s->blass[s->blas_nodes[b].blas];
  if (cap < bl->node_count) return fail(RAYCA_ERR_BAD_ARG, "capacity too small");
  for (uint32_t i = 0; i < bl->node_count; ++i) {
    const BNode* n = &bl->nodes[i];
    out[i].a[0] = n->bounds.a.x; out[i].a[1] = n->bounds.a.y; out[i].a[2] = n->bounds.a.z; out[i].a[3] = n->bounds.a.w;
    out[i].b[0] = n->bounds.b.x; out[i].b[1] = n->bounds.b.y; out[i].b[2] = n->bounds.b.z; out[i].b[3] = n->bounds.b.w;
    out[i].offset = n->offset; out[i].count = n->count;
  }
  return RAYCA_OK;
}
int32_t oracle_scene_primitive_order(const OracleScene* s, uint32_t* out, uint32_t cap) {
  if (cap < s->prim_total) return fail(RAYCA_ERR_BAD_ARG, "capacity too small");
  uint32_t k = 0;
  for (uint32_t i = 0; i < s->blas_count; ++i) {
    const Blas* bl = &s->blass[s->blas_nodes[i].blas];
    for (uint32_t j = 0; j < bl->prim_count; ++j) out[k++] = bl->prims[j].src;
  }
  return RAYCA_OK;
}
int32_t oracle_scene_world_triangles(const OracleScene* s, float* out, uint32_t cap_tris) {
  if (cap_tris < s->flat_prim_count) return fail(RAYCA_ERR_BAD_ARG, "capacity too small");
  memcpy(out, s->world_tris, (size_t)s->flat_prim_count * 36);
  return RAYCA_OK;
}

/* ============================================================================================ */
/* SoftRenderer::draw pixel loop  rayca-soft/src/scene.rs:101-150                                 */
/* ============================================================================================ */
typedef struct {
  OracleScene* s;
  const RaycaConfig* cfg;
  uint32_t width, height;
  const uint32_t* rows; /* frame rows to render, output row r -> frame row rows[r] */
  uint32_t row_count;
  uint8_t* rgba8;
  float* rgba32f;
  volatile uint32_t next_row;
  pthread_mutex_t mu;
  RaycaStats stats;
  int unsupported;
} Job;

static void render_pixel(Ctx* cx, const Job* job, uint32_t x, uint32_t y, uint32_t out_row) {
  const RaycaConfig* cfg = job->cfg;
  const OracleScene* s = job->s;
  float width = (float)job->width, height = (float)job->height;
  float inv_width = 1.0f / width, inv_height = 1.0f / height;
  float aspectratio = width / height;
  float angle = tanf(s->camera_yfov * 0.5f); /* Camera::get_angle  camera.rs:74-76 */
  const Trs* camera_trs = &s->world_trs[s->camera_node];
  Col color = COL_BLACK;
  float strate_count = sqrtf((float)cfg->samples_per_pixel);
  float offset = 0.5f / strate_count;
  float step = 1.0f / strate_count;
  for (uint32_t i = 0; i < cfg->samples_per_pixel; ++i) {
    float ix = (float)(i % (uint32_t)strate_count);
    float iy = (float)(i / (uint32_t)strate_count);
    float xx = (2.0f * (((float)x + ix * step + offset) * inv_width) - 1.0f) * angle * aspectratio;
    float yy = (1.0f - 2.0f * (((float)y + iy * step + offset) * inv_height)) * angle;
    V4 dir = vnormalize(vec3(xx, yy, -1.0f));
    Ray ray = trs_ray(camera_trs, ray_new(point3(0, 0, 0), dir));
    Col c;
    uint32_t key = oracle_rng_root(cfg->seed, y * job->width + x, i);
    if (!trace(cx, ray, 0, key, &c)) c = COL_BLACK; /* draw_pixel  scene.rs:80-86 */
    color = cadd(color, c); /* color += ...  (AddAssign: rgb += rhs.rgb * rhs.a) */
  }
  color = cdivf(color, (float)cfg->samples_per_pixel);
  color = c_gamma(color, cfg->gamma);
  size_t o = ((size_t)out_row * job->width + x) * 4;
  if (job->rgba32f) { job->rgba32f[o] = color.r; job->rgba32f[o + 1] = color.g; job->rgba32f[o + 2] = color.b; job->rgba32f[o + 3] = color.a; }
  if (job->rgba8) { job->rgba8[o] = to_u8(color.r); job->rgba8[o + 1] = to_u8(color.g); job->rgba8[o + 2] = to_u8(color.b); job->rgba8[o + 3] = to_u8(color.a); }
}
static void* render_worker(void* arg) {
  Job* job = (Job*)arg;
  Ctx cx;
  memset(&cx, 0, sizeof cx);
  cx.scene = job->s; cx.cfg = job->cfg;
  uint64_t primary = 0;
  /* work item = 64 consecutive pixels of one row (finer than rows so that a sparse row sample
   * still keeps every core busy) */
  const uint32_t chunks_per_row = (job->width + 63u) / 64u;
  const uint32_t items = job->row_count * chunks_per_row;
  for (;;) {
    uint32_t it = __sync_fetch_and_add(&job->next_row, 1u);
    if (it >= items) break;
    uint32_t r = it / chunks_per_row, x0 = (it % chunks_per_row) * 64u;
    uint32_t x1 = x0 + 64u < job->width ? x0 + 64u : job->width;
    uint32_t y = job->rows[r];
    for (uint32_t x = x0; x < x1; ++x) render_pixel(&cx, job, x, y, r);
    primary += (uint64_t)(x1 - x0) * job->cfg->samples_per_pixel;
  }
  pthread_mutex_lock(&job->mu);
  job->stats.rays_primary += primary;
  job->stats.rays_shadow += cx.rays_shadow;
  job->stats.rays_bounce += cx.rays_bounce;
  job->stats.boxes_tested += cx.tc.boxes;
  job->stats.triangles_tested += cx.tc.tris;
  job->stats.hits_shaded += cx.hits_shaded;
  if (cx.unsupported) job->unsupported = 1;
  pthread_mutex_unlock(&job->mu);
  return NULL;
}
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static int32_t render_rows_list(OracleScene* s, const RaycaConfig* cfg, uint32_t width, uint32_t height, const uint32_t* rows, uint32_t row_count, uint8_t* rgba8, float* rgba32f, RaycaStats* stats_out, double* seconds_out) {
  if (!s || !cfg) return fail(RAYCA_ERR_BAD_ARG, "null argument");
  if (!s->has_camera) return fail(RAYCA_ERR_NO_CAMERA, "scene has no camera (scene.rs:109)");
  if (s->blas_count == 0) return fail(RAYCA_ERR_EMPTY_SCENE, "empty TLAS (tlas.rs:272)");
  if (width == 0 || height == 0) return fail(RAYCA_ERR_BAD_ARG, "empty image");
  if (cfg->samples_per_pixel == 0 || cfg->light_samples == 0) return fail(RAYCA_ERR_BAD_ARG, "samples must be > 0");
  Job job;
  memset(&job, 0, sizeof job);
  job.s = s; job.cfg = cfg; job.width = width; job.height = height; job.rows = rows; job.row_count = row_count;
  job.rgba8 = rgba8; job.rgba32f = rgba32f;
  pthread_mutex_init(&job.mu, NULL);
  uint32_t nt = s->opts.threads;
  {
    uint64_t items = (uint64_t)row_count * ((width + 63u) / 64u);
    if (nt > items) nt = items ? (uint32_t)items : 1;
  }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nt);
  double t0 = now_s();
  for (uint32_t i = 0; i < nt; ++i) pthread_create(&th[i], NULL, render_worker, &job);
  for (uint32_t i = 0; i < nt; ++i) pthread_join(th[i], NULL);
  double t1 = now_s();
  free(th);
  pthread_mutex_destroy(&job.mu);
  job.stats.rows_rendered = row_count;
  if (stats_out) *stats_out = job.stats;
  if (seconds_out) *seconds_out = t1 - t0;
  if (job.unsupported) return fail(RAYCA_ERR_UNSUPPORTED, "configuration reaches a todo!()/unimplemented!() arm of the reference");
  return RAYCA_OK;
}
static uint32_t tile_rows(const RaycaTile* t, uint32_t height, uint32_t* rows) {
  uint32_t n = 0;
  if (!t || t->parts <= 1) { for (uint32_t y = 0; y < height; ++y) { if (rows) rows[n] = y; n++; } return n; }
  uint32_t band = t->band_rows ? t->band_rows : 1;
  for (uint32_t y = 0; y < height; ++y) if ((y / band) % t->parts == t->part) { if (rows) rows[n] = y; n++; }
  return n;
}
int32_t oracle_render(OracleScene* s, const RaycaConfig* cfg, uint32_t width, uint32_t height, const RaycaTile* tile, uint8_t* rgba8, float* rgba32f, RaycaStats* stats_out, double* seconds_out) {
  uint32_t* rows = (uint32_t*)malloc(4 * (height ? height : 1));
  uint32_t n = tile_rows(tile, height, rows);
  int32_t rc = render_rows_list(s, cfg, width, height, rows, n, rgba8, rgba32f, stats_out, seconds_out);
  free(rows);
  return rc;
}
int32_t oracle_render_rows(OracleScene* s, const RaycaConfig* cfg, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, uint8_t* rgba8, float* rgba32f, RaycaStats* stats_out, double* seconds_out) {
  if (row_end > height || row_begin > row_end) return fail(RAYCA_ERR_BAD_ARG, "bad row range");
  uint32_t n = row_end - row_begin;
  uint32_t* rows = (uint32_t*)malloc(4 * (n ? n : 1));
  for (uint32_t i = 0; i < n; ++i) rows[i] = row_begin + i;
  int32_t rc = render_rows_list(s, cfg, width, height, rows, n, rgba8, rgba32f, stats_out, seconds_out);
  free(rows);
  return rc;
}
int32_t oracle_trace_rays(OracleScene* s, uint32_t count, const float* rays, float* t_out, uint32_t* prim_out, float* uv_out, RaycaStats* stats_out) {
  if (!s || !rays) return fail(RAYCA_ERR_BAD_ARG, "null argument");
  if (s->blas_count == 0) return fail(RAYCA_ERR_EMPTY_SCENE, "empty TLAS (tlas.rs:272)");
  TraceCount tc = {0, 0};
  for (uint32_t i = 0; i < count; ++i) {
    const float* r = rays + 6 * (size_t)i;
    Ray ray = ray_new(point3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]));
    Hit h;
    if (tnode_intersects(s, &s->root, &ray, &h, &tc)) {
      if (t_out) t_out[i] = h.depth;
      if (prim_out) prim_out[i] = s->blas_prim_base[h.blas] + h.primitive;
      if (uv_out) { uv_out[2 * i] = h.uv.x; uv_out[2 * i + 1] = h.uv.y; }
    } else {
      if (t_out) t_out[i] = FLT_MAX;
      if (prim_out) prim_out[i] = RAYCA_NONE;
      if (uv_out) { uv_out[2 * i] = 0.0f; uv_out[2 * i + 1] = 0.0f; }
    }
  }
  if (stats_out) { memset(stats_out, 0, sizeof *stats_out); stats_out->rays_primary = count; stats_out->boxes_tested = tc.boxes; stats_out->triangles_tested = tc.tris; }
  return RAYCA_OK;
}

/* ============================================================================================ */
/* known-answer hooks                                                                            */
/* ============================================================================================ */
static void out3(V4 v, float o[3]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void oracle_vec3_rotate(const float v[3], const float q[4], float o[3]) { out3(vrotate(vec3(v[0], v[1], v[2]), v4(q[0], q[1], q[2], q[3])), o); }
void oracle_vec3_normalize(const float v[3], float o[3]) { out3(vnormalize(vec3(v[0], v[1], v[2])), o); }
void oracle_vec3_reciprocal(const float v[3], float o[3]) { out3(vreciprocal(vec3(v[0], v[1], v[2])), o); }
void oracle_vec3_reflect(const float v[3], const float n[3], float o[3]) { out3(vreflect(vec3(v[0], v[1], v[2]), vec3(n[0], n[1], n[2])), o); }
void oracle_vec3_cross(const float a[3], const float b[3], float o[3]) { out3(vcross(vec3(a[0], a[1], a[2]), vec3(b[0], b[1], b[2])), o); }
float oracle_vec3_dot(const float a[3], const float b[3]) { return dot4(vec3(a[0], a[1], a[2]), vec3(b[0], b[1], b[2])); }
int32_t oracle_vec3_close(const float a[3], const float b[3]) { return vclose(vec3(a[0], a[1], a[2]), vec3(b[0], b[1], b[2])); }
void oracle_quat_mul(const float a[4], const float b[4], float o[4]) { V4 r = qmul(v4(a[0], a[1], a[2], a[3]), v4(b[0], b[1], b[2], b[3])); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
void oracle_trs_mul(const RaycaTrs* a, const RaycaTrs* b, RaycaTrs* o) { Trs ta = trs_from_abi(a), tb = trs_from_abi(b); Trs r = trs_mul(&ta, &tb); trs_to_abi(&r, o); }
void oracle_trs_point(const RaycaTrs* t, const float p[3], float o[3]) { Trs tt = trs_from_abi(t); out3(trs_point(&tt, point3(p[0], p[1], p[2])), o); }
void oracle_trs_vec(const RaycaTrs* t, const float v[3], float o[3]) { Trs tt = trs_from_abi(t); out3(trs_vec(&tt, vec3(v[0], v[1], v[2])), o); }
void oracle_inv_trs_vec(const RaycaTrs* t, const float v[3], float o[3]) { /* Mul<Vec3> for &Inversed<Trs>  trs.rs:394-403 */
  Trs tt = trs_from_abi(t);
  V4 r = vadd(vec3(v[0], v[1], v[2]), vneg(tt.translation));
  r = vrotate(r, qconj(tt.rotation));
  r = vmul(r, vreciprocal(tt.scale));
  out3(r, o);
}
void oracle_trs_ray(const RaycaTrs* t, const float origin[3], const float dir[3], float o[9]) {
  Trs tt = trs_from_abi(t);
  Ray r = trs_ray(&tt, ray_new(point3(origin[0], origin[1], origin[2]), vec3(dir[0], dir[1], dir[2])));
  out3(r.origin, o); out3(r.dir, o + 3); out3(r.rdir, o + 6);
}
void oracle_rgba8_from_color(const float c[4], uint8_t o[4]) { for (int i = 0; i < 4; ++i) o[i] = to_u8(c[i]); }
void oracle_color_add(const float a[4], const float b[4], float o[4]) { Col r = cadd(col4(a), col4(b)); o[0] = r.r; o[1] = r.g; o[2] = r.b; o[3] = r.a; }
int32_t oracle_triangle_intersects(const float v[9], const RaycaTrs* trs, const float origin[3], const float dir[3], float* t, float uv[2], float point[3]) {
  Trs tt = trs_from_abi(trs);
  V4 p[3] = {point3(v[0], v[1], v[2]), point3(v[3], v[4], v[5]), point3(v[6], v[7], v[8])};
  Ray ray = ray_new(point3(origin[0], origin[1], origin[2]), vec3(dir[0], dir[1], dir[2]));
  Hit h;
  if (!triangle_intersects(p, &tt, &ray, &h)) return 0;
  if (t) *t = h.depth;
  if (uv) { uv[0] = h.uv.x; uv[1] = h.uv.y; }
  if (point) out3(h.point, point);
  return 1;
}
int32_t oracle_sphere_intersects(const float center[3], float radius, const RaycaTrs* trs, const float origin[3], const float dir[3], float* t, float point[3]) {
  Trs tt = trs_from_abi(trs);
  Ray ray = ray_new(point3(origin[0], origin[1], origin[2]), vec3(dir[0], dir[1], dir[2]));
  Hit h;
  if (!sphere_intersects(point3(center[0], center[1], center[2]), radius * radius, &tt, &ray, &h)) return 0;
  if (t) *t = h.depth;
  if (point) out3(h.point, point);
  return 1;
}
float oracle_aabb_intersects(const float a[3], const float b[3], const float origin[3], const float dir[3]) {
  AABB bx = {point3(a[0], a[1], a[2]), point3(b[0], b[1], b[2])};
  Ray ray = ray_new(point3(origin[0], origin[1], origin[2]), vec3(dir[0], dir[1], dir[2]));
  return aabb_intersects(&bx, &ray);
}
/* Triangle::{get_centroid,min,max}(trs)  triangle.rs:160-177 through the functions the scene path uses */
void oracle_triangle_bounds(const float v[9], const RaycaTrs* trs, float centroid[3], float mn[3], float mx[3]) {
  Trs tt = trs_from_abi(trs);
  V4 p[3] = {point3(v[0], v[1], v[2]), point3(v[3], v[4], v[5]), point3(v[6], v[7], v[8])};
  out3(trs_vec(&tt, tri_model_centroid(p)), centroid);
  out3(tri_min(p, &tt), mn);
  out3(tri_max(p, &tt), mx);
}
/* Sphere::{get_centroid,min,max}(trs)  sphere.rs:164-178 */
void oracle_sphere_bounds(const float center[3], float radius, const RaycaTrs* trs, float centroid[3], float mn[3], float mx[3]) {
  Trs tt = trs_from_abi(trs);
  Prim pr;
  memset(&pr, 0, sizeof pr);
  pr.kind = RAYCA_GEOMETRY_SPHERE;
  pr.center = point3(center[0], center[1], center[2]);
  pr.radius = radius;
  pr.radius2 = radius * radius;
  float r = sphere_world_radius(&pr, &tt);
  V4 c = trs_point(&tt, pr.center);
  out3(c, centroid);
  out3(sub4(c, vec3(r, r, r)), mn);
  out3(add4(c, vec3(r, r, r)), mx);
}
/* BvhPrimitive::intersects(scene, ray)  primitive.rs:95-101 for the primitive whose flatten-order index is `src`:
 * no BVH in front of it, as in the reference's own test (bvh/triangle.rs:83-114).  1 = hit, 0 = miss, -1 = no such primitive */
int32_t oracle_scene_primitive_intersects(const OracleScene* s, uint32_t src, const float origin[3], const float dir[3], float* t, float uv[2]) {
  Ray ray = ray_new(point3(origin[0], origin[1], origin[2]), vec3(dir[0], dir[1], dir[2]));
  for (uint32_t b = 0; b < s->blas_count; ++b) {
    const Blas* bl = &s->blass[b];
    for (uint32_t i = 0; i < bl->prim_count; ++i) {
      if (bl->prims[i].src != src) continue;
      Hit h;
      if (!prim_intersects(s, &bl->prims[i], &ray, &h)) return 0;
      if (t) *t = h.depth;
      if (uv) { uv[0] = h.uv.x; uv[1] = h.uv.y; }
      return 1;
    }
  }
  return -1;
}
/* ---- Mat4 / Quat known-answer hooks (rayca-math/src/mat4.rs:321-421, quat.rs:301-396); matrices are 16 floats, row-major */
static M4 m4_in(const float m[16]) { M4 r; memcpy(r.m, m, 64); return r; }
void oracle_mat4_identity(float o[16]) { M4 r = m4_identity(); memcpy(o, r.m, 64); }
void oracle_mat4_mul(const float a[16], const float b[16], float o[16]) { M4 x = m4_in(a), y = m4_in(b); M4 r = m4_mul(&x, &y); memcpy(o, r.m, 64); }
void oracle_mat4_from_scale(const float s[3], float o[16]) { M4 r = m4_from_scale(vec3(s[0], s[1], s[2])); memcpy(o, r.m, 64); }
void oracle_mat4_from_translation(const float t[3], float o[16]) { M4 r = m4_from_translation(vec3(t[0], t[1], t[2])); memcpy(o, r.m, 64); }
void oracle_mat4_transpose(const float m[16], float o[16]) { M4 x = m4_in(m); M4 r = m4_transpose(&x); memcpy(o, r.m, 64); }
void oracle_mat4_look_at(const float target[3], const float eye[3], const float up[3], float o[16]) {
  M4 r = m4_look_at(vec3(target[0], target[1], target[2]), vec3(eye[0], eye[1], eye[2]), vec3(up[0], up[1], up[2]));
  memcpy(o, r.m, 64);
}
void oracle_mat4_get_rotation(const float m[16], float o[4]) { M4 x = m4_in(m); V4 q = q_from_m4(&x); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
void oracle_mat4_mul_vec3(const float m[16], const float v[3], float o[3]) { M4 x = m4_in(m); out3(m4_vec(&x, vec3(v[0], v[1], v[2])), o); }
void oracle_mat4_mul_point3(const float m[16], const float p[3], float o[3]) { M4 x = m4_in(m); out3(m4_point(&x, point3(p[0], p[1], p[2])), o); }
void oracle_quat_axis_angle(const float axis[3], float angle, float o[4]) { V4 q = q_axis_angle(vec3(axis[0], axis[1], axis[2]), angle); o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }
void oracle_quat_conjugate(const float q[4], float o[4]) { V4 r = qconj(v4(q[0], q[1], q[2], q[3])); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
void oracle_quat_normalize(const float q[4], float o[4]) { V4 r = qnormalize(v4(q[0], q[1], q[2], q[3])); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
int32_t oracle_quat_is_normalized(const float q[4]) { return q_is_normalized(v4(q[0], q[1], q[2], q[3])); }
float oracle_quat_dot(const float a[4], const float b[4]) { return dot4(v4(a[0], a[1], a[2], a[3]), v4(b[0], b[1], b[2], b[3])); }
float oracle_quat_len(const float q[4]) { return qlen(v4(q[0], q[1], q[2], q[3])); }
/* Vec3 arithmetic and min/max  vec3.rs:556-573 */
void oracle_vec3_arith(const float a[3], const float b[3], float s, float add[3], float sub[3], float mul[3], float div[3], float neg[3]) {
  V4 x = vec3(a[0], a[1], a[2]), y = vec3(b[0], b[1], b[2]);
  out3(vadd(x, y), add); out3(vsub(y, x), sub); out3(vscale(x, s), mul); out3(vdivf(y, s), div); out3(vneg(x), neg);
}
void oracle_vec3_min_max(const float a[3], const float b[3], float mn[3], float mx[3]) {
  V4 x = vec3(a[0], a[1], a[2]), y = vec3(b[0], b[1], b[2]);
  out3(min4(x, y), mn); out3(max4(x, y), mx);
}
