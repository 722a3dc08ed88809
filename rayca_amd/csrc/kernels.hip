// kernels.hip -- hand-written gfx950 kernels of the rayca hot path.
//
//   k_generation<..>  one persistent-wavefront kernel per ray generation:
//                     generation 0 = camera rays (rayca-soft/src/scene.rs:117-150),
//                     generation g = the g-th bounce of Pathtracer::trace_impl
//                     (integrator/pathtracer.rs:68-106).  Each wave pulls batches of 64 rays from a
//                     per-XCD work counter, traverses the BVH with a per-lane node stack in LDS
//                     (TlasNode/BvhNode::intersects, bvh/tlas.rs:136-180, bvh/blas.rs:129-177),
//                     intersects triangles (rayca-geometry/src/triangle.rs:84-159), shades the hit
//                     (hit.rs, brdf/ggx.rs, sampler/nee.rs, sampler/cosine.rs) and compacts the
//                     surviving rays of the next generation with __ballot/__popcll/__shfl.
//   k_resolve         folds the per-depth records back in the reference's evaluation order,
//                     accumulates samples, applies gamma and quantises (scene.rs:146-148).
//   k_trace_rays      Tlas::intersects for caller-supplied rays (parity/debug entry).
//
// No MFMA anywhere: there is no dense contraction in this path.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace rayca {
namespace {

constexpr int kBlock = 256;
#ifndef RAYCA_TRACE_MIN_WAVES
#define RAYCA_TRACE_MIN_WAVES 1
#endif
#ifndef RAYCA_MIN_WAVES_FLAT
#define RAYCA_MIN_WAVES_FLAT 5
#endif
#ifndef RAYCA_MIN_WAVES
#define RAYCA_MIN_WAVES 4
#endif

struct DRay {
  F4 o, d, rd;  // origin (w=1), direction (w=0), zero-safe reciprocal
};
struct DHit {
  float t;
  uint32_t prim;
  float u, v;
};

__device__ __forceinline__ DRay make_ray(F4 origin, F4 dir) {  // Ray::new  ray.rs:63-72
  DRay r;
  r.rd = reciprocal(dir);
  origin.w = 1.0f;
  r.o = origin;
  r.d = dir;
  return r;
}

// ---- geometry --------------------------------------------------------------------------------
// AABB::intersects  bvh/aabb.rs:74-93.  (a-o)*rdir goes through Point3::scale = fma(x, s, 0).
__device__ __forceinline__ bool slab(float ax, float ay, float az, float bx, float by, float bz, const DRay& r, float& tmin_out) {
  const float t1x = __fmaf_rn(ax - r.o.x, r.rd.x, 0.0f), t2x = __fmaf_rn(bx - r.o.x, r.rd.x, 0.0f);
  const float t1y = __fmaf_rn(ay - r.o.y, r.rd.y, 0.0f), t2y = __fmaf_rn(by - r.o.y, r.rd.y, 0.0f);
  const float t1z = __fmaf_rn(az - r.o.z, r.rd.z, 0.0f), t2z = __fmaf_rn(bz - r.o.z, r.rd.z, 0.0f);
  const float tmax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fminf(fmaxf(t1z, t2z), FLT_MAX));
  const float tmin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), -FLT_MAX));
  tmin_out = tmin;
  return tmax >= tmin && tmax > 0.0f;
}

// Conservative box test for trees whose boxes are padded (RAYCA_BUILDER_SAH): t = b*rd - o*rd with one
// FMA per plane.  Not the reference's rounding -- it does not have to be: in that mode a triangle is a
// candidate iff the reference's own leaf box passes `slab` (reference_candidate), this test only steers
// the search and the padding of the boxes dominates its rounding error.
struct FastRay {
  float mx, my, mz, cx, cy, cz;  // plane distance t = coordinate * m + c
};
// f32 boxes: t = b*rd - o*rd.  fp16 boxes hold (x - centre) * scale: t = h * (rd / scale) + (centre - o) * rd.
__device__ __forceinline__ FastRay make_fast(const DevScene& sc, const DRay& r, bool half) {
  FastRay f;
  if (half) {
    f.mx = r.rd.x * sc.half_inv_scale; f.my = r.rd.y * sc.half_inv_scale; f.mz = r.rd.z * sc.half_inv_scale;
    f.cx = (sc.half_center[0] - r.o.x) * r.rd.x; f.cy = (sc.half_center[1] - r.o.y) * r.rd.y; f.cz = (sc.half_center[2] - r.o.z) * r.rd.z;
  } else {
    f.mx = r.rd.x; f.my = r.rd.y; f.mz = r.rd.z;
    f.cx = -(r.o.x * r.rd.x); f.cy = -(r.o.y * r.rd.y); f.cz = -(r.o.z * r.rd.z);
  }
  return f;
}
// fp16 -> f32 is exact and free: fmaf((float)half, a, b) is one v_fma_mix_f32
typedef _Float16 rc_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lo16(uint32_t w) { return (float)__builtin_bit_cast(rc_h2, w).x; }
__device__ __forceinline__ float hi16(uint32_t w) { return (float)__builtin_bit_cast(rc_h2, w).y; }
__device__ __forceinline__ bool slab_fast(float ax, float ay, float az, float bx, float by, float bz, const FastRay& f, float& tmin_out) {
  const float t1x = __fmaf_rn(ax, f.mx, f.cx), t2x = __fmaf_rn(bx, f.mx, f.cx);
  const float t1y = __fmaf_rn(ay, f.my, f.cy), t2y = __fmaf_rn(by, f.my, f.cy);
  const float t1z = __fmaf_rn(az, f.mz, f.cz), t2z = __fmaf_rn(bz, f.mz, f.cz);
  const float tmax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
  const float tmin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
  tmin_out = tmin;
  return tmax >= tmin && tmax > 0.0f;
}
// the same arithmetic on decoded fp16 planes (make_fast folds centre and scale into m and c)
__device__ __forceinline__ bool slab_half(float ax, float ay, float az, float bx, float by, float bz, const FastRay& f, float& tmin_out) {
  return slab_fast(ax, ay, az, bx, by, bz, f, tmin_out);
}

// Triangle::intersects  rayca-geometry/src/triangle.rs:84-159 on world-space vertices (identical
// bits to `trs * vertex`, computed once on the host with the same operation sequence).
__device__ __forceinline__ bool tri_test(F4 v0, F4 v1, F4 v2, const DRay& r, float& t_out, float& u_out, float& v_out) {
  const F4 v0v1 = v1 - v0;
  const F4 v0v2 = v2 - v0;
  const F4 n = cross(v0v1, v0v2);
  if (dot(r.d, n) > 0.0f) return false;  // back-face cull
  const float denom = dot(n, n);
  const float ndd = dot(n, r.d);
  if (fabsf(ndd) < FLT_EPSILON) return false;
  const float d = -dot(n, v0);
  const float t = -(dot(n, to_vec(r.o)) + d) / ndd;
  if (t < 0.0f) return false;
  const F4 p = r.o + r.d * t;
  F4 c = cross(v1 - v0, to_vec(p - v0));
  if (dot(n, c) < 0.0f) return false;
  c = cross(v2 - v1, to_vec(p - v1));
  const float u = dot(n, c);
  if (u < 0.0f) return false;
  c = cross(v0 - v2, to_vec(p - v2));
  const float v = dot(n, c);
  if (v < 0.0f) return false;
  t_out = t;
  u_out = u / denom;
  v_out = v / denom;
  return true;
}

// Sphere::intersects  rayca-geometry/src/sphere.rs:101-163: the ray is taken to model space by
// Inversed<&Trs> (translate -T, rotate R^-1, scale 1/S; trs.rs:405-414), intersected, and the hit point is
// brought back by the world Trs.  A sphere occupies one primitive slot whose first float is NaN and
// whose second float carries the sphere index.
__device__ __forceinline__ DRay sphere_local_ray(const DevSphere& sp, const DRay& r) {
  F4 o = r.o + (-f4(sp.translation[0], sp.translation[1], sp.translation[2], 0.0f));  // Ray::translate
  F4 d = r.d;
  const F4 iq = f4(sp.inv_rotation[0], sp.inv_rotation[1], sp.inv_rotation[2], sp.inv_rotation[3]);
  d = rotate(d, iq);  // Ray::rotate
  o = point_rotate(o, iq);
  o.w = 1.0f;
  const F4 is = f4(sp.inv_scale[0], sp.inv_scale[1], sp.inv_scale[2], sp.inv_scale[3]);
  d = d * is;  // Ray::scale
  o = point_scale(o, is);
  DRay l;
  l.o = o;
  l.d = d;
  l.rd = reciprocal(d);
  return l;
}
__device__ __forceinline__ bool sphere_test(const DevSphere& sp, const DRay& r, float& t_out) {
  const DRay l = sphere_local_ray(sp, r);
  const F4 center = f4(sp.center[0], sp.center[1], sp.center[2], sp.center[3]);
  const float a = dot(l.d, l.d);
  const F4 c_to_r = as_vec(l.o - center);
  const float b = dot(c_to_r, l.d);
  const float c = dot(c_to_r, c_to_r) - sp.radius2;
  const float det = b * b - a * c;
  if (det < 0.0f) return false;
  const float det_sqrt = sqrtf(det);
  const float t0 = (-b + det_sqrt) / a;
  const float t1 = (-b - det_sqrt) / a;
  if (t0 < 0.0f && t1 < 0.0f) return false;
  float t;
  if (t0 >= 0.0f && t1 >= 0.0f) t = fminf(t0, t1);
  else if (t0 >= 0.0f) t = t0;
  else t = t1;
  t_out = t;
  return true;
}

// World-space triangle i: nine f32, 36-B stride, in leaf order.  (Compile with -DRAYCA_TRI_SOA to
// read nine SoA planes instead -- kept only for the layout A/B in DESIGN.md section 3.)
struct __attribute__((packed, aligned(4))) Tri9 {
  float v[9];
};
__device__ __forceinline__ void load_tri(const DevScene& sc, uint32_t i, F4& v0, F4& v1, F4& v2) {
#if defined(RAYCA_TRI_SOA)
  const float* p = sc.tris + i;
  const size_t n = sc.prim_count;
  v0 = vec3(p[0], p[n], p[2 * n]);
  v1 = vec3(p[3 * n], p[4 * n], p[5 * n]);
  v2 = vec3(p[6 * n], p[7 * n], p[8 * n]);
#else
  const Tri9 t = *reinterpret_cast<const Tri9*>(sc.tris + 9ull * i);
  v0 = vec3(t.v[0], t.v[1], t.v[2]);
  v1 = vec3(t.v[3], t.v[4], t.v[5]);
  v2 = vec3(t.v[6], t.v[7], t.v[8]);
#endif
}

// exact depth tie: the primitive that comes first in the reference's order wins (strict `<` in a
// left-to-right DFS, blas.rs:151,161,169)
__device__ __forceinline__ bool tie_before(const DevScene& sc, uint32_t a, uint32_t b) {
  if (b == RAYCA_NONE) return true;
  if (sc.tie_rank) return sc.tie_rank[a] < sc.tie_rank[b];
  return a < b;
}

// RAYCA_BUILDER_SAH: the reference only ever tests triangle i if the slab test passes for every box
// above it in ITS tree; those boxes are nested and the slab arithmetic is monotone in the corners, so
// that is equivalent to the slab test of the reference LEAF that holds i (bvh/blas.rs:136-139).
__device__ __forceinline__ bool reference_candidate(const DevScene& sc, uint32_t i, const DRay& r) {
  if (!sc.ref_leaf_of) return true;
  const float4* b = sc.ref_leaf_boxes + 2ull * sc.ref_leaf_of[i];
  const float4 lo = b[0], hi = b[1];
  float tmin;
  return slab(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, r, tmin);
}

struct LaneCounters {
  uint32_t boxes = 0, tris = 0;
  // STATS only: what the lock-step wave pays -- every trip of the node loop / the leaf loop costs all 64 lanes,
  // booked by the first lane that takes the trip
  unsigned long long slot_boxes = 0, slot_tris = 0;
  __device__ __forceinline__ bool books() const { return __lane_id() == (uint32_t)__ffsll((long long)__ballot(1)) - 1u; }
};

constexpr uint32_t kTerminated = 0x7FFFFFFFu;  // "no node left": an inner index that never exists (== kNoChild)

// Per-lane node stack: the first `lds_entries` entries in LDS (entry-major: entry e of thread t at
// lds[e*kBlock + t], so the 64 lanes of a wave always hit 64 different banks whatever their depths),
// deeper entries in a global spill area laid out the same way.  The spill branch is cold: the LDS part
// is sized from the tree on the host and covers every traversal of ordinary trees.
template <bool SPILL>
struct NodeStack {
  uint32_t* lds;
  uint32_t* ovf;       // wave-uniform base of the spill area (a per-lane pointer would be two more registers in every loop)
  uint32_t gthread;    // this lane's column in it
  uint32_t lds_entries, ovf_stride, sp;
  // the newest entry lives in a register: a pop hands it out at once and fetches its successor from LDS behind
  // the node load that follows, instead of in front of it (-3 % on 5-deep paths, neutral on camera rays)
  uint32_t top;
  __device__ __forceinline__ void clear() { sp = 0; top = kTerminated; }
  __device__ __forceinline__ void put(uint32_t i, uint32_t v) {
    if (!SPILL || i < lds_entries) lds[i * kBlock] = v;
    else ovf[(size_t)(i - lds_entries) * ovf_stride + gthread] = v;
  }
  __device__ __forceinline__ uint32_t get(uint32_t i) const {
    return (!SPILL || i < lds_entries) ? lds[i * kBlock] : ovf[(size_t)(i - lds_entries) * ovf_stride + gthread];
  }
  __device__ __forceinline__ void push(uint32_t v) {
    if (top != kTerminated) put(sp++, top);
    top = v;
  }
  __device__ __forceinline__ uint32_t pop() {
    const uint32_t r = top;
    if (sp == 0) top = kTerminated;
    else top = get(--sp);
    return r;
  }
};
template <bool SPILL>
__device__ __forceinline__ NodeStack<SPILL> make_stack(uint32_t* lds_base, const TraceLaunch& tl, uint32_t global_thread) {
  NodeStack<SPILL> st;
  st.lds = lds_base + threadIdx.x;
  st.ovf = tl.ovf;
  st.gthread = global_thread;
  st.lds_entries = tl.lds_entries;
  st.ovf_stride = tl.ovf_stride;
  st.clear();
  return st;
}

// leaf: test primitives [first, first+count) -- shared by both node formats
template <bool ORDERED, bool SPH, bool STATS>
__device__ __forceinline__ void test_leaf(const DevScene& sc, const DRay& r, uint32_t ref, float t_stop, DHit& hit, float& limit,
                                          LaneCounters& cnt) {
  const uint32_t first = ref & kLeafFirstMask;
  const uint32_t count = ((ref >> 25) & 63u) + 1u;
  for (uint32_t i = first; i < first + count; ++i) {
    F4 v0, v1, v2;
    load_tri(sc, i, v0, v1, v2);
    if (STATS) {
      cnt.tris++;
      if (cnt.books()) cnt.slot_tris += 64ull;
    }
    float t, u, v;
    bool got;
    if (SPH && v0.x != v0.x) {  // sphere slot
      u = v = 0.0f;
      got = sphere_test(sc.spheres[__float_as_uint(v0.y)], r, t);
    } else {
      got = tri_test(v0, v1, v2, r, t, u, v);
    }
    if (got) {
      if ((t < hit.t || (t == hit.t && tie_before(sc, i, hit.prim))) && reference_candidate(sc, i, r)) {
        hit.t = t;
        hit.prim = i;
        hit.u = u;
        hit.v = v;
        if (ORDERED) {
          const float b = fminf(t, t_stop);
          limit = b + fabsf(b) * 9.765625e-4f + sc.cull_abs;
        }
      }
    }
  }
}

// One trip through the node loop: test the children of inner node `cur`, push what stays pending, return the
// next node (inner, leaf or kTerminated).
template <bool ORDERED, bool FAST, bool WIDE, bool SPILL, bool STATS, bool HALF>
__device__ __forceinline__ uint32_t node_step(const DevScene& sc, const DRay& r, const FastRay& fr, float limit, uint32_t cur, NodeStack<SPILL>& st,
                                              LaneCounters& cnt) {
  if (STATS && cnt.books()) cnt.slot_boxes += WIDE ? 256ull : 128ull;
  if (WIDE) {
    float key[4];
    uint32_t ref[4];
    if (FAST && HALF) {
      const uint4* np = sc.nodes4_h + 4ull * cur;
      const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
      if (STATS) cnt.boxes += 4;
      ref[0] = q3.x; ref[1] = q3.y; ref[2] = q3.z; ref[3] = q3.w;
      const float ax[4] = {lo16(q0.x), hi16(q0.x), lo16(q0.y), hi16(q0.y)}, ay[4] = {lo16(q0.z), hi16(q0.z), lo16(q0.w), hi16(q0.w)};
      const float az[4] = {lo16(q1.x), hi16(q1.x), lo16(q1.y), hi16(q1.y)}, bx[4] = {lo16(q1.z), hi16(q1.z), lo16(q1.w), hi16(q1.w)};
      const float by[4] = {lo16(q2.x), hi16(q2.x), lo16(q2.y), hi16(q2.y)}, bz[4] = {lo16(q2.z), hi16(q2.z), lo16(q2.w), hi16(q2.w)};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float tc;
        bool h = slab_half(ax[c], ay[c], az[c], bx[c], by[c], bz[c], fr, tc);
        if (ORDERED) h = h && tc <= limit;
        key[c] = h ? (ORDERED ? tc : (float)c) : INFINITY;
      }
    } else {
    const float4* np = sc.nodes4 + 8ull * cur;
    const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], cr = np[6];
    if (STATS) cnt.boxes += 4;
    ref[0] = __float_as_uint(cr.x); ref[1] = __float_as_uint(cr.y); ref[2] = __float_as_uint(cr.z); ref[3] = __float_as_uint(cr.w);
    const float ax[4] = {lx.x, lx.y, lx.z, lx.w}, ay[4] = {ly.x, ly.y, ly.z, ly.w}, az[4] = {lz.x, lz.y, lz.z, lz.w};
    const float bx[4] = {hx.x, hx.y, hx.z, hx.w}, by[4] = {hy.x, hy.y, hy.z, hy.w}, bz[4] = {hz.x, hz.y, hz.z, hz.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float tc;
      bool h = FAST ? slab_fast(ax[c], ay[c], az[c], bx[c], by[c], bz[c], fr, tc) : slab(ax[c], ay[c], az[c], bx[c], by[c], bz[c], r, tc);
      if (ORDERED) h = h && tc <= limit;
      // sort key: entry distance (ORDERED) or the child's index (reference order); misses sort last
      key[c] = h ? (ORDERED ? tc : (float)c) : INFINITY;
    }
    }
    // 5-comparator network, strict `>` so equal keys keep their index order (the chains that
    // split big leaves rely on it: see WideBuilder)
#define RC_CE(i, j)                                  \
  {                                                  \
const bool sw = key[i] > key[j];                 \
const float ka = sw ? key[j] : key[i];           \
const float kb = sw ? key[i] : key[j];           \
const uint32_t ra = sw ? ref[j] : ref[i];        \
const uint32_t rb = sw ? ref[i] : ref[j];        \
key[i] = ka; key[j] = kb; ref[i] = ra; ref[j] = rb; \
  }
    RC_CE(0, 1) RC_CE(2, 3) RC_CE(0, 2) RC_CE(1, 3) RC_CE(1, 2)
#undef RC_CE
    // farthest first, so the nearest pending sibling is popped first
    if (key[3] < INFINITY) st.push(ref[3]);
    if (key[2] < INFINITY) st.push(ref[2]);
    if (key[1] < INFINITY) st.push(ref[1]);
    cur = key[0] < INFINITY ? ref[0] : st.pop();
  } else {
    float tl, tr;
    bool hl, hr;
    uint32_t lref, rref;
    if (FAST && HALF) {
      const uint4* nh = sc.nodes_h + 2ull * cur;
      const uint4 a = nh[0], b = nh[1];
      hl = slab_half(lo16(a.x), hi16(a.x), lo16(a.y), hi16(a.y), lo16(a.z), hi16(a.z), fr, tl);
      hr = slab_half(lo16(a.w), hi16(a.w), lo16(b.x), hi16(b.x), lo16(b.y), hi16(b.y), fr, tr);
      lref = b.z;
      rref = b.w;
    } else {
      const float4* np = sc.nodes + 4ull * cur;
      const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
      if (FAST) {
        hl = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, fr, tl);
        hr = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, fr, tr);
      } else {
        hl = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, tl);
        hr = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r, tr);
      }
      lref = __float_as_uint(q3.x);
      rref = __float_as_uint(q3.y);
    }
    if (STATS) cnt.boxes += 2;
    if (ORDERED) {
      hl = hl && tl <= limit;
      hr = hr && tr <= limit;
    }
    if (hl && hr) {
      const bool left_first = !ORDERED || tl <= tr;
      st.push(left_first ? rref : lref);
      cur = left_first ? lref : rref;
    } else if (hl) {
      cur = lref;
    } else if (hr) {
      cur = rref;
    } else {
      cur = st.pop();
    }
  }
  return cur;
}

// Closest hit along `r` (Tlas::intersects).
//   ORDERED    front-to-back descent, subtrees whose entry distance exceeds the current best are
//              skipped.  The winner is the (t, reference order) lexicographic minimum, which is what
//              the reference's strict-< DFS returns (blas.rs:151,161,169).  The skip test carries a
//              slack (relative 2^-10 plus sc.cull_abs) because a triangle's t and its box's slab
//              entry are computed by different expressions and may disagree in the last bits.
//   !ORDERED   visits every leaf the reference visits (no culling), children in the reference's order.
//   FAST       conservative FMA slabs (only with the reference-leaf filter, see slab_fast).
//   WIDE       4-wide nodes: one 128-B fetch tests four boxes and skips every other level of the
//              binary tree (legal because a box that passes implies its ancestors pass).
//   t_stop     any-hit early out: stop as soon as a hit with t < t_stop is found (shadow rays,
//              "occluded iff closest depth < light distance", nee.rs:152-156).  FLT_MAX = never.
// Structure: "while-while" -- all lanes of the wave first descend inner nodes until each holds a leaf
// (or has finished), then the leaf lanes run the triangle tests together, so the expensive leaf code
// is not serialised against node steps of other lanes.  (A speculative variant -- a lane sets its first leaf
// aside and keeps descending until it holds a second -- was measured: soup -5.6 %, atrium +4 % to +10 %: the
// later leaf test delays the cull bound.  Not kept.)
template <bool ORDERED, bool FAST, bool SPH, bool WIDE, bool SPILL, bool STATS, bool HALF = false>
__device__ __forceinline__ bool trace(const DevScene& sc, const DRay& r, float t_stop, NodeStack<SPILL>& st, DHit& hit, LaneCounters& cnt) {
  const FastRay fr = make_fast(sc, r, HALF);
  hit.t = INFINITY;
  hit.prim = RAYCA_NONE;
  hit.u = hit.v = 0.0f;
  float tmin;
  if (STATS) cnt.boxes++;
  uint32_t cur = WIDE ? sc.root_ref4 : sc.root_ref;
  if (!slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], r, tmin)) cur = kTerminated;
  st.clear();
  const bool any_hit = ORDERED && t_stop < FLT_MAX;
  float limit = INFINITY;  // cull bound (ORDERED only)
  if (any_hit) limit = t_stop + fabsf(t_stop) * 9.765625e-4f + sc.cull_abs;
  while (cur != kTerminated) {
    while (!(cur & kLeafFlag) && cur != kTerminated) {
      cur = node_step<ORDERED, FAST, WIDE, SPILL, STATS, HALF>(sc, r, fr, limit, cur, st, cnt);
    }
    if (cur != kTerminated) {  // a leaf
      test_leaf<ORDERED, SPH, STATS>(sc, r, cur, t_stop, hit, limit, cnt);
      if (any_hit && hit.t < t_stop) break;
      cur = st.pop();
    }
  }
  return hit.prim != RAYCA_NONE;
}

// ---- surface data (HitInfo, rayca-soft/src/hit.rs) ---------------------------------------------
__device__ __forceinline__ Color load_color(const float* c) { return Color{c[0], c[1], c[2], c[3]}; }
__device__ __forceinline__ F4 load_vec3(const float* c) { return vec3(c[0], c[1], c[2]); }

__device__ __forceinline__ uint32_t f32_as_u32_sat(float v) {  // Rust `as u32`
  if (!(v == v) || v <= 0.0f) return 0u;
  if (v >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)v;
}
// Sampler::sample  rayca-model/src/sampler.rs:11-30 (nearest, wrap)
__device__ Color sample_texture(const DevScene& sc, uint32_t tex, F2 uv) {
  const DevTexture t = sc.textures[tex];
  const float fx = (uv.x - floorf(uv.x) + 1.0f) * (float)t.width;
  const float fy = (uv.y - floorf(uv.y) + 1.0f) * (float)t.height;
  const uint32_t x = f32_as_u32_sat(fx) % t.width, y = f32_as_u32_sat(fy) % t.height;
  const size_t idx = (size_t)y * t.width + x;
  const uint8_t* base = sc.image_bytes + t.byte_offset;
  if (t.color_type == RAYCA_COLOR_RGBA32F) {
    const float* f = reinterpret_cast<const float*>(base) + idx * 4;
    return Color{f[0] / 255.0f, f[1] / 255.0f, f[2] / 255.0f, f[3] / 255.0f};
  }
  if (t.color_type == RAYCA_COLOR_RGBA8) {
    const uint8_t* p = base + idx * 4;
    return Color{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
  }
  const uint8_t* p = base + idx * 3;
  return Color{(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, 255.0f / 255.0f};
}

__device__ __forceinline__ bool tex_valid(const DevScene& sc, uint32_t t) { return t != RAYCA_NONE && t < sc.texture_count; }

__device__ __forceinline__ Color pbr_color(const DevScene& sc, const DevMaterial& m, F2 uv) {  // pbr.rs:94-102
  const Color c = load_color(m.color);
  if (tex_valid(sc, m.albedo_texture)) return c * sample_texture(sc, m.albedo_texture, uv);
  return c;
}
__device__ __forceinline__ void pbr_metallic_roughness(const DevScene& sc, const DevMaterial& m, F2 uv, float& metallic, float& roughness) {
  if (tex_valid(sc, m.metallic_roughness_texture)) {  // pbr.rs:125-137: (b, r)
    const Color c = sample_texture(sc, m.metallic_roughness_texture, uv);
    metallic = c.b;
    roughness = c.r;
  } else {
    metallic = m.metallic_factor;
    roughness = m.roughness_factor;
  }
}
__device__ __forceinline__ DevMaterial default_material() {  // Material::DEFAULT -> PbrMaterial::WHITE
  DevMaterial m{};
  m.color[0] = m.color[1] = m.color[2] = m.color[3] = 1.0f;
  m.ambient[3] = m.emission[3] = m.diffuse[3] = m.specular[3] = 1.0f;
  m.kind = RAYCA_MATERIAL_PBR;
  m.albedo_texture = m.normal_texture = m.metallic_roughness_texture = RAYCA_NONE;
  m.metallic_factor = 0.0f;
  m.roughness_factor = 1.0f;
  return m;
}

__device__ __forceinline__ F4 interp3(const float (*a)[3], float bu, float bv) {
  const float w2 = 1.0f - bu - bv;
  return (load_vec3(a[2]) * w2 + load_vec3(a[0]) * bu) + load_vec3(a[1]) * bv;
}

// HitInfo (rayca-soft/src/hit.rs) reduced to the values the rest of the path vertex needs; everything
// here is a pure function of the hit, so evaluating it once instead of lazily gives the same bits.
struct ShadeCtx {
  F4 point, normal, view, next_origin;  // hit point, shading normal, -ray.dir, point + normal*BIAS
  Color kd, ks;                         // get_diffuse(), get_specular()
  float roughness, shininess;
  uint32_t kind;                        // RAYCA_MATERIAL_*
};

// The path kernel keeps a vertex's ShadeCtx in LDS while its shadow rays are traversed (RAYCA_PARK_CTX): the struct is 27
// registers that would otherwise have to survive the traversal loop, which at the 128 VGPRs of four waves per SIMD the
// compiler could only do through scratch (54 spilled VGPRs, 219 MB of scratch writes per 1080p frame).  So do the pending
// sample's contribution and the running sum of direct light (quads 5 and 6).  Seven float4 per lane, quantity-major (lane i of quantity q at [q * kBlock + i]): a wave's ds_read_b128 / ds_write_b128 touch 64
// consecutive 16-B slots, the conflict-free shape.  next_origin and the w lanes are recomputed (same operations, same bits).
#ifndef RAYCA_PARK_CTX
#define RAYCA_PARK_CTX 1
#endif
constexpr uint32_t kCtxQuads = 7;  // ShadeCtx (5) + the pending NEE sample's contribution + the running direct sum
__device__ __forceinline__ void park_ctx(float4* slot, const ShadeCtx& c) {
  slot[0 * kBlock] = make_float4(c.point.x, c.point.y, c.point.z, c.roughness);
  slot[1 * kBlock] = make_float4(c.normal.x, c.normal.y, c.normal.z, c.shininess);
  slot[2 * kBlock] = make_float4(c.view.x, c.view.y, c.view.z, __uint_as_float(c.kind));
  slot[3 * kBlock] = make_float4(c.kd.r, c.kd.g, c.kd.b, c.kd.a);
  slot[4 * kBlock] = make_float4(c.ks.r, c.ks.g, c.ks.b, c.ks.a);
}
__device__ __forceinline__ ShadeCtx unpark_ctx(const float4* slot) {
  const float4 q0 = slot[0 * kBlock], q1 = slot[1 * kBlock], q2 = slot[2 * kBlock], q3 = slot[3 * kBlock], q4 = slot[4 * kBlock];
  ShadeCtx c;
  c.point = f4(q0.x, q0.y, q0.z, 1.0f);   // Hit.point is a Point3
  c.normal = vec3(q1.x, q1.y, q1.z);
  c.view = vec3(q2.x, q2.y, q2.z);
  c.next_origin = c.point + c.normal * kRayBias;  // hit.rs:164-171, as in shade_hit
  c.kd = Color{q3.x, q3.y, q3.z, q3.w};
  c.ks = Color{q4.x, q4.y, q4.z, q4.w};
  c.roughness = q0.w;
  c.shininess = q1.w;
  c.kind = __float_as_uint(q2.w);
  return c;
}

// get_color (primitive.rs:142-148) always; the rest only when `full` (Pathtracer).
// barycentrics: u -> vertex 0, v -> vertex 1, 1-u-v -> vertex 2 (bvh/triangle.rs:34-38)
template <bool SPH>
__device__ __forceinline__ void shade_hit(const DevScene& sc, const DRay& ray, const DHit& hit, bool full, Color& color, bool& emissive,
                                          ShadeCtx& cx) {
  const PrimExt& e = sc.ext[hit.prim];
  const float bu = hit.u, bv = hit.v;
  const float w2 = 1.0f - bu - bv;
  const bool is_sphere = SPH && e.kind == RAYCA_GEOMETRY_SPHERE;
  // BvhGeometry::get_color / get_uv: spheres are white with uv (0,0)  (primitive.rs:15-28)
  const Color geom_color = is_sphere ? white() : (load_color(e.color[2]) * w2 + load_color(e.color[0]) * bu) + load_color(e.color[1]) * bv;
  const F2 uv = is_sphere ? F2{0.0f, 0.0f}
                          : F2{(e.uv[2][0] * w2 + e.uv[0][0] * bu) + e.uv[1][0] * bv, (e.uv[2][1] * w2 + e.uv[0][1] * bu) + e.uv[1][1] * bv};
  const DevMaterial m = (e.material != RAYCA_NONE && e.material < sc.material_count) ? sc.materials[e.material] : default_material();
  Color mc;  // Material::get_color  material/mod.rs:107-113
  if (m.kind == RAYCA_MATERIAL_PBR) mc = pbr_color(sc, m, uv);
  else if (m.kind == RAYCA_MATERIAL_PHONG) mc = load_color(m.ambient) + load_color(m.emission);
  else mc = load_color(m.diffuse);
  color = geom_color * mc;
  emissive = m.emissive != 0u;
  if (!full) return;
  // normal: primitive.rs:172-182 -> material/mod.rs:125-139 -> pbr.rs:104-123
  F4 normal;
  F4 point = ray.o + ray.d * hit.t;  // Hit.point  triangle.rs:122
  F4 hit_dir = ray.d;                // Hit.ray.dir
  if (is_sphere) {  // primitive.rs:183-190 and sphere.rs:155-163
    const DevSphere& sp = sc.spheres[e.node];
    const DRay l = sphere_local_ray(sp, ray);
    // Sphere::intersects builds its Hit from the INVERSE-TRANSFORMED ray (sphere.rs:138,157-159): everything
    // HitInfo later derives from hit.ray -- view vector, reflection, the transmitted ray -- is in the sphere's
    // model space.  Reproduced as is.
    hit_dir = l.d;
    const F4 model_point = l.o + l.d * hit.t;
    const Trs wt{f4(sp.translation[0], sp.translation[1], sp.translation[2], 0.0f), f4(sp.rotation[0], sp.rotation[1], sp.rotation[2], sp.rotation[3]),
                 f4(sp.scale[0], sp.scale[1], sp.scale[2], 0.0f)};
    point = trs_apply_point(wt, model_point);
    const F4 hp = mat4_apply_point(reinterpret_cast<const float(*)[4]>(sp.inv_mat4), point);
    const F4 mn = normalized(as_vec(hp - f4(sp.center[0], sp.center[1], sp.center[2], sp.center[3])));
    const float* nm = sp.normal_mat;
    float out[3] = {0.0f, 0.0f, 0.0f};
    const float in[3] = {mn.x, mn.y, mn.z};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out[i] += nm[4 * i + j] * in[j];
    normal = normalized(vec3(out[0], out[1], out[2]));
  } else {
    normal = normalized(interp3(e.normal, bu, bv));
  }
  if (!is_sphere && m.kind == RAYCA_MATERIAL_PBR && tex_valid(sc, m.normal_texture)) {
    const F4 tangent = normalized(interp3(e.tangent, bu, bv));
    const F4 bitangent = normalized(interp3(e.bitangent, bu, bv));
    F4 sn = premultiplied(sample_texture(sc, m.normal_texture, uv));
    sn = sn * 2.0f - f4(1.0f, 1.0f, 1.0f, 1.0f);
    normal = normalized(mat3_apply(mat3_tbn(tangent, bitangent, normal), sn));
  }
  cx.normal = normal;
  cx.point = point;
  cx.view = -hit_dir;                // ray.rs:149-151
  cx.next_origin = cx.point + normal * kRayBias;  // hit.rs:164-171
  cx.kind = m.kind;
  cx.shininess = m.shininess;
  // get_diffuse  primitive.rs:150-155 ; get_specular / get_roughness  material/mod.rs:141-151,173-185
  if (m.kind == RAYCA_MATERIAL_PBR) {
    float me, ro;
    pbr_metallic_roughness(sc, m, uv, me, ro);
    const Color base = pbr_color(sc, m, uv);
    cx.kd = geom_color * base;
    cx.ks = me * base;
    cx.roughness = ro;
  } else {
    cx.kd = geom_color * load_color(m.diffuse);
    cx.ks = load_color(m.specular);
    cx.roughness = m.kind == RAYCA_MATERIAL_PHONG ? clampf(sqrtf(2.0f / (m.shininess + 2.0f)), 0.0f, 1.0f) : m.roughness_factor;
  }
}

// ---- BRDFs: brdf/ggx.rs:58-129, brdf/lambertian.rs:7-16 -----------------------------------------
// The reference spells these terms with libm compositions -- tan(acos(c))^2, powf(c, 4), powf(x, 5)
// (ggx.rs:58-129) -- ~1500 VALU instructions per BRDF evaluation on this chip, a fifth of the bench frame's
// arithmetic.  RAYCA_GGX_CLOSED_FORM (default) evaluates the same quantities in closed form,
//   tan^2(acos c) = (1-c)(1+c)/c^2,   c^4 (a^2 + tan^2)^2 = (a^2 c^2 + (1-c)(1+c))^2,   x^5 = (x^2)^2 x,
// each within 4e-7 of the exact value, where the f32 composition itself is only good to the rounding of acosf
// amplified by 1/c (several percent at grazing half-vectors).  Shaded pixels are compared with the CPU restatement -- which
// keeps the reference's spelling -- at the 1e-4 tolerance of the parity tests; hit records are unaffected.
// Special cases of the composition are kept: c == 0 gives D = 0 (cos^4 underflows first), c > 1 gives NaN (acosf).
#ifndef RAYCA_GGX_CLOSED_FORM
#define RAYCA_GGX_CLOSED_FORM 1
#endif
__device__ __forceinline__ float ggx_d(float a, F4 h, F4 n) {
  const float a2 = a * a;
  const float cos_theta = clampf(dot(h, n), 0.0f, 1.0f);
#if RAYCA_GGX_CLOSED_FORM
  const float q = a2 * (cos_theta * cos_theta) + (1.0f - cos_theta) * (1.0f + cos_theta);
  const float denominator = q * q;
  if (denominator == 0.0f || cos_theta == 0.0f) return 0.0f;
#else
  const float theta = acosf(cos_theta);
  const float denominator = powf(cos_theta, 4.0f) * powf(a2 + powf(tanf(theta), 2.0f), 2.0f);
  if (denominator == 0.0f) return 0.0f;
#endif
  return a2 * kFrac1Pi / denominator;
}
__device__ __forceinline__ float ggx_g1(float a, F4 omega, F4 n) {
  const float cos_theta = dot(omega, n);
  if (cos_theta <= 0.0f) return 0.0f;
#if RAYCA_GGX_CLOSED_FORM
  const float tan2 = cos_theta > 1.0f ? NAN : ((1.0f - cos_theta) * (1.0f + cos_theta)) / (cos_theta * cos_theta);
  return 2.0f / (1.0f + sqrtf(1.0f + a * a * tan2));
#else
  const float theta = acosf(cos_theta);
  return 2.0f / (1.0f + sqrtf(1.0f + a * a * powf(tanf(theta), 2.0f)));
#endif
}
__device__ __forceinline__ Color ggx_f(Color ks, F4 omega_i, F4 h) {
  const float oh = fabsf(dot(omega_i, h));
#if RAYCA_GGX_CLOSED_FORM
  const float x = 1.0f - oh, x2 = x * x;
  return ks + (white() - ks) * ((x2 * x2) * x);
#else
  return ks + (white() - ks) * powf(1.0f - oh, 5.0f);
#endif
}
// inlined: as a call it kept ShadeCtx in scratch and cost 2.6 % of the depth-1 frame
__device__ __forceinline__ Color surf_brdf(const ShadeCtx& s, F4 omega_i) {  // HitInfo::get_brdf  hit.rs:220-227
  if (s.kind == RAYCA_MATERIAL_PHONG) {  // lambertian::get_brdf
    const Color lambertian = s.kd * kFrac1Pi;
    const float sh = s.shininess;
    const F4 refl = normalized(reflect(-s.view, s.normal));  // hit.rs:93-101 (ray.dir == -view exactly)
    const Color specular = (((s.ks * (sh + 2.0f)) * powf(dot(refl, omega_i), sh)) * kFrac1Pi) / 2.0f;
    return lambertian + specular;
  }
  Color bsdf = black();  // ggx::get_bsdf
  const F4 omega_o = s.view;
  const F4 n = s.normal;
  const float oin = clampf(dot(omega_i, n), 0.0f, 1.0f);
  const float oon = clampf(dot(omega_o, n), 0.0f, 1.0f);
  if (!(oin == 0.0f || oon == 0.0f)) {
    const float a = s.roughness;
    const F4 h = normalized(omega_i + omega_o);
    const Color f = ggx_f(s.ks, omega_i, h);
    const float g = ggx_g1(a, omega_i, n) * ggx_g1(a, omega_o, n);
    const float d = ggx_d(a, h, n);
    const float denominator = 4.0f * oin * oon;
    bsdf = ((f * g) * d) / denominator;
  }
  return s.kd * kFrac1Pi + bsdf;
}

// ---- work distribution ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
// Next 64-wide batch for this wave, or RAYCA_NONE.  The batch list is cut into 8 contiguous
// partitions, one per XCD: waves drain their own XCD's partition first (neighbouring image tiles ->
// the same L2), then steal from the others.  Placement only affects speed, never results.
// Tickets: each of the 8 counters lives on its own 256-B line (kHeadStride dwords apart) -- atomics on
// one line serialise in one L2 channel at ~12 ns each, which for 32k batches would be as long as the
// whole kernel -- and one ticket covers `ticket` consecutive batches: two when the frame has batches to spare, one
// when there are fewer batches than resident waves (a rank's share of a frame on eight GPUs): pairs would then
// leave half the waves without work and double the time of the others.
constexpr uint32_t kHeadStride = 64;
struct WorkCursor {
  uint32_t exhausted = 0;  // partitions found empty so far
  uint32_t next = 0, end = 0;  // batches of the current ticket still to do
};
__device__ __forceinline__ uint32_t next_batch(uint32_t* heads, uint32_t total, uint32_t home, WorkCursor& wc, uint32_t ticket) {
  if (wc.next < wc.end) return wc.next++;
  ticket = max(ticket, 1u);  // a zero ticket would never drain the counters
  while (wc.exhausted < 8u) {
    const uint32_t part = (home + wc.exhausted) & 7u;
    const uint32_t lo = (uint32_t)(((uint64_t)part * total) >> 3), hi = (uint32_t)(((uint64_t)(part + 1u) * total) >> 3);
    uint32_t idx = 0;
    if (__lane_id() == 0) idx = atomicAdd(&heads[part * kHeadStride], ticket);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx < hi - lo) {
      wc.next = lo + idx + 1u;
      wc.end = min(lo + idx + ticket, hi);
      return lo + idx;
    }
    wc.exhausted++;
  }
  return RAYCA_NONE;
}

struct PathBuffers {
  float4* direct;   // [depth][npix] Color: direct lighting, or the terminal colour
  float4* brdf;     // [depth][npix] Color: factor get_radiance applies to the child's radiance
  uint32_t* state;  // [depth][npix] kVertex*
  uint32_t npix;
};

__device__ __forceinline__ float4 as_f4(Color c) { return make_float4(c.r, c.g, c.b, c.a); }
__device__ __forceinline__ Color as_color(float4 c) { return Color{c.x, c.y, c.z, c.w}; }

// scene.rs:146-148: color /= spp; correct_gamma; RGBA8::from
__device__ __forceinline__ void finalize_pixel(const FrameParams& fp, Color acc, uint32_t p, uint8_t* rgba8, float4* rgba32f) {
  Color c = acc / (float)fp.spp;
  if (fp.inv_gamma != 1.0f) {
    c.r = powf(c.r, fp.inv_gamma);
    c.g = powf(c.g, fp.inv_gamma);
    c.b = powf(c.b, fp.inv_gamma);
  }
  if (rgba32f) rgba32f[p] = as_f4(c);
  if (rgba8) {
    const uint32_t packed = (uint32_t)quantize(c.r) | ((uint32_t)quantize(c.g) << 8) | ((uint32_t)quantize(c.b) << 16) | ((uint32_t)quantize(c.a) << 24);
    reinterpret_cast<uint32_t*>(rgba8)[p] = packed;
  }
}

// Camera ray of output pixel (x, packed row r): scene.rs:125-141 + trs.rs:275-284 + ray.rs:74-91
__device__ __forceinline__ DRay camera_ray(const FrameParams& fp, uint32_t x, uint32_t y) {
  const float xx = (2.0f * ((((float)x + fp.sub_step_x) + fp.sub_offset) * fp.inv_width) - 1.0f) * fp.angle * fp.aspect;
  const float yy = (1.0f - 2.0f * ((((float)y + fp.sub_step_y) + fp.sub_offset) * fp.inv_height)) * fp.angle;
  F4 dir = normalized(vec3(xx, yy, -1.0f));
  F4 origin = point3(0.0f, 0.0f, 0.0f);
  // Ray::scale
  dir = dir * fp.camera.scale;
  origin = point_scale(origin, fp.camera.scale);
  // Ray::rotate
  dir = rotate(dir, fp.camera.rotation);
  origin = point_rotate(origin, fp.camera.rotation);
  origin.w = 1.0f;
  // Ray::translate
  origin = origin + fp.camera.translation;
  DRay r;
  r.o = origin;
  r.d = dir;
  r.rd = reciprocal(dir);
  return r;
}

// wave-aggregated append of the next generation's rays
__device__ __forceinline__ void push_ray(bool has, const QueuedRay& qr, QueuedRay* out, uint32_t* out_count) {
  const unsigned long long mask = __ballot(has);
  if (mask == 0ull) return;
  const uint32_t lane = __lane_id();
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(out_count, (uint32_t)__popcll(mask));
  base = __shfl(base, (int)leader);
  if (has) out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = qr;
}

constexpr int kModeFlat = 0, kModePath = 1, kModeGeneral = 2;

// One NEE sample (sampler/nee.rs:72-166), split around its shadow ray: everything that does not depend
// on the shadow ray's outcome is evaluated first (pure functions, same values), so that only the
// candidate contribution `x` has to survive the traversal.
struct NeeSample {
  Color x;       // contribution if the light turns out to be visible
  float t_stop;  // point light: its distance (any-hit bound); quad light: FLT_MAX (closest hit wanted)
  uint32_t quad; // 1: "lit iff the closest hit is emissive" (nee.rs:103-104)
};
__device__ __forceinline__ NeeSample nee_prepare(const DevScene& sc, const FrameParams& fp, const ShadeCtx& s, uint32_t li, uint32_t k,
                                                 uint32_t key, uint32_t& dim, DRay& shadow_ray) {
  const DevLight L = sc.lights[li];
  NeeSample ns;
  const F4 x_point = s.point;
  if (L.kind == RAYCA_LIGHT_POINT) {  // get_point_light_sample  nee.rs:127-166
    const F4 x1 = f4(L.position[0], L.position[1], L.position[2], L.position[3]);
    const F4 x_to_x1 = as_vec(x1 - x_point);
    const float dist = length(x_to_x1);
    const F4 omega = normalized(x_to_x1);
    shadow_ray = make_ray(s.next_origin, omega);
    // PointLight::get_intensity / get_fallof  light/point.rs:37-49
    const F4 dvec = to_vec(x_point) - to_vec(x1);
    const float r2 = norm2(dvec);
    const float rr = sqrtf(r2);
    const float fallof = hsum(f4(L.attenuation[0], L.attenuation[1], L.attenuation[2], 0.0f) * f4(1.0f, rr, r2, 0.0f));
    const Color le = (L.intensity * load_color(L.color)) / fallof;
    const Color brdf = surf_brdf(s, omega);
    const float r_squared = norm2(x_to_x1);
    const float d_omega = 1.0f / r_squared;
    const float n_dot_omega = clampf(dot(s.normal, omega), 0.0f, 1.0f);
    ns.x = ((le * brdf) * n_dot_omega) * d_omega;
    ns.t_stop = dist;
    ns.quad = 0u;
  } else {  // get_quad_light_sample  nee.rs:72-125 ; QuadLight::get_random_point  light/quad.rs:112-135
    const F4 ab = f4(L.ab[0], L.ab[1], L.ab[2], 0.0f), ac = f4(L.ac[0], L.ac[1], L.ac[2], 0.0f);
    const float sc_f = (float)fp.strate_count;
    const float u1 = rng_f32(key, dim++) / sc_f;
    const float u2 = rng_f32(key, dim++) / sc_f;
    const F4 a = f4(L.position[0], L.position[1], L.position[2], L.position[3]);
    F4 x1 = (a + u1 * ab) + u2 * ac;
    if (fp.light_stratify) {
      const float i1 = (float)(k % fp.strate_count), i2 = (float)(k / fp.strate_count);
      x1 = x1 + ((ab / sc_f) * i1 + (ac / sc_f) * i2);
    }
    const F4 x_to_x1 = as_vec(x1 - x_point);
    const F4 omega = normalized(x_to_x1);
    shadow_ray = make_ray(s.next_origin, omega);
    const Color le = L.intensity * load_color(L.color);
    const Color brdf = surf_brdf(s, omega);
    const float r_squared = norm2(x_to_x1);
    const float d_omega = dot(f4(L.normal[0], L.normal[1], L.normal[2], 0.0f), omega) / r_squared;
    const float n_dot_omega = clampf(dot(s.normal, omega), 0.0f, 1.0f);
    ns.x = (((le * L.area) * brdf) * n_dot_omega) * d_omega;
    ns.t_stop = FLT_MAX;
    ns.quad = 1u;
  }
  return ns;
}

// One generation of rays.  Per lane a small state machine around ONE traversal call site:
//   primary ray -> shade -> [NEE shadow ray]* -> bounce sample -> done
// so the traversal loop is instantiated once per kernel and the registers that must survive it are
// the ray, the hit, the compact ShadeCtx and a few colours.
template <int MODE, bool GEN0, bool ORDERED, bool FUSED, bool FAST, bool SPH, bool WIDE, bool SPILL, bool STATS, bool HALF>
__global__ __launch_bounds__(kBlock, MODE == kModeFlat ? RAYCA_MIN_WAVES_FLAT : RAYCA_MIN_WAVES) void k_generation(DevScene sc, FrameParams fp, uint32_t* heads, const QueuedRay* in_rays,
                                                       const uint32_t* in_count, QueuedRay* out_rays, uint32_t* out_count,
                                                       PathBuffers pb, uint32_t depth, uint8_t* rgba8, float4* rgba32f,
                                                       TraceCounters* counters, TraceLaunch tl) {
  extern __shared__ uint32_t lds_stack[];
  NodeStack<SPILL> stack = make_stack<SPILL>(lds_stack, tl, blockIdx.x * kBlock + threadIdx.x);
  // behind the stack rows: this lane's parked ShadeCtx (path frames only; the host sizes the allocation)
  float4* const ctx_slot = reinterpret_cast<float4*>(lds_stack + tl.lds_entries * kBlock) + threadIdx.x;
  constexpr bool PARK = RAYCA_PARK_CTX && MODE == kModePath;
  const uint32_t lane = __lane_id();
  const uint32_t home = xcc_id();
  WorkCursor wc;
  const uint32_t total = GEN0 ? fp.tile_count : (*in_count + 63u) / 64u;
  LaneCounters cnt;
  uint32_t n_shaded = 0, n_shadow = 0, n_bounce = 0;
  const bool collect_emissive = GEN0 ? true : (fp.direct_sampler == RAYCA_SAMPLER_NONE);
  const uint32_t nee_lights = (MODE == kModePath && fp.direct_sampler == RAYCA_SAMPLER_NEE) ? sc.light_count : 0u;

  for (;;) {
    const uint32_t batch = next_batch(heads, total, home, wc, tl.ticket);
    if (batch == RAYCA_NONE) break;
    bool live;
    uint32_t p = 0, key = 0;
    DRay ray;
    if (GEN0) {
      const uint32_t ty = batch / fp.tiles_x, tx = batch - ty * fp.tiles_x;
      const uint32_t x = tx * kTileW + (lane & (kTileW - 1u)), r = ty * kTileH + (lane >> RAYCA_TILE_W_LOG2);
      live = x < fp.width && r < fp.rows;
      const uint32_t y = ((r / fp.band) * fp.parts + fp.part) * fp.band + (r % fp.band);
      p = r * fp.width + x;
      if (live) {
        ray = camera_ray(fp, x, y);
        key = rng_root(fp.seed, y * fp.width + x, fp.sample);
      }
    } else {
      const uint32_t i = batch * 64u + lane;
      live = i < *in_count;
      if (live) {
        const QueuedRay q = in_rays[i];
        ray = make_ray(point3(q.ox, q.oy, q.oz), vec3(q.dx, q.dy, q.dz));
        p = q.pixel;
        key = q.key;
      }
    }
    const size_t slot = (size_t)depth * pb.npix + p;
    bool in_shadow = false;
    // shadow rays: the any-hit bound.  A point light's is its distance (a finite square root, never FLT_MAX); FLT_MAX
    // means "closest hit wanted": camera and bounce rays, and the shadow ray of a quad light (NeeSample.quad)
    float t_stop = FLT_MAX;
    ShadeCtx cx;
    Color ns_x = black();  // the pending NEE sample's contribution if its light turns out to be visible   (PARK: quad 5)
    Color direct = black();  // sum of the direct samples so far                                           (PARK: quad 6)
    uint32_t li = 0, k = 0, dim = 0;
    // FUSED (Flat only): a lane that finishes leaves its pixel sum in `direct`; the gamma + quantise + store tail runs
    // once per batch with the wave reconverged, not inside the divergent state machine
    const bool has_pixel = live;
    // Path: a vertex whose direct samples are all in leaves the loop with `tail` set; its record, the bounce sample and the
    // next generation's ray are made behind the loop, with the wave reconverged -- so the bounce ray is not a loop-carried
    // value (eight registers the traversal loop had to carry for nothing) and the sampling code runs with full lanes
    bool tail = false;

    while (live) {
      DHit hit;
      const bool found = trace<ORDERED, FAST, SPH, WIDE, SPILL, STATS, HALF>(sc, ray, t_stop, stack, hit, cnt);
      if (!in_shadow) {
        if (!found) {
          if (FUSED) direct = black() + black();  // unwrap_or(BLACK), color += it
          else pb.state[slot] = kVertexNone;
          live = false;
        } else {
          n_shaded++;
          Color color;
          bool emissive;
          shade_hit<SPH>(sc, ray, hit, MODE == kModePath, color, emissive, cx);
          if (MODE == kModeFlat) {  // Flat::trace  integrator/flat.rs:16-28
            if (FUSED) direct = black() + color;
            else {
              pb.direct[slot] = as_f4(color);
              pb.state[slot] = kVertexEmissive;
            }
            live = false;
          } else if (collect_emissive && emissive) {  // pathtracer.rs:83-87
            pb.direct[slot] = as_f4(color);
            pb.state[slot] = kVertexEmissive;
            live = false;
          } else {
            in_shadow = true;  // enter the NEE loop (possibly empty)
            if (PARK) {
              park_ctx(ctx_slot, cx);
              ctx_slot[6 * kBlock] = as_f4(black());
            }
          }
        }
      } else {
        // outcome of the pending NEE sample
        bool lit;
        if (t_stop == FLT_MAX) {  // quad light: "lit iff the closest hit is emissive" (nee.rs:103-104)
          lit = false;
          if (found) {
            const uint32_t hm = sc.ext[hit.prim].material;
            lit = hm != RAYCA_NONE && hm < sc.material_count && sc.materials[hm].emissive != 0u;
          }
        } else {
          lit = !(found && hit.t < t_stop);  // nee.rs:152-156
        }
        if (PARK) {
          const Color sum = as_color(ctx_slot[6 * kBlock]) + (lit ? as_color(ctx_slot[5 * kBlock]) : black());
          ctx_slot[6 * kBlock] = as_f4(sum);
        } else {
          direct = direct + (lit ? ns_x : black());
        }
        if (++k == fp.light_samples) {
          k = 0;
          ++li;
        }
      }
      if (live && in_shadow) {
        if (li < nee_lights) {
          if (PARK) cx = unpark_ctx(ctx_slot);  // (its own LDS slot: no barrier; the compiler orders a lane's LDS accesses)
          const NeeSample ns = nee_prepare(sc, fp, cx, li, k, key, dim, ray);
          if (PARK) ctx_slot[5 * kBlock] = as_f4(ns.x);
          else ns_x = ns.x;
          t_stop = ns.t_stop;
        } else {
          tail = true;
          live = false;
        }
      }
    }
    if (FUSED) {
      if (has_pixel) finalize_pixel(fp, direct, p, rgba8, rgba32f);
    } else if (MODE == kModePath) {
      bool want_bounce = false;
      QueuedRay next{};
      if (tail) {
        // all direct samples done: Pathtracer::trace_impl tail  pathtracer.rs:89-105
        if (PARK) {
          cx = unpark_ctx(ctx_slot);
          direct = as_color(ctx_slot[6 * kBlock]);
        }
        const uint32_t limit = fp.direct_sampler != RAYCA_SAMPLER_NONE ? fp.max_depth - 1u : fp.max_depth;
        pb.direct[slot] = as_f4(direct);
        if (depth < limit) {
          // CosineSampler::get_random_dir  sampler/cosine.rs:65-88 ; HemisphereSampler  hemisphere.rs:17-40
          const float e1 = rng_f32(key, dim++), e2 = rng_f32(key, dim++);
          const bool hemi = fp.indirect_sampler == RAYCA_SAMPLER_HEMISPHERE;
          const float theta = hemi ? acosf(e1) : acosf(sqrtf(e1));
          const float omega_a = 2.0f * kPi * e2;
          const F4 sdir = vec3(cosf(omega_a) * sinf(theta), sinf(omega_a) * sinf(theta), cosf(theta));
          const F4 w = cx.normal;
          const F4 a = close(w, vec3(0, 1, 0)) ? vec3(1, 0, 0) : vec3(0, 1, 0);
          const F4 u = normalized(cross(a, w));
          F4 v = cross(w, u);
          if (hemi) v = normalized(v);
          const F4 omega_i = (sdir.x * u + sdir.y * v) + sdir.z * w;
          const Color brdf = surf_brdf(cx, omega_i);
          // the factor SoftSampler::get_radiance applies to the incoming radiance, evaluated in its
          // order up to the point where the child's result enters: cosine.rs:90-99 `PI * brdf`,
          // hemisphere.rs:42-52 `2.0 * PI * brdf * cosine_law`
          Color factor;
          if (hemi) factor = ((2.0f * kPi) * brdf) * clampf(dot(cx.normal, omega_i), 0.0f, 1.0f);
          else factor = kPi * brdf;
          pb.brdf[slot] = as_f4(factor);
          pb.state[slot] = kVertexLit;
          if (depth + 1u < fp.max_depth) {
            want_bounce = true;
            next.ox = cx.next_origin.x; next.oy = cx.next_origin.y; next.oz = cx.next_origin.z;
            next.dx = omega_i.x; next.dy = omega_i.y; next.dz = omega_i.z;
            next.pixel = p;
            next.key = rng_child(key, 0u);
          }
        } else {
          pb.state[slot] = kVertexLitNoIndirect;
        }
      }
      // ray counts, wave-uniform (per-lane counters would be loop-carried registers of the traversal loop): every vertex
      // that reaches `tail` has traced all of its nee_lights x light_samples shadow rays
      n_shadow += (uint32_t)__popcll(__ballot(tail)) * nee_lights * fp.light_samples;
      n_bounce += (uint32_t)__popcll(__ballot(want_bounce));
      push_ray(want_bounce, next, out_rays, out_count);
    }
  }
  if (STATS) {
    // wave reduction, then one atomic per wave and counter
    unsigned long long b = cnt.boxes, t = cnt.tris, sh = n_shaded, sb = cnt.slot_boxes, stt = cnt.slot_tris;
    for (int off = 32; off > 0; off >>= 1) {
      b += __shfl_down(b, off);
      t += __shfl_down(t, off);
      sh += __shfl_down(sh, off);
      sb += __shfl_down(sb, off);
      stt += __shfl_down(stt, off);
    }
    if (lane == 0) {
      atomicAdd(&counters->boxes, b);
      atomicAdd(&counters->tris, t);
      atomicAdd(&counters->shaded, sh);
      atomicAdd(&counters->box_slots, sb);
      atomicAdd(&counters->tri_slots, stt);
    }
  }
  if (MODE == kModePath) {
    const unsigned long long s2 = n_shadow, b2 = n_bounce;  // already whole-wave sums
    if (lane == 0 && (s2 | b2)) {
      atomicAdd(&counters->shadow, s2);
      atomicAdd(&counters->bounce, b2);
    }
  }
}

// Fold the per-depth records in the reference's order (pathtracer.rs:23-66,94-105):
//   L_g = direct_g + (BLACK + (BLACK + (factor_g * L_{g+1}) * weight) / light_samples)
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, PathBuffers pb, uint32_t depths, float4* accum, uint8_t* rgba8, float4* rgba32f) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pb.npix) return;
  bool some = false;
  Color L = black();
  for (int g = (int)depths - 1; g >= 0; --g) {
    const size_t slot = (size_t)g * pb.npix + p;
    const uint32_t st = pb.state[slot];
    if (st == kVertexNone) {
      some = false;
    } else if (st == kVertexEmissive) {
      some = true;
      L = as_color(pb.direct[slot]);
    } else {
      const Color direct = as_color(pb.direct[slot]);
      Color indirect = black();
      if (st == kVertexLit) {
        Color li = black();
        if (some) {
          const Color factor = as_color(pb.brdf[slot]);
          const Color x = (factor * L) * 1.0f;  // ... * indirect_sample * weight (weight = 1 without roulette)
          li = li + x;
        }
        indirect = indirect + li / (float)fp.light_samples;
      }
      L = direct + indirect;
      some = true;
    }
  }
  const Color c = some ? L : black();
  const Color prev = fp.sample == 0 ? black() : as_color(accum[p]);
  const Color acc = prev + c;
  if (fp.sample + 1u == fp.spp) finalize_pixel(fp, acc, p, rgba8, rgba32f);
  else accum[p] = as_f4(acc);
}

template <bool ORDERED, bool FAST, bool SPH, bool WIDE, bool SPILL, bool STATS>
__global__ __launch_bounds__(kBlock, RAYCA_TRACE_MIN_WAVES) void k_trace_rays(DevScene sc, const float* rays, uint32_t count, float* t_out, uint32_t* prim_out, float* uv_out,
                                                       TraceCounters* counters, TraceLaunch tl) {
  extern __shared__ uint32_t lds_stack[];
  NodeStack<SPILL> stack = make_stack<SPILL>(lds_stack, tl, blockIdx.x * kBlock + threadIdx.x);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  LaneCounters cnt;
  if (i < count) {
    const float* r = rays + 6ull * i;
    const DRay ray = make_ray(point3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]));
    DHit hit;
    const bool found = trace<ORDERED, FAST, SPH, WIDE, SPILL, STATS>(sc, ray, FLT_MAX, stack, hit, cnt);
    t_out[i] = found ? hit.t : FLT_MAX;
    prim_out[i] = found ? hit.prim : RAYCA_NONE;
    uv_out[2 * i] = found ? hit.u : 0.0f;
    uv_out[2 * i + 1] = found ? hit.v : 0.0f;
  }
  if (STATS) {
    unsigned long long b = cnt.boxes, t = cnt.tris;
    for (int off = 32; off > 0; off >>= 1) {
      b += __shfl_down(b, off);
      t += __shfl_down(t, off);
    }
    if (__lane_id() == 0) {
      atomicAdd(&counters->boxes, b);
      atomicAdd(&counters->tris, t);
    }
  }
}

#include "general.inc"
#include "wavefront.inc"

}  // namespace
}  // namespace rayca

#include "api.inc"
