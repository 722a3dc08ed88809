// host_scene.cpp -- scene-graph flattening and the reference-equivalent BVH build.
//
// What the reference does at the top of every draw (rayca-soft/src/scene.rs:90-99):
//   SceneDrawInfo::new      scene.rs:190-282        world transforms + draw-info lists
//   BvhScene::from_scene    bvh/primitive.rs:194-395 one BvhModel (primitive list) per model
//   Tlas::new               bvh/tlas.rs:248-269     one BLAS per model (SAH, bvh/blas.rs:229-316)
//                                                   + median-split TLAS (bvh/tlas.rs:74-134)
// The tree produced here has exactly the reference's topology and primitive order (so depth ties
// resolve identically); only the evaluation strategy differs: the 63 x 3 SAH candidates of
// find_best_split_plane (blas.rs:93-123) are priced from one binned sweep per axis instead of 189
// passes over the primitives, and disjoint subtrees are built on separate host threads.
#include "host_scene.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <chrono>
#include <cmath>
#include <thread>

#include <sched.h>
#include <sys/mman.h>
#include <unordered_map>

namespace rayca {

// ---- the pool behind DefaultInitAllocator (host_scene.hpp) ------------------------------------------------------------------
namespace {
struct BigBlockPool {
  std::mutex mu;
  std::unordered_map<void*, size_t> capacity;            // every live block of the pool, handed out or cached
  std::vector<std::pair<size_t, void*>> cached;          // (capacity, block) not handed out
  size_t cached_bytes = 0;
  size_t limit = [] {
    const char* e = getenv("RAYCA_HOST_POOL_MB");
    return (size_t)(e ? std::max(atoi(e), 0) : 1024) << 20;
  }();
  ~BigBlockPool() {
    for (auto& c : cached) std::free(c.second);
  }
};
BigBlockPool& big_pool() {
  static BigBlockPool* p = new BigBlockPool();   // (never destroyed: vectors of static lifetime may give blocks back at exit)
  return *p;
}
}  // namespace
void* big_block_take(size_t bytes) {
  BigBlockPool& bp = big_pool();
  constexpr size_t kAlign = size_t(2) << 20;
  const size_t want = (bytes + kAlign - 1) / kAlign * kAlign;
  {
    std::lock_guard<std::mutex> lock(bp.mu);
    size_t best = bp.cached.size();
    for (size_t i = 0; i < bp.cached.size(); ++i)   // the smallest cached block that fits and is not wastefully large
      if (bp.cached[i].first >= want && bp.cached[i].first <= 2 * want + (size_t(8) << 20) && (best == bp.cached.size() || bp.cached[i].first < bp.cached[best].first)) best = i;
    if (best != bp.cached.size()) {
      void* p = bp.cached[best].second;
      bp.cached_bytes -= bp.cached[best].first;
      bp.cached.erase(bp.cached.begin() + (long)best);
      return p;
    }
  }
  void* p = std::aligned_alloc(kAlign, want);
  if (!p) throw std::bad_alloc();
  (void)madvise(p, want, MADV_HUGEPAGE);   // (where the system allows it: 512 times fewer faults on first touch)
  std::lock_guard<std::mutex> lock(bp.mu);
  bp.capacity[p] = want;
  return p;
}
void big_block_give(void* p) noexcept {
  if (!p) return;
  BigBlockPool& bp = big_pool();
  bool release = false;
  {
    std::lock_guard<std::mutex> lock(bp.mu);
    auto it = bp.capacity.find(p);
    if (it == bp.capacity.end()) return;   // (not ours: cannot happen)
    if (bp.cached_bytes + it->second <= bp.limit && bp.cached.size() < 64) {
      bp.cached.emplace_back(it->second, p);
      bp.cached_bytes += it->second;
    } else {
      bp.capacity.erase(it);
      release = true;
    }
  }
  if (release) std::free(p);
}

unsigned host_threads() {
  static const unsigned n = [] {
    if (const char* e = getenv("RAYCA_HOST_THREADS")) return std::max(1u, (unsigned)atoi(e));
    unsigned avail = std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) avail = (unsigned)CPU_COUNT(&set);
    return std::max(1u, std::min(avail, 16u));
  }();
  return n;
}

namespace {

// fn(begin, end) over [0, n) on all host threads
template <class Fn>
void parallel_chunks(size_t n, Fn fn) {
  const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(host_threads(), n / 4096 + 1));
  if (nt <= 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t per = (n + nt - 1) / nt;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back([=, &fn] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
  fn(0, std::min(n, per));
  for (std::thread& th : pool) th.join();
}

inline Box empty_box() { return Box{point3(FLT_MAX, FLT_MAX, FLT_MAX), point3(-FLT_MAX, -FLT_MAX, -FLT_MAX)}; }
inline Box origin_box() { return Box{point3(0, 0, 0), point3(0, 0, 0)}; }  // AABB::default()  aabb.rs:9-13
inline void grow(Box& bx, F4 p) {
  bx.a = vmin(bx.a, p);
  bx.b = vmax(bx.b, p);
}
inline float area(const Box& bx) {  // aabb.rs:20-23
  const F4 e = as_vec(bx.b - bx.a);
  return e.x * e.y + e.y * e.z + e.z * e.x;
}
inline float axis_of(F4 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

inline Trs trs_from_abi(const RaycaTrs& t) {
  return Trs{vec3(t.translation[0], t.translation[1], t.translation[2]), f4(t.rotation[0], t.rotation[1], t.rotation[2], t.rotation[3]),
             vec3(t.scale[0], t.scale[1], t.scale[2])};
}

float sphere_world_radius(const HostPrim& p, const Trs& t) {  // sphere.rs:90-92
  return p.radius * fmaxf(fmaxf(t.scale.x, t.scale.y), fmaxf(t.scale.z, t.scale.w));
}

// grow a box by a primitive the way AABB::grow_primitive does (aabb.rs:62-72)
void grow_primitive(Box& bx, const HostPrim& p, const Trs& t) {
  if (p.kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
    grow(bx, p.wp[0]);
    grow(bx, p.wp[1]);
    grow(bx, p.wp[2]);
  } else {
    const float r = sphere_world_radius(p, t);
    const F4 c = trs_apply_point(t, p.center);
    grow(bx, c + vec3(-r, 0, 0));
    grow(bx, c + vec3(r, 0, 0));
    grow(bx, c + vec3(0, -r, 0));
    grow(bx, c + vec3(0, r, 0));
    grow(bx, c + vec3(0, 0, -r));
    grow(bx, c + vec3(0, 0, r));
  }
}

void cache_world(HostPrim& p, const Trs& t) {
  if (p.kind == RAYCA_GEOMETRY_TRIANGLE_MESH) {
    for (int i = 0; i < 3; ++i) p.wp[i] = trs_apply_point(t, p.p[i]);  // Triangle::get_vertex  triangle.rs:67-69
    p.wcentroid = to_point(trs_apply_vec(t, p.centroid));              // primitive.rs:71-77 -> triangle.rs:161-163
    F4 mn = point3(FLT_MAX, FLT_MAX, FLT_MAX), mx = point3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < 3; ++i) {
      mn = vmin(mn, p.wp[i]);
      mx = vmax(mx, p.wp[i]);
    }
    p.wmin = mn;
    p.wmax = mx;
  } else {
    const float r = sphere_world_radius(p, t);
    const F4 c = trs_apply_point(t, p.center);
    p.wcentroid = c;
    p.wmin = c - vec3(r, r, r);  // sphere.rs:170-178
    p.wmax = c + vec3(r, r, r);
    p.wp[0] = p.wp[1] = p.wp[2] = point3(0, 0, 0);
  }
}

// ---- BLAS build --------------------------------------------------------------------------------
thread_local BlasBuildFn g_device_builder = nullptr;  // per thread: scenes may be created concurrently on different devices
thread_local uint32_t g_device_ordinal = 0;
// below this many primitives the host builder is faster than a level-by-level sequence of launches
constexpr uint32_t kDeviceBuildMin = 4096;

struct BlasBuilder {
  const HostScene* scene;
  std::vector<uint32_t>* order;  // indices into scene->prims, partitioned in place
  uint32_t max_depth;
  unsigned parallel_levels;
  bool seed_origin;  // true: the reference's AABB::default() seed (blas.rs:66-67); false: empty boxes

  const HostPrim& prim(uint32_t slot) const { return scene->prims[(*order)[slot]]; }

  Box range_bounds(uint32_t offset, uint32_t count) const {  // BvhNode::new  blas.rs:27-36
    Box bx = empty_box();
    for (uint32_t i = offset; i < offset + count; ++i) {
      bx.a = vmin(bx.a, prim(i).wmin);
      bx.b = vmax(bx.b, prim(i).wmax);
    }
    return bx;
  }

  // find_best_split_plane (blas.rs:93-123) + evaluate_sah (blas.rs:64-89), exact binned form.
  // A primitive is left of plane i iff centroid < pos_i; pos_i is non-decreasing in i, so it is left
  // of exactly the planes i > k with k = #{i : pos_i <= centroid}.  Unions of min/max are exact, so
  // each of the 63 (count, box) pairs -- and therefore each cost -- equals the literal evaluation.
  void best_split(const BuildNode& node, int& best_axis, float& split_pos, float& best_cost) const {
    best_cost = FLT_MAX;
    best_axis = 0;
    split_pos = 0.0f;
    for (int axis = 0; axis < 3; ++axis) {
      const float bmin = axis_of(node.bounds.a, axis), bmax = axis_of(node.bounds.b, axis);
      if (bmin == bmax) continue;
      const float scale = (bmax - bmin) / 64.0f;
      float pos[64];
      pos[0] = -FLT_MAX;
      for (int i = 1; i < 64; ++i) pos[i] = bmin + (float)i * scale;
      Box bins[64];
      uint32_t cnt[64];
      for (int b = 0; b < 64; ++b) {
        bins[b] = empty_box();
        cnt[b] = 0;
      }
      for (uint32_t s = node.offset; s < node.offset + node.count; ++s) {
        const HostPrim& p = prim(s);
        const float c = axis_of(p.wcentroid, axis);
        int k;
        if (c != c) {
          k = 63;  // NaN: never `<` any plane
        } else {
          // k = number of planes j in 1..63 with pos[j] <= c  (upper bound in a monotone array)
          int lo = 1, hi = 64;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pos[mid] <= c) lo = mid + 1;
            else hi = mid;
          }
          k = lo - 1;
        }
        cnt[k]++;
        grow_primitive(bins[k], p, scene->world_trs[p.node]);
      }
      Box lbox[65], rbox[65];
      uint32_t lcnt[65], rcnt[65];
      lbox[0] = seed_origin ? origin_box() : empty_box();
      lcnt[0] = 0;
      for (int b = 0; b < 64; ++b) {
        lbox[b + 1] = lbox[b];
        lcnt[b + 1] = lcnt[b] + cnt[b];
        if (cnt[b]) {
          lbox[b + 1].a = vmin(lbox[b + 1].a, bins[b].a);
          lbox[b + 1].b = vmax(lbox[b + 1].b, bins[b].b);
        }
      }
      rbox[64] = seed_origin ? origin_box() : empty_box();
      rcnt[64] = 0;
      for (int b = 63; b >= 0; --b) {
        rbox[b] = rbox[b + 1];
        rcnt[b] = rcnt[b + 1] + cnt[b];
        if (cnt[b]) {
          rbox[b].a = vmin(rbox[b].a, bins[b].a);
          rbox[b].b = vmax(rbox[b].b, bins[b].b);
        }
      }
      for (int i = 1; i < 64; ++i) {
        float cost;
        if (seed_origin) {
          cost = (float)lcnt[i] * area(lbox[i]) + (float)rcnt[i] * area(rbox[i]);
        } else {
          if (lcnt[i] == 0 || rcnt[i] == 0) continue;  // not a split
          cost = (float)lcnt[i] * area(lbox[i]) + (float)rcnt[i] * area(rbox[i]);
        }
        if (!(cost > 0.0f)) cost = FLT_MAX;
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = axis;
          split_pos = pos[i];
        }
      }
    }
  }

  // Blas::set_primitives_recursive  blas.rs:261-316.  Builds the subtree below arena[idx]; new
  // nodes go to `arena` (thread-local when running as a task).
  void split(std::vector<BuildNode>& arena, int32_t idx, uint32_t level) const {
    if (level >= max_depth) return;
    BuildNode node = arena[idx];
    int axis;
    float pos, cost;
    best_split(node, axis, pos, cost);
    const float no_split_cost = (float)node.count * area(node.bounds);
    if (cost > no_split_cost) return;
    uint32_t i = node.offset, j = node.offset + node.count;
    std::vector<uint32_t>& ord = *order;
    while (i < j) {  // partition-in-place, exactly the reference's swap sequence (blas.rs:279-289)
      if (axis_of(scene->prims[ord[i]].wcentroid, axis) < pos) {
        ++i;
      } else {
        std::swap(ord[i], ord[j - 1]);
        --j;
      }
    }
    const uint32_t left_count = i - node.offset, right_count = node.count - left_count;
    if (left_count == 0 || right_count == 0) return;
    BuildNode l, r;
    l.offset = node.offset;
    l.count = left_count;
    l.bounds = range_bounds(l.offset, l.count);
    r.offset = node.offset + left_count;
    r.count = right_count;
    r.bounds = range_bounds(r.offset, r.count);
    if (level < parallel_levels && node.count > 8192) {
      // children in private arenas, spliced afterwards; ranges are disjoint so partitioning is safe
      std::vector<BuildNode> la{l}, ra{r};
      auto fut = std::async(std::launch::async, [&] { split(la, 0, level + 1); });
      split(ra, 0, level + 1);
      fut.get();
      const int32_t lbase = (int32_t)arena.size();
      for (auto n : la) {
        if (n.left >= 0) {
          n.left += lbase;
          n.right += lbase;
        }
        arena.push_back(n);
      }
      const int32_t rbase = (int32_t)arena.size();
      for (auto n : ra) {
        if (n.left >= 0) {
          n.left += rbase;
          n.right += rbase;
        }
        arena.push_back(n);
      }
      arena[idx].left = lbase;
      arena[idx].right = rbase;
    } else {
      const int32_t li = (int32_t)arena.size();
      arena.push_back(l);
      const int32_t ri = (int32_t)arena.size();
      arena.push_back(r);
      arena[idx].left = li;
      arena[idx].right = ri;
      split(arena, li, level + 1);
      split(arena, ri, level + 1);
    }
    arena[idx].count = 0;  // inner
  }
};

// ---- TLAS build  bvh/tlas.rs:74-134 --------------------------------------------------------------
struct TNode {
  Box bounds;
  int32_t left = -1, right = -1;
  uint32_t offset = 0, count = 0;
};

void tlas_split(std::vector<TNode>& tn, int32_t self, std::vector<uint32_t>& blas_order, const std::vector<Box>& blas_root,
                uint32_t offset, uint32_t count) {
  TNode n;
  n.offset = offset;
  n.count = count;
  n.bounds = empty_box();
  for (uint32_t i = offset; i < offset + count; ++i) {
    n.bounds.a = vmin(n.bounds.a, blas_root[blas_order[i]].a);
    n.bounds.b = vmax(n.bounds.b, blas_root[blas_order[i]].b);
  }
  const F4 extent = as_vec(n.bounds.b - n.bounds.a);
  int axis = 0;
  if (extent.y > extent.x) axis = 1;
  if (extent.z > axis_of(extent, axis)) axis = 2;
  const float split_pos = axis_of(n.bounds.a, axis) + axis_of(extent, axis) * 0.5f;
  uint32_t i = offset, j = offset + count;
  while (i < j) {
    const Box& rb = blas_root[blas_order[i]];
    const F4 centroid = to_point(as_vec(rb.b - rb.a) / 2.0f);  // AABB::get_centroid: the half-extent (aabb.rs:95-97)
    if (axis_of(centroid, axis) < split_pos) {
      ++i;
    } else {
      std::swap(blas_order[i], blas_order[j - 1]);
      --j;
    }
  }
  const uint32_t left_count = i - offset, right_count = count - left_count;
  tn[self] = n;
  if (left_count > 0 && right_count > 0) {
    const int32_t l = (int32_t)tn.size();
    tn.emplace_back();
    tlas_split(tn, l, blas_order, blas_root, offset, left_count);
    const int32_t r = (int32_t)tn.size();
    tn.emplace_back();
    tlas_split(tn, r, blas_order, blas_root, offset + left_count, right_count);
    tn[self].left = l;
    tn[self].right = r;
    tn[self].count = 0;
  }
}

// double -> IEEE half, rounded toward -inf (up == false) or +inf (up == true); NaN for NaN.  The arithmetic statement of
// the rounding; the encoder uses the bit form below, the self-test holds the two against each other.
uint16_t to_half_directed_ref(double v, bool up) {
  if (v != v) return 0x7E00u;
  const bool neg = std::signbit(v);
  double a = std::fabs(v);
  const bool mag_up = neg ? !up : up;  // direction on the magnitude
  const uint16_t sign = neg ? 0x8000u : 0u;
  if (a == 0.0) return sign;
  if (a > 65504.0) return (uint16_t)(sign | (mag_up ? 0x7C00u : 0x7BFFu));
  int e;
  (void)std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
  e -= 1;                    // a = m' * 2^e, m' in [1, 2)
  if (e < -14) {             // subnormal halves: multiples of 2^-24
    const double q = a * 16777216.0;
    double r = mag_up ? std::ceil(q) : std::floor(q);
    if (r >= 1024.0) return (uint16_t)(sign | 0x0400u);
    return (uint16_t)(sign | (uint16_t)r);
  }
  const double unit = std::ldexp(1.0, e - 10);
  double q = a / unit;  // in [1024, 2048)
  double r = mag_up ? std::ceil(q) : std::floor(q);
  if (r >= 2048.0) {
    r = 1024.0;
    e += 1;
    if (e > 15) return (uint16_t)(sign | 0x7C00u);
  }
  return (uint16_t)(sign | ((uint16_t)(e + 15) << 10) | ((uint16_t)r - 1024u));
}

// The same on the bits of the double: a value in the normal range of a half keeps its top ten mantissa bits, and moves one
// unit away from zero when anything was cut off and the direction asks for it (the carry runs into the exponent, and out
// of the largest one into infinity, as it should).  Everything else -- zero, subnormal halves, overflow, inf, NaN -- takes
// the arithmetic form.  (A scene build converts twelve of these per node, twice over.)
inline uint16_t to_half_directed(double v, bool up) {
  uint64_t bits;
  std::memcpy(&bits, &v, sizeof bits);
  const uint64_t mag = bits & 0x7FFFFFFFFFFFFFFFull;
  const int e = (int)(mag >> 52) - 1023;
  if (e < -14 || e > 15) return to_half_directed_ref(v, up);
  const bool neg = (bits >> 63) != 0;
  const uint64_t man = mag & 0x000FFFFFFFFFFFFFull;
  uint32_t h = ((uint32_t)(e + 15) << 10) | (uint32_t)(man >> 42);
  if ((neg ? !up : up) && (man & 0x3FFFFFFFFFFull)) ++h;
  return (uint16_t)((neg ? 0x8000u : 0u) | h);
}

// ---- device layout -------------------------------------------------------------------------------

struct DevBuilder {
  HostScene& s;
  std::vector<uint32_t> blas_base;  // first global primitive slot of each BLAS (TLAS order)
  float pad_rel = 0.0f, pad_abs = 0.0f;  // RAYCA_BUILDER_SAH: conservative boxes (see build_host_scene)
  // subtrees laid out as one leaf (host_scene.hpp kLeafNodeCost): the same decisions, by the same arithmetic, as the
  // device layout makes for a tree it holds (bvh_build.hip k_layout_sizes)
  float node_cost = 0.0f;
  uint32_t leaf_max = kLeafCollapseMax;
  std::vector<uint32_t> plan_prims;   // per arena node of the BLAS being laid out: primitives below it
  std::vector<uint8_t> plan_leafed;   // 1: the subtree is one leaf, 2: inside such a subtree
  void plan_leaves(const HostBlas& bl) {
    const std::vector<BuildNode>& a = bl.nodes;
    const size_t n = a.size();
    plan_prims.assign(n, 0u);
    plan_leafed.assign(n, 0);
    std::vector<float> cost(n, 0.0f);
    for (size_t i = n; i-- > 0;) {  // children sit behind their parents
      const BuildNode& bn = a[i];
      if (bn.left < 0) {
        plan_prims[i] = bn.count;
        cost[i] = (float)bn.count * area(bn.bounds);
        continue;
      }
      const uint32_t pl = plan_prims[bn.left], pr = plan_prims[bn.right];
      const float ar = area(bn.bounds), as_leaf = (float)(pl + pr) * ar, as_split = node_cost * ar + cost[bn.left] + cost[bn.right];
      const bool one_leaf = node_cost > 0.0f && pl + pr <= leaf_max && as_leaf <= as_split;
      plan_leafed[i] = one_leaf ? 1 : 0;
      cost[i] = one_leaf ? as_leaf : as_split;
      plan_prims[i] = pl + pr;
    }
    for (size_t i = 0; i < n; ++i) {  // everything below a one-leaf subtree is not laid out at all
      const BuildNode& bn = a[i];
      if (bn.left < 0 || plan_leafed[i] == 0) continue;
      plan_leafed[bn.left] = plan_leafed[bn.right] = 2;
    }
  }

  void put_box(DevNode& n, int side, const Box& b0) {
    Box b = b0;
    if (!(b0.a.x <= b0.b.x)) {  // the inverted box of an empty BLAS: make it unhittable
      b.a = point3(kNowhere, kNowhere, kNowhere);
      b.b = b.a;
    }
    if (pad_rel > 0.0f && b0.a.x <= b0.b.x) {  // (the nowhere box stays a point: padding would make it enterable)
      const float* lo[3] = {&b0.a.x, &b0.a.y, &b0.a.z};
      const float* hi[3] = {&b0.b.x, &b0.b.y, &b0.b.z};
      float* plo[3] = {&b.a.x, &b.a.y, &b.a.z};
      float* phi[3] = {&b.b.x, &b.b.y, &b.b.z};
      for (int k = 0; k < 3; ++k) {
        const float m = fmaxf(fmaxf(fabsf(*lo[k]), fabsf(*hi[k])), *hi[k] - *lo[k]);
        const float pad = m * pad_rel + pad_abs;
        *plo[k] = *lo[k] - pad;
        *phi[k] = *hi[k] + pad;
      }
    }
    put_box_raw(n, side, b);
  }
  static void put_box_raw(DevNode& n, int side, const Box& b) {
    if (side == 0) {
      n.q[0] = b.a.x; n.q[1] = b.a.y; n.q[2] = b.a.z; n.q[3] = b.b.x; n.q[4] = b.b.y; n.q[5] = b.b.z;
    } else {
      n.q[6] = b.a.x; n.q[7] = b.a.y; n.q[8] = b.a.z; n.q[9] = b.b.x; n.q[10] = b.b.y; n.q[11] = b.b.z;
    }
  }
  static uint32_t leaf_ref(uint32_t first, uint32_t count) { return kLeafFlag | ((count - 1u) << 25) | first; }

  uint32_t new_node() {
    s.dev_nodes.emplace_back();
    std::memset(&s.dev_nodes.back(), 0, sizeof(DevNode));
    return (uint32_t)s.dev_nodes.size() - 1;
  }
  // Every emit_* returns the packed child reference and reports, through `need`, how many stack
  // entries a traversal of that subtree can have pending at once: an inner node pushes one child and
  // descends into the other, so need = 1 + max(need(left), need(right)) -- except for the chain nodes
  // below, whose two boxes are identical so the left (leaf) child is always taken first and the
  // pushed entry is popped before the rest of the chain is entered.
  // a leaf range: one packed ref, or a chain of nodes re-testing the same box for > 64 primitives
  uint32_t emit_leaf(uint32_t first, uint32_t count, const Box& box, uint32_t& need) {
    if (count == 0) {
      need = 0;
      return kNoChild;  // a model without primitives: nothing to visit
    }
    if (count <= kLeafMaxPrims) {
      need = 0;
      return leaf_ref(first, count);
    }
    // iterative: chains can be thousands of nodes long with the reference's coarse leaves
    const uint32_t head = new_node();
    uint32_t cur = head;
    for (;;) {
      put_box(s.dev_nodes[cur], 0, box);
      put_box(s.dev_nodes[cur], 1, box);
      s.dev_nodes[cur].left = leaf_ref(first, kLeafMaxPrims);
      first += kLeafMaxPrims;
      count -= kLeafMaxPrims;
      if (count <= kLeafMaxPrims) {
        s.dev_nodes[cur].right = leaf_ref(first, count);
        break;
      }
      const uint32_t nxt = new_node();
      s.dev_nodes[cur].right = nxt;
      cur = nxt;
    }
    need = 1;
    return head;
  }
  // The BLAS below its root, without recursion.  Children sit behind their parents in the arena (both builders append
  // them that way; checked), so one backward pass gives every subtree's device-node count and stack need, one forward
  // pass gives every node its index -- the pre-order numbering the recursive form below produces: a node, its left
  // subtree, its right subtree -- and then every node is written independently, by all host threads.  (The recursive
  // walk over an arena in level order was 36 ms of cache misses for the atrium's 543 k nodes.)
  bool allow_flat = true;  // (the self-test switches it off to compare against the recursive form)
  bool emit_blas_flat(const HostBlas& bl, uint32_t base, uint32_t& root_ref, uint32_t& need) {
    const std::vector<BuildNode>& a = bl.nodes;
    const size_t n = a.size();
    if (n < 4096 || !allow_flat) return false;  // small trees: the recursive form
    std::vector<uint32_t> size(n), stack_need(n);
    for (size_t i = n; i-- > 0;) {
      const BuildNode& bn = a[i];
      if (bn.left < 0) {  // leaf: a packed reference, or a chain of ceil(count / 64) - 1 nodes
        const uint32_t chain = bn.count > kLeafMaxPrims ? (bn.count + kLeafMaxPrims - 1) / kLeafMaxPrims - 1 : 0u;
        size[i] = chain;
        stack_need[i] = chain ? 1u : 0u;
      } else {
        if ((size_t)bn.left <= i || (size_t)bn.right <= i || (size_t)bn.left >= n || (size_t)bn.right >= n) return false;
        const bool one_leaf = plan_leafed[i] != 0;   // (its parent holds a leaf reference, it takes no nodes)
        size[i] = one_leaf ? 0u : 1u + size[bn.left] + size[bn.right];
        stack_need[i] = one_leaf ? 0u : 1u + std::max(stack_need[bn.left], stack_need[bn.right]);
      }
    }
    const uint32_t first = (uint32_t)s.dev_nodes.size();
    std::vector<uint32_t> index(n);  // device index of an inner node / of the head of a leaf's chain
    index[0] = first;
    for (size_t i = 0; i < n; ++i) {
      const BuildNode& bn = a[i];
      if (bn.left < 0) continue;
      index[bn.left] = index[i] + 1u;
      index[bn.right] = index[i] + 1u + size[bn.left];
    }
    s.dev_nodes.resize((size_t)first + size[0]);
    auto ref_of = [&](size_t i) -> uint32_t {  // what the parent stores for child i
      const BuildNode& bn = a[i];
      if (bn.left >= 0) return plan_leafed[i] == 1 ? leaf_ref(base + bn.offset, plan_prims[i]) : index[i];
      if (bn.count == 0) return kNoChild;
      return bn.count <= kLeafMaxPrims ? leaf_ref(base + bn.offset, bn.count) : index[i];
    };
    parallel_chunks(n, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        const BuildNode& bn = a[i];
        if (plan_leafed[i] != 0) continue;  // the root of, or inside, a subtree that is one leaf
        if (bn.left >= 0) {
          DevNode& d = s.dev_nodes[index[i]];
          std::memset(&d, 0, sizeof d);
          put_box(d, 0, a[bn.left].bounds);
          put_box(d, 1, a[bn.right].bounds);
          d.left = ref_of((size_t)bn.left);
          d.right = ref_of((size_t)bn.right);
        } else if (bn.count > kLeafMaxPrims) {  // the chain of emit_leaf
          uint32_t cur = index[i], f = base + bn.offset, count = bn.count;
          for (;;) {
            DevNode& d = s.dev_nodes[cur];
            std::memset(&d, 0, sizeof d);
            put_box(d, 0, bn.bounds);
            put_box(d, 1, bn.bounds);
            d.left = leaf_ref(f, kLeafMaxPrims);
            f += kLeafMaxPrims;
            count -= kLeafMaxPrims;
            if (count <= kLeafMaxPrims) {
              d.right = leaf_ref(f, count);
              break;
            }
            d.right = cur + 1u;
            ++cur;
          }
        }
      }
    });
    root_ref = ref_of(0);
    need = stack_need[0];
    return true;
  }
  uint32_t emit_blas_node(const HostBlas& bl, uint32_t base, uint32_t idx, uint32_t& need) {
    if (idx == 0 && bl.dev_tree) {  // built and kept on the device: a run of node indices is set aside, the device fills it
      const uint32_t first = (uint32_t)s.dev_nodes.size();
      s.dev_nodes.resize((size_t)first + bl.dev_node_count);
      s.dev_segments.push_back(DeviceSegment{first, bl.dev_node_count, base, bl.dev_tree});
      need = bl.dev_need;
      return first;   // (more than 64 primitives: the root is an inner node or the head of a chain, node 0 of the run)
    }
    if (idx == 0) {
      plan_leaves(bl);
      uint32_t ref = 0;
      if (emit_blas_flat(bl, base, ref, need)) return ref;
    }
    const BuildNode& bn = bl.nodes[idx];
    if (bn.left < 0) return emit_leaf(base + bn.offset, bn.count, bn.bounds, need);
    if (plan_leafed[idx] == 1) {  // the whole subtree as one leaf
      need = 0;
      return leaf_ref(base + bn.offset, plan_prims[idx]);
    }
    const uint32_t n = new_node();
    put_box(s.dev_nodes[n], 0, bl.nodes[bn.left].bounds);
    put_box(s.dev_nodes[n], 1, bl.nodes[bn.right].bounds);
    uint32_t nl = 0, nr = 0;
    const uint32_t lr = emit_blas_node(bl, base, (uint32_t)bn.left, nl);
    s.dev_nodes[n].left = lr;
    const uint32_t rr = emit_blas_node(bl, base, (uint32_t)bn.right, nr);
    s.dev_nodes[n].right = rr;
    need = 1 + std::max(nl, nr);
    return n;
  }
  // TLAS leaf holding blas [offset, offset+count): BLAS roots tested one after the other
  uint32_t emit_blas_chain(uint32_t offset, uint32_t count, const Box& leaf_box, uint32_t& need) {
    if (count == 1) return emit_blas_node(s.blas[offset], blas_base[offset], 0, need);
    const uint32_t n = new_node();
    put_box(s.dev_nodes[n], 0, s.blas[offset].nodes[0].bounds);
    put_box(s.dev_nodes[n], 1, leaf_box);
    uint32_t nl = 0, nr = 0;
    const uint32_t lr = emit_blas_node(s.blas[offset], blas_base[offset], 0, nl);
    s.dev_nodes[n].left = lr;
    const uint32_t rr = emit_blas_chain(offset + 1, count - 1, leaf_box, nr);
    s.dev_nodes[n].right = rr;
    need = 1 + std::max(nl, nr);
    return n;
  }
  uint32_t emit_tlas(const std::vector<TNode>& tn, int32_t idx, uint32_t& need) {
    const TNode& t = tn[idx];
    if (t.left < 0) return emit_blas_chain(t.offset, t.count, t.bounds, need);
    const uint32_t n = new_node();
    put_box(s.dev_nodes[n], 0, tn[t.left].bounds);
    put_box(s.dev_nodes[n], 1, tn[t.right].bounds);
    uint32_t nl = 0, nr = 0;
    const uint32_t lr = emit_tlas(tn, t.left, nl);
    s.dev_nodes[n].left = lr;
    const uint32_t rr = emit_tlas(tn, t.right, nr);
    s.dev_nodes[n].right = rr;
    need = 1 + std::max(nl, nr);
    return n;
  }
};

// ---- 4-wide collapse ---------------------------------------------------------------------------
// Each wide node adopts the grandchildren of a binary node (largest box first) until it has four
// children.  Skipping a binary node skips its box test, which cannot change what the traversal finds:
// a child's box is contained in its parent's and the slab arithmetic is monotone in the box corners,
// so whenever a child box passes the test every box above it passes too.
struct WideBuilder {
  HostScene& s;
  struct Kid {
    uint32_t ref;
    float b[6];  // min xyz, max xyz
  };
  static float kid_area(const Kid& k) {
    const float ex = k.b[3] - k.b[0], ey = k.b[4] - k.b[1], ez = k.b[5] - k.b[2];
    return ex * ey + ey * ez + ez * ex;
  }
  void kids_of(uint32_t bin, Kid out[2]) const {
    const DevNode& n = s.dev_nodes[bin];
    out[0].ref = n.left;
    out[1].ref = n.right;
    for (int i = 0; i < 6; ++i) {
      out[0].b[i] = n.q[i];
      out[1].b[i] = n.q[6 + i];
    }
  }
  // the up-to-four children a wide node gets for binary node `bin_ref`
  int expand(uint32_t bin_ref, Kid kids[4]) const {
    int k = 2;
    kids_of(bin_ref, kids);
    while (k < 4) {
      int best = -1;
      float best_area = -1.0f;
      for (int i = 0; i < k; ++i) {
        if ((kids[i].ref & kLeafFlag) || kids[i].ref == kNoChild) continue;
        const float a = kid_area(kids[i]);
        if (best < 0 || a > best_area) {
          best = i;
          best_area = a;
        }
      }
      if (best < 0) break;
      Kid two[2];
      kids_of(kids[best].ref, two);
      // keep the left-to-right order of the binary tree (exhaustive mode visits children in index order)
      for (int i = k; i > best + 1; --i) kids[i] = kids[i - 1];
      kids[best] = two[0];
      kids[best + 1] = two[1];
      ++k;
    }
    return k;
  }

  // Subtrees a few levels down are collapsed by all host threads into arrays of their own (numbered from 0, pre-order)
  // and spliced in when the top of the tree -- built by this thread, in the same pre-order -- reaches them: the result is
  // the array the purely sequential recursion produces, node for node.
  struct Task {
    uint32_t bin_root = 0, need = 0;
    std::vector<DevNode4> nodes;
  };
  std::vector<Task> tasks;
  std::vector<std::pair<uint32_t, uint32_t>> task_of;  // (binary ref, task), sorted by ref
  bool allow_parallel = true;  // (the self-test switches it off to compare against the sequential recursion)

  void collect(uint32_t bin_ref, int depth) {
    if ((bin_ref & kLeafFlag) || bin_ref == kNoChild) return;
    if (depth == 0) {
      tasks.emplace_back();
      tasks.back().bin_root = bin_ref;
      return;
    }
    Kid kids[4];
    const int k = expand(bin_ref, kids);
    for (int i = 0; i < k; ++i) collect(kids[i].ref, depth - 1);
  }

  uint32_t build(uint32_t bin_ref, uint32_t& need) {
    tasks.clear();
    task_of.clear();
    if (allow_parallel && s.dev_nodes.size() >= 65536 && host_threads() > 1) {
      collect(bin_ref, 3);
      std::atomic<size_t> next{0};
      auto worker = [&] {
        for (size_t t; (t = next.fetch_add(1)) < tasks.size();) build_into(tasks[t].nodes, tasks[t].bin_root, tasks[t].need, false);
      };
      std::vector<std::thread> pool;
      const unsigned nt = (unsigned)std::min<size_t>(host_threads(), tasks.size());
      for (unsigned i = 1; i < nt; ++i) pool.emplace_back(worker);
      worker();
      for (std::thread& th : pool) th.join();
      for (size_t t = 0; t < tasks.size(); ++t) task_of.push_back({tasks[t].bin_root, (uint32_t)t});
      std::sort(task_of.begin(), task_of.end());
    }
    return build_into(s.dev_nodes4, bin_ref, need, true);
  }

  uint32_t build_into(std::vector<DevNode4>& out, uint32_t bin_ref, uint32_t& need, bool top) {
    if ((bin_ref & kLeafFlag) || bin_ref == kNoChild) {
      need = 0;
      return bin_ref;
    }
    if (top && !task_of.empty()) {
      const auto it = std::lower_bound(task_of.begin(), task_of.end(), std::make_pair(bin_ref, 0u));
      if (it != task_of.end() && it->first == bin_ref) {  // splice the finished subtree: local indices + base
        Task& t = tasks[it->second];
        const uint32_t base = (uint32_t)out.size();
        out.insert(out.end(), t.nodes.begin(), t.nodes.end());
        for (size_t i = base; i < out.size(); ++i)
          for (int c = 0; c < 4; ++c)
            if (!(out[i].child[c] & kLeafFlag) && out[i].child[c] != kNoChild) out[i].child[c] += base;
        std::vector<DevNode4>().swap(t.nodes);
        need = t.need;
        return base;  // a task's root is its node 0
      }
    }
    // iterative over an explicit work list would be needed for pathological depth only; the binary
    // chains for big leaves are collapsed 3 links at a time here
    Kid kids[4];
    const int k = expand(bin_ref, kids);
    const uint32_t idx = (uint32_t)out.size();
    out.emplace_back();
    uint32_t child_refs[4], child_need[4] = {0, 0, 0, 0};
    uint32_t worst = 0;
    for (int i = 0; i < k; ++i) {
      uint32_t nd = 0;
      child_refs[i] = build_into(out, kids[i].ref, nd, top);
      child_need[i] = nd;
      worst = std::max(worst, nd);
    }
    DevNode4& w = out[idx];
    for (int i = 0; i < 4; ++i) {
      if (i < k) {
        for (int a = 0; a < 3; ++a) {
          w.lo[a][i] = kids[i].b[a];
          w.hi[a][i] = kids[i].b[3 + a];
        }
        w.child[i] = child_refs[i];
      } else {
        // Unused slot: a zero-size box far away.  (An inverted +-FLT_MAX box does NOT work: (MAX - o) * rd
        // overflows to +-inf for |rd| > 1 and the slab test then passes.)  A point box passes only if the
        // three axis distances are equal, and any overflow makes them unordered, so it is never entered.
        for (int a = 0; a < 3; ++a) {
          w.lo[a][i] = kNowhere;
          w.hi[a][i] = kNowhere;
        }
        w.child[i] = kNoChild;
      }
      w.pad[i] = 0;
    }
    // pending entries: the nearest child is entered with k-1 siblings on the stack.  If all boxes are
    // identical (the chains that split big leaves) the children are taken in index order, so child i
    // is entered with k-1-i siblings pending.
    bool same = true;
    for (int i = 1; i < k && same; ++i) same = std::memcmp(kids[i].b, kids[0].b, sizeof kids[0].b) == 0;
    if (same) {
      need = 0;
      for (int i = 0; i < k; ++i) need = std::max(need, (uint32_t)(k - 1 - i) + child_need[i]);
    } else {
      need = (uint32_t)(k - 1) + worst;
    }
    return idx;
  }
};

bool fetch_index(const RaycaSceneDesc& d, const RaycaPrimitive& p, uint32_t i, uint32_t& out) {
  const uint64_t off = p.index_byte_offset;
  switch (p.index_type) {
    case RAYCA_INDEX_U8:
      if (off + i >= d.index_byte_count) return false;
      out = d.index_bytes[off + i];
      return true;
    case RAYCA_INDEX_U16: {
      if (off + 2ull * i + 2 > d.index_byte_count) return false;
      uint16_t v;
      std::memcpy(&v, d.index_bytes + off + 2ull * i, 2);
      out = v;
      return true;
    }
    case RAYCA_INDEX_U32: {
      if (off + 4ull * i + 4 > d.index_byte_count) return false;
      std::memcpy(&out, d.index_bytes + off + 4ull * i, 4);
      return true;
    }
    default:
      return false;  // panic!("Index type not supported")  primitive.rs:258
  }
}

void set_ext(PrimExt& e, int k, Color c, F4 n, F4 t, F4 b, F2 uv) {
  e.color[k][0] = c.r; e.color[k][1] = c.g; e.color[k][2] = c.b; e.color[k][3] = c.a;
  e.normal[k][0] = n.x; e.normal[k][1] = n.y; e.normal[k][2] = n.z;
  e.tangent[k][0] = t.x; e.tangent[k][1] = t.y; e.tangent[k][2] = t.z;
  e.bitangent[k][0] = b.x; e.bitangent[k][1] = b.y; e.bitangent[k][2] = b.z;
  e.uv[k][0] = uv.x; e.uv[k][1] = uv.y;
}

}  // namespace

// Self-check of the two parallel layout passes against their recursive forms on a random tree (uneven depths, mostly small leaves
// and some of 65..264 primitives so that the 64-primitive chains occur): the arrays must be identical, byte for byte.
int32_t selftest_device_layouts(std::string& err) {
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto next = [&rng]() {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return rng;
  };
  auto rnd = [&](float lo, float hi) { return lo + (hi - lo) * (float)((next() >> 40) & 0xFFFFFF) / 16777216.0f; };
  HostBlas bl;
  const uint32_t prims = 1u << 20;
  bl.prims.resize(prims);
  for (uint32_t i = 0; i < prims; ++i) bl.prims[i] = i;
  bl.nodes.emplace_back();
  bl.nodes[0].offset = 0;
  bl.nodes[0].count = prims;
  bl.nodes[0].bounds = Box{point3(-1, -1, -1), point3(1, 1, 1)};
  for (size_t i = 0; i < bl.nodes.size(); ++i) {  // breadth-first: children behind their parent, like the builders
    const BuildNode n = bl.nodes[i];
    const uint32_t leaf_max = (next() % 16u == 0u) ? 65u + (uint32_t)(next() % 200u) : 1u + (uint32_t)(next() % 8u);
    if (n.count <= leaf_max) continue;
    const uint32_t cut = 1u + (uint32_t)(next() % (n.count - 1u));
    BuildNode l, r;
    l.offset = n.offset; l.count = cut;
    r.offset = n.offset + cut; r.count = n.count - cut;
    for (BuildNode* c : {&l, &r}) {
      const float x0 = rnd(-1, 1), y0 = rnd(-1, 1), z0 = rnd(-1, 1);
      c->bounds = Box{point3(x0, y0, z0), point3(x0 + rnd(0, 0.5f), y0 + rnd(0, 0.5f), z0 + rnd(0, 0.5f))};
    }
    bl.nodes[i].left = (int32_t)bl.nodes.size();
    bl.nodes.push_back(l);
    bl.nodes[i].right = (int32_t)bl.nodes.size();
    bl.nodes.push_back(r);
    bl.nodes[i].count = 0;
  }
  for (const float node_cost : {0.0f, kLeafNodeCost}) {   // the tree as built, and with subtrees laid out as single leaves
    HostScene a, b;
    uint32_t ref[2], need[2], ref4[2], need4[2];
    HostScene* scenes[2] = {&a, &b};
    for (int pass = 0; pass < 2; ++pass) {
      HostScene& s = *scenes[pass];
      s.blas.push_back(bl);
      DevBuilder db{s, {0u}};
      db.node_cost = node_cost;
      db.pad_rel = 1.52587890625e-05f;
      db.pad_abs = 1e-6f;
      db.allow_flat = pass == 1;
      ref[pass] = db.emit_blas_node(s.blas[0], 0, 0, need[pass]);
      WideBuilder wb{s};
      wb.allow_parallel = pass == 1;
      ref4[pass] = wb.build(ref[pass], need4[pass]);
    }
    if (node_cost == 0.0f && a.dev_nodes.size() < 65536) { err = "layout self-test: the random tree came out too small to exercise the parallel passes"; return RAYCA_ERR_BAD_ARG; }
    if (ref[0] != ref[1] || need[0] != need[1] || a.dev_nodes.size() != b.dev_nodes.size() ||
        std::memcmp(a.dev_nodes.data(), b.dev_nodes.data(), a.dev_nodes.size() * sizeof(DevNode)) != 0) {
      err = "layout self-test: the flat binary-node pass differs from the recursive one";
      return RAYCA_ERR_BAD_ARG;
    }
    if (ref4[0] != ref4[1] || need4[0] != need4[1] || a.dev_nodes4.size() != b.dev_nodes4.size() ||
        std::memcmp(a.dev_nodes4.data(), b.dev_nodes4.data(), a.dev_nodes4.size() * sizeof(DevNode4)) != 0) {
      err = "layout self-test: the parallel 4-wide collapse differs from the sequential one";
      return RAYCA_ERR_BAD_ARG;
    }
  }
  return RAYCA_OK;
}

// Self-check of the outward fp16 rounding the steering boxes rely on: every finite half maps to itself in both
// directions, and for values between two neighbouring halves `down` and `up` return exactly those neighbours.
int32_t selftest_half_rounding(std::string& err) {
  auto decode = [](uint16_t h) -> double {
    const int e = (h >> 10) & 31, m = h & 1023;
    const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp(1024.0 + m, e - 25);
    return (h & 0x8000u) ? -v : v;
  };
  auto fail = [&](const char* what, double v) {
    char buf[128];
    snprintf(buf, sizeof buf, "half rounding: %s at %.17g", what, v);
    err = buf;
    return (int32_t)RAYCA_ERR_BAD_ARG;
  };
  for (uint32_t h = 0; h < 65536u; ++h) {
    if (((h >> 10) & 31u) == 31u) continue;  // inf / NaN
    const double v = decode((uint16_t)h);
    if (decode(to_half_directed(v, false)) != v || decode(to_half_directed(v, true)) != v) return fail("exact value moved", v);
    // the next half above (same sign ordering on the real line)
    const uint16_t up_bits = (h & 0x8000u) ? (uint16_t)((h & 0x7FFFu) == 0 ? 0x0001u : h - 1u) : (uint16_t)(h + 1u);
    if (((up_bits >> 10) & 31u) == 31u) continue;
    const double w = decode(up_bits);
    if (!(w > v)) continue;  // -0 / +0
    for (const double f : {0.25, 0.5, 0.999}) {
      const double x = v + (w - v) * f;
      const double lo = decode(to_half_directed(x, false)), hi = decode(to_half_directed(x, true));
      if (lo != v || hi != w) return fail("bracket is not the two neighbours", x);
    }
  }
  // the bit form against the arithmetic form: every binade a half can hold and the ones either side, both signs, values
  // on, just above and just below a half's grid
  uint64_t lcg = 0x9E3779B97F4A7C15ull;
  for (int e = -30; e <= 20; ++e)
    for (int i = 0; i < 4096; ++i) {
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      uint64_t man = lcg >> 12;
      if (i % 4 == 1) man &= ~0x3FFFFFFFFFFull;                   // exactly representable
      if (i % 4 == 2) man = (man & ~0x3FFFFFFFFFFull) | 1ull;      // one ulp of the double above
      if (i % 4 == 3) man |= 0x3FFFFFFFFFFull;                     // just below the next
      const uint64_t bits = ((uint64_t)((i >> 2) & 1) << 63) | ((uint64_t)(e + 1023) << 52) | man;
      double x;
      std::memcpy(&x, &bits, sizeof x);
      for (const bool up : {false, true})
        if (to_half_directed(x, up) != to_half_directed_ref(x, up)) return fail("bit form differs from the arithmetic form", x);
    }
  if (to_half_directed(1e6, true) != 0x7C00u || to_half_directed(1e6, false) != 0x7BFFu) return fail("overflow", 1e6);
  if (to_half_directed(-1e6, false) != 0xFC00u || to_half_directed(-1e6, true) != 0xFBFFu) return fail("overflow", -1e6);
  if (to_half_directed(1e-9, true) != 0x0001u || to_half_directed(1e-9, false) != 0x0000u) return fail("underflow", 1e-9);
  return RAYCA_OK;
}

void set_device_blas_builder(BlasBuildFn fn, uint32_t device) {
  g_device_builder = fn;
  g_device_ordinal = device;
}

// The three node formats next to the binary f32 one (RAYCA_BUILDER_SAH scenes): the same tree collapsed to 4-wide nodes, and
// both with fp16 steering boxes.  Reads `s.dev_nodes`, writes `dev_nodes4`, `dev_nodes_h`, `dev_nodes4_h` and their scalars; the
// library runs it on a thread of its own while the first frames are already traversing the binary nodes.
void finish_node_formats(HostScene& s, const std::atomic<bool>* cancel) {
  if (!s.other_formats_wanted) return;
  auto cancelled = [cancel] { return cancel && cancel->load(std::memory_order_relaxed); };
  if (cancelled()) return;
  static const bool timing = getenv("RAYCA_BUILD_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[rayca build] %-28s %8.1f ms\n", what, std::chrono::duration<float, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  // the 4-wide collapse reads the binary nodes and writes its own array: it runs next to the fp16 encoding of the binary
  // nodes below and is joined before its own fp16 copy is made
  std::thread wide_thread([&s] {
    WideBuilder wb{s};
    uint32_t need4 = 0;
    s.root_ref4 = wb.build(s.root_ref, need4);
    s.max_depth4 = need4;
  });
  struct Joiner {
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } wide_joiner{wide_thread};
  // fp16 steering boxes: centre of the scene box, power-of-two scale that maps the half extent to <= 2^14
  float half_extent = 0.0f;
  const float lo3[3] = {s.root_min.x, s.root_min.y, s.root_min.z}, hi3[3] = {s.root_max.x, s.root_max.y, s.root_max.z};
  for (int c = 0; c < 3; ++c) {
    s.half_center[c] = 0.5f * (lo3[c] + hi3[c]);
    if (!(s.half_center[c] == s.half_center[c]) || std::fabs(s.half_center[c]) > 1e30f) s.half_center[c] = 0.0f;
    half_extent = std::max(half_extent, std::max(std::fabs(hi3[c] - s.half_center[c]), std::fabs(lo3[c] - s.half_center[c])));
  }
  int k = 0;
  if (half_extent > 0.0f && half_extent < 1e30f) {
    (void)std::frexp(half_extent * 1.0009765625f, &k);  // half_extent (+ padding) < 2^k
    k = 14 - k;
  }
  k = std::max(-100, std::min(100, k));
  s.half_scale = std::ldexp(1.0f, k);
  auto cv = [&](float x, int axis, bool up) -> uint16_t {
    // only NaNs are unenterable here; the "nowhere" slots are recognised box by box below.  Finite coordinates beyond
    // the fp16 range saturate OUTWARD (to_half_directed: max -> +inf, min -> -inf), so the box stays conservative
    // and the frame cannot depend on the node format chosen, whatever the scene's extent.
    if (!(x == x)) return 0x7E00u;
    return to_half_directed(((double)x - (double)s.half_center[axis]) * (double)s.half_scale, up);
  };
  s.dev_nodes_h.resize(s.dev_nodes.size());
  parallel_chunks(s.dev_nodes.size(), [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const DevNode& n = s.dev_nodes[i];
      DevNodeH& h = s.dev_nodes_h[i];
      // q: l.min xyz, l.max xyz, r.min xyz, r.max xyz
      for (int j = 0; j < 12; ++j) h.h[j] = cv(n.q[j], j % 3, (j / 3) & 1);
      // an inverted (empty) box and the zero-size "nowhere" box of an unused slot must stay unenterable after
      // outward rounding
      for (int side = 0; side < 2; ++side) {
        bool nowhere = true;
        for (int j = 0; j < 6; ++j) nowhere = nowhere && n.q[side * 6 + j] == kNowhere;
        if (nowhere || !(n.q[side * 6] <= n.q[side * 6 + 3]))
          for (int j = 0; j < 6; ++j) h.h[side * 6 + j] = 0x7E00u;
      }
      h.left = n.left;
      h.right = n.right;
    }
  });
  lap("  binary nodes, fp16");
  wide_thread.join();
  lap("  4-wide nodes (rest)");
  if (cancelled()) return;
  s.dev_nodes4_h.resize(s.dev_nodes4.size());
  parallel_chunks(s.dev_nodes4.size(), [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const DevNode4& n = s.dev_nodes4[i];
      DevNode4H& h = s.dev_nodes4_h[i];
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 4; ++c) {
          bool empty = !(n.lo[0][c] <= n.hi[0][c]);
          if (n.lo[0][c] == kNowhere && n.hi[0][c] == kNowhere && n.lo[1][c] == kNowhere && n.hi[1][c] == kNowhere && n.lo[2][c] == kNowhere &&
              n.hi[2][c] == kNowhere)
            empty = true;  // unused slot (WideBuilder)
          h.lo[a][c] = empty ? 0x7E00u : cv(n.lo[a][c], a, false);
          h.hi[a][c] = empty ? 0x7E00u : cv(n.hi[a][c], a, true);
        }
      for (int c = 0; c < 4; ++c) h.child[c] = n.child[c];
    }
  });
  lap("  4-wide nodes, fp16");
}

int32_t build_host_scene(const RaycaSceneDesc& d, bool use_bvh, uint32_t builder, HostScene& s, std::string& err, const BuildHooks& hooks) {
  // RAYCA_BUILD_TIMING=1: phase times of the host build on stderr
  static const bool timing = getenv("RAYCA_BUILD_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[rayca build] %-28s %8.1f ms\n", what, std::chrono::duration<float, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  if (d.abi_version != RAYCA_ABI_VERSION) { err = "abi version mismatch"; return RAYCA_ERR_BAD_ARG; }
  if (d.node_count && !d.nodes) { err = "nodes is null"; return RAYCA_ERR_BAD_ARG; }
  if (d.vertex_count && !d.positions) { err = "positions is null"; return RAYCA_ERR_BAD_ARG; }
  const uint32_t N = d.node_count;

  // ---- SceneDrawInfo::new: world(i) = world(parent) * local(i)  (scene.rs:213-215,228,242) ----
  s.local_trs.resize(N);
  s.world_trs.resize(N);
  std::vector<std::vector<uint32_t>> children(N);
  std::vector<uint32_t> tops;
  for (uint32_t i = 0; i < N; ++i) {
    const RaycaNode& n = d.nodes[i];
    if (n.parent >= (int32_t)i) { err = "node " + std::to_string(i) + ": parent must precede child"; return RAYCA_ERR_BAD_ARG; }
    s.local_trs[i] = trs_from_abi(n.trs);
    if (n.parent < 0) {
      s.world_trs[i] = s.local_trs[i];
      tops.push_back(i);
    } else {
      s.world_trs[i] = trs_compose(s.world_trs[n.parent], s.local_trs[i]);
      children[n.parent].push_back(i);
    }
  }
  // traversal order = DFS pre-order (children ascending)
  std::vector<uint32_t> order;
  order.reserve(N);
  {
    std::vector<uint32_t> stack(tops.rbegin(), tops.rend());
    while (!stack.empty()) {
      const uint32_t n = stack.back();
      stack.pop_back();
      order.push_back(n);
      for (auto it = children[n].rbegin(); it != children[n].rend(); ++it) stack.push_back(*it);
    }
  }
  std::vector<uint32_t> mesh_nodes, light_nodes;
  for (uint32_t n : order) {
    const RaycaNode& nd = d.nodes[n];
    if (nd.mesh != RAYCA_NONE) {
      if (nd.mesh >= d.mesh_count) { err = "mesh index out of range"; return RAYCA_ERR_BAD_ARG; }
      mesh_nodes.push_back(n);
    }
    if (nd.light != RAYCA_NONE) {
      if (nd.light >= d.light_count) { err = "light index out of range"; return RAYCA_ERR_BAD_ARG; }
      light_nodes.push_back(n);
    }
    if (nd.camera != RAYCA_NONE && !s.has_camera) {
      if (nd.camera >= d.camera_count) { err = "camera index out of range"; return RAYCA_ERR_BAD_ARG; }
      s.has_camera = true;
      s.camera_node = n;
      s.camera_yfov = d.cameras[nd.camera].yfov_radians;
    }
  }
  for (uint32_t n : light_nodes) {
    const RaycaLight& l = d.lights[d.nodes[n].light];
    HostLight hl;
    hl.kind = l.kind;
    hl.node = n;
    hl.material = l.material;
    hl.intensity = l.intensity;
    hl.color = rgba(l.color[0], l.color[1], l.color[2], l.color[3]);
    hl.attenuation = vec3(l.attenuation[0], l.attenuation[1], l.attenuation[2]);
    hl.ab = vec3(l.ab[0], l.ab[1], l.ab[2]);
    hl.ac = vec3(l.ac[0], l.ac[1], l.ac[2]);
    hl.local = s.local_trs[n];
    s.lights.push_back(hl);
  }
  s.materials.assign(d.materials, d.materials + d.material_count);
  s.textures.assign(d.textures, d.textures + d.texture_count);
  s.images.assign(d.images, d.images + d.image_count);
  s.image_bytes.assign(d.image_bytes, d.image_bytes + d.image_byte_count);

  // ---- BvhScene::from_scene: models in ascending id; per model meshes then quad lights ----------
  std::vector<uint32_t> models;
  for (uint32_t n : mesh_nodes) models.push_back(d.nodes[n].model);
  for (uint32_t n : light_nodes)
    if (d.lights[d.nodes[n].light].kind == RAYCA_LIGHT_QUAD) models.push_back(d.nodes[n].model);
  std::sort(models.begin(), models.end());
  models.erase(std::unique(models.begin(), models.end()), models.end());

  struct TriJob {  // a run of triangles of one mesh primitive under one node, and where its HostPrims go
    uint32_t node, pi, first_tri, first_out, count;
    Mat3 tangent_matrix, normal_matrix;
    uint32_t model, blas_pos;   // its BLAS and the position of its first triangle in that BLAS's initial order
  };
  struct OtherPrim { uint32_t prim, model, blas_pos; };   // spheres, quad-light triangles: made where they are met, few
  std::vector<OtherPrim> others;
  std::vector<TriJob> jobs;
  std::vector<HostBlas> blas(models.size());
  {  // one allocation for the flatten-order array: growing it piecewise re-copied 0.5 KB per triangle several times over
    size_t total = 2 * light_nodes.size();
    for (uint32_t node : mesh_nodes) {
      const RaycaMesh& mesh = d.meshes[d.nodes[node].mesh];
      for (uint32_t pi = mesh.first_primitive; pi < mesh.first_primitive + mesh.primitive_count && pi < d.primitive_count; ++pi)
        total += d.primitives[pi].geometry == RAYCA_GEOMETRY_SPHERE ? 1u : d.primitives[pi].index_count / 3u;
    }
    s.prims.reserve(total);
    s.ext.reserve(total);
  }
  for (size_t m = 0; m < models.size(); ++m) {
    blas[m].model = models[m];
    for (uint32_t node : mesh_nodes) {
      if (d.nodes[node].model != models[m]) continue;
      const RaycaMesh& mesh = d.meshes[d.nodes[node].mesh];
      const Trs& trs = s.world_trs[node];
      const Mat3 tangent_matrix = mat3_from_trs(trs);                             // primitive.rs:203-204
      const Mat3 normal_matrix = mat3_transpose(mat3_from_inverse_trs(trs));      // primitive.rs:206-207
      for (uint32_t pi = mesh.first_primitive; pi < mesh.first_primitive + mesh.primitive_count; ++pi) {
        if (pi >= d.primitive_count) { err = "primitive index out of range"; return RAYCA_ERR_BAD_ARG; }
        const RaycaPrimitive& p = d.primitives[pi];
        if (p.geometry == RAYCA_GEOMETRY_SPHERE) {  // from_sphere  primitive.rs:262-274
          HostPrim hp{};
          hp.kind = RAYCA_GEOMETRY_SPHERE;
          hp.node = node;
          hp.material = p.material;
          hp.center = point3(p.sphere_center[0], p.sphere_center[1], p.sphere_center[2]);
          hp.radius = p.sphere_radius;
          PrimExt ext;
          std::memset(&ext, 0, sizeof ext);
          ext.material = p.material;
          ext.kind = hp.kind;
          ext.node = node;
          hp.src = (uint32_t)s.prims.size();
          others.push_back(OtherPrim{hp.src, (uint32_t)m, (uint32_t)blas[m].prims.size()});
          blas[m].prims.push_back(hp.src);
          s.prims.push_back(hp);
          s.ext.push_back(ext);
          s.sphere_count++;
          continue;
        }
        // from_triangle_mesh_impl  primitive.rs:209-234: the triangles are independent -- slots are reserved
        // here, in order, and filled by all host threads below
        const uint32_t ntri = p.index_count / 3;
        const uint32_t first_out = (uint32_t)s.prims.size();
        for (uint32_t at = 0; at < ntri; at += 8192)  // pieces, so that one huge mesh still feeds every thread
          jobs.push_back(TriJob{node, pi, at, first_out + at, std::min<uint32_t>(8192u, ntri - at), tangent_matrix, normal_matrix, (uint32_t)m,
                                (uint32_t)blas[m].prims.size() + at});
        s.prims.resize((size_t)first_out + ntri);
        s.ext.resize((size_t)first_out + ntri);
        {
          const size_t at = blas[m].prims.size();
          blas[m].prims.resize(at + ntri);  // (geometric growth: a model of many small meshes must not reallocate per mesh)
          for (uint32_t t = 0; t < ntri; ++t) blas[m].prims[at + t] = first_out + t;
        }
        s.triangle_count += ntri;
      }
    }
    for (uint32_t node : light_nodes) {  // from_quad_light  primitive.rs:310-346
      const RaycaLight& l = d.lights[d.nodes[node].light];
      if (d.nodes[node].model != models[m] || l.kind != RAYCA_LIGHT_QUAD) continue;
      const F4 ab = vec3(l.ab[0], l.ab[1], l.ab[2]), ac = vec3(l.ac[0], l.ac[1], l.ac[2]);
      const F4 normal = normalized(cross(ab, ac));
      const F4 qa = point3(0, 0, 0);
      const F4 a = qa, b = qa + ab, dd = (qa + ab) + ac, c = qa + ac;
      const F4 tri[2][3] = {{a, dd, b}, {a, c, dd}};
      for (int t = 0; t < 2; ++t) {
        HostPrim hp{};
        hp.kind = RAYCA_GEOMETRY_TRIANGLE_MESH;
        hp.node = node;
        hp.material = l.material;
        PrimExt ext;
        std::memset(&ext, 0, sizeof ext);
        for (int k = 0; k < 3; ++k) {
          hp.p[k] = tri[t][k];
          set_ext(ext, k, white(), normal, vec3(0, 0, 0), vec3(0, 0, 0), F2{0, 0});
        }
        hp.centroid = ((to_vec(hp.p[0]) + to_vec(hp.p[1])) + to_vec(hp.p[2])) * 0.3333f;
        ext.material = l.material;
        ext.kind = hp.kind;
        ext.node = node;
        hp.src = (uint32_t)s.prims.size();
        others.push_back(OtherPrim{hp.src, (uint32_t)m, (uint32_t)blas[m].prims.size()});
        blas[m].prims.push_back(hp.src);
        s.prims.push_back(hp);
        s.ext.push_back(ext);
        s.triangle_count++;
      }
    }
  }
  lap("flatten: graph, slots");
  // One pass over the triangles does everything that is per primitive: the flatten-order record and its shading record, the
  // world-space vertices / centroid / box (cache_world), the 9 floats of the device's triangle array, and -- for a BLAS the
  // device builds -- its row of the SoA the device builder reads.  (Three passes of all host threads before: each touched
  // the 0.5 KB per primitive again and paid a round of thread starts.)
  static const bool host_only = getenv("RAYCA_HOST_BUILD") != nullptr;
  const BlasBuildFn device_builder = g_device_builder;   // (thread_local: captured here, the trees are built by other threads)
  using Soa = std::vector<float, DefaultInitAllocator<float>>;   // nine planes of n floats, one block, BLAS order
  std::vector<Soa> soas(blas.size());
  for (size_t m = 0; m < blas.size(); ++m)
    if (device_builder && !host_only && use_bvh && blas[m].prims.size() >= kDeviceBuildMin) soas[m].resize(9 * blas[m].prims.size());
  s.tris.resize(s.prims.size() * 9);
  auto finish_prim = [&](uint32_t prim, uint32_t model, uint32_t blas_pos) {
    HostPrim& p = s.prims[prim];
    cache_world(p, s.world_trs[p.node]);
    float* tv = &s.tris[(size_t)prim * 9];
    for (int k = 0; k < 3; ++k) {
      tv[3 * k + 0] = p.wp[k].x;
      tv[3 * k + 1] = p.wp[k].y;
      tv[3 * k + 2] = p.wp[k].z;
    }
    Soa& soa = soas[model];
    if (!soa.empty()) {
      const size_t n = soa.size() / 9, i = blas_pos;
      float* const q = soa.data();
      q[i] = p.wcentroid.x; q[n + i] = p.wcentroid.y; q[2 * n + i] = p.wcentroid.z;
      q[3 * n + i] = p.wmin.x; q[4 * n + i] = p.wmin.y; q[5 * n + i] = p.wmin.z;
      q[6 * n + i] = p.wmax.x; q[7 * n + i] = p.wmax.y; q[8 * n + i] = p.wmax.z;
    }
  };
  if (!jobs.empty()) {  // fill the reserved triangle slots with all host threads
    std::atomic<int> failed{0};
    const char* first_error = nullptr;
    std::atomic<size_t> next_job{0};
    auto worker = [&] {
      for (;;) {
        const size_t j = next_job.fetch_add(1);
        if (j >= jobs.size() || failed.load(std::memory_order_relaxed)) return;
        const TriJob& job = jobs[j];
        const RaycaPrimitive& p = d.primitives[job.pi];
        for (uint32_t t = 0; t < job.count; ++t) {
          const uint32_t tri = job.first_tri + t;
          HostPrim& hp = s.prims[job.first_out + t];  // written in place, every field (the array is not pre-zeroed)
          hp.kind = RAYCA_GEOMETRY_TRIANGLE_MESH;
          hp.node = job.node;
          hp.material = p.material;
          hp.center = point3(0, 0, 0);
          hp.radius = 0.0f;
          PrimExt& ext = s.ext[job.first_out + t];
          std::memset(&ext, 0, sizeof ext);
          const char* problem = nullptr;
          for (int k = 0; k < 3; ++k) {
            uint32_t idx;
            if (!fetch_index(d, p, tri * 3 + (uint32_t)k, idx)) { problem = "index fetch out of range or unsupported index type"; break; }
            if (idx >= p.vertex_count || p.first_vertex + idx >= d.vertex_count) { problem = "vertex index out of range"; break; }
            const uint32_t v = p.first_vertex + idx;
            hp.p[k] = point3(d.positions[3 * v], d.positions[3 * v + 1], d.positions[3 * v + 2]);
            const Color c = d.colors ? rgba(d.colors[4 * v], d.colors[4 * v + 1], d.colors[4 * v + 2], d.colors[4 * v + 3]) : white();
            const F4 nrm = d.normals ? vec3(d.normals[3 * v], d.normals[3 * v + 1], d.normals[3 * v + 2]) : vec3(0, 0, 1);
            const F4 tan = d.tangents ? vec3(d.tangents[3 * v], d.tangents[3 * v + 1], d.tangents[3 * v + 2]) : vec3(0, 0, 0);
            const F4 bit = d.bitangents ? vec3(d.bitangents[3 * v], d.bitangents[3 * v + 1], d.bitangents[3 * v + 2]) : vec3(0, 0, 0);
            const F2 uv = d.uvs ? F2{d.uvs[2 * v], d.uvs[2 * v + 1]} : F2{0, 0};
            set_ext(ext, k, c, mat3_apply(job.normal_matrix, nrm), mat3_apply(job.tangent_matrix, tan), mat3_apply(job.tangent_matrix, bit), uv);
          }
          if (problem) {
            if (!failed.exchange(1)) first_error = problem;
            return;
          }
          hp.centroid = ((to_vec(hp.p[0]) + to_vec(hp.p[1])) + to_vec(hp.p[2])) * 0.3333f;  // triangle.rs:59-63
          ext.material = p.material;
          ext.kind = hp.kind;
          ext.node = job.node;
          hp.src = job.first_out + t;
          finish_prim(job.first_out + t, job.model, job.blas_pos + t);
        }
      }
    };
    const unsigned nthreads = std::max(1u, std::min<unsigned>(host_threads(), (unsigned)jobs.size()));
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < nthreads; ++i) pool.emplace_back(worker);
    worker();
    for (std::thread& th : pool) th.join();
    if (failed.load()) { err = first_error ? first_error : "bad triangle data"; return RAYCA_ERR_BAD_ARG; }
  }
  for (const OtherPrim& o : others) finish_prim(o.prim, o.model, o.blas_pos);
  lap("flatten primitives, world space, SoA");
  if (hooks.on_prims_ready) hooks.on_prims_ready();

  // ---- Tlas::new: one BLAS per model ------------------------------------------------------------
  const unsigned hw = host_threads();
  unsigned par_levels = 0;
  while ((1u << par_levels) < hw) ++par_levels;
  std::vector<Box> blas_root(blas.size());
  // reference order of every primitive inside its BLAS (needed by both builders: it is the tie rule)
  std::vector<std::vector<uint32_t>> ref_prims(blas.size());
  std::vector<std::vector<BuildNode>> ref_nodes(blas.size());
  const uint32_t device_ordinal = g_device_ordinal;
  std::mutex lap_mu;
  // the traversed tree of a device-built BLAS stays on the device and is laid out there (HostBlas::dev_tree) -- unless the
  // other node formats are to be made right here, on the host, from the host's copy of the binary nodes
  const bool keep_trees = !hooks.with_formats;
  // (RAYCA_LEAF_NODE_COST / RAYCA_LEAF_MAX: experiments with the price of a node step -- 0 lays the tree out as built -- and
  // with the cap on a folded leaf, which hardly ever binds: 4, 8, 16 and 32 give the same 135 379 nodes on the atrium)
  static const float leaf_node_cost = [] { const char* e = getenv("RAYCA_LEAF_NODE_COST"); return e ? (float)atof(e) : kLeafNodeCost; }();
  static const uint32_t leaf_max = [] { const char* e = getenv("RAYCA_LEAF_MAX"); return e ? (uint32_t)std::min(std::max(atoi(e), 1), (int)kLeafMaxPrims) : kLeafCollapseMax; }();
  void* const bstream[2] = {hooks.build_streams ? hooks.build_streams[0] : nullptr, hooks.build_streams ? hooks.build_streams[1] : nullptr};
  s.dev_segments.clear();
  s.dev_trees.clear();
  // what the device builder reads: world centroids and boxes of a BLAS's primitives, SoA, in the BLAS's initial order (the
  // two trees of a RAYCA_BUILDER_SAH scene start from the same order and share one copy)
  auto build_blas = [&](size_t m, bool seed_origin, std::vector<uint32_t>& order, std::vector<BuildNode>& nodes, std::string& build_err, bool quiet,
                        const Soa& soa, BlasDeviceTree* keep, void* stream) {
    auto lap = [&](const char* what) {
      if (!timing || quiet) return;
      std::lock_guard<std::mutex> lock(lap_mu);
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[rayca build] %-28s %8.1f ms\n", what, std::chrono::duration<float, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    BlasBuilder bb{&s, &order, use_bvh ? 255u : 0u, par_levels + 1, seed_origin};
    std::vector<BuildNode> arena(1);
    arena[0].offset = 0;
    arena[0].count = (uint32_t)order.size();
    arena[0].bounds = bb.range_bounds(0, arena[0].count);
    const uint32_t n = arena[0].count;
    if (soa.size() == 9 * (size_t)n && n > 0) {
      // the same recursion, level by level on the GPU (bvh_build.hip): identical tree, boxes and order
      BlasBuildInput in{};
      for (int c = 0; c < 3; ++c) {
        in.cent[c] = soa.data() + (size_t)c * n;
        in.bmin[c] = soa.data() + (size_t)(3 + c) * n;
        in.bmax[c] = soa.data() + (size_t)(6 + c) * n;
      }
      in.count = n;
      in.root_min[0] = arena[0].bounds.a.x; in.root_min[1] = arena[0].bounds.a.y; in.root_min[2] = arena[0].bounds.a.z;
      in.root_max[0] = arena[0].bounds.b.x; in.root_max[1] = arena[0].bounds.b.y; in.root_max[2] = arena[0].bounds.b.z;
      in.seed_origin = seed_origin;
      in.max_depth = 255u;
      in.device = device_ordinal;
      in.stream = stream;
      in.layout_node_cost = seed_origin ? 0.0f : leaf_node_cost;
      in.layout_leaf_max = leaf_max;
      std::vector<uint32_t> perm;
      // (the arena as the host builder would have made it -- or, with `keep`, the tree left on the device and only its
      // root here)
      if (!device_builder(in, perm, arena, build_err, keep)) return false;
      lap("  gpu build (total)");
      std::vector<uint32_t> permuted(n);
      parallel_chunks(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) permuted[i] = order[perm[i]];
      });
      order.swap(permuted);
    } else if (n > 0) {
      bb.split(arena, 0, 0);
    }
    lap("  host: apply order, arena");
    if (!quiet) blas_root[m] = arena[0].bounds;  // (the same box for both trees of a BLAS: written by one of them)
    nodes.swap(arena);
    return true;
  };
  for (size_t m = 0; m < blas.size(); ++m) {
    std::string e1, e2;
    Soa& soa = soas[m];   // (filled by the flatten pass for the BLASes the device builds)
    if (builder == RAYCA_BUILDER_SAH) {
      // two independent trees over the same primitives -- the reference's (tie order + candidate filter) and the one that
      // is traversed -- built side by side: one's host phases (SoA, order, layout) run under the other's GPU levels
      ref_prims[m] = blas[m].prims;
      bool ok_ref = false;
      BlasDeviceTree kept;
      std::thread ref_thread([&] { ok_ref = build_blas(m, true, ref_prims[m], ref_nodes[m], e1, true, soa, nullptr, bstream[0]); });
      const bool ok = build_blas(m, false, blas[m].prims, blas[m].nodes, e2, false, soa, keep_trees ? &kept : nullptr, bstream[1]);
      if (kept.handle) {
        s.dev_trees.push_back(kept.handle);
        blas[m].dev_tree = kept.handle;
        blas[m].dev_node_count = kept.node_count;
        blas[m].dev_need = kept.need;
      }
      ref_thread.join();
      lap("reference tree + SAH tree");
      if (!ok_ref || !ok) { err = !ok_ref ? e1 : e2; return RAYCA_ERR_HIP; }
    } else {
      BlasDeviceTree kept;
      const bool ok = build_blas(m, true, blas[m].prims, blas[m].nodes, e1, false, soa, keep_trees ? &kept : nullptr, bstream[1]);
      if (kept.handle) {
        s.dev_trees.push_back(kept.handle);
        blas[m].dev_tree = kept.handle;
        blas[m].dev_node_count = kept.node_count;
        blas[m].dev_need = kept.need;
      }
      if (!ok) { err = e1; return RAYCA_ERR_HIP; }
    }
  }
  std::vector<uint32_t> blas_order(blas.size());
  for (size_t i = 0; i < blas.size(); ++i) blas_order[i] = (uint32_t)i;
  std::vector<TNode> tn(1);
  if (!blas.empty()) tlas_split(tn, 0, blas_order, blas_root, 0, (uint32_t)blas.size());
  // store BLASes in TLAS order so that a TLAS range [offset, offset+count) indexes s.blas directly
  s.blas.clear();
  std::vector<uint32_t> ref_rank;       // flatten index -> slot in the reference's global order
  std::vector<uint32_t> ref_leaf_flat;  // flatten index -> reference leaf
  s.ref_leaf_boxes.clear();
  if (builder == RAYCA_BUILDER_SAH) {
    ref_rank.assign(s.prims.size(), 0);
    ref_leaf_flat.assign(s.prims.size(), 0);
    uint32_t slot = 0;
    for (uint32_t b : blas_order) {
      for (uint32_t pi : ref_prims[b]) ref_rank[pi] = slot++;
      for (const BuildNode& rn : ref_nodes[b]) {
        if (rn.left >= 0 || rn.count == 0) continue;  // inner node (or the root of an empty BLAS)
        const uint32_t leaf = (uint32_t)(s.ref_leaf_boxes.size() / 8);
        const float bx[8] = {rn.bounds.a.x, rn.bounds.a.y, rn.bounds.a.z, 0.0f, rn.bounds.b.x, rn.bounds.b.y, rn.bounds.b.z, 0.0f};
        s.ref_leaf_boxes.insert(s.ref_leaf_boxes.end(), bx, bx + 8);
        for (uint32_t i = rn.offset; i < rn.offset + rn.count; ++i) ref_leaf_flat[ref_prims[b][i]] = leaf;
      }
    }
  }
  for (uint32_t b : blas_order) s.blas.push_back(std::move(blas[b]));

  lap("TLAS + reference tables");
  // ---- device layout ----------------------------------------------------------------------------
  s.dev_nodes.clear();
  s.prim_order.clear();
  s.max_depth = 0;
  if (!s.blas.empty()) {
    DevBuilder db{s, {}};
    if (builder == RAYCA_BUILDER_SAH) {
      // This tree's boxes only have to be conservative (the reference-leaf filter decides candidacy):
      // pad them by 2^-16 of their magnitude plus 2^-20 of the scene diagonal, two orders of magnitude
      // above the rounding of the slab and triangle arithmetic, so that a triangle the ray hits is
      // never lost because an enclosing box rounds the other way.
      const F4 ext = as_vec(tn[0].bounds.b - tn[0].bounds.a);
      const float diag = sqrtf(ext.x * ext.x + ext.y * ext.y + ext.z * ext.z);
      db.pad_rel = 1.52587890625e-05f;
      db.pad_abs = (diag == diag && diag < FLT_MAX) ? diag * 9.5367431640625e-07f : 0.0f;
    }
    s.pad_rel = db.pad_rel;
    s.pad_abs = db.pad_abs;
    db.node_cost = builder == RAYCA_BUILDER_SAH ? leaf_node_cost : 0.0f;
    db.leaf_max = leaf_max;
    uint32_t base = 0;
    for (const HostBlas& bl : s.blas) {
      db.blas_base.push_back(base);
      for (uint32_t pi : bl.prims) s.prim_order.push_back(pi);
      base += (uint32_t)bl.prims.size();
    }
    if (base > kLeafFirstMask) { err = "too many primitives for the packed leaf reference (max 33554431)"; return RAYCA_ERR_UNSUPPORTED; }
    if (hooks.on_order_ready) hooks.on_order_ready();
    s.root_min = tn[0].bounds.a;
    s.root_max = tn[0].bounds.b;
    uint32_t need = 0;
    s.root_ref = db.emit_tlas(tn, 0, need);
    s.max_depth = need;
    lap("  binary nodes");
    s.tie_rank.clear();
    s.ref_leaf_of.clear();
    if (builder == RAYCA_BUILDER_SAH) {
      s.tie_rank.resize(s.prim_order.size());
      s.ref_leaf_of.resize(s.prim_order.size());
      parallel_chunks(s.prim_order.size(), [&](size_t b, size_t e) {
        for (size_t slot = b; slot < e; ++slot) {
          s.tie_rank[slot] = ref_rank[s.prim_order[slot]];
          s.ref_leaf_of[slot] = ref_leaf_flat[s.prim_order[slot]];
        }
      });
    }
  }
  s.dev_nodes4.clear();
  s.dev_nodes_h.clear();
  s.dev_nodes4_h.clear();
  s.root_ref4 = 0;
  s.max_depth4 = 0;
  // the other three node formats are only ever traversed next to the reference-leaf filter
  s.other_formats_wanted = builder == RAYCA_BUILDER_SAH && !s.blas.empty();
  if (hooks.with_formats) finish_node_formats(s);
  lap("device node layouts");
  return RAYCA_OK;
}

}  // namespace rayca
