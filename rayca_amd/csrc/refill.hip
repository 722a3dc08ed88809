// refill.hip -- camera rays of a Flat frame with LANE REFILL (gfx950).
//
// k_generation gives a wave 64 rays and keeps it until the slowest of them has finished.  Where the length of a ray's
// walk through the tree varies a lot inside a tile -- the 1 M-triangle soup: free paths through a random cloud are
// exponentially distributed, so the longest of 64 is about 4.7 times the mean -- most lanes of a wave idle most of the
// time (measured: 17 % lane utilisation in the node loop, 21.5 % of VALU thread-cycles active).  Here a wave is a
// persistent pool of 64 lanes: when no more than RAYCA_REFILL_THRESHOLD of them are still traversing, the finished
// ones shade and store their pixel (Flat::trace, integrator/flat.rs:16-28; scene.rs:146-148) and take new pixels from
// the wave's current 8x8 tile (work tickets as in k_generation: one counter per XCD), while the long rays keep their
// place.  Same arithmetic, same node_step / test_leaf / shade_hit as every other kernel (trace_core.inc): the frame is
// bit-identical to k_generation's (tests/test_gpu_engines.py); which of the two renders a scene's Flat frames is decided
// by timing both on that scene (RaycaScene::Tune in api.inc).
//
// Coherent scenes lose with it (the atrium's camera rays: new rays at the root next to old rays deep in the tree
// diverge in their node addresses), which is why it is a per-scene choice and not the default.
#include <hip/hip_runtime.h>

#include "device_types.hpp"
#include "refill.hpp"

namespace rayca {
namespace {

#include "trace_core.inc"

#ifndef RAYCA_REFILL_THRESHOLD
#define RAYCA_REFILL_THRESHOLD 44
#endif
#ifndef RAYCA_REFILL_SCHED
#define RAYCA_REFILL_SCHED 0
#endif
#ifndef RAYCA_REFILL_LEAVE_K
#define RAYCA_REFILL_LEAVE_K 32
#endif
#ifndef RAYCA_REFILL_WAVES
#define RAYCA_REFILL_WAVES RAYCA_MIN_WAVES_FLAT
#endif
#ifndef RAYCA_REFILL_NODE_W
#define RAYCA_REFILL_NODE_W 1
#endif
#ifndef RAYCA_REFILL_LEAF_W
#define RAYCA_REFILL_LEAF_W 1
#endif

template <bool SPH, bool WIDE, bool SPILL, bool STATS, bool HALF>
__global__ __launch_bounds__(kBlock, RAYCA_REFILL_WAVES) void k_flat_refill(DevScene sc, FrameParams fp, uint32_t* heads, uint8_t* rgba8, float4* rgba32f,
                                                                             TraceCounters* counters, TraceLaunch tl) {
  extern __shared__ uint32_t lds_stack[];
  NodeStack<SPILL> stack = make_stack<SPILL>(lds_stack, tl, blockIdx.x * kBlock + threadIdx.x);
  const uint32_t lane = __lane_id();
  const uint32_t home = xcc_id();
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  WorkCursor wc;
  LaneCounters cnt;
  uint32_t n_shaded = 0;
  // lane state: a ray that is still traversing (cur != kTerminated), a finished ray waiting to be retired (has), or nothing
  bool has = false;
  uint32_t cur = kTerminated, p = 0;
  DRay ray{};
  FastRay fr{};
  DHit hit{};
  float limit = INFINITY;
  // wave state: pixels [pool_next, pool_end) of the tile list (64 per tile, in tile order) not handed out yet
  uint32_t pool_next = 0, pool_end = 0;
  bool dry = false;  // the work counters are exhausted

  for (;;) {
    const uint32_t n_active = (uint32_t)__popcll(__ballot(cur != kTerminated));
    if (n_active <= (dry ? 0u : (uint32_t)RAYCA_REFILL_THRESHOLD)) {
      if (has && cur == kTerminated) {  // retire: Flat::trace + the pixel store
        Color sum = black() + black();  // unwrap_or(BLACK), color += it
        if (hit.prim != RAYCA_NONE) {
          n_shaded++;
          Color color;
          bool emissive;
          ShadeCtx unused;
          shade_hit<SPH>(sc, ray, hit, false, color, emissive, unused);
          sum = black() + color;
        }
        finalize_pixel(fp, sum, p, rgba8, rgba32f);
        has = false;
      }
      if (dry && n_active == 0u) break;  // every lane reaches this: n_active and dry are wave-uniform
      while (!dry) {
        const unsigned long long idle = __ballot(!has);
        if (idle == 0ull) break;
        if (pool_next == pool_end) {
          const uint32_t batch = next_batch(heads, fp.tile_count, home, wc, tl.ticket);
          if (batch == RAYCA_NONE) {
            dry = true;
            break;
          }
          pool_next = batch * 64u;
          pool_end = pool_next + 64u;
        }
        const uint32_t avail = pool_end - pool_next;
        const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
        if (!has && rank < avail) {
          const uint32_t idx = pool_next + rank, batch = idx >> 6, l = idx & 63u;
          const uint32_t ty = batch / fp.tiles_x, tx = batch - ty * fp.tiles_x;
          const uint32_t x = tx * kTileW + (l & (kTileW - 1u)), r = ty * kTileH + (l >> RAYCA_TILE_W_LOG2);
          if (x < fp.width && r < fp.rows) {  // (pixels of a ragged edge tile outside the frame are skipped: the lane asks again)
            const uint32_t y = ((r / fp.band) * fp.parts + fp.part) * fp.band + (r % fp.band);
            p = r * fp.width + x;
            ray = camera_ray(fp, x, y);
            // prologue of trace(): Tlas::intersects tests the root box first (blas.rs:136-139)
            fr = make_fast(sc, ray, HALF);
            hit.t = INFINITY;
            hit.prim = RAYCA_NONE;
            hit.u = hit.v = 0.0f;
            limit = INFINITY;
            stack.clear();
            float tmin;
            if (STATS) cnt.boxes++;
            cur = slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ray, tmin)
                      ? (WIDE ? sc.root_ref4 : (!HALF && RAYCA_NODE_CH ? sc.root_ref_ch : sc.root_ref))
                      : kTerminated;
            has = true;
          }
        }
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        pool_next += n_idle < avail ? n_idle : avail;
      }
    }
    // One trip of the traversal for the lanes that hold a ray: EITHER a node step for the lanes that are searching (cur is an
    // inner node) OR the leaf test for the lanes that hold a leaf -- whichever side has more lanes (weighted:
    // RAYCA_REFILL_NODE_W searching lanes count like RAYCA_REFILL_LEAF_W leaf lanes).  trace()'s while-while form keeps
    // stepping until the slowest search of the wave has found its leaf; with refilled lanes descending from the root next
    // to lanes deep in the tree that idles most of the wave (measured on the soup: 0.29 lane utilisation in the node loop,
    // 9.5 ms per frame; with the majority rule 0.6-0.7 in both loops).  Every lane still performs its own steps in its own
    // order: only the interleaving of lanes changes, not a single result.
#if RAYCA_REFILL_SCHED == 1
    const bool searching = !(cur & kLeafFlag) && cur != kTerminated;
    const bool at_leaf = (cur & kLeafFlag) != 0u;
    const uint32_t n_search = (uint32_t)__popcll(__ballot(searching)), n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
    if (n_search * (uint32_t)RAYCA_REFILL_NODE_W >= n_leaf * (uint32_t)RAYCA_REFILL_LEAF_W && n_search != 0u) {
      if (searching) cur = node_step<true, true, WIDE, SPILL, STATS, HALF>(sc, ray, fr, limit, cur, stack, cnt);
    } else if (at_leaf) {
      test_leaf<true, SPH, STATS>(sc, ray, cur, FLT_MAX, hit, limit, cnt);
      cur = stack.pop();
    }
#else
    // RAYCA_REFILL_SCHED 0: node phase until fewer than RAYCA_REFILL_LEAVE_K lanes are still searching (and some lane holds
    // a leaf), then the leaf phase for every lane that holds one
    for (;;) {
      const bool searching = !(cur & kLeafFlag) && cur != kTerminated;
      const uint32_t n = (uint32_t)__popcll(__ballot(searching));
      if (n == 0u) break;
      if (n < (uint32_t)RAYCA_REFILL_LEAVE_K && __ballot((cur & kLeafFlag) != 0u) != 0ull) break;
      if (searching) cur = node_step<true, true, WIDE, SPILL, STATS, HALF>(sc, ray, fr, limit, cur, stack, cnt);
    }
    if (cur & kLeafFlag) {
      test_leaf<true, SPH, STATS>(sc, ray, cur, FLT_MAX, hit, limit, cnt);
      cur = stack.pop();
    }
#endif
  }
  if (STATS) {
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
}


// ---- queued rays (the wavefront engine's bounce generations) with lane refill ------------------------------------------
// The same persistent-lane scheme for the rays of an input queue: bounce rays start from scattered points in scattered
// directions, their searches differ in length far more than camera rays' do, and with one ray per lane for the lifetime of
// a wave (k_wf_trace) most lanes of a wave sit out most of its trips (measured on the atrium, 4 bounces: 0.47 of the lanes
// in the node loop, 0.23 in the leaf loop).  Here a lane that has finished its ray writes the hit record and takes the next
// queue entry.  Items are dealt in batches of 64 consecutive entries through the same per-XCD work counters; closest hits
// only (4-wide fp16 nodes, RAYCA_WF_BOUNCE_WIDE / _HALF, as k_wf_trace traverses them), every ray still takes its own
// steps in its own order: the hit records are the ones k_wf_trace writes.
// Measured (atrium 1080p, default 5-deep Config, four frames in flight, tests/gpu_ab_inflight.py; profiles/r02_ab_qrefill.log):
// k_wf_trace 4.995 ms per frame, this kernel with (threshold, K) = (44, 32) 4.792, (32, 16) 4.744, (48, 8) 4.736,
// (24, 8) 4.755, (56, 24) 4.786, (48, 48) 5.29; the 3840x2160 4-bounce frame 19.18 -> 18.18 ms; lanes active in the
// node loop over the whole frame 0.47 -> 0.56.
#ifndef RAYCA_QREFILL_THRESHOLD
#define RAYCA_QREFILL_THRESHOLD 48   // refill when at most this many lanes still hold a live ray
#endif
#ifndef RAYCA_QREFILL_LEAVE_K
#define RAYCA_QREFILL_LEAVE_K 8      // leave the node phase when fewer lanes than this are searching (and a lane holds a leaf)
#endif
#ifndef RAYCA_QREFILL_WAVES
#define RAYCA_QREFILL_WAVES RAYCA_REFILL_WAVES
#endif
template <bool SPH, bool STATS>
__global__ __launch_bounds__(kBlock, RAYCA_QREFILL_WAVES) void k_queue_refill(DevScene sc, const QueuedRay* in_rays, const uint32_t* in_count, float4* hits,
                                                                              uint32_t* heads, TraceCounters* counters, TraceLaunch tl) {
  constexpr bool WIDE = RAYCA_WF_BOUNCE_WIDE != 0, SPILL = true, HALF = RAYCA_WF_BOUNCE_HALF != 0;   // (as k_wf_trace traverses bounce rays)
  extern __shared__ uint32_t lds_stack[];
  NodeStack<SPILL> stack = make_stack<SPILL>(lds_stack, tl, blockIdx.x * kBlock + threadIdx.x);
  const uint32_t lane = __lane_id();
  const uint32_t home = xcc_id();
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  const uint32_t count = *in_count, n_batches = (count + 63u) >> 6;
  WorkCursor wc;
  LaneCounters cnt;
  bool has = false;
  uint32_t cur = kTerminated, item = 0;
  DRay ray{};
  FastRay fr{};
  DHit hit{};
  float limit = INFINITY;
  uint32_t pool_next = 0, pool_end = 0;
  bool dry = false;
  for (;;) {
    const uint32_t n_active = (uint32_t)__popcll(__ballot(cur != kTerminated));
    if (n_active <= (dry ? 0u : (uint32_t)RAYCA_QREFILL_THRESHOLD)) {
      if (has && cur == kTerminated) {  // retire: the hit record of this queue entry
        hits[item] = make_float4(hit.t, __uint_as_float(hit.prim), hit.u, hit.v);
        has = false;
      }
      if (dry && n_active == 0u) break;  // every lane reaches this: n_active and dry are wave-uniform
      while (!dry) {
        const unsigned long long idle = __ballot(!has);
        if (idle == 0ull) break;
        if (pool_next == pool_end) {
          const uint32_t batch = next_batch(heads, n_batches, home, wc, tl.ticket);
          if (batch == RAYCA_NONE) {
            dry = true;
            break;
          }
          pool_next = batch * 64u;
          pool_end = min(pool_next + 64u, count);
        }
        const uint32_t avail = pool_end - pool_next;
        const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
        if (!has && rank < avail) {
          item = pool_next + rank;
          const QueuedRay q = in_rays[item];
          ray = make_ray(point3(q.ox, q.oy, q.oz), vec3(q.dx, q.dy, q.dz));
          fr = make_fast(sc, ray, HALF);   // prologue of trace(): the root box first (blas.rs:136-139)
          hit.t = INFINITY;
          hit.prim = RAYCA_NONE;
          hit.u = hit.v = 0.0f;
          limit = INFINITY;
          stack.clear();
          float tmin;
          if (STATS) cnt.boxes++;
          cur = slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ray, tmin) ? (WIDE ? sc.root_ref4 : (!HALF && RAYCA_NODE_CH ? sc.root_ref_ch : sc.root_ref)) : kTerminated;
          has = true;
        }
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        pool_next += n_idle < avail ? n_idle : avail;
      }
    }
    for (;;) {  // (RAYCA_REFILL_SCHED 0 of k_flat_refill)
      const bool searching = !(cur & kLeafFlag) && cur != kTerminated;
      const uint32_t n = (uint32_t)__popcll(__ballot(searching));
      if (n == 0u) break;
      if (n < (uint32_t)RAYCA_QREFILL_LEAVE_K && __ballot((cur & kLeafFlag) != 0u) != 0ull) break;
      if (searching) cur = node_step<true, true, WIDE, SPILL, STATS, HALF>(sc, ray, fr, limit, cur, stack, cnt);
    }
    if (cur & kLeafFlag) {
      test_leaf<true, SPH, STATS>(sc, ray, cur, FLT_MAX, hit, limit, cnt);
      cur = stack.pop();
    }
  }
  if (STATS) {
    unsigned long long b = cnt.boxes, t = cnt.tris, sb = cnt.slot_boxes, stt = cnt.slot_tris;
    for (int off = 32; off > 0; off >>= 1) {
      b += __shfl_down(b, off);
      t += __shfl_down(t, off);
      sb += __shfl_down(sb, off);
      stt += __shfl_down(stt, off);
    }
    if (lane == 0) {
      atomicAdd(&counters->boxes, b);
      atomicAdd(&counters->tris, t);
      atomicAdd(&counters->box_slots, sb);
      atomicAdd(&counters->tri_slots, stt);
    }
  }
}


// The shadow-ray pass the same way (k_wf_shadow: a lane owns a pixel, traces its shadow rays in sample order and sums the
// direct light exactly like the NEE loop of k_generation).  A shadow ray ends at the first occluder, so the searches of a
// wave differ in length even more than closest-hit searches do; and the items that have no lit vertex never occupy a lane.
// A lane whose ray is done adds its sample, then takes the pixel's next shadow ray or -- after the last -- writes the sum and
// takes the next item.  Items: the tile list for generation 0, the input queue after; the other set of work counters
// (heads_b), since the closest-hit pass of the same generation has used the first.
template <bool GEN0, bool SPH, bool STATS>
__global__ __launch_bounds__(kBlock, RAYCA_QREFILL_WAVES) void k_shadow_refill(DevScene sc, FrameParams fp, const QueuedRay* in_rays, const uint32_t* in_count, WfBuffers wb,
                                                                                PathBuffers pb, uint32_t depth, uint32_t* heads, TraceCounters* counters, TraceLaunch tl) {
  constexpr bool WIDE = RAYCA_WF_SHADOW_WIDE != 0, SPILL = true, HALF = RAYCA_WF_SHADOW_HALF != 0;   // (as k_wf_shadow traverses them)
  extern __shared__ uint32_t lds_stack[];
  NodeStack<SPILL> stack = make_stack<SPILL>(lds_stack, tl, blockIdx.x * kBlock + threadIdx.x);
  const uint32_t lane = __lane_id();
  const uint32_t home = xcc_id();
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  const uint32_t count = GEN0 ? fp.tile_count * 64u : *in_count, n_batches = (count + 63u) >> 6;
  WorkCursor wc;
  LaneCounters cnt;
  bool has = false;          // the lane holds a pixel whose shadow rays are not all done
  uint32_t cur = kTerminated, p = 0, j = 0, quad = 0;
  float t_stop = FLT_MAX;
  Color direct = black();
  DRay ray{};
  FastRay fr{};
  DHit hit{};
  float limit = INFINITY;
  uint32_t pool_next = 0, pool_end = 0;
  bool dry = false;
  // the j-th shadow ray of pixel p: prologue of trace() with the any-hit bound
  auto start_ray = [&]() {
    const size_t e = (size_t)j * pb.npix + p;
    const float4 a = wb.sh_ray[2 * e], b = wb.sh_ray[2 * e + 1];
    quad = __float_as_uint(b.w);
    t_stop = a.w;
    ray = make_ray(point3(a.x, a.y, a.z), vec3(b.x, b.y, b.z));
    fr = make_fast(sc, ray, HALF);
    hit.t = INFINITY;
    hit.prim = RAYCA_NONE;
    hit.u = hit.v = 0.0f;
    limit = t_stop < FLT_MAX ? t_stop + fabsf(t_stop) * 9.765625e-4f + sc.cull_abs : INFINITY;
    stack.clear();
    float tmin;
    if (STATS) cnt.boxes++;
    cur = slab(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1], sc.root_max[2], ray, tmin) ? (WIDE ? sc.root_ref4 : (!HALF && RAYCA_NODE_CH ? sc.root_ref_ch : sc.root_ref)) : kTerminated;
  };
  for (;;) {
    const uint32_t n_active = (uint32_t)__popcll(__ballot(cur != kTerminated));
    if (n_active <= (dry ? 0u : (uint32_t)RAYCA_QREFILL_THRESHOLD)) {
      if (has && cur == kTerminated) {  // retire a shadow ray: k_wf_shadow's statements
        const bool found = hit.prim != RAYCA_NONE;
        bool lit;
        if (quad) {
          lit = false;
          if (found) {
            const uint32_t hm = sc.ext[hit.prim].material;
            lit = hm != RAYCA_NONE && hm < sc.material_count && sc.materials[hm].emissive != 0u;
          }
        } else {
          lit = !(found && hit.t < t_stop);
        }
        const size_t e = (size_t)j * pb.npix + p;
        direct = direct + (lit ? as_color(wb.sh_x[e]) : black());
        if (++j == wb.nls) {
          pb.direct[(size_t)depth * pb.npix + p] = as_f4(direct);
          has = false;
        } else {
          start_ray();
        }
      }
      // (every lane reaches this: both operands are wave-uniform.  A lane that still holds a pixel -- its next ray started by
      // the retire above, or ended at the root box at once -- keeps the wave going; it retires on a later trip)
      if (dry && __ballot(has) == 0ull) break;
      while (!dry) {
        const unsigned long long idle = __ballot(!has);
        if (idle == 0ull) break;
        if (pool_next == pool_end) {
          const uint32_t batch = next_batch(heads, n_batches, home, wc, tl.ticket);
          if (batch == RAYCA_NONE) {
            dry = true;
            break;
          }
          pool_next = batch * 64u;
          pool_end = min(pool_next + 64u, count);
        }
        const uint32_t avail = pool_end - pool_next;
        const uint32_t rank = (uint32_t)__popcll(idle & lanes_below);
        if (!has && rank < avail) {
          const uint32_t item = pool_next + rank;
          uint32_t px = RAYCA_NONE;
          if (GEN0) {
            uint32_t x = 0, r = 0;
            if (tile_item_pixel(fp, item, x, r)) px = r * fp.width + x;
          } else {
            px = in_rays[item].pixel;
          }
          if (px != RAYCA_NONE) {  // (items without a lit vertex are skipped: the lane asks again)
            const uint32_t st = pb.state[(size_t)depth * pb.npix + px];
            if (st == kVertexLit || st == kVertexLitNoIndirect) {
              p = px;
              j = 0;
              direct = black();
              has = true;
              start_ray();
            }
          }
        }
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        pool_next += n_idle < avail ? n_idle : avail;
      }
    }
    for (;;) {
      const bool searching = !(cur & kLeafFlag) && cur != kTerminated;
      const uint32_t n = (uint32_t)__popcll(__ballot(searching));
      if (n == 0u) break;
      if (n < (uint32_t)RAYCA_QREFILL_LEAVE_K && __ballot((cur & kLeafFlag) != 0u) != 0ull) break;
      if (searching) cur = node_step<true, true, WIDE, SPILL, STATS, HALF>(sc, ray, fr, limit, cur, stack, cnt);
    }
    if (cur & kLeafFlag) {
      test_leaf<true, SPH, STATS>(sc, ray, cur, t_stop, hit, limit, cnt);
      cur = (t_stop < FLT_MAX && hit.t < t_stop) ? kTerminated : stack.pop();   // any hit in front of the light ends the search (trace())
    }
  }
  if (STATS) {
    unsigned long long b = cnt.boxes, t = cnt.tris, sb = cnt.slot_boxes, stt = cnt.slot_tris;
    for (int off = 32; off > 0; off >>= 1) {
      b += __shfl_down(b, off);
      t += __shfl_down(t, off);
      sb += __shfl_down(sb, off);
      stt += __shfl_down(stt, off);
    }
    if (lane == 0) {
      atomicAdd(&counters->boxes, b);
      atomicAdd(&counters->tris, t);
      atomicAdd(&counters->box_slots, sb);
      atomicAdd(&counters->tri_slots, stt);
    }
  }
}

using ShadowRefillKernel = void (*)(DevScene, FrameParams, const QueuedRay*, const uint32_t*, WfBuffers, PathBuffers, uint32_t, uint32_t*, TraceCounters*, TraceLaunch);
template <bool GEN0>
ShadowRefillKernel pick_shadow(bool sph, bool stats) {
  if (sph) return stats ? k_shadow_refill<GEN0, true, true> : k_shadow_refill<GEN0, true, false>;
  return stats ? k_shadow_refill<GEN0, false, true> : k_shadow_refill<GEN0, false, false>;
}

using QueueRefillKernel = void (*)(DevScene, const QueuedRay*, const uint32_t*, float4*, uint32_t*, TraceCounters*, TraceLaunch);
QueueRefillKernel pick_queue(bool sph, bool stats) {
  if (sph) return stats ? k_queue_refill<true, true> : k_queue_refill<true, false>;
  return stats ? k_queue_refill<false, true> : k_queue_refill<false, false>;
}

using RefillKernel = void (*)(DevScene, FrameParams, uint32_t*, uint8_t*, float4*, TraceCounters*, TraceLaunch);
template <bool SPH, bool STATS>
RefillKernel pick2(bool wide, bool spill, bool half) {
  if (wide) return half ? k_flat_refill<SPH, true, true, STATS, true> : k_flat_refill<SPH, true, true, STATS, false>;  // 4-wide: always with the spill path
  if (spill) return half ? k_flat_refill<SPH, false, true, STATS, true> : k_flat_refill<SPH, false, true, STATS, false>;
  return half ? k_flat_refill<SPH, false, false, STATS, true> : k_flat_refill<SPH, false, false, STATS, false>;
}
RefillKernel pick(const RefillFlavour& f) {
  if (f.sph) return f.stats ? pick2<true, true>(f.wide, f.spill, f.half) : pick2<true, false>(f.wide, f.spill, f.half);
  return f.stats ? pick2<false, true>(f.wide, f.spill, f.half) : pick2<false, false>(f.wide, f.spill, f.half);
}

}  // namespace

const void* flat_refill_kernel(const RefillFlavour& f) { return reinterpret_cast<const void*>(pick(f)); }

void launch_flat_refill(const RefillFlavour& f, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const FrameParams& fp, uint32_t* heads,
                        uint8_t* rgba8, float4* rgba32f, TraceCounters* counters, const TraceLaunch& tl) {
  hipLaunchKernelGGL(pick(f), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, fp, heads, rgba8, rgba32f, counters, tl);
}

const void* queue_refill_kernel(bool sph, bool stats) { return reinterpret_cast<const void*>(pick_queue(sph, stats)); }

void launch_queue_refill(bool sph, bool stats, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const QueuedRay* in_rays,
                         const uint32_t* in_count, float4* hits, uint32_t* heads, TraceCounters* counters, const TraceLaunch& tl) {
  hipLaunchKernelGGL(pick_queue(sph, stats), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, in_rays, in_count, hits, heads, counters, tl);
}

const void* shadow_refill_kernel(bool gen0, bool sph, bool stats) {
  return reinterpret_cast<const void*>(gen0 ? pick_shadow<true>(sph, stats) : pick_shadow<false>(sph, stats));
}

void launch_shadow_refill(bool gen0, bool sph, bool stats, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const FrameParams& fp,
                          const QueuedRay* in_rays, const uint32_t* in_count, const ShadowRefillArgs& a, uint32_t depth, uint32_t* heads,
                          TraceCounters* counters, const TraceLaunch& tl) {
  WfBuffers wb{};
  wb.sh_ray = a.sh_ray;
  wb.sh_x = a.sh_x;
  wb.nls = a.nls;
  PathBuffers pb{};
  pb.direct = a.direct;
  pb.state = a.state;
  pb.npix = a.npix;
  hipLaunchKernelGGL(gen0 ? pick_shadow<true>(sph, stats) : pick_shadow<false>(sph, stats), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, fp, in_rays, in_count, wb,
                     pb, depth, heads, counters, tl);
}

}  // namespace rayca
