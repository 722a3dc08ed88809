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

#include "trace_core.inc"

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
      constexpr int kLeaveK = !ORDERED ? 0 : (MODE == kModeFlat ? RAYCA_LEAVE_K_CAMERA : (GEN0 ? RAYCA_LEAVE_K_PATH0 : RAYCA_LEAVE_K_BOUNCE));
      const bool found = trace<ORDERED, FAST, SPH, WIDE, SPILL, STATS, HALF, kLeaveK>(sc, ray, t_stop, stack, hit, cnt);
      if (!in_shadow) {
        if (!found) {
          if (FUSED) direct = black() + black();  // unwrap_or(BLACK), color += it   (path frames: k_resolve's `prev + c`, c = BLACK)
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
            if (FUSED) direct = black() + color;  // k_resolve: L = the emissive colour, acc = BLACK + L
            else {
              pb.direct[slot] = as_f4(color);
              pb.state[slot] = kVertexEmissive;
            }
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
    if (FUSED && MODE == kModePath) {
      // One generation, one sample (RAYCA_FUSE_PATH1): no records, no k_resolve -- the vertex's value is folded here the way
      // k_resolve folds a single depth: L = direct + (BLACK + (BLACK [+ 0]) / light_samples), pixel = BLACK + L.
      if (tail) {
        if (PARK) direct = as_color(ctx_slot[6 * kBlock]);
        const uint32_t limit = fp.direct_sampler != RAYCA_SAMPLER_NONE ? fp.max_depth - 1u : fp.max_depth;
        Color indirect = black();
        if (depth < limit) indirect = indirect + black() / (float)fp.light_samples;  // kVertexLit without a child: li stays BLACK
        direct = black() + (direct + indirect);
      }
      n_shadow += (uint32_t)__popcll(__ballot(tail)) * nee_lights * fp.light_samples;
      if (has_pixel) finalize_pixel(fp, direct, p, rgba8, rgba32f);
    } else if (FUSED) {
      if (has_pixel) finalize_pixel(fp, direct, p, rgba8, rgba32f);
    } else if (MODE == kModePath) {
      bool want_bounce = false;
      QueuedRay next{};
      if (tail) {
        // all direct samples done: Pathtracer::trace_impl tail  pathtracer.rs:89-105
        const uint32_t limit = fp.direct_sampler != RAYCA_SAMPLER_NONE ? fp.max_depth - 1u : fp.max_depth;
        const bool hemi = fp.indirect_sampler == RAYCA_SAMPLER_HEMISPHERE;
        F4 sdir = vec3(0.0f, 0.0f, 0.0f);
        if (depth < limit) {
          // CosineSampler::get_random_dir  sampler/cosine.rs:65-88 ; HemisphereSampler  hemisphere.rs:17-40.  The direction in the
          // sampler's own frame only needs the two random numbers: made before the parked context comes back from LDS, so that
          // the double-precision temporaries of libm_exact.hpp (the host libm's bits) and the context are not live together
          const float e1 = rng_f32(key, dim++), e2 = rng_f32(key, dim++);
          const float theta = hemi ? rc_acosf(e1) : rc_acosf(sqrtf(e1));
          const float omega_a = 2.0f * kPi * e2;
          const float st = rc_sinf(theta);
          sdir = vec3(rc_cosf(omega_a) * st, rc_sinf(omega_a) * st, rc_cosf(theta));
        }
        if (PARK) {
          cx = unpark_ctx(ctx_slot);
          direct = as_color(ctx_slot[6 * kBlock]);
        }
        pb.direct[slot] = as_f4(direct);
        if (depth < limit) {
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
// `clear` (optional): `clear_words` dwords zeroed by the first block -- the work counters of this frame context, which the
// generation kernel in front of this one has finished with (same stream), so that the next frame of the context needs no
// memset launch of its own (a frame is three launches then instead of four: what a rank's share of a frame costs on the
// host is what limits an 8-GPU run).
__global__ __launch_bounds__(kBlock) void k_resolve(FrameParams fp, PathBuffers pb, uint32_t depths, float4* accum, uint8_t* rgba8, float4* rgba32f,
                                                    uint32_t* clear, uint32_t clear_words) {
  if (clear && blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < clear_words; i += blockDim.x) clear[i] = 0u;
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
