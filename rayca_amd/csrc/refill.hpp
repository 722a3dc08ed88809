// refill.hpp -- host-side entry of the lane-refill camera-ray kernel (refill.hip), called from api.inc.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace rayca {

struct RefillFlavour {
  bool sph, wide, spill, stats, half;
};
// host stub of the instantiation (for hipFuncGetAttributes) and its launch
const void* flat_refill_kernel(const RefillFlavour& f);
void launch_flat_refill(const RefillFlavour& f, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const FrameParams& fp, uint32_t* heads,
                        uint8_t* rgba8, float4* rgba32f, TraceCounters* counters, const TraceLaunch& tl);

// the same for the rays of a queue (closest hits of a bounce generation of the wavefront engine; 4-wide fp16 nodes)
const void* queue_refill_kernel(bool sph, bool stats);
void launch_queue_refill(bool sph, bool stats, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const QueuedRay* in_rays,
                         const uint32_t* in_count, float4* hits, uint32_t* heads, TraceCounters* counters, const TraceLaunch& tl);

// and for the shadow-ray pass of a generation (k_wf_shadow's work)
struct ShadowRefillArgs {
  float4* sh_ray;
  float4* sh_x;
  uint32_t nls;
  float4* direct;
  uint32_t* state;
  uint32_t npix;
};
const void* shadow_refill_kernel(bool gen0, bool sph, bool stats);
void launch_shadow_refill(bool gen0, bool sph, bool stats, uint32_t grid, size_t lds_bytes, hipStream_t stream, const DevScene& sc, const FrameParams& fp,
                          const QueuedRay* in_rays, const uint32_t* in_count, const ShadowRefillArgs& a, uint32_t depth, uint32_t* heads,
                          TraceCounters* counters, const TraceLaunch& tl);

}  // namespace rayca
