// device_types.hpp -- PODs shared by the host launcher and the gfx950 kernels.
#pragma once

#include <cstdint>

#include "host_scene.hpp"
#include "rayca_math.hpp"

namespace rayca {

struct DevMaterial {  // RaycaMaterial, reordered for 16-B loads
  float color[4];
  float ambient[4], emission[4], diffuse[4], specular[4];
  uint32_t kind, albedo_texture, normal_texture, metallic_roughness_texture;
  float metallic_factor, roughness_factor, shininess;
  uint32_t emissive;  // precomputed PhongMaterial::is_emissive  (material/phong.rs:54-56)
};

struct DevTexture {
  uint32_t width, height, color_type, pad;
  uint64_t byte_offset;
};

// Sphere primitive (rayca-geometry/src/sphere.rs:38-44) with everything its tests and its normal need,
// all derived on the host with the reference's operation order.
struct DevSphere {
  float center[4];        // model-space Point3
  float radius2, pad0, pad1, pad2;
  float translation[4], rotation[4], scale[4];  // WORLD Trs of the node (primitive.rs:95-101)
  float inv_rotation[4], inv_scale[4];           // Inversed<&Trs>: conj(R), reciprocal(S)  (trs.rs:320-331)
  float inv_mat4[16];     // Mat4::from(&Inversed<Trs>) row-major (trs.rs:372-381): world point -> model point
  float normal_mat[12];   // rows of transpose(Mat3::from(&inverse)) padded to 4 (primitive.rs:183-190)
};

struct DevLight {
  uint32_t kind, material;
  float intensity, pad;
  float color[4];
  float attenuation[4];
  float ab[4], ac[4];
  float position[4];  // Point3::from(trs.get_translation()) of the node-LOCAL trs (nee.rs:133-134)
  float normal[4];    // quad: normalized(ab x ac)  (light/quad.rs:36-38)
  float area, pad1, pad2, pad3;
  float direction[4];  // directional: -(rotation * X)  (light/directional.rs:47-51)
  float trs_translation[4], trs_rotation[4], trs_scale[4];  // node-LOCAL Trs (QuadLight::intersects, quad.rs:138-159)
};

struct DevScene {
  const float4* nodes;  // 4 x float4 per DevNode (binary tree)
  const float4* nodes_ch;  // RAYCA_BUILDER_SAH: the same nodes with every box as (centre xyz, half extent xyz): what the conservative binary f32 steps read (trace_core.inc slab_ch)
  const float4* nodes4; // 8 x float4 per DevNode4 (the same tree, 4-wide)
  const uint4* nodes_h;  // 2 x uint4 per DevNodeH, 4 x uint4 per DevNode4H: the same trees with fp16 boxes (or nullptr)
  const uint4* nodes4_h;
  float half_center[3], half_inv_scale;
  const float4* tris;   // 3 x float4 per primitive slot: world-space v0, v1, v2, tie rank, reference leaf (trace_core.inc TriRecord)
  const PrimExt* ext;   // per primitive slot
  // the tables the primitive records' filter fields are filled from (k_fill_tri_filter); the kernels read the records
  const uint32_t* tie_rank;  // nullptr: ties go to the lower slot; else to the lower rank (RAYCA_BUILDER_SAH)
  const uint32_t* ref_leaf_of;    // RAYCA_BUILDER_SAH: slot -> reference leaf (nullptr otherwise: no filter)
  const float4* ref_leaf_boxes;   // 2 x float4 per reference leaf: (min xyz, -), (max xyz, -)
  const DevMaterial* materials;
  const DevLight* lights;
  const DevSphere* spheres;
  const DevTexture* textures;
  const uint8_t* image_bytes;
  uint32_t material_count, light_count, texture_count, prim_count;
  uint32_t root_ref, root_ref4;
  uint32_t root_ref_ch;   // the root as nodes_ch's records reference each other (trace_core.inc kChRefScale)
  float root_min[3], root_max[3];
  float cull_abs;  // absolute slack of the best-t cull (see trace())
  uint32_t reserved;
};

// One launch renders `rows` packed rows of a width x height frame.
struct FrameParams {
  uint32_t width, height;       // full frame
  uint32_t rows;                // rows rendered by this call (packed output)
  uint32_t part, parts, band;   // row r of the output is frame row ((r/band)*parts+part)*band + r%band
  uint32_t tiles_x, tile_count; // 8x8 pixel tiles over the packed output
  float inv_width, inv_height, aspect, angle;
  Trs camera;                   // camera WORLD transform (scene.rs:112)
  // Config
  uint32_t integrator, direct_sampler, indirect_sampler, light_samples, light_stratify, strate_count;
  uint32_t max_depth, russian_roulette, seed, spp, sample;
  float sub_step_x, sub_step_y; // ix*step, iy*step of this sample; the kernel evaluates (x + ix*step) + offset, the reference's
  float sub_offset;             // association (scene.rs:125-137): folding the two terms on the host differs by 1 ulp for spp = 2, 3, 5, 9 ...
  float inv_gamma;
};

// per-launch traversal workspace: the node stack lives in LDS up to `lds_entries` entries per lane
// (entry-major, one 1-KiB row per entry and block) and spills to `ovf` beyond that
// one wave = one tile of 64 pixels: 2^W_LOG2 wide, 2^(6 - W_LOG2) high
#ifndef RAYCA_TILE_W_LOG2
#define RAYCA_TILE_W_LOG2 3
#endif
constexpr uint32_t kTileW = 1u << RAYCA_TILE_W_LOG2, kTileH = 64u >> RAYCA_TILE_W_LOG2;

struct TraceLaunch {
  uint32_t* ovf;         // [entries over the LDS part][ovf_stride] dwords, or nullptr
  uint32_t lds_entries;
  uint32_t ovf_stride;   // total threads of the launch
  uint32_t ticket;       // batches a wave takes from its work counter at a time (persistent kernels)
};

struct TraceCounters {
  unsigned long long boxes, tris, shaded, shadow, bounce, box_slots, tri_slots;
  unsigned long long flags;  // k_general: kFlagUnsupported | kFlagTooDeep
};

// ray queue entry between two generations (32 B)
struct QueuedRay {
  float ox, oy, oz, dx, dy, dz;
  uint32_t pixel;  // index into the packed output (r*width + x)
  uint32_t key;    // RNG key of the path vertex this ray leads to
};

// per generation, per pixel record used to fold the path back in the reference's evaluation order
enum : uint32_t { kVertexNone = 0, kVertexLit = 1, kVertexEmissive = 2, kVertexLitNoIndirect = 3 };

struct WorkQueue {
  uint32_t* heads;   // 8 counters, one per XCD partition (device memory, zeroed before launch)
  uint32_t total;    // number of 64-wide batches
};

}  // namespace rayca
