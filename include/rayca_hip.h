/*
 * rayca_hip.h -- C ABI of librayca_hip.so, the MI355X (gfx950) path-tracing core that sits behind
 * rayca-soft's `Scene` / `Draw::draw()` surface.
 *
 * The reference (Fahien/rayca, Rust) has NO FFI for this path: the interface being replaced is the
 * pure-Rust trait
 *
 *     pub trait Draw { fn draw(&mut self, scene: &Scene, image: &mut Image); }
 *                                                       rayca-soft/src/draw.rs:7-9
 *
 * implemented by `SoftRenderer { pub config: Config }` (rayca-soft/src/scene.rs:11-14,88-154).
 * Everything `draw` does from `SceneDrawInfo::new(scene)` (scene.rs:90) to the RGBA8 store
 * (scene.rs:148) happens behind this ABI; what crosses it is a flat, pointer+size restatement of
 * `&Scene`, `Config` and `&mut Image`:
 *
 *   RaycaSceneDesc  <- rayca-model Scene/Model/Node/Mesh/Primitive/Geometry/Material/Texture/Image/
 *                      Camera/Light                      rayca-model/src/{scene,model,node,...}.rs
 *   RaycaConfig     <- rayca_soft::Config                 rayca-soft/src/config.rs:10-49
 *   rgba8 / rgba32f <- rayca_model::Image (RGBA8, row-major, top-left origin)
 *                                                         rayca-model/src/image.rs:26-36
 *
 * Plain C: fixed-width scalars, pointers and counts only.  No torch / HIP types in signatures
 * (device pointers and streams travel as void*).  Every entry point returns an int32 status
 * (RAYCA_OK == 0, negative on error) and never aborts; `rayca_hip_last_error` returns the
 * thread-local message of the last failure.  The reference panics in the same situations
 * (no camera scene.rs:109, empty TLAS tlas.rs:272, bad index type primitive.rs:258).
 */
#ifndef RAYCA_HIP_H
#define RAYCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAYCA_ABI_VERSION 2u   /* 2: RaycaRenderOptions.wait_event / record_event, RaycaStats.class_ms / class_launches,
                                  RaycaMultiOptions.context, rayca_hip_render_multi_issue / _wait, rayca_hip_scene_reap */
#define RAYCA_NONE 0xFFFFFFFFu /* Handle::NONE, rayca-util/src/pack.rs:61-64 */

/* ---- status codes -------------------------------------------------------------------------- */
enum {
  RAYCA_OK = 0,
  RAYCA_ERR_BAD_ARG = -1,
  RAYCA_ERR_NO_CAMERA = -2,   /* assert!(!camera_draw_infos.is_empty())      scene.rs:109 */
  RAYCA_ERR_EMPTY_SCENE = -3, /* Tlas::intersects assert                      tlas.rs:272  */
  RAYCA_ERR_HIP = -4,
  RAYCA_ERR_OOM = -5,
  RAYCA_ERR_UNSUPPORTED = -6, /* todo!()/unimplemented!() arms of the reference, or a Config the
                                 kernels do not cover yet: fails loudly, never falls back to CPU */
  RAYCA_ERR_NO_DEVICE = -7,
  RAYCA_ERR_BVH_DEPTH = -8,
  RAYCA_ERR_RCCL = -9 /* the frame-end gather of rayca_hip_render_multi: RCCL missing, or one of its calls failed */
};

/* ---- enums crossing the ABI as uint32, in the reference's #[repr(u32)] order ---------------- */
/* rayca-soft/src/integrator/mod.rs:32-41 */
enum {
  RAYCA_INTEGRATOR_SCRATCHER = 0,
  RAYCA_INTEGRATOR_RAYTRACER = 1,
  RAYCA_INTEGRATOR_FLAT = 2,
  RAYCA_INTEGRATOR_ANALYTIC_DIRECT = 3,
  RAYCA_INTEGRATOR_DIRECT = 4,
  RAYCA_INTEGRATOR_PATHTRACER = 5
};
/* rayca-soft/src/sampler/mod.rs:41-50 */
enum {
  RAYCA_SAMPLER_NONE = 0,
  RAYCA_SAMPLER_NEE = 1,
  RAYCA_SAMPLER_HEMISPHERE = 2,
  RAYCA_SAMPLER_COSINE = 3,
  RAYCA_SAMPLER_BRDF = 4,
  RAYCA_SAMPLER_MIS = 5
};
/* rayca-model/src/material/mod.rs:15-20 */
enum { RAYCA_MATERIAL_PBR = 0, RAYCA_MATERIAL_PHONG = 1, RAYCA_MATERIAL_GGX = 2 };
/* rayca-model/src/light/mod.rs:15-19 */
enum { RAYCA_LIGHT_DIRECTIONAL = 0, RAYCA_LIGHT_POINT = 1, RAYCA_LIGHT_QUAD = 2 };
/* rayca-model Geometry enum (TriangleMesh | Sphere) */
enum { RAYCA_GEOMETRY_TRIANGLE_MESH = 0, RAYCA_GEOMETRY_SPHERE = 1 };
/* rayca-geometry/src/triangle.rs:180-201 ComponentType (glTF numbers) */
enum { RAYCA_INDEX_U8 = 5121, RAYCA_INDEX_U16 = 5123, RAYCA_INDEX_U32 = 5125 };
/* rayca-math/src/color/mod.rs:19-25 ColorType */
enum { RAYCA_COLOR_RGB8 = 0, RAYCA_COLOR_RGBA8 = 1, RAYCA_COLOR_RGBA32F = 2 };

/* ---- Config -------------------------------------------------------------------------------- */
/* Field-for-field mirror of rayca_soft::Config (config.rs:10-49); defaults in
 * rayca_hip_config_default().  `seed` is the one addition: the reference draws from a thread-local
 * OS-seeded fastrand (sampler/cosine.rs:66-67), which is not reproducible; here random numbers are
 * a counter-based function of (seed, pixel, sample, depth, dimension). */
typedef struct RaycaConfig {
  uint32_t bvh;               /* bool, default 1.  0 => Tlas max_depth(0): one leaf per model    */
  uint32_t light_samples;     /* default 1 */
  uint32_t light_stratify;    /* bool, default 0 */
  uint32_t samples_per_pixel; /* default 1 */
  uint32_t russian_roulette;  /* bool, default 0 */
  uint32_t direct_sampler;    /* RAYCA_SAMPLER_*, default NEE */
  uint32_t indirect_sampler;  /* RAYCA_SAMPLER_*, default COSINE */
  uint32_t integrator;        /* RAYCA_INTEGRATOR_*, default PATHTRACER */
  uint32_t max_depth;         /* default 5 */
  float gamma;                /* default 1.0 */
  uint32_t seed;              /* counter-based RNG key (no reference counterpart) */
  uint32_t reserved;
} RaycaConfig;

/* ---- scene description --------------------------------------------------------------------- */
/* rayca_math::Trs: translation, rotation quaternion (x,y,z,w), scale.   rayca-math/src/trs.rs:75-86
 * Applied scale -> rotate -> translate (trs.rs:264-273). */
typedef struct RaycaTrs {
  float translation[3];
  float rotation[4];
  float scale[3];
} RaycaTrs;

/* One entry per node of the flattened scene graph: the Scene root, Scene nodes, each Model's root
 * and its nodes (rayca-model/src/node.rs:11-32).  `parent` indexes this same array (-1 for the
 * top).  Parents must precede children.  World transforms are composed inside the library exactly
 * as SceneDrawInfo::traverse_* does (scene.rs:206-282): world(i) = world(parent) * local(i) with
 * the reference's Trs x Trs (trs.rs:211-221); for a top node world = local (scene.rs:207).
 * `model` groups mesh nodes into one BLAS per model (bvh/primitive.rs:385-393); BLASes are created
 * in ascending `model` order (the reference iterates a HashMap, i.e. leaves this unspecified). */
typedef struct RaycaNode {
  int32_t parent;
  uint32_t model;
  uint32_t mesh;   /* index into meshes, or RAYCA_NONE */
  uint32_t camera; /* index into cameras, or RAYCA_NONE */
  uint32_t light;  /* index into lights, or RAYCA_NONE */
  RaycaTrs trs;    /* node-local */
} RaycaNode;

/* rayca_model::Mesh = list of primitives (mesh.rs:32-35): a contiguous range of `primitives`. */
typedef struct RaycaMesh {
  uint32_t first_primitive;
  uint32_t primitive_count;
} RaycaMesh;

/* rayca_model::Primitive { geometry, material } (primitive.rs:9-14) with the Geometry inlined.
 * Triangle mesh: vertices [first_vertex, first_vertex+vertex_count) of the vertex arrays; indices
 * are `index_count` values of `index_type` starting at byte `index_byte_offset` of `index_bytes`
 * (byte-packed like TriangleIndices, rayca-geometry/src/triangle.rs:215-307), relative to
 * first_vertex.  Sphere: model-space center + radius (rayca-geometry/src/sphere.rs:38-44). */
typedef struct RaycaPrimitive {
  uint32_t geometry;  /* RAYCA_GEOMETRY_* */
  uint32_t material;  /* index into materials, or RAYCA_NONE -> Material::DEFAULT (pbr white) */
  uint32_t first_vertex;
  uint32_t vertex_count;
  uint64_t index_byte_offset;
  uint32_t index_count;
  uint32_t index_type; /* RAYCA_INDEX_* */
  float sphere_center[3];
  float sphere_radius;
} RaycaPrimitive;

/* Material (material/mod.rs:15-20) flattened over its three payloads:
 *   Pbr   (material/pbr.rs:58-66): color, albedo/normal/metallic_roughness textures, factors
 *   Phong (material/phong.rs:10-34): ambient, emission, diffuse, specular, shininess
 *   Ggx   (material/ggx.rs:10-23):  diffuse, specular, roughness                      */
typedef struct RaycaMaterial {
  uint32_t kind; /* RAYCA_MATERIAL_* */
  uint32_t albedo_texture;
  uint32_t normal_texture;
  uint32_t metallic_roughness_texture; /* texture indices or RAYCA_NONE */
  float color[4];
  float metallic_factor;
  float roughness_factor;
  float shininess;
  float pad0;
  float ambient[4];
  float emission[4];
  float diffuse[4];
  float specular[4];
} RaycaMaterial;

/* Texture -> image (texture.rs:35-39); sampler is always the default nearest/wrap one
 * (material/pbr.rs:96, sampler.rs:11-30). */
typedef struct RaycaTexture {
  uint32_t image;
} RaycaTexture;

/* Image payload (image.rs:26-36): `color_type` texels, row-major, at `byte_offset` of image_bytes. */
typedef struct RaycaImage {
  uint32_t width;
  uint32_t height;
  uint32_t color_type; /* RAYCA_COLOR_* */
  uint32_t pad0;
  uint64_t byte_offset;
} RaycaImage;

/* Camera: only yfov reaches the hot path (camera.rs:74-76 get_angle). */
typedef struct RaycaCamera {
  float yfov_radians;
} RaycaCamera;

/* Light (light/{point,quad,directional}.rs). */
typedef struct RaycaLight {
  uint32_t kind; /* RAYCA_LIGHT_* */
  uint32_t material; /* quad light material (light/quad.rs:21), or RAYCA_NONE */
  float intensity;
  float pad0;
  float color[4];
  float attenuation[3]; /* point: (const, linear, quadratic), default (0,0,1) point.rs:20 */
  float pad1;
  float ab[3]; /* quad edges, light/quad.rs:17-18 */
  float pad2;
  float ac[3];
  float pad3;
} RaycaLight;

typedef struct RaycaSceneDesc {
  uint32_t abi_version; /* RAYCA_ABI_VERSION */
  uint32_t flags;       /* 0 */

  const RaycaNode* nodes;
  uint32_t node_count;
  const RaycaMesh* meshes;
  uint32_t mesh_count;
  const RaycaPrimitive* primitives;
  uint32_t primitive_count;

  /* vertex attribute arrays (rayca-geometry/src/vertex.rs:137-142), SoA, vertex_count entries.
   * positions is required; any other pointer may be NULL, meaning the Vertex::default() value
   * (color white, normal +Z, tangent/bitangent zero, uv zero  vertex.rs:164-175). */
  uint32_t vertex_count;
  const float* positions;  /* 3 per vertex */
  const float* colors;     /* 4 per vertex */
  const float* normals;    /* 3 per vertex */
  const float* tangents;   /* 3 per vertex */
  const float* bitangents; /* 3 per vertex */
  const float* uvs;        /* 2 per vertex */

  const uint8_t* index_bytes;
  uint64_t index_byte_count;

  const RaycaMaterial* materials;
  uint32_t material_count;
  const RaycaTexture* textures;
  uint32_t texture_count;
  const RaycaImage* images;
  uint32_t image_count;
  const uint8_t* image_bytes;
  uint64_t image_byte_count;

  const RaycaCamera* cameras;
  uint32_t camera_count;
  const RaycaLight* lights;
  uint32_t light_count;
} RaycaSceneDesc;

/* ---- build / render options (no reference counterpart: knobs of this implementation) -------- */
enum {
  /* bit-for-bit restatement of Blas::set_primitives_recursive (bvh/blas.rs:261-316: SAH over 63
   * planes x 3 axes) and TlasNode::replace_models_recursive (bvh/tlas.rs:74-134), evaluated with an
   * exact binned sweep instead of 189 passes over the primitives */
  RAYCA_BUILDER_REFERENCE = 0,
  /* the same builder with ONE change: the candidate boxes of evaluate_sah start empty instead of at
   * AABB::default() (= the origin, bvh/blas.rs:66-67 + bvh/aabb.rs:9-13).  The reference's seed
   * makes every off-origin cluster unsplittable (leaves of 10^2..10^4 triangles on Sponza-class
   * scenes); without it the tree is a regular SAH tree.  Closest hits are unchanged: depth ties are
   * still resolved by the reference's primitive order, which is computed as well and uploaded as a
   * per-primitive rank. */
  RAYCA_BUILDER_SAH = 1
};
enum {
  /* front-to-back, best-t culled traversal (default).  The closest hit is order independent
   * except for exact depth ties, which "lowest primitive index wins" resolves the way the
   * reference's strict-< DFS does (bvh/blas.rs:151,161,169). */
  RAYCA_TRAVERSAL_ORDERED = 0,
  /* visits exactly the boxes BvhNode::intersects visits (bvh/blas.rs:129-177): every child whose
   * box passes the slab test, no culling by the current best t.  Slow; used to verify ORDERED. */
  RAYCA_TRAVERSAL_EXHAUSTIVE = 1
};

enum {
  /* generation-by-generation rendering where the Config allows it (Flat; Pathtracer with Nee/None direct
   * and Cosine/Hemisphere indirect sampling, one indirect sample per vertex, no roulette) -- FUSED up to two
   * generations, WAVEFRONT from three (measured) -- else the stack machine */
  RAYCA_ENGINE_AUTO = 0,
  /* always the per-pixel stack machine (k_general): every IntegratorStrategy / SamplerStrategy the
   * reference has.  Same results as AUTO where both apply (tested); slower. */
  RAYCA_ENGINE_GENERAL = 1,
  /* the generation kernels' frames with traversal split from shading: lean one-thread-per-ray trace
   * kernels + a streaming shade kernel per generation (wavefront.inc).  Same Configs as the generation
   * kernels, same bits (tested). */
  RAYCA_ENGINE_WAVEFRONT = 2,
  /* the fused persistent kernel per generation (k_generation) */
  RAYCA_ENGINE_FUSED = 3
};

enum {
  /* camera rays of a one-sample Flat frame on a RAYCA_BUILDER_SAH scene: the scene times both kernels on its first
   * large frames and keeps the faster (same bits) */
  RAYCA_CAMERA_AUTO = 0,
  RAYCA_CAMERA_GENERATION = 1, /* the fused generation kernel: a wave keeps its 64 rays until the slowest has finished */
  RAYCA_CAMERA_REFILL = 2      /* lane refill (refill.hip): finished lanes take new pixels while long rays keep theirs */
};

typedef struct RaycaBuildOptions {
  uint32_t builder;   /* RAYCA_BUILDER_* */
  uint32_t device;    /* HIP device ordinal */
  /* 1: run the BLAS builder on the host threads instead of the GPU (bvh_build.hip).  Same tree, same boxes,
   * same primitive order either way (tested); the GPU builder is used from 4096 primitives per BLAS up. */
  uint32_t build_on_host;
  uint32_t reserved[5];
} RaycaBuildOptions;

/* Which rows of the frame this call renders (multi-GPU tile sharding): rows are dealt to `parts`
 * participants in bands of `band_rows`; participant `part` renders bands part, part+parts, ...
 * Output holds only those rows, packed in ascending row order.  parts==1 => the whole frame. */
typedef struct RaycaTile {
  uint32_t part;
  uint32_t parts;
  uint32_t band_rows;
  uint32_t reserved;
} RaycaTile;

typedef struct RaycaRenderOptions {
  uint32_t traversal;     /* RAYCA_TRAVERSAL_* */
  uint32_t collect_stats; /* 1: run the instrumented kernel variant that counts boxes/triangles */
  RaycaTile tile;         /* all zero => whole frame */
  void* stream;           /* hipStream_t to launch on, NULL => the scene's own stream */
  uint32_t engine;        /* RAYCA_ENGINE_*: which kernel family renders the frame */
  /* Frame context 0..7.  Each context owns its work buffers and its default stream, so frames rendered with
   * different contexts (and different streams) may be in flight at the same time; calls that use the same
   * context are serialised.  The scene (BVH, triangles, materials) is shared. */
  uint32_t context;
  uint32_t camera_rays;   /* RAYCA_CAMERA_* */
  uint32_t reserved;      /* must be zero (as every `reserved` field of this header) */
  /* Two hipEvent_t handles (or NULL), so that a frame loop needs ONE call per frame: the frame's stream waits for
   * `wait_event` before its first kernel (e.g. "the previous gather out of this buffer has finished") and `record_event`
   * is recorded on it behind the last one (e.g. for the comm stream to wait on).  rayca_hip_render_device only. */
  void* wait_event;
  void* record_event;
} RaycaRenderOptions;

/* Kernel classes RaycaStats.class_ms / class_launches are indexed by */
enum {
  RAYCA_KERNEL_GENERATION = 0,    /* k_generation: the fused persistent kernel of a generation (kernels.hip)        */
  RAYCA_KERNEL_FLAT_REFILL = 1,   /* k_flat_refill: camera rays with lane refill (refill.hip)                        */
  RAYCA_KERNEL_WF_TRACE = 2,      /* k_wf_trace: closest hits, one ray per lane (wavefront.inc)                      */
  RAYCA_KERNEL_QUEUE_REFILL = 3,  /* k_queue_refill: closest hits of a bounce generation, lane refill (refill.hip)   */
  RAYCA_KERNEL_WF_SHADE = 4,      /* k_wf_shade: shading, NEE set-up, bounce sampling (wavefront.inc)                */
  RAYCA_KERNEL_WF_SHADOW = 5,     /* k_wf_shadow: shadow rays + direct sum, one pixel per lane (wavefront.inc)       */
  RAYCA_KERNEL_SHADOW_REFILL = 6, /* k_shadow_refill: the same with lane refill (refill.hip)                         */
  RAYCA_KERNEL_OTHER = 7,         /* k_general (the stack machine), k_resolve, k_trace_rays                          */
  RAYCA_KERNEL_CLASSES = 8
};

/* Filled by every render call (all counters are per call, summed over spp and generations). */
typedef struct RaycaStats {
  uint64_t rays_primary;
  uint64_t rays_shadow;
  uint64_t rays_bounce;
  uint64_t boxes_tested;     /* only with collect_stats: AABB slab tests (32 B each)            */
  uint64_t triangles_tested; /* only with collect_stats: ray/triangle tests (36 B each)         */
  uint64_t hits_shaded;      /* only with collect_stats */
  /* only with collect_stats: SIMD-slot accounting of the traversal.  Every trip a wave
   * takes through the node loop books 64 x (boxes per node) slots, every trip through the leaf loop 64
   * slots, whatever the number of lanes still taking part: what the lock-step wave pays.
   * boxes_tested / wave_box_slots is the lane utilisation of the node loop. */
  uint64_t wave_box_slots;
  uint64_t wave_triangle_slots;
  float kernel_ms;           /* HIP-event time over all kernels of the frame, on the launch stream */
  float trace_kernel_ms;     /* the traversal kernels alone (the roofline kernel)                */
  uint32_t kernel_launches;
  uint32_t trace_kernel_launches;
  uint32_t rows_rendered;
  /* node format of this frame's launches: bit 0 = generation 0 used 4-wide nodes, bit 1 = the bounce
   * generations did, bits 2 / 3 = the same for fp16 node boxes, bit 8 = this was a calibration frame (the scene
   * is still timing the formats), bit 9 = a calibration frame of the camera-ray kernel choice (Flat frames: fused
   * generation kernel or lane-refill kernel, timed FIRST, on 4-wide f32 nodes; the node formats are timed afterwards on
   * the kernel that won), bit 10 = this frame's camera rays ran on
   * the lane-refill kernel, bit 11 = the scene's 4-wide / fp16 node formats were still being made when this frame was
   * issued (a thread started by rayca_hip_scene_create encodes and uploads them; until then frames traverse the binary
   * f32 nodes and nothing is timed -- same pixels either way), bit 12 = where "binary f32" nodes are traversed by the
   * conservative (RAYCA_BUILDER_SAH) kernels they are read as 48-B centre / half-extent records (three 16-B loads per
   * node instead of four; 0: 64-B min / max nodes).  Wavefront-engine frames report the formats their kernels are
   * compiled for (bits 0-3). */
  uint32_t node_format;
  /* HIP-event time (ms, summed over the launches of the call) and number of launches per kernel class (RAYCA_KERNEL_*):
   * class_ms[k] / class_launches[k] is the live average launch duration of that kernel -- what bench.py prices its
   * roofline with, and what the rocprofv3 kernel trace of the same command must agree with. */
  float class_ms[8];
  uint32_t class_launches[8];
} RaycaStats;

typedef struct RaycaSceneInfo {
  uint32_t triangle_count;
  uint32_t sphere_count;
  uint32_t blas_count;
  uint32_t node_count;      /* device BVH nodes (64 B each: two child boxes) */
  uint32_t max_depth;       /* traversal stack entries per ray (LDS) the device BVH needs */
  uint32_t light_count;
  uint64_t device_bytes;    /* HBM resident for this scene */
  float build_ms;           /* rayca_hip_scene_create of this scene: flatten + BVH build + binary-node layout + upload,
                               i.e. until a frame can be rendered (the other node formats follow on their own thread) */
  float runtime_init_ms;    /* what that call spent before it: HIP context + code object load, ~0 except for a
                               process's first scene on a device */
} RaycaSceneInfo;

typedef struct RaycaScene RaycaScene; /* opaque: owns the device-resident scene + BVH */

/* ---- entry points -------------------------------------------------------------------------- */

uint32_t rayca_hip_version(void);
/* number of visible HIP devices; 0 (not an error) when there is none */
int32_t rayca_hip_device_count(void);
/* Host-side self-checks that need no GPU: the outward fp16 rounding of the steering boxes (exhaustive over all finite
 * halves), and the multi-threaded device-layout passes of scene_create against their sequential forms on a random
 * tree (identical arrays).  RAYCA_OK or an error with a message in rayca_hip_last_error. */
int32_t rayca_hip_selftest(void);
/* copies the calling thread's last error message (NUL terminated) */
void rayca_hip_last_error(char* buf, size_t len);

/* Config::default() -- rayca-soft/src/config.rs:51-55 */
void rayca_hip_config_default(RaycaConfig* out);

/* The first half of SoftRenderer::draw (scene.rs:90-99): SceneDrawInfo::new, BvhScene::from_scene,
 * Tlas::builder()...build.  Flattens the graph, builds the BVH (cfg->bvh==0 => max_depth 0) and
 * uploads everything to HBM.  The reference repeats this on every draw; here the handle may be
 * reused for any number of render calls. `cfg` may be NULL (defaults); only cfg->bvh is read. */
int32_t rayca_hip_scene_create(const RaycaSceneDesc* desc, const RaycaConfig* cfg,
                               const RaycaBuildOptions* opts, RaycaScene** out);
/* Drop of the BvhScene / Tlas at the end of draw() (scene.rs:154).  Returns at once: the scene's last frames are waited for and
 * its device memory, streams and host arrays released by a thread of the library (7-27 ms of hipFree / hipStreamDestroy that
 * a host rebuilding the scene for every frame would otherwise pay per frame).  The handle is invalid from the call on.
 * Pending releases are completed before the next rayca_hip_scene_create allocates, by rayca_hip_scene_reap, and when the
 * library is unloaded. */
int32_t rayca_hip_scene_destroy(RaycaScene* scene);
/* Waits until every scene handed to rayca_hip_scene_destroy so far has been released (a host that wants the device memory
 * back at a known point).  No reference counterpart. */
int32_t rayca_hip_scene_reap(void);
int32_t rayca_hip_scene_info(const RaycaScene* scene, RaycaSceneInfo* out);
/* rayca_hip_scene_create returns as soon as frames can be rendered -- on the binary f32 nodes; the 4-wide and fp16
 * node formats a RAYCA_BUILDER_SAH scene times against them are encoded and uploaded by a thread of their own, and
 * frames pick them up when they are there (RaycaStats.node_format bit 11).  This waits for that thread: for hosts
 * that want every frame from the first on to be eligible for every format (benchmarks, tests).  Never required:
 * the pixels are the same with every format.  No reference counterpart. */
int32_t rayca_hip_scene_finish(RaycaScene* scene);

/* The second half of SoftRenderer::draw (scene.rs:101-150): the pixel loop.  Renders
 * width x height with camera_draw_infos[0] and writes RGBA8 (rgba8.rs:75-84) and/or the
 * pre-quantisation float colour (after /spp and gamma, before the u8 conversion) to HOST memory.
 * Either output pointer may be NULL.  Synchronous. */
int32_t rayca_hip_render(RaycaScene* scene, const RaycaConfig* cfg, uint32_t width,
                         uint32_t height, const RaycaRenderOptions* opts, uint8_t* rgba8_out,
                         float* rgba32f_out, RaycaStats* stats_out);

/* Same frame, outputs left in DEVICE memory (hipMalloc'd by the caller, e.g. a torch tensor's
 * data_ptr) on opts->stream; asynchronous unless stats_out is non-NULL (stats need the events).
 * This is the entry the multi-GPU path uses: each rank renders its RaycaTile into device memory
 * and the frame-end gather (RCCL) runs on the same stream. */
int32_t rayca_hip_render_device(RaycaScene* scene, const RaycaConfig* cfg, uint32_t width,
                                uint32_t height, const RaycaRenderOptions* opts,
                                void* d_rgba8_out, void* d_rgba32f_out, RaycaStats* stats_out);

/* ---- several devices, one process ----------------------------------------------------------------------
 * Image rows shard across the devices exactly as RaycaTile shards them across ranks (bands of band_rows rows dealt
 * round-robin); every device holds the whole scene; the only exchange is ONE gather of RGBA8 rows to scenes[0]'s device
 * at frame end, then one de-interleave kernel there.  The multi-PROCESS form of the same frame (one rank per GPU,
 * torch.distributed / RCCL) is rayca_amd/distributed.py on top of rayca_hip_render_device; this entry is for a host that is
 * one process -- a Rust or C program calling this header. */
enum {
  /* ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI; librccl is opened at first use (RAYCA_ERR_RCCL if it is
   * not installed) */
  RAYCA_GATHER_RCCL = 0,
  /* hipMemcpyPeerAsync from every device to scenes[0]'s: no collective library involved */
  RAYCA_GATHER_PEER_COPY = 1
};
typedef struct RaycaMultiOptions {
  uint32_t traversal;        /* RAYCA_TRAVERSAL_* */
  uint32_t collect_stats;    /* as RaycaRenderOptions.collect_stats */
  uint32_t band_rows;        /* 0 => 8 */
  uint32_t gather;           /* RAYCA_GATHER_* */
  uint32_t engine;           /* RAYCA_ENGINE_* */
  uint32_t output_on_device; /* 1: rgba8_out is device memory of scenes[0]'s device, 0: host memory */
  uint32_t context;          /* frame context 0..7 of every scene (RaycaRenderOptions.context): frames issued with different
                                contexts overlap on the devices (rayca_hip_render_multi_issue) */
  uint32_t reserved;
} RaycaMultiOptions;

/* One frame on `count` devices.  scenes[i] is a handle created (from the same RaycaSceneDesc) on the device that renders
 * part i; scenes[0]'s device assembles the frame.  rgba8_out receives width x height RGBA8 (host memory unless
 * opts->output_on_device).  stats_out: NULL or `count` entries, one per device.  count == 1 is rayca_hip_render.
 * Synchronous: rayca_hip_render_multi_issue + rayca_hip_render_multi_wait on frame context opts->context. */
int32_t rayca_hip_render_multi(RaycaScene* const* scenes, uint32_t count, const RaycaConfig* cfg, uint32_t width,
                               uint32_t height, const RaycaMultiOptions* opts, void* rgba8_out,
                               RaycaStats* stats_out);

/* The same frame, asynchronously: queues the rendering of every part (frame context opts->context of every scene, each on
 * that context's own stream), the one exchange, the de-interleave and the copy into rgba8_out, and returns.  A host that
 * issues frames with contexts 0, 1, 2, 3 in turn and waits for a context only before it re-uses it keeps four frames in
 * flight on every device -- the tail of one frame then runs under the head of the next, which is what a frame-at-a-time
 * loop leaves on the table (one GPU, 1080p primary + shadow: 0.51 -> 0.38 ms per frame; a rank's eighth 0.19 -> 0.06 ms).
 * rgba8_out (and host memory it points to) must stay valid until the matching wait, and so must every scene handle: wait
 * for the frames a scene takes part in before rayca_hip_scene_destroy.  Frames of one context are serialised.  No statistics (they need a synchronisation per frame: use rayca_hip_render_multi).
 * The drop-in host loop (draw.rs:7-9 called per frame) is `issue(ctx = f % 4)`, `wait(ctx = (f + 1) % 4)`. */
int32_t rayca_hip_render_multi_issue(RaycaScene* const* scenes, uint32_t count, const RaycaConfig* cfg, uint32_t width,
                                     uint32_t height, const RaycaMultiOptions* opts, void* rgba8_out);
/* Waits until the frame last issued with frame context `context` on these scenes has landed in its rgba8_out; returns
 * that frame's status.  RAYCA_OK at once if there is none. */
int32_t rayca_hip_render_multi_wait(RaycaScene* const* scenes, uint32_t count, uint32_t context);

/* RAYCA_OK if librccl can be opened and offers what RAYCA_GATHER_RCCL uses, else RAYCA_ERR_RCCL with the reason in
 * rayca_hip_last_error.  Needs no GPU. */
int32_t rayca_hip_rccl_status(void);

/* Number of rows a RaycaTile covers in a frame of `height` rows (host-side helper, no GPU). */
uint32_t rayca_hip_tile_rows(const RaycaTile* tile, uint32_t height);

/* Debug/parity entry (the analogue of Tlas::intersects, bvh/tlas.rs:271-275): trace `count`
 * caller-supplied rays (origin xyz, dir xyz; 6 floats per ray, HOST memory) and return per ray
 * t (f32::MAX on miss), primitive index in the scene's post-build primitive order (RAYCA_NONE on
 * miss) and the barycentrics u,v.  */
int32_t rayca_hip_trace_rays(RaycaScene* scene, const RaycaRenderOptions* opts, uint32_t count,
                             const float* rays, float* t_out, uint32_t* prim_out, float* uv_out,
                             RaycaStats* stats_out);

/* Post-build BVH read-back for parity tests against the oracle's literal SAH build:
 * `prim_order[i]` = index (in flatten order) of the primitive stored at slot i.  Buffers may be
 * NULL to query sizes through rayca_hip_scene_info. */
int32_t rayca_hip_scene_primitive_order(const RaycaScene* scene, uint32_t* prim_order,
                                        uint32_t capacity);

/* Node read-back for tests (no reference counterpart): copies the scene's binary nodes to HOST memory.
 * which = 0: the 64-B nodes (12 f32: min xyz, max xyz of the left child's box, then of the right child's; two u32 child
 * references; 8 B padding).  which = 1: the 48-B centre / half-extent records the conservative kernels of a
 * RAYCA_BUILDER_SAH scene read instead (RaycaStats.node_format bit 12; 12 f32: centre xyz, half extent xyz per child;
 * the low 16 bits of the x / y half extents of a child hold the low / high half of its reference, an inner reference
 * being the child record's byte offset, 48 x its index) -- node i of one is node i of the other.  `bytes_out` receives the array's size; with out == NULL only that.  RAYCA_ERR_BAD_ARG if the
 * scene has no such array or `capacity_bytes` is too small. */
int32_t rayca_hip_scene_read_nodes(RaycaScene* scene, uint32_t which, void* out, uint64_t capacity_bytes, uint64_t* bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* RAYCA_HIP_H */
