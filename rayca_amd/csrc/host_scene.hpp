// host_scene.hpp -- host half of SoftRenderer::draw before the pixel loop
// (rayca-soft/src/scene.rs:90-99): SceneDrawInfo::new, BvhScene::from_scene, Tlas::builder().build,
// then conversion of the two-level BVH into the device layout.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rayca_hip.h"
#include "rayca_math.hpp"

namespace rayca {

// VertexExt (rayca-geometry/src/vertex.rs:64-72) x3 + material, as uploaded: 256 B per primitive, two 128-B
// lines.  Everything a hit needs unless its material has a normal texture -- vertex colours, normals, uvs, material --
// sits in the first line, tangents and bitangents in the second: an incoherent hit (bounce generations) costs one
// line instead of the 2-3 a packed 192-B record straddles.
struct PrimExt {
  float color[3][4];
  float normal[3][3];
  float uv[3][2];
  uint32_t material;
  uint32_t kind;       // RAYCA_GEOMETRY_*
  uint32_t node;       // world-transform index (spheres)
  uint32_t pad0[2];
  float tangent[3][3];
  float bitangent[3][3];
  uint32_t pad1[14];
};
static_assert(sizeof(PrimExt) == 256, "PrimExt is 256 B");
static_assert(offsetof(PrimExt, tangent) == 128, "the second line starts with the tangents");

// A BLAS as the builders leave it: an arena of nodes, root at 0, children by index.  (The reference stores the same tree
// with children adjacent and slot 1 unused, bvh/blas.rs:11-15,250-256; nothing outside the builder sees that numbering --
// what is observable is the primitive order and the boxes -- and renumbering half a million nodes cost 23 ms per tree.)
struct Box {
  F4 a, b;
};
struct BuildNode {
  Box bounds;
  uint32_t offset = 0, count = 0;  // primitive range (leaf) -- count == 0 => inner (or the root of an empty BLAS)
  int32_t left = -1, right = -1;   // indices into the arena
};

// Device BVH node, 64 B: both children's boxes + packed child references.
//   q0 = (l.min.x, l.min.y, l.min.z, l.max.x)   q1 = (l.max.y, l.max.z, r.min.x, r.min.y)
//   q2 = (r.min.z, r.max.x, r.max.y, r.max.z)   q3 = (bits: left ref, right ref, 0, 0)
// child ref: bit31 = leaf; leaf: bits30..25 = count-1 (<=64 primitives), bits24..0 = first primitive;
// inner: node index.
struct DevNode {
  float q[12];
  uint32_t left, right, pad0, pad1;
};
static_assert(sizeof(DevNode) == 64, "DevNode is 64 B");
// 4-wide node, 128 B: the boxes of up to four children, SoA (min x[4], min y[4], min z[4], max x[4], ...),
// then four packed child references; unused slots carry an inverted box and kNoChild.
struct DevNode4 {
  float lo[3][4];
  float hi[3][4];
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(DevNode4) == 128, "DevNode4 is 128 B");
// The same two trees with fp16 boxes (RAYCA_BUILDER_SAH only, where boxes merely steer the search): coordinates
// are (x - half_center) * half_scale, minima rounded down and maxima up, so every box contains its f32 original.
// 32 B and 64 B per node: half the bytes through the L1 pipe, decoded for free by v_fma_mix_f32.
struct DevNodeH {
  uint16_t h[12];  // l.min xyz, l.max xyz, r.min xyz, r.max xyz
  uint32_t left, right;
};
static_assert(sizeof(DevNodeH) == 32, "DevNodeH is 32 B");
struct DevNode4H {
  uint16_t lo[3][4];
  uint16_t hi[3][4];
  uint32_t child[4];
};
static_assert(sizeof(DevNode4H) == 64, "DevNode4H is 64 B");
// Box of a slot that must never be entered (an empty BLAS, an unused slot of a 4-wide node): a zero-size box far away.
constexpr float kNowhere = 1.0e18f;
constexpr uint32_t kNoChild = 0x7FFFFFFFu;
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kLeafMaxPrims = 64;
constexpr uint32_t kLeafFirstMask = 0x01FFFFFFu;

struct HostLight {
  uint32_t kind, node, material;
  float intensity;
  Color color;
  F4 attenuation, ab, ac;
  Trs local;  // node-LOCAL transform: what the samplers use (nee.rs:85,133-134)
};

struct HostBlas {
  uint32_t model = 0;
  std::vector<BuildNode> nodes;     // arena, root at 0
  std::vector<uint32_t> prims;      // indices into HostScene::prims (post-build order)
  // A BLAS built by the device builder stays on the device (BlasDeviceTree): `nodes` then holds the root only, and the
  // device lays its nodes out straight into the scene's node array (DeviceSegment, gpu_emit_tree).
  void* dev_tree = nullptr;
  uint32_t dev_node_count = 0, dev_need = 0;
};
// A run of the scene's binary node array that the device writes itself, from a tree it still holds.
struct DeviceSegment {
  uint32_t first = 0, count = 0;    // DevNode indices [first, first + count)
  uint32_t prim_base = 0;           // first global primitive slot of the BLAS
  void* tree = nullptr;
};

struct HostPrim {
  uint32_t kind, node, material, src;
  F4 p[3];       // model space
  F4 centroid;   // model space, (v0+v1+v2)*0.3333
  F4 wp[3];      // world space = trs * p  (bit-identical to what Triangle::intersects recomputes)
  F4 wcentroid, wmin, wmax;
  F4 center;     // sphere
  float radius;
};   // (its shading record lives in HostScene::ext, same index: that array goes to the device as it is)

// std::vector<T>::resize without the zero fill: the flatten order array is ~0.5 KB per triangle (126 MB for the atrium), its
// slots are written exactly once by the threads that fill them, and value-initialising it first costs a single-threaded
// pass over all of it (page faults included) -- a quarter of scene_create on the benchmark scene
//
// Blocks of 1 MiB and more come from (and go back to) a process-wide pool of 2-MiB-aligned blocks (big_block_take / _give,
// host_scene.cpp): a host that builds a scene per draw() -- as the reference's draw() does -- touches ~150 MB of fresh pages per
// build otherwise, and the page faults of those were most of the flatten phases' time (RAYCA_HOST_POOL_MB caps what the pool
// keeps, default 1024; 0 switches it off).
void* big_block_take(size_t bytes);
void big_block_give(void* p) noexcept;
constexpr size_t kBigBlockMin = size_t(1) << 20;
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAllocator<U>;
  };
  T* allocate(size_t n) {
    if (n * sizeof(T) >= kBigBlockMin) return static_cast<T*>(big_block_take(n * sizeof(T)));
    return std::allocator<T>::allocate(n);
  }
  void deallocate(T* p, size_t n) noexcept {
    if (n * sizeof(T) >= kBigBlockMin) big_block_give(p);
    else std::allocator<T>::deallocate(p, n);
  }
  template <class U, class... Args>
  void construct(U* p, Args&&... args) {
    if constexpr (sizeof...(Args) == 0) ::new (static_cast<void*>(p)) U;
    else ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
};

struct HostScene {
  std::vector<Trs> local_trs, world_trs;
  bool has_camera = false;
  uint32_t camera_node = 0;
  float camera_yfov = 0.0f;
  std::vector<HostLight> lights;
  std::vector<RaycaMaterial> materials;
  std::vector<RaycaTexture> textures;
  std::vector<RaycaImage> images;
  std::vector<uint8_t> image_bytes;
  std::vector<HostPrim, DefaultInitAllocator<HostPrim>> prims;      // flatten order
  // what the device reads per primitive, flatten order, written in place by the flatten threads and uploaded from here
  // without another copy (the library releases both once they are on the device): the 256-B shading records, and the
  // world-space triangles, 9 floats each (a sphere's slot: filled by the library when it numbers the sphere table)
  std::vector<PrimExt, DefaultInitAllocator<PrimExt>> ext;
  std::vector<float, DefaultInitAllocator<float>> tris;
  std::vector<HostBlas> blas;       // in TLAS blas_nodes order (post-build)
  uint32_t triangle_count = 0, sphere_count = 0;

  // device layout
  std::vector<DevNode, DefaultInitAllocator<DevNode>> dev_nodes;   // (every node is written whole: no zero fill of 17 MB first)
  // runs of dev_nodes that are NOT filled in here but written on the device (the library emits them into the uploaded
  // array and copies them back when the host needs them: finish_node_formats), and the box padding they are written with
  std::vector<DeviceSegment> dev_segments;
  std::vector<void*> dev_trees;     // every device tree handed out during the build, for the owner to release
  float pad_rel = 0.0f, pad_abs = 0.0f;
  uint32_t root_ref = 0;            // packed ref of the root
  F4 root_min, root_max;            // root box (tested before anything else, blas.rs:136-139)
  std::vector<uint32_t> prim_order; // slot -> flatten index
  uint32_t max_depth = 0;           // stack entries a traversal can have pending at once
  // the same tree collapsed to 4-wide nodes (every other level skipped): what the kernels traverse
  std::vector<DevNode4> dev_nodes4;
  std::vector<DevNodeH, DefaultInitAllocator<DevNodeH>> dev_nodes_h;     // fp16 versions (SAH builder), see DevNodeH
  std::vector<DevNode4H, DefaultInitAllocator<DevNode4H>> dev_nodes4_h;
  bool other_formats_wanted = false;     // finish_node_formats has something to make for this scene
  float half_center[3] = {0, 0, 0};
  float half_scale = 1.0f;               // a power of two
  uint32_t root_ref4 = 0;
  uint32_t max_depth4 = 0;          // pending stack entries for the 4-wide tree
  std::vector<uint32_t> tie_rank;   // RAYCA_BUILDER_SAH only: slot -> position in the reference's order
  // RAYCA_BUILDER_SAH only: the reference tree's leaves.  A triangle is a candidate of the reference
  // iff the slab test passes for its reference LEAF box (every ancestor box contains the leaf box and
  // the slab arithmetic is monotone in the box corners, so the ancestors pass whenever the leaf does).
  std::vector<float> ref_leaf_boxes;   // 8 floats per leaf: min xyz, pad, max xyz, pad
  std::vector<uint32_t> ref_leaf_of;   // slot -> reference leaf index
};

// Host threads a scene build may use: the CPUs this process may run on (sched_getaffinity), at most 16 (a one-GPU job's
// share of a node; std::thread::hardware_concurrency() reports the whole machine, and a few hundred threads for a 50 ms
// phase cost more than they do), RAYCA_HOST_THREADS overrides.
unsigned host_threads();

// Returns RAYCA_OK or an error code with `err` filled.
struct BuildHooks {
  // called once `out.prims` (flatten order: triangles with world-space vertices, shading records, spheres) and
  // `out.world_trs` are final -- before any BVH work -- so that the caller can start moving them to the device
  std::function<void()> on_prims_ready;
  // called once `out.prim_order` (slot -> flatten index) is final -- before the device node layouts are made
  std::function<void()> on_order_ready;
  // false leaves the three node formats other than binary f32 to a later finish_node_formats(out)
  bool with_formats = true;
  // two streams (hipStream_t) for the device builder, read after on_prims_ready has returned: the two trees of a
  // RAYCA_BUILDER_SAH scene are built side by side and only overlap if their streams sit on different hardware queues,
  // which the owner of the streams can arrange and the builder cannot (api.inc make_scene_streams)
  void* const* build_streams = nullptr;
};
int32_t build_host_scene(const RaycaSceneDesc& d, bool use_bvh, uint32_t builder, HostScene& out, std::string& err, const BuildHooks& hooks = {});
// (`cancel`, optional: looked at between the phases; when it reads true the function returns early and the formats are incomplete)
void finish_node_formats(HostScene& s, const std::atomic<bool>* cancel = nullptr);

// Device BLAS builder (bvh_build.hip): the same tree and primitive order as the host builder, built on the GPU.
// host_scene.cpp does not link HIP; the library registers the function before it builds a scene.
struct BlasBuildInput {
  const float* cent[3];   // world centroids, SoA, `count` entries (host memory)
  const float* bmin[3];   // world boxes
  const float* bmax[3];
  uint32_t count;
  float root_min[3], root_max[3];
  bool seed_origin;       // candidate boxes start at the origin (the reference) or empty (RAYCA_BUILDER_SAH)
  uint32_t max_depth;
  uint32_t device;
  void* stream = nullptr;  // hipStream_t to build on (the caller keeps it), or null: the builder makes one of its own
  // layout of a kept tree (BlasDeviceTree): subtrees that are no dearer as one leaf are laid out as one (kLeafNodeCost)
  float layout_node_cost = 0.0f;
  uint32_t layout_leaf_max = 8;
};
// Leaves of the traversed tree (RAYCA_BUILDER_SAH; the reference's own tree is laid out as it is).  The tree only has to be
// conservative -- the reference-leaf filter decides what a ray may hit -- and the reference's cost model puts no price on a
// node step, so it splits down to one or two triangles per leaf even where the children's boxes are the parent's box again
// (the two triangles of a quad).  The layouts price every subtree both ways, bottom-up -- as one leaf: primitives x area of
// its box; split: node cost x area + the children's prices (unit: one triangle test) -- and lay it out as ONE leaf where
// that is no dearer and the leaf stays within kLeafCollapseMax primitives.  Measured: atrium frame
// 0.4044 -> 0.3797 ms (a third of the leaves are such pairs), soup unchanged (random triangles: no subtree qualifies); at
// node cost 1.0 the atrium gains another 0.4 % and the soup loses 10 %.
constexpr float kLeafNodeCost = 0.5f;
constexpr uint32_t kLeafCollapseMax = 8;
// `keep` non-null: the finished tree is not copied to the host (arena is left alone) but kept on the device, with the size of
// its binary-node layout (pre-order, as DevBuilder::emit_blas_flat numbers it) and its stack need already worked out;
// gpu_emit_tree writes that layout into a device node array and releases the tree, gpu_release_tree only releases it.
struct BlasDeviceTree {
  void* handle = nullptr;
  uint32_t node_count = 0;   // DevNodes the layout takes (inner nodes + the chains of leaves above 64 primitives)
  uint32_t need = 0;         // stack entries a traversal of it can have pending
};
using BlasBuildFn = bool (*)(const BlasBuildInput&, std::vector<uint32_t>& order, std::vector<BuildNode>& arena, std::string& err, BlasDeviceTree* keep);
void set_device_blas_builder(BlasBuildFn fn, uint32_t device);
int32_t selftest_half_rounding(std::string& err);
int32_t selftest_device_layouts(std::string& err);
bool gpu_build_blas(const BlasBuildInput& in, std::vector<uint32_t>& order, std::vector<BuildNode>& arena, std::string& err, BlasDeviceTree* keep);
bool gpu_emit_tree(void* tree, DevNode* device_nodes, uint32_t first, uint32_t prim_base, float pad_rel, float pad_abs, std::string& err);
void gpu_release_tree(void* tree);
const void* gpu_builder_any_kernel();  // host stub of one kernel of bvh_build.hip (to load that code object ahead of time)

}  // namespace rayca
