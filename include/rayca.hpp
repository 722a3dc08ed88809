// rayca.hpp -- C++ host mirror of the reference's scene/renderer interface for the hot path.
//
// The reference is a Rust workspace; its interface for this path is the trait
//     pub trait Draw { fn draw(&mut self, scene: &Scene, image: &mut Image); }   rayca-soft/src/draw.rs:7-9
// implemented by SoftRenderer { pub config: Config }                              rayca-soft/src/scene.rs:11-14,88
// over rayca-model's Scene / Model / Node / Mesh / Primitive / Geometry / Material / Camera / Light.
// This header restates those types with the same names, fields, builders and defaults so that a
// program written against rayca-soft reads the same here (compare tests/cpp/host_mirror.cpp with
// rayca-soft/tests/gltf.rs), and lowers them to the C ABI of rayca_hip.h.
//
// Header-only, C++17, no dependency beyond rayca_hip.h / librayca_hip.so.  Nothing in this file
// computes anything that reaches a pixel: world transforms, BVH, intersection and shading all happen
// behind the C ABI.  Errors that the reference raises as panics are thrown as rayca::Error.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "rayca_hip.h"

namespace rayca {

struct Error : std::runtime_error {
  int32_t status;
  Error(int32_t s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// ---- rayca-math ------------------------------------------------------------------------------------
struct Vec3 {  // rayca-math/src/vec3.rs:24-28
  float x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  static Vec3 splat(float v) { return Vec3(v, v, v); }
};
using Point3 = Vec3;  // only ever stored, never transformed, on this side of the ABI

struct Vec2 {
  float x = 0, y = 0;
  Vec2() = default;
  Vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct Quat {  // rayca-math/src/quat.rs:14-18, identity by default
  float x = 0, y = 0, z = 0, w = 1;
  Quat() = default;
  Quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
  // Quat::axis_angle  quat.rs:66-75 (f32 arithmetic, normalised)
  static Quat axis_angle(Vec3 axis, float angle_radians) {
    const float half = angle_radians / 2.0f;
    const float s = std::sin(half), c = std::cos(half);
    Quat q(axis.x * s, axis.y * s, axis.z * s, c);
    const float n = std::sqrt(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
    return q;
  }
};

struct Color {  // rayca-math/src/color/mod.rs:27-35
  float r = 0, g = 0, b = 0, a = 1;
  Color() = default;
  Color(float r_, float g_, float b_, float a_) : r(r_), g(g_), b(b_), a(a_) {}
  // impl From<u32> for Color (0xRRGGBBAA)  color/mod.rs:188-197
  static Color from(uint32_t rgba) {
    return Color((float)(uint8_t)(rgba >> 24) / 255.0f, (float)(uint8_t)(rgba >> 16) / 255.0f, (float)(uint8_t)(rgba >> 8) / 255.0f,
                 (float)(uint8_t)rgba / 255.0f);
  }
  static Color white() { return Color(1, 1, 1, 1); }
  static Color black() { return Color(0, 0, 0, 1); }
};

enum class ColorType : uint32_t { RGB8 = RAYCA_COLOR_RGB8, RGBA8 = RAYCA_COLOR_RGBA8, RGBA32F = RAYCA_COLOR_RGBA32F };  // color/mod.rs:19-25

struct Trs {  // rayca-math/src/trs.rs:75-86
  Vec3 translation;
  Quat rotation;
  Vec3 scale = Vec3(1, 1, 1);
  struct Builder {  // trs.rs:14-73
    Vec3 t;
    Quat r;
    Vec3 s = Vec3(1, 1, 1);
    Builder& translation(Vec3 v) { t = v; return *this; }
    Builder& rotation(Quat q) { r = q; return *this; }
    Builder& scale(Vec3 v) { s = v; return *this; }
    Trs build() const { Trs o; o.translation = t; o.rotation = r; o.scale = s; return o; }
  };
  static Builder builder() { return Builder(); }
};

// ---- rayca-util: Handle / Pack ---------------------------------------------------------------------
template <class T>
struct Handle {  // rayca-util/src/handle.rs: u32 id, NONE = u32::MAX
  uint32_t id = RAYCA_NONE;
  Handle() = default;
  explicit Handle(uint32_t i) : id(i) {}
  bool is_valid() const { return id != RAYCA_NONE; }
  static Handle none() { return Handle(); }
};

template <class T>
struct Pack {  // rayca-util/src/pack.rs: push returns the handle
  std::vector<T> items;
  Handle<T> push(T v) {
    items.push_back(std::move(v));
    return Handle<T>((uint32_t)items.size() - 1u);
  }
  const T* get(Handle<T> h) const { return h.id < items.size() ? &items[h.id] : nullptr; }
  T* get_mut(Handle<T> h) { return h.id < items.size() ? &items[h.id] : nullptr; }
  size_t len() const { return items.size(); }
};

// ---- rayca-geometry ---------------------------------------------------------------------------------
struct VertexExt {  // vertex.rs:64-72, defaults vertex.rs:164-175
  Color color = Color::white();
  Vec3 normal = Vec3(0, 0, 1);
  Vec3 tangent;
  Vec3 bitangent;
  Vec2 uv;
};
struct Vertex {  // vertex.rs:137-142
  Point3 pos;
  VertexExt ext;
  Vertex() = default;
  Vertex(float x, float y, float z) : pos(x, y, z) {}
};

enum class ComponentType : uint32_t { U8 = RAYCA_INDEX_U8, U16 = RAYCA_INDEX_U16, U32 = RAYCA_INDEX_U32 };  // triangle.rs:180-201

struct TriangleIndices {  // triangle.rs:215-307: byte-packed indices + their component type
  std::vector<uint8_t> indices;
  ComponentType index_type = ComponentType::U8;
  static TriangleIndices from_u8(const std::vector<uint8_t>& v) {
    TriangleIndices t;
    t.indices = v;
    t.index_type = ComponentType::U8;
    return t;
  }
  static TriangleIndices from_u16(const std::vector<uint16_t>& v) {
    TriangleIndices t;
    t.indices.resize(v.size() * 2);
    std::memcpy(t.indices.data(), v.data(), t.indices.size());
    t.index_type = ComponentType::U16;
    return t;
  }
  static TriangleIndices from_u32(const std::vector<uint32_t>& v) {
    TriangleIndices t;
    t.indices.resize(v.size() * 4);
    std::memcpy(t.indices.data(), v.data(), t.indices.size());
    t.index_type = ComponentType::U32;
    return t;
  }
  uint32_t get_index_size() const { return index_type == ComponentType::U8 ? 1u : index_type == ComponentType::U16 ? 2u : 4u; }
  uint32_t get_index_count() const { return (uint32_t)(indices.size() / get_index_size()); }
};

struct TriangleMesh {  // triangle.rs:309-314
  std::vector<Vertex> vertices;
  TriangleIndices indices;

  static TriangleMesh unit() {  // triangle.rs:327-342
    TriangleMesh m;
    m.vertices = {Vertex(-1, 0, 0), Vertex(1, 0, 0), Vertex(0, 1, 0)};
    m.indices = TriangleIndices::from_u8({0, 1, 2});
    return m;
  }
  static TriangleMesh quad(Vec2 uv_scale = Vec2(1, 1)) {  // triangle.rs:344-378
    TriangleMesh m;
    const float p[4][2] = {{-0.5f, -0.5f}, {0.5f, -0.5f}, {0.5f, 0.5f}, {-0.5f, 0.5f}};
    const float uv[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    for (int i = 0; i < 4; ++i) {
      Vertex v(p[i][0], p[i][1], 0);
      v.ext.uv = Vec2(uv[i][0] * uv_scale.x, uv[i][1] * uv_scale.y);
      m.vertices.push_back(v);
    }
    m.indices = TriangleIndices::from_u8({0, 1, 2, 2, 3, 0});
    return m;
  }
  static TriangleMesh cube() {  // triangle.rs:380-548: 6 faces x 4 vertices, outward normals
    static const float n[6][3] = {{0, 0, 1}, {1, 0, 0}, {0, 0, -1}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
    static const float q[6][4][3] = {
        {{-.5f, -.5f, .5f}, {.5f, -.5f, .5f}, {.5f, .5f, .5f}, {-.5f, .5f, .5f}},
        {{.5f, -.5f, .5f}, {.5f, -.5f, -.5f}, {.5f, .5f, -.5f}, {.5f, .5f, .5f}},
        {{.5f, -.5f, -.5f}, {-.5f, -.5f, -.5f}, {-.5f, .5f, -.5f}, {.5f, .5f, -.5f}},
        {{-.5f, -.5f, -.5f}, {-.5f, -.5f, .5f}, {-.5f, .5f, .5f}, {-.5f, .5f, -.5f}},
        {{-.5f, .5f, .5f}, {.5f, .5f, .5f}, {.5f, .5f, -.5f}, {-.5f, .5f, -.5f}},
        {{-.5f, -.5f, -.5f}, {.5f, -.5f, -.5f}, {.5f, -.5f, .5f}, {-.5f, -.5f, .5f}}};
    static const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    TriangleMesh m;
    std::vector<uint8_t> idx;
    for (int f = 0; f < 6; ++f) {
      for (int k = 0; k < 4; ++k) {
        Vertex v(q[f][k][0], q[f][k][1], q[f][k][2]);
        v.ext.normal = Vec3(n[f][0], n[f][1], n[f][2]);
        v.ext.uv = Vec2(uv[k][0], uv[k][1]);
        m.vertices.push_back(v);
      }
      const uint8_t b = (uint8_t)(4 * f);
      for (uint8_t o : {0, 1, 2, 0, 2, 3}) idx.push_back((uint8_t)(b + o));
    }
    m.indices = TriangleIndices::from_u8(idx);
    return m;
  }
};

struct Sphere {  // sphere.rs:38-44
  Point3 center;
  float radius = 1.0f;
  Sphere() = default;
  Sphere(Point3 c, float r) : center(c), radius(r) {}
  static Sphere unit() { return Sphere(); }  // sphere.rs:64-66
};

struct Geometry {  // rayca-model Geometry enum
  std::variant<TriangleMesh, Sphere> value;
  Geometry(TriangleMesh m) : value(std::move(m)) {}
  Geometry(Sphere s) : value(s) {}
  static Geometry from_triangle_mesh(TriangleMesh m) { return Geometry(std::move(m)); }
  static Geometry from_sphere(Sphere s) { return Geometry(s); }
};

// ---- rayca-model ------------------------------------------------------------------------------------
struct Image {  // image.rs:26-36: row-major, top-left origin
  uint32_t w = 0, h = 0;
  ColorType color_type = ColorType::RGBA8;
  std::vector<uint8_t> data;
  Image() = default;
  Image(uint32_t width, uint32_t height, ColorType ct) : w(width), h(height), color_type(ct) { data.assign((size_t)w * h * texel_size(), 0); }
  static Image create(uint32_t width, uint32_t height, ColorType ct) { return Image(width, height, ct); }  // Image::new image.rs:49-61
  uint32_t width() const { return w; }
  uint32_t height() const { return h; }
  size_t texel_size() const { return color_type == ColorType::RGB8 ? 3 : color_type == ColorType::RGBA8 ? 4 : 16; }
  const uint8_t* bytes() const { return data.data(); }
  uint8_t* bytes_mut() { return data.data(); }
  // Image::dump_png  image.rs:160-169 (RGBA8 / RGB8 only; stored deflate blocks, no compression)
  void dump_png(const std::string& path) const;
};

struct Texture {  // texture.rs:35-39 (sampler: always the default one on this path, pbr.rs:96)
  Handle<Image> image;
  Texture() = default;
  explicit Texture(Handle<Image> i) : image(i) {}
};

struct PbrMaterial {  // material/pbr.rs:58-66
  Color color = Color::white();
  Handle<Texture> albedo;
  Handle<Texture> normal;
  float metallic_factor = 0.0f;
  float roughness_factor = 0.0f;
  Handle<Texture> metallic_roughness;
};
struct PhongMaterial {  // material/phong.rs:10-34
  Color ambient = Color::black(), emission = Color::black(), diffuse = Color::black(), specular = Color::black();
  float shininess = 0.0f;
};
struct GgxMaterial {  // material/ggx.rs:10-23
  Color diffuse = Color::black(), specular = Color::black();
  float roughness = 0.0f;
};
struct Material {  // material/mod.rs:15-20
  std::variant<PbrMaterial, PhongMaterial, GgxMaterial> value;
  Material() : value(PbrMaterial()) {}
  Material(PbrMaterial m) : value(m) {}
  Material(PhongMaterial m) : value(m) {}
  Material(GgxMaterial m) : value(m) {}
};

struct Camera {  // camera.rs:20-24; default = infinite_perspective(1, pi/4, 0.1)
  float yfov_radians = 0.78539816339744830962f;
};

struct PointLight {  // light/point.rs:12-21
  Color color = Color::white();
  float intensity = 1.0f;
  Vec3 attenuation = Vec3(0, 0, 1);
};
struct DirectionalLight {
  Color color = Color::white();
  float intensity = 1.0f;
};
struct QuadLight {  // light/quad.rs:15-23
  Vec3 ab, ac;
  Color color = Color::white();
  Handle<Material> material;
  float intensity = 1.0f;
  QuadLight() = default;
  QuadLight(Vec3 ab_, Vec3 ac_, Color c, Handle<Material> m) : ab(ab_), ac(ac_), color(c), material(m) {}  // QuadLight::new quad.rs:25-34
};
struct Light {  // light/mod.rs:15-19
  std::variant<DirectionalLight, PointLight, QuadLight> value;
  Light() : value(DirectionalLight()) {}
  Light(PointLight l) : value(l) {}
  Light(QuadLight l) : value(l) {}
  Light(DirectionalLight l) : value(l) {}
  static Light directional() { return Light(DirectionalLight()); }  // mod.rs:22-24
  static Light point() { return Light(PointLight()); }              // mod.rs:26-28
  bool is_quad() const { return std::holds_alternative<QuadLight>(value); }
  void set_intensity(float v) {  // mod.rs:38-44
    std::visit([v](auto& l) { l.intensity = v; }, value);
  }
};

struct Primitive {  // primitive.rs:9-14
  Handle<Geometry> geometry;
  Handle<Material> material;
  struct Builder {
    Handle<Geometry> g;
    Handle<Material> m;
    Builder& geometry(Handle<Geometry> h) { g = h; return *this; }
    Builder& material(Handle<Material> h) { m = h; return *this; }
    Primitive build() const { Primitive p; p.geometry = g; p.material = m; return p; }
  };
  static Builder builder() { return Builder(); }
};

struct Mesh {  // mesh.rs:32-35
  std::vector<Handle<Primitive>> primitives;
  struct Builder {
    std::vector<Handle<Primitive>> p;
    Builder& primitive(Handle<Primitive> h) { p.push_back(h); return *this; }
    Builder& primitives(std::vector<Handle<Primitive>> v) { p = std::move(v); return *this; }
    Mesh build() const { Mesh m; m.primitives = p; return m; }
  };
  static Builder builder() { return Builder(); }
};

struct Model;

struct Node {  // node.rs:11-32
  std::string name;
  Trs trs;
  std::vector<Handle<Node>> children;
  Handle<Mesh> mesh;
  Handle<Camera> camera;
  Handle<Light> light;
  Handle<Model> model;  // scene-level nodes only (scene.rs:107-113)
  struct Builder;
  static Builder builder();
};

struct Node::Builder {
  Node n;
  Builder& name(std::string s) { n.name = std::move(s); return *this; }
  Builder& trs(Trs t) { n.trs = t; return *this; }
  Builder& children(std::vector<Handle<Node>> c) { n.children = std::move(c); return *this; }
  Builder& mesh(Handle<Mesh> h) { n.mesh = h; return *this; }
  Builder& camera(Handle<Camera> h) { n.camera = h; return *this; }
  Builder& light(Handle<Light> h) { n.light = h; return *this; }
  Builder& model(Handle<Model> h) { n.model = h; return *this; }
  Node build() const { return n; }
};
inline Node::Builder Node::builder() { return Builder(); }

struct Model {  // model.rs:30-50
  std::string name = "Unknown";
  Node root;
  Pack<Node> nodes;
  Pack<Mesh> meshes;
  Pack<Primitive> primitives;
  Pack<Geometry> geometries;
  Pack<Material> materials;
  Pack<Texture> textures;
  Pack<Image> images;
  Pack<Camera> cameras;
  Pack<Light> lights;
};

struct Scene {  // rayca-model/src/scene.rs:47-53
  std::string name = "Unknown";
  Pack<Node> nodes;
  Pack<Model> models;
  Node root;
  // Scene::push_model  scene.rs:107-113
  Handle<Node> push_model(Model model) {
    const Handle<Model> mh = models.push(std::move(model));
    const Handle<Node> nh = nodes.push(Node::builder().model(mh).build());
    root.children.push_back(nh);
    return nh;
  }
};

// ---- rayca-soft -------------------------------------------------------------------------------------
enum class IntegratorStrategy : uint32_t {  // integrator/mod.rs:32-41
  Scratcher = 0, Raytracer = 1, Flat = 2, AnalyticDirect = 3, Direct = 4, Pathtracer = 5
};
enum class SamplerStrategy : uint32_t {  // sampler/mod.rs:41-50
  None = 0, Nee = 1, Hemisphere = 2, Cosine = 3, Brdf = 4, Mis = 5
};

struct Config {  // config.rs:10-49, same defaults
  bool bvh = true;
  uint32_t light_samples = 1;
  bool light_stratify = false;
  uint32_t samples_per_pixel = 1;
  bool russian_roulette = false;
  SamplerStrategy direct_sampler = SamplerStrategy::Nee;
  SamplerStrategy indirect_sampler = SamplerStrategy::Cosine;
  IntegratorStrategy integrator = IntegratorStrategy::Pathtracer;
  uint32_t max_depth = 5;
  float gamma = 1.0f;
  uint32_t seed = 0;  // no reference counterpart: key of the counter-based RNG (rayca_hip.h RaycaConfig)
  struct Builder;
  static Builder builder();
  uint32_t get_strate_count() const { return light_stratify ? (uint32_t)std::sqrt((float)light_samples) : 1u; }  // config.rs:72-78
  RaycaConfig to_abi() const {
    RaycaConfig r;
    rayca_hip_config_default(&r);
    r.bvh = bvh; r.light_samples = light_samples; r.light_stratify = light_stratify; r.samples_per_pixel = samples_per_pixel;
    r.russian_roulette = russian_roulette; r.direct_sampler = (uint32_t)direct_sampler; r.indirect_sampler = (uint32_t)indirect_sampler;
    r.integrator = (uint32_t)integrator; r.max_depth = max_depth; r.gamma = gamma; r.seed = seed;
    return r;
  }
};

struct Config::Builder {
  Config c;
  Builder& bvh(bool v) { c.bvh = v; return *this; }
  Builder& light_samples(uint32_t v) { c.light_samples = v; return *this; }
  Builder& light_stratify(bool v) { c.light_stratify = v; return *this; }
  Builder& samples_per_pixel(uint32_t v) { c.samples_per_pixel = v; return *this; }
  Builder& russian_roulette(bool v) { c.russian_roulette = v; return *this; }
  Builder& direct_sampler(SamplerStrategy v) { c.direct_sampler = v; return *this; }
  Builder& indirect_sampler(SamplerStrategy v) { c.indirect_sampler = v; return *this; }
  Builder& integrator(IntegratorStrategy v) { c.integrator = v; return *this; }
  Builder& max_depth(uint32_t v) { c.max_depth = v; return *this; }
  Builder& gamma(float v) { c.gamma = v; return *this; }
  Builder& seed(uint32_t v) { c.seed = v; return *this; }
  Config build() const { return c; }
};
inline Config::Builder Config::builder() { return Builder(); }

// Scene -> RaycaSceneDesc.  Serialisation only, in the order SceneDrawInfo::traverse_scene walks the graph
// (rayca-soft/src/scene.rs:206-282): DFS pre-order, a scene node's model (root, then its subtree) before the
// node's own children.  The arrays own the memory `desc` points into.
struct FlatScene {
  std::vector<RaycaNode> nodes;
  std::vector<RaycaMesh> meshes;
  std::vector<RaycaPrimitive> primitives;
  std::vector<float> positions, colors, normals, tangents, bitangents, uvs;
  std::vector<uint8_t> index_bytes, image_bytes;
  std::vector<RaycaMaterial> materials;
  std::vector<RaycaTexture> textures;
  std::vector<RaycaImage> images;
  std::vector<RaycaCamera> cameras;
  std::vector<RaycaLight> lights;

  explicit FlatScene(const Scene& scene) {
    std::vector<Base> bases(scene.models.len());
    const uint32_t root = emit(scene.root.trs, -1, RAYCA_NONE, nullptr, nullptr);
    for (Handle<Node> c : scene.root.children) walk_scene_node(scene, c, (int32_t)root, bases);
  }

  RaycaSceneDesc desc() const {
    RaycaSceneDesc d;
    std::memset(&d, 0, sizeof d);
    d.abi_version = RAYCA_ABI_VERSION;
    d.nodes = nodes.data(); d.node_count = (uint32_t)nodes.size();
    d.meshes = meshes.data(); d.mesh_count = (uint32_t)meshes.size();
    d.primitives = primitives.data(); d.primitive_count = (uint32_t)primitives.size();
    d.vertex_count = (uint32_t)(positions.size() / 3);
    d.positions = positions.data(); d.colors = colors.data(); d.normals = normals.data();
    d.tangents = tangents.data(); d.bitangents = bitangents.data(); d.uvs = uvs.data();
    d.index_bytes = index_bytes.data(); d.index_byte_count = index_bytes.size();
    d.materials = materials.data(); d.material_count = (uint32_t)materials.size();
    d.textures = textures.data(); d.texture_count = (uint32_t)textures.size();
    d.images = images.data(); d.image_count = (uint32_t)images.size();
    d.image_bytes = image_bytes.data(); d.image_byte_count = image_bytes.size();
    d.cameras = cameras.data(); d.camera_count = (uint32_t)cameras.size();
    d.lights = lights.data(); d.light_count = (uint32_t)lights.size();
    return d;
  }

 private:
  struct Base {
    bool done = false;
    uint32_t mesh = 0, material = 0, texture = 0, image = 0, camera = 0, light = 0;
  };
  static RaycaTrs to_abi(const Trs& t) {
    RaycaTrs r;
    r.translation[0] = t.translation.x; r.translation[1] = t.translation.y; r.translation[2] = t.translation.z;
    r.rotation[0] = t.rotation.x; r.rotation[1] = t.rotation.y; r.rotation[2] = t.rotation.z; r.rotation[3] = t.rotation.w;
    r.scale[0] = t.scale.x; r.scale[1] = t.scale.y; r.scale[2] = t.scale.z;
    return r;
  }
  static void put4(float* d, const Color& c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; d[3] = c.a; }
  static void put3(float* d, const Vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

  uint32_t emit(const Trs& trs, int32_t parent, uint32_t model, const Node* payload, const Base* base) {
    RaycaNode r;
    std::memset(&r, 0, sizeof r);
    r.parent = parent;
    r.model = model;
    r.trs = to_abi(trs);
    r.mesh = (payload && base && payload->mesh.is_valid()) ? base->mesh + payload->mesh.id : RAYCA_NONE;
    r.camera = (payload && base && payload->camera.is_valid()) ? base->camera + payload->camera.id : RAYCA_NONE;
    r.light = (payload && base && payload->light.is_valid()) ? base->light + payload->light.id : RAYCA_NONE;
    nodes.push_back(r);
    return (uint32_t)nodes.size() - 1u;
  }

  static RaycaMaterial material_to_abi(const Material& m, uint32_t tex_base) {
    RaycaMaterial r;
    std::memset(&r, 0, sizeof r);
    r.albedo_texture = r.normal_texture = r.metallic_roughness_texture = RAYCA_NONE;
    put4(r.color, Color::white());
    put4(r.ambient, Color::black()); put4(r.emission, Color::black()); put4(r.diffuse, Color::black()); put4(r.specular, Color::black());
    r.roughness_factor = 1.0f;
    auto tex = [tex_base](Handle<Texture> h) { return h.is_valid() ? tex_base + h.id : RAYCA_NONE; };
    if (const PbrMaterial* p = std::get_if<PbrMaterial>(&m.value)) {
      r.kind = RAYCA_MATERIAL_PBR;
      put4(r.color, p->color);
      r.albedo_texture = tex(p->albedo); r.normal_texture = tex(p->normal); r.metallic_roughness_texture = tex(p->metallic_roughness);
      r.metallic_factor = p->metallic_factor; r.roughness_factor = p->roughness_factor;
    } else if (const PhongMaterial* p = std::get_if<PhongMaterial>(&m.value)) {
      r.kind = RAYCA_MATERIAL_PHONG;
      put4(r.ambient, p->ambient); put4(r.emission, p->emission); put4(r.diffuse, p->diffuse); put4(r.specular, p->specular);
      r.shininess = p->shininess;
    } else {
      const GgxMaterial& g = std::get<GgxMaterial>(m.value);
      r.kind = RAYCA_MATERIAL_GGX;
      put4(r.diffuse, g.diffuse); put4(r.specular, g.specular);
      r.roughness_factor = g.roughness;
    }
    return r;
  }

  const Base& add_model_payload(const Scene& scene, Handle<Model> mh, std::vector<Base>& bases) {
    Base& base = bases[mh.id];
    if (base.done) return base;
    const Model& model = scene.models.items[mh.id];
    base.mesh = (uint32_t)meshes.size(); base.material = (uint32_t)materials.size(); base.texture = (uint32_t)textures.size();
    base.image = (uint32_t)images.size(); base.camera = (uint32_t)cameras.size(); base.light = (uint32_t)lights.size();
    for (const Image& im : model.images.items) {
      RaycaImage r;
      std::memset(&r, 0, sizeof r);
      r.width = im.w; r.height = im.h; r.color_type = (uint32_t)im.color_type; r.byte_offset = image_bytes.size();
      image_bytes.insert(image_bytes.end(), im.data.begin(), im.data.end());
      images.push_back(r);
    }
    for (const Texture& t : model.textures.items) textures.push_back(RaycaTexture{base.image + t.image.id});
    for (const Material& m : model.materials.items) materials.push_back(material_to_abi(m, base.texture));
    for (const Camera& c : model.cameras.items) cameras.push_back(RaycaCamera{c.yfov_radians});
    for (const Light& l : model.lights.items) {
      RaycaLight r;
      std::memset(&r, 0, sizeof r);
      r.material = RAYCA_NONE;
      r.attenuation[2] = 1.0f;
      if (const PointLight* p = std::get_if<PointLight>(&l.value)) {
        r.kind = RAYCA_LIGHT_POINT; r.intensity = p->intensity; put4(r.color, p->color); put3(r.attenuation, p->attenuation);
      } else if (const QuadLight* q = std::get_if<QuadLight>(&l.value)) {
        r.kind = RAYCA_LIGHT_QUAD; r.intensity = q->intensity; put4(r.color, q->color); put3(r.ab, q->ab); put3(r.ac, q->ac);
        r.material = q->material.is_valid() ? base.material + q->material.id : RAYCA_NONE;
      } else {
        const DirectionalLight& d = std::get<DirectionalLight>(l.value);
        r.kind = RAYCA_LIGHT_DIRECTIONAL; r.intensity = d.intensity; put4(r.color, d.color);
      }
      lights.push_back(r);
    }
    for (const Mesh& mesh : model.meshes.items) {
      RaycaMesh rm{(uint32_t)primitives.size(), (uint32_t)mesh.primitives.size()};
      for (Handle<Primitive> ph : mesh.primitives) {
        const Primitive* p = model.primitives.get(ph);
        if (!p) throw Error(RAYCA_ERR_BAD_ARG, "mesh refers to a primitive that does not exist");
        const Geometry* g = model.geometries.get(p->geometry);
        if (!g) throw Error(RAYCA_ERR_BAD_ARG, "primitive refers to a geometry that does not exist");
        RaycaPrimitive rp;
        std::memset(&rp, 0, sizeof rp);
        rp.material = p->material.is_valid() ? base.material + p->material.id : RAYCA_NONE;
        if (const Sphere* s = std::get_if<Sphere>(&g->value)) {
          rp.geometry = RAYCA_GEOMETRY_SPHERE;
          put3(rp.sphere_center, s->center);
          rp.sphere_radius = s->radius;
          rp.index_type = RAYCA_INDEX_U32;
        } else {
          const TriangleMesh& t = std::get<TriangleMesh>(g->value);
          rp.geometry = RAYCA_GEOMETRY_TRIANGLE_MESH;
          rp.first_vertex = (uint32_t)(positions.size() / 3);
          rp.vertex_count = (uint32_t)t.vertices.size();
          rp.index_type = (uint32_t)t.indices.index_type;
          while (index_bytes.size() % 4) index_bytes.push_back(0);
          rp.index_byte_offset = index_bytes.size();
          rp.index_count = t.indices.get_index_count();
          index_bytes.insert(index_bytes.end(), t.indices.indices.begin(), t.indices.indices.end());
          for (const Vertex& v : t.vertices) {
            float f[4];
            put3(f, v.pos); positions.insert(positions.end(), f, f + 3);
            put4(f, v.ext.color); colors.insert(colors.end(), f, f + 4);
            put3(f, v.ext.normal); normals.insert(normals.end(), f, f + 3);
            put3(f, v.ext.tangent); tangents.insert(tangents.end(), f, f + 3);
            put3(f, v.ext.bitangent); bitangents.insert(bitangents.end(), f, f + 3);
            uvs.push_back(v.ext.uv.x); uvs.push_back(v.ext.uv.y);
          }
        }
        primitives.push_back(rp);
      }
      meshes.push_back(rm);
    }
    base.done = true;
    return base;
  }

  void walk_model_node(const Model& model, Handle<Model> mh, Handle<Node> nh, int32_t parent, const Base& base) {
    const Node* node = model.nodes.get(nh);
    if (!node) throw Error(RAYCA_ERR_BAD_ARG, "model node handle out of range");
    const uint32_t me = emit(node->trs, parent, mh.id, node, &base);
    for (Handle<Node> c : node->children) walk_model_node(model, mh, c, (int32_t)me, base);
  }
  void walk_scene_node(const Scene& scene, Handle<Node> nh, int32_t parent, std::vector<Base>& bases) {
    const Node* node = scene.nodes.get(nh);
    if (!node) throw Error(RAYCA_ERR_BAD_ARG, "scene node handle out of range");
    const uint32_t me = emit(node->trs, parent, RAYCA_NONE, nullptr, nullptr);
    if (node->model.is_valid()) {
      const Model* model = scene.models.get(node->model);
      if (!model) throw Error(RAYCA_ERR_BAD_ARG, "scene node refers to a model that does not exist");
      const Base& base = add_model_payload(scene, node->model, bases);
      const uint32_t mroot = emit(model->root.trs, (int32_t)me, node->model.id, nullptr, nullptr);
      for (Handle<Node> c : model->root.children) walk_model_node(*model, node->model, c, (int32_t)mroot, base);
    }
    for (Handle<Node> c : node->children) walk_scene_node(scene, c, (int32_t)me, bases);
  }
};

inline void check(int32_t rc) {
  if (rc != RAYCA_OK) {
    char buf[512];
    rayca_hip_last_error(buf, sizeof buf);
    throw Error(rc, std::string("rayca_hip error ") + std::to_string(rc) + ": " + buf);
  }
}

// A scene kept resident on the device (no reference counterpart: SoftRenderer::draw rebuilds
// SceneDrawInfo, BvhScene and the Tlas on every call, scene.rs:90-99; a viewer keeps one of these).
class DeviceScene {
 public:
  DeviceScene(const Scene& scene, const Config& config, uint32_t builder = RAYCA_BUILDER_SAH, uint32_t device = 0) {
    const FlatScene flat(scene);
    const RaycaSceneDesc d = flat.desc();
    const RaycaConfig c = config.to_abi();
    RaycaBuildOptions o;
    std::memset(&o, 0, sizeof o);
    o.builder = builder;
    o.device = device;
    check(rayca_hip_scene_create(&d, &c, &o, &handle_));
  }
  ~DeviceScene() {
    if (handle_) rayca_hip_scene_destroy(handle_);
  }
  DeviceScene(const DeviceScene&) = delete;
  DeviceScene& operator=(const DeviceScene&) = delete;
  RaycaScene* handle() const { return handle_; }
  RaycaStats draw(const Config& config, Image& image, const RaycaRenderOptions* opts = nullptr) {
    const RaycaConfig c = config.to_abi();
    RaycaStats st;
    std::memset(&st, 0, sizeof st);
    if (image.color_type == ColorType::RGBA8) check(rayca_hip_render(handle_, &c, image.w, image.h, opts, image.bytes_mut(), nullptr, &st));
    else if (image.color_type == ColorType::RGBA32F)
      check(rayca_hip_render(handle_, &c, image.w, image.h, opts, nullptr, reinterpret_cast<float*>(image.bytes_mut()), &st));
    else throw Error(RAYCA_ERR_BAD_ARG, "draw needs an RGBA8 or RGBA32F image");
    return st;
  }

 private:
  RaycaScene* handle_ = nullptr;
};

// trait Draw (rayca-soft/src/draw.rs:7-9) and its implementation (rayca-soft/src/scene.rs:88-154).
struct Draw {
  virtual ~Draw() = default;
  virtual void draw(const Scene& scene, Image& image) = 0;
};

struct SoftRenderer : Draw {
  Config config;
  SoftRenderer() = default;
  static SoftRenderer new_with_config(Config c) {  // scene.rs:57-61
    SoftRenderer r;
    r.config = c;
    return r;
  }
  // SoftRenderer::create_default_model  scene.rs:18-55: camera at (0,0,4), two nodes sharing one point
  // light of intensity 1024 at (-1,4,3) and (1,4,3)
  static Model create_default_model() {
    Model model;
    const Handle<Camera> camera = model.cameras.push(Camera());
    const Handle<Node> cn = model.nodes.push(Node::builder().camera(camera).trs(Trs::builder().translation(Vec3(0, 0, 4)).build()).build());
    model.root.children.push_back(cn);
    Light light = Light::point();
    light.set_intensity(1024.0f);
    const Handle<Light> lh = model.lights.push(light);
    const Handle<Node> l0 = model.nodes.push(Node::builder().trs(Trs::builder().translation(Vec3(-1, 4, 3)).build()).light(lh).build());
    model.root.children.push_back(l0);
    const Handle<Node> l1 = model.nodes.push(Node::builder().light(lh).trs(Trs::builder().translation(Vec3(1, 4, 3)).build()).build());
    model.root.children.push_back(l1);
    return model;
  }
  // Like the reference, every call flattens the scene, builds the acceleration structure and drops it.
  // The tree is the reference's own (RAYCA_BUILDER_REFERENCE) so that depth ties resolve as they do there.
  void draw(const Scene& scene, Image& image) override {
    DeviceScene resident(scene, config, RAYCA_BUILDER_REFERENCE);
    resident.draw(config, image);
  }
};

// ---- PNG write-out (Image::dump_png image.rs:160-169): zlib "stored" blocks, CRC-32, Adler-32 --------
namespace detail {
inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}
inline void be32(std::vector<uint8_t>& o, uint32_t v) {
  o.push_back((uint8_t)(v >> 24)); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v);
}
inline void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& body) {
  be32(out, (uint32_t)body.size());
  std::vector<uint8_t> tb(type, type + 4);
  tb.insert(tb.end(), body.begin(), body.end());
  out.insert(out.end(), tb.begin(), tb.end());
  be32(out, crc32(tb.data(), tb.size()));
}
}  // namespace detail

inline void Image::dump_png(const std::string& path) const {
  if (color_type == ColorType::RGBA32F) throw Error(RAYCA_ERR_BAD_ARG, "dump_png needs an 8-bit image");
  const size_t bpp = texel_size(), stride = (size_t)w * bpp;
  std::vector<uint8_t> raw;
  raw.reserve((stride + 1) * h);
  for (uint32_t y = 0; y < h; ++y) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), data.begin() + (size_t)y * stride, data.begin() + (size_t)(y + 1) * stride);
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) {
    a = (a + v) % 65521u;
    b = (b + a) % 65521u;
  }
  size_t pos = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  detail::be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  detail::be32(ihdr, w);
  detail::be32(ihdr, h);
  ihdr.push_back(8);
  ihdr.push_back(color_type == ColorType::RGBA8 ? 6 : 2);
  ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  detail::chunk(out, "IHDR", ihdr);
  detail::chunk(out, "IDAT", z);
  detail::chunk(out, "IEND", {});
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw Error(RAYCA_ERR_BAD_ARG, "cannot open " + path);
  std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
}

}  // namespace rayca
