// host_mirror.cpp -- the reference's own scene-building tests (rayca-soft/tests/gltf.rs:10-83: `sphere`,
// `triangle`), written against include/rayca.hpp, plus a cube scene built from TriangleMesh::cube().
//
//   host_mirror describe <scene> <out_dir>     flatten only, dump every array of the RaycaSceneDesc (no GPU)
//   host_mirror draw <scene> <out.rgba> [png]  SoftRenderer::draw on device 0, raw RGBA8 (+ PNG)
//   host_mirror image <in.png|jpg> <out_dir>   decode a PNG / JPEG with the loader's decoders, dump w/h/type + texels
//   host_mirror sdtf_config <in.sdtf> <out_dir>  load an SDTF file, dump its SdtfConfig and the Config after Config::apply
// scene "gltf:<path>" = rayca-soft/tests/gltf.rs:191-204 `gltf::cube`: the file + create_default_model()
// scene "sdtf:<path>" = rayca-soft/tests/sdtf.rs:7-25 `run_test`: Scene::push_sdtf_from_path, nothing else
// scene "sdtfstr:<text>" = the loader's own unit tests (rayca-model/src/loader/sdtf.rs:913-947): SdtfBuilder::_str
//
// tests/test_cpp_host.py compares `describe` with rayca_amd.flatten of the same scene built through the
// Python mirror, and `draw` with the Python host's frame.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rayca.hpp"
#include "rayca_gltf.hpp"
#include "rayca_sdtf.hpp"

using namespace rayca;

// rayca-soft/tests/gltf.rs:49-83
static Scene triangle_scene() {
  Model model;
  TriangleMesh triangle = TriangleMesh::unit();
  triangle.vertices[0].ext.color = Color::from(0xFF0000FF);
  triangle.vertices[1].ext.color = Color::from(0x00FF00FF);
  triangle.vertices[2].ext.color = Color::from(0x0000FFFF);
  const auto geometry_handle = model.geometries.push(Geometry(triangle));
  const auto prim_handle = model.primitives.push(Primitive::builder().geometry(geometry_handle).build());
  const auto mesh_handle = model.meshes.push(Mesh::builder().primitive(prim_handle).build());
  const Node node = Node::builder()
                        .mesh(mesh_handle)
                        .trs(Trs::builder().translation(Vec3(0.0f, -1.0f, 0.0f)).scale(Vec3(1.0f, 2.0f, 1.0f)).build())
                        .build();
  const auto node_handle = model.nodes.push(node);
  model.root.children.push_back(node_handle);
  Scene scene;
  scene.push_model(std::move(model));
  scene.push_model(SoftRenderer::create_default_model());
  return scene;
}

// rayca-soft/tests/gltf.rs:10-46
static Scene sphere_scene() {
  Model model;
  const auto geometry_handle = model.geometries.push(Geometry(Sphere::unit()));
  const auto prim_handle = model.primitives.push(Primitive::builder().geometry(geometry_handle).build());
  const auto mesh_handle = model.meshes.push(Mesh::builder().primitive(prim_handle).build());
  const Node node = Node::builder()
                        .mesh(mesh_handle)
                        .trs(Trs::builder().translation(Vec3(0.0f, 0.0f, -1.0f)).scale(Vec3(1.0f, 2.0f, 1.0f)).build())
                        .build();
  model.root.children.push_back(model.nodes.push(node));
  Scene scene;
  scene.push_model(std::move(model));
  scene.push_model(SoftRenderer::create_default_model());
  return scene;
}

// a rotated, GGX-shaded cube over a Phong quad: exercises every material kind and a two-level node chain
static Scene cube_scene() {
  Model model;
  GgxMaterial ggx;
  ggx.diffuse = Color(0.7f, 0.3f, 0.2f, 1.0f);
  ggx.specular = Color(0.2f, 0.2f, 0.2f, 1.0f);
  ggx.roughness = 0.4f;
  const auto ggx_handle = model.materials.push(Material(ggx));
  PhongMaterial phong;
  phong.diffuse = Color(0.2f, 0.6f, 0.3f, 1.0f);
  const auto phong_handle = model.materials.push(Material(phong));
  const auto cube = model.geometries.push(Geometry(TriangleMesh::cube()));
  const auto quad = model.geometries.push(Geometry(TriangleMesh::quad(Vec2(2.0f, 2.0f))));
  const auto cube_prim = model.primitives.push(Primitive::builder().geometry(cube).material(ggx_handle).build());
  const auto quad_prim = model.primitives.push(Primitive::builder().geometry(quad).material(phong_handle).build());
  const auto cube_mesh = model.meshes.push(Mesh::builder().primitive(cube_prim).build());
  const auto quad_mesh = model.meshes.push(Mesh::builder().primitive(quad_prim).build());
  const auto cube_node = model.nodes.push(
      Node::builder().mesh(cube_mesh).trs(Trs::builder().rotation(Quat::axis_angle(Vec3(0.0f, 1.0f, 0.0f), 0.6f)).build()).build());
  const auto group = model.nodes.push(
      Node::builder().children({cube_node}).trs(Trs::builder().translation(Vec3(0.25f, 0.0f, -0.5f)).scale(Vec3::splat(1.2f)).build()).build());
  const auto floor = model.nodes.push(Node::builder()
                                          .mesh(quad_mesh)
                                          .trs(Trs::builder()
                                                   .translation(Vec3(0.0f, -0.8f, 0.0f))
                                                   .rotation(Quat::axis_angle(Vec3(1.0f, 0.0f, 0.0f), -1.5707964f))
                                                   .scale(Vec3::splat(6.0f))
                                                   .build())
                                          .build());
  model.root.children.push_back(group);
  model.root.children.push_back(floor);
  Scene scene;
  scene.push_model(std::move(model));
  scene.push_model(SoftRenderer::create_default_model());
  return scene;
}

// rayca-soft/tests/gltf.rs:191-204
static Scene gltf_scene(const std::string& path) {
  Scene scene;
  push_gltf_from_path(scene, path);
  scene.push_model(SoftRenderer::create_default_model());
  return scene;
}

static Scene sdtf_scene(const std::string& path) {
  Scene scene;
  push_sdtf_from_path(scene, path);
  return scene;
}

static Scene make(const std::string& name) {
  if (name.rfind("gltf:", 0) == 0) return gltf_scene(name.substr(5));
  if (name.rfind("sdtf:", 0) == 0) return sdtf_scene(name.substr(5));
  if (name.rfind("sdtfstr:", 0) == 0) {
    Scene scene;
    scene.push_model(load_sdtf_str(name.substr(8)).first);
    return scene;
  }
  if (name == "triangle") return triangle_scene();
  if (name == "sphere") return sphere_scene();
  if (name == "cube") return cube_scene();
  std::fprintf(stderr, "unknown scene %s\n", name.c_str());
  std::exit(2);
}

template <class T>
static void dump(const std::string& dir, const char* name, const std::vector<T>& v) {
  const std::string path = dir + "/" + name + ".bin";
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) std::exit(3);
  if (!v.empty()) std::fwrite(v.data(), sizeof(T), v.size(), f);
  std::fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: host_mirror describe|draw <scene> <out> [png]\n");
    return 2;
  }
  const std::string mode = argv[1], name = argv[2], out = argv[3];
  try {
    if (mode == "png" || mode == "image") {   // PNG or JPEG, by signature (rayca_gltf.hpp decode_image)
      const Image im = gltf_detail::decode_image(gltf_detail::read_file(name));
      const std::vector<uint32_t> head = {im.w, im.h, (uint32_t)im.color_type};
      dump(out, "png_head", head);
      dump(out, "png_texels", im.data);
      return 0;
    }
    if (mode == "sdtf_config") {
      const SdtfConfig sc = name.rfind("str:", 0) == 0 ? load_sdtf_str(name.substr(4)).second : load_sdtf_path(name).second;
      Config cfg = Config::builder().bvh(false).build();   // as rayca-soft/tests/sdtf.rs builds it
      apply(cfg, sc);
      float g[2] = {sc.gamma, cfg.gamma};
      uint32_t gb[2];
      std::memcpy(gb, g, sizeof gb);
      const std::vector<int64_t> v = {sc.width, sc.height, sc.max_depth, sc.light_samples, sc.light_stratify, sc.samples_per_pixel, (int64_t)sc.direct_sampler,
                                      sc.russian_roulette, (int64_t)sc.indirect_sampler, (int64_t)sc.integrator, (int64_t)sc.brdf, gb[0],
                                      cfg.bvh, cfg.light_samples, cfg.light_stratify, cfg.samples_per_pixel, cfg.russian_roulette, (int64_t)cfg.direct_sampler,
                                      (int64_t)cfg.indirect_sampler, (int64_t)cfg.integrator, cfg.max_depth, gb[1]};
      dump(out, "sdtf_config", v);
      return 0;
    }
    const Scene scene = make(name);
    if (mode == "describe") {
      const FlatScene flat(scene);
      dump(out, "nodes", flat.nodes);
      dump(out, "meshes", flat.meshes);
      dump(out, "primitives", flat.primitives);
      dump(out, "positions", flat.positions);
      dump(out, "colors", flat.colors);
      dump(out, "normals", flat.normals);
      dump(out, "tangents", flat.tangents);
      dump(out, "bitangents", flat.bitangents);
      dump(out, "uvs", flat.uvs);
      dump(out, "index_bytes", flat.index_bytes);
      dump(out, "materials", flat.materials);
      dump(out, "textures", flat.textures);
      dump(out, "images", flat.images);
      dump(out, "image_bytes", flat.image_bytes);
      dump(out, "cameras", flat.cameras);
      dump(out, "lights", flat.lights);
      return 0;
    }
    // the reference's `triangle` test verbatim from here: 256x256 RGBA8, default renderer
    Image image(256, 256, ColorType::RGBA8);
    // rayca-soft/tests/gltf.rs:12-18: the sphere test runs the Scratcher without a BVH
    SoftRenderer renderer = name == "sphere"
                                ? SoftRenderer::new_with_config(Config::builder().bvh(false).integrator(IntegratorStrategy::Scratcher).build())
                                : SoftRenderer();
    renderer.draw(scene, image);
    dump(".", out.c_str(), image.data);  // writes ./<out>.bin
    if (argc > 4) image.dump_png(argv[4]);
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
