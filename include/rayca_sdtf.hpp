// rayca_sdtf.hpp -- C++ host mirror of rayca-model's SDTF ("UCSD scene description") loader and of
// rayca_soft::Config::apply(SdtfConfig).
//
//   reference: rayca-model/src/loader/sdtf.rs      SdtfConfig :89-122, SdtfBuilder::parse_* :160-772,
//                                                  parse_line :774-830, process_material / process_primitive :833-870,
//                                                  build :872-899, Model::load_sdtf_path :902-907
//              rayca-model/src/scene.rs:126-136    Scene::push_sdtf_from_path
//              rayca-soft/src/config.rs:58-71      Config::apply   (maxdepth -1 -> 16)
//
// The step immediately before the hot path: text -> Model (+ the Config the scene asks for).  Host-side, f32, and the few
// pieces of arithmetic that end up in node transforms -- Trs::left_mul (trs.rs:111-118), Quat::axis_angle (quat.rs:67-77),
// Mat4::look_at -> Quat::from(&Mat4) -> get_inverse (mat4.rs:81-95, quat.rs:184-226,99-103), Quat::angle_between
// (quat.rs:118-127), the face normal of `tri` -- are written in the reference's operation order (4-lane sums left to
// right, no contraction: build with -ffp-contract=off like everything else), so that rayca_amd/sdtf.py, which restates them
// with numpy float32 scalars, flattens a file to the same bytes (tests/test_sdtf.py).
//
// Reference behaviour that is kept on purpose:
//   * a `tri` that follows a `sphere` without a material directive in between is DROPPED: the pending primitive is the
//     sphere and `if let Geometry::TriangleMesh` does not match (sdtf.rs:262-289)
//     (a `sphere` itself flushes the pending primitive first, sdtf.rs:312, so every sphere is kept)
//   * translate / rotate / scale without a pushTransform panic in the reference (`last_mut().unwrap()`, sdtf.rs:357): Error
//   * point lights take the `attenuation` current at the time (default (1,0,0): constant, sdtf.rs:146), intensity 1
//   * every primitive gets its OWN copy of the current material (no sharing, sdtf.rs:833-848)
//   * words are separated by spaces only (sdtf.rs:786); lines starting with '#' and lines without an alphanumeric
//     character are skipped; unknown commands are skipped with a warning
#pragma once

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "rayca.hpp"

namespace rayca {

enum class SdtfIntegratorStrategy { Raytracer, AnalyticDirect, Direct, Pathtracer };  // sdtf.rs:15-29
enum class SdtfSamplerStrategy { None, Nee, Hemisphere, Cosine, Brdf, Mis };          // sdtf.rs:46-54
enum class SdtfBrdfStrategy { Phong, Ggx };                                           // sdtf.rs:71-75

struct SdtfConfig {  // sdtf.rs:89-122
  uint32_t width = 0, height = 0;
  int32_t max_depth = 5;
  uint32_t light_samples = 1;
  bool light_stratify = false;
  uint32_t samples_per_pixel = 1;
  SdtfSamplerStrategy direct_sampler = SdtfSamplerStrategy::None;
  bool russian_roulette = false;
  SdtfSamplerStrategy indirect_sampler = SdtfSamplerStrategy::Hemisphere;
  SdtfIntegratorStrategy integrator = SdtfIntegratorStrategy::Raytracer;
  SdtfBrdfStrategy brdf = SdtfBrdfStrategy::Phong;
  float gamma = 1.0f;
};

// impl From<SdtfIntegratorStrategy> for IntegratorStrategy (integrator/mod.rs:74-84), From<SdtfSamplerStrategy> (sampler/mod.rs:94-107)
inline IntegratorStrategy to_integrator(SdtfIntegratorStrategy v) {
  switch (v) {
    case SdtfIntegratorStrategy::Raytracer: return IntegratorStrategy::Raytracer;
    case SdtfIntegratorStrategy::AnalyticDirect: return IntegratorStrategy::AnalyticDirect;
    case SdtfIntegratorStrategy::Direct: return IntegratorStrategy::Direct;
    default: return IntegratorStrategy::Pathtracer;
  }
}
inline SamplerStrategy to_sampler(SdtfSamplerStrategy v) {
  switch (v) {
    case SdtfSamplerStrategy::None: return SamplerStrategy::None;
    case SdtfSamplerStrategy::Nee: return SamplerStrategy::Nee;
    case SdtfSamplerStrategy::Hemisphere: return SamplerStrategy::Hemisphere;
    case SdtfSamplerStrategy::Cosine: return SamplerStrategy::Cosine;
    case SdtfSamplerStrategy::Brdf: return SamplerStrategy::Brdf;
    default: return SamplerStrategy::Mis;
  }
}

// Config::apply  rayca-soft/src/config.rs:58-71.  bvh and russian_roulette are NOT taken from the file (the reference
// does not copy them either).
inline void apply(Config& c, const SdtfConfig& s) {
  c.max_depth = s.max_depth == -1 ? 16u : (uint32_t)s.max_depth;
  c.light_samples = s.light_samples;
  c.light_stratify = s.light_stratify;
  c.samples_per_pixel = s.samples_per_pixel;
  c.direct_sampler = to_sampler(s.direct_sampler);
  c.indirect_sampler = to_sampler(s.indirect_sampler);
  c.integrator = to_integrator(s.integrator);
  c.gamma = s.gamma;
}

namespace sdtf_math {  // rayca-math, f32, the reference's operation order
struct V4 {
  float x, y, z, w;
};
inline float sum4(V4 a) { return (((-0.0f + a.x) + a.y) + a.z) + a.w; }  // f32x4::reduce_sum: ordered
inline V4 mul(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 add(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 sub(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 scale(float f, V4 a) { return {f * a.x, f * a.y, f * a.z, f * a.w}; }  // Mul<Vec3> for f32 / Mul<f32> for Vec3: lane * splat
inline float dot(V4 a, V4 b) { return sum4(mul(a, b)); }
inline V4 vec(Vec3 v) { return {v.x, v.y, v.z, 0.0f}; }
inline Vec3 xyz(V4 v) { return Vec3(v.x, v.y, v.z); }
inline V4 cross(V4 a, V4 b) {  // vec3.rs:134-142
  const V4 t0 = {a.y, a.z, a.x, a.w}, t1 = {b.z, b.x, b.y, b.w};
  const V4 t2 = mul(t0, b), t3 = mul(t0, t1);
  const V4 t4 = {t2.y, t2.z, t2.x, t2.w};
  return sub(t3, t4);
}
constexpr float kEps = 9.765625e-4f;  // rayca-math/src/lib.rs:33
inline V4 normalized(V4 a) {  // vec3.rs:183-193
  const float len = std::sqrt(dot(a, a));
  if (len > kEps) return {a.x / len, a.y / len, a.z / len, a.w / 1.0f};
  return a;
}
inline V4 rotate(V4 v, Quat q) {  // Vec3::rotate vec3.rs:148-159 (Mul<Vec3> for Quat quat.rs:266-273)
  const V4 u = {q.x * 1.0f, q.y * 1.0f, q.z * 1.0f, q.w * 0.0f};
  const float s = q.w;
  const V4 a = scale(2.0f * dot(u, v), u);
  const V4 b = scale(s * s - dot(u, u), v);
  const V4 c = scale(2.0f * s, cross(u, v));
  return add(add(a, b), c);
}
inline Quat qmul(Quat a, Quat b) {  // quat.rs:236-258
  return Quat(a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x, -a.x * b.z + a.y * b.w + a.z * b.x + a.w * b.y,
              a.x * b.y - a.y * b.x + a.z * b.w + a.w * b.z, -a.x * b.x - a.y * b.y - a.z * b.z + a.w * b.w);
}
inline float qlen(Quat q) { return std::sqrt(sum4({q.x * q.x, q.y * q.y, q.z * q.z, q.w * q.w})); }
inline Quat qnormalized(Quat q) {
  const float l = qlen(q);
  return Quat(q.x / l, q.y / l, q.z / l, q.w / l);
}
inline Quat qinverse(Quat q) {  // get_inverse quat.rs:99-103: asserts |len - 1| < 0.001, then conjugates
  if (!(std::fabs(qlen(q) - 1.0f) < 0.001f)) throw Error(RAYCA_ERR_BAD_ARG, "sdtf: inverse of a quaternion that is not normalised (quat.rs:101)");
  return Quat(q.x * -1.0f, q.y * -1.0f, q.z * -1.0f, q.w * 1.0f);
}
inline Quat axis_angle(Vec3 axis, float angle) {  // quat.rs:67-77
  const float factor = std::sin(angle / 2.0f);
  const V4 a = vec(axis);
  const Quat q(a.x * factor + 0.0f, a.y * factor + 0.0f, a.z * factor + 0.0f, a.w * factor + std::cos(angle / 2.0f));
  return qnormalized(q);
}
inline Quat angle_between(Vec3 a_, Vec3 b_) {  // quat.rs:118-127
  const V4 a = vec(a_), b = vec(b_);
  const V4 c = cross(a, b);
  const float w = std::sqrt(dot(a, a) * dot(b, b)) + dot(a, b);
  return qnormalized(Quat(c.x, c.y, c.z, w));
}
inline void left_mul(Trs& self, const Trs& rhs) {  // trs.rs:111-118
  const V4 t = add(vec(self.translation), rotate(mul(vec(self.scale), vec(rhs.translation)), self.rotation));
  const Quat r = qmul(self.rotation, rhs.rotation);
  const V4 s = rotate(mul(vec(self.scale), rotate(vec(rhs.scale), rhs.rotation)), qinverse(rhs.rotation));
  self.translation = xyz(t);
  self.rotation = r;
  self.scale = xyz(s);
}
// Mat4::look_at(target, eye, up).get_rotation()  mat4.rs:81-95,117-119 ; From<&Mat4> for Quat  quat.rs:184-226
inline Quat look_at_rotation(Vec3 target, Vec3 eye, Vec3 up) {
  const V4 z = normalized(sub(vec(eye), vec(target)));
  const V4 x = normalized(cross(vec(up), z));
  const V4 y = cross(z, x);
  const float m[3][3] = {{x.x, x.y, x.z}, {y.x, y.y, y.z}, {z.x, z.y, z.z}};
  Quat r;
  const float t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0f) {
    const float s = 0.5f / std::sqrt(t + 1.0f);
    r = Quat((m[2][1] - m[1][2]) * s, (m[0][2] - m[2][0]) * s, (m[1][0] - m[0][1]) * s, 0.25f / s);
  } else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) {
    const float s = 2.0f * std::sqrt(1.0f + m[0][0] - m[1][1] - m[2][2]);
    r = Quat(0.25f * s, (m[0][1] + m[1][0]) / s, (m[0][2] + m[2][0]) / s, (m[2][1] - m[1][2]) / s);
  } else if (m[1][1] > m[2][2]) {
    const float s = 2.0f * std::sqrt(1.0f + m[1][1] - m[0][0] - m[2][2]);
    r = Quat((m[0][1] + m[1][0]) / s, 0.25f * s, (m[1][2] + m[2][1]) / s, (m[0][2] - m[2][0]) / s);
  } else {
    const float s = 2.0f * std::sqrt(1.0f + m[2][2] - m[0][0] - m[1][1]);
    r = Quat((m[0][2] + m[2][0]) / s, (m[1][2] + m[2][1]) / s, 0.25f * s, (m[1][0] - m[0][1]) / s);
  }
  return qnormalized(r);
}
}  // namespace sdtf_math

// TriangleIndices::add_index  triangle.rs:267-295: the component type widens when index 256 / 65536 arrives
inline void add_index(TriangleIndices& t, size_t last_index) {
  auto expand = [&t]() {  // expand_index_size  triangle.rs:233-265
    if (t.index_type == ComponentType::U8) {
      std::vector<uint16_t> w(t.indices.begin(), t.indices.end());
      t = TriangleIndices::from_u16(w);
    } else if (t.index_type == ComponentType::U16) {
      std::vector<uint32_t> w(t.indices.size() / 2);
      for (size_t i = 0; i < w.size(); ++i) {
        uint16_t v;
        std::memcpy(&v, &t.indices[2 * i], 2);
        w[i] = v;
      }
      t = TriangleIndices::from_u32(w);
    }
  };
  if (t.index_type == ComponentType::U8 && last_index == 256u) expand();
  else if (t.index_type == ComponentType::U16 && last_index == 65536u) expand();
  if (t.index_type == ComponentType::U8) {
    t.indices.push_back((uint8_t)last_index);
  } else if (t.index_type == ComponentType::U16) {
    const uint16_t v = (uint16_t)last_index;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    t.indices.insert(t.indices.end(), p, p + 2);
  } else {
    const uint32_t v = (uint32_t)last_index;
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    t.indices.insert(t.indices.end(), p, p + 4);
  }
}

class SdtfBuilder {  // sdtf.rs:124-147
 public:
  SdtfBuilder& path(const std::string& p) { path_ = p; has_path_ = true; return *this; }
  SdtfBuilder& str(const std::string& s) { string_ = s; has_string_ = true; return *this; }

  std::pair<Model, SdtfConfig> build() {  // sdtf.rs:872-899
    Model model;
    if (has_string_) {
      std::istringstream in(string_);
      parse_stream(in, model);
    } else if (has_path_) {
      std::ifstream in(path_);
      if (!in) throw Error(RAYCA_ERR_BAD_ARG, "Loading UCSD scene from: " + path_ + ": cannot open");
      parse_stream(in, model);
    } else {
      throw Error(RAYCA_ERR_BAD_ARG, "No path or string provided to load UCSD scene");
    }
    process_primitive(model);
    return {std::move(model), config_};
  }

 private:
  using Words = std::vector<std::string>;
  std::string path_, string_;
  bool has_path_ = false, has_string_ = false;
  std::vector<Vertex> vertices_;
  std::vector<Trs> transform_;
  PhongMaterial temp_phong_;
  GgxMaterial temp_ggx_;
  Vec3 attenuation_ = Vec3(1.0f, 0.0f, 0.0f);
  // the pending primitive (temp_model.primitives[0] + its geometry in the reference)
  bool pending_ = false;
  Geometry pending_geometry_ = Geometry(Sphere());
  SdtfConfig config_;

  [[noreturn]] static void bad(const std::string& what) { throw Error(RAYCA_ERR_BAD_ARG, "sdtf: " + what); }
  static const std::string& word(const Words& w, size_t i, const char* what) {
    if (i >= w.size()) bad(std::string("Failed to read ") + what);  // .expect(...) in the reference
    return w[i];
  }
  static float f32(const Words& w, size_t i, const char* what) {  // str::parse::<f32>: correctly rounded, whole word
    const std::string& s = word(w, i, what);
    char* end = nullptr;
    const float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str() || *end != '\0') bad("invalid float literal `" + s + "`");
    return v;
  }
  static long long integer(const Words& w, size_t i, const char* what) {
    const std::string& s = word(w, i, what);
    char* end = nullptr;
    const long long v = std::strtoll(s.c_str(), &end, 10);
    if (end == s.c_str() || *end != '\0') bad("invalid digit found in string `" + s + "`");
    return v;
  }
  static uint32_t u32(const Words& w, size_t i, const char* what) {
    const long long v = integer(w, i, what);
    if (v < 0 || v > 0xFFFFFFFFll) bad(std::string("number out of range for ") + what);
    return (uint32_t)v;
  }
  static Vec3 vec3(const Words& w, size_t i, const char* what) { return Vec3(f32(w, i, what), f32(w, i + 1, what), f32(w, i + 2, what)); }
  static Color rgb(const Words& w, size_t i, const char* what) { return Color(f32(w, i, what), f32(w, i + 1, what), f32(w, i + 2, what), 1.0f); }

  void parse_stream(std::istream& in, Model& model) {
    std::string line;
    while (std::getline(in, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();  // BufRead::lines strips "\r\n"
      parse_line(line, model);
    }
  }

  Trs& current_transform() {
    if (transform_.empty()) bad("translate / rotate / scale outside pushTransform (the reference panics: sdtf.rs:357)");
    return transform_.back();
  }

  void parse_line(const std::string& line, Model& model) {  // sdtf.rs:774-830
    if (!line.empty() && line[0] == '#') return;
    bool any = false;
    for (unsigned char ch : line) any = any || std::isalnum(ch) || ch >= 0x80;  // char::is_alphanumeric (non-ASCII letters count)
    if (!any) return;
    Words w;  // line.split(' ').filter(|w| !w.is_empty())
    size_t pos = 0;
    while (pos <= line.size()) {
      const size_t sp = line.find(' ', pos);
      const size_t end = sp == std::string::npos ? line.size() : sp;
      if (end > pos) w.push_back(line.substr(pos, end - pos));
      if (sp == std::string::npos) break;
      pos = sp + 1;
    }
    if (w.empty()) return;
    const std::string& c = w[0];
    if (c == "size") {
      config_.width = u32(w, 1, "width");
      config_.height = u32(w, 2, "height");
    } else if (c == "camera") parse_camera(w, model);
    else if (c == "maxverts") vertices_.reserve((size_t)u32(w, 1, "max verts"));
    else if (c == "vertex") vertices_.push_back(Vertex(f32(w, 1, "vertex x"), f32(w, 2, "vertex y"), f32(w, 3, "vertex z")));
    else if (c == "tri") parse_tri(w);
    else if (c == "ambient") { process_primitive(model); temp_phong_.ambient = rgb(w, 1, "ambient"); }
    else if (c == "sphere") parse_sphere(w, model);
    else if (c == "translate") {
      Trs t;
      t.translation = vec3(w, 1, "translation");
      sdtf_math::left_mul(current_transform(), t);
    } else if (c == "rotate") {
      const Vec3 axis = vec3(w, 1, "rotate");
      const float degrees = f32(w, 4, "rotate angle");
      Trs t;
      t.rotation = sdtf_math::axis_angle(axis, degrees * (3.14159265358979323846f / 180.0f));  // f32::to_radians
      sdtf_math::left_mul(current_transform(), t);
    } else if (c == "scale") {
      Trs t;
      t.scale = vec3(w, 1, "scale");
      sdtf_math::left_mul(current_transform(), t);
    } else if (c == "pushTransform") transform_.push_back(Trs());
    else if (c == "popTransform") {
      process_primitive(model);
      if (!transform_.empty()) transform_.pop_back();
    } else if (c == "emission") { process_primitive(model); temp_phong_.emission = rgb(w, 1, "emission"); }
    else if (c == "diffuse") { process_primitive(model); temp_phong_.diffuse = temp_ggx_.diffuse = rgb(w, 1, "diffuse"); }
    else if (c == "specular") { process_primitive(model); temp_phong_.specular = temp_ggx_.specular = rgb(w, 1, "specular"); }
    else if (c == "shininess") { process_primitive(model); temp_phong_.shininess = f32(w, 1, "shininess"); }
    else if (c == "roughness") { process_primitive(model); temp_ggx_.roughness = f32(w, 1, "roughness"); }
    else if (c == "brdf") {
      process_primitive(model);
      const std::string& b = word(w, 1, "brdf");
      if (b == "phong") config_.brdf = SdtfBrdfStrategy::Phong;
      else if (b == "ggx") config_.brdf = SdtfBrdfStrategy::Ggx;
      else bad("Failed to find a BRDF for `" + b + "`");
    } else if (c == "point") {  // sdtf.rs:503-547
      PointLight light;
      const Vec3 at = vec3(w, 1, "point light position");
      light.color = rgb(w, 4, "point light colour");
      light.attenuation = attenuation_;
      const Handle<Light> lh = model.lights.push(Light(light));
      const Handle<Node> nh = model.nodes.push(Node::builder().trs(Trs::builder().translation(at).build()).light(lh).build());
      model.root.children.push_back(nh);
    } else if (c == "directional") {  // sdtf.rs:549-601
      const Vec3 d = vec3(w, 1, "light direction");
      DirectionalLight light;
      light.color = rgb(w, 4, "directional light colour");
      light.intensity = 1.0f;
      const Handle<Light> lh = model.lights.push(Light(light));
      const Quat rot = sdtf_math::angle_between(Vec3(1.0f, 0.0f, 0.0f), Vec3(-d.x, -d.y, -d.z));
      const Handle<Node> nh = model.nodes.push(Node::builder().trs(Trs::builder().rotation(rot).build()).light(lh).build());
      model.root.children.push_back(nh);
    } else if (c == "attenuation") attenuation_ = vec3(w, 1, "attenuation");
    else if (c == "maxdepth") {
      const long long v = integer(w, 1, "maxdepth");
      if (v < INT32_MIN || v > INT32_MAX) bad("number out of range for maxdepth");
      config_.max_depth = (int32_t)v;
    } else if (c == "integrator") {
      const std::string& s = word(w, 1, "integrator");
      if (s == "raytracer") config_.integrator = SdtfIntegratorStrategy::Raytracer;
      else if (s == "analyticdirect") config_.integrator = SdtfIntegratorStrategy::AnalyticDirect;
      else if (s == "direct") config_.integrator = SdtfIntegratorStrategy::Direct;
      else if (s == "pathtracer") config_.integrator = SdtfIntegratorStrategy::Pathtracer;
      else bad("Failed to find an integrator for `" + s + "`");
    } else if (c == "quadLight") {  // sdtf.rs:627-706
      const Vec3 a = vec3(w, 1, "quad light a"), ab = vec3(w, 4, "quad light ab"), ac = vec3(w, 7, "quad light ac");
      const Color color = rgb(w, 10, "quad light color");
      PhongMaterial emissive;
      emissive.emission = color;
      const Handle<Material> mh = model.materials.push(Material(emissive));
      const Handle<Light> lh = model.lights.push(Light(QuadLight(ab, ac, color, mh)));
      const Handle<Node> nh = model.nodes.push(Node::builder().trs(Trs::builder().translation(a).build()).light(lh).build());
      model.root.children.push_back(nh);
    } else if (c == "lightsamples") config_.light_samples = u32(w, 1, "light samples");
    else if (c == "lightstratify") config_.light_stratify = word(w, 1, "light_stratify") == "on";
    else if (c == "spp") config_.samples_per_pixel = u32(w, 1, "spp");
    else if (c == "nexteventestimation") config_.direct_sampler = sampler(word(w, 1, "nexteventestimation"));
    else if (c == "russianroulette") config_.russian_roulette = word(w, 1, "russianroulette") == "on";
    else if (c == "importancesampling") config_.indirect_sampler = sampler(word(w, 1, "importancesampling"));
    else if (c == "gamma") config_.gamma = f32(w, 1, "gamma");
    // anything else: "Skipping command" (a log line in the reference)
  }

  static SdtfSamplerStrategy sampler(const std::string& s) {  // sdtf.rs:56-69
    if (s == "on") return SdtfSamplerStrategy::Nee;
    if (s == "mis") return SdtfSamplerStrategy::Mis;
    if (s == "hemisphere") return SdtfSamplerStrategy::Hemisphere;
    if (s == "cosine") return SdtfSamplerStrategy::Cosine;
    if (s == "brdf") return SdtfSamplerStrategy::Brdf;
    bad("Failed to find a sampler for `" + s + "`");
  }

  void parse_camera(const Words& w, Model& model) {  // sdtf.rs:173-229
    const Vec3 eye = vec3(w, 1, "camera"), target = vec3(w, 4, "camera target"), up = vec3(w, 7, "camera up");
    const float yfov_degrees = f32(w, 10, "camera fov");
    const float yfov_radians = yfov_degrees * 3.14159265358979323846f / 180.0f;
    Camera camera;  // Camera::infinite_perspective(1.0, yfov, 0.1): only yfov reaches the hot path
    camera.yfov_radians = yfov_radians;
    const Handle<Camera> ch = model.cameras.push(camera);
    const Quat rotation = sdtf_math::qinverse(sdtf_math::look_at_rotation(target, eye, up));
    const Handle<Node> nh = model.nodes.push(Node::builder().camera(ch).trs(Trs::builder().translation(eye).rotation(rotation).build()).build());
    model.root.children.push_back(nh);
  }

  void parse_tri(const Words& w) {  // sdtf.rs:247-294
    const uint32_t ia = u32(w, 1, "vertex x"), ib = u32(w, 2, "vertex y"), ic = u32(w, 3, "vertex z");
    if (!pending_) {
      pending_geometry_ = Geometry(TriangleMesh());
      pending_ = true;
    }
    TriangleMesh* mesh = std::get_if<TriangleMesh>(&pending_geometry_.value);
    if (!mesh) return;  // the pending primitive is a sphere: the triangle is dropped (`if let` does not match)
    if (ia >= vertices_.size() || ib >= vertices_.size() || ic >= vertices_.size()) bad("tri refers to a vertex that does not exist");
    const size_t last = mesh->vertices.size();
    add_index(mesh->indices, last);
    add_index(mesh->indices, last + 1);
    add_index(mesh->indices, last + 2);
    Vertex a = vertices_[ia], b = vertices_[ib], c = vertices_[ic];
    using namespace sdtf_math;
    // Point3 - Point3 = Vec3 (w: 1 - 1 = 0)
    const V4 ab = sub({b.pos.x, b.pos.y, b.pos.z, 1.0f}, {a.pos.x, a.pos.y, a.pos.z, 1.0f});
    const V4 ac = sub({c.pos.x, c.pos.y, c.pos.z, 1.0f}, {a.pos.x, a.pos.y, a.pos.z, 1.0f});
    const Vec3 n = xyz(normalized(cross(ab, ac)));
    a.ext.normal = b.ext.normal = c.ext.normal = n;
    mesh->vertices.push_back(a);
    mesh->vertices.push_back(b);
    mesh->vertices.push_back(c);
  }

  void parse_sphere(const Words& w, Model& model) {  // sdtf.rs:310-336
    process_primitive(model);
    const Vec3 center = vec3(w, 1, "center");
    const float radius = f32(w, 4, "radius");
    if (!pending_) {
      pending_geometry_ = Geometry(Sphere(center, radius));
      pending_ = true;
    }
  }

  Handle<Material> process_material(Model& model) const {  // sdtf.rs:833-846
    return config_.brdf == SdtfBrdfStrategy::Phong ? model.materials.push(Material(temp_phong_)) : model.materials.push(Material(temp_ggx_));
  }

  void process_primitive(Model& model) {  // sdtf.rs:849-870
    if (!pending_) return;
    pending_ = false;
    Primitive primitive;
    primitive.geometry = model.geometries.push(std::move(pending_geometry_));
    primitive.material = process_material(model);
    const Handle<Primitive> ph = model.primitives.push(primitive);
    const Handle<Mesh> mh = model.meshes.push(Mesh::builder().primitive(ph).build());
    Trs trs;
    for (const Trs& t : transform_) sdtf_math::left_mul(trs, t);
    const Handle<Node> nh = model.nodes.push(Node::builder().trs(trs).mesh(mh).build());
    model.root.children.push_back(nh);
  }
};

// Model::load_sdtf_path  sdtf.rs:902-907
inline std::pair<Model, SdtfConfig> load_sdtf_path(const std::string& path) { return SdtfBuilder().path(path).build(); }
inline std::pair<Model, SdtfConfig> load_sdtf_str(const std::string& text) { return SdtfBuilder().str(text).build(); }

// Scene::push_sdtf_from_path  rayca-model/src/scene.rs:126-136
inline std::pair<Handle<Node>, SdtfConfig> push_sdtf_from_path(Scene& scene, const std::string& path) {
  auto loaded = load_sdtf_path(path);
  return {scene.push_model(std::move(loaded.first)), loaded.second};
}

}  // namespace rayca
