// rayca_gltf.hpp -- glTF 2.0 ingestion for the C++ host mirror (include/rayca.hpp): the step right before the hot
// path.  Restates what the reference's loader accepts (rayca-model/src/loader/gltf.rs:56-578):
//   * .gltf JSON with `data:application/octet-stream;base64,` or external .bin buffers           (:73-99)
//   * f32 VEC2/VEC3/VEC4 attributes through bufferViews with byteStride                             (:121-258)
//   * u8/u16/u32 indices kept byte-packed with their component type                                 (:101-119)
//   * TANGENT vec4 -> bitangent = normal x tangent * w                                              (:205-233)
//   * PBR metallic-roughness materials with base colour / normal / metallic-roughness textures      (:364-407)
//   * images from `data:image/png;base64,` URIs or files next to the .gltf                          (:305-343)
//   * perspective cameras, nodes with TRS or a matrix                                               (:494-566)
// No GLB, no buffer-view images, no lights: the reference has none of these either (:84-99, :313).
// Image decoding: the reference uses the `image` crate; here PNG (8/16-bit, grey/RGB/palette/RGBA, non-interlaced)
// is decoded by the small inflate + unfilter below; other formats raise rayca::Error.
// Matrix nodes: the `gltf` crate's Transform::decomposed() (gltf 1.4.1, not vendored in the reference tree) is
// restated in decompose_matrix() -- column lengths as scale, determinant sign on z, trace-based quaternion -- in
// f32, the same restatement as rayca_amd/gltf.py (parity for matrix-authored nodes is unpinned).
// Header-only, C++17.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rayca.hpp"
#include "rayca_jpeg.hpp"

namespace rayca {
namespace gltf_detail {

// ---- JSON ----------------------------------------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* find(const std::string& k) const {
    for (const auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool has(const std::string& k) const { return find(k) != nullptr; }
  const Json& at(const std::string& k) const {
    const Json* j = find(k);
    if (!j) throw Error(RAYCA_ERR_BAD_ARG, "glTF: missing key '" + k + "'");
    return *j;
  }
  const Json& at(size_t i) const {
    if (kind != Arr || i >= arr.size()) throw Error(RAYCA_ERR_BAD_ARG, "glTF: array index out of range");
    return arr[i];
  }
  size_t size() const { return kind == Arr ? arr.size() : obj.size(); }
  double number(double dflt) const { return kind == Num ? num : dflt; }
  uint32_t u32() const {
    if (kind != Num) throw Error(RAYCA_ERR_BAD_ARG, "glTF: number expected");
    return (uint32_t)num;
  }
};

struct JsonParser {
  const char* p;
  const char* end;
  [[noreturn]] void bad(const char* what) const { throw Error(RAYCA_ERR_BAD_ARG, std::string("glTF JSON: ") + what); }
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
  }
  Json value() {
    ws();
    if (p >= end) bad("unexpected end");
    Json j;
    if (*p == '{') {
      ++p;
      j.kind = Json::Obj;
      ws();
      if (p < end && *p == '}') { ++p; return j; }
      for (;;) {
        ws();
        Json k = string_value();
        ws();
        if (p >= end || *p != ':') bad("':' expected");
        ++p;
        j.obj.emplace_back(k.str, value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return j; }
        bad("',' or '}' expected");
      }
    }
    if (*p == '[') {
      ++p;
      j.kind = Json::Arr;
      ws();
      if (p < end && *p == ']') { ++p; return j; }
      for (;;) {
        j.arr.push_back(value());
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return j; }
        bad("',' or ']' expected");
      }
    }
    if (*p == '"') return string_value();
    if (end - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; j.kind = Json::Bool; j.b = true; return j; }
    if (end - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; j.kind = Json::Bool; return j; }
    if (end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return j; }
    char* e = nullptr;
    j.num = std::strtod(p, &e);
    if (e == p) bad("value expected");
    p = e;
    j.kind = Json::Num;
    return j;
  }
  Json string_value() {
    if (p >= end || *p != '"') bad("string expected");
    ++p;
    Json j;
    j.kind = Json::Str;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) bad("bad escape");
        switch (*p) {
          case 'n': j.str += '\n'; break;
          case 't': j.str += '\t'; break;
          case 'r': j.str += '\r'; break;
          case 'b': j.str += '\b'; break;
          case 'f': j.str += '\f'; break;
          case 'u': {  // BMP code point -> UTF-8
            if (end - p < 5) bad("bad \\u escape");
            const unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            if (cp < 0x80) j.str += (char)cp;
            else if (cp < 0x800) { j.str += (char)(0xC0 | (cp >> 6)); j.str += (char)(0x80 | (cp & 0x3F)); }
            else { j.str += (char)(0xE0 | (cp >> 12)); j.str += (char)(0x80 | ((cp >> 6) & 0x3F)); j.str += (char)(0x80 | (cp & 0x3F)); }
            p += 4;
          } break;
          default: j.str += *p;
        }
        ++p;
      } else {
        j.str += *p++;
      }
    }
    if (p >= end) bad("unterminated string");
    ++p;
    return j;
  }
};

inline std::vector<uint8_t> read_file(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw Error(RAYCA_ERR_BAD_ARG, "cannot open " + path);
  std::vector<uint8_t> data;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
  std::fclose(f);
  return data;
}

inline std::vector<uint8_t> base64_decode(const char* s, size_t n) {
  std::vector<uint8_t> out;
  uint32_t acc = 0;
  int bits = 0;
  for (size_t i = 0; i < n; ++i) {
    const char c = s[i];
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A';
    else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+') v = 62;
    else if (c == '/') v = 63;
    else continue;  // '=', whitespace
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((uint8_t)(acc >> bits));
    }
  }
  return out;
}

// ---- inflate (RFC 1951) + zlib wrapper -----------------------------------------------------------------------------
struct Inflater {
  const uint8_t* in;
  size_t n, pos = 0;
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  std::vector<uint8_t>& out;
  Inflater(const uint8_t* d, size_t len, std::vector<uint8_t>& o) : in(d), n(len), out(o) {}
  [[noreturn]] static void bad() { throw Error(RAYCA_ERR_BAD_ARG, "PNG: corrupt deflate stream"); }
  uint32_t bits(int need) {
    while (bitcnt < need) {
      if (pos >= n) bad();
      bitbuf |= (uint32_t)in[pos++] << bitcnt;
      bitcnt += 8;
    }
    const uint32_t v = bitbuf & ((need < 32 ? (1u << need) : 0u) - 1u);
    bitbuf >>= need;
    bitcnt -= need;
    return v;
  }
  struct Huff {
    uint16_t count[16] = {0};
    std::vector<uint16_t> symbol;
    void build(const uint8_t* lens, int nsym) {
      for (auto& c : count) c = 0;
      for (int i = 0; i < nsym; ++i) count[lens[i]]++;
      count[0] = 0;
      uint16_t offs[16];
      offs[1] = 0;
      for (int i = 1; i < 15; ++i) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
      symbol.assign((size_t)nsym, 0);
      for (int i = 0; i < nsym; ++i)
        if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
    }
  };
  int decode(const Huff& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
      code |= (int)bits(1);
      const int count = h.count[len];
      if (code - count < first) return h.symbol[(size_t)(index + (code - first))];
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    bad();
  }
  void codes(const Huff& lit, const Huff& dist) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(lit);
      if (sym < 256) out.push_back((uint8_t)sym);
      else if (sym == 256) return;
      else {
        sym -= 257;
        if (sym >= 29) bad();
        const size_t len = lbase[sym] + bits(lext[sym]);
        const int ds = decode(dist);
        if (ds >= 30) bad();
        const size_t d = dbase[ds] + bits(dext[ds]);
        if (d > out.size()) bad();
        for (size_t i = 0; i < len; ++i) out.push_back(out[out.size() - d]);
      }
    }
  }
  void run() {
    bits(16);  // zlib header (CMF, FLG)
    int last;
    do {
      last = (int)bits(1);
      const uint32_t type = bits(2);
      if (type == 0) {
        bitbuf = 0;
        bitcnt = 0;
        if (pos + 4 > n) bad();
        const size_t len = in[pos] | ((size_t)in[pos + 1] << 8);
        pos += 4;
        if (pos + len > n) bad();
        out.insert(out.end(), in + pos, in + pos + len);
        pos += len;
      } else if (type == 1) {
        uint8_t lens[320];
        int i = 0;
        for (; i < 144; ++i) lens[i] = 8;
        for (; i < 256; ++i) lens[i] = 9;
        for (; i < 280; ++i) lens[i] = 7;
        for (; i < 288; ++i) lens[i] = 8;
        Huff lit, dist;
        lit.build(lens, 288);
        for (i = 0; i < 30; ++i) lens[i] = 5;
        dist.build(lens, 30);
        codes(lit, dist);
      } else if (type == 2) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        uint8_t lens[320] = {0};
        for (int i = 0; i < ncode; ++i) lens[order[i]] = (uint8_t)bits(3);
        Huff lencode;
        lencode.build(lens, 19);
        int idx = 0;
        uint8_t ll[320] = {0};
        while (idx < nlen + ndist) {
          int sym = decode(lencode);
          if (sym < 16) ll[idx++] = (uint8_t)sym;
          else {
            uint8_t prev = 0;
            int rep;
            if (sym == 16) {
              if (idx == 0) bad();
              prev = ll[idx - 1];
              rep = 3 + (int)bits(2);
            } else if (sym == 17) rep = 3 + (int)bits(3);
            else rep = 11 + (int)bits(7);
            if (idx + rep > nlen + ndist) bad();
            while (rep--) ll[idx++] = prev;
          }
        }
        Huff lit, dist;
        lit.build(ll, nlen);
        dist.build(ll + nlen, ndist);
        codes(lit, dist);
      } else bad();
    } while (!last);
  }
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG -> Image (RGB8 or RGBA8), like image::ImageReader::decode + Image::load_data (rayca-model/src/image.rs:143-158)
inline Image decode_png(const std::vector<uint8_t>& f) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (f.size() < 8 || std::memcmp(f.data(), sig, 8)) throw Error(RAYCA_ERR_BAD_ARG, "PNG: bad signature");
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  for (size_t pos = 8; pos + 12 <= f.size();) {
    const uint32_t len = be32(&f[pos]);
    const char* type = reinterpret_cast<const char*>(&f[pos + 4]);
    const uint8_t* body = &f[pos + 8];
    if (pos + 12 + len > f.size()) throw Error(RAYCA_ERR_BAD_ARG, "PNG: truncated chunk");
    if (!std::strncmp(type, "IHDR", 4)) {
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!std::strncmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!std::strncmp(type, "PLTE", 4)) plte.assign(body, body + len);
    else if (!std::strncmp(type, "tRNS", 4)) trns.assign(body, body + len);
    else if (!std::strncmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!w || !h || interlace || (depth != 8 && depth != 16) ) throw Error(RAYCA_ERR_UNSUPPORTED, "PNG: only non-interlaced 8/16-bit images");
  const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels || (ctype == 3 && depth != 8)) throw Error(RAYCA_ERR_UNSUPPORTED, "PNG: colour type not supported");
  const size_t bpp = (size_t)channels * (depth / 8), stride = (size_t)w * bpp;
  std::vector<uint8_t> raw;
  raw.reserve((stride + 1) * h);
  Inflater(idat.data(), idat.size(), raw).run();
  if (raw.size() < (stride + 1) * h) throw Error(RAYCA_ERR_BAD_ARG, "PNG: image data too short");
  std::vector<uint8_t> px(stride * h);
  for (uint32_t y = 0; y < h; ++y) {  // unfilter
    const uint8_t ft = raw[y * (stride + 1)];
    const uint8_t* src = &raw[y * (stride + 1) + 1];
    uint8_t* dst = &px[y * stride];
    const uint8_t* up = y ? &px[(y - 1) * stride] : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) / 2;
      else if (ft == 4) {
        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      } else if (ft != 0) throw Error(RAYCA_ERR_BAD_ARG, "PNG: bad filter type");
      dst[i] = (uint8_t)(src[i] + pred);
    }
  }
  const bool alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
  Image im(w, h, alpha ? ColorType::RGBA8 : ColorType::RGB8);
  const size_t oc = alpha ? 4 : 3, step = (size_t)(depth / 8);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    const uint8_t* s = &px[i * bpp];
    uint8_t* d = &im.data[i * oc];
    if (ctype == 3) {
      const size_t k = s[0];
      if (3 * k + 2 >= plte.size()) throw Error(RAYCA_ERR_BAD_ARG, "PNG: palette index out of range");
      d[0] = plte[3 * k]; d[1] = plte[3 * k + 1]; d[2] = plte[3 * k + 2];
      if (alpha) d[3] = k < trns.size() ? trns[k] : 255;
    } else if (ctype == 0 || ctype == 4) {
      d[0] = d[1] = d[2] = s[0];
      if (alpha) d[3] = s[step];
    } else {
      d[0] = s[0]; d[1] = s[step]; d[2] = s[2 * step];
      if (alpha) d[3] = s[3 * step];
    }
  }
  return im;
}

// image::ImageReader::with_guessed_format().decode()  (rayca-model/src/image.rs:143-158): PNG or JPEG, by signature
inline Image decode_image(const std::vector<uint8_t>& f) {
  if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF) return decode_jpeg(f);
  if (f.size() >= 8 && f[0] == 0x89 && f[1] == 'P' && f[2] == 'N' && f[3] == 'G') return decode_png(f);
  throw Error(RAYCA_ERR_UNSUPPORTED, "image: neither PNG nor JPEG (the two formats this loader decodes)");
}

inline size_t component_size(uint32_t ct) {
  switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: throw Error(RAYCA_ERR_BAD_ARG, "glTF: unknown componentType");
  }
}
inline size_t type_width(const std::string& t) {
  if (t == "SCALAR") return 1;
  if (t == "VEC2") return 2;
  if (t == "VEC3") return 3;
  if (t == "VEC4") return 4;
  throw Error(RAYCA_ERR_BAD_ARG, "glTF: accessor type " + t + " not supported");
}

struct Document {
  Json doc;
  std::vector<std::vector<uint8_t>> buffers;
  struct View {
    const uint8_t* base;
    size_t count, stride, width, csize;
    uint32_t component;
  };
  View accessor(uint32_t index) const {
    const Json& acc = doc.at("accessors").at(index);
    const Json& view = doc.at("bufferViews").at(acc.at("bufferView").u32());
    View v;
    v.component = acc.at("componentType").u32();
    v.csize = component_size(v.component);
    v.width = type_width(acc.at("type").str);
    v.count = acc.at("count").u32();
    const size_t start = (size_t)(view.has("byteOffset") ? view.at("byteOffset").num : 0) + (size_t)(acc.has("byteOffset") ? acc.at("byteOffset").num : 0);
    v.stride = view.has("byteStride") ? (size_t)view.at("byteStride").num : v.csize * v.width;
    const std::vector<uint8_t>& buf = buffers.at(view.at("buffer").u32());
    if (v.count && start + (v.count - 1) * v.stride + v.csize * v.width > buf.size()) throw Error(RAYCA_ERR_BAD_ARG, "glTF: accessor runs past its buffer");
    v.base = buf.data() + start;
    return v;
  }
  // f32 element (i, c); the reference reads attributes as f32 slices (gltf.rs:121-147)
  static float f32_at(const View& v, size_t i, size_t c) {
    if (v.component != 5126) throw Error(RAYCA_ERR_UNSUPPORTED, "glTF: vertex attributes must be f32 (gltf.rs:121-147)");
    float f;
    std::memcpy(&f, v.base + i * v.stride + c * 4, 4);
    return f;
  }
};

// gltf::scene::Transform::decomposed() for a column-major 4x4 matrix, in f32 (see the header comment)
inline Trs decompose_matrix(const float m[16]) {
  Trs t;
  t.translation = Vec3(m[12], m[13], m[14]);
  float x[3] = {m[0], m[1], m[2]}, y[3] = {m[4], m[5], m[6]}, z[3] = {m[8], m[9], m[10]};
  auto mag = [](const float v[3]) { return std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); };
  const double det = (double)x[0] * ((double)y[1] * z[2] - (double)z[1] * y[2]) - (double)y[0] * ((double)x[1] * z[2] - (double)z[1] * x[2]) +
                     (double)z[0] * ((double)x[1] * y[2] - (double)y[1] * x[2]);
  const float sx = mag(x), sy = mag(y);
  const float sz = (det < 0.0 ? -1.0f : 1.0f) * mag(z);
  const float ix = 1.0f / sx, iy = 1.0f / sy, iz = 1.0f / sz;
  for (int i = 0; i < 3; ++i) { x[i] *= ix; y[i] *= iy; z[i] *= iz; }
  const float trace = (x[0] + y[1]) + z[2];
  Quat q;
  if (trace >= 0.0f) {
    float s = std::sqrt(1.0f + trace);
    const float w = 0.5f * s;
    s = 0.5f / s;
    q = Quat((y[2] - z[1]) * s, (z[0] - x[2]) * s, (x[1] - y[0]) * s, w);
  } else if (x[0] > y[1] && x[0] > z[2]) {
    float s = std::sqrt(((x[0] - y[1]) - z[2]) + 1.0f);
    const float qx = 0.5f * s;
    s = 0.5f / s;
    q = Quat(qx, (y[0] + x[1]) * s, (x[2] + z[0]) * s, (y[2] - z[1]) * s);
  } else if (y[1] > z[2]) {
    float s = std::sqrt(((y[1] - x[0]) - z[2]) + 1.0f);
    const float qy = 0.5f * s;
    s = 0.5f / s;
    q = Quat((y[0] + x[1]) * s, qy, (z[1] + y[2]) * s, (z[0] - x[2]) * s);
  } else {
    float s = std::sqrt(((z[2] - x[0]) - y[1]) + 1.0f);
    const float qz = 0.5f * s;
    s = 0.5f / s;
    q = Quat((x[2] + z[0]) * s, (z[1] + y[2]) * s, qz, (x[1] - y[0]) * s);
  }
  t.rotation = q;
  t.scale = Vec3(sx, sy, sz);
  return t;
}

}  // namespace gltf_detail

// Model::load_gltf_path  rayca-model/src/loader/gltf.rs:291-299
inline Model load_gltf_path(const std::string& path) {
  using namespace gltf_detail;
  const std::vector<uint8_t> text = read_file(path);
  JsonParser jp{reinterpret_cast<const char*>(text.data()), reinterpret_cast<const char*>(text.data()) + text.size()};
  Document d;
  d.doc = jp.value();
  const Json& doc = d.doc;
  const size_t slash = path.find_last_of('/');
  const std::string dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);

  if (const Json* bufs = doc.find("buffers"))
    for (const Json& b : bufs->arr) {
      const std::string& uri = b.at("uri").str;
      static const std::string kData = "data:application/octet-stream;base64,";
      if (uri.compare(0, kData.size(), kData) == 0) d.buffers.push_back(base64_decode(uri.data() + kData.size(), uri.size() - kData.size()));
      else d.buffers.push_back(read_file(dir + "/" + uri));
    }

  Model model;
  // load_images / load_textures  gltf.rs:305-362
  if (const Json* images = doc.find("images"))
    for (const Json& im : images->arr) {
      if (!im.has("uri")) throw Error(RAYCA_ERR_UNSUPPORTED, "glTF: buffer-view images are todo!() in the reference (gltf.rs:313)");
      const std::string& uri = im.at("uri").str;
      static const std::string kPng = "data:image/png;base64,";
      // (the reference only recognises the PNG data URI, gltf.rs:315-321; anything else is a path; the format of the bytes is
      // guessed from their signature, image.rs:143-147)
      if (uri.compare(0, kPng.size(), kPng) == 0) model.images.push(decode_image(base64_decode(uri.data() + kPng.size(), uri.size() - kPng.size())));
      else model.images.push(decode_image(read_file(dir + "/" + uri)));
    }
  if (const Json* textures = doc.find("textures"))
    for (const Json& t : textures->arr) model.textures.push(Texture(Handle<Image>(t.at("source").u32())));
  // load_materials  gltf.rs:364-407 (glTF defaults: base colour 1, metallic 1, roughness 1)
  if (const Json* materials = doc.find("materials"))
    for (const Json& gm : materials->arr) {
      PbrMaterial m;
      m.metallic_factor = 1.0f;
      m.roughness_factor = 1.0f;
      if (const Json* pbr = gm.find("pbrMetallicRoughness")) {
        if (const Json* c = pbr->find("baseColorFactor")) m.color = Color((float)c->at(0).num, (float)c->at(1).num, (float)c->at(2).num, (float)c->at(3).num);
        if (const Json* t = pbr->find("baseColorTexture")) m.albedo = Handle<Texture>(t->at("index").u32());
        if (const Json* t = pbr->find("metallicRoughnessTexture")) m.metallic_roughness = Handle<Texture>(t->at("index").u32());
        if (const Json* v = pbr->find("metallicFactor")) m.metallic_factor = (float)v->num;
        if (const Json* v = pbr->find("roughnessFactor")) m.roughness_factor = (float)v->num;
      }
      if (const Json* t = gm.find("normalTexture")) m.normal = Handle<Texture>(t->at("index").u32());
      model.materials.push(Material(m));
    }
  // load_meshes / load_primitive / load_vertices  gltf.rs:409-492
  if (const Json* meshes = doc.find("meshes"))
    for (const Json& gmesh : meshes->arr) {
      Mesh mesh;
      for (const Json& gp : gmesh.at("primitives").arr) {
        if (gp.has("mode") && gp.at("mode").u32() != 4) throw Error(RAYCA_ERR_UNSUPPORTED, "glTF: only TRIANGLES primitives (gltf.rs:417)");
        const Json& at = gp.at("attributes");
        TriangleMesh tm;
        const Document::View pos = d.accessor(at.at("POSITION").u32());
        tm.vertices.resize(pos.count);
        for (size_t i = 0; i < pos.count; ++i) tm.vertices[i].pos = Point3(Document::f32_at(pos, i, 0), Document::f32_at(pos, i, 1), Document::f32_at(pos, i, 2));
        if (at.has("NORMAL")) {
          const Document::View v = d.accessor(at.at("NORMAL").u32());
          for (size_t i = 0; i < v.count && i < tm.vertices.size(); ++i)
            tm.vertices[i].ext.normal = Vec3(Document::f32_at(v, i, 0), Document::f32_at(v, i, 1), Document::f32_at(v, i, 2));
        }
        if (at.has("TEXCOORD_0")) {
          const Document::View v = d.accessor(at.at("TEXCOORD_0").u32());
          for (size_t i = 0; i < v.count && i < tm.vertices.size(); ++i) tm.vertices[i].ext.uv = Vec2(Document::f32_at(v, i, 0), Document::f32_at(v, i, 1));
        }
        if (at.has("COLOR_0")) {
          const Document::View v = d.accessor(at.at("COLOR_0").u32());
          for (size_t i = 0; i < v.count && i < tm.vertices.size(); ++i)
            tm.vertices[i].ext.color = Color(Document::f32_at(v, i, 0), Document::f32_at(v, i, 1), Document::f32_at(v, i, 2), v.width == 4 ? Document::f32_at(v, i, 3) : 1.0f);
        }
        if (at.has("TANGENT")) {  // load_tangents  gltf.rs:205-233: bitangent = normal x tangent * w, products rounded separately
          const Document::View v = d.accessor(at.at("TANGENT").u32());
          for (size_t i = 0; i < v.count && i < tm.vertices.size(); ++i) {
            Vertex& vx = tm.vertices[i];
            const float tx = Document::f32_at(v, i, 0), ty = Document::f32_at(v, i, 1), tz = Document::f32_at(v, i, 2), tw = Document::f32_at(v, i, 3);
            vx.ext.tangent = Vec3(tx, ty, tz);
            const Vec3 n = vx.ext.normal;
            const float ax = n.y * tz, bx = n.z * ty, ay = n.z * tx, by = n.x * tz, az = n.x * ty, bz = n.y * tx;
            vx.ext.bitangent = Vec3((ax - bx) * tw, (ay - by) * tw, (az - bz) * tw);
          }
        }
        if (gp.has("indices")) {  // load_indices  gltf.rs:101-119: bytes kept as they are
          const Document::View v = d.accessor(gp.at("indices").u32());
          if (v.component != 5121 && v.component != 5123 && v.component != 5125) throw Error(RAYCA_ERR_BAD_ARG, "glTF: index type not supported (primitive.rs:258)");
          tm.indices.index_type = (ComponentType)v.component;
          tm.indices.indices.resize(v.count * v.csize);
          for (size_t i = 0; i < v.count; ++i) std::memcpy(&tm.indices.indices[i * v.csize], v.base + i * v.stride, v.csize);
        }
        const Handle<Geometry> g = model.geometries.push(Geometry(std::move(tm)));
        Primitive prim;
        prim.geometry = g;
        if (gp.has("material")) prim.material = Handle<Material>(gp.at("material").u32());
        mesh.primitives.push_back(model.primitives.push(prim));
      }
      model.meshes.push(std::move(mesh));
    }
  // load_cameras  gltf.rs:494-518
  if (const Json* cameras = doc.find("cameras"))
    for (const Json& gc : cameras->arr) {
      Camera c;
      if (gc.at("type").str == "perspective") c.yfov_radians = (float)gc.at("perspective").at("yfov").num;
      else c.yfov_radians = 1.0f;  // Camera::orthographic sets yfov 1.0 (camera.rs:70)
      model.cameras.push(c);
    }
  // load_nodes / create_node  gltf.rs:520-566
  if (const Json* scenes = doc.find("scenes")) {
    const size_t si = doc.has("scene") ? doc.at("scene").u32() : 0;
    if (si < scenes->arr.size())
      if (const Json* roots = scenes->arr[si].find("nodes"))
        for (const Json& r : roots->arr) model.root.children.push_back(Handle<Node>(r.u32()));
  }
  if (const Json* nodes = doc.find("nodes"))
    for (const Json& gn : nodes->arr) {
      Node n;
      if (const Json* m = gn.find("matrix")) {
        float mm[16];
        for (int i = 0; i < 16; ++i) mm[i] = (float)m->at((size_t)i).num;
        n.trs = decompose_matrix(mm);
      } else {
        if (const Json* t = gn.find("translation")) n.trs.translation = Vec3((float)t->at(0).num, (float)t->at(1).num, (float)t->at(2).num);
        if (const Json* r = gn.find("rotation")) n.trs.rotation = Quat((float)r->at(0).num, (float)r->at(1).num, (float)r->at(2).num, (float)r->at(3).num);
        if (const Json* sc = gn.find("scale")) n.trs.scale = Vec3((float)sc->at(0).num, (float)sc->at(1).num, (float)sc->at(2).num);
      }
      if (const Json* ch = gn.find("children"))
        for (const Json& c : ch->arr) n.children.push_back(Handle<Node>(c.u32()));
      if (gn.has("mesh")) n.mesh = Handle<Mesh>(gn.at("mesh").u32());
      if (gn.has("camera")) n.camera = Handle<Camera>(gn.at("camera").u32());
      n.name = gn.has("name") ? gn.at("name").str : std::string("Unknown");
      model.nodes.push(std::move(n));
    }
  return model;
}

// Scene::push_gltf_from_path  rayca-model/src/scene.rs:117-124
inline Handle<Node> push_gltf_from_path(Scene& scene, const std::string& path) { return scene.push_model(load_gltf_path(path)); }

}  // namespace rayca
