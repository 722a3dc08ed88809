// rayca_jpeg.hpp -- JPEG -> Image (RGB8) for the C++ glTF loader.
//
// The reference decodes textures with `image::ImageReader::with_guessed_format().decode()` (rayca-model/src/image.rs:143-158,
// called from loader/gltf.rs:309-337); for JPEG that is image 0.25.6 -> zune-jpeg 0.4.20 (Cargo.lock:1042,3631), a third-party
// crate that is not vendored under /root/reference.  This is a restatement of the published JPEG decoding process (ITU-T T.81:
// Huffman-coded baseline / extended-sequential / progressive DCT, 8-bit samples, restart intervals) with libjpeg's reference
// arithmetic for the three steps the standard leaves open -- the slow-but-accurate integer IDCT (jidctint), "fancy" triangle
// upsampling of subsampled chroma (jdsample) and 16-bit fixed-point YCbCr -> RGB (jdcolor) -- so that a decode can be checked
// pixel for pixel against a libjpeg decode (tests/test_jpeg.py: committed fixtures + their libjpeg-turbo decodes, +-1 allowed,
// exact in practice).  Parity with zune-jpeg's own rounding is unpinned (its source is not available offline).
//
// Like the reference, only what `from_image_color_type` accepts survives (image.rs:193-200): a colour JPEG gives RGB8; a
// greyscale one (L8) makes the reference panic ("Unsupported image color type") and raises RAYCA_ERR_UNSUPPORTED here.
// Not handled (Error): arithmetic coding, lossless / hierarchical modes, 12-bit samples, CMYK / YCCK.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rayca.hpp"

namespace rayca {
namespace jpeg_detail {

[[noreturn]] inline void bad(const char* what) { throw Error(RAYCA_ERR_BAD_ARG, std::string("JPEG: ") + what); }
[[noreturn]] inline void unsupported(const char* what) { throw Error(RAYCA_ERR_UNSUPPORTED, std::string("JPEG: ") + what); }

// zigzag position -> natural (row-major) position; 16 extra entries guard corrupt run lengths (as libjpeg's jpeg_natural_order)
static const uint8_t kNatural[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                          6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                          39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {  // T.81 Annex C / F.2.2.3: canonical codes, decoded length by length
  bool present = false;
  uint8_t values[256];
  int32_t maxcode[18];   // largest code of each length, -1 if none
  int32_t valoffset[17]; // values[] index of the first code of each length minus that code
  void build(const uint8_t counts[16], const uint8_t* vals, int n) {
    std::memcpy(values, vals, (size_t)n);
    int32_t code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valoffset[len] = k - code;
      k += counts[len - 1];
      code += counts[len - 1];
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
  }
};

struct BitReader {  // entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment (zero bits follow)
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
  void fill() {
    while (nbits <= 24) {
      uint32_t byte = 0;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;
          else {
            hit_marker = true;
            byte = 0;
          }
        } else ++p;
      }
      acc |= byte << (24 - nbits);
      nbits += 8;
    }
  }
  int bit() {
    if (nbits < 1) fill();
    const int b = (int)(acc >> 31);
    acc <<= 1;
    --nbits;
    return b;
  }
  int bits(int n) {  // n in 0..16
    if (n == 0) return 0;
    if (nbits < n) fill();
    const int v = (int)(acc >> (32 - n));
    acc <<= n;
    nbits -= n;
    return v;
  }
  int decode(const Huffman& h) {
    if (!h.present) bad("scan refers to a Huffman table that was not defined");
    int32_t code = 0;
    for (int len = 1; len <= 16; ++len) {
      code = (code << 1) | bit();
      if (code <= h.maxcode[len]) return h.values[(code + h.valoffset[len]) & 0xFF];
    }
    bad("corrupt Huffman code");
  }
  // RSTn: discard the remaining bits, step over the marker
  void restart() {
    acc = 0;
    nbits = 0;
    hit_marker = false;
    while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;  // (stray fill bytes in front of the marker)
    if (p + 1 < end) p += 2;
  }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }  // T.81 F.2.2.1 EXTEND

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int width = 0, height = 0;        // downsampled size in samples: ceil(image * h / hmax)
  int blocks_w = 0, blocks_h = 0;   // blocks of a non-interleaved scan: ceil(width / 8), ceil(height / 8)
  int stride_b = 0, rows_b = 0;     // allocated blocks: padded to whole MCUs
  std::vector<int16_t> coef;        // [rows_b][stride_b][64], natural order
  int dc_table = 0, ac_table = 0, dc_pred = 0;
  std::vector<uint8_t> plane;       // [rows_b * 8][stride_b * 8] after the IDCT
};

// jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2
inline void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
  constexpr int32_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137, F1_961 = 16069,
                    F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  int32_t ws[64];
  auto descale = [](int32_t x, int n) { return (x + (1 << (n - 1))) >> n; };
  for (int c = 0; c < 8; ++c) {
    const int32_t i0 = in[c] * q[c], i1 = in[8 + c] * q[8 + c], i2 = in[16 + c] * q[16 + c], i3 = in[24 + c] * q[24 + c], i4 = in[32 + c] * q[32 + c],
                  i5 = in[40 + c] * q[40 + c], i6 = in[48 + c] * q[48 + c], i7 = in[56 + c] * q[56 + c];
    int32_t z1 = (i2 + i6) * F0_541;
    const int32_t t2 = z1 + i6 * (-F1_847), t3 = z1 + i2 * F0_765;
    const int32_t t0 = (i0 + i4) * 8192, t1 = (i0 - i4) * 8192;
    const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    int32_t o0 = i7, o1 = i5, o2 = i3, o3 = i1;
    z1 = o0 + o3;
    int32_t z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
    const int32_t z5 = (z3 + z4) * F1_175;
    o0 *= F0_298; o1 *= F2_053; o2 *= F3_072; o3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
    ws[c] = descale(t10 + o3, 11); ws[56 + c] = descale(t10 - o3, 11);
    ws[8 + c] = descale(t11 + o2, 11); ws[48 + c] = descale(t11 - o2, 11);
    ws[16 + c] = descale(t12 + o1, 11); ws[40 + c] = descale(t12 - o1, 11);
    ws[24 + c] = descale(t13 + o0, 11); ws[32 + c] = descale(t13 - o0, 11);
  }
  auto sample = [&](int32_t x) {  // range_limit: descale by CONST_BITS + PASS1_BITS + 3, re-centre, clamp
    const int32_t v = descale(x, 18) + 128;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  };
  for (int r = 0; r < 8; ++r) {
    const int32_t* w = ws + 8 * r;
    int32_t z1 = (w[2] + w[6]) * F0_541;
    const int32_t t2 = z1 + w[6] * (-F1_847), t3 = z1 + w[2] * F0_765;
    const int32_t t0 = (w[0] + w[4]) * 8192, t1 = (w[0] - w[4]) * 8192;
    const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    int32_t o0 = w[7], o1 = w[5], o2 = w[3], o3 = w[1];
    z1 = o0 + o3;
    int32_t z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
    const int32_t z5 = (z3 + z4) * F1_175;
    o0 *= F0_298; o1 *= F2_053; o2 *= F3_072; o3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
    uint8_t* o = out + r * out_stride;
    o[0] = sample(t10 + o3); o[7] = sample(t10 - o3);
    o[1] = sample(t11 + o2); o[6] = sample(t11 - o2);
    o[2] = sample(t12 + o1); o[5] = sample(t12 - o1);
    o[3] = sample(t13 + o0); o[4] = sample(t13 - o0);
  }
}

struct Decoder {
  const uint8_t* data;
  size_t size;
  int width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  bool progressive = false, saw_sof = false, jfif = false, adobe = false;
  int adobe_transform = 0, restart_interval = 0;
  std::vector<Component> comps;
  uint16_t quant[4][64];
  bool quant_present[4] = {false, false, false, false};
  Huffman dc[4], ac[4];

  Decoder(const uint8_t* d, size_t n) : data(d), size(n) {}

  static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

  void dqt(const uint8_t* p, int len) {
    while (len > 0) {
      const int pq = p[0] >> 4, tq = p[0] & 15;
      if (tq > 3 || pq > 1) bad("bad quantisation table header");
      const int need = 1 + 64 * (pq + 1);
      if (len < need) bad("truncated DQT");
      for (int i = 0; i < 64; ++i) quant[tq][kNatural[i]] = (uint16_t)(pq ? be16(p + 1 + 2 * i) : p[1 + i]);
      quant_present[tq] = true;
      p += need;
      len -= need;
    }
  }
  void dht(const uint8_t* p, int len) {
    while (len > 0) {
      if (len < 17) bad("truncated DHT");
      const int tc = p[0] >> 4, th = p[0] & 15;
      if (tc > 1 || th > 3) bad("bad Huffman table header");
      int n = 0;
      for (int i = 0; i < 16; ++i) n += p[1 + i];
      if (n > 256 || len < 17 + n) bad("truncated DHT");
      (tc ? ac[th] : dc[th]).build(p + 1, p + 17, n);
      p += 17 + n;
      len -= 17 + n;
    }
  }
  void sof(const uint8_t* p, int len, bool prog) {
    if (saw_sof) bad("more than one frame header");
    if (len < 6) bad("truncated SOF");
    if (p[0] != 8) unsupported("only 8-bit samples");
    height = be16(p + 1);
    width = be16(p + 3);
    const int n = p[5];
    if (!width || !height) unsupported("image size 0 (DNL) is not handled");
    if (n != 1 && n != 3) unsupported(n == 4 ? "CMYK / YCCK images are not handled" : "unexpected number of components");
    if (len < 6 + 3 * n) bad("truncated SOF");
    comps.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
      Component& c = comps[(size_t)i];
      c.id = p[6 + 3 * i];
      c.h = p[7 + 3 * i] >> 4;
      c.v = p[7 + 3 * i] & 15;
      c.tq = p[8 + 3 * i];
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component specification");
      hmax = std::max(hmax, c.h);
      vmax = std::max(vmax, c.v);
    }
    mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
    mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
    for (Component& c : comps) {
      c.width = (width * c.h + hmax - 1) / hmax;
      c.height = (height * c.v + vmax - 1) / vmax;
      c.blocks_w = (c.width + 7) / 8;
      c.blocks_h = (c.height + 7) / 8;
      c.stride_b = mcus_x * c.h;
      c.rows_b = mcus_y * c.v;
      c.coef.assign((size_t)c.stride_b * c.rows_b * 64, 0);
    }
    progressive = prog;
    saw_sof = true;
  }

  // one scan; returns the position behind its entropy-coded data
  size_t scan(size_t pos) {
    const int len = be16(data + pos);
    if (len < 6 || pos + (size_t)len > size) bad("bad scan header");   // (before its first byte is read)
    const uint8_t* p = data + pos + 2;
    const int ns = p[0];
    if (ns < 1 || ns > (int)comps.size() || len != 6 + 2 * ns) bad("bad scan header");
    std::vector<Component*> sc;
    for (int i = 0; i < ns; ++i) {
      Component* c = nullptr;
      for (Component& k : comps)
        if (k.id == p[1 + 2 * i]) c = &k;
      if (!c) bad("scan refers to an unknown component");
      c->dc_table = p[2 + 2 * i] >> 4;
      c->ac_table = p[2 + 2 * i] & 15;
      if (c->dc_table > 3 || c->ac_table > 3) bad("bad table selector");
      sc.push_back(c);
    }
    const int ss = p[1 + 2 * ns], se = p[2 + 2 * ns], ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
    if (progressive) {
      if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss != 0 && ns != 1) || al > 13) bad("bad progressive scan parameters");
    } else if (ss != 0 || se != 63 || ah != 0 || al != 0) bad("bad sequential scan parameters");
    BitReader br(data + pos + len, data + size);
    for (Component* c : sc) c->dc_pred = 0;
    int eobrun = 0;
    const bool interleaved = ns > 1;
    const int units_x = interleaved ? mcus_x : sc[0]->blocks_w, units_y = interleaved ? mcus_y : sc[0]->blocks_h;
    int to_restart = restart_interval;
    for (int uy = 0; uy < units_y; ++uy)
      for (int ux = 0; ux < units_x; ++ux) {
        if (restart_interval && to_restart == 0) {
          br.restart();
          for (Component* c : sc) c->dc_pred = 0;
          eobrun = 0;
          to_restart = restart_interval;
        }
        for (Component* c : sc) {
          const int bh = interleaved ? c->h : 1, bv = interleaved ? c->v : 1;
          for (int by = 0; by < bv; ++by)
            for (int bx = 0; bx < bh; ++bx) {
              const int col = ux * bh + bx, row = uy * bv + by;
              int16_t* blk = &c->coef[((size_t)row * c->stride_b + col) * 64];
              if (!progressive) block_sequential(br, *c, blk);
              else if (ss == 0) {
                if (ah == 0) dc_first(br, *c, blk, al);
                else if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << al));
              } else if (ah == 0) ac_first(br, *c, blk, ss, se, al, eobrun);
              else ac_refine(br, *c, blk, ss, se, al, eobrun);
            }
        }
        --to_restart;
      }
    // behind the entropy-coded segment: the next marker that is not RSTn
    size_t q = (size_t)(br.p - data);
    while (q + 1 < size && !(data[q] == 0xFF && data[q + 1] != 0x00 && data[q + 1] != 0xFF && !(data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7))) ++q;
    return q;
  }

  void block_sequential(BitReader& br, Component& c, int16_t* blk) {  // T.81 F.2.2
    int s = br.decode(dc[c.dc_table]);
    if (s > 16) bad("bad DC size");
    if (s) c.dc_pred += extend(br.bits(s), s);
    blk[0] = (int16_t)c.dc_pred;
    for (int k = 1; k < 64;) {
      const int rs = br.decode(ac[c.ac_table]), r = rs >> 4;
      s = rs & 15;
      if (s) {
        k += r;
        blk[kNatural[k]] = (int16_t)extend(br.bits(s), s);
        ++k;
      } else {
        if (r != 15) break;  // EOB
        k += 16;             // ZRL
      }
    }
  }
  void dc_first(BitReader& br, Component& c, int16_t* blk, int al) {  // T.81 G.1.2.1
    const int s = br.decode(dc[c.dc_table]);
    if (s > 16) bad("bad DC size");
    if (s) c.dc_pred += extend(br.bits(s), s);
    blk[0] = (int16_t)(c.dc_pred * (1 << al));
  }
  void ac_first(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al, int& eobrun) {  // G.1.2.2
    if (eobrun > 0) {
      --eobrun;
      return;
    }
    for (int k = ss; k <= se; ++k) {
      const int rs = br.decode(ac[c.ac_table]), r = rs >> 4, s = rs & 15;
      if (s) {
        k += r;
        blk[kNatural[k]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
      } else if (r == 15) {
        k += 15;
      } else {
        eobrun = (1 << r) - 1;
        if (r) eobrun += br.bits(r);
        break;
      }
    }
  }
  void ac_refine(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al, int& eobrun) {  // G.1.2.3
    const int p1 = 1 << al, m1 = -(1 << al);
    auto correct = [&](int16_t* coef) {
      if (br.bit() && (*coef & p1) == 0) *coef = (int16_t)(*coef >= 0 ? *coef + p1 : *coef + m1);
    };
    int k = ss;
    if (eobrun == 0) {
      for (; k <= se; ++k) {
        const int rs = br.decode(ac[c.ac_table]);
        int r = rs >> 4, s = rs & 15;
        if (s) s = br.bit() ? p1 : m1;  // (size must be 1: a newly nonzero coefficient)
        else if (r != 15) {
          eobrun = 1 << r;
          if (r) eobrun += br.bits(r);
          break;
        }
        do {  // skip r still-zero coefficients, refining every already-nonzero one on the way
          int16_t* coef = blk + kNatural[k];
          if (*coef != 0) correct(coef);
          else if (--r < 0) break;
          ++k;
        } while (k <= se);
        if (s && k <= se) blk[kNatural[k]] = (int16_t)s;
      }
    }
    if (eobrun > 0) {
      for (; k <= se; ++k) {
        int16_t* coef = blk + kNatural[k];
        if (*coef != 0) correct(coef);
      }
      --eobrun;
    }
  }

  // ---- reconstruction --------------------------------------------------------------------------------------------------
  void inverse_dct() {
    for (Component& c : comps) {
      if (!quant_present[c.tq]) bad("component refers to a quantisation table that was not defined");
      const int stride = c.stride_b * 8;
      c.plane.assign((size_t)stride * c.rows_b * 8, 0);
      for (int by = 0; by < c.rows_b; ++by)
        for (int bx = 0; bx < c.stride_b; ++bx)
          idct_islow(&c.coef[((size_t)by * c.stride_b + bx) * 64], quant[c.tq], &c.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride);
    }
  }

  // component plane -> full resolution (width x height), libjpeg's jdsample.c
  std::vector<uint8_t> upsample(const Component& c) const {
    const int stride = c.stride_b * 8, dw = c.width, dh = c.height;
    const int fx = hmax / c.h, fy = vmax / c.v;
    if (hmax % c.h || vmax % c.v) unsupported("fractional sampling ratios");
    std::vector<uint8_t> out((size_t)width * height);
    auto row = [&](int y) { return &c.plane[(size_t)std::min(std::max(y, 0), dh - 1) * stride]; };  // edge rows replicate
    std::vector<uint8_t> line((size_t)dw * fx + 2);
    const bool fancy = dw > 2;
    for (int iy = 0; iy < dh; ++iy)
      for (int v = 0; v < fy; ++v) {
        const int oy = iy * fy + v;
        if (oy >= height) continue;
        const uint8_t* in0 = row(iy);
        if (fx == 1 && fy == 1) {
          std::memcpy(line.data(), in0, (size_t)dw);
        } else if (fx == 2 && fy == 1 && fancy) {  // h2v1_fancy_upsample
          line[0] = in0[0];
          line[1] = (uint8_t)((in0[0] * 3 + in0[1] + 2) >> 2);
          for (int x = 1; x < dw - 1; ++x) {
            const int t = in0[x] * 3;
            line[2 * x] = (uint8_t)((t + in0[x - 1] + 1) >> 2);
            line[2 * x + 1] = (uint8_t)((t + in0[x + 1] + 2) >> 2);
          }
          line[2 * (dw - 1)] = (uint8_t)((in0[dw - 1] * 3 + in0[dw - 2] + 1) >> 2);
          line[2 * (dw - 1) + 1] = in0[dw - 1];
        } else if (fx == 2 && fy == 2 && fancy) {  // h2v2_fancy_upsample: the nearer neighbouring row weighs 1/4
          const uint8_t* in1 = row(v == 0 ? iy - 1 : iy + 1);
          int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
          line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
          line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          for (int x = 1; x < dw - 1; ++x) {
            lastcol = thiscol;
            thiscol = nextcol;
            nextcol = in0[x + 1] * 3 + in1[x + 1];
            line[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            line[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          }
          lastcol = thiscol;
          thiscol = nextcol;
          line[2 * (dw - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          line[2 * (dw - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        } else if (fx == 1 && fy == 2) {  // h1v2_fancy_upsample (libjpeg-turbo)
          const uint8_t* in1 = row(v == 0 ? iy - 1 : iy + 1);
          const int bias = v == 0 ? 1 : 2;
          for (int x = 0; x < dw; ++x) line[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
        } else {  // int_upsample / h2v1_upsample / h2v2_upsample: replication
          for (int x = 0; x < dw; ++x)
            for (int k = 0; k < fx; ++k) line[(size_t)x * fx + k] = in0[x];
        }
        std::memcpy(&out[(size_t)oy * width], line.data(), (size_t)width);
      }
    return out;
  }

  Image run() {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) bad("missing SOI");
    size_t pos = 2;
    bool done = false;
    while (!done) {
      while (pos < size && data[pos] != 0xFF) ++pos;
      while (pos < size && data[pos] == 0xFF) ++pos;
      if (pos >= size) break;
      const int m = data[pos++];
      if (m == 0xD9) break;                                   // EOI
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;    // TEM, stray RSTn
      if (pos + 2 > size) bad("truncated marker segment");
      const int len = be16(data + pos);
      if (len < 2 || pos + (size_t)len > size) bad("truncated marker segment");
      const uint8_t* body = data + pos + 2;
      switch (m) {
        case 0xC0: case 0xC1: sof(body, len - 2, false); break;
        case 0xC2: sof(body, len - 2, true); break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF: unsupported("lossless / hierarchical frames are not handled");
        case 0xC9: case 0xCA: case 0xCC: unsupported("arithmetic coding is not handled");
        case 0xC4: dht(body, len - 2); break;
        case 0xDB: dqt(body, len - 2); break;
        case 0xDD:
          if (len != 4) bad("bad DRI");
          restart_interval = be16(body);
          break;
        case 0xE0: if (len >= 7 && !std::memcmp(body, "JFIF", 5)) jfif = true; break;
        case 0xEE:
          if (len >= 14 && !std::memcmp(body, "Adobe", 5)) {
            adobe = true;
            adobe_transform = body[11];
          }
          break;
        case 0xDA:
          if (!saw_sof) bad("scan before the frame header");
          pos = scan(pos);
          continue;
        default: break;  // APPn, COM, DNL ...: skipped
      }
      pos += (size_t)len;
    }
    if (!saw_sof) bad("no frame header");
    if (comps.size() == 1) unsupported("greyscale (L8) images make the reference panic: Unsupported image color type (rayca-model/src/image.rs:198)");
    inverse_dct();
    std::vector<uint8_t> plane[3];
    for (int i = 0; i < 3; ++i) plane[i] = upsample(comps[(size_t)i]);
    // jdapimin.c default_decompress_parms: JFIF => YCbCr; Adobe => by its transform flag; else component ids R,G,B => RGB
    bool ycc = true;
    if (!jfif) {
      if (adobe) ycc = adobe_transform != 0;
      else if (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
    }
    Image im((uint32_t)width, (uint32_t)height, ColorType::RGB8);
    auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    const size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; ++i) {
      uint8_t* d = &im.data[3 * i];
      if (!ycc) {
        d[0] = plane[0][i]; d[1] = plane[1][i]; d[2] = plane[2][i];
        continue;
      }
      // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: SCALEBITS 16, ONE_HALF 32768, FIX(x) = (int)(x * 65536 + 0.5)
      const int y = plane[0][i], cb = plane[1][i] - 128, cr = plane[2][i] - 128;
      const int cr_r = (91881 * cr + 32768) >> 16, cb_b = (116130 * cb + 32768) >> 16;
      const int g = (-22554 * cb + 32768 + (-46802) * cr) >> 16;
      d[0] = clamp8(y + cr_r);
      d[1] = clamp8(y + g);
      d[2] = clamp8(y + cb_b);
    }
    return im;
  }
};

}  // namespace jpeg_detail

inline Image decode_jpeg(const std::vector<uint8_t>& file) { return jpeg_detail::Decoder(file.data(), file.size()).run(); }

}  // namespace rayca
