// bvh_build.hip -- the BLAS builder on the GPU: the same tree and the same primitive order as BlasBuilder in
// host_scene.cpp, i.e. as Blas::set_primitives_recursive of the reference (rayca-soft/src/bvh/blas.rs:64-123,
// 261-316), for both seeds of the candidate boxes (RAYCA_BUILDER_REFERENCE / RAYCA_BUILDER_SAH).
//
// The reference rebuilds its BVH inside every draw(); its builder is a recursion with an in-place swap
// partition whose exact element order matters downstream (leaf order = primitive index = the tie rule).  Here the
// recursion is run level by level, one workgroup per open node:
//   1. binning: a primitive is left of plane i iff centroid < pos_i and pos_i is monotone in i, so 64 bins per axis
//      (count + box, LDS atomics on order-preserving integer keys) give every one of the 189 candidates' counts
//      and boxes exactly (min/max unions are order independent);
//   2. the 189 costs are evaluated in the reference's order with its arithmetic and strict `<`;
//   3. the swap partition (blas.rs:279-289) is not run but SOLVED: the loop examines the front stream in
//      segments L*R and the back stream in segments R*L alternately, which fixes where every element ends up:
//         left-class at p < nL stays; the k-th right-class in [0,nL) ("hole") goes to n-1 (k = 1) or to one below
//         the (k-1)-th left-class counted from the back ("filler"); the k-th filler goes to the k-th hole; a
//         right-class at p > nL moves to p-1; a right-class at p = nL behaves as hole K+1.
//      Ranks come from block-wide prefix counts, so the permutation is a scatter.
// Checked bit for bit against the host builder (tree, boxes, order) in tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>

#include <thread>

#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "host_scene.hpp"
#include "staging.hpp"

namespace rayca {
namespace {

constexpr int kB = 256;
// k_build_level runs with 256 threads per node, and with 1024 on the levels that hold nodes above kBig primitives: their
// partition is one workgroup's loop over the node's range (levels 0-4 of the atrium: 9.0 ms of the 14 with 256 threads)
constexpr int kBuildLevelMaxBlock = 1024;
#ifndef RAYCA_KBIG
#define RAYCA_KBIG 16384
#endif
constexpr uint32_t kBig = RAYCA_KBIG;    // nodes above this size are binned by many workgroups (k_bin_big), kChunk positions each
constexpr uint32_t kChunk = 4096;
constexpr uint32_t kBinWords = 3 * 64 * 7;  // per node: count + min xyz + max xyz for 64 bins on 3 axes
constexpr uint32_t kSeq = 16;  // subtrees of at most this many primitives are finished by ONE wave (k_build_small)

// The host builder's BuildNode, field for field (host_scene.hpp: two points with w = 1, range, children): the finished
// array is copied straight into the host arena.
struct DNode {
  float a[4], b[4];
  uint32_t offset, count;
  int32_t left, right;
};
static_assert(sizeof(DNode) == sizeof(BuildNode) && offsetof(DNode, b) == offsetof(BuildNode, bounds) + sizeof(F4) &&
              offsetof(DNode, offset) == offsetof(BuildNode, offset) && offsetof(DNode, count) == offsetof(BuildNode, count) &&
              offsetof(DNode, left) == offsetof(BuildNode, left) && offsetof(DNode, right) == offsetof(BuildNode, right),
              "DNode is downloaded into a std::vector<BuildNode>");

// RAYCA_BIG_PARTITION: the swap partition of a node that is binned by many workgroups (more than kBig primitives) is solved by
// many workgroups too -- one per kChunk positions, the chunks k_bin_big already works through -- in a handful of small
// launches per level (k_big_*), instead of by the ONE workgroup that runs the node in k_build_level: that workgroup took
// 2.4 ms for the atrium's root and the top levels were half of the level loop.  Same destinations (same ranks of holes and
// fillers), so the same order.
#ifndef RAYCA_BIG_PARTITION
#define RAYCA_BIG_PARTITION 1
#endif
struct BigSplit {          // one big node of the current level, by its bin slot
  uint32_t node, axis, ok, nl, n, off, holes;
  uint32_t chunk_first;    // its first entry in the level's chunk list (written when the list was made, a level earlier)
  float pos;
  uint32_t box[2][2][3];   // the children's boxes: [child][min / max][xyz], encoded for atomicMin / atomicMax
};
struct BuildState {
  BigSplit* splits[2];   // this level's / the next level's (level parity)
  uint32_t* ch_cnt;      // per chunk of the level's list: holes, fillers in it
  uint32_t* ch_base;     // per chunk: rank of its first hole (ascending), of its last filler (descending)
  const float* cent[3];  // centroid, SoA, by primitive id
  const float* bmin[3];
  const float* bmax[3];
  uint32_t* order;       // primitive ids, partitioned in place (through tmp)
  uint32_t* tmp;
  uint32_t* rank;        // per position: rank of a hole / filler
  uint32_t* hole_pos;    // per node range: position of the k-th hole
  uint32_t* filler_pos;  // per node range: position of the k-th filler
  DNode* nodes;
  uint32_t* big;         // per node: slot of its bins in gbins when it is binned by k_bin_big, else RAYCA_NONE
  uint32_t* node_count;
  uint32_t* small_nodes;  // roots of subtrees left to k_build_small, and their levels
  uint32_t* small_levels;
  uint32_t* small_count;
  uint32_t seed_origin, max_depth;
  uint32_t* gbins;        // [big slot][kBinWords]: bins of the big nodes of the current level
  uint2* chunks[2];       // (node, chunk index) work list of k_bin_big, this level / next level
  uint32_t* chunk_count;  // entries appended to the next level's list
  uint32_t* slot_count;   // big slots handed out for the next level
};

// order-preserving map float -> uint (for LDS atomicMin/atomicMax)
__device__ __forceinline__ uint32_t enc(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__device__ __forceinline__ float area3(const float* a, const float* b) {  // aabb.rs:20-23
  const float ex = b[0] - a[0], ey = b[1] - a[1], ez = b[2] - a[2];
  return ex * ey + ey * ez + ez * ex;
}

// exclusive prefix count of `flag` over the block (thread order), and the block total
__device__ __forceinline__ uint32_t block_prefix(bool flag, uint32_t* wave_tot, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();  // wave_tot may still be read from the previous call
  if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0, tot = 0;
  for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
    if (w < wave) before += wave_tot[w];
    tot += wave_tot[w];
  }
  total = tot;
  return before + in_wave;
}

__device__ __forceinline__ void init_bins(const DNode& nd, uint32_t (*s_cnt)[64], uint32_t (*s_min)[64][3], uint32_t (*s_max)[64][3], float (*s_pos)[64]) {
  for (uint32_t i = threadIdx.x; i < 3 * 64; i += blockDim.x) {
    const uint32_t a = i / 64, b = i % 64;
    s_cnt[a][b] = 0;
    for (int c = 0; c < 3; ++c) {
      s_min[a][b][c] = enc(FLT_MAX);
      s_max[a][b][c] = enc(-FLT_MAX);
    }
    const float lo = nd.a[a], hi = nd.b[a];
    const float scale = (hi - lo) / 64.0f;
    s_pos[a][b] = b == 0 ? -FLT_MAX : lo + (float)b * scale;
  }
}

// positions [begin, end) of the node's range into the 3 x 64 bins
__device__ __forceinline__ void bin_range(const BuildState& st, const DNode& nd, const bool* valid, uint32_t (*s_cnt)[64], uint32_t (*s_min)[64][3],
                                          uint32_t (*s_max)[64][3], float (*s_pos)[64], uint32_t begin, uint32_t end) {
  for (uint32_t s = begin + threadIdx.x; s < end; s += blockDim.x) {
    const uint32_t id = st.order[nd.offset + s];
    float mn[3], mx[3];
    for (int c = 0; c < 3; ++c) {
      mn[c] = st.bmin[c][id];
      mx[c] = st.bmax[c][id];
    }
    for (int a = 0; a < 3; ++a) {
      if (!valid[a]) continue;
      const float c = st.cent[a][id];
      int k;
      if (c != c) k = 63;
      else {
        int lo = 1, hi = 64;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_pos[a][mid] <= c) lo = mid + 1;
          else hi = mid;
        }
        k = lo - 1;
      }
      atomicAdd(&s_cnt[a][k], 1u);
      for (int cc = 0; cc < 3; ++cc) {
        atomicMin(&s_min[a][k][cc], enc(mn[cc]));
        atomicMax(&s_max[a][k][cc], enc(mx[cc]));
      }
    }
  }
}

// bins of the big nodes: one workgroup per kChunk positions, merged into the node's global bins
// The level kernels are launched with grids that BOUND the level's work -- the host does not wait for the previous level
// to learn the exact counts -- and take the counts themselves from `lc`, the record of their level: {open nodes, chunks of
// big nodes, big slots}, written by the level before (k_build_level's atomics on the next record; zeroed once per build).
__global__ __launch_bounds__(kB) void k_bin_init(BuildState st, const uint32_t* lc) {
  const uint32_t slots = lc[2];
  const uint32_t i = blockIdx.x * kB + threadIdx.x;
  if (i >= slots * 3 * 64) return;
  uint32_t* g = st.gbins + (size_t)i * 7;
  g[0] = 0;
  for (int c = 0; c < 3; ++c) {
    g[1 + c] = enc(FLT_MAX);
    g[4 + c] = enc(-FLT_MAX);
  }
}
__global__ __launch_bounds__(kB) void k_bin_big(BuildState st, const uint2* chunks, const uint32_t* lc) {
  __shared__ uint32_t s_cnt[3][64];
  __shared__ uint32_t s_min[3][64][3], s_max[3][64][3];
  __shared__ float s_pos[3][64];
  if (blockIdx.x >= lc[1]) return;   // (block-uniform: before any barrier)
  const uint2 job = chunks[blockIdx.x];
  const DNode nd = st.nodes[job.x];
  init_bins(nd, s_cnt, s_min, s_max, s_pos);
  __syncthreads();
  bool valid[3];
  for (int a = 0; a < 3; ++a) valid[a] = nd.a[a] != nd.b[a];
  const uint32_t begin = job.y * kChunk, end = min(nd.count, begin + kChunk);
  bin_range(st, nd, valid, s_cnt, s_min, s_max, s_pos, begin, end);
  __syncthreads();
  uint32_t* g = st.gbins + (size_t)st.big[job.x] * kBinWords;
  for (uint32_t i = threadIdx.x; i < 3 * 64; i += blockDim.x) {
    const uint32_t a = i / 64, b = i % 64;
    if (s_cnt[a][b] == 0) continue;
    atomicAdd(&g[i * 7], s_cnt[a][b]);
    for (int c = 0; c < 3; ++c) {
      atomicMin(&g[i * 7 + 1 + c], s_min[a][b][c]);
      atomicMax(&g[i * 7 + 4 + c], s_max[a][b][c]);
    }
  }
}

// the two children of `node_id` (range [off, off + n) split at nl): arena nodes, big slots and chunk lists of the big ones,
// and where each goes next -- the next level's list, or k_build_small.  box: [child][min / max][xyz], encoded.  One thread.
__device__ void make_children(const BuildState& st, uint32_t node_id, uint32_t off, uint32_t n, uint32_t nl, const uint32_t* box, uint32_t* next,
                              uint32_t* next_count, uint2* next_chunks, uint32_t level) {
  const uint32_t base = atomicAdd(st.node_count, 2u);
  DNode l, r;
  for (int c = 0; c < 3; ++c) {
    l.a[c] = dec(box[0 * 6 + 0 * 3 + c]); l.b[c] = dec(box[0 * 6 + 1 * 3 + c]);
    r.a[c] = dec(box[1 * 6 + 0 * 3 + c]); r.b[c] = dec(box[1 * 6 + 1 * 3 + c]);
  }
  l.a[3] = l.b[3] = r.a[3] = r.b[3] = 1.0f;
  l.offset = off; l.count = nl; l.left = l.right = -1;
  r.offset = off + nl; r.count = n - nl; r.left = r.right = -1;
  const uint32_t big2[2] = {nl > kBig ? atomicAdd(st.slot_count, 1u) : RAYCA_NONE, n - nl > kBig ? atomicAdd(st.slot_count, 1u) : RAYCA_NONE};
  st.big[base] = big2[0];
  st.big[base + 1] = big2[1];
  for (uint32_t c = 0; c < 2; ++c) {
    const DNode& ch = c ? r : l;
    if (big2[c] == RAYCA_NONE) continue;
    const uint32_t nch = (ch.count + kChunk - 1) / kChunk;
    const uint32_t at = atomicAdd(st.chunk_count, nch);
    for (uint32_t k = 0; k < nch; ++k) next_chunks[at + k] = make_uint2(base + c, k);
    st.splits[(level + 1u) & 1u][big2[c]].chunk_first = at;
  }
  st.nodes[base] = l;
  st.nodes[base + 1] = r;
  st.nodes[node_id].left = (int32_t)base;
  st.nodes[node_id].right = (int32_t)base + 1;
  st.nodes[node_id].count = 0;  // inner
  for (uint32_t c = 0; c < 2; ++c) {
    const uint32_t cn = c ? n - nl : nl;
    if (cn <= kSeq) {
      if (cn > 1) {  // a single primitive can never be split: nothing left to do
        const uint32_t slot = atomicAdd(st.small_count, 1u);
        st.small_nodes[slot] = base + c;
        st.small_levels[slot] = level + 1u;
      }
    } else {
      next[atomicAdd(next_count, 1u)] = base + c;
    }
  }
}

__global__ __launch_bounds__(kBuildLevelMaxBlock) void k_build_level(BuildState st, const uint32_t* active, uint32_t* next, uint32_t* next_count, uint2* next_chunks, uint32_t level,
                                                                  const uint32_t* lc) {
  if (blockIdx.x >= lc[0]) return;   // (block-uniform: before any barrier)
  __shared__ uint32_t s_cnt[3][64];
  __shared__ uint32_t s_min[3][64][3], s_max[3][64][3];
  __shared__ float s_pos[3][64];
  __shared__ float s_cost[3], s_split[3];
  __shared__ uint32_t s_wave[kBuildLevelMaxBlock / 64];
  __shared__ uint32_t s_red[2][2][3];  // [child][min/max][xyz], encoded
  __shared__ int s_axis;
  __shared__ float s_best_pos;
  __shared__ uint32_t s_split_ok, s_left;

  const uint32_t tid = threadIdx.x;
  const uint32_t node_id = active[blockIdx.x];
  const DNode nd = st.nodes[node_id];
  const uint32_t off = nd.offset, n = nd.count;

  // ---- 1. bins ----------------------------------------------------------------------------------------------
  init_bins(nd, s_cnt, s_min, s_max, s_pos);
  __syncthreads();
  bool valid[3];
  for (int a = 0; a < 3; ++a) valid[a] = nd.a[a] != nd.b[a];
  const uint32_t nd_big = st.big[node_id];
  if (nd_big != RAYCA_NONE) {  // binned by k_bin_big: fetch
    const uint32_t* g = st.gbins + (size_t)nd_big * kBinWords;
    for (uint32_t i = tid; i < 3 * 64; i += blockDim.x) {
      const uint32_t a = i / 64, b = i % 64;
      s_cnt[a][b] = g[i * 7];
      for (int c = 0; c < 3; ++c) {
        s_min[a][b][c] = g[i * 7 + 1 + c];
        s_max[a][b][c] = g[i * 7 + 4 + c];
      }
    }
  } else {
    bin_range(st, nd, valid, s_cnt, s_min, s_max, s_pos, 0, n);
  }
  __syncthreads();

  // ---- 2. the 63 candidates of each axis (one thread per axis), reference order and arithmetic -----------------
  if (tid < 3) {
    const int a = (int)tid;
    float best = FLT_MAX, best_pos = 0.0f;
    if (valid[a]) {
      // prefix boxes/counts: L[i] = seed U bins[0..i-1]; suffix: R[i] = seed U bins[i..63]
      float rmin[65][3], rmax[65][3];
      uint32_t rcnt[65];
      for (int c = 0; c < 3; ++c) {
        rmin[64][c] = st.seed_origin ? 0.0f : FLT_MAX;
        rmax[64][c] = st.seed_origin ? 0.0f : -FLT_MAX;
      }
      rcnt[64] = 0;
      for (int b = 63; b >= 0; --b) {
        rcnt[b] = rcnt[b + 1] + s_cnt[a][b];
        for (int c = 0; c < 3; ++c) {
          rmin[b][c] = rmin[b + 1][c];
          rmax[b][c] = rmax[b + 1][c];
          if (s_cnt[a][b]) {
            rmin[b][c] = fminf(rmin[b][c], dec(s_min[a][b][c]));
            rmax[b][c] = fmaxf(rmax[b][c], dec(s_max[a][b][c]));
          }
        }
      }
      float lmin[3], lmax[3];
      for (int c = 0; c < 3; ++c) {
        lmin[c] = st.seed_origin ? 0.0f : FLT_MAX;
        lmax[c] = st.seed_origin ? 0.0f : -FLT_MAX;
      }
      uint32_t lcnt = 0;
      for (int i = 1; i < 64; ++i) {
        const int b = i - 1;  // L[i] adds bin i-1
        lcnt += s_cnt[a][b];
        if (s_cnt[a][b])
          for (int c = 0; c < 3; ++c) {
            lmin[c] = fminf(lmin[c], dec(s_min[a][b][c]));
            lmax[c] = fmaxf(lmax[c], dec(s_max[a][b][c]));
          }
        if (!st.seed_origin && (lcnt == 0 || rcnt[i] == 0)) continue;
        float cost = (float)lcnt * area3(lmin, lmax) + (float)rcnt[i] * area3(rmin[i], rmax[i]);
        if (!(cost > 0.0f)) cost = FLT_MAX;
        if (cost < best) {
          best = cost;
          best_pos = s_pos[a][i];
        }
      }
    }
    s_cost[a] = best;
    s_split[a] = best_pos;
  }
  __syncthreads();
  if (tid == 0) {
    float best = FLT_MAX;
    int axis = 0;
    float pos = 0.0f;
    for (int a = 0; a < 3; ++a)
      if (s_cost[a] < best) {
        best = s_cost[a];
        axis = a;
        pos = s_split[a];
      }
    const float no_split = (float)n * area3(nd.a, nd.b);
    s_axis = axis;
    s_best_pos = pos;
    s_split_ok = best > no_split ? 0u : 1u;
    s_left = 0;
  }
  __syncthreads();
#if RAYCA_BIG_PARTITION
  if (nd_big != RAYCA_NONE) {  // the decision only: k_big_* partition the range and make the children
    if (tid == 0) {
      BigSplit& d = st.splits[level & 1u][nd_big];
      d.node = node_id; d.axis = (uint32_t)s_axis; d.ok = s_split_ok; d.nl = 0u; d.n = n; d.off = off; d.holes = 0u;
      d.pos = s_best_pos;
      for (int c = 0; c < 2; ++c)
        for (int k = 0; k < 3; ++k) {
          d.box[c][0][k] = enc(FLT_MAX);
          d.box[c][1][k] = enc(-FLT_MAX);
        }
    }
    return;
  }
#endif
  if (!s_split_ok) return;  // a leaf: primitives stay as they are

  // ---- 3. the swap partition, solved ------------------------------------------------------------------------------
  const float* cax = st.cent[s_axis];
  const float pos = s_best_pos;
  uint32_t part = 0;
  for (uint32_t s = tid; s < n; s += blockDim.x) part += cax[st.order[off + s]] < pos ? 1u : 0u;
  atomicAdd(&s_left, part);
  __syncthreads();
  const uint32_t nl = s_left;
  if (nl == 0 || nl == n) {
    // everything on one side.  All left: the loop never swaps.  All right: it rotates the range (first element to
    // the end, the others down by one) -- and the node stays a leaf either way (blas.rs:291-293).
    if (nl == 0 && n > 1) {
      for (uint32_t s = tid; s < n; s += blockDim.x) st.tmp[off + (s == 0 ? n - 1 : s - 1)] = st.order[off + s];
      __syncthreads();
      for (uint32_t s = tid; s < n; s += blockDim.x) st.order[off + s] = st.tmp[off + s];
    }
    return;
  }
  // holes: right-class at p < nl, ranked ascending
  uint32_t run = 0;
  for (uint32_t base = 0; base < nl; base += blockDim.x) {
    const uint32_t p = base + tid;
    const bool is_hole = p < nl && !(cax[st.order[off + p]] < pos);
    uint32_t tot;
    const uint32_t r = run + block_prefix(is_hole, s_wave, tot);
    if (is_hole) {
      st.hole_pos[off + r] = p;
      st.rank[off + p] = r;
    }
    run += tot;
  }
  const uint32_t K = run;  // holes == fillers
  // fillers: left-class at p >= nl, ranked descending
  run = 0;
  for (uint32_t base = 0; base < n - nl; base += blockDim.x) {
    const uint32_t q = base + tid;            // distance from the end
    const bool in = q < n - nl;
    const uint32_t p = in ? n - 1 - q : 0;
    const bool is_filler = in && cax[st.order[off + p]] < pos;
    uint32_t tot;
    const uint32_t r = run + block_prefix(is_filler, s_wave, tot);
    if (is_filler) {
      st.filler_pos[off + r] = p;
      st.rank[off + p] = r;
    }
    run += tot;
  }
  __syncthreads();
  for (uint32_t p = tid; p < n; p += blockDim.x) {
    const uint32_t id = st.order[off + p];
    const bool left = cax[id] < pos;
    uint32_t dest;
    if (p < nl) {
      if (left) dest = p;
      else {
        const uint32_t k = st.rank[off + p];
        dest = k == 0 ? n - 1 : st.filler_pos[off + k - 1] - 1;
      }
    } else if (left) {
      dest = st.hole_pos[off + st.rank[off + p]];
    } else if (p == nl) {
      dest = (K == 0 ? n : st.filler_pos[off + K - 1]) - 1;
    } else {
      dest = p - 1;
    }
    st.tmp[off + dest] = id;
  }
  __syncthreads();
  for (uint32_t p = tid; p < n; p += blockDim.x) st.order[off + p] = st.tmp[off + p];
  // ---- 4. children --------------------------------------------------------------------------------------------------
  if (tid < 12) {
    const uint32_t child = tid / 6, mm = (tid / 3) & 1u, c = tid % 3;
    s_red[child][mm][c] = mm ? enc(-FLT_MAX) : enc(FLT_MAX);
  }
  __syncthreads();
  for (uint32_t p = tid; p < n; p += blockDim.x) {
    const uint32_t id = st.tmp[off + p];
    const uint32_t child = p < nl ? 0u : 1u;
    for (int c = 0; c < 3; ++c) {
      atomicMin(&s_red[child][0][c], enc(st.bmin[c][id]));
      atomicMax(&s_red[child][1][c], enc(st.bmax[c][id]));
    }
  }
  __syncthreads();
  if (tid == 0) make_children(st, node_id, off, n, nl, &s_red[0][0][0], next, next_count, next_chunks, level);
}

#if RAYCA_BIG_PARTITION
// ---- the partition of the big nodes, one workgroup per chunk (the level's chunk list; lc[1] entries, lc[2] slots) ---------------
__device__ __forceinline__ bool big_job(const BuildState& st, const uint2* chunks, const uint32_t* lc, uint32_t level, uint2& job, BigSplit*& d, uint32_t& begin, uint32_t& end) {
  if (blockIdx.x >= lc[1]) return false;   // (block-uniform)
  job = chunks[blockIdx.x];
  d = &st.splits[level & 1u][st.big[job.x]];
  begin = job.y * kChunk;
  end = min(d->n, begin + kChunk);
  return d->ok != 0u;
}
// left-class primitives of the node: nl
__global__ __launch_bounds__(kB) void k_big_count(BuildState st, const uint2* chunks, const uint32_t* lc, uint32_t level) {
  __shared__ uint32_t s_left;
  uint2 job; BigSplit* d; uint32_t begin, end;
  if (!big_job(st, chunks, lc, level, job, d, begin, end)) return;
  if (threadIdx.x == 0) s_left = 0;
  __syncthreads();
  const float* cax = st.cent[d->axis];
  const float pos = d->pos;
  uint32_t part = 0;
  for (uint32_t p = begin + threadIdx.x; p < end; p += blockDim.x) part += cax[st.order[d->off + p]] < pos ? 1u : 0u;
  if (part) atomicAdd(&s_left, part);
  __syncthreads();
  if (threadIdx.x == 0 && s_left) atomicAdd(&d->nl, s_left);
}
// holes (right-class below nl) and fillers (left-class from nl on) of every chunk
__global__ __launch_bounds__(kB) void k_big_classify(BuildState st, const uint2* chunks, const uint32_t* lc, uint32_t level) {
  __shared__ uint32_t s_h, s_f;
  uint2 job; BigSplit* d; uint32_t begin, end;
  if (!big_job(st, chunks, lc, level, job, d, begin, end)) return;
  const uint32_t nl = d->nl;
  if (nl == 0u || nl == d->n) return;
  if (threadIdx.x == 0) s_h = s_f = 0;
  __syncthreads();
  const float* cax = st.cent[d->axis];
  const float pos = d->pos;
  uint32_t h = 0, f = 0;
  for (uint32_t p = begin + threadIdx.x; p < end; p += blockDim.x) {
    const bool left = cax[st.order[d->off + p]] < pos;
    h += (p < nl && !left) ? 1u : 0u;
    f += (p >= nl && left) ? 1u : 0u;
  }
  if (h) atomicAdd(&s_h, h);
  if (f) atomicAdd(&s_f, f);
  __syncthreads();
  if (threadIdx.x == 0) {
    st.ch_cnt[2u * blockIdx.x] = s_h;
    st.ch_cnt[2u * blockIdx.x + 1u] = s_f;
  }
}
// per node: rank of every chunk's first hole (holes ranked ascending) and of its last filler (fillers ranked from the end)
__global__ void k_big_scan(BuildState st, const uint32_t* lc, uint32_t level) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= lc[2]) return;
  BigSplit& d = st.splits[level & 1u][slot];
  if (!d.ok || d.nl == 0u || d.nl == d.n) return;
  const uint32_t nch = (d.n + kChunk - 1) / kChunk, cf = d.chunk_first;
  uint32_t run = 0;
  for (uint32_t k = 0; k < nch; ++k) {
    st.ch_base[2u * (cf + k)] = run;
    run += st.ch_cnt[2u * (cf + k)];
  }
  d.holes = run;
  run = 0;
  for (uint32_t k = nch; k-- > 0;) {
    st.ch_base[2u * (cf + k) + 1u] = run;
    run += st.ch_cnt[2u * (cf + k) + 1u];
  }
}
// ranks and positions of the holes and fillers (what the one-workgroup form keeps in rank / hole_pos / filler_pos)
__global__ __launch_bounds__(kB) void k_big_rank(BuildState st, const uint2* chunks, const uint32_t* lc, uint32_t level) {
  __shared__ uint32_t s_wave[kB / 64];
  uint2 job; BigSplit* d; uint32_t begin, end;
  if (!big_job(st, chunks, lc, level, job, d, begin, end)) return;
  const uint32_t nl = d->nl, off = d->off;
  if (nl == 0u || nl == d->n) return;
  const float* cax = st.cent[d->axis];
  const float pos = d->pos;
  uint32_t run = st.ch_base[2u * blockIdx.x];
  for (uint32_t base = begin; base < end; base += blockDim.x) {   // holes, ascending
    const uint32_t p = base + threadIdx.x;
    const bool is_hole = p < end && p < nl && !(cax[st.order[off + p]] < pos);
    uint32_t tot;
    const uint32_t r = run + block_prefix(is_hole, s_wave, tot);
    if (is_hole) {
      st.hole_pos[off + r] = p;
      st.rank[off + p] = r;
    }
    run += tot;
  }
  run = st.ch_base[2u * blockIdx.x + 1u];
  for (uint32_t back = 0; back < end - begin; back += blockDim.x) {   // fillers, from the chunk's end down
    const uint32_t q = back + threadIdx.x;
    const bool in = q < end - begin;
    const uint32_t p = in ? end - 1u - q : 0u;
    const bool is_filler = in && p >= nl && cax[st.order[off + p]] < pos;
    uint32_t tot;
    const uint32_t r = run + block_prefix(is_filler, s_wave, tot);
    if (is_filler) {
      st.filler_pos[off + r] = p;
      st.rank[off + p] = r;
    }
    run += tot;
  }
}
// every primitive to its place (tmp)
__global__ __launch_bounds__(kB) void k_big_scatter(BuildState st, const uint2* chunks, const uint32_t* lc, uint32_t level) {
  uint2 job; BigSplit* d; uint32_t begin, end;
  if (!big_job(st, chunks, lc, level, job, d, begin, end)) return;
  const uint32_t nl = d->nl, n = d->n, off = d->off, K = d->holes;
  if (nl == n) return;   // everything left of the plane: the loop never swaps
  const float* cax = st.cent[d->axis];
  const float pos = d->pos;
  for (uint32_t p = begin + threadIdx.x; p < end; p += blockDim.x) {
    const uint32_t id = st.order[off + p];
    uint32_t dest;
    if (nl == 0u) {   // everything right: the loop rotates the range (first element to the end, the others down by one)
      dest = p == 0u ? n - 1u : p - 1u;
    } else {
      const bool left = cax[id] < pos;
      if (p < nl) {
        if (left) dest = p;
        else {
          const uint32_t k = st.rank[off + p];
          dest = k == 0 ? n - 1 : st.filler_pos[off + k - 1] - 1;
        }
      } else if (left) {
        dest = st.hole_pos[off + st.rank[off + p]];
      } else if (p == nl) {
        dest = (K == 0 ? n : st.filler_pos[off + K - 1]) - 1;
      } else {
        dest = p - 1;
      }
    }
    st.tmp[off + dest] = id;
  }
}
// back into the order, and the children's boxes
__global__ __launch_bounds__(kB) void k_big_finish(BuildState st, const uint2* chunks, const uint32_t* lc, uint32_t level) {
  __shared__ uint32_t s_red[2][2][3];
  uint2 job; BigSplit* d; uint32_t begin, end;
  if (!big_job(st, chunks, lc, level, job, d, begin, end)) return;
  const uint32_t nl = d->nl, n = d->n, off = d->off;
  if (nl == n) return;
  if (threadIdx.x < 12) {
    const uint32_t child = threadIdx.x / 6, mm = (threadIdx.x / 3) & 1u, c = threadIdx.x % 3;
    s_red[child][mm][c] = mm ? enc(-FLT_MAX) : enc(FLT_MAX);
  }
  __syncthreads();
  for (uint32_t p = begin + threadIdx.x; p < end; p += blockDim.x) {
    const uint32_t id = st.tmp[off + p];
    st.order[off + p] = id;
    if (nl != 0u) {
      const uint32_t child = p < nl ? 0u : 1u;
      for (int c = 0; c < 3; ++c) {
        atomicMin(&s_red[child][0][c], enc(st.bmin[c][id]));
        atomicMax(&s_red[child][1][c], enc(st.bmax[c][id]));
      }
    }
  }
  __syncthreads();
  if (nl != 0u && threadIdx.x < 12) {
    const uint32_t child = threadIdx.x / 6, mm = (threadIdx.x / 3) & 1u, c = threadIdx.x % 3;
    if (mm) atomicMax(&d->box[child][1][c], s_red[child][1][c]);
    else atomicMin(&d->box[child][0][c], s_red[child][0][c]);
  }
}
// the children of every big node that was split
__global__ void k_big_children(BuildState st, uint32_t* next, uint32_t* next_count, uint2* next_chunks, uint32_t level, const uint32_t* lc) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= lc[2]) return;
  const BigSplit& d = st.splits[level & 1u][slot];
  if (!d.ok || d.nl == 0u || d.nl == d.n) return;   // a leaf (blas.rs:291-293 for the one-sided cases)
  make_children(st, d.node, d.off, d.n, d.nl, &d.box[0][0][0], next, next_count, next_chunks, level);
}
#endif

// A subtree of at most kSeq primitives, finished by one WAVE: the reference's recursion as it stands -- every candidate
// plane evaluated by a loop over the primitives (evaluate_sah, blas.rs:64-89: the same counts and boxes the bins give,
// min/max being order independent), the swap partition run literally (blas.rs:279-289).  Lane i prices plane i of an
// axis (i = 1..63; the same operations on the same operands as a loop over the planes), the wave keeps the cheapest -- the
// FIRST cheapest in (axis, plane) order, as the loop's strict `<` does -- and lane 0 runs the partition and the children's
// boxes, which are a dozen steps.  The primitives of the subtree sit in LDS, where every lane reads the same one at a time.
// (One thread per subtree, with its sixteen primitives in private arrays, took 5.4 ms for the atrium's 24 000 subtrees.)
__global__ __launch_bounds__(64) void k_build_small(BuildState st, uint32_t n_small) {
  const uint32_t t = blockIdx.x;  // one subtree per 64-thread block
  if (t >= n_small) return;
  const uint32_t lane = threadIdx.x;
  __shared__ uint32_t ids[kSeq];
  __shared__ float cen[3][kSeq], mn[3][kSeq], mx[3][kSeq];
  __shared__ uint32_t stack_node[kSeq + 1], stack_level[kSeq + 1];
  __shared__ uint32_t s_nl;   // lane 0 -> wave: left count of the split just made (0: none was made)
  // The subtree's nodes are made HERE, in LDS -- [0] a copy of its root, then the children in pairs, linked by local indices --
  // and go to the arena when the subtree is finished: ONE atomicAdd on the arena's node counter per subtree (and nodes that
  // are popped come from LDS).  One add per split was 360 000 of them on one address for the atrium's 24 014 subtrees.
  __shared__ DNode ln[2 * kSeq];
  __shared__ uint32_t s_made, s_base;
  const uint32_t root = st.small_nodes[t];
  const uint32_t root_level = st.small_levels[t];
  const uint32_t base_off = st.nodes[root].offset, total = st.nodes[root].count;
  if (lane < total) {
    const uint32_t id = st.order[base_off + lane];
    ids[lane] = id;
    for (int c = 0; c < 3; ++c) {
      cen[c][lane] = st.cent[c][id];
      mn[c][lane] = st.bmin[c][id];
      mx[c][lane] = st.bmax[c][id];
    }
  }
  if (lane == 0) {
    ln[0] = st.nodes[root];
    stack_node[0] = 0;
    stack_level[0] = root_level;
    s_made = 0;
  }
  __syncthreads();
  int sp = 1;  // wave-uniform: every lane tracks it
  while (sp > 0) {
    --sp;
    const uint32_t node_id = stack_node[sp], level = stack_level[sp];   // (node_id: index into ln)
    if (level >= st.max_depth) continue;
    const DNode nd = ln[node_id];
    const uint32_t lo = nd.offset - base_off, n = nd.count;
    // find_best_split_plane  blas.rs:93-123
    float best = FLT_MAX, best_pos = 0.0f;
    int best_axis = 0;
    for (int a = 0; a < 3; ++a) {
      const float bmin = nd.a[a], bmax = nd.b[a];
      if (bmin == bmax) continue;
      const float scale = (bmax - bmin) / 64.0f;
      float cost = FLT_MAX, pos = 0.0f;
      if (lane >= 1) {  // plane `lane`
        pos = bmin + (float)lane * scale;
        float la[3], lb[3], ra[3], rb[3];
        for (int c = 0; c < 3; ++c) {
          la[c] = ra[c] = st.seed_origin ? 0.0f : FLT_MAX;
          lb[c] = rb[c] = st.seed_origin ? 0.0f : -FLT_MAX;
        }
        uint32_t lc = 0, rc = 0;
        for (uint32_t k = lo; k < lo + n; ++k) {
          if (cen[a][k] < pos) {
            ++lc;
            for (int c = 0; c < 3; ++c) {
              la[c] = fminf(la[c], mn[c][k]);
              lb[c] = fmaxf(lb[c], mx[c][k]);
            }
          } else {
            ++rc;
            for (int c = 0; c < 3; ++c) {
              ra[c] = fminf(ra[c], mn[c][k]);
              rb[c] = fmaxf(rb[c], mx[c][k]);
            }
          }
        }
        if (st.seed_origin || (lc != 0 && rc != 0)) {
          cost = (float)lc * area3(la, lb) + (float)rc * area3(ra, rb);
          if (!(cost > 0.0f)) cost = FLT_MAX;
        }
      }
      // the cheapest plane of this axis, the lowest plane index among equals (what a loop over i = 1..63 with `<` keeps)
      float m = cost;
      for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off));
      const unsigned long long at = __ballot(cost == m);
      const int first = __ffsll((long long)at) - 1;
      const float axis_best = __shfl(cost, first), axis_pos = __shfl(pos, first);
      if (axis_best < best) {  // (FLT_MAX never beats the initial FLT_MAX: an axis without a usable plane changes nothing)
        best = axis_best;
        best_axis = a;
        best_pos = axis_pos;
      }
    }
    const float no_split = (float)n * area3(nd.a, nd.b);
    if (best > no_split) continue;
    if (lane == 0) {
      // the swap partition, literally
      uint32_t i = lo, j = lo + n;
      while (i < j) {
        if (cen[best_axis][i] < best_pos) {
          ++i;
        } else {
          const uint32_t q = j - 1;
          const uint32_t tid_ = ids[i]; ids[i] = ids[q]; ids[q] = tid_;
          for (int c = 0; c < 3; ++c) {
            float f = cen[c][i]; cen[c][i] = cen[c][q]; cen[c][q] = f;
            f = mn[c][i]; mn[c][i] = mn[c][q]; mn[c][q] = f;
            f = mx[c][i]; mx[c][i] = mx[c][q]; mx[c][q] = f;
          }
          --j;
        }
      }
      const uint32_t nl = i - lo, nr = n - nl;
      uint32_t made = 0;
      if (nl != 0 && nr != 0) {
        const uint32_t base = 1u + s_made;   // local
        s_made += 2u;
        DNode l, r;
        for (int c = 0; c < 3; ++c) {
          l.a[c] = FLT_MAX; l.b[c] = -FLT_MAX;
          r.a[c] = FLT_MAX; r.b[c] = -FLT_MAX;
        }
        for (uint32_t k = lo; k < lo + nl; ++k)
          for (int c = 0; c < 3; ++c) {
            l.a[c] = fminf(l.a[c], mn[c][k]);
            l.b[c] = fmaxf(l.b[c], mx[c][k]);
          }
        for (uint32_t k = lo + nl; k < lo + n; ++k)
          for (int c = 0; c < 3; ++c) {
            r.a[c] = fminf(r.a[c], mn[c][k]);
            r.b[c] = fmaxf(r.b[c], mx[c][k]);
          }
        l.a[3] = l.b[3] = r.a[3] = r.b[3] = 1.0f;
        l.offset = nd.offset; l.count = nl; l.left = l.right = -1;
        r.offset = nd.offset + nl; r.count = nr; r.left = r.right = -1;
        ln[base] = l;
        ln[base + 1] = r;
        ln[node_id].left = (int32_t)base;
        ln[node_id].right = (int32_t)base + 1;
        ln[node_id].count = 0;
        int p = sp;
        if (nr > 1) { stack_node[p] = base + 1; stack_level[p] = level + 1; ++p; }
        if (nl > 1) { stack_node[p] = base; stack_level[p] = level + 1; ++p; }
        made = nl;
      }
      s_nl = made;
    }
    __syncthreads();
    const uint32_t nl = s_nl;
    if (nl != 0) {  // every lane follows the stack lane 0 has written
      const uint32_t nr = n - nl;
      if (nr > 1) ++sp;
      if (nl > 1) ++sp;
    }
    __syncthreads();
  }
  if (lane < total) st.order[base_off + lane] = ids[lane];
  // the subtree's nodes into the arena: local index k >= 1 becomes arena index base + k - 1
  const uint32_t made = s_made;
  if (made != 0u) {
    if (lane == 0) s_base = atomicAdd(st.node_count, made);
    __syncthreads();
    const uint32_t base = s_base;
    if (lane < made) {
      DNode d = ln[1u + lane];
      if (d.left >= 0) {
        d.left = (int32_t)(base + (uint32_t)d.left - 1u);
        d.right = (int32_t)(base + (uint32_t)d.right - 1u);
      }
      st.nodes[base + lane] = d;
      st.big[base + lane] = RAYCA_NONE;
    }
    if (lane == 0) {
      st.nodes[root].left = (int32_t)(base + (uint32_t)ln[0].left - 1u);
      st.nodes[root].right = (int32_t)(base + (uint32_t)ln[0].right - 1u);
      st.nodes[root].count = 0;
    }
  }
}

#define HB_TRY(expr)                                                                       \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      err = std::string("gpu bvh build: ") + #expr + ": " + hipGetErrorString(e__);        \
      cleanup();                                                                           \
      return false;                                                                        \
    }                                                                                      \
  } while (0)

}  // namespace

const void* gpu_builder_any_kernel() { return reinterpret_cast<const void*>(k_build_small); }

// ---- binary-node layout on the device (what DevBuilder::emit_blas_flat does on the host) ----------------------------------
// The layout numbers the nodes in pre-order -- a node, its left subtree, its right subtree -- and a leaf above 64 primitives
// becomes a chain of nodes.  Sizes and stack needs go up the tree (the second child to arrive at a parent finishes it), a
// node's index is the sum over its ancestors of 1 (+ the left sibling's size where the path turns right), and then every
// node is written on its own.
struct LayoutState {
  const DNode* nodes;
  uint32_t node_count;
  uint32_t* parent;   // arena index of the parent, RAYCA_NONE for the root
  uint4* rec;         // per node: x = DevNodes of the subtree, y = pending stack entries of a traversal of it, z = primitives
                      // below it, w = bits of its expected cost as laid out (triangle tests x box area) -- one 16-B record,
                      // written once by the thread that finishes the node and read by the one that finishes its parent
  uint32_t* arrived;  // children that have reported to this node
  uint32_t* index;    // layout index of the node (of the head of its chain), root = 0
  uint32_t* leafed;   // 1: the subtree is laid out as ONE leaf (its primitives are a contiguous range of the order)
  // subtrees that are no dearer as one leaf are laid out as one (host_scene.hpp kLeafNodeCost; the same arithmetic as
  // DevBuilder::plan_leaves on the host); node_cost 0: the tree as built
  float node_cost;
  uint32_t leaf_max;
};
__device__ __forceinline__ uint32_t chain_nodes(uint32_t count) { return count > kLeafMaxPrims ? (count + kLeafMaxPrims - 1u) / kLeafMaxPrims - 1u : 0u; }

__global__ void k_layout_parents(LayoutState ls) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ls.node_count) return;
  if (i == 0) ls.parent[0] = RAYCA_NONE;
  ls.arrived[i] = 0u;
  const DNode nd = ls.nodes[i];
  if (nd.left >= 0) {
    ls.parent[nd.left] = i;
    ls.parent[nd.right] = i;
  }
}
// The records that travel between threads of DIFFERENT workgroups (a child's record is read by whoever finishes its parent)
// are written and read with agent-scope relaxed atomics -- accesses that are coherent across the XCDs' L2s on their own -- and
// ordered by hand: a record's stores are acknowledged (s_waitcnt) before the arrival count is bumped, and the sibling's record
// is fetched after the bump has returned.  An ACQ_REL agent-scope atomic does the same job with a write-back and an
// invalidate of the whole L2 around EVERY bump: k_layout_sizes took 1.5 ms for the atrium's 543 k nodes that way (average
// vector-memory latency 50 000 cycles, profiles/r03_atrium_diag_pmc.txt).
__device__ __forceinline__ void rec_store(uint4* rec, uint4 v) {
  uint32_t* w = reinterpret_cast<uint32_t*>(rec);
  __hip_atomic_store(w + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 rec_load(const uint4* rec) {
  uint32_t* w = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(rec));
  uint4 v;
  v.x = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ void memory_ops_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__global__ void k_layout_sizes(LayoutState ls) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ls.node_count) return;
  const DNode nd = ls.nodes[i];
  if (nd.left >= 0) return;  // leaves start the climb
  const uint32_t chain = chain_nodes(nd.count);
  rec_store(&ls.rec[i], make_uint4(chain, chain ? 1u : 0u, nd.count, __float_as_uint((float)nd.count * area3(nd.a, nd.b))));
  ls.leafed[i] = 0u;
  uint32_t p = ls.parent[i];
  while (p != RAYCA_NONE) {
    // the record above (or the one written at the end of the last trip) has landed before the count is bumped; the second
    // child to arrive reads its sibling's record after its own bump has come back
    memory_ops_done();
    if (__hip_atomic_fetch_add(&ls.arrived[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // the sibling subtree is not finished: its last thread goes on
    memory_ops_done();
    const DNode pn = ls.nodes[p];
    const uint4 l = rec_load(&ls.rec[pn.left]), r = rec_load(&ls.rec[pn.right]);
    const uint32_t pl = l.z, pr = r.z;
    const float area = area3(pn.a, pn.b), as_leaf = (float)(pl + pr) * area, as_split = ls.node_cost * area + __uint_as_float(l.w) + __uint_as_float(r.w);
    // (the whole subtree becomes one leaf reference in its parent: no nodes of its own)
    const bool one_leaf = ls.node_cost > 0.0f && pl + pr <= ls.leaf_max && as_leaf <= as_split;
    ls.leafed[p] = one_leaf ? 1u : 0u;
    rec_store(&ls.rec[p], make_uint4(one_leaf ? 0u : 1u + l.x + r.x, one_leaf ? 0u : 1u + (l.y > r.y ? l.y : r.y), pl + pr, __float_as_uint(one_leaf ? as_leaf : as_split)));
    p = ls.parent[p];
  }
}
__global__ void k_layout_index(LayoutState ls) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ls.node_count) return;
  uint32_t idx = 0, c = i;
  bool inside = false;   // below a subtree that is laid out as one leaf: this node is not laid out at all
  for (uint32_t p = ls.parent[c]; p != RAYCA_NONE; p = ls.parent[c]) {
    const DNode pn = ls.nodes[p];
    idx += 1u + ((uint32_t)pn.right == c ? ls.rec[pn.left].x : 0u);
    inside = inside || ls.leafed[p] != 0u;
    c = p;
  }
  ls.index[i] = idx;
  if (inside) ls.leafed[i] = 2u;
}

// DevBuilder::put_box (host_scene.cpp), same operations in the same order
__device__ __forceinline__ void layout_box(float* q, const DNode& nd, float pad_rel, float pad_abs) {
  float lo[3] = {nd.a[0], nd.a[1], nd.a[2]}, hi[3] = {nd.b[0], nd.b[1], nd.b[2]};
  if (!(nd.a[0] <= nd.b[0])) {
    for (int k = 0; k < 3; ++k) lo[k] = hi[k] = kNowhere;
  } else if (pad_rel > 0.0f) {
    for (int k = 0; k < 3; ++k) {
      const float m = fmaxf(fmaxf(fabsf(nd.a[k]), fabsf(nd.b[k])), nd.b[k] - nd.a[k]);
      const float pad = m * pad_rel + pad_abs;
      lo[k] = nd.a[k] - pad;
      hi[k] = nd.b[k] + pad;
    }
  }
  q[0] = lo[0]; q[1] = lo[1]; q[2] = lo[2]; q[3] = hi[0]; q[4] = hi[1]; q[5] = hi[2];
}
__device__ __forceinline__ uint32_t layout_leaf_ref(uint32_t first, uint32_t count) { return kLeafFlag | ((count - 1u) << 25) | first; }
__global__ void k_layout_emit(LayoutState ls, DevNode* out, uint32_t first, uint32_t prim_base, float pad_rel, float pad_abs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ls.node_count) return;
  const DNode nd = ls.nodes[i];
  if (nd.left >= 0) {
    if (ls.leafed[i] != 0u) return;   // the root of, or inside, a subtree that is one leaf: the reference sits in a node above
    DevNode d;
    const DNode l = ls.nodes[nd.left], r = ls.nodes[nd.right];
    layout_box(d.q, l, pad_rel, pad_abs);
    layout_box(d.q + 6, r, pad_rel, pad_abs);
    const DNode* kids[2] = {&l, &r};
    uint32_t refs[2];
    for (int c = 0; c < 2; ++c) {
      const DNode& k = *kids[c];
      const uint32_t ki = (uint32_t)(c ? nd.right : nd.left);
      if (k.left >= 0) refs[c] = ls.leafed[ki] == 1u ? layout_leaf_ref(prim_base + k.offset, ls.rec[ki].z) : first + ls.index[ki];
      else if (k.count == 0u) refs[c] = kNoChild;
      else refs[c] = k.count <= kLeafMaxPrims ? layout_leaf_ref(prim_base + k.offset, k.count) : first + ls.index[ki];
    }
    d.left = refs[0]; d.right = refs[1]; d.pad0 = d.pad1 = 0u;
    out[first + ls.index[i]] = d;
  } else if (nd.count > kLeafMaxPrims) {  // the chain of DevBuilder::emit_leaf
    DevNode d;
    layout_box(d.q, nd, pad_rel, pad_abs);
    layout_box(d.q + 6, nd, pad_rel, pad_abs);
    d.pad0 = d.pad1 = 0u;
    uint32_t cur = first + ls.index[i], f = prim_base + nd.offset, count = nd.count;
    for (;;) {
      d.left = layout_leaf_ref(f, kLeafMaxPrims);
      f += kLeafMaxPrims;
      count -= kLeafMaxPrims;
      if (count <= kLeafMaxPrims) {
        d.right = layout_leaf_ref(f, count);
        out[cur] = d;
        break;
      }
      d.right = cur + 1u;
      out[cur] = d;
      ++cur;
    }
  }
}

// a finished tree that stays on the device until its layout has been written (BlasDeviceTree::handle)
struct KeptTree {
  void* pool = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  int device = 0;
  LayoutState ls{};
};

// BlasBuildFn: see host_scene.hpp
bool gpu_build_blas(const BlasBuildInput& in, std::vector<uint32_t>& order, std::vector<BuildNode>& arena, std::string& err, BlasDeviceTree* keep) {
  const uint32_t n = in.count;
  // One device allocation, carved up here, and a stream of its own: a RAYCA_BUILDER_SAH scene builds two trees at the same
  // time from two host threads (host_scene.cpp), and each build is a chain of small launches with a counter read-back per
  // level -- on the null stream the two chains would queue behind each other, and two dozen hipMalloc / hipFree pairs cost
  // more than some of the levels.
  void* pool = nullptr;
  hipStream_t stream = static_cast<hipStream_t>(in.stream);
  const bool own_stream = stream == nullptr;   // (a caller's stream is only drained here, never destroyed)
  auto cleanup = [&] {
    if (stream) {
      (void)hipStreamSynchronize(stream);
      if (own_stream) (void)hipStreamDestroy(stream);
      stream = nullptr;
    }
    if (pool) (void)hipFree(pool);
    pool = nullptr;
  };
  static const bool verbose = getenv("RAYCA_BUILD_TIMING") != nullptr;
  auto tp = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    if (stream) (void)hipStreamSynchronize(stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[rayca build]     gpu %-22s %7.2f ms\n", what, std::chrono::duration<float, std::milli>(now - tp).count());
    tp = now;
  };
  HB_TRY(hipSetDevice((int)in.device));
  if (own_stream) HB_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  const uint32_t max_slots = n / kBig + 2, max_chunks = n / kChunk + max_slots + 2;
  size_t pool_bytes = 0;
  auto reserve = [&](size_t bytes) {  // offsets first, pointers once the pool exists; 256-B aligned pieces
    const size_t at = pool_bytes;
    pool_bytes += (std::max<size_t>(bytes, 4) + 255) / 256 * 256;
    return at;
  };
  size_t off_f[9], off_u[9];
  for (int i = 0; i < 9; ++i) off_f[i] = reserve(sizeof(float) * n);
  for (int i = 0; i < 9; ++i) off_u[i] = reserve(sizeof(uint32_t) * ((size_t)n + 2));
  const size_t off_nodes = reserve(sizeof(DNode) * (2 * (size_t)n + 2)), off_big = reserve(sizeof(uint32_t) * (2 * (size_t)n + 2)), off_counts = reserve(64), off_levels = reserve(sizeof(uint32_t) * 4u * ((size_t)in.max_depth + 2u)), off_gbins = reserve((size_t)max_slots * kBinWords * 4);
  const size_t off_chunks[2] = {reserve((size_t)max_chunks * sizeof(uint2)), reserve((size_t)max_chunks * sizeof(uint2))};
  const size_t off_splits = reserve(2 * (size_t)max_slots * sizeof(BigSplit)), off_chcnt = reserve(2 * (size_t)max_chunks * sizeof(uint32_t)),
               off_chbase = reserve(2 * (size_t)max_chunks * sizeof(uint32_t));
  size_t off_layout[8] = {};
  if (keep)
    for (size_t& o : off_layout) o = reserve(sizeof(uint32_t) * (2 * (size_t)n + 2));
  HB_TRY(hipMalloc(&pool, pool_bytes));
  char* base = static_cast<char*>(pool);
  BuildState st{};
  float* f[9];
  for (int i = 0; i < 9; ++i) f[i] = reinterpret_cast<float*>(base + off_f[i]);
  const float* src[9] = {in.cent[0], in.cent[1], in.cent[2], in.bmin[0], in.bmin[1], in.bmin[2], in.bmax[0], in.bmax[1], in.bmax[2]};
  StagedCopier staged;   // (page-locked staging blocks: staging.hpp)
  for (int i = 0; i < 9; ++i) HB_TRY(staged.copy(f[i], src[i], sizeof(float) * n, stream));
  for (int c = 0; c < 3; ++c) {
    st.cent[c] = f[c];
    st.bmin[c] = f[3 + c];
    st.bmax[c] = f[6 + c];
  }
  uint32_t* u[9];
  for (int i = 0; i < 9; ++i) u[i] = reinterpret_cast<uint32_t*>(base + off_u[i]);
  st.order = u[0]; st.tmp = u[1]; st.rank = u[2]; st.hole_pos = u[3]; st.filler_pos = u[4];
  uint32_t* lists[2] = {u[5], u[6]};
  st.small_nodes = u[7];
  st.small_levels = u[8];
  st.nodes = reinterpret_cast<DNode*>(base + off_nodes);
  st.big = reinterpret_cast<uint32_t*>(base + off_big);
  st.node_count = reinterpret_cast<uint32_t*>(base + off_counts);
  st.chunk_count = st.node_count + 3;
  st.slot_count = st.node_count + 4;
  st.gbins = reinterpret_cast<uint32_t*>(base + off_gbins);
  for (int i = 0; i < 2; ++i) st.chunks[i] = reinterpret_cast<uint2*>(base + off_chunks[i]);
  for (int i = 0; i < 2; ++i) st.splits[i] = reinterpret_cast<BigSplit*>(base + off_splits) + (size_t)i * max_slots;
  st.ch_cnt = reinterpret_cast<uint32_t*>(base + off_chcnt);
  st.ch_base = reinterpret_cast<uint32_t*>(base + off_chbase);
  st.small_count = st.node_count + 2;
  uint32_t* const level_counts = reinterpret_cast<uint32_t*>(base + off_levels);   // [level][open nodes, chunks, big slots, -]
  st.seed_origin = in.seed_origin ? 1u : 0u;
  st.max_depth = in.max_depth;

  std::vector<uint32_t> ident(n);
  for (uint32_t i = 0; i < n; ++i) ident[i] = i;
  HB_TRY(staged.copy(st.order, ident.data(), sizeof(uint32_t) * n, stream));
  DNode root{};
  for (int c = 0; c < 3; ++c) {
    root.a[c] = in.root_min[c];
    root.b[c] = in.root_max[c];
  }
  root.a[3] = root.b[3] = 1.0f;
  root.offset = 0; root.count = n; root.left = root.right = -1;
  const uint32_t root_big = n > kBig ? 0u : RAYCA_NONE;
  HB_TRY(hipMemcpyAsync(st.nodes, &root, sizeof root, hipMemcpyHostToDevice, stream));
  HB_TRY(hipMemcpyAsync(st.big, &root_big, sizeof root_big, hipMemcpyHostToDevice, stream));
  const uint32_t init[5] = {1u, 0u, 0u, 0u, 0u};
  HB_TRY(hipMemcpyAsync(st.node_count, init, sizeof init, hipMemcpyHostToDevice, stream));
  const uint32_t zero = 0;
  HB_TRY(hipMemcpyAsync(lists[0], &zero, 4, hipMemcpyHostToDevice, stream));  // level 0: the root
  // the level records: all zero but the root's {one open node, its chunks, one big slot}
  HB_TRY(hipMemsetAsync(level_counts, 0, sizeof(uint32_t) * 4u * ((size_t)in.max_depth + 2u), stream));
  HB_TRY(hipMemsetAsync(st.splits[0], 0, 2 * (size_t)max_slots * sizeof(BigSplit), stream));   // (the root's chunk_first = 0)
  const uint32_t root_chunks = n > kBig ? (n + kChunk - 1) / kChunk : 0u;
  const uint32_t record0[4] = {n > 0 ? 1u : 0u, root_chunks, n > kBig ? 1u : 0u, 0u};
  HB_TRY(hipMemcpyAsync(level_counts, record0, sizeof record0, hipMemcpyHostToDevice, stream));
  std::vector<uint2> rc(root_chunks);
  for (uint32_t k = 0; k < root_chunks; ++k) rc[k] = make_uint2(0u, k);
  if (root_chunks) HB_TRY(hipMemcpyAsync(st.chunks[0], rc.data(), sizeof(uint2) * root_chunks, hipMemcpyHostToDevice, stream));
  HB_TRY(hipStreamSynchronize(stream));  // (the sources above are locals and caller memory: staged before they go away)
  staged.finish();

  lap("alloc + upload");
  // Levels: every level's record {open nodes, chunks, big slots} lives on the device (level_counts, zeroed once); the
  // kernels of level L read record L for their bounds and count into record L + 1.  The host queues kLevelBatch levels at
  // a time with grids that bound the work (an open node holds more than kSeq primitives, a level has at most 2^L nodes)
  // and reads the records back once per batch -- round 2 read three counters back after EVERY level: 19 stream
  // synchronisations for the atrium, a third of the build.
  constexpr uint32_t kLevelBatch = 8;
  uint32_t levels_run = 0, blocks_run = 0;
  uint32_t n_active = n > 0 ? 1u : 0u;
  bool big_possible = n > kBig;   // (a big node only has big ancestors: once a level has none, no later level has)
  std::vector<uint32_t> lc_host;
  for (uint32_t level0 = 0; level0 < in.max_depth && n_active > 0; level0 += kLevelBatch) {
    const uint32_t batch = std::min<uint32_t>(kLevelBatch, in.max_depth - level0);
    const auto lt0 = std::chrono::steady_clock::now();
    for (uint32_t level = level0; level < level0 + batch; ++level) {
      const uint32_t* lc = level_counts + 4u * level;
      uint32_t* lc_next = level_counts + 4u * (level + 1u);
      const uint64_t by_depth = level < 31u ? (1ull << level) : ~0ull;
      const uint32_t bound = (uint32_t)std::min<uint64_t>(by_depth, (uint64_t)n / (kSeq + 1u) + 1u);
      BuildState sl = st;
      sl.chunk_count = lc_next + 1;
      sl.slot_count = lc_next + 2;
      if (big_possible) {  // big nodes of this level: bins by many workgroups
        hipLaunchKernelGGL(k_bin_init, dim3((std::min<uint64_t>(by_depth, max_slots) * 3 * 64 + kB - 1) / kB), dim3(kB), 0, stream, sl, lc);
        hipLaunchKernelGGL(k_bin_big, dim3((uint32_t)std::min<uint64_t>(by_depth * ((n + kChunk - 1) / kChunk), max_chunks)), dim3(kB), 0, stream, sl, sl.chunks[level & 1], lc);
        HB_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(k_build_level, dim3(bound), dim3(big_possible ? kBuildLevelMaxBlock : kB), 0, stream, sl, lists[level & 1], lists[(level + 1) & 1], lc_next,
                         sl.chunks[(level + 1) & 1], level, lc);
      HB_TRY(hipGetLastError());
#if RAYCA_BIG_PARTITION
      if (big_possible) {  // the big nodes' partitions and children, by their chunks (k_build_level has left the decisions)
        const dim3 cgrid((uint32_t)std::min<uint64_t>(by_depth * ((n + kChunk - 1) / kChunk), max_chunks)), sgrid((uint32_t)((std::min<uint64_t>(by_depth, max_slots) + 63u) / 64u));
        const uint2* cl = sl.chunks[level & 1];
        hipLaunchKernelGGL(k_big_count, cgrid, dim3(kB), 0, stream, sl, cl, lc, level);
        hipLaunchKernelGGL(k_big_classify, cgrid, dim3(kB), 0, stream, sl, cl, lc, level);
        hipLaunchKernelGGL(k_big_scan, sgrid, dim3(64), 0, stream, sl, lc, level);
        hipLaunchKernelGGL(k_big_rank, cgrid, dim3(kB), 0, stream, sl, cl, lc, level);
        hipLaunchKernelGGL(k_big_scatter, cgrid, dim3(kB), 0, stream, sl, cl, lc, level);
        hipLaunchKernelGGL(k_big_finish, cgrid, dim3(kB), 0, stream, sl, cl, lc, level);
        hipLaunchKernelGGL(k_big_children, sgrid, dim3(64), 0, stream, sl, lists[(level + 1) & 1], lc_next, sl.chunks[(level + 1) & 1], level, lc);
        HB_TRY(hipGetLastError());
      }
#endif
    }
    lc_host.resize(4u * (batch + 1u));
    HB_TRY(hipMemcpyAsync(lc_host.data(), level_counts + 4u * level0, sizeof(uint32_t) * lc_host.size(), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    for (uint32_t k = 0; k < batch; ++k) {
      if (lc_host[4 * k] == 0) break;
      ++levels_run;
      blocks_run += lc_host[4 * k];
    }
    for (uint32_t k = 1; k <= batch; ++k)
      if (lc_host[4 * k + 2] > max_slots || lc_host[4 * k + 1] > max_chunks) {
        err = "gpu bvh build: big-node bookkeeping overflow";
        cleanup();
        return false;
      }
    n_active = lc_host[4 * batch];
    big_possible = big_possible && lc_host[4 * batch + 1] != 0;
    if (verbose && getenv("RAYCA_BUILD_LEVELS"))
      fprintf(stderr, "[rayca build]     levels %2u..%2u: %7.2f ms, %u open nodes left\n", level0, level0 + batch - 1,
              std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - lt0).count(), n_active);
  }
  lap("levels");
  uint32_t n_small = 0;
  HB_TRY(hipMemcpyAsync(&n_small, st.small_count, 4, hipMemcpyDeviceToHost, stream));
  HB_TRY(hipStreamSynchronize(stream));
  if (verbose) fprintf(stderr, "[rayca build]   gpu: %u primitives, %u levels, %u node blocks, %u small subtrees\n", n, levels_run, blocks_run, n_small);
  if (n_small) {
    hipLaunchKernelGGL(k_build_small, dim3(n_small), dim3(64), 0, stream, st, n_small);
    HB_TRY(hipGetLastError());
  }
  lap("small subtrees");
  uint32_t node_count = 0;
  HB_TRY(hipMemcpyAsync(&node_count, st.node_count, 4, hipMemcpyDeviceToHost, stream));
  HB_TRY(hipStreamSynchronize(stream));
  if (keep) {  // the tree stays here: work out the layout's size and stack need, hand the pool over
    LayoutState ls{};
    ls.nodes = st.nodes;
    ls.node_count = node_count;
    uint32_t** parts[4] = {&ls.parent, &ls.arrived, &ls.index, &ls.leafed};
    for (int i = 0; i < 4; ++i) *parts[i] = reinterpret_cast<uint32_t*>(base + off_layout[i]);
    ls.rec = reinterpret_cast<uint4*>(base + off_layout[4]);   // (four words per node: off_layout[4..7], contiguous and 256-B aligned)
    ls.node_cost = in.layout_node_cost;
    ls.leaf_max = std::min<uint32_t>(std::max<uint32_t>(in.layout_leaf_max, 1u), kLeafMaxPrims);
    const dim3 grid((node_count + kB - 1) / kB), block(kB);
    hipLaunchKernelGGL(k_layout_parents, grid, block, 0, stream, ls);
    hipLaunchKernelGGL(k_layout_sizes, grid, block, 0, stream, ls);
    hipLaunchKernelGGL(k_layout_index, grid, block, 0, stream, ls);
    HB_TRY(hipGetLastError());
    uint32_t root_rec[4] = {0, 0, 0, 0};
    HB_TRY(hipMemcpyAsync(root_rec, ls.rec, sizeof root_rec, hipMemcpyDeviceToHost, stream));
    order.resize(n);
    HB_TRY(staged.copy_back(order.data(), st.order, sizeof(uint32_t) * n, stream));   // (through the staging blocks, like the uploads)
    HB_TRY(hipStreamSynchronize(stream));
    staged.finish();
    lap("layout sizes + order");
    KeptTree* kt = new KeptTree();
    kt->pool = pool;
    kt->stream = stream;
    kt->own_stream = own_stream;
    kt->device = (int)in.device;
    kt->ls = ls;
    pool = nullptr;     // (cleanup() below must not release what the handle now owns)
    stream = nullptr;
    keep->handle = kt;
    keep->node_count = root_rec[0];
    keep->need = root_rec[1];
    return true;
  }
  arena.resize(node_count);   // DNode == BuildNode (asserted above): no conversion pass
  order.resize(n);
  HB_TRY(staged.copy_back(static_cast<void*>(arena.data()), st.nodes, sizeof(DNode) * node_count, stream));
  HB_TRY(staged.copy_back(order.data(), st.order, sizeof(uint32_t) * n, stream));
  HB_TRY(hipStreamSynchronize(stream));
  staged.finish();
  lap("download");
  cleanup();
  lap("free");
  return true;
}

void gpu_release_tree(void* tree) {
  KeptTree* kt = static_cast<KeptTree*>(tree);
  if (!kt) return;
  (void)hipSetDevice(kt->device);
  if (kt->stream) {
    (void)hipStreamSynchronize(kt->stream);
    if (kt->own_stream) (void)hipStreamDestroy(kt->stream);
  }
  if (kt->pool) (void)hipFree(kt->pool);
  delete kt;
}

// Writes the tree's binary nodes into device_nodes[first, first + node_count) -- child indices absolute, leaf references
// from prim_base on, boxes padded like DevBuilder::put_box pads them -- and releases the tree.
bool gpu_emit_tree(void* tree, DevNode* device_nodes, uint32_t first, uint32_t prim_base, float pad_rel, float pad_abs, std::string& err) {
  KeptTree* kt = static_cast<KeptTree*>(tree);
  if (!kt) { err = "gpu bvh layout: no tree"; return false; }
  hipError_t e = hipSetDevice(kt->device);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_layout_emit, dim3((kt->ls.node_count + kB - 1) / kB), dim3(kB), 0, kt->stream, kt->ls, device_nodes, first, prim_base, pad_rel, pad_abs);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(kt->stream);
  if (e != hipSuccess) err = std::string("gpu bvh layout: ") + hipGetErrorString(e);
  gpu_release_tree(kt);
  return e == hipSuccess;
}

}  // namespace rayca
