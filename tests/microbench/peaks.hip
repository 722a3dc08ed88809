// peaks.hip -- two ceilings the roofline of the traversal kernels rests on, measured on the chip:
//
//   valu   wave-instructions per second the chip sustains on independent VALU instructions, for 1..8 waves per SIMD and
//          a few opcodes (v_fma_f32, v_max_f32, v_max3_f32, v_pk_fma_f32, v_cndmask_b32, v_fma_mix_f32): is a wave64 VALU
//          instruction 4 cycles of its SIMD or 2?
//   l1     vector-L1 (TCP) hit rate: every lane re-reads a 16 KiB per-block working set with 16-B loads
//          (global_load_dwordx4), (a) coherent -- the 64 lanes of a wave read 1 KiB of consecutive bytes, (b) every lane
//          in a 64-B / 128-B line of its own, (c) the traversal's own shape: four 16-B loads of one 64-B record per lane,
//          a record of its own per lane.  Reported as load instructions/s, lane-loads/s, bytes/s, and -- run under
//          rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD -- what one counted access is worth.
//
// One JSON line per measurement on stdout.  Test infrastructure: nothing of the product links or runs this.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/peaks.hip -o tests/microbench/_build/peaks
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(expr)                                                                          \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s: %s (%s:%d)\n", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

constexpr int kBlock = 256;   // one wave per SIMD
constexpr int kIlp = 16;      // independent accumulators per lane

// ---- VALU issue ----------------------------------------------------------------------------------------------------
// OP: see kValuOps below
template <int OP>
__global__ __launch_bounds__(kBlock) void k_valu(float* out, uint32_t iters, float a, float b) {
  extern __shared__ uint32_t occupancy_pad[];   // (dynamic LDS only bounds the blocks per CU)
  float acc[kIlp];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc2[kIlp];
#pragma unroll
  for (int i = 0; i < kIlp; ++i) {
    acc[i] = (float)(threadIdx.x + i) * 1e-3f;
    acc2[i] = f2{acc[i], acc[i] + 1.0f};
  }
  const f2 a2{a, a}, b2{b, b};
  const uint32_t h = __float_as_uint(a) & 0xFFFFu;
  unsigned long long mask = __ballot(threadIdx.x & 1);   // a defined lane mask in an SGPR pair
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a), "v"(acc[0]) : "vcc");   // and a defined vcc
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < kIlp; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : );
        if (OP == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i]) : "v"(h), "v"(a));
        if (OP == 6) {
          if (i & 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
          else asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
        }
        if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(mask));
        if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(acc[i]), "v"(a) : "vcc");
        if (OP == 11) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(acc[i]), "v"(a));
        if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i]) : "v"(a));
        if (OP == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 15) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(acc[i]));
        if (OP == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        if (OP == 17) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(acc[i]));
        if (OP == 18) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
        if (OP == 22) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(acc[i]) : "v"(a), "v"(b));
        if (OP == 23) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");   // two VALU instructions
        if (OP == 24) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(mask));   // two VALU instructions
        if (OP == 25) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : "vcc");   // two VALU instructions
        if (OP == 19) {   // the slab test's own mix: 6 fma, 3 max, 3 min, min3, max3, 2 cmp -- per 16: 6 fma + 10 others
          if (i < 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else if (i < 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
          else if (i < 12) asm volatile("v_min_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
          else if (i < 13) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else if (i < 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(acc[i]), "v"(a) : "vcc");
        }
        if (OP == 20) {   // the same box from centre and half extent: 9 fma, min3, max3, 2 cmp (13 of 16, the rest fma)
          if (i < 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else if (i < 13) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else if (i < 14) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
          else asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(acc[i]), "v"(a) : "vcc");
        }
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kIlp; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
  if (s == 12345.678f || mask == 0x123456789ull) out[blockIdx.x * kBlock + threadIdx.x] = s;   // (never: keeps the chains alive)
}

// ---- L1 hits -------------------------------------------------------------------------------------------------------
// MODE 0: coherent -- lane l of a wave reads 16 B at (l * 16 + step * 1024) mod 16 KiB
// MODE 1: every lane in a 128-B line of its own: lane l reads 16 B of line (l * 2 + step) mod 128   (64 distinct lines per instruction)
// MODE 2: every lane in a 64-B record of its own (two lanes per 128-B line): record (l + step * 64) mod 256
// MODE 3: the traversal's shape: four 16-B loads of one 64-B record, a record of its own per lane (as MODE 2, all four quarters)
// MODE 4: as MODE 3 but one 128-B record per lane read with eight 16-B loads (the 4-wide node)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_l1(const uint4* __restrict__ base, uint4* out, uint32_t iters, uint32_t set_bytes) {
  extern __shared__ uint32_t occupancy_pad[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // every block its own working set, so that the set is this CU's and hits L1 after the first pass (16 KiB of the 32 KiB L1 per
  // resident block at one block per CU; with more blocks per CU the sets compete -- reported as it is)
  const char* set = reinterpret_cast<const char*>(base) + (size_t)blockIdx.x * set_bytes;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint32_t step = wave * 7u;
  const uint32_t mask = set_bytes - 1u;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t off;
      if (MODE == 0) off = (lane * 16u + step * 1024u) & mask;
      else if (MODE == 1) off = (((lane * 2u + step) * 128u) & mask) + ((step & 7u) * 16u);
      else off = ((lane + step * 64u) * 64u) & mask;
      if (MODE == 4) off = ((lane + step * 64u) * 128u) & mask;
      const uint4* p = reinterpret_cast<const uint4*>(set + off);
      constexpr int quarters = MODE == 3 ? 4 : (MODE == 4 ? 8 : 1);
#pragma unroll
      for (int q = 0; q < quarters; ++q) {
        const uint4 v = p[q];   // global_load_dwordx4 (checked in the disassembly); the address depends on `step`, nothing is hoisted
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
      step += 3u;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * kBlock + threadIdx.x] = acc;
}

struct Result {
  double ms;
};

template <typename F>
double time_launch(F launch, int reps = 3) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();   // warm: code object, caches
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, nullptr));
    launch();
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return best;
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "all";
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const double clock_ghz = prop.clockRate * 1e-6;   // kHz -> GHz (the maximum; what the chip holds under load is not read here)
  float* out = nullptr;
  CHECK(hipMalloc(&out, (size_t)cus * 8 * kBlock * sizeof(uint4)));
  const size_t lds_cu = 160u * 1024u;

  if (what == "all" || what == "valu") {
    constexpr int kOps = 26;
    const char* names[kOps] = {"v_fma_f32", "v_max_f32", "v_max3_f32", "v_pk_fma_f32", "v_cndmask_b32 (vcc)", "v_fma_mix_f32", "v_min_f32/v_max_f32",
                               "v_add_f32", "v_mul_f32", "v_cndmask_b32_e64 (sgpr pair)", "v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32_e64 (sgpr pair)", "v_mov_b32",
                               "v_and_b32", "v_add_u32", "v_lshlrev_b32", "v_med3_f32", "v_cvt_f32_ubyte1", "v_sub_f32",
                               "slab mix: 6 fma + 3 max + 3 min + min3 + max3 + 2 cmp", "centre/half mix: 12 fma + min3 + max3 + 2 cmp",
                               "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32_e32 (vcc), dst != src", "pair: v_cmp_e32 vcc + v_cndmask_e32 vcc", "pair: v_cmp_e64 sgpr + v_cndmask_e64 sgpr",
                               "pair: v_cmp_e32 vcc + v_cndmask_e64 vcc"};
    using K = void (*)(float*, uint32_t, float, float);
    K kernels[kOps] = {k_valu<0>, k_valu<1>, k_valu<2>, k_valu<3>, k_valu<4>, k_valu<5>, k_valu<6>, k_valu<7>, k_valu<8>, k_valu<9>, k_valu<10>,
                       k_valu<11>, k_valu<12>, k_valu<13>, k_valu<14>, k_valu<15>, k_valu<16>, k_valu<17>, k_valu<18>, k_valu<19>, k_valu<20>, k_valu<21>, k_valu<22>, k_valu<23>, k_valu<24>, k_valu<25>};
    const uint32_t iters = 2000;
    for (int op = 0; op < kOps; ++op) {
      for (int w : {1, 2, 4, 8}) {   // waves per SIMD = blocks per CU
        const size_t lds = lds_cu / w / 1024 * 1024;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[op]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu));
        const int grid = cus * w;
        const double ms = time_launch([&] { hipLaunchKernelGGL(kernels[op], dim3(grid), dim3(kBlock), lds, nullptr, out, iters, 1.0001f, 0.5f); });
        const double wave_instr = (double)grid * 4.0 * iters * 4.0 * kIlp * (op >= 23 && op <= 25 ? 2.0 : 1.0);
        const double rate = wave_instr / (ms * 1e-3);
        const double cyc = (double)cus * 4.0 * clock_ghz * 1e9 / rate;   // SIMD cycles per wave-instruction at the maximum clock
        printf("{\"bench\": \"valu\", \"op\": \"%s\", \"waves_per_simd\": %d, \"grid\": %d, \"ms\": %.4f, \"G_wave_instr_per_s\": %.2f, "
               "\"simd_cycles_per_instr_at_max_clock\": %.3f, \"max_clock_GHz\": %.3f, \"cus\": %d}\n",
               names[op], w, grid, ms, rate * 1e-9, cyc, clock_ghz, cus);
        fflush(stdout);
      }
    }
  }
  if (what == "all" || what == "l1") {
    const uint32_t set_bytes = 16384;
    uint4* buf = nullptr;
    const size_t buf_bytes = (size_t)cus * 8 * set_bytes + 4096;
    CHECK(hipMalloc(&buf, buf_bytes));
    CHECK(hipMemset(buf, 1, buf_bytes));
    const char* names[5] = {"coherent 16 B/lane (1 KiB per instruction)", "a 128-B line per lane", "a 64-B record per lane, one 16-B load",
                            "a 64-B record per lane, four 16-B loads (binary node)", "a 128-B record per lane, eight 16-B loads (4-wide node)"};
    using K = void (*)(const uint4*, uint4*, uint32_t, uint32_t);
    K kernels[5] = {k_l1<0>, k_l1<1>, k_l1<2>, k_l1<3>, k_l1<4>};
    const int loads_per_r[5] = {1, 1, 1, 4, 8};
    for (int mode = 0; mode < 5; ++mode) {
      for (int w : {1, 2, 4, 8}) {
        const size_t lds = lds_cu / w / 1024 * 1024;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernels[mode]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cu));
        const int grid = cus * w;
        const uint32_t iters = 2000 / loads_per_r[mode] + 1;
        const double ms = time_launch([&] { hipLaunchKernelGGL(kernels[mode], dim3(grid), dim3(kBlock), lds, nullptr, buf, reinterpret_cast<uint4*>(out), iters, set_bytes); });
        const double instr = (double)grid * 4.0 * iters * 8.0 * loads_per_r[mode];
        const double t = ms * 1e-3;
        const double per_cu_clk = instr / t / cus / (clock_ghz * 1e9);
        printf("{\"bench\": \"l1\", \"pattern\": \"%s\", \"mode\": %d, \"waves_per_simd\": %d, \"grid\": %d, \"ms\": %.4f, \"load_instr\": %.0f, "
               "\"G_load_instr_per_s\": %.3f, \"TB_per_s_requested\": %.3f, \"load_instr_per_clk_per_cu\": %.5f, \"clk_per_load_instr_per_cu\": %.2f, "
               "\"B_per_clk_per_cu_requested\": %.2f, \"max_clock_GHz\": %.3f}\n",
               names[mode], mode, w, grid, ms, instr, instr / t * 1e-9, instr * 1024.0 / t * 1e-12, per_cu_clk, 1.0 / per_cu_clk, per_cu_clk * 1024.0, clock_ghz);
        fflush(stdout);
      }
    }
    CHECK(hipFree(buf));
  }
  CHECK(hipFree(out));
  return 0;
}
