// libm_compare.hip -- the bounce sampler's transcendentals, device libm against the host's (test infrastructure).
// The reference draws a bounce direction as theta = acos(sqrt(e1)), phi = 2 pi e2, (cos phi sin theta, sin phi sin theta,
// cos theta)  (sampler/cosine.rs:65-88); e1, e2 are fastrand f32 values, k * 2^-24.  The oracle evaluates this with the host's
// glibc, the kernels with the device's libm.  For EVERY e1 and e2 the generator can return this program evaluates the
// functions on both sides and counts the arguments on which the bits differ, and by how many ulps.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tests/microbench/libm_compare.hip -o tests/microbench/_build/libm_compare
#include <hip/hip_runtime.h>

#include "../../rayca_amd/csrc/libm_exact.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr uint32_t N = 1u << 24;
constexpr float kPi = 3.14159265358979323846f;

// what: 0 sqrtf(e), 1 acosf(sqrtf(e)) [cosine sampler], 2 acosf(e) [hemisphere sampler], 3 sinf(acosf(sqrtf(e))), 4 cosf(acosf(sqrtf(e))),
//       5 sinf(2 pi e), 6 cosf(2 pi e), 7 sinf(acosf(e)), 8 cosf(acosf(e)), 9 acosf(-e)
// EXACT: the device side uses libm_exact.hpp (what the kernels call), the host side always the C library
template <bool EXACT>
__host__ __device__ inline float eval(int what, float e) {
  auto ac = [](float x) { return EXACT ? rayca::rc_acosf(x) : acosf(x); };
  auto sn = [](float x) { return EXACT ? rayca::rc_sinf(x) : sinf(x); };
  auto cs = [](float x) { return EXACT ? rayca::rc_cosf(x) : cosf(x); };
  switch (what) {
    case 0: return sqrtf(e);
    case 1: return ac(sqrtf(e));
    case 2: return ac(e);
    case 3: return sn(ac(sqrtf(e)));
    case 4: return cs(ac(sqrtf(e)));
    case 5: return sn(2.0f * kPi * e);
    case 6: return cs(2.0f * kPi * e);
    case 7: return sn(ac(e));
    case 8: return cs(ac(e));
    default: return ac(-e);
  }
}
template <bool EXACT>
__global__ void k_eval(int what, float* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = eval<EXACT>(what, (float)i * (1.0f / 16777216.0f));   // fastrand: (u32 >> 8) as f32 / 2^24
}
// the inner functions on the HOST's argument, so that a difference is this function's own and not inherited
__global__ void k_eval_arg(int fn, const float* arg, float* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = fn == 0 ? sinf(arg[i]) : cosf(arg[i]);
}

int main() {
  float *d = nullptr, *darg = nullptr;
  CHECK(hipMalloc(&d, N * sizeof(float)));
  CHECK(hipMalloc(&darg, N * sizeof(float)));
  std::vector<float> dev(N), host(N), arg(N);
  const char* names[10] = {"sqrtf(e1)", "acosf(sqrtf(e1))", "acosf(e1)", "sinf(acosf(sqrtf(e1)))", "cosf(acosf(sqrtf(e1)))", "sinf(2 pi e2)", "cosf(2 pi e2)",
                           "sinf(acosf(e1))", "cosf(acosf(e1))", "acosf(-e1)"};
  const char* side = "device libm";
  auto report = [&](const char* name) {
    uint64_t diff = 0, diff1 = 0, diff2 = 0;
    uint32_t worst = 0, worst_i = 0;
    for (uint32_t i = 0; i < N; ++i) {
      uint32_t a, b;
      memcpy(&a, &dev[i], 4);
      memcpy(&b, &host[i], 4);
      if (a == b || (dev[i] != dev[i] && host[i] != host[i])) continue;
      ++diff;
      const uint32_t u = a > b ? a - b : b - a;   // same sign in these ranges (or +-0): distance in ulps
      if (u == 1) ++diff1; else if (u == 2) ++diff2;
      if (u > worst) { worst = u; worst_i = i; }
    }
    printf("{\"device\": \"%s\", \"function\": \"%s\", \"arguments\": %u, \"bits_differ\": %llu, \"fraction\": %.6f, \"by_1_ulp\": %llu, \"by_2_ulp\": %llu, \"worst_ulps\": %u, \"worst_at_k\": %u}\n",
           side, name, N, (unsigned long long)diff, (double)diff / N, (unsigned long long)diff1, (unsigned long long)diff2, worst, worst_i);
    fflush(stdout);
  };
  for (int exact = 0; exact < 2; ++exact) {
    side = exact ? "libm_exact.hpp (what the kernels call)" : "device libm";
    for (int what = 0; what < 10; ++what) {
      if (exact) hipLaunchKernelGGL(k_eval<true>, dim3(N / 256), dim3(256), 0, nullptr, what, d);
      else hipLaunchKernelGGL(k_eval<false>, dim3(N / 256), dim3(256), 0, nullptr, what, d);
      CHECK(hipMemcpy(dev.data(), d, N * sizeof(float), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < N; ++i) host[i] = eval<false>(what, (float)i * (1.0f / 16777216.0f));
      report(names[what]);
    }
  }
  side = "device libm";
  // sinf / cosf of theta on the HOST's theta = acosf(sqrtf(e1)): the functions' own differences
  for (uint32_t i = 0; i < N; ++i) arg[i] = acosf(sqrtf((float)i * (1.0f / 16777216.0f)));
  CHECK(hipMemcpy(darg, arg.data(), N * sizeof(float), hipMemcpyHostToDevice));
  for (int fn = 0; fn < 2; ++fn) {
    hipLaunchKernelGGL(k_eval_arg, dim3(N / 256), dim3(256), 0, nullptr, fn, darg, d);
    CHECK(hipMemcpy(dev.data(), d, N * sizeof(float), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < N; ++i) host[i] = fn == 0 ? sinf(arg[i]) : cosf(arg[i]);
    report(fn == 0 ? "sinf(theta), theta = the host's acosf(sqrtf(e1))" : "cosf(theta), theta = the host's acosf(sqrtf(e1))");
  }
  return 0;
}
