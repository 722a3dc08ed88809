// libm_exact.hpp -- acosf / sinf / cosf with the HOST C library's results, bit for bit, for the kernels' bounce sampler.
//
// The reference draws a bounce direction as theta = acos(sqrt(e1)), phi = 2 pi e2, (cos phi sin theta, sin phi sin theta,
// cos theta) (sampler/cosine.rs:65-88, hemisphere.rs:17-40) through the platform's libm.  A direction that differs in its
// last bit sends a path that grazes a silhouette to the other side of it: the device's own acosf / sinf / cosf differ from
// glibc's on 31 % / 12 % / 15 % of the generator's arguments (tests/microbench/libm_compare.hip, every e = k 2^-24), which
// was the whole of round 2's parity outliers.  These are the algorithms glibc 2.35 uses, restated operation by operation
// (built with -ffp-contract=off like everything else):
//   acosf   fdlibm's single-precision __ieee754_acosf (rational approximation, three ranges)
//   sinf / cosf   the double-precision kernels of ARM's optimized routines that glibc adopted in 2.28: reduction by
//                 pi/2 with a 2^24-scaled reciprocal, degree-7 / degree-8 polynomials in double, one rounding to float.
//                 (glibc's x86-64 build also carries a variant compiled with FMA contraction; the two can differ only
//                 where the double result lies within ~2^-53 relative of a float rounding boundary -- none of the arguments
//                 the sampler produces: checked exhaustively on both the container's and the GPU box's CPU.)
// Arguments beyond the samplers' ranges (|x| >= 120 for sinf / cosf) are not needed and fall back to the platform's function.
// Checked against the host's libm on EVERY argument the samplers can produce: tests/test_libm_exact.py (host side, gcc),
// tests/microbench/libm_compare.hip (device side).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define RC_LIBM_FN __host__ __device__ inline
#else
#define RC_LIBM_FN inline
#endif

namespace rayca {

RC_LIBM_FN uint32_t rc_bits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
RC_LIBM_FN float rc_float(uint32_t u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}

RC_LIBM_FN float rc_acosf(float x) {
  const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
  const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
              pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int32_t hx = (int32_t)rc_bits(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) {  // |x| == 1
    if (hx > 0) return 0.0f;
    return pi + 2.0f * pio2_lo;
  }
  if (ix > 0x3f800000) return (x - x) / (x - x);  // |x| > 1: NaN
  if (ix < 0x3f000000) {                           // |x| < 0.5
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;  // |x| < 2^-57
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {  // x < -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  // x > 0.5
  const float z = (one - x) * 0.5f;
  const float s = sqrtf(z);
  const float df = rc_float(rc_bits(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}

// The table of the published routine -- sign[4] = {1, -1, -1, 1}, hpi_inv = 2/pi * 2^24, hpi = pi/2, c0..c4, s1..s3, and a
// second row with the cosine polynomial's coefficients negated -- spelled as scalars (an array indexed at run time would
// live in scratch memory on the GPU).  A negated row negates every term of the cosine polynomial, i.e. its result, exactly.
RC_LIBM_FN uint32_t rc_abstop12(float x) { return (rc_bits(x) >> 20) & 0x7ffu; }
// sine (n even) or cosine (n odd) polynomial of the reduced argument; negated: the second row of the table
RC_LIBM_FN float rc_sinf_poly(double x, double x2, bool negated, int n) {
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = s2c + x2 * s3c;
    const double x7 = x3 * x2;
    const double s = x + x3 * s1c;
    return (float)(s + x7 * s1);
  }
  const double c0 = 0x1p0, c1c = -0x1.ffffffd0c621cp-2, c2c = 0x1.55553e1068f19p-5, c3c = -0x1.6c087e89a359dp-10, c4c = 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double c2 = c3c + x2 * c4c;
  const double c1 = c0 + x2 * c1c;
  const double x6 = x4 * x2;
  const double c = c1 + x4 * c2c;
  const double r = c + x6 * c2;
  return (float)(negated ? -r : r);
}
RC_LIBM_FN double rc_reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * hpi;
}
RC_LIBM_FN double rc_quadrant_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }   // sign[n & 3] of {1, -1, -1, 1}

RC_LIBM_FN float rc_sinf(float y) {
  double x = y;
  const uint32_t top = rc_abstop12(y);
  if (top < rc_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
    const double s = x * x;
    if (top < rc_abstop12(0x1p-12f)) return y;  // (underflow signalling aside: same value)
    return rc_sinf_poly(x, s, false, 0);
  }
  if (top < rc_abstop12(120.0f)) {
    int n;
    x = rc_reduce_fast(x, &n);
    const double s = rc_quadrant_sign(n);
    return rc_sinf_poly(x * s, x * x, (n & 2) != 0, n);
  }
  return sinf(y);
}

RC_LIBM_FN float rc_cosf(float y) {
  double x = y;
  const uint32_t top = rc_abstop12(y);
  if (top < rc_abstop12(0x1.921FB6p-1f)) {  // |y| < pi/4
    const double x2 = x * x;
    if (top < rc_abstop12(0x1p-12f)) return 1.0f;
    return rc_sinf_poly(x, x2, false, 1);
  }
  if (top < rc_abstop12(120.0f)) {
    int n;
    x = rc_reduce_fast(x, &n);
    const double s = rc_quadrant_sign(n + 1);   // (the sign of the cosine's quadrant)
    return rc_sinf_poly(x * s, x * x, ((n + 1) & 2) != 0, n ^ 1);
  }
  return cosf(y);
}

}  // namespace rayca
