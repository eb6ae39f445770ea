// sfmx_math.h — scalar helpers shared by host and device code.  Everything here must be compiled
// with -ffp-contract=off: the parity contract is "the reference's FP64 operation order, no FMA".
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SFMX_HD __host__ __device__ inline
#else
#define SFMX_HD inline
#endif

namespace sfmx {

// (int)std::floor(v) as x86-64 evaluates it (cvttsd2si): NaN and out-of-range values give INT_MIN
// ("integer indefinite"), which the reference's bounds test `x0 < 0` then rejects (T:184-188).
SFMX_HD int floor_to_int_x86(double v) {
  const double f = floor(v);
  return (f >= -2147483648.0 && f < 2147483648.0) ? (int)f : (int)0x80000000;
}

// glibc 2.35 hypot (sysdeps/ieee754/dbl-64/e_hypot.c, the non-FMA kernel selected on generic
// x86-64), restated so that the device agrees bit-for-bit with the std::hypot calls at T:361,416,937.
// tests/test_host_math.py checks this restatement against the platform libm on >1e6 random inputs.
SFMX_HD double hypot_kernel(double ax, double ay) {
  double h = sqrt(ax * ax + ay * ay);
  double t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}
SFMX_HD double hypot_glibc(double x, double y) {
  if (!isfinite(x) || !isfinite(y)) {
    if (isinf(x) || isinf(y)) return INFINITY;
    return x + y;
  }
  x = fabs(x);
  y = fabs(y);
  double ax = x < y ? y : x;
  double ay = x < y ? x : y;
  const double SCALE = 0x1p-600, LARGE_VAL = 0x1p+511, TINY_VAL = 0x1p-459, EPS = 0x1p-54;
  if (ax > LARGE_VAL) {
    if (ay <= ax * EPS) return ax + ay;
    return hypot_kernel(ax * SCALE, ay * SCALE) / SCALE;
  }
  if (ay < TINY_VAL) {
    if (ax >= ay / EPS) return ax + ay;
    return hypot_kernel(ax / SCALE, ay / SCALE) * SCALE;
  }
  if (ax >= ay / EPS) return ax + ay;
  return hypot_kernel(ax, ay);
}

}  // namespace sfmx
