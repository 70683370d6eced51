// lbft_math.h -- deterministic binary64 exp()/log() shared by the HIP kernels and the oracle's
// "strict" math mode.
//
// Why: the reference samples network delays as `exp(mu + sigma * N)` truncated to i64
// (bft-lib/src/simulator.rs:110-118, rand_distr 0.4 LogNormal) and the ziggurat slow path uses
// exp()/ln() (rand_distr 0.4 `ziggurat`, `StandardNormal::zero_case`).  Rust calls the platform libm
// (glibc).  ROCm's device `exp`/`log` are not bit-identical to glibc, so both sides of the parity test
// use THIS implementation; every operation below is an explicit IEEE-754 binary64 add/mul/div/fma, so
// host (g++ -ffp-contract=off) and device (hipcc -ffp-contract=off) agree bit for bit.
//
//  * lbft_exp: table-driven algorithm published in ARM optimized-routines `exp.c` (the one glibc
//    >= 2.28 ships), N = 128, degree-5 polynomial, < 0.52 ULP.  tests/test_math.py measures its
//    agreement with the host libm (glibc) over millions of points and pins the degenerate
//    "fixed delay" cases (exp(ln 10) -> 10.000000000000002).
//  * lbft_log: fdlibm `e_log.c` algorithm (< 1 ULP).  Only reached in the ziggurat tail
//    (probability ~ 2.7e-4 per normal sample), where only comparisons consume the result.
#ifndef LBFT_MATH_H
#define LBFT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LBFT_HD __host__ __device__ __forceinline__
#else
#define LBFT_HD inline
#endif

LBFT_HD double lbft_asdouble(uint64_t u) {
  union { uint64_t u; double d; } c; c.u = u; return c.d;
}
LBFT_HD uint64_t lbft_asuint64(double d) {
  union { uint64_t u; double d; } c; c.d = d; return c.u;
}

// Main path of exp(): valid for 2^-54 <= |x| < 512.
LBFT_HD double lbft_exp_core(double x, const uint64_t* tab) {
  const double InvLn2N = 0x1.71547652b82fep0 * 128.0;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8;
  const double NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double Shift = 0x1.8p52;
  const double C2 = 0x1.ffffffffffdbdp-2;
  const double C3 = 0x1.555555555543cp-3;
  const double C4 = 0x1.55555cf172b91p-5;
  const double C5 = 0x1.1111167a4d017p-7;
  // Explicit fused multiply-adds in exactly the places the FMA build of glibc's exp contracts them
  // (x86-64 glibc dispatches to that build on every FMA-capable CPU); verified bit-identical to the
  // host libm on 2e7 random points in tests/test_math.py.
  double kd = __builtin_fma(InvLn2N, x, Shift);
  uint64_t ki = lbft_asuint64(kd);
  kd -= Shift;
  double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
  uint64_t idx = 2 * (ki % 128);
  uint64_t top = ki << (52 - 7);
  double tail = lbft_asdouble(tab[idx]);
  uint64_t sbits = tab[idx + 1] + top;
  double r2 = r * r;
  double p1 = __builtin_fma(r, C3, C2);
  double p2 = __builtin_fma(r, C5, C4);
  double tmp = __builtin_fma(r2 * r2, p2, __builtin_fma(r2, p1, tail + r));
  double scale = lbft_asdouble(sbits);
  return __builtin_fma(scale, tmp, scale);
}

// exp(x).  `tab` = LBFT_EXP_TAB (256 x u64).
LBFT_HD double lbft_exp(double x, const uint64_t* tab) {
  uint32_t abstop = (uint32_t)(lbft_asuint64(x) >> 52) & 0x7ff;
  if (abstop < 0x3c9) return 1.0 + x;  // |x| < 2^-54
  if (abstop >= 0x408) {                // |x| >= 512 (or NaN): never reached by a sane delay model
    if (x != x) return x;
    if (x > 709.782712893384) return lbft_asdouble(0x7ff0000000000000ULL);
    if (x < -745.1332191019412) return 0.0;
    double h = lbft_exp_core(0.5 * x, tab);  // deterministic on host and device; not correctly rounded
    return h * h;
  }
  return lbft_exp_core(x, tab);
}

// log(x) for finite x > 0 (fdlibm e_log.c).
LBFT_HD double lbft_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
               Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
               Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t ux = lbft_asuint64(x);
  int32_t hx = (int32_t)(ux >> 32);
  uint32_t lx = (uint32_t)ux;
  int32_t k = 0;
  if (hx < 0x00100000) {  // subnormal / zero / negative
    if (((hx & 0x7fffffff) | lx) == 0) return -lbft_asdouble(0x7ff0000000000000ULL);
    if (hx < 0) return lbft_asdouble(0x7ff8000000000000ULL);
    k -= 54;
    x *= 0x1p54;
    ux = lbft_asuint64(x);
    hx = (int32_t)(ux >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  ux = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (ux & 0xffffffffULL);
  x = lbft_asdouble(ux);  // normalize x or x/2
  k += (i >> 20);
  double f = x - 1.0;
  double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      return dk * ln2_hi + dk * ln2_lo;
    }
    double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  double s = f / (2.0 + f);
  double z = s * s;
  i = hx - 0x6147a;
  double w = z * z;
  int32_t j = 0x6b851 - hx;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  double R = t2 + t1;
  if (i > 0) {
    double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  } else {
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
  }
}

#endif  // LBFT_MATH_H
