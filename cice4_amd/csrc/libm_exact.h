// exp() with the bits of the host libm the reference runs on.
//
// The reference calls the Fortran intrinsic `exp` (source/ice_mechred.F90:2001 `exp(-Gsum*astari)`,
// :1919 Hibler strength; source/ice_therm_vertical.F90:2393 `qsat = qqqice*exp(-TTTice*tmpvar)`), which
// amdflang lowers to glibc's `exp` -- a third-party dependency that is not under /root/reference:
// glibc 2.35 (Ubuntu 2.35-0ubuntu3.11 in this image), sysdeps/ieee754/dbl-64/e_exp.c, the table-driven
// algorithm of S. Nagy (ARM optimized routines, since glibc 2.28): exp(x) = 2^(k/N) exp(r), N = 128,
// k = round(x N/ln2), r = x - k ln2/N in two pieces, 2^(k/N) = scale (1 + tail) from a 128-entry table,
// exp(r) - 1 by a degree-5 polynomial.  Its error is < 0.511 ulp, i.e. NOT correctly rounded, so "any
// good exp" differs from it in ~1 argument of 2000 by one ulp -- and the EVP subcycling amplifies a 1-ulp
// change of `strength` to 1e-10..1e-3 relative within a few steps.  Restating the
// algorithm operation by operation removes that: on x86-64 CPUs with FMA glibc dispatches to the
// `__exp_fma` build of that file, in which the compiler contracts every a*b+c of the source into one
// fused operation; the explicit fma() calls below are that evaluation order.  Verified bit for bit against
// the host libm on 2e8 arguments (tests/test_libm_exact.py runs 1e7 of them on every CPU test run).
//
// The 2^(k/128) table is regenerated from its definition by scripts/gen_exp_table.py.
// Domain of the restatement: 2^-54 <= |x| < 512 (every call site of the hot path: x in [-25, 0]);
// tiny arguments return 1 + x as glibc does, larger ones fall back to the platform exp.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CICE_HD __host__ __device__
#else
#define CICE_HD
#endif

namespace cice {

#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
static const unsigned long long libm_exp_table[256] = {
#include "libm_exp_table.h"
};

CICE_HD inline double exp_libm(double x) {
  constexpr int N = 128;
  constexpr double InvLn2N = 0x1.71547652b82fep0 * N, Shift = 0x1.8p52;
  constexpr double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                   C5 = 0x1.1111167a4d017p-7;
  const double ax = std::fabs(x);
  if (!(ax >= 0x1p-54)) return 1.0 + x;     // also NaN
  if (ax >= 512.0) return std::exp(x);      // outside the restated range (never on the hot path)
  double kd = std::fma(InvLn2N, x, Shift);
  unsigned long long ki;
  std::memcpy(&ki, &kd, 8);
  kd -= Shift;
  const double r = std::fma(kd, NegLn2loN, std::fma(kd, NegLn2hiN, x));
  const unsigned idx = 2u * (unsigned)(ki % N);
  const unsigned long long tbits = libm_exp_table[idx];
  const unsigned long long sbits = libm_exp_table[idx + 1] + (ki << (52 - 7));
  double tail, scale;
  std::memcpy(&tail, &tbits, 8);
  std::memcpy(&scale, &sbits, 8);
  const double r2 = r * r;
  const double tmp = std::fma(r2 * r2, std::fma(r, C5, C4), std::fma(r2, std::fma(r, C3, C2), tail + r));
  return std::fma(scale, tmp, scale);
}

}  // namespace cice
