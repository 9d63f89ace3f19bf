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

// pow() with the host libm's bits: the reference's `deltaT**m2` (source/ice_therm_vertical.F90:716, m2 = 1.36,
// frzmlt_bottom_lateral) is a call of glibc's pow -- sysdeps/ieee754/dbl-64/e_pow.c, the same author's algorithm:
// log(x) = k ln2 + log(c) + log1p(z/c - 1) with a 128-entry table of c (1/c exact on 9 bits, so z/c - 1 is exact with
// one fma) and a degree-7 polynomial, carried as a double-double (hi, lo); then exp(y log x) like exp() above with
// the low part added to the reduced argument.  Again the `__pow_fma` build (x86-64 with FMA) is restated: explicit
// fma() where its compiler fuses a*b+c.  Verified against the host libm on 6e8 argument pairs (tests/
// test_libm_exact.py runs 1e7).  The table is regenerated from its definition by scripts/gen_exp_table.py.
// Domain of the restatement: x positive and normal, 2^-54 <= |y log x| < 512, y of ordinary size; everything else
// (x = 0, negative, subnormal, inf, NaN, huge or tiny y) goes to the platform pow, which agrees there because those
// results are exact (0, 1, inf) or never occur on the hot path.
#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
static const double libm_pow_table[128][3] = {
#include "libm_pow_table.h"
};

CICE_HD inline double pow_libm(double x, double y) {
  constexpr int N = 128;
  constexpr unsigned long long OFF = 0x3FE6955500000000ull;
  constexpr double A0 = -0x1p-1, A1 = 0x1.555555555556p-2 * -2, A2 = -0x1.0000000000006p-2 * -2,
                   A3 = 0x1.999999959554ep-3 * 4, A4 = -0x1.555555529a47ap-3 * 4, A5 = 0x1.2495b9b4845e9p-3 * -8,
                   A6 = -0x1.0002b8b263fc3p-3 * -8;
  constexpr double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  constexpr double InvLn2N = 0x1.71547652b82fep0 * N, Shift = 0x1.8p52;
  constexpr double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
                   C5 = 0x1.1111167a4d017p-7;
  unsigned long long ix, iy;
  std::memcpy(&ix, &x, 8);
  std::memcpy(&iy, &y, 8);
  const unsigned topx = (unsigned)(ix >> 52), topy = (unsigned)(iy >> 52) & 0x7ff;
  if (topx - 0x001u >= 0x7ffu - 0x001u || topy - 0x3beu >= 0x43eu - 0x3beu) return std::pow(x, y);
  // log_inline
  const unsigned long long tmp = ix - OFF;
  const int i = (int)((tmp >> 45) % N);
  const int k = (int)((long long)tmp >> 52);
  const unsigned long long iz = ix - (tmp & (0xfffull << 52));
  double z;
  std::memcpy(&z, &iz, 8);
  const double kd = (double)k;
  const double invc = libm_pow_table[i][0], logc = libm_pow_table[i][1], logctail = libm_pow_table[i][2];
  const double r = std::fma(z, invc, -1.0);
  const double t1 = std::fma(kd, Ln2hi, logc);
  const double t2 = t1 + r;
  const double lo1 = std::fma(kd, Ln2lo, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A0 * r, ar2 = r * ar, ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = std::fma(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double q = std::fma(ar2, std::fma(ar2, std::fma(r, A6, A5), std::fma(r, A4, A3)), std::fma(r, A2, A1));
  const double lo = std::fma(ar3, q, lo1 + lo2 + lo3 + lo4);
  const double lx = hi + lo;
  const double ltail = hi - lx + lo;
  const double ehi = y * lx;
  const double elo = std::fma(y, ltail, std::fma(y, lx, -ehi));
  // exp_inline
  const double aehi = std::fabs(ehi);
  if (!(aehi >= 0x1p-54)) return 1.0 + ehi;
  if (aehi >= 512.0) return std::pow(x, y);
  double kd2 = std::fma(InvLn2N, ehi, Shift);
  unsigned long long ki;
  std::memcpy(&ki, &kd2, 8);
  kd2 -= Shift;
  double rr = std::fma(kd2, NegLn2loN, std::fma(kd2, NegLn2hiN, ehi));
  rr += elo;
  const unsigned idx = 2u * (unsigned)(ki % N);
  const unsigned long long tbits = libm_exp_table[idx];
  const unsigned long long sbits = libm_exp_table[idx + 1] + (ki << (52 - 7));
  double tail, scale;
  std::memcpy(&tail, &tbits, 8);
  std::memcpy(&scale, &sbits, 8);
  const double r2 = rr * rr;
  const double tmp2 = std::fma(r2 * r2, std::fma(rr, C5, C4), std::fma(r2, std::fma(rr, C3, C2), tail + rr));
  return std::fma(scale, tmp2, scale);
}

}  // namespace cice
