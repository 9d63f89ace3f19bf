// Host probe for cice4_amd/csrc/libm_exact.h: counts arguments on which exp_libm differs from the host
// libm's exp.  Built and run by tests/test_libm_exact.py (g++, no GPU).
#include <cstdio>
#include <cstdlib>
#include "../cice4_amd/csrc/libm_exact.h"

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000000;
  unsigned long long s = 88172645463325252ull;
  long bad = 0, used = 0;
  for (long i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    double x;
    switch (i & 3) {
      case 0: x = -700.0 + 1400.0 * u; break;        // whole finite range (|x| >= 512 takes the fallback)
      case 1: x = -25.0 * u; break;                  // ridging participation function, Hibler strength
      case 2: x = -5897.8 / (200.0 + 80.0 * u); break;  // saturation humidity over ice
      default: x = (u - 0.5) * 2e-3; break;
    }
    ++used;
    const double e = std::exp(x), m = cice::exp_libm(x);
    if (!(e == m)) {
      if (bad < 5) std::printf("mismatch x=%a libm=%a restated=%a\n", x, e, m);
      ++bad;
    }
  }
  // a few exact points
  const double pts[] = {0.0, -0.0, -20.0, 1.0, -1.0, 0x1p-60, -0x1p-54, 511.9999, -511.9999};
  for (double x : pts) if (!(std::exp(x) == cice::exp_libm(x))) { std::printf("mismatch at %a\n", x); ++bad; }
  // pow: the hot path's deltaT**1.36 and a general sweep
  for (long i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double v = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    double x, y = 1.36;
    switch (i & 3) {
      case 0: x = 3.0 * u + 1e-6; break;                 // sst - Tbot of a melting ocean
      case 1: x = std::exp(20.0 * (u - 0.5)); break;
      case 2: x = 0.5 + u; y = 0.1 + 3.0 * v; break;
      default: x = 1e-3 * u + 1e-9; break;
    }
    ++used;
    const double e = std::pow(x, y), m = cice::pow_libm(x, y);
    if (!(e == m)) {
      if (bad < 5) std::printf("pow mismatch x=%a y=%a libm=%a restated=%a\n", x, y, e, m);
      ++bad;
    }
  }
  const double px[] = {0.0, 1.0, 1.0 + 0x1p-52, 0x1p-1030, 2.0, 1e300};
  for (double x : px) if (!(std::pow(x, 1.36) == cice::pow_libm(x, 1.36))) { std::printf("pow mismatch at %a\n", x); ++bad; }
  std::printf("checked=%ld mismatches=%ld\n", used, bad);
  return 0;
}
