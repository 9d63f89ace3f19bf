// Can two IEEE divisions that share a denominator be made cheaper WITHOUT changing a bit?  (the column
// solver's Thomas sweep: Tm[k-1] = x / wbeta and wg[k] = sp[k-1] / wbeta.)
//   k_hw:     q1 = a1 / b; q2 = a2 / b as the compiler emits them (v_div_scale x2, v_rcp, 4-5 fma, v_div_fmas, v_div_fixup each)
//   k_shared: ONE refined reciprocal r of b (v_rcp + 4 fma = the hardware sequence's own refinement when nothing is scaled),
//             per numerator q0 = a r, e = fma(-b, q0, a), q = fma(e, r, q0), v_div_fixup(q, b, a) (zero / inf / nan operands)
//             -- taken only where no operand would make v_div_scale scale (|exponent| <= 370 for all three), the plain
//             divisions otherwise.
// Build:  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -save-temps -c div_shared_probe.hip   (instruction counts: the .s)
// Run:    hipcc -O3 -ffp-contract=off --offload-arch=gfx950 div_shared_probe.hip -o div_shared_probe && ./div_shared_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

__global__ void k_hw(const double* a1, const double* a2, const double* b, double* q1, double* q2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  q1[i] = a1[i] / b[i];
  q2[i] = a2[i] / b[i];
}

__device__ __forceinline__ bool mid(double x) {   // zero, or an exponent far from both ends (inf / nan: exponent reads 0, fixup handles them)
  return (unsigned)(__builtin_amdgcn_frexp_exp(x) + 370) <= 740u;
}

__global__ void k_shared(const double* a1, const double* a2, const double* b, double* q1, double* q2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x1 = a1[i], x2 = a2[i], d = b[i];
  if (mid(d) && mid(x1) && mid(x2) && d != 0.0) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    double p = x1 * r;
    double t = __builtin_fma(-d, p, x1);
    q1[i] = __builtin_amdgcn_div_fixup(__builtin_fma(t, r, p), d, x1);
    p = x2 * r;
    t = __builtin_fma(-d, p, x2);
    q2[i] = __builtin_amdgcn_div_fixup(__builtin_fma(t, r, p), d, x2);
  } else {
    q1[i] = x1 / d;
    q2[i] = x2 / d;
  }
}

int main() {
  const int n = 1 << 24;
  std::vector<double> a1(n), a2(n), b(n);
  std::mt19937_64 g(7);
  auto any = [&](int mode) {
    uint64_t u = g();
    if (mode == 0) {   // any bit pattern
    } else if (mode == 1) {   // moderate exponents, random mantissa and sign
      u = (u & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + (g() % 80)) << 52);
    } else {   // specials
      const uint64_t sp[] = {0, 0x8000000000000000ull, 0x7ff0000000000000ull, 0xfff0000000000000ull, 0x7ff8000000000000ull,
                             1, 0x000fffffffffffffull, 0x0010000000000000ull, 0x7fefffffffffffffull, 0x3ff0000000000000ull};
      u = sp[g() % 10];
    }
    double x;
    std::memcpy(&x, &u, 8);
    return x;
  };
  for (int i = 0; i < n; ++i) {
    const int m = i % 8 == 0 ? 0 : (i % 8 == 1 ? 2 : 1);
    a1[i] = any(m); a2[i] = any(i % 16 == 3 ? 2 : m); b[i] = any(i % 32 == 5 ? 2 : m);
  }
  double *da1, *da2, *db, *dq[4];
  hipMalloc(&da1, n * 8); hipMalloc(&da2, n * 8); hipMalloc(&db, n * 8);
  for (auto& p : dq) hipMalloc(&p, n * 8);
  hipMemcpy(da1, a1.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(da2, a2.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_hw, dim3(n / 256), dim3(256), 0, 0, da1, da2, db, dq[0], dq[1], n);
  hipLaunchKernelGGL(k_shared, dim3(n / 256), dim3(256), 0, 0, da1, da2, db, dq[2], dq[3], n);
  std::vector<uint64_t> h[4];
  for (int k = 0; k < 4; ++k) { h[k].resize(n); hipMemcpy(h[k].data(), dq[k], n * 8, hipMemcpyDeviceToHost); }
  long bad = 0, fast = 0;
  for (int i = 0; i < n; ++i) {
    const bool nan1 = (h[0][i] & 0x7fffffffffffffffull) > 0x7ff0000000000000ull, nan2 = (h[1][i] & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
    const bool n1 = (h[2][i] & 0x7fffffffffffffffull) > 0x7ff0000000000000ull, n2 = (h[3][i] & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
    if ((nan1 ? !n1 : h[0][i] != h[2][i]) || (nan2 ? !n2 : h[1][i] != h[3][i])) {
      if (bad < 5) std::printf("mismatch %d: a1 %a a2 %a b %a\n", i, a1[i], a2[i], b[i]);
      ++bad;
    }
  }
  (void)fast;
  std::printf("shared-reciprocal divisions against plain divisions on %d operand triples: %ld mismatches (signed zeros and nan-ness compared)\n", n, bad);
  return bad != 0;
}
