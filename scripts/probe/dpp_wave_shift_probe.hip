#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
  double x = 100.0 + threadIdx.x;
  out[threadIdx.x] = shr1(x);          // lane l gets value of lane l-1 (lane 0 keeps own)
  out[64 + threadIdx.x] = shl1(x);     // lane l gets value of lane l+1 (lane 63 keeps own)
  out[128 + threadIdx.x] = __shfl_up(x, 1);
  out[192 + threadIdx.x] = __shfl_down(x, 1);
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  k<<<1, 64>>>(d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { if (h[i] != h[128 + i]) bad++; if (h[64 + i] != h[192 + i]) bad++; }
  printf("dpp vs shfl mismatches: %d  (shr lane0 %g lane1 %g lane63 %g; shl lane0 %g lane62 %g lane63 %g)\n", bad, h[0], h[1], h[63], h[64], h[126], h[127]);
  return 0;
}
