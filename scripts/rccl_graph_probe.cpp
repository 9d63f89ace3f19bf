// Probe: can RCCL point-to-point calls be captured into a hipGraph on this ROCm?
// One rank, send/recv to self inside a group, between two kernels; replay 50 times.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s -> %d line %d\n", #x, (int)e_, __LINE__); return 1; } } while (0)
__global__ void k_fill(double* a, int n, double v) { int t = blockIdx.x * blockDim.x + threadIdx.x; if (t < n) a[t] = a[t] + v; }
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  int n = 4096;
  double *a, *b;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  printf("start\n"); ncclUniqueId id; CK(ncclGetUniqueId(&id)); printf("got id\n");
  ncclComm_t comm; CK(ncclCommInitRank(&comm, 1, id, 0)); printf("comm ok\n");
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // eager once
  hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, s, a, n, 1.0);
  CK(ncclGroupStart()); CK(ncclRecv(b, n, ncclDouble, 0, comm, s)); CK(ncclSend(a, n, ncclDouble, 0, comm, s)); CK(ncclGroupEnd());
  CK(hipStreamSynchronize(s));
  printf("eager self send/recv ok\n");
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int it = 0; it < 4; ++it) {
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, s, a, n, 1.0);
    CK(ncclGroupStart()); CK(ncclRecv(b, n, ncclDouble, 0, comm, s)); CK(ncclSend(a, n, ncclDouble, 0, comm, s)); CK(ncclGroupEnd());
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<double> h(n); CK(hipMemcpy(h.data(), b, n * 8, hipMemcpyDeviceToHost));
  printf("graph replay ok: b[0]=%g (expect %g), %.2f us per (kernel+p2p)\n", h[0], 1.0 + 200.0, ms * 1e3 / 200.0);
  // eager timing for comparison
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 200; ++r) {
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, s, a, n, 1.0);
    CK(ncclGroupStart()); CK(ncclRecv(b, n, ncclDouble, 0, comm, s)); CK(ncclSend(a, n, ncclDouble, 0, comm, s)); CK(ncclGroupEnd());
  }
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("eager: %.2f us per (kernel+p2p)\n", ms * 1e3 / 200.0);
  ncclCommDestroy(comm);
  return 0;
}
