// pcie_probe.hip -- what do host<->device copies cost on this runtime?  (design input for the host-pointer
// entry points of libfmx: fmx_count_batch moves 40 MB in and 24 MB out per 2^20 x 32 call)
//   hipcc -O2 --offload-arch=gfx950 -o pcie_probe pcie_probe.hip && ./pcie_probe
// Prints ms and GB/s for H2D of 40 MB and D2H of 24 MB: pageable / hipHostMalloc default / non-coherent /
// hipHostRegister'ed memory, 1..8 chunks, alone and with both directions in flight on two streams, and the
// rate of a copy KERNEL reading / writing pinned host memory directly.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t IN = 40u << 20, OUT = 24u << 20;
  void *d_in, *d_out;
  CK(hipMalloc(&d_in, IN)); CK(hipMalloc(&d_out, OUT));
  CK(hipMemset(d_out, 1, OUT));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  struct Kind { const char *name; void *in, *out; } kinds[4];
  // pageable
  kinds[0] = {"pageable", malloc(IN), malloc(OUT)};
  memset(kinds[0].in, 1, IN); memset(kinds[0].out, 1, OUT);
  kinds[1].name = "hostmalloc-default";
  CK(hipHostMalloc(&kinds[1].in, IN, hipHostMallocDefault)); CK(hipHostMalloc(&kinds[1].out, OUT, hipHostMallocDefault));
  kinds[2].name = "hostmalloc-noncoherent";
  CK(hipHostMalloc(&kinds[2].in, IN, hipHostMallocNonCoherent)); CK(hipHostMalloc(&kinds[2].out, OUT, hipHostMallocNonCoherent));
  kinds[3] = {"hostregister", malloc(IN), malloc(OUT)};
  memset(kinds[3].in, 1, IN); memset(kinds[3].out, 1, OUT);
  double t0 = now();
  CK(hipHostRegister(kinds[3].in, IN, hipHostRegisterDefault)); CK(hipHostRegister(kinds[3].out, OUT, hipHostRegisterDefault));
  printf("hipHostRegister of %zu MB: %.2f ms\n", (IN + OUT) >> 20, now() - t0);
  for (int k = 1; k < 3; k++) { memset(kinds[k].in, 1, IN); memset(kinds[k].out, 1, OUT); }
  const int REP = 10;
  for (auto &K : kinds) {
    for (int chunks = 1; chunks <= 8; chunks *= 2) {
      double h2d = 1e9, d2h = 1e9, both = 1e9;
      for (int r = 0; r < REP; r++) {
        double t = now();
        for (int c = 0; c < chunks; c++) CK(hipMemcpyAsync((char *)d_in + IN / chunks * c, (char *)K.in + IN / chunks * c, IN / chunks, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        double a = now() - t; if (a < h2d) h2d = a;
        t = now();
        for (int c = 0; c < chunks; c++) CK(hipMemcpyAsync((char *)K.out + OUT / chunks * c, (char *)d_out + OUT / chunks * c, OUT / chunks, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s2));
        a = now() - t; if (a < d2h) d2h = a;
        t = now();
        for (int c = 0; c < chunks; c++) {
          CK(hipMemcpyAsync((char *)d_in + IN / chunks * c, (char *)K.in + IN / chunks * c, IN / chunks, hipMemcpyHostToDevice, s1));
          CK(hipMemcpyAsync((char *)K.out + OUT / chunks * c, (char *)d_out + OUT / chunks * c, OUT / chunks, hipMemcpyDeviceToHost, s2));
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        a = now() - t; if (a < both) both = a;
      }
      printf("%-24s chunks %d  H2D 40MB %.3f ms (%.1f GB/s)  D2H 24MB %.3f ms (%.1f GB/s)  both at once %.3f ms\n", K.name,
             chunks, h2d, IN / h2d / 1e6, d2h, OUT / d2h / 1e6, both);
    }
  }
  // copy kernels over pinned host memory (zero-copy)
  for (int k = 1; k < 4; k++) {
    void *hin_d = nullptr, *hout_d = nullptr;
    CK(hipHostGetDevicePointer(&hin_d, kinds[k].in, 0)); CK(hipHostGetDevicePointer(&hout_d, kinds[k].out, 0));
    for (int blocks = 64; blocks <= 1024; blocks *= 4) {
      double h2d = 1e9, d2h = 1e9, both = 1e9;
      for (int r = 0; r < REP; r++) {
        double t = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const uint4 *)hin_d, (uint4 *)d_in, IN / 16);
        CK(hipStreamSynchronize(s1));
        double a = now() - t; if (a < h2d) h2d = a;
        t = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)hout_d, OUT / 16);
        CK(hipStreamSynchronize(s2));
        a = now() - t; if (a < d2h) d2h = a;
        t = now();
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s1, (const uint4 *)hin_d, (uint4 *)d_in, IN / 16);
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)hout_d, OUT / 16);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        a = now() - t; if (a < both) both = a;
      }
      printf("copy KERNEL %-22s blocks %4d  H2D %.3f ms (%.1f GB/s)  D2H %.3f ms (%.1f GB/s)  both %.3f ms\n", kinds[k].name,
             blocks, h2d, IN / h2d / 1e6, d2h, OUT / d2h / 1e6, both);
    }
  }
  return 0;
}
