// what hipMalloc / first touch / hipFree of large buffers cost on this runtime (the wide builder holds ~170 GB of scratch
// at n = 2^32: its wall time is dominated by these calls, not by kernels).  hipcc -O2 --offload-arch=gfx950 alloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  (void)hipFree(nullptr);
  const double gb[] = {1, 4, 8, 16, 24, 32, 34.4, 48, 64};
  for (int rep = 0; rep < 2; rep++)
    for (double g : gb) {
      const size_t bytes = (size_t)(g * (1ull << 30));
      void *p = nullptr;
      double t0 = now();
      if (hipMalloc(&p, bytes) != hipSuccess) { printf("%.1f GiB: hipMalloc failed\n", g); continue; }
      double t1 = now();
      (void)hipMemset(p, 1, bytes);
      (void)hipDeviceSynchronize();
      double t2 = now();
      (void)hipFree(p);
      double t3 = now();
      printf("rep %d  %5.1f GiB: hipMalloc %8.2f ms   memset+sync %8.2f ms   hipFree %8.2f ms\n", rep, g, t1 - t0, t2 - t1, t3 - t2);
    }
  // five 34.4 GiB buffers at once (the wide builder's scratch at n = 2^32 + 2^20), freed together
  std::vector<void *> v;
  double t0 = now();
  for (int i = 0; i < 5; i++) { void *p = nullptr; if (hipMalloc(&p, (size_t)(34.4 * (1ull << 30))) == hipSuccess) v.push_back(p); }
  double t1 = now();
  for (void *p : v) (void)hipMemset(p, 1, (size_t)(34.4 * (1ull << 30)));
  (void)hipDeviceSynchronize();
  double t2 = now();
  for (void *p : v) (void)hipFree(p);
  double t3 = now();
  printf("5 x 34.4 GiB: hipMalloc %8.2f ms   memset+sync %8.2f ms   hipFree %8.2f ms\n", t1 - t0, t2 - t1, t3 - t2);
  t0 = now();
  void *q = nullptr;
  (void)hipMalloc(&q, (size_t)34.4 * (1ull << 30));
  t1 = now();
  printf("next 34 GiB hipMalloc after those frees: %8.2f ms\n", t1 - t0);
  (void)hipFree(q);
  return 0;
}
