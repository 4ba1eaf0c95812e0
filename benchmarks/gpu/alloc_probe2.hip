// hipMalloc on a FRESH box (memory nobody has used since boot) against memory this process has freed, and two ways
// around hipMalloc: the virtual-memory API (hipMemCreate / hipMemMap) and a stream-ordered pool that never gives
// memory back (hipMallocAsync).  One process; the parts use disjoint memory so that each sees fresh pages.
//   hipcc -O2 --offload-arch=gfx950 -o alloc_probe2 alloc_probe2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
int main() {
  (void)hipFree(nullptr);
  const size_t G = 1ull << 30, big = (size_t)(34.4 * G);
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GiB of %.1f\n", fr / (double)G, tot / (double)G);
  // A: 3 x 34.4 GiB of fresh memory through hipMalloc
  std::vector<void *> a;
  for (int i = 0; i < 3; i++) {
    void *p = nullptr; double t0 = now();
    CK(hipMalloc(&p, big));
    double t1 = now();
    CK(hipMemset(p, 1, big)); CK(hipDeviceSynchronize());
    double t2 = now();
    printf("A fresh hipMalloc 34.4 GiB #%d: %8.1f ms   memset+sync %8.1f ms\n", i, t1 - t0, t2 - t1);
    a.push_back(p);
  }
  // C: 64 GiB of fresh memory through the virtual-memory API, 2 GiB physical handles
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = 2 * G, total = 64 * G;
    void *va = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t o = 0; o < total; o += chunk) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, chunk, &prop, 0));
      CK(hipMemMap((char *)va + o, chunk, 0, h, 0));
      hs.push_back(h);
    }
    hipMemAccessDesc ad = {};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &ad, 1));
    double t1 = now();
    CK(hipMemset(va, 1, total)); CK(hipDeviceSynchronize());
    double t2 = now();
    printf("C fresh VMM 64 GiB (granularity %zu, 2 GiB handles): reserve+create+map+access %8.1f ms   memset+sync %8.1f ms\n", gran, t1 - t0, t2 - t1);
    t0 = now();
    CK(hipMemUnmap(va, total));
    for (auto h : hs) CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, total));
    printf("C unmap+release %8.1f ms\n", now() - t0);
  }
  // D: 2 x 34.4 GiB through a stream-ordered pool that keeps what it is given back
  {
    hipMemPool_t pool;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (int rep = 0; rep < 2; rep++) {
      void *p[2];
      double t0 = now();
      for (int i = 0; i < 2; i++) CK(hipMallocAsync(&p[i], big, 0));
      CK(hipStreamSynchronize(0));
      double t1 = now();
      for (int i = 0; i < 2; i++) CK(hipMemsetAsync(p[i], 1, big, 0));
      CK(hipStreamSynchronize(0));
      double t2 = now();
      for (int i = 0; i < 2; i++) CK(hipFreeAsync(p[i], 0));
      CK(hipStreamSynchronize(0));
      printf("D pool rep %d: 2 x hipMallocAsync 34.4 GiB %8.1f ms   memset %8.1f ms   free %8.1f ms\n", rep, t1 - t0, t2 - t1, now() - t2);
    }
    CK(hipMemPoolTrimTo(pool, 0));
  }
  // B: what this process has freed, through hipMalloc again
  double t0 = now();
  for (void *p : a) CK(hipFree(p));
  printf("B hipFree 3 x 34.4 GiB %8.1f ms\n", now() - t0);
  for (int i = 0; i < 3; i++) {
    void *p = nullptr; t0 = now();
    CK(hipMalloc(&p, big));
    double t1 = now();
    printf("B recycled hipMalloc 34.4 GiB #%d: %8.1f ms\n", i, t1 - t0);
    a[i] = p;
  }
  for (void *p : a) CK(hipFree(p));
  // E: smaller pieces: is the cost per call or per byte?
  for (size_t sz : {1 * G, 4 * G, 16 * G}) {
    void *p = nullptr; t0 = now();
    CK(hipMalloc(&p, sz));
    printf("E recycled hipMalloc %2zu GiB: %8.1f ms\n", sz / G, now() - t0);
    CK(hipFree(p));
  }
  return 0;
}
