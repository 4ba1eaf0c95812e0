// Do rocPRIM's radix_sort_pairs (u64 keys, u64 values, double_buffer) and inclusive_scan work beyond 2^32 items?
//   hipcc -O2 --offload-arch=gfx950 -o rocprim_large_probe rocprim_large_probe.hip && ./rocprim_large_probe [log2n_extra]
#include <hip/hip_runtime.h>
#include <cstring>
#include <iterator>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __host__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void fill(uint64_t *k, uint64_t *v, uint64_t n, unsigned bits) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    k[i] = mix(i) & ((bits >= 64) ? ~0ull : ((1ull << bits) - 1ull));
    v[i] = i;
  }
}
__global__ void check(const uint64_t *k, const uint64_t *v, uint64_t n, unsigned bits, unsigned long long *bad,
                      unsigned long long *vsum) {
  unsigned long long loc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t want = mix(v[i]) & ((bits >= 64) ? ~0ull : ((1ull << bits) - 1ull));
    if (k[i] != want) atomicAdd(&bad[0], 1ull);                   // the value still belongs to its key
    if (i && k[i - 1] > k[i]) atomicAdd(&bad[1], 1ull);           // sorted
    if (i && k[i - 1] == k[i] && v[i - 1] > v[i]) atomicAdd(&bad[2], 1ull);   // stable
    loc += v[i];
  }
  atomicAdd(vsum, loc);
}
struct MaxOp { __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; } };
__global__ void fill_heads(uint64_t *h, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    h[i] = (i % 5 == 0) ? i : 0;
}
__global__ void check_heads(const uint64_t *h, uint64_t n, unsigned long long *bad) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (h[i] != i - i % 5) atomicAdd(bad, 1ull);
}
int main(int argc, char **argv) {
  const uint64_t n = (1ull << 32) + (1ull << 20);
  uint64_t *ka, *kb, *va, *vb;
  unsigned long long *bad, hb[4];
  CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8)); CK(hipMalloc(&va, n * 8)); CK(hipMalloc(&vb, n * 8));
  CK(hipMalloc(&bad, 32));
  for (unsigned bits : {64u, 34u}) {
    fill<<<4096, 256>>>(ka, va, n, bits);
    rocprim::double_buffer<uint64_t> k(ka, kb), v(va, vb);
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tb, k, v, (size_t)n, 0u, bits, (hipStream_t)0));
    void *tmp; CK(hipMalloc(&tmp, tb));
    CK(rocprim::radix_sort_pairs(tmp, tb, k, v, (size_t)n, 0u, bits, (hipStream_t)0));
    CK(hipMemset(bad, 0, 32));
    check<<<4096, 256>>>(k.current(), v.current(), n, bits, bad, bad + 3);
    CK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost));
    const unsigned long long want = (unsigned long long)((__uint128_t)n * (n - 1) / 2);
    printf("radix_sort_pairs u64/u64 n=%llu bits=%u tmp=%zu: key/value mismatches %llu, out of order %llu, unstable %llu, value sum %s\n",
           (unsigned long long)n, bits, tb, hb[0], hb[1], hb[2], hb[3] == want ? "ok" : "WRONG");
    CK(hipFree(tmp));
  }
  fill_heads<<<4096, 256>>>(ka, n);
  size_t tb = 0;
  CK(rocprim::inclusive_scan(nullptr, tb, ka, ka, (size_t)n, MaxOp(), (hipStream_t)0));
  void *tmp; CK(hipMalloc(&tmp, tb));
  CK(rocprim::inclusive_scan(tmp, tb, ka, ka, (size_t)n, MaxOp(), (hipStream_t)0));
  CK(hipMemset(bad, 0, 32));
  check_heads<<<4096, 256>>>(ka, n, bad);
  CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
  printf("inclusive_scan(max) u64 in place n=%llu: wrong entries %llu\n", (unsigned long long)n, hb[0]);
  return 0;
}
