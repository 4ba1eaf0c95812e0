// Microbenchmark: dependent random line gathers, the access pattern of the count kernel.
// Each LANES-lane group runs U independent chains; every step loads one LANES*16-byte aligned
// chunk at a data-dependent address (next index = hash(loaded word ^ idx)).
// Build: hipcc -O3 --offload-arch=gfx950 gather.hip -o gather ; run: ./gather [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
template <int LANES, int U>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ buf, uint32_t nchunks_mask,
                                               int steps, uint32_t *out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = tid & (LANES - 1);
  const uint32_t gid = tid / LANES;
  uint32_t idx[U];
#pragma unroll
  for (int u = 0; u < U; u++) idx[u] = mix(gid * U + u + 12345u) & nchunks_mask;
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = buf[(size_t)idx[u] * LANES + g];
#pragma unroll
    for (int u = 0; u < U; u++) {
      // group-uniform next index: take lane 0's word via DPP-free trick (all lanes read .x of piece 0?)
      uint32_t w = __shfl(v[u].x, (threadIdx.x & 63) & ~(LANES - 1));
      acc += v[u].y ^ v[u].w;
      idx[u] = mix(w ^ idx[u] ^ ((gid * U + u) * 0x9E3779B9u + (uint32_t)s * 0x85EBCA6Bu)) & nchunks_mask;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void fill(uint4 *buf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = mix((uint32_t)i); buf[i] = make_uint4(h, h * 3u, h * 5u, h * 7u); }
}
template <int LANES, int U>
void run(const uint4 *buf, size_t bytes, int blocks, int steps, uint32_t *out) {
  uint32_t nchunks = (uint32_t)(bytes / (LANES * 16));
  uint32_t mask = nchunks - 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather<LANES, U><<<blocks, 256>>>(buf, mask, 8, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  gather<LANES, U><<<blocks, 256>>>(buf, mask, steps, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double groups = (double)blocks * 256 / LANES;
  double chunks = groups * U * steps;
  printf("buf %5zu MiB  chunk %3d B  U=%d  blocks=%5d  %8.3f ms  %7.2f Gchunk/s  %7.2f TB/s  step-latency %6.2f us\n",
         bytes >> 20, LANES * 16, U, blocks, ms, chunks / ms / 1e6, chunks * LANES * 16 / ms / 1e9,
         ms * 1e3 / steps);
}
int main(int argc, char **argv) {
  size_t mib = argc > 1 ? atoi(argv[1]) : 512;
  size_t bytes = mib << 20;
  uint4 *buf; uint32_t *out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
  fill<<<(unsigned)((bytes / 16 + 255) / 256), 256>>>(buf, bytes / 16);
  CK(hipDeviceSynchronize());
  int steps = 256;
  for (int blocks : {1, 16, 256, 512, 1024, 2048}) {
    run<8, 1>(buf, bytes, blocks, steps, out);
    run<8, 2>(buf, bytes, blocks, steps, out);
    run<8, 4>(buf, bytes, blocks, steps, out);
    run<8, 8>(buf, bytes, blocks, steps, out);
    run<4, 2>(buf, bytes, blocks, steps, out);
    run<4, 4>(buf, bytes, blocks, steps, out);
    run<4, 8>(buf, bytes, blocks, steps, out);
    run<2, 4>(buf, bytes, blocks, steps, out);
    run<2, 8>(buf, bytes, blocks, steps, out);
  }
  return 0;
}
