// Microbenchmark: dependent random gathers, the access pattern of the count / locate kernels.
//
//   gather<LANES,U>   every LANES-lane group runs U independent chains; a step loads one
//                     LANES*16-byte aligned chunk at a data-dependent address (LANES = 1: every lane
//                     its own 16-byte probe -- the shape of a lane-wise B / B' probe of the RLFM index)
//   tgather<Q>        the "endpoint per lane" shape of the round-2 kernels: an 8-lane group owns 8
//                     chains, lane q holds chain q's index; per round the group walks q = 0..Q-1,
//                     broadcasts chain q's index (ds_bpermute), all 8 lanes load their 16-byte piece of
//                     that 128-byte line (Q lines in flight per lane), then each line is reduced with
//                     popcounts + three DPP adds and handed back to lane q
//
// Build: hipcc -O3 --offload-arch=gfx950 gather.hip -o gather ; run: ./gather [MiB ...]
// Output committed per round as profiles/microbench/gather_<round>.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_hm(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t group_sum(uint32_t v) { v += dpp_xor1(v); v += dpp_xor2(v); v += dpp_hm(v); return v; }

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
      uint32_t w = LANES == 1 ? v[u].x : (uint32_t)__shfl((int)v[u].x, (int)((threadIdx.x & 63) & ~(LANES - 1)));
      acc += v[u].y ^ v[u].w;
      idx[u] = mix(w ^ idx[u] ^ ((gid * U + u) * 0x9E3779B9u + (uint32_t)s * 0x85EBCA6Bu)) & nchunks_mask;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// endpoint-per-lane: 8 chains per 8-lane group, Q of them advanced per round (Q = 8: all)
template <int Q, int WAVES>
__global__ __launch_bounds__(256, WAVES) void tgather(const uint4 *__restrict__ buf, uint32_t nlines_mask,
                                                      int steps, uint32_t *out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, base = lane & ~7u;
  uint32_t idx = mix(tid + 12345u) & nlines_mask;      // this lane's chain
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int h = 0; h < 8; h += Q) {
      uint4 v[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const uint32_t li = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + h + q) << 2), (int)idx);
        v[q] = buf[(size_t)li * 8u + g];
      }
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const uint32_t r = group_sum(__popc(v[q].x & 0xFFFFu) + __popc(v[q].y & v[q].z) + (g == 3 ? v[q].w : 0u));
        if (g == (uint32_t)(h + q)) { acc += r; idx = mix(r ^ idx ^ (tid * 0x9E3779B9u + (uint32_t)s * 0x85EBCA6Bu)) & nlines_mask; }
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ void fill(uint4 *buf, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { uint32_t h = mix((uint32_t)i ^ (uint32_t)(i >> 32)); buf[i] = make_uint4(h, h * 3u, h * 5u, h * 7u); }
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
  printf("gather  buf %6zu MiB  chunk %3d B  U=%d  blocks=%5d  %8.3f ms  %7.2f Gchunk/s  %7.2f TB/s  step-latency %6.2f us\n",
         bytes >> 20, LANES * 16, U, blocks, ms, chunks / ms / 1e6, chunks * LANES * 16 / ms / 1e9,
         ms * 1e3 / steps);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}
template <int Q, int WAVES>
void trun(const uint4 *buf, size_t bytes, int blocks, int steps, uint32_t *out) {
  uint32_t mask = (uint32_t)(bytes / 128) - 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  tgather<Q, WAVES><<<blocks, 256>>>(buf, mask, 4, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  tgather<Q, WAVES><<<blocks, 256>>>(buf, mask, steps, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double lines = (double)blocks * 256 * steps;
  printf("tgather buf %6zu MiB  line 128 B  Q=%d waves/SIMD<=%d  blocks=%5d  %8.3f ms  %7.2f Gline/s  %7.2f TB/s  round-latency %6.2f us\n",
         bytes >> 20, Q, WAVES, blocks, ms, lines / ms / 1e6, lines * 128 / ms / 1e9, ms * 1e3 / steps);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}
int main(int argc, char **argv) {
  uint32_t *out;
  CK(hipMalloc(&out, 4));
  const int steps = 256;
  for (int ai = 1; ai < (argc > 1 ? argc : 2); ai++) {
    size_t mib = argc > 1 ? (size_t)atoll(argv[ai]) : 512;
    size_t bytes = mib << 20;
    uint4 *buf;
    CK(hipMalloc(&buf, bytes));
    fill<<<4096, 256>>>(buf, bytes / 16);
    CK(hipDeviceSynchronize());
    for (int blocks : {256, 1024, 2048}) {
      run<8, 1>(buf, bytes, blocks, steps, out);
      run<8, 2>(buf, bytes, blocks, steps, out);
      run<8, 4>(buf, bytes, blocks, steps, out);
      run<8, 8>(buf, bytes, blocks, steps, out);
      run<4, 4>(buf, bytes, blocks, steps, out);
      run<2, 4>(buf, bytes, blocks, steps, out);
      run<2, 8>(buf, bytes, blocks, steps, out);
      run<1, 1>(buf, bytes, blocks, steps, out);
      run<1, 2>(buf, bytes, blocks, steps, out);
      run<1, 4>(buf, bytes, blocks, steps, out);
    }
    for (int blocks : {512, 1024, 1280, 2048}) {
      trun<2, 8>(buf, bytes, blocks, steps, out);
      trun<4, 8>(buf, bytes, blocks, steps, out);
      trun<8, 8>(buf, bytes, blocks, steps, out);
      trun<8, 4>(buf, bytes, blocks, steps, out);
    }
    CK(hipFree(buf));
  }
  return 0;
}
