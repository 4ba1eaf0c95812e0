// Issue rate of the vector-ALU instructions the lane-per-walk record visit is made of (gfx950): a wave runs 16
// independent chains of ONE instruction, 8 waves per SIMD, every CU busy; prints wave-instructions per SIMD cycle
// relative to v_xor_b32 (1.00 = full rate: one wave instruction per 4 cycles) -- which instructions are passes of 8 or
// 16 cycles decides whether "fewer instructions" is "less time" (round 6: the packed-pair visit had 29 % fewer vector
// instructions per record and was 5 % slower).
//   hipcc -O2 --offload-arch=gfx950 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define CHAINS 16
#define BODY(ASM3)                                                                                      \
  uint32_t r[CHAINS];                                                                                   \
  _Pragma("unroll") for (int j = 0; j < CHAINS; j++) r[j] = seed + j * 0x9E3779B9u + threadIdx.x;        \
  uint32_t b = seed * 3u + 1u, c = seed ^ 0x05040100u;                                                   \
  for (int it = 0; it < iters; it++) {                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                     \
      _Pragma("unroll") for (int j = 0; j < CHAINS; j++) { ASM3 }                                       \
    }                                                                                                   \
  }                                                                                                     \
  uint32_t acc = 0;                                                                                     \
  _Pragma("unroll") for (int j = 0; j < CHAINS; j++) acc ^= r[j];                                        \
  if (acc == 0x12345u) out[0] = acc;

#define K3(name, text)                                                                                  \
  __global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed, int iters) {                \
    BODY(asm volatile(text : "+v"(r[j]) : "v"(b), "v"(c));)                                             \
  }
K3(k_xor, "v_xor_b32 %0, %0, %1")
K3(k_perm, "v_perm_b32 %0, %0, %1, %2")
K3(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
K3(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
K3(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
K3(k_lshl_or, "v_lshl_or_b32 %0, %0, 16, %2")
K3(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
K3(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
K3(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
K3(k_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc")
K3(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
K3(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
K3(k_cmp_cnd_vcc, "v_cmp_lt_u32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc")
K3(k_cmp_cnd_sgpr, "v_cmp_lt_u32_e64 s[20:21], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
K3(k_cmp_xor_cnd_vcc, "v_cmp_lt_u32_e32 vcc, %1, %0\n v_xor_b32 %0, %0, %2\n v_cndmask_b32_e32 %0, %0, %1, vcc")
K3(k_addc, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
K3(k_add_co, "v_add_co_u32_e32 %0, vcc, %0, %1")
K3(k_cmp, "v_cmp_eq_u32_e32 vcc, %0, %1")
K3(k_cmp_sgpr, "v_cmp_eq_u32_e64 s[20:21], %0, %1")
K3(k_and, "v_and_b32 %0, %0, %1")
K3(k_or, "v_or_b32 %0, %0, %1")
K3(k_add, "v_add_u32 %0, %0, %1")
K3(k_sub, "v_sub_u32 %0, %0, %1")
K3(k_lshl, "v_lshlrev_b32 %0, 3, %0")
K3(k_not, "v_not_b32 %0, %0")
K3(k_mov, "v_mov_b32 %0, %1")
K3(k_or3, "v_or3_b32 %0, %0, %1, %2")
K3(k_min, "v_min_u32 %0, %0, %1")
K3(k_xor_e64, "v_xor_b32_e64 %0, %0, %1")
K3(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
K3(k_bfe, "v_bfe_u32 %0, %0, 3, 7")
K3(k_lshr, "v_lshrrev_b32 %0, 3, %0")
K3(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
K3(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
K3(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
K3(k_add3, "v_add3_u32 %0, %0, %1, %2")
K3(k_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
K3(k_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
K3(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
K3(k_pack, "v_pack_b32_f16 %0, %0, %1")
K3(k_readlane_like, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

// 64-bit shifts and the 64-bit multiply-add: register pairs
__global__ __launch_bounds__(256) void k_lshl64(uint32_t *out, uint32_t seed, int iters) {
  unsigned long long r[CHAINS];
#pragma unroll
  for (int j = 0; j < CHAINS; j++) r[j] = seed + j * 0x9E3779B97F4A7C15ull + threadIdx.x;
  uint32_t b = (seed & 3u) + 1u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int j = 0; j < CHAINS; j++) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(r[j]) : "v"(b));
    }
  }
  unsigned long long acc = 0;
#pragma unroll
  for (int j = 0; j < CHAINS; j++) acc ^= r[j];
  if (acc == 0x12345ull) out[0] = (uint32_t)acc;
}
__global__ __launch_bounds__(256) void k_lshl_add64(uint32_t *out, uint32_t seed, int iters) {
  unsigned long long r[CHAINS];
#pragma unroll
  for (int j = 0; j < CHAINS; j++) r[j] = seed + j * 0x9E3779B97F4A7C15ull + threadIdx.x;
  unsigned long long b = seed;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int j = 0; j < CHAINS; j++) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(r[j]) : "v"(b));
    }
  }
  unsigned long long acc = 0;
#pragma unroll
  for (int j = 0; j < CHAINS; j++) acc ^= r[j];
  if (acc == 0x12345ull) out[0] = (uint32_t)acc;
}

template <typename K>
static double run(K kern, uint32_t *d, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 7u, 16);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 7u, iters);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms;
}
int main() {
  uint32_t *d = nullptr;
  CK(hipMalloc(&d, 64));
  const int iters = 4096;
  // wave instructions per SIMD: 8 blocks per CU x 4 waves = 8 waves per SIMD, iters * 4 * CHAINS each
  const double per_simd = 8.0 * iters * 4.0 * CHAINS;
  const double base = run(k_xor, d, iters);
  printf("v_xor_b32: %.3f ms for %.0f wave instructions per SIMD -> %.2f cycles each at 2.4 GHz\n", base, per_simd,
         base * 1e-3 * 2.4e9 / per_simd);
#define R(k) printf("%-16s %.2f x the time of v_xor_b32\n", #k, run(k, d, iters) / base)
  R(k_xor); R(k_xor_e64); R(k_and); R(k_or); R(k_add); R(k_sub); R(k_lshl); R(k_not); R(k_mov); R(k_or3); R(k_min); R(k_lshl_add);
  R(k_cndmask_e32); R(k_cndmask_sgpr); R(k_cndmask_e64_vcc); R(k_cmp); R(k_cmp_sgpr);
  printf("pairs / triples (time of the whole group):\n");
  R(k_cmp_cnd_vcc); R(k_cmp_cnd_sgpr); R(k_cmp_xor_cnd_vcc); R(k_addc); R(k_add_co);
  R(k_perm); R(k_bitop3); R(k_bfi); R(k_and_or); R(k_lshl_or); R(k_alignbit); R(k_bcnt); R(k_cndmask); R(k_bfe); R(k_lshr);
  R(k_mul_hi); R(k_mul_lo); R(k_mad_u24); R(k_add3); R(k_sdwa); R(k_dpp); R(k_mbcnt); R(k_pack); R(k_readlane_like);
  R(k_lshl64); R(k_lshl_add64);
  return 0;
}
