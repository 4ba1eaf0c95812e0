/* One batch over three replicas of an index from a plain C99 caller (include/fmx.h: fmx_replicate,
 * fmx_count_batch_multi, fmx_locate_batch_multi -- BASELINE config 5 behind the C ABI).  The replicas sit on the
 * devices named on the command line (default: three on device 0 -- the one-GPU test box); the sharded results must be
 * the one-handle results, byte for byte.  Compiled and linked on CPU by tests/test_abi_cpu.py; run on the GPU box by
 * tests/test_gpu_multi_abi.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmx.h"

static uint64_t mix(uint64_t x) {                    /* SplitMix64 finaliser */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int main(int argc, char **argv) {
  enum { N = 200001, NPAT = 1003, G = 3 };
  uint8_t *t = (uint8_t *)malloc(N);
  for (uint64_t i = 0; i + 1 < N; i++) t[i] = (uint8_t)(1 + (mix(i) & 3));
  t[N - 1] = 0;
  fmx_index *idx[G] = {NULL, NULL, NULL};
  int rc = fmx_build(t, N, 1, 4, FMX_KIND_FM, 2, 0, argc > 1 ? atoi(argv[1]) : 0, &idx[0]);
  if (rc != FMX_OK) { fprintf(stderr, "build: %s\n", fmx_last_error()); return 2; }
  for (int r = 1; r < G; r++) {
    rc = fmx_replicate(idx[0], argc > 1 + r ? atoi(argv[1 + r]) : 0, &idx[r]);
    if (rc != FMX_OK) { fprintf(stderr, "replicate: %s\n", fmx_last_error()); return 3; }
    if (fmx_len(idx[r]) != N || fmx_level(idx[r]) != 2) return 4;
  }
  /* patterns: substrings of the text, 0..12 symbols (a few empty ones), back to back in one buffer */
  uint64_t *off = (uint64_t *)malloc((NPAT + 1) * sizeof *off);
  uint8_t *pat = (uint8_t *)malloc((size_t)NPAT * 12);
  off[0] = 0;
  for (uint64_t k = 0; k < NPAT; k++) {
    const uint64_t len = mix(k * 2 + 1) % 13, at = mix(k * 2) % (N - 13);
    memcpy(pat + off[k], t + at, len);
    off[k + 1] = off[k] + len;
  }
  uint64_t *one = (uint64_t *)malloc(3 * NPAT * sizeof *one), *many = (uint64_t *)malloc(3 * NPAT * sizeof *many);
  rc = fmx_count_batch(idx[0], pat, off, NPAT, NULL, one, one + NPAT, one + 2 * NPAT);
  if (rc != FMX_OK) { fprintf(stderr, "count: %s\n", fmx_last_error()); return 5; }
  rc = fmx_count_batch_multi(idx, G, pat, off, NPAT, NULL, many, many + NPAT, many + 2 * NPAT);
  if (rc != FMX_OK) { fprintf(stderr, "count_multi: %s\n", fmx_last_error()); return 6; }
  if (memcmp(one, many, 3 * NPAT * sizeof *one) != 0) return 7;
  /* the shards are the contiguous ranges of fmx_shard_range and cover the batch */
  uint64_t lo, hi, prev = 0;
  for (uint32_t r = 0; r < G; r++) {
    fmx_shard_range(NPAT, G, r, &lo, &hi);
    if (lo != prev || hi < lo) return 8;
    prev = hi;
  }
  if (prev != NPAT) return 9;
  /* locate: every match of every pattern whose count is small, in the reference's order */
  uint64_t *s = one, *e = one + NPAT, *cnt = one + 2 * NPAT;
  uint64_t *hoff = (uint64_t *)malloc((NPAT + 1) * sizeof *hoff);
  hoff[0] = 0;
  for (uint64_t k = 0; k < NPAT; k++) {
    if (cnt[k] > 5000) e[k] = s[k] + 5000;           /* (the empty pattern matches every row) */
    hoff[k + 1] = hoff[k] + (e[k] - s[k]);
  }
  const uint64_t total = hoff[NPAT];
  uint64_t *p1 = (uint64_t *)malloc((total + 1) * sizeof *p1), *p3 = (uint64_t *)malloc((total + 1) * sizeof *p3);
  memset(p3, 0xFF, (total + 1) * sizeof *p3);
  rc = fmx_locate_batch(idx[0], s, e, NPAT, hoff, p1);
  if (rc != FMX_OK) { fprintf(stderr, "locate: %s\n", fmx_last_error()); return 10; }
  rc = fmx_locate_batch_multi(idx, G, s, e, NPAT, hoff, p3);
  if (rc != FMX_OK) { fprintf(stderr, "locate_multi: %s\n", fmx_last_error()); return 11; }
  if (memcmp(p1, p3, total * sizeof *p1) != 0) return 12;
  for (uint64_t h = 0; h < total; h++)
    if (p1[h] >= N) return 13;
  printf("ok multi replicas=%d patterns=%d hits=%llu\n", (int)G, (int)NPAT, (unsigned long long)total);
  for (int r = 0; r < G; r++) fmx_free(idx[r]);
  free(t); free(off); free(pat); free(one); free(many); free(hoff); free(p1); free(p3);
  return 0;
}
