/* Plain-C use of the ABI (what a cgo / Rust `extern "C"` binding links against): build an index
 * from a host text, count a batch, locate it, read a few characters back, free.  Compiled and
 * linked on CPU by tests/test_abi_cpu.py; run on the GPU box by tests/test_gpu_cpp_mirror.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fmx.h"

int main(void) {
  const char *t = "mississippi";
  const uint64_t n = strlen(t) + 1;                 /* the text ends with its terminator */
  fmx_index *idx = NULL;
  int rc = fmx_build(t, n, 1, 255, FMX_KIND_FM, 2, 0, 0, &idx);
  if (rc != FMX_OK) { fprintf(stderr, "build: %s\n", fmx_last_error()); return 2; }

  const char *pats = "ssi" "pp" "x";
  const uint64_t off[4] = {0, 3, 5, 6};
  uint64_t s[3], e[3], cnt[3];
  rc = fmx_count_batch(idx, pats, off, 3, NULL, s, e, cnt);
  if (rc != FMX_OK) { fprintf(stderr, "count: %s\n", fmx_last_error()); return 3; }
  if (cnt[0] != 2 || cnt[1] != 1 || cnt[2] != 0) return 4;

  uint64_t hits_off[4] = {0, 0, 0, 0}, pos[8];
  for (int k = 0; k < 3; k++) hits_off[k + 1] = hits_off[k] + cnt[k];
  rc = fmx_locate_batch(idx, s, e, 3, hits_off, pos);
  if (rc != FMX_OK) { fprintf(stderr, "locate: %s\n", fmx_last_error()); return 5; }
  /* "ssi" occurs at 2 and 5 (suffix-array order: 5 then 2), "pp" at 8 */
  if (pos[0] != 5 || pos[1] != 2 || pos[2] != 8) return 6;

  uint8_t back[4];
  uint64_t got = 0, row = s[1];                     /* the match of "pp": read "issi" backwards */
  rc = fmx_extract_batch(idx, &row, 1, 4, 0, back, &got, NULL);
  if (rc != FMX_OK || got != 4 || memcmp(back, "issi", 4) != 0) return 7;

  printf("ok len=%llu bytes=%llu\n", (unsigned long long)fmx_len(idx),
         (unsigned long long)fmx_index_bytes(idx));
  fmx_free(idx);
  return 0;
}
