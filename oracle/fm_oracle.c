/*
 * fm_oracle.c -- CPU ORACLE (test infrastructure, see fm_oracle.h).
 *
 * Restates, in plain C, the algorithm of the reference's count/locate path.
 * Citations are file:line under /root/reference.  The bit-vector / wavelet
 * matrix layer restates the *behaviour* of vers-vecs ^1.10.1 (Cargo.toml:16),
 * whose source is not on this machine: per-level bit planes (MSB first, zeros
 * stably before ones), 512-bit blocks + 8192-bit superblocks, rank clamped at
 * len, select1 returning len when out of range.
 */
#define _GNU_SOURCE
#include "fm_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#ifdef _OPENMP
#include <omp.h>
#endif

const char *orc_error_message(int code) {
  switch (code) {
    case ORC_OK: return "ok";
    case ORC_ERR_START_ZERO: /* sais.rs:129-131 */
      return "the given text must not start with zero character";
    case ORC_ERR_END_ZERO: /* sais.rs:135-137 */
      return "the given text must end with exactly one zero character";
    case ORC_ERR_SYMBOL_RANGE: return "symbol exceeds max_character";
    default: return "invalid argument";
  }
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Thread placement for the batch drivers (bench.py's cpu_baseline): when switched on, the threads of a batch
 * pin themselves one per CPU, spread evenly over the CPUs this process is allowed to use (so over both sockets
 * and, while there are fewer threads than cores, one per core); every thread of the team gets the calling thread's
 * mask back when the batch is done.  Done here rather than with OMP_PROC_BIND because that also narrows the mask of the calling
 * (Python) thread for good -- and with it the mask every thread created later inherits. */
static int g_spread = 0;
void orc_set_thread_spread(int on) { g_spread = on; }
typedef struct { cpu_set_t orig; int cpus[CPU_SETSIZE]; int ncpu; int on; } orc_place;
static void orc_place_begin(orc_place *p) {
  p->on = 0;
  p->ncpu = 0;
  if (!g_spread || sched_getaffinity(0, sizeof p->orig, &p->orig) != 0) return;
  for (int c = 0; c < CPU_SETSIZE; c++)
    if (CPU_ISSET(c, &p->orig)) p->cpus[p->ncpu++] = c;
  p->on = p->ncpu > 1;
}
static void orc_place_thread(const orc_place *p, int t, int nt) {
  if (!p->on) return;
  cpu_set_t one;
  CPU_ZERO(&one);
  CPU_SET(p->cpus[(int)(((long long)t * p->ncpu) / (nt > 0 ? nt : 1)) % p->ncpu], &one);
  (void)sched_setaffinity(0, sizeof one, &one);
}
static void orc_place_end(const orc_place *p) {
  if (p->on) (void)sched_setaffinity(0, sizeof p->orig, &p->orig);
}

/* threads an OpenMP team of `nthreads` really gets here (bench.py's cpu_baseline reports it) */
int orc_team_size(int nthreads) {
  int t = 1;
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
  {
#pragma omp single
    t = omp_get_num_threads();
  }
#endif
  return t;
}

/* util.rs:1-3 */
uint32_t orc_log2(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }
/* text.rs:61-63 */
uint32_t orc_max_bits(uint64_t max_character) { return orc_log2(max_character) + 1; }

/* sais.rs:115-139 */
int orc_validate_text(const uint8_t *text, uint64_t n) {
  if (n == 0 || n == 1) return ORC_OK; /* sais.rs:121-126 */
  if (text[0] == 0) return ORC_ERR_START_ZERO;
  /* rposition of the last non-zero must be n-2 (sais.rs:133-134) */
  int64_t last = -1;
  for (int64_t i = (int64_t)n - 1; i >= 0; i--)
    if (text[i] != 0) { last = i; break; }
  if (last != (int64_t)n - 2) return ORC_ERR_END_ZERO;
  return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* suffix array                                                        */
/* ------------------------------------------------------------------ */
typedef struct { const uint8_t *t; uint64_t n; } naive_ctx;
static int naive_cmp(const void *a, const void *b, void *c) {
  const naive_ctx *x = (const naive_ctx *)c;
  uint64_t i = *(const uint32_t *)a, j = *(const uint32_t *)b;
  uint64_t li = x->n - i, lj = x->n - j, l = li < lj ? li : lj;
  int r = memcmp(x->t + i, x->t + j, l); /* slice comparison: sais.rs:553 */
  if (r) return r;
  return li < lj ? -1 : (li > lj ? 1 : 0);
}
void orc_suffix_array_naive(const uint8_t *text, uint64_t n, uint32_t *sa) {
  for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
  naive_ctx c = {text, n};
  qsort_r(sa, n, sizeof(uint32_t), naive_cmp, &c);
}

typedef struct { const uint32_t *rank; uint64_t h, n; } dbl_ctx;
static int dbl_cmp(const void *a, const void *b, void *c) {
  const dbl_ctx *x = (const dbl_ctx *)c;
  uint64_t i = *(const uint32_t *)a + x->h, j = *(const uint32_t *)b + x->h;
  /* a suffix that ends first is the smaller one (slice order) */
  int64_t ri = i < x->n ? (int64_t)x->rank[i] : -1;
  int64_t rj = j < x->n ? (int64_t)x->rank[j] : -1;
  return ri < rj ? -1 : (ri > rj ? 1 : 0);
}
void orc_suffix_array(const uint8_t *text, uint64_t n, uint32_t *sa) {
  if (n == 0) return;
  uint32_t *rank = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint32_t *nrank = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint64_t cnt[257];
  memset(cnt, 0, sizeof cnt);
  for (uint64_t i = 0; i < n; i++) cnt[text[i] + 1]++;
  for (int c = 0; c < 256; c++) cnt[c + 1] += cnt[c];
  for (uint64_t i = 0; i < n; i++) rank[i] = (uint32_t)cnt[text[i]];
  {
    uint64_t pos[256];
    for (int c = 0; c < 256; c++) pos[c] = cnt[c];
    for (uint64_t i = 0; i < n; i++) sa[pos[text[i]]++] = (uint32_t)i;
  }
  for (uint64_t h = 1;; h *= 2) {
    dbl_ctx ctx = {rank, h, n};
    int any = 0;
    uint64_t g0 = 0;
    while (g0 < n) {
      uint64_t g1 = g0 + 1;
      uint32_t r = rank[sa[g0]];
      while (g1 < n && rank[sa[g1]] == r) g1++;
      if (g1 - g0 > 1) {
        any = 1;
        qsort_r(sa + g0, g1 - g0, sizeof(uint32_t), dbl_cmp, &ctx);
        /* new ranks: sub-group head index */
        uint64_t head = g0;
        for (uint64_t p = g0; p < g1; p++) {
          if (p > g0 && dbl_cmp(&sa[p - 1], &sa[p], &ctx) != 0) head = p;
          nrank[sa[p]] = (uint32_t)head;
        }
      } else {
        nrank[sa[g0]] = (uint32_t)g0;
      }
      g0 = g1;
    }
    memcpy(rank, nrank, n * sizeof(uint32_t));
    if (!any) break;
  }
  free(rank);
  free(nrank);
}

/* sais.rs:9-32 (count_chars + get_bucket_start_pos) */
void orc_bucket_start(const uint8_t *text, uint64_t n, uint64_t max_character,
                      uint64_t *cs) {
  uint64_t m = max_character + 1;
  uint64_t *occ = (uint64_t *)calloc(m, sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) occ[text[i]]++;
  uint64_t sum = 0;
  for (uint64_t c = 0; c < m; c++) { cs[c] = sum; sum += occ[c]; }
  free(occ);
}

/* fm_index.rs:44-58 */
void orc_bwt(const uint8_t *text, uint64_t n, const uint32_t *sa, uint8_t *bwt) {
  for (uint64_t i = 0; i < n; i++) {
    uint32_t k = sa[i];
    bwt[i] = k > 0 ? text[k - 1] : 0;
  }
}

/* ------------------------------------------------------------------ */
/* RsVec                                                               */
/* ------------------------------------------------------------------ */
#define BLK_WORDS 8u     /* 512 bits  */
#define SUP_BLKS 16u     /* 8192 bits */
void orc_rsvec_build(orc_rsvec *v, uint64_t *words, uint64_t len) {
  v->len = len;
  v->nwords = (len + 63) / 64;
  v->words = words;
  uint64_t nblk = v->nwords / BLK_WORDS + 1, nsup = nblk / SUP_BLKS + 1;
  v->blk = (uint16_t *)malloc(nblk * sizeof(uint16_t));
  v->sup = (uint64_t *)malloc(nsup * sizeof(uint64_t));
  uint64_t total = 0, insup = 0;
  for (uint64_t b = 0; b < nblk; b++) {
    if (b % SUP_BLKS == 0) { v->sup[b / SUP_BLKS] = total; insup = 0; }
    v->blk[b] = (uint16_t)insup;
    uint64_t w0 = b * BLK_WORDS, w1 = w0 + BLK_WORDS;
    if (w1 > v->nwords) w1 = v->nwords;
    uint64_t c = 0;
    for (uint64_t w = w0; w < w1; w++) c += (uint64_t)__builtin_popcountll(words[w]);
    total += c;
    insup += c;
  }
  v->ones = total;
}
void orc_rsvec_free(orc_rsvec *v) {
  free(v->words); free(v->blk); free(v->sup);
  memset(v, 0, sizeof *v);
}
uint64_t orc_rsvec_rank1(const orc_rsvec *v, uint64_t i) {
  if (i >= v->len) return v->ones; /* clamp */
  uint64_t w = i >> 6, b = w / BLK_WORDS;
  uint64_t r = v->sup[b / SUP_BLKS] + v->blk[b];
  for (uint64_t k = b * BLK_WORDS; k < w; k++) r += (uint64_t)__builtin_popcountll(v->words[k]);
  uint64_t m = (i & 63) ? (v->words[w] & ((1ull << (i & 63)) - 1)) : 0;
  return r + (uint64_t)__builtin_popcountll(m);
}
uint64_t orc_rsvec_rank0(const orc_rsvec *v, uint64_t i) {
  uint64_t ii = i > v->len ? v->len : i;
  return ii - orc_rsvec_rank1(v, i);
}
int orc_rsvec_get(const orc_rsvec *v, uint64_t i) { return (int)((v->words[i >> 6] >> (i & 63)) & 1); }
uint64_t orc_rsvec_select1(const orc_rsvec *v, uint64_t k) {
  if (k >= v->ones) return v->len;
  uint64_t nblk = v->nwords / BLK_WORDS + 1, nsup = nblk / SUP_BLKS + 1;
  uint64_t lo = 0, hi = nsup; /* last superblock with sup <= k */
  while (hi - lo > 1) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (v->sup[mid] <= k) lo = mid; else hi = mid;
  }
  uint64_t rem = k - v->sup[lo];
  uint64_t b = lo * SUP_BLKS, bend = b + SUP_BLKS;
  if (bend > nblk) bend = nblk;
  while (b + 1 < bend && v->blk[b + 1] <= rem) b++;
  rem -= v->blk[b];
  uint64_t w = b * BLK_WORDS;
  for (;; w++) {
    uint64_t c = (uint64_t)__builtin_popcountll(v->words[w]);
    if (rem < c) break;
    rem -= c;
  }
  uint64_t x = v->words[w];
  for (uint64_t j = 0; j < rem; j++) x &= x - 1;
  return w * 64 + (uint64_t)__builtin_ctzll(x);
}

/* ------------------------------------------------------------------ */
/* WaveletMatrix                                                       */
/* ------------------------------------------------------------------ */
#define ORC_DEFINE_WM_BUILD(NAME, T) \
void NAME(orc_wm *w, const T *vals, uint64_t n, uint32_t bits) { \
  w->bits = bits; \
  w->len = n; \
  w->lv = (orc_rsvec *)calloc(bits, sizeof(orc_rsvec)); \
  w->zeros = (uint64_t *)calloc(bits, sizeof(uint64_t)); \
  T *cur = (T *)malloc((n ? n : 1) * sizeof(T)), *nxt = (T *)malloc((n ? n : 1) * sizeof(T)); \
  memcpy(cur, vals, n * sizeof(T)); \
  uint64_t nwords = (n + 63) / 64; \
  int nt = orc_max_threads(); \
  uint64_t *zc = (uint64_t *)calloc((size_t)nt + 1, sizeof(uint64_t)); \
  for (uint32_t l = 0; l < bits; l++) { \
    uint32_t sh = bits - 1 - l; \
    uint64_t *words = (uint64_t *)calloc(nwords + BLK_WORDS, sizeof(uint64_t)); \
    /* chunk boundaries are multiples of 64 so word writes never race */ \
    uint64_t chunk = ((n / (uint64_t)nt) / 64 + 1) * 64; \
_Pragma("omp parallel for schedule(static, 1)") \
    for (int t = 0; t < nt; t++) { \
      uint64_t a = (uint64_t)t * chunk, b = a + chunk; \
      if (a > n) a = n; \
      if (b > n) b = n; \
      uint64_t z = 0; \
      for (uint64_t i = a; i < b; i++) { \
        uint64_t bit = ((uint64_t)cur[i] >> sh) & 1u; \
        words[i >> 6] |= bit << (i & 63); \
        z += bit ^ 1u; \
      } \
      zc[t + 1] = z; \
    } \
    for (int t = 0; t < nt; t++) zc[t + 1] += zc[t]; \
    uint64_t zeros = zc[nt]; \
_Pragma("omp parallel for schedule(static, 1)") \
    for (int t = 0; t < nt; t++) { \
      uint64_t a = (uint64_t)t * chunk, b = a + chunk; \
      if (a > n) a = n; \
      if (b > n) b = n; \
      uint64_t pz = zc[t], po = zeros + (a - zc[t]); \
      for (uint64_t i = a; i < b; i++) { \
        if (((uint64_t)cur[i] >> sh) & 1u) nxt[po++] = cur[i]; else nxt[pz++] = cur[i]; \
      } \
    } \
    zc[0] = 0; \
    w->zeros[l] = zeros; \
    orc_rsvec_build(&w->lv[l], words, n); \
    T *tmp = cur; cur = nxt; nxt = tmp; \
  } \
  free(zc); \
  free(cur); \
  free(nxt); \
}
ORC_DEFINE_WM_BUILD(orc_wm_build, uint8_t)
ORC_DEFINE_WM_BUILD(orc_wm_build_w, uint32_t)
void orc_wm_free(orc_wm *w) {
  for (uint32_t l = 0; l < w->bits; l++) orc_rsvec_free(&w->lv[l]);
  free(w->lv); free(w->zeros);
  memset(w, 0, sizeof *w);
}
uint64_t orc_wm_get(const orc_wm *w, uint64_t i) {
  uint64_t v = 0;
  for (uint32_t l = 0; l < w->bits; l++) {
    const orc_rsvec *b = &w->lv[l];
    uint64_t bit = (uint64_t)orc_rsvec_get(b, i);
    v = (v << 1) | bit;
    i = bit ? w->zeros[l] + orc_rsvec_rank1(b, i) : orc_rsvec_rank0(b, i);
  }
  return v;
}
/* rank_u64_unchecked(i, c) = rank_range(0..i, c): both range ends are mapped
 * through every level */
uint64_t orc_wm_rank(const orc_wm *w, uint64_t i, uint64_t c) {
  uint64_t s = 0, e = i;
  for (uint32_t l = 0; l < w->bits; l++) {
    const orc_rsvec *b = &w->lv[l];
    if ((c >> (w->bits - 1 - l)) & 1u) {
      s = w->zeros[l] + orc_rsvec_rank1(b, s);
      e = w->zeros[l] + orc_rsvec_rank1(b, e);
    } else {
      s = orc_rsvec_rank0(b, s);
      e = orc_rsvec_rank0(b, e);
    }
  }
  return e - s;
}

/* select_u64_unchecked(k, c): position of the k-th (0-based) c, by definition
 * (smallest p with rank(p+1, c) == k+1) -- a binary search over rank is enough for a checker */
uint64_t orc_wm_select(const orc_wm *w, uint64_t k, uint64_t c) {
  uint64_t lo = 0, hi = w->len; /* answer in [lo, hi) */
  while (hi - lo > 1) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (orc_wm_rank(w, mid, c) <= k) lo = mid; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------ */
/* SOSampledSuffixArray                                                */
/* ------------------------------------------------------------------ */
static void bv_append_bits(uint64_t *w, uint64_t *pos, uint64_t v, uint64_t nb) {
  uint64_t p = *pos, o = p & 63;
  w[p >> 6] |= v << o;
  if (o + nb > 64) w[(p >> 6) + 1] |= v >> (64 - o);
  *pos = p + nb;
}
static uint64_t bv_get_bits(const uint64_t *w, uint64_t p, uint64_t nb) {
  uint64_t o = p & 63, v = w[p >> 6] >> o;
  if (o + nb > 64) v |= w[(p >> 6) + 1] << (64 - o);
  return nb == 64 ? v : (v & ((1ull << nb) - 1));
}
static void ssa_init(orc_ssa *s, uint64_t n, uint64_t level) {
  memset(s, 0, sizeof *s);
  if (n == 0) return; /* sample.rs:22-24 (Default) */
  s->word_size = orc_log2(n) + 1;    /* sample.rs:27 */
  if (n <= (1ull << level)) level = 0; /* sample.rs:28-31 */
  s->level = level;
  s->len = n;
  s->nsamples = ((n - 1) >> level) + 1; /* sample.rs:33 */
  s->bits = (uint64_t *)calloc((s->nsamples * s->word_size + 63) / 64 + 1, sizeof(uint64_t));
}
void orc_ssa_from_samples64(orc_ssa *s, const uint64_t *samples, uint64_t n, uint64_t level) {
  ssa_init(s, n, level);
  uint64_t pos = 0;
  for (uint64_t i = 0; i < s->nsamples; i++)
    bv_append_bits(s->bits, &pos, samples[i], s->word_size);
}
void orc_ssa_sample(orc_ssa *s, const uint32_t *sa, uint64_t n, uint64_t level) {
  ssa_init(s, n, level);
  uint64_t pos = 0;
  for (uint64_t i = 0; i < s->nsamples; i++) /* sample.rs:35-37 */
    bv_append_bits(s->bits, &pos, sa[i << s->level], s->word_size);
}
void orc_ssa_from_samples(orc_ssa *s, const uint32_t *samples, uint64_t n, uint64_t level) {
  ssa_init(s, n, level);
  uint64_t pos = 0;
  for (uint64_t i = 0; i < s->nsamples; i++)
    bv_append_bits(s->bits, &pos, samples[i], s->word_size);
}
void orc_ssa_free(orc_ssa *s) { free(s->bits); memset(s, 0, sizeof *s); }
int orc_ssa_get(const orc_ssa *s, uint64_t i, uint64_t *out) { /* sample.rs:46-60 */
  if (i >= s->len) return 0;
  if ((i & ((1ull << s->level) - 1)) == 0) {
    *out = bv_get_bits(s->bits, (i >> s->level) * s->word_size, s->word_size);
    return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* FMIndexBackend                                                      */
/* ------------------------------------------------------------------ */
static uint64_t fm_len(const void *p) { return ((const orc_fm *)p)->bw.len; } /* fm_index.rs:78-80 */
static uint64_t fm_get_l(const void *p, uint64_t i) {                           /* fm_index.rs:82-84 */
  return orc_wm_get(&((const orc_fm *)p)->bw, i);
}
static uint64_t fm_lf_map2(const void *p, uint64_t c, uint64_t i) {             /* fm_index.rs:93-95 */
  const orc_fm *f = (const orc_fm *)p;
  return f->cs[c] + orc_wm_rank(&f->bw, i, c);
}
static uint64_t fm_lf_map(const void *p, uint64_t i) {                          /* fm_index.rs:86-91 */
  const orc_fm *f = (const orc_fm *)p;
  uint64_t c = fm_get_l(p, i);
  return f->cs[c] + orc_wm_rank(&f->bw, i, c);
}
static uint64_t fm_get_sa(const void *p, uint64_t i) {                          /* fm_index.rs:127-140 */
  const orc_fm *f = (const orc_fm *)p;
  uint64_t steps = 0, sa;
  for (;;) {
    if (orc_ssa_get(&f->ssa, i, &sa)) return (sa + steps) % f->bw.len;
    i = fm_lf_map(p, i);
    steps++;
  }
}
static uint64_t cs_upper(const uint64_t *cs, uint64_t m, uint64_t v) { /* fm_index.rs:97-112 */
  uint64_t s = 0, e = m;
  while (e - s > 1) {
    uint64_t mid = s + (e - s) / 2;
    if (cs[mid] <= v) s = mid; else e = mid;
  }
  return s;
}
static uint64_t fm_get_f(const void *p, uint64_t i) {                           /* fm_index.rs:97-112 */
  const orc_fm *f = (const orc_fm *)p;
  return cs_upper(f->cs, f->max_character + 1, i);
}
static uint64_t fm_fl_map(const void *p, uint64_t i) {                          /* fm_index.rs:114-120 */
  const orc_fm *f = (const orc_fm *)p;
  uint64_t c = fm_get_f(p, i);
  return orc_wm_select(&f->bw, i - f->cs[c], c);
}
orc_backend orc_fm_backend(orc_fm *f) {
  orc_backend b = {f, fm_get_l, fm_lf_map, fm_lf_map2, fm_len,
                   f->has_locate ? fm_get_sa : NULL, fm_get_f, fm_fl_map, NULL, 0, f->max_character};
  return b;
}
static int check_symbols(const uint8_t *text, uint64_t n, uint64_t max_character) {
  if (max_character >= 255) return ORC_OK;
  for (uint64_t i = 0; i < n; i++)
    if (text[i] > max_character) return ORC_ERR_SYMBOL_RANGE; /* count_chars would panic, sais.rs:18 */
  return ORC_OK;
}
int orc_fm_new(orc_fm **out, const uint8_t *text, uint64_t n, uint64_t max_character,
               int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  int rc = check_symbols(text, n, max_character);
  if (rc) return rc;
  rc = orc_validate_text(text, n);
  if (rc) return rc;
  orc_fm *f = (orc_fm *)calloc(1, sizeof(orc_fm));
  f->max_character = max_character;
  f->cs = (uint64_t *)calloc(max_character + 1, sizeof(uint64_t));
  orc_bucket_start(text, n, max_character, f->cs);          /* fm_index.rs:32 */
  uint32_t *sa = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_suffix_array(text, n, sa);                             /* fm_index.rs:33 */
  uint8_t *bwt = (uint8_t *)malloc(n ? n : 1);
  orc_bwt(text, n, sa, bwt);                                 /* fm_index.rs:34 */
  orc_wm_build(&f->bw, bwt, n, orc_max_bits(max_character));
  if (level >= 0) {
    orc_ssa_sample(&f->ssa, sa, n, (uint64_t)level);         /* frontend.rs:213-221 */
    f->has_locate = 1;
  }
  free(bwt);
  free(sa);
  *out = f;
  return ORC_OK;
}
int orc_fm_from_bwt(orc_fm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                    const uint64_t *cs, const uint32_t *samples, int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  orc_fm *f = (orc_fm *)calloc(1, sizeof(orc_fm));
  f->max_character = max_character;
  f->cs = (uint64_t *)calloc(max_character + 1, sizeof(uint64_t));
  memcpy(f->cs, cs, (max_character + 1) * sizeof(uint64_t));
  orc_wm_build(&f->bw, bwt, n, orc_max_bits(max_character));
  if (level >= 0 && samples) {
    orc_ssa_from_samples(&f->ssa, samples, n, (uint64_t)level);
    f->has_locate = 1;
  }
  *out = f;
  return ORC_OK;
}
/* the same with 64-bit sample values: texts of 2^32 symbols and more (the reference is usize throughout) */
int orc_fm_from_bwt64(orc_fm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                    const uint64_t *cs, const uint64_t *samples, int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  orc_fm *f = (orc_fm *)calloc(1, sizeof(orc_fm));
  f->max_character = max_character;
  f->cs = (uint64_t *)calloc(max_character + 1, sizeof(uint64_t));
  memcpy(f->cs, cs, (max_character + 1) * sizeof(uint64_t));
  orc_wm_build(&f->bw, bwt, n, orc_max_bits(max_character));
  if (level >= 0 && samples) {
    orc_ssa_from_samples64(&f->ssa, samples, n, (uint64_t)level);
    f->has_locate = 1;
  }
  *out = f;
  return ORC_OK;
}
void orc_fm_free(orc_fm *f) {
  if (!f) return;
  orc_wm_free(&f->bw);
  orc_ssa_free(&f->ssa);
  free(f->cs);
  free(f);
}
uint64_t orc_fm_heap_bytes(const orc_fm *f) {
  uint64_t t = (f->max_character + 1) * 8;
  for (uint32_t l = 0; l < f->bw.bits; l++) {
    const orc_rsvec *v = &f->bw.lv[l];
    uint64_t nblk = v->nwords / BLK_WORDS + 1;
    t += v->nwords * 8 + nblk * 2 + (nblk / SUP_BLKS + 1) * 8;
  }
  t += (f->ssa.nsamples * f->ssa.word_size + 7) / 8;
  return t;
}

/* ------------------------------------------------------------------ */
/* RLFMIndexBackend                                                    */
/* ------------------------------------------------------------------ */
static uint64_t rl_len(const void *p) { return ((const orc_rlfm *)p)->len; } /* rlfmi.rs:118-120 */
static uint64_t rl_get_l(const void *p, uint64_t i) {                        /* rlfmi.rs:122-125 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  return orc_wm_get(&f->s, orc_rsvec_rank1(&f->b, i + 1) - 1);
}
static uint64_t rl_lf_map(const void *p, uint64_t i) {                       /* rlfmi.rs:127-133 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  uint64_t c = rl_get_l(p, i);
  uint64_t j = orc_rsvec_rank1(&f->b, i);
  uint64_t nr = orc_wm_rank(&f->s, j, c);
  return orc_rsvec_select1(&f->bp, f->cs[c] + nr) + i - orc_rsvec_select1(&f->b, j);
}
static uint64_t rl_lf_map2(const void *p, uint64_t c, uint64_t i) {          /* rlfmi.rs:135-143 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  uint64_t j = orc_rsvec_rank1(&f->b, i);
  uint64_t nr = orc_wm_rank(&f->s, j, c);
  if (rl_get_l(p, i) != c) return orc_rsvec_select1(&f->bp, f->cs[c] + nr);
  return orc_rsvec_select1(&f->bp, f->cs[c] + nr) + i - orc_rsvec_select1(&f->b, j);
}
static uint64_t rl_get_sa(const void *p, uint64_t i) {                       /* rlfmi.rs:176-189 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  uint64_t steps = 0, sa;
  for (;;) {
    if (orc_ssa_get(&f->ssa, i, &sa)) return (sa + steps) % f->len;
    i = rl_lf_map(p, i);
    steps++;
  }
}
static uint64_t rl_get_f(const void *p, uint64_t i) {                        /* rlfmi.rs:145-158 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  uint64_t r = orc_rsvec_rank1(&f->bp, i + 1) - 1;
  return cs_upper(f->cs, f->max_character + 1, r);
}
static uint64_t rl_fl_map(const void *p, uint64_t i) {                       /* rlfmi.rs:160-169 */
  const orc_rlfm *f = (const orc_rlfm *)p;
  uint64_t c = rl_get_f(p, i);
  uint64_t j = orc_rsvec_rank1(&f->bp, i + 1) - 1;
  uint64_t pp = orc_rsvec_select1(&f->bp, j);
  uint64_t m = orc_wm_select(&f->s, j - f->cs[c], c);
  uint64_t nn = orc_rsvec_select1(&f->b, m);
  return nn + i - pp;
}
orc_backend orc_rlfm_backend(orc_rlfm *f) {
  orc_backend b = {f, rl_get_l, rl_lf_map, rl_lf_map2, rl_len,
                   f->has_locate ? rl_get_sa : NULL, rl_get_f, rl_fl_map, NULL, 0, f->max_character};
  return b;
}
static void bits_set(uint64_t *w, uint64_t i) { w[i >> 6] |= 1ull << (i & 63); }
int orc_rlfm_new(orc_rlfm **out, const uint8_t *text, uint64_t n, uint64_t max_character,
                 int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  int rc = check_symbols(text, n, max_character);
  if (rc) return rc;
  rc = orc_validate_text(text, n);
  if (rc) return rc;
  orc_rlfm *f = (orc_rlfm *)calloc(1, sizeof(orc_rlfm));
  uint64_t m = max_character + 1;                               /* rlfmi.rs:38 */
  f->len = n;
  f->max_character = max_character;
  uint32_t *sa = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_suffix_array(text, n, sa);                                /* rlfmi.rs:39 */
  uint64_t nw = (n + 63) / 64 + BLK_WORDS;
  uint64_t *bw = (uint64_t *)calloc(nw, 8), *bpw = (uint64_t *)calloc(nw, 8);
  uint8_t *heads = (uint8_t *)malloc(n ? n : 1);
  uint64_t r = 0;
  /* run lengths grouped per symbol, in row order (runs_by_char, rlfmi.rs:47) */
  uint64_t *run_len = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint64_t *runs_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  uint64_t c0 = 0;                                              /* rlfmi.rs:41 */
  for (uint64_t i = 0; i < n; i++) {                            /* rlfmi.rs:48-68 */
    uint32_t k = sa[i];
    uint64_t c = k > 0 ? text[k - 1] : text[n - 1];
    if (c0 != c) {
      heads[r] = (uint8_t)c;
      run_len[r] = 1;
      r++;
      bits_set(bw, i);
      runs_of[c]++;
    } else {
      run_len[r - 1]++;
    }
    c0 = c;
  }
  f->runs = r;
  orc_wm_build(&f->s, heads, r, orc_max_bits(max_character));  /* rlfmi.rs:69-70 */
  f->cs = (uint64_t *)calloc(m, sizeof(uint64_t));
  uint64_t acc = 0;
  for (uint64_t c = 0; c < m; c++) { f->cs[c] = acc; acc += runs_of[c]; } /* rlfmi.rs:72-76 */
  /* B': for c ascending, for each run of c in row order: 1 0^{len-1} (rlfmi.rs:71-83) */
  uint64_t *start_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  {
    uint64_t *chars_of = (uint64_t *)calloc(m, sizeof(uint64_t));
    for (uint64_t k = 0; k < r; k++) chars_of[heads[k]] += run_len[k];
    uint64_t a = 0;
    for (uint64_t c = 0; c < m; c++) { start_of[c] = a; a += chars_of[c]; }
    free(chars_of);
  }
  for (uint64_t k = 0; k < r; k++) {
    uint64_t c = heads[k];
    bits_set(bpw, start_of[c]);
    start_of[c] += run_len[k];
  }
  orc_rsvec_build(&f->b, bw, n);                                /* rlfmi.rs:85 */
  orc_rsvec_build(&f->bp, bpw, n);                              /* rlfmi.rs:86 */
  if (level >= 0) {
    orc_ssa_sample(&f->ssa, sa, n, (uint64_t)level);
    f->has_locate = 1;
  }
  free(start_of); free(runs_of); free(run_len); free(heads); free(sa);
  *out = f;
  return ORC_OK;
}
/* Same construction as orc_rlfm_new (rlfmi.rs:41-96), fed with the L column instead of the text:
 * the symbol the loop reads for row i, T[SA[i]-1] (T[n-1] = 0 when SA[i] = 0), IS bwt[i]
 * (fm_index.rs:50-55).  Used by bench.py's cpu_baseline at n = 2^30, where the CPU suffix sort
 * would take longer than the whole benchmark; `samples` (may be NULL) are the SA samples of `level`. */
static int rlfm_from_bwt_any(orc_rlfm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                             const uint32_t *samples32, const uint64_t *samples64, int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  orc_rlfm *f = (orc_rlfm *)calloc(1, sizeof(orc_rlfm));
  uint64_t m = max_character + 1;
  f->len = n;
  f->max_character = max_character;
  uint64_t nw = (n + 63) / 64 + BLK_WORDS;
  uint64_t *bw = (uint64_t *)calloc(nw, 8), *bpw = (uint64_t *)calloc(nw, 8);
  uint64_t r = 0, c0 = 0;                                       /* rlfmi.rs:41 */
  for (uint64_t i = 0; i < n; i++) {                            /* the runs are counted first: the arrays below are r long */
    if (c0 != bwt[i]) r++;
    c0 = bwt[i];
  }
  uint8_t *heads = (uint8_t *)malloc(r ? r : 1);
  uint64_t *run_len = (uint64_t *)malloc((r ? r : 1) * sizeof(uint64_t));
  uint64_t *runs_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  r = 0; c0 = 0;
  for (uint64_t i = 0; i < n; i++) {                            /* rlfmi.rs:48-68 */
    uint64_t c = bwt[i];
    if (c0 != c) {
      heads[r] = (uint8_t)c;
      run_len[r] = 1;
      r++;
      bits_set(bw, i);
      runs_of[c]++;
    } else {
      run_len[r - 1]++;
    }
    c0 = c;
  }
  f->runs = r;
  orc_wm_build(&f->s, heads, r, orc_max_bits(max_character));  /* rlfmi.rs:69-70 */
  f->cs = (uint64_t *)calloc(m, sizeof(uint64_t));
  uint64_t acc = 0;
  for (uint64_t c = 0; c < m; c++) { f->cs[c] = acc; acc += runs_of[c]; } /* rlfmi.rs:72-76 */
  uint64_t *start_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  {
    uint64_t *chars_of = (uint64_t *)calloc(m, sizeof(uint64_t));
    for (uint64_t k = 0; k < r; k++) chars_of[heads[k]] += run_len[k];
    uint64_t a = 0;
    for (uint64_t c = 0; c < m; c++) { start_of[c] = a; a += chars_of[c]; }
    free(chars_of);
  }
  for (uint64_t k = 0; k < r; k++) {                            /* rlfmi.rs:71-83 */
    uint64_t c = heads[k];
    bits_set(bpw, start_of[c]);
    start_of[c] += run_len[k];
  }
  orc_rsvec_build(&f->b, bw, n);                                /* rlfmi.rs:85 */
  orc_rsvec_build(&f->bp, bpw, n);                              /* rlfmi.rs:86 */
  if (level >= 0 && (samples32 || samples64)) {
    if (samples64) orc_ssa_from_samples64(&f->ssa, samples64, n, (uint64_t)level);
    else orc_ssa_from_samples(&f->ssa, samples32, n, (uint64_t)level);
    f->has_locate = 1;
  }
  free(start_of); free(runs_of); free(run_len); free(heads);
  *out = f;
  return ORC_OK;
}
int orc_rlfm_from_bwt(orc_rlfm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                      const uint32_t *samples, int level) {
  return rlfm_from_bwt_any(out, bwt, n, max_character, samples, NULL, level);
}
/* the same with 64-bit sample values: texts of 2^32 symbols and more (the reference is usize throughout) */
int orc_rlfm_from_bwt64(orc_rlfm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                        const uint64_t *samples, int level) {
  return rlfm_from_bwt_any(out, bwt, n, max_character, NULL, samples, level);
}
void orc_rlfm_free(orc_rlfm *f) {
  if (!f) return;
  orc_wm_free(&f->s);
  orc_rsvec_free(&f->b);
  orc_rsvec_free(&f->bp);
  orc_ssa_free(&f->ssa);
  free(f->cs);
  free(f);
}

/* ------------------------------------------------------------------ */
/* FMIndexMultiPiecesBackend (multi_pieces.rs)                          */
/* ------------------------------------------------------------------ */
static uint64_t mp_len(const void *p) { return ((const orc_multi *)p)->bw.len; }
static uint64_t mp_get_l(const void *p, uint64_t i) { return orc_wm_get(&((const orc_multi *)p)->bw, i); }
static uint64_t mp_zero(const orc_multi *f, uint64_t i, uint64_t rank) { /* multi_pieces.rs:131-137 */
  if (i < f->sa_idx_first_text) return rank + 1;
  if (i == f->sa_idx_first_text) return 0;
  return rank;
}
static uint64_t mp_lf_map2(const void *p, uint64_t c, uint64_t i) {      /* multi_pieces.rs:145-157 */
  const orc_multi *f = (const orc_multi *)p;
  uint64_t rank = orc_wm_rank(&f->bw, i, c);
  return c == 0 ? mp_zero(f, i, rank) : rank + f->cs[c];
}
static uint64_t mp_lf_map(const void *p, uint64_t i) {                   /* multi_pieces.rs:128-143 */
  return mp_lf_map2(p, mp_get_l(p, i), i);
}
static uint64_t mp_get_f(const void *p, uint64_t i) {                    /* multi_pieces.rs:159-174 */
  const orc_multi *f = (const orc_multi *)p;
  return cs_upper(f->cs, f->max_character + 1, i);
}
static uint64_t mp_fl_map(const void *p, uint64_t i) {                   /* multi_pieces.rs:176-187 */
  const orc_multi *f = (const orc_multi *)p;
  uint64_t c = mp_get_f(p, i);
  if (c == 0) return UINT64_MAX; /* None */
  return orc_wm_select(&f->bw, i - f->cs[c], c);
}
static uint64_t mp_get_sa(const void *p, uint64_t i) {                   /* multi_pieces.rs:194-207 */
  const orc_multi *f = (const orc_multi *)p;
  uint64_t steps = 0, sa;
  for (;;) {
    if (orc_ssa_get(&f->ssa, i, &sa)) return (sa + steps) % f->bw.len;
    i = mp_lf_map(p, i);
    steps++;
  }
}
static uint64_t mp_piece_id(const void *p, uint64_t i) {                 /* multi_pieces.rs:213-224 */
  const orc_multi *f = (const orc_multi *)p;
  for (;;) {
    if (mp_get_l(p, i) == 0) {
      uint64_t prev = f->doc[orc_wm_rank(&f->bw, i, 0)];
      return (prev + 1) % f->doc_len; /* modular_add */
    }
    i = mp_lf_map(p, i);
  }
}
orc_backend orc_multi_backend(orc_multi *f) {
  orc_backend b = {f, mp_get_l, mp_lf_map, mp_lf_map2, mp_len, f->has_locate ? mp_get_sa : NULL,
                   mp_get_f, mp_fl_map, mp_piece_id, f->doc_len, f->max_character};
  return b;
}
int orc_multi_new(orc_multi **out, const uint8_t *text, uint64_t n, uint64_t max_character,
                  int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 255) return ORC_ERR_ARG;
  int rc = check_symbols(text, n, max_character);
  if (rc) return rc;
  rc = orc_validate_text(text, n);
  if (rc) return rc;
  orc_multi *f = (orc_multi *)calloc(1, sizeof(orc_multi));
  f->max_character = max_character;
  f->cs = (uint64_t *)calloc(max_character + 1, sizeof(uint64_t));
  orc_bucket_start(text, n, max_character, f->cs);           /* multi_pieces.rs:39 */
  uint32_t *sa = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_suffix_array(text, n, sa);                              /* multi_pieces.rs:40 */
  uint8_t *bwt = (uint8_t *)malloc(n ? n : 1);
  orc_bwt(text, n, sa, bwt);                                  /* multi_pieces.rs:87-103 */
  orc_wm_build(&f->bw, bwt, n, orc_max_bits(max_character));
  /* doc (multi_pieces.rs:57-85) */
  uint64_t count = 0;
  for (uint64_t i = 0; i < n; i++) count += text[i] == 0;
  f->doc = (uint64_t *)calloc(count ? count : 1, sizeof(uint64_t));
  f->doc_len = count;
  uint64_t k = 0;
  for (uint64_t p = 0; p < n; p++) { /* bw.select_u64(k, 0) enumerates the zeros of L in order */
    if (bwt[p] != 0) continue;
    uint64_t idx = sa[p] >= 1 ? sa[p] - 1 : n - 1;            /* modular_sub(sa[p], 1, n) */
    uint64_t piece = 0;
    for (uint64_t q = 0; q < idx; q++) piece += text[q] == 0; /* end_marker_flags.rank1(idx) */
    if (piece == count - 1) f->sa_idx_first_text = p;
    f->doc[k++] = piece;
  }
  if (level >= 0) {
    orc_ssa_sample(&f->ssa, sa, n, (uint64_t)level);
    f->has_locate = 1;
  }
  free(bwt);
  free(sa);
  *out = f;
  return ORC_OK;
}
void orc_multi_free(orc_multi *f) {
  if (!f) return;
  orc_wm_free(&f->bw);
  orc_ssa_free(&f->ssa);
  free(f->cs);
  free(f->doc);
  free(f);
}

/* ------------------------------------------------------------------ */
/* driver (wrapper.rs)                                                 */
/* ------------------------------------------------------------------ */
int orc_search(const orc_backend *b, const uint8_t *pat, uint64_t m, uint64_t *ps,
               uint64_t *pe, uint64_t *steps) {
  uint64_t s = *ps, e = *pe, k = 0;        /* wrapper.rs:105-106 */
  for (uint64_t j = m; j-- > 0;) {         /* wrapper.rs:108: pattern.iter().rev() */
    uint64_t c = pat[j];
    if (c > b->max_character) return ORC_ERR_SYMBOL_RANGE;
    s = b->lf_map2(b->self, c, s);         /* wrapper.rs:109 */
    e = b->lf_map2(b->self, c, e);         /* wrapper.rs:110 */
    k++;
    if (s == e) break;                     /* wrapper.rs:111-113 */
  }
  *ps = s; *pe = e;
  if (steps) *steps = k;
  return ORC_OK;
}
void orc_locate_range(const orc_backend *b, uint64_t s, uint64_t e, uint64_t *out) {
  /* wrapper.rs:203-217 (match_prefix_only == false) + wrapper.rs:238-242 */
  for (uint64_t i = s; i < e; i++) out[i - s] = b->get_sa(b->self, i);
}

int orc_count_batch(const orc_backend *b, const uint8_t *pat, const uint64_t *off,
                    uint64_t npat, const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e,
                    uint64_t *out_steps, int nthreads) {
  int err = 0;
  uint64_t n = b->len(b->self);
  if (nthreads < 1) nthreads = 1;
  orc_place pl;
  orc_place_begin(&pl);
#pragma omp parallel num_threads(nthreads)
  {
#ifdef _OPENMP
    orc_place_thread(&pl, omp_get_thread_num(), omp_get_num_threads());
#endif
#pragma omp for schedule(static)
    for (int64_t k = 0; k < (int64_t)npat; k++) {
      uint64_t s = s0e0 ? s0e0[2 * k] : 0, e = s0e0 ? s0e0[2 * k + 1] : n, st = 0; /* wrapper.rs:41 */
      int rc = orc_search(b, pat + off[k], off[k + 1] - off[k], &s, &e, &st);
      if (rc) {
#pragma omp atomic write
        err = rc;
      }
      out_s[k] = s; out_e[k] = e;
      if (out_steps) out_steps[k] = st;
    }
    /* EVERY thread of the team gets the caller's mask back, not only the calling thread: libgomp keeps its workers
     * between regions, and a worker left on one CPU would run later batches -- a smaller team of the thread sweep,
     * orc_locate_batch -- on that stale one-CPU mask */
    orc_place_end(&pl);
  }
  orc_place_end(&pl);
  return err;
}
void orc_locate_batch(const orc_backend *b, const uint64_t *s, const uint64_t *e,
                      uint64_t npat, const uint64_t *out_off, uint64_t *out_pos,
                      int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
  for (int64_t k = 0; k < (int64_t)npat; k++)
    if (e[k] > s[k]) orc_locate_range(b, s[k], e[k], out_pos + out_off[k]);
}
void orc_lf_map2_batch(const orc_backend *b, const uint64_t *c, const uint64_t *i,
                       uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->lf_map2(b->self, c[j], i[j]);
}
void orc_lf_map_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->lf_map(b->self, i[j]);
}
void orc_get_l_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->get_l(b->self, i[j]);
}
void orc_get_sa_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->get_sa(b->self, i[j]);
}
uint64_t orc_match_rows(const orc_backend *b, uint64_t s, uint64_t e, int match_prefix_only,
                        uint64_t *out, uint64_t cap) {            /* wrapper.rs:203-217 */
  uint64_t cnt = 0;
  for (uint64_t i = s; i < e; i++) {
    if (!match_prefix_only || b->get_l(b->self, i) == 0) {
      if (cnt < cap) out[cnt] = i;
      cnt++;
    }
  }
  return cnt;
}
void orc_piece_id_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->piece_id(b->self, i[j]);
}
void orc_get_f_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->get_f(b->self, i[j]);
}
void orc_fl_map_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out) {
  for (uint64_t j = 0; j < k; j++) out[j] = b->fl_map(b->self, i[j]);
}

/* tests/testutil/mod.rs:62-86 with match_prefix_only = match_suffix_only = false */
uint64_t orc_naive_search(const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m,
                          uint64_t *out, uint64_t cap) {
  uint64_t cnt = 0;
  if (m > n) return 0;
  for (uint64_t i = 0; i + m <= n; i++) {
    if (memcmp(text + i, pat, m) == 0) {
      if (cnt < cap) out[cnt] = i;
      cnt++;
    }
  }
  return cnt;
}

/* ------------------------------------------------------------------ */
/* wide symbols (u16 / u32 texts as uint32_t arrays)                    */
/* ------------------------------------------------------------------ */
int orc_validate_text_w(const uint32_t *text, uint64_t n) { /* sais.rs:115-139 */
  if (n == 0 || n == 1) return ORC_OK;
  if (text[0] == 0) return ORC_ERR_START_ZERO;
  int64_t last = -1;
  for (int64_t i = (int64_t)n - 1; i >= 0; i--)
    if (text[i] != 0) { last = i; break; }
  if (last != (int64_t)n - 2) return ORC_ERR_END_ZERO;
  return ORC_OK;
}
typedef struct { const uint32_t *t; uint64_t n; } naive_ctx_w;
static int naive_cmp_w(const void *a, const void *b, void *c) {
  const naive_ctx_w *x = (const naive_ctx_w *)c;
  uint64_t i = *(const uint32_t *)a, j = *(const uint32_t *)b;
  while (i < x->n && j < x->n) {
    if (x->t[i] != x->t[j]) return x->t[i] < x->t[j] ? -1 : 1;
    i++; j++;
  }
  return i >= x->n ? (j >= x->n ? 0 : -1) : 1; /* the suffix that ends first is smaller */
}
void orc_suffix_array_naive_w(const uint32_t *text, uint64_t n, uint32_t *sa) {
  for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
  naive_ctx_w c = {text, n};
  qsort_r(sa, n, sizeof(uint32_t), naive_cmp_w, &c);
}
static int sym_cmp_w(const void *a, const void *b, void *c) {
  const uint32_t *t = (const uint32_t *)c;
  uint32_t x = t[*(const uint32_t *)a], y = t[*(const uint32_t *)b];
  return x < y ? -1 : (x > y ? 1 : 0);
}
void orc_suffix_array_w(const uint32_t *text, uint64_t n, uint32_t *sa) {
  if (n == 0) return;
  uint32_t *rank = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint32_t *nrank = (uint32_t *)malloc(n * sizeof(uint32_t));
  for (uint64_t i = 0; i < n; i++) sa[i] = (uint32_t)i;
  qsort_r(sa, n, sizeof(uint32_t), sym_cmp_w, (void *)text);
  uint64_t head = 0;
  for (uint64_t p = 0; p < n; p++) {
    if (p > 0 && text[sa[p]] != text[sa[p - 1]]) head = p;
    rank[sa[p]] = (uint32_t)head;
  }
  for (uint64_t h = 1;; h *= 2) {
    dbl_ctx ctx = {rank, h, n};
    int any = 0;
    uint64_t g0 = 0;
    while (g0 < n) {
      uint64_t g1 = g0 + 1;
      uint32_t r = rank[sa[g0]];
      while (g1 < n && rank[sa[g1]] == r) g1++;
      if (g1 - g0 > 1) {
        any = 1;
        qsort_r(sa + g0, g1 - g0, sizeof(uint32_t), dbl_cmp, &ctx);
        uint64_t hd = g0;
        for (uint64_t p = g0; p < g1; p++) {
          if (p > g0 && dbl_cmp(&sa[p - 1], &sa[p], &ctx) != 0) hd = p;
          nrank[sa[p]] = (uint32_t)hd;
        }
      } else {
        nrank[sa[g0]] = (uint32_t)g0;
      }
      g0 = g1;
    }
    memcpy(rank, nrank, n * sizeof(uint32_t));
    if (!any) break;
  }
  free(rank);
  free(nrank);
}
static int check_symbols_w(const uint32_t *text, uint64_t n, uint64_t max_character) {
  for (uint64_t i = 0; i < n; i++)
    if (text[i] > max_character) return ORC_ERR_SYMBOL_RANGE;
  return ORC_OK;
}
static void bucket_start_w(const uint32_t *text, uint64_t n, uint64_t max_character, uint64_t *cs) {
  uint64_t m = max_character + 1;
  uint64_t *occ = (uint64_t *)calloc(m, sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) occ[text[i]]++;
  uint64_t sum = 0;
  for (uint64_t c = 0; c < m; c++) { cs[c] = sum; sum += occ[c]; }
  free(occ);
}
int orc_fm_new_w(orc_fm **out, const uint32_t *text, uint64_t n, uint64_t max_character, int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 0xFFFFFFFFull) return ORC_ERR_ARG;
  int rc = check_symbols_w(text, n, max_character);
  if (rc) return rc;
  rc = orc_validate_text_w(text, n);
  if (rc) return rc;
  orc_fm *f = (orc_fm *)calloc(1, sizeof(orc_fm));
  f->max_character = max_character;
  f->cs = (uint64_t *)calloc(max_character + 1, sizeof(uint64_t));
  bucket_start_w(text, n, max_character, f->cs);             /* fm_index.rs:32 */
  uint32_t *sa = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_suffix_array_w(text, n, sa);                           /* fm_index.rs:33 */
  uint32_t *bwt = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  for (uint64_t i = 0; i < n; i++) bwt[i] = sa[i] > 0 ? text[sa[i] - 1] : 0; /* fm_index.rs:50-55 */
  orc_wm_build_w(&f->bw, bwt, n, orc_max_bits(max_character));
  if (level >= 0) {
    orc_ssa_sample(&f->ssa, sa, n, (uint64_t)level);
    f->has_locate = 1;
  }
  free(bwt);
  free(sa);
  *out = f;
  return ORC_OK;
}
int orc_rlfm_new_w(orc_rlfm **out, const uint32_t *text, uint64_t n, uint64_t max_character,
                   int level) {
  *out = NULL;
  if (max_character == 0 || max_character > 0xFFFFFFFFull) return ORC_ERR_ARG;
  int rc = check_symbols_w(text, n, max_character);
  if (rc) return rc;
  rc = orc_validate_text_w(text, n);
  if (rc) return rc;
  orc_rlfm *f = (orc_rlfm *)calloc(1, sizeof(orc_rlfm));
  uint64_t m = max_character + 1;
  f->len = n;
  f->max_character = max_character;
  uint32_t *sa = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  orc_suffix_array_w(text, n, sa);
  uint64_t nw = (n + 63) / 64 + BLK_WORDS;
  uint64_t *bw = (uint64_t *)calloc(nw, 8), *bpw = (uint64_t *)calloc(nw, 8);
  uint32_t *heads = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  uint64_t *run_len = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint64_t *runs_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  uint64_t r = 0, c0 = 0;
  for (uint64_t i = 0; i < n; i++) {                            /* rlfmi.rs:48-68 */
    uint32_t k = sa[i];
    uint64_t c = k > 0 ? text[k - 1] : text[n - 1];
    if (c0 != c) {
      heads[r] = (uint32_t)c; run_len[r] = 1; r++;
      bits_set(bw, i);
      runs_of[c]++;
    } else {
      run_len[r - 1]++;
    }
    c0 = c;
  }
  f->runs = r;
  orc_wm_build_w(&f->s, heads, r, orc_max_bits(max_character));
  f->cs = (uint64_t *)calloc(m, sizeof(uint64_t));
  uint64_t acc = 0;
  for (uint64_t c = 0; c < m; c++) { f->cs[c] = acc; acc += runs_of[c]; }
  uint64_t *start_of = (uint64_t *)calloc(m, sizeof(uint64_t));
  {
    uint64_t *chars_of = (uint64_t *)calloc(m, sizeof(uint64_t));
    for (uint64_t k = 0; k < r; k++) chars_of[heads[k]] += run_len[k];
    uint64_t a = 0;
    for (uint64_t c = 0; c < m; c++) { start_of[c] = a; a += chars_of[c]; }
    free(chars_of);
  }
  for (uint64_t k = 0; k < r; k++) {
    uint64_t c = heads[k];
    bits_set(bpw, start_of[c]);
    start_of[c] += run_len[k];
  }
  orc_rsvec_build(&f->b, bw, n);
  orc_rsvec_build(&f->bp, bpw, n);
  if (level >= 0) {
    orc_ssa_sample(&f->ssa, sa, n, (uint64_t)level);
    f->has_locate = 1;
  }
  free(start_of); free(runs_of); free(run_len); free(heads); free(sa);
  *out = f;
  return ORC_OK;
}
static int orc_search_w(const orc_backend *b, const uint32_t *pat, uint64_t m, uint64_t *ps,
                        uint64_t *pe, uint64_t *steps) {       /* wrapper.rs:103-124 */
  uint64_t s = *ps, e = *pe, k = 0;
  for (uint64_t j = m; j-- > 0;) {
    uint64_t c = pat[j];
    if (c > b->max_character) return ORC_ERR_SYMBOL_RANGE;
    s = b->lf_map2(b->self, c, s);
    e = b->lf_map2(b->self, c, e);
    k++;
    if (s == e) break;
  }
  *ps = s; *pe = e;
  if (steps) *steps = k;
  return ORC_OK;
}
int orc_count_batch_w(const orc_backend *b, const uint32_t *pat, const uint64_t *off,
                      uint64_t npat, const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e,
                      uint64_t *out_steps, int nthreads) {
  int err = 0;
  uint64_t n = b->len(b->self);
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t k = 0; k < (int64_t)npat; k++) {
    uint64_t s = s0e0 ? s0e0[2 * k] : 0, e = s0e0 ? s0e0[2 * k + 1] : n, st = 0;
    int rc = orc_search_w(b, pat + off[k], off[k + 1] - off[k], &s, &e, &st);
    if (rc) {
#pragma omp atomic write
      err = rc;
    }
    out_s[k] = s; out_e[k] = e;
    if (out_steps) out_steps[k] = st;
  }
  return err;
}
uint64_t orc_naive_search_w(const uint32_t *text, uint64_t n, const uint32_t *pat, uint64_t m,
                            uint64_t *out, uint64_t cap) {      /* tests/testutil/mod.rs:62-86 */
  uint64_t cnt = 0;
  if (m > n) return 0;
  for (uint64_t i = 0; i + m <= n; i++) {
    if (memcmp(text + i, pat, m * sizeof(uint32_t)) == 0) {
      if (cnt < cap) out[cnt] = i;
      cnt++;
    }
  }
  return cnt;
}
