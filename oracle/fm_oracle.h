/*
 * fm_oracle.h -- CPU ORACLE for the count/locate hot path of ajalab/fm-index.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (Rust crate fm-index v0.3.1 + the behaviour of its un-vendored
 * dependency vers-vecs ^1.10.1).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it -- as the checker / the CPU row,
 * never as the thing shipped.  The product (libfmx.so, HIP) never links,
 * imports or calls anything in this directory.
 *
 * Parity pin: the reference cannot be compiled here (no Rust toolchain, and
 * vers-vecs' sources are absent).  The oracle is pinned against every
 * known-answer vector the reference's own tests hold for this path
 * (tests/golden/reference_known_answers.json) and against
 * the brute-force property the reference's integration tests use
 * (tests/testutil/mod.rs:62-86).  See tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (src/error.rs:3-6 has one variant, InvalidText(msg)) ---- */
#define ORC_OK 0
#define ORC_ERR_START_ZERO 1    /* sais.rs:128-132 */
#define ORC_ERR_END_ZERO 2      /* sais.rs:133-138 */
#define ORC_ERR_SYMBOL_RANGE 3  /* reference panics on cs[c] (fm_index.rs:94) */
#define ORC_ERR_ARG 4

const char *orc_error_message(int code);

/* ---- text parameters (text.rs:61-63, util.rs:1-3) ---- */
uint32_t orc_log2(uint64_t x);                 /* util.rs:1-3  floor(log2 x) */
uint32_t orc_max_bits(uint64_t max_character); /* text.rs:61-63             */

/* sais.rs:115-139: validation performed by build_suffix_array */
int orc_validate_text(const uint8_t *text, uint64_t n);

/* sais.rs:546-557 (naive definition of the suffix array; O(n^2 log n)) */
void orc_suffix_array_naive(const uint8_t *text, uint64_t n, uint32_t *sa);
/* same array by prefix doubling (O(n log^2 n)); any correct algorithm yields
 * the identical array because the terminator makes all suffixes distinct */
void orc_suffix_array(const uint8_t *text, uint64_t n, uint32_t *sa);

/* sais.rs:9-32: cs[c] = #{j : T[j] < c}, c in 0..=max_character */
void orc_bucket_start(const uint8_t *text, uint64_t n, uint64_t max_character,
                      uint64_t *cs);
/* fm_index.rs:44-58: bw[i] = T[sa[i]-1], or 0 when sa[i]==0 */
void orc_bwt(const uint8_t *text, uint64_t n, const uint32_t *sa, uint8_t *bwt);

/* ---- vers-vecs behaviour restated ---- */
typedef struct orc_rsvec {
  uint64_t len;        /* bits */
  uint64_t nwords;
  uint64_t *words;     /* LSB-first */
  uint16_t *blk;       /* ones before 512-bit block, inside its superblock */
  uint64_t *sup;       /* ones before 8192-bit superblock */
  uint64_t ones;
} orc_rsvec;

void orc_rsvec_build(orc_rsvec *v, uint64_t *words_take, uint64_t len);
void orc_rsvec_free(orc_rsvec *v);
uint64_t orc_rsvec_rank1(const orc_rsvec *v, uint64_t i); /* clamped: i>=len -> ones */
uint64_t orc_rsvec_rank0(const orc_rsvec *v, uint64_t i);
uint64_t orc_rsvec_select1(const orc_rsvec *v, uint64_t k); /* k>=ones -> len */
int orc_rsvec_get(const orc_rsvec *v, uint64_t i);

typedef struct orc_wm {
  uint32_t bits;
  uint64_t len;
  orc_rsvec *lv;     /* bits levels, MSB first */
  uint64_t *zeros;   /* zeros per level */
} orc_wm;

void orc_wm_build(orc_wm *w, const uint8_t *vals, uint64_t n, uint32_t bits);
void orc_wm_free(orc_wm *w);
uint64_t orc_wm_get(const orc_wm *w, uint64_t i);               /* get_u64_unchecked  */
uint64_t orc_wm_rank(const orc_wm *w, uint64_t i, uint64_t c);  /* rank_u64_unchecked */
uint64_t orc_wm_select(const orc_wm *w, uint64_t k, uint64_t c); /* select_u64_unchecked */

/* ---- SOSampledSuffixArray (suffix_array/sample.rs:21-60) ---- */
typedef struct orc_ssa {
  uint64_t level, word_size, len, nsamples;
  uint64_t *bits; /* BitVec: LSB-first packed words of word_size bits */
} orc_ssa;
void orc_ssa_sample(orc_ssa *s, const uint32_t *sa, uint64_t n, uint64_t level);
/* sample.rs:21-44 fed from an already subsampled array (samples[k] = sa[k<<level]) */
void orc_ssa_from_samples(orc_ssa *s, const uint32_t *samples, uint64_t n, uint64_t level);
void orc_ssa_from_samples64(orc_ssa *s, const uint64_t *samples, uint64_t n, uint64_t level);
void orc_ssa_free(orc_ssa *s);
/* returns 1 and *out when Some, 0 when None (sample.rs:46-60) */
int orc_ssa_get(const orc_ssa *s, uint64_t i, uint64_t *out);

/* ---- backend trait (backend.rs:5-31) as a vtable ---- */
typedef struct orc_backend {
  void *self;
  uint64_t (*get_l)(const void *self, uint64_t i);
  uint64_t (*lf_map)(const void *self, uint64_t i);
  uint64_t (*lf_map2)(const void *self, uint64_t c, uint64_t i);
  uint64_t (*len)(const void *self);
  uint64_t (*get_sa)(const void *self, uint64_t i); /* HasPosition; may be NULL */
  uint64_t (*get_f)(const void *self, uint64_t i);
  uint64_t (*fl_map)(const void *self, uint64_t i); /* UINT64_MAX = None (multi-pieces, c == 0) */
  uint64_t (*piece_id)(const void *self, uint64_t i); /* HasMultiPieces; NULL otherwise */
  uint64_t pieces_count;
  uint64_t max_character;
} orc_backend;

/* ---- FMIndexBackend (fm_index.rs) ---- */
typedef struct orc_fm {
  orc_wm bw;
  uint64_t *cs;
  uint64_t max_character;
  orc_ssa ssa;      /* len==0 when built without locate */
  int has_locate;
} orc_fm;

/* FMIndex::new / FMIndexWithLocate::new (frontend.rs:195-243, fm_index.rs:25-42).
 * level < 0 => count-only. */
int orc_fm_new(orc_fm **out, const uint8_t *text, uint64_t n, uint64_t max_character,
               int level);
/* same structure built from a ready BWT + C-array (+ optional subsampled SA);
 * used only to time the CPU baseline at sizes where the oracle's own suffix
 * sorter would take too long */
int orc_fm_from_bwt(orc_fm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                    const uint64_t *cs, const uint32_t *samples, int level);
int orc_fm_from_bwt64(orc_fm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                    const uint64_t *cs, const uint64_t *samples, int level);
void orc_fm_free(orc_fm *f);
orc_backend orc_fm_backend(orc_fm *f);
uint64_t orc_fm_heap_bytes(const orc_fm *f);

/* ---- RLFMIndexBackend (rlfmi.rs) ---- */
typedef struct orc_rlfm {
  orc_wm s;
  orc_rsvec b, bp;
  uint64_t *cs;
  uint64_t len, max_character, runs;
  orc_ssa ssa;
  int has_locate;
} orc_rlfm;
int orc_rlfm_new(orc_rlfm **out, const uint8_t *text, uint64_t n, uint64_t max_character,
                 int level);
int orc_rlfm_from_bwt(orc_rlfm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                      const uint32_t *samples, int level);
int orc_rlfm_from_bwt64(orc_rlfm **out, const uint8_t *bwt, uint64_t n, uint64_t max_character,
                        const uint64_t *samples, int level);   /* n >= 2^32: 64-bit sample values */
void orc_rlfm_free(orc_rlfm *f);
orc_backend orc_rlfm_backend(orc_rlfm *f);

/* ---- FMIndexMultiPiecesBackend (multi_pieces.rs) ---- */
typedef struct orc_multi {
  orc_wm bw;
  uint64_t *cs;
  uint64_t max_character;
  orc_ssa ssa;
  int has_locate;
  uint64_t *doc;            /* multi_pieces.rs:20 */
  uint64_t doc_len;
  uint64_t sa_idx_first_text; /* multi_pieces.rs:22 */
} orc_multi;
int orc_multi_new(orc_multi **out, const uint8_t *text, uint64_t n, uint64_t max_character,
                  int level);
void orc_multi_free(orc_multi *f);
orc_backend orc_multi_backend(orc_multi *f);

/* ---- driver (wrapper.rs) ---- */
/* SearchWrapper::search, wrapper.rs:103-124; (s,e) in/out. returns ORC_ERR_SYMBOL_RANGE
 * where the reference would panic on cs[c]. *steps (nullable) = executed iterations */
int orc_search(const orc_backend *b, const uint8_t *pat, uint64_t m, uint64_t *s,
               uint64_t *e, uint64_t *steps);
/* iter_matches().map(locate): wrapper.rs:203-217, 238-242. out has e-s slots. */
void orc_locate_range(const orc_backend *b, uint64_t s, uint64_t e, uint64_t *out);

/* batch drivers (OpenMP over patterns); s0e0 nullable => (0,len) */
int orc_count_batch(const orc_backend *b, const uint8_t *pat, const uint64_t *pat_off,
                    uint64_t npat, const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e,
                    uint64_t *out_steps, int nthreads);
void orc_locate_batch(const orc_backend *b, const uint64_t *s, const uint64_t *e,
                      uint64_t npat, const uint64_t *out_off, uint64_t *out_pos,
                      int nthreads);
/* scalar batch probes of the trait methods (for parity of every (c,i)) */
void orc_lf_map2_batch(const orc_backend *b, const uint64_t *c, const uint64_t *i,
                       uint64_t k, uint64_t *out);
void orc_lf_map_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);
void orc_get_l_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);
void orc_get_sa_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);
/* MatchIteratorWrapper::next (wrapper.rs:203-217): rows of [s,e) it yields; returns their number */
uint64_t orc_match_rows(const orc_backend *b, uint64_t s, uint64_t e, int match_prefix_only,
                        uint64_t *out, uint64_t cap);
void orc_piece_id_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);
void orc_get_f_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);
void orc_fl_map_batch(const orc_backend *b, const uint64_t *i, uint64_t k, uint64_t *out);

/* NaiveSearchIndex::search (tests/testutil/mod.rs:62-86): positions ascending;
 * returns the number of matches, writes at most cap of them */
uint64_t orc_naive_search(const uint8_t *text, uint64_t n, const uint8_t *pat, uint64_t m,
                          uint64_t *out, uint64_t cap);

int orc_max_threads(void);
int orc_team_size(int nthreads);
void orc_set_thread_spread(int on);

/* ---- wide symbols: Character = u16 / u32 (character.rs:38-42) as uint32_t arrays ------ */
/* Same algorithms as above with 32-bit symbols; max_character < 2^32, bits = max_bits. */
int orc_validate_text_w(const uint32_t *text, uint64_t n);
void orc_suffix_array_naive_w(const uint32_t *text, uint64_t n, uint32_t *sa);
void orc_suffix_array_w(const uint32_t *text, uint64_t n, uint32_t *sa);
void orc_wm_build_w(orc_wm *w, const uint32_t *vals, uint64_t n, uint32_t bits);
int orc_fm_new_w(orc_fm **out, const uint32_t *text, uint64_t n, uint64_t max_character, int level);
int orc_rlfm_new_w(orc_rlfm **out, const uint32_t *text, uint64_t n, uint64_t max_character,
                   int level);
int orc_count_batch_w(const orc_backend *b, const uint32_t *pat, const uint64_t *pat_off,
                      uint64_t npat, const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e,
                      uint64_t *out_steps, int nthreads);
uint64_t orc_naive_search_w(const uint32_t *text, uint64_t n, const uint32_t *pat, uint64_t m,
                            uint64_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
