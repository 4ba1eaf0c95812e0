// fmx_internal.h -- structures shared by the builder, the query kernels and the C ABI.
//
// HBM layout (see DESIGN.md "Data layout"):
//   The BWT (fm_index.rs:44-58) is stored as a *multi-ary wavelet matrix*: the
//   L = max_bits symbol bits (text.rs:61-63) are split MSB-first into levels of 3 or
//   4 bits.  Every level is an array of 128-byte records; one record holds the bit
//   planes of 256 (3-bit level) or 128 (4-bit level) consecutive entries TOGETHER
//   with the per-code rank counters for the record start -- "level-interleaved"
//   planes + counters in one cache line, so one aligned 128-B load answers one rank.
//   A record is 8 pieces of 16 B; an 8-lane group loads one record with a single
//   dwordx4 per lane and every lane's piece is self-contained:
//     fmt 3:  piece g = { cnt[g], plane0, plane1, plane2 }   of entries [32g, 32g+32)
//     fmt 4:  piece g = { cnt[2g], cnt[2g+1], plane0|plane1<<16, plane2|plane3<<16 }
//                                                           of entries [16g, 16g+16)
//   Between levels the whole sequence is stably sorted by the level's code (the
//   wavelet-matrix trick generalised to 8-/16-ary), so position p maps to
//   C[code] + rank_code(p) on the next level.  The counters are stored ABSOLUTE:
//   a non-last level's cnt[code] already includes C[code] (so cnt + popcount IS the next
//   position), and when there is a single level cnt[c] includes cs[c] (sais.rs:9-32), so
//   cnt + popcount IS lf_map2(c, i) (fm_index.rs:93-95) and K[] is all zero.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstddef>
#include "fmx.h"

#define FMX_MAX_LEVELS 8
#define FMX_SERIES_CAP 64

struct FmxLevel {
  const uint4 *rec;   // nrec records x 8 pieces
  const uint32_t *C;  // 16 entries: #entries of this level with a smaller code
  uint32_t fmt;       // 3 or 4
  uint32_t shift;     // code = (sym >> shift) & mask
  uint32_t mask;
  uint32_t nrec;
  // select hints (fl_map / iter_chars_forward): for code c, selmeta[c] = its counter at record 0,
  // selmeta[16 + c] = start of its hints in sel[], selmeta[32 + c] = its number of entries;
  // sel[start + j] = record holding the code's (j * FMX_WSEL_STEP)-th entry.  NULL = none built.
  const uint32_t *sel;
  const uint32_t *selmeta;
};
#define FMX_WSEL_STEP 128u

struct FmxMwm {
  FmxLevel lv[FMX_MAX_LEVELS];
  uint32_t nlevels;
  uint32_t bits;
  uint32_t len;  // entries
  uint32_t pad;
};

// bit vector with rank/select support in 128-B records (RLFM: B and B', rlfmi.rs:19-20):
//   piece g = { ones before this PIECE (absolute), 96 payload bits }  -> 768 bits / record
struct FmxBits {
  const uint4 *rec;
  const uint32_t *sel;  // select hints: record index holding the (k*FMX_SEL_STEP)-th one
  const uint4 *dsel;    // vectors with >= 0.11 ones per bit, else NULL: one 16-byte SELECT BLOCK per
                        // 2^dsel_shift ones = { position of the block's first one, the 96 bits from there },
                        // so that select1 is ONE load (x == 0xFFFFFFFF: the block's ones do not fit the 96
                        // bits, use the records).  Ones per block by density, so that a block almost always
                        // fits: >= 0.87 -> 64, >= 0.44 -> 32, >= 0.22 -> 16, >= 0.11 -> 8
  const uint32_t *pos;  // sparser vectors only (< 0.11 ones per bit, i.e. runs of 9+ on average), else NULL:
                        // position of every one, so that select1 is ONE load (4 bytes per one)
  uint32_t nrec;
  uint32_t len;
  uint32_t ones;
  uint32_t nsel;
  uint32_t dsel_shift;  // log2(ones per select block): 6, 5, 4 or 3 (0 when dsel is NULL)
  uint32_t pad;
};
// one hint per 64 ones: on sparse vectors (long runs) the record search behind a hint is a chain of
// dependent loads; 512 -> 64 cut the repetitive-text count by 13 % and its locate by 16 % for
// ones/16 bytes of hints
#define FMX_SEL_STEP 64u
#define FMX_BITS_PER_REC 768u
#define FMX_BITS_PER_PIECE 96u

struct FmxDev {  // passed BY VALUE to every query kernel
  FmxMwm bw;            // FM: BWT.  RLFM: run heads S (rlfmi.rs:17)
  const uint32_t *K;    // K[c] = cs[c] - S_c (wrapping u32), c in 0..=max_character
  const uint32_t *samples;  // SOSampledSuffixArray payload as plain u32 (sample.rs:21-44)
  uint32_t *status;     // sticky device-side error bits
  uint32_t n;           // len incl. terminator
  uint32_t max_character;
  uint32_t sa_level;    // effective level; FMX_NO_LOCATE when absent
  uint32_t kind;
  uint32_t sym_bytes;   // width of text / pattern symbols (1, 2 or 4)
  uint32_t nsamples;    // entries of samples[] (bounds checks of the debug build)
  FmxBits b, bp;        // RLFM only
  const uint32_t *doc;    // MULTI: piece id of the k-th end marker in L order (multi_pieces.rs:57-85)
  uint32_t doc_count;     // MULTI: number of pieces
  uint32_t first_row;     // MULTI: sa_idx_first_text (multi_pieces.rs:21-22)
  const uint4 *pair_rec;  // FMX_FLAG_PAIR_INDEX: fmt-4 records over the 2-gram BWT, absolute counters
  uint32_t pair_row0, pair_row1;  // the two rows (SA = 0, 1) that have no 2-gram; stored as code 0
  const uint32_t *cs;   // C array on the device for get_f / fl_map: characters (FM, sais.rs:9-32)
                        // or runs (RLFM, rlfmi.rs:72-76)
  const uint2 *kmer;    // FMX_FLAG_KMER_TABLE: (s, e) of searching each k-mer from (0, n)
  uint32_t kmer_k;      // symbols per entry (0 = no table)
  uint32_t kmer_bits;   // bits per symbol in the table index (symbol c is coded c - 1)
  // Text-order sampling (levels 1..4; NULL = the rows i with i mod 2^level == 0 are the sampled ones,
  // as in sample.rs).  get_sa(i) is SA[i] whatever is sampled, so the structure is ours to choose:
  // sampling the rows whose SA VALUE is a multiple of 2^level makes every walk exactly
  // SA[i] mod 2^level steps long -- (2^level - 1) / 2 on average and never more than 2^level - 1,
  // against 2^level - 1 on average and a geometric tail (48 steps in 2^20 hits) for sampled rows.
  //   phase[] = 16-byte pieces { phase-0 rows before this piece (absolute), 3 words of level-bit
  //             phases SA[row] mod 2^level, floor(32 / level) per word } -> 96, 48, 30, 24 rows per piece;
  //   samples[] then holds SA[row] of the phase-0 rows in row order (same ((n-1) >> level) + 1 entries).
  const uint4 *phase;
  // Walk records (round 4; one 3-bit level, max_character <= 5, text-order sampling at levels 1..3): a second
  // encoding of the BWT for the batched locate walk, DERIVED from bw.lv[0].rec and phase[] (never stored in a file).
  // One 128-byte record = 112 rows: pieces 0..6 hold 16 rows each, piece 7 holds counters only.  Piece g <= 6 = { x, y, z, w }:
  //     y = code plane 0 | code plane 1 << 16,  z = code plane 2 | phase plane 0 << 16,  w = phase plane 1 | plane 2 << 16
  //     x = g <= 4: lf_map2(g + 1, first row of the record)   (cs[] folded in, like the fmt-3 counters)
  //         g == 5: rank0 = number of phase-0 rows before the record
  //         g == 6: rank1[1];   piece 7 = { rank1[2], rank1[3], rank1[4], rank1[5] }
  //     rank1[c] = (row 0 is a phase-0 row) + (phase-1 rows of the index whose BWT symbol is < c)
  //              + (phase-1 rows before the record whose BWT symbol is c)
  // so ONE line answers everything an LF step of a text-order walk asks of a row: L[row], lf_map(row), the row's phase
  // SA[row] mod 2^level (= the steps left, known from the walk's first record on), the index into samples[] of a
  // phase-0 row (rank0 + the phase-0 rows before it in the record) -- and, for a phase-1 row r with symbol c, the index
  // of the sample of the row AFTER it, LF(r): LF keeps the order of rows with the same symbol and sends the phase-1
  // rows onto the phase-0 rows, so rank0(LF(r)) = rank1[c] + (phase-1 rows with symbol c before r in the record).  The
  // walk never reads the record of its final row, and the two 16-byte phase probes of the round-3 text-order walk are
  // gone: a hit costs max(phase, 1) records + 1 sample = 2.75 requests at level 2, where the row-order walk issues 4
  // (+ a geometric tail) and the round-3 text-order walk 4.5.  (Symbol 0 needs no counter: the one row that has it
  // -- SA = 0 -- maps to row 0, and no walk steps from it: its phase is 0.)
  const uint4 *walk;
  // RLFM with locate (round 4): lfrun[j] = lf_map(first row of run j) = the F position of run j (rlfmi.rs:127-133 at a
  // run start).  Rows of a run map to consecutive rows, so lf_map(i) = lfrun[run of i] + (i - start of that run): the
  // batched locate walk takes an LF step with TWO lane-wise requests -- the B piece of row i (run index, run start) and
  // this entry -- instead of B piece + one S record per wavelet level + B' select (4.25 requests on a byte alphabet).
  // 4 bytes per run (a repetitive text has few); built from the builder's F order, stored in index files.
  const uint32_t *lfrun;
};
#define FMX_PHASE_MAX_LEVEL 4u
#define FMX_WALK_MAX_LEVEL 3u        // three phase planes fit the piece
#define FMX_WALK_MAX_CHARACTER 5u    // 5 lf_map2 counters + rank0 + 5 rank1 counters = the 11 counter words of a record
#define FMX_WALK_ROWS 112u

// ---- wide indexes: n >= 2^32 - 16 (usize rows of the reference, fm_index.rs:86-95) ------------------------
// FMIndex / FMIndexWithLocate over byte texts, rows and positions 64 bits wide.  One-level alphabets
// (max_character <= 7: DNA): the same 128-byte fmt-3 records, but a record's eight counters are RELATIVE to the start of its superblock
// (2^31 entries = 2^23 records) and a small table holds the 64-bit absolute value at every superblock start:
//     lf_map2(c, i) = base[i >> 31][c] + cnt32[record(i)][c] + popcount          (cs[] folded into base)
// so a rank still costs one 128-byte line (+ 8 bytes of a table that lives in the caches), the in-group sums
// stay 32 bits wide, and the u32 engine is untouched.  Samples are u64.
#define FMX_WIDE_SB_SHIFT 31u      // log2(rows per superblock); FMX_FLAG_FORCE_WIDE (tests) uses FMX_WIDE_SB_SHIFT_TEST so
#define FMX_WIDE_SB_SHIFT_TEST 12u // that a small text has many superblocks
// Larger byte alphabets (8 <= max_character <= 255, `generic`): the multi-ary wavelet matrix of the 32-bit engine --
// up to FMXW_MAX_LEVELS levels of 3 or 4 bits, the sequence stably sorted by the level's code between levels -- with
// the same superblock scheme per level: lv[l].base[sb][code] is the 64-bit rank of `code` at the superblock's start,
// plus, on every level but the last, the number of entries with a smaller code (so that a level's rank IS the
// position in the next level, as in FmxLevel); K[c] = cs[c] - (rank chain of c at position 0), 64 bits wide:
//     lf_map2(c, i) = K[c] + rank chain of c at i                                  (fm_index.rs:93-95)
#define FMXW_MAX_LEVELS 7
struct FmxWideLevel {
  const uint4 *rec;          // n / per_rec + 1 records (per_rec = 256 for fmt 3, 128 for fmt 4)
  const uint64_t *base;      // [nsb][16]
  uint32_t fmt;              // 3 or 4
  uint32_t shift;            // code = (sym >> shift) & mask
  uint32_t mask;
  uint32_t nrec;
};
// bit vector of a wide RLFM index (B and B', rlfmi.rs:19-20): the 128-byte records of FmxBits -- 8 pieces of { ones before
// the piece, 96 payload bits } -- with the count RELATIVE to the record's superblock (2^sb_shift records: at most
// 2^22 x 768 bits < 2^32 ones) and base[superblock] = the 64-bit count at its start.  select1: the stored positions of
// the ones (sparse vectors: fewer than one bit in nine set, as on the 32-bit engine; 8 bytes per one), else a hint per
// FMX_SEL_STEP ones (the record that holds it) + a search over the record counts between two hints.
struct FmxWideBits {
  const uint4 *rec;
  const uint64_t *base;  // [nsb]
  const uint64_t *pos;   // NULL: hints + records
  const uint32_t *sel;   // [nsel] record indices
  uint64_t len, ones, nsel;
  uint32_t nrec, sb_shift, nsb, pad;
};
#define FMXW_BITS_SB_SHIFT 22u      // records per superblock
#define FMXW_BITS_SB_SHIFT_TEST 1u  // FMX_FLAG_FORCE_WIDE
#define FMXW_PHASE_SB_SHIFT 24u     // phase pieces per superblock (<= 96 rows each: < 2^31 rows)
#define FMXW_PHASE_SB_SHIFT_TEST 3u // FMX_FLAG_FORCE_WIDE
struct FmxWideDev {  // passed BY VALUE to the wide kernels
  const uint4 *rec;          // one 3-bit level (max_character <= 7): n / 256 + 1 records (row n is addressable)
  const uint64_t *base;      // ... [nsb][8], cs[] folded in
  const uint64_t *samples;   // SA[k << level], k = 0 .. (n - 1) >> level       (sample.rs:33-37)
  uint32_t *status;          // sticky device-side error bits
  uint64_t n;                // len incl. terminator
  uint32_t max_character;
  uint32_t sa_level;         // effective level; FMX_NO_LOCATE when absent
  uint32_t nsb;
  uint32_t sb_shift;         // log2(rows per superblock), >= 8
  uint32_t generic;          // 1: lv[] / K / cs below describe the index, rec / base above are NULL
  uint32_t nlevels;
  FmxWideLevel lv[FMXW_MAX_LEVELS];
  const uint64_t *K;         // [max_character + 1]
  const uint64_t *cs;        // [max_character + 1] C array on the device (get_f / fl_map)
  uint32_t sym_bytes;        // width of text / pattern symbols (1, 2 or 4; generic indexes only when > 1)
  uint32_t pad;
  // Walk records of a one-level wide index (round 4; FmxDev::walk above is the 32-bit engine's): the same 112-row
  // records, their 11 counters RELATIVE to a walk superblock (2^wsb_shift records), wbase[superblock][16] = the 64-bit
  // values at the superblock's first record in the order { lf_map2(1..5, .), rank0, rank1[1..5] }.  When `walk` is set
  // the index samples in text order: samples[] = SA of the phase-0 rows in row order, and every get_sa -- batched
  // walk, trait call, sample export -- goes through the walk records (there are no phase pieces on this engine).
  const uint4 *walk;
  const uint64_t *wbase;
  uint32_t nwsb, wsb_shift;
  // RLFMIndex (round 4; kind == FMX_KIND_RLFM): always `generic` -- lv[] / K / nsb describe S, the run heads (slen of
  // them, rlfmi.rs:17), cs[] counts RUNS with a smaller head (rlfmi.rs:72-76) -- plus B, B' and, on indexes that locate,
  // the run table lfrun[j] = lf_map(first row of run j) of FmxDev::lfrun, 64 bits per run.  Samples are the reference's
  // rows (sample.rs:33-37).
  uint32_t kind, pad2;
  uint64_t slen;
  FmxWideBits b, bp;
  const uint64_t *lfrun;
  // Text-order sampling of a wide RLFM index (levels 1..FMX_PHASE_MAX_LEVEL, built together with the run table): the
  // phase pieces of FmxDev::phase -- { phase-0 rows before the piece, 3 words of level-bit phases SA[row] mod 2^level }
  // -- with the count RELATIVE to a superblock of 2^psb_shift pieces and pbase[superblock] = the 64-bit count at its
  // start; samples[] = SA of the phase-0 rows in row order.  A walk is exactly SA[row] mod 2^level steps long: two
  // phase probes (start row: the steps to go; final row: its sample's index), the steps, one sample.
  const uint4 *phase;
  const uint64_t *pbase;
  uint32_t psb_shift, npsb;
  // FMIndexMultiPieces (round 4; kind == FMX_KIND_MULTI; multi_pieces.rs): always `generic`; doc[k] = piece id of the
  // k-th end marker in L order, first_row = sa_idx_first_text (multi_pieces.rs:21-22, 57-85), as FmxDev::doc / first_row
  const uint32_t *doc;
  uint64_t doc_count, first_row;
};
#define FMXW_WALK_SB_SHIFT 24u      // records per walk superblock: 2^24 x 112 rows < 2^31, so relative counters fit 32 bits
#define FMXW_WALK_SB_SHIFT_TEST 5u  // FMX_FLAG_FORCE_WIDE: 32 records, so that a small text has many superblocks

struct fmx_index {
  uint64_t layout;       // FMX_LAYOUT of the library that made the handle (first member: checked before anything else is read)
  FmxDev dev;
  FmxWideDev wide;       // valid when is_wide
  int is_wide;
  uint64_t *d_sa64;      // wide + FMX_FLAG_KEEP_SA
  int device;
  uint64_t n;
  uint32_t sym_bytes;      // width on the device (1, 2, 4)
  uint32_t sym_bytes_abi;  // width the caller uses (8 = u64 narrowed to u32 on the host)
  uint64_t max_character;
  uint32_t kind;
  uint32_t level_requested;
  uint32_t flags;
  uint64_t bytes;        // HBM bytes held
  uint64_t nsamples;
  uint64_t runs;
  double build_ms;
  // owned device allocations
  void **d_alloc;        // grows by doubling (fmx_keep); freed by fmx_free
  int nalloc, cap_alloc;
  uint64_t *h_cs;        // character-based C array (sais.rs:9-32), host copy
  uint8_t *d_text;       // FMX_FLAG_KEEP_SA
  uint32_t *d_sa;        // FMX_FLAG_KEEP_SA
  // instrumentation
  int timing;            // 0 off; 1 the last launch (events + executed-step counter); 2 a series of launches (events only)
  hipEvent_t ev0, ev1;
  int ev_valid;
  hipEvent_t *ev_series; // timing == 2: FMX_SERIES_CAP pairs, created on first use
  int series_n;
  uint64_t *d_steps;     // device counter
};

// The measurement libraries (libfmx_census.so, libfmx_measure.so, libfmx_debug.so) take handles made by libfmx.so: same
// ABI, same structs -- IF they were built from the same sources.  A handle carries the sizes of the structs of the
// library that made it; a library with another layout refuses it (FMX_ERR_ARG) instead of reading pointers at the
// wrong offsets (a stale census library did exactly that in round 4: a GPU memory fault in the middle of bench.py).
static_assert(sizeof(FmxDev) < 1024 && sizeof(fmx_index) < 4096, "FMX_LAYOUT packs the struct sizes into 10 / 12 bits");
#define FMX_LAYOUT (0x464D5800ull << 32 | (uint64_t)sizeof(fmx_index) << 20 | (uint64_t)sizeof(FmxDev) << 10 | \
                    ((uint64_t)sizeof(FmxWideDev) & 0x3FFu))

// ---- internal entry points --------------------------------------------------
void fmx_set_error(int code, const char *detail);
int fmx_hip_fail(hipError_t e, const char *what, int line);
#define FMX_HIP(x)                                        \
  do {                                                    \
    hipError_t _e = (x);                                  \
    if (_e != hipSuccess) return fmx_hip_fail(_e, #x, __LINE__); \
  } while (0)

int fmx_build_impl(fmx_index *idx, const void *d_text);
// per-shard workers of the multi-device entry points (fmx_multi.hip); host pointers, synchronous
//   fmx_locate_batch on entries a .. b of a larger batch's offsets: the hits are out_pos[out_off[0] .. out_off[npat])
int fmx_locate_batch_slice(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                           const uint64_t *out_off, uint64_t *out_pos);
//   patterns resident on the index's device (DEVICE pointers), results into HOST arrays
int fmx_count_resident_slice(const fmx_index *idx, const void *d_pat, const uint64_t *d_pat_off, uint64_t npat,
                             const uint64_t *d_s0e0, uint64_t *out_s, uint64_t *out_e, uint64_t *out_count);
void fmx_set_error_text(const char *text);    // fmx_last_error() of the calling thread, verbatim (a worker thread's message)
// Index-owned (and any other long-lived) device memory.  The process-wide scratch cache of large builds (fmx_build.hip)
// keeps idle blocks out of the driver's hands; an allocation that fails while the cache holds memory of the current
// device returns that memory to the driver and is tried once more, and the room checks that decide the shape of an
// index (text order, walk records, run table, FMX_FLAG_AUTO) count what the cache holds idle as free -- so neither the
// success of a build or query nor the layout of an index depends on what the process built earlier.
hipError_t fmx_dev_malloc(void **p, size_t bytes);
hipError_t fmx_dev_malloc_async(void **p, size_t bytes, hipStream_t st);
hipError_t fmx_dev_mem_info(size_t *free_b, size_t *total_b);
void fmx_release_build_scratch(void);   // the idle small-build buffers of every device (fmx_release_scratch)
// walk records of an index that has the fmt-3 records and the phase pieces (builder, fmx_load): allocates, fills and
// registers dev.walk; no-op (FMX_OK, dev.walk stays NULL) when the index is not eligible
int fmx_make_walk_records(fmx_index *idx);
static inline bool fmx_walk_eligible(const fmx_index *idx) {
  const FmxDev &d = idx->dev;
  return idx->kind == FMX_KIND_FM && !idx->is_wide && idx->sym_bytes == 1 && d.bw.nlevels == 1 && d.bw.lv[0].fmt == 3 &&
         d.max_character <= FMX_WALK_MAX_CHARACTER && d.phase && d.sa_level >= 1 && d.sa_level <= FMX_WALK_MAX_LEVEL && d.n > 0;
}
// wide indexes (fmx_wide.hip / the wide section of fmx_build.hip)
static inline bool fmx_wide_n(uint64_t n) { return n >= 0xFFFFFFF0ull; }
static inline bool fmx_wide_build(const fmx_index *idx) { return fmx_wide_n(idx->n) || (idx->flags & FMX_FLAG_FORCE_WIDE); }
int fmx_build_wide(fmx_index *idx, const void *d_text);
int fmxw_launch_count(const fmx_index *idx, const void *d_pat, const uint64_t *d_off, uint64_t npat,
                      const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e, uint64_t *d_cnt, hipStream_t st);
int fmxw_launch_locate(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                       const uint64_t *d_off, uint64_t total, uint64_t *d_pos, hipStream_t st);
int fmxw_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i, uint64_t k,
                       uint64_t *d_out, hipStream_t st);
int fmxw_launch_export_l(const fmx_index *idx, void *d_out, hipStream_t st);
int fmxw_launch_extract(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len, int forward,
                        void *d_out, uint64_t *d_out_len, uint64_t *d_out_next, hipStream_t st);
int fmxw_launch_match(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat, int prefix_only,
                      const uint64_t *d_off, uint64_t *d_out, hipStream_t st);   // d_off == NULL: counts, else rows
int fmxw_verify_sa(const fmx_index *idx, uint64_t *violations);
int fmxw_launch_compute_K(const FmxWideDev &w, uint64_t *d_K);   // generic wide index: K[c] = cs[c] - rank chain of c at 0
// rows s[k] + j of every interval, 64 bits each (wrapper.rs:203-217), written to out[off[k] + j]
int fmx_launch_expand64(const uint64_t *d_s, const uint64_t *d_e, const uint64_t *d_off, uint64_t npat,
                        uint64_t *d_out, uint64_t total, uint64_t n, uint32_t *status, hipStream_t st);
// The FmxDev a launcher hands to its kernels: idx->dev, with `status` replaced by the calling
// thread's own status word while a host-pointer call is in progress (so that two threads querying
// one handle never read each other's error); the *_dev entry points report into the handle's
// sticky word, read with fmx_stream_status.
uint32_t *fmx_call_status(void);
static inline FmxDev fmx_launch_dev(const fmx_index *idx) {
  FmxDev d = idx->dev;
  if (uint32_t *st = fmx_call_status()) d.status = st;
  return d;
}
// registers a device allocation owned by the index (freed by fmx_free, counted in index bytes)
static inline int fmx_keep(fmx_index *idx, void *p, uint64_t bytes) {
  if (idx->nalloc >= idx->cap_alloc) {
    const int cap = idx->cap_alloc ? idx->cap_alloc * 2 : 64;
    void **grown = (void **)realloc(idx->d_alloc, (size_t)cap * sizeof(void *));
    if (!grown) {
      fmx_set_error(FMX_ERR_ARG, "out of host memory");
      return FMX_ERR_ARG;
    }
    idx->d_alloc = grown;
    idx->cap_alloc = cap;
  }
  idx->d_alloc[idx->nalloc++] = p;
  idx->bytes += bytes;
  return FMX_OK;
}

// max_blocks (0 = the kernel's own choice): cap of the 256-thread grid of the group-per-pattern kernels, for a
// caller that wants workgroup slots left free for kernels of its other streams (fmx_count_batch's copy kernels)
int fmx_launch_count(const fmx_index *idx, const void *d_pat, const uint64_t *d_off,
                     uint64_t npat, const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e,
                     uint64_t *d_cnt, hipStream_t st, unsigned max_blocks = 0);
// rows_ws / tile_ws: caller-provided scratch (fmx_locate_rows_bytes / fmx_offsets_tile_bytes); NULL = take
// it from the stream-ordered allocator for the duration of the call
int fmx_launch_locate(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                      uint64_t npat, const uint64_t *d_off, uint64_t total, uint64_t *d_pos,
                      hipStream_t st, uint32_t *rows_ws = nullptr);
int fmx_launch_offsets(const uint64_t *d_s, const uint64_t *d_e, uint64_t npat, uint64_t *d_off,
                       hipStream_t st, uint64_t *tile_ws = nullptr);
uint64_t fmx_locate_rows_bytes(uint64_t total);
bool fmx_locate_is_one_launch(const fmx_index *idx);   // the default DNA index: no rows array (fmx_locate_f3u_kernel)
uint64_t fmx_offsets_tile_bytes(uint64_t npat);
// fills table[code] for every k-mer code (FMX_FLAG_KMER_TABLE; the index must be complete)
int fmx_launch_kmer_build(const fmx_index *idx, uint2 *d_table, uint32_t k, uint32_t bits, hipStream_t st);
// op: 0 get_l, 1 lf_map, 2 lf_map2, 3 get_sa, 4 get_f, 5 fl_map, 6 piece_id
int fmx_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i,
                      uint64_t k, uint64_t *d_out, hipStream_t st);
int fmx_launch_extract(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len,
                       int forward, void *d_out, uint64_t *d_out_len, uint64_t *d_out_next,
                       hipStream_t st);
// K[c] for every symbol, from the finished wavelet levels (used by the builder)
int fmx_launch_match_counts(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                            uint64_t npat, int prefix_only, uint64_t *d_cnt, hipStream_t st);
int fmx_launch_match_rows(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                          uint64_t npat, int prefix_only, const uint64_t *d_off, uint64_t *d_rows,
                          hipStream_t st);
int fmx_launch_export_l(const fmx_index *idx, void *d_out, hipStream_t st);
int fmx_verify_sa_impl(const fmx_index *idx, uint64_t *violations);
int fmx_launch_compute_K(const FmxMwm &w, const uint64_t *d_cs, uint32_t *d_K,
                         uint32_t max_character, hipStream_t st);
