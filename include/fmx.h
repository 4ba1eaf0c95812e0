/*
 * fmx.h -- C ABI of libfmx: MI355X-native batched FM-index backward search
 * (count) and sampled-suffix-array locate.
 *
 * This is the drop-in boundary for ONE hot path of the Rust crate
 * ajalab/fm-index v0.3.1: Search::search -> lf_map2 loop and the locate walk.
 * The reference has no FFI of its own; the seam is the crate-private trait pair
 * `SearchIndexBackend` + `HasPosition` (src/backend.rs:5-31) consumed by the
 * generic driver in src/wrapper.rs.  Every entry point below names the
 * reference item it replaces (paths relative to the reference repo root).
 * INTEGRATION.md shows the Rust `extern "C"` block + `impl SearchIndexBackend`
 * shim a maintainer would add.
 *
 * Conventions
 *  - rows / positions / counts are uint64_t (= Rust usize).  Texts below 2^32 - 16 symbols run on the
 *    32-bit engine (every kind, every alphabet).  Longer texts are taken by the WIDE engine (64-bit rows,
 *    positions and samples) for every kind, with or without locate (u8 / u16 / u32 symbols, max_character < 2^26;
 *    FMX_KIND_FM over u8 with max_character <= 7 -- DNA -- has the faster one-level kernels): every entry point
 *    works on it (build, count, locate, offsets, the trait calls, piece ids and match rows of a multi-pieces index,
 *    extract, save / load, the exports -- the samples through fmx_export_sa_samples64) except fmx_export_sa and
 *    the opt-in accelerators (flags ignored).
 *  - symbols are `sym_bytes` wide: Character = u8 / u16 / u32 / u64 (character.rs:38-42) =
 *    1 / 2 / 4 / 8.  u64 texts are narrowed to u32 at build time (fmx_build on the host, fmx_build_dev by a
 *    kernel), u64 patterns by the host-pointer entry points; the *_dev query entry points take symbols of
 *    fmx_sym_bytes() bytes -- 1, 2 or 4 (4 for an index built from u64 symbols).  max_character < 2^26.
 *  - `*_dev` entry points take DEVICE pointers and are asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).  The plain
 *    variants take HOST pointers, copy, run the same kernels, and synchronise.
 *  - A handle is immutable after build: any number of threads may issue queries
 *    on it concurrently; build/free are exclusive.  fmx_replicate() copies an index onto another GPU of the node and
 *    the *_multi entry points run ONE batch over several replicas from one host caller (results in place in the caller's
 *    arrays: bit-identical to the one-handle call).
 *  - All compute runs in HIP kernels on the GPU.  There is no CPU fallback:
 *    every call fails with FMX_ERR_HIP when no device is usable.
 */
#ifndef FMX_H
#define FMX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fmx_index fmx_index;

/* ---- status codes ------------------------------------------------------- */
#define FMX_OK 0
/* Error::InvalidText("the given text must not start with zero character")  (sais.rs:128-132) */
#define FMX_ERR_TEXT_START_ZERO 1
/* Error::InvalidText("the given text must end with exactly one zero character") (sais.rs:133-138) */
#define FMX_ERR_TEXT_END_ZERO 2
/* a text or pattern symbol exceeds max_character: the reference panics on cs[c]
 * (fm_index.rs:94, rlfmi.rs:139, sais.rs:18); the ABI reports it instead */
#define FMX_ERR_SYMBOL_RANGE 3
#define FMX_ERR_ARG 4
#define FMX_ERR_UNSUPPORTED 5
#define FMX_ERR_HIP 6
/* locate on an index built without a sampling level (FMIndex / RLFMIndex have no
 * MatchWithLocate impl: frontend.rs:110-133) */
#define FMX_ERR_NO_LOCATE 7

/* ---- index kinds (frontend.rs:110-193) ---------------------------------- */
#define FMX_KIND_FM 0   /* FMIndex / FMIndexWithLocate     (src/fm_index.rs) */
#define FMX_KIND_RLFM 1 /* RLFMIndex / RLFMIndexWithLocate (src/rlfmi.rs)    */
#define FMX_KIND_MULTI 2 /* FMIndexMultiPieces / ...WithLocate (src/multi_pieces.rs): several
                          * \0-separated pieces in one text */
#define FMX_NO_LOCATE 0xFFFFFFFFu /* `level` value for the count-only types */

/* build flags */
#define FMX_FLAG_KEEP_SA 1u /* keep text + full suffix array in HBM (tests / export) */
/* opt-in 2-step acceleration of count for sigma <= 4 texts (max_character <= 4, no interior
 * zeros): a second rank structure over the 2-gram BWT lets one record probe consume TWO pattern
 * symbols (LF(c1, LF(c2, i)) = K2[c1c2] + rank_{c1c2}(BWT2, i)).  Results -- including the
 * (s, e) pair left by the early exit of wrapper.rs:111-113 -- are bit-identical to the 1-step
 * path; +1 byte per text symbol of HBM.  Ignored (1-step index only) when not applicable. */
#define FMX_FLAG_PAIR_INDEX 2u
/* Opt-in k-mer start table (u8 symbols; RLFM, and FM / multi-pieces with max_character <= 15,
 * i.e. one wavelet level): for every k-mer over the symbols
 * 1..max_character the table holds the (s, e) that SearchWrapper::search (wrapper.rs:103-124,
 * early exit included) returns for it from (0, len), so a pattern of >= k symbols starts with ONE
 * 8-byte lookup instead of its first k steps.  k = fmx_kmer_k() = floor(24 / bits per symbol),
 * reduced for short texts so that the table never exceeds len/2 bytes: 12 for DNA and 3 for
 * byte alphabets at len = 2^30 (2^24 entries = 128 MiB).
 * Results stay bit-identical: if the table entry is already an empty range, it is exactly the
 * pair the reference's break would have left.  Patterns shorter than k, patterns whose last k
 * symbols contain 0 or an out-of-range symbol, and refinements from a given (s, e) take the
 * stepwise path.  Ignored when not applicable (wide symbols, texts too short for k >= 2, FM over
 * larger alphabets where the lookup costs more than the steps it replaces). */
#define FMX_FLAG_KMER_TABLE 4u
/* Which rows carry a suffix-array sample (levels 1..4; get_sa() returns SA[i] either way, and
 * fmx_export_sa_samples() always yields the reference's samples SA[k << level], sample.rs:33-37):
 *   text order: the rows whose SA VALUE is a multiple of 2^level (as many samples), plus every row's
 *     phase SA[row] mod 2^level in 16-byte pieces -- a walk is exactly `phase` LF steps, (2^level - 1) / 2
 *     on average and never more than 2^level - 1, where row-order walks are 2^level - 1 on average with a
 *     geometric tail; costs two 16-byte reads per hit and 128 / rows-per-piece bits per row of HBM;
 *   row order: the rows i with i mod 2^level == 0, exactly as SOSampledSuffixArray (sample.rs:21-44).
 * Default: text order for RLFM indexes and FM / multi-pieces indexes over two or more wavelet levels
 * (an LF step there is several dependent requests) and for FM indexes that qualify for walk records (below; the
 * walk records carry the phases, so the two extra reads disappear) when the device has room for them; row order
 * for every other one-level index (an LF step is ONE request, the two extra reads cost more than the steps they
 * save on large batches).  These flags override the default either way; both set, or a level outside 1..4, means
 * row order. */
#define FMX_FLAG_TEXT_ORDER 8u
#define FMX_FLAG_ROW_ORDER 16u
/* A text-order FM index over one 3-bit wavelet level with max_character <= 5 (DNA) and level 1..3 also gets WALK
 * RECORDS (fmx_walk_records() == 1): a second 128-byte-record encoding of the BWT, 112 rows per record, that
 * carries every row's phase, the phase-0 rank and per-symbol phase-1 ranks next to its symbol, so that the batched
 * locate walk needs no phase probe and never reads the record of its last row: max(phase, 1) records and one
 * sample per hit.  +1.14 bytes per text symbol of HBM; derived from the other arrays (not stored in index files).
 * This flag builds the index without them (the round-3 text-order walk). */
#define FMX_FLAG_NO_WALK_RECORDS 64u
/* DEFAULT since round 6 (this flag asked for it in rounds 4-5 and is still accepted): the builder adds the two count
 * accelerators -- FMX_FLAG_PAIR_INDEX | FMX_FLAG_KMER_TABLE -- when they pay and the device has room: FMX_KIND_FM over
 * u8 symbols with max_character <= 4, 2^24 <= n < 2^31, and at least four times the finished index free on the device at
 * build time (2.2 x the count rate on a 1 GiB DNA text for 2.1 x the count structures); FMX_KIND_RLFM over u8 symbols of
 * that size gets the k-mer start table alone (1.22 x the count rate on a 1 GiB byte text for 4 % more index).  Results are bit-identical
 * either way (see the two flags: the reference's (s, e) on every pattern, the pair left by the early exit of
 * wrapper.rs:111-113 included); fmx_has_pair_index() / fmx_kmer_k() tell what the index got.  FMX_FLAG_PLAIN vetoes. */
#define FMX_FLAG_AUTO 128u
/* never add the count accelerators by default: the index runs the reference's loop step for step, one rank per
 * interval end and pattern symbol (the explicit FMX_FLAG_PAIR_INDEX / FMX_FLAG_KMER_TABLE are still honoured) */
#define FMX_FLAG_PLAIN 1024u
/* Keep ALL of this build's temporaries in the library's scratch cache (see fmx_release_scratch) whatever their size,
 * up to three quarters of the device -- for a process that builds several very large indexes in a row (a text of
 * 2^32 symbols needs 137 GB of scratch: above the cache's default cap, so the second such build would pay the
 * runtime ~30 ms per GiB for memory the process has already cycled through: 5 s instead of 0.6).  The default cap
 * (32 GiB or an eighth of the device) covers texts up to ~2^30 symbols without the flag. */
#define FMX_FLAG_KEEP_SCRATCH 256u
/* RLFM indexes that locate: the RUN TABLE lfrun[j] = lf_map(first row of run j) (4 bytes per run; 8 on the wide
 * engine) turns an LF step of the locate walk into two lane-wise reads.  Space policy (round 5): an index type that
 * exists to save space (lib.rs:45-47) gets the table by default only when the text is repetitive enough for it to be a
 * small part of the index -- at most one run per four rows (r <= n / 4: the table is then <= n bytes) -- and the device
 * has room; on a text with about one run per row (random bytes) the table would be 4.3 GB of an 8.9 GB index for a
 * 1 GiB text.  This flag asks for the table whatever r / n is (room permitting); FMX_FLAG_NO_WALK_RECORDS builds
 * without it either way.  fmx_walk_records() tells what the index got. */
#define FMX_FLAG_RUN_TABLE 512u
/* Tests only: build the WIDE engine's index (64-bit rows, see "Conventions") although n < 2^32 - 16, with
 * superblocks of 2^12 rows instead of 2^31 (bit vectors of an RLFM index: 2 records instead of 2^22), so that a
 * small text exercises every part of it.  Every kind; n >= 2; same results. */
#define FMX_FLAG_FORCE_WIDE 32u

/* Message of the last failing call on this thread.  For the two InvalidText codes it
 * is "invalid text: <reference message>" exactly as error.rs:9-15 formats it. */
const char *fmx_last_error(void);
/* Frees the device scratch the library keeps between builds and that no build is using: the cache of large-build
 * temporaries (at most 32 GiB or an eighth of the device per device: on this runtime a hipMalloc of memory the process
 * has already cycled through costs ~30 ms per GiB, so a process that builds several large indexes keeps their
 * temporaries instead of returning them to the driver every time) and the pool of small-build buffers (at most two
 * per device and size class: 4 MiB / 48 MiB, leased by fmx_build* of texts up to 2^17 symbols).  A long-lived
 * service calls it when it is done building; nothing else ever needs to (a failing allocation inside a build empties
 * the cache and retries by itself). */
void fmx_release_scratch(void);
const char *fmx_error_message(int code);

/* ---- construction ------------------------------------------------------- */
/* FMIndex::new / FMIndexWithLocate::new / RLFMIndex::new / RLFMIndexWithLocate::new
 * (frontend.rs:195-267 -> fm_index.rs:25-42, rlfmi.rs:30-96).  `text` is a HOST
 * buffer of n symbols INCLUDING the trailing 0 (Text::text, text.rs:52-54);
 * `max_character` is Text::max_character (text.rs:28-49; 255 for Text::new on u8).
 * Validation is sais.rs:115-139.  Suffix sorting, BWT, rank records and SA samples
 * are all built on the GPU `device`. */
int fmx_build(const void *text, uint64_t n, uint32_t sym_bytes, uint64_t max_character,
              uint32_t kind, uint32_t level, uint32_t flags, int device, fmx_index **out);
/* same, text already resident in HBM on `device` */
int fmx_build_dev(const void *d_text, uint64_t n, uint32_t sym_bytes, uint64_t max_character,
                  uint32_t kind, uint32_t level, uint32_t flags, int device, fmx_index **out);
void fmx_free(fmx_index *idx); /* Drop */
/* flat index file: header + the HBM arrays as they are (the reference has no public on-disk
 * format).  fmx_load uploads it to `device` without rebuilding. */
int fmx_save(const fmx_index *idx, const char *path);
int fmx_load(const char *path, int device, fmx_index **out);

/* ---- SearchIndex (frontend.rs:26-44) ------------------------------------ */
uint64_t fmx_len(const fmx_index *idx);         /* SearchIndexBackend::len, backend.rs:25 */
uint64_t fmx_index_bytes(const fmx_index *idx); /* heap_size (frontend.rs:41-44): HBM bytes */
uint64_t fmx_max_character(const fmx_index *idx);
uint32_t fmx_kind(const fmx_index *idx);
uint32_t fmx_level(const fmx_index *idx); /* effective level (sample.rs:28-31) or FMX_NO_LOCATE */
int fmx_device(const fmx_index *idx);

/* ---- SearchIndexBackend / HasPosition, one call = one trait method ------- */
/* (backend.rs:9-15, 29-31).  Each runs the device kernel on a batch of one; on
 * error they return UINT64_MAX and set fmx_last_error(). */
uint64_t fmx_get_l(const fmx_index *idx, uint64_t i);              /* fm_index.rs:82-84  */
uint64_t fmx_lf_map(const fmx_index *idx, uint64_t i);             /* fm_index.rs:86-91  */
uint64_t fmx_lf_map2(const fmx_index *idx, uint64_t c, uint64_t i); /* fm_index.rs:93-95 */
uint64_t fmx_get_sa(const fmx_index *idx, uint64_t i);             /* fm_index.rs:127-140 */
/* the remaining two trait methods (backend.rs:17-19): the extract path behind
 * Match::iter_chars_forward (wrapper.rs:175-183).  fl_map is always Some for FM / RLFM. */
uint64_t fmx_get_f(const fmx_index *idx, uint64_t i);              /* fm_index.rs:97-112  */
uint64_t fmx_fl_map(const fmx_index *idx, uint64_t i);             /* fm_index.rs:114-120 */

/* batched forms of the same four methods (device pointers, async) */
int fmx_get_l_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                        void *stream);
int fmx_lf_map_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                         void *stream);
int fmx_lf_map2_batch_dev(const fmx_index *idx, const uint64_t *d_c, const uint64_t *d_i,
                          uint64_t k, uint64_t *d_out, void *stream);
int fmx_get_sa_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                         void *stream);
int fmx_get_f_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                        void *stream);
int fmx_fl_map_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                         void *stream);
/* host-pointer forms */
int fmx_get_l_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);
int fmx_lf_map_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);
int fmx_lf_map2_batch(const fmx_index *idx, const uint64_t *c, const uint64_t *i, uint64_t k,
                      uint64_t *out);
int fmx_get_sa_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);
int fmx_get_f_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);
int fmx_fl_map_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);

/* ---- Match::iter_chars_backward / iter_chars_forward for many rows  (wrapper.rs:142-183) -- */
/* For row r = rows[q] the iterator the reference would hand out is run for `len` items:
 *     backward:  c = get_l(i); i = lf_map(i);  yield c            wrapper.rs:154-161 (never ends)
 *     forward :  c = get_f(i); i = fl_map(i)?; yield c            wrapper.rs:175-183
 * out_syms[q*len + t] = t-th yielded character (fmx_sym_bytes() wide, like fmx_export_bwt);
 * out_len[q] (nullable) = number yielded: `len`, except forward on a multi-pieces index, where the
 * iterator ends at the piece's end marker without yielding it (fl_map == None) and the rest of
 * that row's slots are left untouched (dev form) or zero (host form).  out_next[q] (nullable) =
 * the iterator's row after the last item, so a caller can resume in chunks (UINT64_MAX once a
 * forward iterator has ended).  A row >= len() reports FMX_ERR_ARG (out_len = 0). */
int fmx_extract_batch_dev(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint64_t len,
                          int forward, void *d_out_syms, uint64_t *d_out_len, uint64_t *d_out_next,
                          void *stream);
int fmx_extract_batch(const fmx_index *idx, const uint64_t *rows, uint64_t nrows, uint64_t len,
                      int forward, void *out_syms, uint64_t *out_len, uint64_t *out_next);

/* ---- Search::search(..).count()  (wrapper.rs:37-42, 103-134) -------------- */
/* For pattern k = pat[pat_off[k] .. pat_off[k+1]):
 *     (s,e) = s0e0 ? (s0e0[2k], s0e0[2k+1]) : (0, len)          wrapper.rs:41 / 105-106
 *     for c in pattern reversed { s=lf_map2(c,s); e=lf_map2(c,e); if s==e break }
 *     out_s[k]=s; out_e[k]=e; out_count[k]=e-s                   wrapper.rs:132-134
 * `s0e0` (nullable) is the refinement entry: Search::search on an existing Search
 * prepends (wrapper.rs:99-124).  Any of out_s/out_e/out_count may be NULL.
 * A pattern symbol > max_character makes the call report FMX_ERR_SYMBOL_RANGE
 * (that pattern's outputs are 0).  The dev form reports it through
 * fmx_stream_status() after the stream has been synchronised. */
int fmx_count_batch_dev(const fmx_index *idx, const void *d_pat, const uint64_t *d_pat_off,
                        uint64_t npat, const uint64_t *d_s0e0, uint64_t *d_out_s,
                        uint64_t *d_out_e, uint64_t *d_out_count, void *stream);
int fmx_count_batch(const fmx_index *idx, const void *pat, const uint64_t *pat_off, uint64_t npat,
                    const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e, uint64_t *out_count);
/* sticky device-side status of *_dev calls on this handle; reading clears it */
int fmx_stream_status(const fmx_index *idx);

/* ---- Search::iter_matches().map(MatchWithLocate::locate) ------------------ */
/* (wrapper.rs:137-139, 203-217, 238-242 -> fm_index.rs:127-140, sample.rs:46-60)
 *     out_pos[out_off[k] + j] = get_sa(s[k] + j),  j = 0 .. e[k]-s[k]-1
 * i.e. suffix-array order, exactly the reference's iteration order.  `out_off` has
 * npat+1 entries (exclusive scan of the counts; out_off[npat] = total hits).
 * Arguments that are not of this index are reported as FMX_ERR_ARG and never dereferenced: a range with e > len(), a
 * range that does not fit between its offset and total_hits, offsets that leave slots uncovered (the slots take the
 * position of row 0).  What is examined is every range that claims slots of out_pos -- on every path; a range whose
 * slots lie entirely at or beyond total_hits, or whose offsets leave it no slot at all although e > s, writes nothing,
 * and whether it is reported depends on the path the batch takes. */
int fmx_locate_batch_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                         uint64_t npat, const uint64_t *d_out_off, uint64_t total_hits,
                         uint64_t *d_out_pos, void *stream);
int fmx_locate_batch(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                     const uint64_t *out_off, uint64_t *out_pos);
/* exclusive scan helper on the device: d_out_off[k] = sum_{j<k}(e[j]-s[j]), k = 0..npat */
int fmx_offsets_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                    uint64_t *d_out_off, void *stream);
/* Caller-workspace forms of the two calls above.  fmx_locate_batch_dev / fmx_offsets_dev take their
 * scratch (the expanded rows s[k]+j of wrapper.rs:203-217; the scan's tile sums) from the stream-ordered
 * allocator, which keeps them out of hipGraph capture and couples streams that share the pool.  With a
 * workspace of fmx_locate_workspace_bytes(idx, total_hits) / fmx_offsets_workspace_bytes(npat) bytes of
 * device memory (256-byte aligned, owned by the caller, reusable across calls ON THE SAME STREAM) these
 * forms only launch kernels: no allocation, no synchronisation, capturable into a hipGraph, and batches
 * on different streams with different workspaces run concurrently (one batch's long walks under the
 * other's bulk).  Same results, same errors; a workspace that is NULL or too small is FMX_ERR_ARG.
 * (Round 5: on the default DNA index -- text order + walk records -- a locate batch is ONE kernel that expands its
 * slices of the hits in LDS: fmx_locate_batch_dev allocates nothing there either, and the workspace is 256 bytes.) */
uint64_t fmx_locate_workspace_bytes(const fmx_index *idx, uint64_t total_hits);
int fmx_locate_batch_ws_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                            uint64_t npat, const uint64_t *d_out_off, uint64_t total_hits,
                            uint64_t *d_out_pos, void *d_workspace, uint64_t workspace_bytes,
                            void *stream);
uint64_t fmx_offsets_workspace_bytes(uint64_t npat);
int fmx_offsets_ws_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                       uint64_t *d_out_off, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* ---- one batch over several replicas of an index, from ONE host caller ------ */
/* (SURVEY section 8e / BASELINE config 5.)  The driver of the reference (wrapper.rs:103-124, behind Search::search,
 * frontend.rs:70-98) reads only immutable index state: patterns are independent, so a batch shards over G replicas of
 * the index -- one per GPU of the node -- with no exchange but the gather of the results, and for a host caller the
 * gather is the layout of its own arrays: every shard writes its results IN PLACE at its offset of the caller's
 * out_* arrays.  Results are bit-identical to the one-handle call whatever G is.
 *   shard r of G = patterns [ceil(N r / G), ceil(N (r + 1) / G))     (fmx_shard_range; pattern k -> floor(k G / N)),
 * searched on idx[r] by a host thread of its own (shard 0 on the caller's) with that thread's streams, pinned staging
 * and device scratch.  The handles must be replicas of ONE index (fmx_replicate, or fmx_build / fmx_load of the same
 * text with the same parameters); several of them may sit on one device.  Error codes and fmx_last_error() are those
 * of the first failing shard, in shard order. */
/* a second handle holding its own copy of every HBM array of `src`, on `device` (any device of the node, src's own
 * included): arrays travel device to device (hipMemcpyPeer over xGMI), nothing is rebuilt and the text is not needed.
 * FMX_FLAG_KEEP_SA's text + suffix array stay with the source (the replica has neither). */
int fmx_replicate(const fmx_index *src, int device, fmx_index **out);
void fmx_shard_range(uint64_t nitems, uint32_t nshards, uint32_t r, uint64_t *begin, uint64_t *end);
/* fmx_count_batch over `ndev` replicas: same arguments (HOST pointers), same results, same errors */
int fmx_count_batch_multi(fmx_index *const *idx, uint32_t ndev, const void *pat, const uint64_t *pat_off,
                          uint64_t npat, const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e,
                          uint64_t *out_count);
/* the same with every shard's patterns already RESIDENT on its device, results into HOST arrays -- the measurement
 * protocol of SURVEY section 8d ("exclude ... H2D upload from both sides; include D2H of results on the GPU side";
 * benches/count.rs:29-37 times results the caller can read).  d_pat[r] / d_pat_off[r] / d_s0e0[r] (nullable array,
 * nullable entries) are DEVICE pointers on fmx_device(idx[r]) holding shard r's patterns: fmx_sym_bytes() wide
 * symbols, (shard length + 1) offsets relative to d_pat[r], 2 x shard length refinement pairs.  Page-locked out_*
 * arrays are written by the search kernels themselves (posted writes over the host link); pageable ones are staged. */
int fmx_count_batch_multi_resident(fmx_index *const *idx, uint32_t ndev, const void *const *d_pat,
                                   const uint64_t *const *d_pat_off, uint64_t npat,
                                   const uint64_t *const *d_s0e0, uint64_t *out_s, uint64_t *out_e,
                                   uint64_t *out_count);
/* fmx_locate_batch over `ndev` replicas: shard r locates patterns [a, b) of the batch and writes
 * out_pos[out_off[a] .. out_off[b]) -- the reference's iteration order (wrapper.rs:203-217), in place.
 * out_off[0] must be 0. */
int fmx_locate_batch_multi(fmx_index *const *idx, uint32_t ndev, const uint64_t *s, const uint64_t *e,
                           uint64_t npat, const uint64_t *out_off, uint64_t *out_pos);

/* ---- multi-pieces index (src/multi_pieces.rs, frontend.rs:46-68, 100-104) --- */
/* SearchIndexWithMultiPieces::search_prefix / search_suffix / search_exact are the same
 * backward search with a different start and a filter (wrapper.rs:57-82):
 *   search_suffix / search_exact start from (0, pieces_count) -> pass that pair as s0e0;
 *   search_prefix / search_exact keep only matches whose row has L == 0 (wrapper.rs:208)
 *   -> prefix_only = 1 below.  count() stays e - s in the reference (wrapper.rs:132-134). */
uint64_t fmx_pieces_count(const fmx_index *idx);               /* HasMultiPieces::pieces_count */
/* HasMultiPieces::piece_id (multi_pieces.rs:206-219) for SA rows */
uint64_t fmx_piece_id(const fmx_index *idx, uint64_t i);
int fmx_piece_id_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out,
                           void *stream);
int fmx_piece_id_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out);
/* number of matches iter_matches() yields for [s, e): e - s, or with prefix_only the rows of
 * the range whose L symbol is 0 (MatchIteratorWrapper::next, wrapper.rs:203-217) */
int fmx_match_counts_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                         uint64_t npat, int prefix_only, uint64_t *d_out_count, void *stream);
int fmx_match_counts(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                     int prefix_only, uint64_t *out_count);
/* the SA rows iter_matches() visits, in its order: out_rows[out_off[k] + j] (out_off = exclusive
 * scan of the match counts).  Feed them to fmx_get_sa_batch / fmx_piece_id_batch. */
int fmx_match_rows_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                       uint64_t npat, int prefix_only, const uint64_t *d_out_off,
                       uint64_t *d_out_rows, void *stream);
int fmx_match_rows(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                   int prefix_only, const uint64_t *out_off, uint64_t *out_rows);

/* ---- instrumentation / export (tests, bench, checkers) -------------------- */
/* milliseconds the last *_dev query kernel of each type took, measured with HIP
 * events on the launch stream when timing is enabled */
void fmx_set_timing(fmx_index *idx, int enabled);
double fmx_last_kernel_ms(const fmx_index *idx);
/* enabled == 2: a SERIES -- every following count / locate launch of the index (up to 64) gets its own pair of events
 * around its dominant kernel, nothing else changes (no step counter, no synchronisation), so that launches issued back
 * to back are timed as they run back to back.  fmx_series_kernel_ms waits for them, returns their mean duration in
 * milliseconds (-1 when there is none) and starts a new series.  Instrumentation for ONE measuring thread: while timing
 * is enabled (1 or 2) the launches of an index share its event slots and must come from a single thread; with timing
 * off (the default) launches on a const handle are thread-safe as everywhere else. */
double fmx_series_kernel_ms(fmx_index *idx);
/* LF steps executed by the last count / locate call when timing is enabled (else 0) */
uint64_t fmx_last_steps(const fmx_index *idx);
double fmx_build_ms(const fmx_index *idx);
int fmx_export_bwt(const fmx_index *idx, void *host_out);            /* n symbols of fmx_sym_bytes() */
int fmx_export_cs(const fmx_index *idx, uint64_t *host_out);         /* max_character+1 (sais.rs:9-32) */
int fmx_export_sa_samples(const fmx_index *idx, uint32_t *host_out); /* ((n-1)>>level)+1; n < 2^32 */
int fmx_export_sa_samples64(const fmx_index *idx, uint64_t *host_out); /* the same, 64 bits each; also wide indexes */
uint64_t fmx_num_samples(const fmx_index *idx);
int fmx_export_sa(const fmx_index *idx, uint32_t *host_out);         /* needs FMX_FLAG_KEEP_SA */
/* needs FMX_FLAG_KEEP_SA: number of adjacent suffix pairs out of order + number of
 * indices not hit exactly once (0 = the array IS the suffix array) */
int fmx_verify_sa(const fmx_index *idx, uint64_t *violations);
uint64_t fmx_num_runs(const fmx_index *idx);                         /* RLFM: r (rlfmi.rs:43) */
uint32_t fmx_sym_bytes(const fmx_index *idx);                        /* symbol width in HBM / *_dev */
uint32_t fmx_kmer_k(const fmx_index *idx);   /* k of the FMX_FLAG_KMER_TABLE table, 0 = none */
int fmx_has_pair_index(const fmx_index *idx);                        /* FMX_FLAG_PAIR_INDEX honoured? */
int fmx_is_wide(const fmx_index *idx);        /* 1: served by the wide (64-bit rows) engine */
/* 1: the index carries the derived structure that speeds up its batched locate walk -- walk records (FM, DNA-like),
 * or the run table of an RLFM index: lf_map of every run's first row, 4 bytes per run, so that an LF step of the walk
 * is the B piece of the row + one table entry instead of B piece + S records + B' select (rlfmi.rs:127-133).
 * FMX_FLAG_NO_WALK_RECORDS builds without either. */
int fmx_walk_records(const fmx_index *idx);
int fmx_text_order(const fmx_index *idx);     /* 1: suffix-array samples in text order (FMX_FLAG_TEXT_ORDER) */

#ifdef __cplusplus
}
#endif
#endif
