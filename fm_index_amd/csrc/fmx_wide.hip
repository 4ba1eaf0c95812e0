// fmx_wide.hip -- the query kernels of WIDE indexes: texts of n >= 2^32 - 16 symbols, whose rows and positions
// need the reference's `usize` (fm_index.rs:86-95, 127-140; wrapper.rs:103-134, 203-217, 238-242).
//
// A second instantiation next to the 32-bit engine of fmx_query.hip, which it leaves untouched: FMIndex /
// FMIndexWithLocate over byte texts.  Layout: fmx_internal.h (FmxWideDev) -- the same 128-byte records, counters
// relative to a 2^31-row superblock, a table of 64-bit bases.  A rank is one 128-byte line fetched by an 8-lane
// group (one dwordx4 per lane), popcount + three DPP adds in 32 bits, and one 64-bit add of the base.
// First half of the file: one-level alphabets (max_character <= 7, DNA); second half: the generic kernels for
// larger byte alphabets (the multi-ary wavelet levels of the 32-bit engine with per-level bases and a 64-bit K[]).
//
// Shapes: one-level indexes -- a group owns a pattern (count: the shape of the 32-bit engine's fmx_count_f3_kernel); locate:
// the text-order walk over walk records with the hit queue, distributed walk state and write-combining ring of the
// 32-bit engine's fmx_locate_f3t_kernel.  Generic indexes (round 5) -- count with an interval endpoint per lane
// (fmxw_g_count_ep_kernel: FM, RLFM, multi-pieces; the shape of fmx_ep.h), text-order locate a walk per lane
// (fmxw_g_walk_text_ep_kernel; RLFM with the run table: fmxw_r_walk_text_kernel / fmxw_r_walk_kernel), a group per walk for
// row-order samples; the group-per-pattern / four-walks-per-group kernels of round 4 stay for the measurement build's A/B
// (FMXW_R_COUNT_GROUP=1).  Third part: RLFMIndex (B / B' with 64-bit superblock bases, S on the generic levels).
#include "fmx_device.h"

#define FMXW_BLOCK 256
#define FMXW_MAX_BLOCKS 2048
#define FMXW_EP_BLOCKS 1024u     // RLFM count, endpoint per lane: grid cap (as the 32-bit engine's ep_blocks)

static inline unsigned fmxw_grid(uint64_t units) {
  uint64_t blocks = (units * FMX_GROUP + FMXW_BLOCK - 1) / FMXW_BLOCK;
  if (blocks < 1) blocks = 1;
  if (blocks > FMXW_MAX_BLOCKS) blocks = FMXW_MAX_BLOCKS;
  return (unsigned)blocks;
}
static inline FmxWideDev fmxw_dev(const fmx_index *idx) {
  FmxWideDev w = idx->wide;
  if (uint32_t *st = fmx_call_status()) w.status = st;    // a host-pointer call reports into its own status word
  return w;
}

// lf_map2(c, i) = cs[c] + rank_c(BWT, i), i in [0, n]                                        fm_index.rs:93-95
__device__ __forceinline__ uint64_t fmxw_lf_map2(const FmxWideDev &w, uint32_t c, uint64_t i, uint32_t g) {
  FMX_CHECK((i >> 8) < w.n / 256u + 1u && (i >> w.sb_shift) < w.nsb);
  const uint4 p = w.rec[(size_t)(i >> 8) * 8u + g];
  const uint64_t b = w.base[(size_t)(i >> w.sb_shift) * 8u + c];
  return b + fmx_group_sum(fmx_piece_rank<3>(p, (uint32_t)i & 255u, c, g));
}
// get_l(i) and lf_map(i) from the same record                                              fm_index.rs:82-91
__device__ __forceinline__ uint64_t fmxw_lf_map(const FmxWideDev &w, uint64_t i, uint32_t g, uint32_t &sym) {
  FMX_CHECK(i < w.n);
  const uint4 p = w.rec[(size_t)(i >> 8) * 8u + g];
  const uint32_t off = (uint32_t)i & 255u;
  sym = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p, off & 31u) : 0u);
  const uint64_t b = w.base[(size_t)(i >> w.sb_shift) * 8u + sym];
  return b + fmx_group_sum(fmx_piece_rank<3>(p, off, sym, g));
}

// get_f(i) and fl_map(i) (fm_index.rs:97-120): the greatest c with cs[c] <= i -- cs[c] is the base of superblock 0 --
// and the position of the (i - cs[c])-th c of the BWT: the last superblock, then the last record of it, whose
// absolute counter of c is <= i, then the entry inside that record.  (Plain binary searches: the extract path is
// not the hot path.)
__device__ __forceinline__ uint64_t fmxw_fl_map(const FmxWideDev &w, uint64_t i, uint32_t g, uint32_t &sym) {
  uint32_t c = 0;
  for (uint32_t t = 1; t <= w.max_character; t++)
    if (w.base[t] <= i) c = t;                      // cs[] is non-decreasing
  sym = c;
  uint32_t sb = 0;
  for (uint32_t t = 1; t < w.nsb; t++)
    if (w.base[(size_t)t * 8u + c] <= i) sb = t;
  const uint64_t b = w.base[(size_t)sb * 8u + c];
  const uint32_t nrec = (uint32_t)(w.n / 256u + 1u), recs = w.sb_shift - 8u;
  uint32_t lo = sb << recs, hi = ((sb + 1u) << recs) - 1u;
  if (hi > nrec - 1u) hi = nrec - 1u;
  while (lo < hi) {                                 // last record whose counter <= i
    const uint32_t mid = lo + (hi - lo + 1u) / 2u;
    if (b + w.rec[(size_t)mid * 8u + c].x <= i) lo = mid; else hi = mid - 1u;
  }
  const uint32_t rem = (uint32_t)(i - (b + w.rec[(size_t)lo * 8u + c].x));   // rem-th match inside the record
  const uint32_t m = fmx_piece_match<3>(w.rec[(size_t)lo * 8u + g], c);
  const uint32_t mine = __popc(m);
  uint32_t before = 0;
#pragma unroll
  for (uint32_t q = 0; q < FMX_GROUP; q++) {
    const uint32_t cq = fmx_group_sum(g == q ? mine : 0u);
    before += (q < g) ? cq : 0u;
  }
  const bool here = rem >= before && rem < before + mine;
  const uint32_t pos = here ? g * 32u + fmx_select32(m, rem - before) : 0u;
  return (uint64_t)lo * 256u + fmx_group_sum(pos);
}

// counter of the superblock table a step needs: lf_map2 of the row's symbol while the walk goes on (phase >= 2), the
// phase-0 rank on a phase-0 row, rank1[symbol] on a phase-1 row (fmx_walk_step_rel)
__device__ __forceinline__ uint32_t fmxw_walk_counter(uint32_t sym, uint32_t ph) {
  return ph >= 2u ? sym - 1u : (ph == 0u ? 5u : 5u + sym);
}

// get_sa(row) through the walk records, for a group that holds one row (trait call, sample export): at most
// 2^level - 1 LF steps, the last of them without a record of its own                         fm_index.rs:127-140
__device__ __forceinline__ uint64_t fmxw_get_sa_walk(const FmxWideDev &w, uint64_t row, uint32_t g) {
  uint32_t steps = 0;
  for (;;) {
    uint32_t off, sym, ph, si;
    const uint64_t rec = fmx_walk_record64(row, off);
    FMX_CHECK(rec < w.n / FMX_WALK_ROWS + 1u);
    const uint4 p = w.walk[(size_t)rec * 8u + g];
    const uint32_t nr = fmx_walk_step_rel(p, off, g, sym, ph, si);
    if (ph >= 2u && sym == 0u) return ~0ull;          // (cannot happen: the one row with symbol 0 -- SA = 0 -- has phase 0)
    const uint64_t b = w.wbase[(size_t)(rec >> w.wsb_shift) * 16u + fmxw_walk_counter(sym, ph)];
    if (ph <= 1u) {                                   // the sample of this row (phase 0) or of the next one (phase 1)
      steps += ph;
      uint64_t v = w.samples[b + si] + steps;         // (sa + steps) % len
      if (v >= w.n) v -= w.n;
      return v;
    }
    row = b + nr;                                     // i = lf_map(i); steps += 1
    steps++;
  }
}

// The superblock bases are read with a data-dependent symbol in every step.  Up to FMXW_LDS_SB superblocks
// (n < 2^37) every block keeps them in LDS (kernels instantiated with LDSB = true: ds_read, 32-bit address); beyond
// that -- and for the many tiny superblocks of FMX_FLAG_FORCE_WIDE test indexes -- the LDSB = false instantiation
// reads them from global memory.  Two instantiations rather than one pointer that may point to either: a generic
// pointer makes every base read a flat_load, which goes down the vector-memory path next to the record loads
// (0.90 ms against 0.69 ms per 2^20 x 32 batch at n = 2^30, benchmarks/gpu/wide_tune.sh).
#define FMXW_LDS_SB 64u
#define FMXW_BASES(w, LDSB)                                                                          \
  __shared__ uint64_t lds_base[(LDSB) ? FMXW_LDS_SB * 8u : 1u];                                        \
  if (LDSB) {                                                                                        \
    for (uint32_t t_ = threadIdx.x; t_ < (w).nsb * 8u; t_ += blockDim.x) lds_base[t_] = (w).base[t_];  \
    __syncthreads();                                                                                 \
  }                                                                                                  \
  auto base_at = [&](uint64_t row_, uint32_t c_) -> uint64_t {                                       \
    const uint32_t sb_ = (uint32_t)(row_ >> (w).sb_shift);                                           \
    if constexpr (LDSB) return lds_base[sb_ * 8u + c_]; else return (w).base[(size_t)sb_ * 8u + c_]; \
  }

// SearchWrapper::search for a batch (wrapper.rs:103-124): a group per pattern.  Per step the two record loads and
// the NEXT pattern symbol are requested together and waited for once; the base of the step's symbol comes from LDS.
template <bool LDSB>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_count_kernel(
    FmxWideDev w, const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  FMXW_BASES(w, LDSB);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t ptot = npat ? off[npat] : 0;       // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  uint64_t nsteps = 0;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    const uint64_t pbeg = off[k], pend = off[k + 1];
    uint64_t j = pend - pbeg;
    // offsets that go backwards or leave the pattern buffer: refuse, do not read
    bool bad = pend < pbeg || pbeg < pmin || pend > ptot;
    uint64_t s = 0, e = w.n;                        // SearchIndexWrapper::search: (0, len)   wrapper.rs:41
    if (s0e0) {                                     // Search::search on an existing Search   wrapper.rs:105-106
      s = s0e0[2 * k];
      e = s0e0[2 * k + 1];
      bad |= s > w.n || e > w.n;                    // not a range of this index
    }
    if (bad) {
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
      s = 0; e = 0; j = 0;
    }
    uint32_t c = j ? pat[pbeg + j - 1] : 0u;        // for c in pattern.iter().rev()          wrapper.rs:108
    while (j) {
      if (c > w.max_character) {                    // reference: panic on cs[c]
        if (g == 0) atomicOr(w.status, 1u << FMX_ERR_SYMBOL_RANGE);
        s = 0; e = 0;
        break;
      }
      FMX_CHECK((e >> 8) < w.n / 256u + 1u && (s >> w.sb_shift) < w.nsb && (e >> w.sb_shift) < w.nsb);
      const uint4 pa = w.rec[(size_t)(s >> 8) * 8u + g], pb = w.rec[(size_t)(e >> 8) * 8u + g];
      const uint32_t cn = j > 1 ? pat[pbeg + j - 2] : 0u;      // rides along with the record loads
      const uint64_t ba = base_at(s, c), bb = base_at(e, c);
      s = ba + fmx_group_sum(fmx_piece_rank<3>(pa, (uint32_t)s & 255u, c, g));      // wrapper.rs:109
      e = bb + fmx_group_sum(fmx_piece_rank<3>(pb, (uint32_t)e & 255u, c, g));      // wrapper.rs:110
      c = cn;
      j--;
      nsteps++;
      if (s == e) break;                            // wrapper.rs:111-113
    }
    if (g == 0) {
      if (out_s) out_s[k] = s;
      if (out_e) out_e[k] = e;
      if (out_cnt) out_cnt[k] = e - s;              // wrapper.rs:132-134
    }
  }
  if (steps_out && g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// get_sa for a batch of rows (fm_index.rs:127-140; sample.rs:46-60).  `io` holds the row on entry and the position
// on exit -- the expanded rows of iter_matches (wrapper.rs:203-217) are written straight into the caller's
// position array, so a wide locate needs no workspace.  A group keeps WALKS walks going; slot q owns the hits
// gid + (q + WALKS t) ngroups.  One iteration is ONE memory round trip for every slot, all requested before the one
// wait: a walk on an unsampled row requests its record (8 lanes x 16 bytes); a walk on a sampled row requests its
// sample on lane 0 and -- on the other lanes -- the row of the slot's next hit, so a walk of k LF steps costs k + 1
// round trips and nothing that was loaded is waited for in a later iteration.
template <int WALKS, bool LDSB>
__global__ __launch_bounds__(FMXW_BLOCK, 8) void fmxw_walk_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                   uint64_t *__restrict__ steps_out) {
  FMXW_BASES(w, LDSB);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t lmask = (1ull << w.sa_level) - 1ull, stride = (uint64_t)WALKS * ngroups;
  constexpr uint64_t NONE = ~0ull;
  uint64_t h[WALKS], row[WALKS], steps[WALKS];
  bool fresh[WALKS];                                // the slot's first hit: its row has not been read yet
  uint64_t nsteps = 0;
  bool any = false;
#pragma unroll
  for (int q = 0; q < WALKS; q++) {
    h[q] = gid + (uint64_t)q * ngroups;
    if (h[q] >= total) h[q] = NONE;
    row[q] = 0; steps[q] = 0;
    fresh[q] = true;
    any |= h[q] != NONE;
  }
  while (any) {
    uint4 p[WALKS];
    uint64_t aux[WALKS] = {};                       // lane 0: the sample; lanes 1-7: the row of the slot's next hit
    int what[WALKS];                                // 0 idle, 1 first row, 2 not a row of this index, 3 sampled, 4 LF step
#pragma unroll
    for (int q = 0; q < WALKS; q++) {
      what[q] = h[q] == NONE ? 0 : fresh[q] ? 1 : row[q] >= w.n ? 2 : (row[q] & lmask) == 0 ? 3 : 4;
      if (what[q] == 4) {                           // None: the record of lf_map             fm_index.rs:134-135
        p[q] = w.rec[(size_t)(row[q] >> 8) * 8u + g];
      } else {                                      // one load instruction for both kinds of lane
        const uint64_t want = what[q] == 1 ? h[q] : h[q] + stride;
        const bool more = what[q] == 1 || (what[q] != 0 && total - h[q] > stride);
        const uint64_t *from = g == 0 ? w.samples + (row[q] >> w.sa_level) : io + want;   // Some(sa)   fm_index.rs:131
        if (g == 0 ? what[q] == 3 : more) aux[q] = *from;
      }
    }
    any = false;
#pragma unroll
    for (int q = 0; q < WALKS; q++) {
      if (what[q] == 4) {                           // i = lf_map(i); steps += 1              fm_index.rs:135-136
        const uint32_t off = (uint32_t)row[q] & 255u;
        const uint32_t sym = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p[q], off & 31u) : 0u);
        row[q] = base_at(row[q], sym) + fmx_group_sum(fmx_piece_rank<3>(p[q], off, sym, g));
        steps[q]++;
      } else if (what[q] != 0) {
        if (what[q] != 1 && g == 0) {
          uint64_t v = NONE;
          if (what[q] == 2) {                       // refused, nothing was read
            atomicOr(w.status, 1u << FMX_ERR_ARG);
          } else {                                  // (sa + steps) % len                     fm_index.rs:132
            v = aux[q] + steps[q];
            if (v >= w.n) v -= w.n;
          }
          io[h[q]] = v;
        }
        if (what[q] != 1) {
          nsteps += steps[q];
          h[q] = total - h[q] > stride ? h[q] + stride : NONE;
        }
        // the next row stands on lanes 1-7 of the group: lane 1 / lane 5 of the two quads
        row[q] = (uint64_t)fmx_quad_bcast_c<1>((uint32_t)aux[q]) | (uint64_t)fmx_quad_bcast_c<1>((uint32_t)(aux[q] >> 32)) << 32;
        steps[q] = 0;
        fresh[q] = false;
      }
      any |= h[q] != NONE;
    }
  }
  if (steps_out && g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// the trait methods, batched (backend.rs:9-19, 29-31).  op: 0 get_l, 1 lf_map, 2 lf_map2, 3 get_sa, 4 get_f, 5 fl_map
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_scalar_kernel(FmxWideDev w, int op, const uint64_t *__restrict__ cc,
                                                                  const uint64_t *__restrict__ ii, uint64_t k,
                                                                  uint64_t *__restrict__ out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < k; q += ngroups) {
    const uint64_t i = ii[q];
    uint64_t res = ~0ull;
    if (op == 2) {                                  // lf_map2(c, i), i in [0, n]
      const uint64_t c = cc[q];
      if (c > w.max_character || i > w.n) {
        if (g == 0) atomicOr(w.status, 1u << (c > w.max_character ? FMX_ERR_SYMBOL_RANGE : FMX_ERR_ARG));
      } else {
        res = fmxw_lf_map2(w, (uint32_t)c, i, g);
      }
    } else if (i >= w.n) {
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
    } else if (op == 0 || op == 1) {
      uint32_t sym;
      const uint64_t r = fmxw_lf_map(w, i, g, sym);
      res = op == 0 ? (uint64_t)sym : r;
    } else if (op == 4 || op == 5) {                // get_f / fl_map
      uint32_t sym;
      const uint64_t r = fmxw_fl_map(w, i, g, sym);
      res = op == 4 ? (uint64_t)sym : r;
    } else if (w.walk) {                            // get_sa, text-order samples: through the walk records
      res = fmxw_get_sa_walk(w, i, g);
    } else {                                        // get_sa
      const uint64_t lmask = (1ull << w.sa_level) - 1ull;
      uint64_t row = i, steps = 0;
      while (row & lmask) {
        uint32_t sym;
        row = fmxw_lf_map(w, row, g, sym);
        steps++;
      }
      uint64_t v = w.samples[row >> w.sa_level] + steps;
      if (v >= w.n) v -= w.n;
      res = v;
    }
    if (g == 0) out[q] = res;
  }
}

// Match::iter_chars_backward / iter_chars_forward for many rows (wrapper.rs:154-183): one group per row
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_extract_kernel(FmxWideDev w, const uint64_t *__restrict__ rows,
                                                                   uint64_t nrows, uint32_t len, int forward,
                                                                   uint8_t *__restrict__ out, uint64_t *__restrict__ out_len,
                                                                   uint64_t *__restrict__ out_next) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < nrows; q += ngroups) {
    uint64_t i = rows[q], next = ~0ull;
    uint32_t t = 0;
    if (i >= w.n) {
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
    } else {
      uint8_t *dst = out + q * (uint64_t)len;
      for (; t < len; t++) {
        uint32_t sym;
        i = forward ? fmxw_fl_map(w, i, g, sym) : fmxw_lf_map(w, i, g, sym);
        if (g == 0) dst[t] = (uint8_t)sym;
      }
      next = i;
    }
    if (g == 0 && out_len) out_len[q] = t;
    if (g == 0 && out_next) out_next[q] = next;
  }
}

// L column of rows [0, n), one byte per row (get_l)
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_export_l_kernel(FmxWideDev w, uint8_t *__restrict__ out) {
  // one lane per row here: the code of entry `off` is three bits of one piece
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < w.n; i += stride) {
    const uint32_t off = (uint32_t)i & 255u;
    const uint4 p = w.rec[(size_t)(i >> 8) * 8u + (off >> 5)];
    out[i] = (uint8_t)fmx_piece_code<3>(p, off & 31u);
  }
}

// ===========================================================================
// Walk records on the wide engine (round 4; FmxWideDev::walk): the text-order walk of fmx_locate_f3t_kernel
// (fmx_query.hip) with 64-bit rows, positions and sample indices.  A record's counters are relative to its walk
// superblock; the 64-bit bases sit in LDS (LDSW, up to FMXW_LDS_WSB superblocks) or in global memory.
// ===========================================================================
#define FMXW_LDS_WSB 32u
#define FMXW_WBASES(w, LDSW)                                                                          \
  __shared__ uint64_t lds_wb[(LDSW) ? FMXW_LDS_WSB * 16u : 1u];                                        \
  if (LDSW) {                                                                                         \
    for (uint32_t t_ = threadIdx.x; t_ < (w).nwsb * 16u; t_ += blockDim.x) lds_wb[t_] = (w).wbase[t_]; \
    __syncthreads();                                                                                  \
  }                                                                                                   \
  auto wbase_at = [&](uint64_t rec_, uint32_t k_) -> uint64_t {                                       \
    const uint32_t sb_ = (uint32_t)(rec_ >> (w).wsb_shift);                                           \
    if constexpr (LDSW) return lds_wb[sb_ * 16u + k_]; else return (w).wbase[(size_t)sb_ * 16u + k_]; \
  }
// hit queue of a block over 64-bit rows that stand where their positions will be written (fmx_launch_expand64): the
// FmxHitQueue of fmx_query.hip with two registers per resident row
struct FmxwHitQueue {
  const uint64_t *rows;
  uint32_t nhits, chunk, lane, c0, c1, used;
  uint32_t w0lo, w0hi, w1lo, w1hi;
  static constexpr uint32_t NOCHUNK = 0xFFFFFFFFu;
  __device__ __forceinline__ void load_win(uint32_t c, uint32_t &lo, uint32_t &hi) const {
    const uint32_t x = c * chunk + lane;
    const uint64_t v = (c != NOCHUNK && lane < chunk && x < nhits) ? rows[x] : 0ull;
    lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
  }
  __device__ __forceinline__ uint32_t valid(uint32_t c) const { return (c != NOCHUNK && c * chunk < nhits) ? c : NOCHUNK; }
  __device__ __forceinline__ uint32_t draw(unsigned int &counter) const {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  }
  __device__ __forceinline__ void init(const uint64_t *r, uint32_t count, uint32_t rows_per_ticket, uint32_t ln,
                                       unsigned int &counter) {
    rows = r; nhits = count; chunk = rows_per_ticket; lane = ln;
    c0 = valid(draw(counter));
    load_win(c0, w0lo, w0hi);
    __syncthreads();                                  // every wave of the block draws its first ticket before any draws a second
    c1 = valid(draw(counter));
    load_win(c1, w1lo, w1hi);
    used = 0;
  }
  __device__ __forceinline__ bool take(uint32_t rank, uint32_t &x, uint64_t &row, bool &first) const {
    const uint32_t idx = used + rank;
    first = idx < chunk;
    const uint32_t c = first ? c0 : c1, within = first ? idx : idx - chunk;
    const int src = (int)(within & 63u);
    const uint32_t a0 = (uint32_t)__shfl((int)w0lo, src), a1 = (uint32_t)__shfl((int)w0hi, src);
    const uint32_t b0 = (uint32_t)__shfl((int)w1lo, src), b1 = (uint32_t)__shfl((int)w1hi, src);
    row = first ? ((uint64_t)a1 << 32 | a0) : ((uint64_t)b1 << 32 | b0);
    x = c * chunk + within;
    return idx < 2u * chunk && c != NOCHUNK && x < nhits;
  }
  __device__ __forceinline__ bool advance(uint32_t count, unsigned int &counter) {
    used += count;
    if (used >= chunk) {                              // wave-uniform: slide
      used -= chunk;
      c0 = c1;
      c1 = c1 != NOCHUNK ? valid(draw(counter)) : NOCHUNK;
      w0lo = w1lo; w0hi = w1hi;
      load_win(c1, w1lo, w1hi);
      return true;
    }
    return false;
  }
};

#define FMXW_LOC_BLOCK 1024
#define FMXW_WC_SLOTS 4
template <int Q, bool WC, bool LDSW>
__global__ __launch_bounds__(FMXW_LOC_BLOCK) void fmxw_walk_t_kernel(FmxWideDev w, uint64_t total, uint32_t hits_per_block,
                                                                      uint32_t chunk, uint64_t *__restrict__ io,
                                                                      uint64_t *__restrict__ steps_out) {
  static_assert(Q == 1 || Q == 4, "walks per group");
  __shared__ unsigned int lds_q;
  __shared__ uint64_t wc_ring[WC ? (FMXW_LOC_BLOCK / 64) * FMXW_WC_SLOTS * 64 : 1];
  __shared__ uint32_t wc_tag[WC ? (FMXW_LOC_BLOCK / 64) * FMXW_WC_SLOTS : 1];
  if (threadIdx.x == 0) lds_q = 0;
  FMXW_WBASES(w, LDSW);
  __syncthreads();
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane & (FMX_GROUP - 1);
  const uint32_t grp = lane >> 3;
  const uint32_t slot = g & (uint32_t)(Q - 1);        // the walk whose state this lane keeps (8 / Q replicas)
  const bool owner = g < (uint32_t)Q;
  const uint32_t olane = (lane & ~7u) | slot;
  constexpr unsigned long long SLOT0 = Q == 4 ? 0x1111111111111111ull : 0xFFFFFFFFFFFFFFFFull;
  constexpr uint64_t NONE = ~0ull;
  constexpr uint32_t NOCH = FmxwHitQueue::NOCHUNK;
  uint64_t *const out = io + blo;                     // the block's slice: rows on entry, positions on exit
  FmxwHitQueue hq;
  hq.init(out, bn, chunk, lane, lds_q);
  [[maybe_unused]] volatile fmx_lds_u64 *const ring = FMX_LDS_U64(wc_ring + (threadIdx.x >> 6) * (FMXW_WC_SLOTS * 64));
  [[maybe_unused]] volatile fmx_lds_u32 *const ring_tag = FMX_LDS_U32(wc_tag + (threadIdx.x >> 6) * FMXW_WC_SLOTS);
  [[maybe_unused]] uint32_t rs0 = 0, rs1 = 1, rseq = 2;
  if (WC) {
#pragma unroll
    for (uint32_t r = 0; r < FMXW_WC_SLOTS; r++) ring[r * 64u + lane] = NONE;
    if (lane < FMXW_WC_SLOTS) ring_tag[lane] = lane == 0 ? hq.c0 : (lane == 1 ? hq.c1 : NOCH);
  }
  uint32_t hx;
  uint64_t row;
  [[maybe_unused]] uint32_t myslot = 0;
  bool first0;
  bool active = hq.take((slot << 3) | grp, hx, row, first0);
  {
    const bool slid = hq.advance(8u * (uint32_t)Q, lds_q);
    if (WC && slid) {
      const uint32_t ns = rseq & (FMXW_WC_SLOTS - 1u);
      if (lane == 0) ring_tag[ns] = hq.c1;
      rseq++; rs0 = rs1; rs1 = ns;
    }
  }
  if (!active) row = 0;
  constexpr uint32_t FRESH = 0xFFu;
  uint32_t ctl = FRESH;                               // phase of the current row | steps of the whole walk << 8
  uint64_t fin = NONE, nsteps = 0;                    // index of the walk's sample once known
  for (;;) {
    if (!__any(active)) break;
    const bool done = active && fin != NONE;
    const unsigned long long fm = __ballot(done && owner);
    uint64_t fin_si = NONE;
    uint32_t fin_steps = 0, fin_x = 0;
    [[maybe_unused]] uint32_t fin_slot = 0;
    if (fm) {                                         // wave-uniform
      uint32_t x_new;
      uint64_t r_new;
      bool first;
      const bool ok = hq.take((uint32_t)__popcll(fm & ((1ull << olane) - 1ull)), x_new, r_new, first);
      if (done) {
        fin_si = fin;
        fin_steps = ctl >> 8;
        fin_x = hx;
        fin_slot = myslot;
        nsteps += ctl >> 8;
        hx = x_new;
        myslot = first ? rs0 : rs1;
        active = ok;
        row = ok ? r_new : 0;
        ctl = FRESH;
        fin = NONE;
      }
      const bool slid = hq.advance((uint32_t)__popcll(fm), lds_q);
      if (WC && slid) {
        const uint32_t ns = rseq & (FMXW_WC_SLOTS - 1u);
        const uint32_t old_tag = ring_tag[ns];
        const uint64_t v = ring[ns * 64u + lane];
        if (old_tag != NOCH && v != NONE) out[old_tag * 64u + lane] = v;
        ring[ns * 64u + lane] = NONE;
        if (lane == 0) ring_tag[ns] = hq.c1;
        rseq++; rs0 = rs1; rs1 = ns;
      }
    }
    uint64_t sa = 0;
    if (fin_si != NONE) sa = w.samples[fin_si];       // sample.rs:46-60 Some(sa)
    // (every row fmx_launch_expand64 wrote is a row of this index: refused ranges were replaced by row 0)
    FMX_CHECK(!active || row < w.n);
    const uint32_t rlo = active ? (uint32_t)row : 0xFFFFFFFFu, rhi = active ? (uint32_t)(row >> 32) : 0xFFFFFFFFu;
    const unsigned long long wm = __ballot(active);
    uint4 p[Q];
    uint32_t offq[Q];
    uint64_t recq[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      offq[q] = 0xFFFFFFFFu;
      recq[q] = 0;
      if (!(wm & (SLOT0 << q))) continue;
      const uint32_t lo = Q == 1 ? rlo : fmx_quad_bcast(rlo, q), hi = Q == 1 ? rhi : fmx_quad_bcast(rhi, q);
      if ((lo & hi) != 0xFFFFFFFFu) {                 // group-uniform (rows are below 2^38)
        recq[q] = fmx_walk_record64((uint64_t)hi << 32 | lo, offq[q]);
        FMX_CHECK(recq[q] < w.n / FMX_WALK_ROWS + 1u);
        p[q] = w.walk[(size_t)recq[q] * 8u + g];
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (!(wm & (SLOT0 << q))) continue;
      if (offq[q] != 0xFFFFFFFFu) {
        uint32_t sym, ph, si;
        const uint32_t nr = fmx_walk_step_rel(p[q], offq[q], g, sym, ph, si);
        const uint64_t b = (ph >= 2u && sym == 0u) ? 0ull : wbase_at(recq[q], fmxw_walk_counter(sym, ph));
        if (slot == (uint32_t)q) {
          if (ctl == FRESH) ctl = ph * 0x101u;
          FMX_CHECK(ph == (ctl & 0xFFu));
          if (ph <= 1u) {
            fin = b + si;                             // this row's sample (phase 0) or the next row's (phase 1)
          } else {                                    // None: i = lf_map(i); steps += 1   fm_index.rs:134-137
            row = b + nr;
            ctl--;
          }
        }
      }
    }
    if (fin_si != NONE && owner) {
      uint64_t v = sa + fin_steps;                    // fm_index.rs:131-133: (sa + steps) % len
      if (v >= w.n) v -= w.n;
      if (WC && ring_tag[fin_slot] == (fin_x >> 6)) ring[fin_slot * 64u + (fin_x & 63u)] = v;
      else out[fin_x] = v;
    }
  }
  if (WC) {
#pragma unroll
    for (uint32_t r = 0; r < FMXW_WC_SLOTS; r++) {
      const uint32_t tag = ring_tag[r];
      const uint64_t v = ring[r * 64u + lane];
      if (tag != NOCH && v != NONE) out[tag * 64u + lane] = v;
    }
  }
  if (steps_out && owner && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// ===========================================================================
// Generic wide indexes (FmxWideDev::generic: byte alphabets, up to FMXW_MAX_LEVELS wavelet levels).  Same shapes as the
// one-level kernels above -- a group per pattern / per walk --, one record round per level and interval end.
// The level bases and K[] are read with data-dependent indices in every step: LDS (GLDS = true, up to FMXW_GLDS_SB
// superblocks) or global memory, two instantiations as for FMXW_BASES.
// ===========================================================================
#define FMXW_GLDS_SB 8u
#define FMXW_GBASES(w, GLDS)                                                                                       \
  __shared__ uint64_t lds_gb[(GLDS) ? FMXW_MAX_LEVELS * FMXW_GLDS_SB * 16u : 1u];                                    \
  __shared__ uint64_t lds_k[(GLDS) ? 256u : 1u];                                                                    \
  const bool klds_ = (GLDS) && (w).max_character < 256u;      /* K[] of a byte alphabet fits; larger ones stay global */ \
  if (GLDS) {                                                                                                      \
    for (uint32_t l_ = 0; l_ < (w).nlevels; l_++)                                                                  \
      for (uint32_t t_ = threadIdx.x; t_ < (w).nsb * 16u; t_ += blockDim.x)                                        \
        lds_gb[l_ * FMXW_GLDS_SB * 16u + t_] = (w).lv[l_].base[t_];                                                 \
    if (klds_)                                                                                                     \
      for (uint32_t t_ = threadIdx.x; t_ <= (w).max_character; t_ += blockDim.x) lds_k[t_] = (w).K[t_];            \
    __syncthreads();                                                                                               \
  }                                                                                                                \
  auto gbase = [&](uint32_t l_, uint64_t pos_, uint32_t code_) -> uint64_t {                                       \
    const uint32_t sb_ = (uint32_t)(pos_ >> (w).sb_shift);                                                         \
    if constexpr (GLDS) return lds_gb[(l_ * FMXW_GLDS_SB + sb_) * 16u + code_];                                     \
    else return (w).lv[l_].base[(size_t)sb_ * 16u + code_];                                                         \
  };                                                                                                               \
  auto gk = [&](uint32_t c_) -> uint64_t {                                                                         \
    if constexpr (GLDS) { if (klds_) return lds_k[c_]; }                                                            \
    return (w).K[c_];                                                                                              \
  }

// this lane's piece of the record of position `pos` on level L
__device__ __forceinline__ uint4 fmxw_g_piece(const FmxWideLevel &L, uint64_t pos, uint32_t g) {
  FMX_CHECK((pos >> (L.fmt == 3 ? 8 : 7)) < L.nrec);
  return L.rec[(size_t)(pos >> (L.fmt == 3 ? 8 : 7)) * 8u + g];
}
// rank of `code` among the first `pos` entries of the level, relative to the superblock start (all 8 lanes get it)
__device__ __forceinline__ uint32_t fmxw_g_rank32(const FmxWideLevel &L, const uint4 &p, uint64_t pos, uint32_t code, uint32_t g) {
  return L.fmt == 3 ? fmx_group_sum(fmx_piece_rank<3>(p, (uint32_t)pos & 255u, code, g))
                    : fmx_group_sum(fmx_piece_rank<4>(p, (uint32_t)pos & 127u, code, g));
}
// level code of the entry at `pos` (all 8 lanes get it)
__device__ __forceinline__ uint32_t fmxw_g_code(const FmxWideLevel &L, const uint4 &p, uint64_t pos, uint32_t g) {
  if (L.fmt == 3) {
    const uint32_t off = (uint32_t)pos & 255u;
    return fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p, off & 31u) : 0u);
  }
  const uint32_t off = (uint32_t)pos & 127u;
  return fmx_group_sum((g == (off >> 4)) ? fmx_piece_code<4>(p, off & 15u) : 0u);
}
// rank chain of symbol c at position i: lf_map2(c, i) = K[c] + chain                          fm_index.rs:93-95
template <class GB>
__device__ __forceinline__ uint64_t fmxw_g_chain(const FmxWideDev &w, const GB &gbase, uint32_t c, uint64_t pos, uint32_t g) {
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    const uint32_t code = (c >> L.shift) & L.mask;
    const uint4 p = fmxw_g_piece(L, pos, g);
    pos = gbase(l, pos, code) + fmxw_g_rank32(L, p, pos, code, g);
  }
  return pos;
}
// get_l(i) and the rank chain of that symbol at i: lf_map(i) = K[sym] + chain                 fm_index.rs:82-91
template <class GB>
__device__ __forceinline__ uint64_t fmxw_g_lf(const FmxWideDev &w, const GB &gbase, uint64_t pos, uint32_t g, uint32_t &sym) {
  sym = 0;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    const uint4 p = fmxw_g_piece(L, pos, g);
    const uint32_t code = fmxw_g_code(L, p, pos, g);
    sym |= code << L.shift;
    pos = gbase(l, pos, code) + fmxw_g_rank32(L, p, pos, code, g);
  }
  return pos;
}
// FMIndexMultiPiecesBackend::lf_map2 / lf_map for the end marker (multi_pieces.rs:131-137, 147-153): the end markers are
// ordered by piece, the LAST one (row first_row = sa_idx_first_text) maps to row 0; rank0 = cs[0] + rank of 0 at i
__device__ __forceinline__ uint64_t fmxw_multi_zero(const FmxWideDev &w, uint64_t i, uint64_t rank0) {
  return i < w.first_row ? rank0 + 1u : (i == w.first_row ? 0ull : rank0);
}
// get_l(i) and lf_map(i) of an FM (MP = false) or multi-pieces (MP = true) generic index
template <bool MP, class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_g_lf_map(const FmxWideDev &w, const GB &gbase, const GK &gk, uint64_t i, uint32_t g,
                                                  uint32_t &sym) {
  uint64_t r = fmxw_g_lf(w, gbase, i, g, sym);
  r += gk(sym);                                                   // fm_index.rs:86-91
  if (MP && sym == 0u) r = fmxw_multi_zero(w, i, r);
  return r;
}
// counter of `code` at the start of record r, absolute (in the units of the level's ranks); global bases
__device__ __forceinline__ uint64_t fmxw_g_counter(const FmxWideDev &w, const FmxWideLevel &L, uint32_t r, uint32_t code) {
  const uint32_t recs = w.sb_shift - (L.fmt == 3 ? 8u : 7u);
  const uint64_t b = L.base[(size_t)(r >> recs) * 16u + code];
  if (L.fmt == 3) return b + L.rec[(size_t)r * 8u + code].x;
  const uint4 p = L.rec[(size_t)r * 8u + (code >> 1)];
  return b + ((code & 1u) ? p.y : p.x);
}
// position of the entry with level code `code` whose rank (in the level's units) is `target`.  Binary searches over
// the superblock bases and the record counters: the extract path is not the hot path.
__device__ __forceinline__ uint64_t fmxw_g_select(const FmxWideDev &w, const FmxWideLevel &L, uint32_t code, uint64_t target,
                                                  uint32_t g) {
  const uint32_t recs = w.sb_shift - (L.fmt == 3 ? 8u : 7u);
  uint32_t slo = 0, shi = w.nsb - 1u;
  while (slo < shi) {                               // last superblock whose base <= target
    const uint32_t mid = slo + (shi - slo + 1u) / 2u;
    if (L.base[(size_t)mid * 16u + code] <= target) slo = mid; else shi = mid - 1u;
  }
  uint32_t lo = slo << recs, hi = ((slo + 1u) << recs) - 1u;
  if (hi > L.nrec - 1u) hi = L.nrec - 1u;
  while (lo < hi) {                                 // last record of it whose counter <= target
    const uint32_t mid = lo + (hi - lo + 1u) / 2u;
    if (fmxw_g_counter(w, L, mid, code) <= target) lo = mid; else hi = mid - 1u;
  }
  const uint32_t rem = (uint32_t)(target - fmxw_g_counter(w, L, lo, code));   // rem-th match inside the record
  const uint4 p = L.rec[(size_t)lo * 8u + g];
  const uint32_t m = L.fmt == 3 ? fmx_piece_match<3>(p, code) : fmx_piece_match<4>(p, code);
  const uint32_t per = L.fmt == 3 ? 32u : 16u;
  const uint32_t mine = __popc(m);
  uint32_t before = 0;
#pragma unroll
  for (uint32_t q = 0; q < FMX_GROUP; q++) {
    const uint32_t cq = fmx_group_sum(g == q ? mine : 0u);
    before += (q < g) ? cq : 0u;
  }
  const bool here = rem >= before && rem < before + mine;
  const uint32_t pos = here ? g * per + fmx_select32(m, rem - before) : 0u;
  return (uint64_t)lo * (per * 8u) + fmx_group_sum(pos);
}
// get_f(i) and fl_map(i) (fm_index.rs:97-120): the greatest c with cs[c] <= i, then the (i - cs[c])-th c of the BWT:
// down the start chain of c, back up through the level selects (as fmx_mwm_select)
template <class GB>
__device__ __forceinline__ uint64_t fmxw_g_fl(const FmxWideDev &w, const GB &gbase, uint64_t i, uint32_t g, uint32_t &sym) {
  uint32_t s = 0, e = w.max_character + 1u;
  while (e - s > 1u) {
    const uint32_t m = s + (e - s) / 2u;
    if (w.cs[m] <= i) s = m; else e = m;
  }
  sym = s;
  uint64_t target = fmxw_g_chain(w, gbase, s, 0, g) + (i - w.cs[s]);
  for (uint32_t l = w.nlevels; l-- > 0;) {
    const FmxWideLevel &L = w.lv[l];
    target = fmxw_g_select(w, L, (s >> L.shift) & L.mask, target, g);
  }
  return target;
}

// ===========================================================================
// RLFMIndex on the wide engine (FmxWideDev::kind == FMX_KIND_RLFM; rlfmi.rs:15-24, 122-190): S = the generic levels
// above over the run heads, B / B' = FmxWideBits, rows and run indices 64 bits wide.  Group-uniform code, like the
// generic kernels: every lane of a group holds the same row.
// ===========================================================================
#define FMXW_NONE (~0ull)
__device__ __forceinline__ uint64_t fmxw_div3(uint64_t x) { return __umul64hi(x, 0xAAAAAAAAAAAAAAABull) >> 1; }

// the record that holds bit i: this lane's piece, the superblock's base, and where i sits in the record
struct FmxwBitsRec { uint4 pc; uint64_t rec, base; uint32_t p, bit; };
__device__ __forceinline__ FmxwBitsRec fmxw_bits_load(const FmxWideBits &bv, uint64_t i, uint32_t g) {
  FmxwBitsRec r;
  r.rec = fmxw_div3(i >> 8);                                  // i / 768
  const uint32_t within = (uint32_t)(i - r.rec * FMX_BITS_PER_REC);
  r.p = fmx_div3(within >> 5);                                // within / 96
  r.bit = within - r.p * FMX_BITS_PER_PIECE;
  FMX_CHECK(r.rec < bv.nrec && (r.rec >> bv.sb_shift) < bv.nsb);
  r.pc = bv.rec[(size_t)r.rec * 8u + g];
  r.base = bv.base[r.rec >> bv.sb_shift];
  return r;
}
// rank1(i) (clamped like vers-vecs RsVec::rank1), the bit B[i] (0 past the end) and `next` = the position of the first
// one at or after i when it lies in the record just loaded (FMXW_NONE otherwise) -- that position is select1(rank1(i)),
// the run start the RLFM formulas subtract (rlfmi.rs:132, 141)
__device__ __forceinline__ uint64_t fmxw_bits_rank_next(const FmxWideBits &bv, uint64_t i, uint32_t g, uint32_t &bit_i,
                                                        uint64_t &next) {
  if (i > bv.len) i = bv.len;
  const FmxwBitsRec r = fmxw_bits_load(bv, i, g);
  const uint32_t bit = r.bit;
  const uint32_t m0 = fmx_lowmask(bit < 32u ? bit : 32u);
  const uint32_t m1 = bit > 32u ? fmx_lowmask(bit - 32u < 32u ? bit - 32u : 32u) : 0u;
  const uint32_t m2 = bit > 64u ? fmx_lowmask(bit - 64u) : 0u;
  const uint32_t c = __popc(r.pc.y & m0) + __popc(r.pc.z & m1) + __popc(r.pc.w & m2);
  const uint32_t word = bit < 32u ? r.pc.y : (bit < 64u ? r.pc.z : r.pc.w);
  const uint32_t mine = (g == r.p) ? 1u : 0u;
  bit_i = fmx_group_sum(mine * ((word >> (bit & 31u)) & 1u));
  uint32_t y = r.pc.y, z = r.pc.z, ww = r.pc.w;
  if (g == r.p) { y &= ~m0; z &= ~m1; ww &= ~m2; }
  else if (g < r.p) { y = 0u; z = 0u; ww = 0u; }
  uint32_t cand = 0xFFFFFFFFu;
  if (y) cand = (uint32_t)__builtin_ctz(y);
  else if (z) cand = 32u + (uint32_t)__builtin_ctz(z);
  else if (ww) cand = 64u + (uint32_t)__builtin_ctz(ww);
  if (cand != 0xFFFFFFFFu) cand += g * FMX_BITS_PER_PIECE;
  cand = fmx_group_min(cand);
  next = cand != 0xFFFFFFFFu ? r.rec * FMX_BITS_PER_REC + cand : FMXW_NONE;
  return r.base + fmx_group_sum(mine * (r.pc.x + c));
}
// rank1(i + 1) - 1 = the index of the last one at or before i (the run that holds row i), and `prev` = that one's
// position when it lies in the record just loaded (FMXW_NONE otherwise: the run began before the record)
__device__ __forceinline__ uint64_t fmxw_bits_rank_prev(const FmxWideBits &bv, uint64_t i, uint32_t g, uint64_t &prev) {
  if (i >= bv.len) i = bv.len ? bv.len - 1u : 0u;
  const FmxwBitsRec r = fmxw_bits_load(bv, i, g);
  const uint32_t b1 = r.bit + 1u;                             // bits [0, bit] of piece p
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  const uint32_t c = __popc(r.pc.y & m0) + __popc(r.pc.z & m1) + __popc(r.pc.w & m2);
  uint32_t y = r.pc.y, z = r.pc.z, ww = r.pc.w;
  if (g == r.p) { y &= m0; z &= m1; ww &= m2; }
  else if (g > r.p) { y = 0u; z = 0u; ww = 0u; }
  uint32_t cand = 0u;                                         // position in the record + 1; 0 = none (max over the group)
  if (ww) cand = 96u - (uint32_t)__builtin_clz(ww);
  else if (z) cand = 64u - (uint32_t)__builtin_clz(z);
  else if (y) cand = 32u - (uint32_t)__builtin_clz(y);
  if (cand) cand += g * FMX_BITS_PER_PIECE;
  uint32_t mx = cand;
  mx = max(mx, fmx_dpp_xor1(mx));
  mx = max(mx, fmx_dpp_xor2(mx));
  mx = max(mx, fmx_dpp_half_mirror(mx));
  prev = mx ? r.rec * FMX_BITS_PER_REC + (mx - 1u) : FMXW_NONE;
  return r.base + fmx_group_sum((g == r.p) ? r.pc.x + c : 0u) - 1u;
}
// select1(k): position of the k-th one (0-based); len when k >= #ones (vers-vecs RsVec::select1)
__device__ __forceinline__ uint64_t fmxw_bits_select(const FmxWideBits &bv, uint64_t k, uint32_t g) {
  if (k >= bv.ones) return bv.len;
  if (bv.pos) return bv.pos[k];                               // sparse vector: the positions are stored
  const uint64_t h = k / FMX_SEL_STEP;
  FMX_CHECK(h + 1 < bv.nsel);
  uint32_t lo = bv.sel[h], hi = bv.sel[h + 1];
  FMX_CHECK(lo < bv.nrec && hi < bv.nrec);
  while (lo < hi) {                                           // last record whose count <= k (group-uniform)
    const uint32_t mid = lo + (hi - lo + 1u) / 2u;
    if (bv.base[mid >> bv.sb_shift] + bv.rec[(size_t)mid * 8u].x <= k) lo = mid; else hi = mid - 1u;
  }
  const uint4 pc = bv.rec[(size_t)lo * 8u + g];
  const uint64_t ab = bv.base[lo >> bv.sb_shift] + pc.x;      // ones before this lane's piece
  const uint32_t p = fmx_group_sum(ab <= k ? 1u : 0u) - 1u;   // last piece whose count <= k
  const uint32_t rem = (uint32_t)(k - ab);                    // meaningful on lane p only
  const uint32_t c0 = __popc(pc.y), c1 = __popc(pc.z);
  uint32_t pos;
  if (rem < c0) pos = fmx_select32(pc.y, rem);
  else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
  else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
  pos = fmx_group_sum((g == p) ? pos : 0u);
  return (uint64_t)lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
}

// RLFMIndexBackend::lf_map2 for both ends of an interval (rlfmi.rs:135-143), staged: B ranks, then one rank chain per
// end at lo = b.rank1(i + 1) - 1 (the run holding row i) that yields s.rank(lo, c) and m = [s[lo] == c] from the same
// records -- s.rank(j, c) with j = b.rank1(i) in {lo, lo + 1} is s.rank(lo, c) + (j > lo ? m : 0), and get_l(i) == c is m
// (rlfmi.rs:137-138) -- then the B' selects (rlfmi.rs:139) and, when the row's symbol is c, the run start (rlfmi.rs:141).
template <class GB, class GK>
__device__ __forceinline__ void fmxw_r_lf_map2_pair(const FmxWideDev &w, const GB &gbase, const GK &gk, uint32_t c,
                                                    uint64_t &s, uint64_t &e, uint32_t g) {
  uint32_t bs, be;
  uint64_t nxs, nxe;
  const uint64_t js = fmxw_bits_rank_next(w.b, s, g, bs, nxs);    // b.rank1(i)            rlfmi.rs:136
  const uint64_t je = fmxw_bits_rank_next(w.b, e, g, be, nxe);
  uint64_t ps = js - 1u + bs, pe = je - 1u + be;                  // b.rank1(i + 1) - 1    rlfmi.rs:124
  uint32_t ms = 1u, me = 1u;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    const uint32_t code = (c >> L.shift) & L.mask;
    const uint4 pa = fmxw_g_piece(L, ps, g), pb = fmxw_g_piece(L, pe, g);
    const uint64_t ba = gbase(l, ps, code), bb = gbase(l, pe, code);
    ms &= fmxw_g_code(L, pa, ps, g) == code ? 1u : 0u;
    me &= fmxw_g_code(L, pb, pe, g) == code ? 1u : 0u;
    ps = ba + fmxw_g_rank32(L, pa, ps, code, g);
    pe = bb + fmxw_g_rank32(L, pb, pe, code, g);
  }
  const uint64_t kc = gk(c);
  const uint64_t nrs = kc + ps + (bs ? 0u : ms), nre = kc + pe + (be ? 0u : me);   // cs[c] + s.rank(j, c)   rlfmi.rs:137,139
  uint64_t fs = fmxw_bits_select(w.bp, nrs, g), fe = fmxw_bits_select(w.bp, nre, g);
  if (ms) fs = fs + s - (nxs != FMXW_NONE ? nxs : fmxw_bits_select(w.b, js, g));   // + i - b.select1(j)   rlfmi.rs:141
  if (me) fe = fe + e - (nxe != FMXW_NONE ? nxe : fmxw_bits_select(w.b, je, g));
  s = fs;
  e = fe;
}
// RLFMIndexBackend::get_l + lf_map (rlfmi.rs:122-133): one access + rank chain at lo = b.rank1(i + 1) - 1 gives
// c = s[lo] and s.rank(lo, c); s.rank(j, c) = that + (j - lo)
template <class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_r_lf(const FmxWideDev &w, const GB &gbase, const GK &gk, uint64_t i, uint32_t g,
                                              uint32_t &sym) {
  uint32_t bit;
  uint64_t nx;
  const uint64_t j = fmxw_bits_rank_next(w.b, i, g, bit, nx);     // b.rank1(i)
  const uint64_t r = fmxw_g_lf(w, gbase, j - 1u + bit, g, sym);   // s[lo], rank chain of it at lo
  const uint64_t nr = gk(sym) + r + (bit ? 0u : 1u);              // cs[c] + s.rank(j, c)     rlfmi.rs:129-130
  const uint64_t f = fmxw_bits_select(w.bp, nr, g);
  const uint64_t st = nx != FMXW_NONE ? nx : fmxw_bits_select(w.b, j, g);
  return f + i - st;                                              // rlfmi.rs:132
}
// lf_map(i) for a walk that has no use for the symbol (get_sa, rlfmi.rs:183-186): through the run table when the index
// has one -- lf_map(i) = lfrun[run of i] + (i - start of that run) -- the B record of the row + one table entry
template <class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_r_lf_step(const FmxWideDev &w, const GB &gbase, const GK &gk, uint64_t i, uint32_t g) {
  if (w.lfrun) {
    uint64_t st;
    const uint64_t lo = fmxw_bits_rank_prev(w.b, i, g, st);
    FMX_CHECK(lo < w.b.ones);
    const uint64_t f = w.lfrun[lo];
    if (st == FMXW_NONE) st = fmxw_bits_select(w.b, lo, g);       // group-uniform
    return f + i - st;
  }
  uint32_t sym;
  return fmxw_r_lf(w, gbase, gk, i, g, sym);
}
// RLFMIndexBackend::get_f + fl_map (rlfmi.rs:145-169)
template <class GB>
__device__ __forceinline__ uint64_t fmxw_r_fl(const FmxWideDev &w, const GB &gbase, uint64_t i, uint32_t g, uint32_t &sym) {
  uint64_t p;
  const uint64_t j = fmxw_bits_rank_prev(w.bp, i, g, p);          // bp.rank1(i + 1) - 1
  uint32_t s = 0, e = w.max_character + 1u;
  while (e - s > 1u) {                                            // the greatest c with cs[c] <= j
    const uint32_t m = s + (e - s) / 2u;
    if (w.cs[m] <= j) s = m; else e = m;
  }
  sym = s;
  if (p == FMXW_NONE) p = fmxw_bits_select(w.bp, j, g);           // bp.select1(j)
  uint64_t target = fmxw_g_chain(w, gbase, s, 0, g) + (j - w.cs[s]);   // s.select(j - cs[c], c)
  for (uint32_t l = w.nlevels; l-- > 0;) {
    const FmxWideLevel &L = w.lv[l];
    target = fmxw_g_select(w, L, (s >> L.shift) & L.mask, target, g);
  }
  return fmxw_bits_select(w.b, target, g) + i - p;                // b.select1(m) + i - bp.select1(j)
}

// ---- text-order sampling of a wide RLFM index (FmxWideDev::phase) ----
// piece holding `row` and the row's index inside it; rows per piece = 3 * floor(32 / level)
__device__ __forceinline__ uint64_t fmxw_phase_piece(uint64_t row, uint32_t level, uint32_t &t) {
  uint64_t p;
  if (level == 1) { p = fmxw_div3(row >> 5); t = (uint32_t)(row - p * 96u); }
  else if (level == 2) { p = fmxw_div3(row >> 4); t = (uint32_t)(row - p * 48u); }
  else if (level == 3) { p = __umul64hi(row >> 1, 0x8888888888888889ull) >> 3; t = (uint32_t)(row - p * 30u); }   // / 15
  else { p = fmxw_div3(row >> 3); t = (uint32_t)(row - p * 24u); }
  return p;
}
// phase of `row` and the index of its sample when the phase is 0 (one lane, one 16-byte piece)
__device__ __forceinline__ uint32_t fmxw_phase_probe(const FmxWideDev &w, uint64_t row, uint64_t &rank0) {
  uint32_t t, r0;
  const uint64_t pi = fmxw_phase_piece(row, w.sa_level, t);
  FMX_CHECK((pi >> w.psb_shift) < w.npsb);
  const uint32_t ph = fmx_phase_decode(w.phase[pi], t, w.sa_level, r0);
  rank0 = w.pbase[pi >> w.psb_shift] + r0;
  return ph;
}

// get_sa of one row of a text-order RLFM index: SA[row] mod 2^level LF steps (`lf`: row -> lf_map(row))   rlfmi.rs:172-190
template <class LF>
__device__ __forceinline__ uint64_t fmxw_r_get_sa_text(const FmxWideDev &w, const LF &lf, uint64_t row, uint64_t &steps_out) {
  uint64_t r0;
  const uint32_t ph = fmxw_phase_probe(w, row, r0);
  for (uint32_t t = 0; t < ph; t++) row = lf(row);
  if (ph) {
    [[maybe_unused]] const uint32_t p2 = fmxw_phase_probe(w, row, r0);
    FMX_CHECK(p2 == 0u);
  }
  steps_out = ph;
  uint64_t v = w.samples[r0] + ph;                  // (sa + steps) % len
  if (v >= w.n) v -= w.n;
  return v;
}
__global__ __launch_bounds__(64) void fmxw_g_compute_K_kernel(FmxWideDev w, uint64_t *__restrict__ K) {
  FMXW_GBASES(w, false);
  (void)gk;
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  if (c > w.max_character) return;
  const uint64_t sc = fmxw_g_chain(w, gbase, c, 0, g);
  if (g == 0) K[c] = w.cs[c] - sc;
}

// SearchWrapper::search for a batch (wrapper.rs:103-124): a group per pattern; per level the records of both
// interval ends are requested together, the next pattern symbol with the first level's
template <bool GLDS, int KD>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_count_kernel(
    FmxWideDev w, const void *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  constexpr bool RL = KD == FMX_KIND_RLFM, MP = KD == FMX_KIND_MULTI;
  FMXW_GBASES(w, GLDS);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t ptot = npat ? off[npat] : 0;       // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  uint64_t nsteps = 0;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    const uint64_t pbeg = off[k], pend = off[k + 1];
    uint64_t j = pend - pbeg;
    bool bad = pend < pbeg || pbeg < pmin || pend > ptot;          // offsets that go backwards or leave the pattern buffer
    uint64_t s = 0, e = w.n;                        // SearchIndexWrapper::search: (0, len)   wrapper.rs:41
    if (s0e0) {                                     // Search::search on an existing Search   wrapper.rs:105-106
      s = s0e0[2 * k];
      e = s0e0[2 * k + 1];
      bad |= s > w.n || e > w.n;                    // not a range of this index
    }
    if (bad) {                                      // refuse, do not read
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
      s = 0; e = 0; j = 0;
    }
    uint32_t c = j ? fmx_load_sym(pat, w.sym_bytes, pbeg + j - 1) : 0u;        // for c in pattern.iter().rev()          wrapper.rs:108
    while (j) {
      if (c > w.max_character) {                    // reference: panic on cs[c]
        if (g == 0) atomicOr(w.status, 1u << FMX_ERR_SYMBOL_RANGE);
        s = 0; e = 0;
        break;
      }
      uint32_t cn = 0;
      if constexpr (RL) {                           // RLFMIndexBackend::lf_map2             rlfmi.rs:135-143
        if (j > 1) cn = fmx_load_sym(pat, w.sym_bytes, pbeg + j - 2);
        fmxw_r_lf_map2_pair(w, gbase, gk, c, s, e, g);                     // wrapper.rs:109-110
        c = cn;
        j--;
        nsteps++;
        if (s == e) break;                          // wrapper.rs:111-113
        continue;
      }
      uint64_t ps = s, pe = e;
      for (uint32_t l = 0; l < w.nlevels; l++) {
        const FmxWideLevel &L = w.lv[l];
        const uint32_t code = (c >> L.shift) & L.mask;
        const uint4 pa = fmxw_g_piece(L, ps, g), pb = fmxw_g_piece(L, pe, g);
        if (l == 0 && j > 1) cn = fmx_load_sym(pat, w.sym_bytes, pbeg + j - 2);             // rides along with the record loads
        const uint64_t ba = gbase(l, ps, code), bb = gbase(l, pe, code);
        ps = ba + fmxw_g_rank32(L, pa, ps, code, g);
        pe = bb + fmxw_g_rank32(L, pb, pe, code, g);
      }
      const uint64_t kc = gk(c);
      if (MP && c == 0u) {                          // multi_pieces.rs:147-153
        s = fmxw_multi_zero(w, s, kc + ps);
        e = fmxw_multi_zero(w, e, kc + pe);
      } else {
        s = kc + ps;                                // wrapper.rs:109
        e = kc + pe;                                // wrapper.rs:110
      }
      c = cn;
      j--;
      nsteps++;
      if (s == e) break;                            // wrapper.rs:111-113
    }
    if (g == 0) {
      if (out_s) out_s[k] = s;
      if (out_e) out_e[k] = e;
      if (out_cnt) out_cnt[k] = e - s;              // wrapper.rs:132-134
    }
  }
  if (steps_out && g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// get_sa for a batch of rows (fm_index.rs:127-140; sample.rs:46-60): a group per walk, rows in, positions out
template <bool GLDS, int KD>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_walk_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                  uint64_t *__restrict__ steps_out) {
  constexpr bool RL = KD == FMX_KIND_RLFM, MP = KD == FMX_KIND_MULTI;
  FMXW_GBASES(w, GLDS);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t lmask = (1ull << w.sa_level) - 1ull;
  uint64_t nsteps = 0;
  for (uint64_t h = gid; h < total; h += ngroups) {
    uint64_t row = io[h], steps = 0, v = ~0ull;
    if (row >= w.n) {                               // refuse, do not read
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
    } else if (w.phase) {                           // text-order samples (an RLFM index without the run table lands here)
      if constexpr (RL) {
        v = fmxw_r_get_sa_text(w, [&](uint64_t r) { return fmxw_r_lf_step(w, gbase, gk, r, g); }, row, steps);
      } else {
        v = fmxw_r_get_sa_text(w, [&](uint64_t r) { uint32_t sy; return fmxw_g_lf_map<MP>(w, gbase, gk, r, g, sy); }, row, steps);
      }
    } else {
      while (row & lmask) {                         // None: i = lf_map(i); steps += 1        fm_index.rs:134-137, rlfmi.rs:183-186
        if constexpr (RL) {
          row = fmxw_r_lf_step(w, gbase, gk, row, g);
        } else {
          uint32_t sym;
          row = fmxw_g_lf_map<MP>(w, gbase, gk, row, g, sym);
        }
        steps++;
      }
      v = w.samples[row >> w.sa_level] + steps;     // Some(sa): (sa + steps) % len           fm_index.rs:131-133
      if (v >= w.n) v -= w.n;
    }
    if (g == 0) io[h] = v;
    nsteps += steps;
  }
  if (steps_out && g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// get_sa for a batch of rows on a generic FM / multi-pieces index with TEXT-ORDER samples (FmxWideDev::phase; round 4): a
// group owns Q consecutive hits at a time and takes them through the walk together -- start probes, then LF steps
// level by level, final probes, samples -- with the Q requests of every stage in flight before the first is consumed
// (one walk per group, the shape of fmxw_g_walk_kernel, keeps a single line in flight per group: 32 % of the
// request ceiling).  A walk is SA[row] mod 2^level steps long, so the Q walks of a group differ by < 2^level steps.
template <bool GLDS, bool MP>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_walk_text_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                       uint64_t *__restrict__ steps_out) {
  FMXW_GBASES(w, GLDS);
  constexpr int Q = 4;
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  uint64_t nsteps = 0;
  for (uint64_t h0 = gid * Q; h0 < total; h0 += ngroups * Q) {
    uint64_t row[Q], r0[Q];
    uint32_t ph[Q], tt[Q];
    uint4 pp[Q];
    bool ok[Q], bad[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) row[q] = h0 + q < total ? io[h0 + q] : ~0ull;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      bad[q] = h0 + q < total && row[q] >= w.n;     // not a row of this index: refuse, do not read
      ok[q] = h0 + q < total && !bad[q];
      if (bad[q] && g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
      if (!ok[q]) row[q] = 0;
      const uint64_t pi = fmxw_phase_piece(row[q], w.sa_level, tt[q]);
      FMX_CHECK((pi >> w.psb_shift) < w.npsb);
      pp[q] = w.phase[pi];
      r0[q] = w.pbase[pi >> w.psb_shift];
    }
    uint32_t maxph = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      uint32_t rel;
      ph[q] = fmx_phase_decode(pp[q], tt[q], w.sa_level, rel);
      if (!ok[q]) ph[q] = 0;
      r0[q] += rel;
      maxph = ph[q] > maxph ? ph[q] : maxph;
      nsteps += ph[q];
    }
    for (uint32_t t = 0; t < maxph; t++) {          // None: i = lf_map(i); steps += 1        fm_index.rs:134-137
      uint64_t pos[Q];
      uint32_t sym[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { pos[q] = row[q]; sym[q] = 0; }
      for (uint32_t l = 0; l < w.nlevels; l++) {
        const FmxWideLevel &L = w.lv[l];
        uint4 p[Q];
#pragma unroll
        for (int q = 0; q < Q; q++)
          if (t < ph[q]) p[q] = fmxw_g_piece(L, pos[q], g);
#pragma unroll
        for (int q = 0; q < Q; q++) {
          if (t < ph[q]) {
            const uint32_t code = fmxw_g_code(L, p[q], pos[q], g);
            sym[q] |= code << L.shift;
            pos[q] = gbase(l, pos[q], code) + fmxw_g_rank32(L, p[q], pos[q], code, g);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < Q; q++) {
        if (t < ph[q]) {
          uint64_t r = gk(sym[q]) + pos[q];         // fm_index.rs:86-91
          if (MP && sym[q] == 0u) r = fmxw_multi_zero(w, row[q], r);
          row[q] = r;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {                   // the final rows are phase-0 rows: their samples' indices
      if (ph[q]) {
        const uint64_t pi = fmxw_phase_piece(row[q], w.sa_level, tt[q]);
        FMX_CHECK((pi >> w.psb_shift) < w.npsb);
        pp[q] = w.phase[pi];
        r0[q] = w.pbase[pi >> w.psb_shift];
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (ph[q]) {
        uint32_t rel;
        [[maybe_unused]] const uint32_t p2 = fmx_phase_decode(pp[q], tt[q], w.sa_level, rel);
        FMX_CHECK(p2 == 0u);
        r0[q] += rel;
      }
    }
    uint64_t sv[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) sv[q] = ok[q] ? w.samples[r0[q]] : 0ull;       // Some(sa)   fm_index.rs:131
#pragma unroll
    for (int q = 0; q < Q; q++) {
      uint64_t v = sv[q] + ph[q];                   // (sa + steps) % len                      fm_index.rs:132
      if (v >= w.n) v -= w.n;
      if (g == 0 && (ok[q] || bad[q])) io[h0 + q] = ok[q] ? v : ~0ull;
    }
  }
  if (steps_out && g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// select1(k) by ONE lane (k < ones): the stored position, or hint + search over the record counts + the pieces of the record
__device__ __forceinline__ uint64_t fmxw_bits_lane_select(const FmxWideBits &bv, uint64_t k) {
  if (bv.pos) return bv.pos[k];
  const uint64_t h = k / FMX_SEL_STEP;
  uint32_t lo = bv.sel[h], hi = bv.sel[h + 1];
  while (lo < hi) {                                 // last record whose count <= k
    const uint32_t mid = lo + (hi - lo + 1u) / 2u;
    if (bv.base[mid >> bv.sb_shift] + bv.rec[(size_t)mid * 8u].x <= k) lo = mid; else hi = mid - 1u;
  }
  const uint32_t rel = (uint32_t)(k - bv.base[lo >> bv.sb_shift]);
  uint32_t p = 0;
  for (uint32_t q = 1; q < 8u; q++)                 // last piece of it whose count <= k (the record is one line)
    if (bv.rec[(size_t)lo * 8u + q].x <= rel) p = q;
  const uint4 pc = bv.rec[(size_t)lo * 8u + p];
  const uint32_t rem = rel - pc.x, c0 = __popc(pc.y), c1 = __popc(pc.z);
  uint32_t pos;
  if (rem < c0) pos = fmx_select32(pc.y, rem);
  else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
  else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
  return (uint64_t)lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
}
// get_sa for a batch of rows on an RLFM index that has the run table (rlfmi.rs:172-190): a LANE per walk -- every
// probe of a step is a lane-wise request (the 16-byte B piece of the row: run index and, unless the run began before the
// piece, its start; then the table entry), 64 independent walks per wave instruction and no cross-lane traffic.  A lane
// whose walk ends writes its position and takes its next hit (hits tid, tid + threads, ...) while its neighbours walk on.
// LOCKSTEP (intervals of many rows -- a repetitive text's patterns): a wave takes 64 CONSECUTIVE hits and walks them to
// the end together before it takes the next 64.  Neighbouring rows of an interval sit in the same runs and stay
// neighbours under LF, so the wave's 64 requests fall into a few lines -- where the refilling shape scatters them
// over the index after its first iterations and runs at the memory system's request ceiling (2 per step).
// lf_map(row) through the run table by ONE lane: the B piece of the row (run index and, unless the run began before the
// piece, its start), then the table entry                                                   rlfmi.rs:127-133
__device__ __forceinline__ uint64_t fmxw_r_lane_lf(const FmxWideDev &w, uint64_t row) {
  const FmxWideBits &bv = w.b;
  const uint64_t pidx = fmxw_div3(row >> 5);        // row / 96
  const uint32_t b1 = (uint32_t)(row - pidx * FMX_BITS_PER_PIECE) + 1u;     // bits [0, bit] of the piece
  FMX_CHECK((pidx >> 3) < bv.nrec);
  const uint4 pc = bv.rec[pidx];
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  const uint32_t y = pc.y & m0, z = pc.z & m1, ww = pc.w & m2;
  const uint64_t lo = bv.base[(pidx >> 3) >> bv.sb_shift] + pc.x + __popc(y) + __popc(z) + __popc(ww) - 1u;   // run of the row
  FMX_CHECK(lo < bv.ones);
  const uint64_t f = w.lfrun[lo];                   // lf_map(first row of the run)
  uint64_t st;                                      // its first row: the last one at or before the row
  if (ww) st = pidx * FMX_BITS_PER_PIECE + 95u - (uint32_t)__builtin_clz(ww);
  else if (z) st = pidx * FMX_BITS_PER_PIECE + 63u - (uint32_t)__builtin_clz(z);
  else if (y) st = pidx * FMX_BITS_PER_PIECE + 31u - (uint32_t)__builtin_clz(y);
  else st = fmxw_bits_lane_select(bv, lo);
  return f + row - st;
}
// ---- RLFM count with an interval ENDPOINT per lane (round 5; the shape of the 32-bit engine's fmx_count_ep_kernel, fmx_ep.h) ----
// The group-per-pattern kernel above keeps one or two lines in flight per group through four dependent stages (B, S level
// 0, S level 1, B' select).  Here lane 2q / 2q+1 of a group own the s / e end of the group's q-th pattern: the probes that
// need one 16-byte piece (rank1 on B with its superblock base, the selects through the stored positions) are lane-wise --
// 64 independent requests per wave instruction --, the ranks over the 128-byte records of S go in rounds: for q = 0..7
// the group broadcasts endpoint q's (record, offset | code), all 8 lanes load their piece of it, and the popcounts are
// reduced with three DPP adds and handed back to lane q, which adds the 64-bit base of ITS superblock.
__device__ __forceinline__ uint32_t fmxw_grp_bcast(uint32_t v, uint32_t base, uint32_t q) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + q) << 2), (int)v);
}
struct FmxwProbe { uint4 pc; uint64_t pidx, base; uint32_t bit; };
__device__ __forceinline__ FmxwProbe fmxw_ep_probe_issue(const FmxWideBits &bv, uint64_t i) {
  FmxwProbe pr;
  if (i > bv.len) i = bv.len;
  pr.pidx = fmxw_div3(i >> 5);                        // i / 96
  pr.bit = (uint32_t)(i - pr.pidx * FMX_BITS_PER_PIECE);
  FMX_CHECK((pr.pidx >> 3) < bv.nrec && ((pr.pidx >> 3) >> bv.sb_shift) < bv.nsb);
  pr.pc = bv.rec[pr.pidx];
  pr.base = bv.base[(pr.pidx >> 3) >> bv.sb_shift];
  return pr;
}
// rank1(i), the bit B[i] and the first one at or after i when it lies in the piece (FMXW_NONE otherwise)
__device__ __forceinline__ uint64_t fmxw_ep_probe_rank(const FmxwProbe &pr, uint32_t &bit_i, uint64_t &next) {
  const uint32_t bit = pr.bit;
  const uint32_t m0 = fmx_lowmask(bit < 32u ? bit : 32u);
  const uint32_t m1 = bit > 32u ? fmx_lowmask(bit - 32u < 32u ? bit - 32u : 32u) : 0u;
  const uint32_t m2 = bit > 64u ? fmx_lowmask(bit - 64u) : 0u;
  const uint32_t c = __popc(pr.pc.y & m0) + __popc(pr.pc.z & m1) + __popc(pr.pc.w & m2);
  const uint32_t word = bit < 32u ? pr.pc.y : (bit < 64u ? pr.pc.z : pr.pc.w);
  bit_i = (word >> (bit & 31u)) & 1u;
  const uint32_t y = pr.pc.y & ~m0, z = pr.pc.z & ~m1, ww = pr.pc.w & ~m2;
  uint32_t cand = 0xFFFFFFFFu;
  if (y) cand = (uint32_t)__builtin_ctz(y);
  else if (z) cand = 32u + (uint32_t)__builtin_ctz(z);
  else if (ww) cand = 64u + (uint32_t)__builtin_ctz(ww);
  next = cand != 0xFFFFFFFFu ? pr.pidx * FMX_BITS_PER_PIECE + cand : FMXW_NONE;
  return pr.base + pr.pc.x + c;
}
// one round over a level of S: every lane passes ITS endpoint's position and level code and gets back the rank of the
// code before that position RELATIVE to the position's superblock (< 2^31: sb_shift <= 31, checked at the launch) and
// match = [the entry at the position has the code] (bit 31 of the same group sum).  Endpoints dead in every group of the
// wave are skipped.
template <int FMT>
__device__ __forceinline__ void fmxw_ep_round(const uint4 *__restrict__ rec, uint64_t pos, uint32_t code, bool live,
                                              uint32_t base, uint32_t g, uint32_t &rank, uint32_t &match) {
  constexpr int SH = (FMT == 3) ? 8 : 7;
  constexpr uint32_t OM = (FMT == 3) ? 255u : 127u;
  constexpr uint32_t PER = (FMT == 3) ? 32u : 16u;
  constexpr int PSH = (FMT == 3) ? 5 : 4;
  const uint32_t ri = (uint32_t)(pos >> SH), oc = ((uint32_t)pos & OM) | (code << 8);
  uint32_t bo[8];
  uint4 p[8];
  const unsigned long long lv = __ballot(live);
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    const uint32_t br = fmxw_grp_bcast(ri, base, q);
    bo[q] = fmxw_grp_bcast(oc, base, q);
    p[q] = rec[(size_t)br * 8u + g];
  }
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    const uint32_t off = bo[q] & OM, cd = bo[q] >> 8;
    const uint32_t mt = fmx_piece_match<FMT>(p[q], cd);
    int nb = (int)off - (int)(g * PER);
    nb = nb < 0 ? 0 : (nb > (int)PER ? (int)PER : nb);
    uint32_t v = __popc(mt & (uint32_t)((1ull << nb) - 1ull));
    if (FMT == 3) v += (g == cd) ? p[q].x : 0u;
    else v += (g == (cd >> 1)) ? ((cd & 1u) ? p[q].y : p[q].x) : 0u;
    v |= (g == (off >> PSH)) ? (((mt >> (off & (PER - 1u))) & 1u) << 31) : 0u;
    const uint32_t sum = fmx_group_sum(v);
    if (g == q) { rank = sum & 0x7FFFFFFFu; match = sum >> 31; }
  }
}
// the same round for get_l + lf_map (fm_index.rs:82-91): the level code is READ at the position (WaveletMatrix::get) and
// the rank is of that code
template <int FMT>
__device__ __forceinline__ void fmxw_ep_round_access(const uint4 *__restrict__ rec, uint64_t pos, bool live, uint32_t base,
                                                     uint32_t g, uint32_t &rank, uint32_t &code) {
  constexpr int SH = (FMT == 3) ? 8 : 7;
  constexpr uint32_t OM = (FMT == 3) ? 255u : 127u;
  constexpr uint32_t PER = (FMT == 3) ? 32u : 16u;
  constexpr int PSH = (FMT == 3) ? 5 : 4;
  const uint32_t ri = (uint32_t)(pos >> SH), of = (uint32_t)pos & OM;
  uint32_t bo[8];
  uint4 p[8];
  const unsigned long long lv = __ballot(live);
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    const uint32_t br = fmxw_grp_bcast(ri, base, q);
    bo[q] = fmxw_grp_bcast(of, base, q);
    p[q] = rec[(size_t)br * 8u + g];
  }
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    const uint32_t off = bo[q];
    const uint32_t cd = fmx_group_sum((g == (off >> PSH)) ? fmx_piece_code<FMT>(p[q], off & (PER - 1u)) : 0u);
    const uint32_t mt = fmx_piece_match<FMT>(p[q], cd);
    int nb = (int)off - (int)(g * PER);
    nb = nb < 0 ? 0 : (nb > (int)PER ? (int)PER : nb);
    uint32_t v = __popc(mt & (uint32_t)((1ull << nb) - 1ull));
    if (FMT == 3) v += (g == cd) ? p[q].x : 0u;
    else v += (g == (cd >> 1)) ? ((cd & 1u) ? p[q].y : p[q].x) : 0u;
    const uint32_t sum = fmx_group_sum(v);
    if (g == q) { rank = sum; code = cd; }
  }
}
// select1(k) by one lane; len when k >= #ones (vers-vecs RsVec::select1)
__device__ __forceinline__ uint64_t fmxw_ep_select(const FmxWideBits &bv, uint64_t k) {
  return k < bv.ones ? fmxw_bits_lane_select(bv, k) : bv.len;
}
// RLFMIndexBackend::lf_map2 for 8 endpoints per group (rlfmi.rs:135-143): j = b.rank1(i) and the bit b[i] lane-wise; ONE
// rank chain at lo = b.rank1(i + 1) - 1 gives s.rank(lo, c) and m = [s[lo] == c], so s.rank(j, c) = that + (j > lo ? m : 0)
// and get_l(i) == c is m (rlfmi.rs:137-138); then bp.select1(cs[c] + nr) and, when m, + i - b.select1(j) (rlfmi.rs:139-141).
// Dead lanes pass i = 0, c = 0, live = false and ignore the result.
template <class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_r_ep_lf_map2(const FmxWideDev &w, const GB &gbase, const GK &gk, uint32_t c, uint64_t i,
                                                      bool live, uint32_t base, uint32_t g) {
  const FmxwProbe pr = fmxw_ep_probe_issue(w.b, i);
  uint32_t bit;
  uint64_t nx;
  const uint64_t j = fmxw_ep_probe_rank(pr, bit, nx);          // b.rank1(i)            rlfmi.rs:136
  uint64_t pos = j - 1u + bit;                                 // b.rank1(i + 1) - 1    rlfmi.rs:124
  uint32_t m = 1u;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    const uint32_t code = (c >> L.shift) & L.mask;
    uint32_t r = 0, mt = 0;
    FMX_CHECK((pos >> (L.fmt == 3 ? 8 : 7)) < L.nrec);
    if (L.fmt == 3) fmxw_ep_round<3>(L.rec, pos, code, live, base, g, r, mt);
    else fmxw_ep_round<4>(L.rec, pos, code, live, base, g, r, mt);
    m &= mt;
    pos = gbase(l, pos, code) + r;                             // the 64-bit base of the position's superblock
  }
  const uint64_t nr = gk(c) + pos + (bit ? 0u : m);            // cs[c] + s.rank(j, c)  rlfmi.rs:137,139
  const bool need_st = live && m && nx == FMXW_NONE;           // the run start lies before the piece
  uint64_t f = 0, st = nx;
  if (w.bp.pos && w.b.pos) {                                   // stored positions: both selects are one load, issued together
    uint64_t vf = 0, vs = 0;
    if (live && nr < w.bp.ones) vf = w.bp.pos[nr];
    if (need_st && j < w.b.ones) vs = w.b.pos[j];
    f = nr < w.bp.ones ? vf : w.bp.len;
    if (need_st) st = j < w.b.ones ? vs : w.b.len;
  } else {
    if (live) f = fmxw_ep_select(w.bp, nr);                    // bp.select1(cs[c] + nr)
    if (need_st) st = fmxw_ep_select(w.b, j);
  }
  return m ? f + i - st : f;                                   // rlfmi.rs:138-142
}
// FMIndexBackend / FMIndexMultiPiecesBackend::lf_map2 for 8 endpoints per group (fm_index.rs:93-95, multi_pieces.rs:147-153):
// one rank round per wavelet level
template <bool MP, class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_g_ep_lf_map2(const FmxWideDev &w, const GB &gbase, const GK &gk, uint32_t c, uint64_t i,
                                                      bool live, uint32_t base, uint32_t g) {
  uint64_t pos = i;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    const uint32_t code = (c >> L.shift) & L.mask;
    uint32_t r = 0, mt = 0;
    FMX_CHECK((pos >> (L.fmt == 3 ? 8 : 7)) < L.nrec);
    if (L.fmt == 3) fmxw_ep_round<3>(L.rec, pos, code, live, base, g, r, mt);
    else fmxw_ep_round<4>(L.rec, pos, code, live, base, g, r, mt);
    pos = gbase(l, pos, code) + r;
  }
  uint64_t r = gk(c) + pos;
  if (MP && c == 0u) r = fmxw_multi_zero(w, i, r);
  return r;
}
// get_l + lf_map for 8 walks per group (fm_index.rs:82-91): access + rank along the same positions
template <bool MP, class GB, class GK>
__device__ __forceinline__ uint64_t fmxw_g_ep_lf_map(const FmxWideDev &w, const GB &gbase, const GK &gk, uint64_t i, bool live,
                                                     uint32_t base, uint32_t g) {
  uint64_t pos = i;
  uint32_t sym = 0;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxWideLevel &L = w.lv[l];
    uint32_t r = 0, code = 0;
    FMX_CHECK((pos >> (L.fmt == 3 ? 8 : 7)) < L.nrec);
    if (L.fmt == 3) fmxw_ep_round_access<3>(L.rec, pos, live, base, g, r, code);
    else fmxw_ep_round_access<4>(L.rec, pos, live, base, g, r, code);
    sym |= code << L.shift;
    pos = gbase(l, pos, code) + r;
  }
  uint64_t r = gk(sym) + pos;                                  // fm_index.rs:86-91
  if (MP && sym == 0u) r = fmxw_multi_zero(w, i, r);
  return r;
}
// SearchWrapper::search for a batch (wrapper.rs:103-124) on a wide generic index (KD: FM, RLFM, multi-pieces), an interval endpoint per lane.  Pattern
// slots are dealt to the groups first (slot = pair * ngroups + group), so a batch smaller than the grid has one live
// lane pair per group and a round costs one record per level.
template <bool GLDS, int KD>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_count_ep_kernel(
    FmxWideDev w, const void *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  FMXW_GBASES(w, GLDS);
  const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, base = lane & ~7u;
  const uint32_t is_e = g & 1u;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 3;
  const uint64_t slot = (uint64_t)(g >> 1) * ngroups + (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3);
  const uint64_t nslots = ngroups * 4u;
  const uint64_t ptot = npat ? off[npat] : 0;       // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  uint64_t k = slot, pbeg = 0, j = 0, pos = 0, nsteps = 0;
  bool active = k < npat, fresh = true;
  uint32_t c = 0;
  while (__any(active)) {
    if (active && fresh) {
      pbeg = off[k];
      const uint64_t pend = off[k + 1];
      j = pend - pbeg;
      bool bad = pend < pbeg || pbeg < pmin || pend > ptot;          // offsets that go backwards or leave the pattern buffer
      if (s0e0) {                                     // Search::search on an existing Search   wrapper.rs:105-106
        const uint64_t mine = s0e0[2 * k + is_e], other = s0e0[2 * k + (is_e ^ 1u)];
        pos = mine;
        bad |= mine > w.n || other > w.n;             // not a range of this index
      } else {
        pos = is_e ? w.n : 0ull;                      // SearchIndexWrapper::search: (0, len)   wrapper.rs:41
      }
      if (bad) {                                      // refuse, do not read
        if (is_e) atomicOr(w.status, 1u << FMX_ERR_ARG);
        pos = 0; j = 0;
      }
      c = j ? fmx_load_sym(pat, w.sym_bytes, pbeg + j - 1) : 0u;   // for c in pattern.iter().rev()   wrapper.rs:108
      fresh = false;
    }
    bool stepping = active && j != 0;
    if (stepping && c > w.max_character) {            // reference: panic on cs[c]
      if (is_e) atomicOr(w.status, 1u << FMX_ERR_SYMBOL_RANGE);
      pos = 0; j = 0; stepping = false;
    }
    const uint32_t cn = (stepping && j > 1) ? fmx_load_sym(pat, w.sym_bytes, pbeg + j - 2) : 0u;   // rides along with the probes
    uint64_t np;                                      // wrapper.rs:109-110
    if constexpr (KD == FMX_KIND_RLFM) np = fmxw_r_ep_lf_map2(w, gbase, gk, stepping ? c : 0u, stepping ? pos : 0ull, stepping, base, g);
    else np = fmxw_g_ep_lf_map2<KD == FMX_KIND_MULTI>(w, gbase, gk, stepping ? c : 0u, stepping ? pos : 0ull, stepping, base, g);
    if (stepping) {
      pos = np;
      c = cn;
      j--;
      nsteps += is_e ^ 1u;
    }
    const uint64_t other = (uint64_t)fmx_dpp_xor1((uint32_t)pos) | ((uint64_t)fmx_dpp_xor1((uint32_t)(pos >> 32)) << 32);
    if (active && (j == 0 || pos == other)) {         // wrapper.rs:111-113
      if (is_e) {
        if (out_e) out_e[k] = pos;
        if (out_cnt) out_cnt[k] = pos - other;        // wrapper.rs:132-134
      } else if (out_s) {
        out_s[k] = pos;
      }
      k += nslots;
      active = k < npat;
      fresh = true;
    }
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// get_sa for a batch of rows on a generic FM / multi-pieces index with text-order samples, A WALK PER LANE (round 5): a
// wave takes 64 consecutive hits through the walk together -- the phase probe of the start row lane-wise, then LF steps
// (one access + rank round per level and group of eight walks, fmxw_ep_round_access) while any lane has steps to go (a walk
// is SA[row] mod 2^level steps long), the final probe and the sample lane-wise.  64 lines in flight per wave in every
// stage; the four-walks-per-group kernel above had 32 and ran at 0.42 of the request ceiling with 117 VGPRs.
template <bool GLDS, bool MP>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_walk_text_ep_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                          uint64_t *__restrict__ steps_out) {
  FMXW_GBASES(w, GLDS);
  const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, base = lane & ~7u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t nsteps = 0;
  for (uint64_t h0 = wave * 64u; h0 < total; h0 += nwaves * 64u) {
    const uint64_t h = h0 + lane;
    const bool in = h < total;
    uint64_t row = in ? io[h] : 0ull;
    const bool bad = in && row >= w.n;              // not a row of this index: refuse, do not read
    const bool ok = in && !bad;
    if (bad) atomicOr(w.status, 1u << FMX_ERR_ARG);
    if (!ok) row = 0;
    uint64_t r0;
    uint32_t ph = fmxw_phase_probe(w, row, r0);
    if (!ok) ph = 0;
    nsteps += ph;
    uint32_t left = ph;
    while (__any(left != 0u)) {                     // None: i = lf_map(i); steps += 1        fm_index.rs:134-137
      const bool live = left != 0u;
      const uint64_t nr = fmxw_g_ep_lf_map<MP>(w, gbase, gk, live ? row : 0ull, live, base, g);
      if (live) { row = nr; left--; }
    }
    if (ph) {                                       // the final row is a phase-0 row: its sample's index
      [[maybe_unused]] const uint32_t p2 = fmxw_phase_probe(w, row, r0);
      FMX_CHECK(p2 == 0u);
    }
    uint64_t v = (ok ? w.samples[r0] : 0ull) + ph;  // Some(sa) => (sa + steps) % len        fm_index.rs:131-132
    if (v >= w.n) v -= w.n;
    if (in) io[h] = ok ? v : ~0ull;
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// text-order walks, a lane per walk: 64 consecutive hits per wave (the lanes run the same code; the walks differ in
// length by at most 2^level - 1 steps)
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_r_walk_text_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                       uint64_t *__restrict__ steps_out) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  uint64_t nsteps = 0;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total; h += nth) {
    const uint64_t row = io[h];
    uint64_t v = ~0ull, st = 0;
    if (row >= w.n) atomicOr(w.status, 1u << FMX_ERR_ARG);       // refuse, do not read
    else v = fmxw_r_get_sa_text(w, [&](uint64_t r) { return fmxw_r_lane_lf(w, r); }, row, st);
    io[h] = v;
    nsteps += st;
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

template <bool LOCKSTEP>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_r_walk_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                  uint64_t *__restrict__ steps_out) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t lmask = (1ull << w.sa_level) - 1ull;
  uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool have = h < total;
  uint64_t row = have ? io[h] : 0, steps = 0, nsteps = 0;
  [[maybe_unused]] bool done = false;               // LOCKSTEP: this lane's walk is over, its neighbours' are not
  while (__any(have)) {
    if (LOCKSTEP && !__any(have && !done)) {        // wave-uniform: the next 64 hits
      h += nth;
      have = h < total;
      row = have ? io[h] : 0;
      steps = 0;
      done = false;
      continue;
    }
    if (LOCKSTEP && done) {
    } else if (have && (row >= w.n || (row & lmask) == 0)) {   // Some(sa): (sa + steps) % len            rlfmi.rs:178-182
      uint64_t v = ~0ull;
      if (row >= w.n) {                             // refuse, do not read
        atomicOr(w.status, 1u << FMX_ERR_ARG);
      } else {
        v = w.samples[row >> w.sa_level] + steps;
        if (v >= w.n) v -= w.n;
      }
      io[h] = v;
      nsteps += steps;
      if (LOCKSTEP) {
        done = true;
      } else {
        h += nth;
        have = h < total;
        row = have ? io[h] : 0;
        steps = 0;
      }
    } else if (have) {                              // None: i = lf_map(i); steps += 1          rlfmi.rs:183-186
      row = fmxw_r_lane_lf(w, row);
      steps++;
    }
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// the trait methods, batched (backend.rs:9-19, 29-31, 34-40).  op: 0 get_l, 1 lf_map, 2 lf_map2, 3 get_sa, 4 get_f,
// 5 fl_map, 6 piece_id
template <int KD>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_scalar_kernel(FmxWideDev w, int op, const uint64_t *__restrict__ cc,
                                                                    const uint64_t *__restrict__ ii, uint64_t k,
                                                                    uint64_t *__restrict__ out) {
  constexpr bool RL = KD == FMX_KIND_RLFM, MP = KD == FMX_KIND_MULTI;
  FMXW_GBASES(w, false);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < k; q += ngroups) {
    const uint64_t i = ii[q];
    uint64_t res = ~0ull;
    if (op == 2) {                                  // lf_map2(c, i), i in [0, n]
      const uint64_t c = cc[q];
      if (c > w.max_character || i > w.n) {
        if (g == 0) atomicOr(w.status, 1u << (c > w.max_character ? FMX_ERR_SYMBOL_RANGE : FMX_ERR_ARG));
      } else if constexpr (RL) {
        uint64_t a = i, b = i;
        fmxw_r_lf_map2_pair(w, gbase, gk, (uint32_t)c, a, b, g);
        res = a;
      } else {
        res = gk((uint32_t)c) + fmxw_g_chain(w, gbase, (uint32_t)c, i, g);
        if (MP && c == 0u) res = fmxw_multi_zero(w, i, res);       // multi_pieces.rs:147-153
      }
    } else if (i >= w.n) {
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
    } else if (op == 0 || op == 1) {
      uint32_t sym;
      uint64_t r;
      if constexpr (RL) r = fmxw_r_lf(w, gbase, gk, i, g, sym); else r = fmxw_g_lf_map<MP>(w, gbase, gk, i, g, sym);
      res = op == 0 ? (uint64_t)sym : r;
    } else if (op == 6) {                           // HasMultiPieces::piece_id (multi_pieces.rs:206-219)
      if constexpr (MP) {
        uint64_t row = i;
        for (;;) {
          uint32_t sym;
          const uint64_t raw = fmxw_g_lf(w, gbase, row, g, sym);
          if (sym == 0u) {                          // doc[bw.rank(i, 0)] + 1 mod pieces
            FMX_CHECK(gk(0u) + raw < w.doc_count);
            res = ((uint64_t)w.doc[gk(0u) + raw] + 1u) % w.doc_count;
            break;
          }
          row = gk(sym) + raw;
        }
      }
    } else if (op == 4 || op == 5) {                // get_f / fl_map
      uint32_t sym;
      uint64_t r;
      if constexpr (RL) r = fmxw_r_fl(w, gbase, i, g, sym); else r = fmxw_g_fl(w, gbase, i, g, sym);
      if (MP && sym == 0u) r = ~0ull;               // fl_map: None (multi_pieces.rs:176-178)
      res = op == 4 ? (uint64_t)sym : r;
    } else if (!RL && w.walk) {                     // get_sa, text-order samples: through the walk records
      res = fmxw_get_sa_walk(w, i, g);
    } else if (w.phase) {                           // get_sa, text-order samples of a generic index (group-uniform lane code)
      uint64_t st;
      if constexpr (RL) {
        res = fmxw_r_get_sa_text(w, [&](uint64_t r) { return fmxw_r_lf_step(w, gbase, gk, r, g); }, i, st);
      } else {
        res = fmxw_r_get_sa_text(w, [&](uint64_t r) { uint32_t sy; return fmxw_g_lf_map<MP>(w, gbase, gk, r, g, sy); }, i, st);
      }
    } else {                                        // get_sa
      const uint64_t lmask = (1ull << w.sa_level) - 1ull;
      uint64_t row = i, steps = 0;
      while (row & lmask) {
        if constexpr (RL) {
          row = fmxw_r_lf_step(w, gbase, gk, row, g);
        } else {
          uint32_t sym;
          row = fmxw_g_lf_map<MP>(w, gbase, gk, row, g, sym);
        }
        steps++;
      }
      uint64_t v = w.samples[row >> w.sa_level] + steps;
      if (v >= w.n) v -= w.n;
      res = v;
    }
    if (g == 0) out[q] = res;
  }
}

// Match::iter_chars_backward / iter_chars_forward for many rows (wrapper.rs:154-183): one group per row
__device__ __forceinline__ void fmxw_store_sym(void *p, uint32_t sb, uint64_t i, uint32_t v) {
  if (sb == 1) ((uint8_t *)p)[i] = (uint8_t)v;
  else if (sb == 2) ((uint16_t *)p)[i] = (uint16_t)v;
  else ((uint32_t *)p)[i] = v;
}
template <int KD>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_extract_kernel(FmxWideDev w, const uint64_t *__restrict__ rows,
                                                                     uint64_t nrows, uint32_t len, int forward,
                                                                     void *__restrict__ out, uint64_t *__restrict__ out_len,
                                                                     uint64_t *__restrict__ out_next) {
  constexpr bool RL = KD == FMX_KIND_RLFM, MP = KD == FMX_KIND_MULTI;
  FMXW_GBASES(w, false);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < nrows; q += ngroups) {
    uint64_t i = rows[q], next = ~0ull;
    uint32_t t = 0;
    if (i >= w.n) {
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
    } else {
      bool ended = false;
      for (; t < len; t++) {
        uint32_t sym;
        uint64_t nx;
        if constexpr (RL) {
          nx = forward ? fmxw_r_fl(w, gbase, i, g, sym) : fmxw_r_lf(w, gbase, gk, i, g, sym);
        } else if (forward) {
          nx = fmxw_g_fl(w, gbase, i, g, sym);
          if (MP && sym == 0u) { ended = true; break; }           // fl_map: None -- ends without yielding (wrapper.rs:172-183)
        } else {
          nx = fmxw_g_lf_map<MP>(w, gbase, gk, i, g, sym);
        }
        if (g == 0) fmxw_store_sym(out, w.sym_bytes, q * (uint64_t)len + t, sym);
        i = nx;
      }
      if (!ended) next = i;
    }
    if (g == 0 && out_len) out_len[q] = t;
    if (g == 0 && out_next) out_next[q] = next;
  }
}

// L column of rows [0, n), one byte per row (get_l).  A LANE per row (consecutive lanes, consecutive rows): the
// first level's records are read in order, and the rank that leads to the next level is taken by the lane alone
// over the pieces in front of its entry -- one random 16-byte read per row and further level instead of a group's
// dependent round trips (4.3e9 rows at n = 2^32).
template <int FMT>
__device__ __forceinline__ uint32_t fmxw_g_lane_rank(const FmxWideLevel &L, uint32_t r, uint32_t pi, uint32_t bit,
                                                     const uint4 &piece, uint32_t code) {
  uint32_t cnt;
  if (FMT == 3) {
    cnt = L.rec[(size_t)r * 8u + code].x;
  } else {
    const uint4 c = L.rec[(size_t)r * 8u + (code >> 1)];
    cnt = (code & 1u) ? c.y : c.x;
  }
  for (uint32_t q = 0; q < pi; q++) cnt += __popc(fmx_piece_match<FMT>(L.rec[(size_t)r * 8u + q], code));
  return cnt + __popc(fmx_piece_match<FMT>(piece, code) & ((1u << bit) - 1u));
}
// run that holds row i = b.rank1(i + 1) - 1, by one lane from the row's piece alone (rlfmi.rs:122-125)
__device__ __forceinline__ uint64_t fmxw_bits_lane_run(const FmxWideBits &bv, uint64_t i) {
  const uint64_t pidx = fmxw_div3(i >> 5);          // i / 96: records are 8 consecutive pieces
  const uint32_t b1 = (uint32_t)(i - pidx * FMX_BITS_PER_PIECE) + 1u;
  const uint4 pc = bv.rec[pidx];
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  return bv.base[(pidx >> 3) >> bv.sb_shift] + pc.x + __popc(pc.y & m0) + __popc(pc.z & m1) + __popc(pc.w & m2) - 1u;
}
// iter_matches() bookkeeping of a multi-pieces index (wrapper.rs:57-82, 203-217): with match_prefix_only the matches of
// [s, e) are its rows whose L symbol is the end marker
template <bool ROWS>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_match_kernel(FmxWideDev w, const uint64_t *__restrict__ s,
                                                                   const uint64_t *__restrict__ e, const uint64_t *__restrict__ off,
                                                                   uint64_t npat, int prefix_only, uint64_t *__restrict__ out) {
  FMXW_GBASES(w, false);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    const uint64_t a = s[k], b = e[k];
    uint64_t cnt = b > a ? b - a : 0;
    if (a > w.n || b > w.n) {                       // not a range of this index
      if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
      cnt = 0;
    }
    uint64_t ra = 0;
    if (prefix_only && cnt) {
      ra = fmxw_g_chain(w, gbase, 0u, a, g);
      cnt = fmxw_g_chain(w, gbase, 0u, b, g) - ra;
    }
    if (!ROWS) {
      if (g == 0) out[k] = cnt;
    } else if (!prefix_only) {
      for (uint64_t t = g; t < cnt; t += FMX_GROUP) out[off[k] + t] = a + t;
    } else {
      for (uint64_t j = 0; j < cnt; j++) {          // the j-th end marker of L at or after row a
        uint64_t target = ra + j;
        for (uint32_t l = w.nlevels; l-- > 0;) target = fmxw_g_select(w, w.lv[l], 0u, target, g);
        if (g == 0) out[off[k] + j] = target;
      }
    }
  }
  (void)gk;
}

template <bool RL>
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_g_export_l_kernel(FmxWideDev w, void *__restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < w.n; i += stride) {
    uint64_t pos = RL ? fmxw_bits_lane_run(w.b, i) : i;   // RLFM: get_l(i) = s[b.rank1(i + 1) - 1]
    uint32_t sym = 0;
    for (uint32_t l = 0; l < w.nlevels; l++) {
      const FmxWideLevel &L = w.lv[l];
      const bool f3 = L.fmt == 3;
      const uint32_t r = (uint32_t)(pos >> (f3 ? 8 : 7)), off = (uint32_t)pos & (f3 ? 255u : 127u);
      const uint32_t pi = f3 ? off >> 5 : off >> 4, bit = f3 ? off & 31u : off & 15u;
      const uint4 piece = L.rec[(size_t)r * 8u + pi];
      const uint32_t code = f3 ? fmx_piece_code<3>(piece, bit) : fmx_piece_code<4>(piece, bit);
      sym |= code << L.shift;
      if (l + 1 < w.nlevels)
        pos = L.base[(size_t)(pos >> w.sb_shift) * 16u + code] +
              (f3 ? fmxw_g_lane_rank<3>(L, r, pi, bit, piece, code) : fmxw_g_lane_rank<4>(L, r, pi, bit, piece, code));
    }
    fmxw_store_sym(out, w.sym_bytes, i, sym);
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static int fmxw_unsupported(const char *what) {
  fmx_set_error(FMX_ERR_UNSUPPORTED, what);
  return FMX_ERR_UNSUPPORTED;
}
static void fmxw_time_begin(const fmx_index *idx, hipStream_t st) {
  fmx_index *m = const_cast<fmx_index *>(idx);
  if (idx->timing == 1) {
    (void)hipMemsetAsync(m->d_steps, 0, sizeof(uint64_t), st);
    (void)hipEventRecord(m->ev0, st);
  } else if (idx->timing == 2 && m->ev_series && m->series_n < FMX_SERIES_CAP) {
    (void)hipEventRecord(m->ev_series[2 * m->series_n], st);
  }
}
static void fmxw_time_end(const fmx_index *idx, hipStream_t st) {
  fmx_index *m = const_cast<fmx_index *>(idx);
  if (idx->timing == 1) {
    (void)hipEventRecord(m->ev1, st);
    m->ev_valid = 1;
  } else if (idx->timing == 2 && m->ev_series && m->series_n < FMX_SERIES_CAP) {
    (void)hipEventRecord(m->ev_series[2 * m->series_n + 1], st);
    m->series_n++;
  }
}

// A/B switch of the measurement build (read per launch): FMXW_R_COUNT_GROUP=1 = round 4's group-per-pattern count and
// four-walks-per-group text-order walk of the generic kernels; the shipped library reads no environment
static inline bool fmxw_env_group() {
#ifdef FMX_MEASURE
  const char *v = getenv("FMXW_R_COUNT_GROUP");
  return v && atoi(v) != 0;
#else
  return false;
#endif
}

int fmxw_launch_count(const fmx_index *idx, const void *d_pat, const uint64_t *d_off, uint64_t npat,
                      const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e, uint64_t *d_cnt, hipStream_t st) {
  if (npat == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  fmxw_time_begin(idx, st);
  if (w.generic) {
#define FMXW_GCNT(GLDS, RL)                                                                                        \
  hipLaunchKernelGGL((fmxw_g_count_kernel<GLDS, RL>), dim3(fmxw_grid(npat)), dim3(FMXW_BLOCK), 0, st, w, d_pat,       \
                     d_off, npat, d_s0e0, d_s, d_e, d_cnt, idx->timing == 1 ? idx->d_steps : nullptr)
    const bool r_group = fmxw_env_group();
    if (w.sb_shift <= 31u && !r_group) {              // an interval endpoint per lane (32 patterns per block at a time)
      uint64_t eb = (npat + FMXW_BLOCK / 8 - 1) / (FMXW_BLOCK / 8);
      uint64_t cap = FMXW_EP_BLOCKS;
#ifdef FMX_MEASURE
      if (const char *v = getenv("FMXW_EP_BLOCKS")) cap = (uint64_t)atol(v) > 0 ? (uint64_t)atol(v) : cap;   // read per launch: sweeps
#endif
      if (eb > cap) eb = cap;
#define FMXW_RCNT(GLDS, KD)                                                                                         \
  hipLaunchKernelGGL((fmxw_g_count_ep_kernel<GLDS, KD>), dim3((unsigned)eb), dim3(FMXW_BLOCK), 0, st, w, d_pat, d_off,  \
                     npat, d_s0e0, d_s, d_e, d_cnt, idx->timing == 1 ? idx->d_steps : nullptr)
      if (w.kind == FMX_KIND_RLFM) { if (w.nsb <= FMXW_GLDS_SB) FMXW_RCNT(true, FMX_KIND_RLFM); else FMXW_RCNT(false, FMX_KIND_RLFM); }
      else if (w.kind == FMX_KIND_MULTI) { if (w.nsb <= FMXW_GLDS_SB) FMXW_RCNT(true, FMX_KIND_MULTI); else FMXW_RCNT(false, FMX_KIND_MULTI); }
      else if (w.nsb <= FMXW_GLDS_SB) FMXW_RCNT(true, FMX_KIND_FM); else FMXW_RCNT(false, FMX_KIND_FM);
    }
    else if (w.kind == FMX_KIND_RLFM) { if (w.nsb <= FMXW_GLDS_SB) FMXW_GCNT(true, FMX_KIND_RLFM); else FMXW_GCNT(false, FMX_KIND_RLFM); }
    else if (w.kind == FMX_KIND_MULTI) { if (w.nsb <= FMXW_GLDS_SB) FMXW_GCNT(true, FMX_KIND_MULTI); else FMXW_GCNT(false, FMX_KIND_MULTI); }
    else if (w.nsb <= FMXW_GLDS_SB) FMXW_GCNT(true, FMX_KIND_FM); else FMXW_GCNT(false, FMX_KIND_FM);
    fmxw_time_end(idx, st);
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  // group per pattern, one pattern per group at a time: two patterns per group in flight, or one record load for
  // both ends of a narrow interval, measured slower (0.75-0.80 ms against 0.70 ms, benchmarks/gpu/wide_tune.sh)
#define FMXW_CNT(LDSB)                                                                                             \
  hipLaunchKernelGGL(fmxw_count_kernel<LDSB>, dim3(fmxw_grid(npat)), dim3(FMXW_BLOCK), 0, st, w, (const uint8_t *)d_pat, \
                     d_off, npat, d_s0e0, d_s, d_e, d_cnt, idx->timing == 1 ? idx->d_steps : nullptr)
  if (w.nsb <= FMXW_LDS_SB) FMXW_CNT(true); else FMXW_CNT(false);
  fmxw_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_locate(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                       const uint64_t *d_off, uint64_t total, uint64_t *d_pos, hipStream_t st) {
  if (npat == 0 || total == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  // iter_matches: rows s..e-1 ascending (wrapper.rs:203-217), written where their positions will stand
  if (int rc = fmx_launch_expand64(d_s, d_e, d_off, npat, d_pos, total, w.n, w.status, st)) return rc;
  fmxw_time_begin(idx, st);
  if (w.generic) {
    uint64_t *steps = idx->timing == 1 ? idx->d_steps : nullptr;
#define FMXW_GWALK(GLDS, RL)                                                                                        \
    hipLaunchKernelGGL((fmxw_g_walk_kernel<GLDS, RL>), dim3(fmxw_grid(total)), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps)
    if (w.kind == FMX_KIND_RLFM && w.lfrun && w.phase) {   // run table + text-order samples: a lane per walk
      uint64_t blocks = (total + FMXW_BLOCK - 1) / FMXW_BLOCK;
      if (blocks > FMXW_MAX_BLOCKS * 2) blocks = FMXW_MAX_BLOCKS * 2;
      hipLaunchKernelGGL(fmxw_r_walk_text_kernel, dim3((unsigned)blocks), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps);
    } else if (w.kind == FMX_KIND_RLFM && w.lfrun) {       // run table: a lane per walk
      uint64_t blocks = (total + FMXW_BLOCK - 1) / FMXW_BLOCK;
      if (blocks > FMXW_MAX_BLOCKS * 2) blocks = FMXW_MAX_BLOCKS * 2;
#ifdef FMX_MEASURE     // measurement switch (profiles/r04/wide_rlfm_4g_row_order_*.json): the shipped library reads no environment
      static const int force = getenv("FMXW_RL_LOCKSTEP") ? atoi(getenv("FMXW_RL_LOCKSTEP")) : -1;
#else
      constexpr int force = -1;
#endif
      const bool lockstep = force >= 0 ? force != 0 : total / npat >= 64;
      if (lockstep) hipLaunchKernelGGL(fmxw_r_walk_kernel<true>, dim3((unsigned)blocks), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps);
      else hipLaunchKernelGGL(fmxw_r_walk_kernel<false>, dim3((unsigned)blocks), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps);
    } else if (w.kind == FMX_KIND_RLFM) {
      if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALK(true, FMX_KIND_RLFM); else FMXW_GWALK(false, FMX_KIND_RLFM);
    } else if (w.phase && !fmxw_env_group()) {      // FM / multi-pieces with text-order samples: a walk per lane
      uint64_t blocks = (total + FMXW_BLOCK - 1) / FMXW_BLOCK;
      if (blocks > FMXW_MAX_BLOCKS * 2) blocks = FMXW_MAX_BLOCKS * 2;
#define FMXW_GWALKE(GLDS, MPF)                                                                                      \
      hipLaunchKernelGGL((fmxw_g_walk_text_ep_kernel<GLDS, MPF>), dim3((unsigned)blocks), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps)
      if (w.kind == FMX_KIND_MULTI) { if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALKE(true, true); else FMXW_GWALKE(false, true); }
      else if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALKE(true, false); else FMXW_GWALKE(false, false);
    } else if (w.phase) {                           // measurement builds: round 4's four walks per group
      const unsigned grid = fmxw_grid((total + 3) / 4);
#define FMXW_GWALKT(GLDS, MPF)                                                                                      \
      hipLaunchKernelGGL((fmxw_g_walk_text_kernel<GLDS, MPF>), dim3(grid), dim3(FMXW_BLOCK), 0, st, w, total, d_pos, steps)
      if (w.kind == FMX_KIND_MULTI) { if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALKT(true, true); else FMXW_GWALKT(false, true); }
      else if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALKT(true, false); else FMXW_GWALKT(false, false);
    } else if (w.kind == FMX_KIND_MULTI) {
      if (w.nsb <= FMXW_GLDS_SB) FMXW_GWALK(true, FMX_KIND_MULTI); else FMXW_GWALK(false, FMX_KIND_MULTI);
    } else if (w.nsb <= FMXW_GLDS_SB) {
      FMXW_GWALK(true, FMX_KIND_FM);
    } else {
      FMXW_GWALK(false, FMX_KIND_FM);
    }
    fmxw_time_end(idx, st);
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  if (w.walk) {
    // text-order samples + walk records: the shape of the 32-bit engine's fmx_locate_f3t_kernel (block-wide hit queue,
    // four walks per group with distributed state, write-combining ring)
    const int q = total >= (1u << 16) ? 4 : 1;
    const uint64_t nb0 = total >= (4u << 20) ? 512 : 256;
    const uint64_t per_wave = (total + nb0 * (FMXW_LOC_BLOCK / 64) - 1) / (nb0 * (FMXW_LOC_BLOCK / 64));
    uint32_t chunk = (uint32_t)((per_wave + 7) / 8 * 8);
    if (chunk < 8u * (uint32_t)q) chunk = 8u * (uint32_t)q;
    if (chunk > 64u) chunk = 64u;
    uint64_t nb = nb0;
    const uint64_t min_nb = (total >> 31) + 1;
    if (nb < min_nb) nb = min_nb;
    uint64_t per = (total + nb - 1) / nb;
    per = (per + chunk - 1) / chunk * chunk;
    const uint32_t hpb = (uint32_t)per;
    const unsigned gr = (unsigned)((total + per - 1) / per);
    uint64_t *steps = idx->timing == 1 ? idx->d_steps : nullptr;
#define FMXW_WALKT(QQ, WCF, LDSW)                                                                                   \
    hipLaunchKernelGGL((fmxw_walk_t_kernel<QQ, WCF, LDSW>), dim3(gr), dim3(FMXW_LOC_BLOCK), 0, st, w, total, hpb, chunk, \
                       d_pos, steps)
    const bool lds = w.nwsb <= FMXW_LDS_WSB, wc = chunk == 64u;
    if (q == 4) {
      if (wc) { if (lds) FMXW_WALKT(4, true, true); else FMXW_WALKT(4, true, false); }
      else { if (lds) FMXW_WALKT(4, false, true); else FMXW_WALKT(4, false, false); }
    } else {
      if (wc) { if (lds) FMXW_WALKT(1, true, true); else FMXW_WALKT(1, true, false); }
      else { if (lds) FMXW_WALKT(1, false, true); else FMXW_WALKT(1, false, false); }
    }
    fmxw_time_end(idx, st);
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  // one walk per group at a time: with 2 / 3 / 4 the kernel executes the instructions of every slot for every group
  // of a wave and is issue-bound (0.196 / 0.204 / 0.226 ms against 0.181 ms per 2^20 hits, wide_tune.sh)
#define FMXW_WALK(LDSB)                                                                                            \
  hipLaunchKernelGGL((fmxw_walk_kernel<1, LDSB>), dim3(fmxw_grid(total)), dim3(FMXW_BLOCK), 0, st, w, total, d_pos,  \
                     idx->timing == 1 ? idx->d_steps : nullptr)
  if (w.nsb <= FMXW_LDS_SB) FMXW_WALK(true); else FMXW_WALK(false);
  fmxw_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i, uint64_t k,
                       uint64_t *d_out, hipStream_t st) {
  if (op > 5 && idx->wide.kind != FMX_KIND_MULTI) return fmxw_unsupported("piece_id needs a multi-pieces index");
  if (k == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  if (w.kind == FMX_KIND_RLFM) hipLaunchKernelGGL(fmxw_g_scalar_kernel<FMX_KIND_RLFM>, dim3(fmxw_grid(k)), dim3(FMXW_BLOCK), 0, st, w, op, d_c, d_i, k, d_out);
  else if (w.kind == FMX_KIND_MULTI) hipLaunchKernelGGL(fmxw_g_scalar_kernel<FMX_KIND_MULTI>, dim3(fmxw_grid(k)), dim3(FMXW_BLOCK), 0, st, w, op, d_c, d_i, k, d_out);
  else if (w.generic) hipLaunchKernelGGL(fmxw_g_scalar_kernel<FMX_KIND_FM>, dim3(fmxw_grid(k)), dim3(FMXW_BLOCK), 0, st, w, op, d_c, d_i, k, d_out);
  else hipLaunchKernelGGL(fmxw_scalar_kernel, dim3(fmxw_grid(k)), dim3(FMXW_BLOCK), 0, st, w, op, d_c, d_i, k, d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_export_l(const fmx_index *idx, void *d_out, hipStream_t st) {
  const FmxWideDev w = fmxw_dev(idx);
  if (w.kind == FMX_KIND_RLFM) hipLaunchKernelGGL(fmxw_g_export_l_kernel<true>, dim3(FMXW_MAX_BLOCKS * 4), dim3(FMXW_BLOCK), 0, st, w, d_out);
  else if (w.generic) hipLaunchKernelGGL(fmxw_g_export_l_kernel<false>, dim3(FMXW_MAX_BLOCKS * 4), dim3(FMXW_BLOCK), 0, st, w, d_out);
  else hipLaunchKernelGGL(fmxw_export_l_kernel, dim3(FMXW_MAX_BLOCKS * 4), dim3(FMXW_BLOCK), 0, st, w, (uint8_t *)d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_extract(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len, int forward,
                        void *d_out, uint64_t *d_out_len, uint64_t *d_out_next, hipStream_t st) {
  if (nrows == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  if (w.kind == FMX_KIND_RLFM)
    hipLaunchKernelGGL(fmxw_g_extract_kernel<FMX_KIND_RLFM>, dim3(fmxw_grid(nrows)), dim3(FMXW_BLOCK), 0, st, w, d_rows, nrows, len, forward,
                       d_out, d_out_len, d_out_next);
  else if (w.kind == FMX_KIND_MULTI)
    hipLaunchKernelGGL(fmxw_g_extract_kernel<FMX_KIND_MULTI>, dim3(fmxw_grid(nrows)), dim3(FMXW_BLOCK), 0, st, w, d_rows, nrows, len, forward,
                       d_out, d_out_len, d_out_next);
  else if (w.generic)
    hipLaunchKernelGGL(fmxw_g_extract_kernel<FMX_KIND_FM>, dim3(fmxw_grid(nrows)), dim3(FMXW_BLOCK), 0, st, w, d_rows, nrows, len, forward,
                       d_out, d_out_len, d_out_next);
  else
    hipLaunchKernelGGL(fmxw_extract_kernel, dim3(fmxw_grid(nrows)), dim3(FMXW_BLOCK), 0, st, w, d_rows, nrows, len, forward,
                       (uint8_t *)d_out, d_out_len, d_out_next);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// match counts (d_off == NULL) or match rows of every interval (wrapper.rs:57-82, 203-217)
int fmxw_launch_match(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat, int prefix_only,
                      const uint64_t *d_off, uint64_t *d_out, hipStream_t st) {
  if (npat == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  if (prefix_only && w.kind != FMX_KIND_MULTI) return fmxw_unsupported("match_prefix_only needs a multi-pieces index");
  if (!w.generic && prefix_only) return fmxw_unsupported("match_prefix_only needs a multi-pieces index");
  // (without the filter nothing of the index is read: any wide index may be passed; the generic kernel then only
  // reads the level bases of a generic index -- a one-level index has none, so it gets the filter-free instantiation's
  // arithmetic through the same kernel with nlevels == 0)
  if (d_off) hipLaunchKernelGGL(fmxw_g_match_kernel<true>, dim3(fmxw_grid(npat)), dim3(FMXW_BLOCK), 0, st, w, d_s, d_e, d_off, npat, prefix_only, d_out);
  else hipLaunchKernelGGL(fmxw_g_match_kernel<false>, dim3(fmxw_grid(npat)), dim3(FMXW_BLOCK), 0, st, w, d_s, d_e, d_off, npat, prefix_only, d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_compute_K(const FmxWideDev &w, uint64_t *d_K) {
  const unsigned groups = w.max_character + 1u;
  hipLaunchKernelGGL(fmxw_g_compute_K_kernel, dim3((groups * FMX_GROUP + 63) / 64), dim3(64), 0, 0, w, d_K);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
