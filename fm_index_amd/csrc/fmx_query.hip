// fmx_query.hip -- the hot path: batched backward search (count) and the locate walk.
//
// Replaces, for a batch:   SearchWrapper::search      (wrapper.rs:103-124)
//                          FMIndexBackend::lf_map2     (fm_index.rs:93-95)
//                          MatchIteratorWrapper::next  (wrapper.rs:203-217)
//                          FMIndexBackend::get_sa      (fm_index.rs:127-140)
//                          SOSampledSuffixArray::get   (sample.rs:46-60)
//
// Shape: persistent 8-lane groups, one 128-byte record per group per request.  A count group owns one
// pattern at a time and runs a small state machine, so a group that finishes early (the `s == e` break
// of wrapper.rs:111-113) picks up the next pattern while its wave-mates keep stepping.  The DNA walk
// kernel (fmx_locate_f3p_kernel) serves 4 walks per group with the state of each walk kept in ONE lane
// of every quad, so its bookkeeping runs once per round for all of them; the run-length index and
// large multi-level batches use one interval endpoint / one walk per lane (fmx_ep.h).
// Integer / popcount / DPP work only; bound by HBM requests (count) and by the walk chain (locate).
#include <cstdlib>
#include "fmx_ep.h"

#define FMX_BLOCK 256
#define FMX_MAX_BLOCKS 2048  // 256 CUs x 8 resident 256-thread blocks

static inline unsigned fmx_grid_for_groups(uint64_t units);
// the same, but never more blocks than `cap` (a caller that wants CU slots left free for its copy kernels)
static inline unsigned fmx_grid_capped(uint64_t units, unsigned cap) {
  const unsigned g = fmx_grid_for_groups(units);
  return g < cap ? g : cap;
}
static inline unsigned fmx_grid_for_groups(uint64_t units) {
  uint64_t blocks = (units * FMX_GROUP + FMX_BLOCK - 1) / FMX_BLOCK;
  if (blocks < 1) blocks = 1;
  if (blocks > FMX_MAX_BLOCKS) blocks = FMX_MAX_BLOCKS;
  return (unsigned)blocks;
}

// ---------------------------------------------------------------------------
// k-mer start table (FMX_FLAG_KMER_TABLE).  Entry `code` = the (s, e) SearchWrapper::search
// leaves for the k-mer, early exit included, so a lookup replaces the first k steps exactly.
// code: the symbol consumed FIRST (the pattern's last) sits in the top bits, each coded c - 1.
// ---------------------------------------------------------------------------
// the group's lanes read the last `kk` (<= 16) symbols of the pattern (two per lane) and combine
// them; returns false when one of them is 0 or > max_character (then the stepwise path decides)
__device__ __forceinline__ bool fmx_kmer_code(const uint8_t *__restrict__ pat, uint64_t pend, uint32_t kk,
                                              uint32_t bits, uint32_t max_character, uint32_t g,
                                              uint32_t &code) {
  uint32_t part = 0, bad = 0;
#pragma unroll
  for (uint32_t h = 0; h < 2; h++) {
    const uint32_t t = g + 8u * h;                      // t-th symbol from the back
    if (t < kk) {
      const uint32_t cc = pat[pend - 1u - t];
      bad |= (uint32_t)((cc - 1u) >= max_character);
      part |= ((cc - 1u) & ((1u << bits) - 1u)) << (bits * (kk - 1u - t));
    }
  }
  code = fmx_group_sum(part);                           // disjoint bit fields: sum == or
  return fmx_group_sum(bad) == 0u;
}
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_kmer_build_kernel(FmxDev ix, uint2 *__restrict__ table,
                                                                    uint32_t kk, uint32_t bits) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t ncodes = 1ull << (bits * kk);
  for (uint64_t code = gid; code < ncodes; code += ngroups) {
    uint32_t s = 0, e = ix.n;
    for (uint32_t t = 0; t < kk; t++) {
      const uint32_t c = (uint32_t)((code >> (bits * (kk - 1u - t))) & ((1u << bits) - 1u)) + 1u;
      if (c > ix.max_character) { s = 0; e = 0; break; }   // never looked up
      fmx_lf_map2_pair<KIND, 0>(ix, c, s, e, g);           // wrapper.rs:109-110
      if (s == e) break;                                   // wrapper.rs:111-113
    }
    if (g == 0) table[code] = make_uint2(s, e);
  }
}

// ---------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------
// (8 waves per SIMD asked for explicitly: the RLFM instantiations need 65-70 VGPRs otherwise and
// lose a wave of latency hiding to one register)
template <int KIND, int NL, bool KM = false, int SM = -1>
__global__ __launch_bounds__(FMX_BLOCK, 8) void fmx_count_kernel(
    FmxDev ix, const void *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;

  uint64_t k = gid;          // current pattern
  bool active = k < npat;
  bool fresh = true;         // pattern state not loaded yet
  uint64_t pbeg = 0;         // first symbol of the pattern
  uint32_t j = 0;            // symbols still to consume (from the back)
  uint32_t s = 0, e = 0;
  uint32_t c = 0;            // symbol of the coming step (loaded one step ahead)
  uint32_t nsteps = 0;

  const uint64_t ptot = npat ? off[npat] : 0;   // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  while (active) {
    if (fresh) {
      pbeg = off[k];
      const uint64_t pend = off[k + 1];
      j = (uint32_t)(pend - pbeg);
      // offsets that go backwards or leave the pattern buffer: refuse, do not read (j = 0 from here on)
      const bool badoff = pend < pbeg || pbeg < pmin || pend > ptot || pend - pbeg > 0xFFFFFFFFull;
      if (badoff) {
        if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_ARG);
        j = 0;
      }
      if (s0e0) {            // Search::search on an existing Search (wrapper.rs:105-106)
        const uint64_t s64 = s0e0[2 * k], e64 = s0e0[2 * k + 1];
        s = (uint32_t)s64;
        e = (uint32_t)e64;
        if (s64 > ix.n || e64 > ix.n || badoff) {      // not a range of this index: refuse, do not read
          if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_ARG);
          s = 0; e = 0; j = 0;
        }
      } else {               // SearchIndexWrapper::search: (0, len)   (wrapper.rs:41)
        s = 0;
        e = badoff ? 0u : ix.n;
        if (KM && j >= ix.kmer_k) {                    // the first kmer_k steps from the table (u8 symbols)
          uint32_t code;
          if (fmx_kmer_code((const uint8_t *)pat, pbeg + j, ix.kmer_k, ix.kmer_bits, ix.max_character, g,
                            code)) {
            FMX_TOUCH_G0N(g, &ix.kmer[code]);
            const uint2 se = ix.kmer[code];
            s = se.x;
            e = se.y;
            j = se.x == se.y ? 0u : j - ix.kmer_k;     // empty already: the reference's break
            nsteps += ix.kmer_k;
          }
        }
      }
      c = j ? fmx_load_sym(pat, ix.sym_bytes, pbeg + j - 1) : 0u;  // pattern.iter().rev()  wrapper.rs:108
      fresh = false;
    }
    bool done = (j == 0);
    if (!done) {
      if (c > ix.max_character) {                      // reference: panic on cs[c]
        if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_SYMBOL_RANGE);
        s = 0; e = 0; done = true;
      } else {
        // the next symbol rides along with this step's record loads
        const uint32_t cn = j > 1 ? fmx_load_sym(pat, ix.sym_bytes, pbeg + j - 2) : 0u;
        fmx_lf_map2_pair<KIND, NL, SM>(ix, c, s, e, g);   // wrapper.rs:109-110
        c = cn;
        j--;
        nsteps++;
        if (s == e || j == 0) done = true;             // wrapper.rs:111-113
      }
    }
    if (done) {
      if (g == 0) {
        if (out_s) out_s[k] = s;
        if (out_e) out_e[k] = e;
        if (out_cnt) out_cnt[k] = (uint64_t)(e - s);   // wrapper.rs:132-134
      }
      k += ngroups;
      active = k < npat;
      fresh = true;
    }
  }
  if (steps_out) {
    if (g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
  }
}

// count on the index of an EMPTY text (sais.rs:121-123 accepts n = 0): every interval is (0, 0) --
// the start pair (0, len), any refinement pair, and what a step maps them to.  One lane per
// pattern; offsets, given ranges and symbols are still checked like everywhere else.
__global__ __launch_bounds__(FMX_BLOCK) void fmx_count_empty_kernel(
    uint32_t max_character, uint32_t sym_bytes, uint32_t *status, const void *__restrict__ pat,
    const uint64_t *__restrict__ off, uint64_t npat, const uint64_t *__restrict__ s0e0,
    uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e, uint64_t *__restrict__ out_cnt) {
  const uint64_t ptot = npat ? off[npat] : 0;
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < npat; k += stride) {
    const uint64_t a = off[k], b = off[k + 1];
    if (b < a || a < pmin || b > ptot || (s0e0 && (s0e0[2 * k] | s0e0[2 * k + 1]) != 0)) {
      atomicOr(status, 1u << FMX_ERR_ARG);
    } else if (b > a && fmx_load_sym(pat, sym_bytes, b - 1) > max_character) {   // the first step's cs[c]
      atomicOr(status, 1u << FMX_ERR_SYMBOL_RANGE);
    }
    if (out_s) out_s[k] = 0;
    if (out_e) out_e[k] = 0;
    if (out_cnt) out_cnt[k] = 0;
  }
}

// ---------------------------------------------------------------------------
// count, single 3-bit level (L <= 3: DNA, the BASELINE configs 1/2/3/5).
// The record counters are absolute (cs[] folded in), so one 128-B line per endpoint IS
// lf_map2(c, i); no level descriptors, no K[] lookup.  Per step: the two record loads and
// the NEXT pattern byte are issued together, then one wait; the second record load is
// skipped when both interval ends fall into the same record (e - s < 256 most of the time).
// PPG = patterns a group advances concurrently (independent chains -> more loads in flight).
// ---------------------------------------------------------------------------
template <int PPG, bool SKIP, bool KM = false>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_count_f3_kernel(
    const uint4 *__restrict__ rec, uint32_t n, uint32_t max_character, uint32_t *status,
    const uint2 *__restrict__ kmer, uint32_t kmer_k, uint32_t kmer_bits,
    const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  [[maybe_unused]] const uint32_t nrec = n / 256u + 1u;

  uint64_t k[PPG], pbeg[PPG];
  uint32_t j[PPG], s[PPG], e[PPG], c[PPG];
  bool active[PPG], fresh[PPG];
  uint32_t nsteps = 0;
  bool any = false;
#pragma unroll
  for (int q = 0; q < PPG; q++) {
    k[q] = gid + (uint64_t)q * ngroups;
    active[q] = k[q] < npat;
    fresh[q] = true;
    any |= active[q];
    pbeg[q] = 0; j[q] = 0; s[q] = 0; e[q] = 0; c[q] = 0;
  }
  const uint64_t ptot = npat ? off[npat] : 0;   // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  while (any) {
#pragma unroll
    for (int q = 0; q < PPG; q++) {
      if (active[q] && fresh[q]) {
        pbeg[q] = off[k[q]];
        const uint64_t pend = off[k[q] + 1];
        j[q] = (uint32_t)(pend - pbeg[q]);
        // offsets that go backwards or leave the pattern buffer: refuse, do not read (j = 0 from here on;
        // kept AHEAD of the table lookup: a trailing fix-up of (s, e, j) cost the pair kernel 30 %)
        const bool badoff = pend < pbeg[q] || pbeg[q] < pmin || pend > ptot || pend - pbeg[q] > 0xFFFFFFFFull;
        if (badoff) {
          if (g == 0) atomicOr(status, 1u << FMX_ERR_ARG);
          j[q] = 0;
        }
        if (s0e0) {            // Search::search on an existing Search (wrapper.rs:105-106)
          const uint64_t s64 = s0e0[2 * k[q]], e64 = s0e0[2 * k[q] + 1];
          s[q] = (uint32_t)s64;
          e[q] = (uint32_t)e64;
          if (s64 > n || e64 > n || badoff) {          // not a range of this index: refuse, do not read
            if (g == 0) atomicOr(status, 1u << FMX_ERR_ARG);
            s[q] = 0; e[q] = 0; j[q] = 0;
          }
        } else {               // (0, len)   wrapper.rs:41
          s[q] = 0;
          e[q] = badoff ? 0u : n;
          if (KM && j[q] >= kmer_k) {                  // the first kmer_k steps from the table
            uint32_t code;
            if (fmx_kmer_code(pat, pbeg[q] + j[q], kmer_k, kmer_bits, max_character, g, code)) {
              FMX_TOUCH_G0N(g, &kmer[code]);
              const uint2 se = kmer[code];
              s[q] = se.x;
              e[q] = se.y;
              j[q] = se.x == se.y ? 0u : j[q] - kmer_k;  // empty already: the reference's break
              nsteps += kmer_k;
            }
          }
        }
        c[q] = j[q] ? pat[pbeg[q] + j[q] - 1] : 0u;   // last symbol: pattern.iter().rev()
        fresh[q] = false;
      }
    }
    // issue every load of this round first
    uint4 a[PPG], b[PPG];
    uint32_t cn[PPG];
    bool stepping[PPG];
#pragma unroll
    for (int q = 0; q < PPG; q++) {
      stepping[q] = active[q] && j[q] != 0 && c[q] <= max_character;
      cn[q] = 0;
      if (stepping[q]) {
        uint32_t rs = s[q] >> 8, re = e[q] >> 8;
        FMX_CHECK(rs < nrec && re < nrec);
        FMX_TOUCH_G0(g, &rec[(size_t)rs * 8u]);
        if (re != rs) FMX_TOUCH_G0(g, &rec[(size_t)re * 8u]);   // both ends in one record: ONE line
        if (SKIP) {
          a[q] = rec[(size_t)rs * 8u + g];
          b[q] = make_uint4(0u, 0u, 0u, 0u);
          if (re != rs) b[q] = rec[(size_t)re * 8u + g];
        } else {
          a[q] = rec[(size_t)rs * 8u + g];
          b[q] = rec[(size_t)re * 8u + g];
        }
        if (j[q] > 1) cn[q] = pat[pbeg[q] + j[q] - 2];
      }
    }
    any = false;
#pragma unroll
    for (int q = 0; q < PPG; q++) {
      if (!active[q]) continue;
      bool done = (j[q] == 0);
      if (!done) {
        if (!stepping[q]) {                            // reference: panic on cs[c]
          if (g == 0) atomicOr(status, 1u << FMX_ERR_SYMBOL_RANGE);
          s[q] = 0; e[q] = 0; done = true;
        } else {
          uint4 be = b[q];
          if (SKIP && (s[q] >> 8) == (e[q] >> 8)) be = a[q];
          uint32_t ns = fmx_group_sum(fmx_piece_rank<3>(a[q], s[q] & 255u, c[q], g));  // wrapper.rs:109
          uint32_t ne = fmx_group_sum(fmx_piece_rank<3>(be, e[q] & 255u, c[q], g));    // wrapper.rs:110
          s[q] = ns; e[q] = ne;
          c[q] = cn[q];
          j[q]--;
          nsteps++;
          if (ns == ne || j[q] == 0) done = true;      // wrapper.rs:111-113
        }
      }
      if (done) {
        if (g == 0) {
          if (out_s) out_s[k[q]] = s[q];
          if (out_e) out_e[k[q]] = e[q];
          if (out_cnt) out_cnt[k[q]] = (uint64_t)(e[q] - s[q]);   // wrapper.rs:132-134
        }
        k[q] += (uint64_t)PPG * ngroups;
        active[q] = k[q] < npat;
        fresh[q] = true;
      }
      any |= active[q];
    }
  }
  if (steps_out && g == 0 && nsteps)
    atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}


// ---------------------------------------------------------------------------
// count with the opt-in pair index (FMX_FLAG_PAIR_INDEX): while at least two symbols remain
// (both in 1..4) one probe of the 2-gram records per interval end advances TWO pattern symbols:
//     LF(c1, LF(c2, i)) = cnt2[c1c2] + popcount            (absolute counters hold K2)
// The reference's early exit (wrapper.rs:111-113) is reproduced exactly: when the pair step
// collapses the interval, the single step for the LAST symbol decides whether the reference
// would have stopped after one symbol (then ITS (s, e) is returned) or after two.
// ---------------------------------------------------------------------------
// (Round 6 measured two ways of putting more requests in flight per wave here -- the kernel runs at 0.68 of the request
// ceiling where the plain kernel reaches 0.93, three of a pattern's ~13 dependent round trips being its start: offsets,
// the k-mer code's symbols, the table entry -- and dropped both (profiles/r06/pair_kernel_ab.txt; this kernel: 0.324 ms
// per 2^20 x 32 symbols): two patterns per group with the record loads of both issued before one wait, 0.42 ms
// (0.52 with one pattern in that form: loads and rank decode in separate loops cost more than the second chain
// brings); a pattern as a state machine whose start stages take one round each like a step, every load of a round
// issued before one wait, 0.50-0.64 ms -- the groups of a wave sit in different stages, their loads serialise on shared
// destination registers or take the kernel to 104 VGPRs, and the divergent round costs more than it hides.)
template <bool KM>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_count_pair_kernel(
    const uint4 *__restrict__ rec1, const uint4 *__restrict__ rec2, uint32_t n,
    uint32_t max_character, uint32_t row0, uint32_t row1, uint32_t *status,
    const uint2 *__restrict__ kmer, uint32_t kmer_k, uint32_t kmer_bits,
    const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;

  uint64_t k = gid;
  bool active = k < npat, fresh = true;
  uint64_t pbeg = 0;
  uint32_t j = 0, s = 0, e = 0;
  uint32_t c2 = 0, c1 = 0;   // c2 = last unread symbol, c1 = the one before it
  uint32_t nsteps = 0;
  const uint64_t ptot = npat ? off[npat] : 0;   // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)
  // the offsets of the group's NEXT pattern are requested when the current one starts (round 6): the start of a pattern
  // is three dependent round trips -- offsets, the k-mer code's symbols, the table entry -- of its ~13, and this one
  // travels under the current pattern's steps
  uint64_t nbeg = active ? off[k] : 0, nend = active ? off[k + 1] : 0;
  while (active) {
    if (fresh) {
      pbeg = nbeg;
      const uint64_t pend = nend;
      if (k + ngroups < npat) { nbeg = off[k + ngroups]; nend = off[k + ngroups + 1]; }
      j = (uint32_t)(pend - pbeg);
      // offsets that go backwards or leave the pattern buffer: refuse, do not read (j = 0 from here on)
      const bool badoff = pend < pbeg || pbeg < pmin || pend > ptot || pend - pbeg > 0xFFFFFFFFull;
      if (badoff) {
        if (g == 0) atomicOr(status, 1u << FMX_ERR_ARG);
        j = 0;
      }
      if (s0e0) {
        const uint64_t s64 = s0e0[2 * k], e64 = s0e0[2 * k + 1];
        s = (uint32_t)s64;
        e = (uint32_t)e64;
        if (s64 > n || e64 > n || badoff) {            // not a range of this index: refuse, do not read
          if (g == 0) atomicOr(status, 1u << FMX_ERR_ARG);
          s = 0; e = 0; j = 0;
        }
      } else {
        s = 0;
        e = badoff ? 0u : n;
        if (KM && j >= kmer_k) {                       // the first kmer_k steps from the table
          uint32_t code;
          if (fmx_kmer_code(pat, pbeg + j, kmer_k, kmer_bits, max_character, g, code)) {
            FMX_TOUCH_G0N(g, &kmer[code]);
            const uint2 se = kmer[code];
            s = se.x;
            e = se.y;
            j = se.x == se.y ? 0u : j - kmer_k;        // empty already: the reference's break
            nsteps += kmer_k;
          }
        }
      }
      c2 = j ? pat[pbeg + j - 1] : 0u;
      c1 = j > 1 ? pat[pbeg + j - 2] : 0u;
      fresh = false;
    }
    bool done = (j == 0);
    if (!done) {
      if (c2 > max_character) {                        // reference: panic on cs[c]
        if (g == 0) atomicOr(status, 1u << FMX_ERR_SYMBOL_RANGE);
        s = 0; e = 0; done = true;
      } else {
        const bool pair = j >= 2 && (c2 - 1u) < 4u && (c1 - 1u) < 4u;
        // the two symbols after these ride along with the record loads
        const uint32_t n1 = j > 2 ? pat[pbeg + j - 3] : 0u;
        const uint32_t n2 = j > 3 ? pat[pbeg + j - 4] : 0u;
        uint32_t used;
        if (pair) {
          const uint32_t code = (c1 - 1u) * 4u + (c2 - 1u);
          FMX_CHECK((s >> 7) < n / 128u + 1u && (e >> 7) < n / 128u + 1u);
          FMX_TOUCH_G0(g, &rec2[(size_t)(s >> 7) * 8u]);
          if ((e >> 7) != (s >> 7)) FMX_TOUCH_G0(g, &rec2[(size_t)(e >> 7) * 8u]);
          const uint4 a = rec2[(size_t)(s >> 7) * 8u + g];
          const uint4 b = rec2[(size_t)(e >> 7) * 8u + g];
          uint32_t ns = fmx_group_sum(fmx_piece_rank<4>(a, s & 127u, code, g));
          uint32_t ne = fmx_group_sum(fmx_piece_rank<4>(b, e & 127u, code, g));
          if (code == 0u) {  // the two rows without a 2-gram are stored as code 0
            ns -= (uint32_t)(s > row0) + (uint32_t)(s > row1);
            ne -= (uint32_t)(e > row0) + (uint32_t)(e > row1);
          }
          used = 2;
          if (ns == ne) {
            // would the reference already have stopped after the last symbol alone?
            FMX_CHECK((s >> 8) < n / 256u + 1u && (e >> 8) < n / 256u + 1u);
            FMX_TOUCH_G0(g, &rec1[(size_t)(s >> 8) * 8u]);
            if ((e >> 8) != (s >> 8)) FMX_TOUCH_G0(g, &rec1[(size_t)(e >> 8) * 8u]);
            const uint4 a1 = rec1[(size_t)(s >> 8) * 8u + g];
            const uint4 b1 = rec1[(size_t)(e >> 8) * 8u + g];
            const uint32_t s1 = fmx_group_sum(fmx_piece_rank<3>(a1, s & 255u, c2, g));
            const uint32_t e1 = fmx_group_sum(fmx_piece_rank<3>(b1, e & 255u, c2, g));
            if (s1 == e1) { ns = s1; ne = e1; used = 1; }
          }
          s = ns; e = ne;
        } else {
          FMX_CHECK((s >> 8) < n / 256u + 1u && (e >> 8) < n / 256u + 1u);
          FMX_TOUCH_G0(g, &rec1[(size_t)(s >> 8) * 8u]);
          if ((e >> 8) != (s >> 8)) FMX_TOUCH_G0(g, &rec1[(size_t)(e >> 8) * 8u]);
          const uint4 a1 = rec1[(size_t)(s >> 8) * 8u + g];
          const uint4 b1 = rec1[(size_t)(e >> 8) * 8u + g];
          const uint32_t s1 = fmx_group_sum(fmx_piece_rank<3>(a1, s & 255u, c2, g));  // wrapper.rs:109
          const uint32_t e1 = fmx_group_sum(fmx_piece_rank<3>(b1, e & 255u, c2, g));  // wrapper.rs:110
          s = s1; e = e1;
          used = 1;
        }
        nsteps += used;
        j -= used;
        if (used == 2) { c2 = n1; c1 = n2; } else { c2 = c1; c1 = n1; }
        if (s == e || j == 0) done = true;             // wrapper.rs:111-113
      }
    }
    if (done) {
      if (g == 0) {
        if (out_s) out_s[k] = s;
        if (out_e) out_e[k] = e;
        if (out_cnt) out_cnt[k] = (uint64_t)(e - s);
      }
      k += ngroups;
      active = k < npat;
      fresh = true;
    }
  }
  if (steps_out && g == 0 && nsteps)
    atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// ---------------------------------------------------------------------------
// count, endpoint per lane (fmx_ep.h): lane 2q carries s and lane 2q+1 carries e of the group's q-th
// pattern, so a group advances 4 patterns and a wave 32, with 64 probes in flight in every dependent
// stage of a step (RLFM: B piece, S level 0, S level 1, B' select; FM over several wavelet levels:
// one record per level).
// Same state machine as above: a pattern that ends (all symbols consumed, or the `s == e` break of
// wrapper.rs:111-113) is replaced at once; the loop is wave-uniform because the rank rounds are.
// SM = 1 (stored positions) or 2 (select blocks); indexes whose B / B' fall into neither class stay
// on fmx_count_kernel<FMX_KIND_RLFM>.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool fmx_kmer_code_lane(const uint8_t *__restrict__ pat, uint64_t pend, uint32_t kk,
                                                   uint32_t bits, uint32_t max_character, uint32_t &code) {
  uint32_t bad = 0;
  code = 0;
  for (uint32_t t = 0; t < kk; t++) {                     // t-th symbol from the back
    const uint32_t cc = pat[pend - 1u - t];
    bad |= (uint32_t)((cc - 1u) >= max_character);
    code |= ((cc - 1u) & ((1u << bits) - 1u)) << (bits * (kk - 1u - t));
  }
  return bad == 0u;
}
template <int KIND, int NL, int SM, bool KM>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_count_ep_kernel(
    FmxDev ix, const void *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, base = lane & ~7u;
  const uint32_t is_e = g & 1u;
  // pattern slots are dealt to the GROUPS first (slot = pair * ngroups + group): a batch smaller than
  // the grid then has one live lane pair per group, the rank rounds skip the dead endpoints
  // (fmx_ep_round) and a step costs one record per level instead of eight -- small batches are
  // latency-bound, and this is what keeps them at the group-per-pattern kernels' latency
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 3;
  const uint64_t slot = (uint64_t)(g >> 1) * ngroups + (((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3);
  const uint64_t nslots = ngroups * 4u;
  const uint64_t ptot = npat ? off[npat] : 0;   // symbols the caller declares behind `pat`
  const uint64_t pmin = npat ? off[0] : 0;      // ... starting at this symbol (a slice of a larger batch keeps its absolute offsets)

  uint64_t k = slot, pbeg = 0;
  bool active = k < npat, fresh = true;
  uint32_t j = 0, pos = 0, c = 0, nsteps = 0;
  while (__any(active)) {
    if (active && fresh) {
      pbeg = off[k];
      const uint64_t pend = off[k + 1];
      j = (uint32_t)(pend - pbeg);
      bool bad = pend < pbeg || pbeg < pmin || pend > ptot || pend - pbeg > 0xFFFFFFFFull;
      if (s0e0) {            // Search::search on an existing Search (wrapper.rs:105-106)
        const uint64_t mine = s0e0[2 * k + is_e], other = s0e0[2 * k + (is_e ^ 1u)];
        pos = (uint32_t)mine;
        bad |= mine > ix.n || other > ix.n;              // not a range of this index
      } else {               // SearchIndexWrapper::search: (0, len)   (wrapper.rs:41)
        pos = is_e ? ix.n : 0u;
        if (KM && !bad && j >= ix.kmer_k) {              // the first kmer_k steps from the table
          uint32_t code;
          if (fmx_kmer_code_lane((const uint8_t *)pat, pbeg + j, ix.kmer_k, ix.kmer_bits, ix.max_character,
                                 code)) {
            FMX_TOUCH(&ix.kmer[code]);
            const uint2 se = ix.kmer[code];
            pos = is_e ? se.y : se.x;
            j = se.x == se.y ? 0u : j - ix.kmer_k;       // empty already: the reference's break
            nsteps += is_e ? 0u : ix.kmer_k;
          }
        }
      }
      if (bad) {             // refuse, do not read
        if (is_e) atomicOr(ix.status, 1u << FMX_ERR_ARG);
        pos = 0; j = 0;
      }
      c = j ? fmx_load_sym(pat, ix.sym_bytes, pbeg + j - 1) : 0u;  // pattern.iter().rev()  wrapper.rs:108
      fresh = false;
    }
    bool stepping = active && j != 0;
    if (stepping && c > ix.max_character) {              // reference: panic on cs[c]
      if (is_e) atomicOr(ix.status, 1u << FMX_ERR_SYMBOL_RANGE);
      pos = 0; j = 0; stepping = false;
    }
    // the next symbol rides along with this step's probes
    const uint32_t cn = (stepping && j > 1) ? fmx_load_sym(pat, ix.sym_bytes, pbeg + j - 2) : 0u;
    const uint32_t np = KIND == FMX_KIND_RLFM
                            ? fmx_rlfm_ep_lf_map2<NL, (SM > 0 ? SM : 1)>(ix, stepping ? c : 0u, stepping ? pos : 0u, stepping, base, g)
                            : fmx_fm_ep_lf_map2<NL>(ix, stepping ? c : 0u, stepping ? pos : 0u, stepping, base, g);  // wrapper.rs:109-110
    if (stepping) {
      pos = np;
      c = cn;
      j--;
      nsteps += is_e ^ 1u;
    }
    const uint32_t other = fmx_dpp_xor1(pos);            // the pattern's other interval end
    if (active && (j == 0 || pos == other)) {            // wrapper.rs:111-113
      if (is_e) {
        if (out_e) out_e[k] = pos;
        if (out_cnt) out_cnt[k] = (uint64_t)(pos - other);   // wrapper.rs:132-134
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

// ---------------------------------------------------------------------------
// locate
// ---------------------------------------------------------------------------
// exclusive offsets -> rows: rows[off[k] + j] = s[k] + j   (wrapper.rs:203-217: i = s..e-1
// ascending).  One LANE per pattern writes short ranges itself; ranges longer than 32 rows are
// written by the whole wave, one after the other (ballot over the lanes that hold one).
// Ranges of FMX_EXPAND_DEFER rows or more are not written by their pattern's wave (a wave sustains ~17 GB/s of such
// stores: ten patterns of 10^7 hits among 10^6 singletons kept ten waves busy for 3 ms) but listed in
// `longs` -- longs[0] = entries, then {first row, first slot, rows} each, FMX_EXPAND_LONGCAP at most (a range that finds
// the list full is written here after all) -- for fmx_expand_long_kernel, which writes every listed range with the
// whole grid.  The choice is per RANGE, not per batch average.  longs == NULL: every range is written here.
#define FMX_EXPAND_DEFER (1u << 16)
#define FMX_EXPAND_LONGCAP 4096u
template <typename T>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_expand_kernel(
    const uint64_t *__restrict__ s, const uint64_t *__restrict__ e,
    const uint64_t *__restrict__ off, uint64_t npat, T *__restrict__ out_pos, uint64_t total,
    uint64_t n, uint32_t *status, unsigned long long *__restrict__ longs = nullptr) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t base = first - lane; base < npat; base += stride) {   // wave-uniform trip count
    const uint64_t k = base + lane;
    uint64_t a = 0, o = 0, cnt = 0;
    if (k < npat) {
      a = s[k];
      const uint64_t b = e[k];
      o = off[k];
      cnt = b > a ? b - a : 0;
      // a range that is not one of this index, or offsets that do not leave room for it: refuse
      // (row 0 is written so that the walk stays inside the index)
      if (b > n || o > total || cnt > total - o) {
        atomicOr(status, 1u << FMX_ERR_ARG);
        a = 0;
        cnt = o < total ? (cnt < total - o ? cnt : total - o) : 0;
        if (cnt > n) cnt = n;
      }
      // offsets that leave a gap (before the first range, between two ranges, behind the last):
      // reported, and the gap is filled with row 0 so that the walk never sees an unwritten slot
      uint64_t gap_end = k + 1 < npat ? off[k + 1] : total;
      if (gap_end > total) gap_end = total;
      const uint64_t used_end = o < total ? o + cnt : total;
      if (gap_end > used_end || (k == 0 && o != 0)) {
        atomicOr(status, 1u << FMX_ERR_ARG);
        for (uint64_t t = used_end; t < gap_end; t++) out_pos[t] = (T)0;
        if (k == 0) for (uint64_t t = 0; t < (o < total ? o : total); t++) out_pos[t] = (T)0;
      }
    }
    if (cnt <= 32) {
      for (uint64_t t = 0; t < cnt; t++) out_pos[o + t] = (T)(a + t);
    }
    unsigned long long big = __ballot(cnt > 32);
    while (big) {
      const int l = __ffsll((long long)big) - 1;
      big &= big - 1;
      const uint64_t A = ((uint64_t)(uint32_t)__shfl((int)(a >> 32), l) << 32) | (uint32_t)__shfl((int)a, l);
      const uint64_t O = ((uint64_t)(uint32_t)__shfl((int)(o >> 32), l) << 32) | (uint32_t)__shfl((int)o, l);
      const uint64_t N = ((uint64_t)(uint32_t)__shfl((int)(cnt >> 32), l) << 32) | (uint32_t)__shfl((int)cnt, l);
      if (longs && N >= FMX_EXPAND_DEFER) {          // wave-uniform
        unsigned long long q = 0;
        if (lane == 0) q = atomicAdd(&longs[0], 1ull);
        q = ((unsigned long long)(uint32_t)__shfl((int)(q >> 32), 0) << 32) | (uint32_t)__shfl((int)q, 0);
        if (q < FMX_EXPAND_LONGCAP) {
          if (lane == 0) { longs[2u + 3u * q] = A; longs[3u + 3u * q] = O; longs[4u + 3u * q] = N; }
          continue;
        }
      }
      // 16-byte stores (4 rows of 32 bits / 2 of 64) between a scalar head up to the first aligned slot and a scalar
      // tail: a wave writes 1 KB per instruction instead of 256 bytes (config 4b, 750 hits per pattern: the expansion
      // of its 7.9e8 rows was 1.03 ms of the 8.2 ms batch)
      constexpr uint32_t V = 16u / (uint32_t)sizeof(T);
      const uint64_t mis = (((uintptr_t)(out_pos + O)) / sizeof(T)) & (V - 1u);
      const uint64_t h = (V - mis) & (V - 1u);       // rows in front of the first 16-byte slot (N > 32 > h)
      if (lane < h) out_pos[O + lane] = (T)(A + lane);
      const uint64_t nv = (N - h) / V;
      for (uint64_t q = lane; q < nv; q += 64) {
        const uint64_t t = h + q * V;
        if constexpr (V == 4u) {
          const uint32_t r = (uint32_t)(A + t);
          *reinterpret_cast<uint4 *>(out_pos + O + t) = make_uint4(r, r + 1u, r + 2u, r + 3u);
        } else {
          *reinterpret_cast<ulonglong2 *>(out_pos + O + t) = make_ulonglong2(A + t, A + t + 1u);
        }
      }
      const uint64_t done = h + nv * V;
      if (lane < N - done) out_pos[O + done + lane] = (T)(A + done + lane);
    }
  }
}

__global__ void fmx_zero_words_kernel(unsigned long long *p, uint32_t nwords) {
  if (threadIdx.x < nwords) p[threadIdx.x] = 0ull;
}
// the listed long ranges (fmx_expand_kernel), every one of them by the whole grid: 16-byte stores between a scalar head
// and tail, as above.  An empty list costs the launch (~2 us).
template <typename T>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_expand_long_kernel(const unsigned long long *__restrict__ longs,
                                                                    T *__restrict__ out_pos) {
  unsigned long long nl = longs[0];
  if (nl > FMX_EXPAND_LONGCAP) nl = FMX_EXPAND_LONGCAP;
  constexpr uint32_t V = 16u / (uint32_t)sizeof(T);
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  for (unsigned long long i = 0; i < nl; i++) {
    const uint64_t A = longs[2u + 3u * i], O = longs[3u + 3u * i], N = longs[4u + 3u * i];
    const uint64_t mis = (((uintptr_t)(out_pos + O)) / sizeof(T)) & (V - 1u);
    const uint64_t h = (V - mis) & (V - 1u);         // rows in front of the first 16-byte slot (N >= 2^16 > h)
    if (tid < h) out_pos[O + tid] = (T)(A + tid);
    const uint64_t nv = (N - h) / V;
    for (uint64_t q = tid; q < nv; q += nth) {
      const uint64_t t = h + q * V;
      if constexpr (V == 4u) {
        const uint32_t r = (uint32_t)(A + t);
        *reinterpret_cast<uint4 *>(out_pos + O + t) = make_uint4(r, r + 1u, r + 2u, r + 3u);
      } else {
        *reinterpret_cast<ulonglong2 *>(out_pos + O + t) = make_ulonglong2(A + t, A + t + 1u);
      }
    }
    const uint64_t done = h + nv * V;
    if (tid < N - done) out_pos[O + done + tid] = (T)(A + done + tid);
  }
}

// get_sa(row) with text-order sampling (FmxDev::phase), for an 8-lane group that holds one row: the
// row's phase says how many LF steps lead to a sampled row (never more than 2^level - 1), and the
// rank over the phase-0 rows at that row is the index of its sample.  No wrap can occur: the walk
// goes from text position SA[row] down to SA[row] - phase >= 0.
template <int KIND, int NL, int SM>
__device__ __forceinline__ uint64_t fmx_get_sa_text(const FmxDev &ix, uint32_t row, uint32_t g,
                                                    uint32_t &nsteps) {
  uint32_t t, rank0;
  uint32_t p = fmx_phase_piece(row, ix.sa_level, t);
  FMX_TOUCH_G0N(g, &ix.phase[p]);
  const uint32_t phi = fmx_phase_decode(ix.phase[p], t, ix.sa_level, rank0);
  for (uint32_t k = 0; k < phi; k++) {               // i = lf_map(i); steps += 1     fm_index.rs:134-137
    row = fmx_lf_step_any<KIND, NL, SM>(ix, row, g);
  }
  nsteps += phi;
  if (phi) {
    p = fmx_phase_piece(row, ix.sa_level, t);
    FMX_TOUCH_G0N(g, &ix.phase[p]);
    (void)fmx_phase_decode(ix.phase[p], t, ix.sa_level, rank0);
  }
  FMX_CHECK(rank0 < ix.nsamples);
  FMX_TOUCH_G0N(g, &ix.samples[rank0]);
  return (uint64_t)ix.samples[rank0] + phi;          // fm_index.rs:131-133 (sa + steps)
}

// generic locate walk (any kind / any number of levels): same wave-level dynamic hit assignment
// and register row window as fmx_locate_f3w_kernel below; one LF step = fmx_lf_map_any (several
// dependent probes), the sample read is a plain dependent load.
template <int KIND, int NL, int SM = -1>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_locate_kernel(
    FmxDev ix, uint64_t total, uint64_t hits_per_wave, const uint32_t *__restrict__ rows,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane & (FMX_GROUP - 1);
  const uint32_t grp = lane >> 3;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lmask = (1u << ix.sa_level) - 1u;
  uint64_t w0 = wave * hits_per_wave;
  if (w0 >= total) return;                          // wave-uniform
  uint64_t w1 = w0 + hits_per_wave < total ? w0 + hits_per_wave : total;
  uint64_t win_base = w0;
  // rows come from fmx_expand_kernel, which writes every slot (gaps left by inconsistent offsets
  // are filled with row 0 and reported), so every row is inside the index
  auto load_row = [&](uint64_t x) -> uint32_t { return rows[x < total ? x : total - 1]; };
  uint32_t win = load_row(win_base + lane);
  uint64_t h = w0 + grp;
  bool active = h < w1;
  uint32_t row = (uint32_t)__shfl((int)win, (int)grp);
  uint64_t next = w0 + 8 < w1 ? w0 + 8 : w1;
  uint32_t steps = 0, nsteps = 0;
  while (__any(active)) {
    if (next + 8 > win_base + 64 && next < w1) {    // wave-uniform window refill
      win_base = next;
      win = load_row(win_base + lane);
    }
    bool fin = false;
    if (active && ix.phase) {                       // text-order sampling: the whole (short) walk at once
      const uint64_t v = fmx_get_sa_text<KIND, NL, SM>(ix, row, g, nsteps);
      if (g == 0) out_pos[h] = v;
      fin = true;
    } else if (active) {
      if ((row & lmask) == 0) {
        // sample.rs:46-60 Some(sa): fm_index.rs:131-133  (sa + steps) % len
        FMX_CHECK((row >> ix.sa_level) < ix.nsamples);
        FMX_TOUCH_G0N(g, &ix.samples[row >> ix.sa_level]);
        uint64_t v = (uint64_t)ix.samples[row >> ix.sa_level] + steps;
        if (v >= ix.n) v -= ix.n;  // steps < n, sa < n
        if (g == 0) out_pos[h] = v;
        fin = true;
      } else {
        // None: i = lf_map(i); steps += 1      fm_index.rs:134-137
        row = fmx_lf_step_any<KIND, NL, SM>(ix, row, g);
        steps++;
        nsteps++;
      }
    }
    const unsigned long long fmask = __ballot(fin && g == 0);
    if (fmask) {                                                  // wave-uniform
      const uint32_t leader = lane & ~7u;
      const uint32_t my_rank = (uint32_t)__popcll(fmask & ((1ull << leader) - 1ull));
      const uint64_t h_new = next + my_rank;
      const uint32_t r_new = (uint32_t)__shfl((int)win, (int)((h_new - win_base) & 63u));
      if (fin) {
        h = h_new;
        active = h < w1;
        row = r_new;
        steps = 0;
      }
      next += (uint64_t)__popcll(fmask);
    }
  }
  if (steps_out && g == 0 && nsteps)
    atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// ---------------------------------------------------------------------------
// work queue of the locate kernels.  With a private chunk of hits per wave, a wave ran as long as its
// unluckiest slot (walk lengths are geometric: 56 % efficiency at 2^20 hits).  A global ticket
// counter fixes the balance but not the clock: device-scope atomics on one address are served one
// after the other at the memory side, and 2^14 tickets cost more than they saved (measured: DNA
// locate 0.153 -> 0.25 ms).  So the queue is per WORKGROUP, in LDS: a 1024-thread block owns a
// contiguous slice of the hits (large enough for the law of large numbers to balance the blocks:
// 4096 hits -> +-1.4 % of work) and its 16 waves draw 64-row chunks from an LDS counter as they
// run dry.  A wave keeps two chunks resident (their 64 rows each in one register per lane).
// ---------------------------------------------------------------------------
#define FMX_LCHUNK 64u
#define FMX_NOCHUNK 0xFFFFFFFFu
#define FMX_LOC_BLOCK 1024
// LWIN: the rows of the two resident chunks live in LDS (2 x 64 words per wave, `lwin`) instead of two
// registers per lane -- for the one-walk-per-lane kernel, which is short of registers; handing out a hit is
// then one ds_read instead of two ds_bpermutes.
// TL (round 5): the tickets of the slice this queue hands out are listed in `tlist` (nlist entries, LDS) -- the unified DNA
// walk kernel serves the other tickets of its slice another way -- and a draw maps its ticket number through that list.
// runs of consecutive rows in a ticket, minus one: the lanes (of the `in` ones) whose row is not its left neighbour's + 1
__device__ __forceinline__ uint32_t fmx_ticket_breaks(uint32_t r, bool in, uint32_t lane) {
  const uint32_t prev = (uint32_t)__shfl_up((int)r, 1);
  return (uint32_t)__popcll(__ballot(in && lane != 0u && r != prev + 1u));
}
template <bool LWIN = false, bool TL = false>
struct FmxHitQueue {
  const uint32_t *rows;   // rows of this block's slice
  uint64_t lo;            // first hit of the slice (index into out_pos)
  uint32_t nhits;         // hits in the slice
  uint32_t chunk;         // rows per ticket: 64, or 8..56 when the batch has fewer than 64 hits per wave
                          // (every slot of every wave then starts a walk at once -- mid-size batches are
                          // latency-bound, and two walks in a row per slot are twice the chain)
  uint32_t lane;
  uint32_t c0, c1;        // resident chunks of the slice (FMX_NOCHUNK once it is exhausted), wave-uniform
  uint32_t win0, win1;    // rows of the resident chunks, one per lane (lanes >= chunk unused)     [!LWIN]
  volatile fmx_lds_u32 *lwin;  // this wave's 2 x 64 words of LDS; chunk c0 sits in half w0, c1 in the other [LWIN]
  uint32_t w0;
  uint32_t used;          // hits already handed out of c0|c1
  const uint16_t *tlist = nullptr;   // [TL]
  uint32_t nlist = 0;
  // a ticket and its rows (`win`: one per lane)
  __device__ __forceinline__ uint32_t draw_win(unsigned int &counter, uint32_t &win) const {
    const uint32_t c = valid(draw(counter, 1u));
    win = load_win(c);
    return c;
  }
  __device__ __forceinline__ uint32_t load_win(uint32_t c) const {
    const uint32_t x = c * chunk + lane;
    return (c != FMX_NOCHUNK && lane < chunk && x < nhits) ? rows[x] : 0u;   // every slot was written by fmx_expand_kernel
  }
  __device__ __forceinline__ uint32_t valid(uint32_t c) const {
    return (c != FMX_NOCHUNK && c * chunk < nhits) ? c : FMX_NOCHUNK;
  }
  __device__ __forceinline__ uint32_t draw(unsigned int &counter, uint32_t k) const {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&counter, k);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (TL) t = t < nlist ? (uint32_t)tlist[t] : FMX_NOCHUNK;
    return t;
  }
  __device__ __forceinline__ void init(const uint32_t *r, uint64_t first, uint32_t count, uint32_t rows_per_ticket,
                                       uint32_t ln, unsigned int &counter, volatile fmx_lds_u32 *lds_win = nullptr) {
    rows = r; lo = first; nhits = count; chunk = rows_per_ticket; lane = ln;
    lwin = lds_win; w0 = 0; win0 = 0; win1 = 0;
    // every wave of the block draws its FIRST ticket before any draws a second one (called by all
    // threads of the block at kernel start): with fewer tickets than 2 x waves, no wave goes without
    uint32_t wn;
    c0 = draw_win(counter, wn);
    if (LWIN) lwin[lane] = wn; else win0 = wn;
    __syncthreads();
    c1 = draw_win(counter, wn);
    if (LWIN) lwin[64u + lane] = wn; else win1 = wn;
    used = 0;
  }
  // the hit with index `used + rank` (< 2 * chunk): returns false when the slice has run dry
  __device__ __forceinline__ bool take(uint32_t rank, uint64_t &h, uint32_t &row) const {
    uint32_t x;
    const bool ok = take32(rank, x, row);
    h = lo + x;
    return ok;
  }
  // the same with the hit as an index into the slice (out_pos index = lo + x): 32-bit state for the kernels
  // that keep one walk per lane
  __device__ __forceinline__ bool take32(uint32_t rank, uint32_t &x, uint32_t &row) const {
    bool first;
    return take32(rank, x, row, first);
  }
  // `first`: the hit comes from the older resident chunk c0 (else from c1)
  __device__ __forceinline__ bool take32(uint32_t rank, uint32_t &x, uint32_t &row, bool &first) const {
    const uint32_t idx = used + rank;
    first = idx < chunk;
    const uint32_t c = first ? c0 : c1, within = first ? idx : idx - chunk;
    if (LWIN) {
      row = lwin[((first ? w0 : w0 ^ 1u) << 6) + (within & 63u)];
    } else {
      const uint32_t v0 = (uint32_t)__shfl((int)win0, (int)(within & 63u));
      const uint32_t v1 = (uint32_t)__shfl((int)win1, (int)(within & 63u));
      row = first ? v0 : v1;
    }
    x = c * chunk + within;
    return idx < 2u * chunk && c != FMX_NOCHUNK && x < nhits;
  }
  // `count` hits were handed out (wave-uniform, count <= chunk)
  // returns true when the window slid: c1 became c0 and a new c1 was drawn (wave-uniform)
  __device__ __forceinline__ bool advance(uint32_t count, unsigned int &counter) {
    FMX_CHECK(count <= chunk && used < chunk);       // at most one slide per call
    used += count;
    if (used >= chunk) {                             // wave-uniform: slide
      used -= chunk;
      c0 = c1;
      uint32_t wn = 0u;
      c1 = c1 != FMX_NOCHUNK ? draw_win(counter, wn) : FMX_NOCHUNK;
      if (LWIN) {
        lwin[(w0 << 6) + lane] = wn;                 // the half the old c0 occupied
        w0 ^= 1u;
      } else {
        win0 = win1;
        win1 = wn;
      }
      return true;
    }
    return false;
  }
};


// locate walk, single 3-bit level (DNA), walk state DISTRIBUTED over the lanes of a group.  A group of 8
// lanes still serves Q walks at a time with one 128-byte record per LF step (fm_index.rs:134-137), but the
// state of walk q (row, steps, hit index, stage) lives in lane q of each quad of the group instead of being
// repeated in all 8 lanes: the sampled test, the sample read (sample.rs:46-60), the position
// (fm_index.rs:131-133), the hand-over of the next hit and the deferred store are executed ONCE per
// iteration for all Q walks -- lane-wise -- and only the record load + decode run per walk, with the row
// broadcast inside the quad (DPP quad_perm).  fmx_locate_f3w_kernel repeats that bookkeeping per walk in
// every lane and is bound by vector-instruction issue (100 instructions per record load,
// profiles/r02/sweeps.md); this form issues about half.
// TEXT: text-order sampling (FmxDev::phase) -- the phase pieces are lane-wise 16-byte probes like the sample.
template <int Q, bool TEXT>
__global__ __launch_bounds__(FMX_LOC_BLOCK) void fmx_locate_f3q_kernel(
    const uint4 *__restrict__ rec, const uint32_t *__restrict__ samples, const uint4 *__restrict__ phase,
    uint32_t n, uint32_t sa_level, uint64_t total, uint32_t hits_per_block, uint32_t chunk,
    const uint32_t *__restrict__ rows, uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  static_assert(Q == 1 || Q == 2 || Q == 4, "walks per group");
  __shared__ unsigned int lds_q;
  if (threadIdx.x == 0) lds_q = 0;
  __syncthreads();
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane & (FMX_GROUP - 1);
  const uint32_t grp = lane >> 3;
  const uint32_t slot = g & (uint32_t)(Q - 1);        // the walk whose state this lane keeps (8 / Q replicas)
  const bool owner = g < (uint32_t)Q;                 // the replica that counts, stores and is counted
  const uint32_t olane = (lane & ~7u) | slot;         // its lane
  // lanes holding walk 0 of their group: one bit per replica
  constexpr unsigned long long SLOT0 = Q == 4 ? 0x1111111111111111ull : Q == 2 ? 0x5555555555555555ull
                                                                                : 0xFFFFFFFFFFFFFFFFull;
  constexpr uint32_t NONE = 0xFFFFFFFFu;              // no row (n < 2^32 - 16)
  const uint32_t lmask = (1u << sa_level) - 1u;
  FmxHitQueue<> hq;
  hq.init(rows + blo, blo, bn, chunk, lane, lds_q);
  // the first 8 hits go to walk 0 of the wave's 8 groups, the next 8 to walk 1, ...: a wave that gets few
  // hits has few live walk slots and skips the others' loads and decodes
  uint64_t h;
  uint32_t row;
  bool active = hq.take((slot << 3) | grp, h, row);
  hq.advance(8u * (uint32_t)Q, lds_q);
  if (!active) row = 0u;
  uint32_t steps = 0, nsteps = 0;
  // TEXT: stage of the walk (0 phase piece of the start row, 1 LF steps, 2 phase piece of the final row
  // for its rank among the sampled rows, 3 the sample), LF steps still to do, index of the sample
  [[maybe_unused]] uint32_t st = 0, rem = 0, sidx = 0;
  bool pending = false;
  uint64_t pend_h = 0, pend_v = 0;
  for (;;) {
    if (!__any(active || pending)) break;
    const bool probing = TEXT && active && (st == 0u || st == 2u);
    const bool sampled = active && (TEXT ? st == 3u : (row & lmask) == 0u);
    const bool walking = active && !sampled && !probing;
    // lane-wise probes of this round: the sample of a sampled row, the phase piece of a start / final row
    uint32_t sa = 0;
    [[maybe_unused]] uint32_t pt = 0;
    [[maybe_unused]] uint4 pc = make_uint4(0u, 0u, 0u, 0u);
    if (TEXT && probing) {
      const uint32_t pi = fmx_phase_piece(row, sa_level, pt);
      if (owner) FMX_TOUCH(&phase[pi]);
      pc = phase[pi];
    }
    if (sampled) {                                    // sample.rs:46-60 Some(sa)
      const uint32_t si = TEXT ? sidx : row >> sa_level;
      FMX_CHECK((uint64_t)si <= (((uint64_t)n - 1) >> sa_level));
      if (owner) FMX_TOUCH(&samples[si]);
      sa = samples[si];
    }
    // one record per walking walk: the row goes from the lane that keeps it to its quad
    const uint32_t rowx = walking ? row : NONE;
    const unsigned long long wm = __ballot(walking);
    uint4 p[Q];
    uint32_t rq[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      rq[q] = NONE;
      if (!(wm & (SLOT0 << q))) continue;             // walk q idle in every group of the wave
      rq[q] = Q == 1 ? rowx : fmx_quad_bcast(rowx, q);
      if (rq[q] != NONE) {                            // group-uniform
        FMX_CHECK(rq[q] < n && (rq[q] >> 8) < n / 256u + 1u);
        const uint4 *addr = rec + ((size_t)(rq[q] >> 8) * 8u + g);
        FMX_TOUCH_G0(g, addr - g);
        p[q] = *addr;
      }
    }
    // the position finished in the previous round is stored behind these loads
    if (pending && owner) out_pos[pend_h] = pend_v;
    pending = false;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (!(wm & (SLOT0 << q))) continue;
      if (rq[q] != NONE) {                            // None: i = lf_map(i); steps += 1   fm_index.rs:134-137
        const uint32_t off = rq[q] & 255u;
        const uint32_t sym = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p[q], off & 31u) : 0u);
        const uint32_t nr = fmx_group_sum(fmx_piece_rank<3>(p[q], off, sym, g));   // absolute counters
        if (slot == (uint32_t)q) {
          row = nr;
          steps++;
          if (TEXT && --rem == 0u) st = 2u;
        }
      }
    }
    if (TEXT && probing) {                            // phase piece: of the start row (0) or the final row (2)
      uint32_t rank0;
      const uint32_t phi = fmx_phase_decode(pc, pt, sa_level, rank0);
      sidx = rank0;
      rem = phi;
      st = (st == 2u || phi == 0u) ? 3u : 1u;
    }
    const unsigned long long fm = __ballot(sampled && owner);   // one bit per finishing walk
    if (fm) {                                         // wave-uniform
      uint64_t h_new;
      uint32_t r_new;
      const bool ok = hq.take((uint32_t)__popcll(fm & ((1ull << olane) - 1ull)), h_new, r_new);
      if (sampled) {
        uint64_t v = (uint64_t)sa + steps;            // fm_index.rs:131-133: (sa + steps) % len
        if (v >= n) v -= n;
        pend_v = v;
        pend_h = h;
        pending = true;
        nsteps += steps;
        h = h_new;
        active = ok;
        row = ok ? r_new : 0u;
        steps = 0;
        st = 0;
      }
      hq.advance((uint32_t)__popcll(fm), lds_q);
    }
  }
  if (steps_out && owner && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// The same walk for row-order sampling, with the hand-over moved to the TOP of the round: a walk that
// stands on a sampled row gives its slot to the next hit at once and its sample (sample.rs:46-60) is read
// in the same round as the new walk's first record, so a hit costs `steps` rounds of its slot instead of
// steps + 1 (3.25 against 4 at level 2: the fresh hit that is itself sampled -- one in 2^level -- idles a round).
// WC: the positions go through a write-combining ring in LDS.  A wave's hits come in tickets of 64
// consecutive hit indices = 512 contiguous bytes of out_pos, but walks finish in any order, and an 8-byte
// store on its own is one 32-byte sector and one request at the memory side (config 3b: 2.9e8 of them, 40 %
// of the kernel's requests).  Each wave keeps its last FMX_WC_SLOTS tickets as 64 x u32 in LDS; a finished
// walk whose ticket is still there drops its position into it, and when the slot is recycled for a new
// ticket the wave stores what has arrived with ONE contiguous 64 x 8-byte store.  Walks that outlive
// their ticket's residency (~4 tickets = ~26 rounds) store directly as before.  Needs 64-hit tickets.
#define FMX_WC_SLOTS 4
template <int Q, bool WC>
__global__ __launch_bounds__(FMX_LOC_BLOCK) void fmx_locate_f3p_kernel(
    const uint4 *__restrict__ rec, const uint32_t *__restrict__ samples, uint32_t n, uint32_t sa_level,
    uint64_t total, uint32_t hits_per_block, uint32_t chunk, const uint32_t *__restrict__ rows,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  static_assert(Q == 1 || Q == 2 || Q == 4 || Q == 8, "walks per group");
  __shared__ unsigned int lds_q;
  __shared__ uint32_t wc_ring[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS * 64 : 1];
  __shared__ uint32_t wc_tag[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS : 1];
  if (threadIdx.x == 0) lds_q = 0;
  __syncthreads();
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane & (FMX_GROUP - 1);
  const uint32_t grp = lane >> 3;
  const uint32_t slot = g & (uint32_t)(Q - 1);        // the walk whose state this lane keeps (8 / Q replicas)
  const bool owner = g < (uint32_t)Q;                 // the replica that counts, stores and is counted
  const uint32_t olane = (lane & ~7u) | slot;         // its lane
  constexpr unsigned long long SLOT0 = Q == 8   ? 0x0101010101010101ull
                                       : Q == 4 ? 0x1111111111111111ull
                                       : Q == 2 ? 0x5555555555555555ull
                                                : 0xFFFFFFFFFFFFFFFFull;
  constexpr uint32_t NONE = 0xFFFFFFFFu;              // no row / no sample / no position (n < 2^32 - 16)
  const uint32_t lmask = (1u << sa_level) - 1u;
  FmxHitQueue<> hq;
  hq.init(rows + blo, blo, bn, chunk, lane, lds_q);
  uint64_t *const out = out_pos + blo;                // the block's slice of the output (wave-uniform)
  // write-combining ring of this wave: slot r holds ticket ring_tag[r]; entry i of it = position of hit
  // 64 * ticket + i (NONE until its walk has finished).  Only this wave touches its ring.
  // (volatile: lanes read what other lanes of the wave wrote -- every access must be a real LDS operation,
  // in program order; the LDS serves one wave's operations in order)
  [[maybe_unused]] volatile fmx_lds_u32 *const ring = FMX_LDS_U32(wc_ring + (threadIdx.x >> 6) * (FMX_WC_SLOTS * 64));
  [[maybe_unused]] volatile fmx_lds_u32 *const ring_tag = FMX_LDS_U32(wc_tag + (threadIdx.x >> 6) * FMX_WC_SLOTS);
  [[maybe_unused]] uint32_t rs0 = 0, rs1 = 1, rseq = 2;   // ring slots of the resident tickets c0 / c1; tickets drawn
  if (WC) {
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) ring[r * 64u + lane] = NONE;
    if (lane < FMX_WC_SLOTS) ring_tag[lane] = lane == 0 ? hq.c0 : (lane == 1 ? hq.c1 : FMX_NOCHUNK);
  }
  uint32_t hx, row;                                   // hit (index into the slice) and current row of the walk
  [[maybe_unused]] uint32_t myslot = 0;               // WC: ring slot of the ticket the hit came from
  bool active = hq.take32((slot << 3) | grp, hx, row);   // first 8 hits -> walk 0 of the 8 groups, ... (all of c0)
  {
    const bool slid = hq.advance(8u * (uint32_t)Q, lds_q);
    if (WC && slid) {                                 // Q = 8 hands out a whole ticket at once
      const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
      if (lane == 0) ring_tag[ns] = hq.c1;            // slot ns is still empty (all NONE)
      rseq++; rs0 = rs1; rs1 = ns;
    }
  }
  if (!active) row = 0u;
  uint32_t steps = 0, nsteps = 0;
  for (;;) {
    if (!__any(active)) break;                        // nothing is left over: a finished walk is completed below
    // walks standing on a sampled row: the slot goes to the next hit, the walk is completed in this round
    const bool done = active && (row & lmask) == 0u;
    const unsigned long long fm = __ballot(done && owner);     // one bit per finishing walk
    uint32_t fin_si = NONE, fin_steps = 0, fin_x = 0;
    [[maybe_unused]] uint32_t fin_slot = 0;
    if (fm) {                                         // wave-uniform
      uint32_t x_new, r_new;
      bool first;
      const bool ok = hq.take32((uint32_t)__popcll(fm & ((1ull << olane) - 1ull)), x_new, r_new, first);
      if (done) {
        fin_si = row >> sa_level;
        fin_steps = steps;
        fin_x = hx;
        fin_slot = myslot;
        nsteps += steps;
        hx = x_new;
        myslot = first ? rs0 : rs1;
        active = ok;
        row = ok ? r_new : 0u;
        steps = 0;
      }
      const bool slid = hq.advance((uint32_t)__popcll(fm), lds_q);
      if (WC && slid) {
        // a new ticket was drawn: it gets the ring slot of the oldest one, whose arrived positions leave
        // now in one contiguous store (hit index = 64 * ticket + lane)
        const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
        const uint32_t old_tag = ring_tag[ns];
        const uint32_t v = ring[ns * 64u + lane];
        FMX_CHECK(old_tag == FMX_NOCHUNK || v == NONE || old_tag * 64u + lane < bn);
        if (old_tag != FMX_NOCHUNK && v != NONE) out[old_tag * 64u + lane] = (uint64_t)v;
        ring[ns * 64u + lane] = NONE;
        if (lane == 0) ring_tag[ns] = hq.c1;
        rseq++; rs0 = rs1; rs1 = ns;
      }
    }
    uint32_t sa = 0;
    if (fin_si != NONE) {                             // sample.rs:46-60 Some(sa)
      FMX_CHECK((uint64_t)fin_si <= (((uint64_t)n - 1) >> sa_level));
      if (owner) FMX_TOUCH(&samples[fin_si]);
      sa = samples[fin_si];
    }
    // one record per walking walk: the row goes from the lane that keeps it to the group
    const bool walking = active && (row & lmask) != 0u;
    const uint32_t rowx = walking ? row : NONE;
    const unsigned long long wm = __ballot(walking);
    uint4 p[Q];
    uint32_t rq[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      rq[q] = NONE;
      if (!(wm & (SLOT0 << q))) continue;             // walk q idle in every group of the wave
      rq[q] = Q == 1 ? rowx : Q == 8 ? fmx_oct_bcast(rowx, q) : fmx_quad_bcast(rowx, q);
      if (rq[q] != NONE) {                            // group-uniform
        FMX_CHECK(rq[q] < n && (rq[q] >> 8) < n / 256u + 1u);
        const uint4 *addr = rec + ((size_t)(rq[q] >> 8) * 8u + g);
        FMX_TOUCH_G0(g, addr - g);
        p[q] = *addr;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (!(wm & (SLOT0 << q))) continue;
      if (rq[q] != NONE) {                            // None: i = lf_map(i); steps += 1   fm_index.rs:134-137
        const uint32_t off = rq[q] & 255u;
        const uint32_t sym = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p[q], off & 31u) : 0u);
        const uint32_t nr = fmx_group_sum(fmx_piece_rank<3>(p[q], off, sym, g));   // absolute counters
        if (slot == (uint32_t)q) {
          row = nr;
          steps++;
        }
      }
    }
    if (fin_si != NONE && owner) {
      // fm_index.rs:131-133: (sa + steps) % len, in 32 bits: sa < n, steps < n, so the sum is below 2n and
      // one subtraction of n (modulo 2^32) is exact whether or not the addition wrapped
      uint32_t v = sa + fin_steps;
      if (v < sa || v >= n) v -= n;
      FMX_CHECK(fin_x < bn && fin_slot < FMX_WC_SLOTS);
      if (WC && ring_tag[fin_slot] == (fin_x >> 6)) ring[fin_slot * 64u + (fin_x & 63u)] = v;   // its ticket is resident
      else out[fin_x] = (uint64_t)v;
    }
  }
  if (WC) {                                           // what is still in the ring
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) {
      const uint32_t tag = ring_tag[r];
      const uint32_t v = ring[r * 64u + lane];
      FMX_CHECK(tag == FMX_NOCHUNK || v == NONE || tag * 64u + lane < bn);
      if (tag != FMX_NOCHUNK && v != NONE) out[tag * 64u + lane] = (uint64_t)v;
    }
  }
  if (steps_out && owner && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// Text-order walk over WALK RECORDS (FmxDev::walk, fmx_internal.h; round 4).  The shape of fmx_locate_f3p_kernel --
// 8 lanes per record, Q walks per group with walk q's state in lane q of each quad, the block's hit queue, the hand-over
// at the top of the round, the write-combining ring -- but a record of the walk array tells, next to lf_map(row), the
// row's phase SA[row] mod 2^level, its rank among the phase-0 rows and, for a phase-1 row, the rank of the row AFTER it
// among the phase-0 rows.  So a walk needs no probe of its own: its FIRST record gives the number of LF steps (the phase
// of the start row), and the record of its LAST BUT ONE row (phase 1) gives the index of the sample of its last row,
// whose record is never read: max(phase, 1) records and one sample per hit -- 2.75 requests at level 2 where the
// row-order walk issues 4 and the round-3 text-order walk 4.5 -- and no walk is longer than 2^level - 1 steps: the
// geometric tail of row-order sampling (a 2^20-hit batch ends on a chain of ~41 dependent round trips) does not exist.
// get_sa is unchanged as a function: (sample + steps) % len (fm_index.rs:127-140).
// The walk over the block's slice [blo, blo + bn) of the hits, called by all threads of a 1024-thread block.  `rows`: the
// rows of the slice (global memory, or LDS in the unified kernel); TL: only the tickets listed in tlist[0 .. nlist).
template <int Q, bool WC, bool TL>
__device__ __forceinline__ void fmx_f3t_walk(
    const uint4 *__restrict__ walk, const uint32_t *__restrict__ samples, uint32_t n, uint32_t nsamples,
    const uint32_t *rows, uint64_t blo, uint32_t bn, uint32_t chunk, const uint16_t *tlist, uint32_t nlist,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  static_assert(Q == 1 || Q == 2 || Q == 4 || Q == 8, "walks per group");
  __shared__ unsigned int lds_q;
  __shared__ uint32_t wc_ring[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS * 64 : 1];
  __shared__ uint32_t wc_tag[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS : 1];
  if (threadIdx.x == 0) lds_q = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane & (FMX_GROUP - 1);
  const uint32_t grp = lane >> 3;
  const uint32_t slot = g & (uint32_t)(Q - 1);        // the walk whose state this lane keeps (8 / Q replicas)
  const bool owner = g < (uint32_t)Q;                 // the replica that counts, stores and is counted
  const uint32_t olane = (lane & ~7u) | slot;         // its lane
  constexpr unsigned long long SLOT0 = Q == 8   ? 0x0101010101010101ull
                                       : Q == 4 ? 0x1111111111111111ull
                                       : Q == 2 ? 0x5555555555555555ull
                                                : 0xFFFFFFFFFFFFFFFFull;
  constexpr uint32_t NONE = 0xFFFFFFFFu;              // no row / no sample / no position (n < 2^32 - 16)
  FmxHitQueue<false, TL> hq;
  hq.tlist = tlist;
  hq.nlist = nlist;
  hq.init(rows, blo, bn, chunk, lane, lds_q);
  uint64_t *const out = out_pos + blo;                // the block's slice of the output (wave-uniform)
  [[maybe_unused]] volatile fmx_lds_u32 *const ring = FMX_LDS_U32(wc_ring + (threadIdx.x >> 6) * (FMX_WC_SLOTS * 64));
  [[maybe_unused]] volatile fmx_lds_u32 *const ring_tag = FMX_LDS_U32(wc_tag + (threadIdx.x >> 6) * FMX_WC_SLOTS);
  [[maybe_unused]] uint32_t rs0 = 0, rs1 = 1, rseq = 2;   // ring slots of the resident tickets c0 / c1; tickets drawn
  if (WC) {
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) ring[r * 64u + lane] = NONE;
    if (lane < FMX_WC_SLOTS) ring_tag[lane] = lane == 0 ? hq.c0 : (lane == 1 ? hq.c1 : FMX_NOCHUNK);
  }
  uint32_t hx, row;                                   // hit (index into the slice) and current row of the walk
  [[maybe_unused]] uint32_t myslot = 0;               // WC: ring slot of the ticket the hit came from
  bool active = hq.take32((slot << 3) | grp, hx, row);   // first 8 hits -> walk 0 of the 8 groups, ... (all of c0)
  {
    const bool slid = hq.advance(8u * (uint32_t)Q, lds_q);
    if (WC && slid) {                                 // Q = 8 hands out a whole ticket at once
      const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
      if (lane == 0) ring_tag[ns] = hq.c1;            // slot ns is still empty (all NONE)
      rseq++; rs0 = rs1; rs1 = ns;
    }
  }
  if (!active) row = 0u;
  // ctl = phase of the current row (= LF steps still to do) | the steps of the whole walk (= the phase of the start
  // row) << 8; FRESH until the walk's first record has told its phase.  fin: index of the walk's sample once known
  constexpr uint32_t FRESH = 0xFFu;
  uint32_t ctl = FRESH, fin = NONE, nsteps = 0;
  for (;;) {
    if (!__any(active)) break;
    // walks that learnt their sample index in the previous round: the slot goes to the next hit, the walk is completed
    // in this round (its sample travels with the new walk's first record)
    const bool done = active && fin != NONE;
    const unsigned long long fm = __ballot(done && owner);     // one bit per finishing walk
    uint32_t fin_si = NONE, fin_steps = 0, fin_x = 0;
    [[maybe_unused]] uint32_t fin_slot = 0;
    if (fm) {                                         // wave-uniform
      uint32_t x_new, r_new;
      bool first;
      const bool ok = hq.take32((uint32_t)__popcll(fm & ((1ull << olane) - 1ull)), x_new, r_new, first);
      if (done) {
        fin_si = fin;
        fin_steps = ctl >> 8;
        fin_x = hx;
        fin_slot = myslot;
        nsteps += ctl >> 8;
        hx = x_new;
        myslot = first ? rs0 : rs1;
        active = ok;
        row = ok ? r_new : 0u;
        ctl = FRESH;
        fin = NONE;
      }
      const bool slid = hq.advance((uint32_t)__popcll(fm), lds_q);
      if (WC && slid) {
        const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
        const uint32_t old_tag = ring_tag[ns];
        const uint32_t v = ring[ns * 64u + lane];
        FMX_CHECK(old_tag == FMX_NOCHUNK || v == NONE || old_tag * 64u + lane < bn);
        if (old_tag != FMX_NOCHUNK && v != NONE) out[old_tag * 64u + lane] = (uint64_t)v;
        ring[ns * 64u + lane] = NONE;
        if (lane == 0) ring_tag[ns] = hq.c1;
        rseq++; rs0 = rs1; rs1 = ns;
      }
    }
    uint32_t sa = 0;
    if (fin_si != NONE) {                             // sample.rs:46-60 Some(sa)
      FMX_CHECK(fin_si < nsamples);
      if (owner) FMX_TOUCH(&samples[fin_si]);
      sa = samples[fin_si];
    }
    // one record per live walk (a live walk has no sample index yet): its first record, or an LF step
    const uint32_t rowx = active ? row : NONE;
    const unsigned long long wm = __ballot(active);
    uint4 p[Q];
    uint32_t rq[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      rq[q] = NONE;
      if (!(wm & (SLOT0 << q))) continue;             // walk q idle in every group of the wave
      rq[q] = Q == 1 ? rowx : Q == 8 ? fmx_oct_bcast(rowx, q) : fmx_quad_bcast(rowx, q);
      if (rq[q] != NONE) {                            // group-uniform
        FMX_CHECK(rq[q] < n);
        const uint32_t wr = fmx_walk_record(rq[q], rq[q]);   // rq[q] becomes the row's index inside the record
        FMX_CHECK(wr < n / FMX_WALK_ROWS + 1u && rq[q] < FMX_WALK_ROWS);
        const uint4 *addr = walk + ((size_t)wr * 8u + g);
        FMX_TOUCH_G0(g, addr - g);
        p[q] = *addr;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; q++) {
      if (!(wm & (SLOT0 << q))) continue;
      if (rq[q] != NONE) {
        uint32_t ph, si;
        const uint32_t nr = fmx_walk_step(p[q], rq[q], g, ph, si);
        if (slot == (uint32_t)q) {
          if (ctl == FRESH) ctl = ph * 0x101u;        // the walk is exactly SA[row] mod 2^level steps long
          FMX_CHECK(ph == (ctl & 0xFFu));             // an LF step takes the phase down by one
          if (ph <= 1u) {
            // ph == 0 (only a start row): Some(sa), this row carries a sample (sample.rs:46-60).  ph == 1: the row
            // after this one does, and this record knows which -- the walk's last step (fm_index.rs:134-137) needs
            // no record of its own
            fin = si;
          } else {                                    // None: i = lf_map(i); steps += 1   fm_index.rs:134-137
            row = nr;
            ctl--;
          }
        }
      }
    }
    if (fin_si != NONE && owner) {
      // fm_index.rs:131-133: (sa + steps) % len; sa < n and steps < 8, one subtraction is exact
      uint32_t v = sa + fin_steps;
      if (v < sa || v >= n) v -= n;
      FMX_CHECK(fin_x < bn && fin_slot < FMX_WC_SLOTS);
      if (WC && ring_tag[fin_slot] == (fin_x >> 6)) ring[fin_slot * 64u + (fin_x & 63u)] = v;   // its ticket is resident
      else out[fin_x] = (uint64_t)v;
    }
  }
  if (WC) {                                           // what is still in the ring
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) {
      const uint32_t tag = ring_tag[r];
      const uint32_t v = ring[r * 64u + lane];
      FMX_CHECK(tag == FMX_NOCHUNK || v == NONE || tag * 64u + lane < bn);
      if (tag != FMX_NOCHUNK && v != NONE) out[tag * 64u + lane] = (uint64_t)v;
    }
  }
  if (steps_out && owner && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

template <int Q, bool WC>
__global__ __launch_bounds__(FMX_LOC_BLOCK) __attribute__((amdgpu_waves_per_eu(8))) void fmx_locate_f3t_kernel(
    const uint4 *__restrict__ walk, const uint32_t *__restrict__ samples, uint32_t n, uint32_t nsamples,
    uint64_t total, uint32_t hits_per_block, uint32_t chunk, const uint32_t *__restrict__ rows,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  fmx_f3t_walk<Q, WC, false>(walk, samples, n, nsamples, rows + blo, blo, bn, chunk, nullptr, 0u, out_pos, steps_out);
}

// locate walk, one walk per LANE (fmx_ep.h): 64 walks per wave.  RLFM: every LF step = lane-wise B
// probe -> access+rank rounds over the levels of S -> lane-wise B' / B selects; FM over several wavelet
// levels: one access+rank round per level (eight records in flight per lane).  Lanes take
// their hits from the block's queue (FmxHitQueue) as they finish.  K[] is staged in LDS when the
// alphabet is small (it is read with a data-dependent symbol in every step).
// TEXT (compile time since round 3): text-order sampling (FmxDev::phase).  The walk state of a lane is
// packed: row, hit (32-bit index into the block's slice), ring slot, and ONE control word -- the step
// count in row order; stage | steps left | phase in text order (a text-order walk is exactly `phase`
// steps long, so no separate counter) -- 4-5 registers where round 2 kept 8.
// WC: positions leave through the write-combining ring of the DNA walk kernel (fmx_locate_f3p_kernel):
// tickets are 64 consecutive hits = 512 contiguous bytes of out_pos, a finished walk drops its position
// into its ticket's ring slot while that ticket is resident, and a recycled slot is stored with one
// contiguous 64 x 8-byte store.  Round 2 stored every position on its own: an 8-byte store in
// completion order is one 32-byte sector and one request at the memory side (config 4: 22.5 MB written
// for 8.4 MB of positions).
// LFR (RLFM with the run table FmxDev::lfrun, round 4): an LF step is two lane-wise requests (fmx_rlfm_ep_lf_run) and the
// cooperative rank rounds are not run at all.
// The walk over the block's slice [blo, blo + bn) of the hits (rows = the rows of the slice), called by all threads of the block
template <int KIND, int NL, int SM, bool KLDS, bool TEXT, bool WC, bool LFR>
__device__ __forceinline__ void fmx_ep_walk(
    const FmxDev &ix, const uint32_t *__restrict__ rows, uint64_t blo, uint32_t bn,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  __shared__ uint32_t kt_lds[KLDS ? 1024 : 1];
  __shared__ unsigned int lds_q;
  __shared__ uint32_t hq_win[(FMX_LOC_BLOCK / 64) * 128];     // rows of each wave's two resident tickets
  __shared__ uint32_t wc_ring[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS * 64 : 1];
  __shared__ uint32_t wc_tag[WC ? (FMX_LOC_BLOCK / 64) * FMX_WC_SLOTS : 1];
  if (threadIdx.x == 0) lds_q = 0;
  if (KLDS) {
    for (uint32_t t = threadIdx.x; t <= ix.max_character; t += blockDim.x) kt_lds[t] = ix.K[t];
  }
  __syncthreads();
  const uint32_t *kt = KLDS ? kt_lds : ix.K;
  const uint32_t lane = threadIdx.x & 63u, g = lane & 7u, base = lane & ~7u;
  const uint32_t lmask = (1u << ix.sa_level) - 1u;
  constexpr uint32_t NONE = 0xFFFFFFFFu;              // no position (n < 2^32 - 16)
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave of the block (scalar)
  FmxHitQueue<true> hq;
  hq.init(rows, blo, bn, FMX_LCHUNK, lane, lds_q, FMX_LDS_U32(hq_win + wv * 128u));
  uint64_t *const out = out_pos + blo;                // the block's slice of the output (wave-uniform)
  // write-combining ring of this wave (see fmx_locate_f3p_kernel): slot r holds ticket ring_tag[r]
  [[maybe_unused]] volatile fmx_lds_u32 *const ring = FMX_LDS_U32(wc_ring + wv * (FMX_WC_SLOTS * 64));
  [[maybe_unused]] volatile fmx_lds_u32 *const ring_tag = FMX_LDS_U32(wc_tag + wv * FMX_WC_SLOTS);
  [[maybe_unused]] uint32_t rseq = 2;                 // tickets drawn so far (ticket k of the wave -> ring slot k & 3)
  static_assert(FMX_WC_SLOTS == 4, "the tag lookup below reads four tags");
  if (WC) {
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) ring[r * 64u + lane] = NONE;
    if (lane < FMX_WC_SLOTS) ring_tag[lane] = lane == 0 ? hq.c0 : (lane == 1 ? hq.c1 : FMX_NOCHUNK);
  }
  // the first 8 hits of a chunk go to lane 0 of the wave's 8 groups, the next 8 to lane 1, ...: a wave
  // that gets few hits (small batches) has few live endpoint positions and its rank rounds skip the rest
  uint32_t hx, row;                                   // hit (index into the slice) and current row of the walk
  bool active = hq.take32((g << 3) | (lane >> 3), hx, row);     // the whole first ticket (c0, ring slot 0)
  {
    const bool slid = hq.advance(64u, lds_q);
    if (WC && slid) {
      const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
      if (lane == 0) ring_tag[ns] = hq.c1;            // slot ns is still empty (all NONE)
      rseq++;
    }
  }
  if (!active) row = 0u;
  uint32_t nsteps = 0;
  // control word.  Row order: LF steps done so far.  Text order: bits 0-3 steps still to do, bits 4-7 the
  // walk's phase (= its length), bits 8-9 the stage -- 0 phase piece of the start row, 1 LF steps (no test
  // in between), 2 phase piece of the final row for its rank among the sampled rows, 3 the sample.
  // One round = the lane-wise probes of the lanes that need them, chained (phase piece -> sample ->
  // position -> next hit), THEN one LF step of every walking lane: a hit costs `steps` rounds (+ 1 in
  // text order) of its lane instead of one round per stage, and no LF step runs with idle probing lanes.
  uint32_t ctl = 0;
  [[maybe_unused]] uint32_t sidx = 0;                 // text order: index of the sample
  // text order: the phase piece of the lane's row -- of a start row (stage 0: phase -> walk length) or of the
  // final row (stage 2: its rank among the sampled rows) -- one lane-wise 16-byte probe
  auto probe = [&]() {
    const uint32_t st = ctl >> 8;
    const bool probing = active && (st == 0u || st == 2u);
    if (__any(probing)) {                             // wave-uniform
      if (probing) {
        uint32_t pt = 0;
        const uint32_t pi = fmx_phase_piece(row, ix.sa_level, pt);
        FMX_TOUCH(&ix.phase[pi]);
        const uint4 pc = ix.phase[pi];
        uint32_t rank0;
        const uint32_t phi = fmx_phase_decode(pc, pt, ix.sa_level, rank0);
        sidx = rank0;
        if (st == 0u) ctl = phi | (phi << 4) | ((phi == 0u ? 3u : 1u) << 8);   // start row: phase -> walk length
        else ctl = (ctl & 0xFFu) | (3u << 8);                                  // final row: its sample is next
      }
    }
  };
  while (__any(active)) {
    if (TEXT) probe();
    // walks standing on their sampled row: sample -> position; their lanes take the next hits
    const bool sampled = active && (TEXT ? (ctl >> 8) == 3u : (row & lmask) == 0u);
    const unsigned long long fmask = __ballot(sampled);
    if (fmask) {                                      // wave-uniform
      uint32_t sa = 0;
      if (sampled) {                                  // sample.rs:46-60 Some(sa)
        const uint32_t si = TEXT ? sidx : row >> ix.sa_level;
        FMX_CHECK(si < ix.nsamples);
        FMX_TOUCH(&ix.samples[si]);
        sa = ix.samples[si];
      }
      uint32_t x_new, r_new;
      const bool ok = hq.take32((uint32_t)__popcll(fmask & ((1ull << lane) - 1ull)), x_new, r_new);
      if (sampled) {
        // rlfmi.rs:181 / fm_index.rs:131-133: (sa + steps) % len, in 32 bits: sa < n, steps < n, so one
        // subtraction of n (modulo 2^32) is exact whether or not the addition wrapped
        const uint32_t stp = TEXT ? (ctl >> 4) & 15u : ctl;
        uint32_t v = sa + stp;
        if (v < sa || v >= ix.n) v -= ix.n;
        FMX_CHECK(hx < bn);
        // the ring slot that holds the hit's ticket, if it is still resident (the four tags in one LDS read;
        // a ticket id appears in at most one slot)
        uint32_t rslot = FMX_WC_SLOTS;
        if (WC) {
          const uint32_t t = hx >> 6;
          const uint32_t t0 = ring_tag[0], t1 = ring_tag[1], t2 = ring_tag[2], t3 = ring_tag[3];
          rslot = t == t0 ? 0u : t == t1 ? 1u : t == t2 ? 2u : t == t3 ? 3u : (uint32_t)FMX_WC_SLOTS;
        }
        if (WC && rslot < FMX_WC_SLOTS) ring[rslot * 64u + (hx & 63u)] = v;
        else out[hx] = (uint64_t)v;
        nsteps += stp;
        hx = x_new;
        active = ok;
        ctl = 0;
        row = ok ? r_new : 0u;
      }
      const bool slid = hq.advance((uint32_t)__popcll(fmask), lds_q);
      if (WC && slid) {
        // a new ticket was drawn: it gets the ring slot of the oldest one, whose arrived positions leave
        // now in one contiguous store (hit index = 64 * ticket + lane)
        const uint32_t ns = rseq & (FMX_WC_SLOTS - 1u);
        const uint32_t old_tag = ring_tag[ns];
        const uint32_t v = ring[ns * 64u + lane];
        FMX_CHECK(old_tag == FMX_NOCHUNK || v == NONE || old_tag * 64u + lane < bn);
        if (old_tag != FMX_NOCHUNK && v != NONE) out[old_tag * 64u + lane] = (uint64_t)v;
        ring[ns * 64u + lane] = NONE;
        if (lane == 0) ring_tag[ns] = hq.c1;
        rseq++;
      }
    }
    // text order: the hits just taken read their start row's phase piece at once, so that they walk in THIS round
    // like the hits of row order do (round 2 left it to the next round: 2.5 rounds of its lane per hit at level 2,
    // now 1.75, for one more lane-wise probe stage per round)
    if (TEXT && fmask) probe();
    // None: i = lf_map(i); steps += 1   rlfmi.rs:183-186 -- a hit taken above walks in this same round
    const bool walking = active && (TEXT ? (ctl >> 8) == 1u : (row & lmask) != 0u);
    if (__any(walking)) {
      uint32_t sym;
      uint32_t nrow;
      if constexpr (LFR) nrow = fmx_rlfm_ep_lf_run<(SM > 0 ? SM : 1)>(ix, walking ? row : 0u, walking, base, g);
      else nrow = KIND == FMX_KIND_RLFM
                      ? fmx_rlfm_ep_lf_map<NL, (SM > 0 ? SM : 1)>(ix, kt, walking ? row : 0u, walking, base, g, sym)
                      : fmx_fm_ep_lf_map<NL>(ix, kt, walking ? row : 0u, walking, base, g, sym);
      if (walking) {
        row = nrow;
        if (TEXT) {
          ctl -= 1u;                                  // one step less to do
          if ((ctl & 15u) == 0u) ctl = (ctl & 0xFFu) | (2u << 8);
        } else {
          ctl++;
        }
      }
    }
  }
  if (WC) {                                           // what is still in the ring
#pragma unroll
    for (uint32_t r = 0; r < FMX_WC_SLOTS; r++) {
      const uint32_t tag = ring_tag[r];
      const uint32_t v = ring[r * 64u + lane];
      FMX_CHECK(tag == FMX_NOCHUNK || v == NONE || tag * 64u + lane < bn);
      if (tag != FMX_NOCHUNK && v != NONE) out[tag * 64u + lane] = (uint64_t)v;
    }
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

template <int KIND, int NL, int SM, bool KLDS, bool TEXT, bool WC, bool LFR = false>
__global__ __launch_bounds__(FMX_LOC_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void fmx_locate_ep_kernel(
    FmxDev ix, uint64_t total, uint32_t hits_per_block, const uint32_t *__restrict__ rows,
    uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  fmx_ep_walk<KIND, NL, SM, KLDS, TEXT, WC, LFR>(ix, rows + blo, blo, bn, out_pos, steps_out);
}
// counts -> exclusive offsets (single block scan is enough off the hot path? no:
// npat can be 2^23, so do a 3-phase scan with one block per 2048-element tile)
#define FMX_SCAN_TILE 2048
__global__ __launch_bounds__(FMX_BLOCK) void fmx_tile_sums_kernel(const uint64_t *s,
                                                                   const uint64_t *e,
                                                                   uint64_t npat,
                                                                   uint64_t *tile_sum) {
  __shared__ uint64_t red[FMX_BLOCK];
  uint64_t base = (uint64_t)blockIdx.x * FMX_SCAN_TILE, acc = 0;
  for (uint32_t t = threadIdx.x; t < FMX_SCAN_TILE; t += FMX_BLOCK) {
    uint64_t k = base + t;
    if (k < npat) acc += e[k] > s[k] ? e[k] - s[k] : 0;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t w = FMX_BLOCK / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(FMX_BLOCK) void fmx_scan_tiles_kernel(uint64_t *tile_sum,
                                                                    uint64_t ntiles) {
  // one block: exclusive scan over the tile sums, a contiguous chunk per thread
  __shared__ uint64_t part[FMX_BLOCK];
  uint64_t per = (ntiles + FMX_BLOCK - 1) / FMX_BLOCK;
  uint64_t a = (uint64_t)threadIdx.x * per, b = a + per;
  if (a > ntiles) a = ntiles;
  if (b > ntiles) b = ntiles;
  uint64_t acc = 0;
  for (uint64_t t = a; t < b; t++) acc += tile_sum[t];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t d = 1; d < FMX_BLOCK; d <<= 1) {
    uint64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint64_t run = part[threadIdx.x] - acc;
  for (uint64_t t = a; t < b; t++) {
    uint64_t v = tile_sum[t];
    tile_sum[t] = run;
    run += v;
  }
  if (threadIdx.x == FMX_BLOCK - 1) tile_sum[ntiles] = part[FMX_BLOCK - 1];
}
__global__ __launch_bounds__(FMX_BLOCK) void fmx_tile_scan_kernel(const uint64_t *s,
                                                                   const uint64_t *e,
                                                                   uint64_t npat,
                                                                   const uint64_t *tile_sum,
                                                                   uint64_t ntiles,
                                                                   uint64_t *out_off) {
  __shared__ uint64_t part[FMX_BLOCK];
  constexpr int PER = FMX_SCAN_TILE / FMX_BLOCK;  // 8 consecutive elements per thread
  uint64_t base = (uint64_t)blockIdx.x * FMX_SCAN_TILE + (uint64_t)threadIdx.x * PER;
  uint64_t v[PER], acc = 0;
  for (int t = 0; t < PER; t++) {
    uint64_t k = base + t;
    v[t] = (k < npat && e[k] > s[k]) ? e[k] - s[k] : 0;
    acc += v[t];
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 partials
  for (uint32_t d = 1; d < FMX_BLOCK; d <<= 1) {
    uint64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint64_t run = tile_sum[blockIdx.x] + part[threadIdx.x] - acc;
  for (int t = 0; t < PER; t++) {
    uint64_t k = base + t;
    if (k < npat) out_off[k] = run;
    run += v[t];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out_off[npat] = tile_sum[ntiles];
}

// ---------------------------------------------------------------------------
// the four trait methods, batched (backend.rs:9-15, 29-31)
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_scalar_kernel(FmxDev ix, int op,
                                                                const uint64_t *__restrict__ cc,
                                                                const uint64_t *__restrict__ ii,
                                                                uint64_t k,
                                                                uint64_t *__restrict__ out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < k; q += ngroups) {
    uint64_t i64 = ii[q];
    uint64_t res = ~0ull;
    if (op == 2) {  // lf_map2(c, i), i in [0, n]
      uint64_t c = cc[q];
      if (c > ix.max_character || i64 > ix.n) {
        if (g == 0) atomicOr(ix.status, 1u << (c > ix.max_character ? FMX_ERR_SYMBOL_RANGE : FMX_ERR_ARG));
      } else {
        uint32_t a = (uint32_t)i64, b2 = (uint32_t)i64;
        fmx_lf_map2_pair<KIND>(ix, (uint32_t)c, a, b2, g);
        res = a;
      }
    } else if (i64 >= ix.n) {
      if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_ARG);
    } else if (op == 0 || op == 1) {  // get_l / lf_map
      uint32_t sym;
      uint32_t r = fmx_lf_map_any<KIND>(ix, (uint32_t)i64, g, sym);
      res = op == 0 ? (uint64_t)sym : (uint64_t)r;
    } else if (op == 6) {  // HasMultiPieces::piece_id (multi_pieces.rs:206-219)
      uint32_t row = (uint32_t)i64;
      for (;;) {
        uint32_t sym;
        uint32_t raw = fmx_mwm_lf(ix.bw, row, g, sym);
        if (sym == 0u) {  // doc[bw.rank(i, 0)] + 1 mod pieces
          FMX_CHECK(ix.K[0] + raw < ix.doc_count);
          uint32_t prev = ix.doc[ix.K[0] + raw];
          res = (uint64_t)((prev + 1u) % ix.doc_count);
          break;
        }
        row = ix.K[sym] + raw;
      }
    } else if (op == 4 || op == 5) {  // get_f / fl_map (fm_index.rs:97-120, rlfmi.rs:145-169)
      uint32_t sym;
      uint32_t r = fmx_fl_map_any<KIND>(ix, (uint32_t)i64, g, sym);
      res = op == 4 ? (uint64_t)sym : (r == 0xFFFFFFFFu && KIND == FMX_KIND_MULTI ? ~0ull : (uint64_t)r);
    } else if (ix.phase) {  // get_sa, text-order sampling
      uint32_t ns = 0;
      res = fmx_get_sa_text<KIND, 0, -1>(ix, (uint32_t)i64, g, ns);
    } else {  // get_sa (fm_index.rs:127-140)
      uint32_t row = (uint32_t)i64, steps = 0;
      const uint32_t lmask = (1u << ix.sa_level) - 1u;
      while ((row & lmask) != 0) {
        row = fmx_lf_step_any<KIND>(ix, row, g);
        steps++;
      }
      FMX_CHECK((row >> ix.sa_level) < ix.nsamples);
      uint64_t v = (uint64_t)ix.samples[row >> ix.sa_level] + steps;
      if (v >= ix.n) v -= ix.n;
      res = v;
    }
    if (g == 0) out[q] = res;
  }
}

// Match::iter_chars_backward / iter_chars_forward for many rows (wrapper.rs:154-183):
//   backward: c = get_l(i); i = lf_map(i); yield c          (never ends)
//   forward : c = get_f(i); i = fl_map(i)?; yield c         (ends, without yielding, at None)
// one 8-lane group per row, `len` dependent steps each; symbols land row-major in out[r*len + t]
template <int KIND, typename T>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_extract_kernel(FmxDev ix,
                                                                 const uint64_t *__restrict__ rows,
                                                                 uint64_t nrows, uint32_t len, int forward,
                                                                 T *__restrict__ out,
                                                                 uint64_t *__restrict__ out_len,
                                                                 uint64_t *__restrict__ out_next) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < nrows; q += ngroups) {
    const uint64_t i64 = rows[q];
    uint32_t t = 0;
    uint64_t next = ~0ull;
    if (i64 >= ix.n) {
      if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_ARG);
    } else {
      uint32_t i = (uint32_t)i64;
      T *dst = out + q * (uint64_t)len;
      bool ended = false;
      for (; t < len; t++) {
        uint32_t sym;
        const uint32_t nx = forward ? fmx_fl_map_any<KIND>(ix, i, g, sym) : fmx_lf_map_any<KIND>(ix, i, g, sym);
        if (forward && KIND == FMX_KIND_MULTI && nx == 0xFFFFFFFFu) { ended = true; break; }
        if (g == 0) dst[t] = (T)sym;
        i = nx;
      }
      if (!ended) next = i;
    }
    if (g == 0 && out_len) out_len[q] = t;
    if (g == 0 && out_next) out_next[q] = next;
  }
}

// export: L column of rows [0, n) as one byte per row (get_l, fm_index.rs:82-84)
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_export_l_kernel(FmxDev ix, void *__restrict__ out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < ix.n; q += ngroups) {
    uint32_t sym;
    (void)fmx_lf_map_any<KIND>(ix, (uint32_t)q, g, sym);
    if (g == 0) {
      if (ix.sym_bytes == 1) ((uint8_t *)out)[q] = (uint8_t)sym;
      else if (ix.sym_bytes == 2) ((uint16_t *)out)[q] = (uint16_t)sym;
      else ((uint32_t *)out)[q] = sym;
    }
  }
}
int fmx_launch_export_l(const fmx_index *idx, void *d_out, hipStream_t st) {
  if (idx->is_wide) return fmxw_launch_export_l(idx, d_out, st);
  const FmxDev dv = fmx_launch_dev(idx);
  if (idx->n == 0) return FMX_OK;
  if (idx->kind == FMX_KIND_FM || idx->kind == FMX_KIND_MULTI)
    hipLaunchKernelGGL(fmx_export_l_kernel<FMX_KIND_FM>, dim3(fmx_grid_for_groups(idx->n)),
                       dim3(FMX_BLOCK), 0, st, dv, d_out);
  else
    hipLaunchKernelGGL(fmx_export_l_kernel<FMX_KIND_RLFM>, dim3(fmx_grid_for_groups(idx->n)),
                       dim3(FMX_BLOCK), 0, st, dv, d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// ---- iter_matches() bookkeeping (wrapper.rs:203-217) incl. the match_prefix_only filter ------
__global__ __launch_bounds__(FMX_BLOCK) void fmx_match_counts_kernel(
    FmxDev ix, const uint64_t *__restrict__ s, const uint64_t *__restrict__ e, uint64_t npat,
    int prefix_only, uint64_t *__restrict__ out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    uint64_t a = s[k], b = e[k], cnt = b > a ? b - a : 0;
    if (prefix_only && cnt) {  // rows of [a, b) whose L symbol is the end marker
      uint32_t ra = fmx_mwm_rank(ix.bw, 0u, (uint32_t)a, g);
      uint32_t rb = fmx_mwm_rank(ix.bw, 0u, (uint32_t)b, g);
      cnt = rb - ra;
    }
    if (g == 0) out[k] = cnt;
  }
}
__global__ __launch_bounds__(FMX_BLOCK) void fmx_match_rows_kernel(
    FmxDev ix, const uint64_t *__restrict__ s, const uint64_t *__restrict__ e,
    const uint64_t *__restrict__ off, uint64_t npat, int prefix_only, uint64_t *__restrict__ rows) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    uint64_t a = s[k], b = e[k], o = off[k];
    uint64_t cnt = b > a ? b - a : 0;
    if (!prefix_only) {
      for (uint64_t t = g; t < cnt; t += FMX_GROUP) rows[o + t] = a + t;
    } else if (cnt) {
      uint32_t ra = fmx_mwm_rank(ix.bw, 0u, (uint32_t)a, g);
      uint32_t rb = fmx_mwm_rank(ix.bw, 0u, (uint32_t)b, g);
      for (uint32_t j = 0; j < rb - ra; j++) {   // the j-th end marker of L at or after row a
        uint32_t row = fmx_mwm_select(ix.bw, 0u, ix.K[0] + ra + j, g);
        if (g == 0) rows[o + j] = row;
      }
    }
  }
}
int fmx_launch_match_counts(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                            uint64_t npat, int prefix_only, uint64_t *d_cnt, hipStream_t st) {
  if (idx->is_wide) return fmxw_launch_match(idx, d_s, d_e, npat, prefix_only, nullptr, d_cnt, st);
  const FmxDev dv = fmx_launch_dev(idx);
  if (npat == 0) return FMX_OK;
  hipLaunchKernelGGL(fmx_match_counts_kernel, dim3(fmx_grid_for_groups(npat)), dim3(FMX_BLOCK), 0, st,
                     dv, d_s, d_e, npat, prefix_only, d_cnt);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
int fmx_launch_match_rows(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                          uint64_t npat, int prefix_only, const uint64_t *d_off, uint64_t *d_rows,
                          hipStream_t st) {
  if (idx->is_wide) return fmxw_launch_match(idx, d_s, d_e, npat, prefix_only, d_off, d_rows, st);
  const FmxDev dv = fmx_launch_dev(idx);
  if (npat == 0) return FMX_OK;
  hipLaunchKernelGGL(fmx_match_rows_kernel, dim3(fmx_grid_for_groups(npat)), dim3(FMX_BLOCK), 0, st,
                     dv, d_s, d_e, d_off, npat, prefix_only, d_rows);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// K[c] = cs[c] - S_c, S_c = the start chain of rank_range(0..i, c):
// it depends on the symbol only, so it is folded into the C array at build time.
__global__ __launch_bounds__(FMX_BLOCK) void fmx_compute_K_kernel(FmxMwm w, const uint64_t *cs,
                                                                   uint32_t *K,
                                                                   uint32_t max_character) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint32_t c = (blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  if (c > max_character) return;
  uint32_t sc = fmx_mwm_rank(w, c, 0u, g);
  if (g == 0) K[c] = (uint32_t)cs[c] - sc;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

static void fmx_time_begin(const fmx_index *idx, hipStream_t st) {
  fmx_index *m = const_cast<fmx_index *>(idx);
  if (idx->timing == 1) {
    (void)hipMemsetAsync(m->d_steps, 0, sizeof(uint64_t), st);
    (void)hipEventRecord(m->ev0, st);
  } else if (idx->timing == 2 && m->ev_series && m->series_n < FMX_SERIES_CAP) {
    (void)hipEventRecord(m->ev_series[2 * m->series_n], st);      // a series: launches back to back, events only
  }
}
static void fmx_time_end(const fmx_index *idx, hipStream_t st) {
  fmx_index *m = const_cast<fmx_index *>(idx);
  if (idx->timing == 1) {
    (void)hipEventRecord(m->ev1, st);
    m->ev_valid = 1;
  } else if (idx->timing == 2 && m->ev_series && m->series_n < FMX_SERIES_CAP) {
    (void)hipEventRecord(m->ev_series[2 * m->series_n + 1], st);
    m->series_n++;
  }
}

// ---------------------------------------------------------------------------
// Dispatch choices.  The SHIPPED library (libfmx.so) has one path per index kind and reads no environment
// variable on any launch path: fmx_tune() is a constant there and every test on it folds away.  The
// measurement builds (-DFMX_MEASURE: make measure / make debug) fill the same struct from FMX_VARIANT and
// the grid knobs and add the alternative kernels DESIGN.md section 4.1 quotes -- all of that lives in
// fmx_measure.inc, which only those builds include.
// ---------------------------------------------------------------------------
struct FmxTune {
  bool generic = false;          // round-1 group-per-pattern / group-per-walk kernels for every kind
  bool use_kmer = true;          // honour the k-mer start table when the index has one
  bool use_pair = true;          // honour the pair index when the index has one
  bool wc = true;                // locate: positions through the write-combining ring
  bool walk_records = true;      // DNA locate: the walk-record kernel when the index has walk records
  int alt = 0;                   // measurement-only kernel (fmx_measure.inc), 0 = none
  int walks = 0;                 // DNA locate: walks per group, 0 = by batch size
  long ep_blocks = 1024;         // count, endpoint per lane: grid cap (4 waves per SIMD saturate the memory system)
  long loc_blocks = 0;           // DNA locate: blocks, 0 = by batch size
  long loc_threads = FMX_LOC_BLOCK;
  long ep_loc_blocks = 0;        // one-walk-per-lane locate: blocks, 0 = by batch size
  long ep_loc_threads = 0;       // ... threads per block, 0 = by batch size
  bool unified = true;           // DNA locate on walk records: the one-launch kernel (fmx_locate_f3u_kernel); false = the
                                 // round-4 pair expand + f3t / lane kernel chosen from the batch average
  long adj_clusters = 4;         // ... a ticket of at most this many runs of consecutive rows is walked a lane per hit
  long rl_ep_min = 1l << 18;     // RLFM: one walk per lane from this many hits
  bool rl_unified = false;       // measurement builds (FMX_RL_UNIFIED=1): the RLFM lane kernel in one launch (slices expanded in LDS)
  long rl_lane_avg = 2;          // ... with the run table: a lane per walk on consecutive hits from this many hits per pattern
  bool rl_rounds = true;         // ... in rounds, four tickets per wave at a time (fmx_locate_rl_rounds_kernel); false = round 5's kernel
  long rl_rounds_blocks = 32768; // ... grid cap (config 4b: 2048 blocks 9.2 ms, 4096 7.8, 8192 7.5, 32768 7.3, 131072 7.35, 10^6 7.8)
  long fm_ep_min = 4l << 20;     // FM over several levels: one walk per lane from this many hits
};
// everything a count / locate launch needs, for the launch helpers below and in fmx_measure.inc
struct FmxCountCall {
  const fmx_index *idx; FmxDev dv; const void *pat; const uint64_t *off; uint64_t npat; const uint64_t *s0e0;
  uint64_t *s, *e, *cnt, *steps; hipStream_t st; bool km; unsigned grid;
};
struct FmxLocateCall {
  const fmx_index *idx; FmxDev dv; uint64_t total; const uint32_t *rows; uint64_t *pos, *steps; hipStream_t st;
  // 1024-thread blocks that each own a slice of the hits (FmxHitQueue): `nb` blocks wanted -> slice
  // length (a multiple of the ticket size, below 2^31) and the blocks that are really needed
  void slice(uint64_t nb, uint32_t chunk, uint32_t &hpb, unsigned &grid) const {
    const uint64_t min_nb = (total >> 31) + 1;
    if (nb < min_nb) nb = min_nb;
    uint64_t per = (total + nb - 1) / nb;
    per = (per + chunk - 1) / chunk * chunk;
    hpb = (uint32_t)per;
    grid = (unsigned)((total + per - 1) / per);
  }
};
// select structure of an RLFM index's B / B': 1 stored positions, 2 select blocks, 0 hints + records; -1 not RLFM
static inline int fmx_select_mode(const fmx_index *idx, const FmxDev &dv) {
  return idx->kind != FMX_KIND_RLFM ? -1 : (dv.b.pos && dv.bp.pos) ? 1 : (dv.b.dsel && dv.bp.dsel) ? 2 : 0;
}

// ---- count launch helpers (c = FmxCountCall) ----
#define FMX_F3_LAUNCH(c, PPG, SKIP, KM)                                                              \
  hipLaunchKernelGGL((fmx_count_f3_kernel<PPG, SKIP, KM>),                                             \
                     dim3(fmx_grid_capped(((c).npat + PPG - 1) / PPG, (c).grid)), dim3(FMX_BLOCK), 0, (c).st, \
                     (c).dv.bw.lv[0].rec, (c).dv.n, (c).dv.max_character, (c).dv.status, (c).dv.kmer,  \
                     (c).dv.kmer_k, (c).dv.kmer_bits, (const uint8_t *)(c).pat, (c).off, (c).npat,     \
                     (c).s0e0, (c).s, (c).e, (c).cnt, (c).steps)
#define FMX_PAIR_LAUNCH(c, KM)                                                                       \
  hipLaunchKernelGGL(fmx_count_pair_kernel<KM>, dim3((c).grid), dim3(FMX_BLOCK), 0, (c).st,            \
                     (c).dv.bw.lv[0].rec, (c).dv.pair_rec, (c).dv.n, (c).dv.max_character,             \
                     (c).dv.pair_row0, (c).dv.pair_row1, (c).dv.status, (c).dv.kmer, (c).dv.kmer_k,    \
                     (c).dv.kmer_bits, (const uint8_t *)(c).pat, (c).off, (c).npat, (c).s0e0, (c).s,   \
                     (c).e, (c).cnt, (c).steps)
// group per pattern; number of wavelet levels fixed at compile time for the common cases (1, 2)
#define FMX_COUNT_LAUNCH(c, KIND, NL, SM)                                                            \
  do {                                                                                               \
    if ((c).km && (c).idx->sym_bytes == 1)                                                           \
      hipLaunchKernelGGL((fmx_count_kernel<KIND, NL, true, SM>), dim3((c).grid), dim3(FMX_BLOCK), 0,   \
                         (c).st, (c).dv, (c).pat, (c).off, (c).npat, (c).s0e0, (c).s, (c).e, (c).cnt,  \
                         (c).steps);                                                                 \
    else                                                                                             \
      hipLaunchKernelGGL((fmx_count_kernel<KIND, NL, false, SM>), dim3((c).grid), dim3(FMX_BLOCK), 0,  \
                         (c).st, (c).dv, (c).pat, (c).off, (c).npat, (c).s0e0, (c).s, (c).e, (c).cnt,  \
                         (c).steps);                                                                 \
  } while (0)
#define FMX_COUNT_KIND(c, KIND, SM)                                                                  \
  do {                                                                                               \
    if ((c).dv.bw.nlevels == 1) FMX_COUNT_LAUNCH(c, KIND, 1, SM);                                    \
    else if ((c).dv.bw.nlevels == 2) FMX_COUNT_LAUNCH(c, KIND, 2, SM);                               \
    else FMX_COUNT_LAUNCH(c, KIND, 0, SM);                                                           \
  } while (0)
// endpoint per lane: one pattern per group while the grid lasts, `cap` blocks at most
#define FMX_EP_LAUNCH(c, eb, KIND, NL, SM)                                                           \
  do {                                                                                               \
    if ((c).km && (c).idx->sym_bytes == 1)                                                           \
      hipLaunchKernelGGL((fmx_count_ep_kernel<KIND, NL, SM, true>), dim3((unsigned)(eb)),              \
                         dim3(FMX_BLOCK), 0, (c).st, (c).dv, (c).pat, (c).off, (c).npat, (c).s0e0,     \
                         (c).s, (c).e, (c).cnt, (c).steps);                                          \
    else                                                                                             \
      hipLaunchKernelGGL((fmx_count_ep_kernel<KIND, NL, SM, false>), dim3((unsigned)(eb)),             \
                         dim3(FMX_BLOCK), 0, (c).st, (c).dv, (c).pat, (c).off, (c).npat, (c).s0e0,     \
                         (c).s, (c).e, (c).cnt, (c).steps);                                          \
  } while (0)
#define FMX_EP_SM(c, eb, KIND, SM)                                                                   \
  do {                                                                                               \
    if ((c).dv.bw.nlevels == 1) FMX_EP_LAUNCH(c, eb, KIND, 1, SM);                                   \
    else if ((c).dv.bw.nlevels == 2) FMX_EP_LAUNCH(c, eb, KIND, 2, SM);                              \
    else FMX_EP_LAUNCH(c, eb, KIND, 0, SM);                                                          \
  } while (0)
static inline uint64_t fmx_ep_count_blocks(uint64_t npat, long cap) {
  uint64_t eb = (npat + FMX_BLOCK / 8 - 1) / (FMX_BLOCK / 8);
  return eb > (uint64_t)cap ? (uint64_t)cap : eb;
}

// ---- DNA walk records, batches of LONG intervals: a lane per walk on consecutive hits (round 4) -----------------------
// The group-cooperative walk (fmx_locate_f3t_kernel) spends ~12 wave instructions per walk step -- every wave
// instruction serves the 8 walks of its 8 groups -- and config 3b (280 hits per pattern) is bound by exactly that
// (DESIGN.md section 8).  Here a LANE decodes its row's record alone: its own 16-byte piece (symbol, phase), the pieces in
// front of it (popcounts) and the piece that holds the counter it needs -- 5.5 lane-wise 16-byte loads per step on
// average instead of one cooperative line, which would be eight times the requests on random rows; but the hits of a
// pattern are adjacent rows of the same 112-row records, and LF keeps rows with the same symbol adjacent, so the 64
// lanes of a wave ask for a handful of lines per instruction and a wave instruction serves 64 walks.  Round 4 chose it
// from the batch average (64 hits per pattern or more, fmx_locate_walk_lane_kernel below: measurement builds only now);
// since round 5 fmx_locate_f3u_kernel makes the choice per 64-hit ticket and runs this walk in rounds.
// ONE record visit of a walk by ONE lane: the phase of `row` (SA[row] mod 2^level) and
//   phase <= 1: the index of the walk's sample -- of this row (phase 0) or of the row after it (phase 1);
//   else:       lf_map(row)                                                                     fm_index.rs:134-137
__device__ __forceinline__ uint32_t fmx_walk_lane_visit(const uint4 *__restrict__ walk, [[maybe_unused]] uint32_t n, uint32_t row,
                                                        uint32_t &ph) {
  FMX_CHECK(row < n);
  uint32_t off;
  const uint32_t wr = fmx_walk_record(row, off);
  const uint4 *R = walk + (size_t)wr * 8u;
  const uint32_t pi = off >> 4, bit = off & 15u;
  FMX_TOUCH(&R[pi]);
  // the row's piece and the pieces in front of it, all requested at once (round 5: a loop that fetched them one after
  // the other put up to seven dependent round trips into every record visit).  No predication: a lane that needs fewer
  // than six front pieces asks for its own piece again (the same line; its rows are masked out below) -- the kernel is
  // bound by vector-ALU issue, and a branch per piece cost more than the load it saved.
  uint4 front[6];
#pragma unroll
  for (uint32_t q = 0; q < 6u; q++) {
    const uint32_t qq = q < pi ? q : pi;
    if (q < pi) FMX_TOUCH(&R[q]);
    front[q] = R[qq];
  }
  const uint4 own = R[pi];
  const uint32_t sym = ((own.y >> bit) & 1u) | (((own.y >> (bit + 16u)) & 1u) << 1) | (((own.z >> bit) & 1u) << 2);
  ph = ((own.z >> (bit + 16u)) & 1u) | (((own.w >> bit) & 1u) << 1) | (((own.w >> (bit + 16u)) & 1u) << 2);
  const uint32_t m0 = 0u - (sym & 1u), m1 = 0u - ((sym >> 1) & 1u), m2 = 0u - (sym >> 2);      // all ones / zero
  const uint32_t isA = ph == 0u ? 0xFFFFFFFFu : 0u, notA = ~isA, notB = ph == 1u ? 0u : 0xFFFFFFFFu;
  // the rows of a piece that count towards the rank this visit needs (low 16 bits; the caller masks):
  //   phase 0: the phase-0 rows;  phase 1: the phase-1 rows with the row's symbol;  else: the rows with the row's symbol
  auto sel_of = [&](const uint4 &p) -> uint32_t {
    const uint32_t q0 = p.z >> 16, q2 = p.w >> 16;
    const uint32_t A = ~(q0 | p.w | q2);
    const uint32_t B = q0 & ~(p.w | q2);
    const uint32_t M = ~((p.y ^ m0) | ((p.y >> 16) ^ m1) | (p.z ^ m2));
    return (A & isA) | (M & notA & (B | notB));
  };
  uint32_t cnt = (uint32_t)__popc(sel_of(own) & ((1u << bit) - 1u));    // its own piece up to the row
#pragma unroll
  for (uint32_t q = 0; q < 6u; q++) cnt += (uint32_t)__popc(sel_of(front[q]) & (q < pi ? 0xFFFFu : 0u));   // whole pieces in front
  // the counter: phase 0 -> rank0 (piece 5); phase 1 -> rank1[sym] (piece 6 / 7); else lf_map2(sym, .) (piece sym - 1)
  const uint32_t cp = ph == 0u ? 5u : (ph == 1u ? (sym == 1u ? 6u : 7u) : sym - 1u);
  FMX_CHECK(ph == 0u || (sym >= 1u && sym <= FMX_WALK_MAX_CHARACTER));
  FMX_TOUCH(&R[cp]);
  const uint4 cv = R[cp];
  uint32_t ctr = cv.x;
  if (ph == 1u && sym >= 3u) ctr = sym == 3u ? cv.y : (sym == 4u ? cv.z : cv.w);
  return ctr + cnt;
}
// get_sa(row) by ONE lane over the walk records: returns the text position, adds the walk's LF steps to `nsteps`
__device__ __forceinline__ uint64_t fmx_walk_lane_get_sa(const uint4 *__restrict__ walk, const uint32_t *__restrict__ samples,
                                                         uint32_t n, [[maybe_unused]] uint32_t nsamples, uint32_t row,
                                                         uint64_t &nsteps) {
  uint32_t walk_steps = 0xFFFFFFFFu, si;
  for (;;) {
    uint32_t ph;
    const uint32_t v = fmx_walk_lane_visit(walk, n, row, ph);
    if (walk_steps == 0xFFFFFFFFu) walk_steps = ph;             // the walk is exactly SA[row] mod 2^level steps long
    if (ph <= 1u) { si = v; break; }                            // this row's sample (phase 0) or the next row's (phase 1)
    row = v;                                                    // None: i = lf_map(i); steps += 1
  }
  FMX_CHECK(si < nsamples);
  FMX_TOUCH(&samples[si]);
  uint64_t v = (uint64_t)samples[si] + walk_steps;              // (sa + steps) % len          fm_index.rs:131-133
  if (v >= n) v -= n;
  nsteps += walk_steps;
  return v;
}
__global__ __launch_bounds__(FMX_BLOCK) void fmx_locate_walk_lane_kernel(
    const uint4 *__restrict__ walk, const uint32_t *__restrict__ samples, uint32_t n, uint32_t nsamples, uint64_t total,
    const uint32_t *__restrict__ rows, uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  uint64_t nsteps = 0;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total; h += nth)
    out_pos[h] = fmx_walk_lane_get_sa(walk, samples, n, nsamples, rows[h], nsteps);
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}

// ---- DNA walk records: ONE launch per batch, the algorithm chosen per 64-hit TICKET (round 5) ---------------------------
// Until round 4 a locate batch was a stream-ordered allocation of the rows array, fmx_expand_kernel (rows[off[k] + j] =
// s[k] + j), and ONE of two walk kernels chosen from the batch AVERAGE of hits per pattern: the group-cooperative walk
// (fmx_locate_f3t_kernel: 8 lanes per record; right for scattered rows) or the lane-per-walk kernel (right for runs of
// adjacent rows, 1.5 x faster there and 8 x the requests elsewhere).  A batch of 10^7 singletons and a hundred
// patterns with 10^6 hits each averages 11 and sent its 10^8 adjacent hits down the wrong path.  Here
//   * a 1024-thread block owns a slice of <= 4096 hits and expands it ITSELF into LDS: one round of 1024 probes of
//     off[] brackets the slice's first pattern, then thread p takes pattern kc + p (its off, s, e: one round of loads)
//     and writes its rows -- ranges over 32 rows are queued and written by the whole block.  No rows array, no expand
//     launch, no 6 us between two kernels; the argument checks of fmx_expand_kernel are made here (FMX_ERR_ARG).
//   * every ticket (64 consecutive hits) is classified by the adjacency of its rows: at most `adj_clusters` runs of
//     consecutive rows -> the ticket is walked a lane per hit (phase A, tickets dealt to the waves in turn, positions
//     stored as one contiguous 512-byte line); all other tickets go through the block's hit queue to the
//     group-cooperative walk with its write-combining ring (phase B, fmx_f3t_walk).  The choice only picks the faster
//     of two exact algorithms: any classification gives the reference's positions in the reference's order.
#define FMX_U_SLICE 4096u
#define FMX_U_PATS 5120u           // patterns per expansion round (5 per thread): after one probe round over <= 2^20 patterns the
                                   // bracket is 1024 wide, and 1024 + 4096 patterns cover a slice of singletons in ONE round
#define FMX_U_LONG 32u
#define FMX_U_LONGCAP 128u
#define FMX_U_NOROW 0xFFFFFFFFu
// The rows of the block's slice [blo, blo + bn) of the hits into u_rows (LDS), by all 1024 threads of the block: rows[x] =
// s[k] + (blo + x - off[k]) for the pattern k that owns hit blo + x (wrapper.rs:203-217: i = s..e-1 ascending).  Slots no
// range covers (offsets with gaps) take row 0; returns true when this thread met an argument that is not of this index /
// these offsets (the caller reports FMX_ERR_ARG).  Ends with the block synchronised and u_rows complete.
struct FmxSliceLds {
  uint32_t rows[FMX_U_SLICE];
  uint32_t longs[FMX_U_LONGCAP * 3];                  // {first row, first slot, rows} of the long ranges of the slice
  unsigned long long klb;
  unsigned long long cross[3];                        // the range that crosses the slice's end: {first row, first hit, hits}
  unsigned int nlong;
};
// `k_hint` (in / out, block-uniform; ~0 = none): a lower bound of the slice's first pattern within one round of it -- a block
// that expands CONSECUTIVE slices passes the value the previous slice left (the first pattern of its last round) and
// skips the probe rounds.
__device__ __forceinline__ bool fmx_expand_slice(FmxSliceLds &L, const uint64_t *__restrict__ s, const uint64_t *__restrict__ e,
                                                 const uint64_t *__restrict__ off, uint64_t npat, uint64_t total, uint32_t n,
                                                 uint64_t blo, uint32_t bn, uint64_t *k_hint = nullptr) {
  const bool dense = npat >= (total >> 2);            // a pattern per four hits or more (block-uniform, the same in every block)
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint64_t bhi = blo + bn;
  bool bad = false;
  for (uint32_t x = tid; x < bn; x += FMX_LOC_BLOCK) L.rows[x] = FMX_U_NOROW;
  if (tid == 0) { L.nlong = 0; L.cross[2] = 0; }
  // ---- the slice's first pattern, bracketed: k_lo <= (largest k with off[k] <= blo) < k_lo + FMX_U_PATS ----
  // one round of 1024 probes cuts the bracket 1024-fold: none up to 5120 patterns, one up to 2^22, two up to 2^32.
  // (the predicate off[c] <= blo is monotone in c: a wave's best candidate is its highest lane that holds, and only
  // lane 0 of the wave touches the LDS word)
  uint64_t k_lo = 0, span = npat;
  if (k_hint && *k_hint != ~0ull) { k_lo = *k_hint; span = 0; }
  // (Round 6 measured a GUESSED start for batches of about a pattern per hit -- 512 patterns in front of blo * npat /
  // total, checked by the first round of loads, no probe round -- on the same box against this probe round: 0.0877 /
  // 0.0887 ms per 2^20 hits with the guess, 0.0845 / 0.0860 without; dropped.)
  while (span > FMX_U_PATS) {                         // block-uniform
    const uint64_t step = (span + FMX_LOC_BLOCK - 1) / FMX_LOC_BLOCK;
    const uint64_t c = k_lo + (uint64_t)tid * step;
    if (tid == 0) L.klb = k_lo;
    __syncthreads();
    const unsigned long long okm = __ballot(tid != 0 && c < k_lo + span && off[c] <= blo);
    if (okm && lane == 0)
      atomicMax(&L.klb, (unsigned long long)(k_lo + (uint64_t)((tid | 63u) - (uint32_t)__builtin_clzll(okm)) * step));
    __syncthreads();
    const uint64_t best = L.klb;
    span = best + step <= k_lo + span ? step : k_lo + span - best;
    k_lo = best;
  }
  __syncthreads();                                    // rows cleared, counter zeroed
  // ---- expansion: FMX_U_PATS patterns per round (thread p takes patterns kc + p, kc + 1024 + p, ...: their off, s, e
  // in one round of loads), until a round ends on a pattern that starts behind the slice ----
  // (sparse patterns -- fewer than one per four hits: a slice then holds a few patterns, often one -- take 1024 patterns
  // per round, the bracket's width, and load s and e only for the patterns whose offsets reach into the slice: a slice
  // inside one long interval otherwise loaded 5120 x 24 bytes for one pattern, 3 GB for a 10^8-hit batch)
  const uint32_t per_round = dense ? FMX_U_PATS : FMX_LOC_BLOCK;
  for (uint64_t kc = k_lo;; kc += per_round) {
    uint64_t a[FMX_U_PATS / FMX_LOC_BLOCK], b[FMX_U_PATS / FMX_LOC_BLOCK], o[FMX_U_PATS / FMX_LOC_BLOCK];
    bool need[FMX_U_PATS / FMX_LOC_BLOCK];
    if (dense) {                                      // about a pattern per hit: most of the round's patterns are the slice's
#pragma unroll
      for (uint32_t j = 0; j < FMX_U_PATS / FMX_LOC_BLOCK; j++) {
        const uint64_t k = kc + (uint64_t)j * FMX_LOC_BLOCK + tid;
        a[j] = 0; b[j] = 0; o[j] = 0;
        need[j] = k < npat;
        if (k < npat) { a[j] = s[k]; b[j] = e[k]; o[j] = off[k]; }
      }
    } else {
      const uint64_t k = kc + tid;
#pragma unroll
      for (uint32_t j = 0; j < FMX_U_PATS / FMX_LOC_BLOCK; j++) { a[j] = 0; b[j] = 0; o[j] = 0; need[j] = false; }
      uint64_t o1 = 0;
      if (k < npat) { o[0] = off[k]; o1 = k + 1 < npat ? off[k + 1] : total; }
      need[0] = k < npat && o[0] < bhi && o1 > blo;
      if (need[0]) { a[0] = s[k]; b[0] = e[k]; }
    }
    // the round's last pattern + 1: does it still start inside the slice?
    bool more = false;
    if (tid == FMX_LOC_BLOCK - 1u && kc + per_round < npat) more = off[kc + per_round] < bhi;
#pragma unroll
    for (uint32_t j = 0; j < FMX_U_PATS / FMX_LOC_BLOCK; j++) {
      if (!need[j]) continue;
      uint64_t aa = a[j], cnt = b[j] > aa ? b[j] - aa : 0;
      const uint64_t oo = o[j];
      // a range that is not one of this index, or offsets that do not leave room for it: refused (the slots take rows
      // from 0 on, so that the walk stays inside the index)
      if (b[j] > n || oo > total || cnt > total - oo) {
        bad = true;
        aa = 0;
        cnt = oo < total ? (cnt < total - oo ? cnt : total - oo) : 0;
        if (cnt > n) cnt = n;
      }
      if (cnt && oo < bhi && oo + cnt > blo) {        // the part of [oo, oo + cnt) inside the slice
        const uint64_t h0 = oo > blo ? oo : blo, h1 = oo + cnt < bhi ? oo + cnt : bhi;
        const uint32_t len = (uint32_t)(h1 - h0), x0 = (uint32_t)(h0 - blo), r0 = (uint32_t)(aa + (h0 - oo));
        uint32_t q = FMX_U_LONGCAP;
        if (len > FMX_U_LONG) q = atomicAdd(&L.nlong, 1u);
        if (q < FMX_U_LONGCAP) { L.longs[3u * q] = r0; L.longs[3u * q + 1u] = x0; L.longs[3u * q + 2u] = len; }
        else for (uint32_t t = 0; t < len; t++) L.rows[x0 + t] = r0 + t;
        if (oo + cnt > bhi) { L.cross[0] = aa; L.cross[1] = oo; L.cross[2] = cnt; }   // (one range at most, offsets being offsets)
      }
    }
    if (!__syncthreads_or((int)more)) {
      if (k_hint) *k_hint = kc;                       // the next slice's first pattern is one of this round's, or the one after
      break;
    }
  }
  {                                                   // the long ranges, by the whole block
    const uint32_t nl = L.nlong < FMX_U_LONGCAP ? L.nlong : FMX_U_LONGCAP;
    for (uint32_t j = 0; j < nl; j++) {
      const uint32_t r0 = L.longs[3u * j], x0 = L.longs[3u * j + 1u], len = L.longs[3u * j + 2u];
      for (uint32_t t = tid; t < len; t += FMX_LOC_BLOCK) L.rows[x0 + t] = r0 + t;
    }
  }
  __syncthreads();
  for (uint32_t x = tid; x < bn; x += FMX_LOC_BLOCK)   // gaps: row 0, reported
    if (L.rows[x] == FMX_U_NOROW) { L.rows[x] = 0u; bad = true; }
  __syncthreads();
  return bad;
}
// rows[] of a whole batch for the kernels that read them from global memory (every 32-bit path but the one-launch DNA
// kernel): a block per run of consecutive 4096-hit slices.  Replaces fmx_expand_kernel (round 1: a lane per pattern, ranges over 32 rows
// by the lane's wave) -- a thousand patterns of 10^5 hits each kept sixteen waves busy for 1.7 ms there (round 5,
// profiles/r05/locate_mix_*.jsonl: more than the walk of those 10^8 hits); here every slice is written by 1024 threads.
__global__ __launch_bounds__(FMX_LOC_BLOCK) void fmx_expand_slices_kernel(
    const uint64_t *__restrict__ s, const uint64_t *__restrict__ e, const uint64_t *__restrict__ off, uint64_t npat,
    uint64_t total, uint32_t n, uint32_t *__restrict__ rows, uint32_t *__restrict__ status) {
  __shared__ FmxSliceLds L;
  // a block takes `per` CONSECUTIVE slices: only its first one probes off[] for its first pattern
  const uint64_t nslices = (total + FMX_U_SLICE - 1) / FMX_U_SLICE, per = (nslices + gridDim.x - 1) / gridDim.x;
  const uint64_t s0 = (uint64_t)blockIdx.x * per, s1 = s0 + per < nslices ? s0 + per : nslices;
  uint64_t hint = ~0ull, c_row = 0, c_off = 0, c_cnt = 0;     // c_*: the range that crossed the previous slice's end
  for (uint64_t sl = s0; sl < s1; sl++) {
    const uint64_t blo = sl * FMX_U_SLICE;
    const uint32_t bn = (uint32_t)(total - blo < FMX_U_SLICE ? total - blo : FMX_U_SLICE);
    if (c_cnt && c_off <= blo && c_off + c_cnt >= blo + bn) {
      // the whole slice lies inside that range (a long interval): its rows follow from it, nothing is loaded
      for (uint32_t x = threadIdx.x; x < bn; x += FMX_LOC_BLOCK) rows[blo + x] = (uint32_t)(c_row + (blo + x - c_off));
      continue;                                       // (block-uniform; the hint stays: the next slice may start in the same range)
    }
    if (fmx_expand_slice(L, s, e, off, npat, total, n, blo, bn, &hint)) atomicOr(status, 1u << FMX_ERR_ARG);
    for (uint32_t x = threadIdx.x; x < bn; x += FMX_LOC_BLOCK) rows[blo + x] = L.rows[x];
    c_row = L.cross[0]; c_off = L.cross[1]; c_cnt = L.cross[2];
    __syncthreads();
  }
}

template <int Q, bool WC>
__global__ __launch_bounds__(FMX_LOC_BLOCK) __attribute__((amdgpu_waves_per_eu(8))) void fmx_locate_f3u_kernel(
    const uint4 *__restrict__ walk, const uint32_t *__restrict__ samples, uint32_t n, uint32_t nsamples,
    const uint64_t *__restrict__ s, const uint64_t *__restrict__ e, const uint64_t *__restrict__ off, uint64_t npat,
    uint64_t total, uint32_t hits_per_block, uint32_t chunk, uint32_t adj_clusters, uint64_t *__restrict__ out_pos,
    uint64_t *__restrict__ steps_out, uint32_t *__restrict__ status) {
  __shared__ FmxSliceLds L;
  __shared__ uint16_t u_tlist[FMX_U_SLICE / 8];       // tickets for the cooperative walk
  __shared__ uint16_t u_alist[FMX_U_SLICE / 8];       // tickets walked a lane per hit
  __shared__ uint16_t u_walks[2][FMX_U_SLICE];        // ... their unfinished walks, round by round
  __shared__ uint8_t u_wlen[FMX_U_SLICE];             // ... and the length of every finished one
  __shared__ unsigned int u_ntl, u_nal, u_ppos;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  FMX_CHECK(hits_per_block <= FMX_U_SLICE && chunk >= 8u && chunk <= FMX_LCHUNK);
  if (tid == 0) { u_ntl = 0; u_nal = 0; u_ppos = FMX_NOCHUNK; }
  if (fmx_expand_slice(L, s, e, off, npat, total, n, blo, bn)) atomicOr(status, 1u << FMX_ERR_ARG);
  uint32_t *const u_rows = L.rows;
  // ---- tickets: the adjacency of their rows ----
  const uint32_t ntick = (bn + chunk - 1u) / chunk;
  for (uint32_t t = wv; t < ntick; t += FMX_LOC_BLOCK / 64u) {
    const uint32_t x = t * chunk + lane;
    const bool in = lane < chunk && x < bn;
    const uint32_t r = in ? u_rows[x] : 0u;
    const bool adjacent = chunk == FMX_LCHUNK && fmx_ticket_breaks(r, in, lane) < adj_clusters;
    if (lane == 0) {
      if (adjacent) u_alist[atomicAdd(&u_nal, 1u)] = (uint16_t)t;
      else {
        const uint32_t p = atomicAdd(&u_ntl, 1u);
        u_tlist[p] = (uint16_t)t;
        if (t == ntick - 1u && bn % chunk != 0u) u_ppos = p;   // the slice's PARTIAL ticket (the last slice of the batch only)
      }
    }
  }
  __syncthreads();
  // The list fills in the order the waves' atomics land, but the hit queue takes an invalid hit index for "the slice has
  // run dry" (FmxHitQueue::take32): a partial ticket must be the LAST one drawn, as it was when tickets were drawn in
  // ascending order -- a wave that held it as c0 in front of a full c1 would retire its walk slots on the ticket's missing
  // tail and never walk c1.  One thread moves it to the end of the list; the barrier at the top of
  // fmx_f3t_walk orders the swap before any draw.
  if (tid == 0 && u_ppos != FMX_NOCHUNK) {
    const uint32_t p = u_ppos, last = u_ntl - 1u;
    const uint16_t a = u_tlist[p];
    u_tlist[p] = u_tlist[last];
    u_tlist[last] = a;
  }
  // ---- phase A: a lane per walk on the tickets of adjacent rows, in ROUNDS of one record visit per live walk ----
  // A wave that walks its 64 hits to the end runs as long as its longest walk -- 2^level - 1 visits -- with half of its
  // lanes done after the first visit and three quarters after the second (phases are uniform): config 3b spent 709 vector
  // instructions per hit that way and was bound by exactly that (VALU busy 0.93, profiles/r05/kernel_pmc_dna.json).  Here
  // the unfinished walks are kept in a list (LDS: slot | the walk's length << 12; the walk's row stays in
  // u_rows[slot]) that every round compacts: the visits executed are the visits needed, max(phase, 1) per hit.
  const uint32_t nal = u_nal, ntl = u_ntl;
  if (nal) {                                          // block-uniform
    // Round 6: every WAVE runs the rounds of its own tickets (u_alist[wv], u_alist[wv + 16], ...: at most four) on its own
    // quarter-kilobyte of the lists, with no block barrier and no atomic in between.  Rounds shared by the block
    // (round 5) put all sixteen waves into the same step at the same time -- all of them waiting for their records,
    // then all of them decoding -- and the block went at the pace of its slowest wave five times per slice.
    constexpr uint32_t kWaves = FMX_LOC_BLOCK / 64u, kOwn = FMX_U_SLICE / kWaves;     // 16 waves, 256 hits each
    uint16_t *const own0 = u_walks[0] + wv * kOwn, *const own1 = u_walks[1] + wv * kOwn;
    uint32_t nsteps = 0;
    uint32_t cnt = 0;                                 // unfinished walks of this wave (wave-uniform)
    for (uint32_t a = wv; a < nal; a += kWaves) {     // round 0: the tickets themselves
      const uint32_t x = (uint32_t)u_alist[a] * FMX_LCHUNK + lane;
      bool more = false;
      uint32_t wsteps = 0;
      if (x < bn) {
        uint32_t ph;
        const uint32_t v = fmx_walk_lane_visit(walk, n, u_rows[x], ph);
        wsteps = ph;                                  // the walk is exactly SA[row] mod 2^level steps long
        u_rows[x] = v;                                // lf_map(row) -- or, for a walk that ends here, its sample's index
        if (ph <= 1u) {                               // this row's sample (phase 0) or the next row's (phase 1)
          u_wlen[x] = (uint8_t)wsteps;
          nsteps += wsteps;
        } else {                                      // None: i = lf_map(i); steps += 1   fm_index.rs:134-137
          more = true;
        }
      }
      const unsigned long long mm = __ballot(more);   // the wave's unfinished walks, appended in hit order
      if (more) own0[cnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)(x | (wsteps << 12));
      cnt += (uint32_t)__popcll(mm);
    }
    for (uint32_t round = 1; cnt != 0u; round++) {
      const uint16_t *const cur = (round & 1u) ? own0 : own1;
      uint16_t *const nxt = (round & 1u) ? own1 : own0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");        // the list entries other lanes of this wave wrote
      uint32_t ncnt = 0;
      for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
        const uint32_t i = i0 + lane;
        bool more = false;
        uint32_t x = 0, wsteps = 0;
        if (i < cnt) {
          const uint32_t en = cur[i];
          x = en & 0xFFFu;
          wsteps = en >> 12;
          uint32_t ph;
          const uint32_t v = fmx_walk_lane_visit(walk, n, u_rows[x], ph);
          u_rows[x] = v;
          if (ph <= 1u) {
            u_wlen[x] = (uint8_t)wsteps;
            nsteps += wsteps;
          } else {
            more = true;
          }
        }
        const unsigned long long mm = __ballot(more);
        if (more) nxt[ncnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)(x | (wsteps << 12));
        ncnt += (uint32_t)__popcll(mm);
      }
      cnt = ncnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // every walk of these tickets has left the index of its sample in u_rows[slot]: the samples in one sweep (no load
    // waits for a visit any more; adjacent rows' samples are neighbours), the positions as contiguous 512-byte lines
    for (uint32_t a = wv; a < nal; a += kWaves) {
      const uint32_t x = (uint32_t)u_alist[a] * FMX_LCHUNK + lane;
      if (x < bn) {
        const uint32_t si = u_rows[x];
        FMX_CHECK(si < nsamples);
        FMX_TOUCH(&samples[si]);
        uint64_t pos = (uint64_t)samples[si] + u_wlen[x];        // (sa + steps) % len          fm_index.rs:131-133
        if (pos >= n) pos -= n;
        out_pos[blo + x] = pos;
      }
    }
    if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
  }
  // ---- phase B: the group-cooperative walk on the others ----
  if (ntl) fmx_f3t_walk<Q, WC, true>(walk, samples, n, nsamples, u_rows, blo, bn, chunk, u_tlist, ntl, out_pos, steps_out);
}

// ---- RLFM with the run table, batches of LONG intervals: a lane per walk on consecutive hits (round 4) ----------------
// The hits of a pattern are adjacent rows; adjacent rows sit in the same runs and stay neighbours under LF (lf_map(i) =
// lfrun[run] + offset in the run).  A wave that walks 64 CONSECUTIVE hits -- lane per walk, every probe lane-wise,
// no queue, no ring: hit h is read and written by lane h mod 64 -- therefore sends its 64 requests of a step into a
// few lines, where fmx_locate_ep_kernel's refilling lanes scatter them (the same finding as on the wide engine,
// fmxw_r_walk_text_kernel: 9.8e7 hits of the repetitive 1 GiB text in 2.5 ms against 3.45 ms).  Chosen when the
// batch averages two hits per pattern or more (round 5; 64 in round 4 -- see the launcher).
// select1(k) of B by ONE lane (k < ones): stored position / select block / hint + search over the records
__device__ __forceinline__ uint32_t fmx_bits_lane_select(const FmxBits &bv, uint32_t k) {
  if (bv.pos) { FMX_TOUCH(&bv.pos[k]); return bv.pos[k]; }
  if (bv.dsel) {
    FMX_TOUCH(&bv.dsel[k >> bv.dsel_shift]);
    const uint4 blk = bv.dsel[k >> bv.dsel_shift];
    if (blk.x != 0xFFFFFFFFu) return fmx_dsel_pos(blk, k, bv.dsel_shift);
  }
  const uint32_t h = k / FMX_SEL_STEP;
  FMX_CHECK(h + 1 < bv.nsel);
  uint32_t lo = bv.sel[h], hi = bv.sel[h + 1];
  while (lo < hi) {                                   // last record whose count <= k
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (bv.rec[(size_t)mid * 8u].x <= k) lo = mid; else hi = mid - 1u;
  }
  uint32_t p = 0;
  for (uint32_t q = 1; q < 8u; q++)                   // last piece of it whose count <= k (the record is one line)
    if (bv.rec[(size_t)lo * 8u + q].x <= k) p = q;
  FMX_TOUCH(&bv.rec[(size_t)lo * 8u + p]);
  const uint4 pc = bv.rec[(size_t)lo * 8u + p];
  const uint32_t rem = k - pc.x, c0 = __popc(pc.y), c1 = __popc(pc.z);
  uint32_t pos;
  if (rem < c0) pos = fmx_select32(pc.y, rem);
  else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
  else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
  return lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
}
// lf_map(row) through the run table by ONE lane                                              rlfmi.rs:127-133
__device__ __forceinline__ uint32_t fmx_rlfm_lane_lf(const FmxDev &ix, uint32_t row) {
  const uint32_t pidx = fmx_div3(row >> 5);           // row / 96
  const uint32_t b1 = row - pidx * FMX_BITS_PER_PIECE + 1u;   // bits [0, bit] of the piece
  FMX_CHECK(pidx < ix.b.nrec * 8u);
  FMX_TOUCH(&ix.b.rec[pidx]);
  const uint4 pc = ix.b.rec[pidx];
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  const uint32_t y = pc.y & m0, z = pc.z & m1, w = pc.w & m2;
  const uint32_t lo = pc.x + __popc(y) + __popc(z) + __popc(w) - 1u;      // the run of the row
  FMX_CHECK(lo < ix.b.ones);
  FMX_TOUCH(&ix.lfrun[lo]);
  const uint32_t f = ix.lfrun[lo];                    // lf_map(first row of the run)
  uint32_t st;                                        // its first row: the last one at or before the row
  if (w) st = pidx * FMX_BITS_PER_PIECE + 95u - (uint32_t)__builtin_clz(w);
  else if (z) st = pidx * FMX_BITS_PER_PIECE + 63u - (uint32_t)__builtin_clz(z);
  else if (y) st = pidx * FMX_BITS_PER_PIECE + 31u - (uint32_t)__builtin_clz(y);
  else st = fmx_bits_lane_select(ix.b, lo);
  return f + row - st;
}
// get_sa(row) by ONE lane through the run table (rlfmi.rs:172-190): returns the text position, adds the LF steps
template <bool TEXT>
__device__ __forceinline__ uint64_t fmx_rlfm_lane_get_sa(const FmxDev &ix, uint32_t row, uint64_t &nsteps) {
  uint32_t steps = 0, si;
  FMX_CHECK(row < ix.n);                              // (the expand kernels write every slot with a row of this index)
  if (TEXT) {                                         // SA[row] mod 2^level steps; phase probes at both ends
    uint32_t t;
    uint32_t pi = fmx_phase_piece(row, ix.sa_level, t);
    FMX_TOUCH(&ix.phase[pi]);
    steps = fmx_phase_decode(ix.phase[pi], t, ix.sa_level, si);
    for (uint32_t k = 0; k < steps; k++) row = fmx_rlfm_lane_lf(ix, row);
    if (steps) {
      pi = fmx_phase_piece(row, ix.sa_level, t);
      FMX_TOUCH(&ix.phase[pi]);
      [[maybe_unused]] const uint32_t p2 = fmx_phase_decode(ix.phase[pi], t, ix.sa_level, si);
      FMX_CHECK(p2 == 0u);
    }
  } else {                                            // the reference's rows (sample.rs:46-60)
    const uint32_t lmask = (1u << ix.sa_level) - 1u;
    while (row & lmask) { row = fmx_rlfm_lane_lf(ix, row); steps++; }
    si = row >> ix.sa_level;
  }
  FMX_CHECK(si < ix.nsamples);
  FMX_TOUCH(&ix.samples[si]);
  uint64_t v = (uint64_t)ix.samples[si] + steps;      // (sa + steps) % len                      rlfmi.rs:178-182
  if (v >= ix.n) v -= ix.n;
  nsteps += steps;
  return v;
}
// (Round 5 also tried TWO walks per lane with their loads paired -- the kernel spends 82 % of its wave cycles parked on
// s_waitcnt at full occupancy, profiles/r05/kernel_pmc_rep_rlfm_v1.json -- 19 -> 42 VGPRs, no gain: 1.19 -> 1.26 ms on the
// mixed batch of profiles/r05/locate_mix_*.jsonl; the requests in flight per lane are not what it lacks.)
template <bool TEXT>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_locate_rl_lane_kernel(FmxDev ix, uint64_t total,
                                                                        const uint32_t *__restrict__ rows,
                                                                        uint64_t *__restrict__ out_pos,
                                                                        uint64_t *__restrict__ steps_out) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  uint64_t nsteps = 0;
  for (uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < total; h += nth)
    out_pos[h] = fmx_rlfm_lane_get_sa<TEXT>(ix, rows[h], nsteps);
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}
// ---- the same walks in ROUNDS, four tickets per wave at a time (round 6) ---------------------------------------------------
// fmx_locate_rl_lane_kernel above is bound by the chain of dependent round trips of a wave, at full occupancy (82 % of
// its wave cycles parked on s_waitcnt, 22 G requests/s of the 55 the chip serves: profiles/r06/kernel_pmc_rep_rlfm_2p27.json):
// a wave takes 64 hits through rows -> phase probe -> (B piece -> table entry) x its LONGEST walk -> phase probe -> sample,
// ten trips one after the other with half of its lanes idle in the step loop, and 8.0 ms / (1.2e7 tickets / 8192 resident
// waves) / 10 trips = 0.53 us per trip says that is all the time there is.  Here a wave owns 256 consecutive hits: their
// rows, probes, LF steps and samples are requested FOUR tickets at a time (independent loads issued back to back: one
// trip serves 256 hits), and the unfinished walks are kept as a packed list in the wave's own LDS (slot | steps so far),
// compacted every round with a ballot -- no block barrier, no atomic -- so that a round issues loads for the walks that
// still run and for nothing else.  Same results as the kernel above for every hit (any order of the LF steps of
// different walks is the reference's get_sa per hit: rlfmi.rs:172-190).
#define FMX_RLR_BLOCK 256
#define FMX_RLR_HITS 256u            // per wave and pass
struct FmxRlrLds {
  uint32_t row[FMX_RLR_HITS];        // the walk's current row; once it has ended, the index of its sample
  uint32_t list[2][FMX_RLR_HITS];    // unfinished walks: slot | LF steps taken << 8
  uint32_t len[FMX_RLR_HITS];        // LF steps of the finished walk
};
// B piece of `row` requested (stage A of an LF step)
__device__ __forceinline__ uint4 fmx_rlr_piece(const FmxDev &ix, uint32_t row) {
  const uint32_t pidx = fmx_div3(row >> 5);           // row / 96
  FMX_CHECK(pidx < ix.b.nrec * 8u);
  FMX_TOUCH(&ix.b.rec[pidx]);
  return ix.b.rec[pidx];
}
// stage B: the run of the row, its table entry requested, and -- when the run starts in front of the piece -- the one
// load that answers select1 (stored position or select block); st = the run's first row when the piece holds it
struct FmxRlrStep { uint32_t lo, f, st, aux; uint4 blk; int mode; };   // mode 0: st known; 1: aux = position; 2: blk; 3: search
__device__ __forceinline__ FmxRlrStep fmx_rlr_entry(const FmxDev &ix, uint32_t row, const uint4 pc) {
  FmxRlrStep r;
  const uint32_t pidx = fmx_div3(row >> 5);
  const uint32_t b1 = row - pidx * FMX_BITS_PER_PIECE + 1u;   // bits [0, bit] of the piece
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  const uint32_t y = pc.y & m0, z = pc.z & m1, w = pc.w & m2;
  r.lo = pc.x + __popc(y) + __popc(z) + __popc(w) - 1u;       // the run of the row
  FMX_CHECK(r.lo < ix.b.ones);
  FMX_TOUCH(&ix.lfrun[r.lo]);
  r.f = ix.lfrun[r.lo];                                       // lf_map(first row of the run)
  r.mode = 0; r.st = 0; r.aux = 0; r.blk = make_uint4(0, 0, 0, 0);
  if (w) r.st = pidx * FMX_BITS_PER_PIECE + 95u - (uint32_t)__builtin_clz(w);
  else if (z) r.st = pidx * FMX_BITS_PER_PIECE + 63u - (uint32_t)__builtin_clz(z);
  else if (y) r.st = pidx * FMX_BITS_PER_PIECE + 31u - (uint32_t)__builtin_clz(y);
  else if (ix.b.pos) { r.mode = 1; FMX_TOUCH(&ix.b.pos[r.lo]); r.aux = ix.b.pos[r.lo]; }
  else if (ix.b.dsel) { r.mode = 2; FMX_TOUCH(&ix.b.dsel[r.lo >> ix.b.dsel_shift]); r.blk = ix.b.dsel[r.lo >> ix.b.dsel_shift]; }
  else r.mode = 3;
  return r;
}
// stage C: lf_map(row) = table entry + offset of the row in its run                                   rlfmi.rs:127-133
__device__ __forceinline__ uint32_t fmx_rlr_finish(const FmxDev &ix, uint32_t row, const FmxRlrStep &r) {
  uint32_t st = r.st;
  if (r.mode == 1) st = r.aux;
  else if (r.mode == 2 && r.blk.x != 0xFFFFFFFFu) st = fmx_dsel_pos(r.blk, r.lo, ix.b.dsel_shift);
  else if (r.mode >= 2) st = fmx_bits_lane_select(ix.b, r.lo);          // hints + record search (rare)
  return r.f + row - st;
}
template <bool TEXT>
__global__ __launch_bounds__(FMX_RLR_BLOCK) void fmx_locate_rl_rounds_kernel(FmxDev ix, uint64_t total,
                                                                            const uint32_t *__restrict__ rows,
                                                                            uint64_t *__restrict__ out_pos,
                                                                            uint64_t *__restrict__ steps_out) {
  __shared__ FmxRlrLds lds[FMX_RLR_BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63u;
  FmxRlrLds &W = lds[threadIdx.x >> 6];
  const uint64_t nwaves = (uint64_t)gridDim.x * (FMX_RLR_BLOCK / 64);
  const uint64_t wave = (uint64_t)blockIdx.x * (FMX_RLR_BLOCK / 64) + (threadIdx.x >> 6);
  const uint32_t lmask = (1u << ix.sa_level) - 1u;
  uint64_t nsteps = 0;
  for (uint64_t h0 = wave * FMX_RLR_HITS; h0 < total; h0 += nwaves * FMX_RLR_HITS) {       // wave-uniform
    const uint32_t m = (uint32_t)(total - h0 < FMX_RLR_HITS ? total - h0 : FMX_RLR_HITS);
    uint32_t cnt = 0;                                 // unfinished walks (wave-uniform)
    // ---- the hits' rows, and where their walks stand at the start ----
    {
      uint32_t row[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t x = j * 64u + lane;
        row[j] = x < m ? rows[h0 + x] : 0u;
        FMX_CHECK(row[j] < ix.n);                     // (the expand kernels write every slot with a row of this index)
      }
      uint32_t left[4], si[4];
      if (TEXT) {                                     // SA[row] mod 2^level steps to go; phase 0: the row's own sample
        uint4 pp[4]; uint32_t t[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          const uint32_t pi = fmx_phase_piece(row[j], ix.sa_level, t[j]);
          FMX_TOUCH(&ix.phase[pi]);
          pp[j] = ix.phase[pi];
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) left[j] = fmx_phase_decode(pp[j], t[j], ix.sa_level, si[j]);
      } else {                                        // the reference's rows (sample.rs:46-60): a sampled row ends the walk
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { left[j] = row[j] & lmask; si[j] = row[j] >> ix.sa_level; }
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t x = j * 64u + lane;
        const bool more = x < m && left[j] != 0u;
        if (x < m) { W.row[x] = more ? row[j] : si[j]; W.len[x] = TEXT ? left[j] : 0u; }    // TEXT: the walk's length is known
        const unsigned long long mm = __ballot(more);
        if (more) W.list[0][cnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = x;
        cnt += (uint32_t)__popcll(mm);
      }
    }
    // ---- rounds of one LF step per unfinished walk; a walk that ends leaves its sample's index (or, TEXT, its last row) ----
    for (uint32_t round = 0; cnt != 0u; round++) {
      const uint32_t *const cur = W.list[round & 1u];
      uint32_t *const nxt = W.list[(round & 1u) ^ 1u];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // what other lanes of this wave wrote to the lists
      uint32_t ncnt = 0;
      for (uint32_t i0 = 0; i0 < cnt; i0 += 256u) {              // (once: cnt <= 256)
        uint32_t en[4], row[4];
        uint4 pc[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          const uint32_t i = i0 + j * 64u + lane;
          en[j] = i < cnt ? cur[i] : 0u;
          row[j] = i < cnt ? W.row[en[j] & 0xFFu] : 0u;           // (a lane beyond the list steps row 0 and drops the result)
          pc[j] = fmx_rlr_piece(ix, row[j]);
        }
        FmxRlrStep st[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) st[j] = fmx_rlr_entry(ix, row[j], pc[j]);
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
          const uint32_t i = i0 + j * 64u + lane, x = en[j] & 0xFFu, done = (en[j] >> 8) + 1u;
          const uint32_t nr = fmx_rlr_finish(ix, row[j], st[j]);
          bool more = false;
          if (i < cnt) {
            FMX_CHECK(nr < ix.n);
            // TEXT: the walk is exactly `left` steps long and its length was stored at the start; else a sampled row ends it
            const bool last = TEXT ? done == W.len[x] : (nr & lmask) == 0u;
            W.row[x] = (!TEXT && last) ? nr >> ix.sa_level : nr;
            if (!TEXT && last) W.len[x] = done;
            more = !last;
          }
          const unsigned long long mm = __ballot(more);
          if (more) nxt[ncnt + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = x | (done << 8);
          ncnt += (uint32_t)__popcll(mm);
        }
      }
      cnt = ncnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // ---- TEXT: the walks that moved end on a phase-0 row: its sample's index from the phase piece ----
    if (TEXT) {
      uint32_t r2[4], t[4];
      uint4 pp[4];
      bool moved[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t x = j * 64u + lane;
        moved[j] = x < m && W.len[x] != 0u;
        r2[j] = moved[j] ? W.row[x] : 0u;
        const uint32_t pi = fmx_phase_piece(r2[j], ix.sa_level, t[j]);
        if (moved[j]) FMX_TOUCH(&ix.phase[pi]);
        pp[j] = ix.phase[pi];
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        uint32_t si;
        [[maybe_unused]] const uint32_t p2 = fmx_phase_decode(pp[j], t[j], ix.sa_level, si);
        FMX_CHECK(!moved[j] || p2 == 0u);
        if (moved[j]) W.row[j * 64u + lane] = si;
      }
    }
    // ---- the samples, the positions: (sa + steps) % len                                            rlfmi.rs:178-182 ----
    {
      uint32_t si[4], sv[4], ln[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t x = j * 64u + lane;
        si[j] = x < m ? W.row[x] : 0u;
        ln[j] = x < m ? W.len[x] : 0u;
        FMX_CHECK(si[j] < ix.nsamples);
        if (x < m) FMX_TOUCH(&ix.samples[si[j]]);
        sv[j] = ix.samples[si[j]];
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t x = j * 64u + lane;
        uint64_t v = (uint64_t)sv[j] + ln[j];
        if (v >= ix.n) v -= ix.n;
        if (x < m) { out_pos[h0 + x] = v; nsteps += ln[j]; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");        // the next pass reuses the wave's lists
  }
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}
#ifdef FMX_MEASURE
// The same walk in ONE launch (round 5, measurement builds only: FMX_RL_UNIFIED=1): a block expands its slice of at most
// 4096 hits in LDS (fmx_expand_slice: no rows array, no allocation, no expand launch) and its lanes walk them.  Measured
// against the two launches (profiles/r05/locate_mix_rlfm_one_launch.jsonl): repetitive text 10^3 x 10^5 hits 1.31 against
// 1.15 ms, mixed 1.48 against 1.20, config 4b 9.8 against 8.2 ms per batch -- the walk is bound by memory latency at full
// occupancy, and a block that waits for its slice's expansion is 16 waves that do not walk; on a random text with the run
// table (4.4 against 4.8 ms) it wins.  Not shipped.
template <bool TEXT>
__global__ __launch_bounds__(FMX_LOC_BLOCK) __attribute__((amdgpu_waves_per_eu(8))) void fmx_locate_rl_u_kernel(
    FmxDev ix, const uint64_t *__restrict__ s, const uint64_t *__restrict__ e, const uint64_t *__restrict__ off, uint64_t npat,
    uint64_t total, uint32_t hits_per_block, uint64_t *__restrict__ out_pos, uint64_t *__restrict__ steps_out) {
  __shared__ FmxSliceLds L;
  const uint64_t blo = (uint64_t)blockIdx.x * hits_per_block;
  if (blo >= total) return;                           // block-uniform
  const uint32_t bn = (uint32_t)(total - blo < hits_per_block ? total - blo : hits_per_block);
  FMX_CHECK(hits_per_block <= FMX_U_SLICE);
  if (fmx_expand_slice(L, s, e, off, npat, total, ix.n, blo, bn)) atomicOr(ix.status, 1u << FMX_ERR_ARG);
  uint64_t nsteps = 0;
  for (uint32_t x = threadIdx.x; x < bn; x += FMX_LOC_BLOCK)
    out_pos[blo + x] = fmx_rlfm_lane_get_sa<TEXT>(ix, L.rows[x], nsteps);
  if (steps_out && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
}
#endif

// ---- locate launch helpers (c = FmxLocateCall) ----
#define FMX_LOCQ_LAUNCH(c, gr, thr, hpb, chunk, Q, TEXT)                                             \
  hipLaunchKernelGGL((fmx_locate_f3q_kernel<Q, TEXT>), dim3(gr), dim3(thr), 0, (c).st,                 \
                     (c).dv.bw.lv[0].rec, (c).dv.samples, (c).dv.phase, (c).dv.n, (c).dv.sa_level,     \
                     (c).total, hpb, chunk, (c).rows, (c).pos, (c).steps)
#define FMX_LOCP_LAUNCH(c, gr, thr, hpb, chunk, Q, WCF)                                              \
  hipLaunchKernelGGL((fmx_locate_f3p_kernel<Q, WCF>), dim3(gr), dim3(thr), 0, (c).st,                  \
                     (c).dv.bw.lv[0].rec, (c).dv.samples, (c).dv.n, (c).dv.sa_level, (c).total, hpb,   \
                     chunk, (c).rows, (c).pos, (c).steps)
#define FMX_LOCT_LAUNCH(c, gr, thr, hpb, chunk, Q, WCF)                                              \
  hipLaunchKernelGGL((fmx_locate_f3t_kernel<Q, WCF>), dim3(gr), dim3(thr), 0, (c).st,                  \
                     (c).dv.walk, (c).dv.samples, (c).dv.n, (c).dv.nsamples, (c).total, hpb, chunk,    \
                     (c).rows, (c).pos, (c).steps)
// group per walk (any kind / any number of levels)
#define FMX_LOCATE_LAUNCH(c, grid, hpw, KIND, NL, SM)                                                \
  hipLaunchKernelGGL((fmx_locate_kernel<KIND, NL, SM>), dim3(grid), dim3(FMX_BLOCK), 0, (c).st,        \
                     (c).dv, (c).total, hpw, (c).rows, (c).pos, (c).steps)
#define FMX_LOCATE_KIND(c, grid, hpw, KIND, SM)                                                      \
  do {                                                                                               \
    if ((c).dv.bw.nlevels == 1) FMX_LOCATE_LAUNCH(c, grid, hpw, KIND, 1, SM);                        \
    else if ((c).dv.bw.nlevels == 2) FMX_LOCATE_LAUNCH(c, grid, hpw, KIND, 2, SM);                   \
    else FMX_LOCATE_LAUNCH(c, grid, hpw, KIND, 0, SM);                                               \
  } while (0)
// one walk per lane
#define FMX_EPL_LAUNCH3(c, gr, thr, hpb, KIND, NL, SM, KL, TX, WCF)                                  \
  hipLaunchKernelGGL((fmx_locate_ep_kernel<KIND, NL, SM, KL, TX, WCF>), dim3(gr), dim3(thr), 0,        \
                     (c).st, (c).dv, (c).total, hpb, (c).rows, (c).pos, (c).steps)
#ifdef FMX_MEASURE   // direct stores instead of the write-combining ring (FMX_VARIANT=26)
#define FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, KL, TX)                                  \
  do { if (wcf) FMX_EPL_LAUNCH3(c, gr, thr, hpb, KIND, NL, SM, KL, TX, true);                        \
       else FMX_EPL_LAUNCH3(c, gr, thr, hpb, KIND, NL, SM, KL, TX, false); } while (0)
#else
#define FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, KL, TX)                                  \
  FMX_EPL_LAUNCH3(c, gr, thr, hpb, KIND, NL, SM, KL, TX, true)
#endif
#define FMX_EPL_LAUNCH(c, gr, thr, hpb, wcf, KIND, NL, SM)                                           \
  do {                                                                                               \
    const bool klds_ = (c).dv.max_character < 1024u, text_ = (c).dv.phase != nullptr;                \
    if (klds_) { if (text_) FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, true, true);         \
                 else FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, true, false); }            \
    else { if (text_) FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, false, true);              \
           else FMX_EPL_LAUNCH2(c, gr, thr, hpb, wcf, KIND, NL, SM, false, false); }                 \
  } while (0)
// RLFM with the run table: no wavelet level is read (NL, KLDS irrelevant: one instantiation per select structure)
#define FMX_EPL_LFR(c, gr, thr, hpb, SM)                                                             \
  do {                                                                                               \
    if ((c).dv.phase != nullptr)                                                                     \
      hipLaunchKernelGGL((fmx_locate_ep_kernel<FMX_KIND_RLFM, 1, SM, false, true, true, true>), dim3(gr), dim3(thr), 0, \
                         (c).st, (c).dv, (c).total, hpb, (c).rows, (c).pos, (c).steps);              \
    else                                                                                             \
      hipLaunchKernelGGL((fmx_locate_ep_kernel<FMX_KIND_RLFM, 1, SM, false, false, true, true>), dim3(gr), dim3(thr), 0, \
                         (c).st, (c).dv, (c).total, hpb, (c).rows, (c).pos, (c).steps);              \
  } while (0)
#define FMX_EPL_SM(c, gr, thr, hpb, wcf, KIND, SM)                                                   \
  do {                                                                                               \
    if ((c).dv.bw.nlevels == 1) FMX_EPL_LAUNCH(c, gr, thr, hpb, wcf, KIND, 1, SM);                   \
    else if ((c).dv.bw.nlevels == 2) FMX_EPL_LAUNCH(c, gr, thr, hpb, wcf, KIND, 2, SM);              \
    else FMX_EPL_LAUNCH(c, gr, thr, hpb, wcf, KIND, 0, SM);                                          \
  } while (0)

#ifdef FMX_MEASURE
#include "fmx_measure.inc"       // fmx_tune() from the environment, fmx_measure_count(), fmx_measure_locate()
#else
static constexpr FmxTune fmx_tune() { return FmxTune{}; }
#endif

int fmx_launch_count(const fmx_index *idx, const void *d_pat, const uint64_t *d_off,
                     uint64_t npat, const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e,
                     uint64_t *d_cnt, hipStream_t st, unsigned max_blocks) {
  if (idx->is_wide) return fmxw_launch_count(idx, d_pat, d_off, npat, d_s0e0, d_s, d_e, d_cnt, st);
  const FmxDev dv = fmx_launch_dev(idx);
  if (npat == 0) return FMX_OK;
  if (max_blocks == 0 || max_blocks > FMX_MAX_BLOCKS) max_blocks = FMX_MAX_BLOCKS;
  if (idx->n == 0) {                                 // no record of any structure may be probed
    hipLaunchKernelGGL(fmx_count_empty_kernel, dim3(fmx_grid_for_groups((npat + 7) / 8)), dim3(FMX_BLOCK), 0,
                       st, dv.max_character, idx->sym_bytes, dv.status, d_pat, d_off, npat, d_s0e0, d_s, d_e,
                       d_cnt);
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  fmx_time_begin(idx, st);
  const FmxTune tn = fmx_tune();
  const FmxCountCall c{idx, dv, d_pat, d_off, npat, d_s0e0, d_s, d_e, d_cnt, idx->timing == 1 ? idx->d_steps : nullptr,
                       st, dv.kmer != nullptr && tn.use_kmer, fmx_grid_capped(npat, max_blocks)};
  const FmxMwm &w = dv.bw;
  const int sm = fmx_select_mode(idx, dv);
  bool done = false;
#ifdef FMX_MEASURE
  done = fmx_measure_count(c, tn, sm);               // the alternative kernels, when one was asked for
#endif
  if (done) {
  } else if (dv.pair_rec && idx->sym_bytes == 1 && tn.use_pair && !tn.generic) {
    // opt-in pair index: two symbols per probe
    if (c.km) FMX_PAIR_LAUNCH(c, true); else FMX_PAIR_LAUNCH(c, false);
  } else if (idx->kind == FMX_KIND_FM && idx->sym_bytes == 1 && w.nlevels == 1 && w.lv[0].fmt == 3 && !tn.generic) {
    // DNA (one 3-bit level): one 128-byte line per interval end and step, group per pattern
    if (c.km) FMX_F3_LAUNCH(c, 1, false, true); else FMX_F3_LAUNCH(c, 1, false, false);
  } else if (sm > 0 && !tn.generic) {
    // RLFM whose B / B' have a one-load select (stored positions or select blocks -- every index the
    // builder makes today): endpoint per lane (fmx_ep.h), 64 probes in flight per wave and stage.
    // FM indexes keep the group-per-pattern kernels: their steps are one light record probe per level
    // and those kernels already run at the request ceiling (endpoint per lane measured slower there:
    // DNA 1.11 vs 0.67 ms, sigma = 255 two levels 0.97 vs 0.74 ms, benchmarks/gpu/ep_fm_count.sh)
    const uint64_t eb = fmx_ep_count_blocks(npat, tn.ep_blocks);
    if (sm == 1) FMX_EP_SM(c, eb, FMX_KIND_RLFM, 1); else FMX_EP_SM(c, eb, FMX_KIND_RLFM, 2);
  } else if (idx->kind == FMX_KIND_FM) {
    FMX_COUNT_KIND(c, FMX_KIND_FM, -1);
  } else if (idx->kind == FMX_KIND_MULTI) {
    FMX_COUNT_KIND(c, FMX_KIND_MULTI, -1);
  } else {
    FMX_COUNT_KIND(c, FMX_KIND_RLFM, 0);             // hints + record search: valid for every vector
  }
  fmx_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

uint64_t fmx_offsets_tile_bytes(uint64_t npat) {
  uint64_t ntiles = (npat + FMX_SCAN_TILE - 1) / FMX_SCAN_TILE;
  if (ntiles == 0) ntiles = 1;
  return ((ntiles + 1) * sizeof(uint64_t) + 255u) & ~(uint64_t)255u;
}
// rows[total] (u32) + the list of long ranges behind them (fmx_expand_kernel: counter + FMX_EXPAND_LONGCAP entries)
static inline uint64_t fmx_rows_part_bytes(uint64_t total) {
  return ((total ? total : 1) * sizeof(uint32_t) + 255u) & ~(uint64_t)255u;
}
static const uint64_t kLongListBytes = (2u + 3u * (uint64_t)FMX_EXPAND_LONGCAP) * sizeof(unsigned long long);
uint64_t fmx_locate_rows_bytes(uint64_t total) {
  return fmx_rows_part_bytes(total) + ((kLongListBytes + 255u) & ~(uint64_t)255u);
}

int fmx_launch_offsets(const uint64_t *d_s, const uint64_t *d_e, uint64_t npat, uint64_t *d_off,
                       hipStream_t st, uint64_t *tile_ws) {
  uint64_t ntiles = (npat + FMX_SCAN_TILE - 1) / FMX_SCAN_TILE;
  if (ntiles == 0) ntiles = 1;
  uint64_t *tile = tile_ws;
  if (!tile_ws) FMX_HIP(fmx_dev_malloc_async((void **)&tile, (ntiles + 1) * sizeof(uint64_t), st));
  hipLaunchKernelGGL(fmx_tile_sums_kernel, dim3((unsigned)ntiles), dim3(FMX_BLOCK), 0, st, d_s, d_e,
                     npat, tile);
  hipLaunchKernelGGL(fmx_scan_tiles_kernel, dim3(1), dim3(FMX_BLOCK), 0, st, tile, ntiles);
  hipLaunchKernelGGL(fmx_tile_scan_kernel, dim3((unsigned)ntiles), dim3(FMX_BLOCK), 0, st, d_s, d_e,
                     npat, tile, ntiles, d_off);
  FMX_HIP(hipGetLastError());
  if (!tile_ws) FMX_HIP(hipFreeAsync(tile, st));
  return FMX_OK;
}

// does a locate batch of this index run the one-launch kernel (no rows array, no allocation, kernel launches only)?
bool fmx_locate_is_one_launch(const fmx_index *idx) {
  if (!idx || idx->is_wide) return false;
  const FmxTune tn = fmx_tune();
  const FmxMwm &w = idx->dev.bw;
  return idx->kind == FMX_KIND_FM && w.nlevels == 1 && w.lv[0].fmt == 3 && !tn.generic && idx->dev.phase && idx->dev.walk &&
         tn.walk_records && tn.unified && !tn.alt;
}
#ifdef FMX_MEASURE
// RLFM with the run table, batches that take the lane kernel (fmx_launch_locate below): one launch when asked for
static bool fmx_locate_rl_one_launch(const fmx_index *idx, const FmxDev &dv, const FmxTune &tn, uint64_t npat, uint64_t total) {
  return idx->kind == FMX_KIND_RLFM && !tn.generic && !tn.alt && tn.rl_unified && dv.lfrun && tn.walk_records && tn.wc &&
         fmx_select_mode(idx, dv) > 0 && total >= (uint64_t)tn.rl_ep_min && total / npat >= (uint64_t)tn.rl_lane_avg;
}
#endif
int fmx_launch_locate(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                      uint64_t npat, const uint64_t *d_off, uint64_t total, uint64_t *d_pos,
                      hipStream_t st, uint32_t *rows_ws) {
  if (idx->is_wide) return fmxw_launch_locate(idx, d_s, d_e, npat, d_off, total, d_pos, st);   // rows live in d_pos
  const FmxDev dv = fmx_launch_dev(idx);
  if (npat == 0 || total == 0) return FMX_OK;
  const FmxMwm &w = dv.bw;
  const FmxTune tn = fmx_tune();
  const bool dna = idx->kind == FMX_KIND_FM && w.nlevels == 1 && w.lv[0].fmt == 3 && !tn.generic;
  // the default DNA index (text order + walk records): ONE kernel that expands its slices itself -- no rows array
#ifdef FMX_MEASURE
  const bool rl_unified = fmx_locate_rl_one_launch(idx, dv, tn, npat, total);
#else
  constexpr bool rl_unified = false;
#endif
  const bool unified = fmx_locate_is_one_launch(idx) || rl_unified;
  // every other path: rows in their own read-only buffer (the walk's loads never alias its stores).  The caller's
  // workspace when there is one (nothing but kernel launches then: graph-capturable, no pool shared between streams)
  uint32_t *rows = unified ? nullptr : rows_ws;
  const bool own_rows = !unified && !rows_ws;
  if (own_rows) FMX_HIP(fmx_dev_malloc_async((void **)&rows, fmx_locate_rows_bytes(total), st));
  if (!unified && total / npat >= 1024) {
    // few patterns with very many hits each: a block per run of consecutive 4096-hit slices (a lane per pattern left a
    // thousand patterns of 10^5 hits to sixteen waves: 1.7 ms, more than the walk of those 10^8 hits; here 0.2 ms)
    uint64_t eb = (total + FMX_U_SLICE - 1) / FMX_U_SLICE;
    if (eb > 2048) eb = 2048;
    hipLaunchKernelGGL(fmx_expand_slices_kernel, dim3((unsigned)eb), dim3(FMX_LOC_BLOCK), 0, st, d_s, d_e, d_off, npat,
                       total, dv.n, rows, dv.status);
  } else if (!unified) {
    // a lane per pattern (ranges over 32 rows by the lane's wave): the faster of the two up to hundreds of hits per
    // pattern (config 4: 24 against 51 us per 2^20 singletons; config 4b, 750 hits per pattern: 1.0 against 5.1 ms)
    uint64_t eb = (npat + FMX_BLOCK - 1) / FMX_BLOCK;
    if (eb > FMX_MAX_BLOCKS * 4) eb = FMX_MAX_BLOCKS * 4;
    // per RANGE (round 6): a batch that can hold a range of 2^16+ rows lists such ranges for a second, grid-wide pass
    // instead of leaving each to one wave (10^6 singletons + ten patterns of 10^7 hits: 3.4 ms for the batch against
    // 0.15 + 1.09 for its parts); a 16-byte clear and one -- usually empty -- launch per batch of 2^18+ hits
    // (... and only when the batch has at least 2^16 hits MORE than patterns: with a hit per pattern -- configs 3 / 4 --
    // no range can be that long, and the two extra launches cost such a batch 5-8 %: 0.129 -> 0.139 ms with the run table)
    unsigned long long *longs = (total >= (1u << 18) && total >= npat + FMX_EXPAND_DEFER)
                                    ? (unsigned long long *)((uint8_t *)rows + fmx_rows_part_bytes(total)) : nullptr;
    if (longs) hipLaunchKernelGGL(fmx_zero_words_kernel, dim3(1), dim3(64), 0, st, longs, 2u);   // (a kernel, not a memset: the
                                                      // workspace forms promise kernel launches only)
    hipLaunchKernelGGL(fmx_expand_kernel<uint32_t>, dim3((unsigned)eb), dim3(FMX_BLOCK), 0, st, d_s, d_e,
                       d_off, npat, rows, total, dv.n, dv.status, longs);
    // (512 blocks: an empty list -- the usual case -- costs the launch, and 2048 blocks that only read the counter took 5 us)
    if (longs) hipLaunchKernelGGL(fmx_expand_long_kernel<uint32_t>, dim3(512), dim3(FMX_BLOCK), 0, st, longs, rows);
  }
  fmx_time_begin(idx, st);
  const FmxLocateCall c{idx, dv, total, rows, d_pos, idx->timing == 1 ? idx->d_steps : nullptr, st};
  const int sm = fmx_select_mode(idx, dv);
  bool done = false;
#ifdef FMX_MEASURE
  done = fmx_measure_locate(c, tn, sm);              // the alternative kernels, when one was asked for
#endif
  if (done) {
#ifdef FMX_MEASURE
  } else if (rl_unified) {
    const unsigned gr = (unsigned)((total + FMX_U_SLICE - 1) / FMX_U_SLICE);
    if (dv.phase) hipLaunchKernelGGL(fmx_locate_rl_u_kernel<true>, dim3(gr), dim3(FMX_LOC_BLOCK), 0, st, dv, d_s, d_e, d_off, npat, total, (uint32_t)FMX_U_SLICE, d_pos, c.steps);
    else hipLaunchKernelGGL(fmx_locate_rl_u_kernel<false>, dim3(gr), dim3(FMX_LOC_BLOCK), 0, st, dv, d_s, d_e, d_off, npat, total, (uint32_t)FMX_U_SLICE, d_pos, c.steps);
#endif
  } else if (unified) {
    // slices of at most 4096 hits (their rows live in LDS), at least 256 of them -- 512 for large batches: two
    // blocks per CU (64 VGPRs) -- and tickets of 64 hits once every wave gets that many (below: one hit per walk
    // slot of every wave, see the two-kernel path)
    const int q = tn.walks ? tn.walks : (total >= (1u << 16) ? 4 : 1);
    uint64_t nb = tn.loc_blocks ? (uint64_t)tn.loc_blocks : (total >= (4u << 20) ? 512 : 256);
    if (nb < (total + FMX_U_SLICE - 1) / FMX_U_SLICE) nb = (total + FMX_U_SLICE - 1) / FMX_U_SLICE;
    const uint64_t per_wave = (total + nb * (FMX_LOC_BLOCK / 64) - 1) / (nb * (FMX_LOC_BLOCK / 64));
    uint32_t chunk = (uint32_t)((per_wave + 7) / 8 * 8);
    if (chunk < 8u * (uint32_t)q) chunk = 8u * (uint32_t)q;
    if (chunk > FMX_LCHUNK) chunk = FMX_LCHUNK;
    uint32_t hpb;
    unsigned gr;
    c.slice(nb, chunk, hpb, gr);
    if (hpb > FMX_U_SLICE) { hpb = FMX_U_SLICE; gr = (unsigned)((total + hpb - 1) / hpb); }   // (4096 is a multiple of every ticket size)
#define FMX_LOCU_LAUNCH(Q, WCF)                                                                      \
    hipLaunchKernelGGL((fmx_locate_f3u_kernel<Q, WCF>), dim3(gr), dim3(FMX_LOC_BLOCK), 0, st, dv.walk, dv.samples, dv.n, \
                       dv.nsamples, d_s, d_e, d_off, npat, total, hpb, chunk, (uint32_t)tn.adj_clusters, d_pos, c.steps,   \
                       dv.status)
    if (chunk == FMX_LCHUNK && tn.wc) {
      if (q == 4) FMX_LOCU_LAUNCH(4, true); else if (q == 2) FMX_LOCU_LAUNCH(2, true); else FMX_LOCU_LAUNCH(1, true);
    } else {
      if (q == 4) FMX_LOCU_LAUNCH(4, false); else if (q == 2) FMX_LOCU_LAUNCH(2, false); else FMX_LOCU_LAUNCH(1, false);
    }
  } else if (dna) {
    // DNA (one 3-bit level): walk state distributed over the lanes of a group, 4 walks per group when the
    // batch is large enough to keep every group busy with them
    const int q = tn.walks ? tn.walks : (total >= (1u << 16) ? 4 : 1);
    const unsigned lthreads = (unsigned)tn.loc_threads;
    // 1024-thread blocks (one LDS hit queue per 16 waves); 50 VGPRs at Q = 4, so two of them fit a CU.
    // Batches of millions of hits want both (config 3b: 14.7 ms on 512 blocks, 18.7 on 256); a 2^20-hit
    // batch is bound by the chain of its longest walks and as fast on one (benchmarks/gpu/f3q_grid_sweep.sh)
    const uint64_t nb = tn.loc_blocks ? (uint64_t)tn.loc_blocks : (total >= (4u << 20) ? 512 : 256);
    // rows per ticket: 64 once every wave gets that many; below, one hit per slot of every wave
    // (mid-size batches are latency-bound: 1.3e5 hits 54 us against 84 us with 64-row tickets)
    const uint64_t per_wave = (total + nb * (lthreads / 64) - 1) / (nb * (lthreads / 64));
    uint32_t chunk = (uint32_t)((per_wave + 7) / 8 * 8);
    if (chunk < 8u * (uint32_t)q) chunk = 8u * (uint32_t)q;      // the first round hands out 8 q hits at once
    if (chunk > FMX_LCHUNK) chunk = FMX_LCHUNK;
    uint32_t hpb;
    unsigned gr;
    c.slice(nb, chunk, hpb, gr);
    if (dv.phase && dv.walk && tn.walk_records && total / npat >= 64 && total >= (1u << 16) && !tn.walks && !tn.loc_blocks) {
      // long intervals: a lane per walk on consecutive hits
      uint64_t lb = (total + FMX_BLOCK - 1) / FMX_BLOCK;
      if (lb > 8192) lb = 8192;
      hipLaunchKernelGGL(fmx_locate_walk_lane_kernel, dim3((unsigned)lb), dim3(FMX_BLOCK), 0, c.st, dv.walk, dv.samples, dv.n,
                         dv.nsamples, c.total, c.rows, c.pos, c.steps);
    } else if (dv.phase && dv.walk && tn.walk_records) {   // text-order sampling with walk records: no phase probes
      if (chunk == FMX_LCHUNK && tn.wc) {
#ifdef FMX_MEASURE   // FMX_VARIANT=15: 8 walks per group -- 86 VGPRs; walk kernel 0.0805 ms against 0.0774 with 4, config 3b 10.9 against 9.8 ms
        if (q == 8) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 8, true); else
#endif
        if (q == 4) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 4, true);
        else if (q == 2) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 2, true);
        else FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 1, true);
      } else {
#ifdef FMX_MEASURE
        if (q == 8) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 8, false); else
#endif
        if (q == 4) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 4, false);
        else if (q == 2) FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 2, false);
        else FMX_LOCT_LAUNCH(c, gr, lthreads, hpb, chunk, 1, false);
      }
    } else if (dv.phase) {  // text-order sampling (FMX_FLAG_TEXT_ORDER, or a file written that way)
      if (q == 4) FMX_LOCQ_LAUNCH(c, gr, lthreads, hpb, chunk, 4, true);
      else if (q == 2) FMX_LOCQ_LAUNCH(c, gr, lthreads, hpb, chunk, 2, true);
      else FMX_LOCQ_LAUNCH(c, gr, lthreads, hpb, chunk, 1, true);
    } else if (chunk == FMX_LCHUNK && tn.wc) {   // whole 64-hit tickets: positions through the write-combining ring
      if (q == 4) FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 4, true);
      else if (q == 2) FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 2, true);
      else FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 1, true);
    } else {
      if (q == 4) FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 4, false);
      else if (q == 2) FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 2, false);
      else FMX_LOCP_LAUNCH(c, gr, lthreads, hpb, chunk, 1, false);
    }
  } else {
    // FM over several levels: the one-walk-per-lane kernel pays once the batch is throughput-bound
    // (7.9e8 hits: 7.4e9 hits/s against 5.9e9); up to 2^20 hits the group-per-walk kernel has the shorter
    // step (benchmarks/gpu/small_shapes.py: 175 vs 187 us at n = 2^16, 247 vs 297 us at n = 2^27)
    const bool fm_ep = idx->kind == FMX_KIND_FM && w.nlevels >= 2 && total >= (uint64_t)tn.fm_ep_min;
    // RLFM: one walk per lane wins once there are enough hits to keep its 64-wide rounds busy: 2^20 hits
    // 0.22-0.38 ms against 0.44-0.67 ms, but 2^16 hits 0.10-0.16 against 0.09-0.12 ms and fewer about
    // equal (benchmarks/gpu/small_shapes.py) -- below 2^18 hits the group-per-walk kernel runs
    const bool rl_ep = sm > 0 && total >= (uint64_t)tn.rl_ep_min;
    if ((rl_ep || fm_ep) && !tn.generic) {
      // 2^20 hits finish soonest on one 1024-thread block per CU; large batches want every wave the
      // registers admit (96 VGPRs -> 5 per SIMD): two 640-thread blocks per CU
      const bool big = total >= (4u << 20);
      const unsigned thr = tn.ep_loc_threads ? (unsigned)tn.ep_loc_threads : (big ? 640u : (unsigned)FMX_LOC_BLOCK);
      uint32_t hpb;
      unsigned gr;
      c.slice(tn.ep_loc_blocks ? (uint64_t)tn.ep_loc_blocks : (big ? 512 : 256), FMX_LCHUNK, hpb, gr);
      if (fm_ep) FMX_EPL_SM(c, gr, thr, hpb, tn.wc, FMX_KIND_FM, 0);
      else if (dv.lfrun && tn.walk_records && tn.wc && total / npat >= (uint64_t)tn.rl_lane_avg && !tn.alt) {
        // the run table, two hits per pattern or more on average: a lane per walk on consecutive hits, for every hit.
        // Round 5 measured the choice per 64-hit ticket that the DNA kernel makes (lane kernel for the tickets of adjacent
        // rows, the queue kernel for the others, a flag byte per ticket between the two launches): on the repetitive text
        // and on a random text with the run table it lost to this kernel alone on every mixed batch (10^6 singletons +
        // 10^3 x 10^5 hits: 1.73 against 1.19 ms; 4 x 10^6 singletons + 100 x 10^5: 1.44 against 0.64 ms) -- the queue
        // kernel's edge on scattered rows is 15 % at best with the run table (two lane-wise requests per step either
        // way), less than what splitting a batch costs (profiles/r05/locate_mix_*.jsonl).  Round 4 took this kernel from
        // an average of 64 hits per pattern on, which sent the second batch above (average 3.5) to the queue kernel.
        if (tn.rl_rounds) {                                       // round 6: four tickets per wave at a time, in rounds
          const uint64_t per_block = (uint64_t)FMX_RLR_HITS * (FMX_RLR_BLOCK / 64);
          uint64_t lb = (total + per_block - 1) / per_block;
          if (lb > (uint64_t)tn.rl_rounds_blocks) lb = (uint64_t)tn.rl_rounds_blocks;
          if (dv.phase) hipLaunchKernelGGL(fmx_locate_rl_rounds_kernel<true>, dim3((unsigned)lb), dim3(FMX_RLR_BLOCK), 0, c.st, c.dv, c.total, c.rows, c.pos, c.steps);
          else hipLaunchKernelGGL(fmx_locate_rl_rounds_kernel<false>, dim3((unsigned)lb), dim3(FMX_RLR_BLOCK), 0, c.st, c.dv, c.total, c.rows, c.pos, c.steps);
        } else {
          uint64_t lb = (total + FMX_BLOCK - 1) / FMX_BLOCK;
          if (lb > 8192) lb = 8192;
          if (dv.phase) hipLaunchKernelGGL(fmx_locate_rl_lane_kernel<true>, dim3((unsigned)lb), dim3(FMX_BLOCK), 0, c.st, c.dv, c.total, c.rows, c.pos, c.steps);
          else hipLaunchKernelGGL(fmx_locate_rl_lane_kernel<false>, dim3((unsigned)lb), dim3(FMX_BLOCK), 0, c.st, c.dv, c.total, c.rows, c.pos, c.steps);
        }
      }
      else if (dv.lfrun && tn.walk_records && tn.wc) {          // the run table: two lane-wise requests per LF step
        if (sm == 1) FMX_EPL_LFR(c, gr, thr, hpb, 1); else FMX_EPL_LFR(c, gr, thr, hpb, 2);
      }
      else if (sm == 1) FMX_EPL_SM(c, gr, thr, hpb, tn.wc, FMX_KIND_RLFM, 1);
      else FMX_EPL_SM(c, gr, thr, hpb, tn.wc, FMX_KIND_RLFM, 2);
    } else {
      // group per walk: a wave owns a contiguous share of the hits
      uint64_t nwaves = (total + 7) / 8;
      const uint64_t max_waves = (uint64_t)FMX_MAX_BLOCKS * (FMX_BLOCK / 64);
      if (nwaves > max_waves) nwaves = max_waves;
      const uint64_t hpw = (total + nwaves - 1) / nwaves;
      const unsigned grid = (unsigned)((nwaves + FMX_BLOCK / 64 - 1) / (FMX_BLOCK / 64));
      if (idx->kind == FMX_KIND_FM) FMX_LOCATE_KIND(c, grid, hpw, FMX_KIND_FM, -1);
      else if (idx->kind == FMX_KIND_MULTI) FMX_LOCATE_KIND(c, grid, hpw, FMX_KIND_MULTI, -1);
      else if (sm == 1) FMX_LOCATE_KIND(c, grid, hpw, FMX_KIND_RLFM, 1);   // per select structure
      else if (sm == 2) FMX_LOCATE_KIND(c, grid, hpw, FMX_KIND_RLFM, 2);
      else FMX_LOCATE_KIND(c, grid, hpw, FMX_KIND_RLFM, 0);   // hints + record search: valid for every vector
    }
  }
  fmx_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  if (own_rows) FMX_HIP(hipFreeAsync(rows, st));
  return FMX_OK;
}

int fmx_launch_expand64(const uint64_t *d_s, const uint64_t *d_e, const uint64_t *d_off, uint64_t npat,
                        uint64_t *d_out, uint64_t total, uint64_t n, uint32_t *status, hipStream_t st) {
  uint64_t eb = (npat + FMX_BLOCK - 1) / FMX_BLOCK;
  if (eb > FMX_MAX_BLOCKS * 4) eb = FMX_MAX_BLOCKS * 4;
  hipLaunchKernelGGL(fmx_expand_kernel<uint64_t>, dim3((unsigned)eb), dim3(FMX_BLOCK), 0, st, d_s, d_e, d_off, npat,
                     d_out, total, n, status);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmx_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i,
                      uint64_t k, uint64_t *d_out, hipStream_t st) {
  if (idx->is_wide) return fmxw_launch_scalar(idx, op, d_c, d_i, k, d_out, st);
  const FmxDev dv = fmx_launch_dev(idx);
  if (k == 0) return FMX_OK;
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_scalar_kernel<FMX_KIND_FM>, dim3(fmx_grid_for_groups(k)), dim3(FMX_BLOCK),
                       0, st, dv, op, d_c, d_i, k, d_out);
  else if (idx->kind == FMX_KIND_MULTI)
    hipLaunchKernelGGL(fmx_scalar_kernel<FMX_KIND_MULTI>, dim3(fmx_grid_for_groups(k)), dim3(FMX_BLOCK),
                       0, st, dv, op, d_c, d_i, k, d_out);
  else
    hipLaunchKernelGGL(fmx_scalar_kernel<FMX_KIND_RLFM>, dim3(fmx_grid_for_groups(k)),
                       dim3(FMX_BLOCK), 0, st, dv, op, d_c, d_i, k, d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

template <typename T>
static int fmx_launch_extract_t(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len,
                                int forward, T *d_out, uint64_t *d_out_len, uint64_t *d_out_next, hipStream_t st) {
  const FmxDev dv = fmx_launch_dev(idx);
  const dim3 grid(fmx_grid_for_groups(nrows)), block(FMX_BLOCK);
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL((fmx_extract_kernel<FMX_KIND_FM, T>), grid, block, 0, st, dv, d_rows, nrows, len,
                       forward, d_out, d_out_len, d_out_next);
  else if (idx->kind == FMX_KIND_MULTI)
    hipLaunchKernelGGL((fmx_extract_kernel<FMX_KIND_MULTI, T>), grid, block, 0, st, dv, d_rows, nrows,
                       len, forward, d_out, d_out_len, d_out_next);
  else
    hipLaunchKernelGGL((fmx_extract_kernel<FMX_KIND_RLFM, T>), grid, block, 0, st, dv, d_rows, nrows,
                       len, forward, d_out, d_out_len, d_out_next);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
int fmx_launch_extract(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len,
                       int forward, void *d_out, uint64_t *d_out_len, uint64_t *d_out_next,
                       hipStream_t st) {
  if (idx->is_wide) return fmxw_launch_extract(idx, d_rows, nrows, len, forward, d_out, d_out_len, d_out_next, st);
  if (nrows == 0) return FMX_OK;
  if (idx->sym_bytes == 1) return fmx_launch_extract_t(idx, d_rows, nrows, len, forward, (uint8_t *)d_out, d_out_len, d_out_next, st);
  if (idx->sym_bytes == 2) return fmx_launch_extract_t(idx, d_rows, nrows, len, forward, (uint16_t *)d_out, d_out_len, d_out_next, st);
  return fmx_launch_extract_t(idx, d_rows, nrows, len, forward, (uint32_t *)d_out, d_out_len, d_out_next, st);
}

int fmx_launch_kmer_build(const fmx_index *idx, uint2 *d_table, uint32_t k, uint32_t bits, hipStream_t st) {
  const FmxDev dv = fmx_launch_dev(idx);
  const uint64_t ncodes = 1ull << (bits * k);
  const dim3 grid(fmx_grid_for_groups(ncodes)), block(FMX_BLOCK);
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_kmer_build_kernel<FMX_KIND_FM>, grid, block, 0, st, dv, d_table, k, bits);
  else if (idx->kind == FMX_KIND_MULTI)
    hipLaunchKernelGGL(fmx_kmer_build_kernel<FMX_KIND_MULTI>, grid, block, 0, st, dv, d_table, k, bits);
  else
    hipLaunchKernelGGL(fmx_kmer_build_kernel<FMX_KIND_RLFM>, grid, block, 0, st, dv, d_table, k, bits);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmx_launch_compute_K(const FmxMwm &w, const uint64_t *d_cs, uint32_t *d_K,
                         uint32_t max_character, hipStream_t st) {
  uint64_t groups = (uint64_t)max_character + 1;
  unsigned grid = (unsigned)((groups * FMX_GROUP + FMX_BLOCK - 1) / FMX_BLOCK);
  hipLaunchKernelGGL(fmx_compute_K_kernel, dim3(grid), dim3(FMX_BLOCK), 0, st, w, d_cs, d_K,
                     max_character);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// ---------------------------------------------------------------------------
// census build only (make census): line log of the launches that follow fmx_census_begin
// ---------------------------------------------------------------------------
#ifdef FMX_CENSUS
extern "C" int fmx_census_begin(void *d_log, uint64_t cap_entries, void *d_count) {
  FmxCensusDev h;
  h.log = (unsigned long long *)d_log;
  h.count = (unsigned long long *)d_count;
  h.cap = cap_entries;
  FMX_HIP(hipMemset(d_count, 0, 8));
  FMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fmx_census_dev), &h, sizeof h));
  return FMX_OK;
}
extern "C" int fmx_census_end(void) {
  FmxCensusDev h;
  h.log = nullptr; h.count = nullptr; h.cap = 0;
  FMX_HIP(hipDeviceSynchronize());
  FMX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fmx_census_dev), &h, sizeof h));
  return FMX_OK;
}
#endif
