// fmx_wide.hip -- the query kernels of WIDE indexes: texts of n >= 2^32 - 16 symbols, whose rows and positions
// need the reference's `usize` (fm_index.rs:86-95, 127-140; wrapper.rs:103-134, 203-217, 238-242).
//
// A second instantiation next to the 32-bit engine of fmx_query.hip, which it leaves untouched: FMIndex /
// FMIndexWithLocate over a one-level alphabet (max_character <= 7).  Layout: fmx_internal.h (FmxWideDev) -- the
// same 128-byte records, counters relative to a 2^31-row superblock, a table of 64-bit bases.  A rank is one
// 128-byte line fetched by an 8-lane group (one dwordx4 per lane), popcount + three DPP adds in 32 bits, and
// one 64-bit add of the base, whose 8-byte load is issued together with the record load.
//
// Shapes are the simple ones (a group owns a pattern / a walk from start to end): these kernels exist so that
// an index beyond 2^32 symbols answers exactly what the reference answers; the tuned shapes of the 32-bit
// engine (state machines, distributed walk state, write-combining ring) are not repeated here.
#include "fmx_device.h"

#define FMXW_BLOCK 256
#define FMXW_MAX_BLOCKS 2048

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

// the superblock bases in LDS (they are read with a data-dependent symbol in every step): up to FMXW_LDS_SB
// superblocks = n < 2^37; beyond that the kernels read them from global memory
#define FMXW_LDS_SB 64u
__device__ __forceinline__ const uint64_t *fmxw_stage_bases(const FmxWideDev &w, uint64_t *lds) {
  if (w.nsb > FMXW_LDS_SB) return w.base;
  for (uint32_t t = threadIdx.x; t < w.nsb * 8u; t += blockDim.x) lds[t] = w.base[t];
  __syncthreads();
  return lds;
}

// SearchWrapper::search for a batch (wrapper.rs:103-124): a group per pattern.  Per step the two record loads and
// the NEXT pattern symbol are requested together and waited for once; the base of the step's symbol comes from LDS.
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_count_kernel(
    FmxWideDev w, const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
    const uint64_t *__restrict__ s0e0, uint64_t *__restrict__ out_s, uint64_t *__restrict__ out_e,
    uint64_t *__restrict__ out_cnt, uint64_t *__restrict__ steps_out) {
  __shared__ uint64_t lds_base[FMXW_LDS_SB * 8u];
  const uint64_t *base = fmxw_stage_bases(w, lds_base);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t ptot = npat ? off[npat] : 0;       // symbols the caller declares behind `pat`
  uint64_t nsteps = 0;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    const uint64_t pbeg = off[k], pend = off[k + 1];
    uint64_t j = pend - pbeg;
    // offsets that go backwards or leave the pattern buffer: refuse, do not read
    bool bad = pend < pbeg || pend > ptot;
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
      const uint64_t ba = base[(size_t)(s >> w.sb_shift) * 8u + c];
      const uint64_t bb = base[(size_t)(e >> w.sb_shift) * 8u + c];
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
// position array, so a wide locate needs no workspace.  A group runs FMXW_WALKS walks at a time (their record loads
// are requested together) and takes the next hit for a slot the moment its walk ends.
#define FMXW_WALKS 4
__global__ __launch_bounds__(FMXW_BLOCK) void fmxw_walk_kernel(FmxWideDev w, uint64_t total, uint64_t *__restrict__ io,
                                                                uint64_t *__restrict__ steps_out) {
  __shared__ uint64_t lds_base[FMXW_LDS_SB * 8u];
  const uint64_t *base = fmxw_stage_bases(w, lds_base);
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint64_t lmask = (1ull << w.sa_level) - 1ull;
  constexpr uint64_t NONE = ~0ull;
  uint64_t h[FMXW_WALKS], row[FMXW_WALKS], steps[FMXW_WALKS];
  uint64_t next = gid, nsteps = 0;                  // hits gid, gid + ngroups, ... belong to this group
  bool any = false;
#pragma unroll
  for (int q = 0; q < FMXW_WALKS; q++) {
    h[q] = NONE; row[q] = 0; steps[q] = 0;
    if (next < total) { h[q] = next; row[q] = io[next]; next += ngroups; any = true; }
  }
  while (any) {
    // rows that are not of this index, walks that stand on a sampled row: finish, hand the slot to the next hit
#pragma unroll
    for (int q = 0; q < FMXW_WALKS; q++) {
      if (h[q] == NONE) continue;
      bool done = false;
      uint64_t v = NONE;
      if (row[q] >= w.n) {                          // refuse, do not read
        if (g == 0) atomicOr(w.status, 1u << FMX_ERR_ARG);
        done = true;
      } else if ((row[q] & lmask) == 0) {           // Some(sa): (sa + steps) % len           fm_index.rs:131-133
        v = w.samples[row[q] >> w.sa_level] + steps[q];
        if (v >= w.n) v -= w.n;
        done = true;
      }
      if (done) {
        if (g == 0) io[h[q]] = v;
        nsteps += steps[q];
        h[q] = NONE; steps[q] = 0;
        if (next < total) { h[q] = next; row[q] = io[next]; next += ngroups; }
      }
    }
    // one LF step of every walk that is not on a sampled row: the record loads first, then the decodes
    uint4 p[FMXW_WALKS];
    bool walk[FMXW_WALKS];
    any = false;
#pragma unroll
    for (int q = 0; q < FMXW_WALKS; q++) {
      walk[q] = h[q] != NONE && row[q] < w.n && (row[q] & lmask) != 0;
      any |= h[q] != NONE;
      if (walk[q]) p[q] = w.rec[(size_t)(row[q] >> 8) * 8u + g];
    }
#pragma unroll
    for (int q = 0; q < FMXW_WALKS; q++) {
      if (!walk[q]) continue;                       // None: i = lf_map(i); steps += 1        fm_index.rs:134-137
      const uint32_t off = (uint32_t)row[q] & 255u;
      const uint32_t sym = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p[q], off & 31u) : 0u);
      row[q] = base[(size_t)(row[q] >> w.sb_shift) * 8u + sym] + fmx_group_sum(fmx_piece_rank<3>(p[q], off, sym, g));
      steps[q]++;
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

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static int fmxw_unsupported(const char *what) {
  fmx_set_error(FMX_ERR_UNSUPPORTED, what);
  return FMX_ERR_UNSUPPORTED;
}
static void fmxw_time_begin(const fmx_index *idx, hipStream_t st) {
  if (idx->timing) {
    fmx_index *m = const_cast<fmx_index *>(idx);
    (void)hipMemsetAsync(m->d_steps, 0, sizeof(uint64_t), st);
    (void)hipEventRecord(m->ev0, st);
  }
}
static void fmxw_time_end(const fmx_index *idx, hipStream_t st) {
  if (idx->timing) {
    fmx_index *m = const_cast<fmx_index *>(idx);
    (void)hipEventRecord(m->ev1, st);
    m->ev_valid = 1;
  }
}

int fmxw_launch_count(const fmx_index *idx, const void *d_pat, const uint64_t *d_off, uint64_t npat,
                      const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e, uint64_t *d_cnt, hipStream_t st) {
  if (npat == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  fmxw_time_begin(idx, st);
  hipLaunchKernelGGL(fmxw_count_kernel, dim3(fmxw_grid(npat)), dim3(FMXW_BLOCK), 0, st, w, (const uint8_t *)d_pat, d_off,
                     npat, d_s0e0, d_s, d_e, d_cnt, idx->timing ? idx->d_steps : nullptr);
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
  hipLaunchKernelGGL(fmxw_walk_kernel, dim3(fmxw_grid(total)), dim3(FMXW_BLOCK), 0, st, w, total, d_pos,
                     idx->timing ? idx->d_steps : nullptr);
  fmxw_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i, uint64_t k,
                       uint64_t *d_out, hipStream_t st) {
  if (op > 5) return fmxw_unsupported("piece_id needs a multi-pieces index");
  if (k == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  hipLaunchKernelGGL(fmxw_scalar_kernel, dim3(fmxw_grid(k)), dim3(FMXW_BLOCK), 0, st, w, op, d_c, d_i, k, d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_export_l(const fmx_index *idx, void *d_out, hipStream_t st) {
  const FmxWideDev w = fmxw_dev(idx);
  hipLaunchKernelGGL(fmxw_export_l_kernel, dim3(FMXW_MAX_BLOCKS * 4), dim3(FMXW_BLOCK), 0, st, w, (uint8_t *)d_out);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmxw_launch_extract(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint32_t len, int forward,
                        void *d_out, uint64_t *d_out_len, uint64_t *d_out_next, hipStream_t st) {
  if (nrows == 0) return FMX_OK;
  const FmxWideDev w = fmxw_dev(idx);
  hipLaunchKernelGGL(fmxw_extract_kernel, dim3(fmxw_grid(nrows)), dim3(FMXW_BLOCK), 0, st, w, d_rows, nrows, len, forward,
                     (uint8_t *)d_out, d_out_len, d_out_next);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
