// fmx_ep.h -- "endpoint per lane" device functions (round 2).
//
// The round-1 kernels give one 8-lane group ONE pattern: the group's state (s, e, c, ...) is
// replicated in all 8 lanes and a group keeps 1-2 cache lines in flight.  That is enough for the
// one-line-per-step DNA count (it runs at the memory system's request ceiling), but an RLFM step is
// four DEPENDENT lines (B -> S level 0 -> S level 1 -> B' select), and with 1-2 lines per group
// the chip ran at 30 G lines/s of the 55 G/s it sustains (profiles/microbench/gather_r02.txt).
//
// Here every LANE owns one interval endpoint (count: lane 2q = s, lane 2q+1 = e of the group's q-th
// pattern; locate: one walk per lane), so the per-unit state costs one register, not eight:
//   * probes that need a single 16-byte piece -- rank1 on B / B' (each piece carries its absolute
//     count, fmx_internal.h), select blocks, stored positions, SA samples -- are done LANE-WISE:
//     64 independent 16-byte requests per wave-instruction, no cross-lane traffic;
//   * ranks over a 128-byte wavelet record stay cooperative, in ROUNDS: for q = 0..7 the group
//     broadcasts endpoint q's (position, code) (ds_bpermute), all 8 lanes load their piece of that
//     record -- eight lines in flight per lane before the first popcount -- then every record is
//     reduced with popcounts + three DPP adds and the result handed back to lane q.
// A wave therefore has 64 lines in flight in every stage; 2-4 waves per SIMD saturate the memory
// system and the registers are there for it.
#pragma once
#include "fmx_device.h"

#define FMX_NONE 0xFFFFFFFFu

// census only: a lane-wise probe is logged unless PAIRED and the lane's partner (the other interval
// end of the same pattern, lane ^ 1) asks for the same 128-byte line -- that is ONE line of the byte
// model; the even lane logs it
#ifdef FMX_CENSUS
template <bool PAIRED>
__device__ __forceinline__ void fmx_touch_lane(const void *p, bool want = true) {
  const unsigned long long line = want ? (unsigned long long)(uintptr_t)p >> 7 : ~0ull;
  bool log = want;
  if (PAIRED) {
    const uint32_t lo = fmx_dpp_xor1((uint32_t)line), hi = fmx_dpp_xor1((uint32_t)(line >> 32));
    const bool same = lo == (uint32_t)line && hi == (uint32_t)(line >> 32);
    if (same && (threadIdx.x & 1u)) log = false;
  }
  if (log) fmx_touch(p, true);
}
#define FMX_TOUCH_LANE(PAIRED, p, want) fmx_touch_lane<PAIRED>((const void *)(p), (want))
#else
#define FMX_TOUCH_LANE(PAIRED, p, want) do { } while (0)
#endif

// value of lane (base + q) for all lanes of the group
__device__ __forceinline__ uint32_t fmx_grp_bcast(uint32_t v, uint32_t base, uint32_t q) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + q) << 2), (int)v);
}

// ---- lane-wise probes of the RLFM bit vectors ----------------------------------------------
// rank1(i) (clamped like vers-vecs RsVec::rank1), the bit B[i] (0 past the end) and `next` = the
// first one at or after i when it lies in the 96-bit piece just loaded (FMX_NONE otherwise).
// Piece p of the vector is bits [96p, 96p+96) -- records are 8 consecutive pieces, so the piece
// index is simply i / 96.
struct FmxProbe { uint4 pc; uint32_t pidx, bit; };
template <bool PAIRED>
__device__ __forceinline__ FmxProbe fmx_bits_probe_issue(const FmxBits &bv, uint32_t i,
                                                         [[maybe_unused]] bool live = true) {
  FmxProbe pr;
  if (i > bv.len) i = bv.len;
  pr.pidx = fmx_div3(i >> 5);                 // i / 96
  pr.bit = i - pr.pidx * FMX_BITS_PER_PIECE;
  FMX_CHECK(pr.pidx < bv.nrec * 8u);
  FMX_TOUCH_LANE(PAIRED, &bv.rec[pr.pidx], live);
  pr.pc = bv.rec[pr.pidx];
  return pr;
}
__device__ __forceinline__ uint32_t fmx_bits_probe_rank(const FmxProbe &pr, uint32_t &bit_i, uint32_t &next) {
  const uint32_t bit = pr.bit;
  const uint32_t m0 = fmx_lowmask(bit < 32u ? bit : 32u);
  const uint32_t m1 = bit > 32u ? fmx_lowmask(bit - 32u < 32u ? bit - 32u : 32u) : 0u;
  const uint32_t m2 = bit > 64u ? fmx_lowmask(bit - 64u) : 0u;
  const uint32_t c = __popc(pr.pc.y & m0) + __popc(pr.pc.z & m1) + __popc(pr.pc.w & m2);
  const uint32_t word = bit < 32u ? pr.pc.y : (bit < 64u ? pr.pc.z : pr.pc.w);
  bit_i = (word >> (bit & 31u)) & 1u;
  const uint32_t y = pr.pc.y & ~m0, z = pr.pc.z & ~m1, w = pr.pc.w & ~m2;
  uint32_t cand = FMX_NONE;
  if (y) cand = (uint32_t)__builtin_ctz(y);
  else if (z) cand = 32u + (uint32_t)__builtin_ctz(z);
  else if (w) cand = 64u + (uint32_t)__builtin_ctz(w);
  next = cand != FMX_NONE ? cand + pr.pidx * FMX_BITS_PER_PIECE : FMX_NONE;
  return pr.pc.x + c;
}

// select1(k) on a vector with stored positions (SM = 1) or select blocks (SM = 2): ONE lane-wise
// load.  issue() starts it, finish() returns the position, or FMX_NONE when the block says its 64
// ones do not fit (then the caller goes through fmx_ep_select_slow).  k >= ones gives len
// (vers-vecs RsVec::select1).
struct FmxSel { uint4 blk; uint32_t k; bool valid; };
template <int SM, bool PAIRED>
__device__ __forceinline__ FmxSel fmx_ep_select_issue(const FmxBits &bv, uint32_t k, bool want,
                                                      [[maybe_unused]] bool live = true) {
  FmxSel s;
  s.valid = k < bv.ones;
  s.k = s.valid ? k : 0u;
  s.blk = make_uint4(0u, 0u, 0u, 0u);
  want = want && bv.ones != 0u;
  if (SM == 1) FMX_TOUCH_LANE(PAIRED, &bv.pos[s.k], want && live);
  else FMX_TOUCH_LANE(PAIRED, &bv.dsel[s.k >> bv.dsel_shift], want && live);
  if (want) {
    if (SM == 1) s.blk.x = bv.pos[s.k];
    else s.blk = bv.dsel[s.k >> bv.dsel_shift];
  }
  return s;
}
template <int SM>
__device__ __forceinline__ uint32_t fmx_ep_select_finish(const FmxBits &bv, const FmxSel &s) {
  if (!s.valid) return bv.len;
  if (SM == 1) return s.blk.x;
  if (s.blk.x == FMX_NONE) return FMX_NONE;
  return fmx_dsel_pos(s.blk, s.k, bv.dsel_shift);
}
// the rare select a block cannot answer: the group serves its lanes one after the other with the
// cooperative record search (wave-uniform loop; groups without a request run it on a dummy)
__device__ __forceinline__ void fmx_ep_select_slow(const FmxBits &bv, uint32_t k, bool need, uint32_t base,
                                                   uint32_t g, uint32_t &out) {
  unsigned long long pend = __ballot(need);
  while (pend) {
    const uint32_t gm = (uint32_t)(pend >> base) & 0xFFu;
    const uint32_t src = gm ? (uint32_t)__ffs((int)gm) - 1u : 0u;
    uint32_t kk = fmx_grp_bcast(k, base, src);
    if (!gm) kk = 0u;
    // hints + record search, never the block again
    uint32_t r;
    if (kk >= bv.ones) {
      r = bv.len;
    } else {
      const uint32_t h = kk / FMX_SEL_STEP;
      FMX_CHECK(h + 1 < bv.nsel);
      uint32_t lo = bv.sel[h], hi = bv.sel[h + 1];
      while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        FMX_TOUCH_G0(g, &bv.rec[(size_t)mid * 8u]);
        if (bv.rec[(size_t)mid * 8u].x <= kk) lo = mid; else hi = mid - 1u;
      }
      FMX_TOUCH_G0(g, &bv.rec[(size_t)lo * 8u]);
      const uint4 pc = bv.rec[(size_t)lo * 8u + g];
      const uint32_t p = fmx_group_sum(pc.x <= kk ? 1u : 0u) - 1u;
      const uint32_t rem = kk - pc.x;
      const uint32_t c0 = __popc(pc.y), c1 = __popc(pc.z);
      uint32_t pos;
      if (rem < c0) pos = fmx_select32(pc.y, rem);
      else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
      else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
      pos = fmx_group_sum((g == p) ? pos : 0u);
      r = lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
    }
    if (gm && g == src) { out = r; need = false; }
    pend = __ballot(need);
  }
}

// ---- one cooperative round over a wavelet level ----------------------------------------------
// Every lane passes ITS endpoint's position (and, unless ACCESS, its level code).  Returns for the
// lane's own endpoint: rank = counter[code] + #{entries before pos with that code} (absolute
// counters: the next level's position, fmx_internal.h) and match = [entry pos itself has the code].
// ACCESS: the code is READ at entry pos (WaveletMatrix::get) and the rank is of that code.
// `match` rides in bit 31 of the group sum while the index has fewer than 2^31 rows; beyond that
// (`wide`) it is broadcast from the one lane that owns it.
template <int FMT, bool ACCESS, bool PAIRED, bool NOMATCH = false>
__device__ __forceinline__ void fmx_ep_round(const uint4 *__restrict__ rec, uint32_t pos, uint32_t &code,
                                             bool live, uint32_t base, uint32_t g, uint32_t &rank,
                                             uint32_t &match, bool wide = false) {
  constexpr int SH = (FMT == 3) ? 8 : 7;
  constexpr uint32_t OM = (FMT == 3) ? 255u : 127u;
  constexpr uint32_t PER = (FMT == 3) ? 32u : 16u;
  constexpr int PSH = (FMT == 3) ? 5 : 4;
  uint32_t bp[8], bc[8];
  uint4 p[8];
  // endpoint q is skipped when it is dead in EVERY group of the wave (wave-uniform): in the tail of a
  // locate batch a wave holds a few long walks and a round then costs their instructions only
  const unsigned long long lv = __ballot(live);
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    bp[q] = fmx_grp_bcast(pos, base, q);
    bc[q] = ACCESS ? 0u : fmx_grp_bcast(code, base, q);
    const uint4 *r = rec + (size_t)(bp[q] >> SH) * 8u;
#ifdef FMX_CENSUS   // dead endpoints (record 0, cached) are not the algorithm's lines; the odd endpoint is
                    // the same pattern's other interval end: same record = ONE line
    if (fmx_grp_bcast(live ? 1u : 0u, base, q) &&
        !(PAIRED && (q & 1u) && (bp[q] >> SH) == (bp[q - 1u] >> SH)))
      FMX_TOUCH_G0(g, r);
#endif
    p[q] = r[g];
  }
#pragma unroll
  for (uint32_t q = 0; q < 8; q++) {
    if (!(lv & (0x0101010101010101ull << q))) continue;
    const uint32_t off = bp[q] & OM;
    const bool mine = g == (off >> PSH);
    uint32_t cd = bc[q];
    if (ACCESS) cd = fmx_group_sum(mine ? fmx_piece_code<FMT>(p[q], off & (PER - 1u)) : 0u);
    const uint32_t mt = fmx_piece_match<FMT>(p[q], cd);
    int nb = (int)off - (int)(g * PER);
    nb = nb < 0 ? 0 : (nb > (int)PER ? (int)PER : nb);
    uint32_t v = __popc(mt & (uint32_t)((1ull << nb) - 1ull));
    if (FMT == 3) v += (g == cd) ? p[q].x : 0u;
    else v += (g == (cd >> 1)) ? ((cd & 1u) ? p[q].y : p[q].x) : 0u;
    // the match bit of the entry at `pos` lives in ONE lane (`mine`): on indexes below 2^31 rows it
    // rides in bit 31 of the group sum, beyond that (`wide`, wave-uniform) it is broadcast from that lane
    const uint32_t mb = (mt >> (off & (PER - 1u))) & 1u;
    uint32_t msum = 1u;
    if (!ACCESS && !NOMATCH) {
      if (wide) msum = fmx_grp_bcast(mb, base, off >> PSH);
      else v |= mine ? (mb << 31) : 0u;
    }
    const uint32_t sum = fmx_group_sum(v);
    if (g == q) {
      rank = (ACCESS || NOMATCH || wide) ? sum : (sum & 0x7FFFFFFFu);
      match = (ACCESS || NOMATCH) ? 1u : (wide ? msum : (sum >> 31));
      if (ACCESS) code = cd;
    }
  }
}
template <bool ACCESS, bool PAIRED, bool NOMATCH = false>
__device__ __forceinline__ void fmx_ep_level(const FmxLevel &L, uint32_t pos, uint32_t &code, bool live,
                                             uint32_t base, uint32_t g, uint32_t &rank, uint32_t &match,
                                             bool wide = false) {
  FMX_CHECK((pos >> (L.fmt == 3 ? 8 : 7)) < L.nrec);
  if (L.fmt == 3) fmx_ep_round<3, ACCESS, PAIRED, NOMATCH>(L.rec, pos, code, live, base, g, rank, match, wide);
  else fmx_ep_round<4, ACCESS, PAIRED, NOMATCH>(L.rec, pos, code, live, base, g, rank, match, wide);
}

// ---- RLFMIndexBackend::lf_map2 for 8 endpoints per group (rlfmi.rs:135-143) -------------------
// Lane-wise:  j = b.rank1(i), the bit b[i], the run start when it sits in the piece loaded.
// Rounds:     ONE rank chain at lo = b.rank1(i+1) - 1 (the run holding row i) gives both
//             s.rank(lo, c) and m = [s[lo] == c]; s.rank(lo+1, c) = s.rank(lo, c) + m, so
//             nr = s.rank(j, c) with j in {lo, lo+1}, and get_l(i) == c is m  (rlfmi.rs:137-138).
// Lane-wise:  bp.select1(cs[c] + nr); + i - b.select1(j) when m                  (rlfmi.rs:139-141)
// Dead lanes pass i = 0, c = 0, live = false and ignore the result (`live` only keeps their cached
// dummy probes out of the census).
template <int NL, int SM>
__device__ __forceinline__ uint32_t fmx_rlfm_ep_lf_map2(const FmxDev &ix, uint32_t c, uint32_t i, bool live,
                                                        uint32_t base, uint32_t g) {
  const uint32_t kc = ix.K[c];
  const bool wide = ix.n >= (1u << 31);                      // wave-uniform
  const FmxProbe pr = fmx_bits_probe_issue<true>(ix.b, i, live);
  uint32_t bit, nx;
  const uint32_t j = fmx_bits_probe_rank(pr, bit, nx);       // b.rank1(i)            rlfmi.rs:136
  const uint32_t lo = j - 1u + bit;                          // b.rank1(i + 1) - 1    rlfmi.rs:124
  uint32_t pos = lo, r = 0, m = 1u;
  const uint32_t nl = NL ? (uint32_t)NL : ix.bw.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = ix.bw.lv[l];
    uint32_t code = (c >> L.shift) & L.mask, mt;
    fmx_ep_level<false, true>(L, pos, code, live, base, g, r, mt, wide);
    m &= mt;
    pos = r;                                                 // C_l[code] is folded into the counters
  }
  const uint32_t nr = kc + r + (bit ? 0u : m);               // cs[c] + s.rank(j, c)  rlfmi.rs:137,139
  // run start b.select1(j) = first one at or after i: usually in the piece; else one more probe, issued
  // TOGETHER with the B' select (both are consumed below, so it adds no dependent stage; issued ahead of
  // the rank rounds, as in round 2, its block sat in 5 registers through both rounds for nothing)
  const FmxSel ss = fmx_ep_select_issue<SM, true>(ix.b, j, nx == FMX_NONE, live);
  const FmxSel sf = fmx_ep_select_issue<SM, true>(ix.bp, nr, true, live);
  uint32_t f = fmx_ep_select_finish<SM>(ix.bp, sf);          // bp.select1(cs[c] + nr)
  uint32_t st = nx;
  if (nx == FMX_NONE) st = fmx_ep_select_finish<SM>(ix.b, ss);
  if (SM == 2) {                                             // blocks that do not hold their 64 ones
    fmx_ep_select_slow(ix.bp, nr, f == FMX_NONE, base, g, f);
    fmx_ep_select_slow(ix.b, j, m && st == FMX_NONE, base, g, st);
  }
  return m ? f + i - st : f;                                 // rlfmi.rs:138-142
}

// ---- RLFMIndexBackend::get_l + lf_map for 8 walks per group (rlfmi.rs:122-133) -----------------
// One access+rank chain at lo = b.rank1(i+1) - 1 yields c = s[lo] and s.rank(lo, c); since s[lo] = c,
// s.rank(j, c) = s.rank(lo, c) + (j - lo).  K[] comes from LDS (`kt`, staged by the kernel).
template <int NL, int SM>
__device__ __forceinline__ uint32_t fmx_rlfm_ep_lf_map(const FmxDev &ix, const uint32_t *kt, uint32_t i, bool live,
                                                       uint32_t base, uint32_t g, uint32_t &sym) {
  const FmxProbe pr = fmx_bits_probe_issue<false>(ix.b, i, live);
  uint32_t bit, nx;
  const uint32_t j = fmx_bits_probe_rank(pr, bit, nx);       // b.rank1(i)
  const uint32_t lo = j - 1u + bit;
  uint32_t pos = lo, r = 0;
  sym = 0;
  const uint32_t nl = NL ? (uint32_t)NL : ix.bw.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = ix.bw.lv[l];
    uint32_t code = 0, mt;
    fmx_ep_level<true, false>(L, pos, code, live, base, g, r, mt);
    sym |= code << L.shift;
    pos = r;
  }
  const uint32_t nr = kt[sym] + r + (bit ? 0u : 1u);         // cs[c] + s.rank(j, c)   rlfmi.rs:129-130
  const FmxSel ss = fmx_ep_select_issue<SM, false>(ix.b, j, nx == FMX_NONE, live);   // with the B' select
  const FmxSel sf = fmx_ep_select_issue<SM, false>(ix.bp, nr, true, live);
  uint32_t f = fmx_ep_select_finish<SM>(ix.bp, sf);
  uint32_t st = nx;
  if (nx == FMX_NONE) st = fmx_ep_select_finish<SM>(ix.b, ss);
  if (SM == 2) {
    fmx_ep_select_slow(ix.bp, nr, f == FMX_NONE, base, g, f);
    fmx_ep_select_slow(ix.b, j, st == FMX_NONE, base, g, st);
  }
  return f + i - st;                                         // rlfmi.rs:132
}

// ---- the same lf_map through the run table (FmxDev::lfrun; round 4) ----------------------------------
// Rows of one run map to consecutive rows, and lfrun[j] is lf_map of run j's first row, so
//     lf_map(i) = lfrun[lo] + (i - start of run lo),   lo = b.rank1(i + 1) - 1            (rlfmi.rs:122-133)
// Two lane-wise requests -- the B piece of row i, then the table entry (and, only when the run starts before the piece,
// one select on B: stored position / select block, the rare block that does not hold its ones through the group's
// cooperative search) -- and no wavelet record at all.
template <int SM>
__device__ __forceinline__ uint32_t fmx_rlfm_ep_lf_run(const FmxDev &ix, uint32_t i, bool live, uint32_t base, uint32_t g) {
  const FmxProbe pr = fmx_bits_probe_issue<false>(ix.b, i, live);
  uint32_t bit, nx;
  const uint32_t j = fmx_bits_probe_rank(pr, bit, nx);       // b.rank1(i)
  const uint32_t lo = j - 1u + bit;                          // the run holding row i
  // its start = the last one at or before i: in the piece unless the run began before it
  uint32_t st = FMX_NONE;
  {
    const uint32_t b1 = pr.bit + 1u;                         // bits [0, pr.bit] of the 96-bit window
    const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
    const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
    const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
    const uint32_t y = pr.pc.y & m0, z = pr.pc.z & m1, w = pr.pc.w & m2;
    uint32_t cand = FMX_NONE;
    if (w) cand = 95u - (uint32_t)__builtin_clz(w);
    else if (z) cand = 63u - (uint32_t)__builtin_clz(z);
    else if (y) cand = 31u - (uint32_t)__builtin_clz(y);
    if (cand != FMX_NONE) st = cand + pr.pidx * FMX_BITS_PER_PIECE;
  }
  FMX_CHECK(!live || lo < ix.b.ones);
  FMX_TOUCH_LANE(false, &ix.lfrun[live ? lo : 0u], live);
  const uint32_t f = ix.lfrun[live ? lo : 0u];               // lf_map(start of the run)
  const FmxSel ss = fmx_ep_select_issue<SM, false>(ix.b, lo, live && st == FMX_NONE, live);   // b.select1(lo)
  if (st == FMX_NONE) st = fmx_ep_select_finish<SM>(ix.b, ss);
  if (SM == 2) fmx_ep_select_slow(ix.b, lo, live && st == FMX_NONE, base, g, st);
  return f + i - st;
}

// ---- FMIndexBackend::get_l + lf_map for 8 walks per group (fm_index.rs:82-91) -------------------
// access + rank along the same positions, one round per wavelet level
template <int NL>
__device__ __forceinline__ uint32_t fmx_fm_ep_lf_map(const FmxDev &ix, const uint32_t *kt, uint32_t i, bool live,
                                                     uint32_t base, uint32_t g, uint32_t &sym) {
  uint32_t pos = i, r = 0;
  sym = 0;
  const uint32_t nl = NL ? (uint32_t)NL : ix.bw.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = ix.bw.lv[l];
    uint32_t code = 0, mt;
    fmx_ep_level<true, false>(L, pos, code, live, base, g, r, mt);
    sym |= code << L.shift;
    pos = r;                                                 // C_l[code] is folded into the counters
  }
  return kt[sym] + r;                                        // fm_index.rs:86-91
}

// ---- FMIndexBackend::lf_map2 for 8 endpoints per group (fm_index.rs:93-95) ---------------------
// one rank round per wavelet level; the two ends of a pattern sit in adjacent lanes
template <int NL>
__device__ __forceinline__ uint32_t fmx_fm_ep_lf_map2(const FmxDev &ix, uint32_t c, uint32_t i, bool live,
                                                      uint32_t base, uint32_t g) {
  const uint32_t kc = ix.K[c];
  uint32_t pos = i, r = 0;
  const uint32_t nl = NL ? (uint32_t)NL : ix.bw.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = ix.bw.lv[l];
    uint32_t code = (c >> L.shift) & L.mask, mt;
    fmx_ep_level<false, true, true>(L, pos, code, live, base, g, r, mt);
    pos = r;                                                 // C_l[code] is folded into the counters
  }
  return kc + r;
}
