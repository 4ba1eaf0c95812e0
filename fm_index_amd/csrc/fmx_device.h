// fmx_device.h -- gfx950 device functions: 8-lane-group cooperative rank / access
// on the 128-byte records described in fmx_internal.h.
//
// Execution shape: a 64-lane wavefront is 8 groups of 8 lanes.  One group owns one
// query endpoint; lane g of the group loads piece g (16 B) of the record, so one
// global_load_dwordx4 wave-instruction fetches 8 whole, distinct 128-B lines.  The
// per-piece popcounts are summed with three DPP adds inside the group (quad_perm,
// quad_perm, row_half_mirror) -- no LDS traffic, no bpermute.
#pragma once
// LDS words that lanes of one wave hand to each other (write-combining rings, the hit queue's row windows): volatile,
// so that every access is a real LDS operation in program order -- and typed as LDS (address space 3): the compiler's
// address-space inference leaves volatile accesses through generic pointers as FLAT instructions, each followed by a
// wait for every vector-memory operation of the wave in flight (rounds 1-4 shipped that; measured in round 5)
typedef __attribute__((address_space(3))) uint32_t fmx_lds_u32;
typedef __attribute__((address_space(3))) unsigned long long fmx_lds_u64;
#define FMX_LDS_U32(p) ((volatile fmx_lds_u32 *)(p))
#define FMX_LDS_U64(p) ((volatile fmx_lds_u64 *)(p))
#include "fmx_internal.h"

#define FMX_GROUP 8

// Debug build (make debug -> libfmx_debug.so, -DFMX_DEBUG_BOUNDS): every index into the HBM arrays
// is range-checked and a violation traps the kernel (the GPU pool has no address sanitizer).
#ifdef FMX_DEBUG_BOUNDS
#define FMX_CHECK(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#else
#define FMX_CHECK(cond) do { } while (0)
#endif

// Census build (make census -> libfmx_census.so, -DFMX_CENSUS; measurement only, never shipped as
// libfmx.so): every load of an index line appends its 128-byte line address to a log, so that
// bench.py can count the lines a launch REQUESTS and the DISTINCT lines among them (the byte model
// of DESIGN.md section 4).  A cooperative 8-lane record load is logged once (lane 0 of the group).
#ifdef FMX_CENSUS
struct FmxCensusDev { unsigned long long *log; unsigned long long *count; unsigned long long cap; };
static __device__ FmxCensusDev fmx_census_dev;
// A log entry is the 128-byte line address; bit 63 marks a NARROW request (a lane-wise probe of at most 16
// bytes: B / B' pieces, select blocks, stored positions, phase pieces, samples, k-mer entries) as opposed to a
// whole 128-byte record fetched by the 8 lanes of a group -- bench.py prices the two differently.
__device__ __forceinline__ void fmx_touch(const void *p, bool narrow) {
  if (fmx_census_dev.log) {
    const unsigned long long i = atomicAdd(fmx_census_dev.count, 1ull);
    if (i < fmx_census_dev.cap)
      fmx_census_dev.log[i] = ((unsigned long long)(uintptr_t)p >> 7) | (narrow ? 1ull << 63 : 0ull);
  }
}
#define FMX_TOUCH(p) fmx_touch((const void *)(p), true)                                        // lane-wise probe
#define FMX_TOUCH_G0(g, p) do { if ((g) == 0) fmx_touch((const void *)(p), false); } while (0)  // 128-byte record
#define FMX_TOUCH_G0N(g, p) do { if ((g) == 0) fmx_touch((const void *)(p), true); } while (0)  // one word / block, read by the group
#else
#define FMX_TOUCH(p) do { } while (0)
#define FMX_TOUCH_G0(g, p) do { } while (0)
#define FMX_TOUCH_G0N(g, p) do { } while (0)
#endif

// pattern / text symbol i of a buffer whose symbols are sb bytes wide (Character, character.rs)
__device__ __forceinline__ uint32_t fmx_load_sym(const void *p, uint32_t sb, uint64_t i) {
  if (sb == 1) return ((const uint8_t *)p)[i];
  if (sb == 2) return ((const uint16_t *)p)[i];
  return ((const uint32_t *)p)[i];
}

__device__ __forceinline__ uint32_t fmx_dpp_xor1(uint32_t v) {  // quad_perm [1,0,3,2]
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t fmx_dpp_xor2(uint32_t v) {  // quad_perm [2,3,0,1]
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t fmx_dpp_half_mirror(uint32_t v) {  // lane i <- 7-i (per 8)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
}
// sum over the 8 lanes of a group; every lane receives the total
__device__ __forceinline__ uint32_t fmx_group_sum(uint32_t v) {
  v += fmx_dpp_xor1(v);
  v += fmx_dpp_xor2(v);
  v += fmx_dpp_half_mirror(v);
  return v;
}
// minimum over the 8 lanes of a group
__device__ __forceinline__ uint32_t fmx_group_min(uint32_t v) {
  v = min(v, fmx_dpp_xor1(v));
  v = min(v, fmx_dpp_xor2(v));
  v = min(v, fmx_dpp_half_mirror(v));
  return v;
}

// value of lane (4 * (lane / 4) + q) for the four lanes of a quad (DPP quad_perm [q,q,q,q])
template <int QQ>
__device__ __forceinline__ uint32_t fmx_quad_bcast_c(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, QQ * 0x55, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t fmx_quad_bcast(uint32_t v, int q) {
  return q == 0 ? fmx_quad_bcast_c<0>(v) : q == 1 ? fmx_quad_bcast_c<1>(v) : q == 2 ? fmx_quad_bcast_c<2>(v)
                                                                                    : fmx_quad_bcast_c<3>(v);
}

// value of lane (8 * (lane / 8) + q) for the eight lanes of a group: broadcast inside the quad that holds
// lane q, then mirrored into the group's other quad (row_half_mirror, written to that quad's banks only)
template <int QQ>
__device__ __forceinline__ uint32_t fmx_oct_bcast_c(uint32_t v) {
  const int t = __builtin_amdgcn_update_dpp(0, (int)v, (QQ & 3) * 0x55, 0xF, 0xF, true);
  return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, QQ < 4 ? 0xA : 0x5, false);
}
__device__ __forceinline__ uint32_t fmx_oct_bcast(uint32_t v, int q) {
  switch (q) {
    case 0: return fmx_oct_bcast_c<0>(v);
    case 1: return fmx_oct_bcast_c<1>(v);
    case 2: return fmx_oct_bcast_c<2>(v);
    case 3: return fmx_oct_bcast_c<3>(v);
    case 4: return fmx_oct_bcast_c<4>(v);
    case 5: return fmx_oct_bcast_c<5>(v);
    case 6: return fmx_oct_bcast_c<6>(v);
    default: return fmx_oct_bcast_c<7>(v);
  }
}

// ---- per-piece helpers -------------------------------------------------------
// bitmask (over the piece's entries) of entries whose level code == code
template <int FMT>
__device__ __forceinline__ uint32_t fmx_piece_match(const uint4 &p, uint32_t code) {
  // bit k of the code, sign-extended (v_bfe_i32): all ones where the plane must hold a 1
  const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)code, 0u, 1u);
  const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)code, 1u, 1u);
  const uint32_t m2 = (uint32_t)__builtin_amdgcn_sbfe((int)code, 2u, 1u);
  if (FMT == 3) {
    return ~((p.y ^ m0) | (p.z ^ m1) | (p.w ^ m2));
  } else {
    const uint32_t m3 = (uint32_t)__builtin_amdgcn_sbfe((int)code, 3u, 1u);
    const uint32_t lo = (p.z ^ m0) | (p.w ^ m2);         // planes 0 and 2 in the low halves
    const uint32_t hi = (p.z ^ m1) | (p.w ^ m3);         // planes 1 and 3 in the high halves
    return ~(lo | (hi >> 16)) & 0xFFFFu;
  }
}
// code stored at entry `bit` of this piece
template <int FMT>
__device__ __forceinline__ uint32_t fmx_piece_code(const uint4 &p, uint32_t bit) {
  if (FMT == 3) {   // three v_bfe_u32 + two v_lshl_or_b32
    const uint32_t b0 = __builtin_amdgcn_ubfe(p.y, bit, 1u), b1 = __builtin_amdgcn_ubfe(p.z, bit, 1u),
                   b2 = __builtin_amdgcn_ubfe(p.w, bit, 1u);
    return (b2 << 2) | ((b1 << 1) | b0);
  } else {
    return ((p.z >> bit) & 1u) | (((p.z >> (16 + bit)) & 1u) << 1) | (((p.w >> bit) & 1u) << 2) |
           (((p.w >> (16 + bit)) & 1u) << 3);
  }
}
// this lane's contribution to rank_code(pos): masked popcount + the record counter
template <int FMT>
__device__ __forceinline__ uint32_t fmx_piece_rank(const uint4 &p, uint32_t off, uint32_t code,
                                                   uint32_t g) {
  constexpr int PER = (FMT == 3) ? 32 : 16;
  int nb = (int)off - (int)(g * PER);
  nb = nb < 0 ? 0 : (nb > PER ? PER : nb);
  uint32_t mask = (uint32_t)((1ull << nb) - 1ull);
  uint32_t v = __popc(fmx_piece_match<FMT>(p, code) & mask);
  if (FMT == 3) {
    v += (g == code) ? p.x : 0u;
  } else {
    v += (g == (code >> 1)) ? ((code & 1u) ? p.y : p.x) : 0u;
  }
  return v;
}

// `prev` (census only): position whose record the same pattern has just asked for -- a second
// request for the same line is ONE line of the byte model, so it is not logged again
template <int FMT>
__device__ __forceinline__ uint4 fmx_load_piece(const FmxLevel &L, uint32_t pos, uint32_t g,
                                                [[maybe_unused]] uint32_t prev = 0xFFFFFFFFu) {
  constexpr int SH = (FMT == 3) ? 8 : 7;
  FMX_CHECK((pos >> SH) < L.nrec);
#ifdef FMX_CENSUS
  if (prev == 0xFFFFFFFFu || (prev >> SH) != (pos >> SH)) FMX_TOUCH_G0(g, &L.rec[(size_t)(pos >> SH) * 8u]);
#endif
  return L.rec[(size_t)(pos >> SH) * 8u + g];
}
template <int FMT>
__device__ __forceinline__ uint32_t fmx_off(uint32_t pos) {
  return (FMT == 3) ? (pos & 255u) : (pos & 127u);
}

// rank of `code` among the first `pos` entries of level L  (all 8 lanes get it)
__device__ __forceinline__ uint32_t fmx_level_rank(const FmxLevel &L, uint32_t pos, uint32_t code,
                                                   uint32_t g) {
  if (L.fmt == 3) {
    uint4 p = fmx_load_piece<3>(L, pos, g);
    return fmx_group_sum(fmx_piece_rank<3>(p, fmx_off<3>(pos), code, g));
  } else {
    uint4 p = fmx_load_piece<4>(L, pos, g);
    return fmx_group_sum(fmx_piece_rank<4>(p, fmx_off<4>(pos), code, g));
  }
}

// rank_c(i) over the whole multi-ary wavelet matrix, minus the symbol-only start
// chain (folded into K[c]): the value r with lf_map2(c,i) = K[c] + r  (fm_index.rs:93-95)
// NL > 0 fixes the number of levels at compile time: the loops unroll and the level descriptors
// stay in SGPRs instead of being re-read from the kernel arguments inside the search loop.
template <int NL = 0>
__device__ __forceinline__ uint32_t fmx_mwm_rank(const FmxMwm &w, uint32_t c, uint32_t pos,
                                                 uint32_t g) {
  uint32_t r = 0;
  const uint32_t nl = NL ? (uint32_t)NL : w.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = w.lv[l];
    uint32_t code = (c >> L.shift) & L.mask;
    r = fmx_level_rank(L, pos, code, g);
    if (l + 1 < nl) pos = r;  // C_l[code] is folded into the counters
  }
  return r;
}

// both endpoints of one backward-search step, loads issued together
template <int NL = 0>
__device__ __forceinline__ void fmx_mwm_rank2(const FmxMwm &w, uint32_t c, uint32_t ps,
                                              uint32_t pe, uint32_t g, uint32_t &rs,
                                              uint32_t &re) {
  rs = 0;
  re = 0;
  const uint32_t nl = NL ? (uint32_t)NL : w.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = w.lv[l];
    uint32_t code = (c >> L.shift) & L.mask;
    if (L.fmt == 3) {
      uint4 a = fmx_load_piece<3>(L, ps, g);
      uint4 b = fmx_load_piece<3>(L, pe, g, ps);
      rs = fmx_group_sum(fmx_piece_rank<3>(a, fmx_off<3>(ps), code, g));
      re = fmx_group_sum(fmx_piece_rank<3>(b, fmx_off<3>(pe), code, g));
    } else {
      uint4 a = fmx_load_piece<4>(L, ps, g);
      uint4 b = fmx_load_piece<4>(L, pe, g, ps);
      rs = fmx_group_sum(fmx_piece_rank<4>(a, fmx_off<4>(ps), code, g));
      re = fmx_group_sum(fmx_piece_rank<4>(b, fmx_off<4>(pe), code, g));
    }
    if (l + 1 < nl) {  // C_l[code] is folded into the counters
      ps = rs;
      pe = re;
    }
  }
}

// N positions of the same symbol at once: per level all N record loads are issued before the
// first popcount (more lines in flight per group; equal lines are merged by the L1)
template <int N, int NL = 0>
__device__ __forceinline__ void fmx_mwm_rankN(const FmxMwm &w, uint32_t c, uint32_t (&pos)[N],
                                              uint32_t g, uint32_t (&r)[N]) {
  const uint32_t nl = NL ? (uint32_t)NL : w.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = w.lv[l];
    const uint32_t code = (c >> L.shift) & L.mask;
    uint4 p[N];
    if (L.fmt == 3) {
#pragma unroll
      for (int q = 0; q < N; q++) p[q] = fmx_load_piece<3>(L, pos[q], g, q ? pos[q - 1] : 0xFFFFFFFFu);
#pragma unroll
      for (int q = 0; q < N; q++)
        r[q] = fmx_group_sum(fmx_piece_rank<3>(p[q], fmx_off<3>(pos[q]), code, g));
    } else {
#pragma unroll
      for (int q = 0; q < N; q++) p[q] = fmx_load_piece<4>(L, pos[q], g, q ? pos[q - 1] : 0xFFFFFFFFu);
#pragma unroll
      for (int q = 0; q < N; q++)
        r[q] = fmx_group_sum(fmx_piece_rank<4>(p[q], fmx_off<4>(pos[q]), code, g));
    }
    if (l + 1 < nl) {
#pragma unroll
      for (int q = 0; q < N; q++) pos[q] = r[q];  // C_l[code] is folded into the counters
    }
  }
}

// access + rank along the same positions (fm_index.rs:82-91: get_l then rank of that
// symbol): returns r with lf_map(i) = K[sym] + r and the symbol itself.
template <int NL = 0>
__device__ __forceinline__ uint32_t fmx_mwm_lf(const FmxMwm &w, uint32_t pos, uint32_t g,
                                               uint32_t &sym) {
  uint32_t r = 0;
  sym = 0;
  const uint32_t nl = NL ? (uint32_t)NL : w.nlevels;
#pragma unroll
  for (uint32_t l = 0; l < nl; l++) {
    const FmxLevel &L = w.lv[l];
    uint32_t code;
    if (L.fmt == 3) {
      uint4 p = fmx_load_piece<3>(L, pos, g);
      uint32_t off = fmx_off<3>(pos);
      code = fmx_group_sum((g == (off >> 5)) ? fmx_piece_code<3>(p, off & 31u) : 0u);
      r = fmx_group_sum(fmx_piece_rank<3>(p, off, code, g));
    } else {
      uint4 p = fmx_load_piece<4>(L, pos, g);
      uint32_t off = fmx_off<4>(pos);
      code = fmx_group_sum((g == (off >> 4)) ? fmx_piece_code<4>(p, off & 15u) : 0u);
      r = fmx_group_sum(fmx_piece_rank<4>(p, off, code, g));
    }
    sym |= code << L.shift;
    if (l + 1 < nl) pos = r;  // C_l[code] folded
  }
  return r;
}

// position of the r-th (0-based) set bit of w; r < popc(w)
__device__ __forceinline__ uint32_t fmx_select32(uint32_t w, uint32_t r) {
  uint32_t pos = 0;
#pragma unroll
  for (int sh = 16; sh; sh >>= 1) {
    uint32_t c = __popc((w >> pos) & ((1u << sh) - 1u));
    if (r >= c) { r -= c; pos += sh; }
  }
  return pos;
}
// ---------------------------------------------------------------------------
// select on the wavelet levels (extract path: get_f / fl_map, fm_index.rs:97-120).
// Not on the count/locate hot path: a plain binary search over the record counters.
// ---------------------------------------------------------------------------
// counter of `code` at the start of record r (same units as fmx_level_rank's result)
__device__ __forceinline__ uint32_t fmx_level_counter(const FmxLevel &L, uint32_t r, uint32_t code) {
  FMX_CHECK(r < L.nrec);
  if (L.fmt == 3) return L.rec[(size_t)r * 8u + code].x;
  const uint4 p = L.rec[(size_t)r * 8u + (code >> 1)];
  return (code & 1u) ? p.y : p.x;
}
// position p of the entry with level code `code` whose fmx_level_rank(L, p, code) == target
__device__ __forceinline__ uint32_t fmx_level_select(const FmxLevel &L, uint32_t code,
                                                     uint32_t target, uint32_t g) {
  uint32_t lo = 0, hi = L.nrec - 1u;
  if (L.sel) {       // hints: the records of the entries just below and above the target
    const uint32_t base = L.selmeta[code], start = L.selmeta[16u + code];
    const uint32_t t = (target - base) / FMX_WSEL_STEP;
    const uint32_t a = L.sel[start + t], b = L.sel[start + t + 1u];
    if (a <= b && b < L.nrec) { lo = a; hi = b; }
  }
  while (lo < hi) {  // last record whose counter <= target
    const uint32_t mid = (lo + hi + 1u) >> 1;
    if (fmx_level_counter(L, mid, code) <= target) lo = mid; else hi = mid - 1u;
  }
  const uint32_t rem = target - fmx_level_counter(L, lo, code);  // rem-th match inside the record
  uint32_t m, per;
  if (L.fmt == 3) { m = fmx_piece_match<3>(L.rec[(size_t)lo * 8u + g], code); per = 32u; }
  else { m = fmx_piece_match<4>(L.rec[(size_t)lo * 8u + g], code); per = 16u; }
  const uint32_t mine = __popc(m);
  // exclusive prefix of the per-piece counts over the 8 lanes of the group
  uint32_t before = 0;
#pragma unroll
  for (uint32_t q = 0; q < FMX_GROUP; q++) {
    const uint32_t cq = fmx_group_sum(g == q ? mine : 0u);
    before += (q < g) ? cq : 0u;
  }
  const bool here = rem >= before && rem < before + mine;
  const uint32_t pos = here ? g * per + fmx_select32(m, rem - before) : 0u;
  return lo * (L.fmt == 3 ? 256u : 128u) + fmx_group_sum(pos);
}
// WaveletMatrix::select_u64_unchecked(k, c): position of the k-th (0-based) c
__device__ __forceinline__ uint32_t fmx_mwm_select(const FmxMwm &w, uint32_t c, uint32_t k,
                                                   uint32_t g) {
  // down: the start chain of c, in counter units of the last level
  uint32_t pos = 0, r = 0;
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxLevel &L = w.lv[l];
    r = fmx_level_rank(L, pos, (c >> L.shift) & L.mask, g);
    pos = r;
  }
  // up: invert the level mappings
  uint32_t target = r + k;
  for (uint32_t l = w.nlevels; l-- > 0;) {
    const FmxLevel &L = w.lv[l];
    target = fmx_level_select(L, (c >> L.shift) & L.mask, target, g);
  }
  return target;
}

// ===========================================================================
// RLFM (rlfmi.rs): bit vectors B / B' with rank1 / select1, and the LF formulas
// ===========================================================================
// record = 8 pieces of { ones before this piece (absolute), 96 payload bits }
__device__ __forceinline__ uint32_t fmx_lowmask(uint32_t k) {  // k in 0..32
  return k >= 32u ? 0xFFFFFFFFu : ((1u << k) - 1u);
}
__device__ __forceinline__ uint32_t fmx_div3(uint32_t x) {
  return (uint32_t)(((uint64_t)x * 0xAAAAAAABull) >> 33);
}
// rank1(i) (clamped like vers-vecs RsVec::rank1) and the bit B[i] (0 past the end)
__device__ __forceinline__ uint32_t fmx_bits_rank(const FmxBits &bv, uint32_t i, uint32_t g,
                                                  uint32_t &bit_i) {
  if (i > bv.len) i = bv.len;
  uint32_t rec = fmx_div3(i >> 8);            // i / 768
  uint32_t within = i - rec * FMX_BITS_PER_REC;
  uint32_t p = fmx_div3(within >> 5);         // within / 96
  uint32_t bit = within - p * FMX_BITS_PER_PIECE;
  FMX_CHECK(rec < bv.nrec);
  FMX_TOUCH_G0(g, &bv.rec[(size_t)rec * 8u]);
  uint4 pc = bv.rec[(size_t)rec * 8u + g];
  uint32_t c = __popc(pc.y & fmx_lowmask(bit < 32u ? bit : 32u));
  if (bit > 32u) c += __popc(pc.z & fmx_lowmask(bit - 32u < 32u ? bit - 32u : 32u));
  if (bit > 64u) c += __popc(pc.w & fmx_lowmask(bit - 64u));
  uint32_t word = bit < 32u ? pc.y : (bit < 64u ? pc.z : pc.w);
  uint32_t mine = (g == p) ? 1u : 0u;
  bit_i = fmx_group_sum(mine * ((word >> (bit & 31u)) & 1u));
  return fmx_group_sum(mine * (pc.x + c));
}
// rank1(i) as above, plus `next` = position of the first one at or after i when it lies in the
// record just loaded (0xFFFFFFFF otherwise).  That position IS select1(rank1(i)) -- the run start
// the RLFM formulas subtract (rlfmi.rs:132,141) -- so the select (hint word + record search +
// record: 2-3 dependent loads) is only needed when the run is longer than the rest of the record.
__device__ __forceinline__ uint32_t fmx_bits_rank_next(const FmxBits &bv, uint32_t i, uint32_t g,
                                                       uint32_t &bit_i, uint32_t &next) {
  if (i > bv.len) i = bv.len;
  const uint32_t rec = fmx_div3(i >> 8);            // i / 768
  const uint32_t within = i - rec * FMX_BITS_PER_REC;
  const uint32_t p = fmx_div3(within >> 5);         // within / 96
  const uint32_t bit = within - p * FMX_BITS_PER_PIECE;
  FMX_CHECK(rec < bv.nrec);
  FMX_TOUCH_G0(g, &bv.rec[(size_t)rec * 8u]);
  const uint4 pc = bv.rec[(size_t)rec * 8u + g];
  const uint32_t m0 = fmx_lowmask(bit < 32u ? bit : 32u);
  const uint32_t m1 = bit > 32u ? fmx_lowmask(bit - 32u < 32u ? bit - 32u : 32u) : 0u;
  const uint32_t m2 = bit > 64u ? fmx_lowmask(bit - 64u) : 0u;
  const uint32_t c = __popc(pc.y & m0) + __popc(pc.z & m1) + __popc(pc.w & m2);
  const uint32_t word = bit < 32u ? pc.y : (bit < 64u ? pc.z : pc.w);
  const uint32_t mine = (g == p) ? 1u : 0u;
  bit_i = fmx_group_sum(mine * ((word >> (bit & 31u)) & 1u));
  // first one at or after bit `bit` of piece p, or the first one of a later piece
  uint32_t y = pc.y, z = pc.z, w = pc.w;
  if (g == p) { y &= ~m0; z &= ~m1; w &= ~m2; }
  else if (g < p) { y = 0u; z = 0u; w = 0u; }
  uint32_t cand = 0xFFFFFFFFu;
  if (y) cand = (uint32_t)__builtin_ctz(y);
  else if (z) cand = 32u + (uint32_t)__builtin_ctz(z);
  else if (w) cand = 64u + (uint32_t)__builtin_ctz(w);
  if (cand != 0xFFFFFFFFu) cand += rec * FMX_BITS_PER_REC + g * FMX_BITS_PER_PIECE;
  next = fmx_group_min(cand);
  return fmx_group_sum(mine * (pc.x + c));
}
// rank1(i + 1) - 1 = the index of the last one at or before i (the run holding row i of an RLFM index), and `prev` =
// that one's position when it lies in the record just loaded (0xFFFFFFFF otherwise: the run began before the record)
__device__ __forceinline__ uint32_t fmx_bits_rank_prev(const FmxBits &bv, uint32_t i, uint32_t g, uint32_t &prev) {
  if (i >= bv.len) i = bv.len ? bv.len - 1u : 0u;
  const uint32_t rec = fmx_div3(i >> 8);            // i / 768
  const uint32_t within = i - rec * FMX_BITS_PER_REC;
  const uint32_t p = fmx_div3(within >> 5);         // within / 96
  const uint32_t b1 = within - p * FMX_BITS_PER_PIECE + 1u;   // bits [0, bit] of piece p
  FMX_CHECK(rec < bv.nrec);
  FMX_TOUCH_G0(g, &bv.rec[(size_t)rec * 8u]);
  const uint4 pc = bv.rec[(size_t)rec * 8u + g];
  const uint32_t m0 = fmx_lowmask(b1 < 32u ? b1 : 32u);
  const uint32_t m1 = b1 > 32u ? fmx_lowmask(b1 - 32u < 32u ? b1 - 32u : 32u) : 0u;
  const uint32_t m2 = b1 > 64u ? fmx_lowmask(b1 - 64u) : 0u;
  const uint32_t c = __popc(pc.y & m0) + __popc(pc.z & m1) + __popc(pc.w & m2);
  // last one at or before the row: in piece p below the mask, or the last one of an earlier piece
  uint32_t y = pc.y, z = pc.z, w = pc.w;
  if (g == p) { y &= m0; z &= m1; w &= m2; }
  else if (g > p) { y = 0u; z = 0u; w = 0u; }
  uint32_t cand = 0u;                               // position + 1; 0 = none (max over the group)
  if (w) cand = 96u - (uint32_t)__builtin_clz(w);
  else if (z) cand = 64u - (uint32_t)__builtin_clz(z);
  else if (y) cand = 32u - (uint32_t)__builtin_clz(y);
  if (cand) cand += rec * FMX_BITS_PER_REC + g * FMX_BITS_PER_PIECE;
  uint32_t mx = cand;
  mx = max(mx, fmx_dpp_xor1(mx));
  mx = max(mx, fmx_dpp_xor2(mx));
  mx = max(mx, fmx_dpp_half_mirror(mx));
  prev = mx ? mx - 1u : 0xFFFFFFFFu;
  return fmx_group_sum((g == p) ? pc.x + c : 0u) - 1u;
}
// dense-vector select block: position of one number (k & 63) inside the block's 96-bit window
__device__ __forceinline__ uint32_t fmx_dsel_pos(const uint4 blk, uint32_t k, uint32_t shift) {
  const uint32_t r = k & ((1u << shift) - 1u), c0 = __popc(blk.y), c1 = __popc(blk.z);
  uint32_t off;
  if (r < c0) off = fmx_select32(blk.y, r);
  else if (r < c0 + c1) off = 32u + fmx_select32(blk.z, r - c0);
  else off = 64u + fmx_select32(blk.w, r - c0 - c1);
  return blk.x + off;
}
// select1(k): position of the k-th one (0-based); len when k >= #ones (vers-vecs RsVec::select1)
__device__ __forceinline__ uint32_t fmx_bits_select(const FmxBits &bv, uint32_t k, uint32_t g) {
  if (k >= bv.ones) return bv.len;
  if (bv.pos) { FMX_TOUCH_G0N(g, &bv.pos[k]); return bv.pos[k]; }   // sparse vector: the positions are stored
  if (bv.dsel) {                              // dense vector: one 16-byte block answers it
    FMX_TOUCH_G0N(g, &bv.dsel[k >> bv.dsel_shift]);
    const uint4 blk = bv.dsel[k >> bv.dsel_shift];
    if (blk.x != 0xFFFFFFFFu) return fmx_dsel_pos(blk, k, bv.dsel_shift);
  }
  uint32_t h = k / FMX_SEL_STEP;
  FMX_CHECK(h + 1 < bv.nsel);
  FMX_TOUCH_G0N(g, &bv.sel[h]);
  uint32_t lo = bv.sel[h], hi = bv.sel[h + 1];
  FMX_CHECK(lo < bv.nrec && hi < bv.nrec);
  while (lo < hi) {  // group-uniform binary search over record bases
    uint32_t mid = (lo + hi + 1u) >> 1;
    FMX_TOUCH_G0(g, &bv.rec[(size_t)mid * 8u]);
    uint32_t x = bv.rec[(size_t)mid * 8u].x;
    if (x <= k) lo = mid; else hi = mid - 1u;
  }
  FMX_TOUCH_G0(g, &bv.rec[(size_t)lo * 8u]);
  uint4 pc = bv.rec[(size_t)lo * 8u + g];
  uint32_t p = fmx_group_sum(pc.x <= k ? 1u : 0u) - 1u;  // last piece whose base <= k
  uint32_t rem = k - pc.x;                                // meaningful on lane p only
  uint32_t c0 = __popc(pc.y), c1 = __popc(pc.z);
  uint32_t pos;
  if (rem < c0) pos = fmx_select32(pc.y, rem);
  else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
  else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
  pos = fmx_group_sum((g == p) ? pos : 0u);
  return lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
}

// two selects at once, on two vectors of the same index (B and B' always fall into the same
// density class) or twice on one: every stage -- positions / select blocks / hint words, record
// searches, records -- is issued for both before either is consumed, so their latencies overlap.
// SM fixes the structure at compile time (the hot kernels are instantiated per structure, so their
// loops hold no branch on it): 1 = stored positions, 2 = select blocks, 0 = hints + records,
// -1 = decide at run time.
template <int SM = -1>
__device__ __forceinline__ void fmx_bits_select_two(const FmxBits &A, uint32_t k0, const FmxBits &B,
                                                    uint32_t k1, uint32_t g, uint32_t &out0,
                                                    uint32_t &out1) {
  const bool v0 = k0 < A.ones, v1 = k1 < B.ones;
  const uint32_t q0 = v0 ? k0 : 0u, q1 = v1 ? k1 : 0u;
  if (SM == 1 || (SM < 0 && A.pos && B.pos)) {   // sparse vectors: the positions are stored
    FMX_TOUCH_G0N(g, &A.pos[q0]);
    FMX_TOUCH_G0N(g, &B.pos[q1]);
    const uint32_t a0 = A.pos[q0], a1 = B.pos[q1];
    out0 = v0 ? a0 : A.len;
    out1 = v1 ? a1 : B.len;
    return;
  }
  if (SM == 2 || (SM < 0 && A.dsel && B.dsel)) {   // dense vectors: one 16-byte block per select
    FMX_TOUCH_G0N(g, &A.dsel[q0 >> A.dsel_shift]);
    FMX_TOUCH_G0N(g, &B.dsel[q1 >> B.dsel_shift]);
    const uint4 b0 = A.dsel[q0 >> A.dsel_shift], b1 = B.dsel[q1 >> B.dsel_shift];
    if (b0.x != 0xFFFFFFFFu && b1.x != 0xFFFFFFFFu) {
      out0 = v0 ? fmx_dsel_pos(b0, q0, A.dsel_shift) : A.len;
      out1 = v1 ? fmx_dsel_pos(b1, q1, B.dsel_shift) : B.len;
      return;
    }
  }
  FMX_CHECK(q0 / FMX_SEL_STEP + 1 < A.nsel && q1 / FMX_SEL_STEP + 1 < B.nsel);
  FMX_TOUCH_G0N(g, &A.sel[q0 / FMX_SEL_STEP]);
  FMX_TOUCH_G0N(g, &B.sel[q1 / FMX_SEL_STEP]);
  uint32_t lo0 = A.sel[q0 / FMX_SEL_STEP], hi0 = A.sel[q0 / FMX_SEL_STEP + 1];
  uint32_t lo1 = B.sel[q1 / FMX_SEL_STEP], hi1 = B.sel[q1 / FMX_SEL_STEP + 1];
  FMX_CHECK(hi0 < A.nrec && hi1 < B.nrec);
  while (lo0 < hi0 || lo1 < hi1) {  // group-uniform binary searches over record bases
    const uint32_t m0 = (lo0 + hi0 + 1u) >> 1, m1 = (lo1 + hi1 + 1u) >> 1;
    if (lo0 < hi0) FMX_TOUCH_G0(g, &A.rec[(size_t)m0 * 8u]);
    if (lo1 < hi1) FMX_TOUCH_G0(g, &B.rec[(size_t)m1 * 8u]);
    const uint32_t x0 = A.rec[(size_t)m0 * 8u].x, x1 = B.rec[(size_t)m1 * 8u].x;
    if (lo0 < hi0) { if (x0 <= q0) lo0 = m0; else hi0 = m0 - 1u; }
    if (lo1 < hi1) { if (x1 <= q1) lo1 = m1; else hi1 = m1 - 1u; }
  }
  FMX_TOUCH_G0(g, &A.rec[(size_t)lo0 * 8u]);
  FMX_TOUCH_G0(g, &B.rec[(size_t)lo1 * 8u]);
  const uint4 a = A.rec[(size_t)lo0 * 8u + g];
  const uint4 b = B.rec[(size_t)lo1 * 8u + g];
  uint32_t res[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const uint4 pc = t ? b : a;
    const uint32_t k = t ? q1 : q0, lo = t ? lo1 : lo0;
    uint32_t p = fmx_group_sum(pc.x <= k ? 1u : 0u) - 1u;
    uint32_t rem = k - pc.x;
    uint32_t c0 = __popc(pc.y), c1 = __popc(pc.z);
    uint32_t pos;
    if (rem < c0) pos = fmx_select32(pc.y, rem);
    else if (rem < c0 + c1) pos = 32u + fmx_select32(pc.z, rem - c0);
    else pos = 64u + fmx_select32(pc.w, rem - c0 - c1);
    pos = fmx_group_sum((g == p) ? pos : 0u);
    res[t] = lo * FMX_BITS_PER_REC + p * FMX_BITS_PER_PIECE + pos;
  }
  out0 = v0 ? res[0] : A.len;
  out1 = v1 ? res[1] : B.len;
}
template <int SM = -1>
__device__ __forceinline__ void fmx_bits_select2(const FmxBits &bv, uint32_t k0, uint32_t k1,
                                                 uint32_t g, uint32_t &out0, uint32_t &out1) {
  fmx_bits_select_two<SM>(bv, k0, bv, k1, g, out0, out1);
}

// RLFMIndexBackend::lf_map2 (rlfmi.rs:135-143)
__device__ __forceinline__ uint32_t fmx_rlfm_lf_map2(const FmxDev &ix, uint32_t c, uint32_t i,
                                                     uint32_t g) {
  uint32_t bi, nx;
  uint32_t j = fmx_bits_rank_next(ix.b, i, g, bi, nx);    // b.rank1(i)
  uint32_t nr = ix.K[c] + fmx_mwm_rank(ix.bw, c, j, g);   // cs[c] + s.rank(j, c)
  uint32_t l;                                             // get_l(i) = s[b.rank1(i+1) - 1]
  (void)fmx_mwm_lf(ix.bw, j - 1u + bi, g, l);
  uint32_t r = fmx_bits_select(ix.bp, nr, g);             // bp.select1(cs[c] + nr)
  if (l == c) r = r + i - (nx != 0xFFFFFFFFu ? nx : fmx_bits_select(ix.b, j, g));  // + i - b.select1(j)
  return r;
}
// RLFMIndexBackend::get_l + lf_map (rlfmi.rs:122-133)
template <int NL = 0, int SM = -1>
__device__ __forceinline__ uint32_t fmx_rlfm_lf_map(const FmxDev &ix, uint32_t i, uint32_t g,
                                                    uint32_t &sym) {
  uint32_t bi;
  uint32_t j = fmx_bits_rank(ix.b, i, g, bi);
  (void)fmx_mwm_lf<NL>(ix.bw, j - 1u + bi, g, sym);
  uint32_t nr = ix.K[sym] + fmx_mwm_rank<NL>(ix.bw, sym, j, g);
  // (taking the run start from the B record already loaded, as fmx_rlfm_lf_map2_pair does, was
  // measured here too: the locate walk got 10 % slower -- its two selects overlap anyway and the
  // extra lane work sits on every step's dependent chain)
  uint32_t f, st;
  fmx_bits_select_two<SM>(ix.bp, nr, ix.b, j, g, f, st);   // bp.select1(nr), b.select1(j) -- overlapped
  return f + i - st;
}

// both interval ends of one backward-search step on the RLFM index, staged so that the
// independent probes of the two ends overlap:  B ranks -> S ranks -> B' selects.
// get_l(i) == c (rlfmi.rs:138) is decided WITHOUT the wavelet access: with lo = b.rank1(i+1)-1
// (the run holding row i) it is  s.rank(lo+1, c) - s.rank(lo, c) == 1, and b.rank1(i) is lo or
// lo+1, so one rank chain over the adjacent positions {lo, lo+1} yields nr and the comparison
// from the same cache lines.
template <int NL = 0, int SM = -1>
__device__ __forceinline__ void fmx_rlfm_lf_map2_pair(const FmxDev &ix, uint32_t c, uint32_t &s,
                                                      uint32_t &e, uint32_t g) {
  const uint32_t kc = ix.K[c];
  uint32_t bs, be, nxs, nxe;
  const uint32_t js = fmx_bits_rank_next(ix.b, s, g, bs, nxs);  // b.rank1(i)        rlfmi.rs:136
  const uint32_t je = fmx_bits_rank_next(ix.b, e, g, be, nxe);
  const uint32_t los = js - 1u + bs, loe = je - 1u + be;  // b.rank1(i+1) - 1   rlfmi.rs:124
  uint32_t pos[4] = {los, los + 1u, loe, loe + 1u};
  uint32_t r[4];
  fmx_mwm_rankN<4, NL>(ix.bw, c, pos, g, r);
  const uint32_t nrs = kc + (bs ? r[0] : r[1]);       // cs[c] + s.rank(j, c)   rlfmi.rs:137,139
  const uint32_t nre = kc + (be ? r[2] : r[3]);
  const bool eqs = (r[1] - r[0]) == 1u;               // get_l(i) == c          rlfmi.rs:138
  const bool eqe = (r[3] - r[2]) == 1u;
  uint32_t ns, ne;
  fmx_bits_select2<SM>(ix.bp, nrs, nre, g, ns, ne);   // bp.select1(cs[c] + nr)
  // + i - b.select1(j)   rlfmi.rs:141; the run start usually sits in the record already loaded
  if (eqs) ns = ns + s - (nxs != 0xFFFFFFFFu ? nxs : fmx_bits_select(ix.b, js, g));
  if (eqe) ne = ne + e - (nxe != 0xFFFFFFFFu ? nxe : fmx_bits_select(ix.b, je, g));
  s = ns;
  e = ne;
}

// ---- text-order sampling (FmxDev::phase) ------------------------------------------------------
__device__ __forceinline__ uint32_t fmx_div15(uint32_t x) {
  return (uint32_t(((uint64_t)x * 0x88888889ull) >> 35));
}
// piece holding `row` and the row's index inside it; rows per piece = 3 * floor(32 / level)
__device__ __forceinline__ uint32_t fmx_phase_piece(uint32_t row, uint32_t level, uint32_t &t) {
  uint32_t p;
  if (level == 1) { p = fmx_div3(row >> 5); t = row - p * 96u; }
  else if (level == 2) { p = fmx_div3(row >> 4); t = row - p * 48u; }
  else if (level == 3) { p = fmx_div15(row >> 1); t = row - p * 30u; }
  else { p = fmx_div3(row >> 3); t = row - p * 24u; }
  return p;
}
// fields of one word whose phase is 0, as a mask over the fields' lowest bits
__device__ __forceinline__ uint32_t fmx_phase_zero_fields(uint32_t w, uint32_t level) {
  if (level == 1) return ~w;
  if (level == 2) return ~(w | (w >> 1)) & 0x55555555u;
  if (level == 3) return ~(w | (w >> 1) | (w >> 2)) & 0x09249249u;
  return ~(w | (w >> 1) | (w >> 2) | (w >> 3)) & 0x11111111u;
}
// phase of the piece's t-th row, and the number of phase-0 rows before it in the whole index
__device__ __forceinline__ uint32_t fmx_phase_decode(const uint4 pc, uint32_t t, uint32_t level, uint32_t &rank0) {
  // the three words are read into registers BEFORE the data-dependent choice: a choice between
  // members of a by-reference uint4 becomes a dynamically indexed private array (scratch memory)
  const uint32_t py = pc.y, pz = pc.z, pw = pc.w;
  const uint32_t fpw = level == 1 ? 32u : (level == 2 ? 16u : (level == 3 ? 10u : 8u));
  const uint32_t wi = level == 3 ? (uint32_t)(t >= 10u) + (uint32_t)(t >= 20u)
                                 : (level == 1 ? t >> 5 : (level == 2 ? t >> 4 : t >> 3));
  const uint32_t k = t - wi * fpw;
  const uint32_t w = wi == 0 ? py : (wi == 1 ? pz : pw);
  const uint32_t z0 = fmx_phase_zero_fields(py, level), z1 = fmx_phase_zero_fields(pz, level);
  const uint32_t zc = fmx_phase_zero_fields(w, level);
  rank0 = pc.x + (wi > 0 ? __popc(z0) : 0u) + (wi > 1 ? __popc(z1) : 0u) + __popc(zc & fmx_lowmask(k * level));
  return (w >> (k * level)) & ((1u << level) - 1u);
}

// ---- walk records (FmxDev::walk, fmx_internal.h) ----------------------------------------------
// record holding `row` and the row's index inside it (112 rows per record): row / 112 = (row >> 4) / 7, and
// mulhi(x, ceil(2^32 / 7)) is x / 7 exactly for x < 2^28
__device__ __forceinline__ uint32_t fmx_walk_record(uint32_t row, uint32_t &off) {
  const uint32_t rec = __umulhi(row >> 4, 613566757u);
  off = row - rec * FMX_WALK_ROWS;
  return rec;
}
// One LF step of a text-order walk on the walk record holding `row` (off = the row's index in it); p = piece g of that
// record.  Every lane of the group gets the row's symbol and phase SA[row] mod 2^level, the return value r_lf and si:
//     lf_map(row) = r_lf  [+ the superblock's base of counter sym - 1 on a wide index]     (fm_index.rs:86-91)
//     ph == 0: si [+ base of counter 5] = index into samples[] of the row itself (the phase-0 rows before it);
//     ph == 1: si [+ base of counter 5 + sym] = index of the sample of the row LF(row) -- rank1[L[row]] + the phase-1
//              rows with the same symbol before it (fmx_internal.h);   else si is unspecified.
// On the 32-bit engine the record counters are absolute (cs[] folded in) and the bases are zero.
// Three group sums: the row's code and phase (lane that holds it), the rank, the sample index.
__device__ __forceinline__ uint32_t fmx_walk_step_rel(const uint4 &p, uint32_t off, uint32_t g, uint32_t &sym, uint32_t &ph,
                                                      uint32_t &si) {
  const uint32_t bit = off & 15u;
  int nb = (int)off - (int)(g * 16u);
  nb = nb < 0 ? 0 : (nb > 16 ? 16 : nb);
  const uint32_t low = g < 7u ? (1u << nb) - 1u : 0u;                        // piece 7 holds counters, not rows
  uint32_t v = 0;
  if (g == (off >> 4))
    v = __builtin_amdgcn_ubfe(p.y, bit, 1u) | (__builtin_amdgcn_ubfe(p.y, bit + 16u, 1u) << 1) |
        (__builtin_amdgcn_ubfe(p.z, bit, 1u) << 2) | (__builtin_amdgcn_ubfe(p.z, bit + 16u, 1u) << 3) |
        (__builtin_amdgcn_ubfe(p.w, bit, 1u) << 4) | (__builtin_amdgcn_ubfe(p.w, bit + 16u, 1u) << 5);
  v = fmx_group_sum(v);
  sym = v & 7u;
  ph = v >> 3;
  const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)sym, 0u, 1u), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)sym, 1u, 1u),
                 m2 = (uint32_t)__builtin_amdgcn_sbfe((int)sym, 2u, 1u);
  const uint32_t match = ~((p.y ^ m0) | ((p.y >> 16) ^ m1) | (p.z ^ m2)) & low;   // rows before `row` with its symbol
  const uint32_t q0 = p.z >> 16, q1 = p.w, q2 = p.w >> 16;                         // phase planes (high halves: masked by `low`)
  // sample index: phase-0 rows before the row (ph == 0), or phase-1 rows with the row's symbol before it (+ rank1[sym])
  const uint32_t sel = ph == 0u ? ~(q0 | q1 | q2) & low : (q0 & ~(q1 | q2)) & match;
  // the counter word of this lane that the index needs: rank0 in lane 5; rank1[1] in lane 6, rank1[2..5] in lane 7
  const uint32_t c7 = sym == 2u ? p.x : (sym == 3u ? p.y : (sym == 4u ? p.z : p.w));
  const uint32_t cw = ph == 0u ? (g == 5u ? p.x : 0u)
                               : (sym == 1u ? (g == 6u ? p.x : 0u) : (g == 7u ? c7 : 0u));
  si = fmx_group_sum((uint32_t)__popc(sel) + cw);
  return fmx_group_sum((uint32_t)__popc(match) + (g + 1u == sym ? p.x : 0u));     // lanes 0..4: lf_map2(g + 1, .)
}
__device__ __forceinline__ uint32_t fmx_walk_step(const uint4 &p, uint32_t off, uint32_t g, uint32_t &ph, uint32_t &si) {
  uint32_t sym;
  return fmx_walk_step_rel(p, off, g, sym, ph, si);
}
// record holding a 64-bit row (wide engine): row / 112 = (row >> 4) / 7 with mulhi(x, ceil(2^64 / 7)), exact for x < 2^61
__device__ __forceinline__ uint64_t fmx_walk_record64(uint64_t row, uint32_t &off) {
  const uint64_t rec = __umul64hi(row >> 4, 0x2492492492492493ull);
  off = (uint32_t)(row - rec * FMX_WALK_ROWS);
  return rec;
}

// greatest c with cs[c] <= v  (get_f's binary search, fm_index.rs:97-112)
__device__ __forceinline__ uint32_t fmx_cs_upper(const uint32_t *cs, uint32_t max_character,
                                                 uint32_t v) {
  uint32_t s = 0, e = max_character + 1u;
  while (e - s > 1u) {
    const uint32_t m = s + (e - s) / 2u;
    if (cs[m] <= v) s = m; else e = m;
  }
  return s;
}
// get_f + fl_map (fm_index.rs:97-120, rlfmi.rs:145-169); returns fl_map(i), sets sym = get_f(i)
template <int KIND>
__device__ __forceinline__ uint32_t fmx_fl_map_any(const FmxDev &ix, uint32_t i, uint32_t g,
                                                   uint32_t &sym) {
  if (KIND == FMX_KIND_FM || KIND == FMX_KIND_MULTI) {
    sym = fmx_cs_upper(ix.cs, ix.max_character, i);
    if (KIND == FMX_KIND_MULTI && sym == 0u) return 0xFFFFFFFFu;   // fl_map: None (multi_pieces.rs:176-178)
    return fmx_mwm_select(ix.bw, sym, i - ix.cs[sym], g);   // bw.select(i - cs[c], c)
  } else {
    uint32_t bit;
    const uint32_t j = fmx_bits_rank(ix.bp, i + 1u, g, bit) - 1u;   // bp.rank1(i+1) - 1
    sym = fmx_cs_upper(ix.cs, ix.max_character, j);
    const uint32_t p = fmx_bits_select(ix.bp, j, g);                // bp.select1(j)
    const uint32_t m = fmx_mwm_select(ix.bw, sym, j - ix.cs[sym], g);  // s.select(j - cs[c], c)
    const uint32_t nn = fmx_bits_select(ix.b, m, g);                // b.select1(m)
    return nn + i - p;
  }
}

// kind-dispatching forms used by the kernels
// FMIndexMultiPiecesBackend::lf_map2 for c == 0 (multi_pieces.rs:147-153): the end markers are
// ordered by piece, the LAST one (row sa_idx_first_text) maps to row 0
__device__ __forceinline__ uint32_t fmx_multi_zero(const FmxDev &ix, uint32_t i, uint32_t rank0) {
  return i < ix.first_row ? rank0 + 1u : (i == ix.first_row ? 0u : rank0);
}
template <int KIND, int NL = 0, int SM = -1>
__device__ __forceinline__ void fmx_lf_map2_pair(const FmxDev &ix, uint32_t c, uint32_t &s,
                                                 uint32_t &e, uint32_t g) {
  if (KIND == FMX_KIND_FM || KIND == FMX_KIND_MULTI) {
    uint32_t rs, re;
    const uint32_t kc = ix.K[c];  // issued ahead of the record loads
    fmx_mwm_rank2<NL>(ix.bw, c, s, e, g, rs, re);
    if (KIND == FMX_KIND_MULTI && c == 0u) {
      s = fmx_multi_zero(ix, s, kc + rs);
      e = fmx_multi_zero(ix, e, kc + re);
      return;
    }
    s = kc + rs;  // fm_index.rs:93-95
    e = kc + re;
  } else {
    fmx_rlfm_lf_map2_pair<NL, SM>(ix, c, s, e, g);
  }
}
// lf_map(i) for a walk that has no use for the symbol (get_sa: fm_index.rs:134-137, rlfmi.rs:183-186): an RLFM index
// with the run table (FmxDev::lfrun) answers with the B record of the row + one table entry -- lf_map(i) = lfrun[run] +
// (i - start of the run) -- and only a run that began before the record costs a select on B
template <int KIND, int NL = 0, int SM = -1>
__device__ __forceinline__ uint32_t fmx_lf_map_any(const FmxDev &ix, uint32_t i, uint32_t g, uint32_t &sym);
template <int KIND, int NL = 0, int SM = -1>
__device__ __forceinline__ uint32_t fmx_lf_step_any(const FmxDev &ix, uint32_t i, uint32_t g) {
  if (KIND == FMX_KIND_RLFM && ix.lfrun) {
    uint32_t st;
    const uint32_t lo = fmx_bits_rank_prev(ix.b, i, g, st);
    FMX_CHECK(lo < ix.b.ones);
    FMX_TOUCH_G0N(g, &ix.lfrun[lo]);
    const uint32_t f = ix.lfrun[lo];
    if (st == 0xFFFFFFFFu) st = fmx_bits_select(ix.b, lo, g);   // group-uniform
    return f + i - st;
  }
  uint32_t sym;
  return fmx_lf_map_any<KIND, NL, SM>(ix, i, g, sym);
}
template <int KIND, int NL, int SM>
__device__ __forceinline__ uint32_t fmx_lf_map_any(const FmxDev &ix, uint32_t i, uint32_t g,
                                                   uint32_t &sym) {
  if (KIND == FMX_KIND_FM || KIND == FMX_KIND_MULTI) {
    uint32_t r = fmx_mwm_lf<NL>(ix.bw, i, g, sym);
    r += ix.K[sym];  // fm_index.rs:86-91
    if (KIND == FMX_KIND_MULTI && sym == 0u) r = fmx_multi_zero(ix, i, r);  // multi_pieces.rs:131-137
    return r;
  } else {
    return fmx_rlfm_lf_map<NL, SM>(ix, i, g, sym);
  }
}
