// fmx_query.hip -- the hot path: batched backward search (count) and the locate walk.
//
// Replaces, for a batch:   SearchWrapper::search      (wrapper.rs:103-124)
//                          FMIndexBackend::lf_map2     (fm_index.rs:93-95)
//                          MatchIteratorWrapper::next  (wrapper.rs:203-217)
//                          FMIndexBackend::get_sa      (fm_index.rs:127-140)
//                          SOSampledSuffixArray::get   (sample.rs:46-60)
//
// Shape: persistent 8-lane groups.  A group owns one pattern (count) or one hit
// (locate) at a time and runs a small state machine, so a group that finishes
// early (the `s == e` break of wrapper.rs:111-113, or a short locate walk) picks
// up the next unit while its wave-mates keep stepping -- no lane idles on
// divergence except in the tail.  Integer / popcount work only; HBM-bound.
#include "fmx_device.h"

#define FMX_BLOCK 256
#define FMX_MAX_BLOCKS 2048  // 256 CUs x 8 resident 256-thread blocks

static inline unsigned fmx_grid_for_groups(uint64_t units) {
  uint64_t blocks = (units * FMX_GROUP + FMX_BLOCK - 1) / FMX_BLOCK;
  if (blocks < 1) blocks = 1;
  if (blocks > FMX_MAX_BLOCKS) blocks = FMX_MAX_BLOCKS;
  return (unsigned)blocks;
}

// ---------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_count_kernel(
    FmxDev ix, const uint8_t *__restrict__ pat, const uint64_t *__restrict__ off, uint64_t npat,
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
  uint32_t nsteps = 0;

  while (active) {
    if (fresh) {
      pbeg = off[k];
      j = (uint32_t)(off[k + 1] - pbeg);
      if (s0e0) {            // Search::search on an existing Search (wrapper.rs:105-106)
        s = (uint32_t)s0e0[2 * k];
        e = (uint32_t)s0e0[2 * k + 1];
      } else {               // SearchIndexWrapper::search: (0, len)   (wrapper.rs:41)
        s = 0;
        e = ix.n;
      }
      fresh = false;
    }
    bool done = (j == 0);
    if (!done) {
      uint32_t c = pat[pbeg + j - 1];                 // pattern.iter().rev()  wrapper.rs:108
      j--;
      if (c > ix.max_character) {                      // reference: panic on cs[c]
        if (g == 0) atomicOr(ix.status, 1u << FMX_ERR_SYMBOL_RANGE);
        s = 0; e = 0; done = true;
      } else {
        fmx_lf_map2_pair<KIND>(ix, c, s, e, g);        // wrapper.rs:109-110
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
    // one add per group leader; wave-reduced by the compiler's atomic coalescing
    if (g == 0 && nsteps) atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
  }
}

// ---------------------------------------------------------------------------
// locate
// ---------------------------------------------------------------------------
// exclusive offsets -> rows: out_pos[off[k] + j] = s[k] + j   (wrapper.rs:203-217: i = s..e-1
// ascending).  One 8-lane group per pattern, lanes stride over its rows.
__global__ __launch_bounds__(FMX_BLOCK) void fmx_expand_kernel(
    const uint64_t *__restrict__ s, const uint64_t *__restrict__ e,
    const uint64_t *__restrict__ off, uint64_t npat, uint64_t *__restrict__ out_pos) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t k = gid; k < npat; k += ngroups) {
    uint64_t a = s[k], b = e[k], o = off[k];
    uint64_t cnt = b > a ? b - a : 0;
    for (uint64_t t = g; t < cnt; t += FMX_GROUP) out_pos[o + t] = a + t;
  }
}

// in place: out_pos[h] holds SA row i on entry and get_sa(i) on exit.
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_locate_kernel(FmxDev ix, uint64_t total,
                                                                uint64_t *__restrict__ out_pos,
                                                                uint64_t *__restrict__ steps_out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  const uint32_t lmask = (1u << ix.sa_level) - 1u;

  uint64_t h = gid;
  bool active = h < total;
  bool fresh = true;
  uint32_t row = 0, steps = 0, nsteps = 0;
  while (active) {
    if (fresh) {
      row = (uint32_t)out_pos[h];
      steps = 0;
      fresh = false;
    }
    if ((row & lmask) == 0) {
      // sample.rs:46-60 Some(sa): fm_index.rs:131-133  (sa + steps) % len
      uint64_t v = (uint64_t)ix.samples[row >> ix.sa_level] + steps;
      if (v >= ix.n) v -= ix.n;  // steps < n, sa < n
      if (g == 0) out_pos[h] = v;
      h += ngroups;
      active = h < total;
      fresh = true;
    } else {
      // None: i = lf_map(i); steps += 1      fm_index.rs:134-137
      uint32_t sym;
      row = fmx_lf_map_any<KIND>(ix, row, g, sym);
      steps++;
      nsteps++;
    }
  }
  if (steps_out && g == 0 && nsteps)
    atomicAdd((unsigned long long *)steps_out, (unsigned long long)nsteps);
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
    } else {  // get_sa (fm_index.rs:127-140)
      uint32_t row = (uint32_t)i64, steps = 0;
      const uint32_t lmask = (1u << ix.sa_level) - 1u;
      while ((row & lmask) != 0) {
        uint32_t sym;
        row = fmx_lf_map_any<KIND>(ix, row, g, sym);
        steps++;
      }
      uint64_t v = (uint64_t)ix.samples[row >> ix.sa_level] + steps;
      if (v >= ix.n) v -= ix.n;
      res = v;
    }
    if (g == 0) out[q] = res;
  }
}

// export: L column of rows [0, n) as one byte per row (get_l, fm_index.rs:82-84)
template <int KIND>
__global__ __launch_bounds__(FMX_BLOCK) void fmx_export_l_kernel(FmxDev ix, uint8_t *__restrict__ out) {
  const uint32_t g = threadIdx.x & (FMX_GROUP - 1);
  uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / FMX_GROUP;
  const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / FMX_GROUP;
  for (uint64_t q = gid; q < ix.n; q += ngroups) {
    uint32_t sym;
    (void)fmx_lf_map_any<KIND>(ix, (uint32_t)q, g, sym);
    if (g == 0) out[q] = (uint8_t)sym;
  }
}
int fmx_launch_export_l(const fmx_index *idx, uint8_t *d_out, hipStream_t st) {
  if (idx->n == 0) return FMX_OK;
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_export_l_kernel<FMX_KIND_FM>, dim3(fmx_grid_for_groups(idx->n)),
                       dim3(FMX_BLOCK), 0, st, idx->dev, d_out);
  else
    hipLaunchKernelGGL(fmx_export_l_kernel<FMX_KIND_RLFM>, dim3(fmx_grid_for_groups(idx->n)),
                       dim3(FMX_BLOCK), 0, st, idx->dev, d_out);
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
  if (idx->timing) {
    fmx_index *m = const_cast<fmx_index *>(idx);
    (void)hipMemsetAsync(m->d_steps, 0, sizeof(uint64_t), st);
    (void)hipEventRecord(m->ev0, st);
  }
}
static void fmx_time_end(const fmx_index *idx, hipStream_t st) {
  if (idx->timing) {
    fmx_index *m = const_cast<fmx_index *>(idx);
    (void)hipEventRecord(m->ev1, st);
    m->ev_valid = 1;
  }
}

int fmx_launch_count(const fmx_index *idx, const uint8_t *d_pat, const uint64_t *d_off,
                     uint64_t npat, const uint64_t *d_s0e0, uint64_t *d_s, uint64_t *d_e,
                     uint64_t *d_cnt, hipStream_t st) {
  if (npat == 0) return FMX_OK;
  unsigned grid = fmx_grid_for_groups(npat);
  fmx_time_begin(idx, st);
  uint64_t *steps = idx->timing ? idx->d_steps : nullptr;
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_count_kernel<FMX_KIND_FM>, dim3(grid), dim3(FMX_BLOCK), 0, st, idx->dev,
                       d_pat, d_off, npat, d_s0e0, d_s, d_e, d_cnt, steps);
  else
    hipLaunchKernelGGL(fmx_count_kernel<FMX_KIND_RLFM>, dim3(grid), dim3(FMX_BLOCK), 0, st, idx->dev,
                       d_pat, d_off, npat, d_s0e0, d_s, d_e, d_cnt, steps);
  fmx_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmx_launch_offsets(const uint64_t *d_s, const uint64_t *d_e, uint64_t npat, uint64_t *d_off,
                       hipStream_t st) {
  uint64_t ntiles = (npat + FMX_SCAN_TILE - 1) / FMX_SCAN_TILE;
  if (ntiles == 0) ntiles = 1;
  uint64_t *tile = nullptr;
  FMX_HIP(hipMallocAsync((void **)&tile, (ntiles + 1) * sizeof(uint64_t), st));
  hipLaunchKernelGGL(fmx_tile_sums_kernel, dim3((unsigned)ntiles), dim3(FMX_BLOCK), 0, st, d_s, d_e,
                     npat, tile);
  hipLaunchKernelGGL(fmx_scan_tiles_kernel, dim3(1), dim3(FMX_BLOCK), 0, st, tile, ntiles);
  hipLaunchKernelGGL(fmx_tile_scan_kernel, dim3((unsigned)ntiles), dim3(FMX_BLOCK), 0, st, d_s, d_e,
                     npat, tile, ntiles, d_off);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipFreeAsync(tile, st));
  return FMX_OK;
}

int fmx_launch_locate(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                      uint64_t npat, const uint64_t *d_off, uint64_t total, uint64_t *d_pos,
                      hipStream_t st) {
  if (npat == 0 || total == 0) return FMX_OK;
  hipLaunchKernelGGL(fmx_expand_kernel, dim3(fmx_grid_for_groups(npat)), dim3(FMX_BLOCK), 0, st,
                     d_s, d_e, d_off, npat, d_pos);
  fmx_time_begin(idx, st);
  uint64_t *steps = idx->timing ? idx->d_steps : nullptr;
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_locate_kernel<FMX_KIND_FM>, dim3(fmx_grid_for_groups(total)),
                       dim3(FMX_BLOCK), 0, st, idx->dev, total, d_pos, steps);
  else
    hipLaunchKernelGGL(fmx_locate_kernel<FMX_KIND_RLFM>, dim3(fmx_grid_for_groups(total)),
                       dim3(FMX_BLOCK), 0, st, idx->dev, total, d_pos, steps);
  fmx_time_end(idx, st);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int fmx_launch_scalar(const fmx_index *idx, int op, const uint64_t *d_c, const uint64_t *d_i,
                      uint64_t k, uint64_t *d_out, hipStream_t st) {
  if (k == 0) return FMX_OK;
  if (idx->kind == FMX_KIND_FM)
    hipLaunchKernelGGL(fmx_scalar_kernel<FMX_KIND_FM>, dim3(fmx_grid_for_groups(k)), dim3(FMX_BLOCK),
                       0, st, idx->dev, op, d_c, d_i, k, d_out);
  else
    hipLaunchKernelGGL(fmx_scalar_kernel<FMX_KIND_RLFM>, dim3(fmx_grid_for_groups(k)),
                       dim3(FMX_BLOCK), 0, st, idx->dev, op, d_c, d_i, k, d_out);
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
