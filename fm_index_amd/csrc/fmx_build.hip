// fmx_build.hip -- index construction, entirely on the GPU.
//
// Replaces FMIndexBackend::new (fm_index.rs:25-58) / RLFMIndexBackend::new
// (rlfmi.rs:30-96) for the query path's inputs:
//   text validation        sais.rs:115-139
//   C array                sais.rs:9-32  (count_chars + get_bucket_start_pos)
//   suffix array           sais.rs:115-144 (the reference uses SA-IS on the CPU; the array
//                          is uniquely defined -- sais.rs:546-557 -- so it is built here
//                          by GPU prefix doubling over rocPRIM radix sorts instead)
//   BWT                    fm_index.rs:44-58
//   SA samples             suffix_array/sample.rs:21-44
//   rank structure         vers-vecs WaveletMatrix::from_slice -> 128-B multi-ary
//                          wavelet-matrix records (fmx_internal.h)
#include <cstring>
#include <iterator>
#include <rocprim/rocprim.hpp>
#include <vector>
#include <mutex>
#include <chrono>
#include "fmx_device.h"

// FMX_BUILD_TRACE=1 (elapsed ms after each build phase on stderr) is read by the measurement builds only: the shipped
// library reads no environment variable at all
static inline bool fmx_build_trace() {
#ifdef FMX_MEASURE
  return getenv("FMX_BUILD_TRACE") != nullptr;
#else
  return false;
#endif
}

#define BLK 256

namespace {

// Scratch of SMALL builds: every temporary of a text of up to kArenaMaxN symbols is carved from one leased buffer
// instead of ~30 hipMalloc / hipFree pairs (a hipFree waits for the device; at n = 1000 they were most of the build).
// The buffers belong to a process-wide pool per device, behind a mutex: a build leases one for its duration -- sized
// from n (4 MiB up to 8192 symbols, else 48 MiB: 384 bytes per symbol) -- and hands it back; at most kArenaKeep idle
// buffers per device and size class are retained, whatever the number of threads that build (a thread-pool service
// used to pin 48 MiB per worker thread and device until the thread exited), fmx_release_scratch() frees the idle ones,
// and nothing is freed at process exit (the pool is never destroyed: no hipFree can race the runtime's teardown).
// A failed allocation is not retried for the next kArenaBackoff small builds on that device.
const uint64_t kArenaMaxN = 1ull << 17;
const size_t kArenaBytes = 48u << 20, kArenaSmallBytes = 4u << 20;
const uint64_t kArenaSmallN = 8192;
const int kArenaKeep = 2, kArenaBackoff = 64, kArenaDevices = 16;
struct ArenaPool {
  std::mutex mu;
  std::vector<uint8_t *> idle[kArenaDevices][2];
  int backoff[kArenaDevices] = {};
};
inline ArenaPool &arena_pool() {
  static ArenaPool *pool = new ArenaPool;     // leaked on purpose
  return *pool;
}
struct ArenaLease {   // RAII: the buffer goes back to the pool (or is freed when the pool is full) when the build returns
  uint8_t *p = nullptr;
  size_t bytes = 0;
  int device = -1, cls = 0;
  void take(int dev, uint64_t n) {
    if (dev < 0 || dev >= kArenaDevices) return;
    device = dev;
    cls = n <= kArenaSmallN ? 0 : 1;
    bytes = cls ? kArenaBytes : kArenaSmallBytes;
    ArenaPool &ap = arena_pool();
    {
      std::lock_guard<std::mutex> lk(ap.mu);
      if (!ap.idle[dev][cls].empty()) {
        p = ap.idle[dev][cls].back();
        ap.idle[dev][cls].pop_back();
        return;
      }
      if (ap.backoff[dev] > 0) { ap.backoff[dev]--; return; }
    }
    if (hipMalloc((void **)&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      std::lock_guard<std::mutex> lk(ap.mu);
      ap.backoff[dev] = kArenaBackoff;
    }
  }
  ~ArenaLease() {
    if (!p) return;
    ArenaPool &ap = arena_pool();
    {
      std::lock_guard<std::mutex> lk(ap.mu);
      if ((int)ap.idle[device][cls].size() < kArenaKeep) {
        ap.idle[device][cls].push_back(p);
        return;
      }
    }
    (void)hipFree(p);      // the calling thread's current device is the build's device
  }
};

// Scratch of LARGE builds.  On this runtime hipMalloc / hipFree cost 0.2-0.5 ms whatever the size only while the
// process is handed device memory it has not used before; once it has cycled through about the device's memory --
// which a few 2^30-symbol builds (24-30 GB of scratch each) do -- every further hipMalloc pays ~30 ms per GiB
// (benchmarks/gpu/alloc_probe.hip, r04_wide_dirty.sh: the second n = 2^32 build of a process spends 4.9 s in its first
// allocations, the first one 0.7 ms).  So temporaries of at least kScratchMin bytes go back to a process-wide cache per
// device instead of to the driver, and the next build takes them from there (smallest cached block that fits with at
// most a quarter of it wasted): after its first large build a process allocates little that is new.  The cache holds
// at most kScratchKeep bytes or an eighth of the device, whichever is less; what does not fit is freed, a failing
// hipMalloc empties the cache and tries again, and fmx_release_scratch() gives everything back.
const size_t kScratchMin = 1u << 20;
const size_t kScratchKeep = 32ull << 30;
struct ScratchCache {
  struct Blk { void *p; size_t bytes; };
  std::mutex mu;
  std::vector<Blk> idle[kArenaDevices];
  size_t held[kArenaDevices] = {}, cap[kArenaDevices] = {}, total[kArenaDevices] = {};
};
inline ScratchCache &scratch_cache() {
  static ScratchCache *c = new ScratchCache;      // leaked on purpose: nothing is freed at process exit
  return *c;
}
inline void scratch_drop(int dev, std::vector<ScratchCache::Blk> &out) {   // caller holds the lock
  ScratchCache &sc = scratch_cache();
  out.insert(out.end(), sc.idle[dev].begin(), sc.idle[dev].end());
  sc.idle[dev].clear();
  sc.held[dev] = 0;
}
inline hipError_t scratch_get(int dev, size_t bytes, void **out, size_t *got) {
  ScratchCache &sc = scratch_cache();
  const bool cached = dev >= 0 && dev < kArenaDevices && bytes >= kScratchMin;
  if (cached) {
    std::lock_guard<std::mutex> lk(sc.mu);
    auto &v = sc.idle[dev];
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].bytes >= bytes && v[i].bytes - bytes <= bytes / 4 && (best == v.size() || v[i].bytes < v[best].bytes)) best = i;
    if (best != v.size()) {
      *out = v[best].p;
      *got = v[best].bytes;
      sc.held[dev] -= v[best].bytes;
      v.erase(v.begin() + best);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess && cached) {                 // out of memory: give the cached blocks back and try once more
    (void)hipGetLastError();
    std::vector<ScratchCache::Blk> drop;
    { std::lock_guard<std::mutex> lk(sc.mu); scratch_drop(dev, drop); }
    for (auto &b : drop) (void)hipFree(b.p);
    e = hipMalloc(out, bytes);
  }
  *got = bytes;
  return e;
}
// drop the idle scratch of the CURRENT device (an allocation failed); returns the bytes given back to the driver
inline size_t scratch_drop_current() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kArenaDevices) return 0;
  ScratchCache &sc = scratch_cache();
  std::vector<ScratchCache::Blk> drop;
  { std::lock_guard<std::mutex> lk(sc.mu); scratch_drop(dev, drop); }
  size_t bytes = 0;
  for (auto &b : drop) { bytes += b.bytes; (void)hipFree(b.p); }
  return bytes;
}
// keep_all (FMX_FLAG_KEEP_SCRATCH): the caller asked for this build's temporaries to stay whatever their size -- up to
// three quarters of the device
inline void scratch_put(int dev, void *p, size_t bytes, bool keep_all = false) {
  ScratchCache &sc = scratch_cache();
  if (dev >= 0 && dev < kArenaDevices && bytes >= kScratchMin) {
    std::lock_guard<std::mutex> lk(sc.mu);
    if (!sc.cap[dev]) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
      sc.total[dev] = total_b;
      sc.cap[dev] = total_b && total_b / 8 < kScratchKeep ? total_b / 8 : kScratchKeep;
    }
    const size_t cap = keep_all && sc.total[dev] ? sc.total[dev] / 4 * 3 : sc.cap[dev];
    if (sc.held[dev] + bytes <= cap) {
      sc.idle[dev].push_back({p, bytes});
      sc.held[dev] += bytes;
      return;
    }
  }
  (void)hipFree(p);
}

struct DevPool {  // temporaries given back when the builder returns (large ones to the scratch cache)
  struct Ent { void *p; size_t bytes; };
  std::vector<Ent> v;
  int device = -1;
  bool keep_all = false;      // FMX_FLAG_KEEP_SCRATCH
  uint8_t *arena = nullptr;   // small builds: bump allocation from the leased buffer
  size_t arena_cap = 0, arena_off = 0;
  explicit DevPool(int dev, bool keep = false) : device(dev), keep_all(keep) {}
  ~DevPool() {
    for (const Ent &e : v) scratch_put(device, e.p, e.bytes, keep_all);
    if (slab) scratch_put(device, slab, slab_bytes, keep_all);
  }
  void use_arena(uint8_t *base, size_t cap) { arena = base; arena_cap = base ? cap : 0; arena_off = 0; }
  // Very large builds (round 5): ONE block taken before the build's first kernel touches any memory, and every temporary
  // carved out of it (first fit; freed ranges coalesce).  On memory no process has used since boot this driver charges
  // every hipMalloc that FOLLOWS a first touch ~28 ms per GiB touched since the previous allocation (DESIGN.md section
  // 4.3: 5.4 s of a 0.85 s build at n = 2^32 on a fresh box) -- so a build that allocates nothing after its first touch
  // pays nothing.  What does not fit the slab falls back to the scratch cache / the driver as before.
  uint8_t *slab = nullptr;
  size_t slab_bytes = 0;
  std::vector<std::pair<size_t, size_t>> slab_free, slab_used;   // (offset, bytes), slab_free sorted by offset
  hipError_t use_slab(size_t bytes) {
    bytes = (bytes + 255u) & ~(size_t)255u;
    void *p = nullptr;
    size_t got = bytes;
    const hipError_t e = scratch_get(device, bytes, &p, &got);
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    slab = (uint8_t *)p;
    slab_bytes = got;
    slab_free.assign(1, {0, got});
    slab_used.clear();
    return hipSuccess;
  }
  void *slab_take(size_t bytes) {
    for (size_t i = 0; i < slab_free.size(); i++)
      if (slab_free[i].second >= bytes) {
        const size_t off = slab_free[i].first;
        if (slab_free[i].second == bytes) slab_free.erase(slab_free.begin() + i);
        else { slab_free[i].first += bytes; slab_free[i].second -= bytes; }
        slab_used.push_back({off, bytes});
        return slab + off;
      }
    return nullptr;
  }
  bool slab_give(void *p) {
    if (!slab || (uint8_t *)p < slab || (uint8_t *)p >= slab + slab_bytes) return false;
    const size_t off = (size_t)((uint8_t *)p - slab);
    for (size_t i = 0; i < slab_used.size(); i++)
      if (slab_used[i].first == off) {
        const size_t len = slab_used[i].second;
        slab_used.erase(slab_used.begin() + i);
        size_t j = 0;
        while (j < slab_free.size() && slab_free[j].first < off) j++;
        slab_free.insert(slab_free.begin() + j, {off, len});
        if (j + 1 < slab_free.size() && slab_free[j].first + slab_free[j].second == slab_free[j + 1].first) {
          slab_free[j].second += slab_free[j + 1].second;
          slab_free.erase(slab_free.begin() + j + 1);
        }
        if (j > 0 && slab_free[j - 1].first + slab_free[j - 1].second == slab_free[j].first) {
          slab_free[j - 1].second += slab_free[j].second;
          slab_free.erase(slab_free.begin() + j);
        }
        return true;
      }
    return true;                                      // (inside the slab but unknown: nothing to give back)
  }
  template <typename T>
  hipError_t get(T **out, size_t count) {
    const size_t bytes = (((count ? count : 1) * sizeof(T)) + 255u) & ~(size_t)255u;
    if (arena && arena_off + bytes <= arena_cap) {
      *out = (T *)(arena + arena_off);
      arena_off += bytes;
      return hipSuccess;
    }
    if (slab) {
      if (void *q = slab_take(bytes)) { *out = (T *)q; return hipSuccess; }
    }
    void *p = nullptr;
    size_t got = bytes;
    hipError_t e = scratch_get(device, bytes, &p, &got);
    if (e == hipSuccess) v.push_back({p, got});
    *out = (T *)p;
    return e;
  }
  // Before temporaries go back (to the scratch cache or the driver) the kernels that use them must have finished.  A
  // small build whose temporaries all came from its leased arena gives nothing back before it returns -- the bump
  // allocator never reuses a byte -- so it has nothing to wait for here: its launches queue up behind each other on the
  // stream and the builder's last synchronisation covers them (a small build is launch- and round-trip-bound: every
  // wait that goes lets the host enqueue ahead of the device)
  hipError_t quiesce() const { return (arena && v.empty() && !slab) ? hipSuccess : hipDeviceSynchronize(); }
  void release(void *p) {
    if (arena && (uint8_t *)p >= arena && (uint8_t *)p < arena + arena_cap) return;   // goes with the arena
    if (slab_give(p)) return;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].p == p) {
        scratch_put(device, p, v[i].bytes, keep_all);
        v.erase(v.begin() + i);
        return;
      }
  }
  // the buffer leaves the pool for good (handed over to the index)
  void disown(void *p) {
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].p == p) { v.erase(v.begin() + i); return; }
  }
};

inline unsigned nblocks(uint64_t n, unsigned per = BLK) {
  uint64_t b = (n + per - 1) / per;
  return (unsigned)(b ? b : 1);
}

// ---- text statistics: histogram, last non-zero index, (validation inputs) ------
struct TextStats {
  unsigned long long hist[256];
  unsigned long long last_nonzero_plus1;  // 0 when every symbol is zero
  unsigned long long max_sym;
  unsigned long long first_sym;           // t[0] (the validation reads it from here: no read-back of its own)
};
template <typename T>
__global__ __launch_bounds__(BLK) void k_text_stats(const T *__restrict__ t, uint64_t n,
                                                     TextStats *st) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long last = 0, mx = 0;
  const uint64_t stride = (uint64_t)gridDim.x * BLK;
  for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    uint32_t c = (uint32_t)t[i];
    if (c < 256u) atomicAdd(&h[c], 1u);
    if (c) last = i + 1;
    if (c > mx) mx = c;
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
  if (last) atomicMax(&st->last_nonzero_plus1, last);
  if (mx) atomicMax(&st->max_sym, mx);
  if (blockIdx.x == 0 && threadIdx.x == 0 && n) st->first_sym = (unsigned long long)t[0];
}
// large alphabets (max_character > 255): histogram straight into global memory
template <typename T>
__global__ __launch_bounds__(BLK) void k_hist_global(const T *__restrict__ t, uint64_t n,
                                                      uint32_t maxc,
                                                      unsigned long long *__restrict__ hist) {
  const uint64_t stride = (uint64_t)gridDim.x * BLK;
  for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    uint32_t c = (uint32_t)t[i];
    if (c <= maxc) atomicAdd(&hist[c], 1ull);
  }
}

// ---- suffix sorting by prefix doubling -----------------------------------------
// initial key: the first `k` symbols, `bits` bits each, zero-padded past the end
template <typename T>
__global__ __launch_bounds__(BLK) void k_init_keys(const T *__restrict__ t, uint32_t n,
                                                    uint32_t bits, uint32_t k,
                                                    uint64_t *__restrict__ keys,
                                                    uint32_t *__restrict__ idx) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  for (uint32_t j = 0; j < k; j++) {
    uint64_t p = i + j;
    uint64_t c = p < n ? t[p] : 0;
    key = (key << bits) | c;
  }
  keys[i] = key;
  idx[i] = (uint32_t)i;
}
// head[p] = p if sorted position p starts a new key group, else 0; *dup is set when any
// position is NOT a head (i.e. the order is not final yet).  Plain racing stores of the
// same value -- an atomic counter here cost 200 ms at n = 2^30.
__global__ __launch_bounds__(BLK) void k_flag_heads(const uint64_t *__restrict__ keys, uint32_t n,
                                                     uint32_t *__restrict__ head,
                                                     unsigned int *dup) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  bool is_head = (p == 0) || (keys[p] != keys[p - 1]);
  head[p] = is_head ? (uint32_t)p : 0u;
  if (!is_head) *dup = 1u;
}
struct MaxOp {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const {
    return a > b ? a : b;
  }
};
// exclusive prefix sum on the default stream, accumulated in the OUTPUT's value type
template <typename In, typename Out>
static hipError_t exclusive_sum(void *tmp, size_t &bytes, In in, Out out, size_t n) {
  using V = typename std::iterator_traits<Out>::value_type;
  return rocprim::exclusive_scan(tmp, bytes, in, out, V(0), n, rocprim::plus<V>(), (hipStream_t)0);
}
__global__ __launch_bounds__(BLK) void k_scatter_rank(const uint32_t *__restrict__ sa,
                                                       const uint32_t *__restrict__ head,
                                                       uint32_t n, uint32_t *__restrict__ rank) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p < n) rank[sa[p]] = head[p];
}
// doubling key: (rank[i], rank[i+h]+1 or 0 when the suffix ends first)
__global__ __launch_bounds__(BLK) void k_double_keys(const uint32_t *__restrict__ sa,
                                                      const uint32_t *__restrict__ rank,
                                                      uint32_t n, uint64_t h,
                                                      uint64_t *__restrict__ keys) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  uint64_t i = sa[p];
  uint64_t hi = rank[i];
  uint64_t j = i + h;
  uint64_t lo = j < n ? (uint64_t)rank[j] + 1ull : 0ull;
  keys[p] = (hi << 32) | lo;
}

// ---- BWT + samples -------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLK) void k_bwt(const T *__restrict__ t,
                                              const uint32_t *__restrict__ sa, uint32_t n,
                                              T *__restrict__ bwt) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  uint32_t k = sa[p];
  bwt[p] = k > 0 ? t[k - 1] : (T)0;  // fm_index.rs:50-55
}
// RLFM: c_i = T[SA[i]-1], or T[n-1] when SA[i] == 0  (rlfmi.rs:48-53)
template <typename T>
__global__ __launch_bounds__(BLK) void k_bwt_cyclic(const T *__restrict__ t,
                                                     const uint32_t *__restrict__ sa, uint32_t n,
                                                     T *__restrict__ bwt) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  uint32_t k = sa[p];
  bwt[p] = k > 0 ? t[k - 1] : t[n - 1];
}
__global__ __launch_bounds__(BLK) void k_samples(const uint32_t *__restrict__ sa, uint64_t nsamp,
                                                  uint32_t level, uint32_t *__restrict__ out) {
  uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (j < nsamp) out[j] = sa[j << level];  // sample.rs:35-37
}

// text-order sampling (FmxDev::phase): one thread packs the phases SA[row] mod 2^level of one piece
// (floor(32 / level) per word, three words) and counts its phase-0 rows; rows past the end get phase 1
__global__ __launch_bounds__(BLK) void k_phase_pieces(const uint32_t *__restrict__ sa, uint32_t n,
                                                       uint32_t level, uint32_t npieces,
                                                       uint4 *__restrict__ out, uint32_t *__restrict__ zeros) {
  const uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (j >= npieces) return;
  const uint32_t fpw = 32u / level, rpp = 3u * fpw, mask = (1u << level) - 1u;
  uint32_t w[3] = {0u, 0u, 0u}, z = 0;
  for (uint32_t t = 0; t < rpp; t++) {
    const uint64_t row = j * rpp + t;
    const uint32_t ph = row < n ? (sa[row] & mask) : 1u;
    z += ph == 0u;
    w[t / fpw] |= ph << ((t % fpw) * level);
  }
  out[j] = make_uint4(0u, w[0], w[1], w[2]);
  zeros[j] = z;
}
__global__ __launch_bounds__(BLK) void k_phase_counts(const uint32_t *__restrict__ base, uint32_t npieces,
                                                       uint4 *__restrict__ out) {
  const uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (j < npieces) out[j].x = base[j];
}
// ---- walk records (FmxDev::walk, fmx_internal.h) ----------------------------------------------------------------
// Derived from the fmt-3 records (code planes, lf_map2 counters) and the phase pieces (phases, phase-0 rank) alone,
// so that fmx_load can rebuild them; rows past the end have code 0 / phase 1 like their sources.
__device__ __forceinline__ uint32_t kw_code_of(const uint4 *__restrict__ rec, uint64_t row) {
  const uint4 p = rec[(size_t)(row >> 8) * 8u + ((row & 255u) >> 5)];
  return fmx_piece_code<3>(p, (uint32_t)(row & 31u));
}
__device__ __forceinline__ uint32_t kw_phase_of(const uint4 *__restrict__ phase, uint32_t row, uint32_t level) {
  uint32_t t, r0;
  const uint32_t pi = fmx_phase_piece(row, level, t);
  return fmx_phase_decode(phase[pi], t, level, r0);
}
// cnt[(c - 1) * nwalk + j] = rows of walk record j with phase 1 and BWT code c, c = 1..FMX_WALK_MAX_CHARACTER
__global__ __launch_bounds__(BLK) void k_walk_counts(const uint4 *__restrict__ rec, const uint4 *__restrict__ phase,
                                                      uint32_t n, uint32_t level, uint32_t nwalk,
                                                      uint32_t *__restrict__ cnt) {
  const uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (j >= nwalk) return;
  uint32_t c[FMX_WALK_MAX_CHARACTER + 1u];
  for (uint32_t k = 0; k <= FMX_WALK_MAX_CHARACTER; k++) c[k] = 0;
  for (uint32_t t = 0; t < FMX_WALK_ROWS; t++) {
    const uint64_t row = j * FMX_WALK_ROWS + t;
    if (row >= n) break;
    if (kw_phase_of(phase, (uint32_t)row, level) != 1u) continue;
    const uint32_t code = kw_code_of(rec, row);
    // (register array indexed by a loop-invariant compare chain, not by `code`: no scratch)
    for (uint32_t k = 1; k <= FMX_WALK_MAX_CHARACTER; k++) c[k] += code == k;
  }
  for (uint32_t k = 1; k <= FMX_WALK_MAX_CHARACTER; k++) cnt[(size_t)(k - 1u) * nwalk + j] = c[k];
}
// thread = piece g of walk record j.  base[(c - 1) * nwalk + j] = exclusive scan of the counts over the [code][record]
// layout = (phase-1 rows with a smaller code) + (phase-1 rows with code c before record j); `edge` = 1 when row 0
// (SA = n - 1) is a phase-0 row: it is the one phase-0 row that is not the LF image of a phase-1 row.
__global__ __launch_bounds__(BLK) void k_walk_records(const uint4 *__restrict__ rec, const uint4 *__restrict__ phase,
                                                       const uint32_t *__restrict__ base, uint32_t n, uint32_t level,
                                                       uint32_t nsamples, uint32_t nwalk, uint32_t edge,
                                                       uint4 *__restrict__ out) {
  const uint64_t tid = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (tid >= (uint64_t)nwalk * 8u) return;
  const uint32_t j = (uint32_t)(tid >> 3), g = (uint32_t)(tid & 7u);
  const uint64_t first = (uint64_t)j * FMX_WALK_ROWS;                // first row of the record
  auto rank1 = [&](uint32_t c) { return base[(size_t)(c - 1u) * nwalk + j] + edge; };
  if (g == 7u) {                                                     // the counter piece
    out[tid] = make_uint4(rank1(2), rank1(3), rank1(4), rank1(5));
    return;
  }
  uint32_t x;
  if (g < 5u) {
    // lf_map2(g + 1, first): the fmt-3 record holding `first`, its counter of the code + the occurrences before `first`
    const uint64_t fr = first <= n ? first : n;                      // (row n is addressable; beyond it nothing is read)
    const uint4 *R = rec + (size_t)(fr >> 8) * 8u;
    x = 0;
    for (uint32_t pp = 0; pp < 8u; pp++) x += fmx_piece_rank<3>(R[pp], (uint32_t)(fr & 255u), g + 1u, pp);
  } else if (g == 5u) {
    if (first < n) {
      uint32_t k;
      const uint32_t pi = fmx_phase_piece((uint32_t)first, level, k);
      (void)fmx_phase_decode(phase[pi], k, level, x);
    } else {
      x = nsamples;
    }
  } else {
    x = rank1(1);
  }
  // code planes and phase planes of the 16 rows [first + 16 g, + 16): one half of an fmt-3 piece (16 | first)
  const uint64_t row0 = first + g * 16u;
  uint32_t p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
  if (row0 < n) {
    const uint4 src = rec[(size_t)(row0 >> 8) * 8u + ((row0 & 255u) >> 5)];
    const uint32_t sh = (uint32_t)(row0 & 16u);
    p0 = (src.y >> sh) & 0xFFFFu; p1 = (src.z >> sh) & 0xFFFFu; p2 = (src.w >> sh) & 0xFFFFu;
  }
  for (uint32_t t = 0; t < 16u; t++) {
    const uint64_t row = row0 + t;
    const uint32_t ph = row < n ? kw_phase_of(phase, (uint32_t)row, level) : 1u;
    q0 |= (ph & 1u) << t;
    q1 |= ((ph >> 1) & 1u) << t;
    q2 |= ((ph >> 2) & 1u) << t;
  }
  out[tid] = make_uint4(x, p0 | (p1 << 16), p2 | (q0 << 16), q1 | (q2 << 16));
}

struct PhaseZero {   // flag of rocprim::select: rows whose suffix-array value is a multiple of 2^level
  uint32_t mask;
  __host__ __device__ bool operator()(uint32_t v) const { return (v & mask) == 0u; }
};

// 2-gram BWT for the opt-in pair index: code = (T[p-2]-1)*4 + (T[p-1]-1), p = SA[i] >= 2.
// The two rows with p < 2 have no 2-gram: they are stored as code 0 and reported in
// special[] so the query subtracts them from rank_0.
__global__ __launch_bounds__(BLK) void k_bwt2(const uint8_t *__restrict__ t,
                                               const uint32_t *__restrict__ sa, uint32_t n,
                                               uint8_t *__restrict__ out,
                                               uint32_t *__restrict__ special) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  uint32_t p = sa[i];
  if (p >= 2) {
    out[i] = (uint8_t)((t[p - 2] - 1u) * 4u + (t[p - 1] - 1u));
  } else {
    out[i] = 0;
    special[p] = (uint32_t)i;
  }
}

// ---- multi-ary wavelet matrix levels --------------------------------------------
// one thread per 16-B piece; planes written now, counters after the scan
template <int FMT, typename T>
__global__ __launch_bounds__(BLK) void k_mwm_pieces(const T *__restrict__ cur, uint64_t n,
                                                     uint32_t shift, uint32_t mask, uint32_t nrec,
                                                     uint4 *__restrict__ rec,
                                                     uint32_t *__restrict__ hist) {
  constexpr int PER = (FMT == 3) ? 32 : 16;
  constexpr int NCODE = (FMT == 3) ? 8 : 16;
  uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;  // piece index
  if (t >= (uint64_t)nrec * 8) return;                    // whole groups drop out together
  const uint32_t g = (uint32_t)(t & 7);
  const uint32_t r = (uint32_t)(t >> 3);
  const uint64_t base = t * PER;
  uint32_t pl[4] = {0, 0, 0, 0};
  uint32_t valid = 0;
  uint32_t sy[PER];
  if (sizeof(T) == 1 && base + PER <= n) {
    alignas(16) uint8_t raw[PER];
    const uint4 *src = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(cur) + base);
    for (int q = 0; q < PER / 16; q++) *reinterpret_cast<uint4 *>(&raw[q * 16]) = src[q];
    for (int j = 0; j < PER; j++) sy[j] = raw[j];
  } else {
    for (int j = 0; j < PER; j++) sy[j] = (base + j < n) ? (uint32_t)cur[base + j] : 0u;
  }
  for (int j = 0; j < PER; j++) {
    uint64_t p = base + j;
    if (p < n) {
      uint32_t code = (sy[j] >> shift) & mask;
      valid |= 1u << j;
      pl[0] |= (code & 1u) << j;
      pl[1] |= ((code >> 1) & 1u) << j;
      pl[2] |= ((code >> 2) & 1u) << j;
      if (FMT == 4) pl[3] |= ((code >> 3) & 1u) << j;
    }
  }
  uint4 piece;
  piece.x = 0;
  if (FMT == 3) {
    piece.y = pl[0]; piece.z = pl[1]; piece.w = pl[2];
  } else {
    piece.y = 0;
    piece.z = pl[0] | (pl[1] << 16);
    piece.w = pl[2] | (pl[3] << 16);
  }
  rec[t] = piece;
  // per-record histogram: lane g keeps the totals of the codes whose counters it owns
  uint32_t own0 = 0, own1 = 0;
  for (uint32_t code = 0; code < (uint32_t)NCODE; code++) {
    uint32_t m = fmx_piece_match<FMT>(piece, code) & valid;
    uint32_t tot = fmx_group_sum(__popc(m));
    if (FMT == 3) {
      if (g == code) own0 = tot;
    } else {
      if (g == (code >> 1)) { if (code & 1u) own1 = tot; else own0 = tot; }
    }
  }
  if (FMT == 3) {
    hist[(size_t)g * nrec + r] = own0;
  } else {
    hist[(size_t)(2 * g) * nrec + r] = own0;
    hist[(size_t)(2 * g + 1) * nrec + r] = own1;
  }
}
// scan = exclusive sum over hist laid out [code][record]; counter = scan - scan[code][0] + add,
// add = C[code] on a non-last level (the counter then IS the next-level position base) or the
// caller's table (cs[] when the matrix has a single level), else 0.
template <int FMT>
__global__ __launch_bounds__(BLK) void k_mwm_counters(const uint32_t *__restrict__ scan,
                                                       uint32_t nrec, uint4 *__restrict__ rec,
                                                       uint32_t *__restrict__ C, int fold_c,
                                                       const uint32_t *__restrict__ addend) {
  constexpr int NCODE = (FMT == 3) ? 8 : 16;
  uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t < 16) C[t] = t < NCODE ? scan[(size_t)t * nrec] : 0u;
  if (t >= (uint64_t)nrec * 8) return;
  const uint32_t g = (uint32_t)(t & 7);
  const uint32_t r = (uint32_t)(t >> 3);
  uint4 p = rec[t];
  if (FMT == 3) {
    uint32_t b = scan[(size_t)g * nrec];
    p.x = scan[(size_t)g * nrec + r] - b + (fold_c ? b : 0u) + (addend ? addend[g] : 0u);
  } else {
    uint32_t b0 = scan[(size_t)(2 * g) * nrec], b1 = scan[(size_t)(2 * g + 1) * nrec];
    p.x = scan[(size_t)(2 * g) * nrec + r] - b0 + (fold_c ? b0 : 0u) + (addend ? addend[2 * g] : 0u);
    p.y = scan[(size_t)(2 * g + 1) * nrec + r] - b1 + (fold_c ? b1 : 0u) +
          (addend ? addend[2 * g + 1] : 0u);
  }
  rec[t] = p;
}

// ---- select hints of a wavelet level (fl_map: select_u64, fm_index.rs:118 / rlfmi.rs:166) ----
template <int FMT>
__device__ __forceinline__ uint32_t lvl_counter(const uint4 *rec, uint32_t r, uint32_t code) {
  if (FMT == 3) return rec[(size_t)r * 8u + code].x;
  const uint4 p = rec[(size_t)r * 8u + (code >> 1)];
  return (code & 1u) ? p.y : p.x;
}
// meta[c] = counter of code c at record 0; meta[32 + c] = entries with code c; meta[16 + c] = start
// of the code's hints (cnt / STEP + 2 entries each)
template <int FMT>
__global__ void k_wsel_meta(const uint4 *__restrict__ rec, const uint32_t *__restrict__ scan,
                            uint32_t nrec, uint32_t total, uint32_t *__restrict__ meta) {
  constexpr uint32_t NCODE = (FMT == 3) ? 8u : 16u;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t start = 0;
  for (uint32_t c = 0; c < 16u; c++) {
    uint32_t cnt = 0, base = 0;
    if (c < NCODE) {
      const uint32_t lo = scan[(size_t)c * nrec];
      const uint32_t hi = c + 1u < NCODE ? scan[(size_t)(c + 1u) * nrec] : total;
      cnt = hi - lo;
      base = lvl_counter<FMT>(rec, 0, c);
    }
    meta[c] = base;
    meta[16u + c] = start;
    meta[32u + c] = cnt;
    start += cnt / FMX_WSEL_STEP + 2u;
  }
}
template <int FMT>
__global__ __launch_bounds__(BLK) void k_wsel_hints(const uint4 *__restrict__ rec, uint32_t nrec,
                                                     const uint32_t *__restrict__ meta,
                                                     uint32_t *__restrict__ sel) {
  constexpr uint32_t NCODE = (FMT == 3) ? 8u : 16u;
  const uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  const uint32_t r = (uint32_t)(t / NCODE), c = (uint32_t)(t % NCODE);
  if (r >= nrec) return;
  const uint32_t base = meta[c], cnt = meta[32u + c];
  const uint32_t a = lvl_counter<FMT>(rec, r, c) - base;
  const uint32_t b = r + 1u < nrec ? lvl_counter<FMT>(rec, r + 1u, c) - base : cnt;
  uint32_t m = (a + FMX_WSEL_STEP - 1u) / FMX_WSEL_STEP * FMX_WSEL_STEP;
  for (; m < b; m += FMX_WSEL_STEP) sel[meta[16u + c] + m / FMX_WSEL_STEP] = r;
}

// ---- RLFM construction (rlfmi.rs:30-96) --------------------------------------------
// run starts: c0 starts at 0, a run begins wherever c != c0  (rlfmi.rs:41, 56-59)
template <typename T>
__global__ __launch_bounds__(BLK) void k_run_flags(const T *__restrict__ L, uint32_t n,
                                                    uint8_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  T prev = i ? L[i - 1] : (T)0;
  flags[i] = L[i] != prev ? 1 : 0;
}
__global__ __launch_bounds__(BLK) void k_iota(uint32_t *out, uint32_t n) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) out[i] = (uint32_t)i;
}
// run length of the q-th run in (head, row) order
__global__ __launch_bounds__(BLK) void k_sorted_run_lens(const uint32_t *__restrict__ starts,
                                                          const uint32_t *__restrict__ order,
                                                          uint32_t r, uint32_t n,
                                                          uint32_t *__restrict__ lens) {
  uint64_t q = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (q >= r) return;
  uint32_t k = order[q];
  uint32_t end = k + 1 < r ? starts[k + 1] : n;
  lens[q] = end - starts[k];
}
// lfrun[run] = F position of the run (FmxDev::lfrun): order[t] = the run that comes t-th in (head, row) order
__global__ __launch_bounds__(BLK) void k_scatter_lfrun(const uint32_t *__restrict__ order, const uint32_t *__restrict__ fpos,
                                                        uint32_t r, uint32_t *__restrict__ lfrun) {
  const uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t < r) lfrun[order[t]] = fpos[t];
}
__global__ __launch_bounds__(BLK) void k_scatter_ones(const uint32_t *__restrict__ pos, uint32_t r,
                                                       uint8_t *__restrict__ flags) {
  uint64_t q = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (q < r) flags[pos[q]] = 1;
}
// bit-vector records: one thread per 96-bit piece
__global__ __launch_bounds__(BLK) void k_bits_pieces(const uint8_t *__restrict__ flags, uint32_t n,
                                                      uint32_t npieces, uint4 *__restrict__ rec,
                                                      uint32_t *__restrict__ cnt) {
  uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t >= npieces) return;
  uint64_t base = t * FMX_BITS_PER_PIECE;
  uint32_t w[3] = {0, 0, 0};
  for (uint32_t j = 0; j < FMX_BITS_PER_PIECE; j++) {
    uint64_t p = base + j;
    if (p < n && flags[p]) w[j >> 5] |= 1u << (j & 31u);
  }
  uint4 pc;
  pc.x = 0; pc.y = w[0]; pc.z = w[1]; pc.w = w[2];
  rec[t] = pc;
  cnt[t] = __popc(w[0]) + __popc(w[1]) + __popc(w[2]);
}
__global__ __launch_bounds__(BLK) void k_bits_counters(const uint32_t *__restrict__ base,
                                                        uint32_t npieces, uint4 *__restrict__ rec) {
  uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t < npieces) rec[t].x = base[t];
}
__global__ __launch_bounds__(BLK) void k_fill_u32(uint32_t *out, uint32_t n, uint32_t v) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) out[i] = v;
}
// sel[m / STEP] = record holding the m-th one, for every multiple m of STEP
__global__ __launch_bounds__(BLK) void k_select_hints(const uint4 *__restrict__ rec, uint32_t nrec,
                                                       uint32_t ones, uint32_t *__restrict__ sel) {
  uint64_t r = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (r >= nrec) return;
  uint32_t b0 = rec[r * 8].x;
  uint32_t b1 = r + 1 < nrec ? rec[(r + 1) * 8].x : ones;
  uint32_t m = (b0 + FMX_SEL_STEP - 1) / FMX_SEL_STEP * FMX_SEL_STEP;
  for (; m < b1; m += FMX_SEL_STEP) sel[m / FMX_SEL_STEP] = (uint32_t)r;
}

// ---- multi-pieces: doc[] and sa_idx_first_text (multi_pieces.rs:57-85) ----------------
template <typename T>
__global__ __launch_bounds__(BLK) void k_zero_flags(const T *__restrict__ a, uint32_t n,
                                                     uint32_t *__restrict__ flags) {
  uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (i < n) flags[i] = a[i] == 0 ? 1u : 0u;
}
// for the k-th end marker of L (row p): doc[k] = number of end markers before text position
// (SA[p] - 1) mod n; the row whose marker is the LAST of the text is sa_idx_first_text
__global__ __launch_bounds__(BLK) void k_doc(const uint32_t *__restrict__ sa,
                                              const uint32_t *__restrict__ zl_flag,
                                              const uint32_t *__restrict__ zl_rank,
                                              const uint32_t *__restrict__ zt_rank, uint32_t n,
                                              uint32_t pieces, uint32_t *__restrict__ doc,
                                              uint32_t *__restrict__ first_row) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n || !zl_flag[p]) return;
  uint32_t k = sa[p];
  uint32_t pos = k > 0 ? k - 1 : n - 1;   // modular_sub(sa[p], 1, n)
  uint32_t piece = zt_rank[pos];          // end_marker_flags.rank1(end_marker_idx)
  doc[zl_rank[p]] = piece;
  if (piece == pieces - 1) *first_row = (uint32_t)p;
}

// ---- verification (FMX_FLAG_KEEP_SA) ---------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLK) void k_verify_sa(const T *__restrict__ t,
                                                    const uint32_t *__restrict__ sa, uint32_t n,
                                                    uint32_t *__restrict__ mark,
                                                    unsigned long long *bad) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  uint32_t b = sa[p];
  if (b >= n) { atomicAdd(bad, 1ull); return; }
  atomicAdd(&mark[b], 1u);
  if (p == 0) return;
  uint32_t a = sa[p - 1];
  if (a >= n) return;
  // suffix a must be < suffix b (slice order: a suffix that ends first is smaller)
  for (uint64_t j = 0;; j++) {
    uint64_t pa = a + j, pb = b + j;
    if (pa >= n) return;                     // a ended first: ok
    if (pb >= n) { atomicAdd(bad, 1ull); return; }
    T ca = t[pa], cb = t[pb];
    if (ca < cb) return;
    if (ca > cb) { atomicAdd(bad, 1ull); return; }
  }
}
__global__ __launch_bounds__(BLK) void k_count_not_one(const uint32_t *mark, uint32_t n,
                                                        unsigned long long *bad) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p < n && mark[p] != 1u) atomicAdd(bad, 1ull);
}

// ---- helpers ---------------------------------------------------------------------
int keep(fmx_index *idx, void *p, uint64_t bytes) { return fmx_keep(idx, p, bytes); }

// level split: fewest levels (<= 4 bits each), widths as even as possible, wide ones first
// L=3 -> [3]; 5 -> [3,2]; 7 -> [4,3]; 8 -> [4,4]; 9 -> [3,3,3]
void split_levels(uint32_t L, uint32_t *nlv, uint32_t *bits) {
  uint32_t k = (L + 3) / 4;
  if (k == 0) k = 1;
  uint32_t base = L / k, extra = L % k;
  for (uint32_t l = 0; l < k; l++) bits[l] = base + (l < extra ? 1u : 0u);
  *nlv = k;
}

// ---- refinement rounds of the prefix doubling: only the suffixes whose rank is not final yet ----------------
// grp[p] = first sorted position of the group of equal keys position p belongs to (the max-scan of k_flag_heads).
// A position is ACTIVE while its group has more than one member.
__global__ __launch_bounds__(BLK) void k_active_flags(const uint32_t *__restrict__ grp, uint32_t n,
                                                       uint8_t *__restrict__ flags) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  const bool single = grp[p] == (uint32_t)p && (p + 1 == n || grp[p + 1] == (uint32_t)(p + 1));
  flags[p] = single ? 0 : 1;
}
// the same over the compacted list: apos[k] = sorted position of the k-th active suffix, grp[k] its group's first position
__global__ __launch_bounds__(BLK) void k_active_flags_c(const uint32_t *__restrict__ apos, const uint32_t *__restrict__ grp,
                                                         uint32_t m, uint8_t *__restrict__ flags) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  const bool single = grp[k] == apos[k] && (k + 1 == m || grp[k + 1] == apos[k + 1]);
  flags[k] = single ? 0 : 1;
}
// doubling key of the k-th active suffix: (its group, rank[i + h] + 1 or 0 when the suffix ends first)
__global__ __launch_bounds__(BLK) void k_refine_keys(const uint32_t *__restrict__ apos, const uint32_t *__restrict__ sa,
                                                      const uint32_t *__restrict__ rank, uint32_t n, uint64_t h, uint32_t m,
                                                      uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  const uint64_t i = sa[apos[k]];
  const uint64_t j = i + h;
  const uint64_t lo = j < n ? (uint64_t)rank[j] + 1ull : 0ull;
  keys[k] = ((uint64_t)rank[i] << 32) | lo;
  vals[k] = (uint32_t)i;
}
// the sorted active suffixes go back to the active positions (ascending keys <-> ascending positions: a group's
// positions are contiguous and all active); head[k] = apos[k] where a new group starts, else 0
__global__ __launch_bounds__(BLK) void k_refine_write(const uint32_t *__restrict__ apos, const uint64_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ vals, uint32_t m,
                                                       uint32_t *__restrict__ sa, uint32_t *__restrict__ head) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  sa[apos[k]] = vals[k];
  head[k] = (k == 0 || keys[k] != keys[k - 1]) ? apos[k] : 0u;
}
__global__ __launch_bounds__(BLK) void k_refine_rank(const uint32_t *__restrict__ vals, const uint32_t *__restrict__ grp,
                                                      uint32_t m, uint32_t *__restrict__ rank) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k < m) rank[vals[k]] = grp[k];
}
// ---- the first refinement round straight from the text: the key of suffix i + h over its first h symbols IS what
// k_init_keys packs, so the tied suffixes can be told apart to depth 2h before any rank exists (no scan over all
// positions, no scatter of all ranks -- 40 of the 54 ms those cost at n = 2^30 when a few thousand suffixes are tied)
template <typename T>
__device__ __forceinline__ uint64_t pack_key(const T *__restrict__ t, uint32_t n, uint64_t pos, uint32_t bits, uint32_t k) {
  uint64_t key = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint64_t p = pos + j;
    key = (key << bits) | (uint64_t)(p < n ? t[p] : 0);
  }
  return key;
}
// tied positions from the UNSCANNED heads of k_flag_heads (head[p] = p at a group's first position, 0 elsewhere)
__global__ __launch_bounds__(BLK) void k_active_flags_h(const uint32_t *__restrict__ head, uint32_t n, uint8_t *__restrict__ flags) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p >= n) return;
  const bool h0 = p == 0 || head[p] != 0u, h1 = p + 1 == n || head[p + 1] != 0u;
  flags[p] = (h0 && h1) ? 0 : 1;
}
__global__ __launch_bounds__(BLK) void k_text_round_heads(const uint32_t *__restrict__ apos, const uint32_t *__restrict__ head,
                                                           uint32_t m, uint32_t *__restrict__ grp) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  const uint32_t p = apos[k];
  grp[k] = (p == 0 || head[p] != 0u) ? p : 0u;      // position 0 starts a group whose id is 0: the max-scan keeps it
}
template <typename T>
__global__ __launch_bounds__(BLK) void k_text_round_keys(const uint32_t *__restrict__ apos, const uint32_t *__restrict__ sa,
                                                          const T *__restrict__ t, uint32_t n, uint32_t bits, uint32_t ksym,
                                                          uint64_t h, uint32_t m, uint64_t *__restrict__ keys,
                                                          uint32_t *__restrict__ idx) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  keys[k] = pack_key<T>(t, n, (uint64_t)sa[apos[k]] + h, bits, ksym);
  idx[k] = (uint32_t)k;
}
__global__ __launch_bounds__(BLK) void k_text_round_group_keys(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ grp,
                                                                uint32_t m, uint64_t *__restrict__ keys) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k < m) keys[k] = grp[idx[k]];
}
__global__ __launch_bounds__(BLK) void k_text_round_gather(const uint32_t *__restrict__ apos, const uint32_t *__restrict__ idx,
                                                            const uint32_t *__restrict__ sa, uint32_t m, uint32_t *__restrict__ suf) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k < m) suf[k] = sa[apos[idx[k]]];
}
template <typename T>
__global__ __launch_bounds__(BLK) void k_text_round_write(const uint32_t *__restrict__ apos, const uint64_t *__restrict__ gkeys,
                                                           const uint32_t *__restrict__ suf, const T *__restrict__ t, uint32_t n,
                                                           uint32_t bits, uint32_t ksym, uint64_t h, uint32_t m,
                                                           uint32_t *__restrict__ sa, uint32_t *__restrict__ head) {
  uint64_t k = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (k >= m) return;
  const uint32_t i = suf[k];
  sa[apos[k]] = i;
  bool is_head = k == 0 || gkeys[k] != gkeys[k - 1];
  if (!is_head) is_head = pack_key<T>(t, n, (uint64_t)i + h, bits, ksym) != pack_key<T>(t, n, (uint64_t)suf[k - 1] + h, bits, ksym);
  head[k] = is_head ? apos[k] : 0u;
}
// rank[i] = sorted position of suffix i, for every suffix (the still-tied ones get their group's afterwards)
__global__ __launch_bounds__(BLK) void k_rank_identity(const uint32_t *__restrict__ sa, uint32_t n, uint32_t *__restrict__ rank) {
  uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (p < n) rank[sa[p]] = (uint32_t)p;
}
struct AsU32 {
  __device__ __forceinline__ uint32_t operator()(uint8_t f) const { return f; }
};
constexpr uint32_t kRefineMinN = 1u << 19;   // texts from this length on take the refinement rounds
static inline uint32_t refine_min_n() {
#ifdef FMX_MEASURE
  // measurement build: the tests run the refinement rounds on small texts too (tests/test_gpu_fuzz.py)
  if (const char *v = getenv("FMX_REFINE_MIN_N")) return (uint32_t)atol(v);
#endif
  return kRefineMinN;
}

// suffix array of d_text[0..n) into d_sa (u32) -- prefix doubling
template <typename T>
int suffix_sort(const T *d_text, uint32_t n, uint32_t sym_bits, uint32_t *d_sa, DevPool &pool) {
  uint64_t *keys_a, *keys_b;
  uint32_t *vals_b, *rank = nullptr, *head = nullptr;
  unsigned int *d_ng;
  static const bool trace = fmx_build_trace();
  auto ts0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what, uint64_t h) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[fmx build]   sort: %-10s h=%-10llu %8.1f ms\n", what, (unsigned long long)h,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
  };
  FMX_HIP(pool.get(&keys_a, n));
  FMX_HIP(pool.get(&keys_b, n));
  FMX_HIP(pool.get(&vals_b, n));
  FMX_HIP(pool.get(&d_ng, 1));
  // `head` and `rank` live in the radix sort's ALTERNATE buffers, which are free between two sorts
  // (24 instead of 32 bytes of scratch per symbol); assigned after each sort below
  uint32_t k = 64 / sym_bits;
  if (k > 32) k = 32;
  // temp storage: the larger of the sort and scan requirements
  size_t tmp_sort = 0, tmp_scan = 0;
  {
    rocprim::double_buffer<uint64_t> kb(keys_a, keys_b);
    rocprim::double_buffer<uint32_t> vb(d_sa, vals_b);
    FMX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort, kb, vb, (size_t)n, 0u, 64u, (hipStream_t)0));
    FMX_HIP(rocprim::inclusive_scan(nullptr, tmp_scan, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n,
                                    MaxOp(), (hipStream_t)0));
  }
  size_t tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tmp_bytes));

  mark("alloc", 0);
  // rocPRIM calls of the refinement rounds: `tmp` when it is large enough, a buffer of their own otherwise
  auto scratch = [&](size_t need, uint8_t **out, bool *own) -> hipError_t {
    *own = need > tmp_bytes;
    if (!*own) { *out = tmp; return hipSuccess; }
    return pool.get(out, need);
  };
  auto count_flags = [&](const uint8_t *fl, uint32_t cnt, uint32_t *out) -> int {
    auto in = rocprim::make_transform_iterator(fl, AsU32());
    size_t need = 0;
    FMX_HIP(rocprim::reduce(nullptr, need, in, d_ng, 0u, (size_t)cnt, rocprim::plus<uint32_t>(), (hipStream_t)0));
    uint8_t *t; bool own;
    FMX_HIP(scratch(need, &t, &own));
    FMX_HIP(rocprim::reduce(t, need, in, d_ng, 0u, (size_t)cnt, rocprim::plus<uint32_t>(), (hipStream_t)0));
    FMX_HIP(hipMemcpy(out, d_ng, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (own) pool.release(t);
    return FMX_OK;
  };
  auto compact = [&](auto in, const uint8_t *fl, uint32_t cnt, uint32_t *out, uint32_t expect) -> int {
    size_t need = 0;
    FMX_HIP(rocprim::select(nullptr, need, in, fl, out, d_ng, (size_t)cnt, (hipStream_t)0));
    uint8_t *t; bool own;
    FMX_HIP(scratch(need, &t, &own));
    FMX_HIP(rocprim::select(t, need, in, fl, out, d_ng, (size_t)cnt, (hipStream_t)0));
    unsigned int got = 0;
    FMX_HIP(hipMemcpy(&got, d_ng, sizeof got, hipMemcpyDeviceToHost));
    if (own) pool.release(t);
    if (got != expect) { fmx_set_error(FMX_ERR_HIP, "suffix sort: compaction lost count"); return FMX_ERR_HIP; }
    return FMX_OK;
  };
  // refinement: the m suffixes at the sorted positions apos[] are still tied after h symbols; sort them inside their
  // groups by rank[i + h], give the new groups their ranks, drop the ones that are alone now, double h
  auto refine = [&](uint32_t *apos, uint32_t m, uint32_t *sa, uint32_t *rk, uint64_t h) -> int {
    uint64_t *ck_a, *ck_b;
    uint32_t *cv_a, *cv_b, *grp, *apos2;
    uint8_t *fl;
    FMX_HIP(pool.get(&ck_a, m)); FMX_HIP(pool.get(&ck_b, m));
    FMX_HIP(pool.get(&cv_a, m)); FMX_HIP(pool.get(&cv_b, m));
    FMX_HIP(pool.get(&grp, m));  FMX_HIP(pool.get(&apos2, m));
    FMX_HIP(pool.get(&fl, m));
    while (m) {
      if (h >= n) {  // cannot happen for distinct suffixes; guard against an endless loop
        fmx_set_error(FMX_ERR_HIP, "suffix sort did not converge");
        return FMX_ERR_HIP;
      }
      hipLaunchKernelGGL(k_refine_keys, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, sa, rk, n, h, m, ck_a, cv_a);
      rocprim::double_buffer<uint64_t> kb(ck_a, ck_b);
      rocprim::double_buffer<uint32_t> vb(cv_a, cv_b);
      size_t need = 0;
      FMX_HIP(rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)m, 0u, 64u, (hipStream_t)0));
      uint8_t *t; bool own;
      FMX_HIP(scratch(need, &t, &own));
      FMX_HIP(rocprim::radix_sort_pairs(t, need, kb, vb, (size_t)m, 0u, 64u, (hipStream_t)0));
      hipLaunchKernelGGL(k_refine_write, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, kb.current(), vb.current(), m, sa, grp);
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      size_t tb2 = 0;
      FMX_HIP(rocprim::inclusive_scan(nullptr, tb2, grp, grp, (size_t)m, MaxOp(), (hipStream_t)0));
      FMX_HIP(scratch(tb2, &t, &own));
      FMX_HIP(rocprim::inclusive_scan(t, tb2, grp, grp, (size_t)m, MaxOp(), (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      hipLaunchKernelGGL(k_refine_rank, dim3(nblocks(m)), dim3(BLK), 0, 0, vb.current(), grp, m, rk);
      hipLaunchKernelGGL(k_active_flags_c, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, grp, m, fl);
      FMX_HIP(hipGetLastError());
      uint32_t m2 = 0;
      if (int rc = count_flags(fl, m, &m2)) return rc;
      h *= 2;
      mark("refine", h);
      if (m2) {
        if (int rc = compact(apos, fl, m, apos2, m2)) return rc;
        uint32_t *x = apos; apos = apos2; apos2 = x;
      }
      m = m2;
    }
    FMX_HIP(hipDeviceSynchronize());
    pool.release(ck_a); pool.release(ck_b); pool.release(cv_a); pool.release(cv_b);
    pool.release(grp); pool.release(apos); pool.release(apos2); pool.release(fl);
    return FMX_OK;
  };
  // The text round (see k_text_round_*): `flags` marks the m tied positions, `head0` are the unscanned heads.  Sorts
  // the tied suffixes inside their groups by the text key at depth h, and -- if some are still tied -- builds the
  // ranks (identity + the tied groups') and hands over to refine() at depth 2h.  *done = 1: the order is final.
  auto text_round = [&](const uint8_t *fl0, const uint32_t *head0, uint32_t m, uint32_t *sa, uint32_t *rk, uint64_t h,
                        int *done) -> int {
    uint32_t *apos, *grp0, *ix_a, *ix_b, *suf, *grp;
    uint64_t *tk_a, *tk_b;
    uint8_t *fl;
    FMX_HIP(pool.get(&apos, m)); FMX_HIP(pool.get(&grp0, m));
    if (int rc = compact(rocprim::counting_iterator<uint32_t>(0), fl0, n, apos, m)) return rc;
    hipLaunchKernelGGL(k_text_round_heads, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, head0, m, grp0);
    FMX_HIP(hipDeviceSynchronize());
    pool.release(keys_a); pool.release(keys_b);               // flags and heads lived there
    keys_a = keys_b = nullptr;
    FMX_HIP(pool.get(&tk_a, m)); FMX_HIP(pool.get(&tk_b, m));
    FMX_HIP(pool.get(&ix_a, m)); FMX_HIP(pool.get(&ix_b, m));
    FMX_HIP(pool.get(&suf, m));  FMX_HIP(pool.get(&grp, m));
    FMX_HIP(pool.get(&fl, m));
    uint8_t *t; bool own;
    auto scan_max = [&](uint32_t *a) -> int {
      size_t need = 0;
      FMX_HIP(rocprim::inclusive_scan(nullptr, need, a, a, (size_t)m, MaxOp(), (hipStream_t)0));
      FMX_HIP(scratch(need, &t, &own));
      FMX_HIP(rocprim::inclusive_scan(t, need, a, a, (size_t)m, MaxOp(), (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      return FMX_OK;
    };
    if (int rc = scan_max(grp0)) return rc;                   // grp0[k] = first position of the k-th tied suffix's group
    uint64_t *kc = tk_a, *ka = tk_b;
    uint32_t *vc = ix_a, *va = ix_b;
    auto pass = [&](unsigned end_bit) -> int {
      rocprim::double_buffer<uint64_t> kb(kc, ka);
      rocprim::double_buffer<uint32_t> vb(vc, va);
      size_t need = 0;
      FMX_HIP(rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
      FMX_HIP(scratch(need, &t, &own));
      FMX_HIP(rocprim::radix_sort_pairs(t, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      kc = kb.current(); ka = kb.alternate();
      vc = vb.current(); va = vb.alternate();
      return FMX_OK;
    };
    hipLaunchKernelGGL(k_text_round_keys<T>, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, sa, d_text, n, sym_bits, k, h, m, kc, vc);
    if (int rc = pass(k * sym_bits)) return rc;               // by the text key at depth h ...
    hipLaunchKernelGGL(k_text_round_group_keys, dim3(nblocks(m)), dim3(BLK), 0, 0, vc, grp0, m, kc);
    if (int rc = pass(32u)) return rc;                        // ... then, stable, by the group
    hipLaunchKernelGGL(k_text_round_gather, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, vc, sa, m, suf);
    hipLaunchKernelGGL(k_text_round_write<T>, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, kc, suf, d_text, n, sym_bits, k, h, m,
                       sa, grp);
    if (int rc = scan_max(grp)) return rc;
    hipLaunchKernelGGL(k_active_flags_c, dim3(nblocks(m)), dim3(BLK), 0, 0, apos, grp, m, fl);
    FMX_HIP(hipGetLastError());
    uint32_t m2 = 0;
    if (int rc = count_flags(fl, m, &m2)) return rc;
    mark("text round", m2);
    *done = 1;
    uint32_t *apos2 = nullptr;
    if (m2) {
      // some are tied beyond 2h symbols: every suffix gets its rank (its position; a tied one its group's first
      // position), and the doubling rounds take over
      FMX_HIP(pool.get(&apos2, m2));
      if (int rc = compact(apos, fl, m, apos2, m2)) return rc;
      hipLaunchKernelGGL(k_rank_identity, dim3(nblocks(n)), dim3(BLK), 0, 0, sa, n, rk);
      hipLaunchKernelGGL(k_refine_rank, dim3(nblocks(m)), dim3(BLK), 0, 0, suf, grp, m, rk);
      FMX_HIP(hipGetLastError());
    }
    FMX_HIP(hipDeviceSynchronize());
    pool.release(tk_a); pool.release(tk_b); pool.release(ix_a); pool.release(ix_b);
    pool.release(suf); pool.release(grp); pool.release(fl); pool.release(grp0); pool.release(apos);
    if (m2) return refine(apos2, m2, sa, rk, 2 * h);
    return FMX_OK;
  };
  hipLaunchKernelGGL(k_init_keys<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_text, n, sym_bits, k, keys_a,
                     d_sa);
  uint64_t *keys_cur = keys_a, *keys_alt = keys_b;
  uint32_t *sa_cur = d_sa, *sa_alt = vals_b;
  int end_bit = (int)(k * sym_bits);
  uint64_t h = k;
  for (;;) {
    rocprim::double_buffer<uint64_t> kb(keys_cur, keys_alt);
    rocprim::double_buffer<uint32_t> vb(sa_cur, sa_alt);
    size_t tb = tmp_bytes;
    FMX_HIP(rocprim::radix_sort_pairs(tmp, tb, kb, vb, (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)0));
    keys_cur = kb.current(); keys_alt = kb.alternate();
    sa_cur = vb.current();   sa_alt = vb.alternate();
    head = (uint32_t *)keys_alt;
    rank = sa_alt;
    FMX_HIP(hipMemsetAsync(d_ng, 0, sizeof(unsigned int), 0));
    hipLaunchKernelGGL(k_flag_heads, dim3(nblocks(n)), dim3(BLK), 0, 0, keys_cur, n, head, d_ng);
    unsigned int dup = 0;
    FMX_HIP(hipMemcpy(&dup, d_ng, sizeof dup, hipMemcpyDeviceToHost));
    mark("round", h);
    if (!dup) break;
    if (h >= n) {  // cannot happen for distinct suffixes; guard against an endless loop
      fmx_set_error(FMX_ERR_HIP, "suffix sort did not converge");
      return FMX_ERR_HIP;
    }
    if (h == k && n >= refine_min_n()) {
      // first round: when at most an eighth of the suffixes are tied, they are told apart to depth 2h by keys
      // packed from the text, and only if some are STILL tied does any rank get computed
      uint8_t *flags = (uint8_t *)keys_cur;
      uint32_t m = 0;
      hipLaunchKernelGGL(k_active_flags_h, dim3(nblocks(n)), dim3(BLK), 0, 0, head, n, flags);
      if (int rc = count_flags(flags, n, &m)) return rc;
      mark("tied", m);
      if ((uint64_t)m * 8 <= n) {
        int done = 0;
        if (int rc = text_round(flags, head, m, sa_cur, rank, h, &done)) return rc;
        if (done) break;
      }
    }
    tb = tmp_bytes;
    FMX_HIP(rocprim::inclusive_scan(tmp, tb, head, head, (size_t)n, MaxOp(), (hipStream_t)0));
    hipLaunchKernelGGL(k_scatter_rank, dim3(nblocks(n)), dim3(BLK), 0, 0, sa_cur, head, n, rank);
    // how many suffixes are still tied?  (the keys are spent: their buffer takes the flags)  Not asked of small texts:
    // the count, the compaction and the refinement's own launches and round trips (~60 us) cost them more than
    // whole rounds do (n = 10^4: 431 -> 487 us, n = 10^5: 546 -> 614 us; n = 10^6: 2028 -> 1770 us)
    uint8_t *flags = (uint8_t *)keys_cur;
    uint32_t m = n;
    if (n >= refine_min_n()) {
      hipLaunchKernelGGL(k_active_flags, dim3(nblocks(n)), dim3(BLK), 0, 0, head, n, flags);
      if (int rc = count_flags(flags, n, &m)) return rc;
      mark("active", m);
    }
    if ((uint64_t)m * 4 <= n) {
      // few enough: from here on only they are sorted, in buffers of their own (<= 41 m bytes for the 16 n released)
      uint32_t *apos;
      FMX_HIP(pool.get(&apos, m));
      if (int rc = compact(rocprim::counting_iterator<uint32_t>(0), flags, n, apos, m)) return rc;
      FMX_HIP(hipDeviceSynchronize());
      pool.release(keys_a); pool.release(keys_b);
      keys_a = keys_b = nullptr;
      if (int rc = refine(apos, m, sa_cur, rank, h)) return rc;
      break;
    }
    hipLaunchKernelGGL(k_double_keys, dim3(nblocks(n)), dim3(BLK), 0, 0, sa_cur, rank, n, h,
                       keys_cur);
    end_bit = 64;
    h *= 2;
  }
  if (sa_cur != d_sa)
    FMX_HIP(hipMemcpyAsync(d_sa, sa_cur, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, 0));
  FMX_HIP(pool.quiesce());
  if (keys_a) pool.release(keys_a);
  if (keys_b) pool.release(keys_b);
  pool.release(vals_b);
  pool.release(tmp); pool.release(d_ng);
  return FMX_OK;
}

// builds the multi-ary wavelet matrix over d_seq[0..len) (u8 symbols of `L` bits).
// d_seq is consumed (sorted in place between levels).
template <typename T>
int build_mwm(fmx_index *idx, FmxMwm *w, T *d_seq, uint32_t len, uint32_t L, DevPool &pool,
              const uint64_t *single_level_add = nullptr, uint32_t nadd = 0, bool want_select = true) {
  uint32_t nlv, bits[FMX_MAX_LEVELS];
  split_levels(L, &nlv, bits);
  memset(w, 0, sizeof *w);
  w->nlevels = nlv;
  w->bits = L;
  w->len = len;
  T *cur = d_seq, *alt = nullptr;
  if (nlv > 1) FMX_HIP(pool.get(&alt, len));
  uint32_t shift = L;
  for (uint32_t l = 0; l < nlv; l++) {
    shift -= bits[l];
    FmxLevel &lv = w->lv[l];
    lv.fmt = bits[l] == 4 ? 4 : 3;
    lv.shift = shift;
    lv.mask = (1u << bits[l]) - 1u;
    uint32_t per_rec = lv.fmt == 3 ? 256u : 128u;
    lv.nrec = len / per_rec + 1;  // +1: position `len` itself must be addressable
    uint32_t ncode = lv.fmt == 3 ? 8u : 16u;
    uint4 *rec;
    uint32_t *C, *hist, *scan;
    FMX_HIP(fmx_dev_malloc((void **)&rec, (size_t)lv.nrec * 128));
    if (int rc = keep(idx, rec, (uint64_t)lv.nrec * 128)) return rc;
    FMX_HIP(fmx_dev_malloc((void **)&C, 16 * sizeof(uint32_t)));
    if (int rc = keep(idx, C, 64)) return rc;
    size_t nh = (size_t)ncode * lv.nrec;
    FMX_HIP(pool.get(&hist, nh));
    FMX_HIP(pool.get(&scan, nh));
    unsigned grid = nblocks((uint64_t)lv.nrec * 8);
    if (lv.fmt == 3)
      hipLaunchKernelGGL((k_mwm_pieces<3, T>), dim3(grid), dim3(BLK), 0, 0, cur, len, lv.shift, lv.mask,
                         lv.nrec, rec, hist);
    else
      hipLaunchKernelGGL((k_mwm_pieces<4, T>), dim3(grid), dim3(BLK), 0, 0, cur, len, lv.shift, lv.mask,
                         lv.nrec, rec, hist);
    size_t tb = 0;
    FMX_HIP(exclusive_sum(nullptr, tb, hist, scan, nh));
    uint8_t *tmp;
    FMX_HIP(pool.get(&tmp, tb));
    FMX_HIP(exclusive_sum(tmp, tb, hist, scan, nh));
    int fold_c = (l + 1 < nlv) ? 1 : 0;
    uint32_t *d_add = nullptr;
    if (nlv == 1 && single_level_add) {  // cs[] folded into the only level's counters
      uint32_t hadd[16] = {0};
      for (uint32_t c = 0; c < 16 && c < nadd; c++) hadd[c] = (uint32_t)single_level_add[c];
      FMX_HIP(pool.get(&d_add, 16));
      FMX_HIP(hipMemcpy(d_add, hadd, sizeof hadd, hipMemcpyHostToDevice));
    }
    if (lv.fmt == 3)
      hipLaunchKernelGGL(k_mwm_counters<3>, dim3(grid), dim3(BLK), 0, 0, scan, lv.nrec, rec, C, fold_c,
                         (const uint32_t *)d_add);
    else
      hipLaunchKernelGGL(k_mwm_counters<4>, dim3(grid), dim3(BLK), 0, 0, scan, lv.nrec, rec, C, fold_c,
                         (const uint32_t *)d_add);
    FMX_HIP(hipGetLastError());
    lv.rec = rec;
    lv.C = C;
    // hints for select (the forward / fl_map path) -- for sequences beyond a small build's size only: below it the
    // search over all record counters that fmx_level_select falls back to is at most 10 cached steps, and the three
    // launches + two allocations per level are a measurable share of a small build (n = 10^4, RLFM: 730 -> 542 us
    // together with the waits that went, benchmarks/gpu/r04_small_build2.sh)
    if (want_select && len > kArenaMaxN) {
      const uint32_t nsel = len / FMX_WSEL_STEP + 2u * ncode + 2u;
      uint32_t *sel, *meta;
      FMX_HIP(fmx_dev_malloc((void **)&sel, (size_t)nsel * 4));
      if (int rc = keep(idx, sel, (uint64_t)nsel * 4)) return rc;
      FMX_HIP(fmx_dev_malloc((void **)&meta, 48 * 4));
      if (int rc = keep(idx, meta, 48 * 4)) return rc;
      hipLaunchKernelGGL(k_fill_u32, dim3(nblocks(nsel)), dim3(BLK), 0, 0, sel, nsel, lv.nrec - 1);
      const unsigned hgrid = nblocks((uint64_t)lv.nrec * ncode);
      if (lv.fmt == 3) {
        hipLaunchKernelGGL(k_wsel_meta<3>, dim3(1), dim3(1), 0, 0, rec, scan, lv.nrec, len, meta);
        hipLaunchKernelGGL(k_wsel_hints<3>, dim3(hgrid), dim3(BLK), 0, 0, rec, lv.nrec, meta, sel);
      } else {
        hipLaunchKernelGGL(k_wsel_meta<4>, dim3(1), dim3(1), 0, 0, rec, scan, lv.nrec, len, meta);
        hipLaunchKernelGGL(k_wsel_hints<4>, dim3(hgrid), dim3(BLK), 0, 0, rec, lv.nrec, meta, sel);
      }
      FMX_HIP(hipGetLastError());
      lv.sel = sel;
      lv.selmeta = meta;
    }
    if (l + 1 < nlv) {
      // stable sort of the whole sequence by this level's code -> order of the next level
      size_t sb = 0;
      FMX_HIP(rocprim::radix_sort_keys(nullptr, sb, cur, alt, (size_t)len, lv.shift, lv.shift + bits[l],
                                       (hipStream_t)0));
      uint8_t *stmp;
      FMX_HIP(pool.get(&stmp, sb));
      FMX_HIP(rocprim::radix_sort_keys(stmp, sb, cur, alt, (size_t)len, lv.shift, lv.shift + bits[l],
                                       (hipStream_t)0));
      FMX_HIP(pool.quiesce());
      pool.release(stmp);
      T *x = cur; cur = alt; alt = x;
    }
    FMX_HIP(pool.quiesce());
    pool.release(hist); pool.release(scan); pool.release(tmp);
  }
  return FMX_OK;
}


// histogram of a device symbol array over 0..=maxc (host result), plus the TextStats
// (last non-zero index, maximum symbol) used by the validation
template <typename T>
int symbol_histogram(const T *d_sym, uint64_t count, uint32_t maxc, std::vector<uint64_t> &hist,
                     TextStats *st_out, DevPool &pool) {
  TextStats *d_st;
  FMX_HIP(pool.get(&d_st, 1));
  FMX_HIP(hipMemset(d_st, 0, sizeof(TextStats)));
  unsigned grid = nblocks(count, BLK * 16);
  if (grid > 4096) grid = 4096;
  if (count) hipLaunchKernelGGL(k_text_stats<T>, dim3(grid), dim3(BLK), 0, 0, d_sym, count, d_st);
  TextStats st;
  FMX_HIP(hipMemcpy(&st, d_st, sizeof st, hipMemcpyDeviceToHost));
  pool.release(d_st);
  hist.assign((size_t)maxc + 1, 0);
  if (maxc <= 255) {
    for (uint32_t c = 0; c <= maxc; c++) hist[c] = st.hist[c];
  } else {
    unsigned long long *d_h;
    FMX_HIP(pool.get(&d_h, (size_t)maxc + 1));
    FMX_HIP(hipMemset(d_h, 0, ((size_t)maxc + 1) * 8));
    if (count) hipLaunchKernelGGL(k_hist_global<T>, dim3(grid), dim3(BLK), 0, 0, d_sym, count, maxc, d_h);
    FMX_HIP(hipMemcpy(hist.data(), d_h, ((size_t)maxc + 1) * 8, hipMemcpyDeviceToHost));
    pool.release(d_h);
  }
  if (st_out) *st_out = st;
  return FMX_OK;
}

// FmxBits (rank/select records + select hints) from one flag byte per bit
// known_ones >= 0: the caller knows the number of ones (B and B' of an RLFM index: the number of runs) -- no read-back
int build_bits(fmx_index *idx, FmxBits *bv, const uint8_t *d_flags, uint32_t n, DevPool &pool, int64_t known_ones = -1) {
  memset(bv, 0, sizeof *bv);
  bv->len = n;
  bv->nrec = n / FMX_BITS_PER_REC + 1;  // position `len` itself must be addressable
  uint32_t npieces = bv->nrec * 8;
  uint4 *rec;
  uint32_t *cnt, *base;
  FMX_HIP(fmx_dev_malloc((void **)&rec, (size_t)bv->nrec * 128));
  if (int rc = keep(idx, rec, (uint64_t)bv->nrec * 128)) return rc;
  FMX_HIP(pool.get(&cnt, (size_t)npieces + 1));
  FMX_HIP(pool.get(&base, (size_t)npieces + 1));
  FMX_HIP(hipMemset(cnt, 0, ((size_t)npieces + 1) * 4));
  hipLaunchKernelGGL(k_bits_pieces, dim3(nblocks(npieces)), dim3(BLK), 0, 0, d_flags, n, npieces, rec,
                     cnt);
  size_t tb = 0;
  FMX_HIP(exclusive_sum(nullptr, tb, cnt, base, (size_t)npieces + 1));
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tb));
  FMX_HIP(exclusive_sum(tmp, tb, cnt, base, (size_t)npieces + 1));
  hipLaunchKernelGGL(k_bits_counters, dim3(nblocks(npieces)), dim3(BLK), 0, 0, base, npieces, rec);
  uint32_t ones = (uint32_t)known_ones;
  if (known_ones < 0) FMX_HIP(hipMemcpy(&ones, base + npieces, 4, hipMemcpyDeviceToHost));
#ifdef FMX_DEBUG_BOUNDS
  else {
    uint32_t counted = 0;
    FMX_HIP(hipMemcpy(&counted, base + npieces, 4, hipMemcpyDeviceToHost));
    if (counted != ones) { fmx_set_error(FMX_ERR_HIP, "bit vector: the caller's count of ones is wrong"); return FMX_ERR_HIP; }
  }
#endif
  bv->ones = ones;
  bv->nsel = ones / FMX_SEL_STEP + 2;
  uint32_t *sel;
  FMX_HIP(fmx_dev_malloc((void **)&sel, (size_t)bv->nsel * 4));
  if (int rc = keep(idx, sel, (uint64_t)bv->nsel * 4)) return rc;
  hipLaunchKernelGGL(k_fill_u32, dim3(nblocks(bv->nsel)), dim3(BLK), 0, 0, sel, bv->nsel, bv->nrec - 1);
  hipLaunchKernelGGL(k_select_hints, dim3(nblocks(bv->nrec)), dim3(BLK), 0, 0, rec, bv->nrec, ones, sel);
  FMX_HIP(hipGetLastError());
  FMX_HIP(pool.quiesce());
  bv->rec = rec;
  bv->sel = sel;
  pool.release(cnt); pool.release(base); pool.release(tmp);
  return FMX_OK;
}

// select blocks (FmxBits::dsel): one 16-byte block per 2^shift ones = { position of the first of them,
// the 96 bits from there }
__global__ __launch_bounds__(BLK) void k_select_blocks(const uint8_t *__restrict__ flags,
                                                        const uint32_t *__restrict__ pos,
                                                        uint32_t ones, uint32_t len, uint32_t shift,
                                                        uint4 *__restrict__ out) {
  const uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  const uint64_t per = 1ull << shift;
  const uint64_t nblk = ((uint64_t)ones + per - 1u) >> shift;
  if (j >= nblk) return;
  const uint32_t first = pos[j << shift];
  const uint64_t lastk = (j << shift) + per - 1u < ones ? (j << shift) + per - 1u : (uint64_t)ones - 1u;
  const uint32_t last = pos[lastk];
  if (last - first >= 96u) { out[j] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u); return; }
  uint32_t w[3] = {0u, 0u, 0u};
  for (uint32_t b = 0; b < 96u; b++) {
    const uint64_t p = (uint64_t)first + b;
    if (p < len && flags[p]) w[b >> 5] |= 1u << (b & 31u);
  }
  out[j] = make_uint4(first, w[0], w[1], w[2]);
}
// Every vector gets a one-load select: select blocks while a block's ones almost always fit its
// 96-bit window (ones per block chosen by density: span of the block ~ ones / density <= ~74 bits),
// the positions of the ones themselves below that.  Thresholds in 1/256ths of a one per bit.
static uint32_t select_block_shift(const FmxBits *bv) {
  const uint64_t d256 = bv->len ? (uint64_t)bv->ones * 256u / bv->len : 0;
  if (d256 >= 222) return 6;   // >= 0.87: 64 ones per block
  if (d256 >= 112) return 5;   // >= 0.44: 32
  if (d256 >= 56) return 4;    // >= 0.22: 16
  if (d256 >= 28) return 3;    // >= 0.11: 8
  return 0;                    // sparser: stored positions
}
int keep_dense_select(fmx_index *idx, FmxBits *bv, const uint8_t *d_flags, const uint32_t *d_pos) {
  bv->dsel = nullptr;
  bv->dsel_shift = 0;
  const uint32_t shift = select_block_shift(bv);
  if (bv->ones == 0 || shift == 0) return FMX_OK;
#ifdef FMX_MEASURE
  if (const char *v = getenv("FMX_VARIANT")) if (atoi(v) == 17) return FMX_OK;   // measurement: no blocks
#endif
  const uint64_t nblk = ((uint64_t)bv->ones + (1ull << shift) - 1u) >> shift;
  uint4 *d;
  FMX_HIP(fmx_dev_malloc((void **)&d, nblk * 16));
  if (int rc = keep(idx, d, nblk * 16)) return rc;
  hipLaunchKernelGGL(k_select_blocks, dim3(nblocks(nblk)), dim3(BLK), 0, 0, d_flags, d_pos, bv->ones,
                     bv->len, shift, d);
  FMX_HIP(hipGetLastError());
  bv->dsel = d;
  bv->dsel_shift = shift;
  return FMX_OK;
}

// sparse bit vector (< 0.11 ones per bit, i.e. runs of 9+ on average): keep the positions of its ones
// for one-load selects (4 bytes per one: < 0.44 bytes per bit of the vector)
int keep_positions(fmx_index *idx, FmxBits *bv, const uint32_t *d_pos) {
  bv->pos = nullptr;
  if (bv->ones == 0 || select_block_shift(bv) != 0) return FMX_OK;
#ifdef FMX_MEASURE
  if (const char *v = getenv("FMX_VARIANT")) if (atoi(v) == 16) return FMX_OK;   // measurement: no positions
#endif
  uint32_t *p;
  FMX_HIP(fmx_dev_malloc((void **)&p, (size_t)bv->ones * 4));
  if (int rc = keep(idx, p, (uint64_t)bv->ones * 4)) return rc;
  FMX_HIP(hipMemcpy(p, d_pos, (size_t)bv->ones * 4, hipMemcpyDeviceToDevice));
  bv->pos = p;
  return FMX_OK;
}

// RLFMIndexBackend::new (rlfmi.rs:30-96) from the L column (d_L is consumed)
template <typename T>
int build_rlfm(fmx_index *idx, T *d_L, uint32_t n, uint32_t L, DevPool &pool) {
  FmxDev &dv = idx->dev;
  const uint32_t maxc = (uint32_t)idx->max_character;
  uint8_t *flags;
  T *heads;
  uint32_t *starts, *d_num;
  FMX_HIP(pool.get(&flags, n));
  FMX_HIP(pool.get(&heads, n));
  FMX_HIP(pool.get(&starts, n));
  FMX_HIP(pool.get(&d_num, 1));
  hipLaunchKernelGGL(k_run_flags<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_L, n, flags);
  // S = run heads (rlfmi.rs:57), starts = first row of every run
  size_t t1 = 0, t2 = 0;
  rocprim::counting_iterator<uint32_t> rows(0);
  FMX_HIP(rocprim::select(nullptr, t1, d_L, flags, heads, d_num, (size_t)n, (hipStream_t)0));
  FMX_HIP(rocprim::select(nullptr, t2, rows, flags, starts, d_num, (size_t)n, (hipStream_t)0));
  size_t tb = t1 > t2 ? t1 : t2;
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tb));
  size_t tt = tb;
  FMX_HIP(rocprim::select(tmp, tt, d_L, flags, heads, d_num, (size_t)n, (hipStream_t)0));
  tt = tb;
  FMX_HIP(rocprim::select(tmp, tt, rows, flags, starts, d_num, (size_t)n, (hipStream_t)0));
  uint32_t r = 0;
  FMX_HIP(hipMemcpy(&r, d_num, 4, hipMemcpyDeviceToHost));
  idx->runs = r;
  pool.release(tmp);
  // B (rlfmi.rs:46, 58, 61, 85)
  if (int rc = build_bits(idx, &dv.b, flags, n, pool, (int64_t)r)) return rc;
  if (int rc = keep_positions(idx, &dv.b, starts)) return rc;        // run starts = the ones of B
  if (int rc = keep_dense_select(idx, &dv.b, flags, starts)) return rc;
  // cs[c] = number of runs whose head is < c (rlfmi.rs:72-76)
  std::vector<uint64_t> rcs;
  if (int rc = symbol_histogram<T>(heads, r, maxc, rcs, nullptr, pool)) return rc;
  {
    uint64_t acc = 0;
    for (uint32_t c = 0; c <= maxc; c++) { uint64_t v = rcs[c]; rcs[c] = acc; acc += v; }
  }
  // B' (rlfmi.rs:71-83): runs in (head, row) order, each 1 0^{len-1}
  uint32_t *order, *order2, *lens, *fpos;
  T *hk2;
  FMX_HIP(pool.get(&order, r));
  FMX_HIP(pool.get(&order2, r));
  FMX_HIP(pool.get(&lens, r));
  FMX_HIP(pool.get(&fpos, r));
  FMX_HIP(pool.get(&hk2, r));
  hipLaunchKernelGGL(k_iota, dim3(nblocks(r)), dim3(BLK), 0, 0, order, r);
  size_t sb = 0;
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, sb, heads, hk2, order, order2, (size_t)r, 0u, (unsigned)L,
                                    (hipStream_t)0));
  uint8_t *stmp;
  FMX_HIP(pool.get(&stmp, sb));
  FMX_HIP(rocprim::radix_sort_pairs(stmp, sb, heads, hk2, order, order2, (size_t)r, 0u, (unsigned)L,
                                    (hipStream_t)0));
  hipLaunchKernelGGL(k_sorted_run_lens, dim3(nblocks(r)), dim3(BLK), 0, 0, starts, order2, r, n, lens);
  size_t eb = 0;
  FMX_HIP(exclusive_sum(nullptr, eb, lens, fpos, (size_t)r));
  uint8_t *etmp;
  FMX_HIP(pool.get(&etmp, eb));
  FMX_HIP(exclusive_sum(etmp, eb, lens, fpos, (size_t)r));
  // lf_map of every run start (FmxDev::lfrun), for indexes that locate -- when the text is repetitive enough for the
  // table to be a small part of the index (r <= n / 4; FMX_FLAG_RUN_TABLE asks for it whatever r / n is: include/fmx.h)
  // and the device has room for 4 bytes per run four times over; FMX_FLAG_NO_WALK_RECORDS keeps it off.  An optional
  // accelerator: when its allocation fails the index is built without it.
  dv.lfrun = nullptr;
  if (idx->level_requested != FMX_NO_LOCATE && !(idx->flags & FMX_FLAG_NO_WALK_RECORDS) &&
      ((idx->flags & FMX_FLAG_RUN_TABLE) || (uint64_t)r * 4u <= (uint64_t)n)) {
    size_t free_b = 0, total_b = 0;
    if (fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 16ull * r) {
      uint32_t *d_lfrun = nullptr;
      const hipError_t le = fmx_dev_malloc((void **)&d_lfrun, (size_t)(r ? r : 1) * 4);
      if (le == hipSuccess) {
        if (int rc = keep(idx, d_lfrun, (uint64_t)r * 4)) return rc;
        hipLaunchKernelGGL(k_scatter_lfrun, dim3(nblocks(r)), dim3(BLK), 0, 0, order2, fpos, r, d_lfrun);
        dv.lfrun = d_lfrun;
      } else if (le == hipErrorOutOfMemory) {
        (void)hipGetLastError();
      } else {
        return fmx_hip_fail(le, "hipMalloc(run table)", __LINE__);
      }
    }
  }
  FMX_HIP(hipMemsetAsync(flags, 0, n, 0));
  hipLaunchKernelGGL(k_scatter_ones, dim3(nblocks(r)), dim3(BLK), 0, 0, fpos, r, flags);
  FMX_HIP(pool.quiesce());
  if (int rc = build_bits(idx, &dv.bp, flags, n, pool, (int64_t)r)) return rc;
  if (int rc = keep_positions(idx, &dv.bp, fpos)) return rc;         // F positions of the runs = the ones of B'
  if (int rc = keep_dense_select(idx, &dv.bp, flags, fpos)) return rc;
  FMX_HIP(pool.quiesce());                                   // flags / fpos are released below
  pool.release(order); pool.release(order2); pool.release(lens); pool.release(fpos);
  pool.release(hk2); pool.release(stmp); pool.release(etmp); pool.release(flags);
  pool.release(starts);
  // S as a multi-ary wavelet matrix over the r run heads (rlfmi.rs:69-70)
  if (int rc = build_mwm(idx, &dv.bw, heads, r, L, pool, rcs.data(), maxc + 1)) return rc;
  uint64_t *d_cs;
  uint32_t *d_K;
  FMX_HIP(pool.get(&d_cs, maxc + 1));
  FMX_HIP(hipMemcpy(d_cs, rcs.data(), (maxc + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  FMX_HIP(fmx_dev_malloc((void **)&d_K, (maxc + 1) * sizeof(uint32_t)));
  if (int rc = keep(idx, d_K, (maxc + 1) * 4)) return rc;
  if (int rc = fmx_launch_compute_K(dv.bw, d_cs, d_K, maxc, 0)) return rc;
  dv.K = d_K;
  {  // run-based C array as u32 for get_f / fl_map (rlfmi.rs:145-169)
    std::vector<uint32_t> c32((size_t)maxc + 1);
    for (uint32_t c = 0; c <= maxc; c++) c32[c] = (uint32_t)rcs[c];
    uint32_t *d_c32;
    FMX_HIP(fmx_dev_malloc((void **)&d_c32, ((size_t)maxc + 1) * 4));
    if (int rc = keep(idx, d_c32, ((uint64_t)maxc + 1) * 4)) return rc;
    FMX_HIP(hipMemcpy(d_c32, c32.data(), ((size_t)maxc + 1) * 4, hipMemcpyHostToDevice));
    dv.cs = d_c32;
  }
  (void)d_L;
  return FMX_OK;
}

}  // namespace

int fmx_verify_sa_impl(const fmx_index *idx, uint64_t *violations) {
  uint32_t n = (uint32_t)idx->n;
  *violations = 0;
  if (n == 0) return FMX_OK;
  uint32_t *mark;
  unsigned long long *bad;
  FMX_HIP(fmx_dev_malloc((void **)&mark, (size_t)n * 4));
  FMX_HIP(fmx_dev_malloc((void **)&bad, 8));
  FMX_HIP(hipMemset(mark, 0, (size_t)n * 4));
  FMX_HIP(hipMemset(bad, 0, 8));
  if (idx->sym_bytes == 1)
    hipLaunchKernelGGL(k_verify_sa<uint8_t>, dim3(nblocks(n)), dim3(BLK), 0, 0,
                       (const uint8_t *)idx->d_text, idx->d_sa, n, mark, bad);
  else if (idx->sym_bytes == 2)
    hipLaunchKernelGGL(k_verify_sa<uint16_t>, dim3(nblocks(n)), dim3(BLK), 0, 0,
                       (const uint16_t *)idx->d_text, idx->d_sa, n, mark, bad);
  else
    hipLaunchKernelGGL(k_verify_sa<uint32_t>, dim3(nblocks(n)), dim3(BLK), 0, 0,
                       (const uint32_t *)idx->d_text, idx->d_sa, n, mark, bad);
  hipLaunchKernelGGL(k_count_not_one, dim3(nblocks(n)), dim3(BLK), 0, 0, mark, n, bad);
  unsigned long long hb = 0;
  FMX_HIP(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  (void)hipFree(mark);
  (void)hipFree(bad);
  *violations = hb;
  return FMX_OK;
}

static void fmx_async_pool_trim(int dev);
hipError_t fmx_dev_malloc(void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) fmx_async_pool_trim(dev);   // what the library's stream-ordered pool retains
    scratch_drop_current();
    e = hipMalloc(p, bytes);
  }
  return e;
}
// Stream-ordered allocations of the launch paths (the rows array of the locate paths that keep one, the tile sums of
// fmx_offsets_dev) come from a pool the LIBRARY owns, one per device: it keeps up to 2 GiB of freed blocks instead of
// returning them to the driver at every synchronisation (the rows array of the next call then comes out of the pool)
// without touching the attributes of the device's default pool, which the host application and other libraries
// share.  hipMemPoolTrimTo gives the retained blocks back: when an allocation fails, and from
// fmx_release_scratch.  Where the runtime cannot create a pool the allocation falls back to the default pool as it is.
namespace {
struct AsyncPools {
  std::mutex mu;
  hipMemPool_t pool[kArenaDevices] = {};
  bool tried[kArenaDevices] = {};
};
AsyncPools &async_pools() {
  static AsyncPools *p = new AsyncPools;         // leaked on purpose: nothing is freed at process exit
  return *p;
}
hipMemPool_t async_pool_of(int dev) {
  if (dev < 0 || dev >= kArenaDevices) return nullptr;
  AsyncPools &ap = async_pools();
  std::lock_guard<std::mutex> lk(ap.mu);
  if (!ap.tried[dev]) {
    ap.tried[dev] = true;
    hipMemPoolProps props;
    memset(&props, 0, sizeof props);
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    uint64_t keep = 2ull << 30;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool &&
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) {
      ap.pool[dev] = pool;
    } else {
      (void)hipGetLastError();
      if (pool) (void)hipMemPoolDestroy(pool);
    }
  }
  return ap.pool[dev];
}
void async_pool_trim(int dev) {
  if (dev < 0 || dev >= kArenaDevices) return;
  AsyncPools &ap = async_pools();
  hipMemPool_t pool;
  { std::lock_guard<std::mutex> lk(ap.mu); pool = ap.pool[dev]; }
  if (pool && hipMemPoolTrimTo(pool, 0) != hipSuccess) (void)hipGetLastError();
}
}  // namespace
hipError_t fmx_dev_malloc_async(void **p, size_t bytes, hipStream_t st) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  hipMemPool_t pool = async_pool_of(dev);
  auto alloc = [&]() { return pool ? hipMallocFromPoolAsync(p, bytes, pool, st) : hipMallocAsync(p, bytes, st); };
  hipError_t e = alloc();
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    async_pool_trim(dev);
    scratch_drop_current();
    e = alloc();
  }
  return e;
}
static void fmx_async_pool_trim(int dev) { async_pool_trim(dev); }
hipError_t fmx_dev_mem_info(size_t *free_b, size_t *total_b) {
  hipError_t e = hipMemGetInfo(free_b, total_b);
  int dev = -1;
  if (e == hipSuccess && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kArenaDevices) {
    ScratchCache &sc = scratch_cache();
    std::lock_guard<std::mutex> lk(sc.mu);
    *free_b += sc.held[dev];                       // idle scratch goes back to the driver when an allocation needs it
  }
  if (e == hipSuccess && dev >= 0 && dev < kArenaDevices) {   // ... and so does what the library's stream-ordered pool retains
    AsyncPools &ap = async_pools();
    hipMemPool_t pool;
    { std::lock_guard<std::mutex> lk(ap.mu); pool = ap.pool[dev]; }
    uint64_t reserved = 0, used = 0;
    if (pool && hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &reserved) == hipSuccess &&
        hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used) == hipSuccess && reserved > used)
      *free_b += (size_t)(reserved - used);
    else
      (void)hipGetLastError();
  }
  return e;
}
// idle small-build buffers of every device, the cache of large-build temporaries and what the library's stream-ordered
// pools retain (include/fmx.h: fmx_release_scratch)
void fmx_release_build_scratch(void) {
  for (int d = 0; d < kArenaDevices; d++) async_pool_trim(d);
  ArenaPool &ap = arena_pool();
  std::vector<std::pair<int, uint8_t *>> drop;
  {
    std::lock_guard<std::mutex> lk(ap.mu);
    for (int d = 0; d < kArenaDevices; d++)
      for (int c = 0; c < 2; c++) {
        for (uint8_t *p : ap.idle[d][c]) drop.push_back({d, p});
        ap.idle[d][c].clear();
      }
  }
  ScratchCache &sc = scratch_cache();
  {
    std::lock_guard<std::mutex> lk(sc.mu);
    for (int d = 0; d < kArenaDevices; d++) {
      std::vector<ScratchCache::Blk> blks;
      scratch_drop(d, blks);
      for (auto &b : blks) drop.push_back({d, (uint8_t *)b.p});
    }
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (auto &dp : drop)
    if (hipSetDevice(dp.first) == hipSuccess) (void)hipFree(dp.second);
  if (prev >= 0) (void)hipSetDevice(prev);
}

int fmx_make_walk_records(fmx_index *idx) {
  if (!fmx_walk_eligible(idx)) return FMX_OK;
  FmxDev &dv = idx->dev;
  const uint32_t nwalk = dv.n / FMX_WALK_ROWS + 1u;
  const size_t ncnt = (size_t)FMX_WALK_MAX_CHARACTER * nwalk;
  uint4 *d_walk = nullptr;
  uint32_t *d_cnt = nullptr, *d_base = nullptr;
  void *d_tmp = nullptr;
  size_t tb = 0;
  hipError_t e = fmx_dev_malloc((void **)&d_walk, (size_t)nwalk * 128u);
  if (e == hipSuccess) e = fmx_dev_malloc((void **)&d_cnt, ncnt * 4);
  if (e == hipSuccess) e = fmx_dev_malloc((void **)&d_base, ncnt * 4);
  if (e == hipSuccess) e = exclusive_sum(nullptr, tb, d_cnt, d_base, ncnt);
  if (e == hipSuccess) e = fmx_dev_malloc(&d_tmp, tb ? tb : 8);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_walk_counts, dim3(nblocks(nwalk)), dim3(BLK), 0, 0, dv.bw.lv[0].rec, dv.phase, dv.n, dv.sa_level,
                       nwalk, d_cnt);
    e = exclusive_sum(d_tmp, tb, d_cnt, d_base, ncnt);
  }
  if (e == hipSuccess) {
    // row 0 holds SA = n - 1: when that is a multiple of 2^level it is a phase-0 row with no phase-1 row in front of it
    const uint32_t edge = ((dv.n - 1u) & ((1u << dv.sa_level) - 1u)) == 0u ? 1u : 0u;
    hipLaunchKernelGGL(k_walk_records, dim3((unsigned)(((uint64_t)nwalk * 8u + BLK - 1) / BLK)), dim3(BLK), 0, 0,
                       dv.bw.lv[0].rec, dv.phase, d_base, dv.n, dv.sa_level, dv.nsamples, nwalk, edge, d_walk);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (d_tmp) (void)hipFree(d_tmp);
  if (d_base) (void)hipFree(d_base);
  if (d_cnt) (void)hipFree(d_cnt);
  if (e != hipSuccess) {
    if (d_walk) (void)hipFree(d_walk);
    // an optional accelerator: without room for it the index walks through its phase pieces (fmx_walk_records()
    // reports what the index got) -- on a smaller or busier device than the one that saved a file, too
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return FMX_OK; }
    return fmx_hip_fail(e, "walk records", __LINE__);
  }
  if (int rc = fmx_keep(idx, d_walk, (uint64_t)nwalk * 128u)) { (void)hipFree(d_walk); return rc; }
  dv.walk = d_walk;
  return FMX_OK;
}

template <typename T>
static int build_impl_t(fmx_index *idx, const T *d_text) {
  auto t0 = std::chrono::steady_clock::now();
  // FMX_BUILD_TRACE=1: elapsed ms after each phase on stderr (each mark synchronises the device)
  static const bool trace = fmx_build_trace();
  auto mark = [&](const char *what) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[fmx build] %-18s %8.1f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  DevPool pool(idx->device, (idx->flags & FMX_FLAG_KEEP_SCRATCH) != 0);
  const uint32_t n = (uint32_t)idx->n;
  const uint32_t maxc = (uint32_t)idx->max_character;
  const uint32_t L = 32u - (uint32_t)__builtin_clz(maxc);  // text.rs:61-63
  // (FMX_FLAG_KEEP_SA hands the suffix array over to the index: it must be an allocation of its own)
  ArenaLease lease;     // declared after the pool: returned before the pool's own frees run
  if (n <= kArenaMaxN && !(idx->flags & FMX_FLAG_KEEP_SA)) {
    lease.take(idx->device, n);
    pool.use_arena(lease.p, lease.bytes);
  }

  // -- statistics + validation (sais.rs:115-139) --
  std::vector<uint64_t> hist;
  TextStats st;
  if (int rc = symbol_histogram<T>(d_text, n, maxc, hist, &st, pool)) return rc;
  if (st.max_sym > maxc) {  // count_chars would index out of bounds (sais.rs:18)
    fmx_set_error(FMX_ERR_SYMBOL_RANGE, "text symbol exceeds max_character");
    return FMX_ERR_SYMBOL_RANGE;
  }
  if (n >= 2) {  // lengths 0 and 1 bypass validation (sais.rs:121-126)
    const unsigned long long first = st.first_sym;
    if (first == 0) {
      fmx_set_error(FMX_ERR_TEXT_START_ZERO, nullptr);
      return FMX_ERR_TEXT_START_ZERO;
    }
    if (st.last_nonzero_plus1 != (unsigned long long)n - 1) {  // rposition == n-2
      fmx_set_error(FMX_ERR_TEXT_END_ZERO, nullptr);
      return FMX_ERR_TEXT_END_ZERO;
    }
  }
  // -- C array (sais.rs:9-32) --
  idx->h_cs = (uint64_t *)calloc((size_t)maxc + 1, sizeof(uint64_t));
  {
    uint64_t sum = 0;
    for (uint32_t c = 0; c <= maxc; c++) { idx->h_cs[c] = sum; sum += hist[c]; }
  }

  mark("stats");
  // -- suffix array --
  uint32_t *d_sa;
  FMX_HIP(pool.get(&d_sa, n));
  if (n) {
    if (int rc = suffix_sort<T>(d_text, n, L, d_sa, pool)) return rc;
  }

  mark("suffix sort");
  // -- SA samples (sample.rs:21-44) --
  FmxDev &dv = idx->dev;
  dv.n = n;
  dv.max_character = maxc;
  dv.kind = idx->kind;
  dv.sym_bytes = (uint32_t)sizeof(T);
  dv.sa_level = FMX_NO_LOCATE;
  if (idx->level_requested != FMX_NO_LOCATE && n > 0) {
    uint32_t level = idx->level_requested;
    if (level >= 32 || (uint64_t)n <= (1ull << level)) level = 0;  // sample.rs:28-31
    uint64_t nsamp = (((uint64_t)n - 1) >> level) + 1;             // sample.rs:33
    uint32_t *d_samp;
    // +4: the locate kernel reads the sample through an aligned 16-B chunk
    FMX_HIP(fmx_dev_malloc((void **)&d_samp, (nsamp + 4) * sizeof(uint32_t)));
    FMX_HIP(hipMemset(d_samp, 0, (nsamp + 4) * sizeof(uint32_t)));
    if (int rc = keep(idx, d_samp, nsamp * 4)) return rc;
    dv.phase = nullptr;
    // text-order sampling where one LF step costs several dependent requests (RLFM; FM / multi-pieces
    // over two or more wavelet levels): a walk is then SA[row] mod 2^level steps -- half the mean of
    // row-order sampling and no geometric tail -- at the price of two phase-piece reads.  One-level
    // indexes: the two extra reads cost more than the saved steps (an LF step is ONE request there) unless the
    // index also gets WALK RECORDS (fmx_internal.h; FM kind, u8 symbols, max_character <= 5, levels 1..3), which
    // carry the phases inside the records the walk reads anyway: 2.75 requests per hit at level 2 against 4 and a
    // geometric tail (config 3: 0.085 against 0.125 ms per 2^20 hits).  Round 4: that is the default for
    // such an index when the device has room (the two arrays add ~1.5 bytes per text symbol: they must fit four times
    // over in what is free now); FMX_FLAG_TEXT_ORDER / FMX_FLAG_ROW_ORDER override the choice either way,
    // FMX_FLAG_NO_WALK_RECORDS keeps the walk records off (include/fmx.h).
    const bool can_text = level >= 1 && level <= FMX_PHASE_MAX_LEVEL;
    bool text_order = can_text && (idx->kind == FMX_KIND_RLFM || L > 4);
    if (can_text && !text_order && idx->kind == FMX_KIND_FM && sizeof(T) == 1 && L <= 3 && maxc <= FMX_WALK_MAX_CHARACTER &&
        level <= FMX_WALK_MAX_LEVEL && !(idx->flags & FMX_FLAG_NO_WALK_RECORDS)) {
      size_t free_b = 0, total_b = 0;
      const uint64_t extra = ((uint64_t)n / FMX_WALK_ROWS + 1u) * 128u + ((uint64_t)n / (3u * (32u / level)) + 1u) * 16u;
      if (fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4u * extra) text_order = true;
    }
    if (idx->flags & FMX_FLAG_TEXT_ORDER) text_order = can_text;
    if (idx->flags & FMX_FLAG_ROW_ORDER) text_order = false;
    if (text_order) {
      // sample the rows whose SA value is a multiple of 2^level (same number of samples), in row order,
      // and keep every row's phase SA[row] mod 2^level with a rank over the phase-0 rows (fmx_internal.h)
      const uint32_t rpp = 3u * (32u / level);
      const uint32_t npieces = n / rpp + 1u;
      uint4 *d_phase;
      uint32_t *zeros, *zbase, *d_cnt;
      FMX_HIP(fmx_dev_malloc((void **)&d_phase, (size_t)npieces * 16));
      if (int rc = keep(idx, d_phase, (uint64_t)npieces * 16)) return rc;
      FMX_HIP(pool.get(&zeros, (size_t)npieces + 1));
      FMX_HIP(pool.get(&zbase, (size_t)npieces + 1));
      FMX_HIP(pool.get(&d_cnt, 1));
      hipLaunchKernelGGL(k_phase_pieces, dim3(nblocks(npieces)), dim3(BLK), 0, 0, d_sa, n, level, npieces,
                         d_phase, zeros);
      size_t tb = 0, sb = 0;
      FMX_HIP(exclusive_sum(nullptr, tb, zeros, zbase, (size_t)npieces));
      auto flags = rocprim::make_transform_iterator(d_sa, PhaseZero{(1u << level) - 1u});
      FMX_HIP(rocprim::select(nullptr, sb, d_sa, flags, d_samp, d_cnt, (size_t)n, (hipStream_t)0));
      uint8_t *tmp;
      FMX_HIP(pool.get(&tmp, tb > sb ? tb : sb));
      size_t t1 = tb;
      FMX_HIP(exclusive_sum(tmp, t1, zeros, zbase, (size_t)npieces));
      hipLaunchKernelGGL(k_phase_counts, dim3(nblocks(npieces)), dim3(BLK), 0, 0, zbase, npieces, d_phase);
      t1 = sb;
      FMX_HIP(rocprim::select(tmp, t1, d_sa, flags, d_samp, d_cnt, (size_t)n, (hipStream_t)0));
      uint32_t got = 0;
      FMX_HIP(hipMemcpy(&got, d_cnt, 4, hipMemcpyDeviceToHost));
      if (got != nsamp) {
        fmx_set_error(FMX_ERR_HIP, "text-order sampling: unexpected number of samples");
        return FMX_ERR_HIP;
      }
      pool.release(tmp); pool.release(d_cnt); pool.release(zbase); pool.release(zeros);
      dv.phase = d_phase;
    } else {
      hipLaunchKernelGGL(k_samples, dim3(nblocks(nsamp)), dim3(BLK), 0, 0, d_sa, nsamp, level, d_samp);
    }
    dv.samples = d_samp;
    dv.nsamples = (uint32_t)nsamp;
    dv.sa_level = level;
    idx->nsamples = nsamp;
  }

  mark("samples");
  // -- BWT (fm_index.rs:44-58) --
  T *d_bwt;
  FMX_HIP(pool.get(&d_bwt, n));
  if (n) hipLaunchKernelGGL(k_bwt<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_text, d_sa, n, d_bwt);
  FMX_HIP(hipGetLastError());

  if (idx->kind == FMX_KIND_MULTI && n > 0) {
    // doc[] before the BWT buffer is consumed by the wavelet builder
    const uint32_t pieces = (uint32_t)hist[0];
    uint32_t *zl, *zlr, *zt, *ztr, *d_doc, *d_first;
    FMX_HIP(pool.get(&zl, n));
    FMX_HIP(pool.get(&zlr, n));
    FMX_HIP(pool.get(&zt, n));
    FMX_HIP(pool.get(&ztr, n));
    FMX_HIP(pool.get(&d_first, 1));
    FMX_HIP(hipMemset(d_first, 0, 4));
    FMX_HIP(fmx_dev_malloc((void **)&d_doc, (size_t)(pieces ? pieces : 1) * 4));
    if (int rc = keep(idx, d_doc, (uint64_t)pieces * 4)) return rc;
    hipLaunchKernelGGL(k_zero_flags<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_bwt, n, zl);
    hipLaunchKernelGGL(k_zero_flags<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_text, n, zt);
    size_t tb = 0;
    FMX_HIP(exclusive_sum(nullptr, tb, zl, zlr, (size_t)n));
    uint8_t *tmp;
    FMX_HIP(pool.get(&tmp, tb));
    size_t t1 = tb;
    FMX_HIP(exclusive_sum(tmp, t1, zl, zlr, (size_t)n));
    t1 = tb;
    FMX_HIP(exclusive_sum(tmp, t1, zt, ztr, (size_t)n));
    hipLaunchKernelGGL(k_doc, dim3(nblocks(n)), dim3(BLK), 0, 0, d_sa, zl, zlr, ztr, n, pieces, d_doc,
                       d_first);
    uint32_t first = 0;
    FMX_HIP(hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost));
    dv.doc = d_doc;
    dv.doc_count = pieces;
    dv.first_row = first;
    pool.release(zl); pool.release(zlr); pool.release(zt); pool.release(ztr); pool.release(tmp);
    pool.release(d_first);
  }
  if (idx->kind == FMX_KIND_FM || idx->kind == FMX_KIND_MULTI) {
    FMX_HIP(pool.quiesce());
    if (int rc = build_mwm<T>(idx, &dv.bw, d_bwt, n, L, pool, idx->h_cs, maxc + 1)) return rc;
    // K[c] = cs[c] - S_c (all zero when cs[] was folded into a single level)
    uint64_t *d_cs;
    uint32_t *d_K;
    FMX_HIP(pool.get(&d_cs, (size_t)maxc + 1));
    FMX_HIP(hipMemcpy(d_cs, idx->h_cs, ((size_t)maxc + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
    FMX_HIP(fmx_dev_malloc((void **)&d_K, ((size_t)maxc + 1) * sizeof(uint32_t)));
    if (int rc = keep(idx, d_K, ((uint64_t)maxc + 1) * 4)) return rc;
    if (int rc = fmx_launch_compute_K(dv.bw, d_cs, d_K, maxc, 0)) return rc;
    dv.K = d_K;
    {  // C array as u32 for get_f / fl_map
      std::vector<uint32_t> c32((size_t)maxc + 1);
      for (uint32_t c = 0; c <= maxc; c++) c32[c] = (uint32_t)idx->h_cs[c];
      uint32_t *d_c32;
      FMX_HIP(fmx_dev_malloc((void **)&d_c32, ((size_t)maxc + 1) * 4));
      if (int rc = keep(idx, d_c32, ((uint64_t)maxc + 1) * 4)) return rc;
      FMX_HIP(hipMemcpy(d_c32, c32.data(), ((size_t)maxc + 1) * 4, hipMemcpyHostToDevice));
      dv.cs = d_c32;
    }
    // text-order sampling on a DNA-like index: the walk records the batched locate kernel reads (fmx_internal.h)
    dv.walk = nullptr;
    if (!(idx->flags & FMX_FLAG_NO_WALK_RECORDS))
      if (int rc = fmx_make_walk_records(idx)) return rc;
  } else {
    if (n < 2) {  // the reference hits unreachable!() (rlfmi.rs:62-65) / has nothing to index
      fmx_set_error(FMX_ERR_UNSUPPORTED, "RLFM index needs a text of at least 2 symbols");
      return FMX_ERR_UNSUPPORTED;
    }
    // c_i = T[SA[i]-1], or T[n-1] when SA[i] == 0 (rlfmi.rs:48-53)
    hipLaunchKernelGGL(k_bwt_cyclic<T>, dim3(nblocks(n)), dim3(BLK), 0, 0, d_text, d_sa, n, d_bwt);
    FMX_HIP(pool.quiesce());
    if (int rc = build_rlfm<T>(idx, d_bwt, n, L, pool)) return rc;
  }

  mark("rank records");
  // -- opt-in pair index (FMX_FLAG_PAIR_INDEX): sigma <= 4, the terminator is the only zero --
  if ((idx->flags & FMX_FLAG_PAIR_INDEX) && idx->kind == FMX_KIND_FM && sizeof(T) == 1 && maxc <= 4 &&
      n >= 4 && hist[0] == 1 && dv.bw.nlevels == 1) {
    // K2[c1c2] = LF(c1, LF(c2, 0)) = lf_map2(c1, cs[c2])   (fm_index.rs:93-95 applied twice)
    uint64_t hc[16], hi[16], k2[16];
    for (uint32_t code = 0; code < 16; code++) {
      uint32_t c1 = code / 4 + 1, c2 = code % 4 + 1;
      hc[code] = c1 <= maxc ? c1 : 1;
      hi[code] = c2 <= maxc ? idx->h_cs[c2] : 0;
    }
    uint64_t *d_q;
    FMX_HIP(pool.get(&d_q, 48));
    FMX_HIP(hipMemcpy(d_q, hc, sizeof hc, hipMemcpyHostToDevice));
    FMX_HIP(hipMemcpy(d_q + 16, hi, sizeof hi, hipMemcpyHostToDevice));
    if (int rc = fmx_launch_scalar(idx, 2, d_q, d_q + 16, 16, d_q + 32, 0)) return rc;
    FMX_HIP(hipMemcpy(k2, d_q + 32, sizeof k2, hipMemcpyDeviceToHost));
    uint8_t *d_b2;
    uint32_t *d_sp;
    FMX_HIP(pool.get(&d_b2, n));
    FMX_HIP(pool.get(&d_sp, 2));
    hipLaunchKernelGGL(k_bwt2, dim3(nblocks(n)), dim3(BLK), 0, 0, (const uint8_t *)d_text, d_sa, n, d_b2,
                       d_sp);
    uint32_t sp[2];
    FMX_HIP(hipMemcpy(sp, d_sp, sizeof sp, hipMemcpyDeviceToHost));
    FmxMwm pw;
    if (int rc = build_mwm<uint8_t>(idx, &pw, d_b2, n, 4, pool, k2, 16, false)) return rc;   // never selected
    dv.pair_rec = pw.lv[0].rec;
    dv.pair_row0 = sp[0];
    dv.pair_row1 = sp[1];
  }

  mark("pair index");
  // -- opt-in k-mer start table (FMX_FLAG_KMER_TABLE): u8 symbols; RLFM always, FM / multi-pieces
  // when the BWT has one wavelet level (with two levels k is <= 4 and the lookup costs more than
  // the cache-resident steps it replaces: benchmarks/kmer_table_sweep.py) --
  if ((idx->flags & FMX_FLAG_KMER_TABLE) && sizeof(T) == 1 && maxc >= 1 && n >= 2 &&
      (idx->kind == FMX_KIND_RLFM || dv.bw.nlevels == 1)) {
    uint32_t bits = 1;
    while ((1u << bits) < maxc) bits++;             // symbol c is coded c - 1 in 0..maxc-1
    uint32_t kk = 24u / bits;                        // <= 2^24 entries (128 MiB)
    if (kk > 16) kk = 16;                            // two symbols per lane of the 8-lane group
    while (kk > 0 && (1ull << (bits * kk)) > (uint64_t)n / 16u) kk--;   // table <= n/2 bytes
    if (kk >= 2) {
      uint2 *d_tab;
      const uint64_t tab_bytes = (1ull << (bits * kk)) * sizeof(uint2);
      FMX_HIP(fmx_dev_malloc((void **)&d_tab, tab_bytes));
      if (int rc = keep(idx, d_tab, tab_bytes)) return rc;
      if (int rc = fmx_launch_kmer_build(idx, d_tab, kk, bits, 0)) return rc;
      dv.kmer = d_tab;
      dv.kmer_k = kk;
      dv.kmer_bits = bits;
    }
  }

  mark("k-mer table");
  if (idx->flags & FMX_FLAG_KEEP_SA) {
    T *kt;
    FMX_HIP(fmx_dev_malloc((void **)&kt, (n ? n : 1) * sizeof(T)));
    if (n) FMX_HIP(hipMemcpy(kt, d_text, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice));
    if (int rc = keep(idx, kt, (uint64_t)n * sizeof(T))) return rc;
    idx->d_text = (uint8_t *)kt;
    // hand the SA over to the index instead of freeing it
    pool.disown(d_sa);
    if (int rc = keep(idx, d_sa, (uint64_t)n * 4)) return rc;
    idx->d_sa = d_sa;
  }
  FMX_HIP(hipDeviceSynchronize());
  idx->build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return FMX_OK;
}

// Character = u8 / u16 / u32 (character.rs:38-42); u64 texts are narrowed by the caller

// ===========================================================================================================
// WIDE indexes: n >= 2^32 - 16 (fmx_internal.h, FmxWideDev).  FMIndex / FMIndexWithLocate over a one-level
// alphabet.  Same construction as above with 64-bit suffix-array entries and ranks:
//   * prefix doubling cannot pack (rank[i], rank[i+h] + 1) into one 64-bit radix key any more (33-bit ranks), so a
//     round is two stable radix passes over (u64 key, u64 suffix) pairs -- by rank[i+h] + 1, then by rank[i] --
//     with the ranks in their own array: 40 bytes of scratch per symbol (24 in the 32-bit builder);
//   * the record counters are the exclusive scan over [code][record] in 64 bits -- which for a single level IS
//     cs[code] + rank, sais.rs:9-32 -- stored relative to the superblock start, the superblock starts in `base`.
// ===========================================================================================================
namespace {
// HIP refuses a launch whose grid x block exceeds 2^32 - 1 threads -- exactly what one thread per symbol would be
// here -- so every kernel over the symbols is a grid-stride loop on at most 2^22 blocks
#define KW_FOR(v, count) \
  for (uint64_t v = (uint64_t)blockIdx.x * BLK + threadIdx.x; v < (count); v += (uint64_t)gridDim.x * BLK)
inline unsigned wblocks(uint64_t n) {
  const uint64_t b = (n + BLK - 1) / BLK;
  return (unsigned)(b < 1 ? 1 : (b > (1u << 22) ? (1u << 22) : b));
}
template <typename T>
__global__ __launch_bounds__(BLK) void kw_init_keys(const T *__restrict__ t, uint64_t n, uint32_t bits,
                                                     uint32_t k, uint64_t *__restrict__ keys,
                                                     uint64_t *__restrict__ idx) {
  KW_FOR(i, n) {
    uint64_t key = 0;
    for (uint32_t j = 0; j < k; j++) {
      const uint64_t p = i + j;
      key = (key << bits) | (uint64_t)(p < n ? t[p] : 0);
    }
    keys[i] = key;
    idx[i] = i;
  }
}
// head[p] = p when sorted position p starts a new group of equal keys (ka, and kb when given), else 0
__global__ __launch_bounds__(BLK) void kw_flag_heads(const uint64_t *__restrict__ ka, const uint64_t *__restrict__ kb,
                                                      uint64_t n, uint64_t *__restrict__ head, unsigned int *dup) {
  KW_FOR(p, n) {
    const bool is_head = p == 0 || ka[p] != ka[p - 1] || (kb && kb[p] != kb[p - 1]);
    head[p] = is_head ? p : 0ull;
    if (!is_head) *dup = 1u;
  }
}
struct MaxOp64 {
  __device__ __forceinline__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};
__global__ __launch_bounds__(BLK) void kw_scatter_rank(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ head,
                                                        uint64_t n, uint64_t *__restrict__ rank) {
  KW_FOR(p, n) rank[sa[p]] = head[p];
}
// second = rank[i + h] + 1, or 0 when the suffix ends first; first = rank[i]
__global__ __launch_bounds__(BLK) void kw_key_second(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ rank,
                                                      uint64_t n, uint64_t h, uint64_t *__restrict__ keys) {
  KW_FOR(p, n) {
    const uint64_t j = sa[p] + h;
    keys[p] = j < n ? rank[j] + 1ull : 0ull;
  }
}
__global__ __launch_bounds__(BLK) void kw_key_first(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ rank,
                                                     uint64_t n, uint64_t *__restrict__ keys) {
  KW_FOR(p, n) keys[p] = rank[sa[p]];
}
// ---- refinement rounds (as k_refine_* of the 32-bit builder; two 64-bit keys per suffix, two stable passes) ----
__global__ __launch_bounds__(BLK) void kwr_active_flags(const uint64_t *__restrict__ grp, uint64_t n, uint8_t *__restrict__ flags) {
  KW_FOR(p, n) flags[p] = (grp[p] == p && (p + 1 == n || grp[p + 1] == p + 1)) ? 0 : 1;
}
__global__ __launch_bounds__(BLK) void kwr_active_flags_c(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ grp,
                                                           uint64_t m, uint8_t *__restrict__ flags) {
  KW_FOR(k, m) flags[k] = (grp[k] == apos[k] && (k + 1 == m || grp[k + 1] == apos[k + 1])) ? 0 : 1;
}
__global__ __launch_bounds__(BLK) void kwr_init(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ sa,
                                                 const uint64_t *__restrict__ rank, uint64_t n, uint64_t h, uint64_t m,
                                                 uint64_t *__restrict__ keys, uint64_t *__restrict__ vals) {
  KW_FOR(k, m) {
    const uint64_t i = sa[apos[k]], j = i + h;
    vals[k] = i;
    keys[k] = j < n ? rank[j] + 1ull : 0ull;
  }
}
__global__ __launch_bounds__(BLK) void kwr_first(const uint64_t *__restrict__ vals, const uint64_t *__restrict__ rank, uint64_t m,
                                                  uint64_t *__restrict__ keys) {
  KW_FOR(k, m) keys[k] = rank[vals[k]];
}
__global__ __launch_bounds__(BLK) void kwr_write(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ first,
                                                  const uint64_t *__restrict__ vals, const uint64_t *__restrict__ rank, uint64_t n,
                                                  uint64_t h, uint64_t m, uint64_t *__restrict__ sa, uint64_t *__restrict__ head) {
  KW_FOR(k, m) {
    const uint64_t i = vals[k];
    sa[apos[k]] = i;
    bool is_head = k == 0 || first[k] != first[k - 1];
    if (!is_head) {
      const uint64_t j = i + h, jp = vals[k - 1] + h;
      is_head = (j < n ? rank[j] + 1ull : 0ull) != (jp < n ? rank[jp] + 1ull : 0ull);
    }
    head[k] = is_head ? apos[k] : 0ull;
  }
}
__global__ __launch_bounds__(BLK) void kwr_rank(const uint64_t *__restrict__ vals, const uint64_t *__restrict__ grp, uint64_t m,
                                                 uint64_t *__restrict__ rank) {
  KW_FOR(k, m) rank[vals[k]] = grp[k];
}
// ---- the first refinement round from the text (as k_text_round_* of the 32-bit builder) ----
template <typename T>
__device__ __forceinline__ uint64_t kw_pack_key(const T *__restrict__ t, uint64_t n, uint64_t pos, uint32_t bits, uint32_t k) {
  uint64_t key = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint64_t p = pos + j;
    key = (key << bits) | (uint64_t)(p < n ? t[p] : 0);
  }
  return key;
}
__global__ __launch_bounds__(BLK) void kwt_active_flags_h(const uint64_t *__restrict__ head, uint64_t n, uint8_t *__restrict__ flags) {
  KW_FOR(p, n) {
    const bool h0 = p == 0 || head[p] != 0ull, h1 = p + 1 == n || head[p + 1] != 0ull;
    flags[p] = (h0 && h1) ? 0 : 1;
  }
}
__global__ __launch_bounds__(BLK) void kwt_heads(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ head, uint64_t m,
                                                  uint64_t *__restrict__ grp) {
  KW_FOR(k, m) {
    const uint64_t p = apos[k];
    grp[k] = (p == 0 || head[p] != 0ull) ? p : 0ull;
  }
}
template <typename T>
__global__ __launch_bounds__(BLK) void kwt_keys(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ sa,
                                                 const T *__restrict__ t, uint64_t n, uint32_t bits, uint32_t ksym, uint64_t h,
                                                 uint64_t m, uint64_t *__restrict__ keys, uint64_t *__restrict__ idx) {
  KW_FOR(k, m) {
    keys[k] = kw_pack_key(t, n, sa[apos[k]] + h, bits, ksym);
    idx[k] = k;
  }
}
__global__ __launch_bounds__(BLK) void kwt_group_keys(const uint64_t *__restrict__ idx, const uint64_t *__restrict__ grp, uint64_t m,
                                                       uint64_t *__restrict__ keys) {
  KW_FOR(k, m) keys[k] = grp[idx[k]];
}
__global__ __launch_bounds__(BLK) void kwt_gather(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ idx,
                                                   const uint64_t *__restrict__ sa, uint64_t m, uint64_t *__restrict__ suf) {
  KW_FOR(k, m) suf[k] = sa[apos[idx[k]]];
}
template <typename T>
__global__ __launch_bounds__(BLK) void kwt_write(const uint64_t *__restrict__ apos, const uint64_t *__restrict__ gkeys,
                                                  const uint64_t *__restrict__ suf, const T *__restrict__ t, uint64_t n,
                                                  uint32_t bits, uint32_t ksym, uint64_t h, uint64_t m, uint64_t *__restrict__ sa,
                                                  uint64_t *__restrict__ head) {
  KW_FOR(k, m) {
    const uint64_t i = suf[k];
    sa[apos[k]] = i;
    bool is_head = k == 0 || gkeys[k] != gkeys[k - 1];
    if (!is_head) is_head = kw_pack_key(t, n, i + h, bits, ksym) != kw_pack_key(t, n, suf[k - 1] + h, bits, ksym);
    head[k] = is_head ? apos[k] : 0ull;
  }
}
__global__ __launch_bounds__(BLK) void kwt_rank_identity(const uint64_t *__restrict__ sa, uint64_t n, uint64_t *__restrict__ rank) {
  KW_FOR(p, n) rank[sa[p]] = p;
}
struct AsU64 {
  __device__ __forceinline__ unsigned long long operator()(uint8_t f) const { return f; }
};
template <typename T>
__global__ __launch_bounds__(BLK) void kw_bwt(const T *__restrict__ t, const uint64_t *__restrict__ sa, uint64_t n,
                                               T *__restrict__ bwt) {
  KW_FOR(p, n) {
    const uint64_t k = sa[p];
    bwt[p] = k > 0 ? t[k - 1] : (T)0;   // fm_index.rs:50-55
  }
}
__global__ __launch_bounds__(BLK) void kw_samples(const uint64_t *__restrict__ sa, uint64_t nsamp, uint32_t level,
                                                   uint64_t *__restrict__ out) {
  KW_FOR(j, nsamp) out[j] = sa[j << level];   // sample.rs:35-37
}
// scan = exclusive sum (64-bit) over the per-record histograms laid out [code][record]: scan[c][r] = cs[c] +
// #{c before record r}.  Counter of a record = that value minus the value at its superblock's first record.
__global__ __launch_bounds__(BLK) void kw_counters(const uint64_t *__restrict__ scan, uint32_t nrec, uint32_t sb_recs,
                                                    uint4 *__restrict__ rec) {
  const uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t >= (uint64_t)nrec * 8) return;
  const uint32_t g = (uint32_t)(t & 7), r = (uint32_t)(t >> 3);
  const uint32_t r0 = (r >> sb_recs) << sb_recs;
  uint4 p = rec[t];
  p.x = (uint32_t)(scan[(size_t)g * nrec + r] - scan[(size_t)g * nrec + r0]);
  rec[t] = p;
}
__global__ void kw_bases(const uint64_t *__restrict__ scan, uint32_t nrec, uint32_t nsb, uint32_t sb_recs,
                         uint64_t *__restrict__ base) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nsb * 8u) return;
  const uint32_t sb = t >> 3, g = t & 7u;
  base[t] = scan[(size_t)g * nrec + ((size_t)sb << sb_recs)];
}
// the same for a level of the generic wide index (FmxWideLevel): scan laid out [code][record] over NCODE codes;
// a record's counters are relative to its superblock's first record, base[sb][code] is the absolute value there --
// on every level but the last including the entries with a smaller code (scan[code][0]), which makes a level's rank
// the position in the next level
template <int FMT>
__global__ __launch_bounds__(BLK) void kw_counters_g(const uint64_t *__restrict__ scan, uint32_t nrec, uint32_t sb_recs,
                                                      uint4 *__restrict__ rec) {
  const uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
  if (t >= (uint64_t)nrec * 8) return;
  const uint32_t g = (uint32_t)(t & 7), r = (uint32_t)(t >> 3);
  const uint32_t r0 = (r >> sb_recs) << sb_recs;
  uint4 p = rec[t];
  if (FMT == 3) {
    p.x = (uint32_t)(scan[(size_t)g * nrec + r] - scan[(size_t)g * nrec + r0]);
  } else {
    p.x = (uint32_t)(scan[(size_t)(2 * g) * nrec + r] - scan[(size_t)(2 * g) * nrec + r0]);
    p.y = (uint32_t)(scan[(size_t)(2 * g + 1) * nrec + r] - scan[(size_t)(2 * g + 1) * nrec + r0]);
  }
  rec[t] = p;
}
__global__ void kw_bases_g(const uint64_t *__restrict__ scan, uint32_t nrec, uint32_t nsb, uint32_t sb_recs,
                           uint32_t ncode, int fold, uint64_t *__restrict__ base) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nsb * 16u) return;
  const uint32_t sb = t >> 4, c = t & 15u;
  base[t] = c < ncode ? scan[(size_t)c * nrec + ((size_t)sb << sb_recs)] - (fold ? 0ull : scan[(size_t)c * nrec]) : 0ull;
}
template <typename T>
__global__ __launch_bounds__(BLK) void kw_verify_sa(const T *__restrict__ t, const uint64_t *__restrict__ sa,
                                                     uint64_t n, uint32_t *__restrict__ mark, unsigned long long *bad) {
  KW_FOR(p, n) {
    const uint64_t b = sa[p];
    if (b >= n) { atomicAdd(bad, 1ull); continue; }
    atomicAdd(&mark[b >> 2], 1u << (8u * (uint32_t)(b & 3u)));     // one byte per index, four to a word
    if (p == 0) continue;
    const uint64_t a = sa[p - 1];
    if (a >= n) continue;
    for (uint64_t j = 0;; j++) {   // suffix a must be < suffix b (a suffix that ends first is smaller)
      const uint64_t pa = a + j, pb = b + j;
      if (pa >= n) break;
      if (pb >= n) { atomicAdd(bad, 1ull); break; }
      const T ca = t[pa], cb = t[pb];
      if (ca < cb) break;
      if (ca > cb) { atomicAdd(bad, 1ull); break; }
    }
  }
}
__global__ __launch_bounds__(BLK) void kw_count_not_one(const uint8_t *mark, uint64_t n, unsigned long long *bad) {
  KW_FOR(p, n) if (mark[p] != 1u) atomicAdd(bad, 1ull);
}


// ---- walk records of a one-level wide index (FmxWideDev::walk, fmx_internal.h) ---------------------------------
// Built straight from the L column and the 64-bit suffix array (this engine has no phase pieces).  Counter arrays of a
// record, k = 0..10: rows with code k + 1 (k < 5); phase-0 rows (5); phase-1 rows with code k - 5 (6..10).
struct PhaseZero64 {
  uint64_t mask;
  __host__ __device__ bool operator()(uint64_t v) const { return (v & mask) == 0ull; }
};
#define KWW_COUNTERS 11u
__global__ __launch_bounds__(BLK) void kww_counts(const uint8_t *__restrict__ bwt, const uint64_t *__restrict__ sa, uint64_t n,
                                                   uint32_t level, uint32_t nwalk, uint32_t *__restrict__ cnt) {
  const uint64_t mask = (1ull << level) - 1ull;
  KW_FOR(j, nwalk) {
    uint32_t c[KWW_COUNTERS];
    for (uint32_t k = 0; k < KWW_COUNTERS; k++) c[k] = 0;
    for (uint32_t t = 0; t < FMX_WALK_ROWS; t++) {
      const uint64_t row = j * FMX_WALK_ROWS + t;
      if (row >= n) break;
      const uint32_t code = bwt[row];
      const uint64_t ph = sa[row] & mask;
      for (uint32_t k = 0; k < 5u; k++) {
        c[k] += code == k + 1u;
        c[6u + k] += (code == k + 1u) & (ph == 1ull);
      }
      c[5] += ph == 0ull;
    }
    for (uint32_t k = 0; k < KWW_COUNTERS; k++) cnt[(size_t)k * nwalk + j] = c[k];
  }
}
// wbase[superblock][k] = the absolute value of counter k at the superblock's first record (adj[k] turns the scan of
// array k into it: cs[] / the smaller symbols' phase-1 rows / the row-0 edge folded in, the array's own start taken out)
__global__ __launch_bounds__(64) void kww_bases(const uint64_t *__restrict__ scan, const uint64_t *__restrict__ adj,
                                                 uint32_t nwalk, uint32_t nwsb, uint32_t shift, uint64_t *__restrict__ wbase) {
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= nwsb * 16u) return;
  const uint32_t sb = t >> 4, k = t & 15u;
  const uint64_t j = (uint64_t)sb << shift;
  wbase[t] = (k < KWW_COUNTERS && j < nwalk) ? scan[(size_t)k * nwalk + j] + adj[k] : 0ull;
}
// thread = piece g of walk record j (layout: fmx_internal.h, FmxDev::walk), counters relative to the walk superblock
__global__ __launch_bounds__(BLK) void kww_records(const uint8_t *__restrict__ bwt, const uint64_t *__restrict__ sa, uint64_t n,
                                                    uint32_t level, const uint64_t *__restrict__ scan,
                                                    const uint64_t *__restrict__ adj, const uint64_t *__restrict__ wbase,
                                                    uint32_t nwalk, uint32_t shift, uint4 *__restrict__ out) {
  const uint64_t mask = (1ull << level) - 1ull;
  KW_FOR(tid, (uint64_t)nwalk * 8u) {
    const uint32_t j = (uint32_t)(tid >> 3), g = (uint32_t)(tid & 7u);
    const uint64_t *wb = wbase + (size_t)(j >> shift) * 16u;
    auto rel = [&](uint32_t k) { return (uint32_t)(scan[(size_t)k * nwalk + j] + adj[k] - wb[k]); };
    if (g == 7u) {
      out[tid] = make_uint4(rel(7), rel(8), rel(9), rel(10));
      continue;
    }
    uint32_t p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
    const uint64_t row0 = (uint64_t)j * FMX_WALK_ROWS + g * 16u;
    for (uint32_t t = 0; t < 16u; t++) {
      const uint64_t row = row0 + t;
      const uint32_t code = row < n ? bwt[row] : 0u;
      const uint32_t ph = row < n ? (uint32_t)(sa[row] & mask) : 1u;
      p0 |= (code & 1u) << t; p1 |= ((code >> 1) & 1u) << t; p2 |= ((code >> 2) & 1u) << t;
      q0 |= (ph & 1u) << t;   q1 |= ((ph >> 1) & 1u) << t;   q2 |= ((ph >> 2) & 1u) << t;
    }
    out[tid] = make_uint4(rel(g), p0 | (p1 << 16), p2 | (q0 << 16), q1 | (q2 << 16));
  }
}

// ---- RLFM on the wide engine (rlfmi.rs:30-96; FmxWideBits, FmxWideDev::lfrun) ----
template <typename T>
__global__ __launch_bounds__(BLK) void kwb_run_flags(const T *__restrict__ L, uint64_t n, uint8_t *__restrict__ flags) {
  KW_FOR(i, n) {
    const T prev = i ? L[i - 1] : (T)0;             // c0 starts at 0: a run begins wherever c != c0   rlfmi.rs:41, 56-59
    flags[i] = L[i] != prev ? 1 : 0;
  }
}
__global__ __launch_bounds__(BLK) void kwb_iota(uint64_t *out, uint64_t n) {
  KW_FOR(i, n) out[i] = i;
}
// length of the q-th run in (head, row) order
__global__ __launch_bounds__(BLK) void kwb_sorted_run_lens(const uint64_t *__restrict__ starts, const uint64_t *__restrict__ order,
                                                            uint64_t r, uint64_t n, uint64_t *__restrict__ lens) {
  KW_FOR(q, r) {
    const uint64_t k = order[q];
    lens[q] = (k + 1 < r ? starts[k + 1] : n) - starts[k];
  }
}
__global__ __launch_bounds__(BLK) void kwb_scatter_lfrun(const uint64_t *__restrict__ order, const uint64_t *__restrict__ fpos,
                                                          uint64_t r, uint64_t *__restrict__ lfrun) {
  KW_FOR(t, r) lfrun[order[t]] = fpos[t];
}
__global__ __launch_bounds__(BLK) void kwb_scatter_ones(const uint64_t *__restrict__ pos, uint64_t r, uint8_t *__restrict__ flags) {
  KW_FOR(q, r) flags[pos[q]] = 1;
}
// bit-vector records: one thread per 96-bit piece (payload + its popcount)
__global__ __launch_bounds__(BLK) void kwb_pieces(const uint8_t *__restrict__ flags, uint64_t n, uint64_t npieces,
                                                   uint4 *__restrict__ rec, uint32_t *__restrict__ cnt) {
  KW_FOR(t, npieces) {
    const uint64_t base = t * FMX_BITS_PER_PIECE;
    uint32_t w[3] = {0, 0, 0};
    for (uint32_t j = 0; j < FMX_BITS_PER_PIECE; j++) {
      const uint64_t p = base + j;
      if (p < n && flags[p]) w[j >> 5] |= 1u << (j & 31u);
    }
    rec[t] = make_uint4(0u, w[0], w[1], w[2]);
    cnt[t] = __popc(w[0]) + __popc(w[1]) + __popc(w[2]);
  }
}
// counts relative to the record's superblock; scan[] = exclusive 64-bit sums of the piece popcounts
__global__ __launch_bounds__(BLK) void kwb_counters(const uint64_t *__restrict__ scan, uint64_t npieces, uint32_t sb_shift,
                                                     uint4 *__restrict__ rec) {
  KW_FOR(t, npieces) {
    const uint64_t first = (((t >> 3) >> sb_shift) << sb_shift) * 8u;    // first piece of the superblock
    rec[t].x = (uint32_t)(scan[t] - scan[first]);
  }
}
__global__ __launch_bounds__(64) void kwb_bases(const uint64_t *__restrict__ scan, uint32_t nsb, uint32_t sb_shift,
                                                uint64_t *__restrict__ base) {
  const uint32_t sb = blockIdx.x * 64u + threadIdx.x;
  if (sb < nsb) base[sb] = scan[((uint64_t)sb << sb_shift) * 8u];
}
// sel[m / STEP] = record holding the m-th one, for every multiple m of STEP
__global__ __launch_bounds__(BLK) void kwb_select_hints(const uint64_t *__restrict__ scan, uint32_t nrec, uint64_t ones,
                                                         uint32_t *__restrict__ sel) {
  KW_FOR(r, nrec) {
    const uint64_t b0 = scan[r * 8u], b1 = r + 1 < nrec ? scan[(r + 1) * 8u] : ones;
    for (uint64_t m = (b0 + FMX_SEL_STEP - 1) / FMX_SEL_STEP * FMX_SEL_STEP; m < b1; m += FMX_SEL_STEP) sel[m / FMX_SEL_STEP] = (uint32_t)r;
  }
}
__global__ __launch_bounds__(BLK) void kwb_fill_u32(uint32_t *out, uint64_t n, uint32_t v) {
  KW_FOR(i, n) out[i] = v;
}
// text-order sampling of a wide RLFM index (FmxWideDev::phase): one thread packs the phases SA[row] mod 2^level of one
// piece and counts its phase-0 rows; rows past the end get phase 1 (k_phase_pieces with 64-bit rows)
__global__ __launch_bounds__(BLK) void kwp_pieces(const uint64_t *__restrict__ sa, uint64_t n, uint32_t level, uint64_t npieces,
                                                   uint4 *__restrict__ out, uint32_t *__restrict__ zeros) {
  const uint32_t fpw = 32u / level, rpp = 3u * fpw;
  const uint64_t mask = (1ull << level) - 1ull;
  KW_FOR(j, npieces) {
    uint32_t w[3] = {0u, 0u, 0u}, z = 0;
    for (uint32_t t = 0; t < rpp; t++) {
      const uint64_t row = j * rpp + t;
      const uint32_t ph = row < n ? (uint32_t)(sa[row] & mask) : 1u;
      z += ph == 0u;
      w[t / fpw] |= ph << ((t % fpw) * level);
    }
    out[j] = make_uint4(0u, w[0], w[1], w[2]);
    zeros[j] = z;
  }
}
__global__ __launch_bounds__(BLK) void kwp_counters(const uint64_t *__restrict__ scan, uint64_t npieces, uint32_t sb_shift,
                                                     uint4 *__restrict__ out) {
  KW_FOR(j, npieces) out[j].x = (uint32_t)(scan[j] - scan[(j >> sb_shift) << sb_shift]);
}
__global__ __launch_bounds__(64) void kwp_bases(const uint64_t *__restrict__ scan, uint32_t nsb, uint32_t sb_shift,
                                                uint64_t *__restrict__ base) {
  const uint32_t sb = blockIdx.x * 64u + threadIdx.x;
  if (sb < nsb) base[sb] = scan[(uint64_t)sb << sb_shift];
}
// ---- multi-pieces on the wide engine: doc[] and sa_idx_first_text (multi_pieces.rs:57-85) ----
template <typename T>
struct IsZeroSym {
  __device__ __forceinline__ bool operator()(T v) const { return v == (T)0; }
};
// for the k-th end marker of L (row zrows[k]): doc[k] = number of end markers before text position (SA[row] - 1) mod n
// = the index of that position among the ascending marker positions zpos[]; the row whose marker is the LAST of the
// text is sa_idx_first_text
__global__ __launch_bounds__(BLK) void kwm_doc(const uint64_t *__restrict__ sa, const uint64_t *__restrict__ zrows,
                                               const uint64_t *__restrict__ zpos, uint64_t pieces, uint64_t n,
                                               uint32_t *__restrict__ doc, unsigned long long *__restrict__ first_row) {
  KW_FOR(k, pieces) {
    const uint64_t p = zrows[k], v = sa[p];
    const uint64_t pos = v > 0 ? v - 1 : n - 1;     // modular_sub(sa[p], 1, n)
    uint64_t lo = 0, hi = pieces;                   // markers at positions < pos
    while (lo < hi) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (zpos[mid] < pos) lo = mid + 1; else hi = mid;
    }
    doc[k] = (uint32_t)lo;
    if (lo == pieces - 1) *first_row = p;
  }
}

template <typename T>
int suffix_sort_wide(const T *d_text, uint64_t n, uint32_t sym_bits, uint64_t *d_sa, DevPool &pool) {
  // scratch: the (key, suffix) double buffers of the radix sort -- 32 bytes per symbol with d_sa -- and, ONLY when some
  // suffixes are still tied after the text round (or the text has too many ties for it), 8 more for the ranks: random
  // DNA / byte texts never allocate them (round 3 held 40 bytes per symbol from the start: 172 GB at n = 2^32 + 2^20)
  uint64_t *keys_a, *keys_b, *vals_b, *rank = nullptr;
  auto need_rank = [&]() -> hipError_t { return rank ? hipSuccess : pool.get(&rank, n); };
  unsigned int *d_ng;
  static const bool trace = fmx_build_trace();
  auto ts0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what, uint64_t h) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[fmx build]   wide sort: %-10s h=%-10llu %8.1f ms\n", what, (unsigned long long)h,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
  };
  FMX_HIP(pool.get(&keys_a, n));
  FMX_HIP(pool.get(&keys_b, n));
  FMX_HIP(pool.get(&vals_b, n));
  FMX_HIP(pool.get(&d_ng, 1));
  uint32_t k = 63 / sym_bits;
  if (k > 32) k = 32;
  unsigned rank_bits = 1;                                 // bits of the largest key of a doubling pass: n
  while ((n >> rank_bits) != 0) rank_bits++;
  size_t tmp_sort = 0, tmp_scan = 0;
  {
    rocprim::double_buffer<uint64_t> kb(keys_a, keys_b);
    rocprim::double_buffer<uint64_t> vb(d_sa, vals_b);
    FMX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort, kb, vb, (size_t)n, 0u, 64u, (hipStream_t)0));
    FMX_HIP(rocprim::inclusive_scan(nullptr, tmp_scan, (uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)n, MaxOp64(),
                                    (hipStream_t)0));
  }
  const size_t tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tmp_bytes));
  mark("alloc", 0);
  const unsigned nb = wblocks(n);
  hipLaunchKernelGGL(kw_init_keys<T>, dim3(nb), dim3(BLK), 0, 0, d_text, n, sym_bits, k, keys_a, d_sa);
  FMX_HIP(hipGetLastError());
  uint64_t *keys_cur = keys_a, *keys_alt = keys_b, *sa_cur = d_sa, *sa_alt = vals_b;
  auto sort_pass = [&](unsigned end_bit) -> int {
    rocprim::double_buffer<uint64_t> kb(keys_cur, keys_alt);
    rocprim::double_buffer<uint64_t> vb(sa_cur, sa_alt);
    size_t tb = tmp_bytes;
    FMX_HIP(rocprim::radix_sort_pairs(tmp, tb, kb, vb, (size_t)n, 0u, end_bit, (hipStream_t)0));
    keys_cur = kb.current(); keys_alt = kb.alternate();
    sa_cur = vb.current();   sa_alt = vb.alternate();
    return FMX_OK;
  };
  if (int rc = sort_pass(k * sym_bits)) return rc;
  // the refinement rounds' rocPRIM calls: `tmp` when it is large enough, a buffer of their own otherwise
  unsigned long long *d_cnt;
  FMX_HIP(pool.get(&d_cnt, 1));
  auto scratch = [&](size_t need, uint8_t **out, bool *own) -> hipError_t {
    *own = need > tmp_bytes;
    if (!*own) { *out = tmp; return hipSuccess; }
    return pool.get(out, need);
  };
  auto count_flags = [&](const uint8_t *fl, uint64_t cnt, uint64_t *out) -> int {
    auto in = rocprim::make_transform_iterator(fl, AsU64());
    size_t need = 0;
    FMX_HIP(rocprim::reduce(nullptr, need, in, d_cnt, 0ull, (size_t)cnt, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    uint8_t *t; bool own;
    FMX_HIP(scratch(need, &t, &own));
    FMX_HIP(rocprim::reduce(t, need, in, d_cnt, 0ull, (size_t)cnt, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    unsigned long long got = 0;
    FMX_HIP(hipMemcpy(&got, d_cnt, sizeof got, hipMemcpyDeviceToHost));
    if (own) pool.release(t);
    *out = got;
    return FMX_OK;
  };
  auto compact = [&](auto in, const uint8_t *fl, uint64_t cnt, uint64_t *out, uint64_t expect) -> int {
    size_t need = 0;
    FMX_HIP(rocprim::select(nullptr, need, in, fl, out, d_cnt, (size_t)cnt, (hipStream_t)0));
    uint8_t *t; bool own;
    FMX_HIP(scratch(need, &t, &own));
    FMX_HIP(rocprim::select(t, need, in, fl, out, d_cnt, (size_t)cnt, (hipStream_t)0));
    unsigned long long got = 0;
    FMX_HIP(hipMemcpy(&got, d_cnt, sizeof got, hipMemcpyDeviceToHost));
    if (own) pool.release(t);
    if (got != expect) { fmx_set_error(FMX_ERR_HIP, "suffix sort: compaction lost count"); return FMX_ERR_HIP; }
    return FMX_OK;
  };
  auto refine = [&](uint64_t *apos, uint64_t m, uint64_t *sa, uint64_t h) -> int {
    uint64_t *ck_a, *ck_b, *cv_a, *cv_b, *grp, *apos2;
    uint8_t *fl;
    FMX_HIP(pool.get(&ck_a, m)); FMX_HIP(pool.get(&ck_b, m));
    FMX_HIP(pool.get(&cv_a, m)); FMX_HIP(pool.get(&cv_b, m));
    FMX_HIP(pool.get(&grp, m));  FMX_HIP(pool.get(&apos2, m));
    FMX_HIP(pool.get(&fl, m));
    while (m) {
      if (h >= n) {
        fmx_set_error(FMX_ERR_HIP, "suffix sort did not converge");
        return FMX_ERR_HIP;
      }
      const unsigned mb = wblocks(m);
      uint64_t *kc = ck_a, *ka = ck_b, *vc = cv_a, *va = cv_b;
      auto pass = [&](unsigned end_bit) -> int {
        rocprim::double_buffer<uint64_t> kb(kc, ka);
        rocprim::double_buffer<uint64_t> vb(vc, va);
        size_t need = 0;
        FMX_HIP(rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
        uint8_t *t; bool own;
        FMX_HIP(scratch(need, &t, &own));
        FMX_HIP(rocprim::radix_sort_pairs(t, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
        if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
        kc = kb.current(); ka = kb.alternate();
        vc = vb.current(); va = vb.alternate();
        return FMX_OK;
      };
      hipLaunchKernelGGL(kwr_init, dim3(mb), dim3(BLK), 0, 0, apos, sa, rank, n, h, m, kc, vc);
      if (int rc = pass(rank_bits + 1)) return rc;           // by rank[i + h] + 1 ...
      hipLaunchKernelGGL(kwr_first, dim3(mb), dim3(BLK), 0, 0, vc, rank, m, kc);
      if (int rc = pass(rank_bits)) return rc;               // ... then, stable, by rank[i]
      hipLaunchKernelGGL(kwr_write, dim3(mb), dim3(BLK), 0, 0, apos, kc, vc, rank, n, h, m, sa, grp);
      size_t tb2 = 0;
      FMX_HIP(rocprim::inclusive_scan(nullptr, tb2, grp, grp, (size_t)m, MaxOp64(), (hipStream_t)0));
      uint8_t *t; bool own;
      FMX_HIP(scratch(tb2, &t, &own));
      FMX_HIP(rocprim::inclusive_scan(t, tb2, grp, grp, (size_t)m, MaxOp64(), (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      hipLaunchKernelGGL(kwr_rank, dim3(mb), dim3(BLK), 0, 0, vc, grp, m, rank);
      hipLaunchKernelGGL(kwr_active_flags_c, dim3(mb), dim3(BLK), 0, 0, apos, grp, m, fl);
      FMX_HIP(hipGetLastError());
      uint64_t m2 = 0;
      if (int rc = count_flags(fl, m, &m2)) return rc;
      h *= 2;
      mark("refine", h);
      if (m2) {
        if (int rc = compact(apos, fl, m, apos2, m2)) return rc;
        uint64_t *x = apos; apos = apos2; apos2 = x;
      }
      m = m2;
    }
    FMX_HIP(hipDeviceSynchronize());
    pool.release(ck_a); pool.release(ck_b); pool.release(cv_a); pool.release(cv_b);
    pool.release(grp); pool.release(apos); pool.release(apos2); pool.release(fl);
    return FMX_OK;
  };
  auto text_round = [&](const uint8_t *fl0, const uint64_t *head0, uint64_t m, uint64_t *sa, uint64_t h) -> int {
    uint64_t *apos, *grp0, *tk_a, *tk_b, *ix_a, *ix_b, *suf, *grp;
    uint8_t *fl;
    FMX_HIP(pool.get(&apos, m)); FMX_HIP(pool.get(&grp0, m));
    if (int rc = compact(rocprim::counting_iterator<uint64_t>(0), fl0, n, apos, m)) return rc;
    const unsigned mb = wblocks(m);
    hipLaunchKernelGGL(kwt_heads, dim3(mb), dim3(BLK), 0, 0, apos, head0, m, grp0);
    FMX_HIP(hipDeviceSynchronize());
    pool.release(keys_a); pool.release(keys_b);               // flags lived there (heads in the suffix buffer's alternate)
    keys_a = keys_b = nullptr;
    FMX_HIP(pool.get(&tk_a, m)); FMX_HIP(pool.get(&tk_b, m));
    FMX_HIP(pool.get(&ix_a, m)); FMX_HIP(pool.get(&ix_b, m));
    FMX_HIP(pool.get(&suf, m));  FMX_HIP(pool.get(&grp, m));
    FMX_HIP(pool.get(&fl, m));
    uint8_t *t; bool own;
    auto scan_max = [&](uint64_t *a) -> int {
      size_t need = 0;
      FMX_HIP(rocprim::inclusive_scan(nullptr, need, a, a, (size_t)m, MaxOp64(), (hipStream_t)0));
      FMX_HIP(scratch(need, &t, &own));
      FMX_HIP(rocprim::inclusive_scan(t, need, a, a, (size_t)m, MaxOp64(), (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      return FMX_OK;
    };
    if (int rc = scan_max(grp0)) return rc;
    uint64_t *kc = tk_a, *ka = tk_b, *vc = ix_a, *va = ix_b;
    auto pass = [&](unsigned end_bit) -> int {
      rocprim::double_buffer<uint64_t> kb(kc, ka);
      rocprim::double_buffer<uint64_t> vb(vc, va);
      size_t need = 0;
      FMX_HIP(rocprim::radix_sort_pairs(nullptr, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
      FMX_HIP(scratch(need, &t, &own));
      FMX_HIP(rocprim::radix_sort_pairs(t, need, kb, vb, (size_t)m, 0u, end_bit, (hipStream_t)0));
      if (own) { FMX_HIP(hipDeviceSynchronize()); pool.release(t); }
      kc = kb.current(); ka = kb.alternate();
      vc = vb.current(); va = vb.alternate();
      return FMX_OK;
    };
    hipLaunchKernelGGL(kwt_keys<T>, dim3(mb), dim3(BLK), 0, 0, apos, sa, d_text, n, sym_bits, k, h, m, kc, vc);
    if (int rc = pass(k * sym_bits)) return rc;               // by the text key at depth h ...
    hipLaunchKernelGGL(kwt_group_keys, dim3(mb), dim3(BLK), 0, 0, vc, grp0, m, kc);
    if (int rc = pass(rank_bits)) return rc;                  // ... then, stable, by the group
    hipLaunchKernelGGL(kwt_gather, dim3(mb), dim3(BLK), 0, 0, apos, vc, sa, m, suf);
    hipLaunchKernelGGL(kwt_write<T>, dim3(mb), dim3(BLK), 0, 0, apos, kc, suf, d_text, n, sym_bits, k, h, m, sa, grp);
    if (int rc = scan_max(grp)) return rc;
    hipLaunchKernelGGL(kwr_active_flags_c, dim3(mb), dim3(BLK), 0, 0, apos, grp, m, fl);
    FMX_HIP(hipGetLastError());
    uint64_t m2 = 0;
    if (int rc = count_flags(fl, m, &m2)) return rc;
    mark("text round", m2);
    uint64_t *apos2 = nullptr;
    if (m2) {   // tied beyond 2h symbols: ranks for every suffix (position; a tied one its group's first position)
      FMX_HIP(pool.get(&apos2, m2));
      if (int rc = compact(apos, fl, m, apos2, m2)) return rc;
      FMX_HIP(need_rank());                                   // (the 16 n bytes of key buffers are gone by now)
      hipLaunchKernelGGL(kwt_rank_identity, dim3(nb), dim3(BLK), 0, 0, sa, n, rank);
      hipLaunchKernelGGL(kwr_rank, dim3(mb), dim3(BLK), 0, 0, suf, grp, m, rank);
      FMX_HIP(hipGetLastError());
    }
    FMX_HIP(hipDeviceSynchronize());
    pool.release(tk_a); pool.release(tk_b); pool.release(ix_a); pool.release(ix_b);
    pool.release(suf); pool.release(grp); pool.release(fl); pool.release(grp0); pool.release(apos);
    if (m2) return refine(apos2, m2, sa, 2 * h);
    return FMX_OK;
  };
  uint64_t h = k;
  bool two_keys = false;
  for (;;) {
    // the radix sort's alternate buffers are free between two sorts: second keys in keys_alt, heads in sa_alt
    uint64_t *head = sa_alt;
    if (two_keys) hipLaunchKernelGGL(kw_key_second, dim3(nb), dim3(BLK), 0, 0, sa_cur, rank, n, h / 2, keys_alt);
    FMX_HIP(hipMemsetAsync(d_ng, 0, sizeof(unsigned int), 0));
    hipLaunchKernelGGL(kw_flag_heads, dim3(nb), dim3(BLK), 0, 0, keys_cur, two_keys ? keys_alt : nullptr, n, head, d_ng);
    unsigned int dup = 0;
    FMX_HIP(hipMemcpy(&dup, d_ng, sizeof dup, hipMemcpyDeviceToHost));
    mark("round", h);
    if (!dup) break;
    if (h >= n) {
      fmx_set_error(FMX_ERR_HIP, "suffix sort did not converge");
      return FMX_ERR_HIP;
    }
    if (!two_keys) {
      // first round: when at most an eighth of the suffixes are tied, they are told apart to depth 2h by keys packed
      // from the text, and only if some are STILL tied does any rank get computed (see the 32-bit builder)
      uint8_t *flags = (uint8_t *)keys_cur;
      uint64_t m = 0;
      hipLaunchKernelGGL(kwt_active_flags_h, dim3(nb), dim3(BLK), 0, 0, head, n, flags);
      if (int rc = count_flags(flags, n, &m)) return rc;
      mark("tied", m);
      if (m * 8 <= n) {
        if (int rc = text_round(flags, head, m, sa_cur, h)) return rc;
        break;
      }
    }
    size_t tb = tmp_bytes;
    FMX_HIP(rocprim::inclusive_scan(tmp, tb, head, head, (size_t)n, MaxOp64(), (hipStream_t)0));
    FMX_HIP(need_rank());
    hipLaunchKernelGGL(kw_scatter_rank, dim3(nb), dim3(BLK), 0, 0, sa_cur, head, n, rank);
    {
      // how many suffixes are still tied?  (the keys are spent: their buffer takes the flags)
      uint8_t *flags = (uint8_t *)keys_cur;
      hipLaunchKernelGGL(kwr_active_flags, dim3(nb), dim3(BLK), 0, 0, head, n, flags);
      uint64_t m = 0;
      if (int rc = count_flags(flags, n, &m)) return rc;
      mark("active", m);
      if (m * 4 <= n) {
        // few enough: from here on only they are sorted, in buffers of their own (49 m bytes for the 16 n released)
        uint64_t *apos;
        FMX_HIP(pool.get(&apos, m));
        if (int rc = compact(rocprim::counting_iterator<uint64_t>(0), flags, n, apos, m)) return rc;
        FMX_HIP(hipDeviceSynchronize());
        pool.release(keys_a); pool.release(keys_b);
        keys_a = keys_b = nullptr;
        if (int rc = refine(apos, m, sa_cur, h)) return rc;
        break;
      }
    }
    // two stable passes: by rank[i + h] + 1, then by rank[i]
    hipLaunchKernelGGL(kw_key_second, dim3(nb), dim3(BLK), 0, 0, sa_cur, rank, n, h, keys_cur);
    if (int rc = sort_pass(rank_bits + 1)) return rc;
    hipLaunchKernelGGL(kw_key_first, dim3(nb), dim3(BLK), 0, 0, sa_cur, rank, n, keys_cur);
    FMX_HIP(hipGetLastError());
    if (int rc = sort_pass(rank_bits)) return rc;
    two_keys = true;
    h *= 2;
  }
  if (sa_cur != d_sa) FMX_HIP(hipMemcpyAsync(d_sa, sa_cur, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, 0));
  FMX_HIP(hipDeviceSynchronize());
  if (keys_a) pool.release(keys_a);
  if (keys_b) pool.release(keys_b);
  pool.release(vals_b);
  if (rank) pool.release(rank);
  pool.release(tmp); pool.release(d_ng); pool.release(d_cnt);
  return FMX_OK;
}
}  // namespace

// the levels of a generic wide index (FmxWideDev::lv / nsb / sb_shift) over a sequence of `len` symbols of L bits --
// the BWT of an FM index, the run heads of an RLFM index -- which is consumed (sorted level by level)
template <typename T>
static int build_wide_levels(fmx_index *idx, T *d_seq, uint64_t len, uint32_t L, DevPool &pool) {
  FmxWideDev &w = idx->wide;
  const uint64_t n = len;
  const uint32_t sb_shift = fmx_wide_n(idx->n) ? FMX_WIDE_SB_SHIFT : FMX_WIDE_SB_SHIFT_TEST;
  const uint32_t nsb = (uint32_t)(n >> sb_shift) + 1u;
  {
    uint32_t nlv, bits[FMX_MAX_LEVELS];
    split_levels(L, &nlv, bits);
    if (nlv > FMXW_MAX_LEVELS) {
      fmx_set_error(FMX_ERR_UNSUPPORTED, "wide index: more wavelet levels than FMXW_MAX_LEVELS");
      return FMX_ERR_UNSUPPORTED;
    }
    T *cur = d_seq, *alt = nullptr;
    if (nlv > 1) FMX_HIP(pool.get(&alt, n ? n : 1));
    uint32_t shift = L;
    w.generic = 1;
    w.nlevels = nlv;
    w.nsb = nsb;
    w.sb_shift = sb_shift;
    for (uint32_t l = 0; l < nlv; l++) {
      shift -= bits[l];
      FmxWideLevel &lv = w.lv[l];
      lv.fmt = bits[l] == 4 ? 4u : 3u;
      lv.shift = shift;
      lv.mask = (1u << bits[l]) - 1u;
      const uint32_t rec_shift = lv.fmt == 3 ? 8u : 7u, ncode = lv.fmt == 3 ? 8u : 16u;
      lv.nrec = (uint32_t)((n >> rec_shift) + 1u);           // +1: position n itself must be addressable
      uint4 *rec;
      uint64_t *base, *scan;
      uint32_t *hist;
      FMX_HIP(fmx_dev_malloc((void **)&rec, (size_t)lv.nrec * 128));
      if (int rc = keep(idx, rec, (uint64_t)lv.nrec * 128)) return rc;
      FMX_HIP(fmx_dev_malloc((void **)&base, (size_t)nsb * 16 * sizeof(uint64_t)));
      if (int rc = keep(idx, base, (uint64_t)nsb * 128)) return rc;
      const size_t nh = (size_t)ncode * lv.nrec;
      FMX_HIP(pool.get(&hist, nh));
      FMX_HIP(pool.get(&scan, nh));
      const unsigned grid = nblocks((uint64_t)lv.nrec * 8);
      if (lv.fmt == 3)
        hipLaunchKernelGGL((k_mwm_pieces<3, T>), dim3(grid), dim3(BLK), 0, 0, cur, n, lv.shift, lv.mask, lv.nrec, rec, hist);
      else
        hipLaunchKernelGGL((k_mwm_pieces<4, T>), dim3(grid), dim3(BLK), 0, 0, cur, n, lv.shift, lv.mask, lv.nrec, rec, hist);
      size_t tb = 0;
      FMX_HIP(exclusive_sum(nullptr, tb, hist, scan, nh));
      uint8_t *tmp;
      FMX_HIP(pool.get(&tmp, tb));
      FMX_HIP(exclusive_sum(tmp, tb, hist, scan, nh));
      if (lv.fmt == 3)
        hipLaunchKernelGGL(kw_counters_g<3>, dim3(grid), dim3(BLK), 0, 0, scan, lv.nrec, sb_shift - rec_shift, rec);
      else
        hipLaunchKernelGGL(kw_counters_g<4>, dim3(grid), dim3(BLK), 0, 0, scan, lv.nrec, sb_shift - rec_shift, rec);
      hipLaunchKernelGGL(kw_bases_g, dim3((nsb * 16 + 63) / 64), dim3(64), 0, 0, scan, lv.nrec, nsb, sb_shift - rec_shift,
                         ncode, l + 1 < nlv ? 1 : 0, base);
      FMX_HIP(hipGetLastError());
      lv.rec = rec;
      lv.base = base;
      if (l + 1 < nlv) {   // stable sort of the whole sequence by this level's code -> order of the next level
        size_t sb = 0;
        FMX_HIP(rocprim::radix_sort_keys(nullptr, sb, cur, alt, (size_t)n, lv.shift, lv.shift + bits[l], (hipStream_t)0));
        uint8_t *stmp;
        FMX_HIP(pool.get(&stmp, sb));
        FMX_HIP(rocprim::radix_sort_keys(stmp, sb, cur, alt, (size_t)n, lv.shift, lv.shift + bits[l], (hipStream_t)0));
        FMX_HIP(hipDeviceSynchronize());
        pool.release(stmp);
        T *x = cur; cur = alt; alt = x;
      }
      FMX_HIP(hipDeviceSynchronize());
      pool.release(hist); pool.release(scan); pool.release(tmp);
    }
    if (alt) pool.release(cur == d_seq ? alt : cur);   // the buffer that is not the caller's
  }
  return FMX_OK;
}

// FmxWideBits from one flag byte per bit; d_pos = the positions of its ones in order (kept when the vector is sparse)
static int build_bits_wide(fmx_index *idx, FmxWideBits *bv, const uint8_t *d_flags, uint64_t n, const uint64_t *d_pos,
                           DevPool &pool) {
  memset(bv, 0, sizeof *bv);
  bv->len = n;
  bv->nrec = (uint32_t)(n / FMX_BITS_PER_REC + 1);  // position `len` itself must be addressable
  bv->sb_shift = fmx_wide_n(idx->n) ? FMXW_BITS_SB_SHIFT : FMXW_BITS_SB_SHIFT_TEST;
  bv->nsb = ((bv->nrec - 1u) >> bv->sb_shift) + 1u;
  const uint64_t npieces = (uint64_t)bv->nrec * 8;
  uint4 *rec;
  uint32_t *cnt;
  uint64_t *scan, *base;
  FMX_HIP(fmx_dev_malloc((void **)&rec, (size_t)bv->nrec * 128));
  if (int rc = keep(idx, rec, (uint64_t)bv->nrec * 128)) return rc;
  FMX_HIP(fmx_dev_malloc((void **)&base, (size_t)bv->nsb * 8));
  if (int rc = keep(idx, base, (uint64_t)bv->nsb * 8)) return rc;
  FMX_HIP(pool.get(&cnt, (size_t)npieces + 1));
  FMX_HIP(pool.get(&scan, (size_t)npieces + 1));
  FMX_HIP(hipMemsetAsync(cnt + npieces, 0, 4, 0));
  hipLaunchKernelGGL(kwb_pieces, dim3(wblocks(npieces)), dim3(BLK), 0, 0, d_flags, n, npieces, rec, cnt);
  size_t tb = 0;
  FMX_HIP(exclusive_sum(nullptr, tb, cnt, scan, (size_t)npieces + 1));
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tb));
  FMX_HIP(exclusive_sum(tmp, tb, cnt, scan, (size_t)npieces + 1));
  hipLaunchKernelGGL(kwb_counters, dim3(wblocks(npieces)), dim3(BLK), 0, 0, scan, npieces, bv->sb_shift, rec);
  hipLaunchKernelGGL(kwb_bases, dim3((bv->nsb + 63u) / 64u), dim3(64), 0, 0, scan, bv->nsb, bv->sb_shift, base);
  uint64_t ones = 0;
  FMX_HIP(hipMemcpy(&ones, scan + npieces, 8, hipMemcpyDeviceToHost));
  bv->ones = ones;
  bv->nsel = ones / FMX_SEL_STEP + 2;
  uint32_t *sel;
  FMX_HIP(fmx_dev_malloc((void **)&sel, (size_t)bv->nsel * 4));
  if (int rc = keep(idx, sel, bv->nsel * 4)) return rc;
  hipLaunchKernelGGL(kwb_fill_u32, dim3(wblocks(bv->nsel)), dim3(BLK), 0, 0, sel, bv->nsel, bv->nrec - 1);
  hipLaunchKernelGGL(kwb_select_hints, dim3(wblocks(bv->nrec)), dim3(BLK), 0, 0, scan, bv->nrec, ones, sel);
  FMX_HIP(hipGetLastError());
  bv->rec = rec;
  bv->base = base;
  bv->sel = sel;
  // fewer than 0.11 ones per bit (runs of 9+ on average; the 32-bit engine's rule): keep the positions, select1 is one load
  if (ones && n && ones * 256u / n < 28u && d_pos) {
    uint64_t *p;
    FMX_HIP(fmx_dev_malloc((void **)&p, (size_t)ones * 8));
    if (int rc = keep(idx, p, ones * 8)) return rc;
    FMX_HIP(hipMemcpy(p, d_pos, (size_t)ones * 8, hipMemcpyDeviceToDevice));
    bv->pos = p;
  }
  FMX_HIP(hipDeviceSynchronize());
  pool.release(cnt); pool.release(scan); pool.release(tmp);
  return FMX_OK;
}

// RLFMIndexBackend::new (rlfmi.rs:30-96) on the wide engine, from the L column (d_L is consumed)
template <typename T>
static int build_rlfm_wide(fmx_index *idx, T *d_L, uint64_t n, uint32_t L, DevPool &pool) {
  FmxWideDev &w = idx->wide;
  const uint32_t maxc = (uint32_t)idx->max_character;
  uint8_t *flags;
  T *heads;
  uint64_t *starts;
  unsigned long long *d_num;
  FMX_HIP(pool.get(&flags, n));
  FMX_HIP(pool.get(&d_num, 1));
  hipLaunchKernelGGL(kwb_run_flags<T>, dim3(wblocks(n)), dim3(BLK), 0, 0, d_L, n, flags);
  // S = run heads (rlfmi.rs:57), starts = first row of every run: count first, so that the arrays are r long, not n
  size_t t1 = 0, t2 = 0;
  rocprim::counting_iterator<uint64_t> rows(0);
  FMX_HIP(rocprim::select(nullptr, t1, d_L, flags, d_L, d_num, (size_t)n, (hipStream_t)0));
  FMX_HIP(rocprim::select(nullptr, t2, rows, flags, (uint64_t *)nullptr, d_num, (size_t)n, (hipStream_t)0));
  uint64_t r = 0;
  {
    uint64_t *d_r;
    size_t rb = 0;
    FMX_HIP(pool.get(&d_r, 1));
    auto ones = rocprim::make_transform_iterator(flags, AsU64());
    FMX_HIP(rocprim::reduce(nullptr, rb, ones, (unsigned long long *)d_r, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    uint8_t *rtmp;
    FMX_HIP(pool.get(&rtmp, rb));
    FMX_HIP(rocprim::reduce(rtmp, rb, ones, (unsigned long long *)d_r, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), (hipStream_t)0));
    FMX_HIP(hipMemcpy(&r, d_r, 8, hipMemcpyDeviceToHost));
    pool.release(rtmp); pool.release(d_r);
  }
  idx->runs = r;
  w.slen = r;
  FMX_HIP(pool.get(&heads, r ? r : 1));
  FMX_HIP(pool.get(&starts, r ? r : 1));
  const size_t tb = t1 > t2 ? t1 : t2;
  uint8_t *tmp;
  FMX_HIP(pool.get(&tmp, tb));
  size_t tt = tb;
  FMX_HIP(rocprim::select(tmp, tt, d_L, flags, heads, d_num, (size_t)n, (hipStream_t)0));
  tt = tb;
  FMX_HIP(rocprim::select(tmp, tt, rows, flags, starts, d_num, (size_t)n, (hipStream_t)0));
  FMX_HIP(hipDeviceSynchronize());
  pool.release(tmp);
  // B (rlfmi.rs:46, 58, 61, 85)
  if (int rc = build_bits_wide(idx, &w.b, flags, n, starts, pool)) return rc;
  // cs[c] = number of runs whose head is < c (rlfmi.rs:72-76)
  std::vector<uint64_t> rcs;
  if (int rc = symbol_histogram<T>(heads, r, maxc, rcs, nullptr, pool)) return rc;
  {
    uint64_t acc = 0;
    for (uint32_t c = 0; c <= maxc; c++) { const uint64_t v = rcs[c]; rcs[c] = acc; acc += v; }
  }
  // B' (rlfmi.rs:71-83): runs in (head, row) order, each 1 0^{len-1}
  uint64_t *order, *order2, *lens, *fpos;
  T *hk2;
  FMX_HIP(pool.get(&order, r ? r : 1));
  FMX_HIP(pool.get(&order2, r ? r : 1));
  FMX_HIP(pool.get(&hk2, r ? r : 1));
  hipLaunchKernelGGL(kwb_iota, dim3(wblocks(r)), dim3(BLK), 0, 0, order, r);
  size_t sb = 0;
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, sb, heads, hk2, order, order2, (size_t)r, 0u, (unsigned)L, (hipStream_t)0));
  uint8_t *stmp;
  FMX_HIP(pool.get(&stmp, sb));
  FMX_HIP(rocprim::radix_sort_pairs(stmp, sb, heads, hk2, order, order2, (size_t)r, 0u, (unsigned)L, (hipStream_t)0));
  FMX_HIP(hipDeviceSynchronize());
  pool.release(stmp); pool.release(order); pool.release(hk2);
  FMX_HIP(pool.get(&lens, r ? r : 1));
  FMX_HIP(pool.get(&fpos, r ? r : 1));
  hipLaunchKernelGGL(kwb_sorted_run_lens, dim3(wblocks(r)), dim3(BLK), 0, 0, starts, order2, r, n, lens);
  size_t eb = 0;
  FMX_HIP(exclusive_sum(nullptr, eb, lens, fpos, (size_t)r));
  uint8_t *etmp;
  FMX_HIP(pool.get(&etmp, eb));
  FMX_HIP(exclusive_sum(etmp, eb, lens, fpos, (size_t)r));
  FMX_HIP(hipDeviceSynchronize());
  pool.release(etmp); pool.release(lens); pool.release(starts);
  // lf_map of every run start (FmxWideDev::lfrun) for indexes that locate: the 32-bit engine's policy (r <= n / 4, or
  // FMX_FLAG_RUN_TABLE; include/fmx.h) when the device has room for 8 bytes per run four times over;
  // FMX_FLAG_NO_WALK_RECORDS keeps it off; an allocation failure leaves the index without it
  w.lfrun = nullptr;
  if (idx->level_requested != FMX_NO_LOCATE && !(idx->flags & FMX_FLAG_NO_WALK_RECORDS) &&
      ((idx->flags & FMX_FLAG_RUN_TABLE) || (uint64_t)r * 4u <= (uint64_t)n)) {
    size_t free_b = 0, total_b = 0;
    if (fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 32ull * r) {
      uint64_t *d_lfrun = nullptr;
      const hipError_t le = fmx_dev_malloc((void **)&d_lfrun, (size_t)(r ? r : 1) * 8);
      if (le == hipSuccess) {
        if (int rc = keep(idx, d_lfrun, r * 8)) return rc;
        hipLaunchKernelGGL(kwb_scatter_lfrun, dim3(wblocks(r)), dim3(BLK), 0, 0, order2, fpos, r, d_lfrun);
        w.lfrun = d_lfrun;
      } else if (le == hipErrorOutOfMemory) {
        (void)hipGetLastError();
      } else {
        return fmx_hip_fail(le, "hipMalloc(run table)", __LINE__);
      }
    }
  }
  FMX_HIP(hipMemsetAsync(flags, 0, n, 0));
  hipLaunchKernelGGL(kwb_scatter_ones, dim3(wblocks(r)), dim3(BLK), 0, 0, fpos, r, flags);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  if (int rc = build_bits_wide(idx, &w.bp, flags, n, fpos, pool)) return rc;
  pool.release(order2); pool.release(fpos); pool.release(flags); pool.release(d_num);
  // S over the r run heads (rlfmi.rs:69-70): the generic levels; cs[] / K[] count runs
  if (int rc = build_wide_levels<T>(idx, heads, r, L, pool)) return rc;
  uint64_t *d_cs, *d_K;
  FMX_HIP(fmx_dev_malloc((void **)&d_cs, ((size_t)maxc + 1) * 8));
  if (int rc = keep(idx, d_cs, ((uint64_t)maxc + 1) * 8)) return rc;
  FMX_HIP(fmx_dev_malloc((void **)&d_K, ((size_t)maxc + 1) * 8));
  if (int rc = keep(idx, d_K, ((uint64_t)maxc + 1) * 8)) return rc;
  FMX_HIP(hipMemcpy(d_cs, rcs.data(), ((size_t)maxc + 1) * 8, hipMemcpyHostToDevice));
  w.cs = d_cs;
  w.K = d_K;
  w.kind = FMX_KIND_RLFM;
  if (int rc = fmxw_launch_compute_K(w, d_K)) return rc;
  FMX_HIP(hipDeviceSynchronize());
  pool.release(heads);
  (void)d_L;
  return FMX_OK;
}

template <typename T>
static int build_wide_t(fmx_index *idx, const T *d_text) {
  auto t0 = std::chrono::steady_clock::now();
  static const bool trace = fmx_build_trace();
  auto mark = [&](const char *what) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[fmx build wide] %-14s %8.1f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  DevPool pool(idx->device, (idx->flags & FMX_FLAG_KEEP_SCRATCH) != 0);
  const uint64_t n = idx->n;
  const uint32_t maxc = (uint32_t)idx->max_character;
  const uint32_t L = 32u - (uint32_t)__builtin_clz(maxc);   // text.rs:61-63
  // -- every allocation whose size is known from (n, level, alphabet) BEFORE the first kernel touches memory (round 5;
  // DevPool::use_slab says why): the temporaries' slab -- 32 bytes per symbol for the suffix sort's four 8-byte arrays,
  // which everything later fits into once they are given back -- and, for the one-level index, its own arrays
  uint64_t *pre_samp = nullptr, *pre_wbase = nullptr, *pre_base = nullptr;
  uint4 *pre_walk = nullptr, *pre_rec = nullptr;
  bool pre_walk_records = false;
  if (!(idx->flags & FMX_FLAG_KEEP_SA)) {               // (FMX_FLAG_KEEP_SA hands the suffix array over: an allocation of its own)
    (void)pool.use_slab((size_t)n * 32u + ((size_t)96u << 20));
    const bool one_level = idx->kind == FMX_KIND_FM && sizeof(T) == 1 && maxc <= 7;
    if (one_level && pool.slab) {
      if (idx->level_requested != FMX_NO_LOCATE) {
        uint32_t level = idx->level_requested;
        if (level >= 63 || n <= (1ull << level)) level = 0;
        const uint64_t nsamp = ((n - 1) >> level) + 1;
        FMX_HIP(fmx_dev_malloc((void **)&pre_samp, nsamp * sizeof(uint64_t)));
        if (int rc = keep(idx, pre_samp, nsamp * 8)) return rc;
        pre_walk_records = maxc <= FMX_WALK_MAX_CHARACTER && level >= 1 && level <= FMX_WALK_MAX_LEVEL &&
                           !(idx->flags & (FMX_FLAG_ROW_ORDER | FMX_FLAG_NO_WALK_RECORDS));
        if (pre_walk_records && !(idx->flags & FMX_FLAG_TEXT_ORDER)) {
          size_t free_b = 0, total_b = 0;
          const uint64_t extra = (n / FMX_WALK_ROWS + 1u) * 128u;
          pre_walk_records = fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4u * extra;
        }
        if (pre_walk_records) {
          const uint32_t nwalk = (uint32_t)(n / FMX_WALK_ROWS + 1u);
          const uint32_t shift = fmx_wide_n(n) ? FMXW_WALK_SB_SHIFT : FMXW_WALK_SB_SHIFT_TEST;
          const uint32_t nwsb = ((nwalk - 1u) >> shift) + 1u;
          FMX_HIP(fmx_dev_malloc((void **)&pre_walk, (size_t)nwalk * 128));
          if (int rc = keep(idx, pre_walk, (uint64_t)nwalk * 128)) return rc;
          FMX_HIP(fmx_dev_malloc((void **)&pre_wbase, (size_t)nwsb * 16 * sizeof(uint64_t)));
          if (int rc = keep(idx, pre_wbase, (uint64_t)nwsb * 128)) return rc;
        }
      }
      const uint32_t sbs = fmx_wide_n(n) ? FMX_WIDE_SB_SHIFT : FMX_WIDE_SB_SHIFT_TEST;
      const uint32_t nrec0 = (uint32_t)(n / 256u + 1u), nsb0 = (uint32_t)(n >> sbs) + 1u;
      FMX_HIP(fmx_dev_malloc((void **)&pre_rec, (size_t)nrec0 * 128));
      if (int rc = keep(idx, pre_rec, (uint64_t)nrec0 * 128)) return rc;
      FMX_HIP(fmx_dev_malloc((void **)&pre_base, (size_t)nsb0 * 8 * sizeof(uint64_t)));
      if (int rc = keep(idx, pre_base, (uint64_t)nsb0 * 64)) return rc;
    }
  }
  // -- statistics + validation (sais.rs:115-139) --
  std::vector<uint64_t> hist;
  TextStats st;
  if (int rc = symbol_histogram<T>(d_text, n, maxc, hist, &st, pool)) return rc;
  if (st.max_sym > maxc) {
    fmx_set_error(FMX_ERR_SYMBOL_RANGE, "text symbol exceeds max_character");
    return FMX_ERR_SYMBOL_RANGE;
  }
  if (st.first_sym == 0) {
    fmx_set_error(FMX_ERR_TEXT_START_ZERO, nullptr);
    return FMX_ERR_TEXT_START_ZERO;
  }
  if (st.last_nonzero_plus1 != (unsigned long long)n - 1) {   // rposition == n - 2
    fmx_set_error(FMX_ERR_TEXT_END_ZERO, nullptr);
    return FMX_ERR_TEXT_END_ZERO;
  }
  idx->h_cs = (uint64_t *)calloc((size_t)maxc + 1, sizeof(uint64_t));   // sais.rs:9-32
  {
    uint64_t sum = 0;
    for (uint32_t c = 0; c <= maxc; c++) { idx->h_cs[c] = sum; sum += hist[c]; }
  }
  mark("stats");
  // -- suffix array --
  uint64_t *d_sa;
  FMX_HIP(pool.get(&d_sa, n));
  if (int rc = suffix_sort_wide<T>(d_text, n, L, d_sa, pool)) return rc;
  mark("suffix sort");
  FmxWideDev &w = idx->wide;
  w.n = n;
  w.sym_bytes = (uint32_t)sizeof(T);
  w.max_character = maxc;
  w.status = idx->dev.status;
  w.sa_level = FMX_NO_LOCATE;
  idx->dev.sa_level = FMX_NO_LOCATE;
  idx->dev.kind = idx->kind;
  idx->dev.sym_bytes = (uint32_t)sizeof(T);
  // -- SA samples (sample.rs:21-44) --
  // One-level indexes (max_character <= 5, levels 1..3) sample in TEXT order and get walk records, like the 32-bit
  // engine's DNA indexes (fmx_internal.h): the batched walk is max(phase, 1) records + 1 sample per hit and never longer
  // than 2^level - 1 steps.  FMX_FLAG_ROW_ORDER / FMX_FLAG_NO_WALK_RECORDS keep the reference's rows (the round-3 shape).
  bool walk_records = false;
  w.walk = nullptr; w.wbase = nullptr; w.nwsb = 0; w.wsb_shift = 0;
  if (idx->level_requested != FMX_NO_LOCATE) {
    uint32_t level = idx->level_requested;
    if (level >= 63 || n <= (1ull << level)) level = 0;            // sample.rs:28-31
    const uint64_t nsamp = ((n - 1) >> level) + 1;                 // sample.rs:33
    uint64_t *d_samp = pre_samp;
    if (!d_samp) {
      FMX_HIP(fmx_dev_malloc((void **)&d_samp, nsamp * sizeof(uint64_t)));
      if (int rc = keep(idx, d_samp, nsamp * 8)) return rc;
    }
    walk_records = idx->kind == FMX_KIND_FM && sizeof(T) == 1 && maxc <= FMX_WALK_MAX_CHARACTER && level >= 1 && level <= FMX_WALK_MAX_LEVEL &&
                   !(idx->flags & (FMX_FLAG_ROW_ORDER | FMX_FLAG_NO_WALK_RECORDS));
    if (pre_samp) {
      walk_records = pre_walk_records;                             // decided (and allocated) before the first touch
    } else if (walk_records && !(idx->flags & FMX_FLAG_TEXT_ORDER)) {     // by default only when the device has room
      size_t free_b = 0, total_b = 0;
      const uint64_t extra = (n / FMX_WALK_ROWS + 1u) * 128u;
      walk_records = fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4u * extra;
    }
    // RLFM: text-order samples (phase pieces) for levels 1..4, together with the run table -- unless the flags keep the
    // reference's rows, or the device lacks room for the pieces four times over
    // (generic FM / multi-pieces indexes -- the multi-level kernels -- sample the same way: half the LF steps, none
    // longer than 2^level - 1, four walks per group in flight)
    const bool generic_index = idx->kind != FMX_KIND_FM || maxc > 7 || sizeof(T) != 1;
    bool rl_text = generic_index && level >= 1 && level <= FMX_PHASE_MAX_LEVEL &&
                   !(idx->flags & (FMX_FLAG_ROW_ORDER | FMX_FLAG_NO_WALK_RECORDS));
    if (rl_text && !(idx->flags & FMX_FLAG_TEXT_ORDER)) {
      size_t free_b = 0, total_b = 0;
      const uint64_t extra = (n / (3u * (32u / level)) + 1u) * 16u;
      rl_text = fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4u * extra;
    }
    w.phase = nullptr; w.pbase = nullptr; w.psb_shift = 0; w.npsb = 0;
    if (rl_text) {
      const uint32_t rpp = 3u * (32u / level);
      const uint64_t npieces = n / rpp + 1u;
      const uint32_t shift = fmx_wide_n(n) ? FMXW_PHASE_SB_SHIFT : FMXW_PHASE_SB_SHIFT_TEST;
      const uint32_t npsb = (uint32_t)((npieces - 1u) >> shift) + 1u;
      uint4 *d_phase;
      uint64_t *d_pbase, *pscan;
      uint32_t *zeros;
      FMX_HIP(fmx_dev_malloc((void **)&d_phase, (size_t)npieces * 16));
      if (int rc = keep(idx, d_phase, npieces * 16)) return rc;
      FMX_HIP(fmx_dev_malloc((void **)&d_pbase, (size_t)npsb * 8));
      if (int rc = keep(idx, d_pbase, (uint64_t)npsb * 8)) return rc;
      FMX_HIP(pool.get(&zeros, (size_t)npieces));
      FMX_HIP(pool.get(&pscan, (size_t)npieces));
      hipLaunchKernelGGL(kwp_pieces, dim3(wblocks(npieces)), dim3(BLK), 0, 0, d_sa, n, level, npieces, d_phase, zeros);
      size_t tb = 0;
      FMX_HIP(exclusive_sum(nullptr, tb, zeros, pscan, (size_t)npieces));
      uint8_t *ptmp;
      FMX_HIP(pool.get(&ptmp, tb));
      FMX_HIP(exclusive_sum(ptmp, tb, zeros, pscan, (size_t)npieces));
      hipLaunchKernelGGL(kwp_counters, dim3(wblocks(npieces)), dim3(BLK), 0, 0, pscan, npieces, shift, d_phase);
      hipLaunchKernelGGL(kwp_bases, dim3((npsb + 63u) / 64u), dim3(64), 0, 0, pscan, npsb, shift, d_pbase);
      FMX_HIP(hipGetLastError());
      FMX_HIP(hipDeviceSynchronize());
      pool.release(ptmp); pool.release(pscan); pool.release(zeros);
      w.phase = d_phase; w.pbase = d_pbase; w.psb_shift = shift; w.npsb = npsb;
    }
    if (walk_records || rl_text) {
      // the rows whose SA value is a multiple of 2^level (as many as the reference samples), in row order
      unsigned long long *d_got;
      FMX_HIP(pool.get(&d_got, 1));
      auto flags = rocprim::make_transform_iterator(d_sa, PhaseZero64{(1ull << level) - 1ull});
      size_t sb = 0;
      FMX_HIP(rocprim::select(nullptr, sb, d_sa, flags, d_samp, d_got, (size_t)n, (hipStream_t)0));
      uint8_t *stmp;
      FMX_HIP(pool.get(&stmp, sb));
      FMX_HIP(rocprim::select(stmp, sb, d_sa, flags, d_samp, d_got, (size_t)n, (hipStream_t)0));
      unsigned long long got = 0;
      FMX_HIP(hipMemcpy(&got, d_got, sizeof got, hipMemcpyDeviceToHost));
      pool.release(stmp); pool.release(d_got);
      if (got != nsamp) {
        fmx_set_error(FMX_ERR_HIP, "text-order sampling: unexpected number of samples");
        return FMX_ERR_HIP;
      }
    } else {
      hipLaunchKernelGGL(kw_samples, dim3(wblocks(nsamp)), dim3(BLK), 0, 0, d_sa, nsamp, level, d_samp);
    }
    w.samples = d_samp;
    w.sa_level = level;
    idx->dev.sa_level = level;
    idx->nsamples = nsamp;
  }
  mark("samples");
  // -- BWT (fm_index.rs:44-58) --
  T *d_bwt;
  FMX_HIP(pool.get(&d_bwt, n));
  hipLaunchKernelGGL(kw_bwt<T>, dim3(wblocks(n)), dim3(BLK), 0, 0, d_text, d_sa, n, d_bwt);
  FMX_HIP(hipGetLastError());
  if (walk_records) {
    if constexpr (sizeof(T) == 1) {
      const uint32_t level = w.sa_level;
      const uint32_t nwalk = (uint32_t)(n / FMX_WALK_ROWS + 1u);
      const uint32_t shift = fmx_wide_n(n) ? FMXW_WALK_SB_SHIFT : FMXW_WALK_SB_SHIFT_TEST;
      const uint32_t nwsb = ((nwalk - 1u) >> shift) + 1u;
      const size_t ncnt = (size_t)KWW_COUNTERS * nwalk + 1;          // one more: the scan's last entry = the grand total
      uint32_t *d_cnt;
      uint64_t *d_scan, *d_adj, *d_wbase;
      uint4 *d_walk;
      FMX_HIP(pool.get(&d_cnt, ncnt));
      FMX_HIP(pool.get(&d_scan, ncnt));
      FMX_HIP(pool.get(&d_adj, 16));
      d_walk = pre_walk;
      d_wbase = pre_wbase;
      if (!d_walk) {
        FMX_HIP(fmx_dev_malloc((void **)&d_walk, (size_t)nwalk * 128));
        if (int rc = keep(idx, d_walk, (uint64_t)nwalk * 128)) return rc;
        FMX_HIP(fmx_dev_malloc((void **)&d_wbase, (size_t)nwsb * 16 * sizeof(uint64_t)));
        if (int rc = keep(idx, d_wbase, (uint64_t)nwsb * 128)) return rc;
      }
      FMX_HIP(hipMemsetAsync(d_cnt + (ncnt - 1), 0, sizeof(uint32_t), 0));
      hipLaunchKernelGGL(kww_counts, dim3(wblocks(nwalk)), dim3(BLK), 0, 0, (const uint8_t *)d_bwt, d_sa, n, level, nwalk, d_cnt);
      size_t tb = 0;
      FMX_HIP(exclusive_sum(nullptr, tb, d_cnt, d_scan, ncnt));
      uint8_t *tmp;
      FMX_HIP(pool.get(&tmp, tb));
      FMX_HIP(exclusive_sum(tmp, tb, d_cnt, d_scan, ncnt));
      // the scan at the start of every counter array (and its end): what each array's own prefix sums start from
      uint64_t start[KWW_COUNTERS + 1], adj[16] = {};
      for (uint32_t k = 0; k <= KWW_COUNTERS; k++)
        FMX_HIP(hipMemcpy(&start[k], d_scan + (size_t)k * nwalk, sizeof(uint64_t), hipMemcpyDeviceToHost));
      const uint64_t edge = ((n - 1) & ((1ull << level) - 1ull)) == 0 ? 1 : 0;   // row 0 (SA = n - 1) is a phase-0 row
      uint64_t smaller = 0;                                          // phase-1 rows with a smaller symbol
      for (uint32_t k = 0; k < KWW_COUNTERS; k++) {
        uint64_t abs0 = 0;
        if (k < 5u) abs0 = k + 1u <= maxc ? idx->h_cs[k + 1u] : 0;   // lf_map2: cs[] folded in (sais.rs:9-32)
        else if (k > 5u) { abs0 = edge + smaller; smaller += start[k + 1] - start[k]; }
        adj[k] = abs0 - start[k];
      }
      FMX_HIP(hipMemcpy(d_adj, adj, sizeof adj, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(kww_bases, dim3((nwsb * 16u + 63u) / 64u), dim3(64), 0, 0, d_scan, d_adj, nwalk, nwsb, shift, d_wbase);
      hipLaunchKernelGGL(kww_records, dim3(wblocks((uint64_t)nwalk * 8u)), dim3(BLK), 0, 0, (const uint8_t *)d_bwt, d_sa, n,
                         level, d_scan, d_adj, d_wbase, nwalk, shift, d_walk);
      FMX_HIP(hipGetLastError());
      FMX_HIP(hipDeviceSynchronize());
      pool.release(tmp); pool.release(d_adj); pool.release(d_scan); pool.release(d_cnt);
      w.walk = d_walk;
      w.wbase = d_wbase;
      w.nwsb = nwsb;
      w.wsb_shift = shift;
      mark("walk records");
    }
  }
  w.doc = nullptr; w.doc_count = 0; w.first_row = 0;
  if (idx->kind == FMX_KIND_MULTI) {
    const uint64_t pieces = hist[0];
    if (pieces >= (1ull << 32)) {
      fmx_set_error(FMX_ERR_UNSUPPORTED, "wide multi-pieces index: 2^32 pieces or more");
      return FMX_ERR_UNSUPPORTED;
    }
    uint64_t *zrows, *zpos;
    unsigned long long *d_got, *d_first;
    uint32_t *d_doc;
    FMX_HIP(pool.get(&zrows, pieces ? pieces : 1));
    FMX_HIP(pool.get(&zpos, pieces ? pieces : 1));
    FMX_HIP(pool.get(&d_got, 1));
    FMX_HIP(pool.get(&d_first, 1));
    FMX_HIP(hipMemset(d_first, 0, 8));
    FMX_HIP(fmx_dev_malloc((void **)&d_doc, (size_t)(pieces ? pieces : 1) * 4));
    if (int rc = keep(idx, d_doc, pieces * 4)) return rc;
    rocprim::counting_iterator<uint64_t> rows(0);
    auto zl = rocprim::make_transform_iterator((const T *)d_bwt, IsZeroSym<T>());
    auto zt = rocprim::make_transform_iterator(d_text, IsZeroSym<T>());
    size_t t1 = 0, t2 = 0;
    FMX_HIP(rocprim::select(nullptr, t1, rows, zl, zrows, d_got, (size_t)n, (hipStream_t)0));
    FMX_HIP(rocprim::select(nullptr, t2, rows, zt, zpos, d_got, (size_t)n, (hipStream_t)0));
    const size_t tb = t1 > t2 ? t1 : t2;
    uint8_t *tmp;
    FMX_HIP(pool.get(&tmp, tb));
    size_t tt = tb;
    FMX_HIP(rocprim::select(tmp, tt, rows, zl, zrows, d_got, (size_t)n, (hipStream_t)0));
    tt = tb;
    FMX_HIP(rocprim::select(tmp, tt, rows, zt, zpos, d_got, (size_t)n, (hipStream_t)0));
    hipLaunchKernelGGL(kwm_doc, dim3(wblocks(pieces)), dim3(BLK), 0, 0, d_sa, zrows, zpos, pieces, n, d_doc, d_first);
    FMX_HIP(hipGetLastError());
    unsigned long long first = 0;
    FMX_HIP(hipMemcpy(&first, d_first, 8, hipMemcpyDeviceToHost));
    w.doc = d_doc;
    w.doc_count = pieces;
    w.first_row = first;
    pool.release(tmp); pool.release(d_first); pool.release(d_got); pool.release(zpos); pool.release(zrows);
    mark("pieces");
  }
  if (idx->flags & FMX_FLAG_KEEP_SA) {
    T *kt;
    FMX_HIP(fmx_dev_malloc((void **)&kt, n * sizeof(T)));
    FMX_HIP(hipMemcpy(kt, d_text, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice));
    if (int rc = keep(idx, kt, n * sizeof(T))) return rc;
    idx->d_text = (uint8_t *)kt;
    pool.disown(d_sa);
    if (int rc = keep(idx, d_sa, n * 8)) return rc;
    idx->d_sa64 = d_sa;
  } else {
    FMX_HIP(hipDeviceSynchronize());
    pool.release(d_sa);
  }
  w.kind = idx->kind;
  w.slen = 0;
  w.lfrun = nullptr;
  if (idx->kind == FMX_KIND_RLFM) {
    if (int rc = build_rlfm_wide<T>(idx, d_bwt, n, L, pool)) return rc;
    idx->is_wide = 1;
    mark("rlfm");
    idx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FMX_OK;
  }
  const uint32_t sb_shift = fmx_wide_n(n) ? FMX_WIDE_SB_SHIFT : FMX_WIDE_SB_SHIFT_TEST, sb_recs = sb_shift - 8u;
  const uint32_t nsb = (uint32_t)(n >> sb_shift) + 1u;
  if (maxc > 7 || sizeof(T) != 1 || idx->kind == FMX_KIND_MULTI) {
    // -- generic wide index: the levels of the multi-ary wavelet matrix (as build_mwm, 64-bit scans) --
    if (int rc = build_wide_levels<T>(idx, d_bwt, n, L, pool)) return rc;
    uint64_t *d_cs, *d_K;
    FMX_HIP(fmx_dev_malloc((void **)&d_cs, ((size_t)maxc + 1) * 8));
    if (int rc = keep(idx, d_cs, ((uint64_t)maxc + 1) * 8)) return rc;
    FMX_HIP(fmx_dev_malloc((void **)&d_K, ((size_t)maxc + 1) * 8));
    if (int rc = keep(idx, d_K, ((uint64_t)maxc + 1) * 8)) return rc;
    FMX_HIP(hipMemcpy(d_cs, idx->h_cs, ((size_t)maxc + 1) * 8, hipMemcpyHostToDevice));
    w.cs = d_cs;
    w.K = d_K;
    if (int rc = fmxw_launch_compute_K(w, d_K)) return rc;
    FMX_HIP(hipDeviceSynchronize());
    idx->is_wide = 1;
    mark("levels");
    idx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FMX_OK;
  }
  // -- records: planes + per-record histograms, 64-bit scan, relative counters + superblock bases --
  const uint32_t nrec = (uint32_t)(n / 256u + 1u);
  uint4 *d_rec = pre_rec;
  uint32_t *d_hist;
  uint64_t *d_scan, *d_base = pre_base;
  if (!d_rec) {
    FMX_HIP(fmx_dev_malloc((void **)&d_rec, (size_t)nrec * 128));
    if (int rc = keep(idx, d_rec, (uint64_t)nrec * 128)) return rc;
    FMX_HIP(fmx_dev_malloc((void **)&d_base, (size_t)nsb * 8 * sizeof(uint64_t)));
    if (int rc = keep(idx, d_base, (uint64_t)nsb * 64)) return rc;
  }
  FMX_HIP(pool.get(&d_hist, (size_t)nrec * 8));
  FMX_HIP(pool.get(&d_scan, (size_t)nrec * 8));
  hipLaunchKernelGGL((k_mwm_pieces<3, T>), dim3(nblocks((uint64_t)nrec * 8)), dim3(BLK), 0, 0, d_bwt, n, 0u, 7u,
                     nrec, d_rec, d_hist);
  {
    size_t tb = 0;
    FMX_HIP(exclusive_sum(nullptr, tb, d_hist, d_scan, (size_t)nrec * 8));
    uint8_t *tmp;
    FMX_HIP(pool.get(&tmp, tb));
    FMX_HIP(exclusive_sum(tmp, tb, d_hist, d_scan, (size_t)nrec * 8));
    pool.release(tmp);
  }
  hipLaunchKernelGGL(kw_counters, dim3(nblocks((uint64_t)nrec * 8)), dim3(BLK), 0, 0, d_scan, nrec, sb_recs, d_rec);
  hipLaunchKernelGGL(kw_bases, dim3((nsb * 8 + 63) / 64), dim3(64), 0, 0, d_scan, nrec, nsb, sb_recs, d_base);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  w.rec = d_rec;
  w.base = d_base;
  w.nsb = nsb;
  w.sb_shift = sb_shift;
  idx->is_wide = 1;
  mark("records");
  idx->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return FMX_OK;
}

int fmx_build_wide(fmx_index *idx, const void *d_text) {
  switch (idx->sym_bytes) {
    case 1: return build_wide_t<uint8_t>(idx, (const uint8_t *)d_text);
    case 2: return build_wide_t<uint16_t>(idx, (const uint16_t *)d_text);
    case 4: return build_wide_t<uint32_t>(idx, (const uint32_t *)d_text);
    default:
      fmx_set_error(FMX_ERR_UNSUPPORTED, "sym_bytes must be 1, 2 or 4 on the device");
      return FMX_ERR_UNSUPPORTED;
  }
}

int fmxw_verify_sa(const fmx_index *idx, uint64_t *violations) {
  const uint64_t n = idx->n;
  *violations = 0;
  uint32_t *mark;
  unsigned long long *bad;
  const size_t words = (size_t)(n / 4 + 1);
  FMX_HIP(fmx_dev_malloc((void **)&mark, words * 4));
  FMX_HIP(fmx_dev_malloc((void **)&bad, 8));
  FMX_HIP(hipMemset(mark, 0, words * 4));
  FMX_HIP(hipMemset(bad, 0, 8));
  if (idx->sym_bytes == 1)
    hipLaunchKernelGGL(kw_verify_sa<uint8_t>, dim3(wblocks(n)), dim3(BLK), 0, 0, (const uint8_t *)idx->d_text, idx->d_sa64, n, mark, bad);
  else if (idx->sym_bytes == 2)
    hipLaunchKernelGGL(kw_verify_sa<uint16_t>, dim3(wblocks(n)), dim3(BLK), 0, 0, (const uint16_t *)idx->d_text, idx->d_sa64, n, mark, bad);
  else
    hipLaunchKernelGGL(kw_verify_sa<uint32_t>, dim3(wblocks(n)), dim3(BLK), 0, 0, (const uint32_t *)idx->d_text, idx->d_sa64, n, mark, bad);
  hipLaunchKernelGGL(kw_count_not_one, dim3(wblocks(n)), dim3(BLK), 0, 0, (const uint8_t *)mark, n, bad);
  unsigned long long hb = 0;
  FMX_HIP(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  (void)hipFree(mark);
  (void)hipFree(bad);
  *violations = hb;
  return FMX_OK;
}

int fmx_build_impl(fmx_index *idx, const void *d_text) {
  if (fmx_wide_build(idx)) return fmx_build_wide(idx, d_text);   // eligibility checked by the caller
  switch (idx->sym_bytes) {
    case 1: return build_impl_t<uint8_t>(idx, (const uint8_t *)d_text);
    case 2: return build_impl_t<uint16_t>(idx, (const uint16_t *)d_text);
    case 4: return build_impl_t<uint32_t>(idx, (const uint32_t *)d_text);
    default:
      fmx_set_error(FMX_ERR_UNSUPPORTED, "sym_bytes must be 1, 2 or 4 on the device");
      return FMX_ERR_UNSUPPORTED;
  }
}
