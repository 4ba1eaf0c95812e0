// fmx_api.hip -- the C ABI declared in include/fmx.h: handle lifetime, error
// reporting, host-pointer marshalling.  All compute is in fmx_build.hip /
// fmx_query.hip; nothing here has a CPU fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "fmx_internal.h"

// ---------------------------------------------------------------------------
// errors (error.rs:3-15: one variant, InvalidText(msg), Display "invalid text: {msg}")
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

const char *fmx_error_message(int code) {
  switch (code) {
    case FMX_OK: return "ok";
    case FMX_ERR_TEXT_START_ZERO:  // sais.rs:129-131
      return "invalid text: the given text must not start with zero character";
    case FMX_ERR_TEXT_END_ZERO:    // sais.rs:135-137
      return "invalid text: the given text must end with exactly one zero character";
    case FMX_ERR_SYMBOL_RANGE: return "symbol exceeds max_character";
    case FMX_ERR_ARG: return "invalid argument";
    case FMX_ERR_UNSUPPORTED: return "unsupported";
    case FMX_ERR_HIP: return "HIP error";
    case FMX_ERR_NO_LOCATE: return "index was built without a sampled suffix array";
    default: return "unknown error";
  }
}
void fmx_set_error(int code, const char *detail) {
  g_last_error = fmx_error_message(code);
  if (detail) {
    g_last_error += ": ";
    g_last_error += detail;
  }
}
int fmx_hip_fail(hipError_t e, const char *what, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s at line %d: %s", what, line, hipGetErrorString(e));
  fmx_set_error(FMX_ERR_HIP, buf);
  return FMX_ERR_HIP;
}
const char *fmx_last_error(void) { return g_last_error.c_str(); }

static int fail(int code, const char *detail = nullptr) {
  fmx_set_error(code, detail);
  return code;
}

// every entry point runs on the index's device and puts the caller's current device back on return
// (a torch program's current device must not change under it)
namespace {
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t set(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev == device) return hipSuccess;
    changed = true;
    return hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (changed && prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace
// ---------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------
static int select_device(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(FMX_ERR_HIP, "no HIP device available (libfmx has no CPU fallback)");
  if (device < 0 || device >= count) return fail(FMX_ERR_ARG, "device ordinal out of range");
  return FMX_OK;
}

void fmx_release_scratch(void) { fmx_release_build_scratch(); }

void fmx_free(fmx_index *idx) {
  if (!idx) return;
  DeviceGuard dg;
  (void)dg.set(idx->device);
  for (int i = 0; i < idx->nalloc; i++) (void)hipFree(idx->d_alloc[i]);
  free(idx->d_alloc);
  if (idx->dev.status) (void)hipFree(idx->dev.status);     // (d_steps lives in the same 16 bytes)
  if (idx->ev0) (void)hipEventDestroy(idx->ev0);
  if (idx->ev1) (void)hipEventDestroy(idx->ev1);
  if (idx->ev_series) {
    for (int i = 0; i < 2 * FMX_SERIES_CAP; i++) (void)hipEventDestroy(idx->ev_series[i]);
    free(idx->ev_series);
  }
  free(idx->h_cs);
  free(idx);
}

// Character = u64 / usize text already in HBM (fmx_build_dev with sym_bytes 8): narrowed to the u32 the engines work on;
// a symbol above max_character (< 2^26) is reported, as the host path does
__global__ __launch_bounds__(256) void fmx_narrow_u64_kernel(const uint64_t *__restrict__ src, uint64_t n, uint64_t max_character,
                                                             uint32_t *__restrict__ dst, uint32_t *__restrict__ bad) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  bool any = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nth) {
    const uint64_t v = src[i];
    any |= v > max_character;
    dst[i] = (uint32_t)v;
  }
  if (any) atomicOr(bad, 1u);
}

// the two device words of a handle -- the sticky status word and the executed-step counter -- in one 16-byte allocation,
// zeroed by one memset; the timing events are created when timing is switched on (fmx_set_timing), not for every index
// (a small build is launch- and round-trip-bound: four runtime calls fewer per index)
static hipError_t alloc_handle_words(fmx_index *idx) {
  uint8_t *p = nullptr;
  hipError_t e = fmx_dev_malloc((void **)&p, 16);
  if (e != hipSuccess) return e;
  idx->dev.status = (uint32_t *)p;
  idx->d_steps = (uint64_t *)(p + 8);
  return hipMemset(p, 0, 16);
}

static int build_common(const void *text, int text_on_device, uint64_t n, uint32_t sym_bytes,
                        uint64_t max_character, uint32_t kind, uint32_t level, uint32_t flags,
                        int device, fmx_index **out) {
  if (!out) return fail(FMX_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (sym_bytes != 1 && sym_bytes != 2 && sym_bytes != 4 && sym_bytes != 8)
    return fail(FMX_ERR_ARG, "sym_bytes must be 1, 2, 4 or 8 (Character = u8/u16/u32/u64)");
  const uint64_t type_max = sym_bytes == 1 ? 0xFFull : (sym_bytes == 2 ? 0xFFFFull : 0xFFFFFFFFull);
  if (max_character == 0 || (sym_bytes != 8 && max_character > type_max))
    return fail(FMX_ERR_ARG, "max_character must be in 1..=MAX of the symbol type");
  // the C array / K table have max_character+1 entries (the reference allocates the same,
  // sais.rs:16): keep them to a sane size
  if (max_character >= (1ull << 26))
    return fail(FMX_ERR_UNSUPPORTED, "max_character >= 2^26 is not supported");
  if (kind != FMX_KIND_FM && kind != FMX_KIND_RLFM && kind != FMX_KIND_MULTI)
    return fail(FMX_ERR_ARG, "unknown kind");
  // n >= 2^32 - 16: the wide engine (64-bit rows) takes every kind (round 4: RLFM and multi-pieces too)
  if ((fmx_wide_n(n) || (flags & FMX_FLAG_FORCE_WIDE)) && n < 2)
    return fail(FMX_ERR_UNSUPPORTED, "FMX_FLAG_FORCE_WIDE needs a text of two symbols or more");
  // the wide engine's record and superblock indices are 32 bits wide: n / 128 + 1 records, (superblock + 1) << 24
  if (n >= (1ull << 38)) return fail(FMX_ERR_UNSUPPORTED, "n >= 2^38 is not supported");
  if (n && !text) return fail(FMX_ERR_ARG, "text is NULL");
  if (int rc = select_device(device)) return rc;
  DeviceGuard dg;
  FMX_HIP(dg.set(device));

  // The two count accelerators (pair index + k-mer start table: the same (s, e) as SearchWrapper::search on every
  // pattern, early-exit pairs included, with fewer probes -- 2.2 x on the 1 GiB DNA text) are part of the DEFAULT index
  // where they pay and the device has room (round 6; FMX_FLAG_AUTO asked for exactly this in rounds 4-5): a DNA-like
  // FM index of 2^24 .. 2^31 symbols with four times the finished index free on the device now.  Parity is what the
  // caller is promised, not the number of probes; a pure space / throughput trade that FMX_FLAG_PLAIN vetoes.
  if (!(flags & FMX_FLAG_PLAIN) && kind == FMX_KIND_FM && sym_bytes == 1 && max_character <= 4 && n >= (1ull << 24) &&
      n < (1ull << 31) && !fmx_wide_n(n) && !(flags & FMX_FLAG_FORCE_WIDE)) {
    const uint64_t samples = level == FMX_NO_LOCATE ? 0 : (level >= 32 ? n * 4 : ((n - 1) >> level) * 4 + n * 2);
    const uint64_t index_est = n / 2 + n + (128ull << 20) + samples;      // records + pair records + table + locate arrays
    size_t free_b = 0, total_b = 0;
    if (fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4 * index_est)
      flags |= FMX_FLAG_PAIR_INDEX | FMX_FLAG_KMER_TABLE;
  }
  // ... and the run-length index over u8 symbols gets the k-mer start table alone (k = 3 at a byte alphabet: one 8-byte
  // lookup instead of the first three steps; at most len / 8 bytes -- 128 MiB of a 3.2 GB index at n = 2^30 -- for
  // 1.22 x the count rate on config 4's 16-symbol patterns).  Same veto, same room rule.
  if (!(flags & FMX_FLAG_PLAIN) && kind == FMX_KIND_RLFM && sym_bytes == 1 && n >= (1ull << 24) && n < (1ull << 31) &&
      !fmx_wide_n(n) && !(flags & FMX_FLAG_FORCE_WIDE)) {
    size_t free_b = 0, total_b = 0;
    if (fmx_dev_mem_info(&free_b, &total_b) == hipSuccess && (uint64_t)free_b >= 4 * (6 * n))
      flags |= FMX_FLAG_KMER_TABLE;
  }

  fmx_index *idx = (fmx_index *)calloc(1, sizeof(fmx_index));
  idx->layout = FMX_LAYOUT;
  idx->device = device;
  idx->n = n;
  idx->sym_bytes = sym_bytes;
  idx->sym_bytes_abi = sym_bytes;
  idx->max_character = max_character;
  idx->kind = kind;
  idx->level_requested = level;
  idx->flags = flags;

  int rc = FMX_OK;
  uint8_t *d_text = nullptr;
  bool own_text = false;
  do {
    hipError_t e;
    if ((e = alloc_handle_words(idx)) != hipSuccess) {
      rc = fmx_hip_fail(e, "handle resources", __LINE__);
      break;
    }

    if (text_on_device && sym_bytes == 8) {
      uint32_t *d_bad = nullptr, h_bad = 0;
      if ((e = fmx_dev_malloc((void **)&d_text, (n ? n : 1) * 4)) != hipSuccess) { rc = fmx_hip_fail(e, "fmx_dev_malloc(text)", __LINE__); break; }
      own_text = true;
      if ((e = fmx_dev_malloc((void **)&d_bad, 4)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMalloc", __LINE__); break; }
      (void)hipMemset(d_bad, 0, 4);
      uint64_t blocks = (n + 255) / 256;
      if (blocks < 1) blocks = 1;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(fmx_narrow_u64_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, (const uint64_t *)text, n, max_character,
                         (uint32_t *)d_text, d_bad);
      e = hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost);
      (void)hipFree(d_bad);
      if (e != hipSuccess) { rc = fmx_hip_fail(e, "narrowing the text", __LINE__); break; }
      if (h_bad) { rc = fail(FMX_ERR_SYMBOL_RANGE, "text symbol exceeds max_character"); break; }
      idx->sym_bytes = 4;
    } else if (text_on_device) {
      d_text = (uint8_t *)text;
    } else if (sym_bytes == 8) {
      // Character = u64/usize: narrow to u32 (every symbol must be <= max_character < 2^26)
      std::string narrow((size_t)(n ? n : 1) * 4, '\0');
      uint32_t *dst = (uint32_t *)&narrow[0];
      const uint64_t *src = (const uint64_t *)text;
      bool bad = false;
      for (uint64_t i = 0; i < n; i++) {
        if (src[i] > max_character) bad = true;
        dst[i] = (uint32_t)src[i];
      }
      if (bad) { rc = fail(FMX_ERR_SYMBOL_RANGE, "text symbol exceeds max_character"); break; }
      idx->sym_bytes = 4;
      if ((e = fmx_dev_malloc((void **)&d_text, (n ? n : 1) * 4)) != hipSuccess) { rc = fmx_hip_fail(e, "fmx_dev_malloc(text)", __LINE__); break; }
      own_text = true;
      if (n && (e = hipMemcpy(d_text, dst, n * 4, hipMemcpyHostToDevice)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy(text)", __LINE__); break; }
    } else {
      if ((e = fmx_dev_malloc((void **)&d_text, (n ? n : 1) * sym_bytes)) != hipSuccess) { rc = fmx_hip_fail(e, "fmx_dev_malloc(text)", __LINE__); break; }
      own_text = true;
      if (n && (e = hipMemcpy(d_text, text, n * sym_bytes, hipMemcpyHostToDevice)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy(text)", __LINE__); break; }
    }
    rc = fmx_build_impl(idx, d_text);
  } while (0);
  if (own_text && d_text) (void)hipFree(d_text);
  if (rc != FMX_OK) {
    fmx_free(idx);
    return rc;
  }
  *out = idx;
  return FMX_OK;
}

int fmx_build(const void *text, uint64_t n, uint32_t sym_bytes, uint64_t max_character,
              uint32_t kind, uint32_t level, uint32_t flags, int device, fmx_index **out) {
  return build_common(text, 0, n, sym_bytes, max_character, kind, level, flags, device, out);
}
int fmx_build_dev(const void *d_text, uint64_t n, uint32_t sym_bytes, uint64_t max_character,
                  uint32_t kind, uint32_t level, uint32_t flags, int device, fmx_index **out) {
  return build_common(d_text, 1, n, sym_bytes, max_character, kind, level, flags, device, out);
}

// ---------------------------------------------------------------------------
// accessors
// ---------------------------------------------------------------------------
uint64_t fmx_len(const fmx_index *idx) { return idx ? idx->n : 0; }
uint64_t fmx_index_bytes(const fmx_index *idx) { return idx ? idx->bytes : 0; }
uint64_t fmx_max_character(const fmx_index *idx) { return idx ? idx->max_character : 0; }
uint32_t fmx_kind(const fmx_index *idx) { return idx ? idx->kind : 0; }
uint32_t fmx_level(const fmx_index *idx) { return idx ? idx->dev.sa_level : FMX_NO_LOCATE; }
int fmx_device(const fmx_index *idx) { return idx ? idx->device : -1; }
uint64_t fmx_num_samples(const fmx_index *idx) { return idx ? idx->nsamples : 0; }
uint64_t fmx_num_runs(const fmx_index *idx) { return idx ? idx->runs : 0; }
uint32_t fmx_sym_bytes(const fmx_index *idx) { return idx ? idx->sym_bytes : 0; }
int fmx_has_pair_index(const fmx_index *idx) { return idx && idx->dev.pair_rec ? 1 : 0; }
int fmx_is_wide(const fmx_index *idx) { return idx && idx->is_wide ? 1 : 0; }
int fmx_text_order(const fmx_index *idx) {
  return idx && (idx->is_wide ? (idx->wide.walk != nullptr || idx->wide.phase != nullptr) : idx->dev.phase != nullptr) ? 1 : 0;
}
int fmx_walk_records(const fmx_index *idx) {
  return idx && (idx->is_wide ? (idx->wide.walk != nullptr || idx->wide.lfrun != nullptr)
                              : (idx->dev.walk != nullptr || idx->dev.lfrun != nullptr)) ? 1 : 0;
}
uint32_t fmx_kmer_k(const fmx_index *idx) { return idx && idx->dev.kmer ? idx->dev.kmer_k : 0; }
double fmx_build_ms(const fmx_index *idx) { return idx ? idx->build_ms : 0.0; }

void fmx_set_timing(fmx_index *idx, int enabled) {
  if (!idx) return;
  idx->ev_valid = 0;
  idx->series_n = 0;
  if (enabled == 1 && !idx->ev0) {                // the event pair of timing == 1: created on first use
    DeviceGuard dg;
    if (dg.set(idx->device) != hipSuccess) return;
    if (hipEventCreate(&idx->ev0) != hipSuccess || hipEventCreate(&idx->ev1) != hipSuccess) {
      if (idx->ev0) (void)hipEventDestroy(idx->ev0);
      idx->ev0 = nullptr;
      idx->ev1 = nullptr;
      enabled = 0;
    }
  }
  if (enabled == 2 && !idx->ev_series) {          // a series of launches: FMX_SERIES_CAP event pairs, created once
    DeviceGuard dg;
    if (dg.set(idx->device) != hipSuccess) return;
    hipEvent_t *ev = (hipEvent_t *)calloc(2 * FMX_SERIES_CAP, sizeof(hipEvent_t));
    bool ok = ev != nullptr;
    int made = 0;
    for (; ok && made < 2 * FMX_SERIES_CAP; made++) ok = hipEventCreate(&ev[made]) == hipSuccess;
    if (!ok) {
      for (int i = 0; ev && i < made - 1; i++) (void)hipEventDestroy(ev[i]);   // (the failing create made nothing)
      free(ev);
      enabled = 0;
    } else {
      idx->ev_series = ev;
    }
  }
  idx->timing = enabled;
}
double fmx_series_kernel_ms(fmx_index *idx) {
  if (!idx || !idx->ev_series || idx->series_n == 0) return -1.0;
  double sum = 0;
  for (int i = 0; i < idx->series_n; i++) {
    float ms = 0;
    if (hipEventSynchronize(idx->ev_series[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, idx->ev_series[2 * i], idx->ev_series[2 * i + 1]) != hipSuccess)
      return -1.0;
    sum += ms;
  }
  const double mean = sum / idx->series_n;
  idx->series_n = 0;
  return mean;
}
double fmx_last_kernel_ms(const fmx_index *idx) {
  if (!idx || !idx->ev_valid) return -1.0;
  float ms = 0;
  if (hipEventSynchronize(idx->ev1) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, idx->ev0, idx->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}
uint64_t fmx_last_steps(const fmx_index *idx) {
  if (!idx || !idx->ev_valid) return 0;
  (void)hipEventSynchronize(idx->ev1);
  uint64_t v = 0;
  if (hipMemcpy(&v, idx->d_steps, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return v;
}

static int status_to_code(uint32_t bits) {
  if (bits & (1u << FMX_ERR_SYMBOL_RANGE)) return FMX_ERR_SYMBOL_RANGE;
  if (bits & (1u << FMX_ERR_ARG)) return FMX_ERR_ARG;
  return bits ? FMX_ERR_ARG : FMX_OK;
}
int fmx_stream_status(const fmx_index *idx) {
  if (!idx) return FMX_ERR_ARG;
  uint32_t bits = 0;
  int prev = -1;
  (void)hipGetDevice(&prev);
  FMX_HIP(hipSetDevice(idx->device));
  struct Back { int d; ~Back() { if (d >= 0) (void)hipSetDevice(d); } } back{prev == idx->device ? -1 : prev};
  FMX_HIP(hipMemcpy(&bits, idx->dev.status, sizeof bits, hipMemcpyDeviceToHost));
  if (bits) FMX_HIP(hipMemset(idx->dev.status, 0, sizeof bits));
  int code = status_to_code(bits);
  if (code) fmx_set_error(code, code == FMX_ERR_SYMBOL_RANGE ? "pattern symbol exceeds max_character" : "row index out of range");
  return code;
}

// ---------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------
#define CHECK_IDX(idx)                                   \
  if (!(idx)) return fail(FMX_ERR_ARG, "index is NULL"); \
  if ((idx)->layout != FMX_LAYOUT)                       \
    return fail(FMX_ERR_ARG, "the index was made by another build of the library (rebuild libfmx*.so together)"); \
  DeviceGuard _dg;                                       \
  FMX_HIP(_dg.set((idx)->device))

int fmx_count_batch_dev(const fmx_index *idx, const void *d_pat, const uint64_t *d_pat_off,
                        uint64_t npat, const uint64_t *d_s0e0, uint64_t *d_out_s,
                        uint64_t *d_out_e, uint64_t *d_out_count, void *stream) {
  CHECK_IDX(idx);
  if (npat && (!d_pat_off)) return fail(FMX_ERR_ARG, "pat_off is NULL");
  return fmx_launch_count(idx, d_pat, d_pat_off, npat, d_s0e0, d_out_s, d_out_e,
                          d_out_count, (hipStream_t)stream);
}
int fmx_offsets_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                    uint64_t *d_out_off, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_offsets(d_s, d_e, npat, d_out_off, (hipStream_t)stream);
}
int fmx_locate_batch_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                         uint64_t npat, const uint64_t *d_out_off, uint64_t total_hits,
                         uint64_t *d_out_pos, void *stream) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  return fmx_launch_locate(idx, d_s, d_e, npat, d_out_off, total_hits, d_out_pos, (hipStream_t)stream);
}
// caller-workspace forms: kernel launches only (no stream-ordered allocation) -> graph-capturable, and
// batches on different streams share nothing but the index
uint64_t fmx_locate_workspace_bytes(const fmx_index *idx, uint64_t total_hits) {
  if (idx && idx->is_wide) return 256;     // a wide locate expands the rows into the position array itself
  if (fmx_locate_is_one_launch(idx)) return 256;   // the one-launch kernel keeps its rows in LDS
  return fmx_locate_rows_bytes(total_hits);
}
uint64_t fmx_offsets_workspace_bytes(uint64_t npat) { return fmx_offsets_tile_bytes(npat); }
int fmx_locate_batch_ws_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e,
                            uint64_t npat, const uint64_t *d_out_off, uint64_t total_hits,
                            uint64_t *d_out_pos, void *d_workspace, uint64_t workspace_bytes, void *stream) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  if (npat == 0 || total_hits == 0) return FMX_OK;
  if (!d_workspace || workspace_bytes < fmx_locate_workspace_bytes(idx, total_hits) || ((uintptr_t)d_workspace & 15u))
    return fail(FMX_ERR_ARG, "workspace is NULL, misaligned or smaller than fmx_locate_workspace_bytes()");
  return fmx_launch_locate(idx, d_s, d_e, npat, d_out_off, total_hits, d_out_pos, (hipStream_t)stream,
                           (uint32_t *)d_workspace);
}
int fmx_offsets_ws_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                       uint64_t *d_out_off, void *d_workspace, uint64_t workspace_bytes, void *stream) {
  CHECK_IDX(idx);
  if (!d_workspace || workspace_bytes < fmx_offsets_tile_bytes(npat) || ((uintptr_t)d_workspace & 7u))
    return fail(FMX_ERR_ARG, "workspace is NULL, misaligned or smaller than fmx_offsets_workspace_bytes()");
  return fmx_launch_offsets(d_s, d_e, npat, d_out_off, (hipStream_t)stream, (uint64_t *)d_workspace);
}
int fmx_get_l_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_scalar(idx, 0, nullptr, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_lf_map_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_scalar(idx, 1, nullptr, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_lf_map2_batch_dev(const fmx_index *idx, const uint64_t *d_c, const uint64_t *d_i, uint64_t k,
                          uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_scalar(idx, 2, d_c, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_get_f_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_scalar(idx, 4, nullptr, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_fl_map_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_scalar(idx, 5, nullptr, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_get_sa_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  return fmx_launch_scalar(idx, 3, nullptr, d_i, k, d_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// host-pointer entry points (copy in, same kernels, copy out, synchronise)
// ---------------------------------------------------------------------------
namespace {
// Host-pointer calls run on per-thread, per-device contexts: two non-blocking streams (concurrent
// callers never share the legacy default stream), a pinned staging buffer + device mirror for
// small calls, and a grow-only device scratch for batches -- no hipMalloc / hipFree / stream
// creation per call.  Everything is released when the owning thread exits.
const size_t kSmallCap = 256u << 10;
const size_t kSmallUse = kSmallCap - 64;   // the last 64 bytes of both staging buffers hold the call's status word
const size_t kRetainCap = 1ull << 30;   // larger batch scratch is allocated and freed per call
const int kPipeEvents = 16;              // chunks of a pipelined host-pointer batch
const unsigned kPipeSearchBlocks = 1792; // grid cap of a chunk's search: 7 x 256 CUs
struct SmallCtx {
  hipStream_t st = nullptr, st2 = nullptr, st3 = nullptr, st4 = nullptr;
  hipEvent_t ev_in[kPipeEvents], ev_k[kPipeEvents], ev_off[kPipeEvents];   // symbols up / kernel done / offsets up, per chunk
  uint8_t *h = nullptr, *d = nullptr;
  uint8_t *hd = nullptr;                   // the device's address of h (page-locked: the GPU can write it), NULL if unmapped
  uint8_t *big = nullptr;
  size_t big_cap = 0;
};
struct SmallCtxSet {  // released when the owning thread exits
  SmallCtx ctx[16];
  ~SmallCtxSet() {
    for (int dvc = 0; dvc < 16; dvc++) {
      SmallCtx &c = ctx[dvc];
      if (!c.st) continue;
      if (hipSetDevice(dvc) != hipSuccess) continue;
      (void)hipStreamDestroy(c.st);
      (void)hipStreamDestroy(c.st2);
      (void)hipStreamDestroy(c.st3);
      (void)hipStreamDestroy(c.st4);
      for (int k = 0; k < kPipeEvents; k++) {
        (void)hipEventDestroy(c.ev_in[k]); (void)hipEventDestroy(c.ev_k[k]); (void)hipEventDestroy(c.ev_off[k]);
      }
      (void)hipHostFree(c.h);
      (void)hipFree(c.d);
      if (c.big) (void)hipFree(c.big);
      c.st = nullptr;
    }
  }
};
SmallCtx *small_ctx(int device) {
  static thread_local SmallCtxSet set;
  if (device < 0 || device >= 16) return nullptr;
  SmallCtx &c = set.ctx[device];
  if (!c.st) {
    if (hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking) != hipSuccess) { c.st = nullptr; return nullptr; }
    if (hipStreamCreateWithFlags(&c.st2, hipStreamNonBlocking) != hipSuccess) {
      (void)hipStreamDestroy(c.st);
      c.st = nullptr;
      return nullptr;
    }
    if (hipStreamCreateWithFlags(&c.st3, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c.st4, hipStreamNonBlocking) != hipSuccess) {
      (void)hipStreamDestroy(c.st);
      (void)hipStreamDestroy(c.st2);
      if (c.st3) (void)hipStreamDestroy(c.st3);
      c.st = nullptr; c.st3 = nullptr;
      return nullptr;
    }
    bool ok = true;
    int made = 0;
    for (; made < kPipeEvents && ok; made++)
      ok = hipEventCreateWithFlags(&c.ev_in[made], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c.ev_k[made], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c.ev_off[made], hipEventDisableTiming) == hipSuccess;
    if (!ok || hipHostMalloc((void **)&c.h, kSmallCap, hipHostMallocMapped) != hipSuccess ||
        fmx_dev_malloc((void **)&c.d, kSmallCap) != hipSuccess) {
      (void)hipStreamDestroy(c.st);
      (void)hipStreamDestroy(c.st2);
      (void)hipStreamDestroy(c.st3);
      (void)hipStreamDestroy(c.st4);
      c.st = nullptr;
      return nullptr;
    }
  }
  if (!c.hd) {
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, c.h, 0) == hipSuccess) c.hd = (uint8_t *)dv; else (void)hipGetLastError();
  }
  return &c;
}
// device scratch of one batch call, carved from the thread's retained buffer
struct HostCall {
  SmallCtx *sx = nullptr;
  uint8_t *base = nullptr;
  size_t off = 0, cap = 0;
  bool temp = false;
  ~HostCall() { if (temp && base) (void)hipFree(base); }
  static size_t pad(size_t b) { return (b + 255) & ~size_t(255); }
  hipError_t open(int device, size_t bytes) {
    sx = small_ctx(device);
    if (!sx) return hipErrorInvalidDevice;
    if (bytes > kRetainCap) {
      temp = true;
      cap = bytes;
      return fmx_dev_malloc((void **)&base, bytes);
    }
    if (bytes > sx->big_cap) {
      if (sx->big) (void)hipFree(sx->big);
      sx->big = nullptr;
      sx->big_cap = 0;
      size_t want = bytes + bytes / 4;
      if (want > kRetainCap) want = kRetainCap;
      hipError_t e = fmx_dev_malloc((void **)&sx->big, want);
      if (e != hipSuccess) return e;
      sx->big_cap = want;
    }
    base = sx->big;
    cap = sx->big_cap;
    return hipSuccess;
  }
  template <typename T> T *take(size_t bytes) {
    T *p = (T *)(base + off);
    off += pad(bytes ? bytes : 8);
    return p;
  }
};
struct Arena {  // bump allocator over the two mirrored buffers
  SmallCtx *c;
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 15) & ~size_t(15); return o; }
  template <typename T> T *host(size_t o) const { return (T *)(c->h + o); }
  template <typename T> T *dev(size_t o) const { return (T *)(c->d + o); }
  template <typename T> T *mapped(size_t o) const { return (T *)(c->hd + o); }   // the host buffer as the GPU sees it
};
struct Scratch {  // device buffers freed on scope exit
  void *p[12];
  int n = 0;
  ~Scratch() { for (int i = 0; i < n; i++) (void)hipFree(p[i]); }
  hipError_t get(void **out, size_t bytes) {
    hipError_t e = fmx_dev_malloc(out, bytes ? bytes : 8);
    if (e == hipSuccess) p[n++] = *out;
    return e;
  }
};
// A host-pointer call owns a status word for its duration (the last 64 bytes of the thread's
// device staging buffer): its kernels report into it instead of the handle's sticky word, so
// threads that query one handle concurrently can never see -- or clear -- each other's error.
thread_local uint32_t *t_call_status = nullptr;
uint32_t *status_dev(SmallCtx *sx) { return (uint32_t *)(sx->d + kSmallUse); }
uint32_t *status_host(SmallCtx *sx) { return (uint32_t *)(sx->h + kSmallUse); }
struct CallStatus {
  explicit CallStatus(SmallCtx *sx) { t_call_status = status_dev(sx); }
  CallStatus(SmallCtx *sx, bool in_host_buffer) {      // the status word of the page-locked staging buffer itself
    t_call_status = in_host_buffer ? (uint32_t *)(sx->hd + kSmallUse) : status_dev(sx);
  }
  ~CallStatus() { t_call_status = nullptr; }
};
int status_result(uint32_t bits) {
  const int code = status_to_code(bits);
  if (code) fmx_set_error(code, code == FMX_ERR_SYMBOL_RANGE ? "pattern symbol exceeds max_character"
                                                              : "argument does not belong to this index");
  return code;
}
// End of a host-pointer call: read the call's status word through the call's own stream and the
// thread's pinned staging word -- never through the legacy default stream, which would
// synchronise with every other blocking stream of the process -- and wait for the stream(s).
int finish_host_call(SmallCtx *sx, hipStream_t other = nullptr) {
  FMX_HIP(hipMemcpyAsync(status_host(sx), status_dev(sx), sizeof(uint32_t), hipMemcpyDeviceToHost, sx->st));
  FMX_HIP(hipStreamSynchronize(sx->st));
  if (other) FMX_HIP(hipStreamSynchronize(other));
  return status_result(*status_host(sx));
}
// Small calls (round 6): the kernels write their results and the call's status word STRAIGHT into the thread's
// page-locked staging buffer (posted writes over the host link: a few dozen bytes) -- one upload, one launch, one
// synchronise, where rounds 1-5 ran a memset, the upload, the launch, two downloads and the synchronise: the fixed
// cost of a one-pattern call is the runtime's per-operation cost, and there are three fewer of them.  (The INPUTS still
// travel by copy: a kernel that read its pattern symbol by symbol over the host link would pay that link's latency in
// every step.)
int finish_mapped_call(SmallCtx *sx) {
  FMX_HIP(hipStreamSynchronize(sx->st));
  return status_result(*status_host(sx));
}
}  // namespace
uint32_t *fmx_call_status(void) { return t_call_status; }
// ---- page-locked caller memory: seen from the device, moved by kernels ----------------------------------
// device address of a host pointer when the GPU can reach it (hipHostMalloc / hipHostRegister memory), else NULL
static void *device_view(const void *p) {
  if (!p) return nullptr;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (at.type != hipMemoryTypeHost) return nullptr;
  void *d = nullptr;                       // the address of THIS pointer (interior pointers included), not of its allocation
  if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}
// ... of a whole array [p, p + bytes): the LAST byte must be page-locked too and map where the first byte's mapping
// says it should (a caller may have registered only part of an array -- hipHostRegister on a sub-range, a pinned
// allocation shorter than the array: the copy kernels and the DMA engine must not run into the unregistered tail,
// which is a fatal GPU memory fault without XNACK).  Called with the index's device current (CHECK_IDX).
static void *device_view(const void *p, size_t bytes) {
  void *d = device_view(p);
  if (!d || bytes <= 1) return d;
  void *dl = device_view((const uint8_t *)p + (bytes - 1));
  return dl == (uint8_t *)d + (bytes - 1) ? d : nullptr;
}
// dst[0, n) = src[0, n); both sides aligned alike modulo 16 (the callers see to that): bytes up to the first
// 16-byte boundary, 16-byte vectors, bytes again
__global__ __launch_bounds__(256) void fmx_copy_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                       size_t n) {
  size_t head = (16u - ((uintptr_t)src & 15u)) & 15u;
  if (head > n) head = n;
  const size_t body = (n - head) >> 4, tail = (n - head) & 15u;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if (tid < head) dst[tid] = src[tid];
  const uint4 *s4 = (const uint4 *)(src + head);
  uint4 *d4 = (uint4 *)(dst + head);
  for (size_t i = tid; i < body; i += nth) d4[i] = s4[i];
  if (tid < tail) dst[head + (body << 4) + tid] = src[head + (body << 4) + tid];
}
// 8-byte words whose two sides differ by 8 modulo 16 (u64 arrays at an odd word of the caller's buffer)
__global__ __launch_bounds__(256) void fmx_copy8_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src,
                                                        size_t nwords) {
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += nth) dst[i] = src[i];
}
// off[j] = first + j * stride, j = 0..count (a chunk of equally long patterns: the offsets are made here, not uploaded)
__global__ __launch_bounds__(256) void fmx_fill_offsets_kernel(uint64_t *__restrict__ off, uint64_t first, uint64_t stride,
                                                               uint64_t count) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= count; j += nth) off[j] = first + j * stride;
}
// are the offsets off[0..count] an arithmetic progression (equally long patterns)?  The common shape of a batch --
// and then the 8 bytes per pattern need not cross the host link at all.  Runs on the calling thread while the DMA
// engine uploads the chunk's symbols; the chunk's slice (2 MB of a 2^20-pattern batch's four) is read once.
static bool uniform_offsets(const uint64_t *off, uint64_t count, uint64_t *stride) {
  if (count == 0) { *stride = 0; return true; }
  const uint64_t m = off[1] - off[0], first = off[0];
  if (off[1] < off[0]) return false;
  uint64_t bad = 0;
  for (uint64_t j = 2; j <= count; j++) bad |= off[j] ^ (first + j * m);     // branch-free: vectorises
  *stride = m;
  return bad == 0;
}
// 256 blocks move 56 GB/s over the host link and leave the CUs to the search kernels (pcie_probe: 64 blocks the
// same rate alone, 1024 slower)
static unsigned g_copy_blocks = 32;   // 32 x 256 lanes x 16 B in flight carry 56 GB/s over the host link
static void launch_copy(void *dst, const void *src, size_t bytes, hipStream_t S) {
  if (!bytes) return;
  size_t blocks = (bytes / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > g_copy_blocks) blocks = g_copy_blocks;
  if ((((uintptr_t)dst ^ (uintptr_t)src) & 15u) == 0)
    hipLaunchKernelGGL(fmx_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, S, (uint8_t *)dst, (const uint8_t *)src, bytes);
  else   // only the 8-byte arrays can get here: symbols are copied with matching alignment
    hipLaunchKernelGGL(fmx_copy8_kernel, dim3((unsigned)blocks), dim3(256), 0, S, (uint64_t *)dst, (const uint64_t *)src,
                       bytes / 8);
}


int fmx_count_batch(const fmx_index *idx, const void *pat, const uint64_t *pat_off, uint64_t npat,
                    const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e, uint64_t *out_count) {
  CHECK_IDX(idx);
  if (npat == 0) return FMX_OK;
  if (!pat_off) return fail(FMX_ERR_ARG, "pat_off is NULL");
  const uint64_t total = pat_off[npat];
  // The symbols this call reads are pat[first .. total): a caller that hands over a SLICE of a larger batch (entries
  // a .. b of its offsets with the batch's pattern buffer -- fmx_count_batch_multi's shards) moves and stages that
  // span only.  The kernels keep the caller's absolute offsets: they get the address the device copy WOULD have if it
  // began at symbol 0 (never dereferenced below symbol `first`: every pattern's offsets are checked against the span).
  const uint64_t first = pat_off[0];
  if (first > total) return fail(FMX_ERR_ARG, "pat_off is not non-decreasing");
  const uint64_t span = total - first;
  if (span && !pat) return fail(FMX_ERR_ARG, "pat is NULL");
  const uint32_t sb = idx->sym_bytes;  // device symbol width
  if (span * sb + npat * 48 + 128 <= kSmallUse && idx->sym_bytes_abi != 8) {
    if (SmallCtx *sx = small_ctx(idx->device)) {
      Arena a{sx};
      const bool mapped = sx->hd != nullptr;
      CallStatus cs(sx, mapped);
      if (mapped) *status_host(sx) = 0; else FMX_HIP(hipMemsetAsync(status_dev(sx), 0, 4, sx->st));
      const size_t op = a.take(span * sb ? span * sb : 1), oo = a.take((npat + 1) * 8);
      const size_t ose = a.take(s0e0 ? npat * 16 : 0), in_end = a.off;
      const size_t os = a.take(npat * 8), oe = a.take(npat * 8), oc = a.take(npat * 8), ost = a.off;
      if (span) memcpy(a.host<uint8_t>(op), (const uint8_t *)pat + first * sb, span * sb);
      if (first == 0) {
        memcpy(a.host<uint8_t>(oo), pat_off, (npat + 1) * 8);
      } else {                                  // offsets relative to the staged span (an entry below `first` wraps to
        uint64_t *ho = a.host<uint64_t>(oo);    // a huge value: refused by the kernel like any offset that goes backwards)
        for (uint64_t k = 0; k <= npat; k++) ho[k] = pat_off[k] - first;
      }
      if (s0e0) memcpy(a.host<uint8_t>(ose), s0e0, npat * 16);
      FMX_HIP(hipMemcpyAsync(sx->d, sx->h, in_end, hipMemcpyHostToDevice, sx->st));
      int rc;
      if (mapped) {
        if ((rc = fmx_launch_count(idx, a.dev<uint8_t>(op), a.dev<uint64_t>(oo), npat,
                                   s0e0 ? a.dev<uint64_t>(ose) : nullptr, a.mapped<uint64_t>(os),
                                   a.mapped<uint64_t>(oe), a.mapped<uint64_t>(oc), sx->st)) != FMX_OK) {
          (void)hipStreamSynchronize(sx->st);
          return rc;
        }
        rc = finish_mapped_call(sx);
      } else {
        if ((rc = fmx_launch_count(idx, a.dev<uint8_t>(op), a.dev<uint64_t>(oo), npat,
                                   s0e0 ? a.dev<uint64_t>(ose) : nullptr, a.dev<uint64_t>(os),
                                   a.dev<uint64_t>(oe), a.dev<uint64_t>(oc), sx->st)) != FMX_OK)
          return rc;
        FMX_HIP(hipMemcpyAsync(a.host<uint8_t>(os), a.dev<uint8_t>(os), (size_t)(ost - os), hipMemcpyDeviceToHost, sx->st));
        rc = finish_host_call(sx);
      }
      if (out_s) memcpy(out_s, a.host<uint8_t>(os), npat * 8);
      if (out_e) memcpy(out_e, a.host<uint8_t>(oe), npat * 8);
      if (out_count) memcpy(out_count, a.host<uint8_t>(oc), npat * 8);
      return rc;
    }
  }
  // batches
  std::vector<uint32_t> narrow;
  const uint8_t *src = (const uint8_t *)pat;
  if (idx->sym_bytes_abi == 8 && span) {  // u64 patterns: narrow, saturating so out-of-range stays out of range
    narrow.resize((size_t)span);
    const uint64_t *p64 = (const uint64_t *)pat + first;
    for (uint64_t i = 0; i < span; i++) narrow[i] = p64[i] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)p64[i];
    src = (const uint8_t *)narrow.data() - first * sb;     // src + first * sb = the first narrowed symbol
  }
  // Page-locked caller arrays (hipHostMalloc / hipHostRegister / torch pin_memory) can be DMA'd in place and are
  // visible to the GPU: the batch goes through a chunk pipeline over three role streams (below) -- DMA upload,
  // search, download by copy kernels -- that keeps both directions of the host link and the search busy at once.
  // Built from hipMemcpyAsync alone it was slower than no pipeline at all: every hand-over between a DMA copy and
  // a kernel of the same stream costs tens of microseconds on this runtime (4.0 ms per 2^20 x 32 call with eight
  // chunks, 2.0 ms with two; benchmarks/gpu/pcie_probe.hip, hostpipe_sweep.sh).
  const size_t b_pat = (size_t)span * sb, b_out = (size_t)npat * 8;
  // (every array checked over its whole length, with the index's device current; one that is only partly
  // page-locked sends the call down the pageable path)
  const bool all_pinned = idx->sym_bytes_abi != 8 && !idx->timing && npat >= (1u << 16) &&
                          device_view((const uint8_t *)pat + first * sb, b_pat) && device_view(pat_off, (size_t)(npat + 1) * 8) &&
                          (!s0e0 || device_view(s0e0, 2 * b_out)) && (!out_s || device_view(out_s, b_out)) &&
                          (!out_e || device_view(out_e, b_out)) && (!out_count || device_view(out_count, b_out));
  const uint64_t max_ch = 8;
  const size_t b_off = (size_t)(npat + 1 + max_ch) * 8;      // every chunk gets its own slice of the offsets
  HostCall hc;
  FMX_HIP(hc.open(idx->device, HostCall::pad(b_pat + 32) + HostCall::pad(b_off) +
                                   (s0e0 ? HostCall::pad(2 * b_out) : 0) + 3 * HostCall::pad(b_out)));
  uint8_t *d_pat = hc.take<uint8_t>(b_pat + 32);
  uint64_t *d_off = hc.take<uint64_t>(b_off);
  uint64_t *d_se = s0e0 ? hc.take<uint64_t>(2 * b_out) : nullptr;
  uint64_t *d_s = hc.take<uint64_t>(b_out), *d_e = hc.take<uint64_t>(b_out), *d_c = hc.take<uint64_t>(b_out);
  SmallCtx *sx = hc.sx;
  CallStatus cs(sx);
  if (all_pinned) {
    // three ROLE streams linked by events: upload of chunk k+1, search of chunk k and download of chunk k-1
    // run at the same time, each stage at its full rate.  (One stream per CHUNK was measured first: the
    // chunks then march in step -- three uploads share the link, then three searches share the CUs -- and
    // nothing overlaps: 1.75 ms, the sum of the stages.  benchmarks/gpu/hostpipe_trace.sh)
    hipStream_t s_k = sx->st, s_in = sx->st2, s_out = sx->st3, s_off = sx->st4;
    auto drain = [&]() {
      (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(s_off); (void)hipStreamSynchronize(s_k);
      (void)hipStreamSynchronize(s_out);
    };
    // four chunks: the upload is the slowest stage (40 MB at 56 GB/s against 0.65 ms of search and 24 MB out for
    // 2^20 x 32) and every DMA copy costs ~20 us on top of its bytes.  Measured per call (2^20 x 32,
    // benchmarks/gpu/hostpipe_sweep.sh, profiles/r03/hostpipe_*.txt): 4 equal chunks 1.36-1.39 ms, 8 chunks
    // 1.34-1.37 ms but with 2-3 ms outliers on some boxes; all offsets in one copy ahead of the chunks, or chunks
    // that shrink towards the end, 1.46-1.6 ms; the offsets on a copy stream of their own 1.8-2.0 ms (two DMA
    // queues get in each other's way)
    uint64_t nch = 4;
    unsigned search_blocks = kPipeSearchBlocks;
    bool h2d_dma = true;
    bool off_stream = false;   // offsets on a copy stream of their own: measured slower (see above)
    bool synth_off = true;     // equally long patterns: offsets made on the device instead of uploaded
    int direct_out = 2;        // the search writes its results straight into the caller's page-locked arrays: 0 never
                               // (copy kernels per chunk), 1 the last chunk only, 2 every chunk
#ifdef FMX_TUNE_HOSTPIPE   // benchmarks/gpu/hostpipe_sweep.sh only: never defined for the shipped library
    if (const char *v = getenv("FMX_PIPE_CHUNKS")) { const uint64_t u = (uint64_t)atoi(v); if (u >= 1 && u <= max_ch) nch = u; }
    if (const char *v = getenv("FMX_PIPE_BLOCKS")) search_blocks = (unsigned)atoi(v);
    if (const char *v = getenv("FMX_PIPE_COPY_BLOCKS")) g_copy_blocks = (unsigned)atoi(v);
    if (const char *v = getenv("FMX_PIPE_H2D")) h2d_dma = atoi(v) != 0;
    if (const char *v = getenv("FMX_PIPE_OFF_STREAM")) off_stream = atoi(v) != 0;
    if (const char *v = getenv("FMX_PIPE_SYNTH_OFF")) synth_off = atoi(v) != 0;
    if (const char *v = getenv("FMX_PIPE_DIRECT_OUT")) direct_out = atoi(v);
#endif
    // a HIP call that fails in the middle of the pipeline must not leave earlier chunks in flight: their copy
    // kernels would keep writing the caller's arrays, and the thread's retained scratch -- which the next call on
    // this thread overwrites -- would still be read by their searches
#define FMX_HIP_DRAIN(x)                                             \
    do {                                                             \
      hipError_t _e = (x);                                           \
      if (_e != hipSuccess) { drain(); return fmx_hip_fail(_e, #x, __LINE__); } \
    } while (0)
    auto cut = [&](uint64_t k) { return npat * k / nch; };
    FMX_HIP_DRAIN(hipMemsetAsync(status_dev(sx), 0, 4, s_k));   // ahead of every search in the search stream
    // the device copy of the symbols keeps the caller's alignment modulo 16, so that both sides of every
    // chunk's copy are aligned alike whatever pa is
    uint8_t *d_pat_al = d_pat + ((uintptr_t)(src + first * sb) & 15u) - first * sb;   // (virtual: symbol 0 of the caller's buffer)
    const uint8_t *v_pat = (const uint8_t *)device_view(src + first * sb) - first * sb;
    const uint64_t *v_off = (const uint64_t *)device_view(pat_off);
    const uint64_t *v_se = s0e0 ? (const uint64_t *)device_view(s0e0) : nullptr;
    uint64_t *v_os = out_s ? (uint64_t *)device_view(out_s) : nullptr;
    uint64_t *v_oe = out_e ? (uint64_t *)device_view(out_e) : nullptr;
    uint64_t *v_oc = out_count ? (uint64_t *)device_view(out_count) : nullptr;
    for (uint64_t k = 0; k < nch; k++) {
      const uint64_t a = cut(k), b = cut(k + 1);
      if (b == a) continue;
      const uint64_t pa = pat_off[a], pb = pat_off[b];
      if (pb < pa || pb > total || pa < first) {
        drain();
        return fail(FMX_ERR_ARG, "pat_off is not non-decreasing");
      }
      // upload by DMA: the stream holds nothing but copies, so none of them waits for a kernel (a DMA copy that
      // depends on a kernel is handed over by the host: tens of microseconds each); the search's wait for the
      // copy is a barrier on the GPU side.  Upload by copy kernels was measured too: host reads in flight slow
      // the concurrent search five-fold (benchmarks/gpu/hostpipe_trace.sh)
      uint64_t *off_k = d_off + a + k;               // entries a..b of the caller's offsets, this chunk's own copy
      if (h2d_dma) {
        // the symbols first: while the DMA engine moves them, this thread looks at the chunk's offsets.  Equally long
        // patterns (the usual batch) -> the offsets are generated on the device by a tiny kernel ahead of the search
        // and never cross the link: one DMA copy per chunk instead of two, and 8 bytes per pattern less to upload
        // (2^20 x 32: 32 MB instead of 40, 1.36 -> ~1.0 ms per call)
        if (pb > pa)
          FMX_HIP_DRAIN(hipMemcpyAsync(d_pat_al + pa * sb, src + pa * sb, (size_t)(pb - pa) * sb, hipMemcpyHostToDevice, s_in));
        uint64_t stride = 0;
        if (synth_off && uniform_offsets(pat_off + a, b - a, &stride)) {
          // on the fourth stream, long before the search needs them: no kernel of its own between two searches
          hipLaunchKernelGGL(fmx_fill_offsets_kernel, dim3(64), dim3(256), 0, s_off, off_k, pa, stride, b - a);
          FMX_HIP_DRAIN(hipEventRecord(sx->ev_off[k], s_off));
          FMX_HIP_DRAIN(hipStreamWaitEvent(s_k, sx->ev_off[k], 0));
        } else {
          FMX_HIP_DRAIN(hipMemcpyAsync(off_k, pat_off + a, (size_t)(b - a + 1) * 8, hipMemcpyHostToDevice,
                                 off_stream ? s_off : s_in));
          if (off_stream) {
            FMX_HIP_DRAIN(hipEventRecord(sx->ev_off[k], s_off));
            FMX_HIP_DRAIN(hipStreamWaitEvent(s_k, sx->ev_off[k], 0));
          }
        }
        if (s0e0) FMX_HIP_DRAIN(hipMemcpyAsync(d_se + 2 * a, s0e0 + 2 * a, (size_t)(b - a) * 16, hipMemcpyHostToDevice, s_in));
      } else {
        launch_copy(off_k, v_off + a, (size_t)(b - a + 1) * 8, s_in);
        if (pb > pa) launch_copy(d_pat_al + pa * sb, v_pat + pa * sb, (size_t)(pb - pa) * sb, s_in);
        if (s0e0) launch_copy(d_se + 2 * a, v_se + 2 * a, (size_t)(b - a) * 16, s_in);
      }
      FMX_HIP_DRAIN(hipEventRecord(sx->ev_in[k], s_in));
      FMX_HIP_DRAIN(hipStreamWaitEvent(s_k, sx->ev_in[k], 0));
      // the kernel bounds every pattern's offsets by the last entry it is given: pb <= total.  Fewer blocks than
      // the CUs have slots for: the download kernels of the chunk before need somewhere to run WHILE this
      // search runs (a persistent 2048-block grid holds every slot until its last pattern)
      // results: written by the search itself into the caller's page-locked arrays (posted writes over the link,
      // 24 bytes per pattern: far below what it carries) -- no download stage, no copy kernels sharing the CUs with
      // the searches, the whole grid for every search (round 3 downloaded every chunk with three copy kernels and
      // kept an eighth of the workgroup slots free for them: the last chunk's copies, ~135 us, ended every call)
      const bool direct = direct_out == 2 || (direct_out == 1 && b == npat);
      if (int rc = direct ? fmx_launch_count(idx, d_pat_al, off_k, b - a, s0e0 ? d_se + 2 * a : nullptr,
                                             v_os ? v_os + a : nullptr, v_oe ? v_oe + a : nullptr,
                                             v_oc ? v_oc + a : nullptr, s_k, direct_out == 2 ? 0u : search_blocks)
                          : fmx_launch_count(idx, d_pat_al, off_k, b - a, s0e0 ? d_se + 2 * a : nullptr, d_s + a, d_e + a,
                                             d_c + a, s_k, search_blocks)) {
        drain();
        return rc;
      }
      if (!direct) {
        FMX_HIP_DRAIN(hipEventRecord(sx->ev_k[k], s_k));
        FMX_HIP_DRAIN(hipStreamWaitEvent(s_out, sx->ev_k[k], 0));
        // download by copy kernels: kernel after kernel, no hand-over to a DMA engine
        if (v_os) launch_copy(v_os + a, d_s + a, (size_t)(b - a) * 8, s_out);
        if (v_oe) launch_copy(v_oe + a, d_e + a, (size_t)(b - a) * 8, s_out);
        if (v_oc) launch_copy(v_oc + a, d_c + a, (size_t)(b - a) * 8, s_out);
      }
    }
    FMX_HIP_DRAIN(hipGetLastError());
    FMX_HIP_DRAIN(hipStreamSynchronize(s_in));
    FMX_HIP_DRAIN(hipStreamSynchronize(s_off));
    FMX_HIP_DRAIN(hipStreamSynchronize(s_out));            // s_k == sx->st is waited for below
#undef FMX_HIP_DRAIN
    return finish_host_call(sx);
  }
  // pageable arrays: the runtime's copies (pin, copy, unpin), each chunk chained in one of two streams.  Two
  // halves: every such copy has a fixed cost of 50-80 us, so more, smaller chunks lose (2^20 x 32: 1 chunk
  // 1.92 ms, 2: 1.42, 4: 1.52, 8: 2.76, 16: 3.13)
  hipStream_t st[2] = {sx->st, sx->st2};
  FMX_HIP(hipMemsetAsync(status_dev(sx), 0, 4, st[0]));   // ordered before every chunk by the wait below
  uint64_t nch = (npat >= (1u << 17) && !idx->timing) ? 2 : 1;
  // all offsets first (every chunk's kernel reads its own slice plus one entry)
  FMX_HIP(hipMemcpyAsync(d_off, pat_off, (size_t)(npat + 1) * 8, hipMemcpyHostToDevice, st[0]));
  FMX_HIP(hipStreamSynchronize(st[0]));
  auto download = [&](uint64_t k) -> hipError_t {
    const uint64_t a = npat * k / nch, b = npat * (k + 1) / nch;
    hipStream_t S = st[k & 1];
    hipError_t e = hipSuccess;
    if (out_s && e == hipSuccess) e = hipMemcpyAsync(out_s + a, d_s + a, (b - a) * 8, hipMemcpyDeviceToHost, S);
    if (out_e && e == hipSuccess) e = hipMemcpyAsync(out_e + a, d_e + a, (b - a) * 8, hipMemcpyDeviceToHost, S);
    if (out_count && e == hipSuccess) e = hipMemcpyAsync(out_count + a, d_c + a, (b - a) * 8, hipMemcpyDeviceToHost, S);
    return e;
  };
  for (uint64_t k = 0; k < nch; k++) {
    const uint64_t a = npat * k / nch, b = npat * (k + 1) / nch;
    const uint64_t pa = pat_off[a], pb = pat_off[b];
    if (pb < pa || pb > total || pa < first) {
      (void)hipStreamSynchronize(st[0]);
      (void)hipStreamSynchronize(st[1]);
      return fail(FMX_ERR_ARG, "pat_off is not non-decreasing");
    }
    hipStream_t S = st[k & 1];
    uint8_t *const d_pat0 = d_pat - first * sb;          // (virtual: symbol 0 of the caller's buffer)
    if (pb > pa)
      FMX_HIP(hipMemcpyAsync(d_pat0 + pa * sb, src + pa * sb, (size_t)(pb - pa) * sb, hipMemcpyHostToDevice, S));
    if (s0e0)
      FMX_HIP(hipMemcpyAsync(d_se + 2 * a, s0e0 + 2 * a, (size_t)(b - a) * 16, hipMemcpyHostToDevice, S));
    if (int rc = fmx_launch_count(idx, d_pat0, d_off + a, b - a, s0e0 ? d_se + 2 * a : nullptr, d_s + a,
                                  d_e + a, d_c + a, S)) {
      (void)hipStreamSynchronize(st[0]);
      (void)hipStreamSynchronize(st[1]);
      return rc;
    }
    if (k) FMX_HIP(download(k - 1));
  }
  FMX_HIP(download(nch - 1));
  FMX_HIP(hipStreamSynchronize(st[1]));              // st[0] == sx->st is waited for below
  return finish_host_call(sx);
}

// off[k] -= base, k = 0 .. count (a slice of a larger batch's offsets -> offsets of its own)
__global__ __launch_bounds__(256) void fmx_rebase_offsets_kernel(uint64_t *__restrict__ off, uint64_t base, uint64_t count) {
  const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= count; j += nth) off[j] -= base;
}
// `slice`: out_off[0 .. npat] are entries a .. b of a larger batch's offsets -- the slice's hits are
// out_pos[out_off[0] .. out_off[npat]) and nothing in front of them is a gap (fmx_locate_batch_multi's shards)
static int locate_host(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                       const uint64_t *out_off, uint64_t *out_pos, bool slice) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  if (npat == 0) return FMX_OK;
  if (!s || !e || !out_off) return fail(FMX_ERR_ARG, "NULL argument");
  const uint64_t base = slice ? out_off[0] : 0;
  if (out_off[npat] < base) return fail(FMX_ERR_ARG, "out_off is not non-decreasing");
  uint64_t total = out_off[npat] - base;
  if (total == 0) return FMX_OK;
  if (!out_pos) return fail(FMX_ERR_ARG, "out_pos is NULL");
  HostCall hc;
  const size_t b_in = (size_t)npat * 8, b_pos = (size_t)total * 8;
  const size_t b_rows = (size_t)fmx_locate_workspace_bytes(idx, total);
  FMX_HIP(hc.open(idx->device, 2 * HostCall::pad(b_in) + HostCall::pad(b_in + 8) + HostCall::pad(b_pos) +
                                   HostCall::pad(b_rows)));
  uint64_t *d_s = hc.take<uint64_t>(b_in), *d_e = hc.take<uint64_t>(b_in);
  uint64_t *d_off = hc.take<uint64_t>(b_in + 8), *d_pos = hc.take<uint64_t>(b_pos);
  uint32_t *d_rows = hc.take<uint32_t>(b_rows);   // the walk's scratch comes from the thread's retained buffer too
  hipStream_t S = hc.sx->st;
  CallStatus cs(hc.sx);
  FMX_HIP(hipMemsetAsync(status_dev(hc.sx), 0, 4, S));
  FMX_HIP(hipMemcpyAsync(d_s, s, b_in, hipMemcpyHostToDevice, S));
  FMX_HIP(hipMemcpyAsync(d_e, e, b_in, hipMemcpyHostToDevice, S));
  FMX_HIP(hipMemcpyAsync(d_off, out_off, b_in + 8, hipMemcpyHostToDevice, S));
  if (base) {
    uint64_t blocks = (npat + 256) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(fmx_rebase_offsets_kernel, dim3((unsigned)blocks), dim3(256), 0, S, d_off, base, npat);
  }
  if (int rc = fmx_launch_locate(idx, d_s, d_e, npat, d_off, total, d_pos, S, d_rows)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  FMX_HIP(hipMemcpyAsync(out_pos + base, d_pos, b_pos, hipMemcpyDeviceToHost, S));
  return finish_host_call(hc.sx);
}
int fmx_locate_batch(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                     const uint64_t *out_off, uint64_t *out_pos) {
  return locate_host(idx, s, e, npat, out_off, out_pos, false);
}
int fmx_locate_batch_slice(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                           const uint64_t *out_off, uint64_t *out_pos) {
  return locate_host(idx, s, e, npat, out_off, out_pos, true);
}

// Patterns already RESIDENT on the index's device, results into HOST arrays (SURVEY section 8d's protocol: uploads
// excluded, "include D2H of results on the GPU side").  Page-locked result arrays are written by the search itself
// (posted writes over the host link, 24 bytes per pattern); pageable ones go through device scratch and the runtime's
// copies.  Synchronous, like every host-pointer call.
int fmx_count_resident_slice(const fmx_index *idx, const void *d_pat, const uint64_t *d_pat_off, uint64_t npat,
                             const uint64_t *d_s0e0, uint64_t *out_s, uint64_t *out_e, uint64_t *out_count) {
  CHECK_IDX(idx);
  if (npat == 0) return FMX_OK;
  if (!d_pat_off) return fail(FMX_ERR_ARG, "pat_off is NULL");
  const size_t b_out = (size_t)npat * 8;
  uint64_t *v_s = out_s ? (uint64_t *)device_view(out_s, b_out) : nullptr;
  uint64_t *v_e = out_e ? (uint64_t *)device_view(out_e, b_out) : nullptr;
  uint64_t *v_c = out_count ? (uint64_t *)device_view(out_count, b_out) : nullptr;
  const bool direct = (!out_s || v_s) && (!out_e || v_e) && (!out_count || v_c);
  HostCall hc;
  FMX_HIP(hc.open(idx->device, direct ? 256 : 3 * HostCall::pad(b_out)));
  SmallCtx *sx = hc.sx;
  hipStream_t S = sx->st;
  CallStatus cs(sx);
  FMX_HIP(hipMemsetAsync(status_dev(sx), 0, 4, S));
  if (direct) {
    if (int rc = fmx_launch_count(idx, d_pat, d_pat_off, npat, d_s0e0, v_s, v_e, v_c, S)) {
      (void)hipStreamSynchronize(S);
      return rc;
    }
    return finish_host_call(sx);
  }
  uint64_t *d_s = hc.take<uint64_t>(b_out), *d_e = hc.take<uint64_t>(b_out), *d_c = hc.take<uint64_t>(b_out);
  if (int rc = fmx_launch_count(idx, d_pat, d_pat_off, npat, d_s0e0, out_s ? d_s : nullptr, out_e ? d_e : nullptr,
                                out_count ? d_c : nullptr, S)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  if (out_s) FMX_HIP(hipMemcpyAsync(out_s, d_s, b_out, hipMemcpyDeviceToHost, S));
  if (out_e) FMX_HIP(hipMemcpyAsync(out_e, d_e, b_out, hipMemcpyDeviceToHost, S));
  if (out_count) FMX_HIP(hipMemcpyAsync(out_count, d_c, b_out, hipMemcpyDeviceToHost, S));
  return finish_host_call(sx);
}
void fmx_set_error_text(const char *text) { g_last_error = text ? text : ""; }

static int scalar_host(const fmx_index *idx, int op, const uint64_t *c, const uint64_t *i,
                       uint64_t k, uint64_t *out) {
  CHECK_IDX(idx);
  if (op == 3 && idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  if (k == 0) return FMX_OK;
  if (!i || !out) return fail(FMX_ERR_ARG, "NULL argument");
  if (k * 24 + 64 <= kSmallUse) {
    if (SmallCtx *sx = small_ctx(idx->device)) {
      Arena a{sx};
      const bool mapped = sx->hd != nullptr;
      CallStatus cs(sx, mapped);
      if (mapped) *status_host(sx) = 0; else FMX_HIP(hipMemsetAsync(status_dev(sx), 0, 4, sx->st));
      const size_t oi = a.take(k * 8), oc = a.take(c ? k * 8 : 0), in_end = a.off;
      const size_t oo = a.take(k * 8);
      memcpy(a.host<uint8_t>(oi), i, k * 8);
      if (c) memcpy(a.host<uint8_t>(oc), c, k * 8);
      FMX_HIP(hipMemcpyAsync(sx->d, sx->h, in_end, hipMemcpyHostToDevice, sx->st));
      int rc = fmx_launch_scalar(idx, op, c ? a.dev<uint64_t>(oc) : nullptr, a.dev<uint64_t>(oi), k,
                                 mapped ? a.mapped<uint64_t>(oo) : a.dev<uint64_t>(oo), sx->st);
      if (rc != FMX_OK) {
        (void)hipStreamSynchronize(sx->st);
        return rc;
      }
      if (mapped) {
        rc = finish_mapped_call(sx);
      } else {
        FMX_HIP(hipMemcpyAsync(a.host<uint8_t>(oo), a.dev<uint8_t>(oo), k * 8, hipMemcpyDeviceToHost, sx->st));
        rc = finish_host_call(sx);
      }
      memcpy(out, a.host<uint8_t>(oo), k * 8);
      return rc;
    }
  }
  HostCall hc;
  const size_t bk = (size_t)k * 8;
  FMX_HIP(hc.open(idx->device, 3 * HostCall::pad(bk)));
  uint64_t *d_i = hc.take<uint64_t>(bk), *d_o = hc.take<uint64_t>(bk), *d_c = c ? hc.take<uint64_t>(bk) : nullptr;
  hipStream_t S = hc.sx->st;
  CallStatus cs(hc.sx);
  FMX_HIP(hipMemsetAsync(status_dev(hc.sx), 0, 4, S));
  FMX_HIP(hipMemcpyAsync(d_i, i, bk, hipMemcpyHostToDevice, S));
  if (c) FMX_HIP(hipMemcpyAsync(d_c, c, bk, hipMemcpyHostToDevice, S));
  if (int rc = fmx_launch_scalar(idx, op, d_c, d_i, k, d_o, S)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  FMX_HIP(hipMemcpyAsync(out, d_o, bk, hipMemcpyDeviceToHost, S));
  return finish_host_call(hc.sx);
}
int fmx_get_l_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) { return scalar_host(idx, 0, nullptr, i, k, out); }
int fmx_lf_map_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) { return scalar_host(idx, 1, nullptr, i, k, out); }
int fmx_lf_map2_batch(const fmx_index *idx, const uint64_t *c, const uint64_t *i, uint64_t k, uint64_t *out) {
  if (!c) return fail(FMX_ERR_ARG, "c is NULL");
  return scalar_host(idx, 2, c, i, k, out);
}
int fmx_get_sa_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) { return scalar_host(idx, 3, nullptr, i, k, out); }
int fmx_get_f_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) { return scalar_host(idx, 4, nullptr, i, k, out); }
int fmx_fl_map_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) { return scalar_host(idx, 5, nullptr, i, k, out); }

// Match::iter_chars_backward / iter_chars_forward for many rows (wrapper.rs:142-183)
int fmx_extract_batch_dev(const fmx_index *idx, const uint64_t *d_rows, uint64_t nrows, uint64_t len,
                          int forward, void *d_out_syms, uint64_t *d_out_len, uint64_t *d_out_next,
                          void *stream) {
  CHECK_IDX(idx);
  if (len > 0xFFFFFFFFull) return fail(FMX_ERR_ARG, "len is too large");
  if (nrows && (!d_rows || (len && !d_out_syms))) return fail(FMX_ERR_ARG, "NULL argument");
  return fmx_launch_extract(idx, d_rows, nrows, (uint32_t)len, forward, d_out_syms, d_out_len,
                            d_out_next, (hipStream_t)stream);
}
int fmx_extract_batch(const fmx_index *idx, const uint64_t *rows, uint64_t nrows, uint64_t len,
                      int forward, void *out_syms, uint64_t *out_len, uint64_t *out_next) {
  CHECK_IDX(idx);
  if (len > 0xFFFFFFFFull) return fail(FMX_ERR_ARG, "len is too large");
  if (nrows == 0) return FMX_OK;
  if (!rows || (len && !out_syms)) return fail(FMX_ERR_ARG, "NULL argument");
  HostCall hc;
  const size_t b_rows = (size_t)nrows * 8, b_sym = (size_t)nrows * len * idx->sym_bytes;
  FMX_HIP(hc.open(idx->device, 3 * HostCall::pad(b_rows) + HostCall::pad(b_sym ? b_sym : 8)));
  uint64_t *d_rows = hc.take<uint64_t>(b_rows), *d_len = hc.take<uint64_t>(b_rows);
  uint64_t *d_next = hc.take<uint64_t>(b_rows);
  uint8_t *d_sym = hc.take<uint8_t>(b_sym);
  hipStream_t S = hc.sx->st;
  CallStatus cs(hc.sx);
  FMX_HIP(hipMemsetAsync(status_dev(hc.sx), 0, 4, S));
  FMX_HIP(hipMemcpyAsync(d_rows, rows, b_rows, hipMemcpyHostToDevice, S));
  if (b_sym) FMX_HIP(hipMemsetAsync(d_sym, 0, b_sym, S));   // slots past a piece end read as 0 on the host side
  if (int rc = fmx_launch_extract(idx, d_rows, nrows, (uint32_t)len, forward, d_sym, d_len, d_next, S)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  if (b_sym) FMX_HIP(hipMemcpyAsync(out_syms, d_sym, b_sym, hipMemcpyDeviceToHost, S));
  if (out_len) FMX_HIP(hipMemcpyAsync(out_len, d_len, b_rows, hipMemcpyDeviceToHost, S));
  if (out_next) FMX_HIP(hipMemcpyAsync(out_next, d_next, b_rows, hipMemcpyDeviceToHost, S));
  return finish_host_call(hc.sx);
}

// one trait method per call
uint64_t fmx_get_l(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_get_l_batch(idx, &i, 1, &o) ? ~0ull : o; }
uint64_t fmx_lf_map(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_lf_map_batch(idx, &i, 1, &o) ? ~0ull : o; }
uint64_t fmx_lf_map2(const fmx_index *idx, uint64_t c, uint64_t i) { uint64_t o = ~0ull; return fmx_lf_map2_batch(idx, &c, &i, 1, &o) ? ~0ull : o; }
uint64_t fmx_get_f(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_get_f_batch(idx, &i, 1, &o) ? ~0ull : o; }
uint64_t fmx_fl_map(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_fl_map_batch(idx, &i, 1, &o) ? ~0ull : o; }
uint64_t fmx_get_sa(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_get_sa_batch(idx, &i, 1, &o) ? ~0ull : o; }

// ---------------------------------------------------------------------------
// multi-pieces (multi_pieces.rs)
// ---------------------------------------------------------------------------
uint64_t fmx_pieces_count(const fmx_index *idx) {
  return idx && idx->kind == FMX_KIND_MULTI ? (idx->is_wide ? idx->wide.doc_count : idx->dev.doc_count) : 0;
}
int fmx_piece_id_batch_dev(const fmx_index *idx, const uint64_t *d_i, uint64_t k, uint64_t *d_out, void *stream) {
  CHECK_IDX(idx);
  if (idx->kind != FMX_KIND_MULTI) return fail(FMX_ERR_ARG, "piece_id needs a multi-pieces index");
  return fmx_launch_scalar(idx, 6, nullptr, d_i, k, d_out, (hipStream_t)stream);
}
int fmx_piece_id_batch(const fmx_index *idx, const uint64_t *i, uint64_t k, uint64_t *out) {
  if (idx && idx->kind != FMX_KIND_MULTI) return fail(FMX_ERR_ARG, "piece_id needs a multi-pieces index");
  return scalar_host(idx, 6, nullptr, i, k, out);
}
uint64_t fmx_piece_id(const fmx_index *idx, uint64_t i) { uint64_t o = ~0ull; return fmx_piece_id_batch(idx, &i, 1, &o) ? ~0ull : o; }
int fmx_match_counts_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                         int prefix_only, uint64_t *d_out_count, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_match_counts(idx, d_s, d_e, npat, prefix_only, d_out_count, (hipStream_t)stream);
}
int fmx_match_rows_dev(const fmx_index *idx, const uint64_t *d_s, const uint64_t *d_e, uint64_t npat,
                       int prefix_only, const uint64_t *d_out_off, uint64_t *d_out_rows, void *stream) {
  CHECK_IDX(idx);
  return fmx_launch_match_rows(idx, d_s, d_e, npat, prefix_only, d_out_off, d_out_rows, (hipStream_t)stream);
}
int fmx_match_counts(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                     int prefix_only, uint64_t *out_count) {
  CHECK_IDX(idx);
  if (npat == 0) return FMX_OK;
  if (!s || !e || !out_count) return fail(FMX_ERR_ARG, "NULL argument");
  HostCall hc;
  const size_t bk = (size_t)npat * 8;
  FMX_HIP(hc.open(idx->device, 3 * HostCall::pad(bk)));
  uint64_t *d_s = hc.take<uint64_t>(bk), *d_e = hc.take<uint64_t>(bk), *d_c = hc.take<uint64_t>(bk);
  hipStream_t S = hc.sx->st;
  FMX_HIP(hipMemcpyAsync(d_s, s, bk, hipMemcpyHostToDevice, S));
  FMX_HIP(hipMemcpyAsync(d_e, e, bk, hipMemcpyHostToDevice, S));
  if (int rc = fmx_launch_match_counts(idx, d_s, d_e, npat, prefix_only, d_c, S)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  FMX_HIP(hipMemcpyAsync(out_count, d_c, bk, hipMemcpyDeviceToHost, S));
  FMX_HIP(hipStreamSynchronize(S));
  return FMX_OK;
}
int fmx_match_rows(const fmx_index *idx, const uint64_t *s, const uint64_t *e, uint64_t npat,
                   int prefix_only, const uint64_t *out_off, uint64_t *out_rows) {
  CHECK_IDX(idx);
  if (npat == 0) return FMX_OK;
  if (!s || !e || !out_off) return fail(FMX_ERR_ARG, "NULL argument");
  if (out_off[npat] == 0) return FMX_OK;
  if (!out_rows) return fail(FMX_ERR_ARG, "out_rows is NULL");
  uint64_t total = out_off[npat];
  HostCall hc;
  const size_t bk = (size_t)npat * 8, br = (size_t)total * 8;
  FMX_HIP(hc.open(idx->device, 2 * HostCall::pad(bk) + HostCall::pad(bk + 8) + HostCall::pad(br)));
  uint64_t *d_s = hc.take<uint64_t>(bk), *d_e = hc.take<uint64_t>(bk);
  uint64_t *d_o = hc.take<uint64_t>(bk + 8), *d_r = hc.take<uint64_t>(br);
  hipStream_t S = hc.sx->st;
  FMX_HIP(hipMemcpyAsync(d_s, s, bk, hipMemcpyHostToDevice, S));
  FMX_HIP(hipMemcpyAsync(d_e, e, bk, hipMemcpyHostToDevice, S));
  FMX_HIP(hipMemcpyAsync(d_o, out_off, bk + 8, hipMemcpyHostToDevice, S));
  if (int rc = fmx_launch_match_rows(idx, d_s, d_e, npat, prefix_only, d_o, d_r, S)) {
    (void)hipStreamSynchronize(S);
    return rc;
  }
  FMX_HIP(hipMemcpyAsync(out_rows, d_r, br, hipMemcpyDeviceToHost, S));
  FMX_HIP(hipStreamSynchronize(S));
  return FMX_OK;
}

// ---------------------------------------------------------------------------
// export / verification
// ---------------------------------------------------------------------------
int fmx_export_bwt(const fmx_index *idx, void *host_out) {
  CHECK_IDX(idx);
  if (idx->n == 0) return FMX_OK;
  Scratch sc;
  void *d;
  FMX_HIP(sc.get(&d, idx->n * idx->sym_bytes));
  if (int rc = fmx_launch_export_l(idx, d, 0)) return rc;
  FMX_HIP(hipDeviceSynchronize());
  FMX_HIP(hipMemcpy(host_out, d, idx->n * idx->sym_bytes, hipMemcpyDeviceToHost));
  return FMX_OK;
}
int fmx_export_cs(const fmx_index *idx, uint64_t *host_out) {
  if (!idx || !idx->h_cs) return fail(FMX_ERR_ARG);
  memcpy(host_out, idx->h_cs, (idx->max_character + 1) * sizeof(uint64_t));
  return FMX_OK;
}
// SOSampledSuffixArray's payload (sample.rs:33-37): SA[k << level] for k = 0 .. ((n-1) >> level).  With
// text-order sampling the stored samples are other rows', so the values are computed: get_sa of
// those rows (same numbers, whatever is sampled inside).
int fmx_export_sa_samples64(const fmx_index *idx, uint64_t *host_out) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  if (idx->is_wide && !idx->wide.walk && !idx->wide.phase) {
    FMX_HIP(hipMemcpy(host_out, idx->wide.samples, idx->nsamples * 8, hipMemcpyDeviceToHost));
    return FMX_OK;
  }
  if (idx->is_wide) {
    // text-order samples (walk records): the reference's samples SA[k << level] are computed, get_sa of those rows,
    // in chunks of 2^22 rows through the device entry point
    const uint64_t k = idx->nsamples, step = 1ull << idx->wide.sa_level, chunk = 1ull << 22;
    const uint64_t cap = k < chunk ? k : chunk;
    uint64_t *d_rows = nullptr, *d_vals = nullptr;
    uint64_t *h_buf = (uint64_t *)malloc((size_t)(cap ? cap : 1) * 8);
    if (!h_buf) return fail(FMX_ERR_ARG, "out of host memory");
    hipError_t e = fmx_dev_malloc((void **)&d_rows, (size_t)(cap ? cap : 1) * 8);
    if (e == hipSuccess) e = fmx_dev_malloc((void **)&d_vals, (size_t)(cap ? cap : 1) * 8);
    int rc = e == hipSuccess ? FMX_OK : fmx_hip_fail(e, "fmx_dev_malloc(export scratch)", __LINE__);
    for (uint64_t a = 0; a < k && rc == FMX_OK; a += chunk) {
      const uint64_t m = k - a < chunk ? k - a : chunk;
      for (uint64_t j = 0; j < m; j++) h_buf[j] = (a + j) * step;
      if ((e = hipMemcpy(d_rows, h_buf, (size_t)m * 8, hipMemcpyHostToDevice)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy", __LINE__); break; }
      if ((rc = fmx_launch_scalar(idx, 3, nullptr, d_rows, m, d_vals, 0)) != FMX_OK) break;
      if ((e = hipMemcpy(host_out + a, d_vals, (size_t)m * 8, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy", __LINE__); break; }
    }
    if (d_rows) (void)hipFree(d_rows);
    if (d_vals) (void)hipFree(d_vals);
    free(h_buf);
    return rc;
  }
  uint32_t *tmp = (uint32_t *)malloc((size_t)(idx->nsamples ? idx->nsamples : 1) * 4);
  if (!tmp) return fail(FMX_ERR_ARG, "out of host memory");
  const int rc = fmx_export_sa_samples(idx, tmp);
  if (rc == FMX_OK)
    for (uint64_t j = 0; j < idx->nsamples; j++) host_out[j] = tmp[j];
  free(tmp);
  return rc;
}
int fmx_export_sa_samples(const fmx_index *idx, uint32_t *host_out) {
  CHECK_IDX(idx);
  if (idx->dev.sa_level == FMX_NO_LOCATE) return fail(FMX_ERR_NO_LOCATE);
  if (idx->is_wide) return fail(FMX_ERR_UNSUPPORTED, "the samples of an index with n >= 2^32 need fmx_export_sa_samples64");
  if (!idx->dev.phase) {
    FMX_HIP(hipMemcpy(host_out, idx->dev.samples, idx->nsamples * 4, hipMemcpyDeviceToHost));
    return FMX_OK;
  }
  // text-order sampling: the values are computed, get_sa(k << level), in chunks of 2^22 rows through the
  // device entry point (16 B of device scratch + 16 B of pinned-free host staging per row of a CHUNK, not
  // of the whole array; nothing that can throw crosses the C ABI)
  const uint64_t k = idx->nsamples, step = 1ull << idx->dev.sa_level, chunk = 1ull << 22;
  const uint64_t cap = k < chunk ? k : chunk;
  uint64_t *d_rows = nullptr, *d_vals = nullptr;
  uint64_t *h_buf = (uint64_t *)malloc((size_t)cap * 8);
  if (!h_buf) return fail(FMX_ERR_ARG, "out of host memory");
  hipError_t e = fmx_dev_malloc((void **)&d_rows, (size_t)cap * 8);
  if (e == hipSuccess) e = fmx_dev_malloc((void **)&d_vals, (size_t)cap * 8);
  int rc = e == hipSuccess ? FMX_OK : fmx_hip_fail(e, "fmx_dev_malloc(export scratch)", __LINE__);
  for (uint64_t a = 0; a < k && rc == FMX_OK; a += chunk) {
    const uint64_t m = k - a < chunk ? k - a : chunk;
    for (uint64_t j = 0; j < m; j++) h_buf[j] = (a + j) * step;
    if ((e = hipMemcpy(d_rows, h_buf, (size_t)m * 8, hipMemcpyHostToDevice)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy", __LINE__); break; }
    if ((rc = fmx_launch_scalar(idx, 3, nullptr, d_rows, m, d_vals, 0)) != FMX_OK) break;
    if ((e = hipMemcpy(h_buf, d_vals, (size_t)m * 8, hipMemcpyDeviceToHost)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy", __LINE__); break; }
    for (uint64_t j = 0; j < m; j++) host_out[a + j] = (uint32_t)h_buf[j];
  }
  if (d_rows) (void)hipFree(d_rows);
  if (d_vals) (void)hipFree(d_vals);
  free(h_buf);
  return rc;
}
int fmx_export_sa(const fmx_index *idx, uint32_t *host_out) {
  CHECK_IDX(idx);
  if (idx->is_wide) return fail(FMX_ERR_UNSUPPORTED, "the suffix array of an index with n >= 2^32 does not fit 32 bits");
  if (!idx->d_sa) return fail(FMX_ERR_ARG, "index was built without FMX_FLAG_KEEP_SA");
  if (idx->n) FMX_HIP(hipMemcpy(host_out, idx->d_sa, idx->n * 4, hipMemcpyDeviceToHost));
  return FMX_OK;
}
int fmx_verify_sa(const fmx_index *idx, uint64_t *violations) {
  CHECK_IDX(idx);
  if (idx->is_wide) {
    if (!idx->d_sa64 || !idx->d_text) return fail(FMX_ERR_ARG, "index was built without FMX_FLAG_KEEP_SA");
    return fmxw_verify_sa(idx, violations);
  }
  if (!idx->d_sa || !idx->d_text) return fail(FMX_ERR_ARG, "index was built without FMX_FLAG_KEEP_SA");
  return fmx_verify_sa_impl(idx, violations);
}

// ---------------------------------------------------------------------------
// flat index file (the reference has no public on-disk format: its serde derives sit on
// private backend structs only, fm_index.rs:13 / rlfmi.rs:15 / sample.rs:12)
//   header | FmxDev (pointers are rewritten on load) | cs[] | device arrays in a fixed order
// ---------------------------------------------------------------------------
namespace {
struct FileHeader {
  char magic[8];           // "FMXIDX01"
  uint32_t version, dev_struct_bytes;
  uint64_t n, max_character, nsamples, runs, bytes;
  uint32_t sym_bytes, sym_bytes_abi, kind, level_requested;
};
struct Blob { const void **field; uint64_t bytes; };
// every device array the query path reads, in file order
int enumerate_blobs(FmxDev &d, uint64_t nsamples, Blob *out) {
  int k = 0;
  for (uint32_t l = 0; l < d.bw.nlevels; l++) {
    out[k++] = {(const void **)&d.bw.lv[l].rec, (uint64_t)d.bw.lv[l].nrec * 128};
    out[k++] = {(const void **)&d.bw.lv[l].C, 64};
    if (d.bw.lv[l].sel) {
      const uint64_t ncode = d.bw.lv[l].fmt == 3 ? 8 : 16;
      out[k++] = {(const void **)&d.bw.lv[l].sel, ((uint64_t)d.bw.len / FMX_WSEL_STEP + 2 * ncode + 2) * 4};
      out[k++] = {(const void **)&d.bw.lv[l].selmeta, 48 * 4};
    }
  }
  out[k++] = {(const void **)&d.K, ((uint64_t)d.max_character + 1) * 4};
  out[k++] = {(const void **)&d.cs, ((uint64_t)d.max_character + 1) * 4};
  if (d.kind == FMX_KIND_MULTI) out[k++] = {(const void **)&d.doc, (uint64_t)d.doc_count * 4};
  if (d.sa_level != FMX_NO_LOCATE) out[k++] = {(const void **)&d.samples, (nsamples + 4) * 4};
  if (d.phase) out[k++] = {(const void **)&d.phase, ((uint64_t)d.n / (3u * (32u / d.sa_level)) + 1) * 16};
  if (d.kind == FMX_KIND_RLFM) {
    out[k++] = {(const void **)&d.b.rec, (uint64_t)d.b.nrec * 128};
    out[k++] = {(const void **)&d.b.sel, (uint64_t)d.b.nsel * 4};
    out[k++] = {(const void **)&d.bp.rec, (uint64_t)d.bp.nrec * 128};
    out[k++] = {(const void **)&d.bp.sel, (uint64_t)d.bp.nsel * 4};
    if (d.b.dsel) out[k++] = {(const void **)&d.b.dsel, (((uint64_t)d.b.ones + (1ull << d.b.dsel_shift) - 1) >> d.b.dsel_shift) * 16};
    if (d.bp.dsel) out[k++] = {(const void **)&d.bp.dsel, (((uint64_t)d.bp.ones + (1ull << d.bp.dsel_shift) - 1) >> d.bp.dsel_shift) * 16};
    if (d.b.pos) out[k++] = {(const void **)&d.b.pos, (uint64_t)d.b.ones * 4};
    if (d.bp.pos) out[k++] = {(const void **)&d.bp.pos, (uint64_t)d.bp.ones * 4};
    if (d.lfrun) out[k++] = {(const void **)&d.lfrun, (uint64_t)d.b.ones * 4};       // one entry per run
  }
  if (d.pair_rec) out[k++] = {(const void **)&d.pair_rec, ((uint64_t)d.n / 128 + 1) * 128};
  if (d.kmer) out[k++] = {(const void **)&d.kmer, (1ull << (d.kmer_bits * d.kmer_k)) * 8};
  return k;
}
// A file is only trusted as far as its own numbers agree: every field that sizes an array or is
// used as an array bound is checked against the header and the builder's formulas BEFORE
// enumerate_blobs / any allocation looks at it, and the blob sizes must add up to the file size.
const char *validate_loaded(const FileHeader &h, const FmxDev &d) {
  if (h.kind > FMX_KIND_MULTI || d.kind != h.kind) return "kind";
  if ((h.sym_bytes != 1 && h.sym_bytes != 2 && h.sym_bytes != 4) || d.sym_bytes != h.sym_bytes) return "sym_bytes";
  if (h.sym_bytes_abi != 1 && h.sym_bytes_abi != 2 && h.sym_bytes_abi != 4 && h.sym_bytes_abi != 8) return "sym_bytes_abi";
  if (h.n >= 0xFFFFFFF0ull || d.n != h.n) return "n";
  if (h.max_character == 0 || h.max_character >= (1ull << 26) || d.max_character != h.max_character) return "max_character";
  if (h.runs > h.n) return "runs";
  const FmxMwm &w = d.bw;
  if (w.nlevels == 0 || w.nlevels > FMX_MAX_LEVELS) return "nlevels";
  if (w.len != (h.kind == FMX_KIND_RLFM ? h.runs : h.n)) return "wavelet length";
  for (uint32_t l = 0; l < w.nlevels; l++) {
    const FmxLevel &L = w.lv[l];
    if (L.fmt != 3 && L.fmt != 4) return "level format";
    if (L.shift >= 32 || (L.fmt == 3 ? (L.mask != 1u && L.mask != 3u && L.mask != 7u) : L.mask != 15u)) return "level shift / mask";
    if (L.nrec != w.len / (L.fmt == 3 ? 256u : 128u) + 1u) return "level records";
  }
  if (d.sa_level != FMX_NO_LOCATE) {
    if (d.sa_level >= 32 || h.n == 0) return "sa_level";
    if (h.nsamples != ((h.n - 1) >> d.sa_level) + 1 || d.nsamples != h.nsamples) return "nsamples";
  } else if (h.nsamples != 0) return "nsamples";
  if (d.phase && (d.sa_level == FMX_NO_LOCATE || d.sa_level < 1 || d.sa_level > FMX_PHASE_MAX_LEVEL)) return "text-order sampling";
  if (d.kmer) {
    if (d.kmer_k == 0 || d.kmer_bits == 0 || d.kmer_bits > 8 || d.kmer_bits * d.kmer_k > 24) return "k-mer table";
  }
  if (d.kind == FMX_KIND_MULTI) {
    if (d.doc_count > h.n || d.first_row > h.n) return "pieces";
  }
  if (d.kind == FMX_KIND_RLFM) {
    const FmxBits *v[2] = {&d.b, &d.bp};
    for (int t = 0; t < 2; t++) {
      if (v[t]->len != h.n || v[t]->ones != h.runs) return "bit vector length";
      if (v[t]->nrec != v[t]->len / FMX_BITS_PER_REC + 1u) return "bit vector records";
      if (v[t]->nsel != v[t]->ones / FMX_SEL_STEP + 2u) return "select hints";
      if (v[t]->dsel && (v[t]->dsel_shift < 3 || v[t]->dsel_shift > 6)) return "select blocks";
    }
  }
  if (d.pair_rec && (d.pair_row0 > h.n || d.pair_row1 > h.n)) return "pair index";
  if (d.lfrun && (d.kind != FMX_KIND_RLFM || d.sa_level == FMX_NO_LOCATE)) return "run LF table";
  return nullptr;
}
const size_t kChunk = 64u << 20;
const uint32_t kFileVersion = 10;   // 10: wide RLFM indexes (FmxWideDev::b / bp / lfrun); 9: FmxDev::walk (walk records: a presence flag only, rebuilt by fmx_load); 2: select hints every 64 ones (was 512); 3: positions of sparse vectors; 4: wavelet select hints; 5: dense select blocks; 6: pointer fields written as presence flags, fields validated on load; 7: select blocks of 8 / 16 / 32 / 64 ones by density; 8: text-order sampling (phase pieces)
}  // namespace

// wide indexes (n >= 2^32 - 16): header (dev_struct_bytes carries kWideMark) | FmxWideDev with presence flags for
// pointers | cs[] | records | bases | samples
namespace {
const uint32_t kWideMark = 0x80000000u;
struct WideBlobs { const void **field[20 + 2 * FMXW_MAX_LEVELS]; uint64_t bytes[20 + 2 * FMXW_MAX_LEVELS]; int n; };
WideBlobs wide_blobs(FmxWideDev &w, uint64_t nsamples) {
  WideBlobs b;
  b.n = 0;
  if (w.generic) {
    for (uint32_t l = 0; l < w.nlevels && l < FMXW_MAX_LEVELS; l++) {
      b.field[b.n] = (const void **)&w.lv[l].rec;  b.bytes[b.n++] = (uint64_t)w.lv[l].nrec * 128ull;
      b.field[b.n] = (const void **)&w.lv[l].base; b.bytes[b.n++] = (uint64_t)w.nsb * 128ull;
    }
    b.field[b.n] = (const void **)&w.K;  b.bytes[b.n++] = ((uint64_t)w.max_character + 1) * 8ull;
    b.field[b.n] = (const void **)&w.cs; b.bytes[b.n++] = ((uint64_t)w.max_character + 1) * 8ull;
  } else {
    b.field[b.n] = (const void **)&w.rec;  b.bytes[b.n++] = (w.n / 256u + 1u) * 128ull;
    b.field[b.n] = (const void **)&w.base; b.bytes[b.n++] = (uint64_t)w.nsb * 64ull;
  }
  if (w.sa_level != FMX_NO_LOCATE) { b.field[b.n] = (const void **)&w.samples; b.bytes[b.n++] = nsamples * 8ull; }
  if (w.walk) {   // text-order samples: the walk records hold the phases (this engine has no phase pieces to derive them from)
    b.field[b.n] = (const void **)&w.walk;  b.bytes[b.n++] = (w.n / FMX_WALK_ROWS + 1u) * 128ull;
    b.field[b.n] = (const void **)&w.wbase; b.bytes[b.n++] = (uint64_t)w.nwsb * 128ull;
  }
  if (w.phase && w.kind != FMX_KIND_RLFM) {         // text-order samples of a generic FM / multi-pieces index
    b.field[b.n] = (const void **)&w.phase; b.bytes[b.n++] = (w.n / (3u * (32u / w.sa_level)) + 1u) * 16ull;
    b.field[b.n] = (const void **)&w.pbase; b.bytes[b.n++] = (uint64_t)w.npsb * 8ull;
  }
  if (w.kind == FMX_KIND_RLFM) {
    FmxWideBits *v[2] = {&w.b, &w.bp};
    for (int t = 0; t < 2; t++) {
      b.field[b.n] = (const void **)&v[t]->rec;  b.bytes[b.n++] = (uint64_t)v[t]->nrec * 128ull;
      b.field[b.n] = (const void **)&v[t]->base; b.bytes[b.n++] = (uint64_t)v[t]->nsb * 8ull;
      b.field[b.n] = (const void **)&v[t]->sel;  b.bytes[b.n++] = v[t]->nsel * 4ull;
      if (v[t]->pos) { b.field[b.n] = (const void **)&v[t]->pos; b.bytes[b.n++] = v[t]->ones * 8ull; }
    }
    if (w.lfrun) { b.field[b.n] = (const void **)&w.lfrun; b.bytes[b.n++] = w.slen * 8ull; }
    if (w.phase) {
      b.field[b.n] = (const void **)&w.phase; b.bytes[b.n++] = (w.n / (3u * (32u / w.sa_level)) + 1u) * 16ull;
      b.field[b.n] = (const void **)&w.pbase; b.bytes[b.n++] = (uint64_t)w.npsb * 8ull;
    }
  }
  if (w.kind == FMX_KIND_MULTI) { b.field[b.n] = (const void **)&w.doc; b.bytes[b.n++] = w.doc_count * 4ull; }
  return b;
}
}  // namespace
static int save_wide(const fmx_index *idx, FILE *f, FileHeader &h) {
  h.dev_struct_bytes = (uint32_t)sizeof(FmxWideDev) | kWideMark;
  FmxWideDev src = idx->wide, wfile = idx->wide;
  WideBlobs bs = wide_blobs(src, idx->nsamples), bf = wide_blobs(wfile, idx->nsamples);
  for (int b = 0; b < bf.n; b++) *bf.field[b] = (const void *)(uintptr_t)1;
  if (wfile.sa_level == FMX_NO_LOCATE) wfile.samples = nullptr;
  wfile.status = nullptr;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(&wfile, sizeof wfile, 1, f) == 1 &&
            fwrite(idx->h_cs, 8, idx->max_character + 1, f) == idx->max_character + 1;
  std::string buf(kChunk, '\0');
  for (int b = 0; ok && b < bs.n; b++) {
    const uint8_t *p = (const uint8_t *)*bs.field[b];
    for (uint64_t o = 0; ok && o < bs.bytes[b]; o += kChunk) {
      const size_t m = (size_t)(bs.bytes[b] - o < kChunk ? bs.bytes[b] - o : kChunk);
      if (hipMemcpy(&buf[0], p + o, m, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
      ok = fwrite(&buf[0], 1, m, f) == m;
    }
  }
  return ok ? FMX_OK : FMX_ERR_ARG;
}

int fmx_save(const fmx_index *idx, const char *path) {
  CHECK_IDX(idx);
  if (!path) return fail(FMX_ERR_ARG, "path is NULL");
  FILE *f = fopen(path, "wb");
  if (!f) return fail(FMX_ERR_ARG, "cannot open file for writing");
  FileHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "FMXIDX01", 8);
  h.version = kFileVersion;
  h.dev_struct_bytes = (uint32_t)sizeof(FmxDev);
  h.n = idx->n; h.max_character = idx->max_character; h.nsamples = idx->nsamples;
  h.runs = idx->runs; h.bytes = idx->bytes;
  h.sym_bytes = idx->sym_bytes; h.sym_bytes_abi = idx->sym_bytes_abi; h.kind = idx->kind;
  h.level_requested = idx->level_requested;
  if (idx->is_wide) {
    const int rc = save_wide(idx, f, h);
    const bool closed = fclose(f) == 0;
    return rc == FMX_OK && closed ? FMX_OK : fail(FMX_ERR_ARG, "write failed");
  }
  FmxDev d = idx->dev;
  Blob blobs[64];
  int nb = enumerate_blobs(d, idx->nsamples, blobs);
  // the struct that goes into the file keeps which arrays exist, not where they were: every
  // pointer field is written as 1 (present) or 0
  FmxDev dfile = idx->dev;
  {
    Blob fb[64];
    const int nf = enumerate_blobs(dfile, idx->nsamples, fb);
    for (int b = 0; b < nf; b++) *fb[b].field = (const void *)(uintptr_t)1;
    dfile.status = nullptr;
    // walk records are derived from the level-0 records and the phase pieces: the file only says that the index
    // had them, fmx_load rebuilds them
    dfile.walk = idx->dev.walk ? (const uint4 *)(uintptr_t)1 : nullptr;
  }
  bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(&dfile, sizeof dfile, 1, f) == 1 &&
            fwrite(idx->h_cs, 8, idx->max_character + 1, f) == idx->max_character + 1;
  std::string buf(kChunk, '\0');
  for (int b = 0; ok && b < nb; b++) {
    const uint8_t *src = (const uint8_t *)*blobs[b].field;
    for (uint64_t o = 0; ok && o < blobs[b].bytes; o += kChunk) {
      size_t m = (size_t)(blobs[b].bytes - o < kChunk ? blobs[b].bytes - o : kChunk);
      if (hipMemcpy(&buf[0], src + o, m, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
      ok = fwrite(&buf[0], 1, m, f) == m;
    }
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? FMX_OK : fail(FMX_ERR_ARG, "write failed");
}

// a wide index file: every size / bound field is checked against the header and the builder's formulas, the array
// sizes against the file size, before anything is allocated or indexed
static int load_wide(FILE *f, const FileHeader &h, int device, fmx_index *idx) {
  FmxWideDev w;
  if (fread(&w, sizeof w, 1, f) != 1) return fail(FMX_ERR_ARG, "truncated index file");
  const bool locate = w.sa_level != FMX_NO_LOCATE;
  const char *bad = nullptr;
  const bool rl = h.kind == FMX_KIND_RLFM, mp = h.kind == FMX_KIND_MULTI;
  const uint64_t slen = rl ? h.runs : h.n;          // entries of the wavelet levels: run heads (RLFM) or the BWT
  if ((h.kind != FMX_KIND_FM && !rl && !mp) || w.kind != h.kind || (h.sym_bytes != 1 && h.sym_bytes != 2 && h.sym_bytes != 4) ||
      (h.sym_bytes_abi != h.sym_bytes && !(h.sym_bytes_abi == 8 && h.sym_bytes == 4)))
    bad = "kind / symbol width";
  else if (h.n < 2 || h.n >= (1ull << 38) || w.n != h.n) bad = "n";
  else if (h.max_character == 0 || h.max_character >= (1ull << 26) || w.max_character != h.max_character ||
           (h.sym_bytes < 4 && h.max_character >= (1ull << (8 * h.sym_bytes))))
    bad = "max_character";
  else if (w.generic != ((h.max_character > 7 || h.sym_bytes != 1 || rl || mp) ? 1u : 0u) || (w.generic && w.sym_bytes != h.sym_bytes))
    bad = "engine";
  else if (mp ? (w.doc_count == 0 || w.doc_count > h.n || w.doc_count >= (1ull << 32) || w.first_row >= h.n || !w.doc)
              : (w.doc != nullptr || w.doc_count != 0))
    bad = "pieces";
  else if (rl ? (h.runs == 0 || h.runs > h.n || w.slen != h.runs) : (h.runs != 0 || w.slen != 0 || w.lfrun)) bad = "runs";
  else if (w.sb_shift < 8 || w.sb_shift > 31 || w.nsb != (uint32_t)(slen >> w.sb_shift) + 1u) bad = "superblocks";
  else if (locate && (w.sa_level >= 63 || h.nsamples != ((h.n - 1) >> w.sa_level) + 1)) bad = "sampling level";
  else if (!w.generic && (!w.rec || !w.base || (locate && !w.samples))) bad = "array presence";
  else if ((w.walk != nullptr) != (w.wbase != nullptr)) bad = "walk records";
  else if (rl && (w.lfrun != nullptr) && !locate) bad = "run table";
  else if ((w.phase != nullptr) != (w.pbase != nullptr) ||
           (w.phase && (!w.generic || !locate || w.sa_level < 1 || w.sa_level > FMX_PHASE_MAX_LEVEL ||
                        (w.psb_shift != FMXW_PHASE_SB_SHIFT && w.psb_shift != FMXW_PHASE_SB_SHIFT_TEST) ||
                        w.npsb != (uint32_t)((h.n / (3u * (32u / w.sa_level))) >> w.psb_shift) + 1u)))
    bad = "text-order sampling";
  else if (w.walk && (w.generic || !locate || w.sa_level < 1 || w.sa_level > FMX_WALK_MAX_LEVEL || h.sym_bytes != 1 ||
                      h.max_character > FMX_WALK_MAX_CHARACTER ||
                      (w.wsb_shift != FMXW_WALK_SB_SHIFT && w.wsb_shift != FMXW_WALK_SB_SHIFT_TEST) ||
                      w.nwsb != (uint32_t)((h.n / FMX_WALK_ROWS) >> w.wsb_shift) + 1u))
    bad = "walk records";
  else if (w.generic) {
    // the levels must be the builder's split of max_bits (text.rs:61-63) -- the kernels index records, bases and K[]
    // by what these fields say
    const uint32_t L = 32u - (uint32_t)__builtin_clz((uint32_t)h.max_character);
    const uint32_t nlv = (L + 3) / 4, lo = L / nlv, extra = L % nlv;
    uint32_t shift = L;
    if (w.nlevels != nlv || nlv > FMXW_MAX_LEVELS || w.rec || w.base || !w.K || !w.cs || (locate && !w.samples)) bad = "levels";
    for (uint32_t l = 0; !bad && l < nlv; l++) {
      const uint32_t bits = lo + (l < extra ? 1u : 0u), fmt = bits == 4 ? 4u : 3u;
      shift -= bits;
      const FmxWideLevel &v = w.lv[l];
      if (v.fmt != fmt || v.shift != shift || v.mask != (1u << bits) - 1u || !v.rec || !v.base ||
          v.nrec != (uint32_t)((slen >> (fmt == 3 ? 8 : 7)) + 1u) || w.sb_shift < (fmt == 3 ? 8u : 7u))
        bad = "level fields";
    }
  }
  if (!bad && rl) {
    const FmxWideBits *v[2] = {&w.b, &w.bp};
    for (int t = 0; t < 2 && !bad; t++) {
      if (v[t]->len != h.n || v[t]->ones != h.runs) bad = "bit vector length";
      else if (v[t]->nrec != (uint32_t)(h.n / FMX_BITS_PER_REC + 1u)) bad = "bit vector records";
      else if ((v[t]->sb_shift != FMXW_BITS_SB_SHIFT && v[t]->sb_shift != FMXW_BITS_SB_SHIFT_TEST) ||
               v[t]->nsb != ((v[t]->nrec - 1u) >> v[t]->sb_shift) + 1u) bad = "bit vector superblocks";
      else if (v[t]->nsel != v[t]->ones / FMX_SEL_STEP + 2u) bad = "select hints";
      else if (!v[t]->rec || !v[t]->base || !v[t]->sel) bad = "bit vector arrays";
    }
  }
  if (bad) {
    char msg[128];
    snprintf(msg, sizeof msg, "corrupt index file: inconsistent %s", bad);
    return fail(FMX_ERR_ARG, msg);
  }
  idx->device = device;
  idx->n = h.n; idx->max_character = h.max_character; idx->nsamples = locate ? h.nsamples : 0; idx->runs = rl ? h.runs : 0;
  idx->sym_bytes = h.sym_bytes; idx->sym_bytes_abi = h.sym_bytes_abi; idx->kind = h.kind;
  idx->level_requested = h.level_requested;
  idx->h_cs = (uint64_t *)calloc(h.max_character + 1, 8);
  if (fread(idx->h_cs, 8, h.max_character + 1, f) != h.max_character + 1) return fail(FMX_ERR_ARG, "truncated index file");
  const bool has_walk = w.walk != nullptr;          // presence flags of the file, like every blob field
  WideBlobs bs = wide_blobs(w, idx->nsamples);
  uint64_t need = sizeof(FileHeader) + sizeof(FmxWideDev) + (h.max_character + 1) * 8;
  for (int b = 0; b < bs.n; b++) { *bs.field[b] = nullptr; need += bs.bytes[b]; }
  w.samples = nullptr; w.status = nullptr;
  if (!has_walk) { w.walk = nullptr; w.wbase = nullptr; w.nwsb = 0; w.wsb_shift = 0; }
  {
    const long at = ftell(f);
    fseek(f, 0, SEEK_END);
    const uint64_t have = (uint64_t)ftell(f);
    fseek(f, at, SEEK_SET);
    if (have != need) return fail(FMX_ERR_ARG, "corrupt index file: array sizes do not add up to the file size");
  }
  hipError_t e;
  if ((e = alloc_handle_words(idx)) != hipSuccess) return fmx_hip_fail(e, "handle resources", __LINE__);
  std::string buf(kChunk, '\0');
  for (int b = 0; b < bs.n; b++) {
    void *p = nullptr;
    if ((e = fmx_dev_malloc(&p, bs.bytes[b] ? bs.bytes[b] : 8)) != hipSuccess) return fmx_hip_fail(e, "hipMalloc", __LINE__);
    if (int rc = fmx_keep(idx, p, bs.bytes[b])) { (void)hipFree(p); return rc; }
    *bs.field[b] = p;
    for (uint64_t o = 0; o < bs.bytes[b]; o += kChunk) {
      const size_t m = (size_t)(bs.bytes[b] - o < kChunk ? bs.bytes[b] - o : kChunk);
      if (fread(&buf[0], 1, m, f) != m) return fail(FMX_ERR_ARG, "truncated index file");
      if ((e = hipMemcpy((uint8_t *)p + o, &buf[0], m, hipMemcpyHostToDevice)) != hipSuccess) return fmx_hip_fail(e, "hipMemcpy", __LINE__);
    }
  }
  w.status = idx->dev.status;
  idx->wide = w;
  idx->is_wide = 1;
  idx->dev.sa_level = w.sa_level;
  idx->dev.kind = h.kind;
  idx->dev.sym_bytes = h.sym_bytes;
  return FMX_OK;
}

int fmx_load(const char *path, int device, fmx_index **out) {
  if (!out || !path) return fail(FMX_ERR_ARG, "NULL argument");
  *out = nullptr;
  if (int rc = select_device(device)) return rc;
  DeviceGuard dg;
  FMX_HIP(dg.set(device));
  FILE *f = fopen(path, "rb");
  if (!f) return fail(FMX_ERR_ARG, "cannot open index file");
  FileHeader h;
  fmx_index *idx = (fmx_index *)calloc(1, sizeof(fmx_index));
  idx->layout = FMX_LAYOUT;
  int rc = FMX_OK;
  do {
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "FMXIDX01", 8) != 0 || h.version != kFileVersion) {
      rc = fail(FMX_ERR_ARG, "not an fmx index file (or another version)");
      break;
    }
    if (h.dev_struct_bytes == ((uint32_t)sizeof(FmxWideDev) | kWideMark)) {   // an index of the wide engine
      rc = load_wide(f, h, device, idx);
      break;
    }
    if (h.dev_struct_bytes != sizeof(FmxDev)) { rc = fail(FMX_ERR_ARG, "not an fmx index file (or another version)"); break; }
    if (fread(&idx->dev, sizeof(FmxDev), 1, f) != 1) { rc = fail(FMX_ERR_ARG, "truncated index file"); break; }
    if (const char *what = validate_loaded(h, idx->dev)) {
      char msg[128];
      snprintf(msg, sizeof msg, "corrupt index file: inconsistent %s", what);
      memset(&idx->dev, 0, sizeof idx->dev);     // nothing of it is a pointer this process owns
      rc = fail(FMX_ERR_ARG, msg);
      break;
    }
    idx->device = device;
    idx->n = h.n; idx->max_character = h.max_character; idx->nsamples = h.nsamples; idx->runs = h.runs;
    idx->sym_bytes = h.sym_bytes; idx->sym_bytes_abi = h.sym_bytes_abi; idx->kind = h.kind;
    idx->level_requested = h.level_requested;
    idx->h_cs = (uint64_t *)calloc(h.max_character + 1, 8);
    if (fread(idx->h_cs, 8, h.max_character + 1, f) != h.max_character + 1) { rc = fail(FMX_ERR_ARG, "truncated index file"); break; }
    Blob blobs[64];
    int nb = enumerate_blobs(idx->dev, idx->nsamples, blobs);
    uint64_t need = sizeof(FileHeader) + sizeof(FmxDev) + (h.max_character + 1) * 8;
    for (int b = 0; b < nb; b++) {
      *blobs[b].field = nullptr;  // presence markers of the file, not addresses
      need += blobs[b].bytes;
    }
    idx->dev.status = nullptr;
    const bool had_walk = idx->dev.walk != nullptr;   // a presence flag, like the blob fields
    idx->dev.walk = nullptr;
    {
      const long at = ftell(f);
      fseek(f, 0, SEEK_END);
      const uint64_t have = (uint64_t)ftell(f);
      fseek(f, at, SEEK_SET);
      if (have != need) { rc = fail(FMX_ERR_ARG, "corrupt index file: array sizes do not add up to the file size"); break; }
    }
    hipError_t e;
    if ((e = alloc_handle_words(idx)) != hipSuccess) {
      rc = fmx_hip_fail(e, "handle resources", __LINE__);
      break;
    }
    std::string buf(kChunk, '\0');
    for (int b = 0; rc == FMX_OK && b < nb; b++) {
      void *p = nullptr;
      if ((e = fmx_dev_malloc(&p, blobs[b].bytes ? blobs[b].bytes : 8)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMalloc", __LINE__); break; }
      if ((rc = fmx_keep(idx, p, blobs[b].bytes)) != FMX_OK) { (void)hipFree(p); break; }
      *blobs[b].field = p;
      for (uint64_t o = 0; o < blobs[b].bytes; o += kChunk) {
        size_t m = (size_t)(blobs[b].bytes - o < kChunk ? blobs[b].bytes - o : kChunk);
        if (fread(&buf[0], 1, m, f) != m) { rc = fail(FMX_ERR_ARG, "truncated index file"); break; }
        if ((e = hipMemcpy((uint8_t *)p + o, &buf[0], m, hipMemcpyHostToDevice)) != hipSuccess) { rc = fmx_hip_fail(e, "hipMemcpy", __LINE__); break; }
      }
    }
    // walk records: derived (fmx_make_walk_records checks eligibility itself -- a file cannot ask for them on an
    // index whose arrays do not support them)
    if (rc == FMX_OK && had_walk) rc = fmx_make_walk_records(idx);
  } while (0);
  fclose(f);
  if (rc != FMX_OK) { fmx_free(idx); return rc; }
  idx->bytes = h.bytes;  // as reported by the index that was saved
  *out = idx;
  return FMX_OK;
}

// ---------------------------------------------------------------------------
// fmx_replicate: a second handle with its own copy of every HBM array, on any device of the node (SURVEY section 8e:
// "index replicated on every GPU").  The arrays travel device to device (hipMemcpyPeer over xGMI; staged through a
// page-locked buffer when the runtime refuses the peer copy) -- no rebuild, no file, no host copy of the text.
// ---------------------------------------------------------------------------
namespace {
hipError_t copy_between_devices(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (dst_dev == src_dev) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice);
  // (direct copies over xGMI need peer access from the current device -- dst_dev -- to the source; already enabled, or
  // not available between the two, are both fine: hipMemcpyPeer stages through the host where it must)
  if (hipDeviceEnablePeerAccess(src_dev, 0) != hipSuccess) (void)hipGetLastError();
  hipError_t e = hipMemcpyPeer(dst, dst_dev, src, src_dev, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  // through the host, 64 MiB at a time (the current device is dst_dev: the caller's DeviceGuard)
  void *stage = nullptr;
  if ((e = hipHostMalloc(&stage, kChunk, hipHostMallocDefault)) != hipSuccess) return e;
  for (size_t o = 0; o < bytes && e == hipSuccess; o += kChunk) {
    const size_t m = bytes - o < kChunk ? bytes - o : kChunk;
    if ((e = hipSetDevice(src_dev)) != hipSuccess) break;
    e = hipMemcpy(stage, (const uint8_t *)src + o, m, hipMemcpyDeviceToHost);
    const hipError_t back = hipSetDevice(dst_dev);
    if (e == hipSuccess) e = back;
    if (e == hipSuccess) e = hipMemcpy((uint8_t *)dst + o, stage, m, hipMemcpyHostToDevice);
  }
  (void)hipHostFree(stage);
  return e;
}
// one array of the source index -> a new allocation on the current device, owned by `idx`
int replicate_array(fmx_index *idx, const void **field, const void *src, int src_dev, uint64_t bytes) {
  void *p = nullptr;
  hipError_t e = fmx_dev_malloc(&p, bytes ? bytes : 8);
  if (e != hipSuccess) return fmx_hip_fail(e, "fmx_dev_malloc(replica)", __LINE__);
  if (int rc = fmx_keep(idx, p, bytes)) { (void)hipFree(p); return rc; }
  *field = p;
  if ((e = copy_between_devices(p, idx->device, src, src_dev, (size_t)bytes)) != hipSuccess)
    return fmx_hip_fail(e, "copy of an index array between devices", __LINE__);
  return FMX_OK;
}
}  // namespace

int fmx_replicate(const fmx_index *src, int device, fmx_index **out) {
  if (!out) return fail(FMX_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!src) return fail(FMX_ERR_ARG, "index is NULL");
  if (src->layout != FMX_LAYOUT)
    return fail(FMX_ERR_ARG, "the index was made by another build of the library (rebuild libfmx*.so together)");
  if (int rc = select_device(device)) return rc;
  DeviceGuard dg;
  FMX_HIP(dg.set(device));
  // the source's pending work (a build that has just returned is complete; *_dev launches of the caller may not be)
  // does not write the index: nothing to wait for
  fmx_index *idx = (fmx_index *)calloc(1, sizeof(fmx_index));
  if (!idx) return fail(FMX_ERR_ARG, "out of host memory");
  *idx = *src;                                   // every scalar field; what a handle OWNS is reset below
  idx->device = device;
  idx->d_alloc = nullptr; idx->nalloc = 0; idx->cap_alloc = 0; idx->bytes = 0;
  idx->h_cs = nullptr; idx->d_text = nullptr; idx->d_sa = nullptr; idx->d_sa64 = nullptr;   // FMX_FLAG_KEEP_SA arrays stay with the source
  idx->flags &= ~FMX_FLAG_KEEP_SA;
  idx->timing = 0; idx->ev0 = nullptr; idx->ev1 = nullptr; idx->ev_valid = 0; idx->ev_series = nullptr; idx->series_n = 0;
  idx->dev.status = nullptr; idx->d_steps = nullptr;
  int rc = FMX_OK;
  do {
    if (src->h_cs) {
      idx->h_cs = (uint64_t *)calloc(src->max_character + 1, 8);
      if (!idx->h_cs) { rc = fail(FMX_ERR_ARG, "out of host memory"); break; }
      memcpy(idx->h_cs, src->h_cs, (src->max_character + 1) * 8);
    }
    // pointer fields of the copy are cleared BEFORE anything can fail: fmx_free must never see the source's arrays
    if (src->is_wide) {
      WideBlobs bc = wide_blobs(idx->wide, idx->nsamples);
      for (int b = 0; b < bc.n; b++) *bc.field[b] = nullptr;
      idx->wide.status = nullptr;
    } else {
      Blob bc[64];
      const int nc = enumerate_blobs(idx->dev, idx->nsamples, bc);
      for (int b = 0; b < nc; b++) *bc[b].field = nullptr;
      idx->dev.walk = nullptr;
    }
    hipError_t e;
    if ((e = alloc_handle_words(idx)) != hipSuccess) { rc = fmx_hip_fail(e, "handle resources", __LINE__); break; }
    if (src->is_wide) {
      FmxWideDev sw = src->wide;                 // (wide_blobs takes references into the struct it is given)
      WideBlobs bs = wide_blobs(sw, src->nsamples);
      // the same fields of the copy, in the same order: presence is decided by the SOURCE's pointers, so enumerate a struct
      // that still has them and redirect the field addresses into idx->wide
      FmxWideDev probe = src->wide;
      WideBlobs bp = wide_blobs(probe, src->nsamples);
      for (int b = 0; rc == FMX_OK && b < bs.n; b++) {
        const void **field = (const void **)((uint8_t *)&idx->wide + ((const uint8_t *)bp.field[b] - (const uint8_t *)&probe));
        rc = replicate_array(idx, field, *bs.field[b], src->device, bs.bytes[b]);
      }
      idx->wide.status = idx->dev.status;
    } else {
      FmxDev sd = src->dev, probe = src->dev;
      Blob bs[64], bp[64];
      const int nb = enumerate_blobs(sd, src->nsamples, bs);
      (void)enumerate_blobs(probe, src->nsamples, bp);
      for (int b = 0; rc == FMX_OK && b < nb; b++) {
        const void **field = (const void **)((uint8_t *)&idx->dev + ((const uint8_t *)bp[b].field - (const uint8_t *)&probe));
        rc = replicate_array(idx, field, *bs[b].field, src->device, bs[b].bytes);
      }
      // walk records: derived data, but copying 1.14 bytes per symbol beats deriving them again
      if (rc == FMX_OK && src->dev.walk)
        rc = replicate_array(idx, (const void **)&idx->dev.walk, src->dev.walk, src->device,
                             ((uint64_t)src->dev.n / FMX_WALK_ROWS + 1u) * 128u);
    }
  } while (0);
  if (rc != FMX_OK) { fmx_free(idx); return rc; }
  idx->build_ms = 0.0;                           // (idx->bytes = what was copied: the source's FMX_FLAG_KEEP_SA arrays are not)
  *out = idx;
  return FMX_OK;
}
