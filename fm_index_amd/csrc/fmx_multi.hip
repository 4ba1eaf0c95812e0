// fmx_multi.hip -- one batch over several replicas of an index, from ONE host caller (BASELINE
// config 5: "8M length-32 patterns sharded across GPUs, replicated index, gather of counts").
//
// The reference's driver (wrapper.rs:103-124) reads only immutable index state, so patterns are independent: pattern k
// of N goes to replica floor(k * G / N) -- contiguous shards [ceil(N r / G), ceil(N (r + 1) / G)), the partition
// fm_index_amd/sharding.py uses between processes -- and every replica searches its shard with the kernels of the
// single-device entry points.  For a HOST caller the gather is the layout of its own arrays: every shard's results
// are written in place at the shard's offset of out_s / out_e / out_count / out_pos, so the output is bit-identical
// to the one-device call whatever G is.  (Device-resident consumers in one process per GPU keep the RCCL all-gather
// of fm_index_amd/sharding.py.)
//
// One host thread per replica: shard 0 runs on the calling thread, shard r >= 1 on worker r of a process-wide pool --
// the workers are kept (a host-pointer call reuses its thread's streams, pinned staging and device scratch: a thread
// per call would rebuild them every time) and never joined (they park on a condition variable; the pool is leaked on
// purpose so that no destructor runs under a parked thread at process exit).
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "fmx_internal.h"

namespace {
struct Latch {
  std::mutex m;
  std::condition_variable cv;
  unsigned left = 0;
};
struct Job {
  std::function<int()> fn;
  int rc = FMX_OK;
  std::string err;
  Latch *latch = nullptr;
};
struct Worker {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job *> q;
  void loop() {
    for (;;) {
      Job *j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !q.empty(); });
        j = q.front();
        q.pop_front();
      }
      j->rc = j->fn();
      if (j->rc != FMX_OK) j->err = fmx_last_error();
      Latch *l = j->latch;          // (the job may be gone as soon as the latch opens)
      std::lock_guard<std::mutex> lk(l->m);
      if (--l->left == 0) l->cv.notify_all();
    }
  }
  void post(Job *j) {
    {
      std::lock_guard<std::mutex> lk(m);
      q.push_back(j);
    }
    cv.notify_one();
  }
};
struct Pool {
  std::mutex m;
  std::vector<Worker *> w;
  Worker *get(unsigned r) {
    std::lock_guard<std::mutex> lk(m);
    while (w.size() <= r) {
      Worker *nw = new Worker;
      std::thread(&Worker::loop, nw).detach();
      w.push_back(nw);
    }
    return w[r];
  }
};
Pool *pool() {
  static Pool *p = new Pool;   // leaked: see the header comment
  return p;
}

const unsigned kMaxReplicas = 64;

int fail_arg(const char *what) {
  fmx_set_error(FMX_ERR_ARG, what);
  return FMX_ERR_ARG;
}
// the handles of one call: all of one index (same text, same kind, same sampling)
int check_replicas(fmx_index *const *idx, uint32_t ndev) {
  if (!idx || ndev == 0) return fail_arg("no index handles");
  if (ndev > kMaxReplicas) return fail_arg("more than 64 replicas");
  for (uint32_t r = 0; r < ndev; r++) {
    if (!idx[r]) return fail_arg("index is NULL");
    if (idx[r]->layout != FMX_LAYOUT)
      return fail_arg("the index was made by another build of the library (rebuild libfmx*.so together)");
    if (idx[r]->n != idx[0]->n || idx[r]->kind != idx[0]->kind || idx[r]->max_character != idx[0]->max_character ||
        idx[r]->sym_bytes_abi != idx[0]->sym_bytes_abi || idx[r]->dev.sa_level != idx[0]->dev.sa_level)
      return fail_arg("the handles are not replicas of one index");
  }
  return FMX_OK;
}
// runs shard(r) for r = 0 .. ndev - 1 concurrently; the first failing shard's code and message (in shard order)
int run_shards(uint32_t ndev, const std::function<int(uint32_t)> &shard) {
  if (ndev == 1) return shard(0);
  std::vector<Job> jobs(ndev);
  Latch latch;
  latch.left = ndev - 1;
  for (uint32_t r = 1; r < ndev; r++) {
    jobs[r].fn = [&shard, r] { return shard(r); };
    jobs[r].latch = &latch;
    pool()->get(r)->post(&jobs[r]);
  }
  jobs[0].rc = shard(0);
  if (jobs[0].rc != FMX_OK) jobs[0].err = fmx_last_error();
  {
    std::unique_lock<std::mutex> lk(latch.m);
    latch.cv.wait(lk, [&] { return latch.left == 0; });
  }
  for (uint32_t r = 0; r < ndev; r++)
    if (jobs[r].rc != FMX_OK) {
      fmx_set_error_text(jobs[r].err.c_str());
      return jobs[r].rc;
    }
  return FMX_OK;
}
}  // namespace

void fmx_shard_range(uint64_t nitems, uint32_t nshards, uint32_t r, uint64_t *begin, uint64_t *end) {
  uint64_t lo = 0, hi = 0;
  if (nshards && r < nshards) {
    lo = (uint64_t)(((unsigned __int128)nitems * r + nshards - 1) / nshards);
    hi = (uint64_t)(((unsigned __int128)nitems * (r + 1) + nshards - 1) / nshards);
  }
  if (begin) *begin = lo;
  if (end) *end = hi;
}

int fmx_count_batch_multi(fmx_index *const *idx, uint32_t ndev, const void *pat, const uint64_t *pat_off, uint64_t npat,
                          const uint64_t *s0e0, uint64_t *out_s, uint64_t *out_e, uint64_t *out_count) {
  if (int rc = check_replicas(idx, ndev)) return rc;
  if (npat == 0) return FMX_OK;
  if (!pat_off) return fail_arg("pat_off is NULL");
  return run_shards(ndev, [&](uint32_t r) -> int {
    uint64_t a, b;
    fmx_shard_range(npat, ndev, r, &a, &b);
    if (b == a) return FMX_OK;
    // entries a .. b of the caller's offsets with the caller's pattern buffer: fmx_count_batch moves pat[pat_off[a] ..
    // pat_off[b]) only and its kernels keep the absolute offsets
    return fmx_count_batch(idx[r], pat, pat_off + a, b - a, s0e0 ? s0e0 + 2 * a : nullptr, out_s ? out_s + a : nullptr,
                           out_e ? out_e + a : nullptr, out_count ? out_count + a : nullptr);
  });
}

int fmx_count_batch_multi_resident(fmx_index *const *idx, uint32_t ndev, const void *const *d_pat,
                                   const uint64_t *const *d_pat_off, uint64_t npat, const uint64_t *const *d_s0e0,
                                   uint64_t *out_s, uint64_t *out_e, uint64_t *out_count) {
  if (int rc = check_replicas(idx, ndev)) return rc;
  if (npat == 0) return FMX_OK;
  if (!d_pat || !d_pat_off) return fail_arg("d_pat / d_pat_off is NULL");
  return run_shards(ndev, [&](uint32_t r) -> int {
    uint64_t a, b;
    fmx_shard_range(npat, ndev, r, &a, &b);
    if (b == a) return FMX_OK;
    return fmx_count_resident_slice(idx[r], d_pat[r], d_pat_off[r], b - a, d_s0e0 ? d_s0e0[r] : nullptr,
                                    out_s ? out_s + a : nullptr, out_e ? out_e + a : nullptr,
                                    out_count ? out_count + a : nullptr);
  });
}

int fmx_locate_batch_multi(fmx_index *const *idx, uint32_t ndev, const uint64_t *s, const uint64_t *e, uint64_t npat,
                           const uint64_t *out_off, uint64_t *out_pos) {
  if (int rc = check_replicas(idx, ndev)) return rc;
  if (idx[0]->dev.sa_level == FMX_NO_LOCATE) {
    fmx_set_error(FMX_ERR_NO_LOCATE, nullptr);
    return FMX_ERR_NO_LOCATE;
  }
  if (npat == 0) return FMX_OK;
  if (!s || !e || !out_off) return fail_arg("NULL argument");
  // the whole batch's offsets must start at 0 (a gap in front of the first range is FMX_ERR_ARG on one device too:
  // there the kernel reports it; here shard 0 would take it for its slice's origin)
  if (out_off[0] != 0) return fail_arg("out_off[0] must be 0");
  return run_shards(ndev, [&](uint32_t r) -> int {
    uint64_t a, b;
    fmx_shard_range(npat, ndev, r, &a, &b);
    if (b == a) return FMX_OK;
    return fmx_locate_batch_slice(idx[r], s + a, e + a, b - a, out_off + a, out_pos);
  });
}
