// fm_index.hpp -- header-only C++ mirror of the reference's public surface for the
// count / locate path, over the C ABI in include/fmx.h.
//
//   reference (Rust)                                   here (C++)
//   Text::new / Text::with_max_character  text.rs:28-49      fmx::Text
//   FMIndex::new(&text)                   frontend.rs:195    fmx::FMIndex(text)
//   FMIndexWithLocate::new(&text, level)  frontend.rs:205    fmx::FMIndexWithLocate(text, level)
//   RLFMIndex / RLFMIndexWithLocate       frontend.rs:223    fmx::RLFMIndex / fmx::RLFMIndexWithLocate
//   SearchIndex::search / len / heap_size frontend.rs:26-44  .search(p) / .len() / .heap_size()
//   Search::search / count / iter_matches frontend.rs:70-84  Search::search / count / iter_matches
//   MatchWithLocate::locate               frontend.rs:96-98  Match::locate
//   Error::InvalidText(msg)               error.rs:3-15      fmx::Error (what() == Display)
//
// Single queries go through the batched entry points with a batch of one; use
// search_many / locate_many for throughput.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <cstring>
#include <vector>
#include "fmx.h"

namespace fmx {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc != FMX_OK) throw Error(rc, fmx_last_error());
}

class Text {  // text.rs:11-64
 public:
  explicit Text(std::vector<uint8_t> t, uint64_t max_character = 255)
      : t_(std::move(t)), max_(max_character) {}
  static Text with_max_character(std::vector<uint8_t> t, uint64_t m) { return Text(std::move(t), m); }
  const std::vector<uint8_t> &text() const { return t_; }
  uint64_t max_character() const { return max_; }

 private:
  std::vector<uint8_t> t_;
  uint64_t max_;
};

class Index;

class Match {  // frontend.rs:86-98
 public:
  Match(const Index *ix, uint64_t i) : ix_(ix), i_(i) {}
  uint64_t locate() const;  // wrapper.rs:238-242
  uint64_t piece_id() const;  // MatchWithPieceId (frontend.rs:100-104)
  uint64_t row() const { return i_; }
  // iter_chars_forward().take(k) / iter_chars_backward().take(k)  (wrapper.rs:142-183)
  std::vector<uint64_t> chars_forward(size_t k) const;
  std::vector<uint64_t> chars_backward(size_t k) const;

 private:
  const Index *ix_;
  uint64_t i_;
};

class Search {  // frontend.rs:70-84
 public:
  Search(const Index *ix, uint64_t s, uint64_t e, bool fresh, bool prefix_only = false)
      : ix_(ix), s_(s), e_(e), fresh_(fresh), prefix_(prefix_only) {}
  Search search(const std::vector<uint8_t> &pattern) const;   // wrapper.rs:103-124 (prepends)
  Search search(const std::string &p) const { return search(std::vector<uint8_t>(p.begin(), p.end())); }
  uint64_t count() const { return e_ - s_; }                  // wrapper.rs:132-134
  std::pair<uint64_t, uint64_t> get_range() const { return {s_, e_}; }
  std::vector<Match> iter_matches() const;                    // wrapper.rs:203-217
  std::vector<uint64_t> locate_all() const;                   // iter_matches().map(locate)
  std::vector<uint64_t> piece_ids() const;                    // iter_matches().map(piece_id)

 private:
  std::vector<uint64_t> rows() const;
  const Index *ix_;
  uint64_t s_, e_;
  bool fresh_, prefix_;
};

class Index {
 public:
  // `flags`: FMX_FLAG_PLAIN vetoes the count accelerators a DNA-like FM index of 2^24+ symbols gets by default (round 6);
  // FMX_FLAG_PAIR_INDEX | FMX_FLAG_KMER_TABLE ask for them by name
  Index(const Text &text, uint32_t kind, uint32_t level, int device = 0, uint32_t flags = 0) {
    check(fmx_build(text.text().data(), text.text().size(), 1, text.max_character(), kind, level, flags,
                    device, &h_));
  }
  Index(const Index &) = delete;
  Index &operator=(const Index &) = delete;
  ~Index() { fmx_free(h_); }
  Search search(const std::vector<uint8_t> &p) const { return Search(this, 0, 0, true).search(p); }
  Search search(const std::string &p) const { return search(std::vector<uint8_t>(p.begin(), p.end())); }
  uint64_t len() const { return fmx_len(h_); }
  uint64_t heap_size() const { return fmx_index_bytes(h_); }
  const fmx_index *handle() const { return h_; }
  // batched: patterns[k] -> (s, e)
  void search_many(const std::vector<std::vector<uint8_t>> &patterns, std::vector<uint64_t> &s,
                   std::vector<uint64_t> &e) const {
    std::vector<uint8_t> flat;
    std::vector<uint64_t> off(1, 0);
    for (auto &p : patterns) {
      flat.insert(flat.end(), p.begin(), p.end());
      off.push_back(flat.size());
    }
    if (flat.empty()) flat.push_back(0);
    s.assign(patterns.size(), 0);
    e.assign(patterns.size(), 0);
    check(fmx_count_batch(h_, flat.data(), off.data(), patterns.size(), nullptr, s.data(), e.data(),
                          nullptr));
  }
  // fmx_replicate: a second index with its own copy of the HBM arrays on `device` (no rebuild)
  std::unique_ptr<Index> replicate(int device) const {
    fmx_index *h = nullptr;
    check(fmx_replicate(h_, device, &h));
    return std::unique_ptr<Index>(new Index(h));
  }
  // search_many over `replicas` (this index = replica 0): contiguous shards, results in place (fmx_count_batch_multi)
  void search_many_sharded(const std::vector<const Index *> &replicas, const std::vector<std::vector<uint8_t>> &patterns,
                           std::vector<uint64_t> &s, std::vector<uint64_t> &e) const {
    std::vector<uint8_t> flat;
    std::vector<uint64_t> off(1, 0);
    for (auto &p : patterns) {
      flat.insert(flat.end(), p.begin(), p.end());
      off.push_back(flat.size());
    }
    if (flat.empty()) flat.push_back(0);
    s.assign(patterns.size(), 0);
    e.assign(patterns.size(), 0);
    std::vector<fmx_index *> hs(1, h_);
    for (const Index *r : replicas) hs.push_back(r->h_);
    check(fmx_count_batch_multi(hs.data(), (uint32_t)hs.size(), flat.data(), off.data(), patterns.size(), nullptr,
                                s.data(), e.data(), nullptr));
  }

 private:
  explicit Index(fmx_index *h) : h_(h) {}
  fmx_index *h_ = nullptr;
};

struct FMIndex : Index {
  explicit FMIndex(const Text &t, int device = 0, uint32_t flags = 0)
      : Index(t, FMX_KIND_FM, FMX_NO_LOCATE, device, flags) {}
};
struct FMIndexWithLocate : Index {
  FMIndexWithLocate(const Text &t, uint32_t level, int device = 0, uint32_t flags = 0)
      : Index(t, FMX_KIND_FM, level, device, flags) {}
};
struct RLFMIndex : Index {
  explicit RLFMIndex(const Text &t, int device = 0) : Index(t, FMX_KIND_RLFM, FMX_NO_LOCATE, device) {}
};
struct RLFMIndexWithLocate : Index {
  RLFMIndexWithLocate(const Text &t, uint32_t level, int device = 0) : Index(t, FMX_KIND_RLFM, level, device) {}
};
// FMIndexMultiPieces / ...WithLocate (frontend.rs:245-267): SearchIndexWithMultiPieces
struct FMIndexMultiPieces : Index {
  explicit FMIndexMultiPieces(const Text &t, uint32_t level = FMX_NO_LOCATE, int device = 0)
      : Index(t, FMX_KIND_MULTI, level, device) {}
  uint64_t pieces_count() const { return fmx_pieces_count(handle()); }
  Search search_prefix(const std::string &p) const {          // wrapper.rs:57-63
    return Search(this, 0, 0, true, true).search(p);
  }
  Search search_suffix(const std::string &p) const {          // wrapper.rs:66-72
    return Search(this, 0, pieces_count(), false, false).search(p);
  }
  Search search_exact(const std::string &p) const {           // wrapper.rs:75-81
    return Search(this, 0, pieces_count(), false, true).search(p);
  }
};

inline Search Search::search(const std::vector<uint8_t> &pattern) const {
  uint64_t off[2] = {0, pattern.size()};
  uint64_t se[2] = {s_, e_};
  uint64_t s = 0, e = 0;
  uint8_t dummy = 0;
  check(fmx_count_batch(ix_->handle(), pattern.empty() ? &dummy : pattern.data(), off, 1,
                        fresh_ ? nullptr : se, &s, &e, nullptr));
  return Search(ix_, s, e, false, prefix_);
}
inline std::vector<uint64_t> Search::rows() const {
  uint64_t cnt = 0, off[2] = {0, 0};
  check(fmx_match_counts(ix_->handle(), &s_, &e_, 1, prefix_ ? 1 : 0, &cnt));
  off[1] = cnt;
  std::vector<uint64_t> r(cnt);
  if (cnt) check(fmx_match_rows(ix_->handle(), &s_, &e_, 1, prefix_ ? 1 : 0, off, r.data()));
  return r;
}
inline std::vector<Match> Search::iter_matches() const {
  std::vector<Match> m;
  for (uint64_t i : rows()) m.emplace_back(ix_, i);
  return m;
}
inline std::vector<uint64_t> Search::piece_ids() const {
  std::vector<uint64_t> r = rows(), out(r.size());
  if (!r.empty()) check(fmx_piece_id_batch(ix_->handle(), r.data(), r.size(), out.data()));
  return out;
}
inline uint64_t Match::piece_id() const {
  uint64_t v = fmx_piece_id(ix_->handle(), i_);
  if (v == ~0ull) throw Error(FMX_ERR_ARG, fmx_last_error());
  return v;
}
inline std::vector<uint64_t> Search::locate_all() const {
  if (prefix_) {
    std::vector<uint64_t> r = rows(), pos(r.size());
    if (!r.empty()) check(fmx_get_sa_batch(ix_->handle(), r.data(), r.size(), pos.data()));
    return pos;
  }
  uint64_t off[2] = {0, e_ - s_};
  std::vector<uint64_t> pos(e_ - s_);
  if (e_ > s_) check(fmx_locate_batch(ix_->handle(), &s_, &e_, 1, off, pos.data()));
  return pos;
}
// one fmx_extract_batch launch: k steps of the iterator, symbols widened to u64
inline std::vector<uint64_t> extract_one(const fmx_index *h, uint64_t row, size_t k, int forward) {
  const uint32_t sb = fmx_sym_bytes(h);
  std::vector<uint8_t> raw(k * sb + 8);
  uint64_t len = 0;
  check(fmx_extract_batch(h, &row, 1, k, forward, raw.data(), &len, nullptr));
  std::vector<uint64_t> out((size_t)len);
  for (size_t t = 0; t < (size_t)len; t++) {
    uint64_t v = 0;
    std::memcpy(&v, raw.data() + t * sb, sb);   // little-endian host
    out[t] = v;
  }
  return out;
}
inline std::vector<uint64_t> Match::chars_forward(size_t k) const {    // wrapper.rs:175-183
  return extract_one(ix_->handle(), i_, k, 1);
}
inline std::vector<uint64_t> Match::chars_backward(size_t k) const {   // wrapper.rs:154-161
  return extract_one(ix_->handle(), i_, k, 0);
}
inline uint64_t Match::locate() const {
  uint64_t v = fmx_get_sa(ix_->handle(), i_);
  if (v == ~0ull) throw Error(FMX_ERR_NO_LOCATE, fmx_last_error());
  return v;
}

}  // namespace fmx
