// The reference's README example (README.md:35-64) written against the C++ host mirror.
// Built by tests/test_abi_cpu.py (compile + link only on CPU) and run by the -m gpu tests.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include "fm_index.hpp"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);  // the README text incl. trailing \0
  std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  try {
    fmx::Text text(bytes);
    fmx::FMIndexWithLocate index(text, 2);
    auto search = index.search(std::string("dolor"));
    std::printf("count %llu\n", (unsigned long long)search.count());
    std::printf("positions");
    for (auto &m : search.iter_matches()) std::printf(" %llu", (unsigned long long)m.locate());
    std::printf("\n");
    auto ms = search.iter_matches();
    std::printf("forward ");
    for (auto c : ms[3].chars_forward(20)) std::printf("%c", (char)c);       // README.md:78-85
    std::printf("\nbackward ");
    auto back = ms[0].chars_backward(16);                                    // README.md:67-76
    for (size_t t = back.size(); t-- > 0;) std::printf("%c", (char)back[t]);
    std::printf("\n");
    auto refined = index.search(std::string("lor")).search(std::string("do"));
    std::printf("refined %llu\n", (unsigned long long)refined.count());
    if (argc > 2) {  // examples/multi_pieces.rs on the C++ mirror
      std::ifstream f2(argv[2], std::ios::binary);
      std::vector<uint8_t> b2((std::istreambuf_iterator<char>(f2)), std::istreambuf_iterator<char>());
      fmx::FMIndexMultiPieces mp(fmx::Text(b2), 2);
      std::printf("star %llu\n", (unsigned long long)mp.search(std::string("star")).count());
      auto ids = mp.search_suffix(std::string("what you are!\n")).piece_ids();
      std::sort(ids.begin(), ids.end());
      std::printf("suffix");
      for (auto v : ids) std::printf(" %llu", (unsigned long long)v);
      auto pre = mp.search_prefix(std::string("Twinkle")).piece_ids();
      std::printf("\nprefix");
      for (auto v : pre) std::printf(" %llu", (unsigned long long)v);
      std::printf("\n");
    }
    try {
      fmx::FMIndex bad(fmx::Text(std::vector<uint8_t>{'n', 'o'}));
    } catch (const fmx::Error &e) {
      std::printf("error %s\n", e.what());
    }
    {  // one batch over three replicas of the index (fmx_replicate + fmx_count_batch_multi): the one-handle results
      auto r1 = index.replicate(0), r2 = index.replicate(0);
      std::vector<std::vector<uint8_t>> pats;
      for (const char *w : {"dolor", "ipsum", "e", "zzz", "", "in", "o"}) pats.emplace_back(w, w + std::strlen(w));
      std::vector<uint64_t> s1, e1, s3, e3;
      index.search_many(pats, s1, e1);
      index.search_many_sharded({r1.get(), r2.get()}, pats, s3, e3);
      std::printf("sharded %s\n", (s1 == s3 && e1 == e3) ? "same" : "DIFFERENT");
    }
  } catch (const fmx::Error &e) {
    std::printf("FAILED %d %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
