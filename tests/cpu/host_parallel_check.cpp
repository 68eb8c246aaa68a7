// CPU check of csrc/host_parallel.h (built and run by tests/test_host_parallel.py with g++): the threaded merge sort must
// give std::sort's result for a strict total order at every thread count, for_range must cover [0, n) exactly once, and an
// exception thrown in a part must reach the caller after all threads have been joined.
#include "../../visual-odometry_amd/csrc/host_parallel.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <stdexcept>

struct Key { uint64_t key; int idx; };

int main() {
  std::mt19937_64 rng(12345);
  int checks = 0;
  for (long n : {0L, 1L, 2L, 100L, 4095L, 4096L, 4097L, 65537L, 300000L}) {
    std::vector<Key> base((size_t)n);
    for (long i = 0; i < n; ++i) base[(size_t)i] = Key{rng() % 97, (int)i};  // many equal keys: ties broken by idx (total order)
    auto less = [](const Key& a, const Key& b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; };
    std::vector<Key> want(base);
    std::sort(want.begin(), want.end(), less);
    for (int T : {1, 2, 3, 5, 8, 13, 16}) {
      std::vector<Key> got(base);
      hostpar::sort(got, less, T);
      for (long i = 0; i < n; ++i)
        if (got[(size_t)i].key != want[(size_t)i].key || got[(size_t)i].idx != want[(size_t)i].idx) {
          std::printf("sort differs: n %ld T %d at %ld\n", n, T, i);
          return 1;
        }
      ++checks;
    }
    // already sorted and reversed inputs
    for (int T : {4, 16}) {
      std::vector<Key> got(want);
      hostpar::sort(got, less, T);
      std::vector<Key> rev(want.rbegin(), want.rend());
      hostpar::sort(rev, less, T);
      for (long i = 0; i < n; ++i)
        if (got[(size_t)i].idx != want[(size_t)i].idx || rev[(size_t)i].idx != want[(size_t)i].idx) { std::printf("sorted / reversed input: n %ld T %d\n", n, T); return 1; }
      ++checks;
    }
  }
  for (long n : {0L, 1L, 7L, 1000L, 12345L})
    for (int T : {1, 2, 7, 16}) {
      std::vector<std::atomic<int>> hit((size_t)n);
      for (auto& h : hit) h = 0;
      hostpar::for_range(n, T, [&](long i0, long i1) { for (long i = i0; i < i1; ++i) hit[(size_t)i]++; });
      for (long i = 0; i < n; ++i) if (hit[(size_t)i] != 1) { std::printf("for_range: n %ld T %d index %ld hit %d times\n", n, T, i, (int)hit[(size_t)i]); return 1; }
      ++checks;
    }
  {
    std::atomic<int> ran{0};
    bool caught = false;
    try {
      hostpar::run(8, [&](int t, int) { ++ran; if (t == 5) throw std::runtime_error("part 5"); });
    } catch (const std::runtime_error& e) { caught = std::string(e.what()) == "part 5"; }
    if (!caught || ran != 8) { std::printf("exception: caught %d, parts run %d\n", (int)caught, (int)ran); return 1; }
    ++checks;
  }
  {
    hostpar::raw_vector<double> v(1000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (double)i;
    std::vector<double> w(v.begin(), v.end());
    if (w[999] != 999.0) return 1;
    ++checks;
  }
  std::printf("host_parallel: %d checks passed\n", checks);
  return 0;
}
