// Host-side helpers of ba_set_problem: the index tables (tracks, landmark order, chunk / group / write-out / gather
// tables) are sorts and scans over the observations; these run them on a few host threads.  Every helper is
// deterministic: the result does not depend on the number of threads (tests/test_gpu_setup.py compares the digests of
// the tables across thread counts and against the single-threaded set-up of round 2).
// No reference counterpart: the reference's set-up is the jac_sparsity mask (bundle_adjustment.py:90-154).
#pragma once

#include <algorithm>
#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace hostpar {

// std::vector without the zero fill of its elements: the large per-observation arrays are written in full by the threads
// that fill them, and the first touch of their pages then happens on those threads instead of on one
template <class T>
struct no_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = no_init_allocator<U>; };
  no_init_allocator() = default;
  template <class U> no_init_allocator(const no_init_allocator<U>&) {}
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
template <class T> using raw_vector = std::vector<T, no_init_allocator<T>>;

// Threads for a stage that touches `items` elements: one per `grain` elements, at most BA_HOST_THREADS (default: the
// hardware's count, capped at 16: the GPU box gives a process 16 cores of a much larger machine).
inline int threads_for(long items, long grain = 32768) {
  static const int cap = [] {
    const char* s = std::getenv("BA_HOST_THREADS_MAX");
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    int c = std::min(hw, 16);
    if (s && *s) c = std::max(1, std::atoi(s));
    return c;
  }();
  int want = cap;
  if (const char* s = std::getenv("BA_HOST_THREADS"); s && *s) want = std::max(1, std::atoi(s));  // read per call: tests change it
  return (int)std::max<long>(1, std::min<long>(want, items / std::max<long>(grain, 1)));
}

// fn(t, T) on T threads (t = 0 runs on the caller's thread); returns when all are done.  An exception in a part
// (std::bad_alloc from a table that does not fit) is carried to the caller and thrown there after every thread has
// been joined: ba_set_problem turns it into a status.
template <class F>
void run(int T, F&& fn) {
  if (T <= 1) { fn(0, 1); return; }
  std::exception_ptr first;
  std::mutex guard;
  auto part = [&](int t) {
    try {
      fn(t, T);
    } catch (...) {
      std::lock_guard<std::mutex> lk(guard);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  th.reserve((size_t)T - 1);
  for (int t = 1; t < T; ++t) {
    try {
      th.emplace_back(part, t);
    } catch (const std::system_error&) {  // the process may not start another thread: this part runs here
      part(t);
    }
  }
  part(0);
  for (auto& x : th) x.join();
  if (first) std::rethrow_exception(first);
}

// [begin, end) of part t of T over `count` items
inline long part_begin(long count, int t, int T) { return count * t / T; }

// fn(i0, i1) over [0, count) cut into T contiguous parts
template <class F>
void for_range(long count, int T, F&& fn) {
  run(T, [&](int t, int nt) { fn(part_begin(count, t, nt), part_begin(count, t + 1, nt)); });
}

// std::sort's result for a STRICT TOTAL order `less` (no two elements compare equal), on T threads: sorted parts, then
// rounds of pairwise merges, each merge itself cut in two at the median of its left run (co-ranked by binary search) so
// that the last rounds still use the threads
template <class T_, class Less>
void sort(std::vector<T_>& v, Less less, int T) {
  const long n = (long)v.size();
  if (T <= 1 || n < 4096) { std::sort(v.begin(), v.end(), less); return; }
  int parts = 1;
  while (parts < T) parts <<= 1;
  std::vector<long> cut((size_t)parts + 1);
  for (int p = 0; p <= parts; ++p) cut[(size_t)p] = n * p / parts;
  run(T, [&](int t, int nt) {
    for (int p = t; p < parts; p += nt) std::sort(v.begin() + cut[(size_t)p], v.begin() + cut[(size_t)p + 1], less);
  });
  std::vector<T_> tmp((size_t)n);
  std::vector<T_>* src = &v;
  std::vector<T_>* dst = &tmp;
  for (int width = 1; width < parts; width <<= 1) {
    // merges of this round: runs [cut[p], cut[p + width]) and [cut[p + width], cut[p + 2 width]); each split into `split` pieces
    const int n_merge = parts / (2 * width);
    const int split = std::max(1, T / n_merge);
    struct Piece { long a0, a1, b0, b1, out; };
    std::vector<Piece> pieces;
    for (int m = 0; m < n_merge; ++m) {
      const long a0 = cut[(size_t)(2 * m * width)], a1 = cut[(size_t)((2 * m + 1) * width)], b1 = cut[(size_t)((2 * m + 2) * width)];
      const long b0 = a1;
      long pa = a0, pb = b0;
      for (int s = 1; s <= split; ++s) {
        long qa, qb;
        if (s == split) { qa = a1; qb = b1; }
        else {
          qa = a0 + (a1 - a0) * s / split;
          // elements of the right run that go before (*src)[qa]: strictly smaller ones (total order: none is equal)
          qb = qa < a1 ? (long)(std::lower_bound(src->begin() + b0, src->begin() + b1, (*src)[(size_t)qa], less) - src->begin()) : b1;
          qb = std::max(qb, pb);
        }
        pieces.push_back(Piece{pa, qa, pb, qb, a0 + (pa - a0) + (pb - b0)});
        pa = qa; pb = qb;
      }
    }
    run(std::min<int>(T, (int)pieces.size()), [&](int t, int nt) {
      for (size_t i = (size_t)t; i < pieces.size(); i += (size_t)nt) {
        const Piece& pc = pieces[i];
        std::merge(src->begin() + pc.a0, src->begin() + pc.a1, src->begin() + pc.b0, src->begin() + pc.b1, dst->begin() + pc.out, less);
      }
    });
    std::swap(src, dst);
  }
  if (src != &v) v.swap(tmp);
}

}  // namespace hostpar
