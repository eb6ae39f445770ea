// introsort_replay.hpp — the tie order of libstdc++'s std::sort, without sorting everything.
//
// shi_tomasi sorts its candidates with std::sort and a predicate that looks at the score only
// (reference T:286), so the order of equal-score candidates is whatever libstdc++'s introsort leaves
// behind -- and that order decides which corners are picked and which track id they get.  The full
// sort (~150k elements per 640x480 frame) is the single most expensive host step of the pipeline, yet
// only the relative order of a handful of tied, still-eligible candidates is ever needed.
//
// This file replays libstdc++ 11's algorithm (bits/stl_algo.h: __introsort_loop ->
// __unguarded_partition_pivot -> __move_median_to_first / __unguarded_partition, then
// __final_insertion_sort) on the full candidate array, but descends only into partitions that still
// hold two or more "interesting" elements.  Partitioning is exact (same pivots, same swaps), so the
// positions of the interesting elements at the end of the loop phase are the positions the real sort
// gives them; the final insertion sort is stable for equal keys and never moves an element across a
// strictly different key, so their relative order is final.  Cost: O(n) per level along the followed
// paths instead of O(n log n).
//
// tests/test_host_math.py checks the full replay (descend everywhere + insertion sort) against the
// real std::sort on tie-heavy inputs, and the selective replay against the full one.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sfmx_host {

struct SortKey {
  double s;          // score (the only field the predicate reads)
  std::uint32_t id;  // position in the original (row-major) sequence
  std::uint32_t mark;  // 1 = interesting
};

namespace introsort_detail {

inline bool before(const SortKey& a, const SortKey& b) { return a.s > b.s; }  // T:286 predicate

inline void move_median_to_first(SortKey* result, SortKey* a, SortKey* b, SortKey* c) {
  if (before(*a, *b)) {
    if (before(*b, *c)) std::swap(*result, *b);
    else if (before(*a, *c)) std::swap(*result, *c);
    else std::swap(*result, *a);
  } else if (before(*a, *c)) std::swap(*result, *a);
  else if (before(*b, *c)) std::swap(*result, *c);
  else std::swap(*result, *b);
}
inline SortKey* unguarded_partition(SortKey* first, SortKey* last, SortKey* pivot) {
  while (true) {
    while (before(*first, *pivot)) ++first;
    --last;
    while (before(*pivot, *last)) --last;
    if (!(first < last)) return first;
    std::swap(*first, *last);
    ++first;
  }
}
inline SortKey* partition_pivot(SortKey* first, SortKey* last) {
  SortKey* mid = first + (last - first) / 2;
  move_median_to_first(first, first + 1, mid, last - 1);
  return unguarded_partition(first + 1, last, first);
}
inline int count_marked(const SortKey* first, const SortKey* last) {
  int c = 0;
  for (const SortKey* p = first; p != last; ++p) c += (int)p->mark;
  return c;
}
// returns false if the depth limit is hit (the real sort would switch to heapsort: not replayed).
// `marked` = number of interesting elements in [first,last) (only used in selective mode): after each
// partition only the smaller side is counted, the other follows by subtraction.
inline bool loop(SortKey* first, SortKey* last, long depth_limit, bool selective, int marked) {
  while (last - first > 16) {
    if (depth_limit == 0) return false;
    --depth_limit;
    SortKey* cut = partition_pivot(first, last);
    int right = 0, left = 0;
    if (selective) {
      if (last - cut < cut - first) { right = count_marked(cut, last); left = marked - right; }
      else { left = count_marked(first, cut); right = marked - left; }
    }
    if (!selective || right >= 2) {
      if (!loop(cut, last, depth_limit, selective, right)) return false;
    }
    last = cut;
    marked = left;
    if (selective && marked < 2) return true;
  }
  return true;
}
inline long lg(long n) { long k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

}  // namespace introsort_detail

// Selective replay.  On success, keys[] is in "end of loop phase" arrangement along the followed
// paths: comparing the array positions of two marked elements of equal score gives their order in
// the real std::sort output.  Returns false when the replay has to give up (depth limit).
inline bool introsort_replay_selective(SortKey* keys, std::size_t n) {
  if (n < 2) return true;
  const int marked = introsort_detail::count_marked(keys, keys + n);
  if (marked < 2) return true;
  return introsort_detail::loop(keys, keys + n, 2 * introsort_detail::lg((long)n), true, marked);
}
inline bool introsort_replay_selective(std::vector<SortKey>& keys) { return introsort_replay_selective(keys.data(), keys.size()); }

// Full replay: must reproduce std::sort exactly (used by the tests to pin the replica).
inline bool introsort_replay_full(std::vector<SortKey>& keys) {
  using namespace introsort_detail;
  if (keys.size() < 2) return true;
  SortKey* first = keys.data();
  SortKey* last = first + keys.size();
  if (!loop(first, last, 2 * lg((long)keys.size()), false, 0)) return false;
  // __final_insertion_sort: guarded insertion on the first 16, unguarded on the rest
  auto linear_insert = [](SortKey* lastp) {
    SortKey val = *lastp;
    SortKey* next = lastp - 1;
    while (before(val, *next)) { *lastp = *next; lastp = next; --next; }
    *lastp = val;
  };
  auto insertion = [&](SortKey* f, SortKey* l) {
    if (f == l) return;
    for (SortKey* i = f + 1; i != l; ++i) {
      if (before(*i, *f)) {
        SortKey val = *i;
        std::move_backward(f, i, i + 1);
        *f = val;
      } else linear_insert(i);
    }
  };
  if (last - first > 16) {
    insertion(first, first + 16);
    for (SortKey* i = first + 16; i != last; ++i) linear_insert(i);
  } else insertion(first, last);
  return true;
}

}  // namespace sfmx_host
