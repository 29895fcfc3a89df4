// TEST INFRASTRUCTURE (oracle/_ref build only) -- stand-in for orlp/pdqsort, an un-vendored submodule of the reference
// (submods/pdqsort is an empty directory in /root/reference; call site mapper.cpp:531).
//
// pdqsort is an UNSTABLE comparison sort: it fixes the order of elements only up to operator<.  Children that tie on
// (fm_range_, seed_prob_) (mapper.cpp:866-871) may come out in any order, and the walk that follows keeps the LAST of a run
// of equal ranges (mapper.cpp:569-572), so the tie order can decide which of two lineages lives on.  Three orders are
// selectable at run time (ref_set_sort_mode in the harness), so that the exposure can be COUNTED instead of argued:
//   0  stable sort: ties in creation order -- the convention of this project (oracle, C restatement and HIP kernels alike)
//   1  pattern-defeating quicksort restated from its published algorithm (Orson Peters, "Pattern-defeating Quicksort",
//      arXiv:2106.05123, and the description in the project's README: insertion sort below 24 elements, median of 3 / Tukey
//      ninther pivot above 128, partition-right with the pivot's equals to the right, partition-left when the pivot equals its
//      predecessor, a fixed shuffle of eight elements after a bad partition, a bounded attempt at finishing an already
//      partitioned range by insertion, heapsort after log2(n) bad partitions).  Not a copy of upstream's header, and NOT
//      pinned against upstream's object code (none is available here): it shows what AN order of that family does to the PAF
//   2  stable sort with ties in REVERSED creation order: the opposite extreme of mode 0
// The wrapper also counts, after every sort, the adjacent pairs that compare equal and the sorts (= events) that had one.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iterator>
#include <utility>

namespace unc_shim {
inline int &sort_mode() { static int m = 0; return m; }
inline std::atomic<uint64_t> &tie_pairs() { static std::atomic<uint64_t> v(0); return v; }
inline std::atomic<uint64_t> &tie_events() { static std::atomic<uint64_t> v(0); return v; }
inline std::atomic<uint64_t> &sorts() { static std::atomic<uint64_t> v(0); return v; }

namespace pdq {
enum { INSERTION_BELOW = 24, NINTHER_ABOVE = 128, PARTIAL_LIMIT = 8 };

template <class It, class Cmp> inline void order2(It a, It b, Cmp c) { if (c(*b, *a)) std::iter_swap(a, b); }
template <class It, class Cmp> inline void order3(It a, It b, It m, Cmp c) { order2(a, b, c); order2(b, m, c); order2(a, b, c); }

// insertion sort; `guarded` = the range is the leftmost of the whole sort (else an element not larger than all of it sits before it)
template <class It, class Cmp> inline void insertion(It first, It last, Cmp c, bool guarded) {
    typedef typename std::iterator_traits<It>::value_type T;
    if (first == last) return;
    for (It cur = first + 1; cur != last; ++cur) {
        It hole = cur, prev = cur - 1;
        if (c(*hole, *prev)) {
            T tmp = std::move(*hole);
            do { *hole-- = std::move(*prev); } while ((!guarded || hole != first) && c(tmp, *--prev));
            *hole = std::move(tmp);
        }
    }
}
// the same, giving up (false) once more than PARTIAL_LIMIT element moves have been made
template <class It, class Cmp> inline bool insertion_bounded(It first, It last, Cmp c) {
    typedef typename std::iterator_traits<It>::value_type T;
    if (first == last) return true;
    std::size_t moved = 0;
    for (It cur = first + 1; cur != last; ++cur) {
        It hole = cur, prev = cur - 1;
        if (c(*hole, *prev)) {
            T tmp = std::move(*hole);
            do { *hole-- = std::move(*prev); } while (hole != first && c(tmp, *--prev));
            *hole = std::move(tmp);
            moved += (std::size_t)(cur - hole);
        }
        if (moved > PARTIAL_LIMIT) return false;
    }
    return true;
}
// pivot = *first; elements smaller than it to the left, the rest (its equals included) to the right; returns the pivot's place
// and whether no element had to be swapped
template <class It, class Cmp> inline std::pair<It, bool> partition_right(It first, It last, Cmp c) {
    typedef typename std::iterator_traits<It>::value_type T;
    T pivot(std::move(*first));
    It lo = first, hi = last;
    while (c(*++lo, pivot)) {}
    if (lo - 1 == first) { while (lo < hi && !c(*--hi, pivot)) {} }
    else { while (!c(*--hi, pivot)) {} }
    const bool untouched = lo >= hi;
    while (lo < hi) {
        std::iter_swap(lo, hi);
        while (c(*++lo, pivot)) {}
        while (!c(*--hi, pivot)) {}
    }
    It at = lo - 1;
    *first = std::move(*at);
    *at = std::move(pivot);
    return std::make_pair(at, untouched);
}
// pivot = *first; its equals to the LEFT (used when the pivot equals the element before the range: the left part is then done)
template <class It, class Cmp> inline It partition_left(It first, It last, Cmp c) {
    typedef typename std::iterator_traits<It>::value_type T;
    T pivot(std::move(*first));
    It lo = first, hi = last;
    while (c(pivot, *--hi)) {}
    if (hi + 1 == last) { while (lo < hi && !c(pivot, *++lo)) {} }
    else { while (!c(pivot, *++lo)) {} }
    while (lo < hi) {
        std::iter_swap(lo, hi);
        while (c(pivot, *--hi)) {}
        while (!c(pivot, *++lo)) {}
    }
    *first = std::move(*hi);
    *hi = std::move(pivot);
    return hi;
}
template <class It, class Cmp> inline void loop(It first, It last, Cmp c, int bad_left, bool leftmost) {
    typedef typename std::iterator_traits<It>::difference_type D;
    for (;;) {
        const D n = last - first;
        if (n < INSERTION_BELOW) { insertion(first, last, c, leftmost); return; }
        const D h = n / 2;
        if (n > NINTHER_ABOVE) {
            order3(first, first + h, last - 1, c);
            order3(first + 1, first + (h - 1), last - 2, c);
            order3(first + 2, first + (h + 1), last - 3, c);
            order3(first + (h - 1), first + h, first + (h + 1), c);
            std::iter_swap(first, first + h);
        } else order3(first + h, first, last - 1, c);
        if (!leftmost && !c(*(first - 1), *first)) { first = partition_left(first, last, c) + 1; continue; }
        const std::pair<It, bool> pr = partition_right(first, last, c);
        const It at = pr.first;
        const D nl = at - first, nr = last - (at + 1);
        if (nl < n / 8 || nr < n / 8) {            // a bad partition: shuffle a few elements, count it
            if (--bad_left == 0) { std::make_heap(first, last, c); std::sort_heap(first, last, c); return; }
            if (nl >= INSERTION_BELOW) {
                std::iter_swap(first, first + nl / 4);
                std::iter_swap(at - 1, at - nl / 4);
                if (nl > NINTHER_ABOVE) {
                    std::iter_swap(first + 1, first + (nl / 4 + 1));
                    std::iter_swap(first + 2, first + (nl / 4 + 2));
                    std::iter_swap(at - 2, at - (nl / 4 + 1));
                    std::iter_swap(at - 3, at - (nl / 4 + 2));
                }
            }
            if (nr >= INSERTION_BELOW) {
                std::iter_swap(at + 1, at + (1 + nr / 4));
                std::iter_swap(last - 1, last - nr / 4);
                if (nr > NINTHER_ABOVE) {
                    std::iter_swap(at + 2, at + (2 + nr / 4));
                    std::iter_swap(at + 3, at + (3 + nr / 4));
                    std::iter_swap(last - 2, last - (1 + nr / 4));
                    std::iter_swap(last - 3, last - (2 + nr / 4));
                }
            }
        } else if (pr.second && insertion_bounded(first, at, c) && insertion_bounded(at + 1, last, c)) return;
        loop(first, at, c, bad_left, leftmost);
        first = at + 1;
        leftmost = false;
    }
}
template <class It, class Cmp> inline void sort(It first, It last, Cmp c) {
    if (first == last) return;
    int lg = 0;
    for (std::size_t n = (std::size_t)(last - first); n >>= 1;) ++lg;
    loop(first, last, c, lg, true);
}
}  // namespace pdq

template <class It, class Cmp> inline void sort_any(It first, It last, Cmp c) {
    const int mode = sort_mode();
    if (mode == 1) pdq::sort(first, last, c);
    else if (mode == 2) { std::reverse(first, last); std::stable_sort(first, last, c); }
    else std::stable_sort(first, last, c);
    uint64_t pairs = 0;
    if (first != last)
        for (It a = first, b = first + 1; b != last; ++a, ++b) pairs += !c(*a, *b) && !c(*b, *a);
    sorts().fetch_add(1, std::memory_order_relaxed);
    if (pairs) { tie_pairs().fetch_add(pairs, std::memory_order_relaxed); tie_events().fetch_add(1, std::memory_order_relaxed); }
}
}  // namespace unc_shim

template <class It> inline void pdqsort(It first, It last) {
    unc_shim::sort_any(first, last, std::less<typename std::iterator_traits<It>::value_type>());
}
template <class It, class Cmp> inline void pdqsort(It first, It last, Cmp c) { unc_shim::sort_any(first, last, c); }
