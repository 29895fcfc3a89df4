// TEST INFRASTRUCTURE (oracle/_ref build only) -- stand-in for orlp/pdqsort, an
// un-vendored, un-pinned submodule of the reference (call site mapper.cpp:531).
// pdqsort is an unstable comparison sort: it fixes the order of elements only up to
// operator<.  Children that tie on (fm_range_, seed_prob_) (mapper.cpp:866-871) may
// therefore come out in any order; the tie-break chosen for this project -- in the
// oracle, the C restatement and the HIP kernels alike -- is ORIGINAL CHILD INDEX,
// i.e. a stable sort.
#pragma once
#include <algorithm>
template <class It> inline void pdqsort(It first, It last) { std::stable_sort(first, last); }
template <class It, class Cmp> inline void pdqsort(It first, It last, Cmp c) { std::stable_sort(first, last, c); }
