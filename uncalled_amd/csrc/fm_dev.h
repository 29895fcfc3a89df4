// Device FM-index arithmetic over the BWA on-disk block layout:
// one 64-byte block per 128 BWT symbols = 4 x u64 counts-before-block (A,C,G,T) + 8 x u32 of
// 2-bit symbols, symbol j of a word at bits (~j & 15) << 1.  Replaces the libbwa calls behind
// BwaIndex::get_neighbor (bwa_index.hpp:158-162 -> bwt_2occ) and BwaIndex::sa (:176-178 -> bwt_sa).
#pragma once
#include <hip/hip_runtime.h>

#include "unc_dev_types.h"

namespace unc {

// occurrences of base c among the 32 symbols of y
__device__ __forceinline__ uint32_t occ32(uint64_t y, uint32_t c) {
    uint64_t hi = (c & 2) ? y : ~y, lo = (c & 1) ? y : ~y;
    return (uint32_t)__popcll((hi >> 1) & lo & 0x5555555555555555ull);
}

struct FmBlock { uint4 q0, q1, q2, q3; };  // counts A,C | counts G,T | symbols 0..63 | symbols 64..127

__device__ __forceinline__ FmBlock fm_load_block(const DevIndex &ix, uint64_t kk) {
    const uint4 *p = reinterpret_cast<const uint4 *>(ix.bwt + ((kk >> 7) << 4));
    FmBlock b;
    b.q0 = p[0]; b.q1 = p[1]; b.q2 = p[2]; b.q3 = p[3];
    return b;
}

__device__ __forceinline__ uint64_t fm_block_count(const FmBlock &b, uint32_t c) {
    uint64_t cA = ((uint64_t)b.q0.y << 32) | b.q0.x, cC = ((uint64_t)b.q0.w << 32) | b.q0.z;
    uint64_t cG = ((uint64_t)b.q1.y << 32) | b.q1.x, cT = ((uint64_t)b.q1.w << 32) | b.q1.z;
    return c == 0 ? cA : c == 1 ? cC : c == 2 ? cG : cT;
}

// number of c among block symbols [0 .. (kk & 127)] (inclusive), kk already sentinel-adjusted
__device__ __forceinline__ uint32_t fm_block_rank(const FmBlock &b, uint64_t kk, uint32_t c) {
    // 32-symbol words, big-endian pairs of u32 (p[0] << 32 | p[1])
    uint64_t w0 = ((uint64_t)b.q2.x << 32) | b.q2.y, w1 = ((uint64_t)b.q2.z << 32) | b.q2.w;
    uint64_t w2 = ((uint64_t)b.q3.x << 32) | b.q3.y, w3 = ((uint64_t)b.q3.z << 32) | b.q3.w;
    uint32_t full = (uint32_t)(kk & 127) >> 5;                     // whole words before the last one
    uint64_t tailmask = ~((1ull << ((~kk & 31) << 1)) - 1ull);     // keep symbols 0..(kk&31) of the last word
    uint32_t n = 0;
    n += full > 0 ? occ32(w0, c) : (full == 0 ? occ32(w0 & tailmask, c) : 0u);
    n += full > 1 ? occ32(w1, c) : (full == 1 ? occ32(w1 & tailmask, c) : 0u);
    n += full > 2 ? occ32(w2, c) : (full == 2 ? occ32(w2 & tailmask, c) : 0u);
    n += full == 3 ? occ32(w3 & tailmask, c) : 0u;
    if (c == 0) n -= (uint32_t)(~kk & 31);                         // masked-off tail symbols look like A
    return n;
}

// bwt_occ: occurrences of c in BWT rows [0..k] of the matrix that includes the sentinel row
__device__ __forceinline__ uint64_t fm_occ(const DevIndex &ix, uint64_t k, uint32_t c) {
    if (k == ix.seq_len) return ix.L2[c + 1] - ix.L2[c];
    if (k == ~0ull) return 0;
    uint64_t kk = k - (k >= ix.primary ? 1 : 0);
    FmBlock b = fm_load_block(ix, kk);
    return fm_block_count(b, c) + fm_block_rank(b, kk, c);
}

// The part of a block one rank query reads: the count of base c before the block and the symbol words up to the
// query position.  Every per-lane load instruction costs the CU's vector memory pipeline one cache-line request per
// active lane whatever its width, so only what is needed is fetched: 8 B + 16 B, + 16 B when the position lies in the
// upper half of the block.
struct FmPart { uint64_t cnt; uint4 lo, hi; };

__device__ __forceinline__ FmPart fm_load_part(const DevIndex &ix, uint64_t kk, uint32_t c) {
    const uint32_t *p = ix.bwt + ((kk >> 7) << 4);
    FmPart b;
    b.cnt = *reinterpret_cast<const uint64_t *>(p + 2 * c);
    b.lo = reinterpret_cast<const uint4 *>(p)[2];
    b.hi = make_uint4(0u, 0u, 0u, 0u);
    if (kk & 64) b.hi = reinterpret_cast<const uint4 *>(p)[3];
    return b;
}

// rank inside the loaded words: the base folded into an XOR pattern (both bits of a 2-bit field set <=> the symbol equals c, no
// per-word selects on c) and the prefix mask of each word derived arithmetically from the position -- about half the
// instructions of a select-per-word form (round 2, 50 k E. coli reads, same box: k_map 6368 -> 6122 ms)
__device__ __forceinline__ uint32_t fm_word_rank(uint32_t w_hi, uint32_t w_lo, uint64_t npat, int32_t keep) {
    // keep: how many of this word's 32 symbols count (<= 0: none, >= 32: all); symbol j sits at bits 62 - 2j .. 63 - 2j
    const uint64_t w = ((uint64_t)w_hi << 32) | w_lo;
    const uint64_t x = w ^ npat;
    uint64_t m = x & (x >> 1) & 0x5555555555555555ull;
    const int32_t k = keep < 0 ? 0 : (keep > 32 ? 32 : keep);
    const uint64_t pm = k == 0 ? 0ull : (~0ull << (64 - 2 * k));
    m &= pm;
    return (uint32_t)__popcll(m);
}
__device__ __forceinline__ uint64_t fm_part_rank(const FmPart &b, uint64_t kk, uint32_t c) {
    const uint64_t pat = (uint64_t)c * 0x5555555555555555ull;     // c in every 2-bit field
    const uint64_t npat = ~pat;
    const int32_t r1 = (int32_t)(kk & 127) + 1;                   // symbols 0 .. r count
    const uint32_t n = fm_word_rank(b.lo.x, b.lo.y, npat, r1) + fm_word_rank(b.lo.z, b.lo.w, npat, r1 - 32) +
                       fm_word_rank(b.hi.x, b.hi.y, npat, r1 - 64) + fm_word_rank(b.hi.z, b.hi.w, npat, r1 - 96);
    return b.cnt + n;
}

// BwaIndex::get_neighbor: one backward-search step of the range [s,e] with base c (bwt_2occ), in two halves so that a caller
// can have several look-ups in flight: fm_nbr_issue requests the block words (nothing waits on them), fm_nbr_finish
// computes the two ranks.  Like bwt_2occ, the two rank queries share the block when s - 1 and e fall into the same one
// (nearly always for the short ranges that make up most of the path forest); when they do not (young paths, sources), both
// blocks are requested before either rank is computed -- one memory round trip, not two in a row.
struct FmNbr {
    FmPart bl, bk;
    uint64_t kk, ll, ok, ol;      // sentinel-adjusted positions; ok / ol preset for the positions that need no block
    uint32_t c;
    bool k_plain, l_plain, shared;
};
__device__ __forceinline__ FmNbr fm_nbr_issue(const DevIndex &ix, uint64_t s, uint64_t e, uint32_t c) {
    FmNbr q;
    const uint64_t k = s - 1, l = e;
    q.c = c;
    q.k_plain = k != ix.seq_len && k != ~0ull; q.l_plain = l != ix.seq_len && l != ~0ull;
    q.kk = k - (k >= ix.primary ? 1 : 0); q.ll = l - (l >= ix.primary ? 1 : 0);
    q.ok = k == ix.seq_len ? ix.L2[c + 1] - ix.L2[c] : 0; q.ol = l == ix.seq_len ? ix.L2[c + 1] - ix.L2[c] : 0;
    q.shared = q.k_plain && q.l_plain && q.kk <= q.ll && (q.kk >> 7) == (q.ll >> 7);   // the words loaded for l serve k
    q.bl.cnt = 0; q.bl.lo = q.bl.hi = make_uint4(0u, 0u, 0u, 0u);
    q.bk = q.bl;
    if (q.l_plain) q.bl = fm_load_part(ix, q.ll, c);
    if (q.k_plain && !q.shared) q.bk = fm_load_part(ix, q.kk, c);
    return q;
}
__device__ __forceinline__ void fm_nbr_finish(const DevIndex &ix, const FmNbr &q, uint64_t *os, uint64_t *oe) {
    uint64_t ok = q.ok, ol = q.ol;
    if (q.l_plain) ol = fm_part_rank(q.bl, q.ll, q.c);
    if (q.k_plain) {
        FmPart pk;               // the block of k: the one loaded for l when they share it
        pk.cnt = q.shared ? q.bl.cnt : q.bk.cnt;
        pk.lo = q.shared ? q.bl.lo : q.bk.lo;
        pk.hi = q.shared ? q.bl.hi : q.bk.hi;
        ok = fm_part_rank(pk, q.kk, q.c);
    }
    *os = ix.L2[q.c] + ok + 1;
    *oe = ix.L2[q.c] + ol;
}
__device__ __forceinline__ void fm_get_neighbor(const DevIndex &ix, uint64_t s, uint64_t e, uint32_t c,
                                                uint64_t *os, uint64_t *oe) {
    const FmNbr q = fm_nbr_issue(ix, s, e, c);
    fm_nbr_finish(ix, q, os, oe);
}

// bwt_sa: walk LF until a sampled row (multiple of 32); *steps gets the number of LF steps
__device__ __forceinline__ uint64_t fm_sa(const DevIndex &ix, uint64_t k, uint32_t *steps) {
    uint32_t n = 0;
    while (k & 31) {
        ++n;
        if (k == ix.primary) { k = 0; continue; }
        uint64_t kk = k - (k > ix.primary ? 1 : 0);
        FmBlock b = fm_load_block(ix, kk);
        uint32_t j = (uint32_t)(kk & 127);
        uint32_t word = j < 64 ? (j < 32 ? (j < 16 ? b.q2.x : b.q2.y) : (j < 48 ? b.q2.z : b.q2.w))
                               : (j < 96 ? (j < 80 ? b.q3.x : b.q3.y) : (j < 112 ? b.q3.z : b.q3.w));
        uint32_t c = (word >> ((~j & 15) << 1)) & 3;
        k = ix.L2[c] + fm_block_count(b, c) + fm_block_rank(b, kk, c);
    }
    *steps = n;
    return (uint64_t)n + ix.sa[k >> 5];
}

// SA look-up through the dense table built at index load (k_dense_sa): entry = SA value (40 bits) | LF steps the
// BWA-format walk above would have taken << 40 (kept so that the SURVEY 8(d) work counters stay those of
// the reference's algorithm).  One 8-byte read instead of ~31 dependent 64-byte reads.
__device__ __forceinline__ uint64_t fm_sa_dense(const DevIndex &ix, uint64_t k, uint32_t *steps) {
    const uint64_t v = ix.sa_dense[k];
    *steps = (uint32_t)(v >> 40);
    return v & ((1ull << 40) - 1ull);
}

}  // namespace unc
