// Device FM-index arithmetic over the BWA on-disk block layout:
// one 64-byte block per 128 BWT symbols = 4 x u64 counts-before-block (A,C,G,T) + 8 x u32 of
// 2-bit symbols, symbol j of a word at bits (~j & 15) << 1.  Replaces the libbwa calls behind
// BwaIndex::get_neighbor (bwa_index.hpp:158-162 -> bwt_2occ) and BwaIndex::sa (:176-178 -> bwt_sa).
#pragma once
#include <hip/hip_runtime.h>

#include "unc_dev_types.h"
#include "wave_prims.h"

namespace unc {

// occurrences of base c among the 32 symbols of y
__device__ __forceinline__ uint32_t occ32(uint64_t y, uint32_t c) {
    uint64_t hi = (c & 2) ? y : ~y, lo = (c & 1) ? y : ~y;
    return (uint32_t)__popcll((hi >> 1) & lo & 0x5555555555555555ull);
}


// 16 / 8 bytes at a pointer of any address space (one dwordx4 / dwordx2 load)
template <class P> __device__ __forceinline__ uint4 ld16(P p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
template <class P> __device__ __forceinline__ uint64_t ld8(P p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// L2[c] for a per-lane c (0..4): a vector load from the kernel's argument block (the table is five numbers).  Round 5 tried the value
// put together from guarded differences of five scalars instead -- no load, so that the wide FM step does not wait for the table
// before it requests its blocks -- and measured it 2.6 % SLOWER on GRCh38 (profiles/r05_ab_grch38_wide_variants.log): the table is
// L1-resident, the sixteen extra 64-bit selects per call are not free in a kernel whose vector ALUs are half busy.
template <class IX> __device__ __forceinline__ uint64_t fm_L2(const IX &ix, uint32_t c) { return ix.L2[c]; }

struct FmBlock { uint4 q0, q1, q2, q3; };  // counts A,C | counts G,T | symbols 0..63 | symbols 64..127

template <class IX> __device__ __forceinline__ FmBlock fm_load_block(const IX &ix, uint64_t kk) {
    const auto p = ix.bwt + ((kk >> 7) << 4);
    FmBlock b;
    b.q0 = ld16(p); b.q1 = ld16(p + 4); b.q2 = ld16(p + 8); b.q3 = ld16(p + 12);
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
template <class IX> __device__ __forceinline__ uint64_t fm_occ(const IX &ix, uint64_t k, uint32_t c) {
    if (k == ix.seq_len) return fm_L2(ix, c + 1u) - fm_L2(ix, c);
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

template <class IX> __device__ __forceinline__ FmPart fm_load_part(const IX &ix, uint64_t kk, uint32_t c) {
    const auto p = ix.bwt + ((kk >> 7) << 4);
    FmPart b;
    b.cnt = ld8(p + 2 * c);
    b.lo = ld16(p + 8);
    b.hi = make_uint4(0u, 0u, 0u, 0u);
    if (kk & 64) b.hi = ld16(p + 12);
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
template <class IX> __device__ __forceinline__ FmNbr fm_nbr_issue(const IX &ix, uint64_t s, uint64_t e, uint32_t c) {
    FmNbr q;
    const uint64_t k = s - 1, l = e;
    q.c = c;
    q.k_plain = k != ix.seq_len && k != ~0ull; q.l_plain = l != ix.seq_len && l != ~0ull;
    q.kk = k - (k >= ix.primary ? 1 : 0); q.ll = l - (l >= ix.primary ? 1 : 0);
    const uint64_t nc = fm_L2(ix, c + 1u) - fm_L2(ix, c);      // occurrences of c in the whole text
    q.ok = k == ix.seq_len ? nc : 0; q.ol = l == ix.seq_len ? nc : 0;
    q.shared = q.k_plain && q.l_plain && q.kk <= q.ll && (q.kk >> 7) == (q.ll >> 7);   // the words loaded for l serve k
    q.bl.cnt = 0; q.bl.lo = q.bl.hi = make_uint4(0u, 0u, 0u, 0u);
    q.bk = q.bl;
    if (q.l_plain) q.bl = fm_load_part(ix, q.ll, c);
    if (q.k_plain && !q.shared) q.bk = fm_load_part(ix, q.kk, c);
    return q;
}
template <class IX> __device__ __forceinline__ void fm_nbr_finish(const IX &ix, const FmNbr &q, uint64_t *os, uint64_t *oe) {
    uint64_t ok = q.ok, ol = q.ol;
    if (q.l_plain) ol = fm_part_rank(q.bl, q.ll, q.c);
    if (q.k_plain) {
        FmPart pk;               // the block of k: the one loaded for l when they share it
        pk.cnt = q.shared ? q.bl.cnt : q.bk.cnt;
        pk.lo = q.shared ? q.bl.lo : q.bk.lo;
        pk.hi = q.shared ? q.bl.hi : q.bk.hi;
        ok = fm_part_rank(pk, q.kk, q.c);
    }
    const uint64_t l2c = fm_L2(ix, q.c);
    *os = l2c + ok + 1;
    *oe = l2c + ol;
}
template <class IX> __device__ __forceinline__ void fm_get_neighbor(const IX &ix, uint64_t s, uint64_t e, uint32_t c,
                                                uint64_t *os, uint64_t *oe) {
    const FmNbr q = fm_nbr_issue(ix, s, e, c);
    fm_nbr_finish(ix, q, os, oe);
}


// ---- The rank table k_map works on when the reference has fewer than 2^32 rows (DevIndex::fm32; built from the BWA blocks
// at index load, k_taps.hip): one 32-byte block per 64 BWT symbols =
//   4 x u32   L2[c] + occurrences of c before the block  (so a rank needs neither the L2 table nor 64-bit arithmetic)
//   4 x u32   the 64 symbols as two bit planes: hi.lo, hi.hi, lo.lo, lo.hi (bit j = high / low bit of symbol 64 b + j)
// A rank is the count word of its base (one 4-byte load), one 16-byte load, two XORs per plane word and two popcounts.
// Same function as bwt_2occ behind BwaIndex::get_neighbor (bwa_index.hpp:158-162); the row k = seq_len needs no special
// case here (it is the last symbol's block), the row k = -1 cannot occur (path ranges start at row 1 or later).
__device__ __forceinline__ uint32_t fm32_rank(uint32_t cnt, const uint4 &pl, uint32_t xh, uint32_t xl, uint32_t kk) {
    const uint32_t r = kk & 63u;
    const uint64_t mask = (2ull << r) - 1ull;                       // symbols 0 .. r of the block (r = 63: all)
    const uint32_t m_lo = (pl.x ^ xh) & (pl.z ^ xl) & (uint32_t)mask;
    const uint32_t m_hi = (pl.y ^ xh) & (pl.w ^ xl) & (uint32_t)(mask >> 32);
    return cnt + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
}
// one backward-search step of the rows [s, e] with base c: *os > *oe when no row is left
template <class IX> __device__ __forceinline__ void fm32_get_neighbor(const IX &ix, uint32_t s, uint32_t e, uint32_t c, uint32_t *os, uint32_t *oe) {
    const uint32_t primary = (uint32_t)ix.primary;
    const uint32_t k = s - 1u, l = e;
    const uint32_t kk = k - (k >= primary ? 1u : 0u), ll = l - (l >= primary ? 1u : 0u);
    const uint32_t bk = kk >> 6, bl = ll >> 6;
    const auto w = ix.fm32;
    // both blocks are requested before either rank is computed; nearly always they are one and the same.  (Round 6, off the ISA: written
    // as `ck = cl; pk = pl; if (bk != bl) { ck = ...; pk = ...; }` the copies were USES of the first block's words: the wavefront waited
    // for block l before it asked for block k -- two memory round trips in a row in the four passes out of five in which some lane's
    // range straddles two blocks.  The second block gets registers of its own, preset to constants, and is chosen after both requests.)
    const uint32_t cl = w[(bl << 3) + c];
    const uint4 pl = ld16(w + (bl << 3) + 4u);
    const bool two = bk != bl;
    uint32_t ck2 = 0u;
    uint4 pk2 = make_uint4(0u, 0u, 0u, 0u);
    if (two) { ck2 = w[(bk << 3) + c]; pk2 = ld16(w + (bk << 3) + 4u); }
    mem_retire(ck2); mem_retire(pk2);      // (values of their own from here on: else the optimiser folds the selects back into the branch)
    const uint32_t xh = (c & 2u) ? 0u : ~0u, xl = (c & 1u) ? 0u : ~0u;     // plane word ^ x: bit set where the symbol's bit equals c's
    *oe = fm32_rank(cl, pl, xh, xl, ll);
    const uint32_t ck = two ? ck2 : cl;
    const uint4 pk = make_uint4(two ? pk2.x : pl.x, two ? pk2.y : pl.y, two ? pk2.z : pl.z, two ? pk2.w : pl.w);
    *os = fm32_rank(ck, pk, xh, xl, kk) + 1u;
}

// bwt_sa: walk LF until a sampled row (multiple of 32); *steps gets the number of LF steps
template <class IX> __device__ __forceinline__ uint64_t fm_sa(const IX &ix, uint64_t k, uint32_t *steps) {
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
        k = fm_L2(ix, c) + fm_block_count(b, c) + fm_block_rank(b, kk, c);
    }
    *steps = n;
    return (uint64_t)n + ix.sa[k >> 5];
}

// SA look-up through the dense table built at index load (k_dense_sa): SIX bytes per row -- SA value (34 bits: seq_len < 2^34 is the
// library's limit) | LF steps the BWA-format walk above would have taken << 34 (14 bits; kept so that the SURVEY 8(d) work counters
// stay those of the reference's algorithm.  The walk ends at a row that is a multiple of 32, which each LF step reaches with
// probability 1/32: the count is geometric -- 13 % of the rows take more than 63 steps -- and 14 bits are beyond it by a margin of
// e^-500; the load-time self-check refuses a table with a count that does not fit).  One 8-byte read (the two aligned words around
// the entry) instead of ~31 dependent 64-byte reads.  Rounds 2-5 spent 8 bytes per row: 49.6 GB of GRCh38's 252 GB per GPU, now 37.2.
constexpr uint64_t SA_ENTRY_BYTES = 6;
constexpr int SA_STEP_SHIFT = 34;
constexpr uint32_t SA_STEP_MAX = (1u << 14) - 1u;
__device__ __forceinline__ uint64_t sa_pack(uint64_t sa, uint32_t steps) { return (sa & ((1ull << SA_STEP_SHIFT) - 1ull)) | ((uint64_t)(steps & SA_STEP_MAX) << SA_STEP_SHIFT); }
template <class P> __device__ __forceinline__ uint64_t sa_entry_load(P table, uint64_t k) {
    const uint64_t byte = k * SA_ENTRY_BYTES;
    const auto w = (const UNC_AS_GLOBAL uint32_t *)table + (byte >> 2);      // (C-style: the table arrives as a generic or a global pointer)
    const uint64_t two = ((uint64_t)w[1] << 32) | w[0];
    return (two >> ((byte & 3u) << 3)) & ((1ull << 48) - 1ull);
}
template <class IX> __device__ __forceinline__ uint64_t fm_sa_dense(const IX &ix, uint64_t k, uint32_t *steps) {
    const uint64_t v = sa_entry_load(ix.sa_dense, k);
    *steps = (uint32_t)(v >> SA_STEP_SHIFT);
    return v & ((1ull << SA_STEP_SHIFT) - 1ull);
}

}  // namespace unc
