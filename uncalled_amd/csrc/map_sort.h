// k_map.hip, part 4: the children's keys in the reference's order -- bitonic networks (wide keys, small events), and for the
// narrow keys the runs phase E files them in, their repair, the merge-path merge.
#pragma once

namespace unc {

// ---- sorting ------------------------------------------------------------------------------------

__device__ __forceinline__ bool key_gt(uint64_t a1, uint64_t b1, uint64_t a2, uint64_t b2) {
    return a1 > a2 || (a1 == a2 && b1 > b2);
}

// Bitonic network over keys held E per lane (block element q = lane*E + e, global index p = base + q).
// merge_stages runs the stages j = j_from, j_from/2, .., 1 of merge size k: the j < E stages stay inside a
// lane (static register indices), the j >= E stages cross lanes (at most 6 per merge).
template <int E>
__device__ __forceinline__ void merge_stages(uint64_t (&a)[E], uint64_t (&b)[E], uint32_t base, uint32_t k, uint32_t j_from,
                                             int lane) {
    for (uint32_t j = j_from; j > 0; j >>= 1) {
        if (j >= (uint32_t)E) {
            const uint32_t d = j / (uint32_t)E;          // lane distance
            const bool lower = ((uint32_t)lane & d) == 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                uint64_t pa = xor_lane64(a[e], d);
                uint64_t pb = xor_lane64(b[e], d);
                uint32_t p = base + (uint32_t)lane * E + (uint32_t)e;
                bool up = (p & k) == 0;
                bool want_min = lower == up;
                bool gt = key_gt(a[e], b[e], pa, pb);
                if (want_min ? gt : !gt) { a[e] = pa; b[e] = pb; }
            }
        } else {
#pragma unroll
            for (int jj = 1; jj < E; jj <<= 1) {
                if (j == (uint32_t)jj) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (!(e & jj)) {
                            const int pe = e | jj;
                            uint32_t p = base + (uint32_t)lane * E + (uint32_t)e;
                            bool up = (p & k) == 0;
                            bool gt = key_gt(a[e], b[e], a[pe], b[pe]);
                            if (up ? gt : !gt) {
                                uint64_t ta = a[e], tb = b[e];
                                a[e] = a[pe]; b[e] = b[pe];
                                a[pe] = ta; b[pe] = tb;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void block_load(uint64_t (&a)[E], uint64_t (&b)[E], const UNC_AS_GLOBAL SortKey *in, uint32_t base, uint32_t n, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
        if (i < n) { const SortKey k = g_load(in + i); a[e] = k.a; b[e] = k.b; }
        else { a[e] = ~0ull; b[e] = ~0ull; }
    }
}
template <int E>
__device__ __forceinline__ void block_store(const uint64_t (&a)[E], const uint64_t (&b)[E], UNC_AS_GLOBAL SortKey *out, uint32_t base, uint32_t lim, int lane) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
        if (i < lim) { SortKey k; k.a = a[e]; k.b = b[e]; g_store(out + i, k); }
    }
}

// n <= 64*E: the whole sort in registers
template <int E>
static __device__ __noinline__ void sort_regs(const UNC_AS_GLOBAL SortKey *in_, UNC_AS_GLOBAL SortKey *out_, uint32_t n_, int lane) {
    const UNC_AS_GLOBAL SortKey *const in = uniform_ptr(in_);
    UNC_AS_GLOBAL SortKey *const out = uniform_ptr(out_);
    const uint32_t n = uniform32(n_);
    uint64_t a[E], b[E];
    block_load<E>(a, b, in, 0, n, lane);
    for (uint32_t k = 2; k <= 64u * E; k <<= 1) merge_stages<E>(a, b, 0, k, k >> 1, lane);
    block_store<E>(a, b, out, 0, n, lane);
}

constexpr int GS_BATCH = 4;     // passes of a global sort stage whose loads are issued together (N / 2 / 64 >= 8 passes)
// n > 512: 512-key blocks are sorted / merged in registers, only the stages with j >= 512 go through memory
static __device__ __noinline__ void sort_hybrid(const UNC_AS_GLOBAL SortKey *in_, UNC_AS_GLOBAL SortKey *out_, uint32_t n_, int lane) {
    const UNC_AS_GLOBAL SortKey *const in = uniform_ptr(in_);
    UNC_AS_GLOBAL SortKey *const out = uniform_ptr(out_);
    const uint32_t n = uniform32(n_);
    constexpr int E = 8;
    constexpr uint32_t B = 64u * E;
    uint32_t N = 2 * B;
    while (N < n) N <<= 1;
    uint64_t a[E], b[E];
    for (uint32_t base = 0; base < N; base += B) {      // padded blocks sort like any other (keys = max)
        block_load<E>(a, b, in, base, n, lane);
        for (uint32_t k = 2; k <= B; k <<= 1) merge_stages<E>(a, b, base, k, k >> 1, lane);
        block_store<E>(a, b, out, base, N, lane);
    }
    wave_sync();
    for (uint32_t k = 2 * B; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j >= B; j >>= 1) {
            // the pairs of one stage are disjoint: four passes' worth of loads are in flight before the first store (the
            // stage is otherwise one dependent memory round trip per 64 pairs)
            for (uint32_t t0 = 0; t0 < N / 2; t0 += 64 * GS_BATCH) {
                SortKey x[GS_BATCH], y[GS_BATCH];
                uint32_t ii[GS_BATCH];
#pragma unroll
                for (int u = 0; u < GS_BATCH; ++u) {
                    const uint32_t t = t0 + 64u * u + (uint32_t)lane;
                    ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    x[u] = g_load(out + ii[u]); y[u] = g_load(out + (ii[u] | j));
                }
#pragma unroll
                for (int u = 0; u < GS_BATCH; ++u) {
                    const bool up = (ii[u] & k) == 0;
                    const bool gt = key_gt(x[u].a, x[u].b, y[u].a, y[u].b);
                    if (up ? gt : !gt) { g_store(out + ii[u], y[u]); g_store(out + (ii[u] | j), x[u]); }
                }
            }
            wave_sync();
        }
        for (uint32_t base = 0; base < N; base += B) {
            block_load<E>(a, b, out, base, N, lane);
            merge_stages<E>(a, b, base, k, B >> 1, lane);
            block_store<E>(a, b, out, base, N, lane);
        }
        wave_sync();
    }
}

// ---- narrow mode: start, length and creation index of a child fit one 64-bit key (index-dependent, decided at load:
// DevIndex::key_len_bits).  Half the data to move per sort stage; the seed_prob
// ordering inside runs of equal ranges is recovered afterwards by a segmented max over the children's info words.
// The network is the all-ascending form of the bitonic sorter: a merge of size k starts with the "flip" stage
// (element q against q ^ (k - 1), the mirror image inside its k-group) and continues with the half-cleaners
// j = k/4 .. 1 (q against q ^ j); the lower index always keeps the minimum.  Every sorted run is ascending, so the
// +inf padding behind the n real keys never moves and whole blocks / pairs made of padding are skipped.
template <int E>
__device__ __forceinline__ void asc_stages64(uint64_t (&a)[E], uint32_t j_from, int lane) {
    for (uint32_t j = j_from; j > 0; j >>= 1) {
        if (j >= (uint32_t)E) {
            const uint32_t d = j / (uint32_t)E;
            const bool lower = ((uint32_t)lane & d) == 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint64_t pa = xor_lane64(a[e], d);
                const bool gt = a[e] > pa;
                if (lower ? gt : !gt) a[e] = pa;
            }
        } else {
#pragma unroll
            for (int jj = 1; jj < E; jj <<= 1) {
                if (j == (uint32_t)jj) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (!(e & jj)) {
                            const int pe = e | jj;
                            if (a[e] > a[pe]) { const uint64_t t = a[e]; a[e] = a[pe]; a[pe] = t; }
                        }
                    }
                }
            }
        }
    }
}

// flip stage of a merge of size k <= 64 * E inside a block
template <int E>
__device__ __forceinline__ void flip_stage64(uint64_t (&a)[E], uint32_t k, int lane) {
    if (k > (uint32_t)E) {
        const uint32_t kl = k / (uint32_t)E;                 // lanes per k-group; partner lane = lane ^ (kl - 1), register E-1-e
        const bool lower = ((uint32_t)lane & (kl >> 1)) == 0;
        uint64_t pa[E];
#pragma unroll
        for (int e = 0; e < E; ++e) pa[e] = xor_lane64(a[E - 1 - e], kl - 1u);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool gt = a[e] > pa[e];
            if (lower ? gt : !gt) a[e] = pa[e];
        }
    } else {
#pragma unroll
        for (int kk = 2; kk <= E; kk <<= 1) {
            if (k == (uint32_t)kk) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int pe = e ^ (kk - 1);
                    if (pe > e && a[e] > a[pe]) { const uint64_t t = a[e]; a[e] = a[pe]; a[pe] = t; }
                }
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void block_sort64(uint64_t (&a)[E], int lane) {
    for (uint32_t k = 2; k <= 64u * E; k <<= 1) {
        flip_stage64<E>(a, k, lane);
        asc_stages64<E>(a, k >> 2, lane);
    }
}

// ---- narrow keys arrive as RUNS.  Phase E files a child's key by what it is: stays and the moves with base 0..3 of the
// sorted survivors (five runs that come out ascending: a stay keeps its parent's range, and one backward-search step with a
// fixed base maps ascending ranges to ascending ranges; the four bases' rows are disjoint blocks of the index in base
// order), and the children of sources (run 5, no order).  KeyArr<R> is a sequence made of R such runs back to back; every
// field is uniform.
template <int R> struct KeyArr {
    uint32_t adj[R];    // byte offset of run r inside the slot, minus 8 * (first index of run r)
    uint32_t cum[R];    // first index of run r (cum[0] = 0)
    uint32_t n;
};
// byte offset of element i: adj of the run it falls into.  Written as a sum of guarded DIFFERENCES on purpose.  The natural form
// (a = i >= cum[r] ? adj[r] : a) is turned by the optimiser into a load of adj[selected index]: the array then lives in scratch
// memory, every key load is preceded by a scratch load + s_waitcnt vmcnt(0), and the nine loads that stage a merge tile -- meant to
// be ONE memory round trip -- became nine in a row (round 5, read off the ISA: 10-11 scratch loads in each merge function).
template <int R> __device__ __forceinline__ uint32_t ka_adj(const KeyArr<R> &K, uint32_t i) {
    uint32_t a = K.adj[0];
#pragma unroll
    for (int r = 1; r < R; ++r) a += i >= K.cum[r] ? K.adj[r] - K.adj[r - 1] : 0u;
    return a;
}
template <int R> __device__ __forceinline__ uint64_t ka_load(cgptr_t sb, const KeyArr<R> &K, uint32_t i) {
    return gld<uint64_t>(sb, ka_adj(K, i) + (i << 3));
}
template <int R> __device__ __forceinline__ KeyArr<R> ka_uniform(const KeyArr<R> &K) {
    KeyArr<R> U;
#pragma unroll
    for (int r = 0; r < R; ++r) { U.adj[r] = uniform32(K.adj[r]); U.cum[r] = uniform32(K.cum[r]); }
    U.n = uniform32(K.n);
    return U;
}
__device__ __forceinline__ KeyArr<1> ka_single(uint32_t off, uint32_t n) { KeyArr<1> K; K.adj[0] = off; K.cum[0] = 0; K.n = n; return K; }

// n <= 64 * E keys of K -> out (byte offset in the slot), sorted
template <int E>
static __device__ __noinline__ void sort_regs64(gptr_t sb_, KeyArr<6> K_, uint32_t out_off_, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t out_off = uniform32(out_off_);
    const KeyArr<6> K = ka_uniform(K_);
    const uint32_t n = K.n;
    uint64_t a[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)lane * E + (uint32_t)e;
        a[e] = i < n ? ka_load(sb, K, i) : ~0ull;
    }
    block_sort64<E>(a, lane);
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)lane * E + (uint32_t)e;
        if (i < n) gst(sb, out_off + (i << 3), a[e]);
    }
}

static __device__ __noinline__ void sort_hybrid64(gptr_t sb_, KeyArr<6> K_, uint32_t out_off, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    const KeyArr<6> K = ka_uniform(K_);
    const uint32_t n = K.n;
    UNC_AS_GLOBAL uint64_t *const out = reinterpret_cast<UNC_AS_GLOBAL uint64_t *>(sb + uniform32(out_off));
    constexpr int E = 8;
    constexpr uint32_t B = 64u * E;
    uint32_t N = 2 * B;
    while (N < n) N <<= 1;
    uint64_t a[E];
    for (uint32_t base = 0; base < n; base += B) {          // blocks that hold real keys
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
            a[e] = i < n ? ka_load(sb, K, i) : ~0ull;
        }
        block_sort64<E>(a, lane);
        wave_sync();                                          // (in place: the block is loaded before any of it is stored)
        // only positions < n are ever stored or loaded: the output may be a run of exactly n keys' room (the unsorted stream sorted
        // in place), and the padding of a sorted block sits behind its real keys
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
            if (i < n) out[i] = a[e];
        }
    }
    wave_sync();
    for (uint32_t k = 2 * B; k <= N; k <<= 1) {
        // flip: i against i ^ (k - 1); a pair whose upper element is padding (>= n) has nothing to exchange.  The pairs of
        // one stage are disjoint: four passes' worth of loads are in flight before the first store (a stage is otherwise one
        // dependent memory round trip per 64 pairs, and three such stages were most of this sort's time)
        for (uint32_t t0 = 0; t0 < N / 2; t0 += 64 * GS_BATCH) {
            uint64_t x[GS_BATCH], y[GS_BATCH];
            uint32_t ii[GS_BATCH], pp[GS_BATCH];
#pragma unroll
            for (int u = 0; u < GS_BATCH; ++u) {
                const uint32_t t = t0 + 64u * u + (uint32_t)lane;
                ii[u] = ((t & ~((k >> 1) - 1)) << 1) | (t & ((k >> 1) - 1));
                pp[u] = ii[u] ^ (k - 1);
                x[u] = 0; y[u] = 0;
                if (pp[u] < n) { x[u] = out[ii[u]]; y[u] = out[pp[u]]; }
            }
#pragma unroll
            for (int u = 0; u < GS_BATCH; ++u)
                if (pp[u] < n && x[u] > y[u]) { out[ii[u]] = y[u]; out[pp[u]] = x[u]; }
        }
        wave_sync();
        for (uint32_t j = k >> 2; j >= B; j >>= 1) {
            for (uint32_t t0 = 0; t0 < N / 2; t0 += 64 * GS_BATCH) {
                if (uniform32((((t0 & ~(j - 1)) << 1) | (t0 & (j - 1))) | j) >= n) continue;   // p grows with t: the whole batch is padding
                uint64_t x[GS_BATCH], y[GS_BATCH];
                uint32_t ii[GS_BATCH];
#pragma unroll
                for (int u = 0; u < GS_BATCH; ++u) {
                    const uint32_t t = t0 + 64u * u + (uint32_t)lane;
                    ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    x[u] = 0; y[u] = 0;
                    if ((ii[u] | j) < n) { x[u] = out[ii[u]]; y[u] = out[ii[u] | j]; }
                }
#pragma unroll
                for (int u = 0; u < GS_BATCH; ++u)
                    if ((ii[u] | j) < n && x[u] > y[u]) { out[ii[u]] = y[u]; out[ii[u] | j] = x[u]; }
            }
            wave_sync();
        }
        // the register stages of this merge, two blocks per trip to memory where a second one with real keys exists
        for (uint32_t base = 0; base < n; base += 2 * B) {
            const bool two = base + B < n;
            uint64_t b2[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
                a[e] = i < n ? out[i] : ~0ull;
                b2[e] = i + B < n ? out[i + B] : ~0ull;
            }
            asc_stages64<E>(a, B >> 1, lane);
            if (two) asc_stages64<E>(b2, B >> 1, lane);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = base + (uint32_t)lane * E + (uint32_t)e;
                if (i < n) out[i] = a[e];
                if (i + B < n) out[i + B] = b2[e];
            }
        }
        wave_sync();
    }
}

// any number of keys, by size class
static __device__ __noinline__ void sort_any64(gptr_t sb, KeyArr<6> K, uint32_t out_off, int lane) {
    const uint32_t n = uniform32(K.n);
    if (n <= 64) sort_regs64<1>(sb, K, out_off, lane);
    else if (n <= 128) sort_regs64<2>(sb, K, out_off, lane);
    else if (n <= 256) sort_regs64<4>(sb, K, out_off, lane);
    else if (n <= 512) sort_regs64<8>(sb, K, out_off, lane);
    else sort_hybrid64(sb, K, out_off, lane);
}

// ---- merging two ascending runs (merge path).  The output is cut into tiles of MERGE_TILE keys; a tile's share of A
// and B is staged in LDS, each lane finds where its MERGE_C outputs start by a binary search on its diagonal and merges
// them sequentially.  Work per key: one LDS read and a dozen lane instructions, against ~60 compare-exchanges of the
// bitonic network -- provided the inputs ARE ascending.  The caller checks the result (`verify`) and sorts the keys the
// hard way if it is not (an event where the runs of phase E were not ascending after all).
constexpr uint32_t MERGE_MIN = 256;     // fewer children than this go straight through the bitonic network
#ifndef UNC_MERGE_REPAIR
#define UNC_MERGE_REPAIR 1              // (tests build the emulator library with 0: the runs then reach the merge unrepaired, its check
#endif                                  //  must notice and the event must take the bitonic network instead, with the same result)
constexpr bool MERGE_REPAIR = UNC_MERGE_REPAIR != 0;
constexpr uint32_t MERGE_C = 9;
constexpr uint32_t MERGE_TILE = MERGE_C * WAVE;
// LDS slot of tile element i: one pad slot per 8 keys, so that lanes whose reading positions are a multiple of 8 keys
// apart (the typical distance) do not all fall on the same banks
__device__ __forceinline__ uint32_t mslot(uint32_t i) { return i + (i >> 3); }
constexpr uint32_t MERGE_LDS_KEYS = MERGE_TILE + MERGE_TILE / 8 + 1;
static_assert(MERGE_LDS_KEYS <= S_E_WORDS && NKMER * 4 <= S_E_WORDS * 8, "merge tile / source list must fit the staging buffer");

// how many of the first d keys of merge(A, B) come from A (keys distinct): the first mid with !(A[mid] < B[d - 1 - mid]),
// 16 probes per memory round trip (a scattered access costs the memory pipeline per LANE: four rounds of 16 are cheaper
// than three of 64)
constexpr uint32_t SPLIT_PROBES = 16;
template <int RA, int RB>
__device__ __forceinline__ uint32_t merge_split(cgptr_t sb, const KeyArr<RA> &A, const KeyArr<RB> &B, uint32_t d, int lane) {
    uint32_t lo = d > B.n ? d - B.n : 0u, hi = d < A.n ? d : A.n;
    while (lo < hi) {
        const uint32_t span = hi - lo, step = (span + SPLIT_PROBES - 1u) / SPLIT_PROBES;
        const uint32_t p = lo + (uint32_t)lane * step;
        bool less = false;
        if ((uint32_t)lane < SPLIT_PROBES && p < hi) less = ka_load(sb, A, p) < ka_load(sb, B, d - 1u - p);
        const uint32_t c = (uint32_t)__popcll(__ballot(less));       // the predicate is monotone: the first c probes hold
        const uint32_t nlo = c ? lo + (c - 1u) * step + 1u : lo;
        const uint32_t nhi = lo + c * step < hi ? lo + c * step : hi;
        lo = nlo; hi = c ? nhi : lo;
    }
    return lo;
}

// A tile's share of A (na keys from a0) and of B (from b0) into the tile buffer, tn keys in all (tn > 0).  Nine requests in a row and ONE
// wait: every lane asks in every round (a lane past the tile's end for the tile's last key) and the address is a select, not a branch --
// written as `if (i < na) load A; else if (i < tn) load B` each load sat in a block of its own and the compiler drained the memory
// counter between them (round 6, off the ISA: three round trips in a row where one was meant).
template <int RA, int RB>
__device__ __forceinline__ void stage_tile(cgptr_t sb, const KeyArr<RA> &A, const KeyArr<RB> &B, uint32_t a0, uint32_t b0, uint32_t na, uint32_t tn,
                                           uint64_t *s_tile, int lane) {
    uint64_t v[MERGE_C];
#pragma unroll
    for (uint32_t c = 0; c < MERGE_C; ++c) {
        const uint32_t i0 = (uint32_t)lane + c * WAVE, i = i0 < tn ? i0 : tn - 1u;
        const uint32_t ja = a0 + i, jb = b0 + (i - na);
        const uint32_t off = i < na ? ka_adj(A, ja) + (ja << 3) : ka_adj(B, jb) + (jb << 3);
        v[c] = gld<uint64_t>(sb, off);
    }
#pragma unroll
    for (uint32_t c = 0; c < MERGE_C; ++c) {
        const uint32_t i = (uint32_t)lane + c * WAVE;
        if (i < tn) s_tile[mslot(i)] = v[c];
    }
}

template <int RA, int RB>
static __device__ __noinline__ uint32_t merge_runs(gptr_t sb_, KeyArr<RA> A_, KeyArr<RB> B_, uint32_t out_off_, int lane, uint32_t verify_) {
    const gptr_t sb = uniform_ptr(sb_);
    uint64_t *const s_tile = s_e;
    const KeyArr<RA> A = ka_uniform(A_);
    const KeyArr<RB> B = ka_uniform(B_);
    const uint32_t out_off = uniform32(out_off_), verify = uniform32(verify_);
    const uint32_t n = A.n + B.n;
    uint32_t a0 = 0, b0 = 0;
    uint64_t prev_last = 0;          // (keys are > 0: idx and length fields aside, start >= 1)
    bool bad = false;
    for (uint32_t o0 = 0; o0 < n; o0 += MERGE_TILE) {
        const uint32_t d1 = o0 + MERGE_TILE < n ? o0 + MERGE_TILE : n;
        const uint32_t a1 = d1 == n ? A.n : merge_split(sb, A, B, d1, lane);
        const uint32_t b1 = d1 - a1;
        const uint32_t na = a1 - a0, nb = b1 - b0, tn = na + nb;
        // stage the tile: every load is requested before the first key goes into LDS (one memory round trip, not twelve)
        stage_tile(sb, A, B, a0, b0, na, tn, s_tile, lane);
        wave_sync();
        // this lane's outputs [d, d + cnt)
        const uint32_t d = (uint32_t)lane * MERGE_C < tn ? (uint32_t)lane * MERGE_C : tn;
        const uint32_t cnt = tn - d < MERGE_C ? tn - d : MERGE_C;
        uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
        while (__any(lo < hi)) {
            if (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_tile[mslot(mid)] < s_tile[mslot(na + d - 1u - mid)]) lo = mid + 1u; else hi = mid;
            }
        }
        uint32_t ia = lo, ib = d - lo;
        uint64_t va = ia < na ? s_tile[mslot(ia)] : ~0ull, vb = ib < nb ? s_tile[mslot(na + ib)] : ~0ull;
        uint64_t o[MERGE_C];
#pragma unroll
        for (uint32_t c = 0; c < MERGE_C; ++c) {
            const bool ta = va < vb;
            o[c] = ta ? va : vb;
            if (ta) ++ia; else ++ib;
            const uint32_t idx = ta ? ia : na + ib;
            const bool ok = ta ? ia < na : ib < nb;
            uint64_t x = ~0ull;
            if (ok && c + 1u < cnt) x = s_tile[mslot(idx)];
            if (ta) va = x; else vb = x;
        }
        // out through the tile buffer, so that each store instruction writes 512 consecutive bytes: a lane storing its own twelve
        // keys (lanes 96 bytes apart) costs the CU's memory pipeline ten times as much (tools/dev/ubench_vmem.hip)
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGE_C; ++c)
            if (c < cnt) s_tile[mslot(d + c)] = o[c];
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGE_C; ++c) {
            const uint32_t i = (uint32_t)lane + c * WAVE;
            if (i < tn) gst(sb, out_off + ((o0 + i) << 3), s_tile[mslot(i)]);
        }
        if (verify) {
            uint64_t last = o[0];
            bool w = false;
#pragma unroll
            for (uint32_t c = 1; c < MERGE_C; ++c)
                if (c < cnt) { w = w || !(o[c] > last); last = o[c]; }
            uint64_t pl = (uint64_t)__shfl_up((unsigned long long)last, 1);
            if (lane == 0) pl = prev_last;
            if (cnt > 0 && !(o[0] > pl)) w = true;
            if (__any(w)) bad = true;
            const uint32_t ll = (tn - 1u) / MERGE_C;        // the last lane with outputs (tn > 0)
            prev_last = bcast64(last, (int)ll);
        }
        a0 = a1; b0 = b1;
        wave_sync();
    }
    return bad ? 0u : 1u;
}

// The moves of one base (run r of the streams) are ascending by START; two of them with equal starts can be out of order
// when their parents were nested ranges (the outer parent comes first and its child can be the longer range).  A key that
// is smaller than one before it is moved to the unsorted run: what is left is ascending.  Nearly every 64-key chunk has
// no such key (one compare with the neighbour lane says so); a chunk that has one takes the exact running maximum.
static __device__ __noinline__ uint32_t repair_run(gptr_t sb_, uint32_t run_off_, uint32_t n_, uint32_t x_off_, uint32_t nx_, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t run_off = uniform32(run_off_), n = uniform32(n_), x_off = uniform32(x_off_);
    uint32_t nx = uniform32(nx_), shift = 0;
    uint64_t carry = 0;              // the largest key so far (keys are > 0)
    uint64_t knext = (uint32_t)lane < n ? gld<uint64_t>(sb, run_off + ((uint32_t)lane << 3)) : 0ull;
    for (uint32_t c0 = 0; c0 < n; c0 += WAVE) {
        const uint32_t i = c0 + (uint32_t)lane;
        const bool have = i < n;
        const uint64_t k = knext;
        // the next chunk is requested before this one is worked on (what this pass stores lies below what the next one reads)
        knext = i + WAVE < n ? gld<uint64_t>(sb, run_off + ((i + WAVE) << 3)) : 0ull;
        uint64_t pk = (uint64_t)__shfl_up((unsigned long long)k, 1);
        if (lane == 0) pk = carry;
        bool viol = have && k < pk;
        const uint32_t nvalid = n - c0 < (uint32_t)WAVE ? n - c0 : (uint32_t)WAVE;
        if (__any(viol)) {
            const uint64_t inc = seg_incl_max64(k, lane == 0);
            uint64_t ex = (uint64_t)__shfl_up((unsigned long long)inc, 1);
            if (lane == 0) ex = 0;
            if (ex < carry) ex = carry;
            viol = have && k < ex;
            const uint64_t top = bcast64(inc, WAVE - 1);
            if (top > carry) carry = top;
        } else carry = bcast64(k, (int)nvalid - 1);
        const uint64_t vm = __ballot(viol);
        if (vm == 0 && shift == 0) continue;
        if constexpr (!MERGE_REPAIR) { shift += (uint32_t)__popcll(vm); continue; }      // (test build: count, leave in place)
        wave_sync();
        const uint32_t before = (uint32_t)prefix_popc(vm);
        if (have) {
            if (viol) gst(sb, x_off + ((nx + before) << 3), k);
            else gst(sb, run_off + ((i - shift - before) << 3), k);
        }
        const uint32_t nv = (uint32_t)__popcll(vm);
        shift += nv; nx += nv;
        wave_sync();
    }
    return shift;
}

// The four runs of moves repaired TOGETHER: chunk j of every run is requested before any of them is looked at, so that the repair
// costs one memory round trip per 64 keys of the LONGEST run instead of one per 64 keys of every run, one run after the other
// (repair_run above, called four times, was 12 dependent round trips for four runs of 180 keys: 3 % of k_map).  Per run the logic is
// repair_run's; the misfits of all runs go to the unsorted run in whatever order the chunks come (it is sorted afterwards).
// Returns the number of keys taken out of run r in bits 16 (r - 1) .. 16 r - 1 (r = 1..4).
static __device__ __noinline__ uint64_t repair_runs4(gptr_t sb_, uint32_t str_off_, uint32_t run_bytes_, uint32_t n1_, uint32_t n2_, uint32_t n3_, uint32_t n4_,
                                                     uint32_t x_off_, uint32_t nx_, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t str_off = uniform32(str_off_), run_bytes = uniform32(run_bytes_), x_off = uniform32(x_off_);
    const uint32_t n[4] = {uniform32(n1_), uniform32(n2_), uniform32(n3_), uniform32(n4_)};
    uint32_t nx = uniform32(nx_);
    uint32_t shift[4] = {0, 0, 0, 0};
    uint64_t carry[4] = {0, 0, 0, 0};          // the largest key of the run so far (keys are > 0)
    uint32_t nmax = n[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) nmax = n[r] > nmax ? n[r] : nmax;
    for (uint32_t c0 = 0; c0 < nmax; c0 += WAVE) {
        const uint32_t i = c0 + (uint32_t)lane;
        uint64_t kk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)          // (a run of one key needs no repair, and is not touched)
            kk[r] = (n[r] > 1u && i < n[r]) ? gld<uint64_t>(sb, str_off + (uint32_t)(r + 1) * run_bytes + (i << 3)) : 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n[r] <= 1u || c0 >= n[r]) continue;
            const uint32_t run_off = str_off + (uint32_t)(r + 1) * run_bytes;
            const bool have = i < n[r];
            const uint64_t k = kk[r];
            uint64_t pk = (uint64_t)__shfl_up((unsigned long long)k, 1);
            if (lane == 0) pk = carry[r];
            bool viol = have && k < pk;
            const uint32_t nvalid = n[r] - c0 < (uint32_t)WAVE ? n[r] - c0 : (uint32_t)WAVE;
            if (__any(viol)) {
                const uint64_t inc = seg_incl_max64(k, lane == 0);
                uint64_t ex = (uint64_t)__shfl_up((unsigned long long)inc, 1);
                if (lane == 0) ex = 0;
                if (ex < carry[r]) ex = carry[r];
                viol = have && k < ex;
                const uint64_t top = bcast64(inc, WAVE - 1);
                if (top > carry[r]) carry[r] = top;
            } else carry[r] = bcast64(k, (int)nvalid - 1);
            const uint64_t vm = __ballot(viol);
            if (vm == 0 && shift[r] == 0) continue;
            if constexpr (!MERGE_REPAIR) { shift[r] += (uint32_t)__popcll(vm); continue; }      // (test build: count, leave in place)
            wave_sync();
            const uint32_t before = (uint32_t)prefix_popc(vm);
            if (have) {
                if (viol) gst(sb, x_off + ((nx + before) << 3), k);
                else gst(sb, run_off + ((i - shift[r] - before) << 3), k);      // (below what the next chunk of this run reads)
            }
            const uint32_t nv = (uint32_t)__popcll(vm);
            shift[r] += nv; nx += nv;
            wave_sync();
        }
    }
    return (uint64_t)shift[0] | ((uint64_t)shift[1] << 16) | ((uint64_t)shift[2] << 32) | ((uint64_t)shift[3] << 48);
}

}  // namespace unc
