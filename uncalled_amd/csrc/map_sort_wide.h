// k_map.hip, part 5: the runs-and-merge sort of map_sort.h for WIDE keys (human-sized references: 64-bit rows, the 128-bit SortKey
// = range | seed_prob, creation index, flags, k-mer).  Round 3 sorted these with the bitonic network through memory -- a third of
// k_map on GRCh38.  The children's keys leave phase E as the same six runs the narrow mode files them in; an element is the whole
// 16-byte key (it carries what the narrow mode keeps in the separate info word), the order is (a, b) -- a total order, b holds the
// creation index -- and a merge tile holds 320 keys.
#pragma once

namespace unc {

__device__ __forceinline__ bool wk_lt(const SortKey &x, const SortKey &y) { return x.a < y.a || (x.a == y.a && x.b < y.b); }
__device__ __forceinline__ SortKey wk_max() { SortKey k; k.a = ~0ull; k.b = ~0ull; return k; }
__device__ __forceinline__ SortKey wk_bcast(const SortKey &k, int src) { SortKey r; r.a = bcast64(k.a, src); r.b = bcast64(k.b, src); return r; }

// a sequence of R runs of 16-byte keys back to back (adj: byte offset of run r inside the slot minus 16 * its first index)
template <int R> __device__ __forceinline__ SortKey kaw_load(cgptr_t sb, const KeyArr<R> &K, uint32_t i) {
    // (map_sort.h's ka_adj: a sum of guarded differences.  Rounds 3-5 kept a select chain here, for which the optimiser holds `adj` in
    // scratch memory -- every key load behind a scratch load and a full wait; round 5 measured the fix as a loss on GRCh38, on launches
    // whose times come in 10 % quanta.  Round 6 counts round trips instead: the five loads that stage a tile were five trips in a row.)
    return gld<SortKey>(sb, ka_adj(K, i) + (i << 4));
}
__device__ __forceinline__ KeyArr<1> kaw_single(uint32_t off, uint32_t n) { KeyArr<1> K; K.adj[0] = off; K.cum[0] = 0; K.n = n; return K; }

constexpr uint32_t MERGEW_C = 5;
constexpr uint32_t MERGEW_TILE = MERGEW_C * WAVE;           // 320 keys = 5 KB of the staging buffer (+ the key that waits for its successor)
static_assert((MERGEW_TILE + 1) * sizeof(SortKey) <= S_E_WORDS * 8, "wide merge tile must fit the staging buffer");

// a tile's share of A and B into the tile buffer: map_sort.h's stage_tile for 16-byte keys (five requests in a row, one wait)
template <int RA, int RB>
__device__ __forceinline__ void stagew_tile(cgptr_t sb, const KeyArr<RA> &A, const KeyArr<RB> &B, uint32_t a0, uint32_t b0, uint32_t na, uint32_t tn,
                                            SortKey *tile, int lane) {
    SortKey v[MERGEW_C];
#pragma unroll
    for (uint32_t c = 0; c < MERGEW_C; ++c) {
        const uint32_t i0 = (uint32_t)lane + c * WAVE, i = i0 < tn ? i0 : tn - 1u;
        const uint32_t ja = a0 + i, jb = b0 + (i - na);
        const uint32_t off = i < na ? ka_adj(A, ja) + (ja << 4) : ka_adj(B, jb) + (jb << 4);
        v[c] = gld<SortKey>(sb, off);
    }
#pragma unroll
    for (uint32_t c = 0; c < MERGEW_C; ++c) {
        const uint32_t i = (uint32_t)lane + c * WAVE;
        if (i < tn) tile[i] = v[c];
    }
}

template <int RA, int RB>
__device__ __forceinline__ uint32_t mergew_split(cgptr_t sb, const KeyArr<RA> &A, const KeyArr<RB> &B, uint32_t d, int lane) {
    uint32_t lo = d > B.n ? d - B.n : 0u, hi = d < A.n ? d : A.n;
    while (lo < hi) {
        const uint32_t span = hi - lo, step = (span + SPLIT_PROBES - 1u) / SPLIT_PROBES;
        const uint32_t p = lo + (uint32_t)lane * step;
        bool less = false;
        if ((uint32_t)lane < SPLIT_PROBES && p < hi) less = wk_lt(kaw_load(sb, A, p), kaw_load(sb, B, d - 1u - p));
        const uint32_t c = (uint32_t)__popcll(__ballot(less));
        const uint32_t nlo = c ? lo + (c - 1u) * step + 1u : lo;
        const uint32_t nhi = lo + c * step < hi ? lo + c * step : hi;
        lo = nlo; hi = c ? nhi : lo;
    }
    return lo;
}

// one tile of merge(A[a0..a1), B[b0..b1)) staged in `tile` and merged: lane's outputs o[0..cnt) = logical positions d .. d + cnt
__device__ __forceinline__ void mergew_tile(SortKey *tile, uint32_t na, uint32_t nb, int lane, SortKey (&o)[MERGEW_C], uint32_t &d, uint32_t &cnt) {
    const uint32_t tn = na + nb;
    d = (uint32_t)lane * MERGEW_C < tn ? (uint32_t)lane * MERGEW_C : tn;
    cnt = tn - d < MERGEW_C ? tn - d : MERGEW_C;
    uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
    while (__any(lo < hi)) {
        if (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (wk_lt(tile[mid], tile[na + d - 1u - mid])) lo = mid + 1u; else hi = mid;
        }
    }
    uint32_t ia = lo, ib = d - lo;
    // field by field: a select between two STRUCTS is a select between their addresses to the optimiser, which then keeps va, vb and o
    // in scratch memory (17 scratch stores and 11 scratch loads per tile, each load behind a full wait: round 5 saw it and measured the
    // fix inside GRCh38's 10 % launch-time quanta; round 6 counts the waits)
    uint64_t vaa = ~0ull, vab = ~0ull, vba = ~0ull, vbb = ~0ull;
    if (ia < na) { const SortKey t = tile[ia]; vaa = t.a; vab = t.b; }
    if (ib < nb) { const SortKey t = tile[na + ib]; vba = t.a; vbb = t.b; }
#pragma unroll
    for (uint32_t c = 0; c < MERGEW_C; ++c) {
        const bool ta = vaa < vba || (vaa == vba && vab < vbb);
        o[c].a = ta ? vaa : vba; o[c].b = ta ? vab : vbb;
        if (ta) ++ia; else ++ib;
        const uint32_t idx = ta ? ia : na + ib;
        const bool ok = ta ? ia < na : ib < nb;
        uint64_t xa = ~0ull, xb = ~0ull;
        if (ok && c + 1u < cnt) { const SortKey t = tile[idx]; xa = t.a; xb = t.b; }
        vaa = ta ? xa : vaa; vab = ta ? xb : vab;
        vba = ta ? vba : xa; vbb = ta ? vbb : xb;
    }
}

// merge(A, B) -> out (byte offset in the slot); returns 1 (the inputs are ascending by construction: repaired runs, a sorted run)
template <int RA, int RB>
static __device__ __noinline__ void mergew_runs(gptr_t sb_, KeyArr<RA> A_, KeyArr<RB> B_, uint32_t out_off_, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    SortKey *const tile = reinterpret_cast<SortKey *>(s_e);
    const KeyArr<RA> A = ka_uniform(A_);
    const KeyArr<RB> B = ka_uniform(B_);
    const uint32_t out_off = uniform32(out_off_);
    const uint32_t n = A.n + B.n;
    uint32_t a0 = 0, b0 = 0;
    for (uint32_t o0 = 0; o0 < n; o0 += MERGEW_TILE) {
        const uint32_t d1 = o0 + MERGEW_TILE < n ? o0 + MERGEW_TILE : n;
        const uint32_t a1 = d1 == n ? A.n : mergew_split(sb, A, B, d1, lane);
        const uint32_t b1 = d1 - a1;
        const uint32_t na = a1 - a0, nb = b1 - b0, tn = na + nb;
        stagew_tile(sb, A, B, a0, b0, na, tn, tile, lane);
        wave_sync();
        SortKey o[MERGEW_C];
        uint32_t d, cnt;
        mergew_tile(tile, na, nb, lane, o, d, cnt);
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGEW_C; ++c)
            if (c < cnt) tile[d + c] = o[c];
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGEW_C; ++c) {
            const uint32_t i = (uint32_t)lane + c * WAVE;
            if (i < tn) gst(sb, out_off + ((o0 + i) << 4), tile[i]);
        }
        a0 = a1; b0 = b1;
        wave_sync();
    }
}

// a run of moves with one base: keys that are smaller than one before them go to the unsorted run, what is left is ascending
// (map_sort.h, repair_run, for 16-byte keys)
static __device__ __noinline__ uint32_t repairw_run(gptr_t sb_, uint32_t run_off_, uint32_t n_, uint32_t x_off_, uint32_t nx_, int lane) {
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t run_off = uniform32(run_off_), n = uniform32(n_), x_off = uniform32(x_off_);
    uint32_t nx = uniform32(nx_), shift = 0;
    SortKey carry; carry.a = 0; carry.b = 0;             // the largest key so far (keys are > 0)
    SortKey knext; knext.a = 0; knext.b = 0;
    if ((uint32_t)lane < n) knext = gld<SortKey>(sb, run_off + ((uint32_t)lane << 4));
    for (uint32_t c0 = 0; c0 < n; c0 += WAVE) {
        const uint32_t i = c0 + (uint32_t)lane;
        const bool have = i < n;
        const SortKey k = knext;
        knext.a = 0; knext.b = 0;
        if (i + WAVE < n) knext = gld<SortKey>(sb, run_off + ((i + WAVE) << 4));
        SortKey pk;
        pk.a = (uint64_t)__shfl_up((unsigned long long)k.a, 1); pk.b = (uint64_t)__shfl_up((unsigned long long)k.b, 1);
        if (lane == 0) pk = carry;
        bool viol = have && wk_lt(k, pk);
        const uint32_t nvalid = n - c0 < (uint32_t)WAVE ? n - c0 : (uint32_t)WAVE;
        if (__any(viol)) {
            // exact running maximum (rare): a serial pass over the chunk's keys, lane by lane
            SortKey run = carry;
            for (uint32_t j = 0; j < nvalid; ++j) {
                const SortKey kj = wk_bcast(k, (int)j);
                const bool v = wk_lt(kj, run);
                if ((uint32_t)lane == j) viol = v;
                if (!v) run = kj;
            }
            carry = run;
        } else carry = wk_bcast(k, (int)nvalid - 1);
        const uint64_t vm = __ballot(viol);
        if (vm == 0 && shift == 0) continue;
        wave_sync();
        const uint32_t before = (uint32_t)prefix_popc(vm);
        if (have) {
            if (viol) gst(sb, x_off + ((nx + before) << 4), k);
            else gst(sb, run_off + ((i - shift - before) << 4), k);
        }
        const uint32_t nv = (uint32_t)__popcll(vm);
        shift += nv; nx += nv;
        wave_sync();
    }
    return shift;
}

}  // namespace unc
