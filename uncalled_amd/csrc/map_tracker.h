// k_map.hip, part 3: SeedTracker (seed_tracker.cpp:56-73,129-232) as a grid of buckets over ref_en_.start.
#pragma once

namespace unc {

// UNC_DBG_SEED (dev build, tools/dev/build_variants.py dbgseed="-DUNC_DBG_SEED=1"): what add_seeds did, through the profiling pass's
// counters 8..11 (phase E's parts go to counter 1): 8 rounds, 9 nodes walked (all lanes), 10 cycles inside the pool's ring, 11 the
// longest walk of one seed (a maximum, not a sum).  tools/dev/grch38_phase_spread.py reads them per read.
#ifndef UNC_DBG_SEED
#define UNC_DBG_SEED 0
#endif
#if UNC_DBG_SEED
__shared__ unsigned long long s_dbgseed[4];
#endif

struct Tracker {
    uint32_t n, n_lens, max1, max2, status, n_alloc;      // n_alloc: nodes taken from the read's own chunks so far
    float len_sum;
    ClusterVal mm;
};

// SeedTracker's std::set<SeedCluster> (ordered by ref_en_.start descending, evt_en_ descending) as a GRID OF BUCKETS over
// ref_en_.start.  add_seed only ever looks at the clusters whose start lies in [seed start - seed event, seed start]: from
// lower_bound(seed) the reference scans towards smaller starts, takes the best-supported cluster the seed can extend, and stops
// at the first cluster that lies more than `event` rows back (seed_tracker.cpp:169-191: in_range needs r2 - r1 <= e2 - e1 <=
// e2, a cluster further back is never a candidate and ends the scan).  Inside that window the scan's outcome does not depend
// on any order but the set's own tie-break (among equally long candidates the first in set order), so a bucket is an UNORDERED
// array: nodes of NODE_K = 5 clusters (hot key 16 B + cold part 32 B each, 256 B) chained from a per-read table of bucket heads.
// insert = append, erase = move the node's last cluster into the hole.  No directory, nothing to shift, nothing to split.
// Round 4: the seeds of an event are added ONE LANE PER SEED (add_seeds below) -- a lane walks its seed's window by itself, and
// seeds whose windows share no bucket are added in the same round -- where round 3 added them one after the other, the whole
// wavefront gathering one seed's window (lane = bucket x slot) with three to five dependent round trips per seed.
// The bucket width (DevIndex::bucket_shift) is set per index: at least 2^BUCKET_SHIFT_MIN rows and at most 2^15 buckets per
// read (a bucket costs a node of 256 bytes once it is used).  (Rounds 1-2: a sorted array, then a two-level B+-tree.)
constexpr uint32_t NODE_HOT_OFF = 16, NODE_COLD_OFF = 16 + NODE_K * 16;      // (NODE_K = 5 clusters in 256 bytes: unc_dev_types.h)
struct alignas(16) NodeHdr { uint32_t count, next, pad0, pad1; };            // next: node id + 1 (0: end of the chain)
static_assert(NODE_COLD_OFF + NODE_K * 32 <= NODE_BYTES, "node layout");
static_assert(sizeof(ClusterKey) == 16 && sizeof(ClusterCold) == 32, "seed-cluster record layout");
struct PoolView {          // DevPool with its arrays typed as global memory
    gptr_t nodes;
    PoolQueue *q;
    SchedCell *cells;
    uint32_t cap_mask;
};
struct TrackerMem {
    gptr_t sb;             // the read's slot: bucket heads and chunk list live there
    uint32_t off_heads;    // u32 [n_buckets]: first node of the bucket + 1 (0: empty)
    uint32_t off_chunks;   // u32 [max_nodes / CHUNK_NODES + 1]
    uint32_t max_nodes;
    uint32_t n_buckets, shift;
    PoolView pool;         // nodes
};

// a chunk off the pool's ring (lane 0 only; a pop per 768 nodes).  Nothing of what the chunk held is read: no acquire
__device__ __forceinline__ uint32_t pool_pop(const PoolView &P) { return pool_ring_pop(P.q, P.cells, P.cap_mask); }
// the read is over: its chunks go back to the pool.  The release fence: the next owner of a chunk may sit on another XCD, and lines
// of the chunk that are still dirty in THIS XCD's L2 must be in memory before that owner's stores are (a later write-back of them
// would land on top of the new owner's nodes)
__device__ __forceinline__ void tracker_release(Tracker &T, const TrackerMem &M, int lane) {
    const uint32_t n_chunks = (T.n_alloc + CHUNK_NODES - 1) / CHUNK_NODES;
    if (n_chunks) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    for (uint32_t i = (uint32_t)lane; i < n_chunks; i += WAVE)
        pool_ring_push(M.pool.q, M.pool.cells, M.pool.cap_mask, gld<uint32_t>(M.sb, M.off_chunks + (i << 2)));
    T.n_alloc = 0; T.n = 0;
    wave_sync();
}
// a read starts with every bucket empty
__device__ __forceinline__ void tracker_clear_heads(const TrackerMem &M, int lane) {
    const uint32_t n16 = (M.n_buckets + 3) / 4;
    for (uint32_t i = (uint32_t)lane; i < n16; i += WAVE) gst(M.sb, M.off_heads + (i << 4), make_uint4(0u, 0u, 0u, 0u));
    wave_sync();
}

// std::multiset<u32> all_lens_ reduced to what get_final reads: its size and its two largest values.
__device__ __forceinline__ void lens_insert(Tracker &T, uint32_t v) {
    T.n_lens++;
    if (v > T.max1) { T.max2 = T.max1; T.max1 = v; }
    else if (v > T.max2) T.max2 = v;
}
// replace one instance of p by q > p (seed_tracker.cpp:201-203)
__device__ __forceinline__ void lens_replace(Tracker &T, uint32_t p, uint32_t q) {
    if (p == T.max1) { T.max1 = q; }                                   // (q, max2)
    else if (p == T.max2) { if (q > T.max1) { T.max2 = T.max1; T.max1 = q; } else T.max2 = q; }
    else { if (q > T.max1) { T.max2 = T.max1; T.max1 = q; } else if (q > T.max2) T.max2 = q; }
}

__device__ __forceinline__ gptr_t node_ptr(const TrackerMem &M, uint32_t node) { return M.pool.nodes + (size_t)node * NODE_BYTES; }
__device__ __forceinline__ void node_store(const TrackerMem &M, uint32_t node, uint32_t slot, const ClusterKey &k, const ClusterCold &c) {
    const gptr_t p = node_ptr(M, node);
    gst(p, NODE_HOT_OFF + (slot << 4), k);
    gst(p, NODE_COLD_OFF + (slot << 5), c);
}
__device__ __forceinline__ void node_hdr_store(const TrackerMem &M, uint32_t node, uint32_t count, uint32_t next) {
    NodeHdr h; h.count = count; h.next = next; h.pad0 = 0; h.pad1 = 0;
    gst(node_ptr(M, node), 0u, h);
}

// What one lane has found out about its seed's window: the candidates of seed_tracker.cpp:169-191 reduced to the few the outcome
// depends on.  A cluster of the window is ranked by re = (0xFFFF - (r2 - r1)) << 16 | e1: larger = earlier in set order
// (ref_en_.start descending, evt_en_ descending; both differences fit 16 bits because max_events <= 65535).
struct SeedScan {
    uint32_t best_found, best_tl, best_re, best_node, best_sc, best_next;   // best-supported near candidate (sc = slot | node count << 8)
    uint32_t f0_found, f0_tl, f0_node, f0_sc, f0_next;                      // the cluster exactly e2 rows back with evt_en 0, if there is one
    uint32_t f_other;                                                       // another cluster sits exactly e2 rows back (it ends the scan first)
    uint32_t exists;                                                        // an equivalent key (r2, e2) is in the set
    uint32_t lb_have, lb_re;                                                // key at lower_bound(seed): the first in set order not before the seed
    uint32_t ins_node, ins_cnt, ins_next;                                   // a node of the seed's own bucket with room (id + 1)
    uint32_t head0;                                                         // head of the seed's own bucket
    uint32_t walked;                                                        // nodes visited (read by the UNC_DBG_SEED build only)
};

// the window of one seed, walked by ONE lane: buckets from the seed's own downwards, every node of their chains (header + NODE_K
// hot keys = one contiguous piece of the node), all comparisons in the lane's own registers.  The next bucket's head is
// requested before the current chain is walked.
__device__ __forceinline__ void seed_scan(const TrackerMem &M, uint64_t r2, uint32_t e2, uint32_t b_hi, uint32_t b_lo, SeedScan &S) {
    S.best_found = S.best_tl = S.best_re = S.best_node = S.best_sc = S.best_next = 0;
    S.f0_found = S.f0_tl = S.f0_node = S.f0_sc = S.f0_next = 0;
    S.f_other = S.exists = S.lb_have = S.lb_re = 0;
    S.ins_node = S.ins_cnt = S.ins_next = 0;
    S.walked = 0;
    uint32_t node1 = gld<uint32_t>(M.sb, M.off_heads + (b_hi << 2));      // node id + 1
    S.head0 = node1;
    for (uint32_t b = b_hi;;) {
        const uint32_t head_nxt = b > b_lo ? gld<uint32_t>(M.sb, M.off_heads + ((b - 1u) << 2)) : 0u;
        while (node1) {
            const cgptr_t p = node_ptr(M, node1 - 1u);
            const NodeHdr h = gld<NodeHdr>(p, 0u);
            ClusterKey k[NODE_K];
#pragma unroll
            for (uint32_t s = 0; s < NODE_K; ++s) k[s] = gld<ClusterKey>(p, NODE_HOT_OFF + (s << 4));
#pragma unroll
            for (uint32_t s = 0; s < NODE_K; ++s) {
                const uint64_t r1 = k[s].rstart;
                const uint32_t e1 = k[s].evt_en, tl = k[s].total_len;
                // not before the seed in set order, and not further back than e2 rows
                if (s < h.count && r1 <= r2 && !(r1 == r2 && e1 > e2) && (r2 - r1) <= (uint64_t)e2) {
                    const uint32_t dr = (uint32_t)(r2 - r1);
                    const uint32_t re = ((0xFFFFu - dr) << 16) | e1;
                    if (dr == 0u && e1 == e2) S.exists = 1u;
                    if (!S.lb_have || re > S.lb_re) S.lb_re = re;
                    S.lb_have = 1u;
                    const uint32_t sc = s | (h.count << 8);
                    if (dr == e2) {                                                  // r2 - r1 >= e2: a non-candidate here ends the scan
                        if (e1 > 0u) S.f_other = 1u;
                        else { S.f0_found = 1u; S.f0_tl = tl; S.f0_node = node1 - 1u; S.f0_sc = sc; S.f0_next = h.next; }
                    } else if (e1 <= e2) {
                        const uint32_t de = e2 - e1;
                        // near candidates (:178-181): the longest wins, among equally long ones the first in set order
                        if (dr <= de && dr >= de / 12u &&
                            (!S.best_found || tl > S.best_tl || (tl == S.best_tl && re > S.best_re))) {
                            S.best_found = 1u; S.best_tl = tl; S.best_re = re; S.best_node = node1 - 1u; S.best_sc = sc; S.best_next = h.next;
                        }
                    }
                }
            }
            if (b == b_hi && !S.ins_node && h.count < NODE_K) { S.ins_node = node1; S.ins_cnt = h.count; S.ins_next = h.next; }
            node1 = h.next;          // on along the chain
            ++S.walked;
        }
        if (b == b_lo) break;
        --b;
        node1 = head_nxt;
    }
}

// take a cluster out of its node (one lane): the node's last cluster moves into the hole
__device__ __forceinline__ void node_erase(const TrackerMem &M, uint32_t node, uint32_t sc, uint32_t next) {
    const uint32_t slot = sc & 0xFFu, last = (sc >> 8) - 1u;
    if (slot != last) {
        const cgptr_t p = node_ptr(M, node);
        const ClusterKey lk = gld<ClusterKey>(p, NODE_HOT_OFF + (last << 4));
        const ClusterCold lc = gld<ClusterCold>(p, NODE_COLD_OFF + (last << 5));
        node_store(M, node, slot, lk, lc);
    }
    node_hdr_store(M, node, last, next);
}

__device__ __forceinline__ uint32_t lane_get32(uint32_t v, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src); }
__device__ __forceinline__ uint64_t lane_get64(uint64_t v, uint32_t src) {
    return ((uint64_t)lane_get32((uint32_t)(v >> 32), src) << 32) | lane_get32((uint32_t)v, src);
}

// SeedTracker::add_seed (seed_tracker.cpp:157-232) for up to 64 seeds of one event, ONE LANE PER SEED.  A seed reads and changes only the
// clusters whose start lies in its window [r2 - e2, r2], i.e. the buckets b_lo..b_hi: seeds whose bucket intervals are
// disjoint do not see each other, whatever the order they are added in.  So the seeds are added in ROUNDS: a round takes every
// pending seed whose interval no EARLIER pending seed touches (such seeds are pairwise disjoint), each lane scans its
// window, decides and commits on its own -- the dependent memory round trips of a seed are paid once per round, not once
// per seed -- and the seeds that had to wait take the next round, seeing what the earlier ones left: the reference's order
// wherever order matters.  What IS order-dependent across all seeds -- the float sum of the lengths, the two largest lengths and
// the first cluster to reach the record length -- is replayed in seed order from per-lane outcomes afterwards (registers only).
// task = sa_end (40 bits) | event << 40 | length << 56 of lane's seed (lanes >= nt: none).  All of T stays uniform.
static __device__ void add_seeds(Tracker &T, const TrackerMem &M, uint32_t min_map_len, uint64_t task, uint32_t nt, int lane) {
    if (T.status) return;
    const bool have = (uint32_t)lane < nt;
    const uint64_t ref_en = task & ((1ull << 40) - 1ull);
    const uint32_t ref_len = (uint32_t)(task >> 56), e2 = (uint32_t)(task >> 40) & 0xFFFFu;
    const uint64_t r2 = ref_en - ref_len + 1;                     // new_seed.ref_en_.start_ (= ref_st_)
    const uint64_t r_lo = r2 > (uint64_t)e2 ? r2 - (uint64_t)e2 : 0ull;
    const uint32_t b_hi = (uint32_t)(r2 >> M.shift), b_lo = (uint32_t)(r_lo >> M.shift);
    // what the seed did, for the replay: 0 nothing (not reached), 1 joined a cluster, 2 new cluster; the cluster's length before
    // and the cluster after
    uint32_t o_kind = 0, o_prev = 0;
    ClusterVal o_a;
    o_a.ref_st = o_a.rstart = o_a.rend = 0; o_a.evt_st = o_a.evt_en = o_a.total_len = 0;

    uint64_t pend = __ballot(have);
    while (pend) {
#if UNC_DBG_SEED
        if (lane == 0) s_dbgseed[0] += 1ull;
#endif
        // ---- this round's seeds: pending, and no earlier pending seed's buckets overlap theirs
        bool blocked = false;
        for (uint64_t m = pend; m & (m - 1ull); m &= m - 1ull) {              // (the last pending seed blocks nobody)
            const uint32_t i = (uint32_t)__ffsll((unsigned long long)m) - 1u;
            const uint32_t lo_i = lane_get32(b_lo, i), hi_i = lane_get32(b_hi, i);
            if ((uint32_t)lane > i && b_lo <= hi_i && lo_i <= b_hi) blocked = true;
        }
        const bool ready = ((pend >> lane) & 1ull) != 0ull && !blocked;

        // ---- scan, decide, commit (per lane; a new node is taken below, together)
        bool need_node = false;
        int dn = 0;                                                // change of the set's size
        ClusterKey nk; nk.rstart = 0; nk.evt_en = 0; nk.total_len = 0;
        ClusterCold nc; nc.ref_st = 0; nc.rend = 0; nc.evt_st = 0; nc.pad[0] = nc.pad[1] = nc.pad[2] = 0;
        uint32_t head0 = 0;
        if (ready) {
            SeedScan S;
            seed_scan(M, r2, e2, b_hi, b_lo, S);
#if UNC_DBG_SEED
            atomicAdd(&s_dbgseed[1], (unsigned long long)S.walked);
            atomicMax(&s_dbgseed[3], (unsigned long long)S.walked);
#endif
            head0 = S.head0;
            // the scan reaches the clusters e2 rows back after all nearer ones, in order of descending evt_en: any of them with
            // evt_en > 0 is out of range and ends it; the one with evt_en 0 is in range (r2 - r1 = e2 - e1) and is taken when it is longer
            uint32_t mt_found = S.best_found, mt_tl = S.best_tl, mt_re = S.best_re, mt_node = S.best_node, mt_sc = S.best_sc, mt_next = S.best_next;
            if (S.f0_found && !S.f_other && (!S.best_found || S.f0_tl > S.best_tl)) {
                mt_found = 1u; mt_tl = S.f0_tl; mt_re = (0xFFFFu - e2) << 16; mt_node = S.f0_node; mt_sc = S.f0_sc; mt_next = S.f0_next;
            }
            bool want_insert = false;
            if (mt_found) {
                const uint64_t mt_r = r2 - (uint64_t)(0xFFFFu - (mt_re >> 16));
                const ClusterCold mp = gld<ClusterCold>(node_ptr(M, mt_node), NODE_COLD_OFF + ((mt_sc & 0xFFu) << 5));
                ClusterVal a;
                a.ref_st = mp.ref_st; a.rstart = mt_r; a.rend = mp.rend;
                a.evt_st = mp.evt_st; a.evt_en = mt_re & 0xFFFFu; a.total_len = mt_tl;
                // SeedCluster::update, seed_tracker.cpp:56-73 (growth is a u8)
                uint8_t growth = 0;
                if (r2 < a.rend) {
                    if (ref_en > a.rend) { growth = (uint8_t)(ref_en - a.rend); a.rend = ref_en; }
                    a.rstart = r2;
                } else {
                    growth = (uint8_t)ref_len;
                    a.rstart = r2;
                    a.rend = ref_en;
                }
                a.evt_en = e2;
                a.total_len += growth;
                o_kind = 1u; o_prev = mt_tl; o_a = a;
                // erase(loc_match) then insert(hint, a): a's key is now exactly (r2, e2).  The insert collides -- and the cluster is
                // dropped -- when that key is already in the set and is not the matched cluster itself (which then sits at the lower bound)
                nk.rstart = r2; nk.evt_en = e2; nk.total_len = a.total_len;
                nc.ref_st = a.ref_st; nc.rend = a.rend; nc.evt_st = a.evt_st;
                const bool is_lb = S.lb_have && mt_re == S.lb_re;
                if (!is_lb && S.exists) {
                    node_erase(M, mt_node, mt_sc, mt_next);
                    dn = -1;
                } else if ((uint32_t)(mt_r >> M.shift) == b_hi) {
                    node_store(M, mt_node, mt_sc & 0xFFu, nk, nc);          // same bucket: the cluster stays where it is
                } else {
                    node_erase(M, mt_node, mt_sc, mt_next);                 // its start moved into the seed's bucket
                    want_insert = true;      // (the node found for inserting is not the one just shrunk: that one belongs to another bucket)
                }
            } else {
                // new cluster (:218-228): the bookkeeping happens even when the set insert collides
                o_kind = 2u; o_prev = 0u;
                o_a.ref_st = r2; o_a.rstart = r2; o_a.rend = ref_en; o_a.evt_st = e2; o_a.evt_en = e2; o_a.total_len = ref_len;
                if (!S.exists) {
                    nk.rstart = r2; nk.evt_en = e2; nk.total_len = ref_len;
                    nc.ref_st = r2; nc.rend = ref_en; nc.evt_st = e2;
                    want_insert = true;
                    dn = 1;
                }
            }
            // where a new key goes: a free slot of the seed's own bucket, or a fresh node at its head
            if (want_insert) {
                if (S.ins_node) { node_store(M, S.ins_node - 1u, S.ins_cnt, nk, nc); node_hdr_store(M, S.ins_node - 1u, S.ins_cnt + 1u, S.ins_next); }
                else need_node = true;
            }
        }
        // ---- fresh nodes for the lanes that need one: the next ones of the read's newest chunk, then of a chunk popped off the pool's
        // ring (at most one chunk boundary is crossed: 64 < CHUNK_NODES).  Out of allowance or pool dry: the read overflows and is
        // mapped again later, whatever has been committed so far is void.
        const uint64_t nm = __ballot(need_node);
        if (nm) {
            const uint32_t k = (uint32_t)__popcll(nm), a0 = T.n_alloc;
            if (a0 + k > M.max_nodes) { T.status |= UNC_READ_CLUSTER_OVERFLOW; return; }
            const uint32_t c0 = a0 / CHUNK_NODES, c1 = (a0 + k - 1u) / CHUNK_NODES;
            uint32_t ch0 = SCHED_EMPTY, ch1 = SCHED_EMPTY;
#if UNC_DBG_SEED
            const uint64_t dbg_t0 = (uint64_t)clock64();
#endif
            if (lane == 0) {
                if (a0 % CHUNK_NODES == 0u) {
                    ch0 = pool_pop(M.pool);
                    if (ch0 != SCHED_EMPTY) gst(M.sb, M.off_chunks + (c0 << 2), ch0);
                } else ch0 = gld<uint32_t>(M.sb, M.off_chunks + (c0 << 2));
                if (c1 == c0) ch1 = ch0;
                else if (ch0 != SCHED_EMPTY) {
                    ch1 = pool_pop(M.pool);
                    if (ch1 != SCHED_EMPTY) gst(M.sb, M.off_chunks + (c1 << 2), ch1);
                }
            }
            ch0 = lane_get32(ch0, 0u); ch1 = lane_get32(ch1, 0u);
#if UNC_DBG_SEED
            if (lane == 0) s_dbgseed[2] += (uint64_t)clock64() - dbg_t0;
#endif
            if (ch0 == SCHED_EMPTY || ch1 == SCHED_EMPTY) {
                // (a chunk that WAS popped is the read's: n_alloc covers it, so that tracker_release hands it back)
                if (ch0 != SCHED_EMPTY) T.n_alloc = c1 * CHUNK_NODES;
                T.status |= UNC_READ_CLUSTER_OVERFLOW | UNC_READ_POOL_DRY;
                return;
            }
            if (need_node) {
                const uint32_t my = a0 + prefix_popc(nm);
                const uint32_t id = (my / CHUNK_NODES == c0 ? ch0 : ch1) * CHUNK_NODES + my % CHUNK_NODES;
                node_store(M, id, 0u, nk, nc); node_hdr_store(M, id, 1u, head0);
                gst(M.sb, M.off_heads + (b_hi << 2), id + 1u);
            }
            T.n_alloc = a0 + k;
        }
        T.n += (uint32_t)__popcll(__ballot(dn > 0)) - (uint32_t)__popcll(__ballot(dn < 0));
        pend &= ~__ballot(ready);
        wave_sync();          // this round's stores before the next round's loads
    }

    // ---- replay in seed order what depends on it (:193-206, :218-224)
    for (uint32_t j = 0; j < nt; ++j) {
        const uint32_t kind = lane_get32(o_kind, j), prev = lane_get32(o_prev, j), tl = lane_get32(o_a.total_len, j);
        if (kind == 2u) lens_insert(T, tl);
        else if (tl != prev) lens_replace(T, prev, tl);
        if (kind == 2u || tl != prev) {
            T.len_sum = __fadd_rn(T.len_sum, (float)(tl - prev));
            if (tl >= min_map_len && tl > T.mm.total_len) {
                T.mm.ref_st = lane_get64(o_a.ref_st, j); T.mm.rstart = lane_get64(o_a.rstart, j); T.mm.rend = lane_get64(o_a.rend, j);
                T.mm.evt_st = lane_get32(o_a.evt_st, j); T.mm.evt_en = lane_get32(o_a.evt_en, j); T.mm.total_len = tl;
            }
        }
    }
}

}  // namespace unc
