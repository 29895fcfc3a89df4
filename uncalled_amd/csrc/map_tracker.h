// k_map.hip, part 3: SeedTracker (seed_tracker.cpp:56-73,129-232) as a grid of buckets over ref_en_.start.
#pragma once

namespace unc {

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
// array: nodes of NODE_K = 5 clusters (hot key 16 B + cold part 32 B each, 256 B) chained from a per-read table of bucket heads.  A seed
// costs the heads of its window's buckets (one coalesced load), their nodes (one load, one lane per cluster) and a store;
// insert = append, erase = move the node's last cluster into the hole.  No directory, nothing to shift, nothing to split.
// The bucket width (DevIndex::bucket_shift) is set per index: at least 2^12 rows (a window of the default max_events spans at
// most 9 buckets; 12 are gathered at once) and at most 2^15 buckets per read (a bucket costs a node of 256 bytes once it is used).  (Rounds 1-2: a sorted array, then a two-level B+-tree whose
// directory search, leaf load and leaf shift were three to five dependent memory round trips per seed.)
constexpr uint32_t NODE_HOT_OFF = 16, NODE_COLD_OFF = 16 + NODE_K * 16;      // (NODE_K = 5 clusters in 256 bytes: unc_dev_types.h)
constexpr uint32_t NODE_NONE = 0xFFFFFFFFu;
struct alignas(16) NodeHdr { uint32_t count, next, pad0, pad1; };            // next: node id + 1 (0: end of the chain)
static_assert(NODE_COLD_OFF + NODE_K * 32 <= NODE_BYTES && WIN_BUCKETS * NODE_K <= WAVE, "node layout");
static_assert(sizeof(ClusterKey) == 16 && sizeof(ClusterCold) == 32, "seed-cluster record layout");
struct PoolView {          // DevPool with its arrays typed as global memory
    gptr_t nodes;
    SchedQueue *q;
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

// A fresh node for this read: the next one of its newest chunk, or the first of a chunk popped off the pool's ring.
// NODE_NONE when the read has used up its allowance or the pool has run dry (the read then overflows and is mapped again later).
__device__ __forceinline__ uint32_t tracker_new_node(Tracker &T, const TrackerMem &M, int lane) {
    const uint32_t a = T.n_alloc;
    if (a >= M.max_nodes) { T.status |= UNC_READ_CLUSTER_OVERFLOW; return NODE_NONE; }
    uint32_t chunk;
    if (a % CHUNK_NODES == 0) {
        uint32_t c = SCHED_EMPTY;
        if (lane == 0) {
            c = sched_pop(M.pool.q, M.pool.cells, M.pool.cap_mask);
            if (c != SCHED_EMPTY) gst(M.sb, M.off_chunks + ((a / CHUNK_NODES) << 2), c);
        }
        chunk = bcast32(c, 0);
        if (chunk == SCHED_EMPTY) { T.status |= UNC_READ_CLUSTER_OVERFLOW | UNC_READ_POOL_DRY; return NODE_NONE; }
    } else {
        chunk = uniform32(gld<uint32_t>(M.sb, M.off_chunks + ((a / CHUNK_NODES) << 2)));
    }
    T.n_alloc = a + 1;
    return chunk * CHUNK_NODES + a % CHUNK_NODES;
}

// the read is over: its chunks go back to the pool
__device__ __forceinline__ void tracker_release(Tracker &T, const TrackerMem &M, int lane) {
    const uint32_t n_chunks = (T.n_alloc + CHUNK_NODES - 1) / CHUNK_NODES;
    for (uint32_t i = (uint32_t)lane; i < n_chunks; i += WAVE)
        sched_push(M.pool.q, M.pool.cells, M.pool.cap_mask, gld<uint32_t>(M.sb, M.off_chunks + (i << 2)));
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

__device__ __forceinline__ uint64_t wave_max64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, d); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = o > v ? o : v; }
    return v;
}

// one cluster as add_seed carries it between the gather and the commit (all fields uniform)
struct ClusterRef { uint32_t found, node, slot, cnt, next, tl, e; uint64_t r; };
// the lane `src` holds the cluster: its fields to every lane
__device__ __forceinline__ ClusterRef cluster_ref(uint32_t node, uint32_t slot, const NodeHdr &h, const ClusterKey &k, int src) {
    ClusterRef c;
    c.found = 1; c.node = bcast32(node, src); c.slot = bcast32(slot, src); c.cnt = bcast32(h.count, src); c.next = bcast32(h.next, src);
    c.tl = bcast32(k.total_len, src); c.e = bcast32(k.evt_en, src); c.r = bcast64(k.rstart, src);
    return c;
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

// SeedTracker::add_seed, seed_tracker.cpp:157-232 (wave-cooperative; all arguments uniform; stores by lane 0)
static __device__ void add_seed(Tracker &T, const TrackerMem &M, uint32_t min_map_len, uint64_t ref_en, uint32_t ref_len, uint32_t evt, int lane) {
    if (T.status) return;
    const uint64_t r2 = ref_en - ref_len + 1;   // new_seed.ref_en_.start_ (= ref_st_)
    const uint32_t e2 = evt;

    // ---- gather: the buckets that hold starts in [r2 - e2, r2], highest first; lane = (bucket j, slot s) of the bucket's current node
    const uint64_t r_lo = r2 > (uint64_t)e2 ? r2 - (uint64_t)e2 : 0ull;
    const uint32_t b_hi = (uint32_t)(r2 >> M.shift), b_lo = (uint32_t)(r_lo >> M.shift);
    const uint32_t nb = b_hi - b_lo + 1u;                         // <= WIN_BUCKETS for max_events < 2^15 (the default is 30 000)
    const uint32_t j = (uint32_t)lane / NODE_K, s = (uint32_t)lane % NODE_K;
    uint32_t head0 = 0;                                           // head of the seed's own bucket

    ClusterRef best; best.found = 0; best.node = best.slot = best.cnt = best.next = best.tl = best.e = 0; best.r = 0;   // best-supported candidate among the near ones
    ClusterRef f0 = best;                                         // the cluster exactly e2 rows back with evt_en 0, if there is one
    bool f_other = false;                                         // ... and whether another cluster sits exactly e2 rows back (it ends the scan first)
    bool exists = false;                                          // an equivalent key (r2, e2) is in the set
    uint64_t lb_r = 0; uint32_t lb_e = 0; bool lb_have = false;   // key at lower_bound(seed): the first in set order not before the seed
    uint32_t ins_node = 0, ins_cnt = 0, ins_next = 0;             // a node of the seed's own bucket with room (id + 1)
    for (uint32_t jb = 0; jb < nb; jb += WIN_BUCKETS) {
    const bool active = j < WIN_BUCKETS && jb + j < nb;
    uint32_t node1 = active ? gld<uint32_t>(M.sb, M.off_heads + ((b_hi - jb - j) << 2)) : 0u;      // node id + 1
    if (jb == 0) head0 = bcast32(node1, 0);
    while (__any(node1 != 0u)) {
        NodeHdr h; h.count = 0; h.next = 0; h.pad0 = h.pad1 = 0;
        ClusterKey k; k.rstart = 0; k.evt_en = 0; k.total_len = 0;
        if (node1) {
            const cgptr_t p = node_ptr(M, node1 - 1u);
            h = gld<NodeHdr>(p, 0u);
            k = gld<ClusterKey>(p, NODE_HOT_OFF + (s << 4));
        }
        const bool valid = node1 != 0u && s < h.count;
        const uint64_t r1 = k.rstart;
        const uint32_t e1 = k.evt_en, tl = k.total_len;
        // not before the seed in set order, and not further back than e2 rows
        const bool in_win = valid && r1 <= r2 && !(r1 == r2 && e1 > e2) && (r2 - r1) <= (uint64_t)e2;
        const uint64_t dr = r2 - r1, de = (uint64_t)e2 - (uint64_t)e1;
        const bool in_range = in_win && e1 <= e2 && dr <= de && dr >= de / 12;       // :178-181
        const bool farE = in_win && dr == (uint64_t)e2;                              // r2 - r1 >= e2: a non-candidate here ends the scan
        if (__any(valid && r1 == r2 && e1 == e2)) exists = true;
        // a node of the seed's own bucket with a free slot
        if (!ins_node) {
            const uint64_t m = __ballot(jb == 0 && j == 0 && s == 0 && node1 != 0u && h.count < NODE_K);
            if (m) { ins_node = bcast32(node1, 0); ins_cnt = bcast32(h.count, 0); ins_next = bcast32(h.next, 0); }
        }
        // lower bound: the largest (r1, e1) in the window
        {
            const uint64_t mr = wave_max64(in_win ? r1 + 1ull : 0ull);               // (+1: a start of 0 still counts)
            if (mr) {
                const uint32_t me = wave_max32(in_win && r1 + 1ull == mr ? e1 + 1u : 0u);
                if (!lb_have || mr - 1ull > lb_r || (mr - 1ull == lb_r && me - 1u > lb_e)) { lb_r = mr - 1ull; lb_e = me - 1u; }
                lb_have = true;
            }
        }
        // near candidates: the longest wins, among equally long ones the first in set order = the largest (r1, e1)
        {
            const bool cand = in_range && !farE;
            const uint64_t cm = __ballot(cand);
            if (cm) {
                const uint32_t m_tl = wave_max32(cand ? tl + 1u : 0u) - 1u;
                const bool s1 = cand && tl == m_tl;
                const uint64_t m_r = wave_max64(s1 ? r1 + 1ull : 0ull) - 1ull;
                const bool s2 = s1 && r1 == m_r;
                const uint32_t m_e = wave_max32(s2 ? e1 + 1u : 0u) - 1u;
                const uint64_t wm = __ballot(s2 && e1 == m_e);
                const int src = __ffsll((unsigned long long)wm) - 1;
                const bool better = !best.found || m_tl > best.tl || (m_tl == best.tl && (m_r > best.r || (m_r == best.r && m_e > best.e)));
                if (better) best = cluster_ref(node1 - 1u, s, h, k, src);
            }
        }
        // clusters exactly e2 rows back
        {
            if (__any(farE && e1 > 0u)) f_other = true;
            const uint64_t fm = __ballot(farE && e1 == 0u);
            if (fm) f0 = cluster_ref(node1 - 1u, s, h, k, __ffsll((unsigned long long)fm) - 1);
        }
        node1 = node1 ? h.next : 0u;          // on along the chains
    }
    }
    // the scan reaches the clusters e2 rows back after all nearer ones, in order of descending evt_en: any of them with evt_en > 0
    // is out of range and ends it; the one with evt_en 0 is in range (r2 - r1 = e2 - e1) and is taken when it is longer
    ClusterRef mt = best;
    if (f0.found && !f_other && (!best.found || f0.tl > best.tl)) mt = f0;

    // where a new key goes: a free slot of the seed's own bucket, or a fresh node at its head
    auto insert_key = [&](const ClusterKey &nk, const ClusterCold &nc) -> bool {
        if (ins_node) {
            if (lane == 0) { node_store(M, ins_node - 1u, ins_cnt, nk, nc); node_hdr_store(M, ins_node - 1u, ins_cnt + 1u, ins_next); }
        } else {
            const uint32_t id = tracker_new_node(T, M, lane);
            if (id == NODE_NONE) return false;
            if (lane == 0) {
                node_store(M, id, 0u, nk, nc); node_hdr_store(M, id, 1u, head0);
                gst(M.sb, M.off_heads + (b_hi << 2), id + 1u);
            }
        }
        return true;
    };
    // take the cluster c out of its node: the node's last cluster moves into the hole
    auto erase_ref = [&](const ClusterRef &c) {
        const uint32_t last = c.cnt - 1u;
        if (c.slot != last) {
            const cgptr_t p = node_ptr(M, c.node);
            const ClusterKey lk = gld<ClusterKey>(p, NODE_HOT_OFF + (last << 4));
            const ClusterCold lc = gld<ClusterCold>(p, NODE_COLD_OFF + (last << 5));
            wave_sync();
            if (lane == 0) node_store(M, c.node, c.slot, lk, lc);
        }
        if (lane == 0) node_hdr_store(M, c.node, last, c.next);
    };

    if (mt.found) {
        const ClusterCold mp = gld<ClusterCold>(node_ptr(M, mt.node), NODE_COLD_OFF + (mt.slot << 5));
        ClusterVal a;
        a.ref_st = mp.ref_st; a.rstart = mt.r; a.rend = mp.rend;
        a.evt_st = mp.evt_st; a.evt_en = mt.e; a.total_len = mt.tl;
        const uint32_t prev_len = a.total_len;
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
        if (a.total_len != prev_len) {
            T.len_sum = __fadd_rn(T.len_sum, (float)(a.total_len - prev_len));
            lens_replace(T, prev_len, a.total_len);
            if (a.total_len >= min_map_len && a.total_len > T.mm.total_len) T.mm = a;
        }
        // erase(loc_match) then insert(hint, a): a's key is now exactly (r2, e2).  The insert collides -- and the cluster is
        // dropped -- when that key is already in the set and is not the matched cluster itself (which then sits at the lower bound)
        ClusterKey nk; nk.rstart = r2; nk.evt_en = e2; nk.total_len = a.total_len;
        ClusterCold nc; nc.ref_st = a.ref_st; nc.rend = a.rend; nc.evt_st = a.evt_st; nc.pad[0] = nc.pad[1] = nc.pad[2] = 0;
        const bool is_lb = lb_have && mt.r == lb_r && mt.e == lb_e;
        wave_sync();
        if (!is_lb && exists) {
            erase_ref(mt);
            T.n--;
        } else if ((uint32_t)(mt.r >> M.shift) == b_hi) {
            if (lane == 0) node_store(M, mt.node, mt.slot, nk, nc);      // same bucket: the cluster stays where it is
        } else {
            erase_ref(mt);                                               // its start moved into the seed's bucket
            wave_sync();
            // (the node found for inserting is not the one just shrunk: that one belongs to another bucket)
            if (!insert_key(nk, nc)) return;        // (status set by tracker_new_node)
        }
        wave_sync();
    } else {
        // new cluster (:218-228): the bookkeeping happens even when the set insert collides
        lens_insert(T, ref_len);
        T.len_sum = __fadd_rn(T.len_sum, (float)ref_len);
        if (ref_len >= min_map_len && ref_len > T.mm.total_len) {
            T.mm.ref_st = r2; T.mm.rstart = r2; T.mm.rend = ref_en;
            T.mm.evt_st = e2; T.mm.evt_en = e2; T.mm.total_len = ref_len;
        }
        if (!exists) {
            ClusterKey nk; nk.rstart = r2; nk.evt_en = e2; nk.total_len = ref_len;
            ClusterCold nc; nc.ref_st = r2; nc.rend = ref_en; nc.evt_st = e2; nc.pad[0] = nc.pad[1] = nc.pad[2] = 0;
            wave_sync();
            if (!insert_key(nk, nc)) return;        // (status set by tracker_new_node)
            T.n++;
            wave_sync();
        }
    }
}

}  // namespace unc
