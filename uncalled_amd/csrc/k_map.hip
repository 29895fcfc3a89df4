// k_map: the per-read path-forest search, one WAVEFRONT per read, persistent over a read queue.
//
// Replaces Mapper::map_next (mapper.cpp:433-663) + PathBuffer::make_child/make_source (751-807) +
// Mapper::update_seeds (665-700) + SeedTracker::add_seed/get_final (seed_tracker.cpp:129-232) for a
// whole batch.  One event of one read is processed as a sequence of wave-cooperative phases, each an
// out-of-line function:
//
//   P  1024 match log-probs of the normalised event (pore_model.hpp:163-165) -> LDS
//   E  parents, 64 per pass in the reference's visiting order: thresholds -> candidate (parent,base)
//      pairs compacted through LDS -> FM get_neighbor with every lane busy -> child slots by prefix
//      sum (honouring the max_paths cut-off) -> one lane per child writes a 32-byte record and files
//      its 64-bit key into the stream of its class (map_sort.h)
//   S  narrow keys: repair + merge of the streams, the last merge walked in place in LDS;
//      wide keys (human-sized references) and small events: bitonic network
//   W  walk in sorted order: duplicate-range pruning, per-k-mer gap sources from a segmented
//      prefix-max, survivors -> next parent list, seed-valid survivors -> seed list
//   F  full-range sources for k-mers not covered (sources_added_ bitmap in LDS)
//   T  SA look-ups for all seeds in parallel, then the SeedTracker update in the reference's order
//      on the bucket grid (map_tracker.h)
//   G  confidence test (get_final / check_map_conf) -> SUCCESS, or next event
//
// Integer/range results are bit-exact with the reference; float expressions are written one IEEE
// operation at a time (compile with -ffp-contract=off; the reference is built without FMA).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fm_dev.h"
#include "unc_dev_types.h"
#include "wave_prims.h"


#include "map_lds.h"
#include "map_sched.h"
#include "map_tracker.h"
#include "map_sort.h"
#include "map_sort_wide.h"

namespace unc {

__device__ __forceinline__ uint32_t float_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// PathBuffer::make_child, mapper.cpp:775-807.  The prob sums are not copied (see PathRec): a child needs its parent's
// newest sum `last` = prob_sums_[length_] and, once the window is full, `subc` = the parent's prob_sums_[1], which the
// parent lane has already worked out for all of its children together with the slid k-mer history `hist`.
struct ChildHdr { uint32_t moves, meta; float seed_prob, last; uint64_t hist; };
__device__ __forceinline__ ChildHdr make_child(uint32_t pmoves, uint32_t pmeta, float last, float subc, uint64_t hist, uint64_t s, uint64_t e,
                                               uint32_t kmer, float prob, uint32_t move, uint32_t p_seed_len, float p_min_seed_prob,
                                               float p_max_stay, uint32_t child_idx, uint32_t key_len_bits, SortKey &key) {
    const uint32_t PATH_MASK = (1u << SEED_LEN) - 1u, PATH_TAIL_MOVE = 1u << (SEED_LEN - 1);
    const uint32_t plen = (pmeta >> META_LEN_SHIFT) & 31u, pstay = (pmeta >> META_STAY_SHIFT) & 255u;
    const uint32_t stay = 1u - move;
    const bool full = plen == (uint32_t)SEED_LEN;
    const uint32_t len = plen + (full ? 0u : 1u);
    uint32_t moves = ((pmoves << 1) | move) & PATH_MASK;
    const uint32_t cstay = (pstay + stay) * stay;
    ChildHdr c;
    c.last = __fadd_rn(last, prob);
    // full window: (appended - prob_sums_[1]) / seed_len, else appended / len.  One IEEE division serves both cases.
    const float num = full ? __fsub_rn(c.last, subc) : c.last;
    c.seed_prob = __fdiv_rn(num, (float)(full ? (uint32_t)SEED_LEN : len));
    if (full) moves |= PATH_TAIL_MOVE;
    c.moves = moves;
    c.meta = kmer | (len << META_LEN_SHIFT) | (cstay << META_STAY_SHIFT) | (pmeta & META_SA_CHECKED) |
             ((!full && len == (uint32_t)SEED_LEN) ? META_FIRST_FULL : 0u);
    // a move shifts its base into the history (the k-mer's newest base)
    c.hist = move ? (hist & 0x3FFull) | ((((hist >> 10) << 2) | (kmer & 3u)) << 10) : hist;
    // is_seed_valid(path_ended = false), mapper.cpp:842-855, known at creation time
    const uint32_t move_count = (uint32_t)__popc(moves);
    const bool seed_ok = len == p_seed_len && c.seed_prob >= p_min_seed_prob && s == e && (moves & 1u) == 1u &&
                         (float)(len - move_count) <= p_max_stay;      // p_max_stay = max_stay_frac * seed_len
    key.a = key_len_bits ? ((((s << key_len_bits) | (e - s)) << 16) | child_idx) : ((s << KEY_LEN_BITS) | (e - s));
    key.b = ((uint64_t)float_orderable(c.seed_prob) << 32) | ((uint64_t)child_idx << 16) | (move_count << KEYB_MOVES_SHIFT) |
            (seed_ok ? KEYB_SEED_FLAG : 0u) | kmer;
    return c;
}

// PathBuffer::make_source, mapper.cpp:751-772: prob_sums_ = {0, prob}: last = prob, sub = 0, and the window's oldest event is
// this one, with the source's k-mer.  NARROW: rows are 32-bit (start, end), else start << 30 | length - 1.
template <bool NARROW>
__device__ __forceinline__ void write_source(gptr_t buf, uint32_t idx, uint64_t s, uint64_t e, uint32_t kmer, float prob) {
    const uint32_t o = idx << PATH_SHIFT;
    const uint64_t r = NARROW ? (s | (e << 32)) : ((s << KEY_LEN_BITS) | (e - s));
    gst(buf, o, make_uint4((uint32_t)r, (uint32_t)(r >> 32), 1u, kmer | (1u << META_LEN_SHIFT)));
    gst(buf, o + 16u, make_uint4(__float_as_uint(prob), 0u, kmer, 0u));
}


#ifndef UNC_LB
#define UNC_LB 4     // wavefronts per SIMD: 128 VGPRs and under 10 KB of LDS each -> 16 per CU (12 -> 16: -12.6 % on 50 k E. coli reads)
#endif

// ---- One event = a sequence of PHASES, each an out-of-line function.  Inlined into one kernel body, the phases kept every
// loop-invariant address and parameter of every other phase alive in scalar registers: a fifth of the instructions the
// hot loops issued were moves between SGPRs and the lanes of spill VGPRs (round 3, from the disassembly).  A phase function
// reads what it needs from the kernel's argument block (constant address space, scalar loads) and from WaveCtx when it is
// entered, and nothing else is live inside it.
typedef const UNC_AS_CONST MapArgs *kargs_t;

// What the phases hand to each other: uniform scalars in LDS.
struct WaveCtx {
    uint32_t event_i, n_parents, cur, n_surv_par;   // the read: next event, the parent list and which buffer holds it (SlotState)
    uint32_t tstatus;                                // UNC_READ_* bits
    uint32_t nchild, n_seedp;                        // this event: children made, seed paths listed
    uint32_t mixed;                                  // narrow keys: the walk met two children with the SAME range and DIFFERENT k-mers (see phase_S)
    uint32_t kl;                                     // key mode of this event's sorted keys (0: 128-bit)
    uint32_t scnt[6];                                // narrow keys: keys filed in each run
    uint32_t n_surv, n_src;                          // the walk: survivors, gap sources
    uint32_t conf;                                   // the confidence test passed
    uint32_t par_unsorted;                           // narrow keys: the surviving parents were not in ascending range order (never expected)
    uint32_t walked;                                 // phase S walked the keys as it merged them: no separate walk
    uint32_t notes;                                  // UNC_NOTE_* of the read so far
    uint32_t flags_save[NKMER / 32];                 // narrow keys: sources_added_ as the walk found it (restored when the walk is done again)
};
__shared__ WaveCtx s_w;
__shared__ Tracker s_T;       // SeedTracker's scalars between the events that touch them
__shared__ uint64_t s_cyc[12];   // PROF: shader-clock cycles per phase (lane 0)

__device__ __forceinline__ uint32_t ctx_get(const uint32_t &f) { return uniform32(f); }
#define CTX_SET(field, v) do { if (lane == 0) s_w.field = (v); } while (0)

// UNC_PROF_SORT (dev build, tools/dev/build_variants.py sortprof="-DUNC_PROF_SORT=1"): the four counters that normally split
// phase E (8..11) split phase S instead -- 8 repair of the move streams, 9 sort of the unsorted stream, 10 merge of moves and
// unsorted, 11 last merge (with its walk, or through memory) -- and phase E's parts go to counter 1.  Read the profiling pass of
// tools/dev/ab_libs.py with that in mind (the names it prints for 8..11 are phase E's); counter 2 then holds the rest of S.
#ifndef UNC_PROF_SORT
#define UNC_PROF_SORT 0
#endif
template <bool PROF> struct PhaseClock {
    uint64_t tk;
    __device__ __forceinline__ PhaseClock() : tk(PROF ? (uint64_t)clock64() : 0ull) {}
    __device__ __forceinline__ void reset() { if constexpr (PROF) tk = (uint64_t)clock64(); }
    __device__ __forceinline__ void end(int i, int lane) {
        if constexpr (PROF) {
            if ((UNC_PROF_SORT || UNC_DBG_SEED) && i >= 8) i = 1;
            const uint64_t tn = (uint64_t)clock64();
            if (lane == 0) s_cyc[i] += tn - tk;
            tk = tn;
        }
    }
};
#if UNC_PROF_SORT
struct SortClock {          // (counts in both instantiations of the dev build; only the profiling pass reads s_cyc)
    uint64_t tk;
    __device__ __forceinline__ SortClock() : tk((uint64_t)clock64()) {}
    __device__ __forceinline__ void end(int i, int lane) {
        const uint64_t tn = (uint64_t)clock64();
        if (lane == 0) { s_cyc[i] += tn - tk; s_cyc[2] -= tn - tk; }       // (the caller adds the whole of S to counter 2 afterwards)
        tk = tn;
    }
};
#else
struct SortClock { __device__ __forceinline__ void end(int, int) {} };
#endif

__device__ __forceinline__ TrackerMem tracker_mem(kargs_t A, gptr_t sb) {
    TrackerMem M;
    M.sb = sb; M.off_heads = A->sc.off_cl_dir; M.off_chunks = A->sc.off_cl_chunks;
    M.max_nodes = A->sc.max_clusters / 4 ? A->sc.max_clusters / 4 : 1u;
    M.n_buckets = A->ix.n_buckets; M.shift = A->ix.bucket_shift;
    M.pool.nodes = (gptr_t)A->pool.nodes;
    M.pool.q = A->pool.q; M.pool.cells = A->pool.cells; M.pool.cap_mask = A->pool.cap_mask;
    return M;
}

// ---------------- P: match log-probs of the normalised event (pore_model.hpp:163-165) -> s_probs ----------------
static __device__ __noinline__ void phase_P(kargs_t A_, float level, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const UNC_AS_GLOBAL float4 *const model4 = (const UNC_AS_GLOBAL float4 *)A->ix.model4;
#pragma unroll 4
    for (int j = 0; j < NKMER / WAVE; ++j) {
        const uint32_t k = (uint32_t)j * WAVE + (uint32_t)lane;
        const float4 row = g_load(model4 + k);          // one 16-byte row per k-mer: a third of the load instructions
        const float mu = row.x, v2 = row.y, ld = row.z;
        const float d = __fsub_rn(level, mu);
        const double q = -((double)d * (double)d) / (double)v2;
        s_probs[k] = (float)(q - (double)ld);
    }
    wave_sync();
}

// FM index as the phases see it (global-memory pointers; fm_dev.h takes any type with these members)
struct FmView {
    const UNC_AS_GLOBAL uint32_t *bwt;
    const UNC_AS_GLOBAL uint32_t *fm32;
    const UNC_AS_GLOBAL uint64_t *sa, *sa_dense;
    const UNC_AS_CONST uint64_t *L2;
    uint64_t primary, seq_len;
};
__device__ __forceinline__ FmView fm_view(kargs_t A) {
    FmView v;
    v.bwt = (const UNC_AS_GLOBAL uint32_t *)A->ix.bwt; v.fm32 = (const UNC_AS_GLOBAL uint32_t *)A->ix.fm32;
    v.sa = (const UNC_AS_GLOBAL uint64_t *)A->ix.sa; v.sa_dense = (const UNC_AS_GLOBAL uint64_t *)A->ix.sa_dense;
    v.L2 = A->ix.L2; v.primary = A->ix.primary; v.seq_len = A->ix.seq_len;
    return v;
}

// ---------------- E: extend parents ----------------
// parents, 64 per pass in the reference's visiting order: thresholds -> candidate (parent, base) pairs compacted through LDS
// -> FM get_neighbor with every lane busy -> child slots by prefix sum (honouring the max_paths cut-off) -> one lane per
// child writes a 64-byte record (the parent's staged in LDS: no global read), its sort key and its info word.
// Returns this lane's share of the event's get_neighbor count.
template <bool PROF, bool NARROW>
static __device__ __noinline__ uint32_t phase_E(kargs_t A_, gptr_t sb_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    // NARROW: the reference has fewer than 2^32 rows (the host sets key_len_bits only then): rows are 32-bit, FM steps go through
    // the 32-bit rank table (fm32_get_neighbor)
    using Row = std::conditional_t<NARROW, uint32_t, uint64_t>;
    constexpr int RES_BITS = 30;                       // wide rows: packed FM result start << 30 | row count (0 = empty range)
    uint64_t *const s_res = s_e;                      // [parent lane << 2 | base]
    Row *const s_pstart = reinterpret_cast<Row *>(s_e + CAND_MAX), *const s_pend = s_pstart + WAVE;
    uint64_t *const s_phist = reinterpret_cast<uint64_t *>(s_pend + WAVE);          // the parents' k-mer history, slid for their children
    float *const s_plast = reinterpret_cast<float *>(s_phist + WAVE), *const s_psubc = s_plast + WAVE;   // prob_sums_[length_] / the children's prob_sums_[0]
    uint32_t *const s_pmoves = reinterpret_cast<uint32_t *>(s_psubc + WAVE), *const s_pmeta = s_pmoves + WAVE;
    uint16_t *const s_cdesc = reinterpret_cast<uint16_t *>(s_pmeta + WAVE);         // per child: parent lane | type << 6 | child of a source << 9
    uint16_t *const s_ckpos = s_cdesc + CHILD_MAX;                                   // narrow keys: a child's position in its run
    uint16_t *const s_cand = s_cdesc;                                                // (dead before the descriptors are written)
    static_assert(NARROW ? (CAND_MAX * 8 + 2 * WAVE * 4 + WAVE * 8 + 4 * WAVE * 4 + 2 * CHILD_MAX * 2 <= sizeof(s_e))
                         : (CAND_MAX * 8 + 2 * WAVE * 8 + WAVE * 8 + 4 * WAVE * 4 + CHILD_MAX * 2 <= sizeof(s_e)), "phase E staging");
    static_assert(CAND_MAX * 2 <= CHILD_MAX * 2 * 2, "candidate list inside the descriptor area");

    const FmView ix = fm_view(A);
    const uint32_t max_paths = A->sc.max_paths, max_seed_paths = A->sc.max_seed_paths;
    const uint32_t event_i = ctx_get(s_w.event_i), n_parents = ctx_get(s_w.n_parents), cur = ctx_get(s_w.cur), n_surv_par = ctx_get(s_w.n_surv_par);
    const float thr_lane = A->ix.thresholds[lane];   // lane l keeps prob_threshes_[l]
    const uint32_t klb = NARROW ? A->ix.key_len_bits : 0u;
    const uint32_t p_max_consec_stay = A->P.max_consec_stay, p_seed_len = A->P.seed_len, p_max_rep_copy = A->P.max_rep_copy, p_min_rep_len = A->P.min_rep_len;
    const float p_min_seed_prob = A->P.min_seed_prob, p_max_stay = __fmul_rn(A->P.max_stay_frac, (float)A->P.seed_len);
    // byte offsets (inside the slot) of this event's parent / child records and parent-order list
    const uint32_t par_off = A->sc.off_paths + cur * (max_paths << PATH_SHIFT), chd_off = A->sc.off_paths + (cur ^ 1u) * (max_paths << PATH_SHIFT);
    const uint32_t pord_off = A->sc.off_order + cur * (max_paths << 2);
    const uint32_t seedp_off = A->sc.off_seedp, ukeys_off = A->sc.off_keys, str_off = A->sc.off_streams, info_off = A->sc.off_info;
    const uint32_t run_bytes = max_paths << 3;
    // a full-window parent's oldest event is 22 back from this one: its level, for the sum that drops out of the window
    const float level_old = event_i >= (uint32_t)SEED_LEN ? gld<float>(sb, A->sc.off_levels + (((event_i - (uint32_t)SEED_LEN) & (LEVEL_RING - 1u)) << 2)) : 0.0f;
    const UNC_AS_GLOBAL float4 *const model4 = (const UNC_AS_GLOBAL float4 *)A->ix.model4;

    PhaseClock<PROF> clk;
    uint32_t c_nbr = 0;
    uint32_t nchild = 0, n_seedp = 0;
    // narrow keys: keys filed so far in each run (stays / moves by base of the sorted survivors; children of sources)
    uint32_t scnt0 = 0, scnt1 = 0, scnt2 = 0, scnt3 = 0, scnt4 = 0, scntx = 0;
    uint64_t par_carry = 0;
    bool par_bad = false;
    uint64_t wm0 = 0, wm1 = 0, wm2 = 0, wm3 = 0, wm4 = 0;      // wide keys: which parents of the pass have a fitting child in each run ...
    uint32_t ws0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, ws4 = 0, wsx = 0, w_src_cpos0 = 0;   // ... the runs' fill before the pass, where the sources' children begin
    // parent index list and record headers are fetched one / two passes ahead of their use
    uint32_t phys_cur = (uint32_t)lane < n_parents ? gld<uint32_t>(sb, pord_off + ((uint32_t)lane << 2)) : 0u;
    uint32_t phys_nxt = (uint32_t)lane + WAVE < n_parents ? gld<uint32_t>(sb, pord_off + (((uint32_t)lane + WAVE) << 2)) : 0u;
    using Q4 = std::conditional_t<NARROW, u32x4_t, uint4>;     // (one register tuple for mem_retire; the wide instantiation as it was)
    Q4 q0c = Q4{1u, 0u, 1u, 0u}, q1c = Q4{0u, 0u, 0u, 0u};
    if ((uint32_t)lane < n_parents) {
        q0c = gld<Q4>(sb, par_off + (phys_cur << PATH_SHIFT)); q1c = gld<Q4>(sb, par_off + (phys_cur << PATH_SHIFT) + 16u);
    }
    // (waited for HERE, once per event, and in every pass right behind the FM look-ups' wait -- where nothing else is in flight -- so
    // that the top of a pass does not wait for the previous pass's stores to be acknowledged: wave_prims.h, mem_retire.  Narrow keys
    // only: the wide instantiation sits at its register limit, and the same changes cost it 7 % on GRCh38, r05_ab_grch38_3.log)
    if constexpr (NARROW) { mem_retire(phys_nxt); mem_retire(q0c); mem_retire(q1c); }
    for (uint32_t base = 0; base < n_parents && nchild < max_paths; base += WAVE) {
        const uint32_t pi = base + (uint32_t)lane;
        const bool have = pi < n_parents;
        const Q4 q0 = q0c, q1 = q1c;
        phys_cur = phys_nxt;
        if (pi + WAVE < n_parents) {
            q0c = gld<Q4>(sb, par_off + (phys_nxt << PATH_SHIFT)); q1c = gld<Q4>(sb, par_off + (phys_nxt << PATH_SHIFT) + 16u);
        }
        if (pi + 2 * WAVE < n_parents) phys_nxt = gld<uint32_t>(sb, pord_off + ((pi + 2 * WAVE) << 2));
        uint32_t pmoves = 0, pmeta = 0;
        Row pstart = 1, pend = 1;
        float plast = 0.0f, psub = 0.0f;
        uint64_t phist = 0;
        if (have) {
            if constexpr (NARROW) { pstart = q0.x; pend = q0.y; }
            else { const uint64_t r = ((uint64_t)q0.y << 32) | q0.x; pstart = r >> KEY_LEN_BITS; pend = pstart + (r & KEY_LEN_MASK); }
            pmoves = q0.z; pmeta = q0.w;
            plast = __uint_as_float(q1.x); psub = __uint_as_float(q1.y); phist = ((uint64_t)q1.w << 32) | q1.z;
        }
        // A full-window parent hands its children prob_sums_[1] = prob_sums_[0] + the match probability of the window's oldest
        // event with the k-mer the lineage had then (the model row is requested here, the division runs behind the FM round trip)
        const uint32_t plen = (pmeta >> META_LEN_SHIFT) & 31u;
        const bool pfull = plen == (uint32_t)SEED_LEN;
        const uint32_t okmer = (uint32_t)phist & KMASK;
        // (requested by every lane -- a load inside `if (pfull)` is waited for on the spot -- but the lanes that do not need a row all
        // ask for row 0: one address, one request)
        f32x4_t orow = {0.f, 1.f, 0.f, 0.f};
        if constexpr (NARROW) orow = g_load(reinterpret_cast<const UNC_AS_GLOBAL f32x4_t *>(model4) + (pfull ? okmer : 0u));
        else if (pfull) orow = g_load(reinterpret_cast<const UNC_AS_GLOBAL f32x4_t *>(model4) + okmer);      // (wide keys: as rounds 1-4, see the note above the loop)
        const Row plen_fm = pend - pstart + 1;
        {
            // the merge of the children's keys relies on the survivors being in ascending (start, length) order: checked here
            uint64_t pk = ~0ull;
            if (pi < n_surv_par) {
                if constexpr (NARROW) pk = ((uint64_t)pstart << 32) | (uint32_t)(pend - pstart);
                else pk = ((uint64_t)pstart << KEY_LEN_BITS) | (uint64_t)(pend - pstart);
            }
            uint64_t pp = (uint64_t)__shfl_up((unsigned long long)pk, 1);
            if (lane == 0) pp = par_carry;
            if (__any(pi < n_surv_par && !(pk > pp))) par_bad = true;
            par_carry = bcast64(pk, WAVE - 1);
        }
        int thr_bin;                                                         // get_fm_bin = clzll(length), :161-163
        if constexpr (NARROW) thr_bin = 32 + __clz((int)plen_fm); else thr_bin = __clzll((long long)plen_fm);
        const float thr = __shfl(thr_lane, thr_bin);                          // get_prob_thresh, :165-167
        const uint32_t kmer = pmeta & META_KMER_MASK;
        const uint32_t stays = (pmeta >> META_STAY_SHIFT) & 255u;
        const bool stay_ok = have && stays < p_max_consec_stay && s_probs[kmer] >= thr;
        // kmer_neighbor (bp.hpp:105-108): the four successors ((kmer << 2) & KMASK) | b are neighbours in the table
        const float4 np = *reinterpret_cast<const float4 *>(&s_probs[(kmer << 2) & KMASK]);
        uint32_t mask = (!(np.x < thr) ? 1u : 0u) | (!(np.y < thr) ? 2u : 0u) | (!(np.z < thr) ? 4u : 0u) | (!(np.w < thr) ? 8u : 0u);
        if (!have) mask = 0;
        s_pstart[lane] = pstart; s_pend[lane] = pend; s_pmoves[lane] = pmoves; s_pmeta[lane] = pmeta; s_plast[lane] = plast;
        const uint32_t ncand = (uint32_t)__popc(mask);
        uint32_t ctot;
        const uint32_t coff = excl_sum_bits<3>(ncand, &ctot);
        {
            uint32_t w = coff;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if (mask & (1u << b)) s_cand[w++] = (uint16_t)(((uint32_t)lane << 2) | b);
        }
        wave_sync();
        clk.end(8, lane);
        // FM look-ups, every lane busy
        for (uint32_t c0 = 0; c0 < ctot; c0 += WAVE) {
            const uint32_t ci = c0 + (uint32_t)lane;
            if (ci < ctot) {
                const uint32_t cd = s_cand[ci];
                if constexpr (NARROW) {
                    uint32_t ns, ne;
                    fm32_get_neighbor(ix, s_pstart[cd >> 2], s_pend[cd >> 2], cd & 3u, &ns, &ne);
                    s_res[cd] = ns <= ne ? ((uint64_t)(ne - ns + 1u) << 32) | ns : 0ull;      // row count | first row
                } else {
                    uint64_t ns, ne;
                    fm_get_neighbor(ix, s_pstart[cd >> 2], s_pend[cd >> 2], cd & 3u, &ns, &ne);
                    s_res[cd] = ns <= ne ? (ns << RES_BITS) | (ne - ns + 1) : 0ull;
                }
            }
        }
        wave_sync();
        // the FM look-ups have been waited for: everything requested before them has arrived as well
        if constexpr (NARROW) { mem_retire(q0c); mem_retire(q1c); mem_retire(phys_nxt); mem_retire(orow); }
        clk.end(9, lane);
        // children per parent, in the reference's order: stay, then bases 0..3
        // bit b: the step with base b was asked for and left a non-empty range (a lane's four result slots; the slots of
        // bases that were not asked for hold stale words and are masked off)
        const uint32_t rb = (uint32_t)lane << 2;
        uint32_t vmask = (s_res[rb] != 0 ? 1u : 0u) | (s_res[rb + 1] != 0 ? 2u : 0u) | (s_res[rb + 2] != 0 ? 4u : 0u) | (s_res[rb + 3] != 0 ? 8u : 0u);
        vmask &= mask;
        const uint32_t nch = (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask);
        uint32_t chtot;
        const uint32_t choff = excl_sum_bits<3>(nch, &chtot);
        const uint32_t room = max_paths - nchild;                 // > 0 here
        const uint32_t nwrite = chtot < room ? chtot : room;      // children that fit (:480,507,521)
        const bool visited = have && choff < room;                // reached before the buffer filled
        // work counter: the get_neighbor calls the reference makes (it stops at the cut-off)
        if (choff + nch < room) c_nbr += ncand;                   // every call precedes the cut-off
        else
            for (uint32_t b = 0; b < 4; ++b)
                if (((mask >> b) & 1u) && choff + (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask & ((1u << b) - 1u)) < room) c_nbr++;
        wave_sync();             // (the candidate list is dead: its space takes the descriptors)
        if constexpr (NARROW) {
            // creation position (inside this pass) of this lane's child of each type: stay, bases 0..3
            const bool is_src = pi >= n_surv_par;          // children of sources: the unsorted run
            uint32_t cpos[5];
            bool ex[5];
            ex[0] = stay_ok; cpos[0] = choff;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                ex[b + 1] = (vmask >> b) & 1u;
                cpos[b + 1] = choff + (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask & ((1u << b) - 1u));
            }
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t) ex[t] = ex[t] && cpos[t] < nwrite;      // the max_paths cut-off
            const bool pass_has_surv = base < n_surv_par, pass_has_src = base + WAVE > n_surv_par;
            uint32_t kp[5] = {0, 0, 0, 0, 0};
            if (pass_has_surv) {
                // a run takes at most one child per parent: position = keys filed + fitting children of earlier lanes
                const uint64_t m0 = __ballot(ex[0] && !is_src), m1 = __ballot(ex[1] && !is_src), m2 = __ballot(ex[2] && !is_src),
                               m3 = __ballot(ex[3] && !is_src), m4 = __ballot(ex[4] && !is_src);
                kp[0] = scnt0 + (uint32_t)prefix_popc(m0); kp[1] = scnt1 + (uint32_t)prefix_popc(m1);
                kp[2] = scnt2 + (uint32_t)prefix_popc(m2); kp[3] = scnt3 + (uint32_t)prefix_popc(m3);
                kp[4] = scnt4 + (uint32_t)prefix_popc(m4);
                scnt0 += (uint32_t)__popcll(m0); scnt1 += (uint32_t)__popcll(m1); scnt2 += (uint32_t)__popcll(m2);
                scnt3 += (uint32_t)__popcll(m3); scnt4 += (uint32_t)__popcll(m4);
            }
            if (pass_has_src) {
                // the unsorted run keeps creation order
                const uint32_t nfit = is_src ? (ex[0] ? 1u : 0u) + (ex[1] ? 1u : 0u) + (ex[2] ? 1u : 0u) + (ex[3] ? 1u : 0u) + (ex[4] ? 1u : 0u) : 0u;
                uint32_t xtot;
                uint32_t xo = scntx + excl_sum_bits<3>(nfit, &xtot);
                if (is_src) {
#pragma unroll
                    for (uint32_t t = 0; t < 5; ++t) { kp[t] = xo; if (ex[t]) ++xo; }
                }
                scntx += xtot;
            }
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t)
                if (ex[t]) {
                    s_cdesc[cpos[t]] = (uint16_t)((uint32_t)lane | (t << 6) | (is_src ? 1u << 9 : 0u));
                    s_ckpos[cpos[t]] = (uint16_t)kp[t];
                }
        } else {
            // wide keys: the same six runs (map_sort_wide.h).  The staging has no room for a position per child: the lane that
            // writes a child works it out from the five masks (a run takes at most one child per parent) and, for the children
            // of sources -- the tail of the pass in creation order -- from where that tail begins
            const bool is_src = pi >= n_surv_par;
            uint32_t cpos[5];
            bool ex[5];
            ex[0] = stay_ok; cpos[0] = choff;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                ex[b + 1] = (vmask >> b) & 1u;
                cpos[b + 1] = choff + (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask & ((1u << b) - 1u));
            }
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t) ex[t] = ex[t] && cpos[t] < nwrite;      // the max_paths cut-off
            wm0 = __ballot(ex[0] && !is_src); wm1 = __ballot(ex[1] && !is_src); wm2 = __ballot(ex[2] && !is_src);
            wm3 = __ballot(ex[3] && !is_src); wm4 = __ballot(ex[4] && !is_src);
            ws0 = scnt0; ws1 = scnt1; ws2 = scnt2; ws3 = scnt3; ws4 = scnt4; wsx = scntx;
            scnt0 += (uint32_t)__popcll(wm0); scnt1 += (uint32_t)__popcll(wm1); scnt2 += (uint32_t)__popcll(wm2);
            scnt3 += (uint32_t)__popcll(wm3); scnt4 += (uint32_t)__popcll(wm4);
            {
                const uint64_t srcm = __ballot(is_src);
                w_src_cpos0 = srcm ? lane_get32(choff, (uint32_t)__ffsll((unsigned long long)srcm) - 1u) : chtot;
                scntx += nwrite > w_src_cpos0 ? nwrite - w_src_cpos0 : 0u;
            }
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t)
                if (ex[t]) s_cdesc[cpos[t]] = (uint16_t)((uint32_t)lane | (t << 6));
        }
        {
            // the children's prob_sums_[0] and the history slid by one event: the window's oldest event drops out, and if the
            // next one was a move its base (the oldest in the queue) joins the k-mer
            float subc = psub;
            uint64_t hist = phist;
            if (pfull) {
                const float d = __fsub_rn(level_old, orow.x);
                const double q = -((double)d * (double)d) / (double)orow.y;
                subc = __fadd_rn(psub, (float)(q - (double)orow.z));
                if ((pmoves >> (SEED_LEN - 2)) & 1u) {
                    const uint32_t cnt = (uint32_t)__popc(pmoves & ((1u << (SEED_LEN - 1)) - 1u));       // bases queued (>= 1 here)
                    const uint32_t pos = 2u * (cnt - 1u);
                    const uint64_t fifo = hist >> 10;
                    const uint32_t nk = ((okmer << 2) & KMASK) | ((uint32_t)(fifo >> pos) & 3u);
                    hist = (uint64_t)nk | ((fifo & ~(3ull << pos)) << 10);
                }
            }
            s_psubc[lane] = subc; s_phist[lane] = hist;
        }
        // dead ends -> update_seeds(prev_path, true), :513-519 / is_seed_valid :842-863
        bool ended_seed = false;
        uint32_t e_count = 0, e_mc = 0;
        if (visited && nch == 0 && !(pmeta & META_SA_CHECKED)) {
            const uint32_t mc = (uint32_t)__popc(pmoves);
            // seed_prob_ of a full-length path: (prob_sums_[22] - prob_sums_[0]) / 22 (prob_sums_[0] is 0 while the window has not slid)
            const float pprob = __fdiv_rn(__fsub_rn(plast, psub), (float)SEED_LEN);
            const bool base_ok = plen == p_seed_len && pprob >= p_min_seed_prob;
            const bool uniq = plen_fm == 1 && (pmoves & 1u) == 1u && (float)(plen - mc) <= p_max_stay;
            const bool rep = plen_fm <= (Row)p_max_rep_copy && mc >= p_min_rep_len;
            ended_seed = base_ok && (uniq || rep);
            e_count = (uint32_t)plen_fm;
            e_mc = mc;
        }
        {
            const uint64_t em = __ballot(ended_seed);
            const uint32_t pos = n_seedp + (uint32_t)prefix_popc(em);
            if (ended_seed) {
                if (pos < max_seed_paths) {
                    SeedPath sp; sp.start = pstart; sp.count = e_count; sp.evt = event_i - 1u; sp.ref_len = e_mc; sp.pad = 0;
                    gst(sb, seedp_off + pos * (uint32_t)sizeof(SeedPath), sp);
                }
            }
            n_seedp += (uint32_t)__popcll(em);
        }
        wave_sync();
        clk.end(10, lane);
        // one lane per child: everything it needs of its parent sits in LDS, the 64-byte record and the sort key go out
        for (uint32_t l0 = 0; l0 < nwrite; l0 += WAVE) {
            const uint32_t li = l0 + (uint32_t)lane;
            if (li < nwrite) {
                const uint32_t d = s_cdesc[li];
                const uint32_t pl = d & 63u, type = (d >> 6) & 7u, ci = (pl << 2) | ((type - 1u) & 3u);
                const uint32_t pmt = s_pmeta[pl], pmv = s_pmoves[pl];
                const float last = s_plast[pl], subc = s_psubc[pl];
                const uint64_t hist = s_phist[pl];
                const uint32_t pk = pmt & META_KMER_MASK;
                Row cs, ce;
                uint32_t ck, mv;
                if (type == 0) { cs = s_pstart[pl]; ce = s_pend[pl]; ck = pk; mv = 0; }
                else {
                    const uint64_t pr = s_res[ci];
                    if constexpr (NARROW) { cs = (uint32_t)pr; ce = cs + (uint32_t)(pr >> 32) - 1u; }
                    else { cs = pr >> RES_BITS; ce = cs + (pr & ((1ull << RES_BITS) - 1ull)) - 1ull; }
                    ck = ((pk << 2) & KMASK) | (type - 1u); mv = 1;
                }
                SortKey key;
                const uint32_t gi = nchild + li;
                const ChildHdr c = make_child(pmv, pmt, last, subc, hist, cs, ce, ck, s_probs[ck], mv, p_seed_len, p_min_seed_prob, p_max_stay, gi, klb, key);
                const uint32_t co = chd_off + (gi << PATH_SHIFT);
                if constexpr (NARROW) gst(sb, co, make_uint4(cs, ce, c.moves, c.meta));
                else { const uint64_t r = (cs << KEY_LEN_BITS) | (ce - cs); gst(sb, co, make_uint4((uint32_t)r, (uint32_t)(r >> 32), c.moves, c.meta)); }
                gst(sb, co + 16u, make_uint4(__float_as_uint(c.last), __float_as_uint(subc), (uint32_t)c.hist, (uint32_t)(c.hist >> 32)));
                if constexpr (NARROW) {
                    const uint32_t run = (d >> 9) ? 5u : type;
                    gst(sb, str_off + run * run_bytes + ((uint32_t)s_ckpos[li] << 3), key.a);
                    gst(sb, info_off + (gi << 3), key.b);
                } else {
                    const bool csrc = base + pl >= n_surv_par;
                    const uint64_t wm = type == 0u ? wm0 : type == 1u ? wm1 : type == 2u ? wm2 : type == 3u ? wm3 : wm4;
                    const uint32_t ws = type == 0u ? ws0 : type == 1u ? ws1 : type == 2u ? ws2 : type == 3u ? ws3 : ws4;
                    const uint32_t kpos = csrc ? wsx + (li - w_src_cpos0) : ws + (uint32_t)__popcll(wm & ((1ull << pl) - 1ull));
                    gst(sb, str_off + (csrc ? 5u : type) * (run_bytes << 1) + (kpos << 4), key);
                }
            }
        }
        nchild += nwrite;
        wave_sync();
        clk.end(11, lane);
    }
    clk.end(1, lane);
    uint32_t tst = 0;
    if (n_seedp > max_seed_paths) { tst = UNC_READ_SEED_OVERFLOW; n_seedp = max_seed_paths; }
    if (lane == 0) {
        s_w.nchild = nchild; s_w.n_seedp = n_seedp; s_w.tstatus |= tst; s_w.par_unsorted = par_bad ? 1u : 0u;
        s_w.scnt[0] = scnt0; s_w.scnt[1] = scnt1; s_w.scnt[2] = scnt2; s_w.scnt[3] = scnt3; s_w.scnt[4] = scnt4; s_w.scnt[5] = scntx;
    }
    wave_sync();
    return c_nbr;
}

static __device__ __noinline__ void merge_walk(kargs_t A_, gptr_t sb_, KeyArr<1> KA_, KeyArr<4> KB_, int lane);
static __device__ __noinline__ void merge_walk_w(kargs_t A_, gptr_t sb_, KeyArr<1> KA_, KeyArr<4> KB_, int lane);

// ---------------- S: the children's keys in the reference's order (mapper.cpp:531, 866-871) ----------------
template <bool NARROW>
static __device__ __noinline__ void phase_S(kargs_t A_, gptr_t sb_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t n = ctx_get(s_w.nchild);
    const uint32_t max_paths = A->sc.max_paths;
    const uint32_t ukeys_off = A->sc.off_keys, skeys_off = A->sc.off_keys + A->sc.keys_cap * (uint32_t)sizeof(SortKey);
    UNC_AS_GLOBAL SortKey *const ukeys = reinterpret_cast<UNC_AS_GLOBAL SortKey *>(sb + ukeys_off);
    UNC_AS_GLOBAL SortKey *const skeys = reinterpret_cast<UNC_AS_GLOBAL SortKey *>(sb + skeys_off);
    uint32_t kl = NARROW ? A->ix.key_len_bits : 0u;     // key mode of THIS event
    if (lane == 0) { s_w.walked = 0u; s_w.mixed = 0u; }
    if constexpr (NARROW) {
        const uint32_t str_off = A->sc.off_streams, sk_off = skeys_off, run_bytes = max_paths << 3;
        uint32_t scnt0 = ctx_get(s_w.scnt[0]), scnt1 = ctx_get(s_w.scnt[1]), scnt2 = ctx_get(s_w.scnt[2]), scnt3 = ctx_get(s_w.scnt[3]),
                 scnt4 = ctx_get(s_w.scnt[4]), nx = ctx_get(s_w.scnt[5]);
#ifdef LANESIM
        {   // emulator only: every key phase E filed names a child of this event, and the runs hold all of them
            const uint32_t cnt[6] = {scnt0, scnt1, scnt2, scnt3, scnt4, nx};
            uint32_t tot = 0;
            for (uint32_t r = 0; r < 6; ++r) {
                tot += cnt[r];
                for (uint32_t i = (uint32_t)lane; i < cnt[r]; i += WAVE)
                    UNC_SIM_CHECK((gld<uint64_t>(sb, str_off + r * run_bytes + (i << 3)) & 0xFFFFu) < n);
            }
            UNC_SIM_CHECK(tot == n);
        }
#endif
        if (n > MERGE_MIN && !ctx_get(s_w.par_unsorted)) {
            // moves of one base: ascending but for the odd pair of nested parents (repair_run moves those to the unsorted run);
            // then the unsorted run is sorted and merged with the moves, and the result with the stays while it is walked
            const uint32_t x_off = str_off + 5u * run_bytes;
            uint32_t nviol = 0;
            SortClock sclk;
            if (scnt1 > 1 || scnt2 > 1 || scnt3 > 1 || scnt4 > 1) {
                const uint64_t v4 = repair_runs4(sb, str_off, run_bytes, scnt1, scnt2, scnt3, scnt4, x_off, nx, lane);
                const uint32_t v1 = (uint32_t)v4 & 0xFFFFu, v2 = (uint32_t)(v4 >> 16) & 0xFFFFu, v3 = (uint32_t)(v4 >> 32) & 0xFFFFu, v4r = (uint32_t)(v4 >> 48);
                nviol = v1 + v2 + v3 + v4r;
                if constexpr (MERGE_REPAIR) { scnt1 -= v1; scnt2 -= v2; scnt3 -= v3; scnt4 -= v4r; nx += nviol; }
            }
            wave_sync();
            sclk.end(8, lane);
            if (MERGE_REPAIR || nviol == 0) {       // (test build without the repair: an event with such a pair takes the network below)
                if (nx > 1) {
                    KeyArr<6> KX;
#pragma unroll
                    for (int r = 0; r < 6; ++r) { KX.adj[r] = x_off; KX.cum[r] = 0; }
                    KX.n = nx;
                    sort_any64(sb, KX, x_off, lane);          // in place
                    wave_sync();
                }
                sclk.end(9, lane);
                KeyArr<4> KM;       // the moves, base by base
                KM.cum[0] = 0; KM.cum[1] = scnt1; KM.cum[2] = scnt1 + scnt2; KM.cum[3] = scnt1 + scnt2 + scnt3;
                KM.n = KM.cum[3] + scnt4;
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) KM.adj[r] = str_off + (r + 1u) * run_bytes - (KM.cum[r] << 3);
                KeyArr<4> KB = KM;  // what the stays are merged with
                if (nx > 0) {
                    if (KM.n > 0) {
                        merge_runs<4, 1>(sb, KM, ka_single(x_off, nx), A->sc.off_tmp, lane, 0u);
                        wave_sync();
                        KB.adj[0] = A->sc.off_tmp;
                    } else KB.adj[0] = x_off;
                    KB.cum[1] = KB.cum[2] = KB.cum[3] = KB.n = KM.n + nx;
                    KB.adj[1] = KB.adj[2] = KB.adj[3] = KB.adj[0];
                }
                sclk.end(10, lane);
                merge_walk(A, sb, ka_single(str_off, scnt0), KB, lane);
                CTX_SET(kl, kl);
                CTX_SET(walked, 1u);
                wave_sync();
                sclk.end(11, lane);
                return;
            }
        }
        {
            // few children, or parents that were not in order: the bitonic network over all of them
            KeyArr<6> K6;
            const uint32_t c[6] = {scnt0, scnt1, scnt2, scnt3, scnt4, nx};
            uint32_t cum = 0;
#pragma unroll
            for (uint32_t r = 0; r < 6; ++r) { K6.cum[r] = cum; K6.adj[r] = str_off + r * run_bytes - (cum << 3); cum += c[r]; }
            K6.n = cum;
            sort_any64(sb, K6, sk_off, lane);
        }
    }
    if constexpr (!NARROW) {
        // wide keys: the same runs, 16 bytes per key (map_sort_wide.h).  Moves repaired, the unsorted run sorted (into the sorted-keys
        // area: the network pads to a power of two) and merged with the moves, the result merged with the stays and walked in LDS.
        const uint32_t str_off = A->sc.off_streams, run_bytes = max_paths << 4;
        uint32_t scnt0 = ctx_get(s_w.scnt[0]), scnt1 = ctx_get(s_w.scnt[1]), scnt2 = ctx_get(s_w.scnt[2]), scnt3 = ctx_get(s_w.scnt[3]),
                 scnt4 = ctx_get(s_w.scnt[4]), nx = ctx_get(s_w.scnt[5]);
        const uint32_t x_off = str_off + 5u * run_bytes;
        if (n > MERGE_MIN && !ctx_get(s_w.par_unsorted)) {
            if (scnt1 > 1) { const uint32_t v = repairw_run(sb, str_off + run_bytes, scnt1, x_off, nx, lane); scnt1 -= v; nx += v; }
            if (scnt2 > 1) { const uint32_t v = repairw_run(sb, str_off + 2u * run_bytes, scnt2, x_off, nx, lane); scnt2 -= v; nx += v; }
            if (scnt3 > 1) { const uint32_t v = repairw_run(sb, str_off + 3u * run_bytes, scnt3, x_off, nx, lane); scnt3 -= v; nx += v; }
            if (scnt4 > 1) { const uint32_t v = repairw_run(sb, str_off + 4u * run_bytes, scnt4, x_off, nx, lane); scnt4 -= v; nx += v; }
            wave_sync();
            uint32_t xs_off = x_off;              // where the sorted unsorted-run lies
            if (nx > 1) {
                const UNC_AS_GLOBAL SortKey *const xin = reinterpret_cast<const UNC_AS_GLOBAL SortKey *>(sb + x_off);
                if (nx <= 64) sort_regs<1>(xin, skeys, nx, lane);
                else if (nx <= 128) sort_regs<2>(xin, skeys, nx, lane);
                else if (nx <= 256) sort_regs<4>(xin, skeys, nx, lane);
                else if (nx <= 512) sort_regs<8>(xin, skeys, nx, lane);
                else sort_hybrid(xin, skeys, nx, lane);
                xs_off = skeys_off;
                wave_sync();
            }
            KeyArr<4> KM;       // the moves, base by base
            KM.cum[0] = 0; KM.cum[1] = scnt1; KM.cum[2] = scnt1 + scnt2; KM.cum[3] = scnt1 + scnt2 + scnt3;
            KM.n = KM.cum[3] + scnt4;
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) KM.adj[r] = str_off + (r + 1u) * run_bytes - (KM.cum[r] << 4);
            KeyArr<4> KB = KM;  // what the stays are merged with
            if (nx > 0) {
                if (KM.n > 0) {
                    mergew_runs<4, 1>(sb, KM, kaw_single(xs_off, nx), ukeys_off, lane);      // (the unsorted-keys area is free in this mode)
                    wave_sync();
                    KB.adj[0] = ukeys_off;
                } else KB.adj[0] = xs_off;
                KB.cum[1] = KB.cum[2] = KB.cum[3] = KB.n = KM.n + nx;
                KB.adj[1] = KB.adj[2] = KB.adj[3] = KB.adj[0];
            }
            merge_walk_w(A, sb, kaw_single(str_off, scnt0), KB, lane);
            CTX_SET(kl, 0u);
            CTX_SET(walked, 1u);
            wave_sync();
            return;
        }
        // few children, or parents that were not in order: all runs into the unsorted-keys area, then the network as before
        {
            const uint32_t c[6] = {scnt0, scnt1, scnt2, scnt3, scnt4, nx};
            uint32_t cum = 0;
#pragma unroll
            for (uint32_t r = 0; r < 6; ++r) {
                for (uint32_t i = (uint32_t)lane; i < c[r]; i += WAVE) g_store(ukeys + cum + i, gld<SortKey>(sb, str_off + r * run_bytes + (i << 4)));
                cum += c[r];
            }
            UNC_SIM_CHECK(cum == n);
            wave_sync();
        }
    }
    if (!kl) {
        if (n <= 64) sort_regs<1>(ukeys, skeys, n, lane);
        else if (n <= 128) sort_regs<2>(ukeys, skeys, n, lane);
        else if (n <= 256) sort_regs<4>(ukeys, skeys, n, lane);
        else if (n <= 512) sort_regs<8>(ukeys, skeys, n, lane);
        else sort_hybrid(ukeys, skeys, n, lane);
    }
    CTX_SET(kl, kl);
    wave_sync();
}

// ---------------- W: walk in sorted order (mapper.cpp:533-603) ----------------
// duplicate-range pruning, per-k-mer gap sources from a segmented prefix-max (the sequential unchecked_range logic in closed
// form), survivors -> next parent list, seed-valid survivors -> seed list.  One pass = 64 consecutive sorted positions.
template <bool NARROW> struct WalkConst {
    gptr_t sb, chd;                                   // the slot; its child buffer
    const UNC_AS_GLOBAL uint64_t *kmer_ranges;
    uint32_t n, room, nord_off, seedp_off, max_seed_paths, event_i;
    float source_prob;
};
template <bool NARROW> struct WalkState {
    using Row = std::conditional_t<NARROW, uint32_t, uint64_t>;
    uint32_t n_surv = 0, n_src = 0, n_seedp = 0;
    uint32_t carry_kmer = NKMER;
    Row carry_U = 0;
    uint64_t carry_range = ~0ull, carry_w = 0;         // narrow keys: range / running info maximum of the run that is open at a pass boundary
    bool mixed = false;                                // narrow keys: this lane saw two neighbours with equal ranges and different k-mers
};
// BwaIndex::get_base_range starts one row low (bwa_index.hpp:172-174), so neighbouring k-mer ranges can share a boundary row and two
// children with the SAME one-row range may carry DIFFERENT k-mers.  The reference walks them in seed_prob order (mapper.cpp:543-563
// runs per position), which the narrow key cannot express.  Rounds 1-5 tested every one-row child against its k-mer's first and last
// row in phase E (two loads per child) and sent such events through memory; now the walk itself notices the pair -- equal keys are
// neighbours in the sorted order -- and the event's walk is done again on 128-bit keys (walk_begin_narrow / phase_S_wide_redo).  What
// a narrow walk changes before it notices is put back or overwritten: sources_added_ (saved here), the lists it appends to (started
// again), the sa_checked_ bits of seed paths (set again by the second walk on every child that survives it; a child that does not is
// dead).
__device__ __forceinline__ void walk_begin_narrow(int lane) {
    if (lane < NKMER / 32) s_w.flags_save[lane] = s_flags[lane];
}
template <bool NARROW> __device__ __forceinline__ WalkConst<NARROW> walk_const(kargs_t A, gptr_t sb) {
    WalkConst<NARROW> C;
    const uint32_t cur = ctx_get(s_w.cur), max_paths = A->sc.max_paths;
    C.sb = sb; C.chd = sb + A->sc.off_paths + (cur ^ 1u) * (max_paths << PATH_SHIFT);
    C.kmer_ranges = (const UNC_AS_GLOBAL uint64_t *)A->ix.kmer_ranges;
    C.n = ctx_get(s_w.nchild); C.room = max_paths - C.n;      // room: sources that still fit
    C.nord_off = A->sc.off_order + (cur ^ 1u) * (max_paths << 2); C.seedp_off = A->sc.off_seedp; C.max_seed_paths = A->sc.max_seed_paths;
    C.event_i = ctx_get(s_w.event_i);
    C.source_prob = A->ix.thresholds[0];      // Mapper::get_source_prob, mapper.cpp:169-171
    return C;
}
template <bool NARROW> __device__ __forceinline__ void walk_finish(const WalkConst<NARROW> &C, WalkState<NARROW> &S, int lane) {
    if (__any(S.mixed)) {        // (narrow keys only) nothing of this walk counts: phase_S_wide_redo + phase_W follow
        if (lane == 0) s_w.mixed = 1u;
        wave_sync();
        return;
    }
    uint32_t tst = 0;
    if (S.n_seedp > C.max_seed_paths) { tst = UNC_READ_SEED_OVERFLOW; S.n_seedp = C.max_seed_paths; }
    if (lane == 0) { s_w.n_surv = S.n_surv; s_w.n_src = S.n_src; s_w.n_seedp = S.n_seedp; s_w.tstatus |= tst; }
    wave_sync();
}

// narrow keys: this lane's sorted key ki and info word bi and its successor's kn, bn -> range, k-mers, duplicate flag and the
// info word that survives at this position.  Among equal ranges the reference keeps the highest (seed_prob, creation
// order): a running max of the info words over each run, read off at the run's last position.  nv = valid lanes of the pass.
template <bool NARROW>
__device__ __forceinline__ void walk_decode_narrow(WalkState<NARROW> &S, uint32_t kl, uint64_t ki, uint64_t kn, uint64_t bi, uint64_t bn, bool have,
                                                   bool has_next, uint32_t nv, int lane, typename WalkState<NARROW>::Row &start,
                                                   typename WalkState<NARROW>::Row &end, typename WalkState<NARROW>::Row &nstart, uint32_t &kmer,
                                                   uint32_t &nkmer, bool &dup, uint64_t &sb_) {
    using Row = typename WalkState<NARROW>::Row;
    const uint64_t ri = ki >> 16, rn = kn >> 16;
    start = (Row)(ri >> kl); end = start + (Row)(ri & ((1ull << kl) - 1ull));
    nstart = (Row)(rn >> kl);
    kmer = have ? (uint32_t)(bi & META_KMER_MASK) : NKMER + 1u;
    nkmer = has_next ? (uint32_t)(bn & META_KMER_MASK) : NKMER + 2u;
    dup = has_next && rn == ri;
    if (dup && nkmer != kmer) S.mixed = true;
    uint64_t pr = (uint64_t)__shfl_up((unsigned long long)ri, 1);
    if (lane == 0) pr = S.carry_range;
    const bool rhead = !have || ri != pr;
    const uint64_t rheads = __ballot(rhead);
    sb_ = bi;
    if (rheads != ~0ull) {          // (nearly every pass: no two equal ranges, no scan)
        sb_ = seg_incl_max64(bi, rhead);
        if ((rheads & ((2ull << lane) - 1ull)) == 0 && S.carry_w > sb_) sb_ = S.carry_w;
    }
    S.carry_range = bcast64(ri, (int)nv - 1);
    S.carry_w = bcast64(sb_, (int)nv - 1);
}

// the walk proper for one pass.  krc: the full range of this lane's k-mer when the driver has fetched it (narrow keys), else
// it is read here.
template <bool NARROW>
__device__ __forceinline__ void walk_core(const WalkConst<NARROW> &C, WalkState<NARROW> &S, bool have, bool has_next,
                                          typename WalkState<NARROW>::Row start, typename WalkState<NARROW>::Row end,
                                          typename WalkState<NARROW>::Row nstart, uint32_t kmer, uint32_t nkmer, bool dup, uint64_t sb_,
                                          bool have_krc, const ulonglong2 &krc, uint32_t nv, int lane) {
    using Row = typename WalkState<NARROW>::Row;
    const uint32_t n = C.n, room = C.room;
    const uint32_t idx = (uint32_t)(sb_ >> 16) & 0xFFFFu;
    UNC_SIM_CHECK(!(have && !dup) || idx < n);
    uint32_t pk = (uint32_t)__shfl_up((int)kmer, 1);
    if (lane == 0) pk = S.carry_kmer;
    const bool first = have && kmer != pk;              // source_kmer != prev_kmer, :543
    const bool next_same = has_next && nkmer == kmer;
    const bool psrc = have && s_probs[have ? kmer : 0] >= C.source_prob;
    // unchecked_range.start_ when step C runs for i = running max of (end + 1) in the k-mer group
    Row U;
    if constexpr (NARROW) U = seg_incl_max32(have ? end + 1u : 0u, first || !have);
    else U = seg_incl_max64(have ? end + 1 : 0, first || !have);
    const uint64_t heads = __ballot(first || !have);
    const bool headless = (heads & ((2ull << lane) - 1ull)) == 0;   // group began in an earlier pass
    if (headless && S.carry_U > U) U = S.carry_U;
    Row kr_s = 1, kr_e = 0;
    if (have_krc) {
        if (first && psrc) kr_s = (Row)krc.x;
        if (have && !dup && psrc && !next_same) kr_e = (Row)krc.y;
    } else {
        if (first && psrc) kr_s = (Row)C.kmer_ranges[2 * kmer];
        if (have && !dup && psrc && !next_same) kr_e = (Row)C.kmer_ranges[2 * kmer + 1];
    }
    const bool a_valid = first && psrc && kr_s <= start - 1;                       // :549-557
    const Row c_s = U, c_e = next_same ? nstart - 1 : kr_e;                        // :579-589
    const bool c_valid = have && !dup && psrc && c_s <= c_e;                       // :592
    uint32_t stot;
    const uint32_t soff = excl_sum_bits<2>((a_valid ? 1u : 0u) + (c_valid ? 1u : 0u), &stot);
    const uint32_t q0 = S.n_src + soff;                    // sources appended before this child
    const bool not_full0 = q0 < room;                      // next_path != end at step A
    if (first && psrc && not_full0) atomicOr(&s_flags[(kmer & 63u) >> 1], 1u << (((kmer & 1u) << 4) + (kmer >> 6)));   // :547
    if (a_valid && not_full0) write_source<NARROW>(C.chd, n + q0, kr_s, start - 1, kmer, s_probs[kmer]);
    const uint32_t qc = q0 + (a_valid ? 1u : 0u);
    if (c_valid && qc < room) write_source<NARROW>(C.chd, n + qc, c_s, c_e, kmer, s_probs[kmer]);
    // survivors keep sorted order in the next parent list
    const bool surv = have && !dup;
    const uint64_t sm = __ballot(surv);
    if (surv) gst(C.sb, C.nord_off + ((S.n_surv + (uint32_t)prefix_popc(sm)) << 2), idx);
    S.n_surv += (uint32_t)__popcll(sm);
    // update_seeds(child, false), :601 -- validity was decided at creation
    const bool sv = surv && (sb_ & KEYB_SEED_FLAG);
    const uint64_t svm = __ballot(sv);
    if (sv) {
        const uint32_t pos = S.n_seedp + (uint32_t)prefix_popc(svm);
        // path.sa_checked_ = true (no value returned: no round trip)
        __hip_atomic_fetch_or(reinterpret_cast<UNC_AS_GLOBAL uint32_t *>(C.chd + (idx << PATH_SHIFT) + 12u), META_SA_CHECKED, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (pos < C.max_seed_paths) {
            SeedPath sp; sp.start = start; sp.count = 1; sp.evt = C.event_i;
            sp.ref_len = (uint32_t)(sb_ >> KEYB_MOVES_SHIFT) & 31u; sp.pad = 0;
            gst(C.sb, C.seedp_off + pos * (uint32_t)sizeof(SeedPath), sp);
        }
    }
    S.n_seedp += (uint32_t)__popcll(svm);
    // carries into the next pass
    S.carry_kmer = bcast32(kmer, (int)nv - 1);
    if constexpr (NARROW) S.carry_U = bcast32(U, (int)nv - 1); else S.carry_U = bcast64(U, (int)nv - 1);
    S.n_src += stot;
    if (S.n_src > room) S.n_src = room;
}

// the walk over sorted keys that lie in global memory (few children: the bitonic network's output; 128-bit keys)
template <bool NARROW>
static __device__ __noinline__ void phase_W(kargs_t A_, gptr_t sb_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    using Row = std::conditional_t<NARROW, uint32_t, uint64_t>;
    const WalkConst<NARROW> C = walk_const<NARROW>(A, sb);
    WalkState<NARROW> S;
    S.n_seedp = ctx_get(s_w.n_seedp);
    const uint32_t n = C.n, kl = ctx_get(s_w.kl);
    const float source_prob = C.source_prob;
    const UNC_AS_GLOBAL ulonglong2 *const kmer_ranges2 = (const UNC_AS_GLOBAL ulonglong2 *)A->ix.kmer_ranges;
    const UNC_AS_GLOBAL SortKey *const skeys = reinterpret_cast<const UNC_AS_GLOBAL SortKey *>(sb + A->sc.off_keys + A->sc.keys_cap * (uint32_t)sizeof(SortKey));
    const UNC_AS_GLOBAL uint64_t *const skeys64 = reinterpret_cast<const UNC_AS_GLOBAL uint64_t *>(skeys);
    const UNC_AS_GLOBAL uint64_t *const infow = reinterpret_cast<const UNC_AS_GLOBAL uint64_t *>(sb + A->sc.off_info);   // narrow keys: info words by creation index

    // narrow keys: the sorted keys of the pass after next and the info words (seed_prob, idx, flags, k-mer) they
    // point to for the next pass are fetched while this pass is worked on; a lane's successor comes from
    // its neighbour lane, the last lane's from the next pass
    // (every lane asks in every pass -- a lane past the end for the last key, a lane whose k-mer starts no source for entry 0 -- so that
    // the waits can be counted: see merge_walk)
    uint64_t kq0 = ~0ull, kq1 = ~0ull, bq0 = 0, bq1 = 0;
    u32x4_t krq = {1u, 0u, 0u, 0u};
    if (kl) {
        walk_begin_narrow(lane);
        const uint64_t k0 = skeys64[(uint32_t)lane < n ? (uint32_t)lane : n - 1u], k1 = skeys64[(uint32_t)lane + WAVE < n ? (uint32_t)lane + WAVE : n - 1u];
        bq0 = infow[k0 & 0xFFFFu]; bq1 = infow[k1 & 0xFFFFu];
        kq0 = (uint32_t)lane < n ? k0 : ~0ull; kq1 = (uint32_t)lane + WAVE < n ? k1 : ~0ull;
        // ... and so is the k-mer's full range, for the few children whose k-mer may start a source
        const uint32_t km0 = (uint32_t)(bq0 & META_KMER_MASK);
        krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[km0] >= source_prob ? km0 : 0u));
    }
    for (uint32_t base = 0; base < n; base += WAVE) {
        const uint32_t i = base + (uint32_t)lane;
        const bool have = i < n;
        const bool has_next = i + 1 < n;
        const uint32_t nv = n - base < WAVE ? n - base : WAVE;
        Row start, end, nstart;
        uint64_t sbw;                         // info word of the child that survives at this position
        uint32_t kmer, nkmer;
        bool dup;
        ulonglong2 krc = make_ulonglong2(1ull, 0ull);
        uint32_t kq2i = 0;
        if (kl) {
            const uint64_t ki = kq0, bi = bq0;
            uint64_t kn = (uint64_t)__shfl((unsigned long long)ki, (lane + 1) & 63);
            uint64_t bn = (uint64_t)__shfl((unsigned long long)bi, (lane + 1) & 63);
            const uint64_t kf = bcast64(kq1, 0), bf = bcast64(bq1, 0);
            if (lane == WAVE - 1) { kn = kf; bn = bf; }
            kq0 = kq1; bq0 = bq1;
            {
                const uint64_t k2 = skeys64[i + 2 * WAVE < n ? i + 2 * WAVE : n - 1u];
                kq1 = i + 2 * WAVE < n ? k2 : ~0ull;
                kq2i = (uint32_t)(k2 & 0xFFFFu);
            }
            u32x4_t kt = krq;
            mem_retire(kt);
            krc.x = ((uint64_t)kt.y << 32) | kt.x; krc.y = ((uint64_t)kt.w << 32) | kt.z;
            {
                const uint32_t kmn = (uint32_t)(bq0 & META_KMER_MASK);
                krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[kmn] >= source_prob ? kmn : 0u));
            }
            walk_decode_narrow<NARROW>(S, kl, ki, kn, bi, bn, have, has_next, nv, lane, start, end, nstart, kmer, nkmer, dup, sbw);
        } else {
            SortKey ki, kn;
            ki.a = ~0ull; ki.b = 0; kn.a = ~0ull; kn.b = ~0ull;
            if (have) ki = g_load(skeys + i);
            if (has_next) kn = g_load(skeys + i + 1);
            start = (Row)(ki.a >> KEY_LEN_BITS); end = start + (Row)(ki.a & KEY_LEN_MASK);
            nstart = (Row)(kn.a >> KEY_LEN_BITS);
            kmer = have ? (uint32_t)(ki.b & META_KMER_MASK) : NKMER + 1u;
            nkmer = has_next ? (uint32_t)(kn.b & META_KMER_MASK) : NKMER + 2u;
            dup = has_next && kn.a == ki.a;          // equal fm_range_, :569
            sbw = ki.b;                              // sorted by seed_prob inside the run: the last one survives
        }
        walk_core<NARROW>(C, S, have, has_next, start, end, nstart, kmer, nkmer, dup, sbw, kl != 0, krc, nv, lane);
        if (kl) bq1 = infow[kq2i];
    }
    walk_finish<NARROW>(C, S, lane);
}

// narrow keys, many children: the LAST merge -- the stays with everything else -- and the walk in one.  A tile of sorted keys
// is left in LDS and walked there; the sorted keys never go to memory (the kernel is bound by the bytes it moves).  The
// walk looks one key ahead, so a tile's last key waits for the next tile.  Both inputs are ascending by construction
// (phase E checks the parents' order, repair_run the moves, the unsorted run went through the network); should the output
// not be, the read is marked UNC_READ_SORT_FAULT.
static __device__ __noinline__ void merge_walk(kargs_t A_, gptr_t sb_, KeyArr<1> KA_, KeyArr<4> KB_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    uint64_t *const s_tile = s_e;
    const KeyArr<1> KA = ka_uniform(KA_);
    const KeyArr<4> KB = ka_uniform(KB_);
    const uint32_t n = KA.n + KB.n;
    const WalkConst<true> C = walk_const<true>(A, sb);
    WalkState<true> S;
    S.n_seedp = ctx_get(s_w.n_seedp);
    walk_begin_narrow(lane);
    const uint32_t kl = A->ix.key_len_bits;
    const float source_prob = C.source_prob;
    const UNC_AS_GLOBAL ulonglong2 *const kmer_ranges2 = (const UNC_AS_GLOBAL ulonglong2 *)A->ix.kmer_ranges;
    const UNC_AS_GLOBAL uint64_t *const infow = reinterpret_cast<const UNC_AS_GLOBAL uint64_t *>(sb + A->sc.off_info);
    uint32_t a0 = 0, b0 = 0;
    uint64_t pend_key = 0;           // the previous tile's last key: walked first in this tile
    bool have_pend = false, bad = false;
    for (uint32_t o0 = 0; o0 < n; o0 += MERGE_TILE) {
        const uint32_t d1 = o0 + MERGE_TILE < n ? o0 + MERGE_TILE : n;
        const bool last_tile = d1 == n;
        const uint32_t a1 = last_tile ? KA.n : merge_split(sb, KA, KB, d1, lane);
        const uint32_t b1 = d1 - a1;
        const uint32_t na = a1 - a0, nb = b1 - b0, tn = na + nb;
        stage_tile(sb, KA, KB, a0, b0, na, tn, s_tile, lane);
        wave_sync();
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
        {   // the tile must come out ascending, and after the previous tile's last key
            uint64_t last = o[0];
            bool w = false;
#pragma unroll
            for (uint32_t c = 1; c < MERGE_C; ++c)
                if (c < cnt) { w = w || !(o[c] > last); last = o[c]; }
            uint64_t pl = (uint64_t)__shfl_up((unsigned long long)last, 1);
            if (lane == 0) pl = pend_key;
            if (cnt > 0 && !(o[0] > pl)) w = true;
            if (__any(w)) bad = true;
        }
        // sorted tile at logical positions 1 .. tn, the waiting key at 0
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGE_C; ++c)
            if (c < cnt) s_tile[mslot(1u + d + c)] = o[c];
        if (lane == 0 && have_pend) s_tile[mslot(0)] = pend_key;
        wave_sync();
#ifdef LANESIM
        for (uint32_t q = 1u + (uint32_t)lane; q <= tn; q += WAVE) {       // emulator only: the tile holds keys of this event's children
            const uint64_t kq = s_tile[mslot(q)];
            UNC_SIM_CHECK((kq & 0xFFFFu) < n);
            UNC_SIM_CHECK(((infow[kq & 0xFFFFu] >> 16) & 0xFFFFu) == (kq & 0xFFFFu));
        }
        wave_sync();
#endif
        // ---- the walk over logical positions [p0, p1): a key's successor sits one position on (none after the last key of all)
        const uint32_t p0 = have_pend ? 0u : 1u, p1 = last_tile ? tn + 1u : tn;
        // Every load of the walk is issued by EVERY lane in EVERY pass (a lane past the end asks for the tile's last key's word, a lane
        // whose k-mer starts no source for entry 0): a load inside a branch is one the compiler cannot count, and a wait it cannot
        // count is vmcnt(0) -- which drained the info words of the pass after next right behind their request (round 6, off the ISA:
        // each pass of the walk paid a full memory round trip).  With unconditional loads the waits name how many younger requests
        // may stay in flight.
        uint64_t bq0, bq1;
        {
            const uint32_t pa = p0 + (uint32_t)lane, pb = pa + WAVE;
            bq0 = infow[s_tile[mslot(pa <= tn ? pa : tn)] & 0xFFFFu];
            bq1 = infow[s_tile[mslot(pb <= tn ? pb : tn)] & 0xFFFFu];
        }
        // (the k-mer's range as ONE register tuple that stays whole until it is used: of a 16-byte load whose upper words are dead the
        // freed registers go to the next instruction that needs one, which then has to wait for the load)
        u32x4_t krq;
        {
            const uint32_t km0 = (uint32_t)(bq0 & META_KMER_MASK);
            krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[km0] >= source_prob ? km0 : 0u));
        }
        for (uint32_t base = p0; base < p1; base += WAVE) {
            const uint32_t p = base + (uint32_t)lane;
            const bool have = p < p1;
            const bool has_next = have && p < tn;
            const uint32_t nv = p1 - base < WAVE ? p1 - base : WAVE;
            // the info words of the pass after next are requested BEFORE this pass's work, not behind it: the gather then has a whole pass
            // to arrive (round 5, hits identical; -2.9 % in one same-call pair, +0.6 % in the next: within what one library's two
            // regimes differ by, profiles/r05_ab_walk_prefetch.log)
            const uint64_t bq2 = infow[s_tile[mslot(p + 2 * WAVE <= tn ? p + 2 * WAVE : tn)] & 0xFFFFu];
            const uint64_t ki = have ? s_tile[mslot(p)] : ~0ull, kn = has_next ? s_tile[mslot(p + 1u)] : ~0ull;
            const uint64_t bi = bq0;
            uint64_t bn = (uint64_t)__shfl((unsigned long long)bi, (lane + 1) & 63);
            const uint64_t bf = bcast64(bq1, 0);
            if (lane == WAVE - 1) bn = bf;
            bq0 = bq1;
            u32x4_t kt = krq;                 // (used only where this lane's k-mer passes source_prob: the entry asked for was its own)
            mem_retire(kt);                   // requested a pass ago; the info words just asked for stay in flight
            ulonglong2 krc;
            krc.x = ((uint64_t)kt.y << 32) | kt.x; krc.y = ((uint64_t)kt.w << 32) | kt.z;
            {
                const uint32_t kmn = (uint32_t)(bq0 & META_KMER_MASK);
                krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[kmn] >= source_prob ? kmn : 0u));
            }
            uint32_t start, end, nstart, kmer, nkmer;
            uint64_t sbw;
            bool dup;
            walk_decode_narrow<true>(S, kl, ki, kn, bi, bn, have, has_next, nv, lane, start, end, nstart, kmer, nkmer, dup, sbw);
            walk_core<true>(C, S, have, has_next, start, end, nstart, kmer, nkmer, dup, sbw, true, krc, nv, lane);
            bq1 = bq2;
        }
        pend_key = uniform64(s_tile[mslot(tn)]);
        have_pend = true;
        a0 = a1; b0 = b1;
        wave_sync();
    }
    if (bad && lane == 0) s_w.tstatus |= UNC_READ_SORT_FAULT;
    walk_finish<true>(C, S, lane);
}

// wide keys, many children: the last merge -- the stays with everything else -- and the walk in one (merge_walk for 16-byte keys:
// a key carries its own info word, the walk reads range, k-mer, seed probability and creation index off the tile)
static __device__ __noinline__ void merge_walk_w(kargs_t A_, gptr_t sb_, KeyArr<1> KA_, KeyArr<4> KB_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    SortKey *const tile = reinterpret_cast<SortKey *>(s_e);
    const KeyArr<1> KA = ka_uniform(KA_);
    const KeyArr<4> KB = ka_uniform(KB_);
    const uint32_t n = KA.n + KB.n;
    const WalkConst<false> C = walk_const<false>(A, sb);
    WalkState<false> S;
    S.n_seedp = ctx_get(s_w.n_seedp);
    uint32_t a0 = 0, b0 = 0;
    SortKey pend_key; pend_key.a = 0; pend_key.b = 0;      // the previous tile's last key: walked first in this tile
    bool have_pend = false, bad = false;
    for (uint32_t o0 = 0; o0 < n; o0 += MERGEW_TILE) {
        const uint32_t d1 = o0 + MERGEW_TILE < n ? o0 + MERGEW_TILE : n;
        const bool last_tile = d1 == n;
        const uint32_t a1 = last_tile ? KA.n : mergew_split(sb, KA, KB, d1, lane);
        const uint32_t b1 = d1 - a1;
        const uint32_t na = a1 - a0, nb = b1 - b0, tn = na + nb;
        stagew_tile(sb, KA, KB, a0, b0, na, tn, tile, lane);
        wave_sync();
        SortKey o[MERGEW_C];
        uint32_t d, cnt;
        mergew_tile(tile, na, nb, lane, o, d, cnt);
        {   // the tile must come out ascending, and after the previous tile's last key
            SortKey last = o[0];
            bool w = false;
#pragma unroll
            for (uint32_t c = 1; c < MERGEW_C; ++c)
                if (c < cnt) { w = w || !wk_lt(last, o[c]); last = o[c]; }
            SortKey pl;
            pl.a = (uint64_t)__shfl_up((unsigned long long)last.a, 1); pl.b = (uint64_t)__shfl_up((unsigned long long)last.b, 1);
            if (lane == 0) pl = pend_key;
            if (cnt > 0 && !wk_lt(pl, o[0])) w = true;
            if (__any(w)) bad = true;
        }
        // sorted tile at logical positions 1 .. tn, the waiting key at 0
        wave_sync();
#pragma unroll
        for (uint32_t c = 0; c < MERGEW_C; ++c)
            if (c < cnt) tile[1u + d + c] = o[c];
        if (lane == 0 && have_pend) tile[0] = pend_key;
        wave_sync();
        const uint32_t p0 = have_pend ? 0u : 1u, p1 = last_tile ? tn + 1u : tn;
        // the k-mer's full range, for the few children whose k-mer may start a source: asked for a pass ahead, by every lane (one
        // countable request per pass; a lane whose k-mer starts no source asks for entry 0), as in merge_walk
        const UNC_AS_GLOBAL ulonglong2 *const kmer_ranges2 = (const UNC_AS_GLOBAL ulonglong2 *)A->ix.kmer_ranges;
        ulonglong2 krq;
        {
            const uint32_t pa = p0 + (uint32_t)lane;
            const uint32_t km0 = (uint32_t)(tile[pa <= tn ? pa : tn].b & META_KMER_MASK);
            krq = g_load(kmer_ranges2 + (s_probs[km0] >= C.source_prob ? km0 : 0u));
        }
        for (uint32_t base = p0; base < p1; base += WAVE) {
            const uint32_t p = base + (uint32_t)lane;
            const bool have = p < p1;
            const bool has_next = have && p < tn;
            const uint32_t nv = p1 - base < WAVE ? p1 - base : WAVE;
            ulonglong2 krc = krq;
            mem_retire(krc);
            {
                const uint32_t pn = p + WAVE;
                const uint32_t kmn = (uint32_t)(tile[pn <= tn ? pn : tn].b & META_KMER_MASK);
                krq = g_load(kmer_ranges2 + (s_probs[kmn] >= C.source_prob ? kmn : 0u));
            }
            SortKey ki, kn;
            ki.a = ~0ull; ki.b = 0; kn.a = ~0ull; kn.b = ~0ull;
            if (have) ki = tile[p];
            if (has_next) kn = tile[p + 1u];
            const uint64_t start = ki.a >> KEY_LEN_BITS, end = start + (ki.a & KEY_LEN_MASK), nstart = kn.a >> KEY_LEN_BITS;
            const uint32_t kmer = have ? (uint32_t)(ki.b & META_KMER_MASK) : NKMER + 1u;
            const uint32_t nkmer = has_next ? (uint32_t)(kn.b & META_KMER_MASK) : NKMER + 2u;
            const bool dup = has_next && kn.a == ki.a;          // equal fm_range_, :569 (sorted by seed_prob inside the run: the last one survives)
            walk_core<false>(C, S, have, has_next, start, end, nstart, kmer, nkmer, dup, ki.b, true, krc, nv, lane);
        }
        pend_key.a = uniform64(tile[tn].a); pend_key.b = uniform64(tile[tn].b);
        have_pend = true;
        a0 = a1; b0 = b1;
        wave_sync();
    }
    if (bad && lane == 0) s_w.tstatus |= UNC_READ_SORT_FAULT;
    walk_finish<false>(C, S, lane);
}

// narrow keys, after a walk that met equal ranges with different k-mers (WalkState::mixed): the event's children under their full
// 128-bit keys -- range from the child's record, info word as phase E filed it -- through the bitonic network; phase_W follows.
#ifdef LANESIM
}  // namespace unc
extern "C" { unsigned long long unc_sim_wide_redo_count = 0; }      // emulator only: how often the suite takes this path (tests assert it does)
namespace unc {
#endif
static __device__ __noinline__ void phase_S_wide_redo(kargs_t A_, gptr_t sb_, int lane) {
#ifdef LANESIM
    if (lane == 0) ++unc_sim_wide_redo_count;
#endif
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t n = ctx_get(s_w.nchild), cur = ctx_get(s_w.cur), max_paths = A->sc.max_paths;
    const uint32_t chd_off = A->sc.off_paths + (cur ^ 1u) * (max_paths << PATH_SHIFT), info_off = A->sc.off_info;
    UNC_AS_GLOBAL SortKey *const ukeys = reinterpret_cast<UNC_AS_GLOBAL SortKey *>(sb + A->sc.off_keys);
    UNC_AS_GLOBAL SortKey *const skeys = reinterpret_cast<UNC_AS_GLOBAL SortKey *>(sb + A->sc.off_keys + A->sc.keys_cap * (uint32_t)sizeof(SortKey));
    if (lane < NKMER / 32) s_flags[lane] = s_w.flags_save[lane];
    for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) {
        const uint2 r = gld<uint2>(sb, chd_off + (i << PATH_SHIFT));        // 32-bit rows: start, end
        SortKey w;
        w.a = ((uint64_t)r.x << KEY_LEN_BITS) | (uint64_t)(r.y - r.x);
        w.b = gld<uint64_t>(sb, info_off + (i << 3));
        g_store(ukeys + i, w);
    }
    wave_sync();
    if (n <= 64) sort_regs<1>(ukeys, skeys, n, lane);
    else if (n <= 128) sort_regs<2>(ukeys, skeys, n, lane);
    else if (n <= 256) sort_regs<4>(ukeys, skeys, n, lane);
    else if (n <= 512) sort_regs<8>(ukeys, skeys, n, lane);
    else sort_hybrid(ukeys, skeys, n, lane);
    if (lane == 0) { s_w.kl = 0u; s_w.walked = 0u; s_w.mixed = 0u; }
    wave_sync();
}

// ---------------- F: remaining full-range sources, :605-624; the next parent list is complete after it ----------------
// sources_added_[k] for k = j*64 + lane lives in bit j of this lane's 16-bit field (two lanes per word).
// Pass 1 decides, in k-mer order, which k-mers get a full-range source and where (pure ALU + LDS);
// pass 2 fetches their ranges and writes the records, one lane per source, in one memory round trip.
template <bool NARROW>
static __device__ __noinline__ void phase_F(kargs_t A_, gptr_t sb_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    uint32_t *const s_list = reinterpret_cast<uint32_t *>(s_e);   // NKMER entries
    const uint32_t n = ctx_get(s_w.nchild), cur = ctx_get(s_w.cur);
    const uint32_t n_surv = n ? ctx_get(s_w.n_surv) : 0u, n_src = n ? ctx_get(s_w.n_src) : 0u;
    const uint32_t max_paths = A->sc.max_paths;
    const float source_prob = A->ix.thresholds[0];
    const uint32_t kvalid = ((const UNC_AS_GLOBAL uint16_t *)A->ix.kmer_valid)[lane];     // bit j: k-mer j*64+lane occurs in the reference
    const UNC_AS_GLOBAL uint64_t *const kmer_ranges = (const UNC_AS_GLOBAL uint64_t *)A->ix.kmer_ranges;
    const uint32_t chd_off = A->sc.off_paths + (cur ^ 1u) * (max_paths << PATH_SHIFT), nord_off = A->sc.off_order + (cur ^ 1u) * (max_paths << 2);
    uint32_t ent = n + n_src;
    {
        const uint32_t ent0 = ent;
        const uint32_t myflags = (s_flags[lane >> 1] >> ((lane & 1) << 4)) & 0xFFFFu;
        uint32_t newflags = 0;
#pragma unroll 4
        for (int j = 0; j < NKMER / WAVE; ++j) {
            const uint32_t k = (uint32_t)j * WAVE + (uint32_t)lane;
            const uint32_t fl = (myflags >> j) & 1u;
            const bool cond = !fl && s_probs[k] >= source_prob && ((kvalid >> j) & 1u);
            const uint64_t m = __ballot(cond);
            const uint32_t before = ent + (uint32_t)prefix_popc(m);
            const bool exec = before < max_paths;           // loop header: next_path != end
            const bool app = cond && exec;
            if (app) s_list[before - ent0] = k;
            if (!exec && fl) newflags |= 1u << j;           // flags survive only past the cut-off
            ent += (uint32_t)__popcll(__ballot(app));
        }
        const uint32_t other = (uint32_t)__shfl_xor((int)newflags, 1);
        wave_sync();
        if (!(lane & 1)) s_flags[lane >> 1] = newflags | (other << 16);
        for (uint32_t q = (uint32_t)lane; q < ent - ent0; q += WAVE) {
            const uint32_t k = s_list[q];
            write_source<NARROW>(sb + chd_off, ent0 + q, kmer_ranges[2 * k], kmer_ranges[2 * k + 1], k, s_probs[k]);
        }
    }
    wave_sync();
    const uint32_t nsrc_total = ent - n;
    for (uint32_t q = (uint32_t)lane; q < nsrc_total; q += WAVE) gst(sb, nord_off + ((n_surv + q) << 2), n + q);
    if (lane == 0) {
        s_w.n_parents = n_surv + nsrc_total; s_w.n_surv_par = n_surv; s_w.cur = cur ^ 1u;
        if (ent >= max_paths) s_w.notes |= UNC_NOTE_PATHS_FULL;      // next_path == next_paths_.end(): something may have been left out
    }
    wave_sync();
}

// ---------------- T + G: seeds -> SeedTracker, then the confidence test ----------------
// SA look-ups for all seeds in parallel, then SeedTracker::add_seed in the reference's order on the bucket grid;
// get_final / check_map_conf (seed_tracker.cpp:129-143,259-262).  Returns this lane's SA look-ups | LF steps << 32.
template <bool PROF>
static __device__ __noinline__ uint64_t phase_T(kargs_t A_, gptr_t sb_, int lane) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    const FmView ix = fm_view(A);
    const uint32_t n_seedp = ctx_get(s_w.n_seedp);
    const uint32_t seedp_off = A->sc.off_seedp, tasks_off = A->sc.off_tasks;
    const uint32_t p_min_map_len = A->P.min_map_len;
    const float p_min_mean_conf = A->P.min_mean_conf, p_min_top_conf = A->P.min_top_conf;
    PhaseClock<PROF> clk;
    uint32_t c_sa = 0, c_lf = 0;
    Tracker T;
    {
        const Tracker v = s_T;      // uniform values: back into scalar registers
        T.n = uniform32(v.n); T.n_lens = uniform32(v.n_lens); T.max1 = uniform32(v.max1); T.max2 = uniform32(v.max2);
        T.status = uniform32(v.status); T.n_alloc = uniform32(v.n_alloc);
        T.len_sum = __uint_as_float(uniform32(__float_as_uint(v.len_sum)));
        T.mm.ref_st = uniform64(v.mm.ref_st); T.mm.rstart = uniform64(v.mm.rstart); T.mm.rend = uniform64(v.mm.rend);
        T.mm.evt_st = uniform32(v.mm.evt_st); T.mm.evt_en = uniform32(v.mm.evt_en); T.mm.total_len = uniform32(v.mm.total_len);
    }
    const TrackerMem TM = tracker_mem(A, sb);
#if UNC_DBG_SEED
    if (lane < 4) s_dbgseed[lane] = 0ull;
    wave_sync();
#endif
    for (uint32_t sb0 = 0; sb0 < n_seedp && !T.status; sb0 += WAVE) {
        const uint32_t si = sb0 + (uint32_t)lane;
        SeedPath sp; sp.start = 0; sp.count = 0; sp.evt = 0; sp.ref_len = 0;
        if (si < n_seedp) sp = gld<SeedPath>(sb, seedp_off + si * (uint32_t)sizeof(SeedPath));
        uint32_t ttot;
        const uint32_t toff = excl_sum_bits<7>(sp.count, &ttot);
        // one task per FM row of a seed path: the row (40 bits) with the seed's event (16 bits) and move count on top,
        // so that the serial part below gets everything about 64 seeds from one coalesced load
        {
            const uint64_t tag = ((uint64_t)(sp.evt & 0xFFFFu) << 40) | ((uint64_t)sp.ref_len << 56);
            for (uint32_t j = 0; j < sp.count; ++j) gst(sb, tasks_off + ((toff + j) << 3), (sp.start + j) | tag);
        }
        wave_sync();
        for (uint32_t t0 = 0; t0 < ttot; t0 += WAVE) {
            const uint32_t ti = t0 + (uint32_t)lane;
            if (ti < ttot) {
                uint32_t lf;
                const uint64_t v = gld<uint64_t>(sb, tasks_off + (ti << 3));
                const uint64_t row = v & ((1ull << 40) - 1ull);
                const uint64_t sa = ix.sa_dense ? fm_sa_dense(ix, row, &lf) : fm_sa(ix, row, &lf);
                gst(sb, tasks_off + (ti << 3), (ix.seq_len - sa) | (v & ~((1ull << 40) - 1ull)));    // sa_end, mapper.cpp:678
                c_sa++;
                c_lf += lf;
            }
        }
        wave_sync();
        clk.end(5, lane);
        // SeedTracker::add_seed with the reference's outcome: task order = seed paths in list order, their rows ascending; one lane
        // per seed, seeds that cannot see each other together (map_tracker.h)
        for (uint32_t t0 = 0; t0 < ttot && !T.status; t0 += WAVE) {
            const uint32_t nt = ttot - t0 < WAVE ? ttot - t0 : WAVE;
            const uint64_t mine = (uint32_t)lane < nt ? gld<uint64_t>(sb, tasks_off + ((t0 + (uint32_t)lane) << 3)) : 0ull;
            add_seeds(T, TM, p_min_map_len, mine, nt, lane);
        }
        clk.end(6, lane);
    }
#if UNC_DBG_SEED
    wave_sync();
    if (PROF && lane == 0) {
        s_cyc[8] += s_dbgseed[0]; s_cyc[9] += s_dbgseed[1]; s_cyc[10] += s_dbgseed[2];
        if (s_dbgseed[3] > s_cyc[11]) s_cyc[11] = s_dbgseed[3];
    }
#endif
    // ---------------- G: SeedTracker::get_final + check_map_conf, :129-143,259-262 ----------------
    bool conf = false;
    if (T.mm.total_len >= p_min_map_len && T.n_lens >= 2) {
        const float mean_len = __fdiv_rn(T.len_sum, (float)T.n);
        const float second_len = (float)T.max2;
        const float ml = (float)T.mm.total_len;
        conf = (p_min_mean_conf > 0 && __fdiv_rn(ml, mean_len) >= p_min_mean_conf) ||
               (p_min_top_conf > 0 && __fdiv_rn(ml, second_len) >= p_min_top_conf);
    }
    wave_sync();
    if (lane == 0) { s_T = T; s_w.tstatus |= T.status; s_w.conf = conf ? 1u : 0u; }
    wave_sync();
    return (uint64_t)c_sa | ((uint64_t)c_lf << 32);
}

// k_map: the per-read path-forest search, one WAVEFRONT per read, persistent over a read queue.
// PROF: per-phase shader-clock counters (unc_mapper_last_phase_cycles); the plain instantiation carries none of it
// NARROW: the index allows 64-bit sort keys and 32-bit rows (DevIndex::key_len_bits > 0: E. coli, chr20): the children's keys
// leave phase E as sorted runs and are merged; the other instantiation (human-sized references) sorts 128-bit keys
template <bool PROF, bool NARROW>
__global__ __launch_bounds__(64, UNC_LB) void k_map(MapArgs Aval) {
    const kargs_t A = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();     // the argument block itself: Aval is its only member
    (void)Aval;
    const int lane = lane_id();
    const uint64_t wave_t0 = (uint64_t)wall_clock64();
    const bool sliced = A->sched.ctl != nullptr;      // batch mode with time slices (see DevSched)
    const uint32_t resume = A->resume;

    for (;;) {
        // ---------------- fetch or resume a read ----------------
        uint32_t r = 0, event_i, n_parents, cur;
        uint32_t n_surv_par = 0;                     // the first n_surv_par parents are the last walk's survivors (sorted)
        uint32_t tstatus = 0;                        // UNC_READ_* bits (mirror of the tracker's status)
        uint32_t notes = 0;                          // UNC_NOTE_* bits
        Tracker T;
        uint64_t c_nbr = 0, c_sa = 0, c_lf = 0;      // per-lane partial counters
        uint64_t t_start = (uint64_t)wall_clock64();  // when the read was taken up (a parked read brings its own)
        uint32_t slot = (resume && A->slot_map) ? A->slot_map[blockIdx.x] : blockIdx.x;
        bool restore = false;
        if (sliced) {
            // task = a new read in a free slot while both exist, else the longest-parked read
            uint32_t kind = 0, tslot = 0, tread = 0;
            if (lane == 0) {
                SchedCtl *const sc = A->sched.ctl;
                const uint32_t cap_mask = A->sched.cap_mask, n_reads = A->rd.n_reads;
                const uint32_t part = sched_part(A->sched.n_parts);
                SchedCell *const free_cells = A->sched.free_cells + (size_t)part * (cap_mask + 1u);
                SchedCell *const park_cells = A->sched.park_cells + (size_t)part * (cap_mask + 1u);
                SchedQueue *const freeq = &sc->freeq[part], *const parkq = &sc->parkq[part];
                for (int tries = 0; tries < 4096 && !kind; ++tries) {
                    const bool more = ld_rlx(&sc->next_read) < n_reads;
                    // admission control: while the pool of cluster nodes is below an eighth, reads in flight go first (they
                    // give their chunks back when they end); a new read is only started when none of them is waiting
                    const bool low = (int32_t)ld_rlx(reinterpret_cast<const uint32_t *>(&A->pool.q->avail)) < (int32_t)(A->pool.n_chunks >> 3);
                    if (more && (!low || tries >= 8)) {
                        const uint32_t fs = sched_pop(freeq, free_cells, cap_mask);
                        if (fs != SCHED_EMPTY) {
                            const uint32_t t = atomicAdd(&sc->next_read, 1u);
                            if (t < n_reads) { kind = 1; tslot = fs; tread = t; }
                            else sched_push(freeq, free_cells, cap_mask, fs);
                        }
                    }
                    if (!kind) {
                        const uint32_t ps = sched_pop(parkq, park_cells, cap_mask);
                        if (ps != SCHED_EMPTY) { kind = 2; tslot = ps; }
                    }
                    // reads left but every slot is in another wavefront's hands right now: wait for one to come back
                    if (!kind) { if (!more) break; __builtin_amdgcn_s_sleep(32); }
                }
            }
            kind = bcast32(kind, 0); tslot = bcast32(tslot, 0); tread = bcast32(tread, 0);
            if (!kind) break;
            slot = tslot; r = tread; restore = kind == 2;
            // the slot was parked (or freed) by another wavefront: of this XCD when the rings are per XCD -- the L2 is theirs in common,
            // only this CU's vector L1 may hold lines of the slot from an earlier slice --, of any XCD otherwise
            if (A->sched.n_parts > 1u) l1_invalidate();
            else __threadfence();
        }

        // everything this read owns hangs off ONE uniform pointer; regions are 32-bit byte offsets (DevScratch)
        const gptr_t sb = (gptr_t)A->sc.base + (size_t)slot * A->sc.slot_bytes;
        UNC_AS_GLOBAL SlotState *const st = reinterpret_cast<UNC_AS_GLOBAL SlotState *>(sb + A->sc.off_state);
        if constexpr (PROF) { if (lane < 12) s_cyc[lane] = 0; }

        const bool fresh = resume && A->rd.new_read && A->rd.new_read[blockIdx.x];   // first chunk of a read
        if ((resume && !fresh) || restore) {
            r = restore ? uniform32(st->read_idx) : blockIdx.x; event_i = st->event_i; n_parents = st->n_parents; cur = st->cur;
            n_surv_par = uniform32(st->n_surv);
            notes = uniform32(st->notes);
            T.n = st->n_clusters; T.n_lens = st->n_lens; T.max1 = st->len_max1; T.max2 = st->len_max2;
            T.status = st->status; T.len_sum = st->len_sum; T.n_alloc = st->n_alloc;
            T.mm.ref_st = st->max_map.ref_st; T.mm.rstart = st->max_map.rstart; T.mm.rend = st->max_map.rend;
            T.mm.evt_st = st->max_map.evt_st; T.mm.evt_en = st->max_map.evt_en; T.mm.total_len = st->max_map.total_len;
            if (lane == 0) { c_nbr = st->n_nbr; c_sa = st->n_sa; c_lf = st->n_lf; }
            if (restore) t_start = st->t_start;
            if (lane < NKMER / 32) s_flags[lane] = st->sources_added[lane];
            event_i = uniform32(event_i); n_parents = uniform32(n_parents); cur = uniform32(cur);
            if (restore) { if constexpr (PROF) { if (lane < 12) s_cyc[lane] = st->cyc[lane]; } }
            else if (uniform32(st->done)) break;
        } else if (resume) {
            r = blockIdx.x;
            {   // a new read takes the channel over: what the previous one still holds goes back to the pool
                Tracker old;
                old.n_alloc = uniform32(st->n_alloc);
                tracker_release(old, tracker_mem(A, sb), lane);
            }
            tracker_clear_heads(tracker_mem(A, sb), lane);
            event_i = 0; n_parents = 0; cur = 0;
            T.n = 0; T.n_lens = 0; T.max1 = 0; T.max2 = 0; T.status = 0; T.len_sum = 0.0f; T.n_alloc = 0;
            T.mm.ref_st = 0; T.mm.rstart = 1; T.mm.rend = 0; T.mm.evt_st = 1; T.mm.evt_en = 0; T.mm.total_len = 0;
            // sources_added_ is NOT cleared by Mapper::new_read / reset (mapper.cpp:88,219-246,612-623): a channel is one Mapper, and the
            // flags a read leaves behind when its path buffer was full (the else-branch that clears them was not reached) are seen by
            // the channel's next read.  Deterministic per channel, so the chunked path reproduces it (the slot starts zeroed).
            if (lane < NKMER / 32) s_flags[lane] = st->sources_added[lane];
        } else {
            if (!sliced) {
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(A->next_read, 1u);
                r = bcast32(t, 0);
                if (r >= A->rd.n_reads) break;
            }
            if (A->read_list) r = uniform32(A->read_list[r]);
            tracker_clear_heads(tracker_mem(A, sb), lane);     // every bucket of the seed-cluster grid empty
            event_i = 0; n_parents = 0; cur = 0;
            T.n = 0; T.n_lens = 0; T.max1 = 0; T.max2 = 0; T.status = 0; T.len_sum = 0.0f; T.n_alloc = 0;
            T.mm.ref_st = 0; T.mm.rstart = 1; T.mm.rend = 0; T.mm.evt_st = 1; T.mm.evt_en = 0; T.mm.total_len = 0;  // NULL_ALN
            // sources_added_: clear -- every read as a fresh Mapper maps it -- unless the caller hands in what the Mapper's previous
            // read left behind (the `-t 1` order of unc_mapper_set_read_order; mapper.cpp:88,612-623)
            if (lane < NKMER / 32) s_flags[lane] = A->flags_in ? A->flags_in[(size_t)r * (NKMER / 32) + (uint32_t)lane] : 0u;
        }
        tstatus = uniform32(T.status);
        wave_sync();
        if (lane == 0) {
            s_T = T;
            s_w.event_i = event_i; s_w.n_parents = n_parents; s_w.cur = cur; s_w.n_surv_par = n_surv_par; s_w.tstatus = tstatus;
            s_w.notes = notes;
        }
        wave_sync();
        const uint32_t n_events = A->rd.info[r].n_events;
        const float scale = A->rd.info[r].scale, shift = A->rd.info[r].shift;
        const UNC_AS_GLOBAL float *const means = (const UNC_AS_GLOBAL float *)A->rd.means + A->rd.moff[r];
        const uint32_t ring_mod = A->rd.ring_mod, ring_r0 = ring_mod ? A->rd.ring0[r] : 0u;
        const uint32_t max_events = A->P.max_events, max_steps = A->max_steps;
#define MEAN_AT(e) means[ring_mod ? (ring_r0 + (e)) % ring_mod : (e)]

        uint32_t done = 0, steps = 0;
        float next_mean = event_i < n_events ? MEAN_AT(event_i) : 0.0f;
        while (!done && steps < max_steps) {
            // map_next prologue, mapper.cpp:434-437 (norm_.empty() <=> every event popped)
            if (ring_mod && event_i >= n_events && event_i < max_events && !tstatus) break;   // chunk mapped: park
            if (event_i >= n_events || event_i >= max_events || tstatus) { done = 2; break; }
            ++steps;

            PhaseClock<PROF> clk;
            const float level = __fadd_rn(__fmul_rn(scale, next_mean), shift);   // Normalizer::at
            if (event_i + 1 < n_events) next_mean = MEAN_AT(event_i + 1);
            if (lane == 0) gst(sb, A->sc.off_levels + ((event_i & (LEVEL_RING - 1u)) << 2), level);   // (read back 22 events on: PathRec)
            phase_P(A, level, lane);
            clk.end(0, lane);
            c_nbr += phase_E<PROF, NARROW>(A, sb, lane);
            clk.reset();       // (phase E keeps its own clock: parts 8..11, the rest in 1)
            if (ctx_get(s_w.nchild) > 0) {
                phase_S<NARROW>(A, sb, lane);
                clk.end(2, lane);
                if (!ctx_get(s_w.walked)) phase_W<NARROW>(A, sb, lane);
                if constexpr (NARROW) {
                    if (ctx_get(s_w.mixed)) { phase_S_wide_redo(A, sb, lane); phase_W<NARROW>(A, sb, lane); }
                }
                clk.end(3, lane);
            }
            phase_F<NARROW>(A, sb, lane);
            clk.end(4, lane);
            // T: (nothing to add: the tracker, and with it the confidence test, is where the last event left it)
            bool conf = false;
            if (ctx_get(s_w.n_seedp) > 0 && !ctx_get(s_w.tstatus)) {
                const uint64_t c = phase_T<PROF>(A, sb, lane);
                clk.reset();   // (5 and 6 inside)
                c_sa += (uint32_t)c; c_lf += c >> 32;
                conf = ctx_get(s_w.conf) != 0;
            }
            tstatus = ctx_get(s_w.tstatus);
            if (tstatus) { done = 2; }
            else if (conf) { done = 1; }
            else {
                event_i++;
                CTX_SET(event_i, event_i);
                wave_sync();
            }
            clk.end(7, lane);
        }
        n_parents = ctx_get(s_w.n_parents); cur = ctx_get(s_w.cur); n_surv_par = ctx_get(s_w.n_surv_par);
        notes = ctx_get(s_w.notes);
        // sources_added_ flags that are still set when the read is decided: the reference's Mapper takes them into its next read
        if (done && __any(lane < NKMER / 32 && s_flags[lane < NKMER / 32 ? lane : 0] != 0u)) notes |= UNC_NOTE_FLAGS_LEFT;

        // ---------------- publish / park ----------------
        const uint64_t t_nbr = wave_sum64(c_nbr), t_sa = wave_sum64(c_sa), t_lf = wave_sum64(c_lf);
        T = s_T;
        T.status |= tstatus;
        T.n_alloc = uniform32(T.n_alloc);
        if (done && lane == 0) {
            DevResult res;
            res.done = done; res.status = T.status; res.event_i = event_i; res.notes = notes;
            res.cluster = T.mm;
            res.n_nbr = t_nbr; res.n_sa = t_sa; res.n_lf = t_lf;
            res.ticks = resume ? 0ull : (uint64_t)wall_clock64() - t_start;
            // PROF: where the read was decided -- XCC_ID (bits 0-3) | HW_ID's low 16 bits (wave, SIMD, CU, shader array, shader engine) << 8 | the
            // workgroup's number (low 8 bits) << 24: (XCC_ID - workgroup) mod 8 = where the dispatcher started this launch's round over the XCDs
            res.pad2 = PROF ? (uint32_t)__builtin_amdgcn_s_getreg(6164) | ((uint32_t)__builtin_amdgcn_s_getreg(4 | (15 << 11)) << 8) | ((blockIdx.x & 0xFFu) << 24) : 0u;
            for (int i = 0; i < 12; ++i) res.cyc[i] = PROF ? s_cyc[i] : 0ull;
            g_store((UNC_AS_GLOBAL DevResult *)A->results + r, res);
        }
        // the read is decided: its nodes go back to the pool at once -- in batch mode and in chunked (realtime) mode, where an idle
        // channel would otherwise pin them until its next read arrives; the step-wise trace (resume without a ring) keeps them
        // readable for unc_trace_clusters
        if (done && !resume && A->flags_out && lane < NKMER / 32) A->flags_out[(size_t)r * (NKMER / 32) + (uint32_t)lane] = s_flags[lane];
        if (done && (!resume || A->rd.ring_mod)) tracker_release(T, tracker_mem(A, sb), lane);
        if (resume || !done) {
            if (lane == 0) {
                st->read_idx = r; st->event_i = event_i; st->n_parents = n_parents; st->cur = cur; st->done = done;
                st->n_surv = n_surv_par;
                st->status = T.status; st->n_clusters = T.n; st->n_lens = T.n_lens;
                st->len_max1 = T.max1; st->len_max2 = T.max2; st->len_sum = T.len_sum;
                st->max_map.ref_st = T.mm.ref_st; st->max_map.rstart = T.mm.rstart; st->max_map.rend = T.mm.rend;
                st->max_map.evt_st = T.mm.evt_st; st->max_map.evt_en = T.mm.evt_en; st->max_map.total_len = T.mm.total_len;
                st->notes = notes; st->n_alloc = T.n_alloc;        // (a decided read's final notes stay for the chunked path's host side; a new read starts from 0)
                st->n_nbr = t_nbr; st->n_sa = t_sa; st->n_lf = t_lf; st->t_start = t_start;
                if constexpr (PROF) {
                    if (sliced) { for (int i = 0; i < 12; ++i) st->cyc[i] = s_cyc[i]; }
                    else if (resume) { for (int i = 0; i < 12; ++i) st->cyc[i] = (fresh ? 0ull : st->cyc[i]) + s_cyc[i]; }   // chunked mode: summed over a read's launches
                }
            }
            if (lane < NKMER / 32) st->sources_added[lane] = s_flags[lane];
            if (!sliced) break;
        }
        if (sliced) {
            // hand the slot back: to the free ring when the read is finished, else to the end of the parked ring
            // (per-XCD rings: the next wavefront to hold the slot shares this one's L2 -- the stores only have to have arrived there)
            if (A->sched.n_parts > 1u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            else __threadfence();
            if (lane == 0) {
                SchedCtl *const sc = A->sched.ctl;
                const uint32_t cap = A->sched.cap_mask + 1u, part = sched_part_of_slot(A->sched.n_slots, A->sched.n_parts, slot);
                if (done) sched_push(&sc->freeq[part], A->sched.free_cells + (size_t)part * cap, A->sched.cap_mask, slot);
                else sched_push(&sc->parkq[part], A->sched.park_cells + (size_t)part * cap, A->sched.cap_mask, slot);
            }
        }
        wave_sync();
    }
    if (A->wave_ticks && lane == 0) atomicAdd(A->wave_ticks, (unsigned long long)((uint64_t)wall_clock64() - wave_t0));
}

// =====================================================================================================================
// SEVERAL WAVEFRONTS PER READ (the chunked / realtime path).  A flow cell has 512 channels: one wavefront per channel
// leaves three quarters of the chip's SIMDs idle, and a lone wavefront's event is bound by its own chain of dependent
// instructions, not by the memory system.  k_map_team gives a channel a TEAM of W wavefronts (one workgroup): the match
// probabilities and the extension of the parents -- 46 % of a lone wavefront's event -- are split over the team, the
// rest of the event (sort / walk / sources / seeds) is the leader's, on the functions of the one-wavefront kernel.
//
// Phase E in a team: the passes of 64 parents are dealt round by round, wave w taking pass r * W + w.  What a child needs from
// the passes before its own is a handful of COUNTS -- children, keys filed per run, dead-end seeds -- so a round is: every wave
// works its pass up to the child slots (parents, candidates, FM, slots) and publishes the uncut counts; one barrier; every wave
// adds up what the waves before it counted and writes its children where the one-wavefront kernel would have put them.  The
// max_paths cut-off only ever removes a SUFFIX of the creation order: a wave whose pass starts past the cut writes nothing, the
// wave that straddles it cuts as the one-wavefront kernel does, and the counts of the waves before it are exact as published.
// =====================================================================================================================
constexpr int TEAM_MAX = 8;
struct TeamCnt {
    uint32_t chtot, m[5], xtot, ecnt;      // children of the pass, keys per run of sorted survivors' children, children of sources, dead-end seeds
    uint32_t first_is_surv, pad0;
    uint64_t first_pk, last_pk;            // order check of the sorted survivors across passes
};
__shared__ TeamCnt s_team_cnt[2][TEAM_MAX];
__shared__ uint32_t s_team_flags;                  // over the waves of an event: 2 parents unsorted, 4 seed list overflow
__shared__ unsigned long long s_team_nbr;          // get_neighbor calls counted by the followers
__shared__ uint32_t s_team_ctl;                    // leader -> followers after an event: 0 go on, 1 / 2 read decided, 3 chunk mapped (park)
__shared__ __attribute__((aligned(16))) uint64_t s_e_x[TEAM_MAX - 1][S_E_WORDS];   // the followers' staging for phase E

__device__ __forceinline__ void team_barrier() { __syncthreads(); }

template <int W>
static __device__ __noinline__ void phase_P_team(kargs_t A_, float level, int lane, int wave) {
    const kargs_t A = uniform_ptr(A_);
    const UNC_AS_GLOBAL float4 *const model4 = (const UNC_AS_GLOBAL float4 *)A->ix.model4;
#pragma unroll 4
    for (int j = wave; j < NKMER / WAVE; j += W) {
        const uint32_t k = (uint32_t)j * WAVE + (uint32_t)lane;
        const float4 row = g_load(model4 + k);
        const float mu = row.x, v2 = row.y, ld = row.z;
        const float d = __fsub_rn(level, mu);
        const double q = -((double)d * (double)d) / (double)v2;
        s_probs[k] = (float)(q - (double)ld);
    }
}

// Running totals every wave of the team keeps (identically) while the rounds go by
struct TeamTotals { uint32_t nchild, n_seedp, scnt[5], scntx; uint64_t run_last; };

template <bool PROF, bool NARROW, int W>
static __device__ __noinline__ uint32_t phase_E_team(kargs_t A_, gptr_t sb_, int lane, int wave, TeamTotals &TT) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    using Row = std::conditional_t<NARROW, uint32_t, uint64_t>;
    constexpr int RES_BITS = 30;
    constexpr uint32_t PSTRIDE = (uint32_t)WAVE * (uint32_t)W;       // parents of one round
    uint64_t *const stg = wave == 0 ? s_e : s_e_x[wave > 0 ? wave - 1 : 0];
    uint64_t *const s_res = stg;                      // [parent lane << 2 | base]
    Row *const s_pstart = reinterpret_cast<Row *>(stg + CAND_MAX), *const s_pend = s_pstart + WAVE;
    uint64_t *const s_phist = reinterpret_cast<uint64_t *>(s_pend + WAVE);
    float *const s_plast = reinterpret_cast<float *>(s_phist + WAVE), *const s_psubc = s_plast + WAVE;
    uint32_t *const s_pmoves = reinterpret_cast<uint32_t *>(s_psubc + WAVE), *const s_pmeta = s_pmoves + WAVE;
    uint16_t *const s_cdesc = reinterpret_cast<uint16_t *>(s_pmeta + WAVE);         // per child: parent lane | type << 6 | child of a source << 9
    uint16_t *const s_ckpos = s_cdesc + CHILD_MAX;                                   // narrow keys: a child's position among its pass's keys of its run
    uint16_t *const s_cand = s_cdesc;                                                // (dead before the descriptors are written)

    const FmView ix = fm_view(A);
    const uint32_t max_paths = A->sc.max_paths, max_seed_paths = A->sc.max_seed_paths;
    const uint32_t event_i = ctx_get(s_w.event_i), n_parents = ctx_get(s_w.n_parents), cur = ctx_get(s_w.cur), n_surv_par = ctx_get(s_w.n_surv_par);
    const float thr_lane = A->ix.thresholds[lane];
    const uint32_t klb = NARROW ? A->ix.key_len_bits : 0u;
    const uint32_t p_max_consec_stay = A->P.max_consec_stay, p_seed_len = A->P.seed_len, p_max_rep_copy = A->P.max_rep_copy, p_min_rep_len = A->P.min_rep_len;
    const float p_min_seed_prob = A->P.min_seed_prob, p_max_stay = __fmul_rn(A->P.max_stay_frac, (float)A->P.seed_len);
    const uint32_t par_off = A->sc.off_paths + cur * (max_paths << PATH_SHIFT), chd_off = A->sc.off_paths + (cur ^ 1u) * (max_paths << PATH_SHIFT);
    const uint32_t pord_off = A->sc.off_order + cur * (max_paths << 2);
    const uint32_t seedp_off = A->sc.off_seedp, ukeys_off = A->sc.off_keys, str_off = A->sc.off_streams, info_off = A->sc.off_info;
    const uint32_t run_bytes = max_paths << 3;
    const float level_old = event_i >= (uint32_t)SEED_LEN ? gld<float>(sb, A->sc.off_levels + (((event_i - (uint32_t)SEED_LEN) & (LEVEL_RING - 1u)) << 2)) : 0.0f;
    const UNC_AS_GLOBAL float4 *const model4 = (const UNC_AS_GLOBAL float4 *)A->ix.model4;

    uint32_t c_nbr = 0;
    bool par_bad = false, seed_over = false;
    TT.nchild = 0; TT.n_seedp = 0; TT.scntx = 0; TT.run_last = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) TT.scnt[t] = 0;

    // this wave's passes: parents wave * 64 + lane of every round; record headers are fetched one round ahead
    const uint32_t pi0 = (uint32_t)wave * WAVE + (uint32_t)lane;
    uint32_t phys_cur = pi0 < n_parents ? gld<uint32_t>(sb, pord_off + (pi0 << 2)) : 0u;
    uint32_t phys_nxt = pi0 + PSTRIDE < n_parents ? gld<uint32_t>(sb, pord_off + ((pi0 + PSTRIDE) << 2)) : 0u;
    using Q4 = std::conditional_t<NARROW, u32x4_t, uint4>;     // (one register tuple for mem_retire; the wide instantiation as it was)
    Q4 q0c = Q4{1u, 0u, 1u, 0u}, q1c = Q4{0u, 0u, 0u, 0u};
    if (pi0 < n_parents) {
        q0c = gld<Q4>(sb, par_off + (phys_cur << PATH_SHIFT)); q1c = gld<Q4>(sb, par_off + (phys_cur << PATH_SHIFT) + 16u);
    }
    // (where the waits of a round sit: see phase_E and wave_prims.h, mem_retire)
    if constexpr (NARROW) { mem_retire(phys_nxt); mem_retire(q0c); mem_retire(q1c); }
    PhaseClock<PROF> clk;       // (the leader's view of a round: 8 parents + candidates, 9 FM, 10 slots, 11 the wait at the exchange, 1 children)
    for (uint32_t rbase = 0, rnd = 0; rbase < n_parents && TT.nchild < max_paths; rbase += PSTRIDE, ++rnd) {
        const uint32_t base = rbase + (uint32_t)wave * WAVE;
        const uint32_t pi = base + (uint32_t)lane;
        const bool have = pi < n_parents;
        const Q4 q0 = q0c, q1 = q1c;
        phys_cur = phys_nxt;
        if (pi + PSTRIDE < n_parents) {
            q0c = gld<Q4>(sb, par_off + (phys_nxt << PATH_SHIFT)); q1c = gld<Q4>(sb, par_off + (phys_nxt << PATH_SHIFT) + 16u);
        }
        if (pi + 2 * PSTRIDE < n_parents) phys_nxt = gld<uint32_t>(sb, pord_off + ((pi + 2 * PSTRIDE) << 2));
        uint32_t pmoves = 0, pmeta = 0;
        Row pstart = 1, pend = 1;
        float plast = 0.0f, psub = 0.0f;
        uint64_t phist = 0;
        if (have) {
            if constexpr (NARROW) { pstart = q0.x; pend = q0.y; }
            else { const uint64_t r = ((uint64_t)q0.y << 32) | q0.x; pstart = r >> KEY_LEN_BITS; pend = pstart + (r & KEY_LEN_MASK); }
            pmoves = q0.z; pmeta = q0.w;
            plast = __uint_as_float(q1.x); psub = __uint_as_float(q1.y); phist = ((uint64_t)q1.w << 32) | q1.z;
        }
        const uint32_t plen = (pmeta >> META_LEN_SHIFT) & 31u;
        const bool pfull = plen == (uint32_t)SEED_LEN;
        const uint32_t okmer = (uint32_t)phist & KMASK;
        f32x4_t orow = {0.f, 1.f, 0.f, 0.f};
        if constexpr (NARROW) orow = g_load(reinterpret_cast<const UNC_AS_GLOBAL f32x4_t *>(model4) + (pfull ? okmer : 0u));
        else if (pfull) orow = g_load(reinterpret_cast<const UNC_AS_GLOBAL f32x4_t *>(model4) + okmer);      // (wide keys: as rounds 1-4, see the note above the loop)
        const Row plen_fm = pend - pstart + 1;
        // the merge of the children's keys relies on the survivors being in ascending (start, length) order: inside the pass here,
        // across passes after the exchange
        uint64_t first_pk = 0, last_pk = ~0ull;
        if constexpr (NARROW) {
            const uint64_t pk = pi < n_surv_par ? ((uint64_t)pstart << 32) | (uint32_t)(pend - pstart) : ~0ull;
            const uint64_t pp = (uint64_t)__shfl_up((unsigned long long)pk, 1);
            if (__any(lane > 0 && pi < n_surv_par && !(pk > pp))) par_bad = true;
            first_pk = bcast64(pk, 0); last_pk = bcast64(pk, WAVE - 1);
        }
        int thr_bin;
        if constexpr (NARROW) thr_bin = 32 + __clz((int)plen_fm); else thr_bin = __clzll((long long)plen_fm);
        const float thr = __shfl(thr_lane, thr_bin);
        const uint32_t kmer = pmeta & META_KMER_MASK;
        const uint32_t stays = (pmeta >> META_STAY_SHIFT) & 255u;
        const bool stay_ok = have && stays < p_max_consec_stay && s_probs[kmer] >= thr;
        const float4 np = *reinterpret_cast<const float4 *>(&s_probs[(kmer << 2) & KMASK]);
        uint32_t mask = (!(np.x < thr) ? 1u : 0u) | (!(np.y < thr) ? 2u : 0u) | (!(np.z < thr) ? 4u : 0u) | (!(np.w < thr) ? 8u : 0u);
        if (!have) mask = 0;
        s_pstart[lane] = pstart; s_pend[lane] = pend; s_pmoves[lane] = pmoves; s_pmeta[lane] = pmeta; s_plast[lane] = plast;
        const uint32_t ncand = (uint32_t)__popc(mask);
        uint32_t ctot;
        const uint32_t coff = excl_sum_bits<3>(ncand, &ctot);
        {
            uint32_t w = coff;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b)
                if (mask & (1u << b)) s_cand[w++] = (uint16_t)(((uint32_t)lane << 2) | b);
        }
        wave_sync();
        if (wave == 0) clk.end(8, lane);
        for (uint32_t c0 = 0; c0 < ctot; c0 += WAVE) {
            const uint32_t ci = c0 + (uint32_t)lane;
            if (ci < ctot) {
                const uint32_t cd = s_cand[ci];
                if constexpr (NARROW) {
                    uint32_t ns, ne;
                    fm32_get_neighbor(ix, s_pstart[cd >> 2], s_pend[cd >> 2], cd & 3u, &ns, &ne);
                    s_res[cd] = ns <= ne ? ((uint64_t)(ne - ns + 1u) << 32) | ns : 0ull;
                } else {
                    uint64_t ns, ne;
                    fm_get_neighbor(ix, s_pstart[cd >> 2], s_pend[cd >> 2], cd & 3u, &ns, &ne);
                    s_res[cd] = ns <= ne ? (ns << RES_BITS) | (ne - ns + 1) : 0ull;
                }
            }
        }
        wave_sync();
        if constexpr (NARROW) { mem_retire(q0c); mem_retire(q1c); mem_retire(phys_nxt); mem_retire(orow); }      // (behind the FM look-ups' wait: arrived)
        if (wave == 0) clk.end(9, lane);
        // ---- the pass's children as if nothing were cut off: who, where in the pass, where among the pass's keys of its run
        const uint32_t rb = (uint32_t)lane << 2;
        uint32_t vmask = (s_res[rb] != 0 ? 1u : 0u) | (s_res[rb + 1] != 0 ? 2u : 0u) | (s_res[rb + 2] != 0 ? 4u : 0u) | (s_res[rb + 3] != 0 ? 8u : 0u);
        vmask &= mask;
        const uint32_t nch = (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask);
        uint32_t chtot;
        const uint32_t choff = excl_sum_bits<3>(nch, &chtot);
        wave_sync();             // (the candidate list is dead: its space takes the descriptors)
        const bool is_src = pi >= n_surv_par;
        uint32_t mcount[5] = {0, 0, 0, 0, 0}, xtot = 0, src_cpos0 = chtot;
        uint64_t tm[5] = {0, 0, 0, 0, 0};          // which parents of the pass have a child in each run of sorted survivors' children
        {
            uint32_t cpos[5];
            bool ex[5];
            ex[0] = stay_ok; cpos[0] = choff;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                ex[b + 1] = (vmask >> b) & 1u;
                cpos[b + 1] = choff + (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask & ((1u << b) - 1u));
            }
            uint32_t kp[5] = {0, 0, 0, 0, 0};
            if (base < n_surv_par) {
#pragma unroll
                for (uint32_t t = 0; t < 5; ++t) {
                    const uint64_t mt = __ballot(ex[t] && !is_src);
                    kp[t] = (uint32_t)prefix_popc(mt);
                    mcount[t] = (uint32_t)__popcll(mt);
                    tm[t] = mt;
                }
            }
            if (base + WAVE > n_surv_par) {
                const uint32_t nfit = is_src ? (ex[0] ? 1u : 0u) + (ex[1] ? 1u : 0u) + (ex[2] ? 1u : 0u) + (ex[3] ? 1u : 0u) + (ex[4] ? 1u : 0u) : 0u;
                uint32_t xo = excl_sum_bits<3>(nfit, &xtot);
                if (is_src) {
#pragma unroll
                    for (uint32_t t = 0; t < 5; ++t) { kp[t] = xo; if (ex[t]) ++xo; }
                }
                const uint64_t srcm = __ballot(is_src);
                src_cpos0 = srcm ? lane_get32(choff, (uint32_t)__ffsll((unsigned long long)srcm) - 1u) : chtot;
            }
#pragma unroll
            for (uint32_t t = 0; t < 5; ++t)
                if (ex[t]) {
                    s_cdesc[cpos[t]] = (uint16_t)((uint32_t)lane | (t << 6) | (is_src ? 1u << 9 : 0u));
                    if constexpr (NARROW) s_ckpos[cpos[t]] = (uint16_t)kp[t];
                }
        }
        {
            float subc = psub;
            uint64_t hist = phist;
            if (pfull) {
                const float d = __fsub_rn(level_old, orow.x);
                const double q = -((double)d * (double)d) / (double)orow.y;
                subc = __fadd_rn(psub, (float)(q - (double)orow.z));
                if ((pmoves >> (SEED_LEN - 2)) & 1u) {
                    const uint32_t cnt = (uint32_t)__popc(pmoves & ((1u << (SEED_LEN - 1)) - 1u));
                    const uint32_t pos = 2u * (cnt - 1u);
                    const uint64_t fifo = hist >> 10;
                    const uint32_t nk = ((okmer << 2) & KMASK) | ((uint32_t)(fifo >> pos) & 3u);
                    hist = (uint64_t)nk | ((fifo & ~(3ull << pos)) << 10);
                }
            }
            s_psubc[lane] = subc; s_phist[lane] = hist;
        }
        // dead ends that would become seeds if their parent is reached (update_seeds(prev_path, true), :513-519 / is_seed_valid :842-863)
        bool ecand = false;
        uint32_t e_count = 0, e_mc = 0;
        if (have && nch == 0 && !(pmeta & META_SA_CHECKED)) {
            const uint32_t mc = (uint32_t)__popc(pmoves);
            const float pprob = __fdiv_rn(__fsub_rn(plast, psub), (float)SEED_LEN);
            const bool base_ok = plen == p_seed_len && pprob >= p_min_seed_prob;
            const bool uniq = plen_fm == 1 && (pmoves & 1u) == 1u && (float)(plen - mc) <= p_max_stay;
            const bool rep = plen_fm <= (Row)p_max_rep_copy && mc >= p_min_rep_len;
            ecand = base_ok && (uniq || rep);
            e_count = (uint32_t)plen_fm;
            e_mc = mc;
        }
        const uint32_t ecnt = (uint32_t)__popcll(__ballot(ecand));
        if (lane == 0) {
            TeamCnt c;
            c.chtot = chtot; c.xtot = xtot; c.ecnt = ecnt; c.first_is_surv = base < n_surv_par ? 1u : 0u; c.pad0 = 0;
#pragma unroll
            for (int t = 0; t < 5; ++t) c.m[t] = mcount[t];
            c.first_pk = first_pk; c.last_pk = last_pk;
            s_team_cnt[rnd & 1u][wave] = c;
        }
        if (wave == 0) clk.end(10, lane);
        team_barrier();
        if (wave == 0) clk.end(11, lane);
        // ---- what the waves before this one counted; the round's totals
        // (lane l < W looks at wave l's counts, packed three words wide -- no field can overflow into its neighbour: 8 passes hold at
        // most 2 560 children and 512 keys per run -- and an inclusive prefix over those lanes gives every wave its bases and the totals)
        uint32_t b_child = TT.nchild, b_seed = TT.n_seedp, b_x = TT.scntx, b_m[5];
        uint32_t r_child, r_seed, r_x, r_m[5];
        uint64_t prev_last = TT.run_last, round_last;
        {
            const TeamCnt c = s_team_cnt[rnd & 1u][(uint32_t)lane < (uint32_t)W ? lane : 0];
            const bool mine = (uint32_t)lane < (uint32_t)W;
            uint32_t q0 = mine ? (c.chtot | (c.xtot << 16)) : 0u;
            uint32_t q1 = mine ? (c.m[0] | (c.m[1] << 10) | (c.m[2] << 20)) : 0u;
            uint32_t q2 = mine ? (c.m[3] | (c.m[4] << 10) | (c.ecnt << 20)) : 0u;
#pragma unroll
            for (int d = 1; d < W; d <<= 1) {
                const uint32_t y0 = (uint32_t)__shfl_up((int)q0, d), y1 = (uint32_t)__shfl_up((int)q1, d), y2 = (uint32_t)__shfl_up((int)q2, d);
                if (lane >= d) { q0 += y0; q1 += y1; q2 += y2; }
            }
            const uint32_t t0 = lane_get32(q0, (uint32_t)W - 1u), t1 = lane_get32(q1, (uint32_t)W - 1u), t2 = lane_get32(q2, (uint32_t)W - 1u);
            r_child = t0 & 0xFFFFu; r_x = t0 >> 16;
            r_m[0] = t1 & 1023u; r_m[1] = (t1 >> 10) & 1023u; r_m[2] = t1 >> 20;
            r_m[3] = t2 & 1023u; r_m[4] = (t2 >> 10) & 1023u; r_seed = t2 >> 20;
            round_last = lane_get64(c.last_pk, (uint32_t)W - 1u);
#pragma unroll
            for (int t = 0; t < 5; ++t) b_m[t] = TT.scnt[t];
            if (wave > 0) {
                const uint32_t e0 = lane_get32(q0, (uint32_t)wave - 1u), e1 = lane_get32(q1, (uint32_t)wave - 1u), e2 = lane_get32(q2, (uint32_t)wave - 1u);
                b_child += e0 & 0xFFFFu; b_x += e0 >> 16;
                b_m[0] += e1 & 1023u; b_m[1] += (e1 >> 10) & 1023u; b_m[2] += e1 >> 20;
                b_m[3] += e2 & 1023u; b_m[4] += (e2 >> 10) & 1023u; b_seed += e2 >> 20;
                prev_last = lane_get64(c.last_pk, (uint32_t)wave - 1u);
            }
        }
        if constexpr (NARROW) { if (base < n_surv_par && !(first_pk > prev_last)) par_bad = true; }
        // the buffer fills in this round (the same for every wave).  ">=": a round that fills it EXACTLY cuts no child, but the parents
        // behind the last child are not reached any more (mapper.cpp:521-523) -- their dead-end seeds were counted into the published
        // totals and must not stay there (seeded fuzz run, seed 9512: one SA look-up too many)
        const bool cut = TT.nchild + r_child >= max_paths;
        const uint32_t room = b_child >= max_paths ? 0u : max_paths - b_child;
        const uint32_t nwrite = chtot < room ? chtot : room;
        const bool visited = have && choff < room;
        if (room > 0) {
            if (choff + nch < room) c_nbr += ncand;
            else
                for (uint32_t b = 0; b < 4; ++b)
                    if (((mask >> b) & 1u) && choff + (stay_ok ? 1u : 0u) + (uint32_t)__popc(vmask & ((1u << b) - 1u)) < room) c_nbr++;
        }
        const bool ended_seed = ecand && visited;
        const uint64_t em = __ballot(ended_seed);
        {
            const uint32_t pos = b_seed + (uint32_t)prefix_popc(em);
            if (ended_seed) {
                if (pos < max_seed_paths) {
                    SeedPath sp; sp.start = pstart; sp.count = e_count; sp.evt = event_i - 1u; sp.ref_len = e_mc; sp.pad = 0;
                    gst(sb, seedp_off + pos * (uint32_t)sizeof(SeedPath), sp);
                } else seed_over = true;
            }
        }
        // one lane per child that fits
        uint32_t kept_m[5] = {0, 0, 0, 0, 0}, kept_x = 0;
        for (uint32_t l0 = 0; l0 < nwrite; l0 += WAVE) {
            const uint32_t li = l0 + (uint32_t)lane;
            const bool on = li < nwrite;
            uint32_t run = 7u;
            if (on) {
                const uint32_t d = s_cdesc[li];
                const uint32_t pl = d & 63u, type = (d >> 6) & 7u, ci = (pl << 2) | ((type - 1u) & 3u);
                const uint32_t pmt = s_pmeta[pl], pmv = s_pmoves[pl];
                const float last = s_plast[pl], subc = s_psubc[pl];
                const uint64_t hist = s_phist[pl];
                const uint32_t pk = pmt & META_KMER_MASK;
                Row cs, ce;
                uint32_t ck, mv;
                if (type == 0) { cs = s_pstart[pl]; ce = s_pend[pl]; ck = pk; mv = 0; }
                else {
                    const uint64_t pr = s_res[ci];
                    if constexpr (NARROW) { cs = (uint32_t)pr; ce = cs + (uint32_t)(pr >> 32) - 1u; }
                    else { cs = pr >> RES_BITS; ce = cs + (pr & ((1ull << RES_BITS) - 1ull)) - 1ull; }
                    ck = ((pk << 2) & KMASK) | (type - 1u); mv = 1;
                }
                SortKey key;
                const uint32_t gi = b_child + li;
                const ChildHdr c = make_child(pmv, pmt, last, subc, hist, cs, ce, ck, s_probs[ck], mv, p_seed_len, p_min_seed_prob, p_max_stay, gi, klb, key);
                const uint32_t co = chd_off + (gi << PATH_SHIFT);
                if constexpr (NARROW) gst(sb, co, make_uint4(cs, ce, c.moves, c.meta));
                else { const uint64_t r = (cs << KEY_LEN_BITS) | (ce - cs); gst(sb, co, make_uint4((uint32_t)r, (uint32_t)(r >> 32), c.moves, c.meta)); }
                gst(sb, co + 16u, make_uint4(__float_as_uint(c.last), __float_as_uint(subc), (uint32_t)c.hist, (uint32_t)(c.hist >> 32)));
                if constexpr (NARROW) {
                    run = (d >> 9) ? 5u : type;
                    const uint32_t kb = run == 0u ? b_m[0] : run == 1u ? b_m[1] : run == 2u ? b_m[2] : run == 3u ? b_m[3] : run == 4u ? b_m[4] : b_x;
                    gst(sb, str_off + run * run_bytes + ((kb + (uint32_t)s_ckpos[li]) << 3), key.a);
                    gst(sb, info_off + (gi << 3), key.b);
                } else {
                    // wide keys: no room for a position per child in the staging -- from the masks, and for the children of sources
                    // (the tail of the pass in creation order) from where that tail begins
                    const bool csrc = (d >> 9) != 0u;
                    run = csrc ? 5u : type;
                    const uint64_t wm = type == 0u ? tm[0] : type == 1u ? tm[1] : type == 2u ? tm[2] : type == 3u ? tm[3] : tm[4];
                    const uint32_t kb = run == 0u ? b_m[0] : run == 1u ? b_m[1] : run == 2u ? b_m[2] : run == 3u ? b_m[3] : run == 4u ? b_m[4] : b_x;
                    const uint32_t kpos = csrc ? kb + (li - src_cpos0) : kb + (uint32_t)__popcll(wm & ((1ull << pl) - 1ull));
                    gst(sb, str_off + run * (run_bytes << 1) + (kpos << 4), key);
                }
            }
            if (cut) {       // (only the round that hits the cut-off needs the kept counts)
#pragma unroll
                for (uint32_t t = 0; t < 5; ++t) kept_m[t] += (uint32_t)__popcll(__ballot(run == t));
                kept_x += (uint32_t)__popcll(__ballot(run == 5u));
            }
        }
        if (!cut) {
            TT.nchild += r_child; TT.n_seedp += r_seed; TT.scntx += r_x;
#pragma unroll
            for (int t = 0; t < 5; ++t) TT.scnt[t] += r_m[t];
            TT.run_last = round_last;
        } else {
            // the buffer fills in this round: the waves tell each other what they really wrote (the other half of the count slots is free)
            if (lane == 0) {
                TeamCnt c;
                c.chtot = nwrite; c.xtot = kept_x; c.ecnt = (uint32_t)__popcll(em); c.first_is_surv = 0; c.pad0 = 0;
#pragma unroll
                for (int t = 0; t < 5; ++t) c.m[t] = kept_m[t];
                c.first_pk = 0; c.last_pk = 0;
                s_team_cnt[(rnd + 1u) & 1u][wave] = c;
            }
            team_barrier();
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const TeamCnt c = s_team_cnt[(rnd + 1u) & 1u][w];
                TT.nchild += uniform32(c.chtot); TT.n_seedp += uniform32(c.ecnt); TT.scntx += uniform32(c.xtot);
#pragma unroll
                for (int t = 0; t < 5; ++t) TT.scnt[t] += uniform32(c.m[t]);
            }
            team_barrier();        // (the slots are written again two rounds on at the earliest, but the loop ends here: keep it simple)
        }
        wave_sync();
        if (wave == 0) clk.end(1, lane);
    }
    // what the leader needs of the followers: flags, their share of the work counter
    const uint32_t fl = (par_bad ? 2u : 0u) | (__any(seed_over) ? 4u : 0u);
    if (wave != 0) {
        const uint64_t tot = wave_sum64((uint64_t)c_nbr);
        if (lane == 0) { if (fl) atomicOr(&s_team_flags, fl); if (tot) atomicAdd(&s_team_nbr, (unsigned long long)tot); }
        c_nbr = 0;
    } else if (lane == 0 && fl) atomicOr(&s_team_flags, fl);
    return c_nbr;
}

// ---- phase S in a team (narrow keys, an event of more than MERGE_MIN children whose parents were in order and that has no boundary
// child: the common case; everything else is the leader's, on phase_S).  The leader's sort is half of a team's round, and most of it
// is work on independent pieces: the four runs of moves are repaired by four waves at once (their misfits go to disjoint parts of the
// unsorted run's room and are pushed together afterwards), the tiles of both merges are dealt to the waves -- a tile is independent
// of its neighbours once its two merge-path splits are known -- and only the walk, which carries state from key to key, stays with
// the leader: it walks the sorted tiles where the waves left them in LDS.
__shared__ uint32_t s_team_v[TEAM_MAX];            // repair: misfits of run 1 + i; merge: keys of the tile wave i has left in its staging
__shared__ uint32_t s_team_bad;                    // a tile did not come out ascending

__device__ __forceinline__ uint64_t *team_stage(int wave) { return wave == 0 ? s_e : s_e_x[wave > 0 ? wave - 1 : 0]; }

// tile [d0, d1) of merge(A, B): both splits, the tile's share of A and B staged in `tile`, each lane's outputs in o[0 .. cnt) (logical
// positions d .. d + cnt of the tile); tn = keys of the tile
template <int RA, int RB>
__device__ __forceinline__ void team_merge_tile(cgptr_t sb, const KeyArr<RA> &A, const KeyArr<RB> &B, uint32_t d0, uint32_t d1, uint64_t *tile, int lane,
                                                uint64_t (&o)[MERGE_C], uint32_t &d, uint32_t &cnt, uint32_t &tn) {
    const uint32_t n = A.n + B.n;
    const uint32_t a0 = d0 == 0u ? 0u : merge_split(sb, A, B, d0, lane);
    const uint32_t a1 = d1 == n ? A.n : merge_split(sb, A, B, d1, lane);
    const uint32_t b0 = d0 - a0, b1 = d1 - a1;
    const uint32_t na = a1 - a0, nb = b1 - b0;
    tn = na + nb;
    stage_tile(sb, A, B, a0, b0, na, tn, tile, lane);
    wave_sync();
    d = (uint32_t)lane * MERGE_C < tn ? (uint32_t)lane * MERGE_C : tn;
    cnt = tn - d < MERGE_C ? tn - d : MERGE_C;
    uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
    while (__any(lo < hi)) {
        if (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile[mslot(mid)] < tile[mslot(na + d - 1u - mid)]) lo = mid + 1u; else hi = mid;
        }
    }
    uint32_t ia = lo, ib = d - lo;
    uint64_t va = ia < na ? tile[mslot(ia)] : ~0ull, vb = ib < nb ? tile[mslot(na + ib)] : ~0ull;
#pragma unroll
    for (uint32_t c = 0; c < MERGE_C; ++c) {
        const bool ta = va < vb;
        o[c] = ta ? va : vb;
        if (ta) ++ia; else ++ib;
        const uint32_t idx = ta ? ia : na + ib;
        const bool ok = ta ? ia < na : ib < nb;
        uint64_t x = ~0ull;
        if (ok && c + 1u < cnt) x = tile[mslot(idx)];
        if (ta) va = x; else vb = x;
    }
    wave_sync();
}

// returns false when the event is not one for the team (the leader then runs phase_S)
template <int W>
static __device__ __noinline__ bool phase_S_team(kargs_t A_, gptr_t sb_, int lane, int wave) {
    const kargs_t A = uniform_ptr(A_);
    const gptr_t sb = uniform_ptr(sb_);
    const uint32_t n = ctx_get(s_w.nchild);
    if (!(n > MERGE_MIN) || ctx_get(s_w.par_unsorted) || !MERGE_REPAIR) return false;     // (uniform over the team)
    const uint32_t max_paths = A->sc.max_paths;
    const uint32_t str_off = A->sc.off_streams, run_bytes = max_paths << 3, x_off = str_off + 5u * run_bytes;
    uint64_t *const tile = team_stage(wave);
    uint32_t scnt[5] = {ctx_get(s_w.scnt[0]), ctx_get(s_w.scnt[1]), ctx_get(s_w.scnt[2]), ctx_get(s_w.scnt[3]), ctx_get(s_w.scnt[4])};
    uint32_t nx = ctx_get(s_w.scnt[5]);
    // ---- repair: the moves with base r - 1 are run r, taken by wave (r - 1) mod W; its misfits go behind the unsorted run's keys and everything the earlier
    // runs could send there (room: the runs hold n keys together)
    for (uint32_t r = (uint32_t)wave + 1u; r <= 4u; r += (uint32_t)W) {          // (run r's misfits are counted in s_team_v[r - 1])
        uint32_t area = nx;
        for (uint32_t q = 1; q < r; ++q) area += scnt[q];
        const uint32_t v = scnt[r] > 1u ? repair_run(sb, str_off + r * run_bytes, scnt[r], x_off, area, lane) : 0u;
        if (lane == 0) s_team_v[r - 1u] = v;
    }
    if (wave == 0 && lane == 0) s_team_bad = 0;
    team_barrier();
    {
        // the misfits, pushed together behind the unsorted run (few keys; the leader); every wave learns the new counts
        uint32_t v[4], area = nx, dst = nx;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) v[q] = uniform32(s_team_v[q]);
        for (uint32_t q = 0; q < 4; ++q) {
            if (wave == 0 && v[q] && area != dst)
                for (uint32_t i0 = 0; i0 < v[q]; i0 += WAVE) {
                    const uint32_t i = i0 + (uint32_t)lane;
                    uint64_t kq = 0;
                    if (i < v[q]) kq = gld<uint64_t>(sb, x_off + ((area + i) << 3));
                    wave_sync();
                    if (i < v[q]) gst(sb, x_off + ((dst + i) << 3), kq);
                    wave_sync();
                }
            area += scnt[q + 1u]; dst += v[q];
            scnt[q + 1u] -= v[q];
        }
        nx = dst;
    }
    // ---- the unsorted run: the leader (the network is a single wavefront's)
    if (wave == 0 && nx > 1) {
        KeyArr<6> KX;
#pragma unroll
        for (int r = 0; r < 6; ++r) { KX.adj[r] = x_off; KX.cum[r] = 0; }
        KX.n = nx;
        sort_any64(sb, KX, x_off, lane);
        wave_sync();
    }
    team_barrier();
    // ---- moves + unsorted -> tmp, tile by tile
    KeyArr<4> KM;
    KM.cum[0] = 0; KM.cum[1] = scnt[1]; KM.cum[2] = scnt[1] + scnt[2]; KM.cum[3] = scnt[1] + scnt[2] + scnt[3];
    KM.n = KM.cum[3] + scnt[4];
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) KM.adj[r] = str_off + (r + 1u) * run_bytes - (KM.cum[r] << 3);
    KeyArr<4> KB = KM;
    if (nx > 0) {
        if (KM.n > 0) {
            const KeyArr<1> KX1 = ka_single(x_off, nx);
            const uint32_t nm = KM.n + nx, out_off = A->sc.off_tmp;
            for (uint32_t t = (uint32_t)wave; t * MERGE_TILE < nm; t += (uint32_t)W) {
                const uint32_t d0 = t * MERGE_TILE, d1 = d0 + MERGE_TILE < nm ? d0 + MERGE_TILE : nm;
                uint64_t o[MERGE_C];
                uint32_t d, cnt, tn;
                team_merge_tile<4, 1>(sb, KM, KX1, d0, d1, tile, lane, o, d, cnt, tn);
#pragma unroll
                for (uint32_t c = 0; c < MERGE_C; ++c)
                    if (c < cnt) tile[mslot(d + c)] = o[c];
                wave_sync();
#pragma unroll
                for (uint32_t c = 0; c < MERGE_C; ++c) {
                    const uint32_t i = (uint32_t)lane + c * WAVE;
                    if (i < tn) gst(sb, out_off + ((d0 + i) << 3), tile[mslot(i)]);
                }
                wave_sync();
            }
            team_barrier();
            KB.adj[0] = A->sc.off_tmp;
        } else KB.adj[0] = x_off;
        KB.cum[1] = KB.cum[2] = KB.cum[3] = KB.n = KM.n + nx;
        KB.adj[1] = KB.adj[2] = KB.adj[3] = KB.adj[0];
    }
    // ---- stays + the rest, merged tile by tile into the waves' staging and walked there by the leader, W tiles at a time
    const KeyArr<1> KA = ka_single(str_off, scnt[0]);
    const uint32_t nk = KA.n + KB.n;            // == n
    const WalkConst<true> C = walk_const<true>(A, sb);
    WalkState<true> S;
    S.n_seedp = ctx_get(s_w.n_seedp);
    if (wave == 0) { walk_begin_narrow(lane); if (lane == 0) s_w.mixed = 0u; }
    const uint32_t kl = A->ix.key_len_bits;
    const float source_prob = C.source_prob;
    const UNC_AS_GLOBAL ulonglong2 *const kmer_ranges2 = (const UNC_AS_GLOBAL ulonglong2 *)A->ix.kmer_ranges;
    const UNC_AS_GLOBAL uint64_t *const infow = reinterpret_cast<const UNC_AS_GLOBAL uint64_t *>(sb + A->sc.off_info);
    uint64_t pend_key = 0;
    bool have_pend = false, bad = false;
    for (uint32_t g0 = 0; g0 * MERGE_TILE < nk; g0 += (uint32_t)W) {
        const uint32_t t = g0 + (uint32_t)wave;
        if (t * MERGE_TILE < nk) {
            const uint32_t d0 = t * MERGE_TILE, d1 = d0 + MERGE_TILE < nk ? d0 + MERGE_TILE : nk;
            uint64_t o[MERGE_C];
            uint32_t d, cnt, tn;
            team_merge_tile<1, 4>(sb, KA, KB, d0, d1, tile, lane, o, d, cnt, tn);
            {   // the tile must come out ascending (its place after the tile before it is the leader's to check)
                uint64_t last = o[0];
                bool w = false;
#pragma unroll
                for (uint32_t c = 1; c < MERGE_C; ++c)
                    if (c < cnt) { w = w || !(o[c] > last); last = o[c]; }
                const uint64_t pl = (uint64_t)__shfl_up((unsigned long long)last, 1);
                if (lane > 0 && cnt > 0 && !(o[0] > pl)) w = true;
                if (__any(w) && lane == 0) atomicOr(&s_team_bad, 1u);
            }
            // sorted tile at logical positions 1 .. tn (0: the key that waits for its successor, the leader's to fill)
#pragma unroll
            for (uint32_t c = 0; c < MERGE_C; ++c)
                if (c < cnt) tile[mslot(1u + d + c)] = o[c];
            if (lane == 0) s_team_v[wave] = tn;
        }
        team_barrier();
        if (wave == 0) {
            for (uint32_t w = 0; w < (uint32_t)W && (g0 + w) * MERGE_TILE < nk; ++w) {
                uint64_t *const wt = team_stage((int)w);
                const uint32_t tn = uniform32(s_team_v[w]);
                const bool last_tile = (g0 + w + 1u) * MERGE_TILE >= nk;
                if (have_pend) {
                    if (lane == 0) wt[mslot(0)] = pend_key;
                    if (!(uniform64(wt[mslot(1)]) > pend_key)) bad = true;
                }
                wave_sync();
                const uint32_t p0 = have_pend ? 0u : 1u, p1 = last_tile ? tn + 1u : tn;
                uint64_t bq0, bq1;          // (every lane asks in every pass, so that the waits can be counted: see merge_walk)
                {
                    const uint32_t pa = p0 + (uint32_t)lane, pb = pa + WAVE;
                    bq0 = infow[wt[mslot(pa <= tn ? pa : tn)] & 0xFFFFu];
                    bq1 = infow[wt[mslot(pb <= tn ? pb : tn)] & 0xFFFFu];
                }
                u32x4_t krq;
                {
                    const uint32_t km0 = (uint32_t)(bq0 & META_KMER_MASK);
                    krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[km0] >= source_prob ? km0 : 0u));
                }
                for (uint32_t base = p0; base < p1; base += WAVE) {
                    const uint32_t p = base + (uint32_t)lane;
                    const bool have = p < p1;
                    const bool has_next = have && p < tn;
                    const uint32_t nv = p1 - base < WAVE ? p1 - base : WAVE;
                    const uint64_t bq2 = infow[wt[mslot(p + 2 * WAVE <= tn ? p + 2 * WAVE : tn)] & 0xFFFFu];      // (requested ahead of the pass's work: merge_walk)
                    const uint64_t ki = have ? wt[mslot(p)] : ~0ull, kn = has_next ? wt[mslot(p + 1u)] : ~0ull;
                    const uint64_t bi = bq0;
                    uint64_t bn = (uint64_t)__shfl((unsigned long long)bi, (lane + 1) & 63);
                    const uint64_t bf = bcast64(bq1, 0);
                    if (lane == WAVE - 1) bn = bf;
                    bq0 = bq1;
                    u32x4_t kt = krq;
                    mem_retire(kt);
                    ulonglong2 krc;
                    krc.x = ((uint64_t)kt.y << 32) | kt.x; krc.y = ((uint64_t)kt.w << 32) | kt.z;
                    {
                        const uint32_t kmn = (uint32_t)(bq0 & META_KMER_MASK);
                        krq = g_load(reinterpret_cast<const UNC_AS_GLOBAL u32x4_t *>(kmer_ranges2) + (s_probs[kmn] >= source_prob ? kmn : 0u));
                    }
                    uint32_t start, end, nstart, kmer, nkmer;
                    uint64_t sbw;
                    bool dup;
                    walk_decode_narrow<true>(S, kl, ki, kn, bi, bn, have, has_next, nv, lane, start, end, nstart, kmer, nkmer, dup, sbw);
                    walk_core<true>(C, S, have, has_next, start, end, nstart, kmer, nkmer, dup, sbw, true, krc, nv, lane);
                    bq1 = bq2;
                }
                pend_key = uniform64(wt[mslot(tn)]);
                have_pend = true;
                wave_sync();
            }
        }
        team_barrier();
    }
    if (wave == 0) {
        if ((bad || uniform32(s_team_bad)) && lane == 0) s_w.tstatus |= UNC_READ_SORT_FAULT;
        walk_finish<true>(C, S, lane);
        CTX_SET(kl, kl);
        CTX_SET(walked, 1u);
        wave_sync();
    }
    return true;
}

// The chunked path with a team of W wavefronts per channel (workgroup b = chunk descriptor b, slot = slot_map[b]; see the one-wavefront
// kernel's resume branch for the state a channel keeps between chunks).  The leader (wave 0) owns the read's state and every phase
// but P and E; the followers learn after each event whether it goes on.
// (The register budget is the one-wavefront kernel's -- UNC_LB wavefronts per SIMD -- although a flow cell's 512 teams never fill the
// chip: the phases the two kernels share are compiled ONCE, for the looser of their callers' budgets, and a team kernel without the
// bound took the batch kernel from 126 to 197 VGPRs, i.e. from 16 to 8 wavefronts per CU.)
template <bool PROF, bool NARROW, int W>
__global__ __launch_bounds__(64 * W, UNC_LB) void k_map_team(MapArgs Aval) {
    const kargs_t A = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Aval;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const bool lead = wave == 0;
    const uint32_t r = blockIdx.x;
    const uint32_t slot = A->slot_map ? A->slot_map[blockIdx.x] : blockIdx.x;
    const gptr_t sb = (gptr_t)A->sc.base + (size_t)slot * A->sc.slot_bytes;
    UNC_AS_GLOBAL SlotState *const st = reinterpret_cast<UNC_AS_GLOBAL SlotState *>(sb + A->sc.off_state);
    const bool fresh = A->rd.new_read && A->rd.new_read[blockIdx.x];   // first chunk of a read

    uint32_t event_i = 0, tstatus = 0, notes = 0;
    Tracker T;
    T.n = 0; T.n_lens = 0; T.max1 = 0; T.max2 = 0; T.status = 0; T.len_sum = 0.0f; T.n_alloc = 0;
    T.mm.ref_st = 0; T.mm.rstart = 1; T.mm.rend = 0; T.mm.evt_st = 1; T.mm.evt_en = 0; T.mm.total_len = 0;  // NULL_ALN
    uint64_t c_nbr = 0, c_sa = 0, c_lf = 0;      // per-lane partial counters (leader)
    if (!fresh) {
        if (uniform32(st->done)) return;             // (every wave reads the same word)
        event_i = uniform32(st->event_i);
        tstatus = uniform32(st->status);
    }
    if (lead) {
        uint32_t n_parents = 0, cur = 0, n_surv_par = 0;
        if (lane < 12) s_cyc[lane] = 0;
        if (!fresh) {
            n_parents = uniform32(st->n_parents); cur = uniform32(st->cur); n_surv_par = uniform32(st->n_surv);
            notes = uniform32(st->notes);
            T.n = st->n_clusters; T.n_lens = st->n_lens; T.max1 = st->len_max1; T.max2 = st->len_max2;
            T.status = st->status; T.len_sum = st->len_sum; T.n_alloc = st->n_alloc;
            T.mm.ref_st = st->max_map.ref_st; T.mm.rstart = st->max_map.rstart; T.mm.rend = st->max_map.rend;
            T.mm.evt_st = st->max_map.evt_st; T.mm.evt_en = st->max_map.evt_en; T.mm.total_len = st->max_map.total_len;
            if (lane == 0) { c_nbr = st->n_nbr; c_sa = st->n_sa; c_lf = st->n_lf; }
        } else {
            {   // a new read takes the channel over: what the previous one still holds goes back to the pool
                Tracker old;
                old.n_alloc = uniform32(st->n_alloc);
                tracker_release(old, tracker_mem(A, sb), lane);
            }
            tracker_clear_heads(tracker_mem(A, sb), lane);
        }
        // sources_added_ is NOT cleared by Mapper::new_read / reset (mapper.cpp:88,219-246,612-623): a channel is one Mapper
        if (lane < NKMER / 32) s_flags[lane] = st->sources_added[lane];
        if (lane == 0) {
            s_T = T;
            s_w.event_i = event_i; s_w.n_parents = n_parents; s_w.cur = cur; s_w.n_surv_par = n_surv_par; s_w.tstatus = tstatus;
            s_w.notes = notes;
            s_team_flags = 0; s_team_nbr = 0ull; s_team_ctl = 0;
        }
    }
    team_barrier();
    const uint32_t n_events = A->rd.info[r].n_events;
    const float scale = A->rd.info[r].scale, shift = A->rd.info[r].shift;
    const UNC_AS_GLOBAL float *const means = (const UNC_AS_GLOBAL float *)A->rd.means + A->rd.moff[r];
    const uint32_t ring_mod = A->rd.ring_mod, ring_r0 = A->rd.ring0[r];
    const uint32_t max_events = A->P.max_events, max_seed_paths = A->sc.max_seed_paths;

    uint32_t done = 0;
    float next_mean = event_i < n_events ? means[(ring_r0 + event_i) % ring_mod] : 0.0f;
    for (;;) {
        // map_next prologue, mapper.cpp:434-437 (norm_.empty() <=> every event popped)
        if (event_i >= n_events && event_i < max_events && !tstatus) break;        // chunk mapped: park
        if (event_i >= n_events || event_i >= max_events || tstatus) { done = 2; break; }
        PhaseClock<PROF> clk;
        const float level = __fadd_rn(__fmul_rn(scale, next_mean), shift);         // Normalizer::at
        if (event_i + 1 < n_events) next_mean = means[(ring_r0 + event_i + 1) % ring_mod];
        if (lead && lane == 0) gst(sb, A->sc.off_levels + ((event_i & (LEVEL_RING - 1u)) << 2), level);
        phase_P_team<W>(A, level, lane, wave);
        team_barrier();
        if (lead) clk.end(0, lane);
        TeamTotals TT;
        c_nbr += phase_E_team<PROF, NARROW, W>(A, sb, lane, wave, TT);
        team_barrier();
        if (lead) {
            clk.reset();          // (phase E keeps its own clock: parts 8..11, the children in 1)
            const uint32_t fl = uniform32(s_team_flags);
            const unsigned long long nbr_f = s_team_nbr;
            uint32_t n_seedp = TT.n_seedp, tst = 0;
            if (n_seedp > max_seed_paths) { tst = UNC_READ_SEED_OVERFLOW; n_seedp = max_seed_paths; }
            if (lane == 0) {
                s_w.nchild = TT.nchild; s_w.n_seedp = n_seedp; s_w.tstatus |= tst; s_w.par_unsorted = (fl >> 1) & 1u;
                s_w.scnt[0] = TT.scnt[0]; s_w.scnt[1] = TT.scnt[1]; s_w.scnt[2] = TT.scnt[2]; s_w.scnt[3] = TT.scnt[3]; s_w.scnt[4] = TT.scnt[4];
                s_w.scnt[5] = TT.scntx;
                s_w.walked = 0u;
                s_team_flags = 0; s_team_nbr = 0ull;
                c_nbr += nbr_f;
            }
            wave_sync();
            clk.reset();
        }
        bool team_sorted = false;
        if constexpr (NARROW) {
            team_barrier();       // (the event's counts are in the context: every wave decides alike whether the sort is the team's)
            team_sorted = phase_S_team<W>(A, sb, lane, wave);
            if (lead && team_sorted) clk.end(2, lane);
        }
        if (lead) {
            if (ctx_get(s_w.nchild) > 0 && !team_sorted) {
                phase_S<NARROW>(A, sb, lane);
                clk.end(2, lane);
                if (!ctx_get(s_w.walked)) phase_W<NARROW>(A, sb, lane);
            }
            if constexpr (NARROW) {
                if (ctx_get(s_w.nchild) > 0 && ctx_get(s_w.mixed)) { phase_S_wide_redo(A, sb, lane); phase_W<NARROW>(A, sb, lane); }
            }
            if (ctx_get(s_w.nchild) > 0 && !team_sorted) clk.end(3, lane);
            phase_F<NARROW>(A, sb, lane);
            clk.end(4, lane);
            bool conf = false;
            if (ctx_get(s_w.n_seedp) > 0 && !ctx_get(s_w.tstatus)) {
                const uint64_t c = phase_T<PROF>(A, sb, lane);
                clk.reset();
                c_sa += (uint32_t)c; c_lf += c >> 32;
                conf = ctx_get(s_w.conf) != 0;
            }
            const uint32_t ts = ctx_get(s_w.tstatus);
            const uint32_t code = ts ? 2u : conf ? 1u : 0u;
            if (lane == 0) { s_team_ctl = code; if (!code) s_w.event_i = event_i + 1u; }
            tstatus = ts;
            clk.end(7, lane);
        }
        team_barrier();
        const uint32_t code = uniform32(s_team_ctl);
        if (code == 2u) { done = 2; break; }
        if (code == 1u) { done = 1; break; }
        event_i++;
    }
    if (!lead) return;
    // ---------------- publish / park (the leader; as the one-wavefront kernel's chunked branch) ----------------
    const uint32_t n_parents = ctx_get(s_w.n_parents), cur = ctx_get(s_w.cur), n_surv_par = ctx_get(s_w.n_surv_par);
    notes = ctx_get(s_w.notes);
    if (done && __any(lane < NKMER / 32 && s_flags[lane < NKMER / 32 ? lane : 0] != 0u)) notes |= UNC_NOTE_FLAGS_LEFT;
    const uint64_t t_nbr = wave_sum64(c_nbr), t_sa = wave_sum64(c_sa), t_lf = wave_sum64(c_lf);
    T = s_T;
    T.status |= tstatus;
    T.n_alloc = uniform32(T.n_alloc);
    if (done && lane == 0) {
        DevResult res;
        res.done = done; res.status = T.status; res.event_i = event_i; res.notes = notes;
        res.cluster = T.mm;
        res.n_nbr = t_nbr; res.n_sa = t_sa; res.n_lf = t_lf;
        res.ticks = 0ull; res.pad2 = 0;
        for (int i = 0; i < 12; ++i) res.cyc[i] = PROF ? s_cyc[i] : 0ull;
        g_store((UNC_AS_GLOBAL DevResult *)A->results + r, res);
    }
    if (done) tracker_release(T, tracker_mem(A, sb), lane);       // a decided read's nodes go back at once
    if (lane == 0) {
        st->read_idx = r; st->event_i = event_i; st->n_parents = n_parents; st->cur = cur; st->done = done;
        st->n_surv = n_surv_par;
        st->status = T.status; st->n_clusters = T.n; st->n_lens = T.n_lens;
        st->len_max1 = T.max1; st->len_max2 = T.max2; st->len_sum = T.len_sum;
        st->max_map.ref_st = T.mm.ref_st; st->max_map.rstart = T.mm.rstart; st->max_map.rend = T.mm.rend;
        st->max_map.evt_st = T.mm.evt_st; st->max_map.evt_en = T.mm.evt_en; st->max_map.total_len = T.mm.total_len;
        st->notes = notes; st->n_alloc = T.n_alloc;        // (a decided read's final notes stay for the chunked path's host side; a new read starts from 0)
        st->n_nbr = t_nbr; st->n_sa = t_sa; st->n_lf = t_lf; st->t_start = 0;
        if constexpr (PROF) { for (int i = 0; i < 12; ++i) st->cyc[i] = (fresh ? 0ull : st->cyc[i]) + s_cyc[i]; }
    }
    if (lane < NKMER / 32) st->sources_added[lane] = s_flags[lane];
}

}  // namespace unc

#include "unc_kernels.h"
namespace unc {
void launch_map(const DevIndex &ix, const DevScratch &sc, const DevReads &rd, const unc_params_t &P, DevResult *results,
                uint32_t *next_read, uint32_t max_steps, uint32_t resume, const uint32_t *slot_map, uint32_t grid, hipStream_t st,
                const DevPool &pool, const uint32_t *read_list, unsigned long long *wave_ticks, const DevSched *sched, bool profile,
                const uint32_t *flags_in, uint32_t *flags_out, uint32_t team) {
    MapArgs a;
    a.flags_in = flags_in; a.flags_out = flags_out;
    if (sched) a.sched = *sched; else { a.sched.ctl = nullptr; a.sched.free_cells = a.sched.park_cells = nullptr; a.sched.cap_mask = a.sched.n_slots = 0; }
    a.ix = ix; a.sc = sc; a.rd = rd; a.P = P; a.results = results; a.next_read = next_read;
    a.max_steps = max_steps; a.resume = resume; a.slot_map = slot_map; a.read_list = read_list; a.wave_ticks = wave_ticks;
    a.pool = pool;
    const bool narrow = ix.key_len_bits != 0;
    if (team > 1 && resume && rd.ring_mod) {
        // the chunked path with a team of wavefronts per channel (k_map_team): 2, 4 or 8
#define UNC_TEAM_LAUNCH(P_, N_, W_) hipLaunchKernelGGL((k_map_team<P_, N_, W_>), dim3(grid), dim3(WAVE * W_), 0, st, a)
        if (team >= 8) { if (profile) { if (narrow) UNC_TEAM_LAUNCH(true, true, 8); else UNC_TEAM_LAUNCH(true, false, 8); }
                         else { if (narrow) UNC_TEAM_LAUNCH(false, true, 8); else UNC_TEAM_LAUNCH(false, false, 8); } }
        else if (team >= 4) { if (profile) { if (narrow) UNC_TEAM_LAUNCH(true, true, 4); else UNC_TEAM_LAUNCH(true, false, 4); }
                         else { if (narrow) UNC_TEAM_LAUNCH(false, true, 4); else UNC_TEAM_LAUNCH(false, false, 4); } }
        else { if (profile) { if (narrow) UNC_TEAM_LAUNCH(true, true, 2); else UNC_TEAM_LAUNCH(true, false, 2); }
               else { if (narrow) UNC_TEAM_LAUNCH(false, true, 2); else UNC_TEAM_LAUNCH(false, false, 2); } }
#undef UNC_TEAM_LAUNCH
        return;
    }
    if (profile && narrow) hipLaunchKernelGGL((k_map<true, true>), dim3(grid), dim3(WAVE), 0, st, a);
    else if (profile) hipLaunchKernelGGL((k_map<true, false>), dim3(grid), dim3(WAVE), 0, st, a);
    else if (narrow) hipLaunchKernelGGL((k_map<false, true>), dim3(grid), dim3(WAVE), 0, st, a);
    else hipLaunchKernelGGL((k_map<false, false>), dim3(grid), dim3(WAVE), 0, st, a);
}
// every slot free, nothing parked, queue head at the first read
__global__ void k_sched_init(DevSched S) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, cap = S.cap_mask + 1u;
    const uint32_t part = t / cap, i = t % cap, spp = S.n_slots / S.n_parts;       // partition p owns slots p * spp .. (p + 1) * spp - 1
    if (part < S.n_parts) {
        SchedCell f; f.seq = i < spp ? i + 1u : i; f.val = part * spp + i;
        S.free_cells[t] = f;
        SchedCell p; p.seq = i; p.val = 0;
        S.park_cells[t] = p;
    }
    if (t == 0) {
        SchedCtl c;
        memset(&c, 0, sizeof c);
        for (uint32_t q = 0; q < S.n_parts; ++q) c.freeq[q].tail = spp;
        *S.ctl = c;
    }
}
void launch_sched_init(const DevSched &S, hipStream_t st) {
    const uint32_t n = (S.cap_mask + 1u) * S.n_parts;
    hipLaunchKernelGGL(k_sched_init, dim3((n + 255) / 256), dim3(256), 0, st, S);
}
// which XCD a workgroup runs on, per workgroup of a small grid: the host counts the distinct answers (unc_host.cpp: xcd_count)
__global__ void k_xcd_probe(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcd_id(); }
void launch_xcd_probe(uint32_t *out, uint32_t n_blocks, hipStream_t st) { hipLaunchKernelGGL(k_xcd_probe, dim3(n_blocks), dim3(64), 0, st, out); }
// every chunk of the node pool free
__global__ void k_pool_init(DevPool B) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, cap = B.cap_mask + 1u;
    if (i < cap) {
        SchedCell f; f.seq = i < B.n_chunks ? i + 1u : i; f.val = i;
        B.cells[i] = f;
    }
    if (i == 0) {
        PoolQueue q;
        memset(&q, 0, sizeof q);
        q.avail = (int32_t)B.n_chunks;
        q.tail = B.n_chunks;
        q.low_water = B.n_chunks;
        *B.q = q;
    }
}
void launch_pool_init(const DevPool &B, hipStream_t st) {
    const uint32_t cap = B.cap_mask + 1u;
    hipLaunchKernelGGL(k_pool_init, dim3((cap + 255) / 256), dim3(256), 0, st, B);
}
// resident single-wave workgroups per CU for the persistent grid: UNC_LB waves on each of the 4 SIMDs (launch bounds =
// register budget); the LDS footprint (under 10 KB of the CU's 160 KB) admits 16.
uint32_t map_kernel_waves_per_cu() { return 4 * UNC_LB; }
int map_kernel_attributes(bool narrow, bool profile, uint32_t *out4) {
    hipFuncAttributes a;
    const void *f = narrow ? (profile ? (const void *)k_map<true, true> : (const void *)k_map<false, true>)
                           : (profile ? (const void *)k_map<true, false> : (const void *)k_map<false, false>);
    const hipError_t e = hipFuncGetAttributes(&a, f);
    if (e != hipSuccess) return (int)e;
    out4[0] = (uint32_t)a.numRegs; out4[1] = (uint32_t)a.localSizeBytes; out4[2] = (uint32_t)a.sharedSizeBytes; out4[3] = (uint32_t)a.maxThreadsPerBlock;
    return 0;
}
}  // namespace unc
