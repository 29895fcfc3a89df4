// Shared host/device data layout of the MI355X mapper (no HIP headers needed here).
#pragma once
#include <stdint.h>

#include "../../include/uncalled_hip.h"

namespace unc {

constexpr int WAVE = 64;
constexpr int SEED_LEN = UNC_SEED_LEN;
constexpr int NKMER = UNC_NKMER;
constexpr uint32_t KMASK = NKMER - 1;
constexpr int MAX_REP_COPY_LIMIT = 64;

// One path of the forest (Mapper::PathBuffer, mapper.hpp:139-196) = one 32-byte record.  The 23 float prefix sums
// (prob_sums_) are NOT stored: the reference only ever reads prob_sums_[length_] (the newest sum) and, once the window is
// full, prob_sums_[1] (the sum that drops out next), and prob_sums_[1] = prob_sums_[0] + (the match probability of the
// window's oldest event) is recomputed from what that event's k-mer was.  A record carries
//   last   prob_sums_[length_]
//   sub    prob_sums_[0]: 0 until the window has slid for the first time
//   hist   the lineage's k-mer at the OLDEST event of the window (10 bits) | the bases its moves have shifted in since
//          (newest in the low bits; at most 21 of them, 2 bits each) << 10
// so that a child of a full-window path gets  sub' = sub + match_prob(level of the oldest event, its k-mer)  -- the same
// float addition, on the same operands, that made prob_sums_[1] in the reference -- and slides the k-mer one event on.
// seed_prob_ is a function of (last, sub, length) and is recomputed where it is needed.  (Rounds 1-2 kept the sums: 128-byte
// records, then 64-byte records + 96-byte rings copied for every live path every 4th event: most of the kernel's traffic.)
struct alignas(16) PathRec {
    uint32_t r0, r1;            // fm_range_: start, end (fewer than 2^32 rows) or start << 30 | (end - start), low word first
    uint32_t moves;             // event_moves_ (22-bit shift register, LSB = newest event)
    uint32_t meta;              // kmer | length | consec_stays | sa_checked | first_full, see below
    float last, sub;
    uint32_t hist_lo, hist_hi;
};
static_assert(sizeof(PathRec) == 32, "PathRec is two 16-byte quads");
constexpr int PATH_SHIFT = 5;            // log2(sizeof(PathRec))
constexpr uint32_t LEVEL_RING = 32;      // normalised levels of the last events kept per read (the oldest window event is 22 back)

constexpr uint32_t META_KMER_MASK = 0x3FFu;
constexpr int META_LEN_SHIFT = 10;       // 5 bits
constexpr int META_STAY_SHIFT = 16;      // 8 bits
constexpr uint32_t META_SA_CHECKED = 1u << 24;
constexpr uint32_t META_FIRST_FULL = 1u << 25;   // length_ reached seed_len with this event: prob_sums_[0] is still the initial 0

// Sort key of one child (operator< of mapper.cpp:866-871 made total by creation order):
//   a = fm_range_.start << 30 | (fm_range_.length - 1)      (start asc, then end asc)
//   b = orderable(seed_prob_) << 32 | idx << 16 | seedflag << 10 | kmer
// idx (creation order) sits above the payload bits so ties resolve by creation order only.
struct alignas(16) SortKey { uint64_t a, b; };
constexpr int KEY_LEN_BITS = 30;
constexpr uint64_t KEY_LEN_MASK = (1ull << KEY_LEN_BITS) - 1;
constexpr uint32_t KEYB_SEED_FLAG = 1u << 10;
constexpr int KEYB_MOVES_SHIFT = 11;     // 5 bits: popcount(event_moves_)

// A path that passed is_seed_valid (mapper.cpp:842-863): its FM rows become seeds
struct alignas(8) SeedPath { uint64_t start; uint32_t count; uint32_t evt; uint32_t ref_len; uint32_t pad; };

// SeedTracker state (seed_tracker.hpp:69-110): the std::set<SeedCluster> (ordered by ref_en_.start desc, evt_en_ desc) is kept per
// read as a GRID OF BUCKETS over ref_en_.start (map_tracker.h, add_seeds): unordered nodes of NODE_K clusters chained from a per-read
// table of bucket heads.  A cluster = a hot key of 16 bytes (all the scan of add_seed reads: ref_en_.start, evt_en_, total_len_) + a
// cold part of 32 bytes (what a merge needs on top: ref_st_, ref_en_.end, evt_st_).
//   node  = header 16 B (count, next node + 1) | NODE_K hot keys | NODE_K cold parts, padded to a multiple of 128 bytes
//   pool  = ONE per mapper: chunks of POOL_CHUNK_BYTES handed out through a ring of free chunk ids; a read takes what it needs --
//           tens of nodes on a bacterial reference, tens of thousands for an off-target read on a human-sized one -- and gives
//           it back when it is decided
struct alignas(16) ClusterKey { uint64_t rstart; uint32_t evt_en; uint32_t total_len; };      // hot part of a cluster
struct alignas(16) ClusterCold { uint64_t ref_st, rend; uint32_t evt_st; uint32_t pad[3]; };  // cold part
constexpr uint32_t POOL_CHUNK_BYTES = 192u << 10;      // 768 nodes of 256 bytes
#ifndef UNC_NODE_K
#define UNC_NODE_K 5
#endif
constexpr uint32_t NODE_K = UNC_NODE_K;
constexpr uint32_t NODE_BYTES = (16 + NODE_K * 48 + 127) / 128 * 128;      // 256 (384 for 7 clusters, 512 for 10)
constexpr uint32_t CHUNK_NODES = POOL_CHUNK_BYTES / NODE_BYTES;
// the narrowest bucket: 2^12 rows.  A seed's window of `event` rows then spans at most 9 buckets with the default max_events; the lane
// that adds the seed walks them one after the other (map_tracker.h), and wider buckets make more seeds of an event wait for each other
constexpr uint32_t BUCKET_SHIFT_MIN = 12;
static_assert(CHUNK_NODES * NODE_BYTES == POOL_CHUNK_BYTES, "node layout");

struct ClusterVal { uint64_t ref_st, rstart, rend; uint32_t evt_st, evt_en, total_len; };

// Resumable per-slot mapper state (everything Mapper keeps between map_next calls)
struct alignas(16) SlotState {
    uint32_t read_idx;
    uint32_t event_i;
    uint32_t n_parents;      // compacted list of valid parents, in the reference's visiting order
    uint32_t cur;            // which of the two path buffers holds the parents
    uint32_t done;           // 0 = mapping, 1 = SUCCESS, 2 = FAILURE
    uint32_t status;
    uint32_t n_clusters, n_lens, len_max1, len_max2, notes, n_alloc;  // notes: UNC_NOTE_* so far; n_alloc: nodes taken from the read's own chunks so far
    uint32_t n_surv;         // the first n_surv parents are the last walk's survivors, in sorted order (sources follow)
    uint32_t pad_[1];
    float len_sum;
    ClusterVal max_map;
    uint64_t n_nbr, n_sa, n_lf;
    uint64_t t_start;        // device wall clock when the read's first event was taken up
    uint32_t sources_added[NKMER / 32];
    uint64_t cyc[12];        // phase cycle counters of the read so far (sliced batch mode)
};

// ---- sliced batch scheduler (k_map): more reads in flight than resident wavefronts.  A wavefront maps a read for at
// most `slice` events, parks it in its slot and takes the next task: a new read while free slots remain, else the
// longest-parked one.  Long (off-target) reads are thereby discovered early and share the wavefronts until the end,
// instead of a few of them keeping single wavefronts busy long after the queue has drained.
// Both queues are bounded multi-producer/multi-consumer rings of slot ids with a sequence number per cell.  A cell is read and written as
// ONE 64-bit word (sequence number low, value high): value and sequence number can never be seen apart, so the rings need no
// acquire / release pair per operation -- on this chip an agent-scope acquire is an invalidate of the XCD's whole L2 and a release a
// write-back of it (`buffer_inv sc1` / `buffer_wbl2 sc1`), and round 6 found a polling loop of acquire loads to be what held whole XCDs
// up on GRCh38.  The counters sit on lines of their own (128 bytes: the L2's line).
struct alignas(8) SchedCell { uint32_t seq, val; };
struct alignas(128) SchedQueue { uint32_t head; uint32_t pad0[31]; uint32_t tail; uint32_t pad1[31]; };
// One pair of rings PER XCD (SCHED_MAX_PARTS; DevSched::n_parts of them in use): a slot belongs to one XCD for good -- its reads are
// taken up, parked and resumed by wavefronts of that XCD only, whose L2 is coherent among them.  Handing a slot from one XCD to
// another costs a write-back of the giver's whole L2 and an invalidate of the taker's (agent-scope release / acquire on this chip);
// with one pair of rings for the whole device that was paid on every park and every resume, five to six times per read.
constexpr uint32_t SCHED_MAX_PARTS = 8;
struct alignas(128) SchedCtl {
    uint32_t next_read, pad0[31];
    SchedQueue freeq[SCHED_MAX_PARTS], parkq[SCHED_MAX_PARTS];
};
// The node pool of the seed-cluster grids (see ClusterKey): chunk ids travel through a ring as well, but one that is never polled by a
// compare-and-swap loop: `avail` counts the chunks that are in the ring (published) and not yet spoken for -- a pop first takes one
// off that count (or finds the pool dry and puts it back), THEN a ticket off `head`, and waits for that cell's push to be published,
// which is under way by then; a push takes a ticket off `tail`, writes its cell and adds one to `avail`.  Every operation is a fixed
// number of atomics whoever else is at the ring.
// low_water: the fewest chunks `avail` has held since the ring was initialised -- the pool's high-water mark of chunks out at once is
// n_chunks - low_water (unc_mapper_pool_usage: the pool is sized by it, not by a share of the free HBM)
struct alignas(128) PoolQueue {
    int32_t avail; uint32_t pad0[31];
    uint32_t head; uint32_t pad1[31];
    uint32_t tail; uint32_t pad2[31];
    uint32_t low_water; uint32_t pad3[31];
};
struct DevPool {
    char *nodes;                // [n_chunks][POOL_CHUNK_BYTES]
    PoolQueue *q;
    SchedCell *cells;
    uint32_t cap_mask, n_chunks;
};

struct DevSched {
    SchedCtl *ctl;           // null: scheduler off (one slot per wavefront)
    SchedCell *free_cells, *park_cells;   // [n_parts][cap] each
    uint32_t cap_mask;       // cap - 1, cap = power of two >= n_slots / n_parts
    uint32_t n_slots;
    uint32_t n_parts;        // 1: one pair of rings for all wavefronts (slots cross XCDs: release / acquire around every hand-over);
                             // the device's number of XCDs: one pair per XCD, partition p owns slots p * n_slots / n_parts ...
};

struct DevIndex {
    const uint32_t *bwt;        // 64-byte blocks: 4 x u64 counts + 8 x u32 of 2-bit BWT (bwa layout)
    const uint64_t *sa;         // sampled SA, interval 32; sa[0] = -1 (bwa layout)
    const uint64_t *sa_dense;   // [seq_len + 1] x 6 bytes: full SA | LF-steps << 34 (fm_dev.h: sa_entry_load), or null (then the sampled walk is used)
    const uint64_t *kmer_ranges;  // [1024][2]
    const float *model;         // [3][1024]: lv_means, lv_vars_x2, lognorm_denoms
    const float *model4;        // [1024][4]: the same, one 16-byte row per k-mer (a lane that needs ONE k-mer's row: one access)
    const uint16_t *kmer_valid; // [64]: bit j of entry l = range of k-mer j*64+l is non-empty
    const uint32_t *fm32;       // rank table of a reference with fewer than 2^32 rows (8 words per 64 symbols, fm_dev.h), else null
    uint64_t primary, seq_len;
    uint64_t L2[5];
    uint32_t key_len_bits;      // > 0: (start, length, child index) pack into one 64-bit sort key with this many length bits
    uint32_t bucket_shift;      // seed clusters are bucketed by ref_en_.start >> bucket_shift (k_map.hip, add_seed) ...
    uint32_t n_buckets;         // ... into this many buckets per read
    uint32_t pad_;
    float thresholds[64];
};

// Per-slot scratch: ONE allocation, slot s at base + s * slot_bytes, regions at fixed byte offsets inside a slot (all
// below 4 GB, so kernels address them as uniform base + 32-bit offset).  Layout computed by scratch_layout().
struct DevScratch {
    char *base;
    uint64_t slot_bytes;
    uint32_t max_paths, keys_cap, max_seed_paths, max_clusters;
    // region offsets inside a slot
    uint32_t off_paths;    // PathRec [2][max_paths]
    uint32_t off_levels;   // float   [LEVEL_RING]: normalised level of event e at [e % LEVEL_RING]
    uint32_t off_order;    // u32     [2][max_paths]
    uint32_t off_keys;     // SortKey [2][keys_cap]   (unsorted | sorted)
    uint32_t off_seedp;    // SeedPath[max_seed_paths]
    uint32_t off_tasks;    // u64     [WAVE * MAX_REP_COPY_LIMIT]
    uint32_t off_cl_dir;   // u32     [n_buckets]: heads of the read's seed-cluster buckets (node index + 1)
    uint32_t off_cl_chunks;  // u32   [max_clusters / 4 / 512 + 1]: the pool chunks this read holds
    uint32_t off_state;    // SlotState
    // narrow sort keys (DevIndex::key_len_bits > 0): the children's 64-bit keys leave phase E as sorted streams
    uint32_t off_streams;  // u64 (SortKey with 128-bit keys) [6][max_paths]: stays, moves by base 0..3 (children of the sorted survivors), the rest
    uint32_t off_info;     // u64     [max_paths]: a child's info word (SortKey::b) by creation index
    uint32_t off_tmp;      // u64     [max_paths]: intermediate run of the merge
};

// ---- chunked (realtime) path: state a Mapper keeps per channel between chunks (mapper.hpp:209-226) ----
constexpr uint32_t NORM_LEN = 6000;   // Normalizer::PRMS_DEF.len, normalizer.cpp:4-8
constexpr uint32_t PROF_WIN = 25;     // EventProfiler::PRMS_DEF.win_len, event_profiler.cpp:4-10

struct RtDetector { float threshold; uint32_t window_length, masked_to; int32_t peak_pos; float peak_value; uint32_t valid_peak; };

struct alignas(16) RtChan {
    // EventDetector (event_detector.hpp:106-128); cumulative sums by absolute position & 15
    double sum[16], sumsq[16];
    double evt_st_sum, evt_st_sumsq;
    uint32_t t, evt_st, total_events;
    float len_sum;
    RtDetector sd, ld;
    // EventProfiler (event_profiler.hpp:35-48): rolling 25-window + the queued event means
    double pw_mean, pw_varsum;
    float pw_signal[PROF_WIN + 3];
    float evq[PROF_WIN + 3];
    uint32_t pw_n, pw_rd, pw_wr, pw_full, q_head, q_len, prof_full, to_mask;
    // Normalizer in rolling mode (normalizer.hpp:74-79); the 6000-float ring lives in DevRt::norm_ring
    double n_mean, n_varsum;
    uint32_t n_n, n_rd, n_wr, n_full, n_empty;
    // current read
    uint32_t ring0;        // ring slot holding event 0 of this read
    uint32_t n_pushed;     // events pushed for this read so far
    uint32_t status;       // UNC_READ_* bits
    uint32_t pad[2];
};

struct RtChunkDesc {       // one chunk = one channel's next <= 4000 samples (Chunk, chunk.hpp:32-59)
    uint64_t offset;       // into the raw int16 array
    uint32_t n_samples;
    uint32_t channel;
    uint32_t new_read;     // Mapper::new_read(Chunk&) instead of add_chunk
    float cal_range, cal_offset, cal_digit;
};

struct DevReads {
    const int16_t *raw;
    const uint64_t *offsets;      // n_reads + 1
    const unc_calib_t *calib;
    float *means;                 // event means, read i at means[moff[i]..]
    const uint64_t *moff;         // n_reads + 1
    unc_evt_info_t *info;
    uint32_t n_reads;
    float tgt_mean, tgt_stdv;     // PoreModel::get_means_mean/stdv (mapper.cpp:94)
    // chunked path: events live in a per-channel ring, event e of read r at means[moff[r] + (ring0[r] + e) % ring_mod]
    const uint32_t *ring0;        // null in batch mode
    const uint32_t *new_read;     // per descriptor: 1 = start from a fresh Mapper::reset() state
    uint32_t ring_mod;            // 0 in batch mode
};

// what the map kernel hands back per read
struct alignas(16) DevResult {
    uint32_t done, status, event_i, notes;     // notes: UNC_NOTE_* (unc_hit_t::notes)
    ClusterVal cluster;
    uint64_t n_nbr, n_sa, n_lf;
    uint64_t ticks;    // device wall clock ticks from the read's first event to this result
    uint64_t pad2;
    uint64_t cyc[12];  // shader-clock cycles per phase: P, E(rest), S, W, F, T(sa), T(add_seed), G, E1, E2, E3, E4
};

}  // namespace unc
