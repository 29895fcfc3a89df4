/* TEST INFRASTRUCTURE -- NOT part of the product.
 *
 * unc_oracle: a plain-C, single-threaded CPU restatement of the reference's per-read Mapper
 * pipeline (skovaka/UNCALLED v2.3.0): calibration -> event detection -> whole-read
 * normalisation -> r9.4 5-mer match log-probs -> FM-index path forest -> seed clustering ->
 * PAF coordinates.  Every function cites the reference file:line it follows.
 *
 * Pinned against (tests/test_oracle_vs_ref.py, tests/test_golden.py):
 *   - oracle/_ref = the reference's own sources compiled in place (stage taps + full PAF) on the
 *     bundled example read and on seeded synthetic reads, and
 *   - the committed fixtures under tests/golden/ generated from that build.
 * The FM-index arithmetic itself (lh3/bwa 0.7.17, an un-vendored submodule) is restated in
 * minibwa.c from bwa's published on-disk format.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#ifndef UNC_ORACLE_H
#define UNC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UNC_O_SEED_LEN 22
#define UNC_O_NKMER 1024

typedef struct {
    /* Mapper::PRMS, mapper.cpp:29-52 */
    uint32_t seed_len, min_rep_len, max_rep_copy, max_paths, max_consec_stay, max_events;
    float max_stay_frac, min_seed_prob;
    /* EventDetector::PRMS_DEF, event_detector.cpp:17-26 */
    uint32_t window_length1, window_length2;
    float threshold1, threshold2, peak_height, min_mean, max_mean;
    /* SeedTracker::PRMS_DEF, seed_tracker.cpp:28-32 */
    uint32_t min_map_len;
    float min_mean_conf, min_top_conf;
    /* ReadBuffer::PRMS, read_buffer.cpp:26-32 */
    float bp_per_sec, sample_rate;
} unc_o_params_t;

typedef struct { float mean, stdv; uint32_t start, length; } unc_o_event_t;

typedef struct {
    uint64_t fm_start, fm_end;
    uint32_t event_moves;
    float seed_prob;
    uint16_t kmer;
    uint8_t length, consec_stays, sa_checked, pad[3];
    float prob_sums[UNC_O_SEED_LEN + 1];
} unc_o_path_t;

typedef struct {
    uint64_t ref_st, ref_en_start, ref_en_end;
    uint32_t evt_st, evt_en, total_len, pad;
} unc_o_cluster_t;

typedef struct {
    int32_t mapped, fwd;
    int32_t rid;                        /* bns_pos2rid of the hit, -1 if none */
    uint32_t matches;
    uint64_t rd_st, rd_en, rd_len;
    uint64_t rf_st, rf_en, rf_len;
    uint32_t n_events, event_i;
    float mean_event_len;
    uint32_t notes;                     /* UNC_O_NOTE_* */
    uint64_t n_nbr, n_sa, n_lf;         /* SURVEY 8(d) work counters */
    unc_o_cluster_t cluster;            /* the winning SeedTracker::max_map_ (zero if unmapped) */
} unc_o_hit_t;

/* what a read ran into that the reference does not report but that decides whether Mapper state leaks into the NEXT read of the
 * same Mapper: the path buffer was full after some event (next_path == next_paths_.end(), mapper.cpp:480,507,521,543,577,607), and
 * sources_added_ flags were still set when the read ended (mapper.cpp:612-623 clears them only on the way to the buffer's end) */
#define UNC_O_NOTE_PATHS_FULL 1u
#define UNC_O_NOTE_FLAGS_LEFT 2u

typedef struct unc_o_index unc_o_index_t;
typedef struct unc_o_mapper unc_o_mapper_t;

void unc_o_params_default(unc_o_params_t *p);

unc_o_index_t *unc_o_index_load(const char *bwa_prefix, const char *idx_preset);
void unc_o_index_free(unc_o_index_t *ix);
uint64_t unc_o_index_size(const unc_o_index_t *ix);
const char *unc_o_index_ref_name(const unc_o_index_t *ix, int rid);
void unc_o_index_kmer_ranges(const unc_o_index_t *ix, uint64_t *out2048);
void unc_o_index_thresholds(const unc_o_index_t *ix, float *out64);
void unc_o_index_get_neighbor(const unc_o_index_t *ix, uint64_t s, uint64_t e, int base, uint64_t *os, uint64_t *oe);
uint64_t unc_o_index_sa(const unc_o_index_t *ix, uint64_t k);

/* stage functions */
void unc_o_calibrate(const int16_t *raw, uint64_t n, float range, float offset, float digitisation, float *out);
uint32_t unc_o_detect_events(const unc_o_params_t *p, const float *signal, uint32_t n, unc_o_event_t *out,
                             uint32_t cap, float *mean_event_len, uint32_t *total_events);
void unc_o_model_tables(float *means, float *vars_x2, float *lognorm, float *model_mean, float *model_stdv);
void unc_o_normalize(const float *means, uint32_t m, float *levels, float *scale, float *shift);
void unc_o_match_probs(float level, float *out1024);

/* whole path */
unc_o_mapper_t *unc_o_mapper_new(const unc_o_index_t *ix, const unc_o_params_t *p);
void unc_o_mapper_free(unc_o_mapper_t *m);
int unc_o_map_read(unc_o_mapper_t *m, const float *signal, uint32_t n, unc_o_hit_t *out);
/* N threads over an atomic read counter; returns wall seconds of the mapping loop */
double unc_o_map_batch(const unc_o_index_t *ix, const unc_o_params_t *p, int n_threads, uint32_t n_reads,
                       const float *signals, const uint64_t *offsets, unc_o_hit_t *out);

/* chunked (realtime / MAP_ORD) path: one read fed to this mapper chunk by chunk; the rolling normaliser persists
 * across reads on the same mapper (= channel), mapper.cpp:225-226 */
void unc_o_set_max_chunks(unc_o_mapper_t *m, uint32_t max_chunks);
int unc_o_chunk_read(unc_o_mapper_t *m, const float *signal, uint32_t n, uint32_t chunk_len, unc_o_hit_t *out,
                     uint32_t *chunks_used);

/* stage tap of the chunked path: the state this mapper (= channel) carries between chunks below the PAF; ring: NORM_LEN floats */
typedef struct {
    uint32_t det_t, det_total_events;
    float det_len_sum;
    uint32_t norm_n, norm_wr;
    uint32_t prof_n, prof_to_mask, prof_queued;
    double norm_mean, norm_varsum;
    double prof_mean, prof_varsum;
    float prof_queue[28];
} unc_o_rt_tap_t;
void unc_o_rt_tap(const unc_o_mapper_t *m, unc_o_rt_tap_t *out, float *ring);
int unc_o_rt_ended(const unc_o_mapper_t *m);      /* the last chunked read's Paf ENDED flag (mapper.cpp:386) */

/* step-wise trace */
void unc_o_trace_begin(unc_o_mapper_t *m, const float *signal, uint32_t n);
int unc_o_trace_step(unc_o_mapper_t *m);
uint32_t unc_o_trace_paths(const unc_o_mapper_t *m, unc_o_path_t *out, uint32_t cap);
uint32_t unc_o_trace_clusters(const unc_o_mapper_t *m, unc_o_cluster_t *out, uint32_t cap, unc_o_cluster_t *max_map,
                              float *len_sum, uint32_t *n_lens);
uint32_t unc_o_trace_event_i(const unc_o_mapper_t *m);
void unc_o_trace_finish(unc_o_mapper_t *m, unc_o_hit_t *out);
/* per-event statistics of the last map_read/trace: number of parents visited and children made */
void unc_o_stats(const unc_o_mapper_t *m, uint64_t *sum_parents, uint64_t *sum_children, uint32_t *max_children,
                 uint64_t *n_seeds);

#ifdef __cplusplus
}
#endif
#endif
