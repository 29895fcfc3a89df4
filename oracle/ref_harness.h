/* TEST INFRASTRUCTURE -- C ABI of oracle/_ref/libunc_ref.so: the reference's own hot-path
 * sources (/root/reference/src/{mapper,event_detector,normalizer,event_profiler,
 * seed_tracker,range,read_buffer,chunk}.cpp) compiled IN PLACE, unmodified, with the
 * reference's flags (-std=c++11 -O3, setup.py:121) against oracle/shim + oracle/minibwa.c.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. */
#ifndef UNC_REF_HARNESS_H
#define UNC_REF_HARNESS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t mapped, fwd;
    uint64_t rd_st, rd_en, rd_len;      /* PAF cols 3,4,2 */
    uint64_t rf_st, rf_en, rf_len;      /* PAF cols 8,9,7 */
    uint32_t matches;                   /* PAF col 10; col 11 = rf_en - rf_st + 1 */
    uint32_t n_events;                  /* events kept by the detector over the whole read */
    uint32_t event_i;                   /* Mapper::event_i_ when map_read returned */
    float mean_event_len;
    uint64_t n_nbr, n_sa, n_lf;         /* minibwa work counters for this read */
    double map_ms;
    char rf_name[128];
} ref_hit_t;

typedef struct { float mean, stdv; uint32_t start, length; } ref_event_t;

typedef struct {
    uint64_t fm_start, fm_end;
    uint32_t event_moves;
    float seed_prob;
    uint16_t kmer;
    uint8_t length, consec_stays, sa_checked, pad[3];
    float prob_sums[23];
} ref_path_t;

typedef struct {
    uint64_t ref_st, ref_en_start, ref_en_end;
    uint32_t evt_st, evt_en, total_len, pad;
} ref_cluster_t;

/* Sets Mapper::PRMS.{bwa_prefix,idx_preset}; first Mapper construction loads the statics.
 * max_events==0 keeps the reference default (30000).  Returns 0 on success. */
int ref_init(const char *bwa_prefix, const char *idx_preset, uint32_t max_events);
/* Mapper::PRMS.max_paths for Mappers constructed afterwards (mapper.cpp:33,83-86) */
void ref_set_max_paths(uint32_t max_paths);
/* The mapping, event-detector and seed-tracker parameters of Mapper::PRMS (mapper.hpp:49-72, event_detector.hpp Params,
 * seed_tracker.hpp:75-79) for Mappers constructed afterwards; v = the 15 values in the order of the arguments' names */
void ref_set_params(uint32_t min_rep_len, uint32_t max_rep_copy, uint32_t max_paths, uint32_t max_consec_stay, uint32_t max_events,
                    float max_stay_frac, float min_seed_prob, float threshold1, float threshold2, float peak_height,
                    float min_mean, float max_mean, uint32_t min_map_len, float min_mean_conf, float min_top_conf);
/* How oracle/shim/pdqsort.h orders children that TIE on (fm_range_, seed_prob_) at mapper.cpp:531 -- 0 creation order (stable; the
 * project's convention), 1 pattern-defeating quicksort as restated there, 2 reversed creation order -- for every sort from now
 * on, and what the sorts have seen since the last reset: out[0] sorts (events with children), out[1] sorts with at least one
 * tied adjacent pair, out[2] tied adjacent pairs. */
void ref_set_sort_mode(int mode);
void ref_sort_stats(uint64_t *out3, int reset);
int ref_sort_selftest(uint32_t n, uint32_t seed, uint32_t key_range, int shape);   /* the restated sort itself; 0 = ok */
void *ref_mapper_new(void);
void ref_mapper_free(void *m);

/* read_buffer.cpp:239-241 (u16 reinterpretation quirk included). */
void ref_calibrate(const int16_t *raw, uint64_t n, float range, float offset, float digitisation, float *out);

int ref_map_read(void *m, const float *signal, uint32_t n, ref_hit_t *out);
/* N threads, one Mapper each, tight new_read->map_read loop over an interleaved shard
 * (BASELINE.md "B1").  Returns wall seconds of the mapping loop. */
double ref_map_batch(int n_threads, uint32_t n_reads, const float *signals, const uint64_t *offsets, ref_hit_t *out);
/* the same through the as-shipped MapPool hand-shake with its 10 ms polling sleeps (SURVEY 8d "B2") */
double ref_map_batch_pool(int n_threads, uint32_t n_reads, const float *signals, const uint64_t *offsets, ref_hit_t *out);

/* chunked path: one read through new_read(Chunk)/add_chunk/process_chunk/map_chunk on this Mapper (= channel) */
int ref_chunk_read(void *m, const float *signal, uint32_t n, uint32_t chunk_len, uint32_t number, ref_hit_t *out,
                   uint32_t *chunks_used);
void ref_set_max_chunks(uint32_t max_chunks);
int ref_last_ended(void);          /* Paf::is_ended() of the read ref_chunk_read mapped last */
/* stage tap of the chunked path (same layout as unc_o_rt_tap_t): Mapper::evdt_ / evt_prof_ / norm_ after ref_chunk_read */
typedef struct {
    uint32_t det_t, det_total_events;
    float det_len_sum;
    uint32_t norm_n, norm_wr;
    uint32_t prof_n, prof_to_mask, prof_queued;
    double norm_mean, norm_varsum;
    double prof_mean, prof_varsum;
    float prof_queue[28];
} ref_rt_tap_t;
void ref_rt_tap(void *mapper, ref_rt_tap_t *out, float *ring, uint32_t ring_cap);

/* `uncalled index`: FM-range-size trajectories of sampled reference positions (self_align_ref.cpp:34-91) */
uint64_t ref_self_align(const char *bwa_prefix, uint32_t sample_dist, uint64_t *lens, uint64_t lens_cap, uint64_t *offsets,
                        uint64_t offsets_cap, uint64_t *n_paths);

/* stage taps */
uint32_t ref_events(const float *signal, uint32_t n, ref_event_t *out, uint32_t cap, float *mean_event_len, uint32_t *total_events);
void ref_norm_levels(const float *means, uint32_t m, float *levels, float *scale, float *shift);
void ref_match_probs(float level, float *out1024);
void ref_model_tables(float *means1024, float *vars_x2_1024, float *lognorm1024, float *model_mean, float *model_stdv);
void ref_kmer_ranges(uint64_t *out2048);
void ref_thresholds(float *out64);
void ref_get_neighbor(uint64_t s, uint64_t e, int base, uint64_t *os, uint64_t *oe);
uint64_t ref_sa(uint64_t k);
uint64_t ref_fm_size(void);

/* step-wise trace of Mapper::map_read (mapper.cpp:188-200) */
void ref_trace_begin(void *m, const float *signal, uint32_t n);
int ref_trace_step(void *m);                                   /* one map_next(); 1 when done */
uint32_t ref_trace_paths(void *m, ref_path_t *out, uint32_t cap);  /* prev_paths_[0..prev_size_) */
uint32_t ref_trace_clusters(void *m, ref_cluster_t *out, uint32_t cap, ref_cluster_t *max_map, float *len_sum, uint32_t *n_lens);
uint32_t ref_trace_event_i(void *m);
void ref_trace_finish(void *m, ref_hit_t *out);

#ifdef __cplusplus
}
#endif
#endif
