/* uncalled_hip.h -- C ABI of libuncalled_hip.so: the MI355X (gfx950) implementation of UNCALLED's
 * per-read Mapper hot path (event detection -> normalisation -> r9.4 5-mer match -> FM-index path
 * forest -> seed clustering -> PAF coordinates), batched over reads, one wavefront per read.
 *
 * Every entry point names the reference interface it stands in for (skovaka/UNCALLED v2.3.0,
 * paths relative to the reference root).  No exceptions cross this boundary: functions return
 * UNC_OK (0) or a negative unc_status_t; unc_last_error() gives the message for the calling thread.
 * Plain pointers and sizes only -- no C++/torch types.
 */
#ifndef UNCALLED_HIP_H
#define UNCALLED_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    UNC_OK = 0,
    UNC_ERR_ARG = -1,        /* bad argument / unsupported parameter value */
    UNC_ERR_IO = -2,         /* index file missing or malformed (reference: abort(), mapper.cpp:118-127) */
    UNC_ERR_HIP = -3,        /* HIP runtime error */
    UNC_ERR_NOMEM = -4,
    UNC_ERR_OVERFLOW = -5    /* a per-read device scratch area overflowed (see unc_hit_t.status) */
} unc_status_t;

/* Compile-time shape of the kernels (the reference's defaults; mapper.cpp:30, event_detector.cpp:18-19) */
#define UNC_SEED_LEN 22
#define UNC_KLEN 5
#define UNC_NKMER 1024
#define UNC_WINDOW1 3
#define UNC_WINDOW2 6

/* Mapper::PRMS (mapper.cpp:29-52) + EventDetector::PRMS_DEF (event_detector.cpp:17-26) +
 * SeedTracker::PRMS_DEF (seed_tracker.cpp:28-32) + ReadBuffer::PRMS (read_buffer.cpp:26-32);
 * the subset Conf exposes for the map path (conf.hpp:296-340). */
typedef struct {
    uint32_t seed_len;          /* must be 22 */
    uint32_t min_rep_len;
    uint32_t max_rep_copy;      /* <= 64 */
    uint32_t max_paths;         /* <= 65535 */
    uint32_t max_consec_stay;
    uint32_t max_events;
    float max_stay_frac;
    float min_seed_prob;
    uint32_t window_length1;    /* must be 3 */
    uint32_t window_length2;    /* must be 6 */
    float threshold1, threshold2, peak_height, min_mean, max_mean;
    uint32_t min_map_len;
    float min_mean_conf, min_top_conf;
    float bp_per_sec, sample_rate;
    float chunk_time;
    uint32_t max_chunks;
} unc_params_t;

/* fast5 channel calibration, read_buffer.cpp:212-222,239-241 */
typedef struct { float range, offset, digitisation; } unc_calib_t;

/* per-read status bits */
#define UNC_READ_OK 0u
#define UNC_READ_CLUSTER_OVERFLOW 1u   /* seed-cluster scratch exhausted: result for this read is invalid */
#define UNC_READ_SEED_OVERFLOW 2u      /* per-event seed list exhausted: result for this read is invalid */
#define UNC_READ_POOL_DRY 16u          /* with CLUSTER_OVERFLOW: the shared pool of cluster nodes was empty (the read's own allowance was not used up) */
#define UNC_READ_SORT_FAULT 8u        /* internal: the runs of child keys handed to the merge were not ascending (never expected; result invalid) */
#define UNC_READ_NORM_FULL 4u          /* chunked path: >= 6000 unread events.  The reference's #SKIP branch (mapper.cpp:336-351) is OUT OF SCOPE:
                                         * unc_rt_create refuses chunk_time * sample_rate > 12000 samples, below which a chunk cannot yield 6000 events
                                         * (the detector never fires on consecutive samples) -- the bit is the belt to that pair of braces */

/* per-read notes (unc_hit_t::notes): conditions under which the reference's Mapper carries state from one read into the NEXT read
 * mapped by the same thread -- which the batch path, where every read starts from a fresh Mapper, does not reproduce (with
 * `-t N > 1` the reference's own outcome then depends on which thread gets which read).  A batch whose reads carry neither
 * note is mapped exactly as `uncalled map -t 1` maps it, whatever the order.  The chunked path (unc_rt_*) reports the same two
 * bits per read -- there a channel IS one Mapper and the carry-over is reproduced, the bits only say that it happened. */
#define UNC_NOTE_PATHS_FULL 1u         /* after some event the path buffer held max_paths paths (mapper.cpp:480,507,521,543,577,607) */
#define UNC_NOTE_FLAGS_LEFT 2u         /* sources_added_ flags were still set when the read ended (mapper.cpp:88,612-623) */

/* One PAF record's worth of coordinates (Paf, read_buffer.hpp:42-126; set by Mapper::set_ref_loc,
 * mapper.cpp:708-728) plus the work counters of SURVEY.md section 8(d). */
typedef struct {
    int32_t mapped, fwd, rid;
    uint32_t status;
    uint64_t rd_st, rd_en, rd_len;      /* PAF cols 3,4,2 */
    uint64_t rf_st, rf_en, rf_len;      /* PAF cols 8,9,7 */
    uint32_t matches;                   /* PAF col 10; col 11 = rf_en - rf_st + 1 */
    uint32_t n_events;                  /* events kept by the detector over the whole read */
    uint32_t event_i;                   /* Mapper::event_i_ at the end of map_read */
    float mean_event_len;
    uint64_t n_nbr, n_sa, n_lf;         /* get_neighbor calls, SA lookups, LF steps inside them */
    /* winning SeedCluster (seed_tracker.hpp:40-66), zero when unmapped */
    uint64_t cl_ref_st, cl_ref_en_start, cl_ref_en_end;
    uint32_t cl_evt_st, cl_evt_en, cl_total_len;
    float map_ms;                       /* batch path: time the read spent on the device, first event taken up -> result written
                                         * (device wall clock; the PAF `mt` tag, mapper.cpp:197).  Reads share wavefronts in time
                                         * slices, so this is residence, not service time.  0 on the chunked / trace paths */
    uint32_t notes;                     /* UNC_NOTE_* bits: not errors, the result is valid */
    uint32_t pad_;
} unc_hit_t;

/* stage tap: per-read result of the event/normalisation kernel */
typedef struct {
    uint32_t n_events;       /* events with mean in [min_mean, max_mean] */
    uint32_t total_events;   /* all events (EventDetector::total_events_) */
    float len_sum;           /* EventDetector::len_sum_ */
    float scale, shift;      /* Normalizer::at, normalizer.cpp:114-118 */
    uint32_t pad;
} unc_evt_info_t;

typedef struct {
    uint64_t fm_start, fm_end;
    uint32_t event_moves;
    float seed_prob;
    uint16_t kmer;
    uint8_t length, consec_stays, sa_checked, pad[3];
    float prob_sums[UNC_SEED_LEN + 1];
} unc_path_t;

typedef struct {
    uint64_t ref_st, ref_en_start, ref_en_end;
    uint32_t evt_st, evt_en, total_len, pad;
} unc_cluster_t;

typedef struct unc_index unc_index_t;
typedef struct unc_mapper unc_mapper_t;

const char *unc_last_error(void);
const char *unc_version(void);
/* PCI address ("0000:c1:00.0") of HIP device `device` as this library numbers the devices (hipDeviceGetPCIBusId): what the
 * one-process-per-GPU launchers look up under /sys/bus/pci/devices/<address>/numa_node to keep a worker's host threads on its GPU's
 * NUMA node (the reference has no counterpart: its workers are threads of one process, map_pool.cpp:31-42).  UNC_OK or an error. */
int unc_device_pci_address(int device, char *out, int cap);
/* page-locked host memory for batches handed to unc_map_batch with on_device == 0 (the copy to HBM then runs at full
 * PCIe rate and asynchronously); NULL on failure */
void *unc_host_alloc(uint64_t bytes);
void unc_host_free(void *p);

/* Conf defaults: compiled-in PRMS of the reference (SURVEY.md section 5 "Config / flags") */
void unc_params_default(unc_params_t *p);

/* ---- index: replaces Mapper::load_static (mapper.cpp:109-159) = BwaIndex::load_index
 * (bwa_index.hpp:116-135: bwt_restore_bwt / bwt_restore_sa / bns_restore + the 1024 k-mer ranges)
 * + the .uncl threshold parser (mapper.cpp:123-157) + PoreModel tables (pore_model.hpp:58-103).
 * Parses <prefix>.{bwt,sa,ann,amb,uncl} on the host and uploads BWT/Occ blocks, sampled SA, k-mer
 * ranges, thresholds and the 1024x3 model table to HBM of `device`. */
int unc_index_load(const char *bwa_prefix, const char *idx_preset, int device, unc_index_t **out);
void unc_index_free(unc_index_t *ix);
uint64_t unc_index_size(const unc_index_t *ix);                      /* BwaIndex::size, bwa_index.hpp:180-182 */
int32_t unc_index_n_seqs(const unc_index_t *ix);
const char *unc_index_seq_name(const unc_index_t *ix, int32_t rid);  /* BwaIndex::get_ref_name, :197-199 */
uint64_t unc_index_seq_len(const unc_index_t *ix, int32_t rid);      /* BwaIndex::get_ref_len, :201-203 */
/* BwaIndex::translate_loc, bwa_index.hpp:213-220 (bns_pos2rid): returns the sequence length, 0 if none */
uint64_t unc_index_translate_loc(const unc_index_t *ix, uint64_t sa_loc, int32_t *rid, uint64_t *ref_loc);
uint64_t unc_index_device_bytes(const unc_index_t *ix);
/* host copies of the derived tables (parity taps) */
void unc_index_kmer_ranges(const unc_index_t *ix, uint64_t *out2048);       /* BwaIndex::get_kmer_range */
void unc_index_thresholds(const unc_index_t *ix, float *out64);             /* Mapper::prob_threshes_ */
void unc_index_model_tables(const unc_index_t *ix, float *means1024, float *vars_x2_1024, float *lognorm1024,
                            float *model_mean, float *model_stdv);
/* device FM primitives run over arrays (parity taps for bwa_index.hpp:158-162 and :176-178) */
int unc_fm_get_neighbor(const unc_index_t *ix, uint32_t n, const uint64_t *starts, const uint64_t *ends,
                        const uint8_t *bases, uint64_t *out_starts, uint64_t *out_ends);
int unc_fm_sa(const unc_index_t *ix, uint32_t n, const uint64_t *rows, uint64_t *out);
/* device PoreModel::match_prob over all 1024 k-mers (pore_model.hpp:163-165; mapper.cpp:443-445) */
int unc_match_probs(const unc_index_t *ix, uint32_t n, const float *levels, float *out /* n x 1024 */);

/* ---- `uncalled index` (scripts/uncalled:38-78): self-alignment of sampled reference positions, the FM walk whose
 * range-size trajectories IndexParameterizer (uncalled/index.py:53-209) turns into the .uncl thresholds.  Replaces
 * self_align(bwa_prefix, sample_dist) (src/self_align_ref.cpp:34-91): positions are sampled with the same
 * srand(0)/rand() % sample_dist draw per base, the walks run on the device.  lens (host) receives up to `cap` range
 * sizes per trajectory (row-major, n_paths x cap), full_len[i] the trajectory's true length.  Call with lens == NULL
 * to get n_paths only.  Needs <prefix>.pac next to the index files. */
int unc_self_align(const unc_index_t *ix, const char *bwa_prefix, uint32_t sample_dist, uint32_t cap, uint64_t *lens,
                   uint32_t *full_len, uint64_t max_paths, uint64_t *n_paths);

/* ---- mapper: replaces N x (Mapper::new_read + Mapper::map_read) (mapper.cpp:188-207), i.e. the
 * body of MapPool::MapperThread::run (map_pool.cpp:130-158), for a whole batch of reads. */
typedef struct {
    uint32_t n_slots;        /* reads in flight = per-read scratch slots (0 = 3 x n_waves, memory permitting) */
    uint32_t max_clusters;   /* a read's allowance of the shared node pool: max_clusters / 4 nodes (of 5 clusters; 0 = 2^20);
                              * reads that outgrow it are mapped again with a 16x larger allowance */
    uint32_t max_seed_paths; /* seed-valid paths per event (0 = 2 * max_paths, the bound) */
    uint32_t slice_events;   /* with n_slots > n_waves a wavefront parks a read after this many events and takes the
                              * next task (a new read while a slot is free, else the longest-parked read); 0 = 1024 */
    uint32_t n_waves;        /* resident wavefronts of the persistent k_map grid (0 = 16 per CU) */
    uint32_t pool_chunks;    /* the seed-cluster nodes of ALL reads in flight come from one pool, a chunk of 192 KB (768 nodes) at a
                              * time: chunks in the pool (0 = 8 per slot, 24 for references of 2^26 index rows and more, 32 from 2^31 on, at
                              * most 60 % of the free HBM); k_map stops admitting reads while the pool is nearly empty, and a read
                              * that still finds it dry is mapped again after the batch */
    uint32_t sched_parts;    /* pairs of scheduler rings (free slots, parked reads): 0 = one per XCD of the device when n_slots divides
                              * into that many shares of 64 slots or more (a slot then stays with one XCD's wavefronts, whose L2 is
                              * coherent among them: no L2 write-back / invalidate when a read is parked and resumed), else one;
                              * 1 = one pair for the whole device */
    uint32_t events_reads_per_wave;   /* k_events: reads (lanes in use) per wavefront, 1..64 (0 = 64; measured on 50 k reads:
                              * 64 -> 27 ms, 32 -> 36 ms: the kernel is bound by instruction issue) */
} unc_mapper_opts_t;

int unc_mapper_create(const unc_index_t *ix, const unc_params_t *p, const unc_mapper_opts_t *opts, unc_mapper_t **out);
void unc_mapper_free(unc_mapper_t *m);
uint64_t unc_mapper_device_bytes(const unc_mapper_t *m);

/* Map a batch.  raw = concatenated int16 samples, read i = raw[offsets[i] .. offsets[i+1]);
 * calib[i] per read.  `offsets` (n_reads + 1) and `calib` (n_reads) are ALWAYS host arrays (a few bytes per read;
 * they are validated on the host and copied).  `raw` is a host pointer when on_device == 0 (the samples are copied
 * first) and a device pointer when on_device != 0 (samples already resident in HBM).  `stream` is a
 * hipStream_t (NULL = the mapper's own stream).  hits (host, n_reads) receives one record per read. */
int unc_map_batch(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                  const unc_calib_t *calib, int on_device, void *stream, unc_hit_t *hits);
/* The same batch in two halves (unc_map_batch = the two in a row).  _begin stages the reads and launches the kernels on `stream` (NULL:
 * the mapper's own stream) and returns at once; _end waits, maps again the few reads that need it and fills hits[0 .. n_reads).  `raw`
 * (on_device != 0: a device pointer) must stay valid until _end; `offsets` and `calib` are copied by _begin.  One batch per mapper at a
 * time.  What it is for: the worker loop of MapPool::MapperThread::run (map_pool.cpp:130-158) with TWO mappers over one index -- batch
 * k + 1 is begun on the second while batch k's last long reads finish on the first, and moves into the compute units as they fall idle
 * (a persistent launch ends with a few wavefronts on a few long reads: 6 % of a 50 k-read E. coli launch). */
int unc_map_batch_begin(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                        const unc_calib_t *calib, int on_device, void *stream);
int unc_map_batch_end(unc_mapper_t *m, unc_hit_t *hits);
/* wall-clock of the kernels of the last unc_map_batch, from HIP events on the launch stream */
int unc_mapper_last_timing(const unc_mapper_t *m, float *ms_events, float *ms_map);
/* when k_map of the mapper's last batch was submitted and when it ended, in milliseconds on ONE time axis per process (HIP events
 * against a reference event): with batches of two mappers in flight at once (unc_map_batch_begin) the launches overlap, and the time
 * the kernel holds the device per batch is the union of these windows over the number of batches, not the mean of their lengths */
int unc_mapper_last_window(const unc_mapper_t *m, double *start_ms, double *end_ms);
/* shader-clock cycles summed over the reads of the last batch, per k_map phase:
 * [0] match probs, [1] extension (loop overhead), [2] sort, [3] walk, [4] full sources, [5] SA look-ups, [6] add_seed,
 * [7] rest, [8] E1 parent loads + candidates, [9] E2 FM look-ups, [10] E3 child slots, [11] E4 child records */
int unc_mapper_last_phase_cycles(const unc_mapper_t *m, uint64_t *out12);
/* the same per READ of the last batch (n_reads = the batch's): 14 words of 64 bits per read -- the twelve counters above, the read's
 * residence in device wall-clock ticks, and where it was decided (XCC_ID | HW_ID << 8).  Diagnostics: which reads, and which part of the
 * chip, pay when a launch of the same batch runs slower than the last one (the GRCh38 levels) */
int unc_mapper_last_read_cycles(const unc_mapper_t *m, uint32_t n_reads, uint64_t *out14);
/* the counters above are collected only by batches mapped while profiling is on (off by default: the counting
 * instantiation of k_map is about 2 % slower) */
void unc_mapper_set_profile(unc_mapper_t *m, int on);
/* what unc_mapper_create settled on: [0] resident wavefronts, [1] reads in flight (slots), [2] events per time slice
 * (0: one read per wavefront until it is done), [3] chunks in the seed-cluster node pool, [4] max_clusters (a read's allowance x 4) */
void unc_mapper_geometry(const unc_mapper_t *m, uint32_t *out5);
/* diagnostics: device addresses of the mapper's large allocations: [0] slots' base, [1] bytes per slot, [2] node pool's base,
 * [3] its bytes, [4] event means, [5] results */
int unc_mapper_device_addresses(const unc_mapper_t *m, uint64_t *out6);
/* pairs of scheduler rings the mapper settled on (unc_mapper_opts_t.sched_parts); 0: no time slicing (one slot per wavefront) */
uint32_t unc_mapper_sched_parts(const unc_mapper_t *m);
/* The seed-cluster node pool is sized by need.  The reference's SeedTracker is an unbounded std::set per Mapper
 * (src/seed_tracker.hpp:97-110); here its nodes come from ONE pool per mapper.  unc_mapper_create sizes the pool by a rule of
 * thumb for the first batch; after a batch that used less than an eighth of it the library shrinks it to four times the most chunks
 * (192 KB each) that were ever out at once, never below one chunk per slot, and doubles it when a batch found it dry (unless
 * unc_mapper_opts_t.pool_chunks named a size).
 * out4: [0] chunks the pool holds now, [1] high-water mark of the last batch, [2] of all batches, [3] times it was resized. */
int unc_mapper_pool_usage(const unc_mapper_t *m, uint32_t *out4);
/* reads of the last batch that found the seed-cluster node pool dry (or used up their own allowance) and were mapped again
 * after the batch, and the wall-clock milliseconds that took (part of the batch, not of unc_mapper_last_timing) */
void unc_mapper_last_remap(const unc_mapper_t *m, uint32_t *n_reads, float *ms);
/* mean lifetime of the persistent wavefronts of the last batch's k_map launch / the launch duration (both from the
 * device wall clock): 1.0 = every wavefront worked until the end, lower = idle tail behind the longest reads */
double unc_mapper_last_wave_busy(const unc_mapper_t *m);
/* In which order the reads handed to unc_map_batch are to be understood (the reference: N worker threads with one Mapper each,
 * map_pool.cpp:31-42; the ONE piece of Mapper state that Mapper::new_read does not reset is sources_added_, mapper.cpp:88,612-623):
 *   UNC_ORDER_INDEPENDENT  every read as a fresh Mapper maps it (the default).  What `uncalled map -t N` gives for every read
 *                          whose predecessor on its thread left no flag set -- all reads without UNC_NOTE_FLAGS_LEFT neighbours
 *   UNC_ORDER_T1           `uncalled map -t 1`: one Mapper, the reads of a batch in the order given, batch after batch.  A read
 *                          whose predecessor left flags set is mapped again starting from them (and so on down the chain);
 *                          unc_mapper_last_carry_over tells how many reads of the last batch that took, in how many rounds and ms.
 * Setting the order starts a new run (the carried flags are cleared). */
#define UNC_ORDER_INDEPENDENT 0
#define UNC_ORDER_T1 1
int unc_mapper_set_read_order(unc_mapper_t *m, int order);
int unc_mapper_last_carry_over(const unc_mapper_t *m, uint32_t *reads, uint32_t *rounds, float *ms);
/* what the code object says about the k_map instantiation this mapper launches (hipFuncGetAttributes): [0] VGPRs, [1] scratch
 * bytes per lane (spills), [2] static LDS bytes per wavefront, [3] max threads per block, [4] wavefronts per CU the launch
 * bounds are set for, [5] 1 = the 32-bit-row / merged-run instantiation (references below 2^32 rows) */
int unc_mapper_kernel_info(const unc_mapper_t *m, uint32_t *out6);

/* ---- chunked (realtime) path: replaces RealtimePool + per-channel Mapper::new_read(Chunk&) / add_chunk /
 * process_chunk / map_chunk (realtime_pool.cpp:74-142,349-358; mapper.cpp:210-431) with the deterministic
 * semantics of MapPoolOrd (map_pool_ord.cpp:61-112; timeouts disabled as Conf(Mode::MAP_ORD), conf.hpp:88-91):
 * one call processes at most one chunk per channel and maps every chunk completely before returning, so a
 * caller adds the next chunk of a read only after the previous one is mapped.  Per-channel state (streaming
 * event detector, EventProfiler window, rolling Normalizer -- which survives across reads, mapper.cpp:225-226 --
 * path buffers, seed clusters) stays resident in HBM between calls. */
#define UNC_RT_FIRST 1u   /* first chunk of a read: Mapper::new_read(Chunk&) (mapper.cpp:210-217) */
#define UNC_RT_LAST 2u    /* no chunk follows: a read still unmapped afterwards is given up (request_reset, realtime_pool.cpp:115-123) */
typedef struct {
    uint32_t channel;       /* 0-based channel index (Chunk::get_channel_idx) */
    uint32_t read_number;   /* Chunk::get_number */
    uint32_t flags;         /* UNC_RT_FIRST | UNC_RT_LAST */
    uint32_t n_samples;     /* <= chunk_time * sample_rate (4000) */
    uint64_t offset;        /* first sample in `raw` */
    unc_calib_t calib;      /* the channel's calibration */
    uint32_t pad;
} unc_rt_chunk_t;
#define UNC_RT_MAPPING 0    /* chunk fully mapped, read not decided yet (Mapper::chunk_mapped) */
#define UNC_RT_MAPPED 1     /* State::SUCCESS: hit holds the PAF record */
#define UNC_RT_FAILED 2     /* State::FAILURE (max_events, max_chunks, or given up): hit holds the unmapped record */
#define UNC_RT_IGNORED 3    /* chunk of a read this channel is not mapping (already finished / never started) */
typedef struct {
    int32_t state;
    int32_t ended;          /* Paf::is_ended (set_ended, mapper.cpp:389) */
    unc_hit_t hit;
} unc_rt_result_t;
typedef struct unc_rt unc_rt_t;
int unc_rt_create(const unc_index_t *ix, const unc_params_t *p, uint32_t n_channels, unc_rt_t **out);
void unc_rt_free(unc_rt_t *rt);
uint64_t unc_rt_device_bytes(const unc_rt_t *rt);
/* raw: int16 samples (host, or device when on_device != 0); chunks/results: host arrays of n_chunks */
int unc_rt_process_chunks(unc_rt_t *rt, uint32_t n_chunks, const unc_rt_chunk_t *chunks, const int16_t *raw, int on_device,
                          void *stream, unc_rt_result_t *results);
/* the same for chunks that already hold floats -- what the reference's Chunk keeps (src/chunk.cpp:27-66: float32 as sent by
 * MinKNOW, or int16/int32 converted WITHOUT calibration) and what Mapper::new_read(Chunk&)/add_chunk hand to the event
 * detector: `offset` indexes `signal`, the chunk's `calib` is ignored */
int unc_rt_process_chunks_f32(unc_rt_t *rt, uint32_t n_chunks, const unc_rt_chunk_t *chunks, const float *signal, int on_device,
                              void *stream, unc_rt_result_t *results);
int unc_rt_last_timing(const unc_rt_t *rt, float *ms_events, float *ms_map);
/* stage tap (parity tests): what a channel's Mapper carries between chunks below the PAF -- EventDetector counters
 * (event_detector.hpp:106-128), the EventProfiler's window and queue (event_profiler.hpp:35-48) and the rolling Normalizer
 * (normalizer.hpp:74-79) with its ring.  ring (host, 6000 floats) receives the channel's ring, slots [0, norm_n) are in use. */
typedef struct {
    uint32_t det_t, det_total_events;
    float det_len_sum;
    uint32_t norm_n, norm_wr;
    uint32_t prof_n, prof_to_mask, prof_queued;
    double norm_mean, norm_varsum;
    double prof_mean, prof_varsum;
    float prof_queue[28];               /* the queued event means, oldest first (at most 26) */
} unc_rt_tap_t;
int unc_rt_tap_channel(unc_rt_t *rt, uint32_t channel, unc_rt_tap_t *out, float *ring);

/* ---- index building (`uncalled index`; replaces the suffix sort inside bwa_idx_build, src/bwa_index.hpp:92-101)
 * Stable LSD radix sort of n (key, value) pairs of 64 bits each, ascending by the low key_bits bits of the key (1..64; the bits
 * above must be zero), on device memory: keys / vals are sorted in place, tmp_keys / tmp_vals (n words each) are scratch.
 * iota != 0: the values are taken to be 0 .. n - 1 (vals need not be initialised): vals returns the sorting permutation.
 * n < 2^32.  stream: a hipStream_t or NULL. */
int unc_sort_pairs_u64(int device, uint64_t n, uint64_t *keys, uint64_t *vals, uint64_t *tmp_keys, uint64_t *tmp_vals, int key_bits,
                       int iota, void *stream);
/* The suffix array of a text of n < 2^31 symbols (codes 0..3 in host memory) into sa[0 .. n) (host memory): the suffix sort inside
 * bwa_idx_build (src/bwa_index.hpp:92-101) as prefix doubling on the device -- the radix sort above and the steps between the sorts as
 * HIP kernels (k_sort.hip), no torch.  uncalled_amd/build_index.py writes the five BWA-format index files around it. */
int unc_build_suffix_array(int device, const uint8_t *codes, uint64_t n, int64_t *sa);

/* ---- measurement aid: `reps` launches that write, then `reps` that read, n_records (made odd) scattered 64-byte records with
 * one lane per record and four 16-byte accesses per lane -- k_map's access shape with an exactly known byte count, for
 * calibrating the HBM traffic counters of rocprofv3 (tools/dev/pmc_calib.py, profiles/r02_pmc_k_map.json) */
int unc_calib_traffic(int device, uint64_t n_records, int reps);
/* loaded latency of a region of device memory: `waves` (0 = 4096) single-wavefront workgroups, every lane a chain of `steps` (0 = 2000)
 * dependent 16-byte loads over [base, base + bytes); the launch's duration in milliseconds, the best of three (diagnostics) */
int unc_calib_chase(int device, const void *base, uint64_t bytes, uint32_t waves, uint32_t steps, float *ms_out);

/* ---- stage taps (parity tests) */
/* event detection + whole-read normalisation only (EventDetector::get_means, event_detector.cpp:133-145;
 * Normalizer::set_signal, normalizer.cpp:31-44).  means (host) receives the kept event means of read i
 * at means[means_offsets[i] ..]; means_offsets (host, n_reads+1) is filled by the call. */
int unc_detect_events(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                      const unc_calib_t *calib, float *means, uint64_t means_cap, uint64_t *means_offsets,
                      unc_evt_info_t *info);
/* step-wise trace of ONE read (Mapper::map_next, mapper.cpp:433-663): begin, then step until it
 * returns 1; after each step the live path buffer and seed clusters can be read back. */
int unc_trace_begin(unc_mapper_t *m, const int16_t *raw, uint32_t n, const unc_calib_t *calib);
int unc_trace_step(unc_mapper_t *m, uint32_t n_events /* map_next calls to run */, int *done);
int unc_trace_paths(unc_mapper_t *m, unc_path_t *out, uint32_t cap, uint32_t *n_out);
int unc_trace_clusters(unc_mapper_t *m, unc_cluster_t *out, uint32_t cap, uint32_t *n_out, unc_cluster_t *max_map,
                       float *len_sum, uint32_t *n_lens);
int unc_trace_finish(unc_mapper_t *m, unc_hit_t *hit);

#ifdef __cplusplus
}
#endif
#endif
