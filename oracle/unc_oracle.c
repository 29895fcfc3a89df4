#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE
/* TEST INFRASTRUCTURE -- see unc_oracle.h.  Plain C99; build with -ffp-contract=off (the
 * reference is built without FMA: setup.py:121 `-std=c++11 -O3`, x86-64 SSE2). */
#include "unc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "minibwa.h"
#include "r94_model_table.h"

#define KLEN 5
#define KMASK 0x3FFu
#define SEED_LEN UNC_O_SEED_LEN
#define EVT_BUF_LEN 13 /* 1 + window_length2 * 2, event_detector.cpp:30 */

/* ------------------------------------------------------------------ parameters */

void unc_o_params_default(unc_o_params_t *p) {
    /* mapper.cpp:29-40 */
    p->seed_len = 22; p->min_rep_len = 0; p->max_rep_copy = 50; p->max_paths = 10000;
    p->max_consec_stay = 8; p->max_events = 30000; p->max_stay_frac = 0.5f; p->min_seed_prob = -3.75f;
    /* event_detector.cpp:17-26 */
    p->window_length1 = 3; p->window_length2 = 6; p->threshold1 = 1.4f; p->threshold2 = 9.0f;
    p->peak_height = 0.2f; p->min_mean = 0.0f; p->max_mean = 400.0f;
    /* seed_tracker.cpp:28-32 */
    p->min_map_len = 25; p->min_mean_conf = 6.00f; p->min_top_conf = 1.85f;
    /* read_buffer.cpp:26-32 */
    p->bp_per_sec = 450.0f; p->sample_rate = 4000.0f;
}

/* ------------------------------------------------------------------ pore model */

typedef struct {
    float lv_means[UNC_O_NKMER], lv_vars_x2[UNC_O_NKMER], lognorm_denoms[UNC_O_NKMER];
    float model_mean, model_stdv;
} pore_model_t;

static pore_model_t g_model;
static pthread_once_t g_model_once = PTHREAD_ONCE_INIT;

static float bits2f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

/* pore_model.hpp:77-103 (ctor from a means/stdvs vector, cmpl=true: mapper.cpp:57 uses
 * pmodel_r94_complement, model_r94.inl:1036-1037), init_kmer 58-62, init_stdv 48-56 */
static void model_init(void) {
    pore_model_t *m = &g_model;
    m->model_mean = 0;
    for (uint32_t kmer = 0; kmer < UNC_O_NKMER; ++kmer) {
        float mean = bits2f(UNC_R94_MEAN_STDV_BITS[2 * kmer]);
        float stdv = bits2f(UNC_R94_MEAN_STDV_BITS[2 * kmer + 1]);
        uint32_t k = kmer ^ KMASK; /* kmer_comp, bp.hpp:77-80 */
        m->lv_means[k] = mean;
        m->lv_vars_x2[k] = 2 * stdv * stdv;
        m->lognorm_denoms[k] = (float)log(sqrt(M_PI * (double)m->lv_vars_x2[k]));
        m->model_mean += mean;
    }
    m->model_mean /= (float)UNC_O_NKMER;
    m->model_stdv = 0;
    for (uint32_t kmer = 0; kmer < UNC_O_NKMER; ++kmer) {
        float d = m->lv_means[kmer] - m->model_mean;
        m->model_stdv = (float)((double)m->model_stdv + (double)d * (double)d); /* pow(float,2) -> double */
    }
    m->model_stdv = sqrtf(m->model_stdv / (float)UNC_O_NKMER);
}

static const pore_model_t *model_get(void) {
    pthread_once(&g_model_once, model_init);
    return &g_model;
}

/* pore_model.hpp:163-165 */
static inline float match_prob(const pore_model_t *m, float samp, uint32_t kmer) {
    float d = samp - m->lv_means[kmer];
    double q = -((double)d * (double)d) / (double)m->lv_vars_x2[kmer];
    return (float)(q - (double)m->lognorm_denoms[kmer]);
}

void unc_o_model_tables(float *means, float *vars_x2, float *lognorm, float *model_mean, float *model_stdv) {
    const pore_model_t *m = model_get();
    memcpy(means, m->lv_means, sizeof m->lv_means);
    memcpy(vars_x2, m->lv_vars_x2, sizeof m->lv_vars_x2);
    memcpy(lognorm, m->lognorm_denoms, sizeof m->lognorm_denoms);
    *model_mean = m->model_mean;
    *model_stdv = m->model_stdv;
}

void unc_o_match_probs(float level, float *out) {
    const pore_model_t *m = model_get();
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k) out[k] = match_prob(m, level, k);
}

/* ------------------------------------------------------------------ index */

struct unc_o_index {
    bwt_t *bwt;
    bntseq_t *bns;
    uint64_t kmer_start[UNC_O_NKMER], kmer_end[UNC_O_NKMER];
    float prob_threshes[64];
};

/* bwa_index.hpp:158-162 */
static inline void get_neighbor(const unc_o_index_t *ix, uint64_t s, uint64_t e, int base, uint64_t *os, uint64_t *oe) {
    bwtint_t ok, ol;
    bwt_2occ(ix->bwt, s - 1, e, (ubyte_t)base, &ok, &ol);
    *os = ix->bwt->L2[base] + ok + 1;
    *oe = ix->bwt->L2[base] + ol;
}

unc_o_index_t *unc_o_index_load(const char *prefix, const char *preset) {
    char fn[4096];
    unc_o_index_t *ix = (unc_o_index_t *)calloc(1, sizeof *ix);
    /* bwa_index.hpp:116-135 */
    snprintf(fn, sizeof fn, "%s.bwt", prefix);
    ix->bwt = bwt_restore_bwt(fn);
    if (!ix->bwt) { free(ix); return NULL; }
    snprintf(fn, sizeof fn, "%s.sa", prefix);
    bwt_restore_sa(fn, ix->bwt);
    ix->bns = bns_restore(prefix);
    if (!ix->bns) { bwt_destroy(ix->bwt); free(ix); return NULL; }
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k) {
        int head = (k >> (2 * KLEN - 2)) & 3;                          /* kmer_head, bp.hpp:100-103 */
        uint64_t s = ix->bwt->L2[head], e = ix->bwt->L2[head + 1];     /* get_base_range, :172-174 */
        for (int i = 1; i < KLEN; ++i) {
            int b = (k >> (2 * (KLEN - i - 1))) & 3;                   /* kmer_base, bp.hpp:111-114 */
            get_neighbor(ix, s, e, b, &s, &e);
        }
        ix->kmer_start[k] = s;
        ix->kmer_end[k] = e;
    }
    /* mapper.cpp:123-157: "<name>\t<t63,t62,...>\t..." ; remaining lower bins copy the last value */
    snprintf(fn, sizeof fn, "%s.uncl", prefix);
    FILE *fp = fopen(fn, "r");
    if (!fp) { unc_o_index_free(ix); return NULL; }
    char line[65536];
    while (fgets(line, sizeof line, fp)) {
        line[strcspn(line, "\r\n")] = 0;
        char *save1 = NULL;
        char *name = strtok_r(line, "\t", &save1);
        char *fn_str = strtok_r(NULL, "\t", &save1);
        if (!name || !fn_str) continue;
        if (preset && preset[0] && strcmp(name, preset) != 0) continue;
        uint8_t fmbin = 63;
        char *save2 = NULL;
        for (char *tok = strtok_r(fn_str, ",", &save2); tok; tok = strtok_r(NULL, ",", &save2)) {
            ix->prob_threshes[fmbin] = (float)atof(tok);
            fmbin--;
        }
        for (; fmbin < 64; fmbin--) ix->prob_threshes[fmbin] = ix->prob_threshes[fmbin + 1];
    }
    fclose(fp);
    model_get();
    return ix;
}

void unc_o_index_free(unc_o_index_t *ix) {
    if (!ix) return;
    bwt_destroy(ix->bwt);
    bns_destroy(ix->bns);
    free(ix);
}

uint64_t unc_o_index_size(const unc_o_index_t *ix) { return ix->bwt->seq_len; }
const char *unc_o_index_ref_name(const unc_o_index_t *ix, int rid) {
    return (rid >= 0 && rid < ix->bns->n_seqs) ? ix->bns->anns[rid].name : "";
}
void unc_o_index_kmer_ranges(const unc_o_index_t *ix, uint64_t *out) {
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k) { out[2 * k] = ix->kmer_start[k]; out[2 * k + 1] = ix->kmer_end[k]; }
}
void unc_o_index_thresholds(const unc_o_index_t *ix, float *out) { memcpy(out, ix->prob_threshes, sizeof ix->prob_threshes); }
void unc_o_index_get_neighbor(const unc_o_index_t *ix, uint64_t s, uint64_t e, int base, uint64_t *os, uint64_t *oe) {
    get_neighbor(ix, s, e, base, os, oe);
}
uint64_t unc_o_index_sa(const unc_o_index_t *ix, uint64_t k) { return bwt_sa(ix->bwt, k); }

/* ------------------------------------------------------------------ calibration */

/* read_buffer.cpp:239-241: the loop variable is u16, so the i16 sample is reinterpreted */
void unc_o_calibrate(const int16_t *raw, uint64_t n, float range, float offset, float digitisation, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        uint16_t r = (uint16_t)raw[i];
        float t1 = (float)(int)r + offset;
        float t2 = range * t1;
        out[i] = t2 / digitisation;
    }
}

/* ------------------------------------------------------------------ event detector */

typedef struct {
    int32_t DEF_PEAK_POS;
    float DEF_PEAK_VAL;
    float threshold;
    uint32_t window_length;
    uint32_t masked_to;
    int32_t peak_pos;
    float peak_value;
    int valid_peak;
} detector_t;

typedef struct {
    const unc_o_params_t *P;
    double sum[EVT_BUF_LEN], sumsq[EVT_BUF_LEN];
    uint32_t t, buf_mid, evt_st;
    double evt_st_sum, evt_st_sumsq;
    unc_o_event_t event;
    float len_sum;
    uint32_t total_events;
    detector_t short_detector, long_detector;
} evdt_t;

/* event_detector.cpp:47-77.  The ring is NOT cleared by reset() in the reference; only slot 0
 * is.  Stale slots are never read before being rewritten (see compute_tstat's guards), and this
 * restatement zeroes them once at construction so that runs are reproducible. */
static void evdt_reset(evdt_t *d) {
    d->sum[0] = d->sumsq[0] = 0.0;
    d->t = 1;
    d->evt_st = 0;
    d->evt_st_sum = d->evt_st_sumsq = 0.0;
    d->len_sum = 0;
    d->total_events = 0;
    detector_t s = { -1, FLT_MAX, d->P->threshold1, d->P->window_length1, 0, -1, FLT_MAX, 0 };
    detector_t l = { -1, FLT_MAX, d->P->threshold2, d->P->window_length2, 0, -1, FLT_MAX, 0 };
    d->short_detector = s;
    d->long_detector = l;
}

/* event_detector.cpp:174-219 */
static float compute_tstat(evdt_t *d, uint32_t w_length) {
    const float eta = FLT_MIN;
    const float w_lengthf = (float)w_length;
    if (d->t <= 2 * w_length || w_length < 2) return 0;
    uint32_t i = d->buf_mid % EVT_BUF_LEN, st = (d->buf_mid - w_length) % EVT_BUF_LEN,
             en = (d->buf_mid + w_length) % EVT_BUF_LEN;
    double sum1 = d->sum[i] - d->sum[st];
    double sumsq1 = d->sumsq[i] - d->sumsq[st];
    float sum2 = (float)(d->sum[en] - d->sum[i]);
    float sumsq2 = (float)(d->sumsq[en] - d->sumsq[i]);
    float mean1 = (float)(sum1 / (double)w_lengthf);
    float mean2 = sum2 / w_lengthf;
    float m1sq = mean1 * mean1, q2 = sumsq2 / w_lengthf, m2sq = mean2 * mean2;
    float combined_var = (float)(((sumsq1 / (double)w_lengthf - (double)m1sq) + (double)q2) - (double)m2sq);
    combined_var = fmaxf(combined_var, eta);
    const float delta_mean = mean2 - mean1;
    /* <math.h> in C++: fabs/sqrt pick the float overloads here */
    return fabsf(delta_mean) / sqrtf(combined_var / w_lengthf);
}

/* event_detector.cpp:221-279 */
static int peak_detect(evdt_t *d, float current_value, detector_t *det) {
    if (det->masked_to >= d->buf_mid) return 0;
    if (det->peak_pos == det->DEF_PEAK_POS) {
        if (current_value < det->peak_value) {
            det->peak_value = current_value;
        } else if (current_value - det->peak_value > d->P->peak_height) {
            det->peak_value = current_value;
            det->peak_pos = (int32_t)d->buf_mid;
        }
    } else {
        if (current_value > det->peak_value) {
            det->peak_value = current_value;
            det->peak_pos = (int32_t)d->buf_mid;
        }
        if (det->window_length == d->short_detector.window_length) {
            if (det->peak_value > det->threshold) {
                d->long_detector.masked_to = (uint32_t)det->peak_pos + det->window_length;
                d->long_detector.peak_pos = d->long_detector.DEF_PEAK_POS;
                d->long_detector.peak_value = d->long_detector.DEF_PEAK_VAL;
                d->long_detector.valid_peak = 0;
            }
        }
        if (det->peak_value - current_value > d->P->peak_height && det->peak_value > det->threshold) {
            det->valid_peak = 1;
        }
        if (det->valid_peak && (d->buf_mid - (uint32_t)det->peak_pos) > det->window_length / 2) {
            det->peak_pos = det->DEF_PEAK_POS;
            det->peak_value = current_value;
            det->valid_peak = 0;
            return 1;
        }
    }
    return 0;
}

/* event_detector.cpp:296-319 (calibrate() at 160-162 with cal_offset_=0, cal_coef_=1: 31-32) */
static void create_event(evdt_t *d, uint32_t evt_en) {
    uint32_t evt_en_buf = evt_en % EVT_BUF_LEN;
    d->event.start = d->evt_st;
    d->event.length = (uint32_t)(float)(evt_en - d->evt_st);
    d->event.mean = (float)((d->sum[evt_en_buf] - d->evt_st_sum) / (double)d->event.length);
    const float deltasqr = (float)(d->sumsq[evt_en_buf] - d->evt_st_sumsq);
    const float var = deltasqr / (float)d->event.length - d->event.mean * d->event.mean;
    d->event.stdv = sqrtf(fmaxf(var, 0.0f));
    d->event.mean = (d->event.mean + 0.0f) * 1.0f;
    d->event.stdv = (d->event.stdv + 0.0f) * 1.0f;
    d->evt_st = evt_en;
    d->evt_st_sum = d->sum[evt_en_buf];
    d->evt_st_sumsq = d->sumsq[evt_en_buf];
    d->len_sum += (float)d->event.length;
    d->total_events++;
}

/* event_detector.cpp:83-112 */
static int evdt_add_sample(evdt_t *d, float s) {
    uint32_t t_mod = d->t % EVT_BUF_LEN;
    float ss = s * s;
    if (t_mod > 0) {
        d->sum[t_mod] = d->sum[t_mod - 1] + (double)s;
        d->sumsq[t_mod] = d->sumsq[t_mod - 1] + (double)ss;
    } else {
        d->sum[t_mod] = d->sum[EVT_BUF_LEN - 1] + (double)s;
        d->sumsq[t_mod] = d->sumsq[EVT_BUF_LEN - 1] + (double)ss;
    }
    d->t++;
    d->buf_mid = d->t - (EVT_BUF_LEN / 2) - 1;
    float tstat1 = compute_tstat(d, d->P->window_length1), tstat2 = compute_tstat(d, d->P->window_length2);
    int p1 = peak_detect(d, tstat1, &d->short_detector), p2 = peak_detect(d, tstat2, &d->long_detector);
    if (p1 || p2) {
        create_event(d, d->buf_mid - d->P->window_length1 + 1);
        return d->event.mean >= d->P->min_mean && d->event.mean <= d->P->max_mean;
    }
    return 0;
}

/* event_detector.cpp:151-153 */
static float evdt_mean_event_len(const evdt_t *d) { return d->len_sum / (float)d->total_events; }

uint32_t unc_o_detect_events(const unc_o_params_t *p, const float *signal, uint32_t n, unc_o_event_t *out,
                             uint32_t cap, float *mean_event_len, uint32_t *total_events) {
    evdt_t d;
    memset(&d, 0, sizeof d);
    d.P = p;
    evdt_reset(&d);
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (evdt_add_sample(&d, signal[i])) {
            if (m < cap) out[m] = d.event;
            ++m;
        }
    }
    if (mean_event_len) *mean_event_len = evdt_mean_event_len(&d);
    if (total_events) *total_events = d.total_events;
    return m;
}

/* ------------------------------------------------------------------ normaliser (whole-read mode) */

typedef struct {
    float tgt_mean, tgt_stdv;
    float *signal;
    uint32_t cap;
    double mean, varsum;
    uint32_t n, rd, wr;
    int is_full, is_empty;
} norm_t;

/* normalizer.cpp:31-44 */
static void norm_set_signal(norm_t *nm, const float *sig, uint32_t n) {
    if (n > nm->cap) { nm->signal = (float *)realloc(nm->signal, (size_t)n * 4); nm->cap = n; }
    memcpy(nm->signal, sig, (size_t)n * 4);
    nm->n = n;
    nm->rd = nm->wr = 0;
    nm->is_full = 1;
    nm->is_empty = 0;
    nm->mean = 0;
    for (uint32_t i = 0; i < n; ++i) nm->mean += (double)sig[i];
    nm->mean /= (double)n;
    nm->varsum = 0;
    for (uint32_t i = 0; i < n; ++i) { double e = (double)sig[i] - nm->mean; nm->varsum += e * e; }
}

/* normalizer.cpp:114-118 */
static void norm_scale_shift(const norm_t *nm, float *scale, float *shift) {
    *scale = (float)((double)nm->tgt_stdv / sqrt(nm->varsum / (double)nm->n));
    *shift = (float)((double)nm->tgt_mean - (double)*scale * nm->mean);
}

static float norm_at(const norm_t *nm, uint32_t i) {
    float scale, shift;
    norm_scale_shift(nm, &scale, &shift);
    float prod = scale * nm->signal[i];
    return prod + shift;
}

/* normalizer.cpp:120-128.  With an empty signal the reference computes `% 0`; the restatement
 * defines that case as "immediately empty" (the caller never pops then). */
static float norm_pop(norm_t *nm) {
    float e = norm_at(nm, nm->rd);
    nm->rd = (nm->rd + 1) % nm->n;
    nm->is_empty = nm->rd == nm->wr;
    nm->is_full = 0;
    return e;
}

void unc_o_normalize(const float *means, uint32_t m, float *levels, float *scale, float *shift) {
    const pore_model_t *pm = model_get();
    norm_t nm;
    memset(&nm, 0, sizeof nm);
    nm.tgt_mean = pm->model_mean; /* mapper.cpp:94 */
    nm.tgt_stdv = pm->model_stdv;
    if (m == 0) { if (scale) *scale = 0; if (shift) *shift = 0; return; }
    norm_set_signal(&nm, means, m);
    float sc, sh;
    norm_scale_shift(&nm, &sc, &sh);
    if (scale) *scale = sc;
    if (shift) *shift = sh;
    for (uint32_t i = 0; i < m; ++i) levels[i] = norm_pop(&nm);
    free(nm.signal);
}

/* ------------------------------------------------------------------ seed tracker */

typedef struct {
    uint64_t ref_st;
    uint64_t ref_en_start, ref_en_end;
    uint32_t evt_st, evt_en, total_len;
} cluster_t;

typedef struct {
    const unc_o_params_t *P;
    cluster_t *clusters; /* std::set<SeedCluster>: kept sorted by cluster_less */
    uint32_t n_clusters, cap_clusters;
    uint32_t *lens; /* std::multiset<u32>: kept sorted ascending */
    uint32_t n_lens, cap_lens;
    cluster_t max_map;
    float len_sum;
} tracker_t;

static const cluster_t NULL_ALN = { 0, 1, 0, 1, 0, 0 }; /* seed_tracker.cpp:34-38 (Range() = [1,0]) */

/* seed_tracker.cpp:97-102 */
static int cluster_less(const cluster_t *a, const cluster_t *b) {
    if (a->ref_en_start != b->ref_en_start) return a->ref_en_start > b->ref_en_start;
    return a->evt_en > b->evt_en;
}

static void tracker_reset(tracker_t *t) { /* seed_tracker.cpp:117-122 */
    t->n_clusters = 0;
    t->n_lens = 0;
    t->max_map = NULL_ALN;
    t->len_sum = 0;
}

static uint32_t clusters_lower_bound(const tracker_t *t, const cluster_t *key) {
    uint32_t lo = 0, hi = t->n_clusters;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cluster_less(&t->clusters[mid], key)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* std::set::insert: no-op when an equivalent key exists */
static void clusters_insert(tracker_t *t, const cluster_t *c) {
    uint32_t pos = clusters_lower_bound(t, c);
    if (pos < t->n_clusters && !cluster_less(c, &t->clusters[pos])) return; /* equivalent key */
    if (t->n_clusters == t->cap_clusters) {
        t->cap_clusters = t->cap_clusters ? t->cap_clusters * 2 : 256;
        t->clusters = (cluster_t *)realloc(t->clusters, (size_t)t->cap_clusters * sizeof(cluster_t));
    }
    memmove(&t->clusters[pos + 1], &t->clusters[pos], (size_t)(t->n_clusters - pos) * sizeof(cluster_t));
    t->clusters[pos] = *c;
    t->n_clusters++;
}

static void clusters_erase(tracker_t *t, uint32_t pos) {
    memmove(&t->clusters[pos], &t->clusters[pos + 1], (size_t)(t->n_clusters - pos - 1) * sizeof(cluster_t));
    t->n_clusters--;
}

static void lens_insert(tracker_t *t, uint32_t v) {
    if (t->n_lens == t->cap_lens) {
        t->cap_lens = t->cap_lens ? t->cap_lens * 2 : 256;
        t->lens = (uint32_t *)realloc(t->lens, (size_t)t->cap_lens * 4);
    }
    uint32_t lo = 0, hi = t->n_lens;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (t->lens[mid] <= v) lo = mid + 1; else hi = mid; }
    memmove(&t->lens[lo + 1], &t->lens[lo], (size_t)(t->n_lens - lo) * 4);
    t->lens[lo] = v;
    t->n_lens++;
}

static void lens_erase_one(tracker_t *t, uint32_t v) {
    uint32_t lo = 0, hi = t->n_lens;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (t->lens[mid] < v) lo = mid + 1; else hi = mid; }
    if (lo < t->n_lens && t->lens[lo] == v) {
        memmove(&t->lens[lo], &t->lens[lo + 1], (size_t)(t->n_lens - lo - 1) * 4);
        t->n_lens--;
    }
}

/* seed_tracker.cpp:56-73 */
static uint8_t cluster_update(cluster_t *c, const cluster_t *new_seed) {
    uint8_t growth = 0;
    if (new_seed->ref_en_start < c->ref_en_end) {
        if (new_seed->ref_en_end > c->ref_en_end) {
            growth = (uint8_t)(new_seed->ref_en_end - c->ref_en_end);
            c->ref_en_start = new_seed->ref_en_start;
            c->ref_en_end = new_seed->ref_en_end;
        } else {
            c->ref_en_start = new_seed->ref_en_start;
        }
    } else {
        growth = (uint8_t)new_seed->total_len;
        c->ref_en_start = new_seed->ref_en_start;
        c->ref_en_end = new_seed->ref_en_end;
    }
    c->evt_en = new_seed->evt_en;
    c->total_len += growth;
    return growth;
}

/* seed_tracker.cpp:157-232 */
static void tracker_add_seed(tracker_t *t, uint64_t ref_en, uint32_t ref_len, uint32_t evt_st) {
    cluster_t new_seed; /* SeedCluster(Range(ref_en-ref_len+1, ref_en), evt_st), :40-46 */
    new_seed.ref_st = ref_en - ref_len + 1;
    new_seed.ref_en_start = ref_en - ref_len + 1;
    new_seed.ref_en_end = ref_en;
    new_seed.evt_st = new_seed.evt_en = evt_st;
    new_seed.total_len = (uint32_t)(new_seed.ref_en_end - new_seed.ref_en_start + 1);

    uint32_t loc = clusters_lower_bound(t, &new_seed), loc_match = UINT32_MAX;
    uint64_t e2 = new_seed.evt_en, r2 = new_seed.ref_en_start;
    while (loc != t->n_clusters) {
        uint64_t e1 = t->clusters[loc].evt_en, r1 = t->clusters[loc].ref_en_start;
        int higher_sup = loc_match == UINT32_MAX || t->clusters[loc_match].total_len < t->clusters[loc].total_len;
        int in_range = e1 <= e2 && r2 - r1 <= e2 - e1 && (r2 - r1) >= (e2 - e1) / 12;
        if (higher_sup && in_range) {
            loc_match = loc;
        } else if (r2 - r1 >= e2) {
            break;
        }
        loc++;
    }

    if (loc_match != UINT32_MAX) {
        cluster_t a = t->clusters[loc_match];
        uint32_t prev_len = a.total_len;
        cluster_update(&a, &new_seed);
        if (a.total_len != prev_len) {
            t->len_sum += (float)(a.total_len - prev_len);
            lens_insert(t, a.total_len);
            lens_erase_one(t, prev_len);
            if (a.total_len >= t->P->min_map_len && a.total_len > t->max_map.total_len) t->max_map = a;
        }
        clusters_erase(t, loc_match);
        clusters_insert(t, &a);
    } else {
        lens_insert(t, new_seed.total_len);
        t->len_sum += (float)new_seed.total_len;
        if (new_seed.total_len >= t->P->min_map_len && new_seed.total_len > t->max_map.total_len) t->max_map = new_seed;
        clusters_insert(t, &new_seed);
    }
}

/* seed_tracker.cpp:259-262 */
static int check_map_conf(const tracker_t *t, uint32_t seed_len, float mean_len, float second_len) {
    return (t->P->min_mean_conf > 0 && (float)seed_len / mean_len >= t->P->min_mean_conf) ||
           (t->P->min_top_conf > 0 && (float)seed_len / second_len >= t->P->min_top_conf);
}

/* seed_tracker.cpp:129-143 */
static cluster_t tracker_get_final(const tracker_t *t) {
    if (t->max_map.total_len < t->P->min_map_len || t->n_lens < 2) return NULL_ALN;
    float mean_len = t->len_sum / (float)t->n_clusters;
    float second_len = (float)t->lens[t->n_lens - 2];
    if (check_map_conf(t, t->max_map.total_len, mean_len, second_len)) return t->max_map;
    return NULL_ALN;
}

/* ------------------------------------------------------------------ chunked (realtime / MAP_ORD) path */

#define NORM_LEN 6000     /* Normalizer::PRMS_DEF.len, normalizer.cpp:4-8 */
#define PROF_WIN 25       /* EventProfiler::PRMS_DEF.win_len, event_profiler.cpp:4-10 */

/* Normalizer in rolling mode: reset normalizer.cpp:84-95, push 46-75, pop 120-128, unread_size 131-134,
 * skip_unread 136-152, get_mean/get_stdv 97-103 */
typedef struct {
    float *signal;
    uint32_t size;
    double mean, varsum;
    uint32_t n, rd, wr;
    int is_full, is_empty;
} rnorm_t;

static void rnorm_init(rnorm_t *r, uint32_t size) {
    r->signal = (float *)calloc(size, sizeof(float));
    r->size = size;
    r->n = r->rd = r->wr = 0;
    r->mean = r->varsum = 0;
    r->is_full = 0;
    r->is_empty = 1;
}
static void rnorm_reset(rnorm_t *r) {
    r->n = r->rd = r->wr = 0;
    r->mean = r->varsum = 0;
    r->is_full = 0;
    r->is_empty = 1;
    r->signal[0] = 0;
}
static int rnorm_push(rnorm_t *r, float newevt) {
    if (r->is_full) return 0;
    double oldevt = (double)r->signal[r->wr];
    r->signal[r->wr] = newevt;
    if (r->n == r->size) {
        double oldmean = r->mean;
        r->mean += ((double)newevt - oldevt) / (double)r->size;
        r->varsum += ((double)newevt + oldevt - oldmean - r->mean) * ((double)newevt - oldevt);
    } else {
        r->n++;
        double dt1 = (double)newevt - r->mean;
        r->mean += dt1 / (double)r->n;
        double dt2 = (double)newevt - r->mean;
        r->varsum += dt1 * dt2;
    }
    r->wr = (r->wr + 1) % r->size;
    r->is_empty = 0;
    r->is_full = r->wr == r->rd;
    return 1;
}
static uint32_t rnorm_unread(const rnorm_t *r) {
    if (r->rd < r->wr) return r->wr - r->rd;
    return (r->n - r->rd) + r->wr;
}
static float rnorm_at(const rnorm_t *r, float tgt_mean, float tgt_stdv, uint32_t i) {
    float scale = (float)((double)tgt_stdv / sqrt(r->varsum / (double)r->n));
    float shift = (float)((double)tgt_mean - (double)scale * r->mean);
    float prod = scale * r->signal[i];
    return prod + shift;
}
static void rnorm_pop_advance(rnorm_t *r) {
    r->rd = (r->rd + 1) % r->size;
    r->is_empty = r->rd == r->wr;
    r->is_full = 0;
}
static uint32_t rnorm_skip_unread(rnorm_t *r, uint32_t nkeep) {
    if (nkeep >= rnorm_unread(r)) return 0;
    r->is_full = 0;
    r->is_empty = nkeep == 0;
    uint32_t new_rd;
    if (nkeep <= r->wr) new_rd = r->wr - nkeep;
    else new_rd = r->n - (nkeep - r->wr);
    uint32_t nskip;
    if (new_rd > r->rd) nskip = new_rd - r->rd;
    else nskip = (r->n - r->rd) + new_rd;
    r->rd = new_rd;
    return nskip;
}

/* EventProfiler, event_profiler.hpp:35-104 (a 25-long rolling Normalizer + a deque of the same events) */
typedef struct {
    rnorm_t window;
    float evq[PROF_WIN + 1];
    uint32_t q_head, q_len;
    float next_mean;
    int is_full;
    uint32_t to_mask;
    float win_stdv_min;
} evprof_t;

static void evprof_reset(evprof_t *p) {
    rnorm_reset(&p->window);
    p->q_head = p->q_len = 0;
    p->next_mean = 0;
    p->is_full = 0;
    p->to_mask = 0;
}

/* returns event_ready() */
static int evprof_add_event(evprof_t *p, float mean) {
    rnorm_push(&p->window, mean);
    p->evq[(p->q_head + p->q_len) % (PROF_WIN + 1)] = mean;
    p->q_len++;
    if (rnorm_unread(&p->window) <= PROF_WIN / 2) return 0;
    float win_stdv = (float)sqrt(p->window.varsum / (double)p->window.n);   /* Normalizer::get_stdv */
    if (win_stdv < p->win_stdv_min) {
        p->to_mask = PROF_WIN - 1;
    } else if (p->to_mask > 0) {
        p->to_mask--;
    }
    if (p->window.is_full) {
        p->next_mean = p->evq[p->q_head];
        p->q_head = (p->q_head + 1) % (PROF_WIN + 1);
        p->q_len--;
        rnorm_pop_advance(&p->window);     /* window_.pop(): the value is discarded */
        p->is_full = 1;
    }
    return p->is_full && p->to_mask == 0;
}

typedef struct {
    rnorm_t norm;
    evprof_t prof;
    int inited;
    /* ReadBuffer chunk bookkeeping, read_buffer.cpp:248-292 */
    uint64_t raw_len;
    uint32_t chunk_count;
    int chunk_processed, finished, ended, active;
    const float *chunk;
    uint32_t chunk_n;
} rt_state_t;

/* ------------------------------------------------------------------ mapper */

typedef struct {
    uint64_t fm_start, fm_end;
    uint8_t length, consec_stays;
    uint32_t event_moves;
    uint16_t kmer;
    float seed_prob;
    float prob_sums[SEED_LEN + 1];
    uint8_t sa_checked;
    uint32_t order; /* position at creation: stable-sort tie-break (oracle/shim/pdqsort.h) */
} path_t;

struct unc_o_mapper {
    const unc_o_index_t *ix;
    const pore_model_t *model;
    unc_o_params_t P;
    evdt_t evdt;
    norm_t norm;
    tracker_t tracker;
    float kmer_probs[UNC_O_NKMER];
    path_t *prev_paths, *next_paths, *sort_tmp;
    uint8_t sources_added[UNC_O_NKMER];
    uint32_t prev_size, event_i;
    int state_success;
    float *means;
    uint32_t means_cap, n_means;
    uint64_t raw_len;
    unc_o_hit_t hit;
    minibwa_counters_t c0;
    uint64_t stat_parents, stat_children, stat_seeds;
    uint32_t stat_max_children;
    int rt_mode, reset_req;
    uint32_t max_chunks;
    rt_state_t rt;
};

unc_o_mapper_t *unc_o_mapper_new(const unc_o_index_t *ix, const unc_o_params_t *p) {
    if (p->seed_len != SEED_LEN || p->window_length2 * 2 + 1 != EVT_BUF_LEN) return NULL;
    unc_o_mapper_t *m = (unc_o_mapper_t *)calloc(1, sizeof *m);
    m->ix = ix;
    m->model = model_get();
    m->P = *p;
    m->evdt.P = &m->P;
    m->tracker.P = &m->P;
    m->norm.tgt_mean = m->model->model_mean; /* mapper.cpp:94 */
    m->norm.tgt_stdv = m->model->model_stdv;
    m->prev_paths = (path_t *)calloc(p->max_paths, sizeof(path_t));
    m->next_paths = (path_t *)calloc(p->max_paths, sizeof(path_t));
    m->sort_tmp = (path_t *)calloc(p->max_paths, sizeof(path_t));
    tracker_reset(&m->tracker);
    return m;
}

void unc_o_mapper_free(unc_o_mapper_t *m) {
    if (!m) return;
    free(m->prev_paths); free(m->next_paths); free(m->sort_tmp);
    free(m->norm.signal); free(m->tracker.clusters); free(m->tracker.lens); free(m->means);
    if (m->rt.inited) { free(m->rt.norm.signal); free(m->rt.prof.window.signal); }
    free(m);
}

static inline uint64_t range_len(const path_t *p) { return p->fm_end - p->fm_start + 1; }

/* mapper.cpp:751-772 */
static void make_source(path_t *p, uint64_t s, uint64_t e, uint16_t kmer, float prob) {
    p->length = 1;
    p->consec_stays = 0;
    p->event_moves = 1;
    p->seed_prob = prob;
    p->fm_start = s; p->fm_end = e;
    p->kmer = kmer;
    p->sa_checked = 0;
    p->prob_sums[0] = 0;
    p->prob_sums[1] = prob;
}

/* mapper.cpp:775-807 */
static void make_child(path_t *c, const path_t *p, uint64_t s, uint64_t e, uint16_t kmer, float prob, uint8_t move) {
    const uint32_t PATH_MASK = (1u << SEED_LEN) - 1, PATH_TAIL_MOVE = 1u << (SEED_LEN - 1);
    uint8_t stay = 1 - move;
    c->length = p->length + (p->length < SEED_LEN);
    c->fm_start = s; c->fm_end = e;
    c->kmer = kmer;
    c->sa_checked = p->sa_checked;
    c->event_moves = ((p->event_moves << 1) | move) & PATH_MASK;
    c->consec_stays = (uint8_t)((p->consec_stays + stay) * stay);
    if (p->length == SEED_LEN) {
        memcpy(c->prob_sums, &p->prob_sums[1], SEED_LEN * sizeof(float));
        c->prob_sums[SEED_LEN] = c->prob_sums[SEED_LEN - 1] + prob;
        c->seed_prob = (c->prob_sums[SEED_LEN] - c->prob_sums[0]) / (float)SEED_LEN;
        c->event_moves |= PATH_TAIL_MOVE;
    } else {
        memcpy(c->prob_sums, p->prob_sums, c->length * sizeof(float));
        c->prob_sums[c->length] = c->prob_sums[c->length - 1] + prob;
        c->seed_prob = c->prob_sums[c->length] / (float)c->length;
    }
}

/* mapper.cpp:842-863 */
static int is_seed_valid(const unc_o_params_t *P, const path_t *p, int path_ended) {
    uint8_t move_count = (uint8_t)__builtin_popcount(p->event_moves);
    uint8_t stay_count = (uint8_t)(p->length - move_count);
    return (p->length == P->seed_len && p->seed_prob >= P->min_seed_prob) &&
           ((range_len(p) == 1 && (p->event_moves & 1) == 1 &&
             (float)stay_count <= P->max_stay_frac * (float)P->seed_len) ||
            (path_ended && range_len(p) <= P->max_rep_copy && move_count >= P->min_rep_len));
}

/* mapper.cpp:665-700 */
static void update_seeds(unc_o_mapper_t *m, path_t *p, int path_ended) {
    if (!is_seed_valid(&m->P, p, path_ended)) return;
    p->sa_checked = 1;
    uint32_t move_count = (uint8_t)__builtin_popcount(p->event_moves);
    for (uint64_t s = p->fm_start; s <= p->fm_end; ++s) {
        uint64_t sa_end = m->ix->bwt->seq_len - bwt_sa(m->ix->bwt, s);
        tracker_add_seed(&m->tracker, sa_end, move_count, m->event_i - (uint32_t)path_ended);
        m->stat_seeds++;
    }
}

/* mapper.cpp:866-871 + range.cpp:112-115, with the project's tie-break (creation order) */
static int path_less(const path_t *a, const path_t *b) {
    if (a->fm_start != b->fm_start) return a->fm_start < b->fm_start;
    if (a->fm_end != b->fm_end) return a->fm_end < b->fm_end;
    if (a->seed_prob < b->seed_prob) return 1;
    if (b->seed_prob < a->seed_prob) return 0;
    return a->order < b->order;
}

static void sort_paths(path_t *a, path_t *tmp, uint32_t n) { /* bottom-up merge sort */
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint32_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = path_less(&a[j], &a[i]) ? a[j++] : a[i++];
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(path_t));
    }
}

static inline int fm_bin(uint64_t len) { return __builtin_clzll(len); } /* mapper.cpp:161-163 */

/* mapper.cpp:703-706.  The float -> u32 conversion is out of range when evt_st < seed_len
 * wraps (:716); x86-64 GCC emits cvttss2si r64 and keeps the low 32 bits, restated explicitly. */
static uint32_t event_to_bp(const unc_o_mapper_t *m, uint32_t evt_i, int last) {
    float bp_per_samp = m->P.bp_per_sec / m->P.sample_rate;
    float v = ((float)evt_i * evdt_mean_event_len(&m->evdt)) * bp_per_samp + (float)(last * (KLEN - 1));
    return (uint32_t)(int64_t)v;
}

/* mapper.cpp:708-728 + bwa_index.hpp:213-220 + read_buffer.cpp:142-155 */
static void set_ref_loc(unc_o_mapper_t *m, const cluster_t *seeds) {
    const unc_o_index_t *ix = m->ix;
    uint64_t size = ix->bwt->seq_len;
    int fwd = seeds->ref_st < size / 2;
    uint64_t sa_st = fwd ? seeds->ref_st : size - (seeds->ref_en_end + KLEN - 1);
    uint64_t rd_st = event_to_bp(m, seeds->evt_st - m->P.seed_len, 0), rd_en = event_to_bp(m, seeds->evt_en, 1),
             rd_len = event_to_bp(m, m->event_i, 1), rf_st = 0, rf_len = 0;
    int rid = bns_pos2rid(ix->bns, (int64_t)sa_st);
    if (rid >= 0) {
        rf_st = sa_st - (uint64_t)ix->bns->anns[rid].offset;
        rf_len = (uint64_t)ix->bns->anns[rid].len;
    }
    uint64_t rf_en = rf_st + (seeds->ref_en_end - seeds->ref_st + KLEN);
    uint16_t match_count = (uint16_t)(seeds->total_len + KLEN - 1);
    unc_o_hit_t *h = &m->hit;
    h->mapped = 1; h->fwd = fwd; h->rid = rid; h->matches = match_count;
    h->rd_st = rd_st; h->rd_en = rd_en; h->rd_len = rd_len;
    h->rf_st = rf_st; h->rf_en = rf_en; h->rf_len = rf_len;
    h->cluster.ref_st = seeds->ref_st; h->cluster.ref_en_start = seeds->ref_en_start;
    h->cluster.ref_en_end = seeds->ref_en_end; h->cluster.evt_st = seeds->evt_st;
    h->cluster.evt_en = seeds->evt_en; h->cluster.total_len = seeds->total_len;
}

/* mapper.cpp:433-663.  Returns 1 when the read is finished (SUCCESS or FAILURE). */
static int map_next(unc_o_mapper_t *m) {
    const unc_o_index_t *ix = m->ix;
    const uint32_t max_paths = m->P.max_paths;
    float event;
    if (m->rt_mode) {   /* rolling normaliser of the chunked path (persists across reads, mapper.cpp:225-226) */
        if (m->rt.norm.is_empty || m->reset_req || m->event_i >= m->P.max_events) return 1;
        event = rnorm_at(&m->rt.norm, m->norm.tgt_mean, m->norm.tgt_stdv, m->rt.norm.rd);
        rnorm_pop_advance(&m->rt.norm);
    } else {
        if (m->norm.is_empty || m->event_i >= m->P.max_events) return 1; /* State::FAILURE */
        event = norm_pop(&m->norm);
    }
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k) m->kmer_probs[k] = match_prob(m->model, event, k);
    const float *kp = m->kmer_probs;
    const float source_prob = ix->prob_threshes[0]; /* mapper.cpp:169-171 */

    path_t *next = m->next_paths;
    uint32_t nn = 0; /* next_path - next_paths_.begin() */

    /* :455-524 extend previous paths */
    for (uint32_t pi = 0; pi < m->prev_size; ++pi) {
        path_t *prev = &m->prev_paths[pi];
        if (prev->length == 0) continue;
        m->stat_parents++;
        int child_found = 0;
        uint16_t prev_kmer = prev->kmer;
        float thresh = ix->prob_threshes[fm_bin(range_len(prev))];

        if (prev->consec_stays < m->P.max_consec_stay && kp[prev_kmer] >= thresh) {
            make_child(&next[nn], prev, prev->fm_start, prev->fm_end, prev_kmer, kp[prev_kmer], 0);
            next[nn].order = nn;
            child_found = 1;
            if (++nn == max_paths) break;
        }
        for (int b = 0; b < 4; ++b) {
            uint16_t next_kmer = (uint16_t)(((prev_kmer << 2) & KMASK) | b); /* kmer_neighbor, bp.hpp:105-108 */
            if (kp[next_kmer] < thresh) continue;
            uint64_t ns, ne;
            get_neighbor(ix, prev->fm_start, prev->fm_end, b, &ns, &ne);
            if (!(ns <= ne)) continue;
            make_child(&next[nn], prev, ns, ne, next_kmer, kp[next_kmer], 1);
            next[nn].order = nn;
            child_found = 1;
            if (++nn == max_paths) break;
        }
        if (!child_found && !prev->sa_checked) update_seeds(m, prev, 1);
        if (nn == max_paths) break;
    }

    /* :527-603 sources between the gaps */
    if (nn != 0) {
        uint32_t next_size = nn;
        m->stat_children += next_size;
        if (next_size > m->stat_max_children) m->stat_max_children = next_size;
        sort_paths(next, m->sort_tmp, next_size);
        uint32_t prev_kmer = UNC_O_NKMER;
        uint64_t unchecked_s = 1, unchecked_e = 0, source_s, source_e;
        for (uint32_t i = 0; i < next_size; ++i) {
            uint16_t source_kmer = next[i].kmer;
            if (source_kmer != prev_kmer && nn != max_paths && kp[source_kmer] >= source_prob) {
                m->sources_added[source_kmer] = 1;
                source_s = ix->kmer_start[source_kmer];
                source_e = next[i].fm_start - 1;
                if (source_s <= source_e) {
                    make_source(&next[nn], source_s, source_e, source_kmer, kp[source_kmer]);
                    nn++;
                }
                unchecked_s = next[i].fm_end + 1;
                unchecked_e = ix->kmer_end[source_kmer];
            }
            prev_kmer = source_kmer;

            if (i < next_size - 1 && next[i].fm_start == next[i + 1].fm_start && next[i].fm_end == next[i + 1].fm_end) {
                next[i].length = 0; /* invalidate */
                continue;
            }

            if (nn != max_paths && kp[source_kmer] >= source_prob) {
                source_s = unchecked_s;
                source_e = unchecked_e;
                if (i < next_size - 1 && source_kmer == next[i + 1].kmer) {
                    source_e = next[i + 1].fm_start - 1;
                    if (unchecked_s <= next[i + 1].fm_end) unchecked_s = next[i + 1].fm_end + 1;
                }
                if (source_s <= source_e) {
                    make_source(&next[nn], source_s, source_e, source_kmer, kp[source_kmer]);
                    nn++;
                }
            }
            update_seeds(m, &next[i], 0);
        }
    }

    /* :605-624 remaining full-range sources */
    for (uint32_t kmer = 0; kmer < UNC_O_NKMER && nn != max_paths; ++kmer) {
        uint64_t rs = ix->kmer_start[kmer], re = ix->kmer_end[kmer];
        if (!m->sources_added[kmer] && kp[kmer] >= source_prob && nn != max_paths && rs <= re) {
            make_source(&next[nn], rs, re, (uint16_t)kmer, kp[kmer]);
            nn++;
        } else {
            m->sources_added[kmer] = 0;
        }
    }

    if (nn == max_paths) m->hit.notes |= UNC_O_NOTE_PATHS_FULL;   /* next_path == next_paths_.end(): a child or a source may have been left out */
    m->prev_size = nn;
    path_t *tmp = m->prev_paths; m->prev_paths = m->next_paths; m->next_paths = tmp;

    cluster_t sc = tracker_get_final(&m->tracker);
    if (sc.evt_st <= sc.evt_en) { /* SeedCluster::is_valid */
        set_ref_loc(m, &sc);
        m->state_success = 1;
        return 1;
    }
    m->event_i++;
    return 0;
}

/* mapper.cpp:202-207,219-246 (new_read/reset) + 188-200 (map_read prologue) */
void unc_o_trace_begin(unc_o_mapper_t *m, const float *signal, uint32_t n) {
    m->prev_size = 0;
    m->event_i = 0;
    m->state_success = 0;
    tracker_reset(&m->tracker);
    m->stat_parents = m->stat_children = m->stat_seeds = 0;
    m->stat_max_children = 0;
    memset(&m->hit, 0, sizeof m->hit);
    m->hit.rid = -1;
    m->raw_len = n;
    minibwa_counters_get(&m->c0);

    /* evdt_.get_means(full_signal_), event_detector.cpp:133-145 */
    if (m->means_cap < n) { m->means = (float *)realloc(m->means, (size_t)(n ? n : 1) * 4); m->means_cap = n; }
    evdt_reset(&m->evdt);
    m->n_means = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (evdt_add_sample(&m->evdt, signal[i])) m->means[m->n_means++] = m->evdt.event.mean;
    if (m->n_means) {
        norm_set_signal(&m->norm, m->means, m->n_means);
    } else {
        m->norm.n = 0; m->norm.is_empty = 1; /* reference behaviour undefined (x % 0); defined here as FAILURE */
    }
}

int unc_o_trace_step(unc_o_mapper_t *m) { return map_next(m); }

void unc_o_trace_finish(unc_o_mapper_t *m, unc_o_hit_t *out) {
    unc_o_hit_t *h = &m->hit;
    if (!m->state_success) {
        /* ReadBuffer::set_raw_len, read_buffer.cpp:264-267: u64 * float -> float -> u64 */
        float bp_per_samp = m->P.bp_per_sec / m->P.sample_rate;
        h->rd_len = (uint64_t)((float)m->raw_len * bp_per_samp);
    }
    h->n_events = m->n_means;
    h->event_i = m->event_i;
    h->mean_event_len = m->evdt.total_events ? evdt_mean_event_len(&m->evdt) : 0.0f;
    /* sources_added_ is not cleared between reads (mapper.cpp:88,612-623): what is set now is seen by the Mapper's next read */
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k)
        if (m->sources_added[k]) { h->notes |= UNC_O_NOTE_FLAGS_LEFT; break; }
    minibwa_counters_t c1;
    minibwa_counters_get(&c1);
    h->n_nbr = c1.n_2occ - m->c0.n_2occ;
    h->n_sa = c1.n_sa - m->c0.n_sa;
    h->n_lf = c1.n_lf - m->c0.n_lf;
    *out = *h;
}

int unc_o_map_read(unc_o_mapper_t *m, const float *signal, uint32_t n, unc_o_hit_t *out) {
    unc_o_trace_begin(m, signal, n);
    while (!map_next(m)) {}
    unc_o_trace_finish(m, out);
    return 0;
}

uint32_t unc_o_trace_paths(const unc_o_mapper_t *m, unc_o_path_t *out, uint32_t cap) {
    for (uint32_t i = 0; i < m->prev_size && i < cap; ++i) {
        const path_t *p = &m->prev_paths[i];
        unc_o_path_t *o = &out[i];
        memset(o, 0, sizeof *o);
        o->fm_start = p->fm_start; o->fm_end = p->fm_end; o->event_moves = p->event_moves;
        o->seed_prob = p->seed_prob; o->kmer = p->kmer; o->length = p->length;
        o->consec_stays = p->consec_stays; o->sa_checked = p->sa_checked;
        for (int j = 0; j <= p->length && j <= SEED_LEN; ++j) o->prob_sums[j] = p->prob_sums[j];
    }
    return m->prev_size;
}

static void cluster_out(unc_o_cluster_t *o, const cluster_t *c) {
    o->ref_st = c->ref_st; o->ref_en_start = c->ref_en_start; o->ref_en_end = c->ref_en_end;
    o->evt_st = c->evt_st; o->evt_en = c->evt_en; o->total_len = c->total_len; o->pad = 0;
}

uint32_t unc_o_trace_clusters(const unc_o_mapper_t *m, unc_o_cluster_t *out, uint32_t cap, unc_o_cluster_t *max_map,
                              float *len_sum, uint32_t *n_lens) {
    const tracker_t *t = &m->tracker;
    for (uint32_t i = 0; i < t->n_clusters && i < cap; ++i) cluster_out(&out[i], &t->clusters[i]);
    if (max_map) cluster_out(max_map, &t->max_map);
    if (len_sum) *len_sum = t->len_sum;
    if (n_lens) *n_lens = t->n_lens;
    return t->n_clusters;
}

uint32_t unc_o_trace_event_i(const unc_o_mapper_t *m) { return m->event_i; }

void unc_o_stats(const unc_o_mapper_t *m, uint64_t *sum_parents, uint64_t *sum_children, uint32_t *max_children,
                 uint64_t *n_seeds) {
    if (sum_parents) *sum_parents = m->stat_parents;
    if (sum_children) *sum_children = m->stat_children;
    if (max_children) *max_children = m->stat_max_children;
    if (n_seeds) *n_seeds = m->stat_seeds;
}

/* ---- Mapper's chunk API, mapper.cpp:210-431, driven the way MapPoolOrd does (map_pool_ord.cpp:61-112 +
 * realtime_pool.cpp:112-142,349-358): a chunk is added only after the previous one is fully mapped, timeouts are
 * infinite (conf.hpp:88-91), an exhausted read is given up with request_reset(). */

static void rt_set_failed(unc_o_mapper_t *m) { m->rt.finished = 1; m->reset_req = 0; }

static void rt_new_read(unc_o_mapper_t *m, const float *chunk, uint32_t n) {
    if (!m->rt.inited) {
        rnorm_init(&m->rt.norm, NORM_LEN);
        rnorm_init(&m->rt.prof.window, PROF_WIN);
        rnorm_reset(&m->rt.norm);
        m->rt.prof.win_stdv_min = 5.0f;
        m->rt.inited = 1;
    }
    /* Mapper::reset, mapper.cpp:219-246 */
    m->prev_size = 0;
    m->event_i = 0;
    m->reset_req = 0;
    m->state_success = 0;
    rnorm_skip_unread(&m->rt.norm, 0);
    tracker_reset(&m->tracker);
    evdt_reset(&m->evdt);
    evprof_reset(&m->rt.prof);
    memset(&m->hit, 0, sizeof m->hit);
    m->hit.rid = -1;
    minibwa_counters_get(&m->c0);
    m->stat_parents = m->stat_children = m->stat_seeds = 0;
    m->stat_max_children = 0;
    /* ReadBuffer(Chunk&), read_buffer.cpp:248-260 */
    m->rt.chunk_count = 1;
    m->rt.chunk_processed = 0;
    m->rt.finished = m->rt.ended = 0;
    m->rt.raw_len = n;
    m->rt.chunk = chunk;
    m->rt.chunk_n = n;
    m->rt.active = 1;
}

/* Mapper::add_chunk 281-305 + ReadBuffer::add_chunk read_buffer.cpp:268-281 */
static int rt_add_chunk(unc_o_mapper_t *m, const float *chunk, uint32_t n) {
    if (!m->rt.chunk_processed || m->rt.finished || m->reset_req) return 0;
    if (m->rt.chunk_count >= m->max_chunks) { rt_set_failed(m); return 1; }
    m->rt.chunk_processed = 0;
    m->rt.chunk_count++;
    m->rt.raw_len += n;
    m->rt.chunk = chunk;
    m->rt.chunk_n = n;
    return 1;
}

/* Mapper::process_chunk 307-367 */
static uint32_t rt_process_chunk(unc_o_mapper_t *m) {
    if (m->rt.chunk_processed || m->reset_req) return 0;
    uint32_t nevents = 0;
    for (uint32_t i = 0; i < m->rt.chunk_n; ++i) {
        if (evdt_add_sample(&m->evdt, m->rt.chunk[i])) {
            if (!evprof_add_event(&m->rt.prof, m->evdt.event.mean)) continue;
            float evt_mean = m->rt.prof.next_mean;
            if (!rnorm_push(&m->rt.norm, evt_mean)) {
                uint32_t nskip = rnorm_skip_unread(&m->rt.norm, nevents);
                m->event_i += nskip;   /* skip_events, :256-259 */
                m->prev_size = 0;
                if (!rnorm_push(&m->rt.norm, evt_mean)) return nevents;
            }
            nevents++;
        }
    }
    m->rt.chunk_n = 0;
    m->rt.chunk_processed = 1;
    return nevents;
}

static int rt_chunk_mapped(const unc_o_mapper_t *m) { return m->rt.chunk_processed && m->rt.norm.is_empty; }

/* Mapper::map_chunk 381-431 with chunk_timeout = evt_timeout = FLT_MAX */
static int rt_map_chunk(unc_o_mapper_t *m) {
    if (m->reset_req || m->event_i >= m->P.max_events) {
        rt_set_failed(m);
        m->rt.ended = 1;
        return 1;
    } else if (m->rt.norm.is_empty && m->rt.chunk_processed && m->rt.chunk_count >= m->max_chunks) {
        rt_set_failed(m);
        return 1;
    }
    if (m->rt.norm.is_empty) return 0;
    uint32_t nevents = 5;   /* evt_batch_size, mapper.cpp:38,173-177 */
    if (m->event_i + nevents > m->P.max_events) nevents = m->P.max_events - m->event_i;
    for (uint32_t i = 0; i < nevents && !m->rt.norm.is_empty; ++i) {
        if (map_next(m)) {
            rnorm_skip_unread(&m->rt.norm, 0);
            m->rt.finished = 1;
            return 1;
        }
    }
    return 0;
}

void unc_o_rt_tap(const unc_o_mapper_t *m, unc_o_rt_tap_t *out, float *ring) {
    memset(out, 0, sizeof *out);
    out->det_t = m->evdt.t; out->det_total_events = m->evdt.total_events; out->det_len_sum = m->evdt.len_sum;
    if (!m->rt.inited) return;
    const rnorm_t *n = &m->rt.norm, *w = &m->rt.prof.window;
    out->norm_n = n->n; out->norm_wr = n->wr; out->norm_mean = n->mean; out->norm_varsum = n->varsum;
    memcpy(ring, n->signal, (size_t)NORM_LEN * sizeof(float));
    out->prof_n = w->n; out->prof_to_mask = m->rt.prof.to_mask; out->prof_queued = m->rt.prof.q_len;
    out->prof_mean = w->mean; out->prof_varsum = w->varsum;
    for (uint32_t i = 0; i < m->rt.prof.q_len && i < 28; ++i) out->prof_queue[i] = m->rt.prof.evq[(m->rt.prof.q_head + i) % (PROF_WIN + 1)];
}

void unc_o_set_max_chunks(unc_o_mapper_t *m, uint32_t max_chunks) { m->max_chunks = max_chunks; }
/* ReadBuffer::loc_.is_ended() of the read unc_o_chunk_read mapped last (Paf::ENDED, set by Mapper::map_chunk :386) */
int unc_o_rt_ended(const unc_o_mapper_t *m) { return m->rt.ended; }

/* One read on this mapper (= one channel), chunk by chunk.  chunks_used: chunks handed to the mapper. */
int unc_o_chunk_read(unc_o_mapper_t *m, const float *signal, uint32_t n, uint32_t chunk_len, unc_o_hit_t *out,
                     uint32_t *chunks_used) {
    m->rt_mode = 1;
    if (m->max_chunks == 0) m->max_chunks = 1000000;   /* ReadBuffer::PRMS.max_chunks */
    uint32_t used = 0, ci = 0;
    int done = 0;
    for (;; ++ci) {
        uint64_t st = (uint64_t)ci * chunk_len;
        if (st > n) st = n;
        uint32_t ln = (uint32_t)((st + chunk_len > n) ? n - st : chunk_len);   /* ReadBuffer::get_chunk, read_buffer.cpp:303-318 */
        if (ln == 0 && ci > 0) {
            /* RealtimePool::try_add_chunk on an empty chunk, realtime_pool.cpp:115-123 */
            if (rt_chunk_mapped(m) && !m->rt.finished) m->reset_req = 1;
        } else if (ci == 0) {
            rt_new_read(m, signal + st, ln);
            used++;
        } else {
            if (!rt_add_chunk(m, signal + st, ln)) break;   /* cannot happen in ordered mode */
            used++;
        }
        for (;;) {   /* RealtimePool::MapperThread::run, realtime_pool.cpp:349-358 */
            rt_process_chunk(m);
            if (rt_map_chunk(m)) { done = 1; break; }
            if (rt_chunk_mapped(m)) {
                /* the pool's thread comes round again before the next chunk arrives: a read whose chunks are used up, or that has
                 * reached max_events, ends here (map_chunk's first two exits), not on the next chunk */
                if (rt_map_chunk(m)) done = 1;
                break;
            }
        }
        if (done) break;
    }
    /* result: read_.loc_ */
    unc_o_hit_t *h = &m->hit;
    if (!m->state_success) {
        float bp_per_samp = m->P.bp_per_sec / m->P.sample_rate;
        h->rd_len = (uint64_t)((float)m->rt.raw_len * bp_per_samp);
    }
    h->n_events = m->evdt.total_events;
    h->event_i = m->event_i;
    h->mean_event_len = m->evdt.total_events ? evdt_mean_event_len(&m->evdt) : 0.0f;
    /* (as in unc_o_trace_finish: what this read leaves set is what the channel's next read starts with) */
    for (uint32_t k = 0; k < UNC_O_NKMER; ++k)
        if (m->sources_added[k]) { h->notes |= UNC_O_NOTE_FLAGS_LEFT; break; }
    minibwa_counters_t c1;
    minibwa_counters_get(&c1);
    h->n_nbr = c1.n_2occ - m->c0.n_2occ;
    h->n_sa = c1.n_sa - m->c0.n_sa;
    h->n_lf = c1.n_lf - m->c0.n_lf;
    *out = *h;
    if (chunks_used) *chunks_used = used;
    m->rt.active = 0;
    m->rt_mode = 0;
    return 0;
}

/* ------------------------------------------------------------------ threaded batch (cpu_baseline "port") */

typedef struct {
    const unc_o_index_t *ix;
    const unc_o_params_t *p;
    uint32_t n_reads;
    const float *signals;
    const uint64_t *offsets;
    unc_o_hit_t *out;
    uint32_t *next;
} batch_arg_t;

static void *batch_worker(void *vp) {
    batch_arg_t *a = (batch_arg_t *)vp;
    unc_o_mapper_t *m = unc_o_mapper_new(a->ix, a->p);
    for (;;) {
        uint32_t i = __atomic_fetch_add(a->next, 1, __ATOMIC_RELAXED);
        if (i >= a->n_reads) break;
        unc_o_map_read(m, a->signals + a->offsets[i], (uint32_t)(a->offsets[i + 1] - a->offsets[i]), &a->out[i]);
    }
    unc_o_mapper_free(m);
    return NULL;
}

double unc_o_map_batch(const unc_o_index_t *ix, const unc_o_params_t *p, int n_threads, uint32_t n_reads,
                       const float *signals, const uint64_t *offsets, unc_o_hit_t *out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    uint32_t next = 0;
    batch_arg_t a = { ix, p, n_reads, signals, offsets, out, &next };
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, batch_worker, &a);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
