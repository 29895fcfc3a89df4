// TEST INFRASTRUCTURE -- see ref_harness.h.  This translation unit is the only new code
// in oracle/_ref/libunc_ref.so; everything it drives is the reference's own object code.
// It reaches Mapper's private state (prev_paths_, seed_tracker_, map_next) by re-labelling
// `private` for THIS TU only; the reference TUs are compiled untouched, and access
// specifiers do not change the Itanium-ABI layout, so both views of the classes agree.
#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <memory>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

#define private public
#define protected public
#include "mapper.hpp"
#undef private
#undef protected

#include "self_align_ref.hpp"
#include "ref_harness.h"

namespace {

void fill_read(ReadBuffer &r, const float *signal, uint32_t n, uint32_t number) {
    r.id_ = "read" + std::to_string(number);
    r.channel_idx_ = 0;
    r.number_ = number;
    r.start_sample_ = 0;
    r.full_signal_.assign(signal, signal + n);
    r.chunk_.clear();
    r.chunk_count_ = 0;
    r.chunk_processed_ = false;
    r.loc_ = Paf(r.id_, r.get_channel(), r.start_sample_);
    r.set_raw_len(n);
}

void fill_hit(Mapper *m, const Paf &p, ref_hit_t *h) {
    std::memset(h, 0, sizeof *h);
    h->mapped = p.is_mapped_;
    h->fwd = p.fwd_;
    h->rd_st = p.rd_st_; h->rd_en = p.rd_en_; h->rd_len = p.rd_len_;
    h->rf_st = p.rf_st_; h->rf_en = p.rf_en_; h->rf_len = p.rf_len_;
    h->matches = p.matches_;
    h->event_i = m->event_i_;
    h->n_events = m->norm_.n_;
    h->mean_event_len = m->evdt_.mean_event_len();
    std::strncpy(h->rf_name, p.rf_name_.c_str(), sizeof h->rf_name - 1);
}

}  // namespace

extern "C" {

int ref_init(const char *bwa_prefix, const char *idx_preset, uint32_t max_events) {
    Mapper::PRMS.bwa_prefix = bwa_prefix;
    Mapper::PRMS.idx_preset = idx_preset ? idx_preset : "default";
    if (max_events) Mapper::PRMS.max_events = max_events;
    Mapper m;  // load_static(): aborts on a bad index exactly as the reference does
    return Mapper::fmi.is_loaded() ? 0 : 1;
}

void ref_set_max_paths(uint32_t max_paths) { Mapper::PRMS.max_paths = max_paths; }
void ref_set_params(uint32_t min_rep_len, uint32_t max_rep_copy, uint32_t max_paths, uint32_t max_consec_stay, uint32_t max_events,
                    float max_stay_frac, float min_seed_prob, float threshold1, float threshold2, float peak_height,
                    float min_mean, float max_mean, uint32_t min_map_len, float min_mean_conf, float min_top_conf) {
    auto &P = Mapper::PRMS;
    P.min_rep_len = min_rep_len; P.max_rep_copy = max_rep_copy; P.max_paths = max_paths; P.max_consec_stay = max_consec_stay;
    P.max_events = max_events; P.max_stay_frac = max_stay_frac; P.min_seed_prob = min_seed_prob;
    P.event_prms.threshold1 = threshold1; P.event_prms.threshold2 = threshold2; P.event_prms.peak_height = peak_height;
    P.event_prms.min_mean = min_mean; P.event_prms.max_mean = max_mean;
    P.seed_prms.min_map_len = min_map_len; P.seed_prms.min_mean_conf = min_mean_conf; P.seed_prms.min_top_conf = min_top_conf;
}
void *ref_mapper_new(void) { return new Mapper(); }
void ref_mapper_free(void *m) { delete static_cast<Mapper *>(m); }

void ref_calibrate(const int16_t *raw, uint64_t n, float range, float offset, float digitisation, float *out) {
    // read_buffer.cpp:239-241: `for (u16 raw : int_data) cal_range * (raw + cal_offset) / cal_digit`
    for (uint64_t i = 0; i < n; ++i) {
        u16 r = (u16)raw[i];
        float calibrated = range * (r + offset) / digitisation;
        out[i] = calibrated;
    }
}

int ref_map_read(void *mp, const float *signal, uint32_t n, ref_hit_t *out) {
    Mapper *m = static_cast<Mapper *>(mp);
    ReadBuffer r;
    fill_read(r, signal, n, 0);
    minibwa_counters_reset();
    auto t0 = std::chrono::steady_clock::now();
    m->new_read(r);
    Paf p = m->map_read();
    auto t1 = std::chrono::steady_clock::now();
    fill_hit(m, p, out);
    minibwa_counters_t c;
    minibwa_counters_get(&c);
    out->n_nbr = c.n_2occ; out->n_sa = c.n_sa; out->n_lf = c.n_lf;
    out->map_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return 0;
}

double ref_map_batch(int n_threads, uint32_t n_reads, const float *signals, const uint64_t *offsets, ref_hit_t *out) {
    if (n_threads < 1) n_threads = 1;
    std::vector<Mapper *> mappers;
    for (int t = 0; t < n_threads; ++t) mappers.push_back(new Mapper());
    std::atomic<uint32_t> next(0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            for (;;) {
                uint32_t i = next.fetch_add(1);
                if (i >= n_reads) break;
                ref_map_read(mappers[t], signals + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), out + i);
            }
        });
    }
    for (auto &x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    for (auto *m : mappers) delete m;
    return std::chrono::duration<double>(t1 - t0).count();
}

// SURVEY 8(d) baseline "B2": the same reads through the as-shipped MapPool hand-shake (map_pool.cpp:45-69,130-158 driven by
// the polling loop of scripts/uncalled:127-167): every worker waits for its next read and for its result slot to be
// emptied in 10 ms sleeps, and the main thread visits the workers once per <= 10 ms round.  The fast5 reader is left out
// (signals are already in memory); the flags are the reference's unsynchronised bools, atomics here.
double ref_map_batch_pool(int n_threads, uint32_t n_reads, const float *signals, const uint64_t *offsets, ref_hit_t *out) {
    if (n_threads < 1) n_threads = 1;
    struct Worker {
        Mapper mapper;
        std::atomic<bool> running{true}, finished{false}, in_buffered{false}, out_buffered{false};
        uint32_t next = 0, done = 0;
        ref_hit_t hit;
        std::thread th;
    };
    std::vector<std::unique_ptr<Worker>> ws;
    for (int t = 0; t < n_threads; ++t) ws.emplace_back(new Worker());
    auto t0 = std::chrono::steady_clock::now();
    for (auto &wp : ws) {
        Worker *w = wp.get();
        w->th = std::thread([w, signals, offsets]() {
            while (!w->finished) {
                while (!w->in_buffered && !w->finished) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                if (w->finished) break;
                const uint32_t i = w->next;
                w->in_buffered = false;
                ref_hit_t h;
                ref_map_read(&w->mapper, signals + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), &h);
                while (w->out_buffered) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                w->hit = h; w->done = i;
                w->out_buffered = true;
            }
            while (w->out_buffered) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            w->running = false;
        });
    }
    uint32_t handed = 0;
    for (;;) {
        auto r0 = std::chrono::steady_clock::now();
        bool any = false;
        for (auto &wp : ws) {                       // MapPool::update
            Worker *w = wp.get();
            if (w->out_buffered) { out[w->done] = w->hit; w->out_buffered = false; }
            if (!w->in_buffered) {
                if (handed >= n_reads) w->finished = true;
                else { w->next = handed++; w->in_buffered = true; }
            }
            any = any || w->running;
        }
        if (!any) break;
        const auto dt = std::chrono::steady_clock::now() - r0;   // scripts/uncalled:156-160: sleep out the rest of 10 ms
        if (dt < std::chrono::milliseconds(10)) std::this_thread::sleep_for(std::chrono::milliseconds(10) - dt);
    }
    for (auto &wp : ws) wp->th.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static int g_last_ended = 0;
int ref_last_ended() { return g_last_ended; }      // Paf::is_ended() of the read ref_chunk_read mapped last

// One read through the Mapper's chunk API the way MapPoolOrd drives it (map_pool_ord.cpp:61-112 ->
// RealtimePool::try_add_chunk realtime_pool.cpp:112-142 -> MapperThread::run :349-358), single threaded.
int ref_chunk_read(void *mp, const float *signal, uint32_t n, uint32_t chunk_len, uint32_t number, ref_hit_t *out,
                   uint32_t *chunks_used) {
    Mapper *m = static_cast<Mapper *>(mp);
    Mapper::PRMS.chunk_timeout = FLT_MAX;   // Conf(Mode::MAP_ORD), conf.hpp:88-91
    Mapper::PRMS.evt_timeout = FLT_MAX;
    std::vector<float> sig(signal, signal + n);
    std::string id = "read" + std::to_string(number);
    minibwa_counters_reset();
    uint32_t used = 0;
    bool done = false;
    for (uint32_t ci = 0;; ++ci) {
        uint32_t st = ci * chunk_len, ln = chunk_len;        // ReadBuffer::get_chunk, read_buffer.cpp:303-318
        if (st > sig.size()) st = sig.size();
        if (st + ln > sig.size()) ln = sig.size() - st;
        Chunk chunk(id, 1, number, st, sig, st, ln);
        if (chunk.empty() && ci > 0) {
            if (m->chunk_mapped() && !m->finished()) m->request_reset();
        } else if (ci == 0) {
            m->new_read(chunk);
            used++;
        } else {
            if (!m->add_chunk(chunk)) break;
            used++;
        }
        for (;;) {
            m->process_chunk();
            if (m->map_chunk()) { done = true; break; }
            if (m->chunk_mapped()) {
                // the pool's thread comes round again (MapperThread::run is a loop over its mappers) before MapPoolOrd's next update
                // brings a chunk: a read whose chunks are used up, or that has reached max_events, is ended HERE, not on the next chunk
                if (m->map_chunk()) done = true;
                break;
            }
        }
        if (done) break;
    }
    g_last_ended = m->get_read().loc_.is_ended() ? 1 : 0;
    fill_hit(m, m->get_read().loc_, out);
    minibwa_counters_t c;
    minibwa_counters_get(&c);
    out->n_nbr = c.n_2occ; out->n_sa = c.n_sa; out->n_lf = c.n_lf;
    m->deactivate();
    if (chunks_used) *chunks_used = used;
    return 0;
}

void ref_set_max_chunks(uint32_t max_chunks) { ReadBuffer::PRMS.max_chunks = max_chunks; }

void ref_set_sort_mode(int mode) { unc_shim::sort_mode() = mode; }
/* the restated pattern-defeating quicksort on n pseudo-random keys drawn from [0, key_range) (many duplicates when the range is
 * small; ascending / descending / organ-pipe inputs for shape 1 / 2 / 3): 0 when the output is ascending and a permutation of
 * the input, else the first failing check */
int ref_sort_selftest(uint32_t n, uint32_t seed, uint32_t key_range, int shape) {
    struct E { uint32_t key, id; bool operator<(const E &o) const { return key < o.key; } };
    std::vector<E> v(n);
    uint64_t x = 0x9E3779B97F4A7C15ull * (seed + 1u);
    for (uint32_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint32_t k = (uint32_t)(x >> 33) % (key_range ? key_range : 1u);
        if (shape == 1) k = (uint32_t)((uint64_t)i * key_range / (n ? n : 1u));
        else if (shape == 2) k = (uint32_t)((uint64_t)(n - 1u - i) * key_range / (n ? n : 1u));
        else if (shape == 3) k = (uint32_t)((uint64_t)(i < n / 2 ? i : n - 1u - i) * key_range / (n ? n : 1u));
        v[i].key = k; v[i].id = i;
    }
    std::vector<E> in = v;
    unc_shim::pdq::sort(v.begin(), v.end(), std::less<E>());
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (i && v[i].key < v[i - 1].key) return 1;
        if (v[i].id >= n || seen[v[i].id] || in[v[i].id].key != v[i].key) return 2;
        seen[v[i].id] = 1;
    }
    return 0;
}
void ref_sort_stats(uint64_t *out3, int reset) {
    out3[0] = unc_shim::sorts().load(); out3[1] = unc_shim::tie_events().load(); out3[2] = unc_shim::tie_pairs().load();
    if (reset) { unc_shim::sorts().store(0); unc_shim::tie_events().store(0); unc_shim::tie_pairs().store(0); }
}

void ref_rt_tap(void *mp, ref_rt_tap_t *out, float *ring, uint32_t ring_cap) {
    Mapper *m = static_cast<Mapper *>(mp);
    std::memset(out, 0, sizeof *out);
    out->det_t = m->evdt_.t; out->det_total_events = m->evdt_.total_events_; out->det_len_sum = m->evdt_.len_sum_;
    const Normalizer &n = m->norm_, &w = m->evt_prof_.window_;
    out->norm_n = n.n_; out->norm_wr = n.wr_; out->norm_mean = n.mean_; out->norm_varsum = n.varsum_;
    for (uint32_t i = 0; i < ring_cap && i < n.signal_.size(); ++i) ring[i] = n.signal_[i];
    out->prof_n = w.n_; out->prof_to_mask = m->evt_prof_.to_mask_; out->prof_queued = (uint32_t)m->evt_prof_.events_.size();
    out->prof_mean = w.mean_; out->prof_varsum = w.varsum_;
    uint32_t i = 0;
    for (const Event &e : m->evt_prof_.events_) { if (i < 28) out->prof_queue[i] = e.mean; ++i; }
}

// self_align (self_align_ref.cpp:34-91), the FM walk behind `uncalled index`: flattened into CSR form
uint64_t ref_self_align(const char *bwa_prefix, uint32_t sample_dist, uint64_t *lens, uint64_t lens_cap, uint64_t *offsets,
                        uint64_t offsets_cap, uint64_t *n_paths) {
    std::vector<std::vector<u64>> r = self_align(bwa_prefix, sample_dist);
    uint64_t tot = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        if (i < offsets_cap) offsets[i] = tot;
        for (u64 v : r[i]) { if (tot < lens_cap) lens[tot] = v; ++tot; }
    }
    if (r.size() < offsets_cap) offsets[r.size()] = tot;
    *n_paths = r.size();
    return tot;
}

uint32_t ref_events(const float *signal, uint32_t n, ref_event_t *out, uint32_t cap, float *mean_event_len, uint32_t *total_events) {
    EventDetector ed(Mapper::PRMS.event_prms);
    std::vector<float> raw(signal, signal + n);
    std::vector<Event> ev = ed.get_events(raw);
    for (uint32_t i = 0; i < ev.size() && i < cap; ++i) {
        out[i].mean = ev[i].mean; out[i].stdv = ev[i].stdv;
        out[i].start = ev[i].start; out[i].length = ev[i].length;
    }
    if (mean_event_len) *mean_event_len = ed.mean_event_len();
    if (total_events) *total_events = ed.total_events_;
    return (uint32_t)ev.size();
}

void ref_norm_levels(const float *means, uint32_t m, float *levels, float *scale, float *shift) {
    Normalizer norm(Mapper::PRMS.norm_prms);
    norm.set_target(Mapper::model.get_means_mean(), Mapper::model.get_means_stdv());
    std::vector<float> sig(means, means + m);
    norm.set_signal(sig);
    if (scale) *scale = norm.get_scale();
    if (shift) *shift = norm.get_shift();
    for (uint32_t i = 0; i < m; ++i) levels[i] = norm.pop();
}

void ref_match_probs(float level, float *out1024) {
    for (u16 k = 0; k < 1024; ++k) out1024[k] = Mapper::model.match_prob(level, k);
}

void ref_model_tables(float *means, float *vars_x2, float *lognorm, float *model_mean, float *model_stdv) {
    for (u16 k = 0; k < 1024; ++k) {
        means[k] = Mapper::model.lv_means_[k];
        vars_x2[k] = Mapper::model.lv_vars_x2_[k];
        lognorm[k] = Mapper::model.lognorm_denoms_[k];
    }
    *model_mean = Mapper::model.get_means_mean();
    *model_stdv = Mapper::model.get_means_stdv();
}

void ref_kmer_ranges(uint64_t *out2048) {
    for (u16 k = 0; k < 1024; ++k) {
        Range r = Mapper::fmi.get_kmer_range(k);
        out2048[2 * k] = r.start_;
        out2048[2 * k + 1] = r.end_;
    }
}

void ref_thresholds(float *out64) {
    for (int i = 0; i < 64; ++i) out64[i] = Mapper::prob_threshes_[i];
}

void ref_get_neighbor(uint64_t s, uint64_t e, int base, uint64_t *os, uint64_t *oe) {
    Range r = Mapper::fmi.get_neighbor(Range(s, e), (u8)base);
    *os = r.start_; *oe = r.end_;
}

uint64_t ref_sa(uint64_t k) { return Mapper::fmi.sa(k); }
uint64_t ref_fm_size(void) { return Mapper::fmi.size(); }

void ref_trace_begin(void *mp, const float *signal, uint32_t n) {
    Mapper *m = static_cast<Mapper *>(mp);
    ReadBuffer r;
    fill_read(r, signal, n, 0);
    minibwa_counters_reset();
    m->new_read(r);
    // mapper.cpp:191-193
    m->map_timer_.reset();
    m->norm_.set_signal(m->evdt_.get_means(m->read_.full_signal_));
}

int ref_trace_step(void *mp) { return static_cast<Mapper *>(mp)->map_next() ? 1 : 0; }

uint32_t ref_trace_paths(void *mp, ref_path_t *out, uint32_t cap) {
    Mapper *m = static_cast<Mapper *>(mp);
    uint32_t n = m->prev_size_;
    for (uint32_t i = 0; i < n && i < cap; ++i) {
        const Mapper::PathBuffer &p = m->prev_paths_[i];
        ref_path_t &o = out[i];
        std::memset(&o, 0, sizeof o);
        o.fm_start = p.fm_range_.start_; o.fm_end = p.fm_range_.end_;
        o.event_moves = p.event_moves_; o.seed_prob = p.seed_prob_; o.kmer = p.kmer_;
        o.length = p.length_; o.consec_stays = p.consec_stays_; o.sa_checked = p.sa_checked_;
        for (int j = 0; j <= (int)p.length_ && j < 23; ++j) o.prob_sums[j] = p.prob_sums_[j];
    }
    return n;
}

uint32_t ref_trace_clusters(void *mp, ref_cluster_t *out, uint32_t cap, ref_cluster_t *max_map, float *len_sum, uint32_t *n_lens) {
    Mapper *m = static_cast<Mapper *>(mp);
    SeedTracker &st = m->seed_tracker_;
    uint32_t i = 0;
    for (const SeedCluster &c : st.seed_clusters_) {
        if (i < cap) {
            out[i].ref_st = c.ref_st_; out[i].ref_en_start = c.ref_en_.start_; out[i].ref_en_end = c.ref_en_.end_;
            out[i].evt_st = c.evt_st_; out[i].evt_en = c.evt_en_; out[i].total_len = c.total_len_; out[i].pad = 0;
        }
        ++i;
    }
    if (max_map) {
        const SeedCluster &c = st.max_map_;
        max_map->ref_st = c.total_len_ ? c.ref_st_ : 0;
        max_map->ref_en_start = c.ref_en_.start_; max_map->ref_en_end = c.ref_en_.end_;
        max_map->evt_st = c.evt_st_; max_map->evt_en = c.evt_en_; max_map->total_len = c.total_len_; max_map->pad = 0;
    }
    if (len_sum) *len_sum = st.len_sum_;
    if (n_lens) *n_lens = (uint32_t)st.all_lens_.size();
    return i;
}

uint32_t ref_trace_event_i(void *mp) { return static_cast<Mapper *>(mp)->event_i_; }

void ref_trace_finish(void *mp, ref_hit_t *out) {
    Mapper *m = static_cast<Mapper *>(mp);
    fill_hit(m, m->read_.loc_, out);
    minibwa_counters_t c;
    minibwa_counters_get(&c);
    out->n_nbr = c.n_2occ; out->n_sa = c.n_sa; out->n_lf = c.n_lf;
}

}  // extern "C"
