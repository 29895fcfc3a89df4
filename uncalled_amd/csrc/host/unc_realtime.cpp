#include "unc_realtime.hpp"

#include <cstdlib>
#include <cstring>
#include <iostream>

namespace unc_host {

// ------------------------------------------------------------------ Chunk (chunk.cpp:16-125)
Chunk::Chunk(const std::string &id, uint16_t channel, uint32_t number, uint64_t chunk_start, const std::string &dtype,
             const std::string &raw_str)
    : id_(id), channel_idx_((uint16_t)(channel - 1)), number_(number), start_time_(chunk_start) {
    // float32 as MinKNOW sends it, or integers converted WITHOUT calibration (chunk.cpp:27-58: the calibrate() calls are
    // commented out in the reference)
    if (dtype == "float32") {
        raw_data_.resize(raw_str.size() / sizeof(float));
        memcpy(raw_data_.data(), raw_str.data(), raw_data_.size() * sizeof(float));
    } else if (dtype == "int16") {
        const size_t n = raw_str.size() / sizeof(int16_t);
        raw_data_.resize(n);
        for (size_t i = 0; i < n; ++i) { int16_t v; memcpy(&v, raw_str.data() + 2 * i, 2); raw_data_[i] = (float)v; }
    } else if (dtype == "int32") {
        const size_t n = raw_str.size() / sizeof(int32_t);
        raw_data_.resize(n);
        for (size_t i = 0; i < n; ++i) { int32_t v; memcpy(&v, raw_str.data() + 4 * i, 4); raw_data_[i] = (float)v; }
    } else {
        std::cerr << "Error: unsuportted raw signal dtype\n";
    }
}

Chunk::Chunk(const std::string &id, uint16_t channel, uint32_t number, uint64_t start_time, const std::vector<float> &raw_data,
             uint32_t raw_st, uint32_t raw_len)
    : id_(id), channel_idx_((uint16_t)(channel - 1)), number_(number), start_time_(start_time) {
    if (raw_st > raw_data.size()) raw_st = (uint32_t)raw_data.size();
    if ((size_t)raw_st + raw_len > raw_data.size()) raw_len = (uint32_t)(raw_data.size() - raw_st);
    raw_data_.assign(raw_data.begin() + raw_st, raw_data.begin() + raw_st + raw_len);
}

bool Chunk::pop(std::vector<float> &raw_data) {
    raw_data_.swap(raw_data);
    clear();
    return !raw_data.empty();
}

void Chunk::swap(Chunk &c) {
    std::swap(id_, c.id_);
    std::swap(channel_idx_, c.channel_idx_);
    std::swap(number_, c.number_);
    std::swap(start_time_, c.start_time_);
    raw_data_.swap(c.raw_data_);
}

void Chunk::print() const {
    for (float s : raw_data_) std::cout << s << std::endl;
}

// ------------------------------------------------------------------ RealtimePool
RealtimePool::RealtimePool(const Conf &conf) : conf_(conf), chans_(conf.num_channels) {
    unc_params_default(&prms_);
    prms_.max_events = conf.max_events;
    prms_.max_chunks = conf.max_chunks;
    prms_.chunk_time = conf.chunk_time;
    prms_.sample_rate = conf.sample_rate;
    if (unc_index_load(conf.bwa_prefix.c_str(), conf.idx_preset.c_str(), conf.device, &ix_) != UNC_OK ||
        unc_rt_create(ix_, &prms_, conf.num_channels, &rt_) != UNC_OK) {
        std::cerr << "Error: " << unc_last_error() << "\n";   // Mapper::load_static aborts on a bad index, mapper.cpp:118-127
        abort();
    }
}

RealtimePool::~RealtimePool() {
    if (rt_) unc_rt_free(rt_);
    if (ix_) unc_index_free(ix_);
}

void RealtimePool::start_read(Chan &c, Chunk &chunk) {   // Mapper::new_read(Chunk&), mapper.cpp:210-217
    c.id = chunk.get_id(); c.number = chunk.get_number(); c.start = chunk.get_start();
    c.raw_len = 0; c.chunks = 0;
    c.pending.clear();
    c.pending.swap(chunk);
    c.has_pending = true; c.pending_first = true; c.active = false;
}

// request_reset on the read the channel is mapping: what the next update() reports unmapped + ended
static void remember_old(RealtimePool::Chan &c) {
    c.given_up.push_back({c.id, c.number, c.start, c.raw_len});
}

// A chunk longer than chunk_time * sample_rate does not fit the per-channel staging of the device side: it is refused here
// (add_chunk / try_add_chunk return false, as for a channel that is still busy) instead of failing the whole round later.
// Refusals are counted (refused_chunks()).  A refused chunk of the read a channel is busy with ENDS that read -- it is reported
// unmapped + ended by the next update(), as after request_reset -- so that the channel takes the next read instead of waiting for
// a chunk that will never be accepted (round-3 advice).
bool RealtimePool::oversized(const Chunk &chunk) {
    const uint64_t cap = (uint64_t)(prms_.chunk_time * prms_.sample_rate);
    if (chunk.size() <= cap) return false;
    ++refused_chunks_;
    if (!warned_oversized_) {
        std::cerr << "Warning: chunk of " << chunk.size() << " samples on channel " << chunk.get_channel() << " is longer than chunk_time * sample_rate = "
                  << cap << " and was refused; its read is ended (further ones are counted: refused_chunks)\n";
        warned_oversized_ = true;
    }
    const uint16_t ch = chunk.get_channel_idx();
    if (ch < chans_.size()) {
        Chan &c = chans_[ch];
        if ((c.active || c.has_pending) && c.number == chunk.get_number()) {
            remember_old(c);
            c.active = false; c.has_pending = false; c.pending_first = false;
            c.pending.clear();
        }
    }
    return true;
}

// realtime_pool.cpp:74-110
bool RealtimePool::add_chunk(Chunk &chunk) {
    if (stopped_) return false;
    if (oversized(chunk)) return false;
    const uint16_t ch = chunk.get_channel_idx();
    if (ch >= chans_.size()) return false;
    Chan &c = chans_[ch];
    const bool busy = c.active || c.has_pending;
    if (busy && c.number != chunk.get_number()) {
        // the previous read is still aligning: it is reset (reported unmapped + ended by the next update), the chunk
        // of the new read waits in the channel's buffer; a chunk already waiting there is dropped (buffer_chunk)
        if (c.active) remember_old(c);
        start_read(c, chunk);
        return true;
    }
    if (!busy) { start_read(c, chunk); return true; }              // State::INACTIVE -> new_read
    if (c.has_pending) return false;                               // Mapper::add_chunk: the previous chunk is not processed yet
    c.pending.clear();
    c.pending.swap(chunk);
    c.has_pending = true; c.pending_first = false;
    return true;
}

// realtime_pool.cpp:112-142
bool RealtimePool::try_add_chunk(Chunk &chunk) {
    if (stopped_) return false;
    const uint16_t ch = chunk.get_channel_idx();
    if (ch >= chans_.size()) return false;
    Chan &c = chans_[ch];
    if (oversized(chunk)) return false;
    if (chunk.empty()) {
        // all chunks of the read were handed out: give up once the last one is mapped and the read is still undecided
        if (c.active && !c.has_pending) {
            remember_old(c);
            c.active = false;
        }
        return false;
    }
    if (!c.active && !c.has_pending) { start_read(c, chunk); return true; }
    if (c.number == chunk.get_number()) {
        if (c.has_pending) return false;                           // previous chunk still mapping
        c.pending.clear();
        c.pending.swap(chunk);
        c.has_pending = true; c.pending_first = false;
        return true;
    }
    return false;
}

void RealtimePool::end_read(uint16_t ch, uint32_t number) {
    if (ch >= chans_.size()) return;
    Chan &c = chans_[ch];
    if ((c.active || c.has_pending) && c.number == number) {
        if (c.active) remember_old(c);
        c.active = false; c.has_pending = false; c.pending.clear();
    }
}

Paf RealtimePool::unmapped_paf(const std::string &id, uint16_t ch_idx, uint64_t start, uint64_t raw_len) const {
    Paf p(id, (uint16_t)(ch_idx + 1), start);
    p.set_read_len((uint64_t)((float)raw_len * (prms_.bp_per_sec / prms_.sample_rate)));   // Paf::set_read_len, read_buffer.cpp:264-267
    return p;
}

std::vector<MapResult> RealtimePool::update() {
    std::vector<MapResult> ret;
    if (stopped_) return ret;
    // reads given up since the last call: request_reset -> set_failed + set_ended (mapper.cpp:384-390)
    for (size_t ch = 0; ch < chans_.size(); ++ch) {
        Chan &c = chans_[ch];
        for (const Chan::GivenUp &g : c.given_up) {
            Paf p = unmapped_paf(g.id, (uint16_t)ch, g.start, g.raw_len);
            p.set_ended();
            ret.emplace_back((uint16_t)(ch + 1), g.number, p);
        }
        c.given_up.clear();
    }
    // every buffered chunk (at most one per channel) in one call: mapped completely before it returns
    std::vector<unc_rt_chunk_t> chunks;
    std::vector<float> signal;
    std::vector<uint16_t> owner;
    for (size_t ch = 0; ch < chans_.size(); ++ch) {
        Chan &c = chans_[ch];
        if (!c.has_pending) continue;
        unc_rt_chunk_t d;
        memset(&d, 0, sizeof d);
        d.channel = (uint32_t)ch; d.read_number = c.number; d.flags = c.pending_first ? UNC_RT_FIRST : 0u;
        d.n_samples = c.pending.size(); d.offset = signal.size();
        d.calib.range = 1.f; d.calib.offset = 0.f; d.calib.digitisation = 1.f;
        signal.insert(signal.end(), c.pending.data().begin(), c.pending.data().end());
        chunks.push_back(d);
        owner.push_back((uint16_t)ch);
    }
    last_ms_ = 0;
    if (chunks.empty()) return ret;
    if (signal.empty()) signal.push_back(0.f);
    std::vector<unc_rt_result_t> res(chunks.size());
    const int rc = unc_rt_process_chunks_f32(rt_, (uint32_t)chunks.size(), chunks.data(), signal.data(), 0, nullptr, res.data());
    if (rc != UNC_OK && rc != UNC_ERR_OVERFLOW) {
        // the round was refused as a whole (an argument the device side does not accept): the reads it carried are reported
        // unmapped and ended, the pool stays usable
        std::cerr << "Error: " << unc_last_error() << " -- the " << chunks.size() << " reads of this round are reported unmapped\n";
        for (size_t i = 0; i < chunks.size(); ++i) {
            const uint16_t ch = owner[i];
            Chan &c = chans_[ch];
            c.raw_len += chunks[i].n_samples;
            Paf p = unmapped_paf(c.id, ch, c.start, c.raw_len);
            p.set_ended();
            ret.emplace_back((uint16_t)(ch + 1), c.number, p);
            c.active = false; c.has_pending = false; c.pending_first = false;
            c.pending.clear();
        }
        return ret;
    }
    float ms_e = 0, ms_m = 0;
    unc_rt_last_timing(rt_, &ms_e, &ms_m);
    last_ms_ = ms_e + ms_m;
    for (size_t i = 0; i < chunks.size(); ++i) {
        const uint16_t ch = owner[i];
        Chan &c = chans_[ch];
        c.raw_len += chunks[i].n_samples;
        c.chunks++;
        c.has_pending = false; c.pending_first = false;
        c.pending.clear();
        const unc_rt_result_t &r = res[i];
        if (r.state == UNC_RT_MAPPING) { c.active = true; continue; }
        c.active = false;
        const unc_hit_t &h = r.hit;
        Paf p(c.id, (uint16_t)(ch + 1), c.start);
        p.set_read_len(h.rd_len);
        if (r.state == UNC_RT_MAPPED && h.mapped)
            p.set_mapped(h.rd_st, h.rd_en, unc_index_seq_name(ix_, h.rid), h.rf_st, h.rf_en, h.rf_len, h.fwd != 0, (uint16_t)h.matches);
        if (r.ended) p.set_ended();
        p.set_float(Paf::MAP_TIME, last_ms_);     // kernels of this round (the reference times the read's map_chunk calls on its thread)
        if (h.status) std::cerr << "Warning: read " << c.id << ": device scratch overflow (status " << h.status << "), reported unmapped\n";
        ret.emplace_back((uint16_t)(ch + 1), c.number, p);
    }
    return ret;
}

bool RealtimePool::all_finished() {
    for (const Chan &c : chans_) if (c.active || c.has_pending || !c.given_up.empty()) return false;
    return true;
}

uint32_t RealtimePool::active_count() const {
    uint32_t n = 0;
    for (const Chan &c : chans_) n += (c.active || c.has_pending) ? 1u : 0u;
    return n;
}

void RealtimePool::stop_all() {
    stopped_ = true;
    for (Chan &c : chans_) { c.active = c.has_pending = false; c.given_up.clear(); c.pending.clear(); }
}

// ------------------------------------------------------------------ ClientSim-shaped chunk source over fast5 files
ClientSim::ClientSim(const Conf &conf)
    : conf_(conf), reader_(conf), chans_(conf.num_channels), chunk_len_((uint32_t)(conf.chunk_time * conf.sample_rate)) {}

void ClientSim::load_fast5s() {
    while (!reader_.empty()) {
        if (reader_.fill_buffer() == 0 && reader_.buffered() == 0) break;
        while (reader_.buffered()) {
            RawRead r = reader_.pop_read();
            if (r.channel_idx >= chans_.size()) continue;
            SimRead s;
            s.id = r.id; s.number = r.number; s.start = r.start_sample;
            s.signal.resize(r.signal.size());
            for (size_t i = 0; i < s.signal.size(); ++i) {       // ReadBuffer calibration, read_buffer.cpp:239-241 (u16 quirk)
                const float t1 = (float)(int)(uint16_t)r.signal[i] + r.calib.offset;
                const float t2 = r.calib.range * t1;
                s.signal[i] = t2 / r.calib.digitisation;
            }
            chans_[r.channel_idx].reads.push_back(std::move(s));
        }
    }
}

bool ClientSim::run() {
    running_ = true;
    return true;
}

std::vector<std::pair<uint16_t, Chunk>> ClientSim::get_read_chunks() {
    std::vector<std::pair<uint16_t, Chunk>> out;
    if (!running_) return out;
    for (size_t ch = 0; ch < chans_.size(); ++ch) {
        SimChan &c = chans_[ch];
        while (!c.reads.empty() && (uint64_t)c.chunk_i * chunk_len_ >= c.reads.front().signal.size()) {   // the read ran out
            c.reads.pop_front();
            c.chunk_i = 0;
        }
        if (c.reads.empty()) continue;
        const SimRead &r = c.reads.front();
        const uint32_t st = c.chunk_i * chunk_len_;
        out.emplace_back((uint16_t)(ch + 1), Chunk(r.id, (uint16_t)(ch + 1), r.number, r.start + st, r.signal, st, chunk_len_));
        c.chunk_i++;
    }
    rounds_++;
    return out;
}

void ClientSim::end_current(uint16_t channel, uint32_t number) {
    if (channel == 0 || channel > chans_.size()) return;
    SimChan &c = chans_[channel - 1];
    if (!c.reads.empty() && c.reads.front().number == number) { c.reads.pop_front(); c.chunk_i = 0; }
}

void ClientSim::stop_receiving_read(uint16_t channel, uint32_t number) { end_current(channel, number); }

uint32_t ClientSim::unblock_read(uint16_t channel, uint32_t number) {
    end_current(channel, number);
    return 0;      // the reference's simulator returns the ejection delay of its run replay; none is modelled here
}

bool ClientSim::is_running() {
    if (!running_) return false;
    for (const SimChan &c : chans_) if (!c.reads.empty()) return true;
    return false;
}

float ClientSim::get_runtime() const { return (float)rounds_ * conf_.chunk_time; }

}  // namespace unc_host
