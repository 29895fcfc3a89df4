// Host C++ surface of the reference's realtime path, on top of the chunked C ABI (unc_rt_*, include/uncalled_hip.h):
//   Chunk        src/chunk.hpp:32-59, chunk.cpp:16-125 (always floats: MinKNOW's float32, or int16/int32 WITHOUT calibration)
//   RealtimePool src/realtime_pool.hpp:36-70, realtime_pool.cpp:60-261 (add_chunk / try_add_chunk / update / all_finished /
//                stop_all), with the deterministic semantics of MapPoolOrd (map_pool_ord.cpp:61-112; Conf(Mode::MAP_ORD)
//                disables the wall-clock timeouts, conf.hpp:88-91): update() maps every buffered chunk completely
//   ClientSim    the surface of src/client_sim.hpp:39-44 the decision loop uses (scripts/uncalled:216-256), fed from fast5
//                files: one read queue per channel, one chunk per channel per get_read_chunks().  The run-replay patterns
//                of the reference's simulator (gaps, delays, scan intervals) are out of scope.
#pragma once
#include <tuple>

#include "unc_pool.hpp"

namespace unc_host {

class Chunk {
public:
    Chunk() {}
    Chunk(const std::string &id, uint16_t channel, uint32_t number, uint64_t chunk_start, const std::string &dtype, const std::string &raw_str);
    Chunk(const std::string &id, uint16_t channel, uint32_t number, uint64_t start_time, const std::vector<float> &raw_data, uint32_t raw_st,
          uint32_t raw_len);
    bool pop(std::vector<float> &raw_data);      // hands the samples over, leaves the chunk empty
    void swap(Chunk &c);
    void clear() { raw_data_.clear(); }
    bool empty() const { return raw_data_.empty(); }
    uint32_t size() const { return (uint32_t)raw_data_.size(); }
    void print() const;
    uint64_t get_start() const { return start_time_; }
    uint64_t get_end() const { return start_time_ + raw_data_.size(); }
    std::string get_id() const { return id_; }
    uint16_t get_channel() const { return (uint16_t)(channel_idx_ + 1); }
    uint16_t get_channel_idx() const { return channel_idx_; }
    uint32_t get_number() const { return number_; }
    void set_start(uint64_t t) { start_time_ = t; }
    const std::vector<float> &data() const { return raw_data_; }

private:
    std::string id_;
    uint16_t channel_idx_ = 0;
    uint32_t number_ = 0;
    uint64_t start_time_ = 0;
    std::vector<float> raw_data_;
};

using MapResult = std::tuple<uint16_t, uint32_t, Paf>;   // channel (1-based), read number, record

class RealtimePool {
public:
    enum Mode { DEPLETE, ENRICH };
    enum ActiveChs { FULL, EVEN, ODD };
    explicit RealtimePool(const Conf &conf);
    ~RealtimePool();
    RealtimePool(const RealtimePool &) = delete;
    bool add_chunk(Chunk &chunk);
    bool try_add_chunk(Chunk &chunk);
    void end_read(uint16_t ch_idx, uint32_t number);
    std::vector<MapResult> update();
    bool all_finished();
    void stop_all();
    bool is_stopped() const { return stopped_; }
    uint32_t active_count() const;
    float last_round_ms() const { return last_ms_; }
    uint64_t refused_chunks() const { return refused_chunks_; }     // chunks longer than chunk_time * sample_rate, never staged

    struct Chan {
        bool active = false;         // a read is being mapped (Mapper state MAPPING; its last chunk is fully mapped)
        bool has_pending = false;    // a chunk waits for the next update()
        bool pending_first = false;  // ... and it starts a read (Mapper::new_read(Chunk&))
        uint32_t number = 0, chunks = 0;
        uint64_t start = 0, raw_len = 0;
        std::string id;
        Chunk pending;
        // request_reset: reads reported unmapped + ended by the next update().  A list, not one record: between two updates a channel
        // can give up its mapping read (a new read takes it over) AND that new read's first chunk can be refused as oversized before it
        // was ever mapped -- the reference reports every reset read (round-4 advice: one record lost the first of the two)
        struct GivenUp { std::string id; uint32_t number; uint64_t start, raw_len; };
        std::vector<GivenUp> given_up;
    };
private:
    void start_read(Chan &c, Chunk &chunk);
    Paf unmapped_paf(const std::string &id, uint16_t ch_idx, uint64_t start, uint64_t raw_len) const;
    Conf conf_;
    unc_params_t prms_;
    unc_index_t *ix_ = nullptr;
    unc_rt_t *rt_ = nullptr;
    std::vector<Chan> chans_;
    bool stopped_ = false;
    bool warned_oversized_ = false;
    uint64_t refused_chunks_ = 0;
    bool oversized(const Chunk &chunk);
    float last_ms_ = 0;
};

class ClientSim {
public:
    explicit ClientSim(const Conf &conf);
    void add_fast5(const std::string &fname) { reader_.add_fast5(fname); }
    void load_fast5s();
    bool run();
    std::vector<std::pair<uint16_t, Chunk>> get_read_chunks();
    void stop_receiving_read(uint16_t channel, uint32_t number);
    uint32_t unblock_read(uint16_t channel, uint32_t number);
    bool is_running();
    float get_runtime() const;          // simulated seconds = chunk rounds x chunk_time

private:
    struct SimRead { std::string id; uint32_t number; uint64_t start; std::vector<float> signal; };
    struct SimChan { std::deque<SimRead> reads; uint32_t chunk_i = 0; };
    void end_current(uint16_t channel, uint32_t number);
    Conf conf_;
    Fast5Reader reader_;
    std::vector<SimChan> chans_;
    uint32_t chunk_len_, rounds_ = 0;
    bool running_ = false;
};

}  // namespace unc_host
