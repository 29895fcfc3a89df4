// Host C++ surface that keeps the reference's pybind11 shape for the map path -- Conf, Paf, MapPool (and the
// Fast5Reader behind it) -- on top of the C ABI (include/uncalled_hip.h).  Mirrors, without sharing code with:
//   Conf        src/conf.hpp:57-96,296-340 (the properties `uncalled map` sets, uncalled/args.py:264-304)
//   Paf         src/read_buffer.hpp:42-126, read_buffer.cpp:34-160 (print_paf format, tags ch/st/mt)
//   Fast5Reader src/fast5_reader.cpp:62-248 (single- and multi-fast5, read-id filter, max_reads)
//   ReadBuffer  src/read_buffer.cpp:198-246 (attributes, max_chunks truncation; the int16 samples stay int16)
//   MapPool     src/map_pool.cpp:28-158 (add_fast5 / update / running / stop)
#pragma once
#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../../include/uncalled_hip.h"

namespace unc_host {

struct Conf {
    uint16_t threads = 1;            // no host threads are made of it (the GPU path batches); 1 = reads in `-t 1` order (UNC_ORDER_T1), > 1 = independent reads
    std::string bwa_prefix, idx_preset = "default", model_path;
    uint32_t max_events = 30000, seed_len = 22, max_chunks = 1000000, max_reads = 0, max_buffer = 100, num_channels = 512;
    float chunk_time = 1.0f, sample_rate = 4000.0f;
    std::string fast5_list, read_list;
    int device = 0;                  // GPU ordinal
    uint32_t batch_reads = 0;        // reads handed to one unc_map_batch call (0 = four times what the mapper keeps in flight)
    // realtime (conf.hpp:57-96 RealtimeParams): what the decision loop of scripts/uncalled:216-256 reads
    int realtime_mode = 0;           // RealtimePool::DEPLETE / ENRICH
    int active_chs = 0;              // RealtimePool::FULL / EVEN / ODD
    float duration = 0;              // hours (0 = until the source runs dry)
    uint32_t max_active_reads = 512;
};

class Paf {
public:
    enum Tag { MAP_TIME, WAIT_TIME, QUEUE_TIME, RECEIVE_TIME, CHANNEL, EJECT, READ_START, IN_SCAN, TOP_RATIO, MEAN_RATIO, ENDED, KEEP,
               DELAY, SEED_CLUSTER, CONFIDENT_EVENT };
    Paf() {}
    Paf(const std::string &rd_name, uint16_t channel, uint64_t start_sample);
    bool is_mapped() const { return is_mapped_; }
    bool is_ended() const { return ended_; }
    std::string str() const;       // one PAF line, no newline
    void print_paf() const;
    void set_read_len(uint64_t n) { rd_len_ = n; }
    void set_mapped(uint64_t rd_st, uint64_t rd_en, const std::string &rf_name, uint64_t rf_st, uint64_t rf_en, uint64_t rf_len, bool fwd,
                    uint16_t matches);
    void set_ended() { ended_ = true; }
    void set_int(Tag t, int v) { int_tags_.emplace_back(t, v); }
    void set_float(Tag t, float v) { float_tags_.emplace_back(t, v); }
    void set_str(Tag t, const std::string &v) { str_tags_.emplace_back(t, v); }
    const std::string &get_rd_name() const { return rd_name_; }

private:
    bool is_mapped_ = false, ended_ = false, fwd_ = false;
    std::string rd_name_, rf_name_;
    uint64_t rd_st_ = 0, rd_en_ = 0, rd_len_ = 0, rf_st_ = 0, rf_en_ = 0, rf_len_ = 0;
    uint16_t matches_ = 0;
    std::vector<std::pair<Tag, int>> int_tags_;
    std::vector<std::pair<Tag, float>> float_tags_;
    std::vector<std::pair<Tag, std::string>> str_tags_;
};

struct RawRead {   // ReadBuffer of the offline path, samples kept as stored
    std::string id;
    uint16_t channel_idx = 0;
    uint32_t number = 0;
    uint64_t start_sample = 0;
    unc_calib_t calib{1.f, 0.f, 1.f};
    std::vector<int16_t> signal;
};

class Fast5Reader {
public:
    Fast5Reader(const Conf &c);
    void add_fast5(const std::string &path) { fast5_list_.push_back(path); }
    bool add_read(const std::string &read_id);
    bool load_fast5_list(const std::string &fname);
    bool load_read_list(const std::string &fname);
    uint32_t fill_buffer();
    bool empty();
    RawRead pop_read();
    uint32_t buffered() const { return (uint32_t)buffered_.size(); }
    uint64_t front_size() const { return buffered_.empty() ? 0 : buffered_.front().signal.size(); }   // samples of the read pop_read hands out next
    bool all_buffered() const;
    // reads still to come when that is known -- no file left to open: the open file's unread reads + the buffered ones -- else UINT32_MAX
    uint32_t reads_left_if_known() const { return fast5_list_.empty() ? (uint32_t)(read_paths_.size() + buffered_.size()) : 0xFFFFFFFFu; }

private:
    bool open_next();
    bool read_one(const std::string &raw_path, const std::string &ch_path, RawRead &out);
    uint32_t max_reads_, max_buffer_, max_chunks_, chunk_len_, total_buffered_ = 0;
    std::deque<std::string> fast5_list_, read_paths_;
    std::unordered_set<std::string> read_filter_;
    std::deque<RawRead> buffered_;
    int64_t file_ = -1;   // hid_t
    bool multi_ = false;
};

// Writes reads in the two layouts Fast5Reader understands (multi: /read_<id>/{Raw,channel_id}; single: one read under
// /Raw/Reads/Read_<n> + /UniqueGlobalKey/channel_id).  Used by the simulators and the tests; the reference only reads.
bool write_fast5(const std::string &path, const std::vector<RawRead> &reads, bool multi, float sample_rate = 4000.0f);

// MapPool (map_pool.cpp:28-158) on one GPU.  The reference hands reads to worker threads and update() never blocks; here
// a loader thread reads fast5 files into page-locked staging buffers (two of them: batch k+1 is read and flattened while
// batch k is on the GPU), a mapper thread feeds them to unc_map_batch, and update() returns whatever has finished.  Both
// threads live until stop(): add_fast5 works at any time, running() turns true again when it does.
class MapPool {
public:
    explicit MapPool(const Conf &conf);
    ~MapPool();
    MapPool(const MapPool &) = delete;
    void add_fast5(const std::string &fname);
    std::vector<Paf> update();
    bool running();
    void stop();
    uint32_t batch_reads() const { return batch_reads_; }

private:
    struct ReadMeta { std::string id; uint16_t channel_idx; uint64_t start_sample; };
    struct Batch {
        int16_t *raw = nullptr;        // page-locked (pageable when the driver has none left)
        bool pinned = true;
        uint64_t cap = 0, used = 0;
        std::vector<uint64_t> off;
        std::vector<unc_calib_t> cal;
        std::vector<ReadMeta> meta;
    };
    void loader_main();
    void mapper_main();
    bool grow(Batch &b, uint64_t need);
    void release(Batch &b);
    static constexpr uint64_t kBatchBytes = 8ull << 30;     // staged samples of one batch (two batches exist)
    Conf conf_;
    Fast5Reader reader_;
    unc_index_t *ix_ = nullptr;
    unc_mapper_t *mapper_ = nullptr;
    uint32_t batch_reads_ = 0, first_batch_reads_ = 0;
    uint64_t batches_staged_ = 0;
    bool warned_pageable_ = false;
    RawRead carry_;                  // loader thread only: the read a closed batch had no room for; it opens the next batch
    bool carry_valid_ = false;
    float bp_per_samp_ = 0.0f;
    Batch bufs_[2];
    std::deque<int> free_, staged_;       // indices into bufs_
    std::deque<std::string> new_files_;   // add_fast5 -> loader thread
    std::vector<Paf> done_;
    std::mutex mtx_;
    std::condition_variable cv_;
    std::thread loader_, mapper_thread_;
    bool started_ = false, stopped_ = false, loader_idle_ = false, mapper_busy_ = false;
};

}  // namespace unc_host
