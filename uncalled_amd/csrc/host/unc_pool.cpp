#include "unc_pool.hpp"

#include <hdf5.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace unc_host {

// ------------------------------------------------------------------ Paf (read_buffer.cpp:34-160)
static const char *PAF_TAGS[] = {"mt", "wt", "qt", "rt", "ch", "ej", "st", "mx", "tr", "mr", "en", "kp", "dl", "sc", "ce"};

Paf::Paf(const std::string &rd_name, uint16_t channel, uint64_t start_sample) : rd_name_(rd_name) {
    set_int(CHANNEL, channel);
    set_int(READ_START, (int)start_sample);
}

void Paf::set_mapped(uint64_t rd_st, uint64_t rd_en, const std::string &rf_name, uint64_t rf_st, uint64_t rf_en, uint64_t rf_len, bool fwd,
                     uint16_t matches) {
    is_mapped_ = true;
    rd_st_ = rd_st; rd_en_ = rd_en; rf_name_ = rf_name; rf_st_ = rf_st; rf_en_ = rf_en; rf_len_ = rf_len; fwd_ = fwd; matches_ = matches;
}

std::string Paf::str() const {
    std::ostringstream o;
    o << rd_name_ << "\t" << rd_len_ << "\t";
    if (is_mapped_) {
        o << rd_st_ << "\t" << rd_en_ << "\t" << (fwd_ ? '+' : '-') << "\t" << rf_name_ << "\t" << rf_len_ << "\t" << rf_st_ << "\t" << rf_en_
          << "\t" << matches_ << "\t" << (rf_en_ - rf_st_ + 1) << "\t" << 255;
    } else {
        o << "*\t*\t*\t*\t*\t*\t*\t*\t*\t255";
    }
    o << std::fixed;
    for (auto &t : int_tags_) o << "\t" << PAF_TAGS[t.first] << ":i:" << t.second;
    for (auto &t : float_tags_) o << "\t" << PAF_TAGS[t.first] << ":f:" << t.second;
    for (auto &t : str_tags_) o << "\t" << PAF_TAGS[t.first] << ":Z:" << t.second;
    return o.str();
}

void Paf::print_paf() const {
    std::cout << str() << "\n";
    std::cout.flush();
}

// ------------------------------------------------------------------ Fast5Reader (fast5_reader.cpp:62-248) over the HDF5 C API
Fast5Reader::Fast5Reader(const Conf &c)
    : max_reads_(c.max_reads), max_buffer_(c.max_buffer ? c.max_buffer : 100), max_chunks_(c.max_chunks),
      chunk_len_((uint16_t)(c.chunk_time * c.sample_rate)) {
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);   // errors are reported by return values
    if (!c.read_list.empty()) load_read_list(c.read_list);
    if (!c.fast5_list.empty()) load_fast5_list(c.fast5_list);
}

bool Fast5Reader::add_read(const std::string &read_id) {
    if (max_reads_ != 0 && read_filter_.size() >= max_reads_) return false;
    read_filter_.insert(read_id);
    return true;
}

bool Fast5Reader::load_fast5_list(const std::string &fname) {
    std::ifstream f(fname);
    if (!f.is_open()) { std::cerr << "Error: failed to open fast5 list \"" << fname << "\".\n"; return false; }
    std::string line;
    while (std::getline(f, line)) if (!line.empty()) add_fast5(line);
    return true;
}

bool Fast5Reader::load_read_list(const std::string &fname) {
    std::ifstream f(fname);
    if (!f.is_open()) { std::cerr << "Error: failed to open read list \"" << fname << "\".\n"; return false; }
    std::string line;
    while (std::getline(f, line)) if (!add_read(line)) break;
    return true;
}

bool Fast5Reader::all_buffered() const {
    return (max_reads_ > 0 && total_buffered_ >= max_reads_) || (!read_filter_.empty() && total_buffered_ >= read_filter_.size());
}

bool Fast5Reader::empty() { return buffered_.empty() && read_paths_.empty() && (fast5_list_.empty() || all_buffered()); }

static herr_t collect_names(hid_t, const char *name, const H5L_info_t *, void *op) {
    static_cast<std::vector<std::string> *>(op)->push_back(name);
    return 0;
}
static std::vector<std::string> list_group(hid_t f, const std::string &path) {
    std::vector<std::string> names;
    hid_t g = H5Gopen2(f, path.c_str(), H5P_DEFAULT);
    if (g < 0) return names;
    H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, nullptr, collect_names, &names);
    H5Gclose(g);
    return names;
}
static bool attr_string(hid_t f, const std::string &obj, const char *name, std::string &out) {
    hid_t a = H5Aopen_by_name(f, obj.c_str(), name, H5P_DEFAULT, H5P_DEFAULT);
    if (a < 0) return false;
    hid_t t = H5Aget_type(a);
    bool ok = false;
    if (H5Tget_class(t) == H5T_STRING) {
        if (H5Tis_variable_str(t)) {
            char *p = nullptr;
            hid_t mt = H5Tcopy(H5T_C_S1);
            H5Tset_size(mt, H5T_VARIABLE);
            if (H5Aread(a, mt, &p) >= 0 && p) { out = p; free(p); ok = true; }
            H5Tclose(mt);
        } else {
            size_t n = H5Tget_size(t);
            std::vector<char> buf(n + 1, 0);
            hid_t mt = H5Tcopy(H5T_C_S1);
            H5Tset_size(mt, n);
            if (H5Aread(a, mt, buf.data()) >= 0) { out = std::string(buf.data(), strnlen(buf.data(), n)); ok = true; }
            H5Tclose(mt);
        }
    }
    H5Tclose(t);
    H5Aclose(a);
    return ok;
}
static bool attr_double(hid_t f, const std::string &obj, const char *name, double &out) {
    hid_t a = H5Aopen_by_name(f, obj.c_str(), name, H5P_DEFAULT, H5P_DEFAULT);
    if (a < 0) return false;
    hid_t t = H5Aget_type(a);
    bool ok = false;
    if (H5Tget_class(t) == H5T_STRING) {
        H5Tclose(t); H5Aclose(a);
        std::string s;
        if (!attr_string(f, obj, name, s)) return false;
        out = atof(s.c_str());
        return true;
    }
    ok = H5Aread(a, H5T_NATIVE_DOUBLE, &out) >= 0;
    if (ok && H5Tget_class(t) == H5T_FLOAT) {
        // ReadBuffer takes every attribute through get_attr_map's strings and atof (read_buffer.cpp:202-222); the fast5
        // library renders a floating attribute with a default-precision stream, i.e. 6 significant digits
        // (range 1534.141357421875 -> "1534.14").  Keep that rounding so the calibrated samples agree.
        std::ostringstream oss;
        oss << out;
        out = atof(oss.str().c_str());
    }
    H5Tclose(t);
    H5Aclose(a);
    return ok;
}

bool Fast5Reader::open_next() {
    read_paths_.clear();
    if (file_ >= 0) { H5Fclose((hid_t)file_); file_ = -1; }
    if (fast5_list_.empty()) return false;
    const std::string fn = fast5_list_.front();
    fast5_list_.pop_front();
    file_ = (int64_t)H5Fopen(fn.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (file_ < 0) { std::cerr << "Error: failed to open fast5 \"" << fn << "\"\n"; return true; }
    const hid_t f = (hid_t)file_;
    multi_ = true;
    for (const std::string &s : list_group(f, "/")) if (s == "Raw") { multi_ = false; break; }
    if (!multi_) {
        for (const std::string &read : list_group(f, "/Raw/Reads")) {
            std::string id;
            if (!attr_string(f, "/Raw/Reads/" + read, "read_id", id) || id.empty()) { std::cerr << "Error: failed to find read_id\n"; return false; }
            if (read_filter_.empty() || read_filter_.count(id)) read_paths_.push_back("/" + read);
        }
    } else {
        for (const std::string &read : list_group(f, "/")) {
            const std::string id = read.substr(read.find('_') + 1);
            if (read_filter_.empty() || read_filter_.count(id)) read_paths_.push_back("/" + read);
        }
    }
    return true;
}

// ReadBuffer(file, raw_path, ch_path), read_buffer.cpp:198-246
bool Fast5Reader::read_one(const std::string &raw_path, const std::string &ch_path, RawRead &r) {
    const hid_t f = (hid_t)file_;
    double v;
    std::string s;
    if (attr_string(f, raw_path, "read_id", s)) r.id = s;
    if (attr_double(f, raw_path, "read_number", v)) r.number = (uint32_t)(int)v;
    if (attr_double(f, raw_path, "start_time", v)) r.start_sample = (uint64_t)(int)v;
    float dig = 1, rng = 1, off = 0;
    if (attr_string(f, ch_path, "channel_number", s)) r.channel_idx = (uint16_t)(atoi(s.c_str()) - 1);
    else if (attr_double(f, ch_path, "channel_number", v)) r.channel_idx = (uint16_t)((int)v - 1);
    if (attr_double(f, ch_path, "digitisation", v)) dig = (float)v;
    if (attr_double(f, ch_path, "range", v)) rng = (float)v;
    if (attr_double(f, ch_path, "offset", v)) off = (float)v;
    r.calib.range = rng; r.calib.offset = off; r.calib.digitisation = dig;
    hid_t d = H5Dopen2(f, (raw_path + "/Signal").c_str(), H5P_DEFAULT);
    if (d < 0) return false;
    hid_t sp = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    r.signal.resize(n > 0 ? (size_t)n : 0);
    bool ok = n <= 0 || H5Dread(d, H5T_NATIVE_INT16, H5S_ALL, H5S_ALL, H5P_DEFAULT, r.signal.data()) >= 0;
    H5Sclose(sp);
    H5Dclose(d);
    // chunk_count_ > max_chunks: the signal is cut to whole chunks (read_buffer.cpp:229-234)
    if (chunk_len_) {
        const uint64_t cc = r.signal.size() / chunk_len_ + (r.signal.size() % chunk_len_ != 0);
        if (cc > max_chunks_) r.signal.resize((size_t)max_chunks_ * chunk_len_);
    }
    return ok;
}

uint32_t Fast5Reader::fill_buffer() {
    uint32_t count = 0;
    while (buffered_.size() < max_buffer_) {
        if (all_buffered()) { read_paths_.clear(); fast5_list_.clear(); break; }
        while (read_paths_.empty()) if (!open_next()) break;
        if (read_paths_.empty()) break;
        std::string raw_path, ch_path;
        if (!multi_) { raw_path = "/Raw/Reads" + read_paths_.front(); ch_path = "/UniqueGlobalKey/channel_id"; }
        else { raw_path = read_paths_.front() + "/Raw"; ch_path = read_paths_.front() + "/channel_id"; }
        read_paths_.pop_front();
        RawRead r;
        if (read_one(raw_path, ch_path, r)) {
            buffered_.push_back(std::move(r));
            count++;
            total_buffered_++;
        }
    }
    return count;
}

RawRead Fast5Reader::pop_read() {
    if (buffered_.empty()) fill_buffer();             // fast5_reader.cpp:236-238
    if (buffered_.empty()) return RawRead();          // exhausted: an empty read (id "", no samples), never UB
    RawRead r = std::move(buffered_.front());
    buffered_.pop_front();
    return r;
}

// ------------------------------------------------------------------ fast5 writer (simulators / tests)
static void put_attr_str(hid_t obj, const char *name, const std::string &v) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, v.size() ? v.size() : 1);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, v.c_str());
    H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}
template <class T> static void put_attr_num(hid_t obj, const char *name, hid_t file_t, hid_t mem_t, T v) {
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, file_t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, mem_t, &v);
    H5Aclose(a); H5Sclose(sp);
}
static void write_read(hid_t raw_g, hid_t ch_g, const RawRead &r, float sample_rate) {
    put_attr_str(raw_g, "read_id", r.id);
    put_attr_num<int32_t>(raw_g, "read_number", H5T_STD_I32LE, H5T_NATIVE_INT32, (int32_t)r.number);
    put_attr_num<uint64_t>(raw_g, "start_time", H5T_STD_U64LE, H5T_NATIVE_UINT64, r.start_sample);
    put_attr_num<uint32_t>(raw_g, "duration", H5T_STD_U32LE, H5T_NATIVE_UINT32, (uint32_t)r.signal.size());
    hsize_t n = r.signal.size();
    hid_t sp = H5Screate_simple(1, &n, nullptr);
    hid_t d = H5Dcreate2(raw_g, "Signal", H5T_STD_I16LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (n) H5Dwrite(d, H5T_NATIVE_INT16, H5S_ALL, H5S_ALL, H5P_DEFAULT, r.signal.data());
    H5Dclose(d); H5Sclose(sp);
    put_attr_str(ch_g, "channel_number", std::to_string(r.channel_idx + 1));
    put_attr_num<double>(ch_g, "digitisation", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, r.calib.digitisation);
    put_attr_num<double>(ch_g, "range", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, r.calib.range);
    put_attr_num<double>(ch_g, "offset", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, r.calib.offset);
    put_attr_num<double>(ch_g, "sampling_rate", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, sample_rate);
}

bool write_fast5(const std::string &path, const std::vector<RawRead> &reads, bool multi, float sample_rate) {
    if (!multi && reads.size() != 1) return false;
    hid_t f = H5Fcreate(path.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return false;
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);
    for (const RawRead &r : reads) {
        const std::string raw = multi ? "/read_" + r.id + "/Raw" : "/Raw/Reads/Read_" + std::to_string(r.number);
        const std::string ch = multi ? "/read_" + r.id + "/channel_id" : "/UniqueGlobalKey/channel_id";
        hid_t rg = H5Gcreate2(f, raw.c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT);
        hid_t cg = H5Gcreate2(f, ch.c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT);
        write_read(rg, cg, r, sample_rate);
        H5Gclose(rg); H5Gclose(cg);
    }
    H5Pclose(lcpl);
    H5Fclose(f);
    return true;
}

// ------------------------------------------------------------------ MapPool (map_pool.cpp:28-158), batched on one GPU
MapPool::MapPool(const Conf &conf) : conf_(conf), reader_(conf) {
    unc_params_t p;
    unc_params_default(&p);
    p.max_events = conf.max_events;
    p.max_chunks = conf.max_chunks;
    p.chunk_time = conf.chunk_time;
    p.sample_rate = conf.sample_rate;
    bp_per_samp_ = p.bp_per_sec / p.sample_rate;
    if (unc_index_load(conf.bwa_prefix.c_str(), conf.idx_preset.c_str(), conf.device, &ix_) != UNC_OK ||
        unc_mapper_create(ix_, &p, nullptr, &mapper_) != UNC_OK) {
        std::cerr << "Error: " << unc_last_error() << "\n";   // Mapper::load_static aborts on a bad index, mapper.cpp:118-127
        abort();
    }
    // -t 1 (the reference's default): ONE Mapper maps the reads in input order, and what a read leaves set in sources_added_ is seen
    // by the next (mapper.cpp:88,612-623).  The C ABI reproduces that order exactly (UNC_ORDER_T1: the few reads whose predecessor
    // left flags set are mapped a second time); -t N > 1, where the reference's own outcome depends on which thread gets which
    // read, maps every read independently.
    if (conf.threads <= 1 && unc_mapper_set_read_order(mapper_, UNC_ORDER_T1) != UNC_OK) {
        std::cerr << "Error: " << unc_last_error() << "\n";
        abort();
    }
    // a batch = four times as many reads as the mapper keeps in flight: every unc_map_batch call ends with the few reads that
    // run to max_events on a nearly idle chip, and at one load of the slots that tail is a third of the call (round 3: 11.5 k
    // reads/s end to end against 17 k in HBM); the FIRST batch is one load of the slots, so that the first PAF lines do not
    // wait for 65 k reads to be read from disk.  --batch-reads overrides; kBatchBytes caps the page-locked memory of a batch.
    uint32_t geo[5] = {0, 0, 0, 0, 0};
    unc_mapper_geometry(mapper_, geo);
    first_batch_reads_ = geo[1] ? geo[1] : 4096;
    batch_reads_ = conf.batch_reads ? conf.batch_reads : 4 * first_batch_reads_;
    if (first_batch_reads_ > batch_reads_) first_batch_reads_ = batch_reads_;
    free_.push_back(0);
    free_.push_back(1);
}

MapPool::~MapPool() {
    stop();
    for (Batch &b : bufs_) release(b);
    if (mapper_) unc_mapper_free(mapper_);
    if (ix_) unc_index_free(ix_);
}

void MapPool::add_fast5(const std::string &fname) {
    std::lock_guard<std::mutex> lk(mtx_);
    new_files_.push_back(fname);
    cv_.notify_all();       // an idle loader picks it up
}

void MapPool::release(Batch &b) {
    if (b.raw) { if (b.pinned) unc_host_free(b.raw); else free(b.raw); }
    b.raw = nullptr; b.cap = 0;
}

// Room for `need` samples.  Page-locked memory while the driver hands it out; pageable memory (the C ABI takes either, the
// copy to the device is then staged by the runtime) with one warning when it does not -- a full pinned pool is no reason to
// end a run.  The old buffer is released before the new one is the only copy only in the sense that both exist during memcpy.
bool MapPool::grow(Batch &b, uint64_t need) {
    if (need <= b.cap) return true;
    uint64_t cap = b.cap ? 2 * b.cap : (32ull << 20);            // at least double, at least what is asked for (in 32 M-sample steps)
    if (cap < need) cap = (need + (32ull << 20) - 1) / (32ull << 20) * (32ull << 20);
    if (cap > kBatchBytes / 2 && need <= kBatchBytes / 2) cap = kBatchBytes / 2;      // never past the byte cap of a batch
    // UNC_STAGING_MAX_KB: a ceiling on one staging buffer for hosts short of (lockable) memory -- and how the tests reach the
    // out-of-memory paths of the loader
    const char *lim_env = getenv("UNC_STAGING_MAX_KB");
    const uint64_t limit = lim_env ? (uint64_t)atoll(lim_env) * 512ull : 0ull;       // in samples
    if (limit) {
        if (need > limit) return false;
        if (cap > limit) cap = limit;
    }
    bool pinned = true;
    int16_t *p = static_cast<int16_t *>(unc_host_alloc(cap * 2));
    if (!p) {
        pinned = false;
        p = static_cast<int16_t *>(malloc(cap * 2));
        if (!p) return false;
        if (!warned_pageable_) {
            std::cerr << "Warning: no page-locked host memory for a staging buffer of " << (cap * 2 >> 20) << " MB: using pageable memory (slower copies)\n";
            warned_pageable_ = true;
        }
    }
    if (b.used) memcpy(p, b.raw, b.used * 2);
    release(b);
    b.raw = p; b.cap = cap; b.pinned = pinned;
    return true;
}

// fast5 files -> staged batches (flattened int16 samples in page-locked memory + per-read offsets / calibration).  Like the
// reference's worker threads (map_pool.cpp:31-42) the two threads live until stop(): files may be added at any time.
void MapPool::loader_main() {
    for (;;) {
        int bi;
        {
            std::unique_lock<std::mutex> lk(mtx_);
            cv_.wait(lk, [&] { return stopped_ || !free_.empty(); });
            if (stopped_) break;
            bi = free_.front();
            free_.pop_front();
            while (!new_files_.empty()) { reader_.add_fast5(new_files_.front()); new_files_.pop_front(); }
        }
        Batch &b = bufs_[bi];
        b.used = 0; b.off.assign(1, 0); b.cal.clear(); b.meta.clear();
        const uint32_t target = batches_staged_ == 0 ? first_batch_reads_ : batch_reads_;
        std::vector<Paf> unstaged;      // reads no staging buffer could be had for: reported unmapped, like every read the reference is given
        while (b.meta.size() < target) {
            // the GPU has nothing to do and one load of its slots is staged: hand that over now.  Batches thus grow from one
            // load towards four as long as reading keeps ahead of mapping, and the GPU never waits for a full batch.
            if (b.meta.size() >= first_batch_reads_ && (b.meta.size() & 255u) == 0) {
                std::lock_guard<std::mutex> lk(mtx_);
                if (!mapper_busy_ && staged_.empty()) break;
            }
            if (!carry_valid_ && reader_.buffered() == 0 && (reader_.empty() || reader_.fill_buffer() == 0)) break;
            // (a batch is also closed by bytes: whole reads with -c 1000000 are tens of MB each)
            const uint64_t next_size = carry_valid_ ? carry_.signal.size() : reader_.front_size();
            if (!b.meta.empty() && (b.used + next_size + 1) * 2 > kBatchBytes) break;
            RawRead r;
            if (carry_valid_) { r = std::move(carry_); carry_valid_ = false; }      // the read the last batch had no room for
            else r = reader_.pop_read();
            // a buffer that must grow goes straight to what a FULL batch will need, also while the short first batch is being
            // read: page-locking is slow (seconds for the 4-5 GB of a batch) and holds up the HIP calls of the mapping thread
            // meanwhile, so both buffers reach their final size during the first two batches and never grow again
            uint64_t need = b.used + r.signal.size() + 1;
            if (need > b.cap && b.meta.size() >= 16) {
                // (... but never for more reads than are left: a small input must not page-lock gigabytes it will never fill -- round-3 advice)
                const uint32_t left = reader_.reads_left_if_known();
                const uint64_t full = left == 0xFFFFFFFFu ? batch_reads_ : std::min<uint64_t>(batch_reads_, b.meta.size() + 1u + left);
                const uint64_t est = (uint64_t)((double)b.used / (double)b.meta.size() * (double)full * 1.1);
                if (est > need) need = est < kBatchBytes / 2 ? est : kBatchBytes / 2;
                if (need < b.used + r.signal.size() + 1) need = b.used + r.signal.size() + 1;
            }
            if (!grow(b, need) && !grow(b, b.used + r.signal.size() + 1)) {
                // no memory for the buffer to grow.  A batch that holds reads is handed over as it is and this read opens the next
                // one (whose buffer is then empty); a read that does not even fit an empty buffer gets its unmapped PAF line
                // (every read gets a line, map_pool.cpp:130-158) and the loader goes on with the next read
                if (!b.meta.empty()) {
                    carry_ = std::move(r); carry_valid_ = true;
                    break;
                }
                std::cerr << "Error: out of host memory for a staging buffer of " << ((r.signal.size() + 1) * 2 >> 20) << " MB; read " << r.id
                          << " is reported unmapped\n";
                Paf pf(r.id, (uint16_t)(r.channel_idx + 1), r.start_sample);
                pf.set_read_len((uint64_t)((float)r.signal.size() * bp_per_samp_));     // (an unmapped read's length: read_buffer.cpp:133-155)
                unstaged.push_back(std::move(pf));
                continue;
            }
            if (!r.signal.empty()) memcpy(b.raw + b.used, r.signal.data(), r.signal.size() * 2);
            b.used += r.signal.size();
            b.off.push_back(b.used);
            b.cal.push_back(r.calib);
            b.meta.push_back(ReadMeta{r.id, r.channel_idx, r.start_sample});
        }
        std::unique_lock<std::mutex> lk(mtx_);
        for (Paf &pf : unstaged) done_.push_back(std::move(pf));
        if (b.meta.empty()) {          // the files ran dry: idle until another one is added
            free_.push_front(bi);
            loader_idle_ = true;
            cv_.notify_all();
            cv_.wait(lk, [&] { return stopped_ || !new_files_.empty(); });
            loader_idle_ = false;
            if (stopped_) break;
            continue;
        }
        ++batches_staged_;
        staged_.push_back(bi);
        cv_.notify_all();
    }
}

void MapPool::mapper_main() {
    for (;;) {
        int bi;
        {
            std::unique_lock<std::mutex> lk(mtx_);
            cv_.wait(lk, [&] { return stopped_ || !staged_.empty(); });
            if (stopped_) break;
            bi = staged_.front();
            staged_.pop_front();
            mapper_busy_ = true;
        }
        Batch &b = bufs_[bi];
        const size_t n = b.meta.size();
        std::vector<unc_hit_t> hits(n);
        if (b.used == 0) { grow(b, 1); b.raw[0] = 0; }
        const int rc = unc_map_batch(mapper_, (uint32_t)n, b.raw, b.off.data(), b.cal.data(), 0, nullptr, hits.data());
        if (rc != UNC_OK && rc != UNC_ERR_OVERFLOW) { std::cerr << "Error: " << unc_last_error() << "\n"; abort(); }
        std::vector<Paf> out;
        out.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const unc_hit_t &h = hits[i];
            Paf p(b.meta[i].id, (uint16_t)(b.meta[i].channel_idx + 1), b.meta[i].start_sample);
            p.set_read_len(h.rd_len);
            if (h.mapped) p.set_mapped(h.rd_st, h.rd_en, unc_index_seq_name(ix_, h.rid), h.rf_st, h.rf_en, h.rf_len, h.fwd != 0, (uint16_t)h.matches);
            // the read's residence on the device, first event taken up -> result written (the reference times
            // Mapper::map_read on the read's own thread, mapper.cpp:197; here reads share wavefronts in time slices)
            p.set_float(Paf::MAP_TIME, h.map_ms);
            if (h.status)   // the reference's SeedTracker is unbounded: say so instead of printing a silent unmapped line
                std::cerr << "Warning: read " << b.meta[i].id << ": device scratch overflow (status " << h.status
                          << ") even after re-mapping with more room; reported unmapped\n";
            out.push_back(std::move(p));
        }
        std::lock_guard<std::mutex> lk(mtx_);
        for (Paf &p : out) done_.push_back(std::move(p));
        free_.push_back(bi);
        mapper_busy_ = false;
        cv_.notify_all();
    }
}

std::vector<Paf> MapPool::update() {
    std::vector<Paf> ret;
    std::lock_guard<std::mutex> lk(mtx_);
    if (!started_ && !stopped_) {       // the first update starts the pipeline: every add_fast5 of the CLI has happened by then
        started_ = true;
        loader_ = std::thread(&MapPool::loader_main, this);
        mapper_thread_ = std::thread(&MapPool::mapper_main, this);
    }
    ret.swap(done_);
    return ret;
}

bool MapPool::running() {
    std::lock_guard<std::mutex> lk(mtx_);
    if (stopped_) return false;
    if (!started_) return !new_files_.empty() || !reader_.empty();
    // MapPool::running, map_pool.cpp:71-81: files left, a thread at work, or records not yet handed out
    return !(loader_idle_ && new_files_.empty() && staged_.empty() && !mapper_busy_ && done_.empty());
}

void MapPool::stop() {
    {
        std::lock_guard<std::mutex> lk(mtx_);
        stopped_ = true;
        cv_.notify_all();
    }
    if (loader_.joinable()) loader_.join();
    if (mapper_thread_.joinable()) mapper_thread_.join();
}

}  // namespace unc_host
