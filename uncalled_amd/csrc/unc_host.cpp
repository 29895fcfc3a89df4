// Host side of libuncalled_hip.so: index loader (BWA on-disk format -> HBM), per-GPU scratch, batch
// driver for the two kernels, PAF coordinate arithmetic (Mapper::set_ref_loc, mapper.cpp:703-728) and
// the extern "C" boundary declared in include/uncalled_hip.h.  Everything that maps reads runs on the
// device; there is no CPU mapping path in this library.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <chrono>
#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "r94_model_table.h"
#include "unc_dev_types.h"
#include "unc_kernels.h"

using namespace unc;

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(UNC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                               \
    } while (0)

extern "C" const char *unc_last_error(void) { return g_err; }
extern "C" const char *unc_version(void) { return "uncalled_hip 0.2 (gfx950)"; }
extern "C" int unc_device_pci_address(int device, char *out, int cap) {
    if (!out || cap < 13) return fail(UNC_ERR_ARG, "unc_device_pci_address: room for 13 characters is needed");
    HIPCHK(hipDeviceGetPCIBusId(out, cap, device));
    for (char *p = out; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells addresses in lower case
    return UNC_OK;
}
extern "C" void *unc_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void unc_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" void unc_params_default(unc_params_t *p) {
    // mapper.cpp:29-40
    p->seed_len = 22; p->min_rep_len = 0; p->max_rep_copy = 50; p->max_paths = 10000;
    p->max_consec_stay = 8; p->max_events = 30000; p->max_stay_frac = 0.5f; p->min_seed_prob = -3.75f;
    // event_detector.cpp:17-26
    p->window_length1 = 3; p->window_length2 = 6; p->threshold1 = 1.4f; p->threshold2 = 9.0f;
    p->peak_height = 0.2f; p->min_mean = 0.0f; p->max_mean = 400.0f;
    // seed_tracker.cpp:28-32
    p->min_map_len = 25; p->min_mean_conf = 6.00f; p->min_top_conf = 1.85f;
    // read_buffer.cpp:26-32
    p->bp_per_sec = 450.0f; p->sample_rate = 4000.0f; p->chunk_time = 1.0f; p->max_chunks = 1000000;
}

// ------------------------------------------------------------------ index
struct SeqAnn { std::string name; uint64_t offset, len; };

struct unc_index {
    int device = 0;
    uint64_t primary = 0, seq_len = 0, L2[5] = {0, 0, 0, 0, 0};
    int64_t l_pac = 0;
    std::vector<SeqAnn> seqs;
    std::vector<uint64_t> kmer_ranges;       // 2048, host copy
    float thresholds[64];
    std::vector<float> model;                // [3][1024] host copy
    float model_mean = 0, model_stdv = 0;
    // device
    uint32_t *d_bwt = nullptr;
    uint64_t *d_sa = nullptr;
    uint64_t *d_kmer_ranges = nullptr;
    float *d_model = nullptr;
    uint16_t *d_kmer_valid = nullptr;
    uint64_t *d_sa_dense = nullptr;
    uint32_t *d_fm32 = nullptr;
    float *d_model4 = nullptr;
    uint64_t device_bytes = 0;
    DevIndex dev;
};

static bool read_file(const std::string &fn, std::vector<char> &out) {
    FILE *fp = fopen(fn.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize((size_t)sz);
    bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
    fclose(fp);
    return ok;
}

// PoreModel(vector, cmpl=true): pore_model.hpp:77-103, init_kmer 58-62, init_stdv 48-56.  Host libm
// builds the three float tables once; the device only ever consumes the resulting float32 values.
static void build_model(unc_index *ix) {
    ix->model.assign(3 * NKMER, 0.0f);
    float *mu = ix->model.data(), *v2 = mu + NKMER, *ld = mu + 2 * NKMER;
    float model_mean = 0;
    for (uint32_t kmer = 0; kmer < (uint32_t)NKMER; ++kmer) {
        float mean, stdv;
        memcpy(&mean, &UNC_R94_MEAN_STDV_BITS[2 * kmer], 4);
        memcpy(&stdv, &UNC_R94_MEAN_STDV_BITS[2 * kmer + 1], 4);
        const uint32_t k = kmer ^ KMASK;   // complement model, mapper.cpp:57 / bp.hpp:77-80
        mu[k] = mean;
        float tv = 2 * stdv;
        tv = tv * stdv;
        v2[k] = tv;
        ld[k] = (float)log(sqrt(M_PI * (double)v2[k]));
        model_mean = model_mean + mean;
    }
    model_mean = model_mean / (float)NKMER;
    float acc = 0;
    for (uint32_t k = 0; k < (uint32_t)NKMER; ++k) {
        float d = mu[k] - model_mean;
        acc = (float)((double)acc + (double)d * (double)d);
    }
    ix->model_mean = model_mean;
    ix->model_stdv = sqrtf(acc / (float)NKMER);
}

// mapper.cpp:123-157
static int parse_uncl(const std::string &fn, const char *preset, float *thr) {
    std::vector<char> buf;
    if (!read_file(fn, buf)) return fail(UNC_ERR_IO, "failed to load uncalled index %s", fn.c_str());
    buf.push_back(0);
    bool found = false;
    char *save = nullptr;
    for (char *line = strtok_r(buf.data(), "\n", &save); line; line = strtok_r(nullptr, "\n", &save)) {
        char *s1 = nullptr;
        char *name = strtok_r(line, "\t", &s1);
        char *fn_str = strtok_r(nullptr, "\t", &s1);
        if (!name || !fn_str) continue;
        if (preset && preset[0] && strcmp(name, preset) != 0) continue;
        uint8_t bin = 63;
        char *s2 = nullptr;
        for (char *tok = strtok_r(fn_str, ",", &s2); tok; tok = strtok_r(nullptr, ",", &s2)) {
            thr[bin] = (float)atof(tok);
            bin--;
        }
        for (; bin < 64; bin--) thr[bin] = thr[bin + 1];
        found = true;
    }
    if (!found) return fail(UNC_ERR_IO, "preset '%s' not found in %s", preset ? preset : "", fn.c_str());
    return UNC_OK;
}

static int parse_ann(const std::string &fn, unc_index *ix) {
    FILE *fp = fopen(fn.c_str(), "r");
    if (!fp) return fail(UNC_ERR_IO, "failed to load BWA index: %s", fn.c_str());
    long long l_pac; int n_seqs; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &l_pac, &n_seqs, &seed) != 3) { fclose(fp); return fail(UNC_ERR_IO, "bad header in %s", fn.c_str()); }
    ix->l_pac = l_pac;
    char line[8192], name[4096];
    for (int i = 0; i < n_seqs; ++i) {
        unsigned gi;
        if (fscanf(fp, "%u%4095s", &gi, name) != 2) { fclose(fp); return fail(UNC_ERR_IO, "bad record in %s", fn.c_str()); }
        if (!fgets(line, sizeof line, fp)) line[0] = 0;
        long long off; int len, n_ambs;
        if (fscanf(fp, "%lld%d%d", &off, &len, &n_ambs) != 3) { fclose(fp); return fail(UNC_ERR_IO, "bad record in %s", fn.c_str()); }
        ix->seqs.push_back(SeqAnn{name, (uint64_t)off, (uint64_t)len});
    }
    fclose(fp);
    return UNC_OK;
}

extern "C" void unc_index_free(unc_index_t *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->d_bwt) (void)hipFree(ix->d_bwt);
    if (ix->d_sa) (void)hipFree(ix->d_sa);
    if (ix->d_kmer_ranges) (void)hipFree(ix->d_kmer_ranges);
    if (ix->d_model) (void)hipFree(ix->d_model);
    if (ix->d_kmer_valid) (void)hipFree(ix->d_kmer_valid);
    if (ix->d_sa_dense) (void)hipFree(ix->d_sa_dense);
    if (ix->d_fm32) (void)hipFree(ix->d_fm32);
    if (ix->d_model4) (void)hipFree(ix->d_model4);
    delete ix;
}

// WHERE in the HBM a mapper's slots and node pool lie decides which of two speeds it runs at, for the life of the allocation: the same
// library maps the same 50 000 E. coli reads in 2 040 or in 2 250 ms (round 6, tools/dev/placement_probe.py: mapper instances created
// and freed one after the other at the SAME virtual addresses alternate between the two; one created while 60 GB of other memory is
// held landed on the fast one three times out of three).  Virtual addresses say nothing; what differs is the physical memory behind them: what the
// process -- e.g. an index build -- has used and given back is handed out again first, in pieces, and a kernel that walks tens of
// gigabytes at random pays for the pieces (in latency, not in throughput: DESIGN.md section 5).  So the big random-access allocations (the slots, the node pool;
// the index's tables likewise) are made while a SPACER holds what was given back, up to 64 GB, and the spacer is freed right after:
// the streaming buffers allocated later (raw signal, event means) take its place.  UNC_PLACEMENT_SPACER_GB names another size (0: none).
struct PlacementSpacer {
    void *p = nullptr;
    size_t bytes = 0;
    explicit PlacementSpacer(size_t need_after) {
        size_t cap = 64ull << 30;
        if (const char *e = getenv("UNC_PLACEMENT_SPACER_GB")) cap = (size_t)strtoull(e, nullptr, 10) << 30;
        // (a mapper of a few slots, an index of a few megabytes: what is about to be allocated sits in the caches or is walked by a
        // handful of wavefronts -- where it lies does not matter, and 64 GB allocated and freed per creation would be all cost)
        if (need_after < (1ull << 30)) return;
        size_t free_b = 0, total_b = 0;
        if (!cap || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
        const size_t margin = 16ull << 30;       // (left alone beside what is about to be allocated)
        if (free_b <= need_after + margin) return;
        const size_t want = std::min(cap, free_b - need_after - margin);
        if (want < (2ull << 30)) return;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return; }
        bytes = want;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    ~PlacementSpacer() { release(); }
    PlacementSpacer(const PlacementSpacer &) = delete;
    PlacementSpacer &operator=(const PlacementSpacer &) = delete;
};

extern "C" int unc_index_load(const char *bwa_prefix, const char *idx_preset, int device, unc_index_t **out) {
    if (!bwa_prefix || !out) return fail(UNC_ERR_ARG, "null argument");
    *out = nullptr;
    std::string prefix(bwa_prefix);
    unc_index *ix = new unc_index();
    ix->device = device;
    struct Guard { unc_index *p; ~Guard() { if (p) unc_index_free(p); } } guard{ix};

    // .bwt = u64 primary; u64 L2[1..4]; u32 words (Occ counts interleaved every 128 symbols)
    std::vector<char> bwt;
    if (!read_file(prefix + ".bwt", bwt) || bwt.size() < 40) return fail(UNC_ERR_IO, "failed to load BWA index: %s.bwt", bwa_prefix);
    memcpy(&ix->primary, bwt.data(), 8);
    memcpy(&ix->L2[1], bwt.data() + 8, 32);
    ix->seq_len = ix->L2[4];
    const size_t n_words = (bwt.size() - 40) / 4;
    const uint64_t n = ix->seq_len;
    if (n_words != (n + 15) / 16 + 8 * ((n + 127) / 128 + 1)) return fail(UNC_ERR_IO, "%s.bwt: size does not match seq_len", bwa_prefix);
    if (n >= (1ull << 34)) return fail(UNC_ERR_ARG, "reference too large: seq_len %llu >= 2^34", (unsigned long long)n);

    // .sa = u64 primary; u64 x4; u64 intv; u64 seq_len; u64 sa[1..]
    std::vector<char> sa;
    if (!read_file(prefix + ".sa", sa) || sa.size() < 56) return fail(UNC_ERR_IO, "failed to load BWA index: %s.sa", bwa_prefix);
    uint64_t sa_primary, sa_intv, sa_seq_len;
    memcpy(&sa_primary, sa.data(), 8);
    memcpy(&sa_intv, sa.data() + 40, 8);
    memcpy(&sa_seq_len, sa.data() + 48, 8);
    if (sa_primary != ix->primary || sa_seq_len != n || sa_intv != 32) return fail(UNC_ERR_IO, "%s.sa does not match the .bwt (interval must be 32)", bwa_prefix);
    const uint64_t n_sa = (n + 32) / 32;
    if (sa.size() != 56 + (n_sa - 1) * 8) return fail(UNC_ERR_IO, "%s.sa: unexpected size", bwa_prefix);

    int rc = parse_ann(prefix + ".ann", ix);
    if (rc) return rc;
    rc = parse_uncl(prefix + ".uncl", idx_preset ? idx_preset : "default", ix->thresholds);
    if (rc) return rc;
    build_model(ix);

    HIPCHK(hipSetDevice(device));
    // pad the BWT to whole 64-byte blocks so that block loads of the last (partial) block stay in bounds
    const size_t bwt_bytes = ((n_words * 4 + 63) / 64 + 1) * 64;
    // (the index's tables are read at random by every FM step: allocated behind a spacer like the mapper's slots -- PlacementSpacer)
    PlacementSpacer spacer(bwt_bytes + n_sa * 8 + ((size_t)n_words * 16 + 2) / 2 * 12 + (size_t)n_words * 2 + (64u << 20));
    HIPCHK(hipMalloc((void **)&ix->d_bwt, bwt_bytes));
    HIPCHK(hipMemset(ix->d_bwt, 0, bwt_bytes));
    HIPCHK(hipMemcpy(ix->d_bwt, bwt.data() + 40, n_words * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&ix->d_sa, n_sa * 8));
    {
        std::vector<uint64_t> h(n_sa);
        h[0] = ~0ull;   // bwt_restore_sa: sa[0] = -1
        memcpy(h.data() + 1, sa.data() + 56, (n_sa - 1) * 8);
        HIPCHK(hipMemcpy(ix->d_sa, h.data(), n_sa * 8, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void **)&ix->d_kmer_ranges, 2 * NKMER * 8));
    HIPCHK(hipMalloc((void **)&ix->d_model, 3 * NKMER * 4));
    HIPCHK(hipMemcpy(ix->d_model, ix->model.data(), 3 * NKMER * 4, hipMemcpyHostToDevice));
    {
        std::vector<float> m4(4 * NKMER, 0.0f);
        for (int k = 0; k < NKMER; ++k) { m4[4 * k] = ix->model[k]; m4[4 * k + 1] = ix->model[NKMER + k]; m4[4 * k + 2] = ix->model[2 * NKMER + k]; }
        HIPCHK(hipMalloc((void **)&ix->d_model4, 4 * NKMER * 4));
        HIPCHK(hipMemcpy(ix->d_model4, m4.data(), 4 * NKMER * 4, hipMemcpyHostToDevice));
    }
    ix->device_bytes = bwt_bytes + n_sa * 8 + 2 * NKMER * 8 + 3 * NKMER * 4;

    DevIndex &d = ix->dev;
    d.bwt = ix->d_bwt; d.sa = ix->d_sa; d.kmer_ranges = ix->d_kmer_ranges; d.model = ix->d_model; d.model4 = ix->d_model4;
    d.primary = ix->primary; d.seq_len = n;
    for (int i = 0; i < 5; ++i) d.L2[i] = ix->L2[i];
    memcpy(d.thresholds, ix->thresholds, sizeof d.thresholds);

    d.sa_dense = nullptr;
    // Dense SA (6 bytes per BWT row, fm_dev.h: 56 MB for E. coli, 37 GB for GRCh38; 8 bytes in rounds 2-5): every row walks LF to
    // its sampled row once, here, instead of on every seed.  UNC_DENSE_SA=0 keeps the BWA sampling only.
    {
        const char *env = getenv("UNC_DENSE_SA");
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        free_b += spacer.bytes;
        const size_t need = ((n + 2) / 2) * 12 + 16;      // rows 0 .. n in pairs (12 bytes) + the word a load reads past the last entry
        if (!(env && env[0] == '0') && need < free_b / 2) {
            HIPCHK(hipMalloc((void **)&ix->d_sa_dense, need));
            launch_dense_sa(d, ix->d_sa_dense, nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
            {   // self-check: 16 384 rows spread over the whole table (the last row included) against the BWA-format walk
                uint32_t *d_bad = nullptr, bad = 0;
                HIPCHK(hipMalloc((void **)&d_bad, 4));
                HIPCHK(hipMemset(d_bad, 0, 4));
                launch_dense_sa_check(d, ix->d_sa_dense, 16384, d_bad, nullptr);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
                (void)hipFree(d_bad);
                if (bad) return fail(UNC_ERR_HIP, "dense SA self-check failed on %u of 16384 probe rows", bad);
            }
            d.sa_dense = ix->d_sa_dense;
            ix->device_bytes += need;
        }
    }
    // References of fewer than 2^32 rows (the reference's own 32-bit regime, range.hpp:37) get the 32-bit rank table k_map
    // works on there (fm_dev.h: 32 bytes per 64 symbols, counts with L2 folded in, symbols as bit planes), checked against the
    // BWA-format arithmetic on 16 384 pseudo-random steps.
    d.fm32 = nullptr;
    if (n < 0xFFFFFF00ull) {
        const uint32_t n_blk = (uint32_t)((n + 63) / 64 + 1);
        HIPCHK(hipMalloc((void **)&ix->d_fm32, (size_t)n_blk * 32));
        launch_build_fm32(d, ix->d_fm32, n_blk, nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
        d.fm32 = ix->d_fm32;
        uint32_t *d_bad = nullptr, bad = 0;
        HIPCHK(hipMalloc((void **)&d_bad, 4));
        HIPCHK(hipMemset(d_bad, 0, 4));
        launch_fm32_check(d, 16384, d_bad, nullptr);
        const hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        HIPCHK(e1); HIPCHK(e2);
        if (bad) return fail(UNC_ERR_HIP, "32-bit rank table self-check failed on %u of 16384 probe steps", bad);
        ix->device_bytes += (size_t)n_blk * 32;
    }
    // the 1024 k-mer ranges, derived on the device exactly as BwaIndex::load_index does (bwa_index.hpp:124-132)
    launch_kmer_ranges(d, ix->d_kmer_ranges, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    ix->kmer_ranges.resize(2 * NKMER);
    HIPCHK(hipMemcpy(ix->kmer_ranges.data(), ix->d_kmer_ranges, 2 * NKMER * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < NKMER; ++k) {
        uint64_t s = ix->kmer_ranges[2 * k], e = ix->kmer_ranges[2 * k + 1];
        if (s <= e && e - s + 1 >= (1ull << KEY_LEN_BITS)) return fail(UNC_ERR_ARG, "k-mer %d occurs more than 2^30 times: unsupported", k);
    }
    {
        // narrow sort keys: start | length | 16-bit creation index in 64 bits when the reference allows it
        uint64_t max_len = 0;
        for (int k = 0; k < NKMER; ++k) {
            uint64_t s = ix->kmer_ranges[2 * k], e = ix->kmer_ranges[2 * k + 1];
            if (s <= e && e - s > max_len) max_len = e - s;
        }
        uint32_t start_bits = 1, len_bits = 1;
        while ((n >> start_bits) != 0) ++start_bits;
        while ((max_len >> len_bits) != 0) ++len_bits;
        const char *wide = getenv("UNC_WIDE_KEYS");
        // (narrow keys also mean 32-bit rows in k_map: only with the 32-bit rank table)
        ix->dev.key_len_bits = (start_bits + len_bits + 16 <= 64 && ix->dev.fm32 && !(wide && wide[0] == '1')) ? len_bits : 0;
        ix->dev.pad_ = 0;
    }
    {
        // the grid the seed clusters of a read are bucketed in (k_map.hip, add_seed): buckets of 2^shift rows, at most 2^15 per read,
        // at least 2^12 rows wide, so that a window of 32768 rows (max_events <= 65535 is checked per mapper) spans few buckets
        uint32_t bits = 1;
        while ((n >> bits) != 0) ++bits;
        uint32_t shift = bits > 15 + BUCKET_SHIFT_MIN ? bits - 15 : BUCKET_SHIFT_MIN;
        // UNC_BUCKET_SHIFT (tests): narrower buckets, so that the small test references have windows of many buckets
        // (several gathers per seed) and clusters that move between buckets
        if (const char *e = getenv("UNC_BUCKET_SHIFT")) { const long v = atol(e); if (v >= 2 && v <= 30 && (n >> v) < (1ull << 22)) shift = (uint32_t)v; }
        ix->dev.bucket_shift = shift;
        ix->dev.n_buckets = (uint32_t)(n >> shift) + 2u;
    }
    {
        uint16_t valid[WAVE];
        for (int l = 0; l < WAVE; ++l) {
            valid[l] = 0;
            for (int j = 0; j < NKMER / WAVE; ++j) {
                const int k = j * WAVE + l;
                if (ix->kmer_ranges[2 * k] <= ix->kmer_ranges[2 * k + 1]) valid[l] |= (uint16_t)(1u << j);
            }
        }
        HIPCHK(hipMalloc((void **)&ix->d_kmer_valid, sizeof valid));
        HIPCHK(hipMemcpy(ix->d_kmer_valid, valid, sizeof valid, hipMemcpyHostToDevice));
        ix->dev.kmer_valid = ix->d_kmer_valid;
    }
    guard.p = nullptr;
    *out = ix;
    return UNC_OK;
}

extern "C" uint64_t unc_index_size(const unc_index_t *ix) { return ix->seq_len; }
extern "C" int32_t unc_index_n_seqs(const unc_index_t *ix) { return (int32_t)ix->seqs.size(); }
extern "C" const char *unc_index_seq_name(const unc_index_t *ix, int32_t rid) {
    return (rid >= 0 && (size_t)rid < ix->seqs.size()) ? ix->seqs[rid].name.c_str() : "";
}
extern "C" uint64_t unc_index_seq_len(const unc_index_t *ix, int32_t rid) {
    return (rid >= 0 && (size_t)rid < ix->seqs.size()) ? ix->seqs[rid].len : 0;
}
extern "C" uint64_t unc_index_device_bytes(const unc_index_t *ix) { return ix->device_bytes; }

// bns_pos2rid behind BwaIndex::translate_loc, bwa_index.hpp:213-220
extern "C" uint64_t unc_index_translate_loc(const unc_index_t *ix, uint64_t sa_loc, int32_t *rid, uint64_t *ref_loc) {
    *rid = -1;
    if ((int64_t)sa_loc >= ix->l_pac || ix->seqs.empty()) return 0;
    int left = 0, mid = 0, right = (int)ix->seqs.size();
    while (left < right) {
        mid = (left + right) >> 1;
        if (sa_loc >= ix->seqs[mid].offset) {
            if (mid == (int)ix->seqs.size() - 1) break;
            if (sa_loc < ix->seqs[mid + 1].offset) break;
            left = mid + 1;
        } else {
            right = mid;
        }
    }
    *rid = mid;
    *ref_loc = sa_loc - ix->seqs[mid].offset;
    return ix->seqs[mid].len;
}

extern "C" void unc_index_kmer_ranges(const unc_index_t *ix, uint64_t *out) { memcpy(out, ix->kmer_ranges.data(), 2 * NKMER * 8); }
extern "C" void unc_index_thresholds(const unc_index_t *ix, float *out) { memcpy(out, ix->thresholds, sizeof ix->thresholds); }
extern "C" void unc_index_model_tables(const unc_index_t *ix, float *mu, float *v2, float *ld, float *mm, float *ms) {
    memcpy(mu, ix->model.data(), NKMER * 4);
    memcpy(v2, ix->model.data() + NKMER, NKMER * 4);
    memcpy(ld, ix->model.data() + 2 * NKMER, NKMER * 4);
    *mm = ix->model_mean;
    *ms = ix->model_stdv;
}

template <class T> struct DevBuf;
extern "C" int unc_self_align(const unc_index_t *ix, const char *bwa_prefix, uint32_t sample_dist, uint32_t cap, uint64_t *lens,
                              uint32_t *full_len, uint64_t max_paths, uint64_t *n_paths);

template <class T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

extern "C" int unc_fm_get_neighbor(const unc_index_t *ix, uint32_t n, const uint64_t *starts, const uint64_t *ends,
                                   const uint8_t *bases, uint64_t *out_s, uint64_t *out_e) {
    HIPCHK(hipSetDevice(ix->device));
    DevBuf<uint64_t> s, e, os, oe;
    DevBuf<uint8_t> b;
    HIPCHK(s.alloc(n)); HIPCHK(e.alloc(n)); HIPCHK(os.alloc(n)); HIPCHK(oe.alloc(n)); HIPCHK(b.alloc(n));
    HIPCHK(hipMemcpy(s.p, starts, n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e.p, ends, n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.p, bases, n, hipMemcpyHostToDevice));
    launch_fm_neighbor(ix->dev, n, s.p, e.p, b.p, os.p, oe.p, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_s, os.p, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_e, oe.p, n * 8, hipMemcpyDeviceToHost));
    return UNC_OK;
}

extern "C" int unc_fm_sa(const unc_index_t *ix, uint32_t n, const uint64_t *rows, uint64_t *out) {
    HIPCHK(hipSetDevice(ix->device));
    DevBuf<uint64_t> r, o;
    HIPCHK(r.alloc(n)); HIPCHK(o.alloc(n));
    HIPCHK(hipMemcpy(r.p, rows, n * 8, hipMemcpyHostToDevice));
    launch_fm_sa(ix->dev, n, r.p, o.p, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, o.p, n * 8, hipMemcpyDeviceToHost));
    return UNC_OK;
}

extern "C" int unc_match_probs(const unc_index_t *ix, uint32_t n, const float *levels, float *out) {
    HIPCHK(hipSetDevice(ix->device));
    DevBuf<float> l, o;
    HIPCHK(l.alloc(n)); HIPCHK(o.alloc((size_t)n * NKMER));
    HIPCHK(hipMemcpy(l.p, levels, n * 4, hipMemcpyHostToDevice));
    launch_match_probs(ix->dev, n, l.p, o.p, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, o.p, (size_t)n * NKMER * 4, hipMemcpyDeviceToHost));
    return UNC_OK;
}

// ------------------------------------------------------------------ mapper
struct unc_mapper {
    const unc_index *ix = nullptr;
    unc_params_t P;
    uint32_t n_slots = 0, n_waves = 0, slice_events = 1024, ev_rpw = 64;
    DevSched sched{};             // ctl != null: sliced batch scheduler (n_slots > n_waves)
    DevScratch sc;
    uint64_t device_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double win_start_ms = 0, win_end_ms = 0;      // k_map of the last batch on the process-wide time axis (unc_mapper_last_window)
    float ms_events = 0, ms_map = 0;
    double wave_busy = 0;          // mean wave lifetime / k_map duration of the last batch (1 = no queue tail)
    float wall_khz = 0;            // device wall clock rate (ticks per ms)
    bool profile = false;          // launch the instantiation of k_map that counts cycles per phase
    DevPool pool{};                // nodes of the seed-cluster grids, shared by every read in flight
    // the pool is sized by NEED: created by a rule of thumb, cut to four times the most chunks that were ever out at once when it
    // holds more than eight times that, doubled when found dry (pool_fit, unc_mapper_pool_usage); a caller who named pool_chunks
    // keeps that number
    bool pool_auto = false;
    uint32_t pool_floor = 16, pool_created = 0, pool_hw_last = 0, pool_hw_max = 0, pool_resizes = 0;
    // a batch that unc_map_batch_begin has launched and unc_map_batch_end has not yet collected
    struct Pending {
        bool active = false;
        uint32_t n_reads = 0, grid = 0;
        hipStream_t st = nullptr;
        DevReads rd{};
        bool t1 = false;
        std::vector<uint64_t> lens;      // samples per read (fill_hit)
    } pend;
    DevScratch big{};              // scratch with a larger node allowance for the reads that outgrew a slot's (kept between batches)
    uint64_t big_cap = 0;
    size_t big_slots = 0;
    bool big_at_limit = false;     // big_slots is all the free HBM allowed
    uint32_t *d_list = nullptr; size_t list_cap = 0;     // read ids of a re-map round
    uint32_t remap_reads = 0;      // reads of the last batch that were mapped again with more room, and what that cost
    float remap_ms = 0;
    // read order UNC_ORDER_T1 (unc_mapper_set_read_order): sources_added_ travels from a read to the next one, across batches
    int read_order = 0;
    uint32_t *d_flags_in = nullptr, *d_flags_out = nullptr; size_t flags_cap = 0;
    uint32_t carry_flags[NKMER / 32] = {0};     // what the last read of the previous batch left set
    uint32_t carry_reads = 0, carry_rounds = 0; // reads of the last batch mapped again because their predecessor left flags set
    float carry_ms = 0;
    uint32_t *d_next = nullptr;
    // per-batch buffers (grown on demand)
    int16_t *d_raw = nullptr; size_t raw_cap = 0;
    uint64_t *d_offsets = nullptr; uint64_t *d_moff = nullptr; unc_calib_t *d_calib = nullptr;
    unc_evt_info_t *d_info = nullptr; DevResult *d_results = nullptr; size_t reads_cap = 0;
    float *d_means = nullptr; size_t means_cap = 0;
    std::vector<uint64_t> h_moff;
    std::vector<unc_evt_info_t> h_info;
    std::vector<DevResult> h_results;
    // trace state
    uint32_t trace_n = 0;
    bool trace_active = false;
};

static void free_scratch(DevScratch &sc) {
    if (sc.base) (void)hipFree(sc.base);
    memset(&sc, 0, sizeof sc);
}

// Regions of one slot (DevScratch), each 256-byte aligned; everything below 4 GB so that kernels address a slot as
// uniform base + 32-bit offset.
static int scratch_layout(DevScratch &sc, const unc_params_t &P, uint32_t max_clusters, uint32_t max_seed_paths, const DevIndex &dix) {
    const uint32_t n_buckets = dix.n_buckets;
    const bool wide = dix.key_len_bits == 0;          // 128-bit sort keys: the runs of child keys hold 16-byte elements
    memset(&sc, 0, sizeof sc);
    sc.max_paths = P.max_paths;
    uint32_t kc = 64;
    while (kc < P.max_paths) kc <<= 1;
    sc.keys_cap = kc;
    sc.max_seed_paths = max_seed_paths;
    sc.max_clusters = max_clusters;
    uint64_t off = 0;
    auto region = [&off](uint64_t bytes) { const uint64_t o = off; off += (bytes + 255) & ~255ull; return o; };
    const uint64_t max_nodes = max_clusters / 4 ? max_clusters / 4 : 1;     // nodes of seed clusters a read may take from the pool
    const uint64_t o_paths = region(2ull * sc.max_paths * sizeof(PathRec));
    const uint64_t o_levels = region(LEVEL_RING * 4);
    const uint64_t o_order = region(2ull * sc.max_paths * 4);
    const uint64_t o_keys = region(2ull * sc.keys_cap * sizeof(SortKey));
    const uint64_t o_seedp = region((uint64_t)max_seed_paths * sizeof(SeedPath));
    const uint64_t o_tasks = region((uint64_t)WAVE * MAX_REP_COPY_LIMIT * 8);
    const uint64_t o_cld = region(((uint64_t)n_buckets + 4) * 4);
    const uint64_t o_clc = region((max_nodes / CHUNK_NODES + 1) * 4);
    const uint64_t o_state = region(sizeof(SlotState));
    const uint64_t o_streams = region(6ull * sc.max_paths * (wide ? 16 : 8));
    const uint64_t o_info = region((uint64_t)sc.max_paths * 8);
    const uint64_t o_tmp = region((uint64_t)sc.max_paths * 8);
    if (off >= (1ull << 32)) return fail(UNC_ERR_ARG, "per-read scratch of %llu bytes does not fit 32-bit offsets (max_clusters %u)", (unsigned long long)off, max_clusters);
    sc.off_paths = (uint32_t)o_paths; sc.off_levels = (uint32_t)o_levels; sc.off_order = (uint32_t)o_order; sc.off_keys = (uint32_t)o_keys;
    sc.off_seedp = (uint32_t)o_seedp; sc.off_tasks = (uint32_t)o_tasks; sc.off_cl_dir = (uint32_t)o_cld; sc.off_cl_chunks = (uint32_t)o_clc;
    sc.off_state = (uint32_t)o_state;
    sc.off_streams = (uint32_t)o_streams; sc.off_info = (uint32_t)o_info; sc.off_tmp = (uint32_t)o_tmp;
    sc.slot_bytes = off;
    return UNC_OK;
}

static uint64_t scratch_slot_bytes(const unc_params_t &P, uint32_t max_clusters, uint32_t max_seed_paths, const DevIndex &dix) {
    DevScratch t;
    return scratch_layout(t, P, max_clusters, max_seed_paths, dix) == UNC_OK ? t.slot_bytes : ~0ull;
}

static int alloc_scratch(DevScratch &sc, const unc_params_t &P, size_t n_slots, uint32_t max_clusters, uint32_t max_seed_paths,
                         size_t *bytes_out, const DevIndex &dix) {
    int rc = scratch_layout(sc, P, max_clusters, max_seed_paths, dix);
    if (rc) return rc;
    const size_t bytes = (size_t)n_slots * sc.slot_bytes;
    HIPCHK(hipMalloc((void **)&sc.base, bytes));
    // SlotState of every slot starts zeroed (done = 0, nothing parked)
    HIPCHK(hipMemset2D(sc.base + sc.off_state, sc.slot_bytes, 0, sizeof(SlotState), n_slots));
    if (bytes_out) *bytes_out = bytes;
    return UNC_OK;
}

static void free_pool(DevPool &p) {
    void *ptrs[] = {p.nodes, p.q, p.cells};
    for (void *x : ptrs) if (x) (void)hipFree(x);
    memset(&p, 0, sizeof p);
}

// XCDs of a device: workgroups of a small grid report the XCD they run on (hardware register), the distinct answers are counted; once
// per device and process.  (1 on a device with one XCD -- or whose partition mode shows one --: the rings then stay one pair.)
static uint32_t xcd_count(int device) {
    static std::mutex mu;
    static std::map<int, uint32_t> known;
    std::lock_guard<std::mutex> lk(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    uint32_t n = 1;
    uint32_t *d = nullptr;
    constexpr uint32_t NB = 256;
    if (hipMalloc((void **)&d, NB * sizeof(uint32_t)) == hipSuccess) {
        uint32_t h[NB];
        launch_xcd_probe(d, NB, nullptr);
        if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            uint32_t seen = 0;
            for (uint32_t i = 0; i < NB; ++i) seen |= 1u << (h[i] & 15u);
            n = (uint32_t)__builtin_popcount(seen);
            // the ids must be 0 .. n - 1 (sched_part takes the id modulo n_parts)
            if (seen != (n >= 32 ? ~0u : (1u << n) - 1u)) n = 1;
        }
        (void)hipFree(d);
    }
    (void)hipGetLastError();
    known[device] = n;
    return n;
}

static int alloc_pool(DevPool &p, uint32_t n_chunks, size_t *bytes_out) {
    memset(&p, 0, sizeof p);
    uint32_t cap = 64;
    while (cap < n_chunks) cap <<= 1;
    p.cap_mask = cap - 1; p.n_chunks = n_chunks;
    HIPCHK(hipMalloc((void **)&p.nodes, (size_t)n_chunks * POOL_CHUNK_BYTES));
    HIPCHK(hipMalloc((void **)&p.q, sizeof(PoolQueue)));
    HIPCHK(hipMalloc((void **)&p.cells, (size_t)cap * sizeof(SchedCell)));
    launch_pool_init(p, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (bytes_out) *bytes_out += (size_t)n_chunks * POOL_CHUNK_BYTES + (size_t)cap * sizeof(SchedCell);
    return UNC_OK;
}

static SlotState *slot_state(const DevScratch &sc, size_t slot) { return reinterpret_cast<SlotState *>(sc.base + slot * sc.slot_bytes + sc.off_state); }

extern "C" void unc_mapper_free(unc_mapper_t *m) {
    if (!m) return;
    (void)hipSetDevice(m->ix->device);
    free_scratch(m->sc);
    free_scratch(m->big);
    void *ptrs[] = {m->d_next, m->d_raw, m->d_offsets, m->d_moff, m->d_calib, m->d_info, m->d_results, m->d_means,
                    m->sched.ctl, m->sched.free_cells, m->sched.park_cells, m->d_list, m->d_flags_in, m->d_flags_out};
    free_pool(m->pool);
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : m->ev) if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

// one reference event per process: the k_map windows of batches on DIFFERENT mappers (streams) on one time axis
static hipEvent_t g_ref_event = nullptr;
static std::mutex g_ref_mutex;
static int ref_event(hipEvent_t *out) {
    std::lock_guard<std::mutex> lock(g_ref_mutex);
    if (!g_ref_event) {
        HIPCHK(hipEventCreate(&g_ref_event));
        HIPCHK(hipEventRecord(g_ref_event, nullptr));
        HIPCHK(hipEventSynchronize(g_ref_event));
    }
    *out = g_ref_event;
    return UNC_OK;
}
extern "C" int unc_mapper_create(const unc_index_t *ix, const unc_params_t *p, const unc_mapper_opts_t *opts, unc_mapper_t **out) {
    if (!ix || !p || !out) return fail(UNC_ERR_ARG, "null argument");
    *out = nullptr;
    if (p->seed_len != UNC_SEED_LEN) return fail(UNC_ERR_ARG, "seed_len must be %d", UNC_SEED_LEN);
    if (p->window_length1 != UNC_WINDOW1 || p->window_length2 != UNC_WINDOW2) return fail(UNC_ERR_ARG, "event windows must be %d/%d", UNC_WINDOW1, UNC_WINDOW2);
    if (p->max_paths == 0 || p->max_paths > 65535) return fail(UNC_ERR_ARG, "max_paths must be in 1..65535");
    if (p->max_rep_copy > (uint32_t)MAX_REP_COPY_LIMIT) return fail(UNC_ERR_ARG, "max_rep_copy must be <= %d", MAX_REP_COPY_LIMIT);
    if (p->max_consec_stay > 255) return fail(UNC_ERR_ARG, "max_consec_stay must be <= 255");
    if (p->max_events > 65535) return fail(UNC_ERR_ARG, "max_events must be <= 65535");
    HIPCHK(hipSetDevice(ix->device));
    unc_mapper *m = new unc_mapper();
    struct Guard { unc_mapper *p; ~Guard() { if (p) unc_mapper_free(p); } } guard{m};
    m->ix = ix;
    m->P = *p;
    memset(&m->sc, 0, sizeof m->sc);
    uint32_t n_slots = opts ? opts->n_slots : 0, n_waves = opts ? opts->n_waves : 0;
    if (n_waves == 0) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ix->device));
        n_waves = (uint32_t)prop.multiProcessorCount * map_kernel_waves_per_cu();
        if (n_slots && n_slots < n_waves) n_waves = n_slots;
    }
    if (n_slots == 0) {
        // three times more reads in flight than wavefronts, so that the long reads of a batch are found early (DevSched); 2 - 3 MB of
        // scratch per slot, bounded by a third of the free HBM.  (Rounds 2-5: four times.  With the rings per XCD, E. coli maps 5 - 8 %
        // faster at 2.5 - 3 and at 4.25 - 5 reads per wavefront than at 3.5 - 4, chr20 3 % faster, GRCh38 the same, and a quarter of the
        // slots' memory is saved: profiles/r06_ab_slots_*.log.)
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const uint32_t msp0 = (opts && opts->max_seed_paths) ? opts->max_seed_paths : 2 * p->max_paths;
        const uint32_t mcl0 = (opts && opts->max_clusters) ? opts->max_clusters : (1u << 20);
        const size_t per_slot = scratch_slot_bytes(*p, mcl0, msp0, ix->dev);
        size_t want = (size_t)n_waves * 3, fit = free_b / 3 / per_slot;
        n_slots = (uint32_t)(want < fit ? want : fit);
        if (n_slots > 512) n_slots -= n_slots % 512;      // (whole shares of 64 slots and more for the rings of up to eight XCDs)
        if (n_slots < n_waves) n_slots = n_waves;
    }
    if (n_slots < n_waves) n_waves = n_slots;
    m->n_slots = n_slots;
    m->n_waves = n_waves;
    m->slice_events = (opts && opts->slice_events) ? opts->slice_events : 1024;
    m->ev_rpw = (opts && opts->events_reads_per_wave) ? opts->events_reads_per_wave : 64;
    if (m->ev_rpw > (uint32_t)WAVE) return fail(UNC_ERR_ARG, "events_reads_per_wave must be in 1..64");
    size_t bytes = 0;
    // every seed of an event is either an ended parent or a surviving child: 2 * max_paths bounds the per-event list
    const uint32_t msp = (opts && opts->max_seed_paths) ? opts->max_seed_paths : 2 * p->max_paths;
    // per read: the table of bucket heads of its seed-cluster grid and the list of its pool chunks; max_clusters / 4 is the number
    // of nodes a read may take from the pool below (its allowance)
    const uint32_t mcl = (opts && opts->max_clusters) ? opts->max_clusters : (1u << 20);
    // (what the slots and the pool will take, for the spacer: see PlacementSpacer)
    const size_t pool_per_slot = ix->seq_len >= (1ull << 31) ? 32 : ix->seq_len >= (1ull << 26) ? 24 : 8;
    const size_t pool_want = ((opts && opts->pool_chunks) ? (size_t)opts->pool_chunks : (size_t)n_slots * pool_per_slot) * POOL_CHUNK_BYTES;
    PlacementSpacer spacer((size_t)n_slots * scratch_slot_bytes(*p, mcl, msp, ix->dev) + pool_want);
    int rc_ = alloc_scratch(m->sc, *p, n_slots, mcl, msp, &bytes, ix->dev);
    if (rc_) return rc_;
    HIPCHK(hipMalloc((void **)&m->d_next, 64));
    {
        // the pool of cluster nodes: 8 chunks (1.5 MB, 6144 nodes) per read in flight on average, 64 on references of 2^26
        // index rows and more, where a read touches thousands of buckets and off-target reads collect hundreds of thousands
        // of clusters; at most 60% of what is left of the HBM.  k_map stops taking up new reads while the pool is nearly
        // empty; a read that still finds it dry is mapped again after the batch (below).
        // That rule sizes the pool for the FIRST batch only: after every batch pool_fit (below) may cut it to four times the
        // high-water mark of chunks out at once (never below one chunk per slot) or double it.
        uint32_t n_chunks = opts ? opts->pool_chunks : 0;
        if (n_chunks == 0) {
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            free_b += spacer.bytes;             // (held only while the pool is allocated)
            const size_t chunk_bytes = POOL_CHUNK_BYTES;
            // (per read in flight: 8 chunks on a bacterial reference, 24 from 2^26 index rows on -- chr20: eleven out at the peak --, 32 from
            // 2^31 on (rounds 2-5 and most of round 6: 64, for every reference past 2^26 rows -- 155 GB for chr20, which left no room for a
            // second mapper beside the first)
            // Round 6, after the ring change: GRCh38 with 12 288 reads in flight has 287 000 chunks out at the peak, the same in every
            // launch (23 per read in flight): 32 per slot = 393 000 chunks = 75 GB = 1.37 x the peak, where 64 (capped at 60 % of the
            // free HBM) had grown to 131 GB once the dense SA and the slots had shrunk.
            const size_t per_slot = pool_per_slot;
            const size_t want = (size_t)n_slots * per_slot;
            n_chunks = (uint32_t)std::max<size_t>(16, std::min<size_t>(want, free_b / 5 * 3 / chunk_bytes));
            m->pool_auto = true;
            m->pool_floor = std::max<uint32_t>(16, std::min<uint32_t>(n_slots, n_chunks));
            m->pool_created = n_chunks;
        }
        int rc2 = alloc_pool(m->pool, n_chunks, &bytes);
        if (rc2) return rc2;
    }
    spacer.release();
    if (n_slots > n_waves) {
        // one pair of rings per XCD when every XCD's share of the slots is a fair number (SchedCtl): slots then never cross XCDs
        uint32_t parts = (opts && opts->sched_parts) ? opts->sched_parts : xcd_count(ix->device);
        if (parts > SCHED_MAX_PARTS || parts == 0 || n_slots % parts || n_slots / parts < 64 || (opts && opts->sched_parts > 1 && opts->sched_parts != xcd_count(ix->device))) {
            if (opts && opts->sched_parts > 1) return fail(UNC_ERR_ARG, "sched_parts: 0, 1 or the device's number of XCDs (%u), dividing n_slots into shares of 64 or more", xcd_count(ix->device));
            parts = 1;
        }
        const uint32_t spp = n_slots / parts;
        uint32_t cap = 64;
        while (cap < spp) cap <<= 1;
        m->sched.cap_mask = cap - 1; m->sched.n_slots = n_slots; m->sched.n_parts = parts;
        HIPCHK(hipMalloc((void **)&m->sched.ctl, sizeof(SchedCtl)));
        HIPCHK(hipMalloc((void **)&m->sched.free_cells, (size_t)parts * cap * sizeof(SchedCell)));
        HIPCHK(hipMalloc((void **)&m->sched.park_cells, (size_t)parts * cap * sizeof(SchedCell)));
        bytes += sizeof(SchedCtl) + 2 * (size_t)parts * cap * sizeof(SchedCell);
    }
    m->device_bytes = bytes;
    {
        int khz = 0;
        HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ix->device));
        m->wall_khz = (float)khz;
    }
    HIPCHK(hipStreamCreate(&m->stream));
    for (auto &e : m->ev) HIPCHK(hipEventCreate(&e));
    { hipEvent_t ref = nullptr; int rcr = ref_event(&ref); if (rcr) return rcr; }      // (before the first batch of any mapper)
    guard.p = nullptr;
    *out = m;
    return UNC_OK;
}

extern "C" uint64_t unc_mapper_device_bytes(const unc_mapper_t *m) { return m->device_bytes; }

static int ensure_batch(unc_mapper *m, uint32_t n_reads, uint64_t total_samples, bool need_raw) {
    if (need_raw && total_samples > m->raw_cap) {
        if (m->d_raw) (void)hipFree(m->d_raw);
        m->d_raw = nullptr;
        HIPCHK(hipMalloc((void **)&m->d_raw, (total_samples + 64) * 2));
        m->raw_cap = total_samples;
    }
    if (n_reads > m->reads_cap) {
        void *ptrs[] = {m->d_offsets, m->d_moff, m->d_calib, m->d_info, m->d_results};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        m->d_offsets = m->d_moff = nullptr; m->d_calib = nullptr; m->d_info = nullptr; m->d_results = nullptr;
        HIPCHK(hipMalloc((void **)&m->d_offsets, ((size_t)n_reads + 1) * 8));
        HIPCHK(hipMalloc((void **)&m->d_moff, ((size_t)n_reads + 1) * 8));
        HIPCHK(hipMalloc((void **)&m->d_calib, (size_t)n_reads * sizeof(unc_calib_t)));
        HIPCHK(hipMalloc((void **)&m->d_info, (size_t)n_reads * sizeof(unc_evt_info_t)));
        HIPCHK(hipMalloc((void **)&m->d_results, (size_t)n_reads * sizeof(DevResult)));
        m->reads_cap = n_reads;
    }
    const uint64_t means_need = total_samples / 8 * 5 + 24ull * n_reads + 16;      // (>= the sum of the reads' rooms, stage_batch)
    if (means_need > m->means_cap) {
        if (m->d_means) (void)hipFree(m->d_means);
        m->d_means = nullptr;
        HIPCHK(hipMalloc((void **)&m->d_means, means_need * 4));
        m->means_cap = means_need;
    }
    return UNC_OK;
}

// uploads metadata, returns the DevReads view
static int stage_batch(unc_mapper *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets, const unc_calib_t *calib,
                       int on_device, hipStream_t st, DevReads *rd) {
    if (n_reads == 0) return fail(UNC_ERR_ARG, "empty batch");
    for (uint32_t i = 0; i < n_reads; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(UNC_ERR_ARG, "offsets must be non-decreasing");
        if (offsets[i + 1] - offsets[i] >= (1ull << 31)) return fail(UNC_ERR_ARG, "read %u too long", i);
    }
    const uint64_t base = offsets[0], total = offsets[n_reads] - base;
    int rc = ensure_batch(m, n_reads, total, !on_device);
    if (rc) return rc;
    m->h_moff.resize((size_t)n_reads + 1);
    // a read's room for its kept event means: 5/8 of its samples + 16.  Neither peak detector can fire on consecutive samples (after a
    // peak it needs one sample to find the next candidate and one to see it fall, event_detector.cpp:221-279) and the short one masks the
    // long one for three samples, so n / 2 bounds the events; k_events checks the room and reports a read that overran it
    // (unc_evt_info_t.pad), which fails the call below.  (Rounds 1-4: n + 16 floats per read, 32 GB for 250 k reads.)
    {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < n_reads; ++i) { m->h_moff[i] = acc; acc += (offsets[i + 1] - offsets[i]) / 8 * 5 + 16; }
        m->h_moff[n_reads] = acc;
    }
    HIPCHK(hipMemcpyAsync(m->d_offsets, offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->d_moff, m->h_moff.data(), ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->d_calib, calib, (size_t)n_reads * sizeof(unc_calib_t), hipMemcpyHostToDevice, st));
    const int16_t *d_raw = raw;
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(m->d_raw, raw + base, total * 2, hipMemcpyHostToDevice, st));
        d_raw = m->d_raw - base;   // kernels index raw[offsets[i]..]
    }
    rd->raw = d_raw; rd->offsets = m->d_offsets; rd->calib = m->d_calib; rd->means = m->d_means; rd->moff = m->d_moff;
    rd->info = m->d_info; rd->n_reads = n_reads;
    rd->tgt_mean = m->ix->model_mean; rd->tgt_stdv = m->ix->model_stdv;
    rd->ring0 = nullptr; rd->new_read = nullptr; rd->ring_mod = 0;
    return UNC_OK;
}

// Mapper::event_to_bp, mapper.cpp:703-706.  float -> u32 of an out-of-range value (evt_st - seed_len can
// wrap, :716) follows x86-64's cvttss2si r64 + truncation, as the reference binary does.
static uint32_t event_to_bp(const unc_params_t &P, uint32_t evt_i, float mean_event_len, bool last) {
    float bp_per_samp = P.bp_per_sec / P.sample_rate;
    float v = (float)evt_i * mean_event_len;
    v = v * bp_per_samp;
    v = v + (float)((int)last * (UNC_KLEN - 1));
    return (uint32_t)(int64_t)v;
}

// Mapper::set_ref_loc (mapper.cpp:708-728) + Paf::set_mapped / set_read_len (read_buffer.cpp:133-155,264-267)
static void fill_hit(const unc_index *ix, const unc_params_t &P, const DevResult &res, const unc_evt_info_t &inf, uint64_t raw_len,
                     unc_hit_t *h, float ticks_per_ms = 0.0f) {
    memset(h, 0, sizeof *h);
    h->map_ms = ticks_per_ms > 0.0f ? (float)((double)res.ticks / (double)ticks_per_ms) : 0.0f;
    h->rid = -1;
    h->status = res.status;
    h->notes = res.notes; h->pad_ = 0;
    h->n_events = inf.n_events;
    h->event_i = res.event_i;
    float mel = inf.len_sum / (float)inf.total_events;   // EventDetector::mean_event_len, event_detector.cpp:151-153
    h->mean_event_len = inf.total_events ? mel : 0.0f;
    h->n_nbr = res.n_nbr; h->n_sa = res.n_sa; h->n_lf = res.n_lf;
    if (res.done == 1 && res.status == 0) {
        const ClusterVal &c = res.cluster;
        const uint64_t size = ix->seq_len;
        const bool fwd = c.ref_st < size / 2;
        const uint64_t sa_st = fwd ? c.ref_st : size - (c.rend + UNC_KLEN - 1);
        h->rd_st = event_to_bp(P, c.evt_st - P.seed_len, mel, false);
        h->rd_en = event_to_bp(P, c.evt_en, mel, true);
        h->rd_len = event_to_bp(P, res.event_i, mel, true);
        uint64_t rf_st = 0;
        int32_t rid;
        const uint64_t rf_len = unc_index_translate_loc(ix, sa_st, &rid, &rf_st);
        h->rid = rid;
        h->rf_st = rf_st;
        h->rf_len = rf_len;
        h->rf_en = rf_st + (c.rend - c.ref_st + UNC_KLEN);
        h->matches = (uint16_t)(c.total_len + UNC_KLEN - 1);
        h->mapped = 1;
        h->fwd = fwd ? 1 : 0;
        h->cl_ref_st = c.ref_st; h->cl_ref_en_start = c.rstart; h->cl_ref_en_end = c.rend;
        h->cl_evt_st = c.evt_st; h->cl_evt_en = c.evt_en; h->cl_total_len = c.total_len;
    } else {
        float bp_per_samp = P.bp_per_sec / P.sample_rate;
        h->rd_len = (uint64_t)((float)raw_len * bp_per_samp);
    }
}

// chunks that were out at once in the launches since the pool was last initialised (PoolQueue::low_water)
static int pool_note_high_water(unc_mapper *m, hipStream_t st) {
    uint32_t lw = m->pool.n_chunks;
    HIPCHK(hipMemcpyAsync(&lw, &m->pool.q->low_water, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint32_t hw = m->pool.n_chunks - std::min(lw, m->pool.n_chunks);
    m->pool_hw_last = std::max(m->pool_hw_last, hw);
    m->pool_hw_max = std::max(m->pool_hw_max, hw);
    return UNC_OK;
}

// After a batch (the pool is idle): a pool that went dry doubles (the reads concerned were mapped again, correctly but late); one
// that holds more than eight times the most chunks that were ever out at once shrinks to four times that.
static int pool_fit(unc_mapper *m, bool went_dry, uint32_t n_reads) {
    if (!m->pool_auto) return UNC_OK;
    const uint64_t cur = m->pool.n_chunks;
    uint64_t target = cur;
    if (went_dry) {
        // found dry: not one doubling per batch (a pool cut after a small first call took several full batches to recover, each with
        // reads mapped again one by one: round-5 advice) but straight back to what is known to have been enough -- the size it was
        // created with -- or to four times the most chunks that were out when it ran dry, whichever is more
        target = std::max<uint64_t>(std::max<uint64_t>(cur * 2, m->pool_created), 4ull * m->pool_hw_max);
    } else if (n_reads >= m->n_slots) {
        // cut only after a batch that FILLED the slots (a warm-up call, a handful of reads or a profiling pass over part of a batch say
        // nothing about what a full load of reads in flight takes): to four times the most chunks ever out at once, and only when the
        // pool is more than twice that -- the peak of ONE batch moves by tens of per cent from launch to launch (which reads are deep
        // in their forests at the same time is a matter of scheduling), and a pool cut to twice one launch's peak was found dry by a
        // later launch of the same batch (15 reads mapped again)
        const uint64_t need = std::max<uint64_t>(m->pool_floor, 4ull * m->pool_hw_max);
        if (need * 2 < cur) target = need;
    }
    if (target > cur) {     // growing: at most 60 % of what would be free without the pool
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const uint64_t most = (free_b + cur * (uint64_t)POOL_CHUNK_BYTES) / 5 * 3 / POOL_CHUNK_BYTES;
        target = std::max(cur, std::min(target, most));
    }
    if (target == cur) return UNC_OK;
    HIPCHK(hipDeviceSynchronize());
    // the batch's hits are already filled: a resize that fails must neither fail the call nor leave the mapper without a pool.  A
    // smaller pool is allocated BEFORE the old one is freed (both fit); a larger one may need the old one's memory, and if it cannot
    // be had the old size is allocated again.
    DevPool fresh;
    size_t bytes = 0;
    if (target < cur) {
        if (alloc_pool(fresh, (uint32_t)target, &bytes) != UNC_OK) { free_pool(fresh); (void)hipGetLastError(); return UNC_OK; }
        free_pool(m->pool);
    } else {
        free_pool(m->pool);
        if (alloc_pool(fresh, (uint32_t)target, &bytes) != UNC_OK) {
            free_pool(fresh); (void)hipGetLastError();
            target = cur;
            int rc = alloc_pool(fresh, (uint32_t)cur, &bytes);
            if (rc) { free_pool(fresh); memset(&m->pool, 0, sizeof m->pool); return rc; }      // (the memory was there a moment ago)
        }
    }
    m->pool = fresh;
    m->device_bytes += (uint64_t)target * POOL_CHUNK_BYTES;
    m->device_bytes -= (uint64_t)cur * POOL_CHUNK_BYTES;
    if (target != cur) m->pool_resizes++;
    return UNC_OK;
}

extern "C" int unc_mapper_pool_usage(const unc_mapper_t *m, uint32_t *out4) {
    if (!m || !out4) return fail(UNC_ERR_ARG, "null argument");
    out4[0] = m->pool.n_chunks; out4[1] = m->pool_hw_last; out4[2] = m->pool_hw_max; out4[3] = m->pool_resizes;
    return UNC_OK;
}

// A batch in two halves: _begin stages the reads and launches the kernels on the stream and returns; _end waits for them, maps the few
// reads again that need it, and fills the hits.  Between the two the host is free -- and so is the GPU's tail: the persistent k_map
// ends with a few long reads on a few wavefronts (wavefronts alive 94 % of a 50 k-read E. coli launch), and a second mapper's batch,
// begun on its own stream meanwhile, moves into the compute units as they fall idle.  unc_map_batch = the two in a row.
extern "C" int unc_map_batch_begin(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                                   const unc_calib_t *calib, int on_device, void *stream) {
    if (!m || !raw || !offsets || !calib) return fail(UNC_ERR_ARG, "null argument");
    if (m->pend.active) return fail(UNC_ERR_ARG, "unc_map_batch_begin: the mapper's previous batch has not been collected (unc_map_batch_end)");
    HIPCHK(hipSetDevice(m->ix->device));
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    DevReads rd;
    int rc = stage_batch(m, n_reads, raw, offsets, calib, on_device, st, &rd);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(m->d_next, 0, 16, st));   // [0] queue head, [2..3] wave-lifetime ticks
    HIPCHK(hipEventRecord(m->ev[0], st));
    launch_events(rd, m->P, st, m->ev_rpw);
    HIPCHK(hipEventRecord(m->ev[1], st));
    const uint32_t grid = n_reads < m->n_waves ? n_reads : m->n_waves;
    const bool sliced = m->sched.ctl != nullptr && n_reads > m->n_waves;
    if (sliced) launch_sched_init(m->sched, st);
    launch_pool_init(m->pool, st);       // every chunk free: nothing outlives a batch
    constexpr size_t FW = NKMER / 32;      // words of one read's sources_added_ bitmap
    const bool t1 = m->read_order == UNC_ORDER_T1;
    if (t1) {
        if (n_reads > m->flags_cap) {
            if (m->d_flags_in) (void)hipFree(m->d_flags_in);
            if (m->d_flags_out) (void)hipFree(m->d_flags_out);
            m->d_flags_in = m->d_flags_out = nullptr; m->flags_cap = 0;
            HIPCHK(hipMalloc((void **)&m->d_flags_in, (size_t)n_reads * FW * 4));
            HIPCHK(hipMalloc((void **)&m->d_flags_out, (size_t)n_reads * FW * 4));
            m->flags_cap = n_reads;
        }
        HIPCHK(hipMemsetAsync(m->d_flags_in, 0, (size_t)n_reads * FW * 4, st));
        HIPCHK(hipMemcpyAsync(m->d_flags_in, m->carry_flags, FW * 4, hipMemcpyHostToDevice, st));    // read 0 follows the previous batch's last read
    }
    const uint32_t *const fl_in = t1 ? m->d_flags_in : nullptr;
    uint32_t *const fl_out = t1 ? m->d_flags_out : nullptr;
    launch_map(m->ix->dev, m->sc, rd, m->P, m->d_results, m->d_next, sliced ? m->slice_events : 0xFFFFFFFFu, 0, nullptr, grid, st, m->pool,
               nullptr, reinterpret_cast<unsigned long long *>(m->d_next + 2), sliced ? &m->sched : nullptr, m->profile, fl_in, fl_out);
    HIPCHK(hipEventRecord(m->ev[2], st));
    HIPCHK(hipGetLastError());
    m->pend.active = true; m->pend.n_reads = n_reads; m->pend.grid = grid; m->pend.st = st; m->pend.rd = rd; m->pend.t1 = t1;
    m->pend.lens.resize(n_reads);
    for (uint32_t i = 0; i < n_reads; ++i) m->pend.lens[i] = offsets[i + 1] - offsets[i];
    return UNC_OK;
}

extern "C" int unc_map_batch_end(unc_mapper_t *m, unc_hit_t *hits) {
    if (!m || !hits) return fail(UNC_ERR_ARG, "null argument");
    if (!m->pend.active) return fail(UNC_ERR_ARG, "unc_map_batch_end: no batch has been begun on this mapper");
    m->pend.active = false;       // (whatever happens below, the batch is over)
    HIPCHK(hipSetDevice(m->ix->device));
    const uint32_t n_reads = m->pend.n_reads, grid = m->pend.grid;
    hipStream_t st = m->pend.st;
    const DevReads rd = m->pend.rd;
    const bool t1 = m->pend.t1;
    constexpr size_t FW = NKMER / 32;
    const uint32_t *const fl_in = t1 ? m->d_flags_in : nullptr;
    uint32_t *const fl_out = t1 ? m->d_flags_out : nullptr;
    int rc = UNC_OK;
    m->h_info.resize(n_reads);
    m->h_results.resize(n_reads);
    // (the copies into pageable host memory would hold the host until the kernels are done: they are issued here, not in _begin)
    HIPCHK(hipMemcpyAsync(m->h_info.data(), m->d_info, (size_t)n_reads * sizeof(unc_evt_info_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(m->h_results.data(), m->d_results, (size_t)n_reads * sizeof(DevResult), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&m->ms_events, m->ev[0], m->ev[1]));
    HIPCHK(hipEventElapsedTime(&m->ms_map, m->ev[1], m->ev[2]));
    {
        hipEvent_t ref = nullptr;
        float a = 0, b = 0;
        if (ref_event(&ref) == UNC_OK && hipEventElapsedTime(&a, ref, m->ev[1]) == hipSuccess && hipEventElapsedTime(&b, ref, m->ev[2]) == hipSuccess) {
            m->win_start_ms = a; m->win_end_ms = b;
        } else (void)hipGetLastError();
    }
    for (uint32_t i = 0; i < n_reads; ++i)
        if (m->h_info[i].pad) return fail(UNC_ERR_OVERFLOW, "read %u: more events than the room for event means holds (5/8 of its samples + 16)", i);
    m->pool_hw_last = 0;
    rc = pool_note_high_water(m, st);
    if (rc) return rc;
    bool pool_went_dry = false;
    {
        unsigned long long ticks = 0;
        int khz = 0;
        HIPCHK(hipMemcpy(&ticks, m->d_next + 2, 8, hipMemcpyDeviceToHost));
        HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m->ix->device));
        m->wave_busy = (grid && m->ms_map > 0 && khz > 0) ? (double)ticks / ((double)grid * (double)m->ms_map * (double)khz) : 0.0;
    }
    // The reference's SeedTracker is an unbounded std::set.  Reads whose set did not fit are mapped again after the batch, by
    // cause: a read that found the pool of nodes DRY runs again on the same scratch with fewer and fewer reads sharing the
    // pool (n_waves, a quarter of that, ... down to one read with the whole pool); a read that used up its own ALLOWANCE
    // (max_clusters / 4 nodes) runs again on scratch with a 16x larger allowance, up to 2^26 clusters.
    m->remap_reads = 0;
    m->remap_ms = 0;
    // (`subset`: the reads whose results are new -- all of the batch after the first pass, the re-mapped ones of a `-t 1` round)
    auto resolve_overflows = [&](const std::vector<uint32_t> *subset) -> int {
        std::vector<uint32_t> dry, full, work;
        auto classify = [&](uint32_t i) {
            const uint32_t stt = m->h_results[i].status;
            if (stt & UNC_READ_POOL_DRY) dry.push_back(i);
            else if (stt & UNC_READ_CLUSTER_OVERFLOW) full.push_back(i);
        };
        if (subset) { for (uint32_t i : *subset) classify(i); }
        else { for (uint32_t i = 0; i < n_reads; ++i) classify(i); }
        m->remap_reads += (uint32_t)(dry.size() + full.size());
        if (!dry.empty()) pool_went_dry = true;
        const auto t_redo = std::chrono::steady_clock::now();
        uint64_t cap = m->sc.max_clusters;
        size_t limit = m->n_waves;
        // one pass over `work` with at most `slots` reads in flight; the reads that overflowed again are sorted by cause
        auto run_round = [&](const DevScratch &sc, size_t slots) -> int {
            if (work.size() > m->list_cap) {
                if (m->d_list) (void)hipFree(m->d_list);
                m->d_list = nullptr; m->list_cap = 0;
                HIPCHK(hipMalloc((void **)&m->d_list, work.size() * 4));
                m->list_cap = work.size();
            }
            HIPCHK(hipMemcpyAsync(m->d_list, work.data(), work.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(m->d_next, 0, 4, st));
            launch_pool_init(m->pool, st);
            DevReads rd2 = rd;
            rd2.n_reads = (uint32_t)work.size();
            launch_map(m->ix->dev, sc, rd2, m->P, m->d_results, m->d_next, 0xFFFFFFFFu, 0, nullptr, (uint32_t)slots, st, m->pool, m->d_list,
                       nullptr, nullptr, false, fl_in, fl_out);
            HIPCHK(hipGetLastError());
            const uint32_t lo = *std::min_element(work.begin(), work.end()), hi = *std::max_element(work.begin(), work.end());
            HIPCHK(hipMemcpyAsync(m->h_results.data() + lo, m->d_results + lo, (size_t)(hi - lo + 1) * sizeof(DevResult), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (int rc3 = pool_note_high_water(m, st)) return rc3;
            for (uint32_t i : work) classify(i);
            work.clear();
            return UNC_OK;
        };
        auto scratch_now = [&]() -> const DevScratch & { return cap == m->sc.max_clusters ? m->sc : m->big; };
        while (!dry.empty() || !full.empty()) {
            if (!dry.empty()) {
                const DevScratch &sc = scratch_now();
                const size_t have = cap == m->sc.max_clusters ? m->n_slots : m->big_slots;
                const size_t slots = std::min({dry.size(), have, limit});
                work.swap(dry);
                // (reads that ran out of allowance wait in `full` meanwhile: at this allowance they would only overflow again)
                int rc2 = run_round(sc, slots);
                if (rc2) return rc2;
                if (!dry.empty()) {
                    if (slots <= 1) {
                        // every read of this round had the whole pool to itself and is still dry: reported with its status (raise
                        // pool_chunks); the reads waiting in `full` for a larger allowance go on (round-3 advice: they were abandoned)
                        dry.clear();
                        continue;
                    }
                    limit = std::max<size_t>(1, slots / 4);
                }
                continue;
            }
            cap *= 16;
            if (cap > (1ull << 26)) break;
            const size_t want = std::min<size_t>(std::max<size_t>(full.size(), 64), m->n_waves);
            if (m->big_cap != cap || (m->big_slots < want && !m->big_at_limit)) {
                free_scratch(m->big);
                m->big_cap = 0; m->big_slots = 0;
                size_t free_b = 0, total_b = 0;
                HIPCHK(hipMemGetInfo(&free_b, &total_b));
                const size_t per_slot = scratch_slot_bytes(m->P, (uint32_t)cap, m->sc.max_seed_paths, m->ix->dev);
                const size_t fit = std::max<size_t>(1, free_b / 4 / per_slot);
                const size_t n = std::min(want, fit);
                int rc2 = alloc_scratch(m->big, m->P, n, (uint32_t)cap, m->sc.max_seed_paths, nullptr, m->ix->dev);
                if (rc2) { free_scratch(m->big); return rc2; }
                m->big_cap = cap; m->big_slots = n; m->big_at_limit = n == fit;
            }
            const size_t slots = std::min({full.size(), m->big_slots, limit});
            work.swap(full);
            int rc2 = run_round(m->big, slots);
            if (rc2) return rc2;
        }
        m->remap_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_redo).count();
        return UNC_OK;
    };
    rc = resolve_overflows(nullptr);
    if (rc) return rc;
    m->carry_reads = 0; m->carry_rounds = 0; m->carry_ms = 0;
    if (t1) {
        // `uncalled map -t 1`: ONE Mapper maps the reads one after another, and sources_added_ is the one piece of its state that
        // Mapper::new_read does not reset (mapper.cpp:88,219-246): what a read leaves set (only a read whose path buffer was full
        // can, :612-623; UNC_NOTE_FLAGS_LEFT) is what its successor starts with.  The batch was mapped with every read but the
        // first starting clear; a read whose predecessor turned out to leave something else than it assumed is mapped again
        // with that, and so on down the chain until every read has started from what its predecessor really left.
        const auto t_c = std::chrono::steady_clock::now();
        typedef std::array<uint32_t, FW> Flags;
        const Flags zero{};
        std::map<uint32_t, Flags> left;        // read -> flags its CURRENT result leaves set (absent: none)
        std::map<uint32_t, Flags> assumed;     // read -> flags its current result started from (absent: none)
        Flags f0; memcpy(f0.data(), m->carry_flags, sizeof f0);
        if (f0 != zero) assumed[0] = f0;
        // (a read that is reported with an overflow status has no valid run behind it: it hands NO flags on -- round-4 advice; the
        // rows of all flagged reads are fetched together, one synchronisation per call)
        auto fetch_left = [&](const std::vector<uint32_t> &reads) -> int {
            std::vector<uint32_t> rows;
            for (uint32_t i : reads) {
                left.erase(i);
                if (m->h_results[i].status == 0 && (m->h_results[i].notes & UNC_NOTE_FLAGS_LEFT)) rows.push_back(i);
            }
            std::vector<Flags> got(rows.size());
            for (size_t k = 0; k < rows.size(); ++k)
                HIPCHK(hipMemcpyAsync(got[k].data(), m->d_flags_out + (size_t)rows[k] * FW, sizeof(Flags), hipMemcpyDeviceToHost, st));
            if (!rows.empty()) HIPCHK(hipStreamSynchronize(st));
            for (size_t k = 0; k < rows.size(); ++k) left[rows[k]] = got[k];
            return UNC_OK;
        };
        {
            std::vector<uint32_t> flagged;
            for (uint32_t i = 0; i < n_reads; ++i) if (m->h_results[i].status == 0 && (m->h_results[i].notes & UNC_NOTE_FLAGS_LEFT)) flagged.push_back(i);
            rc = fetch_left(flagged);
            if (rc) return rc;
        }
        auto get = [&](const std::map<uint32_t, Flags> &mp, uint32_t i) -> const Flags & { auto it = mp.find(i); return it == mp.end() ? zero : it->second; };
        std::vector<uint32_t> todo;
        for (const auto &kv : left) if (kv.first + 1 < n_reads) todo.push_back(kv.first + 1);
        while (!todo.empty()) {
            std::vector<uint32_t> work;
            for (uint32_t j : todo) if (get(left, j - 1) != get(assumed, j)) work.push_back(j);
            if (work.empty()) break;
            for (uint32_t j : work) {
                const Flags &f = get(left, j - 1);
                if (f == zero) assumed.erase(j); else assumed[j] = f;
                HIPCHK(hipMemcpyAsync(m->d_flags_in + (size_t)j * FW, f.data(), sizeof f, hipMemcpyHostToDevice, st));
            }
            if (work.size() > m->list_cap) {
                if (m->d_list) (void)hipFree(m->d_list);
                m->d_list = nullptr; m->list_cap = 0;
                HIPCHK(hipMalloc((void **)&m->d_list, work.size() * 4));
                m->list_cap = work.size();
            }
            HIPCHK(hipMemcpyAsync(m->d_list, work.data(), work.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(m->d_next, 0, 4, st));
            launch_pool_init(m->pool, st);
            DevReads rd2 = rd;
            rd2.n_reads = (uint32_t)work.size();
            launch_map(m->ix->dev, m->sc, rd2, m->P, m->d_results, m->d_next, 0xFFFFFFFFu, 0, nullptr,
                       (uint32_t)std::min<size_t>(work.size(), m->n_slots), st, m->pool, m->d_list, nullptr, nullptr, false, fl_in, fl_out);
            HIPCHK(hipGetLastError());
            for (uint32_t j : work)
                HIPCHK(hipMemcpyAsync(m->h_results.data() + j, m->d_results + j, sizeof(DevResult), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            rc = resolve_overflows(&work);
            if (rc) return rc;
            rc = fetch_left(work);
            if (rc) return rc;
            m->carry_reads += (uint32_t)work.size();
            m->carry_rounds++;
            todo.clear();
            for (uint32_t j : work) if (j + 1 < n_reads) todo.push_back(j + 1);
        }
        const Flags &last = get(left, n_reads - 1);
        memcpy(m->carry_flags, last.data(), sizeof m->carry_flags);
        m->carry_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_c).count();
    }
    int worst = UNC_OK;
    for (uint32_t i = 0; i < n_reads; ++i) {
        fill_hit(m->ix, m->P, m->h_results[i], m->h_info[i], m->pend.lens[i], &hits[i], m->wall_khz);
        if (hits[i].status) worst = UNC_ERR_OVERFLOW;
    }
    rc = pool_fit(m, pool_went_dry, n_reads);
    if (rc) return rc;
    if (worst) return fail(worst, "device scratch overflow on at least one read (see unc_hit_t.status); raise pool_chunks / max_clusters / max_seed_paths");
    return UNC_OK;
}

extern "C" int unc_map_batch(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                             const unc_calib_t *calib, int on_device, void *stream, unc_hit_t *hits) {
    if (!hits) return fail(UNC_ERR_ARG, "null argument");
    const int rc = unc_map_batch_begin(m, n_reads, raw, offsets, calib, on_device, stream);
    if (rc) return rc;
    return unc_map_batch_end(m, hits);
}

extern "C" int unc_mapper_last_phase_cycles(const unc_mapper_t *m, uint64_t *out8) {
    for (int i = 0; i < 12; ++i) out8[i] = 0;
    for (const DevResult &r : m->h_results)
        for (int i = 0; i < 12; ++i) out8[i] += r.cyc[i];
    return UNC_OK;
}

// per read of the last batch (profiling instantiation): the twelve phase counters, the residence in wall-clock ticks, and where the read was
// decided (XCC_ID | HW_ID << 8) -- 14 words of 64 bits per read
extern "C" int unc_mapper_last_read_cycles(const unc_mapper_t *m, uint32_t n_reads, uint64_t *out14) {
    if (!m || !out14) return fail(UNC_ERR_ARG, "null argument");
    if (n_reads > m->h_results.size()) return fail(UNC_ERR_ARG, "unc_mapper_last_read_cycles: the last batch had %zu reads", m->h_results.size());
    for (uint32_t r = 0; r < n_reads; ++r) {
        const DevResult &d = m->h_results[r];
        for (int i = 0; i < 12; ++i) out14[(size_t)r * 14 + i] = d.cyc[i];
        out14[(size_t)r * 14 + 12] = d.ticks;
        out14[(size_t)r * 14 + 13] = d.pad2;
    }
    return UNC_OK;
}

// where the mapper's large allocations lie (diagnostics: the same library maps the same batch 4 - 8 % faster or slower from one mapper
// instance to the next, tools/dev/placement_probe.py): [0] slots' base, [1] bytes per slot, [2] node pool's base, [3] its bytes,
// [4] event means, [5] results
extern "C" int unc_mapper_device_addresses(const unc_mapper_t *m, uint64_t *out6) {
    if (!m || !out6) return fail(UNC_ERR_ARG, "null argument");
    out6[0] = (uint64_t)(uintptr_t)m->sc.base; out6[1] = m->sc.slot_bytes;
    out6[2] = (uint64_t)(uintptr_t)m->pool.nodes; out6[3] = (uint64_t)m->pool.n_chunks * POOL_CHUNK_BYTES;
    out6[4] = (uint64_t)(uintptr_t)m->d_means; out6[5] = (uint64_t)(uintptr_t)m->d_results;
    return UNC_OK;
}
extern "C" uint32_t unc_mapper_sched_parts(const unc_mapper_t *m) { return (m && m->sched.ctl) ? m->sched.n_parts : 0u; }
extern "C" double unc_mapper_last_wave_busy(const unc_mapper_t *m) { return m ? m->wave_busy : 0.0; }
extern "C" int unc_mapper_set_read_order(unc_mapper_t *m, int order) {
    if (!m || (order != UNC_ORDER_INDEPENDENT && order != UNC_ORDER_T1)) return fail(UNC_ERR_ARG, "unc_mapper_set_read_order: UNC_ORDER_INDEPENDENT or UNC_ORDER_T1");
    m->read_order = order;
    memset(m->carry_flags, 0, sizeof m->carry_flags);      // a new run: the Mapper's flags start clear (mapper.cpp:88)
    return UNC_OK;
}
extern "C" int unc_mapper_last_carry_over(const unc_mapper_t *m, uint32_t *reads, uint32_t *rounds, float *ms) {
    if (!m) return fail(UNC_ERR_ARG, "null argument");
    if (reads) *reads = m->carry_reads;
    if (rounds) *rounds = m->carry_rounds;
    if (ms) *ms = m->carry_ms;
    return UNC_OK;
}
extern "C" void unc_mapper_set_profile(unc_mapper_t *m, int on) { if (m) m->profile = on != 0; }
extern "C" int unc_mapper_kernel_info(const unc_mapper_t *m, uint32_t *out6) {
    if (!m || !out6) return fail(UNC_ERR_ARG, "null argument");
    const bool narrow = m->ix->dev.key_len_bits != 0;
    if (map_kernel_attributes(narrow, false, out6)) return fail(UNC_ERR_HIP, "hipFuncGetAttributes failed");
    out6[4] = map_kernel_waves_per_cu(); out6[5] = narrow ? 1u : 0u;
    return UNC_OK;
}
extern "C" void unc_mapper_geometry(const unc_mapper_t *m, uint32_t *out5) {
    out5[0] = m->n_waves; out5[1] = m->n_slots; out5[2] = m->sched.ctl ? m->slice_events : 0u;
    out5[3] = m->pool.n_chunks; out5[4] = m->sc.max_clusters;
}
extern "C" void unc_mapper_last_remap(const unc_mapper_t *m, uint32_t *n_reads, float *ms) {
    if (n_reads) *n_reads = m ? m->remap_reads : 0;
    if (ms) *ms = m ? m->remap_ms : 0.0f;
}

extern "C" int unc_mapper_last_timing(const unc_mapper_t *m, float *ms_events, float *ms_map) {
    if (ms_events) *ms_events = m->ms_events;
    if (ms_map) *ms_map = m->ms_map;
    return UNC_OK;
}

extern "C" int unc_mapper_last_window(const unc_mapper_t *m, double *start_ms, double *end_ms) {
    if (!m || !start_ms || !end_ms) return fail(UNC_ERR_ARG, "null argument");
    *start_ms = m->win_start_ms; *end_ms = m->win_end_ms;
    return UNC_OK;
}

extern "C" int unc_detect_events(unc_mapper_t *m, uint32_t n_reads, const int16_t *raw, const uint64_t *offsets,
                                 const unc_calib_t *calib, float *means, uint64_t means_cap, uint64_t *means_offsets,
                                 unc_evt_info_t *info) {
    if (!m || !raw || !offsets || !calib || !means || !means_offsets || !info) return fail(UNC_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(m->ix->device));
    hipStream_t st = m->stream;
    DevReads rd;
    int rc = stage_batch(m, n_reads, raw, offsets, calib, 0, st, &rd);
    if (rc) return rc;
    HIPCHK(hipEventRecord(m->ev[0], st));
    launch_events(rd, m->P, st, m->ev_rpw);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->ev[1], st));
    HIPCHK(hipMemcpyAsync(info, m->d_info, (size_t)n_reads * sizeof(unc_evt_info_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&m->ms_events, m->ev[0], m->ev[1]));     // unc_mapper_last_timing: k_events alone
    m->ms_map = 0;
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n_reads; ++i)
        if (info[i].pad) return fail(UNC_ERR_OVERFLOW, "read %u: more events than the room for event means holds (5/8 of its samples + 16)", i);
    for (uint32_t i = 0; i < n_reads; ++i) { means_offsets[i] = tot; tot += info[i].n_events; }
    means_offsets[n_reads] = tot;
    if (tot > means_cap) return fail(UNC_ERR_ARG, "means buffer too small: need %llu", (unsigned long long)tot);
    for (uint32_t i = 0; i < n_reads; ++i)
        HIPCHK(hipMemcpyAsync(means + means_offsets[i], m->d_means + m->h_moff[i], (size_t)info[i].n_events * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return UNC_OK;
}

// ------------------------------------------------------------------ radix sort of the index builders (k_sort.hip)
extern "C" int unc_sort_pairs_u64(int device, uint64_t n, uint64_t *keys, uint64_t *vals, uint64_t *tmp_keys, uint64_t *tmp_vals,
                                  int key_bits, int iota, void *stream) {
    if (!keys || !vals || !tmp_keys || !tmp_vals) return fail(UNC_ERR_ARG, "null argument");
    if (key_bits < 1 || key_bits > 64) return fail(UNC_ERR_ARG, "key_bits must be in 1..64");
    if (n >= (1ull << 32)) return fail(UNC_ERR_ARG, "at most 2^32 - 1 pairs");
    if (n == 0) return UNC_OK;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const uint64_t ntiles = (n + rsort_tile() - 1) / rsort_tile(), m = 256 * ntiles, nsums = (m + rsort_tile() - 1) / rsort_tile();
    uint32_t *counts = nullptr, *sums = nullptr;
    HIPCHK(hipMalloc((void **)&counts, m * 4));
    if (hipMalloc((void **)&sums, nsums * 4) != hipSuccess) { (void)hipFree(counts); return fail(UNC_ERR_HIP, "hipMalloc failed"); }
    uint64_t *ki = keys, *vi = vals, *ko = tmp_keys, *vo = tmp_vals;
    const int passes = (key_bits + 7) / 8;
    for (int p = 0; p < passes; ++p) {
        launch_rsort_pass(ki, vi, ko, vo, n, (uint32_t)(8 * p), (iota && p == 0) ? 1u : 0u, counts, sums, st);
        std::swap(ki, ko); std::swap(vi, vo);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && ki != keys) {      // an odd number of passes: the result sits in the scratch arrays
        e = hipMemcpyAsync(keys, ki, n * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(vals, vi, n * 8, hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(counts); (void)hipFree(sums);
    if (e != hipSuccess) return fail(UNC_ERR_HIP, "radix sort: %s", hipGetErrorString(e));
    return UNC_OK;
}

// The suffix array of a text of n < 2^31 symbols (codes 0..3, host memory) -> sa[0 .. n) (host memory, int64): prefix doubling on the
// device -- 21 symbols first, then 42, 84, ... -- with the radix sort above and the steps between the sorts as kernels of their own
// (k_sort.hip).  What bwa_idx_build's suffix sort does for `uncalled index` (bwa_index.hpp:92-101); uncalled_amd/build_index.py builds
// the five index files around it.  Device memory: 37 bytes per symbol.
extern "C" int unc_build_suffix_array(int device, const uint8_t *codes, uint64_t n, int64_t *sa) {
    if (!codes || !sa) return fail(UNC_ERR_ARG, "null argument");
    if (n == 0) return UNC_OK;
    if (n >= (1ull << 31)) return fail(UNC_ERR_ARG, "unc_build_suffix_array: texts of fewer than 2^31 symbols (the chunked builder of "
                                                    "uncalled_amd/build_index_big.py takes the larger ones)");
    HIPCHK(hipSetDevice(device));
    hipStream_t st = nullptr;
    struct Bufs {
        uint8_t *text = nullptr; uint64_t *k[2] = {nullptr, nullptr}, *v[2] = {nullptr, nullptr};
        uint32_t *rank = nullptr, *flags = nullptr, *sums = nullptr, *counts = nullptr, *ngroups = nullptr;
        ~Bufs() { void *p[] = {text, k[0], k[1], v[0], v[1], rank, flags, sums, counts, ngroups}; for (void *x : p) if (x) (void)hipFree(x); }
    } B;
    const uint64_t ntiles = (n + rsort_tile() - 1) / rsort_tile(), m = 256 * ntiles;
    const uint64_t nsums = (std::max<uint64_t>(m, n) + rsort_tile() - 1) / rsort_tile();
    HIPCHK(hipMalloc((void **)&B.text, n));
    for (int i = 0; i < 2; ++i) { HIPCHK(hipMalloc((void **)&B.k[i], n * 8)); HIPCHK(hipMalloc((void **)&B.v[i], n * 8)); }
    HIPCHK(hipMalloc((void **)&B.rank, n * 4));
    HIPCHK(hipMalloc((void **)&B.flags, n * 4));
    HIPCHK(hipMalloc((void **)&B.sums, nsums * 4));
    HIPCHK(hipMalloc((void **)&B.counts, m * 4));
    HIPCHK(hipMalloc((void **)&B.ngroups, 4));
    HIPCHK(hipMemcpyAsync(B.text, codes, n, hipMemcpyHostToDevice, st));
    // keys in B.k[0], sorted with the suffix numbers as values (iota); returns which pair of buffers holds the result
    auto sort_keys = [&](int key_bits) -> int {
        int cur = 0;
        const int passes = (key_bits + 7) / 8;
        for (int p = 0; p < passes; ++p) {
            launch_rsort_pass(B.k[cur], B.v[cur], B.k[cur ^ 1], B.v[cur ^ 1], n, (uint32_t)(8 * p), p == 0 ? 1u : 0u, B.counts, B.sums, st);
            cur ^= 1;
        }
        return cur;
    };
    auto rerank = [&](int key_bits, uint32_t *groups) -> int {
        const int cur = sort_keys(key_bits);
        launch_sa_ranks(B.k[cur], B.v[cur], n, B.flags, B.sums, B.rank, B.ngroups, st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(groups, B.ngroups, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return UNC_OK;
    };
    uint32_t groups = 0;
    launch_sa_first_key(B.text, n, B.k[0], st);
    int rc = rerank(63, &groups);
    if (rc) return rc;
    int bits = 1;
    while ((((unsigned __int128)(n + 1) * (n + 1)) >> bits) != 0) ++bits;      // bits of the largest doubled key
    for (uint64_t k = 21; groups < n; k *= 2) {
        launch_sa_next_key(B.rank, n, k, B.k[0], st);
        rc = rerank(bits, &groups);
        if (rc) return rc;
        if (k > n) return fail(UNC_ERR_HIP, "unc_build_suffix_array: %u groups of %llu suffixes after comparing whole suffixes", groups, (unsigned long long)n);
    }
    launch_sa_invert(B.rank, n, reinterpret_cast<int64_t *>(B.k[0]), st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sa, B.k[0], n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return UNC_OK;
}

// Known-byte traffic in k_map's access shape, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/dev/pmc_calib.py):
// `reps` passes that write n_records scattered 64-byte records (one lane each, 4 x 16 B), then `reps` passes that read them.
extern "C" int unc_calib_traffic(int device, uint64_t n_records, int reps) {
    HIPCHK(hipSetDevice(device));
    n_records |= 1;     // odd: the multiplicative scatter is then a permutation
    uint4 *buf = nullptr;
    uint32_t *sink = nullptr;
    HIPCHK(hipMalloc((void **)&buf, n_records * 64));
    HIPCHK(hipMalloc((void **)&sink, 4));
    for (int i = 0; i < reps; ++i) launch_calib(buf, n_records, 1, sink, nullptr);
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < reps; ++i) launch_calib(buf, n_records, 0, sink, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(buf); (void)hipFree(sink);
    return UNC_OK;
}

// Loaded latency of a region of device memory (diagnostics): `waves` single-wavefront workgroups, every lane a chain of `steps` dependent
// 16-byte loads over [base, base + bytes) (k_calib_chase); the launch's duration in milliseconds, the best of three.
extern "C" int unc_calib_chase(int device, const void *base, uint64_t bytes, uint32_t waves, uint32_t steps, float *ms_out) {
    if (!base || bytes < 16 || !ms_out) return fail(UNC_ERR_ARG, "unc_calib_chase: a region of 16 bytes or more");
    HIPCHK(hipSetDevice(device));
    uint32_t *sink = nullptr;
    hipEvent_t e0, e1;
    HIPCHK(hipMalloc((void **)&sink, 4));
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HIPCHK(hipEventRecord(e0, nullptr));
        launch_calib_chase((const uint4 *)base, bytes / 16, waves ? waves : 4096, steps ? steps : 2000, sink, nullptr);
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
    *ms_out = best;
    return UNC_OK;
}

// ------------------------------------------------------------------ step-wise trace of one read
extern "C" int unc_trace_begin(unc_mapper_t *m, const int16_t *raw, uint32_t n, const unc_calib_t *calib) {
    if (!m || !raw || !calib) return fail(UNC_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(m->ix->device));
    hipStream_t st = m->stream;
    uint64_t offsets[2] = {0, n};
    DevReads rd;
    int rc = stage_batch(m, 1, raw, offsets, calib, 0, st, &rd);
    if (rc) return rc;
    launch_events(rd, m->P, st, m->ev_rpw);
    launch_pool_init(m->pool, st);      // the traced read starts with every chunk free
    SlotState s0;
    memset(&s0, 0, sizeof s0);
    s0.max_map.rstart = 1; s0.max_map.evt_st = 1;   // NULL_ALN
    HIPCHK(hipMemcpyAsync(slot_state(m->sc, 0), &s0, sizeof s0, hipMemcpyHostToDevice, st));
    unc_evt_info_t info0;
    HIPCHK(hipMemcpyAsync(&info0, m->d_info, sizeof info0, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // (as the batch path: a read whose events overran the room for their means is reported, never traced on a truncated list)
    if (info0.pad) return fail(UNC_ERR_OVERFLOW, "traced read: more events than the room for event means holds (5/8 of its samples + 16)");
    m->trace_n = n;
    m->trace_active = true;
    return UNC_OK;
}

static int trace_reads(unc_mapper *m, DevReads *rd) {
    rd->raw = m->d_raw; rd->offsets = m->d_offsets; rd->calib = m->d_calib; rd->means = m->d_means; rd->moff = m->d_moff;
    rd->info = m->d_info; rd->n_reads = 1; rd->tgt_mean = m->ix->model_mean; rd->tgt_stdv = m->ix->model_stdv;
    rd->ring0 = nullptr; rd->new_read = nullptr; rd->ring_mod = 0;
    return UNC_OK;
}

extern "C" int unc_trace_step(unc_mapper_t *m, uint32_t n_events, int *done) {
    if (!m || !m->trace_active) return fail(UNC_ERR_ARG, "no trace in progress");
    HIPCHK(hipSetDevice(m->ix->device));
    DevReads rd;
    trace_reads(m, &rd);
    launch_map(m->ix->dev, m->sc, rd, m->P, m->d_results, m->d_next, n_events, 1, nullptr, 1, m->stream, m->pool);
    HIPCHK(hipGetLastError());
    SlotState s;
    HIPCHK(hipMemcpyAsync(&s, slot_state(m->sc, 0), sizeof s, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (done) *done = s.done ? 1 : 0;
    return UNC_OK;
}

extern "C" int unc_trace_paths(unc_mapper_t *m, unc_path_t *out, uint32_t cap, uint32_t *n_out) {
    if (!m || !m->trace_active) return fail(UNC_ERR_ARG, "no trace in progress");
    HIPCHK(hipSetDevice(m->ix->device));
    SlotState s;
    HIPCHK(hipMemcpy(&s, slot_state(m->sc, 0), sizeof s, hipMemcpyDeviceToHost));
    const DevScratch &sc = m->sc;
    std::vector<uint32_t> ord(s.n_parents ? s.n_parents : 1);
    std::vector<PathRec> recs(sc.max_paths);
    // The paths belong to event `gen`.  prob_sums_ is not stored (PathRec): entry 0 is `sub`, entry j the sum after the j-th
    // event of the window, rebuilt here exactly as the reference built it -- one float addition per event of the match
    // probability of that event's level with the k-mer the lineage had then (the record's k-mer history); the last entry must
    // come out as the record's `last`, which checks the history the kernel keeps.
    const int64_t gen = s.done == 1 ? (int64_t)s.event_i : (int64_t)s.event_i - 1;
    unc_evt_info_t inf;
    uint64_t moff = 0;
    HIPCHK(hipMemcpy(&inf, m->d_info, sizeof inf, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&moff, m->d_moff, 8, hipMemcpyDeviceToHost));
    std::vector<float> means(gen >= 0 ? (size_t)gen + 1 : 1);
    if (gen >= 0) HIPCHK(hipMemcpy(means.data(), m->d_means + moff, ((size_t)gen + 1) * 4, hipMemcpyDeviceToHost));
    const bool narrow = m->ix->dev.key_len_bits != 0;
    HIPCHK(hipMemcpy(ord.data(), sc.base + sc.off_order + (size_t)s.cur * sc.max_paths * 4, (size_t)s.n_parents * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(recs.data(), sc.base + sc.off_paths + (size_t)s.cur * sc.max_paths * sizeof(PathRec),
                     (size_t)sc.max_paths * sizeof(PathRec), hipMemcpyDeviceToHost));
    const float *mu = m->ix->model.data(), *v2 = mu + NKMER, *ld = mu + 2 * NKMER;
    for (uint32_t i = 0; i < s.n_parents && i < cap; ++i) {
        const PathRec &r = recs[ord[i]];
        unc_path_t &o = out[i];
        memset(&o, 0, sizeof o);
        if (narrow) { o.fm_start = r.r0; o.fm_end = r.r1; }
        else { const uint64_t v = ((uint64_t)r.r1 << 32) | r.r0; o.fm_start = v >> KEY_LEN_BITS; o.fm_end = o.fm_start + (v & KEY_LEN_MASK); }
        o.event_moves = r.moves;
        o.kmer = (uint16_t)(r.meta & META_KMER_MASK);
        o.length = (uint8_t)((r.meta >> META_LEN_SHIFT) & 31u);
        o.consec_stays = (uint8_t)((r.meta >> META_STAY_SHIFT) & 255u);
        o.sa_checked = (r.meta & META_SA_CHECKED) ? 1 : 0;
        const int L = o.length;
        // seed_prob_ as make_child / make_source left it (mapper.cpp:751-807)
        const bool slid = L == UNC_SEED_LEN && !(r.meta & META_FIRST_FULL);
        volatile float num = slid ? r.last - r.sub : r.last;
        volatile float sp = num / (float)L;
        o.seed_prob = sp;
        const uint64_t hist = ((uint64_t)r.hist_hi << 32) | r.hist_lo;
        uint32_t kmer = (uint32_t)hist & KMASK;
        const uint64_t fifo = hist >> 10;
        int queued = 0;
        for (int j = 0; j + 1 < L; ++j) queued += (r.moves >> j) & 1u;      // moves after the window's oldest event
        o.prob_sums[0] = slid ? r.sub : 0.0f;
        for (int j = 1; j <= L; ++j) {
            const int64_t t = gen - (L - j);                                 // the event behind entry j
            if (t < 0) return fail(UNC_ERR_HIP, "path %u is longer than the read so far", i);
            if (j > 1 && ((r.moves >> (L - j)) & 1u)) {                      // that event was a move: the next queued base joins the k-mer
                if (queued <= 0) return fail(UNC_ERR_HIP, "path %u: k-mer history has fewer bases than moves", i);
                --queued;
                kmer = ((kmer << 2) & KMASK) | (uint32_t)((fifo >> (2 * queued)) & 3u);
            }
            volatile float level = inf.scale * means[(size_t)t];
            level = level + inf.shift;
            volatile float d = level - mu[kmer];
            const double q = -((double)d * (double)d) / (double)v2[kmer];
            const float pr = (float)(q - (double)ld[kmer]);
            volatile float sum = o.prob_sums[j - 1] + pr;
            o.prob_sums[j] = sum;
        }
        if (L > 0 && kmer != o.kmer) return fail(UNC_ERR_HIP, "path %u: k-mer history ends in %u, the path's k-mer is %u", i, kmer, (unsigned)o.kmer);
    }
    *n_out = s.n_parents;
    return UNC_OK;
}

extern "C" int unc_trace_clusters(unc_mapper_t *m, unc_cluster_t *out, uint32_t cap, uint32_t *n_out, unc_cluster_t *max_map,
                                  float *len_sum, uint32_t *n_lens) {
    if (!m || !m->trace_active) return fail(UNC_ERR_ARG, "no trace in progress");
    HIPCHK(hipSetDevice(m->ix->device));
    SlotState s;
    HIPCHK(hipMemcpy(&s, slot_state(m->sc, 0), sizeof s, hipMemcpyDeviceToHost));
    // flatten the bucket grid (k_map.hip, add_seed): heads -> chains of nodes (hot keys, then cold parts), then set order
    // (ref_en_.start descending, evt_en_ descending: seed_tracker.cpp:97-102)
    const uint32_t n_buckets = m->ix->dev.n_buckets;
    std::vector<uint32_t> heads(n_buckets);
    HIPCHK(hipMemcpy(heads.data(), m->sc.base + m->sc.off_cl_dir, (size_t)n_buckets * 4, hipMemcpyDeviceToHost));
    const uint32_t n_chunks = (s.n_alloc + CHUNK_NODES - 1) / CHUNK_NODES;
    std::vector<uint32_t> chunk_ids(n_chunks ? n_chunks : 1);
    HIPCHK(hipMemcpy(chunk_ids.data(), m->sc.base + m->sc.off_cl_chunks, (size_t)n_chunks * 4, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::vector<char>> chunks;            // pool chunk -> its bytes
    for (uint32_t c = 0; c < n_chunks; ++c) {
        std::vector<char> &buf = chunks[chunk_ids[c]];
        buf.resize((size_t)CHUNK_NODES * NODE_BYTES);
        HIPCHK(hipMemcpy(buf.data(), m->pool.nodes + (size_t)chunk_ids[c] * POOL_CHUNK_BYTES, buf.size(), hipMemcpyDeviceToHost));
    }
    std::vector<unc_cluster_t> all;
    for (uint32_t b = 0; b < n_buckets; ++b) {
        uint32_t node1 = heads[b], guard = 0;
        while (node1) {
            const uint32_t node = node1 - 1;
            auto it = chunks.find(node / CHUNK_NODES);
            if (it == chunks.end() || ++guard > s.n_alloc + 1) return fail(UNC_ERR_HIP, "seed-cluster grid is inconsistent: bucket %u points outside the read's chunks", b);
            const char *np = it->second.data() + (size_t)(node % CHUNK_NODES) * NODE_BYTES;
            uint32_t hdr[4];
            memcpy(hdr, np, 16);
            if (hdr[0] > NODE_K) return fail(UNC_ERR_HIP, "seed-cluster grid is inconsistent: a node of bucket %u holds %u clusters", b, hdr[0]);
            const ClusterKey *hot = reinterpret_cast<const ClusterKey *>(np + 16);
            const ClusterCold *cold = reinterpret_cast<const ClusterCold *>(np + 16 + NODE_K * 16);
            for (uint32_t e = 0; e < hdr[0]; ++e) {
                if ((hot[e].rstart >> m->ix->dev.bucket_shift) != b) return fail(UNC_ERR_HIP, "seed-cluster grid is inconsistent: a cluster sits in the wrong bucket");
                unc_cluster_t c;
                c.ref_st = cold[e].ref_st; c.ref_en_start = hot[e].rstart; c.ref_en_end = cold[e].rend;
                c.evt_st = cold[e].evt_st; c.evt_en = hot[e].evt_en; c.total_len = hot[e].total_len; c.pad = 0;
                all.push_back(c);
            }
            node1 = hdr[1];
        }
    }
    std::sort(all.begin(), all.end(), [](const unc_cluster_t &a, const unc_cluster_t &b) {
        return a.ref_en_start > b.ref_en_start || (a.ref_en_start == b.ref_en_start && a.evt_en > b.evt_en);
    });
    uint32_t n = (uint32_t)all.size();
    for (uint32_t i = 0; i < n && i < cap; ++i) out[i] = all[i];
    if (n != s.n_clusters) return fail(UNC_ERR_HIP, "seed-cluster set is inconsistent: %u keys, %u clusters", n, s.n_clusters);
    *n_out = s.n_clusters;
    if (max_map) {
        max_map->ref_st = s.max_map.ref_st; max_map->ref_en_start = s.max_map.rstart; max_map->ref_en_end = s.max_map.rend;
        max_map->evt_st = s.max_map.evt_st; max_map->evt_en = s.max_map.evt_en; max_map->total_len = s.max_map.total_len; max_map->pad = 0;
    }
    if (len_sum) *len_sum = s.len_sum;
    if (n_lens) *n_lens = s.n_lens;
    return UNC_OK;
}

extern "C" int unc_trace_finish(unc_mapper_t *m, unc_hit_t *hit) {
    if (!m || !m->trace_active) return fail(UNC_ERR_ARG, "no trace in progress");
    HIPCHK(hipSetDevice(m->ix->device));
    SlotState s;
    HIPCHK(hipMemcpy(&s, slot_state(m->sc, 0), sizeof s, hipMemcpyDeviceToHost));
    unc_evt_info_t inf;
    HIPCHK(hipMemcpy(&inf, m->d_info, sizeof inf, hipMemcpyDeviceToHost));
    DevResult res;
    memset(&res, 0, sizeof res);
    res.done = s.done; res.status = s.status; res.event_i = s.event_i; res.cluster = s.max_map;
    res.n_nbr = s.n_nbr; res.n_sa = s.n_sa; res.n_lf = s.n_lf;
    fill_hit(m->ix, m->P, res, inf, m->trace_n, hit);
    m->trace_active = false;
    return UNC_OK;
}

// ------------------------------------------------------------------ chunked (realtime) path
struct RtHostChan { int state = 0; /* 0 inactive, 1 mapping */ uint32_t number = 0, chunk_count = 0; uint64_t raw_len = 0; };

struct unc_rt {
    const unc_index *ix = nullptr;
    unc_params_t P;
    uint32_t n_channels = 0;
    uint32_t team = 8;            // wavefronts per channel in k_map (UNC_RT_TEAM)
    DevScratch sc;
    DevPool pool{};               // nodes of the channels' seed-cluster grids (a channel keeps its chunks until its read is decided)
    RtChan *d_chans = nullptr;
    float *d_ring = nullptr;
    RtChunkDesc *d_desc = nullptr;
    unc_evt_info_t *d_info = nullptr;
    uint32_t *d_ring0 = nullptr, *d_newread = nullptr, *d_slotmap = nullptr, *d_next = nullptr;
    uint64_t *d_moff = nullptr;
    DevResult *d_results = nullptr;
    int16_t *d_raw = nullptr;
    size_t raw_cap = 0;
    uint64_t device_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double win_start_ms = 0, win_end_ms = 0;      // k_map of the last batch on the process-wide time axis (unc_mapper_last_window)
    float ms_events = 0, ms_map = 0;
    std::vector<RtHostChan> chans;
    std::vector<SlotState> h_state;
    std::vector<unc_evt_info_t> h_info;
    // dev: UNC_RT_PROFILE=1 launches the cycle-counting instantiation of k_map and prints the phase shares of all finished reads on free
    bool profile = false;
    uint64_t cyc_sum[12] = {0};
};

extern "C" void unc_rt_free(unc_rt_t *rt) {
    if (!rt) return;
    if (rt->profile) {
        static const char *names[12] = {"probs", "extend_rest", "sort", "walk", "sources", "sa", "add_seed", "rest", "e1_parents", "e2_fm", "e3_slots", "e4_children"};
        double tot = 0;
        for (uint64_t c : rt->cyc_sum) tot += (double)c;
        fprintf(stderr, "UNC_RT_PROFILE phase cycle shares of the finished reads:");
        for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.3f", names[i], tot > 0 ? (double)rt->cyc_sum[i] / tot : 0.0);
        fprintf(stderr, "\n");
    }
    (void)hipSetDevice(rt->ix->device);
    free_pool(rt->pool);
    void *ptrs[] = {rt->sc.base,
                    rt->d_chans, rt->d_ring, rt->d_desc, rt->d_info, rt->d_ring0, rt->d_newread, rt->d_slotmap, rt->d_next, rt->d_moff,
                    rt->d_results, rt->d_raw};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : rt->ev) if (e) (void)hipEventDestroy(e);
    if (rt->stream) (void)hipStreamDestroy(rt->stream);
    delete rt;
}

extern "C" int unc_rt_create(const unc_index_t *ix, const unc_params_t *p, uint32_t n_channels, unc_rt_t **out) {
    if (!ix || !p || !out || n_channels == 0) return fail(UNC_ERR_ARG, "bad argument");
    *out = nullptr;
    if (p->seed_len != UNC_SEED_LEN || p->window_length1 != UNC_WINDOW1 || p->window_length2 != UNC_WINDOW2)
        return fail(UNC_ERR_ARG, "seed_len/window lengths must be %d/%d/%d", UNC_SEED_LEN, UNC_WINDOW1, UNC_WINDOW2);
    if (p->max_paths == 0 || p->max_paths > 65535 || p->max_rep_copy > (uint32_t)MAX_REP_COPY_LIMIT || p->max_consec_stay > 255 ||
        p->max_events > 65535)
        return fail(UNC_ERR_ARG, "unsupported parameter value");
    // A chunk may hold at most 2 * NORM_LEN samples (3 s at 4 kHz; the reference's default is 1 s).  The detector never fires on two
    // consecutive samples (a peak needs a sample to be recorded and more than window / 2 further ones to be confirmed, and a short-window
    // peak above the threshold resets the long-window detector: event_detector.cpp:221-279), so such a chunk yields at most NORM_LEN events
    // and the rolling normaliser, which update() leaves empty after every chunk, cannot fill: the #SKIP branch of Mapper::process_chunk
    // (mapper.cpp:336-351) -- which on a full ring drops the forest and, when the second push fails too, leaves the chunk unprocessed
    // so that it is fed to the detector AGAIN -- is unreachable for every chunk this interface accepts, here and in the reference.
    if (!(p->chunk_time > 0.0f) || !(p->sample_rate > 0.0f) || (double)p->chunk_time * (double)p->sample_rate > 2.0 * (double)NORM_LEN)
        return fail(UNC_ERR_ARG, "chunk_time * sample_rate = %.0f samples: chunks of more than %u samples are not supported (mapper.cpp:336-351, "
                                 "the #SKIP branch of the reference, is out of scope)", (double)p->chunk_time * (double)p->sample_rate, 2u * NORM_LEN);
    HIPCHK(hipSetDevice(ix->device));
    unc_rt *rt = new unc_rt();
    struct Guard { unc_rt *p; ~Guard() { if (p) unc_rt_free(p); } } guard{rt};
    rt->ix = ix; rt->P = *p; rt->n_channels = n_channels;
    { const char *e = getenv("UNC_RT_PROFILE"); rt->profile = e && e[0] == '1'; }
    // wavefronts per channel (k_map_team): 8 unless UNC_RT_TEAM says 1 (the one-wavefront kernel), 2 or 4.  512 channels x 8 = the
    // 4096 wavefronts the chip holds; measured on E. coli thresholds, ms per round of 512 chunks: 110 / 102 / 90 / 84 with 1 / 2 / 4 / 8
    {
        const char *e = getenv("UNC_RT_TEAM");
#ifdef LANESIM
        // the emulator suite's default (teams of 2: 512 fibers per emulated workgroup are three times the run time) is a variable only
        // the emulator build reads, so that it cannot change what the gfx950 library does in the same process (round-4 advice)
        if (!e) e = getenv("UNC_SIM_RT_TEAM");
#endif
        const long v = e ? atol(e) : 8;
        rt->team = v >= 8 ? 8u : v >= 4 ? 4u : v >= 2 ? 2u : 1u;
    }
    const size_t S = n_channels;
    size_t bytes = 0;
    {
        PlacementSpacer spacer(S * scratch_slot_bytes(*p, 1u << 20, 2 * p->max_paths, ix->dev) + S * 64 * (size_t)POOL_CHUNK_BYTES);
        int rc = alloc_scratch(rt->sc, *p, S, 1u << 20, 2 * p->max_paths, &bytes, ix->dev);
        if (rc) return rc;
        HIPCHK(hipMemset(rt->sc.base, 0, bytes));
        // the channels' node pool: 16 chunks (12 288 nodes) per channel on average, 64 on references of 2^26 rows and more (a read
        // there touches tens of thousands of buckets); UNC_RT_POOL_CHUNKS overrides.  A channel's chunks go back when its read is
        // decided; a read that finds the pool dry fails with its status set (there is no second pass in chunked mode).
        size_t per_ch = ix->seq_len >= (1ull << 26) ? 64 : 16;
        size_t n_chunks = S * per_ch;
        {
            // never more than 60 % of the HBM that is free now (the batch path's rule): next to a dense SA, or on a smaller GPU,
            // the pool shrinks -- with a warning, and never below four chunks per channel -- instead of failing the whole create
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            free_b += spacer.bytes;
            const size_t fit = free_b / 5 * 3 / POOL_CHUNK_BYTES;
            if (n_chunks > fit) {
                const size_t floor_chunks = std::max<size_t>(64, (size_t)S * 4);
                const size_t clamped = std::max(fit, floor_chunks);
                fprintf(stderr, "Warning: realtime node pool clamped from %zu to %zu chunks (%.1f GB of HBM free)\n", n_chunks, clamped, (double)free_b / 1e9);
                n_chunks = clamped;
            }
        }
        n_chunks = std::max<size_t>(64, n_chunks);
        // (the override is taken as it is, also below the floor: how the tests make a pool small enough to notice a chunk that is not
        // handed back)
        if (const char *e = getenv("UNC_RT_POOL_CHUNKS")) { const long v = atol(e); if (v > 0) n_chunks = (size_t)v; }
        rc = alloc_pool(rt->pool, (uint32_t)n_chunks, &bytes);
        if (rc) return rc;
    }
#define RALLOC(ptr, type, count)                                   \
    do {                                                           \
        size_t b_ = (size_t)(count) * sizeof(type);                \
        HIPCHK(hipMalloc((void **)&(ptr), b_));                    \
        HIPCHK(hipMemset((ptr), 0, b_));                           \
        bytes += b_;                                               \
    } while (0)
    RALLOC(rt->d_chans, RtChan, S);
    RALLOC(rt->d_ring, float, S * NORM_LEN);
    RALLOC(rt->d_desc, RtChunkDesc, S);
    RALLOC(rt->d_info, unc_evt_info_t, S);
    RALLOC(rt->d_ring0, uint32_t, S);
    RALLOC(rt->d_newread, uint32_t, S);
    RALLOC(rt->d_slotmap, uint32_t, S);
    RALLOC(rt->d_next, uint32_t, 16);
    RALLOC(rt->d_moff, uint64_t, S + 1);
    RALLOC(rt->d_results, DevResult, S);
#undef RALLOC
    rt->device_bytes = bytes;
    rt->chans.resize(n_channels);
    HIPCHK(hipStreamCreate(&rt->stream));
    for (auto &e : rt->ev) HIPCHK(hipEventCreate(&e));
    guard.p = nullptr;
    *out = rt;
    return UNC_OK;
}

extern "C" uint64_t unc_rt_device_bytes(const unc_rt_t *rt) { return rt->device_bytes; }
extern "C" int unc_rt_last_timing(const unc_rt_t *rt, float *ms_events, float *ms_map) {
    if (ms_events) *ms_events = rt->ms_events;
    if (ms_map) *ms_map = rt->ms_map;
    return UNC_OK;
}

extern "C" int unc_rt_tap_channel(unc_rt_t *rt, uint32_t channel, unc_rt_tap_t *out, float *ring) {
    if (!rt || !out || !ring || channel >= rt->n_channels) return fail(UNC_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(rt->ix->device));
    RtChan c;
    HIPCHK(hipMemcpy(&c, rt->d_chans + channel, sizeof c, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ring, rt->d_ring + (size_t)channel * NORM_LEN, (size_t)NORM_LEN * 4, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->det_t = c.t; out->det_total_events = c.total_events; out->det_len_sum = c.len_sum;
    out->norm_n = c.n_n; out->norm_wr = c.n_wr; out->norm_mean = c.n_mean; out->norm_varsum = c.n_varsum;
    out->prof_n = c.pw_n; out->prof_to_mask = c.to_mask; out->prof_queued = c.q_len; out->prof_mean = c.pw_mean; out->prof_varsum = c.pw_varsum;
    for (uint32_t i = 0; i < c.q_len && i < 28; ++i) out->prof_queue[i] = c.evq[(c.q_head + i) % (PROF_WIN + 1)];
    return UNC_OK;
}

// unc_hit_t::notes of a channel's read: what the kernel has noted so far (path buffer full) and, once the read has ENDED -- by the
// kernel's decision or by the host's (max_chunks, last chunk) --, whether sources_added_ flags are left for the channel's next read
static uint32_t rt_notes(const SlotState &s, bool ended) {
    uint32_t notes = s.notes;
    if (ended)
        for (uint32_t w : s.sources_added) if (w) { notes |= UNC_NOTE_FLAGS_LEFT; break; }
    return notes;
}

static void rt_unmapped(const unc_rt *rt, const RtHostChan &hc, const SlotState &s, const unc_evt_info_t *inf, unc_hit_t *h, bool ended = true) {
    DevResult res;
    memset(&res, 0, sizeof res);
    res.done = 2; res.status = s.status; res.event_i = s.event_i; res.notes = rt_notes(s, ended);
    res.n_nbr = s.n_nbr; res.n_sa = s.n_sa; res.n_lf = s.n_lf;
    unc_evt_info_t z;
    memset(&z, 0, sizeof z);
    fill_hit(rt->ix, rt->P, res, inf ? *inf : z, hc.raw_len, h);
}

// raw (int16 + per-chunk calibration) or raw_pa (floats taken as they are): exactly one of the two
static int rt_process(unc_rt_t *rt, uint32_t n_chunks, const unc_rt_chunk_t *chunks, const int16_t *raw, const float *raw_pa, int on_device,
                      void *stream, unc_rt_result_t *results) {
    if (!rt || !chunks || (!raw && !raw_pa) || !results) return fail(UNC_ERR_ARG, "null argument");
    const size_t esz = raw_pa ? sizeof(float) : sizeof(int16_t);
    if (n_chunks > rt->n_channels) return fail(UNC_ERR_ARG, "more chunks than channels");
    HIPCHK(hipSetDevice(rt->ix->device));
    hipStream_t st = stream ? (hipStream_t)stream : rt->stream;
    const uint32_t chunk_len = (uint32_t)(rt->P.chunk_time * rt->P.sample_rate);

    // ---- host decisions that the reference takes before any signal is touched (RealtimePool::add_chunk /
    //      try_add_chunk, Mapper::add_chunk): which chunks start a read, continue one, or are dropped
    std::vector<RtChunkDesc> desc;
    std::vector<uint32_t> slotmap, newread, active;   // active[i] = index into chunks[]
    std::vector<uint64_t> moff;
    std::vector<char> seen(rt->n_channels, 0);
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t i = 0; i < n_chunks; ++i) {
        const unc_rt_chunk_t &c = chunks[i];
        memset(&results[i], 0, sizeof results[i]);
        results[i].hit.rid = -1;
        if (c.channel >= rt->n_channels) return fail(UNC_ERR_ARG, "chunk %u: channel %u out of range", i, c.channel);
        if (seen[c.channel]) return fail(UNC_ERR_ARG, "two chunks for channel %u in one call", c.channel);
        if (c.n_samples > chunk_len) return fail(UNC_ERR_ARG, "chunk %u longer than chunk_time * sample_rate", i);
        seen[c.channel] = 1;
        RtHostChan &hc = rt->chans[c.channel];
        if (c.flags & UNC_RT_FIRST) {
            hc.state = 1; hc.number = c.read_number; hc.chunk_count = 1; hc.raw_len = c.n_samples;   // ReadBuffer(Chunk&)
        } else if (hc.state != 1 || hc.number != c.read_number) {
            results[i].state = UNC_RT_IGNORED;
            continue;
        } else if (hc.chunk_count >= rt->P.max_chunks) {
            // Mapper::add_chunk: read_.chunks_maxed() -> set_failed(), mapper.cpp:289-296 (the chunk is dropped)
            results[i].state = UNC_RT_FAILED;
            hc.state = 0;
            active.push_back(i | 0x80000000u);   // needs the slot state for the record, nothing to launch
            continue;
        } else {
            hc.chunk_count++;
            hc.raw_len += c.n_samples;
        }
        RtChunkDesc d;
        d.offset = c.offset; d.n_samples = c.n_samples; d.channel = c.channel; d.new_read = (c.flags & UNC_RT_FIRST) ? 1u : 0u;
        d.cal_range = c.calib.range; d.cal_offset = c.calib.offset; d.cal_digit = c.calib.digitisation;
        desc.push_back(d);
        slotmap.push_back(c.channel);
        newread.push_back(d.new_read);
        moff.push_back((uint64_t)c.channel * NORM_LEN);
        active.push_back(i);
        if (c.n_samples) { lo = c.offset < lo ? c.offset : lo; hi = c.offset + c.n_samples > hi ? c.offset + c.n_samples : hi; }
    }
    const uint32_t n_act = (uint32_t)desc.size();
    rt->ms_events = rt->ms_map = 0;
    if (n_act) {
        const char *d_base = raw_pa ? reinterpret_cast<const char *>(raw_pa) : reinterpret_cast<const char *>(raw);
        if (!on_device) {
            const uint64_t span = hi > lo ? hi - lo : 0;
            if (span * 2 > rt->raw_cap) {              // raw_cap counts int16 elements; floats take two each
                if (rt->d_raw) (void)hipFree(rt->d_raw);
                rt->d_raw = nullptr;
                HIPCHK(hipMalloc((void **)&rt->d_raw, (span * 2 + 64) * 2));
                rt->raw_cap = span * 2;
            }
            if (span) HIPCHK(hipMemcpyAsync(rt->d_raw, d_base + lo * esz, span * esz, hipMemcpyHostToDevice, st));
            d_base = reinterpret_cast<const char *>(rt->d_raw) - lo * esz;
        }
        const int16_t *d_raw = raw_pa ? nullptr : reinterpret_cast<const int16_t *>(d_base);
        const float *d_pa = raw_pa ? reinterpret_cast<const float *>(d_base) : nullptr;
        moff.push_back(0);
        HIPCHK(hipMemcpyAsync(rt->d_desc, desc.data(), n_act * sizeof(RtChunkDesc), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(rt->d_slotmap, slotmap.data(), n_act * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(rt->d_newread, newread.data(), n_act * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(rt->d_moff, moff.data(), (n_act + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(rt->ev[0], st));
        launch_rt_events(d_raw, d_pa, rt->d_desc, n_act, rt->d_chans, rt->d_ring, rt->P, rt->ix->model_mean, rt->ix->model_stdv, rt->d_info,
                         rt->d_ring0, st);
        HIPCHK(hipEventRecord(rt->ev[1], st));
        DevReads rd;
        memset(&rd, 0, sizeof rd);
        rd.means = rt->d_ring; rd.moff = rt->d_moff; rd.info = rt->d_info; rd.n_reads = n_act;
        rd.tgt_mean = rt->ix->model_mean; rd.tgt_stdv = rt->ix->model_stdv;
        rd.ring0 = rt->d_ring0; rd.new_read = rt->d_newread; rd.ring_mod = NORM_LEN;
        launch_map(rt->ix->dev, rt->sc, rd, rt->P, rt->d_results, rt->d_next, 0xFFFFFFFFu, 1, rt->d_slotmap, n_act, st, rt->pool, nullptr, nullptr, nullptr,
                   rt->profile, nullptr, nullptr, rt->team);
        HIPCHK(hipEventRecord(rt->ev[2], st));
        HIPCHK(hipGetLastError());
        rt->h_info.resize(n_act);
        HIPCHK(hipMemcpyAsync(rt->h_info.data(), rt->d_info, n_act * sizeof(unc_evt_info_t), hipMemcpyDeviceToHost, st));
    }
    rt->h_state.resize(rt->n_channels);
    HIPCHK(hipMemcpy2DAsync(rt->h_state.data(), sizeof(SlotState), rt->sc.base + rt->sc.off_state, rt->sc.slot_bytes, sizeof(SlotState), rt->n_channels,
                             hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_act) {
        HIPCHK(hipEventElapsedTime(&rt->ms_events, rt->ev[0], rt->ev[1]));
        HIPCHK(hipEventElapsedTime(&rt->ms_map, rt->ev[1], rt->ev[2]));
    }

    // ---- outcome per chunk (Mapper::map_chunk's exits, mapper.cpp:381-431)
    int worst = UNC_OK;
    uint32_t di = 0;
    for (uint32_t a : active) {
        const bool dropped = a & 0x80000000u;
        const uint32_t i = a & 0x7FFFFFFFu;
        const unc_rt_chunk_t &c = chunks[i];
        RtHostChan &hc = rt->chans[c.channel];
        const SlotState &s = rt->h_state[c.channel];
        if (dropped) { rt_unmapped(rt, hc, s, nullptr, &results[i].hit); continue; }
        const unc_evt_info_t &inf = rt->h_info[di++];
        if (inf.pad || s.status) {   // RtChan status travels in info.pad
            results[i].state = UNC_RT_FAILED;
            rt_unmapped(rt, hc, s, &inf, &results[i].hit);
            results[i].hit.status = s.status | inf.pad;
            hc.state = 0;
            worst = UNC_ERR_OVERFLOW;
        } else if (s.done == 1) {
            DevResult res;
            memset(&res, 0, sizeof res);
            res.done = 1; res.event_i = s.event_i; res.cluster = s.max_map; res.n_nbr = s.n_nbr; res.n_sa = s.n_sa; res.n_lf = s.n_lf;
            res.notes = rt_notes(s, true);
            fill_hit(rt->ix, rt->P, res, inf, hc.raw_len, &results[i].hit);
            results[i].state = UNC_RT_MAPPED;
            hc.state = 0;
        } else if (s.done == 2 || s.event_i >= rt->P.max_events) {
            results[i].state = UNC_RT_FAILED; results[i].ended = 1;        // event_i_ >= max_events: set_failed + set_ended
            rt_unmapped(rt, hc, s, &inf, &results[i].hit);
            hc.state = 0;
        } else if (hc.chunk_count >= rt->P.max_chunks) {
            results[i].state = UNC_RT_FAILED;                               // norm_.empty() && chunks_maxed(), :392-405
            rt_unmapped(rt, hc, s, &inf, &results[i].hit);
            hc.state = 0;
        } else if (c.flags & UNC_RT_LAST) {
            results[i].state = UNC_RT_FAILED; results[i].ended = 1;        // request_reset -> set_failed + set_ended
            rt_unmapped(rt, hc, s, &inf, &results[i].hit);
            hc.state = 0;
        } else {
            results[i].state = UNC_RT_MAPPING;
            rt_unmapped(rt, hc, s, &inf, &results[i].hit, false);           // progress so far (event_i, counters)
        }
        if (rt->profile && hc.state == 0) for (int k = 0; k < 12; ++k) rt->cyc_sum[k] += s.cyc[k];
    }
    if (worst) return fail(worst, "device scratch overflow on at least one channel (see hit.status)");
    return UNC_OK;
}

extern "C" int unc_rt_process_chunks(unc_rt_t *rt, uint32_t n_chunks, const unc_rt_chunk_t *chunks, const int16_t *raw, int on_device,
                                     void *stream, unc_rt_result_t *results) {
    if (!raw) return fail(UNC_ERR_ARG, "null argument");
    return rt_process(rt, n_chunks, chunks, raw, nullptr, on_device, stream, results);
}

extern "C" int unc_rt_process_chunks_f32(unc_rt_t *rt, uint32_t n_chunks, const unc_rt_chunk_t *chunks, const float *signal, int on_device,
                                         void *stream, unc_rt_result_t *results) {
    if (!signal) return fail(UNC_ERR_ARG, "null argument");
    return rt_process(rt, n_chunks, chunks, nullptr, signal, on_device, stream, results);
}

// ------------------------------------------------------------------ uncalled index: self alignment
extern "C" int unc_self_align(const unc_index_t *ix, const char *bwa_prefix, uint32_t sample_dist, uint32_t cap, uint64_t *lens,
                              uint32_t *full_len, uint64_t max_paths, uint64_t *n_paths) {
    if (!ix || !bwa_prefix || !n_paths || sample_dist == 0) return fail(UNC_ERR_ARG, "bad argument");
    // the reference draws rand() once per base of every sequence after srand(0) (self_align_ref.cpp:37,67)
    std::vector<uint64_t> pos, remain;
    srand(0);
    uint64_t st = 0;
    for (const SeqAnn &sq : ix->seqs) {
        for (uint64_t i = 0; i < sq.len; ++i) {
            if (rand() % (int)sample_dist != 0) continue;
            pos.push_back(st + i);
            remain.push_back(sq.len - i);
        }
        st += sq.len;
    }
    *n_paths = pos.size();
    if (!lens) return UNC_OK;
    if (pos.size() > max_paths) return fail(UNC_ERR_ARG, "output too small: %zu trajectories", pos.size());
    if (pos.empty()) return UNC_OK;
    std::vector<char> pac;
    if (!read_file(std::string(bwa_prefix) + ".pac", pac)) return fail(UNC_ERR_IO, "cannot read %s.pac", bwa_prefix);
    if ((int64_t)pac.size() < ix->l_pac / 4 + 1) return fail(UNC_ERR_IO, "%s.pac too short", bwa_prefix);
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t n = (uint32_t)pos.size();
    DevBuf<uint8_t> d_pac;
    DevBuf<uint64_t> d_pos, d_rem, d_out;
    DevBuf<uint32_t> d_len;
    HIPCHK(d_pac.alloc(pac.size() + 16)); HIPCHK(d_pos.alloc(n)); HIPCHK(d_rem.alloc(n)); HIPCHK(d_out.alloc((size_t)n * cap)); HIPCHK(d_len.alloc(n));
    HIPCHK(hipMemcpy(d_pac.p, pac.data(), pac.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_pos.p, pos.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_rem.p, remain.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_out.p, 0, (size_t)n * cap * 8));
    launch_self_align(ix->dev, d_pac.p, n, d_pos.p, d_rem.p, d_out.p, cap, d_len.p, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(lens, d_out.p, (size_t)n * cap * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(full_len, d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return UNC_OK;
}
