// Small kernels around the FM index and the pore model: the k-mer range table built at index load
// (BwaIndex::load_index, bwa_index.hpp:124-132) and array-at-a-time taps used by the parity tests.
#include <hip/hip_runtime.h>

#include "fm_dev.h"
#include "unc_dev_types.h"
#include "unc_kernels.h"

namespace unc {

__global__ void k_kmer_ranges(DevIndex ix, uint64_t *out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint32_t)NKMER) return;
    const uint32_t head = (k >> (2 * UNC_KLEN - 2)) & 3u;          // kmer_head, bp.hpp:100-103
    uint64_t s = ix.L2[head], e = ix.L2[head + 1];                 // get_base_range, bwa_index.hpp:172-174
    for (int i = 1; i < UNC_KLEN; ++i) {
        const uint32_t b = (k >> (2 * (UNC_KLEN - i - 1))) & 3u;   // kmer_base, bp.hpp:111-114
        fm_get_neighbor(ix, s, e, b, &s, &e);
    }
    out[2 * k] = s;
    out[2 * k + 1] = e;
}

__global__ void k_fm_neighbor(DevIndex ix, uint32_t n, const uint64_t *s, const uint64_t *e, const uint8_t *b, uint64_t *os, uint64_t *oe) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t a, c;
    if (ix.fm32 && s[i] >= 1 && s[i] <= e[i] && e[i] <= ix.seq_len) {      // what k_map does on a reference of fewer than 2^32 rows
        uint32_t a32, c32;
        fm32_get_neighbor(ix, (uint32_t)s[i], (uint32_t)e[i], b[i], &a32, &c32);
        a = a32; c = c32;
        if (a32 > c32) fm_get_neighbor(ix, s[i], e[i], b[i], &a, &c);      // an empty result: report the BWA-format pair
    } else fm_get_neighbor(ix, s[i], e[i], b[i], &a, &c);
    os[i] = a;
    oe[i] = c;
}

__global__ void k_fm_sa(DevIndex ix, uint32_t n, const uint64_t *rows, uint64_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t steps;
    // row 0 is the sentinel suffix (SA stored as -1); path ranges never contain it
    out[i] = (ix.sa_dense && rows[i] != 0) ? fm_sa_dense(ix, rows[i], &steps) : fm_sa(ix, rows[i], &steps);
}

// `uncalled index` self-alignment (self_align_ref.cpp:34-91): from every sampled reference position walk
// the complemented forward strand through the FM index and record the range size before each step, until the
// range is unique or the contig ends.  One lane per sample; the first SA_CAP sizes are stored, the full
// length is always counted (IndexParameterizer only reads the head of each trajectory, index.py:84-100).
__global__ void k_self_align(DevIndex ix, const uint8_t *pac, uint32_t n_samples, const uint64_t *pos, const uint64_t *remain,
                             uint64_t *out, uint32_t cap, uint32_t *out_len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const uint64_t p0 = pos[i], rem = remain[i];      // rem = contig length - offset within the contig
    uint64_t *o = out + (size_t)i * cap;
    uint32_t n = 0;
    uint32_t b = 3u - ((pac[p0 >> 2] >> (((3u ^ (uint32_t)p0) & 3u) << 1)) & 3u);   // BASE_COMP_B[get_base(st+i)]
    uint64_t s = ix.L2[b], e = ix.L2[b + 1];                                          // get_base_range
    uint64_t j = 1;
    for (; j < rem && e - s + 1 > 1; ++j) {
        if (n < cap) o[n] = e - s + 1;
        ++n;
        const uint64_t pj = p0 + j;
        b = 3u - ((pac[pj >> 2] >> (((3u ^ (uint32_t)pj) & 3u) << 1)) & 3u);
        fm_get_neighbor(ix, s, e, b, &s, &e);
    }
    if (e - s + 1 > 0) {   // "happens on Ns"
        if (n < cap) o[n] = e - s + 1;
        ++n;
    }
    out_len[i] = n;
}

// Dense SA: every row walks to its sampled row once, at index load (bwt_sa for all rows in parallel).  Grid-stride: a
// launch of one thread per row would need more than 2^32 threads for GRCh38 (6.2 G rows), which HIP does not dispatch.
// (two rows per thread: their 12 bytes are three aligned words)
__global__ void k_dense_sa(DevIndex ix, uint64_t *out_) {
    uint32_t *const out = reinterpret_cast<uint32_t *>(out_);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, groups = (ix.seq_len + 2) / 2;      // rows 0 .. seq_len
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        uint64_t v[2];
        for (int j = 0; j < 2; ++j) {
            const uint64_t k = 2 * g + (uint64_t)j;
            v[j] = 0;
            if (k <= ix.seq_len) {
                uint32_t steps;
                const uint64_t sa = fm_sa(ix, k, &steps);      // row 0 (SA = -1) is never looked up
                v[j] = sa_pack(sa, steps);
            }
        }
        uint32_t *const p = out + 3 * g;
        p[0] = (uint32_t)v[0];
        p[1] = (uint32_t)(v[0] >> 32) | (uint32_t)(v[1] << 16);
        p[2] = (uint32_t)(v[1] >> 16);
    }
}

// Load-time self-check of the dense table: `n` probe rows spread over the whole index (the last row included) must agree
// with the BWA-format walk; *bad counts the ones that do not.
__global__ void k_dense_sa_check(DevIndex ix, const uint64_t *dense, uint32_t n, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = n > 1 ? (uint64_t)((__uint128_t)ix.seq_len * i / (n - 1)) : ix.seq_len;
    if (k == 0) k = 1;
    uint32_t steps;
    const uint64_t sa = fm_sa(ix, k, &steps);
    if (sa_entry_load(dense, k) != sa_pack(sa, steps) || steps > SA_STEP_MAX) atomicAdd(bad, 1u);
}


// The 32-bit rank table (fm_dev.h, fm32_get_neighbor) from the BWA blocks: one thread per block of 64 symbols.
__global__ void k_build_fm32(DevIndex ix, uint32_t *out, uint32_t n_blk) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blk) return;
    const uint64_t base = (uint64_t)b << 6;
    uint32_t cnt[4];
    for (uint32_t c = 0; c < 4; ++c) {
        uint64_t n = ix.L2[c];
        if (base > 0 && base <= ix.seq_len) {       // occurrences among the symbols [0 .. base - 1] (symbol indices: no sentinel)
            const FmBlock blk = fm_load_block(ix, base - 1);
            n += fm_block_count(blk, c) + fm_block_rank(blk, base - 1, c);
        } else if (base > ix.seq_len) n = ix.L2[c + 1];
        cnt[c] = (uint32_t)n;
    }
    uint32_t h_lo = 0, h_hi = 0, l_lo = 0, l_hi = 0;
    for (uint32_t j = 0; j < 64; ++j) {
        const uint64_t p = base + j;
        if (p >= ix.seq_len) break;
        const uint32_t sym = (ix.bwt[((p >> 7) << 4) + 8 + ((p & 127) >> 4)] >> ((~p & 15) << 1)) & 3u;
        if (j < 32) { h_lo |= (sym >> 1) << j; l_lo |= (sym & 1u) << j; }
        else { h_hi |= (sym >> 1) << (j - 32); l_hi |= (sym & 1u) << (j - 32); }
    }
    uint4 *o = reinterpret_cast<uint4 *>(out) + ((size_t)b << 1);
    o[0] = make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]);
    o[1] = make_uint4(h_lo, h_hi, l_lo, l_hi);
}
// Load-time self-check of the table: `n` pseudo-random steps (rows spread over the whole index, short and long ranges, every
// base) must give what the BWA-format arithmetic gives; *bad counts the ones that do not.
__global__ void k_fm32_check(DevIndex ix, uint32_t n, uint32_t *bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint64_t s = 1 + x % ix.seq_len;                       // 1 .. seq_len
    uint64_t len = (i & 3u) == 0 ? (x >> 40) % ix.seq_len : (x >> 40) % 200;
    uint64_t e = s + len;
    if (e > ix.seq_len) e = ix.seq_len;
    if (i == 0) e = ix.seq_len;
    const uint32_t c = (uint32_t)(x >> 33) & 3u;
    uint64_t ws, we;
    fm_get_neighbor(ix, s, e, c, &ws, &we);
    uint32_t gs, ge;
    fm32_get_neighbor(ix, (uint32_t)s, (uint32_t)e, c, &gs, &ge);
    const bool want_empty = ws > we, got_empty = gs > ge;
    if (want_empty != got_empty || (!want_empty && (ws != gs || we != ge))) atomicAdd(bad, 1u);
}
void launch_build_fm32(const DevIndex &ix, uint32_t *out, uint32_t n_blk, hipStream_t st) {
    hipLaunchKernelGGL(k_build_fm32, dim3((n_blk + 255) / 256), dim3(256), 0, st, ix, out, n_blk);
}
void launch_fm32_check(const DevIndex &ix, uint32_t n, uint32_t *bad, hipStream_t st) {
    hipLaunchKernelGGL(k_fm32_check, dim3((n + 255) / 256), dim3(256), 0, st, ix, n, bad);
}

// PoreModel::match_prob for all 1024 k-mers of each level (mapper.cpp:443-445)
__global__ void k_match_probs(DevIndex ix, uint32_t n, const float *levels, float *out) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const float level = levels[i];
    for (uint32_t k = threadIdx.x; k < (uint32_t)NKMER; k += blockDim.x) {
        const float d = __fsub_rn(level, ix.model[k]);
        const double q = -((double)d * (double)d) / (double)ix.model[NKMER + k];
        out[(size_t)i * NKMER + k] = (float)(q - (double)ix.model[2 * NKMER + k]);
    }
}

// Calibration of the HBM traffic counters (rocprofv3 FETCH_SIZE / WRITE_SIZE) in k_map's own access shape: one lane per
// 64-byte record, four 16-byte accesses per lane, records scattered by a multiplicative permutation over a buffer far larger
// than L2 + Infinity Cache.  The byte counts are known exactly: n_rec * 64 written, n_rec * 64 read.
__global__ void k_calib_write(uint4 *buf, uint64_t n_rec, uint64_t mult) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += stride) {
        const uint64_t r = (i * mult) % n_rec;
        uint4 *p = buf + r * 4;
        const uint32_t v = (uint32_t)i;
        p[0] = make_uint4(v, v + 1, v + 2, v + 3); p[1] = make_uint4(v, v, v, v); p[2] = make_uint4(v, 1, 2, 3); p[3] = make_uint4(3, 2, 1, v);
    }
}
__global__ void k_calib_read(const uint4 *buf, uint64_t n_rec, uint64_t mult, uint32_t *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += stride) {
        const uint64_t r = (i * mult) % n_rec;
        const uint4 *p = buf + r * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) *sink = acc;    // keeps the loads alive
}
// Dependent loads over a region, in k_map's shape: 4 096 single-wavefront workgroups, every lane its own chain of `steps` 16-byte loads, the
// next address hashed from the last load's data (whatever the region holds) -- what a round trip into THAT memory costs while the whole
// chip is chasing through it.  Diagnostics (unc_calib_chase, tools/dev/placement_probe.py): the two speeds of one library.
__global__ __launch_bounds__(64, 4) void k_calib_chase(const uint4 *buf, uint64_t n16, uint32_t steps, uint32_t *sink) {
    uint64_t x = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < steps; ++i) {
        const uint4 v = buf[x % n16];
        acc += v.x ^ v.w;
        x = (x ^ v.x ^ ((uint64_t)v.y << 32) ^ i) * 0xD6E8FEB86659FD93ull;
        x ^= x >> 29;
    }
    if (acc == 0x12345678u) sink[0] = acc;      // (keeps the loads alive)
}
void launch_calib_chase(const uint4 *buf, uint64_t n16, uint32_t waves, uint32_t steps, uint32_t *sink, hipStream_t st) {
    hipLaunchKernelGGL(k_calib_chase, dim3(waves), dim3(64), 0, st, buf, n16, steps, sink);
}
void launch_calib(uint4 *buf, uint64_t n_rec, int write, uint32_t *sink, hipStream_t st) {
    const uint64_t mult = 2654435761ull;    // odd, coprime with any power-of-two-free n_rec the host passes (n_rec is made odd)
    if (write) hipLaunchKernelGGL(k_calib_write, dim3(256 * 16), dim3(256), 0, st, buf, n_rec, mult);
    else hipLaunchKernelGGL(k_calib_read, dim3(256 * 16), dim3(256), 0, st, buf, n_rec, mult, sink);
}

void launch_kmer_ranges(const DevIndex &ix, uint64_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_kmer_ranges, dim3(NKMER / 64), dim3(64), 0, st, ix, out);
}
void launch_fm_neighbor(const DevIndex &ix, uint32_t n, const uint64_t *s, const uint64_t *e, const uint8_t *b, uint64_t *os,
                        uint64_t *oe, hipStream_t st) {
    hipLaunchKernelGGL(k_fm_neighbor, dim3((n + 63) / 64), dim3(64), 0, st, ix, n, s, e, b, os, oe);
}
void launch_self_align(const DevIndex &ix, const uint8_t *pac, uint32_t n, const uint64_t *pos, const uint64_t *remain, uint64_t *out,
                       uint32_t cap, uint32_t *out_len, hipStream_t st) {
    hipLaunchKernelGGL(k_self_align, dim3((n + 63) / 64), dim3(64), 0, st, ix, pac, n, pos, remain, out, cap, out_len);
}
void launch_dense_sa(const DevIndex &ix, uint64_t *out, hipStream_t st) {
    const uint64_t n = ix.seq_len + 1;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_dense_sa, dim3((unsigned)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, st, ix, out);
}
void launch_dense_sa_check(const DevIndex &ix, const uint64_t *dense, uint32_t n, uint32_t *bad, hipStream_t st) {
    hipLaunchKernelGGL(k_dense_sa_check, dim3((n + 255) / 256), dim3(256), 0, st, ix, dense, n, bad);
}
void launch_fm_sa(const DevIndex &ix, uint32_t n, const uint64_t *rows, uint64_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_fm_sa, dim3((n + 63) / 64), dim3(64), 0, st, ix, n, rows, out);
}
void launch_match_probs(const DevIndex &ix, uint32_t n, const float *levels, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_match_probs, dim3(n), dim3(64), 0, st, ix, n, levels, out);
}

}  // namespace unc
