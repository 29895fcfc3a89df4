// k_sort: LSD radix sort of (64-bit key, 64-bit value) pairs on the device, 8 bits per pass -- the suffix sorts of the BWA-format
// index builders (uncalled_amd/build_index.py; SURVEY 8f-3: the replacement for bwa_idx_build, bwa_index.hpp:92-101).  Rounds 2-4
// sorted with torch.sort (rocPRIM); this is the hand-written path.  One WAVEFRONT per tile of 2 048 pairs, three steps per digit:
//   k_rsort_hist     a tile's 256-bin histogram in LDS -> counts[digit][tile]
//   scan             exclusive prefix sum over counts in (digit, tile) order: where each tile's keys of each digit go
//                    (k_scan_sums / k_scan_top / k_scan_apply: sums of 2 048-element chunks, their scan by one wavefront, the chunks again)
//   k_rsort_scatter  the tile again, 64 keys per round IN ORDER: a key's rank among the round's keys of its digit by eight ballots
//                    (stable: earlier lanes first), its position = the digit's running offset + rank
// HBM-bound integer work (40 bytes per pair and pass, the scatter's 8-byte stores uncoalesced): no LDS tiling beyond the histogram,
// no MFMA.  Stable, so that a pass leaves the order of the passes before it intact.
#include <hip/hip_runtime.h>

#include "unc_dev_types.h"
#include "unc_kernels.h"
#include "wave_prims.h"

namespace unc {

constexpr uint32_t RS_ROUNDS = 32, RS_TILE = (uint32_t)WAVE * RS_ROUNDS, RS_BINS = 256;

__global__ __launch_bounds__(64) void k_rsort_hist(const uint64_t *keys, uint64_t n, uint32_t shift, uint32_t *counts, uint32_t ntiles) {
    __shared__ uint32_t h[RS_BINS];
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t b = lane; b < RS_BINS; b += WAVE) h[b] = 0;
    wave_sync();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint64_t i = base + (uint64_t)r * WAVE + lane;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (RS_BINS - 1u)], 1u);
    }
    wave_sync();
    for (uint32_t b = lane; b < RS_BINS; b += WAVE) counts[(uint64_t)b * ntiles + blockIdx.x] = h[b];
}

// ---- exclusive prefix sum of m 32-bit counts, in place (m up to 2^26: 2 048 x 2 048 x 16)
__global__ __launch_bounds__(64) void k_scan_sums(const uint32_t *v, uint64_t m, uint32_t *sums) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint64_t i = base + (uint64_t)r * WAVE + lane;
        acc += i < m ? v[i] : 0u;
    }
    uint32_t tot;
    (void)excl_sum32(acc, &tot);
    if (lane == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(64) void k_scan_top(uint32_t *sums, uint32_t nsums) {      // one wavefront
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nsums; c0 += WAVE) {
        const uint32_t i = c0 + lane;
        const uint32_t x = i < nsums ? sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = excl_sum32(x, &tot);
        if (i < nsums) sums[i] = carry + ex;
        carry += tot;
    }
}
__global__ __launch_bounds__(64) void k_scan_apply(uint32_t *v, uint64_t m, const uint32_t *sums) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    uint32_t carry = sums[blockIdx.x];
    for (uint32_t r = 0; r < RS_ROUNDS; ++r) {
        const uint64_t i = base + (uint64_t)r * WAVE + lane;
        const uint32_t x = i < m ? v[i] : 0u;
        uint32_t tot;
        const uint32_t ex = excl_sum32(x, &tot);
        if (i < m) v[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(64) void k_rsort_scatter(const uint64_t *kin, const uint64_t *vin, uint64_t *kout, uint64_t *vout, uint64_t n,
                                                      uint32_t shift, const uint32_t *offs, uint32_t ntiles, uint32_t iota) {
    __shared__ uint32_t pos[RS_BINS];
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t b = lane; b < RS_BINS; b += WAVE) pos[b] = offs[(uint64_t)b * ntiles + blockIdx.x];
    wave_sync();
    const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
    const uint64_t below = lanemask_lt();
    for (uint32_t r = 0; r < RS_ROUNDS && base + (uint64_t)r * WAVE < n; ++r) {
        const uint64_t i = base + (uint64_t)r * WAVE + lane;
        const bool have = i < n;
        const uint64_t key = have ? kin[i] : 0ull;
        const uint64_t val = iota ? i : (have ? vin[i] : 0ull);       // iota: the values are the positions (an argsort's first pass)
        const uint32_t d = (uint32_t)(key >> shift) & (RS_BINS - 1u);
        uint64_t peers = __ballot(have);                              // lanes of this round with my digit
#pragma unroll
        for (uint32_t b = 0; b < 8; ++b) {
            const uint64_t bb = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below), cnt = (uint32_t)__popcll(peers);
        const uint32_t p = have ? pos[d] : 0u;                        // (every lane reads before the group's first lane moves the offset on)
        wave_sync();
        if (have && rank == 0) pos[d] = p + cnt;
        wave_sync();
        if (have) { kout[(uint64_t)p + rank] = key; vout[(uint64_t)p + rank] = val; }
    }
}

// one pass (digit at `shift`) of the sort: (kin, vin) -> (kout, vout); counts: 256 * ntiles words, sums: ceil(256 * ntiles / 2048) words
void launch_rsort_pass(const uint64_t *kin, const uint64_t *vin, uint64_t *kout, uint64_t *vout, uint64_t n, uint32_t shift, uint32_t iota,
                       uint32_t *counts, uint32_t *sums, hipStream_t st) {
    const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const uint64_t m = (uint64_t)RS_BINS * ntiles;
    const uint32_t nsums = (uint32_t)((m + RS_TILE - 1) / RS_TILE);
    hipLaunchKernelGGL(k_rsort_hist, dim3(ntiles), dim3(WAVE), 0, st, kin, n, shift, counts, ntiles);
    hipLaunchKernelGGL(k_scan_sums, dim3(nsums), dim3(WAVE), 0, st, (const uint32_t *)counts, m, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(WAVE), 0, st, sums, nsums);
    hipLaunchKernelGGL(k_scan_apply, dim3(nsums), dim3(WAVE), 0, st, counts, m, (const uint32_t *)sums);
    hipLaunchKernelGGL(k_rsort_scatter, dim3(ntiles), dim3(WAVE), 0, st, kin, vin, kout, vout, n, shift, (const uint32_t *)counts, ntiles, iota);
}
uint32_t rsort_tile() { return RS_TILE; }

// ---- the suffix array of a text of n < 2^31 symbols by prefix doubling, everything between the radix sorts (unc_build_suffix_array:
// the whole of what bwa_idx_build's suffix sort does for the index builder, bwa_index.hpp:92-101, without torch).  Rounds 3-5 ran these
// steps as torch tensor ops around the hand-written sort.  All of them are one pass over n elements, HBM-bound:
//   k_sa_first_key   the first 21 symbols of every suffix as a 63-bit key (symbol + 1 in 3 bits, 0 past the end)
//   k_sa_flags       1 where a sorted key differs from its predecessor (a new group of equal prefixes)
//   (exclusive scan of the flags: k_scan_sums / _top / _apply above)
//   k_sa_ranks       rank[suffix] = groups before it in sorted order; the number of groups
//   k_sa_next_key    key = rank[i] * (n + 1) + (rank[i + k] + 1, 0 past the end): suffixes by their first 2k symbols
//   k_sa_invert      sa[rank[i]] = i once every suffix has a rank of its own
__global__ void k_sa_first_key(const uint8_t *text, uint64_t n, uint64_t *key) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
    for (uint32_t j = 0; j < 21; ++j) k = (k << 3) | (i + j < n ? (uint64_t)text[i + j] + 1u : 0u);
    key[i] = k;
}
__global__ void k_sa_flags(const uint64_t *sk, uint64_t n, uint32_t *flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1u : 0u;
}
__global__ void k_sa_ranks(const uint64_t *sk, const uint64_t *order, const uint32_t *before, uint64_t n, uint32_t *rank, uint32_t *n_groups) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t head = (i == 0 || sk[i] != sk[i - 1]) ? 1u : 0u;
    const uint32_t r = before[i] + head - 1u;          // groups up to and including mine, minus one
    rank[order[i]] = r;
    if (i == n - 1) *n_groups = r + 1u;
}
__global__ void k_sa_next_key(const uint32_t *rank, uint64_t n, uint64_t k, uint64_t *key) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t nxt = i + k < n ? (uint64_t)rank[i + k] + 1u : 0u;
    key[i] = (uint64_t)rank[i] * (n + 1u) + nxt;        // < 2^62 for n < 2^31
}
__global__ void k_sa_invert(const uint32_t *rank, uint64_t n, int64_t *sa) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sa[rank[i]] = (int64_t)i;
}
void launch_sa_first_key(const uint8_t *text, uint64_t n, uint64_t *key, hipStream_t st) {
    hipLaunchKernelGGL(k_sa_first_key, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, text, n, key);
}
// sorted keys + their order -> rank per suffix and the number of groups; flags: n words of scratch, sums: ceil(n / 2048) words
void launch_sa_ranks(const uint64_t *sk, const uint64_t *order, uint64_t n, uint32_t *flags, uint32_t *sums, uint32_t *rank, uint32_t *n_groups,
                     hipStream_t st) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const uint32_t nsums = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    hipLaunchKernelGGL(k_sa_flags, dim3(blocks), dim3(256), 0, st, sk, n, flags);
    hipLaunchKernelGGL(k_scan_sums, dim3(nsums), dim3(WAVE), 0, st, (const uint32_t *)flags, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(WAVE), 0, st, sums, nsums);
    hipLaunchKernelGGL(k_scan_apply, dim3(nsums), dim3(WAVE), 0, st, flags, n, (const uint32_t *)sums);
    hipLaunchKernelGGL(k_sa_ranks, dim3(blocks), dim3(256), 0, st, sk, order, (const uint32_t *)flags, n, rank, n_groups);
}
void launch_sa_next_key(const uint32_t *rank, uint64_t n, uint64_t k, uint64_t *key, hipStream_t st) {
    hipLaunchKernelGGL(k_sa_next_key, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rank, n, k, key);
}
void launch_sa_invert(const uint32_t *rank, uint64_t n, int64_t *sa, hipStream_t st) {
    hipLaunchKernelGGL(k_sa_invert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rank, n, sa);
}

}  // namespace unc
