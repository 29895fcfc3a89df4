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

}  // namespace unc
