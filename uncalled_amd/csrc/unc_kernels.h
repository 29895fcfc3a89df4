// Launch wrappers implemented next to the kernels (k_events.hip, k_map.hip, k_taps.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "unc_dev_types.h"

namespace unc {
void launch_events(const DevReads &rd, const unc_params_t &P, hipStream_t st, uint32_t reads_per_wave = 0);
void launch_rt_events(const int16_t *raw, const float *raw_pa, const RtChunkDesc *chunks, uint32_t n_chunks, RtChan *chans, float *norm_ring,
                      const unc_params_t &P, float tgt_mean, float tgt_stdv, unc_evt_info_t *info, uint32_t *ring0_out, hipStream_t st);
void launch_map(const DevIndex &ix, const DevScratch &sc, const DevReads &rd, const unc_params_t &P, DevResult *results,
                uint32_t *next_read, uint32_t max_steps, uint32_t resume, const uint32_t *slot_map, uint32_t grid, hipStream_t st,
                const DevPool &pool, const uint32_t *read_list = nullptr, unsigned long long *wave_ticks = nullptr,
                const DevSched *sched = nullptr, bool profile = false, const uint32_t *flags_in = nullptr, uint32_t *flags_out = nullptr,
                uint32_t team = 1);   // chunked path only: wavefronts per channel (1, 2 or 4)
void launch_sched_init(const DevSched &S, hipStream_t st);
void launch_xcd_probe(uint32_t *out, uint32_t n_blocks, hipStream_t st);
void launch_pool_init(const DevPool &B, hipStream_t st);
uint32_t map_kernel_waves_per_cu();
int map_kernel_attributes(bool narrow, bool profile, uint32_t *out4);   // VGPRs, scratch bytes / lane, LDS bytes, max threads
void launch_kmer_ranges(const DevIndex &ix, uint64_t *out2048, hipStream_t st);
void launch_fm_neighbor(const DevIndex &ix, uint32_t n, const uint64_t *s, const uint64_t *e, const uint8_t *b, uint64_t *os,
                        uint64_t *oe, hipStream_t st);
void launch_self_align(const DevIndex &ix, const uint8_t *pac, uint32_t n, const uint64_t *pos, const uint64_t *remain, uint64_t *out,
                       uint32_t cap, uint32_t *out_len, hipStream_t st);
void launch_dense_sa(const DevIndex &ix, uint64_t *out, hipStream_t st);
void launch_build_fm32(const DevIndex &ix, uint32_t *out, uint32_t n_blk, hipStream_t st);
void launch_fm32_check(const DevIndex &ix, uint32_t n, uint32_t *bad, hipStream_t st);
void launch_dense_sa_check(const DevIndex &ix, const uint64_t *dense, uint32_t n, uint32_t *bad, hipStream_t st);
void launch_fm_sa(const DevIndex &ix, uint32_t n, const uint64_t *rows, uint64_t *out, hipStream_t st);
void launch_calib(uint4 *buf, uint64_t n_rec, int write, uint32_t *sink, hipStream_t st);
void launch_calib_chase(const uint4 *buf, uint64_t n16, uint32_t waves, uint32_t steps, uint32_t *sink, hipStream_t st);
void launch_rsort_pass(const uint64_t *kin, const uint64_t *vin, uint64_t *kout, uint64_t *vout, uint64_t n, uint32_t shift, uint32_t iota,
                       uint32_t *counts, uint32_t *sums, hipStream_t st);      // k_sort.hip: one digit of the LSD radix sort
uint32_t rsort_tile();
// k_sort.hip: the steps of the prefix-doubling suffix sort between the radix sorts (unc_build_suffix_array)
void launch_sa_first_key(const uint8_t *text, uint64_t n, uint64_t *key, hipStream_t st);
void launch_sa_ranks(const uint64_t *sk, const uint64_t *order, uint64_t n, uint32_t *flags, uint32_t *sums, uint32_t *rank, uint32_t *n_groups,
                     hipStream_t st);
void launch_sa_next_key(const uint32_t *rank, uint64_t n, uint64_t k, uint64_t *key, hipStream_t st);
void launch_sa_invert(const uint32_t *rank, uint64_t n, int64_t *sa, hipStream_t st);
void launch_match_probs(const DevIndex &ix, uint32_t n, const float *levels, float *out, hipStream_t st);
}  // namespace unc
