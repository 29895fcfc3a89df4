// Event detection + whole-read normalisation.  Replaces, for a batch of reads:
//   ReadBuffer calibration            read_buffer.cpp:239-241   (u16 reinterpretation included)
//   EventDetector::get_means          event_detector.cpp:133-145 (add_sample 83-112, compute_tstat
//                                     174-219, peak_detect 221-279, create_event 296-319)
//   Normalizer::set_signal / at       normalizer.cpp:31-44,114-118 (scale/shift only; the affine map
//                                     itself is applied per event in k_map)
// Work decomposition (k_events): the detector is a strictly sequential machine per read -- two peak-detector FSMs that
// consume the two t-statistics of EVERY sample and whose state after a sample depends on all earlier ones -- so one
// lane follows one read, and the double cumulative sums are added in exactly the reference's order (no exactness
// argument needed).  What is parallel inside a read, the t-statistics, is exposed as instruction-level parallelism: a
// lane takes its samples eight at a time (one aligned 16-byte load), keeps the 20 cumulative sums the eight window
// pairs need in registers, computes the sixteen t-statistics independently and only then runs the sixteen FSM steps.
// A wavefront carries `reads_per_wave` reads (64, or fewer so that a batch spreads over more wavefronts).
// (k_rt_events below, one chunk per lane, keeps its 16-slot ring of sums in LDS.)
// All float expressions are written one IEEE operation at a time and the file is compiled with
// -ffp-contract=off: the reference is built without FMA (setup.py:121).
#include <hip/hip_runtime.h>
#include <float.h>

#include "unc_dev_types.h"
#include "wave_prims.h"

namespace unc {

constexpr int RING = 16;   // >= 13 = 1 + 2*window_length2 (event_detector.cpp:30); power of two

struct Detector {
    float threshold;
    uint32_t window_length, masked_to;
    int32_t peak_pos;
    float peak_value;
    bool valid_peak;
};

// event_detector.cpp:221-279 as straight-line code: 64 lanes follow 64 reads whose detectors are in different states at every
// sample, so every branch of the written-out form below is taken by some lane and the wavefront pays for all of them plus the
// exec-mask bookkeeping in between (round 3, profiles/r03_ab_k_events.log: 31.6 -> 27.4 ms per 50 k reads); here both arms are computed and selected (the arithmetic is the same single operations:
// one float subtraction per comparison against peak_height, no contraction).  `long_det` is the long detector when `det` is
// the short one.
__device__ __forceinline__ bool peak_detect(Detector &det, Detector *long_det, float cur, uint32_t buf_mid, float peak_height) {
    const bool act = !(det.masked_to >= buf_mid);
    const bool idle = det.peak_pos == -1;                       // no candidate yet: following the minimum
    const float pv = det.peak_value;
    // -- idle arm
    const bool a_lt = cur < pv;
    const bool a_rise = !a_lt && (__fsub_rn(cur, pv) > peak_height);
    const float a_pv = (a_lt || a_rise) ? cur : pv;
    const int32_t a_pos = a_rise ? (int32_t)buf_mid : -1;
    // -- tracking arm
    const bool b_gt = cur > pv;
    const float b_pv = b_gt ? cur : pv;
    const int32_t b_pos = b_gt ? (int32_t)buf_mid : det.peak_pos;
    const bool b_over = b_pv > det.threshold;
    const bool b_valid = det.valid_peak || ((__fsub_rn(b_pv, cur) > peak_height) && b_over);
    const bool b_emit = b_valid && ((buf_mid - (uint32_t)b_pos) > det.window_length / 2);
    const bool trk = act && !idle;
    if (long_det != nullptr) {
        const bool m = trk && b_over;
        long_det->masked_to = m ? (uint32_t)b_pos + det.window_length : long_det->masked_to;
        long_det->peak_pos = m ? -1 : long_det->peak_pos;
        long_det->peak_value = m ? FLT_MAX : long_det->peak_value;
        long_det->valid_peak = m ? false : long_det->valid_peak;
    }
    const float n_pv = idle ? a_pv : (b_emit ? cur : b_pv);
    const int32_t n_pos = idle ? a_pos : (b_emit ? -1 : b_pos);
    const bool n_valid = idle ? det.valid_peak : (b_emit ? false : b_valid);
    det.peak_value = act ? n_pv : pv;
    det.peak_pos = act ? n_pos : det.peak_pos;
    det.valid_peak = act ? n_valid : det.valid_peak;
    return trk && b_emit;
}

// sqrt of a float, CORRECTLY ROUNDED (the reference's sqrt resolves to the float overload = one sqrtss, event_detector.cpp:218).
// hipcc's __fsqrt_rn is NOT that unless OCML_BASIC_ROUNDED_OPERATIONS is defined: it is __ocml_native_sqrt_f32, the bare v_sqrt_f32 with
// its 1 ulp of error (clang's __clang_hip_math.h) -- rounds 1-4 used it, and one read in seven thousand came out with an event more or
// less than the reference detects (found by the 10 240-read parity sweeps of round 5: tests/dev/parity_sweep.py; pinned by
// tests/golden/sweep_reads_r05.npz).  sqrtf() is lowered to v_sqrt_f32 + a two-FMA correction that is exact.
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }

// ---- x / 3 and x / 6, correctly rounded, in three operations instead of the division sequence (a dozen dependent FP64
// operations for a double).  With y = RN(1 / d):  q = RN(x * y);  r = x - d * q (one FMA: exact, q is within 2 ulp of x / d and r a
// small multiple of ulp(q));  q' = RN(q + r * y).  The real number q + r * y differs from x / d by |r * y * eps| < 2^-50 ulp, and
// for d = 3 * 2^k the quotient of a p-bit significand m is never closer than 1/6 ulp to a rounding boundary (2m - 6k - 3 is an
// odd integer), and equally far from the next representable number unless it is one -- so the last rounding lands on RN(x / d),
// the value the reference's division instruction (built without FMA contraction, setup.py:121) produces.  Outside the range
// where neither q nor r can leave the normal numbers (zero, subnormal quotients, infinities, NaN) the division itself is
// used; tests/dev/check_div_const.c checks the float version over every float and the double version on 2^33 samples.
#ifndef UNC_EXACT_DIV_CONST
#define UNC_EXACT_DIV_CONST 1
#endif
// FAST = the three-operation form, unconditionally (straight-line code: the sixteen t-statistics of a block interleave);
// `bad` collects whether any operand was outside the range the argument covers -- the caller then recomputes with FAST = false.
template <uint32_t W, bool FAST> __device__ __forceinline__ double div_w(double x, bool &bad) {
    static_assert(W == 3 || W == 6, "the argument above is for 3 * 2^k");
    if constexpr (FAST) {
        const double ax = fabs(x);
        bad |= !(ax >= 0x1p-900 && ax <= 0x1p900);
        constexpr double d = (double)W, y = 1.0 / (double)W;
        const double q = __dmul_rn(x, y);
        const double r = __fma_rn(-d, q, x);
        return __fma_rn(r, y, q);
    } else {
        return x / (double)W;
    }
}
template <uint32_t W, bool FAST> __device__ __forceinline__ float div_w(float x, bool &bad) {
    static_assert(W == 3 || W == 6, "the argument above is for 3 * 2^k");
    if constexpr (FAST) {
        const float ax = fabsf(x);
        bad |= !(ax >= 0x1p-100f && ax <= 0x1p100f);
        constexpr float d = (float)W, y = 1.0f / (float)W;
        const float q = __fmul_rn(x, y);
        const float r = __fmaf_rn(-d, q, x);
        return __fmaf_rn(r, y, q);
    } else {
        return __fdiv_rn(x, (float)W);
    }
}
constexpr bool EV_FAST_DIV = UNC_EXACT_DIV_CONST != 0;

// event_detector.cpp:174-219 with the ring addressed by ABSOLUTE position (slot = pos & 15).
// `st_pos` is the position the reference's `(buf_mid - w) % 13` lands on (it wraps for
// buf_mid < w, where slot (2^32 + buf_mid - w) % 13 == buf_mid + 6 still holds C[buf_mid + 6]).
template <uint32_t W, bool FAST>
__device__ __forceinline__ float tstat_ring(const double *sum, const double *sumsq, int lane, uint32_t buf_mid, bool &bad) {
    constexpr uint32_t w = W;
    uint32_t st_pos = buf_mid >= w ? buf_mid - w : buf_mid + 6;
    uint32_t i = (buf_mid & (RING - 1)) * WAVE + lane, st = (st_pos & (RING - 1)) * WAVE + lane,
             en = ((buf_mid + w) & (RING - 1)) * WAVE + lane;
    double sum1 = sum[i] - sum[st];
    double sumsq1 = sumsq[i] - sumsq[st];
    float sum2 = (float)(sum[en] - sum[i]);
    float sumsq2 = (float)(sumsq[en] - sumsq[i]);
    float mean1 = (float)div_w<W, FAST>(sum1, bad);
    float mean2 = div_w<W, FAST>(sum2, bad);
    float m1sq = __fmul_rn(mean1, mean1), q2 = div_w<W, FAST>(sumsq2, bad), m2sq = __fmul_rn(mean2, mean2);
    float combined_var = (float)(((div_w<W, FAST>(sumsq1, bad) - (double)m1sq) + (double)q2) - (double)m2sq);
    combined_var = fmaxf(combined_var, FLT_MIN);
    float delta_mean = __fsub_rn(mean2, mean1);
    return __fdiv_rn(fabsf(delta_mean), sqrt_rn(div_w<W, FAST>(combined_var, bad)));
}
template <uint32_t W>
__device__ __forceinline__ float tstat(const double *sum, const double *sumsq, int lane, uint32_t t, uint32_t buf_mid) {
    if (t <= 2 * W) return 0.0f;
    bool bad = false;
    float v = tstat_ring<W, EV_FAST_DIV>(sum, sumsq, lane, buf_mid, bad);
    if (bad) v = tstat_ring<W, false>(sum, sumsq, lane, buf_mid, bad);
    return v;
}

// ---- k_events: register-window detector ----------------------------------------------------------------------------
constexpr int EB = 8;                 // samples per block of the fast path
constexpr int WIN = 12 + EB;          // window of cumulative sums: positions base-11 .. base+EB (13 = the reference's ring)

// compute_tstat (event_detector.cpp:174-219) for the sample whose newest cumulative sum sits at window index `ni`:
// buf_mid = ni - 6, the two windows end / start there.  `st` = index of C[buf_mid - w] (the caller resolves the
// reference's wrap for the first samples).
template <uint32_t W, bool FAST>
__device__ __forceinline__ float tstat_win(const double *C, const double *Q, int ni, int st, bool &bad) {
    const int i = ni - 6, en = i + (int)W;
    double sum1 = C[i] - C[st];
    double sumsq1 = Q[i] - Q[st];
    float sum2 = (float)(C[en] - C[i]);
    float sumsq2 = (float)(Q[en] - Q[i]);
    float mean1 = (float)div_w<W, FAST>(sum1, bad);
    float mean2 = div_w<W, FAST>(sum2, bad);
    float m1sq = __fmul_rn(mean1, mean1), q2 = div_w<W, FAST>(sumsq2, bad), m2sq = __fmul_rn(mean2, mean2);
    float combined_var = (float)(((div_w<W, FAST>(sumsq1, bad) - (double)m1sq) + (double)q2) - (double)m2sq);
    combined_var = fmaxf(combined_var, FLT_MIN);
    float delta_mean = __fsub_rn(mean2, mean1);
    return __fdiv_rn(fabsf(delta_mean), sqrt_rn(div_w<W, FAST>(combined_var, bad)));
}

struct EvRun {                        // what EventDetector carries from sample to sample
    Detector sd, ld;
    uint32_t t, evt_st, total_events, n_kept, over;      // over: an event found the read's room in `means` full (reported, never silent)
    double evt_st_sum, mean_sum;
    float len_sum;
};

// both peak detectors on one sample's t-statistics, then create_event (event_detector.cpp:101-110,296-319); csum = C[evt_en]
__device__ __forceinline__ void fsm_step(EvRun &E, const unc_params_t &P, float t1, float t2, uint32_t buf_mid, double csum, float *means,
                                         uint32_t mcap) {
    const bool p1 = peak_detect(E.sd, &E.ld, t1, buf_mid, P.peak_height);
    const bool p2 = peak_detect(E.ld, nullptr, t2, buf_mid, P.peak_height);
    if (p1 || p2) {
        const uint32_t evt_en = buf_mid - UNC_WINDOW1 + 1;
        const uint32_t length = (uint32_t)(float)(evt_en - E.evt_st);
        float mean = (float)((csum - E.evt_st_sum) / (double)length);
        E.evt_st = evt_en;
        E.evt_st_sum = csum;
        E.len_sum = __fadd_rn(E.len_sum, (float)length);
        E.total_events++;
        mean = __fmul_rn(__fadd_rn(mean, 0.0f), 1.0f);   // calibrate(): cal_offset_=0, cal_coef_=1
        if (mean >= P.min_mean && mean <= P.max_mean) {
            if (E.n_kept < mcap) {
                means[E.n_kept++] = mean;
                E.mean_sum += (double)mean;              // Normalizer::set_signal's first sum, in index order
            } else E.over = 1u;
        }
    }
}

__global__ __launch_bounds__(64) void k_events(DevReads R, unc_params_t P, uint32_t reads_per_wave) {
    const int lane = lane_id();
    const uint32_t r = blockIdx.x * reads_per_wave + (uint32_t)lane;
    const bool active = (uint32_t)lane < reads_per_wave && r < R.n_reads;
    if (!active) return;              // no collectives in this kernel

    const uint64_t off = R.offsets[r], n = R.offsets[r + 1] - off, moff = R.moff[r];
    const uint32_t mcap = (uint32_t)(R.moff[r + 1] - moff);
    const float cal_range = R.calib[r].range, cal_offset = R.calib[r].offset, cal_digit = R.calib[r].digitisation;
    const int16_t *raw = R.raw + off;
    float *means = R.means + moff;

    // EventDetector::reset, event_detector.cpp:47-77
    EvRun E;
    E.sd = Detector{P.threshold1, P.window_length1, 0u, -1, FLT_MAX, false};
    E.ld = Detector{P.threshold2, P.window_length2, 0u, -1, FLT_MAX, false};
    E.t = 1; E.evt_st = 0; E.total_events = 0; E.n_kept = 0; E.over = 0; E.evt_st_sum = 0.0; E.mean_sum = 0.0; E.len_sum = 0.0f;
    double C[WIN], Q[WIN];            // C[i] = cumulative sum at position (next sample's position) - 12 + i
#pragma unroll
    for (int i = 0; i < WIN; ++i) { C[i] = 0.0; Q[i] = 0.0; }

    // calibration: u16 reinterpretation of the stored i16, three float roundings
    auto calibrate = [&](int16_t v) {
        return __fdiv_rn(__fmul_rn(cal_range, __fadd_rn((float)(int)(uint16_t)v, cal_offset)), cal_digit);
    };
    // one sample, every special case of the first samples included (the u32 wrap of buf_mid, `t <= 2w`, and the slot the
    // reference's `(buf_mid - w) % 13` lands on while buf_mid < w: the newest sum)
    auto step1 = [&](float s) {
        C[12] = C[11] + (double)s;
        Q[12] = Q[11] + (double)__fmul_rn(s, s);
        E.t++;
        const uint32_t buf_mid = E.t - 7;
        bool bad = false;      // (head and tail samples: the division itself)
        const float t1 = E.t <= 2 * UNC_WINDOW1 ? 0.0f : tstat_win<UNC_WINDOW1, false>(C, Q, 12, buf_mid >= UNC_WINDOW1 ? 6 - UNC_WINDOW1 : 12, bad);
        const float t2 = E.t <= 2 * UNC_WINDOW2 ? 0.0f : tstat_win<UNC_WINDOW2, false>(C, Q, 12, buf_mid >= UNC_WINDOW2 ? 6 - UNC_WINDOW2 : 12, bad);
        fsm_step(E, P, t1, t2, buf_mid, C[4], means, mcap);
#pragma unroll
        for (int i = 0; i < 12; ++i) { C[i] = C[i + 1]; Q[i] = Q[i + 1]; }
    };

    uint64_t k = 0;
    // head: the first 16 samples (special cases) and up to the first 16-byte boundary of this read's samples
    const uint64_t misalign = (8u - (uint32_t)(((uintptr_t)raw >> 1) & 7u)) & 7u;     // samples until raw + k is 16-byte aligned
    uint64_t head = 16 + misalign;                                                    // 16 is a multiple of EB
    if (head > n) head = n;
    for (; k < head; ++k) step1(calibrate(raw[k]));
    // middle: EB samples per iteration, everything about them that does not depend on the detectors first
    for (; k + EB <= n; k += EB) {
        const uint4 q = *reinterpret_cast<const uint4 *>(raw + k);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float t1[EB], t2[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const float s = calibrate((int16_t)(uint16_t)(w[j >> 1] >> ((j & 1) << 4)));
            C[12 + j] = C[11 + j] + (double)s;
            Q[12 + j] = Q[11 + j] + (double)__fmul_rn(s, s);
        }
        bool bad = false;
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            t1[j] = tstat_win<UNC_WINDOW1, EV_FAST_DIV>(C, Q, 12 + j, 6 + j - UNC_WINDOW1, bad);
            t2[j] = tstat_win<UNC_WINDOW2, EV_FAST_DIV>(C, Q, 12 + j, 6 + j - UNC_WINDOW2, bad);
        }
        if (bad) {       // an operand outside the range of the short division (a flat-zero window, say): the block again, dividing
#pragma unroll      // (static indices: C / Q / t1 / t2 stay in registers)
            for (int j = 0; j < EB; ++j) {
                t1[j] = tstat_win<UNC_WINDOW1, false>(C, Q, 12 + j, 6 + j - UNC_WINDOW1, bad);
                t2[j] = tstat_win<UNC_WINDOW2, false>(C, Q, 12 + j, 6 + j - UNC_WINDOW2, bad);
            }
        }
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            E.t++;
            fsm_step(E, P, t1[j], t2[j], E.t - 7, C[4 + j], means, mcap);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) { C[i] = C[i + EB]; Q[i] = Q[i + EB]; }
    }
    for (; k < n; ++k) step1(calibrate(raw[k]));     // tail

    // Normalizer::set_signal: sequential double sums in index order, then scale/shift (::at)
    float scale = 0.0f, shift = 0.0f;
    const uint32_t n_kept = E.n_kept;
    if (n_kept > 0) {
        const double mean = E.mean_sum / (double)n_kept;
        double varsum = 0.0;
        for (uint32_t i = 0; i < n_kept; ++i) {
            double e = (double)means[i] - mean;
            varsum += e * e;
        }
        const float tgt_mean = R.tgt_mean, tgt_stdv = R.tgt_stdv;
        scale = (float)((double)tgt_stdv / sqrt(varsum / (double)n_kept));
        shift = (float)((double)tgt_mean - (double)scale * mean);
    }
    unc_evt_info_t inf;
    inf.n_events = n_kept;
    inf.total_events = E.total_events;
    inf.len_sum = E.len_sum;
    inf.scale = scale;
    inf.shift = shift;
    inf.pad = E.over;        // 1: the read has more events than its room in `means` (the host fails the call loudly)
    R.info[r] = inf;
}

}  // namespace unc

namespace unc {
// ---------------------------------------------------------------------------------------------------
// Chunked (realtime / MAP_ORD) path: Mapper::process_chunk (mapper.cpp:307-367) for one chunk per channel,
// one lane per channel, with the per-channel state a Mapper keeps between chunks held in HBM (RtChan):
// the streaming detector, the 25-event EventProfiler window (event_profiler.hpp:71-104: events are released
// 24 late and masked while the window's stdv < 5) and the rolling 6000-event Normalizer
// (normalizer.cpp:46-75), which deliberately survives across reads (mapper.cpp:225-226).
__device__ __forceinline__ bool roll_push(float *signal, uint32_t size, double &mean, double &varsum, uint32_t &n, uint32_t &rd,
                                          uint32_t &wr, uint32_t &full, float newevt) {
    if (full) return false;
    const double oldevt = (double)signal[wr];
    signal[wr] = newevt;
    if (n == size) {
        const double oldmean = mean;
        mean += ((double)newevt - oldevt) / (double)size;
        varsum += ((double)newevt + oldevt - oldmean - mean) * ((double)newevt - oldevt);
    } else {
        n++;
        const double dt1 = (double)newevt - mean;
        mean += dt1 / (double)n;
        const double dt2 = (double)newevt - mean;
        varsum += dt1 * dt2;
    }
    wr = wr + 1 == size ? 0 : wr + 1;
    full = wr == rd ? 1u : 0u;
    return true;
}
__device__ __forceinline__ uint32_t roll_unread(uint32_t n, uint32_t rd, uint32_t wr) {
    return rd < wr ? wr - rd : (n - rd) + wr;   // Normalizer::unread_size, normalizer.cpp:131-134
}

// lanes_per_wave: how many lanes of a wavefront follow a channel.  The lanes of a wavefront are different channels in different
// detector states, so whenever ANY of them closes an event the whole wavefront walks the event path (event, profiler window, rolling
// normaliser: two thirds of an iteration's instructions); with 64 channels per wavefront that is every sample.  A flow cell's 512
// channels do not fill the chip anyway: one channel per wavefront pays the event path on one sample in nine.
__global__ __launch_bounds__(64) void k_rt_events(const int16_t *raw, const float *raw_pa, const RtChunkDesc *chunks, uint32_t n_chunks, RtChan *chans,
                                                  float *norm_ring, unc_params_t P, float tgt_mean, float tgt_stdv,
                                                  unc_evt_info_t *info, uint32_t *ring0_out, uint32_t lanes_per_wave) {
    __shared__ double s_sum[RING * WAVE];
    __shared__ double s_sumsq[RING * WAVE];
    const int lane = lane_id();
    const uint32_t ci = blockIdx.x * lanes_per_wave + (uint32_t)lane;
    if ((uint32_t)lane >= lanes_per_wave || ci >= n_chunks) return;   // no collectives in this kernel
    const RtChunkDesc cd = chunks[ci];
    RtChan *C = chans + cd.channel;
    float *ring = norm_ring + (size_t)cd.channel * NORM_LEN;

    // ---- load (or reset) the per-channel state
    uint32_t t, evt_st, total_events, n_pushed, ring0, status = 0;
    double evt_st_sum, evt_st_sumsq;
    float len_sum;
    Detector sd, ld;
    double n_mean = C->n_mean, n_varsum = C->n_varsum;
    uint32_t n_n = C->n_n, n_rd = C->n_rd, n_wr = C->n_wr, n_full = C->n_full;
    double pw_mean, pw_varsum;
    uint32_t pw_n, pw_rd, pw_wr, pw_full, q_head, q_len, prof_full, to_mask;
    if (cd.new_read) {
        // Mapper::reset (mapper.cpp:219-246): evdt_.reset(), evt_prof_.reset(), norm_.skip_unread()
        s_sum[lane] = 0.0; s_sumsq[lane] = 0.0;
        t = 1; evt_st = 0; total_events = 0; evt_st_sum = evt_st_sumsq = 0.0; len_sum = 0.0f;
        sd = Detector{P.threshold1, P.window_length1, 0u, -1, FLT_MAX, false};
        ld = Detector{P.threshold2, P.window_length2, 0u, -1, FLT_MAX, false};
        pw_mean = pw_varsum = 0.0; pw_n = pw_rd = pw_wr = pw_full = 0; q_head = q_len = 0; prof_full = 0; to_mask = 0;
        C->pw_signal[0] = 0.0f;
        n_rd = n_wr;            // skip_unread(0): drop whatever the previous read left unread
        n_full = 0;
        ring0 = n_wr;
        n_pushed = 0;
    } else {
#pragma unroll
        for (int i = 0; i < RING; ++i) { s_sum[i * WAVE + lane] = C->sum[i]; s_sumsq[i * WAVE + lane] = C->sumsq[i]; }
        t = C->t; evt_st = C->evt_st; total_events = C->total_events; evt_st_sum = C->evt_st_sum; evt_st_sumsq = C->evt_st_sumsq;
        len_sum = C->len_sum;
        sd = Detector{C->sd.threshold, C->sd.window_length, C->sd.masked_to, C->sd.peak_pos, C->sd.peak_value, C->sd.valid_peak != 0};
        ld = Detector{C->ld.threshold, C->ld.window_length, C->ld.masked_to, C->ld.peak_pos, C->ld.peak_value, C->ld.valid_peak != 0};
        pw_mean = C->pw_mean; pw_varsum = C->pw_varsum; pw_n = C->pw_n; pw_rd = C->pw_rd; pw_wr = C->pw_wr; pw_full = C->pw_full;
        q_head = C->q_head; q_len = C->q_len; prof_full = C->prof_full; to_mask = C->to_mask;
        ring0 = C->ring0; n_pushed = C->n_pushed; status = C->status;
        // every event of the previous chunk has been popped by the time the next chunk is added (a chunk is only
        // added once Mapper::chunk_mapped(), realtime_pool.cpp:133-138): the read pointer has caught up
        n_rd = n_wr;
        n_full = 0;
    }

    // samples: stored int16 + the channel's calibration, or floats taken as they are (Chunk keeps floats, chunk.cpp:27-66)
    const int16_t *rp = raw_pa ? nullptr : raw + cd.offset;
    const float *fp = raw_pa ? raw_pa + cd.offset : nullptr;
    for (uint32_t k = 0; k < cd.n_samples; ++k) {
        float s;
        if (fp) s = fp[k];
        else {
            const uint16_t ru = (uint16_t)rp[k];
            s = __fdiv_rn(__fmul_rn(cd.cal_range, __fadd_rn((float)(int)ru, cd.cal_offset)), cd.cal_digit);
        }
        const uint32_t cur = (t & (RING - 1)) * WAVE + lane, prv = ((t - 1) & (RING - 1)) * WAVE + lane;
        const float ss = __fmul_rn(s, s);
        s_sum[cur] = s_sum[prv] + (double)s;
        s_sumsq[cur] = s_sumsq[prv] + (double)ss;
        t++;
        const uint32_t buf_mid = t - 7;
        const float t1 = tstat<UNC_WINDOW1>(s_sum, s_sumsq, lane, t, buf_mid);
        const float t2 = tstat<UNC_WINDOW2>(s_sum, s_sumsq, lane, t, buf_mid);
        const bool p1 = peak_detect(sd, &ld, t1, buf_mid, P.peak_height);
        const bool p2 = peak_detect(ld, nullptr, t2, buf_mid, P.peak_height);
        if (!(p1 || p2)) continue;
        const uint32_t evt_en = buf_mid - UNC_WINDOW1 + 1;
        const uint32_t eb = (evt_en & (RING - 1)) * WAVE + lane;
        const uint32_t length = (uint32_t)(float)(evt_en - evt_st);
        const double csum = s_sum[eb], csq = s_sumsq[eb];
        float mean = (float)((csum - evt_st_sum) / (double)length);
        evt_st = evt_en; evt_st_sum = csum; evt_st_sumsq = csq;
        len_sum = __fadd_rn(len_sum, (float)length);
        total_events++;
        mean = __fmul_rn(__fadd_rn(mean, 0.0f), 1.0f);
        if (!(mean >= P.min_mean && mean <= P.max_mean)) continue;
        // EventProfiler::add_event
        roll_push(C->pw_signal, PROF_WIN, pw_mean, pw_varsum, pw_n, pw_rd, pw_wr, pw_full, mean);
        {
            uint32_t qi = q_head + q_len; if (qi >= PROF_WIN + 1) qi -= PROF_WIN + 1;
            C->evq[qi] = mean;
            q_len++;
        }
        if (roll_unread(pw_n, pw_rd, pw_wr) <= PROF_WIN / 2) continue;
        const float win_stdv = (float)sqrt(pw_varsum / (double)pw_n);
        if (win_stdv < 5.0f) to_mask = PROF_WIN - 1;      // win_stdv_min, event_profiler.cpp:7
        else if (to_mask > 0) to_mask--;
        float next_mean = 0.0f;
        if (pw_full) {
            next_mean = C->evq[q_head];
            q_head = q_head + 1 == PROF_WIN + 1 ? 0 : q_head + 1;
            q_len--;
            pw_rd = pw_rd + 1 == PROF_WIN ? 0 : pw_rd + 1;   // window_.pop()
            pw_full = 0;
            prof_full = 1;
        }
        if (!(prof_full && to_mask == 0)) continue;
        // norm_.push(evt_mean), mapper.cpp:336-351.  A full ring (>= 6000 unread events) cannot happen when every
        // chunk is mapped before the next one is added; it is reported instead of reproducing the #SKIP path.
        if (!roll_push(ring, NORM_LEN, n_mean, n_varsum, n_n, n_rd, n_wr, n_full, next_mean)) { status |= UNC_READ_NORM_FULL; break; }
        n_pushed++;
    }

    // ---- save state
#pragma unroll
    for (int i = 0; i < RING; ++i) { C->sum[i] = s_sum[i * WAVE + lane]; C->sumsq[i] = s_sumsq[i * WAVE + lane]; }
    C->t = t; C->evt_st = evt_st; C->total_events = total_events; C->evt_st_sum = evt_st_sum; C->evt_st_sumsq = evt_st_sumsq;
    C->len_sum = len_sum;
    C->sd = RtDetector{sd.threshold, sd.window_length, sd.masked_to, sd.peak_pos, sd.peak_value, sd.valid_peak ? 1u : 0u};
    C->ld = RtDetector{ld.threshold, ld.window_length, ld.masked_to, ld.peak_pos, ld.peak_value, ld.valid_peak ? 1u : 0u};
    C->pw_mean = pw_mean; C->pw_varsum = pw_varsum; C->pw_n = pw_n; C->pw_rd = pw_rd; C->pw_wr = pw_wr; C->pw_full = pw_full;
    C->q_head = q_head; C->q_len = q_len; C->prof_full = prof_full; C->to_mask = to_mask;
    C->n_mean = n_mean; C->n_varsum = n_varsum; C->n_n = n_n; C->n_rd = n_rd; C->n_wr = n_wr; C->n_full = n_full;
    C->ring0 = ring0; C->n_pushed = n_pushed; C->status = status;

    // Normalizer::at: scale / shift from the rolling statistics after the whole chunk has been pushed
    float scale = 0.0f, shift = 0.0f;
    if (n_n > 0) {
        scale = (float)((double)tgt_stdv / sqrt(n_varsum / (double)n_n));
        shift = (float)((double)tgt_mean - (double)scale * n_mean);
    }
    unc_evt_info_t inf;
    inf.n_events = n_pushed; inf.total_events = total_events; inf.len_sum = len_sum; inf.scale = scale; inf.shift = shift;
    inf.pad = status;
    info[ci] = inf;
    ring0_out[ci] = ring0;
}
}  // namespace unc

#include "unc_kernels.h"
namespace unc {
void launch_rt_events(const int16_t *raw, const float *raw_pa, const RtChunkDesc *chunks, uint32_t n_chunks, RtChan *chans, float *norm_ring,
                      const unc_params_t &P, float tgt_mean, float tgt_stdv, unc_evt_info_t *info, uint32_t *ring0_out, hipStream_t st) {
    // as few channels per wavefront as a grid of 2048 wavefronts allows (512 channels: one each)
    uint32_t lpw = (n_chunks + 2047u) / 2048u;
    if (lpw < 1u) lpw = 1u;
    if (lpw > (uint32_t)WAVE) lpw = WAVE;
    hipLaunchKernelGGL(k_rt_events, dim3((n_chunks + lpw - 1) / lpw), dim3(WAVE), 0, st, raw, raw_pa, chunks, n_chunks, chans, norm_ring, P,
                       tgt_mean, tgt_stdv, info, ring0_out, lpw);
}
void launch_events(const DevReads &rd, const unc_params_t &P, hipStream_t st, uint32_t reads_per_wave) {
    if (reads_per_wave == 0 || reads_per_wave > (uint32_t)WAVE) reads_per_wave = WAVE;
    hipLaunchKernelGGL(k_events, dim3((rd.n_reads + reads_per_wave - 1) / reads_per_wave), dim3(WAVE), 0, st, rd, P, reads_per_wave);
}
}  // namespace unc
