// Event detection + whole-read normalisation, one LANE per read (each lane runs the reference's
// strictly sequential two-window t-test detector over its own read, so the double cumulative sums
// are added in exactly the reference's order).  Replaces, for a batch of reads:
//   ReadBuffer calibration            read_buffer.cpp:239-241   (u16 reinterpretation included)
//   EventDetector::get_means          event_detector.cpp:133-145 (add_sample 83-112, compute_tstat
//                                     174-219, peak_detect 221-279, create_event 296-319)
//   Normalizer::set_signal / at       normalizer.cpp:31-44,114-118 (scale/shift only; the affine map
//                                     itself is applied per event in k_map)
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

// event_detector.cpp:221-279.  `other` is the long detector when `det` is the short one.
__device__ __forceinline__ bool peak_detect(Detector &det, Detector *long_det, float current_value, uint32_t buf_mid,
                                            float peak_height) {
    if (det.masked_to >= buf_mid) return false;
    if (det.peak_pos == -1) {
        if (current_value < det.peak_value) {
            det.peak_value = current_value;
        } else if (__fsub_rn(current_value, det.peak_value) > peak_height) {
            det.peak_value = current_value;
            det.peak_pos = (int32_t)buf_mid;
        }
    } else {
        if (current_value > det.peak_value) {
            det.peak_value = current_value;
            det.peak_pos = (int32_t)buf_mid;
        }
        if (long_det != nullptr) {
            if (det.peak_value > det.threshold) {
                long_det->masked_to = (uint32_t)det.peak_pos + det.window_length;
                long_det->peak_pos = -1;
                long_det->peak_value = FLT_MAX;
                long_det->valid_peak = false;
            }
        }
        if (__fsub_rn(det.peak_value, current_value) > peak_height && det.peak_value > det.threshold) {
            det.valid_peak = true;
        }
        if (det.valid_peak && (buf_mid - (uint32_t)det.peak_pos) > det.window_length / 2) {
            det.peak_pos = -1;
            det.peak_value = current_value;
            det.valid_peak = false;
            return true;
        }
    }
    return false;
}

// event_detector.cpp:174-219 with the ring addressed by ABSOLUTE position (slot = pos & 15).
// `st_pos` is the position the reference's `(buf_mid - w) % 13` lands on (it wraps for
// buf_mid < w, where slot (2^32 + buf_mid - w) % 13 == buf_mid + 6 still holds C[buf_mid + 6]).
__device__ __forceinline__ float tstat(const double *sum, const double *sumsq, int lane, uint32_t t, uint32_t buf_mid,
                                       uint32_t w) {
    if (t <= 2 * w) return 0.0f;
    const float wf = (float)w;
    uint32_t st_pos = buf_mid >= w ? buf_mid - w : buf_mid + 6;
    uint32_t i = (buf_mid & (RING - 1)) * WAVE + lane, st = (st_pos & (RING - 1)) * WAVE + lane,
             en = ((buf_mid + w) & (RING - 1)) * WAVE + lane;
    double sum1 = sum[i] - sum[st];
    double sumsq1 = sumsq[i] - sumsq[st];
    float sum2 = (float)(sum[en] - sum[i]);
    float sumsq2 = (float)(sumsq[en] - sumsq[i]);
    float mean1 = (float)(sum1 / (double)wf);
    float mean2 = __fdiv_rn(sum2, wf);
    float m1sq = __fmul_rn(mean1, mean1), q2 = __fdiv_rn(sumsq2, wf), m2sq = __fmul_rn(mean2, mean2);
    float combined_var = (float)(((sumsq1 / (double)wf - (double)m1sq) + (double)q2) - (double)m2sq);
    combined_var = fmaxf(combined_var, FLT_MIN);
    float delta_mean = __fsub_rn(mean2, mean1);
    return __fdiv_rn(fabsf(delta_mean), __fsqrt_rn(__fdiv_rn(combined_var, wf)));
}

__global__ __launch_bounds__(64) void k_events(DevReads R, unc_params_t P) {
    __shared__ double s_sum[RING * WAVE];
    __shared__ double s_sumsq[RING * WAVE];
    const int lane = lane_id();
    const uint32_t r = blockIdx.x * WAVE + lane;
    const bool active = r < R.n_reads;

    uint64_t off = 0, n = 0, moff = 0;
    uint32_t mcap = 0;
    float cal_range = 1.f, cal_offset = 0.f, cal_digit = 1.f;
    if (active) {
        off = R.offsets[r];
        n = R.offsets[r + 1] - off;
        moff = R.moff[r];
        mcap = (uint32_t)(R.moff[r + 1] - moff);
        cal_range = R.calib[r].range;
        cal_offset = R.calib[r].offset;
        cal_digit = R.calib[r].digitisation;
    }
    // EventDetector::reset, event_detector.cpp:47-77
    s_sum[lane] = 0.0;
    s_sumsq[lane] = 0.0;
    uint32_t t = 1, evt_st = 0, total_events = 0, n_kept = 0;
    double evt_st_sum = 0.0, evt_st_sumsq = 0.0;
    float len_sum = 0.0f;
    Detector sd{P.threshold1, P.window_length1, 0u, -1, FLT_MAX, false};
    Detector ld{P.threshold2, P.window_length2, 0u, -1, FLT_MAX, false};
    const int16_t *raw = R.raw + off;
    float *means = R.means + moff;

    for (uint64_t k = 0; k < n; ++k) {
        // calibration: u16 reinterpretation of the stored i16, three float roundings
        uint16_t ru = (uint16_t)raw[k];
        float s = __fdiv_rn(__fmul_rn(cal_range, __fadd_rn((float)(int)ru, cal_offset)), cal_digit);
        // add_sample: position t gets C[t] = C[t-1] + s, Q[t] = Q[t-1] + (float)(s*s)
        uint32_t cur = (t & (RING - 1)) * WAVE + lane, prv = ((t - 1) & (RING - 1)) * WAVE + lane;
        float ss = __fmul_rn(s, s);
        s_sum[cur] = s_sum[prv] + (double)s;
        s_sumsq[cur] = s_sumsq[prv] + (double)ss;
        t++;
        uint32_t buf_mid = t - 7;   // t - BUF_LEN/2 - 1, wraps for the first samples exactly as the u32 does
        float t1 = tstat(s_sum, s_sumsq, lane, t, buf_mid, UNC_WINDOW1);
        float t2 = tstat(s_sum, s_sumsq, lane, t, buf_mid, UNC_WINDOW2);
        bool p1 = peak_detect(sd, &ld, t1, buf_mid, P.peak_height);
        bool p2 = peak_detect(ld, nullptr, t2, buf_mid, P.peak_height);
        if (p1 || p2) {
            // create_event(buf_mid - window_length1 + 1)
            uint32_t evt_en = buf_mid - UNC_WINDOW1 + 1;
            uint32_t eb = (evt_en & (RING - 1)) * WAVE + lane;
            uint32_t length = (uint32_t)(float)(evt_en - evt_st);
            double csum = s_sum[eb], csq = s_sumsq[eb];
            float mean = (float)((csum - evt_st_sum) / (double)length);
            evt_st = evt_en;
            evt_st_sum = csum;
            evt_st_sumsq = csq;
            len_sum = __fadd_rn(len_sum, (float)length);
            total_events++;
            mean = __fmul_rn(__fadd_rn(mean, 0.0f), 1.0f);   // calibrate(): cal_offset_=0, cal_coef_=1
            if (mean >= P.min_mean && mean <= P.max_mean && n_kept < mcap) means[n_kept++] = mean;
        }
    }

    // Normalizer::set_signal: sequential double sums in index order, then scale/shift (::at)
    float scale = 0.0f, shift = 0.0f;
    if (active && n_kept > 0) {
        double mean = 0.0;
        for (uint32_t i = 0; i < n_kept; ++i) mean += (double)means[i];
        mean /= (double)n_kept;
        double varsum = 0.0;
        for (uint32_t i = 0; i < n_kept; ++i) {
            double e = (double)means[i] - mean;
            varsum += e * e;
        }
        const float tgt_mean = R.tgt_mean, tgt_stdv = R.tgt_stdv;
        scale = (float)((double)tgt_stdv / sqrt(varsum / (double)n_kept));
        shift = (float)((double)tgt_mean - (double)scale * mean);
    }
    if (active) {
        unc_evt_info_t inf;
        inf.n_events = n_kept;
        inf.total_events = total_events;
        inf.len_sum = len_sum;
        inf.scale = scale;
        inf.shift = shift;
        inf.pad = 0;
        R.info[r] = inf;
    }
}

}  // namespace unc

#include "unc_kernels.h"
namespace unc {
void launch_events(const DevReads &rd, const unc_params_t &P, hipStream_t st) {
    hipLaunchKernelGGL(k_events, dim3((rd.n_reads + WAVE - 1) / WAVE), dim3(WAVE), 0, st, rd, P);
}
}  // namespace unc
