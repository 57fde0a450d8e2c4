// host_calib.cpp -- the multiplier search of the reference's INT8 calibration tool, host side.
//
// Behavioural mirror of entropy_calibration (src/yolov2_forward_network_quantized.c:1292-1400),
// split at its only data-parallel part: the histogram H[b] = #{ x : lround(|x| / bin_width) == b }
// (saturated at max_bin - 1) is counted on the GPU (hist_abs_kernel, layers.hip, exact integers);
// everything after it -- the KL(P || Q) scan over the clip point i = 128 .. max_bin-1 -- runs here
// with the reference's types and evaluation order (float accumulators, the uint64 outlier counter
// that passes through float on every add, log in double), so that the same histogram gives the
// same multiplier bit for bit (tests/test_calibration.py against the reference-built library).
//
// One exact shortcut: a bin with P[j] == 0 also has Q[j] == 0 ("preserve empty bins"), its KL term
// is 0 * log(FLT_MIN / FLT_MIN) = +0 and adding +0 changes nothing, so empty bins are skipped in the
// innermost sum (activations fill a few hundred of the 4096 bins).
#include "yl_internal.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace yl {

float entropy_from_counts(const uint32_t *counts, int max_bin, float bin_width)
{
    std::vector<float> m_array(max_bin, 0.f), H(max_bin), P(max_bin, 0.f), Q(max_bin, 0.f);
    // the reference increments a float per element: exact up to 2^24, then it sticks
    for (int b = 0; b < max_bin; ++b) H[b] = counts[b] >= 16777216u ? 16777216.f : (float)counts[b];
    float qQ[128];
    uint64_t qcount[128];
    for (int i = 128; i < max_bin; ++i) {
        uint64_t outliers = 0;
        const int last_bin = i - 1;
        for (int j = 0; j < max_bin; ++j) {
            if (j <= last_bin) P[j] = H[j];
            else outliers = (uint64_t)((float)outliers + H[j]);
        }
        const float expand = i / 128.0F;
        for (int j = 0; j < 128; ++j) { qQ[j] = 0; qcount[j] = 0; }
        for (int j = 0; j < i; ++j) {
            int qb = (int)lround((double)(j / expand));
            if (qb > 127) qb = 127;
            qQ[qb] += P[j];
            if (P[j] != 0) qcount[qb]++;
        }
        for (int j = 0; j < i; ++j) Q[j] = 0;
        for (int j = 0; j < i; ++j) {
            int qb = (int)lround((double)(j / expand));
            if (qb > 127) qb = 127;
            if (P[j] != 0) Q[j] = qQ[qb] / (float)qcount[qb];
        }
        P[last_bin] = P[last_bin] + (float)outliers;
        float sum_P = 0, sum_Q = 0;
        for (int j = 0; j < i; ++j) { sum_P += P[j]; sum_Q += Q[j]; }
        for (int j = 0; j < i; ++j) { P[j] /= sum_P; Q[j] /= sum_Q; }
        float m = m_array[i];
        for (int j = 0; j < i; ++j) {
            if (P[j] == 0) continue;                     // exact: the term is +0 (see header)
            m = (float)((double)m + (double)P[j] * log((double)((P[j] + FLT_MIN) / (Q[j] + FLT_MIN))));
        }
        m_array[i] = m;
    }
    float m_index = 128, min_m = FLT_MAX;
    for (int i = 128; i < max_bin; ++i)
        if (m_array[i] < min_m) { min_m = m_array[i]; m_index = (float)i; }
    const float threshold = (float)(((double)m_index + 0.5) * (double)bin_width);
    return 127 / threshold;
}

}  // namespace yl
