// pairhmm_device.h -- types shared by the PairHMM kernels (pairhmm_kernels.hip) and the C-ABI host layer
// (pairhmm_capi.cpp).
#ifndef MGL_PAIRHMM_DEVICE_H
#define MGL_PAIRHMM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgl_ph_dev {

constexpr int MAX_QUAL = 254;                                       // Context.h:7
constexpr int M2M_SIZE = ((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1;    // Context.h:24

struct PhArgs {
    int64_t n_pairs;
    const uint8_t *reads;     // per read: bases | quals | ins | del | gcp, len bytes each, at 5 * read_off[r]
    const int64_t *read_off;
    const uint8_t *haps;
    const int64_t *hap_off;
    const int32_t *pair_read, *pair_hap;
    const float *ph2pr_f, *m2m_f;   // Context<float> tables (Context.h:141-146, 49-59), built on the host
    const double *ph2pr_d, *m2m_d;  // Context<double> tables
    double log10_initial_f, log10_initial_d; // (double)log10f(2^120), log10(2^1020): Context.h:148,109, from the host's libm
    int hap_cap;                    // upper bound of the haplotype lengths (sizes the LDS carve)
    double *out;                    // log10 likelihood per pair
    int32_t *need_double;           // per pair: written by the float pass, read by the rescue pass
    int rescue_only;                // double pass: 1 = only the pairs flagged by the float pass, 0 = every pair
};

// rows = read rows per stripe = lanes per pair: 16 (four pairs per wave) or 64 (one pair per wave)
int ph_lds_bytes(int hap_cap, int rows, int elem_bytes);
hipError_t launch_pairhmm_float(const PhArgs &a, int rows, int rows_per_lane, hipStream_t stream);
hipError_t launch_pairhmm_double(const PhArgs &a, int rows, int rows_per_lane, hipStream_t stream);

} // namespace mgl_ph_dev
#endif
