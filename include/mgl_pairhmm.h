/*
 * mgl_pairhmm.h -- C ABI of the MI355X PairHMM forward kernel (mgl's other
 * native kernel, src/main/native/mgl_pairhmm/).  Drop-in boundary of libmgl_pairhmm_hip.so: plain pointers
 * and sizes, no C++ or torch types.  File:line citations are relative to
 * /root/reference/src/main/native/mgl_pairhmm/ unless they name a Java file.
 *
 * What is computed: for a read (bases + base / insertion / deletion / continuation qualities) and a
 * haplotype, log10 of the forward-algorithm likelihood exactly as compute_prob_scalar.cc:5-45,47-342 defines
 * it, in float with the reference's rescue in double when the float sum falls below 1e-28
 * (com_microsoft_mgl_pairhmm_MicrosoftPairHmm.cc:131-216), or in double throughout (initNative's
 * use_double, :50-53).  Results agree with the reference's known answers within its own test tolerance,
 * 1e-5 on the log10 likelihood (MicrosoftPairHmmUnitTest.java:55,103).
 *
 * There is no CPU fallback: without a HIP device every compute entry returns MGL_PAIRHMM_ERR_DEVICE.
 */
#ifndef MGL_PAIRHMM_H
#define MGL_PAIRHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGL_PAIRHMM_VERSION 1

enum {
    MGL_PAIRHMM_OK = 0,
    MGL_PAIRHMM_ERR_BAD_ARG = 1,    /* null pointer, negative count, empty read or haplotype */
    MGL_PAIRHMM_ERR_NOMEM = 3,      /* host or device allocation failed */
    MGL_PAIRHMM_ERR_DEVICE = 4,     /* no HIP device / HIP runtime error (see mgl_pairhmm_last_error) */
    MGL_PAIRHMM_ERR_UNSUPPORTED = 5 /* haplotype longer than mgl_pairhmm_max_haplotype_len() */
};

typedef struct mgl_pairhmm_ctx mgl_pairhmm_ctx;

int mgl_pairhmm_version(void);
const char *mgl_pairhmm_strerror(int status);
const char *mgl_pairhmm_last_error(const mgl_pairhmm_ctx *ctx);
int mgl_pairhmm_device_count(void);
int mgl_pairhmm_max_haplotype_len(int use_double);

/* one context per device and caller thread (calls on one context are serialised) */
int mgl_pairhmm_ctx_create(int device, mgl_pairhmm_ctx **out);
void mgl_pairhmm_ctx_destroy(mgl_pairhmm_ctx *ctx);

/* replaces initNative(use_double, max_threads), …PairHmm.cc:47-70 (max_threads is ignored there too) */
int mgl_pairhmm_initialize(mgl_pairhmm_ctx *ctx, int use_double, int max_threads);

/* lanes per pair of the kernels: 0 (default) = per batch (16 lanes x four pairs per wave for very short reads; 21 lanes x
 * three pairs or 32 lanes x two pairs per wave, up to five read rows per lane, for reads of 24 .. 160 bases and more
 * than 1 024 pairs; else 64 lanes x one pair per wave, up to four rows per lane and as many stripes as the read needs);
 * 16 / 21 / 32 / 64 force one (tests; 21 and 32 fall back to 64 for reads beyond 105 / 160 bases; every cell and the
 * column-order final sum are computed the same way, so the results agree to float rounding: only the contraction of
 * multiply-adds may differ between the variants) */
int mgl_pairhmm_set_stripe_rows(mgl_pairhmm_ctx *ctx, int rows);

/* replaces computeLikelihoodsNative(lengthBuffer, readsBuffer, haplotypesBuffer, likelihoodBuffer),
 * …PairHmm.cc:77-222, with the very same buffer contents (MicrosoftPairHmm.java:62-112):
 *   lengths   = { n_reads, len(read 0), ..., n_haps, len(hap 0), ... }              int32
 *   reads     = per read: bases | quals | insertion GOP | deletion GOP | overall GCP, len(read) bytes each
 *   haps      = haplotype bases, concatenated
 *   likelihoods[r * n_haps + h] = log10 likelihood of read r given haplotype h          double, host memory
 * All pointers are host memory. */
int mgl_pairhmm_compute_likelihoods(mgl_pairhmm_ctx *ctx, const int32_t *lengths, const uint8_t *reads, const uint8_t *haps,
                                    double *likelihoods);

/* Flat pair list (many regions in one call): read r = reads_data[5 * read_off[r] ...], five tracks of
 * read_off[r+1] - read_off[r] bytes each in the order above; hap h = haps_data[hap_off[h] .. hap_off[h+1]);
 * out[k] for pair (pair_read[k], pair_hap[k]).  Host memory. */
int mgl_pairhmm_compute_pairs(mgl_pairhmm_ctx *ctx, int64_t n_pairs, int64_t n_reads, const uint8_t *reads_data,
                              const int64_t *read_off, int64_t n_haps, const uint8_t *haps_data, const int64_t *hap_off,
                              const int32_t *pair_read, const int32_t *pair_hap, double *out);

/* The same with every pointer in device memory, enqueued on `stream` (a hipStream_t; NULL = the context's
 * own stream); nothing is copied or synchronised.  max_read_len / max_hap_len bound the lengths in the batch.
 * d_used_double (optional, int32 per pair) reports which pairs were rescued in double. */
int mgl_pairhmm_compute_pairs_device(mgl_pairhmm_ctx *ctx, void *stream, int64_t n_pairs, const uint8_t *d_reads_data,
                                     const int64_t *d_read_off, const uint8_t *d_haps_data, const int64_t *d_hap_off,
                                     const int32_t *d_pair_read, const int32_t *d_pair_hap, int max_read_len, int max_hap_len,
                                     double *d_out, int32_t *d_used_double);

/* kernel time of the last call (HIP events), for the benches: float pass and double pass in ms, pairs that
 * needed the double pass (device variant: -1, not read back) */
typedef struct {
    float float_ms, double_ms;
    int64_t cells;        /* sum of read_len * hap_len over the pairs (host entries only, else 0) */
    int64_t rescued;
} mgl_pairhmm_timing;
int mgl_pairhmm_set_profiling(mgl_pairhmm_ctx *ctx, int enable);
int mgl_pairhmm_get_timing(mgl_pairhmm_ctx *ctx, mgl_pairhmm_timing *out);

#ifdef __cplusplus
}
#endif
#endif
