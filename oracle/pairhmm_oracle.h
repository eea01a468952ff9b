/*
 * pairhmm_oracle.h -- CPU restatement of mgl's PairHMM forward algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and the CPU-baseline legs of the bench scripts may load this library, and only as
 * the checker / the timed CPU baseline.
 *
 * Parity status: PINNED by the reference's own fixture.  The reference's PairHMM sources cannot be compiled
 * here (pairhmm_common.h:16-17 includes <tbb/tbb.h>, absent from this image; no stand-ins are written), so
 * this restatement of the scalar path is checked (tests/test_pairhmm_oracle.py) against the known answers
 * the reference's own tests hold: src/test/resources/pairhmm-testdata.txt (104 cases, kept as the data
 * fixture tests/golden/pairhmm-testdata.txt) with the reference's tolerance 1e-5 on log10 likelihood, in
 * float and double mode (MicrosoftPairHmmUnitTest.java:58-117), and simpleTest's -6.022797e-01 (:22-56).
 *
 * All citations are relative to /root/reference/src/main/native/mgl_pairhmm/.
 */
#ifndef PAIRHMM_ORACLE_H
#define PAIRHMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHO_MIN_ACCEPTED 1e-28f /* pairhmm_common.h:32 */

/* One read: bases and the four quality tracks, each `len` bytes, ALREADY normalised by the caller the way the
 * Java side hands them over (phred values, not +33; MicrosoftPairHmmUnitTest.java:119-129). */
typedef struct {
    int32_t len;
    const uint8_t *bases, *qual, *ins, *del, *gcp;
} pho_read;

/* Raw forward sums (before the log10), compute_prob_scalar.cc:47-342:
 * float: scaled by 2^120, double: scaled by 2^1020 (Context.h:108,147). */
float pho_forward_float(const pho_read *read, const uint8_t *hap, int32_t hap_len);
double pho_forward_double(const pho_read *read, const uint8_t *hap, int32_t hap_len);

/* log10 likelihood of one (read, haplotype) pair with the float -> double rescue of
 * com_microsoft_mgl_pairhmm_MicrosoftPairHmm.cc:131-216; use_double = initNative's flag (:50-53).
 * *used_double (optional) reports which precision produced the answer. */
double pho_log10_likelihood(const pho_read *read, const uint8_t *hap, int32_t hap_len, int use_double, int *used_double);

/* computeLikelihoodsNative with the JNI's buffer layout (…PairHmm.cc:77-128): lengths = {n_reads, len_r...,
 * n_haps, len_h...}; reads = per read bases|qual|ins|del|gcp; haps = concatenated bases;
 * out[r * n_haps + h].  nthreads > 1 splits the reads over pthreads (the reference uses tbb::parallel_for). */
int pho_compute_likelihoods(const int32_t *lengths, const uint8_t *reads, const uint8_t *haps, double *out, int use_double,
                            int nthreads);

/* flat pair list: read r = reads_data + 5 * read_off[r] (five tracks of read_off[r+1]-read_off[r] bytes), hap h =
 * haps_data[hap_off[h] .. hap_off[h+1]); out[k] for pair (pair_read[k], pair_hap[k]) */
int pho_compute_pairs(int64_t n_pairs, const uint8_t *reads_data, const int64_t *read_off, const uint8_t *haps_data,
                      const int64_t *hap_off, const int32_t *pair_read, const int32_t *pair_hap, double *out,
                      int use_double, int nthreads, int32_t *used_double);

#ifdef __cplusplus
}
#endif
#endif
