/*
 * sw_oracle.h -- CPU restatement of mgl's Smith-Waterman affine-gap core.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py)
 * against golden vectors in tests/golden/ that were generated in the authoring
 * container from the reference's own sources compiled in place
 * (oracle/Makefile -> oracle/_ref/libmgl_ref.so, generator
 * tests/golden/make_golden.py).  The reference ships no Smith-Waterman tests
 * or vectors of its own.
 *
 * All citations are relative to /root/reference/src/main/native/mgl_sw/.
 */
#ifndef SW_ORACLE_H
#define SW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* overhang strategies, sw_common.h:22-25 */
#define SWO_SOFTCLIP 1
#define SWO_INDEL 2
#define SWO_LEAD_INDEL 4
#define SWO_IGNORE 8

#define SWO_NEG_INF (-0x40000000) /* sw_common.h:33 */

/* ScoreMax, sw_common.h:36-40 (same field meaning, plain C) */
typedef struct {
    int32_t mqe, mqe_t;
    int32_t max, max_t, max_q;
    int32_t seg_length;
} swo_score;

enum { SWO_OK = 0, SWO_BAD_ARG = 1, SWO_CIGAR_OVERFLOW = 2, SWO_NOMEM = 3 };

/* sign normalisation of the JNI boundary,
 * com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman.cpp:51-55 */
void swo_normalize_params(int *match, int *mismatch, int *gopen, int *gext);

/* calculateMatrix (sw.cpp:5-146): fills the logical backtrack matrix
 * btr[(tl+1)*(ql+1)] (row-major, row 0 / column 0 left untouched) and *ez.
 * h_end (optional) receives H[tl][ql]. */
int swo_fill(const uint8_t *t, int tl, const uint8_t *q, int ql, int match,
             int mismatch, int gopen, int gext, int strategy, int32_t *btr,
             swo_score *ez, int32_t *h_end);

/* calculateCigar (sw.cpp:149-255): traceback + CIGAR text.  Writes at most
 * cap bytes (no terminator), *len = bytes produced, *offset = alignment
 * offset. */
int swo_cigar(const int32_t *btr, int tl, int ql, int strategy,
              const swo_score *ez, char *cigar, int cap, int *len,
              int *offset);

/* align_scalar (sw.cpp:258-272) == align_avx (sw_avx.cpp:6-108) as a function */
int swo_align(const uint8_t *t, int tl, const uint8_t *q, int ql, int match,
              int mismatch, int gopen, int gext, int strategy, char *cigar,
              int cap, int *len, int *offset, swo_score *ez, int32_t *h_end);

/* CRC-32 (zlib polynomial) of the logical backtrack matrix restricted to
 * i=1..tl, j=1..ql, row-major, little-endian int32 -- the fixture checksum. */
uint32_t swo_btr_crc32(const int32_t *btr, int tl, int ql);

/* ---- substitution-matrix scoring ("protein"): NOT in the reference, which scores by
 * byte equality only (sw.cpp:55) -- PARITY UNPINNED for this mode, never claimed.  The same recurrence with
 * diag = H[i-1][j-1] + mat[code[t] * 32 + code[q]] (code: byte -> 0..31, mat: 32 x 32 int8); gap open / extend
 * are taken as given (positive). */
#define SWO_MATRIX_DIM 32
int swo_fill_matrix(const uint8_t *t, int tl, const uint8_t *q, int ql, const uint8_t *code, const int8_t *mat,
                    int gopen, int gext, int strategy, int32_t *btr, swo_score *ez, int32_t *h_end);
int swo_align_matrix(const uint8_t *t, int tl, const uint8_t *q, int ql, const uint8_t *code, const int8_t *mat,
                     int gopen, int gext, int strategy, char *cigar, int cap, int *len, int *offset, swo_score *ez);
int swo_align_batch_matrix(int n, const uint8_t *targets, const int64_t *t_off, const uint8_t *queries,
                           const int64_t *q_off, const uint8_t *code, const int8_t *mat, int gopen, int gext,
                           int strategy, int nthreads, int32_t *offset_out, swo_score *score_out, char *cigar_out,
                           int cigar_stride, int32_t *cigar_len);

/* Batch driver over nthreads POSIX threads (one pair per task) -- used for
 * bench.py's cpu_baseline "port" leg.  Sequences are concatenated; pair k is
 * targets[t_off[k]..t_off[k+1]) vs queries[q_off[k]..q_off[k+1]).  cigar_out
 * is n*cigar_stride bytes, zero padded. */
int swo_align_batch(int n, const uint8_t *targets, const int64_t *t_off,
                    const uint8_t *queries, const int64_t *q_off, int match,
                    int mismatch, int gopen, int gext, int strategy,
                    int nthreads, int32_t *offset_out, swo_score *score_out,
                    char *cigar_out, int cigar_stride, int32_t *cigar_len);

#ifdef __cplusplus
}
#endif
#endif
