/*
 * sw_oracle.c -- CPU restatement of mgl's Smith-Waterman affine-gap core.
 *
 * TEST INFRASTRUCTURE ONLY (see sw_oracle.h).  Parity status: PINNED against
 * tests/golden/ (vectors produced by the compiled reference).
 *
 * The algorithm restated here is the one defined by the reference's scalar
 * path sw.cpp (which its AVX2 path sw_avx.cpp reproduces cell for cell):
 *   - no zero floor; best score searched only on the last column / last row
 *   - priority diag >= right(F, insertion) >= down(E, deletion)
 *   - gap extension wins ties against gap open
 *   - the backtrack matrix holds signed gap run lengths (+k = k rows up,
 *     -k = k columns left, 0 = diagonal)
 * Citations are relative to /root/reference/src/main/native/mgl_sw/.
 */
#include "sw_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman.cpp:51-55 */
void swo_normalize_params(int *match, int *mismatch, int *gopen, int *gext)
{
    if (*match < 0) *match = -*match;
    if (*mismatch > 0) *mismatch = -*mismatch;
    if (*gopen < 0) *gopen = -*gopen;
    if (*gext < 0) *gext = -*gext;
}

/* H on the matrix border: 0 for SOFTCLIP/IGNORE, -(o + (k-1)e) for the two
 * INDEL strategies; H[0][0] is always 0.  sw.cpp:29-40,47-49 */
static inline int border(int k, int gopen, int gext, int indel)
{
    return (indel && k > 0) ? -gopen - (k - 1) * gext : 0;
}

/* code / mat: substitution-matrix scoring (swo_fill_matrix below), NULL for the reference's match / mismatch */
static int fill_core(const uint8_t *t, int tl, const uint8_t *q, int ql, int match,
                     int mismatch, const uint8_t *code, const int8_t *mat, int gopen, int gext, int strategy,
                     int32_t *btr, swo_score *ez, int32_t *h_end)
{
    if (!t || !q || !btr || !ez || tl < 1 || ql < 1) return SWO_BAD_ARG;
    const int m = ql + 1;
    const int indel = (strategy & (SWO_INDEL | SWO_LEAD_INDEL)) != 0;

    /* per-column state: H of the previous row, E entering the current row
     * and its vertical run length (sw.cpp:10-18) */
    int32_t *hrow = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)m);
    if (!hrow) return SWO_NOMEM;
    int32_t *ecol = hrow + m, *vrun = ecol + m;
    for (int j = 0; j <= ql; j++) {
        hrow[j] = border(j, gopen, gext, indel);
        ecol[j] = hrow[j] - gopen; /* E[1][j] = H[0][j] - o, sw.cpp:16,34 */
        vrun[j] = 1;
    }

    ez->mqe = SWO_NEG_INF; /* sw_common.h:37-39 */
    ez->mqe_t = -1;
    ez->max = SWO_NEG_INF;
    ez->max_t = ez->max_q = -1;
    ez->seg_length = 0;

    for (int i = 1; i <= tl; i++) {
        const int hleft0 = border(i, gopen, gext, indel); /* H[i][0] */
        int f = hleft0 - gopen; /* F[i][1] = H[i][0] - o, sw.cpp:24,38 */
        int hrun = 1;
        int hdiag = hrow[0]; /* H[i-1][0] */
        hrow[0] = hleft0;
        const uint8_t a = t[i - 1];
        int h = 0;
        for (int j = 1; j <= ql; j++) {
            /* sw.cpp:55 raw byte comparison (or the substitution matrix of the protein extension) */
            const int diag = hdiag + (mat ? (int)mat[code[a] * SWO_MATRIX_DIM + code[q[j - 1]]]
                                          : (a == q[j - 1] ? match : mismatch));
            const int down = ecol[j], right = f;
            int32_t mark;
            /* sw.cpp:60-71 */
            if (diag >= down && diag >= right) {
                h = diag;
                mark = 0;
            } else if (right >= down) {
                h = right;
                mark = -hrun;
            } else {
                h = down;
                mark = vrun[j];
            }
            btr[(size_t)i * m + j] = mark;
            /* sw.cpp:73-82 (vertical) and :84-93 (horizontal): a new gap
             * only wins when strictly better than extending */
            if (h - gopen > ecol[j] - gext) {
                ecol[j] = h - gopen;
                vrun[j] = 1;
            } else {
                ecol[j] -= gext;
                vrun[j]++;
            }
            if (h - gopen > f - gext) {
                f = h - gopen;
                hrun = 1;
            } else {
                f -= gext;
                hrun++;
            }
            hdiag = hrow[j];
            hrow[j] = h;
        }
        /* last column, later row wins ties: sw.cpp:100-104 */
        if (h >= ez->mqe) {
            ez->mqe = h;
            ez->mqe_t = i;
        }
    }

    /* last row scan, sw.cpp:116-127 */
    ez->max = ez->mqe;
    ez->max_t = ez->mqe_t;
    ez->max_q = ql;
    for (int j = 1; j <= ql; j++) {
        const int sc = hrow[j];
        if (sc > ez->max ||
            (sc == ez->max && abs(tl - j) < abs(ez->max_t - ez->max_q))) {
            ez->max = sc;
            ez->max_t = tl;
            ez->max_q = j;
            ez->seg_length = ql - j;
        }
    }
    if (h_end) *h_end = hrow[ql];
    free(hrow);
    return SWO_OK;
}

int swo_fill(const uint8_t *t, int tl, const uint8_t *q, int ql, int match,
             int mismatch, int gopen, int gext, int strategy, int32_t *btr,
             swo_score *ez, int32_t *h_end)
{
    return fill_core(t, tl, q, ql, match, mismatch, NULL, NULL, gopen, gext, strategy, btr, ez, h_end);
}

int swo_fill_matrix(const uint8_t *t, int tl, const uint8_t *q, int ql, const uint8_t *code, const int8_t *mat,
                    int gopen, int gext, int strategy, int32_t *btr, swo_score *ez, int32_t *h_end)
{
    if (!code || !mat) return SWO_BAD_ARG;
    return fill_core(t, tl, q, ql, 0, 0, code, mat, gopen, gext, strategy, btr, ez, h_end);
}

typedef struct {
    char op;
    int len;
} cig_el;

int swo_cigar(const int32_t *btr, int tl, int ql, int strategy,
              const swo_score *ez, char *cigar, int cap, int *len,
              int *offset)
{
    if (!btr || !ez || !cigar || !len || !offset || tl < 1 || ql < 1)
        return SWO_BAD_ARG;
    const int m = ql + 1;
    int I, J, seg = 0;
    /* start cell, sw.cpp:155-170 */
    if (strategy == SWO_INDEL) {
        I = tl;
        J = ql;
    } else if (strategy != SWO_LEAD_INDEL) {
        I = ez->max_t;
        J = ez->max_q;
        seg = ez->seg_length;
    } else {
        I = ez->mqe_t;
        J = ql;
    }
    if (I < 1 || J < 1 || I > tl || J > ql) return SWO_BAD_ARG;

    /* elements are produced back to front (the reference push_front()s onto
     * a std::list, sw.cpp:172-214); el[] is filled from its end */
    const int max_el = tl + ql + 4;
    cig_el *el = (cig_el *)malloc(sizeof(cig_el) * (size_t)max_el);
    if (!el) return SWO_NOMEM;
    int head = max_el;
#define PUSH_FRONT(o, l)      \
    do {                      \
        --head;               \
        el[head].op = (o);    \
        el[head].len = (l);   \
    } while (0)

    if (seg > 0 && strategy == SWO_SOFTCLIP) { /* sw.cpp:173-176 */
        PUSH_FRONT('S', seg);
        seg = 0;
    }
    char state = 'M';
    do { /* sw.cpp:182-214 */
        const int b = btr[(size_t)I * m + J];
        char next;
        int step = 1;
        if (b > 0) {
            next = 'D';
            step = b;
            I -= step;
        } else if (b < 0) {
            next = 'I';
            step = -b;
            J -= step;
        } else {
            next = 'M';
            I--;
            J--;
        }
        if (next == state) {
            seg += step;
        } else {
            PUSH_FRONT(state, seg);
            seg = step;
            state = next;
        }
    } while (I > 0 && J > 0);

    int off;
    if (strategy == SWO_SOFTCLIP) { /* sw.cpp:225-229 */
        PUSH_FRONT(state, seg);
        if (J > 0) PUSH_FRONT('S', J);
        off = I;
    } else if (strategy == SWO_IGNORE) { /* sw.cpp:230-233 */
        PUSH_FRONT(state, seg + J);
        off = I - J;
    } else { /* sw.cpp:234-248 */
        PUSH_FRONT(state, seg);
        if (I > 0)
            PUSH_FRONT('D', I);
        else if (J > 0)
            PUSH_FRONT('I', J);
        off = 0;
    }
#undef PUSH_FRONT

    /* sw.cpp:251-252: zero-length elements are skipped, nothing is merged */
    int n = 0, rc = SWO_OK;
    for (int k = head; k < max_el; k++) {
        if (el[k].len <= 0) continue;
        char tmp[16];
        const int w = snprintf(tmp, sizeof tmp, "%d%c", el[k].len, el[k].op);
        if (n + w > cap) {
            rc = SWO_CIGAR_OVERFLOW;
            break;
        }
        memcpy(cigar + n, tmp, (size_t)w);
        n += w;
    }
    free(el);
    *len = n;
    *offset = off;
    return rc;
}

int swo_align(const uint8_t *t, int tl, const uint8_t *q, int ql, int match,
              int mismatch, int gopen, int gext, int strategy, char *cigar,
              int cap, int *len, int *offset, swo_score *ez, int32_t *h_end)
{
    if (tl < 1 || ql < 1) return SWO_BAD_ARG;
    swo_score local;
    if (!ez) ez = &local;
    /* sw.cpp:262-265 */
    int32_t *btr =
        (int32_t *)calloc((size_t)(tl + 1) * (size_t)(ql + 1), sizeof(int32_t));
    if (!btr) return SWO_NOMEM;
    int rc = swo_fill(t, tl, q, ql, match, mismatch, gopen, gext, strategy, btr,
                      ez, h_end);
    if (rc == SWO_OK)
        rc = swo_cigar(btr, tl, ql, strategy, ez, cigar, cap, len, offset);
    free(btr);
    return rc;
}

int swo_align_matrix(const uint8_t *t, int tl, const uint8_t *q, int ql, const uint8_t *code, const int8_t *mat,
                     int gopen, int gext, int strategy, char *cigar, int cap, int *len, int *offset, swo_score *ez)
{
    if (tl < 1 || ql < 1) return SWO_BAD_ARG;
    swo_score local;
    if (!ez) ez = &local;
    int32_t *btr = (int32_t *)calloc((size_t)(tl + 1) * (size_t)(ql + 1), sizeof(int32_t));
    if (!btr) return SWO_NOMEM;
    int rc = swo_fill_matrix(t, tl, q, ql, code, mat, gopen, gext, strategy, btr, ez, NULL);
    if (rc == SWO_OK) rc = swo_cigar(btr, tl, ql, strategy, ez, cigar, cap, len, offset);
    free(btr);
    return rc;
}

uint32_t swo_btr_crc32(const int32_t *btr, int tl, int ql)
{
    static uint32_t table[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
        ready = 1;
    }
    uint32_t crc = 0xFFFFFFFFu;
    const int m = ql + 1;
    for (int i = 1; i <= tl; i++)
        for (int j = 1; j <= ql; j++) {
            uint32_t v = (uint32_t)btr[(size_t)i * m + j];
            for (int b = 0; b < 4; b++) {
                crc = table[(crc ^ (v & 0xFF)) & 0xFF] ^ (crc >> 8);
                v >>= 8;
            }
        }
    return crc ^ 0xFFFFFFFFu;
}

typedef struct {
    int n, tid, nthreads;
    const uint8_t *targets, *queries;
    const int64_t *t_off, *q_off;
    int match, mismatch, gopen, gext, strategy;
    int32_t *offset_out, *cigar_len;
    swo_score *score_out;
    char *cigar_out;
    int cigar_stride;
    int rc;
    const uint8_t *code; /* matrix mode (swo_align_batch_matrix), else NULL */
    const int8_t *mat;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *jb = (batch_job *)arg;
    for (int k = jb->tid; k < jb->n; k += jb->nthreads) {
        const int tl = (int)(jb->t_off[k + 1] - jb->t_off[k]);
        const int ql = (int)(jb->q_off[k + 1] - jb->q_off[k]);
        int len = 0, off = 0;
        swo_score ez;
        char *cg = jb->cigar_out + (size_t)k * jb->cigar_stride;
        memset(cg, 0, (size_t)jb->cigar_stride);
        int rc = jb->mat ? swo_align_matrix(jb->targets + jb->t_off[k], tl, jb->queries + jb->q_off[k], ql, jb->code,
                                            jb->mat, jb->gopen, jb->gext, jb->strategy, cg, jb->cigar_stride, &len,
                                            &off, &ez)
                         : swo_align(jb->targets + jb->t_off[k], tl,
                           jb->queries + jb->q_off[k], ql, jb->match,
                           jb->mismatch, jb->gopen, jb->gext, jb->strategy, cg,
                           jb->cigar_stride, &len, &off, &ez, NULL);
        if (rc != SWO_OK && jb->rc == SWO_OK) jb->rc = rc;
        jb->offset_out[k] = off;
        if (jb->score_out) jb->score_out[k] = ez;
        if (jb->cigar_len) jb->cigar_len[k] = len;
    }
    return NULL;
}

static int align_batch_core(int n, const uint8_t *targets, const int64_t *t_off,
                    const uint8_t *queries, const int64_t *q_off, int match,
                    int mismatch, int gopen, int gext, int strategy,
                    int nthreads, int32_t *offset_out, swo_score *score_out,
                    char *cigar_out, int cigar_stride, int32_t *cigar_len, const uint8_t *code, const int8_t *mat)
{
    if (n < 0 || !targets || !queries || !t_off || !q_off || !offset_out ||
        !cigar_out || cigar_stride < 1)
        return SWO_BAD_ARG;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (!mat) swo_normalize_params(&match, &mismatch, &gopen, &gext);
    pthread_t th[256];
    batch_job jobs[256];
    for (int w = 0; w < nthreads; w++) {
        batch_job jb = {n,        w,        nthreads, targets,    queries,
                        t_off,    q_off,    match,    mismatch,   gopen,
                        gext,     strategy, offset_out, cigar_len, score_out,
                        cigar_out, cigar_stride, SWO_OK, code, mat};
        jobs[w] = jb;
    }
    for (int w = 1; w < nthreads; w++)
        pthread_create(&th[w], NULL, batch_worker, &jobs[w]);
    batch_worker(&jobs[0]);
    int rc = jobs[0].rc;
    for (int w = 1; w < nthreads; w++) {
        pthread_join(th[w], NULL);
        if (rc == SWO_OK) rc = jobs[w].rc;
    }
    return rc;
}

int swo_align_batch(int n, const uint8_t *targets, const int64_t *t_off,
                    const uint8_t *queries, const int64_t *q_off, int match,
                    int mismatch, int gopen, int gext, int strategy,
                    int nthreads, int32_t *offset_out, swo_score *score_out,
                    char *cigar_out, int cigar_stride, int32_t *cigar_len)
{
    return align_batch_core(n, targets, t_off, queries, q_off, match, mismatch, gopen, gext, strategy, nthreads,
                            offset_out, score_out, cigar_out, cigar_stride, cigar_len, NULL, NULL);
}

int swo_align_batch_matrix(int n, const uint8_t *targets, const int64_t *t_off, const uint8_t *queries,
                           const int64_t *q_off, const uint8_t *code, const int8_t *mat, int gopen, int gext,
                           int strategy, int nthreads, int32_t *offset_out, swo_score *score_out, char *cigar_out,
                           int cigar_stride, int32_t *cigar_len)
{
    if (!code || !mat) return SWO_BAD_ARG;
    return align_batch_core(n, targets, t_off, queries, q_off, 0, 0, gopen, gext, strategy, nthreads, offset_out,
                            score_out, cigar_out, cigar_stride, cigar_len, code, mat);
}
