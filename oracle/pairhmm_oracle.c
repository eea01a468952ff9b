/*
 * pairhmm_oracle.c -- CPU restatement of mgl's PairHMM (see pairhmm_oracle.h: TEST INFRASTRUCTURE ONLY,
 * parity pinned by the reference's own known answers).  The scalar path is the specification; the same
 * text is instantiated for float and double through PHO_T, as the reference does with templates.
 * Compile with -ffp-contract=off: the scalar reference multiplies and adds separately.
 */
#include "pairhmm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAX_QUAL 254                      /* Context.h:7 */
#define JAC_TOL 8.0                       /* Context.h:8 */
#define JAC_STEP 0.0001                   /* Context.h:9 */
#define JAC_INV_STEP (1.0 / JAC_STEP)     /* Context.h:10 */
enum { JAC_SIZE = 80001 };               /* Context.h:11: (int)(8.0 / 0.0001) + 1, checked in init_all() */
#define M2M_SIZE (((MAX_QUAL + 1) * (MAX_QUAL + 2)) >> 1)

/* ------------------------------------------------------------------ float context (Context.h:136-175) */
#define PHO_T float
#define PHO_SUFFIX(x) x##_f
#define PHO_POW10_NEG_TENTH(x) powf(10.f, -((float)(x)) / 10.f) /* Context.h:144 */
#define PHO_INITIAL ldexpf(1.f, 120)                             /* Context.h:147 */
#include "pairhmm_oracle_impl.h"
#undef PHO_T
#undef PHO_SUFFIX
#undef PHO_POW10_NEG_TENTH
#undef PHO_INITIAL

/* ------------------------------------------------------------------ double context (Context.h:96-134) */
#define PHO_T double
#define PHO_SUFFIX(x) x##_d
#define PHO_POW10_NEG_TENTH(x) pow(10.0, -((double)(x)) / 10.0) /* Context.h:105 */
#define PHO_INITIAL ldexp(1.0, 1020)                             /* Context.h:108 */
#include "pairhmm_oracle_impl.h"

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_all(void)
{
    if ((int)(JAC_TOL / JAC_STEP) + 1 != JAC_SIZE) abort();
    init_context_f();
    init_context_d();
}

float pho_forward_float(const pho_read *read, const uint8_t *hap, int32_t hap_len)
{
    pthread_once(&g_once, init_all);
    return forward_f(read, hap, hap_len);
}

double pho_forward_double(const pho_read *read, const uint8_t *hap, int32_t hap_len)
{
    pthread_once(&g_once, init_all);
    return forward_d(read, hap, hap_len);
}

/* …PairHmm.cc:131-216: every haplotype starts as "non matching" (the exact-match prefilter is commented
 * out, :140-143, and hap.score starts at 0), so: float unless use_double; float results below MIN_ACCEPTED
 * are recomputed in double; log10(result) - log10(initial constant) */
double pho_log10_likelihood(const pho_read *read, const uint8_t *hap, int32_t hap_len, int use_double, int *used_double)
{
    pthread_once(&g_once, init_all);
    if (!use_double) {
        const double rf = (double)forward_f(read, hap, hap_len);
        if (!(rf < (double)PHO_MIN_ACCEPTED)) {
            if (used_double) *used_double = 0;
            return log10(rf) - (double)log10f(ldexpf(1.f, 120)); /* :185, LOG10_INITIAL_CONSTANT is a float, Context.h:148 */
        }
    }
    if (used_double) *used_double = 1;
    return log10(forward_d(read, hap, hap_len)) - log10(ldexp(1.0, 1020)); /* :207 */
}

typedef struct {
    int64_t n_pairs;
    const uint8_t *reads_data;
    const int64_t *read_off;
    const uint8_t *haps_data;
    const int64_t *hap_off;
    const int32_t *pair_read, *pair_hap;
    double *out;
    int32_t *used_double;
    int use_double, tid, nthreads;
} pair_job;

static void *pair_worker(void *arg)
{
    pair_job *j = (pair_job *)arg;
    /* contiguous blocks of pairs per thread */
    const int64_t lo = j->n_pairs * j->tid / j->nthreads, hi = j->n_pairs * (j->tid + 1) / j->nthreads;
    for (int64_t k = lo; k < hi; ++k) {
        const int32_t r = j->pair_read[k], h = j->pair_hap[k];
        pho_read rd;
        rd.len = (int32_t)(j->read_off[r + 1] - j->read_off[r]);
        const uint8_t *base = j->reads_data + 5 * j->read_off[r];
        rd.bases = base;
        rd.qual = base + rd.len;
        rd.ins = base + 2 * (int64_t)rd.len;
        rd.del = base + 3 * (int64_t)rd.len;
        rd.gcp = base + 4 * (int64_t)rd.len;
        int ud = 0;
        j->out[k] = pho_log10_likelihood(&rd, j->haps_data + j->hap_off[h], (int32_t)(j->hap_off[h + 1] - j->hap_off[h]),
                                         j->use_double, &ud);
        if (j->used_double) j->used_double[k] = ud;
    }
    return NULL;
}

int pho_compute_pairs(int64_t n_pairs, const uint8_t *reads_data, const int64_t *read_off, const uint8_t *haps_data,
                      const int64_t *hap_off, const int32_t *pair_read, const int32_t *pair_hap, double *out,
                      int use_double, int nthreads, int32_t *used_double)
{
    pthread_once(&g_once, init_all);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pair_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) {
        pair_job j = {n_pairs, reads_data, read_off, haps_data, hap_off, pair_read, pair_hap, out, used_double, use_double, t, nthreads};
        jobs[t] = j;
    }
    if (nthreads == 1) {
        pair_worker(&jobs[0]);
        return 0;
    }
    for (int t = 0; t < nthreads; ++t)
        if (pthread_create(&th[t], NULL, pair_worker, &jobs[t]) != 0) return 3;
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return 0;
}

/* …PairHmm.cc:77-128: unpack the JNI buffers, then every read against every haplotype */
int pho_compute_likelihoods(const int32_t *lengths, const uint8_t *reads, const uint8_t *haps, double *out, int use_double,
                            int nthreads)
{
    const int32_t n_reads = lengths[0];
    const int32_t *read_len = lengths + 1;
    const int32_t n_haps = lengths[1 + n_reads];
    const int32_t *hap_len = lengths + 2 + n_reads;
    int64_t *read_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_reads + 1));
    int64_t *hap_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_haps + 1));
    const int64_t n_pairs = (int64_t)n_reads * n_haps;
    int32_t *pr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_pairs ? n_pairs : 1));
    int32_t *ph = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_pairs ? n_pairs : 1));
    if (!read_off || !hap_off || !pr || !ph) return 3;
    read_off[0] = hap_off[0] = 0;
    for (int32_t r = 0; r < n_reads; ++r) read_off[r + 1] = read_off[r] + read_len[r];
    for (int32_t h = 0; h < n_haps; ++h) hap_off[h + 1] = hap_off[h] + hap_len[h];
    for (int32_t r = 0; r < n_reads; ++r)
        for (int32_t h = 0; h < n_haps; ++h) {
            pr[(int64_t)r * n_haps + h] = r;
            ph[(int64_t)r * n_haps + h] = h;
        }
    const int rc = pho_compute_pairs(n_pairs, reads, read_off, haps, hap_off, pr, ph, out, use_double, nthreads, NULL);
    free(read_off);
    free(hap_off);
    free(pr);
    free(ph);
    return rc;
}
