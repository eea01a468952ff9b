/*
 * ref_wrap.cpp -- thin extern "C" face over the reference's own alignment
 * core so that Python (ctypes) can call it.  TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code: it #includes the reference headers
 * from /root/reference (never copied into this repo) and is linked with the
 * reference's sw.cpp / sw_avx.cpp compiled where they lie (oracle/Makefile).
 * The resulting oracle/_ref/libmgl_ref.so is git-ignored.
 */
#include "sw_scalar.h" /* /root/reference/src/main/native/mgl_sw/sw_scalar.h:7-9 */
#include "sw_avx.h"    /* /root/reference/src/main/native/mgl_sw/sw_avx.h:6 */

#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
swParameters make_params(int match, int mismatch, int gopen, int gext)
{
    /* same normalisation as the JNI boundary (..._MicrosoftSmithWaterman.cpp:51-55) */
    swParameters p;
    p.sc_match = match > 0 ? match : -match;
    p.sc_mismatch = mismatch < 0 ? mismatch : -mismatch;
    p.g_open = gopen > 0 ? gopen : -gopen;
    p.g_ext = gext > 0 ? gext : -gext;
    return p;
}
int emit(const std::string &s, char *cigar, int cap, int *len)
{
    *len = (int)s.size();
    if ((int)s.size() > cap) return 2;
    memcpy(cigar, s.data(), s.size());
    return 0;
}
} // namespace

extern "C" {

int ref_align_scalar(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen,
                     int gext, int strategy, char *cigar, int cap, int *len, int *offset)
{
    std::string s;
    *offset = align_scalar(t, tl, q, ql, make_params(match, mismatch, gopen, gext), strategy, &s);
    return emit(s, cigar, cap, len);
}

int ref_align_avx(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen,
                  int gext, int strategy, char *cigar, int cap, int *len, int *offset)
{
    std::string s;
    *offset = align_avx(t, tl, q, ql, make_params(match, mismatch, gopen, gext), strategy, &s);
    return emit(s, cigar, cap, len);
}

/* calculateMatrix (sw.cpp:5) on a caller-provided zeroed (tl+1)*(ql+1) matrix;
 * ez6 = {mqe, mqe_t, max, max_t, max_q, seg_length} */
void ref_calculate_matrix(const char *t, int tl, const char *q, int ql, int match, int mismatch,
                          int gopen, int gext, int strategy, int *btr, int *ez6)
{
    ScoreMax ez;
    calculateMatrix(t, tl, q, ql, btr, make_params(match, mismatch, gopen, gext), strategy, &ez);
    ez6[0] = ez.mqe;
    ez6[1] = ez.mqe_t;
    ez6[2] = ez.max;
    ez6[3] = ez.max_t;
    ez6[4] = ez.max_q;
    ez6[5] = ez.seg_length;
}

/* calculateCigar (sw.cpp:149) on a matrix produced by ref_calculate_matrix */
int ref_calculate_cigar(int *btr, int tl, int ql, int strategy, const int *ez6, char *cigar, int cap,
                        int *len, int *offset)
{
    ScoreMax ez;
    ez.mqe = ez6[0];
    ez.mqe_t = ez6[1];
    ez.max = ez6[2];
    ez.max_t = ez6[3];
    ez.max_q = ez6[4];
    ez.seg_length = ez6[5];
    std::string s;
    *offset = calculateCigar(btr, tl + 1, ql + 1, strategy, &ez, &s);
    return emit(s, cigar, cap, len);
}

/* Whole batch through the reference's dispatch rule
 * (..._MicrosoftSmithWaterman.cpp:62-70: AVX2 when ql >= 8, else scalar),
 * one pair per task over nthreads std::threads.  use_avx=0 forces scalar. */
int ref_align_batch(int n, const char *targets, const long long *t_off, const char *queries,
                    const long long *q_off, int match, int mismatch, int gopen, int gext, int strategy,
                    int use_avx, int nthreads, int *offset_out, char *cigar_out, int cigar_stride,
                    int *cigar_len)
{
    if (nthreads < 1) nthreads = 1;
    const swParameters p = make_params(match, mismatch, gopen, gext);
    std::atomic<int> next(0), rc(0);
    auto work = [&]() {
        for (;;) {
            const int base = next.fetch_add(64);
            if (base >= n) break;
            const int end = base + 64 < n ? base + 64 : n;
            for (int k = base; k < end; k++) {
                const int tl = (int)(t_off[k + 1] - t_off[k]), ql = (int)(q_off[k + 1] - q_off[k]);
                std::string s;
                int off;
                if (use_avx && __builtin_cpu_supports("avx2") && ql >= 8)
                    off = align_avx(targets + t_off[k], tl, queries + q_off[k], ql, p, strategy, &s);
                else
                    off = align_scalar(targets + t_off[k], tl, queries + q_off[k], ql, p, strategy, &s);
                offset_out[k] = off;
                char *cg = cigar_out + (size_t)k * cigar_stride;
                memset(cg, 0, cigar_stride);
                int len;
                if (emit(s, cg, cigar_stride, &len)) rc = 2;
                if (cigar_len) cigar_len[k] = len;
            }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nthreads; w++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    return rc;
}

int ref_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

/* One band through the reference's calculateMatrix_avx (sw_avx.h:7, sw_avx.cpp:110-322) on the caller's arrays, laid
 * out as align_avx lays them out (sw_avx.cpp:16-49); mqe2 = {mqe, mqe_t} in and out. */
void ref_band_fill(int *target, int tl, int *query_rev, int ql, int *bcktrack, int band_count, int default_bw, int actual_bw,
                   int *score, int *step, int *gap, int match, int mismatch, int gopen, int gext, int strategy, int *mqe2)
{
    ScoreMax ez;
    ez.mqe = mqe2[0];
    ez.mqe_t = mqe2[1];
    calculateMatrix_avx(target, tl, query_rev, ql, bcktrack, band_count, default_bw, actual_bw, score, step, gap,
                        make_params(match, mismatch, gopen, gext), strategy, &ez);
    mqe2[0] = ez.mqe;
    mqe2[1] = ez.mqe_t;
}
}
