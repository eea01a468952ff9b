// mgl_sw.hpp -- the reference's C++ alignment-core API surface over the C ABI of mgl_sw.h.
//
// A translation unit that used to #include "sw_avx.h" / "sw_scalar.h" from mgl
// (/root/reference/src/main/native/mgl_sw/) can include this header instead and link
// libmgl_sw_hip.so: the names, signatures, argument meaning and return values are the
// reference's, the work happens on the MI355X.  Header-only; C++11.
//
//   reference declaration                                   here
//   ------------------------------------------------------  ---------------------------------
//   swParameters, ScoreMax, CigarElement, SW_OS_*, STATE_*,  same names, same layout
//     SW_NEG_INF, resetScoreMax        (sw_common.h:22-62)
//   int align_avx(...)                  (sw_avx.h:6)         -> mgl_sw_align
//   int align_scalar(...)               (sw_scalar.h:9)      -> mgl_sw_align
//   void calculateMatrix(...)           (sw_scalar.h:7)      -> mgl_sw_backtrack_matrix
//   int calculateCigar(...)             (sw_scalar.h:8)      -> mgl_sw_cigar_from_backtrack
//   (new) int align_gpu(..., ScoreMax* ez = nullptr)         -> mgl_sw_align, also returns the score
//
//   void calculateMatrix_avx(...)       (sw_avx.h:7)         -> mgl_sw_band_fill (one band per call, the caller's arrays)
//   int calculateCigar_avx(...)         (sw_avx.h:8)         -> mgl_sw_cigar_from_backtrack on the band layout
//   bcktrMatrix_index                   (sw_avx.h:33-40)     same
//
// Also calculateMatrix_banded: the whole matrix in the AVX2 path's band layout in one call.
//
// Error behaviour: the reference's functions cannot fail.  These throw std::runtime_error
// carrying the mgl_sw_status text when the library reports an error (no GPU, bad length, ...).
#ifndef MGL_SW_HPP
#define MGL_SW_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "mgl_sw.h"

#ifndef SW_COMMON_H /* do not clash if the reference's own header is also in the include path */
#define SW_COMMON_H

#define SW_OS_SOFTCLIP MGL_SW_OS_SOFTCLIP /* sw_common.h:22 */
#define SW_OS_INDEL MGL_SW_OS_INDEL       /* sw_common.h:23 */
#define SW_OS_LEAD_ID MGL_SW_OS_LEAD_ID   /* sw_common.h:24 */
#define SW_OS_IGNORE MGL_SW_OS_IGNORE     /* sw_common.h:25 */

#define STATE_MATCH 'M' /* sw_common.h:27-30 */
#define STATE_INS 'I'
#define STATE_DEL 'D'
#define STATE_CLIP 'S'

#define SW_NEG_INF MGL_SW_NEG_INF /* sw_common.h:33 */

struct ScoreMax { /* sw_common.h:36-40; field order differs from mgl_sw_score, so converted by name */
    int mqe = SW_NEG_INF, mqe_t = -1;
    int max = SW_NEG_INF, max_t = -1, max_q = -1;
    int seg_length = 0;
};

struct swParameters { /* sw_common.h:42-47 */
    int sc_match;
    int sc_mismatch;
    int g_open;
    int g_ext;
};

struct CigarElement { /* sw_common.h:49-56 */
    char state;
    int length;
    CigarElement(char a, int b) : state(a), length(b) {}
};

static inline void resetScoreMax(ScoreMax *ez) /* sw_common.h:58-62 */
{
    ez->mqe_t = ez->max_q = ez->max_t = -1;
    ez->mqe = ez->max = SW_NEG_INF;
}

#endif /* SW_COMMON_H */

namespace mgl_sw_detail {
inline void check(int rc, const char *what)
{
    if (rc != MGL_SW_OK) throw std::runtime_error(std::string(what) + ": " + mgl_sw_strerror(rc));
}
inline void to_ref(const mgl_sw_score &s, ScoreMax *ez)
{
    ez->mqe = s.mqe;
    ez->mqe_t = s.mqe_t;
    ez->max = s.max;
    ez->max_t = s.max_t;
    ez->max_q = s.max_q;
    ez->seg_length = s.seg_length;
}
inline mgl_sw_score from_ref(const ScoreMax &e)
{
    mgl_sw_score s;
    s.mqe = e.mqe;
    s.mqe_t = e.mqe_t;
    s.max = e.max;
    s.max_t = e.max_t;
    s.max_q = e.max_q;
    s.seg_length = e.seg_length;
    return s;
}
} // namespace mgl_sw_detail

// Returns the alignment offset and APPENDS the CIGAR to *result_cigar, like sw.cpp:252 / sw_avx.cpp:427.
inline int align_gpu(const char *tseq, int target_length, const char *qseq, int query_length, swParameters parameters,
                     int strategy, std::string *result_cigar, ScoreMax *ez = nullptr)
{
    // worst case text: one element per row/column plus overhangs, <= 11 characters each
    std::vector<char> buf(64);
    for (;;) {
        int len = 0, off = 0;
        mgl_sw_score sc;
        const int rc = mgl_sw_align(tseq, target_length, qseq, query_length, parameters.sc_match, parameters.sc_mismatch,
                                    parameters.g_open, parameters.g_ext, strategy, buf.data(), (int)buf.size(), &len,
                                    &off, &sc);
        if (rc == MGL_SW_ERR_CIGAR_OVERFLOW && len > (int)buf.size()) {
            buf.resize((size_t)len);
            continue;
        }
        mgl_sw_detail::check(rc, "mgl_sw_align");
        result_cigar->append(buf.data(), (size_t)len);
        if (ez) mgl_sw_detail::to_ref(sc, ez);
        return off;
    }
}

// sw_avx.h:6 -- same signature; there is no AVX2 anywhere behind it.
inline int align_avx(const char *tseq, int target_length, const char *qseq, int query_length, swParameters parameters,
                     int strategy, std::string *result_cigar)
{
    return align_gpu(tseq, target_length, qseq, query_length, parameters, strategy, result_cigar);
}

// sw_scalar.h:9
inline int align_scalar(const char *tseq, int target_length, const char *qseq, int query_length, swParameters parameters,
                        int strategy, std::string *result_cigar)
{
    return align_gpu(tseq, target_length, qseq, query_length, parameters, strategy, result_cigar);
}

// sw_scalar.h:7 -- bcktrack is (target_length+1)*(query_length+1) ints, row-major; row 0 and column 0 are zeroed.
inline void calculateMatrix(const char *target, int target_length, const char *query, int query_length, int *bcktrack,
                            swParameters parameters, int overhangStrategy, ScoreMax *ez)
{
    mgl_sw_score sc;
    mgl_sw_detail::check(mgl_sw_backtrack_matrix(target, target_length, query, query_length, parameters.sc_match,
                                                 parameters.sc_mismatch, parameters.g_open, parameters.g_ext,
                                                 overhangStrategy, bcktrack, &sc),
                         "mgl_sw_backtrack_matrix");
    mgl_sw_detail::to_ref(sc, ez);
}

// sw_scalar.h:8 -- n = target_length + 1, m = query_length + 1; appends to *cigar, returns the offset.
inline int calculateCigar(int *bcktrack, int n, int m, int overhangStrategy, ScoreMax *ez, std::string *cigar)
{
    const mgl_sw_score sc = mgl_sw_detail::from_ref(*ez);
    std::vector<char> buf(64);
    for (;;) {
        int len = 0, off = 0;
        const int rc =
            mgl_sw_cigar_from_backtrack(bcktrack, n - 1, m - 1, overhangStrategy, &sc, buf.data(), (int)buf.size(), &len, &off);
        if (rc == MGL_SW_ERR_CIGAR_OVERFLOW && len > (int)buf.size()) {
            buf.resize((size_t)len);
            continue;
        }
        mgl_sw_detail::check(rc, "mgl_sw_cigar_from_backtrack");
        cigar->append(buf.data(), (size_t)len);
        return off;
    }
}

// sw_avx.h:33-40 -- position of cell (i, j) (0-based) in the AVX2 path's anti-diagonal band layout: bands of bw
// target rows, inside a band bw consecutive ints per anti-diagonal.
inline int bcktrMatrix_index(int i, int j, int n_col, int bw)
{
    const int band = i / bw, J = i % bw, I = j + J;
    return band * bw * n_col + I * bw + J;
}

// The band-layout backtrack matrix of sw_avx.cpp:33-34,173 for callers that hold their traceback in that form:
// (query_length + bw - 1) * ceil(target_length / bw) * bw ints, cells outside the matrix zero (the reference leaves
// garbage there).  Filled from the logical matrix of calculateMatrix().
inline std::vector<int> calculateMatrix_banded(const char *target, int target_length, const char *query, int query_length,
                                               swParameters parameters, int overhangStrategy, ScoreMax *ez, int bw = 8)
{
    std::vector<int> logical((size_t)(target_length + 1) * (size_t)(query_length + 1));
    calculateMatrix(target, target_length, query, query_length, logical.data(), parameters, overhangStrategy, ez);
    const int rows = (target_length + bw - 1) / bw * bw, n_col = query_length + bw - 1;
    std::vector<int> banded((size_t)rows * (size_t)n_col, 0);
    for (int i = 1; i <= target_length; ++i)
        for (int j = 1; j <= query_length; ++j)
            banded[(size_t)bcktrMatrix_index(i - 1, j - 1, n_col, bw)] = logical[(size_t)i * (query_length + 1) + j];
    return banded;
}

// sw_avx.h:7 -- one band of `actual_bw` target rows (band number band_count, default_bw = 8 rows per band) over the
// caller's arrays in the reference's layouts: what align_avx's driver loop calls once per band (sw_avx.cpp:71-80).
inline void calculateMatrix_avx(int *target, int target_length, int *query, int query_length, int *bcktrack, int band_count,
                                int default_bw, int actual_bw, int *score, int *step, int *gap, swParameters parameters,
                                int overhangStrategy, ScoreMax *ez)
{
    mgl_sw_score sc = mgl_sw_detail::from_ref(*ez);
    mgl_sw_detail::check(mgl_sw_band_fill(target, target_length, query, query_length, bcktrack, band_count, default_bw, actual_bw,
                                          score, step, gap, parameters.sc_match, parameters.sc_mismatch, parameters.g_open,
                                          parameters.g_ext, overhangStrategy, &sc),
                         "mgl_sw_band_fill");
    ez->mqe = sc.mqe;
    ez->mqe_t = sc.mqe_t;
}

// sw_avx.h:8 -- traceback on a band-layout matrix (n = target_length + 1, m = query_length + 1, bw = the band
// width it was written with, 8 in the reference); appends to *cigar, returns the offset.
inline int calculateCigar_avx(int *bcktrack, int n, int m, int bw, int overhangStrategy, ScoreMax *ez, std::string *cigar)
{
    const int tl = n - 1, ql = m - 1, n_col = ql + bw - 1;
    std::vector<int> logical((size_t)n * (size_t)m, 0);
    for (int i = 1; i <= tl; ++i)
        for (int j = 1; j <= ql; ++j) logical[(size_t)i * m + j] = bcktrack[bcktrMatrix_index(i - 1, j - 1, n_col, bw)];
    return calculateCigar(logical.data(), n, m, overhangStrategy, ez, cigar);
}

#endif /* MGL_SW_HPP */
