// dropin_caller.cpp -- a caller written against the REFERENCE's C++ API names
// (align_avx / align_scalar / calculateMatrix / calculateCigar, swParameters, ScoreMax; sw_avx.h:6,
// sw_scalar.h:7-9 under /root/reference/src/main/native/mgl_sw/), compiled against include/mgl_sw.hpp
// and linked with libmgl_sw_hip.so.  Reads "t q match mismatch open ext strategy" lines on stdin and
// prints, per line:  <offset> <cigar> <offset_via_matrix> <cigar_via_matrix> <6 ScoreMax fields> <matrix crc32>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "mgl_sw.hpp"

static uint32_t crc32_le(const int *btr, int tl, int ql)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
        ready = true;
    }
    uint32_t crc = 0xFFFFFFFFu;
    for (int i = 1; i <= tl; i++)
        for (int j = 1; j <= ql; j++) {
            uint32_t v = (uint32_t)btr[(size_t)i * (ql + 1) + j];
            for (int b = 0; b < 4; b++, v >>= 8) crc = table[(crc ^ (v & 0xFF)) & 0xFF] ^ (crc >> 8);
        }
    return crc ^ 0xFFFFFFFFu;
}

int main()
{
    std::string t, q;
    swParameters p;
    int strategy;
    while (std::cin >> t >> q >> p.sc_match >> p.sc_mismatch >> p.g_open >> p.g_ext >> strategy) {
        const int tl = (int)t.size(), ql = (int)q.size();
        std::string cigar_a, cigar_s, cigar_m;
        const int off_a = align_avx(t.data(), tl, q.data(), ql, p, strategy, &cigar_a);
        const int off_s = align_scalar(t.data(), tl, q.data(), ql, p, strategy, &cigar_s);
        if (off_a != off_s || cigar_a != cigar_s) {
            std::fprintf(stderr, "align_avx != align_scalar\n");
            return 2;
        }
        // the two-step form of sw.cpp:258-272
        std::vector<int> btr((size_t)(tl + 1) * (ql + 1), 0);
        ScoreMax ez;
        calculateMatrix(t.data(), tl, q.data(), ql, btr.data(), p, strategy, &ez);
        const int off_m = calculateCigar(btr.data(), tl + 1, ql + 1, strategy, &ez, &cigar_m);
        // the AVX2 driver's form (sw_avx.cpp:97): traceback on the band-layout matrix, band width 8
        ScoreMax ez_b;
        std::vector<int> banded = calculateMatrix_banded(t.data(), tl, q.data(), ql, p, strategy, &ez_b);
        std::string cigar_b;
        const int off_b = calculateCigar_avx(banded.data(), tl + 1, ql + 1, 8, strategy, &ez_b, &cigar_b);
        if (off_b != off_m || cigar_b != cigar_m || banded[(size_t)bcktrMatrix_index(tl - 1, ql - 1, ql + 7, 8)] != btr[(size_t)tl * (ql + 1) + ql]) {
            std::fprintf(stderr, "calculateCigar_avx != calculateCigar\n");
            return 3;
        }
        // the AVX2 driver itself, band by band over the caller's arrays (sw_avx.cpp:16-97): calculateMatrix_avx per band of
        // eight rows, the last-row scan over score[], calculateCigar_avx -- must land on the same alignment
        {
            const int bw = 8, ncol = ql + 1, pad_t = (tl % bw) ? bw - tl % bw : 0;
            std::vector<int> rq((size_t)ql + 2 * bw, 0), et((size_t)tl + pad_t, 0), bt((size_t)(ql + bw - 1) * (tl + pad_t), 0);
            std::vector<int> score((size_t)ncol + bw, 0), step((size_t)ncol + bw, 0), gap((size_t)ql + 2 * bw, 1);
            for (int k = 0; k < ql; k++) rq[(size_t)bw + ql - 1 - k] = (int)q[(size_t)k];
            for (int k = 0; k < tl; k++) et[(size_t)k] = (int)t[(size_t)k];
            const int go = p.g_open < 0 ? -p.g_open : p.g_open, ge = p.g_ext < 0 ? -p.g_ext : p.g_ext;
            for (int k = 0; k < ncol; k++) step[(size_t)k] = -go;
            if ((strategy & SW_OS_INDEL) | (strategy & SW_OS_LEAD_ID))
                for (int k = 1; k < ncol; k++) {
                    score[(size_t)k] = -go - (k - 1) * ge;
                    step[(size_t)k] += -go - (k - 1) * ge;
                }
            ScoreMax ez_d;
            swParameters pn = p; // the driver receives normalised parameters from the JNI layer (.cpp:51-55)
            pn.sc_match = p.sc_match < 0 ? -p.sc_match : p.sc_match;
            pn.sc_mismatch = p.sc_mismatch > 0 ? -p.sc_mismatch : p.sc_mismatch;
            pn.g_open = go;
            pn.g_ext = ge;
            int rows_left = tl;
            for (int band = 0; rows_left > 0; band++) {
                const int rows = rows_left >= bw ? bw : rows_left;
                rows_left -= rows;
                calculateMatrix_avx(et.data(), tl, rq.data(), ql, bt.data(), band, bw, rows, score.data(), step.data(), gap.data(), pn,
                                    strategy, &ez_d);
            }
            ez_d.max = ez_d.mqe;
            ez_d.max_t = ez_d.mqe_t;
            ez_d.max_q = ql;
            for (int k = 1; k < ncol; k++) {
                const int sc = score[(size_t)k];
                const int da = tl - k < 0 ? k - tl : tl - k, db = ez_d.max_t - ez_d.max_q < 0 ? ez_d.max_q - ez_d.max_t : ez_d.max_t - ez_d.max_q;
                if (sc > ez_d.max || (sc == ez_d.max && da < db)) {
                    ez_d.max_t = tl;
                    ez_d.max_q = k;
                    ez_d.max = sc;
                    ez_d.seg_length = ql - k;
                }
            }
            std::string cigar_d;
            const int off_d = calculateCigar_avx(bt.data(), tl + 1, ql + 1, bw, strategy, &ez_d, &cigar_d);
            if (off_d != off_a || cigar_d != cigar_a || ez_d.max != ez.max || ez_d.max_t != ez.max_t || ez_d.max_q != ez.max_q ||
                ez_d.mqe != ez.mqe || ez_d.mqe_t != ez.mqe_t || ez_d.seg_length != ez.seg_length) {
                std::fprintf(stderr, "band-by-band driver (calculateMatrix_avx) != align_avx\n");
                return 4;
            }
        }
        std::printf("%d %s %d %s %d %d %d %d %d %d %u\n", off_a, cigar_a.c_str(), off_m, cigar_m.c_str(), ez.mqe, ez.mqe_t,
                    ez.max, ez.max_t, ez.max_q, ez.seg_length, crc32_le(btr.data(), tl, ql));
    }
    return 0;
}
