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
        std::printf("%d %s %d %s %d %d %d %d %d %d %u\n", off_a, cigar_a.c_str(), off_m, cigar_m.c_str(), ez.mqe, ez.mqe_t,
                    ez.max, ez.max_t, ez.max_q, ez.seg_length, crc32_le(btr.data(), tl, ql));
    }
    return 0;
}
