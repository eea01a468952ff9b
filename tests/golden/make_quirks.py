"""Regenerates tests/golden/quirks.tsv from the reference compiled in place (oracle/_ref; only where /root/reference exists):
one small case per quirk of SURVEY.md Appendix B, all four overhang strategies, scalar path and (ql >= 8) AVX2 path agreeing.
    python tests/golden/make_quirks.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol

P = (200, -150, 260, 11)
CASES = [
    ("case_sensitive_bytes", b"ACGTACGTAC", b"acgtacgtac", P),          # B6: raw byte compare (sw.cpp:55)
    ("n_matches_n", b"ACGNNNGTAC", b"ACGNNNGTAC", P),                   # B6
    ("n_is_a_mismatch_to_a_base", b"ACGTACGTAC", b"ACGNACGTAC", P),     # B6
    ("last_row_tie_closest_to_diagonal", b"ACGTACGT", b"ACGTACGTACGTACGTACGT", P),  # B1 (sw.cpp:116-127)
    ("last_row_tie_unit_scores", b"AAAA", b"AAAAAAAAAAAA", (1, -1, 1, 1)),          # B1
    ("first_move_is_a_gap_zero_length_m", b"ACGTTTTTACGT", b"ACGTACGTA", P),        # B4 (sw.cpp:180,209,252)
    ("indel_tail_d_else_i", b"GGGGACGTACGT", b"ACGTACGTCCCC", P),                   # B3 (sw.cpp:240-245)
    ("ignore_negative_offset", b"ACGT", b"TTTTACGTACGTGG", P),                      # B2 (sw.cpp:230-233)
    ("indel_score_is_still_row_column_max", b"AAAAAAAAAAAAAAAA", b"AAAAAAAA", (25, -50, 110, 6)),  # B5
]
rows = []
for name, t, q, p in CASES:
    for s in ol.STRATEGIES:
        off, cigar = ol.ref_align(t, q, p, s, avx=False)
        if len(q) >= 8:
            assert ol.ref_align(t, q, p, s, avx=True) == (off, cigar)
        full = ol.ref_full(t, q, p, s)
        rows.append("\t".join([name, t.decode(), q.decode(), *map(str, p), str(s), str(off), cigar, ",".join(map(str, full["score"]))]))
open(os.path.join(HERE, "quirks.tsv"), "w").write("\n".join(rows) + "\n")
print(len(rows), "records")
