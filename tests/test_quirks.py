"""One regression case per quirk of the reference that SURVEY.md Appendix B lists (raw byte comparison, the last-row
tie rule as coded, negative IGNORE offsets, the INDEL tail, zero-length elements, `max` under INDEL), expected values
from the reference compiled in place (tests/golden/quirks.tsv, tests/golden/make_quirks.py)."""
import os

import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def _rows():
    out = []
    for line in open(os.path.join(HERE, "golden", "quirks.tsv")):
        f = line.rstrip("\n").split("\t")
        out.append(dict(name=f[0], t=f[1].encode(), q=f[2].encode(), params=tuple(int(x) for x in f[3:7]), strategy=int(f[7]),
                        offset=int(f[8]), cigar=f[9], score=tuple(int(x) for x in f[10].split(","))))
    return out


def test_restatement_reproduces_every_quirk():
    rows = _rows()
    assert len(rows) == 36
    for r in rows:
        o = ol.oracle_align(r["t"], r["q"], r["params"], r["strategy"])
        assert (o["offset"], o["cigar"], tuple(o["score"])) == (r["offset"], r["cigar"], r["score"]), r["name"]
    by = {(r["name"], r["strategy"]): r for r in rows}
    # the properties the cases were built for
    assert by[("case_sensitive_bytes", ol.IGNORE)]["score"][2] == -150          # 'a' != 'A': not one match in ten
    assert by[("n_matches_n", ol.SOFTCLIP)]["score"][2] == 2000                 # N == N
    assert by[("ignore_negative_offset", ol.IGNORE)]["offset"] == -4
    assert by[("last_row_tie_closest_to_diagonal", ol.SOFTCLIP)]["cigar"] == "8M12S"
    assert by[("first_move_is_a_gap_zero_length_m", ol.INDEL)]["cigar"].endswith("1I")
    assert by[("indel_tail_d_else_i", ol.INDEL)]["cigar"] == "4D8M4I"
    assert by[("indel_score_is_still_row_column_max", ol.INDEL)]["score"][2] == 200


@pytest.mark.gpu
def test_gpu_reproduces_every_quirk():
    from mgl_amd import smithwaterman as sw

    rows = _rows()
    with sw.MicrosoftSmithWaterman(0) as a:
        # one pair at a time (int32 kernel), then the same pairs as uniform batches of eight (packed kernel)
        for r in rows:
            res = a.align_batch([r["t"]], [r["q"]], r["params"], r["strategy"])
            assert (int(res.offsets[0]), res.cigars[0], tuple(int(x) for x in res.scores[0])) == (r["offset"], r["cigar"], r["score"]), r["name"]
            res = a.align_batch([r["t"]] * 8, [r["q"]] * 8, r["params"], r["strategy"])
            for k in range(8):
                assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (r["offset"], r["cigar"], r["score"]), r["name"]
