"""The CPU restatement (oracle/sw_oracle.c) against the golden vectors produced by the
compiled reference, and -- where oracle/_ref is present -- against the reference live."""
import hashlib

import numpy as np
import pytest

import golden_io
import oracle_lib as ol


def check(g, r):
    assert r["offset"] == g.offset, (g.t, g.q, g.params, g.strategy)
    if g.cigar.startswith("sha1:"):
        assert "sha1:" + hashlib.sha1(r["cigar"].encode()).hexdigest() == g.cigar
    else:
        assert r["cigar"] == g.cigar, (g.t, g.q, g.params, g.strategy)
    assert r["score"] == g.score, (g.t, g.q, g.params, g.strategy)
    assert r["crc"] == g.crc, (g.t, g.q, g.params, g.strategy)


@pytest.mark.parametrize("suite", golden_io.SUITES)
def test_oracle_matches_golden(suite):
    rows = golden_io.load(suite)
    assert rows
    for g in rows:
        check(g, ol.oracle_align(g.t, g.q, g.params, g.strategy))


def test_golden_counts():
    # the fixture inventory promised in DESIGN.md
    n = {s: len(golden_io.load(s)) for s in golden_io.SUITES}
    assert n["known"] == 24 and n["tiny"] == 14880 and n["random"] == 2000
    assert n["config1"] == 1000 and n["window"] == 512 and n["long"] == 9 and n["long2"] == 10 and n["long3"] == 6


def test_survey_known_answers():
    # SURVEY.md section 8a known answers (captured from the reference at survey time)
    P = (200, -150, 260, 11)
    r = ol.oracle_align(b"ACGTACGTACGTTTGACCA", b"CGTACGTTGACC", P, ol.SOFTCLIP)
    assert (r["offset"], r["cigar"], r["score"]) == (5, "6M1D6M", (2140, 18, 2140, 18, 12, 0))
    r = ol.oracle_align(b"ACGT", b"TTTTACGTACGTGG", P, ol.IGNORE)
    assert (r["offset"], r["cigar"]) == (-4, "14M")
    r = ol.oracle_align(b"ACGT", b"TTTTACGTACGTGG", P, ol.SOFTCLIP)
    assert (r["offset"], r["cigar"], r["score"]) == (0, "4S4M6S", (529, 4, 800, 4, 8, 6))
    r = ol.oracle_align(b"GATTACA", b"TTAC", P, ol.INDEL)
    assert (r["offset"], r["cigar"]) == (0, "2D4M1D")


def test_param_normalisation():
    import ctypes as C

    v = [C.c_int(x) for x in (-200, 150, -260, -11)]
    ol.oracle().swo_normalize_params(*[C.byref(x) for x in v])
    assert [x.value for x in v] == [200, -150, 260, 11]


def test_batch_driver_threads_agree():
    rows = golden_io.load("window")[:64]
    ts, qs = [g.t for g in rows], [g.q for g in rows]
    off1, sc1, cg1 = ol.oracle_align_batch(ts, qs, rows[0].params, ol.SOFTCLIP, nthreads=1)
    off4, sc4, cg4 = ol.oracle_align_batch(ts, qs, rows[0].params, ol.SOFTCLIP, nthreads=4)
    assert (off1 == off4).all() and (sc1 == sc4).all() and cg1 == cg4
    for k, g in enumerate(rows):
        if g.strategy == ol.SOFTCLIP:
            assert off1[k] == g.offset and cg1[k] == g.cigar and tuple(sc1[k]) == g.score


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference_fuzz():
    rng = np.random.default_rng(2718)
    psets = [(200, -150, 260, 11), (1, -1, 1, 1), (5, -4, 10, 1), (3, -1, 4, 3)]
    for k in range(600):
        tl, ql = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        alpha = np.frombuffer(b"ACGT" if k % 2 else b"AG", np.uint8)
        t = alpha[rng.integers(0, len(alpha), tl)].tobytes()
        q = alpha[rng.integers(0, len(alpha), ql)].tobytes()
        p, s = psets[k % 4], ol.STRATEGIES[(k // 4) % 4]
        a = ol.oracle_align(t, q, p, s, want_btr=True)
        b = ol.ref_full(t, q, p, s, want_btr=True)
        assert (a["offset"], a["cigar"], a["score"]) == (b["offset"], b["cigar"], b["score"])
        assert (a["btr"][1:, 1:] == b["btr"][1:, 1:]).all()
        if ql >= 8:
            assert ol.ref_align(t, q, p, s, avx=True) == (a["offset"], a["cigar"])
