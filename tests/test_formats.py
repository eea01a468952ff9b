"""Wire / on-disk formats (mgl_amd/formats.py): FASTA / FASTQ round trips, the BAM reader on the reference repo's
own test resource (kept as a data fixture), CIGAR text <-> BAM binary."""
import os

import numpy as np
import pytest

import golden_io
from mgl_amd import formats, synth

BAM = os.path.join(golden_io.GOLDEN_DIR, "HiSeq.1mb.1RG.2k_lines.bam")


def test_fasta_fastq_round_trip(tmp_path):
    rng = synth.rng_for(3)
    recs = [("chr%d" % k, "len=%d" % n, synth.random_genome(rng, n).tobytes()) for k, n in enumerate((1, 79, 80, 81, 1000))]
    for name in ("a.fa", "a.fa.gz"):
        path = tmp_path / name
        formats.write_fasta(path, recs, width=80)
        assert list(formats.read_fasta(path)) == recs
    reads = [("r%d" % k, "pos=%d" % (k * 7), synth.random_genome(rng, 150).tobytes(), bytes(rng.integers(35, 74, size=150, dtype=np.uint8)))
             for k in range(50)]
    for name in ("r.fq", "r.fq.gz"):
        path = tmp_path / name
        formats.write_fastq(path, reads)
        assert list(formats.read_fastq(path)) == reads
    bad = tmp_path / "bad.fq"
    bad.write_text("@x\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        list(formats.read_fastq(bad))


def test_cigar_text_binary():
    for text in ("6M1D6M", "4S4M6S", "1D6M5D6M1D", "101M", "1M2D100M", "14M"):
        el = formats.cigar_text_to_elements(text)
        words = formats.cigar_elements_to_binary(el)
        assert formats.cigar_binary_to_text(words) == text
    assert list(formats.cigar_elements_to_binary([(6, "M"), (1, "I"), (2, "D"), (3, "S")])) == [0x60, 0x11, 0x22, 0x34]
    with pytest.raises(ValueError):
        formats.cigar_text_to_elements("6M1Q")


def test_bam_reader_and_reference_reconstruction():
    text, refs, recs = formats.read_bam(BAM)
    assert text.startswith("@HD") and len(refs) == 45 and refs[1][0] == "chr1"
    assert len(recs) == 1677 and all(len(r.seq) == 101 == len(r.qual) for r in recs)
    assert recs[0].cigar == [(101, "M")] and recs[0].pos == 10069920 and recs[0].flag == 99
    # coordinate sorted
    keys = [(r.ref_id if r.ref_id >= 0 else 1 << 30, r.pos) for r in recs]
    assert keys == sorted(keys)
    n = 0
    for r in recs:
        rebuilt = formats.reference_under_read(r)
        if rebuilt is None:
            continue
        ref, edits = rebuilt
        if "OC" not in r.tags:
            assert len(ref) == sum(k for k, op in r.cigar if op in "MDN=X")
            assert edits == r.tags["NM"], (r.name, edits, r.tags["NM"])  # the aligner's own edit distance
        n += 1
    assert n >= 1650
    ts, qs, used = formats.bam_pairs(BAM, window=256)
    assert len(ts) == len(qs) == len(used) > 1600 and all(len(t) == 256 for t in ts)
    # deterministic in the seed
    assert formats.bam_pairs(BAM, window=256)[0] == ts


def test_bam_golden_suite_matches_fixture_inputs():
    """The 'bam' golden suite was generated from exactly these pairs (make_golden.py suite_bam)."""
    rows = golden_io.load("bam")
    ts, qs, _ = formats.bam_pairs(BAM)
    assert [(g.t, g.q) for g in rows if g.suite == "bam"] == list(zip(ts, qs))
    ts, qs, _ = formats.bam_pairs(BAM, window=256)
    assert [(g.t, g.q) for g in rows if g.suite == "bamwin"] == list(zip(ts, qs))
