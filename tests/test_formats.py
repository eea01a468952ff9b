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


def test_database_search_layout_on_the_host():
    """mgl_amd.protein.DatabaseSearch (no GPU: CPU tensors): the tiles of `shared` are aligned blocks of 128 pairs with ONE target and one
    query length, longest database sequence first; `rest` holds the queries beyond whole tiles in blocks of eight of one geometry; `long`
    the sequences beyond shared_max_tl; where() is a bijection onto the three batches' pairs and names the right (target, query)."""
    import torch

    from mgl_amd import protein

    rng = np.random.default_rng(3)
    lens = rng.integers(5, 900, 37)
    db_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    db = protein.random_proteins(rng, 1, int(db_off[-1]))[0]
    Q, QL = 272, 33                       # two whole tiles + 16 queries per database sequence
    queries = protein.random_proteins(rng, Q, QL)
    ds = protein.DatabaseSearch(db, db_off, queries, torch.device("cpu"), shared_max_tl=600)
    n_long = int((lens > 600).sum())
    assert ds.Qs == 256 and ds.Qr == 16 and ds.shared.n == (37 - n_long) * 256 and ds.rest.n == (37 - n_long) * 16 and ds.long.n == n_long * Q
    t_off, t_len, q_len = ds.shared.t_off.numpy().reshape(-1, 128), ds.shared.t_len.numpy().reshape(-1, 128), ds.shared.q_len.numpy().reshape(-1, 128)
    assert (t_off == t_off[:, :1]).all() and (t_len == t_len[:, :1]).all() and (q_len == QL).all()      # the promise of MGL_SW_FLAG_SHARED_TARGET
    assert (np.diff(t_len[:, 0]) <= 0).all() and t_len.max() <= 600                                   # longest first, nothing beyond the bound
    r_len = ds.rest.t_len.numpy().reshape(-1, 8)
    assert (r_len == r_len[:, :1]).all()                                                                # blocks of eight of one geometry
    seen = set()
    for d in range(37):
        for q in (0, 1, 127, 128, 255, 256, 271):
            b, p = ds.where(d, q)
            assert (id(b), p) not in seen
            seen.add((id(b), p))
            assert int(b.t_off[p]) == db_off[d] and int(b.t_len[p]) == lens[d] and int(b.q_off[p]) == q * QL and int(b.q_len[p]) == QL
            assert (b is ds.long) == (lens[d] > 600) and (b is ds.rest) == (lens[d] <= 600 and q >= 256)
    # by default nothing is sent to `long`; a workspace bound sends what would not get a region of its own length on every slot
    assert protein.DatabaseSearch(db, db_off, queries, torch.device("cpu")).long is None
    small = protein.DatabaseSearch(db, db_off, queries, torch.device("cpu"), workspace_bytes=protein.shared_target_region_bytes(320, QL) * 3072)
    bound = protein.shared_target_region_bytes(320, QL)
    assert protein.shared_target_region_bytes(small.shared_max_tl, QL) <= bound < protein.shared_target_region_bytes(small.shared_max_tl + 32, QL)
    assert small.long.n == int((lens > small.shared_max_tl).sum()) * Q
