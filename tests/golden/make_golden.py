#!/usr/bin/env python3
"""Generate the golden Smith-Waterman vectors from the COMPILED REFERENCE.

Run in the authoring container only (needs oracle/_ref/libmgl_ref.so, built by
``make -C oracle ref`` from the sources under /root/reference):

    python tests/golden/make_golden.py

Each record is produced by the reference's own calculateMatrix + calculateCigar
(sw.cpp:5-255).  For every AVX-eligible case (ql >= 8, the dispatch rule of
..._MicrosoftSmithWaterman.cpp:62) the reference's align_avx (sw_avx.cpp:6) is
also run and asserted to give the same offset and CIGAR as the scalar path, so
a record pins "the reference CPU SIMD path" as well.

Output: tests/golden/*.tsv.gz, tab separated, one header line:
  suite t q match mismatch gopen gext strategy offset cigar mqe mqe_t max max_t max_q seg crc
``crc`` = zlib CRC-32 of the logical backtrack matrix over i=1..tl, j=1..ql.
For the ``long`` suite the cigar column holds ``sha1:<hex>`` of the CIGAR text.
"""
import gzip
import hashlib
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from mgl_amd import synth  # noqa: E402

GATK = (200, -150, 260, 11)
PARAM_SETS = [GATK, (25, -50, 110, 6), (10, -15, 30, 5), (3, -1, 4, 3), (1, -1, 1, 1), (1, -4, 6, 1), (5, -4, 10, 1)]
HEADER = "suite t q match mismatch gopen gext strategy offset cigar mqe mqe_t max max_t max_q seg crc".split()

n_avx_checked = 0


def record(suite, t, q, params, strategy, hash_cigar=False):
    global n_avx_checked
    r = ol.ref_full(t, q, params, strategy)
    # calculateMatrix+calculateCigar must equal align_scalar as a whole
    assert ol.ref_align(t, q, params, strategy, avx=False) == (r["offset"], r["cigar"])
    if len(q) >= 8:
        a = ol.ref_align(t, q, params, strategy, avx=True)
        assert a == (r["offset"], r["cigar"]), ("scalar != avx", t, q, params, strategy, a, r)
        n_avx_checked += 1
    cigar = r["cigar"]
    if hash_cigar:
        cigar = "sha1:" + hashlib.sha1(cigar.encode()).hexdigest()
    return [suite, t.decode("latin1"), q.decode("latin1"), *params, strategy, r["offset"], cigar, *r["score"], r["crc"]]


def write(name, rows):
    path = os.path.join(HERE, name + ".tsv.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(("\t".join(HEADER) + "\n").encode("latin1"))
        for r in rows:
            f.write(("\t".join(str(x) for x in r) + "\n").encode("latin1"))
    print(f"{name}: {len(rows)} records, {os.path.getsize(path)} bytes")


def rand_seq(rng, n, alphabet=b"ACGT"):
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes()


def mutate(rng, s, sub=0.05, ins=0.02, dele=0.02, alphabet=b"ACGT"):
    out = bytearray()
    for c in s:
        r = rng.random(3)
        if r[0] < ins:
            out.append(alphabet[rng.integers(0, len(alphabet))])
        if r[1] < dele:
            continue
        if r[2] < sub:
            c = alphabet[rng.integers(0, len(alphabet))]
        out.append(c)
    return bytes(out) or alphabet[:1]


def suite_known():
    rows = []
    cases = [
        (b"ACGTACGTACGTTTGACCA", b"CGTACGTTGACC", GATK),
        (b"AAACCCGGGTTTACGT", b"CCCGGTTTAC", (25, -50, 110, 6)),
        (b"GATTACA", b"TTAC", GATK),
        (b"ACGT", b"TTTTACGTACGTGG", GATK),
        # raw byte comparison: case sensitive, N == N (sw.cpp:55)
        (b"ACGTNNNNACGTacgtACGT", b"GTNNNNACGTACGTAC", GATK),
        (b"AAAAAAAAAAAAAAAA", b"AAAAAAAA", GATK),
    ]
    for t, q, p in cases:
        for s in ol.STRATEGIES:
            rows.append(record("known", t, q, p, s))
    return rows


def suite_tiny():
    rows = []
    seqs_t = [bytes(x) for n in range(1, 6) for x in itertools.product(b"AC", repeat=n)]
    seqs_q = [bytes(x) for n in range(1, 5) for x in itertools.product(b"AC", repeat=n)]
    for p in (GATK, (1, -1, 1, 1)):
        for t in seqs_t:
            for q in seqs_q:
                for s in ol.STRATEGIES:
                    rows.append(record("tiny", t, q, p, s))
    return rows


def suite_random(n=2000):
    rng = synth.rng_for(1234)
    rows = []
    for k in range(n):
        tl = int(rng.integers(1, 401))
        ql = int(rng.integers(8, 301))
        alphabet = b"ACGT" if k % 3 else b"AC"
        t = rand_seq(rng, tl, alphabet)
        if k % 2:
            # related pair: query is a noisy substring of the target (or vice versa)
            a = int(rng.integers(0, max(1, tl - 1)))
            q = mutate(rng, t[a:a + ql], alphabet=alphabet)
            if len(q) < 8:
                q = q + rand_seq(rng, 8 - len(q), alphabet)
        else:
            q = rand_seq(rng, ql, alphabet)
        p = PARAM_SETS[k % len(PARAM_SETS)]
        s = ol.STRATEGIES[(k // len(PARAM_SETS)) % 4]
        rows.append(record("random", t, q, p, s))
    return rows


def suite_ties(n=400):
    rng = synth.rng_for(99)
    rows = []
    for k in range(n):
        tl = int(rng.integers(1, 80))
        ql = int(rng.integers(1, 80))
        if k % 4 == 0:
            t, q = b"A" * tl, b"A" * ql
        elif k % 4 == 1:
            t, q = b"AC" * (tl // 2 + 1), b"CA" * (ql // 2 + 1)
        elif k % 4 == 2:
            t, q = rand_seq(rng, tl, b"AC"), rand_seq(rng, ql, b"AC")
        else:
            t, q = b"A" * tl, rand_seq(rng, ql, b"AT")
        p = PARAM_SETS[k % len(PARAM_SETS)]
        for s in ol.STRATEGIES:
            rows.append(record("ties", t, q, p, s))
    return rows


def suite_shapes():
    rng = synth.rng_for(7)
    rows = []
    for ql in (8, 9, 15, 16, 17, 63, 64, 65, 150, 151):
        for tl in (1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000):
            t = rand_seq(rng, tl)
            q = mutate(rng, (t * (ql // tl + 2))[:ql + 4])[:ql]
            q = q + rand_seq(rng, ql - len(q))
            for s in ol.STRATEGIES:
                rows.append(record("shapes", t, q, GATK, s))
    return rows


def suite_config1():
    ref, reads = synth.config1()
    t = ref.tobytes()
    return [record("config1", t, reads[k].tobytes(), GATK, ol.SOFTCLIP) for k in range(len(reads))]


def suite_window():
    # BASELINE.json configs[1] shape at fixture size: 256-base windows, 150 bp reads, all strategies
    genome, ws, reads = synth.window_batch(2024, 512, genome_len=1 << 16)
    rows = []
    for k in range(len(reads)):
        t = genome[ws[k]:ws[k] + 256].tobytes()
        rows.append(record("window", t, reads[k].tobytes(), GATK, ol.STRATEGIES[k % 4] if k >= 256 else ol.SOFTCLIP))
    return rows


def suite_long():
    rng = synth.rng_for(5)
    rows = []
    for length in (2000, 2000, 2000, 2000):
        t, q = synth.ont_pair(rng, length, 0.05, 0.05, 0.05)
        for s in (ol.SOFTCLIP, ol.INDEL):
            rows.append(record("ont2k", t.tobytes(), q.tobytes(), GATK, s))
    t, q = synth.ont_pair(rng, 10000, 0.05, 0.05, 0.05)
    rows.append(record("long", t.tobytes(), q.tobytes(), GATK, ol.SOFTCLIP, hash_cigar=True))
    return rows


def suite_long2():
    """Ten more long pairs (ONT-style 5 % substitutions / insertions / deletions), CIGARs hashed: every overhang strategy at
    10 kb x 10 kb, unequal lengths either way, target lengths that are not multiples of 64 (the long-read kernel's stripe
    height) or 32, a second parameter set, and one pair beyond 30 kb."""
    rng = synth.rng_for(11)
    rows = []

    def pair(tl, ql):
        # a noisy copy of the target's middle, cut or padded with random flanks to the lengths asked for
        t, q = synth.ont_pair(rng, max(tl, ql) + 64, 0.05, 0.05, 0.05)
        t = t[:tl]
        if len(q) >= ql:
            a = (len(q) - ql) // 2
            q = q[a:a + ql]
        else:
            q = np.concatenate([q, synth.random_genome(rng, ql - len(q))])
        return t.tobytes(), q.tobytes()

    t, q = pair(10000, 10000)
    for s in (ol.INDEL, ol.LEAD_INDEL, ol.IGNORE):
        rows.append(record("long2", t, q, GATK, s, hash_cigar=True))
    for tl, ql, params, s in ((12000, 7000, GATK, ol.SOFTCLIP), (6500, 9000, GATK, ol.INDEL), (10007, 10000, GATK, ol.SOFTCLIP),
                              (5000, 11000, GATK, ol.IGNORE), (8001, 8100, (5, -4, 10, 1), ol.LEAD_INDEL),
                              (9999, 4097, (25, -50, 110, 6), ol.SOFTCLIP), (31000, 30500, GATK, ol.SOFTCLIP)):
        t, q = pair(tl, ql)
        rows.append(record("long2", t, q, params, s, hash_cigar=True))
    return rows


def suite_long3():
    """Long pairs whose paths cross the seams of the checkpointed long-read path with LONG gaps (round 3: the strip kernel keeps the
    rows below bands of 60 rows and column checkpoints every 256 columns, the walk recomputes blocks in between): deletions and
    insertions of 70 .. 900 bases laid over band seams and checkpoint columns, several per pair, every strategy."""
    rng = synth.rng_for(33)
    rows = []
    for length, gaps, params, s in ((4200, ((1190, "D", 130), (2500, "I", 300), (3300, "D", 75)), GATK, ol.SOFTCLIP),
                                    (5000, ((600, "I", 70), (1780, "D", 520), (3999, "I", 260)), GATK, ol.INDEL),
                                    (6100, ((2040, "D", 900), (4090, "I", 128)), GATK, ol.LEAD_INDEL),
                                    (4800, ((958, "D", 61), (1918, "D", 62), (2878, "I", 257), (3500, "I", 255)), GATK, ol.IGNORE),
                                    (4500, ((1500, "D", 240), (3000, "I", 240)), (100, -100, 300, 10), ol.SOFTCLIP),
                                    (9000, ((2500, "D", 700), (5200, "I", 512), (7700, "D", 64)), GATK, ol.SOFTCLIP)):
        t, q = synth.ont_pair(rng, length, 0.02, 0.01, 0.01)
        q = q.copy()
        for at, kind, n in sorted(gaps, reverse=True):  # (positions in the read; applied from the back so that they stay put)
            if kind == "D":   # the read lacks n bases of the target: a vertical run
                q = np.concatenate([q[:at], q[at + n:]])
            else:             # the read has n extra bases: a horizontal run
                q = np.concatenate([q[:at], synth.random_genome(rng, n), q[at:]])
        rows.append(record("long3", t.tobytes(), q.tobytes(), params, s, hash_cigar=True))
    return rows


def suite_bam():
    # real Illumina reads: the reference repo's own test resource (src/test/resources/HiSeq.1mb.1RG.2k_lines.bam,
    # kept here as a data fixture); target = the reference bases under the read rebuilt from CIGAR + MD
    # ("bam"), and the same padded to a 256-base window with seeded random flanks ("bamwin", one geometry)
    from mgl_amd import formats
    path = os.path.join(HERE, "HiSeq.1mb.1RG.2k_lines.bam")
    rows = []
    ts, qs, _ = formats.bam_pairs(path)
    for k, (t, q) in enumerate(zip(ts, qs)):
        rows.append(record("bam", t, q, GATK, ol.STRATEGIES[k % 4] if k % 5 == 4 else ol.SOFTCLIP))
    ts, qs, _ = formats.bam_pairs(path, window=256)
    for k, (t, q) in enumerate(zip(ts, qs)):
        rows.append(record("bamwin", t, q, GATK, ol.SOFTCLIP))
    return rows


SUITE_FUNCS = {"known": suite_known, "tiny": suite_tiny, "random": suite_random, "ties": suite_ties,
               "shapes": suite_shapes, "config1": suite_config1, "window": suite_window, "long": suite_long, "long2": suite_long2,
               "long3": suite_long3, "bam": suite_bam}


def main():
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle ref"
    for name in (sys.argv[1:] or list(SUITE_FUNCS)):
        write(name, SUITE_FUNCS[name]())
    print("AVX2 path asserted equal to scalar on", n_avx_checked, "records")


if __name__ == "__main__":
    main()
