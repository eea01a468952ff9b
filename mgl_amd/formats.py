"""Wire / on-disk formats either side of the alignment path.

Inputs: FASTA / FASTQ readers and writers (plain or gzip), a minimal BAM reader (BGZF is multi-member gzip,
so the standard library is enough) that also rebuilds the reference bases under each aligned read from
SEQ + CIGAR + MD -- which turns the reference repo's only real sequencing data,
``src/test/resources/HiSeq.1mb.1RG.2k_lines.bam`` (kept as a data fixture in tests/golden/), into realistic
(target, query) pairs.  Outputs: BAM-style binary CIGAR (``len << 4 | op``, what
``MGL_SW_FLAG_BINARY_CIGAR`` makes the kernels emit) <-> the text of sw.cpp:251-252.

The reference moves sequences as ASCII inside a direct ByteBuffer (MicrosoftSmithWaterman.java:73-75) and
CIGARs as ASCII text (…SmithWaterman.cpp:65-69); these helpers are the formats a caller actually holds.
"""
import gzip
import io
import re
import struct
from collections import namedtuple

import numpy as np

BAM_CIGAR_OPS = "MIDNSHP=X"
_SEQ_NT16 = "=ACMGRSVTWYHKDBN"


def _open(path, mode="rt"):
    path = str(path)
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic == b"\x1f\x8b":
        return gzip.open(path, mode)
    return open(path, mode)


# --------------------------------------------------------------------------------------------- FASTA / FASTQ
def read_fasta(path):
    """Yield (name, description, sequence bytes) per record; sequence lines are joined, case kept
    (the aligner compares raw bytes, sw.cpp:55)."""
    name, desc, parts = None, "", []
    with _open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, desc, "".join(parts).encode()
                head = line[1:].split(None, 1)
                name, desc, parts = (head[0] if head else ""), (head[1] if len(head) > 1 else ""), []
            elif line and name is not None:
                parts.append(line)
    if name is not None:
        yield name, desc, "".join(parts).encode()


def write_fasta(path, records, width=80):
    """records: iterable of (name, description, sequence bytes / uint8 array)."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wt") as f:
        for name, desc, seq in records:
            seq = bytes(seq).decode()
            f.write(">" + name + (" " + desc if desc else "") + "\n")
            for k in range(0, len(seq), width):
                f.write(seq[k:k + width] + "\n")


def read_fastq(path):
    """Yield (name, description, sequence bytes, quality bytes) per four-line record."""
    with _open(path) as f:
        while True:
            head = f.readline()
            if not head:
                return
            seq = f.readline().rstrip("\r\n")
            plus = f.readline()
            qual = f.readline().rstrip("\r\n")
            if not head.startswith("@") or not plus.startswith("+") or len(seq) != len(qual):
                raise ValueError("malformed FASTQ record near %r" % head[:40])
            h = head[1:].rstrip("\r\n").split(None, 1)
            yield (h[0] if h else ""), (h[1] if len(h) > 1 else ""), seq.encode(), qual.encode()


def write_fastq(path, records):
    """records: iterable of (name, description, sequence bytes, quality bytes or None -> 'I' * len)."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wt") as f:
        for name, desc, seq, qual in records:
            seq = bytes(seq).decode()
            qual = "I" * len(seq) if qual is None else bytes(qual).decode()
            f.write("@" + name + (" " + desc if desc else "") + "\n" + seq + "\n+\n" + qual + "\n")


# --------------------------------------------------------------------------------------------- CIGAR
def cigar_text_to_elements(text):
    """'6M1D6M' -> [(6, 'M'), (1, 'D'), (6, 'M')]"""
    out = [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", text)]
    if "".join("%d%s" % e for e in out) != text:
        raise ValueError("not a CIGAR: %r" % text)
    return out


def cigar_elements_to_binary(elements):
    """[(len, op)] -> uint32 array, BAM encoding (len << 4 | op code; M=0 I=1 D=2 N=3 S=4 H=5 P=6 '='=7 X=8)."""
    return np.array([(n << 4) | BAM_CIGAR_OPS.index(op) for n, op in elements], dtype=np.uint32)


def cigar_binary_to_text(words):
    return "".join("%d%s" % (int(w) >> 4, BAM_CIGAR_OPS[int(w) & 15]) for w in words)


# --------------------------------------------------------------------------------------------- BAM
BamRecord = namedtuple("BamRecord", "name ref_id pos flag mapq cigar seq qual tags")


def _bam_tags(buf, q, end):
    tags = {}
    while q < end:
        tag, ty = buf[q:q + 2].decode(), chr(buf[q + 2])
        q += 3
        if ty == "A":
            tags[tag], q = chr(buf[q]), q + 1
        elif ty in "cCsSiIf":
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[ty]
            tags[tag], q = struct.unpack_from(fmt, buf, q)[0], q + struct.calcsize(fmt)
        elif ty in "ZH":
            e = buf.index(b"\0", q)
            tags[tag], q = buf[q:e].decode(), e + 1
        elif ty == "B":
            st, cnt = chr(buf[q]), struct.unpack_from("<i", buf, q + 1)[0]
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
            tags[tag] = struct.unpack_from("<%d%s" % (cnt, fmt), buf, q + 5)
            q += 5 + cnt * struct.calcsize(fmt)
        else:
            raise ValueError("unknown BAM tag type %r" % ty)
    return tags


def read_bam(path):
    """Returns (header text, [(reference name, length)], [BamRecord]).  ``cigar`` is a list of (len, op),
    ``seq`` ASCII bytes, ``qual`` raw phred bytes (not +33)."""
    with gzip.open(path, "rb") as f:  # BGZF blocks are gzip members
        buf = f.read()
    if buf[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    text = buf[8:8 + l_text].rstrip(b"\0").decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", buf, p)[0]
        refs.append((buf[p + 4:p + 4 + ln - 1].decode(), struct.unpack_from("<i", buf, p + 4 + ln)[0]))
        p += 8 + ln
    records = []
    while p + 4 <= len(buf):
        size = struct.unpack_from("<i", buf, p)[0]
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", buf, p + 4)
        q = p + 36
        name = buf[q:q + l_name - 1].decode()
        q += l_name
        cigar = [(w >> 4, BAM_CIGAR_OPS[w & 15]) for w in struct.unpack_from("<%dI" % n_cig, buf, q)]
        q += 4 * n_cig
        packed = np.frombuffer(buf, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q)
        codes = np.empty(2 * len(packed), dtype=np.uint8)
        codes[0::2], codes[1::2] = packed >> 4, packed & 15
        seq = "".join(_SEQ_NT16[c] for c in codes[:l_seq]).encode()
        q += (l_seq + 1) // 2
        qual = bytes(buf[q:q + l_seq])
        q += l_seq
        records.append(BamRecord(name, ref_id, pos, flag, mapq, cigar, seq, qual, _bam_tags(buf, q, p + 4 + size)))
        p += 4 + size
    return text, refs, records


def reference_under_read(rec):
    """Rebuild the reference bases a read is aligned to from SEQ + CIGAR + MD (SAM spec section 1.5, MD tag).
    Returns (reference bytes, edit distance implied by the reconstruction), or None without a usable MD tag.
    Indel-realigned records keep the MD of their ORIGINAL alignment (tag OC): that CIGAR is tried second."""
    if rec.tags.get("MD") is None:
        return None
    for cigar in (rec.cigar, cigar_text_to_elements(rec.tags["OC"]) if "OC" in rec.tags else None):
        if cigar is None:
            continue
        try:
            return _rebuild(rec._replace(cigar=cigar))
        except (AssertionError, IndexError):
            pass
    return None


def _rebuild(rec):
    md = rec.tags["MD"]
    # 1) the read bases that sit on reference positions (M/=/X), in reference order, with deletions as gaps
    aligned, k = [], 0
    for n, op in rec.cigar:
        if op in "M=X":
            aligned.extend(rec.seq[k:k + n])
            k += n
        elif op in "IS":
            k += n
        elif op in "DN":
            aligned.extend([None] * n)
    # 2) walk MD over them: numbers = matches, letters = reference base at a mismatch, ^letters = deleted bases
    out, pos, edits = [], 0, 0
    for num, dele, sub in re.findall(r"(\d+)|\^([A-Za-z]+)|([A-Za-z])", md):
        if num:
            for _ in range(int(num)):
                out.append(aligned[pos])
                pos += 1
        elif dele:
            for ch in dele:
                assert aligned[pos] is None, "MD deletion does not line up with the CIGAR"
                out.append(ord(ch))
                pos += 1
            edits += len(dele)
        else:
            out.append(ord(sub))
            pos += 1
            edits += 1
    assert pos == len(aligned) and None not in out, "MD does not cover the alignment"
    edits += sum(n for n, op in rec.cigar if op == "I")
    return bytes(out), edits


def bam_pairs(path, window=None, seed=7, min_mapq=0):
    """Realistic (target, query) pairs from a BAM with MD tags: query = the read as sequenced, target = the
    reference bases under it; with ``window`` the target is padded to that many bases with seeded random
    flanks (the read then sits at a random offset inside, as in BASELINE configs[1]).
    Returns (targets list of bytes, queries list of bytes, records used)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    _, _, records = read_bam(path)
    ts, qs, used = [], [], []
    for rec in records:
        if rec.flag & 0x4 or rec.mapq < min_mapq:
            continue
        rebuilt = reference_under_read(rec)
        if rebuilt is None:
            continue
        ref = rebuilt[0]
        if window is not None:
            if len(ref) > window:
                continue
            pad = window - len(ref)
            left = int(rng.integers(0, pad + 1))
            flank = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=pad)].tobytes()
            ref = flank[:left] + ref + flank[left:]
        ts.append(ref)
        qs.append(rec.seq)
        used.append(rec)
    return ts, qs, used
