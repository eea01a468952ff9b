"""Reader for the committed golden vectors (tests/golden/*.tsv.gz, made by make_golden.py)."""
import gzip
import os
from collections import namedtuple

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")

Golden = namedtuple("Golden", "suite t q params strategy offset cigar score crc")

SUITES = ("known", "tiny", "random", "ties", "shapes", "config1", "window", "long", "long2", "long3", "bam")


def load(suite):
    path = os.path.join(GOLDEN_DIR, suite + ".tsv.gz")
    out = []
    with gzip.open(path, "rb") as f:
        header = f.readline().decode("latin1").rstrip("\n").split("\t")
        assert header[0] == "suite" and header[-1] == "crc"
        for line in f:
            c = line.decode("latin1").rstrip("\n").split("\t")
            out.append(Golden(c[0], c[1].encode("latin1"), c[2].encode("latin1"), tuple(int(x) for x in c[3:7]),
                              int(c[7]), int(c[8]), c[9], tuple(int(x) for x in c[10:16]), int(c[16])))
    return out
