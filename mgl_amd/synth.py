"""Seeded synthetic read batches (the bench / fixture workloads of SURVEY.md section 8d).

Everything is numpy-vectorised so a 10 M-pair batch can be produced in seconds;
the generators are deterministic in (seed, arguments).
"""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def rng_for(seed):
    return np.random.Generator(np.random.MT19937(seed))


def random_genome(rng, n):
    """n uniform-random ACGT bases as a uint8 array."""
    return BASES[rng.integers(0, 4, size=n, dtype=np.uint8)]


def illumina_reads(rng, genome, starts, read_len=150, sub=0.01, ins=0.001, dele=0.001):
    """Fixed-length reads copied from ``genome`` at ``starts`` with per-base errors.

    For output base k of a read: with probability ``ins`` it is a random inserted
    base (consumes no genome base); otherwise it copies the next genome base,
    after skipping one genome base with probability ``dele``, and is substituted
    by a different base with probability ``sub``.  Returns uint8 [n, read_len].
    """
    n = len(starts)
    is_ins = rng.random((n, read_len)) < ins
    is_del = rng.random((n, read_len)) < dele
    is_sub = rng.random((n, read_len)) < sub
    copied = ~is_ins
    # genome index for each copied base: start + #copied-before + #deleted-up-to-here
    src = np.cumsum(copied, axis=1) - copied + np.cumsum(is_del & copied, axis=1)
    src = src + np.asarray(starts, dtype=np.int64)[:, None]
    np.clip(src, 0, len(genome) - 1, out=src)
    reads = genome[src]
    code = np.searchsorted(BASES, reads).astype(np.uint8)  # A,C,G,T -> 0..3 (BASES is sorted)
    shift = rng.integers(1, 4, size=(n, read_len), dtype=np.uint8)
    code = np.where(is_sub, (code + shift) & 3, code)
    rnd = rng.integers(0, 4, size=(n, read_len), dtype=np.uint8)
    code = np.where(is_ins, rnd, code)
    return BASES[code]


def config1(seed=42, n_reads=1000, ref_len=1000, read_len=150):
    """BASELINE.json configs[0]: n_reads 150 bp reads, each against the full 1 kb reference."""
    rng = rng_for(seed)
    ref = random_genome(rng, ref_len)
    starts = rng.integers(0, ref_len - read_len + 1, size=n_reads)
    # keep room for a few deletions at the right edge
    starts = np.minimum(starts, ref_len - read_len - 8)
    reads = illumina_reads(rng, ref, starts, read_len)
    return ref, reads


def window_batch(seed, n_pairs, window=256, read_len=150, genome_len=1 << 24):
    """BASELINE.json configs[1]: per-pair target window of ``window`` bases cut from a random
    genome, read = read_len-bp copy starting U[0, window-read_len-8] into the window, Illumina errors.

    Returns (genome uint8[genome_len], win_start int64[n], reads uint8[n, read_len]).
    """
    rng = rng_for(seed)
    genome = random_genome(rng, genome_len)
    win_start = rng.integers(0, genome_len - window, size=n_pairs, dtype=np.int64)
    inner = rng.integers(0, window - read_len - 8 + 1, size=n_pairs, dtype=np.int64)
    reads = illumina_reads(rng, genome, win_start + inner, read_len)
    return genome, win_start, reads


def ont_pair(rng, length, sub=0.05, ins=0.05, dele=0.05):
    """One ONT-style pair: random target of ``length`` and a noisy copy (variable length)."""
    t = random_genome(rng, length)
    out = []
    code_t = np.searchsorted(BASES, t)
    r = rng.random((length, 3))
    extra = rng.integers(0, 4, size=length)
    shift = rng.integers(1, 4, size=length)
    for k in range(length):
        if r[k, 0] < ins:
            out.append(extra[k])
        if r[k, 1] < dele:
            continue
        c = code_t[k]
        if r[k, 2] < sub:
            c = (c + shift[k]) & 3
        out.append(c)
    q = BASES[np.asarray(out, dtype=np.int64)]
    return t, q
