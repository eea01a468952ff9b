"""GPU parity of the PairHMM kernels (through the C ABI of include/mgl_pairhmm.h) against the reference's known
answers and the CPU restatement.  Tolerance: 1e-5 on the log10 likelihood, the reference's own test tolerance
(MicrosoftPairHmmUnitTest.java:55,103); against the restatement the float path is held to 2e-5 relative on the
likelihood itself for pairs computed in float (different summation association: fma contraction) and to 1e-9 in
double."""
import numpy as np
import pytest

import pairhmm_oracle_lib as pol
from mgl_amd import pairhmm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hmm():
    p = pairhmm.MicrosoftPairHmm(0)
    assert p.load(), "libmgl_pairhmm_hip.so could not create a GPU context"
    yield p
    p.done()


@pytest.mark.parametrize("use_double", [False, True])
def test_data_file(hmm, use_double):
    """dataFileTest (:58-117): one read, one haplotype per call, both precisions."""
    hmm.initialize(pairhmm.PairHMMNativeArguments(use_double, 1))
    n = 0
    for hap, rb, q, i, d, c, expected in pol.testdata():
        out = np.zeros(1)
        hmm.computeLikelihoods([pairhmm.ReadDataHolder(rb, q, i, d, c)], [pairhmm.HaplotypeDataHolder(hap)], out)
        assert abs(out[0] - expected) < 1e-5, (hap, rb, out[0], expected)
        n += 1
    assert n == 104


def test_simple(hmm):
    hmm.initialize(None)
    out = [0.0]
    hmm.computeLikelihoods([pairhmm.ReadDataHolder(b"ACGT", b"++++", b"++++", b"++++", b"++++")],
                           [pairhmm.HaplotypeDataHolder(b"ACGT")], out)
    assert abs(out[0] - (-6.022797e-01)) < 1e-5


def _region(rng, n_reads, n_haps, alphabet=b"ACGT", with_n=False, max_read=260, max_hap=420):
    base = rng.choice(list(alphabet), size=max_hap + 50).astype(np.uint8)
    haps = []
    for _ in range(n_haps):
        h = base[: int(rng.integers(1, max_hap + 1))].copy()
        flips = rng.random(len(h)) < 0.02
        h[flips] = rng.choice(list(alphabet), size=int(flips.sum()))
        if with_n and len(h) > 3:
            h[rng.integers(0, len(h))] = ord("N")
        haps.append(pairhmm.HaplotypeDataHolder(h.tobytes()))
    reads = []
    for k in range(n_reads):
        n = int(rng.integers(1, max_read + 1))
        if k % 7 == 3:   # unrelated read: underflows float, rescued in double
            r = rng.choice(list(alphabet), size=n).astype(np.uint8)
        else:
            s = int(rng.integers(0, max(1, len(base) - n)))
            r = base[s:s + n].copy()
            flips = rng.random(n) < 0.03
            r[flips] = rng.choice(list(alphabet), size=int(flips.sum()))
        if with_n and n > 2:
            r[rng.integers(0, n)] = ord("N")
        n = len(r)
        reads.append(pairhmm.ReadDataHolder(r.tobytes(), rng.integers(6, 42, size=n).astype(np.uint8).tobytes(),
                                            rng.integers(20, 46, size=n).astype(np.uint8).tobytes(),
                                            rng.integers(20, 46, size=n).astype(np.uint8).tobytes(),
                                            np.full(n, 10, np.uint8).tobytes()))
    return reads, haps


@pytest.mark.parametrize("rows", [0, 16, 64])
@pytest.mark.parametrize("use_double", [False, True])
def test_regions_against_restatement(hmm, use_double, rows):
    """Whole regions through the JNI-layout entry: every read against every haplotype, ragged lengths from 1 up,
    N bases, reads that need the double rescue; with four pairs per wave (16 rows), one pair per wave (64 rows) and
    the per-batch choice."""
    rng = np.random.default_rng(11)
    hmm.initialize(pairhmm.PairHMMNativeArguments(use_double, 1))
    hmm.set_stripe_rows(rows)
    for trial in range(6):
        reads, haps = _region(rng, int(rng.integers(1, 40)), int(rng.integers(1, 9)), with_n=(trial % 2 == 1))
        got = np.zeros(len(reads) * len(haps))
        hmm.computeLikelihoods(reads, haps, got)
        rd, roff = pairhmm.pack_reads(reads)
        hd, hoff = pairhmm.pack_haps(haps)
        pr = np.repeat(np.arange(len(reads), dtype=np.int32), len(haps))
        ph = np.tile(np.arange(len(haps), dtype=np.int32), len(reads))
        want, used = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, use_double, nthreads=4)
        assert np.isfinite(got).all()
        err = np.abs(got - want)
        # pairs computed in double (all of them in double mode, the rescued ones in float mode): 1e-9 on the log10;
        # pairs computed in float: 1e-5, the reference's own tolerance (fma contraction differs from the scalar code)
        dbl = (used != 0) | use_double
        assert (err[dbl] < 1e-9 * np.maximum(1.0, np.abs(want[dbl]))).all(), err[dbl].max()
        assert (err[~dbl] < 1e-5).all(), err[~dbl].max()
        if not use_double:
            assert used.sum() > 0 or len(reads) < 4
            assert hmm.timing().rescued == int(used.sum())
    hmm.set_stripe_rows(0)


def test_quality_bytes_are_masked_like_the_reference(hmm):
    """compute_prob_scalar.cc:68-86 masks every quality byte with 127: bytes with the top bit set, and qualities of 0,
    behave accordingly (raw bytes 0..255 in all four tracks)."""
    rng = np.random.default_rng(21)
    hmm.initialize(None)
    hap = bytes(rng.choice(list(b"ACGT"), size=90).astype(np.uint8))
    reads = []
    for _ in range(40):
        n = int(rng.integers(1, 80))
        s = int(rng.integers(0, 90 - n + 1))
        q, i, d, c = (rng.integers(0, 256, size=n).astype(np.uint8).tobytes() for _ in range(4))
        reads.append(pairhmm.ReadDataHolder(hap[s:s + n], q, i, d, c))
    got = np.zeros(len(reads))
    hmm.computeLikelihoods(reads, [pairhmm.HaplotypeDataHolder(hap)], got)
    for k, r in enumerate(reads):
        want, used = pol.log10_likelihood(hap, r.readBases, r.readQuals, r.insertionGOP, r.deletionGOP, r.overallGCP)
        masked, _ = pol.log10_likelihood(hap, r.readBases, *(bytes(b & 127 for b in t) for t in
                                                             (r.readQuals, r.insertionGOP, r.deletionGOP, r.overallGCP)))
        assert want == masked
        assert abs(got[k] - want) < (1e-9 * max(1.0, abs(want)) if used else 1e-5), (k, got[k], want)


@pytest.mark.parametrize("rows,max_reads", [(0, (160, 150, 129, 97, 70, 160)), (32, (160, 150, 129, 97, 70, 160)),
                                            (21, (105, 101, 84, 64, 42, 21, 105)), (0, (105, 101, 76))])
def test_two_pairs_per_wave(hmm, rows, max_reads):
    """Reads of up to 160 bases: 32 lanes x up to five rows per pair, two pairs per wave; up to 105 bases: 21 lanes, three
    pairs per wave (forced, and as the per-batch choice); ragged lengths from 1, odd pair counts (a partly filled last
    wave), N bases, pairs that need the double rescue."""
    rng = np.random.default_rng(57)
    hmm.initialize(pairhmm.PairHMMNativeArguments(False, 1))
    hmm.set_stripe_rows(rows)
    for trial, max_read in enumerate(max_reads):
        reads, haps = _region(rng, 2 * int(rng.integers(1, 16)) + 1, 2 * int(rng.integers(0, 4)) + 1, with_n=(trial % 2 == 1),
                              max_read=max_read, max_hap=330)
        got = np.zeros(len(reads) * len(haps))
        hmm.computeLikelihoods(reads, haps, got)
        rd, roff = pairhmm.pack_reads(reads)
        hd, hoff = pairhmm.pack_haps(haps)
        pr = np.repeat(np.arange(len(reads), dtype=np.int32), len(haps))
        ph = np.tile(np.arange(len(haps), dtype=np.int32), len(reads))
        want, used = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, False, nthreads=4)
        assert np.isfinite(got).all()
        err = np.abs(got - want)
        dbl = used != 0
        assert (err[dbl] < 1e-9 * np.maximum(1.0, np.abs(want[dbl]))).all(), err[dbl].max()
        assert (err[~dbl] < 1e-5).all(), err[~dbl].max()
        assert hmm.timing().rescued == int(used.sum())
        # the same region one pair per wave: same cells, same summation order (the compiler may contract the
        # multiply-adds of the two instantiations differently, so equal only to float rounding)
        hmm.set_stripe_rows(64)
        again = np.zeros_like(got)
        hmm.computeLikelihoods(reads, haps, again)
        hmm.set_stripe_rows(rows)
        assert (np.abs(again - got) < 2e-6).all(), np.abs(again - got).max()
    hmm.set_stripe_rows(0)


@pytest.mark.parametrize("use_double", [False, True])
def test_long_reads_and_haplotypes(hmm, use_double):
    """Reads far beyond one stripe (4 rows x 64 lanes): several stripes with the carry ring between them, haplotypes
    near the LDS limit."""
    rng = np.random.default_rng(31)
    hmm.initialize(pairhmm.PairHMMNativeArguments(use_double, 1))
    limit = pairhmm.lib().mgl_pairhmm_max_haplotype_len(0)
    reads, haps = [], []
    for R, H in ((257, 300), (600, 700), (1000, 1500), (1234, limit)):
        hap = rng.choice(list(b"ACGT"), size=H).astype(np.uint8)
        s = int(rng.integers(0, H - R + 1))
        r = hap[s:s + R].copy()
        flips = rng.random(R) < 0.02
        r[flips] = rng.choice(list(b"ACGT"), size=int(flips.sum()))
        reads.append(pairhmm.ReadDataHolder(r.tobytes(), rng.integers(10, 41, size=R).astype(np.uint8).tobytes(),
                                            np.full(R, 45, np.uint8).tobytes(), np.full(R, 45, np.uint8).tobytes(),
                                            np.full(R, 10, np.uint8).tobytes()))
        haps.append(pairhmm.HaplotypeDataHolder(hap.tobytes()))
    rd, roff = pairhmm.pack_reads(reads)
    hd, hoff = pairhmm.pack_haps(haps)
    pr = np.arange(4, dtype=np.int32)
    ph = np.arange(4, dtype=np.int32)
    got = hmm.compute_pairs(rd, roff, hd, hoff, pr, ph)
    want, used = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, use_double, 4)
    for k in range(4):
        tol = 1e-9 * max(1.0, abs(want[k])) if (use_double or used[k]) else 1e-5
        assert abs(got[k] - want[k]) < tol, (k, got[k], want[k], used[k])


def test_large_host_call_takes_the_copy_path(hmm):
    """More than 1 MiB of inputs: the host entry copies each array to HBM instead of staging one pinned buffer
    (pairhmm_capi.cpp); same answers."""
    rng = np.random.default_rng(77)
    hmm.initialize(pairhmm.PairHMMNativeArguments(False, 1))
    reads, haps = _region(rng, 4000, 4, max_read=160, max_hap=330)
    rd, roff = pairhmm.pack_reads(reads)
    hd, hoff = pairhmm.pack_haps(haps)
    assert rd.nbytes > (1 << 20)
    pr = np.arange(len(reads), dtype=np.int32)
    ph = (pr % len(haps)).astype(np.int32)
    got = hmm.compute_pairs(rd, roff, hd, hoff, pr, ph)
    want, used = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, False, nthreads=4)
    err = np.abs(got - want)
    dbl = used != 0
    assert (err[dbl] < 1e-9 * np.maximum(1.0, np.abs(want[dbl]))).all()
    assert (err[~dbl] < 1e-5).all()
    assert hmm.timing().rescued == int(used.sum())
    # and a slice of it through the staged path gives the same numbers (a small call runs one pair per wave, the large
    # one two: equal to the rounding of differently contracted multiply-adds)
    small = hmm.compute_pairs(rd, roff, hd, hoff, pr[:64], ph[:64])
    assert (np.abs(small - got[:64]) < 2e-6).all()


def test_pair_list_and_errors(hmm):
    hmm.initialize(None)
    rng = np.random.default_rng(3)
    reads, haps = _region(rng, 12, 5)
    rd, roff = pairhmm.pack_reads(reads)
    hd, hoff = pairhmm.pack_haps(haps)
    pr = rng.integers(0, 12, size=200).astype(np.int32)
    ph = rng.integers(0, 5, size=200).astype(np.int32)
    got = hmm.compute_pairs(rd, roff, hd, hoff, pr, ph)
    want, _ = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, False, 4)
    assert np.abs(got - want).max() < 1e-5
    # same pairs in another order give the same numbers (no cross-pair state)
    perm = rng.permutation(200)
    again = hmm.compute_pairs(rd, roff, hd, hoff, pr[perm], ph[perm])
    assert (again == got[perm]).all()
    with pytest.raises(pairhmm.PairHmmError) as e:
        hmm.compute_pairs(rd, roff, hd, hoff, np.array([12], np.int32), np.array([0], np.int32))
    assert e.value.status == pairhmm.ERR_BAD_ARG
    long_hap = np.full(pairhmm.lib().mgl_pairhmm_max_haplotype_len(0) + 1, ord("A"), np.uint8)
    with pytest.raises(pairhmm.PairHmmError) as e:
        hmm.compute_pairs(rd, roff, long_hap, np.array([0, len(long_hap)], np.int64), np.array([0], np.int32), np.array([0], np.int32))
    assert e.value.status == pairhmm.ERR_UNSUPPORTED


def test_device_resident_entry(hmm):
    import torch

    hmm.initialize(None)
    rng = np.random.default_rng(9)
    reads, haps = _region(rng, 64, 8)
    rd, roff = pairhmm.pack_reads(reads)
    hd, hoff = pairhmm.pack_haps(haps)
    pr = np.repeat(np.arange(64, dtype=np.int32), 8)
    ph = np.tile(np.arange(8, dtype=np.int32), 64)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = torch.zeros(len(pr), dtype=torch.float64, device=dev)
    used = torch.zeros(len(pr), dtype=torch.int32, device=dev)
    hmm.compute_pairs_device(t(rd), t(roff), t(hd), t(hoff), t(pr), t(ph), int(np.diff(roff).max()), int(np.diff(hoff).max()), out, used)
    torch.cuda.synchronize()
    host = hmm.compute_pairs(rd, roff, hd, hoff, pr, ph)
    assert (out.cpu().numpy() == host).all()
    want, wused = pol.compute_pairs(rd, roff, hd, hoff, pr, ph, False, 4)
    assert (used.cpu().numpy() == wused).all()
