"""GPU parity: the HIP path, called through the C ABI, against the golden vectors produced by
the compiled reference and against the CPU restatement (oracle) on fresh seeded inputs.
Bit-exact: offset, CIGAR text, all six ScoreMax fields and the logical backtrack matrix."""
import hashlib
import os
import zlib
from collections import defaultdict

import numpy as np
import pytest

import golden_io
import oracle_lib as ol
from mgl_amd import smithwaterman as sw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aligner():
    a = sw.MicrosoftSmithWaterman(0)
    assert a.load(), "libmgl_sw_hip.so could not create a GPU context"
    # the tests of this module were written for the batched kernels and assert which one ran; sw_small_kernel (small batches on a
    # default context) has its own tests below and is switched on by test_golden_suite[default]
    a.set_small_kernel(1)
    yield a
    a.close()


def run_groups(aligner, rows):
    """Batch rows by (params, strategy) -- one parameter set per batch -- and compare with goldens."""
    groups = defaultdict(list)
    for g in rows:
        groups[(g.params, g.strategy)].append(g)
    n = 0
    for (params, strategy), gs in groups.items():
        res = aligner.align_batch([g.t for g in gs], [g.q for g in gs], params, strategy)
        for k, g in enumerate(gs):
            ctx = (g.suite, g.t, g.q, params, strategy)
            assert int(res.offsets[k]) == g.offset, ctx
            if g.cigar.startswith("sha1:"):
                assert "sha1:" + hashlib.sha1(res.cigars[k].encode()).hexdigest() == g.cigar, ctx
            else:
                assert res.cigars[k] == g.cigar, ctx
            assert tuple(int(x) for x in res.scores[k]) == g.score, ctx
            n += 1
    return n


@pytest.mark.parametrize("small", [0, 1], ids=["default", "no_small_kernel"])
@pytest.mark.parametrize("suite", ["known", "tiny", "random", "ties", "shapes", "config1", "window", "bam"])
def test_golden_suite(aligner, suite, small):
    """default: batches of up to 2 048 pairs whose score matrix fits LDS take sw_small_kernel; with it switched off the same
    batches take the general kernels, as larger ones do"""
    rows = golden_io.load(suite)
    aligner.set_small_kernel(small)
    try:
        assert run_groups(aligner, rows) == len(rows)
    finally:
        aligner.set_small_kernel(1)


def _gapped_pairs(rng, n, tl, ql, ragged=False):
    """pairs with one long gap each (either way), substitutions, and every fifth pair unrelated"""
    alpha = np.frombuffer(b"ACGT", np.uint8)
    ts, qs = [], []
    for k in range(n):
        tlk = int(rng.integers(max(1, tl // 2), tl + 1)) if ragged and k % 3 else tl
        qlk = int(rng.integers(max(1, ql // 2), ql + 1)) if ragged and k % 3 else ql
        t = alpha[rng.integers(0, 4, tlk)]
        if k % 5 == 4:
            q = alpha[rng.integers(0, 4, qlk)]
        else:
            src = np.resize(t[int(rng.integers(0, max(1, tlk // 3))):], qlk + 200).copy()
            gap = int(rng.integers(1, 70))
            at = int(rng.integers(1, max(2, qlk - 1)))
            if k % 5 in (0, 1):
                src = np.concatenate([src[:at], src[at + gap:]])
            elif k % 5 == 2:
                src = np.concatenate([src[:at], alpha[rng.integers(0, 4, gap)], src[at:]])
            sub = rng.random(len(src)) < 0.03
            src[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
            q = src[:qlk]
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    return ts, qs


@pytest.mark.parametrize("tl,ql", [(256, 150), (512, 150), (1, 1), (5, 3), (64, 65), (65, 64), (129, 40), (128, 300), (300, 200), (400, 190),
                                   (37, 500), (2, 700), (511, 1)])
def test_small_kernel(tl, ql):
    """sw_small_kernel (one wave per pair, H kept in LDS, the walk reads every move off the scores): uniform and ragged batches, gaps
    up to 70 cells either way, unrelated pairs, every strategy; parameter sets whose scores fit 16 bits and ones that need the
    32-bit form of the kept matrix; ties between gap lengths (1, -1, 1, 1 and 5, -4, 10, 1 make many)."""
    rng = np.random.default_rng(tl * 11 + ql)
    a = sw.MicrosoftSmithWaterman(0)
    took = 0
    try:
        for ragged in (False, True):
            ts, qs = _gapped_pairs(rng, 120, tl, ql, ragged)
            for params in [(200, -150, 260, 11), (1, -1, 1, 1), (5, -4, 10, 1), (10, -30, 40, 1), (1000, -800, 1500, 50), (3, -3, 0, 0)]:
                for strategy in ol.STRATEGIES:
                    res = a.align_batch(ts, qs, params, strategy)
                    took += a.timing().fill_kernel == 8
                    off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
                    assert (res.offsets == off).all(), (params, strategy, ragged)
                    assert (res.scores == sc).all(), (params, strategy, ragged)
                    assert res.cigars == cg, (params, strategy, ragged)
        # (256 x 150 with scores beyond 16 bits is the largest matrix that fits LDS in 32-bit form; larger ones take the general kernels)
        assert took >= (40 if tl * ql <= 256 * 150 else 32), took
    finally:
        a.close()


def test_parameters_beyond_24_bits_leave_the_one_wave_kernel():
    """small_pair() multiplies by `mismatch - match` and by the gap extension with 24-bit multiplies (one pass on a lone wave where the
    32-bit one takes four).  The reference takes any int (sw.cpp:5-146 computes in plain int), and the library's own bound --
    parameters x lengths < 2^30 -- admits match = 2^24 on a tiny pair: such parameters must go to the kernels that multiply in 32 bits
    (small_mul24_ok), through the batch entry and through the one-pair front-ends alike."""
    rng = np.random.default_rng(24)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    ts = [alpha[rng.integers(0, 4, 20)].tobytes() for _ in range(40)]
    qs = [t[3:15] if k % 3 else alpha[rng.integers(0, 4, 12)].tobytes() for k, t in enumerate(ts)]
    a = sw.MicrosoftSmithWaterman(0)
    try:
        for params in [(1 << 24, -3, 5, 2), (7, -(1 << 24), 5, 2), (5, -3, (1 << 23) + 9, 1 << 23), (1 << 22, -(1 << 22) - 1, 9, 3)]:
            for strategy in ol.STRATEGIES:
                off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=2)
                res = a.align_batch(ts, qs, params, strategy)
                assert a.timing().fill_kernel != 8, "not the one-wave-per-pair kernel"
                assert (res.offsets == off).all() and (res.scores == sc).all() and res.cigars == cg, (params, strategy)
                for k in (0, 1, 5):
                    cigar, offset, ez = sw.align(ts[k], qs[k], params, strategy)
                    assert (offset, cigar, tuple(ez)) == (int(off[k]), cg[k], tuple(int(x) for x in sc[k])), (params, strategy, k)
    finally:
        a.close()


def test_bam_pairs_device_formats(aligner):
    """Real Illumina reads (tests/golden/HiSeq.1mb.1RG.2k_lines.bam) against 256-base windows, through the
    device-resident entry in both wire formats (ASCII, 2-bit packed) and with BAM-style binary CIGAR output:
    identical to the goldens (one geometry, so this is the packed-int16 kernel)."""
    import torch
    from mgl_amd import device_batch, formats

    rows = [g for g in golden_io.load("bam") if g.suite == "bamwin"]
    keep = [g for g in rows if set(g.t) <= set(b"ACGT") and set(g.q) <= set(b"ACGT")]  # 2-bit has no N
    assert len(keep) > 1000
    td, toff = sw.concat([g.t for g in keep])
    qd, qoff = sw.concat([g.q for g in keep])
    b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=512)  # the Java side's 2*max(tl,ql)
    assert b.uniform
    b.run(aligner)
    cg = b.cigar_strings()
    for k, g in enumerate(keep):
        assert (int(b.offsets[k]), cg[k], tuple(int(x) for x in b.scores[k])) == (g.offset, g.cigar, g.score)
    assert aligner.timing().packed16 == 1
    # binary CIGAR == BAM encoding of the same text
    b.run(aligner, binary_cigar=True)
    raw, ln = b.cigars.cpu().numpy(), b.cigar_len.cpu().numpy()
    for k, g in enumerate(keep):
        words = np.frombuffer(raw[k, : ln[k]].tobytes(), dtype="<u4")
        assert (words == formats.cigar_elements_to_binary(formats.cigar_text_to_elements(g.cigar))).all()
        assert formats.cigar_binary_to_text(words) == g.cigar
    # 2-bit packed inputs
    tp = device_batch.pack2bit(td)
    qp = device_batch.pack2bit(qd)
    pb = device_batch.PackedBatch(torch.from_numpy(tp).cuda(), torch.from_numpy(toff[:-1].copy()).cuda(),
                                  torch.from_numpy(np.diff(toff).astype(np.int32)).cuda(), torch.from_numpy(qp).cuda(),
                                  torch.from_numpy(qoff[:-1].copy()).cuda(), torch.from_numpy(np.diff(qoff).astype(np.int32)).cuda(),
                                  256, 101, cigar_stride=512)
    pb.run(aligner)
    torch.cuda.synchronize()
    assert torch.equal(pb.offsets, b.offsets) and torch.equal(pb.scores, b.scores)
    assert pb.cigar_strings() == cg


def test_golden_long(aligner):
    """2 kb ONT-style pairs (full CIGAR) and the 10 kb x 10 kb pair (CIGAR hash): the latter is beyond the LDS
    carve, so its stripe carry lives in the HBM scratch (sw_dp_scratch_kernel)."""
    from mgl_amd import _lib

    rows = golden_io.load("long")
    assert max(len(g.q) for g in rows) > _lib.lib().mgl_sw_max_lds_query_len()
    assert run_groups(aligner, rows) == len(rows) == 9


def test_golden_long2(aligner):
    """Ten more long pairs from the compiled reference (tests/golden/make_golden.py::suite_long2): every overhang strategy
    at 10 kb x 10 kb, unequal lengths either way, target lengths that are no multiple of the stripe height, a second and a
    third parameter set, one pair beyond 30 kb; offset, all six ScoreMax fields, CIGAR by hash.  Default kernel choice (one
    pair per workgroup for these query lengths), then the same through 3 and 16 waves per pair and through the one-pair entry."""
    rows = golden_io.load("long2")
    assert len(rows) == 10 and max(len(g.t) for g in rows) >= 30000
    assert run_groups(aligner, rows) == 10
    assert aligner.timing().fill_kernel in (3, 5, 6)  # (the last group's lengths and parameters decide which long-read kernel)
    for waves in (3, 16):
        forced = sw.MicrosoftSmithWaterman(0)
        forced.set_cooperative(waves)
        assert run_groups(forced, rows) == 10
        forced.close()
    g = rows[5]  # 10 007 x 10 000
    cigar, off, ez = sw.align(g.t, g.q, g.params, g.strategy)
    assert off == g.offset and "sha1:" + hashlib.sha1(cigar.encode()).hexdigest() == g.cigar and tuple(ez) == g.score


def test_golden_long3_gaps_across_the_seams(aligner):
    """Six long pairs from the compiled reference whose paths carry gaps of 61 .. 900 cells laid over the seams of the
    checkpointed long-read path (bands of 60 rows, checkpoint columns every 256 columns): runs that cross several recomputed blocks
    upwards and to the left, every strategy, two parameter sets -- default choice (strip kernel, no flags stored), the same with
    every flag stored, and the workgroup kernels."""
    rows = golden_io.load("long3")
    assert len(rows) == 6
    assert run_groups(aligner, rows) == 6
    assert aligner.timing().fill_kernel == 6 and aligner.slot_layout(0) == 6
    for mode in ("stored", "coop"):
        forced = sw.MicrosoftSmithWaterman(0)
        if mode == "stored":
            forced.set_lane_checkpoint(1)
        else:
            forced.set_strip_kernel(1)
        assert run_groups(forced, rows) == 6
        assert forced.timing().fill_kernel == (6 if mode == "stored" else 5)
        forced.close()


@pytest.mark.parametrize("rows,carry", [(64, 0), (64, 1), (16, 1)])
def test_stripe_rows_and_carry_variants(rows, carry):
    """The int32 fill kernel with 64-row stripes (one pair per wave, wave_shr DPP) and / or the carry in the HBM
    scratch, forced onto ordinary mixed batches: identical to the goldens, traceback included."""
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_precision(32)
    forced.set_stripe_rows(rows)
    forced.set_carry_memory(carry)
    rows_ = golden_io.load("known") + golden_io.load("shapes") + golden_io.load("random")[:500] + golden_io.load("ties")[::9]
    assert run_groups(forced, rows_) == len(rows_)
    gs = [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP][:24]
    forced.align_batch([g.t for g in gs], [g.q for g in gs], gs[0].params, ol.SOFTCLIP)
    for k, g in enumerate(gs):
        btr = forced.expand_slot(k, len(g.t), len(g.q))
        assert zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF == g.crc
    # long pairs (2 kb, and the 10 kb x 10 kb one) through the same variant
    assert run_groups(forced, golden_io.load("long")) == 9
    forced.close()


@pytest.mark.parametrize("waves,precision", [(2, 32), (3, 32), (8, 32), (16, 32), (2, 0), (3, 0), (8, 16), (16, 0)])
def test_cooperative_long_read_kernel(waves, precision):
    """sw_dp_coop_kernel (one pair per workgroup, `waves` waves pipelined over the 64-row stripes, LDS rings between
    them and the HBM row from the last wave back to wave 0) and its 16-bit form sw_dp_coop16_kernel (precision 0 / 16: 128
    rows per wave) forced onto ordinary mixed batches and the long goldens: identical results, traceback included."""
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_cooperative(waves)
    forced.set_precision(precision)
    rows_ = golden_io.load("known") + golden_io.load("shapes") + golden_io.load("random")[:500] + golden_io.load("ties")[::9]
    assert run_groups(forced, rows_) == len(rows_)
    gs = [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP][:24]
    forced.align_batch([g.t for g in gs], [g.q for g in gs], gs[0].params, ol.SOFTCLIP)
    for k, g in enumerate(gs):
        btr = forced.expand_slot(k, len(g.t), len(g.q))
        assert zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF == g.crc
    assert run_groups(forced, golden_io.load("long")) == 9
    forced.close()


def _oracle_rows(pairs, params, strategy):
    return [(t, q, ol.oracle_align(t, q, params, strategy)) for t, q in pairs]


@pytest.mark.parametrize("waves", [2, 5, 16])
def test_cooperative_16bit_kernel_and_its_fallback(waves):
    """sw_dp_coop16_kernel: 128 target rows per wave in packed int16 relative to two moving baselines, a window check at
    every move, and the int32 body for a pair whose window does not fit.
    (a) GATK parameters: every pair stays in 16 bits (layout 3), results = goldens (run by the test above as well);
    (b) scores ten times larger than 16 bits on 2 kb .. 6 kb pairs: still layout 3, identical to the oracle;
    (c) parameters whose windows cannot fit (precision 16 forces the attempt): the pairs come back in layout 0 -- redone in
        32 bits inside the kernel -- and identical to the oracle."""
    from mgl_amd import synth

    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_cooperative(waves)
    gs = [g for g in golden_io.load("long") + golden_io.load("long2") if g.params == (200, -150, 260, 11)]
    assert run_groups(forced, gs) == len(gs)
    assert forced.timing().fill_kernel == 5
    rng = synth.rng_for(4242)
    pairs = [tuple(x.tobytes() for x in synth.ont_pair(rng, n)) for n in (130, 257, 700, 2000, 3100, 6000)]
    pairs += [(pairs[4][0], pairs[2][1]), (pairs[2][0], pairs[5][1])]  # unrelated sequences, unequal lengths
    for strategy in (ol.SOFTCLIP, ol.INDEL, ol.LEAD_INDEL, ol.IGNORE):
        want = _oracle_rows(pairs, (200, -150, 260, 11), strategy)
        res = forced.align_batch([p[0] for p in pairs], [p[1] for p in pairs], (200, -150, 260, 11), strategy, cigar_stride=20000)
        for k, (t, q, o) in enumerate(want):
            assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (strategy, k)
            assert forced.slot_layout(k) == 3
    # (c) windows that cannot fit
    big = (400, -300, 500, 20)
    forced.set_precision(16)
    want = _oracle_rows(pairs, big, ol.SOFTCLIP)
    res = forced.align_batch([p[0] for p in pairs], [p[1] for p in pairs], big, ol.SOFTCLIP, cigar_stride=20000)
    layouts = [forced.slot_layout(k) for k in range(len(pairs))]
    for k, (t, q, o) in enumerate(want):
        assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), k
    assert forced.timing().fill_kernel == 5 and layouts.count(0) >= 4, layouts
    # the same parameters without the forcing: the host does not try (int32 kernel)
    forced.set_precision(0)
    forced.align_batch([p[0] for p in pairs], [p[1] for p in pairs], big, ol.SOFTCLIP, cigar_stride=20000)
    assert forced.timing().fill_kernel == 3
    # backtrack matrices out of the 16-bit layout, cell for cell
    g24 = [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP][:24]
    forced.align_batch([g.t for g in g24], [g.q for g in g24], g24[0].params, ol.SOFTCLIP)
    for k, g in enumerate(g24):
        assert forced.slot_layout(k) == 3
        btr = forced.expand_slot(k, len(g.t), len(g.q))
        assert zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF == g.crc
    forced.close()


@pytest.mark.parametrize("two_bit", [False, True])
def test_device_batch_of_mixed_lengths_is_sorted_on_the_device(two_bit):
    """The same promise for a DEVICE-resident batch (mgl_sw_align_batch_device[_2bit], no flag): the library sorts every chunk
    by geometry with a counting sort on the GPU (sw_regroup_*_kernel), full blocks of eight through the packed kernel, the
    left-over pairs through the int32 kernel, results at the caller's indices; several chunks (small workspace)."""
    import torch
    from mgl_amd import device_batch, synth

    rng = synth.rng_for(78)
    n = 30_000
    genome = synth.random_genome(rng, 1 << 20)
    starts = rng.integers(0, len(genome) - 300, size=n)
    tls = rng.choice([200, 256], size=n)
    reads = synth.illumina_reads(rng, genome, starts + 40, read_len=150, sub=0.02, ins=0.004, dele=0.004)
    qls = rng.integers(100, 151, size=n)
    qls[::101] = rng.integers(1, 40, size=len(qls[::101]))
    tseqs = [genome[s: s + tl].tobytes() for s, tl in zip(starts, tls)]
    qseqs = [r[:q].tobytes() for r, q in zip(reads, qls)]
    td, toff = sw.concat(tseqs)
    qd, qoff = sw.concat(qseqs)
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(160 << 20)  # about four chunks
    if two_bit:
        dev = torch.device("cuda", 0)
        pb = device_batch.PackedBatch(torch.from_numpy(device_batch.pack2bit(td)).to(dev), torch.from_numpy(toff[:-1].copy()).to(dev),
                                      torch.from_numpy(np.diff(toff).astype(np.int32)).to(dev), torch.from_numpy(device_batch.pack2bit(qd)).to(dev),
                                      torch.from_numpy(qoff[:-1].copy()).to(dev), torch.from_numpy(np.diff(qoff).astype(np.int32)).to(dev),
                                      256, 150, cigar_stride=256)
        b = pb
    else:
        b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=256)
        assert not b.uniform
    b.run(a)
    torch.cuda.synchronize()
    tm = a.timing()
    assert tm.packed16 == 1 and tm.dp_launches >= 3
    woff, wsc, wcg = ol.oracle_align_batch(tseqs, qseqs, (200, -150, 260, 11), ol.SOFTCLIP, nthreads=8)
    assert int((b.status != 0).sum()) == 0
    assert (b.offsets.cpu().numpy() == woff).all() and (b.scores.cpu().numpy() == wsc).all()
    cg = b.cigar_strings()
    assert all(cg[k] == wcg[k] for k in range(n))
    # the same batch with the int32 kernel forced: identical bytes
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_precision(32)
    keep = (b.offsets.clone(), b.scores.clone(), b.cigars.clone(), b.cigar_len.clone())
    b.run(forced)
    torch.cuda.synchronize()
    assert forced.timing().packed16 == 0
    assert torch.equal(b.offsets, keep[0]) and torch.equal(b.scores, keep[1]) and torch.equal(b.cigars, keep[2]) and torch.equal(b.cigar_len, keep[3])
    forced.close()
    a.close()


@pytest.mark.parametrize("stored", [False, True, "bytes"], ids=["checkpointed", "stored", "checkpointed_byte_compare"])
def test_strip_kernel_long_reads(stored, monkeypatch):
    """sw_dp16_strip_kernel (one pair per workgroup, one 32-row strip per lane-half, per-strip 16-bit baselines, hand-over by DPP
    and an LDS mailbox) forced onto the long goldens, onto pairs of awkward lengths under every strategy, and onto ordinary
    short pairs (one wave, mostly idle strips): identical results, traceback included.  Both forms: no flags stored -- kept
    rows and checkpoints, sw_strip_ck_walk_kernel recomputes the blocks the path crosses (layout 6, the default) -- and the
    flags of every cell stored (layout 4, what mgl_sw_ctx_expand_slot needs)."""
    from mgl_amd import synth

    if stored == "bytes":  # the byte compare for every pair (default: base codes wherever a pair's target is all ACGT and the query's tables fit the LDS carve)
        monkeypatch.setenv("MGL_SW_DEBUG_STRIP_CODES", "0")
        stored = False
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_strip_kernel(2)
    forced.set_lane_checkpoint(1 if stored else 0)
    # (targets beyond 16 384 rows: several passes over the target, the kernels without stored flags only)
    gs = [g for g in golden_io.load("long") + golden_io.load("long2") + golden_io.load("long3") if g.params == (200, -150, 260, 11) and (len(g.t) <= 16384 or not stored)]
    assert len(gs) >= 17 and (stored or max(len(g.t) for g in gs) > 16384)
    assert run_groups(forced, gs) == len(gs)
    assert forced.timing().fill_kernel == 6
    rng = synth.rng_for(77)
    pairs = [tuple(x.tobytes() for x in synth.ont_pair(rng, n)) for n in (33, 130, 257, 1000, 4097, 5000, 8191, 9000)]
    pairs += [(pairs[5][0], pairs[3][1]), (pairs[3][0], pairs[6][1]), (pairs[7][0][:4100], pairs[7][1][:37])]
    for strategy in (ol.SOFTCLIP, ol.INDEL, ol.LEAD_INDEL, ol.IGNORE):
        res = forced.align_batch([p[0] for p in pairs], [p[1] for p in pairs], (200, -150, 260, 11), strategy, cigar_stride=24000)
        assert forced.timing().fill_kernel == 6
        for k, (t, q) in enumerate(pairs):
            o = ol.oracle_align(t, q, (200, -150, 260, 11), strategy)
            assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (strategy, k)
            assert forced.slot_layout(k) == (4 if stored else 6)
    # the same pairs as 2-bit packed inputs at unaligned base offsets (base codes by construction): identical to the ASCII run
    if not stored:
        import torch
        from collections import namedtuple
        Row = namedtuple("Row", "t q")
        pb = _packed_from_rows([Row(*p) for p in pairs], torch.device("cuda", 0))
        pb.run(forced, (200, -150, 260, 11), ol.IGNORE)
        torch.cuda.synchronize()
        assert forced.timing().fill_kernel == 6 and int((pb.status != 0).sum()) == 0
        off, sc, cg = pb.offsets.cpu().numpy(), pb.scores.cpu().numpy(), pb.cigar_strings()
        for k in range(len(pairs)):
            assert (int(off[k]), cg[k], tuple(int(x) for x in sc[k])) == (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])), k
    # four waves per pair (targets of 12 289 .. 16 384 rows), a short query against a long target and the reverse
    wide = [(synth.random_genome(rng, 15000).tobytes(), synth.random_genome(rng, 1800).tobytes())]
    wide.append((wide[0][0][:16384 - 3], wide[0][0][200:1500]))
    wide.append((wide[0][1][:1700], wide[0][0][:9000]))
    for strategy in (ol.SOFTCLIP, ol.LEAD_INDEL):
        res = forced.align_batch([p[0] for p in wide], [p[1] for p in wide], (200, -150, 260, 11), strategy, cigar_stride=40000)
        assert forced.timing().fill_kernel == 6
        for k, (t, q) in enumerate(wide):
            o = ol.oracle_align(t, q, (200, -150, 260, 11), strategy)
            assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (strategy, k)
    # bytes outside ACGT (sw.cpp:55 compares bytes: N == N, a != A): in the query alone -- the base-code form, whose table for such a
    # byte differs from every code --, in the target -- that pair takes the byte compare --, in both at the same place, lower case
    gn = bytearray(synth.random_genome(rng, 3000).tobytes())
    qn = bytearray(gn[100:2600])
    for x in (5, 700, 701, 702, 2499):
        qn[x] = ord("N")
    tn = bytearray(gn)
    for x in (105, 800, 801, 802, 2599, 2999):
        tn[x] = ord("N")
    odd = [(bytes(gn), bytes(qn)), (bytes(tn), bytes(gn[100:2600])), (bytes(tn), bytes(qn)), (bytes(gn).lower(), bytes(gn[50:2000])), (bytes(gn), bytes(gn[50:2000]).lower()),
           (bytes(gn), bytes(gn[50:1000]) + b"RYKM" + bytes(gn[1004:2000]))]
    for strategy in (ol.SOFTCLIP, ol.INDEL):
        res = forced.align_batch([p[0] for p in odd], [p[1] for p in odd], (200, -150, 260, 11), strategy, cigar_stride=8192)
        assert forced.timing().fill_kernel == 6
        for k, (t, q) in enumerate(odd):
            o = ol.oracle_align(t, q, (200, -150, 260, 11), strategy)
            assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (strategy, k)
    # sequences that stretch a strip's static window (identical, disjoint, homopolymers, a long gap) under two parameter sets that
    # pass strip16_range_ok; sets that do not go to the workgroup kernels by themselves (16-bit with its checked window, or int32)
    g = synth.random_genome(rng, 5000).tobytes()
    adv = [(g, g), (g, synth.random_genome(rng, 4400).tobytes()), (b"A" * 4200, b"A" * 3900), (b"A" * 4100, b"C" * 2000), (g[:2500] + g[3300:], g)]
    for params, kernel in (((200, -150, 260, 11), 6), ((100, -100, 300, 10), 6), ((50, -200, 400, 1), 5), ((400, -300, 500, 20), 3)):
        res = forced.align_batch([p[0] for p in adv], [p[1] for p in adv], params, ol.SOFTCLIP, cigar_stride=16384)
        assert forced.timing().fill_kernel == kernel, (params, forced.timing().fill_kernel)
        for k, (t, q) in enumerate(adv):
            o = ol.oracle_align(t, q, params, ol.SOFTCLIP)
            assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (params, k)
    rows_ = golden_io.load("known") + golden_io.load("shapes")[:200] + [g for g in golden_io.load("random")[:400] if g.params[0] <= 300]
    assert run_groups(forced, rows_) == len(rows_)
    g24 = [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP][:24]
    forced.align_batch([g.t for g in g24], [g.q for g in g24], g24[0].params, ol.SOFTCLIP)
    for k, g in enumerate(g24):
        if not stored:
            with pytest.raises(RuntimeError):
                forced.expand_slot(k, len(g.t), len(g.q))   # no stored traceback to expand
            break
        btr = forced.expand_slot(k, len(g.t), len(g.q))
        assert zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF == g.crc
    forced.close()


def test_cooperative_16bit_equals_32bit_on_many_long_pairs():
    """160 ONT-style pairs of 1.5 kb .. 12 kb (lengths no multiple of anything, both sequences of a pair different in
    length), every overhang strategy: the 16-bit long-read kernel and the int32 kernel must agree on every output byte;
    a sample is checked against the oracle."""
    import torch
    from mgl_amd import device_batch, synth

    rng = synth.rng_for(2024)
    lens = rng.integers(1500, 12001, size=160)
    pairs = [tuple(x.tobytes() for x in synth.ont_pair(rng, int(n))) for n in lens]
    td, toff = sw.concat([p[0] for p in pairs])
    qd, qoff = sw.concat([p[1] for p in pairs])
    b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=2 * 14000)
    a16, a32, astrip = sw.MicrosoftSmithWaterman(0), sw.MicrosoftSmithWaterman(0), sw.MicrosoftSmithWaterman(0)
    a16.set_strip_kernel(1)
    a32.set_precision(32)
    for strategy in (ol.SOFTCLIP, ol.INDEL, ol.LEAD_INDEL, ol.IGNORE):
        b.run(astrip, overhang_strategy=strategy)  # default choice for targets of 4 096 .. 16 384 rows: the strip kernel
        torch.cuda.synchronize()
        assert astrip.timing().fill_kernel == 6
        got_strip = (b.offsets.clone(), b.scores.clone(), b.cigars.clone(), b.cigar_len.clone(), b.status.clone())
        b.run(a16, overhang_strategy=strategy)
        torch.cuda.synchronize()
        assert a16.timing().fill_kernel == 5
        got = (b.offsets.clone(), b.scores.clone(), b.cigars.clone(), b.cigar_len.clone(), b.status.clone())
        assert all(torch.equal(x, y) for x, y in zip(got, got_strip))
        b.run(a32, overhang_strategy=strategy)
        torch.cuda.synchronize()
        assert a32.timing().fill_kernel == 3
        assert all(torch.equal(x, y) for x, y in zip(got, (b.offsets, b.scores, b.cigars, b.cigar_len, b.status)))
        assert int((b.status != 0).sum()) == 0
        if strategy == ol.SOFTCLIP:
            idx = [0, 57, 159]
            cg = b.cigar_strings(idx)
            for k, c in zip(idx, cg):
                o = ol.oracle_align(pairs[k][0], pairs[k][1], (200, -150, 260, 11), strategy)
                assert (int(b.offsets[k]), c, tuple(int(x) for x in b.scores[k])) == (o["offset"], o["cigar"], o["score"])
    a16.close()
    a32.close()
    astrip.close()


def test_cooperative_16bit_kernel_adversarial_windows():
    """Sequences chosen to stretch the 16-bit window of sw_dp_coop16_kernel: identical sequences (the steepest rise along the
    diagonal against the flattest fall beside it), nothing in common, a long insertion / deletion in the middle, tandem
    repeats, one base repeated -- under two parameter sets.  Whatever the kernel decides per pair (16 bits or its int32
    body), the results are the oracle's."""
    from mgl_amd import synth

    rng = synth.rng_for(99)
    g = synth.random_genome(rng, 3000).tobytes()
    h = synth.random_genome(rng, 2600).tobytes()
    rep = (b"ACGTTGCA" * 400)[:2900]
    pairs = [(g, g), (g, h), (g[:1500] + g[2100:], g), (g, g[:1200] + g[1900:]), (rep, rep[3:2500]), (b"A" * 2000, b"A" * 1700),
             (b"A" * 1500, b"C" * 1500), (g[:700], g), (g, g[1000:1400])]
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_cooperative(7)
    for params in ((200, -150, 260, 11), (100, -100, 300, 10), (50, -200, 400, 1)):
        for strategy in (ol.SOFTCLIP, ol.INDEL):
            res = forced.align_batch([p[0] for p in pairs], [p[1] for p in pairs], params, strategy, cigar_stride=8192)
            assert forced.timing().fill_kernel == 5
            for k, (t, q) in enumerate(pairs):
                o = ol.oracle_align(t, q, params, strategy)
                assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (o["offset"], o["cigar"], o["score"]), (params, strategy, k)
    forced.close()


def test_scratch_carry_equals_lds():
    """The long-query path (carry ring + query copies in HBM, agent-scope accesses) forced onto ordinary
    batches must reproduce the goldens exactly."""
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_carry_memory(1)
    forced.set_precision(32)
    rows = golden_io.load("known") + golden_io.load("shapes") + golden_io.load("random")[:600] + golden_io.load("ties")[::7]
    assert run_groups(forced, rows) == len(rows)
    # mixed geometry inside one batch + backtrack matrices
    gs = [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP][:24]
    forced.align_batch([g.t for g in gs], [g.q for g in gs], gs[0].params, ol.SOFTCLIP)
    for k, g in enumerate(gs):
        btr = forced.expand_slot(k, len(g.t), len(g.q))
        assert zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF == g.crc
    forced.close()


def test_backtrack_matrix_bit_exact():
    """Logical backtrack matrix (the reference's int32 run lengths) cell for cell."""
    rows = (golden_io.load("known") + golden_io.load("ties")[::16] + golden_io.load("shapes")[::11]
            + golden_io.load("random")[::25])
    for g in rows:
        btr, ez = sw.backtrack_matrix(g.t, g.q, g.params, g.strategy)
        crc = zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF
        assert crc == g.crc, (g.suite, g.t, g.q, g.params, g.strategy)
        assert tuple(ez) == g.score
        assert not btr[0].any() and not btr[:, 0].any()


def test_single_pair_entry():
    g = golden_io.load("known")[0]
    cigar, off, ez = sw.align(g.t, g.q, g.params, g.strategy)
    assert (cigar, off, tuple(ez)) == (g.cigar, g.offset, g.score)
    # sign normalisation (..._MicrosoftSmithWaterman.cpp:51-55): any sign convention, same answer
    m, x, o, e = g.params
    assert sw.align(g.t, g.q, (-m, -x, -o, -e), g.strategy)[:2] == (g.cigar, g.offset)


def test_operator_interface(aligner):
    # MicrosoftSmithWaterman.align(ref, alt, params, strategy) -> (cigar, offset)
    r = aligner.align(b"ACGTACGTACGTTTGACCA", b"CGTACGTTGACC", sw.GATK_PARAMETERS, sw.SWOverhangStrategy.SOFTCLIP)
    assert r == sw.SWNativeAlignerResult("6M1D6M", 5)
    r = aligner.align(b"ACGT", b"TTTTACGTACGTGG", sw.GATK_PARAMETERS, sw.SWOverhangStrategy.IGNORE)
    assert r == sw.SWNativeAlignerResult("14M", -4)


def test_fuzz_vs_oracle(aligner):
    """Fresh seeded inputs with ragged lengths inside one batch (so waves mix geometries)."""
    rng = np.random.default_rng(20260101)
    psets = [(200, -150, 260, 11), (1, -1, 1, 1), (5, -4, 10, 1), (3, -1, 4, 3), (25, -50, 110, 6)]
    for it in range(10):
        n = 257
        alpha = np.frombuffer(b"ACGT" if it % 2 else b"AC", np.uint8)
        ts, qs = [], []
        for k in range(n):
            tl = int(rng.integers(1, 300)) if it < 8 else int(rng.integers(1, 40))
            ql = int(rng.integers(1, 200)) if it < 8 else int(rng.integers(1, 40))
            t = alpha[rng.integers(0, len(alpha), tl)].tobytes()
            if k % 3 == 0 and tl > 2:
                a = int(rng.integers(0, tl - 1))
                q = t[a:a + ql] or t[:1]
            else:
                q = alpha[rng.integers(0, len(alpha), ql)].tobytes()
            ts.append(t)
            qs.append(q)
        params, strategy = psets[it % len(psets)], ol.STRATEGIES[it % 4]
        res = aligner.align_batch(ts, qs, params, strategy)
        off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
        assert (res.offsets == off).all()
        assert (res.scores == sc).all()
        assert res.cigars == cg


def test_chunked_equals_unchunked(aligner):
    """A tiny workspace forces many chunks; results must not depend on chunking."""
    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP][:200]
    ts, qs = [g.t for g in rows], [g.q for g in rows]
    a = aligner.align_batch(ts, qs, rows[0].params, ol.SOFTCLIP)
    small = sw.MicrosoftSmithWaterman(0)
    small.set_workspace(1 << 20)
    b = small.align_batch(ts, qs, rows[0].params, ol.SOFTCLIP)
    small.close()
    assert (a.offsets == b.offsets).all() and (a.scores == b.scores).all() and a.cigars == b.cigars
    for k, g in enumerate(rows):
        assert a.cigars[k] == g.cigar and int(a.offsets[k]) == g.offset


def test_cigar_overflow_is_reported(aligner):
    from mgl_amd import _lib
    g = golden_io.load("known")[1]  # 1D6M5D6M1D
    with pytest.raises(_lib.MglSwError) as e:
        aligner.align_batch([g.t], [g.q], g.params, g.strategy, cigar_stride=4)
    assert e.value.status == _lib.ERR_CIGAR_OVERFLOW


def test_bad_arguments(aligner):
    from mgl_amd import _lib
    with pytest.raises(_lib.MglSwError) as e:
        aligner.align_batch([b""], [b"ACGT"])
    assert e.value.status == _lib.ERR_BAD_ARG
    with pytest.raises(_lib.MglSwError) as e:
        aligner.align_batch([b"ACGT"], [b"ACGT"], overhang_strategy=3)
    assert e.value.status == _lib.ERR_BAD_ARG


def test_cpp_dropin_api():
    """A C++ caller written against the reference's names (align_avx, align_scalar, calculateMatrix,
    calculateCigar) built on include/mgl_sw.hpp reproduces the golden answers, including through the
    two-step calculateMatrix -> calculateCigar form of sw.cpp:258-272."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp")])
    rows = [g for g in golden_io.load("known") + golden_io.load("shapes")[::9] + golden_io.load("random")[::40]
            if b" " not in g.t]
    inp = "".join(f"{g.t.decode()} {g.q.decode()} {g.params[0]} {g.params[1]} {g.params[2]} {g.params[3]} {g.strategy}\n"
                  for g in rows)
    out = subprocess.run([os.path.join(root, "tests", "cpp", "dropin_caller")], input=inp.encode(), capture_output=True,
                         check=True).stdout.decode().splitlines()
    assert len(out) == len(rows)
    for g, line in zip(rows, out):
        f = line.split()
        assert (int(f[0]), f[1]) == (g.offset, g.cigar), (g, line)
        assert (int(f[2]), f[3]) == (g.offset, g.cigar), (g, line)
        assert tuple(int(x) for x in f[4:10]) == g.score
        assert int(f[10]) == g.crc


def test_band_stage_matches_the_reference_band_by_band():
    """calculateMatrix_avx (sw_avx.h:7) as a stage: mgl_sw_band_fill on the caller's arrays against the reference's own
    calculateMatrix_avx (oracle/_ref), compared after EVERY band -- score[], step[], the in-matrix entries of gap[], the
    band's backtrack cells, mqe / mqe_t -- then the finished band-layout matrix against the reference's logical one."""
    import ctypes as C

    from mgl_amd import _lib

    if not ol.have_ref() or not hasattr(ol.ref(), "ref_band_fill"):
        pytest.skip("oracle/_ref (the compiled reference) is not available")
    R, L = ol.ref(), _lib.lib()
    i32p = C.POINTER(C.c_int32)
    # (queries of eight bases and more: below that the reference dispatches to its scalar path and its AVX2 stage is not valid)
    rows = [g for g in golden_io.load("known")] + golden_io.load("random")[:40] + golden_io.load("ties")[:24] + \
           [g for g in golden_io.load("shapes") if len(g.q) >= 1][:80]
    rows = [g for g in rows if len(g.q) >= 8]
    checked = 0
    for g in rows:
        tl, ql, bw = len(g.t), len(g.q), 8
        m, x, o, e = g.params
        ncol, pad_t = ql + 1, (bw - tl % bw) % bw
        rq = np.zeros(ql + 2 * bw, np.int32)
        rq[bw + ql - 1 - np.arange(ql)] = np.frombuffer(g.q, np.uint8)
        et = np.zeros(tl + pad_t, np.int32)
        et[:tl] = np.frombuffer(g.t, np.uint8)
        state = []
        for _ in range(2):  # [reference, ours]
            score, step, gap = np.zeros(ncol + bw, np.int32), np.zeros(ncol + bw, np.int32), np.ones(ql + 2 * bw, np.int32)
            step[:ncol] = -abs(o)
            if g.strategy in (ol.INDEL, ol.LEAD_INDEL):
                k = np.arange(1, ncol)
                score[1:ncol] = -abs(o) - (k - 1) * abs(e)
                step[1:ncol] += -abs(o) - (k - 1) * abs(e)
            state.append(dict(score=score, step=step, gap=gap, bt=np.zeros((ql + bw - 1) * (tl + pad_t), np.int32),
                              mqe=np.array([-0x40000000, -1], np.int32)))
        ref, mine = state
        ez = _lib.Score(-0x40000000, -1, -0x40000000, -1, -1, 0)
        P = lambda a: a.ctypes.data_as(i32p)
        rows_left, band = tl, 0
        while rows_left > 0:
            nrow = min(bw, rows_left)
            rows_left -= nrow
            R.ref_band_fill(P(et), tl, P(rq), ql, P(ref["bt"]), band, bw, nrow, P(ref["score"]), P(ref["step"]), P(ref["gap"]), m, x, o, e,
                            g.strategy, P(ref["mqe"]))
            rc = L.mgl_sw_band_fill(et.ctypes.data, tl, rq.ctypes.data, ql, mine["bt"].ctypes.data, band, bw, nrow, mine["score"].ctypes.data,
                                    mine["step"].ctypes.data, mine["gap"].ctypes.data, m, x, o, e, g.strategy, C.byref(ez))
            assert rc == 0
            ctx = (g.t, g.q, g.params, g.strategy, band)
            assert (mine["score"][:ncol] == ref["score"][:ncol]).all(), ctx
            assert (mine["step"][:ncol] == ref["step"][:ncol]).all(), ctx
            if nrow == bw:  # (after the last, partial band the reference's gap[] holds the run lengths of its PADDING rows -- lane 7
                assert (mine["gap"][bw:bw + ql] == ref["gap"][bw:bw + ql]).all(), ctx  # stores last -- and nothing reads it any more)
            assert (ez.mqe, ez.mqe_t) == (int(ref["mqe"][0]), int(ref["mqe"][1])), ctx
            for J in range(nrow):  # the band's in-matrix cells: row J of the band, column j -> diagonal j - 1 + J
                idx = (ql + bw - 1) * bw * band + (np.arange(ql) + J) * bw + J
                assert (mine["bt"][idx] == ref["bt"][idx]).all(), ctx + (J,)
            band += 1
        # the finished band-layout matrix == the reference's logical matrix (sw_avx.h:33-40 indexing)
        o_full = ol.ref_full(g.t, g.q, g.params, g.strategy, want_btr=True)
        ii, jj = np.meshgrid(np.arange(tl), np.arange(ql), indexing="ij")
        pos = (ii // bw) * bw * (ql + bw - 1) + (jj + ii % bw) * bw + ii % bw
        assert (mine["bt"][pos] == o_full["btr"][1:, 1:]).all(), (g.t, g.q, g.strategy)
        assert (ez.mqe, ez.mqe_t) == (g.score[0], g.score[1])
        checked += 1
    assert checked == len(rows) and checked > 80
    # argument checks: band outside the target, more rows than the band width
    z = np.zeros(64, np.int32)
    assert L.mgl_sw_band_fill(z.ctypes.data, 4, z.ctypes.data, 4, z.ctypes.data, 1, 8, 4, z.ctypes.data, z.ctypes.data, z.ctypes.data,
                              1, -1, 1, 1, 1, C.byref(ez)) == _lib.ERR_BAD_ARG
    assert L.mgl_sw_band_fill(z.ctypes.data, 4, z.ctypes.data, 4, z.ctypes.data, 0, 8, 9, z.ctypes.data, z.ctypes.data, z.ctypes.data,
                              1, -1, 1, 1, 1, C.byref(ez)) == _lib.ERR_BAD_ARG


def test_grouped_geometry_variable_length_reads(aligner):
    """Reads of varying length (trimmed reads) against windows of two sizes, addressed by (start, length), sorted by
    geometry and padded to blocks of eight (MGL_SW_FLAG_GROUPED_GEOMETRY): the packed-int16 kernel runs one geometry
    per wave; results equal the oracle's and the int32 kernel's on the same pairs."""
    import torch
    from mgl_amd import device_batch, synth

    rng = synth.rng_for(77)
    n = 3000
    genome = synth.random_genome(rng, 1 << 16)
    tl = rng.choice([200, 256], size=n)
    ql = rng.integers(118, 151, size=n)
    ql[::97] = 9          # a few odd ones: too short for the chained schedule / a bucket of its own
    ql[5], tl[5] = 77, 231   # geometries that occur once or twice: they take the mixed (int32) part
    ql[6], tl[6] = 78, 231
    ql[7], tl[7] = 78, 231
    ts = rng.integers(0, len(genome) - 300, size=n)
    reads = synth.illumina_reads(rng, genome, ts + rng.integers(0, 40, size=n), 150)
    q_start = np.arange(n, dtype=np.int64) * 150
    dev = torch.device("cuda", 0)
    gb = device_batch.GroupedBatch(torch.from_numpy(genome).to(dev), torch.from_numpy(ts).to(dev), torch.from_numpy(tl).to(dev),
                                   torch.from_numpy(reads.reshape(-1)).to(dev), torch.from_numpy(q_start).to(dev),
                                   torch.from_numpy(ql).to(dev), cigar_stride=128)
    assert gb.n_grouped % 8 == 0 and gb.n >= n and gb.n_rest == 3
    # the promise itself: every aligned block of eight slots of the grouped part has one geometry
    gt, gq = gb.t_len[:gb.n_grouped].view(-1, 8), gb.q_len[:gb.n_grouped].view(-1, 8)
    assert (gt == gt[:, :1]).all() and (gq == gq[:, :1]).all()
    gb.run(aligner)
    torch.cuda.synchronize()
    assert int((gb.status != 0).sum()) == 0
    assert torch.equal(gb.gather()[0], gb.offsets[gb.first_slot])
    slot = gb.first_slot.cpu().numpy()
    assert (gb.order.cpu().numpy()[slot] == np.arange(n)).all()
    off, sc = gb.offsets.cpu().numpy()[slot], gb.scores.cpu().numpy()[slot]
    cg = gb.cigar_strings(slot.tolist())
    tseqs = [genome[ts[k]:ts[k] + tl[k]].tobytes() for k in range(n)]
    qseqs = [reads[k, :ql[k]].tobytes() for k in range(n)]
    woff, wsc, wcg = ol.oracle_align_batch(tseqs, qseqs, (200, -150, 260, 11), ol.SOFTCLIP, nthreads=8)
    assert (off == woff).all() and (sc == wsc).all() and cg == wcg
    # the host-buffer entry notices a batch that is sorted this way by itself
    o_np = gb.order.cpu().numpy()
    o_np = o_np[:gb.n_grouped]
    res = aligner.align_batch([tseqs[i] for i in o_np], [qseqs[i] for i in o_np], (200, -150, 260, 11), ol.SOFTCLIP, cigar_stride=128)
    assert aligner.timing().packed16 == 1
    assert (res.offsets == woff[o_np]).all() and (res.scores == wsc[o_np]).all() and list(res.cigars) == [wcg[i] for i in o_np]
    # duplicates (padding slots) carry the same answers as the pair they repeat
    o = gb.order.cpu().numpy()
    assert (gb.offsets.cpu().numpy() == off[o]).all() and (gb.scores.cpu().numpy() == sc[o]).all()


def test_grouped_geometry_traceback_regions_do_not_overlap(aligner):
    """Grouped batches run one geometry per wave, and the packed kernel's step count is not monotone in tl or ql (a
    partial last stripe runs stand-alone; queries below 28 bases are not chained): a group just under the batch
    maximum must not write past its traceback region into its neighbour's.  Blocks of eight at the maxima sit between
    blocks that need MORE steps than the maximum geometry; reads come from the first rows of their windows, so a
    clobbered first block of a region would show in the CIGAR / offset."""
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    for geoms in ([(256, 150), (255, 150), (241, 148), (256, 150), (250, 149), (256, 150), (243, 150)],
                  [(256, 28), (256, 25), (256, 28), (250, 9), (256, 27), (256, 28), (255, 26)],
                  [(32, 150), (31, 150), (17, 149), (32, 150)]):
        ts, qs = [], []
        for tl, ql in geoms * 3:
            for _ in range(8):
                t = alpha[rng.integers(0, 4, tl)]
                a = int(rng.integers(0, min(6, max(1, tl - ql + 1))))
                q = np.resize(t[a:a + ql], ql).copy()
                if ql > 12:
                    q[rng.integers(0, ql)] = alpha[rng.integers(0, 4)]
                    cut = int(rng.integers(4, ql - 4))
                    q = np.concatenate([q[:cut], q[cut + 2:], alpha[rng.integers(0, 4, 2)]])  # a short deletion
                ts.append(t.tobytes())
                qs.append(q.tobytes())
        for strategy in (ol.SOFTCLIP, ol.INDEL):
            res = aligner.align_batch(ts, qs, (200, -150, 260, 11), strategy, cigar_stride=256)
            assert aligner.timing().packed16 == 1
            off, sc, cg = ol.oracle_align_batch(ts, qs, (200, -150, 260, 11), strategy, nthreads=4)
            assert (res.offsets == off).all() and (res.scores == sc).all() and list(res.cigars) == cg


def test_host_entry_sorts_mixed_batches_by_geometry_itself():
    """The reference takes any pair (sw_avx.cpp:6-108): a host batch of variable-length reads, unsorted and unflagged, must
    reach the packed kernel without the caller's help.  mgl_sw_align_batch sorts every chunk by (tl, ql) on the host, runs
    full blocks of eight through the packed kernel and the left-over pairs through the int32 kernel, and hands the
    results back in the caller's order -- identical to the oracle and to the int32-only run; several chunks; real reads."""
    from mgl_amd import synth

    rng = synth.rng_for(5)
    n = 30000
    genome = synth.random_genome(rng, 1 << 16)
    tl = rng.choice([200, 256], size=n)
    ql = rng.integers(100, 151, size=n)
    ql[::211] = rng.integers(1, 30, size=len(ql[::211]))          # a sprinkling of odd geometries (left-over pairs)
    ts = rng.integers(0, len(genome) - 300, size=n)
    reads = synth.illumina_reads(rng, genome, ts + rng.integers(0, 40, size=n), 150)
    tseqs = [genome[ts[k]:ts[k] + tl[k]].tobytes() for k in range(n)]
    qseqs = [reads[k, :ql[k]].tobytes() for k in range(n)]
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(200 << 20)                                      # ~9 000 pairs per chunk: four chunks, both halves reused
    for strategy in (ol.SOFTCLIP, ol.LEAD_INDEL):
        res = a.align_batch(tseqs, qseqs, (200, -150, 260, 11), strategy, cigar_stride=128)
        tm = a.timing()
        assert tm.packed16 == 1 and tm.fill_kernel == 1 and tm.dp_launches >= 3, "a mixed host batch should be sorted onto the packed kernel"
        woff, wsc, wcg = ol.oracle_align_batch(tseqs, qseqs, (200, -150, 260, 11), strategy, nthreads=8)
        assert (res.offsets == woff).all() and (res.scores == wsc).all() and list(res.cigars) == wcg
    a.set_precision(32)                                             # no packed kernel: nothing to sort for
    res32 = a.align_batch(tseqs, qseqs, (200, -150, 260, 11), ol.LEAD_INDEL, cigar_stride=128)
    assert a.timing().packed16 == 0
    assert (res32.offsets == res.offsets).all() and (res32.scores == res.scores).all() and list(res32.cigars) == list(res.cigars)
    a.close()
    # real Illumina reads against their exact reference spans (variable tl): the golden records of the reference's BAM resource
    rows = [g for g in golden_io.load("bam") if g.suite != "bamwin" and g.strategy == ol.SOFTCLIP]
    assert len(rows) > 1024 and len({(len(g.t), len(g.q)) for g in rows}) > 10
    b = sw.MicrosoftSmithWaterman(0)
    b.set_small_kernel(1)                                           # (3 304 / 2 = 1 652 pairs: a default context would run one wave per pair)
    res = b.align_batch([g.t for g in rows], [g.q for g in rows], rows[0].params, ol.SOFTCLIP)
    assert b.timing().packed16 == 1
    for k, g in enumerate(rows):
        assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (g.offset, g.cigar, g.score)
    b.close()


@pytest.mark.parametrize("resident", [True, False])
def test_sorted_chunks_run_big_geometries_on_the_lane_kernel(monkeypatch, resident):
    """A chunk sorted by geometry (either entry) hands the geometries that fill whole waves of 128 pairs to the checkpointed lane
    kernel -- one geometry per wave -- the remaining full blocks of eight to the packed kernel and the left-over pairs to the int32
    kernel.  The threshold (half a round of the chip) is lowered so that a test-sized batch takes all three."""
    import torch
    from mgl_amd import device_batch, synth

    monkeypatch.setenv("MGL_SW_DEBUG_LANE_GROUP_MIN", "128")
    rng = synth.rng_for(91)
    n = 36_000
    genome = synth.random_genome(rng, 1 << 18)
    starts = rng.integers(0, len(genome) - 300, size=n)
    tls = rng.choice([256, 256, 256, 200], size=n)
    qls = rng.choice([150, 150, 125, 101, 64], size=n)
    qls[::53] = rng.integers(1, 151, size=len(qls[::53]))   # odd geometries: blocks of eight and left-over pairs
    reads = synth.illumina_reads(rng, genome, starts + 30, read_len=150, sub=0.02, ins=0.004, dele=0.004)
    tseqs = [genome[s: s + tl].tobytes() for s, tl in zip(starts, tls)]
    qseqs = [r[:q].tobytes() for r, q in zip(reads, qls)]
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(256 << 20)  # a few chunks
    for strategy in (ol.SOFTCLIP, ol.INDEL):
        woff, wsc, wcg = ol.oracle_align_batch(tseqs, qseqs, (200, -150, 260, 11), strategy, nthreads=8)
        if resident:
            td, toff = sw.concat(tseqs)
            qd, qoff = sw.concat(qseqs)
            b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=128)
            b.run(a, overhang_strategy=strategy)
            torch.cuda.synchronize()
            off, sc, cg = b.offsets.cpu().numpy(), b.scores.cpu().numpy(), b.cigar_strings()
            assert int((b.status != 0).sum()) == 0
        else:
            res = a.align_batch(tseqs, qseqs, (200, -150, 260, 11), strategy, cigar_stride=128)
            off, sc, cg = res.offsets, res.scores, list(res.cigars)
        tm = a.timing()
        assert tm.dp_launches >= 2 and tm.fill_kernel == 7, "the sorted chunks' big geometries should take the lane kernel"
        assert (off == woff).all() and (sc == wsc).all()
        assert all(cg[k] == wcg[k] for k in range(n))
    a.close()


def test_a_sorted_device_batch_is_one_chunk_sized_by_wave_slots(monkeypatch):
    """Round 4: a device-resident batch of mixed geometries, large enough for lane launches, is ONE chunk in an 8 GiB workspace -- the
    lane kernel's persistent grid keeps a region per wave slot, the left-over pairs (here: many, a third of the batch has odd
    geometries) go through the packed and int32 kernels in pieces of what their area holds.  Identical to the per-pair sizing of
    round 3 (itself checked against the reference's path above), with the area made small enough for several pieces, and to the
    CPU restatement on a sample."""
    import torch
    from mgl_amd import device_batch, synth

    rng = synth.rng_for(404)
    n = 260_000   # (two thirds in whole waves of one geometry: above the 131 072 from which a chunk's bulk gets its own lane launch)
    genome = synth.random_genome(rng, 1 << 19)
    starts = rng.integers(0, len(genome) - 300, size=n)
    tls = rng.choice([256, 256, 256, 200], size=n)
    qls = rng.choice([150, 150, 125, 101], size=n)
    odd = rng.random(n) < 0.33
    qls[odd] = rng.integers(1, 151, size=int(odd.sum()))
    tls[odd] = rng.integers(150, 257, size=int(odd.sum()))
    reads = synth.illumina_reads(rng, genome, starts + 30, read_len=150, sub=0.02, ins=0.004, dele=0.004)
    tseqs = [genome[s: s + tl].tobytes() for s, tl in zip(starts, tls)]
    qseqs = [r[:q].tobytes() for r, q in zip(reads, qls)]
    td, toff = sw.concat(tseqs)
    qd, qoff = sw.concat(qseqs)
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(8 << 30)
    outs = []
    for env in ({}, {"MGL_SW_DEBUG_LEFT_AREA": str(8 << 20)}, {"MGL_SW_DEBUG_GROUP_REGIONS": "0"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=128)
        b.run(a, overhang_strategy=ol.SOFTCLIP)
        torch.cuda.synchronize()
        tm = a.timing()
        assert int((b.status != 0).sum()) == 0
        if "MGL_SW_DEBUG_GROUP_REGIONS" not in env:
            assert tm.dp_launches == 1 and tm.fill_kernel == 7, "one chunk, its bulk on the lane kernel"
        else:
            assert tm.dp_launches > 1, "sized per pair: several chunks in 8 GiB (the last one too small for a lane launch)"
        outs.append((b.offsets.cpu().numpy(), b.scores.cpu().numpy(), b.cigars.cpu().numpy(), b.cigar_len.cpu().numpy()))
        for k_ in env:
            monkeypatch.delenv(k_)
    for o in outs[1:]:
        assert all((x == y).all() for x, y in zip(outs[0], o))
    sample = rng.choice(n, size=4000, replace=False)
    woff, wsc, wcg = ol.oracle_align_batch([tseqs[k] for k in sample], [qseqs[k] for k in sample], (200, -150, 260, 11), ol.SOFTCLIP, nthreads=8)
    cg = b.cigar_strings()
    assert (outs[0][0][sample] == woff).all() and (outs[0][1][sample] == wsc).all() and all(cg[k] == wcg[j] for j, k in enumerate(sample))
    a.close()


def test_grouping_helper_feeds_the_indexed_entry(aligner):
    """What a C caller does with variable-length reads: mgl_sw_group_by_geometry on the host, then the grouped part through
    mgl_sw_align_batch_device_indexed with MGL_SW_FLAG_GROUPED_GEOMETRY (packed kernel) and the rest without the flag."""
    import ctypes as C

    import torch

    from mgl_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(21)
    n = 3000
    alpha = np.frombuffer(b"ACGT", np.uint8)
    tl = np.full(n, 256, np.int32)
    ql = rng.integers(120, 151, n).astype(np.int32)
    ts = [alpha[rng.integers(0, 4, 256)] for _ in range(n)]
    qs = []
    for k in range(n):
        a = int(rng.integers(0, 256 - ql[k] + 1))
        q = ts[k][a:a + ql[k]].copy()
        q[rng.integers(0, ql[k])] = alpha[rng.integers(0, 4)]
        qs.append(q)
    order = np.zeros(n, np.int64)
    ng = C.c_int64()
    assert L.mgl_sw_group_by_geometry(n, tl.ctypes.data, ql.ctypes.data, order.ctypes.data, C.byref(ng)) == 0
    assert ng.value > n - 8 * 31
    dev = torch.device("cuda", 0)
    t_start = np.arange(n, dtype=np.int64) * 256
    q_start = np.concatenate([[0], np.cumsum(ql[:-1], dtype=np.int64)])
    d_t = torch.from_numpy(np.concatenate(ts)).to(dev)
    d_q = torch.from_numpy(np.concatenate(qs)).to(dev)
    perm = lambda a: torch.from_numpy(np.ascontiguousarray(a[order])).to(dev)
    d_ts, d_tl, d_qs, d_ql = perm(t_start), perm(tl), perm(q_start), perm(ql)
    off = torch.zeros(n, dtype=torch.int32, device=dev)
    sc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
    cg = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
    ln = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    for lo, cnt, flags in ((0, ng.value, _lib.FLAG_GROUPED_GEOMETRY), (ng.value, n - ng.value, 0)):
        if cnt == 0:
            continue
        rc = L.mgl_sw_align_batch_device_indexed(
            aligner.ctx, C.c_void_p(stream.cuda_stream), cnt, d_t.data_ptr(), d_ts[lo:].data_ptr(), d_tl[lo:].data_ptr(),
            d_q.data_ptr(), d_qs[lo:].data_ptr(), d_ql[lo:].data_ptr(), 256, 150, 200, -150, 260, 11, ol.SOFTCLIP,
            off[lo:].data_ptr(), sc[lo:].data_ptr(), cg[lo:].data_ptr(), 64, ln[lo:].data_ptr(), st[lo:].data_ptr(), flags)
        assert rc == 0
        if flags:
            assert aligner.timing().packed16 == 1
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    want_off, want_sc, want_cg = ol.oracle_align_batch([t.tobytes() for t in ts], [q.tobytes() for q in qs], (200, -150, 260, 11),
                                                      ol.SOFTCLIP, nthreads=4)
    got_off, got_sc, got_ln, got_cg = off.cpu().numpy(), sc.cpu().numpy(), ln.cpu().numpy(), cg.cpu().numpy()
    for slot in range(n):
        k = int(order[slot])
        assert got_off[slot] == want_off[k] and (got_sc[slot] == want_sc[k]).all()
        assert got_cg[slot, : got_ln[slot]].tobytes().decode() == want_cg[k]


def test_single_long_pair_through_the_one_pair_entry():
    """mgl_sw_align (the alignNative replacement, per-thread context) on the 10 kb x 10 kb golden pair: its 50 MB
    traceback must fit without the caller configuring anything, and one pair must not reserve sixteen pairs' worth."""
    g = [r for r in golden_io.load("long") if r.suite == "long"][0]
    cigar, off, ez = sw.align(g.t, g.q, g.params, g.strategy)
    assert off == g.offset and "sha1:" + hashlib.sha1(cigar.encode()).hexdigest() == g.cigar
    assert tuple(ez) == g.score
    # a small workspace holds exactly as many pairs as fit, down to one per chunk
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(24 << 20)   # two halves of 12 MB: one pair's kept rows and checkpoints each (10.1 MB as packed entries -- checkpoint columns every 128 from round 4 on; with every flag stored: 50 MB)
    res = a.align_batch([g.t] * 3, [g.q] * 3, g.params, g.strategy, cigar_stride=24000)
    assert all(int(res.offsets[k]) == g.offset and "sha1:" + hashlib.sha1(res.cigars[k].encode()).hexdigest() == g.cigar for k in range(3))
    assert a.timing().dp_launches == 3
    a.close()


def test_score_only_flag(aligner):
    """MGL_SW_FLAG_SCORE_ONLY: all six ScoreMax fields identical to the full call; on the packed kernel no traceback is
    formed (offsets 0, lengths 0); on other kernels the flag is a no-op."""
    import torch
    from mgl_amd import device_batch

    dev = torch.device("cuda", 0)
    b = device_batch.window_batch(5, 4096, dev)
    b.run(aligner)
    torch.cuda.synchronize()
    assert aligner.timing().packed16 == 1
    full_scores, full_off = b.scores.clone(), b.offsets.clone()
    b.cigar_len.fill_(-1)
    for strategy in ol.STRATEGIES:
        b.run(aligner, overhang_strategy=strategy)
        torch.cuda.synchronize()
        want = b.scores.clone()
        b.scores.zero_()
        b.run(aligner, overhang_strategy=strategy, score_only=True)
        torch.cuda.synchronize()
        assert torch.equal(b.scores, want) and int(b.offsets.abs().sum()) == 0 and int(b.cigar_len.abs().sum()) == 0
        assert aligner.timing().tb_ms == 0 or True
    # mixed batch (int32 kernel): the hint is ignored, everything is produced as usual
    rows = golden_io.load("random")[:64]
    by = [g for g in rows if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP]
    td, toff = sw.concat([g.t for g in by]); qd, qoff = sw.concat([g.q for g in by])
    mb = device_batch.from_host(td, toff, qd, qoff, dev, cigar_stride=1024)
    mb.run(aligner, (200, -150, 260, 11), ol.SOFTCLIP, score_only=True)
    torch.cuda.synchronize()
    cg = mb.cigar_strings()
    for k, g in enumerate(by):
        assert (int(mb.offsets[k]), cg[k], tuple(int(x) for x in mb.scores[k])) == (g.offset, g.cigar, g.score)


def test_native_threads_through_the_coalescer():
    """GATK's calling pattern from native threads (tests/cpp/coalesce_bench.cpp): 48 threads, one pair per mgl_sw_align call, served by
    their mailboxes' resident waves or merged into device batches by the dispatcher; every answer equals the direct call's."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp")])
    import re

    # the default front-end (mailboxes, requests over the BAR where there is one); mailboxes whose requests stay in pinned host memory;
    # a grid that gives up after 30 us of silence and lives 1 ms (calls keep finding their wave gone: the launch races of
    # sw_service.cpp on the real device); eight mailboxes for 48 threads (the rest through the coalescer); the coalescer alone
    for env, want_mailboxes in (({}, True), ({"MGL_SW_SERVICE_BAR": "0"}, True), ({"MGL_SW_SERVICE_IDLE_US": "30", "MGL_SW_SERVICE_LIFE_MS": "1"}, True),
                                ({"MGL_SW_SERVICE_SLOTS": "8"}, True), ({"MGL_SW_SERVICE_SLOTS": "0"}, False)):
        for args in (["48", "60", "50"], ["7", "40", "0", "97", "33"]):
            r = subprocess.run([os.path.join(root, "tests", "cpp", "coalesce_bench")] + args, capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
            assert r.returncode == 0 and "wrong results 0" in r.stdout, (env, r.stdout, r.stderr)
            served = int(re.search(r"(\d+) through mailboxes", r.stdout).group(1))
            assert (served > 0) == want_mailboxes, (env, r.stdout)


def test_batch_mode_backtrack_matrix(aligner):
    """Traceback bits of pairs aligned inside a mixed-geometry batch (wave mates of different
    tl/ql) expand to the reference's matrix as well."""
    rows = golden_io.load("random")[:96:1]
    by = defaultdict(list)
    for g in rows:
        by[(g.params, g.strategy)].append(g)
    checked = 0
    for (params, strategy), gs in by.items():
        aligner.align_batch([g.t for g in gs], [g.q for g in gs], params, strategy)
        for k, g in enumerate(gs):
            btr = aligner.expand_slot(k, len(g.t), len(g.q))
            crc = zlib.crc32(np.ascontiguousarray(btr[1:, 1:]).astype("<i4").tobytes()) & 0xFFFFFFFF
            assert crc == g.crc, (k, g.params, g.strategy)
            checked += 1
    assert checked == len(rows)


# ---------------------------------------------------------------------------------------------
# packed-int16 fill kernel (sw_dp16_kernel): taken for batches with one geometry whose score
# range fits 16 bits.  It must be indistinguishable from the int32 kernel and from the oracle.

def _uniform_batch(rng, n, tl, ql, alphabet=b"ACGT", related=True):
    alpha = np.frombuffer(alphabet, np.uint8)
    ts, qs = [], []
    for k in range(n):
        t = alpha[rng.integers(0, len(alpha), tl)]
        if related and k % 4 != 3:
            # read = noisy copy of a window of the target (substitutions + one indel), padded/cut to ql
            a = int(rng.integers(0, max(1, tl - ql + 1))) if tl >= ql else 0
            src = np.resize(t[a:], ql + 2).copy()
            sub = rng.random(ql + 2) < 0.05
            src[sub] = alpha[rng.integers(0, len(alpha), int(sub.sum()))]
            if k % 4 == 1:
                src = np.delete(src, int(rng.integers(0, ql)))
            elif k % 4 == 2:
                src = np.insert(src, int(rng.integers(0, ql)), alpha[0])
            q = src[:ql]
        else:
            q = alpha[rng.integers(0, len(alpha), ql)]
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    return ts, qs


@pytest.mark.parametrize("tl,ql", [(256, 150), (1000, 150), (16, 16), (17, 15), (1, 1), (5, 3), (33, 8), (64, 65),
                                   (100, 151), (300, 7),
                                   # chained-stripe schedule of sw_dp16_kernel: every gap P - ql in 1..4, whole and
                                   # partial last stripes, the shortest period (32), one / two stripes
                                   (256, 147), (256, 149), (256, 151), (256, 152), (250, 150), (32, 28), (48, 31),
                                   (16, 150), (15, 150), (31, 40), (160, 27)])
def test_packed16_uniform_batches(aligner, tl, ql):
    rng = np.random.default_rng(tl * 1000 + ql)
    n = 37  # odd: the last group has a lone pair
    ts, qs = _uniform_batch(rng, n, tl, ql, b"ACGT" if ql % 2 else b"AC")
    for params in [(200, -150, 260, 11), (25, -50, 110, 6), (3, -1, 4, 3), (1, -1, 1, 1), (5, -4, 10, 1)]:
        for strategy in ol.STRATEGIES:
            res = aligner.align_batch(ts, qs, params, strategy)
            assert aligner.timing().packed16 == 1, "uniform small-range batch should take the packed kernel"
            off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
            assert (res.offsets == off).all(), (params, strategy)
            assert (res.scores == sc).all(), (params, strategy)
            assert res.cigars == cg, (params, strategy)
            # logical backtrack matrix of a pair in each half of a lane
            for slot in (0, 1, n - 1):
                btr = aligner.expand_slot(slot, tl, ql)
                o = ol.oracle_align(ts[slot], qs[slot], params, strategy, want_btr=True)
                assert (btr[1:, 1:] == o["btr"][1:, 1:]).all(), (params, strategy, slot)


# ---------------------------------------------------------------------------------------------
# two pairs per LANE (sw_dp16_lane_kernel): large uniform batches; forced here onto small ones.  Strips of 16 / 32 rows,
# partial last strips, tl below one strip, every strategy, the traceback layout through the expansion to the
# reference's int32 matrix.

# Both forms: flags of every cell stored (sw_dp16_lane_kernel, 4) and checkpoints + recomputed blocks (sw_dp16_lane_ck_kernel, 7:
# the default wherever the strips have 32 rows and CIGARs are written).

@pytest.fixture(scope="module", params=[1, 0], ids=["stored", "checkpointed"])
def lane_aligner(request):
    a = sw.MicrosoftSmithWaterman(0)
    a.set_lane_kernel(2)
    a.set_lane_checkpoint(request.param)
    a.lane_kernels = (4,) if request.param == 1 else (4, 7)
    yield a
    a.close()


@pytest.mark.parametrize("tl,ql", [(256, 150), (1000, 150), (16, 16), (17, 15), (1, 1), (5, 3), (33, 8), (64, 65), (100, 151),
                                   (300, 7), (32, 149), (31, 150), (48, 2), (250, 150), (255, 1)])
def test_lane_kernel_uniform_batches(lane_aligner, tl, ql):
    from mgl_amd import _lib

    rng = np.random.default_rng(tl * 1000 + ql + 1)
    n = 131  # more than one wave (128 pairs), odd: the last lane has a lone pair
    ts, qs = _uniform_batch(rng, n, tl, ql, b"ACGT" if ql % 2 else b"AC")
    for params in [(200, -150, 260, 11), (25, -50, 110, 6), (1, -1, 1, 1), (5, -4, 10, 1)]:
        for strategy in ol.STRATEGIES:
            res = lane_aligner.align_batch(ts, qs, params, strategy)
            tm = lane_aligner.timing()
            assert tm.packed16 == 1 and tm.fill_kernel in lane_aligner.lane_kernels, "a uniform small-range batch should take the lane kernel when forced"
            if 7 in lane_aligner.lane_kernels and (tl, ql) in ((256, 150), (1000, 150), (64, 65), (32, 149), (250, 150)):
                assert tm.fill_kernel == 7, "32-row strips: the checkpointed form is the default"
            off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
            assert (res.offsets == off).all(), (params, strategy)
            assert (res.scores == sc).all(), (params, strategy)
            assert res.cigars == cg, (params, strategy)
            for slot in (0, 1, 77, n - 1):
                if tm.fill_kernel == 7:
                    with pytest.raises(RuntimeError):
                        lane_aligner.expand_slot(slot, tl, ql)  # no stored traceback to expand
                    break
                btr = lane_aligner.expand_slot(slot, tl, ql)
                o = ol.oracle_align(ts[slot], qs[slot], params, strategy, want_btr=True)
                assert (btr[1:, 1:] == o["btr"][1:, 1:]).all(), (params, strategy, slot)


@pytest.mark.parametrize("tl,ql", [(256, 150), (300, 200), (120, 260), (64, 33), (1000, 150), (513, 31), (60, 300)])
def test_lane_checkpointed_long_gaps(tl, ql):
    """sw_dp16_lane_ck_kernel: paths with long gaps -- runs that cross the 16-row bands and 32-column blocks its walk recomputes, and
    horizontal runs longer than one look-ahead round inside a block -- plus unrelated pairs (many short gaps), every strategy."""
    rng = np.random.default_rng(tl * 7 + ql)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    n = 150
    ts, qs = [], []
    for k in range(n):
        t = alpha[rng.integers(0, 4, tl)]
        if k % 5 == 4:
            q = alpha[rng.integers(0, 4, ql)]
        else:
            src = np.resize(t[int(rng.integers(0, max(1, tl // 3))):], ql + 200).copy()
            gap = int(rng.integers(17, 70))
            at = int(rng.integers(1, max(2, ql - 1)))
            if k % 5 in (0, 1):   # the read lacks `gap` bases of the target: a long vertical run
                src = np.concatenate([src[:at], src[at + gap:]])
            elif k % 5 == 2:      # the read has `gap` extra bases: a long horizontal run
                src = np.concatenate([src[:at], alpha[rng.integers(0, 4, gap)], src[at:]])
            sub = rng.random(len(src)) < 0.03
            src[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
            q = src[:ql]
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    a = sw.MicrosoftSmithWaterman(0)
    a.set_lane_kernel(2)
    ran_ck = 0
    try:
        for params in [(200, -150, 260, 11), (25, -50, 110, 6), (1, -1, 1, 1), (5, -4, 10, 1), (10, -30, 40, 1)]:
            for strategy in ol.STRATEGIES:
                res = a.align_batch(ts, qs, params, strategy)
                ran_ck += a.timing().fill_kernel == 7  # (scores of some of these parameter sets leave 16 bits at the larger geometries)
                off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
                assert (res.offsets == off).all(), (params, strategy)
                assert (res.scores == sc).all(), (params, strategy)
                assert res.cigars == cg, (params, strategy)
        assert ran_ck >= 8, "32-row strips: most of these runs should take the checkpointed lane kernel"
    finally:
        a.close()


def test_lane_kernels_alphabets_binary_cigar_and_overflow(lane_aligner):
    """Both lane kernels on what their other tests leave out: arbitrary byte alphabets and unusual scoring parameters (bytes compared
    raw, sw.cpp:55; zero gap costs, many ties), BAM-style binary CIGARs, and CIGAR slots too small for some pairs (per-pair overflow
    status and needed length, the other pairs unaffected) -- against the oracle and against the default aligner's bytes."""
    import torch
    from mgl_amd import device_batch as db

    rng = np.random.default_rng(2718)
    psets = [(200, -150, 260, 11), (1, 0, 0, 0), (5, -3, 0, 1), (5, -3, 7, 0), (2, -7, 1, 1), (127, -128, 255, 1), (9, -9, 9, 9), (1, -1, 50, 50)]
    ran = 0
    for it, params in enumerate(psets):
        tl, ql = [(256, 150), (64, 65), (96, 40), (130, 33)][it % 4]
        hi = 256 if it % 2 else 3
        ts, qs = [], []
        for k in range(131):
            t = rng.integers(0, hi, tl, dtype=np.uint8)
            q = np.resize(t[int(rng.integers(0, max(1, tl - ql + 1))):], ql).copy() if k % 3 else rng.integers(0, hi, ql, dtype=np.uint8)
            if k % 3 == 1:
                q[rng.integers(0, ql, 3)] ^= 1
            ts.append(t.tobytes())
            qs.append(q.tobytes())
        for strategy in ol.STRATEGIES:
            res = lane_aligner.align_batch(ts, qs, params, strategy)
            ran += lane_aligner.timing().fill_kernel in lane_aligner.lane_kernels  # (gap open below gap extend: outside the 16-bit guard)
            off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
            assert (res.offsets == off).all() and (res.scores == sc).all() and list(res.cigars) == cg, (params, strategy)
    assert ran >= 16, "most of these parameter sets fit the lane kernels"
    # binary CIGARs and small slots: reads with an indel each, so that the text needs more than eight bytes for some pairs only
    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP][:300]
    td, toff = sw.concat([g.t for g in rows])
    qd, qoff = sw.concat([g.q for g in rows])
    b = db.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=4 * 64)
    b.run(lane_aligner, rows[0].params, ol.SOFTCLIP, binary_cigar=True)
    torch.cuda.synchronize()
    assert lane_aligner.timing().fill_kernel in lane_aligner.lane_kernels and int((b.status != 0).sum()) == 0
    assert b.cigar_elements() == [g.cigar for g in rows]
    small = db.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=8)
    small.run(lane_aligner, rows[0].params, ol.SOFTCLIP)
    torch.cuda.synchronize()
    st, need = small.status.cpu().numpy(), small.cigar_len.cpu().numpy()
    cgs = small.cigar_strings()
    assert 0 < int((st != 0).sum()) < len(rows)
    for k, g in enumerate(rows):
        assert need[k] == len(g.cigar) and int(small.offsets[k]) == g.offset
        assert (st[k] == 2 and len(g.cigar) > 8) or (st[k] == 0 and cgs[k] == g.cigar), (k, st[k], g.cigar)


def test_lane_kernel_goldens_unaligned_chunked_score_only(lane_aligner):
    """The window goldens (one geometry) through the lane kernel: sequences at odd byte offsets (the kernel reads aligned
    dwords and shifts), several chunks, and the score-only hint."""
    import torch
    from mgl_amd import _lib, device_batch

    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP]
    assert len(rows) > 100 and len({(len(g.t), len(g.q)) for g in rows}) == 1
    rows = (rows * 12)[:1500]
    res = lane_aligner.align_batch([g.t for g in rows], [g.q for g in rows], rows[0].params, ol.SOFTCLIP)
    assert lane_aligner.timing().fill_kernel in lane_aligner.lane_kernels
    for k, g in enumerate(rows):
        assert (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) == (g.offset, g.cigar, g.score)
    # device-resident, with a pad byte in front of both arrays: every sequence starts at an odd address
    td, toff = sw.concat([g.t for g in rows])
    qd, qoff = sw.concat([g.q for g in rows])
    tl, ql = len(rows[0].t), len(rows[0].q)
    for pad in (1, 2, 3):
        b = device_batch.from_host(np.concatenate([np.zeros(pad, np.uint8), td]), toff + pad,
                                   np.concatenate([np.zeros(pad, np.uint8), qd]), qoff + pad, "cuda:0", cigar_stride=64)
        assert b.uniform
        small = sw.MicrosoftSmithWaterman(0)
        small.set_lane_kernel(2)
        small.set_workspace(7 << 20)     # room for TWO wave slots of the persistent grid: every wave takes tile after tile off the counter, one region each
        b.run(small)
        torch.cuda.synchronize()
        tm = small.timing()
        assert tm.fill_kernel == 7 and tm.dp_launches == 1
        assert _lib.explain(len(rows), tl, ql, rows[0].params, flags=_lib.FLAG_UNIFORM_GEOMETRY, ctx=small.ctx).resident_waves == 2
        cg = b.cigar_strings()
        for k, g in enumerate(rows):
            assert (int(b.offsets[k]), cg[k], tuple(int(x) for x in b.scores[k])) == (g.offset, g.cigar, g.score)
        full = b.scores.clone()
        b.scores.zero_()
        b.run(small, score_only=True)
        torch.cuda.synchronize()
        assert torch.equal(full, b.scores)
        small.close()


def test_packed16_equals_forced_int32(aligner):
    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP][:128]
    ts, qs = [g.t for g in rows], [g.q for g in rows]
    a = aligner.align_batch(ts, qs, rows[0].params, ol.SOFTCLIP)
    assert aligner.timing().packed16 == 1
    forced = sw.MicrosoftSmithWaterman(0)
    forced.set_precision(32)
    b = forced.align_batch(ts, qs, rows[0].params, ol.SOFTCLIP)
    assert forced.timing().packed16 == 0
    forced.close()
    assert (a.offsets == b.offsets).all() and (a.scores == b.scores).all() and a.cigars == b.cigars
    for k, g in enumerate(rows):
        assert (int(a.offsets[k]), a.cigars[k], tuple(int(x) for x in a.scores[k])) == (g.offset, g.cigar, g.score)


def test_packed16_range_guard(aligner):
    """Scores that do not fit 16 bits must fall back to the int32 kernel -- and still be exact."""
    rng = np.random.default_rng(99)
    ts, qs = _uniform_batch(rng, 9, 400, 300)  # match * ql = 60000: with the gap terms beyond a 16-bit span
    res = aligner.align_batch(ts, qs, (200, -150, 260, 11), ol.INDEL)
    assert aligner.timing().packed16 == 0
    off, sc, cg = ol.oracle_align_batch(ts, qs, (200, -150, 260, 11), ol.INDEL, nthreads=4)
    assert (res.offsets == off).all() and (res.scores == sc).all() and res.cigars == cg
    # just inside the guard: homopolymer pairs drive H to its extremes (all-match and all-mismatch)
    for t, q in ((b"A" * 300, b"A" * 150), (b"A" * 300, b"C" * 150), (b"AC" * 150, b"CA" * 75),
                 (b"A" * 300, b"A" * 289), (b"A" * 300, b"C" * 289), (b"A" * 289, b"A" * 300), (b"C" * 289, b"A" * 300),
                 (b"AC" * 144, b"CA" * 144), (b"A" * 1200, b"A" * 240), (b"A" * 1200, b"C" * 240)):
        for strategy in ol.STRATEGIES:
            r = aligner.align_batch([t] * 3, [q] * 3, (200, -150, 260, 11), strategy)
            assert aligner.timing().packed16 == 1
            o = ol.oracle_align(t, q, (200, -150, 260, 11), strategy)
            for k in range(3):
                assert (int(r.offsets[k]), r.cigars[k], tuple(int(x) for x in r.scores[k])) == (
                    o["offset"], o["cigar"], o["score"])


def test_packed16_guard_edge_fuzz():
    """Random parameter sets at the largest geometry dp16_range_ok admits, extreme sequences (scripts/range_fuzz.py)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("range_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "range_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, packed = mod.run(24, 11, lambda *x: None)
    assert bad == 0 and packed > 20


def test_odd_parameter_sets():
    """Gap open < gap extend, zero penalties, a mismatch of -30000: uniform, mixed and forced-cooperative batches
    (scripts/odd_params_check.py)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("odd_params_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "odd_params_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(lambda *x: None) == 0


@pytest.mark.parametrize("mailboxes", [0, 8, 64], ids=["coalescer", "8_mailboxes", "mailboxes"])
def test_coalescing_front_end(mailboxes):
    """Many threads calling the one-pair entry (the way GATK drives alignNative): merged into device batches by the coalescer, or
    -- with the mailbox service in front of it (sw_service.hip: one resident wave per calling thread, no launch per call) -- served
    by their own waves, threads beyond the mailboxes still through the coalescer.  Each gets exactly the answer of the direct call."""
    import ctypes as C
    import threading

    from mgl_amd import _lib

    L = _lib.lib()
    rows = (golden_io.load("window")[:256] + golden_io.load("random")[:256] + golden_io.load("known"))
    results = [None] * len(rows)

    def call(k):
        g = rows[k]
        cap = 12 * (len(g.t) + len(g.q) + 4)
        buf = C.create_string_buffer(cap)
        ln, off, ez = C.c_int(), C.c_int(), _lib.Score()
        rc = L.mgl_sw_align(g.t, len(g.t), g.q, len(g.q), *g.params, g.strategy, buf, cap, C.byref(ln), C.byref(off),
                            C.byref(ez))
        results[k] = (rc, off.value, buf.raw[: ln.value].decode(),
                      (ez.mqe, ez.mqe_t, ez.max, ez.max_t, ez.max_q, ez.seg_length))

    b0, p0, c0, l0 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.mgl_sw_coalescing_stats(C.byref(b0), C.byref(p0))
    L.mgl_sw_service_stats(C.byref(c0), C.byref(l0))
    assert L.mgl_sw_set_coalescing(256, 2000) == 0
    assert L.mgl_sw_set_service(mailboxes, 300) == 0    # (waves that give up after 300 us of silence: some calls find theirs gone)
    try:
        threads = [threading.Thread(target=lambda lo=lo: [call(k) for k in range(lo, len(rows), 32)]) for lo in range(32)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        # a caller whose buffer is too small gets the overflow status and the length it needs, others are unaffected
        g = golden_io.load("known")[1]
        small = C.create_string_buffer(4)
        ln, off = C.c_int(), C.c_int()
        rc = L.mgl_sw_align(g.t, len(g.t), g.q, len(g.q), *g.params, g.strategy, small, 4, C.byref(ln), C.byref(off), None)
        assert rc == _lib.ERR_CIGAR_OVERFLOW and ln.value == len(g.cigar)
    finally:
        assert L.mgl_sw_set_coalescing(0, 0) == 0
        assert L.mgl_sw_set_service(64, 1000) == 0
    b1, p1, c1, l1 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.mgl_sw_coalescing_stats(C.byref(b1), C.byref(p1))
    L.mgl_sw_service_stats(C.byref(c1), C.byref(l1))
    served = c1.value - c0.value
    assert p1.value - p0.value + served == len(rows) + 1
    if mailboxes == 0:
        assert served == 0
        # (how many share a batch depends on how fast Python's threads come back against a device round trip of a few dozen microseconds)
        assert b1.value - b0.value < len(rows) * 3 // 4, "calls were not merged into batches"
    else:
        # every pair of these suites fits a mailbox except the few whose matrix of scores is beyond a workgroup's LDS
        assert served >= (len(rows) // 8 if mailboxes == 8 else len(rows) * 3 // 4) and 1 <= l1.value - l0.value <= served
    for k, g in enumerate(rows):
        assert results[k] == (0, g.offset, g.cigar, g.score), (k, results[k], g)
    # and the direct path still works after switching it off
    assert sw.align(rows[0].t, rows[0].q, rows[0].params, rows[0].strategy)[:2] == (rows[0].cigar, rows[0].offset)


# ---------------------------------------------------------------------------------------------
# 2-bit packed inputs (mgl_sw_align_batch_device_2bit)

def _packed_from_rows(rows, dev):
    """Pack golden (ACGT-only) pairs into one target array and one query array at unaligned base offsets."""
    import torch
    from mgl_amd import device_batch as db

    tparts, qparts, ts, qs, tlens, qlens = [], [], [], [], [], []
    tpos = qpos = 0
    for k, g in enumerate(rows):
        lead_t, lead_q = (k * 7) % 5, (k * 3) % 4  # gaps so that starts are not multiples of 4
        tparts.append(b"A" * lead_t + g.t)
        qparts.append(b"C" * lead_q + g.q)
        ts.append(tpos + lead_t)
        qs.append(qpos + lead_q)
        tpos += lead_t + len(g.t)
        qpos += lead_q + len(g.q)
        tlens.append(len(g.t))
        qlens.append(len(g.q))
    T = torch.from_numpy(db.pack2bit(b"".join(tparts))).to(dev)
    Q = torch.from_numpy(db.pack2bit(b"".join(qparts))).to(dev)
    i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    return db.PackedBatch(T, i64(ts), i32(tlens), Q, i64(qs), i32(qlens), max(tlens), max(qlens),
                          cigar_stride=2 * max(max(tlens), max(qlens)) + 16)


def test_2bit_inputs_match_goldens(aligner):
    import torch

    dev = torch.device("cuda", 0)
    acgt = set(b"ACGT")
    rows = [g for g in golden_io.load("random") + golden_io.load("shapes") + golden_io.load("window")
            if set(g.t) <= acgt and set(g.q) <= acgt]
    groups = defaultdict(list)
    for g in rows:
        groups[(g.params, g.strategy)].append(g)
    n = 0
    for (params, strategy), gs in groups.items():
        b = _packed_from_rows(gs, dev)
        b.run(aligner, params, strategy)
        torch.cuda.synchronize()
        assert int((b.status != 0).sum()) == 0
        off, sc, cg = b.offsets.cpu().numpy(), b.scores.cpu().numpy(), b.cigar_strings()
        for k, g in enumerate(gs):
            assert (int(off[k]), cg[k], tuple(int(x) for x in sc[k])) == (g.offset, g.cigar, g.score), (params, strategy, k)
            n += 1
    assert n == len(rows) and n > 1500


def test_2bit_window_batch_equals_ascii(aligner):
    """Windows into ONE packed genome + packed reads (the SURVEY config-2 layout) give byte-identical outputs to
    the ASCII batch of the same bases; the uniform batch takes the packed-int16 kernel in both forms."""
    import torch
    from mgl_amd import device_batch as db

    dev = torch.device("cuda", 0)
    pb, ab = db.window_batch_2bit(77, 20000, dev, genome_len=1 << 20)
    ab.run(aligner)
    torch.cuda.synchronize()
    assert aligner.timing().packed16 == 1
    pb.run(aligner)
    torch.cuda.synchronize()
    assert aligner.timing().packed16 == 1
    assert torch.equal(pb.offsets, ab.offsets) and torch.equal(pb.scores, ab.scores)
    assert torch.equal(pb.cigars, ab.cigars) and torch.equal(pb.cigar_len, ab.cigar_len)


@pytest.mark.parametrize("tl,ql", [(256, 150), (1000, 150), (64, 65), (250, 150), (33, 8), (96, 131)])
def test_2bit_inputs_on_the_lane_kernel(tl, ql):
    """The 2-bit wire format on sw_dp16_lane_ck_kernel (forced onto a small batch): uniform pairs packed at unaligned base
    offsets -- target windows overlapping inside one packed array -- every strategy, against the oracle; and byte-identical
    to the ASCII batch of the same bases on the same kernel."""
    import torch
    from mgl_amd import device_batch as db

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(tl * 31 + ql)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    n = 259
    genome = alpha[rng.integers(0, 4, 3 * tl + 2 * n + 64)]
    t_start = rng.integers(0, len(genome) - tl, n)            # overlapping windows, any alignment
    ts, qs = [], []
    for k in range(n):
        t = genome[t_start[k]: t_start[k] + tl]
        q = np.resize(t[int(rng.integers(0, max(1, tl // 2))):], ql + 8).copy()
        if k % 3 == 1 and ql > 12:
            at = int(rng.integers(2, ql - 4))
            q = np.concatenate([q[:at], q[at + 3:]])
        elif k % 3 == 2 and ql > 12:
            at = int(rng.integers(2, ql - 4))
            q = np.concatenate([q[:at], alpha[rng.integers(0, 4, 2)], q[at:]])
        sub = rng.random(len(q)) < 0.03
        q[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
        ts.append(t.tobytes())
        qs.append(q[:ql].tobytes())
    lead = 3                                                   # reads packed back to back behind three bases of padding
    G = torch.from_numpy(db.pack2bit(genome.tobytes() + b"A" * ((-len(genome)) % 4))).to(dev)
    qcat = b"C" * lead + b"".join(qs)
    Q = torch.from_numpy(db.pack2bit(qcat + b"A" * ((-len(qcat)) % 4))).to(dev)
    i64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.int64, device=dev)
    pb = db.PackedBatch(G, i64(t_start), None, Q, i64(lead + np.arange(n) * ql), None, tl, ql, cigar_stride=2 * max(tl, ql) + 16)
    td, toff = sw.concat(ts)
    qd, qoff = sw.concat(qs)
    ab = db.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=pb.cigar_stride)
    ab.uniform = True
    a = sw.MicrosoftSmithWaterman(0)
    a.set_lane_kernel(2)
    try:
        for params in [(200, -150, 260, 11), (25, -50, 110, 6), (1, -1, 1, 1)]:
            for strategy in ol.STRATEGIES:
                pb.run(a, params, strategy)
                torch.cuda.synchronize()
                # (33 rows: the planner prefers 16-row strips there, which only the kernels that store every flag have -- sw_dp16_kernel for 2-bit)
                assert a.timing().fill_kernel == (1 if tl == 33 else 7), "a uniform 2-bit batch takes the checkpointed lane kernel"
                assert int((pb.status != 0).sum()) == 0
                off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
                assert (pb.offsets.cpu().numpy() == off).all() and (pb.scores.cpu().numpy() == sc).all(), (params, strategy)
                assert pb.cigar_strings() == cg, (params, strategy)
                ab.run(a, params, strategy)
                torch.cuda.synchronize()
                assert a.timing().fill_kernel in ((4, 7) if tl == 33 else (7,))
                assert torch.equal(pb.cigars, ab.cigars) and torch.equal(pb.offsets, ab.offsets) and torch.equal(pb.scores, ab.scores)
    finally:
        a.close()


def test_lane_kernel_bases_outside_acgt():
    """sw_dp16_lane_ck_kernel stages base codes when every TARGET byte of a wave is one of ACGT and raw bytes otherwise
    (sw.cpp:55 compares bytes): reads with N, lower case and arbitrary bytes against clean targets (codes, the query byte
    becomes "no target base"), targets with N (that wave falls back to bytes), N against N (equal bytes match)."""
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    tl, ql, n = 256, 150, 300                  # three waves: clean / N in some reads / N in some targets
    ts, qs = [], []
    for k in range(n):
        t = alpha[rng.integers(0, 4, tl)].copy()
        a0 = int(rng.integers(0, tl - ql))
        q = t[a0:a0 + ql].copy()
        if k >= 128 and k % 3 == 0:            # reads with N / lower case / junk
            pos = rng.integers(0, ql, 4)
            q[pos] = np.frombuffer(b"NnaX", np.uint8)[rng.integers(0, 4, 4)]
        if k >= 256 and k % 4 == 1:            # targets with N, some aligned with an N of the read
            pos = rng.integers(a0, a0 + ql, 3)
            t[pos] = ord("N")
            if k % 8 == 1:
                q[pos - a0] = ord("N")
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    a = sw.MicrosoftSmithWaterman(0)
    a.set_lane_kernel(2)
    try:
        for strategy in ol.STRATEGIES:
            res = a.align_batch(ts, qs, (200, -150, 260, 11), strategy)
            assert a.timing().fill_kernel == 7
            off, sc, cg = ol.oracle_align_batch(ts, qs, (200, -150, 260, 11), strategy, nthreads=4)
            assert (res.offsets == off).all() and (res.scores == sc).all() and list(res.cigars) == cg, strategy
    finally:
        a.close()


def test_host_entries_packed_and_registered(aligner):
    """mgl_sw_align_batch_2bit (packed bases in host memory: ascending starts -> the arrays travel chunk by chunk; windows in any
    order -> whole arrays first) and mgl_sw_register_host_buffer (page-locked caller arrays: results copied straight into them),
    a uniform batch large enough for several chunks and a ragged one, against the oracle and the ASCII host entry."""
    from mgl_amd import device_batch as db

    rng = np.random.default_rng(4242)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    genome = alpha[rng.integers(0, 4, 1 << 16)]
    n, tl, ql = 6000, 256, 150
    win = rng.integers(0, len(genome) - tl, n).astype(np.int64)       # windows in any order
    ts, qs = [], []
    for k in range(n):
        t = genome[win[k]: win[k] + tl]
        q = t[int(rng.integers(0, tl - ql)):][:ql].copy()
        q[rng.integers(0, ql, 2)] = alpha[rng.integers(0, 4, 2)]
        if k % 7 == 3:
            q = np.concatenate([q[:60], q[63:], alpha[rng.integers(0, 4, 3)]])
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    G = db.pack2bit(genome.tobytes())
    Q = db.pack2bit(b"".join(qs))
    qst = np.arange(n, dtype=np.int64) * ql
    params = (200, -150, 260, 11)
    off, sc, cg = ol.oracle_align_batch(ts, qs, params, ol.SOFTCLIP, nthreads=8)
    aligner.set_workspace(64 << 20)   # several chunks
    try:
        res = aligner.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64)
        assert (res.offsets == off).all() and (res.scores == sc).all() and list(res.cigars) == cg
        # ragged lengths, ascending starts in both arrays, every strategy
        rts, rqs = [], []
        for k in range(500):
            a, b = int(rng.integers(1, 300)), int(rng.integers(1, 200))
            t = alpha[rng.integers(0, 4, a)]
            rts.append(t.tobytes())
            rqs.append((np.resize(t[a // 3:], b) if k % 2 else alpha[rng.integers(0, 4, b)]).tobytes())
        T2, Q2 = db.pack2bit(b"".join(rts)), db.pack2bit(b"".join(rqs))
        tl2, ql2 = np.array([len(x) for x in rts], np.int32), np.array([len(x) for x in rqs], np.int32)
        ts2, qs2 = np.concatenate([[0], np.cumsum(tl2)[:-1]]).astype(np.int64), np.concatenate([[0], np.cumsum(ql2)[:-1]]).astype(np.int64)
        for strategy in ol.STRATEGIES:
            r2 = aligner.align_packed_2bit(T2, int(tl2.sum()), ts2, tl2, Q2, int(ql2.sum()), qs2, ql2, int(tl2.max()), int(ql2.max()), params, strategy)
            o2, s2, c2 = ol.oracle_align_batch(rts, rqs, params, strategy, nthreads=8)
            assert (r2.offsets == o2).all() and (r2.scores == s2).all() and list(r2.cigars) == c2, strategy
        # registered arrays on the ASCII entry: same bytes out as from pageable ones
        td, toff = sw.concat(ts)
        qd, qoff = sw.concat(qs)
        plain = aligner.align_packed(td, toff, qd, qoff, params, ol.SOFTCLIP, 64)
        arrays = [np.ascontiguousarray(x) for x in (td, toff, qd, qoff)]
        outs = (np.zeros(n, np.int32), np.zeros((n, 6), np.int32), np.zeros(n * 64, np.uint8), np.zeros(n, np.int32))
        for a_ in arrays + list(outs):
            aligner.register_host_buffer(a_)
        try:
            from mgl_amd import _lib
            rc = _lib.lib().mgl_sw_align_batch(aligner.ctx, n, arrays[0].ctypes.data, arrays[1].ctypes.data, arrays[2].ctypes.data, arrays[3].ctypes.data,
                                               200, -150, 260, 11, int(ol.SOFTCLIP), outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, 64,
                                               outs[3].ctypes.data)
            assert rc == 0
            assert (outs[0] == plain.offsets).all() and (outs[1] == plain.scores).all() and (outs[0] == off).all()
            assert [outs[2].reshape(n, 64)[k, : outs[3][k]].tobytes().decode() for k in range(n)] == cg
            # and the packed entry writing into the registered outputs
            aligner.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64, out=outs)
            assert (outs[0] == off).all() and (outs[1] == sc).all()
        finally:
            for a_ in arrays + list(outs):
                aligner.unregister_host_buffer(a_)
    finally:
        aligner.set_workspace(4 << 30)


def test_direct_host_entry_one_gated_launch_results_in_place(monkeypatch):
    """mgl_sw_align_batch_2bit with every array page-locked (round 4, the direct form): ONE launch of the persistent grid while the copy
    engines bring the inputs in behind a gate, results written by the waves straight into the caller's arrays in whole lines out of LDS.
    Against the chunked form (MGL_SW_DEBUG_HOST_DIRECT=0) byte for byte -- an odd number of pairs: the last tile is partial, the last
    lane holds one pair -- and against the oracle on a sample; then the same pairs device resident with and without the LDS hand-over."""
    import torch
    from mgl_amd import _lib, device_batch as db

    rng = np.random.default_rng(20264)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    genome = alpha[rng.integers(0, 4, 1 << 18)]
    n, tl, ql = 300_001, 256, 150
    win = rng.integers(0, len(genome) - tl, n).astype(np.int64)
    start = rng.integers(0, tl - ql - 4, n)
    idx = win[:, None] + start[:, None] + np.arange(ql)[None, :]
    reads = genome[idx]
    sub = rng.random(reads.shape) < 0.01
    reads = np.where(sub, alpha[rng.integers(0, 4, reads.shape)], reads).astype(np.uint8)
    gap = np.nonzero(rng.random(n) < 0.3)[0]          # a deletion of three bases in a third of the reads: the walks need their blocks
    for k in gap[:20000]:
        reads[k, 60:-3] = reads[k, 63:]
        reads[k, -3:] = genome[win[k] + start[k] + ql: win[k] + start[k] + ql + 3]
    G = db.pack2bit(genome.tobytes())
    Q = db.pack2bit(reads.tobytes())
    qst = np.arange(n, dtype=np.int64) * ql
    params = (200, -150, 260, 11)
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(8 << 30)

    def outs():
        return (np.zeros(n, np.int32), np.zeros((n, 6), np.int32), np.full(n * 64, 7, np.uint8), np.zeros(n, np.int32))

    direct, chunked = outs(), outs()
    regs = [G, win, Q, qst] + list(direct) + list(chunked)
    for x in regs:
        a.register_host_buffer(x)
    try:
        a.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64, out=direct)
        assert a.timing().fill_kernel == 7 and a.timing().dp_launches == 1, "the direct form: one launch of sw_dp16_lane_ck_kernel"
        monkeypatch.setenv("MGL_SW_DEBUG_HOST_DIRECT", "0")
        a.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64, out=chunked)
        monkeypatch.delenv("MGL_SW_DEBUG_HOST_DIRECT")
        for d, c in zip(direct, chunked):
            assert (d == c).all()
        sample = np.sort(rng.choice(n, 3000, replace=False))
        sample[-1] = n - 1
        ts = [genome[win[k]: win[k] + tl].tobytes() for k in sample]
        qs = [reads[k].tobytes() for k in sample]
        off, sc, cg = ol.oracle_align_batch(ts, qs, params, ol.SOFTCLIP, nthreads=8)
        assert (direct[0][sample] == off).all() and (direct[1][sample] == sc).all()
        assert [direct[2].reshape(n, 64)[k, : direct[3][k]].tobytes().decode() for k in sample] == cg
        assert (direct[2].reshape(n, 64)[np.arange(64)[None, :] >= direct[3][:, None]] == 0).all(), "zero behind the text"
        # a pair outside its array, found by the direct form's own check while the grid is already running: the grid is called off, the
        # call returns the argument error -- and the next call works
        win_ok = int(win[n - 5])
        win[n - 5] = len(genome) - tl + 1
        with pytest.raises(_lib.MglSwError) as bad:
            a.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64, out=chunked)
        assert bad.value.status == _lib.ERR_BAD_ARG
        win[n - 5] = win_ok
        a.align_packed_2bit(G, len(genome), win, None, Q, n * ql, qst, None, tl, ql, params, ol.SOFTCLIP, 64, out=chunked)
        for d, c in zip(direct, chunked):
            assert (d == c).all()
    finally:
        for x in regs:
            a.unregister_host_buffer(x)
    # device resident: the LDS hand-over of results on (default) and off
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy
    t_start, q_start = tq(win).to(dev), tq(qst).to(dev)
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("MGL_SW_DEBUG_COALESCED_OUT", mode)
        b = db.PackedBatch(tq(G).to(dev), t_start, None, tq(Q).to(dev), q_start, None, tl, ql, 64) if hasattr(db, "PackedBatch") else None
        if b is None:
            break
        b.run(a, params, ol.SOFTCLIP)
        torch.cuda.synchronize()
        res.append((b.offsets.cpu().numpy(), b.scores.cpu().numpy(), b.cigars.cpu().numpy(), b.cigar_len.cpu().numpy()))
    monkeypatch.delenv("MGL_SW_DEBUG_COALESCED_OUT", raising=False)
    if len(res) == 2:
        for x, y in zip(*res):
            assert (x == y).all()
        assert (res[0][0] == direct[0]).all() and (res[0][2].reshape(-1) == direct[2]).all()
    a.close()


def test_persistent_grid_counter_after_a_called_off_launch(monkeypatch):
    """Round 4's review: the persistent grid's tile counter was never reset and the host kept a copy of where each of its 64 ring words
    stood -- which a gated launch that ends early (a bad index array, a gate that stands still) did not move as far: 64 counter-using
    launches later the grid on that word left every tile beyond its first 2 048 undone, status 0.  Now the grid's last wave out zeroes the
    counter however the launch ended (sw_dp16_lane_ck.hip).  Here: a direct-form call of 700 000 pairs with the bad pair in the FIRST chunk
    (no gate ever opens: all 5 469 tiles undone), one called off in its LAST chunk, one whose gate "stands still" (a time-out of one tick:
    every wave gives up, the call goes the chunked way), then 70 further counter-using launches on the SAME context, each compared byte
    for byte with what a fresh context returns, and mgl_sw_ctx_check."""
    import torch
    from mgl_amd import _lib, device_batch as db

    rng = np.random.default_rng(5150)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    genome = alpha[rng.integers(0, 4, 1 << 18)]
    n, tl, ql = 700_000, 256, 150
    win = rng.integers(0, len(genome) - tl, n).astype(np.int64)
    start = rng.integers(0, tl - ql - 4, n)
    reads = genome[win[:, None] + start[:, None] + np.arange(ql)[None, :]]
    sub = rng.random(reads.shape) < 0.01
    reads = np.where(sub, alpha[rng.integers(0, 4, reads.shape)], reads).astype(np.uint8)
    for k in np.nonzero(rng.random(n) < 0.2)[0][:30000]:  # a deletion of three bases: these walks need their blocks
        reads[k, 60:-3] = reads[k, 63:]
        reads[k, -3:] = genome[win[k] + start[k] + ql: win[k] + start[k] + ql + 3]
    G, Q = db.pack2bit(genome.tobytes()), db.pack2bit(reads.tobytes())
    qst = np.arange(n, dtype=np.int64) * ql
    params = (200, -150, 260, 11)

    def outs():
        return (np.full(n, -77, np.int32), np.full((n, 6), -77, np.int32), np.full(n * 64, 7, np.uint8), np.full(n, -77, np.int32))

    def run(a, out, count=n):
        a.align_packed_2bit(G, len(genome), win[:count], None, Q, n * ql, qst[:count], None, tl, ql, params, ol.SOFTCLIP, 64,
                            out=tuple(x[: count * (64 if x.ndim == 1 and x.dtype == np.uint8 else 1)] for x in out))

    fresh = sw.MicrosoftSmithWaterman(0)
    fresh.set_workspace(8 << 30)
    good, work = outs(), outs()
    regs = [G, win, Q, qst] + list(good) + list(work)
    for x in regs:
        fresh.register_host_buffer(x)
    run(fresh, good)
    assert fresh.timing().fill_kernel == 7 and fresh.timing().dp_launches == 1, "the direct form: one gated launch"
    sample = np.sort(rng.choice(n, 2000, replace=False))
    off, sc, cg = ol.oracle_align_batch([genome[win[k]: win[k] + tl].tobytes() for k in sample], [reads[k].tobytes() for k in sample], params, ol.SOFTCLIP, nthreads=8)
    assert (good[0][sample] == off).all() and (good[1][sample] == sc).all()
    assert [good[2].reshape(n, 64)[k, : good[3][k]].tobytes().decode() for k in sample] == cg
    for x in regs:
        fresh.unregister_host_buffer(x)
    fresh.close()

    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(8 << 30)
    for x in regs:
        a.register_host_buffer(x)
    try:
        for bad_at, bad_value in ((1000, len(genome) - tl + 1), (n - 5, -1)):  # first chunk (262 400 pairs), last chunk
            keep = int(win[bad_at])
            win[bad_at] = bad_value
            with pytest.raises(_lib.MglSwError) as bad:
                run(a, work)
            assert bad.value.status == _lib.ERR_BAD_ARG
            win[bad_at] = keep
        a.check()
        # (many tiles through the gate: a grid of 64 wave slots takes its 5 469 tiles one after the other while the chunks arrive)
        monkeypatch.setenv("MGL_SW_DEBUG_LANE_SLOTS", "64")
        monkeypatch.setenv("MGL_SW_DEBUG_DIRECT_CHUNK", "65536")
        run(a, work)
        assert a.timing().dp_launches == 1
        monkeypatch.delenv("MGL_SW_DEBUG_LANE_SLOTS")
        monkeypatch.delenv("MGL_SW_DEBUG_DIRECT_CHUNK")
        for g, w in zip(good, work):
            assert (g == w).all()
        # the ring goes round: 70 counter-using launches (300 001 pairs: 2 344 tiles on 2 048 slots), the direct form each time
        m = 300_001
        for it in range(70):
            for w in work:
                w[...] = 9
            run(a, work, m)
            assert a.timing().dp_launches == 1
            for g, w in zip(good, work):
                k = m * (64 if w.ndim == 1 and w.dtype == np.uint8 else 1)
                assert (g[:k] == w[:k]).all(), it
        # a gate that stands still: every wave gives up at its first look, the call is done again the chunked way -- and is right
        monkeypatch.setenv("MGL_SW_DEBUG_GATE_TIMEOUT_TICKS", "1")
        for w in work:
            w[...] = 9
        run(a, work)
        monkeypatch.delenv("MGL_SW_DEBUG_GATE_TIMEOUT_TICKS")
        assert a.timing().dp_launches > 1, "the chunked form took over"
        for g, w in zip(good, work):
            assert (g == w).all()
        a.check()
    finally:
        for x in regs:
            a.unregister_host_buffer(x)
    # ... and once more round the ring through the device entry on the same context
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy
    m = 300_001
    b = db.PackedBatch(tq(G).to(dev), tq(win[:m]).to(dev), None, tq(Q).to(dev), tq(qst[:m]).to(dev), None, tl, ql, 64)
    for it in range(70):
        b.offsets.fill_(9)
        b.cigars.fill_(9)
        b.run(a, params, ol.SOFTCLIP)
        torch.cuda.synchronize()
        a.check()
        assert (b.offsets.cpu().numpy() == good[0][:m]).all() and (b.cigars.cpu().numpy().reshape(-1) == good[2][: m * 64]).all(), it
        assert (b.scores.cpu().numpy() == good[1][:m]).all() and (b.cigar_len.cpu().numpy() == good[3][:m]).all()
    a.close()


def test_ascii_direct_host_entry(monkeypatch):
    """mgl_sw_align_batch_status with every array page-locked and one geometry (round 5): the reference's own wire format -- ASCII bases
    in host memory, MicrosoftSmithWaterman.java:71-86 -- through ONE gated launch of the persistent grid while the copy engines bring the
    bases in; the offsets never cross the link (a kernel writes k * tl), the results are written by the waves into the caller's arrays.
    Against the chunked form byte for byte, with targets that hold N and lower-case letters (raw byte semantics, sw.cpp:55), small chunks
    on a small grid (many tiles through the gate), and the oracle on a sample."""
    rng = np.random.default_rng(808)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    n, tl, ql = 300_001, 256, 150
    t = alpha[rng.integers(0, 4, (n, tl))]
    start = rng.integers(0, tl - ql - 4, n)
    q = t[np.arange(n)[:, None], start[:, None] + np.arange(ql)[None, :]].copy()
    q = np.where(rng.random(q.shape) < 0.01, alpha[rng.integers(0, 4, q.shape)], q).astype(np.uint8)
    odd = rng.choice(n, 4000, replace=False)
    t[odd[:2000], 100] = ord("N")          # N == N matches (sw.cpp:55), and the wave that holds it stages raw bytes
    q[odd[:1000], 100 - start[odd[:1000]].clip(0, 100)] = ord("N")
    t[odd[2000:], 30:40] |= 0x20           # lower case differs from upper case
    for k in np.nonzero(rng.random(n) < 0.2)[0][:20000]:
        q[k, 70:-3] = q[k, 73:]
    td, qd = np.ascontiguousarray(t.reshape(-1)), np.ascontiguousarray(q.reshape(-1))
    toff, qoff = np.arange(n + 1, dtype=np.int64) * tl, np.arange(n + 1, dtype=np.int64) * ql
    params = (200, -150, 260, 11)
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(8 << 30)

    def outs():
        return (np.full(n, -7, np.int32), np.full((n, 6), -7, np.int32), np.full(n * 64, 7, np.uint8), np.full(n, -7, np.int32))

    from mgl_amd import _lib
    L = _lib.lib()

    def run(o):
        rc = L.mgl_sw_align_batch_status(a.ctx, n, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, *params, int(ol.SOFTCLIP), o[0].ctypes.data,
                                         o[1].ctypes.data, o[2].ctypes.data, 64, o[3].ctypes.data, None)
        assert rc == 0, rc

    direct, small, chunked = outs(), outs(), outs()
    regs = [td, qd] + list(direct) + list(small) + list(chunked)
    for x in regs:
        a.register_host_buffer(x)
    try:
        run(direct)
        assert a.timing().fill_kernel == 7 and a.timing().dp_launches == 1, "the direct form: one launch of sw_dp16_lane_ck_kernel"
        monkeypatch.setenv("MGL_SW_DEBUG_LANE_SLOTS", "96")
        monkeypatch.setenv("MGL_SW_DEBUG_DIRECT_CHUNK", "16384")
        run(small)
        assert a.timing().dp_launches == 1
        monkeypatch.delenv("MGL_SW_DEBUG_LANE_SLOTS")
        monkeypatch.delenv("MGL_SW_DEBUG_DIRECT_CHUNK")
        monkeypatch.setenv("MGL_SW_DEBUG_HOST_DIRECT", "0")
        run(chunked)
        assert a.timing().dp_launches > 1
        monkeypatch.delenv("MGL_SW_DEBUG_HOST_DIRECT")
        for d, s_, c in zip(direct, small, chunked):
            assert (d == c).all() and (s_ == c).all()
        sample = np.sort(np.concatenate([rng.choice(n, 2000, replace=False), odd[:300], odd[2000:2300], [n - 1]]))
        off, sc, cg = ol.oracle_align_batch([t[k].tobytes() for k in sample], [q[k].tobytes() for k in sample], params, ol.SOFTCLIP, nthreads=8)
        assert (direct[0][sample] == off).all() and (direct[1][sample] == sc).all()
        assert [direct[2].reshape(n, 64)[k, : direct[3][k]].tobytes().decode() for k in sample] == cg
    finally:
        for x in regs:
            a.unregister_host_buffer(x)
    a.close()


def test_mixed_lengths_from_host_memory_are_sorted_on_the_device(monkeypatch):
    """mgl_sw_align_batch_2bit WITHOUT the uniform flag (round 5): the chunks of a host batch of mixed geometries are sorted by (tl, ql)
    on the device -- whole waves of one geometry through the checkpointed lane kernel, the rest through the packed and int32 kernels --
    each chunk's inputs brought in front of its sort, two chunks ahead of the fills, results back in the caller's order.  400 000 reads of
    100-150 bases (a few odd geometries, a few one-base queries) against 256-base windows, from page-locked and from pageable arrays, in
    one chunk and in four: identical to the device-resident entry, and to the oracle on a sample."""
    import torch
    from mgl_amd import device_batch as db

    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    n, tl = 400_000, 256
    genome = alpha[rng.integers(0, 4, 1 << 20)]
    win = rng.integers(0, len(genome) - tl, n).astype(np.int64)
    qlen = rng.integers(100, 151, n).astype(np.int32)
    qlen[rng.choice(n, 300, replace=False)] = rng.integers(1, 100, 300)
    tlen = np.full(n, tl, np.int32)
    tlen[rng.choice(n, 300, replace=False)] = rng.integers(150, 256, 300)
    start = rng.integers(0, 100, n)
    reads = genome[win[:, None] + start[:, None] + np.arange(150)[None, :]]
    reads = np.where(rng.random(reads.shape) < 0.02, alpha[rng.integers(0, 4, reads.shape)], reads).astype(np.uint8)
    for k in np.nonzero(rng.random(n) < 0.1)[0]:
        reads[k, 50:-2] = reads[k, 52:]   # a deletion of two bases
    G, Q = db.pack2bit(genome.tobytes()), db.pack2bit(reads.tobytes())
    qst = np.arange(n, dtype=np.int64) * 150
    params = (200, -150, 260, 11)
    a = sw.MicrosoftSmithWaterman(0)
    a.set_workspace(8 << 30)
    dev = torch.device("cuda", 0)
    tq = torch.from_numpy
    want = db.PackedBatch(tq(G).to(dev), tq(win).to(dev), tq(tlen).to(dev), tq(Q).to(dev), tq(qst).to(dev), tq(qlen).to(dev), tl, 150, 64)
    want.run(a, params, ol.SOFTCLIP)
    torch.cuda.synchronize()
    w = (want.offsets.cpu().numpy(), want.scores.cpu().numpy(), want.cigars.cpu().numpy().reshape(-1), want.cigar_len.cpu().numpy())
    assert int((want.status != 0).sum()) == 0
    sample = np.sort(rng.choice(n, 1500, replace=False))
    ts = [genome[win[k]: win[k] + tlen[k]].tobytes() for k in sample]
    qs = [reads[k, : qlen[k]].tobytes() for k in sample]
    off, sc, cg = ol.oracle_align_batch(ts, qs, params, ol.SOFTCLIP, nthreads=8)
    assert (w[0][sample] == off).all() and (w[1][sample] == sc).all()
    assert [w[2].reshape(n, 64)[k, : w[3][k]].tobytes().decode() for k in sample] == cg

    def outs():
        return (np.full(n, -7, np.int32), np.full((n, 6), -7, np.int32), np.full(n * 64, 7, np.uint8), np.full(n, -7, np.int32))

    pinned = [G, win, Q, qst, tlen, qlen]
    for chunk_env, pin in ((None, True), ("131072", True), ("131072", False)):
        got = outs()
        regs = (pinned + list(got)) if pin else []
        for x in regs:
            a.register_host_buffer(x)
        if chunk_env:
            monkeypatch.setenv("MGL_SW_DEBUG_HOST_SORT_CHUNK", chunk_env)
            monkeypatch.setenv("MGL_SW_DEBUG_LANE_GROUP_MIN", "128")  # (a chunk's whole waves of one geometry get their lane-kernel launch below 131 072 pairs too)
        try:
            a.align_packed_2bit(G, len(genome), win, tlen, Q, n * 150, qst, qlen, tl, 150, params, ol.SOFTCLIP, 64, out=got)
            tm = a.timing()
            assert tm.fill_kernel == 7, "the bulk of every chunk: sw_dp16_lane_ck_kernel"
            assert tm.dp_launches == (1 if chunk_env is None else 4), tm.dp_launches  # (65 536, 131 072, 131 072, 72 320 pairs: a short first chunk, no launch for a sliver)
        finally:
            monkeypatch.delenv("MGL_SW_DEBUG_HOST_SORT_CHUNK", raising=False)
            monkeypatch.delenv("MGL_SW_DEBUG_LANE_GROUP_MIN", raising=False)
            for x in regs:
                a.unregister_host_buffer(x)
        for g_, w_ in zip(got, w):
            assert (g_ == w_).all(), (chunk_env, pin)
    a.close()


def test_binary_cigar_output(aligner):
    """MGL_SW_FLAG_BINARY_CIGAR: BAM-style uint32 elements carry the same elements in the same order as the text."""
    import torch
    from mgl_amd import device_batch as db

    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP][:300] + \
           [g for g in golden_io.load("random") if g.params == (200, -150, 260, 11) and g.strategy == ol.SOFTCLIP]
    td, toff = sw.concat([g.t for g in rows])
    qd, qoff = sw.concat([g.q for g in rows])
    b = db.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=4 * 700)
    b.run(aligner, rows[0].params, ol.SOFTCLIP, binary_cigar=True)
    torch.cuda.synchronize()
    assert int((b.status != 0).sum()) == 0
    got = b.cigar_elements()
    for k, g in enumerate(rows):
        assert got[k] == g.cigar, (k, got[k], g.cigar)
    assert (b.offsets.cpu().numpy() == np.array([g.offset for g in rows])).all()


def test_parameter_and_alphabet_fuzz(aligner):
    """Unusual scoring parameters (zero gap costs, zero mismatch, huge match) and arbitrary byte alphabets
    (0..255, compared raw like sw.cpp:55) through both fill kernels: uniform batches take the packed-int16
    kernel when the score range allows it, ragged ones the int32 kernel."""
    rng = np.random.default_rng(314159)
    psets = [(200, -150, 260, 11), (1, 0, 0, 0), (5, -3, 0, 1), (5, -3, 7, 0), (1000, -1000, 3000, 100), (2, -7, 1, 1),
             (127, -128, 255, 1), (9, -9, 9, 9), (1, -1, 50, 50)]
    for it, params in enumerate(psets):
        for uniform in (True, False):
            ts, qs = [], []
            tl0, ql0 = int(rng.integers(20, 90)), int(rng.integers(20, 90))
            for k in range(41):
                tl = tl0 if uniform else int(rng.integers(1, 90))
                ql = ql0 if uniform else int(rng.integers(1, 90))
                hi = 256 if it % 2 else 3  # full byte range or a tiny alphabet (many ties)
                t = rng.integers(0, hi, tl, dtype=np.uint8)
                q = t[: ql].copy() if (k % 2 and tl >= ql) else rng.integers(0, hi, ql, dtype=np.uint8)
                if k % 4 == 1 and len(q) > 4:
                    q[len(q) // 2] ^= 1
                ts.append(t.tobytes())
                qs.append(np.resize(q, ql).tobytes())
            for strategy in ol.STRATEGIES:
                res = aligner.align_batch(ts, qs, params, strategy)
                off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
                ctx = (params, strategy, uniform, bool(aligner.timing().packed16))
                assert (res.offsets == off).all(), ctx
                assert (res.scores == sc).all(), ctx
                assert res.cigars == cg, ctx


def test_contexts_on_many_threads():
    """One mgl_sw_ctx per host thread, all on the same GPU, running batches concurrently (the reference's
    alignNative is re-entrant, ..._MicrosoftSmithWaterman.cpp:44-71)."""
    import threading

    rows = [g for g in golden_io.load("window") if g.strategy == ol.SOFTCLIP]
    ts, qs = [g.t for g in rows], [g.q for g in rows]
    errors = []

    def worker(seed):
        try:
            a = sw.MicrosoftSmithWaterman(0)
            for rep in range(3):
                res = a.align_batch(ts, qs, rows[0].params, ol.SOFTCLIP)
                for k, g in enumerate(rows):
                    if (int(res.offsets[k]), res.cigars[k], tuple(int(x) for x in res.scores[k])) != (g.offset, g.cigar, g.score):
                        errors.append((seed, rep, k))
                        return
            a.close()
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
