"""Substitution-matrix ("protein") scoring, SURVEY.md section 8f rank 4 / BASELINE configs[4].  The reference has no
such path, so NO parity with it is claimed: the GPU is compared (bit-exactly) with the CPU restatement's own
extension (oracle swo_*_matrix), and that extension with an independent textbook DP written here."""
import numpy as np
import pytest

import oracle_lib as ol
from mgl_amd import device_batch, protein, smithwaterman as sw


def textbook(t, q, code, mat, o, e, indel):
    """Affine-gap DP straight from the recurrence (no zero floor, mgl's borders): returns H as a numpy array."""
    tl, ql = len(t), len(q)
    NEG = -10 ** 9
    H = np.zeros((tl + 1, ql + 1), dtype=np.int64)
    E = np.full((tl + 2, ql + 1), NEG, dtype=np.int64)  # E[i][j]: gap entering (i, j) from above
    F = np.full((tl + 1, ql + 2), NEG, dtype=np.int64)  # F[i][j]: gap entering (i, j) from the left
    b = lambda k: (-o - (k - 1) * e) if (indel and k > 0) else 0
    for j in range(ql + 1):
        H[0][j] = b(j)
        E[1][j] = H[0][j] - o
    for i in range(1, tl + 1):
        H[i][0] = b(i)
        F[i][1] = H[i][0] - o
        for j in range(1, ql + 1):
            diag = H[i - 1][j - 1] + int(mat[code[t[i - 1]], code[q[j - 1]]])
            H[i][j] = max(diag, F[i][j], E[i][j])
            E[i + 1][j] = max(H[i][j] - o, E[i][j] - e)
            F[i][j + 1] = max(H[i][j] - o, F[i][j] - e)
    return H


def oracle_matrix_batch(ts, qs, code, mat, o, e, strategy, stride):
    import ctypes as C

    td, toff = sw.concat(ts)
    qd, qoff = sw.concat(qs)
    n = len(ts)
    off = np.zeros(n, np.int32); sc = np.zeros((n, 6), np.int32); cg = np.zeros(n * stride, np.uint8); ln = np.zeros(n, np.int32)
    L = ol.oracle()
    L.swo_align_batch_matrix.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
    rc = L.swo_align_batch_matrix(n, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, code.ctypes.data,
                                  mat.ctypes.data, o, e, strategy, 4, off.ctypes.data, sc.ctypes.data, cg.ctypes.data, stride,
                                  ln.ctypes.data)
    assert rc == 0
    return off, sc, [cg[k * stride: k * stride + ln[k]].tobytes().decode() for k in range(n)]


def test_restatement_extension_against_textbook():
    """CPU only: last-column / last-row maxima of the extension equal those of the textbook DP."""
    rng = np.random.default_rng(2)
    code, mat = protein.blosum62()
    for trial in range(40):
        tl, ql = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        t = protein.random_proteins(rng, 1, tl)[0].tobytes()
        q = protein.random_proteins(rng, 1, ql)[0].tobytes()
        for strategy in ol.STRATEGIES:
            indel = strategy in (ol.INDEL, ol.LEAD_INDEL)
            H = textbook(t, q, code, mat, 11, 1, indel)
            _, sc, _ = oracle_matrix_batch([t], [q], code, mat, 11, 1, strategy, 4 * (tl + ql) + 16)
            mqe, mqe_t, mx = int(sc[0][0]), int(sc[0][1]), int(sc[0][2])
            col = H[1:, ql]
            assert mqe == col.max() and mqe_t == max(i + 1 for i in range(tl) if col[i] == col.max())
            assert mx == max(col.max(), H[tl, 1:].max())


@pytest.mark.gpu
def test_blosum62_batches_bit_exact():
    import torch

    rng = np.random.default_rng(4)
    code, mat = protein.blosum62()
    a = sw.MicrosoftSmithWaterman(0)
    assert a.load()
    for strategy, (o, e) in zip(ol.STRATEGIES, [(11, 1), (10, 2), (5, 5), (12, 1)]):
        ts, qs = [], []
        for k in range(300):
            tl, ql = int(rng.integers(1, 700)), int(rng.integers(1, 400))
            t = protein.random_proteins(rng, 1, tl)[0]
            if k % 3 and tl > 20:   # a diverged homologue of part of the target
                s = int(rng.integers(0, tl - 10)); q = t[s:s + ql].copy()
                mut = rng.random(len(q)) < 0.3
                q[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0] if mut.any() else q[mut]
                if len(q) > 8: q = np.delete(q, rng.integers(0, len(q), size=2))
            else:
                q = protein.random_proteins(rng, 1, ql)[0]
            if k % 17 == 0: q = np.concatenate([q, np.frombuffer(b"BZX*ux", np.uint8)])   # ambiguity codes, lower case, junk
            ts.append(t.tobytes()); qs.append(q.tobytes())
        td, toff = sw.concat(ts); qd, qoff = sw.concat(qs)
        stride = 2 * 800
        b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=stride)
        protein.run_matrix(b, a, code, mat, o, e, strategy)
        torch.cuda.synchronize()
        assert int((b.status != 0).sum()) == 0
        off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, o, e, strategy, stride)
        assert (b.offsets.cpu().numpy() == off).all()
        assert (b.scores.cpu().numpy() == sc).all()
        assert b.cigar_strings() == cg
    # one geometry per block of eight pairs: the packed-int16 kernel with the table S - max S in LDS, all strategies
    for strategy, (o, e) in zip(ol.STRATEGIES, [(11, 1), (10, 2), (7, 3), (12, 1)]):
        ts, qs = [], []
        for g in range(40):
            tl, ql = int(rng.integers(20, 500)), int(rng.integers(33, 330))
            for k in range(8):
                t = protein.random_proteins(rng, 1, tl)[0]
                if k % 2 and tl > ql:
                    s0 = int(rng.integers(0, tl - ql + 1)); q = t[s0:s0 + ql].copy()
                    mut = rng.random(ql) < 0.35
                    q[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0] if mut.any() else q[mut]
                else:
                    q = protein.random_proteins(rng, 1, ql)[0]
                ts.append(t.tobytes()); qs.append(q.tobytes())
        td, toff = sw.concat(ts); qd, qoff = sw.concat(qs)
        b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=1200)
        protein.run_matrix(b, a, code, mat, o, e, strategy, grouped=True)
        torch.cuda.synchronize()
        assert a.timing().packed16 == 1 and int((b.status != 0).sum()) == 0
        off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, o, e, strategy, 1200)
        assert (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
        protein.run_matrix(b, a, code, mat, o, e, strategy, grouped=False)   # the same pairs through the int32 kernel
        torch.cuda.synchronize()
        assert a.timing().packed16 == 0
        assert (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
    # a DNA match / mismatch matrix reproduces the reference's scoring exactly
    dcode = np.zeros(256, np.uint8)
    for k, ch in enumerate(b"ACGT"): dcode[ch] = k + 1
    dmat = np.full((32, 32), -75, np.int8)   # int8 range: (100, -75) instead of (200, -150)
    for k in range(1, 5): dmat[k, k] = 100
    gs = [g for g in __import__("golden_io").load("window")[:64]]
    ts = [g.t for g in gs]; qs = [g.q for g in gs]
    td, toff = sw.concat(ts); qd, qoff = sw.concat(qs)
    b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=512)
    protein.run_matrix(b, a, dcode, dmat, 130, 6, ol.SOFTCLIP)
    torch.cuda.synchronize()   # the host entry below runs on the context's own stream and shares its workspace
    ref = a.align_batch(ts, qs, (100, -75, 130, 6), ol.SOFTCLIP, cigar_stride=512)
    assert (b.offsets.cpu().numpy() == ref.offsets).all() and (b.scores.cpu().numpy() == ref.scores).all()
    assert b.cigar_strings() == list(ref.cigars)
    # a 2 000-residue query takes the one-pair-per-wave variant of the matrix kernel
    long_t = protein.random_proteins(rng, 1, 2500)[0]
    long_q = np.concatenate([long_t[300:1500], protein.random_proteins(rng, 1, 800)[0]])
    b = device_batch.from_host(*sw.concat([long_t.tobytes()] * 3), *sw.concat([long_q.tobytes()] * 3), "cuda:0", cigar_stride=8192)
    protein.run_matrix(b, a, code, mat, 11, 1, ol.SOFTCLIP)
    torch.cuda.synchronize()
    off, sc, cg = oracle_matrix_batch([long_t.tobytes()] * 3, [long_q.tobytes()] * 3, code, mat, 11, 1, ol.SOFTCLIP, 8192)
    assert (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
    # too long a query for the matrix kernel's LDS carve
    long_q = protein.random_proteins(rng, 1, 5000)[0].tobytes()
    b = device_batch.from_host(*sw.concat([long_q]), *sw.concat([long_q]), "cuda:0")
    with pytest.raises(Exception) as ex:
        protein.run_matrix(b, a, code, mat)
    assert "too long" in str(ex.value)
    a.close()


def _tiles(rng, shapes, last_count):
    """Tiles of 128 pairs that share their target (the last one `last_count` pairs): queries of one length per tile, diverged
    fragments of the target, unrelated sequences, ambiguity codes, lower case and junk bytes among them."""
    ts, qs = [], []
    for k, (tl, ql) in enumerate(shapes):
        t = protein.random_proteins(rng, 1, tl)[0]
        for p in range(last_count if k == len(shapes) - 1 else 128):
            if p % 3 and tl >= ql:
                s0 = int(rng.integers(0, tl - ql + 1)); q = t[s0:s0 + ql].copy()
                mut = rng.random(ql) < 0.3
                if mut.any(): q[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0]
            else:
                q = protein.random_proteins(rng, 1, ql)[0]
            if p % 19 == 0 and ql >= 6: q[-6:] = np.frombuffer(b"BZX*u!", np.uint8)
            ts.append(t.tobytes()); qs.append(q.tobytes())
    return ts, qs


def _shared_batch(ts, qs, dev, stride):
    """IndexedBatch over ONE copy of each tile's target: pair k points at its tile's target (the promise of MGL_SW_FLAG_SHARED_TARGET)."""
    import torch

    tdata, t_start, t_len, seen = [], [], [], {}
    pos = 0
    for k, t in enumerate(ts):
        key = (k // 128, t)
        if key not in seen:
            seen[key] = pos; tdata.append(np.frombuffer(t, np.uint8)); pos += len(t)
        t_start.append(seen[key]); t_len.append(len(t))
    qd, qoff = sw.concat(qs)
    return protein.IndexedBatch(torch.from_numpy(np.concatenate(tdata)).to(dev), torch.tensor(t_start, dtype=torch.int64, device=dev),
                                torch.tensor(t_len, dtype=torch.int32, device=dev), torch.from_numpy(qd).to(dev),
                                torch.from_numpy(qoff[:-1].copy()).to(dev), torch.from_numpy(np.diff(qoff).astype(np.int32)).to(dev),
                                max(len(t) for t in ts), max(len(q) for q in qs), stride)


@pytest.mark.gpu
def test_tiles_that_share_their_target_bit_exact(monkeypatch):
    """MGL_SW_FLAG_SHARED_TARGET (sw_dp16_lane_matrix.hip): tiles of 128 pairs on one target, two pairs per lane, the scores of a column out
    of the strip's profile -- against the CPU restatement's extension, every strategy, strips that end inside a target, queries of 1 .. 301
    residues, a short last tile with an odd pair count; a grid of five wave slots (the tiles outnumber them: the counter, twice on one
    context); parameters the byte table cannot hold (the flag is then read as the grouped promise); a tile that breaks the promise."""
    import torch

    rng = np.random.default_rng(11)
    code, mat = protein.blosum62()
    dev = torch.device("cuda", 0)
    shapes = [(1, 1), (5, 3), (31, 4), (32, 5), (33, 7), (63, 33), (64, 150), (65, 301), (100, 2), (257, 64), (700, 300), (96, 299), (40, 40), (333, 130)]
    a = sw.MicrosoftSmithWaterman(0)
    for strategy, (o, e) in zip(ol.STRATEGIES, [(11, 1), (10, 2), (5, 5), (12, 1)]):
        ts, qs = _tiles(rng, shapes, 77)
        b = _shared_batch(ts, qs, dev, 1024)
        off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, o, e, strategy, 1024)
        for slots in (None, "5", "5"):
            if slots: monkeypatch.setenv("MGL_SW_DEBUG_LANE_SLOTS", slots)
            b.offsets.fill_(-7); b.scores.fill_(-7); b.status.fill_(-7)
            protein.run_matrix(b, a, code, mat, o, e, strategy, shared_target=True)
            torch.cuda.synchronize()
            monkeypatch.delenv("MGL_SW_DEBUG_LANE_SLOTS", raising=False)
            assert a.fill_kernel_name(a.timing()) == "sw_dp16_lane_matrix_kernel"
            assert int((b.status != 0).sum()) == 0
            assert (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
        # MGL_SW_FLAG_SCORE_ONLY: the same six fields without flags, regions or walk
        b.scores.fill_(-7); b.offsets.fill_(-7); b.cigar_len.fill_(-7)
        protein.run_matrix(b, a, code, mat, o, e, strategy, shared_target=True, score_only=True)
        torch.cuda.synchronize()
        assert a.fill_kernel_name(a.timing()) == "sw_dp16_lane_matrix_kernel"
        assert (b.scores.cpu().numpy() == sc).all() and int(b.offsets.abs().sum()) == 0 and int(b.cigar_len.abs().sum()) == 0 and int((b.status != 0).sum()) == 0
        # MGL_SW_FLAG_BINARY_CIGAR: BAM-style uint32 elements, the same elements in the same order as the text
        protein.run_matrix(b, a, code, mat, o, e, strategy, shared_target=True, binary_cigar=True)
        torch.cuda.synchronize()
        raw, ln = b.cigars.cpu().numpy(), b.cigar_len.cpu().numpy()
        for k in range(0, len(cg), 37):
            el = np.frombuffer(raw[k, : ln[k]].tobytes(), dtype="<u4")
            assert "".join(f"{int(v) >> 4}{'MIDNS'[int(v) & 15]}" for v in el) == cg[k]
        a.check()
    # gap penalties under which an entry S + e + o is negative: the byte table cannot hold them, the batch takes the packed kernel
    ts, qs = _tiles(rng, shapes[3:9], 128)
    b = _shared_batch(ts, qs, dev, 1024)
    protein.run_matrix(b, a, code, mat, 2, 1, ol.SOFTCLIP, shared_target=True)
    torch.cuda.synchronize()
    assert a.fill_kernel_name(a.timing()) == "sw_dp16_kernel"
    off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, 2, 1, ol.SOFTCLIP, 1024)
    assert (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
    # a broken promise: one pair of tile 2 points one residue further into the targets, one pair of tile 4 has a shorter query -- those two
    # tiles are NOT computed (MGL_SW_ERR_BAD_ARG for their pairs), the others are right
    ts, qs = _tiles(rng, [(64, 50)] * 6, 128)
    b = _shared_batch(ts, qs, dev, 512)
    b.t_off[2 * 128 + 77] += 1
    b.q_len[4 * 128 + 5] -= 1
    b.status.fill_(0)
    protein.run_matrix(b, a, code, mat, 11, 1, ol.SOFTCLIP, shared_target=True)
    torch.cuda.synchronize()
    st = b.status.cpu().numpy().reshape(6, 128)
    assert (st[[2, 4]] == 1).all() and (st[[0, 1, 3, 5]] == 0).all()
    off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, 11, 1, ol.SOFTCLIP, 512)
    good = np.repeat(np.array([1, 1, 0, 1, 0, 1], bool), 128)
    assert (b.offsets.cpu().numpy()[good] == off[good]).all() and (b.scores.cpu().numpy()[good] == sc[good]).all()
    assert [c for c, g in zip(b.cigar_strings(), good) if g] == [c for c, g in zip(cg, good) if g]
    # a caller whose max_tl is smaller than a tile's target: that tile does not fit its region and is refused the same way
    ts, qs = _tiles(rng, [(40, 30), (90, 30), (40, 30)], 128)
    b = _shared_batch(ts, qs, dev, 512)
    b.max_tl = 64
    protein.run_matrix(b, a, code, mat, 11, 1, ol.SOFTCLIP, shared_target=True)
    torch.cuda.synchronize()
    st = b.status.cpu().numpy().reshape(3, 128)
    assert (st[1] == 1).all() and (st[[0, 2]] == 0).all()
    a.close()


@pytest.mark.gpu
def test_database_search_layout_every_pair_where_it_says():
    """mgl_amd.protein.DatabaseSearch: 136 queries against 9 database sequences -- tiles of 128 per sequence (longest first), 8 queries beyond
    whole tiles, two sequences too long for the tile class -- every (d, q) found by where() and equal to the CPU restatement's extension."""
    import torch

    rng = np.random.default_rng(5)
    code, mat = protein.blosum62()
    lens = np.array([60, 333, 41, 700, 129, 64, 1200, 95, 256])
    db_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    db = protein.random_proteins(rng, 1, int(db_off[-1]))[0]
    Q, QL = 136, 70
    queries = protein.random_proteins(rng, Q, QL)
    for q in range(0, Q, 3):   # diverged fragments of database sequences
        d = int(rng.integers(0, len(lens)))
        if lens[d] >= QL:
            s0 = int(rng.integers(0, lens[d] - QL + 1)); frag = db[db_off[d] + s0: db_off[d] + s0 + QL].copy()
            mut = rng.random(QL) < 0.3
            frag[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0]
            queries[q] = frag
    ds = protein.DatabaseSearch(db, db_off, queries, torch.device("cuda", 0), cigar_stride=1024, shared_max_tl=512)
    assert ds.shared.n == 7 * 128 and ds.rest.n == 7 * 8 and ds.long.n == 2 * Q and ds.shared_max_tl == 512
    a = sw.MicrosoftSmithWaterman(0)
    ds.run(a, code, mat, 11, 1, ol.SOFTCLIP)
    torch.cuda.synchronize()
    ts = [db[db_off[d]:db_off[d + 1]].tobytes() for d in range(len(lens)) for q in range(Q)]
    qs = [queries[q].tobytes() for d in range(len(lens)) for q in range(Q)]
    off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, 11, 1, ol.SOFTCLIP, 1024)
    host = {id(b): (b.offsets.cpu().numpy(), b.scores.cpu().numpy(), b.status.cpu().numpy(), b.cigar_strings()) for b in ds.batches()}
    seen = set()
    for d in range(len(lens)):
        for q in range(Q):
            b, p = ds.where(d, q)
            o_, s_, st_, c_ = host[id(b)]
            k = d * Q + q
            assert (id(b), p) not in seen and st_[p] == 0 and o_[p] == off[k] and (s_[p] == sc[k]).all() and c_[p] == cg[k], (d, q)
            seen.add((id(b), p))
    assert len(seen) == len(lens) * Q == sum(b.n for b in ds.batches())
    a.close()
