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
