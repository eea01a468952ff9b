"""Substitution-matrix ("protein") scoring on top of the Smith-Waterman core (SURVEY.md section 8f rank 4,
BASELINE.json configs[4]).  The reference has no such path (it scores by byte equality, sw.cpp:55): this is an
extension -- the same recurrence, overhang strategies and traceback with a 32 x 32 int8 matrix in LDS -- and no
parity with the reference is claimed for it.

BLOSUM62 below is the standard NCBI table (Henikoff & Henikoff 1992), order ARNDCQEGHILKMFPSTWYVBZX*.
"""
import ctypes as C

import numpy as np

from . import _lib

AMINO = "ARNDCQEGHILKMFPSTWYVBZX*"
_BLOSUM62_ROWS = """
 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4
-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4
-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4
-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4
 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4
-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4
-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4
-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4
-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4
-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4
-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4
-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4
-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4
 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4
-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4
 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4
-2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4
-1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4
-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1
"""


def blosum62():
    """(code uint8[256], matrix int8[32, 32]): residues in AMINO order get codes 0..23 (lower case too), every other
    byte the code of 'X'; unused matrix rows / columns score like 'X'."""
    rows = np.array([[int(x) for x in line.split()] for line in _BLOSUM62_ROWS.strip().splitlines()], dtype=np.int8)
    assert rows.shape == (24, 24) and (rows == rows.T).all()
    x = AMINO.index("X")
    mat = np.empty((32, 32), dtype=np.int8)
    mat[:] = rows[x, x]
    mat[:24, :24] = rows
    mat[24:, :24] = rows[x]
    mat[:24, 24:] = rows[:, x:x + 1]
    code = np.full(256, x, dtype=np.uint8)
    for k, ch in enumerate(AMINO):
        code[ord(ch)] = k
        code[ord(ch.lower())] = k
    return code, mat


# Robinson & Robinson (1991) background frequencies of the 20 standard residues, for synthetic proteins
_FREQ = dict(A=7.8, R=5.1, N=4.5, D=5.4, C=1.9, Q=4.3, E=6.3, G=7.4, H=2.2, I=5.1, L=9.1, K=5.7, M=2.2, F=3.9, P=5.2,
             S=7.1, T=5.8, W=1.3, Y=3.2, V=6.4)


def random_proteins(rng, n, length):
    """n random sequences of ``length`` residues (uint8 [n, length]) with natural background frequencies."""
    letters = np.frombuffer("".join(_FREQ).encode(), dtype=np.uint8)
    p = np.array(list(_FREQ.values()))
    return letters[rng.choice(len(letters), size=(n, length), p=p / p.sum())]


class IndexedBatch:
    """Pairs addressed by (start, length) into shared sequence arrays -- a database search: pair k = targets[t_off[k] :
    t_off[k] + t_len[k]] against queries[q_off[k] : q_off[k] + q_len[k]] (torch CUDA tensors)."""

    def __init__(self, targets, t_off, t_len, queries, q_off, q_len, max_tl, max_ql, cigar_stride):
        import torch

        self.targets, self.t_off, self.t_len = targets, t_off, t_len
        self.queries, self.q_off, self.q_len = queries, q_off, q_len
        self.n, self.max_tl, self.max_ql, self.cigar_stride = t_off.numel(), int(max_tl), int(max_ql), int(cigar_stride)
        dev = targets.device
        self.offsets = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.scores = torch.empty((self.n, 6), dtype=torch.int32, device=dev)
        self.cigars = torch.empty((self.n, self.cigar_stride), dtype=torch.uint8, device=dev)
        self.cigar_len = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.status = torch.empty(self.n, dtype=torch.int32, device=dev)

    def cigar_strings(self, idx=None):
        cg = self.cigars if idx is None else self.cigars[idx]
        ln = self.cigar_len if idx is None else self.cigar_len[idx]
        cg, ln = cg.cpu().numpy(), ln.cpu().numpy()
        return [cg[k, : ln[k]].tobytes().decode() for k in range(len(ln))]


def run_matrix(batch, aligner, code, matrix, gap_open=11, gap_extend=1, overhang_strategy=1, stream=None, binary_cigar=False,
               grouped=False, score_only=False, shared_target=False):
    """mgl_sw_align_batch_device_matrix on a device_batch.DeviceBatch / IndexedBatch (ASCII wire format); no sync.
    ``grouped``: every aligned block of eight pairs has one (tl, ql) (MGL_SW_FLAG_GROUPED_GEOMETRY) -- true for a
    database search laid out as pair = d * Q + q with Q a multiple of eight -- which makes the packed-int16 kernel
    eligible.  ``shared_target``: every aligned block of 128 pairs shares its target and has one query length
    (MGL_SW_FLAG_SHARED_TARGET, DatabaseSearch below): two pairs per lane, the scores of a column out of a per-strip profile."""
    import torch

    if stream is None:
        stream = torch.cuda.current_stream(batch.targets.device)
    code = np.ascontiguousarray(code, dtype=np.uint8)
    matrix = np.ascontiguousarray(matrix, dtype=np.int8)
    assert code.shape == (256,) and matrix.shape == (32, 32)
    L = _lib.lib()
    rc = L.mgl_sw_align_batch_device_matrix(
        aligner.ctx, C.c_void_p(stream.cuda_stream), batch.n, batch.targets.data_ptr(), batch.t_off.data_ptr(),
        None if getattr(batch, "t_len", None) is None else batch.t_len.data_ptr(), batch.queries.data_ptr(),
        batch.q_off.data_ptr(), None if getattr(batch, "q_len", None) is None else batch.q_len.data_ptr(), batch.max_tl,
        batch.max_ql, matrix.ctypes.data, code.ctypes.data,
        int(gap_open), int(gap_extend), int(overhang_strategy), batch.offsets.data_ptr(), batch.scores.data_ptr(),
        batch.cigars.data_ptr(), batch.cigar_stride, batch.cigar_len.data_ptr(), batch.status.data_ptr(),
        (_lib.FLAG_BINARY_CIGAR if binary_cigar else 0) | (_lib.FLAG_GROUPED_GEOMETRY if grouped else 0) |
        (_lib.FLAG_UNIFORM_GEOMETRY if getattr(batch, "uniform", False) else 0) | (_lib.FLAG_SCORE_ONLY if score_only else 0) |
        (_lib.FLAG_SHARED_TARGET if shared_target else 0))
    if rc != _lib.OK:
        raise _lib.MglSwError(rc, L.mgl_sw_last_error(aligner.ctx).decode())


def shared_target_region_bytes(tl, ql):
    """What one wave slot of the shared-target kernel keeps for a tile of geometry tl x ql (sw_device.h: lane_tb_words * 4 +
    lane_scratch_bytes at 32 rows per strip): the flags of 128 pairs and the carry row."""
    strips = (tl + 31) // 32
    return strips * ql * 2 * 64 * 16 + (ql + 1) * 64 * 8 + ((ql + 3) // 4 + strips * 8) * 2 * 64 * 4


class DatabaseSearch:
    """Q queries of ONE length against D database sequences, laid out for MGL_SW_FLAG_SHARED_TARGET: the first Qs = Q - Q % 128
    queries of every database sequence are tiles of 128 pairs that share it (batch ``shared``, pair = rank(d) * Qs + q, longest
    database sequence first), the other Qr = Q % 128 queries a second batch (``rest``, pair = rank(d) * Qr + (q - Qs); blocks of
    eight of one geometry when Qr % 8 == 0).  The library sizes a wave slot's region by the tile it starts on (it looks at every tile's
    geometry and draws the largest first), so a database's long tail costs what it holds, not the number of slots times the longest
    sequence.  ``shared_max_tl`` / ``workspace_bytes`` (optional; default: neither) send database sequences longer than a bound -- given,
    or the length up to which ``workspace_bytes`` would hold a region of THAT length for each of ``wave_slots`` slots -- with all their
    queries into a third batch (``long``, pair = k * Q + q) for the packed kernel.
    ``db``: uint8 residues of all database sequences, ``db_off`` int64[D + 1]; ``queries``: uint8 [Q, QL].  where(d, q) -> (batch, pair)."""

    def __init__(self, db, db_off, queries, device, cigar_stride=256, workspace_bytes=None, wave_slots=256 * 12, shared_max_tl=None):
        import torch

        db_off = np.asarray(db_off, dtype=np.int64)
        all_lens = np.diff(db_off)
        D, (Q, QL) = len(all_lens), queries.shape
        self.D, self.Q, self.QL = D, Q, QL
        self.Qs, self.Qr = Q - Q % 128, Q % 128
        self.cells = int(all_lens.sum()) * Q * QL
        if shared_max_tl is None:
            shared_max_tl = int(all_lens.max())
            if workspace_bytes:
                while shared_max_tl > 32 and shared_target_region_bytes(shared_max_tl, QL) * wave_slots > workspace_bytes:
                    shared_max_tl -= 32
        self.shared_max_tl = shared_max_tl
        # the longest database sequences FIRST: the kernel is a persistent grid that draws tile after tile, and a tile's work is its
        # target's length -- 2 000-residue tiles at the end of the queue would leave most of the chip waiting for the last few waves
        order = np.argsort(-all_lens, kind="stable")
        n_long = int((all_lens > shared_max_tl).sum())
        self.kind = np.zeros(D, np.int8)            # 1: its pairs are in `long`
        self.kind[order[:n_long]] = 1
        self.rank = np.empty(D, np.int64)           # position among the sequences of its kind
        self.rank[order[:n_long]] = np.arange(n_long)
        self.rank[order[n_long:]] = np.arange(D - n_long)
        t = torch.from_numpy(np.ascontiguousarray(db)).to(device)
        qd = torch.from_numpy(np.ascontiguousarray(queries).reshape(-1)).to(device)

        def part(ds_, q0, qn):
            if qn == 0 or len(ds_) == 0:
                return None
            starts, lens = db_off[:-1][ds_], all_lens[ds_]
            t_start = torch.from_numpy(np.repeat(starts, qn)).to(device)
            t_len = torch.from_numpy(np.repeat(lens, qn).astype(np.int32)).to(device)
            q_start = torch.from_numpy(np.tile((q0 + np.arange(qn, dtype=np.int64)) * QL, len(ds_))).to(device)
            q_len = torch.full((len(ds_) * qn,), QL, dtype=torch.int32, device=device)
            return IndexedBatch(t, t_start, t_len, qd, q_start, q_len, int(lens.max()), QL, cigar_stride)

        self.shared, self.rest = part(order[n_long:], 0, self.Qs), part(order[n_long:], self.Qs, self.Qr)
        self.long = part(order[:n_long], 0, Q)

    def run(self, aligner, code, matrix, gap_open=11, gap_extend=1, overhang_strategy=1, stream=None, score_only=False):
        """Enqueue the search on ``stream`` (default: torch's current stream); no sync.  ``score_only``: MGL_SW_FLAG_SCORE_ONLY, the
        pre-filter mode (all six ScoreMax fields, no CIGAR).  (The two small batches on a second context and stream, to fill the tail
        of the tiles' persistent grid, measured slower: 63.5 against 58.2 ms per pass.)"""
        kw = dict(stream=stream, score_only=score_only)
        if self.shared is not None:
            run_matrix(self.shared, aligner, code, matrix, gap_open, gap_extend, overhang_strategy, shared_target=True, **kw)
        if self.long is not None:
            run_matrix(self.long, aligner, code, matrix, gap_open, gap_extend, overhang_strategy, grouped=self.Q % 8 == 0, **kw)
        if self.rest is not None:
            run_matrix(self.rest, aligner, code, matrix, gap_open, gap_extend, overhang_strategy, grouped=self.Qr % 8 == 0, **kw)

    def batches(self):
        return [b for b in (self.long, self.shared, self.rest) if b is not None]

    def where(self, d, q):
        r = int(self.rank[d])
        if self.kind[d]:
            return self.long, r * self.Q + q
        return (self.shared, r * self.Qs + q) if q < self.Qs else (self.rest, r * self.Qr + q - self.Qs)
