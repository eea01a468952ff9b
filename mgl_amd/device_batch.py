"""Device-resident batches: torch is used only for HBM allocations, streams and (in dist.py)
the RCCL gather; the alignment itself is mgl_sw_align_batch_device (include/mgl_sw.h)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .smithwaterman import GATK_PARAMETERS, SWOverhangStrategy, SWParameters, _check

_BASES = b"ACGT"


class DeviceBatch:
    """Inputs and outputs of one batch, all resident in one GPU's HBM.

    targets / queries: uint8 concatenated bases; t_off / q_off: int64 [n+1].
    Outputs: offsets int32[n], scores int32[n,6] (ScoreMax), cigars uint8[n,stride] zero padded,
    cigar_len int32[n], status int32[n].
    """

    def __init__(self, targets, t_off, queries, q_off, max_tl, max_ql, cigar_stride=64, uniform=False):
        assert targets.dtype == torch.uint8 and t_off.dtype == torch.int64
        self.targets, self.t_off, self.queries, self.q_off = targets, t_off, queries, q_off
        self.n = t_off.numel() - 1
        self.max_tl, self.max_ql = int(max_tl), int(max_ql)
        self.cigar_stride = int(cigar_stride)
        self.uniform = bool(uniform)  # every pair exactly max_tl x max_ql
        dev = targets.device
        self.offsets = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.scores = torch.empty((self.n, 6), dtype=torch.int32, device=dev)
        self.cigars = torch.empty((self.n, self.cigar_stride), dtype=torch.uint8, device=dev)
        self.cigar_len = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.status = torch.empty(self.n, dtype=torch.int32, device=dev)

    @property
    def cells(self):
        tl = (self.t_off[1:] - self.t_off[:-1])
        ql = (self.q_off[1:] - self.q_off[:-1])
        return int((tl * ql).sum().item())

    def run(self, aligner, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP, stream=None,
            binary_cigar=False, score_only=False):
        """Enqueue fill + traceback on ``stream`` (default: torch's current stream); no sync.  ``score_only``:
        MGL_SW_FLAG_SCORE_ONLY (only ``scores`` is wanted; batches on the packed kernel skip the traceback)."""
        assert self.targets.is_cuda, "the batch must be resident on a GPU (there is no CPU path)"
        if stream is None:
            stream = torch.cuda.current_stream(self.targets.device)
        p = SWParameters(*parameters)
        flags = (_lib.FLAG_UNIFORM_GEOMETRY if self.uniform else 0) | (_lib.FLAG_BINARY_CIGAR if binary_cigar else 0) | (
            _lib.FLAG_SCORE_ONLY if score_only else 0)
        rc = _lib.lib().mgl_sw_align_batch_device(
            aligner.ctx, C.c_void_p(stream.cuda_stream), self.n, self.targets.data_ptr(), self.t_off.data_ptr(),
            self.queries.data_ptr(), self.q_off.data_ptr(), self.max_tl, self.max_ql, p.match, p.mismatch,
            p.gap_open, p.gap_extend, int(overhang_strategy), self.offsets.data_ptr(), self.scores.data_ptr(),
            self.cigars.data_ptr(), self.cigar_stride, self.cigar_len.data_ptr(), self.status.data_ptr(), flags)
        _check(rc, aligner.ctx)

    def cigar_elements(self, idx=None):
        """Decode BAM-style binary CIGARs (run(..., binary_cigar=True)) into text."""
        cg = self.cigars if idx is None else self.cigars[idx]
        ln = self.cigar_len if idx is None else self.cigar_len[idx]
        cg, ln = cg.cpu().numpy(), ln.cpu().numpy()
        out = []
        for k in range(len(ln)):
            el = np.frombuffer(cg[k, : ln[k]].tobytes(), dtype="<u4")
            out.append("".join(f"{int(v) >> 4}{'MIDNS'[int(v) & 15]}" for v in el))
        return out

    def cigar_strings(self, idx=None):
        cg = self.cigars if idx is None else self.cigars[idx]
        ln = self.cigar_len if idx is None else self.cigar_len[idx]
        cg, ln = cg.cpu().numpy(), ln.cpu().numpy()
        return [cg[k, : ln[k]].tobytes().decode() for k in range(len(ln))]

    def host_pairs(self, idx):
        """(targets, queries) as lists of bytes for the pairs ``idx`` (1-D LongTensor / list)."""
        idx = torch.as_tensor(idx, device=self.targets.device, dtype=torch.int64)
        t0, t1 = self.t_off[idx].cpu().numpy(), self.t_off[idx + 1].cpu().numpy()
        q0, q1 = self.q_off[idx].cpu().numpy(), self.q_off[idx + 1].cpu().numpy()
        lo_t, hi_t, lo_q, hi_q = int(t0.min()), int(t1.max()), int(q0.min()), int(q1.max())
        T = self.targets[lo_t:hi_t].cpu().numpy()
        Q = self.queries[lo_q:hi_q].cpu().numpy()
        ts = [T[a - lo_t: b - lo_t].tobytes() for a, b in zip(t0, t1)]
        qs = [Q[a - lo_q: b - lo_q].tobytes() for a, b in zip(q0, q1)]
        return ts, qs


WORKLOAD_BLOCK = 1 << 16  # pairs per generation block of the window workload (small batches materialise one block, not a million pairs)


def window_batch(seed, n_pairs, device, window=256, read_len=150, genome_len=1 << 24, sub=0.01, ins=0.001,
                 dele=0.001, cigar_stride=64, first=0):
    """BASELINE.json configs[1] generated on the GPU: per-pair ``window``-base target cut from a
    seeded random genome, ``read_len``-bp read copied from inside the window with Illumina-style
    errors (the model of synth.illumina_reads).  Deterministic in (seed, arguments, device type).

    The workload is ONE seeded sequence of pairs, defined block by block (WORKLOAD_BLOCK pairs, each block from its
    own generator seeded by (seed, block index)); this call materialises pairs [first, first + n_pairs) of it, so
    the ranks of a multi-GPU run each generate only their own contiguous shard of the same global batch
     and a one-rank run of the whole range reproduces their concatenation."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    bases = torch.tensor(list(_BASES), dtype=torch.uint8, device=device)
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)
    ar_w = torch.arange(window, device=device, dtype=torch.int64)
    targets = torch.empty((n_pairs, window), dtype=torch.uint8, device=device)
    reads = torch.empty((n_pairs, read_len), dtype=torch.uint8, device=device)
    win_all = torch.empty(n_pairs, dtype=torch.int64, device=device)
    first, last = int(first), int(first) + int(n_pairs)
    for blk in range(first // WORKLOAD_BLOCK, (last + WORKLOAD_BLOCK - 1) // WORKLOAD_BLOCK if n_pairs else 0):
        m = WORKLOAD_BLOCK
        gb = torch.Generator(device=device)
        gb.manual_seed((int(seed) * 1_000_003 + blk + 1) & 0x7FFFFFFFFFFFFFFF)
        win = torch.randint(0, genome_len - window, (m,), generator=gb, device=device, dtype=torch.int64)
        inner = torch.randint(0, window - read_len - 8 + 1, (m,), generator=gb, device=device, dtype=torch.int64)
        r = torch.rand((3, m, read_len), generator=gb, device=device)
        is_ins, is_del, is_sub = r[0] < ins, r[1] < dele, r[2] < sub
        copied = ~is_ins
        src = torch.cumsum(copied, 1) - copied.long() + torch.cumsum(is_del & copied, 1)
        src = (src + (win + inner)[:, None]).clamp_(0, genome_len - 1)
        code = genome[src]
        shift = torch.randint(1, 4, (m, read_len), generator=gb, device=device, dtype=torch.uint8)
        code = torch.where(is_sub, (code + shift) & 3, code)
        rnd = torch.randint(0, 4, (m, read_len), generator=gb, device=device, dtype=torch.uint8)
        code = torch.where(is_ins, rnd, code)
        # the part of this block inside [first, last)
        lo, hi = max(first, blk * WORKLOAD_BLOCK), min(last, (blk + 1) * WORKLOAD_BLOCK)
        a, b = lo - blk * WORKLOAD_BLOCK, hi - blk * WORKLOAD_BLOCK
        reads[lo - first:hi - first] = bases[code[a:b].long()]
        targets[lo - first:hi - first] = bases[genome[win[a:b, None] + ar_w].long()]
        win_all[lo - first:hi - first] = win[a:b]
        del r, is_ins, is_del, is_sub, copied, src, code, shift, rnd, win, inner
    t_off = torch.arange(n_pairs + 1, device=device, dtype=torch.int64) * window
    q_off = torch.arange(n_pairs + 1, device=device, dtype=torch.int64) * read_len
    out = DeviceBatch(targets.reshape(-1), t_off, reads.reshape(-1), q_off, window, read_len, cigar_stride,
                      uniform=True)
    out.win = win_all  # window start of every pair in the genome (window_batch_2bit addresses the packed genome by it)
    return out


def from_host(targets, t_off, queries, q_off, device, cigar_stride=None):
    """Upload a host batch (numpy uint8 / int64 arrays as taken by mgl_sw_align_batch)."""
    t_off = np.asarray(t_off, dtype=np.int64)
    q_off = np.asarray(q_off, dtype=np.int64)
    dt, dq = np.diff(t_off), np.diff(q_off)
    max_tl, max_ql = int(dt.max()), int(dq.max())
    uniform = bool((dt == max_tl).all() and (dq == max_ql).all())
    if cigar_stride is None:
        cigar_stride = max(16, 2 * max(max_tl, max_ql))
    dev = torch.device(device)
    return DeviceBatch(torch.from_numpy(np.ascontiguousarray(targets, dtype=np.uint8)).to(dev),
                       torch.from_numpy(t_off).to(dev),
                       torch.from_numpy(np.ascontiguousarray(queries, dtype=np.uint8)).to(dev),
                       torch.from_numpy(q_off).to(dev), max_tl, max_ql, cigar_stride, uniform=uniform)


# ---------------------------------------------------------------------------------------------
# variable-length batches at packed-kernel speed: sort by geometry, pad every geometry to a multiple of eight
class GroupedBatch(DeviceBatch):
    """ASCII pairs addressed by (start, length) (mgl_sw_align_batch_device_indexed), reordered for the packed kernel:
    pairs are sorted by (tl, ql); every geometry with at least ``min_bucket`` pairs is padded to a multiple of eight
    (its last pair repeated) and goes into the GROUPED part, where every aligned block of eight slots has one geometry
    (MGL_SW_FLAG_GROUPED_GEOMETRY); the pairs of rarer geometries follow as an ordinary mixed part (int32 kernel),
    still sorted so that wave mates are alike.  ``order[k]`` is the original pair index of slot k;
    ``first_slot[i]`` is a slot holding original pair i: its results are ``offsets[first_slot[i]]`` etc.
    (``gather()`` returns them in the original order)."""

    def __init__(self, targets, t_start, t_len, queries, q_start, q_len, cigar_stride=64, min_bucket=8):
        dev = targets.device
        n = t_start.numel()
        tl, ql = t_len.to(torch.int64), q_len.to(torch.int64)
        key = tl * (1 << 32) + ql
        order = torch.argsort(key, stable=True)
        uniq, inverse, counts = torch.unique_consecutive(key[order], return_inverse=True, return_counts=True)
        big = counts >= min_bucket
        in_big = big[inverse]                                  # per sorted pair
        # ---- grouped part: the big buckets, each padded to a multiple of eight
        cb = counts[big]
        padded = (cb + 7) // 8 * 8
        n_grouped = int(padded.sum())
        sorted_big = order[in_big]                             # pairs of big buckets, bucket by bucket
        if n_grouped:
            ends = torch.cumsum(cb, 0)
            bucket = torch.repeat_interleave(torch.arange(len(cb), device=dev), padded)
            pos = torch.arange(n_grouped, device=dev) - torch.repeat_interleave(torch.cumsum(padded, 0) - padded, padded)
            src = (ends - cb)[bucket] + torch.minimum(pos, cb[bucket] - 1)
            grouped = sorted_big[src]
        else:
            grouped = order[:0]
        rest = order[~in_big]
        self.n_grouped, self.n_rest = n_grouped, int(rest.numel())
        self.order = torch.cat([grouped, rest])
        self.n = self.n_grouped + self.n_rest
        self.first_slot = torch.full((n,), self.n, dtype=torch.int64, device=dev).scatter_reduce_(
            0, self.order, torch.arange(self.n, device=dev), "amin")
        self.targets, self.queries = targets, queries
        self.t_off, self.q_off = t_start[self.order].contiguous(), q_start[self.order].contiguous()
        self.t_len, self.q_len = t_len[self.order].to(torch.int32).contiguous(), q_len[self.order].to(torch.int32).contiguous()
        self.max_tl, self.max_ql = int(tl.max()), int(ql.max())
        self.cigar_stride = int(cigar_stride)
        self.uniform = False
        self.offsets = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.scores = torch.empty((self.n, 6), dtype=torch.int32, device=dev)
        self.cigars = torch.empty((self.n, self.cigar_stride), dtype=torch.uint8, device=dev)
        self.cigar_len = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.status = torch.empty(self.n, dtype=torch.int32, device=dev)

    @property
    def cells(self):
        return int((self.t_len.to(torch.int64) * self.q_len.to(torch.int64)).sum())

    def _launch(self, aligner, stream, lo, count, p, overhang_strategy, flags):
        if count == 0:
            return
        sl = slice(lo, lo + count)
        rc = _lib.lib().mgl_sw_align_batch_device_indexed(
            aligner.ctx, C.c_void_p(stream.cuda_stream), count, self.targets.data_ptr(), self.t_off[sl].data_ptr(),
            self.t_len[sl].data_ptr(), self.queries.data_ptr(), self.q_off[sl].data_ptr(), self.q_len[sl].data_ptr(),
            int(self.t_len[sl].max()), int(self.q_len[sl].max()), p.match, p.mismatch, p.gap_open, p.gap_extend,
            int(overhang_strategy), self.offsets[sl].data_ptr(), self.scores[sl].data_ptr(), self.cigars[sl].data_ptr(),
            self.cigar_stride, self.cigar_len[sl].data_ptr(), self.status[sl].data_ptr(), flags)
        _check(rc, aligner.ctx)

    def run(self, aligner, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP, stream=None,
            binary_cigar=False):
        """Two enqueues on ``stream``: the grouped part (packed kernel when its score range allows), then the rest."""
        if stream is None:
            stream = torch.cuda.current_stream(self.targets.device)
        p = SWParameters(*parameters)
        bc = _lib.FLAG_BINARY_CIGAR if binary_cigar else 0
        self._launch(aligner, stream, 0, self.n_grouped, p, overhang_strategy, _lib.FLAG_GROUPED_GEOMETRY | bc)
        self.grouped_packed16 = bool(aligner.timing().packed16) if (self.n_grouped and not self.n_rest) else None
        self._launch(aligner, stream, self.n_grouped, self.n_rest, p, overhang_strategy, bc)

    def gather(self):
        """(offsets, scores, cigars, cigar_len, status) in the ORIGINAL pair order (device tensors)."""
        s = self.first_slot
        return self.offsets[s], self.scores[s], self.cigars[s], self.cigar_len[s], self.status[s]


# ---------------------------------------------------------------------------------------------
# 2-bit packed inputs (mgl_sw_align_batch_device_2bit)

_CODE = np.full(256, 255, dtype=np.uint8)
for _k, _c in enumerate(b"ACGT"):
    _CODE[_c] = _k


def pack2bit(seq_bytes, lead=0):
    """ACGT bytes (numpy uint8 / bytes) -> 2-bit packed uint8 array, four bases per byte, base k in bits
    2*(k%4) of byte k//4.  ``lead`` unused base slots (zeros) are put in front, so the first real base gets
    base index ``lead``."""
    a = np.frombuffer(bytes(seq_bytes), dtype=np.uint8) if not isinstance(seq_bytes, np.ndarray) else seq_bytes
    code = _CODE[a]
    if (code == 255).any():
        raise ValueError("2-bit packing needs ACGT only")
    code = np.concatenate([np.zeros(lead, np.uint8), code])
    pad = (-len(code)) % 4
    code = np.concatenate([code, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    return (code[:, 0] | (code[:, 1] << 2) | (code[:, 2] << 4) | (code[:, 3] << 6)).astype(np.uint8)


class PackedBatch(DeviceBatch):
    """A batch whose sequences are 2-bit packed: pair k = t_len[k] bases from base index t_start[k] of
    ``target_bases`` against q_len[k] bases from q_start[k] of ``query_bases`` (t_len / q_len None: uniform)."""

    def __init__(self, target_bases, t_start, t_len, query_bases, q_start, q_len, max_tl, max_ql, cigar_stride=64):
        self.target_bases, self.query_bases = target_bases, query_bases
        self.t_start, self.q_start, self.t_len, self.q_len = t_start, q_start, t_len, q_len
        self.n = t_start.numel()
        self.max_tl, self.max_ql = int(max_tl), int(max_ql)
        self.cigar_stride = int(cigar_stride)
        self.uniform = t_len is None and q_len is None
        dev = target_bases.device
        self.offsets = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.scores = torch.empty((self.n, 6), dtype=torch.int32, device=dev)
        self.cigars = torch.empty((self.n, self.cigar_stride), dtype=torch.uint8, device=dev)
        self.cigar_len = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.status = torch.empty(self.n, dtype=torch.int32, device=dev)

    def run(self, aligner, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream(self.target_bases.device)
        p = SWParameters(*parameters)
        rc = _lib.lib().mgl_sw_align_batch_device_2bit(
            aligner.ctx, C.c_void_p(stream.cuda_stream), self.n, self.target_bases.data_ptr(), self.t_start.data_ptr(),
            None if self.t_len is None else self.t_len.data_ptr(), self.query_bases.data_ptr(), self.q_start.data_ptr(),
            None if self.q_len is None else self.q_len.data_ptr(), self.max_tl, self.max_ql, p.match, p.mismatch,
            p.gap_open, p.gap_extend, int(overhang_strategy), self.offsets.data_ptr(), self.scores.data_ptr(),
            self.cigars.data_ptr(), self.cigar_stride, self.cigar_len.data_ptr(), self.status.data_ptr(),
            _lib.FLAG_UNIFORM_GEOMETRY if self.uniform else 0)
        _check(rc, aligner.ctx)


def _pack2bit_torch(code):
    """uint8 codes 0..3 (length multiple of 4) -> packed bytes, on the tensor's device."""
    c = code.view(-1, 4)
    return c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)


def window_batch_2bit(seed, n_pairs, device, window=256, read_len=150, genome_len=1 << 24, cigar_stride=64, **kw):
    """The window_batch workload in its packed form (SURVEY.md 8d "Config 2"): ONE 2-bit packed genome, target
    windows addressed by base offset into it, reads packed back to back.  Same bases as window_batch(seed, ...)."""
    b = window_batch(seed, n_pairs, device, window, read_len, genome_len, cigar_stride=cigar_stride, **kw)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)
    win = b.win
    lut = torch.zeros(256, dtype=torch.uint8, device=device)
    for k, ch in enumerate(_BASES):
        lut[ch] = k
    reads_code = lut[b.queries.long()]
    pad = (-reads_code.numel()) % 4
    if pad:
        reads_code = torch.cat([reads_code, torch.zeros(pad, dtype=torch.uint8, device=device)])
    q_start = torch.arange(n_pairs, device=device, dtype=torch.int64) * read_len
    return PackedBatch(_pack2bit_torch(genome), win, None, _pack2bit_torch(reads_code), q_start, None, window, read_len,
                       cigar_stride), b
