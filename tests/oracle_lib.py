"""ctypes bindings for the two CPU checkers (test infrastructure only).

* ``oracle/libsw_oracle.so``   -- this repo's C restatement (oracle/sw_oracle.c)
* ``oracle/_ref/libmgl_ref.so`` -- the reference's own sw.cpp / sw_avx.cpp compiled
  in place (only buildable where /root/reference exists; the prebuilt .so
  travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.
"""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libsw_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmgl_ref.so")

SOFTCLIP, INDEL, LEAD_INDEL, IGNORE = 1, 2, 4, 8
STRATEGIES = (SOFTCLIP, INDEL, LEAD_INDEL, IGNORE)


class Score(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mqe", "mqe_t", "max", "max_t", "max_q", "seg_length")]

    def astuple(self):
        return (self.mqe, self.mqe_t, self.max, self.max_t, self.max_q, self.seg_length)


def build_oracle():
    """(Re)build the C restatement if it is stale or missing; gcc only."""
    src = os.path.join(ORACLE_DIR, "sw_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        u8p, i32p, i64p = C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.swo_fill.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 5 + [i32p, C.POINTER(Score), i32p]
        lib.swo_cigar.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.POINTER(Score), C.c_char_p, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.swo_align.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 5 + [
            C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Score), i32p]
        lib.swo_btr_crc32.argtypes = [i32p, C.c_int, C.c_int]
        lib.swo_btr_crc32.restype = C.c_uint32
        lib.swo_align_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.swo_normalize_params.argtypes = [C.POINTER(C.c_int)] * 4
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        sig = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 5 + [
            C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.ref_align_scalar.argtypes = sig
        lib.ref_align_avx.argtypes = sig
        i32p = C.POINTER(C.c_int32)
        lib.ref_calculate_matrix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int] + [C.c_int] * 5 + [i32p, i32p]
        lib.ref_calculate_matrix.restype = None
        lib.ref_calculate_cigar.argtypes = [i32p, C.c_int, C.c_int, C.c_int, i32p, C.c_char_p, C.c_int,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if hasattr(lib, "ref_band_fill"):
            lib.ref_band_fill.argtypes = [i32p, C.c_int, i32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p] + [C.c_int] * 5 + [i32p]
            lib.ref_band_fill.restype = None
        lib.ref_align_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [
            C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _ref = lib
    return _ref


def _cap(tl, ql):
    return 12 * (tl + ql + 4)


def oracle_align(t: bytes, q: bytes, params, strategy, want_btr=False):
    """Full result from the C restatement: dict(offset, cigar, score, h_end, crc[, btr])."""
    lib = oracle()
    tl, ql = len(t), len(q)
    m, x, o, e = params
    btr = np.zeros((tl + 1) * (ql + 1), dtype=np.int32)
    bp = btr.ctypes.data_as(C.POINTER(C.c_int32))
    ez = Score()
    h_end = C.c_int32()
    rc = lib.swo_fill(t, tl, q, ql, m, x, o, e, strategy, bp, C.byref(ez), C.byref(h_end))
    assert rc == 0, rc
    cap = _cap(tl, ql)
    buf = C.create_string_buffer(cap)
    ln, off = C.c_int(), C.c_int()
    rc = lib.swo_cigar(bp, tl, ql, strategy, C.byref(ez), buf, cap, C.byref(ln), C.byref(off))
    assert rc == 0, rc
    out = dict(offset=off.value, cigar=buf.raw[: ln.value].decode(), score=ez.astuple(), h_end=h_end.value,
               crc=int(lib.swo_btr_crc32(bp, tl, ql)))
    if want_btr:
        out["btr"] = btr.reshape(tl + 1, ql + 1)
    return out


def ref_align(t: bytes, q: bytes, params, strategy, avx=False):
    """(offset, cigar) from the compiled reference (align_scalar or align_avx)."""
    lib = ref()
    tl, ql = len(t), len(q)
    cap = _cap(tl, ql)
    buf = C.create_string_buffer(cap)
    ln, off = C.c_int(), C.c_int()
    fn = lib.ref_align_avx if avx else lib.ref_align_scalar
    rc = fn(t, tl, q, ql, *params, strategy, buf, cap, C.byref(ln), C.byref(off))
    assert rc == 0
    return off.value, buf.raw[: ln.value].decode()


def ref_full(t: bytes, q: bytes, params, strategy, want_btr=False):
    """dict(offset, cigar, score, crc[, btr]) from the reference's calculateMatrix + calculateCigar."""
    lib = ref()
    tl, ql = len(t), len(q)
    btr = np.zeros((tl + 1) * (ql + 1), dtype=np.int32)
    bp = btr.ctypes.data_as(C.POINTER(C.c_int32))
    ez = (C.c_int32 * 6)()
    lib.ref_calculate_matrix(t, tl, q, ql, *params, strategy, bp, ez)
    cap = _cap(tl, ql)
    buf = C.create_string_buffer(cap)
    ln, off = C.c_int(), C.c_int()
    rc = lib.ref_calculate_cigar(bp, tl, ql, strategy, ez, buf, cap, C.byref(ln), C.byref(off))
    assert rc == 0
    out = dict(offset=off.value, cigar=buf.raw[: ln.value].decode(), score=tuple(ez), crc=btr_crc(btr, tl, ql))
    if want_btr:
        out["btr"] = btr.reshape(tl + 1, ql + 1)
    return out


def btr_crc(btr, tl, ql):
    """zlib CRC-32 of the logical backtrack matrix over i=1..tl, j=1..ql (row-major, LE int32)."""
    a = np.asarray(btr, dtype="<i4").reshape(tl + 1, ql + 1)[1:, 1:]
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def concat(seqs):
    """list of bytes -> (uint8 array, int64 offsets[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return data, off


def oracle_align_batch(ts, qs, params, strategy, nthreads=1, cigar_stride=None):
    """Batch through the C restatement; returns (offsets, scores[n,6], cigars list)."""
    lib = oracle()
    n = len(ts)
    td, toff = concat(ts)
    qd, qoff = concat(qs)
    if cigar_stride is None:
        cigar_stride = max(16, max((_cap(len(a), len(b)) for a, b in zip(ts, qs)), default=16))
    off = np.zeros(n, np.int32)
    sc = np.zeros((n, 6), np.int32)
    cg = np.zeros(n * cigar_stride, np.uint8)
    ln = np.zeros(n, np.int32)
    rc = lib.swo_align_batch(n, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, *params, strategy,
                             nthreads, off.ctypes.data, sc.ctypes.data, cg.ctypes.data, cigar_stride, ln.ctypes.data)
    assert rc == 0, rc
    cigars = [cg[i * cigar_stride: i * cigar_stride + ln[i]].tobytes().decode() for i in range(n)]
    return off, sc, cigars
