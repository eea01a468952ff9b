"""Host-side mirror of the reference's Smith-Waterman operator interface, over the C ABI.

The reference's entry point is ``MicrosoftSmithWaterman.align(ref, alt, parameters,
overhangStrategy) -> SWNativeAlignerResult(cigar, alignment_offset)``
(src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86) on top of
``align_avx`` / ``align_scalar`` (src/main/native/mgl_sw/sw_avx.h:6, sw_scalar.h:9).  The
names, argument meaning and the (cigar, offset) result are kept; a batch form is added because
one pair per launch cannot feed a GPU.  All arithmetic happens in libmgl_sw_hip.so.
"""
import ctypes as C
from collections import namedtuple
from enum import IntEnum

import numpy as np

from . import _lib


class SWOverhangStrategy(IntEnum):
    """org.broadinstitute.gatk.nativebindings SWOverhangStrategy -> sw_common.h:22-25 codes
    (the mapping of MicrosoftSmithWaterman.java:39-56)."""
    SOFTCLIP = 0x01
    INDEL = 0x02
    LEADING_INDEL = 0x04
    IGNORE = 0x08


SWParameters = namedtuple("SWParameters", "match mismatch gap_open gap_extend")
SWParameters.__doc__ = "swParameters (sw_common.h:42-47); any sign convention, normalised natively"

# GATK's NEW_SW_PARAMETERS, the set the reference is exercised with
GATK_PARAMETERS = SWParameters(200, -150, -260, -11)

SWNativeAlignerResult = namedtuple("SWNativeAlignerResult", "cigar alignment_offset")
ScoreMax = namedtuple("ScoreMax", "mqe mqe_t max max_t max_q seg_length")

BatchResult = namedtuple("BatchResult", "offsets scores cigars cigar_len")


class CigarColumn:
    """The CIGAR texts of a batch, decoded on access (a 10 M-pair batch would otherwise spend ten times the
    alignment time building Python strings).  Behaves like a read-only list of str."""

    def __init__(self, slots, lengths):
        self.slots, self.lengths = slots, lengths      # uint8 [n, stride], int32 [n]

    def __len__(self):
        return len(self.lengths)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        return self.slots[k, : self.lengths[k]].tobytes().decode()

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __repr__(self):
        return "CigarColumn(%r)" % (self[:8] + (["..."] if len(self) > 8 else []),)


def _check(rc, ctx=None):
    if rc != _lib.OK:
        detail = _lib.lib().mgl_sw_last_error(ctx).decode() if ctx else ""
        raise _lib.MglSwError(rc, detail)


def concat(seqs):
    """list of bytes -> (uint8 array, int64 offsets[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    if seqs:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return data, off


class MicrosoftSmithWaterman:
    """Drop-in for the reference's SWAlignerNativeBinding implementation.

    ``load()`` / ``align()`` / ``close()`` follow MicrosoftSmithWaterman.java:35,66,89.
    One instance owns one GPU context; instances may be used from different threads.
    """

    def __init__(self, device=0):
        self._device = device
        self._ctx = None

    # -- SWAlignerNativeBinding ------------------------------------------------------------
    def load(self, temp_dir=None):
        """True when the native library is present and a GPU context could be made
        (the reference returns False when the library cannot be used, .java:27-37)."""
        try:
            self._ensure()
            return True
        except (OSError, _lib.MglSwError):
            return False

    def align(self, ref, alt, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP):
        ref, alt = bytes(ref), bytes(alt)
        # the Java side allocates 2*max(refLength, altLength) CIGAR bytes (.java:71)
        cap = max(16, 2 * max(len(ref), len(alt)))
        res = self.align_batch([ref], [alt], parameters, overhang_strategy, cigar_stride=cap)
        return SWNativeAlignerResult(res.cigars[0], int(res.offsets[0]))

    def close(self):
        if self._ctx is not None:
            _lib.lib().mgl_sw_ctx_destroy(self._ctx)
            self._ctx = None

    # -- batch form ------------------------------------------------------------------------
    def align_batch(self, refs, alts, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP,
                    cigar_stride=None):
        """Align refs[k] against alts[k] for every k; returns BatchResult with numpy arrays."""
        td, toff = concat([bytes(x) for x in refs])
        qd, qoff = concat([bytes(x) for x in alts])
        return self.align_packed(td, toff, qd, qoff, parameters, overhang_strategy, cigar_stride)

    def align_packed(self, targets, t_off, queries, q_off, parameters=GATK_PARAMETERS,
                     overhang_strategy=SWOverhangStrategy.SOFTCLIP, cigar_stride=None):
        """Concatenated-bytes form of align_batch (what mgl_sw_align_batch takes)."""
        ctx = self._ensure()
        n = len(t_off) - 1
        assert len(q_off) - 1 == n
        targets = np.ascontiguousarray(targets, dtype=np.uint8)
        queries = np.ascontiguousarray(queries, dtype=np.uint8)
        t_off = np.ascontiguousarray(t_off, dtype=np.int64)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        if cigar_stride is None:
            longest = int(max(np.diff(t_off).max(initial=1), np.diff(q_off).max(initial=1)))
            cigar_stride = max(16, 2 * longest)
        off = np.empty(n, np.int32)
        sc = np.empty((n, 6), np.int32)
        cg = np.empty(n * cigar_stride, np.uint8)
        ln = np.empty(n, np.int32)
        p = SWParameters(*parameters)
        rc = _lib.lib().mgl_sw_align_batch(ctx, n, targets.ctypes.data, t_off.ctypes.data, queries.ctypes.data,
                                           q_off.ctypes.data, p.match, p.mismatch, p.gap_open, p.gap_extend,
                                           int(overhang_strategy), off.ctypes.data, sc.ctypes.data, cg.ctypes.data,
                                           cigar_stride, ln.ctypes.data)
        _check(rc, ctx)
        return BatchResult(off, sc, CigarColumn(cg.reshape(n, cigar_stride), ln), ln)

    def align_packed_2bit(self, target_bases, target_base_count, t_start, t_len, query_bases, query_base_count, q_start, q_len,
                          max_tl, max_ql, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP, cigar_stride=None,
                          out=None):
        """mgl_sw_align_batch_2bit: 2-bit packed bases in host memory (four per byte), pair k = t_len[k] bases from base index
        t_start[k] against q_len[k] from q_start[k]; t_len = q_len = None: every pair max_tl x max_ql.  ``out``: (offsets,
        scores, cigars, lengths) arrays to write into (e.g. registered ones) instead of fresh ones."""
        ctx = self._ensure()
        n = len(t_start)
        uniform = t_len is None and q_len is None
        t_start = np.ascontiguousarray(t_start, dtype=np.int64)
        q_start = np.ascontiguousarray(q_start, dtype=np.int64)
        if not uniform:
            t_len = np.ascontiguousarray(t_len, dtype=np.int32)
            q_len = np.ascontiguousarray(q_len, dtype=np.int32)
        if cigar_stride is None:
            cigar_stride = max(16, 2 * max(max_tl, max_ql))
        if out is None:
            out = (np.empty(n, np.int32), np.empty((n, 6), np.int32), np.empty(n * cigar_stride, np.uint8), np.empty(n, np.int32))
        off, sc, cg, ln = out
        p = SWParameters(*parameters)
        rc = _lib.lib().mgl_sw_align_batch_2bit(
            ctx, n, target_bases.ctypes.data, int(target_base_count), t_start.ctypes.data, None if uniform else t_len.ctypes.data,
            query_bases.ctypes.data, int(query_base_count), q_start.ctypes.data, None if uniform else q_len.ctypes.data, int(max_tl),
            int(max_ql), p.match, p.mismatch, p.gap_open, p.gap_extend, int(overhang_strategy), off.ctypes.data, sc.ctypes.data,
            cg.ctypes.data, cigar_stride, ln.ctypes.data, None, _lib.FLAG_UNIFORM_GEOMETRY if uniform else 0)
        _check(rc, ctx)
        return BatchResult(off, sc, CigarColumn(cg.reshape(n, cigar_stride), ln), ln)

    def register_host_buffer(self, array):
        """Page-lock a numpy array for this context (mgl_sw_register_host_buffer): the host entries then copy from / into it
        asynchronously.  Keep the array alive and unregister it before it is freed."""
        _check(_lib.lib().mgl_sw_register_host_buffer(self._ensure(), array.ctypes.data, array.nbytes), self._ctx)

    def unregister_host_buffer(self, array):
        _check(_lib.lib().mgl_sw_unregister_host_buffer(self._ensure(), array.ctypes.data), self._ctx)

    def expand_slot(self, slot, tl, ql):
        """Logical backtrack matrix of pair ``slot`` of the last chunk of the last batch call."""
        btr = np.zeros((tl + 1, ql + 1), dtype=np.int32)
        _check(_lib.lib().mgl_sw_ctx_expand_slot(self._ensure(), slot, tl, ql,
                                                 btr.ctypes.data_as(C.POINTER(C.c_int32))), self._ctx)
        return btr

    def slot_layout(self, slot):
        """Traceback layout of pair ``slot`` of the last chunk (0 int32, 1 packed16, 2 lane16, 3 coop16)."""
        v = C.c_int(-1)
        _check(_lib.lib().mgl_sw_ctx_slot_layout(self._ensure(), slot, C.byref(v)), self._ctx)
        return v.value

    # -- extras the reference keeps internal -----------------------------------------------
    def set_workspace(self, nbytes):
        _check(_lib.lib().mgl_sw_ctx_set_workspace(self._ensure(), int(nbytes)))

    def set_precision(self, bits):
        """0 = per batch (packed int16 when possible), 32 = always the int32 fill kernels, 16 = like 0 but the self-checking
        16-bit long-read kernel is tried whenever its constants fit."""
        _check(_lib.lib().mgl_sw_ctx_set_precision(self._ensure(), int(bits)))

    def set_strip_kernel(self, mode):
        """Long reads, one 32-row strip per lane-half: 0 = by size, 1 = never, 2 = whenever eligible."""
        _check(_lib.lib().mgl_sw_ctx_set_strip_kernel(self._ensure(), int(mode)))

    def set_carry_memory(self, mode):
        """0 = stripe carry in LDS when the query fits, 1 = always in the HBM scratch (long-query path)."""
        _check(_lib.lib().mgl_sw_ctx_set_carry_memory(self._ensure(), int(mode)))

    def set_stripe_rows(self, rows):
        """Lanes per pair of the int32 fill kernel: 0 = by query length, 16 or 64 = forced."""
        _check(_lib.lib().mgl_sw_ctx_set_stripe_rows(self._ensure(), int(rows)))

    def set_cooperative(self, mode):
        """Long-read fill kernel (one pair per workgroup): 0 = when the query does not fit the one-wave LDS carve,
        1 = never, 2..16 = always with that many waves per pair."""
        _check(_lib.lib().mgl_sw_ctx_set_cooperative(self._ensure(), int(mode)))

    def set_lane_kernel(self, mode):
        """Packed kernel for uniform batches: 0 = by batch size, 1 = never the two-pairs-per-lane kernel, 2 = always when eligible."""
        _check(_lib.lib().mgl_sw_ctx_set_lane_kernel(self._ensure(), int(mode)))

    def set_lane_checkpoint(self, mode):
        """The lane kernel's checkpointed form (no stored traceback, the walk recomputes the blocks it crosses): 0 = default (on),
        1 = never (needed before expand_slot), 2 = on."""
        _check(_lib.lib().mgl_sw_ctx_set_lane_checkpoint(self._ensure(), int(mode)))

    def set_small_kernel(self, mode):
        """Small batches (up to 5 120 pairs of one promised geometry, 8 192 without a promise -- half where one matrix fills more than half a CU's LDS --: one wave per pair, fill + walk in one launch.
        0 = default (on, unless another kernel choice is forced), 1 = never, 2 = whenever the bounds allow."""
        _check(_lib.lib().mgl_sw_ctx_set_small_kernel(self._ensure(), int(mode)))

    def set_profiling(self, on=True):
        _check(_lib.lib().mgl_sw_ctx_set_profiling(self._ensure(), int(on)))

    def timing(self):
        t = _lib.Timing()
        _check(_lib.lib().mgl_sw_ctx_get_timing(self._ensure(), C.byref(t)))
        return t

    def check(self):
        """mgl_sw_ctx_check: what a kernel found out about itself after the (asynchronous) device entry that enqueued it returned --
        synchronise the stream first.  Raises MglSwError(ERR_DEVICE) when a persistent grid found its tile counter out of range."""
        _check(_lib.lib().mgl_sw_ctx_check(self._ensure()), self._ctx)

    @staticmethod
    def fill_kernel_name(timing):
        """Name of the fill kernel a Timing record belongs to (MGL_SW_KERNEL_*)."""
        return _lib.FILL_KERNEL_NAMES[timing.fill_kernel]

    @property
    def ctx(self):
        return self._ensure()

    def _ensure(self):
        if self._ctx is None:
            h = C.c_void_p()
            _check(_lib.lib().mgl_sw_ctx_create(self._device, C.byref(h)))
            self._ctx = h
        return self._ctx

    def __enter__(self):
        self._ensure()
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGpuSmithWaterman:
    """Several GPUs from one process (mgl_sw_align_batch_multi): contiguous shards of a host batch, one per device,
    balanced by DP cells, each on its own context and host thread; results in the caller's order.
    ``devices``: HIP ordinals (an ordinal may repeat: every entry gets its own context and thread)."""

    def __init__(self, devices):
        self._devices = [int(d) for d in devices]
        arr = (C.c_int * len(self._devices))(*self._devices)
        h = C.c_void_p()
        rc = _lib.lib().mgl_sw_multi_create(len(self._devices), arr, C.byref(h))
        if rc != _lib.OK:
            raise _lib.MglSwError(rc, "mgl_sw_multi_create")
        self._h = h

    def set_workspace(self, nbytes_per_device):
        rc = _lib.lib().mgl_sw_multi_set_workspace(self._h, int(nbytes_per_device))
        if rc != _lib.OK:
            raise _lib.MglSwError(rc, "mgl_sw_multi_set_workspace")

    def align_packed(self, targets, t_off, queries, q_off, parameters=GATK_PARAMETERS,
                     overhang_strategy=SWOverhangStrategy.SOFTCLIP, cigar_stride=64, per_pair_status=False):
        L = _lib.lib()
        n = len(t_off) - 1
        targets = np.ascontiguousarray(targets, dtype=np.uint8)
        queries = np.ascontiguousarray(queries, dtype=np.uint8)
        t_off = np.ascontiguousarray(t_off, dtype=np.int64)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        off = np.empty(n, np.int32)
        sc = np.empty((n, 6), np.int32)
        cg = np.empty(n * cigar_stride, np.uint8)
        ln = np.empty(n, np.int32)
        st = np.zeros(n, np.int32) if per_pair_status else None
        p = SWParameters(*parameters)
        rc = L.mgl_sw_align_batch_multi(self._h, n, targets.ctypes.data, t_off.ctypes.data, queries.ctypes.data, q_off.ctypes.data,
                                        p.match, p.mismatch, p.gap_open, p.gap_extend, int(overhang_strategy), off.ctypes.data,
                                        sc.ctypes.data, cg.ctypes.data, cigar_stride, ln.ctypes.data,
                                        st.ctypes.data if st is not None else None)
        if rc != _lib.OK:
            raise _lib.MglSwError(rc, L.mgl_sw_multi_last_error(self._h).decode())
        res = BatchResult(off, sc, CigarColumn(cg.reshape(n, cigar_stride), ln), ln)
        return (res, st) if per_pair_status else res

    def align_batch(self, refs, alts, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP, cigar_stride=64):
        td, toff = concat([bytes(x) for x in refs])
        qd, qoff = concat([bytes(x) for x in alts])
        return self.align_packed(td, toff, qd, qoff, parameters, overhang_strategy, cigar_stride)

    def last_shards(self):
        first = np.zeros(len(self._devices) + 1, np.int64)
        _lib.lib().mgl_sw_multi_last_shards(self._h, first.ctypes.data)
        return first

    def close(self):
        if self._h is not None:
            _lib.lib().mgl_sw_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align(ref, alt, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP):
    """One pair through mgl_sw_align (thread-local context), returning (cigar, offset, ScoreMax)."""
    ref, alt = bytes(ref), bytes(alt)
    L = _lib.lib()
    cap = 12 * (len(ref) + len(alt) + 4)
    buf = C.create_string_buffer(cap)
    ln, off, ez = C.c_int(), C.c_int(), _lib.Score()
    p = SWParameters(*parameters)
    rc = L.mgl_sw_align(ref, len(ref), alt, len(alt), p.match, p.mismatch, p.gap_open, p.gap_extend,
                        int(overhang_strategy), buf, cap, C.byref(ln), C.byref(off), C.byref(ez))
    _check(rc)
    return buf.raw[: ln.value].decode(), off.value, ScoreMax(ez.mqe, ez.mqe_t, ez.max, ez.max_t, ez.max_q,
                                                            ez.seg_length)


def set_coalescing(max_batch, max_wait_us):
    """The front-ends of the one-pair entry (mgl_sw_align / alignNative): max_batch = 0 makes every call a direct call."""
    _check(_lib.lib().mgl_sw_set_coalescing(int(max_batch), int(max_wait_us)))


def set_service(slots, idle_us=0):
    """Mailboxes of the one-pair entry (one resident wave per calling thread, no launch per call): `slots` of them at most, 0 = off
    (every call through the coalescer); the service grid ends after idle_us of silence (0: keep the current value)."""
    _check(_lib.lib().mgl_sw_set_service(int(slots), int(idle_us)))


def service_stats():
    """(calls served through mailboxes, launches of the service grid) so far."""
    calls, launches = C.c_int64(), C.c_int64()
    _check(_lib.lib().mgl_sw_service_stats(C.byref(calls), C.byref(launches)))
    return calls.value, launches.value


def backtrack_matrix(ref, alt, parameters=GATK_PARAMETERS, overhang_strategy=SWOverhangStrategy.SOFTCLIP):
    """The reference's logical backtrack matrix (calculateMatrix, sw.cpp:5-146) rebuilt on the GPU."""
    ref, alt = bytes(ref), bytes(alt)
    btr = np.zeros((len(ref) + 1, len(alt) + 1), dtype=np.int32)
    ez = _lib.Score()
    p = SWParameters(*parameters)
    rc = _lib.lib().mgl_sw_backtrack_matrix(ref, len(ref), alt, len(alt), p.match, p.mismatch, p.gap_open,
                                            p.gap_extend, int(overhang_strategy),
                                            btr.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ez))
    _check(rc)
    return btr, ScoreMax(ez.mqe, ez.mqe_t, ez.max, ez.max_t, ez.max_q, ez.seg_length)
