"""Host-side mirror of the reference's PairHMM operator over the C ABI of
include/mgl_pairhmm.h / libmgl_pairhmm_hip.so.

Same names and argument meaning as com.microsoft.mgl.pairhmm.MicrosoftPairHmm
(src/main/java/com/microsoft/mgl/pairhmm/MicrosoftPairHmm.java:35-112): ``load()``, ``initialize(args)``,
``computeLikelihoods(readDataArray, haplotypeDataArray, likelihoodArray)``, ``done()``; the holders are the
gatk-native-bindings ones (ReadDataHolder, HaplotypeDataHolder, PairHMMNativeArguments).  There is no CPU
fallback: without the HIP library or a GPU the calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGL_PAIRHMM_LIB") or os.path.join(HERE, "libmgl_pairhmm_hip.so")
CSRC = os.path.join(HERE, "csrc")

OK, ERR_BAD_ARG, _, ERR_NOMEM, ERR_DEVICE, ERR_UNSUPPORTED = range(6)

# every symbol include/mgl_pairhmm.h declares (tests check that the library exports them all)
SYMBOLS = (
    "mgl_pairhmm_version", "mgl_pairhmm_strerror", "mgl_pairhmm_last_error", "mgl_pairhmm_device_count",
    "mgl_pairhmm_max_haplotype_len", "mgl_pairhmm_ctx_create", "mgl_pairhmm_ctx_destroy", "mgl_pairhmm_initialize",
    "mgl_pairhmm_compute_likelihoods", "mgl_pairhmm_compute_pairs", "mgl_pairhmm_compute_pairs_device",
    "mgl_pairhmm_set_profiling", "mgl_pairhmm_get_timing", "mgl_pairhmm_set_stripe_rows",
)


class PairHmmError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().mgl_pairhmm_strerror(status).decode()
        super().__init__(f"mgl_pairhmm status {status}: {msg}" + (f" ({detail})" if detail else ""))


class Timing(C.Structure):
    _fields_ = [("float_ms", C.c_float), ("double_ms", C.c_float), ("cells", C.c_int64), ("rescued", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    L = C.CDLL(LIB_PATH, mode=os.RTLD_NOW)
    vp = C.c_void_p
    L.mgl_pairhmm_strerror.restype = C.c_char_p
    L.mgl_pairhmm_strerror.argtypes = [C.c_int]
    L.mgl_pairhmm_last_error.restype = C.c_char_p
    L.mgl_pairhmm_last_error.argtypes = [vp]
    L.mgl_pairhmm_max_haplotype_len.argtypes = [C.c_int]
    L.mgl_pairhmm_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mgl_pairhmm_ctx_destroy.argtypes = [vp]
    L.mgl_pairhmm_ctx_destroy.restype = None
    L.mgl_pairhmm_initialize.argtypes = [vp, C.c_int, C.c_int]
    L.mgl_pairhmm_compute_likelihoods.argtypes = [vp, vp, vp, vp, vp]
    L.mgl_pairhmm_compute_pairs.argtypes = [vp, C.c_int64, C.c_int64, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L.mgl_pairhmm_compute_pairs_device.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.mgl_pairhmm_set_profiling.argtypes = [vp, C.c_int]
    L.mgl_pairhmm_set_stripe_rows.argtypes = [vp, C.c_int]
    L.mgl_pairhmm_get_timing.argtypes = [vp, C.POINTER(Timing)]
    _lib = L
    return L


def _check(rc, ctx=None):
    if rc != OK:
        raise PairHmmError(rc, lib().mgl_pairhmm_last_error(ctx).decode() if ctx else "")


class ReadDataHolder:
    """org.broadinstitute.gatk.nativebindings.pairhmm.ReadDataHolder: bases + four quality tracks (phred
    values as bytes, already normalised by the caller)."""

    def __init__(self, readBases=b"", readQuals=b"", insertionGOP=b"", deletionGOP=b"", overallGCP=b""):
        self.readBases, self.readQuals = bytes(readBases), bytes(readQuals)
        self.insertionGOP, self.deletionGOP, self.overallGCP = bytes(insertionGOP), bytes(deletionGOP), bytes(overallGCP)


class HaplotypeDataHolder:
    def __init__(self, haplotypeBases=b""):
        self.haplotypeBases = bytes(haplotypeBases)


class PairHMMNativeArguments:
    def __init__(self, useDoublePrecision=False, maxNumberOfThreads=1):
        self.useDoublePrecision, self.maxNumberOfThreads = useDoublePrecision, maxNumberOfThreads


def pack_reads(reads):
    """[ReadDataHolder] -> (uint8 data, int64 offsets[n+1]): per read bases|quals|ins|del|gcp, the readsBuffer of
    MicrosoftPairHmm.java:88-95; read r starts at byte 5 * offsets[r]."""
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    parts = []
    for k, r in enumerate(reads):
        n = len(r.readBases)
        if not (len(r.readQuals) == len(r.insertionGOP) == len(r.deletionGOP) == len(r.overallGCP) == n):
            raise ValueError("read %d: the five tracks must have one length" % k)
        off[k + 1] = off[k] + n
        parts += [r.readBases, r.readQuals, r.insertionGOP, r.deletionGOP, r.overallGCP]
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), off


def pack_haps(haps):
    off = np.zeros(len(haps) + 1, dtype=np.int64)
    for k, h in enumerate(haps):
        off[k + 1] = off[k] + len(h.haplotypeBases)
    return np.frombuffer(b"".join(h.haplotypeBases for h in haps), dtype=np.uint8).copy(), off


class MicrosoftPairHmm:
    """Drop-in for the reference's PairHMMNativeBinding implementation (MicrosoftPairHmm.java)."""

    def __init__(self, device=0):
        self._device = device
        self._ctx = None

    def load(self, temp_dir=None):
        """MicrosoftPairHmm.java:35-37: True iff the native library (and here: a GPU context) is usable."""
        try:
            self._ensure()
            return True
        except (OSError, PairHmmError, subprocess.CalledProcessError):
            return False

    def _ensure(self):
        if self._ctx is None:
            ctx = C.c_void_p()
            _check(lib().mgl_pairhmm_ctx_create(self._device, C.byref(ctx)))
            self._ctx = ctx
        return self._ctx

    @property
    def ctx(self):
        return self._ensure()

    def initialize(self, args=None):
        """MicrosoftPairHmm.java:44-53: null args = float, one thread."""
        if args is None:
            args = PairHMMNativeArguments(False, 1)
        _check(lib().mgl_pairhmm_initialize(self._ensure(), int(bool(args.useDoublePrecision)), int(args.maxNumberOfThreads)))

    def computeLikelihoods(self, readDataArray, haplotypeDataArray, likelihoodArray):
        """MicrosoftPairHmm.java:62-112: fills likelihoodArray[r * nHaplotypes + h] (numpy float64 or list)."""
        n_reads, n_haps = len(readDataArray), len(haplotypeDataArray)
        lengths = np.array([n_reads] + [len(r.readBases) for r in readDataArray] + [n_haps] +
                           [len(h.haplotypeBases) for h in haplotypeDataArray], dtype=np.int32)
        reads, _ = pack_reads(readDataArray)
        haps, _ = pack_haps(haplotypeDataArray)
        out = np.zeros(n_reads * n_haps, dtype=np.float64)
        _check(lib().mgl_pairhmm_compute_likelihoods(self._ensure(), lengths.ctypes.data, reads.ctypes.data if len(reads) else None,
                                                     haps.ctypes.data if len(haps) else None, out.ctypes.data), self._ctx)
        likelihoodArray[:] = out if isinstance(likelihoodArray, np.ndarray) else out.tolist()
        return likelihoodArray

    def compute_pairs(self, reads, read_off, haps, hap_off, pair_read, pair_hap):
        """Flat pair list over packed host arrays (mgl_pairhmm_compute_pairs); returns float64[n_pairs]."""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        haps = np.ascontiguousarray(haps, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        hap_off = np.ascontiguousarray(hap_off, dtype=np.int64)
        pair_read = np.ascontiguousarray(pair_read, dtype=np.int32)
        pair_hap = np.ascontiguousarray(pair_hap, dtype=np.int32)
        out = np.zeros(len(pair_read), dtype=np.float64)
        _check(lib().mgl_pairhmm_compute_pairs(self._ensure(), len(pair_read), len(read_off) - 1, reads.ctypes.data,
                                               read_off.ctypes.data, len(hap_off) - 1, haps.ctypes.data, hap_off.ctypes.data,
                                               pair_read.ctypes.data, pair_hap.ctypes.data, out.ctypes.data), self._ctx)
        return out

    def compute_pairs_device(self, reads, read_off, haps, hap_off, pair_read, pair_hap, max_read_len, max_hap_len, out,
                             used_double=None, stream=None):
        """The same over torch CUDA tensors, enqueued on ``stream`` (default: torch's current stream); no sync."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(reads.device)
        _check(lib().mgl_pairhmm_compute_pairs_device(
            self._ensure(), C.c_void_p(stream.cuda_stream), pair_read.numel(), reads.data_ptr(), read_off.data_ptr(),
            haps.data_ptr(), hap_off.data_ptr(), pair_read.data_ptr(), pair_hap.data_ptr(), int(max_read_len), int(max_hap_len),
            out.data_ptr(), None if used_double is None else used_double.data_ptr()), self._ctx)
        return out

    def set_stripe_rows(self, rows):
        """Lanes per pair of the kernels: 0 = per batch, 16 (four pairs per wave), 21 (three), 32 (two) or 64 (one pair per wave)."""
        _check(lib().mgl_pairhmm_set_stripe_rows(self._ensure(), int(rows)))

    def set_profiling(self, on=True):
        _check(lib().mgl_pairhmm_set_profiling(self._ensure(), int(on)))

    def timing(self):
        t = Timing()
        _check(lib().mgl_pairhmm_get_timing(self._ensure(), C.byref(t)))
        return t

    def done(self):
        """MicrosoftPairHmm.java:117-119 (doneNative is a no-op there; here the GPU context is released)."""
        if self._ctx is not None:
            lib().mgl_pairhmm_ctx_destroy(self._ctx)
            self._ctx = None

    close = done
