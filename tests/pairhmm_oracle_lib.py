"""ctypes access to oracle/libpairhmm_oracle.so (TEST INFRASTRUCTURE: the PairHMM checker) and the reference's
known-answer file (tests/golden/pairhmm-testdata.txt, a data fixture copied from the reference's test resources)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libpairhmm_oracle.so")
TESTDATA = os.path.join(ROOT, "tests", "golden", "pairhmm-testdata.txt")


class Read(C.Structure):
    _fields_ = [("len", C.c_int32), ("bases", C.c_char_p), ("qual", C.c_char_p), ("ins", C.c_char_p), ("del_", C.c_char_p),
                ("gcp", C.c_char_p)]


_lib = None


def oracle():
    global _lib
    if _lib is None:
        src = [os.path.join(ROOT, "oracle", f) for f in ("pairhmm_oracle.c", "pairhmm_oracle_impl.h", "pairhmm_oracle.h")]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = C.CDLL(LIB)
        L.pho_log10_likelihood.restype = C.c_double
        L.pho_log10_likelihood.argtypes = [C.POINTER(Read), C.c_char_p, C.c_int32, C.c_int, C.POINTER(C.c_int)]
        L.pho_forward_float.restype = C.c_float
        L.pho_forward_float.argtypes = [C.POINTER(Read), C.c_char_p, C.c_int32]
        L.pho_forward_double.restype = C.c_double
        L.pho_forward_double.argtypes = [C.POINTER(Read), C.c_char_p, C.c_int32]
        L.pho_compute_likelihoods.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.pho_compute_pairs.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def log10_likelihood(hap, bases, qual, ins, dele, gcp, use_double=False):
    r = Read(len(bases), bytes(bases), bytes(qual), bytes(ins), bytes(dele), bytes(gcp))
    ud = C.c_int()
    v = oracle().pho_log10_likelihood(C.byref(r), bytes(hap), len(hap), int(use_double), C.byref(ud))
    return v, bool(ud.value)


def compute_pairs(reads, read_off, haps, hap_off, pair_read, pair_hap, use_double=False, nthreads=1):
    out = np.zeros(len(pair_read), dtype=np.float64)
    used = np.zeros(len(pair_read), dtype=np.int32)
    rc = oracle().pho_compute_pairs(len(pair_read), reads.ctypes.data, read_off.ctypes.data, haps.ctypes.data, hap_off.ctypes.data,
                                    pair_read.ctypes.data, pair_hap.ctypes.data, out.ctypes.data, int(use_double), nthreads,
                                    used.ctypes.data)
    assert rc == 0
    return out, used


def normalize(scores, minimum=0):
    """MicrosoftPairHmmUnitTest.java:119-129: phred+33 text -> phred bytes, floor ``minimum``."""
    return bytes(max(b - 33, minimum) for b in scores)


def testdata():
    """Yield (hap, bases, qual, ins, del, gcp, expected) with the qualities normalised as dataFileTest does (:88-93)."""
    with open(TESTDATA, "rb") as f:
        for line in f:
            if line.startswith(b"#") or not line.strip():
                continue
            hap, rb, q, i, d, c, exp = line.split()
            yield hap, rb, normalize(q, 6), normalize(i), normalize(d), normalize(c), float(exp)
