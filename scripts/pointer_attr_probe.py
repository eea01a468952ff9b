"""What does the runtime say about a PAGEABLE numpy array before and after a framework's copy from it (hipPointerGetAttributes)?  If a
pageable array reads as page-locked host memory for a while after a copy, `is_registered()`'s fallback (sw_capi.cpp) would take it for
the caller's own pinned memory.  python scripts/pointer_attr_probe.py"""
import ctypes as C, sys, os
import numpy as np
import torch

torch.cuda.init()
hip = None
for l in open("/proc/self/maps"):
    p = l.split()[-1]
    if "libamdhip64" in p:
        hip = C.CDLL(p); break


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p), ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]


hip.hipPointerGetAttributes.argtypes = [C.POINTER(Attr), C.c_void_p]
hip.hipGetLastError.restype = C.c_int


def say(label, ptr):
    a = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(a), ptr)
    hip.hipGetLastError()
    print(f"{label:58s} rc {rc:3d} type {a.type} devicePointer {a.devicePointer} hostPointer {a.hostPointer} flags {a.allocationFlags}", flush=True)


dev = torch.device("cuda", 0)
for size in (200_000, 3_200_000, 40_000_000):
    x = np.full(size, 7, np.uint8)
    say(f"pageable numpy array of {size} bytes, fresh", x.ctypes.data)
    t = torch.from_numpy(x).to(dev)
    say("  ... right after torch copied it to the device", x.ctypes.data)
    torch.cuda.synchronize()
    say("  ... after a device synchronisation", x.ctypes.data)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t2 = torch.from_numpy(x).to(dev, non_blocking=True)
    say("  ... right after a non_blocking copy on a side stream", x.ctypes.data)
    s.synchronize()
    say("  ... after that stream's synchronisation", x.ctypes.data)
p = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
say("torch pinned memory", p.data_ptr())
d = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
say("device memory", d.data_ptr())
