"""One-pair-per-call throughput (the GATK calling pattern) with and without the coalescing front-end."""
import ctypes as C, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from mgl_amd import _lib, synth

L = _lib.lib()
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 400
g, ws, reads = synth.window_batch(7, nthreads * per_thread, genome_len=1 << 20)
ts = [g[w:w + 256].tobytes() for w in ws]
qs = [r.tobytes() for r in reads]

def worker(lo):
    buf = C.create_string_buffer(512)
    ln, off = C.c_int(), C.c_int()
    for k in range(lo, len(ts), nthreads):
        rc = L.mgl_sw_align(ts[k], 256, qs[k], 150, 200, -150, 260, 11, 1, buf, 512, C.byref(ln), C.byref(off), None)
        assert rc == 0

def run(label):
    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    n = len(ts)
    print(f"{label:34s} {nthreads} threads: {n} pairs in {dt*1e3:8.1f} ms = {n/dt:10.0f} pairs/s = {n*256*150/dt/1e9:7.2f} GCUPS", flush=True)

worker(0)  # warm (creates the thread-local context of this thread only)
run("direct (one launch per pair)")
for us in (200, 1000):
    L.mgl_sw_set_coalescing(4096, us)
    run(f"coalesced (max_wait {us} us)")
    b, p = C.c_int64(), C.c_int64()
    L.mgl_sw_coalescing_stats(C.byref(b), C.byref(p))
    print(f"    batches so far {b.value}, pairs {p.value}, avg batch {p.value / max(1, b.value):.1f}")
L.mgl_sw_set_coalescing(0, 0)
