"""Regions per second of the JNI-shaped PairHMM entry when several host threads each drive their own context (what the
per-thread contexts of pairhmm_jni_exports.cpp give GATK): 100 reads x 8 haplotypes per call."""
import ctypes as C, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from mgl_amd import pairhmm

rng = np.random.default_rng(3)
base = rng.choice(list(b"ACGT"), size=400).astype(np.uint8)
haps = [pairhmm.HaplotypeDataHolder(base[: 300 + 2 * h].tobytes()) for h in range(8)]
reads = []
for _ in range(100):
    s = int(rng.integers(0, 150))
    reads.append(pairhmm.ReadDataHolder(base[s:s + 150].tobytes(), bytes([30] * 150), bytes([45] * 150), bytes([45] * 150), bytes([10] * 150)))
lengths = np.array([100] + [150] * 100 + [8] + [len(h.haplotypeBases) for h in haps], dtype=np.int32)
rd, _ = pairhmm.pack_reads(reads)
hd, _ = pairhmm.pack_haps(haps)
L = pairhmm.lib()
calls = 400

def worker(out):
    hmm = pairhmm.MicrosoftPairHmm(0)
    hmm.initialize(None)
    res = np.zeros(800)
    for _ in range(calls):
        assert L.mgl_pairhmm_compute_likelihoods(hmm.ctx, lengths.ctypes.data, rd.ctypes.data, hd.ctypes.data, res.ctypes.data) == 0
    out.append(res.copy())
    hmm.done()

for T in (1, 2, 4, 8, 16):
    outs = []
    th = [threading.Thread(target=worker, args=(outs,)) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    assert all((o == outs[0]).all() for o in outs)
    cells = sum(150 * len(h.haplotypeBases) for h in haps) * 100
    print(f"{T} threads: {T*calls/dt:.0f} regions/s = {T*calls*cells/dt/1e9:.0f} GCUPS (incl. context creation)", flush=True)
