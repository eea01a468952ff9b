"""Per-call latency of the JNI-shaped PairHMM entry (one active region per call, the way GATK drives
computeLikelihoodsNative): mgl_pairhmm_compute_likelihoods on host buffers, region sizes 10 x 4 .. 200 x 16."""
import ctypes as C, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from mgl_amd import pairhmm

rng = np.random.default_rng(3)
hmm = pairhmm.MicrosoftPairHmm(0)
hmm.initialize(None)
L = pairhmm.lib()
for n_reads, n_haps in ((10, 4), (50, 8), (100, 8), (200, 16)):
    base = rng.choice(list(b"ACGT"), size=400).astype(np.uint8)
    haps = [pairhmm.HaplotypeDataHolder(base[: 300 + 2 * h].tobytes()) for h in range(n_haps)]
    reads = []
    for _ in range(n_reads):
        s = int(rng.integers(0, 150))
        r = base[s:s + 150].copy()
        reads.append(pairhmm.ReadDataHolder(r.tobytes(), bytes([30] * 150), bytes([45] * 150), bytes([45] * 150), bytes([10] * 150)))
    lengths = np.array([n_reads] + [150] * n_reads + [n_haps] + [len(h.haplotypeBases) for h in haps], dtype=np.int32)
    rd, _ = pairhmm.pack_reads(reads)
    hd, _ = pairhmm.pack_haps(haps)
    out = np.zeros(n_reads * n_haps)
    call = lambda: L.mgl_pairhmm_compute_likelihoods(hmm.ctx, lengths.ctypes.data, rd.ctypes.data, hd.ctypes.data, out.ctypes.data)
    for _ in range(20):
        assert call() == 0
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    cells = sum(150 * len(h.haplotypeBases) for h in haps) * n_reads
    print(f"{n_reads} reads x {n_haps} haplotypes = {n_reads*n_haps} pairs: {dt*1e6:.0f} us per call = {cells/dt/1e9:.1f} GCUPS", flush=True)
hmm.done()
