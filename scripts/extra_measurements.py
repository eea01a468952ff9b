"""Measurements quoted in DESIGN.md that are not bench.py's `value`: the PCIe-inclusive rate of the
host-buffer entry (mgl_sw_align_batch: H2D of ASCII inputs, kernels, D2H of results) and variants."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tl, ql = 256, 150
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev, window=tl, read_len=ql)
t = b.targets.cpu().numpy(); q = b.queries.cpu().numpy()
toff = b.t_off.cpu().numpy(); qoff = b.q_off.cpu().numpy()
a = MicrosoftSmithWaterman(0)
a.set_workspace(8 << 30)
a.align_packed(t[: tl * 1000], toff[:1001], q[: ql * 1000], qoff[:1001], GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, 64)
for rep in range(2):
    t0 = time.perf_counter()
    res = a.align_packed(t, toff, q, qoff, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, 64)
    dt = time.perf_counter() - t0
    print(f"host-buffer API (PCIe inclusive, pageable host memory, incl. python list building): {n} pairs in {dt*1e3:.1f} ms = "
          f"{n*tl*ql/dt/1e9:.1f} GCUPS, {n/dt/1e6:.2f} M reads/s", flush=True)
# same call through ctypes only (no Python post-processing of CIGAR strings)
import ctypes as C
from mgl_amd import _lib
off = np.zeros(n, np.int32); sc = np.zeros((n, 6), np.int32); cg = np.zeros(n * 64, np.uint8); ln = np.zeros(n, np.int32)
L = _lib.lib()
for rep in range(2):
    t0 = time.perf_counter()
    rc = L.mgl_sw_align_batch(a.ctx, n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, 200, -150, 260, 11, 1,
                              off.ctypes.data, sc.ctypes.data, cg.ctypes.data, 64, ln.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"mgl_sw_align_batch only: {n} pairs in {dt*1e3:.1f} ms = {n*tl*ql/dt/1e9:.1f} GCUPS, {n/dt/1e6:.2f} M reads/s "
          f"({(t.nbytes+q.nbytes+toff.nbytes+qoff.nbytes)/1e6:.0f} MB in, {(off.nbytes+sc.nbytes+cg.nbytes+ln.nbytes)/1e6:.0f} MB out)", flush=True)

# score-only mode (MGL_SW_FLAG_SCORE_ONLY): the packed kernel without the traceback flags, no path walk
b.run(a); torch.cuda.synchronize()
full = b.scores.clone()
for rep in range(2):
    t0 = time.perf_counter()
    b.run(a, score_only=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"score-only device batch: {n} pairs in {dt*1e3:.1f} ms = {n*tl*ql/dt/1e9:.1f} GCUPS, {n/dt/1e6:.2f} M reads/s", flush=True)
assert torch.equal(full, b.scores)
