"""BASELINE.json configs[3] shape: ONT-style 10 kb x 10 kb pairs (5 % sub / 5 % ins / 5 % del), int32 scores,
full matrix, on-device traceback and full CIGAR.  The traceback needs tl*ql/2 bytes per pair (50 MB), so the
number of pairs in flight -- hence GPU occupancy -- is set by the workspace."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from mgl_amd import device_batch, synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ws_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 32
length = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
check = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rng = synth.rng_for(11)
base = [synth.ont_pair(rng, length) for _ in range(min(n, 32))]
ts = [base[k % len(base)][0].tobytes() for k in range(n)]
qs = [base[k % len(base)][1].tobytes() for k in range(n)]
from mgl_amd.smithwaterman import concat
td, toff = concat(ts); qd, qoff = concat(qs)
b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=2 * (length + 2000))
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(ws_gib * (1 << 30)))
cells = b.cells
b.run(a); torch.cuda.synchronize()
a.set_profiling(1)
t0 = time.perf_counter()
b.run(a); torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm = a.timing()
print(f"{n} pairs of ~{length} x {length}: {dt*1e3:.1f} ms = {cells/dt/1e9:.1f} GCUPS "
      f"(fill {tm.dp_ms:.1f} ms in {tm.dp_launches} launches, traceback {tm.tb_ms:.1f} ms, "
      f"traceback workspace {tm.tb_bytes/2**30:.1f} GiB per pass)", flush=True)
assert int((b.status != 0).sum()) == 0
if check:
    import oracle_lib as ol
    idx = list(range(min(check, len(base))))
    cg = b.cigar_strings(idx)
    for k in idx:
        o = ol.oracle_align(ts[k], qs[k], (200, -150, 260, 11), ol.SOFTCLIP)
        assert (int(b.offsets[k]), cg[k], tuple(int(x) for x in b.scores[k])) == (o["offset"], o["cigar"], o["score"]), k
    print(f"checked {len(idx)} pairs against the oracle: identical", flush=True)
