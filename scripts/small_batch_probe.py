"""Latency anatomy of a small batch (the coalescing front-end's unit of work): kernel times from HIP events vs wall time."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from mgl_amd import synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g, ws, reads = synth.window_batch(7, n, genome_len=1 << 20)
ts = [g[w:w + 256].tobytes() for w in ws]
qs = [r.tobytes() for r in reads]
for coop, rows in ((1, 16), (1, 64), (4, 0), (2, 0)):
    a = MicrosoftSmithWaterman(0)
    a.set_cooperative(coop)
    a.set_precision(32)
    if rows:
        a.set_stripe_rows(rows)
    for prof in (0, 1):
        a.set_profiling(prof)
        for _ in range(20):
            a.align_batch(ts, qs, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            a.align_batch(ts, qs, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
        dt = (time.perf_counter() - t0) / reps
        tm = a.timing()
        print(f"n={n} cooperative={coop} rows={rows} profiling={prof}: {dt*1e6:.1f} us per call (python incl.), fill {tm.dp_ms*1e3:.1f} us, traceback {tm.tb_ms*1e3:.1f} us", flush=True)
    a.close()
