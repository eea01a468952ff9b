"""Timing probe of the two-pairs-per-lane kernel: full call vs score-only (no traceback flags / stores), device resident.
Usage: python scripts/lane_probe.py [pairs] [tl] [ql]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_097_152
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ql = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev, window=tl, read_len=ql)
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(os.environ.get("WS_GIB", "8")) << 30)   # the persistent grid: 4.2 GB of regions for the whole chip at 256 x 150 (tl = 1000: WS_GIB=24)
a.set_lane_kernel(int(os.environ.get("LANE_MODE", "2")))
for so in ((False,) if os.environ.get('FULL_ONLY') else (False, True)):
    b.run(a, score_only=so); torch.cuda.synchronize()
    a.set_profiling(1)
    t0 = time.perf_counter()
    reps = int(os.environ.get("REPS", "3"))
    for _ in range(reps):
        b.run(a, score_only=so)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tm = a.timing()
    a.set_profiling(0)
    print(f"{'score-only' if so else 'full      '}: {n} pairs {tl}x{ql}: {dt*1e3:.2f} ms per call = {n*tl*ql/dt/1e9:.0f} GCUPS; fill kernel "
          f"{a.fill_kernel_name(tm)} {tm.dp_ms:.2f} ms in {tm.dp_launches} launch(es) = {n*tl*ql/tm.dp_ms/1e6:.0f} GCUPS", flush=True)
